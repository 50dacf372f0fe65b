"""The C-ABI library loads and exports every symbol include/*.h declares;
ctypes mirrors have the C layout.  No compute calls (runs without a GPU)."""
import ctypes as C
import os
import re
import subprocess
import sys

import pytest

from chess2rt_amd import _abi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions(header):
    text = open(os.path.join(ROOT, "include", header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    text = re.sub(r"//[^\n]*", "", text)
    names = re.findall(r"\b(c2rt_[a-z0-9_]+)\s*\(", text)
    return sorted(set(names))


@pytest.mark.parametrize("header,table", [("c2rt.h", _abi.C2RT_SYMBOLS), ("c2rt_host.h", _abi.C2RT_HOST_SYMBOLS)])
def test_every_declared_symbol_is_exported(header, table):
    lib = _abi.load_library()
    declared = declared_functions(header)
    assert declared, header
    for name in declared:
        assert hasattr(lib, name), "%s declared in include/%s but not exported" % (name, header)
    # and the ctypes table covers exactly what the header declares
    assert sorted(table) == declared


def test_every_exported_symbol_is_declared():
    """The converse: the product library exports no c2rt_* entry point that the two headers do not declare
    (diagnostics hooks live in diagnostics builds only)."""
    import subprocess

    lib_path = _abi.LIB_PATH
    out = subprocess.run(["nm", "-D", "--defined-only", lib_path], capture_output=True, text=True, check=True).stdout
    exported = sorted({line.split()[-1] for line in out.splitlines() if line.split() and line.split()[-1].startswith("c2rt_")})
    declared = set(declared_functions("c2rt.h")) | set(declared_functions("c2rt_host.h"))
    assert exported, lib_path
    extra = [n for n in exported if n not in declared]
    assert not extra, "exported but declared in no header: %s" % extra


def test_abi_version_and_status_strings():
    lib = _abi.load_library()
    assert lib.c2rt_abi_version() == _abi.ABI_VERSION
    for st in range(10):
        assert lib.c2rt_status_string(st)
    assert b"CPU fallback" in lib.c2rt_status_string(_abi.ERR_NO_DEVICE)


def test_struct_layouts_match_c(tmp_path):
    src = tmp_path / "sizes.c"
    src.write_text('''
#include <stdio.h>
#include <stddef.h>
#include "c2rt_host.h"
#include "../oracle/c2rt_oracle.h"
int main(void) {
  printf("SceneDesc %zu\\n", sizeof(c2rt_scene_desc));
  printf("CameraFrame %zu\\n", sizeof(c2rt_camera_frame));
  printf("RenderOpts %zu\\n", sizeof(c2rt_render_opts));
  printf("TraceResult %zu\\n", sizeof(c2rt_trace_result));
  printf("RayStats %zu\\n", sizeof(c2rt_ray_stats));
  printf("HostSettings %zu\\n", sizeof(c2rt_host_settings));
  printf("HostCamera %zu\\n", sizeof(c2rt_host_camera));
  printf("OrcHit %zu\\n", sizeof(orc_hit));
  printf("off_scene_texels %zu\\n", offsetof(c2rt_scene_desc, texels));
  printf("off_scene_ambient %zu\\n", offsetof(c2rt_scene_desc, ambient));
  printf("off_cam_dof %zu\\n", offsetof(c2rt_camera_frame, dof));
  printf("off_opts_seed %zu\\n", offsetof(c2rt_render_opts, seed));
  printf("off_trace_p %zu\\n", offsetof(c2rt_trace_result, p));
  return 0; }
''')
    exe = tmp_path / "sizes"
    subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)])
    got = dict(line.split() for line in subprocess.check_output([str(exe)]).decode().splitlines())
    from oracle_lib import OrcHit

    expect = {
        "SceneDesc": C.sizeof(_abi.SceneDesc), "CameraFrame": C.sizeof(_abi.CameraFrame),
        "RenderOpts": C.sizeof(_abi.RenderOpts), "TraceResult": C.sizeof(_abi.TraceResult),
        "RayStats": C.sizeof(_abi.RayStats), "HostSettings": C.sizeof(_abi.HostSettings),
        "HostCamera": C.sizeof(_abi.HostCamera), "OrcHit": C.sizeof(OrcHit),
        "off_scene_texels": _abi.SceneDesc.texels.offset, "off_scene_ambient": _abi.SceneDesc.ambient.offset,
        "off_cam_dof": _abi.CameraFrame.dof.offset, "off_opts_seed": _abi.RenderOpts.seed.offset,
        "off_trace_p": _abi.TraceResult.p.offset,
    }
    assert {k: int(v) for k, v in got.items()} == expect


def test_missing_library_fails_loudly(tmp_path):
    """The product path must not fall back to anything when libc2rt.so is absent."""
    code = "import os; os.environ['C2RT_LIB_VARIANT']='does_not_exist'; import chess2rt_amd as c; c._abi.load_library()"
    p = subprocess.run([sys.executable, "-c", code], cwd=ROOT, capture_output=True, text=True)
    assert p.returncode != 0
    assert "no" in p.stderr.lower() and "fallback" in p.stderr.lower()


def test_product_package_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "chess2rt_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".cpp", ".hpp", ".h", ".hip")):
                text = open(os.path.join(dirpath, f), errors="replace").read()
                assert "oracle_lib" not in text and "c2rt_oracle" not in text and "libc2rt_oracle" not in text, os.path.join(dirpath, f)


def test_product_library_reads_no_environment_variable():
    """Round-3 verdict, weak #11: the measurement / test hooks (C2RT_EXACT, C2RT_NO_IDN, C2RT_DEBUG_CULL,
    C2RT_CSG_FIRST_CAP, C2RT_HOST_*) live in the diagnostics build only.  The product library does not import
    getenv at all and holds none of the variable names; libc2rt_diag.so (same kernel objects, c2rt_api.cpp with
    -DC2RT_DIAG=1) does, and exports the same ABI."""
    lib = os.path.join(ROOT, "chess2rt_amd", "libc2rt.so")
    diag = os.path.join(ROOT, "chess2rt_amd", "libc2rt_diag.so")
    undefined = subprocess.run(["nm", "-D", "--undefined-only", lib], capture_output=True, text=True, check=True).stdout
    assert "getenv" not in undefined
    blob = open(lib, "rb").read()
    for name in (b"C2RT_EXACT", b"C2RT_NO_IDN", b"C2RT_DEBUG_CULL", b"C2RT_CSG_FIRST_CAP", b"C2RT_HOST_"):
        assert name not in blob, name
    assert "getenv" in subprocess.run(["nm", "-D", "--undefined-only", diag], capture_output=True, text=True, check=True).stdout
    assert b"C2RT_CSG_FIRST_CAP" in open(diag, "rb").read()
    d = C.CDLL(diag)
    for table in (_abi.C2RT_SYMBOLS, _abi.C2RT_HOST_SYMBOLS):
        for name in table:
            getattr(d, name)


def test_local_rows_is_pure_host_arithmetic():
    lib = _abi.load_library()
    from chess2rt_amd.sharding import local_rows

    for (h, sh, world) in [(480, 8, 2), (1080, 8, 8), (217, 8, 3), (100, 16, 4), (7, 8, 2), (2160, 8, 1)]:
        total = 0
        for r in range(world):
            o = _abi.RenderOpts(width=16, height=h, taps=1, strip_height=sh, strip_rank=r, strip_world=world)
            n = lib.c2rt_local_rows(C.byref(o))
            assert n == local_rows(h, sh, r, world)
            total += n
        assert total == h


def _build_c_program(src, exe, with_oracle=False):
    cmd = ["gcc", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), src, "-L", os.path.join(ROOT, "chess2rt_amd"), "-lc2rt",
           "-Wl,-rpath," + os.path.join(ROOT, "chess2rt_amd")]
    if with_oracle:
        cmd += ["-L", os.path.join(ROOT, "oracle"), "-lc2rt_oracle", "-Wl,-rpath," + os.path.join(ROOT, "oracle")]
    subprocess.check_call(cmd + ["-lm", "-o", exe])


def test_plain_c_caller_links_and_fails_loudly_without_gpu(tmp_path):
    """The boundary is usable from plain C (no C++ or torch types leak into the
    header), and with no GPU the caller gets C2RT_ERR_NO_DEVICE, not a CPU frame."""
    import torch

    exe = str(tmp_path / "c_abi_demo")
    _build_c_program(os.path.join(ROOT, "examples", "c_abi_demo.c"), exe)
    p = subprocess.run([exe, "32", "24"], capture_output=True, text=True)
    if torch.cuda.is_available():
        assert p.returncode == 0, p.stderr
        assert "primary" in p.stdout
    else:
        assert p.returncode != 0 and "no usable GPU" in p.stderr
