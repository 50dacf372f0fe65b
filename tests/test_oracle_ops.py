"""The counting build of the oracle (oracle/libc2rt_oracle_count.so, -DORC_COUNT_OPS): the
algorithmic floating-point operation count bench.py's roofline.flops rests on."""
import os

import numpy as np

import chess2rt_amd as c2
import oracle_lib as orc
from golden_configs import SCENES, load_config


def test_counting_build_renders_the_same_frame_and_counts_deterministically():
    scene, cam, opts = load_config("lecture5_333x217_t4")
    ref = orc.render_frame(scene.desc, cam, opts, 2)
    ops1, img1 = orc.op_counts(scene.desc, cam, opts, 1)
    ops4, img4 = orc.op_counts(scene.desc, cam, opts, 4)
    assert np.array_equal(img1, ref) and np.array_equal(img4, ref)
    assert ops1 == ops4                       # thread-local tallies, flushed once per worker
    assert ops1["primary_rays"] == 333 * 217 * 4
    assert ops1["fp64"] == ops1["dadd"] + ops1["dmul"] + ops1["ddiv"] + ops1["dsqrt"] + ops1["dlibm"]
    # lecture5 has spheres, a cube, a CsgDiff, Phong and bitmap textures: every class of operation occurs
    assert all(ops1[k] > 0 for k in ("dadd", "dmul", "ddiv", "dsqrt", "dlibm", "fadd", "fmul", "fdiv"))


def test_sky_rays_cost_exactly_what_the_source_says():
    """lecture4 with the camera pitched up: every ray misses the one Plane node.  Per ray the
    reference executes getScreenRay (rt/camera.d:123-147), Node.intersect's transform of the ray
    (rt/node.d:28-36) and Plane.intersect's first rejection (rt/geometry.d:33-34, no arithmetic):
      getScreenRay: 2 vector subs + 2 divs + 2 vector*scalar + 2 vector adds + (target - pos) + normalize
      Node.intersect: (orig - offset), 2 row-vector x matrix products, magnitude, dist *= len, normalize
    = 39 fp64 add/sub, 40 mul, 4 div, 3 sqrt, no libm, no colour arithmetic."""
    scene = c2.parseSceneFromFile(os.path.join(SCENES, "lecture4.sdl"))
    scene.setFrameSize(64, 48)
    scene.setAA(False)
    cam = scene.camera
    cam.pitch = 80.0
    scene.camera = cam
    frame = scene.beginFrame()
    opts = scene.renderOpts()
    ops, img = orc.op_counts(scene.desc, frame, opts, 2)
    n = 64 * 48
    assert not img.any() and ops["shadow_rays"] == 0 and ops["primary_rays"] == n
    assert (ops["dadd"], ops["dmul"], ops["ddiv"], ops["dsqrt"], ops["dlibm"]) == (39 * n, 40 * n, 4 * n, 3 * n, 0)
    assert ops["fp32"] == 0


def test_plain_build_reports_that_it_does_not_count():
    import ctypes as C

    L = orc.lib()
    L.orc_op_counts_take.argtypes = [C.POINTER(orc.OpCounts)]
    L.orc_op_counts_take.restype = C.c_int
    oc = orc.OpCounts()
    assert L.orc_op_counts_take(C.byref(oc)) == 0 and oc.dadd == 0


def test_x87_emulation_matches_the_compilers_long_double(tmp_path):
    """chess2rt_amd/csrc/x87.h — the integer emulation the device uses for Sphere.intersect's u, v
    (`PI` is an 80-bit real in the reference, rt/geometry.d:119-120) — against this host's own x87
    `long double` arithmetic: special values, every binade with tie patterns, 4 M random operands."""
    import subprocess

    from golden_configs import ROOT

    exe = str(tmp_path / "x87_check")
    subprocess.check_call(["gcc", "-O2", "-Wall", "-Werror", os.path.join(ROOT, "tests", "x87_check.c"), "-lm", "-o", exe])
    p = subprocess.run([exe, "1000000"], capture_output=True, text=True, timeout=120)
    assert p.returncode == 0 and "mismatches 0" in p.stdout, p.stdout + p.stderr


def test_the_csg_hit_cap_takes_no_part_in_ordinary_frames(tmp_path):
    """findAllIntersections is capped at C2RT_MAX_CSG_HITS = 8 hits per CsgOp child (the reference loops
    `while (true)`).  With the cap lifted to 64 the oracle renders the same frames — golden CSG configs and
    40 random CSG scenes up to depth 4 — and no hit list ever reaches even the cap of 8."""
    import shutil

    from scene_fuzz import random_scene_sdl

    L = orc.lib()
    shutil.copy(os.path.join(SCENES, "floor.bmp"), tmp_path / "floor.bmp")
    cases = [load_config(n) for n in ("csg_stress_320x240_t1", "csg_corner_256x192_t1", "lecture5_333x217_t4")]
    for seed in range(40):
        path = tmp_path / ("f%d.sdl" % seed)
        path.write_text(random_scene_sdl(seed, max_depth=4))
        sc = c2.parseSceneFromFile(str(path))
        cases.append((sc, sc.beginFrame(), sc.renderOpts()))
    try:
        for scene, cam, opts in cases:
            L.orc_take_csg_truncations()
            a = orc.render_frame(scene.desc, cam, opts, 4)
            assert L.orc_take_csg_truncations() == 0
            L.orc_set_csg_hit_cap(64)
            b = orc.render_frame(scene.desc, cam, opts, 4)
            assert L.orc_take_csg_truncations() == 0
            L.orc_set_csg_hit_cap(8)
            assert np.array_equal(a, b, equal_nan=True)
    finally:
        L.orc_set_csg_hit_cap(8)
