"""ctypes binding of the CPU oracle (oracle/libc2rt_oracle.so).

TEST INFRASTRUCTURE: only tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg may import this.  The product package never does.
"""
import ctypes as C
import os

import numpy as np

from chess2rt_amd._abi import CameraFrame, RayStats, RenderOpts, SceneDesc, TraceResult

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_PATH = os.path.join(ROOT, "oracle", "libc2rt_oracle.so")


class OrcHit(C.Structure):
    _fields_ = [
        ("p", C.c_double * 3), ("normal", C.c_double * 3),
        ("dist", C.c_double), ("u", C.c_double), ("v", C.c_double),
        ("g", C.c_int32),
        ("dNdx", C.c_double * 3), ("dNdy", C.c_double * 3),
    ]


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(ORACLE_PATH):
            raise RuntimeError("oracle not built: run `make oracle/libc2rt_oracle.so`")
        L = C.CDLL(ORACLE_PATH)
        d3 = C.POINTER(C.c_double)
        L.orc_camera_begin_frame.argtypes = [d3, C.c_double, C.c_double, C.c_double, C.c_double, C.c_uint32, C.c_uint32, C.POINTER(CameraFrame)]
        L.orc_camera_begin_frame.restype = None
        for n in ("orc_transform_reset",):
            getattr(L, n).argtypes = [d3]
            getattr(L, n).restype = None
        L.orc_transform_scale.argtypes = [d3, C.c_double, C.c_double, C.c_double]
        L.orc_transform_scale.restype = None
        L.orc_transform_rotate.argtypes = [d3, C.c_double, C.c_double, C.c_double]
        L.orc_transform_rotate.restype = None
        L.orc_transform_translate.argtypes = [d3, d3]
        L.orc_transform_translate.restype = None
        L.orc_bmp_decode.argtypes = [C.c_void_p, C.c_size_t, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.POINTER(C.POINTER(C.c_uint32))]
        L.orc_bmp_decode.restype = C.POINTER(C.c_float)
        L.orc_texture_gamma.argtypes = [C.c_void_p, C.c_size_t, C.c_float]
        L.orc_texture_gamma.restype = None
        L.orc_free.argtypes = [C.c_void_p]
        L.orc_free.restype = None
        L.orc_render_frame.argtypes = [C.POINTER(SceneDesc), C.POINTER(CameraFrame), C.POINTER(RenderOpts), C.c_void_p, C.c_uint32, C.POINTER(RayStats)]
        L.orc_render_frame.restype = C.c_int
        L.orc_render_pixel.argtypes = [C.POINTER(SceneDesc), C.POINTER(CameraFrame), C.POINTER(RenderOpts), C.c_int, C.c_int, C.POINTER(TraceResult)]
        L.orc_render_pixel.restype = C.c_int
        L.orc_geom_intersect.argtypes = [C.POINTER(SceneDesc), C.c_int32, d3, d3, C.POINTER(OrcHit)]
        L.orc_geom_intersect.restype = C.c_int
        L.orc_geom_is_inside.argtypes = [C.POINTER(SceneDesc), C.c_int32, d3]
        L.orc_geom_is_inside.restype = C.c_int
        L.orc_node_intersect.argtypes = [C.POINTER(SceneDesc), C.c_int32, d3, d3, C.POINTER(OrcHit)]
        L.orc_node_intersect.restype = C.c_int
        L.orc_tex_color.argtypes = [C.POINTER(SceneDesc), C.c_int32, C.c_double, C.c_double, C.POINTER(C.c_float)]
        L.orc_tex_color.restype = None
        L.orc_screen_ray.argtypes = [C.POINTER(CameraFrame), C.c_double, C.c_double, d3, d3]
        L.orc_screen_ray.restype = None
        L.orc_test_visibility.argtypes = [C.POINTER(SceneDesc), d3, d3]
        L.orc_test_visibility.restype = C.c_int
        L.orc_shell_sort_hits.argtypes = [C.POINTER(OrcHit), C.c_size_t]
        L.orc_shell_sort_hits.restype = None
        L.orc_color_to_rgb32.argtypes = [C.POINTER(C.c_float)]
        L.orc_color_to_rgb32.restype = C.c_uint32
        L.orc_set_csg_hit_cap.argtypes = [C.c_uint]
        L.orc_set_csg_hit_cap.restype = None
        L.orc_take_csg_truncations.argtypes = []
        L.orc_take_csg_truncations.restype = C.c_ulonglong
        L.orc_rng_uniform.argtypes = [C.c_uint64, C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32]
        L.orc_rng_uniform.restype = C.c_double
        _lib = L
    return _lib


class OpCounts(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in ("dadd", "dmul", "ddiv", "dsqrt", "dlibm", "fadd", "fmul", "fdiv")]


_count_lib = None


def op_counts(desc, cam, opts, n_threads=0):
    """Algorithmic floating-point operations of one oracle frame (the counting build,
    oracle/libc2rt_oracle_count.so): dict of fp64 add/mul/div/sqrt/libm and fp32 add/mul/div tallies
    plus "fp64" = their fp64 sum (div, sqrt and libm calls counted as one each)."""
    global _count_lib
    if _count_lib is None:
        path = os.path.join(ROOT, "oracle", "libc2rt_oracle_count.so")
        if not os.path.exists(path):
            raise RuntimeError("counting oracle not built: run `make oracle/libc2rt_oracle_count.so`")
        L = C.CDLL(path)
        L.orc_render_frame.argtypes = [C.POINTER(SceneDesc), C.POINTER(CameraFrame), C.POINTER(RenderOpts), C.c_void_p, C.c_uint32, C.POINTER(RayStats)]
        L.orc_render_frame.restype = C.c_int
        L.orc_op_counts_take.argtypes = [C.POINTER(OpCounts)]
        L.orc_op_counts_take.restype = C.c_int
        _count_lib = L
    L = _count_lib
    out = np.zeros((local_rows(opts), opts.width, 3), dtype=np.float32)
    st = RayStats()
    oc = OpCounts()
    L.orc_op_counts_take(C.byref(oc))  # clear
    rc = L.orc_render_frame(desc, C.byref(cam), C.byref(opts), out.ctypes.data_as(C.c_void_p), n_threads, C.byref(st))
    if rc != 0:
        raise RuntimeError("orc_render_frame (counting build) status %d" % rc)
    if L.orc_op_counts_take(C.byref(oc)) != 1:
        raise RuntimeError("libc2rt_oracle_count.so was built without -DORC_COUNT_OPS")
    d = {n: int(getattr(oc, n)) for n, _ in OpCounts._fields_}
    d["fp64"] = d["dadd"] + d["dmul"] + d["ddiv"] + d["dsqrt"] + d["dlibm"]
    d["fp32"] = d["fadd"] + d["fmul"] + d["fdiv"]
    d["primary_rays"], d["shadow_rays"] = int(st.primary_rays), int(st.shadow_rays)
    return d, out


def vec3(x, y, z):
    return (C.c_double * 3)(x, y, z)


def local_rows(opts):
    if opts.strip_world <= 1:
        return opts.height
    sh = opts.strip_height or 1
    return sum(1 for y in range(opts.height) if (y // sh) % opts.strip_world == opts.strip_rank)


def render_frame(desc, cam, opts, n_threads=0, stats=None):
    """Oracle frame: (local_rows, W, 3) float32."""
    out = np.zeros((local_rows(opts), opts.width, 3), dtype=np.float32)
    st = RayStats()
    rc = lib().orc_render_frame(desc, C.byref(cam), C.byref(opts), out.ctypes.data_as(C.c_void_p), n_threads, C.byref(st))
    if rc != 0:
        raise RuntimeError("orc_render_frame status %d" % rc)
    if stats is not None:
        stats["primary"] = int(st.primary_rays)
        stats["shadow"] = int(st.shadow_rays)
    return out


def render_pixel(desc, cam, opts, x, y):
    r = TraceResult()
    rc = lib().orc_render_pixel(desc, C.byref(cam), C.byref(opts), x, y, C.byref(r))
    if rc != 0:
        raise RuntimeError("orc_render_pixel status %d" % rc)
    return r


def bmp_decode(data):
    """-> (float (H,W,3), raw uint32 (H,W)) as the reference's loadBmp!Color / loadBmp!uint."""
    buf = (C.c_uint8 * len(data)).from_buffer_copy(data)
    w, h = C.c_uint32(), C.c_uint32()
    raw = C.POINTER(C.c_uint32)()
    p = lib().orc_bmp_decode(buf, len(data), C.byref(w), C.byref(h), C.byref(raw))
    if not p:
        return None, None
    arr = np.ctypeslib.as_array(p, shape=(h.value, w.value, 3)).copy()
    rawa = np.ctypeslib.as_array(raw, shape=(h.value, w.value)).copy()
    lib().orc_free(p)
    lib().orc_free(raw)
    return arr, rawa
