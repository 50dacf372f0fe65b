"""ctypes mirror of include/c2rt.h and include/c2rt_host.h.

The product path lives in ``libc2rt.so`` (HIP kernels + C ABI + the C++ host
mirror).  There is no Python or CPU fallback: if the library is missing,
loading fails loudly.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# C2RT_LIB_VARIANT is a development knob (A/B builds from `make VARIANT=x`)
_VARIANT = os.environ.get("C2RT_LIB_VARIANT", "")
LIB_PATH = os.path.join(_HERE, "libc2rt%s.so" % ("_" + _VARIANT if _VARIANT else ""))

# ---- enums (include/c2rt.h) -------------------------------------------------
OK, ERR_INVALID_ARG, ERR_NO_DEVICE, ERR_HIP, ERR_UNSUPPORTED, ERR_LIMIT, ERR_NO_SCENE, ERR_CANCELLED, ERR_IO, ERR_PARSE = range(10)
GEOM_PLANE, GEOM_SPHERE, GEOM_CUBE, GEOM_CSG_UNION, GEOM_CSG_INTER, GEOM_CSG_DIFF = range(6)
SHADER_LAMBERT, SHADER_PHONG = 0, 1
TEX_CHECKER, TEX_PROCEDURE2, TEX_BITMAP = 0, 1, 2
LIGHT_POINT = 0
TAPS_1, TAPS_REF5, TAPS_4 = 1, 5, 4
ABI_VERSION = 1
MAX_CSG_DEPTH = 4
MAX_CSG_HITS = 8

_i32p = C.POINTER(C.c_int32)
_u32p = C.POINTER(C.c_uint32)
_u64p = C.POINTER(C.c_uint64)
_f32p = C.POINTER(C.c_float)
_f64p = C.POINTER(C.c_double)


class SceneDesc(C.Structure):
    _fields_ = [
        ("abi_version", C.c_uint32),
        ("n_geoms", C.c_uint32),
        ("geom_type", _i32p),
        ("geom_param", _f64p),
        ("geom_child", _i32p),
        ("n_textures", C.c_uint32),
        ("tex_type", _i32p),
        ("tex_color", _f32p),
        ("tex_param", _f64p),
        ("tex_scaling", _f32p),
        ("tex_width", _u32p),
        ("tex_height", _u32p),
        ("tex_offset", _u64p),
        ("n_texels", C.c_uint64),
        ("texels", _f32p),
        ("n_shaders", C.c_uint32),
        ("shader_type", _i32p),
        ("shader_color", _f32p),
        ("shader_texture", _i32p),
        ("shader_exponent", _f64p),
        ("shader_strength", _f32p),
        ("n_lights", C.c_uint32),
        ("light_type", _i32p),
        ("light_pos", _f64p),
        ("light_color", _f32p),
        ("light_power", _f32p),
        ("n_nodes", C.c_uint32),
        ("node_geom", _i32p),
        ("node_shader", _i32p),
        ("node_bump", _i32p),
        ("node_transform", _f64p),
        ("ambient", C.c_float * 3),
        ("max_trace_depth", C.c_uint32),
        ("gi_enabled", C.c_uint32),
    ]


class CameraFrame(C.Structure):
    _fields_ = [
        ("pos", C.c_double * 3),
        ("up_left", C.c_double * 3),
        ("up_right", C.c_double * 3),
        ("down_left", C.c_double * 3),
        ("right_dir", C.c_double * 3),
        ("up_dir", C.c_double * 3),
        ("front_dir", C.c_double * 3),
        ("frame_width", C.c_double),
        ("frame_height", C.c_double),
        ("dof", C.c_uint32),
        ("num_samples", C.c_uint32),
        ("focal_plane_dist", C.c_double),
        ("disc_multiplier", C.c_double),
        ("stereo_separation", C.c_double),
    ]


class RenderOpts(C.Structure):
    _fields_ = [
        ("width", C.c_uint32),
        ("height", C.c_uint32),
        ("taps", C.c_uint32),
        ("strip_height", C.c_uint32),
        ("strip_rank", C.c_uint32),
        ("strip_world", C.c_uint32),
        ("seed", C.c_uint64),
        ("count_rays", C.c_uint32),
        ("prepass_bucket", C.c_uint32),
    ]


class TraceResult(C.Structure):
    _fields_ = [
        ("color", C.c_float * 3),
        ("closest_node", C.c_int32),
        ("leaf_geom", C.c_int32),
        ("p", C.c_double * 3),
        ("normal", C.c_double * 3),
        ("dist", C.c_double),
        ("u", C.c_double),
        ("v", C.c_double),
        ("ray_orig", C.c_double * 3),
        ("ray_dir", C.c_double * 3),
    ]


class RayStats(C.Structure):
    _fields_ = [("primary_rays", C.c_uint64), ("shadow_rays", C.c_uint64)]


class HostSettings(C.Structure):
    _fields_ = [
        ("frame_width", C.c_uint32), ("frame_height", C.c_uint32),
        ("fullscreen", C.c_uint32), ("allow_resize", C.c_uint32),
        ("dynamic_aspect_ratio", C.c_uint32), ("interactive", C.c_uint32),
        ("bucket_size", C.c_uint32), ("thread_count", C.c_uint32),
        ("prepass_enabled", C.c_uint32), ("prepass_only", C.c_uint32),
        ("gi_enabled", C.c_uint32), ("aa_enabled", C.c_uint32),
        ("aa_threshold", C.c_double),
        ("paths_per_pixel", C.c_uint32), ("max_trace_depth", C.c_uint32),
        ("ambient", C.c_float * 3),
        ("debug_enabled", C.c_uint32),
    ]


class HostCamera(C.Structure):
    _fields_ = [
        ("frame_width", C.c_uint64), ("frame_height", C.c_uint64),
        ("aspect", C.c_double),
        ("pos", C.c_double * 3),
        ("yaw", C.c_double), ("pitch", C.c_double), ("roll", C.c_double), ("fov", C.c_double),
        ("focal_plane_dist", C.c_double), ("f_number", C.c_double), ("disc_multiplier", C.c_double),
        ("dof", C.c_uint32),
        ("num_samples", C.c_uint64),
        ("stereo_separation", C.c_double),
    ]


_SCENE_P = C.POINTER(SceneDesc)
_CAM_P = C.POINTER(CameraFrame)
_OPTS_P = C.POINTER(RenderOpts)
_VP = C.c_void_p

# every symbol include/c2rt.h declares: name -> (restype, argtypes)
C2RT_SYMBOLS = {
    "c2rt_init": (C.c_int, [C.c_int, C.POINTER(_VP)]),
    "c2rt_init_multi": (C.c_int, [C.c_int, C.POINTER(C.c_int), C.POINTER(_VP)]),
    "c2rt_device_count": (C.c_int, [_VP]),
    "c2rt_scene_generation": (C.c_uint64, [_VP]),
    "c2rt_get_csg_truncations": (C.c_int, [_VP, C.POINTER(C.c_uint64)]),
    "c2rt_get_exact_redos": (C.c_int, [_VP, C.POINTER(C.c_uint64)]),
    "c2rt_destroy": (None, [_VP]),
    "c2rt_last_error": (C.c_char_p, [_VP]),
    "c2rt_status_string": (C.c_char_p, [C.c_int]),
    "c2rt_abi_version": (C.c_uint32, []),
    "c2rt_upload_scene": (C.c_int, [_VP, _SCENE_P]),
    "c2rt_local_rows": (C.c_uint32, [_OPTS_P]),
    "c2rt_render_frame": (C.c_int, [_VP, _CAM_P, _OPTS_P, _VP, _VP]),
    "c2rt_render_frame_device": (C.c_int, [_VP, _CAM_P, _OPTS_P, _VP, _VP]),
    "c2rt_pin_host_buffer": (C.c_int, [_VP, _VP, C.c_size_t]),
    "c2rt_unpin_host_buffer": (C.c_int, [_VP, _VP]),
    "c2rt_get_ray_stats": (C.c_int, [_VP, C.POINTER(RayStats)]),
    "c2rt_render_pixel": (C.c_int, [_VP, _CAM_P, _OPTS_P, C.c_int, C.c_int, C.POINTER(TraceResult)]),
    "c2rt_deinterleave_strips": (C.c_int, [_VP, _VP, _VP, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, _VP]),
    "c2rt_encode_rgb32": (C.c_int, [_VP, _VP, _VP, C.c_uint64, _VP]),
    "c2rt_render_frame_rgb32": (C.c_int, [_VP, _CAM_P, _OPTS_P, _VP, _VP]),
    "c2rt_deinterleave_strips_rgb32": (C.c_int, [_VP, _VP, _VP, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, _VP]),
}

# every symbol include/c2rt_host.h declares
C2RT_HOST_SYMBOLS = {
    "c2rt_host_scene_load": (C.c_int, [C.c_char_p, C.POINTER(_VP), C.c_char_p, C.c_size_t]),
    "c2rt_host_scene_free": (None, [_VP]),
    "c2rt_host_scene_name": (C.c_char_p, [_VP]),
    "c2rt_host_scene_desc": (_SCENE_P, [_VP]),
    "c2rt_host_scene_get_settings": (None, [_VP, C.POINTER(HostSettings)]),
    "c2rt_host_scene_get_camera": (None, [_VP, C.POINTER(HostCamera)]),
    "c2rt_host_scene_set_camera": (None, [_VP, C.POINTER(HostCamera)]),
    "c2rt_host_scene_set_frame_size": (None, [_VP, C.c_uint32, C.c_uint32]),
    "c2rt_host_scene_set_aa": (None, [_VP, C.c_uint32]),
    "c2rt_host_scene_set_dof": (None, [_VP, C.c_uint32]),
    "c2rt_host_scene_begin_frame": (None, [_VP, _CAM_P]),
    "c2rt_host_camera_move": (None, [_VP, C.c_double, C.c_double, C.c_double]),
    "c2rt_host_camera_rotate": (None, [_VP, C.c_double, C.c_double, C.c_double]),
    "c2rt_host_render_rt": (C.c_int, [_VP, _VP, _VP, _VP]),
    "c2rt_host_render_scene_async": (C.c_int, [_VP, _VP, _VP, _VP, _VP]),
    "c2rt_host_render_wait": (C.c_int, [_VP]),
    "c2rt_host_render_pixel": (C.c_int, [_VP, _VP, C.c_int, C.c_int, C.POINTER(TraceResult)]),
    "c2rt_host_bmp_decode": (C.c_int, [_VP, C.c_size_t, _u32p, _u32p, C.POINTER(_f32p)]),
    "c2rt_host_texture_gamma": (None, [_VP, C.c_size_t, C.c_float]),
    "c2rt_host_bmp_encode": (C.c_int, [_VP, C.c_uint32, C.c_uint32, C.POINTER(C.POINTER(C.c_uint8)), C.POINTER(C.c_size_t)]),
    "c2rt_host_color_to_rgb32": (C.c_uint32, [_f32p]),
    "c2rt_host_free": (None, [_VP]),
    "c2rt_host_transform_reset": (None, [_f64p]),
    "c2rt_host_transform_scale": (None, [_f64p, C.c_double, C.c_double, C.c_double]),
    "c2rt_host_transform_rotate": (None, [_f64p, C.c_double, C.c_double, C.c_double]),
    "c2rt_host_transform_translate": (None, [_f64p, _f64p]),
    "c2rt_host_transform_point": (None, [_f64p, _f64p, _f64p]),
}

_lib = None


def load_library():
    """Loads libc2rt.so (built by ``make`` / ``__graft_entry__.build()``)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "chess2rt_amd: %s is missing. Build it with `make -j8` (or "
            "`python -c 'import __graft_entry__ as g; g.build()'`). There is no "
            "CPU fallback for the render path." % LIB_PATH)
    lib = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
    for table in (C2RT_SYMBOLS, C2RT_HOST_SYMBOLS):
        for name, (restype, argtypes) in table.items():
            fn = getattr(lib, name)  # AttributeError if the export is missing
            fn.restype = restype
            fn.argtypes = argtypes
    _lib = lib
    return lib
