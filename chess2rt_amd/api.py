"""Python face of the Chess2RT GPU render path.

Thin wrappers over the C ABI (include/c2rt.h) and the C++ host mirror
(include/c2rt_host.h) that keep the reference's names: ``parseSceneFromFile``
(rt/scene_loader.d:20-41), ``Scene.beginFrame`` (rt/scene.d:55-58),
``Renderer.renderRT`` / ``renderPixelNoAA`` (rt/renderer.d:83,223),
``renderPixel`` (rt/renderer.d:46-57).  Python holds no algorithm: all
rendering happens in libc2rt.so on the GPU.
"""
import ctypes as C

import numpy as np

from . import _abi
from ._abi import (CameraFrame, HostCamera, HostSettings, RayStats, RenderOpts, SceneDesc, TraceResult)


class C2rtError(RuntimeError):
    def __init__(self, status, message):
        super().__init__("c2rt status %d (%s): %s" % (status, _status_string(status), message))
        self.status = status


def _status_string(status):
    return _abi.load_library().c2rt_status_string(status).decode()


class Scene:
    """Host-side scene (mirror of rt/scene.d Scene), loaded from .sdl/.json."""

    def __init__(self, handle):
        self._lib = _abi.load_library()
        self._h = C.c_void_p(handle)

    def close(self):
        if self._h:
            self._lib.c2rt_host_scene_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def name(self):
        return self._lib.c2rt_host_scene_name(self._h).decode()

    @property
    def settings(self):
        s = HostSettings()
        self._lib.c2rt_host_scene_get_settings(self._h, C.byref(s))
        return s

    @property
    def camera(self):
        c = HostCamera()
        self._lib.c2rt_host_scene_get_camera(self._h, C.byref(c))
        return c

    @camera.setter
    def camera(self, cam):
        self._lib.c2rt_host_scene_set_camera(self._h, C.byref(cam))

    @property
    def desc(self):
        """POINTER(SceneDesc): the flat tables (owned by the scene)."""
        return self._lib.c2rt_host_scene_desc(self._h)

    def setFrameSize(self, width, height):
        """RTDemo.updateToWindowSize (gui/raytracer_demo.d:126-143)."""
        self._lib.c2rt_host_scene_set_frame_size(self._h, int(width), int(height))

    def setAA(self, enabled):
        self._lib.c2rt_host_scene_set_aa(self._h, 1 if enabled else 0)

    def setDof(self, enabled):
        self._lib.c2rt_host_scene_set_dof(self._h, 1 if enabled else 0)

    def beginFrame(self):
        cam = CameraFrame()
        self._lib.c2rt_host_scene_begin_frame(self._h, C.byref(cam))
        return cam

    def moveCamera(self, dx, dy, dz):
        self._lib.c2rt_host_camera_move(self._h, dx, dy, dz)

    def rotateCamera(self, dyaw, droll, dpitch):
        self._lib.c2rt_host_camera_rotate(self._h, dyaw, droll, dpitch)

    def renderOpts(self, **kw):
        s = self.settings
        o = RenderOpts()
        o.width, o.height = s.frame_width, s.frame_height
        o.taps = _abi.TAPS_REF5 if s.aa_enabled else _abi.TAPS_1
        for k, v in kw.items():
            setattr(o, k, v)
        return o


def parseSceneFromFile(path):
    lib = _abi.load_library()
    h = C.c_void_p()
    err = C.create_string_buffer(512)
    st = lib.c2rt_host_scene_load(str(path).encode(), C.byref(h), err, len(err))
    if st != _abi.OK:
        raise C2rtError(st, err.value.decode(errors="replace"))
    return Scene(h.value)


class Context:
    """One GPU context (c2rt_ctx).  Raises if no GPU is usable.

    ``Context(device)``: one device (c2rt_init).  ``Context(devices=[...])``: ONE context over several
    device slots of this process (c2rt_init_multi; ids may repeat, ``devices=0`` = every visible GPU):
    host-output frames are dealt to the slots in interleaved strips, device-output frames are stored by
    every slot straight into the lead device's frame."""

    def __init__(self, device=-1, devices=None):
        self._lib = _abi.load_library()
        h = C.c_void_p()
        if devices is None:
            st = self._lib.c2rt_init(int(device), C.byref(h))
        elif isinstance(devices, int):
            st = self._lib.c2rt_init_multi(int(devices), None, C.byref(h))
        else:
            ids = (C.c_int * len(devices))(*[int(d) for d in devices])
            st = self._lib.c2rt_init_multi(len(devices), ids, C.byref(h))
        self._h = h if h.value else None
        if st != _abi.OK:
            msg = self._lib.c2rt_last_error(self._h).decode() if self._h else ""
            if self._h:
                self._lib.c2rt_destroy(self._h)
                self._h = None
            raise C2rtError(st, msg)

    def close(self):
        if self._h:
            self._lib.c2rt_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def handle(self):
        return self._h

    def _check(self, st):
        if st != _abi.OK:
            raise C2rtError(st, self._lib.c2rt_last_error(self._h).decode(errors="replace"))

    @property
    def deviceCount(self):
        return int(self._lib.c2rt_device_count(self._h))

    @property
    def sceneGeneration(self):
        return int(self._lib.c2rt_scene_generation(self._h))

    def uploadScene(self, desc):
        """desc: POINTER(SceneDesc) or SceneDesc."""
        if isinstance(desc, SceneDesc):
            desc = C.pointer(desc)
        self._check(self._lib.c2rt_upload_scene(self._h, desc))

    def localRows(self, opts):
        return int(self._lib.c2rt_local_rows(C.byref(opts)))

    def renderFrame(self, cam, opts, stop_flag=None):
        """Blocking render into a new host array of shape (local_rows, W, 3)."""
        rows = self.localRows(opts)
        out = np.empty((rows, opts.width, 3), dtype=np.float32)
        stop = stop_flag.ctypes.data_as(C.c_void_p) if stop_flag is not None else None
        self._check(self._lib.c2rt_render_frame(self._h, C.byref(cam), C.byref(opts), out.ctypes.data_as(C.c_void_p), stop))
        return out

    def renderFrameInto(self, cam, opts, out, stop_flag=None):
        """Blocking render into a caller-owned float32 array (e.g. one pinned with pinHostBuffer)."""
        assert out.dtype == np.float32 and out.flags["C_CONTIGUOUS"] and out.size >= self.localRows(opts) * opts.width * 3
        stop = stop_flag.ctypes.data_as(C.c_void_p) if stop_flag is not None else None
        self._check(self._lib.c2rt_render_frame(self._h, C.byref(cam), C.byref(opts), out.ctypes.data_as(C.c_void_p), stop))
        return out

    def pinHostBuffer(self, arr):
        self._check(self._lib.c2rt_pin_host_buffer(self._h, arr.ctypes.data_as(C.c_void_p), arr.nbytes))

    def unpinHostBuffer(self, arr):
        self._check(self._lib.c2rt_unpin_host_buffer(self._h, arr.ctypes.data_as(C.c_void_p)))

    def renderFrameDevice(self, cam, opts, out_ptr, stream=0):
        """Enqueue a render into device memory (e.g. tensor.data_ptr())."""
        self._check(self._lib.c2rt_render_frame_device(self._h, C.byref(cam), C.byref(opts), C.c_void_p(out_ptr), C.c_void_p(stream)))

    def rayStats(self):
        s = RayStats()
        self._check(self._lib.c2rt_get_ray_stats(self._h, C.byref(s)))
        return int(s.primary_rays), int(s.shadow_rays)

    def csgTruncations(self):
        """CsgOp child hit lists that reached C2RT_MAX_CSG_HITS during the last counted frame."""
        n = C.c_uint64()
        self._check(self._lib.c2rt_get_csg_truncations(self._h, C.byref(n)))
        return int(n.value)

    def exactRedos(self):
        """Tiles this context has rendered a second time through the compiler's divide / sqrt (cumulative)."""
        n = C.c_uint64()
        self._check(self._lib.c2rt_get_exact_redos(self._h, C.byref(n)))
        return int(n.value)

    def renderPixel(self, cam, opts, x, y):
        r = TraceResult()
        self._check(self._lib.c2rt_render_pixel(self._h, C.byref(cam), C.byref(opts), int(x), int(y), C.byref(r)))
        return r

    def deinterleaveStrips(self, gathered_ptr, frame_ptr, width, height, strip_height, world, stream=0):
        self._check(self._lib.c2rt_deinterleave_strips(self._h, C.c_void_p(gathered_ptr), C.c_void_p(frame_ptr), width, height,
                                                       strip_height, world, C.c_void_p(stream)))

    def renderFrameRGB32(self, cam, opts, stop_flag=None):
        """Blocking render in display format: (local_rows, W) uint32 0x00RRGGBB (Color.toRGB32)."""
        out = np.empty((self.localRows(opts), opts.width), dtype=np.uint32)
        stop = stop_flag.ctypes.data_as(C.c_void_p) if stop_flag is not None else None
        self._check(self._lib.c2rt_render_frame_rgb32(self._h, C.byref(cam), C.byref(opts), out.ctypes.data_as(C.c_void_p), stop))
        return out

    def renderFrameRGB32Into(self, cam, opts, out, stop_flag=None):
        """Same into a caller-owned (local_rows, W) uint32 array (pin it with pinHostBuffer for overlapped copies)."""
        assert out.dtype == np.uint32 and out.flags["C_CONTIGUOUS"] and out.size == self.localRows(opts) * opts.width
        stop = stop_flag.ctypes.data_as(C.c_void_p) if stop_flag is not None else None
        self._check(self._lib.c2rt_render_frame_rgb32(self._h, C.byref(cam), C.byref(opts), out.ctypes.data_as(C.c_void_p), stop))
        return out

    def deinterleaveStripsRGB32(self, gathered_ptr, frame_ptr, width, height, strip_height, world, stream=0):
        self._check(self._lib.c2rt_deinterleave_strips_rgb32(self._h, C.c_void_p(gathered_ptr), C.c_void_p(frame_ptr), width, height,
                                                             strip_height, world, C.c_void_p(stream)))

    def encodeRGB32(self, frame_ptr, out_ptr, n_pixels, stream=0):
        self._check(self._lib.c2rt_encode_rgb32(self._h, C.c_void_p(frame_ptr), C.c_void_p(out_ptr), int(n_pixels), C.c_void_p(stream)))


class Renderer:
    """struct Renderer (rt/renderer.d:59-81) with the GPU behind renderRT."""

    def __init__(self, scene, ctx=None):
        self.scene = scene
        self.ctx = ctx if ctx is not None else Context()
        self._lib = _abi.load_library()

    def renderRT(self, stop_flag=None):
        s = self.scene.settings
        out = np.empty((s.frame_height, s.frame_width, 3), dtype=np.float32)
        stop = stop_flag.ctypes.data_as(C.c_void_p) if stop_flag is not None else None
        self.ctx._check(self._lib.c2rt_host_render_rt(self.ctx.handle, self.scene._h, out.ctypes.data_as(C.c_void_p), stop))
        return out

    def renderSceneAsync(self, out, is_rendering, needs_rendering=None):
        """renderSceneAsync (rt/renderer.d:23-44); `out`, flags: numpy arrays kept alive by the caller."""
        stop = needs_rendering.ctypes.data_as(C.c_void_p) if needs_rendering is not None else None
        self.ctx._check(self._lib.c2rt_host_render_scene_async(self.ctx.handle, self.scene._h, out.ctypes.data_as(C.c_void_p),
                                                               is_rendering.ctypes.data_as(C.c_void_p), stop))

    def wait(self):
        self.ctx._check(self._lib.c2rt_host_render_wait(self.scene._h))

    def renderPixelNoAA(self, x, y):
        r = TraceResult()
        self.ctx._check(self._lib.c2rt_host_render_pixel(self.ctx.handle, self.scene._h, int(x), int(y), C.byref(r)))
        return r


def renderPixel(scene, x, y, ctx=None):
    """renderPixel (rt/renderer.d:46-57): returns the TraceResult incl. colour."""
    return Renderer(scene, ctx).renderPixelNoAA(x, y)


def loadBmpImage(data):
    """loadBmpImage!Color (imageio/bmp.d:31-34): bytes -> (H, W, 3) float32."""
    lib = _abi.load_library()
    buf = (C.c_uint8 * len(data)).from_buffer_copy(data)
    w, h = C.c_uint32(), C.c_uint32()
    p = C.POINTER(C.c_float)()
    st = lib.c2rt_host_bmp_decode(buf, len(data), C.byref(w), C.byref(h), C.byref(p))
    if st != _abi.OK:
        raise C2rtError(st, "BMP decode failed")
    arr = np.ctypeslib.as_array(p, shape=(h.value, w.value, 3)).copy()
    lib.c2rt_host_free(p)
    return arr


def saveBmp(rgb):
    """Bitmap.saveImage -> saveBmp (imageio/bmp.d:195-237): (H, W, 3) float32 -> bytes."""
    lib = _abi.load_library()
    rgb = np.ascontiguousarray(rgb, dtype=np.float32)
    out = C.POINTER(C.c_uint8)()
    n = C.c_size_t()
    st = lib.c2rt_host_bmp_encode(rgb.ctypes.data_as(C.c_void_p), rgb.shape[1], rgb.shape[0], C.byref(out), C.byref(n))
    if st != _abi.OK:
        raise C2rtError(st, "BMP encode failed")
    data = bytes(bytearray(out[: n.value]))
    lib.c2rt_host_free(out)
    return data
