"""chess2rt_amd — MI355X-native render hot path for Chess2RT.

Hand-written HIP kernels (gfx950) behind a plain-C ABI (include/c2rt.h), with
the reference's Scene / Camera / Renderer API mirrored in C++
(include/c2rt_host.h) and exposed here through ctypes.  No CPU fallback.
"""
from . import _abi
from ._abi import (CameraFrame, HostCamera, HostSettings, RayStats, RenderOpts, SceneDesc, TraceResult,
                   TAPS_1, TAPS_4, TAPS_REF5)
from .api import (C2rtError, Context, Renderer, Scene, loadBmpImage, parseSceneFromFile, renderPixel, saveBmp)
from .sharding import (StripPlan, deinterleave_strips_torch, exchange_strips_p2p, local_rows, plan_strips, render_frame_sharded)

__all__ = [
    "C2rtError", "Context", "Renderer", "Scene", "parseSceneFromFile", "renderPixel", "loadBmpImage", "saveBmp",
    "CameraFrame", "HostCamera", "HostSettings", "RayStats", "RenderOpts", "SceneDesc", "TraceResult",
    "TAPS_1", "TAPS_4", "TAPS_REF5",
    "StripPlan", "plan_strips", "local_rows", "render_frame_sharded", "deinterleave_strips_torch", "exchange_strips_p2p",
]
