"""PCIe-inclusive rate: c2rt_render_frame with a HOST output buffer (kernel + D2H copy)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import chess2rt_amd as c2
ctx = c2.Context(0)
import numpy as np
s = c2.parseSceneFromFile(os.path.join(ROOT, "tests/golden/scenes/lecture5.sdl"))
for (w, h, taps) in [(3840, 2160, 5), (3840, 2160, 1), (1920, 1080, 1)]:
    s.setFrameSize(w, h)
    cam = s.beginFrame(); opts = s.renderOpts(taps=taps, count_rays=1)
    ctx.uploadScene(s.desc)
    ctx.renderFrame(cam, opts); pr, sh = ctx.rayStats()
    opts = s.renderOpts(taps=taps)
    t = time.perf_counter(); n = 10
    for _ in range(n):
        ctx.renderFrame(cam, opts)
    dt = (time.perf_counter() - t) / n
    print("lecture5 %dx%d %d tap(s): host-output %.3f ms/frame, %.0f Mray/s (kernel + D2H of %.1f MB into pageable memory)" % (w, h, taps, dt * 1e3, (pr + sh) / dt / 1e6, w * h * 12 / 1e6))
    out = np.empty((h, w, 3), np.float32)
    ctx.pinHostBuffer(out)
    ctx.renderFrameInto(cam, opts, out)
    t = time.perf_counter()
    for _ in range(n):
        ctx.renderFrameInto(cam, opts, out)
    dt = (time.perf_counter() - t) / n
    ctx.unpinHostBuffer(out)
    print("    same into a buffer pinned with c2rt_pin_host_buffer: %.3f ms/frame, %.0f Mray/s" % (dt * 1e3, (pr + sh) / dt / 1e6))
    out32 = np.empty((h, w), np.uint32)
    ctx.renderFrameRGB32Into(cam, opts, out32)
    t = time.perf_counter()
    for _ in range(n):
        ctx.renderFrameRGB32Into(cam, opts, out32)
    dt = (time.perf_counter() - t) / n
    print("    RGB32 (Color.toRGB32 on the GPU, %.1f MB over PCIe) into pageable memory: %.3f ms/frame" % (w * h * 4 / 1e6, dt * 1e3))
    ctx.pinHostBuffer(out32)
    ctx.renderFrameRGB32Into(cam, opts, out32)
    t = time.perf_counter()
    for _ in range(n):
        ctx.renderFrameRGB32Into(cam, opts, out32)
    dt = (time.perf_counter() - t) / n
    ctx.unpinHostBuffer(out32)
    print("    RGB32 into a pinned buffer (row chunks, copies overlapped): %.3f ms/frame, %.0f Mray/s" % (dt * 1e3, (pr + sh) / dt / 1e6))
