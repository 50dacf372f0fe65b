"""Kernel time of arbitrary scenes: scripts/scene_time.py scene.sdl W H taps [scene2 ...]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import chess2rt_amd as c2
ctx = c2.Context(0)
args = sys.argv[1:]
for i in range(0, len(args), 4):
    name, w, h, taps = args[i], int(args[i + 1]), int(args[i + 2]), int(args[i + 3])
    s = c2.parseSceneFromFile(os.path.join(ROOT, "tests/golden/scenes", name))
    s.setFrameSize(w, h); s.setDof(False)
    cam = s.beginFrame(); ctx.uploadScene(s.desc)
    out = torch.empty((h, w, 3), dtype=torch.float32, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    o = s.renderOpts(taps=taps, count_rays=1)
    ctx.renderFrameDevice(cam, o, out.data_ptr(), st); pr, sh = ctx.rayStats()
    o = s.renderOpts(taps=taps)
    for _ in range(3): ctx.renderFrameDevice(cam, o, out.data_ptr(), st)
    torch.cuda.synchronize(); t = time.perf_counter(); n = 20
    for _ in range(n): ctx.renderFrameDevice(cam, o, out.data_ptr(), st)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t) / n
    print("%-24s %dx%d %d tap(s): %.3f ms/frame, %.0f Mray/s (%d primary + %d shadow)" % (name, w, h, taps, dt * 1e3, (pr + sh) / dt / 1e6, pr, sh))
