#!/usr/bin/env python3
"""ms per frame at the reference's actual boundary (renderRT writes the caller's HOST Image!Color): c2rt_render_frame
into a pinned float frame and c2rt_render_frame_rgb32 into pinned display words, lecture5 4K x5, for the pipeline
parameters in the environment (C2RT_HOST_CHUNK_MB, C2RT_HOST_FIRST_FRAC, C2RT_HOST_COPY_STREAMS,
C2RT_HOST_DIRECT_STORE).  Prints one line."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import chess2rt_amd as c2
from bench import SCENES, WORKLOADS, boundary_timings

name = sys.argv[1] if len(sys.argv) > 1 else "lecture5_4k_aa5"
scene_file, w, h, taps, dof = WORKLOADS[name]
s = c2.parseSceneFromFile(os.path.join(SCENES, scene_file)); s.setFrameSize(w, h); s.setDof(dof)
cam = s.beginFrame(); ctx = c2.Context(0); ctx.uploadScene(s.desc)
b = boundary_timings(np, ctx, cam, s.renderOpts(taps=taps), n=20)
tag = " ".join("%s=%s" % (k[10:], v) for k, v in sorted(os.environ.items()) if k.startswith("C2RT_HOST_"))
print("%-60s float %.3f ms  rgb32 %.3f ms" % (tag or "(defaults)", b["host_float_pinned_ms"], b["host_rgb32_pinned_ms"]))
if os.environ.get("C2RT_HOST_MEASURE_COPY"):
    import torch
    d = torch.empty((h, w, 3), dtype=torch.float32, device="cuda")
    hp = torch.empty((h, w, 3), dtype=torch.float32, pin_memory=True)
    for _ in range(3):
        hp.copy_(d, non_blocking=True); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(20):
        hp.copy_(d, non_blocking=True); torch.cuda.synchronize()
    dt = (time.perf_counter() - t) / 20
    print("plain pinned D2H of the float frame (%.1f MB): %.3f ms = %.1f GB/s" % (d.numel() * 4 / 1e6, dt * 1e3, d.numel() * 4 / dt / 1e9))
