"""Quick GPU-vs-oracle parity + timing probe (development helper)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import chess2rt_amd as c
import oracle_lib as o

ctx = c.Context()
scenes = sys.argv[1:] or ["lecture4.sdl", "lecture4-proc-texture.sdl", "lecture5.sdl", "zaphod.sdl"]
for name in scenes:
    s = c.parseSceneFromFile(os.path.join(ROOT, "tests/golden/scenes", name))
    s.setDof(False)
    for (w, h, aa) in [(640, 480, False), (640, 480, True), (1920, 1080, False)]:
        s.setFrameSize(w, h); s.setAA(aa)
        cam = s.beginFrame(); opts = s.renderOpts(count_rays=1)
        ctx.uploadScene(s.desc)
        t = time.time(); g = ctx.renderFrame(cam, opts); tg = time.time() - t
        pr, sh = ctx.rayStats()
        t = time.time(); g = ctx.renderFrame(cam, opts); tg2 = time.time() - t
        st = {}
        t = time.time(); r = o.render_frame(s.desc, cam, opts, 0, st); tc = time.time() - t
        d = np.abs(g.astype(np.float64) - r.astype(np.float64))
        d = np.where(np.isnan(g) & np.isnan(r), 0, d)
        print("%-26s %4dx%-4d aa=%d gpu %.1f/%.1f ms cpu %.1f ms  maxdiff %.3g  n>1e-4 %d  n!=0 %d  rays gpu %d/%d cpu %d/%d" % (
            name, w, h, aa, tg * 1e3, tg2 * 1e3, tc * 1e3, np.nanmax(d), int((d > 1e-4).sum()), int((d != 0).sum()),
            pr, sh, st["primary"], st["shadow"]), flush=True)
        if (d > 1e-4).any():
            ys, xs, cs = np.nonzero(d > 1e-4)
            for k in range(min(5, len(ys))):
                print("   bad px", xs[k], ys[k], g[ys[k], xs[k]], r[ys[k], xs[k]])
        np.save(os.path.join(ROOT, "gpurun_out", "gpu_%s_%dx%d_%d.npy" % (name, w, h, aa)), g) if w == 640 else None
