"""Offline sweep over render modes: odd frame sizes, 1/4/5 taps, strip sharding, on random scenes of all three generators."""
import os, sys, shutil, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, ROOT + '/tests')
import chess2rt_amd as c2, oracle_lib as orc

class BothContext(c2.Context):
    """count_rays = 1 frames come from the counting (exact::) instances; render the same frame with the
    production (lean:: + redo) instance as well and insist on the same bits — round 3: the two differ in code."""
    redone = 0
    def renderFrame(self, cam, opts, stop_flag=None):
        if not opts.count_rays:
            return super().renderFrame(cam, opts, stop_flag)
        plain = type(opts).from_buffer_copy(opts); plain.count_rays = 0
        before = self.exactRedos()
        a = super().renderFrame(cam, plain, stop_flag)
        BothContext.redone += self.exactRedos() - before
        b = super().renderFrame(cam, opts, stop_flag)
        if not np.array_equal(a.view(np.uint32), b.view(np.uint32)):
            print('LEAN != EXACT', int((a.view(np.uint32) != b.view(np.uint32)).sum()), 'words', flush=True)
            raise SystemExit(3)
        return b
from scene_fuzz import random_scene_sdl, ground_scene_sdl, planes_scene_sdl
d = '/tmp/fzm'; os.makedirs(d, exist_ok=True); shutil.copy(ROOT + '/tests/golden/scenes/floor.bmp', d + '/floor.bmp')
ctx = BothContext(0); bad = 0; nne = 0
START = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
N = int(sys.argv[2]) if len(sys.argv) > 2 else 600
sizes = [(97, 61), (64, 48), (131, 33), (8, 8), (200, 9), (57, 120)]
for seed in range(START, START + N):
    gen = (random_scene_sdl, ground_scene_sdl, planes_scene_sdl)[seed % 3]
    open(d + "/f.sdl", "w").write(gen(seed))
    s = c2.parseSceneFromFile(d + '/f.sdl'); s.setFrameSize(*sizes[seed % len(sizes)]); cam = s.beginFrame()
    kw = dict(count_rays=1, taps=(1, 4, 5)[(seed // 3) % 3])
    if seed % 4 == 0:
        kw.update(strip_height=(8, 16, 5)[(seed // 4) % 3], strip_rank=seed % 3, strip_world=3)
    opts = s.renderOpts(**kw)
    ctx.uploadScene(s.desc); a = ctx.renderFrame(cam, opts); pr, sh = ctx.rayStats(); st = {}
    r = orc.render_frame(s.desc, cam, opts, 8, st)
    same = a.shape == r.shape and np.array_equal(np.isnan(a), np.isnan(r)) and np.array_equal(np.isinf(a), np.isinf(r))
    fin = np.isfinite(r); dd = np.abs(np.where(fin, a, 0).astype(np.float64) - np.where(fin, r, 0)) if same else np.array([1.0])
    nne += int((dd != 0).sum())
    if not same or (dd.size and dd.max() > 1e-4) or (pr, sh) != (st['primary'], st['shadow']):
        bad += 1; print('MISMATCH seed', seed, kw, float(dd.max()) if dd.size else None, (pr, sh), st)
    if seed % 200 == 0: print('progress', seed, bad, nne, flush=True)
print('tiles redone through exact::', BothContext.redone); print('done: bad scenes', bad, 'differing floats', nne)
