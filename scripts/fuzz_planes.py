"""Offline sweep of planes-only scenes (kernel instances with plane_points_away) against the oracle."""
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
from scene_fuzz import planes_scene_sdl
d = '/tmp/fzp'; os.makedirs(d, exist_ok=True); shutil.copy(ROOT + '/tests/golden/scenes/floor.bmp', d + '/floor.bmp')
ctx = BothContext(0); bad = 0; nne = 0
START = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
for seed in range(START, START + (int(sys.argv[1]) if len(sys.argv) > 1 else 1000)):
    open(d + "/f.sdl", "w").write(planes_scene_sdl(seed))
    s = c2.parseSceneFromFile(d + '/f.sdl'); s.setFrameSize(64, 48); cam = s.beginFrame(); opts = s.renderOpts(count_rays=1)
    ctx.uploadScene(s.desc); a = ctx.renderFrame(cam, opts); pr, sh = ctx.rayStats(); st = {}
    r = orc.render_frame(s.desc, cam, opts, 8, st)
    same = np.array_equal(np.isnan(a), np.isnan(r)) and np.array_equal(np.isinf(a), np.isinf(r))
    fin = np.isfinite(r); dd = np.abs(np.where(fin, a, 0).astype(np.float64) - np.where(fin, r, 0))
    nne += int((dd != 0).sum())
    if not same or dd.max() > 1e-4 or (pr, sh) != (st['primary'], st['shadow']):
        bad += 1; print('MISMATCH seed', seed, float(dd.max()), (pr, sh), st)
    if seed % 250 == 0: print('progress', seed, bad, nne, flush=True)
print('tiles redone through exact::', BothContext.redone); print('done: bad scenes', bad, 'differing floats', nne)
