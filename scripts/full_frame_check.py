"""One-off: a full BASELINE-size frame on the GPU against the oracle, float for float.
usage: full_frame_check.py <scene.sdl> <W> <H> <taps> [dof]"""
import os, sys, time, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, ROOT + '/tests')
import chess2rt_amd as c2, oracle_lib as orc
scene, W, H, taps = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
s = c2.parseSceneFromFile(os.path.join(ROOT, 'tests/golden/scenes', scene)); s.setFrameSize(W, H); s.setDof(len(sys.argv) > 5)
cam = s.beginFrame(); o = s.renderOpts(taps=taps, count_rays=1)
ctx = c2.Context(0); ctx.uploadScene(s.desc)
a = ctx.renderFrame(cam, o); pr = ctx.rayStats()
t = time.time(); st = {}; r = orc.render_frame(s.desc, cam, o, 0, st); dt = time.time() - t
d = np.abs(a.astype(np.float64) - r.astype(np.float64))
print('%s %dx%d x%d%s: max|d| %.3g, differing floats %d of %d, rays gpu %s oracle %s (oracle %.1f s)' % (
    scene, W, H, taps, ' dof' if len(sys.argv) > 5 else '', float(d.max()), int((d != 0).sum()), a.size, pr, (st['primary'], st['shadow']), dt))
