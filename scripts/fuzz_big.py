import os, sys, shutil, numpy as np
ROOT='/root/repo'
sys.path.insert(0, ROOT); sys.path.insert(0, ROOT+'/tests')
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
from scene_fuzz import random_scene_sdl
d='/tmp/fz'; os.makedirs(d, exist_ok=True); shutil.copy(ROOT+'/tests/golden/scenes/floor.bmp', d+'/floor.bmp')
ctx = BothContext(0); worst=0; bad=0; nne=0
from scene_fuzz import many_lights_scene_sdl
START = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
COUNT = int(sys.argv[2]) if len(sys.argv) > 2 else 1500
W, H = (int(sys.argv[3]), int(sys.argv[4])) if len(sys.argv) > 4 else (64, 48)   # frame size: more tiles, other culling rectangles
for seed in range(START, START + COUNT):
    open(d+"/f.sdl","w").write(many_lights_scene_sdl(seed) if seed % 5 == 0 else random_scene_sdl(seed, max_depth=4 if seed%2 else 3))
    s=c2.parseSceneFromFile(d+'/f.sdl'); s.setFrameSize(W,H); cam=s.beginFrame(); opts=s.renderOpts(count_rays=1)
    ctx.uploadScene(s.desc); a=ctx.renderFrame(cam,opts); pr,sh=ctx.rayStats(); st={}
    r=orc.render_frame(s.desc,cam,opts,8,st)
    dd=np.abs(a.astype(np.float64)-r.astype(np.float64)); dd=np.where(np.isnan(a)&np.isnan(r),0,dd)
    m=float(np.nanmax(dd)); worst=max(worst,m); nne+=int((dd!=0).sum())
    if 0 < m <= 1e-4: print('WITHIN-TOLERANCE seed',seed,'max',m,'floats',int((dd!=0).sum()),'at',[tuple(int(v) for v in ix) for ix in np.argwhere(dd!=0)[:4]],flush=True)
    if m>1e-4 or not np.array_equal(np.isnan(a),np.isnan(r)) or (pr,sh)!=(st['primary'],st['shadow']):
        bad+=1; print('MISMATCH seed',seed,m,(pr,sh),(st['primary'],st['shadow']))
    if seed%250==0: print('progress',seed,worst,nne,flush=True)
print('tiles redone through exact::', BothContext.redone); print('done: worst',worst,'bad scenes',bad,'differing floats',nne)
