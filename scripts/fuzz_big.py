import os, sys, shutil, numpy as np
ROOT='/root/repo'
sys.path.insert(0, ROOT); sys.path.insert(0, ROOT+'/tests')
import chess2rt_amd as c2, oracle_lib as orc
from scene_fuzz import random_scene_sdl
d='/tmp/fz'; os.makedirs(d, exist_ok=True); shutil.copy(ROOT+'/tests/golden/scenes/floor.bmp', d+'/floor.bmp')
ctx=c2.Context(0); worst=0; bad=0; nne=0
from scene_fuzz import many_lights_scene_sdl
START = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
COUNT = int(sys.argv[2]) if len(sys.argv) > 2 else 1500
for seed in range(START, START + COUNT):
    open(d+"/f.sdl","w").write(many_lights_scene_sdl(seed) if seed % 5 == 0 else random_scene_sdl(seed, max_depth=4 if seed%2 else 3))
    s=c2.parseSceneFromFile(d+'/f.sdl'); s.setFrameSize(64,48); cam=s.beginFrame(); opts=s.renderOpts(count_rays=1)
    ctx.uploadScene(s.desc); a=ctx.renderFrame(cam,opts); pr,sh=ctx.rayStats(); st={}
    r=orc.render_frame(s.desc,cam,opts,8,st)
    dd=np.abs(a.astype(np.float64)-r.astype(np.float64)); dd=np.where(np.isnan(a)&np.isnan(r),0,dd)
    m=float(np.nanmax(dd)); worst=max(worst,m); nne+=int((dd!=0).sum())
    if m>1e-4 or not np.array_equal(np.isnan(a),np.isnan(r)) or (pr,sh)!=(st['primary'],st['shadow']):
        bad+=1; print('MISMATCH seed',seed,m,(pr,sh),(st['primary'],st['shadow']))
    if seed%250==0: print('progress',seed,worst,nne,flush=True)
print('done: worst',worst,'bad scenes',bad,'differing floats',nne)
