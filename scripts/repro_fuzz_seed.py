"""One seed of scripts/fuzz_big.py again, verbosely: differing pixels of the production and the counting instance against the
oracle, then the probe (c2rt_render_pixel) beside the oracle's for the first of them.  usage: repro_fuzz_seed.py <seed> [print-scene]
(C2RT_DEBUG_CULL=1|2|4 switches parts of the culling off: c2rt_api.cpp fill_params.)"""
import os, sys, shutil, numpy as np
ROOT='/root/repo'
sys.path.insert(0, ROOT); sys.path.insert(0, ROOT+'/tests')
import chess2rt_amd as c2
import oracle_lib as orc
from scene_fuzz import random_scene_sdl, many_lights_scene_sdl
seed=int(sys.argv[1])
d='/tmp/fzr'; os.makedirs(d, exist_ok=True); shutil.copy(ROOT+'/tests/golden/scenes/floor.bmp', d+'/floor.bmp')
txt = many_lights_scene_sdl(seed) if seed % 5 == 0 else random_scene_sdl(seed, max_depth=4 if seed%2 else 3)
open(d+"/f.sdl","w").write(txt)
if len(sys.argv)>2: print(txt)
s=c2.parseSceneFromFile(d+'/f.sdl'); W,H=(int(os.environ.get("FUZZ_W","64")),int(os.environ.get("FUZZ_H","48"))); s.setFrameSize(W,H); cam=s.beginFrame()
ctx=c2.Context(0); ctx.uploadScene(s.desc)
r=orc.render_frame(s.desc,cam,s.renderOpts(),8,{})
for cr in (0,1):
    opts=s.renderOpts(count_rays=cr)
    a=ctx.renderFrame(cam,opts)
    bad=np.argwhere((a.view(np.uint32)!=r.view(np.uint32)).any(axis=2) & ~(np.isnan(a)&np.isnan(r)).all(axis=2))
    print('variant',os.environ.get('C2RT_LIB_VARIANT') or 'base','count_rays',cr,'differing pixels',len(bad),[ (int(y),int(x)) for y,x in bad[:10]])
    for y,x in bad[:3]: print('   gpu',a[y,x],'oracle',r[y,x])
print('levels', s.desc.contents.n_geoms, 'nodes', s.desc.contents.n_nodes, 'lights', s.desc.contents.n_lights, 'exact redos', ctx.exactRedos())
opts=s.renderOpts()
for (y,x) in [(int(y),int(x)) for y,x in bad[:4]]:
    g=ctx.renderPixel(cam,opts,x,y); o=orc.render_pixel(s.desc,cam,opts,x,y)
    f=lambda r:(list(r.color), r.closest_node, r.leaf_geom, r.dist, list(r.p), list(r.normal), r.u, r.v)
    print((y,x)); print('  gpu   ',f(g)); print('  oracle',f(o))
