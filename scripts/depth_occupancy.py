"""Frame time of csg_stress.sdl cut down to CSG depth 2 / 3 / 4 (4K, 4 taps), for one library variant and one
first-pass hit-stack capacity (environment: C2RT_LIB_VARIANT, C2RT_CSG_FIRST_CAP): the occupancy / LDS trade of the
nested-CSG instances.  usage: depth_occupancy.py <depth> [frames]"""
import os, re, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import chess2rt_amd as c2
depth = int(sys.argv[1]); frames = int(sys.argv[2]) if len(sys.argv) > 2 else 10
src = open(os.path.join(ROOT, "tests/golden/scenes/csg_stress.sdl")).read()
drop = {4: [], 3: ["n_i4"], 2: ["n_i4", "n_i3"]}[depth]
for n in drop:
    src = re.sub(r'^\s*Node "%s".*\n' % n, "", src, flags=re.M)
d = tempfile.mkdtemp()
path = os.path.join(d, "csg_d%d.sdl" % depth)
open(path, "w").write(src)
s = c2.parseSceneFromFile(path); s.setFrameSize(3840, 2160)
cam = s.beginFrame(); opts = s.renderOpts(taps=4)
ctx = c2.Context(0); ctx.uploadScene(s.desc)
out = torch.empty((2160, 3840, 3), dtype=torch.float32, device="cuda:0")
st = torch.cuda.current_stream().cuda_stream
for _ in range(3): ctx.renderFrameDevice(cam, opts, out.data_ptr(), st)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(frames): ctx.renderFrameDevice(cam, opts, out.data_ptr(), st)
e1.record(); torch.cuda.synchronize()
print("depth %d  variant %-6s first cap %-3s  %.3f ms/frame  checksum %.6f" % (depth, os.environ.get("C2RT_LIB_VARIANT") or "base", os.environ.get("C2RT_CSG_FIRST_CAP") or "-", e0.elapsed_time(e1) / frames, float(out.double().mean())))
