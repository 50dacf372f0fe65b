"""Would a HIP graph of [tile-mask pre-pass, frame kernel] shorten a small frame's period?  Captures one
c2rt_render_frame_device call into a graph (static camera) and times N replays against N direct calls.
usage: graph_replay_probe.py [workload] [frames]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import chess2rt_amd as c2
from bench import SCENES, WORKLOADS
name = sys.argv[1] if len(sys.argv) > 1 else "lecture5_1080p"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 400
scene_file, w, h, taps, dof = WORKLOADS[name]
ctx = c2.Context(0)
s = c2.parseSceneFromFile(os.path.join(SCENES, scene_file)); s.setFrameSize(w, h); s.setDof(dof)
cam = s.beginFrame(); opts = s.renderOpts(taps=taps)
ctx.uploadScene(s.desc)
dev = torch.device("cuda", 0)
out = torch.empty((h, w, 3), dtype=torch.float32, device=dev)
st = torch.cuda.Stream(dev)
def direct(k):
    for _ in range(k):
        ctx.renderFrameDevice(cam, opts, out.data_ptr(), st.cuda_stream)
direct(20); torch.cuda.synchronize()
t0 = time.perf_counter(); direct(n); torch.cuda.synchronize(); t_direct = (time.perf_counter() - t0) / n * 1e6
ref = out.clone()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g, stream=st):
    ctx.renderFrameDevice(cam, opts, out.data_ptr(), torch.cuda.current_stream(dev).cuda_stream)
out.zero_()
with torch.cuda.stream(st):
    for _ in range(20): g.replay()
torch.cuda.synchronize()
assert torch.equal(out, ref), "graph replay frame differs"
t0 = time.perf_counter()
with torch.cuda.stream(st):
    for _ in range(n): g.replay()
torch.cuda.synchronize(); t_graph = (time.perf_counter() - t0) / n * 1e6
print("%s: direct %.2f us/frame, graph replay %.2f us/frame" % (name, t_direct, t_graph))
