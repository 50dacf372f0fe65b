import os, sys, time
sys.path.insert(0, '/root/repo')
import torch, chess2rt_amd as c2
from bench import SCENES, WORKLOADS
name = sys.argv[1]; nf = int(sys.argv[2])
scene_file, w, h, taps, dof = WORKLOADS[name]
s = c2.parseSceneFromFile(os.path.join(SCENES, scene_file)); s.setFrameSize(w, h); s.setDof(dof)
cam = s.beginFrame(); opts = s.renderOpts(taps=taps)
dev = torch.device('cuda', 0)
ctxs = [c2.Context(0) for _ in range(nf)]
for c in ctxs: c.uploadScene(s.desc)
streams = [torch.cuda.Stream(dev) for _ in range(nf)]
outs = [torch.empty((h, w, 3), dtype=torch.float32, device=dev) for _ in range(nf)]
def frame(i): ctxs[i % nf].renderFrameDevice(cam, opts, outs[i % nf].data_ptr(), streams[i % nf].cuda_stream)
for i in range(4 * nf): frame(i)
torch.cuda.synchronize()
steps = 200
t = time.perf_counter()
for i in range(steps): frame(i)
torch.cuda.synchronize()
dt = (time.perf_counter() - t) / steps
ctxs[0].renderFrameDevice(cam, s.renderOpts(taps=taps, count_rays=1), outs[0].data_ptr(), streams[0].cuda_stream)
p, sh = ctxs[0].rayStats()
print("%s, %d in flight: %.4f ms/frame, %.0f Mray/s" % (name, nf, dt * 1e3, (p + sh) / dt / 1e6))
