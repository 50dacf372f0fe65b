"""Where a frame's wave-cycles go, by tile class (diagnostics build):
    make VARIANT=tilestats EXTRA_HIPFLAGS=-DC2RT_TILE_STATS=1
    C2RT_LIB_VARIANT=tilestats python scripts/tile_stats.py [workload]
Every wave stamps the shader clock at the start and end of its tile (s_memtime; with 4 waves per SIMD a
wave's lifetime includes the time it waits for its turn, so shares are shares of resident time)."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import chess2rt_amd as c2
from chess2rt_amd import _abi
from bench import SCENES, WORKLOADS

name = sys.argv[1] if len(sys.argv) > 1 else "lecture5_4k_aa5"
scene_file, w, h, taps, dof = WORKLOADS[name]
ctx = c2.Context(0)
s = c2.parseSceneFromFile(os.path.join(SCENES, scene_file)); s.setFrameSize(w, h); s.setDof(dof)
cam = s.beginFrame(); opts = s.renderOpts(taps=taps)
ctx.uploadScene(s.desc)
tx, ty = (w + 7) // 8, (h + 7) // 8
stats = torch.zeros((ty * tx + 4 + 3 * 9, 2), dtype=torch.int32, device="cuda:0")   # + four 64-bit lane counters + 9 regions x {ticks, lanes, slots} at the end
lib = _abi.load_library()
lib.c2rt_debug_set_tile_stats.argtypes = [C.c_void_p, C.c_void_p]; lib.c2rt_debug_set_tile_stats.restype = None
lib.c2rt_debug_set_tile_stats(ctx.handle, C.c_void_p(stats.data_ptr()))
out = torch.empty((h, w, 3), dtype=torch.float32, device="cuda:0")
for _ in range(3):
    ctx.renderFrameDevice(cam, opts, out.data_ptr(), torch.cuda.current_stream().cuda_stream)
torch.cuda.synchronize()
raw = stats.cpu().numpy()
lane = raw[ty * tx:].view(np.uint64).ravel()    # {active, slots} x {CSG evaluations entered, child stepping calls}, over the 3 frames
a = raw[:ty * tx].astype(np.int64) & 0xFFFFFFFF
cyc, cls = a[:, 0], a[:, 1]
if not cyc.any():
    raise SystemExit("no stamps: the loaded library was not built with -DC2RT_TILE_STATS=1")
nn = s.desc.contents.n_nodes
pm = (cls >> 8) & ((1 << min(nn, 24)) - 1)
kind = np.where(cls & 2, "ground only (primary + shadow)", np.where(cls & 1, "ground primary, objects may shadow", "objects in view"))
total = cyc.sum()
print("%s: %d tiles, mean %.0f cycles per tile-wave" % (name, len(cyc), cyc.mean()))
for k in ("ground only (primary + shadow)", "ground primary, objects may shadow", "objects in view"):
    m = kind == k
    if m.any():
        print("  %-36s %5.1f %% of tiles  %5.1f %% of wave-cycles  mean %7.0f cycles" % (k, 100 * m.mean(), 100 * cyc[m].sum() / total, cyc[m].mean()))
for n in range(min(nn, 24)):
    m = ((pm >> n) & 1) == 1
    print("  node %2d in the primary mask: %5.1f %% of tiles  %5.1f %% of wave-cycles  mean %7.0f cycles" % (n, 100 * m.mean(), 100 * cyc[m].sum() / total, cyc[m].mean() if m.any() else 0))
if lane[1]:
    print("  depth-1 CSG evaluations entered: %.1f of 64 lanes active on average (%.3f); child stepping calls: %.1f of 64 (%.3f)" % (
        64.0 * lane[0] / lane[1], lane[0] / lane[1], 64.0 * lane[2] / max(lane[3], 1), lane[2] / max(lane[3], 1)))
reg = lane[4:4 + 27].reshape(9, 3).astype(np.float64)
names = ["primary node loop (whole)", "  CsgOp evaluation (primary + shadow rays)", "  replay of the winning hit (inside it)", "closest hit's surface (hit_surface waterfall)",
         "shade (whole, incl. shadow ray)", "  shadow ray (test_visibility)", "  texture lookup", "  lit branch (cos terms, Phong pow)", "  (inside the surface waterfall) Sphere u,v: atan2, asin, x87 emulation"]
whole = float(cyc.sum()) * 3       # the per-tile stamps hold the last of the 3 frames; the region counters add up all 3
if reg[:, 2].any():
    print("  regions (share of all wave-ticks; lanes active at entry, of 64; idle lane share x time share = what perfect regrouping of that region could win):")
    for i, nm in enumerate(names):
        t, act, slots = reg[i]
        if slots:
            occ = act / slots
            print("    %-52s %5.1f %% of ticks  entered %9d times  %5.1f lanes (%.3f)  idle x share = %4.1f %%" % (nm, 100 * t / whole, slots / 64, 64 * occ, occ, 100 * t / whole * (1 - occ)))
