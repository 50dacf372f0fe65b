"""Renders lecture5 at 4K with subsets of its nodes (one dispatch each, in a
fixed order) so that a PMC run attributes VALU instructions per node type."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import chess2rt_amd as c2
SUBSETS = [("floor", [0]), ("floor+globe", [0, 1]), ("floor+csg", [0, 2]), ("floor+3balls", [0, 3, 4, 5]), ("floor+1ball", [0, 3]), ("all", [0, 1, 2, 3, 4, 5]), ("csg only", [2]), ("globe only", [1])]
if __name__ == "__main__":
    ctx = c2.Context(0)
    s = c2.parseSceneFromFile(os.path.join(ROOT, "tests/golden/scenes/lecture5.sdl"))
    s.setFrameSize(3840, 2160); s.setAA(False)
    cam = s.beginFrame(); opts = s.renderOpts()
    d = s.desc.contents
    geom = [d.node_geom[i] for i in range(6)]; shader = [d.node_shader[i] for i in range(6)]
    tr = [[d.node_transform[30 * i + k] for k in range(30)] for i in range(6)]
    for name, idx in SUBSETS:
        n = len(idx)
        g = (C.c_int32 * n)(*[geom[i] for i in idx]); sh = (C.c_int32 * n)(*[shader[i] for i in idx]); b = (C.c_int32 * n)(*([-1] * n))
        t = (C.c_double * (30 * n))(*[x for i in idx for x in tr[i]])
        d.n_nodes, d.node_geom, d.node_shader, d.node_bump, d.node_transform = n, g, sh, b, t
        ctx.uploadScene(s.desc)
        img = ctx.renderFrame(cam, opts)
        print(name, float(img.mean()))
