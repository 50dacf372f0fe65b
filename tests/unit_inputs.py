"""Fixed inputs for per-function vectors (SURVEY.md section 8(c) 'unit vectors
for each a10-a23 function').  unit_cases() evaluates them with the oracle."""
import ctypes as C
import math
import os

import numpy as np

from golden_configs import SCENES


def _hit(orc, dist):
    h = orc.OrcHit()
    h.dist = dist
    h.g = -1
    for f in ("p", "normal", "dNdx", "dNdy"):
        for i in range(3):
            getattr(h, f)[i] = float("nan")
    h.u = h.v = float("nan")
    return h


def _hit_dict(ok, h):
    def clean(v):
        return None if v != v else v
    return {"hit": int(ok), "dist": clean(h.dist), "g": h.g if ok else -1,
            "p": [clean(x) for x in h.p] if ok else None, "normal": [clean(x) for x in h.normal] if ok else None,
            "u": clean(h.u) if ok else None, "v": clean(h.v) if ok else None,
            "dNdx": [clean(x) for x in h.dNdx] if ok else None, "dNdy": [clean(x) for x in h.dNdy] if ok else None}


def rays_towards(origin, targets):
    out = []
    for t in targets:
        d = np.array(t, dtype=np.float64) - np.array(origin, dtype=np.float64)
        d = d * (1.0 / math.sqrt(float((d * d).sum())))
        out.append((list(map(float, origin)), [float(x) for x in d]))
    return out


def unit_cases(orc, c2):
    L = orc.lib()
    out = {}
    l4 = c2.parseSceneFromFile(os.path.join(SCENES, "lecture4.sdl"))
    l4p = c2.parseSceneFromFile(os.path.join(SCENES, "lecture4-proc-texture.sdl"))
    l5 = c2.parseSceneFromFile(os.path.join(SCENES, "lecture5.sdl"))
    cs = c2.parseSceneFromFile(os.path.join(SCENES, "csg_stress.sdl"))

    # a6 Camera.getScreenRay
    cam = l4.beginFrame()
    cases = []
    for (x, y) in [(0, 0), (320, 240), (100, 400), (639, 479), (320.3, 240.3), (10.6, 7.0), (0.0, 479.6)]:
        o, d = orc.vec3(0, 0, 0), orc.vec3(0, 0, 0)
        L.orc_screen_ray(C.byref(cam), x, y, o, d)
        cases.append({"x": x, "y": y, "orig": list(o), "dir": list(d)})
    out["screen_ray_lecture4_640x480"] = cases

    # a10-a16 Geometry.intersect on lecture5's geometries (file order) from the camera
    origin = (0.0, 165.0, 0.0)
    targets = [(100, 50, 320), (100, 95, 320), (149.9, 50, 320), (-100, 60, 200), (-60, 100, 160), (-100, 110, 200),
               (-140, 20, 240), (100, 15, 256), (0, -0.01, 300), (500, -0.01, 900), (0, 400, 100), (-100, 60, 150),
               (-52, 60, 152), (-148, 108, 152)]
    rays = rays_towards(origin, targets) + rays_towards((-100, 60, 200), [(0, 165, 0), (-100, 300, 200), (50, 60, 200)])
    cases = []
    for g in range(l5.desc.contents.n_geoms):
        for (o, d) in rays:
            for dist in (1e99, 150.0):
                h = _hit(orc, dist)
                ok = L.orc_geom_intersect(l5.desc, g, orc.vec3(*o), orc.vec3(*d), C.byref(h))
                cases.append({"geom": g, "orig": o, "dir": d, "dist_in": dist, "out": _hit_dict(ok, h)})
    out["geom_intersect_lecture5"] = cases

    # nested CSG + transformed nodes: Node.intersect on csg_stress
    cam = cs.beginFrame()
    cases = []
    origin = (20.0, 140.0, -60.0)
    targets = [(-90, 40, 180), (-90, 80, 140), (-60, 70, 150), (90, 50, 220), (120, 80, 190), (60, 90, 180), (0, 25, 90),
               (20, 40, 80), (10, 50, 70), (-20, 30, 30), (-20, 55, 30), (110, 12, 60), (130, 15, 70), (40, -2, 20), (45, 8, 25)]
    for n in range(cs.desc.contents.n_nodes):
        for (o, d) in rays_towards(origin, targets):
            h = _hit(orc, 1e99)
            ok = L.orc_node_intersect(cs.desc, n, orc.vec3(*o), orc.vec3(*d), C.byref(h))
            cases.append({"node": n, "orig": o, "dir": d, "out": _hit_dict(ok, h)})
    out["node_intersect_csg_stress"] = cases

    # isInside
    cases = []
    pts = [(-90, 40, 180), (-90, 79, 180), (-50, 40, 180), (90, 50, 220), (120, 80, 190), (0, 25, 90), (25, 45, 75), (0, 0, 0), (-90, 100, 180)]
    for g in range(cs.desc.contents.n_geoms):
        for p in pts:
            cases.append({"geom": g, "p": list(map(float, p)), "inside": int(L.orc_geom_is_inside(cs.desc, g, orc.vec3(*p)))})
    out["is_inside_csg_stress"] = cases

    # a21-a23 textures
    def texcases(scene, tex, uvs):
        r = []
        for (u, v) in uvs:
            c = (C.c_float * 3)()
            L.orc_tex_color(scene.desc, tex, u, v, c)
            r.append({"tex": tex, "u": u, "v": v, "rgb": list(c)})
        return r
    uvs = [(0.0, 0.0), (4.999, 0.0), (5.0, 0.0), (-0.001, 0.0), (-5.0, -5.0), (-7.5, 12.5), (150.0, -230.0), (1e12, 3.0), (-1e12, -1e12), (282.3243, 0.0)]
    out["tex_checker_lecture4"] = texcases(l4, 0, uvs)
    out["tex_procedure2"] = texcases(l4p, 0, [(0.0, 0.0), (1.0, 2.0), (-105.9, 128.26), (1000.5, -333.25), (3.14159, 2.71828)])
    buv = [(0.0, 0.0), (0.5, 0.5), (0.999999, 0.999999), (0.99999999, 0.5), (0.5, 0.99999999), (1.0, 1.0), (-0.25, 3.75),
           (123.456, -77.7), (0.0012, 0.9988), (float("nan"), 0.5)]
    out["tex_bitmap_floor_scaling"] = texcases(l5, 0, [(u * 200, v * 200) for (u, v) in buv[:9]])
    out["tex_bitmap_world"] = texcases(l5, 1, buv)
    for k in ("tex_bitmap_world", "tex_bitmap_floor_scaling"):
        for c in out[k]:
            c["u"] = None if c["u"] != c["u"] else c["u"]

    # a17 testVisibility on lecture5
    cases = []
    light = (-90.0, 700.0, 350.0)
    for p in [(0, 0, 300), (100, 0, 380), (-100, 0, 120), (100, 100.000001, 320), (-100, 110.000001, 200), (-60, 0, 60), (300, 0, 100), (100, 0, 230)]:
        cases.append({"from": list(map(float, p)), "to": list(light), "visible": int(L.orc_test_visibility(l5.desc, orc.vec3(*p), orc.vec3(*light)))})
    out["test_visibility_lecture5"] = cases

    # a15 shell sort incl. ties (stability behaviour of util.array.sort)
    cases = []
    rng = np.random.RandomState(5)
    for n in (0, 1, 2, 3, 4, 5, 8, 11, 16):
        d = rng.randint(0, 6, size=n).astype(np.float64)
        arr = (orc.OrcHit * max(n, 1))()
        for i in range(n):
            arr[i].dist = d[i]
            arr[i].g = i
        L.orc_shell_sort_hits(arr, n)
        cases.append({"dist": d.tolist(), "order": [arr[i].g for i in range(n)]})
    out["shell_sort"] = cases

    # display encode (rt/color.d toRGB32)
    cases = []
    for rgb in [(0, 0, 0), (1, 1, 1), (0.5, 0.25, 0.75), (0.003, 0.0031308, 0.0032), (-1, 2, 0.999999), (0.2, 0.2, 0.2), (1e-6, 0.9, 0.1)]:
        c = (C.c_float * 3)(*rgb)
        cases.append({"rgb": list(c), "rgb32": int(L.orc_color_to_rgb32(c))})
    out["color_to_rgb32"] = cases

    # build-defined counter RNG (depth of field only)
    out["rng_uniform"] = [{"seed": s, "pixel": p, "tap": t, "sample": k, "dim": d, "u": L.orc_rng_uniform(s, p, t, k, d)}
                          for (s, p, t, k, d) in [(0, 0, 0, 0, 0), (7, 12345, 0, 3, 2), (7, 12345, 4, 24, 3), (2 ** 63 + 5, 2 ** 31, 1, 100, 7)]]
    return out
