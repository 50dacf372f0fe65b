"""Seeded random scene generator (SDL text) for GPU-vs-oracle fuzzing."""
import random


def _v(r, lo, hi, n=3):
    return " ".join("%.6g" % r.uniform(lo, hi) for _ in range(n))


def random_scene_sdl(seed, max_depth=3):
    r = random.Random(seed)
    geoms, names = [], []

    def prim(kind=None):
        kind = kind or r.choice(["Sphere", "Sphere", "Cube", "Cube", "Plane"])
        name = "g%d" % len(names)
        if kind == "Sphere":
            geoms.append('Sphere "%s" { center %s; R %.6g }' % (name, _v(r, -40, 40), r.uniform(5, 30)))
        elif kind == "Cube":
            geoms.append('Cube "%s" { center %s; side %.6g }' % (name, _v(r, -40, 40), r.uniform(8, 45)))
        else:
            geoms.append('Plane "%s" { y %.6g }' % (name, r.uniform(-20, 20)))
        names.append(name)
        return name

    def csg(depth):
        if depth == 0 or r.random() < 0.25:
            return prim(r.choice(["Sphere", "Cube", "Cube", "Sphere", "Plane"]) if r.random() < 0.2 else r.choice(["Sphere", "Cube"]))
        # children first (the loader resolves names at deserialize time)
        left = csg(depth - 1) if r.random() < 0.6 else prim(r.choice(["Sphere", "Cube"]))
        right = csg(depth - 1) if r.random() < 0.6 else prim(r.choice(["Sphere", "Cube"]))
        if r.random() < 0.15:
            right = left                      # Op(a, a)
        name = "g%d" % len(names)
        geoms.append('%s "%s" { left "%s"; right "%s" }' % (r.choice(["CsgUnion", "CsgInter", "CsgDiff", "CsgDiff"]), name, left, right))
        names.append(name)
        return name

    roots = [prim("Plane")]
    for _ in range(r.randint(2, 5)):
        roots.append(csg(r.randint(0, max_depth)))
    textures = ['Checker "chk" { color1 %s; color2 %s; size %.6g }' % (_v(r, 0, 1), _v(r, 0, 1), r.uniform(3, 30)),
                'Procedure2 "proc" { freqU %s; freqV %s; colorU { color %s; color %s; color %s }; colorV { color %s; color %s; color %s } }'
                % (_v(r, 0.01, 0.5), _v(r, 0.01, 0.5), _v(r, 0, 0.5), _v(r, 0, 0.5), _v(r, 0, 0.5), _v(r, 0, 0.5), _v(r, 0, 0.5), _v(r, 0, 0.5)),
                'BitmapTexture "bmp" { file "floor.bmp"; scaling %.6g }' % r.uniform(0.005, 0.05)]
    shaders = ['Lambert "s0" { texture "chk" }', 'Lambert "s1" { texture "proc" }', 'Lambert "s2" { texture "bmp" }',
               'Lambert "s3" { color %s }' % _v(r, 0.1, 1),
               'Phong "s4" { color %s; exponent %.6g; strength %.6g }' % (_v(r, 0.1, 1), r.uniform(2, 90), r.uniform(0.2, 1)),
               'Phong "s5" { texture "bmp"; exponent %.6g }' % r.uniform(5, 40)]
    nodes = []
    for i, g in enumerate(roots):
        xf = ""
        if i > 0 and r.random() < 0.5:
            xf += "; scale %s" % _v(r, 0.5, 2.0)
        if i > 0 and r.random() < 0.3:
            xf += "; rotate %s" % _v(r, 0.7, 1.4)
        if i > 0 and r.random() < 0.6:
            xf += "; translate %s" % _v(r, -30, 30)
        nodes.append('Node "n%d" { geometry "%s"; shader "s%d"%s }' % (i, g, r.randint(0, 5), xf))
    lights = ['PointLight "l%d" { pos %s; color %s; power %.6g }' % (i, _v(r, -150, 150), _v(r, 0.3, 1), r.uniform(8000, 60000))
              for i in range(r.randint(1, 3))]
    lights[0] = 'PointLight "l0" { pos %.6g %.6g %.6g; color 1 1 1; power 50000 }' % (r.uniform(-100, 100), r.uniform(80, 200), r.uniform(-100, 100))
    cam = "Camera { pos %.6g %.6g %.6g; yaw %.6g; pitch %.6g; roll %.6g; fov %.6g }" % (
        r.uniform(-30, 30), r.uniform(30, 90), r.uniform(-160, -90), r.uniform(-15, 15), r.uniform(-35, -10), r.uniform(-5, 5), r.uniform(50, 95))
    return "\n".join([
        "Scene {", '  Name "fuzz%d"' % seed,
        "  GlobalSettings { frameWidth 96; frameHeight 72; AAEnabled false; ambientLightColor %s }" % _v(r, 0, 0.2),
        "  " + cam,
        "  Lights {\n    " + "\n    ".join(lights) + "\n  }",
        "  Geometries {\n    " + "\n    ".join(geoms) + "\n  }",
        "  Textures {\n    " + "\n    ".join(textures) + "\n  }",
        "  Shaders {\n    " + "\n    ".join(shaders) + "\n  }",
        "  Nodes {\n    " + "\n    ".join(nodes) + "\n  }",
        "}", ""])


def many_lights_scene_sdl(seed):
    """Like random_scene_sdl but with 4-6 lights placed all around (also behind and beside the camera):
    exercises the per-light shadow-culling thresholds."""
    r = random.Random(10_000 + seed)
    text = random_scene_sdl(seed, max_depth=2)
    lights = []
    for i in range(r.randint(4, 6)):
        lights.append('PointLight "m%d" { pos %s; color %s; power %.6g }' % (i, _v(r, -250, 250), _v(r, 0.2, 1), r.uniform(5000, 40000)))
    head, rest = text.split("  Lights {\n", 1)
    _, tail = rest.split("\n  }\n", 1)
    return head + "  Lights {\n    " + "\n    ".join(lights) + "\n  }\n" + tail


def planes_scene_sdl(seed):
    """Scenes made of Plane nodes only (the kernel instances that decide a plane's miss before the
    ray is normalised): 1-3 planes, bounded or not, under identity / diagonal scale (negative in
    x or z, non-uniform) / translate / (sometimes) the "rotate" entry the loader applies as a
    scale; lights above, below, level with a plane, absurdly far (1e155, beyond the shortcut's
    bounds) or absurdly close; cameras above, below and looking up at the sky."""
    r = random.Random(20_000 + seed)
    geoms, nodes = [], []
    for i in range(r.randint(1, 3)):
        lim = "; limit %.6g" % r.uniform(40, 400) if r.random() < 0.4 else ""
        geoms.append('Plane "p%d" { y %.6g%s }' % (i, r.choice([0, -0.01, r.uniform(-30, 60)]), lim))
        xf = ""
        k = r.random()
        if k < 0.35:
            sx, sy, sz = (r.choice([-1, 1]) * r.uniform(0.2, 12), r.uniform(0.2, 12), r.choice([-1, 1]) * r.uniform(0.2, 12))
            xf += "; scale %.6g %.6g %.6g" % (sx, sy, sz)
        elif k < 0.45:
            xf += "; scale 10 10 10"
        elif k < 0.5:
            xf += "; scale %.6g %.6g %.6g" % (r.uniform(0.5, 2), -r.uniform(0.5, 2), r.uniform(0.5, 2))   # negative y: not an axis plane
        if r.random() < 0.1:
            xf += "; rotate %s" % _v(r, 0.7, 1.4)
        if r.random() < 0.4:
            xf += "; translate %s" % _v(r, -40, 40)
        nodes.append('Node "n%d" { geometry "p%d"; shader "s%d"%s }' % (i, i, r.randint(0, 5), xf))
    textures = ['Checker "chk" { color1 %s; color2 %s; size %.6g }' % (_v(r, 0, 1), _v(r, 0, 1), r.uniform(3, 30)),
                'Procedure2 "proc" { freqU %s; freqV %s; colorU { color %s; color %s; color %s }; colorV { color %s; color %s; color %s } }'
                % (_v(r, 0.01, 0.5), _v(r, 0.01, 0.5), _v(r, 0, 0.5), _v(r, 0, 0.5), _v(r, 0, 0.5), _v(r, 0, 0.5), _v(r, 0, 0.5), _v(r, 0, 0.5)),
                'BitmapTexture "bmp" { file "floor.bmp"; scaling %.6g }' % r.uniform(0.005, 0.05)]
    shaders = ['Lambert "s0" { texture "chk" }', 'Lambert "s1" { texture "proc" }', 'Lambert "s2" { texture "bmp" }',
               'Lambert "s3" { color %s }' % _v(r, 0.1, 1),
               'Phong "s4" { color %s; exponent %.6g; strength %.6g }' % (_v(r, 0.1, 1), r.uniform(2, 90), r.uniform(0.2, 1)),
               'Phong "s5" { texture "bmp"; exponent %.6g }' % r.uniform(5, 40)]
    lights = []
    for i in range(r.randint(1, 3)):
        k = r.random()
        if k < 0.5:
            pos = "%.6g %.6g %.6g" % (r.uniform(-200, 200), r.uniform(20, 300), r.uniform(-200, 200))
        elif k < 0.65:
            pos = "%.6g %.6g %.6g" % (r.uniform(-200, 200), -r.uniform(20, 300), r.uniform(-200, 200))      # below
        elif k < 0.75:
            pos = "%.6g %.6g %.6g" % (r.uniform(-200, 200), r.choice([0, -0.01, 1e-160, -1e-160]), r.uniform(-200, 200))  # level
        elif k < 0.85:
            pos = "%s %s %s" % (r.choice(["1e155", "-3e160", "5"]), r.choice(["1e155", "2e151", "7e149"]), r.choice(["-1e155", "40"]))
        elif k < 0.93:
            pos = "%.6g %s %.6g" % (r.uniform(-50, 50), r.choice(["1e-155", "3e-151", "1e-149"]), r.uniform(-50, 50))
        else:
            pos = "1e308 1e308 -1e308"
        lights.append('PointLight "l%d" { pos %s; color %s; power %s }' % (i, pos, _v(r, 0.3, 1), r.choice(["50000", "1e300", "8000"])))
    if r.random() < 0.25:   # below the plane(s), or looking up
        cam = "Camera { pos %.6g %.6g %.6g; yaw %.6g; pitch %.6g; roll %.6g; fov %.6g }" % (
            r.uniform(-30, 30), -r.uniform(5, 90), r.uniform(-160, -90), r.uniform(-15, 15), r.uniform(-10, 40), r.uniform(-5, 5), r.uniform(50, 95))
    else:
        cam = "Camera { pos %.6g %.6g %.6g; yaw %.6g; pitch %.6g; roll %.6g; fov %.6g }" % (
            r.uniform(-30, 30), r.uniform(3, 120), r.uniform(-160, -20), r.uniform(-25, 25), r.uniform(-60, 15), r.uniform(-8, 8), r.uniform(40, 100))
    return "\n".join([
        "Scene {", '  Name "planes%d"' % seed,
        "  GlobalSettings { frameWidth 96; frameHeight 72; AAEnabled %s; ambientLightColor %s }" % (r.choice(["false", "true"]), _v(r, 0, 0.2)),
        "  " + cam,
        "  Lights {\n    " + "\n    ".join(lights) + "\n  }",
        "  Geometries {\n    " + "\n    ".join(geoms) + "\n  }",
        "  Textures {\n    " + "\n    ".join(textures) + "\n  }",
        "  Shaders {\n    " + "\n    ".join(shaders) + "\n  }",
        "  Nodes {\n    " + "\n    ".join(nodes) + "\n  }",
        "}", ""])


def ground_scene_sdl(seed):
    """One ground plane (identity node 0) plus 2-6 small objects above, below, resting on or cutting
    through it; light 0 above, grazing, below the ground, among or inside the objects; camera above
    or below the ground: exercises the ground-plane shadow rectangles (projection of each node's box
    from the light onto the plane) against the literal shadow rays."""
    r = random.Random(30_000 + seed)
    gy = r.choice([0, -0.01, r.uniform(-10, 10)])
    lim = "; limit %.6g" % r.uniform(150, 600) if r.random() < 0.2 else ""
    geoms = ['Plane "ground" { y %.6g%s }' % (gy, lim)]
    nodes = ['Node "n0" { geometry "ground"; shader "s%d" }' % r.choice([0, 1, 2, 3])]
    for i in range(r.randint(2, 6)):
        kind = r.choice(["Sphere", "Cube", "Csg"])
        size = r.uniform(4, 25)
        k = r.random()
        cy = gy + (size * r.uniform(1.0, 4.0) if k < 0.55 else size * r.uniform(-0.5, 1.0) if k < 0.85 else -size * r.uniform(1.0, 3.0))
        c = "%.6g %.6g %.6g" % (r.uniform(-70, 70), cy, r.uniform(-40, 90))
        if kind == "Sphere":
            geoms.append('Sphere "g%d" { center %s; R %.6g }' % (i, c, size))
        elif kind == "Cube":
            geoms.append('Cube "g%d" { center %s; side %.6g }' % (i, c, 2 * size))
        else:
            geoms.append('Cube "g%da" { center %s; side %.6g }' % (i, c, 2 * size))
            geoms.append('Sphere "g%db" { center %s; R %.6g }' % (i, c, size * r.uniform(0.9, 1.35)))
            geoms.append('%s "g%d" { left "g%da"; right "g%db" }' % (r.choice(["CsgDiff", "CsgInter", "CsgUnion"]), i, i, i))
        xf = ""
        if r.random() < 0.3:
            xf += "; scale %s" % _v(r, 0.6, 1.8)
        if r.random() < 0.3:
            xf += "; translate %s" % _v(r, -25, 25)
        nodes.append('Node "n%d" { geometry "g%d"; shader "s%d"%s }' % (i + 1, i, r.randint(0, 5), xf))
    textures = ['Checker "chk" { color1 %s; color2 %s; size %.6g }' % (_v(r, 0, 1), _v(r, 0, 1), r.uniform(3, 30)),
                'Procedure2 "proc" { freqU %s; freqV %s; colorU { color %s; color %s; color %s }; colorV { color %s; color %s; color %s } }'
                % (_v(r, 0.01, 0.5), _v(r, 0.01, 0.5), _v(r, 0, 0.5), _v(r, 0, 0.5), _v(r, 0, 0.5), _v(r, 0, 0.5), _v(r, 0, 0.5), _v(r, 0, 0.5)),
                'BitmapTexture "bmp" { file "floor.bmp"; scaling %.6g }' % r.uniform(0.005, 0.05)]
    shaders = ['Lambert "s0" { texture "chk" }', 'Lambert "s1" { texture "proc" }', 'Lambert "s2" { texture "bmp" }',
               'Lambert "s3" { color %s }' % _v(r, 0.1, 1),
               'Phong "s4" { color %s; exponent %.6g; strength %.6g }' % (_v(r, 0.1, 1), r.uniform(2, 90), r.uniform(0.2, 1)),
               'Phong "s5" { texture "bmp"; exponent %.6g }' % r.uniform(5, 40)]
    k = r.random()
    if k < 0.4:
        lpos = "%.6g %.6g %.6g" % (r.uniform(-200, 200), gy + r.uniform(60, 400), r.uniform(-200, 300))
    elif k < 0.6:
        lpos = "%.6g %.6g %.6g" % (r.uniform(-300, 300), gy + r.uniform(0.5, 12), r.uniform(-300, 300))     # grazing
    elif k < 0.75:
        lpos = "%.6g %.6g %.6g" % (r.uniform(-60, 60), gy + r.uniform(5, 60), r.uniform(-30, 80))           # among / inside the objects
    elif k < 0.9:
        lpos = "%.6g %.6g %.6g" % (r.uniform(-200, 200), gy - r.uniform(1, 200), r.uniform(-200, 300))      # below the ground
    else:
        lpos = "%.6g %.6g %.6g" % (r.uniform(-200, 200), gy, r.uniform(-200, 300))                          # in the plane
    lights = ['PointLight "l0" { pos %s; color 1 1 1; power %.6g }' % (lpos, r.uniform(8000, 90000))]
    for i in range(r.randint(0, 2)):
        lights.append('PointLight "l%d" { pos %s; color %s; power %.6g }' % (i + 1, _v(r, -150, 150), _v(r, 0.3, 1), r.uniform(8000, 60000)))
    if r.random() < 0.2:
        cam = "Camera { pos %.6g %.6g %.6g; yaw %.6g; pitch %.6g; roll %.6g; fov %.6g }" % (
            r.uniform(-30, 30), gy - r.uniform(5, 60), r.uniform(-160, -90), r.uniform(-15, 15), r.uniform(-5, 35), r.uniform(-5, 5), r.uniform(50, 95))
    else:
        cam = "Camera { pos %.6g %.6g %.6g; yaw %.6g; pitch %.6g; roll %.6g; fov %.6g }" % (
            r.uniform(-40, 40), gy + r.uniform(3, 120), r.uniform(-180, -60), r.uniform(-25, 25), r.uniform(-50, 5), r.uniform(-8, 8), r.uniform(40, 100))
    return "\n".join([
        "Scene {", '  Name "ground%d"' % seed,
        "  GlobalSettings { frameWidth 96; frameHeight 72; AAEnabled %s; ambientLightColor %s }" % (r.choice(["false", "false", "true"]), _v(r, 0, 0.2)),
        "  " + cam,
        "  Lights {\n    " + "\n    ".join(lights) + "\n  }",
        "  Geometries {\n    " + "\n    ".join(geoms) + "\n  }",
        "  Textures {\n    " + "\n    ".join(textures) + "\n  }",
        "  Shaders {\n    " + "\n    ".join(shaders) + "\n  }",
        "  Nodes {\n    " + "\n    ".join(nodes) + "\n  }",
        "}", ""])


def many_nodes_scene_sdl(seed, n_objects):
    """A ground plane plus `n_objects` small spheres / cubes / depth-1 CSG objects scattered over it (every
    third one under a scaled / translated node): scenes whose node count crosses the 32-node width of the
    per-tile culling masks (kMaxCullNodes) — nodes 32.. are always tested, nodes below it may be culled."""
    r = random.Random(40_000 + 1000 * n_objects + seed)
    geoms = ['Plane "ground" { y 0 }']
    nodes = ['Node "n0" { geometry "ground"; shader "s%d" }' % r.choice([0, 2, 3])]
    for i in range(n_objects):
        kind = r.choice(["Sphere", "Sphere", "Cube", "Csg"])
        size = r.uniform(3, 9)
        c = "%.6g %.6g %.6g" % (r.uniform(-110, 110), r.uniform(0.4, 2.5) * size, r.uniform(-30, 190))
        if kind == "Sphere":
            geoms.append('Sphere "g%d" { center %s; R %.6g }' % (i, c, size))
        elif kind == "Cube":
            geoms.append('Cube "g%d" { center %s; side %.6g }' % (i, c, 2 * size))
        else:
            geoms.append('Cube "g%da" { center %s; side %.6g }' % (i, c, 2 * size))
            geoms.append('Sphere "g%db" { center %s; R %.6g }' % (i, c, size * r.uniform(0.9, 1.35)))
            geoms.append('%s "g%d" { left "g%da"; right "g%db" }' % (r.choice(["CsgDiff", "CsgInter", "CsgUnion"]), i, i, i))
        xf = ""
        if i % 3 == 1:
            xf += "; scale %s" % _v(r, 0.7, 1.5)
        if i % 3 == 2:
            xf += "; translate %s" % _v(r, -12, 12)
        nodes.append('Node "n%d" { geometry "g%d"; shader "s%d"%s }' % (i + 1, i, r.randint(0, 5), xf))
    textures = ['Checker "chk" { color1 %s; color2 %s; size %.6g }' % (_v(r, 0, 1), _v(r, 0, 1), r.uniform(6, 30)),
                'Procedure2 "proc" { freqU %s; freqV %s; colorU { color %s; color %s; color %s }; colorV { color %s; color %s; color %s } }'
                % (_v(r, 0.01, 0.5), _v(r, 0.01, 0.5), _v(r, 0, 0.5), _v(r, 0, 0.5), _v(r, 0, 0.5), _v(r, 0, 0.5), _v(r, 0, 0.5), _v(r, 0, 0.5)),
                'BitmapTexture "bmp" { file "floor.bmp"; scaling %.6g }' % r.uniform(0.005, 0.05)]
    shaders = ['Lambert "s0" { texture "chk" }', 'Lambert "s1" { texture "proc" }', 'Lambert "s2" { texture "bmp" }',
               'Lambert "s3" { color %s }' % _v(r, 0.1, 1),
               'Phong "s4" { color %s; exponent %.6g; strength %.6g }' % (_v(r, 0.1, 1), r.uniform(2, 90), r.uniform(0.2, 1)),
               'Phong "s5" { texture "bmp"; exponent %.6g }' % r.uniform(5, 40)]
    lights = ['PointLight "l0" { pos %.6g %.6g %.6g; color 1 1 1; power 90000 }' % (r.uniform(-150, 150), r.uniform(90, 300), r.uniform(-100, 200))]
    if seed % 2:
        lights.append('PointLight "l1" { pos %s; color %s; power 30000 }' % (_v(r, -150, 150), _v(r, 0.3, 1)))
    cam = "Camera { pos %.6g %.6g %.6g; yaw %.6g; pitch %.6g; roll %.6g; fov %.6g }" % (
        r.uniform(-20, 20), r.uniform(40, 110), r.uniform(-150, -80), r.uniform(-12, 12), r.uniform(-40, -15), r.uniform(-4, 4), r.uniform(55, 90))
    return "\n".join([
        "Scene {", '  Name "many%d_%d"' % (n_objects, seed),
        "  GlobalSettings { frameWidth 320; frameHeight 180; AAEnabled false; ambientLightColor %s }" % _v(r, 0, 0.2),
        "  " + cam,
        "  Lights {\n    " + "\n    ".join(lights) + "\n  }",
        "  Geometries {\n    " + "\n    ".join(geoms) + "\n  }",
        "  Textures {\n    " + "\n    ".join(textures) + "\n  }",
        "  Shaders {\n    " + "\n    ".join(shaders) + "\n  }",
        "  Nodes {\n    " + "\n    ".join(nodes) + "\n  }",
        "}", ""])
