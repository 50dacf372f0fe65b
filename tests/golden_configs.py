"""Frame configurations shared by the golden generator and the tests."""
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SCENES = os.path.join(ROOT, "tests", "golden", "scenes")

# name: scene file, width, height, taps, dof (1 = as the file says), seed
CONFIGS = {
    "lecture4_640x480_t1": dict(scene="lecture4.sdl", w=640, h=480, taps=1, dof=0),
    "lecture4_640x480_t5": dict(scene="lecture4.sdl", w=640, h=480, taps=5, dof=0),
    "lecture4json_640x480_t1": dict(scene="lecture4.json", w=640, h=480, taps=1, dof=0),
    "lecture4proc_640x480_t1": dict(scene="lecture4-proc-texture.sdl", w=640, h=480, taps=1, dof=0),
    "lecture5_640x480_t1": dict(scene="lecture5.sdl", w=640, h=480, taps=1, dof=0),
    "lecture5_640x480_t5": dict(scene="lecture5.sdl", w=640, h=480, taps=5, dof=0),
    "lecture5_333x217_t4": dict(scene="lecture5.sdl", w=333, h=217, taps=4, dof=0),   # ragged: not a tile multiple
    "zaphod_645x430_t1": dict(scene="zaphod.sdl", w=645, h=430, taps=1, dof=0),
    "zaphod_645x430_t4": dict(scene="zaphod.sdl", w=645, h=430, taps=4, dof=0),
    "zaphod_215x143_dof25": dict(scene="zaphod.sdl", w=215, h=143, taps=1, dof=1, seed=7),  # as shipped: 25 DOF samples, build RNG
    "csg_stress_320x240_t1": dict(scene="csg_stress.sdl", w=320, h=240, taps=1, dof=0),
    "csg_stress_320x240_t5": dict(scene="csg_stress.sdl", w=320, h=240, taps=5, dof=0),
    "csg_corner_256x192_t1": dict(scene="csg_corner.sdl", w=256, h=192, taps=1, dof=0),
}


def crop_offsets(w, h):
    xs = [0, max(0, w // 2 - 32), max(0, w - 64), max(0, w // 4)]
    ys = [0, max(0, h // 2 - 32), max(0, h - 64), max(0, (3 * h) // 4 - 32)]
    return list(zip(xs, ys))


def load_config(name, **opt_overrides):
    import chess2rt_amd as c2

    cfg = CONFIGS[name]
    scene = c2.parseSceneFromFile(os.path.join(SCENES, cfg["scene"]))
    scene.setFrameSize(cfg["w"], cfg["h"])
    scene.setDof(bool(cfg["dof"]))
    cam = scene.beginFrame()
    kw = dict(taps=cfg["taps"], seed=cfg.get("seed", 0))
    kw.update(opt_overrides)
    opts = scene.renderOpts(**kw)
    return scene, cam, opts
