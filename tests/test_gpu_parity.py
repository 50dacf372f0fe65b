"""Parity tests proper: the HIP path (through the C ABI) against the oracle
on the same inputs, against the committed golden fixtures, and — at
BASELINE.json's full sizes — through size-independent properties.

Tolerance (BASELINE.json north_star): |gpu - reference| <= 1e-4 per RGB
channel on every pixel.  Geometry is fp64 in reference order, so in practice
the frames are bit-identical except where device libm (atan2/asin/sin/cos/
pow) differs from glibc by an ulp; the tests also report the mismatch count.
"""
import ctypes as C
import json
import os

import numpy as np
import pytest

import chess2rt_amd as c2
import oracle_lib as orc
from chess2rt_amd import _abi
from golden_configs import CONFIGS, SCENES, crop_offsets, load_config
from parity_util import TOL, maxdiff  # TOL = 1e-4 per RGB channel, BASELINE.json; one-sided NaN / inf fail

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", sorted(CONFIGS))
def test_frame_matches_oracle_and_golden(name, gpu_ctx, golden_dir):
    scene, cam, opts = load_config(name, count_rays=1)
    gpu_ctx.uploadScene(scene.desc)
    gpu = gpu_ctx.renderFrame(cam, opts)
    primary, shadow = gpu_ctx.rayStats()
    stats = {}
    ref = orc.render_frame(scene.desc, cam, opts, 0, stats)
    md, nbad, nne = maxdiff(gpu, ref)
    print("%s: max|d|=%.3g, >1e-4: %d, !=: %d of %d" % (name, md, nbad, nne, gpu.size))
    assert gpu.shape == ref.shape == (opts.height, opts.width, 3)
    assert md <= TOL and nbad == 0
    assert (primary, shadow) == (stats["primary"], stats["shadow"])
    # committed fixtures
    e = json.load(open(os.path.join(golden_dir, "frames.json")))[name]
    crops = np.load(os.path.join(golden_dir, "frames_crops.npz"))
    assert (primary, shadow) == (e["primary_rays"], e["shadow_rays"])
    for k, (x0, y0) in enumerate(crop_offsets(opts.width, opts.height)):
        np.testing.assert_allclose(gpu[y0:y0 + 64, x0:x0 + 64], crops["%s/%d" % (name, k)], rtol=0, atol=TOL)


@pytest.mark.parametrize("scene_file,size", [("lecture4.sdl", (1920, 1080)), ("lecture5.sdl", (1920, 1080)), ("lecture5.sdl", (1001, 563))])
def test_baseline_1080p_configs_bit_level(scene_file, size, gpu_ctx):
    """BASELINE configs C2/C3 at full size against the oracle (it finishes in well under a minute)."""
    s = c2.parseSceneFromFile(os.path.join(SCENES, scene_file))
    s.setFrameSize(*size)
    s.setAA(False)
    cam = s.beginFrame()
    opts = s.renderOpts()
    gpu_ctx.uploadScene(s.desc)
    gpu = gpu_ctx.renderFrame(cam, opts)
    ref = orc.render_frame(s.desc, cam, opts, 0)
    md, nbad, nne = maxdiff(gpu, ref)
    print("%s %s: max|d|=%.3g, !=: %d" % (scene_file, size, md, nne))
    assert md <= TOL and nbad == 0


def test_probe_matches_oracle_trace_results(gpu_ctx):
    """renderPixel (rt/renderer.d:46-57): p, normal, dist, u, v, node, leaf per pixel."""
    for name in ("lecture5_640x480_t1", "csg_stress_320x240_t1", "zaphod_645x430_t1"):
        scene, cam, opts = load_config(name)
        gpu_ctx.uploadScene(scene.desc)
        rng = np.random.RandomState(3)
        pts = [(int(rng.randint(0, opts.width)), int(rng.randint(0, opts.height))) for _ in range(120)]
        pts += [(0, 0), (opts.width - 1, opts.height - 1), (opts.width // 2, opts.height // 2)]
        for (x, y) in pts:
            g = gpu_ctx.renderPixel(cam, opts, x, y)
            r = orc.render_pixel(scene.desc, cam, opts, x, y)
            assert g.closest_node == r.closest_node and g.leaf_geom == r.leaf_geom, (name, x, y)
            np.testing.assert_allclose(list(g.color), list(r.color), rtol=0, atol=TOL)
            assert list(g.ray_orig) == list(r.ray_orig) and list(g.ray_dir) == list(r.ray_dir)
            if r.closest_node >= 0:
                assert g.dist == r.dist and list(g.p) == list(r.p), (name, x, y)
                np.testing.assert_allclose(list(g.normal), list(r.normal), rtol=0, atol=1e-15)
                np.testing.assert_allclose([g.u, g.v], [r.u, r.v], rtol=0, atol=1e-12)


def test_renderer_api_mirrors_the_reference(gpu_ctx):
    """Renderer.renderRT / renderSceneAsync / renderPixel through the host mirror."""
    scene = c2.parseSceneFromFile(os.path.join(SCENES, "lecture5.sdl"))
    scene.setFrameSize(200, 150)                         # AAEnabled true in the file -> 5 taps
    r = c2.Renderer(scene, gpu_ctx)
    img = r.renderRT()
    cam = scene.beginFrame()
    ref = orc.render_frame(scene.desc, cam, scene.renderOpts())
    assert scene.renderOpts().taps == c2.TAPS_REF5
    assert maxdiff(img, ref)[0] <= TOL
    # async: isRendering is cleared when the frame is complete
    out = np.full((150, 200, 3), -1.0, np.float32)
    is_rendering = np.ones(1, np.uint8)
    needs = np.zeros(1, np.uint8)
    r.renderSceneAsync(out, is_rendering, needs)
    r.wait()
    assert is_rendering[0] == 0 and np.array_equal(out, img)
    # a raised stop flag cancels before the pass (isStopReq, rt/renderer.d:129)
    needs[0] = 1
    with pytest.raises(c2.C2rtError) as e:
        r.renderRT(stop_flag=needs)
    assert e.value.status == _abi.ERR_CANCELLED
    tr = r.renderPixelNoAA(100, 100)
    rr = orc.render_pixel(scene.desc, cam, scene.renderOpts(), 100, 100)
    np.testing.assert_allclose(list(tr.color), list(rr.color), rtol=0, atol=TOL)
    assert tr.closest_node == rr.closest_node


def test_strip_sharded_render_equals_full_render(gpu_ctx):
    """Multi-GPU decomposition on one GPU: every rank's strips, gathered and
    de-interleaved by the HIP copy kernel, are the full frame bit for bit."""
    import torch

    scene, cam, opts = load_config("lecture5_333x217_t4")
    gpu_ctx.uploadScene(scene.desc)
    W, H = opts.width, opts.height
    full = gpu_ctx.renderFrame(cam, opts)
    for world in (2, 3, 8):
        plan = c2.plan_strips(H, world, 8)
        gathered = torch.zeros((world, plan.rows_pad, W, 3), dtype=torch.float32, device="cuda")
        for r in range(world):
            _, _, o = load_config("lecture5_333x217_t4", strip_height=8, strip_rank=r, strip_world=world)
            rows = gpu_ctx.localRows(o)
            assert rows == c2.local_rows(H, 8, r, world) <= plan.rows_pad
            gpu_ctx.renderFrameDevice(cam, o, gathered[r].data_ptr(), torch.cuda.current_stream().cuda_stream)
            host = gpu_ctx.renderFrame(cam, o)               # same strips through the host-output entry point
            assert np.array_equal(host, full[[y for y in range(H) if (y // 8) % world == r]])
        frame = torch.empty((H, W, 3), dtype=torch.float32, device="cuda")
        gpu_ctx.deinterleaveStrips(gathered.data_ptr(), frame.data_ptr(), W, H, 8, world, torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        assert np.array_equal(frame.cpu().numpy(), full)
        assert torch.equal(c2.deinterleave_strips_torch(gathered.cpu(), H, 8, world), frame.cpu())


def test_display_encode_matches_reference_table(gpu_ctx):
    import torch

    scene, cam, opts = load_config("lecture5_640x480_t1")
    gpu_ctx.uploadScene(scene.desc)
    img = gpu_ctx.renderFrame(cam, opts)
    img[0, 0] = (-1.0, 2.0, np.nan)
    img[0, 1] = (0.003, 0.0031308, 1.0)
    t = torch.from_numpy(img).cuda()
    out = torch.empty((opts.height, opts.width), dtype=torch.int32, device="cuda")
    gpu_ctx.encodeRGB32(t.data_ptr(), out.data_ptr(), opts.height * opts.width, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    got = out.cpu().numpy().astype(np.uint32)
    L = orc.lib()
    rng = np.random.RandomState(1)
    for _ in range(3000):
        y, x = int(rng.randint(0, opts.height)), int(rng.randint(0, opts.width))
        c = (C.c_float * 3)(*img[y, x])
        assert int(got[y, x]) == L.orc_color_to_rgb32(c)
    for (y, x) in [(0, 0), (0, 1)]:
        c = (C.c_float * 3)(*img[y, x])
        assert int(got[y, x]) == L.orc_color_to_rgb32(c)


def test_full_size_properties_4k_and_8k(gpu_ctx):
    """At BASELINE's full sizes the oracle is too slow for every test run, so
    check properties that do not depend on the size: idempotence, strips union
    == frame, sampled pixels against the oracle's pixel probe, ray counts."""
    s = c2.parseSceneFromFile(os.path.join(SCENES, "lecture5.sdl"))
    for (w, h, taps) in [(3840, 2160, 1), (7680, 4320, 4)]:
        s.setFrameSize(w, h)
        cam = s.beginFrame()
        opts = s.renderOpts(taps=taps, count_rays=1)
        gpu_ctx.uploadScene(s.desc)
        a = gpu_ctx.renderFrame(cam, opts)
        primary, shadow = gpu_ctx.rayStats()
        assert primary == w * h * taps and 0 < shadow <= primary
        b = gpu_ctx.renderFrame(cam, opts)
        assert np.array_equal(a, b)                                   # deterministic / idempotent
        assert np.isfinite(a).all() and a.min() >= 0
        part = gpu_ctx.renderFrame(cam, s.renderOpts(taps=taps, strip_height=8, strip_rank=3, strip_world=8))
        rows = [y for y in range(h) if (y // 8) % 8 == 3]
        assert np.array_equal(part, a[rows])
        if taps == 1:
            rng = np.random.RandomState(11)
            for _ in range(400):
                x, y = int(rng.randint(0, w)), int(rng.randint(0, h))
                r = orc.render_pixel(s.desc, cam, opts, x, y)
                np.testing.assert_allclose(a[y, x], list(r.color), rtol=0, atol=TOL)
        # linearity of the AA taps: the 4-tap frame is the /4 average of four shifted 1-tap frames
        if taps == 4:
            sub = s.renderOpts(taps=4, strip_height=8, strip_rank=100, strip_world=540)   # one 8-row strip
            four = gpu_ctx.renderFrame(cam, sub)
            ref = orc.render_frame(s.desc, cam, sub, 0)
            assert maxdiff(four, ref)[0] <= TOL


def test_dof_with_counter_rng_matches_oracle(gpu_ctx):
    scene, cam, opts = load_config("zaphod_215x143_dof25")
    assert cam.dof == 1 and cam.num_samples == 25
    gpu_ctx.uploadScene(scene.desc)
    a = gpu_ctx.renderFrame(cam, opts)
    ref = orc.render_frame(scene.desc, cam, opts, 0)
    assert maxdiff(a, ref)[0] <= TOL
    _, _, o2 = load_config("zaphod_215x143_dof25", seed=8)
    assert not np.array_equal(gpu_ctx.renderFrame(cam, o2), a)         # the seed matters
    # stereo (combineStereo, rt/color.d:10-15) through the same kernel family
    cam.dof = 0
    cam.stereo_separation = 0.5
    a = gpu_ctx.renderFrame(cam, opts)
    ref = orc.render_frame(scene.desc, cam, opts, 0)
    assert maxdiff(a, ref)[0] <= TOL
    # every camera mode through every kernel family: mono / stereo x depth of field on / off, on a
    # planes-only scene, a depth-1 CSG scene with one light and a depth-4 CSG scene with several lights
    for name in ("zaphod_215x143_dof25", "lecture5_333x217_t4", "csg_stress_320x240_t1"):
        scene, cam, opts = load_config(name, count_rays=1, seed=3)
        scene_cam = cam
        gpu_ctx.uploadScene(scene.desc)
        for dof, sep in ((1, 0.0), (1, 0.7), (0, 0.7)):
            scene_cam.dof = dof
            scene_cam.num_samples = 6
            scene_cam.focal_plane_dist = 180.0
            scene_cam.disc_multiplier = 10.0 / 4.0
            scene_cam.stereo_separation = sep
            a = gpu_ctx.renderFrame(scene_cam, opts)
            rays = gpu_ctx.rayStats()
            st = {}
            ref = orc.render_frame(scene.desc, scene_cam, opts, 0, st)
            md, nbad, nne = maxdiff(a, ref)
            assert md <= TOL and nbad == 0, (name, dof, sep, md)
            assert rays == (st["primary"], st["shadow"]), (name, dof, sep)


def test_error_behaviour(scenes_dir):
    ctx = c2.Context(0)
    s = c2.parseSceneFromFile(os.path.join(scenes_dir, "lecture4.sdl"))
    cam = s.beginFrame()
    with pytest.raises(c2.C2rtError) as e:
        ctx.renderFrame(cam, s.renderOpts())
    assert e.value.status == _abi.ERR_NO_SCENE
    d = s.desc.contents
    d.gi_enabled = 1
    with pytest.raises(c2.C2rtError) as e:
        ctx.uploadScene(s.desc)
    assert e.value.status == _abi.ERR_UNSUPPORTED
    d.gi_enabled = 0
    d.abi_version = 99
    with pytest.raises(c2.C2rtError) as e:
        ctx.uploadScene(s.desc)
    assert e.value.status == _abi.ERR_INVALID_ARG
    d.abi_version = _abi.ABI_VERSION
    ctx.uploadScene(s.desc)
    for bad in (dict(width=0), dict(taps=3), dict(strip_world=2, strip_rank=2), dict(height=1 << 17)):
        with pytest.raises(c2.C2rtError) as e:
            ctx.renderFrame(cam, s.renderOpts(**bad))
        assert e.value.status == _abi.ERR_INVALID_ARG
    ctx.close()


def test_csg_depth_limit_and_cycles_are_rejected(gpu_ctx):
    n = 8
    d = _abi.SceneDesc()
    d.abi_version = _abi.ABI_VERSION
    d.n_geoms = n
    types = (C.c_int32 * n)(_abi.GEOM_SPHERE, _abi.GEOM_SPHERE, *([_abi.GEOM_CSG_UNION] * 6))
    params = (C.c_double * (4 * n))(*([0, 0, 5, 1] * n))
    ch = (C.c_int32 * (2 * n))(-1, -1, -1, -1, 0, 1, 0, 2, 0, 3, 0, 4, 0, 5, 0, 6)   # chain: depth 1..6
    d.geom_type, d.geom_param, d.geom_child = types, params, ch
    d.n_shaders = 1
    st, sc, stx, se, ss = (C.c_int32 * 1)(0), (C.c_float * 3)(1, 1, 1), (C.c_int32 * 1)(-1), (C.c_double * 1)(16), (C.c_float * 1)(1)
    d.shader_type, d.shader_color, d.shader_texture, d.shader_exponent, d.shader_strength = st, sc, stx, se, ss
    d.n_nodes = 1
    ng, ns, nb = (C.c_int32 * 1)(5), (C.c_int32 * 1)(0), (C.c_int32 * 1)(-1)
    I = [1, 0, 0, 0, 1, 0, 0, 0, 1]
    nt = (C.c_double * 30)(*(I + I + I + [0, 0, 0]))
    d.node_geom, d.node_shader, d.node_bump, d.node_transform = ng, ns, nb, nt
    gpu_ctx.uploadScene(d)                      # depth 4: accepted
    ng[0] = 6                                   # depth 5
    with pytest.raises(c2.C2rtError) as e:
        gpu_ctx.uploadScene(d)
    assert e.value.status == _abi.ERR_LIMIT
    ng[0] = 5
    ch[4] = 5                                   # geometry 2's left child -> 5 : a cycle
    with pytest.raises(c2.C2rtError) as e:
        gpu_ctx.uploadScene(d)
    assert e.value.status == _abi.ERR_INVALID_ARG
    ng[0] = 42
    with pytest.raises(c2.C2rtError) as e:
        gpu_ctx.uploadScene(d)
    assert e.value.status == _abi.ERR_INVALID_ARG
    # hit lists tag their leaves in 12 bits: a scene WITH CsgOps may hold 4096 geometries, one without any number
    big = 4097
    types2 = (C.c_int32 * big)(*([_abi.GEOM_SPHERE] * (big - 1) + [_abi.GEOM_CSG_UNION]))
    params2 = (C.c_double * (4 * big))(*([0, 0, 5, 1] * big))
    ch2 = (C.c_int32 * (2 * big))(*([-1, -1] * (big - 1) + [0, 1]))
    d.n_geoms, d.geom_type, d.geom_param, d.geom_child = big, types2, params2, ch2
    ng[0] = big - 1
    with pytest.raises(c2.C2rtError) as e:
        gpu_ctx.uploadScene(d)
    assert e.value.status == _abi.ERR_LIMIT
    ng[0] = 7
    types2[big - 1] = _abi.GEOM_SPHERE
    ch2[2 * (big - 1)] = ch2[2 * (big - 1) + 1] = -1
    gpu_ctx.uploadScene(d)                      # no CsgOp left: accepted


def _load_text(tmp_path, text, name="s.sdl"):
    p = tmp_path / name
    p.write_text(text)
    return c2.parseSceneFromFile(str(p))


@pytest.mark.parametrize("w,h", [(1, 1), (7, 3), (8, 8), (9, 17), (64, 1), (1, 70)])
def test_edge_frame_sizes(w, h, gpu_ctx):
    scene, _, _ = load_config("lecture5_640x480_t1")
    scene.setFrameSize(w, h)
    cam = scene.beginFrame()
    for taps in (1, 5):
        opts = scene.renderOpts(taps=taps)
        gpu_ctx.uploadScene(scene.desc)
        a = gpu_ctx.renderFrame(cam, opts)
        ref = orc.render_frame(scene.desc, cam, opts, 1)
        assert a.shape == (h, w, 3) and maxdiff(a, ref)[0] <= TOL


def test_empty_and_degenerate_scenes(gpu_ctx, tmp_path):
    cases = {
        # no nodes at all: every ray returns the (black) environment
        "empty": 'Scene { Camera { pos 0 1 0; fov 90 } }',
        # geometry but no lights: ambient only
        "nolight": '''Scene { GlobalSettings { ambientLightColor 0.3 0.2 0.1 }
            Camera { pos 0 5 -10; pitch -20; fov 80 }
            Geometries { Plane "p" { y 0 }; Sphere "s" { center 0 2 5; R 2 } }
            Shaders { Lambert "l" { color 0.5 0.5 0.5 } }
            Nodes { Node "a" { geometry "p"; shader "l" }; Node "b" { geometry "s"; shader "l" } } }''',
        # a light whose colour * power has zero intensity casts no shadow ray (rt/shader.d:88)
        "darklight": '''Scene { Camera { pos 0 5 -10; pitch -20; fov 80 }
            Lights { PointLight "z" { pos 0 10 0; color 0 0 0; power 100 }; PointLight "n" { pos 3 10 0; color 1 1 1; power 0 } }
            Geometries { Plane "p" { y 0 } }
            Shaders { Phong "l" { color 0.5 0.5 0.5 } }
            Nodes { Node "a" { geometry "p"; shader "l" } } }''',
        # Plane() without y: y is NaN, every comparison fails -> never... (reference semantics, whatever they are)
        "nanplane": '''Scene { Camera { pos 0 5 -10; pitch -20; fov 80 }
            Lights { PointLight "k" { pos 0 10 0; color 1 1 1; power 500 } }
            Geometries { Plane "p" { }; Cube "c" { center 0 1 4; side 2 } }
            Shaders { Lambert "l" { } }
            Nodes { Node "a" { geometry "p"; shader "l" }; Node "b" { geometry "c"; shader "l"; scale 1 0 1 } } }''',
        # far-away geometry and huge texture coordinates (cast(int) overflow path of Checker)
        "huge": '''Scene { Camera { pos 0 1e7 0; pitch -89; fov 120 }
            Lights { PointLight "k" { pos 0 2e7 0; color 1 1 1; power 1e16 } }
            Geometries { Plane "p" { y 0 } }
            Textures { Checker "c" { color1 1 0 0; color2 0 1 0; size 0.001 } }
            Shaders { Lambert "l" { texture "c" } }
            Nodes { Node "a" { geometry "p"; shader "l" } } }''',
    }
    for name, text in cases.items():
        scene = _load_text(tmp_path, text, name + ".sdl")
        scene.setFrameSize(96, 64)
        scene.setAA(False)
        cam = scene.beginFrame()
        opts = scene.renderOpts(count_rays=1)
        gpu_ctx.uploadScene(scene.desc)
        a = gpu_ctx.renderFrame(cam, opts)
        pr, sh = gpu_ctx.rayStats()
        st = {}
        ref = orc.render_frame(scene.desc, cam, opts, 1, st)
        md, nbad, _ = maxdiff(a, ref)
        assert np.array_equal(np.isnan(a), np.isnan(ref)), name
        assert md <= TOL and nbad == 0, (name, md)
        assert (pr, sh) == (st["primary"], st["shadow"]), name
        if name == "empty":
            assert not a.any() and sh == 0
        if name == "darklight":
            assert sh == 0


def test_host_output_chunks_pinned_buffer_and_midframe_cancel(gpu_ctx):
    """c2rt_render_frame renders in row chunks that stream back while later
    chunks render; a pinned caller buffer and a pageable one get the same bits."""
    scene, cam, opts = load_config("lecture5_640x480_t5")
    gpu_ctx.uploadScene(scene.desc)
    a = gpu_ctx.renderFrame(cam, opts)                       # pageable numpy buffer
    ref = orc.render_frame(scene.desc, cam, opts, 0)
    assert maxdiff(a, ref)[0] <= TOL
    out = np.full((opts.height, opts.width, 3), -1.0, np.float32)
    gpu_ctx.pinHostBuffer(out)
    gpu_ctx.pinHostBuffer(out)                               # idempotent
    gpu_ctx.renderFrameInto(cam, opts, out)
    assert np.array_equal(out, a)
    gpu_ctx.unpinHostBuffer(out)
    with pytest.raises(c2.C2rtError):
        gpu_ctx.unpinHostBuffer(out)
    # ray statistics accumulate over the chunks
    _, _, o2 = load_config("lecture5_640x480_t5", count_rays=1)
    gpu_ctx.renderFrame(cam, o2)
    st = {}
    orc.render_frame(scene.desc, cam, opts, 0, st)
    assert gpu_ctx.rayStats() == (st["primary"], st["shadow"])


def test_chunks_of_a_pinned_4k_frame_share_one_mask_table(gpu_ctx, scenes_dir):
    """A 4K float frame into a pinned buffer is rendered in row chunks (several launches of the frame kernel,
    each streaming back while the next renders) that read ONE per-tile mask table written by one pre-pass launch;
    a pageable buffer gets the frame from a single launch.  Same bits — whole frame and the strips of rank 1 of 2
    (chunks of interleaved strips: the table's rows are local rows)."""
    scene = c2.parseSceneFromFile(os.path.join(scenes_dir, "lecture5.sdl"))
    scene.setFrameSize(3840, 2160)
    scene.setAA(False)
    cam = scene.beginFrame()
    gpu_ctx.uploadScene(scene.desc)
    for kw in ({}, dict(strip_height=8, strip_rank=1, strip_world=2)):
        opts = scene.renderOpts(**kw)
        rows = gpu_ctx.localRows(opts)
        one_launch = gpu_ctx.renderFrame(cam, opts)
        assert one_launch.shape[0] == rows
        out = np.full((rows, opts.width, 3), -1.0, np.float32)
        gpu_ctx.pinHostBuffer(out)
        try:
            gpu_ctx.renderFrameInto(cam, opts, out)
            assert np.array_equal(out.view(np.uint32), one_launch.view(np.uint32))
        finally:
            gpu_ctx.unpinHostBuffer(out)


def test_random_scene_fuzz(gpu_ctx, tmp_path, scenes_dir):
    """120 seeded random scenes (CSG trees up to depth 3-4 incl. planes as operands and Op(a, a), scaled /
    'rotated' / translated nodes, 1-3 lights, every shader and texture type): GPU == oracle within 1e-4,
    same NaN pattern, same ray counts."""
    import shutil
    from scene_fuzz import random_scene_sdl

    shutil.copy(os.path.join(scenes_dir, "floor.bmp"), tmp_path / "floor.bmp")
    worst = 0.0
    for seed in range(120):
        path = tmp_path / ("fuzz%d.sdl" % seed)
        path.write_text(random_scene_sdl(seed, max_depth=4 if seed % 3 == 0 else 3))
        scene = c2.parseSceneFromFile(str(path))
        cam = scene.beginFrame()
        for taps in ((1, 5) if seed % 8 == 0 else (1,)):
            opts = scene.renderOpts(taps=taps, count_rays=1)
            gpu_ctx.uploadScene(scene.desc)
            a = gpu_ctx.renderFrame(cam, opts)
            pr, sh = gpu_ctx.rayStats()
            st = {}
            ref = orc.render_frame(scene.desc, cam, opts, 2, st)
            md, nbad, _ = maxdiff(a, ref)
            worst = max(worst, md)
            assert np.array_equal(np.isnan(a), np.isnan(ref)), seed
            assert md <= TOL and nbad == 0, (seed, md)
            assert (pr, sh) == (st["primary"], st["shadow"]), seed
    print("fuzz: worst max|d| over 120 scenes = %.3g" % worst)
    # 4-6 lights all around the scene, incl. behind the camera: per-light shadow culling
    from scene_fuzz import many_lights_scene_sdl

    for seed in range(40):
        path = tmp_path / ("lights%d.sdl" % seed)
        path.write_text(many_lights_scene_sdl(seed))
        scene = c2.parseSceneFromFile(str(path))
        if seed % 2:
            scene.setFrameSize(133, 75)
        cam = scene.beginFrame()
        opts = scene.renderOpts(count_rays=1)
        gpu_ctx.uploadScene(scene.desc)
        a = gpu_ctx.renderFrame(cam, opts)
        pr, sh = gpu_ctx.rayStats()
        st = {}
        ref = orc.render_frame(scene.desc, cam, opts, 2, st)
        assert np.array_equal(np.isnan(a), np.isnan(ref)), seed
        assert maxdiff(a, ref)[0] <= TOL, seed
        assert (pr, sh) == (st["primary"], st["shadow"]), seed


def test_prepass_only_preview(gpu_ctx, tmp_path):
    scene, cam, _ = load_config("lecture5_333x217_t4")
    gpu_ctx.uploadScene(scene.desc)
    for bucket in (48, 40):
        _, _, po = load_config("lecture5_333x217_t4", prepass_bucket=bucket)
        a = gpu_ctx.renderFrame(cam, po)
        ref = orc.render_frame(scene.desc, cam, po, 0)
        assert maxdiff(a, ref)[0] <= TOL
    # with depth of field the lens jitter spans the (clipped) 16x16 block: renderPixelNoAA(x, y, ex - dx, ey - dy)
    zs, zcam, _ = load_config("zaphod_215x143_dof25")
    gpu_ctx.uploadScene(zs.desc)
    for bucket in (48, 40):
        _, _, po = load_config("zaphod_215x143_dof25", prepass_bucket=bucket, count_rays=1)
        a = gpu_ctx.renderFrame(zcam, po)
        pr = gpu_ctx.rayStats()
        st = {}
        ref = orc.render_frame(zs.desc, zcam, po, 0, st)
        assert maxdiff(a, ref)[0] <= TOL
        assert len(np.unique(a.reshape(-1, 3), axis=0)) > 20          # block colours, not one flat value
    gpu_ctx.uploadScene(scene.desc)
    # through the host mirror: GlobalSettings.prepassOnly
    text = open(os.path.join(SCENES, "lecture4.sdl")).read().replace("frameHeight\t\t\t480", "frameHeight 480\n        prepassOnly true")
    assert "prepassOnly" in text
    s = _load_text(tmp_path, text, "pre.sdl")
    assert s.settings.prepass_only == 1 and s.settings.prepass_enabled == 1
    img = c2.Renderer(s, gpu_ctx).renderRT()
    cam4 = s.beginFrame()
    ref = orc.render_frame(s.desc, cam4, s.renderOpts(prepass_bucket=48), 0)
    assert maxdiff(img, ref)[0] <= TOL
    assert np.array_equal(img[0:16, 16:32], np.broadcast_to(img[0, 16], (16, 16, 3)))   # blocky
    # prepassOnly without prepassEnabled: the reference returns at once, frame untouched
    s2 = _load_text(tmp_path, text.replace("prepassOnly true", "prepassOnly true\n        prepassEnabled false"), "pre2.sdl")
    out = np.full((480, 640, 3), 7.0, np.float32)
    flag = np.ones(1, np.uint8)
    r = c2.Renderer(s2, gpu_ctx)
    r.renderSceneAsync(out, flag)
    r.wait()
    assert flag[0] == 0 and (out == 7.0).all()


def test_csg_children_with_coincident_surfaces_match_the_oracle(gpu_ctx, tmp_path):
    """Depth-1 CsgOps whose children share surfaces, so that hit lists hold EQUAL distances (ties), odd counts
    (eye inside a child) and tangent contacts: the regular case of the device's CsgOp (0 or 2 hits per child, all
    distances distinct, decided by comparisons) must hand exactly these rays to the literal shell sort and walk —
    whose order of equal keys and leaf-identity toggles decide the pixel.  Bit-equal to the oracle."""
    prims = """Cube "a" { center 0 50 200; side 100 }
               Cube "b" { center 50 50 200; side 100 }        // shares the y and z face planes of "a"
               Cube "c" { center 0 50 200; side 60 }          // concentric, inside "a"
               Sphere "t" { center 0 50 200; R 50 }           // tangent to the faces of "a" from inside
               Sphere "o" { center 0 50 150; R 50 }           // pokes through the front face
               Cube "e" { center 0 165 0; side 40 }           // the eye sits at its centre
               Plane "f" { y 0 }"""
    ops = [("CsgUnion", "a", "b"), ("CsgInter", "a", "b"), ("CsgDiff", "a", "b"), ("CsgDiff", "b", "a"),
           ("CsgDiff", "a", "c"), ("CsgInter", "a", "t"), ("CsgDiff", "a", "t"), ("CsgUnion", "t", "a"),
           ("CsgDiff", "a", "o"), ("CsgInter", "o", "a"), ("CsgUnion", "e", "a"), ("CsgDiff", "a", "a"),
           ("CsgInter", "t", "t")]
    for i, (op, l, r) in enumerate(ops):
        text = """Scene { GlobalSettings { ambientLightColor 0.1 0.1 0.1 }
            Camera { pos 0 165 0; pitch -30; fov 90 }
            Lights { PointLight "k" { pos -90 700 350; color 1 1 1; power 800000 } }
            Geometries { %s
               %s "x" { left "%s"; right "%s" } }
            Shaders { Phong "p" { color 0.5 0.5 0; exponent 60 }; Lambert "l" { color 0.7 0.7 0.7 } }
            Nodes { Node "floor" { geometry "f"; shader "l" }; Node "n" { geometry "x"; shader "p" } } }""" % (prims, op, l, r)
        scene = _load_text(tmp_path, text, "coincident%d.sdl" % i)
        scene.setFrameSize(320, 200)
        for aa in (False, True):
            scene.setAA(aa)
            cam = scene.beginFrame()
            opts = scene.renderOpts(count_rays=1)
            gpu_ctx.uploadScene(scene.desc)
            a = gpu_ctx.renderFrame(cam, opts)
            st = {}
            ref = orc.render_frame(scene.desc, cam, opts, 2, st)
            assert np.array_equal(np.isnan(a), np.isnan(ref)), (op, l, r)
            assert np.array_equal(a.view(np.uint32), ref.view(np.uint32)), (op, l, r, maxdiff(a, ref))
            assert gpu_ctx.rayStats() == (st["primary"], st["shadow"]), (op, l, r)


def test_random_scene_sweep(gpu_ctx, tmp_path, scenes_dir):
    """3 000 more seeded random scenes at 64x48 (the generators of the offline sweeps, scripts/fuzz_big.py: general
    scenes with CSG trees to depth 4 and many-lights scenes; ~15 s): every float of every frame equal to the
    oracle's, same NaN pattern, same ray counts — through the production and the counting instances (conftest)."""
    import shutil
    from scene_fuzz import many_lights_scene_sdl, random_scene_sdl

    shutil.copy(os.path.join(scenes_dir, "floor.bmp"), tmp_path / "floor.bmp")
    path = tmp_path / "sweep.sdl"
    differing = 0
    for seed in range(200000, 203000):
        path.write_text(many_lights_scene_sdl(seed) if seed % 5 == 0 else random_scene_sdl(seed, max_depth=4 if seed % 2 else 3))
        scene = c2.parseSceneFromFile(str(path))
        scene.setFrameSize(64, 48)
        cam = scene.beginFrame()
        opts = scene.renderOpts(count_rays=1)
        gpu_ctx.uploadScene(scene.desc)
        a = gpu_ctx.renderFrame(cam, opts)
        st = {}
        ref = orc.render_frame(scene.desc, cam, opts, 8, st)
        both_nan = np.isnan(a) & np.isnan(ref)
        assert np.array_equal(np.isnan(a), np.isnan(ref)), seed
        differing += int(((a.view(np.uint32) != ref.view(np.uint32)) & ~both_nan).sum())
        assert maxdiff(a, ref)[0] <= TOL, seed
        assert gpu_ctx.rayStats() == (st["primary"], st["shadow"]), seed
    assert differing == 0


def test_occluder_outside_the_left_childs_box(gpu_ctx, tmp_path, scenes_dir):
    """Scene 108921 of the offline sweep (scripts/fuzz_big.py): Diff / Inter nodes whose LEFT child is a CsgOp.  The
    reference's walk compares leaves with `left` (rt/geometry.d:314-317), so no entry ever toggles inL there: inL is
    the parity of the left list, and the operator can come out "in" at an entry of the RIGHT child anywhere along the
    ray — outside the left child's box.  Three floor pixels lost a shadow to the view-pyramid culling of shadow rays
    while the culling boxes of such nodes were their left child's box (c2rt_api.cpp: box_of)."""
    import shutil
    from scene_fuzz import random_scene_sdl

    shutil.copy(os.path.join(scenes_dir, "floor.bmp"), tmp_path / "floor.bmp")
    path = tmp_path / "fuzz108921.sdl"
    path.write_text(random_scene_sdl(108921, max_depth=4))
    scene = c2.parseSceneFromFile(str(path))
    for w, h in ((64, 48), (256, 192)):
        scene.setFrameSize(w, h)
        cam = scene.beginFrame()
        opts = scene.renderOpts(count_rays=1)
        gpu_ctx.uploadScene(scene.desc)
        a = gpu_ctx.renderFrame(cam, opts)
        st = {}
        ref = orc.render_frame(scene.desc, cam, opts, 2, st)
        assert np.array_equal(np.isnan(a), np.isnan(ref))
        assert np.array_equal(a.view(np.uint32), ref.view(np.uint32)), maxdiff(a, ref)
        assert gpu_ctx.rayStats() == (st["primary"], st["shadow"])


def test_nested_csg_trees_that_only_shadow_the_view(gpu_ctx, tmp_path):
    """The shadow masks of lights 1.. come from the per-tile table the pre-pass writes (c2rt_trace.inc:
    light_shadow_mask; until round 4 every sample derived them with a ballot).  Here the deep trees (depth 3 and 4)
    are OUT of view and only their shadows — from two lights on opposite sides, and from five lights, one more than
    the table holds masks for — fall into the frame, next to a depth-1 and a depth-2 tree in view: a mask that culled
    one of them for a tile its shadow crosses would lose that shadow.  Frames and ray counts equal the oracle's, at a
    frame with clipped tiles too."""
    geoms = """Plane "floor" { y 0 }
        Cube "c1" { center 0 40 0; side 60 }
        Sphere "s1" { center 0 40 0; R 38 }
        CsgDiff "d1" { left "c1"; right "s1" }
        Sphere "s2" { center 0 40 0; R 20 }
        CsgUnion "u2" { left "s2"; right "d1" }
        Cube "c3" { center 0 40 0; side 80 }
        CsgInter "i3" { left "c3"; right "u2" }
        Sphere "s4" { center 10 50 5; R 30 }
        CsgDiff "d4" { left "i3"; right "s4" }
        Cube "k1" { center 0 15 0; side 30 }
        Sphere "k2" { center 0 15 0; R 19 }
        CsgDiff "e1" { left "k1"; right "k2" }
        Sphere "k3" { center 0 22 0; R 9 }
        CsgUnion "e2" { left "e1"; right "k3" }"""
    lights3 = """PointLight "a" { pos -600 500 -200; color 1 0.9 0.8; power 900000 }
        PointLight "b" { pos 700 300 100; color 0.5 0.6 1; power 500000 }"""
    lights5 = lights3 + """
        PointLight "c" { pos 0 900 -900; color 0.3 0.3 0.3; power 900000 }
        PointLight "d" { pos 100 400 900; color 0.2 0.4 0.2; power 900000 }
        PointLight "e" { pos -900 200 900; color 0.4 0.2 0.2; power 900000 }"""
    for name, lights, size in (("two_lights", lights3, (320, 240)), ("two_lights_clipped", lights3, (333, 217)), ("five_lights", lights5, (200, 152))):
        text = """Scene { GlobalSettings { ambientLightColor 0.05 0.05 0.05; AAEnabled false }
            Camera { pos 0 120 -260; pitch -22; fov 60 }
            Lights { %s }
            Geometries { %s }
            Shaders {
              Lambert "w" { color 0.9 0.9 0.9 }
              Phong "g" { color 0.8 0.6 0.1; exponent 20 }
            }
            Nodes {
              Node "floor" { geometry "floor"; shader "w" }
              Node "deep_left"  { geometry "d4"; shader "g"; translate -330 0 60 }     // depth 4, left of the view
              Node "deep_right" { geometry "i3"; shader "g"; translate 340 0 40 }      // depth 3, right of the view
              Node "mid" { geometry "e2"; shader "g"; translate 60 0 -20 }             // depth 2, in view
              Node "small" { geometry "e1"; shader "w"; translate -70 0 -40 }          // depth 1, in view
            } }""" % (lights, geoms)
        nne, _ = _check_scene_text(gpu_ctx, tmp_path / (name + ".sdl"), text, size=size)
        assert nne == 0, name


def test_pathological_scenes_terminate_and_match(gpu_ctx, tmp_path):
    """Inputs on which the reference itself never terminates (findAllIntersections loops for ever when
    `p + dir*1e-6 == p` or the hit is NaN) or degenerates (NaN camera, zero-size primitives): the device
    path and the oracle both cap the hit lists at C2RT_MAX_CSG_HITS, must terminate and must agree."""
    cases = {
        # stepping by 1e-6 is absorbed at 1e12: the reference would never leave the first hit
        "absorbed_eps": ('''Scene { Camera { pos 0 0 -3e12; fov 40 }
            Lights { PointLight "k" { pos 0 4e12 -4e12; color 1 1 1; power 1e27 } }
            Geometries { Cube "c" { center 0 0 0; side 1e12 }; Sphere "s" { center 0 0 0; R 6e11 }; CsgDiff "d" { left "c"; right "s" } }
            Shaders { Lambert "l" { } }
            Nodes { Node "n" { geometry "d"; shader "l" } } }''', True),
        "nan_camera": ('''Scene { Camera { fov 60 }
            Lights { PointLight "k" { pos 0 10 0; color 1 1 1; power 100 } }
            Geometries { Sphere "s" { center 0 0 5; R 1 }; Cube "c" { center 0 0 5; side 1 }; CsgUnion "u" { left "s"; right "c" } }
            Shaders { Phong "l" { } }
            Nodes { Node "n" { geometry "u"; shader "l" } } }''', True),
        "degenerate_prims": ('''Scene { Camera { pos 0 1 -5; fov 60 }
            Lights { PointLight "k" { pos 0 10 0; color 1 1 1; power 100 } }
            Geometries { Sphere "s" { center 0 1 0; R 0 }; Cube "c" { center 0 1 0; side 0 }; CsgInter "i" { left "s"; right "c" }; Plane "p" { y 0 } }
            Shaders { Lambert "l" { } }
            Nodes { Node "a" { geometry "i"; shader "l" }; Node "b" { geometry "s"; shader "l"; scale 0 0 0 }; Node "f" { geometry "p"; shader "l" } } }''', True),
    }
    truncated = {}
    for name, (text, compare) in cases.items():
        scene = _load_text(tmp_path, text, name + ".sdl")
        scene.setFrameSize(64, 48)
        scene.setAA(False)
        cam = scene.beginFrame()
        opts = scene.renderOpts(count_rays=1)
        gpu_ctx.uploadScene(scene.desc)
        a = gpu_ctx.renderFrame(cam, opts)          # must return (bounded loops)
        truncated[name] = gpu_ctx.csgTruncations()
        assert a.shape == (48, 64, 3)
        if compare:
            ref = orc.render_frame(scene.desc, cam, opts, 1)
            assert np.array_equal(np.isnan(a), np.isnan(ref)), name
            assert maxdiff(a, ref)[0] <= TOL, name
    # the cap on findAllIntersections is reported where it bites (the absorbed 1e-6 step: the hit list of
    # every ray that meets the object fills up) and nowhere else
    assert truncated["absorbed_eps"] > 0, truncated
    for name in sorted(CONFIGS):
        scene, cam, opts = load_config(name, count_rays=1)
        gpu_ctx.uploadScene(scene.desc)
        gpu_ctx.renderFrame(cam, opts)
        assert gpu_ctx.csgTruncations() == 0, name


def test_rgb32_frame_and_strips(gpu_ctx):
    """Display-format output: render + Color.toRGB32 on the device, 4 B/pixel back to the host; packed strips
    de-interleave like float ones."""
    import torch

    scene, cam, opts = load_config("lecture5_333x217_t4")
    gpu_ctx.uploadScene(scene.desc)
    W, H = opts.width, opts.height
    flt = gpu_ctx.renderFrame(cam, opts)
    packed = gpu_ctx.renderFrameRGB32(cam, opts)
    assert packed.shape == (H, W) and packed.dtype == np.uint32
    L = orc.lib()
    rng = np.random.RandomState(2)
    for _ in range(2000):
        y, x = int(rng.randint(0, H)), int(rng.randint(0, W))
        c = (C.c_float * 3)(*flt[y, x])
        assert int(packed[y, x]) == L.orc_color_to_rgb32(c)
    world = 3
    plan = c2.plan_strips(H, world, 8)
    gathered = torch.zeros((world, plan.rows_pad, W), dtype=torch.int32, device="cuda")
    for r in range(world):
        _, _, o = load_config("lecture5_333x217_t4", strip_height=8, strip_rank=r, strip_world=world)
        part = gpu_ctx.renderFrameRGB32(cam, o)
        assert np.array_equal(part, packed[[y for y in range(H) if (y // 8) % world == r]])
        gathered[r, : part.shape[0]] = torch.from_numpy(part.astype(np.int32)).cuda()
    frame = torch.empty((H, W), dtype=torch.int32, device="cuda")
    gpu_ctx.deinterleaveStripsRGB32(gathered.data_ptr(), frame.data_ptr(), W, H, 8, world, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    assert np.array_equal(frame.cpu().numpy().astype(np.uint32), packed)
    # pinned destination: the frame comes back in row chunks (render + encode of chunk i+1 overlap the copy of chunk i)
    scene.setFrameSize(1283, 1001)
    cam2 = scene.beginFrame()
    o2 = scene.renderOpts(taps=1, count_rays=1)
    whole = gpu_ctx.renderFrameRGB32(cam2, o2)
    pr = gpu_ctx.rayStats()
    pinned = np.zeros((1001, 1283), np.uint32)
    gpu_ctx.pinHostBuffer(pinned)
    try:
        gpu_ctx.renderFrameRGB32Into(cam2, o2, pinned)
        assert gpu_ctx.rayStats() == pr
    finally:
        gpu_ctx.unpinHostBuffer(pinned)
    assert np.array_equal(pinned, whole)
    # the PRODUCTION (uncounted, lean::) instance storing display words straight into the page-locked frame over
    # PCIe (render_to_host's direct-store branch: no staging buffer), whole frame and strips
    o3 = scene.renderOpts(taps=1)
    pinned = np.zeros((1001, 1283), np.uint32)
    gpu_ctx.pinHostBuffer(pinned)
    try:
        gpu_ctx.renderFrameRGB32Into(cam2, o3, pinned)
        assert np.array_equal(pinned, whole)
        o4 = scene.renderOpts(taps=1, strip_height=8, strip_rank=2, strip_world=5)
        rows = [y for y in range(1001) if (y // 8) % 5 == 2]
        pinned[:] = 0
        gpu_ctx.renderFrameRGB32Into(cam2, o4, pinned[: len(rows)])
        assert np.array_equal(pinned[: len(rows)], whole[rows]) and not pinned[len(rows):].any()
    finally:
        gpu_ctx.unpinHostBuffer(pinned)


def test_float_frame_stored_straight_into_the_pinned_buffer():
    """C2RT_HOST_DIRECT_STORE=2 (diagnostics build): the float frame too is stored by the kernel straight into the
    page-locked destination (12-byte stores over PCIe; slower than the chunked copy, which is why it is not the
    default) — production and counting instances, frames equal the staged ones bit for bit."""
    import subprocess
    import sys

    code = r'''
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import chess2rt_amd as c2
from golden_configs import load_config
assert c2._abi.LIB_PATH.endswith("libc2rt_diag.so")
ctx = c2.Context(0)
scene, cam, opts = load_config("lecture5_333x217_t4")
ctx.uploadScene(scene.desc)
staged = ctx.renderFrame(cam, opts)                      # pageable destination: staging buffer + one copy
for count in (0, 1):
    _, _, o = load_config("lecture5_333x217_t4", count_rays=count)
    pinned = np.full(staged.shape, -1.0, np.float32)
    ctx.pinHostBuffer(pinned)
    try:
        ctx.renderFrameInto(cam, o, pinned)
    finally:
        ctx.unpinHostBuffer(pinned)
    assert np.array_equal(pinned.view(np.uint32), staged.view(np.uint32)), count
print("ok")
'''
    env = dict(os.environ, C2RT_HOST_DIRECT_STORE="2", C2RT_LIB_VARIANT="diag")
    p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300, env=env,
                       cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    assert p.returncode == 0 and "ok" in p.stdout, p.stdout + p.stderr


def test_plain_c_caller_matches_oracle(tmp_path):
    """examples/c_abi_demo.c's hand-built tables (CsgDiff, Phong, checker) through
    the C ABI from a C program, against the oracle linked into the same program."""
    import subprocess

    from test_abi_exports import ROOT, _build_c_program

    exe = str(tmp_path / "c_abi_check")
    _build_c_program(os.path.join(ROOT, "tests", "c_abi_check.c"), exe, with_oracle=True)
    p = subprocess.run([exe, "200", "152"], capture_output=True, text=True, timeout=120)
    assert p.returncode == 0, p.stdout + p.stderr
    assert "beyond_tol 0" in p.stdout


@pytest.mark.parametrize("slots,w,h", [(2, 203, 149), (3, 333, 217), (8, 640, 360)])
def test_multi_device_context_from_plain_c(tmp_path, slots, w, h):
    """c2rt_init_multi (SURVEY.md 8(b) `device_count_or_0`) driven from a C program: N device slots (all on
    HIP device 0 here) deal the frame in interleaved 8-row strips and copy them straight into the host
    frame; float (pageable, pinned) and RGB32 frames equal the one-device frames bit for bit, the ray
    counts add up, the one-device frame matches the oracle."""
    import subprocess

    from test_abi_exports import ROOT, _build_c_program

    exe = str(tmp_path / "c_multi_check")
    _build_c_program(os.path.join(ROOT, "tests", "c_multi_check.c"), exe, with_oracle=True)
    p = subprocess.run([exe, str(w), str(h), str(slots)], capture_output=True, text=True, timeout=180)
    assert p.returncode == 0, p.stdout + p.stderr
    assert "differ 0 differ32 0 beyond_tol 0" in p.stdout


def test_multi_device_context_device_output_and_modes():
    """The same through the Python face, incl. the device-output entry point (every slot's kernel stores
    its strips straight into the lead device's frame — RenderParams::frame_rows), 1/4/5 taps, depth of
    field, prepassOnly and a frame whose last strip is partial; against one-device frames and the oracle."""
    import torch

    one = c2.Context(0)
    multi = c2.Context(devices=[0, 0, 0])
    assert multi.deviceCount == 3 and one.deviceCount == 1
    try:
        for name, kw in [("lecture5_333x217_t4", {}), ("lecture5_640x480_t5", {}), ("csg_stress_320x240_t1", {}),
                         ("zaphod_215x143_dof25", {}), ("lecture4_640x480_t1", {"prepass_bucket": 48})]:
            scene, cam, opts = load_config(name, count_rays=1, **kw)
            one.uploadScene(scene.desc)
            multi.uploadScene(scene.desc)
            a = one.renderFrame(cam, opts)
            ra = one.rayStats()
            b = multi.renderFrame(cam, opts)
            assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), name
            assert multi.rayStats() == ra, name
            dev = torch.full((opts.height, opts.width, 3), -1.0, dtype=torch.float32, device="cuda:0")
            multi.renderFrameDevice(cam, opts, dev.data_ptr(), torch.cuda.current_stream().cuda_stream)
            torch.cuda.synchronize()
            assert np.array_equal(dev.cpu().numpy().view(np.uint32), a.view(np.uint32)), name
            assert multi.rayStats() == ra, name
            assert np.array_equal(multi.renderFrameRGB32(cam, opts), one.renderFrameRGB32(cam, opts)), name
            ref = orc.render_frame(scene.desc, cam, opts, 0)
            md, nbad, nne = maxdiff(b, ref)
            assert md <= TOL, (name, md)
    finally:
        multi.close()
        one.close()


def test_two_scenes_alternating_on_one_context(gpu_ctx):
    """A context holds ONE scene.  Two Renderers (host scene objects) sharing a Context must each get
    their own scene rendered whenever it is their turn — also after a direct uploadScene of something
    else — never the other scene's tables under this scene's camera."""
    sa = c2.parseSceneFromFile(os.path.join(SCENES, "lecture4.sdl"))
    sb = c2.parseSceneFromFile(os.path.join(SCENES, "lecture5.sdl"))
    for s in (sa, sb):
        s.setFrameSize(160, 120)
    ra, rb = c2.Renderer(sa, gpu_ctx), c2.Renderer(sb, gpu_ctx)
    refs = {}
    for key, s in (("a", sa), ("b", sb)):
        refs[key] = orc.render_frame(s.desc, s.beginFrame(), s.renderOpts(), 0)
    gens = set()
    for turn in ("a", "b", "a", "a", "b", "upload", "a", "b"):
        if turn == "upload":
            other = c2.parseSceneFromFile(os.path.join(SCENES, "zaphod.sdl"))
            gpu_ctx.uploadScene(other.desc)
            continue
        img = (ra if turn == "a" else rb).renderRT()
        gens.add(gpu_ctx.sceneGeneration)
        md, nbad, nne = maxdiff(img, refs[turn])
        assert md <= TOL, (turn, md)
        s = sa if turn == "a" else sb
        probe = (ra if turn == "a" else rb).renderPixelNoAA(80, 100)   # one sample, no AA (rt/renderer.d:46-57)
        want = orc.render_pixel(s.desc, s.beginFrame(), s.renderOpts(), 80, 100)
        assert probe.closest_node == want.closest_node and np.allclose(np.array(probe.color), np.array(want.color), atol=TOL), turn
    assert len(gens) >= 5      # every switch of scene was a fresh upload with a new generation


@pytest.mark.parametrize("cap", [1, 3, 6, 11])
def test_csg_hit_stack_overflow_is_redone_at_full_capacity(cap):
    """Nested-CSG scenes run with a reduced LDS hit stack first; tiles in which a lane's nested lists
    outgrow it are rendered again by the full-capacity launch.  C2RT_CSG_FIRST_CAP (a test hook of the
    DIAGNOSTICS build, chess2rt_amd/libc2rt_diag.so: the same kernel objects under a c2rt_api.cpp compiled with
    the environment hooks) forces tiny stacks, so that most CSG tiles of ordinary scenes take that path: the
    frames, the ray counts and a strip-sharded render must not change."""
    import subprocess
    import sys

    code = r'''
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import chess2rt_amd as c2, oracle_lib as orc
from golden_configs import load_config
ctx = c2.Context(0)
for name in ("csg_stress_320x240_t5", "csg_corner_256x192_t1"):
    scene, cam, opts = load_config(name, count_rays=1)
    ctx.uploadScene(scene.desc)
    a = ctx.renderFrame(cam, opts)
    rays = ctx.rayStats()
    st = {}
    ref = orc.render_frame(scene.desc, cam, opts, 0, st)
    assert np.array_equal(a.view(np.uint32), ref.view(np.uint32)), name
    assert rays == (st["primary"], st["shadow"]), name
    _, _, o2 = load_config(name, strip_height=8, strip_rank=1, strip_world=3)
    b = ctx.renderFrame(cam, o2)
    rows = [y for y in range(opts.height) if (y // 8) % 3 == 1]
    assert np.array_equal(b.view(np.uint32), ref[rows].view(np.uint32)), name
print("ok")
'''
    env = dict(os.environ, C2RT_CSG_FIRST_CAP=str(cap), C2RT_LIB_VARIANT="diag")
    p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300, env=env,
                       cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    assert p.returncode == 0 and "ok" in p.stdout, p.stdout + p.stderr


def _check_scene_text(gpu_ctx, path, text, size=None, taps=None):
    path.write_text(text)
    scene = c2.parseSceneFromFile(str(path))
    if size:
        scene.setFrameSize(*size)
    cam = scene.beginFrame()
    opts = scene.renderOpts(count_rays=1) if taps is None else scene.renderOpts(count_rays=1, taps=taps)
    gpu_ctx.uploadScene(scene.desc)
    a = gpu_ctx.renderFrame(cam, opts)
    rays = gpu_ctx.rayStats()
    st = {}
    ref = orc.render_frame(scene.desc, cam, opts, 0, st)
    assert np.array_equal(np.isnan(a), np.isnan(ref)), path
    md, nbad, nne = maxdiff(a, ref)
    assert md <= TOL and nbad == 0, (str(path), md)
    assert rays == (st["primary"], st["shadow"]), path
    return nne, scene.desc.contents.n_nodes


@pytest.mark.parametrize("n_objects", [30, 31, 32, 33, 47, 63, 64, 90])
def test_many_node_scenes_cross_the_cull_mask_boundary(gpu_ctx, tmp_path, scenes_dir, n_objects):
    """Ground plane + 30..90 objects: 31..91 nodes.  The per-tile culling masks cover nodes 0..31
    (kMaxCullNodes); nodes from 32 on are always tested (next_node's n >= 32 path), and the ground
    refinement needs n_nodes <= 32.  Frames and ray counts must equal the oracle's on both sides of
    that boundary, with one and two lights, at 320x180 and 5 taps for one seed."""
    import shutil

    from scene_fuzz import many_nodes_scene_sdl

    shutil.copy(os.path.join(scenes_dir, "floor.bmp"), tmp_path / "floor.bmp")
    differing = 0
    for seed in range(3):
        nne, nn = _check_scene_text(gpu_ctx, tmp_path / ("many%d_%d.sdl" % (n_objects, seed)), many_nodes_scene_sdl(seed, n_objects),
                                    taps=5 if seed == 2 else 1)
        assert nn == n_objects + 1
        differing += nne
    print("many-node fuzz (%d objects): %d differing floats" % (n_objects, differing))


def test_fuzz_scenes_at_larger_frames(gpu_ctx, tmp_path, scenes_dir):
    """20 fuzz scenes (general CSG trees, ground-plane scenes, many-lights, planes-only, many-node) at
    640x360 .. 800x450: thousands of tiles per frame instead of the 108 of the small fuzz frames, so
    that the per-tile culling masks, the dispatch rotation and the XCD tile map are exercised the
    way a full frame exercises them."""
    import shutil

    from scene_fuzz import ground_scene_sdl, many_lights_scene_sdl, many_nodes_scene_sdl, planes_scene_sdl, random_scene_sdl

    shutil.copy(os.path.join(scenes_dir, "floor.bmp"), tmp_path / "floor.bmp")
    cases = [("gen%d" % s, random_scene_sdl(1000 + s, max_depth=3 + s % 2), (640, 360)) for s in range(6)]
    cases += [("ground%d" % s, ground_scene_sdl(500 + s), (800, 450)) for s in range(6)]
    cases += [("lights%d" % s, many_lights_scene_sdl(700 + s), (640, 360)) for s in range(3)]
    cases += [("planes%d" % s, planes_scene_sdl(900 + s), (723, 407)) for s in range(3)]
    cases += [("many%d" % s, many_nodes_scene_sdl(10 + s, 40 + 20 * s), (640, 360)) for s in range(2)]
    assert len(cases) == 20
    differing = 0
    for name, text, size in cases:
        nne, _ = _check_scene_text(gpu_ctx, tmp_path / (name + ".sdl"), text, size=size)
        differing += nne
    print("large-frame fuzz: %d differing floats over 20 scenes" % differing)


@pytest.mark.parametrize("scene_file,w,h,taps,rays", [("zaphod.sdl", 3840, 2160, 4, None), ("lecture5.sdl", 7680, 4320, 4, 132710400)])
def test_baseline_configs_4_and_5_full_frames(gpu_ctx, scene_file, w, h, taps, rays):
    """BASELINE configs[3] (zaphod.sdl 3840x2160, "4 spp", DOF off) and configs[4]'s frame (lecture5.sdl
    7680x4320, "4 spp") at FULL size, float for float against the oracle on all host cores."""
    s = c2.parseSceneFromFile(os.path.join(SCENES, scene_file))
    s.setFrameSize(w, h)
    s.setDof(False)
    cam = s.beginFrame()
    opts = s.renderOpts(taps=taps, count_rays=1)
    gpu_ctx.uploadScene(s.desc)
    gpu = gpu_ctx.renderFrame(cam, opts)
    got = gpu_ctx.rayStats()
    st = {}
    ref = orc.render_frame(s.desc, cam, opts, 0, st)
    md, nbad, nne = maxdiff(gpu, ref)
    print("%s %dx%d x%d: max|d|=%.3g, !=: %d of %d floats" % (scene_file, w, h, taps, md, nne, gpu.size))
    assert md <= TOL and nbad == 0
    assert got == (st["primary"], st["shadow"])
    if rays:
        assert got[0] == rays


def test_zaphod_as_shipped_full_size_dof25(gpu_ctx):
    """zaphod.sdl exactly as the file ships — depth of field on, 25 lens samples per pixel (rt/camera.d:154-173,
    rt/renderer.d:270-287; the lens draws from the build's counter RNG on both sides) — at
    3840x2160, the frame bench.py times as `zaphod_4k_dof25`: every float against the oracle on all host cores."""
    s = c2.parseSceneFromFile(os.path.join(SCENES, "zaphod.sdl"))
    s.setFrameSize(3840, 2160)
    cam = s.beginFrame()
    assert cam.dof == 1 and cam.num_samples == 25
    opts = s.renderOpts(taps=c2.TAPS_1, count_rays=1, seed=7)
    gpu_ctx.uploadScene(s.desc)
    gpu = gpu_ctx.renderFrame(cam, opts)
    got = gpu_ctx.rayStats()
    st = {}
    ref = orc.render_frame(s.desc, cam, opts, 0, st)
    md, nbad, nne = maxdiff(gpu, ref)
    print("zaphod.sdl 3840x2160 DOF x25: max|d|=%.3g, !=: %d of %d floats" % (md, nne, gpu.size))
    assert md <= TOL and nbad == 0
    assert got == (st["primary"], st["shadow"]) and got[0] == 3840 * 2160 * 25


def test_zaphod_anchor_pixels_on_the_gpu(gpu_ctx, golden_dir):
    """The independent yaw / roll anchors (tests/golden/zaphod_anchors.json, derived by
    tests/golden/make_zaphod_anchors.py without the oracle or the host mirror) against the HIP path itself:
    the probe's ray, hit and colour, and the frame kernel's pixel."""
    a = json.load(open(os.path.join(golden_dir, "zaphod_anchors.json")))
    s = c2.parseSceneFromFile(os.path.join(SCENES, a["scene"]))
    s.setFrameSize(a["width"], a["height"])
    s.setAA(False)
    s.setDof(False)
    cam = s.beginFrame()
    opts = s.renderOpts()
    gpu_ctx.uploadScene(s.desc)
    frame = gpu_ctx.renderFrame(cam, opts)
    for px in a["pixels"]:
        g = gpu_ctx.renderPixel(cam, opts, px["x"], px["y"])
        assert g.closest_node == 0
        np.testing.assert_allclose(list(g.ray_dir), px["dir"], atol=a["dir_tolerance"], rtol=0)
        np.testing.assert_allclose(list(g.p), px["p"], atol=a["geom_tolerance"], rtol=0)
        np.testing.assert_allclose([g.dist, g.u, g.v], [px["t"], px["u"], px["v"]], atol=a["geom_tolerance"], rtol=0)
        np.testing.assert_allclose(list(g.color), px["rgb"], atol=a["rgb_tolerance"], rtol=0)
        np.testing.assert_allclose(frame[px["y"], px["x"]], px["rgb"], atol=a["rgb_tolerance"], rtol=0)


LIBM_RESIDUAL = dict(seed=12023564, sizes=((64, 48), (128, 96), (200, 152), (320, 240)))


def test_device_libm_residual_is_inside_the_tolerance(gpu_ctx, tmp_path, scenes_dir):
    """The one KNOWN float (of 2.7e9 swept offline) that is not the oracle's: fuzz seed
    12023564, a Phong lobe with exponent 2.04391 where the device's libm `pow` and glibc's round the double
    differently — 1 ulp of one fp32 channel of one pixel.  This is why the claim is "<= 1e-4 with bit-identical
    frames on everything else", not "bit-identical by construction"; the suite holds the case so the residual is
    visible where the driver looks: the difference is there (> 0), far inside the tolerance (<= 1e-4, in fact
    <= 1e-6), in a handful of floats at most, and the probe kernel returns what the frame kernel returns (it is
    the device's arithmetic, not a frame-kernel shortcut)."""
    import shutil

    from scene_fuzz import random_scene_sdl

    shutil.copy(os.path.join(scenes_dir, "floor.bmp"), tmp_path / "floor.bmp")
    seed = LIBM_RESIDUAL["seed"]
    path = tmp_path / "residual.sdl"
    path.write_text(random_scene_sdl(seed, max_depth=4 if seed % 2 else 3))     # scripts/fuzz_big.py's generator call
    s = c2.parseSceneFromFile(str(path))
    seen = []
    for (w, h) in LIBM_RESIDUAL["sizes"]:
        s.setFrameSize(w, h)
        cam = s.beginFrame()
        opts = s.renderOpts(count_rays=1)
        gpu_ctx.uploadScene(s.desc)
        a = gpu_ctx.renderFrame(cam, opts)
        got = gpu_ctx.rayStats()
        st = {}
        ref = orc.render_frame(s.desc, cam, opts, 0, st)
        md, nbad, nne = maxdiff(a, ref)
        assert md <= TOL and nbad == 0 and got == (st["primary"], st["shadow"]), (w, h, md)
        assert md <= 1e-6 and nne <= 4, (w, h, md, nne)
        for (y, x, ch) in np.argwhere(a.view(np.uint32) != ref.view(np.uint32)):
            g = gpu_ctx.renderPixel(cam, s.renderOpts(), int(x), int(y))
            o = orc.render_pixel(s.desc, cam, s.renderOpts(), int(x), int(y))
            assert np.float32(g.color[ch]) == a[y, x, ch] != ref[y, x, ch] == np.float32(o.color[ch])
            assert abs(int(a.view(np.uint32)[y, x, ch]) - int(ref.view(np.uint32)[y, x, ch])) == 1     # one ulp
            assert (g.closest_node, g.leaf_geom, g.dist, list(g.p)) == (o.closest_node, o.leaf_geom, o.dist, list(o.p))
            seen.append((w, h, int(x), int(y), int(ch), float(md)))
    print("libm residual:", seen)
    assert seen, "the device-libm residual of seed %d did not show at any swept frame size" % seed


def test_planes_only_scenes(gpu_ctx, tmp_path, scenes_dir):
    """Scenes made of Plane nodes only run the kernel instances that decide a plane's miss from the
    un-normalised ray (plane_points_away): scaled / mirrored / translated / bounded planes, lights
    above, below, level with the plane, 1e155 away (outside the shortcut's bounds) or 1e-155 close,
    cameras below the planes or looking at the sky; with and without depth of field."""
    import shutil

    from scene_fuzz import planes_scene_sdl

    shutil.copy(os.path.join(scenes_dir, "floor.bmp"), tmp_path / "floor.bmp")
    differing = 0
    for seed in range(80):
        path = tmp_path / ("planes%d.sdl" % seed)
        path.write_text(planes_scene_sdl(seed))
        scene = c2.parseSceneFromFile(str(path))
        if seed % 3 == 0:
            scene.setFrameSize(131, 77)
        if seed % 10 == 9:
            scene.setDof(True)
        cam = scene.beginFrame()
        opts = scene.renderOpts(count_rays=1, seed=seed)
        gpu_ctx.uploadScene(scene.desc)
        a = gpu_ctx.renderFrame(cam, opts)
        pr, sh = gpu_ctx.rayStats()
        st = {}
        ref = orc.render_frame(scene.desc, cam, opts, 2, st)
        assert np.array_equal(np.isnan(a), np.isnan(ref)), seed
        assert np.array_equal(np.isinf(a), np.isinf(ref)), seed
        fin = np.isfinite(ref)
        md, nbad, nne = maxdiff(np.where(fin, a, 0), np.where(fin, ref, 0))
        # depth-of-field seeds too: the lens sample is libm-free and bit-identical on both sides
        assert md <= TOL, (seed, md)
        assert (pr, sh) == (st["primary"], st["shadow"]), seed
        differing += nne
    print("planes-only fuzz: %d differing floats over 80 scenes" % differing)


def test_ground_plane_shadow_rectangles(gpu_ctx, tmp_path, scenes_dir):
    """Ground plane + small objects above / below / through it, light 0 above, grazing, below the
    ground or among the objects, camera above or below: tiles that only see the ground drop the
    nodes whose light-projected box misses the tile's footprint; frames must stay identical."""
    import shutil

    from scene_fuzz import ground_scene_sdl

    shutil.copy(os.path.join(scenes_dir, "floor.bmp"), tmp_path / "floor.bmp")
    differing = 0
    for seed in range(80):
        path = tmp_path / ("ground%d.sdl" % seed)
        path.write_text(ground_scene_sdl(seed))
        scene = c2.parseSceneFromFile(str(path))
        if seed % 3 == 0:
            scene.setFrameSize(163, 101)
        cam = scene.beginFrame()
        opts = scene.renderOpts(count_rays=1)
        gpu_ctx.uploadScene(scene.desc)
        a = gpu_ctx.renderFrame(cam, opts)
        pr, sh = gpu_ctx.rayStats()
        st = {}
        ref = orc.render_frame(scene.desc, cam, opts, 2, st)
        assert np.array_equal(np.isnan(a), np.isnan(ref)), seed
        md, nbad, nne = maxdiff(a, ref)
        assert md <= TOL, (seed, md)
        assert (pr, sh) == (st["primary"], st["shadow"]), seed
        differing += nne
    print("ground fuzz: %d differing floats over 80 scenes" % differing)


def test_headline_config_full_frame(gpu_ctx):
    """The bench workload itself — lecture5.sdl, 3840x2160, the reference's 5-tap AA — float for
    float against the oracle (all host cores; a few seconds)."""
    s = c2.parseSceneFromFile(os.path.join(SCENES, "lecture5.sdl"))
    s.setFrameSize(3840, 2160)
    s.setAA(True)
    cam = s.beginFrame()
    opts = s.renderOpts(count_rays=1)
    assert opts.taps == 5
    gpu_ctx.uploadScene(s.desc)
    gpu = gpu_ctx.renderFrame(cam, opts)
    primary, shadow = gpu_ctx.rayStats()
    st = {}
    ref = orc.render_frame(s.desc, cam, opts, 0, st)
    md, nbad, nne = maxdiff(gpu, ref)
    print("lecture5 4K x5: max|d|=%.3g, !=: %d of %d floats" % (md, nne, gpu.size))
    assert md <= TOL and nbad == 0
    assert (primary, shadow) == (st["primary"], st["shadow"]) == (41472000, 41472000)


def test_fp64_lean_sequences_match_the_compilers_expansions_on_the_device():
    """chess2rt_amd/csrc/fp64_lean.h (what the production kernel instances divide, take square roots and
    normalise with) against hipcc's own expansions of `/` and `sqrt` evaluated in the same kernel, and against
    the host's IEEE operations on a sampled slice: 2^30 operands per routine incl. structured significands
    (all ones, single bits, both sides of powers of two), every squared length within 64 ulp of 1, window
    edges, zeros / subnormals / infinities / NaNs refused (tests/fp64_lean_check.hip; the long sweep is
    scripts/fp64_lean_sweep.sh)."""
    import subprocess

    exe = os.path.join(os.path.dirname(os.path.abspath(__file__)), "fp64_lean_check")
    out = subprocess.run([exe, "30"], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout + out.stderr
    rep = json.loads(out.stdout.strip().splitlines()[-1])
    print(rep)
    assert rep["operands_per_routine"] >= 1 << 30 and rep["host_sample"] >= 1 << 20
    for k in ("div_window", "div_tracer_range", "sqrt_lean", "inv_len", "renormalise", "unit_len_exhaustive", "window_edges",
              "div_by_rounded_reciprocal", "div_by_unit_len", "div_f32", "host_mismatches", "not_near_one"):
        assert rep[k] == 0, (k, rep)
    assert rep["inv_len_all_ones_significands"] > 0   # the patched tie case was exercised


def test_tiles_outside_the_lean_windows_are_redone_through_the_exact_path(gpu_ctx, tmp_path):
    """The production instances render a tile optimistically with the shortened divide / sqrt sequences and
    render it AGAIN with the compiler's IEEE expansions when a lane met an operand outside their window
    (c2rt_trace.inc).  (a) On the shipped scenes nothing is redone: the lean path is what runs and what
    every other test compares with the oracle.  (b) A zero numerator (the eye exactly on the plane of a cube
    face), coordinates beyond 1e+72 in a sum of squares, a NaN camera: tiles are redone (counted by
    c2rt_get_exact_redos) and the frames still match the oracle, float for float."""
    for name in sorted(CONFIGS):
        scene, cam, opts = load_config(name)
        gpu_ctx.uploadScene(scene.desc)
        before = gpu_ctx.exactRedos()
        gpu_ctx.renderFrame(cam, opts)
        redone = gpu_ctx.exactRedos() - before
        if name.startswith("csg_corner"):
            # build-authored corner cases (Op(a, a), a Diff whose right subtree contains its left primitive): stepped
            # origins coincide with surfaces exactly, numerators are exact zeros — a few tiles take the exact path
            assert 0 < redone < 300, (name, redone)
        else:
            assert redone == 0, (name, redone)
    cases = {
        # cube top face at y = 1 = the eye's height: (o.y - yhi) is exactly 0 for every ray that reaches the face test
        "eye_on_face_plane": '''Scene { Camera { pos 0 1 -5; fov 60 }
            Lights { PointLight "k" { pos 3 10 -4; color 1 1 1; power 300 } }
            Geometries { Cube "c" { center 0 0.5 0; side 1 }; Sphere "s" { center 0 0.5 0; R 0.6 }; CsgDiff "d" { left "c"; right "s" }; Plane "p" { y 0 } }
            Shaders { Phong "l" { color 0.9 0.4 0.1 }; Lambert "f" { } }
            Nodes { Node "n" { geometry "d"; shader "l" }; Node "g" { geometry "p"; shader "f" } } }''',
        # squared distances of 1e160: far outside [2^-240, 2^240)
        "astronomic_scale": '''Scene { Camera { pos 0 0 0; fov 50 }
            Lights { PointLight "k" { pos 0 6e80 0; color 1 1 1; power 1e164 } }
            Geometries { Sphere "s" { center 0 0 5e80; R 1e80 }; Cube "c" { center 2.5e80 0 6e80; side 1.5e80 } }
            Shaders { Lambert "l" { } }
            Nodes { Node "a" { geometry "s"; shader "l" }; Node "b" { geometry "c"; shader "l" } } }''',
        "nan_camera": '''Scene { Camera { fov 60 }
            Lights { PointLight "k" { pos 0 10 0; color 1 1 1; power 100 } }
            Geometries { Sphere "s" { center 0 0 5; R 1 } }
            Shaders { Phong "l" { } }
            Nodes { Node "n" { geometry "s"; shader "l" } } }''',
    }
    for name, text in cases.items():
        scene = _load_text(tmp_path, text, name + ".sdl")
        scene.setFrameSize(160, 120)
        for taps in (1, 5):
            scene.setAA(taps == 5)
            cam = scene.beginFrame()
            opts = scene.renderOpts(taps=taps, count_rays=1)
            gpu_ctx.uploadScene(scene.desc)
            before = gpu_ctx.exactRedos()
            gpu = gpu_ctx.renderFrame(cam, opts)   # the production instance AND the exact-only counting instance, bit-equal
            redone = gpu_ctx.exactRedos() - before
            ref = orc.render_frame(scene.desc, cam, opts, 0)
            assert np.array_equal(np.isnan(gpu), np.isnan(ref)), name
            md, nbad, nne = maxdiff(gpu, ref)
            print("%s x%d: %d tiles redone, max|d|=%.3g, !=: %d" % (name, taps, redone, md, nne))
            assert redone > 0, name
            assert md <= TOL and nbad == 0, name


def test_interactive_camera_sequence_matches_the_oracle(gpu_ctx):
    """The GUI loop (gui/raytracer_demo.d:268-340): Camera.move / Camera.rotate (rt/camera.d:181-229) ->
    beginFrame -> renderRT, ten frames in a row on one context, each compared with the oracle.  The path walks the
    eye INTO the CsgDiff's box and into the globe, turns until nodes straddle the eye plane and looks straight down
    with the light behind the eye: the per-frame culling rectangles, hulls and light half-spaces are recomputed for
    all of those (c2rt_api.cpp: cull_rect_of, light_side_of)."""
    scene = c2.parseSceneFromFile(os.path.join(SCENES, "lecture5.sdl"))
    scene.setFrameSize(320, 240)
    gpu_ctx.uploadScene(scene.desc)
    steps = [
        ((0, 0, 0), (0, 0, 0)),          # the file's camera
        ((-25, 0, 0), (0, 0, 100)),      # turn towards the CSG object, walk forward
        ((0, 0, 0), (0, 0, 120)),        # ... to just in front of / inside its bounding box
        ((0, 0, 60), (0, 0, 0)),         # look up: floor leaves the view
        ((90, 0, 0), (10, 0, 0)),        # quarter turn: objects straddle the eye plane
        ((0, 0, -100), (0, 0, 0)),       # pitch clamps at -90: straight down, the light behind the eye
        ((0, 0, 45), (0, 300, 0)),       # high above the scene
        ((180, 0, 0), (0, 0, -50)),      # look back
        ((-60, 30, -20), (150, -200, 80)),  # rolled camera, towards the globe
        ((0, -30, 0), (0, 0, 60)),       # ... and closer
    ]
    cam = scene.beginFrame()
    for i, (rot, mov) in enumerate(steps):
        scene.rotateCamera(*rot)
        cam = scene.beginFrame()         # the GUI calls beginFrame between rotate and move as part of the frame
        scene.moveCamera(*mov)
        cam = scene.beginFrame()
        scene.setAA(i % 2 == 1)
        opts = scene.renderOpts(count_rays=1)
        gpu = gpu_ctx.renderFrame(cam, opts)
        primary, shadow = gpu_ctx.rayStats()
        stats = {}
        ref = orc.render_frame(scene.desc, cam, opts, 0, stats)
        md, nbad, nne = maxdiff(gpu, ref)
        print("step %d pos %s yaw %.0f pitch %.0f roll %.0f: max|d|=%.3g, !=: %d" % (i, [round(v, 1) for v in scene.camera.pos], scene.camera.yaw, scene.camera.pitch, scene.camera.roll, md, nne))
        assert md <= TOL and nbad == 0, i
        assert (primary, shadow) == (stats["primary"], stats["shadow"]), i
    # the eye really went inside boxes: at some step the CSG node's cull rectangle covered the frame
    assert scene.camera.pos[1] != 165.0


def test_frames_of_one_context_on_two_streams(gpu_ctx):
    """c2rt_render_frame_device on two different streams of ONE context, back to back, for a nested-CSG scene
    (its frames use a retry list and a tile-mask table): include/c2rt.h promises that frames on different streams
    are independent — each stream has its own scratch slot in the context — so both frames are complete and equal
    the frames rendered alone, whatever their overlap, and no call waits on the host for an earlier frame.  Counted
    frames (shared ray counters) on two streams are ordered on the device and each reports its own counts."""
    import torch

    scene, cam_a, opts = load_config("csg_stress_320x240_t1")
    gpu_ctx.uploadScene(scene.desc)
    scene.rotateCamera(20, 0, -5)
    cam_b = scene.beginFrame()
    dev = torch.device("cuda", 0)
    H, W = opts.height, opts.width
    alone = []
    for cam in (cam_a, cam_b):
        t = torch.empty((H, W, 3), dtype=torch.float32, device=dev)
        gpu_ctx.renderFrameDevice(cam, opts, t.data_ptr(), torch.cuda.current_stream(dev).cuda_stream)
        torch.cuda.synchronize(dev)
        alone.append(t.clone())
    s1, s2 = torch.cuda.Stream(dev), torch.cuda.Stream(dev)
    for rep in range(5):
        a = torch.full((H, W, 3), -1.0, dtype=torch.float32, device=dev)
        b = torch.full((H, W, 3), -1.0, dtype=torch.float32, device=dev)
        torch.cuda.synchronize(dev)
        gpu_ctx.renderFrameDevice(cam_a, opts, a.data_ptr(), s1.cuda_stream)
        gpu_ctx.renderFrameDevice(cam_b, opts, b.data_ptr(), s2.cuda_stream)
        gpu_ctx.renderFrameDevice(cam_a, opts, a.data_ptr(), s1.cuda_stream)
        torch.cuda.synchronize(dev)
        assert torch.equal(a, alone[0]) and torch.equal(b, alone[1]), rep
    # and a blocking host-output frame right after a stream-async one
    gpu_ctx.renderFrameDevice(cam_b, opts, b.data_ptr(), s2.cuda_stream)
    host = gpu_ctx.renderFrame(cam_a, opts)
    torch.cuda.synchronize(dev)
    assert np.array_equal(host, alone[0].cpu().numpy()) and torch.equal(b, alone[1])
    # counted frames on two streams: ordered among themselves (the counters are shared), frames and counts intact
    _, _, copts = load_config("csg_stress_320x240_t1", count_rays=1)
    gpu_ctx.renderFrame(cam_b, copts)
    rays_b = gpu_ctx.rayStats()
    for rep in range(3):
        gpu_ctx.renderFrameDevice(cam_a, copts, a.data_ptr(), s1.cuda_stream)
        gpu_ctx.renderFrameDevice(cam_b, copts, b.data_ptr(), s2.cuda_stream)
        assert gpu_ctx.rayStats() == rays_b
        torch.cuda.synchronize(dev)
        assert torch.equal(a, alone[0]) and torch.equal(b, alone[1]), rep
    # no call blocks the host: a long frame (2048x1536 x5 taps of the depth-4 scene, several ms) on
    # s1, then a frame on s2 — the second CALL is back while the first FRAME is still running
    big = scene.renderOpts(taps=c2.TAPS_REF5)
    scene.setFrameSize(2048, 1536)
    cam_big = scene.beginFrame()
    big = scene.renderOpts(taps=c2.TAPS_REF5)
    t_big = torch.empty((1536, 2048, 3), dtype=torch.float32, device=dev)
    gpu_ctx.renderFrameDevice(cam_big, big, t_big.data_ptr(), s1.cuda_stream)      # warm-up (allocations)
    torch.cuda.synchronize(dev)
    done = torch.cuda.Event()
    gpu_ctx.renderFrameDevice(cam_big, big, t_big.data_ptr(), s1.cuda_stream)
    done.record(s1)
    gpu_ctx.renderFrameDevice(cam_b, opts, b.data_ptr(), s2.cuda_stream)
    first_still_running = not done.query()
    torch.cuda.synchronize(dev)
    assert first_still_running, "c2rt_render_frame_device waited on the host for another stream's frame"
    assert torch.equal(b, alone[1])
    scene.setFrameSize(W, H)
    small = opts
    # more streams than scratch slots (16): the least recently used slot is recycled, frames unchanged
    many = [torch.cuda.Stream(dev) for _ in range(19)]
    outs = [torch.full((H, W, 3), -1.0, dtype=torch.float32, device=dev) for _ in many]
    for i, st in enumerate(many):
        gpu_ctx.renderFrameDevice(cam_a if i % 2 == 0 else cam_b, small, outs[i].data_ptr(), st.cuda_stream)
    torch.cuda.synchronize(dev)
    for i in range(len(many)):
        assert torch.equal(outs[i], alone[i % 2]), i


def test_the_callers_stream_may_be_destroyed_after_the_call(gpu_ctx):
    """The library keeps no reference to the caller's stream (round-3 advisor): render on a stream, destroy it,
    then every kind of call on the same context still works — blocking frames, the ray counters of the frame that
    ran on the destroyed stream, a frame on another stream, pin / unpin."""
    # the HIP runtime libc2rt.so itself is bound to: symbols looked up through ITS handle resolve in its own dependency
    # tree.  (C.CDLL("libamdhip64.so") can map a second copy of the runtime — torch bundles one — and a stream of one
    # runtime handed to the other aborts inside HIP.)
    hip = _abi.load_library()
    hip.hipStreamCreate.argtypes = [C.POINTER(C.c_void_p)]
    hip.hipStreamDestroy.argtypes = [C.c_void_p]
    hip.hipStreamSynchronize.argtypes = [C.c_void_p]
    hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
    hip.hipFree.argtypes = [C.c_void_p]
    hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    scene, cam, opts = load_config("csg_stress_320x240_t1", count_rays=1)
    gpu_ctx.uploadScene(scene.desc)
    want = gpu_ctx.renderFrame(cam, opts)
    rays = gpu_ctx.rayStats()
    dev = C.c_void_p()
    assert hip.hipMalloc(C.byref(dev), want.nbytes) == 0
    try:
        for _ in range(3):
            st = C.c_void_p()
            assert hip.hipStreamCreate(C.byref(st)) == 0
            gpu_ctx.renderFrameDevice(cam, opts, dev.value, st.value)
            assert hip.hipStreamSynchronize(st) == 0       # (what a caller does before it lets go of a stream)
            assert hip.hipStreamDestroy(st) == 0           # the handle is gone; the library must not touch it again
            assert gpu_ctx.rayStats() == rays              # counters of the frame that ran on it
            st2 = C.c_void_p()
            assert hip.hipStreamCreate(C.byref(st2)) == 0
            gpu_ctx.renderFrameDevice(cam, opts, dev.value, st2.value)     # a counted frame: ordered behind the dead stream's event
            host = gpu_ctx.renderFrame(cam, opts)                          # blocking entry point drains first
            assert np.array_equal(host, want)
            got = np.empty_like(want)
            assert hip.hipMemcpy(got.ctypes.data, dev, want.nbytes, 2) == 0  # hipMemcpyDeviceToHost
            assert np.array_equal(got, want)
            assert hip.hipStreamSynchronize(st2) == 0
            assert hip.hipStreamDestroy(st2) == 0
            pinned = np.zeros_like(want)
            gpu_ctx.pinHostBuffer(pinned)
            gpu_ctx.renderFrameInto(cam, opts, pinned)
            gpu_ctx.unpinHostBuffer(pinned)
            assert np.array_equal(pinned, want)
    finally:
        hip.hipFree(dev)
