"""Host-side mirror (C++ in libc2rt.so): scene loader, Camera, Transform,
BMP decode / encode, display encode — against the reference's documented
behaviour and against the oracle's independent restatement.  No GPU."""
import ctypes as C
import json
import os

import numpy as np
import pytest

import chess2rt_amd as c2
import oracle_lib as orc
from chess2rt_amd import _abi


def load_text(tmp_path, text, ext=".sdl", name="s"):
    p = tmp_path / (name + ext)
    p.write_text(text)
    return c2.parseSceneFromFile(str(p))


def test_lecture4_sdl_and_json_describe_the_same_scene(scenes_dir):
    a = c2.parseSceneFromFile(os.path.join(scenes_dir, "lecture4.sdl"))
    b = c2.parseSceneFromFile(os.path.join(scenes_dir, "lecture4.json"))
    assert a.name == "lecture4" and b.name == ""           # the JSON file has no "Name"
    sa, sb = a.settings, b.settings
    assert (sa.frame_width, sa.frame_height) == (640, 480) == (sb.frame_width, sb.frame_height)
    assert sa.aa_enabled == 1 and sb.aa_enabled == 0        # AAEnabled defaults to true (rt/global_settings.d:23)
    assert sa.prepass_enabled == 1 and sb.prepass_enabled == 0 and sa.bucket_size == 48 and sa.max_trace_depth == 4
    da, db = a.desc.contents, b.desc.contents
    for f in ("n_geoms", "n_textures", "n_shaders", "n_lights", "n_nodes"):
        assert getattr(da, f) == getattr(db, f) == 1
    assert da.geom_type[0] == _abi.GEOM_PLANE and da.geom_param[0] == 2.0 and np.isnan(da.geom_param[1])
    assert [da.tex_color[i] for i in range(6)] == [0, 0, 0, 0, 0.5, 1.0] and da.tex_param[0] == 5.0
    assert da.shader_type[0] == _abi.SHADER_LAMBERT and da.shader_texture[0] == 0
    assert [da.shader_color[i] for i in range(3)] == [1, 1, 1]   # Lambert default colour
    assert [da.light_pos[i] for i in range(3)] == [-30, 100, 250] and da.light_power[0] == 50000
    t = [da.node_transform[i] for i in range(30)]
    I = [1, 0, 0, 0, 1, 0, 0, 0, 1]
    assert t == I + I + I + [0, 0, 0]
    ca, cb = a.camera, b.camera
    assert list(ca.pos) == [0, 165, 0] and ca.pitch == -30 and ca.fov == 90 and ca.aspect == 640 / 480 == cb.aspect


def test_lecture5_tables(scenes_dir):
    s = c2.parseSceneFromFile(os.path.join(scenes_dir, "lecture5.sdl"))
    d = s.desc.contents
    assert (d.n_geoms, d.n_textures, d.n_shaders, d.n_lights, d.n_nodes) == (6, 2, 4, 1, 6)
    assert [d.geom_type[i] for i in range(6)] == [_abi.GEOM_PLANE, _abi.GEOM_SPHERE, _abi.GEOM_CUBE, _abi.GEOM_SPHERE, _abi.GEOM_CSG_DIFF, _abi.GEOM_SPHERE]
    assert (d.geom_child[8], d.geom_child[9]) == (2, 3)          # diff: left cube, right sphere
    assert [d.geom_param[20 + i] for i in range(4)] == [0, 0, 0, 15]   # Sphere "S": centre reset to 0 when absent
    assert d.geom_param[0] == -0.01                                # `Plane "floor" {` name-as-value form
    assert (d.tex_width[0], d.tex_height[0], d.tex_width[1], d.tex_height[1]) == (256, 256, 800, 400)
    assert d.tex_offset[1] == 256 * 256 and d.n_texels == 256 * 256 + 800 * 400
    assert d.tex_scaling[0] == np.float32(0.005) and d.tex_scaling[1] == 1.0
    assert [d.shader_type[i] for i in range(4)] == [0, 0, 1, 1] and d.shader_exponent[2] == 60 and d.shader_strength[3] == 1.0
    assert [d.node_transform[30 * 3 + 27 + i] for i in range(3)] == [100, 15, 256]     # translate
    assert list(d.ambient) == [np.float32(0.2)] * 3
    assert s.settings.prepass_enabled == 0 and s.settings.dynamic_aspect_ratio == 1


def test_zaphod_ignores_misplaced_keys_and_file_order(scenes_dir):
    s = c2.parseSceneFromFile(os.path.join(scenes_dir, "zaphod.sdl"))
    cam = s.camera
    # numSamples/fNumber sit inside the PointLight block where nothing reads them
    assert cam.dof == 1 and cam.num_samples == 25 and cam.f_number == 1.0 and cam.disc_multiplier == 10.0
    assert cam.focal_plane_dist == 25.29 and (cam.frame_width, cam.frame_height) == (645, 430)
    d = s.desc.contents
    t = [d.node_transform[i] for i in range(30)]
    assert t[0:9] == [10, 0, 0, 0, 10, 0, 0, 0, 10]
    np.testing.assert_allclose(t[9:18], [0.1, 0, 0, 0, 0.1, 0, 0, 0, 0.1], rtol=1e-15)
    assert (d.tex_width[0], d.tex_height[0]) == (768, 768)


def test_defaults_and_reference_quirks(tmp_path):
    s = load_text(tmp_path, '''Scene {
      Lights { PointLight "l" { pos 1 2 3; color 1 1 1; power 10 } }
      Geometries { Plane "p" { }; Sphere "s" { }; Cube "c" { } }
      Textures { Checker "k" { } }
      Shaders { Phong "ph" { exponent 1e9; strength -3 }; Lambert "la" { color 0.5 0.25 0.125; texture "nope" } }
      Nodes {
        Node "a" { geometry "s"; shader "ph"; scale 2 4 8; rotate 3 5 7; translate 1 1 1; translate 9 8 7 }
        Node "b" { geometry "c"; shader "la"; bump "k" }
      }
    }''')
    st = s.settings
    assert (st.frame_width, st.frame_height, st.aa_enabled, st.gi_enabled, st.paths_per_pixel) == (640, 480, 1, 0, 40)
    assert list(st.ambient) == [0, 0, 0]
    d = s.desc.contents
    assert np.isnan(d.geom_param[0]) and np.isnan(d.geom_param[1])              # Plane(): y, limit = NaN
    assert [d.geom_param[4 + i] for i in range(4)] == [0, 0, 0, 1] and [d.geom_param[8 + i] for i in range(4)] == [0, 0, 0, 1]
    assert [d.tex_color[i] for i in range(6)] == [0, 0, 0, 1, 1, 1] and d.tex_param[0] == 1.0
    assert d.shader_exponent[0] == 1e6 and d.shader_strength[0] == 0.0          # clamp [1e-6,1e6], [0,1e6]
    assert d.shader_texture[1] == -1                                            # unknown texture name -> none
    assert d.node_bump[1] == 0 and d.node_bump[0] == -1
    t = [d.node_transform[i] for i in range(30)]
    assert t[0:9] == [6, 0, 0, 0, 20, 0, 0, 0, 56]                              # "rotate" applied as a SCALE (rt/node.d:89-90)
    assert t[27:30] == [1, 1, 1]                                                # getChild takes the FIRST `translate`


@pytest.mark.parametrize("text,needle", [
    ('Scene { Geometries { Sphere "a" { }; Cube "a" { } } }', "Duplicate"),
    ('Scene { Geometries { Torus "a" { } } }', "Unknown object type"),
    ('Scene { Lights { Sphere "a" { } } }', "Unknown object type"),
    ('Scene { Nodes { Node "n" { geometry "missing"; shader "x" } } }', "unknown geometry"),
    ('Scene { Geometries { CsgDiff "d" { left "later"; right "later" }; Sphere "later" { } } }', "unknown geometry"),
    ('Scene { GlobalSettings { frameWidth "wide" } }', "expected an integer"),
    ('Scene { Textures { BitmapTexture "t" { file "nope.bmp" } } }', "cannot read"),
    ('Scene { Textures { BitmapTexture "t" { file "x.exr" } } }', "UnknownImageType"),
    ('Scene { Geometries { Sphere "a" { ', "missing '}'"),
])
def test_loader_errors(tmp_path, text, needle):
    with pytest.raises(c2.C2rtError) as e:
        load_text(tmp_path, text)
    assert needle.lower() in str(e.value).lower()
    assert e.value.status in (_abi.ERR_PARSE, _abi.ERR_IO)


def test_missing_scene_file_and_unknown_extension(tmp_path):
    with pytest.raises(c2.C2rtError) as e:
        c2.parseSceneFromFile(str(tmp_path / "nope.sdl"))
    assert e.value.status == _abi.ERR_IO
    with pytest.raises(c2.C2rtError) as e:
        load_text(tmp_path, "{}", ext=".yaml")
    assert e.value.status == _abi.ERR_PARSE


def test_sdl_syntax_subset(tmp_path):
    s = load_text(tmp_path, '''
    # hash comment
    -- dash comment
    Scene {   // trailing comment
      Name `raw name`
      GlobalSettings { frameWidth 100L; frameHeight 50 /* inline */; AAEnabled off; prepassOnly false
        ambientLightColor 0.5f 1 \\
           0.25d }
      Camera { pos 1.5 -2 3; fov 60; dof on; numSamples 9 }
    }''')
    assert s.name == "raw name"
    st = s.settings
    assert (st.frame_width, st.frame_height, st.aa_enabled) == (100, 50, 0) and list(st.ambient) == [0.5, 1.0, 0.25]
    assert list(s.camera.pos) == [1.5, -2, 3] and s.camera.dof == 1 and s.camera.num_samples == 9 and s.camera.aspect == 2.0


def test_camera_begin_frame_matches_the_oracle_bitwise(scenes_dir):
    s = c2.parseSceneFromFile(os.path.join(scenes_dir, "zaphod.sdl"))   # yaw, pitch and roll all non-zero
    for (w, h) in [(645, 430), (1920, 1080), (333, 217)]:
        s.setFrameSize(w, h)
        cam = s.beginFrame()
        hc = s.camera
        o = _abi.CameraFrame()
        orc.lib().orc_camera_begin_frame(orc.vec3(*hc.pos), hc.yaw, hc.pitch, hc.roll, hc.fov, w, h, C.byref(o))
        for k in ("pos", "up_left", "up_right", "down_left", "right_dir", "up_dir", "front_dir"):
            assert list(getattr(o, k)) == list(getattr(cam, k)), k
        assert (cam.frame_width, cam.frame_height) == (w, h) and cam.dof == 1 and cam.disc_multiplier == 10.0
    # pitch -30 looks DOWN: the sign convention of gfm rotateX under row-vector mul
    s4 = c2.parseSceneFromFile(os.path.join(scenes_dir, "lecture4.sdl"))
    f = list(s4.beginFrame().front_dir)
    assert f[0] == 0 and abs(f[1] + 0.5) < 1e-15 and abs(f[2] - np.sqrt(0.75)) < 1e-15


def test_camera_move_and_rotate(scenes_dir):
    s = c2.parseSceneFromFile(os.path.join(scenes_dir, "lecture4.sdl"))
    cam = s.beginFrame()
    s.moveCamera(2.0, 3.0, 4.0)
    exp = np.array([0, 165, 0.0]) + 2 * np.array(cam.right_dir) + 3 * np.array(cam.up_dir) + 4 * np.array(cam.front_dir)
    np.testing.assert_allclose(list(s.camera.pos), exp, rtol=0, atol=1e-12)
    s.rotateCamera(10, 5, -100)
    c = s.camera
    assert (c.yaw, c.roll, c.pitch) == (10, 5, -90)      # pitch clamped to [-90, 90]


def test_transform_algebra_matches_the_oracle(tmp_path):
    s = load_text(tmp_path, '''Scene {
      Geometries { Sphere "s" { } }
      Shaders { Lambert "l" { } }
      Nodes { Node "n" { geometry "s"; shader "l"; scale 2 3 5; rotate 0.5 0.25 4; translate -1 2 -3 } }
    }''')
    t = (C.c_double * 30)()
    L = orc.lib()
    L.orc_transform_reset(t)
    L.orc_transform_scale(t, 2, 3, 5)
    L.orc_transform_scale(t, 0.5, 0.25, 4)
    L.orc_transform_translate(t, orc.vec3(-1, 2, -3))
    assert [s.desc.contents.node_transform[i] for i in range(30)] == list(t)
    # the real rotate (never reached from scene files) is orthonormal and inverts
    L.orc_transform_reset(t)
    L.orc_transform_rotate(t, 30, 20, 10)
    m, inv = np.array(t[0:9]).reshape(3, 3), np.array(t[9:18]).reshape(3, 3)
    np.testing.assert_allclose(m @ inv, np.eye(3), atol=1e-15)
    np.testing.assert_allclose(np.array(t[18:27]).reshape(3, 3), inv.T, atol=0)


def test_bmp_decode_known_answers_and_oracle_agreement(golden_dir, scenes_dir):
    for case in json.load(open(os.path.join(golden_dir, "bmp_known_answers.json")))["cases"]:
        data = bytes.fromhex(case["bytes_hex"])
        img = c2.loadBmpImage(data)
        assert img.shape == (case["height"], case["width"], 3)
        for key, word in case["pixels_xy"].items():
            x, y = map(int, key.split(","))
            exp = [np.float32((word >> sft) & 0xFF) * np.float32(1.0 / 255.0) for sft in (16, 8, 0)]
            assert list(img[y, x]) == exp
    for f in ("floor.bmp", "world.bmp", "texture/zaphod.bmp"):     # 24-bpp, 8-bpp paletted (161 colours), 8-bpp
        data = open(os.path.join(scenes_dir, f), "rb").read()
        a = c2.loadBmpImage(data)
        b, _ = orc.bmp_decode(data)
        assert np.array_equal(a, b), f


def test_bmp_decode_of_the_references_other_textures(golden_dir):
    """data/texture/{heightfield (8-bpp), hf_color, wood, zar-bump, zar-texture (24-bpp), lava
    (32-bpp)}.bmp through the host decoder (imageio/bmp.d:60-193 mirrored) and the oracle's: equal each other and
    equal the committed SHA-256 of the decoded float frame (tests/golden/make_bmp_texture_hashes.py).  The files
    are read where the reference keeps them; they exist in the build container only."""
    import hashlib

    fx = json.load(open(os.path.join(golden_dir, "bmp_texture_hashes.json")))["files"]
    src = "/root/reference/data/texture"
    if not os.path.isdir(src):
        pytest.skip("the reference's texture files are not on this machine")
    assert sorted(fx) == ["heightfield.bmp", "hf_color.bmp", "lava.bmp", "wood.bmp", "zar-bump.bmp", "zar-texture.bmp"]
    assert {e["bpp"] for e in fx.values()} == {8, 24, 32}
    for f, e in fx.items():
        data = open(os.path.join(src, f), "rb").read()
        assert hashlib.sha256(data).hexdigest() == e["file_sha256"], f
        a = c2.loadBmpImage(data)
        b, _ = orc.bmp_decode(data)
        assert a.shape == (e["height"], e["width"], 3) and a.dtype == np.float32, f
        assert np.array_equal(a, b), f
        assert hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest() == e["decoded_sha256"], f
        assert a.min() >= 0.0 and a.max() <= 1.0


def test_bmp_rejects_what_the_reference_cannot_decode():
    hdr = bytearray(70)
    hdr[0:2] = b"BM"
    hdr[10] = 54
    hdr[14] = 40
    hdr[18] = 2
    hdr[22] = 2
    hdr[26] = 1
    for bpp in (16, 4, 1, 7):
        hdr[28] = bpp
        with pytest.raises(c2.C2rtError):
            c2.loadBmpImage(bytes(hdr))
    with pytest.raises(c2.C2rtError):
        c2.loadBmpImage(b"PNG" + bytes(100))


def test_texture_gamma_matches_the_oracle():
    x = (np.arange(256, dtype=np.float32) * np.float32(1.0 / 255.0)).astype(np.float32)
    for gamma in (2.2, 1.0, 1.8, 0.0, 10.0):
        a, b = x.copy(), x.copy()
        _abi.load_library().c2rt_host_texture_gamma(a.ctypes.data_as(C.c_void_p), a.size, gamma)
        orc.lib().orc_texture_gamma(b.ctypes.data_as(C.c_void_p), b.size, gamma)
        assert np.array_equal(a, b)
    a = x.copy()
    _abi.load_library().c2rt_host_texture_gamma(a.ctypes.data_as(C.c_void_p), a.size, 2.2)
    assert a[0] == 0 and a[255] == 1 and a[10] == x[10] / np.float32(12.92) and abs(a[128] - 0.2158605) < 1e-6


def test_display_encode_and_bmp_writer(golden_dir):
    lib = _abi.load_library()
    for case in json.load(open(os.path.join(golden_dir, "unit_vectors.json")))["color_to_rgb32"]:
        c = (C.c_float * 3)(*case["rgb"])
        assert lib.c2rt_host_color_to_rgb32(c) == case["rgb32"] == orc.lib().orc_color_to_rgb32(c)
    # 12.02 (not 12.92) below the knee — rt/color.d:200-201
    c = (C.c_float * 3)(0.003, 0.003, 0.003)
    assert lib.c2rt_host_color_to_rgb32(c) & 0xFF == int(np.floor(np.float32(np.float32(int(np.float32(0.003) * np.float32(4096)) / np.float32(4096)) * np.float32(12.02)) * np.float32(255)))
    img = np.zeros((2, 3, 3), np.float32)
    img[0, 0] = (1, 0, 0)
    img[1, 2] = (0, 0, 1)
    data = c2.saveBmp(img)
    assert data[:2] == b"BM" and len(data) == 14 + 40 + 3 * 2 * 3        # rows are NOT padded (imageio/bmp.d:199-200, as written)
    assert int.from_bytes(data[2:6], "little") == len(data) and int.from_bytes(data[10:14], "little") == 54
    assert int.from_bytes(data[18:22], "little") == 3 and int.from_bytes(data[22:26], "little") == 2 and data[28] == 24
    assert int.from_bytes(data[38:42], "little") == 2835
    px = data[54:]
    assert px[0:9] == bytes([0, 0, 0, 0, 0, 0, 255, 0, 0])                # bottom row first: (0,1),(1,1),(2,1)=blue as B,G,R
    assert px[9:12] == bytes([0, 0, 255])                                 # top-left red
