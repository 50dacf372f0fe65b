#!/usr/bin/env python3
"""Fixture generator: decodes the reference's remaining texture files (data/texture/*.bmp — 8-, 24- and 32-bpp,
SURVEY.md 8(f3); the three that scenes of tests/golden/scenes use are committed there already) with the host
decoder (the C++ mirror of imageio/bmp.d:60-193 behind c2rt_host_bmp_decode) and records, per file, the SHA-256
of the FILE (so that the test knows it is looking at the same input), its header fields and the SHA-256 of the
decoded float32 (H, W, 3) array.  Run in the build container, where /root/reference exists:
    python tests/golden/make_bmp_texture_hashes.py > tests/golden/bmp_texture_hashes.json
The BMP files themselves stay in /root/reference (5 MB of data the parity suite does not need on the GPU box)."""
import hashlib
import json
import os
import struct
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import chess2rt_amd as c2  # noqa: E402
import oracle_lib as orc  # noqa: E402
import numpy as np  # noqa: E402

DIR = "/root/reference/data/texture"
FILES = ["heightfield.bmp", "hf_color.bmp", "lava.bmp", "wood.bmp", "zar-bump.bmp", "zar-texture.bmp"]
out = {"source_dir": "data/texture (reference repository)", "files": {}}
for f in FILES:
    data = open(os.path.join(DIR, f), "rb").read()
    w, h = struct.unpack_from("<ii", data, 18)
    bpp, = struct.unpack_from("<H", data, 28)
    img = c2.loadBmpImage(data)
    ref, _ = orc.bmp_decode(data)
    assert np.array_equal(img, ref), f
    out["files"][f] = {"file_sha256": hashlib.sha256(data).hexdigest(), "width": w, "height": abs(h), "bpp": bpp,
                       "decoded_sha256": hashlib.sha256(np.ascontiguousarray(img).tobytes()).hexdigest(),
                       "mean_rgb": [float(x) for x in img.reshape(-1, 3).astype(np.float64).mean(axis=0)]}
json.dump(out, sys.stdout, indent=1)
print()
