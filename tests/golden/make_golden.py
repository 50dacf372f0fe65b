#!/usr/bin/env python3
"""Generates the committed golden fixtures from the CPU oracle.

The reference (D) cannot be run anywhere in this pipeline, so these vectors
pin the ORACLE (oracle/c2rt_oracle.c) against regressions and give the GPU
tests fixed expected outputs; what pins the oracle to the reference is in
bmp_known_answers.json (the reference's own unittests) and
lecture4_anchors.json (hand-derived).

Run from the repo root:  python tests/golden/make_golden.py
Writes tests/golden/frames.json, frames_crops.npz, unit_vectors.json.
"""
import hashlib
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import chess2rt_amd as c2  # host loader only (no GPU needed)
import oracle_lib as orc
from golden_configs import CONFIGS, crop_offsets, load_config  # noqa: E402


def main():
    frames = {}
    crops = {}
    for name in CONFIGS:
        scene, cam, opts = load_config(name)
        stats = {}
        img = orc.render_frame(scene.desc, cam, opts, 0, stats)
        entry = {
            "config": CONFIGS[name],
            "sha256": hashlib.sha256(np.ascontiguousarray(img).tobytes()).hexdigest(),
            "min": [float(x) for x in img.min(axis=(0, 1))],
            "max": [float(x) for x in img.max(axis=(0, 1))],
            "mean": [float(x) for x in img.astype(np.float64).mean(axis=(0, 1))],
            "primary_rays": stats["primary"],
            "shadow_rays": stats["shadow"],
            "crops": [],
        }
        for k, (x0, y0) in enumerate(crop_offsets(opts.width, opts.height)):
            crops["%s/%d" % (name, k)] = img[y0:y0 + 64, x0:x0 + 64].copy()
            entry["crops"].append([x0, y0])
        frames[name] = entry
        print(name, entry["sha256"][:12], entry["mean"])
    with open(os.path.join(HERE, "frames.json"), "w") as f:
        json.dump(frames, f, indent=1, sort_keys=True)
    np.savez_compressed(os.path.join(HERE, "frames_crops.npz"), **crops)

    # ---- unit vectors: oracle outputs for fixed inputs (a6, a8, a10-a23) ----
    import ctypes as C
    from unit_inputs import unit_cases  # noqa: E402

    out = unit_cases(orc, c2)
    with open(os.path.join(HERE, "unit_vectors.json"), "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
    print("unit vectors:", {k: len(v) for k, v in out.items()})


if __name__ == "__main__":
    main()
