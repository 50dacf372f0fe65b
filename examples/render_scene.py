#!/usr/bin/env python3
"""Render a Chess2RT scene (.sdl / .json) on the GPU and save it as a BMP —
what `chess2rt --file=<scene>` + F12 (gui/raytracer_demo.d:227-238) produce,
without the SDL2 window.

  python examples/render_scene.py tests/golden/scenes/lecture5.sdl out.bmp --size 1920 1080
"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import chess2rt_amd as c2


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("scene")
    ap.add_argument("out")
    ap.add_argument("--size", type=int, nargs=2, metavar=("W", "H"))
    ap.add_argument("--no-aa", action="store_true")
    ap.add_argument("--no-dof", action="store_true")
    args = ap.parse_args()
    scene = c2.parseSceneFromFile(args.scene)
    if args.size:
        scene.setFrameSize(*args.size)
    if args.no_aa:
        scene.setAA(False)
    if args.no_dof:
        scene.setDof(False)
    renderer = c2.Renderer(scene)
    t = time.perf_counter()
    frame = renderer.renderRT()           # (H, W, 3) float32 linear RGB, like Image!Color
    dt = time.perf_counter() - t
    with open(args.out, "wb") as f:
        f.write(c2.saveBmp(frame))        # sRGB-encoded 24-bpp BMP (rt/color.d toRGB32 + imageio/bmp.d saveBmp)
    s = scene.settings
    print("%s: %dx%d, AA %s -> %s (%.1f ms incl. upload and copy-back)" % (scene.name, s.frame_width, s.frame_height, bool(s.aa_enabled), args.out, dt * 1e3))


if __name__ == "__main__":
    main()
