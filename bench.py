#!/usr/bin/env python3
"""bench.py — Mray/s and ms/frame of the Chess2RT render hot path on MI355X.

A "step" is one frame: every pixel of the workload traced and shaded by the
HIP kernel behind the C ABI (scene tables, textures and the frame buffer are
resident in HBM when the timed region starts).

N=1 workload (default): data/lecture5.sdl at 3840x2160, 1 sample/pixel — the
configuration BASELINE.json's north_star target is quoted on.
N>1 (one process per GPU, launched by torch.distributed.run): WEAK scaling —
the frame keeps its 16:9 shape and grows to N x the N=1 pixel count, ranks
render interleaved 8-row strips and one RCCL gather brings them to rank 0,
where a copy kernel de-interleaves them.

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import math
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

SCENES = os.path.join(ROOT, "tests", "golden", "scenes")

WORKLOADS = {
    # name: (scene file, width, height, taps, dof)
    "lecture5_4k": ("lecture5.sdl", 3840, 2160, 1, False),       # north_star target config
    "lecture5_4k_aa5": ("lecture5.sdl", 3840, 2160, 5, False),   # as the scene file ships (AAEnabled)
    "lecture5_1080p": ("lecture5.sdl", 1920, 1080, 1, False),    # BASELINE configs[2]
    "lecture4_1080p": ("lecture4.sdl", 1920, 1080, 1, False),    # BASELINE configs[1]
    "zaphod_4k_4spp": ("zaphod.sdl", 3840, 2160, 4, False),      # BASELINE configs[3], DOF off
    "lecture5_8k_4spp": ("lecture5.sdl", 7680, 4320, 4, False),  # BASELINE configs[4] (per-frame size)
}

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def weak_frame(width, height, n):
    """16:9-preserving frame with n x the pixels, rounded to the 8x8 tile."""
    if n == 1:
        return width, height
    s = math.sqrt(n)
    return int(round(width * s / 8)) * 8, int(round(height * s / 8)) * 8


def cpu_baseline(scene, cam, opts, rays_per_frame, budget_s=12.0):
    """The CPU oracle (restatement of the reference algorithm, NOT the D
    binary — no D toolchain exists here) timed on this box's host cores."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib

    cores = len(os.sched_getaffinity(0))
    times = []
    t_all = time.time()
    oracle_lib.render_frame(scene.desc, cam, opts, cores)  # warm-up
    warm = time.time() - t_all
    while len(times) < 5 and (len(times) < 2 or time.time() - t_all < budget_s):
        t = time.time()
        oracle_lib.render_frame(scene.desc, cam, opts, cores)
        times.append(time.time() - t)
        if warm > budget_s and len(times) >= 1:
            break
    med = statistics.median(times)
    return {
        "value": rays_per_frame / med / 1e6,
        "unit": "Mray/s",
        "ms_per_frame": med * 1e3,
        "cores": cores,
        "kind": "port",
        "sample": "%d full frames of the same workload (%dx%d, %d tap(s)) after 1 warm-up, median; pthread pool over 48x48 buckets"
                  % (len(times), opts.width, opts.height, opts.taps),
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="lecture5_4k", choices=sorted(WORKLOADS))
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--strip-height", type=int, default=8)
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    import chess2rt_amd as c2

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d (launch with torch.distributed.run)" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py: no GPU visible; the render path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)

    scene_file, w0, h0, taps, dof = WORKLOADS[args.workload]
    width, height = (weak_frame(w0, h0, world) if args.scaling == "weak" else (w0, h0))
    scene = c2.parseSceneFromFile(os.path.join(SCENES, scene_file))
    scene.setFrameSize(width, height)
    scene.setDof(dof)
    cam = scene.beginFrame()

    ctx = c2.Context(local_rank)
    ctx.uploadScene(scene.desc)
    plan = c2.plan_strips(height, world, args.strip_height)
    opts = scene.renderOpts(taps=taps, strip_height=plan.strip_height, strip_rank=rank, strip_world=world)
    my_rows = ctx.localRows(opts)

    stream = torch.cuda.current_stream(dev)
    local = torch.zeros((plan.rows_pad, width, 3), dtype=torch.float32, device=dev)
    gathered = frame = None
    if world > 1 and rank == 0:
        gathered = torch.empty((world, plan.rows_pad, width, 3), dtype=torch.float32, device=dev)
        frame = torch.empty((height, width, 3), dtype=torch.float32, device=dev)

    def step():
        ctx.renderFrameDevice(cam, opts, local.data_ptr(), stream.cuda_stream)
        if world > 1:
            if rank == 0:
                dist.gather(local, list(gathered.unbind(0)), dst=0)
                ctx.deinterleaveStrips(gathered.data_ptr(), frame.data_ptr(), width, height, plan.strip_height, world,
                                       stream.cuda_stream)
            else:
                dist.gather(local, None, dst=0)

    # rays per frame (deterministic): one untimed counting pass
    copts = scene.renderOpts(taps=taps, strip_height=plan.strip_height, strip_rank=rank, strip_world=world, count_rays=1)
    ctx.renderFrameDevice(cam, copts, local.data_ptr(), stream.cuda_stream)
    primary, shadow = ctx.rayStats()
    rays = torch.tensor([primary, shadow], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(rays)
    primary, shadow = int(rays[0].item()), int(rays[1].item())
    rays_per_frame = primary + shadow

    for _ in range(args.warmup):
        step()

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    # kernel-only duration: events on the launch stream around each render launch
    ev0 = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps)]
    ev1 = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps)]

    barrier()
    t0 = time.perf_counter()
    for i in range(args.steps):
        ev0[i].record(stream)
        ctx.renderFrameDevice(cam, opts, local.data_ptr(), stream.cuda_stream)
        ev1[i].record(stream)
        if world > 1:
            if rank == 0:
                dist.gather(local, list(gathered.unbind(0)), dst=0)
                ctx.deinterleaveStrips(gathered.data_ptr(), frame.data_ptr(), width, height, plan.strip_height, world,
                                       stream.cuda_stream)
            else:
                dist.gather(local, None, dst=0)
    barrier()
    elapsed = time.perf_counter() - t0

    el = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(el, op=dist.ReduceOp.MAX)
    elapsed = float(el.item())
    kernel_ms = statistics.mean(a.elapsed_time(b) for a, b in zip(ev0, ev1))

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        value = rays_per_frame * args.steps / elapsed / 1e6
        # algorithmic HBM bytes of one launch: 12 B per pixel written + every bitmap texel once
        d = scene.desc.contents
        tex_bytes = int(d.n_texels) * 12
        alg_bytes = my_rows * width * 12 + tex_bytes
        achieved = alg_bytes / (kernel_ms * 1e-3) / 1e9
        traffic = None
        prof = os.path.join(ROOT, "profiles", "traffic_%s.json" % args.workload)
        if world == 1 and os.path.exists(prof):
            try:
                traffic = json.load(open(prof)).get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        out = {
            "metric": "Mray/s (primary + shadow rays actually cast per second; ms/frame in ms_per_step)",
            "value": value,
            "unit": "Mray/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": ms_per_step,
            "higher_is_better": True,
            "scaling": args.scaling,
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic: the reference's own scene file and textures (tests/golden/scenes), no camera motion",
            "config": {
                "workload": "%s %dx%d, %d tap(s)/pixel, dof off%s" % (
                    scene_file, width, height, taps,
                    "" if world == 1 else "; %d interleaved %d-row strip sets + RCCL gather to rank 0" % (world, plan.strip_height)),
                "name": args.workload,
                "primary_rays_per_frame": primary,
                "shadow_rays_per_frame": shadow,
                "Msample_per_s": primary * args.steps / elapsed / 1e6,
                "frames_per_s": args.steps / elapsed,
                "kernel_ms_rank0": kernel_ms,
            },
            "roofline": {
                "bound": "hbm",
                "achieved": achieved,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS,
                "traffic": traffic,
                "algorithmic_bytes_per_launch": alg_bytes,
                "kernel_ms": kernel_ms,
                "note": "the path is fp64-VALU bound, not HBM bound (DESIGN.md): 12 B/pixel is all it must move",
            },
        }
        if world == 1 and not args.no_cpu_baseline:
            full = scene.renderOpts(taps=taps)
            out["cpu_baseline"] = cpu_baseline(scene, cam, full, rays_per_frame)
        print(json.dumps(out), flush=True)

    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
