"""The N>1 path on CPU: world_size-2 (and 3) gloo groups run the product's
strip plan + gather + de-interleave plumbing (chess2rt_amd/sharding.py) with
the ORACLE standing in as the strip producer (tests only; the product's
producer is the HIP kernel)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import chess2rt_amd as c2
from chess2rt_amd.sharding import deinterleave_strips_torch, exchange_strips_p2p, local_rows, plan_strips, render_frame_sharded


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, name, out_path):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import oracle_lib as orc
    from golden_configs import load_config

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    scene, cam, full_opts = load_config(name)
    W, H = full_opts.width, full_opts.height
    plan = plan_strips(H, world, 8)
    _, _, opts = load_config(name, strip_height=plan.strip_height, strip_rank=rank, strip_world=world)

    def producer():
        mine = orc.render_frame(scene.desc, cam, opts, 2)
        assert mine.shape[0] == local_rows(H, plan.strip_height, rank, world)
        local = torch.zeros((plan.rows_pad, W, 3), dtype=torch.float32)
        local[: mine.shape[0]] = torch.from_numpy(mine)
        return local

    frame, local = render_frame_sharded(producer, W, H, plan, rank)
    # the gather-free exchange (one receive per remote strip straight into the frame) assembles the same frame
    direct = exchange_strips_p2p(local, torch.full((H, W, 3), -1.0) if rank == 0 else None, H, plan, rank)
    if rank == 0:
        assert torch.equal(direct, frame)
        np.save(out_path, frame.numpy())
    else:
        assert frame is None
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,name", [(2, "lecture5_333x217_t4"), (3, "csg_stress_320x240_t1")])
def test_strips_gather_to_the_full_frame(world, name, tmp_path):
    import oracle_lib as orc
    from golden_configs import load_config

    out = str(tmp_path / "frame.npy")
    mp.spawn(_worker, args=(world, _free_port(), name, out), nprocs=world, join=True)
    scene, cam, opts = load_config(name)
    full = orc.render_frame(scene.desc, cam, opts, 0)
    assert np.array_equal(np.load(out), full)


def test_plan_and_deinterleave_roundtrip():
    for (h, world, sh) in [(217, 3, 8), (2160, 8, 8), (4320, 8, 16), (100, 4, 8), (8, 2, 8), (5, 4, 8)]:
        plan = plan_strips(h, world, sh)
        assert plan.rows_pad == local_rows(h, sh, 0, world) >= max(local_rows(h, sh, r, world) for r in range(world))
        assert sum(local_rows(h, sh, r, world) for r in range(world)) == h
        frame = torch.arange(h * 2 * 3, dtype=torch.float32).reshape(h, 2, 3)
        gathered = torch.full((world, plan.rows_pad, 2, 3), -1.0)
        for r in range(world):
            rows = [y for y in range(h) if (y // sh) % world == r]
            gathered[r, : len(rows)] = frame[rows]
        assert torch.equal(deinterleave_strips_torch(gathered, h, sh, world), frame)
    with pytest.raises(ValueError):
        plan_strips(100, 2, 5)
