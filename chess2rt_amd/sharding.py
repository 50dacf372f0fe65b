"""Framebuffer sharding across the GPUs of one node.

Pixels are independent, so the frame shards with no data-path collective
until the very end: rank r renders the interleaved row strips r, r+G, r+2G, ...
(contiguous bands would be badly unbalanced: the top of these scenes is sky)
into a compact buffer, and ONE gather over RCCL/xGMI brings the strips to
rank 0, where a copy kernel de-interleaves them into the frame.  One process
per GPU, torch.distributed for the plumbing.
"""
from collections import namedtuple

StripPlan = namedtuple("StripPlan", "strip_height world rows_pad")


def local_rows(height, strip_height, rank, world):
    """Rows rank `rank` owns (mirror of c2rt_local_rows)."""
    if world <= 1:
        return height
    n_strips = (height + strip_height - 1) // strip_height
    rows = 0
    for s in range(rank, n_strips, world):
        rows += min(strip_height, height - s * strip_height)
    return rows


def plan_strips(height, world, strip_height=8):
    """Strip height (a multiple of the 8-row wavefront tile) and the padded
    per-rank row count (rank 0 always owns the most rows)."""
    if strip_height <= 0 or strip_height % 8:
        raise ValueError("strip_height must be a positive multiple of 8 (tile height)")
    return StripPlan(strip_height, world, local_rows(height, strip_height, 0, world))


def deinterleave_strips_torch(gathered, height, strip_height, world):
    """CPU-tensor restatement of c2rt_deinterleave_strips for gloo runs:
    gathered (world, rows_pad, W, 3) -> frame (height, W, 3)."""
    import torch

    frame = torch.empty((height,) + tuple(gathered.shape[2:]), dtype=gathered.dtype, device=gathered.device)
    n_strips = (height + strip_height - 1) // strip_height
    for s in range(n_strips):
        y0 = s * strip_height
        h = min(strip_height, height - y0)
        lr = (s // world) * strip_height
        frame[y0:y0 + h] = gathered[s % world, lr:lr + h]
    return frame


def render_frame_sharded(render_strips, width, height, plan, rank, group=None, deinterleave=None):
    """Renders this rank's strips and gathers the frame on rank 0.

    render_strips(local) fills the (rows_pad, width, 3) float32 tensor `local`
    with this rank's strips (rows beyond its own count are padding).
    deinterleave(gathered, frame) de-interleaves on the device (rank 0 only);
    None uses the torch restatement (CPU tensors).
    Returns (frame or None, local).
    """
    import torch
    import torch.distributed as dist

    world = plan.world
    local = render_strips()
    assert local.shape == (plan.rows_pad, width, 3) and local.dtype == torch.float32
    if world <= 1:
        return local[:height], local
    gathered = None
    if rank == 0:
        gathered = torch.empty((world,) + tuple(local.shape), dtype=local.dtype, device=local.device)
        dist.gather(local, list(gathered.unbind(0)), dst=0, group=group)
    else:
        dist.gather(local, None, dst=0, group=group)
        return None, local
    if deinterleave is not None:
        frame = torch.empty((height, width, 3), dtype=local.dtype, device=local.device)
        deinterleave(gathered, frame)
    else:
        frame = deinterleave_strips_torch(gathered, height, plan.strip_height, world)
    return frame, local


def exchange_strips_p2p(local, frame, height, plan, rank, group=None):
    """The gather-free exchange: a strip of full rows is contiguous in the frame, so rank 0 copies its
    own strips into place and posts one receive per remote strip straight into `frame[y0:y0+n]`; every
    other rank posts one send per own strip (batched isend/irecv — RCCL send/recv over xGMI with the
    `nccl` backend).  No rank-major gather buffer and no de-interleave pass.  `local` is this rank's
    compact (rows_pad, ...) strip buffer, `frame` the (height, ...) result on rank 0 (None elsewhere).
    Blocks until the exchange is complete."""
    import torch.distributed as dist

    world, sh = plan.world, plan.strip_height
    ops = []
    for s in range((height + sh - 1) // sh):
        y0, n, owner, lr = s * sh, min(sh, height - s * sh), s % world, (s // world) * sh
        if rank == 0:
            if owner == 0:
                frame[y0:y0 + n].copy_(local[lr:lr + n])
            else:
                ops.append(dist.P2POp(dist.irecv, frame[y0:y0 + n], owner, group))
        elif owner == rank:
            ops.append(dist.P2POp(dist.isend, local[lr:lr + n], 0, group))
    for w in (dist.batch_isend_irecv(ops) if ops else []):
        w.wait()
    return frame
