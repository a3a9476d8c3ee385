"""Multi-GPU sharding of the path: pixel tiles across ranks + ONE sum-reduce of the float3 accumulator.

The path shards embarrassingly: every pixel-sample is independent (seed = hash(pixel) + hash(iter),
reference src/pathtracer.cu:888) and reads only read-only scene data.  Each rank holds the whole scene,
renders the 8x8-pixel tiles t with t % world == rank into a zero-initialised full-frame accumulator, and the
frame is assembled by one reduce (sum, fp32, W*H*3) to rank 0 — RCCL over xGMI when the backend is "nccl".
Supports are disjoint, so the sum adds zeros only and the result is bit-identical to a 1-GPU render.
Tonemapping (Output) runs on the root after the reduce.
"""
import os

import numpy as np


def tile_owner_mask(width, height, rank, world):
    """Boolean (H, W) mask of the pixels rank owns: tile index (x/8) + (y/8)*tiles_x, t % world == rank,
    over the reference's launch geometry stride = 32*(W/32), rows = 4*(H/4)."""
    stride, rows = 32 * (width // 32), 4 * (height // 4)
    tiles_x = (stride + 7) // 8
    y, x = np.mgrid[0:height, 0:width]
    tile = (x // 8) + (y // 8) * tiles_x
    return (tile % world == rank) & (x < stride) & (y < rows)


def film_owner_mask(width, height, rank, world):
    """Boolean mask over the W*H pixel slots of the FILM BUFFER (accumulator / last-sample planes): slot
    p = x + y * stride with stride = 32*(W/32) — the reference's pixel index (src/pathtracer.cu:881-883), which is
    plain row-major only when W is a multiple of 32 (include/gpt.h "Film state").  Use this one to index the buffer
    gpt_read_accum returns; tile_owner_mask is the same set as (y, x) coordinates."""
    stride, rows = 32 * (width // 32), 4 * (height // 4)
    own = tile_owner_mask(width, height, rank, world)[:rows, :stride]
    y, x = np.nonzero(own)
    flat = np.zeros(width * height, dtype=bool)
    flat[x + y * stride] = True
    return flat


def init_process_group(backend=None):
    import torch
    import torch.distributed as dist
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world == 1:
        return None, rank, world
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    backend = backend or ("nccl" if torch.cuda.is_available() else "gloo")
    if not dist.is_initialized():
        dist.init_process_group(backend, rank=rank, world_size=world)
    return dist, rank, world


def reduce_framebuffer(acc, dist, root=0):
    """acc: torch tensor (W*H*3,) float32 — this rank's accumulator (zeros outside its tiles).
    In place; after the call the root holds the whole frame."""
    if dist is not None:
        dist.reduce(acc, dst=root, op=dist.ReduceOp.SUM)
    return acc
