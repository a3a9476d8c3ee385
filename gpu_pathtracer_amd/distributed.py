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


class FilmReducer:
    """The one collective of the path.  native=True: the library's own RCCL reduce (gpt_comm_init / gpt_reduce_film) on the
    renderer's stream - torch.distributed only carries the 128-byte RCCL id to the ranks.  native=False: the same N-rank
    code path (tile ownership, a receive buffer that is not the accumulator, Output from the reduced frame) over a
    torch.distributed backend that can run every rank on ONE GPU (gloo) - a functional check where RCCL, which refuses
    duplicate devices, cannot run."""

    def __init__(self, renderer, dist, rank, world, native=True):
        from . import api
        self.r, self.dist, self.rank, self.world, self.native = renderer, dist, rank, world, native
        self.kind = "rccl ncclReduce issued by libgpt.so (gpt_reduce_film)" if native else f"torch.distributed {dist.get_backend()} reduce (all ranks share one GPU)"
        self.native_error = None
        if native:
            # Every rank must end up on the same path, and ncclCommInitRank is itself a collective: a rank that cannot even load
            # RCCL must not leave the others blocked in it.  So the agreement has two steps.
            # Step 1 (no RCCL collective yet): can every rank load RCCL, and could rank 0 create the id?
            import torch
            ok, uid = 1, None
            try:
                uid = api.comm_unique_id()        # loads librccl; the id itself is used from rank 0 only
            except Exception as e:
                ok, self.native_error = 0, str(e)
            flag = torch.tensor([ok], dtype=torch.int32, device="cuda")
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            if int(flag.item()) == 0:
                self.native = native = False
                self.kind = (f"torch.distributed {dist.get_backend()} reduce (RCCL could not be loaded on every rank: "
                             f"{self.native_error or 'another rank failed'})")
            else:
                # Step 2: everybody enters ncclCommInitRank with rank 0's id; its outcome is agreed on as well
                box = [uid if rank == 0 else None]
                dist.broadcast_object_list(box, src=0)
                try:
                    renderer.comm_init(rank, world, box[0])
                except Exception as e:
                    ok, self.native_error = 0, str(e)
                flag = torch.tensor([ok], dtype=torch.int32, device="cuda")
                dist.all_reduce(flag, op=dist.ReduceOp.MIN)
                if int(flag.item()) == 0:
                    renderer.comm_destroy()
                    self.native = native = False
                    self.kind = f"torch.distributed {dist.get_backend()} reduce (the library's RCCL path failed: {self.native_error or 'on another rank'})"
        if not native:
            import torch
            renderer.set_tile_owner(rank, world)
            n = renderer.width * renderer.height * 3
            self.acc = torch.zeros(n, dtype=torch.float32, device="cuda")       # the film itself: bound as the accumulator
            self.col = torch.zeros(n, dtype=torch.float32, device="cuda")
            self.recv = torch.zeros(n, dtype=torch.float32, device="cuda")      # the root's reduced frame
            renderer.bind_film(self.acc.data_ptr(), self.col.data_ptr())

    def reduce(self, root=0):
        if self.native:
            self.r.reduce_film(root)
            return
        import torch
        self.r.synchronize()                    # torch's streams do not know the renderer's
        self.recv.copy_(self.acc)
        self.dist.reduce(self.recv, dst=root, op=self.dist.ReduceOp.SUM)
        torch.cuda.synchronize()

    def reduced_ptr(self):
        return self.r.reduced_ptr() if self.native else self.recv.data_ptr()

    def tonemap_reduced(self, iteration, filmic, out_dev):
        self.r.tonemap_from(self.reduced_ptr(), iteration, filmic, out_dev)

    def read_reduced(self):
        if self.native:
            return self.r.read_reduced()
        self.r.synchronize()
        return self.recv.cpu().numpy()


def reduce_framebuffer(acc, dist, root=0):
    """acc: torch tensor (W*H*3,) float32 - this rank's accumulator (zeros outside its tiles).  Returns the reduced frame on
    the root (a NEW tensor: acc itself keeps only this rank's tiles, so a later progressive render + reduce stays right)."""
    if dist is None:
        return acc
    out = acc.clone()
    dist.reduce(out, dst=root, op=dist.ReduceOp.SUM)
    return out
