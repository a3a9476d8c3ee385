"""The N>1 path on CPU: two gloo ranks, each rendering its tile shard (with the oracle standing in for the
GPU renderer — test infrastructure), one sum-reduce of the accumulator; the root's frame must be bit-identical
to a single-rank render."""
import os
import socket
import subprocess
import sys

import numpy as np

import oracle_lib as ol
from gpu_pathtracer_amd import distributed as gd

WORKER = r'''
import os, sys
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, os.path.join(sys.argv[1], "tests"))
import numpy as np, torch
import oracle_lib as ol
from gpu_pathtracer_amd import distributed as gd
dist, rank, world = gd.init_process_group("gloo")
W, H, spp = 96, 64, 3
scene, meta = ol.load_cornell(5)
cam = ol.cornell_camera(meta, W, H)
acc, _ = ol.render(scene, cam, W, H, 0.001, 1, spp, rank=rank, n_ranks=world, threads=2)
own = gd.film_owner_mask(W, H, rank, world)
assert not acc.reshape(-1, 3)[~own].any(), "a rank wrote outside its tiles"
t = torch.from_numpy(acc.copy())
out = gd.reduce_framebuffer(t, dist, root=0)
assert t.numpy().tobytes() == acc.tobytes(), "the send buffer keeps this rank's tiles only"
if rank == 0:
    np.save(sys.argv[2], out.numpy())
dist.barrier()
dist.destroy_process_group()
'''


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_two_rank_gloo_reduce_equals_single_rank(tmp_path):
    out = str(tmp_path / "frame.npy")
    port = free_port()
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   OMP_NUM_THREADS="2")
        procs.append(subprocess.Popen([sys.executable, "-c", WORKER, ol.ROOT, out], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    for p in procs:
        o, _ = p.communicate(timeout=300)
        assert p.returncode == 0, o.decode()[-2000:]
    W, H, spp = 96, 64, 3
    scene, meta = ol.load_cornell(5)
    cam = ol.cornell_camera(meta, W, H)
    full, _ = ol.render(scene, cam, W, H, 0.001, 1, spp)
    assert np.load(out).tobytes() == full.tobytes()


def test_film_mask_follows_the_buffer_layout_for_widths_that_are_not_multiples_of_32():
    """pixel slot = x + y * stride, stride = 32*(W/32) (the reference's index, pathtracer.cu:881-883): the oracle's film for a
    100-pixel-wide frame has its samples exactly on the slots the flat mask names."""
    W, H = 100, 70
    scene, meta = ol.load_cornell(3)
    cam = ol.cornell_camera(meta, W, H)
    lit = np.zeros(W * H, dtype=bool)
    for rank in range(2):
        acc, _ = ol.render(scene, cam, W, H, 0.001, 1, 2, rank=rank, n_ranks=2, threads=2)
        own = gd.film_owner_mask(W, H, rank, 2)
        px = acc.reshape(-1, 3)
        assert not px[~own].any(), "a rank wrote outside the slots of its tiles"
        lit |= px.any(axis=1)
    assert lit.sum() > 0.5 * 96 * 68 and not lit[96 * 68:].any()


def test_tile_masks_partition_the_launch_region():
    for (W, H) in ((1920, 1080), (100, 70), (64, 64)):
        for world in (1, 2, 3, 8):
            masks = [gd.tile_owner_mask(W, H, r, world) for r in range(world)]
            total = np.sum(masks, axis=0)
            stride, rows = 32 * (W // 32), 4 * (H // 4)
            assert (total[:rows, :stride] == 1).all() and total[rows:].sum() == 0 and total[:, stride:].sum() == 0
            if (W, H) == (1920, 1080) and world == 8:
                counts = [int(m.sum()) for m in masks]
                assert max(counts) - min(counts) <= 64        # 32400 tiles / 8 ranks: balanced to one tile
