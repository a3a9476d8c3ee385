"""Kernel time of one 1/8 tile shard (and of the full frame) against iterations per launch: fixed cost + slope."""
import sys, os
sys.path.insert(0, '.')
from gpu_pathtracer_amd import api, host
W, H, D = 1920, 1080, 8
scene, meta = host.load_baked("tests/golden/cornell_pt.npz", D)
cam = host.camera_from_meta(meta, W, H)
for label, rank, n in (("shard0of8", 0, 8), ("full", 0, 1)):
    r = api.Renderer(scene.desc, W, H, 0.001)
    r.set_tile_owner(rank, n)
    r.render(cam, 1, 8, reset=True); r.synchronize()
    row = []
    for spp in (4, 8, 16, 32, 64):
        best = 1e9
        for rep in range(3):
            r.kernel_time_reset(); r.render(cam, 1, spp, reset=True); r.synchronize()
            best = min(best, r.kernel_time()[1])
        row.append((spp, best))
    r.close()
    (s0, t0), (s1, t1) = row[-2], row[-1]
    slope = (t1 - t0) / (s1 - s0)
    print(label, " ".join(f"{s}spp:{t:.3f}ms" for s, t in row), f"| slope {slope*1e3:.1f} us/iteration, intercept {t1 - slope*s1:.3f} ms", flush=True)
