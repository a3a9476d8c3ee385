"""Traversal alone: Mrays/s of the two loops (reference order / 4-wide) through gpt_debug_trace, one ray per lane per
round, on the Cornell box (LDS and global memory) and the config-5 stand-in, for incoherent rays (random origins and directions) and for
coherent ones (a pinhole camera's primary rays).  usage (GPU box): python tools/gpu_trace_rate.py [million rays]"""
import sys, tempfile
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np
import scenes, oracle_lib as ol
from gpu_pathtracer_amd import api
n = int(float(sys.argv[1]) * 1e6) if len(sys.argv) > 1 else 4_000_000
rng = np.random.default_rng(1)
inc = np.zeros((n, 8), np.float32)
inc[:, 0:3] = np.float32([-0.95, 0.05, -0.95]) + np.float32([1.9, 1.9, 1.9]) * rng.random((n, 3)).astype(np.float32)
d = rng.standard_normal((n, 3)).astype(np.float32)
inc[:, 3:6] = d / np.sqrt((d * d).sum(-1, keepdims=True), dtype=np.float32)
inc[:, 6] = np.inf
side = int(np.sqrt(n))
ys, xs = np.mgrid[0:side, 0:side].astype(np.float32)
coh = np.zeros((side * side, 8), np.float32)
coh[:, 0:3] = np.float32([0, 1.0, 6.8])
dd = np.stack([(xs.ravel() / side - 0.5) * 0.34, (ys.ravel() / side - 0.5) * 0.34, -np.ones(side * side, np.float32)], -1).astype(np.float32)
coh[:, 3:6] = dd / np.sqrt((dd * dd).sum(-1, keepdims=True), dtype=np.float32)
coh[:, 6] = np.inf
scene_c, _ = ol.load_cornell(4)
ls = api.LoadedScene(scenes.write_standin_scene(tempfile.mkdtemp(), "c5"))
for name, scene, lds in (("cornell, LDS", scene_c, 1), ("cornell, global memory", scene_c, 0), ("config-5 stand-in", ls, 0)):
    with api.Renderer(scene.desc, 64, 64, 0.001) as r:
        r.set_option("lds_scene", lds)
        for order, oname in ((0, "reference"), (2, "wide")):
            if lds and order:
                continue
            r.set_traversal_order(order)
            for rname, rays in (("incoherent", inc), ("coherent", coh)):
                r.trace_rays(rays[:65536])
                prim, tb = r.trace_rays(rays)
                us = r.get_option("last_trace_us")
                print(f"{name:24s} {oname:10s} {rname:10s} {len(rays) / us:8.1f} Mrays/s  ({len(rays)} closest-hit rays, {100 * (prim >= 0).mean():.0f} % hit, {us / 1e3:.1f} ms)", flush=True)
