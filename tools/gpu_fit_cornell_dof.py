"""result/cornell_dof.png of the reference shows the Cornell box with its two boxes (the scene of BASELINE configs 1-2: every
mesh ships) through the thin-lens camera (camera.h:62-78).  The scene file it was rendered from is not in the repository, so
the lens parameters are unknown; the shipped scene.json carries "focalDistance": 7.0 beside "apertureRadius": 0.0.  This
script renders the scene on the GPU over a grid of (maxDepth, apertureRadius, focalDistance) and prints the distance to the
published picture, to find the setting the author used.
usage (GPU box): python tools/gpu_fit_cornell_dof.py [spp [depths [focal distances [aperture radii]]]]   (comma-separated lists)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import oracle_lib as ol, refimg
from gpu_pathtracer_amd import api

spp = int(sys.argv[1]) if len(sys.argv) > 1 else 512
want = refimg.load("reference_cornell_dof_64.npy")
W = H = 512
rows = []
DEPTHS = [int(x) for x in sys.argv[2].split(',')] if len(sys.argv) > 2 else (5, 8, 17)
FOCALS = [float(x) for x in sys.argv[3].split(',')] if len(sys.argv) > 3 else (6.0, 6.5, 7.0, 7.5, 8.0)
APERTURES = [float(x) for x in sys.argv[4].split(',')] if len(sys.argv) > 4 else (0.0, 0.02, 0.05, 0.1, 0.15)
for depth in DEPTHS:
    scene, meta = ol.load_cornell(depth)
    with api.Renderer(scene.desc, W, H, meta["epsilon"]) as r:
        for focal in FOCALS:
            for ap in APERTURES:
                c = meta["camera"]
                cam = ol.make_camera(c["position"], c["lookat"], c["up"], (W, H), c["fov"], ap, focal, c["distance"], c["filmic"])
                r.render(cam, 1, spp, reset=True)
                m, bm, bx, means = refimg.compare(r.read_accum(), spp, W, H, want)
                rows.append((bm, depth, focal, ap, m, bx, means))
                print(f"depth {depth:2d} focal {focal:4.1f} aperture {ap:5.3f}: mean diff {m:.4f} block mean {bm:.4f} block max {bx:.3f} means {means.round(4)}", flush=True)
rows.sort(key=lambda t: t[0])
print("# best by block mean:")
for t in rows[:8]:
    print(f"# depth {t[1]} focal {t[2]} aperture {t[3]}: block mean {t[0]:.4f} mean diff {t[4]:.4f} block max {t[5]:.3f}")
print("# reference frame means", want.mean(axis=(0, 1)).round(4))
