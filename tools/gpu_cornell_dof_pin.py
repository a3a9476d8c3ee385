"""Power of the pin against result/cornell_dof.png (tests/golden/reference_cornell_dof_64.npy): the GPU renders the Cornell box
through the thin-lens camera with the identified parameters and with deliberately wrong ones, 4096 spp each.
usage (GPU box): python tools/gpu_cornell_dof_pin.py [spp]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import oracle_lib as ol, test_oracle_golden as tg
from gpu_pathtracer_amd import api

spp = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
W = H = 512
print(f"# GPU film of the Cornell box (configs 1 / 2 geometry) through the thin lens against result/cornell_dof.png: 512x512, {spp} spp,")
print("# filmic + 8-bit, 64x63 blocks (the picture's last 6 pixel columns are black); the test requires mean diff < 0.002, block mean < 0.003")
for name, depth, ap, focal, light in (("identified: depth 8, focal 7.0, aperture 0.5", 8, 0.5, 7.0, 1.0),
                                      ("depth 17", 17, 0.5, 7.0, 1.0), ("depth 7", 7, 0.5, 7.0, 1.0), ("depth 5", 5, 0.5, 7.0, 1.0),
                                      ("depth 3", 3, 0.5, 7.0, 1.0),
                                      ("pinhole (aperture 0)", 8, 0.0, 7.0, 1.0), ("aperture 0.4", 8, 0.4, 7.0, 1.0), ("aperture 0.6", 8, 0.6, 7.0, 1.0),
                                      ("focal distance 6.5", 8, 0.5, 6.5, 1.0), ("focal distance 7.5", 8, 0.5, 7.5, 1.0),
                                      ("light radiance x 0.9", 8, 0.5, 7.0, 0.9), ("light radiance x 0.97", 8, 0.5, 7.0, 0.97)):
    scene, meta = ol.load_cornell(depth)
    if light != 1.0:
        scene.lights["radiance"]["x"] *= np.float32(light); scene.lights["radiance"]["y"] *= np.float32(light); scene.lights["radiance"]["z"] *= np.float32(light)
    with api.Renderer(scene.desc, W, H, meta["epsilon"]) as r:
        r.render(tg.dof_camera(meta, ap, focal), 1, spp, reset=True)
        m, bm, bx = tg.dof_compare(r.read_accum(), spp)
    print(f"{name:48s} mean diff {m:.4f}  block mean {bm:.4f}  block max {bx:.3f}", flush=True)
