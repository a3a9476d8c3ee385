"""Volpath timing on the GPU: the three-rays-per-bounce kernel (homogeneous fog in the Cornell box) and the
one-ray-at-a-time kernel (a density grid in a material-less box, the shape of the reference's shipped
scenes/cornell_box/scene.json: 512x512, 17 bounces, 100x100x40 grid, ratio tracking, iterMax 2000)."""
import sys, os
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np
import scenes, oracle_lib as ol
import test_gpu_parity as tg
from gpu_pathtracer_amd import api, scene_types as st


def timed(name, scene, cam, W, H, spp, opt=None):
    with api.Renderer(scene.desc, W, H, 0.001) as r:
        if opt: r.set_option(opt, 1)
        r.render(cam, 1, 2, reset=True); r.synchronize()
        best = 1e9
        for rep in range(2):
            r.kernel_time_reset(); r.render(cam, 1, spp, reset=True); r.synchronize()
            best = min(best, r.kernel_time()[1])
        acc = r.read_accum()
    print(f"{name}: {W}x{H} {spp} spp: {best:.1f} ms -> {W*H*spp/best/1e3:.1f} Msamples/s, mean radiance {acc.reshape(-1,3).mean(0)/spp}", flush=True)


only_shipped = len(sys.argv) > 1 and sys.argv[1] == "shipped"       # (counter passes: one kernel, one workload)
fog = st.make_medium((0.0014, 0.0025, 0.0142), (0.70, 1.22, 1.90), 0.0, 0.3)
scene, meta = ol.load_cornell(8)
scene.set_mediums([fog])
scene.desc.set_integrator("vpt", 8)
W, H = 1920, 1080
cam = ol.cornell_camera(meta, W, H)
cam.medium = 0
if not only_shipped:
    timed("fog cornell, three-ray kernel", scene, cam, W, H, 64)
    timed("fog cornell, one-ray kernel (forced)", scene, cam, W, H, 64, opt="vpt_walk_kernel")
    scene.desc.set_integrator("pt", 8)
    timed("same scene, Path (media ignored)", scene, cam, W, H, 64)

scene, cam, W, H, spp = tg.walk_case("shipped_like")
W = H = 512
cam = ol.make_camera((0, 1.0, 6.8), (0, 1.0, 0), (0, 1, 0), (W, H), 19.5, 0.0, 7.0)
cam.medium = -1
timed("shipped-like density grid, one-ray kernel", scene, cam, W, H, 64)
