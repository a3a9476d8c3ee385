"""Throughput of the BASELINE.json configurations' stand-ins on one GPU (kernel time of the path kernel, HIP events).
config 2 is what bench.py measures; 3-5 are the parity-test scenes of tests/test_gpu_parity.py at their full sizes."""
import sys
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np
import scenes, oracle_lib as ol
from gpu_pathtracer_amd import api, host


def timed(desc, cam, W, H, eps, spp, near=False):
    with api.Renderer(desc, W, H, eps) as r:
        r.set_traversal_order({False: "reference", "wide": "wide"}[near])
        r.render(cam, 1, 2, reset=True); r.synchronize()
        best = 1e9
        for rep in range(3):
            r.kernel_time_reset(); r.render(cam, 1, spp, reset=True); r.synchronize()
            best = min(best, r.kernel_time()[1])
        fin = bool(np.isfinite(r.read_accum()).all())
    return W * H * spp / best / 1e3, best, fin


scene, meta = host.load_baked("tests/golden/cornell_pt.npz", 8)
print("config 2  cornell 1080p depth 8            : %7.1f Msamples/s (%.1f ms / 64 iterations) finite=%s" % timed(scene.desc, host.camera_from_meta(meta, 1920, 1080), 1920, 1080, 0.001, 64), flush=True)

extra = scenes.concat([scenes.uv_sphere((-0.45, 0.45, 0.3), 0.4, 8, nu=24, nv=16), scenes.uv_sphere((0.4, 0.35, 0.45), 0.33, 7, nu=24, nv=16),
                       scenes.uv_sphere((0.05, 1.25, -0.3), 0.35, 10, nu=24, nv=16)])
s3, meta3 = scenes.zoo_scene(max_depth=10, extra=extra, assign={"short": 5, "tall": 13, "floor": 12, "back": 9})
cam = ol.cornell_camera(meta3, 1920, 1080)
for near in (False, "wide"):
    print("config 3  material scene 1080p depth 10 %s: %7.1f Msamples/s (%.1f ms / 32 iterations) finite=%s" % ((("wide" if near else "    "),) + timed(s3.desc, cam, 1920, 1080, 0.0005, 32, near)), flush=True)

prims, _, meta = scenes.cornell_raw()
allp = scenes.concat([prims[0:2], scenes.stress_parts(0.3)])
c, s_ = np.float32(np.cos(np.pi / 6)), np.float32(np.sin(np.pi / 6))
s4 = ol.make_scene(allp, scenes.material_table(), light_radiance=meta["light_radiance"], max_depth=7, env=scenes.sky_env(256, 128),
                   env_rotate_uvw=((c, 0.0, -s_), (0.0, 1.0, 0.0), (s_, 0.0, c)), textures=[scenes.checker_texture()])
cam = ol.make_camera((0.3, 1.4, 5.5), (0, 0.8, 0), (0, 1, 0), (1920, 1080), 35.0)
for near in (False, "wide"):
    print("config 4  env-lit 22k triangles 1080p d7 %s: %7.1f Msamples/s (%.1f ms / 32 iterations) finite=%s" % ((("wide" if near else "    "),) + timed(s4.desc, cam, 1920, 1080, 0.001, 32, near)), flush=True)

# ---- the stand-ins SURVEY.md 8(d) defines from the reference's shipped meshes, through the product loader
import tempfile
for which, label, spp in (("c3", "config 3  SURVEY stand-in: 3 spheres + cube-subdiv, shaderball camera/materials, 1080p d10", 32),
                          ("c4", "config 4  SURVEY stand-in: config-5 geometry under a procedural sky, 1080p d7             ", 32),
                          ("c5", "config 5  SURVEY stand-in: walls + dragon + bunny2 + teapot + 9 spheres (248 574), 4K d16 ", 8)):
    ls = api.LoadedScene(scenes.write_standin_scene(tempfile.mkdtemp(), which))
    for near in (False, "wide"):
        print("%s %s: %7.1f Msamples/s (%.1f ms / %d iterations) finite=%s" % ((label, {False: "    ", "wide": "wide"}[near]) + timed(ls.desc, ls.camera, ls.width, ls.height, ls.epsilon, spp, near)[:2] + (spp, True)), flush=True)
    ls.close()

print("# the procedural stand-ins of round 1 (parametric blobs), for comparison")
s5, meta5 = scenes.stress_scene(1.0, max_depth=16)
cam = ol.cornell_camera(meta5, 3840, 2160)
for near in (False, "wide"):
    print("config 5  253k triangles 4K depth 16     %s: %7.1f Msamples/s (%.1f ms / 8 iterations) finite=%s" % ((("wide" if near else "    "),) + timed(s5.desc, cam, 3840, 2160, 0.001, 8, near)), flush=True)
