"""Config-5 stand-in (253k triangles, 16 bounces) timing + work counters at 1080p and 4K."""
import sys, os, time
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np
import scenes, oracle_lib as ol
from gpu_pathtracer_amd import api
scene, meta = scenes.stress_scene(1.0, max_depth=16)
print("stress scene:", len(scene.prims), "triangles,", len(scene.nodes), "nodes", flush=True)
NEAR = len(sys.argv) > 1 and sys.argv[1] == "wide"
if NEAR: print("traversal order: 4-wide walk", flush=True)
for (W, H, spp) in ((1920, 1080, 16), (3840, 2160, 8)):
    cam = ol.cornell_camera(meta, W, H)
    with api.Renderer(scene.desc, W, H, 0.001) as r:
        r.set_traversal_order("wide" if NEAR else "reference")
        r.render(cam, 1, 2, reset=True); r.synchronize()
        best = 1e9
        for rep in range(2):
            r.kernel_time_reset(); r.render(cam, 1, spp, reset=True); r.synchronize()
            best = min(best, r.kernel_time()[1])
        r.enable_counters(True); r.render(cam, 1, 2, reset=True); c = r.read_probe_counters()
    s = c["samples"]
    balg = (40.0 * c["node_visits"] + 176.0 * c["prim_tests"] + 72.0 * c["bounce_iters"] + 192.0 * c["shadow_rays"]) / s + 60
    print(f"{W}x{H} {spp} spp: {best:.1f} ms -> {W*H*spp/best/1e3:.1f} Msamples/s | per sample: nodes {c['node_visits']/s:.1f} prims {c['prim_tests']/s:.1f} "
          f"bounces {c['bounce_iters']/s:.2f} shadow {c['shadow_rays']/s:.2f} closest {c['closest_rays']/s:.2f} | B_alg {balg/1e3:.1f} KB -> {W*H*spp/best*1e3*balg/1e12:.1f} TB/s algorithmic | "
          f"lane util node {c['node_visits']/(64.0*c['w_node']):.2f} tri {c['prim_tests']/(64.0*c['w_prim']):.2f}", flush=True)
