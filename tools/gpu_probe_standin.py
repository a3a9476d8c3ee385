"""Work counters of the path kernel's counting build on a SURVEY stand-in (c3 / c4 / c5): rays, node visits and triangle tests
per sample, and how full the wave is in the node and triangle trips of the traversal loop.
usage (GPU box): python tools/gpu_probe_standin.py c5 [wide]"""
import sys, tempfile
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import scenes, oracle_lib as ol
from gpu_pathtracer_amd import api
which = sys.argv[1] if len(sys.argv) > 1 else "c5"
near = {"wide": "wide"}.get(sys.argv[2], False) if len(sys.argv) > 2 else False
ls = api.LoadedScene(scenes.write_standin_scene(tempfile.mkdtemp(), which, 1920, 1080))
W, H = 1920, 1080
cam = ls.camera
with api.Renderer(ls.desc, W, H, ls.epsilon) as r:
    r.set_traversal_order(near or "reference")
    r.enable_counters(True); r.render(cam, 1, 4, reset=True); r.synchronize()
    c = r.read_probe_counters()
    r.enable_counters(False)
    r.render(cam, 1, 8, reset=True); r.synchronize(); r.kernel_time_reset()
    r.render(cam, 1, 16, reset=True); r.synchronize()
    n, ms = r.kernel_time()
s = c["samples"]
print(f"{which} {near if near else 'reference order'}: {ls.desc.n_prims} triangles, {ls.desc.n_nodes} nodes, 1920x1080, depth {ls.desc.max_depth}: {W*H*16/ms/1e3:.1f} Msamples/s")
print(f"per sample: closest-hit rays {c['closest_rays']/s:.2f}, shadow rays {c['shadow_rays']/s:.2f}, bounces {c['bounce_iters']/s:.2f}, "
      f"node visits {c['node_visits']/s:.1f}, triangle tests {c['prim_tests']/s:.1f}")
rays = c['closest_rays'] + c['shadow_rays']
print(f"per ray: node visits {c['node_visits']/rays:.1f}, triangle tests {c['prim_tests']/rays:.1f}")
if near == "wide":
    print(f"wide: trips per 64 samples {c['w_trip']*64/s:.1f}, busy lanes per trip {c['l_trip']/max(1,c['w_trip']):.2f} of 64, lanes per node block {c['node_visits']/max(1,c['w_node']):.1f}, per triangle block {c['prim_tests']/max(1,c['w_prim']):.1f}, trips with a node block {c['w_node']/max(1,c['w_trip']):.2f}, with a leaf block {c['w_prim']/max(1,c['w_trip']):.2f}")
elif c["w_node"]:
    print(f"wave trips per sample-lane: node {c['w_node']*64/s:.1f} (lanes active {c['node_visits']/c['w_node']:.1f} of 64), "
          f"triangle {c['w_prim']*64/s:.1f} (lanes active {c['prim_tests']/max(1,c['w_prim']):.1f} of 64)")
tot = c["cyc_trace"] + c["cyc_shade"]
if tot:
    print("wave time: traversal %.1f %%, everything else %.1f %%" % (100.0 * c["cyc_trace"] / tot, 100.0 * c["cyc_shade"] / tot))
