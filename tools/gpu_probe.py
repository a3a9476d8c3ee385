import sys, os
sys.path.insert(0, '.')
from gpu_pathtracer_amd import api, host
W, H, D = 1920, 1080, 8
scene, meta = host.load_baked("tests/golden/cornell_pt.npz", D)
cam = host.camera_from_meta(meta, W, H)
r = api.Renderer(scene.desc, W, H, 0.001)
r.enable_counters(True); r.render(cam, 1, 16, reset=True); r.synchronize()
c = r.read_probe_counters(); print(c)
u = lambda l, w: c[l] / (64.0 * c[w]) if c[w] else float('nan')
print("lane utilisation: node loop %.3f  triangle loop %.3f  bounce trip %.3f" %
      (u("node_visits", "w_node"), u("prim_tests", "w_prim"), u("l_trip", "w_trip")))
tot = c["cyc_trace"] + c["cyc_shade"]
print("split of the rest: direct-light resolution %.1f %%  hit shading %.1f %%  finish + regeneration %.1f %%  pool deposit/pickup + item fetch %.1f %%" % (
    100.0 * c["cyc_direct"] / tot, 100.0 * c["cyc_hit"] / tot, 100.0 * c["cyc_regen"] / tot,
    100.0 * (c["cyc_shade"] - c["cyc_direct"] - c["cyc_hit"] - c["cyc_regen"]) / tot))
print("traversal trips per sample: %.3f wave-trips, busy lanes per trip %.1f/64" % (c["w_trip"]/c["samples"]*1.0, c["l_trip"]/max(1,c["w_trip"])))
print("per sample: wave node trips x64 = %.1f lane-slots (useful %.1f); tri %.1f (useful %.1f); trips %.2f" % (
    64.0*c["w_node"]/c["samples"], c["node_visits"]/c["samples"], 64.0*c["w_prim"]/c["samples"], c["prim_tests"]/c["samples"], 64.0*c["w_trip"]/c["samples"]))
print("wave time split (s_memtime, counting build): traversal %.1f %%  shading+rest %.1f %%" % (
    100.0 * c["cyc_trace"] / (c["cyc_trace"] + c["cyc_shade"]), 100.0 * c["cyc_shade"] / (c["cyc_trace"] + c["cyc_shade"])))
if c["unused13"]:
    t = tot
    print("PT_SUBPROBES build: make_hit+material %.1f %%  light sample+eval %.1f %%  MIS sample+pretest %.1f %%  continuation+roulette %.1f %%  (of all wave time)" % (
        100.0*c["w_node"]/t, 100.0*c["w_prim"]/t, 100.0*c["w_trip"]/t, 100.0*c["l_trip"]/t))
