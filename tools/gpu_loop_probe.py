"""Where the lanes go in the hand-scheduled binary loop (probe build: tools/build_variant.sh loopprobe -DPT_LOOP_PROBE=1):
node / triangle trips, lanes active in them, lanes holding a ray, and the tail after the pool ran dry.
usage (GPU box): GPT_LIB_PATH=var/libgpt_loopprobe.so GPT_ALLOW_OLD_LIB=1 python tools/gpu_loop_probe.py [c2|c2sq|c3|c4|c5]"""
import os, sys, tempfile
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from gpu_pathtracer_amd import api, host
which = sys.argv[1] if len(sys.argv) > 1 else "c2"
if which.startswith("c2"):
    W, H = (1088, 1080) if which == "c2sq" else (1920, 1080)
    scene, meta = host.load_baked("tests/golden/cornell_pt.npz", 8)
    desc, cam, eps, spp = scene.desc, host.camera_from_meta(meta, W, H), 0.001, 64
else:
    import standins
    ls = api.LoadedScene(standins.write_standin_scene(tempfile.mkdtemp(), which, 1920, 1080))
    desc, cam, eps, spp, W, H = ls.desc, ls.camera, ls.epsilon, 8, 1920, 1080
with api.Renderer(desc, W, H, eps) as r:
    r.render(cam, 1, 2, reset=True); r.synchronize(); r.kernel_time_reset()
    r.render(cam, 1, spp, reset=True); r.synchronize()
    n, ms = r.kernel_time()
    c = r.read_probe_counters()
tn, ln, tt, lt, busy, dry, dry_busy = c["w_node"], c["node_visits"], c["w_prim"], c["prim_tests"], c["l_trip"], c["w_trip"], c["unused13"]
trips = tn + tt
s = W * H * spp
print(f"LOOP {which}: {s/ms/1e3:.0f} Msamples/s; per 64 samples {trips*64/s:.1f} trips ({tn*64/s:.1f} node with {ln/max(1,tn):.1f} lanes, {tt*64/s:.1f} triangle with {lt/max(1,tt):.1f} lanes)")
print(f"LOOP   lane-trips: active {100*(ln+lt)/(64*trips):.1f} %, waiting for the other kind of trip {100*(busy-ln-lt)/(64*trips):.1f} %, "
      f"without a ray {100*(64*trips-busy)/(64*trips):.1f} % (of which after the pool ran dry: {100*(64*dry-dry_busy)/(64*trips):.1f} % in {100*dry/trips:.1f} % of the trips, "
      f"{dry_busy/max(1,dry):.1f} lanes with a ray there)")
