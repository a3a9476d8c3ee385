"""What the hand-scheduled wide loop counts in a probe build (tools/build_variant.sh probe -DPT_WIDE_PROBE=1): trips, node and
triangle blocks and the lanes in them.  usage (GPU box): GPT_LIB_PATH=var/libgpt_probe.so GPT_ALLOW_OLD_LIB=1 python tools/gpu_wide_probe.py c5"""
import sys, tempfile
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import scenes
from gpu_pathtracer_amd import api
which = sys.argv[1] if len(sys.argv) > 1 else "c5"
ls = api.LoadedScene(scenes.write_standin_scene(tempfile.mkdtemp(), which, 1920, 1080))
W, H, spp = 1920, 1080, 8
with api.Renderer(ls.desc, W, H, ls.epsilon) as r:
    r.set_traversal_order("wide")
    r.render(ls.camera, 1, 2, reset=True); r.synchronize(); r.kernel_time_reset()
    r.render(ls.camera, 1, spp, reset=True); r.synchronize()
    n, ms = r.kernel_time()
    c = r.read_probe_counters()
s = W * H * spp
t = max(1, c["w_trip"])
print(f"PROBE {which} wide (asm): {W*H*spp/ms/1e3:.1f} Msamples/s; per 64 samples: {c['w_trip']*64/s:.1f} trips, {c['w_node']*64/s:.1f} node blocks "
      f"({c['node_visits']/max(1,c['w_node']):.1f} lanes), {c['w_prim']*64/s:.1f} triangle blocks ({c['prim_tests']/max(1,c['w_prim']):.1f} lanes); "
      f"busy lanes per trip {c['l_trip']/t:.1f}; trips through the slow push {100*c['unused13']/t:.1f} %; node visits / sample {c['node_visits']/s:.1f}, triangle tests / sample {c['prim_tests']/s:.1f}")
