"""Probe build (tools/build_variant.sh wflanes -DPT_WF_PROBE=3): the hand-scheduled trace phase's trips and the lanes in them.
GPT_LIB_PATH=var/libgpt_wflanes.so python tools/gpu_wf_lanes.py c5"""
import sys, tempfile
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import scenes
from gpu_pathtracer_amd import api
which = sys.argv[1] if len(sys.argv) > 1 else "c5"
spp = int(sys.argv[2]) if len(sys.argv) > 2 else {"c3": 32, "c4": 32, "c5": 8}[which]
ls = api.LoadedScene(scenes.write_standin_scene(tempfile.mkdtemp(), which))
with api.Renderer(ls.desc, ls.width, ls.height, ls.epsilon) as r:
    r.set_traversal_order("wide")
    r.set_option("scheduler", 1)
    r.render(ls.camera, 1, spp, reset=True); r.synchronize()
    c = list(r.read_probe_counters().values())
trips, busy, node, leaf = c[8:12]
n = ls.width * ls.height * spp
print(f"WFLANES {which}: {trips / n * 64:.1f} trips per 64 samples; lanes holding a ray {busy / trips:.1f}, in the node block {node / trips:.1f}, in the triangle block {leaf / trips:.1f} of 64", flush=True)
