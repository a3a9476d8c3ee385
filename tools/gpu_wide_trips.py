"""Trips of the wide walk in the counting build (C++ loops, drains run to the end), for whichever wide order the loaded library walks:
   GPT_LIB_PATH=var/libgpt_wide8.so python tools/gpu_wide_trips.py c5"""
import sys, tempfile
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import scenes
from gpu_pathtracer_amd import api
for which in (sys.argv[1] if len(sys.argv) > 1 else "c5").split(","):
    ls = api.LoadedScene(scenes.write_standin_scene(tempfile.mkdtemp(), which))
    with api.Renderer(ls.desc, ls.width, ls.height, ls.epsilon) as r:
        order = r.get_option("traversal_order")
        r.enable_counters(True); r.render(ls.camera, 1, 2, reset=True); r.synchronize()
        c = r.read_probe_counters()
    s = c["samples"]
    print(f"TRIPS {which} order {order}: per sample node visits {c['node_visits']/s:.1f}, triangle tests {c['prim_tests']/s:.1f}; trips per 64 samples {c['w_trip']*64/s:.1f}, "
          f"busy lanes per trip {c['l_trip']/max(1,c['w_trip']):.1f}, lanes per node block {c['node_visits']/max(1,c['w_node']):.1f}, node block in {c['w_node']/max(1,c['w_trip']):.2f} of the trips, "
          f"leaf block in {c['w_prim']/max(1,c['w_trip']):.2f}; drain cycles per trip {c['cyc_trace']/max(1,c['w_trip']):.0f}, drain share of the wave time {c['cyc_trace']/(c['cyc_trace']+c['cyc_shade']):.2f}", flush=True)
