"""Probe build (tools/build_variant.sh wfprobe -DPT_WF_PROBE=1 -DPT_WF_WIDE_ASM=0): how many trips the rays of the trace stage take, as a
histogram by powers of two - what a round of the decoupled scheduler waits for is its LONGEST ray.  GPT_LIB_PATH=var/libgpt_wfprobe.so python tools/gpu_wf_probe.py c5 wide"""
import sys, tempfile
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import scenes
from gpu_pathtracer_amd import api
which = sys.argv[1] if len(sys.argv) > 1 else "c5"
mode = sys.argv[2] if len(sys.argv) > 2 else "wide"
spp = int(sys.argv[3]) if len(sys.argv) > 3 else 2
ls = api.LoadedScene(scenes.write_standin_scene(tempfile.mkdtemp(), which))
with api.Renderer(ls.desc, ls.width, ls.height, ls.epsilon) as r:
    r.set_traversal_order(mode)
    r.set_option("scheduler", 1)
    r.render(ls.camera, 1, spp, reset=True); r.synchronize()
    c = r.read_probe_counters()
    rounds = r.get_option("last_rounds")
vals = list(c.values())
total = sum(vals)
print(f"WFPROBE {which} {mode}: {total} rays in {rounds} rounds; trips per ray, by power of two:", flush=True)
for b, v in enumerate(vals):
    if v:
        print(f"WFPROBE   {'< 2' if b == 0 else f'{1 << b} .. {(2 << b) - 1}':>16s}: {v:12d}  ({100.0 * v / total:8.4f} %)")
