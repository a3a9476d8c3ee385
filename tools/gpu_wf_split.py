"""Probe build (tools/build_variant.sh wfsplit -DPT_WF_PROBE=2): where the waves of the decoupled scheduler's kernel spend their time - shade
phase, trace phase, waiting at the workgroup barriers - and how many rounds a workgroup makes.  GPT_LIB_PATH=var/libgpt_wfsplit.so python tools/gpu_wf_split.py c5 wide"""
import sys, tempfile
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import scenes
from gpu_pathtracer_amd import api
which = sys.argv[1] if len(sys.argv) > 1 else "c5"
mode = sys.argv[2] if len(sys.argv) > 2 else "wide"
spp = int(sys.argv[3]) if len(sys.argv) > 3 else {"c3": 32, "c4": 32, "c5": 8}[which]
ls = api.LoadedScene(scenes.write_standin_scene(tempfile.mkdtemp(), which))
with api.Renderer(ls.desc, ls.width, ls.height, ls.epsilon) as r:
    r.set_traversal_order(mode)
    r.set_option("scheduler", 1)
    r.render(ls.camera, 1, 2, reset=True); r.synchronize()
    r.kernel_time_reset(); r.render(ls.camera, 1, spp, reset=True); r.synchronize()
    ms = r.kernel_time()[1]
    c = list(r.read_probe_counters().values())
shade, trace, wait, rounds, chunks, waves = c[:6]
tot = shade + trace + wait
print(f"WFSPLIT {which} {mode}: {ms:.2f} ms per batch of {spp} iterations; per wave: shade {shade / tot:.3f}, trace {trace / tot:.3f}, barriers {wait / tot:.3f} of {tot / waves / 1e6:.2f} M cycles; "
      f"{rounds / waves:.1f} rounds per workgroup, {chunks / waves:.1f} chunks shaded per wave ({chunks / rounds:.2f} per round)", flush=True)
nt, nl, lt, ll, inf = c[6:11]
if nt + lt:
    print(f"WFSTREAM {which} {mode}: node trips {nt} with {nl / max(nt, 1):.1f} lanes, leaf trips {lt} with {ll / max(lt, 1):.1f} lanes, {inf / (nt + lt):.1f} rays in flight per trip, "
          f"{(nt + lt) * 64 / (ls.width * ls.height * spp):.1f} trips per 64 samples", flush=True)
    pre, wt, proc = c[11:14]
    if pre + wt + proc:
        print(f"WFSTREAM {which} {mode}: cycles per trip of a wave: {pre / (nt + lt):.0f} up to the loads, {wt / (nt + lt):.0f} until they are back, {proc / (nt + lt):.0f} the step "
              f"(trace phase {trace / (nt + lt):.0f} per trip)", flush=True)
