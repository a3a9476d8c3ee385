"""Wave-time split of the render kernel (probe build: tools/build_variant.sh asmcount -DPT_ASM_IN_COUNT=1 keeps the hand-scheduled
loops in the counting kernels): draining the pool vs everything else, and the parts of everything else.
usage (GPU box): GPT_LIB_PATH=var/libgpt_asmcount.so GPT_ALLOW_OLD_LIB=1 python tools/gpu_timesplit.py c5 wide"""
import sys, tempfile
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import scenes
from gpu_pathtracer_amd import api
which = sys.argv[1] if len(sys.argv) > 1 else "c5"
mode = sys.argv[2] if len(sys.argv) > 2 else "wide"
W, H, spp = 1920, 1080, 8
if which == "c2":
    from gpu_pathtracer_amd import host
    _scene, _meta = host.load_baked("tests/golden/cornell_pt.npz", 8)
    class ls: pass
    ls.desc, ls.camera, ls.epsilon = _scene.desc, host.camera_from_meta(_meta, W, H), 0.001
else:
    ls = api.LoadedScene(scenes.write_standin_scene(tempfile.mkdtemp(), which, 1920, 1080))
with api.Renderer(ls.desc, W, H, ls.epsilon) as r:
    if which == "c2" and mode != "reference":
        r.set_option("lds_scene", 0)
    r.set_traversal_order(mode)
    r.enable_counters(True)
    r.render(ls.camera, 1, 2, reset=True); r.synchronize(); r.kernel_time_reset()
    r.render(ls.camera, 1, spp, reset=True); r.synchronize()
    n, ms = r.kernel_time()
    c = r.read_probe_counters()
tot = c["cyc_trace"] + c["cyc_shade"]
s = c["samples"]
print(f"SPLIT {which} {mode}: counting kernel {W*H*spp/ms/1e3:.0f} Msamples/s; wave time: drain {100*c['cyc_trace']/tot:.1f} %, rest {100*c['cyc_shade']/tot:.1f} % "
      f"(direct-light resolution {100*c['cyc_direct']/tot:.1f}, hit shading {100*c['cyc_hit']/tot:.1f}, finish + regeneration {100*c['cyc_regen']/tot:.1f}, "
      f"deposit / pick-up {100*(c['cyc_shade']-c['cyc_direct']-c['cyc_hit']-c['cyc_regen'])/tot:.1f}); "
      f"per sample: bounces {c['bounce_iters']/s:.2f}, closest rays {c['closest_rays']/s:.2f}, shadow rays {c['shadow_rays']/s:.2f}; cycles per wave per 64 samples {tot*64/s:.0f}")
import os
if "subprobe" in os.environ.get("GPT_LIB_PATH", ""):
    # PT_SUBPROBES build: counters 6..9 and 13 hold cycles of the hit-shading sub-phases (booked by the first active lane)
    sub = [c["unused13"], c["w_node"], c["w_prim"], c["w_trip"], c["l_trip"]]
    names = ["between rounds (incl. drain)", "make_hit + material", "light sample + BSDF eval", "MIS sample + emitter pre-test", "continuation sample + roulette"]
    hs = sum(sub[1:])
    print("SUB   " + "; ".join(f"{n} {100*v/max(1,hs):.1f} %" for n, v in zip(names[1:], sub[1:])) + f"  (of the hit-shading block; block = {hs*64/s:.0f} cycles per 64 samples)")
if "laneprobe" in os.environ.get("GPT_LIB_PATH", ""):
    # PT_LANEPROBE build: counters 6..9 = rounds with a hit-shading block, lanes in it, drains, lanes that deposited rays for a drain
    print(f"LANES {which} {mode}: rounds per 64 samples {c['w_trip']*64/s:.2f}; lanes whose path takes part in a round {c['l_trip']/max(1,c['w_trip']):.1f}; "
          f"rounds with hit shading {100*c['w_node']/max(1,c['w_trip']):.0f} %, lanes in the hit-shading block {c['w_prim']/max(1,c['w_node']):.1f} of 64; "
          f"hit-shading lane-rounds per sample {c['w_prim']/s:.2f}")

