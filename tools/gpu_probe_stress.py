"""Time split of the path kernel on the 253k-triangle stand-in (probe build: PT_ASM_IN_COUNT=1)."""
import sys
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import scenes, oracle_lib as ol
from gpu_pathtracer_amd import api
scene, meta = scenes.stress_scene(1.0, max_depth=16)
W, H = 1920, 1080
cam = ol.cornell_camera(meta, W, H)
with api.Renderer(scene.desc, W, H, 0.001) as r:
    r.enable_counters(True); r.render(cam, 1, 8, reset=True); r.synchronize()
    c = r.read_probe_counters()
tot = c["cyc_trace"] + c["cyc_shade"]
print("traversal %.1f %%  direct %.1f %%  hit shading %.1f %%  finish+regen %.1f %%  pool+rest %.1f %%" % (
    100.0 * c["cyc_trace"] / tot, 100.0 * c["cyc_direct"] / tot, 100.0 * c["cyc_hit"] / tot, 100.0 * c["cyc_regen"] / tot,
    100.0 * (c["cyc_shade"] - c["cyc_direct"] - c["cyc_hit"] - c["cyc_regen"]) / tot))
