"""Time split of the path kernel on the config-3 stand-in (probe build: PT_ASM_IN_COUNT=1)."""
import sys
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import scenes, oracle_lib as ol
from gpu_pathtracer_amd import api
extra = scenes.concat([scenes.uv_sphere((-0.45, 0.45, 0.3), 0.4, 8, nu=24, nv=16), scenes.uv_sphere((0.4, 0.35, 0.45), 0.33, 7, nu=24, nv=16),
                       scenes.uv_sphere((0.05, 1.25, -0.3), 0.35, 10, nu=24, nv=16)])
scene, meta = scenes.zoo_scene(max_depth=10, extra=extra, assign={"short": 5, "tall": 13, "floor": 12, "back": 9})
W, H = 1920, 1080
cam = ol.cornell_camera(meta, W, H)
with api.Renderer(scene.desc, W, H, 0.0005) as r:
    r.enable_counters(True); r.render(cam, 1, 8, reset=True); r.synchronize()
    c = r.read_probe_counters()
tot = c["cyc_trace"] + c["cyc_shade"]
print("triangles", len(scene.prims), "| traversal %.1f %%  direct %.1f %%  hit shading %.1f %%  finish+regen %.1f %%  pool+rest %.1f %%" % (
    100.0 * c["cyc_trace"] / tot, 100.0 * c["cyc_direct"] / tot, 100.0 * c["cyc_hit"] / tot, 100.0 * c["cyc_regen"] / tot,
    100.0 * (c["cyc_shade"] - c["cyc_direct"] - c["cyc_hit"] - c["cyc_regen"]) / tot))
print({k: c[k] / c["samples"] for k in ("bounce_iters", "closest_rays", "shadow_rays")})
