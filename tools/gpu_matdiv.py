"""How much of a stand-in's time is material divergence?  The same geometry, camera and light with EVERY material replaced by one lambertian
(an upper bound for what sorting paths by material could return; the paths themselves differ, so it is indicative only).
python tools/gpu_matdiv.py [c3,c4,c5]"""
import json, sys, tempfile
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import scenes
from gpu_pathtracer_amd import api

for which in (sys.argv[1] if len(sys.argv) > 1 else "c5").split(","):
    spp = {"c3": 32, "c4": 32, "c5": 8}[which]
    for variant in ("as defined", "all lambertian", "all mirror", "all roughconduct"):
        path = scenes.write_standin_scene(tempfile.mkdtemp(), which)
        if variant != "as defined":
            js = json.load(open(path))
            for m in js["material"]:
                if m["name"] in ("Emission",):
                    continue
                keep = m["name"]
                m.clear()
                if variant == "all lambertian": m.update({"name": keep, "bsdf": "lambertian", "diffuse": [0.6, 0.6, 0.6]})
                elif variant == "all mirror": m.update({"name": keep, "bsdf": "mirror"})
                else: m.update({"name": keep, "bsdf": "roughconduct", "alphaU": 0.025, "alphaV": 0.025, "eta": [1.0, 1.0, 1.0], "k": [1.0, 1.0, 1.0], "remap": False})
            json.dump(js, open(path, "w"))
        ls = api.LoadedScene(path)
        for sched in (0,):
            with api.Renderer(ls.desc, ls.width, ls.height, ls.epsilon) as r:
                r.render(ls.camera, 1, 2, reset=True); r.synchronize()
                best = 1e9
                for _ in range(2):
                    r.kernel_time_reset(); r.render(ls.camera, 1, spp, reset=True); r.synchronize()
                    best = min(best, r.kernel_time()[1])
                r.enable_counters(True) if sched == 0 else None
                c = None
                if sched == 0:
                    r.render(ls.camera, 1, 1, reset=True); r.synchronize()
                    c = r.read_counters()
                    r.enable_counters(False)
            extra = "" if c is None else f" closest rays/sample {c['closest_rays'] / c['samples']:.2f} shadow {c['shadow_rays'] / c['samples']:.2f} bounces {c['bounce_iters'] / c['samples']:.2f} node visits {c['node_visits'] / c['samples']:.1f} tri tests {c['prim_tests'] / c['samples']:.1f}"
            print(f"MATDIV {which} {variant:16s} scheduler {sched}: {ls.width * ls.height * spp / best / 1e3:8.1f} Msamples/s ({best:7.2f} ms){extra}", flush=True)
        ls.close()
