#!/usr/bin/env python3
"""Write scenes/cornell_pt/ (scene.json + OBJ meshes) from the baked fixture tests/golden/cornell_pt.npz, so the
C++ loader (gpt_scene_load) has an on-disk scene that exists on the GPU box too.  Floats are written with 9
significant digits, which round-trips float32 exactly."""
import json, os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from gpu_pathtracer_amd import scene_types as st

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
z = np.load(os.path.join(ROOT, "tests", "golden", "cornell_pt.npz"))
prims = z["prims"].view(st.PRIMITIVE)
meta = json.loads(str(z["meta"]))
raw_n = z["raw_normals"]        # the OBJ's own (un-normalised) normals: the loader normalises once, like the reference
parts = {"floor": (0, 2), "ceil": (2, 4), "back": (4, 6), "left": (6, 8), "right": (8, 10), "short": (10, 22),
         "tall": (22, 34), "light": (34, 36)}
g = lambda x: f"{float(x):.9g}"
out = os.path.join(ROOT, "scenes", "cornell_pt")
for name, (a, b) in parts.items():
    with open(os.path.join(out, "geometry", name + ".obj"), "w") as f:
        f.write(f"# {name}: {b - a} triangles, one v/vt/vn triple per corner\n")
        k = 0
        for i in range(a, b):
            t = prims[i]["triangle"]
            for c, vn in enumerate(("v1", "v2", "v3")):
                v = t[vn]
                f.write(f"v {g(v['v']['x'])} {g(v['v']['y'])} {g(v['v']['z'])}\n")
                f.write(f"vn {g(raw_n[i][c][0])} {g(raw_n[i][c][1])} {g(raw_n[i][c][2])}\n")
                f.write(f"vt {g(v['uv'][0])} {g(v['uv'][1])}\n")
            f.write(f"f {k+1}/{k+1}/{k+1} {k+2}/{k+2}/{k+2} {k+3}/{k+3}/{k+3}\n")
            k += 3
scene = {
    "screen_width": 512, "screen_height": 512, "integrator": "pt", "maxDepth": 8, "epsilon": meta["epsilon"],
    "camera": {"position": meta["camera"]["position"], "lookat": meta["camera"]["lookat"], "fov": meta["camera"]["fov"],
               "apertureRadius": 0.0, "focalDistance": 7.0},
    "material": [
        {"name": "Left", "bsdf": "lambertian", "diffuse": [0.63, 0.065, 0.05]},
        {"name": "Right", "bsdf": "lambertian", "diffuse": [0.14, 0.45, 0.091]},
        {"name": "General", "bsdf": "lambertian", "diffuse": [0.725, 0.725, 0.725]},
        {"name": "General", "bsdf": "lambertian", "diffuse": [0.725, 0.725, 0.725]},
        {"name": "Emission", "bsdf": "lambertian", "diffuse": [0, 0, 0]},
        {"name": "Mirror", "bsdf": "mirror"},
        {"name": "metal", "bsdf": "roughconduct", "alphaU": 0.025, "alphaV": 0.025, "eta": [1, 1, 1], "k": [1, 1, 1], "remap": False},
        {"name": "Glass", "bsdf": "dielectric", "insideIOR": 1.5, "outsideIOR": 1.0},
    ],
    "scene": [{"mesh": f"geometry/{n}.obj", "material": m} for n, m in
              (("floor", "General"), ("ceil", "General"), ("back", "General"), ("left", "Left"), ("right", "Right"),
               ("short", "General"), ("tall", "General"))],
    "light": [{"mesh": "geometry/light.obj", "material": "Emission", "radiance": meta["light_radiance"]}],
}
json.dump(scene, open(os.path.join(out, "scene.json"), "w"), indent=1)
print("wrote", out)
