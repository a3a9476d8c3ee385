"""The BASELINE.json configurations 3 - 5 as concrete inputs: the stand-ins SURVEY.md 8(d) defines from the reference's shipped
meshes (tests/golden/meshes.npz, baked by tools/bake_d_inputs.py), written as a scene directory (OBJ files + scene.json) that goes
through the product's loader like a scene directory of the reference.  Used by the tests and by bench.py's other_configs leg; nothing
here touches the oracle."""
import json
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")


def checker_texture(n=8, cell=4, a=(230, 230, 230, 255), b=(40, 60, 200, 255)):
    t = np.zeros((n * cell, n * cell, 4), dtype=np.uint8)
    for i in range(n * cell):
        for j in range(n * cell):
            t[i, j] = a if ((i // cell) + (j // cell)) % 2 == 0 else b
    return np.ascontiguousarray(t)


def sky_env(w=64, h=32):
    """closed-form lat-long sky: vertical gradient + a sun lobe; float32, no RNG"""
    v = (np.arange(h, dtype=np.float32) + 0.5) / h
    u = (np.arange(w, dtype=np.float32) + 0.5) / w
    vv, uu = np.meshgrid(v, u, indexing="ij")
    base = np.stack([0.35 + 0.4 * (1 - vv), 0.45 + 0.45 * (1 - vv), 0.6 + 0.6 * (1 - vv)], -1)
    sun = np.exp(-(((uu - 0.3) * 6) ** 2 + ((vv - 0.25) * 6) ** 2)).astype(np.float32)[..., None] * np.float32(12.0)
    return np.ascontiguousarray((base + sun * np.array([1.0, 0.9, 0.7], np.float32)).astype(np.float32))


# The reference ships no shaderball / whiteroom / sponza geometry; SURVEY.md defines their stand-ins from the meshes of
# scenes/cornell_box/geometry.  Those meshes are kept as tests/golden/meshes.npz (tools/bake_d_inputs.py); here they are written
# back as OBJ files next to a scene.json, so that everything goes through the product's loader (OBJ reader, fan triangulation,
# smooth normals for the files without vn, TRS transforms, BVH build) exactly like a scene directory of the reference.

_MESHES = None


def mesh_fixture(name):
    global _MESHES
    if _MESHES is None:
        _MESHES = np.load(os.path.join(GOLDEN, "meshes.npz"))
    key = name.replace("-", "_")
    return {k: _MESHES[key + "__" + k] for k in ("v", "vn", "vt", "counts", "fv", "fvt", "fvn")}


def _f32_text(a):
    """float32 array -> decimal text that parses back to the same float32 (9 significant digits)"""
    return np.char.mod("%.9g", a.astype(np.float64))


def write_mesh_obj(path, name):
    m = mesh_fixture(name)
    lines = []
    for tag, arr in (("v", m["v"]), ("vn", m["vn"]), ("vt", m["vt"])):
        if len(arr):
            t = _f32_text(arr)
            lines.append("\n".join(tag + " " + " ".join(row) for row in t))
    has_vt, has_vn = len(m["vt"]) > 0, len(m["vn"]) > 0
    fv, fvt, fvn = m["fv"].astype(str), m["fvt"].astype(str), m["fvn"].astype(str)
    if has_vn:
        corner = np.char.add(np.char.add(np.char.add(np.char.add(fv, "/"), fvt if has_vt else ""), "/"), fvn)
    elif has_vt:
        corner = np.char.add(np.char.add(fv, "/"), fvt)
    else:
        corner = fv
    ends = np.cumsum(m["counts"])
    starts = ends - m["counts"]
    if (m["counts"] == m["counts"][0]).all():
        c = corner.reshape(-1, int(m["counts"][0]))
        lines.append("\n".join("f " + " ".join(row) for row in c))
    else:
        lines.append("\n".join("f " + " ".join(corner[a:b]) for a, b in zip(starts, ends)))
    with open(path, "w") as f:
        f.write("\n".join(lines) + "\n")


def write_png_rgba(path, img):
    """8-bit RGBA PNG, rows top-down"""
    import struct
    import zlib
    h, w, _ = img.shape
    raw = b"".join(b"\x00" + img[y].tobytes() for y in range(h))

    def chunk(t, b):
        return struct.pack(">I", len(b)) + t + b + struct.pack(">I", zlib.crc32(t + b))
    with open(path, "wb") as f:
        f.write(b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 6, 0, 0, 0)) + chunk(b"IDAT", zlib.compress(raw)) + chunk(b"IEND", b""))


def procedural_sky(w=1024, h=512):
    """closed-form lat-long sky, vertical gradient + sun lobe, float32, no RNG (rows top-down)"""
    return sky_env(w, h)


STANDIN_MATERIALS = [
    {"name": "Left", "bsdf": "lambertian", "diffuse": [0.63, 0.065, 0.05]},                      # cornell_box/scene.json:16-31
    {"name": "Right", "bsdf": "lambertian", "diffuse": [0.14, 0.45, 0.091]},
    {"name": "General", "bsdf": "lambertian", "diffuse": [0.725, 0.725, 0.725]},
    {"name": "Emission", "bsdf": "lambertian", "diffuse": [0, 0, 0]},
    {"name": "Mirror", "bsdf": "mirror"},                                                         # :42-45
    {"name": "metal", "bsdf": "roughconduct", "alphaU": 0.025, "alphaV": 0.025, "eta": [1.0, 1.0, 1.0], "k": [1.0, 1.0, 1.0], "remap": False},   # :46-54
    {"name": "Glass", "bsdf": "dielectric", "insideIOR": 1.5, "outsideIOR": 1.0},                 # :55-60
    {"name": "LTELogo", "bsdf": "roughconduct", "alpha": 0.01, "eta": [0.143119, 0.374957, 1.442479], "k": [3.98316, 2.385721, 1.603215], "remap": True},   # shaderball/scene.json:23-30
    {"name": "Outer", "bsdf": "roughconduct", "alphaU": 0.0025, "alphaV": 0.25, "eta": [0.155265, 0.116723, 0.138381], "k": [4.828343, 3.122246, 2.14695], "remap": True},   # :31-39
    {"name": "Plastic_Black", "bsdf": "substrate", "alpha": 0.1, "specular": [0.04, 0.04, 0.04], "diffuse": [0.00631, 0.00631, 0.00631]},   # coffee/scene.json:32-38
    {"name": "BackGround", "bsdf": "lambertian", "diffuse": "textures/Checker.png"},              # shaderball/scene.json:13-17 (procedural 8x8 checker)
]
C5_CORE = [("dragon", "metal", {"scale": [0.08, 0.08, 0.08], "translate": [-0.35, 0, -0.3], "rotate": [0, 30, 0]}),
           ("bunny2", "Glass", {"scale": [0.06, 0.06, 0.06], "translate": [0.45, 0, 0.3]}),
           ("teapot", "Plastic_Black", {"scale": [0.08, 0.08, 0.08], "translate": [-0.5, 0, 0.55]})]
C5_SPHERE_MATERIALS = ["Mirror", "Left", "metal", "Glass", "General", "Plastic_Black", "Right", "LTELogo", "Outer"]


def write_standin_scene(directory, which, width=None, height=None):
    """which: "c3" (shaderball stand-in: sphere x 3 + cube-subdiv on the floor, shaderball camera / light radiance / metals,
    glass, substrate, checker), "c5core" (Cornell walls + dragon + bunny2 + teapot + light: the 175 998-primitive scene whose
    BVH SURVEY.md measured at 112 947 nodes), "c5" (c5core + 9 spheres: ~250k triangles, the sponza stand-in), "c4" (the
    geometry of c5 under a procedural sky instead of the area light: the whiteroom-with-env stand-in).  Returns the json path."""
    import shutil
    src = os.path.join(ROOT, "scenes", "cornell_pt", "geometry")
    os.makedirs(os.path.join(directory, "geometry"), exist_ok=True)
    os.makedirs(os.path.join(directory, "textures"), exist_ok=True)
    for name in ("floor", "ceil", "back", "left", "right", "light"):
        shutil.copy(os.path.join(src, name + ".obj"), os.path.join(directory, "geometry", name + ".obj"))
    js = {"integrator": "pt", "material": [dict(m) for m in STANDIN_MATERIALS if which == "c3" or m["name"] != "BackGround"]}
    units, lights = [], []
    if which == "c3":
        js.update({"screen_width": 1920, "screen_height": 1080, "maxDepth": 10, "epsilon": 0.0005,
                   "camera": {"position": [-0.3, 0.5, -0.5], "lookat": [0.0, 0.075, 0.0], "up": [0.0, 1.0, 0.0], "fov": 37.0}})   # shaderball/scene.json:2-11
        write_png_rgba(os.path.join(directory, "textures", "Checker.png"), checker_texture(8, 8))
        for name in ("sphere", "cube-subdiv"):
            write_mesh_obj(os.path.join(directory, "geometry", name + ".obj"), name)
        units = [{"mesh": "geometry/floor.obj", "material": "BackGround"},
                 {"mesh": "geometry/sphere.obj", "material": "LTELogo", "scale": [0.15, 0.15, 0.15], "translate": [0.0, 0.075, 0.0]},
                 {"mesh": "geometry/sphere.obj", "material": "Glass", "scale": [0.12, 0.12, 0.12], "translate": [-0.17, 0.06, 0.06]},
                 {"mesh": "geometry/sphere.obj", "material": "Plastic_Black", "scale": [0.1, 0.1, 0.1], "translate": [0.16, 0.05, 0.1]},
                 {"mesh": "geometry/cube-subdiv.obj", "material": "Outer", "scale": [0.06, 0.06, 0.06], "translate": [0.02, 0.0505, 0.2], "rotate": [0, 25, 0]}]
        lights = [{"mesh": "geometry/light.obj", "material": "Emission", "radiance": [9.5, 9.5, 9.5], "translate": [0.0, -1.38, 0.0]}]   # :72-78
    else:
        cam = {"position": [0, 1.0, 6.8], "lookat": [0, 1.0, 0], "fov": 19.5, "apertureRadius": 0.0, "focalDistance": 7.0}
        if which == "c4":
            js.update({"screen_width": 1920, "screen_height": 1080, "maxDepth": 7, "epsilon": 0.001, "camera": cam})
        else:
            js.update({"screen_width": 3840, "screen_height": 2160, "maxDepth": 16, "epsilon": 0.001, "camera": cam})
        units = [{"mesh": "geometry/%s.obj" % n, "material": m} for n, m in
                 (("floor", "General"), ("ceil", "General"), ("back", "General"), ("left", "Left"), ("right", "Right"))]
        for name, mat, trs in C5_CORE:
            write_mesh_obj(os.path.join(directory, "geometry", name + ".obj"), name)
            units.append(dict({"mesh": "geometry/%s.obj" % name, "material": mat}, **trs))
        if which != "c5core":
            write_mesh_obj(os.path.join(directory, "geometry", "sphere.obj"), "sphere")
            for i, mat in enumerate(C5_SPHERE_MATERIALS):
                x, z = -0.6 + 0.6 * (i % 3), -0.6 + 0.6 * (i // 3)
                units.append({"mesh": "geometry/sphere.obj", "material": mat, "scale": [0.3, 0.3, 0.3], "translate": [x, 1.45 + 0.05 * (i % 2), z]})
        if which == "c4":
            from gpu_pathtracer_amd import api
            env = procedural_sky(1024, 512)
            api.save_pfm(os.path.join(directory, "textures", "sky.pfm"), 1024, 512, env[::-1].copy())     # PFM is bottom-up on disk
            lights = [{"infinite": "textures/sky.pfm", "rotate": [0, 30, 0]}]
        else:
            lights = [{"mesh": "geometry/light.obj", "material": "Emission", "radiance": [17.0, 12.0, 4.0]}]
    if width:
        js["screen_width"], js["screen_height"] = int(width), int(height)
    js["scene"], js["light"] = units, lights
    path = os.path.join(directory, "scene.json")
    json.dump(js, open(path, "w"), indent=1)
    return path


def write_smoke_scene(directory):
    """The reference's default scene (scenes/cornell_box/scene.json: Volpath, 17 bounces, a 100 x 100 x 40 density grid in
    a material-less box, bare Cornell walls) rebuilt on disk from what this repository holds: the wall / light meshes of
    scenes/cornell_pt (the same geometry), the density grid fixture (tests/golden/reference_density_grid.npz, written back
    as the text file the loader reads) and the box mesh from its corner coordinates.  Returns the json path.
    tests/test_scene_loader.py checks, where /root/reference exists, that it loads to the same scene as the shipped file."""
    import json
    import shutil
    src = os.path.join(ROOT, "scenes", "cornell_pt")
    os.makedirs(os.path.join(directory, "geometry"), exist_ok=True)
    for name in ("floor", "ceil", "back", "left", "right", "light"):
        shutil.copy(os.path.join(src, "geometry", name + ".obj"), os.path.join(directory, "geometry", name + ".obj"))
    g = np.load(os.path.join(ROOT, "tests", "golden", "reference_density_grid.npz"))
    with open(os.path.join(directory, "geometry", "density.d"), "w") as f:
        f.write("".join("%d.%06d\n" % divmod(int(v), 1000000) for v in g["millionths"]))
    lo, hi = (-0.63, 0.27, -0.2415), (0.693, 1.593, 0.2415)
    x0, y0, z0 = lo
    x1, y1, z1 = hi
    with open(os.path.join(directory, "geometry", "density_render.obj"), "w") as f:
        for v in ((x0, y0, z0), (x1, y0, z0), (x1, y1, z0), (x0, y1, z0), (x0, y0, z1), (x1, y0, z1), (x1, y1, z1), (x0, y1, z1)):
            f.write("v %g %g %g\n" % v)
        f.write("vn -1 0 0\nvn 1 0 0\nvn 0 0 1\nvn 0 0 -1\nvn 0 1 0\nvn 0 -1 0\n")
        f.write("f 1//4 2//4 3//4 4//4\nf 5//3 6//3 7//3 8//3\nf 8//5 7//5 3//5 4//5\nf 5//6 6//6 2//6 1//6\nf 1//1 5//1 8//1 4//1\nf 6//2 7//2 3//2 2//2\n")
    js = json.load(open(os.path.join(src, "scene.json")))
    js.update({"screen_width": 512, "screen_height": 512, "integrator": "vpt", "maxDepth": 17, "epsilon": 0.001})
    js["medium"] = [{"type": "homogeneous", "sigmaA": [0.0014, 0.0025, 0.0142], "sigmaS": [0.70, 1.22, 1.90], "scale": 25.0, "name": "vol"},
                    {"type": "heterogeneous", "sigmaA": [10.0, 10.0, 10.0], "sigmaS": [90.0, 90.0, 90.0], "nx": int(g["nx"]), "ny": int(g["ny"]),
                     "nz": int(g["nz"]), "p0": list(lo), "p1": list(hi), "density": "geometry/density.d", "iterMax": 2000, "name": "hhh"}]
    js["scene"] = [{"mesh": "geometry/%s.obj" % n, "material": m} for n, m in
                   (("floor", "General"), ("ceil", "General"), ("back", "General"), ("left", "Left"), ("right", "Right"))]
    js["scene"].append({"mesh": "geometry/density_render.obj", "inside": "hhh", "outside": ""})
    path = os.path.join(directory, "scene.json")
    json.dump(js, open(path, "w"))
    return path
