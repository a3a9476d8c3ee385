"""Deterministic test scenes built from the Cornell fixture and procedural parts (test infrastructure)."""
import json
import os

import numpy as np

import oracle_lib as ol
from gpu_pathtracer_amd import scene_types as st

PRIM = st.PRIMITIVE

# primitive ranges of tests/golden/cornell_pt.npz before BVH reordering
CORNELL_PARTS = {"floor": (0, 2), "ceil": (2, 4), "back": (4, 6), "left": (6, 8), "right": (8, 10),
                 "short": (10, 22), "tall": (22, 34), "light": (34, 36)}


def cornell_raw():
    z = np.load(os.path.join(ol.GOLDEN, "cornell_pt.npz"))
    meta = json.loads(str(z["meta"]))
    return z["prims"].view(st.PRIMITIVE).copy(), z["materials"].view(st.MATERIAL).copy(), meta


from standins import checker_texture, sky_env


def make_tri(p1, p2, p3, n1, n2, n3, uv1=(0, 0), uv2=(1, 0), uv3=(1, 1), mat=0, light=-1):
    p = np.zeros((), dtype=st.PRIMITIVE)
    t = p["triangle"]
    for name, pos, nor, uv in (("v1", p1, n1, uv1), ("v2", p2, n2, uv2), ("v3", p3, n3, uv3)):
        t[name]["v"] = st.f3(pos)
        t[name]["n"] = st.f3(nor)
        t[name]["uv"] = np.asarray(uv, np.float32)
    t["matIdx"], t["bssrdfIdx"], t["lightIdx"], t["mediumInside"], t["mediumOutside"] = mat, -1, light, -1, -1
    return p


def uv_sphere(center, radius, mat, nu=12, nv=8):
    """smooth-shaded sphere: interpolated normals differ from the face normal, uvs are non-degenerate"""
    c = np.asarray(center, np.float32)
    prims = []

    def vert(i, j):
        th = np.float32(np.pi) * np.float32(j) / np.float32(nv)
        ph = np.float32(2 * np.pi) * np.float32(i) / np.float32(nu)
        n = np.array([np.sin(th) * np.cos(ph), np.cos(th), np.sin(th) * np.sin(ph)], np.float32)
        n = n / np.sqrt(np.float32(n @ n), dtype=np.float32)
        return c + np.float32(radius) * n, n, (np.float32(i) / nu, np.float32(j) / nv)

    for j in range(nv):
        for i in range(nu):
            a, b, cc, d = vert(i, j), vert(i + 1, j), vert(i + 1, j + 1), vert(i, j + 1)
            if j != 0:
                prims.append(make_tri(a[0], b[0], cc[0], a[1], b[1], cc[1], a[2], b[2], cc[2], mat))
            if j != nv - 1:
                prims.append(make_tri(a[0], cc[0], d[0], a[1], cc[1], d[1], a[2], cc[2], d[2], mat))
    out = np.zeros(len(prims), dtype=st.PRIMITIVE)
    for k, p in enumerate(prims):
        out[k] = p
    return out


def box_mesh(lo, hi, mat, inside=-1, outside=-1):
    """axis-aligned box, 12 triangles, face normals pointing outwards; mat = -1 makes it a material-less surface that
    only separates the media `inside` / `outside` (parsescene.cpp:345-357)"""
    lo, hi = np.asarray(lo, np.float32), np.asarray(hi, np.float32)

    def corner(ix, iy, iz):
        return np.array([hi[0] if ix else lo[0], hi[1] if iy else lo[1], hi[2] if iz else lo[2]], np.float32)

    faces = [  # (normal, four corners counter-clockwise seen from outside)
        ((-1, 0, 0), [(0, 0, 0), (0, 0, 1), (0, 1, 1), (0, 1, 0)]), ((1, 0, 0), [(1, 0, 0), (1, 1, 0), (1, 1, 1), (1, 0, 1)]),
        ((0, -1, 0), [(0, 0, 0), (1, 0, 0), (1, 0, 1), (0, 0, 1)]), ((0, 1, 0), [(0, 1, 0), (0, 1, 1), (1, 1, 1), (1, 1, 0)]),
        ((0, 0, -1), [(0, 0, 0), (0, 1, 0), (1, 1, 0), (1, 0, 0)]), ((0, 0, 1), [(0, 0, 1), (1, 0, 1), (1, 1, 1), (0, 1, 1)]),
    ]
    out = np.zeros(12, dtype=st.PRIMITIVE)
    k = 0
    for n, cs in faces:
        a, b, c, d = (corner(*ix) for ix in cs)
        out[k] = make_tri(a, b, c, n, n, n, (0, 0), (1, 0), (1, 1), mat)
        out[k + 1] = make_tri(a, c, d, n, n, n, (0, 0), (1, 1), (0, 1), mat)
        k += 2
    out["triangle"]["mediumInside"] = inside
    out["triangle"]["mediumOutside"] = outside
    return out


def smoke_grid(nx=24, ny=20, nz=16, seed=3):
    """a lumpy density field (float32, shape (nz, ny, nx)) with empty regions and a few dense blobs"""
    rng = np.random.default_rng(seed)
    z, y, x = np.meshgrid(np.linspace(0, 1, nz), np.linspace(0, 1, ny), np.linspace(0, 1, nx), indexing="ij")
    g = np.zeros((nz, ny, nx), np.float64)
    for _ in range(5):
        c = rng.uniform(0.15, 0.85, 3)
        r = rng.uniform(0.12, 0.3)
        g += rng.uniform(0.4, 1.5) * np.exp(-(((x - c[0]) ** 2 + (y - c[1]) ** 2 + (z - c[2]) ** 2) / (r * r)))
    g[g < 0.08] = 0.0
    return np.ascontiguousarray(g.astype(np.float32))


def random_soup(n, seed, lo=(-0.9, 0.05, -0.9), hi=(0.9, 1.9, 0.9), size=0.12, mats=(2,)):
    rng = np.random.default_rng(seed)
    out = np.zeros(n, dtype=st.PRIMITIVE)
    lo, hi = np.asarray(lo, np.float32), np.asarray(hi, np.float32)
    for k in range(n):
        c = lo + (hi - lo) * rng.random(3).astype(np.float32)
        p = [c + (rng.random(3).astype(np.float32) - np.float32(0.5)) * np.float32(size) for _ in range(3)]
        fn = np.cross(p[1] - p[0], p[2] - p[0]).astype(np.float32)
        ln = np.sqrt(np.float32(fn @ fn), dtype=np.float32)
        fn = fn / ln if ln > 0 else np.array([0, 1, 0], np.float32)
        uvs = rng.random((3, 2)).astype(np.float32)
        out[k] = make_tri(p[0], p[1], p[2], fn, fn, fn, uvs[0], uvs[1], uvs[2], int(mats[k % len(mats)]))
    return out


def big_soup(n, seed, lo=(-0.95, 0.02, -0.95), hi=(0.95, 1.95, 0.95), size=0.02, mats=(2, 5, 7, 13)):
    """n small random triangles, vectorised (random_soup builds them one by one): for scenes of a million primitives"""
    rng = np.random.default_rng(seed)
    lo, hi = np.asarray(lo, np.float32), np.asarray(hi, np.float32)
    c = (lo + (hi - lo) * rng.random((n, 1, 3)).astype(np.float32)).astype(np.float32)
    p = (c + (rng.random((n, 3, 3)).astype(np.float32) - np.float32(0.5)) * np.float32(size)).astype(np.float32)
    fn = np.cross(p[:, 1] - p[:, 0], p[:, 2] - p[:, 0]).astype(np.float32)
    ln = np.sqrt((fn * fn).sum(-1, keepdims=True), dtype=np.float32)
    fn = np.where(ln > 0, fn / np.where(ln > 0, ln, 1), np.array([0, 1, 0], np.float32)).astype(np.float32)
    uv = rng.random((n, 3, 2)).astype(np.float32)
    out = np.zeros(n, dtype=PRIM)
    t = out["triangle"]
    for k, name in enumerate(("v1", "v2", "v3")):
        for a, comp in enumerate("xyz"):
            t[name]["v"][comp] = p[:, k, a]
            t[name]["n"][comp] = fn[:, a]
        t[name]["uv"] = uv[:, k]
    t["matIdx"] = np.asarray(mats, np.int32)[np.arange(n) % len(mats)]
    t["bssrdfIdx"] = t["lightIdx"] = t["mediumInside"] = t["mediumOutside"] = -1
    return out


def concat(parts):
    n = sum(len(p) for p in parts)
    out = np.zeros(n, dtype=st.PRIMITIVE)
    o = 0
    for p in parts:
        out[o:o + len(p)] = p
        o += len(p)
    return out


def material_table():
    """index: 0 Left 1 Right 2 General 3 dup 4 Emission 5 Mirror 6 metal 7 Glass (Cornell json) + extras"""
    _, mats, _ = cornell_raw()
    extra = [
        # 8: anisotropic rough conductor (shaderball "Outer"-like), already-remapped alphas
        st.make_material(st.MT_ROUGHCONDUCTOR, alphaU=0.05, alphaV=0.4, eta=(0.2, 0.92, 1.1), k=(3.9, 2.45, 2.14),
                         specular=(0.9, 0.85, 0.8)),
        # 9: substrate with texture 0
        st.make_material(st.MT_SUBSTRATE, alphaU=0.1, alphaV=0.1, specular=(0.04, 0.04, 0.04), textureIdx=0),
        # 10: substrate, plain diffuse
        st.make_material(st.MT_SUBSTRATE, alphaU=0.2, alphaV=0.05, diffuse=(0.4, 0.1, 0.1), specular=(0.1, 0.1, 0.1)),
        # 11: rough dielectric
        st.make_material(st.MT_ROUGHDIELECTRIC, alphaU=0.15, alphaV=0.15, insideIOR=1.5, outsideIOR=1.0),
        # 12: textured lambertian
        st.make_material(st.MT_LAMBERTIAN, textureIdx=0),
        # 13: isotropic rough conductor, coarse
        st.make_material(st.MT_ROUGHCONDUCTOR, alphaU=0.3, alphaV=0.3, eta=(1.0, 1.0, 1.0), k=(1.0, 1.0, 1.0)),
    ]
    out = np.zeros(len(mats) + len(extra), dtype=st.MATERIAL)
    out[:len(mats)] = mats
    for i, m in enumerate(extra):
        out[len(mats) + i] = m
    return out


def zoo_scene(max_depth=8, with_env=False, with_area_light=True, assign=None, extra=None):
    """Cornell box whose parts carry the material types under test."""
    prims, _, meta = cornell_raw()
    mats = material_table()
    assign = assign or {"short": 7, "tall": 5, "floor": 12, "back": 9, "left": 8, "right": 10, "ceil": 2}
    for part, m in assign.items():
        a, b = CORNELL_PARTS[part]
        prims["triangle"]["matIdx"][a:b] = m
    parts = [prims if with_area_light else prims[:34]]
    if extra is not None:
        parts.append(extra)
    allp = concat(parts)
    env = sky_env() if with_env else None
    uvw = None
    if with_env:
        # a rotation about y by 30 degrees, written out (u, v, w rows)
        c, s = np.float32(np.cos(np.pi / 6)), np.float32(np.sin(np.pi / 6))
        uvw = ((c, 0.0, -s), (0.0, 1.0, 0.0), (s, 0.0, c))
    scene = ol.make_scene(allp, mats, light_radiance=meta["light_radiance"], max_depth=max_depth, env=env,
                          env_rotate_uvw=uvw, textures=[checker_texture()])
    return scene, meta


def displaced_sphere(center, radius, mat, nu, nv, amp=0.08, freq=9.0):
    """dense smooth-shaded blob: 2*nu*(nv-1) triangles, vectorised (used for the 250k-triangle stress scene)"""
    c = np.asarray(center, np.float32)
    j, i = np.meshgrid(np.arange(nv + 1, dtype=np.float32), np.arange(nu + 1, dtype=np.float32), indexing="ij")
    th = np.float32(np.pi) * j / np.float32(nv)
    ph = np.float32(2 * np.pi) * i / np.float32(nu)
    d = np.stack([np.sin(th) * np.cos(ph), np.cos(th), np.sin(th) * np.sin(ph)], -1).astype(np.float32)
    r = (np.float32(radius) * (1 + np.float32(amp) * np.sin(np.float32(freq) * th) * np.cos(np.float32(freq) * ph))).astype(np.float32)
    pos = (c + d * r[..., None]).astype(np.float32)
    nrm = d / np.sqrt((d * d).sum(-1, keepdims=True), dtype=np.float32)
    uv = np.stack([i / np.float32(nu), j / np.float32(nv)], -1).astype(np.float32)
    quads = [(jj, ii) for jj in range(nv) for ii in range(nu)]
    tris = []
    for jj, ii in quads:
        a, b, cc, dd = (jj, ii), (jj, ii + 1), (jj + 1, ii + 1), (jj + 1, ii)
        if jj != 0:
            tris.append((a, b, cc))
        if jj != nv - 1:
            tris.append((a, cc, dd))
    out = np.zeros(len(tris), dtype=st.PRIMITIVE)
    idx = np.array(tris)            # (T, 3, 2)
    for k, name in enumerate(("v1", "v2", "v3")):
        jj, ii = idx[:, k, 0], idx[:, k, 1]
        for ax, comp in enumerate("xyz"):
            out["triangle"][name]["v"][comp] = pos[jj, ii, ax]
            out["triangle"][name]["n"][comp] = nrm[jj, ii, ax]
        out["triangle"][name]["uv"] = uv[jj, ii]
    out["triangle"]["matIdx"] = mat
    out["triangle"]["bssrdfIdx"] = -1
    out["triangle"]["lightIdx"] = -1
    out["triangle"]["mediumInside"] = -1
    out["triangle"]["mediumOutside"] = -1
    return out


def stress_parts(scale=1.0):
    """Cornell walls + three dense blobs (mirror-ish metal, glass, substrate): ~250k triangles at scale=1
    (stand-in for BASELINE config 5 "sponza": the reference ships no such mesh)."""
    n1 = max(8, int(182 * scale))
    n2 = max(8, int(140 * scale))
    n3 = max(8, int(105 * scale))
    return concat([
        displaced_sphere((-0.35, 0.55, -0.25), 0.45, 13, 2 * n1, n1),
        displaced_sphere((0.45, 0.4, 0.3), 0.35, 7, 2 * n2, n2, amp=0.05, freq=13.0),
        displaced_sphere((-0.1, 1.45, 0.1), 0.3, 10, 2 * n3, n3, amp=0.12, freq=7.0),
    ])


def stress_scene(scale=1.0, max_depth=16, extra=None):
    prims, _, meta = cornell_raw()
    keep = concat([prims[0:10], prims[34:36]])      # walls + light, no boxes
    keep["triangle"]["lightIdx"][10:] = [0, 1]
    allp = concat([keep, stress_parts(scale)] + ([extra] if extra is not None else []))
    scene = ol.make_scene(allp, material_table(), light_radiance=meta["light_radiance"], max_depth=max_depth,
                          textures=[checker_texture()])
    return scene, meta


# ---- BASELINE config 3-5 stand-ins from the reference's shipped meshes (SURVEY.md 8(d) "D-inputs") -------------------------
from standins import (mesh_fixture, write_mesh_obj, write_png_rgba, procedural_sky, STANDIN_MATERIALS, C5_CORE, C5_SPHERE_MATERIALS,
                      write_standin_scene, write_smoke_scene)      # the BASELINE stand-ins live in tests/standins.py (bench.py uses them too)


def write_vol_caustic_scene(directory, sphere="mesh", integrator="vpt", max_depth=17):
    """The reference's scenes/cornell_box/vol_caustic.json - Cornell walls, a glass sphere (centre 0, 1.2, 0, radius 0.3) in a
    scattering gas that a material-less front face closes in - rebuilt from what this repository holds.  Two substitutions: the
    sphere is the json's analytic `"sphere": true` primitive (not supported here), replaced by the shipped sphere.obj (8 064 smooth-
    shaded triangles, radius 0.5) scaled to radius 0.3; and the light is the Cornell light.obj (the json as shipped names a
    5 x 4 mm mesh_6.obj).  A parity scene (glass + homogeneous medium + interface through the loader), NOT a pin: rendered with
    Volpath it converges to a frame 7 - 12 % darker than result/volume_caustic.png, whose own scene file is not in the repository."""
    import shutil
    src = os.path.join(ol.ROOT, "scenes", "cornell_pt", "geometry")
    os.makedirs(os.path.join(directory, "geometry"), exist_ok=True)
    for name in ("floor", "ceil", "back", "left", "right", "light"):
        shutil.copy(os.path.join(src, name + ".obj"), os.path.join(directory, "geometry", name + ".obj"))
    with open(os.path.join(directory, "geometry", "front.obj"), "w") as f:          # mesh_3.obj of the reference: the z = 1 face
        f.write("v -1 2 1\nv -1 0 1\nv 1 0 1\nv 1 2 1\nvn 0 0 1\nvt 0 0\nvt 1 0\nvt 1 1\nvt 0 1\nf 1/1/1 2/2/1 3/3/1\nf 1/1/1 3/3/1 4/4/1\n")
    write_mesh_obj(os.path.join(directory, "geometry", "sphere.obj"), "sphere")
    js = {"screen_width": 512, "screen_height": 512, "integrator": integrator, "maxDepth": max_depth, "epsilon": 0.001,
          "camera": {"position": [0, 1.0, 6.8], "lookat": [0, 1.0, 0], "fov": 19.5, "apertureRadius": 0.0, "focalDistance": 7.0},
          "material": [{"name": "Left", "bsdf": "lambertian", "diffuse": [0.63, 0.065, 0.05]},
                       {"name": "Right", "bsdf": "lambertian", "diffuse": [0.14, 0.45, 0.091]},
                       {"name": "General", "bsdf": "lambertian", "diffuse": [0.725, 0.725, 0.725]},
                       {"name": "Emission", "bsdf": "lambertian", "diffuse": [0, 0, 0]},
                       {"name": "Glass", "bsdf": "dielectric", "insideIOR": 1.5, "outsideIOR": 1.0}],
          "medium": [{"type": "homogeneous", "sigmaA": [0, 0, 0], "sigmaS": [1.0, 1.0, 1.0], "scale": 1.0, "name": "gas"}],
          "scene": [{"mesh": "geometry/floor.obj", "material": "General"}, {"mesh": "geometry/ceil.obj", "material": "General"},
                    {"mesh": "geometry/back.obj", "material": "General"},
                    {"mesh": "geometry/front.obj", "material": "", "inside": "gas", "outside": ""},
                    {"mesh": "geometry/right.obj", "material": "Right"}, {"mesh": "geometry/left.obj", "material": "Left"},
                    {"mesh": "geometry/sphere.obj", "material": "Glass", "inside": "", "outside": "gas",
                     "scale": [0.6, 0.6, 0.6], "translate": [0, 1.2, 0]}],
          "light": [{"mesh": "geometry/light.obj", "material": "Emission", "radiance": [17.0, 12.0, 4.0]}]}
    path = os.path.join(directory, "scene.json")
    json.dump(js, open(path, "w"), indent=1)
    return path
