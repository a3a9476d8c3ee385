"""Mesh placement and the environment light's frame against the glm the reference vendors.

src/parsescene.cpp:349-355 builds t * r * s with glm, src/mesh.cpp:49-57 moves vertices and normals with it, and
src/parsescene.cpp:552-569 turns the environment light with it.  glm is header-only and compiles with g++ where it lies
(oracle/ref_transform.cpp -> oracle/_ref/libref_transform.so), so the scene loader's own matrix code
(gpu_pathtracer_amd/csrc/scene_loader.cpp) is compared with glm's results bit for bit, through the loader's public entry
(gpt_scene_load on a scene file), the way a scene reaches the renderer.
"""
import ctypes as C
import json
import os
import shutil
import subprocess

import numpy as np
import pytest

import oracle_lib as ol
from gpu_pathtracer_amd import api, scene_types as st

ROOT = ol.ROOT
REF_LIB = os.path.join(ROOT, "oracle", "_ref", "libref_transform.so")
SCENE = os.path.join(ROOT, "scenes", "cornell_pt")


@pytest.fixture(scope="module")
def ref():
    if not os.path.exists(REF_LIB) and os.path.isdir("/root/reference"):
        subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "ref"], check=True)
    if not os.path.exists(REF_LIB):
        pytest.skip("oracle/_ref/libref_transform.so is not built and /root/reference is not here to build it from")
    lib = C.CDLL(REF_LIB)
    for name in ("ref_mesh_trs", "ref_transform_vertices", "ref_infinite_frame"):
        getattr(lib, name).restype = None
    return lib


def f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


CASES = [dict(), dict(scale=[0.5, 0.5, 0.5]), dict(translate=[0.1, -2.5, 1e-3]), dict(rotate=[0, 30, 0]), dict(rotate=[90, 180, 270]),
         dict(scale=[0.08, 0.08, 0.08], translate=[-0.35, 0, -0.3], rotate=[0, 30, 0]),                 # the C5 stand-in's dragon
         dict(scale=[1.7, 0.3, -2.0], translate=[3, 4, 5], rotate=[12.5, -47.25, 333.0]),               # anisotropic, mirrored
         dict(scale=[1e-3, 1e3, 1], rotate=[0.001, 359.999, 45]), dict(rotate=[-0.0, 1e-6, 720.5], translate=[1e6, -1e-6, 0])]


@pytest.mark.parametrize("case", range(len(CASES)))
def test_mesh_vertices_and_normals_equal_glm(ref, tmp_path, case):
    kw = CASES[case]
    rng = np.random.default_rng(case)
    n_tri = 40
    verts = f32(rng.uniform(-1, 1, (3 * n_tri, 3)))
    normals = rng.normal(0, 1, (3 * n_tri, 3))
    normals = f32(normals / np.linalg.norm(normals, axis=1, keepdims=True))
    d = tmp_path / "scene"
    shutil.copytree(SCENE, d)
    with open(d / "geometry" / "cloud.obj", "w") as f:
        for v in verts:
            f.write("v %.9g %.9g %.9g\n" % tuple(v))
        for n in normals:
            f.write("vn %.9g %.9g %.9g\n" % tuple(n))
        for i in range(3 * n_tri):
            f.write("vt %.9g 0.25\n" % (i / 256.0))            # the corner's number, to find it again after the BVH reorders
        for t in range(n_tri):
            f.write("f %d/%d/%d %d/%d/%d %d/%d/%d\n" % tuple(np.repeat(3 * t + np.arange(1, 4), 3)))
    js = json.load(open(d / "scene.json"))
    js["material"].append({"name": "cloud", "bsdf": "lambertian", "diffuse": [0.5, 0.5, 0.5]})
    unit = {"mesh": "geometry/cloud.obj", "material": "cloud"}
    unit.update(kw)
    js["scene"].append(unit)
    json.dump(js, open(d / "scene.json", "w"))
    ls = api.LoadedScene(str(d / "scene.json"))
    prims = ls.array("prims", "n_prims", st.PRIMITIVE)
    tri = prims[prims["triangle"]["matIdx"] == len(js["material"]) - 1]["triangle"]
    assert len(tri) == n_tri
    trs = np.zeros(16, np.float32)
    ref.ref_mesh_trs(f32(kw.get("scale", [1, 1, 1])).ctypes.data_as(C.c_void_p), f32(kw.get("translate", [0, 0, 0])).ctypes.data_as(C.c_void_p),
                     f32(kw.get("rotate", [0, 0, 0])).ctypes.data_as(C.c_void_p), trs.ctypes.data_as(C.c_void_p))
    v_want, n_want = np.zeros_like(verts), np.zeros_like(normals)
    ref.ref_transform_vertices(trs.ctypes.data_as(C.c_void_p), 3 * n_tri, verts.ctypes.data_as(C.c_void_p), normals.ctypes.data_as(C.c_void_p),
                               v_want.ctypes.data_as(C.c_void_p), n_want.ctypes.data_as(C.c_void_p))
    seen = 0
    for corner in ("v1", "v2", "v3"):
        idx = np.rint(tri[corner]["uv"][:, 0] * 256).astype(int)
        got_v = np.stack([tri[corner]["v"][c] for c in "xyz"], -1)
        got_n = np.stack([tri[corner]["n"][c] for c in "xyz"], -1)
        assert np.array_equal(got_v.view(np.uint32), v_want[idx].view(np.uint32)), kw
        assert np.array_equal(got_n.view(np.uint32), n_want[idx].view(np.uint32)), kw
        seen += len(idx)
    assert seen == 3 * n_tri


def load_with_env(tmp_path, name, light):
    d = tmp_path / name
    shutil.copytree(SCENE, d)
    api.save_pfm(str(d / "sky.pfm"), 4, 2, np.ones((2, 4, 3), np.float32))
    js = json.load(open(d / "scene.json"))
    light = dict(light)
    light["infinite"] = "sky.pfm"
    js["light"].append(light)
    json.dump(js, open(d / "scene.json", "w"))
    ls = api.LoadedScene(str(d / "scene.json"))
    inf = C.cast(ls.desc.infinite, C.POINTER(st.Infinite))[0]
    return np.array([[getattr(getattr(inf, a), c) for c in "xyz"] for a in "uvw"], np.float32)


def test_environment_light_frame_equals_glm(ref, tmp_path):
    rng = np.random.default_rng(21)
    for i, rot in enumerate(([0, 30, 0], [10, 20, 30], [-95.5, 181.25, 0.001], [360, 720, -360])):
        want = np.zeros(9, np.float32)
        ref.ref_infinite_frame(f32(rot).ctypes.data_as(C.c_void_p), None, want.ctypes.data_as(C.c_void_p))
        got = load_with_env(tmp_path, f"rot{i}", {"rotate": rot})
        assert np.array_equal(got.reshape(-1).view(np.uint32), want.view(np.uint32)), rot
    for i in range(4):
        m = np.eye(4) + rng.normal(0, 0.4, (4, 4))
        m16 = [float(np.float32(x)) for x in m.reshape(-1)]
        want = np.zeros(9, np.float32)
        ref.ref_infinite_frame(None, f32(m16).ctypes.data_as(C.c_void_p), want.ctypes.data_as(C.c_void_p))
        got = load_with_env(tmp_path, f"mat{i}", {"matrix": m16})
        assert np.array_equal(got.reshape(-1).view(np.uint32), want.view(np.uint32)), m16
