#!/usr/bin/env python3
"""Bake the reference's Cornell-box geometry into a small triangle-soup fixture.

Runs only in the build container (reads /root/reference, which does not exist
on the GPU box).  Output: tests/golden/cornell_pt.npz = the `cornell_pt` scene of
SURVEY.md §8(d) "D-inputs" C1/C2 — meshes floor, ceil, back, left, right, short,
tall, then light — as the `Primitive` array the reference's loader would hand to
Scene::Init (reference src/parsescene.cpp:333-392,492-541, src/mesh.cpp:29-91),
BEFORE BVH reordering, plus the material table and camera/light parameters of
scenes/cornell_box/scene.json:7-61,111-117.

OBJ reading rule (assimp with aiProcess_Triangulate on files that carry vn/vt):
one vertex per face corner, fan triangulation, identity transform, normals
re-normalised as n * (1/sqrt(dot(n,n))) in float32 (glm::normalize,
src/mesh.cpp:56).  The fixture is data (vertex positions / normals / uvs); no
reference source text is stored.
"""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from gpu_pathtracer_amd import scene_types as st  # noqa: E402

REF = "/root/reference/scenes/cornell_box"


def read_obj(path):
    v, vn, vt, faces = [], [], [], []
    with open(path) as f:
        for line in f:
            p = line.split()
            if not p:
                continue
            if p[0] == "v":
                v.append([np.float32(x) for x in p[1:4]])
            elif p[0] == "vn":
                vn.append([np.float32(x) for x in p[1:4]])
            elif p[0] == "vt":
                vt.append([np.float32(x) for x in p[1:3]])
            elif p[0] == "f":
                corners = []
                for c in p[1:]:
                    idx = (c.split("/") + ["", ""])[:3]
                    corners.append(tuple(int(i) if i else 0 for i in idx))
                for k in range(1, len(corners) - 1):
                    faces.append((corners[0], corners[k], corners[k + 1]))
    return v, vn, vt, faces


def normalize32(n):
    n = np.asarray(n, dtype=np.float32)
    d = np.float32(n[0] * n[0]) + np.float32(n[1] * n[1])
    d = np.float32(d + np.float32(n[2] * n[2]))
    inv = np.float32(1.0) / np.sqrt(d, dtype=np.float32)
    return (n * inv).astype(np.float32)


RAW_NORMALS = []   # OBJ normals before normalisation, one (3,3) block per triangle, for tools/export_cornell_scene.py


def mesh_prims(path, mat_idx, light_base=None):
    v, vn, vt, faces = read_obj(path)
    prims = np.zeros(len(faces), dtype=st.PRIMITIVE)
    for face in faces:
        RAW_NORMALS.append(np.array([vn[c[2] - 1] for c in face], dtype=np.float32))
    for i, face in enumerate(faces):
        tri = prims[i]["triangle"]
        for name, (iv, it, inn) in zip(("v1", "v2", "v3"), face):
            vert = tri[name]
            vert["v"] = st.f3(v[iv - 1])
            vert["n"] = st.f3(normalize32(vn[inn - 1]))
            vert["uv"] = np.asarray(vt[it - 1], dtype=np.float32) if it else np.zeros(2, np.float32)
        tri["matIdx"] = mat_idx
        tri["bssrdfIdx"] = -1
        tri["lightIdx"] = -1 if light_base is None else light_base + i
        tri["mediumInside"] = -1
        tri["mediumOutside"] = -1
        prims[i]["type"] = 0
    return prims


def main():
    # material table of scenes/cornell_box/scene.json:16-61 (index 3 is the dead duplicate "General")
    mats = np.zeros(8, dtype=st.MATERIAL)
    mats[0] = st.make_material(diffuse=(0.63, 0.065, 0.05))            # Left
    mats[1] = st.make_material(diffuse=(0.14, 0.45, 0.091))            # Right
    mats[2] = st.make_material(diffuse=(0.725, 0.725, 0.725))          # General
    mats[3] = st.make_material(diffuse=(0.725, 0.725, 0.725))          # General (dup)
    mats[4] = st.make_material(diffuse=(0, 0, 0))                      # Emission
    mats[5] = st.make_material(st.MT_MIRROR)                           # Mirror
    mats[6] = st.make_material(st.MT_ROUGHCONDUCTOR, alphaU=0.025, alphaV=0.025, eta=(1, 1, 1), k=(1, 1, 1))
    mats[7] = st.make_material(st.MT_DIELECTRIC, insideIOR=1.5, outsideIOR=1.0)

    parts = []
    for name, m in (("floor", 2), ("ceil", 2), ("back", 2), ("left", 0), ("right", 1), ("short", 2), ("tall", 2)):
        parts.append(mesh_prims(f"{REF}/geometry/{name}.obj", m))
    parts.append(mesh_prims(f"{REF}/geometry/light.obj", 4, light_base=0))
    prims = np.zeros(sum(len(p) for p in parts), dtype=st.PRIMITIVE)   # np.concatenate would re-pack the records
    o = 0
    for p in parts:
        prims[o:o + len(p)] = p
        o += len(p)
    assert prims.dtype.itemsize == 176 and mats.dtype.itemsize == 72
    meta = {
        "camera": {"position": [0, 1.0, 6.8], "lookat": [0, 1.0, 0], "up": [0, 1, 0], "fov": 19.5,
                   "apertureRadius": 0.0, "focalDistance": 7.0, "distance": 0.1, "filmic": True},
        "epsilon": 0.001,
        "light_radiance": [17.0, 12.0, 4.0],
        "n_light_prims": int(len(parts[-1])),
    }
    out = os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "cornell_pt.npz")
    # raw bytes: np.save would re-pack the padded record layouts
    np.savez_compressed(out, prims=np.frombuffer(prims.tobytes(), np.uint8),
                        materials=np.frombuffer(mats.tobytes(), np.uint8), meta=json.dumps(meta),
                        raw_normals=np.stack(RAW_NORMALS))
    print("wrote", out, len(prims), "prims")


if __name__ == "__main__":
    main()
