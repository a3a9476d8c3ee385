"""tests/golden/meshes.npz: the meshes of the reference's only shipped scene directory that SURVEY.md 8(d) names as the raw
material of the BASELINE config 3-5 stand-ins ("D-inputs"): sphere (8 064 triangles, vn + vt), cube-subdiv (1 536 quads, vt, no
vn), dragon (100 000 triangles), bunny2 (69 666), teapot (6 320) - the last three without vn / vt, i.e. they exercise the
loader's smooth-normal rule.  Input DATA of the reference kept as a compressed fixture (positions / normals / uvs as the
float32 the text parses to, faces as index arrays), so that the GPU box, which has no /root/reference, can write the OBJ files
back (tests/scenes.py: write_mesh_obj) and run them through the product loader; tests/test_standins.py checks, where
/root/reference exists, that the written files load to the same triangles as the shipped ones.
Run where /root/reference exists:  python tools/bake_d_inputs.py"""
import os, sys
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = "/root/reference/scenes/cornell_box/geometry"
NAMES = ["sphere", "cube-subdiv", "dragon", "bunny2", "teapot"]


def parse_obj(path):
    v, vn, vt, counts, fv, fvt, fvn = [], [], [], [], [], [], []
    for line in open(path, errors="replace"):
        t = line.split()
        if not t:
            continue
        if t[0] == "v":
            v.append([np.float32(float(x)) for x in t[1:4]])
        elif t[0] == "vn":
            vn.append([np.float32(float(x)) for x in t[1:4]])
        elif t[0] == "vt":
            vt.append([np.float32(float(x)) for x in t[1:3]])
        elif t[0] == "f":
            counts.append(len(t) - 1)
            for c in t[1:]:
                idx = (c.split("/") + ["", ""])[:3]
                fv.append(int(idx[0]))
                fvt.append(int(idx[1]) if idx[1] else 0)
                fvn.append(int(idx[2]) if idx[2] else 0)
    return {"v": np.array(v, np.float32).reshape(-1, 3), "vn": np.array(vn, np.float32).reshape(-1, 3),
            "vt": np.array(vt, np.float32).reshape(-1, 2), "counts": np.array(counts, np.int32),
            "fv": np.array(fv, np.int32), "fvt": np.array(fvt, np.int32), "fvn": np.array(fvn, np.int32)}


if __name__ == "__main__":
    out = {}
    for name in NAMES:
        m = parse_obj(os.path.join(SRC, name + ".obj"))
        assert m["fv"].min() >= 1, "negative (relative) indices are not used by these files"
        tris = int((m["counts"] - 2).sum())
        print(f"{name:12s} v {len(m['v']):6d} vn {len(m['vn']):6d} vt {len(m['vt']):6d} faces {len(m['counts']):6d} -> {tris} triangles")
        for k, a in m.items():
            out[name.replace("-", "_") + "__" + k] = a
    path = os.path.join(ROOT, "tests", "golden", "meshes.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")
