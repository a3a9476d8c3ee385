"""numpy / ctypes mirrors of the records in include/gpt_types.h.

Plumbing only: lets Python (tests, bench.py) build scene arrays in the
reference's own struct layouts (reference src/mesh.h:13-26,
src/primitive.h:15-23, src/bvh.h:19-29, src/material.h:19-27, src/area.h:7-11,
src/infinite.h:6-13, src/camera.h:8-26) and hand them across the C ABI.
"""
import ctypes as C

import numpy as np

F3 = np.dtype([("x", "<f4"), ("y", "<f4"), ("z", "<f4")])

VERTEX = np.dtype({
    "names": ["v", "n", "uv", "t"],
    "formats": [F3, F3, ("<f4", (2,)), F3],
    "offsets": [0, 12, 24, 32],
    "itemsize": 48,
})

TRIANGLE = np.dtype({
    "names": ["v1", "v2", "v3", "matIdx", "bssrdfIdx", "lightIdx", "mediumInside", "mediumOutside"],
    "formats": [VERTEX, VERTEX, VERTEX, "<i4", "<i4", "<i4", "<i4", "<i4"],
    "offsets": [0, 48, 96, 144, 148, 152, 156, 160],
    "itemsize": 168,
})

PRIMITIVE = np.dtype({
    "names": ["type", "triangle"],
    "formats": ["<i4", TRIANGLE],
    "offsets": [0, 8],
    "itemsize": 176,
})

BVH_NODE = np.dtype({
    "names": ["fmin", "fmax", "second_child_offset", "is_leaf", "start", "end"],
    "formats": [F3, F3, "<i4", "u1", "<i4", "<i4"],
    "offsets": [0, 12, 24, 28, 32, 36],
    "itemsize": 40,
})

MATERIAL = np.dtype({
    "names": ["type", "alphaU", "alphaV", "insideIOR", "outsideIOR", "k", "eta", "diffuse", "specular", "textureIdx"],
    "formats": ["<i4", "<f4", "<f4", "<f4", "<f4", F3, F3, F3, F3, "<i4"],
    "offsets": [0, 4, 8, 12, 16, 20, 32, 44, 56, 68],
    "itemsize": 72,
})

AREA = np.dtype({
    "names": ["radiance", "triangle", "medium"],
    "formats": [F3, TRIANGLE, "<i4"],
    "offsets": [0, 16, 184],
    "itemsize": 192,
})

MT_LAMBERTIAN, MT_MIRROR, MT_DIELECTRIC, MT_ROUGHDIELECTRIC, MT_ROUGHCONDUCTOR, MT_SUBSTRATE = range(6)
IT_AO = 0
IT_PT = 1
IT_VPT = 2


class Float3(C.Structure):
    _fields_ = [("x", C.c_float), ("y", C.c_float), ("z", C.c_float)]


class Float2(C.Structure):
    _fields_ = [("x", C.c_float), ("y", C.c_float)]


# gpt_medium (104 bytes): type, g, then the union at offset 8 - the coefficients sit at the same offsets in both views,
# the density grid (src/medium.h:53-62) follows them
MEDIUM = np.dtype({"names": ["type", "g", "sigmaA", "sigmaS", "sigmaT", "nx", "ny", "nz", "density", "invMaxDensity",
                             "p0", "p1", "iterMax", "evalTransmittanceType"],
                   "formats": [np.int32, np.float32, F3, F3, F3, np.int32, np.int32, np.int32, np.uint64, np.float32,
                               F3, F3, np.int32, np.int32],
                   "offsets": [0, 4, 8, 20, 32, 44, 48, 52, 56, 64, 68, 80, 92, 96], "itemsize": 104})


def make_medium(sigma_a, sigma_s, g=0.0, scale=1.0):
    """homogeneous medium as parsescene.cpp:86-98 builds it: sigmaA, sigmaS scaled, sigmaT = their sum"""
    m = np.zeros((), dtype=MEDIUM)
    a = (np.asarray(sigma_a, np.float32) * np.float32(scale)).astype(np.float32)
    s_ = (np.asarray(sigma_s, np.float32) * np.float32(scale)).astype(np.float32)
    m["type"], m["g"] = 0, np.float32(g)
    m["sigmaA"], m["sigmaS"], m["sigmaT"] = f3(a), f3(s_), f3((a + s_).astype(np.float32))
    return m


def make_het_medium(sigma_a, sigma_s, grid, p0, p1, iter_max=1000, tr_type=1, g=0.0, scale=1.0):
    """heterogeneous medium as parsescene.cpp:99-132 builds it.  `grid` is a C-contiguous float32 array of shape
    (nz, ny, nx) (x fastest, medium.h:176-181) that the CALLER keeps alive; sigmaA + sigmaS must be grey."""
    assert grid.dtype == np.float32 and grid.ndim == 3 and grid.flags["C_CONTIGUOUS"]
    m = make_medium(sigma_a, sigma_s, g, scale)
    m["type"] = 1
    m["nz"], m["ny"], m["nx"] = grid.shape
    m["density"] = grid.ctypes.data
    m["invMaxDensity"] = np.float32(1.0) / np.float32(max(np.float32(0), grid.max()))
    m["p0"], m["p1"] = f3(np.asarray(p0, np.float32)), f3(np.asarray(p1, np.float32))
    m["iterMax"], m["evalTransmittanceType"] = iter_max, tr_type
    return m


class Infinite(C.Structure):
    _fields_ = [
        ("data", C.c_void_p), ("width", C.c_int32), ("height", C.c_int32),
        ("center", Float3), ("radius", C.c_float),
        ("u", Float3), ("v", Float3), ("w", Float3),
        ("isvalid", C.c_uint8), ("_pad", C.c_uint8 * 3),
    ]


class Camera(C.Structure):
    _fields_ = [
        ("position", Float3), ("u", Float3), ("v", Float3), ("w", Float3),
        ("resolution", Float2), ("distance", C.c_float), ("fov", C.c_float),
        ("apertureRadius", C.c_float), ("focalDistance", C.c_float),
        ("filmic", C.c_uint8), ("environment", C.c_uint8), ("_pad", C.c_uint8 * 2),
        ("medium", C.c_int32),
        ("width", C.c_float), ("height", C.c_float), ("pixel2screen", Float2),
        ("ratio", C.c_float), ("area", C.c_float),
    ]


class Texture(C.Structure):
    _fields_ = [("data", C.c_void_p), ("width", C.c_int32), ("height", C.c_int32)]


class SceneDesc(C.Structure):
    _fields_ = [
        ("prims", C.c_void_p), ("n_prims", C.c_int32),
        ("nodes", C.c_void_p), ("n_nodes", C.c_int32),
        ("materials", C.c_void_p), ("n_materials", C.c_int32),
        ("lights", C.c_void_p), ("n_lights", C.c_int32),
        ("light_distribution", C.c_void_p), ("n_light_distribution", C.c_int32),
        ("infinite", C.c_void_p),
        ("textures", C.c_void_p), ("n_textures", C.c_int32),
        ("integrator_type", C.c_int32), ("max_depth", C.c_int32),     # max_depth shares storage with max_dist (ao)
        ("mediums", C.c_void_p), ("n_mediums", C.c_int32),
    ]

    def set_integrator(self, kind, value):
        """kind "pt": value = maxDepth (int); kind "ao": value = maxDist (float) - the reference's union."""
        import struct
        if kind == "pt":
            self.integrator_type, self.max_depth = IT_PT, int(value)
        elif kind == "vpt":
            self.integrator_type, self.max_depth = IT_VPT, int(value)
        elif kind == "ao":
            self.integrator_type = IT_AO
            self.max_depth = struct.unpack("<i", struct.pack("<f", float(value)))[0]
        else:
            raise ValueError(f"integrator {kind!r} is not supported (pt, vpt, ao)")


assert C.sizeof(Infinite) == 72 and Infinite.center.offset == 16 and Infinite.isvalid.offset == 68
assert C.sizeof(Camera) == 104 and Camera.resolution.offset == 48 and Camera.pixel2screen.offset == 88
assert C.sizeof(Texture) == 16


def f3(a):
    """float32 triple -> 0-d F3 record."""
    r = np.zeros((), dtype=F3)
    r["x"], r["y"], r["z"] = np.float32(a[0]), np.float32(a[1]), np.float32(a[2])
    return r


def make_material(mtype=MT_LAMBERTIAN, diffuse=(1, 1, 1), specular=(1, 1, 1), alphaU=0.01, alphaV=0.01,
                  insideIOR=1.0, outsideIOR=1.0, k=(0, 0, 0), eta=(0, 0, 0), textureIdx=-1):
    """Defaults follow the loader (reference src/parsescene.cpp:273-299)."""
    m = np.zeros((), dtype=MATERIAL)
    m["type"] = mtype
    m["alphaU"], m["alphaV"] = alphaU, alphaV
    m["insideIOR"], m["outsideIOR"] = insideIOR, outsideIOR
    m["k"], m["eta"], m["diffuse"], m["specular"] = f3(k), f3(eta), f3(diffuse), f3(specular)
    m["textureIdx"] = textureIdx
    return m


def ptr(arr):
    return arr.ctypes.data_as(C.c_void_p) if arr is not None and arr.size else None
