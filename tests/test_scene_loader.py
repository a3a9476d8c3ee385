"""C++ scene loader (JSON + OBJ + PNG/PFM + bvh.cache) — host code only, no GPU."""
import json
import os
import shutil
import struct
import zlib

import numpy as np
import pytest

import oracle_lib as ol
import scenes
from gpu_pathtracer_amd import api, scene_types as st

SCENE = os.path.join(ol.ROOT, "scenes", "cornell_pt", "scene.json")


def tri_fields_equal(a, b):
    ok = True
    for v in ("v1", "v2", "v3"):
        for f in ("v", "n"):
            for c in "xyz":
                ok &= np.array_equal(a[v][f][c].view(np.uint32), b[v][f][c].view(np.uint32))
        ok &= np.array_equal(a[v]["uv"], b[v]["uv"])
    for f in ("matIdx", "lightIdx", "bssrdfIdx", "mediumInside", "mediumOutside"):
        ok &= np.array_equal(a[f], b[f])
    return bool(ok)


def test_cornell_scene_file_equals_baked_fixture():
    ls = api.LoadedScene(SCENE)
    scene, meta = ol.load_cornell(8)
    d = ls.desc
    assert (d.n_prims, d.n_nodes, d.n_materials, d.n_lights, d.n_light_distribution) == (36, 27, 8, 2, 3)
    assert d.integrator_type == st.IT_PT and d.max_depth == 8
    assert tri_fields_equal(ls.array("prims", "n_prims", st.PRIMITIVE)["triangle"], scene.prims["triangle"])
    assert ls.array("materials", "n_materials", st.MATERIAL).tobytes() == scene.materials.tobytes()
    nodes = ls.array("nodes", "n_nodes", st.BVH_NODE)
    assert all(np.array_equal(nodes[f], scene.nodes[f]) for f in nodes.dtype.names)
    assert ls.array("light_distribution", "n_light_distribution", np.float32).tolist() == [0.0, 0.5, 1.0]
    lights = ls.array("lights", "n_lights", st.AREA)
    assert tri_fields_equal(lights["triangle"], scene.lights["triangle"])
    assert [float(lights["radiance"][0][c]) for c in "xyz"] == [17.0, 12.0, 4.0]
    assert (ls.width, ls.height) == (512, 512) and abs(ls.epsilon - 0.001) < 1e-9
    assert bytes(ls.camera) == bytes(ol.cornell_camera(meta, 512, 512))


def write_png(path, img):
    """8-bit RGBA PNG, rows top-down (test helper: zlib from the standard library)"""
    h, w, _ = img.shape
    raw = b"".join(b"\x00" + img[y].tobytes() for y in range(h))

    def chunk(t, d):
        return struct.pack(">I", len(d)) + t + d + struct.pack(">I", zlib.crc32(t + d) & 0xffffffff)
    open(path, "wb").write(b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 6, 0, 0, 0)) +
                           chunk(b"IDAT", zlib.compress(raw, 9)) + chunk(b"IEND", b""))


@pytest.fixture()
def scene_dir(tmp_path):
    d = tmp_path / "scene"
    shutil.copytree(os.path.dirname(SCENE), d)
    return d


def test_transform_defaults_textures_env_and_materials(scene_dir):
    rng = np.random.default_rng(5)
    tex = rng.integers(0, 256, (6, 5, 4), dtype=np.uint8)
    write_png(scene_dir / "checker.png", tex)
    env = scenes.sky_env(16, 8)
    # PFM is bottom-up on disk; the loader hands rows top-down like the reference's EXR reader
    api.save_pfm(str(scene_dir / "sky.pfm"), 16, 8, env[::-1].copy())
    js = json.load(open(scene_dir / "scene.json"))
    js["material"] += [
        {"name": "tex", "bsdf": "substrate", "diffuse": "checker.png", "alpha": 0.1, "remap": True, "specular": [0.04, 0.04, 0.04]},
        {"name": "aniso", "bsdf": "roughconduct", "alphaU": 0.0025, "alphaV": 0.25, "remap": True, "eta": [0.2, 0.9, 1.1], "k": [3.9, 2.4, 2.1]},
        {"name": "unknown-bsdf", "bsdf": "velvet"},
    ]
    js["scene"][5]["material"] = "tex"
    js["scene"][6].update({"material": "aniso", "scale": [0.5, 0.5, 0.5], "translate": [0.1, 0.0, -0.2], "rotate": [0, 30, 10]})
    js["light"].append({"infinite": "sky.pfm", "rotate": [0, 30, 0]})
    js.pop("screen_width")
    json.dump(js, open(scene_dir / "scene.json", "w"))
    ls = api.LoadedScene(str(scene_dir / "scene.json"))
    d = ls.desc
    assert (ls.width, ls.height) == (512, 512)             # default when either size key is missing
    mats = ls.array("materials", "n_materials", st.MATERIAL)
    assert d.n_materials == 11 and d.n_textures == 1
    assert mats[8]["type"] == st.MT_SUBSTRATE and mats[8]["textureIdx"] == 0
    x = np.log(np.float32(0.1))
    remap = np.float32(1.62142) + np.float32(0.819955) * x + np.float32(0.1734) * x * x + np.float32(0.0171201) * x * x * x + np.float32(0.000640711) * x * x * x * x
    assert abs(float(mats[8]["alphaU"]) - float(remap)) < 1e-6 and mats[8]["alphaU"] == mats[8]["alphaV"]
    assert mats[9]["alphaU"] != mats[9]["alphaV"]
    assert mats[10]["type"] == st.MT_LAMBERTIAN               # unknown bsdf name -> 0, as std::map::operator[]
    # texture: flipped vertically, sRGB->linear pow 2.2, truncated to 8 bit (reference imageio.cpp:11-59, texture.h:21-25)
    import ctypes as C
    trec = C.cast(d.textures, C.POINTER(st.Texture))[0]
    assert (trec.width, trec.height) == (5, 6)
    got = np.ctypeslib.as_array(C.cast(trec.data, C.POINTER(C.c_uint8)), shape=(6, 5, 4))
    lin = np.power((tex[::-1].astype(np.float32) * np.float32(1 / 255.0))[..., :3], np.float32(2.2), dtype=np.float32)
    want = np.concatenate([(lin * np.float32(255)).astype(np.uint8), tex[::-1][..., 3:]], -1)
    assert np.abs(got.astype(int) - want.astype(int)).max() <= 1
    # transformed mesh: the tall box is scaled/rotated/translated; normals stay unit
    prims = ls.array("prims", "n_prims", st.PRIMITIVE)
    moved = prims[prims["triangle"]["matIdx"] == 9]["triangle"]
    assert len(moved) == 12
    n = np.stack([moved["v1"]["n"][c] for c in "xyz"], -1)
    assert np.allclose((n ** 2).sum(-1), 1.0, atol=1e-6)
    ys = np.concatenate([moved[v]["v"]["y"] for v in ("v1", "v2", "v3")])
    assert abs(ys.max() - 0.6 * 1.0) < 0.2 and ys.min() > -0.2
    # env light: cdf gets one more entry, bounding sphere from the scene box, u/v/w from "rotate"
    assert d.n_light_distribution == 4
    inf = C.cast(d.infinite, C.POINTER(st.Infinite))[0]
    assert inf.isvalid and (inf.width, inf.height) == (16, 8) and inf.radius > 1.0
    assert abs(inf.u.x - np.cos(np.pi / 6)) < 1e-6 and abs(inf.v.y - 1) < 1e-6
    data = np.ctypeslib.as_array(C.cast(inf.data, C.POINTER(C.c_float)), shape=(8, 16, 3))
    assert np.array_equal(data, env)


def test_loader_errors(tmp_path, scene_dir):
    lib = api.load()
    with pytest.raises(api.GptError) as e:
        api.LoadedScene(str(tmp_path / "missing.json"))
    assert "is not good" in str(e.value)
    (tmp_path / "bad.json").write_text('{"camera": {"position": [0,0,0]}, "scene": [')
    with pytest.raises(api.GptError) as e:
        api.LoadedScene(str(tmp_path / "bad.json"))
    assert "Parse scene error" in str(e.value) and "gpt error -6" in str(e.value)
    (tmp_path / "nocam.json").write_text('{"scene": []}')
    with pytest.raises(api.GptError) as e:
        api.LoadedScene(str(tmp_path / "nocam.json"))
    assert "must define camera" in str(e.value)
    js = json.load(open(scene_dir / "scene.json"))
    js["scene"][0]["material"] = "DoesNotExist"
    json.dump(js, open(scene_dir / "scene.json", "w"))
    with pytest.raises(api.GptError) as e:
        api.LoadedScene(str(scene_dir / "scene.json"))
    assert "no material named" in str(e.value)
    js["scene"][0] = {"sphere": True, "material": "General"}
    json.dump(js, open(scene_dir / "scene.json", "w"))
    with pytest.raises(api.GptError) as e:
        api.LoadedScene(str(scene_dir / "scene.json"))
    assert "sphere" in str(e.value)


def test_shipped_vpt_scene_settings_are_refused_at_begin(scene_dir):
    """The reference's own cornell json is a "vpt" scene; the loader accepts the key, the renderer refuses it."""
    js = json.load(open(scene_dir / "scene.json"))
    js["integrator"] = "vpt"
    js["maxDepth"] = 17
    json.dump(js, open(scene_dir / "scene.json", "w"))
    ls = api.LoadedScene(str(scene_dir / "scene.json"))
    assert ls.desc.integrator_type == 2 and ls.desc.max_depth == 17
    ls.set_integrator(st.IT_PT, 8)
    assert ls.desc.integrator_type == st.IT_PT and ls.desc.max_depth == 8


def test_png_writer_follows_savepng(tmp_path):
    """flip Y, clamp, truncate (reference src/imageio.cpp:61-78)"""
    from PIL import Image
    w, h = 7, 5
    rng = np.random.default_rng(2)
    img = (rng.random((h, w, 3)) * 1.4 - 0.2).astype(np.float32)
    api.save_png(str(tmp_path / "o.png"), w, h, img)
    got = np.asarray(Image.open(tmp_path / "o.png"))
    want = (np.clip(img[::-1], 0, 1) * np.float32(255)).astype(np.uint8)
    assert np.array_equal(got, want)
