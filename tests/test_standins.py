"""The BASELINE config 3-5 stand-ins SURVEY.md 8(d) defines from the reference's shipped meshes, through the product loader
(host code only here; the GPU parity tests are in tests/test_gpu_parity.py)."""
import os

import numpy as np
import pytest

import oracle_lib as ol
import scenes
from gpu_pathtracer_amd import api, scene_types as st

REF_GEOMETRY = "/root/reference/scenes/cornell_box/geometry"


@pytest.mark.skipif(not os.path.isdir(REF_GEOMETRY), reason="needs the reference's scene directory (build container only)")
@pytest.mark.parametrize("name", ["sphere", "cube-subdiv", "dragon", "bunny2", "teapot"])
def test_written_mesh_loads_like_the_shipped_file(tmp_path, name):
    """tests/golden/meshes.npz written back as OBJ text is, for the loader, the file the reference ships: the same triangles,
    normals (given or generated) and uvs, bit for bit, in the same order."""
    import json
    import shutil
    out = []
    for variant in ("shipped", "written"):
        d = tmp_path / variant
        os.makedirs(d / "geometry")
        shutil.copy(os.path.join(ol.ROOT, "scenes", "cornell_pt", "geometry", "light.obj"), d / "geometry" / "light.obj")
        if variant == "shipped":
            shutil.copy(os.path.join(REF_GEOMETRY, name + ".obj"), d / "geometry" / (name + ".obj"))
        else:
            scenes.write_mesh_obj(str(d / "geometry" / (name + ".obj")), name)
        js = {"camera": {"position": [0, 1, 6.8], "lookat": [0, 1, 0], "fov": 19.5},
              "material": [{"name": "m", "bsdf": "lambertian", "diffuse": [0.5, 0.5, 0.5]}],
              "scene": [{"mesh": "geometry/%s.obj" % name, "material": "m", "scale": [0.08, 0.08, 0.08], "rotate": [0, 30, 0]}],
              "light": [{"mesh": "geometry/light.obj", "material": "m", "radiance": [1, 1, 1]}]}
        json.dump(js, open(d / "scene.json", "w"))
        ls = api.LoadedScene(str(d / "scene.json"))
        out.append((ls.array("prims", "n_prims", st.PRIMITIVE).tobytes(), ls.array("nodes", "n_nodes", st.BVH_NODE).tobytes()))
        ls.close()
    assert out[0][0] == out[1][0] and out[0][1] == out[1][1]


def test_config5_core_scene_has_the_surveyed_bvh(tmp_path):
    """Cornell walls + dragon (scale 0.08, translate -0.35,0,-0.3, rotate 0,30,0) + bunny2 (scale
    0.06, translate 0.45,0,0.3) + teapot (scale 0.08, translate -0.5,0,0.55) + light = 175 998 primitives -> 112 947 BVH nodes
    with the reference's builder (bvh.cache 35 493 560 bytes)."""
    ls = api.LoadedScene(scenes.write_standin_scene(str(tmp_path / "c5core"), "c5core"))
    assert ls.desc.n_prims == 175998
    assert ls.desc.n_nodes == 112947
    assert 32 + 176 * ls.desc.n_prims + 40 * ls.desc.n_nodes == 35493560
    tri = ls.array("prims", "n_prims", st.PRIMITIVE)["triangle"]
    n = np.stack([tri["v1"]["n"][c] for c in "xyz"], -1)
    assert np.isfinite(n).all() and np.allclose((n * n).sum(-1), 1.0, atol=1e-4)      # generated smooth normals are unit vectors
    ls.close()


def test_standin_scenes_load(tmp_path):
    c3 = api.LoadedScene(scenes.write_standin_scene(str(tmp_path / "c3"), "c3"))
    assert (c3.width, c3.height, c3.desc.max_depth) == (1920, 1080, 10) and abs(c3.epsilon - 0.0005) < 1e-9
    assert c3.desc.n_prims == 2 + 3 * 8064 + 3072 + 2 and c3.desc.n_textures == 1 and c3.desc.n_lights == 2
    mats = c3.array("materials", "n_materials", st.MATERIAL)
    kinds = set(int(t) for t in mats["type"])
    assert {st.MT_ROUGHCONDUCTOR, st.MT_DIELECTRIC, st.MT_SUBSTRATE, st.MT_LAMBERTIAN} <= kinds
    assert mats[8]["alphaU"] != mats[8]["alphaV"]               # "Outer": anisotropic, remapped
    c3.close()
    c4 = api.LoadedScene(scenes.write_standin_scene(str(tmp_path / "c4"), "c4", 64, 36))
    assert c4.desc.n_prims == 175998 - 2 + 9 * 8064 and c4.desc.n_lights == 0 and c4.desc.n_light_distribution == 2
    assert c4.desc.max_depth == 7
    c4.close()
