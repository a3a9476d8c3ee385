"""The reference-shaped C++ interface (csrc/pathtracer.h) used from a C++ caller."""
import json
import os
import shutil
import subprocess

import numpy as np
import pytest

import oracle_lib as ol

CXX = os.path.join(ol.ROOT, "tests", "cxx", "mirror_main.cpp")
LIBDIR = os.path.join(ol.ROOT, "gpu_pathtracer_amd")
SCENE = os.path.join(ol.ROOT, "scenes", "cornell_pt", "scene.json")


def build(tmp_path):
    exe = str(tmp_path / "mirror_main")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", CXX, "-o", exe,
                           f"-L{LIBDIR}", "-lgpt", "-L/opt/rocm/lib", "-lamdhip64", f"-Wl,-rpath,{LIBDIR}", "-Wl,-rpath,/opt/rocm/lib"])
    return exe


def test_cxx_caller_builds_and_runs_host_side(tmp_path):
    exe = build(tmp_path)
    out = subprocess.run([exe, SCENE, "1", str(tmp_path / "o.bin"), "host"], capture_output=True, text=True)
    assert out.returncode == 0, out.stderr
    assert "host-only ok: 36 prims 27 nodes" in out.stdout


@pytest.mark.gpu
def test_cxx_render_calls_match_oracle(tmp_path):
    exe = build(tmp_path)
    out_bin = str(tmp_path / "acc.bin")
    out = subprocess.run([exe, SCENE, "3", out_bin], capture_output=True, text=True)
    assert out.returncode == 0, out.stderr + out.stdout
    got = np.fromfile(out_bin, dtype=np.float32)
    scene, meta = ol.load_cornell(8)
    cam = ol.cornell_camera(meta, 512, 512)
    ref, _ = ol.render(scene, cam, 512, 512, 0.001, 1, 3)
    assert got.tobytes() == ref.tobytes()


@pytest.mark.gpu
def test_cxx_output_buffer_is_ordered_with_the_default_stream(tmp_path):
    """The reference launches on the default stream, so its caller reads `output` right after Render() (src/main.cpp:139-143).
    The context's stream is a blocking stream: the same caller, with no synchronisation call, gets the finished tonemapped
    frame - bit for bit the oracle's Output."""
    exe = build(tmp_path)
    out_bin = str(tmp_path / "out.bin")
    spp = 6
    out = subprocess.run([exe, SCENE, str(spp), out_bin, "output"], capture_output=True, text=True)
    assert out.returncode == 0, out.stderr + out.stdout
    got = np.fromfile(out_bin, dtype=np.float32)
    scene, meta = ol.load_cornell(8)
    cam = ol.cornell_camera(meta, 512, 512)
    _, _, ref = ol.render(scene, cam, 512, 512, 0.001, 1, spp, want_out=True)
    assert ref.max() > 0.5 and got.tobytes() == ref.tobytes()


@pytest.mark.gpu
def test_cxx_volpath_scene_matches_oracle(tmp_path):
    """A "vpt" scene on disk in the shape of the reference's scenes/cornell_box/scene.json (a homogeneous camera medium,
    a density grid read from a text file inside a mesh without a material) through LoadScene / BeginRender / Render."""
    import scenes
    from gpu_pathtracer_amd import api, scene_types as st
    d = tmp_path / "scene"
    shutil.copytree(os.path.dirname(SCENE), d)
    js = json.load(open(d / "scene.json"))
    W = H = 128
    js.update({"screen_width": W, "screen_height": H, "integrator": "vpt", "maxDepth": 9})
    grid = scenes.smoke_grid(12, 10, 8, seed=5)
    with open(d / "geometry" / "density.d", "w") as f:
        for v in grid.ravel():
            f.write(f"{v:.6f}\n")
    js["medium"] = [{"type": "homogeneous", "sigmaA": [0.0014, 0.0025, 0.0142], "sigmaS": [0.70, 1.22, 1.90], "scale": 0.1, "name": "vol"},
                    {"type": "heterogeneous", "sigmaA": [1, 1, 1], "sigmaS": [9, 9, 9], "nx": 12, "ny": 10, "nz": 8, "g": 0.4,
                     "p0": [-0.6, 0.0, -0.6], "p1": [0.6, 1.3, 0.6], "density": "geometry/density.d", "iterMax": 300, "name": "smoke"}]
    js["camera"]["medium"] = "vol"
    # the tall box loses its material and becomes the smoke's container
    for unit in js["scene"]:
        if unit["mesh"].endswith("tall.obj"):
            del unit["material"]
            unit.update({"inside": "smoke", "outside": "vol"})
    json.dump(js, open(d / "scene.json", "w"))
    exe = build(tmp_path)
    out_bin = str(tmp_path / "acc.bin")
    spp = 4
    out = subprocess.run([exe, str(d / "scene.json"), str(spp), out_bin], capture_output=True, text=True)
    assert out.returncode == 0, out.stderr + out.stdout
    got = np.fromfile(out_bin, dtype=np.float32)
    ls = api.LoadedScene(str(d / "scene.json"))
    assert ls.desc.integrator_type == st.IT_VPT and ls.desc.n_mediums == 2 and ls.camera.medium == 0
    _, meta = ol.load_cornell(8)
    cam = ol.cornell_camera(meta, W, H)
    cam.medium = 0
    ref, _ = ol.render(ls, cam, W, H, 0.001, 1, spp)
    assert np.isfinite(ref).all() and ref.mean() > 0
    assert got.tobytes() == ref.tobytes()
