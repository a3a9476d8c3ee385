"""The reference-shaped C++ interface (csrc/pathtracer.h) used from a C++ caller."""
import os
import subprocess

import numpy as np
import pytest

import oracle_lib as ol

CXX = os.path.join(ol.ROOT, "tests", "cxx", "mirror_main.cpp")
LIBDIR = os.path.join(ol.ROOT, "gpu_pathtracer_amd")
SCENE = os.path.join(ol.ROOT, "scenes", "cornell_pt", "scene.json")


def build(tmp_path):
    exe = str(tmp_path / "mirror_main")
    subprocess.check_call(["g++", "-std=c++17", "-O1", CXX, "-o", exe, f"-L{LIBDIR}", "-lgpt", f"-Wl,-rpath,{LIBDIR}",
                           "-Wl,-rpath,/opt/rocm/lib"])
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
