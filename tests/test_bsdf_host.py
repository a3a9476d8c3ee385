"""The product's surface scattering (gpu_pathtracer_amd/csrc/pt_bsdf.h), compiled for the HOST by tests/cxx/bsdf_host.cpp, against the
oracle's SampleBSDF / Fr restatement (oracle/pt_oracle.c) - bit for bit, on constructed edge cases and random inputs, without a GPU.

pt_bsdf.h is organised by operation (shared Surface terms, one closing routine for all rough lobes), the oracle follows the reference
statement by statement (src/pathtracer.cu:491-826): two different decompositions of the same arithmetic.  This is the fast gate for
changes to that header; the same cases run on the device through gpt_debug_bsdf (tests/test_gpu_parity.py).
"""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

import bsdf_cases as bc
import oracle_lib as ol
from gpu_pathtracer_amd import scene_types as st

SRC = os.path.join(ol.ROOT, "tests", "cxx", "bsdf_host.cpp")
HDRS = [os.path.join(ol.ROOT, "gpu_pathtracer_amd", "csrc", h) for h in ("pt_bsdf.h", "pt_device.h", "pt_layout.h", "pt_vec.h")]
HIPCC = "/opt/rocm/bin/hipcc"


def P(a):
    return a.ctypes.data_as(C.c_void_p)


@pytest.fixture(scope="module")
def host_lib(tmp_path_factory):
    if not os.path.exists(HIPCC):
        pytest.skip("no hipcc")
    so = str(tmp_path_factory.mktemp("bsdf_host") / "libbsdf_host.so")
    # the float contract of csrc/Makefile: no contraction, IEEE divide and square root (the host's are)
    subprocess.check_call([HIPCC, "-x", "hip", "--cuda-host-only", "-O2", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off",
                           "-fno-fast-math", "-fno-slp-vectorize", SRC, "-o", so])
    return C.CDLL(so)


def run_both(host_lib, m, tex, g, a, mode):
    n = len(g)
    mine, ref = np.zeros((n, 7), np.float32), np.zeros((n, 7), np.float32)
    rec = st.Texture()
    rec.data, rec.height, rec.width = tex.ctypes.data, tex.shape[0], tex.shape[1]
    host_lib.host_bsdf_batch(P(m), P(tex), tex.shape[1], tex.shape[0], P(g), P(a), n, mode, P(mine))
    ol.load("soft").oracle_bsdf_batch(P(m), C.byref(rec), P(g), P(a), n, mode, P(ref))
    return mine, ref


@pytest.mark.parametrize("name", list(bc.materials()))
def test_host_build_of_the_product_bsdf_is_bit_equal_to_the_oracle(host_lib, name):
    m = bc.materials()[name]
    tex = bc.texture()
    n = 200_000
    g, u, wi = bc.cases(name, m, n, seed=abs(hash(name)) % 10007 + 1)
    for mode, a, what in ((1, u, "scatter"), (0, wi, "respond")):
        mine, ref = run_both(host_lib, m, tex, g, a, mode)
        same = bc.same_bits(mine, ref)
        bad = np.nonzero(~same.all(1))[0]
        assert bad.size == 0, (f"{name} {what}: {bad.size}/{n} cases differ; first {bad[0]}: geom {g[bad[0]]} in {a[bad[0]]} "
                               f"mine {mine[bad[0]]} oracle {ref[bad[0]]}")
        if what == "scatter" and name != "unknown_kind":
            assert np.isfinite(ref[:, 6]).mean() > 0.5          # the cases are not all degenerate
