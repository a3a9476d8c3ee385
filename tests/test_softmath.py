"""include/gpt_softmath.h on the host: correctly rounded against float64 libm, edge cases (no GPU)."""
import numpy as np
import pytest

import oracle_lib as ol
from gpu_pathtracer_amd import scene_types as st


def batch(kind, fn, x, y=None):
    x = np.ascontiguousarray(x, np.float32)
    y = np.ascontiguousarray(y if y is not None else x, np.float32)
    o = np.zeros_like(x)
    ol.load(kind).oracle_math_batch(fn, st.ptr(x), st.ptr(y), st.ptr(o), len(x))
    return o


CASES = [
    (0, "sin", np.sin, lambda r, n: r.random(n) * 8.5 - 0.5),
    (1, "cos", np.cos, lambda r, n: r.random(n) * 8.5 - 0.5),
    (2, "tan", np.tan, lambda r, n: r.random(n) * 8.5 - 0.5),
    (3, "atan", np.arctan, lambda r, n: r.standard_normal(n) * np.exp(r.standard_normal(n) * 4)),
    (4, "acos", np.arccos, lambda r, n: r.random(n) * 2 - 1),
    (9, "exp", np.exp, lambda r, n: r.standard_normal(n) * 20),
    (10, "log", np.log, lambda r, n: np.exp(r.standard_normal(n) * 12)),
]


@pytest.mark.parametrize("fn,name,ref,gen", CASES, ids=[c[1] for c in CASES])
def test_soft_functions_are_correctly_rounded(fn, name, ref, gen):
    rng = np.random.default_rng(fn)
    x = gen(rng, 400_000).astype(np.float32)
    got = batch("soft", fn, x)
    want = ref(x.astype(np.float64)).astype(np.float32)
    bad = np.count_nonzero(got != want)
    assert bad <= 2, f"{name}: {bad} results differ from the correctly rounded value"


def test_soft_pow_gamma_tonemap_range():
    rng = np.random.default_rng(9)
    x = (rng.random(300_000) * 40 + 1e-5).astype(np.float32)
    y = np.full_like(x, np.float32(1 / 2.2))
    got = batch("soft", 5, x, y)
    want = np.power(x.astype(np.float64), y.astype(np.float64)).astype(np.float32)
    assert np.count_nonzero(got != want) <= 2


def test_soft_edge_cases():
    x = np.array([0.0, -0.0, 1.0, -1.0], np.float32)
    assert batch("soft", 4, x).tolist() == [np.float32(np.pi / 2), np.float32(np.pi / 2), 0.0, np.float32(np.pi)]
    assert np.isnan(batch("soft", 4, np.array([1.0000001, -2.0, np.nan], np.float32))).all()
    assert batch("soft", 0, np.zeros(1, np.float32))[0] == 0 and batch("soft", 1, np.zeros(1, np.float32))[0] == 1
    big = np.array([np.inf, -np.inf, 1e30, -1e30], np.float32)
    assert np.allclose(batch("soft", 3, big), [np.pi / 2, -np.pi / 2, np.pi / 2, -np.pi / 2])


def test_soft_exp_log_edge_cases():
    x = np.array([0.0, -0.0, 1.0, np.inf, -1.0, 1e-45, 88.9, -105.0, -np.inf], np.float32)
    with np.errstate(all="ignore"):
        lg = batch("soft", 10, x)
        ex = batch("soft", 9, x)
    assert lg[0] == -np.inf and lg[1] == -np.inf and lg[2] == 0 and lg[3] == np.inf and np.isnan(lg[4])
    assert lg[5] == np.float32(np.log(np.float64(np.float32(1e-45))))
    assert ex[0] == 1 and ex[2] == np.float32(np.e) and ex[3] == np.inf and ex[6] == np.inf and ex[7] == 0 and ex[8] == 0


def test_ieee_ops_match_numpy():
    rng = np.random.default_rng(4)
    a = (rng.standard_normal(100_000) * np.exp(rng.standard_normal(100_000) * 6)).astype(np.float32)
    b = (rng.standard_normal(100_000) * np.exp(rng.standard_normal(100_000) * 6)).astype(np.float32)
    assert (batch("soft", 6, a, b) == a / b).all()
    p = np.abs(a)
    assert (batch("soft", 7, p) == np.sqrt(p)).all()
    assert (batch("soft", 8, p[p > 0]) == np.float32(1) / np.sqrt(p[p > 0])).all()
