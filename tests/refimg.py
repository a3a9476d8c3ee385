"""Comparison of a linear film with one of the reference's published renders (test infrastructure).

The reference's repository holds PNGs its author rendered (result/*.png).  They went through Output's tone curve
(pathtracer.cu:187-204, 2516-2531) and the PNG writer's flip, clamp and 8-bit truncation (imageio.cpp:61-78).  A fixture
under tests/golden/ keeps such a picture box-filtered to 64 x 64 blocks; `compare` pushes a linear accumulator through the
same chain and reports how far the two pictures are apart."""
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def filmic_png(acc, spp, width, height):
    """linear accumulator (row 0 = bottom) -> what SavePng would have written: filmic curve, flipped, 8-bit, as [0,1] floats"""
    lin = np.asarray(acc, dtype=np.float64).reshape(height, width, 3) / spp
    c = np.maximum(0.0, lin - 0.004)
    img = (c * (6.2 * c + 0.5)) / (c * (6.2 * c + 1.7) + 0.06)               # FilmicTonemapping, pathtracer.cu:199-204
    img = np.floor(np.clip(img, 0.0, 1.0) * 255.0) / 255.0                   # imageio.cpp:61-78
    return img[::-1]


def blocks(img, n=64):
    h, w, _ = img.shape
    return img.reshape(n, h // n, n, w // n, 3).mean(axis=(1, 3))


def compare(acc, spp, width, height, want):
    """-> (worst-channel |frame-mean difference|, mean |block difference|, max |block difference|, got frame means)"""
    got = blocks(filmic_png(acc, spp, width, height), want.shape[0])
    d = np.abs(got - want)
    return float(np.abs(got.mean(axis=(0, 1)) - want.mean(axis=(0, 1))).max()), float(d.mean()), float(d.max()), got.mean(axis=(0, 1))


def load(name):
    return np.load(os.path.join(GOLDEN, name)).astype(np.float64)
