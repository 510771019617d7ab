"""Shared helpers of the parity tests."""
import numpy as np

from magr_ray_tracer_amd import scenes

DEFAULT = dict(shading=1, sampling=1, accel=0, russian_roulette=True, filter_fireflies=True)


def bits_equal(a, b):
    """Bit-exact for floats except that +0 == -0."""
    a, b = np.asarray(a), np.asarray(b)
    if a.shape != b.shape:
        return False
    if a.dtype == np.float32:
        return bool(np.all((a.view(np.uint32) == b.view(np.uint32)) | ((a == 0) & (b == 0))))
    return bool(np.array_equal(a, b))


def mismatch(a, b):
    a, b = np.asarray(a), np.asarray(b)
    assert a.shape == b.shape, (a.shape, b.shape)
    if a.dtype == np.float32:
        bad = (a.view(np.uint32) != b.view(np.uint32)) & ~((a == 0) & (b == 0))
    else:
        bad = a != b
    return int(bad.sum())


def assert_bits(a, b, what):
    n = mismatch(a, b)
    assert n == 0, f"{what}: {n}/{np.asarray(a).size} elements differ"


def max_rel(a, b, floor=1e-6):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.max(np.abs(a - b) / np.maximum(np.abs(b), floor))) if a.size else 0.0


def build(scene_fn, W, H):
    s, view = scene_fn()
    sa = s.arrays()
    cam = scenes.camera_for(view, W, H)
    return s, sa, cam
