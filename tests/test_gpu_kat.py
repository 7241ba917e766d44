"""The hand-derived known answers of tests/golden/kat.json through the product itself: every
vector becomes a one-triangle or one-sphere scene and goes through rt_trace on the GPU (the CPU
suite runs the same vectors through the oracle). Covers each early-out of intersects_triangle
(src/main.rs:205, 218, 228, 234) and both roots of intersects_sphere (src/main.rs:253-259)."""
import json
import os

import numpy as np
import pytest

import raytracer_amd as rt
from raytracer_amd import _ffi

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
KAT = json.load(open(os.path.join(HERE, "golden", "kat.json")))
GREY = rt.Opaque((0.5, 0.5, 0.5), 0.0, 1.0, 0.0)


def tri_scene(a, b, c):
    tv = np.array([[a, b, c]], dtype=np.float64)
    tn = np.tile(np.array([0.0, 0.0, 1.0]), (1, 3, 1))
    s = rt.Scene.from_desc([_ffi.RT_OBJ_MESH], [GREY], tv, tn, np.zeros((1, 3, 2)), [0], [0], np.zeros((0, 4)), [])
    s.upload(0)
    return s


def sphere_scene(c, r):
    s = rt.Scene.from_desc([_ffi.RT_OBJ_SPHERE], [GREY], np.zeros((0, 3, 3)), np.zeros((0, 3, 3)), np.zeros((0, 3, 2)), [], [],
                           [[c[0], c[1], c[2], r]], [0])
    s.upload(0)
    return s


def one(scene, o, d):
    return scene.hit(np.array([o], dtype=np.float64), np.array([d], dtype=np.float64))[0]


def test_triangle_known_answers_on_the_gpu():
    if rt.device_count() < 1:
        pytest.fail("no HIP device visible")
    for name in ("floor_down", "unit_tri", "unit_tri_back", "unit_tri_dir2"):
        k = KAT[name]
        h = one(tri_scene(k["a"], k["b"], k["c"]), k["o"], k["d"])
        assert h["obj"] == 0 and h["tri"] == 0, name
        assert h["t"] == pytest.approx(k["t"], rel=1e-15, abs=1e-15), name
        assert h["u"] == pytest.approx(k["u"], rel=1e-14, abs=1e-15), name
        assert h["v"] == pytest.approx(k["v"], rel=1e-14, abs=1e-15), name


def test_triangle_early_outs_on_the_gpu():
    for i, k in enumerate(KAT["misses"]):
        assert one(tri_scene(k["a"], k["b"], k["c"]), k["o"], k["d"])["obj"] == -1, i
    for i, k in enumerate(KAT["edge_hits"]):
        h = one(tri_scene(k["a"], k["b"], k["c"]), k["o"], k["d"])
        assert (h["obj"], h["tri"], h["t"], h["u"], h["v"]) == (0, 0, k["t"], k["u"], k["v"]), i
    # |det| < 1e-8 is rejected, just above accepted (src/main.rs:205); dist must exceed 1e-8 (234)
    s = tri_scene([0, 0, -1], [1, 0, -1], [0, 1, -1])
    assert one(s, [0.25, 0.25, 0], [0, 0, -0.9e-8])["obj"] == -1
    assert one(s, [0.25, 0.25, 0], [0, 0, -1.1e-8])["obj"] == 0
    assert one(s, [0.25, 0.25, -1 + 0.9e-8], [0, 0, -1])["obj"] == -1
    assert one(s, [0.25, 0.25, -1 + 1.1e-8], [0, 0, -1])["obj"] == 0


def test_sphere_known_answers_on_the_gpu():
    for i, k in enumerate(KAT["sphere"]):
        h = one(sphere_scene(k["c"], k["r"]), k["o"], k["d"])
        if k["t"] is None:
            assert h["obj"] == -1, i
        else:
            assert h["obj"] == 0 and h["tri"] == -1 and h["t"] == pytest.approx(k["t"], rel=1e-15), i
