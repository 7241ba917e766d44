"""Randomly made scenes rendered by the HIP path and by the oracle: triangle soups split over several
mesh objects, from none to fifteen spheres (nine or more get places in the tree, which takes the
kernels' general form), opaque materials from purely diffuse to mirror with and without emission,
colour textures, normal maps, dielectrics, arbitrary vertex normals and uvs. Equal ray counts and
images within 1e-12, as for the built-in scenes; every scheduling form on some of them."""
import numpy as np
import pytest

import raytracer_amd as rt
from raytracer_amd import _ffi

pytestmark = pytest.mark.gpu
IMG_RTOL = 1e-12


def _material(rng, n_tex, allow_emission):
    kind = rng.random()
    if kind < 0.15:
        return rt.Dielectric(rng.uniform(1.1, 1.9))
    color = int(rng.integers(0, n_tex)) if n_tex and rng.random() < 0.4 else tuple(rng.uniform(0.05, 0.95, 3))
    how = (0.0, 1.0, rng.random())[int(rng.integers(0, 3))]
    emission = tuple(rng.uniform(0.0, 0.5, 3)) if allow_emission and rng.random() < 0.4 else (0.0, 0.0, 0.0)
    normal = int(rng.integers(0, n_tex)) if n_tex and rng.random() < 0.3 else None
    return rt.Opaque(color, rng.random(), rng.random(), how, emission=emission, normal=normal)


def _random_scene(seed):
    rng = np.random.default_rng(seed)
    n_tri = int(rng.choice([0, 1, 30, 400, 3000]))
    n_mesh = 0 if n_tri == 0 else int(rng.integers(1, 4))
    n_sph = int(rng.choice([0, 2, 8, 9, 15])) if n_tri else int(rng.choice([2, 9]))
    n_tex = int(rng.integers(0, 3))
    textures = [rng.integers(0, 256, size=(int(rng.choice([1, 4, 16])), int(rng.choice([1, 8, 5])), 4), dtype=np.uint8) for _ in range(n_tex)]
    allow_emission = rng.random() < 0.5
    kinds = [_ffi.RT_OBJ_MESH] * n_mesh + [_ffi.RT_OBJ_SPHERE] * n_sph
    mats = [_material(rng, n_tex, allow_emission) for _ in kinds]
    # triangles in the camera's view (it looks down -z from the origin), a large one as a floor
    size = 10.0 ** rng.uniform(-1.5, 0.0, (n_tri, 1, 1))
    tv = rng.normal(size=(n_tri, 1, 3)) * np.array([1.5, 1.0, 1.0]) + np.array([0.0, 0.0, -4.0]) + rng.normal(size=(n_tri, 3, 3)) * size
    if n_tri > 1:
        tv[0] = [[30, -1.2, -30], [-30, -1.2, -30], [0, -1.2, 5]]
    tn = rng.normal(size=(n_tri, 3, 3))
    geometric = rng.random(n_tri) < 0.5  # half of them carry their plane's normal at every corner
    if n_tri:
        g = np.cross(tv[:, 1] - tv[:, 0], tv[:, 2] - tv[:, 0])
        tn[geometric] = np.repeat(g[geometric, None, :], 3, axis=1)
    tuv = rng.uniform(-0.5, 1.5, size=(n_tri, 3, 2))
    tri_obj = rng.integers(0, max(1, n_mesh), n_tri)
    tri_idx = np.zeros(n_tri, dtype=np.int64)
    for m in range(n_mesh):
        tri_idx[tri_obj == m] = np.arange(np.count_nonzero(tri_obj == m))
    sph = np.concatenate([rng.normal(size=(n_sph, 3)) * np.array([1.5, 0.7, 1.0]) + np.array([0.0, 0.0, -3.5]),
                          rng.uniform(0.1, 0.6, size=(n_sph, 1))], axis=1)
    return (kinds, mats, tv, tn, tuv, tri_obj, tri_idx, sph, list(range(n_mesh, n_mesh + n_sph)), textures)


@pytest.mark.parametrize("seed", range(12))
def test_random_scene_renders_as_the_oracle_does(oracle, seed):
    if rt.device_count() < 1:
        pytest.fail("no HIP device visible")
    args = _random_scene(1000 + seed)
    s = rt.Scene.from_desc(*args)
    o = oracle.OracleScene.from_desc_arrays(*args)
    # nearest hits of the camera's rays first: bit for bit
    co, cd = oracle.camera_rays(48, 36)
    assert s.hit(co, cd).tobytes() == o.hit(co, cd, oracle.HIT_BRUTE).tobytes()
    w, h, spp = 48, 36, 3
    want = np.zeros((h, w, 3))
    p = dict(width=w, height=h, spp=spp, seed=seed, dof_strength=(0.0, 0.05)[seed & 1])
    orays, _ = o.render(want, rt.render_params(**p))
    forms = (0, _ffi.RT_FLAG_MEGAKERNEL, _ffi.RT_FLAG_NO_REGEN) if seed % 3 == 0 else (0,)
    for flags in forms:
        got = np.zeros((h, w, 3))
        assert s.render(got, rt.render_params(flags=flags, **p)) == orays, flags
        err = np.max(np.abs(got - want) / np.maximum(1.0, np.abs(want)))
        assert err < IMG_RTOL, (flags, err)
    # the same scene with its tree built on the GPU
    if len(args[2]):
        s.rebuild_on_device(0, ("ploc", "lbvh")[seed & 1])
        got = np.zeros((h, w, 3))
        assert s.render(got, rt.render_params(**p)) == orays
        assert np.max(np.abs(got - want) / np.maximum(1.0, np.abs(want))) < IMG_RTOL
