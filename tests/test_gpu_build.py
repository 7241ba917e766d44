"""The acceleration structure built on the GPU (rt_scene_rebuild_on_device) must be a valid tree
and leave every result of the library unchanged: the nearest hit does not depend on the tree."""
import os
import time

import numpy as np
import pytest

import raytracer_amd as rt
from raytracer_amd import _ffi, scenes
from test_host_parity import _adversarial_rays, _check_tree, _tree_contents

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.mark.parametrize("method", ["lbvh", "ploc"])
@pytest.mark.parametrize("name", ["spot", "sphere_floor", "ogre"])
def test_device_built_tree_is_valid_and_changes_no_result(oracle, name, method):
    if rt.device_count() < 1:
        pytest.fail("no HIP device visible")
    host = scenes.build(name)
    dev = scenes.build(name)
    dev.rebuild_on_device(0, method)
    i = dev.info()
    assert i.n_triangles == host.info().n_triangles and 0 < i.bvh_nodes < i.n_triangles and i.bvh_depth <= 62
    nodes, order = dev.export_bvh()
    tv = dev.export_flat()[0]
    ids, tv_all = _tree_contents(dev)
    _check_tree(nodes, order, ids, tv_all, i.bvh_depth)

    # the CPU restatement of the traversal over the device-built tree equals brute force ...
    o = oracle.OracleScene.build(name)
    rng = np.random.default_rng(7)
    org, d = _adversarial_rays(rng, tv, 1500)
    got, _ = o.bvh_hit(nodes, order, 0, org, d)
    want = o.hit(org, d, oracle.HIT_BRUTE)
    assert got.tobytes() == want.tobytes()  # spheres included: they have their places in the tree
    # ... and the kernels give the same bits with either tree
    assert dev.hit(org, d).tobytes() == host.hit(org, d).tobytes()
    w, h = (160, 120) if name == "spot" else (96, 54)
    p = rt.render_params(width=w, height=h, spp=3, seed=21)
    for a, b in zip(dev.primary_hits(p), host.primary_hits(p)):
        assert a.tobytes() == b.tobytes()
    ca = np.zeros((h, w, 3)); cb = np.zeros((h, w, 3))
    assert dev.render(ca, p) == host.render(cb, p)
    assert ca.tobytes() == cb.tobytes()


def _scene(tv, oracle=None):
    mat = rt.Opaque((0.5, 0.5, 0.5), 0, 1, 0)
    k = len(tv)
    tn = np.tile(np.array([0, 0, 1.0]), (k, 3, 1)); tuv = np.zeros((k, 3, 2))
    return rt.Scene.from_desc([_ffi.RT_OBJ_MESH], [mat], tv, tn, tuv, [0] * k, list(range(k)), np.zeros((0, 4)), [])


def test_small_and_degenerate_inputs(oracle):
    one = np.array([[-1, -1, -2], [1, -1, -2], [0, 1, -2]], dtype=np.float64)
    co, cd = oracle.camera_rays(32, 24)
    rng = np.random.default_rng(3)
    cases = {
        "one triangle": one[None],
        "three triangles (a single leaf)": one[None] + rng.normal(size=(3, 1, 3)) * 0.3,
        "five triangles": one[None] + rng.normal(size=(5, 1, 3)) * 0.3,
        "forty coincident triangles (equal Morton codes)": np.tile(one, (40, 1, 1)),
        "a line of equal centres and one far away": np.concatenate([np.tile(one, (9, 1, 1)), one[None] + 100.0]),
    }
    cases["a regular grid (every neighbour ties)"] = np.array(
        [[[x, y, -3.0], [x + 1, y, -3.0], [x, y + 1, -3.0]] for y in range(-8, 8) for x in range(-8, 8)], dtype=np.float64)
    # boxes that all overlap: clustering chains up here and hands over to the Morton hierarchy
    cases["a soup of scene-sized triangles"] = rng.normal(size=(5000, 3, 3)) * 0.01 + np.array([0, 0, -3.0])
    for (what, tv), method in [(c, m) for c in cases.items() for m in ("lbvh", "ploc")]:
        a, b = _scene(tv), _scene(tv)
        b.rebuild_on_device(0, method)
        i = b.info()
        nodes, order = b.export_bvh()
        assert sorted(order.tolist()) == list(range(len(tv))), what
        _check_tree(nodes, order, len(tv), b.export_flat()[0], i.bvh_depth)
        assert b.hit(co, cd).tobytes() == a.hit(co, cd).tobytes(), what
    # spheres only: nothing to build, nothing changes
    mat = rt.Opaque((0.5, 0.5, 0.5), 0, 1, 0)
    s = rt.Scene.from_desc([_ffi.RT_OBJ_SPHERE], [mat], np.zeros((0, 9)), [], [], [], [], [[0, 0, -3, 1]], [0])
    want = s.hit(co, cd).tobytes()
    s.rebuild_on_device(0)
    assert s.info().bvh_nodes == 0 and s.hit(co, cd).tobytes() == want
    with pytest.raises(rt.RtError, match="out of range"):
        s.rebuild_on_device(99)
    with pytest.raises(rt.RtError, match="unknown method"):
        s.rebuild_on_device(0, 7)


@pytest.mark.parametrize("method", ["lbvh", "ploc"])
def test_million_triangle_scene_builds_in_milliseconds(method):
    s = scenes.build("spot_grid:14x14")
    t0 = time.perf_counter()
    s.rebuild_on_device(0, method)
    first = time.perf_counter() - t0
    t0 = time.perf_counter()
    s.rebuild_on_device(0, method)
    again = time.perf_counter() - t0
    i = s.info()
    print(f"{method} build of {i.n_triangles} triangles: {first * 1e3:.1f} ms, again {again * 1e3:.1f} ms; "
          f"{i.bvh_nodes} nodes, depth {i.bvh_depth}")
    assert i.bvh_depth <= 62 and again < 2.0
    p = rt.render_params(width=160, height=90, spp=1, seed=1)
    host = scenes.build("spot_grid:14x14")
    for a, b in zip(s.primary_hits(p), host.primary_hits(p)):
        assert a.tobytes() == b.tobytes()


def test_scene_without_a_host_tree(oracle):
    """RT_TREE_NONE: creation skips the host build; nothing runs until the tree is built on the
    GPU; afterwards every result equals the host-built scene's."""
    host = scenes.build("ogre")
    bare = scenes.build("ogre", tree="none")
    i = bare.info()
    assert i.n_triangles == host.info().n_triangles and i.bvh_nodes == 0
    for call in (lambda: bare.upload(0), lambda: bare.hit([[0, 0, 0]], [[0, 0, -1]]), lambda: bare.export_bvh(),
                 lambda: bare.render(np.zeros((8, 8, 3)), rt.render_params(width=8, height=8, spp=1)), lambda: rt.Group(bare, [0])):
        with pytest.raises(rt.RtError, match="without a tree"):
            call()
    bare.rebuild_on_device(0)
    assert 0 < bare.info().bvh_nodes < i.n_triangles
    p = rt.render_params(width=96, height=54, spp=3, seed=2)
    ca = np.zeros((54, 96, 3)); cb = np.zeros((54, 96, 3))
    assert bare.render(ca, p) == host.render(cb, p)
    assert ca.tobytes() == cb.tobytes()
    # a scene of spheres has no tree to wait for
    mat = rt.Opaque((0.5, 0.5, 0.5), 0, 1, 0)
    objs = rt.Objects()
    objs.sphere((0, 0, -3), 1.0, mat)
    s = rt.Scene.from_objects(objs, tree="none")
    assert s.hit([[0, 0, 0]], [[0, 0, -1]])["t"][0] == 2.0



def _soup(n, seed=7):
    """Large slivers through a common region: every box overlaps most others, so bottom-up clustering
    chains the clusters into a tree several dozen levels deep."""
    rng = np.random.default_rng(seed)
    c = rng.normal(size=(n, 1, 3)) * 0.05 + np.array([0.0, 0.0, -3.0])
    tv = c + rng.normal(size=(n, 3, 3)) * np.array([1.0, 1.0, 0.3]) * 10.0 ** rng.uniform(-2.0, 0.0, (n, 1, 1))
    nrm = np.cross(tv[:, 1] - tv[:, 0], tv[:, 2] - tv[:, 0])
    nrm /= np.maximum(1e-30, np.linalg.norm(nrm, axis=1, keepdims=True))
    mat = rt.Opaque((0.6, 0.6, 0.6), 0, 0, 0.9)
    return ([_ffi.RT_OBJ_MESH], [mat], tv, np.repeat(nrm[:, None, :], 3, axis=1), np.zeros((n, 3, 2)), [0] * n, list(range(n)), np.zeros((0, 4)), [])


def test_deep_tree_short_stack(oracle, monkeypatch):
    """A tree of 38 levels or more runs the kernels' form that keeps sixteen stack entries in LDS and
    the deeper ones in private memory: same hits as the exhaustive test, and the same image and ray
    count as the full LDS stack."""
    args = _soup(4000)
    images = {}
    for deep in ("1", "0"):
        monkeypatch.setenv("RT_DEEP_STACK", deep)
        s = rt.Scene.from_desc(*args)
        s.rebuild_on_device(0, "ploc")
        assert s.info().bvh_depth >= 42, s.info().bvh_depth
        if deep == "1":
            o = oracle.OracleScene.from_desc_arrays(*args)
            rng = np.random.default_rng(3)
            org = rng.normal(size=(3000, 3)) * 2.0 + np.array([0.0, 0.0, -3.0])
            d = rng.normal(size=(3000, 3))
            co, cd = oracle.camera_rays(64, 48)
            org, d = np.concatenate([org, co]), np.concatenate([d, cd])
            assert s.hit(org, d).tobytes() == o.hit(org, d, oracle.HIT_BRUTE).tobytes()
        canvas = np.zeros((54, 96, 3))
        rays = s.render(canvas, rt.render_params(width=96, height=54, spp=4, seed=5))
        images[deep] = (rays, canvas.tobytes())
    monkeypatch.delenv("RT_DEEP_STACK")
    s = rt.Scene.from_desc(*args)
    s.rebuild_on_device(0, "ploc")  # by itself a tree this deep takes the short stack
    canvas = np.zeros((54, 96, 3))
    rays = s.render(canvas, rt.render_params(width=96, height=54, spp=4, seed=5))
    assert images["1"] == images["0"] == (rays, canvas.tobytes())
