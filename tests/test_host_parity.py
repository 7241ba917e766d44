"""Host half of the library against the oracle, no GPU: OBJ loading, transforms, flattening
(Scene::from_objects), the exported acceleration structure, quantiser, error behaviour."""
import ctypes as C
import os

import numpy as np
import pytest

import raytracer_amd as rt
from raytracer_amd import _ffi, scenes

SCENES = ["spot", "sphere_floor", "ogre", "lens"]


@pytest.mark.parametrize("name", SCENES)
def test_flatten_bit_exact(oracle, name):
    s = scenes.build(name)
    o = oracle.OracleScene.build(name)
    got = s.export_flat()
    want = o.flat()
    for g, w, what in zip(got, want, ("vertices", "normals", "uvs", "obj", "tri")):
        assert g.shape == w.shape, what
        # bitwise (NaNs included: a vertex no face uses has a NaN normal in the reference too)
        assert g.tobytes() == w.tobytes(), (name, what)
    i = s.info()
    nt, ns, no = o.counts()
    assert (i.n_triangles, i.n_spheres, i.n_objects) == (nt, ns, no)


def test_grid_scene_flatten(oracle):
    s = scenes.build("spot_grid:3x2")
    o = oracle.OracleScene.build("spot_grid:3x2")
    assert s.info().n_triangles == 1 + 6 * 5856
    for g, w in zip(s.export_flat(), o.flat()):
        assert g.tobytes() == w.tobytes()


def test_mesh_counts_and_transforms(oracle):
    m = rt.load_obj(os.path.join(scenes.ASSETS, "spot_triangulated.obj"))
    assert m.counts() == oracle.load_obj_counts(os.path.join(scenes.ASSETS, "spot_triangulated.obj"))
    c = m.clone()
    c.rotate_x(0.3); c.rotate_z(-1.2); c.translate((1, 2, 3))
    assert c.counts() == m.counts()


def test_load_obj_errors(tmp_path):
    with pytest.raises(_ffi.RtError) as e:
        rt.load_obj(tmp_path / "missing.obj")
    assert e.value.code == _ffi.RT_ERR_IO
    bad = tmp_path / "bad.obj"
    bad.write_text("v 0 0 0\nv 1 0 0\nv 0 1 0\nf 1 2 9\n")
    with pytest.raises(_ffi.RtError):
        rt.load_obj(bad)
    bad.write_text("v 0 0 zero\n")
    with pytest.raises(_ffi.RtError):
        rt.load_obj(bad)
    # blank lines, comments, vn lines and extra face corners are tolerated (vn ignored, quads cut)
    ok = tmp_path / "ok.obj"
    ok.write_text("# c\n\nv 0 0 0\nv 1 0 0\nv 0 1 0\nv 1 1 0\nvn 0 0 1\nvt 0.5 0.5\nf 1/1/1 2/1/1 3/1/1 4/1/1\n")
    assert rt.load_obj(ok).counts() == (4, 1, 2)


def test_numbers_are_parsed_as_rust_parses_them(tmp_path):
    """load_obj reads coordinates with str::parse::<f64> and indices with parse::<usize>
    (src/main.rs:1147, 1161): decimal notation, optional sign and exponent, no hexadecimal floats,
    no negative indices; and not through strtod, whose decimal point follows the process locale."""
    tri = "v 0 0 0\nv 1 0 0\nv 0 1 0\n"
    ok = tmp_path / "ok.obj"
    ok.write_text("v +0.5 -1e-3 2.5E+1\nv 1 0 0\nv 0 1 0\nf +1 2 3\n")
    assert rt.load_obj(ok).counts()[:2] == (3, 1)
    for bad_line in ("v 0x1p3 0 0\n", "v 1,5 0 0\n", "v 1.5x 0 0\n", "v ++1 0 0\n", "v  \n"):
        f = tmp_path / "bad.obj"
        f.write_text(bad_line + tri + "f 1 2 3\n")
        with pytest.raises(_ffi.RtError):
            rt.load_obj(f)
    for bad_face in ("f -1 2 3\n", "f 1 2 3x\n", "f 0x1 2 3\n", "f 1.0 2 3\n"):
        f = tmp_path / "badf.obj"
        f.write_text(tri + bad_face)
        with pytest.raises(_ffi.RtError):
            rt.load_obj(f)


def test_scene_create_validation():
    mat = rt.Opaque((1, 1, 1), 0, 1, 0)
    tv = np.zeros((1, 3, 3)); tn = np.zeros((1, 3, 3)); tuv = np.zeros((1, 3, 2))
    # triangle naming a sphere object
    with pytest.raises(_ffi.RtError) as e:
        rt.Scene.from_desc([_ffi.RT_OBJ_SPHERE], [mat], tv, tn, tuv, [0], [0], [[0, 0, 0, 1]], [0])
    assert e.value.code == _ffi.RT_ERR_INVALID
    # sphere count mismatch
    with pytest.raises(_ffi.RtError):
        rt.Scene.from_desc([_ffi.RT_OBJ_SPHERE], [mat], np.zeros((0, 9)), [], [], [], [], np.zeros((0, 4)), [])
    # texture id out of range
    with pytest.raises(_ffi.RtError):
        rt.Scene.from_desc([_ffi.RT_OBJ_MESH], [rt.Opaque(3, 0, 1, 0)], tv, tn, tuv, [0], [0], np.zeros((0, 4)), [])
    # no objects at all
    with pytest.raises(_ffi.RtError):
        rt.Scene.from_desc([], [], np.zeros((0, 9)), [], [], [], [], np.zeros((0, 4)), [])
    # a sphere-only scene is fine and has an empty tree
    s = rt.Scene.from_desc([_ffi.RT_OBJ_SPHERE], [mat], np.zeros((0, 9)), [], [], [], [], [[0, 0, -3, 1]], [0])
    assert s.info().bvh_nodes == 0 and s.info().n_spheres == 1


def _tree_contents(scene):
    """What the scene's tree is built over: (ids, vertices). Ids 0 .. n_triangles - 1 are the flat
    triangles; n_triangles + i is the place of sphere i, found like a triangle's from three
    vertices (two opposite corners of the sphere's box and its centre). vertices is indexed by id
    and covers every sphere; ids lists what must sit in the tree (the first 64 finite spheres)."""
    tv = scene.export_flat()[0]
    _, _, sph, _ = scene.export_objects()
    r = np.abs(sph[:, 3:4])
    places = np.stack([sph[:, :3] - r, sph[:, :3] + r, sph[:, :3]], axis=1).reshape(-1, 3, 3)
    in_tree = [i for i in range(min(len(sph), 64)) if np.all(np.isfinite(sph[i]))]
    if scene.info().bvh_spheres == 0:  # a scene of fewer than nine spheres: every ray tests them all
        in_tree = []
    assert scene.info().bvh_spheres == len(in_tree)
    return list(range(len(tv))) + [len(tv) + i for i in in_tree], np.concatenate([tv.reshape(-1, 3, 3), places])


def _check_tree(nodes, order, ids, tv, depth_reported):
    """Everything the tree is built over is in exactly one leaf; every child box contains what
    lies below it. ids, tv: see _tree_contents (or plain triangles: a count and their vertices)."""
    if isinstance(ids, (int, np.integer)):
        ids = list(range(int(ids)))
    assert sorted(np.asarray(order).tolist()) == sorted(ids)
    n_tris = len(ids)
    f = nodes.view(np.float32)
    seen = np.zeros(len(tv), np.int32)
    max_depth = 0

    def walk(ref, depth):
        nonlocal max_depth
        if ref < 0:
            leaf = ~ref
            first, cnt = leaf >> 2, (leaf & 3) + 1
            max_depth = max(max_depth, depth)
            flats = order[first:first + cnt]
            seen[flats] += 1
            lo = tv[flats].reshape(-1, 3).min(axis=0); hi = tv[flats].reshape(-1, 3).max(axis=0)
            return lo, hi
        n = nodes[ref].view(np.int32)
        boxes = []
        for c in range(2):
            lo, hi = walk(int(n[12 + c]), depth + 1)
            blo = np.array([f[ref][k * 4 + c * 2] for k in range(3)], np.float64)
            bhi = np.array([f[ref][k * 4 + c * 2 + 1] for k in range(3)], np.float64)
            assert np.all(blo < lo) and np.all(bhi > hi), "child box does not strictly contain its triangles"
            boxes.append((lo, hi))
        return np.minimum(boxes[0][0], boxes[1][0]), np.maximum(boxes[0][1], boxes[1][1])

    import sys
    sys.setrecursionlimit(10000)
    root = 0 if len(nodes) else (~((0 << 2) | (n_tris - 1)) if n_tris else None)
    if root is not None:
        walk(root, 0)
    assert np.all(seen[ids] == 1) and seen.sum() == len(ids)
    assert max_depth == depth_reported


def test_tree_invariants_with_the_six_spheres_placed(monkeypatch, oracle):
    monkeypatch.setenv("RT_SPHERE_TREE_FROM", "1")
    s = scenes.build("spot")
    i = s.info()
    assert i.bvh_spheres == 6
    nodes, order = s.export_bvh()
    ids, tv = _tree_contents(s)
    _check_tree(nodes, order, ids, tv, i.bvh_depth)
    co, cd = oracle.camera_rays(96, 72)
    o = oracle.OracleScene.build("spot")
    got, _ = o.bvh_hit(nodes, order, 0, co, cd)
    assert got.tobytes() == o.hit(co, cd, oracle.HIT_BRUTE).tobytes()


@pytest.mark.parametrize("name", ["spot", "sphere_floor"])
def test_tree_invariants(name):
    s = scenes.build(name)
    nodes, order = s.export_bvh()
    i = s.info()
    ids, tv = _tree_contents(s)
    assert i.bvh_spheres == 0
    assert i.bvh_depth <= 62
    _check_tree(nodes, order, ids, tv, i.bvh_depth)


def test_spheres_have_places_in_the_tree(oracle):
    """The tree culls spheres like triangles: the first 64 finite spheres each sit in one leaf, by
    their boxes; a ray tests the spheres whose place it reached and those without a place. The
    CPU restatement of that traversal equals the exhaustive test of every sphere (src/main.rs:802-810)."""
    rng = np.random.default_rng(31)
    k = 72
    cen = rng.normal(size=(k, 3)) * 2.0 + np.array([0.0, 0.0, -6.0])
    sph = np.concatenate([cen, rng.uniform(0.1, 0.6, size=(k, 1))], axis=1)
    sph[7, 3] = -sph[7, 3]  # a negative radius is a sphere like any other (only r * r enters, src/main.rs:247)
    sph[9, 0] = np.nan      # no place for this one: it is tested (and missed) by every ray
    mat = rt.Opaque((0.5, 0.5, 0.5), 0, 1, 0)
    tv = np.array([[[-30, -3, -30], [30, -3, -30], [0, -3, 5]]], dtype=np.float64)
    tn = np.tile(np.array([0, 1.0, 0]), (1, 3, 1)); tuv = np.zeros((1, 3, 2))
    args = ([_ffi.RT_OBJ_MESH] + [_ffi.RT_OBJ_SPHERE] * k, [mat] * (k + 1), tv, tn, tuv, [0], [0], sph, list(range(1, k + 1)))
    s = rt.Scene.from_desc(*args)
    o = oracle.OracleScene.from_desc_arrays(*args)
    i = s.info()
    assert i.bvh_spheres == 63 and i.n_spheres == k
    nodes, order = s.export_bvh()
    ids, verts = _tree_contents(s)
    assert 1 + 9 not in ids and 1 + 64 not in ids and 1 + 7 in ids
    _check_tree(nodes, order, ids, verts, i.bvh_depth)
    co, cd = oracle.camera_rays(120, 90)
    ro = rng.normal(size=(4000, 3)) * 3.0 + np.array([0.0, 0.0, -6.0]); rd = rng.normal(size=(4000, 3))
    org = np.concatenate([co, ro]); d = np.concatenate([cd, rd])
    want = o.hit(org, d, oracle.HIT_BRUTE)
    got, (n_node, n_tri) = o.bvh_hit(nodes, order, 0 if i.bvh_nodes else int(~np.int32(i.n_triangles + i.bvh_spheres - 1)), org, d)
    assert got.tobytes() == want.tobytes()
    assert np.count_nonzero(want["tri"] == -1) > 2000
    assert n_tri / len(org) < 12  # places visited per ray: far fewer than the 72 tests of the exhaustive loop


def _adversarial_rays(rng, tv, n):
    """Rays that stress box culling: axis-parallel, starting on vertices/edges/faces, through
    vertices and edge midpoints, grazing, far origins, tiny and huge direction lengths."""
    tris = tv[rng.integers(0, len(tv), n)]
    a, b, c = tris[:, 0], tris[:, 1], tris[:, 2]
    w = rng.random((n, 3)); w /= w.sum(axis=1, keepdims=True)
    inside = w[:, :1] * a + w[:, 1:2] * b + w[:, 2:] * c
    kinds = rng.integers(0, 8, n)
    target = np.where((kinds == 0)[:, None], a, np.where((kinds == 1)[:, None], 0.5 * (a + b), inside))
    org = rng.normal(size=(n, 3)) * 3.0
    org[kinds == 2] *= 1e4                                   # far away
    d = target - org
    ax = rng.integers(0, 3, n)
    sel = kinds == 3                                         # axis-parallel through the target
    d[sel] = 0.0
    d[sel, ax[sel]] = rng.choice([-1.0, 1.0], sel.sum())
    org[sel] = target[sel] - 5.0 * d[sel]
    sel = kinds == 4                                         # start exactly on the triangle
    org[sel] = inside[sel]
    d[sel] = rng.normal(size=(sel.sum(), 3))
    sel = kinds == 5                                         # start on a vertex, leave along an edge
    org[sel] = a[sel]
    d[sel] = (b - a)[sel]
    d[kinds == 6] *= 1e-6                                    # tiny direction (distances ~1e6)
    d[kinds == 7] *= 1e6
    return org, d


@pytest.mark.parametrize("name", ["spot", "sphere_floor", "ogre"])
def test_tree_traversal_equals_brute_force(oracle, name):
    """The kernel's traversal, restated on the CPU over the exported tree, returns the brute-force
    nearest hit bit for bit — on camera rays, random rays and rays built to stress the culling."""
    s = scenes.build(name)
    o = oracle.OracleScene.build(name)
    nodes, order = s.export_bvh()
    tv = s.export_flat()[0]
    rng = np.random.default_rng(5)
    w, h = (96, 72) if name != "ogre" else (48, 36)
    co, cd = oracle.camera_rays(w, h)
    n = 3000 if name != "ogre" else 1200
    ro = rng.normal(size=(n, 3)) * 2.0 + np.array([0, 0.5, -2.5]); rd = rng.normal(size=(n, 3))
    ao, ad = _adversarial_rays(rng, tv, n)
    org = np.concatenate([co, ro, ao]); d = np.concatenate([cd, rd, ad])
    want = o.hit(org, d, oracle.HIT_BRUTE)
    got, (n_node, n_tri) = o.bvh_hit(nodes, order, 0, org, d)
    assert np.array_equal(got["obj"], want["obj"]) and np.array_equal(got["tri"], want["tri"])
    assert got.tobytes() == want.tobytes()
    assert (want["obj"] >= 0).mean() > 0.3
    # pruning works: far fewer triangle tests than brute force
    assert n_tri / len(org) < 0.02 * s.info().n_triangles + 8


def test_quantiser_matches_reference_semantics():
    c = np.array([[[0.5, 1.0, 2.0], [-1.0, float("nan"), 0.999]]])
    q = rt.quantize_rgba8(c, 1)
    assert q.tolist() == [[[127, 255, 255, 255], [0, 0, 254, 255]]]
    q2 = rt.quantize_rgba8(c * 4, 4)
    assert np.array_equal(q, q2)


def test_device_calls_fail_loudly_without_gpu():
    if rt.device_count() > 0:
        pytest.skip("a GPU is present")
    s = scenes.build("sphere_floor")
    with pytest.raises(_ffi.RtError) as e:
        s.upload(0)
    assert e.value.code == _ffi.RT_ERR_NO_DEVICE
    with pytest.raises(_ffi.RtError) as e:
        s.hit([[0, 0, 0]], [[0, 0, -1]])
    assert e.value.code == _ffi.RT_ERR_NO_DEVICE
    canvas = np.zeros((8, 8, 3))
    with pytest.raises(_ffi.RtError):
        s.render(canvas, rt.render_params(width=8, height=8, spp=1))
    assert not canvas.any()
    with pytest.raises(_ffi.RtError) as e:
        rt.Group(s, [0, 1])
    assert e.value.code == _ffi.RT_ERR_NO_DEVICE


def test_group_arguments_are_checked_before_any_device_work():
    s = scenes.build("sphere_floor")
    for devices in ([], list(range(_ffi.RT_GROUP_MAX + 1))):
        with pytest.raises(_ffi.RtError) as e:
            rt.Group(s, devices)
        assert e.value.code == _ffi.RT_ERR_INVALID and "members" in str(e.value)
    import ctypes as C
    n = C.c_uint32()
    assert _ffi.lib.rt_group_size(None, C.byref(n)) == _ffi.RT_ERR_INVALID
    assert _ffi.lib.rt_group_member_info(None, 0, None, None, None) == _ffi.RT_ERR_INVALID
    _ffi.lib.rt_group_destroy(None)


def test_tree_does_not_depend_on_builder_threads(monkeypatch):
    """Large subtrees are built on separate threads; the tree must be the same for any count."""
    trees = []
    for threads in ("1", "3", "16"):
        monkeypatch.setenv("RT_BUILD_THREADS", threads)
        s = scenes.build("ogre")  # 39857 triangles: forks at the top two levels
        trees.append(s.export_bvh() + (s.info().bvh_depth, s.info().bvh_leaves))
    for nodes, order, depth, leaves in trees[1:]:
        assert np.array_equal(nodes, trees[0][0]) and np.array_equal(order, trees[0][1])
        assert (depth, leaves) == trees[0][2:]


def test_scene_created_without_a_tree_waits_for_one():
    bare = scenes.build("sphere_floor", tree="none")
    full = scenes.build("sphere_floor")
    assert bare.info().n_triangles == full.info().n_triangles == 321
    assert bare.info().bvh_nodes == 0 and full.info().bvh_nodes > 0
    for a, b in zip(bare.export_flat(), full.export_flat()):
        assert np.array_equal(a, b, equal_nan=True)
    with pytest.raises(_ffi.RtError, match="without a tree"):
        bare.export_bvh()
    import ctypes as C
    out = C.c_void_p()
    assert _ffi.lib.rt_scene_from_objects_with(None, 0, C.byref(out)) == _ffi.RT_ERR_INVALID
    objs = rt.Objects()
    assert _ffi.lib.rt_scene_from_objects_with(objs._h, 9, C.byref(out)) == _ffi.RT_ERR_INVALID

