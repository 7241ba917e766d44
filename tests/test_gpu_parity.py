"""Parity of the HIP path with the oracle, through the C ABI, on a real MI355X.

Bars: hit records (object id, triangle id, distance, barycentrics) bit-exact; ray counts equal;
images within 1e-12 relative (the kernels carry throughput as a running product where the
reference recursion multiplies on the way back, and evaluate the sky's x^6 in double-double
where the reference calls pow — both affect colour only, in the last bits)."""
import hashlib
import os

import numpy as np
import pytest

import raytracer_amd as rt
from raytracer_amd import _ffi, scenes

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
IMG_RTOL = 1e-12


@pytest.fixture(scope="module")
def gpu_scenes():
    if rt.device_count() < 1:
        pytest.fail("no HIP device visible: the GPU tests cannot run (there is no CPU fallback)")
    cache = {}

    def get(name):
        if name not in cache:
            s = scenes.build(name)
            s.upload(0)
            cache[name] = s
        return cache[name]

    return get


def rel_err(a, b):
    return np.max(np.abs(a - b) / np.maximum(1.0, np.abs(b)))


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


@pytest.mark.parametrize("name,w,h", [("spot", 160, 120), ("sphere_floor", 192, 108), ("ogre", 160, 90), ("lens", 80, 60)])
def test_primary_hits_bit_exact(gpu_scenes, oracle, name, w, h):
    s = gpu_scenes(name)
    o = oracle.OracleScene.build(name)
    obj, tri, t, u, v = s.primary_hits(rt.render_params(width=w, height=h))
    co, cd = oracle.camera_rays(w, h)
    want = o.hit(co, cd)
    assert np.array_equal(obj.ravel(), want["obj"])
    assert np.array_equal(tri.ravel(), want["tri"])
    for got, f in ((t, "t"), (u, "u"), (v, "v")):
        assert got.ravel().tobytes() == want[f].tobytes(), f
    g = os.path.join(HERE, "golden", f"primary_{name}_{w}x{h}.npz")
    if os.path.exists(g):
        g = np.load(g)
        for got, f in ((obj, "obj"), (tri, "tri"), (t, "t"), (u, "u"), (v, "v")):
            assert sha(got) == str(g[f"sha_{f}"]), f


def test_primary_hits_full_resolution_fixture(gpu_scenes):
    """Spot at the reference's default 1600x1200 against the committed fixture (hashes + crop)."""
    g = np.load(os.path.join(HERE, "golden", "primary_spot_1600x1200.npz"))
    obj, tri, t, u, v = gpu_scenes("spot").primary_hits(rt.render_params(width=1600, height=1200))
    for got, f in ((obj, "obj"), (tri, "tri"), (t, "t"), (u, "u"), (v, "v")):
        assert sha(got) == str(g[f"sha_{f}"]), f
    y0, x0 = g["crop_origin"]
    crop = g["crop"]
    assert np.array_equal(obj[y0:y0 + 16, x0:x0 + 16], crop["obj"])
    assert np.array_equal(t[y0:y0 + 16, x0:x0 + 16], crop["t"])


@pytest.mark.parametrize("name", ["spot", "sphere_floor", "ogre"])
def test_trace_random_and_adversarial_rays(gpu_scenes, oracle, name):
    from test_host_parity import _adversarial_rays

    s = gpu_scenes(name)
    o = oracle.OracleScene.build(name)
    tv = s.export_flat()[0]
    rng = np.random.default_rng(11)
    n = 6000 if name != "ogre" else 2500
    ro = rng.normal(size=(n, 3)) * 2.0 + np.array([0, 0.5, -2.5]); rd = rng.normal(size=(n, 3))
    ao, ad = _adversarial_rays(rng, tv, n)
    org = np.concatenate([ro, ao]); d = np.concatenate([rd, ad])
    got = s.hit(org, d)
    want = o.hit(org, d)
    assert got.tobytes() == want.tobytes()
    st = s.last_stats()
    assert st.rays == len(org) and st.tri_tests > 0


def test_trace_edge_cases(gpu_scenes, oracle):
    s = gpu_scenes("spot")
    o = oracle.OracleScene.build("spot")
    # empty batch
    assert len(s.hit(np.zeros((0, 3)), np.zeros((0, 3)))) == 0
    # zero direction, NaN, infinities: no hit, no hang
    org = np.array([[0, 0, 0], [0, 0, 0], [np.nan, 0, 0], [0, 0, 0], [1e300, 0, 0]], dtype=np.float64)
    d = np.array([[0, 0, 0], [np.nan, 0, -1], [0, 0, -1], [np.inf, 0, -1], [0, 0, -1]], dtype=np.float64)
    got = s.hit(org, d)
    want = o.hit(org, d)
    assert np.array_equal(got["obj"], want["obj"])
    # a ray from inside a sphere returns the far root (src/main.rs:254-259)
    got = s.hit([[-0.3, -0.2, -2.8]], [[0, 0, 1]])
    want = o.hit([[-0.3, -0.2, -2.8]], [[0, 0, 1]])
    assert got.tobytes() == want.tobytes() and got["tri"][0] == -1 and got["t"][0] == 0.5


def _sphere_rays(rng, spheres, n):
    """Rays that graze, leave, cross and start inside the spheres: the cases in which the box around
    the spheres and the early exits of the sphere test (negative discriminant; both roots behind
    the origin) decide."""
    org, d = [], []
    for c4 in spheres:
        c, r = c4[:3], c4[3]
        for _ in range(n):
            o = c + rng.normal(size=3) * r * rng.choice([0.5, 1.5, 4.0, 40.0])
            # towards a point on or just off the silhouette
            u = rng.normal(size=3); u /= np.linalg.norm(u)
            to_c = c - o
            side = np.cross(to_c, u); side /= max(np.linalg.norm(side), 1e-300)
            target = c + side * r * rng.choice([0.0, 0.999999, 1.0, 1.000001, 1.0 + 1e-12, 1.0 - 1e-12, 2.0])
            org.append(o); d.append((target - o) * rng.choice([1.0, 1e-3, 1e3, 1e-9]))
            # the same ray backwards: the sphere lies behind the origin
            org.append(o); d.append(o - target)
            # from the surface, outwards and inwards, as scattered rays start (src/main.rs:903)
            p = c + u * r
            w = rng.normal(size=3); w /= np.linalg.norm(w)
            org.append(p + 1e-8 * w); d.append(w)
            org.append(p); d.append(-u + 0.3 * w)
    return np.array(org), np.array(d)


def test_sphere_culling_keeps_every_hit(gpu_scenes, oracle, monkeypatch):
    s = gpu_scenes("spot")
    o = oracle.OracleScene.build("spot")
    rng = np.random.default_rng(29)
    _, _, sph, _ = s.export_objects()
    org, d = _sphere_rays(rng, sph, 400)
    got = s.hit(org, d)
    want = o.hit(org, d)
    assert got.tobytes() == want.tobytes()
    assert np.count_nonzero(got["tri"] == -1) > 2000 and np.count_nonzero(got["obj"] < 0) > 1000
    # the six spheres given places in the tree (by default only scenes of nine spheres or more do that;
    # read when the tree is built): a ray then tests the spheres whose place it reached instead of
    # all of them — the same records, the same image
    monkeypatch.setenv("RT_SPHERE_TREE_FROM", "1")
    s2 = scenes.build("spot")
    assert s2.info().bvh_spheres == 6 and s.info().bvh_spheres == 0
    assert s2.hit(org, d).tobytes() == want.tobytes()
    w, h, spp = 72, 40, 5
    a = np.zeros((h, w, 3)); b = np.zeros((h, w, 3))
    assert s.render(a, rt.render_params(width=w, height=h, spp=spp, seed=5)) == s2.render(b, rt.render_params(width=w, height=h, spp=spp, seed=5))
    assert a.tobytes() == b.tobytes()
    monkeypatch.delenv("RT_SPHERE_TREE_FROM")
    # spheres only, more of them than the candidate mask has bits (72 > 64), one of them not finite
    k = 72
    cen = rng.normal(size=(k, 3)) * 2.0 + np.array([0.0, 0.0, -6.0])
    sph = np.concatenate([cen, rng.uniform(0.1, 0.6, size=(k, 1))], axis=1)
    mat = rt.Opaque((0.5, 0.5, 0.5), 0, 1, 0)
    for variant in ("finite", "one infinite radius"):
        if variant != "finite":
            sph[5, 3] = np.inf
        args = ([_ffi.RT_OBJ_SPHERE] * k, [mat] * k, np.zeros((0, 9)), [], [], [], [], sph, list(range(k)))
        many = rt.Scene.from_desc(*args)
        assert many.info().bvh_spheres == (64 if variant == "finite" else 63) and many.info().bvh_nodes > 8
        om = oracle.OracleScene.from_desc_arrays(*args)
        co, cd = oracle.camera_rays(96, 64)
        org2, d2 = _sphere_rays(rng, sph[:, :4][np.isfinite(sph[:, 3])][:12], 60)
        org2 = np.concatenate([co, org2]); d2 = np.concatenate([cd, d2])
        assert many.hit(org2, d2).tobytes() == om.hit(org2, d2).tobytes(), variant
        for method in ("lbvh", "ploc"):  # the device builders place the spheres too
            many.rebuild_on_device(0, method)
            assert many.info().bvh_spheres == (64 if variant == "finite" else 63)
            assert many.hit(org2, d2).tobytes() == om.hit(org2, d2).tobytes(), (variant, method)


def test_sphere_only_and_single_triangle_scenes(oracle):
    mat = rt.Opaque((0.5, 0.5, 0.5), 0, 1, 0)
    args = ([_ffi.RT_OBJ_SPHERE], [mat], np.zeros((0, 9)), [], [], [], [], [[0, 0, -3, 1]], [0])
    s = rt.Scene.from_desc(*args)
    o = oracle.OracleScene.from_desc_arrays(*args)
    co, cd = oracle.camera_rays(32, 24)
    assert s.hit(co, cd).tobytes() == o.hit(co, cd).tobytes()
    tv = np.array([[[-1, -1, -2], [1, -1, -2], [0, 1, -2]]], dtype=np.float64)
    tn = np.tile(np.array([0, 0, 1.0]), (1, 3, 1)); tuv = np.zeros((1, 3, 2))
    args = ([_ffi.RT_OBJ_MESH], [mat], tv, tn, tuv, [0], [0], np.zeros((0, 4)), [])
    s = rt.Scene.from_desc(*args)
    o = oracle.OracleScene.from_desc_arrays(*args)
    got = s.hit(co, cd)
    assert got.tobytes() == o.hit(co, cd).tobytes() and (got["obj"] >= 0).any()
    p = rt.render_params(width=32, height=24, spp=3, seed=2)
    c = np.zeros((24, 32, 3)); oc = np.zeros((24, 32, 3))
    assert s.render(c, p) == o.render(oc, p)[0]
    assert rel_err(c, oc) < IMG_RTOL


def test_equal_distance_ties_take_lowest_flat_index(oracle):
    """Two coincident triangles: the reference's strict < keeps the first in flat order."""
    mat = rt.Opaque((0.5, 0.5, 0.5), 0, 1, 0)
    one = np.array([[-1, -1, -2], [1, -1, -2], [0, 1, -2]], dtype=np.float64)
    k = 9
    tv = np.tile(one, (k, 1, 1)); tn = np.tile(np.array([0, 0, 1.0]), (k, 3, 1)); tuv = np.zeros((k, 3, 2))
    args = ([_ffi.RT_OBJ_MESH] * 3, [mat] * 3, tv, tn, tuv, [0, 0, 0, 1, 1, 1, 2, 2, 2], [0, 1, 2] * 3, np.zeros((0, 4)), [])
    s = rt.Scene.from_desc(*args)
    o = oracle.OracleScene.from_desc_arrays(*args)
    co, cd = oracle.camera_rays(24, 18)
    got = s.hit(co, cd)
    assert got.tobytes() == o.hit(co, cd).tobytes()
    hit = got[got["obj"] >= 0]
    assert len(hit) and np.all(hit["obj"] == 0) and np.all(hit["tri"] == 0)


@pytest.mark.parametrize("name,w,h,spp", [("spot", 96, 72, 4), ("sphere_floor", 96, 54, 6), ("ogre", 64, 36, 3), ("lens", 64, 48, 3)])
def test_render_matches_oracle(gpu_scenes, oracle, name, w, h, spp):
    s = gpu_scenes(name)
    o = oracle.OracleScene.build(name)
    p = rt.render_params(width=w, height=h, spp=spp, seed=42)
    c = np.zeros((h, w, 3)); oc = np.zeros((h, w, 3))
    rays = s.render(c, p)
    orays, _ = o.render(oc, p)
    assert rays == orays
    assert rel_err(c, oc) < IMG_RTOL
    assert c.mean() > 0.1


@pytest.mark.parametrize("name", ["spot", "lens", "sphere_floor", "ogre"])
def test_render_matches_committed_vectors(gpu_scenes, name):
    """Against the fixtures alone (tests/golden/render_*.npz: the oracle's image and ray count)."""
    g = np.load(os.path.join(HERE, "golden", f"render_{name}.npz"))
    w, h, spp, seed = (int(x) for x in g["params"])
    c = np.zeros((h, w, 3))
    rays = gpu_scenes(name).render(c, rt.render_params(width=w, height=h, spp=spp, seed=seed))
    assert rays == int(g["rays"])
    assert rel_err(c, g["canvas"]) < IMG_RTOL


def test_render_accumulates_and_is_progressive(gpu_scenes, oracle):
    """render adds into the canvas (src/main.rs:1035-1037); successive frames use successive
    samples; frames of spp a and b equal one frame's sample streams a+b."""
    s = gpu_scenes("spot")
    o = oracle.OracleScene.build("spot")
    w, h = 64, 48
    c = np.full((h, w, 3), 0.25); oc = np.full((h, w, 3), 0.25)
    for its in range(3):
        p = rt.render_params(width=w, height=h, spp=2, first_sample=2 * its, seed=9)
        s.render(c, p)
        o.render(oc, p)
    assert rel_err(c, oc) < IMG_RTOL
    one = np.zeros((h, w, 3))
    s.render(one, rt.render_params(width=w, height=h, spp=6, seed=9))
    assert np.allclose((c - 0.25) / 3.0, one, rtol=1e-12, atol=1e-14)


def test_render_variants_are_bitwise_identical(gpu_scenes, monkeypatch):
    """Path regeneration, one-thread-per-path, the counting build and sample chunking all produce
    the same bits: the result depends only on (seed, pixel, sample)."""
    s = gpu_scenes("spot")
    w, h, spp = 72, 40, 5
    base = np.zeros((h, w, 3))
    rays = s.render(base, rt.render_params(width=w, height=h, spp=spp, seed=5))
    for flags in (_ffi.RT_FLAG_NO_REGEN, _ffi.RT_FLAG_COUNT_WORK, _ffi.RT_FLAG_NO_REGEN | _ffi.RT_FLAG_COUNT_WORK, _ffi.RT_FLAG_TIMING,
                  _ffi.RT_FLAG_MEGAKERNEL, _ffi.RT_FLAG_MEGAKERNEL | _ffi.RT_FLAG_COUNT_WORK, _ffi.RT_FLAG_TIMING | _ffi.RT_FLAG_COUNT_WORK):
        c = np.zeros((h, w, 3))
        assert s.render(c, rt.render_params(width=w, height=h, spp=spp, seed=5, flags=flags)) == rays
        assert c.tobytes() == base.tobytes(), flags
    st_paths = None
    c = np.zeros((h, w, 3))
    s.render(c, rt.render_params(width=w, height=h, spp=spp, seed=5, flags=_ffi.RT_FLAG_COUNT_WORK))
    st = s.last_stats()
    assert st.rays == rays and st.paths == w * h * spp and st.node_visits > st.rays and st.shaded_hits < st.rays
    # the wavefront form's launch log: first bounce over every path, then shrinking queues
    log = s.last_launches()
    assert log[0].kind == 0 and log[0].items == ((w + 7) // 8) * ((h + 7) // 8) * 64 * spp
    assert log[-1].kind == 4 and sum(l.rays for l in log) == rays
    assert all(a.items >= b.items for a, b in zip(log[:-1], log[1:-1])) and log[0].items > log[-2].items
    s2 = scenes.build("spot")
    for env in ({"RT_WORK_BUFFER_MIB": "1"},          # several sample chunks
                {"RT_TAIL_THRESHOLD": "1"},           # (almost) never switch to the looping tail kernel
                {"RT_TAIL_THRESHOLD": "100000000"},   # tail kernel right after the first bounce
                {"RT_LONG_STEPS": "0"},               # one run per segment: hits not filed by the cost of their ray
                {"RT_LONG_STEPS": "1", "RT_TAIL_THRESHOLD": "1"},   # every hit in the back run
                {"RT_LONG_STEPS": "6", "RT_FUSE_BOUNCES": "0"},     # first stage queues its primary hits, both runs in use
                {"RT_FUSE_BOUNCES": "3", "RT_WORK_BUFFER_MIB": "1"},
                {"RT_WORK_BUFFER_MIB": "1", "RT_NO_OVERLAP": "1"},  # chunks one after the other, no side stream
                {"RT_WORK_BUFFER_MIB": "2", "RT_TAIL_THRESHOLD": "300"}):  # overlapped tails, queues that empty before the threshold
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        for flags in (0, _ffi.RT_FLAG_MEGAKERNEL, _ffi.RT_FLAG_COUNT_WORK | _ffi.RT_FLAG_TIMING):
            c = np.zeros((h, w, 3))
            assert s2.render(c, rt.render_params(width=w, height=h, spp=spp, seed=5, flags=flags)) == rays
            assert c.tobytes() == base.tobytes(), (env, flags)
            if flags & _ffi.RT_FLAG_COUNT_WORK:
                assert s2.last_stats().rays == rays and s2.last_stats().paths == w * h * spp
                assert sum(l.rays for l in s2.last_launches()) == rays, env
        for k in env:
            monkeypatch.delenv(k)
    # RT_LEAVE (read when a scene is uploaded): when a wave's node steps pause for its triangle
    # tests changes how the lanes interleave, not what a ray visits
    for leave in ("0", "8", "48", "64"):
        monkeypatch.setenv("RT_LEAVE", leave)
        s3 = scenes.build("spot")
        c = np.zeros((h, w, 3))
        assert s3.render(c, rt.render_params(width=w, height=h, spp=spp, seed=5, flags=_ffi.RT_FLAG_COUNT_WORK)) == rays
        assert c.tobytes() == base.tobytes(), leave
        assert (s3.last_stats().node_visits, s3.last_stats().tri_tests) == (st.node_visits, st.tri_tests), leave
        monkeypatch.delenv("RT_LEAVE")


def test_grid_nodes_and_binary32_nodes_give_the_same_bits(gpu_scenes, oracle, monkeypatch):
    """The launches over the ray queue walk the 16-bit grid copy of the tree when the scene
    qualifies for it, the binary32 nodes otherwise (RT_NO_GRID, read at upload, forces that): same
    image, same rays. A scene shrunk until its coordinates are ten times its extent still
    qualifies and must agree with the oracle; the same scene a thousand units off does not
    qualify, and its hit records must."""
    from test_grid_nodes import _moved

    w, h, spp = 72, 40, 5
    base = np.zeros((h, w, 3))
    rays = gpu_scenes("spot").render(base, rt.render_params(width=w, height=h, spp=spp, seed=5))
    monkeypatch.setenv("RT_NO_GRID", "1")
    s2 = scenes.build("spot")
    c = np.zeros((h, w, 3))
    assert s2.render(c, rt.render_params(width=w, height=h, spp=spp, seed=5)) == rays
    assert c.tobytes() == base.tobytes()
    monkeypatch.delenv("RT_NO_GRID")
    # the first stage walks the grid copy only for trees far larger than the caches (RT_QUANT_FIRST=1 forces it)
    monkeypatch.setenv("RT_QUANT_FIRST", "1")
    s3 = scenes.build("spot")
    c = np.zeros((h, w, 3))
    assert s3.render(c, rt.render_params(width=w, height=h, spp=spp, seed=5)) == rays
    assert c.tobytes() == base.tobytes()
    monkeypatch.delenv("RT_QUANT_FIRST")

    args = _moved("spot", scale=0.15, about=(0.0, 0.0, -3.0), drop_objects=(0,))
    small = rt.Scene.from_desc(*args)
    assert small.export_grid_bvh() is not None
    o = oracle.OracleScene.from_desc_arrays(*args)
    p = rt.render_params(width=96, height=72, spp=6, seed=3, fov_y_deg=8.0)
    got = np.zeros((72, 96, 3)); want = np.zeros((72, 96, 3))
    n_got = small.render(got, p)
    n_want, _ = o.render(want, p)
    assert n_got == n_want and n_got > 96 * 72 * 6 * 1.1, "the shrunk scene must be in view and scatter rays"
    assert rel_err(got, want) < IMG_RTOL

    args = _moved("spot", offset=(0.0, 1000.0, 0.0))
    far = rt.Scene.from_desc(*args)
    assert far.export_grid_bvh() is None
    o = oracle.OracleScene.from_desc_arrays(*args)
    rng = np.random.default_rng(23)
    org = rng.normal(size=(4000, 3)) * 2.0 + np.array([0.0, 1000.5, -2.5]); d = rng.normal(size=(4000, 3))
    hits = far.hit(org, d)
    assert hits.tobytes() == o.hit(org, d).tobytes() and np.count_nonzero(hits["obj"] >= 0) > 1000
    # and a frame of it (nothing but the floor's far edge in view: the queue launches still run)
    got = np.zeros((30, 40, 3)); want = np.zeros((30, 40, 3))
    p = rt.render_params(width=40, height=30, spp=2, seed=1)
    assert far.render(got, p) == o.render(want, p)[0]
    assert rel_err(got, want) < IMG_RTOL


def test_emissive_materials(oracle):
    """Emission travels with the path (src/main.rs:925-929): an emissive floor under a grey sphere
    and a glass sphere, every scheduling form against the oracle."""
    floor = rt.Opaque((0.8, 0.7, 0.6), 0.0, 1.0, 0.0, emission=(0.3, 0.2, 0.1))
    ball = rt.Opaque((0.5, 0.5, 0.9), 0.2, 0.8, 0.5, emission=(0.0, 0.1, 0.4))
    glass = rt.Dielectric(1.5)
    tv = np.array([[[30, -0.7, -30], [-30, -0.7, -30], [0, -0.7, 5]]], dtype=np.float64)
    tn = np.tile(np.array([0, 1.0, 0]), (1, 3, 1)); tuv = np.zeros((1, 3, 2))
    args = ([_ffi.RT_OBJ_MESH, _ffi.RT_OBJ_SPHERE, _ffi.RT_OBJ_SPHERE], [floor, ball, glass], tv, tn, tuv, [0], [0],
            [[0.5, 0, -3, 0.7], [-0.7, -0.2, -2.2, 0.5]], [1, 2])
    s = rt.Scene.from_desc(*args)
    o = oracle.OracleScene.from_desc_arrays(*args)
    w, h, spp = 64, 48, 4
    oc = np.zeros((h, w, 3))
    orays, _ = o.render(oc, rt.render_params(width=w, height=h, spp=spp, seed=6))
    for flags in (0, _ffi.RT_FLAG_MEGAKERNEL, _ffi.RT_FLAG_NO_REGEN):
        c = np.zeros((h, w, 3))
        assert s.render(c, rt.render_params(width=w, height=h, spp=spp, seed=6, flags=flags)) == orays
        assert rel_err(c, oc) < IMG_RTOL, flags


def test_tile_split_union_equals_whole(gpu_scenes):
    s = gpu_scenes("sphere_floor")
    w, h, spp = 200, 120, 3  # not multiples of the tile size
    whole = np.zeros((h, w, 3))
    rays = s.render(whole, rt.render_params(width=w, height=h, spp=spp, seed=1))
    parts = np.zeros((h, w, 3))
    total = 0
    for r in range(3):
        total += s.render(parts, rt.render_params(width=w, height=h, spp=spp, seed=1, tile_size=32, tile_rank=r, tile_ranks=3))
    assert total == rays and parts.tobytes() == whole.tobytes()


def test_sample_split_sums_to_whole(gpu_scenes):
    """Samples split over ranks: each renders its share with spp_total as the divisor."""
    s = gpu_scenes("spot")
    w, h = 64, 48
    whole = np.zeros((h, w, 3))
    s.render(whole, rt.render_params(width=w, height=h, spp=8, seed=3))
    parts = np.zeros((h, w, 3))
    for r in range(4):
        s.render(parts, rt.render_params(width=w, height=h, spp=2, first_sample=2 * r, spp_total=8, seed=3))
    assert np.allclose(parts, whole, rtol=1e-13, atol=1e-15)


def test_depth_cap_and_dof(gpu_scenes, oracle):
    s = gpu_scenes("spot")
    o = oracle.OracleScene.build("spot")
    for kw in (dict(max_depth=0), dict(max_depth=1), dict(max_depth=3), dict(dof_strength=0.05), dict(fov_y_deg=35.0)):
        p = rt.render_params(width=48, height=36, spp=2, seed=8, **kw)
        c = np.zeros((36, 48, 3)); oc = np.zeros((36, 48, 3))
        assert s.render(c, p) == o.render(oc, p)[0], kw
        assert rel_err(c, oc) < IMG_RTOL, kw


def test_parameter_errors(gpu_scenes):
    s = gpu_scenes("sphere_floor")
    c = np.zeros((8, 8, 3))
    for kw in (dict(max_depth=-1), dict(fov_y_deg=0.0), dict(dof_strength=-1.0), dict(tile_size=12, tile_ranks=2)):
        with pytest.raises(_ffi.RtError) as e:
            s.render(c, rt.render_params(width=8, height=8, spp=1, **kw))
        assert e.value.code == _ffi.RT_ERR_INVALID
    assert s.render(c, rt.render_params(width=8, height=8, spp=0)) == 0 and not c.any()


def test_screenshot_pin_gpu(gpu_scenes):
    """The HIP path's Spot image against the four screenshots the reference holds of this scene as it
    is at HEAD (three at 1600x1200, one at 640x480; tests/golden/make_golden.py): rendered at the
    screenshot's own size with 128 samples per pixel, quantised per pixel as the reference does
    (truncation, src/main.rs:32-35), averaged over the fixtures' 8x8 blocks."""
    from PIL import Image

    def blocks(c):
        h, w, _ = c.shape
        return (np.floor(np.clip(c, 0.0, 1.0) * 255.0) / 255.0).reshape(h // 8, 8, w // 8, 8, 3).mean(axis=(1, 3))

    def load(name):
        return np.asarray(Image.open(os.path.join(HERE, "golden", name)), dtype=np.float64) / 255.0

    c = np.zeros((1200, 1600, 3))
    gpu_scenes("spot").render(c, rt.render_params(width=1600, height=1200, spp=128, seed=3))
    img = blocks(c)
    for name in ("ref_screenshot_blocks_200x150.png", "ref_screenshot_170012_blocks_200x150.png", "ref_screenshot_164420_blocks_200x150.png"):
        ref = load(name)
        diff = img - ref
        print(f"{name}: mean |diff| {np.abs(diff).mean():.5f}, mean diff {diff.mean():+.5f}, p99 {np.percentile(np.abs(diff), 99):.4f}")
        # measured against 170941: 0.00138, -0.00010, 0.0082 (the reference's own screenshots differ by 0.0005 from each other)
        assert np.abs(diff).mean() < 0.0019, (name, np.abs(diff).mean())
        assert abs(diff.mean()) < 0.0005, (name, diff.mean())
        assert np.percentile(np.abs(diff), 99) < 0.013, name
        for ch in range(3):
            assert np.corrcoef(img[..., ch].ravel(), ref[..., ch].ravel())[0, 1] > 0.9993, name
    c = np.zeros((480, 640, 3))
    gpu_scenes("spot").render(c, rt.render_params(width=640, height=480, spp=128, seed=3))
    img, ref = blocks(c), load("ref_screenshot_163744_blocks_80x60.png")
    diff = img - ref
    print(f"640x480: mean |diff| {np.abs(diff).mean():.5f}, mean diff {diff.mean():+.5f}, p99 {np.percentile(np.abs(diff), 99):.4f}")
    assert np.abs(diff).mean() < 0.0025, np.abs(diff).mean()
    assert abs(diff.mean()) < 0.0006, diff.mean()
    for ch in range(3):
        assert np.corrcoef(img[..., ch].ravel(), ref[..., ch].ravel())[0, 1] > 0.999


def test_full_size_properties(gpu_scenes):
    """BASELINE configuration 2 at full size (1920x1080, 64 spp): determinism, sample-stream
    additivity and ray-count additivity — properties that need no CPU reference."""
    s = gpu_scenes("sphere_floor")
    w, h = 1920, 1080
    a = np.zeros((h, w, 3))
    rays = s.render(a, rt.render_params(width=w, height=h, spp=64, seed=0))
    b = np.zeros((h, w, 3))
    assert s.render(b, rt.render_params(width=w, height=h, spp=64, seed=0)) == rays
    assert sha(a) == sha(b)
    halves = np.zeros((h, w, 3))
    r1 = s.render(halves, rt.render_params(width=w, height=h, spp=32, first_sample=0, spp_total=64, seed=0))
    r2 = s.render(halves, rt.render_params(width=w, height=h, spp=32, first_sample=32, spp_total=64, seed=0))
    assert r1 + r2 == rays
    assert np.allclose(halves, a, rtol=1e-13, atol=1e-15)
    assert rays > w * h * 64 and np.isfinite(a).all() and a.min() >= 0.0


def test_default_scene_with_its_20000_pixel_texture(oracle):
    """The texture file the reference's default scene actually opens (src/main.rs:1273): 1.6 GB of
    texels in HBM instead of 4 MB in cache. Same bars as every other render."""
    s = scenes.build("spot_20k")
    s.upload(0)
    assert s.info().device_bytes > 1_600_000_000
    o = oracle.OracleScene.build("spot_20k")
    w, h, spp = 160, 120, 3
    p = rt.render_params(width=w, height=h, spp=spp, seed=9)
    c = np.zeros((h, w, 3)); oc = np.zeros((h, w, 3))
    rays = s.render(c, p)
    orays, _ = o.render(oc, p)
    assert rays == orays and rel_err(c, oc) < IMG_RTOL
    # not the small texture's image: the finer texels show
    s1 = scenes.build("spot")
    c1 = np.zeros((h, w, 3))
    assert s1.render(c1, p) == rays  # texels colour paths, they never steer them
    assert 1e-5 < np.abs(c1 - c).mean() < 0.02


def psnr(a, b, peak=1.0):
    mse = float(np.mean((np.clip(a, 0, peak) - np.clip(b, 0, peak)) ** 2))
    return np.inf if mse == 0.0 else 10.0 * np.log10(peak * peak / mse)


@pytest.mark.parametrize("name,w,h", [("spot", 160, 120), ("sphere_floor", 160, 90)])
def test_image_error_against_the_noise_floor(gpu_scenes, oracle, name, w, h):
    """Image agreement in the units a renderer is judged in. Two independent seeds at the same
    sample count differ by the Monte-Carlo noise (the floor any comparison of images has to live
    with); the HIP image and the oracle's image of the SAME seed must differ by nothing like that.
    Stated bound: RMSE(HIP, oracle) < 1e-12 (PSNR > 240 dB), against a floor of 20-35 dB."""
    s, o = gpu_scenes(name), oracle.OracleScene.build(name)

    def gpu(seed, spp):
        c = np.zeros((h, w, 3))
        s.render(c, rt.render_params(width=w, height=h, spp=spp, seed=seed))
        return c

    a16, b16 = gpu(11, 16), gpu(12, 16)
    oc = np.zeros((h, w, 3))
    o.render(oc, rt.render_params(width=w, height=h, spp=16, seed=11))
    rmse_same = float(np.sqrt(np.mean((a16 - oc) ** 2)))
    floor16 = float(np.sqrt(np.mean((a16 - b16) ** 2)))
    print(f"{name}: RMSE HIP-oracle {rmse_same:.2e} ({psnr(a16, oc):.0f} dB); two seeds at 16 spp {floor16:.4f} ({psnr(a16, b16):.1f} dB)")
    assert rmse_same < 1e-12 and psnr(a16, oc) > 240.0
    assert 15.0 < psnr(a16, b16) < 45.0
    # the floor is sampling noise and nothing else: it falls like 1/sqrt(spp), and the seeds agree in the mean
    a64, b64 = gpu(11, 64), gpu(12, 64)
    floor64 = float(np.sqrt(np.mean((a64 - b64) ** 2)))
    assert 1.6 < floor16 / floor64 < 2.4, (floor16, floor64)
    assert abs(a64.mean() - b64.mean()) < 4.0 * floor64 / np.sqrt(w * h)


@pytest.mark.parametrize("name,w,h,spp", [("ogre", 3840, 2160, 256), ("spot", 3840, 2160, 1024)])
def test_full_size_sample_additivity(gpu_scenes, name, w, h, spp):
    """BASELINE configurations 3 and 4 at full size (4K, 256 and 1024 spp; the calls run in several
    sample chunks with overlapped tails): a frame equals the sum of two calls that take half of its
    sample stream each — what the sample split over GPUs relies on — and the ray counts add up."""
    s = gpu_scenes(name)
    whole = np.zeros((h, w, 3))
    rays = s.render(whole, rt.render_params(width=w, height=h, spp=spp, seed=0))
    halves = np.zeros((h, w, 3))
    r1 = s.render(halves, rt.render_params(width=w, height=h, spp=spp // 2, first_sample=0, spp_total=spp, seed=0))
    r2 = s.render(halves, rt.render_params(width=w, height=h, spp=spp // 2, first_sample=spp // 2, spp_total=spp, seed=0))
    assert r1 + r2 == rays and rays > w * h * spp
    assert np.allclose(halves, whole, rtol=1e-13, atol=1e-15)
    assert np.isfinite(whole).all() and whole.min() >= 0.0 and 0.2 < whole.mean() < 2.0


def test_full_size_tile_union(gpu_scenes):
    """BASELINE configuration 5 at full size (1.15 M triangles, 7680x4320): the interleaved 64x64
    tiles of eight ranks are disjoint and their union is the whole frame, bit for bit."""
    s = gpu_scenes("spot_grid:14x14")
    w, h, spp = 7680, 4320, 1
    whole = np.zeros((h, w, 3))
    rays = s.render(whole, rt.render_params(width=w, height=h, spp=spp, seed=0))
    union = np.zeros((h, w, 3))
    parts = 0
    for r in range(8):
        before = np.count_nonzero(union.any(axis=2))
        parts += s.render(union, rt.render_params(width=w, height=h, spp=spp, seed=0, tile_size=64, tile_rank=r, tile_ranks=8))
        assert np.count_nonzero(union.any(axis=2)) - before > w * h // 10  # every rank owns about an eighth
    assert parts == rays
    assert sha(union) == sha(whole)


@pytest.mark.parametrize("name", ["spot", "sphere_floor", "ogre"])
def test_work_counters_equal_the_cpu_restatement(gpu_scenes, oracle, name):
    """The node visits and triangle tests the kernels count (what the roofline's algorithmic bytes
    are computed from) against the oracle's restatement of the same traversal over the same tree,
    ray for ray the same walk: equal up to the handful of box tests that the hardware reciprocal
    decides differently (the restatement perturbs the reciprocal by an ulp to stand in for it)."""
    s = gpu_scenes(name)
    o = oracle.OracleScene.build(name)
    co, cd = oracle.camera_rays(320, 240)
    rng = np.random.default_rng(1)
    org = np.concatenate([co, rng.normal(size=(50000, 3)) * 2 + np.array([0, 0, -3.0])])
    d = np.concatenate([cd, rng.normal(size=(50000, 3))])
    s.hit(org, d)
    st = s.last_stats()
    nodes, order = s.export_bvh()
    _, (node_visits, tri_tests) = o.bvh_hit(nodes, order, 0, org, d)
    assert st.rays == len(org)
    assert abs(st.node_visits - node_visits) <= 1e-4 * node_visits, (st.node_visits, node_visits)
    assert abs(st.tri_tests - tri_tests) <= 1e-4 * tri_tests, (st.tri_tests, tri_tests)


def test_million_triangle_scene_against_brute_force(gpu_scenes, oracle):
    """BASELINE configuration 5's scene (1.15 M triangles): the exhaustive binary64 test over every
    triangle, on a sample of camera and scattered rays small enough for the CPU, bit for bit."""
    s = gpu_scenes("spot_grid:14x14")
    o = oracle.OracleScene.build("spot_grid:14x14")
    assert o.counts()[0] == s.info().n_triangles == 1147777
    co, cd = oracle.camera_rays(32, 18)
    first = s.hit(co, cd)
    hit = first["tri"] >= 0
    rng = np.random.default_rng(4)
    d2 = rng.normal(size=(int(hit.sum()), 3))
    p2 = co[hit] + first["t"][hit, None] * cd[hit] + 1e-6 * d2
    org, d = np.concatenate([co, p2]), np.concatenate([cd, d2])
    got = s.hit(org, d)
    want = o.hit(org, d, oracle.HIT_BRUTE, threads=16)
    for f in ("obj", "tri", "t", "u", "v"):
        assert got[f].tobytes() == want[f].tobytes(), f
    assert (got["tri"] >= 0).sum() > 300
