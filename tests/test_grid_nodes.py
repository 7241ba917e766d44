"""The half-size copy of the tree (16-bit grid planes) that the launches over the ray queue walk.

A grid box must contain the binary32 box it stands for with a margin that covers the kernels' own
binary32 arithmetic, or a scattered ray could be culled from a box it enters; scenes for which
that cannot be shown (far from the origin relative to their size) must not get the copy at all.
No GPU: the copy is made on the host and exported."""
import numpy as np
import pytest

import raytracer_amd as rt
from raytracer_amd import _ffi, scenes


def _planes(scene):
    """binary32 planes [n][3 axes][4: c0.min c0.max c1.min c1.max] and the grid's q, lo, step."""
    nodes, _ = scene.export_bvh()
    f = nodes[:, :12].copy().view(np.float32).reshape(-1, 3, 4)
    g = scene.export_grid_bvh()
    if g is None:
        return f, None, None, None
    qn, lo, step = g
    q16 = qn[:, :6].copy().view(np.uint16).reshape(-1, 3, 4)
    assert np.array_equal(qn[:, 6:8].view(np.int32), nodes[:, 12:14].view(np.int32)), "child refs differ"
    return f, q16, lo, step


def _moved(name, offset=(0.0, 0.0, 0.0), scale=1.0, about=(0.0, 0.0, 0.0), drop_objects=()):
    """The scene `name` scaled about a point and then moved, built through rt_scene_create."""
    s = scenes.build(name)
    tv, tn, tuv, tobj, tidx = s.export_flat()
    kind, mats, sph, sobj = s.export_objects()
    keep = ~np.isin(tobj, list(drop_objects))
    about = np.asarray(about, dtype=np.float64)
    offset = np.asarray(offset, dtype=np.float64)
    tv = (tv[keep] - about) * scale + about + offset
    sph = sph.copy()
    sph[:, :3] = (sph[:, :3] - about) * scale + about + offset
    sph[:, 3] *= scale
    tex = [s.export_texture(i) for i in range(s.info().n_textures)]
    return (kind, list(mats), tv, tn[keep], tuv[keep], tobj[keep], tidx[keep], sph, sobj, tex)


@pytest.mark.parametrize("name", ["spot", "sphere_floor", "ogre", "lens"])
def test_grid_boxes_contain_the_binary32_boxes(name):
    f, q, lo, step = _planes(scenes.build(name))
    assert q is not None, "the reference's scenes sit at the origin and qualify for the grid"
    lo64, st64 = lo.astype(np.float64)[None, :, None], step.astype(np.float64)[None, :, None]
    plane = lo64 + q.astype(np.float64) * st64
    is_max = np.array([False, True, False, True])[None, None, :]
    gap = np.where(is_max, plane - f, f - plane)
    assert np.all(gap >= st64 * (1 - 1e-9)), "a grid plane lies less than one step outside its binary32 plane"
    # the scene's largest coordinate is within 2^20 steps: what bounds the kernels' rounding
    assert np.all(np.abs(f).max(axis=(0, 2)) <= step.astype(np.float64) * 2.0**20)


def _slab_f32(planes, ix, m_hi, m_lo):
    """{min, max} * inv + {m_hi, m_lo} in binary32 with one rounding per plane (fused), per axis."""
    t_min = (planes[..., 0].astype(np.float64) * ix.astype(np.float64) + m_hi.astype(np.float64)).astype(np.float32)
    t_max = (planes[..., 1].astype(np.float64) * ix.astype(np.float64) + m_lo.astype(np.float64)).astype(np.float32)
    return np.minimum(t_min, t_max), np.maximum(t_min, t_max)


@pytest.mark.parametrize("name", ["spot", "sphere_floor"])
def test_grid_intervals_contain_the_binary32_intervals_in_the_kernels_arithmetic(name):
    """The per-axis entry/exit distances as trav_set_ray + trav_run form them, in emulated binary32:
    the grid's interval must contain the binary32 nodes' for every box and ray."""
    f, q, lo, step = _planes(scenes.build(name))
    rng = np.random.default_rng(5)
    n_rays = 300
    pick = rng.integers(0, f.shape[0], size=n_rays)
    centre = np.array([0.0, 0.5, -2.5])
    org = np.concatenate([rng.normal(size=(n_rays // 2, 3)) * 2.0 + centre, rng.normal(size=(n_rays - n_rays // 2, 3)) * 40.0])
    d = rng.normal(size=(n_rays, 3))
    d[::7, 0] = 0.0  # axis-parallel directions
    d[::11, 1] = 1e-12
    f32 = np.float32
    for a in range(3):
        o = org[:, a]
        dd = d[:, a].copy()
        dd[np.abs(dd) < 1e-30] = 1e-30
        omax = np.abs(org).max(axis=1)
        pad = np.nextafter((omax * 2.0**-21).astype(f32), f32(np.inf))
        ox = o.astype(f32)
        ix = (f32(1.0) / dd.astype(f32)).astype(f32)
        m_lo = (-(ox - pad) * ix).astype(f32)
        m_hi = (-(ox + pad) * ix).astype(f32)
        for child in range(2):
            pf = f[pick, a, 2 * child:2 * child + 2]
            lo_f, hi_f = _slab_f32(pf, ix, m_hi, m_lo)
            # grid form: m += lo * inv (one rounding for the product, one for the sum), inv *= step
            prod = (lo[a] * ix).astype(f32)
            gm_lo = (m_lo + prod).astype(f32)
            gm_hi = (m_hi + prod).astype(f32)
            gix = (ix * step[a]).astype(f32)
            pq = q[pick, a, 2 * child:2 * child + 2].astype(f32)
            lo_q, hi_q = _slab_f32(pq, gix, gm_hi, gm_lo)
            ok = np.isfinite(lo_f) & np.isfinite(hi_f) & np.isfinite(lo_q) & np.isfinite(hi_q)
            assert np.all(lo_q[ok] <= lo_f[ok]) and np.all(hi_q[ok] >= hi_f[ok])


def test_scenes_far_from_the_origin_keep_their_binary32_nodes():
    # the whole Spot scene a thousand units away: its y extent is two units
    far = rt.Scene.from_desc(*_moved("spot", offset=(0.0, 1000.0, 0.0)))
    assert far.export_grid_bvh() is None
    # bounds [1000, 1001] and [5000, 5000.5]: the cases in which a clamped top plane would fall inside the box
    for x0, w in ((1000.0, 1.0), (5000.0, 0.5)):
        n, h = 16, w / 64
        x = x0 + np.linspace(0.0, w - h, n)
        tv = np.stack([np.stack([x, np.zeros(n), np.full(n, -3.0)], 1), np.stack([x + h, np.zeros(n), np.full(n, -3.0)], 1),
                       np.stack([x, np.full(n, h), np.full(n, -3.0 - h)], 1)], 1)
        tn = np.tile(np.array([0, 0, 1.0]), (n, 3, 1)); tuv = np.zeros((n, 3, 2))
        two = rt.Scene.from_desc([_ffi.RT_OBJ_MESH], [rt.Opaque((0.5, 0.5, 0.5), 0, 1, 0)], tv, tn, tuv, [0] * n, list(range(n)), np.zeros((0, 4)), [])
        assert two.info().bvh_nodes >= 1
        assert two.export_grid_bvh() is None


def test_a_small_scene_near_the_limit_still_qualifies_and_is_contained():
    # the cow and the spheres shrunk to a seventh about a point three units in front of the camera:
    # coordinates about ten times the extent
    s = rt.Scene.from_desc(*_moved("spot", scale=0.15, about=(0.0, 0.0, -3.0), drop_objects=(0,)))
    f, q, lo, step = _planes(s)
    assert q is not None
    ratio = np.abs(f).max(axis=(0, 2)) / (f.max(axis=(0, 2)) - f.min(axis=(0, 2)))
    assert ratio.max() > 5.0
    plane = lo.astype(np.float64)[None, :, None] + q.astype(np.float64) * step.astype(np.float64)[None, :, None]
    is_max = np.array([False, True, False, True])[None, None, :]
    gap = np.where(is_max, plane - f, f - plane)
    assert np.all(gap >= step.astype(np.float64)[None, :, None] * (1 - 1e-9))
