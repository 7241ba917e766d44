#!/usr/bin/env python3
"""Random triangle soups and spheres, batch nearest-hit queries on the GPU (host-built tree and
device-built tree) against the oracle's brute force: clustered and scattered triangles, slivers,
zero-area and repeated triangles, coordinates from 1e-6 to 1e6. Bit-exact or it stops.

    python tools/fuzz_trace.py [cases] [seed]
"""
import sys

import numpy as np

sys.path.insert(0, "."); sys.path.insert(0, "tests")
import oracle_ffi as orc
import raytracer_amd as rt
from raytracer_amd import _ffi

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
mat = rt.Opaque((0.5, 0.5, 0.5), 0, 1, 0)
for k in range(cases):
    n = int(rng.choice([1, 2, 4, 5, 17, 100, 1000, 5000]))
    scale = float(10.0 ** rng.integers(-6, 7))
    centres = rng.normal(size=(n, 1, 3)) * scale * (1.0 if rng.random() < 0.5 else 0.01)
    if rng.random() < 0.3:  # a few tight clusters
        centres = centres[rng.integers(0, max(1, n // 50), n)]
    size = scale * 10.0 ** rng.uniform(-4, 0, (n, 1, 1))
    tv = centres + rng.normal(size=(n, 3, 3)) * size
    kind = rng.random(n)
    tv[kind < 0.05, 2] = tv[kind < 0.05, 1]          # zero area: two equal corners
    tv[(kind >= 0.05) & (kind < 0.1), 2] = 0.5 * (tv[(kind >= 0.05) & (kind < 0.1), 0] + tv[(kind >= 0.05) & (kind < 0.1), 1])  # collinear
    rep = rng.random(n) < 0.1
    tv[rep] = tv[rng.integers(0, n, int(rep.sum()))]  # repeated triangles: distance ties
    n_sph = int(rng.integers(0, 4))
    spheres = np.concatenate([rng.normal(size=(n_sph, 3)) * scale, np.abs(rng.normal(size=(n_sph, 1))) * scale * 0.3], axis=1)
    kinds = [_ffi.RT_OBJ_MESH] + [_ffi.RT_OBJ_SPHERE] * n_sph
    args = (kinds, [mat] * len(kinds), tv, np.tile(np.array([0, 0, 1.0]), (n, 3, 1)), np.zeros((n, 3, 2)), [0] * n, list(range(n)),
            spheres, list(range(1, n_sph + 1)))
    host = rt.Scene.from_desc(*args)
    dev = rt.Scene.from_desc(*args)
    dev.rebuild_on_device(0, ("ploc", "lbvh")[k % 2])  # both device builders take turns
    o = orc.OracleScene.from_desc_arrays(*args)
    m = 4000
    org = rng.normal(size=(m, 3)) * scale * 3.0
    tgt = tv.reshape(-1, 3)[rng.integers(0, 3 * n, m)] + rng.normal(size=(m, 3)) * scale * 10.0 ** rng.uniform(-8, -1, (m, 1))
    d = (tgt - org) * 10.0 ** rng.uniform(-3, 3, (m, 1))
    half = m // 2  # the other half starts on a corner, an edge or inside a triangle
    w = rng.random((half, 3)); w /= w.sum(axis=1, keepdims=True)
    w[rng.random(half) < 0.3, 2] = 0.0
    t = tv[rng.integers(0, n, half)]
    org[:half] = (w[:, :, None] * t).sum(axis=1)
    want = o.hit(org, d, orc.HIT_BRUTE)
    for what, s in (("host tree", host), ("device tree", dev)):
        got = s.hit(org, d)
        if got.tobytes() != want.tobytes():
            bad = np.nonzero([got[i].tobytes() != want[i].tobytes() for i in range(m)])[0]
            print(f"case {k}: {what} differs on {len(bad)} rays, first {bad[0]}: got {got[bad[0]]} want {want[bad[0]]}")
            sys.exit(1)
    i = dev.info()
    print(f"{k:3d} n {n:5d} spheres {n_sph} scale {scale:g}: {int((want['obj'] >= 0).sum())} of {m} rays hit; device tree {i.bvh_nodes} nodes depth {i.bvh_depth}", flush=True)
print(f"all {cases} cases bit-exact")
