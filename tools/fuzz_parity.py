#!/usr/bin/env python3
"""Random small renders, GPU against the oracle: scenes, image sizes (ragged tiles included),
samples, depth caps, depth of field, field of view, flags, chunking and tail thresholds drawn at
random. Run on a GPU box from the repo root:

    python tools/fuzz_parity.py [cases] [seed]
"""
import os
import sys

import numpy as np

sys.path.insert(0, "."); sys.path.insert(0, "tests")
import oracle_ffi as orc
import raytracer_amd as rt
from raytracer_amd import _ffi, scenes

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
names = ["spot", "lens", "sphere_floor", "ogre"]
gpu = {n: scenes.build(n) for n in names}
for s in gpu.values():
    s.upload(0)
cpu = {n: orc.OracleScene.build(n) for n in names}
worst = 0.0
for k in range(cases):
    name = names[rng.integers(0, 4)]
    w, h = int(rng.integers(1, 71)), int(rng.integers(1, 71))
    spp = int(rng.integers(1, 10))
    flags = int(rng.choice([0, 0, _ffi.RT_FLAG_NO_JITTER, _ffi.RT_FLAG_COUNT_WORK, _ffi.RT_FLAG_PIPELINE, _ffi.RT_FLAG_TIMING,
                            _ffi.RT_FLAG_MEGAKERNEL, _ffi.RT_FLAG_NO_REGEN]))
    p = rt.render_params(width=w, height=h, spp=spp, first_sample=int(rng.integers(0, 1000)), seed=int(rng.integers(0, 2**62)),
                         max_depth=int(rng.choice([0, 1, 2, 5, 200])), flags=flags, fov_y_deg=float(rng.uniform(20, 120)),
                         dof_strength=float(rng.choice([0.0, 0.0, 0.05])), spp_total=int(rng.choice([0, spp, 3 * spp])))
    env = {}
    if rng.random() < 0.4:
        env["RT_WORK_BUFFER_MIB"] = "1"
    if rng.random() < 0.5:
        env["RT_TAIL_THRESHOLD"] = str(int(rng.choice([1, 50, 500, 5000])))
    if rng.random() < 0.3:
        env["RT_FUSE_BOUNCES"] = str(int(rng.integers(0, 4)))
    if rng.random() < 0.2:
        env["RT_LONG_STEPS"] = str(int(rng.integers(1, 6)))
    for key in ("RT_WORK_BUFFER_MIB", "RT_TAIL_THRESHOLD", "RT_FUSE_BOUNCES", "RT_LONG_STEPS"):
        os.environ.pop(key, None)
    os.environ.update(env)
    base = rng.random((h, w, 3))
    c = base.copy(); oc = base.copy()
    rays = gpu[name].render(c, p)
    q = rt.render_params(width=w, height=h, spp=spp, first_sample=p.first_sample, seed=p.seed, max_depth=p.max_depth,
                         flags=flags & _ffi.RT_FLAG_NO_JITTER, fov_y_deg=p.fov_y_deg, dof_strength=p.dof_strength, spp_total=p.spp_total)
    orays, _ = cpu[name].render(oc, q)
    err = float(np.max(np.abs(c - oc) / np.maximum(1.0, np.abs(oc))))
    worst = max(worst, err)
    ok = rays == orays and err < 1e-12
    print(f"{k:3d} {name:12s} {w:2d}x{h:2d} spp {spp} depth {p.max_depth:3d} dof {p.dof_strength} flags {flags:2d} {env} rays {rays} err {err:.2e} {'ok' if ok else 'MISMATCH'}", flush=True)
    if not ok:
        sys.exit(1)
print(f"all {cases} cases agree; largest relative difference {worst:.2e}")
