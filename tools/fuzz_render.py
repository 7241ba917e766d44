#!/usr/bin/env python3
"""Many randomly made scenes (the generator of tests/test_gpu_fuzz_render.py) rendered on the GPU and
by the oracle: equal ray counts, images within 1e-12, nearest hits of the camera's rays bit for bit.
Stops at the first difference and prints the seed.

    python tools/fuzz_render.py [scenes] [first seed]
"""
import sys
import time

import numpy as np

sys.path.insert(0, "."); sys.path.insert(0, "tests")
import oracle_ffi as oracle
import raytracer_amd as rt
from raytracer_amd import _ffi
from test_gpu_fuzz_render import _random_scene

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100
first = int(sys.argv[2]) if len(sys.argv) > 2 else 5000
worst, rays_total, t0 = 0.0, 0, time.time()
for seed in range(first, first + n):
    args = _random_scene(seed)
    s = rt.Scene.from_desc(*args)
    o = oracle.OracleScene.from_desc_arrays(*args)
    co, cd = oracle.camera_rays(48, 36)
    if s.hit(co, cd).tobytes() != o.hit(co, cd, oracle.HIT_BRUTE).tobytes():
        sys.exit(f"seed {seed}: nearest hits differ")
    w, h, spp = 48, 36, 3
    p = dict(width=w, height=h, spp=spp, seed=seed, dof_strength=(0.0, 0.05)[seed & 1], max_depth=(200, 7)[seed % 5 == 0])
    want = np.zeros((h, w, 3))
    orays, _ = o.render(want, rt.render_params(**p))
    for flags in (0, _ffi.RT_FLAG_MEGAKERNEL, _ffi.RT_FLAG_NO_REGEN)[: 3 if seed % 4 == 0 else 1]:
        got = np.zeros((h, w, 3))
        rays = s.render(got, rt.render_params(flags=flags, **p))
        err = float(np.max(np.abs(got - want) / np.maximum(1.0, np.abs(want))))
        if rays != orays or not err < 1e-12:
            sys.exit(f"seed {seed} flags {flags}: rays {rays} against {orays}, image error {err}")
        worst = max(worst, err)
    rays_total += orays
    if (seed - first) % 20 == 19:
        print(f"{seed - first + 1} scenes, {rays_total} rays, worst image error {worst:.2e}, {time.time() - t0:.0f} s", flush=True)
print(f"{n} scenes from seed {first}: ray counts equal, nearest hits bit for bit, worst image error {worst:.2e} (bar 1e-12); {rays_total} rays")
