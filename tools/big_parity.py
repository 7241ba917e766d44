#!/usr/bin/env python3
"""Larger parity sweep than the test suite makes: four scenes, 0.4-1.9 M rays each, GPU image and
ray count against the brute-force oracle (about 15 s of CPU). Run on a GPU box from the repo root:

    python tools/big_parity.py
"""
import sys, time, numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import raytracer_amd as rt, oracle_ffi as orc
from raytracer_amd import scenes
for name, w, h, spp in (("spot", 400, 300, 6), ("ogre", 320, 180, 4), ("lens", 320, 240, 4), ("sphere_floor", 480, 270, 8)):
    s = scenes.build(name); s.upload(0)
    o = orc.OracleScene.build(name)
    p = rt.render_params(width=w, height=h, spp=spp, seed=77)
    c = np.zeros((h, w, 3)); oc = np.zeros((h, w, 3))
    t0 = time.time(); rays = s.render(c, p); t1 = time.time()
    orays, _ = o.render(oc, p, threads=16); t2 = time.time()
    err = np.max(np.abs(c - oc) / np.maximum(1.0, np.abs(oc)))
    print(name, rays, orays, rays == orays, "max rel err %.3g" % err, "gpu %.2fs oracle %.1fs" % (t1 - t0, t2 - t1), flush=True)
    assert rays == orays and err < 1e-12
print("ok")
