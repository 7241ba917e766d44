#!/usr/bin/env python3
"""Prints the per-launch log (kind, items, rays, ms) of one frame, for each scheduling form."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import raytracer_amd as rt
from raytracer_amd import _ffi, scenes

ap = argparse.ArgumentParser()
ap.add_argument("--scene", default="sphere_floor")
ap.add_argument("--width", type=int, default=1920)
ap.add_argument("--height", type=int, default=1080)
ap.add_argument("--spp", type=int, default=64)
ap.add_argument("--reps", type=int, default=3)
ap.add_argument("--forms", default="wave,mega,noregen")
ap.add_argument("--count", action="store_true")
ap.add_argument("--show", type=int, default=14)
ap.add_argument("--device-build", nargs="?", const="default", default=None, choices=["default", "lbvh", "ploc"],
                help="replace the host-built tree by one built on the GPU")
a = ap.parse_args()
s = scenes.build(a.scene)
if a.device_build:
    import time as _t
    for _ in range(2):
        t0 = _t.perf_counter()
        s.rebuild_on_device(0, None if a.device_build == "default" else a.device_build)
        print(f"device build ({a.device_build}): {(_t.perf_counter() - t0) * 1e3:.1f} ms")
s.upload(0)
print(f"tree: {s.info().bvh_nodes} nodes, {s.info().bvh_leaves} leaves, depth {s.info().bvh_depth}")
canvas = torch.zeros((a.height, a.width, 3), dtype=torch.float64, device="cuda")
KIND = {0: "first", 1: "bounce", 2: "tail", 3: "single", 4: "resolve", 5: "continue"}
FORMS = {"wave": 0, "mega": _ffi.RT_FLAG_MEGAKERNEL, "noregen": _ffi.RT_FLAG_NO_REGEN}
for form in a.forms.split(","):
    flags = FORMS[form] | _ffi.RT_FLAG_TIMING | (_ffi.RT_FLAG_COUNT_WORK if a.count else 0)
    for rep in range(a.reps):
        import time
        torch.cuda.synchronize(); t0 = time.perf_counter()
        s.render_device(canvas.data_ptr(), rt.render_params(width=a.width, height=a.height, spp=a.spp, seed=0, first_sample=rep * a.spp, flags=flags))
        torch.cuda.synchronize(); wall = (time.perf_counter() - t0) * 1e3
    log = s.last_launches()
    tm = s.last_timing()
    rays = sum(l.rays for l in log)
    print(f"== {form}: wall {wall:.2f} ms, kernels {tm.ms_paths:.2f} + resolve {tm.ms_resolve:.2f} ms, {rays} rays, {rays / wall / 1e3:.0f} Mrays/s, {len(log)} launches")
    for l in log[:a.show]:
        extra = (f" nodes/ray {l.node_visits / max(1, l.rays):.2f} tris/ray {l.tri_tests / max(1, l.rays):.2f} | per wave: inner {l.wave_iters_inner / max(1, l.waves):.1f}"
                 f" leaf {l.wave_iters_leaf / max(1, l.waves):.1f} reject {l.wave_iters_reject / max(1, l.waves):.1f}"
                 f" | lanes busy: node steps {l.node_visits / max(1, 64 * l.wave_iters_inner):.2f} triangle tests {l.tri_tests / max(1, 64 * l.wave_iters_leaf):.2f}") if a.count else ""
        print(f"   {KIND[l.kind]:8s} items {l.items:>10d} rays {l.rays:>10d} {l.ms:8.3f} ms  {l.rays / max(l.ms, 1e-6) / 1e3:8.0f} Mrays/s{extra}")
