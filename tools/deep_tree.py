#!/usr/bin/env python3
"""Frame time over a deep tree: a soup of large, mutually overlapping triangles in front of the
camera, whose bottom-up clustering (PLOC) chains the clusters into a tree several dozen levels deep,
rendered at 1920x1080. The traversal stack is sized from the tree's depth, so this is the scene on
which a full stack in LDS costs occupancy.

    python tools/deep_tree.py [--triangles 3000] [--spp 8] [--frames 5]
"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import raytracer_amd as rt
from raytracer_amd import _ffi

ap = argparse.ArgumentParser()
ap.add_argument("--triangles", type=int, default=3000)
ap.add_argument("--spp", type=int, default=8)
ap.add_argument("--frames", type=int, default=5)
ap.add_argument("--builder", default="ploc")
a = ap.parse_args()
rng = np.random.default_rng(7)
n = a.triangles
# thin slivers through a common region: every bounding box overlaps most of the others
c = rng.normal(size=(n, 1, 3)) * 0.05 + np.array([0.0, 0.0, -3.0])
tv = c + rng.normal(size=(n, 3, 3)) * np.array([1.0, 1.0, 0.3]) * 10.0 ** rng.uniform(-2.0, 0.0, (n, 1, 1))
mat = rt.Opaque((0.6, 0.6, 0.6), 0, 0, 0.9)
nrm = np.cross(tv[:, 1] - tv[:, 0], tv[:, 2] - tv[:, 0])
nrm /= np.maximum(1e-30, np.linalg.norm(nrm, axis=1, keepdims=True))
s = rt.Scene.from_desc([_ffi.RT_OBJ_MESH], [mat], tv, np.repeat(nrm[:, None, :], 3, axis=1), np.zeros((n, 3, 2)), [0] * n, list(range(n)),
                       np.zeros((0, 4)), [])
s.rebuild_on_device(0, a.builder)
i = s.info()
W, H = 1920, 1080
canvas = torch.zeros((H, W, 3), dtype=torch.float64, device="cuda")
times = []
for f in range(a.frames + 2):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    s.render_device(canvas.data_ptr(), rt.render_params(width=W, height=H, spp=a.spp, seed=0, first_sample=f * a.spp))
    s.wait()
    torch.cuda.synchronize()
    times.append((time.perf_counter() - t0) * 1e3)
st = s.last_stats()
lds = (i.bvh_depth + 2) * 256
print(f"{n} triangles, {a.builder} tree: {i.bvh_nodes} nodes, depth {i.bvh_depth} (full stack: {lds} B of LDS per wave, {min(32, 160 * 1024 // lds)} waves per CU by LDS); "
      f"{W}x{H}x{a.spp}: {np.median(times[2:]):.2f} ms per frame, {st.rays} rays, {st.rays / np.median(times[2:]) / 1e3:.0f} Mrays/s, "
      f"{st.node_visits / max(1, st.rays):.1f} node visits per ray" if st.node_visits else
      f"{n} triangles, {a.builder} tree: {i.bvh_nodes} nodes, depth {i.bvh_depth} (full stack: {lds} B of LDS per wave, {min(32, 160 * 1024 // lds)} waves per CU by LDS); "
      f"{W}x{H}x{a.spp}: {np.median(times[2:]):.2f} ms per frame, {st.rays} rays, {st.rays / np.median(times[2:]) / 1e3:.0f} Mrays/s")
