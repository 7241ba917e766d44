#!/usr/bin/env python3
"""Per-ray traversal work of every benchmark scene from the CPU restatement of the kernels' traversal
(oracle/), on the pixel-centre camera rays of a 480x270 frame and on rays scattered from their
hit points: node visits and triangle tests per ray, and the algorithmic bytes and flops per ray
that follow (64 B per node, 192 B per triangle, 88 B per ray; 24, 52 and 30 flops per node,
triangle and sphere). No GPU needed. Writes profiles/per_ray_cpu_restatement.json."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np

import oracle_ffi as orc
from raytracer_amd import scenes

out = {"_doc": __doc__.strip()}
for name in ("spot", "sphere_floor", "ogre", "lens", "spot_grid:14x14"):
    s = scenes.build(name)
    o = orc.OracleScene.build(name)
    nodes, order = s.export_bvh()
    info = s.info()
    co, cd = orc.camera_rays(480, 270)
    hits, (nv, tt) = o.bvh_hit(nodes, order, 0, co, cd)
    entry = {"triangles": int(info.n_triangles), "tree_nodes": int(info.bvh_nodes), "spheres": int(info.n_spheres)}

    def per_ray(nv, tt, n):
        nn, nt = nv / n, tt / n
        return {"rays": int(n), "nodes_per_ray": round(nn, 4), "tris_per_ray": round(nt, 4),
                "bytes_per_ray": round(64 * nn + 192 * nt + 88, 1), "flops_per_ray": round(24 * nn + 52 * nt + 30 * info.n_spheres, 1)}

    entry["primary"] = per_ray(nv, tt, len(co))
    hit = hits["tri"] >= 0
    rng = np.random.default_rng(0)
    p = co[hit] + hits["t"][hit, None] * cd[hit]
    d = rng.normal(size=p.shape)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    p = p + 1e-6 * d
    _, (nv2, tt2) = o.bvh_hit(nodes, order, 0, p, d)
    entry["scattered"] = per_ray(nv2, tt2, len(p))
    out[name] = entry
    print(name, entry["primary"], entry["scattered"])
json.dump(out, open(os.path.join(ROOT, "profiles", "per_ray_cpu_restatement.json"), "w"), indent=1)
