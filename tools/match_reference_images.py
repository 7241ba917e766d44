#!/usr/bin/env python3
"""Which of the reference's committed screenshots (images/*.png: no spp, no seed, many of them of
scenes and code states long gone) show the scenes this repository renders? Renders each scene
with the CPU restatement (oracle/), reduces it and every 4:3 screenshot to 100 x 75 block means
and lists, per scene, the screenshots by correlation. Needs /root/reference (build container
only); writes profiles/reference_image_matches.json.

    python tools/match_reference_images.py [--spp 8]
"""
import argparse
import glob
import json
import os
import sys

import numpy as np
from PIL import Image

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
IMAGES = "/root/reference/images"


def blocks(img):
    """[300][400][3] in 0..1 -> [75][100][3] block means."""
    return img.reshape(75, 4, 100, 4, 3).mean(axis=(1, 3))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--spp", type=int, default=8)
    a = ap.parse_args()
    import oracle_ffi as orc
    import raytracer_amd as rt

    shots = {}
    for f in sorted(glob.glob(os.path.join(IMAGES, "*.png"))):
        with Image.open(f) as im:
            w, h = im.size
            if w * 3 != h * 4 or w < 400:
                continue
            shots[os.path.basename(f)] = blocks(np.asarray(im.convert("RGB").resize((400, 300), Image.BOX), dtype=np.float64) / 255.0)
    print(f"{len(shots)} screenshots of aspect 4:3 and at least 400 pixels wide")
    out = {"_doc": "Per scene: the reference screenshots (images/*.png) that correlate best with the CPU restatement's render, "
                   "both as 100x75 block means (correlation over all three channels; mean absolute difference in 0..1). "
                   f"Renders: 400x300, {a.spp} spp, seed 3, quantised as src/main.rs:32-35. Made by tools/match_reference_images.py."}
    for scene in ("spot", "lens", "sphere_floor", "ogre"):
        c = np.zeros((300, 400, 3))
        orc.OracleScene.build(scene).render(c, rt.render_params(width=400, height=300, spp=a.spp, seed=3), threads=8)
        mine = blocks(np.floor(np.clip(c, 0.0, 1.0) * 255.0) / 255.0)
        rows = []
        for name, b in shots.items():
            rows.append((float(np.corrcoef(mine.ravel(), b.ravel())[0, 1]), float(np.abs(mine - b).mean()), name))
        rows.sort(reverse=True)
        out[scene] = [{"image": n, "corr": round(r, 5), "mean_abs_diff": round(d, 5)} for r, d, n in rows[:5]]
        print(scene)
        for r, d, n in rows[:5]:
            print(f"   {n:28s} corr {r:.5f}  mean |diff| {d:.5f}")
    json.dump(out, open(os.path.join(ROOT, "profiles", "reference_image_matches.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
