"""Regenerates the fixtures in tests/golden/. Run from the repository root, in the build container
(needs /root/reference for the screenshot fixture only):

    python tests/golden/make_golden.py

Fixtures
  ref_screenshot_*blocks_*.png       8x8 block means of the reference's own committed outputs that show
                                     sphere_scene at HEAD (images/20220206_170941, _170012, _164420
                                     at 1600x1200, _163744 at 640x480): data derived from reference
                                     data files, the only artefacts of the reference that pin the
                                     oracle end to end.
  primary_<scene>_<w>x<h>.npz        primary-hit buffer of the oracle (brute-force Scene::hit on
                                     pixel-centre rays): sha256 of obj/tri/t/u/v plus a 16x16 crop.
  render_<scene>.npz                 the oracle's image (binary64 canvas) and ray count for a small
                                     render of each scene with the shared counter-based sampler:
                                     the vector the GPU path is held to (1e-12 relative) and
                                     that pins the oracle itself against drift (bitwise).
  kat.json                           analytic known-answer vectors (values derived by hand from
                                     the formulas of src/main.rs, not produced by the oracle).
"""
import hashlib
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


# The screenshots of the reference that show sphere_scene() as it is at HEAD (the four that
# tools/match_reference_images.py finds at correlation 0.998; no screenshot matches any other scene).
SCREENSHOTS = {
    "20220206_170941": "ref_screenshot_blocks_200x150.png",
    "20220206_170012": "ref_screenshot_170012_blocks_200x150.png",
    "20220206_164420": "ref_screenshot_164420_blocks_200x150.png",
    "20220206_163744": "ref_screenshot_163744_blocks_80x60.png",  # 640x480
}


def screenshot_blocks():
    from PIL import Image

    for stem, fixture in SCREENSHOTS.items():
        src = f"/root/reference/images/{stem}.png"
        if not os.path.exists(src):
            print("skip screenshot fixtures: reference not mounted")
            return
        im = np.asarray(Image.open(src).convert("RGB"), dtype=np.float64)  # [h][w][3]
        h, w, _ = im.shape
        blocks = im.reshape(h // 8, 8, w // 8, 8, 3).mean(axis=(1, 3))
        Image.fromarray(np.round(blocks).astype(np.uint8)).save(os.path.join(HERE, fixture))
        print("wrote", fixture)


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def primary(scene_name, w, h):
    import oracle_ffi as orc

    s = orc.OracleScene.build(scene_name)
    o, d = orc.camera_rays(w, h)
    hit = s.hit(o, d, orc.HIT_BRUTE)
    y0, x0 = h // 2 - 8, w // 2 - 8
    crop = hit.reshape(h, w)[y0:y0 + 16, x0:x0 + 16].copy()
    np.savez_compressed(
        os.path.join(HERE, f"primary_{scene_name}_{w}x{h}.npz"),
        sha_obj=sha(hit["obj"]), sha_tri=sha(hit["tri"]), sha_t=sha(hit["t"]), sha_u=sha(hit["u"]), sha_v=sha(hit["v"]),
        crop=crop, crop_origin=np.array([y0, x0]), hit_fraction=float((hit["obj"] >= 0).mean()),
    )
    print(f"wrote primary_{scene_name}_{w}x{h}.npz  hit fraction {(hit['obj'] >= 0).mean():.4f}")


RENDERS = {  # scene: (width, height, spp, seed)
    "spot": (48, 36, 2, 11),
    "lens": (40, 30, 2, 12),
    "sphere_floor": (48, 27, 3, 13),
    "ogre": (40, 22, 2, 14),
}


def render(scene_name):
    import oracle_ffi as orc
    import raytracer_amd as rt

    w, h, spp, seed = RENDERS[scene_name]
    s = orc.OracleScene.build(scene_name)
    c = np.zeros((h, w, 3))
    rays, _ = s.render(c, rt.render_params(width=w, height=h, spp=spp, seed=seed))
    np.savez_compressed(os.path.join(HERE, f"render_{scene_name}.npz"), canvas=c, rays=np.uint64(rays),
                        params=np.array([w, h, spp, seed], dtype=np.int64))
    print(f"wrote render_{scene_name}.npz  {rays} rays, mean {c.mean():.4f}")


def kat():
    # Hand-derived from src/main.rs formulas.
    k = {
        # floor triangle of triangle_floor.obj, ray straight down from the origin:
        # plane y = -0.7 -> t = 0.7; hit point (0,-0.7,0) = a + u*(b-a) + v*(c-a) with
        # a=(30,-.7,-30) b=(-30,-.7,-30) c=(0,-.7,5): z: -30 + 35 v = 0 -> v = 6/7;
        # x: 30 - 60 u - 30 v = 0 -> u = (30 - 180/7)/60 = 1/14.
        "floor_down": {"o": [0, 0, 0], "d": [0, -1, 0], "a": [30.0, -0.7, -30.0], "b": [-30.0, -0.7, -30.0],
                       "c": [0.0, -0.7, 5.0], "t": 0.7, "u": 1.0 / 14.0, "v": 6.0 / 7.0},
        # unit right triangle in z = -1, ray along -z through (0.25, 0.5): t=1,u=.25,v=.5
        "unit_tri": {"o": [0.25, 0.5, 0], "d": [0, 0, -1], "a": [0, 0, -1], "b": [1, 0, -1], "c": [0, 1, -1],
                     "t": 1.0, "u": 0.25, "v": 0.5},
        # same from behind (two-sided: det < 0 is accepted, src/main.rs:205)
        "unit_tri_back": {"o": [0.25, 0.5, -2], "d": [0, 0, 1], "a": [0, 0, -1], "b": [1, 0, -1], "c": [0, 1, -1],
                          "t": 1.0, "u": 0.25, "v": 0.5},
        # direction of length 2: distances are in units of |dir| (no normalisation in hit)
        "unit_tri_dir2": {"o": [0.25, 0.5, 0], "d": [0, 0, -2], "a": [0, 0, -1], "b": [1, 0, -1], "c": [0, 1, -1],
                          "t": 0.5, "u": 0.25, "v": 0.5},
        "misses": [
            # parallel to the plane: det == 0 (:205)
            {"o": [0, 0, 0], "d": [1, 0, 0], "a": [0, 0, -1], "b": [1, 0, -1], "c": [0, 1, -1]},
            # u < 0 (:218)
            {"o": [-0.1, 0.5, 0], "d": [0, 0, -1], "a": [0, 0, -1], "b": [1, 0, -1], "c": [0, 1, -1]},
            # v < 0 (:228)
            {"o": [0.5, -0.1, 0], "d": [0, 0, -1], "a": [0, 0, -1], "b": [1, 0, -1], "c": [0, 1, -1]},
            # u + v > 1 (:228)
            {"o": [0.75, 0.75, 0], "d": [0, 0, -1], "a": [0, 0, -1], "b": [1, 0, -1], "c": [0, 1, -1]},
            # behind the origin: dist < 0 (:234)
            {"o": [0.25, 0.5, -2], "d": [0, 0, -1], "a": [0, 0, -1], "b": [1, 0, -1], "c": [0, 1, -1]},
            # origin on the plane: dist == 0 is not > epsilon (:234)
            {"o": [0.25, 0.5, -1], "d": [0, 0, -1], "a": [0, 0, -1], "b": [1, 0, -1], "c": [0, 1, -1]},
        ],
        "edge_hits": [
            # exactly on the edge u = 0 and on the vertex a: inclusive bounds (:218, :228)
            {"o": [0.0, 0.5, 0], "d": [0, 0, -1], "a": [0, 0, -1], "b": [1, 0, -1], "c": [0, 1, -1], "t": 1.0, "u": 0.0, "v": 0.5},
            {"o": [0.0, 0.0, 0], "d": [0, 0, -1], "a": [0, 0, -1], "b": [1, 0, -1], "c": [0, 1, -1], "t": 1.0, "u": 0.0, "v": 0.0},
            {"o": [0.5, 0.5, 0], "d": [0, 0, -1], "a": [0, 0, -1], "b": [1, 0, -1], "c": [0, 1, -1], "t": 1.0, "u": 0.5, "v": 0.5},
        ],
        # sphere c=(0,0,-3) r=1: from the origin along -z the near root is 2; from the centre
        # the near root is -1 < epsilon so the far root 1 is returned; from beyond, none.
        "sphere": [
            {"c": [0, 0, -3], "r": 1.0, "o": [0, 0, 0], "d": [0, 0, -1], "t": 2.0},
            {"c": [0, 0, -3], "r": 1.0, "o": [0, 0, -3], "d": [0, 0, -1], "t": 1.0},
            {"c": [0, 0, -3], "r": 1.0, "o": [0, 0, -5], "d": [0, 0, -1], "t": None},
            {"c": [0, 0, -3], "r": 1.0, "o": [0, 2, 0], "d": [0, 0, -1], "t": None},
            # non-unit direction: a = 4, roots in units of |dir|
            {"c": [0, 0, -3], "r": 1.0, "o": [0, 0, 0], "d": [0, 0, -2], "t": 1.0},
        ],
        # interp_sky (:302-312): looking along the light direction t = 1 -> 1.5 everywhere;
        # opposite, t = 0 -> dir.y*c2 + (1-dir.y)*c1 with dir.y = -2/sqrt(6).
        "sky_light": {"d": [1 / 6 ** 0.5, 2 / 6 ** 0.5, 1 / 6 ** 0.5], "rgb": [1.5, 1.5, 1.5]},
        # straight up: d.L = 2/sqrt6; t = ((2/sqrt6+1)/2)^6; rgb = (1-t)*0.8 + t*1.5
        "sky_up": {"d": [0, 1, 0]},
        # reflectance (:329-333): at normal incidence (cosine 1) = ((1-n)/(1+n))^2 = 0.04 for 1.5
        "reflectance": [{"cos": 1.0, "ri": 1.5, "r": 0.04}, {"cos": 0.0, "ri": 1.5, "r": 1.0}],
        # refract (:320-327): normal incidence passes straight through
        "refract_normal": {"uv": [0, 0, 1], "n": [0, 0, 1], "eta": 1 / 1.5, "out": [0, 0, -1]},
    }
    with open(os.path.join(HERE, "kat.json"), "w") as f:
        json.dump(k, f, indent=1)
    print("wrote kat.json")


if __name__ == "__main__":
    screenshot_blocks()
    kat()
    primary("spot", 160, 120)
    primary("spot", 1600, 1200)
    primary("sphere_floor", 192, 108)
    primary("ogre", 160, 90)
    for name in RENDERS:
        render(name)
