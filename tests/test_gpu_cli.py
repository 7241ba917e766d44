"""The command-line driver (the reference's main() without the window, src/main.rs:1054-1128)
against the same loop made through the Python binding and against the oracle, on a real GPU."""
import os
import subprocess

import numpy as np
import pytest

import raytracer_amd as rt

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLI = os.path.join(ROOT, "raytracer_amd", "rt_render")


def test_cli_image_equals_library_loop_and_oracle(tmp_path, oracle):
    assert os.path.exists(CLI), "raytracer_amd/rt_render is not built (make -C raytracer_amd/csrc)"
    w, h, spp, its, seed = 96, 72, 3, 4, 7
    out = tmp_path / "spot.png"
    scene_file = os.path.join(ROOT, "scenes", "spot.scene")
    r = subprocess.run([CLI, scene_file, "-o", str(out), "--width", str(w), "--height", str(h), "--spp", str(spp),
                        "--its", str(its), "--seed", str(seed)], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    assert "5857 triangles, 6 spheres" in r.stderr and "Mrays/s" in r.stderr
    got = rt.load_png(out)

    scene, p, _ = rt.Scene.from_file(scene_file)
    scene.upload(0)
    o = oracle.OracleScene.build("spot")
    c = np.zeros((h, w, 3)); oc = np.zeros((h, w, 3))
    for it in range(its):  # src/main.rs:1099-1100: one render call per event, its += 1
        q = rt.render_params(width=w, height=h, spp=spp, first_sample=it * spp, seed=seed)
        scene.render(c, q)
        o.render(oc, q)
    assert np.array_equal(got, rt.quantize_rgba8(c, its))
    # against the oracle the canvas agrees to 1e-12, so a byte can differ only where canvas/its*255
    # sits within that distance of an integer: allow a handful of one-step differences
    d = np.abs(got.astype(int) - rt.quantize_rgba8(oc, its).astype(int))
    assert d.max() <= 1 and (d > 0).sum() <= 3
    assert got[..., :3].mean() > 40 and np.all(got[..., 3] == 255)


def test_cli_progress_and_errors(tmp_path):
    out = tmp_path / "p.png"
    scene_file = os.path.join(ROOT, "scenes", "sphere_floor.scene")
    r = subprocess.run([CLI, scene_file, "-o", str(out), "--width", "64", "--height", "36", "--spp", "2", "--its", "2",
                        "--progress"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    assert rt.load_png(out).shape == (36, 64, 4)
    r = subprocess.run([CLI, str(tmp_path / "none.scene")], capture_output=True, text=True, timeout=60)
    assert r.returncode == 1 and "cannot open" in r.stderr
    r = subprocess.run([CLI, scene_file, "--bogus"], capture_output=True, text=True, timeout=60)
    assert r.returncode == 2 and "usage" in r.stderr
    # the tree built on the GPU instead of the host: the same picture
    a, b = tmp_path / "a.png", tmp_path / "b.png"
    common = [scene_file, "--width", "64", "--height", "36", "--spp", "2", "--its", "2"]
    assert subprocess.run([CLI] + common + ["-o", str(a)], capture_output=True, text=True, timeout=120).returncode == 0
    r = subprocess.run([CLI] + common + ["-o", str(b), "--device-tree"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    assert np.array_equal(rt.load_png(a), rt.load_png(b))


@pytest.mark.parametrize("split", ["samples", "tiles"])
def test_cli_on_a_group_of_members(tmp_path, split):
    """--devices: the same frames from a group (members sharing the one GPU of the test box). A byte may
    differ from the single-GPU image only where a channel sits within 1e-7 of a quantisation step."""
    scene_file = os.path.join(ROOT, "scenes", "spot.scene")
    common = ["--width", "120", "--height", "80", "--spp", "6", "--its", "3", "--seed", "2"]
    one, grp = tmp_path / "one.png", tmp_path / "grp.png"
    r = subprocess.run([CLI, scene_file, "-o", str(one)] + common, capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    r = subprocess.run([CLI, scene_file, "-o", str(grp), "--devices", "0,0,0", "--split", split] + common, capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    assert r.stderr.count("member ") == 3
    a, b = rt.load_png(one).astype(int), rt.load_png(grp).astype(int)
    d = np.abs(a - b)
    assert d.max() <= 1 and (d > 0).sum() <= 3
    rays = [line.split(" rays in")[0].split()[-1] for line in r.stderr.splitlines() if "Mrays/s" in line]
    r1 = subprocess.run([CLI, scene_file, "-o", str(one), "--devices", "0"] + common, capture_output=True, text=True, timeout=120)
    assert r1.returncode == 0 and rays == [line.split(" rays in")[0].split()[-1] for line in r1.stderr.splitlines() if "Mrays/s" in line]
    r = subprocess.run([CLI, scene_file, "--devices", "0,x"], capture_output=True, text=True, timeout=60)
    assert r.returncode == 2 and "usage" in r.stderr
