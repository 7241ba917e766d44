"""examples/render_loop.c: the reference's main loop (src/main.rs:1077-1112) written in C99 against
include/rt_amd.h — the header must be plain C, every call must link, and without a GPU the
program must stop with the library's own message instead of producing a picture."""
import os
import shutil
import subprocess

import numpy as np
import pytest

import raytracer_amd as rt

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIBDIR = os.path.join(ROOT, "raytracer_amd")


@pytest.fixture(scope="module")
def program(tmp_path_factory):
    if shutil.which("gcc") is None:
        pytest.skip("no gcc")
    exe = str(tmp_path_factory.mktemp("c_example") / "render_loop")
    r = subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-I", os.path.join(ROOT, "include"),
                        os.path.join(ROOT, "examples", "render_loop.c"), "-L", LIBDIR, "-lrt_amd", f"-Wl,-rpath,{LIBDIR}", "-o", exe],
                       capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    return exe


def small_scene(tmp_path):
    text = open(os.path.join(ROOT, "scenes", "spot.scene")).read().replace("../assets", os.path.join(ROOT, "assets"))
    lines = [l for l in text.splitlines() if not l.startswith("render")] + ["render width 96 height 72 spp 3 its 2 seed 4"]
    path = tmp_path / "small.scene"
    path.write_text("\n".join(lines) + "\n")
    return str(path)


def test_c_program_builds_and_refuses_to_run_without_a_gpu(program, tmp_path):
    if rt.device_count() > 0:
        pytest.skip("a GPU is present")
    out = tmp_path / "o.png"
    r = subprocess.run([program, small_scene(tmp_path), str(out)], capture_output=True, text=True, timeout=120)
    assert r.returncode == 1 and "no HIP device" in r.stderr and not out.exists()
    r = subprocess.run([program], capture_output=True, text=True, timeout=60)
    assert r.returncode == 2 and "usage" in r.stderr


@pytest.mark.gpu
def test_c_program_renders_what_the_binding_renders(program, tmp_path):
    scene_file = small_scene(tmp_path)
    out = tmp_path / "o.png"
    r = subprocess.run([program, scene_file, str(out)], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    scene, p, its = rt.Scene.from_file(scene_file)
    assert (p.width, p.height, p.spp, its) == (96, 72, 3, 2)
    c = np.zeros((72, 96, 3))
    rays = 0
    for it in range(its):
        p.first_sample = it * p.spp
        rays += scene.render(c, p)
    assert np.array_equal(rt.load_png(out), rt.quantize_rgba8(c, its))
    assert f"{rays} rays" in r.stdout
