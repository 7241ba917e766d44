"""Scene description files replay the calls of the reference's scene functions
(src/main.rs:1270-1450, 1223-1268): a file-built scene must be the same scene, bit for bit, as
the one assembled call by call."""
import ctypes as C
import os

import numpy as np
import pytest

import raytracer_amd as rt
from raytracer_amd import scenes

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def same_scene(a, b):
    for x, y in zip(a.export_flat(), b.export_flat()):
        assert np.array_equal(x, y)
    ka, ma, sa, oa = a.export_objects()
    kb, mb, sb, ob = b.export_objects()
    assert np.array_equal(ka, kb) and np.array_equal(sa, sb) and np.array_equal(oa, ob)
    assert bytes(ma) == bytes(mb)
    ia, ib = a.info(), b.info()
    assert ia.n_textures == ib.n_textures
    for t in range(ia.n_textures):
        assert np.array_equal(a.export_texture(t), b.export_texture(t))
    na, la = a.export_bvh()
    nb, lb = b.export_bvh()
    assert np.array_equal(na, nb) and np.array_equal(la, lb)


@pytest.mark.parametrize("name,size", [("spot", (1600, 1200, 10)), ("lens", (1600, 1200, 10)),
                                       ("sphere_floor", (1920, 1080, 64)), ("ogre", (3840, 2160, 256))])
def test_shipped_scene_files_equal_the_assembled_scenes(name, size):
    scene, p, its = rt.Scene.from_file(os.path.join(ROOT, "scenes", name + ".scene"))
    same_scene(scene, scenes.build(name))
    assert (p.width, p.height, p.spp) == size and its == 1
    assert (p.max_depth, p.fov_y_deg, p.dof_strength, p.seed, p.flags) == (200, 60.0, 0.0, 0, 0)


def test_defaults_are_the_reference_constants():
    from raytracer_amd import _ffi

    p = _ffi.rt_render_params()
    _ffi.lib.rt_render_params_default(C.byref(p))
    # src/main.rs:1041-1052, 978
    assert (p.width, p.height, p.spp, p.max_depth, p.fov_y_deg, p.dof_strength) == (1600, 1200, 10, 200, 60.0, 0.0)
    assert (p.first_sample, p.seed, p.flags, p.spp_total, p.tile_size) == (0, 0, 0, 0, 0)


def test_statements_transform_order_comments_and_render_line(tmp_path):
    assets = os.path.join(ROOT, "assets")
    f = tmp_path / "s.scene"
    f.write_text(
        f"""
        # transforms apply in the order written
        mesh {assets}/sphere.obj translate 1 2 3 rotate_x 90 rotate_z -30 opaque color 0.1 0.2 0.3 emission 4 5 6  # trailing
        sphere 0 1 -2 0.25 opaque
        sphere 0 1 -3 0.5 dielectric 1.33
        render width 64 height 48 spp 3 its 7 seed 12345678901234 max_depth 9 fov 45.5 dof 0.01
        """
    )
    scene, p, its = rt.Scene.from_file(f)
    m = rt.load_obj(os.path.join(assets, "sphere.obj"))
    m.translate((1.0, 2.0, 3.0))
    m.rotate_x(rt.host.deg(90.0))
    m.rotate_z(rt.host.deg(-30.0))
    objs = rt.Objects()
    objs.from_mesh(m, rt.Opaque(color=(0.1, 0.2, 0.3), metallic=0.0, how_metallic=0.0, diffuse=1.0, emission=(4.0, 5.0, 6.0)))
    objs.sphere((0.0, 1.0, -2.0), 0.25, rt.Opaque(color=(1.0, 1.0, 1.0), metallic=0.0, how_metallic=0.0, diffuse=1.0))
    objs.sphere((0.0, 1.0, -3.0), 0.5, rt.Dielectric(1.33))
    same_scene(scene, rt.Scene.from_objects(objs))
    assert (p.width, p.height, p.spp, its, p.seed, p.max_depth, p.fov_y_deg, p.dof_strength) == (64, 48, 3, 7, 12345678901234, 9, 45.5, 0.01)


def test_one_obj_file_used_twice_gives_independent_meshes(tmp_path):
    assets = os.path.join(ROOT, "assets")
    f = tmp_path / "two.scene"
    f.write_text(f"mesh {assets}/sphere.obj translate 5 0 0 opaque\nmesh {assets}/sphere.obj opaque\n")
    scene, _, _ = rt.Scene.from_file(f)
    tv, _, _, tobj, _ = scene.export_flat()
    n = len(tobj) // 2
    assert np.array_equal(tv[:n, :, 0], tv[n:, :, 0] + 5.0) and np.array_equal(tv[:n, :, 1:], tv[n:, :, 1:])


@pytest.mark.parametrize("text,msg", [
    ("bogus 1 2 3", "unknown statement"),
    ("sphere 0 0 0 opaque", "sphere needs"),
    ("sphere 0 0 0 1", "material expected"),
    ("sphere 0 0 0 1 shiny", "unknown material"),
    ("sphere 0 0 0 1 opaque color texture nope", "unknown texture"),
    ("sphere 0 0 0 1 opaque gloss 3", "unknown material property"),
    ("sphere 0 0 0 1 dielectric", "index of refraction"),
    ("sphere 0 0 0 1 dielectric 1.5 extra", "unexpected text"),
    ("mesh missing.obj opaque", "cannot open"),
    ("mesh {assets}/sphere.obj spin 3 opaque", "unknown mesh transform"),
    ("mesh {assets}/sphere.obj translate 1 2 opaque", "translate needs"),
    ("texture t missing.png", "cannot open"),
    ("texture t", "texture needs"),
    ("render width -3", "bad value"),
    ("render width 1.5", "bad value"),
    ("render colour 3", "unknown setting"),
])
def test_errors_name_file_and_line(tmp_path, text, msg):
    f = tmp_path / "bad.scene"
    f.write_text("# first line\n" + text.format(assets=os.path.join(ROOT, "assets")) + "\n")
    with pytest.raises(rt.RtError, match=msg) as e:
        rt.Scene.from_file(f)
    if "cannot open" not in msg:
        assert "bad.scene:2:" in str(e.value)


def test_missing_file():
    with pytest.raises(rt.RtError, match="cannot open"):
        rt.Objects.from_file("/nonexistent/x.scene")
