"""The reference's scenes, assembled through the C ABI exactly as src/main.rs assembles them.

sphere_scene   src/main.rs:1270-1450   the default scene ("Spot": floor, cow, six spheres); "spot"
               takes the cow's texture from the 1024x1024 file its material library names,
               "spot_20k" from the 20000x20000 rendering of the same picture that
               src/main.rs:1273 opens (1.6 GB of texels once decoded)
lens_scene     src/main.rs:1223-1268
sphere_floor_scene / ogre_scene: the default scene's floor plus the objects that are present
but commented out at src/main.rs:1353-1364 and 1434-1445 (benchmark configurations)
spot_grid_scene: floor plus nx x nz duplicated cows (the reference has no instancing)
"""
import os

from .host import Dielectric, Objects, Opaque, Scene, deg, load_obj

ASSETS = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "assets")


def load_texture(path):
    """Decodes a PNG to [h][w][4] uint8 with the library's own decoder, which ignores
    gamma/colour-profile chunks as the reference's does (src/main.rs:1226-1232 via image 0.23)."""
    from .host import load_png

    return load_png(path)


def _floor(objects, assets):
    # src/main.rs:1284-1296
    floor = load_obj(os.path.join(assets, "triangle_floor.obj"))
    objects.from_mesh(floor, Opaque(color=(0.7, 0.7, 0.7), metallic=0.1, how_metallic=0.0, diffuse=1.0))


def sphere_scene(assets=ASSETS, spot_texture="spot_texture.png", tree="host"):
    spot_poly = load_obj(os.path.join(assets, "spot_triangulated.obj"))
    objects = Objects()
    spot_text = objects.add_texture(load_texture(os.path.join(assets, spot_texture)))
    _floor(objects, assets)

    mirror = spot_poly.clone()  # src/main.rs:1339-1351
    mirror.rotate_y(deg(220.0))
    mirror.translate((1.0, 0.0, -2.5))
    objects.from_mesh(mirror, Opaque(color=spot_text, metallic=0.0, how_metallic=0.0, diffuse=1.0))

    objects.sphere((-0.3, -0.2, -2.8), 0.5, Opaque((0.7, 0.7, 0.75), metallic=0.0, how_metallic=1.0, diffuse=0.9))
    objects.sphere((-1.0, -0.5, -2.8), 0.2, Opaque((1.0, 1.0, 1.0), metallic=0.0, how_metallic=1.0, diffuse=0.9))
    objects.sphere((-1.5, -0.4, -2.3), 0.3, Opaque((0.0, 0.0, 0.6), metallic=0.0, how_metallic=1.0, diffuse=0.9))
    objects.sphere((0.3, -0.4, -1.4), 0.3, Dielectric(1.5))
    objects.sphere((-0.15, -0.6, -2.0), 0.1, Opaque((0.7, 0.0, 0.0), metallic=0.0, how_metallic=0.0, diffuse=1.0))
    objects.sphere((0.0, 0.3, -1.9), 0.1, Dielectric(1.5))
    return Scene.from_objects(objects, tree)


def lens_scene(assets=ASSETS, spot_texture="spot_texture.png", tree="host"):
    spot_poly = load_obj(os.path.join(assets, "spot_triangulated.obj"))
    objects = Objects()
    spot_text = objects.add_texture(load_texture(os.path.join(assets, spot_texture)))
    _floor(objects, assets)
    mirror = spot_poly.clone()
    mirror.rotate_y(deg(220.0))
    mirror.rotate_z(deg(180.0))
    mirror.translate((0.0, 0.9, -2.2))
    objects.from_mesh(mirror, Opaque(color=spot_text, metallic=0.0, how_metallic=0.0, diffuse=1.0))
    objects.sphere((0.0, -0.3, -1.9), 0.3, Dielectric(1.5))
    return Scene.from_objects(objects, tree)


def sphere_floor_scene(assets=ASSETS, tree="host"):
    objects = Objects()
    _floor(objects, assets)
    sphere = load_obj(os.path.join(assets, "sphere.obj"))  # src/main.rs:1353-1364
    sphere.translate((0.0, 2.0, -7.0))
    objects.from_mesh(sphere, Opaque((1.0, 0.3, 0.0), metallic=0.0, how_metallic=0.0, diffuse=0.9))
    return Scene.from_objects(objects, tree)


def ogre_scene(assets=ASSETS, tree="host"):
    objects = Objects()
    ogre_text = objects.add_texture(load_texture(os.path.join(assets, "ogre", "diffuse.png")))
    ogre_normal = objects.add_texture(load_texture(os.path.join(assets, "ogre", "normalmap.png")))
    _floor(objects, assets)
    ogre = load_obj(os.path.join(assets, "ogre", "bs_rest.obj"))  # src/main.rs:1434-1445
    ogre.rotate_y(deg(20.0))
    ogre.translate((0.0, 0.0, -2.0))
    objects.from_mesh(ogre, Opaque(color=ogre_text, metallic=0.0, how_metallic=0.0, diffuse=0.9, normal=ogre_normal))
    return Scene.from_objects(objects, tree)


def spot_grid_scene(nx, nz, assets=ASSETS, spot_texture="spot_texture.png", tree="host"):
    objects = Objects()
    spot_text = objects.add_texture(load_texture(os.path.join(assets, spot_texture)))
    _floor(objects, assets)
    turned = load_obj(os.path.join(assets, "spot_triangulated.obj"))
    turned.rotate_y(deg(220.0))
    for iz in range(nz):
        for ix in range(nx):
            m = turned.clone()
            m.translate(((ix - 0.5 * (nx - 1)) * 1.2, 0.0, -2.5 - iz * 1.2))
            objects.from_mesh(m, Opaque(color=spot_text, metallic=0.0, how_metallic=0.0, diffuse=1.0))
    return Scene.from_objects(objects, tree)


SCENES = {
    "spot": sphere_scene,
    "spot_20k": lambda assets=ASSETS, tree="host": sphere_scene(assets, "spot_texture_20k.png", tree),
    "lens": lens_scene,
    "sphere_floor": sphere_floor_scene,
    "ogre": ogre_scene,
}


def build(name, assets=ASSETS, tree="host"):
    """tree="none": no tree is built on the host (the caller builds it on the GPU with
    Scene.rebuild_on_device before the first upload)."""
    if name.startswith("spot_grid:"):
        nx, nz = name.split(":", 1)[1].split("x")
        return spot_grid_scene(int(nx), int(nz), assets, tree=tree)
    return SCENES[name](assets, tree=tree)
