"""ctypes binding of oracle/librt_oracle.so — the CPU restatement used as the checker.
Test infrastructure: only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg use it."""
import ctypes as C
import os

import numpy as np

from raytracer_amd import _ffi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB_PATH = os.path.join(ROOT, "oracle", "librt_oracle.so")
ASSETS = os.path.join(ROOT, "assets")

HIT_BRUTE = 0
HIT_REFLIKE = 1

_P = C.c_void_p
lib = C.CDLL(LIB_PATH)
lib.orc_register_texture.argtypes = [C.c_char_p, _P, C.c_uint32, C.c_uint32]
lib.orc_scene_build.restype = _P
lib.orc_scene_build.argtypes = [C.c_char_p, C.c_char_p, C.c_char_p]
lib.orc_scene_from_desc.restype = _P
lib.orc_scene_from_desc.argtypes = [C.POINTER(_ffi.rt_scene_desc)]
lib.orc_scene_free.argtypes = [_P]
lib.orc_scene_counts.argtypes = [_P, C.POINTER(C.c_uint64), C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
lib.orc_scene_flat.argtypes = [_P, _P, _P, _P, _P, _P]
lib.orc_scene_objects.argtypes = [_P, _P, _P, _P, _P]
lib.orc_trace.argtypes = [_P, C.c_int, _P, _P, C.c_size_t, _P, C.c_int]
lib.orc_camera_rays.argtypes = [C.c_uint32, C.c_uint32, C.c_double, _P, _P]
lib.orc_camera_ray_sample.argtypes = [C.POINTER(_ffi.rt_render_params), C.c_uint32, C.c_uint32, C.c_uint32, _P, _P]
lib.orc_render.argtypes = [_P, C.POINTER(_ffi.rt_render_params), _P, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.c_int, C.c_int]
lib.orc_path_radiance.argtypes = [_P, C.POINTER(_ffi.rt_render_params), C.c_uint32, C.c_uint32, C.c_uint32, C.c_int, _P, C.POINTER(C.c_uint64)]
lib.orc_intersects_triangle.argtypes = [_P] * 6
lib.orc_intersects_sphere.argtypes = [_P, C.c_double, _P, _P, C.POINTER(C.c_double)]
lib.orc_interp_sky.argtypes = [_P, _P]
lib.orc_refract.argtypes = [_P, _P, C.c_double, _P]
lib.orc_reflectance.restype = C.c_double
lib.orc_reflectance.argtypes = [C.c_double, C.c_double]
lib.orc_rng_u01.restype = C.c_double
lib.orc_rng_u01.argtypes = [C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint32, C.c_uint32]
lib.orc_random_unit_sphere_surface.argtypes = [C.c_uint64, C.c_uint32, C.c_uint32, _P, C.POINTER(C.c_uint32)]
lib.orc_normal_map.argtypes = [_P, _P, _P]
lib.orc_load_obj_counts.argtypes = [C.c_char_p, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
lib.orc_bvh_trace.argtypes = [_P, _P, C.c_uint32, C.c_int32, _P, _P, _P, C.c_size_t, _P, _P, C.c_int]


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


def _vec(x):
    return np.ascontiguousarray(x, dtype=np.float64)


_registered = set()


def register_texture(name, rgba):
    a = np.ascontiguousarray(rgba, dtype=np.uint8)
    lib.orc_register_texture(name.encode(), _ptr(a), a.shape[1], a.shape[0])
    _registered.add(name)


def decode_texture(path):
    """[h][w][4] uint8 of a PNG, decoded by Pillow and not by the product's own decoder: the oracle's
    texels must not depend on the code under test. Pillow, like the reference's image 0.23, ignores
    the gamma and colour-profile chunks some of the files carry."""
    from PIL import Image

    Image.MAX_IMAGE_PIXELS = None  # the 20000 x 20000 texture is no decompression bomb
    with Image.open(path) as im:
        return np.ascontiguousarray(np.asarray(im.convert("RGBA"), dtype=np.uint8))


def register_default_textures():
    for name in ("spot_texture.png", "ogre/diffuse.png", "ogre/normalmap.png"):
        if name not in _registered:
            register_texture(name, decode_texture(os.path.join(ASSETS, name)))


class OracleScene:
    def __init__(self, handle):
        if not handle:
            raise RuntimeError("oracle scene construction failed")
        self._h = C.c_void_p(handle)

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            lib.orc_scene_free(h)

    @staticmethod
    def build(name, assets=ASSETS, spot_texture="spot_texture.png"):
        register_default_textures()
        if name == "spot_20k":  # the default scene with the 20000x20000 texture of src/main.rs:1273
            name, spot_texture = "spot", "spot_texture_20k.png"
            if spot_texture not in _registered:
                register_texture(spot_texture, decode_texture(os.path.join(assets, spot_texture)))
        return OracleScene(lib.orc_scene_build(name.encode(), assets.encode(), spot_texture.encode()))

    @staticmethod
    def from_desc_arrays(object_kind, materials, tri_vertices, tri_normals, tri_uvs, tri_obj, tri_idx, spheres, sphere_obj, textures=()):
        keep = []

        def arr(a, dt):
            a = np.ascontiguousarray(a, dtype=dt)
            keep.append(a)
            return a

        d = _ffi.rt_scene_desc()
        ok = arr(object_kind, np.int32)
        mats = (_ffi.rt_material * len(materials))(*materials)
        d.n_objects = len(ok)
        d.object_kind = _ptr(ok).value
        d.materials = C.cast(mats, C.c_void_p).value
        tv = arr(tri_vertices, np.float64).reshape(-1, 9)
        d.n_triangles = tv.shape[0]
        if tv.shape[0]:
            d.tri_vertices = _ptr(tv).value
            d.tri_normals = _ptr(arr(tri_normals, np.float64)).value
            d.tri_uvs = _ptr(arr(tri_uvs, np.float64)).value
            d.tri_obj = _ptr(arr(tri_obj, np.int32)).value
            d.tri_idx = _ptr(arr(tri_idx, np.int32)).value
        sp = arr(spheres, np.float64).reshape(-1, 4)
        d.n_spheres = sp.shape[0]
        if sp.shape[0]:
            d.spheres = _ptr(sp).value
            d.sphere_obj = _ptr(arr(sphere_obj, np.int32)).value
        tex = (_ffi.rt_texture * max(1, len(textures)))()
        for i, t in enumerate(textures):
            a = arr(t, np.uint8)
            tex[i].rgba = _ptr(a).value
            tex[i].width = a.shape[1]
            tex[i].height = a.shape[0]
        d.n_textures = len(textures)
        d.textures = C.cast(tex, C.c_void_p).value if textures else None
        return OracleScene(lib.orc_scene_from_desc(C.byref(d)))

    def counts(self):
        nt, ns, no = C.c_uint64(), C.c_uint32(), C.c_uint32()
        lib.orc_scene_counts(self._h, C.byref(nt), C.byref(ns), C.byref(no))
        return nt.value, ns.value, no.value

    def flat(self):
        n = self.counts()[0]
        tv = np.empty((n, 3, 3)); tn = np.empty((n, 3, 3)); tuv = np.empty((n, 3, 2))
        tobj = np.empty(n, np.int32); tidx = np.empty(n, np.int32)
        lib.orc_scene_flat(self._h, _ptr(tv), _ptr(tn), _ptr(tuv), _ptr(tobj), _ptr(tidx))
        return tv, tn, tuv, tobj, tidx

    def objects(self):
        _, ns, no = self.counts()
        kinds = np.empty(no, np.int32)
        mats = (_ffi.rt_material * no)()
        sph = np.empty((ns, 4)); sobj = np.empty(ns, np.int32)
        lib.orc_scene_objects(self._h, _ptr(kinds), C.cast(mats, C.c_void_p), _ptr(sph), _ptr(sobj))
        return kinds, mats, sph, sobj

    def hit(self, origins, dirs, mode=HIT_BRUTE, threads=8):
        from raytracer_amd.host import HIT_DTYPE

        o = _vec(origins).reshape(-1, 3); d = _vec(dirs).reshape(-1, 3)
        out = np.zeros(o.shape[0], HIT_DTYPE)
        lib.orc_trace(self._h, mode, _ptr(o), _ptr(d), o.shape[0], _ptr(out), threads)
        return out

    def bvh_hit(self, nodes, leaf_order, root_ref, origins, dirs, threads=8):
        """Traversal of the library's exported tree, restated on the CPU. Returns hits and
        (node visits, triangle tests)."""
        from raytracer_amd.host import HIT_DTYPE

        o = _vec(origins).reshape(-1, 3); d = _vec(dirs).reshape(-1, 3)
        out = np.zeros(o.shape[0], HIT_DTYPE)
        counts = np.zeros(2, np.uint64)
        nodes = np.ascontiguousarray(nodes, dtype=np.uint32)
        leaf_order = np.ascontiguousarray(leaf_order, dtype=np.int32)
        lib.orc_bvh_trace(self._h, _ptr(nodes), nodes.shape[0], int(root_ref), _ptr(leaf_order), _ptr(o), _ptr(d), o.shape[0], _ptr(out), _ptr(counts), threads)
        return out, (int(counts[0]), int(counts[1]))

    def render(self, canvas, params, threads=8, mode=HIT_BRUTE):
        rays, cands = C.c_uint64(), C.c_uint64()
        lib.orc_render(self._h, C.byref(params), _ptr(canvas), C.byref(rays), C.byref(cands), threads, mode)
        return rays.value, cands.value

    def path_radiance(self, params, u, v, sample, mode=HIT_BRUTE):
        rgb = np.zeros(3)
        rays = C.c_uint64()
        lib.orc_path_radiance(self._h, C.byref(params), u, v, sample, mode, _ptr(rgb), C.byref(rays))
        return rgb, rays.value


def camera_rays(w, h, fov_y_deg=60.0):
    o = np.empty((h * w, 3)); d = np.empty((h * w, 3))
    lib.orc_camera_rays(w, h, fov_y_deg, _ptr(o), _ptr(d))
    return o, d


def camera_ray_sample(params, u, v, sample):
    o = np.empty(3); d = np.empty(3)
    lib.orc_camera_ray_sample(C.byref(params), u, v, sample, _ptr(o), _ptr(d))
    return o, d


def intersects_triangle(o, d, a, b, c):
    tuv = np.zeros(3)
    hit = lib.orc_intersects_triangle(_ptr(_vec(o)), _ptr(_vec(d)), _ptr(_vec(a)), _ptr(_vec(b)), _ptr(_vec(c)), _ptr(tuv))
    return tuple(tuv) if hit else None


def intersects_sphere(center, radius, o, d):
    t = C.c_double()
    hit = lib.orc_intersects_sphere(_ptr(_vec(center)), radius, _ptr(_vec(o)), _ptr(_vec(d)), C.byref(t))
    return t.value if hit else None


def interp_sky(d):
    rgb = np.zeros(3)
    lib.orc_interp_sky(_ptr(_vec(d)), _ptr(rgb))
    return rgb


def refract(uv, n, eta):
    out = np.zeros(3)
    lib.orc_refract(_ptr(_vec(uv)), _ptr(_vec(n)), eta, _ptr(out))
    return out


def reflectance(c, ri):
    return lib.orc_reflectance(c, ri)


def rng_u01(seed, pixel, sample, stage, index):
    return lib.orc_rng_u01(seed, pixel, sample, stage, index)


def random_unit_sphere_surface(key, stage, index):
    out = np.zeros(3)
    after = C.c_uint32()
    lib.orc_random_unit_sphere_surface(key, stage, index, _ptr(out), C.byref(after))
    return out, after.value


def normal_map(rgb, world_normal):
    out = np.zeros(3)
    px = np.asarray(rgb, dtype=np.uint8)
    lib.orc_normal_map(_ptr(px), _ptr(_vec(world_normal)), _ptr(out))
    return out


def load_obj_counts(path):
    nv, nt, nuv = C.c_uint64(), C.c_uint64(), C.c_uint64()
    rc = lib.orc_load_obj_counts(path.encode(), C.byref(nv), C.byref(nt), C.byref(nuv))
    if rc != 0:
        raise IOError(path)
    return nv.value, nt.value, nuv.value
