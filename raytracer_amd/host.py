"""Host-side mirror of the reference's scene interface, over the C ABI.

Names follow src/main.rs: PolygonMesh (566-574) with rotate_x/y/z and translate (627-664),
load_obj (1130-1221), Material Opaque / Dielectric (414-425), Object::from_mesh / Object::sphere
(454-469), Scene::from_objects (750-774), Scene::hit (776-814) and render (977-1039).
All work is done by librt_amd.so; this file only marshals arguments.
"""
import ctypes as C
import math

import numpy as np

from . import _ffi
from ._ffi import check, lib

# constants of src/main.rs:1041-1052
MAX_DEPTH = 200
NUM_SAMPLES_PER_PIXEL = 10
DOF_STRENGTH = 0.0
WIDTH = 1600
HEIGHT = 1200
FOV_Y_DEG = 60.0


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


class PolygonMesh:
    """PolygonMesh, src/main.rs:566-574."""

    def __init__(self, handle):
        self._h = C.c_void_p(handle)

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            lib.rt_mesh_destroy(h)

    def clone(self):
        out = C.c_void_p()
        check(lib.rt_mesh_clone(self._h, C.byref(out)), "rt_mesh_clone")
        return PolygonMesh(out.value)

    def counts(self):
        nv, nt, nuv = C.c_uint64(), C.c_uint64(), C.c_uint64()
        check(lib.rt_mesh_counts(self._h, C.byref(nv), C.byref(nt), C.byref(nuv)), "rt_mesh_counts")
        return nv.value, nt.value, nuv.value

    def rotate_x(self, angle):
        check(lib.rt_mesh_rotate_x(self._h, float(angle)), "rt_mesh_rotate_x")

    def rotate_y(self, angle):
        check(lib.rt_mesh_rotate_y(self._h, float(angle)), "rt_mesh_rotate_y")

    def rotate_z(self, angle):
        check(lib.rt_mesh_rotate_z(self._h, float(angle)), "rt_mesh_rotate_z")

    def translate(self, t):
        v = (C.c_double * 3)(*[float(x) for x in t])
        check(lib.rt_mesh_translate(self._h, v), "rt_mesh_translate")


def load_obj(filename):
    """load_obj, src/main.rs:1130-1221."""
    out = C.c_void_p()
    check(lib.rt_mesh_load_obj(str(filename).encode(), C.byref(out)), "rt_mesh_load_obj")
    return PolygonMesh(out.value)


def Opaque(color, metallic, diffuse, how_metallic, emission=(0.0, 0.0, 0.0), normal=None):
    """Material::Opaque, src/main.rs:417-424. `color` is an (r, g, b) SolidColor or a texture id
    returned by Objects.add_texture; `normal` is None (SimpleNormal) or a texture id (NormalMap)."""
    m = _ffi.rt_material()
    m.kind = _ffi.RT_MAT_OPAQUE
    if isinstance(color, int):
        m.color_tex = color
    else:
        m.color_tex = -1
        m.color = (C.c_double * 3)(*[float(x) for x in color])
    m.normal_tex = -1 if normal is None else int(normal)
    m.metallic = float(metallic)
    m.diffuse = float(diffuse)
    m.how_metallic = float(how_metallic)
    m.emission = (C.c_double * 3)(*[float(x) for x in emission])
    m.ior = 1.0
    return m


def Dielectric(index_of_refraction):
    """Material::Dielectric, src/main.rs:416."""
    m = _ffi.rt_material()
    m.kind = _ffi.RT_MAT_DIELECTRIC
    m.color_tex = -1
    m.normal_tex = -1
    m.ior = float(index_of_refraction)
    return m


class Objects:
    """The Vec<Object> a scene function fills (src/main.rs:1282), plus its textures."""

    def __init__(self):
        out = C.c_void_p()
        check(lib.rt_objects_create(C.byref(out)), "rt_objects_create")
        self._h = out

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            lib.rt_objects_destroy(h)

    @staticmethod
    def from_file(path):
        """Replays a scene description file; returns (objects, render params, its)."""
        out, p, its = C.c_void_p(), _ffi.rt_render_params(), C.c_uint32()
        check(lib.rt_scene_file_load(str(path).encode(), C.byref(out), C.byref(p), C.byref(its)), "rt_scene_file_load")
        objs = Objects.__new__(Objects)
        objs._h = out
        return objs, p, its.value

    def add_texture(self, rgba):
        a = np.ascontiguousarray(rgba, dtype=np.uint8)
        assert a.ndim == 3 and a.shape[2] == 4, "texture must be [h][w][4] uint8"
        return check(lib.rt_objects_add_texture(self._h, _ptr(a), a.shape[1], a.shape[0]), "rt_objects_add_texture")

    def from_mesh(self, mesh, material):
        """Object::from_mesh + push; returns objects.len() - 1."""
        return check(lib.rt_objects_push_mesh(self._h, mesh._h, C.byref(material)), "rt_objects_push_mesh")

    def sphere(self, center, radius, material):
        """Object::sphere + push; returns objects.len() - 1."""
        c = (C.c_double * 3)(*[float(x) for x in center])
        return check(lib.rt_objects_push_sphere(self._h, c, float(radius), C.byref(material)), "rt_objects_push_sphere")


def render_params(width=WIDTH, height=HEIGHT, spp=NUM_SAMPLES_PER_PIXEL, first_sample=0, seed=0, max_depth=MAX_DEPTH,
                  flags=0, fov_y_deg=FOV_Y_DEG, dof_strength=DOF_STRENGTH, spp_total=0, tile_size=0, tile_rank=0,
                  tile_ranks=1):
    p = _ffi.rt_render_params()
    p.width, p.height, p.spp, p.first_sample = int(width), int(height), int(spp), int(first_sample)
    p.seed, p.max_depth, p.flags = int(seed), int(max_depth), int(flags)
    p.fov_y_deg, p.dof_strength = float(fov_y_deg), float(dof_strength)
    p.spp_total, p.tile_size, p.tile_rank, p.tile_ranks = int(spp_total), int(tile_size), int(tile_rank), int(tile_ranks)
    return p


HIT_DTYPE = np.dtype([("t", "<f8"), ("u", "<f8"), ("v", "<f8"), ("obj", "<i4"), ("tri", "<i4")])


class Scene:
    """Scene, src/main.rs:747."""

    def __init__(self, handle):
        self._h = C.c_void_p(handle)

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            lib.rt_scene_destroy(h)

    @staticmethod
    def from_objects(objs, tree="host"):
        """Scene::from_objects, src/main.rs:750-774. tree="none" skips the host's tree build for
        callers that build the tree on the GPU next (rebuild_on_device)."""
        out = C.c_void_p()
        t = {"host": _ffi.RT_TREE_HOST, "none": _ffi.RT_TREE_NONE}[tree]
        check(lib.rt_scene_from_objects_with(objs._h, t, C.byref(out)), "rt_scene_from_objects_with")
        return Scene(out.value)

    @staticmethod
    def from_file(path):
        """Scene of a description file; returns (scene, render params, its)."""
        objs, p, its = Objects.from_file(path)
        return Scene.from_objects(objs), p, its

    @staticmethod
    def from_desc(object_kind, materials, tri_vertices, tri_normals, tri_uvs, tri_obj, tri_idx, spheres, sphere_obj,
                  textures=()):
        """rt_scene_create from flat numpy arrays (layouts of rt_scene_desc)."""
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
        out = C.c_void_p()
        check(lib.rt_scene_create(C.byref(d), C.byref(out)), "rt_scene_create")
        return Scene(out.value)

    # ---- inspection -------------------------------------------------------------------
    def info(self):
        i = _ffi.rt_scene_info()
        check(lib.rt_scene_get_info(self._h, C.byref(i)), "rt_scene_get_info")
        return i

    def export_flat(self):
        n = self.info().n_triangles
        tv = np.empty((n, 3, 3)); tn = np.empty((n, 3, 3)); tuv = np.empty((n, 3, 2))
        tobj = np.empty(n, np.int32); tidx = np.empty(n, np.int32)
        check(lib.rt_scene_export_flat(self._h, _ptr(tv), _ptr(tn), _ptr(tuv), _ptr(tobj), _ptr(tidx)), "rt_scene_export_flat")
        return tv, tn, tuv, tobj, tidx

    def export_objects(self):
        """(object kinds, materials as a ctypes array, spheres [n][4], sphere object indices)."""
        i = self.info()
        kind = np.zeros(i.n_objects, np.int32)
        mats = (_ffi.rt_material * i.n_objects)()
        sph = np.zeros((i.n_spheres, 4)); sobj = np.zeros(i.n_spheres, np.int32)
        check(lib.rt_scene_export_objects(self._h, _ptr(kind), C.cast(mats, C.c_void_p), _ptr(sph), _ptr(sobj)),
              "rt_scene_export_objects")
        return kind, mats, sph, sobj

    def export_texture(self, tex_id):
        w, h = C.c_uint32(), C.c_uint32()
        check(lib.rt_scene_export_texture(self._h, tex_id, C.byref(w), C.byref(h), None), "rt_scene_export_texture")
        px = np.zeros((h.value, w.value, 4), np.uint8)
        check(lib.rt_scene_export_texture(self._h, tex_id, None, None, _ptr(px)), "rt_scene_export_texture")
        return px

    def export_bvh(self):
        i = self.info()
        nodes = np.zeros((i.bvh_nodes, 16), np.uint32)
        order = np.zeros(i.n_triangles + i.bvh_spheres, np.int32)
        check(lib.rt_scene_export_bvh(self._h, _ptr(nodes), _ptr(order)), "rt_scene_export_bvh")
        return nodes, order

    def export_grid_bvh(self):
        """The half-size copy of the tree the launches over the ray queue walk: (nodes as
        [bvh_nodes][8] uint32, lo[3], step[3]), or None when the scene does not qualify for it."""
        i = self.info()
        nodes = np.zeros((i.bvh_nodes, 8), np.uint32)
        lo_step = np.zeros(6, np.float32)
        usable = C.c_int32()
        check(lib.rt_scene_export_grid_bvh(self._h, _ptr(nodes), _ptr(lo_step), C.byref(usable)), "rt_scene_export_grid_bvh")
        if not usable.value:
            return None
        return nodes, lo_step[:3].copy(), lo_step[3:].copy()

    # ---- device -----------------------------------------------------------------------
    def rebuild_on_device(self, device=0, method=None):
        """Replaces the host-built tree by one built on the GPU (for scenes that change):
        method "lbvh" (Morton hierarchy), "ploc" (bottom-up clustering) or None for the default."""
        if method is None:
            check(lib.rt_scene_rebuild_on_device(self._h, int(device)), "rt_scene_rebuild_on_device")
        else:
            m = {"lbvh": _ffi.RT_BUILD_LBVH, "ploc": _ffi.RT_BUILD_PLOC}.get(method, method)
            check(lib.rt_scene_rebuild_on_device_with(self._h, int(device), int(m)), "rt_scene_rebuild_on_device_with")

    def upload(self, device=0):
        check(lib.rt_scene_upload(self._h, int(device)), "rt_scene_upload")

    def hit(self, origins, dirs):
        """Batch Scene::hit, src/main.rs:776-814. Returns a structured array (t,u,v,obj,tri)."""
        o = np.ascontiguousarray(origins, dtype=np.float64).reshape(-1, 3)
        d = np.ascontiguousarray(dirs, dtype=np.float64).reshape(-1, 3)
        assert o.shape == d.shape
        out = np.zeros(o.shape[0], HIT_DTYPE)
        check(lib.rt_trace(self._h, _ptr(o), _ptr(d), o.shape[0], _ptr(out)), "rt_trace")
        return out

    def primary_hits(self, params):
        n = params.width * params.height
        obj = np.empty(n, np.int32); tri = np.empty(n, np.int32)
        t = np.empty(n); u = np.empty(n); v = np.empty(n)
        check(lib.rt_primary_hits(self._h, C.byref(params), _ptr(obj), _ptr(tri), _ptr(t), _ptr(u), _ptr(v)), "rt_primary_hits")
        shape = (params.height, params.width)
        return obj.reshape(shape), tri.reshape(shape), t.reshape(shape), u.reshape(shape), v.reshape(shape)

    def render(self, canvas, params):
        """render::<W,H>(&mut canvas, &scene), src/main.rs:977: adds the mean of params.spp samples
        into canvas ([H][W][3] float64, host). Returns the number of nearest-hit queries."""
        assert canvas.dtype == np.float64 and canvas.flags.c_contiguous
        assert canvas.shape == (params.height, params.width, 3)
        rays = C.c_uint64()
        check(lib.rt_render_accumulate(self._h, C.byref(params), _ptr(canvas), C.byref(rays)), "rt_render_accumulate")
        return rays.value

    def render_device(self, device_canvas_ptr, params, stream=None):
        check(lib.rt_render_device(self._h, C.byref(params), C.c_void_p(device_canvas_ptr), C.c_void_p(stream or 0)), "rt_render_device")

    def render_device_noalloc(self, device_canvas_ptr, params, stream=None):
        """render_device out of the library's static work memory: allocates nothing; the scene must
        already be uploaded and the image is bounded (see rt_render_device_noalloc)."""
        check(lib.rt_render_device_noalloc(self._h, C.byref(params), C.c_void_p(device_canvas_ptr), C.c_void_p(stream or 0)), "rt_render_device_noalloc")

    def render_sum_f32_device(self, device_sum_ptr, params, stream=None):
        check(lib.rt_render_sum_f32_device(self._h, C.byref(params), C.c_void_p(device_sum_ptr), C.c_void_p(stream or 0)), "rt_render_sum_f32_device")

    def wait(self):
        """Waits for everything the scene has queued (ends a run of pipelined calls)."""
        check(lib.rt_scene_wait(self._h), "rt_scene_wait")

    def stream_wait(self, stream):
        """Makes `stream` (a hipStream_t as an integer) wait on the device for everything the scene
        has queued so far; returns at once."""
        check(lib.rt_scene_stream_wait(self._h, C.c_void_p(stream or 0)), "rt_scene_stream_wait")

    def last_stats(self):
        s = _ffi.rt_stats()
        check(lib.rt_scene_last_stats(self._h, C.byref(s)), "rt_scene_last_stats")
        return s

    def last_launches(self):
        n = C.c_uint32()
        check(lib.rt_scene_last_launches(self._h, None, 0, C.byref(n)), "rt_scene_last_launches")
        out = (_ffi.rt_launch_info * max(1, n.value))()
        check(lib.rt_scene_last_launches(self._h, out, n.value, C.byref(n)), "rt_scene_last_launches")
        return [out[i] for i in range(n.value)]

    def last_timing(self):
        t = _ffi.rt_timing()
        check(lib.rt_scene_last_timing(self._h, C.byref(t)), "rt_scene_last_timing")
        return t


class Group:
    """Several GPUs of one node driven from this one process, the way the reference's single
    process spreads render over its cores (src/main.rs:988, 994): one member per listed device,
    each with its own copy of the scene. The first device combines the frame."""

    def __init__(self, scene, devices):
        devs = (C.c_int * len(devices))(*[int(d) for d in devices])
        h = C.c_void_p()
        check(lib.rt_group_create(scene._h, devs, len(devices), C.byref(h)), "rt_group_create")
        self._h = h

    def __del__(self):
        if getattr(self, "_h", None):
            lib.rt_group_destroy(self._h)
            self._h = None

    def __len__(self):
        n = C.c_uint32()
        check(lib.rt_group_size(self._h, C.byref(n)), "rt_group_size")
        return n.value

    def render(self, canvas, params, split=_ffi.RT_SPLIT_SAMPLES):
        """render::<W,H>(&mut canvas, &scene) over the group: adds the mean of params.spp samples
        into the host canvas. Returns the number of nearest-hit queries of all members."""
        assert canvas.dtype == np.float64 and canvas.flags.c_contiguous
        assert canvas.shape == (params.height, params.width, 3)
        rays = C.c_uint64()
        check(lib.rt_group_render_accumulate(self._h, C.byref(params), int(split), _ptr(canvas), C.byref(rays)), "rt_group_render_accumulate")
        return rays.value

    def render_device(self, device_canvas_ptr, params, split=_ffi.RT_SPLIT_SAMPLES, stream=None):
        """Queues one frame into a canvas in the first device's memory and returns; `stream` orders
        the canvas (the frame is added behind what the stream held, later work on it sees the frame)."""
        check(lib.rt_group_render_device(self._h, C.byref(params), int(split), C.c_void_p(device_canvas_ptr), C.c_void_p(stream or 0)), "rt_group_render_device")

    def wait(self):
        """Waits for every frame queued so far; returns the nearest-hit queries of all members in the
        frames since the previous wait."""
        rays = C.c_uint64()
        check(lib.rt_group_wait(self._h, C.byref(rays)), "rt_group_wait")
        return rays.value

    def member_info(self, member):
        """(device, nearest-hit queries, milliseconds) of one member: queries in the frames the latest
        wait covered, time of its share of the latest of them."""
        dev, rays, ms = C.c_int(), C.c_uint64(), C.c_double()
        check(lib.rt_group_member_info(self._h, int(member), C.byref(dev), C.byref(rays), C.byref(ms)), "rt_group_member_info")
        return dev.value, rays.value, ms.value


def load_png(path):
    """image::io::Reader::open(path).decode() as [h][w][4] uint8 (src/main.rs:1226-1232)."""
    img = C.c_void_p()
    check(lib.rt_image_load_png(str(path).encode(), C.byref(img)), "rt_image_load_png")
    try:
        w, h, px = C.c_uint32(), C.c_uint32(), C.c_void_p()
        check(lib.rt_image_info(img, C.byref(w), C.byref(h), C.byref(px)), "rt_image_info")
        buf = (C.c_uint8 * (w.value * h.value * 4)).from_address(px.value)
        return np.frombuffer(buf, dtype=np.uint8).reshape(h.value, w.value, 4).copy()
    finally:
        lib.rt_image_destroy(img)


def write_png(path, rgba):
    a = np.ascontiguousarray(rgba, dtype=np.uint8)
    assert a.ndim == 3 and a.shape[2] == 4
    check(lib.rt_image_write_png(str(path).encode(), _ptr(a), a.shape[1], a.shape[0]), "rt_image_write_png")


def quantize_rgba8(canvas, its):
    """The display loop's canvas update, src/main.rs:1106-1112 with 29-37."""
    h, w, _ = canvas.shape
    out = np.empty((h, w, 4), np.uint8)
    a = np.ascontiguousarray(canvas, dtype=np.float64)
    check(lib.rt_quantize_rgba8(_ptr(a), w, h, int(its), _ptr(out)), "rt_quantize_rgba8")
    return out


def selftest_division(n, seed=0):
    """Runs n operand triples through the kernels' shared-reciprocal division and through plain IEEE
    division on device 0; returns how many differ in any bit (expected 0)."""
    bad = C.c_uint64()
    check(lib.rt_selftest_division(int(n), int(seed), C.byref(bad)), "rt_selftest_division")
    return bad.value


def device_count():
    return lib.rt_device_count()


def deg(x):
    return x * math.pi / 180.0
