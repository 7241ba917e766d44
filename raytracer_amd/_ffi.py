"""ctypes binding of librt_amd.so (the C ABI declared in include/rt_amd.h).

This module only loads the shared library and declares signatures; there is no Python or CPU
implementation of anything behind it. If the library is missing, importing fails loudly.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("RT_AMD_LIB") or os.path.join(_HERE, "librt_amd.so")  # override: experiment builds

RT_OK = 0
RT_ERR_INVALID = -1
RT_ERR_NO_DEVICE = -2
RT_ERR_HIP = -3
RT_ERR_IO = -4
RT_ERR_OOM = -5

RT_MAT_OPAQUE = 0
RT_MAT_DIELECTRIC = 1
RT_OBJ_MESH = 0
RT_OBJ_SPHERE = 1

RT_FLAG_NO_JITTER = 1
RT_FLAG_TIMING = 2
RT_FLAG_COUNT_WORK = 4
RT_FLAG_NO_REGEN = 8
RT_FLAG_MEGAKERNEL = 16
RT_FLAG_PIPELINE = 32
RT_FLAG_STORE = 64

RT_TREE_HOST = 0
RT_TREE_NONE = 1

RT_BUILD_LBVH = 0
RT_BUILD_PLOC = 1
RT_BUILD_DEFAULT = RT_BUILD_PLOC

RT_GROUP_MAX = 16
RT_NOALLOC_SCRATCH_MIB = 320
RT_SPLIT_SAMPLES = 0
RT_SPLIT_TILES = 1


class rt_material(C.Structure):
    _fields_ = [
        ("kind", C.c_int32),
        ("color_tex", C.c_int32),
        ("normal_tex", C.c_int32),
        ("reserved", C.c_int32),
        ("metallic", C.c_double),
        ("diffuse", C.c_double),
        ("how_metallic", C.c_double),
        ("emission", C.c_double * 3),
        ("color", C.c_double * 3),
        ("ior", C.c_double),
    ]


class rt_texture(C.Structure):
    _fields_ = [("rgba", C.c_void_p), ("width", C.c_uint32), ("height", C.c_uint32)]


class rt_scene_desc(C.Structure):
    _fields_ = [
        ("n_objects", C.c_uint32),
        ("object_kind", C.c_void_p),
        ("materials", C.c_void_p),
        ("n_triangles", C.c_uint64),
        ("tri_vertices", C.c_void_p),
        ("tri_normals", C.c_void_p),
        ("tri_uvs", C.c_void_p),
        ("tri_obj", C.c_void_p),
        ("tri_idx", C.c_void_p),
        ("n_spheres", C.c_uint32),
        ("spheres", C.c_void_p),
        ("sphere_obj", C.c_void_p),
        ("n_textures", C.c_uint32),
        ("textures", C.c_void_p),
    ]


class rt_hit(C.Structure):
    _fields_ = [("t", C.c_double), ("u", C.c_double), ("v", C.c_double), ("obj", C.c_int32), ("tri", C.c_int32)]


class rt_render_params(C.Structure):
    _fields_ = [
        ("width", C.c_uint32),
        ("height", C.c_uint32),
        ("spp", C.c_uint32),
        ("first_sample", C.c_uint32),
        ("seed", C.c_uint64),
        ("max_depth", C.c_int32),
        ("flags", C.c_uint32),
        ("fov_y_deg", C.c_double),
        ("dof_strength", C.c_double),
        ("spp_total", C.c_uint32),
        ("tile_size", C.c_uint32),
        ("tile_rank", C.c_uint32),
        ("tile_ranks", C.c_uint32),
    ]


class rt_scene_info(C.Structure):
    _fields_ = [
        ("n_triangles", C.c_uint64),
        ("n_spheres", C.c_uint32),
        ("n_objects", C.c_uint32),
        ("n_textures", C.c_uint32),
        ("bvh_nodes", C.c_uint32),
        ("bvh_leaves", C.c_uint32),
        ("bvh_depth", C.c_uint32),
        ("uploaded", C.c_uint32),
        ("bvh_spheres", C.c_uint32),
        ("reserved", C.c_uint32),
        ("device_bytes", C.c_uint64),
    ]


class rt_stats(C.Structure):
    _fields_ = [
        ("rays", C.c_uint64),
        ("node_visits", C.c_uint64),
        ("tri_tests", C.c_uint64),
        ("sphere_tests", C.c_uint64),
        ("shaded_hits", C.c_uint64),
        ("paths", C.c_uint64),
        ("waves", C.c_uint64),
        ("wave_iters_inner", C.c_uint64),
        ("wave_iters_leaf", C.c_uint64),
        ("wave_iters_reject", C.c_uint64),
    ]


class rt_timing(C.Structure):
    _fields_ = [
        ("ms_paths", C.c_float),
        ("ms_resolve", C.c_float),
        ("path_launches", C.c_uint32),
        ("resolve_launches", C.c_uint32),
    ]


class rt_launch_info(C.Structure):
    _fields_ = [
        ("kind", C.c_uint32),
        ("items", C.c_uint32),
        ("ms", C.c_float),
        ("folded", C.c_uint32),
        ("rays", C.c_uint64),
        ("node_visits", C.c_uint64),
        ("tri_tests", C.c_uint64),
        ("waves", C.c_uint64),
        ("wave_iters_inner", C.c_uint64),
        ("wave_iters_leaf", C.c_uint64),
        ("wave_iters_reject", C.c_uint64),
    ]


# Every symbol include/rt_amd.h declares: name -> (restype, argtypes)
_P = C.c_void_p
SIGNATURES = {
    "rt_mesh_load_obj": (C.c_int, [C.c_char_p, C.POINTER(_P)]),
    "rt_mesh_clone": (C.c_int, [_P, C.POINTER(_P)]),
    "rt_mesh_destroy": (None, [_P]),
    "rt_mesh_counts": (C.c_int, [_P, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    "rt_mesh_rotate_x": (C.c_int, [_P, C.c_double]),
    "rt_mesh_rotate_y": (C.c_int, [_P, C.c_double]),
    "rt_mesh_rotate_z": (C.c_int, [_P, C.c_double]),
    "rt_mesh_translate": (C.c_int, [_P, C.POINTER(C.c_double)]),
    "rt_image_load_png": (C.c_int, [C.c_char_p, C.POINTER(_P)]),
    "rt_image_destroy": (None, [_P]),
    "rt_image_info": (C.c_int, [_P, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.POINTER(_P)]),
    "rt_image_write_png": (C.c_int, [C.c_char_p, _P, C.c_uint32, C.c_uint32]),
    "rt_objects_create": (C.c_int, [C.POINTER(_P)]),
    "rt_objects_destroy": (None, [_P]),
    "rt_objects_add_texture": (C.c_int, [_P, _P, C.c_uint32, C.c_uint32]),
    "rt_objects_push_mesh": (C.c_int, [_P, _P, C.POINTER(rt_material)]),
    "rt_objects_push_sphere": (C.c_int, [_P, C.POINTER(C.c_double), C.c_double, C.POINTER(rt_material)]),
    "rt_scene_file_load": (C.c_int, [C.c_char_p, C.POINTER(_P), _P, C.POINTER(C.c_uint32)]),
    "rt_render_params_default": (None, [_P]),
    "rt_scene_from_objects": (C.c_int, [_P, C.POINTER(_P)]),
    "rt_scene_create": (C.c_int, [C.POINTER(rt_scene_desc), C.POINTER(_P)]),
    "rt_scene_from_objects_with": (C.c_int, [_P, C.c_uint32, C.POINTER(_P)]),
    "rt_scene_create_with": (C.c_int, [C.POINTER(rt_scene_desc), C.c_uint32, C.POINTER(_P)]),
    "rt_scene_destroy": (None, [_P]),
    "rt_scene_get_info": (C.c_int, [_P, C.POINTER(rt_scene_info)]),
    "rt_scene_export_flat": (C.c_int, [_P, _P, _P, _P, _P, _P]),
    "rt_scene_export_objects": (C.c_int, [_P, _P, _P, _P, _P]),
    "rt_scene_export_texture": (C.c_int, [_P, C.c_uint32, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), _P]),
    "rt_scene_export_bvh": (C.c_int, [_P, _P, _P]),
    "rt_scene_export_grid_bvh": (C.c_int, [_P, _P, _P, _P]),
    "rt_device_count": (C.c_int, []),
    "rt_scene_rebuild_on_device": (C.c_int, [_P, C.c_int]),
    "rt_scene_rebuild_on_device_with": (C.c_int, [_P, C.c_int, C.c_uint32]),
    "rt_scene_upload": (C.c_int, [_P, C.c_int]),
    "rt_trace": (C.c_int, [_P, _P, _P, C.c_size_t, _P]),
    "rt_primary_hits": (C.c_int, [_P, C.POINTER(rt_render_params), _P, _P, _P, _P, _P]),
    "rt_render_accumulate": (C.c_int, [_P, C.POINTER(rt_render_params), _P, C.POINTER(C.c_uint64)]),
    "rt_render_device": (C.c_int, [_P, C.POINTER(rt_render_params), _P, _P]),
    "rt_render_device_noalloc": (C.c_int, [_P, C.POINTER(rt_render_params), _P, _P]),
    "rt_render_sum_f32_device": (C.c_int, [_P, C.POINTER(rt_render_params), _P, _P]),
    "rt_scene_wait": (C.c_int, [_P]),
    "rt_scene_stream_wait": (C.c_int, [_P, _P]),
    "rt_group_create": (C.c_int, [_P, C.POINTER(C.c_int), C.c_uint32, C.POINTER(_P)]),
    "rt_group_destroy": (None, [_P]),
    "rt_group_size": (C.c_int, [_P, C.POINTER(C.c_uint32)]),
    "rt_group_render_accumulate": (C.c_int, [_P, C.POINTER(rt_render_params), C.c_uint32, _P, C.POINTER(C.c_uint64)]),
    "rt_group_render_device": (C.c_int, [_P, C.POINTER(rt_render_params), C.c_uint32, _P, _P]),
    "rt_group_wait": (C.c_int, [_P, C.POINTER(C.c_uint64)]),
    "rt_group_member_info": (C.c_int, [_P, C.c_uint32, C.POINTER(C.c_int), C.POINTER(C.c_uint64), C.POINTER(C.c_double)]),
    "rt_scene_last_stats": (C.c_int, [_P, C.POINTER(rt_stats)]),
    "rt_scene_last_timing": (C.c_int, [_P, C.POINTER(rt_timing)]),
    "rt_scene_last_launches": (C.c_int, [_P, C.POINTER(rt_launch_info), C.c_uint32, C.POINTER(C.c_uint32)]),
    "rt_quantize_rgba8": (C.c_int, [_P, C.c_uint32, C.c_uint32, C.c_uint32, _P]),
    "rt_selftest_division": (C.c_int, [C.c_uint64, C.c_uint64, C.POINTER(C.c_uint64)]),
    "rt_last_error": (C.c_char_p, []),
    "rt_source_digest": (C.c_char_p, []),
}


def _share_torch_hip_runtime():
    """A PyTorch-ROCm wheel carries its own libamdhip64.so, which torch's libraries ask for by file
    name. If this library pulled in the system's copy first, a later `import torch` would load a
    second HIP runtime into the process, and that one finds no GPU. Loading torch's copy (if there
    is one) before librt_amd.so makes both resolve to the same runtime, whichever is imported
    first. Does not import torch."""
    import importlib.util

    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    for loc in (spec.submodule_search_locations or []) if spec else []:
        cand = os.path.join(loc, "lib", "libamdhip64.so")
        if os.path.exists(cand):
            try:
                C.CDLL(cand, mode=C.RTLD_GLOBAL)
            except OSError:
                pass
            return


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(or `make -C raytracer_amd/csrc`). There is no fallback implementation."
        )
    _share_torch_hip_runtime()
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the library does not export the symbol
        fn.restype = res
        fn.argtypes = args
    return lib


lib = _load()


class RtError(RuntimeError):
    def __init__(self, code, where):
        msg = lib.rt_last_error()
        super().__init__(f"{where} failed with code {code}: {msg.decode() if msg else ''}")
        self.code = code


def check(code, where):
    if code < 0:
        raise RtError(code, where)
    return code
