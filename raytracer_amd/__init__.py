"""MI355X (gfx950) implementation of the skius/raytracer render hot path, behind a C ABI.

`raytracer_amd.host` mirrors the reference's scene interface (load_obj, PolygonMesh, Object,
Scene::from_objects, Scene::hit, render); `raytracer_amd.scenes` assembles the reference's
scenes. Everything executes in librt_amd.so (HIP kernels + C++ host code); importing this
package fails if that library has not been built.
"""
from . import _ffi  # noqa: F401  (loads librt_amd.so or raises)
from ._ffi import RtError  # noqa: F401
from .host import (  # noqa: F401
    Dielectric,
    Group,
    Objects,
    Opaque,
    PolygonMesh,
    Scene,
    device_count,
    load_obj,
    load_png,
    write_png,
    quantize_rgba8,
    render_params,
)
