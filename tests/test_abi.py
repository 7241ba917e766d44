"""The C-ABI library loads and exports exactly what include/rt_amd.h declares (no compute)."""
import ctypes as C
import os
import re

from raytracer_amd import _ffi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    text = open(os.path.join(ROOT, "include", "rt_amd.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(rt_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_are_exported():
    names = _declared()
    assert len(names) >= 28
    lib = C.CDLL(_ffi.LIB_PATH)
    for n in names:
        assert hasattr(lib, n), f"{n} is declared in include/rt_amd.h but not exported"


def test_binding_covers_header():
    assert sorted(_ffi.SIGNATURES) == _declared()


def test_struct_sizes_match_header():
    # the layouts the header documents (natural alignment, no packing)
    assert C.sizeof(_ffi.rt_material) == 96
    assert C.sizeof(_ffi.rt_hit) == 32
    assert C.sizeof(_ffi.rt_texture) == 16
    assert C.sizeof(_ffi.rt_render_params) == 64
    assert C.sizeof(_ffi.rt_scene_desc) == 112
    assert C.sizeof(_ffi.rt_stats) == 80
    assert C.sizeof(_ffi.rt_timing) == 16


def test_last_error_is_a_string():
    assert isinstance(_ffi.lib.rt_last_error(), bytes)
    digest = _ffi.lib.rt_source_digest()
    assert isinstance(digest, bytes) and len(digest) == 12 and int(digest, 16) >= 0  # the build passed it
    assert _ffi.lib.rt_device_count() >= 0
