import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _ensure_built():
    need = [os.path.join(ROOT, "raytracer_amd", "librt_amd.so"), os.path.join(ROOT, "oracle", "librt_oracle.so")]
    if all(os.path.exists(p) for p in need):
        return
    import __graft_entry__ as g

    g.build()


_ensure_built()


@pytest.fixture(scope="session")
def oracle():
    import oracle_ffi

    return oracle_ffi
