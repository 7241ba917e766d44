"""The kernels divide a vector by its length and colour bytes by 255 with one refined reciprocal
per denominator (rt_kernels.hip.h: div3_by_length, div3_in_range). Results must be the bits of
the plain IEEE divisions src/main.rs:87-90, 359, 404-406 perform: checked on the device over two
thousand million operand triples (moderate magnitudes, the whole exponent range, zeros of either
sign, denormals, single tiny components, lengths at both edges of the fast range, raw bit patterns
with NaNs and infinities)."""
import pytest

import raytracer_amd as rt
from raytracer_amd import host

pytestmark = pytest.mark.gpu


def test_shared_reciprocal_division_is_ieee_division():
    if rt.device_count() < 1:
        pytest.fail("no HIP device visible")
    for seed in (0, 0x1234567):
        assert host.selftest_division(1 << 30, seed) == 0
