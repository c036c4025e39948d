"""The kernel's exact 'divide by a per-ray constant' against true IEEE division, on the GPU
(csrc/exact_div.h, csrc/kernel_selftest.hip): zero mismatches over 2^30 operand pairs."""
import ctypes as C

import pytest

pytestmark = pytest.mark.gpu


def test_division_by_constant_is_exact(pkg, gpu):
    bad = C.c_uint64(123)
    for seed in (1, 2):
        pkg._native.check(gpu.shray_selftest_division(1 << 29, seed, C.byref(bad)))
        assert bad.value == 0, f"{bad.value} of 2^29 quotients differ from true division (seed {seed})"


def test_three_instruction_reciprocal_is_exact_on_its_whole_domain(pkg, gpu):
    """reciprocal_in_range (v_rcp_f32 + one Newton step in FMAs; the triangle tests' 1 / det and the ray set-up's 1 / D)
    against true IEEE division on every float with 2^-100 <= |x| < 2^100: all 3.4e9 of them."""
    bad = C.c_uint64(123)
    pkg._native.check(gpu.shray_selftest_reciprocal(C.byref(bad)))
    assert bad.value == 0, f"{bad.value} reciprocals differ from true division"
