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
