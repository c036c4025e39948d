import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def pkg():
    from __graft_entry__ import load_package
    import subprocess
    pkg_dir = os.path.join(ROOT, "shader-ray_amd")
    if not (os.path.exists(os.path.join(pkg_dir, "libshray_host.so")) and os.path.exists(os.path.join(pkg_dir, "libshray_hip.so"))):
        subprocess.run(["make", "-C", pkg_dir, "-j4", "host", "hip"], check=True, stdout=subprocess.DEVNULL)
    return load_package()


@pytest.fixture(scope="session")
def oracle_mod():
    import oracle
    oracle.load()
    return oracle


@pytest.fixture(scope="session")
def gpu(pkg):
    """The HIP layer with a device behind it; GPU tests fail (not skip) if it is missing."""
    import ctypes as C
    lib = pkg._native.load_hip()
    n = C.c_int()
    rc = lib.shray_device_count(C.byref(n))
    assert rc == 0 and n.value >= 1, "GPU tests need a HIP device: " + lib.shray_last_error().decode()
    return lib
