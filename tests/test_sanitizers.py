"""The host layer and the checker under AddressSanitizer + UndefinedBehaviorSanitizer (CPU build only -- the
GPU pool refuses sanitizers).  `make -C shader-ray_amd sanitize` and `make -C oracle sanitize` build _san/libshray_host.so and
oracle/_san/libshader_oracle.so; a CHILD process that preloads the two sanitizer runtimes loads them through SHRAY_HOST_LIB /
SHRAY_ORACLE_LIB, re-runs the loader / flattener / background / host-vs-reference / oracle KAT tests and then a seeded byte-mutation
fuzz of the trisrc, OBJ and .hdr loaders (tests/sanitizer_child.py).  A report of either sanitizer aborts the child."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _runtime(name):
    path = subprocess.run(["gcc", f"-print-file-name={name}"], capture_output=True, text=True).stdout.strip()
    return os.path.realpath(path) if os.path.isabs(path) and os.path.exists(path) else None


def test_loaders_flattener_and_oracle_under_asan_and_ubsan():
    asan, ubsan = _runtime("libasan.so"), _runtime("libubsan.so")
    if not (asan and ubsan):
        pytest.skip("this toolchain has no libasan / libubsan")
    subprocess.run(["make", "-C", os.path.join(ROOT, "shader-ray_amd"), "-j4", "sanitize"], check=True, stdout=subprocess.DEVNULL)
    subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "sanitize"], check=True, stdout=subprocess.DEVNULL)
    env = dict(os.environ,
               LD_PRELOAD=f"{asan}:{ubsan}",
               # (the interpreter and numpy leak by design; everything else is on)
               ASAN_OPTIONS="detect_leaks=0:abort_on_error=1:handle_segv=1",
               UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1",
               SHRAY_HOST_LIB=os.path.join(ROOT, "shader-ray_amd", "_san", "libshray_host.so"),
               SHRAY_ORACLE_LIB=os.path.join(ROOT, "oracle", "_san", "libshader_oracle.so"))
    run = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "sanitizer_child.py"), "60", "20261005"], env=env, capture_output=True,
                         text=True, timeout=900)
    tail = (run.stdout[-3000:] + "\n" + run.stderr[-6000:])
    assert "AddressSanitizer" not in run.stderr and "runtime error:" not in run.stderr, tail
    assert run.returncode == 0 and "SANITIZER_CHILD_OK" in run.stdout, tail
    # the fuzz met both outcomes for the text loaders: mutants that still load and mutants that are refused with a message
    tally = eval(run.stdout[run.stdout.index("SANITIZER_CHILD_OK") + len("SANITIZER_CHILD_OK"):].strip().splitlines()[0])   # noqa: S307 (our own child's repr)
    assert set(tally) == {"lobed_528.trisrc", "quads_mixed.obj", "quads_nonormals.obj", "seed.hdr"}
    assert all(loaded + refused == 60 for loaded, refused in tally.values())
    assert tally["lobed_528.trisrc"][1] > 0 and tally["quads_mixed.obj"][0] > 0 and tally["seed.hdr"][1] > 0, tally
