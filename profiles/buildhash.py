"""Hash of everything that determines the compiled kernels (csrc sources + the Makefile's flags), so that
bench.py can tell whether a committed rocprofv3 counter file (profiles/rNN/pmc_*.json) was measured on the
kernels it is timing."""
import glob
import hashlib
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def kernel_source_hash() -> str:
    h = hashlib.sha256()
    pkg = os.path.join(ROOT, "shader-ray_amd")
    for path in sorted(glob.glob(os.path.join(pkg, "csrc", "*"))) + [os.path.join(pkg, "Makefile")]:
        h.update(os.path.basename(path).encode())
        h.update(open(path, "rb").read())
    return h.hexdigest()[:16]
