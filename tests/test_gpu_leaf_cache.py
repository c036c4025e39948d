"""The library with the leaf cache compiled in (shader-ray_amd/libshray_hip_leafcache.so: a leaf stage's distinct leaves
fetched once, as consecutive 16-byte chunks, straight into LDS and read from there -- north_star's "triangle data staged
into LDS tiles with coalesced loads", csrc/leaf_cache.h).  It is not the shipped library (measured slower,
profiles/EXPERIMENTS.md R5.1); this test keeps it parity-green: the oracle-parity tests and the fuzz run once more against
it, in ONE child process (SHRAY_HIP_LIB selects the library when the package is first imported)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "shader-ray_amd", "libshray_hip_leafcache.so")

# the tests that put rays through every leaf-stage path: whole frames against the oracle (one and several samples, gold and
# plaster: the dealt, the crowded and the plain stages), work counters of the timed instances and of the counting twins, leaves
# with more triangles than the cache takes (it must stand aside), NaN candidates, ranges' ends, the 1M-triangle tree, the fuzz
SELECTION = ("bunny_full_path or counters_of_the_timed_instances or nan_candidates or ends_of_a_leaf_range or "
             "depth_capped_tree_with_large_leaves or million_triangle_scene or sample_lanes_every_group_size or "
             "frame_batches_equal_single_launches or hand_built_edge_cases or kernels_match_the_reference_shaders or fuzz")


def test_the_leaf_cache_build_is_bit_identical(gpu):
    assert os.path.exists(LIB), f"{LIB} is not built (make -C shader-ray_amd leafcache)"
    env = dict(os.environ, SHRAY_HIP_LIB=LIB)
    run = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_gpu_parity.py"),
                          os.path.join(ROOT, "tests", "test_gpu_fuzz.py"), "-x", "-q", "-m", "gpu", "-k", SELECTION,
                          "-p", "no:cacheprovider"], env=env, capture_output=True, text=True, timeout=1500, cwd=ROOT)
    tail = (run.stdout + run.stderr)[-3000:]
    assert run.returncode == 0, tail
    assert " passed" in run.stdout and "failed" not in run.stdout, tail
