"""make_bvh on the GPU (second half; reference bvh.cpp:288-358): shray_bvh_build_device
(shader-ray_amd/csrc/bvh_build.hip) builds the tree level by level on the device; the tree, every box and the post-build order of
the triangles must equal the host builder's -- which is pinned, bit for bit, to the reference's own (tests/test_host_vs_reference.py)
-- because the flattened arrays are the data contract of the tracer: a leaf's triangle ORDER decides ties between equal hit
distances (fs:333-340).  Compared here as the pre-order arrays of shray_tree_desc and as the flattened scene_shader_data arrays."""
import ctypes as C
import os
import subprocess
import sys

import numpy as np
import pytest

import helpers
from test_gpu_fuzz import soup

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")


def tree_arrays(tree):
    n, t = tree.node_count, tree.triangle_count
    take = lambda ptr, count, dtype: np.ctypeslib.as_array(ptr, shape=(count,)).view(dtype).copy()   # noqa: E731
    return {"parent": take(tree.node_parent, n, np.int32), "negative": take(tree.node_negative, n, np.int32),
            "positive": take(tree.node_positive, n, np.int32), "box": take(tree.node_box, 6 * n, np.uint32),
            "direction": take(tree.node_direction, 3 * n, np.uint32), "start": take(tree.node_start, n, np.int32),
            "triangles": take(tree.node_triangles, n, np.int32), "triangle_vertices": take(tree.triangle_vertices, 3 * t, np.int32)}


def assert_same_world(host, device, what):
    a, b = tree_arrays(host.export_tree()), tree_arrays(device.export_tree())
    for key in a:
        assert a[key].shape == b[key].shape, (what, key, a[key].shape, b[key].shape)
        differing = int((a[key] != b[key]).sum())
        assert differing == 0, f"{what}: {differing} entries of the tree's {key} differ from the host build's"
    for field in ("node_count", "leaf_count", "max_level", "large_leaves", "triangle_count"):
        assert getattr(host.info, field) == getattr(device.info, field), (what, field)
    ha, da = host.arrays(), device.arrays()
    for key, value in ha.items():
        if isinstance(value, np.ndarray):
            assert np.array_equal(value.view(np.uint32), da[key].view(np.uint32)), (what, key)
        else:
            assert value == da[key], (what, key)


@pytest.mark.parametrize("name", ["lobed_528.trisrc", "quads_mixed.obj", "quads_nonormals.obj"])
def test_device_build_equals_the_host_build_on_the_golden_scenes(pkg, gpu, name):
    path = os.path.join(GOLDEN, name)
    host, device = pkg.World(path), pkg.World(path, build="gpu")
    assert_same_world(host, device, name)
    # and the reference's own dump of the flattened arrays (made by the compiled reference: tests/golden/*.ref.npz)
    ref = dict(np.load(os.path.splitext(path)[0] + ".ref.npz"))
    mine = device.arrays()
    for key in ("group_boxmin", "group_boxmax", "group_objects", "vertex_positions", "group_hitmiss_0", "group_hitmiss_7"):
        assert np.array_equal(np.ascontiguousarray(ref[key], np.float32).view(np.uint32), mine[key].view(np.uint32)), key
    host.close()
    device.close()


def test_device_build_of_the_benchmark_scenes(pkg, gpu, oracle_mod):
    """The 69k-triangle benchmark mesh and the 1M-triangle OBJ (depth 26, 291k nodes): the same tree, and a frame of the
    device-built world equals the oracle's frame of the host-built one."""
    for path in (helpers.bunny_trisrc(), helpers.million_obj()):
        host, device = pkg.World(path), pkg.World(path, build="gpu")
        assert_same_world(host, device, os.path.basename(path))
        assert device.bvh_device_seconds is not None and device.bvh_device_seconds > 0
        if host.triangle_count < 100000:
            env = pkg.scenes.environment_hdr_sky(128)
            params = device.frame_params(160, 96, material=6)
            want, _ = oracle_mod.render(host.flatten(), env, params, 160, 96, 1)
            scene = pkg.Scene(device.flatten(), env, device=0)
            assert np.array_equal(scene.render(params, 160, 96, 1).view(np.uint32), want.view(np.uint32))
            scene.close()
        host.close()
        device.close()


# 14 soups in the suite; SHRAY_FUZZ_ROUNDS=n builds n times as many (seeds go on counting)
@pytest.mark.parametrize("seed,kind", [(k, kind) for k, kind in enumerate(["uniform", "clusters", "sizes", "duplicates", "planes", "degenerate", "dense"]
                                                                           * (2 * max(1, int(os.environ.get("SHRAY_FUZZ_ROUNDS", "1")))))])
def test_device_build_of_random_soups(pkg, gpu, tmp_path, seed, kind):
    """Seeded random triangle soups (tests/test_gpu_fuzz.py's generator): duplicates (no split separates them: large leaves),
    axis-aligned sheets (boxes of zero thickness, barycentres on one plane), degenerate and huge triangles."""
    rng = np.random.default_rng(5000 + seed)
    pos, tri = soup(rng, kind)
    path = str(tmp_path / f"soup_{seed}.trisrc")
    pkg.scenes.write_trisrc(path, pos, tri)
    host, device = pkg.World(path), pkg.World(path, build="gpu")
    assert_same_world(host, device, f"{kind} soup {seed}")
    host.close()
    device.close()


def test_small_and_odd_inputs(pkg, gpu, tmp_path):
    """One triangle, eleven identical triangles (count > leaf_max and nothing to split: a large leaf), twelve triangles in a row
    (one split), and a refusal: a triangle that names a vertex that does not exist."""
    N = pkg._native
    one = np.asarray([[0, 0, 0], [1, 0, 0], [0, 1, 0]], np.float32)
    cases = {"one": one, "eleven identical": np.tile(one, (11, 1)),
             "a row of twelve": np.concatenate([one + [2.5 * k, 0, 0] for k in range(12)])}
    for what, pos in cases.items():
        path = str(tmp_path / (what.replace(" ", "_") + ".trisrc"))
        pkg.scenes.write_trisrc(path, pos.astype(np.float32), np.arange(len(pos)).reshape(-1, 3))
        host, device = pkg.World(path), pkg.World(path, build="gpu")
        assert_same_world(host, device, what)
        host.close()
        device.close()
    hip = N.load_hip()
    bad = (C.c_int32 * 3)(0, 1, 7)
    verts = (C.c_float * 27)(*([0.0] * 27))
    handle = C.c_void_p()
    assert hip.shray_bvh_build_device(bad, 1, verts, 3, 9, None, C.byref(handle)) == -1 and not handle      # SHRAY_ERR_INVALID_ARGUMENT
    assert b"vertex that does not exist" in hip.shray_last_error()


def test_build_options_follow_the_reference_environment(pkg, gpu, tmp_path):
    """BVH_MAX_DEPTH, BVH_LEAF_MAX, SAH_CTRAV, SAH_CISEC (bvh.cpp:60-79) are read once per process by the host builder; the device
    build takes them as shray_bvh_options.  In a child process: a shallow tree with large leaves, and another SAH."""
    script = r'''
import sys, numpy as np
sys.path[:0] = [%r, %r]
from __graft_entry__ import load_package
import helpers
from test_gpu_bvh_build import assert_same_world
pkg = load_package()
N = pkg._native
options = N.BvhOptions(0, 6, 3, 2.0, 1.5)
host = pkg.World(helpers.small_trisrc())
device = pkg.World(helpers.small_trisrc(), build="gpu", options=options)
assert host.info.max_level <= 6 and host.info.large_leaves >= 0
assert_same_world(host, device, "max depth 6, leaf max 3, SAH 2 + 1.5 n")
print("ok", host.info.node_count, host.info.max_level)
''' % (ROOT, os.path.join(ROOT, "tests"))
    env = dict(os.environ, BVH_MAX_DEPTH="6", BVH_LEAF_MAX="3", SAH_CTRAV="2.0", SAH_CISEC="1.5")
    run = subprocess.run([sys.executable, "-c", script], env=env, capture_output=True, text=True, timeout=600)
    assert run.returncode == 0 and run.stdout.startswith("ok"), run.stderr[-3000:]


def test_the_device_tree_feeds_the_device_flattener(pkg, gpu):
    """File -> triangles on the host; BVH and flattening both on the device (shray_bvh_build_device -> shray_flatten_device): the
    arrays of the host's load_world + get_shader_data."""
    N = pkg._native
    hip, lib = N.load_hip(), N.load_host()
    path = helpers.bunny_trisrc()
    handle = C.c_void_p()
    assert lib.shray_host_load_triangles(path.encode(), C.byref(handle)) == 0
    tv, vd, nt, nv = C.POINTER(C.c_int32)(), C.POINTER(C.c_float)(), C.c_int32(), C.c_int32()
    assert lib.shray_host_triangles(handle, C.byref(tv), C.byref(nt), C.byref(vd), C.byref(nv)) == 0
    built = C.c_void_p()
    N.check(hip.shray_bvh_build_device(tv, nt, vd, nv, 9, None, C.byref(built)))
    tree = N.TreeDesc()
    N.check(hip.shray_device_tree_download(built, C.byref(tree), None))
    flat = pkg.tracer.DeviceFlat(tree)
    mine = flat.arrays()
    host = pkg.World(path)
    want = host.arrays()
    for key, value in want.items():
        if isinstance(value, np.ndarray):
            assert np.array_equal(value.view(np.uint32), mine[key].view(np.uint32)), key
        else:
            assert value == mine[key], key
    flat.close()
    hip.shray_device_tree_destroy(built)
    lib.shray_host_free_world(handle)
    host.close()


def test_render_tool_with_the_device_built_tree(gpu, tmp_path):
    """tools/shray_render -b gpu / -b device: the reference-shaped main() with make_bvh replaced by the device build -- and, for
    `device`, get_shader_data and the upload by the device-resident pipeline -- writes the same picture."""
    exe = os.path.join(ROOT, "shader-ray_amd", "tools", "shray_render")
    if not os.path.exists(exe):
        subprocess.run(["make", "-C", os.path.dirname(os.path.dirname(exe)), "tools"], check=True, stdout=subprocess.DEVNULL)
    model = os.path.join(GOLDEN, "quads_nonormals.obj")
    pictures = []
    for where in ("host", "gpu", "device"):     # device: build, flattening and scene creation on the device, nothing downloaded (round 6)
        out = str(tmp_path / f"{where}.ppm")
        run = subprocess.run([exe, model, "grid", "-o", out, "-w", "96", "-h", "64", "-m", "6", "-b", where], capture_output=True, text=True)
        assert run.returncode == 0, run.stderr[-2000:]
        pictures.append(open(out, "rb").read())
        assert ("the tree never left the device" in run.stderr) == (where == "device")
    assert len(pictures[0]) > 96 * 64 * 3 and pictures[0] == pictures[1] == pictures[2]
