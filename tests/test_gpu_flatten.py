"""GPU-side flattener (reference world.cpp:179-288, :298-347): the scene_shader_data arrays
computed by shader-ray_amd/csrc/flatten.hip equal the REFERENCE's own (tests/golden/*.ref.npz, dumped by the
compiled reference) bit for bit, a scene created from them renders the frame of the host-flattened scene, and
malformed trees are refused before any kernel runs."""
import ctypes as C
import os

import numpy as np
import pytest

import helpers

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
ARRAYS = ["vertex_positions", "vertex_normals", "vertex_colors", "group_boxmin", "group_boxmax", "group_children",
          "group_objects", "group_directions"] + [f"group_hitmiss_{c}" for c in range(8)]
SCALARS = ["vertex_count", "vertex_data_rows", "group_count", "group_data_rows", "tree_root"]


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


@pytest.mark.parametrize("name", ["lobed_528.trisrc", "quads_mixed.obj", "quads_nonormals.obj"])
def test_device_flattening_equals_the_reference_dump(pkg, gpu, name):
    path = os.path.join(GOLDEN, name)
    ref = dict(np.load(os.path.splitext(path)[0] + ".ref.npz"))
    world = pkg.World(path)
    flat = pkg.tracer.DeviceFlat(world.export_tree())
    mine = flat.arrays()
    for k in SCALARS:
        assert int(ref[k][0]) == int(mine[k]), k
    for k in ARRAYS:
        assert mine[k].shape == ref[k].shape, k
        assert int((bits(mine[k]) != bits(ref[k])).sum()) == 0, k
    flat.close()
    world.close()


def test_device_flattening_of_the_benchmark_scenes(pkg, gpu):
    """69k and 1M triangles: identical to the host flattener (itself pinned to the reference), padding included;
    the scene created from the device-flattened arrays renders the same frame."""
    for path in (helpers.bunny_trisrc(), helpers.million_obj()):
        world = pkg.World(path)
        host = world.flatten()
        flat = pkg.tracer.DeviceFlat(world.export_tree())
        dev = flat.download()
        for field in SCALARS + ["data_texture_width"]:
            assert getattr(host, field) == getattr(dev, field), field
        vt, nt = 3 * host.data_texture_width * host.vertex_data_rows, host.data_texture_width * host.group_data_rows
        for field, floats in (("vertex_positions", vt), ("vertex_normals", vt), ("vertex_colors", vt), ("group_boxmin", 3 * nt),
                              ("group_boxmax", 3 * nt), ("group_directions", 3 * nt), ("group_children", 2 * nt),
                              ("group_objects", 2 * nt), ("group_hitmiss", 16 * nt)):
            a = np.ctypeslib.as_array(getattr(host, field), shape=(floats,)).view(np.uint32)
            b = np.ctypeslib.as_array(getattr(dev, field), shape=(floats,)).view(np.uint32)
            assert np.array_equal(a, b), field
        if world.triangle_count < 100000:
            env = pkg.scenes.environment_hdr_sky(128)
            params = world.frame_params(160, 96, material=6)
            a = pkg.Scene(host, env, device=0)
            b = pkg.Scene(dev, env, device=0)
            assert np.array_equal(a.render(params, 160, 96, 2), b.render(params, 160, 96, 2))
            a.close()
            b.close()
        flat.close()
        world.close()


def test_malformed_trees_are_refused(pkg, gpu):
    N = pkg._native
    lib = N.load_hip()
    world = pkg.World(os.path.join(GOLDEN, "lobed_528.trisrc"))
    tree = world.export_tree()
    n = tree.node_count
    out = C.c_void_p()

    def attempt(mutate):
        t = N.TreeDesc.from_buffer_copy(tree)
        keep = mutate(t)
        rc = lib.shray_flatten_device(C.byref(t), 2048, C.byref(out))
        del keep
        return rc, lib.shray_last_error().decode()

    def swap_children(t):   # no longer pre-order
        neg = (C.c_int32 * n)(*tree.node_negative[:n])
        neg[0] = tree.node_positive[0]
        t.node_negative = C.cast(neg, C.POINTER(C.c_int32))
        return neg

    def leaf_out_of_range(t):
        cnt = (C.c_int32 * n)(*tree.node_triangles[:n])
        leaf = [g for g in range(n) if tree.node_negative[g] < 0][0]
        cnt[leaf] = tree.triangle_count + 5
        t.node_triangles = C.cast(cnt, C.POINTER(C.c_int32))
        return cnt

    def bad_size(t):
        t.struct_size = 4
        return None

    for mutate, word in ((swap_children, "pre-order"), (leaf_out_of_range, "outside the mesh"), (bad_size, "struct_size")):
        rc, message = attempt(mutate)
        assert rc != 0 and word in message, (rc, message)
    assert lib.shray_flatten_device(None, 2048, C.byref(out)) != 0
    world.close()
