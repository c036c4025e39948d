"""The device-resident scene pipeline (round 6): shray_bvh_build_device -> shray_flatten_device_tree ->
shray_scene_create_from_device -- the tree is built, flattened and turned into a scene without leaving the device.  Everything
the host path derives (shray_scene_create: the packed tree and its eight octant copies, the packed triangles, the pair records, the
fp16 normals, the deepest ray stack) is read back from both scenes and compared bit for bit; frames and work counters of the two
scenes are identical; the host's group tree (world.h:48-51) can still be had on demand."""
import ctypes as C
import os

import numpy as np
import pytest

import helpers

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def both_scenes(pkg, path):
    host_world = pkg.World(path)
    host_scene = pkg.Scene(host_world.flatten(), pkg.scenes.environment_constant())
    device = pkg.tracer.DeviceWorld(path, pkg.scenes.environment_constant())
    return host_world, host_scene, device


def assert_same_derived_arrays(host_scene, device_scene, what):
    want, got = host_scene.derived_arrays(), device_scene.derived_arrays()
    assert want["stack_levels"] == got["stack_levels"], (what, want["stack_levels"], got["stack_levels"])
    for key in ("packed_nodes", "packed_tris", "normals16", "pair_nodes"):
        assert want[key].shape == got[key].shape and want[key].size > 0, (what, key, want[key].shape, got[key].shape)
        differing = int((want[key] != got[key]).sum())
        assert differing == 0, f"{what}: {differing} words of {key} differ from the host path's"


@pytest.mark.parametrize("name", ["lobed_528.trisrc", "quads_mixed.obj", "quads_nonormals.obj"])
def test_device_pipeline_equals_the_host_path_on_the_golden_scenes(pkg, gpu, name):
    path = os.path.join(GOLDEN, name)
    host_world, host_scene, device = both_scenes(pkg, path)
    assert_same_derived_arrays(host_scene, device.scene, name)
    # the reference-layout arrays too (what the literal kernel reads), against the reference's own dump
    flat, ref = device.flat_arrays(), dict(np.load(os.path.splitext(path)[0] + ".ref.npz"))
    for key, value in host_world.arrays().items():
        if isinstance(value, np.ndarray):
            assert np.array_equal(value.view(np.uint32), flat[key].view(np.uint32)), key
    for key in ("group_boxmin", "group_boxmax", "group_objects", "vertex_positions", "group_hitmiss_0", "group_hitmiss_7"):
        assert np.array_equal(np.ascontiguousarray(ref[key], np.float32).view(np.uint32), flat[key].view(np.uint32)), key
    # frames and work counters: every kernel, gold and plaster
    for material in (0, 6):
        params = host_world.frame_params(96, 64, material=material)
        mine = device.frame_params(96, 64, material=material)
        assert bytes(params) == bytes(mine)          # (the frame block needs the mesh's extent, not its tree)
        for kernel in (0, 1):
            host_scene.set_kernel(kernel)
            device.scene.set_kernel(kernel)
            want, want_counters = host_scene.render_counters(params, 96, 64, 1)
            got, got_counters = device.scene.render_counters(params, 96, 64, 1)
            assert np.array_equal(want.view(np.uint32), got.view(np.uint32)) and want_counters == got_counters, (name, material, kernel)
    for obj in (host_scene, host_world, device):
        obj.close()


def test_device_pipeline_on_the_benchmark_scenes(pkg, gpu, oracle_mod):
    """BASELINE's two scenes: 69,168 and 1,000,000 triangles.  The derived arrays equal the host path's; a frame of the bunny-class
    scene equals the oracle's; the host's group tree, asked for afterwards, is the host build's."""
    for path, what in ((helpers.bunny_trisrc(), "bunny-class"), (helpers.million_obj(), "1M triangles")):
        host_world, host_scene, device = both_scenes(pkg, path)
        assert_same_derived_arrays(host_scene, device.scene, what)
        assert device.stats.node_count == host_world.info.node_count and device.stats.max_level == host_world.info.max_level
        assert device.seconds["triangles_to_resident"] > 0
        if what == "bunny-class":
            params = device.frame_params(160, 120, material=0)
            want, counters = oracle_mod.render(host_world.flatten(), pkg.scenes.environment_constant(), params, 160, 120, 1)
            got, got_counters = device.scene.render_counters(params, 160, 120, 1)
            assert np.array_equal(want.view(np.uint32), got.view(np.uint32))
            assert all(got_counters[k] == counters[k] for k in ("node_visits", "leaf_visits", "triangle_tests", "shaded_hits"))
        # the reference's `world`, on demand: adopt the device's tree and flatten on the host -- the host build's arrays
        handle = device.host_world()
        desc = pkg._native.SceneDesc()
        assert pkg._native.load_host().shray_host_flatten(handle, 2048, C.byref(desc)) == 0
        mine, want = pkg.host.desc_arrays(desc), host_world.arrays()
        for key, value in want.items():
            if isinstance(value, np.ndarray):
                assert np.array_equal(value.view(np.uint32), mine[key].view(np.uint32)), (what, key)
        for obj in (host_scene, host_world, device):
            obj.close()


def test_device_pipeline_refuses_what_it_cannot_take(pkg, gpu):
    """A flattening of another tree, NULL handles: error codes, no crash."""
    N = pkg._native
    hip = N.load_hip()
    out = C.c_void_p()
    assert hip.shray_scene_create_from_device(None, None, C.byref(out)) != 0 and not out
    assert hip.shray_flatten_device_tree(None, 2048, C.byref(out)) != 0 and not out
    small = pkg.tracer.DeviceWorld(os.path.join(GOLDEN, "lobed_528.trisrc"))
    other = pkg.tracer.DeviceWorld(os.path.join(GOLDEN, "quads_mixed.obj"))
    assert hip.shray_scene_create_from_device(small._tree, other._flat, C.byref(out)) != 0 and not out
    assert b"not those of this tree" in hip.shray_last_error()
    small.close()
    other.close()
