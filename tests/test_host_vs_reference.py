"""The repo's own host layer (loaders, BVH build, flattener, frame parameters) against the
REFERENCE's outputs for the same files: committed fixtures (tests/golden/*.ref.npz, dumped
by the compiled reference, see tests/golden/make_golden.py) and, when oracle/_ref/ref_host
is present, a live run on the benchmark-sized scenes.  Everything is compared bit for bit."""
import os
import subprocess

import numpy as np
import pytest

import helpers
from refdump import read_dump

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_HOST = os.path.join(ROOT, "oracle", "_ref", "ref_host")

ARRAYS = ["vertex_positions", "vertex_normals", "vertex_colors", "group_boxmin", "group_boxmax", "group_children",
          "group_objects", "group_directions"] + [f"group_hitmiss_{c}" for c in range(8)]
SCALARS = ["vertex_count", "vertex_data_rows", "group_count", "group_data_rows", "tree_root"]
MATRICES = ["camera_matrix", "camera_normal_matrix", "object_matrix", "object_inverse", "object_normal_matrix",
            "object_normal_inverse"]


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def check_world_against(pkg, scene_path, ref):
    world = pkg.World(scene_path)
    mine = world.arrays()
    for k in SCALARS:
        assert int(ref[k][0]) == int(mine[k]), k
    for k in ARRAYS:
        assert mine[k].shape == ref[k].shape, k
        diff = int((bits(mine[k]) != bits(ref[k])).sum())
        assert diff == 0, f"{k}: {diff} of {mine[k].size} floats differ from the reference"
    assert int(ref["triangle_count"][0]) == world.info.triangle_count
    assert int(ref["independent_vertices"][0]) == world.info.independent_vertex_count
    assert np.array_equal(bits(ref["scene_center"]), bits(np.array(world.info.scene_center[:])))
    assert np.float32(ref["scene_extent"][0]) == np.float32(world.info.scene_extent)

    # start-up view, 1920x1080 (ray.cpp:1077-1088, :648-704)
    view = world.default_view()
    assert np.float32(view.zoom) == ref["zoom"][0] and np.float32(view.fov) == ref["fov"][0]
    fp = world.frame_params(1920, 1080)
    for k in MATRICES + ["right", "up", "light_dir"]:
        assert np.array_equal(bits(np.array(getattr(fp, k)[:])), bits(ref[k])), k
    assert np.float32(fp.image_plane_width) == ref["image_plane_width"][0]
    assert np.float32(fp.aspect) == ref["aspect"][0]

    # a rotated object + rotated light + closer camera (values fixed in oracle/ref_driver.cpp)
    view.object_rotation[:] = [0.9, 0.26726124, 0.53452248, 0.80178373]
    view.light_rotation[:] = [1.1, 0.0, 0.6, 0.8]
    view.zoom = float(np.float32(view.zoom) * np.float32(0.75))
    fp = world.frame_params(1920, 1080, view)
    for k in MATRICES + ["light_dir"]:
        assert np.array_equal(bits(np.array(getattr(fp, k)[:])), bits(ref["v2_" + k])), "v2_" + k
    world.close()


@pytest.mark.parametrize("name", ["lobed_528.trisrc", "quads_mixed.obj", "quads_nonormals.obj"])
def test_against_committed_reference_dumps(pkg, name):
    path = os.path.join(GOLDEN, name)
    ref = dict(np.load(os.path.splitext(path)[0] + ".ref.npz"))
    check_world_against(pkg, path, ref)


@pytest.mark.skipif(not os.path.exists(REF_HOST), reason="oracle/_ref/ref_host not built (needs /root/reference)")
@pytest.mark.parametrize("which", ["bunny", "obj_small"])
def test_against_live_reference(pkg, tmp_path, which):
    path = helpers.bunny_trisrc() if which == "bunny" else helpers.small_obj_no_normals()
    dump = str(tmp_path / "ref.bin")
    subprocess.run([REF_HOST, path, dump, "1920", "1080"], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    check_world_against(pkg, path, read_dump(dump))


def test_survey_probe_values_unit_sphere(pkg, tmp_path):
    """golden values measured with the compiled reference for a
    unit sphere at 1920x1080."""
    pos, tri = pkg.scenes.lobed_sphere_mesh(16, 32, bumpiness=0.0, ears=False)
    path = str(tmp_path / "unit.trisrc")
    pkg.scenes.write_trisrc(path, pos, tri)
    world = pkg.World(path)
    assert abs(world.info.scene_extent - 2.0) < 1e-3
    view = world.default_view()
    fp = world.frame_params(1920, 1080)
    assert abs(view.zoom - 2.92380619) < 2e-3          # extent is 2 only up to tessellation
    assert np.float32(fp.image_plane_width) == np.float32(0.72794044)
    assert np.float32(fp.aspect) == np.float32(0.5625)
    assert np.float32(fp.right[0]) == np.float32(0.000379135658) == np.float32(fp.up[1])
    assert np.allclose(fp.light_dir[:], [0.241816789, 0.241816789, 0.939725816], rtol=0, atol=1e-7)
    ident = np.eye(4, dtype=np.float32).reshape(-1)
    cam = np.array(fp.camera_matrix[:], dtype=np.float32)
    assert cam[14] == np.float32(view.zoom) and np.array_equal(np.delete(cam, 14), np.delete(ident, 14))
