"""The C-ABI libraries load and export every symbol include/*.h declares; structure
layouts agree between the headers (as compiled) and the ctypes mirror; without a GPU
the HIP layer reports an error code instead of computing anything."""
import ctypes as C
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions(header):
    text = open(os.path.join(ROOT, "include", header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(shray_\w+)\s*\(", text)))


def test_hip_library_exports_every_declared_symbol(pkg):
    lib = pkg._native.load_hip()
    names = declared_functions("shader_ray_hip.h")
    assert len(names) >= 13
    for name in names:
        assert hasattr(lib, name), f"libshray_hip.so does not export {name}"
    assert sorted(n for n, _, _ in pkg._native.HIP_SYMBOLS) == names


def test_host_library_exports_every_declared_symbol(pkg):
    lib = pkg._native.load_host()
    names = [n for n in declared_functions("shader_ray_host.h")]
    for name in names:
        assert hasattr(lib, name), f"libshray_host.so does not export {name}"
    assert sorted(n for n, _, _ in pkg._native.HOST_SYMBOLS) == names


def test_dist_library_exports_every_declared_symbol(pkg):
    lib = pkg._native.load_dist()
    names = declared_functions("shader_ray_dist.h")
    assert len(names) >= 12
    for name in names:
        assert hasattr(lib, name), f"libshray_dist.so does not export {name}"
    assert sorted(n for n, _, _ in pkg._native.DIST_SYMBOLS) == names


def test_abi_version_matches_the_header(pkg):
    text = open(os.path.join(ROOT, "include", "shader_ray_hip.h")).read()
    declared = int(re.search(r"#define SHRAY_ABI_VERSION (\d+)", text).group(1))
    assert pkg._native.ABI_VERSION == declared == pkg._native.load_hip().shray_abi_version()


def test_struct_layouts_match_the_headers(pkg, tmp_path):
    """Compiles a tiny C program against the headers and compares sizeof/offsetof."""
    import subprocess
    src = tmp_path / "sizes.c"
    src.write_text(r'''
#include <stdio.h>
#include <stddef.h>
#include "shader_ray_hip.h"
#include "shader_ray_host.h"
#include "shader_ray_dist.h"
int main(void) {
  printf("%zu %zu %zu %zu %zu %zu\n", sizeof(shray_scene_desc), sizeof(shray_frame_params), sizeof(shray_tile_set),
         sizeof(shray_counters), sizeof(shray_host_view), sizeof(shray_host_world_info));
  printf("%zu %zu %zu %zu %zu %zu\n", sizeof(shray_dist_config), sizeof(shray_dist_xfer), sizeof(shray_dist_plan),
         sizeof(shray_dist_callbacks), offsetof(shray_dist_plan, owned_tiles), offsetof(shray_dist_config, transport));
  printf("%zu %zu %zu %zu\n", offsetof(shray_scene_desc, group_hitmiss), offsetof(shray_frame_params, image_plane_width),
         offsetof(shray_frame_params, bounce_count), offsetof(shray_host_view, which_material));
  return 0; }''')
    exe = tmp_path / "sizes"
    subprocess.run(["gcc", "-std=c99", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)], check=True)
    out = subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.split()
    N = pkg._native
    want = [C.sizeof(N.SceneDesc), C.sizeof(N.FrameParams), C.sizeof(N.TileSet), C.sizeof(N.Counters),
            C.sizeof(N.HostView), C.sizeof(N.HostWorldInfo),
            C.sizeof(N.DistConfig), C.sizeof(N.DistXfer), C.sizeof(N.DistPlan), C.sizeof(N.DistCallbacks),
            N.DistPlan.owned_tiles.offset, N.DistConfig.transport.offset,
            N.SceneDesc.group_hitmiss.offset, N.FrameParams.image_plane_width.offset,
            N.FrameParams.bounce_count.offset, N.HostView.which_material.offset]
    assert [int(x) for x in out] == want


def test_frame_params_defaults_are_the_shader_constants(pkg):
    p = pkg._native.FrameParams()
    pkg._native.load_hip().shray_frame_params_init(C.byref(p))
    assert p.struct_size == C.sizeof(pkg._native.FrameParams)
    # raytracer.es.fs:550, :381, :382, :445, :525; ray.cpp:474; ray.cpp:46
    assert (p.bounce_count, p.max_bvh_iterations, p.max_leaf_tests, p.cast_shadows, p.tonemap, p.normals_fp16, p.which) == \
        (3, 400, 10, 1, 1, 1, 0)
    assert np.array_equal(np.array(p.camera_matrix[:]).reshape(4, 4), np.eye(4))


def test_tile_buffer_bytes(pkg):
    N = pkg._native
    from shader_ray_amd.tracer import tile_buffer_bytes
    assert tile_buffer_bytes(1920, 1080, None) == 1920 * 1080 * 16
    # 1920x1080 in 32x32 tiles: 60 x 34 = 2040 tiles; 8 ranks -> 255 each
    for phase in range(8):
        assert tile_buffer_bytes(1920, 1080, N.TileSet(32, 32, 8, phase)) == 255 * 32 * 32 * 16
    # 7 tiles over 3 ranks: 3, 2, 2
    sizes = [tile_buffer_bytes(7 * 16, 16, N.TileSet(16, 16, 3, ph)) // (16 * 16 * 16) for ph in range(3)]
    assert sizes == [3, 2, 2]
    # uneven shares: phases [0, 2) and [2, 5) of a period of 5 over 13 tiles -> 2+2+2 = 6 and 3+3+1 = 7 tiles
    sizes = [tile_buffer_bytes(13 * 16, 16, N.TileSet(16, 16, 5, ph, cnt)) // (16 * 16 * 16) for ph, cnt in ((0, 2), (2, 3))]
    assert sizes == [6, 7]
    assert tile_buffer_bytes(64, 64, N.TileSet(16, 16, 5, 3, 3)) == 0     # phase + count beyond the period


def test_argument_errors_are_reported_not_fatal(pkg):
    N = pkg._native
    lib = N.load_hip()
    handle = C.c_void_p()
    d = N.SceneDesc()
    assert lib.shray_scene_create(C.byref(d), C.byref(handle)) == -1       # struct_size 0
    assert b"struct_size" in lib.shray_last_error()
    assert lib.shray_scene_create(None, C.byref(handle)) == -1
    assert lib.shray_scene_destroy(None) == 0
    assert lib.shray_render(None, None, 4, 4, 1, None) == -1


def test_without_a_gpu_the_product_path_fails_loudly(pkg):
    """No CPU fallback: on a machine without a HIP device scene creation returns an error."""
    import helpers
    N = pkg._native
    lib = N.load_hip()
    n = C.c_int()
    if lib.shray_device_count(C.byref(n)) == 0 and n.value > 0:
        pytest.skip("a GPU is present")
    scene = helpers.single_leaf_scene([[[0, 0, 0], [1, 0, 0], [0, 1, 0]]])
    with pytest.raises(N.ShrayError) as err:
        pkg.Scene(scene.desc)
    assert err.value.code in (-2, -3)
