"""ctypes mirror of include/shader_ray_hip.h and include/shader_ray_host.h.

Plumbing only: structure layouts, library loading, error translation.  There is no
CPU fallback -- if a shared library is missing the loaders raise, loudly.
"""
from __future__ import annotations

import ctypes as C
import os

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
# SHRAY_HOST_LIB selects another build of the host layer (the sanitizer build, `make -C shader-ray_amd sanitize`: tests/test_sanitizers.py)
HOST_LIB = os.environ.get("SHRAY_HOST_LIB") or os.path.join(PKG_DIR, "libshray_host.so")
DIST_LIB = os.path.join(PKG_DIR, "libshray_dist.so")
# SHRAY_HIP_LIB selects an experiment build of the same library (profiles/variant_sweep.sh); unset in normal use
HIP_LIB = os.environ.get("SHRAY_HIP_LIB") or os.path.join(PKG_DIR, "libshray_hip.so")

c_float_p = C.POINTER(C.c_float)
ENV_FLOAT32, ENV_UNORM8 = 0, 1
ABI_VERSION = 4   # SHRAY_ABI_VERSION of the header these structures mirror (tests/test_abi.py compares the two)


class SceneDesc(C.Structure):
    _fields_ = [
        ("struct_size", C.c_uint32), ("data_texture_width", C.c_uint32),
        ("vertex_count", C.c_uint32), ("vertex_data_rows", C.c_uint32),
        ("vertex_positions", c_float_p), ("vertex_normals", c_float_p), ("vertex_colors", c_float_p),
        ("group_count", C.c_int32), ("group_data_rows", C.c_int32), ("tree_root", C.c_int32),
        ("group_boxmin", c_float_p), ("group_boxmax", c_float_p),
        ("group_directions", c_float_p), ("group_children", c_float_p),
        ("group_hitmiss", c_float_p), ("group_objects", c_float_p),
    ]


class FrameParams(C.Structure):
    _fields_ = [
        ("struct_size", C.c_uint32), ("which", C.c_int32),
        ("camera_matrix", C.c_float * 16), ("camera_normal_matrix", C.c_float * 16),
        ("object_matrix", C.c_float * 16), ("object_inverse", C.c_float * 16),
        ("object_normal_matrix", C.c_float * 16), ("object_normal_inverse", C.c_float * 16),
        ("image_plane_width", C.c_float), ("aspect", C.c_float),
        ("right", C.c_float * 3), ("up", C.c_float * 3), ("light_dir", C.c_float * 3),
        ("specular_color", C.c_float * 3), ("diffuse_color", C.c_float * 3),
        ("bounce_count", C.c_int32), ("max_bvh_iterations", C.c_int32), ("max_leaf_tests", C.c_int32),
        ("cast_shadows", C.c_int32), ("tonemap", C.c_int32), ("normals_fp16", C.c_int32),
    ]

    def copy(self) -> "FrameParams":
        other = FrameParams()
        C.memmove(C.byref(other), C.byref(self), C.sizeof(FrameParams))
        return other


class TileSet(C.Structure):
    _fields_ = [("tile_w", C.c_int32), ("tile_h", C.c_int32), ("tile_stride", C.c_int32), ("tile_phase", C.c_int32),
                ("tile_phase_count", C.c_int32)]


class Counters(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in (
        "node_visits", "leaf_visits", "triangle_tests", "shaded_hits", "env_lookups",
        "traversals", "bad_hits", "samples")]

    def as_dict(self) -> dict:
        return {n: int(getattr(self, n)) for n, _ in self._fields_}


class TreeDesc(C.Structure):
    """shray_tree_desc (include/shader_ray_hip.h): the BVH as pre-order arrays, input of the GPU flattener."""
    _fields_ = [
        ("struct_size", C.c_uint32), ("node_count", C.c_int32),
        ("node_parent", C.POINTER(C.c_int32)), ("node_negative", C.POINTER(C.c_int32)), ("node_positive", C.POINTER(C.c_int32)),
        ("node_box", C.POINTER(C.c_float)), ("node_direction", C.POINTER(C.c_float)),
        ("node_start", C.POINTER(C.c_int32)), ("node_triangles", C.POINTER(C.c_int32)),
        ("triangle_count", C.c_int32), ("triangle_vertices", C.POINTER(C.c_int32)),
        ("vertex_count", C.c_int32), ("vertex_data", C.POINTER(C.c_float)),
    ]


class HostView(C.Structure):
    _fields_ = [
        ("fov", C.c_float), ("zoom", C.c_float), ("object_rotation", C.c_float * 4),
        ("object_position", C.c_float * 3), ("light_rotation", C.c_float * 4),
        ("which", C.c_int32), ("which_material", C.c_int32), ("which_diffuse_color", C.c_int32),
    ]


class HostWorldInfo(C.Structure):
    _fields_ = [
        ("triangle_count", C.c_int32), ("independent_vertex_count", C.c_int32),
        ("scene_center", C.c_float * 3), ("scene_extent", C.c_float),
        ("node_count", C.c_int32), ("leaf_count", C.c_int32), ("max_level", C.c_int32),
        ("large_leaves", C.c_int32), ("parse_seconds", C.c_double), ("build_seconds", C.c_double),
    ]


class BvhOptions(C.Structure):
    """shray_bvh_options (include/shader_ray_hip.h): the reference's build parameters (bvh.cpp:28-58)."""
    _fields_ = [("struct_size", C.c_uint32), ("max_depth", C.c_int32), ("leaf_max", C.c_int32), ("sah_ctrav", C.c_float), ("sah_cisec", C.c_float)]


class BvhStats(C.Structure):
    _fields_ = [("node_count", C.c_int32), ("leaf_count", C.c_int32), ("max_level", C.c_int32), ("large_leaves", C.c_int32),
                ("device_seconds", C.c_double)]


# Every symbol include/shader_ray_hip.h declares: (name, restype, argtypes)
HIP_SYMBOLS = [
    ("shray_abi_version", C.c_int, []),
    ("shray_last_error", C.c_char_p, []),
    ("shray_device_count", C.c_int, [C.POINTER(C.c_int)]),
    ("shray_set_device", C.c_int, [C.c_int]),
    ("shray_frame_params_init", None, [C.POINTER(FrameParams)]),
    ("shray_scene_create", C.c_int, [C.POINTER(SceneDesc), C.POINTER(C.c_void_p)]),
    ("shray_scene_set_environment", C.c_int, [C.c_void_p, c_float_p, C.c_int, C.c_int]),
    ("shray_scene_set_environment_storage", C.c_int, [C.c_void_p, c_float_p, C.c_int, C.c_int, C.c_int]),
    ("shray_scene_destroy", C.c_int, [C.c_void_p]),
    ("shray_scene_device", C.c_int, [C.c_void_p, C.POINTER(C.c_int)]),
    ("shray_scene_set_kernel", C.c_int, [C.c_void_p, C.c_int]),
    ("shray_render", C.c_int, [C.c_void_p, C.POINTER(FrameParams), C.c_int, C.c_int, C.c_int, c_float_p]),
    ("shray_render_host_async", C.c_int, [C.c_void_p, C.POINTER(FrameParams), C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    ("shray_pinned_alloc", C.c_int, [C.c_size_t, C.POINTER(C.c_void_p)]),
    ("shray_pinned_free", C.c_int, [C.c_void_p]),
    ("shray_flatten_device", C.c_int, [C.POINTER(TreeDesc), C.c_uint32, C.POINTER(C.c_void_p)]),
    ("shray_device_flat_describe", C.c_int, [C.c_void_p, C.POINTER(SceneDesc)]),
    ("shray_device_flat_download", C.c_int, [C.c_void_p, C.POINTER(SceneDesc)]),
    ("shray_device_flat_destroy", C.c_int, [C.c_void_p]),
    ("shray_bvh_build_device", C.c_int, [C.POINTER(C.c_int32), C.c_int32, C.POINTER(C.c_float), C.c_int32, C.c_int32, C.POINTER(BvhOptions),
                                         C.POINTER(C.c_void_p)]),
    ("shray_device_tree_download", C.c_int, [C.c_void_p, C.POINTER(TreeDesc), C.POINTER(C.POINTER(C.c_int32))]),
    ("shray_device_tree_stats", C.c_int, [C.c_void_p, C.POINTER(BvhStats)]),
    ("shray_device_tree_destroy", C.c_int, [C.c_void_p]),
    ("shray_flatten_device_tree", C.c_int, [C.c_void_p, C.c_uint32, C.POINTER(C.c_void_p)]),
    ("shray_scene_create_from_device", C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(C.c_void_p)]),
    ("shray_scene_derived_sizes", C.c_int, [C.c_void_p, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64),
                                            C.POINTER(C.c_uint64), C.POINTER(C.c_int32)]),
    ("shray_scene_derived_download", C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    ("shray_render_device", C.c_int, [C.c_void_p, C.POINTER(FrameParams), C.c_int, C.c_int, C.c_int,
                                      C.POINTER(TileSet), C.c_void_p, C.c_void_p]),
    ("shray_render_batch_device", C.c_int, [C.c_void_p, C.POINTER(FrameParams), C.c_int, C.c_int, C.c_int, C.c_int,
                                            C.POINTER(TileSet), C.c_void_p, C.c_int64, C.c_void_p]),
    ("shray_tile_buffer_bytes", C.c_int64, [C.c_int, C.c_int, C.POINTER(TileSet)]),
    ("shray_assemble_tiles_device", C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int64, C.c_int64,
                                              C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    ("shray_assemble_tiles_split_device", C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int64, C.c_int64,
                                                    C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    ("shray_render_counters", C.c_int, [C.c_void_p, C.POINTER(FrameParams), C.c_int, C.c_int, C.c_int,
                                        c_float_p, C.POINTER(Counters)]),
    ("shray_render_counters_timed", C.c_int, [C.c_void_p, C.POINTER(FrameParams), C.c_int, C.c_int, C.c_int, C.c_int,
                                              c_float_p, C.POINTER(Counters)]),
    ("shray_scene_dispatch_order", C.c_int, [C.c_void_p, C.POINTER(C.c_uint32), C.c_uint32, C.POINTER(C.c_uint32)]),
    ("shray_selftest_division", C.c_int, [C.c_uint64, C.c_uint64, C.POINTER(C.c_uint64)]),
    ("shray_selftest_reciprocal", C.c_int, [C.POINTER(C.c_uint64)]),
    ("shray_probe_vector_cache", C.c_int, [C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint64, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_uint64)]),
]

HOST_SYMBOLS = [
    ("shray_host_load_world", C.c_int, [C.c_char_p, C.POINTER(C.c_void_p)]),
    ("shray_host_free_world", None, [C.c_void_p]),
    ("shray_host_load_triangles", C.c_int, [C.c_char_p, C.POINTER(C.c_void_p)]),
    ("shray_host_triangles", C.c_int, [C.c_void_p, C.POINTER(C.POINTER(C.c_int32)), C.POINTER(C.c_int32), C.POINTER(C.POINTER(C.c_float)),
                                       C.POINTER(C.c_int32)]),
    ("shray_host_adopt_tree", C.c_int, [C.c_void_p, C.POINTER(TreeDesc), C.POINTER(C.c_int32), C.c_double]),
    ("shray_host_bvh_options", C.c_int, [C.POINTER(BvhOptions)]),
    ("shray_host_get_world_info", C.c_int, [C.c_void_p, C.POINTER(HostWorldInfo)]),
    ("shray_host_flatten", C.c_int, [C.c_void_p, C.c_uint, C.POINTER(SceneDesc)]),
    ("shray_host_export_tree", C.c_int, [C.c_void_p, C.POINTER(TreeDesc)]),
    ("shray_host_default_view", C.c_int, [C.c_void_p, C.POINTER(HostView)]),
    ("shray_host_frame_params", C.c_int, [C.c_void_p, C.POINTER(HostView), C.c_int, C.c_int, C.POINTER(FrameParams)]),
    ("shray_host_trackball_motion", C.c_int, [C.POINTER(C.c_float), C.c_float, C.c_float, C.POINTER(C.c_float)]),
    ("shray_host_load_background", C.c_int, [C.c_char_p, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(c_float_p)]),
    ("shray_host_free_background", None, [c_float_p]),
    ("shray_host_set_quiet", None, [C.c_int]),
]

# include/shader_ray_dist.h ------------------------------------------------------------------------------
DIST_ROOT0, DIST_ROTATE = 0, 1
DIST_RCCL, DIST_LOOPBACK, DIST_CALLBACK = 0, 1, 2
DIST_UNIQUE_ID_BYTES = 128
MAX_BATCH = 64
DIST_MAX_WORLD = 64


class DistConfig(C.Structure):
    _fields_ = [("struct_size", C.c_uint32), ("rank", C.c_int32), ("world", C.c_int32),
                ("width", C.c_int32), ("height", C.c_int32), ("spp", C.c_int32),
                ("tile_w", C.c_int32), ("tile_h", C.c_int32), ("max_frames", C.c_int32), ("root_mode", C.c_int32),
                ("rank0_phases", C.c_int32), ("other_phases", C.c_int32), ("rgb_wire", C.c_int32),
                ("transport", C.c_int32), ("buffer_sets", C.c_int32)]


class DistXfer(C.Structure):
    _fields_ = [("peer", C.c_int32), ("frame", C.c_int32), ("offset_bytes", C.c_int64), ("bytes", C.c_int64)]


class DistPlan(C.Structure):
    _fields_ = [("struct_size", C.c_uint32), ("tiles", TileSet), ("rank0_phases", C.c_int32), ("other_phases", C.c_int32),
                ("channels", C.c_int32), ("max_assembled", C.c_int32), ("owned_tiles", C.c_int64), ("max_tiles", C.c_int64),
                ("render_frame_stride_bytes", C.c_int64), ("wire_frame_stride_bytes", C.c_int64),
                ("gather_rank_stride_bytes", C.c_int64), ("gather_frame_stride_bytes", C.c_int64)]


DIST_EXCHANGE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.POINTER(DistXfer), C.c_int, C.c_void_p, C.POINTER(DistXfer),
                               C.c_int, C.c_void_p)


class DistCallbacks(C.Structure):
    _fields_ = [("user", C.c_void_p), ("exchange", DIST_EXCHANGE_FN)]


DIST_SYMBOLS = [
    ("shray_dist_last_error", C.c_char_p, []),
    ("shray_dist_balanced_shares", C.c_int, [C.c_int, C.c_double, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    ("shray_dist_make_plan", C.c_int, [C.POINTER(DistConfig), C.POINTER(DistPlan)]),
    ("shray_dist_frame_owner", C.c_int, [C.POINTER(DistConfig), C.c_int]),
    ("shray_dist_step_xfers", C.c_int, [C.POINTER(DistConfig), C.c_int, C.POINTER(DistXfer), C.POINTER(C.c_int),
                                        C.POINTER(DistXfer), C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int),
                                        C.POINTER(C.c_int)]),
    ("shray_dist_unique_id", C.c_int, [C.c_void_p]),
    ("shray_dist_hub_create", C.c_int, [C.c_int, C.POINTER(C.c_void_p)]),
    ("shray_dist_hub_destroy", C.c_int, [C.c_void_p]),
    ("shray_dist_create", C.c_int, [C.c_void_p, C.POINTER(DistConfig), C.c_void_p, C.POINTER(C.c_void_p)]),
    ("shray_dist_destroy", C.c_int, [C.c_void_p]),
    ("shray_dist_world", C.c_int, [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    ("shray_dist_step", C.c_int, [C.c_void_p, C.c_int, C.POINTER(FrameParams), C.c_int, C.c_void_p]),
    ("shray_dist_set_timing", C.c_int, [C.c_void_p, C.c_int]),
    ("shray_dist_step_times", C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_float), C.POINTER(C.c_float), C.POINTER(C.c_float)]),
    ("shray_dist_output", C.c_int, [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int),
                                    C.POINTER(C.c_void_p)]),
    ("shray_dist_copy_output", C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    ("shray_dist_copy_to_host", C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]),
    ("shray_dist_copy_to_device", C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]),
]

_host = None
_hip = None
_dist = None


def _bind(lib, table):
    for name, restype, argtypes in table:
        fn = getattr(lib, name)          # AttributeError here = header/library mismatch
        fn.restype = restype
        fn.argtypes = argtypes
    return lib


def load_host():
    global _host
    if _host is None:
        if not os.path.exists(HOST_LIB):
            raise RuntimeError(f"{HOST_LIB} is not built; run `python __graft_entry__.py build` (or `make -C shader-ray_amd`)")
        _host = _bind(C.CDLL(HOST_LIB), HOST_SYMBOLS)
    return _host


def load_hip():
    """Loads the HIP layer.  torch (when installed) is imported first so that both share
    one HIP runtime: torch bundles its own libamdhip64 under the same SONAME."""
    global _hip
    if _hip is None:
        if not os.path.exists(HIP_LIB):
            raise RuntimeError(f"{HIP_LIB} is not built; run `python __graft_entry__.py build` (or `make -C shader-ray_amd`). "
                               "There is no CPU fallback for the tracer.")
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        _hip = _bind(C.CDLL(HIP_LIB), HIP_SYMBOLS)
        if _hip.shray_abi_version() != ABI_VERSION:
            raise RuntimeError(f"libshray_hip.so reports ABI version {_hip.shray_abi_version()}, these bindings mirror "
                               f"version {ABI_VERSION} of include/shader_ray_hip.h: rebuild the library")
    return _hip


def load_dist():
    """Loads the multi-GPU frame loop (libshray_dist.so: depends on libshray_hip.so and RCCL)."""
    global _dist
    if _dist is None:
        load_hip()    # torch first, then the HIP layer: one HIP runtime, one RCCL (torch bundles both)
        if not os.path.exists(DIST_LIB):
            raise RuntimeError(f"{DIST_LIB} is not built; run `python __graft_entry__.py build` (or `make -C shader-ray_amd`)")
        _dist = _bind(C.CDLL(DIST_LIB), DIST_SYMBOLS)
    return _dist


def check_dist(code: int):
    if code != 0:
        msg = load_dist().shray_dist_last_error()
        raise ShrayError(code, msg.decode() if msg else "")


class ShrayError(RuntimeError):
    def __init__(self, code: int, message: str):
        super().__init__(f"shray error {code}: {message}")
        self.code = code


def check(code: int):
    if code != 0:
        msg = load_hip().shray_last_error()
        raise ShrayError(code, msg.decode() if msg else "")
