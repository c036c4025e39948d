"""Python face of the host layer: load a scene file, build the BVH, flatten it, and
produce frame parameters -- the same steps the reference's `main` performs before its
first draw call (reference ray.cpp:954-1092), minus the window."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _native as N


class World:
    """A loaded scene (`load_world`, reference world.cpp:46) plus its flattened arrays."""

    def __init__(self, filename: str, quiet: bool = True, build: str = "host", options: N.BvhOptions | None = None):
        """build = "host": make_bvh on the host (bvh.cpp:288-358; sub-trees on the box's threads); "gpu": the same tree built by
        shray_bvh_build_device (csrc/bvh_build.hip) -- load_triangles, the device build, adopt_tree -- bit-identical arrays."""
        lib = N.load_host()
        lib.shray_host_set_quiet(1 if quiet else 0)
        handle = C.c_void_p()
        self._lib = lib
        self._desc = None
        self.bvh_device_seconds = None
        if build == "host":
            if lib.shray_host_load_world(filename.encode(), C.byref(handle)) != 0 or not handle:
                raise RuntimeError(f"load_world failed for {filename!r} (see stderr)")
            self._handle = handle
        elif build == "gpu":
            import time
            if lib.shray_host_load_triangles(filename.encode(), C.byref(handle)) != 0 or not handle:
                raise RuntimeError(f"load_triangles failed for {filename!r} (see stderr)")
            self._handle = handle
            hip = N.load_hip()
            tv, vd = C.POINTER(C.c_int32)(), C.POINTER(C.c_float)()
            nt, nv = C.c_int32(), C.c_int32()
            if lib.shray_host_triangles(handle, C.byref(tv), C.byref(nt), C.byref(vd), C.byref(nv)) != 0:
                raise RuntimeError("shray_host_triangles failed")
            then = time.perf_counter()
            tree_handle = C.c_void_p()
            if options is None:
                # the host builder reads BVH_MAX_DEPTH / BVH_LEAF_MAX / SAH_* from the environment (bvh.cpp:60-79): so does this one
                # (with the defaults the "bit-identical" device tree silently parted from the host's)
                options = bvh_options_from_environment()
            options.struct_size = C.sizeof(N.BvhOptions)
            N.check(hip.shray_bvh_build_device(tv, nt, vd, nv, 9, C.byref(options), C.byref(tree_handle)))
            try:
                tree, order = N.TreeDesc(), C.POINTER(C.c_int32)()
                N.check(hip.shray_device_tree_download(tree_handle, C.byref(tree), C.byref(order)))
                stats = N.BvhStats()
                N.check(hip.shray_device_tree_stats(tree_handle, C.byref(stats)))
                self.bvh_device_seconds = stats.device_seconds
                if lib.shray_host_adopt_tree(handle, C.byref(tree), order, time.perf_counter() - then) != 0:
                    raise RuntimeError("shray_host_adopt_tree refused the device-built tree")
            finally:
                hip.shray_device_tree_destroy(tree_handle)
        else:
            raise ValueError(f"build = {build!r}: 'host' or 'gpu'")
        info = N.HostWorldInfo()
        lib.shray_host_get_world_info(self._handle, C.byref(info))
        self.info = info

    def close(self):
        if self._handle:
            self._lib.shray_host_free_world(self._handle)
            self._handle = None
            self._desc = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def triangle_count(self) -> int:
        return self.info.triangle_count

    def flatten(self, data_texture_width: int = 2048) -> N.SceneDesc:
        """`get_shader_data` (reference world.cpp:298).  The returned descriptor points
        into memory owned by this World."""
        if self._desc is not None and self._desc.data_texture_width == data_texture_width:
            return self._desc
        desc = N.SceneDesc()
        if self._lib.shray_host_flatten(self._handle, data_texture_width, C.byref(desc)) != 0:
            raise RuntimeError("get_shader_data failed (tree deeper than the 64-entry link stack?)")
        desc._owner = self   # the arrays live in the C-side world: keep it alive as long as the descriptor
        self._desc = desc
        return desc

    def export_tree(self) -> N.TreeDesc:
        """The BVH as pre-order arrays (shray_host_export_tree): input of the GPU flattener.  The arrays
        are owned by this World."""
        tree = N.TreeDesc()
        if self._lib.shray_host_export_tree(self._handle, C.byref(tree)) != 0:
            raise RuntimeError("shray_host_export_tree failed")
        tree._owner = self
        return tree

    def arrays(self, data_texture_width: int = 2048) -> dict:
        """The flattened arrays as numpy copies (populated prefix only), keyed like
        scene_shader_data (reference world.h:68-93)."""
        d = self._desc if self._desc is not None else self.flatten(data_texture_width)
        return desc_arrays(d)


    def default_view(self) -> N.HostView:
        return default_view_of(self._handle)

    def frame_params(self, width: int, height: int, view: N.HostView | None = None, *, material: int | None = None,
                     diffuse: int | None = None) -> N.FrameParams:
        """Frame block for a width x height frame (reference ray.cpp:648-704).  `material`
        indexes the reference's table: 0 = gold ... 6 = glazed plaster (ray.cpp:54-65)."""
        return frame_params_of(self._handle, width, height, view, material, diffuse)


def default_view_of(world_handle) -> N.HostView:
    """The start-up view (ray.cpp:1077-1088) of a libshray_host world -- loaded with or without its tree."""
    view = N.HostView()
    N.load_host().shray_host_default_view(world_handle, C.byref(view))
    return view


def frame_params_of(world_handle, width: int, height: int, view: N.HostView | None = None, material: int | None = None,
                    diffuse: int | None = None) -> N.FrameParams:
    if view is None:
        view = default_view_of(world_handle)
    if material is not None:
        view.which_material = material
    if diffuse is not None:
        view.which_diffuse_color = diffuse
    params = N.FrameParams()
    if N.load_host().shray_host_frame_params(world_handle, C.byref(view), width, height, C.byref(params)) != 0:
        raise RuntimeError("frame parameter computation failed")
    return params


def bvh_options_from_environment() -> N.BvhOptions:
    """BVH_MAX_DEPTH / BVH_LEAF_MAX / SAH_CTRAV / SAH_CISEC as the host builder of this process reads them (bvh.cpp:60-79)."""
    options = N.BvhOptions()
    if N.load_host().shray_host_bvh_options(C.byref(options)) != 0:
        raise RuntimeError("shray_host_bvh_options failed")
    return options


def desc_arrays(d) -> dict:
    """numpy copies of the arrays a SceneDesc with HOST pointers names (populated prefix only)."""
    nv, ng = d.vertex_count, d.group_count
    stride = d.data_texture_width * d.group_data_rows

    def take(ptr, n):
        return np.ctypeslib.as_array(ptr, shape=(n,)).copy() if n else np.zeros(0, np.float32)

    out = {
        "vertex_count": nv, "vertex_data_rows": d.vertex_data_rows, "group_count": ng,
        "group_data_rows": d.group_data_rows, "tree_root": d.tree_root,
        "vertex_positions": take(d.vertex_positions, 3 * nv), "vertex_normals": take(d.vertex_normals, 3 * nv),
        "vertex_colors": take(d.vertex_colors, 3 * nv),
        "group_boxmin": take(d.group_boxmin, 3 * ng), "group_boxmax": take(d.group_boxmax, 3 * ng),
        "group_directions": take(d.group_directions, 3 * ng), "group_children": take(d.group_children, 2 * ng),
        "group_objects": take(d.group_objects, 2 * ng),
    }
    hm = np.ctypeslib.as_array(d.group_hitmiss, shape=(8, stride, 2)) if stride else np.zeros((8, 0, 2), np.float32)
    for code in range(8):
        out[f"group_hitmiss_{code}"] = hm[code, :ng].reshape(-1).copy()
    return out


def trackball_motion(rotation, dx: float, dy: float):
    """ray.cpp:91-98: `rotation` (a 4-float ctypes array of a HostView) after a drag by (dx, dy), in place."""
    N.load_host().shray_host_trackball_motion(rotation, dx, dy, rotation)
    return rotation


def load_background(spec: str) -> np.ndarray:
    """The reference's background argument (ray.cpp:1002-1075): "r, g, b", "grid", "rrggbb" or a
    Radiance .hdr file -> float32 [height, width, 3], row 0 = bottom row of the picture."""
    lib = N.load_host()
    w, h = C.c_int(), C.c_int()
    px = N.c_float_p()
    if lib.shray_host_load_background(spec.encode(), C.byref(w), C.byref(h), C.byref(px)) != 0:
        raise RuntimeError(f"cannot load background {spec!r} (see stderr)")
    try:
        return np.ctypeslib.as_array(px, shape=(h.value, w.value, 3)).copy()
    finally:
        lib.shray_host_free_background(px)
