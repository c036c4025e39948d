"""Python face of the HIP layer (include/shader_ray_hip.h): upload a flattened scene and
an environment, render frames.  Device memory and streams are plumbing (torch tensors
/ the current torch stream when torch is used); every pixel is produced by the gfx950
kernels behind the C ABI.  There is no CPU fallback."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _native as N

KERNEL_STACK = 0      # per-ray LDS stack over the packed BVH (default)
KERNEL_THREADED = 1   # literal hit/miss-table traversal over the reference arrays


class Scene:
    """One scene resident on one GPU (replaces the reference's texture upload,
    ray.cpp:470-510)."""

    def __init__(self, desc: N.SceneDesc, environment: np.ndarray | None = None, device: int | None = None):
        self._lib = N.load_hip()
        if device is not None:
            N.check(self._lib.shray_set_device(device))
        handle = C.c_void_p()
        N.check(self._lib.shray_scene_create(C.byref(desc), C.byref(handle)))
        self._handle = handle
        if environment is not None:
            self.set_environment(environment)

    @classmethod
    def from_device(cls, tree_handle, flat_handle, environment: np.ndarray | None = None):
        """A scene from a tree that never left the device (shray_scene_create_from_device): `tree_handle` from
        shray_bvh_build_device, `flat_handle` from shray_flatten_device_tree (both may be destroyed afterwards)."""
        self = cls.__new__(cls)
        self._lib = N.load_hip()
        handle = C.c_void_p()
        N.check(self._lib.shray_scene_create_from_device(tree_handle, flat_handle, C.byref(handle)))
        self._handle = handle
        if environment is not None:
            self.set_environment(environment)
        return self

    def derived_arrays(self) -> dict:
        """What scene creation derived, read back (tests): packed_nodes uint32 [8, nodes, 8], packed_tris uint32 [triangles, 9],
        normals16 uint16 [corners * 3], pair_nodes uint32 [nodes, 16], stack_levels."""
        a, b, c, d, levels = C.c_uint64(), C.c_uint64(), C.c_uint64(), C.c_uint64(), C.c_int32()
        N.check(self._lib.shray_scene_derived_sizes(self._handle, C.byref(a), C.byref(b), C.byref(c), C.byref(d), C.byref(levels)))
        nodes, tris = np.zeros(a.value // 4, np.uint32), np.zeros(b.value // 4, np.uint32)
        halves, pairs = np.zeros(c.value // 2, np.uint16), np.zeros(d.value // 4, np.uint32)
        N.check(self._lib.shray_scene_derived_download(self._handle, nodes.ctypes.data_as(C.c_void_p), tris.ctypes.data_as(C.c_void_p),
                                                       halves.ctypes.data_as(C.c_void_p), pairs.ctypes.data_as(C.c_void_p)))
        return {"packed_nodes": nodes.reshape(8, -1, 8), "packed_tris": tris.reshape(-1, 9), "normals16": halves,
                "pair_nodes": pairs.reshape(-1, 16), "stack_levels": levels.value}

    def close(self):
        if getattr(self, "_handle", None):
            self._lib.shray_scene_destroy(self._handle)
            self._handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_environment(self, rgb: np.ndarray, storage: int = N.ENV_FLOAT32):
        """`rgb` is [height, width, 3] float32, row 0 = straight down (texture t = 0).  storage = ENV_UNORM8 keeps
        it the way most drivers keep the reference's unsized GL_RGB upload (ray.cpp:508): 8 bits, clamped to [0, 1]."""
        rgb = np.ascontiguousarray(rgb, dtype=np.float32)
        h, w, c = rgb.shape
        assert c == 3
        N.check(self._lib.shray_scene_set_environment_storage(self._handle, rgb.ctypes.data_as(N.c_float_p), w, h, storage))

    def set_kernel(self, kernel_id: int):
        N.check(self._lib.shray_scene_set_kernel(self._handle, kernel_id))

    def render(self, params: N.FrameParams, width: int, height: int, spp: int = 1, out: np.ndarray | None = None) -> np.ndarray:
        """Blocking render to host memory: RGBA float32 [height, width, 4], row 0 = bottom.  `out`: a C-contiguous
        float32 array of that shape to fill (a frame loop reuses one: a fresh 33 MB array costs its page faults)."""
        if out is None:
            out = np.empty((height, width, 4), dtype=np.float32)
        elif out.shape != (height, width, 4) or out.dtype != np.float32 or not out.flags.c_contiguous:
            raise ValueError("out must be a C-contiguous float32 array of shape (height, width, 4)")
        N.check(self._lib.shray_render(self._handle, C.byref(params), width, height, spp,
                                       out.ctypes.data_as(N.c_float_p)))
        return out

    def render_to_pinned(self, params: N.FrameParams, width: int, height: int, spp: int, pinned: "PinnedFrame",
                         stream_ptr: int = 0, wait: bool = True):
        """Render + DMA into pinned host memory on a HIP stream (shray_render_host_async).  With wait=False
        the caller synchronises the stream before reading `pinned.array`."""
        N.check(self._lib.shray_render_host_async(self._handle, C.byref(params), width, height, spp,
                                                  C.c_void_p(pinned.ptr), C.c_void_p(stream_ptr)))
        if wait:
            import torch
            torch.cuda.synchronize()
        return pinned.array

    def render_counters(self, params: N.FrameParams, width: int, height: int, spp: int = 1, want_image: bool = True):
        out = np.empty((height, width, 4), dtype=np.float32) if want_image else None
        counters = N.Counters()
        N.check(self._lib.shray_render_counters(
            self._handle, C.byref(params), width, height, spp,
            out.ctypes.data_as(N.c_float_p) if want_image else None, C.byref(counters)))
        return out, counters.as_dict()

    def render_counters_timed(self, params: N.FrameParams, width: int, height: int, spp: int = 1, frames_per_launch: int = 1,
                              want_image: bool = True):
        """Tallies of the instance the timed launches run (shadow rays stop at their first hit, samples in neighbouring
        lanes): shray_render_counters_timed."""
        out = np.empty((height, width, 4), dtype=np.float32) if want_image else None
        counters = N.Counters()
        N.check(self._lib.shray_render_counters_timed(
            self._handle, C.byref(params), width, height, spp, frames_per_launch,
            out.ctypes.data_as(N.c_float_p) if want_image else None, C.byref(counters)))
        return out, counters.as_dict()

    def dispatch_order(self) -> np.ndarray:
        """The patch permutation the next batch launch of the current shape would read (shray_scene_dispatch_order);
        empty while the identity is in use."""
        n = C.c_uint32(0)
        N.check(self._lib.shray_scene_dispatch_order(self._handle, None, 0, C.byref(n)))
        out = np.empty(n.value, dtype=np.uint32)
        if n.value:
            N.check(self._lib.shray_scene_dispatch_order(self._handle, out.ctypes.data_as(C.POINTER(C.c_uint32)), n.value, C.byref(n)))
        return out

    def render_into(self, params: N.FrameParams, width: int, height: int, spp: int, out_ptr: int,
                    stream_ptr: int = 0, tiles: N.TileSet | None = None):
        """Asynchronous render into device memory (`out_ptr`, e.g. tensor.data_ptr()) on a
        HIP stream (`stream_ptr`, e.g. torch.cuda.current_stream().cuda_stream)."""
        N.check(self._lib.shray_render_device(
            self._handle, C.byref(params), width, height, spp,
            C.byref(tiles) if tiles is not None else None, C.c_void_p(out_ptr), C.c_void_p(stream_ptr)))

    def render_batch_into(self, params_list, width: int, height: int, spp: int, out_ptr: int, frame_stride_bytes: int,
                          stream_ptr: int = 0, tiles: N.TileSet | None = None):
        """`len(params_list)` frames in one launch; frame k goes to out_ptr + k * frame_stride_bytes
        (shray_render_batch_device)."""
        count = len(params_list)
        array = (N.FrameParams * count)(*params_list)
        N.check(self._lib.shray_render_batch_device(
            self._handle, array, count, width, height, spp,
            C.byref(tiles) if tiles is not None else None, C.c_void_p(out_ptr), frame_stride_bytes,
            C.c_void_p(stream_ptr)))


class DeviceFlat:
    """get_shader_data on the GPU (shray_flatten_device): the flattened arrays of a host-built BVH, resident
    on the device.  `download()` gives a SceneDesc with host pointers (owned by this object)."""

    def __init__(self, tree: N.TreeDesc, data_texture_width: int = 2048):
        self._lib = N.load_hip()
        self._tree = tree
        handle = C.c_void_p()
        N.check(self._lib.shray_flatten_device(C.byref(tree), data_texture_width, C.byref(handle)))
        self._handle = handle

    def download(self) -> N.SceneDesc:
        desc = N.SceneDesc()
        N.check(self._lib.shray_device_flat_download(self._handle, C.byref(desc)))
        desc._owner = self
        return desc

    def describe(self) -> N.SceneDesc:
        desc = N.SceneDesc()
        N.check(self._lib.shray_device_flat_describe(self._handle, C.byref(desc)))
        desc._owner = self
        return desc

    def arrays(self) -> dict:
        from .host import desc_arrays
        return desc_arrays(self.download())

    def close(self):
        if getattr(self, "_handle", None):
            self._lib.shray_device_flat_destroy(self._handle)
            self._handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class DeviceWorld:
    """File -> resident scene with the BVH, the flattening and everything scene creation derives done ON THE DEVICE (round 6):
    the host parses the file (libshray_host: load_triangles -- world.cpp:46-134 without make_bvh); shray_bvh_build_device,
    shray_flatten_device_tree and shray_scene_create_from_device do the rest where the data lies.  No tree is downloaded and no
    group tree is built on the host unless somebody asks for one (`host_world()`).  `seconds`: the stages' wall times."""

    def __init__(self, filename: str, environment: np.ndarray | None = None, options: "N.BvhOptions | None" = None, device: int | None = None,
                 data_texture_width: int = 2048, quiet: bool = True):
        import time
        from . import host
        hip, lib = N.load_hip(), N.load_host()
        lib.shray_host_set_quiet(1 if quiet else 0)
        if device is not None:
            N.check(hip.shray_set_device(device))
        self._hip, self._host, self.filename = hip, lib, filename
        self.seconds = {}
        t0 = time.perf_counter()
        handle = C.c_void_p()
        if lib.shray_host_load_triangles(filename.encode(), C.byref(handle)) != 0 or not handle:
            raise RuntimeError(f"load_triangles failed for {filename!r} (see stderr)")
        self._world_handle = handle
        tv, vd, nt, nv = C.POINTER(C.c_int32)(), C.POINTER(C.c_float)(), C.c_int32(), C.c_int32()
        if lib.shray_host_triangles(handle, C.byref(tv), C.byref(nt), C.byref(vd), C.byref(nv)) != 0:
            raise RuntimeError("shray_host_triangles failed")
        t1 = time.perf_counter()
        if options is None:
            options = host.bvh_options_from_environment()
        options.struct_size = C.sizeof(N.BvhOptions)
        self._options = options
        self._tree = C.c_void_p()
        N.check(hip.shray_bvh_build_device(tv, nt, vd, nv, 9, C.byref(options), C.byref(self._tree)))
        t2 = time.perf_counter()
        self._flat = C.c_void_p()
        N.check(hip.shray_flatten_device_tree(self._tree, data_texture_width, C.byref(self._flat)))
        t3 = time.perf_counter()
        self.scene = Scene.from_device(self._tree, self._flat, None)
        t4 = time.perf_counter()
        if environment is not None:
            self.scene.set_environment(environment)
        stats = N.BvhStats()
        N.check(hip.shray_device_tree_stats(self._tree, C.byref(stats)))
        self.stats = stats
        self.triangle_count = nt.value
        self.seconds = {"parse": t1 - t0, "bvh": t2 - t1, "bvh_on_the_device": stats.device_seconds, "flatten": t3 - t2, "scene": t4 - t3,
                        "triangles_to_resident": t4 - t1, "total": t4 - t0}
        self._adopted = False

    def frame_params(self, width: int, height: int, view=None, material: int | None = None, diffuse: int | None = None):
        """The frame block (ray.cpp:648-704) -- it needs the mesh's extent, not its tree."""
        from . import host
        return host.frame_params_of(self._world_handle, width, height, view, material, diffuse)

    def default_view(self):
        from . import host
        return host.default_view_of(self._world_handle)

    def flat_arrays(self) -> dict:
        """The flattened (reference-layout) arrays, downloaded (tests)."""
        from .host import desc_arrays
        desc = N.SceneDesc()
        N.check(self._hip.shray_device_flat_download(self._flat, C.byref(desc)))
        return desc_arrays(desc)

    def host_world(self):
        """The reference's `world` with its group tree (world.h:48-51), built NOW from the device's tree: shray_device_tree_download +
        shray_host_adopt_tree.  Returns the libshray_host world handle (owned by this object)."""
        if not self._adopted:
            tree, order = N.TreeDesc(), C.POINTER(C.c_int32)()
            N.check(self._hip.shray_device_tree_download(self._tree, C.byref(tree), C.byref(order)))
            if self._host.shray_host_adopt_tree(self._world_handle, C.byref(tree), order, float(self.seconds.get("bvh", 0.0))) != 0:
                raise RuntimeError("shray_host_adopt_tree refused the device-built tree")
            self._adopted = True
        return self._world_handle

    def close(self):
        if getattr(self, "scene", None) is not None:
            self.scene.close()
            self.scene = None
        for name, destroy in (("_flat", "shray_device_flat_destroy"), ("_tree", "shray_device_tree_destroy")):
            h = getattr(self, name, None)
            if h:
                getattr(self._hip, destroy)(h)
                setattr(self, name, None)
        if getattr(self, "_world_handle", None):
            self._host.shray_host_free_world(self._world_handle)
            self._world_handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class PinnedFrame:
    """RGBA float32 [height, width, 4] in pinned host memory (shray_pinned_alloc): the destination of the
    PCIe-speed readback forms."""

    def __init__(self, width: int, height: int):
        self._lib = N.load_hip()
        p = C.c_void_p()
        N.check(self._lib.shray_pinned_alloc(width * height * 16, C.byref(p)))
        self.ptr = p.value
        self.array = np.ctypeslib.as_array((C.c_float * (width * height * 4)).from_address(self.ptr)).reshape(height, width, 4)

    def close(self):
        if getattr(self, "ptr", None):
            self.array = None
            self._lib.shray_pinned_free(C.c_void_p(self.ptr))
            self.ptr = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def tile_buffer_bytes(width: int, height: int, tiles: N.TileSet | None) -> int:
    return int(N.load_hip().shray_tile_buffer_bytes(width, height, C.byref(tiles) if tiles is not None else None))


def algorithmic_bytes(counters: dict, pixels: int, normals_fp16: bool = True, out_bytes: int = 16) -> int:
    """Cache-less byte count of the reference's own fetches:
    32 B per node visit (24 B box + 8 B links), +8 B per leaf visit (start, count),
    36 B per triangle test (3 x 12 B), 18 B per shaded hit (3 fp16 normals; 36 B if
    fp32), 48 B per environment lookup (4 texels x 12 B), 16 B per output pixel."""
    c = counters
    return (32 * c["node_visits"] + 8 * c["leaf_visits"] + 36 * c["triangle_tests"]
            + (18 if normals_fp16 else 36) * c["shaded_hits"] + 48 * c["env_lookups"] + out_bytes * pixels)
