"""Shared helpers for the test-suite: small scene files, hand-built flattened scenes,
image comparison."""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

from __graft_entry__ import load_package

REL_TOL = 1e-4     # north_star: pixels match the CPU evaluation within 1e-4 relative
ABS_FLOOR = 1e-6   # tone-mapped values live in [0, 1]; below this "relative" is meaningless

END = np.float32(2147483648.0)   # 0x7fffffff stored as float32 (world.cpp:229)


def scene_file(name: str, maker):
    """Generates `name` once per machine under the scene cache and returns its path."""
    pkg = load_package()
    path = pkg.scenes.cached_path(name)
    if not os.path.exists(path):
        tmp = path + ".tmp%d" % os.getpid() + os.path.splitext(path)[1]
        maker(tmp)
        os.replace(tmp, path)
    return path


def small_trisrc():
    pkg = load_package()

    def make(path):
        pos, tri = pkg.scenes.lobed_sphere_mesh(24, 48, bumpiness=0.22, ears=True)
        pkg.scenes.write_trisrc(path, pos, tri)
    return scene_file("small_lobed_24x48.trisrc", make)


def bunny_trisrc():
    pkg = load_package()
    return scene_file("bunny_class_132x264.trisrc", lambda p: pkg.scenes.bunny_class_trisrc(p))


def small_obj_no_normals():
    pkg = load_package()

    def make(path):
        pos, tri = pkg.scenes.lobed_sphere_mesh(40, 64, bumpiness=0.15, ears=False, scale=3.0, center=(5.0, -2.0, 1.0))
        pkg.scenes.write_obj(path, pos, tri)
    return scene_file("small_sphere_40x64.obj", make)


def million_obj():
    pkg = load_package()
    return scene_file("million_501x1000.obj", lambda p: pkg.scenes.million_triangle_obj(p))


class HandScene:
    """A flattened scene assembled by hand (numpy arrays in scene_shader_data layout)."""

    def __init__(self, positions, normals, boxmin, boxmax, hitmiss8, objects, tree_root, width=2048):
        pkg = load_package()
        N = pkg._native
        positions = np.ascontiguousarray(positions, dtype=np.float32).reshape(-1, 3)
        normals = np.ascontiguousarray(normals, dtype=np.float32).reshape(-1, 3)
        nv = len(positions)
        ng = len(boxmin)
        vrows = (nv + width - 1) // width
        grows = (ng + width - 1) // width

        def padded(a, rows, comps):
            buf = np.zeros((max(rows, 1) * width, comps), dtype=np.float32)
            a = np.asarray(a, dtype=np.float32).reshape(-1, comps)
            buf[:len(a)] = a
            return buf

        self.keep = {
            "pos": padded(positions, vrows, 3), "nrm": padded(normals, vrows, 3),
            "bmin": padded(boxmin, grows, 3), "bmax": padded(boxmax, grows, 3),
            "obj": padded(objects, grows, 2),
        }
        hm = np.zeros((8, grows * width, 2), dtype=np.float32)
        hm[:, :ng] = np.asarray(hitmiss8, dtype=np.float32).reshape(8, ng, 2)
        self.keep["hm"] = hm
        d = N.SceneDesc()
        d.struct_size = C.sizeof(N.SceneDesc)
        d.data_texture_width = width
        d.vertex_count = nv
        d.vertex_data_rows = vrows
        d.group_count = ng
        d.group_data_rows = grows
        d.tree_root = tree_root
        fp = N.c_float_p
        d.vertex_positions = self.keep["pos"].ctypes.data_as(fp)
        d.vertex_normals = self.keep["nrm"].ctypes.data_as(fp)
        d.group_boxmin = self.keep["bmin"].ctypes.data_as(fp)
        d.group_boxmax = self.keep["bmax"].ctypes.data_as(fp)
        d.group_hitmiss = self.keep["hm"].ctypes.data_as(fp)
        d.group_objects = self.keep["obj"].ctypes.data_as(fp)
        d._keepalive = self.keep   # the descriptor only holds raw pointers into these arrays
        self.desc = d


def single_leaf_scene(tri_positions, tri_normals=None, count=None):
    """One leaf node holding all the triangles given ([T,3,3] positions)."""
    tp = np.asarray(tri_positions, dtype=np.float32).reshape(-1, 3, 3)
    T = len(tp)
    if tri_normals is None:
        n = np.cross(tp[:, 1] - tp[:, 0], tp[:, 2] - tp[:, 0])
        n /= np.maximum(np.linalg.norm(n, axis=1, keepdims=True), 1e-30)
        tri_normals = np.repeat(n[:, None, :], 3, axis=1)
    pts = tp.reshape(-1, 3)
    lo, hi = pts.min(0) - 1e-5, pts.max(0) + 1e-5
    hm = np.full((8, 1, 2), END, dtype=np.float32)
    return HandScene(pts, np.asarray(tri_normals, np.float32).reshape(-1, 3), [lo], [hi], hm,
                     [[0, T if count is None else count]], 0)


def default_params(pkg, width, height, zoom=4.0, fov_deg=40.0, material=0):
    """Frame block for a camera at (0,0,zoom) looking down -z at an un-rotated object at
    the origin -- the reference's start-up pose (ray.cpp:1077-1088) without a World."""
    N = pkg._native
    p = N.FrameParams()
    lib = pkg._native.load_hip()
    lib.shray_frame_params_init(C.byref(p))
    p.camera_matrix[14] = zoom
    fov = np.float32(fov_deg) / np.float32(180) * np.pi
    ipw = np.float32(2) * np.tan(np.float32(np.float32(fov) / 2.0), dtype=np.float32)
    p.image_plane_width = float(ipw)
    p.aspect = float(np.float32(height) / np.float32(width))
    p.right[0] = float(np.float32(ipw) / np.float32(width))
    p.up[1] = float(np.float32(ipw) * np.float32(p.aspect) / np.float32(height))
    p.light_dir[:] = [0.241816789, 0.241816789, 0.939725816]
    mats = [(1, .71, .29), (.95, .95, .88), (.95, .64, .54), (.56, .57, .58), (.91, .92, .92), (.03, .03, .03), (.05, .05, .05)]
    p.specular_color[:] = mats[material]
    p.diffuse_color[:] = [0, 0, 0] if material < 5 else [1, 1, 1]
    return p


def mismatch_mask(a: np.ndarray, b: np.ndarray) -> np.ndarray:
    """Per-pixel True where any channel differs by more than REL_TOL relative (with ABS_FLOOR)."""
    a = a.astype(np.float64)
    b = b.astype(np.float64)
    tol = REL_TOL * np.maximum(np.abs(a), np.abs(b)) + ABS_FLOOR
    return (np.abs(a - b) > tol).any(axis=-1) | np.isnan(a).any(axis=-1) | np.isnan(b).any(axis=-1)


def assert_images_match(got: np.ndarray, want: np.ndarray, what: str, allowed_bad: int = 0):
    assert got.shape == want.shape, (got.shape, want.shape)
    bad = mismatch_mask(got, want)
    n = int(bad.sum())
    if n > allowed_bad:
        ys, xs = np.nonzero(bad)
        first = [(int(x), int(y), got[y, x].tolist(), want[y, x].tolist()) for y, x in list(zip(ys, xs))[:5]]
        raise AssertionError(f"{what}: {n} of {bad.size} pixels outside {REL_TOL} relative; first: {first}")
