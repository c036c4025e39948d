"""Deterministic synthetic inputs for tests and benchmarks.

The reference's benchmark assets (`bunny.trisrc`, `pisa.hdr`) live in a separate,
un-vendored repository (reference README.md:14) and are not available offline, so the
workloads are generated here with fixed seeds:

* a "bunny-class" closed, smooth, partly concave mesh of ~69k triangles, written as
  trisrc TEXT so the real parser runs (grammar: reference trisrc-support.cpp:50-84);
* a ~1M-triangle displaced sphere written as Wavefront OBJ *without* `vn` lines, which
  exercises the loader's area-weighted normal synthesis (reference obj-support.cpp:104-146);
* environments: a constant colour (reference ray.cpp:1004-1008), the reference's own
  procedural `grid` (ray.cpp:1009-1029) and a seeded float32 lat-long HDR sky with a
  sun well above 1.0.

Only numpy is used; nothing here touches the GPU.
"""
from __future__ import annotations

import os

import numpy as np

__all__ = [
    "lobed_sphere_mesh", "write_trisrc", "write_obj", "bunny_class_trisrc", "million_triangle_obj",
    "environment_constant", "environment_grid", "environment_hdr_sky",
    "scene_file", "bunny_trisrc", "million_obj", "small_trisrc", "small_obj_no_normals",
]


def _radius(theta: np.ndarray, phi: np.ndarray, bumpiness: float, ears: bool) -> np.ndarray:
    """Radial displacement of the unit sphere: low-frequency lobes, optional two 'ears'."""
    r = 1.0 + bumpiness * (0.55 * np.sin(3.0 * phi) * np.sin(theta) ** 2 * np.sin(2.0 * theta)
                           + 0.35 * np.cos(5.0 * phi + 0.7) * np.sin(theta) ** 3
                           + 0.25 * np.cos(4.0 * theta))
    if ears:
        # two gaussian lobes near the top, tilted apart: long, thin, mutually visible
        d = np.stack([np.sin(theta) * np.cos(phi), np.cos(theta), np.sin(theta) * np.sin(phi)], axis=-1)
        for ex, ez in ((0.38, 0.10), (-0.38, 0.10)):
            e = np.array([ex, 0.90, ez])
            e /= np.linalg.norm(e)
            cosang = np.clip(d @ e, -1.0, 1.0)
            r = r + 0.85 * np.exp(-((np.arccos(cosang) / 0.22) ** 2))
    return r


def lobed_sphere_mesh(n_lat: int, n_lon: int, bumpiness: float = 0.22, ears: bool = True,
                      scale: float = 1.0, center=(0.0, 0.0, 0.0)):
    """Closed UV-sphere mesh with pole fans.

    Returns (positions float32 [V,3], triangles int32 [T,3]) with
    T = 2*n_lon + 2*n_lon*(n_lat-2) and outward-facing counter-clockwise winding.
    """
    assert n_lat >= 3 and n_lon >= 3
    theta = np.linspace(0.0, np.pi, n_lat + 1)[1:-1]            # interior rings
    phi = np.arange(n_lon) * (2.0 * np.pi / n_lon)
    tt, pp = np.meshgrid(theta, phi, indexing="ij")
    rr = _radius(tt, pp, bumpiness, ears)
    ring = np.stack([rr * np.sin(tt) * np.cos(pp), rr * np.cos(tt), rr * np.sin(tt) * np.sin(pp)], axis=-1)
    top = np.array([[0.0, float(_radius(np.array(0.0), np.array(0.0), bumpiness, ears)), 0.0]])
    bot = np.array([[0.0, -float(_radius(np.array(np.pi), np.array(0.0), bumpiness, ears)), 0.0]])
    pos = np.concatenate([top, ring.reshape(-1, 3), bot], axis=0)
    pos = pos * scale + np.asarray(center, dtype=np.float64)

    n_rings = n_lat - 1

    def vid(i, j):
        return 1 + i * n_lon + (j % n_lon)

    j = np.arange(n_lon)
    tris = [np.stack([np.zeros(n_lon, dtype=np.int64), vid(0, j + 1), vid(0, j)], axis=-1)]
    for i in range(n_rings - 1):
        a, b, c, d = vid(i, j), vid(i, j + 1), vid(i + 1, j), vid(i + 1, j + 1)
        tris.append(np.stack([a, b, d], axis=-1))
        tris.append(np.stack([a, d, c], axis=-1))
    last = 1 + n_rings * n_lon
    tris.append(np.stack([np.full(n_lon, last, dtype=np.int64), vid(n_rings - 1, j), vid(n_rings - 1, j + 1)], axis=-1))
    tri = np.concatenate(tris, axis=0).astype(np.int32)
    return pos.astype(np.float32), tri


def _smooth_normals(pos: np.ndarray, tri: np.ndarray) -> np.ndarray:
    p = pos.astype(np.float64)
    fn = np.cross(p[tri[:, 1]] - p[tri[:, 0]], p[tri[:, 2]] - p[tri[:, 0]])
    n = np.zeros_like(p)
    for k in range(3):
        np.add.at(n, tri[:, k], fn)
    n /= np.maximum(np.linalg.norm(n, axis=1, keepdims=True), 1e-30)
    return n.astype(np.float32)


def _fmt(a: np.ndarray) -> np.ndarray:
    """float32 -> shortest-roundtrip-safe decimal strings (9 significant digits)."""
    return np.char.mod("%.9g", a.astype(np.float64))


def write_trisrc(path: str, pos: np.ndarray, tri: np.ndarray, normals: np.ndarray | None = None,
                 color=(0.8, 0.8, 0.8)) -> int:
    """Writes the mesh in trisrc text form; returns the triangle count."""
    if normals is None:
        normals = _smooth_normals(pos, tri)
    ps, ns = _fmt(pos), _fmt(normals)
    vline = np.array([" ".join(ps[i]) + " " + " ".join(ns[i]) for i in range(len(pos))], dtype=object)
    tail = " %.9g %.9g %.9g 1 0 0\n" % tuple(color)
    head = '"*" mesh 0.5 0.5 0.5 1 20\n'
    with open(path, "w") as f:
        chunk = []
        for t in tri:
            chunk.append(head)
            chunk.append(vline[t[0]] + tail)
            chunk.append(vline[t[1]] + tail)
            chunk.append(vline[t[2]] + tail)
            if len(chunk) >= 1 << 16:
                f.write("".join(chunk))
                chunk = []
        f.write("".join(chunk))
    return len(tri)


def write_obj(path: str, pos: np.ndarray, tri: np.ndarray, normals: np.ndarray | None = None,
              quads: np.ndarray | None = None) -> int:
    """Writes `v` (and `vn` when normals are given) plus `f` lines.  `quads` ([Q,4] indices)
    are written as 4-corner faces, which the loader fan-triangulates.  Returns the number
    of triangles the loader will produce."""
    with open(path, "w") as f:
        f.write("# synthetic mesh written by shader-ray_amd/scenes.py\n")
        ps = _fmt(pos)
        f.write("".join("v " + " ".join(r) + "\n" for r in ps))
        if normals is not None:
            ns = _fmt(normals)
            f.write("".join("vn " + " ".join(r) + "\n" for r in ns))
            f.write("".join("f %d//%d %d//%d %d//%d\n" % (a, a, b, b, c, c) for a, b, c in (tri + 1)))
        else:
            f.write("".join("f %d %d %d\n" % (a, b, c) for a, b, c in (tri + 1)))
        n = len(tri)
        if quads is not None:
            f.write("".join("f %d %d %d %d\n" % tuple(q) for q in (quads + 1)))
            n += 2 * len(quads)
    return n


def bunny_class_trisrc(path: str, n_lat: int = 132, n_lon: int = 264) -> int:
    """The benchmark stand-in for bunny.trisrc: 69,168 triangles at the default size
    (the Stanford bunny has 69,451)."""
    pos, tri = lobed_sphere_mesh(n_lat, n_lon, bumpiness=0.22, ears=True)
    return write_trisrc(path, pos, tri)


def million_triangle_obj(path: str, n_lat: int = 501, n_lon: int = 1000) -> int:
    """Deep-BVH stress mesh: 1,000,000 triangles at the default size, bumpy, no normals."""
    pos, tri = lobed_sphere_mesh(n_lat, n_lon, bumpiness=0.12, ears=False)
    rng = np.random.default_rng(20240611)
    # fine radial noise so neighbouring leaves overlap and rays graze many boxes
    pos = (pos.astype(np.float64) * (1.0 + 0.004 * rng.standard_normal((len(pos), 1)))).astype(np.float32)
    return write_obj(path, pos, tri)


def environment_constant(rgb=(0.5, 0.6, 0.7)) -> np.ndarray:
    return np.asarray(rgb, dtype=np.float32).reshape(1, 1, 3)


def environment_grid(width: int = 2048) -> np.ndarray:
    """The reference's `grid` background: 8-pixel tiles with 1-pixel white bars (ray.cpp:1009-1029)."""
    height = width // 2
    i = np.arange(width)[None, :]
    j = np.arange(height)[:, None]
    on = ((i % 8) < 1) | ((j % 8) < 1)
    img = np.zeros((height, width, 3), dtype=np.float32)
    img[on] = 1.0
    return img


def environment_hdr_sky(width: int = 2048, seed: int = 7) -> np.ndarray:
    """Seeded float32 lat-long HDR: horizon gradient, ground, soft clouds, a sun peaking near 60."""
    height = width // 2
    rng = np.random.default_rng(seed)
    s = (np.arange(width) + 0.5) / width
    t = (np.arange(height) + 0.5) / height                      # row 0 = texture t near 0 = straight down
    ss, tt = np.meshgrid(s, t)
    elev = (tt - 0.5) * np.pi                                   # -pi/2 (down) .. +pi/2 (up)
    sky = np.stack([0.25 + 0.45 * (1 - tt), 0.40 + 0.40 * (1 - tt), 0.75 + 0.20 * (1 - tt)], axis=-1)
    ground = np.stack([0.22 + 0.1 * tt, 0.18 + 0.1 * tt, 0.12 + 0.1 * tt], axis=-1)
    img = np.where((elev > 0)[..., None], sky, ground)
    # low-frequency "clouds": a few random cosine waves
    clouds = np.zeros_like(ss)
    for _ in range(6):
        fx, fy = rng.integers(1, 9), rng.integers(1, 6)
        ph = rng.uniform(0, 2 * np.pi)
        clouds += np.cos(2 * np.pi * (fx * ss + fy * tt) + ph) / 6.0
    img = img * (1.0 + 0.25 * clouds[..., None] * (elev > 0)[..., None])
    # sun at azimuth s=0.31, elevation 38 degrees
    az = 2 * np.pi * (ss - 0.31)
    cosd = np.sin(elev) * np.sin(np.radians(38)) + np.cos(elev) * np.cos(np.radians(38)) * np.cos(az)
    ang = np.arccos(np.clip(cosd, -1, 1))
    sun = 60.0 * np.exp(-((ang / 0.035) ** 2)) + 1.5 * np.exp(-((ang / 0.25) ** 2))
    img = img + sun[..., None] * np.array([1.0, 0.93, 0.80])
    return np.ascontiguousarray(img.astype(np.float32))


def cached_path(name: str) -> str:
    """Scratch location for generated scene files (kept out of the repository)."""
    root = os.environ.get("SHRAY_SCENE_CACHE", os.path.join(os.environ.get("TMPDIR", "/tmp"), "shray_scenes"))
    os.makedirs(root, exist_ok=True)
    return os.path.join(root, name)


def scene_file(name: str, maker) -> str:
    """Generates `name` once per machine under the scene cache (maker(path) writes it) and returns its path."""
    path = cached_path(name)
    if not os.path.exists(path):
        tmp = path + ".tmp%d" % os.getpid() + os.path.splitext(path)[1]
        maker(tmp)
        os.replace(tmp, path)
    return path


def bunny_trisrc() -> str:
    """The benchmark mesh of BASELINE configs 1, 2, 3 and 5 (stand-in for bunny.trisrc) as a trisrc file."""
    return scene_file("bunny_class_132x264.trisrc", bunny_class_trisrc)


def million_obj() -> str:
    """BASELINE config 4's 1M-triangle OBJ (no `vn` lines: the loader synthesizes the normals)."""
    return scene_file("million_501x1000.obj", million_triangle_obj)


def small_trisrc() -> str:
    def make(path):
        pos, tri = lobed_sphere_mesh(24, 48, bumpiness=0.22, ears=True)
        write_trisrc(path, pos, tri)
    return scene_file("small_lobed_24x48.trisrc", make)


def small_obj_no_normals() -> str:
    def make(path):
        pos, tri = lobed_sphere_mesh(40, 64, bumpiness=0.15, ears=False, scale=3.0, center=(5.0, -2.0, 1.0))
        write_obj(path, pos, tri)
    return scene_file("small_sphere_40x64.obj", make)
