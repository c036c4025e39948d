"""TEST INFRASTRUCTURE: ctypes access to the CPU restatement (oracle/shader_oracle.cpp).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this module;
the product package (shader-ray_amd/) never does.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
# SHRAY_ORACLE_LIB selects another build of the checker (the sanitizer build, `make -C oracle sanitize`: tests/test_sanitizers.py)
LIB = os.environ.get("SHRAY_ORACLE_LIB") or os.path.join(HERE, "_build", "libshader_oracle.so")
REF_HOST = os.path.join(HERE, "_ref", "ref_host")

_lib = None


def build(ref: bool = True) -> None:
    subprocess.run(["make", "-C", HERE, "oracle"] + (["ref"] if ref else []), check=True,
                   stdout=subprocess.DEVNULL)


def load():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB):
            build(ref=False)
        _lib = C.CDLL(LIB)
        _lib.shray_oracle_render.restype = C.c_int
        _lib.shray_oracle_filmic.restype = C.c_float
        _lib.shray_oracle_filmic.argtypes = [C.c_float]
        _lib.shray_oracle_half.restype = C.c_float
        _lib.shray_oracle_half.argtypes = [C.c_float]
        for name, nargs in (("shray_oracle_atan2", 2), ("shray_oracle_acos", 1), ("shray_oracle_pow5", 1), ("shray_oracle_log2", 1)):
            fn = getattr(_lib, name)
            fn.restype = C.c_float
            fn.argtypes = [C.c_float] * nargs
    return _lib


def render(desc, env: np.ndarray, params, width: int, height: int, spp: int = 1, rows=None, threads: int = 0):
    """Full-frame RGBA float32 [height, width, 4] (row 0 = bottom) plus a counters dict.
    `desc`/`params` are the ctypes structs of shader-ray_amd/_native.py (same C layout)."""
    lib = load()
    env = np.ascontiguousarray(env, dtype=np.float32)
    eh, ew, _ = env.shape
    out = np.zeros((height, width, 4), dtype=np.float32)
    r0, r1 = (0, height) if rows is None else rows
    counters = (C.c_uint64 * 8)()
    rc = lib.shray_oracle_render(C.byref(desc), env.ctypes.data_as(C.c_void_p), C.c_int(ew), C.c_int(eh),
                                 C.byref(params), C.c_int(width), C.c_int(height), C.c_int(spp),
                                 C.c_int(r0), C.c_int(r1), C.c_int(threads),
                                 out.ctypes.data_as(C.c_void_p), C.byref(counters))
    if rc != 0:
        raise RuntimeError("oracle rejected its arguments")
    names = ("node_visits", "leaf_visits", "triangle_tests", "shaded_hits", "env_lookups", "traversals",
             "bad_hits", "samples")
    return out, dict(zip(names, (int(c) for c in counters)))


def render_with_paths(desc, env: np.ndarray, params, width: int, height: int, threads: int = 0, with_decisions: bool = False):
    """render() of a 1 spp plain frame plus its path planes: (frame, counters, path uint32 [h, w], first_triangle int32 [h, w],
    edge_margin float32 [h, w] = the smallest barycentric coordinate of any of the path's hits, env_dy float32 [h, w] = D.y
    of the environment lookup before the oracle clamps it into acos's domain);
    path: bit 2i = bounce i hit a triangle, bit 2i + 1 = that hit was lit, bits 24-27 = bounces that hit, bit 30 = the
    iteration-cap marker (shader_oracle.cpp: g_path_map).  Where the path of a pixel differs from a neighbour's the frame is
    discontinuous (a silhouette, a shadow edge, a ray that leaves a bounce earlier).
    with_decisions: a seventh plane, decision_margin float32 [h, w] = how close ANY triangle test of the path, shadow rays
    included, came to deciding the other way (shader_oracle.cpp: Ctx::decision_margin)."""
    lib = load()
    path = np.zeros((height, width), dtype=np.uint32)
    first = np.full((height, width), -1, dtype=np.int32)
    margin = np.ones((height, width), dtype=np.float32)
    env_dy = np.zeros((height, width), dtype=np.float32)
    decisions = np.ones((height, width), dtype=np.float32)
    lib.shray_oracle_set_path_map(path.ctypes.data_as(C.c_void_p), first.ctypes.data_as(C.c_void_p), margin.ctypes.data_as(C.c_void_p),
                                  env_dy.ctypes.data_as(C.c_void_p))
    if with_decisions:
        lib.shray_oracle_set_decision_map(decisions.ctypes.data_as(C.c_void_p))
    try:
        frame, counters = render(desc, env, params, width, height, 1, threads=threads)
    finally:
        lib.shray_oracle_set_path_map(None, None, None, None)
        lib.shray_oracle_set_decision_map(None)
    if with_decisions:
        return frame, counters, path, first, margin, env_dy, decisions
    return frame, counters, path, first, margin, env_dy


GLSL_REF_LIB = os.path.join(HERE, "_ref", "libglsl_ref.so")
REFERENCE_DIR = os.environ.get("SHRAY_REFERENCE_DIR", "/root/reference")
MESA_SOFTWARE_DRIVER = os.environ.get("SHRAY_MESA_DRIVER", "/usr/lib/x86_64-linux-gnu/dri/swrast_dri.so")
_glsl = None


def reference_shader_available() -> bool:
    """The reference's GLSL can run here: its sources, Mesa's software driver and the harness are present."""
    return (os.path.exists(GLSL_REF_LIB) and os.path.exists(os.path.join(REFERENCE_DIR, "raytracer.es.fs"))
            and os.path.exists(MESA_SOFTWARE_DRIVER))


def render_reference_shader(desc, env: np.ndarray, params, width: int, height: int, background_mode: int = 0,
                            anisotropy: float | None = None):
    """One frame of the REFERENCE'S OWN shaders (raytracer.vs + raytracer.es.fs, read from the reference tree, compiled
    unmodified as "#version 140" by Mesa's llvmpipe on the CPU; oracle/glsl_ref/glsl_ref.cpp restates ray.cpp's GL
    calls around them).  RGBA float32 [height, width, 4], row 0 = bottom, plus the harness's log (GL version, shader
    logs).  background_mode 0: sized GL_RGB32F; 1: the reference's literal unsized GL_RGB (ray.cpp:508).  anisotropy:
    override of ray.cpp:506's 4.0 (None = 4.0).  Only tests/golden/make_glsl_reference.py and the live check of
    tests/test_reference_shader.py call this."""
    global _glsl
    if _glsl is None:
        _glsl = C.CDLL(GLSL_REF_LIB)
        _glsl.shray_glsl_ref_render.restype = C.c_int
    env = np.ascontiguousarray(env, dtype=np.float32)
    out = np.zeros((height, width, 4), dtype=np.float32)
    log = C.create_string_buffer(1 << 16)
    previous = os.environ.get("SHRAY_GLSL_REF_ANISOTROPY")
    if anisotropy is not None:
        os.environ["SHRAY_GLSL_REF_ANISOTROPY"] = repr(float(anisotropy))
    try:
        rc = _glsl.shray_glsl_ref_render(REFERENCE_DIR.encode(), MESA_SOFTWARE_DRIVER.encode(), C.byref(desc),
                                         env.ctypes.data_as(C.c_void_p), C.c_int(env.shape[1]), C.c_int(env.shape[0]),
                                         C.c_int(background_mode), C.byref(params), C.c_int(width), C.c_int(height),
                                         out.ctypes.data_as(C.c_void_p), log, C.c_int(len(log)))
    finally:
        if anisotropy is not None:
            if previous is None:
                del os.environ["SHRAY_GLSL_REF_ANISOTROPY"]
            else:
                os.environ["SHRAY_GLSL_REF_ANISOTROPY"] = previous
    if rc != 0:
        raise RuntimeError(f"the reference shaders did not run (code {rc}): {log.value.decode(errors='replace')}")
    return out, log.value.decode(errors="replace")


def set_env_storage(storage: int) -> None:
    """0 = the environment's floats as given (default); 1 = stored as 8-bit normalized fixed point, mip levels included
    (what the reference's unsized GL_RGB upload, ray.cpp:508, becomes on most drivers).  Applies to later renders."""
    load().shray_oracle_set_env_storage(C.c_int(storage))


def filmic(c: float) -> float:
    return float(load().shray_oracle_filmic(C.c_float(c)))


def half(f: float) -> float:
    return float(load().shray_oracle_half(C.c_float(f)))


def atan2(y: float, x: float) -> float:
    return float(load().shray_oracle_atan2(y, x))


def acos(x: float) -> float:
    return float(load().shray_oracle_acos(x))


def pow5(x: float) -> float:
    return float(load().shray_oracle_pow5(x))


def log2(x: float) -> float:
    return float(load().shray_oracle_log2(x))


def texture_grad(image: np.ndarray, s: float, t: float, dudx: float, dvdx: float, dudy: float, dvdy: float):
    image = np.ascontiguousarray(image, dtype=np.float32)
    h, w, _ = image.shape
    out = (C.c_float * 3)()
    load().shray_oracle_texture_grad(image.ctypes.data_as(C.c_void_p), C.c_int(w), C.c_int(h), C.c_float(s), C.c_float(t),
                                     C.c_float(dudx), C.c_float(dvdx), C.c_float(dudy), C.c_float(dvdy), out)
    return np.array(out[:], dtype=np.float32)


def schlick(cspec, v, r):
    out = (C.c_float * 3)()
    load().shray_oracle_schlick((C.c_float * 3)(*cspec), (C.c_float * 3)(*v), (C.c_float * 3)(*r), out)
    return np.array(out[:], dtype=np.float32)


def primary_ray(params, u: float, v: float):
    o, d = (C.c_float * 3)(), (C.c_float * 3)()
    load().shray_oracle_primary_ray(C.byref(params), C.c_float(u), C.c_float(v), o, d)
    return np.array(o[:], dtype=np.float32), np.array(d[:], dtype=np.float32)
