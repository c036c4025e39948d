"""Why a pixel of the oracle's frame is further than 1e-4 from the reference shaders' frame: a DERIVED budget.

The reference's GLSL runs on a compiler whose built-ins are not the oracle's explicit fp32 sequences.  Measured on the
driver that made the fixtures (tests/golden/measure_glsl_functions.py -> tests/golden/glsl_reference/driver_functions.json):
acos is up to 1.6e-4 rad off (the oracle's: 2.8e-7), atan 3.3e-6, pow(x, 5) 4.2e-6 relative; normalize, division, sqrt and
inversesqrt are correctly rounded.  Such differences move a pixel by more than 1e-4 only where the frame is ill-conditioned:

  discontinuity   the pixel, or one of its eight neighbours, takes another PATH (oracle.render_with_paths: which bounces
                  hit, which hits were lit, the iteration-cap marker): a silhouette, a shadow edge, a ray that leaves one
                  bounce earlier -- a last-bit difference decides which side the pixel falls on;
  sensitivity     the oracle's OWN pixel moves when its inputs are perturbed by what the driver's functions are off by:
                  the camera frame turned by +-DIRECTION_EPS about two axes (arithmetic noise of a ray direction, amplified by
                  every curved mirror bounce: the 1M-facet sphere), the environment lookup shifted by the driver's acos / atan
                  deviation (|grad env| x deviation, through the Fresnel factors and the tone map).  The pixel's difference
                  must stay within 1e-4 + SAFETY x the largest such move in its 3x3 neighbourhood -- or that move is itself
                  beyond CHAOTIC = ten times the tolerance: a pixel that a turn of the camera by a millionth of a radian moves by
                  1e-3 is not determined to 1e-4 by ANY fp32 evaluation of the shader (the third bounce lands on another facet);
                  the budget is linear, such a pixel's response is not.
  edge            one of the pixel's rays -- its shadow rays included (round 5) -- passes within EDGE_MARGIN / DECISION_MARGIN (a
                  barycentric coordinate) of a triangle's boundary, e.g. of an edge two triangles share:
                  the triangle test of fs:333-340 is not watertight, and in another arithmetic such a ray misses BOTH triangles
                  (it goes on through the mesh and bounces inside) or hits the other one.  Two pixels of a 1080p frame of the
                  bunny-class mesh, found that way: margins 3.6e-6 and 6.9e-6, where 76 of 2,073,600 pixels are below 1e-5.

  swallowed       the reference's pixel is exactly black -- a NaN through tonemap_and_gamma -- where the oracle's is lit, among
                  neighbours whose LATER bounces take other paths: the weakly weighted tail of its path ran into one of the shader's
                  NaN corners (fs:481, fs:130) in the reference's arithmetic.  Pinned by a frame of the reference's own shaders:
                  million_matte_traps_constant_384 (tests/glsl_cases.py; test_a_nan_in_the_tail_of_a_path_swallows_the_pixel).

A pixel outside 1e-4 that is neither is UNEXPLAINED; tests allow none (tests/test_reference_shader.py).

THE CLASSIFIER IS FROZEN (round 6).  Every arm above is exhibited by a committed frame of the reference's own shaders.  A new way to
explain a pixel needs such a fixture FIRST -- a small scene, rendered by tests/golden/make_glsl_reference.py, in which the mechanism
is the only explanation of at least one pixel, and a test that asserts exactly that -- before it may be added here.
"""
from __future__ import annotations

import json
import os

import numpy as np

import glsl_cases

EDGE_MARGIN = 2.0e-5       # barycentric; fp32 rounding of u, v at a distance of ~300 triangle sizes is a few 1e-5
DIRECTION_EPS = (1.0e-6, 3.0e-6)   # radians: ~10 and ~30 x the rounding of one fp32 operation on a unit vector, for the ~50
                                   # operations between a pixel and its third bounce (the driver's normalize / division / sqrt
                                   # are correctly rounded; its contraction of a * b + c is not the oracle's)
DECISION_MARGIN = 2.0e-5   # any decision of the path's triangle tests (shadow rays included) this close to falling the other way: the
                           # tested point from the triangle's boundary (barycentric), its distance from an end of the leaf's range
                           # (relative), a determinant from fs:312's threshold, Schlick's pow base from zero (oracle:
                           # Ctx::decision_margin).  Found that way in the 1M-triangle scene's matte 1080p frame: a primary ray
                           # through a crack 2.6e-5 wide, 1.4e-5 from the nearer triangle
SAFETY = 2.0
CHAOTIC = 1.0e-3


def driver_functions():
    return json.load(open(os.path.join(glsl_cases.FIXTURES, "driver_functions.json")))


def deviation(got, want):
    """per pixel: the largest (|got - want| - 1e-6) / |want| over R, G, B -- the quantity helpers.mismatch_mask holds to 1e-4"""
    diff = np.abs(got[..., :3].astype(np.float64) - want[..., :3].astype(np.float64)) - 1e-6
    return np.max(np.maximum(diff, 0.0) / np.maximum(np.abs(want[..., :3].astype(np.float64)), 1e-30), axis=-1)


def neighbourhood_max(plane):
    padded = np.pad(plane, 1, mode="edge")
    h, w = plane.shape
    return np.max([padded[1 + dy:1 + dy + h, 1 + dx:1 + dx + w] for dy in (-1, 0, 1) for dx in (-1, 0, 1)], axis=0)


def path_changes_nearby(path):
    """True where the pixel's path differs from one of its eight neighbours'"""
    padded = np.pad(path, 1, mode="edge")
    h, w = path.shape
    differs = np.zeros(path.shape, dtype=bool)
    for dy in (-1, 0, 1):
        for dx in (-1, 0, 1):
            differs |= padded[1 + dy:1 + dy + h, 1 + dx:1 + dx + w] != path
    return differs


def turned_camera(params, axis, angle):
    """the frame block with the camera's direction frame turned by `angle` about its own x (axis 0) or y (axis 1) axis"""
    q = params.copy()
    m = np.array(params.camera_normal_matrix[:], dtype=np.float64).reshape(4, 4).T     # column-major -> rows
    s = angle
    turn = (np.array([[1, 0, 0, 0], [0, 1, -s, 0], [0, s, 1, 0], [0, 0, 0, 1.0]]) if axis == 0
            else np.array([[1, 0, s, 0], [0, 1, 0, 0], [-s, 0, 1, 0], [0, 0, 0, 1.0]]))
    flat = (m @ turn).T.reshape(-1)
    for k in range(16):
        q.camera_normal_matrix[k] = float(np.float32(flat[k]))
    return q


def shifted_environment(env, ds, dt):
    """the environment as a lookup at (s + ds, t + dt) would see it (first order: linear interpolation between
    neighbouring texels; s wraps, t clamps)"""
    h, w, _ = env.shape
    out = env.astype(np.float64)
    fx, fy = ds * w, dt * h
    if fx:
        other = np.roll(out, -1 if fx > 0 else 1, axis=1)
        out = out + abs(fx) * (other - out)
    if fy:
        other = np.concatenate([out[1:], out[-1:]], axis=0) if fy > 0 else np.concatenate([out[:1], out[:-1]], axis=0)
        out = out + abs(fy) * (other - out)
    return np.ascontiguousarray(out, dtype=np.float32)


def classify(oracle_mod, case, want, threads=0):
    """dict(bad, discontinuity, sensitivity, edge, unexplained: pixel counts; unexplained_mask; worst_unexplained)."""
    desc, env, params, w, h = case["scene"][0], case["env"], case["params"], case["width"], case["height"]
    dev = driver_functions()
    try:
        oracle_mod.set_env_storage(case["env_storage"])
        base, _, path, _, margin, _, decisions = oracle_mod.render_with_paths(desc, env, params, w, h, threads=threads, with_decisions=True)
        moved = np.zeros((h, w))
        for eps in DIRECTION_EPS:
            for axis in (0, 1):
                for sign in (1.0, -1.0):
                    frame, _ = oracle_mod.render(desc, env, turned_camera(params, axis, sign * eps), w, h, 1, threads=threads)
                    moved = np.maximum(moved, deviation(frame, base))
        if float(env.max()) != float(env.min()):        # (a constant environment has no gradient)
            ds = dev["atan_abs_dev_driver"] / (2.0 * np.pi)
            dt = dev["acos_abs_dev_driver"] / np.pi
            for shift in ((ds, 0.0), (-ds, 0.0), (0.0, dt), (0.0, -dt)):
                frame, _ = oracle_mod.render(desc, shifted_environment(env, *shift), params, w, h, 1, threads=threads)
                moved = np.maximum(moved, deviation(frame, base))
    finally:
        oracle_mod.set_env_storage(0)
    err = deviation(base, want)
    bad = err > 1e-4
    # (a case may name the path bits its pixels' values depend on: glsl_cases.MATTE_PATH_BITS)
    on_edge = path_changes_nearby(path & np.uint32(case.get("path_bits", 0xffffffff)))
    nearby = neighbourhood_max(moved)
    within_budget = err <= 1e-4 + SAFETY * nearby
    sensitive = within_budget | (nearby >= CHAOTIC)
    # (round 5: `decisions` extends the hits' margins to every triangle test of the path, shadow rays included -- a shadow ray
    # that grazes an occluder's silhouette or slips between two triangles of it, a primary ray that passes a crack 2.6e-5 wide)
    near_an_edge = (margin < EDGE_MARGIN) | (decisions < DECISION_MARGIN)
    # a pixel the reference's frame has BLACK (all channels exactly 0: tonemap_and_gamma's max(0, c - .004) of a NaN) where the
    # oracle's is not, among neighbours whose LATER bounces take other paths (the unmasked path): the chaotic tail of its path --
    # weighted by a small Fresnel factor, so that the perturbations hardly move the pixel -- ran into one of the shader's NaN
    # corners in the reference's arithmetic (fs:481's pow of a negative base at normal incidence, fs:130's acos outside [-1, 1]),
    # and a NaN swallows the pixel whatever its weight.  One pixel of the 1M-triangle scene's matte 1080p frame (round 5).
    swallowed = bad & np.all(want[..., :3] == 0.0, axis=-1) & np.any(base[..., :3] != 0.0, axis=-1) & path_changes_nearby(path)
    unexplained = bad & ~on_edge & ~sensitive & ~near_an_edge & ~swallowed
    # what the CHAOTIC escape alone lets through: bad pixels off every change of path whose error EXCEEDS the linear budget and
    # which are admitted only because their neighbourhood moves by >= CHAOTIC under the perturbations.  The escape has no bound
    # of its own on the error; the tests bound how many pixels may take it and report the worst of them
    chaotic_only = bad & ~on_edge & ~within_budget & (nearby >= CHAOTIC)
    return {"bad": int(bad.sum()), "pixels": int(bad.size), "discontinuity": int((bad & on_edge).sum()),
            "sensitivity": int((bad & ~on_edge & sensitive).sum()), "edge": int((bad & ~on_edge & ~sensitive & near_an_edge).sum()),
            "edge_candidates": int(near_an_edge.sum()), "unexplained": int(unexplained.sum()),
            "swallowed_by_nan": int((bad & ~on_edge & ~sensitive & ~near_an_edge & swallowed).sum()),
            "unexplained_mask": unexplained, "worst_unexplained": float(err[unexplained].max()) if unexplained.any() else 0.0,
            "chaotic_admitted": int(chaotic_only.sum()), "chaotic_admitted_worst": float(err[chaotic_only].max()) if chaotic_only.any() else 0.0,
            "ill_conditioned_pixels": int((on_edge | near_an_edge | (nearby > 1e-4)).sum()), "frame": base}
