#!/usr/bin/env python3
"""The reference's own shaders against the CPU oracle at BASELINE's full frame size (runs only where the reference tree
is: this container).  raytracer.vs + raytracer.es.fs, unmodified, on Mesa's llvmpipe (oracle/glsl_ref/glsl_ref.cpp);
prints per configuration how many pixels of the 1920 x 1080 frame agree within 1e-4 relative and where the rest sit.

    python tests/reference_shader_agreement.py > profiles/history/r03/reference_shader_agreement.txt
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
from __graft_entry__ import load_package  # noqa: E402
import oracle  # noqa: E402

pkg = load_package()
W, H = 1920, 1080
sky = pkg.scenes.environment_hdr_sky(2048)
constant = pkg.scenes.environment_constant((0.5, 0.25, 2.0))
print("reference shaders: raytracer.vs + raytracer.es.fs, '#version 140', anisotropy 1 (see tests/test_reference_shader.py); "
      "oracle: oracle/shader_oracle.cpp; relative difference per pixel = max over R, G, B of |a - b| / max(|b|, 1e-2)")
for what, path, env, env_name, material in (
        ("configs[1]: bunny-class mesh, gold", pkg.scenes.bunny_trisrc(), sky, "HDR sky 2048x1024", 0),
        ("configs[2] at 1 spp: bunny-class mesh, glazed plaster (shadow rays)", pkg.scenes.bunny_trisrc(), sky, "HDR sky 2048x1024", 6),
        ("configs[1] with a constant environment (no texture filter in the way)", pkg.scenes.bunny_trisrc(), constant, "constant", 0),
        ("configs[3] at 1 spp: 1M-triangle OBJ, gold", pkg.scenes.million_obj(), constant, "constant", 0)):
    world = pkg.World(path)
    desc = world.flatten()
    params = world.frame_params(W, H, material=material)
    t0 = time.time()
    ref, log = oracle.render_reference_shader(desc, env, params, W, H, 0, 1.0)
    t1 = time.time()
    got, counters = oracle.render(desc, np.ascontiguousarray(env, dtype=np.float32), params, W, H, 1)
    t2 = time.time()
    rel = (np.abs(got - ref)[..., :3] / np.maximum(np.abs(ref[..., :3]), 1e-2)).max(axis=-1)
    n = rel.size
    print(f"\n{what}; {world.triangle_count} triangles, {env_name}, {W}x{H}, 1 spp   [{log.splitlines()[0]}]")
    print(f"  reference shaders {t1 - t0:.1f} s, oracle {t2 - t1:.1f} s on {os.cpu_count()} cores")
    print(f"  median {np.median(rel):.1e}  p90 {np.percentile(rel, 90):.1e}  p99 {np.percentile(rel, 99):.1e}  p99.9 {np.percentile(rel, 99.9):.1e}  max {rel.max():.1e}")
    for bound in (1e-4, 1e-3, 1e-2, 1e-1):
        k = int((rel > bound).sum())
        print(f"  pixels beyond {bound:g}: {k} ({100.0 * k / n:.4f} %)")
    red = oracle.filmic(1.0)                                      # the marker (1, 0, 0), tone-mapped (fs:436-438, :566-568)
    marked = lambda f: (np.abs(f[..., 0] - red) < 1e-5) & (f[..., 1] == 0) & (f[..., 2] == 0)   # noqa: E731
    print(f"  iteration-cap marker pixels: reference {int(marked(ref).sum())}, oracle {int(marked(got).sum())}, the same pixels: "
          f"{bool(np.array_equal(marked(ref), marked(got)))}; oracle bad_hits {counters['bad_hits']}")
    ys, xs = np.nonzero(rel > 0.1)
    for y, x in list(zip(ys, xs))[:4]:
        print(f"  pixel ({x}, {y}): reference {ref[y, x, :3].round(5).tolist()}  oracle {got[y, x, :3].round(5).tolist()}")
