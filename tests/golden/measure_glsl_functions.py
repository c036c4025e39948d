#!/usr/bin/env python3
"""How far this GL implementation's built-ins are from the exact functions: the perturbation sizes behind the pixel
classifier of tests/test_reference_shader.py.

The oracle states atan / acos / pow(x, 5) / normalize as explicit fp32 sequences; the GLSL compiler that runs the
reference's shaders here (Mesa llvmpipe) has its own.  This script runs SMALL PROBE SHADERS OF ITS OWN (written below --
not the reference's) through the same harness (oracle/glsl_ref: the context, the quad, an RGBA32F target) and compares
the driver's atan(y, x), acos(x), pow(x, 5.0), normalize() and division with float64 and with the oracle's sequences.
Output: tests/golden/glsl_reference/driver_functions.json -- the measured deviations, committed; the classifier derives
its perturbations from them.  Runs only where the harness library is built (the container with the reference tree).

    python tests/golden/measure_glsl_functions.py
"""
import json
import os
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]

from __graft_entry__ import load_package  # noqa: E402
import glsl_cases  # noqa: E402
import oracle  # noqa: E402

N = 512   # the probe frames are N x N

VS = """
in vec4 pos;
in vec2 vtex;
uniform mat4 modelview;
void main() { gl_Position = modelview * pos; }
"""

# `which` (a uniform the harness sets from the frame parameters) selects the probe; every frame also returns its inputs,
# so that the comparison uses exactly the floats the driver worked on
FS = """
precision highp float;
uniform int which;
out vec4 fragment_color;
void main()
{
    float a = -1.0 + 2.0 * (floor(gl_FragCoord.x) + 0.5) / %(N)d.0;
    float b = -1.0 + 2.0 * (floor(gl_FragCoord.y) + 0.5) / %(N)d.0;
    float c = (floor(gl_FragCoord.y) * %(N)d.0 + floor(gl_FragCoord.x) + 0.5) / (%(N)d.0 * %(N)d.0);   // (0, 1)
    if (which == 0)
        fragment_color = vec4(atan(a, b), a, b, 1.0);
    else if (which == 1)
        fragment_color = vec4(acos(-1.0 + 2.0 * c), -1.0 + 2.0 * c, pow(c, 5.0), c);
    else if (which == 2)
        fragment_color = vec4(normalize(vec3(a, b, -1.0)), 1.0);
    else if (which == 3)
        fragment_color = vec4(a / (1.5 + b), sqrt(1.5 + b), inversesqrt(1.5 + b), 1.0);
    else    // the corners GLSL leaves undefined: pow of a negative base, acos outside [-1, 1]
        fragment_color = vec4(pow(-c * 1.0e-6, 5.0), pow(-c, 5.0), acos(1.0 + c * 1.0e-5), acos(-1.0 - c * 1.0e-5));
}
""" % {"N": N}


def main():
    if not os.path.exists(oracle.GLSL_REF_LIB):
        raise SystemExit("oracle/_ref/libglsl_ref.so is not built (it is built where the reference tree is)")
    pkg = load_package()
    case = glsl_cases.cases(pkg)["kat_mirror_quad"]       # any scene: the probe shaders read none of it
    frames = {}
    with tempfile.TemporaryDirectory() as tmp:
        open(os.path.join(tmp, "raytracer.vs"), "w").write(VS)
        open(os.path.join(tmp, "raytracer.es.fs"), "w").write(FS)
        saved = oracle.REFERENCE_DIR
        oracle.REFERENCE_DIR = tmp
        try:
            for which in range(5):
                params = case["params"].copy()
                params.which = which
                frames[which], log = oracle.render_reference_shader(case["scene"][0], case["env"], params, N, N, 0, 1.0)
        finally:
            oracle.REFERENCE_DIR = saved
    f64 = np.float64
    out = {"gl": log.splitlines()[0], "probe_points": N * N}
    # atan(y, x)
    f = frames[0]
    a, b = f[..., 1].astype(f64), f[..., 2].astype(f64)
    true = np.arctan2(a, b)
    ours = np.vectorize(oracle.atan2)(f[..., 1], f[..., 2]).astype(f64)
    out["atan_abs_dev_driver"] = float(np.max(np.abs(f[..., 0] - true)))
    out["atan_abs_dev_oracle"] = float(np.max(np.abs(ours - true)))
    # acos, pow(x, 5)
    f = frames[1]
    x, c = f[..., 1].astype(f64), f[..., 3].astype(f64)
    out["acos_abs_dev_driver"] = float(np.max(np.abs(f[..., 0] - np.arccos(x))))
    out["acos_abs_dev_oracle"] = float(np.max(np.abs(np.vectorize(oracle.acos)(f[..., 1]).astype(f64) - np.arccos(x))))
    out["pow5_rel_dev_driver"] = float(np.max(np.abs(f[..., 2] - c ** 5) / c ** 5))
    out["pow5_rel_dev_oracle"] = float(np.max(np.abs(np.vectorize(oracle.pow5)(f[..., 3]).astype(f64) - c ** 5) / c ** 5))
    # normalize: the largest error of a component relative to the unit length (an angle, in radians)
    f = frames[2]
    gx = -1.0 + 2.0 * (np.arange(N) + 0.5) / N
    v = np.stack(np.broadcast_arrays(gx[None, :].astype(np.float32), gx[:, None].astype(np.float32), np.float32(-1.0)), axis=-1).astype(f64)
    unit = v / np.linalg.norm(v, axis=-1, keepdims=True)
    out["normalize_abs_dev_driver"] = float(np.max(np.abs(f[..., :3] - unit)))
    v32 = v.astype(np.float32)
    len32 = np.sqrt((v32[..., 0] * v32[..., 0] + v32[..., 1] * v32[..., 1] + v32[..., 2] * v32[..., 2]).astype(np.float32)).astype(np.float32)
    out["normalize_abs_dev_oracle"] = float(np.max(np.abs((v32 / len32[..., None]).astype(f64) - unit)))
    # division, sqrt, inversesqrt (relative)
    f = frames[3]
    a32 = (-1.0 + 2.0 * (np.arange(N) + 0.5) / N).astype(np.float32)
    num, den = a32[None, :].astype(f64), (np.float32(1.5) + a32[:, None]).astype(f64)
    out["division_rel_dev_driver"] = float(np.max(np.abs(f[..., 0] - num / den) / np.abs(num / den)))
    out["sqrt_rel_dev_driver"] = float(np.max(np.abs(f[..., 1] - np.sqrt(den)) / np.sqrt(den)))
    out["inversesqrt_rel_dev_driver"] = float(np.max(np.abs(f[..., 2] - 1 / np.sqrt(den)) / (1 / np.sqrt(den))))
    # the undefined corners: what fraction of the probe points came back NaN
    f = frames[4]
    out["pow_negative_base_is_nan"] = float(np.mean(np.isnan(f[..., 0]) & np.isnan(f[..., 1])))
    out["acos_outside_domain_is_nan"] = float(np.mean(np.isnan(f[..., 2]) & np.isnan(f[..., 3])))
    path = os.path.join(glsl_cases.FIXTURES, "driver_functions.json")
    json.dump(out, open(path, "w"), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
