#!/usr/bin/env python3
"""Writes tests/golden/glsl_reference/<case>.npz: frames rendered by the REFERENCE'S OWN SHADERS.

Runs only where the reference tree is (this container): raytracer.vs and raytracer.es.fs are read from /root/reference
at run time, compiled unmodified as "#version 140" (ray.cpp:401) by Mesa's llvmpipe -- a CPU implementation of desktop
OpenGL shipped in the image -- in a 3.2 core context (ray.cpp:964-967) made through the driver's own loader interface,
behind oracle/glsl_ref/glsl_ref.cpp, which restates ray.cpp's texture uploads, uniforms and draw call.  The inputs of
every case are rebuilt from tests/glsl_cases.py (committed scenes, deterministic generators); a fixture holds the
reference's output frame, a hash of the inputs, and what the GL implementation said about itself.

    python tests/golden/make_glsl_reference.py [case ...]
"""
import os
import subprocess
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]

from __graft_entry__ import load_package  # noqa: E402
import glsl_cases  # noqa: E402
import oracle  # noqa: E402


def main():
    subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "glsl_ref"], check=True, stdout=subprocess.DEVNULL)
    if not oracle.reference_shader_available():
        raise SystemExit("the reference's shaders cannot run here (reference tree, Mesa software driver or harness missing)")
    pkg = load_package()
    wanted = sys.argv[1:]
    os.makedirs(glsl_cases.FIXTURES, exist_ok=True)
    for name, case in glsl_cases.cases(pkg).items():
        if wanted and name not in wanted:
            continue
        t0 = time.time()
        frame, log = oracle.render_reference_shader(case["scene"][0], case["env"], case["params"], case["width"], case["height"],
                                                    case["background_mode"], case["anisotropy"])
        np.savez_compressed(os.path.join(glsl_cases.FIXTURES, name + ".npz"), frame=frame, input_hash=glsl_cases.input_hash(case),
                            gl=log.splitlines()[0], anisotropy=-1.0 if case["anisotropy"] is None else case["anisotropy"],
                            background_mode=case["background_mode"])
        print(f"{name}: {case['width']}x{case['height']} in {time.time() - t0:.1f} s; {log.splitlines()[0]}", flush=True)


if __name__ == "__main__":
    main()
