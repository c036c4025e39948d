#!/usr/bin/env python3
"""DESIGN-TIME DIAGNOSTIC (test infrastructure, like everything under oracle/): records, with the CPU
oracle, the per-ray event trace of a BASELINE configuration -- for every sample, every traversal's node
visits and the triangle tests each visit ran -- so that oracle/tools/wave_sim.cpp can replay the rays
through candidate wave-scheduling policies of the HIP kernel without a GPU.

    python oracle/tools/dump_trace.py <config 2|3|4> <out prefix> [width height spp]
"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
from __graft_entry__ import load_package  # noqa: E402
import helpers  # noqa: E402
import oracle  # noqa: E402

config = int(sys.argv[1])
prefix = sys.argv[2]
W, H, SPP = (int(a) for a in sys.argv[3:6]) if len(sys.argv) > 5 else (1920, 1080, 1)
pkg = load_package()
if config == 4:
    world = pkg.World(helpers.million_obj())
    material = 0
else:
    world = pkg.World(helpers.bunny_trisrc())
    material = 6 if config == 3 else 0
desc = world.flatten()
env = pkg.scenes.environment_hdr_sky(2048)
params = world.frame_params(W, H, material=material)
lib = oracle.load()
samples = W * H * SPP
cap = samples * 400
buf = np.zeros(cap, dtype=np.uint8)
offs = np.zeros(samples + 1, dtype=np.uint64)
lib.shray_oracle_trace_begin.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64]
lib.shray_oracle_trace_end.restype = C.c_uint64
lib.shray_oracle_trace_begin(buf.ctypes.data, cap, offs.ctypes.data, samples)
_, counters = oracle.render(desc, env, params, W, H, SPP, threads=1)
n = lib.shray_oracle_trace_end()
assert n <= cap, (n, cap)
print(counters, "trace bytes", n)
np.save(prefix + "_bytes.npy", buf[:n])
np.save(prefix + "_offsets.npy", offs)
open(prefix + "_meta.txt", "w").write(f"{W} {H} {SPP}\n")
