#!/usr/bin/env python3
"""EXPERIMENTS R5.4: how often would a 16-byte node record with 16-bit quantised planes leave a visit undecided?  Runs the CPU
oracle (no GPU) on the benchmark frame with its probe on: a visit counts as undecided when the box grown by one quantisation step
and the box shrunk by one step answer the visit's question differently.
    python tests/quantised_record_probe.py [width height]"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))   # (tests/: the oracle is test infrastructure)
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
from __graft_entry__ import load_package  # noqa: E402
import oracle  # noqa: E402

pkg = load_package()
W, H = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (960, 540)
lib = oracle.load()
lib.shray_oracle_quant_probe.argtypes = [C.c_float, C.POINTER(C.c_ulonglong)]
lib.shray_oracle_quant_probe.restype = None
for name, path, spp in (("bunny-class, gold", pkg.scenes.bunny_trisrc(), 1), ("1M triangles, gold", pkg.scenes.million_obj(), 1)):
    world = pkg.World(path)
    env = pkg.scenes.environment_hdr_sky(256)
    params = world.frame_params(W, H, material=0)
    for bits in (16, 20):
        step = world.info.scene_extent / float(1 << bits)
        lib.shray_oracle_quant_probe(step, None)
        oracle.render(world.flatten(), env, params, W, H, spp)
        counts = (C.c_ulonglong * 2)()
        lib.shray_oracle_quant_probe(0.0, counts)
        visits, undecided = counts[0], counts[1]
        p = undecided / max(1, visits)
        print(f"{name}, {W}x{H}: planes in steps of extent / 2^{bits}: {undecided:,} of {visits:,} visits undecided = {100 * p:.3f} % per lane; "
              f"a wave-visit of 54 walking lanes has an undecided lane with probability {100 * (1 - (1 - p) ** 54):.1f} %", flush=True)
