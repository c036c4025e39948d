#!/usr/bin/env python3
"""Leaf stages of the dealt form by the number of parked lanes (diagnostic build with -DSHRAY_DIAG_KHIST):
    make -C shader-ray_amd variant VARIANT=khist HIP_EXTRA="-DSHRAY_DIAGNOSTICS -DSHRAY_DIAG_KHIST=1"
    SHRAY_DIAG_LIB=shader-ray_amd/_variants/libshray_hip_khist.so python profiles/leaf_stage_histogram.py [--million] [--material 6]
Per bin of K: stages, rounds of three strided fetches they run today, 16-byte-per-lane fetches they would run if each
group fetched its leaf as consecutive chunks (leaf_stage.h, variants/diag_khist_stage.inc, SHRAY_DIAG_KHIST)."""
import argparse
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--material", type=int, default=0)
    ap.add_argument("--million", action="store_true")
    ap.add_argument("--distinct", action="store_true", help="the library was built with -DSHRAY_DIAG_KHIST=2: stages by distinct leaves")
    args = ap.parse_args()
    import torch  # noqa: F401
    from __graft_entry__ import load_package
    pkg = load_package()
    N = pkg._native
    N.HIP_LIB = os.environ["SHRAY_DIAG_LIB"]
    lib = N.load_hip()
    lib.shray_debug_timeline.restype = C.c_int
    world = pkg.World(pkg.scenes.million_obj() if args.million else pkg.scenes.bunny_trisrc())
    scene = pkg.Scene(world.flatten(), pkg.scenes.environment_hdr_sky(2048), device=0)
    W, H = args.width, args.height
    params = world.frame_params(W, H, material=args.material)
    patches = ((W + 15) // 16) * ((H + 15) // 16)
    stamps = np.zeros((patches * 4, 16), dtype=np.uint64)
    N.check(lib.shray_debug_timeline(scene._handle, C.byref(params), W, H, 1, stamps.ctypes.data_as(C.c_void_p)))
    t = stamps[:, 4:12]
    if args.distinct:
        names = ["D = 1", "D = 2", "D = 3", "D = 4", "D = 5-8", "D = 9-16", "D > 16"]
        stages = np.array([(t[:, k] & np.uint64(0xffffff)).astype(np.float64).sum() for k in range(7)])
        rounds = np.array([((t[:, k] >> np.uint64(24)) & np.uint64(0xfffff)).astype(np.float64).sum() for k in range(7)])
        dsum = (t[:, 7] & np.uint64(0xffffffff)).astype(np.float64).sum()
        ksum = (t[:, 7] >> np.uint64(32)).astype(np.float64).sum()
        print(f"{'scene':>12}: {'1M triangles' if args.million else 'bunny-class'}, material {args.material}, {W}x{H}; stages with more parked lanes than the build's SHRAY_DIAG_KHIST_FROM, by distinct leaves")
        print(f"{'bin':>12} {'stages':>12} {'share':>7} {'rounds':>12} {'share':>7}")
        for n, s_, r in zip(names, stages, rounds):
            print(f"{n:>12} {s_:12.0f} {s_ / max(stages.sum(), 1):7.3f} {r:12.0f} {r / max(rounds.sum(), 1):7.3f}")
        print(f"{'all':>12} {stages.sum():12.0f} {'':7} {rounds.sum():12.0f}; mean distinct leaves {dsum / max(stages.sum(), 1):.2f}, mean K {ksum / max(stages.sum(), 1):.2f}")
        return
    names = ["K = 1", "K = 2", "K = 3-4", "K = 5-8", "K = 9-16", "K = 17-32", "K > 32 (plain loop)"]
    stages = np.array([(t[:, k] & np.uint64(0xffffff)).astype(np.float64).sum() for k in range(7)])
    rounds = np.array([((t[:, k] >> np.uint64(24)) & np.uint64(0xfffff)).astype(np.float64).sum() for k in range(7)])
    staged = np.array([(t[:, k] >> np.uint64(44)).astype(np.float64).sum() for k in range(7)])
    most = (t[:, 7] & np.uint64(0xffffffff)).astype(np.float64).sum()
    ksum = (t[:, 7] >> np.uint64(32)).astype(np.float64).sum()
    print(f"{'scene':>22}: {'1M triangles' if args.million else 'bunny-class'}, material {args.material}, {W}x{H}")
    print(f"{'bin':>22} {'stages':>12} {'share':>7} {'rounds now':>12} {'share':>7} {'fetch instr now (3/round)':>26} {'staged 16-B fetches':>20}")
    for n, s, r, g in zip(names, stages, rounds, staged):
        print(f"{n:>22} {s:12.0f} {s / stages.sum():7.3f} {r:12.0f} {r / rounds.sum():7.3f} {3 * r:26.0f} {g:20.0f}")
    print(f"{'all':>22} {stages.sum():12.0f} {'':7} {rounds.sum():12.0f} {'':7} {3 * rounds.sum():26.0f} {staged.sum() + 3 * rounds[6]:20.0f} (plain loop unchanged)")
    print(f"mean K {ksum / stages.sum():.2f}, mean longest leaf of a stage {most / stages.sum():.2f} triangles")
    # the texture addresser's cycles by R4.3's probe: 14.4 per 16-byte-per-lane instruction, 4.7 per 4-byte one
    now = rounds.sum() * (2 * 14.4 + 4.7)
    then = staged.sum() * 14.4 + rounds[6] * (2 * 14.4 + 4.7)
    print(f"vector-memory pipeline cycles of the triangle fetches at the probe's prices: {now:.4g} now, {then:.4g} staged ({then / now:.2f})")


if __name__ == "__main__":
    main()
