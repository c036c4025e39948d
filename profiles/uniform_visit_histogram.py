#!/usr/bin/env python3
"""Wave-uniform node visits by outcome (diagnostic build with -DSHRAY_DIAG_UNIFORM):
    make -C shader-ray_amd variant VARIANT=uniform HIP_EXTRA="-DSHRAY_DIAGNOSTICS -DSHRAY_DIAG_UNIFORM"
    SHRAY_DIAG_LIB=shader-ray_amd/_variants/libshray_hip_uniform.so python profiles/uniform_visit_histogram.py [--million] [--material 6]
Of the node stage's wave-visits whose walking lanes are all at one record -- the ones a packet-level (scalar interval) test could
decide for the whole wave -- how many end with every lane entering, with no lane entering, or mixed."""
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
    t = stamps[:, 4:12].astype(np.float64).sum(axis=0)
    visits, uniform, all_enter, all_miss, mixed, two, three_four, lanes = t
    print(f"scene: {'1M triangles' if args.million else 'bunny-class'}, material {args.material}, {W}x{H}, one frame (the start-up view)")
    print(f"wave-visits of the node stage          {visits:14.0f}   ({lanes / visits:.1f} walking lanes on average)")
    print(f"  distinct records among the walking lanes: 1: {uniform / visits:.3f}   2: {two / visits:.3f}   3-4: {three_four / visits:.3f}   "
          f"more: {(visits - uniform - two - three_four) / visits:.3f}")
    print(f"  wave-uniform (one record)            {uniform:14.0f}   {uniform / visits:6.3f} of all")
    print(f"    every lane enters                  {all_enter:14.0f}   {all_enter / max(uniform, 1):6.3f} of the uniform ones")
    print(f"    no lane enters                     {all_miss:14.0f}   {all_miss / max(uniform, 1):6.3f}")
    print(f"    mixed                              {mixed:14.0f}   {mixed / max(uniform, 1):6.3f}")
    print(f"  a packet-level decision could settle {all_enter + all_miss:14.0f}   {(all_enter + all_miss) / max(uniform, 1):6.3f} of the uniform visits = "
          f"{(all_enter + all_miss) / visits:6.3f} of all wave-visits")


if __name__ == "__main__":
    main()
