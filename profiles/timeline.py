#!/usr/bin/env python3
"""Per-wave residency timeline of one bench frame (diagnostic build, `make -C shader-ray_amd diag`).

Prints: kernel span, distribution of wave durations, occupancy over time (resident waves per
CU in 20 slices of the kernel span), and the busiest / idlest CUs.  Usage (GPU box):
    python profiles/timeline.py [--kernel 0|1] [--width 1920 --height 1080]
"""
import argparse
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--kernel", type=int, default=0)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--material", type=int, default=0)
    ap.add_argument("--million", action="store_true", help="the 1M-triangle scene instead of the bunny-class mesh")
    args = ap.parse_args()
    import torch  # noqa: F401  (one HIP runtime)
    from __graft_entry__ import load_package
    pkg = load_package()
    N = pkg._native
    N.HIP_LIB = os.environ.get("SHRAY_DIAG_LIB") or os.path.join(N.PKG_DIR, "libshray_hip_diag.so")
    lib = N.load_hip()
    lib.shray_debug_timeline.restype = C.c_int
    world = pkg.World(pkg.scenes.million_obj() if args.million else pkg.scenes.bunny_trisrc())
    desc = world.flatten()
    scene = pkg.Scene(desc, pkg.scenes.environment_hdr_sky(2048), device=0)
    scene.set_kernel(args.kernel)
    W, H = args.width, args.height
    params = world.frame_params(W, H, material=args.material)
    patches = ((W + 15) // 16) * ((H + 15) // 16)
    stamps = np.zeros((patches * 4, 16), dtype=np.uint64)
    N.check(lib.shray_debug_timeline(scene._handle, C.byref(params), W, H, 1, stamps.ctypes.data_as(C.c_void_p)))
    if args.kernel == 2:
        raw = stamps.reshape(-1)
        rows = raw[: (len(raw) // 16) * 16].reshape(-1, 16)
        rows = rows[rows[:, 1] > 0]
        t0 = rows[:, 0].astype(np.float64)
        t1 = rows[:, 1].astype(np.float64)
        span = (t1.max() - t0.min()) * 10e-3
        dur = (t1 - t0) * 10e-3
        print(f"persistent kernel: span {span:.1f} us, {len(rows)} waves; wave life us min {dur.min():.1f} p50 {np.median(dur):.1f} max {dur.max():.1f}")
        tot = rows[:, 2:10].astype(np.float64).sum(axis=0)
        names = ["outer iterations", "fetch passes", "shade passes", "node-loop iterations", "leaf-loop iterations",
                 "cycles in fetch", "cycles in walk (node+leaf)", "cycles in shade"]
        for n, v in zip(names, tot):
            print(f"  {n:28s} {v:.4g}")
        print(f"  per node+leaf iteration: {tot[6] / (tot[3] + tot[4]):.0f} cycles; per fetch pass {tot[5] / tot[1]:.0f}; per shade pass {tot[7] / tot[2]:.0f}")
        ends = np.sort(t1 - t0.min()) * 10e-3
        print("  wave end times us: p10 %.0f p50 %.0f p90 %.0f p99 %.0f max %.0f" % tuple(np.percentile(ends, [10, 50, 90, 99, 100])))
        return
    t0 = stamps[:, 0].astype(np.float64)
    t1 = stamps[:, 1].astype(np.float64)
    hw = (stamps[:, 2] & np.uint64(0xffffffff)).astype(np.uint32)
    xcc = (stamps[:, 2] >> np.uint64(32)).astype(np.uint32)
    # HW_ID: wave_id[3:0] simd_id[5:4] pipe[7:6] cu_id[11:8] sh_id[12] se_id[15:13]
    cu = (hw >> 8) & 0xf
    sh = (hw >> 12) & 0x1
    se = (hw >> 13) & 0x7
    cu_key = xcc * 1000 + se * 100 + sh * 10 + cu
    start = t0.min()
    span = t1.max() - start
    dur = (t1 - t0) * 10e-3   # 100 MHz ticks -> microseconds
    print(f"kernel span {span * 10e-3:.1f} us, {len(dur)} waves, {len(np.unique(cu_key))} distinct CUs seen")
    cyc = (stamps[:, 13] - stamps[:, 12]).astype(np.float64)
    ticks = (t1 - t0)
    ok = ticks > 200   # waves resident for more than 2 us
    if ok.any():
        ghz = cyc[ok] / ticks[ok] * 0.1
        print("in-kernel clock (d s_memtime / d s_memrealtime x 100 MHz) GHz: p10 %.2f median %.2f p90 %.2f  [SHRAY_DIAG_REPS=%s]"
              % (*np.percentile(ghz, [10, 50, 90]), os.environ.get("SHRAY_DIAG_REPS", "3")))
    q = np.percentile(dur, [0, 10, 50, 90, 99, 100])
    print("wave duration us: min %.1f p10 %.1f p50 %.1f p90 %.1f p99 %.1f max %.1f; mean %.1f" % (*q, dur.mean()))
    print(f"sum of wave durations / span = {dur.sum() / (span * 10e-3):.0f} waves resident on average "
          f"({dur.sum() / (span * 10e-3) / 256:.2f} per CU)")
    slices = 20
    edges = start + np.linspace(0, span, slices + 1)
    print("slice  resident-waves(avg)  waves-started")
    for k in range(slices):
        a, b = edges[k], edges[k + 1]
        overlap = np.clip(np.minimum(t1, b) - np.maximum(t0, a), 0, None).sum() / (b - a)
        started = int(((t0 >= a) & (t0 < b)).sum())
        print(f"{k:5d}  {overlap:10.0f}  {started:8d}")
    # heavy waves: where are they in the frame and when do they start?
    order = np.argsort(-dur)[:10]
    px = ((W + 15) // 16)
    for i in order:
        blk = i // 4
        ni, li, nc, lc = (int(v) for v in stamps[i, 4:8])
        print(f"heavy wave: block ({blk % px},{blk // px}) wave {i % 4} dur {dur[i]:.1f} us start +{(t0[i] - start) * 10e-3:.1f} us; "
              f"node loop {ni} iterations x {nc / max(ni, 1):.0f} cycles, leaf loop {li} iterations x {lc / max(li, 1):.0f} cycles, "
              f"in loops {(nc + lc) / 2.4e3:.0f} us at 2.4 GHz; load wait per iteration: node {int(stamps[i, 8]) / max(ni, 1):.0f}, triangle {int(stamps[i, 9]) / max(li, 1):.0f} cycles")
    for i in order[:5]:
        print(f"  wave {i}: {int(stamps[i, 10])} leaf stages, {int(stamps[i, 5])} leaf-loop turns now, "
              f"{int(stamps[i, 11])} 64-wide rounds if triangles of all parked lanes were dealt across the wave")
    st = stamps[:, 10:12].astype(np.float64).sum(axis=0)
    print(f"all waves: {st[0]:.3g} leaf stages, {stamps[:, 5].astype(np.float64).sum():.3g} leaf-loop turns, {st[1]:.3g} 64-wide rounds")
    tot = stamps[:, 4:8].astype(np.float64).sum(axis=0)
    print(f"all waves: node loop {tot[0]:.3g} iterations x {tot[2] / tot[0]:.0f} cycles; leaf loop {tot[1]:.3g} iterations x {tot[3] / max(tot[1], 1):.0f} cycles")
    w = stamps[:, 8:10].astype(np.float64).sum(axis=0)
    print(f"all waves: load wait per iteration: node {w[0] / tot[0]:.0f} cycles, triangle {w[1] / max(tot[1], 1):.0f} cycles")
    busy = {}
    for k, d in zip(cu_key, dur):
        busy[k] = busy.get(k, 0.0) + d
    vals = np.array(sorted(busy.values()))
    print("per-CU summed wave time us: min %.0f p50 %.0f max %.0f" % (vals[0], vals[len(vals) // 2], vals[-1]))


if __name__ == "__main__":
    main()
