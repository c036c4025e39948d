#!/usr/bin/env python3
"""Experiment: how much of a rank's lone launch (its tiles of K frames of the bench orbit, N ranks) is the late start of
its heavy waves?  Historical: it ran against an experiment build of commit 5a6d6ea's predecessor that took an explicit patch
permutation for batch launches (-DSHRAY_EXPERIMENTS -DSHRAY_BATCH_PATCH_ORDER, since replaced by the learnt order of
capi.hip: DispatchOrder); its output is quoted in profiles/EXPERIMENTS.md R3.9.
Per-tile cost = the duration of a launch of that tile alone (its slowest wave), summed over the orbit's views; the
rank's patches are then dispatched heaviest tile first.   python profiles/heavy_first_probe.py [N=8] [K=20]"""
import ctypes as C
import os
import sys
import time

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np  # noqa: E402
import torch  # noqa: E402
from __graft_entry__ import load_package  # noqa: E402
import helpers  # noqa: E402
import bench  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 8
K = int(sys.argv[2]) if len(sys.argv) > 2 else 20
pkg = load_package()
W, H = 1920, 1080
world = pkg.World(helpers.bunny_trisrc())
scene = pkg.Scene(world.flatten(), pkg.scenes.environment_hdr_sky(2048), device=0)
orbit = bench.orbit_params(pkg, world, W, H, 0)
Nn = pkg._native
lib = Nn.load_hip() if hasattr(Nn, "load_hip") else None
streams = [torch.cuda.Stream() for _ in range(4)]
tiles_x, tiles_y = -(-W // 32), -(-H // 32)
n_tiles = tiles_x * tiles_y
rank = N - 1
owned = [t for t in range(n_tiles) if t % N == rank]


def region(tiles, batch, lanes, frames, stride_bytes, outs):
    done = j = 0
    while done < frames:
        count = min(batch, frames - done)
        views = [orbit[(done + k) % len(orbit)] for k in range(count)]
        st = streams[j % lanes]
        scene.render_batch_into(views, W, H, 1, outs[j % lanes].data_ptr(), stride_bytes, st.cuda_stream, tiles)
        done += count
        j += 1


def measure(tiles, batch, lanes, frames, stride_bytes):
    outs = [torch.empty(max(batch, 1) * stride_bytes // 4, dtype=torch.float32, device="cuda") for _ in range(lanes)]
    for _ in range(3):
        region(tiles, batch, lanes, frames, stride_bytes, outs)
    torch.cuda.synchronize()
    t = []
    for _ in range(15):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        region(tiles, batch, lanes, frames, stride_bytes, outs)
        torch.cuda.synchronize()
        t.append(time.perf_counter() - t0)
    return sorted(t)[len(t) // 2] * 1e3


for _ in range(200):
    scene.render_into(orbit[0], W, H, 1, torch.empty(W * H * 4, device="cuda").data_ptr(), streams[0].cuda_stream, None)
torch.cuda.synchronize()

# cost of each owned tile: a launch of that tile alone, every view of the orbit
cost = np.zeros(len(owned))
one = torch.empty(32 * 32 * 4, dtype=torch.float32, device="cuda")
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
t0 = time.time()
for i, t in enumerate(owned):
    ts = Nn.TileSet(32, 32, n_tiles, t, 1)
    for v in orbit[:: max(1, len(orbit) // 5)]:
        a.record()
        scene.render_into(v, W, H, 1, one.data_ptr(), torch.cuda.current_stream().cuda_stream, ts)
        b.record()
        b.synchronize()
        cost[i] += a.elapsed_time(b)
print(f"{len(owned)} tiles costed in {time.time() - t0:.1f} s: per-tile ms (5 views) min {cost.min():.3f} median {np.median(cost):.3f} "
      f"p90 {np.quantile(cost, .9):.3f} max {cost.max():.3f}")

tiles = Nn.TileSet(32, 32, N, rank, 1)
stride = pkg.tracer.tile_buffer_bytes(W, H, tiles)
shapes = [(min(K, 64), 1), (min(-(-K // 2), 64), 2), (min(4 * N, 64), 4)]
base = {s: measure(tiles, s[0], s[1], K, stride) for s in shapes}
# the k-th owned tile holds patches 4k .. 4k + 3 (a 32x32 tile is 2x2 patches of 16x16); edge tiles too (clipped patches exist)
patches_per_tile = 4
for name, key in (("heaviest first", -cost), ("lightest first", cost), ("heavy tiles spread evenly", None)):
    if key is None:
        heavy = list(np.argsort(-cost))
        cut = max(1, len(heavy) // 8)
        hv, rest = heavy[:cut], sorted(heavy[cut:])
        step = max(1, len(rest) // len(hv))
        order_tiles = []
        for i, h in enumerate(hv):
            order_tiles.append(h)
            order_tiles.extend(rest[i * step:(i + 1) * step])
        order_tiles.extend(rest[len(hv) * step:])
    else:
        order_tiles = list(np.argsort(key, kind="stable"))
    order = np.array([patches_per_tile * k + q for k in order_tiles for q in range(patches_per_tile)], dtype=np.uint32)
    total = len(owned) * patches_per_tile
    assert sorted(order.tolist()) == list(range(total))
    rc = scene._lib.shray_debug_set_patch_order(scene._handle, order.ctypes.data_as(C.c_void_p), C.c_uint32(total))
    assert rc == 0, rc
    for s in shapes:
        ms = measure(tiles, s[0], s[1], K, stride)
        print(f"  {name:28s} {s[0]:3d} frames per launch x {s[1]} streams: {ms:.3f} ms (row-major {base[s]:.3f})")
scene._lib.shray_debug_set_patch_order(scene._handle, None, C.c_uint32(0))
