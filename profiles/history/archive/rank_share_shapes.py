#!/usr/bin/env python3
"""How long ONE rank of N needs for its share of K frames of the bench orbit, by launch shape (frames per launch x
streams) -- the compute side of bench.py --gpus N, measured on one GPU (no exchange: the tiles are rendered and dropped).
Ideal = the N = 1 loop's time for K frames / N.   python profiles/rank_share_shapes.py [N=8] [K=20]"""
import os
import sys
import time

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import torch  # noqa: E402
from __graft_entry__ import load_package  # noqa: E402
import helpers  # noqa: E402

import bench  # noqa: E402  (the orbit)

N = int(sys.argv[1]) if len(sys.argv) > 1 else 8
K = int(sys.argv[2]) if len(sys.argv) > 2 else 20
pkg = load_package()
W, H = 1920, 1080
world = pkg.World(helpers.bunny_trisrc())
scene = pkg.Scene(world.flatten(), pkg.scenes.environment_hdr_sky(2048), device=0)
orbit = bench.orbit_params(pkg, world, W, H, 0)
Nn = pkg._native
streams = [torch.cuda.Stream() for _ in range(4)]


def region(tiles, batch, lanes, frames, stride_bytes, outs):
    done = j = 0
    while done < frames:
        count = min(batch, frames - done)
        views = [orbit[(done + k) % len(orbit)] for k in range(count)]
        st = streams[j % lanes]
        if count == 1:
            scene.render_into(views[0], W, H, 1, outs[j % lanes].data_ptr(), st.cuda_stream, tiles)
        else:
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


# warm the clock
for _ in range(200):
    scene.render_into(orbit[0], W, H, 1, torch.empty(W * H * 4, device="cuda").data_ptr(), streams[0].cuda_stream, None)
torch.cuda.synchronize()
whole = measure(None, 2, 4, K, W * H * 16)
print(f"N = 1 loop (2 frames per launch x 4 streams), {K} frames: {whole:.3f} ms; ideal per rank at N = {N}: {whole / N:.3f} ms")
tiles = Nn.TileSet(32, 32, N, N - 1, 1)
stride = pkg.tracer.tile_buffer_bytes(W, H, tiles)
custom = [(int(b), 4) for b in os.environ["SHAPES"].split(",")] if os.environ.get("SHAPES") else None   # SHAPES=8,16,20: frames per launch
for batch, lanes in custom or sorted({(N, 2), (2 * N, 2), (2 * N, 4), (3 * N, 4), (4 * N, 4), (4 * N, 2), (-(-K // 4), 4), (-(-K // 2), 2), (-(-K // 3), 3), (N, 4), (max(1, N // 2), 4)}):
    if batch > 64:
        continue
    ms = measure(tiles, batch, lanes, K, stride)
    print(f"  rank {N - 1} of {N}: {batch:3d} frames per launch x {lanes} streams: {ms:.3f} ms  -> {whole / ms:.2f} x one GPU (compute only)")
