#!/usr/bin/env python3
"""One GPU playing each rank of an 8-rank job in turn (8 frames per launch, two streams, render only):
how evenly do interleaved tiles of a given size split the frame's work?"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import torch
from __graft_entry__ import load_package
import helpers
pkg = load_package()
from shader_ray_amd import multigpu, _native as N
W, H = 1920, 1080
world = pkg.World(helpers.bunny_trisrc())
scene = pkg.Scene(world.flatten(), pkg.scenes.environment_hdr_sky(2048), device=0)
params = world.frame_params(W, H, material=0)
streams = [torch.cuda.current_stream(), torch.cuda.Stream()]
nranks = int(os.environ.get("NRANKS", "8"))
batch = int(os.environ.get("BATCH", str(nranks)))
for tw, th in ((32, 32), (16, 16), (64, 64), (128, 16), (16, 128)):
    floats = multigpu.max_tiles_per_rank(W, H, tw, th, nranks) * tw * th * 4
    outs = [torch.zeros(batch, floats, dtype=torch.float32, device="cuda") for _ in streams]
    times = []
    for rank in range(nranks):
        tiles = N.TileSet(tw, th, nranks, rank)
        def step(k):
            scene.render_batch_into([params] * batch, W, H, 1, outs[k % 2].data_ptr(), floats * 4, streams[k % 2].cuda_stream, tiles)
        for k in range(4):
            step(k)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for k in range(24):
            step(k)
        torch.cuda.synchronize()
        times.append((time.perf_counter() - t0) / (24 * batch) * 1e3)
    print(f"tiles {tw}x{th}: per-rank ms/frame " + " ".join(f"{t:.4f}" for t in times) + f"  max {max(times):.4f} mean {sum(times) / len(times):.4f}", flush=True)
