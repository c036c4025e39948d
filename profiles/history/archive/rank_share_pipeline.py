#!/usr/bin/env python3
"""One GPU playing ONE rank of an N-rank job: renders only that rank's interleaved tiles
(no gather) with 1..8 frames in flight.  Shows how deep the frame pipeline must be before a
rank's share of the frame is throughput- rather than latency-bound (the frame's heavy waves
take ~0.25 ms however few tiles a rank owns)."""
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
tile = int(os.environ.get("TILE", multigpu.DEFAULT_TILE))
LANES = [int(x) for x in os.environ.get("LANES", "1,2,4,8").split(",")]
for nranks in [int(x) for x in os.environ.get("NRANKS", "8,4,2").split(",")]:
    for rank in (range(nranks) if os.environ.get("ALL_RANKS") else ((0, 3) if nranks == 8 else (0,))):
        for lanes in LANES:
            streams = [torch.cuda.Stream() for _ in range(lanes)]
            per_rank = multigpu.max_tiles_per_rank(W, H, tile, tile, nranks)
            outs = [torch.zeros(per_rank * tile * tile * 4, dtype=torch.float32, device="cuda") for _ in range(lanes)]
            tiles = N.TileSet(tile, tile, nranks, rank)
            def step(k):
                scene.render_into(params, W, H, 1, outs[k % lanes].data_ptr(), streams[k % lanes].cuda_stream, tiles)
            for k in range(16):
                step(k)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for k in range(400):
                step(k)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / 400
            print(f"rank {rank} of {nranks}: {lanes} frames in flight: {dt * 1e3:.4f} ms/frame", flush=True)
