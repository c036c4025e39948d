#!/usr/bin/env python3
"""One GPU playing RANK 0 of an 8-rank job, without the network: per step it renders its share of
`batch` frames in one launch, packs RGB, receives (stand-in: 7 device-to-device copies of a peer-sized
buffer) and de-interleaves `batch` frames.  Shows what rank 0's GPU must sustain per frame besides the
gather itself."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import torch
from __graft_entry__ import load_package
import helpers
pkg = load_package()
from shader_ray_amd import multigpu, _native as N
W, H, tile, world = 1920, 1080, 32, 8
w = pkg.World(helpers.bunny_trisrc())
scene = pkg.Scene(w.flatten(), pkg.scenes.environment_hdr_sky(2048), device=0)
params = w.frame_params(W, H, material=0)
for batch in (8, 16):
    lanes = 2
    streams = [torch.cuda.current_stream(), torch.cuda.Stream()]
    splits = []
    for _ in range(lanes):
        s = multigpu.DistributedFrame(W, H, tile, tile, device="cuda", frames=batch)
        # lay the object out as rank 0 of 8 (no process group here)
        s.world, s.per_rank = world, multigpu.max_tiles_per_rank(W, H, tile, tile, world)
        s.pixels = s.per_rank * tile * tile
        s.mine = torch.zeros(batch, s.pixels * 4, device="cuda")
        s.wire = torch.zeros(batch, s.pixels * 3, device="cuda")
        s.tiles = N.TileSet(tile, tile, world, 0)
        s.received = torch.rand(world, batch, s.pixels * 3, device="cuda")
        s.output = torch.ones(batch, H, W, 4, device="cuda")
        splits.append(s)
    peer = torch.rand(batch, splits[0].pixels * 3, device="cuda")

    def step(j, parts):
        s, st = splits[j % lanes], streams[j % lanes]
        with torch.cuda.stream(st):
            if "render" in parts:
                scene.render_batch_into([params] * batch, W, H, 1, s.mine.data_ptr(), s.frame_stride_bytes, st.cuda_stream, s.tiles)
            if "pack" in parts:
                s.wire.view(batch, s.pixels, 3).copy_(s.mine.view(batch, s.pixels, 4)[:, :, :3])
            if "recv" in parts:
                for r in range(1, world):
                    s.received[r].copy_(peer)
            if "assemble" in parts:
                s._assemble(s.received, batch)

    for parts in (("render",), ("render", "pack"), ("render", "pack", "assemble"), ("render", "pack", "recv", "assemble"), ("assemble",)):
        for j in range(4):
            step(j, parts)
        torch.cuda.synchronize()
        n = 40
        t0 = time.perf_counter()
        for j in range(n):
            step(j, parts)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / (n * batch)
        print(f"batch {batch:2d}: {'+'.join(parts):32s} {dt * 1e3:.4f} ms/frame", flush=True)
