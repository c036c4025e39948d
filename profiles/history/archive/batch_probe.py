#!/usr/bin/env python3
"""Frames per launch (shray_render_batch_device): ms/frame for batch sizes and stream counts, for the
whole frame and for one rank's share of an 8-rank job; checks the batch against single launches."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import torch
from __graft_entry__ import load_package
import helpers
pkg = load_package()
from shader_ray_amd import multigpu, _native as N
W, H, tile = 1920, 1080, 32
world = pkg.World(helpers.bunny_trisrc())
scene = pkg.Scene(world.flatten(), pkg.scenes.environment_hdr_sky(2048), device=0)
params = world.frame_params(W, H, material=0)
default, side = torch.cuda.current_stream(), torch.cuda.Stream()

single = torch.empty(H * W * 4, dtype=torch.float32, device="cuda")
scene.render_into(params, W, H, 1, single.data_ptr(), default.cuda_stream)
three = torch.zeros(3, H * W * 4, dtype=torch.float32, device="cuda")
scene.render_batch_into([params] * 3, W, H, 1, three.data_ptr(), H * W * 16, default.cuda_stream)
torch.cuda.synchronize()
print("batch of 3 equals the single frame:", all(bool(torch.equal(three[k], single)) for k in range(3)), flush=True)

for nranks in (1, 8, 4, 2):
    tiles = None if nranks == 1 else N.TileSet(tile, tile, nranks, 0)
    floats = H * W * 4 if nranks == 1 else multigpu.max_tiles_per_rank(W, H, tile, tile, nranks) * tile * tile * 4
    for batch in (1, 2, 4, 8, 16):
        for nstreams in (1, 2):
            streams = [default, side][:nstreams]
            outs = [torch.zeros(batch, floats, dtype=torch.float32, device="cuda") for _ in streams]
            def step(k):
                scene.render_batch_into([params] * batch, W, H, 1, outs[k % nstreams].data_ptr(), floats * 4,
                                        streams[k % nstreams].cuda_stream, tiles)
            launches = max(8, 256 // batch)
            for k in range(4):
                step(k)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for k in range(launches):
                step(k)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / (launches * batch)
            print(f"share 1/{nranks}: {batch:2d} frames per launch, {nstreams} stream(s): {dt * 1e3:.4f} ms/frame", flush=True)
