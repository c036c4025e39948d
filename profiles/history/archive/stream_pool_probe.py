#!/usr/bin/env python3
"""Does the rate of an 8-deep frame pipeline depend on WHICH torch pool streams carry it?
Same work (rank 3 of 8's tiles) repeated; each repetition takes the next 8 streams of torch's pool."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import torch
from __graft_entry__ import load_package
import helpers
pkg = load_package()
from shader_ray_amd import multigpu, _native as N
W, H, tile, nranks, rank, lanes = 1920, 1080, 32, 8, 3, int(os.environ.get("LANES", "8"))
world = pkg.World(helpers.bunny_trisrc())
scene = pkg.Scene(world.flatten(), pkg.scenes.environment_hdr_sky(2048), device=0)
params = world.frame_params(W, H, material=0)
per_rank = multigpu.max_tiles_per_rank(W, H, tile, tile, nranks)
outs = [torch.zeros(per_rank * tile * tile * 4, dtype=torch.float32, device="cuda") for _ in range(lanes)]
tiles = N.TileSet(tile, tile, nranks, rank)
for rep in range(10):
    streams = [torch.cuda.Stream() for _ in range(lanes)]
    def step(k):
        scene.render_into(params, W, H, 1, outs[k % lanes].data_ptr(), streams[k % lanes].cuda_stream, tiles)
    for k in range(16):
        step(k)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(400):
        step(k)
    torch.cuda.synchronize()
    print(f"rep {rep}: streams {[hex(s.cuda_stream)[-5:] for s in streams][:3]}...: {(time.perf_counter() - t0) / 400 * 1e3:.4f} ms/frame", flush=True)
