#!/usr/bin/env python3
"""Single-GPU breakdown of the per-frame cost of the multi-GPU code path (one rank):
tile-mode render only / + assemble / + NCCL gather, with 1 and 2 frames in flight."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
import torch, torch.distributed as dist
from __graft_entry__ import load_package
import helpers
pkg = load_package()
from shader_ray_amd import multigpu
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
W, H = 1920, 1080
world = pkg.World(helpers.bunny_trisrc())
scene = pkg.Scene(world.flatten(), pkg.scenes.environment_hdr_sky(2048), device=0)
params = world.frame_params(W, H, material=0)


def run(label, lanes, gather, assemble, tile=32):
    streams = [torch.cuda.current_stream()] + [torch.cuda.Stream() for _ in range(lanes - 1)]
    frames = [multigpu.DistributedFrame(W, H, tile, tile, device="cuda", always_gather=gather) for _ in range(lanes)]

    def step(k):
        lane = k % lanes
        st = streams[lane]
        f = frames[lane]
        with torch.cuda.stream(st):
            scene.render_into(params, W, H, 1, f.mine.data_ptr(), st.cuda_stream, f.tiles)
            if gather:
                dist.gather(f.mine, f.sink, dst=0)
            if assemble:
                src = f.received if gather else f.mine
                multigpu.assemble_tiles_torch(src.view(1, f.per_rank, tile, tile, 4), W, H, tile, tile)
    for k in range(20):
        step(k)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(200):
        step(k)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 200
    print(f"{label:55s} {dt * 1e3:.3f} ms/frame")


for lanes in (1, 2):
    run(f"{lanes} in flight: tile-mode render only", lanes, False, False)
    run(f"{lanes} in flight: render + assemble", lanes, False, True)
    run(f"{lanes} in flight: render + gather", lanes, True, False)
    run(f"{lanes} in flight: render + gather + assemble", lanes, True, True)
run("2 in flight: render only, 16x16 tiles", 2, False, False, tile=16)
dist.destroy_process_group()
