#!/usr/bin/env python3
"""How much of a multi-sample frame is its tail?  Times frames of a configuration one at a time on one stream and
overlapped on four streams (the bulk of one frame fills the tail of another).
usage: python3 profiles/tail_probe.py <config 2|3|4> <spp> [frames]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import torch
from __graft_entry__ import load_package
import helpers
pkg = load_package()
cfg, spp = int(sys.argv[1]), int(sys.argv[2])
frames = int(sys.argv[3]) if len(sys.argv) > 3 else 12
W, H = 1920, 1080
if cfg == 4:
    world, material = pkg.World(helpers.million_obj()), 0
else:
    world, material = pkg.World(helpers.bunny_trisrc()), (6 if cfg == 3 else 0)
scene = pkg.Scene(world.flatten(), pkg.scenes.environment_hdr_sky(2048), device=0)
params = world.frame_params(W, H, material=material)
streams = [torch.cuda.Stream() for _ in range(4)]
outs = [torch.empty(H * W * 4, dtype=torch.float32, device="cuda") for _ in range(4)]
for lanes in (1, 4):
    for rep in range(2):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for k in range(frames):
            scene.render_into(params, W, H, spp, outs[k % lanes].data_ptr(), streams[k % lanes].cuda_stream)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / frames
    print(f"config {cfg} {spp} spp, {lanes} stream(s): {dt * 1e3:.3f} ms/frame  {W * H * spp / dt / 1e6:.0f} Mrays/s", flush=True)
