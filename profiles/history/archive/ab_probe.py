#!/usr/bin/env python3
"""A/B of library builds on the benchmark frame (config 2 with the HDR sky, as bench.py): for the library
named by SHRAY_HIP_LIB (or the shipped one) prints one-frame-at-a-time median ms and two-streams
pipelined ms per frame; optional argument 4 = the 1M-triangle scene (one frame at a time only)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import torch
from __graft_entry__ import load_package
import helpers
pkg = load_package()
cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 2
W, H = 1920, 1080
world = pkg.World(helpers.million_obj() if cfg == 4 else helpers.bunny_trisrc())
scene = pkg.Scene(world.flatten(), pkg.scenes.environment_hdr_sky(2048), device=0)
params = world.frame_params(W, H, material=0)
streams = [torch.cuda.current_stream(), torch.cuda.Stream()]
outs = [torch.empty(H * W * 4, dtype=torch.float32, device="cuda") for _ in streams]
for _ in range(5):
    scene.render_into(params, W, H, 1, outs[0].data_ptr(), streams[0].cuda_stream)
torch.cuda.synchronize()
singles = []
for _ in range(40 if cfg == 2 else 12):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); scene.render_into(params, W, H, 1, outs[0].data_ptr(), streams[0].cuda_stream); b.record()
    torch.cuda.synchronize(); singles.append(a.elapsed_time(b))
singles.sort()
line = f"{os.path.basename(os.environ.get('SHRAY_HIP_LIB', 'shipped')):28s} one at a time: median {singles[len(singles) // 2]:.4f} min {singles[0]:.4f} ms"
if cfg == 2:
    best = 1e9
    for rep in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for k in range(200):
            scene.render_into(params, W, H, 1, outs[k % 2].data_ptr(), streams[k % 2].cuda_stream)
        torch.cuda.synchronize(); best = min(best, (time.perf_counter() - t0) / 200 * 1e3)
    line += f"; two in flight: {best:.4f} ms/frame"
print(line, flush=True)
