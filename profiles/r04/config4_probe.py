#!/usr/bin/env python3
"""Runs BASELINE config 4 (1M-triangle OBJ, 1920x1080, 4 spp) a few times; meant to be wrapped by
rocprofv3 (profiles/run_profile.sh style) to read cache counters for the deep-tree scene."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import torch
from __graft_entry__ import load_package
pkg = load_package()
world = pkg.World(pkg.scenes.million_obj())
scene = pkg.Scene(world.flatten(), pkg.scenes.environment_hdr_sky(2048), device=0)
W, H, spp = 1920, 1080, int(os.environ.get("SPP", "4"))
params = world.frame_params(W, H, material=int(os.environ.get("MATERIAL", "0")))
out = torch.empty(H * W * 4, dtype=torch.float32, device="cuda")
st = torch.cuda.current_stream().cuda_stream
for _ in range(2):
    scene.render_into(params, W, H, spp, out.data_ptr(), st)
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
n = int(os.environ.get("REPS", "5"))
for _ in range(n):
    scene.render_into(params, W, H, spp, out.data_ptr(), st)
b.record()
torch.cuda.synchronize()
ms = a.elapsed_time(b) / n
print(f"config 4: {ms:.3f} ms/frame, {W * H * spp / ms / 1e3:.1f} Mrays/s")
