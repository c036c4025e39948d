#!/usr/bin/env python3
"""Renders one BASELINE configuration a few times, one launch at a time; meant to be wrapped by rocprofv3
(profiles/r04/r04_config_profile.sh).  SCENE=bunny|million WIDTH HEIGHT SPP MATERIAL REPS in the environment."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import torch
from __graft_entry__ import load_package
pkg = load_package()
million = os.environ.get("SCENE", "bunny") == "million"
world = pkg.World(pkg.scenes.million_obj() if million else pkg.scenes.bunny_trisrc())
scene = pkg.Scene(world.flatten(), pkg.scenes.environment_hdr_sky(2048), device=0)
W, H, spp = int(os.environ.get("WIDTH", "1920")), int(os.environ.get("HEIGHT", "1080")), int(os.environ.get("SPP", "1"))
material = int(os.environ.get("MATERIAL", "0"))
params = world.frame_params(W, H, material=material)
out = torch.empty(H * W * 4, dtype=torch.float32, device="cuda")
st = torch.cuda.current_stream().cuda_stream
for _ in range(3):
    scene.render_into(params, W, H, spp, out.data_ptr(), st)
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
n = int(os.environ.get("REPS", "8"))
for _ in range(n):
    scene.render_into(params, W, H, spp, out.data_ptr(), st)
b.record()
torch.cuda.synchronize()
ms = a.elapsed_time(b) / n
print(f"{'1M-triangle' if million else 'bunny-class'} {W}x{H} {spp} spp material {material}: {ms:.3f} ms/frame, {W * H * spp / ms / 1e3:.1f} Mrays/s")
