#!/usr/bin/env python3
"""Config 3 (glazed plaster: diffuse + shadow rays) at 8 spp, for A/B of library builds (SHRAY_HIP_LIB)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import torch
from __graft_entry__ import load_package
import helpers
pkg = load_package()
W, H, spp = 1920, 1080, 8
world = pkg.World(helpers.bunny_trisrc())
scene = pkg.Scene(world.flatten(), pkg.scenes.environment_hdr_sky(2048), device=0)
params = world.frame_params(W, H, material=6)
out = torch.empty(H * W * 4, dtype=torch.float32, device="cuda")
st = torch.cuda.current_stream().cuda_stream
for _ in range(2):
    scene.render_into(params, W, H, spp, out.data_ptr(), st)
torch.cuda.synchronize()
ts = []
for _ in range(8):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); scene.render_into(params, W, H, spp, out.data_ptr(), st); b.record(); torch.cuda.synchronize()
    ts.append(a.elapsed_time(b))
ts.sort()
print(f"{os.path.basename(os.environ.get('SHRAY_HIP_LIB', 'shipped')):28s} plaster {spp} spp: median {ts[len(ts) // 2]:.3f} ms  {W * H * spp / ts[len(ts) // 2] / 1e3:.1f} Mrays/s", flush=True)
