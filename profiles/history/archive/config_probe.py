#!/usr/bin/env python3
"""Times one BASELINE config with each kernel and prints the work counters per ray.
usage: python3 profiles/config_probe.py [config=4] [spp] ; env KERNELS=0,1,2 LIB=<alternate libshray_hip .so>"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import torch
from __graft_entry__ import load_package
import helpers
pkg = load_package()
cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 4
W, H = 1920, 1080
if cfg == 4:
    world, env, material, spp = pkg.World(helpers.million_obj()), pkg.scenes.environment_hdr_sky(2048), 0, 4
elif cfg == 3:
    world, env, material, spp = pkg.World(helpers.bunny_trisrc()), pkg.scenes.environment_hdr_sky(2048), 6, 4
else:
    world, env, material, spp = pkg.World(helpers.bunny_trisrc()), pkg.scenes.environment_constant(), 0, 1
if len(sys.argv) > 2:
    spp = int(sys.argv[2])
scene = pkg.Scene(world.flatten(), env, device=0)
params = world.frame_params(W, H, material=material)
out = torch.empty(H * W * 4, dtype=torch.float32, device="cuda")
st = torch.cuda.current_stream().cuda_stream
_, c = scene.render_counters(params, W, H, spp, want_image=False)
n = c["samples"]
print(f"config {cfg} spp {spp}: per sample: traversals {c['traversals'] / n:.2f} node visits {c['node_visits'] / n:.1f} "
      f"leaves {c['leaf_visits'] / n:.2f} tri tests {c['triangle_tests'] / n:.2f} hits {c['shaded_hits'] / n:.2f} bad {c['bad_hits']}")
for k in [int(x) for x in os.environ.get("KERNELS", "0,1,2").split(",")]:
    scene.set_kernel(k)
    for _ in range(2):
        scene.render_into(params, W, H, spp, out.data_ptr(), st)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 5
    a.record()
    for _ in range(reps):
        scene.render_into(params, W, H, spp, out.data_ptr(), st)
    b.record(); torch.cuda.synchronize()
    ms = a.elapsed_time(b) / reps
    print(f"  kernel {k}: {ms:.3f} ms/frame  {W * H * spp / ms / 1e3:.1f} Mrays/s", flush=True)
