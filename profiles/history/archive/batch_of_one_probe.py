#!/usr/bin/env python3
"""One frame per launch through the by-value-FrameView kernel (shray_render_device) and through the
FrameView-in-memory batch kernel (a -DSHRAY_BATCH_ALWAYS build runs batches of one through it)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import torch
from __graft_entry__ import load_package
import helpers
pkg = load_package()
W, H = 1920, 1080
world = pkg.World(helpers.bunny_trisrc())
scene = pkg.Scene(world.flatten(), pkg.scenes.environment_hdr_sky(2048), device=0)
params = world.frame_params(W, H, material=0)
streams = [torch.cuda.current_stream(), torch.cuda.Stream()]
outs = [torch.empty(H * W * 4, dtype=torch.float32, device="cuda") for _ in streams]
plain = lambda k: scene.render_into(params, W, H, 1, outs[k % 2].data_ptr(), streams[k % 2].cuda_stream)
batch = lambda k: scene.render_batch_into([params], W, H, 1, outs[k % 2].data_ptr(), H * W * 16, streams[k % 2].cuda_stream)
for label, fn in (("render_device", plain), ("batch of one", batch), ("render_device", plain), ("batch of one", batch)):
    for k in range(6): fn(k)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for k in range(200): fn(k)
    torch.cuda.synchronize(); print(f"{label:14s} two in flight: {(time.perf_counter() - t0) / 200 * 1e3:.4f} ms/frame", flush=True)
