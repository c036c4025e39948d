#!/usr/bin/env python3
"""Times the BASELINE.json configurations other than the headline one on one MI355X
(kernel 0, scene and environment resident, HIP events on the launch stream).  Writes
profiles/<tag>_configs.json.  Usage: python profiles/run_configs.py [tag]"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import torch
from __graft_entry__ import load_package

tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
KERNEL = int(sys.argv[2]) if len(sys.argv) > 2 else 0
ONLY = sys.argv[3].split(",") if len(sys.argv) > 3 else None   # e.g. "2,4": config numbers to run
pkg = load_package()
env = pkg.scenes.environment_hdr_sky(2048)
stream = torch.cuda.current_stream().cuda_stream
results = []


def run(name, world, scene, params, W, H, spp, reps):
    if ONLY and name.split(":")[0].split()[0] not in ONLY:
        return
    scene.set_kernel(KERNEL)
    out = torch.empty(H * W * 4, dtype=torch.float32, device="cuda")
    for _ in range(2):
        scene.render_into(params, W, H, spp, out.data_ptr(), stream)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        scene.render_into(params, W, H, spp, out.data_ptr(), stream)
    b.record()
    torch.cuda.synchronize()
    ms = a.elapsed_time(b) / reps
    _, counters = scene.render_counters(params, W, H, spp, want_image=False)
    algo = pkg.tracer.algorithmic_bytes(counters, W * H)
    row = {"config": name, "kernel": KERNEL, "width": W, "height": H, "spp": spp, "ms_per_frame": round(ms, 4),
           "mrays_per_s": round(W * H * spp / ms / 1e3, 1), "algorithmic_gb_per_s": round(algo / ms / 1e6, 1),
           "bytes_per_ray": round(algo / (W * H * spp), 1), "bad_hit_fraction": counters["bad_hits"] / counters["samples"],
           "counters": counters}
    print(json.dumps({k: v for k, v in row.items() if k != "counters"}), flush=True)
    results.append(row)


t0 = time.time()
bunny = pkg.World(pkg.scenes.bunny_trisrc())
scene = pkg.Scene(bunny.flatten(), env, device=0)
run("1: 256x256 primary rays only (parity anchor)", bunny, scene, (lambda p: (setattr(p, "bounce_count", 1), p)[1])(bunny.frame_params(256, 256, material=0)), 256, 256, 1, 200)
run("2: 1920x1080 1 spp gold (headline)", bunny, scene, bunny.frame_params(1920, 1080, material=0), 1920, 1080, 1, 100)
run("3: 1920x1080 64 spp glazed plaster", bunny, scene, bunny.frame_params(1920, 1080, material=6), 1920, 1080, 64, 3)
run("5 (one GPU's view): 3840x2160 16 spp gold, whole frame on one GPU", bunny, scene, bunny.frame_params(3840, 2160, material=0), 3840, 2160, 16, 3)
scene.close()
big = pkg.World(pkg.scenes.million_obj())
print("1M-triangle scene loaded: %d triangles, %d nodes, depth %d (%.1f s since start)" % (
    big.triangle_count, big.info.node_count, big.info.max_level, time.time() - t0), flush=True)
scene = pkg.Scene(big.flatten(), env, device=0)
run("4: 1M-triangle OBJ 1920x1080 4 spp gold", big, scene, big.frame_params(1920, 1080, material=0), 1920, 1080, 4, 5)
json.dump(results, open(os.path.join(ROOT, "gpurun_out", f"{tag}_configs_k{KERNEL}.json"), "w"), indent=1)
