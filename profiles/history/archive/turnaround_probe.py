#!/usr/bin/env python3
"""Scene turnaround: seconds for each step from a model file to the first frame."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import torch
from __graft_entry__ import load_package
import helpers
pkg = load_package()
torch.cuda.init(); torch.zeros(1, device="cuda")
for name, path in (("bunny-class trisrc", helpers.bunny_trisrc()), ("1M-triangle obj", helpers.million_obj())):
    for rep in range(2):
        t0 = time.perf_counter(); world = pkg.World(path)
        t1 = time.perf_counter(); desc = world.flatten()
        t2 = time.perf_counter(); scene = pkg.Scene(desc, None, device=0)
        t3 = time.perf_counter(); scene.set_environment(pkg.scenes.environment_constant()) if hasattr(scene, "set_environment") else None
        t4 = time.perf_counter()
        print(f"{name}: load_world {t1 - t0:.3f} s  get_shader_data {t2 - t1:.3f} s  shray_scene_create {t3 - t2:.3f} s", flush=True)
        del scene
