#!/usr/bin/env python3
"""Turnaround of the host flattener (get_shader_data, threaded) against the GPU flattener (shray_flatten_device,
incl. its uploads) on the benchmark scenes.  usage: python profiles/flatten_probe.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
from __graft_entry__ import load_package
import helpers
pkg = load_package()
for name, path in (("bunny-class 69k", helpers.bunny_trisrc()), ("1M-triangle OBJ", helpers.million_obj())):
    t0 = time.perf_counter(); world = pkg.World(path); t_load = time.perf_counter() - t0
    t0 = time.perf_counter(); world.flatten(); t_host = time.perf_counter() - t0
    t0 = time.perf_counter(); tree = world.export_tree(); t_export = time.perf_counter() - t0
    pkg.tracer.DeviceFlat(tree).close()   # warm-up (context, code objects)
    t0 = time.perf_counter(); flat = pkg.tracer.DeviceFlat(tree); t_dev = time.perf_counter() - t0
    t0 = time.perf_counter(); flat.download(); t_down = time.perf_counter() - t0
    print(f"{name}: load_world {t_load:.3f} s | host get_shader_data {t_host*1e3:.1f} ms | export tree {t_export*1e3:.1f} ms + "
          f"shray_flatten_device {t_dev*1e3:.1f} ms (+ download {t_down*1e3:.1f} ms)", flush=True)
