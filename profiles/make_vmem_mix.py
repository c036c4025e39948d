#!/usr/bin/env python3
"""Measures the class mix bench.py prices the kernel's vector-memory instructions with (roofline.vmem), on the box, over the orbit:
    make -C shader-ray_amd variant VARIANT=uniform HIP_EXTRA="-DSHRAY_DIAGNOSTICS -DSHRAY_DIAG_UNIFORM"
    make -C shader-ray_amd variant VARIANT=khist2 HIP_EXTRA="-DSHRAY_DIAGNOSTICS -DSHRAY_DIAG_KHIST=2 -DSHRAY_DIAG_KHIST_FROM=0"
    python profiles/make_vmem_mix.py <SQ_INSTS_VMEM_RD per frame> [out.json]
Two diagnostic builds of THESE sources render the 20 views of bench.py's orbit: how many distinct records the walking lanes of a
node visit are at (variants/diag_uniform_visit.inc), how many distinct leaves the parked lanes of a leaf stage are in, by rounds
(variants/diag_khist_stage.inc).  The file is keyed by the hash of the PRODUCT library's device code (profiles/buildhash.py): the
build whose instructions the mix describes."""
import ctypes as C
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import ctypes as C, json, os, sys
import numpy as np
sys.path[:0] = [%r, %r]
import torch
from __graft_entry__ import load_package
import bench
pkg = load_package()
N = pkg._native
N.HIP_LIB = os.environ["SHRAY_DIAG_LIB"]
lib = N.load_hip()
lib.shray_debug_timeline.restype = C.c_int
world = pkg.World(pkg.scenes.bunny_trisrc())
scene = pkg.Scene(world.flatten(), pkg.scenes.environment_hdr_sky(2048), device=0)
W, H = 1920, 1080
total = np.zeros(8)
packed = []
for params in bench.orbit_params(pkg, world, W, H, 0):
    patches = ((W + 15) // 16) * ((H + 15) // 16)
    stamps = np.zeros((patches * 4, 16), dtype=np.uint64)
    N.check(lib.shray_debug_timeline(scene._handle, C.byref(params), W, H, 1, stamps.ctypes.data_as(C.c_void_p)))
    t = stamps[:, 4:12]
    if os.environ.get("SHRAY_MIX_KIND") == "leaves":
        # words {stages, bits 0-23; rounds, bits 24-43}: bins D = 1, 2, 3, 4, 5-8, 9-16, > 16
        total[:7] += np.array([((t[:, k] >> np.uint64(24)) & np.uint64(0xfffff)).astype(np.float64).sum() for k in range(7)])
    else:
        total += t.astype(np.float64).sum(axis=0)
print(json.dumps([float(x) / 20.0 for x in total]))
''' % (ROOT, os.path.join(ROOT, "tests"))


def run(lib, kind):
    env = dict(os.environ, SHRAY_DIAG_LIB=os.path.join(ROOT, "shader-ray_amd", "_variants", lib), SHRAY_MIX_KIND=kind)
    out = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True, check=True).stdout
    return json.loads(out.strip().splitlines()[-1])


def main():
    vmem_per_frame = float(sys.argv[1])
    out_path = sys.argv[2] if len(sys.argv) > 2 else os.path.join(ROOT, "gpurun_out", "vmem_class_mix.json")
    visits, uniform, _, _, _, two, three_four, _ = run("libshray_hip_uniform.so", "nodes")
    rounds = run("libshray_hip_khist2.so", "leaves")[:7]
    more = visits - uniform - two - three_four
    node_insts = 2.0 * (visits - uniform)            # two 16-byte loads per wave-visit the scalar cache does not serve
    tri_insts = 3.0 * sum(rounds)                     # three loads per round of a leaf stage
    other = max(0.0, vmem_per_frame - node_insts - tri_insts)
    r = [x / sum(rounds) for x in rounds]
    sys.path.insert(0, os.path.join(ROOT, "profiles"))
    from buildhash import kernel_source_hash
    mix = {
        "what": "How many distinct records the lanes of one vector-memory wave-instruction of the headline kernel read (BASELINE configs[1]: "
                "1920x1080, 1 spp, gold, the 20 views of the orbit), by kind of fetch; bench.py prices SQ_INSTS_VMEM_RD with it "
                "(roofline.vmem.peak = 1 / sum(share x cost), each class's cost probed in the bench run).",
        "build_hash": kernel_source_hash(),
        "build_hash_is": "the product library's device code when the histograms were taken (two diagnostic builds of the same sources)",
        "measured": {"wave_visits_per_frame": visits, "uniform": uniform / visits, "two_records": two / visits, "three_or_four": three_four / visits,
                     "more": more / visits, "leaf_rounds_per_frame": sum(rounds), "rounds_by_distinct_leaves_1_2_3_4_5to8_9to16_more": r,
                     "vmem_insts_per_frame": vmem_per_frame},
        "kinds": [
            {"name": "node_fetch", "share_of_insts": round(node_insts / vmem_per_frame, 4), "probe_bytes_per_lane": 32,
             "records_per_instruction": {"2": round(two / (visits - uniform), 4), "4": round(three_four / (visits - uniform), 4),
                                         "8": round(more / (visits - uniform), 4)}},
            {"name": "triangle_fetch", "share_of_insts": round(tri_insts / vmem_per_frame, 4), "probe_bytes_per_lane": 12,
             "records_per_instruction": {"1": round(r[0], 4), "2": round(r[1], 4), "4": round(r[2] + r[3], 4), "8": round(r[4] + r[5] + r[6], 4)}},
            {"name": "other", "share_of_insts": round(other / vmem_per_frame, 4), "probe_bytes_per_lane": 16, "records_per_instruction": {"64": 1.0}},
        ],
    }
    json.dump(mix, open(out_path, "w"), indent=1)
    print(json.dumps(mix["measured"]), json.dumps([(k["name"], k["share_of_insts"], k["records_per_instruction"]) for k in mix["kinds"]]))


if __name__ == "__main__":
    main()
