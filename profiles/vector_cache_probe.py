#!/usr/bin/env python3
"""What the vector L1 delivers to the node-visit access pattern (shray_probe_vector_cache), on the GPU box:
    python profiles/vector_cache_probe.py [out.json]
Rates in bytes per clock per CU at 2.4 GHz over 256 CUs; TCP accesses = 64-byte quarters of a 16-byte-per-lane
wave-instruction (16 per instruction: TCP_TOTAL_CACHE_ACCESSES of the product's kernel reads 17 per vector-memory
instruction, profiles/r04/default_bench_command_summary.txt)."""
import ctypes as C
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package

pkg = load_package()
lib = pkg._native.load_hip()
CUS, GHZ = 256, 2.4
rows = []
for records, table in [(256, "8 KB table (L1-resident)"), (32768, "1 MB table (the bunny's 0.64 MB of nodes: L2-resident)"),
                       (1 << 19, "16 MB table (the 1M-triangle scene's 9.4 MB of nodes)")]:
    for spread in (1, 2, 4, 8, 16, 32, 64):
        for runs in (False, True):
            if spread in (1, 64) and runs:
                continue
            waves = CUS * 4 * 7 * 8
            sec, nbytes = C.c_double(), C.c_uint64()
            pkg._native.check(lib.shray_probe_vector_cache(records, spread | (0x80000000 if runs else 0), 512, waves, 0xffffffffffffffff, 32, C.byref(sec), C.byref(nbytes)))
            rate = nbytes.value / sec.value
            row = {"table": table, "records_per_wave_instruction": spread, "lanes_sharing_a_record": "runs of neighbours" if runs else "pseudo-random",
                   "seconds": sec.value, "TB_per_s": round(rate / 1e12, 2), "bytes_per_clk_per_cu": round(rate / (CUS * GHZ * 1e9), 1),
                   "cycles_per_16B_wave_instruction": round(1024.0 / (rate / (CUS * GHZ * 1e9)), 1)}
            rows.append(row)
            print(json.dumps(row), flush=True)
# partly filled wave-instructions: does an instruction cost by its active quads (groups of four neighbouring lanes)?
for name, mask in [("all 64 lanes", 0xffffffffffffffff), ("lanes 0-31", 0xffffffff), ("lanes 0-15", 0xffff), ("lanes 0-3 (one quad)", 0xf),
                   ("one lane of every quad (16 lanes)", 0x1111111111111111), ("one lane of every second quad (8 lanes)", 0x0101010101010101),
                   ("lane 0", 1)]:
    for spread in (1, 64):
        waves = CUS * 4 * 7 * 8
        sec, nbytes = C.c_double(), C.c_uint64()
        pkg._native.check(lib.shray_probe_vector_cache(32768, spread, 512, waves, mask, 32, C.byref(sec), C.byref(nbytes)))
        instr = waves * 512 * 2
        row = {"table": "1 MB table", "lanes": name, "records_per_wave_instruction": spread, "seconds": sec.value,
               "cycles_per_wave_instruction": round(sec.value * GHZ * 1e9 * CUS / instr, 2)}
        rows.append(row)
        print(json.dumps(row), flush=True)
# one load of 4 ... 16 bytes per lane: is a wave-instruction's cost its width?
for width in (16, 12, 8, 4):
    for spread in (1, 64):
        waves = CUS * 4 * 7 * 8
        sec, nbytes = C.c_double(), C.c_uint64()
        pkg._native.check(lib.shray_probe_vector_cache(32768, spread, 512, waves, 0xffffffffffffffff, width, C.byref(sec), C.byref(nbytes)))
        row = {"table": "1 MB table", "bytes_per_lane": width, "records_per_wave_instruction": spread, "seconds": sec.value,
               "cycles_per_wave_instruction": round(sec.value * GHZ * 1e9 * CUS / (waves * 512), 2)}
        rows.append(row)
        print(json.dumps(row), flush=True)
out = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "vector_cache_probe.json")
json.dump({"cus": CUS, "clock_ghz": GHZ, "rows": rows}, open(out, "w"), indent=1)
