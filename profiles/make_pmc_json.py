#!/usr/bin/env python3
"""Turns a profiles/run_profile.sh summary (rocprofv3 kernel trace + separate --pmc passes of the default
bench command) into the counter file bench.py reads for its roofline object:

    python profiles/make_pmc_json.py <summary.txt> <out.json> [kernel name substring]

HBM bytes follow /opt/skills/guides/MI355X_MICROARCH.md (HBM section): rocprofv3 reports FETCH_SIZE and
WRITE_SIZE in KiB; on gfx950 FETCH_SIZE counts the 128-byte requests of 16-byte-per-lane loads as 64 bytes,
so the read side is doubled; WRITE_SIZE is exact for 16-byte-per-lane stores."""
import json
import os
import re
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from buildhash import kernel_source_hash  # noqa: E402

summary_path, out_path = sys.argv[1], sys.argv[2]
want = sys.argv[3] if len(sys.argv) > 3 else "trace_stack_batch_kernel<true, true, false>"
frames_per_launch = int(sys.argv[4]) if len(sys.argv) > 4 else 2
text = open(summary_path).read()
trace, pmc = text.split("== PMC counters", 1)
avg_us = None
for line in trace.splitlines():
    if want in line:
        avg_us = float(re.search(r"avg\s+([\d.]+)", line).group(1))
        calls = int(re.search(r"calls\s+(\d+)", line).group(1))
        break
block = pmc[pmc.index(want):]
nxt = re.search(r"\n(?=\S)", block[1:])
block = block if not nxt else block[:nxt.start() + 1]
vals = {m.group(1): float(m.group(2)) for m in re.finditer(r"^\s+(\w+)\s+n=\s*\d+\s+avg\s+([\d.]+)", block, re.M)}
fetch, write = vals.get("FETCH_SIZE"), vals.get("WRITE_SIZE")


def serialized_launch():
    """(median duration in ms of the kernel's launches in the counter passes -- rocprofv3 runs one dispatch at a time while it
    counts --, the clock GRBM_GUI_ACTIVE / 8 XCDs ticks at over those launches in GHz), from the pass that counted
    GRBM_GUI_ACTIVE; (None, None) when the per-dispatch files are not beside the summary."""
    import csv
    import glob
    import statistics
    for path in sorted(glob.glob(os.path.join(os.path.dirname(summary_path), "pmc*", "*", "*counter_collection.csv"))):
        per = {}
        for r in csv.DictReader(open(path)):
            if want in r["Kernel_Name"] and r["Counter_Name"] == "GRBM_GUI_ACTIVE":
                per[r["Dispatch_Id"]] = (float(r["Counter_Value"]) / 8.0, int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
        if per:
            ns = statistics.median(d for _, d in per.values())
            ghz = statistics.median(c / d for c, d in per.values() if d > 0)
            return ns / 1e6, ghz
    return None, None


serial_ms, serial_ghz = serialized_launch()
out = {
    "kernel": want,
    "workload": {"width": 1920, "height": 1080, "spp": 1, "material": 0, "kernel_id": 0, "frames_per_launch": frames_per_launch,
                 "streams": int(sys.argv[5]) if len(sys.argv) > 5 else 4,
                 "orbit": int(os.environ.get("SHRAY_PMC_ORBIT", "20"))},   # distinct views the frames cycle through (bench.py: ORBIT)
    "build_hash": kernel_source_hash(),
    "kernel_trace_avg_us": avg_us, "kernel_trace_calls": calls,
    "counters_per_launch": vals,
    "valu_insts_per_launch": vals.get("SQ_INSTS_VALU"),
    "lane_util": (vals["SQ_THREAD_CYCLES_VALU"] / (vals["SQ_INSTS_VALU"] * 64.0)) if "SQ_THREAD_CYCLES_VALU" in vals else None,
    "hbm_bytes_per_launch": int((2 * fetch + write) * 1024) if fetch is not None and write is not None else None,
    "hbm_note": "2 x FETCH_SIZE (gfx950 half-count of 16 B/lane loads) + WRITE_SIZE, KiB -> bytes, per launch",
    # the vector memory pipeline (one texture-addresser / texture-data pair per CU, shared by the CU's four SIMDs):
    # wave-instructions it was handed, and how busy it was while the kernel ran (TA_BUSY_avr = busy cycles averaged over
    # the 256 instances; GRBM_GUI_ACTIVE is summed over the 8 XCDs)
    "vmem_insts_per_launch": vals.get("SQ_INSTS_VMEM_RD"),
    "smem_insts_per_launch": vals.get("SQ_INSTS_SMEM"),
    # (busy cycles summed over the instances / 256 active CUs, over the kernel's own cycles GRBM_GUI_ACTIVE / 8 XCDs; rocprofv3
    # runs one dispatch at a time while it counts, so these are fractions of a launch that has the GPU to itself)
    "ta_busy_frac": (vals["TA_TA_BUSY_sum"] / 256.0 / (vals["GRBM_GUI_ACTIVE"] / 8.0)) if "TA_TA_BUSY_sum" in vals and vals.get("GRBM_GUI_ACTIVE") else None,
    "td_busy_frac": (vals["TD_TD_BUSY_sum"] / 256.0 / (vals["GRBM_GUI_ACTIVE"] / 8.0)) if "TD_TD_BUSY_sum" in vals and vals.get("GRBM_GUI_ACTIVE") else None,
    "valu_busy_frac_profiled": (vals["SQ_INSTS_VALU"] * 2.0 / 1024.0 / (vals["GRBM_GUI_ACTIVE"] / 8.0)) if "SQ_INSTS_VALU" in vals and vals.get("GRBM_GUI_ACTIVE") else None,
    # the mix of vector instructions by the classes the hardware counts (profiles/r04/valu_costs_probe.txt: f32 add / sub / mul
    # issue one per 2 cycles, fma one per 4, transcendentals one per 8; what is in none of the five -- min / max, compares,
    # selects, moves, conversions -- is of either class)
    "valu_mix_per_launch": {k[len("SQ_INSTS_VALU_"):].lower(): vals[k] for k in ("SQ_INSTS_VALU_ADD_F32", "SQ_INSTS_VALU_MUL_F32",
                            "SQ_INSTS_VALU_FMA_F32", "SQ_INSTS_VALU_TRANS_F32", "SQ_INSTS_VALU_INT32") if k in vals} or None,
    "serialized_launch_ms": serial_ms, "serialized_clock_ghz": serial_ghz,
    "kernel_cycles_profiled": (vals["GRBM_GUI_ACTIVE"] / 8.0) if vals.get("GRBM_GUI_ACTIVE") else None,
    "wait_frac": (vals["SQ_WAIT_ANY"] / vals["SQ_WAVE_CYCLES"]) if "SQ_WAIT_ANY" in vals and vals.get("SQ_WAVE_CYCLES") else None,
    "source": summary_path,
}
json.dump(out, open(out_path, "w"), indent=1)
print(json.dumps(out))
