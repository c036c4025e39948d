#!/usr/bin/env python3
"""Condenses a rocprofv3 output tree written by profiles/run_profile.sh into a short text
summary: per-kernel average duration from the kernel trace, and per-dispatch averages of
each PMC counter for the trace kernels."""
import csv
import glob
import os
import sys
from collections import defaultdict

root = sys.argv[1]


def rows(pattern):
    # gpurun merges every call's files into the same local tree: per directory keep only the newest run
    # (rocprofv3 prefixes its files with the process id)
    newest = {}
    for path in glob.glob(os.path.join(root, pattern), recursive=True):
        d = os.path.dirname(path)
        if d not in newest or os.path.getmtime(path) > os.path.getmtime(newest[d]):
            newest[d] = path
    for path in sorted(newest.values()):
        with open(path, newline="") as f:
            for r in csv.DictReader(f):
                yield r


print("== kernel trace (durations in us) ==")
dur = defaultdict(list)
for r in rows("trace/**/*kernel_trace.csv"):
    name = r.get("Kernel_Name", "?")
    try:
        dur[name].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    except Exception:
        pass
for name, d in sorted(dur.items(), key=lambda kv: -sum(kv[1])):
    d2 = sorted(d)
    print(f"{name[:100]:100s} calls {len(d):5d}  avg {sum(d)/len(d):10.2f}  median {d2[len(d2)//2]:10.2f}  min {d2[0]:10.2f}  max {d2[-1]:10.2f}")

print("\n== PMC counters (average per dispatch, trace kernels only) ==")
acc = defaultdict(lambda: defaultdict(list))
meta = {}
for r in rows("pmc*/**/*counter_collection.csv"):
    name = r.get("Kernel_Name", "?")
    if "trace_" not in name:
        continue
    try:
        acc[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
        meta[name] = (r.get("VGPR_Count"), r.get("SGPR_Count"), r.get("LDS_Block_Size"), r.get("Grid_Size"), r.get("Workgroup_Size"))
    except Exception:
        pass
for name, counters in acc.items():
    print(name[:120], "VGPR/SGPR/LDS/grid/wg =", meta.get(name))
    for c, v in sorted(counters.items()):
        print(f"   {c:32s} n={len(v):4d} avg {sum(v)/len(v):18.1f}")
