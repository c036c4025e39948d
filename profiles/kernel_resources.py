#!/usr/bin/env python3
"""Registers / scratch / occupancy of every kernel instance of one translation unit, from the compiler's own
remarks (no GPU needed):   python profiles/kernel_resources.py [kernel_stack] [extra -D flags ...]"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "shader-ray_amd")


def resources(unit="kernel_stack", extra=()):
    cmd = ["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-ffp-contract=off", "-fno-fast-math",
           "-fno-slp-vectorize", "-fhip-fp32-correctly-rounded-divide-sqrt", f"-I{ROOT}/include", f"-I{PKG}/csrc", *extra,
           "-Rpass-analysis=kernel-resource-usage", "-c", f"{PKG}/csrc/{unit}.hip", "-o", "/dev/null"]
    text = subprocess.run(cmd, capture_output=True, text=True).stderr
    rows, cur = [], {}
    for line in text.splitlines():
        m = re.search(r"remark:\s+(Function Name|VGPRs|AGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|TotalSGPRs|LDS Size \[bytes/block\]): (\S+)", line)
        if not m:
            continue
        key, val = m.group(1), m.group(2)
        if key == "Function Name":
            cur = {"name": val}
            rows.append(cur)
        else:
            cur[key.split()[0]] = val
    try:
        names = subprocess.run(["c++filt"], input="\n".join(r["name"] for r in rows), capture_output=True, text=True).stdout.splitlines()
    except FileNotFoundError:
        names = [r["name"] for r in rows]
    for r, n in zip(rows, names):
        r["pretty"] = re.sub(r"\(.*", "", n.replace("shray::", "").replace("void ", ""))
    return rows


if __name__ == "__main__":
    unit = sys.argv[1] if len(sys.argv) > 1 else "kernel_stack_batch"
    for r in resources(unit, sys.argv[2:]):
        print(f"{r['pretty']:<64} vgpr {r.get('VGPRs', '?'):>3} sgpr {r.get('TotalSGPRs', '?'):>3} scratch {r.get('ScratchSize', '?'):>4} "
              f"occupancy {r.get('Occupancy', '?')}")
