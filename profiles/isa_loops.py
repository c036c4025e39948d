#!/usr/bin/env python3
"""Where a kernel's scratch / global / LDS instructions sit relative to its loops, from the compiler's assembly (no GPU):
    python profiles/isa_loops.py [kernel_stack_batch] [substring of the kernel's name] [-D flags ...]
A loop = a label with a branch back to it; depth = number of such intervals a line lies in.  Prints per kernel the scratch
stores / loads by loop depth with the line range of the innermost loop, and instruction totals by class and depth."""
import collections
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "shader-ray_amd")
FLAGS = ["-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-ffp-contract=off", "-fno-fast-math", "-fno-slp-vectorize",
         "-fhip-fp32-correctly-rounded-divide-sqrt", f"-I{ROOT}/include", f"-I{PKG}/csrc"]


def assembly(unit, extra=()):
    with tempfile.TemporaryDirectory() as tmp:
        out = os.path.join(tmp, "k.s")
        subprocess.run(["/opt/rocm/bin/hipcc", *FLAGS, *extra, "--cuda-device-only", "-S", f"{PKG}/csrc/{unit}.hip", "-o", out],
                       check=True, capture_output=True)
        return open(out).read().split("\n")


def kernels(lines):
    cur = None
    for i, l in enumerate(lines):
        m = re.match(r"^(_Z\w+):", l)
        if m:
            cur = [m.group(1), i, None]
        if l.startswith(".Lfunc_end") and cur:
            cur[2] = i
            yield tuple(cur)
            cur = None


def loops_of(body):
    labels = {}
    for i, l in enumerate(body):
        m = re.match(r"^(\.LBB\w+):", l)
        if m:
            labels[m.group(1)] = i
    loops = []
    for i, l in enumerate(body):
        m = re.search(r"\bs_c?branch\w*\s+(\.LBB\w+)", l)
        if m and m.group(1) in labels and labels[m.group(1)] <= i:
            loops.append((labels[m.group(1)], i))
    return loops


def classify(op):
    if op.startswith("scratch_"):
        return "scratch"
    if op.startswith(("global_", "buffer_", "flat_")):
        return "vmem"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith("v_"):
        return "valu"
    if op.startswith(("s_load", "s_buffer_load")):
        return "smem"
    if op.startswith("s_"):
        return "salu"
    return None


def main():
    args = sys.argv[1:]
    extra = [a for a in args if a.startswith("-")]
    args = [a for a in args if not a.startswith("-")]
    unit = args[0] if args else "kernel_stack_batch"
    want = args[1] if len(args) > 1 else ""
    lines = assembly(unit, extra)
    for name, a, b in kernels(lines):
        pretty = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
        pretty = re.sub(r"\(.*", "", pretty.replace("shray::", "").replace("void ", ""))
        if want not in pretty:
            continue
        body = lines[a:b]
        loops = loops_of(body)
        depth = [0] * len(body)
        inner = [None] * len(body)
        for lo, hi in sorted(loops, key=lambda t: t[1] - t[0], reverse=True):
            for i in range(lo, hi + 1):
                depth[i] += 1
                inner[i] = (lo, hi)
        totals = collections.Counter()
        spots = collections.Counter()
        for i, l in enumerate(body):
            t = l.strip().split()
            if not t or t[0].startswith((".", ";")) or t[0].endswith(":"):
                continue
            c = classify(t[0])
            if not c:
                continue
            totals[(c, depth[i])] += 1
            if c == "scratch":
                spots[(t[0].split("_")[1], depth[i], inner[i])] += 1
        print(f"== {pretty}: {len(loops)} loops")
        for c in ("valu", "salu", "vmem", "smem", "lds", "scratch"):
            row = {d: n for (cc, d), n in totals.items() if cc == c}
            print(f"   {c:8s} " + "  ".join(f"depth {d}: {row[d]}" for d in sorted(row)))
        for (kind, d, rng), n in sorted(spots.items(), key=lambda kv: (kv[0][1], str(kv[0][2]))):
            where = f"loop at lines {rng[0]}-{rng[1]} ({rng[1] - rng[0]} lines)" if rng else "outside every loop"
            print(f"   scratch {kind:5s} x{n:<3d} depth {d}  {where}")


if __name__ == "__main__":
    main()
