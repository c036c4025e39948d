#!/usr/bin/env python3
"""DESIGN-TIME DIAGNOSTIC (test infrastructure): rewrites a multi-sample trace of dump_trace.py as a 1 spp trace of a
virtual image in which the g = min(spp, 64) samples of a pixel are g neighbouring pixels, so that wave_sim.py's 8x8
wave tile holds 64 / g pixels x g samples: the "lanes = samples" mapping of a multi-sample frame.

    python oracle/tools/remap_trace.py <in prefix> <out prefix>
"""
import sys

import numpy as np

src, dst = sys.argv[1:3]
W, H, spp = (int(x) for x in open(src + "_meta.txt").read().split())
offs = np.load(src + "_offsets.npy").astype(np.int64)
data = np.load(src + "_bytes.npy", mmap_mode="r")
g = min(spp, 64)
gx = {1: 1, 2: 2, 4: 2, 8: 4, 16: 4, 32: 8, 64: 8}[g]
gy = g // gx
assert spp == g, "one group of samples per pixel"
VW, VH = W * gx, H * gy
vy, vx = np.mgrid[0:VH, 0:VW]
s = (vy % gy) * gx + (vx % gx)
idx = (((vy // gy) * W + (vx // gx)) * spp + s).reshape(-1)
length = (offs[1:] - offs[:-1])[idx]
new_offs = np.zeros(len(idx) + 1, dtype=np.uint64)
new_offs[1:] = np.cumsum(length)
out = np.empty(int(new_offs[-1]), dtype=np.uint8)
start = offs[:-1][idx]
pos = 0
CH = 1 << 18
for a in range(0, len(idx), CH):
    st, ln = start[a:a + CH], length[a:a + CH]
    total = int(ln.sum())
    src_index = np.repeat(st - np.concatenate(([0], np.cumsum(ln)[:-1])), ln) + np.arange(total)
    out[pos:pos + total] = data[src_index]
    pos += total
np.save(dst + "_bytes.npy", out)
np.save(dst + "_offsets.npy", new_offs)
open(dst + "_meta.txt", "w").write(f"{VW} {VH} 1\n")
print("virtual image", VW, VH, "bytes", pos)
