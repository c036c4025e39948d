#!/usr/bin/env python3
"""Experiment: how much would a cost-aware workgroup order buy the stack kernel?
Per-patch cost comes from the CPU oracle's per-pixel visit map (diagnostics only); orders tried:
identity, heaviest-first, heaviest-first interleaved with lightest.  Needs libshray_hip.so
built with -DSHRAY_EXPERIMENTS."""
import ctypes as C
import os
import sys
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import torch
from __graft_entry__ import load_package
import helpers
import oracle

pkg = load_package()
N = pkg._native
lib = N.load_hip()
W, H = 1920, 1080
world = pkg.World(helpers.bunny_trisrc())
desc = world.flatten()
env = pkg.scenes.environment_hdr_sky(2048)
params = world.frame_params(W, H, material=0)
scene = pkg.Scene(desc, env, device=0)
olib = oracle.load()
vm = np.zeros((H, W), dtype=np.uint32)
olib.shray_oracle_set_visit_map(vm.ctypes.data_as(C.c_void_p))
oracle.render(desc, env, params, W, H)
olib.shray_oracle_set_visit_map(None)
cost_px = (vm & 0xffff).astype(np.float64) + (vm >> 16).astype(np.float64)
px, py = (W + 15) // 16, (H + 15) // 16
pad = np.zeros((py * 16, px * 16))
pad[:H, :W] = cost_px
# patch cost = sum over its four 8x8 wave tiles of the tile's max pixel cost (a wave lasts as long as its longest lane)
t8 = pad.reshape(py * 2, 8, px * 2, 8).max(axis=(1, 3))
patch_cost = t8.reshape(py, 2, px, 2).sum(axis=(1, 3)).reshape(-1)
n = px * py
out = torch.empty(H * W * 4, dtype=torch.float32, device="cuda")
stream = torch.cuda.current_stream().cuda_stream


def timed(label, order):
    if order is None:
        lib.shray_debug_set_patch_order(scene._handle, None, 0)
    else:
        arr = np.ascontiguousarray(order, dtype=np.uint32)
        assert sorted(arr.tolist()) == list(range(n))
        lib.shray_debug_set_patch_order(scene._handle, arr.ctypes.data_as(C.c_void_p), n)
    for _ in range(10):
        scene.render_into(params, W, H, 1, out.data_ptr(), stream)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(100):
        scene.render_into(params, W, H, 1, out.data_ptr(), stream)
    b.record()
    torch.cuda.synchronize()
    print(f"{label:40s} {a.elapsed_time(b) / 100:.4f} ms")
    return out.cpu().numpy().copy()


ref = timed("identity", None)
desc_order = np.argsort(-patch_cost, kind="stable")
img = timed("heaviest first", desc_order)
assert np.array_equal(img, ref)
# interleave: one heavy then three from the light end
heavy, light = list(desc_order[: n // 4]), list(desc_order[n // 4:][::-1])
mix = []
while heavy or light:
    if heavy:
        mix.append(heavy.pop(0))
    for _ in range(3):
        if light:
            mix.append(light.pop(0))
img = timed("1 heavy + 3 lightest, repeated", np.array(mix))
assert np.array_equal(img, ref)
rng = np.random.default_rng(0)
timed("random", rng.permutation(n))
# heavy first for the top 5% only, rest in natural order
k = n // 20
top = desc_order[:k]
rest = np.array([p for p in range(n) if p not in set(top.tolist())])
timed("top 5% first, rest natural", np.concatenate([top, rest]))

# ---- locality-oriented orders (no cost knowledge needed)
def morton_key(x, y):
    k = 0
    for b in range(8):
        k |= ((x >> b) & 1) << (2 * b) | ((y >> b) & 1) << (2 * b + 1)
    return k


ids = np.arange(n)
mort = np.array(sorted(ids, key=lambda p: morton_key(p % px, p // px)))
timed("morton", mort)


def xcd_chunked(seq, xcds=8):
    """workgroup b runs on XCD b % 8 (round-robin dispatch): give XCD k the k-th contiguous chunk of seq"""
    seq = np.asarray(seq)
    chunk = (len(seq) + xcds - 1) // xcds
    out = np.empty(len(seq), dtype=np.int64)
    pos = 0
    for i in range(chunk):
        for k in range(xcds):
            j = k * chunk + i
            if j < len(seq):
                out[pos] = seq[j]
                pos += 1
    return out[:pos]


timed("row-major, XCD-contiguous chunks", xcd_chunked(ids))
timed("morton, XCD-contiguous chunks", xcd_chunked(mort))
# column-major strips: a CU's consecutive workgroups stay in a narrow vertical band
cols = np.array(sorted(ids, key=lambda p: ((p % px) // 8, p // px, p % px)))
timed("8-patch-wide vertical strips", cols)
timed("strips, XCD-contiguous chunks", xcd_chunked(cols))
