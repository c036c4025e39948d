#!/usr/bin/env python3
"""DESIGN-TIME DIAGNOSTIC (test infrastructure): drives oracle/tools/wave_sim.cpp over a trace written by
oracle/tools/dump_trace.py and prints, per scheduling policy, the modelled wave-instruction count, the
fraction of lanes active, and the heaviest instruction streams (what bounds a lone frame's latency).

    python oracle/tools/wave_sim.py /tmp/trace/c2 [policy=value ...]
"""
import ctypes as C
import os
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
SO = "/tmp/wave_sim.so"
if not os.path.exists(SO) or os.path.getmtime(SO) < os.path.getmtime(os.path.join(HERE, "wave_sim.cpp")):
    subprocess.run(["g++", "-O2", "-shared", "-fPIC", "-o", SO, os.path.join(HERE, "wave_sim.cpp")], check=True)
lib = C.CDLL(SO)


class Costs(C.Structure):
    _fields_ = [(n, C.c_double) for n in ("gen", "setup", "node", "tri", "shade", "env", "stage_switch", "loop_iter",
                                          "deal_setup", "deal_round", "deal_finish", "compact", "event")]


class Policy(C.Structure):
    _fields_ = [(n, C.c_int) for n in ("keep_num", "keep_floor", "node_turns", "deal_max_parked", "compact", "deal_group", "async_lanes", "event_min", "pixels_per_wave", "epoch_turns", "epoch_slack", "interleave", "split_heavy", "leaf_cap")]


def run(prefix, costs, policy):
    bytes_ = np.load(prefix + "_bytes.npy", mmap_mode="r")
    bytes_ = np.ascontiguousarray(bytes_)
    offs = np.load(prefix + "_offsets.npy")
    W, H, spp = (int(x) for x in open(prefix + "_meta.txt").read().split())
    groups = ((W + 15) // 16) * ((H + 15) // 16)
    out = np.zeros(24)
    ws = np.zeros(groups * 4)
    gp = np.zeros(groups)
    lib.wave_sim(bytes_.ctypes.data_as(C.c_void_p), offs.ctypes.data_as(C.c_void_p), W, H, spp, C.byref(costs), C.byref(policy),
                 out.ctypes.data_as(C.c_void_p), ws.ctypes.data_as(C.c_void_p), gp.ctypes.data_as(C.c_void_p))
    return out, ws, gp


def report(label, out, ws, gp):
    wi, li = out[0], out[1]
    top = np.sort(ws)[::-1]
    gtop = np.sort(gp)[::-1]
    print(f"{label:46s} wave-instr {wi/1e6:8.1f} M  lanes {li/wi/64*100:5.1f} %  | node {out[2]/1e6:7.1f} M @ {out[3]/out[2]/64*100:4.1f} %"
          f"  leaf {out[4]/1e6:7.1f} M @ {out[5]/max(out[4],1)/64*100:4.1f} %  other {out[6]/1e6:6.1f} M"
          f" | turns node {out[8]/1e6:6.2f} M leaf {out[9]/1e6:6.2f} M | heaviest wave {top[0]/1e3:6.1f} k, #60 {top[59]/1e3:6.1f} k,"
          f" mean {ws.mean()/1e3:5.2f} k; heaviest group path {gtop[0]/1e3:6.1f} k")
    tot = wi * 64
    if os.environ.get("IDLE"):
        print(f"    idle lane-slots, % of all: node stage: parked {out[11]/tot*100:4.1f}  ended {out[12]/tot*100:4.1f}  no ray {out[13]/tot*100:4.1f}"
              f" | leaf stage: walkers waiting {out[14]/tot*100:4.1f}  ended {out[15]/tot*100:4.1f}  own leaf done {out[16]/tot*100:4.1f}  no ray {out[17]/tot*100:4.1f}")


if __name__ == "__main__":
    prefix = sys.argv[1]
    costs = Costs(gen=90, setup=70, node=64, tri=66, shade=120, env=220, stage_switch=8, loop_iter=10,
                  deal_setup=30, deal_round=85, deal_finish=35, compact=60, event=280)
    base = dict(keep_num=36, keep_floor=2, node_turns=3, deal_max_parked=0, compact=0, deal_group=0, async_lanes=0, event_min=16, pixels_per_wave=64, epoch_turns=0, epoch_slack=0, interleave=0, split_heavy=0, leaf_cap=0)
    variants = [("current (keep 36/64, 3 node turns)", {})]
    for kn in (16, 40, 48):
        variants.append((f"keep {kn}/64", dict(keep_num=kn)))
    variants += [
        ("dealt leaves when <= 16 parked (general)", dict(deal_max_parked=16)),
        ("dealt leaves when <= 32 parked (general)", dict(deal_max_parked=32)),
        ("dealt leaves always (general)", dict(deal_max_parked=64)),
        ("dealt leaves always, keep 48/64", dict(deal_max_parked=64, keep_num=48)),
        ("dealt leaves <= 16 (pow2 groups)", dict(deal_max_parked=16, deal_group=1)),
        ("dealt leaves <= 32 (pow2 groups)", dict(deal_max_parked=32, deal_group=1)),
        ("workgroup compaction", dict(compact=1)),
        ("compaction + dealt <= 32 (general)", dict(compact=1, deal_max_parked=32)),
        ("compaction + dealt always, keep 48", dict(compact=1, deal_max_parked=64, keep_num=48)),
    ]
    for em in (8, 16, 32):
        variants.append((f"async lanes, event stage at >= {em}", dict(async_lanes=1, event_min=em)))
    variants.append(("async lanes, event >= 16, dealt always keep 48", dict(async_lanes=1, event_min=16, deal_max_parked=64, keep_num=48)))
    for ppw in (128, 256):
        variants.append((f"async, {ppw} px per wave, event >= 16", dict(async_lanes=1, event_min=16, pixels_per_wave=ppw)))
        variants.append((f"async, {ppw} px per wave, event >= 16, dealt, keep 48", dict(async_lanes=1, event_min=16, pixels_per_wave=ppw, deal_max_parked=64, keep_num=48)))
    for et in (4, 8, 16, 32):
        variants.append((f"epoch compaction every {et} node turns", dict(compact=1, epoch_turns=et)))
    for et in (8, 16):
        variants.append((f"epoch {et} + dealt always keep 48", dict(compact=1, epoch_turns=et, deal_max_parked=64, keep_num=48)))
        variants.append((f"epoch {et} slack 32 + dealt always keep 48", dict(compact=1, epoch_turns=et, epoch_slack=32, deal_max_parked=64, keep_num=48)))
    variants.append(("dealt always keep 48, interleaved waves", dict(deal_max_parked=64, keep_num=48, interleave=1)))
    variants.append(("current, interleaved waves", dict(interleave=1)))
    for thr in (20000, 30000, 40000):
        variants.append((f"dealt always keep 48, heavy waves (> {thr}) split in 4", dict(deal_max_parked=64, keep_num=48, split_heavy=thr)))
    for cap in (1, 2, 3, 4, 6):
        variants.append((f"leaf stage capped at {cap} rounds while rays walk", dict(leaf_cap=cap)))
        variants.append((f"leaf cap {cap}, keep 48/64", dict(leaf_cap=cap, keep_num=48)))
    if os.environ.get("ONLY"):
        variants = [v for v in variants if os.environ["ONLY"] in v[0] or v[0].startswith("current")]
    for label, over in variants:
        pol = Policy(**{**base, **over})
        report(label, *run(prefix, costs, pol))
