#!/usr/bin/env python3
"""Benchmark of the per-pixel hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W]

Workload (BASELINE.json configs[1]): bunny-class mesh (69,168 triangles, written as
trisrc text and loaded through the real parser + BVH builder), seeded HDR sky environment,
1920x1080, 1 spp, default gold material, 3 bounces.  A "step" is one frame: every rank
renders its interleaved tiles with the HIP kernel and (N > 1) the packed tile buffers are
gathered to rank 0 over RCCL and de-interleaved.  Scene and environment are resident in
HBM before the timed region.  Rank 0 prints ONE JSON line.

Successive frames are independent.  N = 1: the frame loop hands the C ABI two frames per launch and
alternates launches over four HIP streams (profiles/r02/leaf_stage_ab.txt section 10).  N > 1: a
launch carries N consecutive frames (the rank's tiles of each; shray_render_batch_device), one
gather moves all N, and two such launches alternate on two streams -- a rank's share of ONE
frame is latency-bound, see DESIGN.md section 6.  Exactly K frames are rendered in the timed
region either way (the last launch is shorter when N does not divide K).

For N > 1 launch as
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

# Two frames are kept in flight on two HIP streams; they only overlap when the streams land on
# different hardware queues.  With the runtime's default queue count the side stream was seen to
# alias the queue RCCL's stream uses (no overlap at all); an explicit count avoids that.  Must be
# set before the HIP runtime initialises.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

WIDTH, HEIGHT, SPP = 1920, 1080, 1
HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: HBM3E peak 8.0 TB/s (spec)
# VALU issue peak (MI355X_MICROARCH.md): 256 CUs x 4 SIMDs, one wave64 VALU instruction per 2 cycles per SIMD, 2.4 GHz
VALU_PEAK_GINST = 256 * 4 * 2.4 / 2.0   # = 1228.8 G wave-instructions / s
PMC_FILE = os.path.join("profiles", "r02", "pmc_headline.json")   # written by profiles/make_pmc_json.py from rocprofv3 --pmc passes


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def cpu_baseline(pkg, desc, env, params, budget_s=12.0):
    """The CPU oracle (oracle/, a port of the shader: the reference has no CPU tracer) on the
    same frame, all host cores; repeated for ~budget_s of wall time."""
    import oracle
    threads = os.cpu_count() or 1
    oracle.render(desc, env, params, WIDTH, HEIGHT, SPP, threads=threads)   # warm-up (page faults, thread pool)
    reps = 0
    t0 = time.perf_counter()
    while True:
        oracle.render(desc, env, params, WIDTH, HEIGHT, SPP, threads=threads)
        reps += 1
        elapsed = time.perf_counter() - t0
        if elapsed >= budget_s or reps >= 2000:
            break
    dt = elapsed / reps
    return {"value": round(WIDTH * HEIGHT * SPP / dt / 1e6, 4), "unit": "Mrays/s", "cores": threads, "kind": "port",
            "sample": f"the full {WIDTH}x{HEIGHT} frame of the same workload, {reps} repetitions in {elapsed:.1f} s, "
                      f"CPU oracle with {threads} threads"}


def main():
    global WIDTH, HEIGHT, SPP
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--trials", type=int, default=10,
                    help="the timed K-step loop is repeated this many times (each bracketed by barrier + synchronize); the median trial is reported")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--kernel", type=int, default=0)
    ap.add_argument("--width", type=int, default=WIDTH, help="frame width (default: the headline configuration)")
    ap.add_argument("--height", type=int, default=HEIGHT)
    ap.add_argument("--spp", type=int, default=SPP, help="samples per pixel; --width 3840 --height 2160 --spp 16 is BASELINE configs[4]")
    ap.add_argument("--frames-in-flight", type=int, default=0,
                    help="independent launches alternate over this many HIP streams (1 = strictly one at a time; "
                         "default: 4 for N = 1, 2 for N > 1)")
    ap.add_argument("--frames-per-launch", type=int, default=0,
                    help="consecutive frames per launch (shray_render_batch_device); N > 1: also per gather.  Default: 2 for "
                         "N = 1, the number of GPUs for N > 1 (a launch then carries one frame's worth of pixels per GPU)")
    ap.add_argument("--rgba-wire", action="store_true", help="N > 1 only: gather RGBA instead of RGB (alpha is the constant 1)")
    args = ap.parse_args()
    WIDTH, HEIGHT, SPP = args.width, args.height, args.spp

    import torch
    import torch.distributed as dist

    from __graft_entry__ import load_package
    import helpers

    world_size = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world_size:
        if world_size == 1 and args.gpus > 1:
            raise SystemExit("--gpus N > 1 must be launched with torch.distributed.run (one process per GPU)")
    # SHRAY_FORCE_DIST=1 rehearses the multi-GPU code path (process group, barrier, gather) with one rank
    distributed = world_size > 1 or os.environ.get("SHRAY_FORCE_DIST") == "1"
    # Rehearsal on a single-GPU box: SHRAY_BENCH_ONE_GPU=1 puts every rank on cuda:0 and
    # SHRAY_BENCH_BACKEND=gloo swaps RCCL (which refuses two ranks on one GPU) for gloo, staging the
    # gather through host memory.  Only the control flow is rehearsed that way, never a reported number.
    one_gpu = os.environ.get("SHRAY_BENCH_ONE_GPU") == "1"
    backend = os.environ.get("SHRAY_BENCH_BACKEND", "nccl")
    if one_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if distributed:
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=device)
        else:
            dist.init_process_group(backend)

    pkg = load_package()
    # rank 0 generates the scene file once; the others wait for it
    if rank == 0:
        path = helpers.bunny_trisrc()
    if distributed:
        dist.barrier()
    path = helpers.bunny_trisrc()
    world = pkg.World(path)
    desc = world.flatten()
    env = pkg.scenes.environment_hdr_sky(2048)
    params = world.frame_params(WIDTH, HEIGHT, material=0)
    scene = pkg.Scene(desc, env, device=local_rank)
    scene.set_kernel(args.kernel)
    stream = torch.cuda.current_stream().cuda_stream

    from shader_ray_amd import multigpu

    tile = multigpu.DEFAULT_TILE

    # Successive frames are independent.  With one frame at a time the last ~35 % of a 1 spp 1080p
    # frame is a tail of a few heavy waves on an otherwise idle GPU; alternating
    # frames over two HIP streams (double buffering, as any frame loop does) lets frame k+1's bulk
    # fill frame k's tail -- and, with several GPUs, lets the gather of frame k overlap the render
    # of frame k+1.  Every frame is still rendered completely into its own buffer.
    lanes = max(1, args.frames_in_flight or (2 if distributed else 4))
    # N > 1: a rank's share of one 1080p frame is latency-bound (its long-running waves take ~0.5 ms
    # wherever they land), so a launch carries `batch` consecutive frames (shray_render_batch_device) and
    # one gather moves them all: fewer, larger collectives, and the GPU stays full.
    # N = 1: the frame loop hands the C ABI two frames per launch (shray_render_batch_device, the throughput form of
    # the frame loop, ray.cpp:1096-1131); --frames-per-launch 1 --frames-in-flight 1 is strictly one frame at a time
    batch = max(1, min(64, args.frames_per_launch or (world_size if distributed else 2)))
    streams = [torch.cuda.current_stream()] + [torch.cuda.Stream(device=device) for _ in range(lanes - 1)]
    frame_outs = [torch.empty(batch * HEIGHT * WIDTH * 4, dtype=torch.float32, device=device) for _ in range(lanes)]
    splits = [multigpu.DistributedFrame(WIDTH, HEIGHT, tile, tile, device=device, always_gather=True,
                                        stage_through_host=(backend != "nccl"), frames=batch,
                                        rgb_wire=not args.rgba_wire) for _ in range(lanes)] if distributed else None
    trials = max(1, args.trials)
    starts, stops, launch_frames = [], [], []
    EVENT_STRIDE = max(1, int(os.environ.get("SHRAY_BENCH_EVENT_STRIDE", "4")))
    torch.cuda.synchronize()

    def step(j, count, timed=None):
        """launch j of the single-GPU path: `count` consecutive frames; timed: HIP events bracket the launch"""
        lane = j % lanes
        st = streams[lane]
        # HIP events bracket every EVENT_STRIDE-th launch of the timed region (an event is a barrier packet on its stream:
        # bracketing every launch costs the one-frame-at-a-time form ~1 % of its step)
        timed = timed if (timed is not None and j % EVENT_STRIDE == 0) else None
        if timed is not None:
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            starts.append(a)
            stops.append(b)
            launch_frames.append(count)
            a.record(st)
        if count == 1:
            scene.render_into(params, WIDTH, HEIGHT, SPP, frame_outs[lane].data_ptr(), st.cuda_stream, None)
        else:
            scene.render_batch_into([params] * count, WIDTH, HEIGHT, SPP, frame_outs[lane].data_ptr(), HEIGHT * WIDTH * 16,
                                    st.cuda_stream, None)
        if timed is not None:
            b.record(st)
        return frame_outs[lane]

    def launch(j, count):
        """`count` frames: this rank's tiles in one launch, one gather to rank 0, one de-interleave"""
        lane = j % lanes
        st = streams[lane]
        split = splits[lane]

        def render_tiles(tile_set, out):
            scene.render_batch_into([params] * count, WIDTH, HEIGHT, SPP, out.data_ptr(), split.frame_stride_bytes,
                                    st.cuda_stream, tile_set)
        with torch.cuda.stream(st):
            return split.render(render_tiles, count)

    def run(frames, timed=None):
        """exactly `frames` frames; returns what the last launch produced"""
        last = None
        if not distributed:
            done = j = 0
            while done < frames:
                count = min(batch, frames - done)
                last = step(j, count, timed)
                done += count
                j += 1
            return last
        done = j = 0
        while done < frames:
            count = min(batch, frames - done)
            last = launch(j, count)
            done += count
            j += 1
        return last

    def fence():
        if distributed:
            dist.barrier()
        torch.cuda.synchronize()

    run(args.warmup)
    fence()

    # `trials` repetitions of the timed region; each one is EXACTLY K steps between two fences (barrier +
    # synchronize), its time the MAX over ranks.  The median trial is what the JSON line reports, so that a
    # short K (the driver's --steps 20 is 7 ms of GPU time) is not a single noisy sample.
    trial_s = []
    for trial in range(trials):
        t0 = time.perf_counter()
        run(args.steps, timed=trial)
        fence()
        dt = time.perf_counter() - t0
        if distributed:
            t = torch.tensor([dt], dtype=torch.float64, device=device)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        trial_s.append(dt)
    elapsed = sorted(trial_s)[len(trial_s) // 2]

    if distributed and os.environ.get("SHRAY_BENCH_CHECK") == "1":
        # rehearsal aid (every rank takes part in the extra launch): every assembled frame must
        # equal a single-GPU render of the whole frame
        last = run(batch)
        torch.cuda.synchronize()
        if rank == 0:
            whole = torch.empty(HEIGHT * WIDTH * 4, dtype=torch.float32, device=device)
            scene.render_into(params, WIDTH, HEIGHT, SPP, whole.data_ptr(), torch.cuda.current_stream().cuda_stream, None)
            torch.cuda.synchronize()
            last = last if last.dim() == 4 else last.unsqueeze(0)
            same = all(bool(torch.equal(last[f].reshape(-1), whole)) for f in range(last.shape[0]))
            log(f"all {last.shape[0]} assembled frame(s) equal the single-GPU frame:", same)

    result = None
    if rank == 0:
        rays = WIDTH * HEIGHT * SPP * args.steps
        result = {
            "metric": "Mrays/s at 1920x1080 1spp (bunny.trisrc); 1/2/4/8-GPU scaling" if (WIDTH, HEIGHT, SPP) == (1920, 1080, 1)
            else f"Mrays/s at {WIDTH}x{HEIGHT} {SPP}spp (bunny.trisrc)",
            "value": round(rays / elapsed / 1e6, 3), "unit": "Mrays/s", "n_gpus": world_size, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 5), "higher_is_better": True,
            "trials": trials, "trial_ms": [round(t * 1e3, 4) for t in trial_s],
            "timing": f"median of {trials} trials of exactly {args.steps} steps, each between barrier + synchronize fences",
            "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "bunny-class trisrc (69,168 triangles, synthetic stand-in for bunny.trisrc) + seeded "
                                   f"2048x1024 HDR sky, {WIDTH}x{HEIGHT}, {SPP} spp, gold, 3 bounces"
                                   + (" (BASELINE configs[1])" if (WIDTH, HEIGHT, SPP) == (1920, 1080, 1) else ""),
                       "width": WIDTH, "height": HEIGHT, "spp": SPP, "kernel": {0: "stack", 1: "threaded", 2: "pool"}[args.kernel],
                       "parallelism": f"tiles{tile}x{tile}-interleaved-x{world_size}" if distributed else "single-gpu",
                       "frames_in_flight": lanes * batch, "frames_per_launch": batch, "streams": lanes,
                       "wire": ("rgb32f" if not args.rgba_wire else "rgba32f") if distributed else None},
        }
    if distributed and rank == 0:
        # all ranks together execute exactly the fetches of the whole frame: per-GPU algorithmic rate
        _, counters = scene.render_counters(params, WIDTH, HEIGHT, SPP, want_image=False)
        algo_bytes = pkg.tracer.algorithmic_bytes(counters, WIDTH * HEIGHT, normals_fp16=True)
        per_gpu = algo_bytes * args.steps / elapsed / 1e9 / world_size
        result["roofline"] = {"bound": "valu_issue", "achieved": None, "peak": VALU_PEAK_GINST, "unit": "Gwaveinst/s", "frac": None,
                              "traffic": None, "traffic_source": "not collected for N > 1 (see the N = 1 line)",
                              "algorithmic_cacheless": {"bytes_per_frame": algo_bytes, "gbs_per_gpu": round(per_gpu, 2),
                                                        "note": "cache-less count of the reference's fetches / wall time / n_gpus "
                                                                "(gather and de-interleave included in the time); not a bound"}}
        result["counters"] = counters
    if not distributed:
        # per-launch kernel time: HIP events recorded around every EVENT_STRIDE-th launch of every trial, on the stream that
        # launch went to.  With frames_in_flight > 1 two launches share the GPU, so each lasts longer than it
        # would alone while the pair finishes sooner: rates below use WALL time, not per-launch time.
        kernel_ms = sorted(s.elapsed_time(e) for s, e in zip(starts, stops))
        avg_ms = sum(kernel_ms) / len(kernel_ms)   # per LAUNCH (a launch carries `batch` frames)
        _, counters = scene.render_counters(params, WIDTH, HEIGHT, SPP, want_image=False)
        algo_bytes = pkg.tracer.algorithmic_bytes(counters, WIDTH * HEIGHT, normals_fp16=True)
        frames_per_s = args.steps / elapsed
        # hardware counters of the dominant kernel come from a committed rocprofv3 --pmc run of THIS command
        # (they cannot be read from inside the process); used only if they were taken on this workload
        pmc, pmc_note = None, "no counter file"
        try:
            sys.path.insert(0, os.path.join(ROOT, "profiles"))
            from buildhash import kernel_source_hash
            cand = json.load(open(os.path.join(ROOT, PMC_FILE)))
            wl = cand["workload"]
            if (wl["width"], wl["height"], wl["spp"], wl["kernel_id"], wl.get("frames_per_launch", 1)) == \
                    (WIDTH, HEIGHT, SPP, args.kernel, batch) and cand["valu_insts_per_launch"]:
                pmc = cand
                pmc_note = PMC_FILE + (" (same kernel sources as this build)" if cand["build_hash"] == kernel_source_hash()
                                       else " (STALE: measured on different kernel sources than this build)")
            else:
                pmc_note = PMC_FILE + " is for another workload"
        except Exception as exc:   # noqa: BLE001
            pmc_note = f"{PMC_FILE} unreadable: {exc}"
        roof = {"bound": "valu_issue", "achieved": None, "peak": VALU_PEAK_GINST, "unit": "Gwaveinst/s", "frac": None,
                "lane_util": None, "traffic": None, "traffic_source": pmc_note, "hbm_frac": None,
                "counter_source": pmc_note,
                "why": "the scene + environment working set (32 MB) is cache-resident: the kernel is bound by VALU issue, "
                       "not by HBM (DESIGN.md section 4.4); HBM use is reported as hbm_frac"}
        if pmc:
            fpl = pmc["workload"].get("frames_per_launch", 1)   # the profiled launches carried this many frames each
            ginst = pmc["valu_insts_per_launch"] / fpl * frames_per_s / 1e9
            roof.update({"achieved": round(ginst, 2), "frac": round(ginst / VALU_PEAK_GINST, 5),
                         "lane_util": round(pmc["lane_util"], 4) if pmc.get("lane_util") else None,
                         "useful_frac": round(ginst / VALU_PEAK_GINST * pmc["lane_util"], 5) if pmc.get("lane_util") else None,
                         "valu_insts_per_frame": pmc["valu_insts_per_launch"] / fpl,
                         "profiled_kernel_us": pmc.get("kernel_trace_avg_us")})
            if pmc.get("hbm_bytes_per_launch"):
                hbm_gbs = pmc["hbm_bytes_per_launch"] / fpl * frames_per_s / 1e9
                roof.update({"traffic": pmc["hbm_bytes_per_launch"], "traffic_frames": fpl, "hbm_gbs": round(hbm_gbs, 2),
                             "hbm_frac": round(hbm_gbs / HBM_PEAK_GBS, 5)})
        algo_gbs = algo_bytes * frames_per_s / 1e9
        roof["algorithmic_cacheless"] = {
            "bytes_per_frame": algo_bytes, "bytes_per_ray": round(algo_bytes / (WIDTH * HEIGHT * SPP), 1),
            "gbs": round(algo_gbs, 2),
            "note": "SURVEY 8(d)'s cache-less count of the REFERENCE's fetches x frames / wall time; these bytes are served by "
                    "L1/L2, not by HBM: not a bound (it exceeds the 8000 GB/s HBM peak), no fraction is formed from it"}
        roof.update({"kernel_ms_avg": round(avg_ms, 5), "kernel_ms_median": round(kernel_ms[len(kernel_ms) // 2], 5),
                     "concurrent_launches": lanes, "frames_per_launch": batch,
                     "kernel_events": f"{len(kernel_ms)} launches of the timed region bracketed (every {EVENT_STRIDE}th)"})
        result["roofline"] = roof
        result["counters"] = counters
        # the C ABI's host-buffer forms, PCIe-inclusive, for the record (never `value`): the blocking call into
        # pageable memory (the runtime's staged copy, into a buffer the loop reuses) and the stream form into pinned memory, double-buffered
        from shader_ray_amd.tracer import PinnedFrame
        import numpy as np
        host_frame = np.empty((HEIGHT, WIDTH, 4), dtype=np.float32)   # a frame loop's own (pageable) buffer, reused
        scene.render(params, WIDTH, HEIGHT, SPP, out=host_frame)      # first touch of its pages
        t0 = time.perf_counter()
        for _ in range(10):
            scene.render(params, WIDTH, HEIGHT, SPP, out=host_frame)
        result["host_readback_mrays"] = round(WIDTH * HEIGHT * SPP * 10 / (time.perf_counter() - t0) / 1e6, 2)
        pinned = [PinnedFrame(WIDTH, HEIGHT) for _ in range(2)]
        scene2 = pkg.Scene(desc, env, device=local_rank)   # one in-flight readback per scene: two scenes double-buffer
        pair = [scene, scene2]
        for k in range(4):
            pair[k % 2].render_to_pinned(params, WIDTH, HEIGHT, SPP, pinned[k % 2], streams[k % lanes].cuda_stream, wait=False)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for k in range(20):
            pair[k % 2].render_to_pinned(params, WIDTH, HEIGHT, SPP, pinned[k % 2], streams[k % lanes].cuda_stream, wait=False)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        result["host_readback_pinned_mrays"] = round(WIDTH * HEIGHT * SPP * 20 / dt / 1e6, 2)
        result["host_readback_pinned_gbs"] = round(WIDTH * HEIGHT * 16 * 20 / dt / 1e9, 2)
        scene2.close()
        if not args.no_cpu_baseline:
            result["cpu_baseline"] = cpu_baseline(pkg, desc, env, params)
    if rank == 0:
        print(json.dumps(result), flush=True)
    if distributed:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
