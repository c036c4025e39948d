#!/usr/bin/env python3
"""Benchmark of the per-pixel hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W]

Workload (BASELINE.json configs[1]): bunny-class mesh (69,168 triangles, written as trisrc text and loaded
through the real parser + BVH builder), seeded HDR sky environment, 1920x1080, 1 spp, default gold material,
3 bounces.  A "step" is one frame of a trackball orbit (the frame loop of ray.cpp:1096-1131 with a mouse drag,
ray.cpp:879-918: ORBIT distinct views, replayed): scene and environment are resident in HBM before the timed
region.  Rank 0 prints ONE JSON line.

N = 1: the frame loop hands the C ABI four frames per launch (shray_render_batch_device) and alternates launches
over four HIP streams.  N > 1 (one process per GPU): every rank drives libshray_dist.so (shray_dist_step): a step of
the library carries 4 N consecutive frames (2 N in runs of fewer than 16 N) -- the rank's interleaved tiles of each in one launch, RGB tile buffers
exchanged with grouped ncclSend / ncclRecv over xGMI, frame f of the step de-interleaved on rank f % N (rotating
roots; --root-mode root0 gathers every frame on rank 0) -- and four such steps alternate on four streams and buffer
sets.  Exactly K frames are rendered in the timed region either way (the last launch is shorter when the frames per
launch do not divide K).  torch.distributed (gloo) is the control plane only: rendezvous, barrier, max-over-ranks.

N > 1 runs one process per GPU.  Either launch them yourself,
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \\
        --master-port P bench.py --gpus N --steps K --warmup W
or just run `python bench.py --gpus N ...`: without WORLD_SIZE in the environment this process -- which makes no GPU call
-- starts exactly that command as a child, relays rank 0's JSON line and exits with the child's code.

The N > 1 line times BOTH root modes in the same run (`value` = the mode named in config.parallelism, the other under
`alt_root_mode`) and carries `rccl_ranks` (ncclCommCount of the communicator the pixels travelled on), `frames_verified`
(assembled frames compared bit for bit with single-GPU renders of the whole frame, after the timed region) and
`transport_fallback`.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

# Independent launches are kept in flight on several HIP streams; they only overlap when the streams land on
# different hardware queues.  Must be set before the HIP runtime initialises.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

WIDTH, HEIGHT, SPP = 1920, 1080, 1
ORBIT = 20                 # distinct views of the orbit; frame k of a run is view k % ORBIT
ORBIT_DRAG = (0.025, 0.010)  # the mouse drag per frame, in window fractions (trackball_motion, ray.cpp:91-98)
HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: HBM3E peak 8.0 TB/s (spec)
# VALU issue peak (MI355X_MICROARCH.md): 256 CUs x 4 SIMDs, one wave64 VALU instruction per 2 cycles per SIMD, 2.4 GHz
VALU_PEAK_GINST = 256 * 4 * 2.4 / 2.0   # = 1228.8 G wave-instructions / s
PMC_FILE = os.path.join("profiles", "r04", "pmc_headline.json")   # written by profiles/make_pmc_json.py from rocprofv3 --pmc passes
ISA_COSTS = os.path.join("profiles", "r04", "isa_costs.json")     # written by profiles/isa_costs.py from the kernels' ISA
WARM_SECONDS = 0.15        # back-to-back frames before the first trial, beyond the W warm-up steps: the GPU's clock ramps


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def orbit_params(pkg, world, width, height, material=0):
    """ORBIT frame blocks: the start-up view dragged ORBIT_DRAG further every frame."""
    view = world.default_view()
    view.which_material = material
    out = []
    for _ in range(ORBIT):
        pkg.host.trackball_motion(view.object_rotation, *ORBIT_DRAG)
        out.append(world.frame_params(width, height, view))
    return out


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
            "sample": f"the full {WIDTH}x{HEIGHT} frame of the same workload (first view of the orbit), {reps} repetitions in "
                      f"{elapsed:.1f} s, CPU oracle with {threads} threads"}


def algorithmic_ops(counters, costs):
    """Lane-instructions of the shader's own arithmetic for the work the counters describe (profiles/isa_costs.py)."""
    c, k = counters, costs
    return (k["c_node"] * c["node_visits"] + k["c_tri_distance"] * c["triangle_tests"]
            + (k["c_tri_barycentric"] + k["c_shade"]) * c["shaded_hits"] + k["c_setup"] * c["traversals"]
            + k["c_env"] * c["env_lookups"] + k["c_pixel"] * c["samples"])


def launch_ranks(n):
    """`python bench.py --gpus N` without a launcher: this process has made no GPU call (torch is not even imported); it
    starts one rank per GPU with torch.distributed.run as a CHILD process (never an exec), passes its own arguments on,
    relays the child's output (rank 0 prints the JSON line) and returns the child's exit code -- non-zero if any rank fails
    (torch.distributed.run ends the other ranks and reports the failure)."""
    import socket
    import subprocess
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    log("bench.py: no WORLD_SIZE in the environment, launching", " ".join(cmd))
    return subprocess.call(cmd)


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
    ap.add_argument("--material", type=int, default=0, help="0 = gold (the headline), 6 = glazed plaster with diffuse white (configs[2])")
    ap.add_argument("--frames-in-flight", type=int, default=0,
                    help="independent launches alternate over this many HIP streams (1 = strictly one at a time; default: 4)")
    ap.add_argument("--frames-per-launch", type=int, default=0,
                    help="consecutive frames per launch (shray_render_batch_device / shray_dist_step).  Default: 4 for "
                         "N = 1, 4 N for N > 1 (a step then carries four frames' worth of pixels per GPU; 2 N when --steps < 16 N)")
    ap.add_argument("--root-mode", choices=["rotate", "root0"], default="rotate",
                    help="N > 1: rotate = frame f of a step is assembled on rank f % N (all-to-all over every xGMI link); "
                         "root0 = every frame on rank 0 (gather)")
    ap.add_argument("--rgba-wire", action="store_true", help="N > 1 only: exchange RGBA instead of RGB (alpha is the constant 1)")
    ap.add_argument("--same-view", action="store_true", help="every frame renders the first view of the orbit (round 2's loop)")
    args = ap.parse_args()
    WIDTH, HEIGHT, SPP = args.width, args.height, args.spp

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(launch_ranks(args.gpus))

    import torch
    import torch.distributed as dist

    from __graft_entry__ import load_package

    world_size = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world_size and not (world_size == 1 and args.gpus == 1):
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE is {world_size}: one process per GPU")
    # SHRAY_FORCE_DIST=1 rehearses the multi-GPU code path (process group, barrier, shray_dist_step) with one rank
    distributed = world_size > 1 or os.environ.get("SHRAY_FORCE_DIST") == "1"
    # Rehearsal on a single-GPU box: SHRAY_BENCH_ONE_GPU=1 puts every rank on cuda:0 and SHRAY_BENCH_TRANSPORT=gloo swaps
    # RCCL (which refuses two ranks on one GPU) for the library's CALLBACK transport over gloo, staging the tile buffers
    # through host memory.  Only the control flow is rehearsed that way, never a reported number.
    one_gpu = os.environ.get("SHRAY_BENCH_ONE_GPU") == "1"
    transport_name = os.environ.get("SHRAY_BENCH_TRANSPORT", "rccl")
    if one_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if distributed:
        # control plane only; the pixels travel over RCCL inside libshray_dist.so.  (gloo reports its connections on the
        # process's stdout -- "[Gloo] Rank 0 is connected to ..." --, which belongs to rank 0's ONE JSON line: while the
        # group comes up, file descriptor 1 points at stderr)
        sys.stdout.flush()
        saved_stdout = os.dup(1)
        os.dup2(2, 1)
        try:
            dist.init_process_group("gloo")
            dist.barrier()
        finally:
            sys.stdout.flush()
            os.dup2(saved_stdout, 1)
            os.close(saved_stdout)

    pkg = load_package()
    # rank 0 generates the scene file once; the others wait for it
    if rank == 0:
        path = pkg.scenes.bunny_trisrc()
    if distributed:
        dist.barrier()
    path = pkg.scenes.bunny_trisrc()
    t_load = time.perf_counter()
    world = pkg.World(path)
    t_flatten = time.perf_counter()
    desc = world.flatten()
    t_flattened = time.perf_counter()
    env = pkg.scenes.environment_hdr_sky(2048)
    orbit = orbit_params(pkg, world, WIDTH, HEIGHT, args.material)
    if args.same_view:
        orbit = [orbit[0]] * ORBIT
    t_create = time.perf_counter()
    scene = pkg.Scene(desc, env, device=local_rank)
    torch.cuda.synchronize()
    # file -> resident scene (the reference prints the same pieces: world.cpp:93-116)
    turnaround = {"parse_s": round(world.info.parse_seconds, 4), "bvh_build_s": round(world.info.build_seconds, 4),
                  "flatten_s": round(t_flattened - t_flatten, 4), "validate_repack_upload_s": round(time.perf_counter() - t_create, 4),
                  "load_world_s": round(t_flatten - t_load, 4), "triangles": int(world.triangle_count),
                  "host_threads": int(os.environ.get("SHRAY_LOAD_THREADS", 0)) or min(os.cpu_count() or 1, 32),
                  "what": "bunny-class trisrc (21 MB of text) -> triangle_set -> BVH -> get_shader_data arrays -> shray_scene_create + environment"}
    scene.set_kernel(args.kernel)

    from shader_ray_amd import multigpu

    tile = multigpu.DEFAULT_TILE
    # A GPU wants about eight frames' worth of rays in flight, at least two per launch (profiles/r03/loop_shapes.txt); a
    # rank of N renders 1 / N of every frame, so its launches carry 4 N frames (its tiles of each) over four streams and
    # buffer sets (profiles/r03/rank_share_shapes.txt: with N frames per launch on two streams one rank of 8 reached 5.5 x
    # of one GPU's rate before any exchange, with 4 N on four 7.6 x) -- never more than the K frames there are
    lanes = max(1, args.frames_in_flight or 4)
    if distributed:
        lanes = min(lanes, 4)      # buffer sets of a shray_dist object
    # A run of fewer than four such steps takes 2 N frames per step instead: its compute side is the same within noise
    # (profiles/r03/rank_share_steps.txt: a rank of 8 / 4 / 2 needs 0.65-0.73 / 1.24 / 2.44 ms for its share of 20 frames in
    # steps of 2 N against 0.63-0.66 / 1.22 / 2.43 in one), but only the LAST step's exchange is left uncovered by rendering
    # -- 4 of the driver's 20 frames at N = 8 instead of all 20 (9.3 MB per link, 0.09-0.19 ms of 0.9)
    per_step = 4 * world_size if args.steps >= 16 * world_size else 2 * world_size
    batch = max(1, min(64, args.frames_per_launch or (min(per_step, max(1, args.steps)) if distributed else 4)))
    streams = [torch.cuda.current_stream()] + [torch.cuda.Stream(device=device) for _ in range(lanes - 1)]
    frame_outs = None
    me = None
    ranks = {}             # root mode name -> multigpu.Rank (N > 1: both modes are timed, the chosen one first)
    transport_fallback = False
    rccl_ranks = None
    if distributed:
        mode_ids = {"rotate": multigpu.ROTATE, "root0": multigpu.ROOT0}

        def make_rank(root_mode, transport):
            """Collective.  (Rank or None, failure text or None) for `transport` = multigpu.RCCL / CALLBACK."""
            cfg = multigpu.make_config(rank, world_size, WIDTH, HEIGHT, SPP, batch, root_mode, transport, None, tile, tile,
                                       not args.rgba_wire, buffer_sets=lanes)
            if transport == multigpu.CALLBACK:
                return multigpu.Rank(scene, cfg, multigpu.HostExchange()), None
            failure, made = None, None
            try:
                ids = [multigpu.unique_id() if rank == 0 else None]
            except Exception as exc:   # noqa: BLE001
                ids, failure = [None], repr(exc)
            dist.broadcast_object_list(ids, src=0)
            if ids[0] is not None:
                try:
                    made = multigpu.Rank(scene, cfg, ids[0])      # collective: ncclCommInitRank
                except Exception as exc:   # noqa: BLE001
                    failure = repr(exc)
            else:
                failure = failure or "rank 0 could not make an RCCL id"
            failures = [None] * world_size
            dist.all_gather_object(failures, failure)
            if any(failures):
                if made is not None:
                    made.close()
                return None, f"ranks {[k for k, f in enumerate(failures) if f]}: " + next(f for f in failures if f)
            return made, None

        for name in [args.root_mode] + [m for m in mode_ids if m != args.root_mode]:
            if transport_name == "gloo":
                ranks[name], _ = make_rank(mode_ids[name], multigpu.CALLBACK)
                continue
            made, failure = (None, "an earlier communicator failed") if transport_fallback else make_rank(mode_ids[name], multigpu.RCCL)
            if made is None:
                # a communicator that does not come up must not cost the whole measurement: every rank falls back to the
                # host-staged exchange (slower: the tile buffers cross PCIe twice); the line says so at top level
                # (`transport_fallback`) and carries rccl_ranks = 0
                if rank == 0:
                    log("RCCL communicator failed:", failure)
                if not transport_fallback:
                    transport_name = "gloo, host-staged (RCCL failed: " + failure[:120] + ")"
                transport_fallback = True
                made, _ = make_rank(mode_ids[name], multigpu.CALLBACK)
            ranks[name] = made
        me = ranks[args.root_mode]
        rccl_ranks = min(r.world()[1] for r in ranks.values())
    else:
        frame_outs = [torch.empty(batch * HEIGHT * WIDTH * 4, dtype=torch.float32, device=device) for _ in range(lanes)]
    trials = max(1, args.trials)
    starts, stops = [], []
    EVENT_STRIDE = max(1, int(os.environ.get("SHRAY_BENCH_EVENT_STRIDE", "4")))
    torch.cuda.synchronize()

    active = {"rank": me}      # the shray_dist object the loop drives (N > 1: one per root mode)

    def views(first, count):
        return [orbit[(first + k) % ORBIT] for k in range(count)]

    def step(j, first, count, timed=None):
        """launch j: frames first .. first + count - 1 of the orbit; timed: HIP events bracket the launch"""
        lane = j % lanes
        st = streams[lane]
        # HIP events bracket every EVENT_STRIDE-th launch of the timed region (an event is a barrier packet on its stream:
        # bracketing every launch costs the one-frame-at-a-time form ~1 % of its step)
        timed = timed if (timed is not None and j % EVENT_STRIDE == 0) else None
        if timed is not None:
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            starts.append(a)
            stops.append(b)
            a.record(st)
        if distributed:
            active["rank"].step(views(first, count), lane, st.cuda_stream)
        elif count == 1:
            scene.render_into(orbit[first % ORBIT], WIDTH, HEIGHT, SPP, frame_outs[lane].data_ptr(), st.cuda_stream, None)
        else:
            scene.render_batch_into(views(first, count), WIDTH, HEIGHT, SPP, frame_outs[lane].data_ptr(), HEIGHT * WIDTH * 16,
                                    st.cuda_stream, None)
        if timed is not None:
            b.record(st)

    def run(frames, timed=None):
        """exactly `frames` frames"""
        done = j = 0
        while done < frames:
            count = min(batch, frames - done)
            step(j, done, count, timed)
            done += count
            j += 1

    def fence():
        if distributed:
            dist.barrier()
        torch.cuda.synchronize()

    def timed_trials():
        """W warm-up steps, the clock ramp, then `trials` repetitions of the timed region; each one is EXACTLY K steps
        between two fences (barrier + synchronize), its time the MAX over ranks.  Returns (seconds per trial, ms of the
        untimed ramp, frames of it)."""
        run(args.warmup)
        fence()
        # the clock of an idle GPU ramps for tens of milliseconds under load (round 2: ten 5 ms trials rose
        # monotonically); keep rendering, untimed, until WARM_SECONDS have passed
        t0 = time.perf_counter()
        ramp_frames = 0
        while True:
            run(ORBIT)             # whole periods of the orbit: a profile of this command averages over every view alike
            ramp_frames += ORBIT
            fence()
            go_on = time.perf_counter() - t0 < WARM_SECONDS
            if distributed:        # every rank leaves the ramp after the same number of (collective) steps
                flag = torch.tensor([1 if go_on else 0])
                dist.all_reduce(flag, op=dist.ReduceOp.MAX)
                go_on = bool(flag.item())
            if not go_on:
                break
        ramp_ms = (time.perf_counter() - t0) * 1e3
        seconds = []
        for trial in range(trials):
            fence()
            t0 = time.perf_counter()
            run(args.steps, timed=trial)
            fence()
            dt = time.perf_counter() - t0
            if distributed:
                t = torch.tensor([dt], dtype=torch.float64)
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                dt = float(t.item())
            seconds.append(dt)
        return seconds, ramp_ms, ramp_frames

    def verify_frames():
        """Outside the timed region: one more step, every frame this rank assembled compared bit for bit with a render
        of the WHOLE frame on this rank's GPU (N = 1: the frames of one launch of the loop against one-frame launches).
        Returns (frames compared over all ranks, frames that differed)."""
        whole = torch.empty(HEIGHT * WIDTH * 4, dtype=torch.float32, device=device)
        cur = torch.cuda.current_stream().cuda_stream
        compared = differing = 0
        if distributed:
            for r in ranks.values():
                r.step(views(0, batch), 0, streams[0].cuda_stream)
                mine = r.frames(0, batch, streams[0].cuda_stream)
                torch.cuda.synchronize()
                for f, got in mine.items():
                    scene.render_into(orbit[f % ORBIT], WIDTH, HEIGHT, SPP, whole.data_ptr(), cur, None)
                    torch.cuda.synchronize()
                    compared += 1
                    differing += 0 if torch.equal(got.reshape(-1), whole) else 1
            both = torch.tensor([compared, differing])
            dist.all_reduce(both)
            compared, differing = int(both[0]), int(both[1])
        else:
            step(0, 0, batch)
            torch.cuda.synchronize()
            for k in range(batch):
                scene.render_into(orbit[k % ORBIT], WIDTH, HEIGHT, SPP, whole.data_ptr(), cur, None)
                torch.cuda.synchronize()
                compared += 1
                differing += 0 if torch.equal(frame_outs[0][k * HEIGHT * WIDTH * 4:(k + 1) * HEIGHT * WIDTH * 4], whole) else 1
        return compared, differing

    trial_s, warm_ms, warm_frames = timed_trials()
    elapsed = sorted(trial_s)[len(trial_s) // 2]
    alt = None
    if distributed:
        for name, r in ranks.items():
            if name == args.root_mode:
                continue
            active["rank"] = r
            starts.clear()
            stops.clear()
            alt_s, _, _ = timed_trials()
            alt_elapsed = sorted(alt_s)[len(alt_s) // 2]
            alt = {"mode": name, "value": round(WIDTH * HEIGHT * SPP * args.steps / alt_elapsed / 1e6, 3), "unit": "Mrays/s",
                   "ms_per_step": round(alt_elapsed / args.steps * 1e3, 5), "trial_ms": [round(t * 1e3, 4) for t in alt_s],
                   "what": ("every frame gathered on rank 0 (north_star's gather; link-bound at 1 spp: DESIGN.md section 6)"
                            if name == "root0" else "frame f of a step assembled on rank f % N (all-to-all over every xGMI link)")}
        active["rank"] = me
    frames_compared, frames_differing = verify_frames()

    result = None
    frames_per_s = args.steps / elapsed
    if rank == 0:
        rays = WIDTH * HEIGHT * SPP * args.steps
        headline = (WIDTH, HEIGHT, SPP, args.material) == (1920, 1080, 1, 0)
        result = {
            "metric": "Mrays/s at 1920x1080 1spp (bunny.trisrc); 1/2/4/8-GPU scaling" if headline
            else f"Mrays/s at {WIDTH}x{HEIGHT} {SPP}spp (bunny.trisrc)",
            "value": round(rays / elapsed / 1e6, 3), "unit": "Mrays/s", "n_gpus": world_size, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 5), "higher_is_better": True,
            "trials": trials, "trial_ms": [round(t * 1e3, 4) for t in trial_s],
            "timing": f"median of {trials} trials of exactly {args.steps} steps, each between barrier + synchronize fences; "
                      f"after the {args.warmup} warm-up steps the loop ran untimed for {warm_ms:.0f} ms ({warm_frames} frames) so that the "
                      "clock has ramped before the first trial",
            "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "frames_verified": frames_compared - frames_differing, "frames_mismatched": frames_differing,
            "rccl_ranks": rccl_ranks, "transport_fallback": transport_fallback, "scene_turnaround": turnaround,
            "config": {"workload": "bunny-class trisrc (69,168 triangles, synthetic stand-in for bunny.trisrc) + seeded "
                                   f"2048x1024 HDR sky, {WIDTH}x{HEIGHT}, {SPP} spp, "
                                   + ("gold" if args.material == 0 else f"material {args.material}") + ", 3 bounces"
                                   + (" (BASELINE configs[1])" if headline else "")
                                   + (f"; the frames are a trackball orbit of {ORBIT} views, replayed" if not args.same_view else "; one view"),
                       "width": WIDTH, "height": HEIGHT, "spp": SPP, "kernel": {0: "stack", 1: "threaded", 2: "pool", 3: "stack, pair turns", 4: "wavefront"}[args.kernel],
                       "parallelism": (f"tiles{tile}x{tile}-interleaved-x{world_size}, libshray_dist ({transport_name}), "
                                       f"{'rotating roots (all-to-all)' if args.root_mode == 'rotate' else 'gather to rank 0'}")
                       if distributed else "single-gpu",
                       "frames_in_flight": lanes * batch, "frames_per_launch": batch, "streams": lanes,
                       "wire": ("rgb32f" if not args.rgba_wire else "rgba32f") if distributed else None},
        }

    # work counters of the orbit: the counting kernels' (the reference's traversals, equal to the CPU oracle's) and
    # the timed instances' own (shadow rays stop at their first hit; equal for the gold headline)
    if rank == 0:
        sums, timed_sums = {}, {}
        for view in (orbit[:1] if args.same_view else orbit):
            _, c = scene.render_counters(view, WIDTH, HEIGHT, SPP, want_image=False)
            _, ct = scene.render_counters_timed(view, WIDTH, HEIGHT, SPP, batch if not distributed else 1, want_image=False)
            for k in c:
                sums[k] = sums.get(k, 0) + c[k]
                timed_sums[k] = timed_sums.get(k, 0) + ct[k]
        nviews = 1 if args.same_view else ORBIT
        counters = {k: v / nviews for k, v in sums.items()}
        counters_timed = {k: v / nviews for k, v in timed_sums.items()}
        result["counters"] = {"per_frame_mean_over_the_orbit": counters, "of": "the counting kernels = the reference's traversal (every shadow "
                              "ray walked to its end), equal to the CPU oracle's"}
        result["counters_timed"] = {"per_frame_mean_over_the_orbit": counters_timed,
                                    "of": "the instance the timed launches run (shray_render_counters_timed)"}
        result["traversals_per_s"] = round(counters_timed["traversals"] * frames_per_s, 1)
        algo_bytes = pkg.tracer.algorithmic_bytes({k: int(v) for k, v in counters.items()}, WIDTH * HEIGHT, normals_fp16=True)
        costs, costs_note = None, ISA_COSTS
        try:
            costs = json.load(open(os.path.join(ROOT, ISA_COSTS)))
        except Exception as exc:   # noqa: BLE001
            costs_note = f"{ISA_COSTS} unreadable: {exc}"
        roof = {"bound": "valu_issue", "achieved": None, "peak": VALU_PEAK_GINST, "unit": "Gwaveinst/s", "frac": None,
                "traffic": None,
                "why": "the scene + environment working set (32 MB) is cache-resident: the kernel is co-limited by VALU issue (four "
                       "SIMDs per CU) and by the CU's one vector memory pipeline, not by HBM (DESIGN.md sections 4, 5); both are "
                       "reported (valu_issue, vector_memory), the headline pair is the busier one; HBM use is hbm_frac"}
        if costs:
            # wave-instructions a frame would take if every lane of every instruction did arithmetic the shader asks for
            ops = algorithmic_ops(counters_timed, costs) / 64.0
            gops = ops * frames_per_s / 1e9 / (world_size if distributed else 1)
            roof["algorithmic_ops"] = {
                "wave_insts_per_frame": round(ops, 1), "gwaveinst_per_s_per_gpu": round(gops, 2),
                "costs": {k: v for k, v in costs.items() if k.startswith("c_")}, "costs_source": costs_note,
                "formula": "(c_node Nv + c_tri_distance Tt + (c_tri_barycentric + c_shade) H + c_setup Tr + c_env E + c_pixel S) / 64 "
                           "from counters_timed; tests that reach the barycentric part are counted as H (a lower bound)"}
            roof["necessary_frac"] = round(gops / VALU_PEAK_GINST, 5)
        if not distributed:
            # per-launch kernel time: HIP events recorded around every EVENT_STRIDE-th launch of every trial, on the stream that
            # launch went to.  With frames_in_flight > 1 several launches share the GPU, so each lasts longer than it
            # would alone while together they finish sooner: rates below use WALL time, not per-launch time.
            kernel_ms = sorted(s.elapsed_time(e) for s, e in zip(starts, stops))
            avg_ms = sum(kernel_ms) / len(kernel_ms)   # per LAUNCH (a launch carries `batch` frames)
            # hardware counters of the dominant kernel come from a committed rocprofv3 --pmc run of THIS command
            # (they cannot be read from inside the process); used only if they were taken on this workload
            pmc, pmc_note = None, "no counter file"
            try:
                sys.path.insert(0, os.path.join(ROOT, "profiles"))
                from buildhash import kernel_source_hash
                cand = json.load(open(os.path.join(ROOT, PMC_FILE)))
                wl = cand["workload"]
                if (wl["width"], wl["height"], wl["spp"], wl["kernel_id"], wl.get("frames_per_launch", 1), wl.get("orbit", 1)) == \
                        (WIDTH, HEIGHT, SPP, args.kernel, batch, 1 if args.same_view else ORBIT) and cand["valu_insts_per_launch"] \
                        and args.material == 0:
                    pmc = cand
                    pmc_note = PMC_FILE + (" (same kernel sources as this build)" if cand["build_hash"] == kernel_source_hash()
                                           else " (STALE: measured on different kernel sources than this build)")
                else:
                    pmc_note = PMC_FILE + " is for another workload"
            except Exception as exc:   # noqa: BLE001
                pmc_note = f"{PMC_FILE} unreadable: {exc}"
            roof.update({"traffic_source": pmc_note, "counter_source": pmc_note, "lane_util": None, "hbm_frac": None})
            if pmc:
                fpl = pmc["workload"].get("frames_per_launch", 1)   # the profiled launches carried this many frames each
                ginst = pmc["valu_insts_per_launch"] / fpl * frames_per_s / 1e9
                roof.update({"achieved": round(ginst, 2), "frac": round(ginst / VALU_PEAK_GINST, 5),
                             "lane_util": round(pmc["lane_util"], 4) if pmc.get("lane_util") else None,
                             "useful_frac": round(ginst / VALU_PEAK_GINST * pmc["lane_util"], 5) if pmc.get("lane_util") else None,
                             "valu_insts_per_frame": pmc["valu_insts_per_launch"] / fpl,
                             "profiled_kernel_us": pmc.get("kernel_trace_avg_us")})
                if costs:
                    roof["necessary_of_issued"] = round(roof["algorithmic_ops"]["wave_insts_per_frame"] / (pmc["valu_insts_per_launch"] / fpl), 4)
                if pmc.get("hbm_bytes_per_launch"):
                    hbm_gbs = pmc["hbm_bytes_per_launch"] / fpl * frames_per_s / 1e9
                    roof.update({"traffic": pmc["hbm_bytes_per_launch"], "traffic_frames": fpl, "hbm_gbs": round(hbm_gbs, 2),
                                 "hbm_frac": round(hbm_gbs / HBM_PEAK_GBS, 5)})
            # The vector memory pipeline: one per CU (texture addresser + vector L1 + texture data), shared by the CU's four
            # SIMDs.  Its peak is measured HERE, on this GPU, by shray_probe_vector_cache: wave-instructions of 16 bytes per
            # lane per second when nothing else is done and every lane reads the same record (the pipeline's floor of ~14
            # cycles per instruction: profiles/r04/vector_cache_probe.json; lanes that read different cache lines cost more);
            # achieved = the kernel's vector-memory instructions (SQ_INSTS_VMEM_RD of the profiled run) x frames per second
            import ctypes as C
            probe_waves, probe_visits = 256 * 4 * 7 * 8, 512
            sec, nbytes = C.c_double(), C.c_uint64()
            pkg._native.check(pkg._native.load_hip().shray_probe_vector_cache(32768, 1, probe_visits, probe_waves, 0xffffffffffffffff, 32,
                                                                              C.byref(sec), C.byref(nbytes)))
            vmem_peak = probe_waves * probe_visits * 2 / sec.value / 1e9          # G wave-instructions / s
            vmem = {"peak": round(vmem_peak, 2), "unit": "Gwaveinst/s (16 bytes per lane)",
                    "peak_source": "shray_probe_vector_cache in this run: every lane of a wave-instruction at one 32-byte record of a "
                                   f"1 MB table, {probe_waves} one-wave workgroups x {probe_visits} visits x 2 loads in {sec.value * 1e3:.3f} ms",
                    "achieved": None, "frac": None}
            if pmc and pmc.get("vmem_insts_per_launch"):
                fpl = pmc["workload"].get("frames_per_launch", 1)
                gv = pmc["vmem_insts_per_launch"] / fpl * frames_per_s / 1e9
                vmem.update({"achieved": round(gv, 2), "frac": round(gv / vmem_peak, 5), "vmem_insts_per_frame": pmc["vmem_insts_per_launch"] / fpl,
                             "scalar_mem_insts_per_frame": (pmc.get("smem_insts_per_launch") or 0) / fpl,
                             "ta_busy_frac": round(pmc["ta_busy_frac"], 4) if pmc.get("ta_busy_frac") else None,
                             "td_busy_frac": round(pmc["td_busy_frac"], 4) if pmc.get("td_busy_frac") else None,
                             "note": "frac prices every instruction at the floor; ta_busy_frac / td_busy_frac (TA_BUSY_avr, TD_BUSY_avr over "
                                     "GRBM_GUI_ACTIVE / 8 of the profiled run) are how busy the pipeline was with what the lanes really read"})
                roof["wait_frac"] = round(pmc["wait_frac"], 4) if pmc.get("wait_frac") else None
            roof["vector_memory"] = vmem
            roof["valu_issue"] = {"achieved": roof.get("achieved"), "peak": VALU_PEAK_GINST, "frac": roof.get("frac"), "unit": "Gwaveinst/s",
                                  "busy_frac_profiled": round(pmc["valu_busy_frac_profiled"], 4) if pmc and pmc.get("valu_busy_frac_profiled") else None}
            # `frac` prices every vector instruction at the peak's 2 cycles.  Only f32 add / sub / mul, 32-bit integer add and
            # logic, register moves and VCC selects issue at that rate; fma, min / max, compares, other selects and anything
            # with a scalar-register operand take 4, transcendentals 8 (profiles/r04/valu_costs_probe.txt).  From the mix the
            # hardware counts (one more --pmc pass of the same command): the VALU's busy fraction lies between "only what is
            # counted as fma / transcendental is slow" and "only what is counted as f32 add / mul is fast"
            mix = pmc.get("valu_mix_per_launch") if pmc else None
            if mix and pmc.get("valu_insts_per_launch") and roof.get("frac"):
                total = pmc["valu_insts_per_launch"]
                fast_sure = mix.get("add_f32", 0.0) + mix.get("mul_f32", 0.0)
                fma, trans = mix.get("fma_f32", 0.0), mix.get("trans_f32", 0.0)
                lo = (2.0 * (total - fma - trans) + 4.0 * fma + 8.0 * trans) / (2.0 * total)
                hi = (2.0 * fast_sure + 8.0 * trans + 4.0 * (total - fast_sure - trans)) / (2.0 * total)
                roof["valu_issue"].update({
                    "frac_class_weighted": [round(roof["frac"] * lo, 4), round(roof["frac"] * hi, 4)],
                    "mix_of_vector_instructions": {k: round(v / total, 4) for k, v in mix.items()},
                    "frac_class_weighted_is": "frac x (cycles per instruction by class / 2): lower bound with only the counted fma and "
                                              "transcendental instructions at 4 and 8 cycles, upper bound with only the counted f32 add / mul at 2"})
            # The headline pair names the busier pipe.  Compared in ONE run, the profiled one (rocprofv3 runs a launch at a time
            # while it counts: there a four-frame launch has the GPU to itself and every pipe is less busy than in the timed
            # loop, whose launches overlap): VALU = 2 cycles x SQ_INSTS_VALU / (1024 SIMDs x the kernel's cycles), the vector
            # memory pipeline = the busier of its two stages (texture addresser, texture data)
            if pmc and pmc.get("valu_busy_frac_profiled") and (pmc.get("ta_busy_frac") or pmc.get("td_busy_frac")):
                stage, busy = max((("texture addresser", pmc.get("ta_busy_frac") or 0.0), ("texture data", pmc.get("td_busy_frac") or 0.0)),
                                  key=lambda kv: kv[1])
                if busy > pmc["valu_busy_frac_profiled"]:
                    roof.update({"bound": "vector_memory_pipeline", "frac": round(busy, 5), "peak": round(vmem_peak, 2),
                                 "achieved": round(busy * vmem_peak, 2), "unit": "Gwaveinst/s (16 bytes per lane)",
                                 "frac_is": f"the {stage} stage's busy fraction in the profiled run (VALU in the same run: "
                                            f"{pmc['valu_busy_frac_profiled']:.3f}); achieved = frac x the peak measured in this run; "
                                            "the timed loop's VALU fraction is valu_issue.frac"})
            # The fractions above are of the PROFILED run, where rocprofv3 runs one launch at a time (a four-frame launch alone:
            # serialized_launch_ms).  The timed loop overlaps launches and finishes a frame sooner; the work a frame hands each
            # pipe is the same, so the pipe's busy fraction in the timed loop is its busy cycles per frame over the timed
            # loop's cycles per frame (at the clock the profiled launches ran at): an estimate, reported beside the measurement
            if pmc and pmc.get("serialized_launch_ms") and pmc.get("serialized_clock_ghz"):
                fpl = pmc["workload"].get("frames_per_launch", 1)
                # (the launch's own cycles, GRBM_GUI_ACTIVE / 8 -- what the profiled fractions are fractions of --, at that clock)
                serial_ms = pmc["kernel_cycles_profiled"] / pmc["serialized_clock_ghz"] / 1e6 / fpl
                scale = serial_ms / (elapsed / args.steps * 1e3)
                est = {"serialized_ms_per_frame": round(serial_ms, 5), "clock_ghz": round(pmc["serialized_clock_ghz"], 3),
                       "is": "profiled busy fraction x (profiled time per frame / timed time per frame): same work per frame, less time"}
                for key, name in (("td_busy_frac", "texture_data_busy"), ("ta_busy_frac", "texture_addresser_busy")):
                    if pmc.get(key):
                        est[name] = round(pmc[key] * scale, 4)
                if roof["valu_issue"].get("frac_class_weighted"):
                    est["valu_busy_class_weighted"] = roof["valu_issue"]["frac_class_weighted"]
                roof["timed_loop_estimate"] = est
            roof.update({"kernel_ms_avg": round(avg_ms, 5), "kernel_ms_median": round(kernel_ms[len(kernel_ms) // 2], 5),
                         "concurrent_launches": lanes, "frames_per_launch": batch,
                         "kernel_events": f"{len(kernel_ms)} launches of the timed region bracketed (every {EVENT_STRIDE}th)"})
        else:
            roof["traffic_source"] = "not collected for N > 1 (see the N = 1 line)"
        algo_gbs = algo_bytes * frames_per_s / 1e9
        roof["algorithmic_cacheless"] = {
            "bytes_per_frame": algo_bytes, "bytes_per_ray": round(algo_bytes / (WIDTH * HEIGHT * SPP), 1),
            "gbs": round(algo_gbs, 2),
            "note": "SURVEY 8(d)'s cache-less count of the REFERENCE's fetches x frames / wall time (all GPUs together); these bytes are "
                    "served by L1/L2, not by HBM: not a bound (it exceeds the 8000 GB/s HBM peak), no fraction is formed from it"}
        result["roofline"] = roof

    if not distributed:
        # one frame at a time, in the same run: the latency form of the loop (one frame per launch, no second stream).
        # `ms`: 2 x ORBIT launches back to back on one stream (each starts when the one before it has drained);
        # `ms_host_synchronised`: the host waits for every frame before it submits the next
        solo = torch.empty(HEIGHT * WIDTH * 4, dtype=torch.float32, device=device)
        for k in range(5):
            scene.render_into(orbit[k % ORBIT], WIDTH, HEIGHT, SPP, solo.data_ptr(), streams[0].cuda_stream, None)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for k in range(2 * ORBIT):
            scene.render_into(orbit[k % ORBIT], WIDTH, HEIGHT, SPP, solo.data_ptr(), streams[0].cuda_stream, None)
        torch.cuda.synchronize()
        lat_ms = (time.perf_counter() - t0) / (2 * ORBIT) * 1e3
        lat = []
        for k in range(2 * ORBIT):
            t0 = time.perf_counter()
            scene.render_into(orbit[k % ORBIT], WIDTH, HEIGHT, SPP, solo.data_ptr(), streams[0].cuda_stream, None)
            torch.cuda.synchronize()
            lat.append(time.perf_counter() - t0)
        result["latency"] = {"ms": round(lat_ms, 5), "mrays": round(WIDTH * HEIGHT * SPP / lat_ms / 1e3, 2),
                             "ms_host_synchronised": round(sum(lat) / len(lat) * 1e3, 5),
                             "what": f"one frame per launch, one launch at a time on one stream: mean of {2 * ORBIT} frames of the orbit"}
        # the C ABI's host-buffer forms, PCIe-inclusive, for the record (never `value`): the blocking call into
        # pageable memory (the runtime's staged copy, into a buffer the loop reuses) and the stream form into pinned memory, double-buffered
        from shader_ray_amd.tracer import PinnedFrame
        import numpy as np
        host_frame = np.empty((HEIGHT, WIDTH, 4), dtype=np.float32)   # a frame loop's own (pageable) buffer, reused
        scene.render(orbit[0], WIDTH, HEIGHT, SPP, out=host_frame)    # first touch of its pages
        t0 = time.perf_counter()
        for k in range(10):
            scene.render(orbit[k % ORBIT], WIDTH, HEIGHT, SPP, out=host_frame)
        result["host_readback_mrays"] = round(WIDTH * HEIGHT * SPP * 10 / (time.perf_counter() - t0) / 1e6, 2)
        pinned = [PinnedFrame(WIDTH, HEIGHT) for _ in range(2)]
        scene2 = pkg.Scene(desc, env, device=local_rank)   # one in-flight readback per scene: two scenes double-buffer,
        pair = [scene, scene2]                             # each on its own stream (calls on one scene stay ordered)
        two = [torch.cuda.Stream(device=device) for _ in range(2)]
        for k in range(4):
            pair[k % 2].render_to_pinned(orbit[k % ORBIT], WIDTH, HEIGHT, SPP, pinned[k % 2], two[k % 2].cuda_stream, wait=False)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for k in range(20):
            pair[k % 2].render_to_pinned(orbit[k % ORBIT], WIDTH, HEIGHT, SPP, pinned[k % 2], two[k % 2].cuda_stream, wait=False)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        result["host_readback_pinned_mrays"] = round(WIDTH * HEIGHT * SPP * 20 / dt / 1e6, 2)
        result["host_readback_pinned_gbs"] = round(WIDTH * HEIGHT * 16 * 20 / dt / 1e9, 2)
        scene2.close()
        if not args.no_cpu_baseline:
            result["cpu_baseline"] = cpu_baseline(pkg, desc, env, orbit[0])
    if rank == 0:
        if alt is not None:
            result["alt_root_mode"] = alt
        print(json.dumps(result), flush=True)
    if distributed:
        dist.barrier()
        for r in ranks.values():
            r.close()
        dist.destroy_process_group()
    if frames_differing:
        raise SystemExit(f"{frames_differing} of {frames_compared} verified frames differ from the single-GPU render")


if __name__ == "__main__":
    main()
