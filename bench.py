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
PMC_FILE = os.path.join("profiles", "r06", "pmc_headline.json")   # written by profiles/make_pmc_json.py from rocprofv3 --pmc passes
ISA_COSTS = os.path.join("profiles", "r06", "isa_costs.json")     # written by profiles/isa_costs.py from the kernels' ISA
VMEM_MIX = os.path.join("profiles", "r06", "vmem_class_mix.json")  # distinct records per vector-memory instruction, measured (histogram builds)
WARM_SECONDS = 0.15        # back-to-back frames before the first trial, beyond the W warm-up steps: the GPU's clock ramps


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def build_hash():
    """Hash of the device code of the library this run loads (profiles/buildhash.py), or None."""
    try:
        sys.path.insert(0, os.path.join(ROOT, "profiles"))
        from buildhash import kernel_source_hash
        return kernel_source_hash()
    except Exception:   # noqa: BLE001
        return None


def keyed_input(relative):
    """A committed input of the roofline object (per-unit instruction costs, the vector-memory class mix): (dict or None, note).  Every
    such file carries the hash of the device code it was derived from or measured on (`build_hash`), as the counter file does; the
    note says whether that is this run's device code -- STALE if not."""
    try:
        data = json.load(open(os.path.join(ROOT, relative)))
    except Exception as exc:   # noqa: BLE001
        return None, f"{relative} unreadable: {exc}"
    mine, theirs = build_hash(), data.get("build_hash")
    if theirs is None:
        return data, f"{relative} (STALE: carries no build hash)"
    if mine is None:
        return data, f"{relative} (build hash of this run unknown)"
    return data, relative + (" (same device code as this build)" if mine == theirs else
                             f" (STALE: derived from device code {theirs}, this build is {mine})")


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


LIVE_GROUPS = ["SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_INSTS_SALU SQ_INSTS_LDS", "FETCH_SIZE", "WRITE_SIZE"]


def live_counters(argv_tail, batch, budget_s=100.0):
    """Hardware counters of the dominant kernel from THIS invocation (VERDICT round 4, weak point 7): after the timed region, rank 0
    at N = 1 runs this script again as a child process under `rocprofv3 --pmc <group>` (one group per pass, nothing traced beside
    them; the program itself behind `--`, started as a child, never an exec) in its --counter-child form -- scene load, 2 x ORBIT
    frames of the timed loop's launches, exit -- and averages each counter over the kernel's launches that carried `batch` frames.
    Returns (dict like profiles/make_pmc_json.py writes, note) or (None, why not): any failure (no rocprofv3, a pass that
    does not end within its share of `budget_s`) falls back to the committed counter file."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    tool = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(tool):
        return None, "rocprofv3 is not on this box"
    began = time.time()
    sums, counts = {}, {}
    kernel = None
    for group in LIVE_GROUPS:
        left = budget_s - (time.time() - began)
        if left < 10.0:
            return None, f"the counter passes did not fit their {budget_s:.0f} s"
        out = tempfile.mkdtemp(prefix="shray_pmc_", dir="/tmp")
        cmd = [tool, "--pmc", *group.split(), "--output-format", "csv", "-d", out, "--", sys.executable, os.path.abspath(__file__),
               "--counter-child", "--steps", str(2 * ORBIT), "--warmup", "0"] + argv_tail
        child = subprocess.Popen(cmd, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp", GPU_MAX_HW_QUEUES="8"),
                                 stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, text=True, start_new_session=True)
        try:
            _, err = child.communicate(timeout=left)
        except subprocess.TimeoutExpired:
            _kill_tree(child.pid)          # the profiler AND the program it started
            shutil.rmtree(out, ignore_errors=True)
            return None, f"a counter pass ({group}) did not end within {left:.0f} s"
        if child.returncode != 0:
            shutil.rmtree(out, ignore_errors=True)
            return None, f"a counter pass ({group}) left with code {child.returncode}: {(err or '')[-200:]}"
        rows = 0
        for path in glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True):
            for r in csv.DictReader(open(path)):
                name = r.get("Kernel_Name", "")
                # the timed loop's kernel: the batch kernels of the stack tracer; whole launches only (grid = batch frames' worth)
                if "trace_stack_batch" not in name:
                    continue
                kernel = kernel or name.split("(")[0]
                key = r["Counter_Name"]
                sums[key] = sums.get(key, 0.0) + float(r["Counter_Value"])
                counts[key] = counts.get(key, 0) + 1
                rows += 1
        shutil.rmtree(out, ignore_errors=True)
        if not rows:
            return None, f"a counter pass ({group}) reported no launch of the tracer's kernel"
    avg = {k: sums[k] / counts[k] for k in sums}
    launches = counts.get("SQ_INSTS_VALU", 0)
    need = ("SQ_INSTS_VALU", "SQ_THREAD_CYCLES_VALU", "SQ_INSTS_VMEM_RD", "FETCH_SIZE", "WRITE_SIZE")
    if any(k not in avg for k in need) or launches * batch != 2 * ORBIT:
        return None, f"the counter passes saw {launches} launches of {batch} frames for {2 * ORBIT} frames"
    return {"kernel": kernel, "counters_per_launch": avg, "launches_averaged": launches,
            "workload": {"frames_per_launch": batch},
            "valu_insts_per_launch": avg["SQ_INSTS_VALU"], "lane_util": avg["SQ_THREAD_CYCLES_VALU"] / (avg["SQ_INSTS_VALU"] * 64.0),
            "vmem_insts_per_launch": avg["SQ_INSTS_VMEM_RD"], "smem_insts_per_launch": avg.get("SQ_INSTS_SMEM"),
            "hbm_bytes_per_launch": int((2 * avg["FETCH_SIZE"] + avg["WRITE_SIZE"]) * 1024)}, \
        (f"rocprofv3 --pmc passes of this invocation's own child runs ({len(LIVE_GROUPS)} passes, {launches} launches of {batch} frames = two "
         f"periods of the orbit each, {time.time() - began:.0f} s)")


def algorithmic_ops(counters, costs):
    """Lane-instructions of the shader's own arithmetic for the work the counters describe (profiles/isa_costs.py)."""
    c, k = counters, costs
    return (k["c_node"] * c["node_visits"] + k["c_tri_distance"] * c["triangle_tests"]
            + (k["c_tri_barycentric"] + k["c_shade"]) * c["shaded_hits"] + k["c_setup"] * c["traversals"]
            + k["c_env"] * c["env_lookups"] + k["c_pixel"] * c["samples"])


def _kill_tree(pid):
    """Ends process `pid`, its process group and every descendant (torch.distributed.run's workers): SIGTERM, five seconds, SIGKILL."""
    import signal
    victims = []
    try:
        import psutil
        root = psutil.Process(pid)
        victims = root.children(recursive=True) + [root]
    except Exception:   # noqa: BLE001
        pass
    for sig in (signal.SIGTERM, signal.SIGKILL):
        try:
            os.killpg(pid, sig)          # the child was started as the leader of its own process group
        except (ProcessLookupError, PermissionError):
            pass
        for v in victims:
            try:
                v.send_signal(sig)
            except Exception:   # noqa: BLE001
                pass
        deadline = time.time() + 5.0
        while time.time() < deadline and any(v.is_running() for v in victims):
            time.sleep(0.1)


def timeout_line(n, budget, where):
    return json.dumps({"error": "rank timeout", "n_gpus": n, "timeout_s": budget, "where": where, "value": None,
                       "metric": "Mrays/s at 1920x1080 1spp (bunny.trisrc); 1/2/4/8-GPU scaling", "unit": "Mrays/s"})


def launch_ranks(n, budget):
    """`python bench.py --gpus N` without a launcher: this process has made no GPU call (torch is not even imported); it
    starts one rank per GPU with torch.distributed.run as a CHILD process (never an exec), passes its own arguments on,
    relays the child's output (rank 0 prints the JSON line) and returns the child's exit code -- non-zero if any rank fails
    (torch.distributed.run ends the other ranks and reports the failure).  The child gets `budget` seconds of wall clock
    (--rank-timeout): a communicator that never comes up or an exchange that never ends must not hang the caller -- on expiry
    the child's process group and all its descendants are killed, ONE line {"error": "rank timeout", ...} goes to stdout and
    the exit code is 124."""
    import random
    import socket
    import subprocess

    def free_port():
        # below the kernel's ephemeral range (32768-60999): a port the kernel hands to nobody between this probe and the
        # child's listen() a second or two later (a port from bind(0) was taken in that gap once: EADDRINUSE, round 5)
        for _ in range(64):
            port = random.randrange(20000, 32000)
            with socket.socket() as sock:
                try:
                    sock.bind(("127.0.0.1", port))
                    return port
                except OSError:
                    continue
        with socket.socket() as sock:
            sock.bind(("127.0.0.1", 0))
            return sock.getsockname()[1]

    import threading

    deadline = time.time() + budget + float(os.environ.get("SHRAY_BENCH_PARENT_SLACK", "120"))   # (slack: a fresh box's first `import torch`)
    for attempt in range(2):
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
               "--master-port", str(free_port()), os.path.abspath(__file__)] + sys.argv[1:]
        log("bench.py: no WORLD_SIZE in the environment, launching", " ".join(cmd), f"(budget {budget:.0f} s)")
        # the ranks' own watchdogs (arm_watchdog) fire first and say where they were; this budget is the backstop behind them.
        # The child's stdout comes through a pipe: the parent relays it line by line and KNOWS whether a record went out, so
        # that exactly one does (a rank other than 0 whose timer fires first makes the launcher end rank 0
        # before rank 0 has spoken; a rank 0 that has printed its result and then hangs in the shutdown barrier must not add
        # a second record).
        started = time.time()
        child = subprocess.Popen(cmd, start_new_session=True, stdout=subprocess.PIPE, text=True, bufsize=1)
        records = []

        def relay(pipe=child.stdout, seen=records):
            for line in pipe:
                if line.lstrip().startswith("{"):
                    if seen:
                        log("bench.py: a second record from the ranks was dropped:", line.strip()[:200])
                        continue
                    seen.append(line)
                sys.stdout.write(line)
                sys.stdout.flush()

        reader = threading.Thread(target=relay, daemon=True)
        reader.start()
        try:
            rc = child.wait(timeout=max(1.0, deadline - time.time()) if budget > 0 else None)
        except subprocess.TimeoutExpired:
            log(f"bench.py: the ranks did not finish within {budget:.0f} s: killing them")
            _kill_tree(child.pid)
            reader.join(timeout=5.0)
            if records:      # the result went out and the ranks then hung in their shutdown: the run's line stands
                return 0
            print(timeout_line(n, budget, "parent: the child process group was killed"), flush=True)
            return 124
        reader.join(timeout=5.0)
        # a launcher that died at once (its rendezvous port taken after all) gets one more try on another port; a rank's own
        # failure takes longer than that (it has imported torch) and is final
        if rc == 0 or attempt == 1 or time.time() - started > 8.0:
            if rc != 0 and not records and budget > 0 and time.time() - started >= budget - 1.0:
                # a rank's watchdog fired and the launcher ended the others before rank 0 printed the record
                print(timeout_line(n, budget, f"parent: a rank's watchdog ended the run (launcher exit code {rc})"), flush=True)
                return 124
            return rc
        log(f"bench.py: the launcher left with code {rc} after {time.time() - started:.1f} s: once more on another port")
    return rc


_watchdog = {"stage": "start-up", "timer": None}


def arm_watchdog(rank, n, budget):
    """Inside a rank (also when a launcher other than launch_ranks started it, as the driver's torch.distributed.run does): a
    timer thread that, `budget` seconds after start-up, reports the stage the rank was in on stderr, has rank 0 print the
    one-line {"error": "rank timeout"} and leaves with os._exit(124) -- a hung ncclCommInitRank or grouped exchange blocks
    the main thread inside a C call, which no Python exception reaches."""
    import threading

    def fire():
        log(f"bench.py: rank {rank} of {n} is still in '{_watchdog['stage']}' after {budget:.0f} s: giving up")
        if rank == 0:
            print(timeout_line(n, budget, f"rank 0 was in: {_watchdog['stage']}"), flush=True)
        os._exit(124)

    if budget > 0:
        # rank 0 speaks first: the others fire three seconds later (their os._exit makes the launcher end rank 0)
        t = threading.Timer(budget if rank == 0 else budget + 3.0, fire)
        t.daemon = True
        t.start()
        _watchdog["timer"] = t


def stage(name):
    _watchdog["stage"] = name


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
    ap.add_argument("--no-live-counters", action="store_true",
                    help="N = 1: take the roofline's instruction and byte counts from the committed counter file only (default: "
                         "measure them in this invocation with short `rocprofv3 --pmc` child runs after the timed region, ~30 s)")
    ap.add_argument("--counter-child", action="store_true", help=argparse.SUPPRESS)   # the child run live_counters() profiles
    ap.add_argument("--rank-timeout", type=float, default=300.0,
                    help="N > 1: wall-clock budget in seconds for the whole multi-rank run; on expiry the ranks are killed, ONE line "
                         '{"error": "rank timeout", ...} is printed and the exit code is 124 (0 = no budget)')
    args = ap.parse_args()
    WIDTH, HEIGHT, SPP = args.width, args.height, args.spp

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(launch_ranks(args.gpus, args.rank_timeout))

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
    if distributed and os.environ.get("SHRAY_BENCH_NO_RANK_WATCHDOG") != "1":
        arm_watchdog(rank, world_size, args.rank_timeout)
    # SHRAY_BENCH_STALL_RANK=k|all (a test knob): that rank stops for good -- at SHRAY_BENCH_STALL_AT=start (before its first GPU
    # call) or =communicator (the default: it never arrives at the communicators, what a hung ncclCommInitRank looks like to
    # the others); the run must end with the one-line timeout record (tests/test_multigpu_cpu.py, tests/test_gpu_dist.py)
    stalled = os.environ.get("SHRAY_BENCH_STALL_RANK") in (str(rank), "all")
    if stalled and os.environ.get("SHRAY_BENCH_STALL_AT") == "start":
        stage("stalled on purpose (SHRAY_BENCH_STALL_RANK, at start)")
        time.sleep(10 ** 6)
    stage("process group (gloo rendezvous)")
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

    stage("scene load")
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
    # A GPU wants about eight frames' worth of rays in flight, at least two per launch (profiles/history/r03/loop_shapes.txt); a
    # rank of N renders 1 / N of every frame, so its launches carry 4 N frames (its tiles of each) over four streams and
    # buffer sets (profiles/history/r03/rank_share_shapes.txt: with N frames per launch on two streams one rank of 8 reached 5.5 x
    # of one GPU's rate before any exchange, with 4 N on four 7.6 x) -- never more than the K frames there are
    lanes = max(1, args.frames_in_flight or 4)
    if distributed:
        lanes = min(lanes, 4)      # buffer sets of a shray_dist object
    # A run of fewer than four such steps takes 2 N frames per step instead: its compute side is the same within noise
    # (profiles/history/r03/rank_share_steps.txt: a rank of 8 / 4 / 2 needs 0.65-0.73 / 1.24 / 2.44 ms for its share of 20 frames in
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

        if stalled:
            stage("stalled on purpose (SHRAY_BENCH_STALL_RANK, in front of the communicators)")
            time.sleep(10 ** 6)
        for name in [args.root_mode] + [m for m in mode_ids if m != args.root_mode]:
            stage(f"communicator for root mode {name} (ncclCommInitRank / callback transport)")
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
        for r in ranks.values():
            r.set_timing(True)      # four stamps per step: where a step's time goes
    else:
        frame_outs = [torch.empty(batch * HEIGHT * WIDTH * 4, dtype=torch.float32, device=device) for _ in range(lanes)]
    trials = max(1, args.trials)
    starts, stops = [], []
    EVENT_STRIDE = max(1, int(os.environ.get("SHRAY_BENCH_EVENT_STRIDE", "4")))
    torch.cuda.synchronize()

    active = {"rank": me}      # the shray_dist object the loop drives (N > 1: one per root mode)
    stage_ms = []              # N > 1: per trial, this rank's [render, exchange, assemble] ms (timed_trials)

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
        stage_ms.clear()
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
                # the stage stamps of the LAST step each buffer set ran in this trial, averaged over the sets it used
                used = min(lanes, -(-args.steps // batch))
                times = [active["rank"].step_times(lane) for lane in range(used)]
                stage_ms.append([sum(t3[k] for t3 in times) / len(times) for k in range(3)])
            seconds.append(dt)
        return seconds, ramp_ms, ramp_frames

    def stages_report(trial_seconds, rank_obj):
        """Collective.  The stage times of the median trial on every rank, and the bytes each directed link carries per step."""
        median_trial = sorted(range(len(trial_seconds)), key=lambda k: trial_seconds[k])[len(trial_seconds) // 2]
        mine = {"rank": rank, "render_ms": round(stage_ms[median_trial][0], 4), "exchange_ms": round(stage_ms[median_trial][1], 4),
                "assemble_ms": round(stage_ms[median_trial][2], 4), "link_bytes_to_peer": rank_obj.link_bytes(batch)}
        everyone = [None] * world_size
        dist.all_gather_object(everyone, mine)
        links = [row["link_bytes_to_peer"] for row in everyone]
        busiest = max((links[a][b], a, b) for a in range(world_size) for b in range(world_size))
        return {"trial": median_trial, "frames_per_step": batch,
                "per_rank": [{k: v for k, v in row.items() if k != "link_bytes_to_peer"} for row in everyone],
                "link_bytes_per_step": links,
                "busiest_link": {"bytes_per_step": busiest[0], "from": busiest[1], "to": busiest[2],
                                 "bytes_per_frame": round(busiest[0] / batch, 1)},
                "into_rank0_bytes_per_step": sum(links[a][0] for a in range(world_size)),
                "what": "per rank, the last step of each buffer set in the median trial (shray_dist_step_times): render_ms = start -> render + pack "
                        "done, exchange_ms = -> the set's exchange done (what the step waited for the links, net of the other sets' work it "
                        "overlapped), assemble_ms = -> de-interleaved; link_bytes_per_step[a][b] = bytes rank a sends to rank b in a step (the plan)"}

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

    if args.counter_child:
        # what live_counters() profiles: exactly `steps` frames of the timed loop's launches (whole periods of the orbit), nothing else
        run(args.steps)
        fence()
        return
    stage("warm-up and timed trials (shray_dist_step: render, pack, grouped exchange, de-interleave)" if distributed else "timed trials")
    trial_s, warm_ms, warm_frames = timed_trials()
    elapsed = sorted(trial_s)[len(trial_s) // 2]
    alt = None
    stages = stages_report(trial_s, me) if distributed else None
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
                   "stages": stages_report(alt_s, r),
                   "what": ("every frame gathered on rank 0 (north_star's gather; link-bound at 1 spp: DESIGN.md section 6)"
                            if name == "root0" else "frame f of a step assembled on rank f % N (all-to-all over every xGMI link)")}
        active["rank"] = me
    stage("frame verification")
    frames_compared, frames_differing = verify_frames()
    stage("counters, roofline and the report")

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
        costs, costs_note = keyed_input(ISA_COSTS)
        gpus = world_size if distributed else 1
        # ---- the roofline object (definition frozen in round 5; DESIGN.md section 5 derives every number) ----------------
        # Three resources, each as achieved / peak with achieved = a COUNT per frame x this run's frames per second:
        #   hbm    bytes: SURVEY 8(d)'s cache-less count of the reference's fetches (algorithmic) and the bytes the HBM
        #          counters saw (measured); peak 8000 GB/s.  The algorithmic rate exceeds the peak (the working set is
        #          cache-resident), so north_star's ">= 60 % of the HBM roofline" has no meaning here; frac_measured is
        #          what the path really asks of HBM
        #   vmem   vector-memory wave-instructions (SQ_INSTS_VMEM_RD) against the rate the CU's vector memory pipeline
        #          sustains for the kernel's MEASURED mix of distinct records per instruction (probed in this run)
        #   valu   vector-ALU wave-instructions (SQ_INSTS_VALU) against 256 CUs x 4 SIMDs x 2.4 GHz / 2
        # Top level: bound / achieved / peak / unit / frac are those of the resource with the largest frac; traffic = the
        # measured HBM bytes per launch.  Counts per frame come from committed rocprofv3 --pmc passes of THIS command on THIS
        # device code (instruction and byte counts of a deterministic kernel; keyed by the code objects' hash); the frame
        # rate, the launch durations (HIP events) and the vmem peak are measured in this run.  Busy counters of the
        # profiled run are reported under busy_profiled, named *_busy, and enter no frac.
        algo_gbs = algo_bytes * frames_per_s / 1e9
        roof = {"bound": None, "achieved": None, "peak": None, "unit": None, "frac": None, "traffic": None,
                "definition": "r05: frac = max over {hbm.frac_measured, vmem.frac, valu.frac}, each = count per frame x frames/s of this run / peak; "
                              "`bound` names the resource with the LARGEST of the three fractions -- it is not a claim about what binds the kernel "
                              "(R5.1 removed 38 % of the vector-memory instructions and the frame got slower; R6.2: without the scalar-cache path "
                              "it loses 10 %; DESIGN.md section 5: instruction issue of the waves' own streams, with both pipes two-thirds full)",
                "hbm": {"algorithmic_bytes_per_frame": algo_bytes, "algorithmic_bytes_per_ray": round(algo_bytes / (WIDTH * HEIGHT * SPP), 1),
                        "algorithmic_gbs": round(algo_gbs, 2), "peak_gbs": HBM_PEAK_GBS * gpus,
                        "algorithmic_over_peak": round(algo_gbs / (HBM_PEAK_GBS * gpus), 4),
                        "measured_bytes_per_frame": None, "measured_gbs": None, "frac_measured": None,
                        "north_star_60_percent": "not applicable: the algorithmic bytes (32 Nv + 8 Lv + 36 Tt + 18 H + 48 E + 16 P, SURVEY 8(d)) are "
                                                 "served by the vector L1 / L2 (scene + environment = 32 MB working set), so their rate exceeds "
                                                 "the HBM peak (algorithmic_over_peak > 1) while HBM itself carries frac_measured of its peak"},
                "vmem": {"insts_per_frame": None, "insts_per_s_g": None, "peak_g": None, "frac": None},
                "valu": {"insts_per_frame": None, "insts_per_s_g": None, "peak_g": VALU_PEAK_GINST, "frac": None, "necessary_frac": None,
                         "lane_util": None}}
        if costs:
            # wave-instructions a frame would take if every lane of every instruction did arithmetic the shader asks for
            ops = algorithmic_ops(counters_timed, costs) / 64.0
            gops = ops * frames_per_s / 1e9 / gpus
            roof["valu"]["necessary_frac"] = round(gops / VALU_PEAK_GINST, 5)
            roof["valu"]["necessary"] = {
                "wave_insts_per_frame": round(ops, 1), "costs": {k: v for k, v in costs.items() if k.startswith("c_")}, "costs_source": costs_note,
                "formula": "(c_node Nv + c_tri_distance Tt + (c_tri_barycentric + c_shade) H + c_setup Tr + c_env E + c_pixel S) / 64 "
                           "from counters_timed; tests that reach the barycentric part are counted as H (a lower bound)"}
        if not distributed:
            # per-launch kernel time: HIP events recorded around every EVENT_STRIDE-th launch of every trial, on the stream that
            # launch went to.  With frames_in_flight > 1 several launches share the GPU, so each lasts longer than it
            # would alone while together they finish sooner: rates use WALL time, not per-launch time.
            kernel_ms = sorted(s.elapsed_time(e) for s, e in zip(starts, stops))
            avg_ms = sum(kernel_ms) / len(kernel_ms)   # per LAUNCH (a launch carries `batch` frames)
            roof.update({"kernel_ms_avg": round(avg_ms, 5), "kernel_ms_median": round(kernel_ms[len(kernel_ms) // 2], 5),
                         "concurrent_launches": lanes, "frames_per_launch": batch,
                         "kernel_events": f"{len(kernel_ms)} launches of the timed region bracketed by HIP events on their own streams (every {EVENT_STRIDE}th)"})
            pmc, pmc_note = None, "no counter file"
            kernel_source_hash = lambda: None   # noqa: E731  (replaced below; a counter file never carries None)
            try:
                sys.path.insert(0, os.path.join(ROOT, "profiles"))
                from buildhash import kernel_source_hash
                cand = json.load(open(os.path.join(ROOT, PMC_FILE)))
                wl = cand["workload"]
                if (wl["width"], wl["height"], wl["spp"], wl["kernel_id"], wl.get("frames_per_launch", 1), wl.get("orbit", 1)) == \
                        (WIDTH, HEIGHT, SPP, args.kernel, batch, 1 if args.same_view else ORBIT) and cand["valu_insts_per_launch"] \
                        and args.material == 0:
                    pmc = cand
                    pmc_note = PMC_FILE + (" (same device code as this build)" if cand["build_hash"] == kernel_source_hash()
                                           else " (STALE: measured on other device code than this build's)")
                else:
                    pmc_note = PMC_FILE + " is for another workload"
            except Exception as exc:   # noqa: BLE001
                pmc_note = f"{PMC_FILE} unreadable: {exc}"
            # the counts of THIS invocation where they can be had (the busy counters stay the archived run's: they are not rates)
            archived = pmc
            # (not when this process is itself being profiled -- profiles/run_profile.sh, the driver's own rocprofv3 run --: the
            # profiler's environment would be inherited by the children)
            profiled = any(k.startswith(("ROCPROF", "ROCP_TOOL", "ROCTRACER")) for k in os.environ) or "rocprof" in os.environ.get("LD_PRELOAD", "")
            if profiled:
                pmc_note += "; live counters not taken: this process is running under a profiler"
            if not profiled and not args.no_live_counters and os.environ.get("SHRAY_BENCH_LIVE_COUNTERS", "1") != "0":
                tail = ["--width", str(WIDTH), "--height", str(HEIGHT), "--spp", str(SPP), "--material", str(args.material),
                        "--kernel", str(args.kernel), "--frames-per-launch", str(batch), "--frames-in-flight", str(lanes)] + \
                       (["--same-view"] if args.same_view else [])
                live_began = time.time()
                try:
                    live, live_note = live_counters(tail, batch, budget_s=float(os.environ.get("SHRAY_BENCH_LIVE_BUDGET", "60")))
                except Exception as exc:   # noqa: BLE001  (a measurement aid must not cost the run its line)
                    live, live_note = None, f"live_counters raised {exc!r}"
                # (what the counter passes add to the invocation's wall time, after the timed region, is in the line)
                roof["live_counter_passes_s"] = round(time.time() - live_began, 1)
                if live:
                    if archived and archived["build_hash"] == kernel_source_hash():
                        for key in ("td_busy_frac", "ta_busy_frac", "valu_busy_frac_profiled", "wait_frac", "serialized_launch_ms",
                                    "kernel_trace_avg_us", "valu_mix_per_launch"):
                            live[key] = archived.get(key)
                        if live.get("valu_mix_per_launch"):      # (a mix of the archived run's instructions: scale to this count)
                            scale = live["valu_insts_per_launch"] / archived["valu_insts_per_launch"]
                            live["valu_mix_per_launch"] = {k: v * scale for k, v in live["valu_mix_per_launch"].items()}
                        roof["archived_counters_agree"] = {
                            k: round(live[k] / archived[k], 5) for k in ("valu_insts_per_launch", "vmem_insts_per_launch", "hbm_bytes_per_launch")
                            if archived.get(k)}
                    pmc, pmc_note = live, live_note
                else:
                    pmc_note += f"; live counters not taken: {live_note}"
            roof["counter_source"] = pmc_note
            # The vector memory pipeline: one per CU (texture addresser + vector L1 + texture data), shared by the CU's four
            # SIMDs.  What it sustains depends on how many distinct records the lanes of an instruction read
            # (shray_probe_vector_cache: ~12.5 CU-cycles with every lane at one record, ~24 with two to four records spread over
            # the lanes, ~44 with eight, ~54 with every lane elsewhere).  The kernel's instructions are priced by the mix of
            # those classes that was MEASURED on this workload (profiles/r05/vmem_class_mix.json: histograms of distinct nodes
            # per wave-visit and of distinct leaves per leaf stage); each class's cost is probed here, in this run.
            import ctypes as C
            probe_waves, probe_visits = 256 * 4 * 7 * 8, 512
            hip = pkg._native.load_hip()

            def probe(spread, nbytes_lane, neighbours=False):
                """seconds per wave-instruction, chip-wide; neighbours: the lanes that share a record are runs of neighbouring lanes"""
                sec, nb = C.c_double(), C.c_uint64()
                pkg._native.check(hip.shray_probe_vector_cache(32768, spread | (0x80000000 if neighbours and spread > 1 else 0), probe_visits, probe_waves,
                                                               0xffffffffffffffff, nbytes_lane, C.byref(sec), C.byref(nb)))
                return sec.value / (probe_waves * probe_visits * (2 if nbytes_lane == 32 else 1))

            vm = roof["vmem"]
            try:
                mix, mix_note = keyed_input(VMEM_MIX)
                if mix is None:
                    raise RuntimeError(mix_note)
                per_inst, per_inst_scattered, classes = 0.0, 0.0, {}
                for kind in mix["kinds"]:
                    for records, share in kind["records_per_instruction"].items():
                        t_near = probe(int(records), kind["probe_bytes_per_lane"], neighbours=True)
                        t_far = probe(int(records), kind["probe_bytes_per_lane"])
                        classes[f"{kind['name']}:{records}"] = {"share": round(kind["share_of_insts"] * share, 4),
                                                                 "cu_cycles_per_inst": round(t_near * 256 * 2.4e9, 2),
                                                                 "cu_cycles_per_inst_scattered": round(t_far * 256 * 2.4e9, 2)}
                        per_inst += kind["share_of_insts"] * share * t_near
                        per_inst_scattered += kind["share_of_insts"] * share * t_far
                floor = probe(1, 32)
                vm.update({"peak_g": round(1.0 / per_inst / 1e9, 2), "peak_scattered_g": round(1.0 / per_inst_scattered / 1e9, 2),
                           "peak_at_one_record_g": round(1.0 / floor / 1e9, 2), "classes": classes, "mix_source": mix_note,
                           "peak_is": "1 / sum over classes (share x seconds per wave-instruction of that class), the classes = the kernel's measured "
                                      "mix of distinct records per instruction, their costs probed in this run (shray_probe_vector_cache, 1 MB table): "
                                      "peak_g with the lanes that share a record in runs of neighbouring lanes (a wave is an 8x8 pixel tile: "
                                      "rays at one node are neighbours), peak_scattered_g with a pseudo-random lane per record (the pipeline's worst "
                                      "case for that many records), peak_at_one_record_g with every lane at one record (the floor rounds 1-4 priced "
                                      "against).  frac = insts_per_s / peak_g; frac_scattered is its upper bound.  R5.1 (38 % of these instructions "
                                      "removed, no gain) says the truth is near frac"})
            except Exception as exc:   # noqa: BLE001
                vm["mix_source"] = f"{VMEM_MIX} unusable: {exc}"
            if pmc:
                fpl = pmc["workload"].get("frames_per_launch", 1)   # the profiled launches carried this many frames each
                vi = pmc["valu_insts_per_launch"] / fpl
                roof["valu"].update({"insts_per_frame": vi, "insts_per_s_g": round(vi * frames_per_s / 1e9, 2),
                                     "frac": round(vi * frames_per_s / 1e9 / VALU_PEAK_GINST, 5),
                                     "lane_util": round(pmc["lane_util"], 4) if pmc.get("lane_util") else None})
                if costs:
                    roof["valu"]["necessary_of_issued"] = round(roof["valu"]["necessary"]["wave_insts_per_frame"] / vi, 4)
                mixv = pmc.get("valu_mix_per_launch")
                if mixv:
                    roof["valu"]["mix"] = {k: round(v / pmc["valu_insts_per_launch"], 4) for k, v in mixv.items()}
                if pmc.get("vmem_insts_per_launch"):
                    mi = pmc["vmem_insts_per_launch"] / fpl
                    vm.update({"insts_per_frame": mi, "insts_per_s_g": round(mi * frames_per_s / 1e9, 3),
                               "scalar_mem_insts_per_frame": (pmc.get("smem_insts_per_launch") or 0) / fpl})
                    if vm.get("peak_g"):
                        vm["frac"] = round(mi * frames_per_s / 1e9 / vm["peak_g"], 5)
                        vm["frac_scattered"] = round(mi * frames_per_s / 1e9 / vm["peak_scattered_g"], 5)
                        vm["frac_at_one_record_peak"] = round(mi * frames_per_s / 1e9 / vm["peak_at_one_record_g"], 5)
                if pmc.get("hbm_bytes_per_launch"):
                    hb = pmc["hbm_bytes_per_launch"] / fpl
                    roof["hbm"].update({"measured_bytes_per_frame": hb, "measured_gbs": round(hb * frames_per_s / 1e9, 2),
                                        "frac_measured": round(hb * frames_per_s / 1e9 / HBM_PEAK_GBS, 5),
                                        "measured_over_algorithmic": round(hb / algo_bytes, 5),
                                        "measured_is": "(2 x FETCH_SIZE + WRITE_SIZE) KiB per launch / frames per launch, MI355X_MICROARCH.md's gfx950 correction"})
                    roof["traffic"] = pmc["hbm_bytes_per_launch"]
                    roof["traffic_frames"] = fpl
                # busy counters of the PROFILED run (rocprofv3 runs one launch at a time while it counts: a four-frame launch alone on
                # the GPU); none of them is a fraction of a peak rate
                roof["busy_profiled"] = {
                    "texture_data_busy": round(pmc["td_busy_frac"], 4) if pmc.get("td_busy_frac") else None,
                    "texture_addresser_busy": round(pmc["ta_busy_frac"], 4) if pmc.get("ta_busy_frac") else None,
                    "valu_busy_at_2_cycles": round(pmc["valu_busy_frac_profiled"], 4) if pmc.get("valu_busy_frac_profiled") else None,
                    "wave_cycles_waiting": round(pmc["wait_frac"], 4) if pmc.get("wait_frac") else None,
                    "serialized_launch_ms": pmc.get("serialized_launch_ms"), "profiled_kernel_us": pmc.get("kernel_trace_avg_us"),
                    "of": "TD_TD_BUSY_sum / 256 CUs, TA_TA_BUSY_sum / 256, 2 x SQ_INSTS_VALU / 1024 SIMDs over GRBM_GUI_ACTIVE / 8; SQ_WAIT_ANY / SQ_WAVE_CYCLES"}
            pairs = [("hbm", roof["hbm"]["frac_measured"], roof["hbm"]["measured_gbs"], HBM_PEAK_GBS, "GB/s"),
                     ("vmem", vm.get("frac"), vm.get("insts_per_s_g"), vm.get("peak_g"), "Gwaveinst/s (vector memory)"),
                     ("valu", roof["valu"]["frac"], roof["valu"]["insts_per_s_g"], VALU_PEAK_GINST, "Gwaveinst/s (vector ALU)")]
            pairs = [q for q in pairs if q[1] is not None]
            if pairs:
                name, frac, achieved, peak, unit = max(pairs, key=lambda q: q[1])
                roof.update({"bound": name, "frac": frac, "achieved": achieved, "peak": peak, "unit": unit})
        else:
            roof["counter_source"] = "not collected for N > 1 (see the N = 1 line)"
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
        if stages is not None:
            result["stages"] = stages
        print(json.dumps(result), flush=True)
        # the record is out: from here on the watchdog must not add a second one (a hang in the shutdown barrier is the
        # launcher's or the parent's to end)
        if _watchdog["timer"] is not None:
            _watchdog["timer"].cancel()
            _watchdog["timer"] = None
    if distributed:
        stage("shutdown")
        dist.barrier()
        for r in ranks.values():
            r.close()
        dist.destroy_process_group()
    if _watchdog["timer"] is not None:
        _watchdog["timer"].cancel()
    if frames_differing:
        raise SystemExit(f"{frames_differing} of {frames_compared} verified frames differ from the single-GPU render")


if __name__ == "__main__":
    main()
