"""The multi-GPU frame loop through its C ABI (include/shader_ray_dist.h) on ONE MI355X.

RCCL refuses two ranks on one device, so the N-rank step is rehearsed with the LOOPBACK transport: the ranks are
threads of this process, each with its own scene replica, HIP stream and shray_dist object, all on device 0, and the
tile buffers travel through the in-process hub.  Everything else is the product path: shray_render_batch_device on
the rank's tile set, the pack kernel, the transfer lists of the plan, shray_assemble_tiles_split_device.  Every
assembled frame must equal a single full-frame render bit for bit, in both root modes.  The RCCL transport itself is
exercised with one rank (communicator creation, a step without peers); a two-process gloo run covers the CALLBACK
transport and with it bench.py's rehearsal path."""
import os
import socket
import sys
import threading

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")


def step_frames(s, steps, frames):
    """frames of step s: the last of several steps is one frame short (a frame loop's tail)"""
    return frames - 1 if (steps > 1 and s == steps - 1 and frames > 1) else frames


def run_ranks(pkg, world_file, env, W, H, spp, ranks, frames, mode, shares, rgb, steps, materials):
    """`steps` steps of `frames` frames on `ranks` loopback ranks; returns {(step, frame): [H, W, 4] array}."""
    import torch
    from shader_ray_amd import multigpu
    hub = multigpu.Hub(ranks)
    world = pkg.World(world_file)
    desc = world.flatten()
    params = [[world.frame_params(W, H, material=materials[(s * frames + f) % len(materials)]) for f in range(frames)] for s in range(steps)]
    got, errors = {}, []
    lock = threading.Lock()
    barrier = threading.Barrier(ranks)

    def body(rank):
        try:
            torch.cuda.set_device(0)
            scene = pkg.Scene(desc, env, device=0)
            cfg = multigpu.make_config(rank, ranks, W, H, spp, frames, mode, multigpu.LOOPBACK, shares, 32, 32, rgb, buffer_sets=2)
            me = multigpu.Rank(scene, cfg, hub)
            streams = [torch.cuda.Stream(device=0) for _ in range(2)]
            barrier.wait()
            mine = {}
            for s in range(steps):
                me.step(params[s][:step_frames(s, steps, frames)], s % 2, streams[s % 2].cuda_stream)
                # frames are fetched a step late: the next step is already enqueued on the other stream / buffer set
                if s > 0:
                    for f, t in me.frames((s - 1) % 2, step_frames(s - 1, steps, frames), streams[(s - 1) % 2].cuda_stream).items():
                        mine[(s - 1, f)] = t
            for f, t in me.frames((steps - 1) % 2, step_frames(steps - 1, steps, frames), streams[(steps - 1) % 2].cuda_stream).items():
                mine[(steps - 1, f)] = t
            torch.cuda.synchronize()
            with lock:
                for k, t in mine.items():
                    assert k not in got, f"frame {k} assembled twice"
                    got[k] = t.cpu().numpy()
            barrier.wait()
            me.close()
            scene.close()
        except Exception as exc:   # noqa: BLE001
            errors.append((rank, repr(exc)))
            barrier.abort()

    threads = [threading.Thread(target=body, args=(r,)) for r in range(ranks)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=300)
    hub.close()
    assert not errors, errors
    return got, params


@pytest.mark.parametrize("ranks,frames,mode,shares,rgb", [
    (2, 1, "root0", None, True), (3, 2, "root0", (2, 3), True), (8, 8, "root0", None, True), (4, 3, "root0", (1, 1), False),
    (2, 2, "rotate", None, True), (3, 3, "rotate", None, False), (8, 8, "rotate", None, True), (3, 7, "rotate", None, True),
    (4, 2, "rotate", None, True)])
def test_loopback_ranks_assemble_the_full_frames(pkg, gpu, ranks, frames, mode, shares, rgb):
    from shader_ray_amd import multigpu
    W, H, spp = 333, 200, 2
    env = pkg.scenes.environment_hdr_sky(128)
    path = os.path.join(GOLDEN, "lobed_528.trisrc")
    steps = 3
    got, params = run_ranks(pkg, path, env, W, H, spp, ranks, frames, multigpu.ROTATE if mode == "rotate" else multigpu.ROOT0,
                            shares, rgb, steps, materials=(0, 6, 3))
    scene = pkg.Scene(pkg.World(path).flatten(), env, device=0)
    expected = set()
    for s in range(steps):
        for f in range(step_frames(s, steps, frames)):
            expected.add((s, f))
            want = scene.render(params[s][f], W, H, spp)
            assert np.array_equal(got[(s, f)].view(np.uint32), want.view(np.uint32)), (ranks, frames, mode, s, f)
    assert set(got) == expected
    scene.close()


def test_config5_shape_on_eight_loopback_ranks(pkg, gpu):
    """BASELINE configs[4] scaled to a quarter of the pixels: 1920x1080, 16 spp, eight ranks, both root modes."""
    import helpers
    from shader_ray_amd import multigpu
    env = pkg.scenes.environment_hdr_sky(256)
    path = helpers.bunny_trisrc()
    scene = pkg.Scene(pkg.World(path).flatten(), env, device=0)
    for mode in (multigpu.ROOT0, multigpu.ROTATE):
        got, params = run_ranks(pkg, path, env, 1920, 1080, 16, 8, 8 if mode == multigpu.ROTATE else 2, mode, None, True, 1, materials=(0, 6))
        for (s, f), frame in got.items():
            want = scene.render(params[s][f], 1920, 1080, 16)
            assert np.array_equal(frame.view(np.uint32), want.view(np.uint32)), (mode, f)
        assert len(got) == (8 if mode == multigpu.ROTATE else 2)
    scene.close()


def test_rccl_transport_with_one_rank(pkg, gpu):
    """ncclGetUniqueId / ncclCommInitRank behind shray_dist_create, and a step that has no peers."""
    import torch
    from shader_ray_amd import multigpu
    W, H = 200, 136
    world = pkg.World(os.path.join(GOLDEN, "lobed_528.trisrc"))
    scene = pkg.Scene(world.flatten(), pkg.scenes.environment_hdr_sky(128), device=0)
    for mode in (multigpu.ROOT0, multigpu.ROTATE):
        uid = multigpu.unique_id()          # one id per communicator
        assert len(uid) == 128 and any(uid)
        cfg = multigpu.make_config(0, 1, W, H, 1, 3, mode, multigpu.RCCL)
        me = multigpu.Rank(scene, cfg, uid)
        assert me.world() == (1, 1)         # (configured world, ncclCommCount of the communicator shray_dist_create made)
        params = [world.frame_params(W, H, material=m) for m in (0, 6, 2)]
        me.step(params, 0, torch.cuda.current_stream().cuda_stream)
        frames = me.frames(0, 3, torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        assert sorted(frames) == [0, 1, 2]
        for f, t in frames.items():
            assert np.array_equal(t.cpu().numpy(), scene.render(params[f], W, H, 1))
        me.close()
    N = pkg._native
    with pytest.raises(N.ShrayError):
        multigpu.Rank(scene, multigpu.make_config(0, 1, W, H, 1, 65), uid)          # more frames than SHRAY_MAX_BATCH (refused before RCCL)
    me = multigpu.Rank(scene, multigpu.make_config(0, 1, W, H, 1, 2), multigpu.unique_id())
    with pytest.raises(N.ShrayError):
        me.step([world.frame_params(W, H)] * 3, 0, 0)                                # more frames than the object was made for
    with pytest.raises(N.ShrayError):
        me.step([world.frame_params(W, H)], 5, 0)                                    # no such buffer set
    me.close()
    scene.close()


def test_one_buffer_set_driven_from_two_streams(pkg, gpu):
    """A buffer set is not tied to a stream: step k + 1 on the same set but another stream must wait, on
    the device, for step k's exchange and de-interleave before its render and pack overwrite the set's buffers.  Three
    loopback ranks, ONE buffer set, the stream alternating from step to step, nothing synchronised in between; every
    step's frames are copied out on the stream of that step."""
    import torch
    from shader_ray_amd import multigpu
    W, H, ranks, frames, steps = 333, 200, 3, 3, 6
    env = pkg.scenes.environment_hdr_sky(128)
    path = os.path.join(GOLDEN, "lobed_528.trisrc")
    world = pkg.World(path)
    desc = world.flatten()
    view = world.default_view()
    params = []
    for s in range(steps):
        row = []
        for f in range(frames):
            pkg.host.trackball_motion(view.object_rotation, 0.03, 0.01)
            row.append(world.frame_params(W, H, view, material=(0, 6, 3)[(s + f) % 3]))
        params.append(row)
    hub = multigpu.Hub(ranks)
    got, errors = {}, []
    lock = threading.Lock()
    barrier = threading.Barrier(ranks)

    def body(rank):
        try:
            torch.cuda.set_device(0)
            scene = pkg.Scene(desc, env, device=0)
            me = multigpu.Rank(scene, multigpu.make_config(rank, ranks, W, H, 1, frames, multigpu.ROTATE, multigpu.LOOPBACK, None, 32, 32,
                                                           True, buffer_sets=1), hub)
            streams = [torch.cuda.Stream(device=0) for _ in range(2)]
            barrier.wait()
            mine = {}
            for s in range(steps):
                st = streams[s % 2].cuda_stream
                me.step(params[s], 0, st)
                for f, t in me.frames(0, frames, st).items():
                    mine[(s, f)] = t
            torch.cuda.synchronize()
            with lock:
                got.update({k: t.cpu().numpy() for k, t in mine.items()})
            barrier.wait()
            me.close()
            scene.close()
        except Exception as exc:   # noqa: BLE001
            errors.append((rank, repr(exc)))
            barrier.abort()

    threads = [threading.Thread(target=body, args=(r,)) for r in range(ranks)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=300)
    hub.close()
    assert not errors, errors
    scene = pkg.Scene(desc, env, device=0)
    assert len(got) == steps * frames
    for (s, f), frame in got.items():
        assert np.array_equal(frame, scene.render(params[s][f], W, H, 1)), (s, f)
    scene.close()


@pytest.mark.gpu
@pytest.mark.parametrize("ranks", [1, 2])
def test_a_late_copy_out_holds_the_next_step_back(pkg, gpu, ranks):
    """the copy-out of a step's frames reads the set's output, which the set's NEXT step overwrites (a lone
    rank renders straight into it, the others de-interleave into it) -- from another stream if the caller says so.  Made
    deterministic: the copy-out's stream is held up for ~50 ms (a spin kernel) before the copy is enqueued, while the next
    step runs on a free stream.  The copy must still deliver the frames of ITS step: shray_dist_copy_output leaves the
    set's `finished` event behind the copy, and the next step waits for it."""
    import torch
    from shader_ray_amd import multigpu
    W, H, frames, steps = 333, 200, 2, 4
    env = pkg.scenes.environment_hdr_sky(128)
    world = pkg.World(os.path.join(GOLDEN, "lobed_528.trisrc"))
    desc = world.flatten()
    view = world.default_view()
    params = []
    for s in range(steps):
        row = []
        for f in range(frames):
            pkg.host.trackball_motion(view.object_rotation, 0.05, 0.02)
            row.append(world.frame_params(W, H, view, material=(0, 6)[(s + f) % 2]))
        params.append(row)
    hub = multigpu.Hub(ranks)
    got, errors = {}, []
    lock = threading.Lock()
    barrier = threading.Barrier(ranks)

    def body(rank):
        try:
            torch.cuda.set_device(0)
            scene = pkg.Scene(desc, env, device=0)
            me = multigpu.Rank(scene, multigpu.make_config(rank, ranks, W, H, 1, frames, multigpu.ROTATE, multigpu.LOOPBACK, None, 32, 32,
                                                           True, buffer_sets=1), hub)
            step_streams = [torch.cuda.Stream(device=0) for _ in range(2)]
            copy_stream = torch.cuda.Stream(device=0)
            barrier.wait()
            mine = {}
            for s in range(steps):
                me.step(params[s], 0, step_streams[s % 2].cuda_stream)
                with torch.cuda.stream(copy_stream):
                    torch.cuda._sleep(100_000_000)          # ~50 ms at the shader clock: the copy below starts late
                for f, t in me.frames(0, frames, copy_stream.cuda_stream).items():
                    mine[(s, f)] = t
            torch.cuda.synchronize()
            with lock:
                got.update({k: t.cpu().numpy() for k, t in mine.items()})
            barrier.wait()
            me.close()
            scene.close()
        except Exception as exc:   # noqa: BLE001
            errors.append((rank, repr(exc)))
            barrier.abort()

    threads = [threading.Thread(target=body, args=(r,)) for r in range(ranks)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=300)
    hub.close()
    assert not errors, errors
    scene = pkg.Scene(desc, env, device=0)
    assert len(got) == steps * frames
    for (s, f), frame in got.items():
        assert np.array_equal(frame, scene.render(params[s][f], W, H, 1)), (s, f)
    scene.close()


def test_bench_launches_its_own_ranks(gpu, tmp_path):
    """`python bench.py --gpus 2` with no launcher typed by hand: the parent makes no GPU call, starts
    torch.distributed.run as a child, relays rank 0's line.  On this one-GPU box both ranks share cuda:0 and the tile
    buffers travel over gloo (SHRAY_BENCH_ONE_GPU / SHRAY_BENCH_TRANSPORT: the rehearsal switches); the line must name
    both root modes, the verified frames and the (absent) RCCL communicator."""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env.update(SHRAY_BENCH_ONE_GPU="1", SHRAY_BENCH_TRANSPORT="gloo")
    run = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "6", "--warmup", "2", "--trials", "2",
                          "--width", "640", "--height", "360"], env=env, capture_output=True, text=True, timeout=900)
    assert run.returncode == 0, run.stderr[-3000:]
    lines = [ln for ln in run.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1 and lines[0].startswith("{"), run.stdout      # ONE line on stdout: rank 0's (gloo's chatter goes to stderr)
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["steps"] == 6 and line["value"] > 0
    assert line["frames_mismatched"] == 0 and line["frames_verified"] >= 8       # a step of 4 frames in each root mode
    assert line["rccl_ranks"] == 0 and line["transport_fallback"] is False       # gloo was asked for, nothing fell back
    assert "rotating roots" in line["config"]["parallelism"]
    assert line["alt_root_mode"]["mode"] == "root0" and line["alt_root_mode"]["value"] > 0
    # where a step's time goes, per rank, and what every directed link carries: both root modes
    for stages, mode in ((line["stages"], "rotate"), (line["alt_root_mode"]["stages"], "root0")):
        assert [row["rank"] for row in stages["per_rank"]] == [0, 1]
        for row in stages["per_rank"]:
            assert row["render_ms"] > 0 and row["exchange_ms"] >= 0 and row["assemble_ms"] >= 0, (mode, row)
        links = stages["link_bytes_per_step"]
        assert len(links) == 2 and links[0][0] == 0 and links[1][1] == 0 and links[1][0] > 0
        # ROOT0: nothing leaves rank 0; ROTATE: both directions carry tiles
        assert (links[0][1] == 0) == (mode == "root0"), (mode, links)
        assert stages["into_rank0_bytes_per_step"] == links[1][0] and stages["busiest_link"]["bytes_per_step"] == max(links[0][1], links[1][0])
    # a rank that fails must fail the command: an impossible frame size is refused by every rank's validation
    bad = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--trials", "1",
                          "--width", "0", "--height", "360"], env=env, capture_output=True, text=True, timeout=900)
    assert bad.returncode != 0


def test_bench_launches_four_ranks_and_survives_a_stalled_one(gpu):
    """The launcher form with more ranks, and the first-contact guard.  Four ranks share this box's one
    GPU (with the test process that is five processes on the card; the pool allows six, so the N = 8 form cannot be rehearsed
    here): uneven tile shares, a step of 8 frames, both root modes, every assembled frame verified.  Then the same command with
    rank 1 stalled in front of the communicators -- what a hung ncclCommInitRank looks like to the other ranks: the run ends
    within its --rank-timeout with the one-line record, a non-zero exit code and no process left."""
    import json
    import subprocess
    import time
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env.update(SHRAY_BENCH_ONE_GPU="1", SHRAY_BENCH_TRANSPORT="gloo")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--steps", "8", "--warmup", "2", "--trials", "1",
           "--width", "640", "--height", "360"]
    run = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert run.returncode == 0, run.stderr[-3000:]
    lines = [ln for ln in run.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1 and lines[0].startswith("{"), run.stdout
    line = json.loads(lines[0])
    assert line["n_gpus"] == 4 and line["value"] > 0 and line["frames_mismatched"] == 0 and line["frames_verified"] >= 16
    assert line["alt_root_mode"]["mode"] == "root0" and line["alt_root_mode"]["value"] > 0
    t0 = time.time()
    stalled = subprocess.run(cmd + ["--rank-timeout", "45"], env=dict(env, SHRAY_BENCH_STALL_RANK="1"), capture_output=True, text=True, timeout=600)
    took = time.time() - t0
    assert stalled.returncode != 0
    records = [ln for ln in stalled.stdout.splitlines() if ln.strip().startswith("{")]
    assert len(records) == 1, stalled.stdout
    record = json.loads(records[0])
    assert record["error"] == "rank timeout" and record["n_gpus"] == 4 and record["value"] is None
    # (over gloo the callback transport is made without a collective: rank 0 gets as far as the first exchange and waits there)
    assert any(word in record["where"] for word in ("communicator", "trials", "process")), record
    assert took < 45 + 150, took


def _gloo_worker(rank, world_size, port, mode, out_path):
    sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world_size)
    from __graft_entry__ import load_package
    pkg = load_package()
    from shader_ray_amd import multigpu
    torch.cuda.set_device(0)
    W, H = 200, 136
    world = pkg.World(os.path.join(GOLDEN, "lobed_528.trisrc"))
    scene = pkg.Scene(world.flatten(), pkg.scenes.environment_hdr_sky(128), device=0)
    frames = 2
    cfg = multigpu.make_config(rank, world_size, W, H, 1, frames, mode, multigpu.CALLBACK)
    me = multigpu.Rank(scene, cfg, multigpu.HostExchange())
    params = [world.frame_params(W, H, material=m) for m in (0, 6)]
    stream = torch.cuda.current_stream().cuda_stream
    me.step(params, 0, stream)
    mine = me.frames(0, frames, stream)
    torch.cuda.synchronize()
    ok = all(np.array_equal(t.cpu().numpy(), scene.render(params[f], W, H, 1)) for f, t in mine.items())
    everyone = [None] * world_size
    dist.all_gather_object(everyone, (sorted(mine), ok))
    if rank == 0:
        np.save(out_path, np.asarray([int(all(o for _, o in everyone)), sum(len(k) for k, _ in everyone)]))
    dist.barrier()
    me.close()
    scene.close()
    dist.destroy_process_group()


@pytest.mark.parametrize("mode", ["root0", "rotate"])
def test_callback_transport_over_gloo_two_processes(pkg, gpu, tmp_path, mode):
    """Two processes share the GPU; the tile buffers travel through host memory with gloo (multigpu.HostExchange),
    driven by the C library's CALLBACK transport -- the path bench.py's one-GPU rehearsal takes."""
    import torch.multiprocessing as mp
    from shader_ray_amd import multigpu
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    out = str(tmp_path / "ok.npy")
    mp.spawn(_gloo_worker, args=(2, port, multigpu.ROTATE if mode == "rotate" else multigpu.ROOT0, out), nprocs=2, join=True)
    ok, frames = np.load(out)
    assert ok == 1 and frames == 2


@pytest.mark.parametrize("ranks,root", [(3, "rotate"), (4, "root0"), (8, "rotate")])
def test_command_line_harness_on_several_ranks(pkg, gpu, tmp_path, ranks, root):
    """`shray_render -g N -t loopback -n frames`: the C++ frame loop (one thread per rank, steps of N frames, two buffer
    sets in flight) renders the same trackball animation as the single-GPU loop, frame for frame, and saves the last one."""
    import subprocess
    exe = os.path.join(ROOT, "shader-ray_amd", "tools", "shray_render")
    model = os.path.join(GOLDEN, "lobed_528.trisrc")
    W, H, frames = 160, 96, 2 * ranks + 3
    single, multi = str(tmp_path / "one"), str(tmp_path / "many")
    common = [exe, model, "grid", "-w", str(W), "-h", str(H), "-n", str(frames), "-s", "2"]
    subprocess.run(common + ["-o", single + ".ppm", "-f", single], check=True, capture_output=True, text=True)
    run = subprocess.run(common + ["-o", multi + ".ppm", "-f", multi, "-g", str(ranks), "-t", "loopback", "-r", root],
                         check=True, capture_output=True, text=True)
    assert f"{frames} frames on {ranks} GPUs" in run.stdout and "Mrays/s" in run.stdout, run.stdout
    for frame in range(frames):
        a = np.fromfile(f"{single}{frame:03d}.rgba", dtype=np.float32)
        b = np.fromfile(f"{multi}{frame:03d}.rgba", dtype=np.float32)
        assert a.size == W * H * 4 and np.array_equal(a.view(np.uint32), b.view(np.uint32)), frame
    assert open(single + ".ppm", "rb").read() == open(multi + ".ppm", "rb").read()
    # without -f only the last frame is read back
    subprocess.run(common + ["-o", multi + "2.ppm", "-g", str(ranks), "-t", "loopback", "-r", root], check=True, capture_output=True)
    assert open(single + ".ppm", "rb").read() == open(multi + "2.ppm", "rb").read()
