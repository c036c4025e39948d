"""The multi-GPU frame loop without a GPU: the C plan (libshray_dist.so: who renders which tiles, who assembles which
frame, which bytes travel where) checked against a numpy statement of the tile mapping, and world-size-2 / 3 runs
under gloo in which every rank executes ITS transfer lists from the C plan with torch.distributed point-to-point
calls (multigpu.HostExchange -- the same class the CALLBACK transport uses on a GPU box).  The CPU oracle stands in for
the renderer of a rank's tiles (this is a test: the product path has no CPU fallback); every assembled frame must equal
a single full-frame oracle render bit for bit, in both root modes."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
W, H, TILE = 88, 56, 16      # 6 x 4 tiles, the right and top edges are partial


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def pack_like_the_kernel(cfg, plan, rendered, count, multigpu):
    """numpy restatement of dist_capi.hip: pack_tiles_kernel -- frame f of the step goes to the wire buffer, or, when
    this rank assembles it, straight into its own row of its gather buffer."""
    c = plan.channels
    pixels = plan.owned_tiles * cfg.tile_w * cfg.tile_h
    wire = np.zeros(plan.wire_frame_stride_bytes * cfg.max_frames, dtype=np.uint8)
    gather = np.zeros(plan.gather_rank_stride_bytes * cfg.world, dtype=np.uint8) if plan.max_assembled else np.zeros(0, np.uint8)
    for f in range(count):
        body = np.ascontiguousarray(rendered[f].reshape(-1, 4)[:pixels, :c]).view(np.uint8).reshape(-1)
        if multigpu.frame_owner(cfg, f) == cfg.rank:
            slot = f // cfg.world if cfg.root_mode == multigpu.ROTATE else f
            at = cfg.rank * plan.gather_rank_stride_bytes + slot * plan.gather_frame_stride_bytes
            gather[at:at + body.size] = body
        else:
            at = f * plan.wire_frame_stride_bytes
            wire[at:at + body.size] = body
    return wire, gather


def worker(rank, world, port, out_path, frames, rgb_wire, shares, root_mode):
    sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from __graft_entry__ import load_package
    import helpers
    import oracle
    pkg = load_package()
    from shader_ray_amd import multigpu

    scene_world = pkg.World(helpers.small_trisrc())
    desc = scene_world.flatten()
    env = pkg.scenes.environment_hdr_sky(64)
    # every frame of a step has its own parameters (here: its own material)
    params = [scene_world.frame_params(W, H, material=(6, 0, 3)[k % 3]) for k in range(frames)]
    wanted = [oracle.render(desc, env, p, W, H, 1, threads=2)[0] for p in params]

    cfg = multigpu.make_config(rank, world, W, H, 1, frames, root_mode, multigpu.CALLBACK, shares, TILE, TILE, rgb_wire)
    plan = multigpu.plan(cfg)
    got_shares = (plan.rank0_phases, plan.other_phases)
    if shares is None and world > 1 and root_mode == multigpu.ROOT0:
        assert got_shares[0] < got_shares[1]     # rank 0 owns the smaller share by default
    if root_mode == multigpu.ROTATE:
        assert got_shares == (1, 1)

    def render_tiles(count):
        # stand-in for shray_render_batch_device: this rank's packed tiles of `count` frames from the oracle
        ts = plan.tiles
        tiles, tiles_x, _ = multigpu.owned_tiles(W, H, ts.tile_w, ts.tile_h, ts.tile_stride, ts.tile_phase, ts.tile_phase_count)
        assert len(tiles) == plan.owned_tiles
        out = np.zeros((count, plan.max_tiles, ts.tile_h, ts.tile_w, 4), dtype=np.float32)   # (0, 0, 0, 0) outside the frame
        for f in range(count):
            for k, t in enumerate(tiles):
                x0, y0 = (t % tiles_x) * ts.tile_w, (t // tiles_x) * ts.tile_h
                w, h = min(ts.tile_w, W - x0), min(ts.tile_h, H - y0)
                out[f, k, :h, :w] = wanted[f][y0:y0 + h, x0:x0 + w]
        return out

    exchange = multigpu.HostExchange()
    results = {}
    for count in (frames, max(1, frames - 1)):          # a shorter last step reuses the same buffers
        wire, gather = pack_like_the_kernel(cfg, plan, render_tiles(count), count, multigpu)
        sends, recvs, assembled, first, step = multigpu.step_xfers(cfg, count)
        exchange.exchange_host(wire, sends, gather, recvs)
        for slot in range(assembled):
            f = first + slot * step
            parts = [gather[r * plan.gather_rank_stride_bytes + slot * plan.gather_frame_stride_bytes:][:plan.gather_frame_stride_bytes]
                     .view(np.float32) for r in range(world)]
            frame = multigpu.assemble_tiles(parts, W, H, TILE, TILE, got_shares, plan.channels)
            assert np.array_equal(frame, wanted[f]), (count, f)
            results[(count, f)] = True
    # every frame of both steps was assembled by exactly one rank
    mine = sorted(results)
    everyone = [None] * world
    dist.all_gather_object(everyone, mine)
    if rank == 0:
        seen = sorted(x for part in everyone for x in part)
        assert seen == sorted({(frames, f) for f in range(frames)} | {(max(1, frames - 1), f) for f in range(max(1, frames - 1))})
        np.save(out_path, np.asarray([len(seen)]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,frames,rgb_wire,shares,rotate", [
    (2, 1, False, None, False), (3, 1, False, (1, 1), False), (2, 3, True, None, False), (3, 2, True, (2, 3), False),
    (3, 1, True, (1, 4), False), (2, 2, True, None, True), (3, 3, True, None, True), (3, 5, False, None, True), (2, 1, True, None, True)])
def test_tiles_exchange_and_reassemble(world, frames, rgb_wire, shares, rotate, tmp_path, pkg, oracle_mod):
    """Even and uneven shares, gather to rank 0 and rotating roots: the assembled frames equal single full-frame
    oracle renders bit for bit (asserted in the workers)."""
    from shader_ray_amd import multigpu
    out = str(tmp_path / "frames.npy")
    mp.spawn(worker, args=(world, free_port(), out, frames, rgb_wire, shares, multigpu.ROTATE if rotate else multigpu.ROOT0),
             nprocs=world, join=True)
    assert int(np.load(out)[0]) >= frames


def test_tile_ownership_covers_the_frame_once(pkg):
    from shader_ray_amd import multigpu
    for (w, h, tw, th, world) in ((1920, 1080, 32, 32, 8), (3840, 2160, 32, 32, 8), (100, 70, 16, 32, 3), (16, 16, 16, 16, 4)):
        seen = []
        for r in range(world):
            tiles, tx, ty = multigpu.owned_tiles(w, h, tw, th, world, r)
            assert len(tiles) <= multigpu.max_tiles_per_rank(w, h, tw, th, world)
            seen += tiles
        assert sorted(seen) == list(range(tx * ty))
    # the 8-way split of a 1080p frame is balanced to within one tile
    counts = [len(multigpu.owned_tiles(1920, 1080, 32, 32, 8, r)[0]) for r in range(8)]
    assert max(counts) - min(counts) <= 1
    # uneven shares: every tile has exactly one owner, slots are dense, rank 0 owns c0 / c1 of a peer's share
    for world, shares in ((8, (2, 3)), (8, multigpu.balanced_shares(8)), (4, (3, 4)), (2, (7, 8)), (3, (1, 4))):
        seen, sizes = {}, []
        for r in range(world):
            period, phase, count = multigpu.rank_phases(world, r, shares)
            tiles, tx, ty = multigpu.owned_tiles(1920, 1080, 32, 32, period, phase, count)
            sizes.append(len(tiles))
            assert len(tiles) <= multigpu.max_tiles_per_rank(1920, 1080, 32, 32, world, shares)
            for k, t in enumerate(tiles):
                assert t not in seen and multigpu.tile_owner(t, world, shares) == (r, k)
                seen[t] = r
        assert sorted(seen) == list(range(tx * ty))
        assert abs(sizes[0] / sizes[1] - shares[0] / shares[1]) < 0.02 and max(sizes[1:]) - min(sizes[1:]) <= shares[1]
    assert multigpu.balanced_shares(1) == (1, 1) and multigpu.balanced_shares(8, overhead=0.0) == (1, 1)
    c0, c1 = multigpu.balanced_shares(8)
    assert (c0, c1) == (1, 2) and multigpu.balanced_shares(4) == (3, 4) and multigpu.balanced_shares(2) == (7, 8)


def test_the_c_plan_agrees_with_the_numpy_mapping(pkg):
    """shray_dist_make_plan / shray_dist_step_xfers against multigpu's numpy statement: tile sets, buffer shapes, and
    transfer lists that pair up rank against rank, in order, byte for byte."""
    from shader_ray_amd import multigpu
    N = pkg._native
    for (w, h, tile, world, shares, frames, mode, rgb) in (
            (1920, 1080, 32, 8, None, 8, multigpu.ROOT0, True), (1920, 1080, 32, 8, None, 8, multigpu.ROTATE, True),
            (3840, 2160, 32, 8, (1, 1), 3, multigpu.ROOT0, False), (100, 70, 16, 3, (2, 3), 7, multigpu.ROTATE, True),
            (100, 70, 16, 3, (2, 3), 7, multigpu.ROOT0, True), (16, 16, 16, 4, None, 9, multigpu.ROTATE, True),
            (333, 100, 16, 5, (1, 4), 64, multigpu.ROTATE, False), (64, 64, 32, 1, None, 4, multigpu.ROOT0, True)):
        cfgs = [multigpu.make_config(r, world, w, h, 1, frames, mode, multigpu.LOOPBACK, shares, tile, tile, rgb) for r in range(world)]
        plans = [multigpu.plan(c) for c in cfgs]
        eff = (plans[0].rank0_phases, plans[0].other_phases)
        if mode == multigpu.ROTATE:
            assert eff == (1, 1)
        elif shares is None:
            assert eff == multigpu.balanced_shares(world)
        else:
            assert eff == shares
        channels = 3 if rgb else 4
        most = multigpu.max_tiles_per_rank(w, h, tile, tile, world, eff)
        for r, p in enumerate(plans):
            period, phase, count = multigpu.rank_phases(world, r, eff)
            assert (p.tiles.tile_w, p.tiles.tile_h, p.tiles.tile_stride, p.tiles.tile_phase, p.tiles.tile_phase_count) == \
                (tile, tile, period, phase, count)
            assert p.owned_tiles == len(multigpu.owned_tiles(w, h, tile, tile, period, phase, count)[0])
            assert p.max_tiles == most and p.channels == channels
            assert p.render_frame_stride_bytes == most * tile * tile * 16
            assert p.wire_frame_stride_bytes == p.gather_frame_stride_bytes == most * tile * tile * channels * 4
            assert p.gather_rank_stride_bytes == p.gather_frame_stride_bytes * max(1, p.max_assembled)
        for count in sorted({1, frames, max(1, frames - 1), min(frames, world)}):
            lists = [multigpu.step_xfers(c, count) for c in cfgs]
            # every frame has exactly one owner, and the owner's slots are dense
            owners = [multigpu.frame_owner(cfgs[0], f) for f in range(count)]
            for r, (_s, _r, assembled, first, step) in enumerate(lists):
                assert [first + k * step for k in range(assembled)] == [f for f in range(count) if owners[f] == r]
                assert assembled <= plans[r].max_assembled
            # rank a's sends to b, in order, are b's receives from a, in order; sizes = the sender's owned bytes
            for a in range(world):
                for b in range(world):
                    out = [(f, n) for peer, f, _o, n in lists[a][0] if peer == b]
                    inn = [(f, n) for peer, f, _o, n in lists[b][1] if peer == a]
                    assert out == inn, (a, b, count)
                assert all(peer != a for peer, *_ in lists[a][0] + lists[a][1])
            for r, (sends, recvs, assembled, first, step) in enumerate(lists):
                body = plans[r].owned_tiles * tile * tile * channels * 4
                for peer, f, off, n in sends:
                    if mode == multigpu.ROTATE:
                        assert owners[f] == peer and off == f * plans[r].wire_frame_stride_bytes and n == body
                    else:
                        assert peer == 0 and f == -1 and off == 0 and n == (count - 1) * plans[r].wire_frame_stride_bytes + body
                    assert off + n <= plans[r].wire_frame_stride_bytes * frames
                for peer, f, off, n in recvs:
                    slot = 0 if f < 0 else (f - first) // step
                    assert off == peer * plans[r].gather_rank_stride_bytes + slot * plans[r].gather_frame_stride_bytes
                    assert off + n <= (peer + 1) * plans[r].gather_rank_stride_bytes
            # bytes per directed link: ROTATE spreads a step over all world * (world - 1) links
            if mode == multigpu.ROTATE and count == world and world > 1 and all(p.owned_tiles > 0 for p in plans):
                links = {(a, peer) for a in range(world) for peer, *_ in lists[a][0]}
                assert len(links) == world * (world - 1)
    # a malformed configuration is refused with a message, not a crash
    bad = multigpu.make_config(3, 3, 64, 64)
    with pytest.raises(N.ShrayError):
        multigpu.plan(bad)
    bad = multigpu.make_config(0, 2, 64, 64, tile_w=24)
    with pytest.raises(N.ShrayError):
        multigpu.plan(bad)


def test_assemble_numpy_handles_rgb_and_rgba(pkg):
    from shader_ray_amd import multigpu
    rng = np.random.default_rng(3)
    w, h, tw, th, world = 100, 70, 16, 32, 3
    per = multigpu.max_tiles_per_rank(w, h, tw, th, world)
    parts4 = [rng.random((per, th, tw, 4), dtype=np.float32) for _ in range(world)]
    for p in parts4:
        p[..., 3] = 1.0
    a = multigpu.assemble_tiles([p.reshape(-1) for p in parts4], w, h, tw, th)
    b = multigpu.assemble_tiles([np.ascontiguousarray(p[..., :3]).reshape(-1) for p in parts4], w, h, tw, th, channels=3)
    assert np.array_equal(a, b)


def _bench_env(**extra):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env.update(extra)
    return env


def test_a_stalled_rank_ends_in_the_timeout_record():
    """`python bench.py --gpus 2` whose ranks never come back must not hang its caller: within the
    --rank-timeout budget the run ends with ONE line {"error": "rank timeout", ...} on stdout and a non-zero exit code.  Both forms
    run without a GPU, because the ranks stall in front of their first GPU call (SHRAY_BENCH_STALL_AT=start):
      * the ranks' own watchdogs (a timer thread per rank; what also guards a run some other launcher started): rank 0 prints
        the record and the ranks leave with os._exit(124);
      * the parent's budget alone (the ranks' watchdogs off): it kills the child's process group and descendants, prints the
        record itself and returns 124."""
    import json
    import subprocess
    import time
    bench = os.path.join(ROOT, "bench.py")
    for knobs, want_rc in ((dict(), None), (dict(SHRAY_BENCH_NO_RANK_WATCHDOG="1", SHRAY_BENCH_PARENT_SLACK="2"), 124)):
        env = _bench_env(SHRAY_BENCH_STALL_RANK="all", SHRAY_BENCH_STALL_AT="start", **knobs)
        t0 = time.time()
        run = subprocess.run([sys.executable, bench, "--gpus", "2", "--steps", "2", "--warmup", "1", "--rank-timeout", "8"],
                             env=env, capture_output=True, text=True, timeout=240)
        took = time.time() - t0
        assert run.returncode != 0 and (want_rc is None or run.returncode == want_rc), (run.returncode, run.stderr[-2000:])
        lines = [ln for ln in run.stdout.splitlines() if ln.strip().startswith("{")]
        assert len(lines) == 1, run.stdout
        record = json.loads(lines[0])
        assert record["error"] == "rank timeout" and record["n_gpus"] == 2 and record["value"] is None and record["timeout_s"] == 8
        assert took < 120, took
        # nothing of the run is left behind
        time.sleep(0.5)
        import psutil
        left = [p for p in psutil.process_iter(["cmdline"]) if p.info["cmdline"] and bench in p.info["cmdline"] and "--rank-timeout" in p.info["cmdline"]]
        assert not left, [p.info["cmdline"] for p in left]
