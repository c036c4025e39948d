"""World-size-2 (and 3) rehearsal of the multi-GPU frame split on CPU with the gloo backend.

The real path renders each rank's interleaved tiles with the HIP kernel and gathers the
packed tile buffers to rank 0 over RCCL (shader-ray_amd/multigpu.py).  Here the same
gather + de-interleave code runs under gloo, with the CPU oracle standing in for the
renderer of a rank's tiles (this is a test: the product path itself has no CPU fallback),
and the assembled frame must equal a single full-frame oracle render bit for bit."""
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


def worker(rank, world, port, out_path, frames, rgb_wire, shares=None):
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

    def render_tiles(tile_set, out):
        # stand-in for Scene.render_batch_into: fill this rank's packed tiles of out.shape[0] frames from the oracle
        tiles, tiles_x, _ = multigpu.owned_tiles(W, H, tile_set.tile_w, tile_set.tile_h, tile_set.tile_stride, tile_set.tile_phase,
                                                 tile_set.tile_phase_count)
        for f in range(out.shape[0]):
            packed = out[f].view(-1, tile_set.tile_h, tile_set.tile_w, 4)
            packed.zero_()                                       # the kernel writes (0, 0, 0, 0) outside the frame
            for k, t in enumerate(tiles):
                x0, y0 = (t % tiles_x) * tile_set.tile_w, (t // tiles_x) * tile_set.tile_h
                w, h = min(tile_set.tile_w, W - x0), min(tile_set.tile_h, H - y0)
                packed[k, :h, :w] = torch.from_numpy(wanted[f][y0:y0 + h, x0:x0 + w])

    if frames == 1 and not rgb_wire:
        got = [multigpu.render_frame_distributed(render_tiles, W, H, TILE, TILE, device="cpu", shares=shares)]
        short = None
    else:
        split = multigpu.DistributedFrame(W, H, TILE, TILE, device="cpu", frames=frames, rgb_wire=rgb_wire, shares=shares)
        if shares is None and world > 1:
            assert split.shares[0] < split.shares[1]     # rank 0 owns the smaller share by default
        out = split.render(render_tiles)
        got = None if out is None else ([out.clone()] if frames == 1 else list(out.clone()))
        # a shorter last step reuses the same buffers
        short = split.render(render_tiles, count=max(1, frames - 1))
        got = [None] if got is None else got
    if rank == 0:
        assert all(g is not None for g in got)
        if short is not None:
            short = short if short.dim() == 4 else short.unsqueeze(0)
            assert short.shape[0] == max(1, frames - 1)
            for f in range(short.shape[0]):
                assert np.array_equal(short[f].numpy(), wanted[f])
        np.save(out_path, np.stack([np.stack([g.numpy() for g in got]), np.stack(wanted)]))
    else:
        assert got == [None]
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,frames,rgb_wire,shares", [(2, 1, False, None), (3, 1, False, (1, 1)), (2, 3, True, None),
                                                           (3, 2, True, (2, 3)), (2, 1, True, (1, 1)), (3, 1, True, (1, 4))])
def test_tiles_gather_and_reassemble(world, frames, rgb_wire, shares, tmp_path, pkg, oracle_mod):
    """Even and uneven shares (rank 0 owns c0 of every c0 + (world - 1) c1 tile phases): the assembled frames
    equal single full-frame oracle renders bit for bit."""
    out = str(tmp_path / "frames.npy")
    mp.spawn(worker, args=(world, free_port(), out, frames, rgb_wire, shares), nprocs=world, join=True)
    got, want = np.load(out)
    assert got.shape == (frames, H, W, 4)
    assert np.array_equal(got, want)


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
    assert c0 < c1 and max(c0 / (c0 + 7 * c1) + multigpu.RANK0_OVERHEAD, c1 / (c0 + 7 * c1)) < 1 / 8 + multigpu.RANK0_OVERHEAD


def test_assemble_torch_matches_numpy(pkg):
    from shader_ray_amd import multigpu
    rng = np.random.default_rng(3)
    w, h, tw, th, world = 100, 70, 16, 32, 3
    per = multigpu.max_tiles_per_rank(w, h, tw, th, world)
    parts = [rng.random((per, th, tw, 4), dtype=np.float32) for _ in range(world)]
    a = multigpu.assemble_tiles([p.reshape(-1) for p in parts], w, h, tw, th)
    b = multigpu.assemble_tiles_torch(torch.from_numpy(np.stack(parts)), w, h, tw, th).numpy()
    assert np.array_equal(a, b)
