"""Frame tiling across the GPUs of one node.

Pixels are independent (reference raytracer.es.fs:613-682 reads no neighbour), so a frame
shards by pixel: the scene and environment are replicated on every GPU, the frame is cut
into tile_w x tile_h tiles numbered row-major, rank r renders the tiles with
index % world_size == r (interleaved, because the object sits mid-frame and contiguous
bands would be unbalanced) and packs them densely; one gather of the packed tile buffers to
rank 0 (RCCL over xGMI: every peer sends on its own link) is the only exchange step; rank 0
de-interleaves.  Samples of a pixel never leave their GPU, so there is no reduction.

Shares: rank 0 also packs nothing less than its peers but additionally receives every buffer and
de-interleaves the frames, so it owns fewer tiles: the tiles are dealt in periods of
c0 + (world - 1) * c1 phases, of which rank 0 takes the first c0 and every other rank c1 (c0 <= c1;
`balanced_shares`).  shray_tile_set / shray_assemble_tiles_split_device speak the same scheme.

Frames per step: a rank's share of ONE 1080p frame is latency-bound -- the frame's few
long-running waves take ~0.5 ms wherever they land, whatever the share (profiles/r01) -- so a
step carries `frames` consecutive frames: one launch (shray_render_batch_device), one larger
gather, one de-interleave.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _native as N

DEFAULT_TILE = 32


# rank 0's extra work per frame (receive + de-interleave + its larger pack) as a fraction of ONE GPU's time for a
# whole frame: 0.017 ms of 0.28 ms on MI355X; only the shares depend on it, never the image
RANK0_OVERHEAD = 0.06


def balanced_shares(world: int, overhead: float = RANK0_OVERHEAD, largest: int = 8):
    """(c0, c1): phases per period for rank 0 and for every other rank that minimise the slowest rank's
    time, c0 / period + overhead against c1 / period, period = c0 + (world - 1) * c1."""
    if world <= 1:
        return 1, 1
    best = None
    for c1 in range(1, largest + 1):
        for c0 in range(1, c1 + 1):
            period = c0 + (world - 1) * c1
            cost = max(c0 / period + overhead, c1 / period)
            if best is None or cost < best[0] - 1e-12:
                best = (cost, c0, c1)
    return best[1], best[2]


def rank_phases(world: int, rank: int, shares=(1, 1)):
    """(period, first phase, phase count) of `rank` under shares (c0, c1)."""
    c0, c1 = shares
    period = c0 + (world - 1) * c1
    return (period, 0, c0) if rank == 0 else (period, c0 + (rank - 1) * c1, c1)


def owned_tiles(width: int, height: int, tile_w: int, tile_h: int, stride: int, phase: int, count: int = 1):
    """Row-major tile indices of the set {t : phase <= t % stride < phase + count}, in the order the
    kernel packs them, plus the grid size (tiles_x, tiles_y)."""
    tiles_x = (width + tile_w - 1) // tile_w
    tiles_y = (height + tile_h - 1) // tile_h
    count = max(1, count)
    return [t for t in range(tiles_x * tiles_y) if phase <= t % stride < phase + count], tiles_x, tiles_y


def tile_owner(t: int, world: int, shares=(1, 1)):
    """(rank, slot in that rank's packed buffer) of tile t."""
    c0, c1 = shares
    period = c0 + (world - 1) * c1
    phase, rnd = t % period, t // period
    if phase < c0:
        return 0, rnd * c0 + phase
    return 1 + (phase - c0) // c1, rnd * c1 + (phase - c0) % c1


def assemble_tiles(parts, width: int, height: int, tile_w: int, tile_h: int, shares=(1, 1)) -> np.ndarray:
    """De-interleaves the packed tile buffers of all ranks (list index = rank) into one
    [height, width, 4] frame.  Works on numpy arrays; see assemble_tiles_torch for the
    device-side form of the even split."""
    world = len(parts)
    frame = np.zeros((height, width, 4), dtype=np.float32)
    tiles_x = (width + tile_w - 1) // tile_w
    tiles_y = (height + tile_h - 1) // tile_h
    packed = [np.asarray(flat, dtype=np.float32).reshape(-1, tile_h, tile_w, 4) for flat in parts]
    for t in range(tiles_x * tiles_y):
        rank, k = tile_owner(t, world, shares)
        x0, y0 = (t % tiles_x) * tile_w, (t // tiles_x) * tile_h
        w, h = min(tile_w, width - x0), min(tile_h, height - y0)
        frame[y0:y0 + h, x0:x0 + w] = packed[rank][k, :h, :w]
    return frame


def assemble_tiles_torch(gathered, width: int, height: int, tile_w: int, tile_h: int):
    """`gathered` is a [world, max_tiles, tile_h, tile_w, 4] tensor (rank-major, as gathered, EVEN split);
    returns the [height, width, 4] frame on the same device using one permute + crop."""
    import torch
    world, max_tiles = gathered.shape[0], gathered.shape[1]
    tiles_x = (width + tile_w - 1) // tile_w
    tiles_y = (height + tile_h - 1) // tile_h
    # tile t lives at gathered[t % world, t // world]
    by_tile = gathered.permute(1, 0, 2, 3, 4).reshape(max_tiles * world, tile_h, tile_w, 4)[: tiles_x * tiles_y]
    grid = by_tile.reshape(tiles_y, tiles_x, tile_h, tile_w, 4).permute(0, 2, 1, 3, 4)
    return grid.reshape(tiles_y * tile_h, tiles_x * tile_w, 4)[:height, :width].contiguous()


def max_tiles_per_rank(width: int, height: int, tile_w: int, tile_h: int, world: int, shares=(1, 1)) -> int:
    """The largest packed buffer among the ranks, in tiles (a gather moves equal-sized buffers)."""
    tiles_x = (width + tile_w - 1) // tile_w
    tiles_y = (height + tile_h - 1) // tile_h
    total = tiles_x * tiles_y
    most = 0
    for rank in range(world):
        period, phase, count = rank_phases(world, rank, shares)
        most = max(most, total // period * count + min(count, max(0, total % period - phase)))
    return most


class DistributedFrame:
    """`frames` consecutive frames split across the ranks of a torch.distributed group, reusable.

    All buffers are allocated once: this rank's packed tile buffer ([frames, tiles, tile_h,
    tile_w, 4], what the kernel writes), its wire buffer and, on rank 0, one [world, frames, ...]
    receive buffer whose rows are the gather targets and the assembled [frames, height, width, 4]
    output.  A step is: render (one launch for all `frames`) -> pack -> ONE gather -> one
    permute/crop copy.  Nothing synchronises with the host.

    Wire format: alpha is the constant 1 for every pixel of a frame (raytracer.es.fs:676), so
    with `rgb_wire` only R, G, B travel (12 instead of 16 bytes per pixel over xGMI) and rank 0
    writes them into an output whose alpha plane is already 1."""

    def __init__(self, width: int, height: int, tile_w: int = DEFAULT_TILE, tile_h: int = DEFAULT_TILE, group=None,
                 device=None, always_gather: bool = False, stage_through_host: bool = False, frames: int = 1,
                 rgb_wire: bool = True, shares=None):
        import torch
        import torch.distributed as dist
        self.width, self.height, self.tile_w, self.tile_h = width, height, tile_w, tile_h
        self.group = group
        self.frames = frames
        self.channels = 3 if rgb_wire else 4
        self.always_gather = always_gather   # run the collective even with one rank (rehearsal)
        self.stage_through_host = stage_through_host   # gloo rehearsal: the collective moves host copies
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        # rank 0 owns fewer tiles than its peers (it also receives and de-interleaves): (c0, c1) phases per period
        self.shares = tuple(shares) if shares is not None else balanced_shares(self.world)
        self.per_rank = max_tiles_per_rank(width, height, tile_w, tile_h, self.world, self.shares)
        self.pixels = self.per_rank * tile_h * tile_w          # per frame, padded to whole tiles
        self.mine = torch.zeros(frames, self.pixels * 4, dtype=torch.float32, device=device)
        self.wire = torch.zeros(frames, self.pixels * 3, dtype=torch.float32, device=device) if rgb_wire else self.mine
        period, phase, count = rank_phases(self.world, self.rank, self.shares)
        self.tiles = N.TileSet(tile_w, tile_h, period, phase, count)
        self.received = None
        self.output = None
        if self.rank == 0:
            self.received = torch.zeros(self.world, frames, self.pixels * self.channels, dtype=torch.float32, device=device)
            self.output = torch.ones(frames, height, width, 4, dtype=torch.float32, device=device)

    @property
    def frame_stride_bytes(self) -> int:
        return self.pixels * 16

    def render(self, render_tiles, count: int | None = None):
        """`render_tiles(tile_set, out_tensor)` fills this rank's packed tiles of `count` (default:
        all `frames`) frames, frame k at out_tensor[k] (on a GPU box that is
        Scene.render_batch_into; the CPU rehearsal passes an oracle-backed stand-in).  Returns the
        assembled frames on rank 0 -- [height, width, 4] when the object holds one frame, else
        [count, height, width, 4] -- and None elsewhere."""
        import torch.distributed as dist
        count = self.frames if count is None else count
        assert 1 <= count <= self.frames
        mine = self.mine[:count]
        render_tiles(self.tiles, mine)
        wire = mine
        if self.channels == 3:
            wire = self.wire[:count]
            wire.view(count, self.pixels, 3).copy_(mine.view(count, self.pixels, 4)[:, :, :3])
        if self.world == 1 and not self.always_gather:
            return self._assemble(wire.unsqueeze(0), count)
        sink = [self.received[r, :count] for r in range(self.world)] if self.rank == 0 else None
        if self.stage_through_host:
            wire_host = wire.cpu()
            sink_host = [t.cpu() for t in sink] if self.rank == 0 else None
            dist.gather(wire_host, sink_host, dst=0, group=self.group)
            if self.rank == 0:
                for dst, src in zip(sink, sink_host):
                    dst.copy_(src)
        else:
            dist.gather(wire, sink, dst=0, group=self.group)
        if self.rank != 0:
            return None
        return self._assemble(self.received[:, :count], count)

    def _assemble(self, gathered, count: int):
        """gathered: [world, count, pixels * channels], rank-major as gathered."""
        import torch
        c = self.channels
        if self.output is None:   # world == 1 without a process group
            self.output = torch.ones(self.frames, self.height, self.width, 4, dtype=torch.float32, device=gathered.device)
        if gathered.is_cuda:
            # the library's de-interleave kernel (shray_assemble_tiles_device), on the current stream
            assert gathered.stride(2) == 1 and gathered.stride(1) == self.pixels * c
            N.check(N.load_hip().shray_assemble_tiles_split_device(
                C.c_void_p(gathered.data_ptr()), self.world, self.shares[0], self.shares[1], count, c,
                gathered.stride(0) * 4, gathered.stride(1) * 4,
                self.width, self.height, self.tile_w, self.tile_h, C.c_void_p(self.output.data_ptr()),
                C.c_void_p(torch.cuda.current_stream(gathered.device).cuda_stream)))
            return self.output[0] if self.frames == 1 else self.output[:count]
        # host tensors (the gloo rehearsal): the same mapping through an index table (tile -> rank, slot)
        tiles_x = (self.width + self.tile_w - 1) // self.tile_w
        tiles_y = (self.height + self.tile_h - 1) // self.tile_h
        owners = [tile_owner(t, self.world, self.shares) for t in range(tiles_x * tiles_y)]
        rank_of = torch.tensor([o[0] for o in owners], dtype=torch.long)
        slot_of = torch.tensor([o[1] for o in owners], dtype=torch.long)
        per_tile = gathered.reshape(self.world, count, self.per_rank, self.tile_h, self.tile_w, c)
        by_tile = per_tile[rank_of, :, slot_of].permute(1, 0, 2, 3, 4)          # [count, tiles, tile_h, tile_w, c]
        grid = by_tile.reshape(count, tiles_y, tiles_x, self.tile_h, self.tile_w, c).permute(0, 1, 3, 2, 4, 5)
        image = grid.reshape(count, tiles_y * self.tile_h, tiles_x * self.tile_w, c)[:, : self.height, : self.width]
        self.output[:count, :, :, :c].copy_(image)
        return self.output[0] if self.frames == 1 else self.output[:count]


def render_frame_distributed(render_tiles, width: int, height: int, tile_w: int = DEFAULT_TILE,
                             tile_h: int = DEFAULT_TILE, group=None, device=None, shares=None):
    """One-shot convenience wrapper around DistributedFrame."""
    return DistributedFrame(width, height, tile_w, tile_h, group, device, rgb_wire=False, shares=shares).render(render_tiles)
