"""Frame tiling across the GPUs of one node.

Pixels are independent (reference raytracer.es.fs:613-682 reads no neighbour), so a frame
shards by pixel: the scene and environment are replicated on every GPU, the frame is cut
into tile_w x tile_h tiles numbered row-major, rank r renders the tiles with
index % world_size == r (interleaved, because the object sits mid-frame and contiguous
bands would be unbalanced) and packs them densely; one gather of the packed tile buffers to
rank 0 (RCCL over xGMI: every peer sends on its own link) is the only exchange step; rank 0
de-interleaves.  Samples of a pixel never leave their GPU, so there is no reduction.
"""
from __future__ import annotations

import numpy as np

from . import _native as N

DEFAULT_TILE = 32


def owned_tiles(width: int, height: int, tile_w: int, tile_h: int, stride: int, phase: int):
    """Row-major tile indices owned by `phase`, plus the grid size (tiles_x, tiles_y)."""
    tiles_x = (width + tile_w - 1) // tile_w
    tiles_y = (height + tile_h - 1) // tile_h
    return list(range(phase, tiles_x * tiles_y, stride)), tiles_x, tiles_y


def assemble_tiles(parts, width: int, height: int, tile_w: int, tile_h: int) -> np.ndarray:
    """De-interleaves the packed tile buffers of all ranks (list index = rank = phase) into
    one [height, width, 4] frame.  Works on numpy arrays; see assemble_tiles_torch for the
    device-side form rank 0 uses."""
    stride = len(parts)
    frame = np.zeros((height, width, 4), dtype=np.float32)
    for phase, flat in enumerate(parts):
        tiles, tiles_x, _ = owned_tiles(width, height, tile_w, tile_h, stride, phase)
        packed = np.asarray(flat, dtype=np.float32).reshape(-1, tile_h, tile_w, 4)
        assert len(packed) >= len(tiles)
        for k, t in enumerate(tiles):
            x0, y0 = (t % tiles_x) * tile_w, (t // tiles_x) * tile_h
            w, h = min(tile_w, width - x0), min(tile_h, height - y0)
            frame[y0:y0 + h, x0:x0 + w] = packed[k, :h, :w]
    return frame


def assemble_tiles_torch(gathered, width: int, height: int, tile_w: int, tile_h: int):
    """`gathered` is a [world, max_tiles, tile_h, tile_w, 4] tensor (rank-major, as gathered);
    returns the [height, width, 4] frame on the same device using one permute + crop."""
    import torch
    world, max_tiles = gathered.shape[0], gathered.shape[1]
    tiles_x = (width + tile_w - 1) // tile_w
    tiles_y = (height + tile_h - 1) // tile_h
    # tile t lives at gathered[t % world, t // world]
    by_tile = gathered.permute(1, 0, 2, 3, 4).reshape(max_tiles * world, tile_h, tile_w, 4)[: tiles_x * tiles_y]
    grid = by_tile.reshape(tiles_y, tiles_x, tile_h, tile_w, 4).permute(0, 2, 1, 3, 4)
    return grid.reshape(tiles_y * tile_h, tiles_x * tile_w, 4)[:height, :width].contiguous()


def max_tiles_per_rank(width: int, height: int, tile_w: int, tile_h: int, world: int) -> int:
    tiles_x = (width + tile_w - 1) // tile_w
    tiles_y = (height + tile_h - 1) // tile_h
    return (tiles_x * tiles_y + world - 1) // world


class DistributedFrame:
    """One frame split across the ranks of a torch.distributed group, reusable across frames.

    All buffers are allocated once: this rank's packed tile buffer and, on rank 0, one
    [world, max_tiles, tile_h, tile_w, 4] receive buffer whose rows are the gather targets, so a
    frame is: render -> gather -> one permute/crop kernel.  Nothing synchronises with the host."""

    def __init__(self, width: int, height: int, tile_w: int = DEFAULT_TILE, tile_h: int = DEFAULT_TILE, group=None,
                 device=None, always_gather: bool = False, stage_through_host: bool = False):
        import torch
        import torch.distributed as dist
        self.width, self.height, self.tile_w, self.tile_h = width, height, tile_w, tile_h
        self.group = group
        self.always_gather = always_gather   # run the collective even with one rank (rehearsal)
        self.stage_through_host = stage_through_host   # gloo rehearsal: the collective moves host copies
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.per_rank = max_tiles_per_rank(width, height, tile_w, tile_h, self.world)
        n = self.per_rank * tile_h * tile_w * 4
        self.mine = torch.zeros(n, dtype=torch.float32, device=device)
        self.tiles = N.TileSet(tile_w, tile_h, self.world, self.rank)
        self.received = None
        self.sink = None
        if self.rank == 0:
            self.received = torch.zeros(self.world, n, dtype=torch.float32, device=device)
            self.sink = [self.received[r] for r in range(self.world)]

    def render(self, render_tiles):
        """`render_tiles(tile_set, out_tensor)` fills this rank's packed tiles (on a GPU box that
        is Scene.render_into; the CPU rehearsal passes an oracle-backed stand-in).  Returns the
        assembled [height, width, 4] frame on rank 0, None elsewhere."""
        import torch.distributed as dist
        render_tiles(self.tiles, self.mine)
        if self.world == 1 and not self.always_gather:
            gathered = self.mine.view(1, self.per_rank, self.tile_h, self.tile_w, 4)
            return assemble_tiles_torch(gathered, self.width, self.height, self.tile_w, self.tile_h)
        if self.stage_through_host:
            mine_host = self.mine.cpu()
            sink_host = [t.cpu() for t in self.sink] if self.rank == 0 else None
            dist.gather(mine_host, sink_host, dst=0, group=self.group)
            if self.rank == 0:
                for dst, src in zip(self.sink, sink_host):
                    dst.copy_(src)
        else:
            dist.gather(self.mine, self.sink, dst=0, group=self.group)
        if self.rank != 0:
            return None
        gathered = self.received.view(self.world, self.per_rank, self.tile_h, self.tile_w, 4)
        return assemble_tiles_torch(gathered, self.width, self.height, self.tile_w, self.tile_h)


def render_frame_distributed(render_tiles, width: int, height: int, tile_w: int = DEFAULT_TILE,
                             tile_h: int = DEFAULT_TILE, group=None, device=None):
    """One-shot convenience wrapper around DistributedFrame."""
    return DistributedFrame(width, height, tile_w, tile_h, group, device).render(render_tiles)
