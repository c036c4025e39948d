"""Frame tiling across the GPUs of one node: a thin Python face of libshray_dist.so
(include/shader_ray_dist.h), where one rank's whole step lives in C++:

    this rank's tiles of `count` frames (one launch) -> pack (RGB on the wire) -> grouped
    ncclSend / ncclRecv over xGMI -> de-interleave on the rank that assembles the frame

Pixels are independent (reference raytracer.es.fs:613-682 reads no neighbour), so a frame shards by
pixel: scene and environment are replicated on every GPU, tiles are dealt round robin (the object
sits mid-frame, contiguous bands would be unbalanced) and all samples of a pixel stay on its GPU --
no reduction, one exchange step.  Two root modes: ROOT0 gathers every frame on rank 0 (which may own
a smaller share of the tiles); ROTATE assembles frame f of a step on rank f % world, so that all
world * (world - 1) directed links carry pixels instead of the 7 into rank 0.

What is left here: the ctypes marshalling (`Rank`), the plan queries (`plan`, `step_xfers`: host-only,
usable without a GPU), a numpy statement of the tile mapping for tests (`owned_tiles`, `tile_owner`,
`assemble_tiles`), and `HostExchange`: the CALLBACK transport over torch.distributed / gloo, which is
how `bench.py`'s multi-rank control flow is rehearsed on a one-GPU box.  Nothing here renders.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _native as N

DEFAULT_TILE = 32
ROOT0, ROTATE = N.DIST_ROOT0, N.DIST_ROTATE
RCCL, LOOPBACK, CALLBACK = N.DIST_RCCL, N.DIST_LOOPBACK, N.DIST_CALLBACK


# ---- the plan (host only) ----------------------------------------------------------------------------
def balanced_shares(world: int, overhead: float = -1.0):
    """(c0, c1): tile phases per period for rank 0 and for every other rank (shray_dist_balanced_shares)."""
    c0, c1 = C.c_int(), C.c_int()
    N.check_dist(N.load_dist().shray_dist_balanced_shares(world, overhead, C.byref(c0), C.byref(c1)))
    return c0.value, c1.value


def make_config(rank: int, world: int, width: int, height: int, spp: int = 1, frames: int = 1, root_mode: int = ROOT0,
                transport: int = RCCL, shares=None, tile_w: int = DEFAULT_TILE, tile_h: int = DEFAULT_TILE,
                rgb_wire: bool = True, buffer_sets: int = 2) -> N.DistConfig:
    cfg = N.DistConfig()
    cfg.struct_size = C.sizeof(N.DistConfig)
    cfg.rank, cfg.world = rank, world
    cfg.width, cfg.height, cfg.spp = width, height, spp
    cfg.tile_w, cfg.tile_h = tile_w, tile_h
    cfg.max_frames = frames
    cfg.root_mode = root_mode
    cfg.rank0_phases, cfg.other_phases = shares if shares is not None else (0, 0)
    cfg.rgb_wire = 1 if rgb_wire else 0
    cfg.transport = transport
    cfg.buffer_sets = buffer_sets
    return cfg


def plan(cfg: N.DistConfig) -> N.DistPlan:
    p = N.DistPlan()
    N.check_dist(N.load_dist().shray_dist_make_plan(C.byref(cfg), C.byref(p)))
    return p


def frame_owner(cfg: N.DistConfig, frame: int) -> int:
    rank = N.load_dist().shray_dist_frame_owner(C.byref(cfg), frame)
    if rank < 0:
        N.check_dist(rank)
    return rank


def step_xfers(cfg: N.DistConfig, count: int):
    """(sends, recvs, assembled, first_frame, frame_step) of a step of `count` frames for cfg.rank; sends / recvs
    are lists of (peer, frame, offset_bytes, bytes)."""
    cap = N.MAX_BATCH + N.DIST_MAX_WORLD
    sends, recvs = (N.DistXfer * cap)(), (N.DistXfer * cap)()
    ns, nr, asm, first, step = C.c_int(), C.c_int(), C.c_int(), C.c_int(), C.c_int()
    N.check_dist(N.load_dist().shray_dist_step_xfers(C.byref(cfg), count, sends, C.byref(ns), recvs, C.byref(nr), C.byref(asm),
                                                     C.byref(first), C.byref(step)))
    as_list = lambda arr, n: [(x.peer, x.frame, x.offset_bytes, x.bytes) for x in arr[:n]]   # noqa: E731
    return as_list(sends, ns.value), as_list(recvs, nr.value), asm.value, first.value, step.value


# ---- the tile mapping in numpy: an independent statement for tests -------------------------------------
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


def max_tiles_per_rank(width: int, height: int, tile_w: int, tile_h: int, world: int, shares=(1, 1)) -> int:
    tiles_x = (width + tile_w - 1) // tile_w
    tiles_y = (height + tile_h - 1) // tile_h
    total = tiles_x * tiles_y
    most = 0
    for rank in range(world):
        period, phase, count = rank_phases(world, rank, shares)
        most = max(most, total // period * count + min(count, max(0, total % period - phase)))
    return most


def assemble_tiles(parts, width: int, height: int, tile_w: int, tile_h: int, shares=(1, 1), channels: int = 4) -> np.ndarray:
    """De-interleaves the packed tile buffers of all ranks (list index = rank, `channels` floats per pixel) into one
    [height, width, 4] frame (alpha = 1 when only R, G, B travelled)."""
    world = len(parts)
    frame = np.ones((height, width, 4), dtype=np.float32)
    tiles_x = (width + tile_w - 1) // tile_w
    tiles_y = (height + tile_h - 1) // tile_h
    per_tile = tile_h * tile_w * channels
    for t in range(tiles_x * tiles_y):
        rank, k = tile_owner(t, world, shares)
        tile = np.asarray(parts[rank], dtype=np.float32).reshape(-1)[k * per_tile:(k + 1) * per_tile].reshape(tile_h, tile_w, channels)
        x0, y0 = (t % tiles_x) * tile_w, (t // tiles_x) * tile_h
        w, h = min(tile_w, width - x0), min(tile_h, height - y0)
        frame[y0:y0 + h, x0:x0 + w, :channels] = tile[:h, :w]
    return frame


# ---- one rank of the frame loop ------------------------------------------------------------------------
class Hub:
    """The LOOPBACK transport's in-process meeting point: ranks are threads of this process sharing one GPU."""

    def __init__(self, world: int):
        self._lib = N.load_dist()
        h = C.c_void_p()
        N.check_dist(self._lib.shray_dist_hub_create(world, C.byref(h)))
        self.handle = h

    def close(self):
        if getattr(self, "handle", None):
            self._lib.shray_dist_hub_destroy(self.handle)
            self.handle = None


def unique_id() -> bytes:
    """An RCCL unique id (rank 0 makes one and hands it to the others over any side channel)."""
    buf = C.create_string_buffer(N.DIST_UNIQUE_ID_BYTES)
    N.check_dist(N.load_dist().shray_dist_unique_id(buf))
    return buf.raw


class Rank:
    """shray_dist: this rank's step of the multi-GPU frame loop.  `scene` is the rank's replica
    (tracer.Scene); `transport_arg`: RCCL -> the unique id bytes, LOOPBACK -> a Hub, CALLBACK -> an object with
    an `exchange_device(d_wire, sends, d_gather, recvs, stream_ptr)` method (HostExchange)."""

    def __init__(self, scene, cfg: N.DistConfig, transport_arg):
        self._lib = N.load_dist()
        self.cfg = cfg
        self.scene = scene                    # keeps the replica alive
        self.plan = plan(cfg)
        self._keep = transport_arg
        if cfg.transport == RCCL:
            arg = C.create_string_buffer(bytes(transport_arg), N.DIST_UNIQUE_ID_BYTES)
            arg_p = C.cast(arg, C.c_void_p)
        elif cfg.transport == LOOPBACK:
            arg_p = transport_arg.handle
        else:
            def trampoline(_user, d_wire, sends, ns, d_gather, recvs, nr, stream):
                try:
                    transport_arg.exchange_device(d_wire, [(x.peer, x.frame, x.offset_bytes, x.bytes) for x in sends[:ns]],
                                                  d_gather, [(x.peer, x.frame, x.offset_bytes, x.bytes) for x in recvs[:nr]],
                                                  stream)
                    return 0
                except Exception as exc:   # noqa: BLE001  (an exception must not unwind through the C frames)
                    print(f"shray exchange callback: {exc!r}", flush=True)
                    return 1
            self._fn = N.DIST_EXCHANGE_FN(trampoline)
            self._cb = N.DistCallbacks(None, self._fn)
            arg_p = C.cast(C.pointer(self._cb), C.c_void_p)
        handle = C.c_void_p()
        N.check_dist(self._lib.shray_dist_create(scene._handle, C.byref(cfg), arg_p, C.byref(handle)))
        self._handle = handle

    def step(self, params_list, buffer_set: int = 0, stream_ptr: int = 0):
        count = len(params_list)
        array = (N.FrameParams * count)(*params_list)
        N.check_dist(self._lib.shray_dist_step(self._handle, buffer_set, array, count, C.c_void_p(stream_ptr)))

    def set_timing(self, enable: bool = True):
        """Later steps record stage stamps (shray_dist_set_timing)."""
        N.check_dist(self._lib.shray_dist_set_timing(self._handle, 1 if enable else 0))

    def step_times(self, buffer_set: int):
        """(render_ms, exchange_ms, assemble_ms) of the set's most recent step; waits for it (shray_dist_step_times)."""
        r, x, a = C.c_float(), C.c_float(), C.c_float()
        N.check_dist(self._lib.shray_dist_step_times(self._handle, buffer_set, C.byref(r), C.byref(x), C.byref(a)))
        return r.value, x.value, a.value

    def link_bytes(self, count: int):
        """[bytes this rank sends to peer p in a step of `count` frames for p in range(world)] (the plan's transfers: host only)."""
        sends, _recvs, *_ = step_xfers(self.cfg, count)
        out = [0] * self.cfg.world
        for peer, _frame, _offset, nbytes in sends:
            out[peer] += nbytes
        return out

    def output(self, buffer_set: int, count: int):
        """(assembled, first_frame, frame_step, device pointer) after a step of `count` frames."""
        asm, first, step, ptr = C.c_int(), C.c_int(), C.c_int(), C.c_void_p()
        N.check_dist(self._lib.shray_dist_output(self._handle, buffer_set, count, C.byref(asm), C.byref(first), C.byref(step),
                                                 C.byref(ptr)))
        return asm.value, first.value, step.value, ptr.value

    def frames(self, buffer_set: int, count: int, stream_ptr: int = 0):
        """{frame index in the step: [height, width, 4] torch tensor on the device} of the frames this rank
        assembled (copies, enqueued on the stream)."""
        import torch
        asm, first, step, _ = self.output(buffer_set, count)
        if asm == 0:
            return {}
        dev = torch.device("cuda", self.device_index())
        out = torch.empty(asm, self.cfg.height, self.cfg.width, 4, dtype=torch.float32, device=dev)
        N.check_dist(self._lib.shray_dist_copy_output(self._handle, buffer_set, count, C.c_void_p(out.data_ptr()), C.c_void_p(stream_ptr)))
        return {first + k * step: out[k] for k in range(asm)}

    def world(self):
        """(world of the configuration, ranks the transport's communicator reports: ncclCommCount for RCCL, else 0)."""
        w, ranks = C.c_int(), C.c_int()
        N.check_dist(self._lib.shray_dist_world(self._handle, C.byref(w), C.byref(ranks)))
        return w.value, ranks.value

    def device_index(self) -> int:
        d = C.c_int()
        N.check(N.load_hip().shray_scene_device(self.scene._handle, C.byref(d)))
        return d.value

    def close(self):
        if getattr(self, "_handle", None):
            self._lib.shray_dist_destroy(self._handle)
            self._handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


# ---- the CALLBACK transport over torch.distributed (gloo): rehearsal only ------------------------------
class HostExchange:
    """Moves a step's transfers between ranks through host memory with torch.distributed point-to-point calls
    (gloo).  `exchange_host` works on host byte arrays (what the CPU tests drive); `exchange_device` wraps it for
    the C library's CALLBACK transport: device -> host, exchange, host -> device."""

    def __init__(self, group=None):
        self.group = group

    def exchange_host(self, wire: np.ndarray, sends, gather: np.ndarray, recvs):
        """wire / gather: uint8 views of this rank's wire and gather buffers."""
        import torch
        import torch.distributed as dist
        ops, landing = [], []
        for peer, _frame, offset, nbytes in recvs:
            t = torch.empty(nbytes, dtype=torch.uint8)
            landing.append((t, offset, nbytes))
            ops.append(dist.P2POp(dist.irecv, t, peer, self.group))
        for peer, _frame, offset, nbytes in sends:
            ops.append(dist.P2POp(dist.isend, torch.from_numpy(np.ascontiguousarray(wire[offset:offset + nbytes])), peer, self.group))
        if ops:
            for req in dist.batch_isend_irecv(ops):
                req.wait()
        for t, offset, nbytes in landing:
            gather[offset:offset + nbytes] = t.numpy()

    def exchange_device(self, d_wire, sends, d_gather, recvs, stream_ptr):
        lib = N.load_dist()
        wire_end = max((o + b for _, _, o, b in sends), default=0)
        wire = np.empty(wire_end, dtype=np.uint8)
        if wire_end:
            N.check_dist(lib.shray_dist_copy_to_host(wire.ctypes.data_as(C.c_void_p), C.c_void_p(d_wire), wire_end, C.c_void_p(stream_ptr)))
        hi = max((o + b for _, _, o, b in recvs), default=0)
        gather = np.empty(hi, dtype=np.uint8)
        self.exchange_host(wire, sends, gather, recvs)
        for _peer, _frame, offset, nbytes in recvs:
            N.check_dist(lib.shray_dist_copy_to_device(C.c_void_p(d_gather + offset), gather[offset:].ctypes.data_as(C.c_void_p), nbytes,
                                                       C.c_void_p(stream_ptr)))
