// plan.h -- who renders which tiles, who assembles which frame, and which bytes travel where: the host-only
// half of the multi-GPU frame loop (include/shader_ray_dist.h).  No GPU, no RCCL: plain integer arithmetic that
// every rank evaluates identically, so the transfer lists of two ranks always pair up.
//
// Tiles of a frame are numbered row-major and dealt in periods of c0 + (world - 1) c1 phases: rank 0 owns the
// first c0 phases of every period, rank r >= 1 the c1 phases from c0 + (r - 1) c1 (shray_tile_set,
// shray_assemble_tiles_split_device speak the same scheme).  A rank packs its tiles densely, in increasing tile
// index; every per-frame stride is sized for the largest share, so that all buffers have one shape.
#pragma once

#include <algorithm>
#include <cstdint>
#include <cstring>

#include "shader_ray_dist.h"

namespace shray_dist_plan_detail {

// rank 0's extra work per frame (receive + de-interleave) as a fraction of ONE GPU's time for a whole frame:
// 0.017 ms of 0.28 ms on MI355X; only the shares depend on it, never the image
constexpr double kRank0Overhead = 0.06;

inline void balanced_shares(int world, double overhead, int *c0_out, int *c1_out)
{
    *c0_out = *c1_out = 1;
    if (world <= 1)
        return;
    if (overhead < 0.0)
        overhead = kRank0Overhead;
    double best = 1e300;
    for (int c1 = 1; c1 <= 8; c1++)
        for (int c0 = 1; c0 <= c1; c0++) {
            const double period = c0 + (double)(world - 1) * c1;
            const double cost = std::max(c0 / period + overhead, c1 / period);
            if (cost < best - 1e-12) {
                best = cost;
                *c0_out = c0;
                *c1_out = c1;
            }
        }
}

// tiles t < total with phase <= t % period < phase + count
inline int64_t owned_tile_count(int64_t total, int64_t period, int64_t phase, int64_t count)
{
    return total / period * count + std::min<int64_t>(count, std::max<int64_t>(0, total % period - phase));
}

struct Resolved {
    int rank, world, width, height, tile_w, tile_h, max_frames, root_mode, c0, c1, channels, buffer_sets;
    int64_t tiles_total, period;
};

// validates the configuration and fills in its defaults; returns nullptr or what is wrong with it
inline const char *resolve(const shray_dist_config *cfg, Resolved *r)
{
    if (!cfg)
        return "config is NULL";
    if (cfg->struct_size != sizeof(shray_dist_config))
        return "shray_dist_config.struct_size does not match this library";
    if (cfg->world < 1 || cfg->world > SHRAY_DIST_MAX_WORLD || cfg->rank < 0 || cfg->rank >= cfg->world)
        return "rank / world out of range";
    if (cfg->width <= 0 || cfg->height <= 0 || cfg->width > 65536 || cfg->height > 65535 || cfg->spp <= 0)
        return "bad frame geometry";
    r->tile_w = cfg->tile_w ? cfg->tile_w : 32;
    r->tile_h = cfg->tile_h ? cfg->tile_h : 32;
    if (r->tile_w <= 0 || r->tile_h <= 0 || r->tile_w % 16 || r->tile_h % 16)
        return "tile sizes must be positive multiples of 16";
    if (cfg->max_frames < 1 || cfg->max_frames > SHRAY_MAX_BATCH)
        return "max_frames out of range (1..SHRAY_MAX_BATCH)";
    if (cfg->root_mode != SHRAY_DIST_ROOT0 && cfg->root_mode != SHRAY_DIST_ROTATE)
        return "unknown root_mode";
    if (cfg->transport != SHRAY_DIST_RCCL && cfg->transport != SHRAY_DIST_LOOPBACK && cfg->transport != SHRAY_DIST_CALLBACK)
        return "unknown transport";
    if (cfg->buffer_sets < 0 || cfg->buffer_sets > 4)
        return "buffer_sets out of range (0..4)";
    r->rank = cfg->rank;
    r->world = cfg->world;
    r->width = cfg->width;
    r->height = cfg->height;
    r->max_frames = cfg->max_frames;
    r->root_mode = cfg->root_mode;
    r->channels = cfg->rgb_wire ? 3 : 4;
    r->buffer_sets = cfg->buffer_sets ? cfg->buffer_sets : 2;
    if (cfg->root_mode == SHRAY_DIST_ROTATE) {
        r->c0 = r->c1 = 1;   // every rank renders, receives and de-interleaves alike
    } else if (cfg->rank0_phases == 0 && cfg->other_phases == 0) {
        balanced_shares(cfg->world, -1.0, &r->c0, &r->c1);
    } else {
        if (cfg->rank0_phases < 1 || cfg->other_phases < 1 || cfg->rank0_phases > 4096 || cfg->other_phases > 4096)
            return "shares out of range (1..4096 phases)";
        r->c0 = cfg->rank0_phases;
        r->c1 = cfg->other_phases;
    }
    r->period = r->c0 + (int64_t)(r->world - 1) * r->c1;
    r->tiles_total = (int64_t)((r->width + r->tile_w - 1) / r->tile_w) * ((r->height + r->tile_h - 1) / r->tile_h);
    return nullptr;
}

inline int first_phase(const Resolved &r, int rank) { return rank == 0 ? 0 : r.c0 + (rank - 1) * r.c1; }
inline int phase_count(const Resolved &r, int rank) { return rank == 0 ? r.c0 : r.c1; }
inline int64_t owned_tiles(const Resolved &r, int rank)
{
    return owned_tile_count(r.tiles_total, r.period, first_phase(r, rank), phase_count(r, rank));
}
inline int64_t max_tiles(const Resolved &r)
{
    int64_t most = 0;
    for (int k = 0; k < r.world; k++)
        most = std::max(most, owned_tiles(r, k));
    return most;
}
inline int frame_owner(const Resolved &r, int frame) { return r.root_mode == SHRAY_DIST_ROTATE ? frame % r.world : 0; }
// frames of a step of `count` that `rank` assembles
inline int assembled_frames(const Resolved &r, int rank, int count)
{
    if (r.root_mode == SHRAY_DIST_ROOT0)
        return rank == 0 ? count : 0;
    return count > rank ? (count - rank + r.world - 1) / r.world : 0;
}

inline void make_plan(const Resolved &r, shray_dist_plan *p)
{
    memset(p, 0, sizeof(*p));
    p->struct_size = (uint32_t)sizeof(*p);
    p->tiles.tile_w = r.tile_w;
    p->tiles.tile_h = r.tile_h;
    p->tiles.tile_stride = (int32_t)r.period;
    p->tiles.tile_phase = first_phase(r, r.rank);
    p->tiles.tile_phase_count = phase_count(r, r.rank);
    p->rank0_phases = r.c0;
    p->other_phases = r.c1;
    p->channels = r.channels;
    p->max_assembled = assembled_frames(r, r.rank, r.max_frames);
    p->owned_tiles = owned_tiles(r, r.rank);
    p->max_tiles = max_tiles(r);
    const int64_t pixels = p->max_tiles * r.tile_w * r.tile_h;
    p->render_frame_stride_bytes = pixels * 16;
    p->wire_frame_stride_bytes = pixels * r.channels * 4;
    p->gather_frame_stride_bytes = p->wire_frame_stride_bytes;
    p->gather_rank_stride_bytes = p->gather_frame_stride_bytes * std::max(1, p->max_assembled);
}

// The transfers of a step of `count` frames for rank r.rank.  ROOT0: every peer sends its `count` wire frames to
// rank 0 in one piece (the padding between frames is at most one tile: peers own equal shares).  ROTATE: frame f
// goes to rank f % world, one transfer per frame, in frame order -- so rank a's k-th send to b is b's k-th receive
// from a.  A rank's own tiles never travel: its pack writes them straight into its gather buffer.
inline void step_xfers(const Resolved &r, int count, shray_dist_xfer *sends, int *ns, shray_dist_xfer *recvs, int *nr)
{
    shray_dist_plan p;
    make_plan(r, &p);
    *ns = *nr = 0;
    auto frame_bytes = [&](int rank) { return owned_tiles(r, rank) * r.tile_w * r.tile_h * r.channels * 4; };
    if (r.world == 1)
        return;
    if (r.root_mode == SHRAY_DIST_ROOT0) {
        if (r.rank != 0) {
            if (frame_bytes(r.rank) > 0)
                sends[(*ns)++] = shray_dist_xfer{0, -1, 0, (int64_t)(count - 1) * p.wire_frame_stride_bytes + frame_bytes(r.rank)};
        } else {
            for (int peer = 1; peer < r.world; peer++)
                if (frame_bytes(peer) > 0)
                    recvs[(*nr)++] = shray_dist_xfer{peer, -1, (int64_t)peer * p.gather_rank_stride_bytes,
                                                     (int64_t)(count - 1) * p.gather_frame_stride_bytes + frame_bytes(peer)};
        }
        return;
    }
    for (int f = 0; f < count; f++) {
        const int owner = f % r.world, slot = f / r.world;
        if (owner != r.rank) {
            if (frame_bytes(r.rank) > 0)
                sends[(*ns)++] = shray_dist_xfer{owner, f, (int64_t)f * p.wire_frame_stride_bytes, frame_bytes(r.rank)};
        } else {
            for (int peer = 0; peer < r.world; peer++)
                if (peer != r.rank && frame_bytes(peer) > 0)
                    recvs[(*nr)++] = shray_dist_xfer{peer, f, (int64_t)peer * p.gather_rank_stride_bytes +
                                                                  (int64_t)slot * p.gather_frame_stride_bytes, frame_bytes(peer)};
        }
    }
}

}   // namespace shray_dist_plan_detail
