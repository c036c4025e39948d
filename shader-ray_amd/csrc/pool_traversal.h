// pool_traversal.h -- group_intersect (raytracer.es.fs:386-443) for a whole 256-thread workgroup at once:
// the workgroup's rays form a pool, and its four waves MERGE while they traverse.
//
// Why: in the stack kernel a wave keeps the 64 rays it was born with.  Rays end at very different times
// (a miss at the root box, a deep walk along a silhouette), secondary rays exist only where something
// was hit, and every instruction the wave issues costs the same whether 64 lanes or 3 are active.  On the
// 1M-triangle scene only 36 % of the lanes are active per vector instruction (profiles/r01); replaying
// the CPU oracle's per-ray traces through the scheduling policies (oracle/tools/wave_sim.py) shows most
// idle lanes belong to rays that have ended or never existed in this traversal -- and that compaction at
// bounce boundaries alone recovers little, while merging half-empty waves DURING a traversal recovers most.
//
// How: a ray's traversal state (wave_traversal.h: LaneTraversal) is position-independent -- its stack
// lives in an LDS column `col` that travels with the ray, so any lane of any wave can continue it.
// The waves run the traversal in epochs of kPoolEpochTurns node turns.  Between epochs they meet at a
// barrier, publish how many live rays each holds, and all evaluate the same plan: if the two waves with
// the fewest live rays fit into one, the smaller (donor) writes its rays to a 5 KB exchange buffer and
// the other (receiver) adopts them in its free lanes.  A wave without rays only keeps the barriers
// company.  When a ray ends, the lane that holds it writes the hit into the first four levels of the
// ray's own stack column (free by then), where the owning thread -- the pixel's thread, which keeps the
// shading state -- collects it after the last barrier.
//
// Per-ray arithmetic, visit order, iteration cap and leaf cap are exactly those of the stack kernel:
// only WHICH lane executes a ray's next step changes, so frames and work counters stay bit-identical.
#pragma once

#include "wave_traversal.h"

namespace shray {

#ifndef SHRAY_POOL_EPOCH_TURNS
#define SHRAY_POOL_EPOCH_TURNS 8
#endif
#ifndef SHRAY_POOL_KEEP
#define SHRAY_POOL_KEEP 28
#endif
#ifndef SHRAY_POOL_KEEP_FLOOR
#define SHRAY_POOL_KEEP_FLOOR 2
#endif
#ifndef SHRAY_POOL_MERGE_SLACK
#define SHRAY_POOL_MERGE_SLACK 0   // merge two waves when their live rays sum to <= 64 - slack
#endif
constexpr int kPoolEpochTurns = SHRAY_POOL_EPOCH_TURNS;
constexpr int kPoolXbufDwords = 64 * 20;    // one wave's rays, 20 dwords each
constexpr int kPoolCountDwords = 8;         // live-ray counts of the four waves, double-buffered by epoch parity
constexpr int kPoolMinLevels = 4;           // a column must hold the four result words

template <int BLOCK>
struct PoolTraversal {
    static constexpr int block_size = BLOCK;
    uint32_t *stack;    // LDS: levels x BLOCK, column-major ([level][column])
    uint32_t *xbuf;     // LDS: kPoolXbufDwords
    uint32_t *counts;   // LDS: kPoolCountDwords
    // which half of `counts` the next epoch publishes into.  It alternates per epoch ACROSS closest() calls: a
    // parity restarted at every call would let a fast wave's epoch-0 write of the next call land in the half a
    // slower wave is still reading for the last epoch of this one (no barrier lies between the two)
    unsigned int parity = 0;
#ifdef SHRAY_DIAGNOSTICS
    unsigned long long diag_tally[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif

    // Collective: every thread of the workgroup calls it (has_ray = this thread's pixel has a ray to trace).
    // Returns the number of rays the workgroup traced (uniform); `hit` is set for has_ray threads.
    // (ANY_HIT: accepted for the driver's shadow rays; this kernel walks them to the end)
    template <bool COUNT, bool ANY_HIT = false, bool TIMED_FORM = false>
    __device__ __forceinline__ int closest(const SceneView &sc, const FrameView &fr, bool has_ray, V3 P, V3 D, Hit &hit,
                                           RayCounters &rc)
    {
        const unsigned int lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
        LaneTraversal t;
        lane_begin<COUNT>(sc, fr, t, stack + threadIdx.x, P, D, rc, has_ray);
        bool busy = has_ray;            // this lane holds a live ray (its own or an adopted one)
        int state = busy ? LT_WALK : LT_ENDED;
        uint32_t col = threadIdx.x;     // the held ray's stack column = its owner's thread index; travels with the ray
        int first_total = 0;

        for (int epoch = 0;; epoch++) {
            // ---- the waves meet: live rays per wave (every live ray is in LT_WALK here)
            const unsigned long long live_mask = wave_ballot(busy);
            uint32_t *cnt = counts + 4 * parity;
            parity ^= 1u;
            if (lane == 0)
                cnt[wave] = (uint32_t)__popcll(live_mask);
            __syncthreads();
            int c[4];
#pragma unroll
            for (int w = 0; w < 4; w++)
                c[w] = __builtin_amdgcn_readfirstlane((int)cnt[w]);
            const int total = c[0] + c[1] + c[2] + c[3];
            if (epoch == 0)
                first_total = total;
            if (total == 0)
                break;

            // ---- the plan, identical in every wave: the two waves with the fewest live rays merge if they fit
            int ia = -1, ib = -1, ca = 0x7fffffff, cb = 0x7fffffff;
#pragma unroll
            for (int w = 0; w < 4; w++) {
                if (c[w] == 0)
                    continue;
                if (c[w] < ca) {
                    ib = ia;
                    cb = ca;
                    ia = w;
                    ca = c[w];
                } else if (c[w] < cb) {
                    ib = w;
                    cb = c[w];
                }
            }
            if (ib >= 0 && ca + cb <= 64 - SHRAY_POOL_MERGE_SLACK) {
                if ((int)wave == ia && busy) {                  // donor: every live ray leaves
                    const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(live_mask >> 32),
                                                                    __builtin_amdgcn_mbcnt_lo((uint32_t)live_mask, 0u));
                    float4 *slot = reinterpret_cast<float4 *>(xbuf) + 5u * rank;
                    slot[0] = make_float4(t.P.x, t.P.y, t.P.z, t.D.x);
                    slot[1] = make_float4(t.D.y, t.D.z, t.Y.x, t.Y.y);
                    slot[2] = make_float4(t.Y.z, 0.0f, 0.0f, 0.0f);
                    slot[3] = make_float4(t.hit.t, t.hit.which, t.hit.bu, t.hit.bv);
                    slot[4] = make_float4(__uint_as_float(t.node), __uint_as_float((uint32_t)(t.top - (stack + col))), __uint_as_float((uint32_t)t.left),
                                          __uint_as_float(col | (t.divide ? 0x80000000u : 0u)));
                    busy = false;
                    state = LT_ENDED;
                }
                __syncthreads();
                if ((int)wave == ib) {                          // receiver: its free lanes adopt them
                    const unsigned long long free_mask = ~live_mask;
                    const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(free_mask >> 32),
                                                                    __builtin_amdgcn_mbcnt_lo((uint32_t)free_mask, 0u));
                    if (!busy && rank < (uint32_t)ca) {
                        const float4 *slot = reinterpret_cast<const float4 *>(xbuf) + 5u * rank;
                        const float4 s0 = slot[0], s1 = slot[1], s2 = slot[2], s3 = slot[3], s4 = slot[4];
                        t.P = mk(s0.x, s0.y, s0.z);
                        t.D = mk(s0.w, s1.x, s1.y);
                        t.Y = mk(s1.z, s1.w, s2.x);
                        t.hit = Hit{s3.x, s3.y, s3.z, s3.w};
                        t.node = __float_as_uint(s4.x);
                        t.left = (int)__float_as_uint(s4.z);
                        const uint32_t packed = __float_as_uint(s4.w);
                        col = packed & 0x7fffffffu;
                        t.top = stack + col + __float_as_uint(s4.y);   // the ray's column travels with it; depth in words
                        t.divide = (packed >> 31) != 0u;
                        t.fx = t.D.x >= 0.0f;
                        t.fy = t.D.y >= 0.0f;
                        t.fz = t.D.z >= 0.0f;
                        t.positive_dir = (t.D.x > 0.0f ? 1u : 0u) | (t.D.y > 0.0f ? 2u : 0u) | (t.D.z > 0.0f ? 4u : 0u);
                        t.octant = octant_offset(sc, t.fx, t.fy, t.fz);
                        busy = true;
                        state = LT_WALK;
                    }
                }
            }

            // ---- one epoch of the wave-cooperative traversal (wave_traversal.h).  A leaf a lane parks in is
            //      finished before the epoch ends, so every live ray is in LT_WALK when the waves meet again.
            uint32_t *column = stack + col;
            t.divide_mask = wave_ballot(t.divide);     // rays have moved between lanes
            int turns = 0;
            while (turns < kPoolEpochTurns && wave_ballot(state != LT_ENDED)) {
                const int alive = __popcll(wave_ballot(state != LT_ENDED));
                const int keep = max(SHRAY_POOL_KEEP_FLOOR, (alive * SHRAY_POOL_KEEP + 32) >> 6);
                while (wave_ballot(state == LT_WALK)) {
#pragma unroll
                    for (int turn = 0; turn < 2; turn++) {
                        if (state == LT_WALK) {
                            lane_count_visit(t);
                            lane_apply_cap(t, state);     // (at every visit: a capped ray must not travel to another wave)
                            if (state == LT_WALK) {
                                float4 lo, hi;
                                load_packed_node(sc, node_address(t, t.node), lo, hi);
                                state = lane_visit_loaded<COUNT, BLOCK>(fr, t, column, rc, lo, hi);
                            }
                        }
                    }
                    turns += 2;
                    const int walking = __popcll(wave_ballot(state == LT_WALK));
                    if ((walking < keep && wave_ballot(state == LT_LEAF)) || turns >= kPoolEpochTurns)
                        break;
                }
                leaf_stage<COUNT, BLOCK>(sc, fr, t, state, column, rc, nullptr SHRAY_DIAG_ARG);
            }

            // ---- a ray that ended in this epoch leaves its hit in the first four levels of its own column
            if (busy && state == LT_ENDED) {
                column[0 * BLOCK] = __float_as_uint(t.hit.t);
                column[1 * BLOCK] = __float_as_uint(t.hit.which);
                column[2 * BLOCK] = __float_as_uint(t.hit.bu);
                column[3 * BLOCK] = __float_as_uint(t.hit.bv);
                busy = false;
            }
        }
        // the owner collects (the last barrier above ordered every wave's result stores before this)
        if (has_ray) {
            const uint32_t *own = stack + threadIdx.x;
            hit.t = __uint_as_float(own[0 * BLOCK]);
            hit.which = __uint_as_float(own[1 * BLOCK]);
            hit.bu = __uint_as_float(own[2 * BLOCK]);
            hit.bv = __uint_as_float(own[3 * BLOCK]);
        }
        return first_total;
    }
};

}   // namespace shray
