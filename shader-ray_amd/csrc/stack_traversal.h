// stack_traversal.h -- per-ray stack traversal over the packed layout (packed_layout.h),
// one traversal per lane at a time (used by kernel_stack.hip through uniform_driver.h / trace_common.h).
// Same visits, same order, same arithmetic as the reference's threaded traversal
// (raytracer.es.fs:386-443); what changes is where the data comes from and how the wave
// schedules the work:
//   * one node = two dwordx4 loads (box + links) instead of 3-4 texture fetches
//   * the ray's pending far children live in LDS, laid out [level][thread] so that every
//     lane always addresses its own bank (no conflicts at any mix of depths)
//   * triangle edges are precomputed
//   * the slab test's six true divisions (fs:204-213) become multiplications by the ray's
//     correctly rounded reciprocal plus two FMA refinements that land on the same correctly
//     rounded quotient (exact_div.h); rays or scenes outside the proven operand ranges keep
//     dividing
//   * node visits and triangle tests are stages of one wave-cooperative loop (wave_traversal.h: inner_stage; leaf_stage.h:
//     leaf_stage / leaf_stage_dealt) so that one lane's leaf does not stall the other 63
#pragma once

#include "wave_traversal.h"
#include "visit_asm.h"

namespace shray {

// The node loop yields to the leaf stage (when lanes are parked) once fewer than a threshold of lanes
// are still walking: SHRAY_KEEP_WALKING / 64 of the lanes still in this traversal (at least SHRAY_KEEP_FLOOR)
// -- a wave with eight live lanes should not leave the node loop after every visit.  Numeric knobs, -D overrides
// them for A/B builds (round 2 sweeps: profiles/EXPERIMENTS.md R2.7).
#ifndef SHRAY_KEEP_WALKING
#define SHRAY_KEEP_WALKING 36
#endif
constexpr int kStackKeepWalking = SHRAY_KEEP_WALKING;
// with the dealt leaf stage a leaf stage with few parked lanes is cheap: the node loop yields as soon as ANY lane is parked
// (64 of 64).  Measured 48 ... 64 for the one-sample instances on the orbit workload in round 3 (+1.3 % at 64,
// profiles/history/r03/dealt_keep_walking_ab.txt) and for the multi-sample dealing instance of the 1M-triangle scene in round 4
// (40 / 48 / 56 / 64: 2.49 / 2.43 / 2.44 / 2.40 ms, profiles/r04/dealt_keep_walking_ab.txt)
#ifndef SHRAY_KEEP_WALKING_DEALT
#define SHRAY_KEEP_WALKING_DEALT 64
#endif
constexpr int kStackKeepWalkingDealt = SHRAY_KEEP_WALKING_DEALT;
#ifndef SHRAY_KEEP_FLOOR
#define SHRAY_KEEP_FLOOR 2
#endif
// DEAL: the convergent form's leaf stage deals triangles to idle lanes (leaf_stage.h: leaf_stage_dealt)
// PAIR: both children of a node per turn (variants/pair_traversal.h: inner_stage_pair); convergent form only
// CACHE: the sequential leaf loop reads a stage's distinct leaves from the wave's LDS slab (leaf_cache.h; `ids` is followed by it)
// ROOMY: an instance compiled for six waves per SIMD: the dealt stage's rounds are hand-scheduled too (leaf_stage.h: dealt_search)
template <int BLOCK, bool DEAL = true, bool PAIR = false, bool CACHE = false, bool ROOMY = false>
struct StackTraversal {
    static constexpr int block_size = BLOCK;
    uint32_t *stack;   // LDS, this thread's column: stack[level * BLOCK]
    uint8_t *ids;      // LDS, 64 bytes per wave: scratch of the dealt leaf stage (leaf_stage.h)
#ifdef SHRAY_DIAGNOSTICS
    unsigned long long diag_tally[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif

    // Convergent form (uniform_driver.h): all 64 lanes of the wave call it together; has_ray = this lane's pixel
    // has a ray to trace.  Lanes without one take part in the dealt leaf stage as workers.  Returns the number
    // of rays the wave traced.
    // ANY_HIT (shadow rays of the timed kernels, uniform_driver.h): the caller only asks whether hit.t stays at
    // "infinitely far" (fs:516-521: lit = shadow.t >= far).  Once a leaf has produced a hit the answer is "no" whatever
    // the rest of the walk finds -- a closer hit, or the iteration cap's bad hit (t = -1) -- so the ray stops there.
    // The counting twins walk on: their tallies are compared with the reference's full traversal -- unless they are
    // asked for the tallies of the timed form itself (TIMED_FORM).
    template <bool COUNT, bool ANY_HIT = false, bool TIMED_FORM = false>
    __device__ __forceinline__ int closest(const SceneView &sc, const FrameView &fr, bool has_ray, V3 P, V3 D, Hit &hit,
                                           RayCounters &rc)
    {
        const int traced = __popcll(wave_ballot(has_ray));
        if (!traced)
            return 0;
        LaneTraversal t;
        lane_begin<COUNT>(sc, fr, t, stack, P, D, rc, has_ray);
        int state = has_ray ? LT_WALK : LT_ENDED;
        if (PAIR) {
            // the root's visit (its own box, fs:395's first iteration) is made by the first retest stage
            t.node = sc.pair_root_link;
            state = has_ray ? LT_RETEST : LT_ENDED;
            if (COUNT && has_ray) {
                rc.node_visits++;
                if (sc.pair_root_link & kLeafFlag)
                    rc.leaf_visits++;
            }
        }
        run<COUNT, true, ANY_HIT && (!COUNT || TIMED_FORM)>(sc, fr, t, state, rc);
        hit = t.hit;
        return traced;
    }

    template <bool COUNT>
    __device__ __forceinline__ void closest(const SceneView &sc, const FrameView &fr, V3 P, V3 D, Hit &hit,
                                            RayCounters &rc)
    {
        LaneTraversal t;
        lane_begin<COUNT>(sc, fr, t, stack, P, D, rc);
        int state = LT_WALK;
        run<COUNT, false>(sc, fr, t, state, rc);
        hit = t.hit;
    }

    // CONVERGED: every lane of the wave is executing (the dealt leaf stage may use them all)
    template <bool COUNT, bool CONVERGED, bool ANY_HIT = false>
    __device__ __forceinline__ void run(const SceneView &sc, const FrameView &fr, LaneTraversal &t, int &state, RayCounters &rc)
    {
        do {
#if defined(SHRAY_DIAGNOSTICS) && !defined(SHRAY_DIAG_KHIST) && !defined(SHRAY_DIAG_UNIFORM)
            const unsigned long long c0 = __builtin_amdgcn_s_memtime();
#endif
            // the threshold scales with the lanes still in this traversal: kStackKeepWalking of 64
            const int alive = __popcll(wave_ballot(state != LT_ENDED));
            const int keep = max(SHRAY_KEEP_FLOOR, (alive * ((DEAL && CONVERGED) ? kStackKeepWalkingDealt : kStackKeepWalking) + 32) >> 6);
            if constexpr (PAIR) {        // (variants/pair_traversal.h: kernel id 3's translation unit includes it)
                inner_stage_pair<COUNT, BLOCK>(sc, t, state, stack, rc, keep);
                retest_stage<COUNT, BLOCK>(sc, t, state, stack, rc);
            } else {
#if SHRAY_ASM_VISIT && !defined(SHRAY_DIAGNOSTICS)
                // the timed instances: the node stage as one hand-scheduled statement (visit_asm.h); the counting twins keep
                // the compiler's form (their tallies, and the cap in front of every visit)
                if (!COUNT)
                    inner_stage_scheduled<BLOCK>(sc, fr, t, state, stack, rc, keep);
                else
#endif
                    inner_stage<COUNT, BLOCK>(sc, fr, t, state, stack, rc, keep, false SHRAY_DIAG_ARG);
            }
#if defined(SHRAY_DIAGNOSTICS) && !defined(SHRAY_DIAG_KHIST) && !defined(SHRAY_DIAG_UNIFORM)
            const unsigned long long c1 = __builtin_amdgcn_s_memtime();
#endif
            if (DEAL && CONVERGED)
                leaf_stage_dealt<COUNT, BLOCK, PAIR, CACHE, ROOMY>(sc, fr, t, state, stack, rc, ids SHRAY_DIAG_ARG);
            else
                leaf_stage<COUNT, BLOCK, PAIR, CACHE>(sc, fr, t, state, stack, rc, ids SHRAY_DIAG_ARG);
            if (ANY_HIT && state != LT_ENDED && t.hit.t < kFar)
                state = LT_ENDED;   // a hit: the shadow query is answered (a capped ray, t = -1, has ended already)
#if defined(SHRAY_DIAGNOSTICS) && !defined(SHRAY_DIAG_KHIST) && !defined(SHRAY_DIAG_UNIFORM)
            const unsigned long long c2 = __builtin_amdgcn_s_memtime();
            diag_tally[2] += c1 - c0;
            diag_tally[3] += c2 - c1;
#endif
        } while (wave_ballot(state != LT_ENDED));
    }
};

}   // namespace shray
