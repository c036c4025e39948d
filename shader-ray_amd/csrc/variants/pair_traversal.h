// variants/pair_traversal.h -- kernel id 3: both children of a node per turn (R3.2: exact, slower).  Included by
// kernel_stack_tally.hip, the one translation unit that instantiates the pair instances; wave_traversal.h / leaf_stage.h only
// declare inner_stage_pair, retest_stage and lane_pop.
#pragma once

#include "stack_traversal.h"

namespace shray {

// ---------------------------------------------------------------------------------------------------
// Pair traversal: both children of a node in one turn.
//
// The reference visits a branch, then its near child, and its far child when everything under the near child is
// done (fs:395-433): two visits, two dependent fetches.  Here the record of an entered branch holds BOTH children's
// boxes (PackedPair, packed_layout.h): one fetch, two slab tests -- the near child's visit is decided at once, the
// far child's visit is prepared and made when the reference makes it: when its word comes off the ray's stack.
// What the far child's visit decides then is `!(r0 >= r1) && r0 < hit.t` (fs:400) with the hit distance of THAT
// moment; r0 and r1 do not depend on the moment, so the word carries the first clause as a marker and r0 truncated
// to the bits the node index leaves free (lower bound L <= r0 < U, its neighbour): `hit.t <= L` fails and
// `U <= hit.t` passes with certainty; in between, and for every leaf (whose triangle tests need the exact clipped
// range, fs:327-331), the child's own record is fetched and tested exactly as the reference does (LT_RETEST, served
// at the start of the next leaf stage).  Every visit is counted where the reference makes it -- the near child's in
// its parent's turn, a far child's when it is popped, whether it then passes, fails or was known to fail -- so the
// iteration cap (fs:426-438) and the work counters come out exactly as in the one-visit-per-turn form.
#ifndef SHRAY_PAIR_TURNS
#define SHRAY_PAIR_TURNS 2
#endif

// the stack word of a pending child: (index | axis << IB | leaf << (IB + 2)) | r0's top 29 - IB bits << (IB + 3)
__device__ __forceinline__ uint32_t pair_stack_word(uint32_t link, float r0, bool range_not_empty, uint32_t ib)
{
    const uint32_t s = ib + 3u;
    const uint32_t low = (link & ((1u << ib) - 1u)) | ((link >> kPairAxisShift) << ib);
    const uint32_t q = range_not_empty ? ((__float_as_uint(r0) & 0x7fffffffu) >> (s - 1u)) : (0xffffffffu >> s);
    return low | (q << s);
}

// The visit is over: take the next pending child off the stack -- or several, while they fail.  Returns the lane's
// next state: LT_WALK (t.node = a branch that is entered), LT_RETEST (t.node = a child whose own record decides),
// LT_ENDED (the stack is empty, or the iteration cap struck: hit.t = -1).
template <bool COUNT, int BLOCK>
__device__ __forceinline__ int lane_pop(const SceneView &sc, LaneTraversal &t, uint32_t *stack, RayCounters &rc)
{
    const uint32_t ib = sc.pair_index_bits, s = ib + 3u;
    for (;;) {
        if (t.top == stack)
            return LT_ENDED;           // finished: the cap does not apply to a finished ray
        t.top -= BLOCK;
        const uint32_t word = *t.top;
        if (--t.left == 0) {           // fs:426-438: the visit that was about to be made is one too many
            t.hit.t = -1.0f;
            return LT_ENDED;
        }
        const uint32_t link = (word & ((1u << ib) - 1u)) | (((word >> ib) & 7u) << kPairAxisShift);
        if (COUNT) {
            rc.node_visits++;
            if (link & kLeafFlag)
                rc.leaf_visits++;
        }
        const uint32_t q = word >> s;
        if (q == (0xffffffffu >> s))
            continue;                  // its box range was empty: the visit fails whatever hit.t is
        const float lower = __uint_as_float(q << (s - 1u)), upper = __uint_as_float((q + 1u) << (s - 1u));
        if (t.hit.t <= lower)
            continue;                  // r0 >= lower >= hit.t
        t.node = link;
        return (!(link & kLeafFlag) && upper <= t.hit.t) ? LT_WALK : LT_RETEST;   // r0 < upper <= hit.t: entered
    }
}

// One turn of a lane in LT_WALK: t.node is a branch that has been entered; its near child is visited now.
template <bool COUNT, int BLOCK>
__device__ __forceinline__ int lane_pair_turn(const SceneView &sc, LaneTraversal &t, uint32_t *stack, RayCounters &rc)
{
    const uint32_t node = t.node;
    const bool neg_first = (t.positive_dir >> ((node >> kPairAxisShift) & 3u)) & 1u;
    // the near child's half of the record first (32-byte halves: negative child, positive child)
    const uint32_t near_at = ((node & kPairIndexMask) << 6) + (neg_first ? 0u : 32u);
    const char *base = reinterpret_cast<const char *>(sc.pair_nodes);
    const float4 *np = reinterpret_cast<const float4 *>(base + near_at), *fp = reinterpret_cast<const float4 *>(base + (near_at ^ 32u));
    const float4 nlo = np[0], nhi = np[1], flo = fp[0], fhi = fp[1];
    if (--t.left == 0) {               // the near child's visit would be one too many (fs:426-438)
        t.hit.t = -1.0f;
        return LT_ENDED;
    }
    const uint32_t near_link = __float_as_uint(nlo.w), far_link = __float_as_uint(flo.w);
    if (COUNT) {
        rc.node_visits++;
        if (near_link & kLeafFlag)
            rc.leaf_visits++;
    }
    float n0, n1, f0, f1;
    slab_range<false>(t, nlo, nhi, n0, n1);
    slab_range<false>(t, flo, fhi, f0, f1);
    *t.top = pair_stack_word(far_link, f0, !(f0 >= f1), sc.pair_index_bits);
    t.top += BLOCK;
    if (!(n0 >= n1) && (n0 < t.hit.t)) {
        if (near_link & kLeafFlag) {
            const uint32_t count = min((near_link >> kPairCountShift) & kPairCountMask, t.leaf_cap);
            if (count > 0) {
                asm volatile("v_mov_b32 %0, %4\n\tv_mov_b32 %1, %5\n\tv_mov_b32 %2, %6\n\tv_mov_b32 %3, %7"
                             : "+v"(t.leaf_first), "+v"(t.leaf_count), "+v"(t.leaf_r0), "+v"(t.leaf_r1)
                             : "v"(__float_as_uint(nhi.w)), "v"(count), "v"(n0), "v"(n1));
                return LT_LEAF;
            }
        } else {
            t.node = near_link;
            return LT_WALK;
        }
    }
    return lane_pop<COUNT, BLOCK>(sc, t, stack, rc);
}

template <bool COUNT, int BLOCK>
__device__ __forceinline__ void inner_stage_pair(const SceneView &sc, LaneTraversal &t, int &state, uint32_t *stack,
                                                 RayCounters &rc, int keep_walking)
{
    for (;;) {
        if (!wave_ballot(state == LT_WALK))
            return;
#pragma unroll
        for (int turn = 0; turn < SHRAY_PAIR_TURNS; turn++)
            if (state == LT_WALK)
                state = lane_pair_turn<COUNT, BLOCK>(sc, t, stack, rc);
        const int walking = __popcll(wave_ballot(state == LT_WALK));
        if (walking < keep_walking && wave_ballot((state & 1) == 0))
            return;
    }
}

// Start of a leaf stage: the lanes in LT_RETEST fetch their node's own record and make its visit's test exactly.
template <bool COUNT, int BLOCK>
__device__ __forceinline__ void retest_stage(const SceneView &sc, LaneTraversal &t, int &state, uint32_t *stack, RayCounters &rc)
{
    if (!wave_ballot(state == LT_RETEST))
        return;
    if (state == LT_RETEST) {
        float4 lo, hi;
        // (the node's own record, from the copy that holds the scene's boxes as they are: the last one)
        load_packed_node(sc, ((t.node & kPairIndexMask) << kNodeShift) + 7u * sc.packed_nodes_bytes, lo, hi);
        float r0, r1;
        slab_range<false>(t, lo, hi, r0, r1);
        const uint32_t a = __float_as_uint(lo.w), b = __float_as_uint(hi.w);
        state = LT_ENDED;              // placeholder: decided below
        bool entered = false;
        if (!(r0 >= r1) && (r0 < t.hit.t)) {
            if (b & kLeafFlag) {
                const uint32_t count = min(b & ~kLeafFlag, t.leaf_cap);
                if (count > 0) {
                    t.leaf_first = a;
                    t.leaf_count = count;
                    t.leaf_r0 = r0;
                    t.leaf_r1 = r1;
                    state = LT_LEAF;
                    entered = true;
                }
            } else {
                state = LT_WALK;       // t.node already carries the branch's index and split axis
                entered = true;
            }
        }
        if (!entered)
            state = lane_pop<COUNT, BLOCK>(sc, t, stack, rc);
    }
}

}   // namespace shray
