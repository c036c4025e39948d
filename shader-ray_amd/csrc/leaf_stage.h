// leaf_stage.h -- the leaf side of the wave-cooperative BVH walk (wave_traversal.h has the node side): triangle_intersect
// (raytracer.es.fs:297-346) in its two halves, the sequential leaf loop, the end of a leaf stage, and the DEALT leaf stage (a parked
// ray's triangles go to the wave's idle lanes).  The timed instances run the loops as hand-scheduled statements (leaf_asm.h, included
// at the end); what is here is the compiler's form: the counting twins, the rare rounds the statements hand back, the variants.
#pragma once

#include "wave_traversal.h"

namespace shray {

// (the pair traversal, variants/pair_traversal.h: kernel id 3 only)
template <bool COUNT, int BLOCK>
__device__ __forceinline__ void inner_stage_pair(const SceneView &sc, LaneTraversal &t, int &state, uint32_t *stack, RayCounters &rc, int keep_walking);
template <bool COUNT, int BLOCK>
__device__ __forceinline__ void retest_stage(const SceneView &sc, LaneTraversal &t, int &state, uint32_t *stack, RayCounters &rc);

// 1 / det of triangle_intersect (fs:314).  A determinant the shader goes on with is at least 1e-7 (its early-out, fs:312);
// below 2^100 (every scene of finite size) the three-instruction reciprocal of exact_div.h IS the correctly rounded
// quotient; a larger one, or NaN, takes the true division (the wave skips it).  (triangle_candidate applies the early-out
// after the arithmetic -- a conjunction --: what this returns for a determinant below 1e-7 is never looked at.)
__device__ __forceinline__ float reciprocal_of_determinant(float det)
{
    float inv = reciprocal_in_range(det);
#ifdef SHRAY_COST_MAIN_PATH     // profiles/isa_costs.hip counts the path every wave takes, not the division the rare one adds
    return inv;
#endif
    const bool large = !(fabsf(det) < 0x1p100f);
    if (__builtin_expect(wave_ballot(large) != 0ull, 0)) {
        asm volatile("; determinant outside the reciprocal's domain" ::: "memory");   // keeps this a branch
        if (large)
            inv = 1.0f / det;
    }
    return inv;
}

// triangle_intersect (fs:297-346) in its two halves (profiles/isa_costs.py counts each in isolation).
// First half, fs:307-331: determinant, distance, the early-outs against the determinant's epsilon, the closest hit so
// far and the leaf's clipped range.  Returns false where the shader returns.
struct TriangleSetup {
    V3 M, T, Q;
    float inv_det, dist;
};
__device__ __forceinline__ bool triangle_distance(const LaneTraversal &t, const float4 q0, const float4 q1, const float4 q2,
                                                  TriangleSetup &s)
{
    const V3 v0 = mk(q0.x, q0.y, q0.z), e0 = mk(q0.w, q1.x, q1.y), e1 = mk(q1.z, q1.w, q2.x);
    s.M = cross3(e1, t.D);
    const float det = dot3(e0, s.M);
    if (det > -0.0000001f && det < 0.0000001f)
        return false;
    s.inv_det = reciprocal_of_determinant(det);
    s.T = t.P - v0;
    s.Q = cross3(s.T, e0);
    s.dist = -dot3(e1, s.Q) * s.inv_det;
    // `d > hit.t || d > r1` is `d > min(hit.t, r1)` whatever is NaN (the hardware minimum returns the other operand, and a
    // comparison with NaN is false either way); the compiler makes the same fold but canonicalises both operands first
    // (two v_max x, x per test).  One bare v_min_f32:
    float upper;
    asm("v_min_f32 %0, %1, %2" : "=v"(upper) : "v"(t.hit.t), "v"(t.leaf_r1));
    return !(s.dist > upper || s.dist < t.leaf_r0);
}
// Second half, fs:333-346: the barycentric tests and the store.
// BOUNDS: the lane parked bounds of its leaf's range, not the range (lane_visit_loaded): a candidate that has passed
// everything else and lies within 2^-19 of an end is held against the exact range before it is stored.
template <bool BOUNDS>
__device__ __forceinline__ void triangle_barycentrics(const SceneView &sc, LaneTraversal &t, uint32_t which, const TriangleSetup &s)
{
    const float u = dot3(s.T, s.M) * s.inv_det;
    if (u < 0.0f || u > 1.0f)
        return;
    const float w = dot3(t.D, s.Q) * s.inv_det;
    if (w < 0.0f || u + w > 1.0f)
        return;
    if (BOUNDS) {
        const bool near_end = near_range_end(s.dist, t.leaf_r0, t.leaf_r1);
        if (__builtin_expect(wave_ballot(near_end) != 0ull, 0)) {
            asm volatile("; a candidate at an end of its leaf's range: the exact range" ::: "memory");   // keeps this a branch
            if (near_end) {
                float e0, e1;
                exact_leaf_range(sc, t, e0, e1);
                if (s.dist < e0 || s.dist > e1)
                    return;                       // fs:329-331
            }
        }
    }
    t.hit.which = (float)which;
    t.hit.t = s.dist;
    t.hit.bu = u;
    t.hit.bv = w;
}

// triangle_intersect of triangle `which` (its three 16-byte words) for a parked lane
template <bool COUNT, bool BOUNDS>
__device__ __forceinline__ void lane_test_triangle_loaded(const SceneView &sc, LaneTraversal &t, uint32_t which, RayCounters &rc,
                                                          const float4 q0, const float4 q1, const float4 q2)
{
    if (COUNT)
        rc.triangle_tests++;
    TriangleSetup s;
    if (triangle_distance(t, q0, q1, q2, s))
        triangle_barycentrics<BOUNDS>(sc, t, which, s);
}

// The nine floats of a packed triangle, fetched as three 12-byte loads issued back to back and handed on as the
// three words {v0, e0.x} {e0.yz, e1.xy} {e1.z} the tests unpack.  (Left to itself the compiler splits the loads and
// sinks part of them behind the `det` early-out of the test, which costs a second dependent memory round trip per
// triangle: hence the pin below.)
struct PackedF3 {
    float x, y, z;
};
__device__ __forceinline__ void load_packed_triangle_at(const SceneView &sc, uint32_t byte_offset, float4 &q0, float4 &q1, float4 &q2);
__device__ __forceinline__ void load_packed_triangle(const SceneView &sc, uint32_t index, float4 &q0, float4 &q1, float4 &q2)
{
    load_packed_triangle_at(sc, index * 36u, q0, q1, q2);
}
__device__ __forceinline__ void load_packed_triangle_at(const SceneView &sc, uint32_t byte_offset, float4 &q0, float4 &q1, float4 &q2)
{
    // base + 32-bit byte offset, as for the nodes
    // 36 bytes as 16 + 16 + 4 (what the back end makes of three 12-byte loads anyway), pinned as the register tuples the
    // loads fill: pinned component by component, every test began with five or six moves out of those tuples
    typedef float f4 __attribute__((ext_vector_type(4)));
    typedef f4 __attribute__((aligned(4), may_alias)) packed_f4;
    const char *p = reinterpret_cast<const char *>(sc.packed_tris) + byte_offset;
    f4 a = *reinterpret_cast<const packed_f4 *>(p), b = *reinterpret_cast<const packed_f4 *>(p + 16);
    float c = *reinterpret_cast<const float *>(p + 32);
    asm volatile("" : "+v"(a), "+v"(b), "+v"(c));
    q0 = make_float4(a.x, a.y, a.z, a.w);
    q1 = make_float4(b.x, b.y, b.z, b.w);
    q2 = make_float4(c, 0.0f, 0.0f, 0.0f);
}


template <bool COUNT, int BLOCK>
__device__ __forceinline__ int lane_pop(const SceneView &sc, LaneTraversal &t, uint32_t *stack, RayCounters &rc);

// triangles a parked lane tests: its leaf's count (the flag bit may still be on it), at most max_leaf_tests (fs:411)
__device__ __forceinline__ uint32_t parked_count(uint32_t leaf_count, uint32_t leaf_cap) { return min(leaf_count & ~kLeafFlag, leaf_cap); }

// Leaf stage: every parked lane tests its leaf's triangles in order, then follows its link.
// PAIR: the lane belongs to the pair traversal (below): "follow the link" is lane_pop.
// its loop (some lane must be parked) ...
template <bool COUNT, bool BOUNDS>
__device__ __forceinline__ void leaf_loop(const SceneView &sc, LaneTraversal &t, int state, RayCounters &rc SHRAY_DIAG_PARAM)
{
#if defined(SHRAY_DIAGNOSTICS) && !defined(SHRAY_DIAG_KHIST) && !defined(SHRAY_DIAG_UNIFORM)
#include "variants/diag_leaf_loop.inc"
#endif
    // a lane that is not parked has no triangles: ONE comparison per round decides both who works and whether anyone does
    // (as `state == LT_LEAF && j < count` the wave-level test cost a select and a second comparison per round; the pin
    // keeps the compiler from turning it back into that)
    uint32_t mine = state == LT_LEAF ? parked_count(t.leaf_count, t.leaf_cap) : 0u;
    asm volatile("" : "+v"(mine));
    // bottom-tested: some lane is parked, and a parked lane has at least one triangle (a top-tested loop over a wave-level
    // condition is not rotated by the compiler, and then carries the hit's four fields in two register sets with a copy at
    // every level of the test's early-outs)
    uint32_t j = 0;
    do {
        SHRAY_DIAG_COUNT(1);
        if (j < mine) {
            float4 q0, q1, q2;
            SHRAY_DIAG_T0
            load_packed_triangle(sc, t.leaf_first + j, q0, q1, q2);
            SHRAY_DIAG_WAIT(5);
            lane_test_triangle_loaded<COUNT, BOUNDS>(sc, t, t.leaf_first + j, rc, q0, q1, q2);
        }
        j++;
    } while (wave_ballot(j < mine));
}

// The same loop for the timed instances as one hand-scheduled statement (leaf_asm.h, included at the end of this file); the
// counting twins, the pair traversal (exact parked ranges) and the diagnostic build keep the compiler's form.
#ifndef SHRAY_ASM_LEAF
#define SHRAY_ASM_LEAF 1
#endif
__device__ __forceinline__ void leaf_loop_scheduled(const SceneView &sc, LaneTraversal &t, int state, RayCounters &rc);
__device__ __forceinline__ bool dealt_rounds_scheduled(const SceneView &sc, const LaneTraversal &t, int source, uint32_t end, uint32_t G,
                                                       uint32_t tri, uint32_t where, float &best_d, float &best_u, float &best_w, uint32_t &best);
template <bool COUNT, bool BOUNDS>
__device__ __forceinline__ void leaf_loop_timed_or_counted(const SceneView &sc, LaneTraversal &t, int state, RayCounters &rc SHRAY_DIAG_PARAM)
{
#if SHRAY_ASM_LEAF && !defined(SHRAY_DIAGNOSTICS)
    if (!COUNT && BOUNDS) {
        leaf_loop_scheduled(sc, t, state, rc);
        return;
    }
#endif
    leaf_loop<COUNT, BOUNDS>(sc, t, state, rc SHRAY_DIAG_ARG_FWD);
}

}   // namespace shray

// the leaf cache (north_star's "triangle data staged into LDS tiles"; -DSHRAY_LEAF_CACHE=1 compiles it in): constants and helpers
#include "leaf_cache.h"

namespace shray {

// ... and its end: the parked lanes move on (fs:416-433)
template <bool COUNT, int BLOCK, bool PAIR>
__device__ __forceinline__ void leaf_finish(const SceneView &sc, LaneTraversal &t, int &state, uint32_t *stack, RayCounters &rc)
{
    if constexpr (PAIR) {    // (variants/pair_traversal.h)
        if (state == LT_LEAF)
            state = lane_pop<COUNT, BLOCK>(sc, t, stack, rc);
        return;
    }
    // a hit distance that is NaN (an unordered candidate was accepted, see leaf_stage_dealt) fails `r0 < hit.t` at every
    // later visit; the visit's fast test does not look for it: such a lane takes the exact branch from here on
    {
        const unsigned long long no_distance = wave_ballot(state == LT_LEAF && t.hit.t != t.hit.t);
        if (__builtin_expect(no_distance != 0ull, 0)) {
            asm volatile("; a hit distance that is NaN" ::: "memory");   // keeps this a branch
            if (state == LT_LEAF && t.hit.t != t.hit.t)
                t.divide = true;
            t.divide_mask |= no_distance;
        }
    }
    if (state == LT_LEAF)
        state = lane_advance<BLOCK>(t, stack, false, 0u);
}

// CACHE: `ids` is followed by the wave's leaf cache (above)
template <bool COUNT, int BLOCK, bool PAIR = false, bool CACHE = false>
__device__ __forceinline__ void leaf_stage(const SceneView &sc, const FrameView &fr, LaneTraversal &t, int &state,
                                           uint32_t *stack, RayCounters &rc, uint8_t *ids SHRAY_DIAG_PARAM)
{
    if (!wave_ballot(state == LT_LEAF))
        return;
    if (CACHE && t.leaf_cap <= kCacheTriangles)     // (uniform)
        leaf_loop_cached<COUNT, !PAIR>(sc, t, state, rc, reinterpret_cast<char *>(ids) + kIdsBytes SHRAY_DIAG_ARG_FWD);
    else
        leaf_loop_timed_or_counted<COUNT, !PAIR>(sc, t, state, rc SHRAY_DIAG_ARG_FWD);
    leaf_finish<COUNT, BLOCK, PAIR>(sc, t, state, stack, rc);
}

// ---------------------------------------------------------------------------------------------------
// Dealt leaf stage.  The plain stage above takes max(count) turns whoever is parked: one lane sitting in a
// 10-triangle leaf costs ten dependent fetch-and-test rounds with one lane active (the normal case in
// divergent waves: on the 1M-triangle scene the leaf loop runs at 24 % of its lanes, oracle/tools/wave_sim.py).
// Here the wave's idle lanes do the work: with K <= 32 lanes parked, each parked ray gets a group of
// G = 2, 4, 8 or 16 worker lanes (G * K <= 64); worker i of a group pulls the ray (ds_bpermute) and tests
// triangles i, i + G, ... of its leaf, so the stage takes ceil(count / G) rounds -- one memory round trip
// instead of up to ten when few lanes are parked.
//
// Exactness: triangle_intersect's outcome for one triangle depends on hit.t only through the early-out
// `d > hit.t` (fs:327); every other test is a pure function of (ray, triangle, leaf range).  Testing the
// leaf's triangles in order therefore ends with: among the candidates that pass those tests and have
// d <= the hit.t the leaf started with, the smallest d, and of equal d the LAST in order (equal d
// overwrites, the test is `>`).  Each worker keeps that rule over its own increasing j, the group combines
// by (smaller d, then larger j), the parked lane applies the winner.  Same arithmetic on the same values:
// bit-identical hits; the counting twin tallies the same triangle tests (in the worker lanes).
//
// That argument needs the candidates' d to be ORDERED.  A candidate whose d is NaN (a triangle so large that its
// determinant overflows to inf - inf, or a ray that already carries NaNs) fails none of the shader's comparisons:
// the sequential loop accepts it, and after it accepts whatever candidate comes next -- an order-dependent
// outcome no (d, j) ranking reproduces.  A worker that accepts an unordered d raises a flag; if any lane of the
// wave did, the stage discards the dealt result and runs the plain sequential loop over the untouched parked
// rays (tests/test_gpu_parity.py::test_nan_candidates_in_a_dealt_leaf).
#ifndef SHRAY_DEAL_MAX_PARKED
#define SHRAY_DEAL_MAX_PARKED 32   // at most 32: a group is at least two lanes
#endif

__device__ __forceinline__ float lane_pull(int src_lane, float v)
{
    return __int_as_float(__builtin_amdgcn_ds_bpermute(src_lane << 2, __float_as_int(v)));
}
__device__ __forceinline__ int lane_pull(int src_lane, int v) { return __builtin_amdgcn_ds_bpermute(src_lane << 2, v); }

// triangle_intersect without the store and without the `d > hit.t` early-out, for a ray held in plain values
// (the worker's copy of another lane's ray): the candidate (dist, u, w) and whether it passes every other early-out of
// fs:312-340 (a conjunction, so their order does not matter; NaN operands fail the same comparisons as upstream)
__device__ __forceinline__ bool triangle_candidate(V3 P, V3 D, float r0, float r1, const float4 q0, const float4 q1,
                                                   const float4 q2, float &dist, float &u, float &w)
{
    const V3 v0 = mk(q0.x, q0.y, q0.z), e0 = mk(q0.w, q1.x, q1.y), e1 = mk(q1.z, q1.w, q2.x);
    const V3 M = cross3(e1, D);
    const float det = dot3(e0, M);
    const float inv_det = reciprocal_of_determinant(det);
    const V3 T = P - v0;
    const V3 Q = cross3(T, e0);
    dist = -dot3(e1, Q) * inv_det;
    u = dot3(T, M) * inv_det;
    w = dot3(D, Q) * inv_det;
    if (det > -0.0000001f && det < 0.0000001f)
        return false;
    if (dist < r0 || dist > r1)
        return false;
    if (u < 0.0f || u > 1.0f)
        return false;
    if (w < 0.0f || u + w > 1.0f)
        return false;
    return true;
}

// The search of a dealt stage: `parked` = the lanes in LT_LEAF (K of them, K <= SHRAY_DEAL_MAX_PARKED).  Returns true if
// a worker accepted an unordered candidate (the caller then runs the plain loop); else the parked lane's winner in
// (won, wd, wu, ww), won = 0xffffffff for none.  `ids`: 64 bytes of LDS owned by this wave (rank of a parked lane -> its
// lane number)
// CACHED: the stage's first kCacheSlots distinct leaves come through the wave's leaf cache (`ids` is followed by it); the
// workers of a ray whose leaf got no slot fetch their triangles themselves, as before.
// ROOMY: the instance is compiled for six waves per SIMD (80 registers): its rounds run as one hand-scheduled statement
// (leaf_asm.h: dealt_rounds_scheduled).  Measured (profiles/EXPERIMENTS.md R6.3): a lone frame 0.356 -> 0.349 ms there, but the
// seven- and eight-wave instances lose 2 % with it (throughput form 11,951 -> 11,743 Mrays/s, config 4 2.18 -> 2.21 ms): they keep
// the compiler's rounds.
template <bool COUNT, bool BOUNDS, bool CACHED, bool ROOMY = false>
__device__ __forceinline__ bool dealt_search(const SceneView &sc, const LaneTraversal &t, int state, RayCounters &rc, uint8_t *ids,
                                             unsigned long long parked, int K, float &wd, float &wu, float &ww,
                                             uint32_t &won SHRAY_DIAG_PARAM)
{
    constexpr uint32_t kNoSlot = 0xffffffffu;
    char *cache = reinterpret_cast<char *>(ids) + kIdsBytes;
    uint32_t my_slot = kNoSlot;
    const bool cached = CACHED && t.leaf_cap <= kCacheTriangles;   // (uniform; a larger leaf cap: every group fetches for itself)
    if (cached) {
        bool todo = state == LT_LEAF, now;
        uint32_t at;
        leaf_cache_fill(sc, t, todo, now, at, cache);
        my_slot = now ? at : kNoSlot;
    }
    const int log_g = K <= 4 ? 4 : (K <= 8 ? 3 : (K <= 16 ? 2 : 1));   // G = 16, 8, 4, 2
    const int G = 1 << log_g;
    const int lane = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
    const int rank = (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(parked >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)parked, 0u));
    if (state == LT_LEAF)
        ids[rank] = (uint8_t)lane;
    const int group = lane >> log_g, sub = lane & (G - 1);
    const bool worker = group < K;
    const int src = worker ? (int)ids[group] : lane;    // same wave, LDS operations complete in order
    // the timed instances run their rounds as one hand-scheduled statement (leaf_asm.h: dealt_rounds_scheduled), which also
    // pulls the ray -- behind the first round's fetches, so that the pulls' LDS round trips and the fetch overlap
#if SHRAY_ASM_LEAF && !defined(SHRAY_DIAGNOSTICS)
    constexpr bool SCHEDULED = ROOMY && !COUNT && BOUNDS && !CACHED;
#else
    constexpr bool SCHEDULED = false;
#endif
    // the parked ray, as its workers see it
    V3 P = mk(0, 0, 0), D = mk(0, 0, 0);
    float r0 = 0.0f, r1 = 0.0f;
    if (!SCHEDULED) {
        P = mk(lane_pull(src, t.P.x), lane_pull(src, t.P.y), lane_pull(src, t.P.z));
        D = mk(lane_pull(src, t.D.x), lane_pull(src, t.D.y), lane_pull(src, t.D.z));
        r0 = lane_pull(src, t.leaf_r0);
        r1 = lane_pull(src, t.leaf_r1);
    }
    // worker i of a group walks triangles i, i + G, ... < count of its ray's leaf; the winner is kept as that number
    const uint32_t first = (uint32_t)lane_pull(src, (int)t.leaf_first);
    // (every pull is a statement of its own, executed by all 64 lanes: ds_bpermute returns 0 for a source lane that
    // is masked off, so a pull must never sit inside a conditional expression)
    const uint32_t count = parked_count((uint32_t)lane_pull(src, (int)t.leaf_count), t.leaf_cap);
    uint32_t end = worker ? count : 0u;
    asm volatile("" : "+v"(end));   // one comparison per round (see leaf_stage)
    // where the worker's next triangle is: a byte offset into packed_tris -- or, if the ray's leaf has a slot in the cache,
    // into the wave's slab (`served`)
    uint32_t where = __umul24(first, 36u) + __umul24((uint32_t)sub, 36u);
    bool served = false;
    if (CACHED) {
        const uint32_t slot = (uint32_t)lane_pull(src, (int)my_slot);
        served = slot != kNoSlot;
        if (served)
            where = slot + __umul24((uint32_t)sub, 36u);
        if (cached)
            leaf_cache_wait(where);
    }
    float best_d = 0.0f, best_u = 0.0f, best_w = 0.0f;
    if (!SCHEDULED)
        best_d = lane_pull(src, t.hit.t);
    uint32_t best = 0xffffffffu;    // no candidate accepted
    uint32_t unordered_flag = 0u;   // accepted a candidate whose d is NaN (see above); set by the tied form below
    SHRAY_DIAG_COUNT(6);
    if (SCHEDULED) {
        if (dealt_rounds_scheduled(sc, t, src << 2, end, (uint32_t)G, (uint32_t)sub, where, best_d, best_u, best_w, best)) {
            asm volatile("; unordered candidate: sequential leaf loop" ::: "memory");
            return true;
        }
    } else {
    // bottom-tested, like leaf_stage's loop: group 0's first worker always has a triangle
    uint32_t tri = (uint32_t)sub;
    do {
        SHRAY_DIAG_COUNT(1);
        if (tri < end) {
            float4 q0, q1, q2;
            if (CACHED && served)
                load_cached_triangle(cache + where, q0, q1, q2);
            else
                load_packed_triangle_at(sc, where, q0, q1, q2);
            if (COUNT)
                rc.triangle_tests++;
            float d, u, w;
            if (triangle_candidate(P, D, r0, r1, q0, q1, q2, d, u, w) && !(d > best_d)) {
                // the worker's best candidate so far is rewritten HERE only, deep inside the test's early-outs: as plain
                // assignments the four values (and the flag, a lane mask) are copied back and forth at every level of that
                // nest, ~24 moves and a dozen scalar mask operations per triangle; tied to their registers, none
                asm volatile("v_mov_b32 %0, %5\n\tv_mov_b32 %1, %6\n\tv_mov_b32 %2, %7\n\tv_mov_b32 %3, %8\n\t"
                             "v_cmp_u_f32 vcc, %5, %5\n\tv_cndmask_b32 %4, %4, 1, vcc"
                             : "+v"(best_d), "+v"(best_u), "+v"(best_w), "+v"(best), "+v"(unordered_flag)
                             : "v"(d), "v"(u), "v"(w), "v"(tri)
                             : "vcc");
                if (BOUNDS) {
                    // r0, r1 are bounds of the leaf's range (lane_visit_loaded): a candidate within 2^-19 of an end is for the
                    // exact range to decide -- the stage then runs the sequential loop, which does that (same flag)
                    float scaled;
                    asm volatile("v_mul_f32 %1, %4, %2\n\tv_cmp_lt_f32 vcc, %1, %3\n\tv_cndmask_b32 %0, %0, 1, vcc"
                                 : "+v"(unordered_flag), "=&v"(scaled) : "v"(d), "v"(r0), "s"(kCheckDown) : "vcc");
                    asm volatile("v_mul_f32 %1, %4, %2\n\tv_cmp_gt_f32 vcc, %1, %3\n\tv_cndmask_b32 %0, %0, 1, vcc"
                                 : "+v"(unordered_flag), "=&v"(scaled) : "v"(d), "v"(r1), "s"(kCheckUp) : "vcc");
                }
            }
        }
        tri += (uint32_t)G;
        where += (uint32_t)G * 36u;
    } while (wave_ballot(tri < end));
    if (__builtin_expect(wave_ballot(unordered_flag != 0u) != 0ull, 0)) {
        asm volatile("; unordered candidate: sequential leaf loop" ::: "memory");   // keeps this a branch
        return true;    // the parked rays have not been touched yet; the triangle tests were tallied above
    }
    }
    // combine inside each group: smaller d, of equal d the later triangle (no candidate = 0xffffffff loses)
    for (int step = 1; step < G; step <<= 1) {
        const int other = lane ^ step;
        const float od = lane_pull(other, best_d);
        const uint32_t ob = (uint32_t)lane_pull(other, (int)best);
        const bool take = ob != 0xffffffffu && (best == 0xffffffffu || od < best_d || (od == best_d && ob > best));
        const float ou = lane_pull(other, best_u), ow = lane_pull(other, best_w);
        best_d = take ? od : best_d;
        best_u = take ? ou : best_u;
        best_w = take ? ow : best_w;
        best = take ? ob : best;
    }
    // the parked lane collects its group's winner and moves on (fs:416-433)
    const int from = rank << log_g;
    wd = lane_pull(from, best_d);
    wu = lane_pull(from, best_u);
    ww = lane_pull(from, best_w);
    won = (uint32_t)lane_pull(from, (int)best);
    return false;
}

// One call site of the plain loop serves both the crowded stage (more than SHRAY_DEAL_MAX_PARKED lanes parked) and the
// unordered fallback in the timed instances, and one end (leaf_finish) serves every path: each inlined copy is another
// 150 instructions and another set of register copies where its results meet the other paths'.
// CACHE: `ids` is followed by the wave's leaf cache; the crowded stage's sequential loop reads its triangles from there
template <bool COUNT, int BLOCK, bool PAIR = false, bool CACHE = false, bool ROOMY = false>
__device__ __forceinline__ void leaf_stage_dealt(const SceneView &sc, const FrameView &fr, LaneTraversal &t, int &state,
                                                 uint32_t *stack, RayCounters &rc, uint8_t *ids SHRAY_DIAG_PARAM)
{
    constexpr bool BOUNDS = !PAIR;   // the parked leaf range is a pair of bounds (lane_visit_loaded)
    const unsigned long long parked = wave_ballot(state == LT_LEAF);
    if (!parked)
        return;
    const int K = __popcll(parked);
#if defined(SHRAY_DIAGNOSTICS) && defined(SHRAY_DIAG_KHIST)
#include "variants/diag_khist_stage.inc"
#endif
    float wd = 0.0f, wu = 0.0f, ww = 0.0f;
    uint32_t won = 0xffffffffu;
    bool plain = K > SHRAY_DEAL_MAX_PARKED, tallied = false;
    if (!plain) {
        plain = dealt_search<COUNT, BOUNDS, CACHE && SHRAY_LEAF_CACHE_DEALT != 0, ROOMY>(sc, t, state, rc, ids, parked, K, wd, wu, ww, won SHRAY_DIAG_ARG_FWD);
        tallied = true;
    }
    if (plain) {
        if (CACHE && t.leaf_cap <= kCacheTriangles) {     // (uniform)
            char *cache = reinterpret_cast<char *>(ids) + kIdsBytes;
            if (COUNT && !tallied)
                leaf_loop_cached<COUNT, BOUNDS>(sc, t, state, rc, cache SHRAY_DIAG_ARG_FWD);
            else
                leaf_loop_cached<false, BOUNDS>(sc, t, state, rc, cache SHRAY_DIAG_ARG_FWD);
        } else if (COUNT && !tallied)
            leaf_loop<COUNT, BOUNDS>(sc, t, state, rc SHRAY_DIAG_ARG_FWD);
        else
            leaf_loop_timed_or_counted<false, BOUNDS>(sc, t, state, rc SHRAY_DIAG_ARG_FWD);
    } else if (state == LT_LEAF && won != 0xffffffffu) {
        // the parked lane takes its group's winner (its number in the leaf)
        t.hit.which = (float)(t.leaf_first + won);
        t.hit.t = wd;
        t.hit.bu = wu;
        t.hit.bv = ww;
    }
    leaf_finish<COUNT, BLOCK, PAIR>(sc, t, state, stack, rc);
}

}   // namespace shray

#include "leaf_asm.h"
