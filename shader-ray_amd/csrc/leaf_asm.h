// leaf_asm.h -- the sequential leaf loop of the timed instances, hand-scheduled for gfx950 (round 6).
//
// leaf_stage.h: leaf_loop is the same loop as the compiler writes it: per round ~80 vector and ~32 scalar instructions -- five
// nested early-outs of triangle_intersect (fs:312-340), each an exec save / and / branch / restore ladder around a block that, in
// a wave with fifty testing lanes, is almost never skipped.  Here a round is straight-line code: every early-out is ONE v_cmpx
// that takes the failing lanes out of EXEC (their later arithmetic is simply not executed for them; nothing is saved, EXEC is
// set again from a scalar copy when the round is over), the loop's own test is one v_cmpx per round instead of two compares,
// and the stores happen under whatever EXEC is left.  Same operations on the same operands in the same order as
// triangle_distance / triangle_barycentrics (leaf_stage.h): bit-identical hits.  68 vector (+ 3 fetches) and 8 scalar
// instructions per round.
// What is rare leaves the statement BEFORE anything is stored, and the round is made by the compiler's form (which holds the
// true division and the exact leaf range): a determinant outside the three-instruction reciprocal's domain (>= 2^100, or NaN),
// and a candidate about to be accepted within 2^-19 of an end of its leaf's parked bounds (near_range_end).
#pragma once

#include "wave_traversal.h"

namespace shray {

// leaf_loop<false, true> (leaf_stage.h): every lane in LT_LEAF tests its leaf's triangles in order.  Some lane is parked.
// Registers of the statement: v[2:5] v[6:9] v10 the triangle { v0.xyz, e0.x } { e0.yz, e1.xy } { e1.z }; v11-v13 M = e1 x D;
// v14, v15 products; v16 det, then the distance; v17 1 / det; v18-v20 T = P - v0; v2-v4 again: Q = T x e0; v14 u, v15 w.
__device__ __forceinline__ void leaf_loop_scheduled(const SceneView &sc, LaneTraversal &t, int state, RayCounters &rc)
{
    uint32_t mine = state == LT_LEAF ? parked_count(t.leaf_count, t.leaf_cap) : 0u;
    uint32_t where = __umul24(t.leaf_first, 36u);   // the lane's next triangle: a byte offset into packed_tris
    uint32_t j = 0;
    for (;;) {
        uint32_t reason, j_next;
        unsigned long long saved, other;
        asm volatile(
            "s_mov_b64 %[saved], exec\n\t"
            "v_readfirstlane_b32 %[J], %[J0]\n\t"         // (the round counter: uniform, handed over in a lane register)
            "s_nop 1\n\t"                                  // (a VALU-written SGPR read by a VALU: two wait states)
            "v_cmpx_lt_u32_e32 %[J], %[MINE]\n\t"
            "s_cbranch_execz ldone_%=\n"
            "lround_%=:\n\t"
            "global_load_dwordx4 v[2:5], %[W], %[base]\n\t"
            "global_load_dwordx4 v[6:9], %[W], %[base] offset:16\n\t"
            "global_load_dword v10, %[W], %[base] offset:32\n\t"
            "s_waitcnt vmcnt(0)\n\t"
            // M = cross(e1, D)                                                                       (fs:307)
            "v_mul_f32_e32 v14, %[Dz], v9\n\t"
            "v_mul_f32_e32 v15, %[Dy], v10\n\t"
            "v_sub_f32_e32 v11, v14, v15\n\t"
            "v_mul_f32_e32 v14, %[Dx], v10\n\t"
            "v_mul_f32_e32 v15, %[Dz], v8\n\t"
            "v_sub_f32_e32 v12, v14, v15\n\t"
            "v_mul_f32_e32 v14, %[Dy], v8\n\t"
            "v_mul_f32_e32 v15, %[Dx], v9\n\t"
            "v_sub_f32_e32 v13, v14, v15\n\t"
            // det = dot(e0, M); |det| < 1e-7: no hit                                                  (fs:309-313)
            "v_mul_f32_e32 v14, v5, v11\n\t"
            "v_mul_f32_e32 v15, v6, v12\n\t"
            "v_add_f32_e32 v14, v14, v15\n\t"
            "v_mul_f32_e32 v15, v7, v13\n\t"
            "v_add_f32_e32 v16, v15, v14\n\t"
            "v_cmpx_nlt_f32_e64 vcc, |v16|, %[eps]\n\t"
            // 1 / det: v_rcp_f32 and one Newton step (exact_div.h: reciprocal_in_range); outside its domain: the compiler's round
            "v_rcp_f32_e32 v17, v16\n\t"
            "v_cmp_nlt_f32_e64 vcc, |v16|, %[big]\n\t"
            "v_fma_f32 v14, -v16, v17, 1.0\n\t"
            "v_fmac_f32_e32 v17, v14, v17\n\t"
            "s_cbranch_vccnz lslow_%=\n\t"
            // T = P - v0, Q = cross(T, e0), d = -dot(e1, Q) / det                                     (fs:315-325)
            "v_sub_f32_e32 v18, %[Px], v2\n\t"
            "v_sub_f32_e32 v19, %[Py], v3\n\t"
            "v_sub_f32_e32 v20, %[Pz], v4\n\t"
            "v_mul_f32_e32 v14, v19, v7\n\t"
            "v_mul_f32_e32 v15, v20, v6\n\t"
            "v_sub_f32_e32 v2, v14, v15\n\t"
            "v_mul_f32_e32 v14, v20, v5\n\t"
            "v_mul_f32_e32 v15, v18, v7\n\t"
            "v_sub_f32_e32 v3, v14, v15\n\t"
            "v_mul_f32_e32 v14, v18, v6\n\t"
            "v_mul_f32_e32 v15, v19, v5\n\t"
            "v_sub_f32_e32 v4, v14, v15\n\t"
            "v_mul_f32_e32 v14, v8, v2\n\t"
            "v_mul_f32_e32 v15, v9, v3\n\t"
            "v_add_f32_e32 v14, v14, v15\n\t"
            "v_mul_f32_e32 v15, v10, v4\n\t"
            "v_add_f32_e32 v14, v15, v14\n\t"
            "v_mul_f32_e64 v16, v17, -v14\n\t"
            // d > hit.t || d outside the leaf's (parked) range: no hit                                 (fs:327-331)
            "v_min_f32_e32 v14, %[HT], %[LR1]\n\t"
            "v_cmpx_nlt_f32_e32 v16, %[LR0]\n\t"
            "v_cmpx_ngt_f32_e32 v16, v14\n\t"
            "s_cbranch_execz lnext_%=\n\t"
            // u = dot(T, M) / det in [0, 1]                                                            (fs:333-336)
            "v_mul_f32_e32 v14, v18, v11\n\t"
            "v_mul_f32_e32 v15, v19, v12\n\t"
            "v_add_f32_e32 v14, v14, v15\n\t"
            "v_mul_f32_e32 v15, v20, v13\n\t"
            "v_add_f32_e32 v14, v15, v14\n\t"
            "v_mul_f32_e32 v14, v14, v17\n\t"
            "v_cmpx_ngt_f32_e32 0, v14\n\t"
            "v_cmpx_nlt_f32_e32 1.0, v14\n\t"
            // w = dot(D, Q) / det >= 0, u + w <= 1                                                     (fs:337-340)
            "v_mul_f32_e32 v15, %[Dx], v2\n\t"
            "v_mul_f32_e32 v11, %[Dy], v3\n\t"
            "v_add_f32_e32 v15, v15, v11\n\t"
            "v_mul_f32_e32 v11, %[Dz], v4\n\t"
            "v_add_f32_e32 v15, v11, v15\n\t"
            "v_mul_f32_e32 v15, v15, v17\n\t"
            "v_add_f32_e32 v11, v14, v15\n\t"
            "v_cmpx_ngt_f32_e32 0, v15\n\t"
            "v_cmpx_nlt_f32_e32 1.0, v11\n\t"
            "s_cbranch_execz lnext_%=\n\t"
            // a candidate within 2^-19 of an end of the parked bounds: the exact range decides (near_range_end): the compiler's round
            "v_mul_f32_e32 v11, 0x3f7fffe0, v16\n\t"
            "v_mul_f32_e32 v12, 0x3f800010, v16\n\t"
            "v_cmp_lt_f32_e32 vcc, v11, %[LR0]\n\t"
            "v_cmp_gt_f32_e64 %[other], v12, %[LR1]\n\t"
            "s_or_b64 vcc, vcc, %[other]\n\t"
            "s_cbranch_vccnz lslow_%=\n\t"
            // the hit                                                                                  (fs:342-345)
            "v_add_u32_e32 v11, %[J], %[LF]\n\t"
            "v_cvt_f32_u32_e32 %[HW], v11\n\t"
            "v_mov_b32_e32 %[HT], v16\n\t"
            "v_mov_b32_e32 %[HU], v14\n\t"
            "v_mov_b32_e32 %[HV], v15\n"
            "lnext_%=:\n\t"
            "s_mov_b64 exec, %[saved]\n\t"
            "s_add_u32 %[J], %[J], 1\n\t"
            "v_add_u32_e32 %[W], 36, %[W]\n\t"
            "v_cmpx_lt_u32_e32 %[J], %[MINE]\n\t"
            "s_cbranch_execnz lround_%=\n"
            "ldone_%=:\n\t"
            "s_mov_b32 %[reason], 0\n\t"
            "s_branch lout_%=\n"
            "lslow_%=:\n\t"
            "s_mov_b32 %[reason], 1\n"
            "lout_%=:\n\t"
            "s_mov_b64 exec, %[saved]\n\t"
            : [HT] "+v"(t.hit.t), [HW] "+v"(t.hit.which), [HU] "+v"(t.hit.bu), [HV] "+v"(t.hit.bv), [W] "+v"(where), [J] "=&s"(j_next),
              [saved] "=&s"(saved), [other] "=&s"(other), [reason] "=&s"(reason)
            : [Px] "v"(t.P.x), [Py] "v"(t.P.y), [Pz] "v"(t.P.z), [Dx] "v"(t.D.x), [Dy] "v"(t.D.y), [Dz] "v"(t.D.z), [LR0] "v"(t.leaf_r0),
              [LR1] "v"(t.leaf_r1), [LF] "v"(t.leaf_first), [MINE] "v"(mine), [base] "s"(sc.packed_tris), [eps] "s"(0.0000001f),
              [big] "s"(0x1p100f), [J0] "v"(j)
            : "vcc", "scc", "memory", "v2", "v3", "v4", "v5", "v6", "v7", "v8", "v9", "v10", "v11", "v12", "v13", "v14", "v15", "v16",
              "v17", "v18", "v19", "v20");
        // (an asm statement's results count as divergent whatever register class they are in: say they are not)
        if (__builtin_amdgcn_readfirstlane((int)reason) == 0)
            return;
        j = (uint32_t)__builtin_amdgcn_readfirstlane((int)j_next);
        asm volatile("; a round the scheduled leaf loop leaves to the compiler's test" ::: "memory");
        if (j < mine) {
            float4 q0, q1, q2;
            load_packed_triangle(sc, t.leaf_first + j, q0, q1, q2);
            lane_test_triangle_loaded<false, true>(sc, t, t.leaf_first + j, rc, q0, q1, q2);
        }
        j++;
        where += 36u;
        if (!wave_ballot(j < mine))
            return;
    }
}

// The rounds of a dealt leaf stage (leaf_stage.h: dealt_search) the same way: worker lane `sub` of a group tests triangles
// sub, sub + G, ... < end of its ray's leaf -- the ray (P, D), its parked bounds (r0, r1) and the hit distance it starts from
// (best_d) pulled from the parked lane -- and keeps, over its own increasing triangle numbers, the candidate the sequential loop
// would keep: every early-out of triangle_candidate is one v_cmpx, then `!(d > best_d)`.  Returns true where dealt_search's flag
// would be raised -- a candidate about to be kept whose distance is NaN or within 2^-19 of an end of the parked bounds -- and for
// a determinant outside the reciprocal's domain: the stage then discards what the workers found and runs the sequential loop
// over the untouched parked rays (which holds the true division and the exact range).  60 vector (+ 3 fetches) and 7 scalar
// instructions per round.
// (registers as above; v14 u, v15 w)
// `source`: four times the number of the parked lane this worker serves (ds_bpermute's address; the lane's own for a lane that
// serves nobody).  The statement pulls that lane's ray, parked bounds and hit distance itself, all 64 lanes executing, and waits
// for them together with the first round's fetches.
__device__ __forceinline__ bool dealt_rounds_scheduled(const SceneView &sc, const LaneTraversal &t, int source, uint32_t end, uint32_t G,
                                                       uint32_t tri, uint32_t where, float &best_d, float &best_u, float &best_w, uint32_t &best)
{
    uint32_t flagged;
    unsigned long long saved, other;
    const uint32_t stride = G * 36u;
    float px, py, pz, dx, dy, dz, r0, r1;
    asm volatile(
        "s_mov_b64 %[saved], exec\n\t"
        "ds_bpermute_b32 %[Dx], %[SRC], %[tDx]\n\t"
        "ds_bpermute_b32 %[Dy], %[SRC], %[tDy]\n\t"
        "ds_bpermute_b32 %[Dz], %[SRC], %[tDz]\n\t"
        "ds_bpermute_b32 %[Px], %[SRC], %[tPx]\n\t"
        "ds_bpermute_b32 %[Py], %[SRC], %[tPy]\n\t"
        "ds_bpermute_b32 %[Pz], %[SRC], %[tPz]\n\t"
        "ds_bpermute_b32 %[R0], %[SRC], %[tR0]\n\t"
        "ds_bpermute_b32 %[R1], %[SRC], %[tR1]\n\t"
        "ds_bpermute_b32 %[BD], %[SRC], %[tHT]\n\t"
        "v_cmpx_lt_u32_e32 %[TRI], %[END]\n\t"
        "s_cbranch_execz ddone_%=\n"
        "dround_%=:\n\t"
        "global_load_dwordx4 v[2:5], %[W], %[base]\n\t"
        "global_load_dwordx4 v[6:9], %[W], %[base] offset:16\n\t"
        "global_load_dword v10, %[W], %[base] offset:32\n\t"
        "s_waitcnt vmcnt(0) lgkmcnt(0)\n\t"
        "v_mul_f32_e32 v14, %[Dz], v9\n\t"
        "v_mul_f32_e32 v15, %[Dy], v10\n\t"
        "v_sub_f32_e32 v11, v14, v15\n\t"
        "v_mul_f32_e32 v14, %[Dx], v10\n\t"
        "v_mul_f32_e32 v15, %[Dz], v8\n\t"
        "v_sub_f32_e32 v12, v14, v15\n\t"
        "v_mul_f32_e32 v14, %[Dy], v8\n\t"
        "v_mul_f32_e32 v15, %[Dx], v9\n\t"
        "v_sub_f32_e32 v13, v14, v15\n\t"
        "v_mul_f32_e32 v14, v5, v11\n\t"
        "v_mul_f32_e32 v15, v6, v12\n\t"
        "v_add_f32_e32 v14, v14, v15\n\t"
        "v_mul_f32_e32 v15, v7, v13\n\t"
        "v_add_f32_e32 v16, v15, v14\n\t"
        "v_cmpx_nlt_f32_e64 vcc, |v16|, %[eps]\n\t"
        "v_rcp_f32_e32 v17, v16\n\t"
        "v_cmp_nlt_f32_e64 vcc, |v16|, %[big]\n\t"
        "v_fma_f32 v14, -v16, v17, 1.0\n\t"
        "v_fmac_f32_e32 v17, v14, v17\n\t"
        "s_cbranch_vccnz dflag_%=\n\t"
        "v_sub_f32_e32 v18, %[Px], v2\n\t"
        "v_sub_f32_e32 v19, %[Py], v3\n\t"
        "v_sub_f32_e32 v20, %[Pz], v4\n\t"
        "v_mul_f32_e32 v14, v19, v7\n\t"
        "v_mul_f32_e32 v15, v20, v6\n\t"
        "v_sub_f32_e32 v2, v14, v15\n\t"
        "v_mul_f32_e32 v14, v20, v5\n\t"
        "v_mul_f32_e32 v15, v18, v7\n\t"
        "v_sub_f32_e32 v3, v14, v15\n\t"
        "v_mul_f32_e32 v14, v18, v6\n\t"
        "v_mul_f32_e32 v15, v19, v5\n\t"
        "v_sub_f32_e32 v4, v14, v15\n\t"
        "v_mul_f32_e32 v14, v8, v2\n\t"
        "v_mul_f32_e32 v15, v9, v3\n\t"
        "v_add_f32_e32 v14, v14, v15\n\t"
        "v_mul_f32_e32 v15, v10, v4\n\t"
        "v_add_f32_e32 v14, v15, v14\n\t"
        "v_mul_f32_e64 v16, v17, -v14\n\t"
        // d outside the parked bounds: no candidate; d > the best so far: not kept            (triangle_candidate; dealt_search)
        "v_cmpx_nlt_f32_e32 v16, %[R0]\n\t"
        "v_cmpx_ngt_f32_e32 v16, %[R1]\n\t"
        "v_cmpx_ngt_f32_e32 v16, %[BD]\n\t"
        "s_cbranch_execz dnext_%=\n\t"
        "v_mul_f32_e32 v14, v18, v11\n\t"
        "v_mul_f32_e32 v15, v19, v12\n\t"
        "v_add_f32_e32 v14, v14, v15\n\t"
        "v_mul_f32_e32 v15, v20, v13\n\t"
        "v_add_f32_e32 v14, v15, v14\n\t"
        "v_mul_f32_e32 v14, v14, v17\n\t"
        "v_cmpx_ngt_f32_e32 0, v14\n\t"
        "v_cmpx_nlt_f32_e32 1.0, v14\n\t"
        "v_mul_f32_e32 v15, %[Dx], v2\n\t"
        "v_mul_f32_e32 v11, %[Dy], v3\n\t"
        "v_add_f32_e32 v15, v15, v11\n\t"
        "v_mul_f32_e32 v11, %[Dz], v4\n\t"
        "v_add_f32_e32 v15, v11, v15\n\t"
        "v_mul_f32_e32 v15, v15, v17\n\t"
        "v_add_f32_e32 v11, v14, v15\n\t"
        "v_cmpx_ngt_f32_e32 0, v15\n\t"
        "v_cmpx_nlt_f32_e32 1.0, v11\n\t"
        "s_cbranch_execz dnext_%=\n\t"
        // kept -- unless its distance is unordered, or at an end of the bounds: the sequential loop decides
        "v_mul_f32_e32 v11, 0x3f7fffe0, v16\n\t"
        "v_mul_f32_e32 v12, 0x3f800010, v16\n\t"
        "v_cmp_lt_f32_e32 vcc, v11, %[R0]\n\t"
        "v_cmp_gt_f32_e64 %[other], v12, %[R1]\n\t"
        "s_or_b64 %[other], vcc, %[other]\n\t"
        "v_cmp_u_f32_e32 vcc, v16, v16\n\t"
        "s_or_b64 vcc, vcc, %[other]\n\t"
        "s_cbranch_vccnz dflag_%=\n\t"
        "v_mov_b32_e32 %[BD], v16\n\t"
        "v_mov_b32_e32 %[BU], v14\n\t"
        "v_mov_b32_e32 %[BW], v15\n\t"
        "v_mov_b32_e32 %[BEST], %[TRI]\n"
        "dnext_%=:\n\t"
        "s_mov_b64 exec, %[saved]\n\t"
        "v_add_u32_e32 %[TRI], %[G], %[TRI]\n\t"
        "v_add_u32_e32 %[W], %[stride], %[W]\n\t"
        "v_cmpx_lt_u32_e32 %[TRI], %[END]\n\t"
        "s_cbranch_execnz dround_%=\n"
        "ddone_%=:\n\t"
        "s_mov_b32 %[flagged], 0\n\t"
        "s_branch dout_%=\n"
        "dflag_%=:\n\t"
        "s_mov_b32 %[flagged], 1\n"
        "dout_%=:\n\t"
        "s_mov_b64 exec, %[saved]\n\t"
        "s_waitcnt lgkmcnt(0)\n\t"
        : [BD] "=&v"(best_d), [BU] "+v"(best_u), [BW] "+v"(best_w), [BEST] "+v"(best), [W] "+v"(where), [TRI] "+v"(tri),
          [Px] "=&v"(px), [Py] "=&v"(py), [Pz] "=&v"(pz), [Dx] "=&v"(dx), [Dy] "=&v"(dy), [Dz] "=&v"(dz), [R0] "=&v"(r0), [R1] "=&v"(r1),
          [saved] "=&s"(saved), [other] "=&s"(other), [flagged] "=&s"(flagged)
        : [tPx] "v"(t.P.x), [tPy] "v"(t.P.y), [tPz] "v"(t.P.z), [tDx] "v"(t.D.x), [tDy] "v"(t.D.y), [tDz] "v"(t.D.z), [tR0] "v"(t.leaf_r0),
          [tR1] "v"(t.leaf_r1), [tHT] "v"(t.hit.t), [SRC] "v"(source), [END] "v"(end),
          [base] "s"(sc.packed_tris), [eps] "s"(0.0000001f), [big] "s"(0x1p100f), [G] "s"(G), [stride] "s"(stride)
        : "vcc", "scc", "memory", "v2", "v3", "v4", "v5", "v6", "v7", "v8", "v9", "v10", "v11", "v12", "v13", "v14", "v15", "v16", "v17",
          "v18", "v19", "v20");
    return __builtin_amdgcn_readfirstlane((int)flagged) != 0;
}

}   // namespace shray
