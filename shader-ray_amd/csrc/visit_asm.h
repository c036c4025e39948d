// visit_asm.h -- the node stage of the timed instances, hand-scheduled for gfx950 (round 6).
//
// wave_traversal.h: inner_stage is the same stage as the compiler writes it: ~38 vector and ~35 scalar instructions per turn when
// a wave's lanes take every path of a visit (the structurizer's exec save / or / andn2 ladders around five nested branches, a
// state word moved on every path, 64-bit moves of a scalar-fetched record into lanes' registers), and a wave alone issues one
// instruction per ~4 cycles: a visit's instruction count IS its latency, and the kernel is bound by issue slots (DESIGN.md
// section 5).  Here the stage is ONE asm statement:
//   * who walks, who has parked, who has ended are lane MASKS in scalar registers (the state word is read once on entry and
//     written once on exit); EXEC is set by one s_mov / s_andn2 / v_cmpx per path, never saved and restored;
//   * fs:400's decision `!(r0 >= r1) && r0 < hit.t` from the same one-multiplication bounds as visit_decision, but compared as
//     INTEGERS: r0~ >= +0 (a signed-integer maximum with 0 clamps negative products and -0 to +0), below = min(r1~, hit.t);
//     d = int(below) - int(r0~) with saturation.  Floats of one sign order like their bit patterns and neighbours differ by a
//     relative 2^-24 ... 2^-23, so d > 16 means below > r0~ (1 + 2^-20): entered; d < -16 means r0~ > below (1 + 2^-20), or below is
//     negative (any negative float is an integer below -2^23) or -0: not entered -- the bounds of visit_decision's proof (the
//     quotient approximations are within 2^-22 of the quotients and have their signs; products are zero or normal numbers: exact_div.h's
//     ranges).  One subtraction and two comparisons instead of two multiplications, a three-operand minimum and two
//     comparisons; the parked leaf's bounds (r0~ (1 - 2^-20), min(r1~, 1e8) (1 + 2^-20)) are formed on the leaf path only.
//     Undecided lanes (about one visit in 10^5), lanes that divide, and empty leaves leave the statement: the turn is made by the
//     compiler's form of the visit (lane_visit_loaded), which holds the exact quotients;
//   * a record every walking lane is at comes through the scalar cache as before, and its planes are the subtractions' scalar
//     operands (no moves into lanes' registers; four v_mov_b64 and the vector path's own subtractions measure 1 % slower, and
//     without the scalar path at all the throughput form loses 10 %: profiles/EXPERIMENTS.md R6.2);
//   * the pop is `top != stack` as v_cmpx (the lanes whose stacks are empty drop out of EXEC and of the walking mask in that one
//     instruction), the push a write under the branch lanes' mask.
// Per turn with lanes on both the descend and the pop path: 31 vector (2 of them fetches, 2 LDS) and
// 16 scalar instructions (3 of them waits / a two-state nop), + 8 / 4 when a lane parks.
// Same visits, same order, same counts (`left`): frames and tallies bit-identical (the GPU parity suite runs through this stage).
#pragma once

#include "wave_traversal.h"

#ifndef SHRAY_ASM_VISIT
#define SHRAY_ASM_VISIT 1
#endif

namespace shray {

// One turn: every lane in EXEC (= the walking mask, not empty) visits its node.  Registers: v[2:5] v[6:9] the record
// { entry.x, entry.y, exit.x, exit.y } { entry.z, exit.z, a, b } (DeviceNode, packed_layout.h), then in place the six differences,
// the six products; v2 = r0~, v3 = below, then d; v4 = r1~ on the leaf path.  s[64:71]: a record fetched through the scalar cache.
// (-DSHRAY_VISIT_SCALAR=0, an A/B build: every record through the vector memory pipeline -- the throughput form loses 10 %, R6.2)
#ifndef SHRAY_VISIT_SCALAR
#define SHRAY_VISIT_SCALAR 1
#endif
// (-DSHRAY_VISIT_RUNS=0, an A/B build: no uniform runs -- every turn finds out again whether its lanes are at one record, R6.8)
#ifndef SHRAY_VISIT_RUNS
#define SHRAY_VISIT_RUNS 1
#endif

// the six products, the decision (see the file's header), and the way out for a turn the statement does not make
#define SHRAY_VISIT_DECIDE                                                                                                       \
    "v_mul_f32_e32 v2, v2, %[Yx]\n\t"                                                                                           \
    "v_mul_f32_e32 v3, v3, %[Yy]\n\t"                                                                                           \
    "v_mul_f32_e32 v6, v6, %[Yz]\n\t"                                                                                           \
    "v_mul_f32_e32 v4, v4, %[Yx]\n\t"                                                                                           \
    "v_mul_f32_e32 v5, v5, %[Yy]\n\t"                                                                                           \
    "v_mul_f32_e32 v7, v7, %[Yz]\n\t"                                                                                           \
    "v_max3_f32 v2, v2, v3, v6\n\t"                                                                                             \
    "v_max_i32_e32 v2, 0, v2\n\t"                           /* r0~ = max(0, ...), and never -0 */                                 \
    "v_min3_f32 v3, v4, v5, %[HT]\n\t"                      /* below = min(r1~, hit.t) in two instructions: hit.t takes the */    \
    "v_min_f32_e32 v3, v3, v7\n\t"                          /* place of the third product (r1~ itself: the leaf path)       */    \
    "v_sub_i32 v3, v3, v2 clamp\n\t"                        /* d */                                                               \
    "v_cmp_gt_i32_e64 %[sE], v3, 16\n\t"                    /* entered */                                                         \
    "v_cmp_gt_i32_e32 vcc, -16, v3\n\t"                     /* not entered */                                                     \
    "s_or_b64 %[sT], %[sE], vcc\n\t"                                                                                            \
    "s_andn2_b64 %[sT], %[sT], %[sDIV]\n\t"                 /* decided, and not a lane that divides */                            \
    "s_andn2_b64 %[sT], exec, %[sT]\n\t"                                                                                        \
    "s_cbranch_scc1 vslow_%=\n\t"                           /* some lane is not: the compiler's visit makes this turn */

// A turn whose walking lanes are all at ONE record (60 % of the headline's wave-visits; of those 56 % end with every lane entering,
// profiles/r06/uniform_wave_visits_by_outcome.txt).  The record comes through the scalar cache and its planes are the subtractions'
// scalar operands.  UNIFORM RUNS (round 6, R6.8): if the record is a branch's and EVERY walking lane enters it, every lane goes to
// the same child and the next turn's lanes are at one record again -- its address is ((child's name) << 3) + the run's octant
// offset, two scalar instructions -- so the next turn skips finding out (v_lshl_add, v_readfirstlane, v_cmp: three second-class
// vector instructions and a two-state nop), and this turn's tail is the push alone: no leaf test, no pop, no lane masks.
#if SHRAY_VISIT_SCALAR && SHRAY_VISIT_RUNS
#define SHRAY_VISIT_HEAD(K)                                                                                                      \
    "s_cmp_lg_u32 %[sKU], 0\n\t"                            /* inside a uniform run: the record's address is in sF already */     \
    "s_cbranch_scc1 vk" #K "_%=\n\t"                                                                                            \
    "s_waitcnt lgkmcnt(0)\n\t"                              /* the node the last turn took off the stack */                       \
    "v_lshl_add_u32 %[A], %[N], 3, %[OCT]\n\t"              /* node_address(): (name << 3) + octant */                            \
    "v_add_u32_e32 %[L], -1, %[L]\n\t"                      /* lane_count_visit (and the wait state in front of readfirstlane) */ \
    "v_readfirstlane_b32 %[sF], %[A]\n\t"                                                                                       \
    "s_nop 1\n\t"                                           /* a VALU-written SGPR read by a VALU: two wait states */             \
    "v_cmp_ne_u32_e32 vcc, %[sF], %[A]\n\t"                                                                                     \
    "s_cbranch_vccnz vv" #K "_%=\n\t"                                                                                           \
    "v_readfirstlane_b32 %[sOCT], %[OCT]\n\t"               /* one record: one octant copy -- its offset, should a run begin */  \
    "s_branch vu" #K "_%=\n"                                                                                                    \
    "vk" #K "_%=:\n\t"                                                                                                          \
    "v_add_u32_e32 %[L], -1, %[L]\n"                                                                                            \
    "vu" #K "_%=:\n\t"
#define SHRAY_VISIT_UNIFORM_TAIL(K)                                                                                              \
    "s_mov_b32 %[sKU], 0\n\t"                                                                                                   \
    SHRAY_VISIT_DECIDE                                                                                                           \
    "s_cmp_eq_u64 %[sE], exec\n\t"                          /* every walking lane enters ... */                                   \
    "s_cbranch_scc0 vm" #K "_%=\n\t"                                                                                            \
    "s_bitcmp1_b32 s71, 31\n\t"                             /* ... a branch's record (b without the leaf flag) */                 \
    "s_cbranch_scc1 vm" #K "_%=\n\t"                                                                                            \
    "v_mov_b32_e32 v9, s71\n\t"                             /* the run goes on: push the other child, go to the first */          \
    "ds_write_b32 %[T], v9\n\t"                                                                                                 \
    "v_add_u32_e32 %[T], %[up], %[T]\n\t"                                                                                       \
    "v_mov_b32_e32 %[N], s70\n\t"                                                                                               \
    "s_lshl_b32 %[sF], s70, 3\n\t"                          /* (the shift drops the axis bits of the child's name) */             \
    "s_add_u32 %[sF], %[sF], %[sOCT]\n\t"                                                                                       \
    "s_mov_b32 %[sKU], 1\n\t"                                                                                                   \
    "s_branch ve" #K "_%=\n"                                                                                                    \
    "vm" #K "_%=:\n\t"                                                                                                          \
    "v_mov_b32_e32 v8, s70\n\t"                                                                                                 \
    "v_mov_b32_e32 v9, s71\n\t"                                                                                                 \
    "s_branch vt" #K "_%=\n"
#elif SHRAY_VISIT_SCALAR
#define SHRAY_VISIT_HEAD(K)                                                                                                      \
    "s_waitcnt lgkmcnt(0)\n\t"                              /* the node the last turn took off the stack */                       \
    "v_lshl_add_u32 %[A], %[N], 3, %[OCT]\n\t"              /* node_address(): (name << 3) + octant */                            \
    "v_add_u32_e32 %[L], -1, %[L]\n\t"                      /* lane_count_visit (and the wait state in front of readfirstlane) */ \
    "v_readfirstlane_b32 %[sF], %[A]\n\t"                                                                                       \
    "s_nop 1\n\t"                                           /* a VALU-written SGPR read by a VALU: two wait states */             \
    "v_cmp_ne_u32_e32 vcc, %[sF], %[A]\n\t"                                                                                     \
    "s_cbranch_vccnz vv" #K "_%=\n\t"
#define SHRAY_VISIT_UNIFORM_TAIL(K)                                                                                              \
    "v_mov_b32_e32 v8, s70\n\t"                                                                                                 \
    "v_mov_b32_e32 v9, s71\n\t"                                                                                                 \
    "s_branch vj" #K "_%=\n"
#else
#define SHRAY_VISIT_HEAD(K)                                                                                                      \
    "s_waitcnt lgkmcnt(0)\n\t"                                                                                                  \
    "v_lshl_add_u32 %[A], %[N], 3, %[OCT]\n\t"                                                                                  \
    "v_add_u32_e32 %[L], -1, %[L]\n\t"                                                                                          \
    "s_branch vv" #K "_%=\n\t"
#define SHRAY_VISIT_UNIFORM_TAIL(K)                                                                                              \
    "v_mov_b32_e32 v8, s70\n\t"                                                                                                 \
    "v_mov_b32_e32 v9, s71\n\t"                                                                                                 \
    "s_branch vj" #K "_%=\n"
#endif
#define SHRAY_VISIT_TURN(K)                                                                                                      \
    SHRAY_VISIT_HEAD(K)                                                                                                          \
    "s_load_dwordx8 s[64:71], %[base], %[sF]\n\t"           /* every lane at one record: once, through the scalar cache */       \
    "s_waitcnt lgkmcnt(0)\n\t"                                                                                                  \
    "v_sub_f32_e32 v2, s64, %[Px]\n\t"                                                                                          \
    "v_sub_f32_e32 v3, s65, %[Py]\n\t"                                                                                          \
    "v_sub_f32_e32 v4, s66, %[Px]\n\t"                                                                                          \
    "v_sub_f32_e32 v5, s67, %[Py]\n\t"                                                                                          \
    "v_sub_f32_e32 v6, s68, %[Pz]\n\t"                                                                                          \
    "v_sub_f32_e32 v7, s69, %[Pz]\n\t"                                                                                          \
    SHRAY_VISIT_UNIFORM_TAIL(K)                                                                                                  \
    "vv" #K "_%=:\n\t"                                                                                                          \
    "global_load_dwordx4 v[2:5], %[A], %[base]\n\t"                                                                             \
    "global_load_dwordx4 v[6:9], %[A], %[base] offset:16\n\t"                                                                   \
    "s_waitcnt vmcnt(1)\n\t"                                                                                                    \
    "v_sub_f32_e32 v2, v2, %[Px]\n\t"                                                                                           \
    "v_sub_f32_e32 v3, v3, %[Py]\n\t"                                                                                           \
    "v_sub_f32_e32 v4, v4, %[Px]\n\t"                                                                                           \
    "v_sub_f32_e32 v5, v5, %[Py]\n\t"                                                                                           \
    "s_waitcnt vmcnt(0)\n\t"                                                                                                    \
    "v_sub_f32_e32 v6, v6, %[Pz]\n\t"                                                                                           \
    "v_sub_f32_e32 v7, v7, %[Pz]\n"                                                                                             \
    "vj" #K "_%=:\n\t"                                                                                                          \
    SHRAY_VISIT_DECIDE                                                                                                           \
    "\nvt" #K "_%=:\n\t"                                                                                                        \
    "v_cmp_gt_i32_e32 vcc, 0, v9\n\t"                       /* b's flag: a leaf's record */                                       \
    "s_and_b64 %[sL], %[sE], vcc\n\t"                       /* lanes that enter a leaf */                                         \
    "s_cbranch_scc0 vb" #K "_%=\n\t"                                                                                            \
    "s_mov_b64 exec, %[sL]\n\t"                                                                                                 \
    "v_cmp_eq_u32_e32 vcc, 0x80000000, v9\n\t"              /* a leaf without triangles: the compiler's visit */                 \
    "s_cbranch_vccnz vslow_%=\n\t"                                                                                              \
    "v_min3_f32 v4, v4, v5, v7\n\t"                         /* r1~ */                                                             \
    "v_min_f32_e32 v4, 0x4cbebc20, v4\n\t"                  /* the range's end is at most 1e8 (fs:392) */                         \
    "v_mul_f32_e32 %[LR1], 0x3f800008, v4\n\t"              /* parked: an upper bound of r1, */                                   \
    "v_mul_f32_e32 %[LR0], 0x3f7ffff0, v2\n\t"              /* a lower bound of r0, */                                            \
    "v_mov_b32_e32 %[LF], v8\n\t"                           /* the first triangle, */                                             \
    "v_mov_b32_e32 %[LC], v9\n\t"                           /* the count word as it is (parked_count) */                          \
    "s_or_b64 %[sP], %[sP], %[sL]\n"                                                                                            \
    "vb" #K "_%=:\n\t"                                                                                                          \
    "s_andn2_b64 exec, %[sW], %[sE]\n\t"                    /* not entered: the next node comes off the stack ... */             \
    "s_cbranch_scc0 vp" #K "_%=\n\t"                        /* (nobody: no LDS read to wait for at the next turn) */             \
    "v_cmpx_ne_u32_e32 %[T], %[B]\n\t"                      /* ... unless it is empty: those lanes have ended */                  \
    "v_add_u32_e32 %[T], %[down], %[T]\n\t"                                                                                     \
    "ds_read_b32 %[N], %[T]\n"                                                                                                  \
    "vp" #K "_%=:\n\t"                                                                                                          \
    "s_andn2_b64 %[sT], %[sE], %[sL]\n\t"                   /* lanes that enter a branch */                                       \
    "s_or_b64 %[sW], %[sT], exec\n\t"                       /* still walking: they, and the lanes that took a node off */        \
    "s_mov_b64 exec, %[sT]\n\t"                             /* descend: push the other child, go to the first */                  \
    "ds_write_b32 %[T], v9\n\t"                                                                                                 \
    "v_add_u32_e32 %[T], %[up], %[T]\n\t"                                                                                       \
    "v_mov_b32_e32 %[N], v8\n\t"                                                                                                \
    "s_mov_b64 exec, %[sW]\n\t"                                                                                                 \
    "s_cbranch_execz vg_%=\n"                                                                                                   \
    "ve" #K "_%=:\n\t"

// what the statement reports: the stage is over; a turn for the compiler's visit (`left` already counted); a lane is past the cap
enum : uint32_t { VISIT_STAGE_OVER = 0, VISIT_SLOW_TURN = 1, VISIT_CAP = 2 };

// inner_stage<false, BLOCK> (wave_traversal.h) for the timed instances of the stack kernel: same arguments, same effect on
// (t, state), the same exit rule -- no walking lane left, or fewer than keep_walking walking while some lane is parked, evaluated once
// per SHRAY_NODE_TURNS turns, like the iteration cap (lane_apply_cap).
template <int BLOCK>
__device__ __forceinline__ void inner_stage_scheduled(const SceneView &sc, const FrameView &fr, LaneTraversal &t, int &state,
                                                      uint32_t *stack, RayCounters &rc, int keep_walking)
{
    static_assert(SHRAY_NODE_TURNS == 4, "the statement below is four turns per evaluation of the exit tests");
    // one turn as the compiler writes it (its visit holds the exact quotients, the zero-component child order, empty leaves and
    // a leaf cap of zero)
    auto compiled_turn = [&](bool count) {
        if (state == LT_WALK) {
            if (count)
                lane_count_visit(t);
            float4 lo, hi;
            load_packed_node_shared(sc, node_address(t, t.node), lo, hi);
            state = lane_visit_loaded<false, BLOCK>(fr, t, stack, rc, lo, hi);
        }
    };
    auto past_the_cap = [&]() {
        if (__builtin_expect(__builtin_amdgcn_sicmp(t.left, 0, 40 /* slt */) != 0ull, 0)) {
            asm volatile("; iteration cap" ::: "memory");
            if (t.left < 0) {
                t.hit.t = -1.0f;
                state = LT_ENDED;
                t.left = 0x7fffffff;   // (an ended lane's count means nothing any more; the statement's test sees every lane)
            }
        }
    };
    if (__builtin_expect(t.leaf_cap == 0u, 0)) {   // (uniform) no leaf is ever parked: the compiler's stage
        for (;;) {
            if (!wave_ballot(state == LT_WALK))
                return;
            compiled_turn(true);
            past_the_cap();
        }
    }
    lds_word *top = (lds_word *)t.top, *const base = (lds_word *)stack;
    for (;;) {
        uint32_t reason, first, address, in_a_run, run_octant;
        unsigned long long walking, parked, walked_in, entered, mask, leaves, saved;
        asm volatile(
            "s_mov_b64 %[saved], exec\n\t"
            "v_cmp_eq_u32_e64 %[sW], 1, %[ST]\n\t"             // LT_WALK
            "v_cmp_eq_u32_e64 %[sP], 2, %[ST]\n\t"             // LT_LEAF (parked by a turn the compiler's visit made)
            "s_mov_b64 %[sW0], %[sW]\n\t"
            "s_mov_b32 %[sKU], 0\n\t"                           // no uniform run yet
            "s_cmp_eq_u64 %[sW], 0\n\t"
            "s_cbranch_scc1 vdone_%=\n"
            "vloop_%=:\n\t"
            "s_mov_b64 exec, %[sW]\n\t"
            SHRAY_VISIT_TURN(0) SHRAY_VISIT_TURN(1) SHRAY_VISIT_TURN(2) SHRAY_VISIT_TURN(3)
            "\nvg_%=:\n\t"
            "s_mov_b64 exec, %[saved]\n\t"
            "v_cmp_gt_i32_e32 vcc, 0, %[L]\n\t"                 // fs:426-438, once per four visits (lane_apply_cap)
            "s_cbranch_vccnz vcap_%=\n\t"
            "s_cmp_eq_u64 %[sW], 0\n\t"
            "s_cbranch_scc1 vdone_%=\n\t"
            "s_bcnt1_i32_b64 %[reason], %[sW]\n\t"          // (sF may hold a uniform run's next address: not a scratch register)
            "s_cmp_ge_u32 %[reason], %[keep]\n\t"
            "s_cbranch_scc1 vloop_%=\n\t"
            "s_cmp_eq_u64 %[sP], 0\n\t"
            "s_cbranch_scc1 vloop_%=\n"
            "vdone_%=:\n\t"
            "s_mov_b32 %[reason], 0\n\t"
            "s_branch vout_%=\n"
            "vcap_%=:\n\t"
            "s_mov_b32 %[reason], 2\n\t"
            "s_branch vout_%=\n"
            "vslow_%=:\n\t"
            "s_mov_b32 %[reason], 1\n"
            "vout_%=:\n\t"
            "s_mov_b64 exec, %[saved]\n\t"
            "s_waitcnt lgkmcnt(0)\n\t"
            "s_or_b64 %[sT], %[sW], %[sP]\n\t"
            "s_andn2_b64 %[sT], %[sW0], %[sT]\n\t"              // walked in, neither walking nor parked now: ended
            "v_cndmask_b32_e64 %[ST], %[ST], 3, %[sT]\n\t"
            "v_cndmask_b32_e64 %[ST], %[ST], 2, %[sP]\n\t"
            : [N] "+v"(t.node), [T] "+v"(top), [L] "+v"(t.left), [ST] "+v"(state), [LF] "+v"(t.leaf_first), [LC] "+v"(t.leaf_count),
              [LR0] "+v"(t.leaf_r0), [LR1] "+v"(t.leaf_r1), [A] "=&v"(address), [sW] "=&s"(walking), [sP] "=&s"(parked),
              [sW0] "=&s"(walked_in), [sE] "=&s"(entered), [sT] "=&s"(mask), [sL] "=&s"(leaves), [saved] "=&s"(saved),
              [sF] "=&s"(first), [reason] "=&s"(reason), [sKU] "=&s"(in_a_run), [sOCT] "=&s"(run_octant)
            : [Px] "v"(t.P.x), [Py] "v"(t.P.y), [Pz] "v"(t.P.z), [Yx] "v"(t.Y.x), [Yy] "v"(t.Y.y), [Yz] "v"(t.Y.z), [OCT] "v"(t.octant),
              [HT] "v"(t.hit.t), [B] "v"(base), [base] "s"(sc.packed_nodes), [sDIV] "s"(t.divide_mask), [keep] "s"(keep_walking),
              [up] "i"((unsigned int)(4 * BLOCK)), [down] "i"((unsigned int)(-4 * BLOCK))
            : "vcc", "scc", "memory", "v2", "v3", "v4", "v5", "v6", "v7", "v8", "v9", "s64", "s65", "s66", "s67", "s68", "s69", "s70", "s71");
        t.top = (uint32_t *)top;
        reason = (uint32_t)__builtin_amdgcn_readfirstlane((int)reason);   // (an asm statement's results count as divergent: say it is not)
        if (reason == VISIT_STAGE_OVER)
            return;
        if (reason == VISIT_CAP) {
            past_the_cap();
            continue;
        }
        asm volatile("; a turn the scheduled stage leaves to the compiler's visit" ::: "memory");
        compiled_turn(false);
        past_the_cap();   // (the statement left in the middle of its four turns: the cap's test that ends them is made here)
        top = (lds_word *)t.top;
    }
}

}   // namespace shray
