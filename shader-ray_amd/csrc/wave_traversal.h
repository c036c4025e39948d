// wave_traversal.h -- the BVH walk as two wave-cooperative stages over per-lane state.
//
// A lane's traversal is a strictly sequential program (raytracer.es.fs:386-443): visit a
// node, if it is a leaf whose box is hit test its triangles in order, then follow the hit /
// miss link.  Inside a wave, lanes reach their leaves at different times; if each lane runs
// its <= 10 dependent triangle tests the moment it arrives, the other 63 lanes wait for it
// on almost every iteration (measured: the frame lasted as long as its most divergent wave,
// profiles/r01).  Here every lane keeps its own order, but the WAVE decides which kind of
// step to run next:
//
//   inner_stage   lanes in WALK visit one node each (slab test, push far child / pop, the
//                 iteration-cap bookkeeping); a lane whose leaf box is hit parks in LEAF
//   leaf_stage    all parked lanes run their triangle loops together, then move on; or
//   leaf_stage_dealt  the parked lanes' triangles are dealt to the wave's idle lanes (round 2)
//
// The node loop keeps running while enough lanes are still walking and hands over to the leaf
// stage once enough lanes are parked, so triangle tests run with many lanes active instead of one
// or two.  Used by kernel_stack.hip (a wave keeps its rays) and kernel_pool.hip (the waves of a
// workgroup merge their rays between epochs).
//
// Arithmetic, visit order and iteration counting are exactly those of stack_traversal.h's
// first version and of the literal threaded kernel; tests require bit-identical frames and
// equal work counters across all of them.
//
// Where things are (round 6: this file was 1,200 lines, two fifths of them variants that are off):
//   wave_traversal.h   (this file) a lane's traversal state, ONE NODE VISIT as the compiler writes it (lane_visit_loaded: the
//                      counting twins, and the rare turns the scheduled stage hands back), the node stage inner_stage
//   visit_asm.h        the node stage of the timed instances, hand-scheduled: what the shipped kernels run
//   leaf_stage.h       triangle_intersect, the sequential leaf loop, the dealt leaf stage (compiler's form)
//   leaf_asm.h         the leaf loops of the timed instances, hand-scheduled: what the shipped kernels run
//   leaf_cache.h       north_star's LDS staging of leaf triangles (libshray_hip_leafcache.so; measured slower, R5.1)
//   variants/          what was built, measured and is off: pair_traversal.h (kernel id 3), packed_slab.h (-DSHRAY_PK_SLAB),
//                      parked_state.h (-DSHRAY_PARK), diagnostics.h + *.inc (-DSHRAY_DIAGNOSTICS)
// The shipped instances' ISA is the same instruction for instruction before and after the split (profiles/r06/prune_isa_identity.txt).
#pragma once

#include "exact_div.h"
#include "packed_layout.h"
#include "trace_common.h"

namespace shray {

// LT_RETEST (pair traversal only): the lane holds a node taken off its stack whose own record must be fetched and
// tested before it is entered (a leaf, or a branch whose stored entry distance cannot decide r0 < hit.t); like
// LT_LEAF it is "parked" -- served by the next leaf stage.  Parked states are even.
enum : int { LT_WALK = 1, LT_LEAF = 2, LT_ENDED = 3, LT_RETEST = 4 };

// Variants that were built, measured and removed (LDS staging of the top of the tree, MUBUF loads, a DPP combine in the
// dealt stage, untied register moves, ...): profiles/EXPERIMENTS.md, with their numbers.
// node visits per lane between two evaluations of inner_stage's exit tests (fewer instructions against more
// registers; three measured best in round 2, four since the visit got shorter in round 4: a lone frame 0.468 -> 0.456 ms,
// the throughput form +0.2 %, profiles/r04/node_turns_ab.txt)
#ifndef SHRAY_NODE_TURNS
#define SHRAY_NODE_TURNS 4
#endif

// (the diagnostic build threads wave-level tallies through the stages: variants/diagnostics.h; empty here)
#ifdef SHRAY_DIAGNOSTICS
#include "variants/diagnostics.h"
#else
#define SHRAY_DIAG_DECL
#define SHRAY_DIAG_T0
#define SHRAY_DIAG_WAIT(k) ((void)0)
#define SHRAY_DIAG_COUNT(k) ((void)0)
#define SHRAY_DIAG_PARAM
#define SHRAY_DIAG_ARG
#define SHRAY_DIAG_ARG_FWD
#endif

struct LaneTraversal {
    V3 P, D, Y;               // object-space ray and its reciprocal direction RN(1/D) (exact_div.h); the residual of the
                              // reciprocal is recomputed where the exact quotients are needed (slab_range: a rare branch)
    bool fx, fy, fz, divide;  // direction signs; divide = operands outside exact_div.h's ranges
    unsigned long long divide_mask;   // the wave's lanes with `divide` set (uniform; kept beside the per-lane flag because a
                              // ballot of a flag that lives in a lane mask is materialised in a vector register first)
    uint32_t positive_dir;    // bit k: D[k] > 0 (the pair traversal's form)
    uint32_t octant;          // byte offset of the copy of the node array this ray's visits read: the one whose records hold the
                              // planes a ray of its direction octant enters and leaves a box by, and the children in the
                              // order it visits them (packed_layout.h)
    Hit hit;
    uint32_t node;
    uint32_t *top;            // LDS: the next free slot of this ray's stack column (slots are BLOCK words apart)
    int left;                 // node visits left before the iteration cap (fs:426-438), counted down
    uint32_t leaf_cap;        // fr.max_leaf_tests, held in a scalar register (not re-read from the view at every visit)
    float leaf_r0, leaf_r1;
    uint32_t leaf_first, leaf_count;
};

__device__ __forceinline__ unsigned long long wave_ballot(bool p) { return __builtin_amdgcn_ballot_w64(p); }

// The stack position moves by one slot.  Written as a plain `top += BLOCK` in one branch of the visit it becomes a
// loop-carried phi that the compiler resolves with a register copy on each of the OTHER branches and one more where
// they meet (three moves per node turn in the ISA); an add tied to its own register keeps it in place on every path.
typedef __attribute__((address_space(3))) uint32_t lds_word;
template <int BYTES>
__device__ __forceinline__ void top_move(uint32_t *&top)
{
    lds_word *p = (lds_word *)top;
    asm volatile("v_add_u32_e32 %0, %1, %0" : "+v"(p) : "i"((unsigned int)BYTES));
    top = (uint32_t *)p;
}

// The copy of the node array a ray reads (packed_layout.h: one per direction octant; bit k of the octant = D[k] >= 0, the
// predicate range_intersect_box selects a box's entry plane by, fs:204-213), as a byte offset; and the address of a visit's
// record in it.  A ray names a node by its byte offset / 8 with the node's split axis in the three bits above (the word the
// parent's record holds): the address is ONE instruction, (node << 3) + octant -- the shift drops the axis bits.
__device__ __forceinline__ uint32_t octant_offset(const SceneView &sc, bool fx, bool fy, bool fz)
{
    return ((fx ? 1u : 0u) | (fy ? 2u : 0u) | (fz ? 4u : 0u)) * sc.packed_nodes_bytes;
}
__device__ __forceinline__ uint32_t node_address(const LaneTraversal &t, uint32_t node) { return (node << kNodeNameShift) + t.octant; }

// group_intersect set-up for the object-space ray (P, D)                      (fs:388-392, :486)
template <bool COUNT>
__device__ __forceinline__ void lane_begin(const SceneView &sc, const FrameView &fr, LaneTraversal &t, uint32_t *stack, V3 P, V3 D,
                                           RayCounters &rc, bool counted = true)
{
    t.P = P;
    t.D = D;
    t.divide = !(sc.exact_div_ok && divisor_in_range(D.x) && divisor_in_range(D.y) && divisor_in_range(D.z) &&
                 coordinate_in_range(P.x) && coordinate_in_range(P.y) && coordinate_in_range(P.z));
    // RN(1 / D): a direction inside exact_div.h's divisor range is inside the three-instruction reciprocal's domain too;
    // anything else divides (and its lane will go on dividing: t.divide)
    t.Y = mk(reciprocal_in_range(D.x), reciprocal_in_range(D.y), reciprocal_in_range(D.z));
#ifndef SHRAY_COST_MAIN_PATH
    if (__builtin_expect(wave_ballot(t.divide) != 0ull, 0)) {
        asm volatile("; a lane that divides" ::: "memory");   // keeps this a branch the wave skips
        if (t.divide)
            t.Y = mk(1.0f / D.x, 1.0f / D.y, 1.0f / D.z);
    }
#endif
    t.divide_mask = wave_ballot(t.divide);
    t.fx = D.x >= 0.0f;
    t.fy = D.y >= 0.0f;
    t.fz = D.z >= 0.0f;
    t.positive_dir = (D.x > 0.0f ? 1u : 0u) | (D.y > 0.0f ? 2u : 0u) | (D.z > 0.0f ? 4u : 0u);
    t.octant = octant_offset(sc, t.fx, t.fy, t.fz);
    t.hit = Hit{kFar, -1.0f, 0.0f, 0.0f};
    t.node = sc.packed_root;
    t.top = stack;
    t.left = fr.max_bvh_iterations > 0 ? fr.max_bvh_iterations : 0x7fffffff;   // counted down to the cap; never zero without one
    t.leaf_cap = (uint32_t)fr.max_leaf_tests;
    asm volatile("" : "+s"(t.leaf_cap));
    if (COUNT && counted)
        rc.traversals++;
}

// A node visit (with its triangles, if any) is over: follow the link.  Returns the lane's next state.
template <int BLOCK>
__device__ __forceinline__ int lane_advance(LaneTraversal &t, uint32_t *stack, bool descended, uint32_t near_child)
{
    if (descended) {
        t.node = near_child;
    } else if (t.top == stack) {
        return LT_ENDED;        // finished: `left` is not counted down (the cap does not apply to a finished ray)
    } else {
        top_move<-4 * BLOCK>(t.top);
        t.node = *t.top;
    }
    return LT_WALK;             // (the visit this leads to is counted where it is made: lane_count_visit)
}

// The cap of fs:426-438: a ray that has used max_bvh_iterations visits and is not finished becomes a bad hit (set_bad_hit).
// A visit is counted where it is made, by ONE subtraction in front of the node's fetch (counted where the previous visit
// moved on, the subtraction sat on both of that visit's paths -- descend, pop -- and a wave whose lanes take both issued it
// twice): `left` goes below zero on the visit that is one too many.
//   * The counting twins test there and do not make that visit: their tallies are the reference's.
//   * The timed instances test once per SHRAY_NODE_TURNS visits (inner_stage: lane_apply_cap): a lane that is past its last
//     visit walks on for up to SHRAY_NODE_TURNS more -- visits change neither its hit nor anything another lane sees, its
//     stack stays within the tree's depth --; whatever state that leaves it in (walking, parked in a leaf whose triangles
//     are then never tested, finished), it ends as the bad hit it became.  A comparison per visit is 1.7 % of the headline
//     (profiles/r04/cap_check_ab.txt).
__device__ __forceinline__ void lane_count_visit(LaneTraversal &t) { t.left--; }

__device__ __forceinline__ void lane_apply_cap(LaneTraversal &t, int &state)
{
    if (__builtin_expect(__builtin_amdgcn_sicmp(t.left, 0, 40 /* slt */) != 0ull, 0)) {
        asm volatile("; iteration cap" ::: "memory");   // keeps this a branch (the compiler would predicate three moves into every visit)
        if (t.left < 0) {
            t.hit.t = -1.0f;
            state = LT_ENDED;
        }
    }
}

// A packed node's two 16-byte words.  `at` is the record's byte offset in the eight copies (node_address): the address is
// the (scalar) base plus that 32-bit offset, which the load instruction takes as is (SGPR base + VGPR offset) -- no 64-bit
// add.  (shray_scene_create admits at most 2^21 nodes and 2^24 vertices -- the shader's float32 indices -- so the eight
// copies end below 2^29 bytes and triangle byte offsets stay far below 2^32.)
// (the record lies in memory as DeviceNode, packed_layout.h: { entry.x, entry.y, exit.x, exit.y } { entry.z, exit.z, a, b }; it
// comes back as lo = { entry planes, a }, hi = { exit planes, b } -- names for registers, no instruction)
__device__ __forceinline__ void load_packed_node(const SceneView &sc, uint32_t at, float4 &lo, float4 &hi)
{
    const float4 *p = reinterpret_cast<const float4 *>(reinterpret_cast<const char *>(sc.packed_nodes) + at);
    const float4 q0 = p[0], q1 = p[1];
    lo = make_float4(q0.x, q0.y, q1.x, q1.z);
    hi = make_float4(q0.z, q0.w, q1.y, q1.w);
}

// The same for the lanes of a wave that are executing, when they are all at ONE record (one node, one octant: the top of the tree, coherent primary
// rays: 51 % of the headline's wave-visits, profiles/r04/distinct_nodes_per_wave_visit.txt).  A CU's vector memory
// pipeline -- one per CU, shared by its four SIMDs -- spends about 14 cycles on a 16-byte-per-lane instruction whatever its
// lanes read, even with one lane active (profiles/r04/vector_cache_probe.json), and that pipeline is what bounds the node
// loop; a node every lane wants is fetched once, through the scalar cache, and copied into the
// lanes' registers (eight moves).  +3.1 % on the headline, +2.5 ... +3.6 % on configs 3-5 (profiles/r04/scalar_nodes_ab.txt).
// Serving SOME lanes that way and the others with the vector load -- one, two, three or four groups, one after the other --
// puts two dependent fetches into one visit and is 15-30 % slower (same file): only the all-at-one-node case is taken.
__device__ __forceinline__ void load_packed_node_shared(const SceneView &sc, uint32_t node, float4 &lo, float4 &hi)
{
    const uint32_t first = (uint32_t)__builtin_amdgcn_readfirstlane((int)node);
    if (wave_ballot(node != first) == 0ull) {
        typedef unsigned long long q4 __attribute__((ext_vector_type(4)));
        typedef __attribute__((address_space(4))) const q4 constant_q4;
        const q4 v = *reinterpret_cast<constant_q4 *>(reinterpret_cast<uintptr_t>(sc.packed_nodes) + (uintptr_t)first);
        // into the lanes' registers two words at a time (v_mov_b64: four moves instead of the eight the compiler makes of it)
        unsigned long long w0, w1, w2, w3;
        asm("v_mov_b64 %0, %1" : "=v"(w0) : "s"(v[0]));
        asm("v_mov_b64 %0, %1" : "=v"(w1) : "s"(v[1]));
        asm("v_mov_b64 %0, %1" : "=v"(w2) : "s"(v[2]));
        asm("v_mov_b64 %0, %1" : "=v"(w3) : "s"(v[3]));
        // (DeviceNode's words: w0 = entry.xy, w1 = exit.xy, w2 = { entry.z, exit.z }, w3 = { a, b })
        lo = make_float4(__uint_as_float((uint32_t)w0), __uint_as_float((uint32_t)(w0 >> 32)), __uint_as_float((uint32_t)w2),
                         __uint_as_float((uint32_t)w3));
        hi = make_float4(__uint_as_float((uint32_t)w1), __uint_as_float((uint32_t)(w1 >> 32)), __uint_as_float((uint32_t)(w2 >> 32)),
                         __uint_as_float((uint32_t)(w3 >> 32)));
    } else {
        load_packed_node(sc, node, lo, hi);
    }
}

// range_intersect_box of the node's box against [0, 1e8] (fs:200-217, :272-275): the entry plane is the box's low
// side when D >= 0, else its high side.  The arithmetic the shader demands of a node visit: 6 selects, 6 subtractions,
// 6 quotients, 6 min / max (profiles/isa_costs.py counts it in isolation).
// PRESELECTED: (lo, hi) is a record of the ray's octant copy -- its entry planes, its exit planes: the selects were made when
// the scene was created.  (The pair traversal's records hold the scene's boxes as they are: it selects.)
template <bool PRESELECTED = true>
__device__ __forceinline__ void slab_range(const LaneTraversal &t, const float4 lo, const float4 hi, float &r0, float &r1)
{
    const float ex = ((PRESELECTED || t.fx) ? lo.x : hi.x) - t.P.x, ey = ((PRESELECTED || t.fy) ? lo.y : hi.y) - t.P.y,
                ez = ((PRESELECTED || t.fz) ? lo.z : hi.z) - t.P.z;
    const float xx = ((PRESELECTED || t.fx) ? hi.x : lo.x) - t.P.x, xy = ((PRESELECTED || t.fy) ? hi.y : lo.y) - t.P.y,
                xz = ((PRESELECTED || t.fz) ? hi.z : lo.z) - t.P.z;
    // all six quotients are finite on this path, so hardware min/max equal GLSL's select forms
    // (the residuals of the reciprocals are wanted HERE only, in a branch the wave rarely takes; computed from values the
    // optimizer can see are loop-invariant they are hoisted in front of the node loop and spilled there -- three scratch
    // stores per traversal on the pipe that bounds the kernel --, so the reciprocals pass through an opaque copy first)
    V3 Y = t.Y;
    asm volatile("" : "+v"(Y.x), "+v"(Y.y), "+v"(Y.z));
    const V3 YL = mk(reciprocal_residual(t.D.x, Y.x), reciprocal_residual(t.D.y, Y.y), reciprocal_residual(t.D.z, Y.z));
    r0 = fmaxf(fmaxf(fmaxf(0.0f, div_by_constant4(ex, t.D.x, Y.x, YL.x)), div_by_constant4(ey, t.D.y, Y.y, YL.y)),
               div_by_constant4(ez, t.D.z, Y.z, YL.z));
    r1 = fminf(fminf(fminf(kRangeMax, div_by_constant4(xx, t.D.x, Y.x, YL.x)), div_by_constant4(xy, t.D.y, Y.y, YL.y)),
               div_by_constant4(xz, t.D.z, Y.z, YL.z));
    if (t.divide) {   // operands outside the proven ranges of exact_div.h: true division, GLSL min/max
        r0 = sel_max(sel_max(sel_max(0.0f, ex / t.D.x), ey / t.D.y), ez / t.D.z);
        r1 = sel_min(sel_min(sel_min(kRangeMax, xx / t.D.x), xy / t.D.y), xz / t.D.z);
    }
}

// The visit's decision without the exact quotients (round 4).  fs:400 asks `!(r0 >= r1) && r0 < hit.t` of the correctly
// rounded quotients; a quotient's one-multiplication approximation q~ = RN(a * RN(1 / b)) differs from RN(a / b) by less
// than 3 * 2^-24 of it and has its sign, so r0~ = max(0, q~...) and r1~ = min(1e8, q~...) are within 2^-22 (relative) of
// r0 and r1, and
//      r0~ (1 + 2^-20) <  min(r1~, hit.t)   =>  the box is entered       (r0 <= r0~ (1 + 2^-22) < r1~ (1 - 2^-22) <= r1, < hit.t)
//      r0~ (1 - 2^-20) >= min(r1~, hit.t)   =>  it is not                (r0 >= r0~ (1 - 2^-22) >= r1 or >= hit.t; r1~ <= 0: r1 <= 0 <= r0)
// decide all but about one visit in 10^5 exactly as the quotients would: 22 + 5 vector instructions instead of 40.  A lane
// that neither test decides -- or whose operands are outside exact_div.h's ranges (t.divide: also a lane whose hit.t is NaN,
// leaf_finish) -- evaluates slab_range in a branch the wave skips.  What a lane parks for its leaf's triangle tests is the
// pair of bounds lo0 = r0~ (1 - 2^-20) <= r0 and hi1 = r1~ (1 + 2^-20) >= r1 (or r0, r1 themselves from that branch):
// a candidate outside [lo0, hi1] is outside [r0, r1]; one inside that is about to be ACCEPTED is looked at again
// (leaf_range_check: within 2^-19 of an end the leaf's exact range is computed and decides, fs:327-331).
// Frames and work counters stay bit-identical to the reference's divisions (the GPU parity suite).
constexpr float kBandUp = 1.0f + 0x1p-20f, kBandDown = 1.0f - 0x1p-20f;       // node test, parked bounds
constexpr float kCheckUp = 1.0f + 0x1p-19f, kCheckDown = 1.0f - 0x1p-19f;     // leaf_range_check

// (r1 comes back WITHOUT its clamp to kRangeMax: the visit folds the clamp into its three-operand minimum with hit.t, a leaf
// that is entered applies it to what it parks)
// (lo = the entry planes, hi = the exit planes of the ray's octant: no select)
// (-DSHRAY_PK_SLAB=1, variants/packed_slab.h: the same twelve operations as six packed instructions; bit-identical, slower, R5.7)
#ifndef SHRAY_PK_SLAB
#define SHRAY_PK_SLAB 0
#endif
#if SHRAY_PK_SLAB
#include "variants/packed_slab.h"
#else
__device__ __forceinline__ void slab_range_fast(const LaneTraversal &t, const float4 lo, const float4 hi, float &r0, float &r1)
{
    const float ex = lo.x - t.P.x, ey = lo.y - t.P.y, ez = lo.z - t.P.z;
    const float xx = hi.x - t.P.x, xy = hi.y - t.P.y, xz = hi.z - t.P.z;
    r0 = fmaxf(fmaxf(fmaxf(0.0f, ex * t.Y.x), ey * t.Y.y), ez * t.Y.z);
    r1 = fminf(fminf(xx * t.Y.x, xy * t.Y.y), xz * t.Y.z);
}
#endif

// The leaf's exact clipped range (fs:406: the range the leaf's box test left), from the leaf's own record: t.node is
// still the leaf while a lane is parked (not in the pair traversal, which parks exact ranges and never asks).
__device__ __forceinline__ void exact_leaf_range(const SceneView &sc, const LaneTraversal &t, float &e0, float &e1)
{
    float4 lo, hi;
    load_packed_node(sc, node_address(t, t.node), lo, hi);
    slab_range(t, lo, hi, e0, e1);
}

// A candidate distance that passed `d < leaf_r0 || d > leaf_r1` against the parked bounds and is about to be accepted:
// true if it is so close to an end that the exact range could still reject it.
__device__ __forceinline__ bool near_range_end(float d, float lo0, float hi1) { return d * kCheckDown < lo0 || d * kCheckUp > hi1; }

// fs:400's decision for one visit; r0 = a lower bound of the box's entry distance, r1 = an approximation of its exit
// distance within 2^-22 (or both exact, from the branch): see above.
// (a, b: the record's two link words -- a branch's children in the order the octant visits them; a lane whose direction has a
// ZERO component corrects that order here, see below)
__device__ __forceinline__ bool visit_decision(const LaneTraversal &t, const float4 lo, const float4 hi, float &r0, float &r1,
                                               uint32_t &a, uint32_t &b)
{
    slab_range_fast(t, lo, hi, r0, r1);
    float below;   // min(r1~, 1e8, hit.t): one bare v_min3_f32 (a NaN hit.t makes its lane divide, so the branch below decides it)
    asm("v_min3_f32 %0, %1, %2, %3" : "=v"(below) : "v"(r1), "s"(kRangeMax), "v"(t.hit.t));
    const float lo0 = r0 * kBandDown;
    bool enter = r0 * kBandUp < below;
    const bool miss = lo0 >= below;
    r0 = lo0;
#ifndef SHRAY_COST_MAIN_PATH     // profiles/isa_costs.hip counts the path every wave takes
    // (the wave-level test from the two comparisons' own lane masks: as one ballot of the combined condition the compiler
    // materialises it in a vector register and compares again -- three more vector instructions per visit)
    const unsigned long long active = wave_ballot(true);
    const unsigned long long undecided = (active & ~(wave_ballot(enter) | wave_ballot(miss))) | (active & t.divide_mask);
    if (__builtin_expect(undecided != 0ull, 0)) {
        asm volatile("; a visit the bounds do not decide: the exact quotients" ::: "memory");   // keeps this a branch
        const bool unsure = !(enter || miss) || t.divide;
        if (unsure) {
            slab_range(t, lo, hi, r0, r1);
            enter = !(r0 >= r1) && (r0 < t.hit.t);
        }
        // The children's order.  The reference descends into the negative child first when the direction component along
        // the split axis is > 0 (world.cpp:259-265, :214-220), and takes a box's low plane as its entry plane when it is
        // >= 0 (fs:204-213): one octant copy serves both except for a component that IS zero (+0 or -0), whose copy holds
        // the order of a positive one.  Such a lane divides (zero is outside exact_div.h's range) and so is always here:
        // it swaps the words back.
        if (t.divide && !(b & kLeafFlag)) {
            const uint32_t zero_axes = (t.D.x == 0.0f ? 1u << kAxisHotShift : 0u) | (t.D.y == 0.0f ? 2u << kAxisHotShift : 0u) |
                                       (t.D.z == 0.0f ? 4u << kAxisHotShift : 0u);
            if (a & zero_axes) {
                const uint32_t first = a & kChildNameMask;
                a = (a & ~kChildNameMask) | b;
                b = first;
            }
        }
    }
#endif
    return enter;
}

// One node visit for a lane in LT_WALK, given the node's two 16-byte words; returns its next state.
// (A predicated form of the bookkeeping below -- every side effect once, under its own condition: 75 instead of
// 96 vector instructions per visit in the ISA -- was measured in rounds 1 and 2 and is 3-8 % SLOWER on every
// workload, profiles/history/r02/leaf_stage_ab.txt; the branches let the wave skip whole blocks with s_cbranch_execz.)
template <bool COUNT, int BLOCK>
__device__ __forceinline__ int lane_visit_loaded(const FrameView &fr, LaneTraversal &t, uint32_t *stack, RayCounters &rc,
                                                 const float4 lo, const float4 hi)
{
    if (COUNT)
        rc.node_visits++;
    uint32_t a = __float_as_uint(lo.w), b = __float_as_uint(hi.w);
    if (COUNT && (b & kLeafFlag))
        rc.leaf_visits++;   // the reference fetches (start, count) before the box test, fs:263-267

    float r0, r1;
    const bool enter = visit_decision(t, lo, hi, r0, r1, a, b);
    if (enter) {
        if (b & kLeafFlag) {
            // (the leaf's upper bound; a lane that took the exact branch parks a bound 2^-20 above its r1: still a bound;
            // the clamp slab_range_fast left to its caller: the exact branch's r1 has it already)
            // (a bare v_min_f32: fminf() canonicalises its operand first, one more instruction)
            asm("v_min_f32 %0, %1, %0" : "+v"(r1) : "s"(kRangeMax));
            r1 = r1 * kBandUp;
            // the leaf's count word is parked as it is (flag bit and all); the leaf stages clamp it to the leaf cap
            // once per stage (parked_count) instead of every visit masking, clamping and testing it
            const uint32_t count = b;
            if (b != kLeafFlag && t.leaf_cap != 0u) {
                // the parked leaf's four fields are written HERE only.  As plain assignments they become loop-carried
                // phis that the compiler resolves with eight register copies on the inner-node path (the common one);
                // a move whose destination is tied to the old value keeps each field in one register on every path.
                asm volatile("v_mov_b32 %0, %4\n\tv_mov_b32 %1, %5\n\tv_mov_b32 %2, %6\n\tv_mov_b32 %3, %7"
                             : "+v"(t.leaf_first), "+v"(t.leaf_count), "+v"(t.leaf_r0), "+v"(t.leaf_r1)
                             : "v"(a), "v"(count), "v"(r0), "v"(r1));
                return LT_LEAF;
            }
            return lane_advance<BLOCK>(t, stack, false, 0u);
        }
        // a = the child this ray's octant visits first (under the split axis' bit, which node_address() shifts out), b = the
        // other one: nothing to decide
        *t.top = b;
        top_move<4 * BLOCK>(t.top);
        return lane_advance<BLOCK>(t, stack, true, a);
    }
    return lane_advance<BLOCK>(t, stack, false, 0u);
}

// Node loop: lanes whose state is LT_WALK visit nodes until fewer than `keep_walking` of
// them remain while other lanes are parked (state == LT_LEAF) or `others_waiting`.
// The iteration cap: the counting instances (COUNT: the reference's tallies AND the timed form's tallying twins, TALLY == 1)
// apply it in front of every visit; the timed instances once per SHRAY_NODE_TURNS visits, so that a capped ray of theirs makes
// up to SHRAY_NODE_TURNS - 1 node visits the tallying twin does not count (a capped ray's frame is the same marker either way;
// bench.py's instruction model is off by at most 3 visits per capped ray: 1.9e-5 of config 4's samples, none of the headline's).
template <bool COUNT, int BLOCK>
__device__ __forceinline__ void inner_stage(const SceneView &sc, const FrameView &fr, LaneTraversal &t, int &state,
                                            uint32_t *stack, RayCounters &rc, int keep_walking, bool others_waiting SHRAY_DIAG_PARAM)
{
    for (;;) {
        if (!wave_ballot(state == LT_WALK))
            return;
#pragma unroll
        for (int turn = 0; turn < SHRAY_NODE_TURNS; turn++) {   // the exit tests below run once per SHRAY_NODE_TURNS visits
            SHRAY_DIAG_COUNT(0);
            if (state == LT_WALK) {
                lane_count_visit(t);
                if (COUNT)
                    lane_apply_cap(t, state);
                if (!COUNT || state == LT_WALK) {
                    SHRAY_DIAG_T0
                    float4 lo, hi;
                    load_packed_node_shared(sc, node_address(t, t.node), lo, hi);
                    SHRAY_DIAG_WAIT(4);
#if defined(SHRAY_DIAGNOSTICS) && defined(SHRAY_DIAG_UNIFORM)
#include "variants/diag_uniform_visit.inc"
#endif
                    state = lane_visit_loaded<COUNT, BLOCK>(fr, t, stack, rc, lo, hi);
                }
            }
        }
        if (!COUNT)
            lane_apply_cap(t, state);
        const int walking = __popcll(wave_ballot(state == LT_WALK));
        if (walking < keep_walking && (wave_ballot(state == LT_LEAF) || others_waiting))
            return;
    }
}

}   // namespace shray

#include "leaf_stage.h"
