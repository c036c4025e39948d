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

// Diagnostic build only: wave-level tallies {node-loop iterations, leaf-loop iterations,
// cycles in the node loop, cycles in the leaf loop}, read by profiles/timeline.py.

// SHRAY_DIAG_KHIST (with SHRAY_DIAGNOSTICS; profiles/leaf_stage_histogram.py): the eight tallies are instead a histogram of
// the dealt leaf stages by the number of parked lanes K -- bins K = 1, 2, 3-4, 5-8, 9-16, 17-32, > 32 (the plain loop) --,
// each word {stages, bits 0-23; rounds of three strided fetches the stage runs, bits 24-43; 16-byte-per-lane fetches a
// stage would run if every group fetched its leaf's bytes as consecutive chunks, bits 44-63}; nothing is timed.
#if defined(SHRAY_DIAGNOSTICS) && defined(SHRAY_DIAG_KHIST)
#ifndef SHRAY_DIAG_KHIST_FROM
#define SHRAY_DIAG_KHIST_FROM 32
#endif
#define SHRAY_DIAG_DECL unsigned long long diag_tally[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#define SHRAY_DIAG_T0
#define SHRAY_DIAG_WAIT(k) ((void)0)
#define SHRAY_DIAG_COUNT(k) ((void)0)
#define SHRAY_DIAG_PARAM , unsigned long long *diag_tally_ref
#define SHRAY_DIAG_ARG , diag_tally
#define SHRAY_DIAG_ARG_FWD , diag_tally_ref
#elif defined(SHRAY_DIAGNOSTICS)
#define SHRAY_DIAG_DECL unsigned long long diag_tally[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#define SHRAY_DIAG_T0 const unsigned long long diag_t0 = __builtin_amdgcn_s_memtime();
#define SHRAY_DIAG_WAIT(k) do { __builtin_amdgcn_s_waitcnt(0); diag_tally_ref[k] += __builtin_amdgcn_s_memtime() - diag_t0; } while (0)
#define SHRAY_DIAG_COUNT(k) (diag_tally_ref[k]++)
#define SHRAY_DIAG_PARAM , unsigned long long *diag_tally_ref
#define SHRAY_DIAG_ARG , diag_tally
#define SHRAY_DIAG_ARG_FWD , diag_tally_ref
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
// Round 5 (R5.7), built, bit-identical, SLOWER, off (-DSHRAY_PK_SLAB=1): the six subtractions and six multiplications as six PACKED
// fp32 instructions (v_pk_add_f32 with its second source negated, v_pk_mul_f32: IEEE per component, the same values), on register
// PAIRS -- { entry.x, entry.y }, { exit.x, exit.y }, { entry.z, exit.z } as the record's loads leave them (DeviceNode), { P.x, P.y },
// { Y.x, Y.y } and { P.z, Y.z } of the ray, the z pair serving both halves of its instructions through op_sel; no pairing moves in
// the ISA.  Six issue slots fewer per visit -- and the headline 1 % slower, configs 3 / 5 2.3 % (profiles/r05/packed_slab_ab.txt):
// a packed instruction occupies the VALU as long as the two it replaces (R4.16) and its results arrive later in the visit's
// dependent chain (subtract -> multiply -> max3 -> compare).
typedef float f32x2 __attribute__((ext_vector_type(2)));
#ifndef SHRAY_PK_SLAB
#define SHRAY_PK_SLAB 0
#endif
__device__ __forceinline__ void slab_range_fast(const LaneTraversal &t, const float4 lo, const float4 hi, float &r0, float &r1)
{
#if SHRAY_PK_SLAB
    f32x2 e = {lo.x, lo.y}, x = {hi.x, hi.y}, z = {lo.z, hi.z};
    const f32x2 pxy = {t.P.x, t.P.y}, yxy = {t.Y.x, t.Y.y}, pzyz = {t.P.z, t.Y.z};
    asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(e) : "v"(e), "v"(pxy));
    asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(x) : "v"(x), "v"(pxy));
    asm("v_pk_add_f32 %0, %1, %2 op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]" : "=v"(z) : "v"(z), "v"(pzyz));
    asm("v_pk_mul_f32 %0, %1, %2" : "=v"(e) : "v"(e), "v"(yxy));
    asm("v_pk_mul_f32 %0, %1, %2" : "=v"(x) : "v"(x), "v"(yxy));
    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1]" : "=v"(z) : "v"(z), "v"(pzyz));
    // (a bare v_max_f32: fmaxf() of an asm result is canonicalised first -- one more instruction)
    float first;
    asm("v_max_f32_e32 %0, 0, %1" : "=v"(first) : "v"(e.x));
    r0 = fmaxf(fmaxf(first, e.y), z.x);
    r1 = fminf(fminf(x.x, x.y), z.y);
#else
    const float ex = lo.x - t.P.x, ey = lo.y - t.P.y, ez = lo.z - t.P.z;
    const float xx = hi.x - t.P.x, xy = hi.y - t.P.y, xz = hi.z - t.P.z;
    r0 = fmaxf(fmaxf(fmaxf(0.0f, ex * t.Y.x), ey * t.Y.y), ez * t.Y.z);
    r1 = fminf(fminf(xx * t.Y.x, xy * t.Y.y), xz * t.Y.z);
#endif
}

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
#if defined(SHRAY_DIAGNOSTICS) && !defined(SHRAY_DIAG_KHIST)
    {   // how much a triangle-parallel leaf stage could save: stages, and 64-wide rounds over all parked triangles
        unsigned int total = (state == LT_LEAF) ? parked_count(t.leaf_count, t.leaf_cap) : 0u;
        for (int off = 32; off > 0; off >>= 1)
            total += __shfl_xor(total, off, 64);
        diag_tally_ref[6] += 1;
        diag_tally_ref[7] += (total + 63u) / 64u;
    }
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

// Leaf cache (round 5).  The lanes of a wave that are parked TOGETHER mostly sit in the same few leaves: the rays of an 8x8
// tile reach a leaf side by side.  Stages with more than 32 parked lanes, the plain loop's share of the throughput form:
// ONE distinct leaf in 51 % of them, two in 31 %, three in 11 %, at most four in 96 % (1M-triangle scene: 15 / 28 / 25 %,
// at most four in 84 %; profiles/r05/leaf_stage_histograms.txt) -- and those stages are 71 % of all triangle rounds.
// The plain loop fetched every lane's triangle in every round: three strided fetches (16 + 16 + 4 bytes per lane) per
// round on the CU's one vector memory pipeline, which charges an instruction by its width, not by what its lanes read
// (DESIGN.md section 5).  Here a stage first names the distinct leaves among its parked lanes (a scalar loop: the first
// parked lane's leaf, a ballot of the lanes in the same one, the next ...), and each distinct leaf -- 36 x count
// consecutive bytes of packed_tris -- is fetched ONCE, as consecutive 16-byte chunks by the wave's first lanes, straight
// into a slot of the wave's slab in LDS (global_load_lds_dwordx4: no registers, nothing waits until the slots are read).
// Every parked lane then runs its triangles in order, as before, reading them from its leaf's slot: a round is five LDS
// reads, no fetch.  The lanes of leaves beyond the slab's kCacheSlots fetch their own triangles, as before, in the same rounds.
// (A leaf has at most kCacheTriangles triangles when the frame's leaf cap is that low -- the shader's is 10 -- else the
// uncached loop runs.)
#ifndef SHRAY_LEAF_CACHE
#define SHRAY_LEAF_CACHE 0                // measured slower (profiles/EXPERIMENTS.md R5.1): built as a variant library only
#endif
#ifndef SHRAY_LEAF_CACHE_SLOTS
#define SHRAY_LEAF_CACHE_SLOTS 3
#endif
#ifndef SHRAY_LEAF_CACHE_DEALT
#define SHRAY_LEAF_CACHE_DEALT 1          // the dealt stage's workers read cached leaves too
#endif
constexpr int kCacheSlots = SHRAY_LEAF_CACHE_SLOTS;
constexpr uint32_t kCacheTriangles = 10;                         // 90 words = 23 chunks of 16 bytes
constexpr uint32_t kCacheSlotBytes = 400;                        // 368 used; 100 words: consecutive slots start 4 banks apart
constexpr uint32_t kCacheBytes = (uint32_t)kCacheSlots * kCacheSlotBytes;
constexpr uint32_t kIdsBytes = 64;                               // the wave's `ids` table in front of its slab
typedef __attribute__((address_space(1))) const void cache_global_ptr;
typedef __attribute__((address_space(3))) void cache_lds_ptr;

// One pass of the cache's fill: the first kCacheSlots distinct leaves among the lanes in `todo` (parked, not yet served) are
// fetched into the slots; `now` = the lanes they serve (taken out of `todo`), `at` = a served lane's slot as a byte offset
// into the slab.  Nothing waits: the fetches are in flight when this returns (leaf_cache_wait).
__device__ __forceinline__ void leaf_cache_fill(const SceneView &sc, const LaneTraversal &t, bool &todo, bool &now, uint32_t &at, char *cache)
{
    const uint32_t lane = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
    const char *tris = reinterpret_cast<const char *>(sc.packed_tris);
    unsigned long long left = wave_ballot(todo);
    now = false;
    at = 0;
#pragma unroll
    for (int n = 0; n < kCacheSlots; n++) {
        if (left != 0ull) {             // (uniform)
            const int lead = __builtin_ctzll(left);
            const uint32_t first = (uint32_t)__builtin_amdgcn_readlane((int)t.leaf_first, lead);
            const uint32_t count = parked_count((uint32_t)__builtin_amdgcn_readlane((int)t.leaf_count, lead), t.leaf_cap);
            const bool same = todo && t.leaf_first == first;      // (one leaf, one record: the same count)
            left &= ~wave_ballot(same);
            if (same)
                at = (uint32_t)n;   // (the slot's number: an inline constant; its byte offset below)
            now = now || same;
            todo = todo && !same;
            // whatever lies behind the leaf's last word comes along (the array ends in a spare record, capi.hip) and is
            // never read back
            const uint32_t chunks = (count << 1) + ((count + 3u) >> 2);   // ceil(9 count / 4)
            if (lane < chunks)
                __builtin_amdgcn_global_load_lds((cache_global_ptr *)(tris + (size_t)(first * 36u) + (lane << 4)),
                                                 (cache_lds_ptr *)(cache + (uint32_t)n * kCacheSlotBytes), 16, 0, 0);
        }
    }
    at = __umul24(at, kCacheSlotBytes);
}
// The slots are read by other lanes than wrote them: the fetches have landed (their counter says so), and the compiler keeps
// the reads behind this point.  (`pin`: any value the reads' addresses depend on.)
__device__ __forceinline__ void leaf_cache_wait(uint32_t &pin) { asm volatile("s_waitcnt vmcnt(0)" : "+v"(pin) : : "memory"); }

// a triangle's nine words from a slot (five LDS reads), as the three words the tests unpack
__device__ __forceinline__ void load_cached_triangle(const char *p, float4 &q0, float4 &q1, float4 &q2)
{
    const float *q = reinterpret_cast<const float *>(p);
    float a0 = q[0], a1 = q[1], a2 = q[2], a3 = q[3], a4 = q[4], a5 = q[5], a6 = q[6], a7 = q[7], a8 = q[8];
    asm volatile("" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7), "+v"(a8));
    q0 = make_float4(a0, a1, a2, a3);
    q1 = make_float4(a4, a5, a6, a7);
    q2 = make_float4(a8, 0.0f, 0.0f, 0.0f);
}

// The sequential loop over a stage whose first kCacheSlots distinct leaves come through the cache: a lane whose leaf has a
// slot reads its triangles from there, a lane whose leaf has none (a stage with more distinct leaves: the divergent waves,
// the ones a lone frame waits for) fetches them itself as before -- in the same rounds.
template <bool COUNT, bool BOUNDS>
__device__ __forceinline__ void leaf_loop_cached(const SceneView &sc, LaneTraversal &t, int state, RayCounters &rc, char *cache SHRAY_DIAG_PARAM)
{
    bool todo = state == LT_LEAF, served;
    uint32_t where;                         // the lane's next triangle: a byte offset into the slab (served) or into packed_tris
    leaf_cache_fill(sc, t, todo, served, where, cache);
    if (!served)
        where = __umul24(t.leaf_first, 36u);
    uint32_t mine = state == LT_LEAF ? parked_count(t.leaf_count, t.leaf_cap) : 0u;
    leaf_cache_wait(mine);
    uint32_t j = 0;
    do {
        SHRAY_DIAG_COUNT(1);
        if (j < mine) {
            float4 q0, q1, q2;
            if (served)
                load_cached_triangle(cache + where, q0, q1, q2);
            else
                load_packed_triangle_at(sc, where, q0, q1, q2);
            lane_test_triangle_loaded<COUNT, BOUNDS>(sc, t, t.leaf_first + j, rc, q0, q1, q2);
        }
        j++;
        where += 36u;
    } while (wave_ballot(j < mine));
}

// ... and its end: the parked lanes move on (fs:416-433)
template <bool COUNT, int BLOCK, bool PAIR>
__device__ __forceinline__ void leaf_finish(const SceneView &sc, LaneTraversal &t, int &state, uint32_t *stack, RayCounters &rc)
{
    if (PAIR) {
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
    {
        unsigned int most = (state == LT_LEAF) ? parked_count(t.leaf_count, t.leaf_cap) : 0u;
        for (int off = 32; off > 0; off >>= 1)
            most = max(most, (unsigned int)__shfl_xor((int)most, off, 64));
        const int bin = K == 1 ? 0 : (K == 2 ? 1 : (K <= 4 ? 2 : (K <= 8 ? 3 : (K <= 16 ? 4 : (K <= 32 ? 5 : 6)))));
        const unsigned int g = K <= 4 ? 16u : (K <= 8 ? 8u : (K <= 16 ? 4u : (K <= 32 ? 2u : 1u)));
        const unsigned long long rounds = (most + g - 1u) / g, chunks = (most * 9u + 3u) / 4u, staged = K > 32 ? 0ull : (chunks + g - 1u) / g;
#if SHRAY_DIAG_KHIST == 1
        diag_tally_ref[bin] += 1ull | (rounds << 24) | (staged << 44);
        diag_tally_ref[7] += (unsigned long long)most | ((unsigned long long)K << 32);   // sums of the longest leaf and of K
#else
        // SHRAY_DIAG_KHIST == 2: the stages with more than SHRAY_DIAG_KHIST_FROM parked lanes by the number of DISTINCT leaves
        // among them -- bins D = 1, 2, 3, 4, 5-8, 9-16, > 16 --, each word {stages; rounds}; [7] = sums of D and of K
        if (K > SHRAY_DIAG_KHIST_FROM) {
            unsigned long long left = parked;
            int distinct = 0;
            while (left) {
                const int lead = __builtin_ctzll(left);
                const uint32_t f = (uint32_t)__builtin_amdgcn_readlane((int)t.leaf_first, lead);
                left &= ~wave_ballot(state == LT_LEAF && t.leaf_first == f);
                distinct++;
            }
            const int dbin = distinct <= 4 ? distinct - 1 : (distinct <= 8 ? 4 : (distinct <= 16 ? 5 : 6));
            diag_tally_ref[dbin] += 1ull | (rounds << 24);
            diag_tally_ref[7] += (unsigned long long)distinct | ((unsigned long long)K << 32);
        }
        (void)bin;
        (void)staged;
#endif
    }
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

#include "leaf_asm.h"
