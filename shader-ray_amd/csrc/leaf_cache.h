// leaf_cache.h -- north_star's "triangle data staged into LDS tiles": a stage's distinct leaves fetched once, coalesced, straight into
// LDS (round 5).  Compiled in by -DSHRAY_LEAF_CACHE=1 (libshray_hip_leafcache.so, kept parity-green by tests/test_gpu_leaf_cache.py);
// measured slower than the shipped loops (profiles/EXPERIMENTS.md R5.1), so the product build only takes the constants from here.
#pragma once

namespace shray {

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

}   // namespace shray
