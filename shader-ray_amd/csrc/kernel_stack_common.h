// kernel_stack_common.h -- what the translation units of kernel id 0 share (kernel_stack*.hip): block sizes, the
// occupancy each class of instance is compiled for, the body of the batch kernels, and the shape of a batch launch.
// The instances are spread over three translation units so that an edit of the traversal does not cost one 45-second
// compile on one core: kernel_stack_batch.hip (the timed batch instances), kernel_stack_tally.hip (their tallying twins
// and the pair traversal), kernel_stack.hip (one frame per launch: the counting twins and the shader's debug views).
//
// LDS: BLOCK * stack_levels * 4 bytes of dynamic shared memory (+ 64 bytes per wave for the dealt leaf stage);
// stack_levels is the deepest stack the tree can ask for (computed at scene creation): 3.9 KB per wave for the
// bunny-class tree (15 levels), 6.6 KB for the 1M-triangle tree (26 levels: 24 waves per CU).
#pragma once

#include "launch.h"
#include "stack_traversal.h"
#include "uniform_driver.h"

namespace shray {

constexpr int kBlock = 256;
// the convergent batch instances (every timed launch) are one-wave workgroups: a wave tile's LDS and wave slot are
// released when THAT wave ends instead of when the slowest wave of its 16x16 patch does
constexpr int kBatchBlock = 64;

// Waves per SIMD the register allocator must leave room for, per class of instance (-D overrides them for A/B builds:
// `make variant`).  Measured: profiles/history/r02/leaf_stage_ab.txt, profiles/history/r03/dealt_occupancy_ab2.txt, profiles/r04/occupancy_ab.txt.
#ifndef SHRAY_MIN_WAVES
#define SHRAY_MIN_WAVES 8                 // zero-diffuse multi-sample instances with the plain leaf loop (64 registers): config 5's 4K
                                          // 16 spp frame 10.02 / 9.84 ms at 7 / 8 (10.60 at 6), gold 4 spp +3 % (round 4)
#endif
#ifndef SHRAY_MIN_WAVES_DEALT
#define SHRAY_MIN_WAVES_DEALT 6           // dealt leaf stage, one sample, one frame per launch: a second ray's worth of values (<= 80)
#endif
#ifndef SHRAY_MIN_WAVES_DEALT_DENSE
#define SHRAY_MIN_WAVES_DEALT_DENSE 7     // ... several frames per launch, the throughput form (72 registers, 28 B of scratch outside the
#endif                                    // loops): 9,369 / 9,450 / 9,247 Mrays/s at 8 / 7 / 6 (round 4)
#ifndef SHRAY_MIN_WAVES_DEALT_MULTI
#define SHRAY_MIN_WAVES_DEALT_MULTI 8     // ... multi-sample (the divergent scenes are latency-bound, every wave is worth its spills:
#endif                                    // the 1M-triangle scene at 4 spp 2.56 / 2.68 / 2.90 ms at 8 / 7 / 6, round 4)
#ifndef SHRAY_MIN_WAVES_GENERAL
#define SHRAY_MIN_WAVES_GENERAL 7         // diffuse / shadow-ray instances with the dealt stage, one sample: a plaster lone frame 0.700 /
#endif                                    // 0.688 / 0.675 / 0.683 ms at 5 / 6 / 7 / 8 (round 4: profiles/r04/general_occupancy_ab.txt)
#ifndef SHRAY_MIN_WAVES_GENERAL_MULTI
#define SHRAY_MIN_WAVES_GENERAL_MULTI 8   // ... multi-sample (trees larger than an L2 share): the 1M-triangle scene, plaster, 4 spp
#endif                                    // 4.99 / 4.50 / 4.17 / 4.06 ms at 5 / 6 / 7 / 8
#ifndef SHRAY_MIN_WAVES_VIEW
#define SHRAY_MIN_WAVES_VIEW 5            // the shader's debug views and the counting twins (256-thread workgroups, never timed)
#endif
#ifndef SHRAY_MIN_WAVES_GENERAL_PLAIN
#define SHRAY_MIN_WAVES_GENERAL_PLAIN 8   // ... with the plain leaf loop (cache-resident scenes), one sample: plaster 1 spp in the throughput form 7,842 ->
                                          // 7,998 Mrays/s at 8 (round 5: profiles/r05/one_sample_occupancy_ab.txt; round 4 had 6,649 -> 7,082 from 5 to 7)
#endif
#ifndef SHRAY_MIN_WAVES_GENERAL_PLAIN_MULTI
#define SHRAY_MIN_WAVES_GENERAL_PLAIN_MULTI 8   // ... multi-sample (config 3).  Round 5, after the sample loop stopped carrying the lane's
                                                // pixel: 13.01 / 12.49 / 12.23 ms at 6 / 7 / 8 (profiles/r05/occupancy_ab.txt) -- the eighth
                                                // wave wins although it costs 32 B more scratch (124 against 92 B per lane)
#endif
#ifndef SHRAY_MIN_WAVES_PAIR
#define SHRAY_MIN_WAVES_PAIR 6
#endif
#ifndef SHRAY_MIN_WAVES_PAIR_GENERAL
#define SHRAY_MIN_WAVES_PAIR_GENERAL 5
#endif

constexpr int min_waves(bool metal, bool deal, bool one_sample = true)
{
    return metal ? (deal ? (one_sample ? SHRAY_MIN_WAVES_DEALT : SHRAY_MIN_WAVES_DEALT_MULTI) : SHRAY_MIN_WAVES)
                 : (deal ? (one_sample ? SHRAY_MIN_WAVES_GENERAL : SHRAY_MIN_WAVES_GENERAL_MULTI)
                         : (one_sample ? SHRAY_MIN_WAVES_GENERAL_PLAIN : SHRAY_MIN_WAVES_GENERAL_PLAIN_MULTI));
}

inline bool one_sample(const FrameView &fr) { return fr.spp == 1; }
inline bool metal(const FrameView &fr)
{
    return !(fr.diffuse_color[0] > 0.0f && fr.diffuse_color[1] > 0.0f && fr.diffuse_color[2] > 0.0f);
}
inline bool plain_view(const FrameView &fr) { return !(fr.which == 1 || fr.which == 2 || fr.which == 3 || fr.which == 5); }

#ifndef SHRAY_LDS_PAD
#define SHRAY_LDS_PAD 0u   // unused LDS per wave (occupancy experiments)
#endif
// the batch instances whose sequential leaf loop reads its triangles from the wave's leaf cache in LDS (leaf_cache.h)
constexpr bool caches_leaves(bool pair) { return SHRAY_LEAF_CACHE != 0 && !pair; }

// The multi-sample instances with the plain leaf loop (cache-resident scenes: configs 3 and 5) keep the words of the sample loop
// that are only touched BETWEEN traversals in LDS, beside the wave's stack, instead of in registers the traversal spills
// (uniform_driver.h: ParkedState): four words per lane with a zero diffuse colour, five with a diffuse term.  Measured room (round 5,
// profiles/r05/lds_pad_sweep_configs.txt): a wave of the bunny-class scene (3,904 bytes) has 1,024 bytes to spare at eight waves per
// SIMD and 1,280 at seven; the 1M-triangle scene's 26-level stacks have none (any padding costs 16 %), and its instances deal.
#ifndef SHRAY_PARK
#define SHRAY_PARK 0
#endif
constexpr bool parks_state(bool one_sample, bool deal, bool pair) { return SHRAY_PARK != 0 && !one_sample && !deal && !pair; }
constexpr uint32_t park_bytes(bool metal) { return metal ? 1024u : 1280u; }

inline size_t stack_lds_bytes(int stack_levels, int block = kBlock, bool cache = false, uint32_t park = 0u)
{
    // stack columns + per wave: the dealt leaf stage's id table (64 bytes), in the instances that have one the leaf cache, and the
    // parked sample-loop state
    return (size_t)block * (size_t)stack_levels * sizeof(uint32_t) + (size_t)(block / 64) * (kIdsBytes + (cache ? kCacheBytes : 0u) + park + SHRAY_LDS_PAD);
}

template <bool DEAL, int BLOCK = kBlock, bool PAIR = false, bool CACHE = false, bool ROOMY = false>
__device__ __forceinline__ StackTraversal<BLOCK, DEAL, PAIR, CACHE, ROOMY> make_traversal(uint32_t *lds, int stack_levels)
{
    StackTraversal<BLOCK, DEAL, PAIR, CACHE, ROOMY> trav;
    trav.stack = lds + threadIdx.x;
    trav.ids = reinterpret_cast<uint8_t *>(lds + (size_t)stack_levels * BLOCK) + (threadIdx.x >> 6) * (kIdsBytes + (CACHE ? kCacheBytes : 0u));
    return trav;
}

// The body of every convergent batch kernel: one-wave workgroups, four (times the lanes per pixel of a multi-sample
// frame) per 16x16 patch.
// DEAL: the dealt leaf stage (leaf_stage.h) instead of the plain leaf loop.
// TALLY: 0 = the timed kernels; 1 = the same form with per-ray work tallies (what the timed form does); 2 = tallies of
// the reference's walk (one lane per pixel, every shadow ray to its end) -- the counting twin of the pair traversal.
// PAIR: both children of a node per turn (variants/pair_traversal.h).  ORDERED: the launch reads a dispatch order (capi.hip).
// ROOMY: the kernel is compiled for six waves per SIMD (stack_traversal.h).
template <bool ONE_SAMPLE, bool METAL, bool DEAL, int TALLY, bool PAIR, bool ORDERED = false, bool ROOMY = false>
__device__ __forceinline__ void stack_batch_body(const SceneView &sc, const FrameView *__restrict__ frames, float4 *out, size_t frame_stride,
                                                 int stack_levels, int frame_count_arg, DeviceCounters *counters)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t lds_stack[];
    using Traversal = StackTraversal<kBatchBlock, DEAL, PAIR, caches_leaves(PAIR), ROOMY>;
    Traversal trav = make_traversal<DEAL, kBatchBlock, PAIR, caches_leaves(PAIR), ROOMY>(lds_stack, stack_levels);
    // the frames of a launch share grid.x, frame index fastest after the (XCD, wave-of-patch) bits: the same patch
    // of every frame starts at about the same time on the same XCD, so the last frame's long-running waves do not
    // start when the launch is half over (what a lone launch, or the last of a run, then waits for)
    const unsigned int frame_count = gridDim.y == 1 ? (unsigned int)frame_count_arg : 1u;
    unsigned int frame = blockIdx.y, block_index = blockIdx.x;
    if (frame_count > 1u) {
        const FrameView &f0 = frames[0];
        const unsigned int log_waves = 2u + (ONE_SAMPLE ? 0u : f0.sample_log_x + f0.sample_log_y);
        const unsigned int b = blockIdx.x, k = b >> 3, rest = k >> log_waves;
        frame = rest % frame_count;
        block_index = ((((rest / frame_count) << log_waves) | (k & ((1u << log_waves) - 1u))) << 3) | (b & 7u);
    }
    // (the full-walk tallying twins, TALLY == 2, run one lane per pixel: a form of their own, nothing parked)
    constexpr bool PARK = parks_state(ONE_SAMPLE, DEAL, PAIR) && TALLY != 2;
    float *park = PARK ? reinterpret_cast<float *>(trav.ids + kIdsBytes + (caches_leaves(PAIR) ? kCacheBytes : 0u)) : nullptr;
    trace_pixels_uniform<Traversal, TALLY != 0, ONE_SAMPLE, METAL, TALLY == 1, ORDERED, PARK>(sc, frames[frame], out + (size_t)frame * frame_stride, counters, trav,
                                                                                     block_index, park);
}

// The shape of one batch launch (kernel_stack.hip: launch_stack_batch works it out) and the choices that select an instance.
struct BatchLaunch {
    dim3 grid, block;
    size_t lds_bytes;
    hipStream_t stream;
    const FrameView *d_frames;
    float4 *out;
    size_t frame_stride;
    int stack_levels, count;
    bool one, metallic, deal, dense, ordered;   // one sample per pixel; every frame zero-diffuse; dealt leaf stage; the
                                                // throughput form's seven-wave dealing instance; reads a dispatch order
};
// kernel_stack_batch.hip: the timed instances;  kernel_stack_tally.hip: the tallying twins, the pair traversal (timed and
// tallying: `tally` == nullptr selects the timed form; full_walk: the reference's walk instead of the timed form's)
void launch_stack_batch_timed(const SceneView &sc, const BatchLaunch &b);
void launch_stack_batch_tally(const SceneView &sc, const BatchLaunch &b, DeviceCounters *tally);
void launch_stack_batch_pair(const SceneView &sc, const BatchLaunch &b, DeviceCounters *tally, bool full_walk);

}   // namespace shray
