// kernel_stack.hip -- kernel id 0: per-ray LDS stack over the packed BVH (stack_traversal.h).
//
// LDS: BLOCK * stack_levels * 4 bytes of dynamic shared memory (+ 64 bytes per wave for the dealt leaf stage);
// stack_levels is the deepest stack the tree can ask for (computed at scene creation): 3.9 KB per wave for the
// bunny-class tree (15 levels), 6.6 KB for the 1M-triangle tree (26 levels: 24 waves per CU).
#include "launch.h"
#include "stack_traversal.h"
#include "uniform_driver.h"

namespace shray {

constexpr int kBlock = 256;
// the convergent batch instances (every timed launch) as one-wave workgroups: a wave tile's LDS and wave slot are
// released when THAT wave ends instead of when the slowest wave of its 16x16 patch does
constexpr int kBatchBlock = SHRAY_WAVE_BLOCKS ? 64 : 256;
#ifdef SHRAY_DIAGNOSTICS
bool g_diag_plain_kernel = false;
#endif

// Waves per SIMD the register allocator must leave room for (measured per instance, profiles/r02/leaf_stage_ab.txt):
// the plain gold instances are asked for seven (<= 72 registers); the spp == 1 one needs 63 since the library is
// built without the SLP vectoriser and runs eight, the multi-sample one 72 with its spills outside the loops
// (forced to eight it is slower).
#ifndef SHRAY_MIN_WAVES
#define SHRAY_MIN_WAVES 7
#endif
// the instances whose leaf stage deals triangles to idle lanes (wave_traversal.h) hold a second ray's worth of
// values while they do: six waves per SIMD (<= 80 registers) keeps the spills out of the loops
#ifndef SHRAY_MIN_WAVES_DEALT
#define SHRAY_MIN_WAVES_DEALT 6
#endif
// ... and its multi-sample form (the divergent scenes: latency-bound, every extra wave is worth its spills: the
// 1M-triangle scene at 4 spp 3.10 / 2.96 / 2.85 ms at 6 / 7 / 8 waves per SIMD, profiles/r02/leaf_stage_ab.txt section 22)
#ifndef SHRAY_MIN_WAVES_DEALT_MULTI
#define SHRAY_MIN_WAVES_DEALT_MULTI 8
#endif
// the instances with the diffuse / shadow-ray branch carry more state: one wave fewer
#ifndef SHRAY_MIN_WAVES_GENERAL
#define SHRAY_MIN_WAVES_GENERAL 5
#endif
// the diffuse / shadow-ray instances follow the leaf-stage policy too: on cache-resident scenes the plain leaf loop,
// which fits six waves per SIMD (plaster 8 spp 2.57 -> 2.42 ms, profiles/r02/leaf_stage_ab.txt section 22)
#ifndef SHRAY_GENERAL_PLAIN
#define SHRAY_GENERAL_PLAIN 1
#endif
#ifndef SHRAY_MIN_WAVES_GENERAL_PLAIN
#define SHRAY_MIN_WAVES_GENERAL_PLAIN 6
#endif
// launch the spp == 1 / metal instances of the plain kernel where they apply
// the frames of a launch interleaved along grid.x instead of stacked on grid.y (convergent instances)
#ifndef SHRAY_INTERLEAVE_FRAMES
#define SHRAY_INTERLEAVE_FRAMES 1
#endif
#ifndef SHRAY_SPECIALIZE
#define SHRAY_SPECIALIZE 1
#endif
#ifndef SHRAY_LDS_PAD
#define SHRAY_LDS_PAD 0
#endif

// Two drivers share the traversal.  Plain frames (which == 0, every timed launch) run trace() in its
// convergent form (uniform_driver.h), where all 64 lanes enter each traversal together and the dealt leaf
// stage can use the idle ones; the shader's debug views (which = 1, 2, 3, 5) and the patch-order experiment
// keep the per-lane driver of trace_common.h.
static bool one_sample(const FrameView &fr) { return fr.spp == 1; }
static bool metal(const FrameView &fr)
{
    return !(fr.diffuse_color[0] > 0.0f && fr.diffuse_color[1] > 0.0f && fr.diffuse_color[2] > 0.0f);
}
static bool plain_view(const FrameView &fr) { return !(fr.which == 1 || fr.which == 2 || fr.which == 3 || fr.which == 5); }

template <bool DEAL, int BLOCK = kBlock, bool PAIR = false>
__device__ __forceinline__ StackTraversal<BLOCK, DEAL, PAIR> make_traversal(uint32_t *lds, int stack_levels, const SceneView &sc)
{
    StackTraversal<BLOCK, DEAL, PAIR> trav;
    trav.stack = lds + threadIdx.x;
    trav.ids = reinterpret_cast<uint8_t *>(lds + (size_t)stack_levels * BLOCK) + (threadIdx.x & ~63u);
#if SHRAY_LDS_TOP
    // experiment: the workgroup stages the top of the tree (the first SHRAY_LDS_TOP nodes, numbered breadth first
    // by capi.hip under the same flag) behind the stack columns and the id tables
    float4 *top = reinterpret_cast<float4 *>(lds + (size_t)stack_levels * kBlock + kBlock / 4);
    const float4 *nodes = reinterpret_cast<const float4 *>(sc.packed_nodes);
    for (unsigned int k = threadIdx.x; k < 2u * SHRAY_LDS_TOP && k < 2u * sc.group_count; k += kBlock)
        top[k] = nodes[k];
    __syncthreads();
    trav.top = top;
#endif
    return trav;
}

constexpr int min_waves(bool metal, bool deal, bool one_sample = true)
{
    return metal ? (deal ? (one_sample ? SHRAY_MIN_WAVES_DEALT : SHRAY_MIN_WAVES_DEALT_MULTI) : SHRAY_MIN_WAVES)
                 : (deal ? SHRAY_MIN_WAVES_GENERAL : SHRAY_MIN_WAVES_GENERAL_PLAIN);
}

// spp == 1 and a zero diffuse colour get instances without the sample loop / the diffuse branch
template <bool COUNT, bool ONE_SAMPLE, bool METAL>
__global__ void __launch_bounds__(kBlock, min_waves(METAL, true)) trace_stack_kernel(SceneView sc, FrameView fr, float4 *out, DeviceCounters *counters,
                                                                                    int stack_levels)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t lds_stack[];
    StackTraversal<kBlock, true> trav = make_traversal<true>(lds_stack, stack_levels, sc);
    trace_pixels_uniform<StackTraversal<kBlock, true>, COUNT, ONE_SAMPLE, METAL>(sc, fr, out, counters, trav);
}

template <bool COUNT, bool DIFF>
__global__ void __launch_bounds__(kBlock, SHRAY_MIN_WAVES_GENERAL) trace_stack_view_kernel(SceneView sc, FrameView fr, float4 *out, DeviceCounters *counters,
                                                                                          int stack_levels)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t lds_stack[];
    StackTraversal<kBlock, false> trav = make_traversal<false>(lds_stack, stack_levels, sc);
    trace_pixels<StackTraversal<kBlock, false>, COUNT, DIFF>(sc, fr, out, counters, trav);
}

// Batch forms: workgroup (x, y) renders patch x of frame y.  Workgroups are dispatched x-fastest, so
// frame 0 starts first and later frames fill the SIMDs its long-running waves leave idle.
// DEAL = false is the throughput instance (several spp == 1 frames per launch): one wave more per SIMD, plain leaf loop
// TALLY: 0 = the timed kernels; 1 = the same form with per-ray work tallies (what the timed form does); 2 = tallies of
// the reference's walk (one lane per pixel, every shadow ray to its end) -- the counting twin of the pair traversal.
// PAIR: both children of a node per turn (wave_traversal.h)
template <bool ONE_SAMPLE, bool METAL, bool DEAL, int TALLY, bool PAIR, bool ORDERED = false>
__device__ __forceinline__ void stack_batch_body(const SceneView &sc, const FrameView *__restrict__ frames, float4 *out, size_t frame_stride,
                                                 int stack_levels, int frame_count_arg, DeviceCounters *counters)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t lds_stack[];
    using Traversal = StackTraversal<kBatchBlock, DEAL, PAIR>;
    Traversal trav = make_traversal<DEAL, kBatchBlock, PAIR>(lds_stack, stack_levels, sc);
    if (ONE_SAMPLE)
        trav.keep_dealt = SHRAY_KEEP_WALKING_DEALT_ONE;
#if SHRAY_WAVE_BLOCKS == 2 && SHRAY_INTERLEAVE_FRAMES
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
    trace_pixels_uniform<Traversal, TALLY != 0, ONE_SAMPLE, METAL, TALLY == 1, ORDERED>(sc, frames[frame], out + (size_t)frame * frame_stride, counters, trav,
                                                                               block_index);
#else
    trace_pixels_uniform<Traversal, TALLY != 0, ONE_SAMPLE, METAL, TALLY == 1, ORDERED>(sc, frames[blockIdx.y], out + (size_t)blockIdx.y * frame_stride,
                                                                               counters, trav);
#endif
}

template <bool ONE_SAMPLE, bool METAL, bool DEAL>
__global__ void __launch_bounds__(kBatchBlock, min_waves(METAL, DEAL, ONE_SAMPLE)) trace_stack_batch_kernel(SceneView sc, const FrameView *__restrict__ frames,
                                                                                               float4 *out, size_t frame_stride, int stack_levels,
                                                                                               int frame_count_arg)
{
    stack_batch_body<ONE_SAMPLE, METAL, DEAL, 0, false>(sc, frames, out, frame_stride, stack_levels, frame_count_arg, nullptr);
}

// Several spp == 1 zero-diffuse frames per launch (the throughput form): the dealt leaf stage at SEVEN waves per SIMD
// (72 registers, 24 B of scratch).  Round 2 gave these launches the plain leaf loop for its eighth wave; since the dealt
// loop lost its register copies (round 3, wave_traversal.h) seven dealing waves beat eight plain ones by 2.7 %, while a
// lone frame still does best with six (no scratch): profiles/r03/dealt_occupancy_ab2.txt.
#ifndef SHRAY_THROUGHPUT_DEALT
#define SHRAY_THROUGHPUT_DEALT 1
#endif
#ifndef SHRAY_MIN_WAVES_DEALT_DENSE
#define SHRAY_MIN_WAVES_DEALT_DENSE 7
#endif
__global__ void __launch_bounds__(kBatchBlock, SHRAY_MIN_WAVES_DEALT_DENSE)
    trace_stack_batch_dense_kernel(SceneView sc, const FrameView *__restrict__ frames, float4 *out, size_t frame_stride, int stack_levels,
                                   int frame_count_arg)
{
    stack_batch_body<true, true, true, 0, false>(sc, frames, out, frame_stride, stack_levels, frame_count_arg, nullptr);
}

// The zero-diffuse dealing instances once more for launches that read a dispatch order (capi.hip: DispatchOrder -- lone
// frames, tile sets): DENSE = the seven-wave dealt instance of the throughput form (ONE_SAMPLE, DEAL)
template <bool ONE_SAMPLE, bool DEAL, bool DENSE>
__global__ void __launch_bounds__(kBatchBlock, DENSE ? SHRAY_MIN_WAVES_DEALT_DENSE : min_waves(true, DEAL, ONE_SAMPLE))
    trace_stack_batch_ordered_kernel(SceneView sc, const FrameView *__restrict__ frames, float4 *out, size_t frame_stride, int stack_levels,
                                     int frame_count_arg)
{
    stack_batch_body<ONE_SAMPLE, true, DEAL, 0, false, true>(sc, frames, out, frame_stride, stack_levels, frame_count_arg, nullptr);
}

// The pair traversal (dealt leaf stage): for launches that are bound by dependent round trips -- a lone frame, a tree
// larger than the L2 (capi.hip: pair_policy)
#ifndef SHRAY_MIN_WAVES_PAIR
#define SHRAY_MIN_WAVES_PAIR 6
#endif
#ifndef SHRAY_MIN_WAVES_PAIR_GENERAL
#define SHRAY_MIN_WAVES_PAIR_GENERAL 5
#endif
template <bool ONE_SAMPLE, bool METAL>
__global__ void __launch_bounds__(kBatchBlock, METAL ? SHRAY_MIN_WAVES_PAIR : SHRAY_MIN_WAVES_PAIR_GENERAL)
    trace_stack_batch_pair_kernel(SceneView sc, const FrameView *__restrict__ frames, float4 *out, size_t frame_stride, int stack_levels,
                                  int frame_count_arg)
{
    stack_batch_body<ONE_SAMPLE, METAL, true, 0, true>(sc, frames, out, frame_stride, stack_levels, frame_count_arg, nullptr);
}

// The same instances with per-ray work tallies (shray_render_counters_timed): what the TIMED form does -- sample lanes,
// shadow rays that stop at their first hit -- as opposed to the counting twins of trace_stack_kernel, which reproduce
// the reference's full traversals.  Never timed.
template <bool ONE_SAMPLE, bool METAL, bool DEAL>
__global__ void __launch_bounds__(kBatchBlock, 4) trace_stack_batch_tally_kernel(SceneView sc, const FrameView *__restrict__ frames, float4 *out,
                                                                                 size_t frame_stride, int stack_levels, int frame_count_arg,
                                                                                 DeviceCounters *counters)
{
    stack_batch_body<ONE_SAMPLE, METAL, DEAL, 1, false>(sc, frames, out, frame_stride, stack_levels, frame_count_arg, counters);
}
// ... and of the pair traversal: FULL_WALK = the reference's walk (its counting twin), else the timed form
template <bool ONE_SAMPLE, bool METAL, bool FULL_WALK>
__global__ void __launch_bounds__(kBatchBlock, 4) trace_stack_batch_pair_tally_kernel(SceneView sc, const FrameView *__restrict__ frames, float4 *out,
                                                                                      size_t frame_stride, int stack_levels, int frame_count_arg,
                                                                                      DeviceCounters *counters)
{
    stack_batch_body<ONE_SAMPLE, METAL, true, FULL_WALK ? 2 : 1, true>(sc, frames, out, frame_stride, stack_levels, frame_count_arg, counters);
}

template <bool DIFF>
__global__ void __launch_bounds__(kBlock, SHRAY_MIN_WAVES_GENERAL) trace_stack_view_batch_kernel(SceneView sc, const FrameView *__restrict__ frames,
                                                                                                float4 *out, size_t frame_stride, int stack_levels)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t lds_stack[];
    StackTraversal<kBlock, false> trav = make_traversal<false>(lds_stack, stack_levels, sc);
    trace_pixels<StackTraversal<kBlock, false>, false, DIFF>(sc, frames[blockIdx.y], out + (size_t)blockIdx.y * frame_stride, nullptr, trav);
}

static size_t stack_lds_bytes(int stack_levels, int block = kBlock)
{
    // stack columns + the dealt leaf stage's id tables (64 bytes per wave)
    return (size_t)block * (size_t)stack_levels * sizeof(uint32_t) + (size_t)block + SHRAY_LDS_PAD + (size_t)SHRAY_LDS_TOP * 32;
}

// `all_metal`: every frame of the batch has a zero diffuse colour; `all_plain`: every frame has which == 0;
// `deal`: the dealt leaf stage instead of the plain one (capi.hip: leaf_stage_policy)
hipError_t launch_stack_batch(const SceneView &sc, const FrameView *d_frames, int count, const FrameView &first, bool all_metal,
                              bool all_plain, bool deal, float4 *out, size_t frame_stride, hipStream_t stream, int stack_levels,
                              DeviceCounters *tally, bool pair, bool tally_full_walk, bool ordered)
{
    // the view instances run 256-thread workgroups (a patch each), the convergent ones kBatchBlock-thread workgroups
    const bool view_instance = !all_plain;
    // one-wave workgroups: four per patch, times the lanes per pixel of a multi-sample frame (uniform_driver.h)
    const unsigned int sample_lanes = (kBatchBlock == 64 && !one_sample(first) && !(tally && tally_full_walk))
                                          ? (1u << (first.sample_log_x + first.sample_log_y)) : 1u;   // (the reference's walk: one lane per pixel)
    const unsigned int per_patch = view_instance ? 1u : (unsigned int)(kBlock / kBatchBlock) * sample_lanes;
    const unsigned int grid_patches = (SHRAY_WAVE_BLOCKS == 2 && !view_instance) ? ((first.total_patches + 7u) & ~7u) : first.total_patches;
    // convergent instances: the frames of the launch interleaved along grid.x (see the kernel); the view instances keep grid.y = frame
    // (HIP rejects a launch whose gridDim.x * blockDim.x reaches 2^32: a batch that large -- 4K at 16 spp from 33 frames
    // on -- goes back to grid.y = frame, which the kernel tells by gridDim.y > 1)
    const bool interleave = SHRAY_WAVE_BLOCKS == 2 && SHRAY_INTERLEAVE_FRAMES && !view_instance && count > 1 &&
                            (unsigned long long)grid_patches * per_patch * (unsigned long long)count * kBatchBlock < (1ull << 32);
    const dim3 grid(grid_patches * per_patch * (interleave ? (unsigned)count : 1u), interleave ? 1u : (unsigned)count),
        block(view_instance ? kBlock : kBatchBlock);
    const size_t lds_bytes = view_instance ? stack_lds_bytes(stack_levels) : stack_lds_bytes(stack_levels, kBatchBlock);
    const bool one = SHRAY_SPECIALIZE && one_sample(first), metallic = SHRAY_SPECIALIZE && all_metal;
#define SHRAY_LAUNCH_VIEW_BATCH(K) hipLaunchKernelGGL((K), grid, block, lds_bytes, stream, sc, d_frames, out, frame_stride, stack_levels)
#define SHRAY_LAUNCH_BATCH(K) hipLaunchKernelGGL((K), grid, block, lds_bytes, stream, sc, d_frames, out, frame_stride, stack_levels, count)
#define SHRAY_LAUNCH_TALLY(K) hipLaunchKernelGGL((K), grid, block, lds_bytes, stream, sc, d_frames, out, frame_stride, stack_levels, count, tally)
#define SHRAY_LAUNCH_PAIR_TALLY(O, M)                                                                    \
    do {                                                                                                \
        if (tally_full_walk)                                                                            \
            SHRAY_LAUNCH_TALLY((trace_stack_batch_pair_tally_kernel<O, M, true>));                      \
        else                                                                                            \
            SHRAY_LAUNCH_TALLY((trace_stack_batch_pair_tally_kernel<O, M, false>));                     \
    } while (0)
    if (tally && all_plain && pair) {
        if (one && metallic)
            SHRAY_LAUNCH_PAIR_TALLY(true, true);
        else if (one)
            SHRAY_LAUNCH_PAIR_TALLY(true, false);
        else if (metallic)
            SHRAY_LAUNCH_PAIR_TALLY(false, true);
        else
            SHRAY_LAUNCH_PAIR_TALLY(false, false);
        return hipGetLastError();
    }
#undef SHRAY_LAUNCH_PAIR_TALLY
    // the throughput form of the headline workload deals its leaves too, at its own occupancy (trace_stack_batch_dense_kernel)
    const bool dense = SHRAY_THROUGHPUT_DEALT && one && metallic && !deal && !pair && all_plain;
    if (tally && all_plain) {   // the same choice of instance as below, with tallies
        if (one && metallic && (deal || dense))
            SHRAY_LAUNCH_TALLY((trace_stack_batch_tally_kernel<true, true, true>));
        else if (one && metallic)
            SHRAY_LAUNCH_TALLY((trace_stack_batch_tally_kernel<true, true, false>));
#if SHRAY_GENERAL_PLAIN
        else if (one && !deal)
            SHRAY_LAUNCH_TALLY((trace_stack_batch_tally_kernel<true, false, false>));
        else if (!one && !metallic && !deal)
            SHRAY_LAUNCH_TALLY((trace_stack_batch_tally_kernel<false, false, false>));
#endif
        else if (one)
            SHRAY_LAUNCH_TALLY((trace_stack_batch_tally_kernel<true, false, true>));
        else if (metallic && deal)
            SHRAY_LAUNCH_TALLY((trace_stack_batch_tally_kernel<false, true, true>));
        else if (metallic)
            SHRAY_LAUNCH_TALLY((trace_stack_batch_tally_kernel<false, true, false>));
        else
            SHRAY_LAUNCH_TALLY((trace_stack_batch_tally_kernel<false, false, true>));
        return hipGetLastError();
    }
#undef SHRAY_LAUNCH_TALLY
    if (!all_plain && (first.which == 1 || first.which == 2))
        SHRAY_LAUNCH_VIEW_BATCH(trace_stack_view_batch_kernel<true>);
    else if (!all_plain)
        SHRAY_LAUNCH_VIEW_BATCH(trace_stack_view_batch_kernel<false>);
    // `pair` (capi.hip: pair_policy): both children per node turn, dealt leaf stage
    else if (pair && one && metallic)
        SHRAY_LAUNCH_BATCH((trace_stack_batch_pair_kernel<true, true>));
    else if (pair && one)
        SHRAY_LAUNCH_BATCH((trace_stack_batch_pair_kernel<true, false>));
    else if (pair && metallic)
        SHRAY_LAUNCH_BATCH((trace_stack_batch_pair_kernel<false, true>));
    else if (pair)
        SHRAY_LAUNCH_BATCH((trace_stack_batch_pair_kernel<false, false>));
    // `ordered` (capi.hip: DispatchOrder; zero-diffuse launches only): the same choice among the instances that read a
    // dispatch order
    else if (ordered && metallic && dense)
        SHRAY_LAUNCH_BATCH((trace_stack_batch_ordered_kernel<true, true, true>));
    else if (ordered && metallic && one && deal)
        SHRAY_LAUNCH_BATCH((trace_stack_batch_ordered_kernel<true, true, false>));
    else if (ordered && metallic && !one && deal)
        SHRAY_LAUNCH_BATCH((trace_stack_batch_ordered_kernel<false, true, false>));
    // `deal` (chosen in capi.hip: leaf_stage_policy) selects the leaf stage of each class of instances
    else if (dense)
        SHRAY_LAUNCH_BATCH(trace_stack_batch_dense_kernel);
    else if (one && metallic && deal)
        SHRAY_LAUNCH_BATCH((trace_stack_batch_kernel<true, true, true>));
    else if (one && metallic)
        SHRAY_LAUNCH_BATCH((trace_stack_batch_kernel<true, true, false>));
#if SHRAY_GENERAL_PLAIN
    else if (one && !deal)
        SHRAY_LAUNCH_BATCH((trace_stack_batch_kernel<true, false, false>));
    else if (!one && !metallic && !deal)
        SHRAY_LAUNCH_BATCH((trace_stack_batch_kernel<false, false, false>));
#endif
    else if (one)
        SHRAY_LAUNCH_BATCH((trace_stack_batch_kernel<true, false, true>));
    else if (metallic && deal)
        SHRAY_LAUNCH_BATCH((trace_stack_batch_kernel<false, true, true>));
    else if (metallic)
        SHRAY_LAUNCH_BATCH((trace_stack_batch_kernel<false, true, false>));
    else
        SHRAY_LAUNCH_BATCH((trace_stack_batch_kernel<false, false, true>));
#undef SHRAY_LAUNCH_BATCH
#undef SHRAY_LAUNCH_VIEW_BATCH
    return hipGetLastError();
}

hipError_t launch_stack(const SceneView &sc, const FrameView &fr, float4 *out, DeviceCounters *counters,
                        hipStream_t stream, int stack_levels)
{
    const dim3 grid(fr.total_patches), block(kBlock);
    const size_t lds_bytes = stack_lds_bytes(stack_levels);
#define SHRAY_LAUNCH(K) hipLaunchKernelGGL((K), grid, block, lds_bytes, stream, sc, fr, out, counters, stack_levels)
    const bool diff = fr.which == 1 || fr.which == 2;   // only those views need the ray differentials carried along
    if (!plain_view(fr) || fr.patch_order) {
        if (counters && diff)
            SHRAY_LAUNCH((trace_stack_view_kernel<true, true>));
        else if (counters)
            SHRAY_LAUNCH((trace_stack_view_kernel<true, false>));
        else if (diff)
            SHRAY_LAUNCH((trace_stack_view_kernel<false, true>));
        else
            SHRAY_LAUNCH((trace_stack_view_kernel<false, false>));
    }
#ifdef SHRAY_DIAGNOSTICS
    else if (counters && g_diag_plain_kernel)
        SHRAY_LAUNCH((trace_stack_kernel<false, false, false>));
#endif
    else if (counters)
        SHRAY_LAUNCH((trace_stack_kernel<true, false, false>));
    else if (SHRAY_SPECIALIZE && one_sample(fr) && metal(fr))
        SHRAY_LAUNCH((trace_stack_kernel<false, true, true>));
    else if (SHRAY_SPECIALIZE && one_sample(fr))
        SHRAY_LAUNCH((trace_stack_kernel<false, true, false>));
    else if (SHRAY_SPECIALIZE && metal(fr))
        SHRAY_LAUNCH((trace_stack_kernel<false, false, true>));
    else
        SHRAY_LAUNCH((trace_stack_kernel<false, false, false>));
#undef SHRAY_LAUNCH
    return hipGetLastError();
}

}   // namespace shray
