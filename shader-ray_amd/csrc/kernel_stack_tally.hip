// kernel_stack_tally.hip -- kernel id 0: the batch instances with per-ray work tallies (shray_render_counters_timed: what
// the TIMED form does -- sample lanes, shadow rays that stop at their first hit -- as opposed to the counting twins of
// kernel_stack.hip, which reproduce the reference's full traversals; never timed), and kernel id 3, the pair traversal
// (both children of a node per turn, variants/pair_traversal.h: inner_stage_pair), timed and tallying.
#include "kernel_stack_common.h"
#include "variants/pair_traversal.h"   // the pair instances (kernel id 3) live in this translation unit only

namespace shray {

template <bool ONE_SAMPLE, bool METAL, bool DEAL>
__global__ void __launch_bounds__(kBatchBlock, 4) trace_stack_batch_tally_kernel(SceneView sc, const FrameView *__restrict__ frames, float4 *out,
                                                                                 size_t frame_stride, int stack_levels, int frame_count_arg,
                                                                                 DeviceCounters *counters)
{
    stack_batch_body<ONE_SAMPLE, METAL, DEAL, 1, false>(sc, frames, out, frame_stride, stack_levels, frame_count_arg, counters);
}

template <bool ONE_SAMPLE, bool METAL>
__global__ void __launch_bounds__(kBatchBlock, METAL ? SHRAY_MIN_WAVES_PAIR : SHRAY_MIN_WAVES_PAIR_GENERAL)
    trace_stack_batch_pair_kernel(SceneView sc, const FrameView *__restrict__ frames, float4 *out, size_t frame_stride, int stack_levels,
                                  int frame_count_arg)
{
    stack_batch_body<ONE_SAMPLE, METAL, true, 0, true>(sc, frames, out, frame_stride, stack_levels, frame_count_arg, nullptr);
}
// FULL_WALK = the reference's walk (the pair traversal's counting twin), else the timed form
template <bool ONE_SAMPLE, bool METAL, bool FULL_WALK>
__global__ void __launch_bounds__(kBatchBlock, 4) trace_stack_batch_pair_tally_kernel(SceneView sc, const FrameView *__restrict__ frames, float4 *out,
                                                                                      size_t frame_stride, int stack_levels, int frame_count_arg,
                                                                                      DeviceCounters *counters)
{
    stack_batch_body<ONE_SAMPLE, METAL, true, FULL_WALK ? 2 : 1, true>(sc, frames, out, frame_stride, stack_levels, frame_count_arg, counters);
}

#define SHRAY_LAUNCH_BATCH(K) hipLaunchKernelGGL((K), b.grid, b.block, b.lds_bytes, b.stream, sc, b.d_frames, b.out, b.frame_stride, b.stack_levels, b.count)
#define SHRAY_LAUNCH_TALLY(K) hipLaunchKernelGGL((K), b.grid, b.block, b.lds_bytes, b.stream, sc, b.d_frames, b.out, b.frame_stride, b.stack_levels, b.count, tally)

// the same choice of instance as launch_stack_batch_timed, with tallies
void launch_stack_batch_tally(const SceneView &sc, const BatchLaunch &b, DeviceCounters *tally)
{
    const bool one = b.one, metallic = b.metallic, deal = b.deal || b.dense;
    if (one && metallic)
        SHRAY_LAUNCH_TALLY((trace_stack_batch_tally_kernel<true, true, true>));
    else if (one && !deal)
        SHRAY_LAUNCH_TALLY((trace_stack_batch_tally_kernel<true, false, false>));
    else if (one)
        SHRAY_LAUNCH_TALLY((trace_stack_batch_tally_kernel<true, false, true>));
    else if (!metallic && !deal)
        SHRAY_LAUNCH_TALLY((trace_stack_batch_tally_kernel<false, false, false>));
    else if (!metallic)
        SHRAY_LAUNCH_TALLY((trace_stack_batch_tally_kernel<false, false, true>));
    else if (deal)
        SHRAY_LAUNCH_TALLY((trace_stack_batch_tally_kernel<false, true, true>));
    else
        SHRAY_LAUNCH_TALLY((trace_stack_batch_tally_kernel<false, true, false>));
}

void launch_stack_batch_pair(const SceneView &sc, const BatchLaunch &b, DeviceCounters *tally, bool full_walk)
{
    const bool one = b.one, metallic = b.metallic;
#define SHRAY_LAUNCH_PAIR(O, M)                                                                          \
    do {                                                                                                \
        if (!tally)                                                                                     \
            SHRAY_LAUNCH_BATCH((trace_stack_batch_pair_kernel<O, M>));                                  \
        else if (full_walk)                                                                             \
            SHRAY_LAUNCH_TALLY((trace_stack_batch_pair_tally_kernel<O, M, true>));                      \
        else                                                                                            \
            SHRAY_LAUNCH_TALLY((trace_stack_batch_pair_tally_kernel<O, M, false>));                     \
    } while (0)
    if (one && metallic)
        SHRAY_LAUNCH_PAIR(true, true);
    else if (one)
        SHRAY_LAUNCH_PAIR(true, false);
    else if (metallic)
        SHRAY_LAUNCH_PAIR(false, true);
    else
        SHRAY_LAUNCH_PAIR(false, false);
#undef SHRAY_LAUNCH_PAIR
}
#undef SHRAY_LAUNCH_TALLY
#undef SHRAY_LAUNCH_BATCH

}   // namespace shray
