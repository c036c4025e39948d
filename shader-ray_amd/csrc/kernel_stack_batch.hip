// kernel_stack_batch.hip -- kernel id 0, the timed instances: `count` frames per launch, one-wave workgroups, the
// convergent driver (uniform_driver.h) over the per-ray LDS stack (stack_traversal.h).  Which instance a launch runs:
// capi.hip (leaf_stage_policy, DispatchOrder) and launch_stack_batch (kernel_stack.hip).
#include "kernel_stack_common.h"

namespace shray {

// spp == 1 and a zero diffuse colour get instances without the sample loop / the diffuse branch
template <bool ONE_SAMPLE, bool METAL, bool DEAL>
__global__ void __launch_bounds__(kBatchBlock, min_waves(METAL, DEAL, ONE_SAMPLE)) trace_stack_batch_kernel(SceneView sc, const FrameView *__restrict__ frames,
                                                                                               float4 *out, size_t frame_stride, int stack_levels,
                                                                                               int frame_count_arg)
{
    // (one sample, zero diffuse, dealing: the six-wave instance, SHRAY_MIN_WAVES_DEALT)
    stack_batch_body<ONE_SAMPLE, METAL, DEAL, 0, false, false, ONE_SAMPLE && METAL && DEAL && SHRAY_MIN_WAVES_DEALT <= 6>(sc, frames, out, frame_stride, stack_levels,
                                                                                                                   frame_count_arg, nullptr);
}

// Several spp == 1 zero-diffuse frames per launch (the throughput form): the dealt leaf stage at SEVEN waves per SIMD
// (72 registers, 28 B of scratch outside the loops).  Round 2 gave these launches the plain leaf loop for its eighth wave;
// since the dealt loop lost its register copies (round 3, leaf_stage.h) seven dealing waves beat eight plain ones by
// 2.7 %, while a lone frame still does best with six (no scratch): profiles/history/r03/dealt_occupancy_ab2.txt.
__global__ void __launch_bounds__(kBatchBlock, SHRAY_MIN_WAVES_DEALT_DENSE)
    trace_stack_batch_dense_kernel(SceneView sc, const FrameView *__restrict__ frames, float4 *out, size_t frame_stride, int stack_levels,
                                   int frame_count_arg)
{
    stack_batch_body<true, true, true, 0, false>(sc, frames, out, frame_stride, stack_levels, frame_count_arg, nullptr);
}

// The zero-diffuse dealing instances once more for launches that read a dispatch order (capi.hip: DispatchOrder -- lone
// frames, tile sets): DENSE = the seven-wave dealt instance of the throughput form (ONE_SAMPLE, DEAL).  Instances of their
// own because the two scalars an ordered launch carries through the kernel cost the others 2 % (R3.9).
template <bool ONE_SAMPLE, bool DEAL, bool DENSE>
__global__ void __launch_bounds__(kBatchBlock, DENSE ? SHRAY_MIN_WAVES_DEALT_DENSE : min_waves(true, DEAL, ONE_SAMPLE))
    trace_stack_batch_ordered_kernel(SceneView sc, const FrameView *__restrict__ frames, float4 *out, size_t frame_stride, int stack_levels,
                                     int frame_count_arg)
{
    stack_batch_body<ONE_SAMPLE, true, DEAL, 0, false, true, ONE_SAMPLE && DEAL && !DENSE && SHRAY_MIN_WAVES_DEALT <= 6>(sc, frames, out, frame_stride, stack_levels,
                                                                                                                 frame_count_arg, nullptr);
}

void launch_stack_batch_timed(const SceneView &sc, const BatchLaunch &b)
{
    const bool one = b.one, metallic = b.metallic, deal = b.deal;
#define SHRAY_LAUNCH_BATCH(K) hipLaunchKernelGGL((K), b.grid, b.block, b.lds_bytes, b.stream, sc, b.d_frames, b.out, b.frame_stride, b.stack_levels, b.count)
    // `ordered` (capi.hip: DispatchOrder; zero-diffuse launches only): the instances that read a dispatch order
    if (b.ordered && metallic && b.dense)
        SHRAY_LAUNCH_BATCH((trace_stack_batch_ordered_kernel<true, true, true>));
    else if (b.ordered && metallic && one && deal)
        SHRAY_LAUNCH_BATCH((trace_stack_batch_ordered_kernel<true, true, false>));
    else if (b.ordered && metallic && !one && deal)
        SHRAY_LAUNCH_BATCH((trace_stack_batch_ordered_kernel<false, true, false>));
    // the throughput form of the headline workload deals its leaves at its own occupancy
    else if (b.dense)
        SHRAY_LAUNCH_BATCH(trace_stack_batch_dense_kernel);
    // `deal` (capi.hip: leaf_stage_policy) selects the leaf stage of each class of instances
    else if (one && metallic)
        SHRAY_LAUNCH_BATCH((trace_stack_batch_kernel<true, true, true>));
    else if (one && !deal)
        SHRAY_LAUNCH_BATCH((trace_stack_batch_kernel<true, false, false>));
    else if (one)
        SHRAY_LAUNCH_BATCH((trace_stack_batch_kernel<true, false, true>));
    else if (!metallic && !deal)
        SHRAY_LAUNCH_BATCH((trace_stack_batch_kernel<false, false, false>));
    else if (!metallic)
        SHRAY_LAUNCH_BATCH((trace_stack_batch_kernel<false, false, true>));
    else if (deal)
        SHRAY_LAUNCH_BATCH((trace_stack_batch_kernel<false, true, true>));
    else
        SHRAY_LAUNCH_BATCH((trace_stack_batch_kernel<false, true, false>));
#undef SHRAY_LAUNCH_BATCH
}

}   // namespace shray
