// kernel_stack.hip -- kernel id 0: per-ray LDS stack over the packed BVH (stack_traversal.h).  This translation unit:
// the one-frame-per-launch kernels (the counting twins, whose tallies equal the reference's traversals, and the shader's
// debug views) and the host side of every stack-kernel launch; the batch instances live in kernel_stack_batch.hip and
// kernel_stack_tally.hip (kernel_stack_common.h).
#include "kernel_stack_common.h"

namespace shray {

#ifdef SHRAY_DIAGNOSTICS
bool g_diag_plain_kernel = false;
#endif

// Two drivers share the traversal.  Plain frames (which == 0, every timed launch) run trace() in its
// convergent form (uniform_driver.h), where all 64 lanes enter each traversal together and the dealt leaf
// stage can use the idle ones; the shader's debug views (which = 1, 2, 3, 5) keep the per-lane driver of
// trace_common.h.
// spp == 1 and a zero diffuse colour get instances without the sample loop / the diffuse branch
template <bool COUNT, bool ONE_SAMPLE, bool METAL>
__global__ void __launch_bounds__(kBlock, SHRAY_MIN_WAVES_VIEW) trace_stack_kernel(SceneView sc, FrameView fr, float4 *out, DeviceCounters *counters,
                                                                                    int stack_levels)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t lds_stack[];
    using Traversal = StackTraversal<kBlock, true, false, caches_leaves(false)>;
    Traversal trav = make_traversal<true, kBlock, false, caches_leaves(false)>(lds_stack, stack_levels);
    trace_pixels_uniform<Traversal, COUNT, ONE_SAMPLE, METAL>(sc, fr, out, counters, trav);
}

template <bool COUNT, bool DIFF>
__global__ void __launch_bounds__(kBlock, SHRAY_MIN_WAVES_VIEW) trace_stack_view_kernel(SceneView sc, FrameView fr, float4 *out, DeviceCounters *counters,
                                                                                          int stack_levels)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t lds_stack[];
    StackTraversal<kBlock, false> trav = make_traversal<false>(lds_stack, stack_levels);
    trace_pixels<StackTraversal<kBlock, false>, COUNT, DIFF>(sc, fr, out, counters, trav);
}

// the debug views, `count` frames per launch: workgroup (x, y) renders patch x of frame y
template <bool DIFF>
__global__ void __launch_bounds__(kBlock, SHRAY_MIN_WAVES_VIEW) trace_stack_view_batch_kernel(SceneView sc, const FrameView *__restrict__ frames,
                                                                                                float4 *out, size_t frame_stride, int stack_levels)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t lds_stack[];
    StackTraversal<kBlock, false> trav = make_traversal<false>(lds_stack, stack_levels);
    trace_pixels<StackTraversal<kBlock, false>, false, DIFF>(sc, frames[blockIdx.y], out + (size_t)blockIdx.y * frame_stride, nullptr, trav);
}

// `all_metal`: every frame of the batch has a zero diffuse colour; `all_plain`: every frame has which == 0;
// `deal`: the dealt leaf stage instead of the plain one (capi.hip: leaf_stage_policy)
hipError_t launch_stack_batch(const SceneView &sc, const FrameView *d_frames, int count, const FrameView &first, bool all_metal,
                              bool all_plain, bool deal, float4 *out, size_t frame_stride, hipStream_t stream, int stack_levels,
                              DeviceCounters *tally, bool pair, bool tally_full_walk, bool ordered)
{
    // the view instances run 256-thread workgroups (a patch each), the convergent ones one-wave workgroups
    const bool view_instance = !all_plain;
    // one-wave workgroups: four per patch, times the lanes per pixel of a multi-sample frame (uniform_driver.h)
    const unsigned int sample_lanes = (!one_sample(first) && !(tally && tally_full_walk))
                                          ? (1u << (first.sample_log_x + first.sample_log_y)) : 1u;   // (the reference's walk: one lane per pixel)
    const unsigned int per_patch = view_instance ? 1u : (unsigned int)(kBlock / kBatchBlock) * sample_lanes;
    // workgroups go to the eight XCDs round robin; a patch's waves stay on one XCD (uniform_driver.h): whole groups of 8 patches
    const unsigned int grid_patches = !view_instance ? ((first.total_patches + 7u) & ~7u) : first.total_patches;
    // convergent instances: the frames of the launch interleaved along grid.x (stack_batch_body); the view instances keep grid.y = frame
    // (HIP rejects a launch whose gridDim.x * blockDim.x reaches 2^32: a batch that large -- 4K at 16 spp from 33 frames
    // on -- goes back to grid.y = frame, which the kernel tells by gridDim.y > 1)
    const bool interleave = !view_instance && count > 1 &&
                            (unsigned long long)grid_patches * per_patch * (unsigned long long)count * kBatchBlock < (1ull << 32);
    BatchLaunch b;
    b.grid = dim3(grid_patches * per_patch * (interleave ? (unsigned)count : 1u), interleave ? 1u : (unsigned)count);
    b.block = dim3(view_instance ? kBlock : kBatchBlock);
    b.lds_bytes = view_instance ? stack_lds_bytes(stack_levels) : stack_lds_bytes(stack_levels, kBatchBlock);
    b.stream = stream;
    b.d_frames = d_frames;
    b.out = out;
    b.frame_stride = frame_stride;
    b.stack_levels = stack_levels;
    b.count = count;
    b.one = one_sample(first);
    b.metallic = all_metal;
    b.deal = deal;
    // the throughput form of the headline workload deals its leaves too, at its own occupancy (trace_stack_batch_dense_kernel)
    b.dense = b.one && b.metallic && !deal && !pair && all_plain;
    b.ordered = ordered;
    if (!view_instance)
        b.lds_bytes = stack_lds_bytes(stack_levels, kBatchBlock, caches_leaves(pair),
                                      (parks_state(b.one, deal, pair) && !(tally && tally_full_walk)) ? park_bytes(all_metal) : 0u);
    if (view_instance) {
        if (first.which == 1 || first.which == 2)
            hipLaunchKernelGGL(trace_stack_view_batch_kernel<true>, b.grid, b.block, b.lds_bytes, stream, sc, d_frames, out, frame_stride, stack_levels);
        else
            hipLaunchKernelGGL(trace_stack_view_batch_kernel<false>, b.grid, b.block, b.lds_bytes, stream, sc, d_frames, out, frame_stride, stack_levels);
    } else if (pair)
        launch_stack_batch_pair(sc, b, tally, tally_full_walk);
    else if (tally)
        launch_stack_batch_tally(sc, b, tally);
    else
        launch_stack_batch_timed(sc, b);
    return hipGetLastError();
}

// One frame per launch WITH per-ray work tallies: the counting twins (plain frames: one lane per pixel, every shadow ray
// walked to its end -- the reference's traversals, equal to the CPU oracle's counters) and the debug views.  Launches
// without tallies never come here: they are batches of one (capi.hip: launch -> launch_stack_views).
hipError_t launch_stack(const SceneView &sc, const FrameView &fr, float4 *out, DeviceCounters *counters,
                        hipStream_t stream, int stack_levels)
{
    const dim3 grid(fr.total_patches), block(kBlock);
    const size_t lds_bytes = stack_lds_bytes(stack_levels, kBlock, plain_view(fr) && caches_leaves(false));
#define SHRAY_LAUNCH(K) hipLaunchKernelGGL((K), grid, block, lds_bytes, stream, sc, fr, out, counters, stack_levels)
    if (!counters)
        return hipErrorInvalidValue;
    if (!plain_view(fr)) {
        if (fr.which == 1 || fr.which == 2)   // only those views need the ray differentials carried along
            SHRAY_LAUNCH((trace_stack_view_kernel<true, true>));
        else
            SHRAY_LAUNCH((trace_stack_view_kernel<true, false>));
    }
#ifdef SHRAY_DIAGNOSTICS
    else if (g_diag_plain_kernel)
        SHRAY_LAUNCH((trace_stack_kernel<false, false, false>));
#endif
    else
        SHRAY_LAUNCH((trace_stack_kernel<true, false, false>));
#undef SHRAY_LAUNCH
    return hipGetLastError();
}

}   // namespace shray
