// kernel_stack.hip -- kernel id 0: per-ray LDS stack over the packed BVH (stack_traversal.h).
//
// LDS: BLOCK * stack_levels * 4 bytes of dynamic shared memory; stack_levels
// is the tree's depth (computed at scene creation), so the bunny-class tree
// (depth 20) costs 20 KB per 256-thread workgroup and leaves room for 8
// workgroups (32 waves) per CU.
#include "launch.h"
#include "stack_traversal.h"

namespace shray {

constexpr int kBlock = 256;
#ifdef SHRAY_DIAGNOSTICS
bool g_diag_plain_kernel = false;
#endif

// Waves per SIMD the register allocator must leave room for.  Left alone, the plain kernel sits at the
// 128-register boundary (arch VGPRs + the AGPRs that hold spilled SGPRs): a few registers more and only
// three waves fit.  Asking for six (<= 85 registers) spills more to scratch and wins it back with
// occupancy -- with the plain one-triangle / one-visit loops of wave_traversal.h, which need fewer
// registers: one frame 0.58 -> 0.558 ms, pipelined 0.389 -> 0.357 ms, 1M-triangle scene 2.04 -> 1.77 ms.
// Seven and eight lose again on the benchmark frame (profiles/ab_sweep.sh).
#ifndef SHRAY_MIN_WAVES
#define SHRAY_MIN_WAVES 7
#endif
// the instances with the diffuse / shadow-ray branch carry more state: one wave fewer
#ifndef SHRAY_MIN_WAVES_GENERAL
#define SHRAY_MIN_WAVES_GENERAL 5
#endif
// launch the spp == 1 / metal instances of the plain kernel where they apply
#ifndef SHRAY_SPECIALIZE
#define SHRAY_SPECIALIZE 1
#endif
#ifndef SHRAY_LDS_PAD
#define SHRAY_LDS_PAD 0
#endif

template <bool COUNT, bool DIFF, bool ONE_SAMPLE = false, bool METAL = false>
__global__ void __launch_bounds__(kBlock, METAL ? SHRAY_MIN_WAVES : SHRAY_MIN_WAVES_GENERAL) trace_stack_kernel(SceneView sc, FrameView fr, float4 *out, DeviceCounters *counters)
{
    extern __shared__ uint32_t lds_stack[];
    StackTraversal<kBlock> trav;
    trav.stack = lds_stack + threadIdx.x;
    trace_pixels<StackTraversal<kBlock>, COUNT, DIFF, ONE_SAMPLE, METAL>(sc, fr, out, counters, trav);
}

// spp == 1 and a zero diffuse colour: the instances without a sample loop / without the diffuse branch
static bool one_sample(const FrameView &fr) { return fr.spp == 1; }
static bool metal(const FrameView &fr)
{
    return !(fr.diffuse_color[0] > 0.0f && fr.diffuse_color[1] > 0.0f && fr.diffuse_color[2] > 0.0f);
}

// Batch form: workgroup (x, y) renders patch x of frame y.  Workgroups are dispatched x-fastest, so
// frame 0 starts first and later frames fill the SIMDs its long-running waves leave idle.
template <bool DIFF, bool ONE_SAMPLE, bool METAL>
__global__ void __launch_bounds__(kBlock, METAL ? SHRAY_MIN_WAVES : SHRAY_MIN_WAVES_GENERAL) trace_stack_batch_kernel(SceneView sc, const FrameView *__restrict__ frames,
                                                                                   float4 *out, size_t frame_stride)
{
    extern __shared__ uint32_t lds_stack[];
    StackTraversal<kBlock> trav;
    trav.stack = lds_stack + threadIdx.x;
    trace_pixels<StackTraversal<kBlock>, false, DIFF, ONE_SAMPLE, METAL>(sc, frames[blockIdx.y], out + (size_t)blockIdx.y * frame_stride,
                                                                         nullptr, trav);
}

// `all_metal`: every frame of the batch has a zero diffuse colour
hipError_t launch_stack_batch(const SceneView &sc, const FrameView *d_frames, int count, const FrameView &first, bool all_metal,
                              float4 *out, size_t frame_stride, hipStream_t stream, int stack_levels)
{
    const dim3 grid(first.total_patches, (unsigned)count), block(kBlock);
    const size_t lds_bytes = (size_t)kBlock * (size_t)stack_levels * sizeof(uint32_t) + SHRAY_LDS_PAD;
    const bool one = SHRAY_SPECIALIZE && one_sample(first), metallic = SHRAY_SPECIALIZE && all_metal;
#define SHRAY_LAUNCH_BATCH(D, O, M) \
    hipLaunchKernelGGL((trace_stack_batch_kernel<D, O, M>), grid, block, lds_bytes, stream, sc, d_frames, out, frame_stride)
    if (first.which == 1 || first.which == 2)
        SHRAY_LAUNCH_BATCH(true, false, false);
    else if (one && metallic)
        SHRAY_LAUNCH_BATCH(false, true, true);
    else if (one)
        SHRAY_LAUNCH_BATCH(false, true, false);
    else if (metallic)
        SHRAY_LAUNCH_BATCH(false, false, true);
    else
        SHRAY_LAUNCH_BATCH(false, false, false);
#undef SHRAY_LAUNCH_BATCH
    return hipGetLastError();
}

hipError_t launch_stack(const SceneView &sc, const FrameView &fr, float4 *out, DeviceCounters *counters,
                        hipStream_t stream, int stack_levels)
{
    const dim3 grid(fr.total_patches), block(kBlock);
    const size_t lds_bytes = (size_t)kBlock * (size_t)stack_levels * sizeof(uint32_t) + SHRAY_LDS_PAD;
#ifdef SHRAY_DIAGNOSTICS
    if (counters && g_diag_plain_kernel) {
        hipLaunchKernelGGL((trace_stack_kernel<false, false>), grid, block, lds_bytes, stream, sc, fr, out, counters);
        return hipGetLastError();
    }
#endif
    const bool diff = fr.which == 1 || fr.which == 2;   // only those views need the ray differentials carried along
    if (counters && diff)
        hipLaunchKernelGGL((trace_stack_kernel<true, true>), grid, block, lds_bytes, stream, sc, fr, out, counters);
    else if (counters)
        hipLaunchKernelGGL((trace_stack_kernel<true, false>), grid, block, lds_bytes, stream, sc, fr, out, counters);
    else if (diff)
        hipLaunchKernelGGL((trace_stack_kernel<false, true>), grid, block, lds_bytes, stream, sc, fr, out, counters);
    else if (SHRAY_SPECIALIZE && one_sample(fr) && metal(fr))
        hipLaunchKernelGGL((trace_stack_kernel<false, false, true, true>), grid, block, lds_bytes, stream, sc, fr, out, counters);
    else if (SHRAY_SPECIALIZE && one_sample(fr))
        hipLaunchKernelGGL((trace_stack_kernel<false, false, true, false>), grid, block, lds_bytes, stream, sc, fr, out, counters);
    else if (SHRAY_SPECIALIZE && metal(fr))
        hipLaunchKernelGGL((trace_stack_kernel<false, false, false, true>), grid, block, lds_bytes, stream, sc, fr, out, counters);
    else
        hipLaunchKernelGGL((trace_stack_kernel<false, false>), grid, block, lds_bytes, stream, sc, fr, out, counters);
    return hipGetLastError();
}

}   // namespace shray
