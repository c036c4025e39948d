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
#define SHRAY_MIN_WAVES 6
#endif
#ifndef SHRAY_LDS_PAD
#define SHRAY_LDS_PAD 0
#endif

template <bool COUNT, bool DIFF>
__global__ void __launch_bounds__(kBlock, SHRAY_MIN_WAVES) trace_stack_kernel(SceneView sc, FrameView fr, float4 *out, DeviceCounters *counters)
{
    extern __shared__ uint32_t lds_stack[];
    StackTraversal<kBlock> trav;
    trav.stack = lds_stack + threadIdx.x;
    trace_pixels<StackTraversal<kBlock>, COUNT, DIFF>(sc, fr, out, counters, trav);
}

// Batch form: workgroup (x, y) renders patch x of frame y.  Workgroups are dispatched x-fastest, so
// frame 0 starts first and later frames fill the SIMDs its long-running waves leave idle.
template <bool DIFF>
__global__ void __launch_bounds__(kBlock, SHRAY_MIN_WAVES) trace_stack_batch_kernel(SceneView sc, const FrameView *__restrict__ frames,
                                                                                   float4 *out, size_t frame_stride)
{
    extern __shared__ uint32_t lds_stack[];
    StackTraversal<kBlock> trav;
    trav.stack = lds_stack + threadIdx.x;
    trace_pixels<StackTraversal<kBlock>, false, DIFF>(sc, frames[blockIdx.y], out + (size_t)blockIdx.y * frame_stride, nullptr, trav);
}

hipError_t launch_stack_batch(const SceneView &sc, const FrameView *d_frames, int count, const FrameView &first,
                              float4 *out, size_t frame_stride, hipStream_t stream, int stack_levels)
{
    const dim3 grid(first.total_patches, (unsigned)count), block(kBlock);
    const size_t lds_bytes = (size_t)kBlock * (size_t)stack_levels * sizeof(uint32_t) + SHRAY_LDS_PAD;
    if (first.which == 1 || first.which == 2)
        hipLaunchKernelGGL((trace_stack_batch_kernel<true>), grid, block, lds_bytes, stream, sc, d_frames, out, frame_stride);
    else
        hipLaunchKernelGGL((trace_stack_batch_kernel<false>), grid, block, lds_bytes, stream, sc, d_frames, out, frame_stride);
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
    else
        hipLaunchKernelGGL((trace_stack_kernel<false, false>), grid, block, lds_bytes, stream, sc, fr, out, counters);
    return hipGetLastError();
}

}   // namespace shray
