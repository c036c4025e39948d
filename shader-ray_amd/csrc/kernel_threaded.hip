// kernel_threaded.hip -- kernel id 1: literal threaded traversal (threaded_traversal.h).
#include "launch.h"
#include "threaded_traversal.h"

namespace shray {

template <bool COUNT, bool DIFF>
__global__ void __launch_bounds__(256) trace_threaded_kernel(SceneView sc, FrameView fr, float4 *out, DeviceCounters *counters)
{
    ThreadedTraversal trav;
    trace_pixels<ThreadedTraversal, COUNT, DIFF>(sc, fr, out, counters, trav);
}

hipError_t launch_threaded(const SceneView &sc, const FrameView &fr, float4 *out, DeviceCounters *counters,
                           hipStream_t stream)
{
    const dim3 grid(fr.total_patches), block(256);
    const bool diff = fr.which == 1 || fr.which == 2;
    if (counters && diff)
        hipLaunchKernelGGL((trace_threaded_kernel<true, true>), grid, block, 0, stream, sc, fr, out, counters);
    else if (counters)
        hipLaunchKernelGGL((trace_threaded_kernel<true, false>), grid, block, 0, stream, sc, fr, out, counters);
    else if (diff)
        hipLaunchKernelGGL((trace_threaded_kernel<false, true>), grid, block, 0, stream, sc, fr, out, counters);
    else
        hipLaunchKernelGGL((trace_threaded_kernel<false, false>), grid, block, 0, stream, sc, fr, out, counters);
    return hipGetLastError();
}

}   // namespace shray
