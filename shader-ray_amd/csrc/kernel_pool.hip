// kernel_pool.hip -- kernel id 2: the workgroup's rays as a pool whose waves merge while they traverse
// (pool_traversal.h).  Same pixels, same patches, same per-ray arithmetic as the stack kernel (id 0);
// what differs is the control flow around the traversal, which has to be uniform over the workgroup:
// every thread runs every bounce of every sample and enters each traversal together with the others
// (has_ray says whether its pixel still carries a ray), because the traversal synchronises the four
// waves.  trace() of raytracer.es.fs:552-582 in that form is uniform_driver.h.
//
// LDS per workgroup: max(levels, 4) x 1 KB of stack columns + 5 KB exchange buffer + 32 B of counts.
#include "launch.h"
#include "pool_traversal.h"
#include "uniform_driver.h"

namespace shray {

constexpr int kPoolBlock = 256;
#ifndef SHRAY_POOL_MIN_WAVES
#define SHRAY_POOL_MIN_WAVES 5
#endif

__device__ __forceinline__ PoolTraversal<kPoolBlock> make_pool(uint32_t *lds, int levels)
{
    PoolTraversal<kPoolBlock> pool;
    pool.stack = lds;
    pool.xbuf = lds + (size_t)levels * kPoolBlock;
    pool.counts = pool.xbuf + kPoolXbufDwords;
    return pool;
}

template <bool COUNT, bool ONE_SAMPLE, bool METAL>
__global__ void __launch_bounds__(kPoolBlock, SHRAY_POOL_MIN_WAVES) trace_pool_kernel(SceneView sc, FrameView fr, float4 *out,
                                                                                    DeviceCounters *counters, int levels)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t lds_pool[];
    PoolTraversal<kPoolBlock> pool = make_pool(lds_pool, levels);
    trace_pixels_uniform<PoolTraversal<kPoolBlock>, COUNT, ONE_SAMPLE, METAL>(sc, fr, out, counters, pool);
}

// batch form: workgroup (x, y) renders patch x of frame y (as kernel_stack.hip's)
template <bool ONE_SAMPLE, bool METAL>
__global__ void __launch_bounds__(kPoolBlock, SHRAY_POOL_MIN_WAVES) trace_pool_batch_kernel(SceneView sc, const FrameView *__restrict__ frames,
                                                                                          float4 *out, size_t frame_stride, int levels)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t lds_pool[];
    PoolTraversal<kPoolBlock> pool = make_pool(lds_pool, levels);
    trace_pixels_uniform<PoolTraversal<kPoolBlock>, false, ONE_SAMPLE, METAL>(sc, frames[blockIdx.y], out + (size_t)blockIdx.y * frame_stride, nullptr, pool);
}

static int pool_levels(int stack_levels) { return stack_levels < kPoolMinLevels ? kPoolMinLevels : stack_levels; }
static size_t pool_lds_bytes(int levels)
{
    return ((size_t)levels * kPoolBlock + kPoolXbufDwords + kPoolCountDwords) * sizeof(uint32_t);
}
static bool pool_metal(const FrameView &fr)
{
    return !(fr.diffuse_color[0] > 0.0f && fr.diffuse_color[1] > 0.0f && fr.diffuse_color[2] > 0.0f);
}

// which == 0 frames only (the launcher in capi.hip sends the shader's debug views to the stack kernel)
hipError_t launch_pool(const SceneView &sc, const FrameView &fr, float4 *out, DeviceCounters *counters, hipStream_t stream,
                       int stack_levels)
{
    const dim3 grid(fr.total_patches), block(kPoolBlock);
    const int levels = pool_levels(stack_levels);
    const size_t lds = pool_lds_bytes(levels);
    const bool one = fr.spp == 1, metal = pool_metal(fr);
#define SHRAY_LAUNCH_POOL(C, O, M) hipLaunchKernelGGL((trace_pool_kernel<C, O, M>), grid, block, lds, stream, sc, fr, out, counters, levels)
    if (counters)
        SHRAY_LAUNCH_POOL(true, false, false);
    else if (one && metal)
        SHRAY_LAUNCH_POOL(false, true, true);
    else if (one)
        SHRAY_LAUNCH_POOL(false, true, false);
    else if (metal)
        SHRAY_LAUNCH_POOL(false, false, true);
    else
        SHRAY_LAUNCH_POOL(false, false, false);
#undef SHRAY_LAUNCH_POOL
    return hipGetLastError();
}

hipError_t launch_pool_batch(const SceneView &sc, const FrameView *d_frames, int count, const FrameView &first, bool all_metal,
                             float4 *out, size_t frame_stride, hipStream_t stream, int stack_levels)
{
    const dim3 grid(first.total_patches, (unsigned)count), block(kPoolBlock);
    const int levels = pool_levels(stack_levels);
    const size_t lds = pool_lds_bytes(levels);
    const bool one = first.spp == 1;
#define SHRAY_LAUNCH_POOL_BATCH(O, M) \
    hipLaunchKernelGGL((trace_pool_batch_kernel<O, M>), grid, block, lds, stream, sc, d_frames, out, frame_stride, levels)
    if (one && all_metal)
        SHRAY_LAUNCH_POOL_BATCH(true, true);
    else if (one)
        SHRAY_LAUNCH_POOL_BATCH(true, false);
    else if (all_metal)
        SHRAY_LAUNCH_POOL_BATCH(false, true);
    else
        SHRAY_LAUNCH_POOL_BATCH(false, false);
#undef SHRAY_LAUNCH_POOL_BATCH
    return hipGetLastError();
}

}   // namespace shray
