// kernel_assemble.hip -- rank 0's de-interleave of gathered tile buffers into whole frames.
//
// Input, as the gather leaves it:  gathered[rank][frame][tile k of that rank][tile_h][tile_w][C]
// with C = 3 (R, G, B on the wire; alpha is the constant 1, raytracer.es.fs:676) or 4; tile t
// of a frame (row-major over the frame) has phase t % period, period = c0 + (world - 1) * c1; rank 0 owns
// phases [0, c0), rank r >= 1 phases [c0 + (r - 1) c1, c0 + r c1); it is its owner's tile
// (t / period) * (phases owned) + (phase - first owned phase).  c0 = c1 = 1 is the even split.
// Output: out[frame][height][width] RGBA, row 0 = bottom.
//
// One thread per output pixel: the 16-byte stores are fully coalesced; the loads are
// contiguous runs of tile_w pixels.  HBM-bound: (4 * C + 16) bytes per pixel.
#include "launch.h"

namespace shray {

template <int C>
__global__ void __launch_bounds__(256) assemble_tiles_kernel(const float *__restrict__ gathered, float4 *__restrict__ out,
                                                             int period, int c0, int c1, int frames, int width, int height,
                                                             int tile_w, int tile_h, int tiles_x, size_t rank_stride,
                                                             size_t frame_stride)
{
    const int x = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    const int y = (int)blockIdx.y, f = (int)blockIdx.z;
    if (x >= width)
        return;
    const int tx = x / tile_w, ty = y / tile_h;
    const int t = ty * tiles_x + tx;
    const int phase = t % period, round = t / period;
    const int rank = phase < c0 ? 0 : 1 + (phase - c0) / c1;
    const int k = phase < c0 ? round * c0 + phase : round * c1 + (phase - c0) % c1;
    const size_t pixel = ((size_t)k * tile_h + (size_t)(y - ty * tile_h)) * tile_w + (size_t)(x - tx * tile_w);
    const float *src = gathered + (size_t)rank * rank_stride + (size_t)f * frame_stride + pixel * C;
    float4 v;
    v.x = src[0];
    v.y = src[1];
    v.z = src[2];
    v.w = C == 4 ? src[C - 1] : 1.0f;
    out[((size_t)f * height + y) * width + x] = v;
}

hipError_t launch_assemble_tiles(const float *gathered, float4 *out, int world, int c0, int c1, int frames, int channels,
                                 int width, int height, int tile_w, int tile_h, size_t rank_stride, size_t frame_stride,
                                 hipStream_t stream)
{
    const int tiles_x = (width + tile_w - 1) / tile_w;
    const int period = c0 + (world - 1) * c1;
    const dim3 block(256), grid((unsigned)((width + 255) / 256), (unsigned)height, (unsigned)frames);
    if (channels == 3)
        hipLaunchKernelGGL((assemble_tiles_kernel<3>), grid, block, 0, stream, gathered, out, period, c0, c1, frames, width, height,
                           tile_w, tile_h, tiles_x, rank_stride, frame_stride);
    else
        hipLaunchKernelGGL((assemble_tiles_kernel<4>), grid, block, 0, stream, gathered, out, period, c0, c1, frames, width, height,
                           tile_w, tile_h, tiles_x, rank_stride, frame_stride);
    return hipGetLastError();
}


// ---------------------------------------------------------------------------------------------------
// launch_dispatch_order: a stable counting sort of the patches by cost class, heaviest class first.
#ifndef SHRAY_ORDER_CLASSES
#define SHRAY_ORDER_CLASSES 32
#endif
constexpr int kOrderClasses = SHRAY_ORDER_CLASSES, kOrderThreads = 4096 / kOrderClasses;   // 16 KB of class counts in LDS
constexpr uint32_t kOrderStaged = 8192;     // shapes of up to this many patches (a 1080p frame has 8,160) are sorted out of LDS

__global__ void __launch_bounds__(kOrderThreads) dispatch_order_kernel(uint32_t *__restrict__ live_cost, uint32_t *__restrict__ order, uint32_t n,
                                                                       uint32_t bulk_class)
{
    // render kernels keep reporting into live_cost while this runs: every pass below must see the SAME costs (or the
    // classes' counts and the scatter disagree and `order` is no permutation), so they are copied first -- into the n
    // words behind them -- and decay in place
    // -- in LDS when the shape fits (one coalesced read of the costs instead of three strided passes over them: 39 -> ~10 us
    // for a 1080p frame's patches), else in the n words behind them
    __shared__ uint32_t staged[kOrderStaged];
    uint32_t *__restrict__ cost = n <= kOrderStaged ? staged : live_cost + n;
    __shared__ uint32_t hist[kOrderClasses * kOrderThreads];   // hist[(class rank) * threads + thread], class rank 0 = heaviest
    __shared__ uint32_t partial[kOrderThreads];
    __shared__ uint32_t top;
    const uint32_t t = threadIdx.x;
    const uint32_t chunk = (n + kOrderThreads - 1) / kOrderThreads, first = t * chunk, last = min(n, first + chunk);
    // the largest cost
    uint32_t m = 0;
    if (n <= kOrderStaged) {
        for (uint32_t e = t; e < n; e += kOrderThreads) {      // coalesced
            const uint32_t v = live_cost[e];
            staged[e] = v;
            live_cost[e] = v >> 1;     // (a wave that reports meanwhile may be overwritten: the next frame reports again)
            m = max(m, v);
        }
    } else {
        for (uint32_t e = first; e < last; e++) {
            const uint32_t v = live_cost[e];
            cost[e] = v;
            live_cost[e] = v >> 1;
            m = max(m, v);
        }
    }
    partial[t] = m;
    __syncthreads();
    for (uint32_t step = kOrderThreads / 2; step > 0; step >>= 1) {
        if (t < step)
            partial[t] = max(partial[t], partial[t + step]);
        __syncthreads();
    }
    if (t == 0)
        top = partial[0];
    __syncthreads();
    const unsigned long long scale = (unsigned long long)top + 1ull;
    // this thread's patches per class
#pragma unroll
    for (int c = 0; c < kOrderClasses; c++)
        hist[c * kOrderThreads + t] = 0;
    for (uint32_t e = first; e < last; e++) {
        const uint32_t c = min(bulk_class, (kOrderClasses - 1) - (uint32_t)(((unsigned long long)cost[e] * kOrderClasses) / scale));
        hist[c * kOrderThreads + t]++;
    }
    __syncthreads();
    // exclusive scan over (class rank, thread): thread t owns kOrderClasses consecutive entries of the flattened table
    uint32_t sum = 0;
    for (int k = 0; k < kOrderClasses; k++)
        sum += hist[t * kOrderClasses + k];
    partial[t] = sum;
    __syncthreads();
    for (uint32_t step = 1; step < kOrderThreads; step <<= 1) {
        const uint32_t add = t >= step ? partial[t - step] : 0u;
        __syncthreads();
        partial[t] += add;
        __syncthreads();
    }
    uint32_t run = partial[t] - sum;
    for (int k = 0; k < kOrderClasses; k++) {
        const uint32_t h = hist[t * kOrderClasses + k];
        hist[t * kOrderClasses + k] = run;
        run += h;
    }
    __syncthreads();
    // scatter, in patch order inside a class
    for (uint32_t e = first; e < last; e++) {
        const uint32_t v = cost[e];
        const uint32_t c = min(bulk_class, (kOrderClasses - 1) - (uint32_t)(((unsigned long long)v * kOrderClasses) / scale));
        order[hist[c * kOrderThreads + t]++] = e;
    }
}

hipError_t launch_dispatch_order(uint32_t *cost, uint32_t *order, uint32_t n, hipStream_t stream, int bulk_class)
{
    if (n == 0)
        return hipSuccess;
    const uint32_t bulk = (uint32_t)(bulk_class < 0 ? 0 : (bulk_class > kOrderClasses - 1 ? kOrderClasses - 1 : bulk_class));
    hipLaunchKernelGGL(dispatch_order_kernel, dim3(1), dim3(kOrderThreads), 0, stream, cost, order, n, bulk);
    return hipGetLastError();
}

}   // namespace shray
