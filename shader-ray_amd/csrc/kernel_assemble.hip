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

}   // namespace shray
