// kernel_pool.hip -- kernel id 2: the workgroup's rays as a pool whose waves merge while they traverse
// (pool_traversal.h).  Same pixels, same patches, same per-ray arithmetic as the stack kernel (id 0);
// what differs is the control flow around the traversal, which has to be uniform over the workgroup:
// every thread runs every bounce of every sample and enters each traversal together with the others
// (has_ray says whether its pixel still carries a ray), because the traversal synchronises the four
// waves.  trace() of raytracer.es.fs:552-582 is restated in that form below; the per-lane statements are
// the ones of trace_common.h: trace_ray.
//
// LDS per workgroup: max(levels, 4) x 1 KB of stack columns + 5 KB exchange buffer + 32 B of counts.
#include "launch.h"
#include "pool_traversal.h"

namespace shray {

constexpr int kPoolBlock = 256;
#ifndef SHRAY_POOL_MIN_WAVES
#define SHRAY_POOL_MIN_WAVES 5
#endif

template <bool COUNT, bool ONE_SAMPLE, bool METAL>
__device__ __forceinline__ void trace_pixels_pooled(const SceneView &sc, const FrameView &fr, float4 *__restrict__ out,
                                                    DeviceCounters *counters, PoolTraversal<kPoolBlock> &pool)
{
    int px, py;
    size_t out_index;
    bool store, inside;
    locate_pixel(fr, blockIdx.x, px, py, out_index, store, inside);

    RayCounters rc = {0, 0, 0, 0, 0, 0, 0};
    const V3 light = mk(fr.light_dir[0], fr.light_dir[1], fr.light_dir[2]);
    const V3 spec = mk(fr.specular_color[0], fr.specular_color[1], fr.specular_color[2]);
    const V3 diff = mk(fr.diffuse_color[0], fr.diffuse_color[1], fr.diffuse_color[2]);
    const bool has_diffuse = !METAL && diff.x > 0.0f && diff.y > 0.0f && diff.z > 0.0f;   // fs:570, uniform
    const float fw = (float)fr.width, fh = (float)fr.height, fn = (float)fr.spp;
    const int samples = ONE_SAMPLE ? 1 : fr.spp;

    V3 sum = mk(0, 0, 0);
    for (int s = 0; s < samples; s++) {
        // primary ray (vs:39-60, fs:619), sub-pixel pattern of the oracle
        const float ox = ((float)s + 0.5f) / fn;
        const float oy = (float)__brev((unsigned int)s) * 2.3283064365386963e-10f + 0.5f / fn;
        const float u = ((float)px + ox) / fw;
        const float v = ((float)py + oy) / fh;
        const V3 eye = unit(mk(fr.image_plane_width * (u - 0.5f), fr.image_plane_width * (v - 0.5f) * fr.aspect, -1.0f));
        V3 P = xform(fr.camera_matrix, mk(0, 0, 0), 1.0f);
        V3 D = unit(xform(fr.camera_normal_matrix, eye, 0.0f));

        V3 accumulated = mk(0, 0, 0), modulation = mk(1, 1, 1);
        bool alive = inside;      // still inside trace()'s bounce loop
        bool marker = false;      // returned the bad-hit colour (fs:566-568): no environment term
        for (int bounce = 0; bounce < fr.bounce_count; bounce++) {
            Hit hit{kFar, -1.0f, 0.0f, 0.0f};
            const int traced = pool.template closest<COUNT>(sc, fr, alive, xform(fr.object_matrix, P, 1.0f),
                                                            xform(fr.object_normal_matrix, D, 0.0f), hit, rc);
            if (traced == 0)
                break;            // uniform: no thread of the workgroup has a ray left
            bool shade = alive;
            if (alive && hit.t >= kFar) {
                alive = false;
                shade = false;
            }
            if (alive && hit.t == -1.0f) {
                if (COUNT)
                    rc.bad_hits++;
                marker = true;
                alive = false;
                shade = false;
            }
            V3 n = mk(0, 0, 0), R = D, P2 = P;
            V3 object_specular = mk(0, 0, 0);
            if (shade) {
                if (COUNT)
                    rc.shaded_hits++;
                const V3 object_normal = interpolated_normal(sc, fr.normals_fp16 != 0, hit.which, hit.bu, hit.bv);
                n = xform(fr.object_normal_inverse, object_normal, 0.0f);
                if (dot3(n, D) > 0.0f)
                    n = n * -1.0f;
                const V3 at = P + D * hit.t;                      // ray_transfer, fs:69
                R = D - n * (2.0f * dot3(n, D));                  // reflect(), fs:86
                P2 = at + n * .0001f;                             // surface fudge, fs:87
                const float fresnel = pow5(dot3(D, R) * .5f + .5f);
                object_specular = spec + (mk(1.0f, 1.0f, 1.0f) - spec) * fresnel;   // f_schlick_vr, fs:479-482
            }
            if (has_diffuse) {                                    // uniform
                bool lit = true;
                if (fr.cast_shadows) {                            // uniform
                    Hit shadow{kFar, -1.0f, 0.0f, 0.0f};
                    pool.template closest<COUNT>(sc, fr, shade, xform(fr.object_matrix, P2, 1.0f),
                                                 xform(fr.object_normal_matrix, light, 0.0f), shadow, rc);
                    lit = shadow.t >= kFar;
                }
                if (shade) {
                    const float lcos = sel_max(0.0f, dot3(n, light));
                    V3 irradiance = mk(0, 0, 0);
                    if (lit)
                        irradiance = irradiance + mk(1.0f, 1.0f, 1.0f) * lcos;
                    accumulated = accumulated + modulation * diff * irradiance;
                }
            }
            if (shade) {
                modulation = modulation * object_specular;
                P = P2;
                D = R;
            }
        }
        V3 radiance = mk(1.0f, 0.0f, 0.0f);
        if (inside && !marker) {
            if (COUNT)
                rc.env_lookups++;
            radiance = accumulated + modulation * environment(sc, D);
        }
        sum = (ONE_SAMPLE || fr.spp == 1) ? radiance : sum + radiance;
    }
    V3 result = (ONE_SAMPLE || fr.spp == 1) ? sum : sum / fn;
    if (fr.tonemap)
        result = mk(filmic(result.x), filmic(result.y), filmic(result.z));
    if (store)
        out[out_index] = inside ? make_float4(result.x, result.y, result.z, 1.0f) : make_float4(0, 0, 0, 0);
    if (COUNT)
        add_counters(rc, counters);
}

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
    trace_pixels_pooled<COUNT, ONE_SAMPLE, METAL>(sc, fr, out, counters, pool);
}

// batch form: workgroup (x, y) renders patch x of frame y (as kernel_stack.hip's)
template <bool ONE_SAMPLE, bool METAL>
__global__ void __launch_bounds__(kPoolBlock, SHRAY_POOL_MIN_WAVES) trace_pool_batch_kernel(SceneView sc, const FrameView *__restrict__ frames,
                                                                                          float4 *out, size_t frame_stride, int levels)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t lds_pool[];
    PoolTraversal<kPoolBlock> pool = make_pool(lds_pool, levels);
    trace_pixels_pooled<false, ONE_SAMPLE, METAL>(sc, frames[blockIdx.y], out + (size_t)blockIdx.y * frame_stride, nullptr, pool);
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
