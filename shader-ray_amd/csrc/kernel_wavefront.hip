// kernel_wavefront.hip -- kernel id 4: trace() of raytracer.es.fs:552-582 split by bounce (a "wavefront" form).
//
// One launch per bounce.  Bounce 0 generates the primary rays (one lane per pixel-sample, 8x8 pixel tiles per wave,
// like the other kernels); every bounce runs ONE closest-hit traversal per live path (plus the shadow traversal of
// fs:447-472 for diffuse materials), shades, and either finishes the path -- its radiance goes to the sample's slot
// -- or appends its state (64 bytes: P, D, modulation, accumulated, pixel, sample) to the next bounce's queue:
// wave-aggregated, one atomic per wave, so that the next launch runs on full waves of live paths only.  A last launch
// adds a pixel's samples in order (fs:622-636), divides and tone-maps.  Per-path arithmetic is trace_common.h's and
// uniform_driver.h's, statement for statement: frames are bit-identical to the other kernels'.
//
// Built as the measured answer to "would a bounce-split form with compaction between launches shorten a lone frame or
// the 1M-triangle scene?" (profiles/EXPERIMENTS.md R3.3); selectable with shray_scene_set_kernel(scene, 4), whole
// frames of the plain view only (tile sets, debug views and counters go to kernel 0's instances).
#include "launch.h"
#include "stack_traversal.h"

namespace shray {

constexpr int kWaveBlock = 64;

struct alignas(16) PathState {       // 64 bytes
    float P[3], Dx;
    float Dyz[2], modulation_xy[2];
    float modulation_z, accumulated[3];
    uint32_t pixel, sample, pad0, pad1;
};
static_assert(sizeof(PathState) == 64, "PathState must be 64 bytes");

// queue header in device memory: counts[b] = paths queued for bounce b
struct WavefrontQueues {
    PathState *queue[2];
    unsigned int *counts;      // [bounces + 1]
    float4 *radiance;          // per pixel-sample (spp > 1) -- or the frame itself (spp == 1, tone-mapped on the way)
};

template <bool METAL>
__global__ void __launch_bounds__(kWaveBlock, 6) wavefront_bounce_kernel(SceneView sc, const FrameView *__restrict__ frames, WavefrontQueues q,
                                                                          int bounce, int stack_levels, float4 *__restrict__ out)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t lds_stack[];
    const FrameView &fr = frames[0];
    StackTraversal<kWaveBlock, true> trav;
    trav.stack = lds_stack + threadIdx.x;
    trav.ids = reinterpret_cast<uint8_t *>(lds_stack + (size_t)stack_levels * kWaveBlock);
    RayCounters rc = {0, 0, 0, 0, 0, 0, 0};

    const V3 light = mk(fr.light_dir[0], fr.light_dir[1], fr.light_dir[2]);
    const V3 spec = mk(fr.specular_color[0], fr.specular_color[1], fr.specular_color[2]);
    const V3 diff = mk(fr.diffuse_color[0], fr.diffuse_color[1], fr.diffuse_color[2]);
    const bool has_diffuse = !METAL && diff.x > 0.0f && diff.y > 0.0f && diff.z > 0.0f;
    const float fn = (float)fr.spp;

    V3 P, D, modulation = mk(1, 1, 1), accumulated = mk(0, 0, 0);
    uint32_t pixel = 0, sample = 0;
    bool alive;
    if (bounce == 0) {
        // primary rays: workgroup b = (sample, patch, wave-of-patch) as a one-wave 8x8 tile (vs:39-60, fs:619)
        const unsigned int waves_per_sample = fr.total_patches * 4u;
        const unsigned int s = blockIdx.x / waves_per_sample, w = blockIdx.x % waves_per_sample;
        int px, py;
        size_t out_index;
        bool store, inside;
        locate_pixel(fr, w >> 2, px, py, out_index, store, inside, w & 3u);
        alive = inside;
        pixel = (uint32_t)out_index;
        sample = s;
        const float ox = ((float)s + 0.5f) / fn;
        const float oy = (float)__brev(s) * 2.3283064365386963e-10f + 0.5f / fn;
        const float u = ((float)px + ox) / (float)fr.width;
        const float v = ((float)py + oy) / (float)fr.height;
        const V3 eye = unit(mk(fr.image_plane_width * (u - 0.5f), fr.image_plane_width * (v - 0.5f) * fr.aspect, -1.0f));
        P = xform(fr.camera_matrix, mk(0, 0, 0), 1.0f);
        D = unit(xform(fr.camera_normal_matrix, eye, 0.0f));
    } else {
        const unsigned int i = blockIdx.x * kWaveBlock + threadIdx.x;
        const unsigned int count = q.counts[bounce];
        if (blockIdx.x * kWaveBlock >= count)
            return;     // the whole wave is beyond the queue
        alive = i < count;
        const PathState &st = q.queue[bounce & 1][alive ? i : 0];
        const float4 a = reinterpret_cast<const float4 *>(&st)[0], b = reinterpret_cast<const float4 *>(&st)[1],
                     c = reinterpret_cast<const float4 *>(&st)[2], d = reinterpret_cast<const float4 *>(&st)[3];
        P = mk(a.x, a.y, a.z);
        D = mk(a.w, b.x, b.y);
        modulation = mk(b.z, b.w, c.x);
        accumulated = mk(c.y, c.z, c.w);
        pixel = __float_as_uint(d.x);
        sample = __float_as_uint(d.y);
    }

    // ---- one iteration of trace()'s loop (uniform_driver.h's statements); bounce_count == 0: the loop does not run at all
    const bool in_loop = bounce < fr.bounce_count;
    Hit hit{kFar, -1.0f, 0.0f, 0.0f};
    if (in_loop)
        trav.template closest<false>(sc, fr, alive, xform(fr.object_matrix, P, 1.0f), xform(fr.object_normal_matrix, D, 0.0f), hit, rc);
    bool shade = alive && in_loop, marker = false, ended = alive && !in_loop;
    if (alive && hit.t >= kFar) {
        shade = false;
        ended = true;
    }
    if (alive && hit.t == -1.0f) {
        marker = true;
        shade = false;
        ended = true;
    }
    V3 n = mk(0, 0, 0), R = D, P2 = P, object_specular = mk(0, 0, 0);
    if (shade) {
        const ShadedHit sh = shade_hit(sc, fr, spec, P, D, hit);
        n = sh.n;
        R = sh.R;
        P2 = sh.P2;
        object_specular = sh.object_specular;
    }
    if (has_diffuse && in_loop) {
        bool lit = true;
        if (fr.cast_shadows) {
            Hit shadow{kFar, -1.0f, 0.0f, 0.0f};
            trav.template closest<false, true>(sc, fr, shade, xform(fr.object_matrix, P2, 1.0f), xform(fr.object_normal_matrix, light, 0.0f),
                                               shadow, rc);
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
        if (bounce + 1 >= fr.bounce_count)
            ended = true;       // the loop is over: the environment term follows (fs:580)
    }

    // ---- finished paths: the sample's radiance
    if (alive && ended) {
        V3 radiance = mk(1.0f, 0.0f, 0.0f);
        if (!marker)
            radiance = accumulated + modulation * environment(sc, D);
        if (fr.spp == 1) {
            if (fr.tonemap)
                radiance = mk(filmic(radiance.x), filmic(radiance.y), filmic(radiance.z));
            out[pixel] = make_float4(radiance.x, radiance.y, radiance.z, 1.0f);
        } else
            q.radiance[(size_t)pixel * (size_t)fr.spp + sample] = make_float4(radiance.x, radiance.y, radiance.z, 0.0f);
    }
    // ---- continuing paths: append to the next bounce's queue, one atomic per wave
    const bool go_on = alive && !ended;
    const unsigned long long movers = wave_ballot(go_on);
    if (movers) {
        const unsigned int lane = threadIdx.x & 63u;
        unsigned int base = 0;
        if (lane == (unsigned int)__builtin_ctzll(movers))
            base = atomicAdd(&q.counts[bounce + 1], (unsigned int)__popcll(movers));
        base = __shfl(base, __builtin_ctzll(movers), 64);
        if (go_on) {
            const unsigned int rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(movers >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)movers, 0u));
            float4 *dst = reinterpret_cast<float4 *>(&q.queue[(bounce + 1) & 1][base + rank]);
            dst[0] = make_float4(P.x, P.y, P.z, D.x);
            dst[1] = make_float4(D.y, D.z, modulation.x, modulation.y);
            dst[2] = make_float4(modulation.z, accumulated.x, accumulated.y, accumulated.z);
            dst[3] = make_float4(__uint_as_float(pixel), __uint_as_float(sample), 0.0f, 0.0f);
        }
    }
}

// spp > 1: a pixel's samples added in order, divided, tone-mapped once (fs:622-640)
__global__ void __launch_bounds__(256) wavefront_resolve_kernel(const FrameView *__restrict__ frames, const float4 *__restrict__ radiance,
                                                                float4 *__restrict__ out)
{
    const FrameView &fr = frames[0];
    const unsigned int p = blockIdx.x * 256u + threadIdx.x;
    if (p >= (unsigned int)(fr.width * fr.height))
        return;
    V3 sum = mk(0, 0, 0);
    for (int s = 0; s < fr.spp; s++) {
        const float4 r = radiance[(size_t)p * (size_t)fr.spp + s];
        sum = sum + mk(r.x, r.y, r.z);
    }
    V3 result = sum / (float)fr.spp;
    if (fr.tonemap)
        result = mk(filmic(result.x), filmic(result.y), filmic(result.z));
    out[p] = make_float4(result.x, result.y, result.z, 1.0f);
}

// d_view: the frame's view in device memory; scratch: queues / counts / radiance (capi.hip sizes them)
hipError_t launch_wavefront(const SceneView &sc, const FrameView *d_view, const FrameView &fr, bool metal, PathState *queue0, PathState *queue1,
                            unsigned int *counts, float4 *radiance, float4 *out, hipStream_t stream, int stack_levels)
{
    WavefrontQueues q;
    q.queue[0] = queue0;
    q.queue[1] = queue1;
    q.counts = counts;
    q.radiance = radiance;
    const size_t lds_bytes = (size_t)kWaveBlock * (size_t)stack_levels * sizeof(uint32_t) + kWaveBlock;
    const unsigned long long paths = (unsigned long long)fr.width * fr.height * fr.spp;
    hipError_t e = hipMemsetAsync(counts, 0, sizeof(unsigned int) * (size_t)(fr.bounce_count + 2), stream);
    if (e != hipSuccess)
        return e;
    for (int bounce = 0; bounce < (fr.bounce_count > 0 ? fr.bounce_count : 1); bounce++) {
        // bounce 0: every pixel-sample (whole 8x8 tiles); later bounces: at most every path, waves beyond the queue leave at once
        const unsigned int grid = bounce == 0 ? fr.total_patches * 4u * (unsigned int)fr.spp : (unsigned int)((paths + kWaveBlock - 1) / kWaveBlock);
        if (metal)
            hipLaunchKernelGGL((wavefront_bounce_kernel<true>), dim3(grid), dim3(kWaveBlock), lds_bytes, stream, sc, d_view, q, bounce, stack_levels, out);
        else
            hipLaunchKernelGGL((wavefront_bounce_kernel<false>), dim3(grid), dim3(kWaveBlock), lds_bytes, stream, sc, d_view, q, bounce, stack_levels, out);
    }
    if (fr.spp > 1)
        hipLaunchKernelGGL(wavefront_resolve_kernel, dim3((unsigned int)(((size_t)fr.width * fr.height + 255) / 256)), dim3(256), 0, stream, d_view,
                           radiance, out);
    return hipGetLastError();
}

}   // namespace shray
