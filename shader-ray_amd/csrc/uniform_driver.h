// uniform_driver.h -- trace() of raytracer.es.fs:552-582 with CONVERGENT control flow: every thread of the
// group that traverses together (a wave for the stack kernel, the workgroup for the pool kernel) runs every
// bounce of every sample and enters each traversal with the others; `has_ray` says whether its pixel still
// carries a ray.  The traversals need that: the stack kernel's dealt leaf stage (leaf_stage.h) uses the
// wave's idle lanes as workers, the pool kernel synchronises its four waves.  The per-lane statements are
// those of trace_common.h: trace_ray (same arithmetic, same order); only which == 0 frames come here.
//
// Traversal::closest(sc, fr, has_ray, P, D, hit, rc) -> number of rays traced by the group (uniform).
//
// Multi-sample frames in one-wave workgroups (SAMPLE LANES): with one lane per pixel a wave runs its 64 pixels' spp
// samples one after another, so the few heavy waves of a frame (the grazing silhouette) run spp times longer than
// the others and a lone frame is half tail (profiles/history/r02/leaf_stage_ab.txt section 22).  Here the G = 2^(x+y) <= 32
// samples of a pixel run in a block of 2^x by 2^y neighbouring lanes of the wave's 8x8 lane grid: a 16x16 patch is
// 4 G waves, a wave is 64 / G pixels x G samples -- the heavy pixels' work is spread over G lanes, and the 64 rays
// of a wave are closer together.  The shader adds a pixel's samples in order (fs:622-636: ((r0 + r1) + r2) + ...);
// so does this: each round of G samples is staged through the wave's idle stack columns and added, in sample order,
// by one lane per colour channel.
#pragma once

#include "trace_common.h"

// 1: the lane's pixel is located again wherever it is needed instead of carried across the traversals (A/B builds: 0)
#ifndef SHRAY_RECOMPUTE_PIXEL
#define SHRAY_RECOMPUTE_PIXEL 1
#endif

namespace shray {

// shadow rays stop at their first hit (stack_traversal.h: closest<COUNT, ANY_HIT>); 0 = walk them to the end

// COUNT: tally per-ray work.  The counting twins walk every shadow ray to its end and run one lane per pixel, so that
// their tallies equal the reference's full traversals (the oracle's); with TIMED_FORM they keep the timed instances'
// form instead -- sample lanes, shadow rays that stop at their first hit -- and tally what THOSE do
// (shray_render_counters_timed).
// ORDERED: the launch reads a dispatch order and its waves report their running times (capi.hip: DispatchOrder) -- an
// instance of its own, because the two scalars it carries through the kernel cost the others 2 % (12 B more scratch)
// unit() of the eye-space ray (x, y, -1): three quotients by one length.  components_in_range (uniform): see the caller; the
// length itself is checked here (a lane far outside the frame could have an absurd one); the wave takes the true divisions
// unless every lane qualifies -- the two forms give the same quotients wherever both apply.
__device__ __forceinline__ V3 unit_of_eye(V3 a, bool components_in_range)
{
    const float length = sqrtf(dot3(a, a));
    const bool exact = components_in_range && divisor_in_range(length);
#ifndef SHRAY_COST_MAIN_PATH     // profiles/isa_costs.hip counts the path every wave takes
    if (__builtin_expect(__builtin_amdgcn_ballot_w64(!exact) != 0ull, 0)) {
        asm volatile("; an eye ray outside exact_div.h's ranges" ::: "memory");   // keeps this a branch
        return a / length;
    }
#endif
    const float y = reciprocal_in_range(length), yl = reciprocal_residual(length, y);
    return mk(div_by_constant4(a.x, length, y, yl), div_by_constant4(a.y, length, y, yl), div_by_constant4(a.z, length, y, yl));
}

// The sample loop's `accumulated`, `modulation` and channel sums between traversals: registers.  (-DSHRAY_PARK=1 keeps them in the
// wave's spare LDS instead -- variants/parked_state.h: built, bit-identical, 1 % slower, R5.2 -- behind the same interface.)
#ifndef SHRAY_PARK
#define SHRAY_PARK 0
#endif
#if SHRAY_PARK
}   // namespace shray
#include "variants/parked_state.h"
namespace shray {
#else
template <bool PARK, bool METAL>
struct ParkedState {
    static_assert(!PARK, "the parked form is variants/parked_state.h (-DSHRAY_PARK=1)");
    float *slab;                                   // (the parked form's; kept so that both forms lay their members out alike)
    V3 product = mk(1, 1, 1), sum = mk(0, 0, 0);   // modulation, accumulated
    float channel = 0.0f, channel2 = 0.0f;
    __device__ __forceinline__ void begin(float *) { slab = nullptr; }
    __device__ __forceinline__ V3 modulation() const { return product; }
    __device__ __forceinline__ void set_modulation(V3 v) { product = v; }
    __device__ __forceinline__ V3 accumulated() const { return sum; }
    __device__ __forceinline__ void set_accumulated(V3 v) { sum = v; }
    __device__ __forceinline__ float channel_sum() const { return channel; }
    __device__ __forceinline__ void set_channel_sum(float v) { channel = v; }
    __device__ __forceinline__ float channel_sum2() const { return channel2; }
    __device__ __forceinline__ void set_channel_sum2(float v) { channel2 = v; }
};
#endif

template <class Traversal, bool COUNT, bool ONE_SAMPLE, bool METAL, bool TIMED_FORM = false, bool ORDERED = false, bool PARK = false>
__device__ __forceinline__ void trace_pixels_uniform(const SceneView &sc, const FrameView &fr, float4 *__restrict__ out,
                                                     DeviceCounters *counters, Traversal &pool,
                                                     unsigned int block_index = 0xffffffffu,   // default: blockIdx.x
                                                     float *park = nullptr)
{
    if (block_index == 0xffffffffu)
        block_index = blockIdx.x;
#ifdef SHRAY_DIAGNOSTICS
    // diagnostic build only (profiles/timeline.py): per-wave residency stamps, written to a buffer nothing else reads
    const unsigned long long t_begin = __builtin_amdgcn_s_memrealtime();
    const unsigned long long c_begin = __builtin_amdgcn_s_memtime();   // shader cycles: in-kernel clock = d(memtime) / d(memrealtime) x 100 MHz
#endif
    // Which pixel (and which of its samples) this lane renders.  Evaluated where it is needed -- in front of a round of samples
    // for the primary ray, behind the last traversal for the store -- instead of once at the top: px, py, the output index and the
    // lane's place in its pixel's block of sample lanes are a dozen instructions of integer arithmetic on the workgroup's index
    // and the lane's number, and carried across the traversals they are six vector registers the traversal spills (round 5).
    // The asm statements make each evaluation's inputs opaque, so that the compiler does not keep the first one's results.
    const bool sample_lanes = !ONE_SAMPLE && (!COUNT || TIMED_FORM) && Traversal::block_size == 64;
    const unsigned int log_gx = sample_lanes ? fr.sample_log_x : 0u, log_gy = sample_lanes ? fr.sample_log_y : 0u;
    const unsigned int G = 1u << (log_gx + log_gy);
    struct LanePixel {
        int px, py;
        size_t out_index;
        bool store, inside;
        unsigned int sub, base_lane;   // lanes per pixel G = gx * gy: this lane runs samples sub, sub + G, ... of its pixel;
                                       // base_lane = the pixel's lane with sub == 0
        uint32_t patch;
    };
    auto locate = [&](bool again) -> LanePixel {
        LanePixel lp;
        lp.sub = 0;
        lp.patch = 0;
        unsigned int b = block_index;
        // a 256-thread workgroup is a 16x16 patch (four 8x8 wave tiles); a 64-thread workgroup is one of those tiles
        if (Traversal::block_size == 64) {
            // (one-wave workgroups: the lane's number is threadIdx.x; recomputed, it does not keep v0 alive)
            unsigned int lane = (again && SHRAY_RECOMPUTE_PIXEL) ? __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)) : threadIdx.x;
            if (again && SHRAY_RECOMPUTE_PIXEL) {
                asm volatile("" : "+s"(b));
                asm volatile("" : "+v"(lane));
            }
            lp.base_lane = lane;
            // workgroups go to the eight XCDs round robin: keep the waves of a patch (four, or 4 G with sample lanes) on one
            // XCD (one L2), as the 256-thread form does.  The grid is rounded up to whole groups of 8 patches; the surplus
            // waves leave at once (the caller's test of `slot`).
            const unsigned int log_waves = 2u + log_gx + log_gy;
            const unsigned int k = b >> 3, slot = ((k >> log_waves) << 3) + (b & 7u), wave = k & ((1u << log_waves) - 1u);
            // heaviest patches first (capi.hip: DispatchOrder): which patch this slot of the launch renders
            lp.patch = slot;
            if (ORDERED)
                lp.patch = fr.dispatch_order ? fr.dispatch_order[slot] : slot;
            // the patch as (16 gx) x (16 gy) lane positions, cut into 8x8 wave tiles: position (vx, vy) is sample
            // (vy % gy) * gx + vx % gx of pixel (vx / gx, vy / gy)
            const unsigned int tiles_across = 2u << log_gx;
            const unsigned int vx = (wave & (tiles_across - 1u)) * 8u + (lane & 7u), vy = (wave >> (1u + log_gx)) * 8u + (lane >> 3);
            const unsigned int sx = vx & ((1u << log_gx) - 1u), sy = vy & ((1u << log_gy) - 1u);
            lp.sub = (sy << log_gx) | sx;
            lp.base_lane = lane - (sy * 8u + sx);
            locate_patch_pixel(fr, lp.patch, (int)(vx >> log_gx), (int)(vy >> log_gy), lp.px, lp.py, lp.out_index, lp.store, lp.inside);
        } else {
            lp.base_lane = threadIdx.x & 63u;
            locate_pixel(fr, b, lp.px, lp.py, lp.out_index, lp.store, lp.inside);
        }
        return lp;
    };
    uint32_t cost_begin = 0;   // ORDERED: when the wave began (the low half of the shader clock, a scalar carried through the kernel)
    if (Traversal::block_size == 64) {
        const unsigned int log_waves = 2u + log_gx + log_gy;
        const unsigned int k = block_index >> 3, slot = ((k >> log_waves) << 3) + (block_index & 7u);
        if (slot >= fr.total_patches)
            return;
        if (ORDERED && fr.dispatch_cost)
            cost_begin = (uint32_t)__builtin_amdgcn_s_memtime();
    }
    const LanePixel first = locate(false);

    RayCounters rc = {0, 0, 0, 0, 0, 0, 0};
    const V3 light = mk(fr.light_dir[0], fr.light_dir[1], fr.light_dir[2]);
    const V3 spec = mk(fr.specular_color[0], fr.specular_color[1], fr.specular_color[2]);
    const V3 diff = mk(fr.diffuse_color[0], fr.diffuse_color[1], fr.diffuse_color[2]);
    const bool has_diffuse = !METAL && diff.x > 0.0f && diff.y > 0.0f && diff.z > 0.0f;   // fs:570, uniform
    // (a ONE_SAMPLE instance is launched for spp == 1 only: its divisions by the sample count fold away)
    const float fw = (float)fr.width, fh = (float)fr.height, fn = ONE_SAMPLE ? 1.0f : (float)fr.spp;
    const int samples = ONE_SAMPLE ? 1 : fr.spp;
    // the pixel coordinates' divisions by the frame's width and height (vs:39-60): dividends px + ox, py + oy are 0 or of
    // magnitude in [2^-25, 2^25) -- a pixel index plus an offset in (0, 1] --, the divisors shared by the whole launch
    const SharedDivisor by_width = shared_divisor(fw), by_height = shared_divisor(fh);
    // the eye ray's normalization: its length is at least 1 (the -1 component) and its components are 0 or no smaller than
    // 2^-85 when the image plane's width and the aspect ratio are ordinary numbers (u - 0.5 is 0 or at least 2^-25)
    const bool eye_in_range = magnitude_in(fr.image_plane_width, -30, 8) && magnitude_in(fr.aspect, -30, 8);

    V3 sum = mk(0, 0, 0);
    // sample lanes: this lane's colour channel (lane 0 of a pair: also blue) -- and, in the same object, the sample's
    // `accumulated` and `modulation`: in registers, or parked in LDS
    ParkedState<PARK, METAL> kept;
    kept.begin(park);
    for (int s0 = 0; s0 < samples; s0 += (int)G) {
        const LanePixel at = s0 == 0 ? first : locate(true);
        const int s = s0 + (int)at.sub;
        const bool has_sample = at.inside && s < samples;
        // primary ray (vs:39-60, fs:619), sub-pixel pattern of the oracle
        const float ox = ((float)s + 0.5f) / fn;
        const float oy = (float)__brev((unsigned int)s) * 2.3283064365386963e-10f + 0.5f / fn;
        const float u = divide_by_shared((float)at.px + ox, by_width);
        const float v = divide_by_shared((float)at.py + oy, by_height);
        const V3 eye = unit_of_eye(mk(fr.image_plane_width * (u - 0.5f), fr.image_plane_width * (v - 0.5f) * fr.aspect, -1.0f), eye_in_range);
        V3 P = xform(fr.camera_matrix, mk(0, 0, 0), 1.0f);
        V3 D = unit(xform(fr.camera_normal_matrix, eye, 0.0f));

        kept.set_accumulated(mk(0, 0, 0));
        kept.set_modulation(mk(1, 1, 1));
        bool alive = has_sample;  // still inside trace()'s bounce loop
        bool marker = false;      // returned the bad-hit colour (fs:566-568): no environment term
        for (int bounce = 0; bounce < fr.bounce_count; bounce++) {
            Hit hit{kFar, -1.0f, 0.0f, 0.0f};
            const int traced = pool.template closest<COUNT>(sc, fr, alive, xform(fr.object_matrix, P, 1.0f),
                                                            xform(fr.object_normal_matrix, D, 0.0f), hit, rc);
            if (traced == 0)
                break;            // uniform: none of the threads that traverse together has a ray left
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
                const ShadedHit sh = shade_hit(sc, fr, spec, P, D, hit);
                n = sh.n;
                R = sh.R;
                P2 = sh.P2;
                object_specular = sh.object_specular;
            }
            if (has_diffuse) {                                    // uniform
                // What the diffuse term needs of this hit is formed in FRONT of the shadow traversal -- the cosine (one word
                // instead of the normal) and modulation * diff, the first product of fs:571's modulation * diffuse * irradiance
                // -- and the ray moves on to its next bounce there too: across the shadow traversal a lane then carries
                // accumulated, that product, the cosine and the next ray (16 words instead of 18 + the old ray)
                const float lcos = sel_max(0.0f, dot3(n, light));
                const V3 modulation = kept.modulation();
                const V3 md = modulation * diff;
                if (shade) {
                    kept.set_modulation(modulation * object_specular);
                    P = P2;
                    D = R;
                }
                bool lit = true;
                if (fr.cast_shadows) {                            // uniform
                    Hit shadow{kFar, -1.0f, 0.0f, 0.0f};
                    pool.template closest<COUNT, true, TIMED_FORM>(sc, fr, shade, xform(fr.object_matrix, P2, 1.0f),
                                                 xform(fr.object_normal_matrix, light, 0.0f), shadow, rc);
                    lit = shadow.t >= kFar;
                }
                if (shade) {
                    V3 irradiance = mk(0, 0, 0);
                    if (lit)
                        irradiance = irradiance + mk(1.0f, 1.0f, 1.0f) * lcos;
                    kept.set_accumulated(kept.accumulated() + md * irradiance);
                }
            } else if (shade) {
                kept.set_modulation(kept.modulation() * object_specular);
                P = P2;
                D = R;
            }
        }
        V3 radiance = mk(1.0f, 0.0f, 0.0f);
        if (has_sample && !marker) {
            if (COUNT)
                rc.env_lookups++;
            const V3 sky = environment(sc, D);
            radiance = kept.accumulated() + kept.modulation() * sky;
        }
        if (G == 1u) {
            sum = (ONE_SAMPLE || fr.spp == 1) ? radiance : sum + radiance;
        } else {
            // this round's G radiances of every pixel, through levels 0-2 of the wave's stack columns (the stacks are
            // empty between traversals), then added in sample order: lane `sub` of a pixel sums channel `sub`
            // (of a pair of lanes, lane 0 sums red and blue); the other lanes of the pixel add along, unused
            const LanePixel me = locate(true);
            uint32_t *lds = pool.stack - threadIdx.x;
            lds[threadIdx.x] = __float_as_uint(radiance.x);
            lds[64u + threadIdx.x] = __float_as_uint(radiance.y);
            lds[128u + threadIdx.x] = __float_as_uint(radiance.z);
            __syncthreads();   // one wave: orders the exchange for the compiler, costs nothing
            const unsigned int channel = me.sub < 3u ? me.sub : 2u;
            const int valid = min((int)G, samples - s0);
            float channel_sum = kept.channel_sum(), channel_sum2 = G == 2u ? kept.channel_sum2() : 0.0f;
            for (int k = 0; k < valid; k++) {
                const unsigned int src = me.base_lane + (((unsigned int)k >> log_gx) << 3) + ((unsigned int)k & ((1u << log_gx) - 1u));
                channel_sum = channel_sum + __uint_as_float(lds[channel * 64u + src]);
                if (G == 2u)
                    channel_sum2 = channel_sum2 + __uint_as_float(lds[128u + src]);
            }
            kept.set_channel_sum(channel_sum);
            if (G == 2u)
                kept.set_channel_sum2(channel_sum2);
            __syncthreads();
        }
    }
    const LanePixel me = locate(true);
    if (G > 1u) {
        // sum / n and the tone map are per channel (fs:636-640); the pixel's first lane collects the three and stores
        float c0 = kept.channel_sum() / fn, c2 = (G == 2u ? kept.channel_sum2() : 0.0f) / fn;
        if (fr.tonemap) {
            c0 = filmic(c0);
            c2 = filmic(c2);
        }
        uint32_t *lds = pool.stack - threadIdx.x;
        if (me.sub < 3u)
            lds[me.sub * 64u + me.base_lane] = __float_as_uint(c0);
        if (G == 2u && me.sub == 0u)
            lds[128u + me.base_lane] = __float_as_uint(c2);
        __syncthreads();
        if (me.sub == 0u && me.store)
            out[me.out_index] = me.inside ? make_float4(__uint_as_float(lds[threadIdx.x]), __uint_as_float(lds[64u + threadIdx.x]),
                                                        __uint_as_float(lds[128u + threadIdx.x]), 1.0f)
                                          : make_float4(0, 0, 0, 0);
    } else {
        V3 result = (ONE_SAMPLE || fr.spp == 1) ? sum : sum / fn;
        if (fr.tonemap)
            result = mk(filmic(result.x), filmic(result.y), filmic(result.z));
        if (me.store)
            out[me.out_index] = me.inside ? make_float4(result.x, result.y, result.z, 1.0f) : make_float4(0, 0, 0, 0);
    }
    if (ORDERED && fr.dispatch_cost && (threadIdx.x & 63u) == 0u) {
        // the wave's running time in units of 64 shader-clock ticks (32-bit difference: good for 1.7 s): its patch keeps the
        // longest of its waves' (and frames')
        atomicMax(fr.dispatch_cost + me.patch, ((uint32_t)__builtin_amdgcn_s_memtime() - cost_begin) >> 6);
    }
#ifdef SHRAY_DIAGNOSTICS
    if (counters && (threadIdx.x & 63u) == 0) {
        unsigned long long *tl = reinterpret_cast<unsigned long long *>(counters + kCounterShards) + 16ull * (blockIdx.x * 4u + (threadIdx.x >> 6));
        unsigned int hw_id, xcc_id;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw_id));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc_id));
        tl[0] = t_begin;
        tl[1] = __builtin_amdgcn_s_memrealtime();
        tl[2] = ((unsigned long long)xcc_id << 32) | hw_id;
        tl[3] = rc.node_visits;
        for (int k = 0; k < 8; k++)
            tl[4 + k] = pool.diag_tally[k];
        tl[12] = c_begin;
        tl[13] = __builtin_amdgcn_s_memtime();
    }
#endif
    if (COUNT)
        add_counters(rc, counters);
}

}   // namespace shray
