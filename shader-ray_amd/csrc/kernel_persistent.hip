// kernel_persistent.hip -- kernel id 2: persistent waves, ray replacement, phase batching.
//
// Why: with one thread per pixel (kernel_stack.hip) a frame takes as long as its slowest
// wave -- a wave over the object whose 64 rays need different numbers of bounces and node
// visits, and in which a single lane sitting in a leaf (<= 10 dependent triangle tests,
// raytracer.es.fs:412-417) stalls the 63 others (profiles/r01: average residency 8 of 16
// wave slots per CU, kernel span == longest wave).  Here a fixed grid of waves stays
// resident and every LANE is a small state machine that walks the reference's per-pixel
// path (raytracer.es.fs:552-582) one step at a time:
//
//     FETCH  take the next sample of my pixel, or a new pixel from the wave's pool
//            (pool = one 8x8 tile = 64 pixels, claimed with one atomic per wave)
//     INNER  visit one BVH node: slab test, push far child / pop            (fs:399-429)
//     LEAF   test the triangles of the leaf I am in                          (fs:412-417)
//     SHADE  a traversal ended: shade the hit, start the shadow ray or the next bounce,
//            or look up the environment and finish the sample               (fs:484-522, :447-472, :563-581)
//
// The wave runs the stage most lanes are waiting for, so node visits execute with (mostly)
// full waves, triangle tests are batched, and a lane whose path ends early is re-used for
// another pixel instead of idling until the longest path of its wave is done.
//
// Nothing about a ray's own sequence changes: same visits in the same order with the same
// hit.t at every test, same iteration count (so the 400-iteration marker and the 10-triangle
// cap are reproduced), same arithmetic as trace_common.h / stack_traversal.h.  Frames and
// work counters are bit-identical to kernels 0 and 1 and to the CPU oracle.
#include "launch.h"
#include "wave_traversal.h"

namespace shray {

constexpr int kPBlock = 256;

enum : int { PH_FETCH = 0, PH_INNER = LT_WALK, PH_LEAF = LT_LEAF, PH_SHADE = LT_ENDED, PH_DONE = 4 };
enum : int { MODE_CLOSEST = 0, MODE_SHADOW = 1 };

// stage-switch thresholds (lanes); see DESIGN.md "Persistent kernel"
#ifndef SHRAY_FETCH_MIN
#define SHRAY_FETCH_MIN 8
#endif
#ifndef SHRAY_SHADE_MIN
#define SHRAY_SHADE_MIN 8
#endif
constexpr int kFetchMin = SHRAY_FETCH_MIN;   // refill when this many lanes are idle (or nothing else can run)
constexpr int kShadeMin = SHRAY_SHADE_MIN;   // shade when this many traversals have ended (or nothing else can run)

__device__ __forceinline__ int popc64(unsigned long long m) { return __popcll(m); }
__device__ __forceinline__ unsigned long long ballot(bool p) { return __builtin_amdgcn_ballot_w64(p); }
__device__ __forceinline__ unsigned int lane_rank(unsigned long long mask)   // set bits of mask below this lane
{
    return __builtin_amdgcn_mbcnt_hi((unsigned int)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned int)mask, 0u));
}

template <bool COUNT>
__global__ void __launch_bounds__(kPBlock) trace_persistent_kernel(SceneView sc, FrameView fr, float4 *__restrict__ out,
                                                                   DeviceCounters *counters, unsigned int *next_patch)
{
    extern __shared__ uint32_t lds_stack[];
    uint32_t *const stack = lds_stack + threadIdx.x;   // [level * kPBlock]
    const unsigned int lane = threadIdx.x & 63u;
#ifdef SHRAY_DIAGNOSTICS
    const unsigned long long t_begin = __builtin_amdgcn_s_memrealtime();
    unsigned long long diag_outer = 0, diag_fetch = 0, diag_shade = 0, diag_c_fetch = 0, diag_c_shade = 0, diag_c_walk = 0;
#endif

    const V3 light = mk(fr.light_dir[0], fr.light_dir[1], fr.light_dir[2]);
    const V3 spec = mk(fr.specular_color[0], fr.specular_color[1], fr.specular_color[2]);
    const V3 diff = mk(fr.diffuse_color[0], fr.diffuse_color[1], fr.diffuse_color[2]);
    const bool has_diffuse = diff.x > 0.0f && diff.y > 0.0f && diff.z > 0.0f;   // fs:570
    const float fw = (float)fr.width, fh = (float)fr.height, fn = (float)fr.spp;

    // ---- wave-uniform pixel pool: indices [pool_next, pool_end) of patch-major pixel numbering
    unsigned int pool_next = 0, pool_end = 0;
    bool pool_dry = false;

    // ---- per-lane state
    int phase = PH_FETCH;
    bool have_pixel = false;
    unsigned int pixel = 0;            // patch * 256 + index in patch
    int sample = 0;
    V3 sum = mk(0, 0, 0);
    V3 Pw = mk(0, 0, 0), Dw = mk(0, 0, 1);                 // world-space ray of the current bounce
    V3 accumulated = mk(0, 0, 0), modulation = mk(1, 1, 1);
    int bounce = 0, mode = MODE_CLOSEST;
    V3 next_D = mk(0, 0, 0), next_spec = mk(0, 0, 0);      // kept across a shadow traversal
    float lcos = 0.0f;
    LaneTraversal t;
    t.hit = Hit{kFar, -1.0f, 0.0f, 0.0f};
    t.node = 0;
    t.sp = 0;
    t.iter = 0;
    t.leaf_count = 0;
    RayCounters rc = {0, 0, 0, 0, 0, 0, 0};
    SHRAY_DIAG_DECL

    // start a traversal of the object-space image of (origin, direction)       (fs:489-491, :462-463)
    auto begin_traversal = [&](V3 origin, V3 direction) {
        lane_begin<COUNT>(sc, t, xform(fr.object_matrix, origin, 1.0f), xform(fr.object_normal_matrix, direction, 0.0f), rc);
        phase = PH_INNER;
    };

    for (;;) {
#ifdef SHRAY_DIAGNOSTICS
        diag_outer++;
        const unsigned long long dc0 = __builtin_amdgcn_s_memtime();
#endif
        // ------------------------------------------------------------------ FETCH
        {
            const unsigned long long want = ballot(phase == PH_FETCH);
            const unsigned long long busy = ballot(phase == PH_INNER || phase == PH_LEAF || phase == PH_SHADE);
            if (want && (popc64(want) >= kFetchMin || !busy)) {
#ifdef SHRAY_DIAGNOSTICS
                diag_fetch++;
#endif
                bool need = phase == PH_FETCH;
                // next sample of the same pixel?
                bool new_pixel = need && !(have_pixel && sample < fr.spp);
                // claim pixels for the lanes that need a new one
                unsigned long long m = ballot(new_pixel);
                unsigned int rank = lane_rank(m);
                bool waiting = new_pixel;
                while (ballot(waiting)) {
                    unsigned int avail = pool_end - pool_next;
                    if (avail == 0) {
                        if (pool_dry)
                            break;
                        unsigned int tile = 0;   // 64-pixel wave tile: 4 per 16x16 patch
                        if (lane == 0)
                            tile = atomicAdd(next_patch, 1u);
                        tile = __builtin_amdgcn_readfirstlane(tile);
                        if (tile >= fr.total_patches * 4u) {
                            pool_dry = true;
                            break;
                        }
                        pool_next = tile * 64u;
                        pool_end = pool_next + 64u;
                        avail = 64u;
                    }
                    const unsigned int wanted = (unsigned int)popc64(ballot(waiting));
                    if (waiting && rank < avail) {
                        pixel = pool_next + rank;
                        waiting = false;
                        have_pixel = true;
                        sample = 0;
                        sum = mk(0, 0, 0);
                    } else if (waiting) {
                        rank -= avail;
                    }
                    pool_next += wanted < avail ? wanted : avail;
                }
                if (waiting) {   // no work left for this lane
                    phase = PH_DONE;
                    have_pixel = false;
                    need = false;
                }
                if (need) {
                    // pixel coordinates: same patch / wave-tile mapping as trace_pixels()
                    const unsigned int patch = pixel >> 8, k = pixel & 255u;
                    const unsigned int wv = k >> 6, ln = k & 63u;
                    const int lx = (int)((wv & 1u) * 8u + (ln & 7u)), ly = (int)((wv >> 1) * 8u + (ln >> 3));
                    int px, py;
                    if (fr.tile_stride == 0) {
                        px = (int)(patch % (unsigned int)fr.patches_x) * 16 + lx;
                        py = (int)(patch / (unsigned int)fr.patches_x) * 16 + ly;
                    } else {
                        const unsigned int kk = patch / (unsigned int)fr.patches_per_unit;
                        const unsigned int q = patch % (unsigned int)fr.patches_per_unit;
                        const unsigned int tile = kk * (unsigned int)fr.tile_stride + (unsigned int)fr.tile_phase;
                        px = (int)(tile % (unsigned int)fr.tiles_x) * fr.tile_w + (int)(q % (unsigned int)fr.patches_x) * 16 + lx;
                        py = (int)(tile / (unsigned int)fr.tiles_x) * fr.tile_h + (int)(q / (unsigned int)fr.patches_x) * 16 + ly;
                    }
                    if (px >= fr.width || py >= fr.height) {
                        // padding of an edge patch: tiled output gets zeros, untiled output has no such texel
                        if (fr.tile_stride != 0) {
                            const unsigned int kk = patch / (unsigned int)fr.patches_per_unit;
                            const unsigned int q = patch % (unsigned int)fr.patches_per_unit;
                            const int tlx = (int)(q % (unsigned int)fr.patches_x) * 16 + lx;
                            const int tly = (int)(q / (unsigned int)fr.patches_x) * 16 + ly;
                            out[(size_t)kk * fr.tile_w * fr.tile_h + (size_t)tly * fr.tile_w + tlx] = make_float4(0, 0, 0, 0);
                        }
                        have_pixel = false;   // stays in FETCH: asks for another pixel next time round
                    } else {
                        // primary ray of sample `sample`                         (vs:39-60, fs:617-619)
                        const float ox = ((float)sample + 0.5f) / fn;
                        const float oy = (float)__brev((unsigned int)sample) * 2.3283064365386963e-10f + 0.5f / fn;
                        const float u = ((float)px + ox) / fw;
                        const float v = ((float)py + oy) / fh;
                        const V3 eye = unit(mk(fr.image_plane_width * (u - 0.5f), fr.image_plane_width * (v - 0.5f) * fr.aspect, -1.0f));
                        Pw = xform(fr.camera_matrix, mk(0, 0, 0), 1.0f);
                        Dw = unit(xform(fr.camera_normal_matrix, eye, 0.0f));
                        accumulated = mk(0, 0, 0);
                        modulation = mk(1, 1, 1);
                        bounce = 0;
                        mode = MODE_CLOSEST;
                        if (fr.bounce_count > 0) {
                            begin_traversal(Pw, Dw);
                        } else {
                            t.hit = Hit{kFar, -1.0f, 0.0f, 0.0f};   // no bounces: straight to the environment (fs:556, :580)
                            phase = PH_SHADE;
                        }
                    }
                }
            }
        }
        if (ballot(phase != PH_DONE) == 0ull)
            break;

#ifdef SHRAY_DIAGNOSTICS
        const unsigned long long dc1 = __builtin_amdgcn_s_memtime();
#endif
        // ------------------------------------------------------------------ WALK: one node step + one triangle step
        {
            const unsigned long long walkers = wave_ballot(phase == PH_INNER);
            const unsigned long long parked = wave_ballot(phase == PH_LEAF);
            if (walkers | parked)
                walk_step<COUNT, kPBlock>(sc, fr, t, phase, stack, rc, walkers, parked, 1 SHRAY_DIAG_ARG);
        }
#ifdef SHRAY_DIAGNOSTICS
        const unsigned long long dc2 = __builtin_amdgcn_s_memtime();
#endif

        // ------------------------------------------------------------------ SHADE
        {
            const unsigned long long ended = ballot(phase == PH_SHADE);
            const unsigned long long walking = ballot(phase == PH_INNER || phase == PH_LEAF);
            if (ended && (popc64(ended) >= kShadeMin || !walking)) {
#ifdef SHRAY_DIAGNOSTICS
                diag_shade++;
#endif
                if (phase == PH_SHADE) {
                    bool path_done = false;
                    bool next_bounce = false;
                    V3 radiance = mk(0, 0, 0);
                    if (mode == MODE_CLOSEST) {
                        if (t.hit.t >= kFar) {                         // nothing hit: leave the bounce loop (fs:563-565)
                            if (COUNT)
                                rc.env_lookups++;
                            radiance = accumulated + modulation * environment(sc, Dw);
                            path_done = true;
                        } else if (t.hit.t == -1.0f) {                 // iteration-cap marker (fs:566-568)
                            if (COUNT)
                                rc.bad_hits++;
                            radiance = mk(1.0f, 0.0f, 0.0f);
                            path_done = true;
                        } else {                                     // intersect_and_shade, fs:503-521
                            if (COUNT)
                                rc.shaded_hits++;
                            const V3 object_normal = interpolated_normal(sc, fr.normals_fp16 != 0, t.hit.which, t.hit.bu, t.hit.bv);
                            V3 n = xform(fr.object_normal_inverse, object_normal, 0.0f);
                            if (dot3(n, Dw) > 0.0f)
                                n = n * -1.0f;
                            const V3 at = Pw + Dw * t.hit.t;
                            const V3 R = Dw - n * (2.0f * dot3(n, Dw));
                            const V3 P2 = at + n * .0001f;
                            const float fresnel = pow5(dot3(Dw, R) * .5f + .5f);
                            next_spec = spec + (mk(1.0f, 1.0f, 1.0f) - spec) * fresnel;
                            next_D = R;
                            Pw = P2;
                            if (has_diffuse) {                       // approximate_diffuse, fs:447-472
                                lcos = sel_max(0.0f, dot3(n, light));
                                if (fr.cast_shadows) {
                                    mode = MODE_SHADOW;
                                    begin_traversal(P2, light);
                                } else {
                                    accumulated = accumulated + modulation * diff * (mk(0, 0, 0) + mk(1.0f, 1.0f, 1.0f) * lcos);
                                    next_bounce = true;
                                }
                            } else {
                                next_bounce = true;
                            }
                        }
                    } else {                                         // the shadow ray came back (fs:464-466)
                        V3 irradiance = mk(0, 0, 0);
                        if (t.hit.t >= kFar)
                            irradiance = irradiance + mk(1.0f, 1.0f, 1.0f) * lcos;
                        accumulated = accumulated + modulation * diff * irradiance;
                        mode = MODE_CLOSEST;
                        next_bounce = true;
                    }
                    if (next_bounce) {                               // fs:576-578
                        modulation = modulation * next_spec;
                        Dw = next_D;
                        bounce++;
                        if (bounce < fr.bounce_count) {
                            begin_traversal(Pw, Dw);
                        } else {                                     // loop ran out: environment along the last ray (fs:580)
                            if (COUNT)
                                rc.env_lookups++;
                            radiance = accumulated + modulation * environment(sc, Dw);
                            path_done = true;
                        }
                    }
                    if (path_done) {
                        sum = (fr.spp == 1) ? radiance : sum + radiance;
                        sample++;
                        if (sample == fr.spp) {
                            V3 result = (fr.spp == 1) ? sum : sum / fn;
                            if (fr.tonemap)
                                result = mk(filmic(result.x), filmic(result.y), filmic(result.z));
                            // output slot of `pixel`
                            const unsigned int patch = pixel >> 8, k = pixel & 255u;
                            const unsigned int wv = k >> 6, ln = k & 63u;
                            const int lx = (int)((wv & 1u) * 8u + (ln & 7u)), ly = (int)((wv >> 1) * 8u + (ln >> 3));
                            size_t index;
                            if (fr.tile_stride == 0) {
                                const int px = (int)(patch % (unsigned int)fr.patches_x) * 16 + lx;
                                const int py = (int)(patch / (unsigned int)fr.patches_x) * 16 + ly;
                                index = (size_t)py * fr.width + px;
                            } else {
                                const unsigned int kk = patch / (unsigned int)fr.patches_per_unit;
                                const unsigned int q = patch % (unsigned int)fr.patches_per_unit;
                                const int tlx = (int)(q % (unsigned int)fr.patches_x) * 16 + lx;
                                const int tly = (int)(q / (unsigned int)fr.patches_x) * 16 + ly;
                                index = (size_t)kk * fr.tile_w * fr.tile_h + (size_t)tly * fr.tile_w + tlx;
                            }
                            out[index] = make_float4(result.x, result.y, result.z, 1.0f);
                            have_pixel = false;
                        }
                        phase = PH_FETCH;
                    }
                }
            }
        }
#ifdef SHRAY_DIAGNOSTICS
        const unsigned long long dc3 = __builtin_amdgcn_s_memtime();
        diag_c_fetch += dc1 - dc0;
        diag_c_walk += dc2 - dc1;
        diag_c_shade += dc3 - dc2;
#endif
    }

#ifdef SHRAY_DIAGNOSTICS
    if (counters && lane == 0) {   // diagnostic build: per-wave life stamps and stage tallies (profiles/timeline.py)
        unsigned long long *tl = reinterpret_cast<unsigned long long *>(counters + kCounterShards) + 16ull * (blockIdx.x * 4u + (threadIdx.x >> 6));
        tl[0] = t_begin;
        tl[1] = __builtin_amdgcn_s_memrealtime();
        tl[2] = diag_outer;
        tl[3] = diag_fetch;
        tl[4] = diag_shade;
        tl[5] = diag_tally[0];
        tl[6] = diag_tally[1];
        tl[7] = diag_c_fetch;
        tl[8] = diag_c_walk;
        tl[9] = diag_c_shade;
    }
#endif

    if (COUNT) {
        const unsigned int vals[7] = {rc.node_visits, rc.leaf_visits, rc.triangle_tests, rc.shaded_hits,
                                      rc.env_lookups, rc.traversals, rc.bad_hits};
        unsigned long long *dst = &counters[blockIdx.x % kCounterShards].node_visits;
#pragma unroll
        for (int k = 0; k < 7; k++) {
            const unsigned long long s = wave_sum(vals[k]);
            if (lane == 0 && s)
                atomicAdd(dst + k, s);
        }
    }
}

int persistent_blocks_per_cu(int stack_levels)
{
    int blocks = 0;
    const size_t lds_bytes = (size_t)kPBlock * (size_t)stack_levels * sizeof(uint32_t);
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&blocks, trace_persistent_kernel<false>, kPBlock, lds_bytes) != hipSuccess)
        blocks = 0;
    return blocks > 0 ? blocks : 1;
}

hipError_t launch_persistent(const SceneView &sc, const FrameView &fr, float4 *out, DeviceCounters *counters,
                             hipStream_t stream, int stack_levels, unsigned int *work_counter, int resident_blocks)
{
    hipError_t e = hipMemsetAsync(work_counter, 0, sizeof(unsigned int), stream);
    if (e != hipSuccess)
        return e;
    const unsigned int blocks = (unsigned int)resident_blocks < fr.total_patches ? (unsigned int)resident_blocks : fr.total_patches;
    const dim3 grid(blocks), block(kPBlock);
    const size_t lds_bytes = (size_t)kPBlock * (size_t)stack_levels * sizeof(uint32_t);
#ifdef SHRAY_DIAGNOSTICS
    if (counters && g_diag_plain_kernel) {
        hipLaunchKernelGGL(trace_persistent_kernel<false>, grid, block, lds_bytes, stream, sc, fr, out, counters, work_counter);
        return hipGetLastError();
    }
#endif
    if (counters)
        hipLaunchKernelGGL(trace_persistent_kernel<true>, grid, block, lds_bytes, stream, sc, fr, out, counters, work_counter);
    else
        hipLaunchKernelGGL(trace_persistent_kernel<false>, grid, block, lds_bytes, stream, sc, fr, out, counters, work_counter);
    return hipGetLastError();
}

}   // namespace shray
