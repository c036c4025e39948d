// isa_costs.hip -- the arithmetic the shader demands per unit of work, counted in isolation.
//
// Each stage of the per-pixel path is wrapped in a kernel that runs it REPS times on independent inputs read from
// memory; profiles/isa_costs.py compiles this file with the product's flags and takes the difference of the VALU
// instruction counts of the REPS = 2 and REPS = 1 instances: per-ray set-up, addressing and stores cancel, what is
// left is one repetition of the stage -- the product's own inline functions (slab_range, triangle_distance,
// triangle_barycentrics, shade_hit, environment, filmic, the primary-ray statements), compiled as the kernels compile
// them.  Not linked into any library, never launched.
#define SHRAY_COST_MAIN_PATH     // (the rare lanes' true divisions are not part of the count: wave_traversal.h)
#include "stack_traversal.h"
#include "uniform_driver.h"

using namespace shray;

__device__ __forceinline__ LaneTraversal load_ray(const float *in)
{
    LaneTraversal t;
    const unsigned int i = blockIdx.x * blockDim.x + threadIdx.x;
    const float *r = in + 16u * i;
    t.P = mk(r[0], r[1], r[2]);
    t.D = mk(r[3], r[4], r[5]);
    t.Y = mk(r[6], r[7], r[8]);
    t.fx = t.D.x >= 0.0f;
    t.fy = t.D.y >= 0.0f;
    t.fz = t.D.z >= 0.0f;
    t.divide = false;
    t.hit = Hit{r[12], -1.0f, 0.0f, 0.0f};
    t.leaf_r0 = r[13];
    t.leaf_r1 = r[14];
    return t;
}

// one node visit's slab test + the hit decision (fs:200-217, :272-275, :400) as the kernels make it since round 4
// (wave_traversal.h: visit_decision -- the bounds that decide all but one visit in 10^5; cost_node_exact: the quotients)
template <int REPS>
__global__ void cost_node(const float *rays, const float4 *nodes, float *out)
{
    LaneTraversal t = load_ray(rays);
    const unsigned int i = blockIdx.x * blockDim.x + threadIdx.x;
#pragma unroll
    for (int k = 0; k < REPS; k++) {
        const float4 lo = nodes[(2u * i + 0u) + 4096u * k], hi = nodes[(2u * i + 1u) + 4096u * k];
        float r0, r1;
        uint32_t a = __float_as_uint(lo.w), b = __float_as_uint(hi.w);
        const bool enter = visit_decision(t, lo, hi, r0, r1, a, b);   // (lo, hi: a record of the ray's octant copy: no selects)
        out[i + 65536u * k] = enter ? r0 : r1;
    }
}

template <int REPS>
__global__ void cost_node_exact(const float *rays, const float4 *nodes, float *out)
{
    LaneTraversal t = load_ray(rays);
    const unsigned int i = blockIdx.x * blockDim.x + threadIdx.x;
#pragma unroll
    for (int k = 0; k < REPS; k++) {
        const float4 lo = nodes[(2u * i + 0u) + 4096u * k], hi = nodes[(2u * i + 1u) + 4096u * k];
        float r0, r1;
        slab_range(t, lo, hi, r0, r1);
        out[i + 65536u * k] = (!(r0 >= r1) && (r0 < t.hit.t)) ? r0 : r1;
    }
}

// first half of a triangle test: through the distance early-outs (fs:307-331)
template <int REPS>
__global__ void cost_triangle_distance(const float *rays, const float4 *tris, float *out)
{
    LaneTraversal t = load_ray(rays);
    const unsigned int i = blockIdx.x * blockDim.x + threadIdx.x;
#pragma unroll
    for (int k = 0; k < REPS; k++) {
        const float4 q0 = tris[3u * i + 8192u * k], q1 = tris[3u * i + 1u + 8192u * k], q2 = tris[3u * i + 2u + 8192u * k];
        TriangleSetup s;
        const bool pass = triangle_distance(t, q0, q1, q2, s);
        // everything the second half reads stays live, as in the product
        float *o = out + 16u * i + 1048576u * k;
        o[0] = pass ? s.dist : 0.0f;
        o[1] = s.M.x; o[2] = s.M.y; o[3] = s.M.z;
        o[4] = s.T.x; o[5] = s.T.y; o[6] = s.T.z;
        o[7] = s.Q.x; o[8] = s.Q.y; o[9] = s.Q.z;
        o[10] = s.inv_det;
    }
}

// the whole triangle test (fs:297-346); the difference to the first half is the barycentric part
template <int REPS>
__global__ void cost_triangle_full(SceneView sc, const float *rays, const float4 *tris, float *out)
{
    LaneTraversal t = load_ray(rays);
    const unsigned int i = blockIdx.x * blockDim.x + threadIdx.x;
    RayCounters rc = {};
#pragma unroll
    for (int k = 0; k < REPS; k++) {
        const float4 q0 = tris[3u * i + 8192u * k], q1 = tris[3u * i + 1u + 8192u * k], q2 = tris[3u * i + 2u + 8192u * k];
        lane_test_triangle_loaded<false, false>(sc, t, i + 7u * k, rc, q0, q1, q2);
    }
    out[4u * i] = t.hit.t;
    out[4u * i + 1] = t.hit.which;
    out[4u * i + 2] = t.hit.bu;
    out[4u * i + 3] = t.hit.bv;
}

// a shaded hit: interpolated normal, facing, transfer + reflect, Fresnel (fs:288-295, :503-521, :65-96, :479-482)
template <int REPS>
__global__ void cost_shade(SceneView sc, const FrameView *frames, const float *rays, float *out)
{
    const FrameView &fr = frames[0];
    const unsigned int i = blockIdx.x * blockDim.x + threadIdx.x;
    const V3 spec = mk(fr.specular_color[0], fr.specular_color[1], fr.specular_color[2]);
    V3 acc = mk(0, 0, 0);
#pragma unroll
    for (int k = 0; k < REPS; k++) {
        const float *r = rays + 16u * i + 1048576u * k;
        const Hit hit{r[6], r[7], r[8], r[9]};
        const ShadedHit s = shade_hit(sc, fr, spec, mk(r[0], r[1], r[2]), mk(r[3], r[4], r[5]), hit);
        float *o = out + 16u * i + 1048576u * k;
        o[0] = s.R.x; o[1] = s.R.y; o[2] = s.R.z;
        o[3] = s.P2.x; o[4] = s.P2.y; o[5] = s.P2.z;
        o[6] = s.object_specular.x; o[7] = s.object_specular.y; o[8] = s.object_specular.z;
    }
}

// the world-to-object transform of a ray at the start of a traversal + the per-ray set-up of the exact slab test
// (fs:486-489; exact_div.h's reciprocals stand for the shader's six divisions per visit)
template <int REPS>
__global__ void cost_traversal_setup(SceneView sc, const FrameView *frames, const float *rays, float *out)
{
    const FrameView &fr = frames[0];
    const unsigned int i = blockIdx.x * blockDim.x + threadIdx.x;
#pragma unroll
    for (int k = 0; k < REPS; k++) {
        const float *r = rays + 16u * i + 1048576u * k;
        const V3 P = xform(fr.object_matrix, mk(r[0], r[1], r[2]), 1.0f), D = xform(fr.object_normal_matrix, mk(r[3], r[4], r[5]), 0.0f);
        const V3 Y = mk(reciprocal_in_range(D.x), reciprocal_in_range(D.y), reciprocal_in_range(D.z));   // (as lane_begin)
        float *o = out + 16u * i + 1048576u * k;
        o[0] = P.x; o[1] = P.y; o[2] = P.z; o[3] = D.x; o[4] = D.y; o[5] = D.z;
        o[6] = Y.x; o[7] = Y.y; o[8] = Y.z;
    }
}

// sample_environment with which == 0 (fs:127-155): lat-long coordinates + bilinear lookup + the modulate-and-add of fs:580
template <int REPS>
__global__ void cost_environment(SceneView sc, const float *rays, float *out)
{
    const unsigned int i = blockIdx.x * blockDim.x + threadIdx.x;
#pragma unroll
    for (int k = 0; k < REPS; k++) {
        const float *r = rays + 16u * i + 1048576u * k;
        const V3 radiance = mk(r[3], r[4], r[5]) + mk(r[6], r[7], r[8]) * environment(sc, mk(r[0], r[1], r[2]));
        float *o = out + 4u * i + 1048576u * k;
        o[0] = radiance.x; o[1] = radiance.y; o[2] = radiance.z;
    }
}

// primary ray (vs:39-60, fs:619) and tone map (fs:527-548) of one sample
template <int REPS>
__global__ void cost_primary_and_tonemap(const FrameView *frames, const float *rays, float *out)
{
    const FrameView &fr = frames[0];
    const unsigned int i = blockIdx.x * blockDim.x + threadIdx.x;
    const float fw = (float)fr.width, fh = (float)fr.height;
    // (as uniform_driver.h makes them: the divisors' reciprocals once, outside the count of one repetition)
    const SharedDivisor by_width = shared_divisor(fw), by_height = shared_divisor(fh);
    const bool eye_in_range = magnitude_in(fr.image_plane_width, -30, 8) && magnitude_in(fr.aspect, -30, 8);
#pragma unroll
    for (int k = 0; k < REPS; k++) {
        const float *r = rays + 16u * i + 1048576u * k;
        const float u = divide_by_shared(r[0] + 0.5f, by_width), v = divide_by_shared(r[1] + 0.5f, by_height);
        const V3 eye = unit_of_eye(mk(fr.image_plane_width * (u - 0.5f), fr.image_plane_width * (v - 0.5f) * fr.aspect, -1.0f), eye_in_range);
        const V3 P = xform(fr.camera_matrix, mk(0, 0, 0), 1.0f);
        const V3 D = unit(xform(fr.camera_normal_matrix, eye, 0.0f));
        float *o = out + 16u * i + 1048576u * k;
        o[0] = P.x; o[1] = P.y; o[2] = P.z; o[3] = D.x; o[4] = D.y; o[5] = D.z;
        o[6] = filmic(r[2]); o[7] = filmic(r[3]); o[8] = filmic(r[4]);
    }
}

#define INSTANTIATE(K, ...)                              \
    template __global__ void K<1>(__VA_ARGS__);          \
    template __global__ void K<2>(__VA_ARGS__);
INSTANTIATE(cost_node, const float *, const float4 *, float *)
INSTANTIATE(cost_triangle_distance, const float *, const float4 *, float *)
INSTANTIATE(cost_triangle_full, SceneView, const float *, const float4 *, float *)
INSTANTIATE(cost_node_exact, const float *, const float4 *, float *)
INSTANTIATE(cost_shade, SceneView, const FrameView *, const float *, float *)
INSTANTIATE(cost_traversal_setup, SceneView, const FrameView *, const float *, float *)
INSTANTIATE(cost_environment, SceneView, const float *, float *)
INSTANTIATE(cost_primary_and_tonemap, const FrameView *, const float *, float *)
