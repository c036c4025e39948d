// trace_common.h -- device-side per-pixel path shared by the gfx950 kernels:
// primary-ray generation, the bounce loop, shading, shadow rays, environment
// lookup and tone mapping.  The BVH traversal itself is a policy class
// (threaded_traversal.h / stack_traversal.h) plugged into trace_pixels<>.
//
// Restates, as one HIP kernel, the reference's two GLSL stages:
//   raytracer.vs:39-60      primary ray through the image plane
//   raytracer.es.fs:552-582 trace(): <= bounce_count closest-hit traversals,
//                           Fresnel-weighted mirror bounces, shadow rays for
//                           non-metals, one environment lookup at the end
//   raytracer.es.fs:527-548 filmic tone map
// Arithmetic contract (identical to the CPU oracle's; DESIGN.md "Arithmetic"):
// single-rounded IEEE fp32 operations in the shader's order, compiled with
// -ffp-contract=off and correctly rounded divide/sqrt.  atan / acos / pow(x,5)
// are explicit fp32 operation sequences (atan_yx, acos_clamped, pow5 below),
// not library calls, so a frame is bit-reproducible against the CPU oracle.
#pragma once

#include <hip/hip_runtime.h>

#include "device_types.h"

// 0: the convergent batch instances run 256-thread workgroups (one 16x16 patch); 1: one-wave workgroups (one 8x8 tile),
// tiles of a patch on consecutive workgroup ids; 2: one-wave workgroups, the four tiles of a patch on one XCD

namespace shray {

struct V3 {
    float x, y, z;
};
__device__ __forceinline__ V3 mk(float x, float y, float z) { return V3{x, y, z}; }
__device__ __forceinline__ V3 operator+(V3 a, V3 b) { return mk(a.x + b.x, a.y + b.y, a.z + b.z); }
__device__ __forceinline__ V3 operator-(V3 a, V3 b) { return mk(a.x - b.x, a.y - b.y, a.z - b.z); }
__device__ __forceinline__ V3 operator*(V3 a, V3 b) { return mk(a.x * b.x, a.y * b.y, a.z * b.z); }
__device__ __forceinline__ V3 operator*(V3 a, float s) { return mk(a.x * s, a.y * s, a.z * s); }
__device__ __forceinline__ V3 operator/(V3 a, float s) { return mk(a.x / s, a.y / s, a.z / s); }
__device__ __forceinline__ float dot3(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__device__ __forceinline__ V3 cross3(V3 a, V3 b)
{
    return mk(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
__device__ __forceinline__ V3 unit(V3 a) { return a / sqrtf(dot3(a, a)); }
// GLSL max/min: the SECOND operand wins only on a strict compare
__device__ __forceinline__ float sel_max(float x, float y) { return x < y ? y : x; }
__device__ __forceinline__ float sel_min(float x, float y) { return y < x ? y : x; }

// atan(y, x): octant reduction, then z + z^3 * P(z^2) on |z| <= tan(pi/8);
// coefficients from oracle/tools/fit_atan.py (max error 2.8 ulp).
__device__ __forceinline__ float atan_yx(float y, float x)
{
    const float ax = fabsf(x), ay = fabsf(y);
    const float hi = ax < ay ? ay : ax, lo = ax < ay ? ax : ay;
    if (hi == 0.0f)
        return 0.0f;
    const float q = lo / hi;
    const bool fold = q > 0.414213562f;
    const float z = fold ? (q - 1.0f) / (q + 1.0f) : q;
    const float zz = z * z;
    float poly = 0.0803788006f * zz;
    poly = (poly + -0.138722613f) * zz;
    poly = (poly + 0.199771404f) * zz;
    poly = poly + -0.33332932f;
    float angle = z + (z * zz) * poly;
    angle = fold ? 0.785398163f + angle : angle;
    angle = ay > ax ? 1.57079633f - angle : angle;
    angle = x < 0.0f ? 3.14159265f - angle : angle;
    return y < 0.0f ? -angle : angle;
}
__device__ __forceinline__ float acos_clamped(float c) { return atan_yx(sqrtf((1.0f - c) * (1.0f + c)), c); }
// pow(b, 5.0), fs:481: b^5 for b >= 0; NaN for a negative base, as the GL implementations' exp2(5 log2 b) gives (the
// oracle's sr_pow5 has the measurement): such a pixel ends black through the tone map's max(0, c - .004)
__device__ __forceinline__ float pow5(float b)
{
    const float b2 = b * b;
    const float p = (b2 * b2) * b;
    return b < 0.0f ? __uint_as_float(0x7fc00000u) : p;
}

// column-major mat4 times (v, w)
__device__ __forceinline__ V3 xform(const float *m, V3 v, float w)
{
    return mk(m[0] * v.x + m[4] * v.y + m[8] * v.z + m[12] * w,
              m[1] * v.x + m[5] * v.y + m[9] * v.z + m[13] * w,
              m[2] * v.x + m[6] * v.y + m[10] * v.z + m[14] * w);
}

struct Hit {        // surface_hit, raytracer.es.fs:108-113
    float t;
    float which;    // triangle index as a float; -1 = none
    float bu, bv;   // barycentric u, v (uvw = (1-u-v, u, v)); for a bad hit the marker colour is implied
};

constexpr float kFar = 10000000.0f;        // infinitely_far, fs:115
constexpr float kRangeMax = 100000000.0f;  // traversal range, fs:491
constexpr float kPi = 3.14159265259f;      // fs:116
constexpr float kTerminator = 16777215.0f; // fs:384

struct RayCounters {
    unsigned int node_visits, leaf_visits, triangle_tests, shaded_hits, env_lookups, traversals, bad_hits;
};

__device__ __forceinline__ float half_bits_to_float(uint16_t h)
{
    return (float)__builtin_bit_cast(_Float16, h);   // exact widening (v_cvt_f32_f16)
}

// triangle_interpolate_normal, fs:288-295 (vertex normals from the fp16 or fp32 copy)
__device__ __forceinline__ V3 interpolated_normal(const SceneView &sc, bool fp16, float which, float bu, float bv)
{
    // one address per triangle, the nine components at immediate offsets from it (indexed as [base + k] with an unsigned
    // 32-bit base every component got its own 64-bit address: ~25 instructions per hit)
    const size_t base = (size_t)(9u * (unsigned int)which);
    float n[9];
    if (fp16) {
        const auto *p = sc.normals16 + base;
#pragma unroll
        for (int k = 0; k < 9; k++)
            n[k] = half_bits_to_float(p[k]);
    } else {
        const float *p = sc.normals32 + base;
#pragma unroll
        for (int k = 0; k < 9; k++)
            n[k] = p[k];
    }
    const float bw = 1.0f - bu - bv;
    return mk(n[0], n[1], n[2]) * bw + mk(n[3], n[4], n[5]) * bu + mk(n[6], n[7], n[8]) * bv;
}

// REPEAT wrap of a texel index: (i % n, made non-negative) for EVERY int i and n >= 1.  A remainder by a divisor that is not a
// compile-time constant is ~28 vector instructions, and a lookup makes four; the lookup's own coordinates (s in [1/2, 3/2],
// t in [0, 1]) put i in [-n, 2n), where the remainder is one conditional subtraction and the fix-up one conditional addition.
// Anything else (the filtered view's probes, coordinates that are not numbers) takes the remainders, in ONE branch the wave skips.
__device__ __forceinline__ int wrap_near(int i, int n)
{
    const int r = i >= n ? i - n : i;
    return r < 0 ? r + n : r;
}
__device__ __forceinline__ bool wrap_is_near(int i, int n) { return (unsigned int)i + (unsigned int)n < 3u * (unsigned int)n; }   // -n <= i < 2n
__device__ __forceinline__ int wrap_any(int i, int n)
{
    const int r = i % n;
    return r < 0 ? r + n : r;
}

// LINEAR lookup with REPEAT wrap in one level of the environment pyramid
__device__ __forceinline__ V3 environment_level(const float *texels, int w, int h, float s, float t)
{
    const float u = s * (float)w - 0.5f;
    const float v = t * (float)h - 0.5f;
    const float fu = floorf(u), fv = floorf(v);
    const float a = u - fu, b = v - fv;
    const int iu0 = (int)fu, iu1 = (int)(fu + 1.0f), iv0 = (int)fv, iv1 = (int)(fv + 1.0f);
    int i0 = wrap_near(iu0, w), i1 = wrap_near(iu1, w), j0 = wrap_near(iv0, h), j1 = wrap_near(iv1, h);
#ifndef SHRAY_COST_MAIN_PATH     // profiles/isa_costs.hip counts the path every wave takes
    const bool far = !(wrap_is_near(iu0, w) && wrap_is_near(iu1, w) && wrap_is_near(iv0, h) && wrap_is_near(iv1, h));
    if (__builtin_expect(__builtin_amdgcn_ballot_w64(far) != 0ull, 0)) {
        asm volatile("; a texel index outside [-n, 2n): the remainders" ::: "memory");   // keeps this a branch
        if (far) {
            i0 = wrap_any(iu0, w);
            i1 = wrap_any(iu1, w);
            j0 = wrap_any(iv0, h);
            j1 = wrap_any(iv1, h);
        }
    }
#endif
    const float *r0 = texels + 3 * (size_t)j0 * w;
    const float *r1 = texels + 3 * (size_t)j1 * w;
    const V3 t00 = mk(r0[3 * i0], r0[3 * i0 + 1], r0[3 * i0 + 2]);
    const V3 t10 = mk(r0[3 * i1], r0[3 * i1 + 1], r0[3 * i1 + 2]);
    const V3 t01 = mk(r1[3 * i0], r1[3 * i0 + 1], r1[3 * i0 + 2]);
    const V3 t11 = mk(r1[3 * i1], r1[3 * i1 + 1], r1[3 * i1 + 2]);
    const float w00 = (1.0f - a) * (1.0f - b), w10 = a * (1.0f - b), w01 = (1.0f - a) * b, w11 = a * b;
    return t00 * w00 + t10 * w10 + t01 * w01 + t11 * w11;
}

// lat-long lookup coordinates of a direction, fs:130
__device__ __forceinline__ void lookup_coords(V3 d, float &s, float &t)
{
    const float dy = sel_min(sel_max(d.y, -1.0f), 1.0f);
    s = 1.0f + atan_yx(-d.z, d.x) / (2 * kPi);
    t = 1.0f - acos_clamped(dy) / kPi;
}

// sample_environment, fs:127-155 with which == 0: level-0 bilinear, REPEAT wrap
// acos(d.y) is undefined for |d.y| > 1 (d is not a unit vector: reflections about normals that are not renormalized), NaN on
// the GL implementations that run the reference: the lookup's colour is NaN there (the oracle's sample_environment has the
// measurement), which the tone map's max(0, c - .004) turns into a black pixel.
__device__ __forceinline__ V3 outside_acos_is_nan(V3 colour, float dy)
{
    const float nan = __uint_as_float(0x7fc00000u);
    return fabsf(dy) <= 1.0f ? colour : mk(nan, nan, nan);
}
__device__ __forceinline__ V3 environment(const SceneView &sc, V3 d)
{
    float s, t;
    lookup_coords(d, s, t);
    return outside_acos_is_nan(environment_level(sc.env, sc.env_w, sc.env_h, s, t), d.y);
}

// log2 of a finite positive float as an explicit fp32 sequence (identical in the oracle):
// exponent + 2/ln2 * atanh((m-1)/(m+1)) by its odd series to z^9, m in [1, 2)
__device__ __forceinline__ float log2_explicit(float x)
{
    int e;
    const float m = 2.0f * frexpf(x, &e);
    const float z = (m - 1.0f) / (m + 1.0f);
    const float z2 = z * z;
    float p = 0.111111111f * z2;
    p = (p + 0.142857143f) * z2;
    p = (p + 0.2f) * z2;
    p = (p + 0.333333333f) * z2;
    p = (p + 1.0f) * z;
    return (float)(e - 1) + 2.88539008f * p;
}

__device__ __forceinline__ V3 pyramid_level(const SceneView &sc, int level, float s, float t)
{
    return environment_level(sc.env + sc.mip_offset[level], sc.mip_w[level], sc.mip_h[level], s, t);
}

// one probe at level-of-detail lambda: LINEAR at level 0 when magnifying, else the two nearest
// levels blended by frac(lambda), clamped to the 1x1 level
__device__ __forceinline__ V3 trilinear_probe(const SceneView &sc, float s, float t, float lambda)
{
    const int last = sc.mip_levels - 1;
    if (!(lambda > 0.0f))
        return pyramid_level(sc, 0, s, t);
    if (lambda >= (float)last)
        return pyramid_level(sc, last, s, t);
    const float fl = floorf(lambda);
    const int d1 = (int)fl;
    const float f = lambda - fl;
    return pyramid_level(sc, d1, s, t) * (1.0f - f) + pyramid_level(sc, d1 + 1, s, t) * f;
}

// textureGrad on the mip-mapped environment (fs:146; LINEAR_MIPMAP_LINEAR + 4x anisotropy,
// ray.cpp:503-509) by the rule fixed in oracle/shader_oracle.cpp's header: GL 3.1 scale factors,
// EXT_texture_filter_anisotropic's probe count / level / probe positions
__device__ __forceinline__ V3 texture_grad(const SceneView &sc, float s, float t, float dudx, float dvdx, float dudy, float dvdy)
{
    const float fw = (float)sc.env_w, fh = (float)sc.env_h;
    const float ax = dudx * fw, ay = dvdx * fh, bx = dudy * fw, by = dvdy * fh;
    const float px = sqrtf(ax * ax + ay * ay), py = sqrtf(bx * bx + by * by);
    const float pmax = sel_max(px, py), pmin = sel_min(px, py);
    if (!(pmax <= 3.0e38f))
        return pyramid_level(sc, sc.mip_levels - 1, s, t);
    if (!(pmax > 0.0f))
        return pyramid_level(sc, 0, s, t);
    float n = 4.0f;
    if (pmin > 0.0f)
        n = sel_min(ceilf(pmax / pmin), 4.0f);
    const float lambda = log2_explicit(pmax / n);
    const bool along_x = px >= py;
    const float mu = along_x ? dudx : dudy, mv = along_x ? dvdx : dvdy;
    V3 sum = mk(0, 0, 0);
    const int probes = (int)n;
    for (int i = 1; i <= probes; i++) {
        const float o = (float)i / (n + 1.0f) - 0.5f;
        sum = sum + trilinear_probe(sc, s + mu * o, t + mv * o, lambda);
    }
    return sum / n;
}

// filmic, fs:527-531
__device__ __forceinline__ float filmic(float c)
{
    const float x = sel_max(0.0f, c - 0.004f);
    return (x * (6.2f * x + 0.5f)) / (x * (6.2f * x + 1.7f) + 0.06f);
}

// intersect_and_shade's tail for a valid hit (fs:503-521 with shade, fs:362-377, ray_transfer + ray_reflect, fs:65-96,
// f_schlick_vr, fs:479-482): the facing normal, the reflected ray and the Fresnel-weighted specular colour.
// (profiles/isa_costs.py counts it in isolation.)
struct ShadedHit {
    V3 n, R, P2, object_specular;
};
__device__ __forceinline__ ShadedHit shade_hit(const SceneView &sc, const FrameView &fr, V3 spec, V3 P, V3 D, const Hit &hit)
{
    ShadedHit s;
    const V3 object_normal = interpolated_normal(sc, fr.normals_fp16 != 0, hit.which, hit.bu, hit.bv);
    s.n = xform(fr.object_normal_inverse, object_normal, 0.0f);
    if (dot3(s.n, D) > 0.0f)
        s.n = s.n * -1.0f;
    const V3 at = P + D * hit.t;                      // ray_transfer, fs:69
    s.R = D - s.n * (2.0f * dot3(s.n, D));            // reflect(), fs:86
    s.P2 = at + s.n * .0001f;                         // surface fudge, fs:87
    const float fresnel = pow5(dot3(D, s.R) * .5f + .5f);
    s.object_specular = spec + (mk(1.0f, 1.0f, 1.0f) - spec) * fresnel;   // f_schlick_vr, fs:479-482
    return s;
}

// Ray differentials of fs:58-63, carried only by the which == 2 view.
struct Differentials {
    V3 dPdx, dDdx, dPdy, dDdy;
};

// fs:621-625; pow(dot(d, d), 1.5) is evaluated as x * sqrt(x) (as in the oracle)
__device__ __forceinline__ void primary_differentials(const FrameView &fr, V3 d, Differentials &df)
{
    const V3 right = mk(fr.right[0], fr.right[1], fr.right[2]), up = mk(fr.up[0], fr.up[1], fr.up[2]);
    const float dd = dot3(d, d);
    const float dd15 = dd * sqrtf(dd);
    df.dPdx = mk(0, 0, 0);
    df.dDdx = (right * dd - d * dot3(d, right)) / dd15;
    df.dPdy = mk(0, 0, 0);
    df.dDdy = (up * dd - d * dot3(d, up)) / dd15;
}

// ray_transfer (fs:65-81) then ray_reflect (fs:83-96) applied to the differentials; the direction
// differentials lose a SCALAR per component in the reflection, exactly as the shader is written
__device__ __forceinline__ void bounce_differentials(Differentials &df, V3 D, float t, V3 n)
{
    const float dn = dot3(D, n);
    const V3 ax = df.dPdx + df.dDdx * t, ay = df.dPdy + df.dDdy * t;
    const float dtdx = -dot3(ax, n) / dn, dtdy = -dot3(ay, n) / dn;
    df.dPdx = ax + D * dtdx;
    df.dPdy = ay + D * dtdy;
    const float sx = 2 * dot3(df.dDdx, n), sy = 2 * dot3(df.dDdy, n);
    df.dDdx = mk(df.dDdx.x - sx, df.dDdx.y - sx, df.dDdx.z - sx);
    df.dDdy = mk(df.dDdy.x - sy, df.dDdy.y - sy, df.dDdy.z - sy);
}

// sample_environment when the ray carries differentials (fs:135-149): which == 2 draws
// |d(s,t)/dy| * 100 instead of a texel, which == 1 looks the texel up through textureGrad
__device__ __forceinline__ V3 environment_with_differentials(const SceneView &sc, const FrameView &fr, V3 D,
                                                             const Differentials &df)
{
    const float two_pi_rxz = 2.0f * kPi * (D.x * D.x + D.z * D.z);
    const float dudx = (D.x * df.dDdx.z - D.z * df.dDdx.x) / two_pi_rxz;
    const float dudy = (D.x * df.dDdy.z - D.z * df.dDdy.x) / two_pi_rxz;
    const float pi_ryy = kPi * sqrtf(1.0f - D.y * D.y);
    const float dvdx = df.dDdx.y / pi_ryy;
    const float dvdy = df.dDdy.y / pi_ryy;
    if (fr.which == 2)
        return mk(fabsf(dudy) * 1.0f * 100, fabsf(dvdy) * 1.0f * 100, 0.0f);
    float s, t;
    lookup_coords(D, s, t);
    return outside_acos_is_nan(texture_grad(sc, s, t, dudx, dvdx, dudy, dvdy), D.y);
}

// get_environment_map_coords, fs:121-125
__device__ __forceinline__ void environment_coords(V3 d, float &s, float &t)
{
    s = 1.0f + atan_yx(-d.z, d.x) / (2 * kPi);
    t = 1.0f - acos_clamped(sel_min(sel_max(d.y, -1.0f), 1.0f)) / kPi;
}

// trace(), fs:552-582, with intersect_and_shade (fs:484-522) and
// approximate_diffuse (fs:447-472) inlined.  Traversal::closest() runs
// group_intersect (fs:386-443) on an object-space ray.
// METAL: the caller guarantees a zero diffuse colour (the shader's metals, ray.cpp:698-704), so the
// diffuse / shadow-ray branch and the `accumulated` sum do not exist in this instance.
template <class Traversal, bool COUNT, bool DIFF, bool METAL = false>
__device__ __forceinline__ V3 trace_ray(const SceneView &sc, const FrameView &fr, Traversal &trav, V3 P, V3 D,
                                       RayCounters &rc, Differentials df = Differentials())
{
    V3 accumulated = mk(0, 0, 0);
    V3 modulation = mk(1, 1, 1);
    const V3 light = mk(fr.light_dir[0], fr.light_dir[1], fr.light_dir[2]);
    const V3 spec = mk(fr.specular_color[0], fr.specular_color[1], fr.specular_color[2]);
    const V3 diff = mk(fr.diffuse_color[0], fr.diffuse_color[1], fr.diffuse_color[2]);
    const bool has_diffuse = !METAL && diff.x > 0.0f && diff.y > 0.0f && diff.z > 0.0f;   // fs:570 (object_color is white)

    for (int bounce = 0; bounce < fr.bounce_count; bounce++) {
        Hit hit{kFar, -1.0f, 0.0f, 0.0f};
        trav.template closest<COUNT>(sc, fr, xform(fr.object_matrix, P, 1.0f), xform(fr.object_normal_matrix, D, 0.0f),
                                     hit, rc);
        if (hit.t >= kFar)
            break;
        if (hit.t == -1.0f) {   // iteration cap hit: the marker colour, unmodulated, no environment (fs:566-568)
            if (COUNT)
                rc.bad_hits++;
            return mk(1.0f, 0.0f, 0.0f);
        }
        if (COUNT)
            rc.shaded_hits++;
        const V3 object_normal = interpolated_normal(sc, fr.normals_fp16 != 0, hit.which, hit.bu, hit.bv);
        V3 n = xform(fr.object_normal_inverse, object_normal, 0.0f);
        if (dot3(n, D) > 0.0f)
            n = n * -1.0f;

        if (DIFF)
            bounce_differentials(df, D, hit.t, n);
        const V3 at = P + D * hit.t;                      // ray_transfer, fs:69
        const V3 R = D - n * (2.0f * dot3(n, D));         // reflect(), fs:86
        const V3 P2 = at + n * .0001f;                    // surface fudge, fs:87
        const float fresnel = pow5(dot3(D, R) * .5f + .5f);
        const V3 object_specular = spec + (mk(1.0f, 1.0f, 1.0f) - spec) * fresnel;   // f_schlick_vr, fs:479-482

        if (has_diffuse) {
            const float lcos = sel_max(0.0f, dot3(n, light));
            V3 irradiance = mk(0, 0, 0);
            bool lit = true;
            if (fr.cast_shadows) {
                Hit shadow{kFar, -1.0f, 0.0f, 0.0f};
                trav.template closest<COUNT>(sc, fr, xform(fr.object_matrix, P2, 1.0f),
                                             xform(fr.object_normal_matrix, light, 0.0f), shadow, rc);
                lit = shadow.t >= kFar;
            }
            if (lit)
                irradiance = irradiance + mk(1.0f, 1.0f, 1.0f) * lcos;
            accumulated = accumulated + modulation * diff * irradiance;
        }
        modulation = modulation * object_specular;
        P = P2;
        D = R;
    }
    if (COUNT)
        rc.env_lookups++;
    if (DIFF)
        return accumulated + modulation * environment_with_differentials(sc, fr, D, df);
    return accumulated + modulation * environment(sc, D);
}

__device__ __forceinline__ unsigned long long wave_sum(unsigned int v)
{
    unsigned long long s = v;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1)
        s += __shfl_down(s, off, 64);
    return s;
}

// the counting twins: a wave's tallies go to one of kCounterShards copies (the host sums them)
__device__ __forceinline__ void add_counters(const RayCounters &rc, DeviceCounters *counters)
{
    const unsigned int vals[7] = {rc.node_visits, rc.leaf_visits, rc.triangle_tests, rc.shaded_hits,
                                  rc.env_lookups, rc.traversals, rc.bad_hits};
    unsigned long long *dst = &counters[blockIdx.x % kCounterShards].node_visits;
#pragma unroll
    for (int k = 0; k < 7; k++) {
        const unsigned long long s = wave_sum(vals[k]);
        if ((threadIdx.x & 63u) == 0 && s)
            atomicAdd(dst + k, s);
    }
}

// Pixel (lx, ly) of the 16x16 pixel patch `patch` and where it goes: a patch of the whole frame (row-major output) or
// of the launch's k-th owned tile (packed output).
__device__ __forceinline__ void locate_patch_pixel(const FrameView &fr, unsigned int patch, int lx, int ly, int &px, int &py,
                                                   size_t &out_index, bool &store, bool &inside)
{
    store = true;
    if (fr.tile_stride == 0) {
        px = (int)(patch % (unsigned int)fr.patches_x) * 16 + lx;
        py = (int)(patch / (unsigned int)fr.patches_x) * 16 + ly;
        out_index = (size_t)py * fr.width + px;
        store = px < fr.width && py < fr.height;
    } else {
        const unsigned int k = patch / (unsigned int)fr.patches_per_unit;
        const unsigned int q = patch % (unsigned int)fr.patches_per_unit;
        const unsigned int pc = (unsigned int)fr.tile_phase_count;   // the set's k-th tile: period k / pc, phase offset k % pc
        const unsigned int tile = (k / pc) * (unsigned int)fr.tile_stride + (unsigned int)fr.tile_phase + k % pc;
        const int tx = (int)(tile % (unsigned int)fr.tiles_x), ty = (int)(tile / (unsigned int)fr.tiles_x);
        const int tlx = (int)(q % (unsigned int)fr.patches_x) * 16 + lx;
        const int tly = (int)(q / (unsigned int)fr.patches_x) * 16 + ly;
        px = tx * fr.tile_w + tlx;
        py = ty * fr.tile_h + tly;
        out_index = (size_t)k * fr.tile_w * fr.tile_h + (size_t)tly * fr.tile_w + tlx;
    }
    inside = px < fr.width && py < fr.height;
}

// Which pixel a thread renders: workgroup `patch` is a 16x16 pixel patch = four 8x8 wave tiles.
// (`wave` = which of the patch's four 8x8 tiles; by default the thread's wave within its 256-thread workgroup)
__device__ __forceinline__ void locate_pixel(const FrameView &fr, unsigned int patch, int &px, int &py, size_t &out_index,
                                             bool &store, bool &inside, unsigned int wave = 0xffffffffu)
{
    const unsigned int lane = threadIdx.x & 63u;
    if (wave == 0xffffffffu)
        wave = threadIdx.x >> 6;
    const int lx = (int)((wave & 1u) * 8u + (lane & 7u));
    const int ly = (int)((wave >> 1) * 8u + (lane >> 3));
    locate_patch_pixel(fr, patch, lx, ly, px, py, out_index, store, inside);
}

// One thread per pixel; a 256-thread workgroup covers a 16x16 patch as four
// 8x8 wave tiles so that the 64 rays of a wave stay spatially coherent.
// ONE_SAMPLE / METAL: instances for spp == 1 and for a zero diffuse colour (checked by the launcher):
// no sample loop, no radiance sum, no diffuse branch -- fewer live registers across the traversal.
template <class Traversal, bool COUNT, bool DIFF = false, bool ONE_SAMPLE = false, bool METAL = false>
__device__ __forceinline__ void trace_pixels(const SceneView &sc, const FrameView &fr, float4 *__restrict__ out,
                                             DeviceCounters *counters, Traversal &trav)
{
    const unsigned int patch = blockIdx.x;
#ifdef SHRAY_DIAGNOSTICS
    const unsigned int lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    // diagnostic build only (profiles/timeline.py): per-wave residency stamps, written to a
    // buffer nothing else reads
    const unsigned long long t_begin = __builtin_amdgcn_s_memrealtime();
#endif
    int px, py;
    size_t out_index;
    bool store, inside;
    locate_pixel(fr, patch, px, py, out_index, store, inside);

    RayCounters rc = {0, 0, 0, 0, 0, 0, 0};
    V3 result = mk(0, 0, 0);
    if (inside && fr.which == 3) {
        // fs:642-650: this pixel's own dY differential of the lookup coordinates; no trace, no tone map
        const float u = ((float)px + 0.5f) / (float)fr.width, v = ((float)py + 0.5f) / (float)fr.height;
        const V3 eye = unit(mk(fr.image_plane_width * (u - 0.5f), fr.image_plane_width * (v - 0.5f) * fr.aspect, -1.0f));
        const V3 D = unit(xform(fr.camera_normal_matrix, eye, 0.0f));
        Differentials df;
        primary_differentials(fr, D, df);
        float sb, tb, sa, ta;
        environment_coords(D - df.dDdy / 2.0f, sb, tb);
        environment_coords(D + df.dDdy / 2.0f, sa, ta);
        result = mk(fabsf(sa - sb) * 1.0f * 100, fabsf(ta - tb) * 1.0f * 100, 0.0f);
    } else if (inside && fr.which == 5) {
        // fs:654-673: 5 x 5 supersampled reference image around the interpolated varying direction
        const float u = ((float)px + 0.5f) / (float)fr.width, v = ((float)py + 0.5f) / (float)fr.height;
        const float hx = fr.image_plane_width * (1.0f - 0.5f), hy = fr.image_plane_width * (1.0f - 0.5f) * fr.aspect;
        const float corner_length = sqrtf(dot3(mk(hx, hy, -1.0f), mk(hx, hy, -1.0f)));
        const V3 eye = mk(fr.image_plane_width * (u - 0.5f), fr.image_plane_width * (v - 0.5f) * fr.aspect, -1.0f) / corner_length;
        const V3 P = xform(fr.camera_matrix, mk(0, 0, 0), 1.0f);
        const V3 dir = xform(fr.camera_normal_matrix, eye, 0.0f);
        const V3 right = mk(fr.right[0], fr.right[1], fr.right[2]), up = mk(fr.up[0], fr.up[1], fr.up[2]);
        V3 acc = mk(0, 0, 0);
        for (int i = 0; i < 5; i++) {
            for (int j = 0; j < 5; j++) {
                const float du = ((float)i / 5.0f - .5f), dv = ((float)j / 5.0f - .5f);
                const V3 D = unit(dir + right * (du * .2f) + up * (dv * .2f));
                acc = acc + trace_ray<Traversal, COUNT, false, METAL>(sc, fr, trav, P, D, rc);
            }
        }
        result = acc / 25.0f;
        if (fr.tonemap)
            result = mk(filmic(result.x), filmic(result.y), filmic(result.z));
    } else if (inside) {
        const float fw = (float)fr.width, fh = (float)fr.height, fn = (float)fr.spp;
        V3 sum = mk(0, 0, 0);
        const int samples = ONE_SAMPLE ? 1 : fr.spp;
        for (int s = 0; s < samples; s++) {
            // sub-pixel pattern of the oracle (oracle/shader_oracle.cpp header): centred Hammersley
            const float ox = ((float)s + 0.5f) / fn;
            const float oy = (float)__brev((unsigned int)s) * 2.3283064365386963e-10f + 0.5f / fn;
            const float u = ((float)px + ox) / fw;
            const float v = ((float)py + oy) / fh;
            // image_plane_ray + ray_transform, vs:39-60; normalize, fs:619
            const V3 eye = unit(mk(fr.image_plane_width * (u - 0.5f), fr.image_plane_width * (v - 0.5f) * fr.aspect, -1.0f));
            const V3 P = xform(fr.camera_matrix, mk(0, 0, 0), 1.0f);
            const V3 D = unit(xform(fr.camera_normal_matrix, eye, 0.0f));
            Differentials df;
            if (DIFF)
                primary_differentials(fr, D, df);
            const V3 radiance = trace_ray<Traversal, COUNT, DIFF, METAL>(sc, fr, trav, P, D, rc, df);
            sum = (ONE_SAMPLE || fr.spp == 1) ? radiance : sum + radiance;
        }
        result = (ONE_SAMPLE || fr.spp == 1) ? sum : sum / fn;
        if (fr.tonemap)
            result = mk(filmic(result.x), filmic(result.y), filmic(result.z));
    }
    if (store)
        out[out_index] = inside ? make_float4(result.x, result.y, result.z, 1.0f) : make_float4(0, 0, 0, 0);

#ifdef SHRAY_DIAGNOSTICS
    if (counters && lane == 0) {
        unsigned long long *tl = reinterpret_cast<unsigned long long *>(counters + kCounterShards) + 16ull * (blockIdx.x * 4u + wave);
        unsigned int hw_id;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw_id));
        unsigned int xcc_id;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc_id));
        tl[0] = t_begin;
        tl[1] = __builtin_amdgcn_s_memrealtime();
        tl[2] = ((unsigned long long)xcc_id << 32) | hw_id;
        tl[3] = rc.node_visits;
        for (int k = 0; k < 8; k++)
            tl[4 + k] = trav.diag_tally[k];
    }
#endif
    if (COUNT)
        add_counters(rc, counters);
}

}   // namespace shray
