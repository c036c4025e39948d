// exact_div.h -- IEEE-exact fp32 division by a per-ray constant, in 5 VALU ops.
//
// The reference's slab test divides by the ray direction six times per node
// visit (raytracer.es.fs:204-213).  A correctly rounded a / b costs ~12 VALU
// instructions when the compiler emits it (v_div_scale x2, v_rcp, 6 FMAs,
// v_div_fmas, v_div_fixup); b is the same for every node a ray visits, so its
// correctly rounded reciprocal y = RN(1 / b) is computed once per ray and each
// quotient becomes
//      q0 = a * y
//      r0 = fma(-b, q0, a)      q1 = fma(r0, y, q0)     (q1 within 1 ulp of a / b)
//      r1 = fma(-b, q1, a)      q  = fma(r1, y, q1)     (exact residual; Markstein's
//                                                        theorem: q == RN(a / b))
// -- the same two refinement steps the compiler's own sequence performs, minus
// the scaling that protects it against overflow / underflow of intermediates.
// The scaling is replaced by range conditions under which no intermediate can
// overflow, underflow or lose bits (exponent ranges):
//      |b| in [2^-40, 2^20]                        (checked per ray)
//      a == 0  or  |a| in [2^-93, 2^61]            (follows from: every box
//        coordinate and every ray-origin component is 0 or has magnitude in
//        [2^-70, 2^60]; checked per scene and per ray)
// A ray (or scene) outside these ranges takes the true-division path.
// Equality with true division is verified on the GPU over ~10^9 operand pairs by
// shray_selftest (tests/test_gpu_selftest.py) and on the CPU with fmaf.
#pragma once

#include <hip/hip_runtime.h>

namespace shray {

__device__ __forceinline__ float div_by_constant(float a, float b, float y)
{
    const float q0 = a * y;
    const float r0 = __builtin_fmaf(-b, q0, a);
    const float q1 = __builtin_fmaf(r0, y, q0);
    const float r1 = __builtin_fmaf(-b, q1, a);
    return __builtin_fmaf(r1, y, q1);
}

// Four-operation form with the reciprocal split in two: y = RN(1 / b) and its residual
// yl = fma(-b, y, 1) * y  (so y + yl = 1/b to ~2^-46 relative).  a*(y + yl), evaluated as
// fma(a, yl, a*y), is already within one ulp of a / b, which is all Markstein's final step
// needs:   q1 = fma(a, yl, a * y);   r = fma(-b, q1, a);   q = fma(r, y, q1) == RN(a / b).
// Same operand ranges as div_by_constant; verified by the same self-test.
__device__ __forceinline__ float reciprocal_residual(float b, float y) { return __builtin_fmaf(-b, y, 1.0f) * y; }
__device__ __forceinline__ float div_by_constant4(float a, float b, float y, float yl)
{
    const float q1 = __builtin_fmaf(a, yl, a * y);
    const float r = __builtin_fmaf(-b, q1, a);
    return __builtin_fmaf(r, y, q1);
}

// The correctly rounded reciprocal in three operations: the hardware's v_rcp_f32 (within 1 ulp) and ONE Newton step with
// fused multiply-adds,   r = rcp(x);  e = fma(-x, r, 1);  r = fma(e, r, r),   equals RN(1 / x) for EVERY float with
// 2^-100 <= |x| < 2^100 -- verified exhaustively, all 3.4e9 of them, by shray_selftest_reciprocal
// (tests/test_gpu_selftest.py); the compiler's own 1.0f / x is ten (v_div_scale x2, v_rcp, five FMAs, v_div_fmas,
// v_div_fixup: the scaling that protects intermediates outside that range).  Callers establish the range (or divide).
__device__ __forceinline__ float reciprocal_in_range(float x)
{
    const float r = __builtin_amdgcn_rcpf(x);
    return __builtin_fmaf(__builtin_fmaf(-x, r, 1.0f), r, r);
}

// exponent-field tests on the raw bits (NaN / inf fail every one of them)
__device__ __forceinline__ bool magnitude_in(float v, int lo_exp, int hi_exp)   // 2^lo <= |v| < 2^(hi+1)
{
    const int e = (int)((__float_as_uint(v) >> 23) & 0xffu) - 127;
    return e >= lo_exp && e <= hi_exp;
}
__device__ __forceinline__ bool divisor_in_range(float b) { return magnitude_in(b, -40, 19); }
__device__ __forceinline__ bool reciprocal_domain(float x) { return magnitude_in(x, -100, 99); }
__device__ __forceinline__ bool coordinate_in_range(float c) { return c == 0.0f || magnitude_in(c, -70, 59); }

// A divisor many dividends share (a frame's width and height): RN(1 / b) and its residual once, then div_by_constant4's four
// operations per quotient instead of the compiler's eleven -- when b is inside the proven range and the CALLER vouches for
// the dividends (0, or a magnitude in [2^-93, 2^61]); otherwise the true division.  `b` is wave-uniform where this is used,
// so `exact` is a scalar and the choice a scalar branch.
struct SharedDivisor {
    float b, y, yl;
    bool exact;
};
__device__ __forceinline__ SharedDivisor shared_divisor(float b)
{
    SharedDivisor d;
    d.b = b;
    d.exact = divisor_in_range(b);
    d.y = reciprocal_in_range(b);           // (not looked at when b is outside the range)
    d.yl = reciprocal_residual(b, d.y);
    return d;
}
__device__ __forceinline__ float divide_by_shared(float a, const SharedDivisor &d)
{
#ifndef SHRAY_COST_MAIN_PATH     // profiles/isa_costs.hip counts the path every wave takes
    if (__builtin_expect(!d.exact, 0)) {
        asm volatile("; a shared divisor outside exact_div.h's range" ::: "memory");   // keeps this a branch
        return a / d.b;
    }
#endif
    return div_by_constant4(a, d.b, d.y, d.yl);
}

}   // namespace shray
