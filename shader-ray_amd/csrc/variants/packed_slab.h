// variants/packed_slab.h -- -DSHRAY_PK_SLAB=1: the slab test's six subtractions and six multiplications as six packed fp32
// instructions.  Built, bit-identical, measured slower (profiles/EXPERIMENTS.md R5.7); not part of the product build.
#pragma once

// Round 5 (R5.7), built, bit-identical, SLOWER, off (-DSHRAY_PK_SLAB=1): the six subtractions and six multiplications as six PACKED
// fp32 instructions (v_pk_add_f32 with its second source negated, v_pk_mul_f32: IEEE per component, the same values), on register
// PAIRS -- { entry.x, entry.y }, { exit.x, exit.y }, { entry.z, exit.z } as the record's loads leave them (DeviceNode), { P.x, P.y },
// { Y.x, Y.y } and { P.z, Y.z } of the ray, the z pair serving both halves of its instructions through op_sel; no pairing moves in
// the ISA.  Six issue slots fewer per visit -- and the headline 1 % slower, configs 3 / 5 2.3 % (profiles/r05/packed_slab_ab.txt):
// a packed instruction occupies the VALU as long as the two it replaces (R4.16) and its results arrive later in the visit's
// dependent chain (subtract -> multiply -> max3 -> compare).
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void slab_range_fast(const LaneTraversal &t, const float4 lo, const float4 hi, float &r0, float &r1)
{
    f32x2 e = {lo.x, lo.y}, x = {hi.x, hi.y}, z = {lo.z, hi.z};
    const f32x2 pxy = {t.P.x, t.P.y}, yxy = {t.Y.x, t.Y.y}, pzyz = {t.P.z, t.Y.z};
    asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(e) : "v"(e), "v"(pxy));
    asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(x) : "v"(x), "v"(pxy));
    asm("v_pk_add_f32 %0, %1, %2 op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]" : "=v"(z) : "v"(z), "v"(pzyz));
    asm("v_pk_mul_f32 %0, %1, %2" : "=v"(e) : "v"(e), "v"(yxy));
    asm("v_pk_mul_f32 %0, %1, %2" : "=v"(x) : "v"(x), "v"(yxy));
    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1]" : "=v"(z) : "v"(z), "v"(pzyz));
    // (a bare v_max_f32: fmaxf() of an asm result is canonicalised first -- one more instruction)
    float first;
    asm("v_max_f32_e32 %0, 0, %1" : "=v"(first) : "v"(e.x));
    r0 = fmaxf(fmaxf(first, e.y), z.x);
    r1 = fminf(fminf(x.x, x.y), z.y);
}
