// Issue cost of the vector instructions a node visit is made of, on a gfx950 SIMD: eight independent chains per wave, 8 waves per
// SIMD on every CU, all 64 lanes; cycles per wave-instruction per SIMD at 2.4 GHz (v_mul_f32 is the yardstick: the probe does
// not know the clock the chip holds).
//   hipcc -O3 --offload-arch=gfx950 profiles/probes/valu_costs.hip -o profiles/probes/valu_costs && profiles/probes/valu_costs
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef float f2 __attribute__((ext_vector_type(2)));

#define CHAIN8(S) S(0) S(1) S(2) S(3) S(4) S(5) S(6) S(7)

#define I_MUL(j) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[j].x) : "v"(m.x));
#define I_MUL_S(j) asm volatile("v_mul_f32 %0, %1, %0" : "+v"(a[j].x) : "s"(sm));
#define I_SUB(j) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(a[j].x) : "v"(c.x));
#define I_ADD(j) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[j].x) : "v"(c.x));
#define I_FMA(j) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[j].x) : "v"(m.x), "v"(c.x));
#define I_FMA_S(j) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[j].x) : "s"(sm), "v"(c.x));
#define I_FMAC(j) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a[j].x) : "v"(m.x), "v"(c.x));
#define I_MIN(j) asm volatile("v_min_f32 %0, %0, %1" : "+v"(a[j].x) : "v"(m.x));
#define I_MAX(j) asm volatile("v_max_f32 %0, %0, %1" : "+v"(a[j].x) : "v"(m.x));
#define I_MIN3(j) asm volatile("v_min3_f32 %0, %0, %1, %2" : "+v"(a[j].x) : "v"(m.x), "v"(c.x));
#define I_MAX3(j) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(a[j].x) : "v"(m.x), "v"(c.x));
#define I_MED3(j) asm volatile("v_med3_f32 %0, %0, %1, %2" : "+v"(a[j].x) : "v"(m.x), "v"(c.x));
#define I_CND_VCC(j) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[j].x) : "v"(m.x));
#define I_CND_S(j) asm volatile("v_cndmask_b32 %0, %0, %1, %2" : "+v"(a[j].x) : "v"(m.x), "s"(mask));
#define I_CND_OUT(j) asm volatile("v_cndmask_b32 %0, %1, %2, %3" : "=v"(a[j].x) : "v"(c.x), "v"(m.x), "s"(mask));
#define I_CMP_VCC(j) asm volatile("v_cmp_lt_f32 vcc, %0, %1" : : "v"(a[j].x), "v"(m.x) : "vcc");
#define I_CMP_S(j) asm volatile("v_cmp_lt_f32 %0, %1, %2" : "=s"(out_mask) : "v"(a[j].x), "v"(m.x));
#define I_MOV(j) asm volatile("v_mov_b32 %0, %1" : "=v"(a[j].x) : "v"(m.x));
#define I_MOV_S(j) asm volatile("v_mov_b32 %0, %1" : "=v"(a[j].x) : "s"(sm));
#define I_MOV64(j) asm volatile("v_mov_b64 %0, %1" : "=v"(a[j]) : "v"(m));
#define I_MOV64_S(j) asm volatile("v_mov_b64 %0, %1" : "=v"(a[j]) : "s"(mask));
#define I_ADDU(j) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[j].x) : "v"(m.x));
#define I_SUBU_C(j) asm volatile("v_add_u32 %0, -1, %0" : "+v"(a[j].x));
#define I_AND(j) asm volatile("v_and_b32 %0, %0, %1" : "+v"(a[j].x) : "v"(m.x));
#define I_AND_S(j) asm volatile("v_and_b32 %0, %1, %0" : "+v"(a[j].x) : "s"(sm));
#define I_PK_FMA(j) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(a[j]) : "v"(m), "v"(c));
#define I_PK_MUL(j) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(a[j]) : "v"(m));
#define I_PK_ADD(j) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(a[j]) : "v"(c));
#define I_RCP(j) asm volatile("v_rcp_f32 %0, %0" : "+v"(a[j].x));
#define I_CMPX(j) asm volatile("v_cmp_class_f32 vcc, %0, %1" : : "v"(a[j].x), "v"(m.x) : "vcc");
#define I_BALLOTISH(j) asm volatile("v_cmp_ne_u32 %0, 0, %1" : "=s"(out_mask) : "v"(a[j].x));
#define I_READFIRST(j) asm volatile("v_readfirstlane_b32 %0, %1" : "=s"(out_s) : "v"(a[j].x));
#define I_DSW(j) asm volatile("ds_write_b32 %0, %1" : : "v"(lds_at), "v"(a[j].x) : "memory");
#define I_DSR(j) asm volatile("ds_read_b32 %0, %1" : "=v"(a[j].x) : "v"(lds_at) : "memory");
#define I_BPERM(j) asm volatile("ds_bpermute_b32 %0, %1, %0" : "+v"(a[j].x) : "v"(lds_at) : "memory");

#define I_CMPCND_VCC(j) asm volatile("v_cmp_lt_f32 vcc, %0, %1\n\tv_cndmask_b32 %0, %0, %2, vcc" : "+v"(a[j].x) : "v"(m.x), "v"(c.x) : "vcc");
#define I_CMPCND_S(j) asm volatile("v_cmp_lt_f32 %3, %0, %1\n\tv_cndmask_b32 %0, %0, %2, %3" : "+v"(a[j].x) : "v"(m.x), "v"(c.x), "s"(mask));
#define I_CND_VCC_SET(j) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[j].x) : "v"(m.x));
#define I_BFI(j) asm volatile("v_bfi_b32 %0, %1, %0, %2" : "+v"(a[j].x) : "v"(m.x), "v"(c.x));
#define I_MUL_LIT(j) asm volatile("v_mul_f32 %0, 0x3f800008, %0" : "+v"(a[j].x));
#define I_MUL_INL(j) asm volatile("v_mul_f32 %0, 1.0, %0" : "+v"(a[j].x));
#define I_SUB_S(j) asm volatile("v_sub_f32 %0, %1, %0" : "+v"(a[j].x) : "s"(sm));
#define I_XOR(j) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(a[j].x) : "v"(m.x));
#define I_OR(j) asm volatile("v_or_b32 %0, %0, %1" : "+v"(a[j].x) : "v"(m.x));
#define I_LSHL(j) asm volatile("v_lshlrev_b32 %0, 1, %0" : "+v"(a[j].x));
#define I_SUBU(j) asm volatile("v_sub_u32 %0, %0, %1" : "+v"(a[j].x) : "v"(m.x));
#define I_MINU(j) asm volatile("v_min_u32 %0, %0, %1" : "+v"(a[j].x) : "v"(m.x));
#define I_CMPU(j) asm volatile("v_cmp_lt_u32 %0, %1, %2" : "=s"(out_mask) : "v"(a[j].x), "v"(m.x));
#define I_CMP_E64_LIT(j) asm volatile("v_cmp_lt_f32 %0, %1, %2" : "=s"(out_mask) : "v"(a[j].x), "s"(sm));
#define I_AND_OR(j) asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(a[j].x) : "v"(m.x), "v"(c.x));
#define I_ADD3(j) asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(a[j].x) : "v"(m.x), "v"(c.x));
#define I_MUL_ABS(j) asm volatile("v_mul_f32 %0, |%0|, %1" : "+v"(a[j].x) : "v"(m.x));
#define I_SUB_NEG(j) asm volatile("v_sub_f32 %0, -%0, %1" : "+v"(a[j].x) : "v"(m.x));
#define I_MAX_SELF(j) asm volatile("v_max_f32 %0, %1, %1" : "=v"(a[j].x) : "v"(m.x));
#define I_MUL_OUT(j) asm volatile("v_mul_f32 %0, %1, %2" : "=v"(a[j].x) : "v"(m.x), "v"(c.x));
#define I_DSWU8(j) asm volatile("ds_write_b8 %0, %1" : : "v"(lds_at), "v"(a[j].x) : "memory");
#define I_SALU(j) asm volatile("s_and_b64 %0, %0, %1" : "+s"(out_mask) : "s"(mask) : "scc");
#define I_SMOV(j) asm volatile("s_mov_b32 %0, %1" : "=s"(out_s) : "s"(sm));

#define KINDS(X) \
    X(0, I_MUL, "v_mul_f32 v,v,v") X(1, I_MUL_S, "v_mul_f32 v,s,v") X(2, I_SUB, "v_sub_f32") X(3, I_ADD, "v_add_f32") \
    X(4, I_FMA, "v_fma_f32 v,v,v,v") X(5, I_FMA_S, "v_fma_f32 v,v,s,v") X(6, I_FMAC, "v_fmac_f32") X(7, I_MIN, "v_min_f32") \
    X(8, I_MAX, "v_max_f32") X(9, I_MIN3, "v_min3_f32") X(10, I_MAX3, "v_max3_f32") X(11, I_MED3, "v_med3_f32") \
    X(12, I_CND_VCC, "v_cndmask_b32 d,d,v,vcc") X(13, I_CND_S, "v_cndmask_b32 d,d,v,s[]") X(14, I_CND_OUT, "v_cndmask_b32 d,v,v,s[]") \
    X(15, I_CMP_VCC, "v_cmp_lt_f32 vcc") X(16, I_CMP_S, "v_cmp_lt_f32 s[]") X(17, I_MOV, "v_mov_b32 v,v") X(18, I_MOV_S, "v_mov_b32 v,s") \
    X(19, I_MOV64, "v_mov_b64 v,v") X(20, I_MOV64_S, "v_mov_b64 v,s") X(21, I_ADDU, "v_add_u32 v,v,v") X(22, I_SUBU_C, "v_add_u32 v,-1,v") \
    X(23, I_AND, "v_and_b32 v,v,v") X(24, I_AND_S, "v_and_b32 v,s,v") X(25, I_PK_FMA, "v_pk_fma_f32") X(26, I_PK_MUL, "v_pk_mul_f32") \
    X(27, I_PK_ADD, "v_pk_add_f32") X(28, I_RCP, "v_rcp_f32") X(29, I_CMPX, "v_cmp_class_f32 vcc") X(30, I_BALLOTISH, "v_cmp_ne_u32 s[],0,v") \
    X(31, I_READFIRST, "v_readfirstlane_b32") X(32, I_DSW, "ds_write_b32") X(33, I_DSR, "ds_read_b32") X(34, I_BPERM, "ds_bpermute_b32") \
    X(35, I_CMPCND_VCC, "v_cmp vcc + v_cndmask vcc (PAIR)") X(36, I_CMPCND_S, "v_cmp s[] + v_cndmask s[] (PAIR)") X(37, I_BFI, "v_bfi_b32") \
    X(38, I_MUL_LIT, "v_mul_f32 v,literal,v") X(39, I_MUL_INL, "v_mul_f32 v,1.0,v") X(40, I_SUB_S, "v_sub_f32 v,s,v") X(41, I_XOR, "v_xor_b32") \
    X(42, I_OR, "v_or_b32") X(43, I_LSHL, "v_lshlrev_b32 v,1,v") X(44, I_SUBU, "v_sub_u32") X(45, I_MINU, "v_min_u32") X(46, I_CMPU, "v_cmp_lt_u32 s[]") \
    X(47, I_CMP_E64_LIT, "v_cmp_lt_f32 s[],v,s") X(48, I_AND_OR, "v_and_or_b32") X(49, I_ADD3, "v_add3_u32") X(50, I_MUL_ABS, "v_mul_f32 v,|v|,v (e64)") \
    X(51, I_SUB_NEG, "v_sub_f32 v,-v,v (e64)") X(52, I_MAX_SELF, "v_max_f32 d,v,v") X(53, I_MUL_OUT, "v_mul_f32 d,v,v") \
    X(56, I_DSWU8, "ds_write_b8") X(57, I_SALU, "s_and_b64 (scalar)") X(58, I_SMOV, "s_mov_b32 (scalar)")

template <int KIND>
__global__ void __launch_bounds__(64) loop(float *out, int iters, float sm_in, unsigned long long mask_in)
{
    __shared__ float lds[64 * 8];
    const unsigned lane = threadIdx.x;
    f2 a[8];
    for (int k = 0; k < 8; k++)
        a[k] = f2{(float)(lane + k) + 1.0f, (float)(lane + k) + 1.5f};
    f2 m = f2{1.0000001f, 0.9999999f}, c = f2{0.5f, 0.25f};
    float sm = sm_in;
    unsigned long long mask = mask_in, out_mask = 0;
    unsigned out_s = 0;
    unsigned lds_at = (unsigned)(size_t)(&lds[lane]) ;
    lds[lane] = 1.0f;
    asm volatile("" : "+v"(m), "+v"(c), "+s"(sm), "+s"(mask), "+v"(lds_at));
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int k = 0; k < 8; k++) {
#define X(id, STMT, name) if (KIND == id) { CHAIN8(STMT) }
            KINDS(X)
#undef X
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    float s = (float)(out_mask & 1) + (float)out_s;
    for (int k = 0; k < 8; k++)
        s += a[k].x + a[k].y;
    out[blockIdx.x * 64 + lane] = s;
}

template <int KIND>
void run(const char *name, float *out, int cus)
{
    const int iters = 2048;
    for (int waves : {1, 8}) {
        const int blocks = cus * 4 * waves;
        hipEvent_t a, b;
        (void)hipEventCreate(&a);
        (void)hipEventCreate(&b);
        loop<KIND><<<blocks, 64>>>(out, 16, 1.0000001f, 0x5555aaaa3333ccccull);
        (void)hipDeviceSynchronize();
        (void)hipEventRecord(a);
        loop<KIND><<<blocks, 64>>>(out, iters, 1.0000001f, 0x5555aaaa3333ccccull);
        (void)hipEventRecord(b);
        (void)hipEventSynchronize(b);
        float ms;
        (void)hipEventElapsedTime(&ms, a, b);
        const double insts = (double)iters * 64.0 * waves;   // per SIMD
        printf("%-28s %d waves per SIMD  %8.3f ms  %6.2f ns per wave-instruction per SIMD (2.4 GHz: %5.2f cycles)\n", name, waves, ms,
               ms * 1e6 / insts, ms * 1e6 / insts * 2.4);
        fflush(stdout);
    }
}

int main()
{
    hipDeviceProp_t prop;
    (void)hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    float *out;
    (void)hipMalloc(&out, sizeof(float) * 64 * cus * 4 * 8);
    run<0>("warm-up (v_mul_f32)", out, cus);
#define X(id, STMT, name) run<id>(name, out, cus);
    KINDS(X)
#undef X
    return 0;
}
