// Does a wave64 vector instruction whose upper (or lower) 32 lanes are all masked off cost the SIMD one pass instead of two?
// Dependent-free v_fma chains, `waves` waves per SIMD on every CU, EXEC = all 64 / lower 32 / upper 32 / even lanes / one lane.
//   hipcc -O3 --offload-arch=gfx950 profiles/probes/valu_half.hip -o profiles/probes/valu_half && profiles/probes/valu_half
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

__global__ void __launch_bounds__(64) fma_loop(float *out, int iters, unsigned long long mask)
{
    const unsigned lane = threadIdx.x;
    float a0 = lane, a1 = lane + 1, a2 = lane + 2, a3 = lane + 3, a4 = lane + 4, a5 = lane + 5, a6 = lane + 6, a7 = lane + 7;
    const float m = 1.0000001f, c = 0.5f;
    if ((mask >> lane) & 1ull) {
        for (int i = 0; i < iters; i++) {
#pragma unroll
            for (int k = 0; k < 8; k++) {
                a0 = __builtin_fmaf(a0, m, c); a1 = __builtin_fmaf(a1, m, c); a2 = __builtin_fmaf(a2, m, c); a3 = __builtin_fmaf(a3, m, c);
                a4 = __builtin_fmaf(a4, m, c); a5 = __builtin_fmaf(a5, m, c); a6 = __builtin_fmaf(a6, m, c); a7 = __builtin_fmaf(a7, m, c);
            }
        }
    }
    out[blockIdx.x * 64 + lane] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}

int main()
{
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    float *out;
    const int max_blocks = cus * 4 * 8;
    hipMalloc(&out, sizeof(float) * 64 * max_blocks);
    const int iters = 4096;
    struct { const char *name; unsigned long long mask; } masks[] = {
        {"all 64 lanes", ~0ull}, {"lower 32", 0xffffffffull}, {"upper 32", 0xffffffff00000000ull},
        {"even lanes", 0x5555555555555555ull}, {"lower 16", 0xffffull}, {"lanes 0-15 and 32-47", 0x0000ffff0000ffffull}, {"one lane", 1ull}};
    for (int waves : {1, 2, 4, 8}) {
        for (auto &mk : masks) {
            const int blocks = cus * 4 * waves;
            hipEvent_t a, b;
            hipEventCreate(&a);
            hipEventCreate(&b);
            fma_loop<<<blocks, 64>>>(out, 16, mk.mask);
            hipDeviceSynchronize();
            hipEventRecord(a);
            fma_loop<<<blocks, 64>>>(out, iters, mk.mask);
            hipEventRecord(b);
            hipEventSynchronize(b);
            float ms;
            hipEventElapsedTime(&ms, a, b);
            const double insts = (double)iters * 64.0 * waves;   // per SIMD
            printf("%d waves per SIMD, %-22s %8.3f ms  %6.2f ns per wave-instruction per SIMD (2.4 GHz: %5.2f cycles)\n", waves, mk.name, ms,
                   ms * 1e6 / insts, ms * 1e6 / insts * 2.4);
        }
    }
    return 0;
}
