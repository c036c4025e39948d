// kernel_selftest.hip -- device-side check that div_by_constant (exact_div.h) equals
// true IEEE division wherever the traversal uses it.
#include "exact_div.h"
#include "launch.h"

namespace shray {

__device__ __forceinline__ unsigned long long mix64(unsigned long long z)
{
    z += 0x9e3779b97f4a7c15ull;
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
    return z ^ (z >> 31);
}

// a float with a random sign and 23-bit fraction and an exponent drawn from [lo, hi]
__device__ __forceinline__ float random_in_exponent_range(unsigned long long bits, int lo, int hi, unsigned int shape)
{
    unsigned int frac = (unsigned int)bits & 0x7fffffu;
    // structured fractions: all ones, all zeros, single bits, near-all-ones
    switch (shape & 7u) {
    case 1: frac = 0x7fffffu; break;
    case 2: frac = 0u; break;
    case 3: frac = 1u << ((bits >> 40) % 23); break;
    case 4: frac = 0x7fffffu ^ (1u << ((bits >> 40) % 23)); break;
    default: break;
    }
    const int e = lo + (int)((bits >> 24) % (unsigned long long)(hi - lo + 1));
    const unsigned int sign = (unsigned int)(bits >> 63) << 31;
    return __uint_as_float(sign | ((unsigned int)(e + 127) << 23) | frac);
}

__global__ void __launch_bounds__(256) division_selftest_kernel(unsigned long long pairs, unsigned long long seed,
                                                                unsigned long long *mismatches)
{
    unsigned long long bad = 0;
    const unsigned long long stride = (unsigned long long)gridDim.x * blockDim.x;
    for (unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; i < pairs; i += stride) {
        const unsigned long long h0 = mix64(seed + 2 * i), h1 = mix64(seed + 2 * i + 1);
        // divisor: the ray-direction range; numerator: zero, or a box-minus-origin difference
        const float b = random_in_exponent_range(h0, -40, 19, (unsigned int)(h0 >> 48));
        float a = random_in_exponent_range(h1, -93, 60, (unsigned int)(h1 >> 48));
        if (((h1 >> 56) & 63u) == 0u)
            a = 0.0f;
        if (((h1 >> 56) & 63u) == 1u)
            a = b * random_in_exponent_range(mix64(h1), -3, 3, (unsigned int)(h1 >> 51));   // quotient near a "nice" value
        const float y = 1.0f / b;
        const float fast = div_by_constant(a, b, y);
        const float fast4 = div_by_constant4(a, b, y, reciprocal_residual(b, y));
        const float exact = a / b;
        const bool same = __float_as_uint(fast) == __float_as_uint(exact) || (fast == 0.0f && exact == 0.0f);
        const bool same4 = __float_as_uint(fast4) == __float_as_uint(exact) || (fast4 == 0.0f && exact == 0.0f);
        bad += (same && same4) ? 0ull : 1ull;
    }
    if (bad)
        atomicAdd(mismatches, bad);
}

hipError_t launch_division_selftest(unsigned long long pairs, unsigned long long seed, unsigned long long *mismatches,
                                    hipStream_t stream)
{
    hipLaunchKernelGGL(division_selftest_kernel, dim3(2048), dim3(256), 0, stream, pairs, seed, mismatches);
    return hipGetLastError();
}


// shray_selftest_reciprocal: reciprocal_in_range against the compiler's correctly rounded 1.0f / x on every float of its
// domain (exponent field 27 .. 226, both signs: 200 * 2^23 * 2 values)
__global__ void __launch_bounds__(256) reciprocal_selftest_kernel(unsigned long long *mismatches)
{
    const unsigned long long stride = (unsigned long long)gridDim.x * blockDim.x;
    unsigned long long bad = 0;
    for (unsigned long long u = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; u < (1ull << 32); u += stride) {
        const float x = __uint_as_float((uint32_t)u);
        if (!reciprocal_domain(x))
            continue;
        if (__float_as_uint(reciprocal_in_range(x)) != __float_as_uint(1.0f / x))
            bad++;
    }
    if (bad)
        atomicAdd(mismatches, bad);
}

hipError_t launch_reciprocal_selftest(unsigned long long *mismatches, hipStream_t stream)
{
    hipLaunchKernelGGL(reciprocal_selftest_kernel, dim3(4096), dim3(256), 0, stream, mismatches);
    return hipGetLastError();
}

}   // namespace shray
