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


// shray_probe_vector_cache: what a CU's vector L1 delivers to the access pattern of a node visit.  One-wave workgroups,
// every lane reads whole 32-byte records (two global_load_dwordx4, as load_packed_node does) of a table of `records`
// records; the 64 lanes of a wave-instruction pick theirs among `spread` consecutive records around a base that changes
// pseudo-randomly from visit to visit (spread 1: the wave at one node; 64: every lane elsewhere), eight visits in flight
// per wave.  Nothing else is done with the data than a sum, so the rate is the memory pipeline's, not the arithmetic's.
// BYTES: what a lane reads of its record per visit: 32 = the node's two 16-byte loads; 16, 12, 8, 4 = one load of that width
template <int BYTES>
__global__ void __launch_bounds__(64, 7) vector_cache_probe_kernel(const float4 *__restrict__ table, uint32_t record_mask, uint32_t spread_mask,
                                                                   uint32_t visits, unsigned long long lanes, float *__restrict__ out)
{
    const uint32_t lane = threadIdx.x;
    uint32_t base = (blockIdx.x * 2654435761u) ^ 0x9e3779b9u;
    // this lane's place inside the spread: pseudo-random, or (bit 31 of spread_mask) runs of neighbouring lanes together
    const bool runs = (spread_mask >> 31) != 0u;
    spread_mask &= 0x7fffffffu;
    const uint32_t mine = runs ? (lane * (spread_mask + 1u)) >> 6 : (lane * 0x61c88647u) >> 7;
    float acc0 = 0.0f, acc1 = 0.0f;
    if (!((lanes >> lane) & 1ull))
        visits = 0;      // only the lanes of the mask take part
    typedef float f3 __attribute__((ext_vector_type(3)));
    typedef f3 __attribute__((aligned(4), may_alias)) packed_f3;
    for (uint32_t v = 0; v < visits; v += 8u) {
        float4 lo[8], hi[8];
#pragma unroll
        for (int k = 0; k < 8; k++) {
            base = base * 1664525u + 1013904223u;        // uniform over the wave (a scalar sequence)
            const uint32_t record = ((base >> 9) + (mine & spread_mask)) & record_mask;
            const char *p = reinterpret_cast<const char *>(table) + (record << 5);
            lo[k] = hi[k] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
            if (BYTES >= 16)
                lo[k] = *reinterpret_cast<const float4 *>(p);
            if (BYTES == 32)
                hi[k] = *reinterpret_cast<const float4 *>(p + 16);
            if (BYTES == 12) {
                const f3 w = *reinterpret_cast<const packed_f3 *>(p);
                lo[k] = make_float4(w.x, w.y, w.z, 0.0f);
            }
            if (BYTES == 8) {
                const float2 w = *reinterpret_cast<const float2 *>(p);
                lo[k] = make_float4(w.x, w.y, 0.0f, 0.0f);
            }
            if (BYTES == 4)
                lo[k].x = *reinterpret_cast<const float *>(p);
        }
#pragma unroll
        for (int k = 0; k < 8; k++) {
            acc0 += lo[k].x + hi[k].w + lo[k].y;
            acc1 += lo[k].w + hi[k].x + lo[k].z;
        }
    }
    out[blockIdx.x * 64u + lane] = acc0 + acc1;
}

hipError_t launch_vector_cache_probe(const float4 *table, uint32_t records, uint32_t spread, uint32_t visits, uint32_t waves, unsigned long long lanes,
                                     int bytes_per_lane, float *out, hipStream_t stream)
{
    const uint32_t spread_arg = ((spread & 0x7fffffffu) - 1u) | (spread & 0x80000000u);
#define SHRAY_PROBE(B) hipLaunchKernelGGL(vector_cache_probe_kernel<B>, dim3(waves), dim3(64), 0, stream, table, records - 1u, spread_arg, visits, lanes, out)
    switch (bytes_per_lane) {
    case 32: SHRAY_PROBE(32); break;
    case 16: SHRAY_PROBE(16); break;
    case 12: SHRAY_PROBE(12); break;
    case 8: SHRAY_PROBE(8); break;
    case 4: SHRAY_PROBE(4); break;
    default: return hipErrorInvalidValue;
    }
#undef SHRAY_PROBE
    return hipGetLastError();
}

}   // namespace shray
