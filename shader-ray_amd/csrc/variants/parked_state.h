// variants/parked_state.h -- -DSHRAY_PARK=1: the sample loop's state between traversals parked in LDS instead of registers.  Built,
// bit-identical, measured 1 % slower (profiles/EXPERIMENTS.md R5.2: a spilled word costs its two instructions, not its bytes); not part
// of the product build, where uniform_driver.h defines ParkedState as plain registers.
#pragma once

namespace shray {

// PARK (kernel_stack_common.h: parks_state): the words of the sample loop that are only touched BETWEEN traversals live in the
// wave's LDS slab `park` instead of in registers -- [lane][4] = {x, y, z of the running product `modulation` (zero diffuse colour) or
// of `accumulated` (with a diffuse term), the lane's channel sum}, and with a diffuse term one more row [lane] for the second
// channel sum.  A lane reads and writes its own words only; every read sits behind a compiler barrier (the traversal in between
// writes LDS through other pointers).
typedef float parked_f3 __attribute__((ext_vector_type(3)));
template <bool PARK, bool METAL>
struct ParkedState {
    float *slab;                       // this lane's four words (PARK only)
    V3 product = mk(1, 1, 1), sum = mk(0, 0, 0);   // modulation, accumulated: whichever is not parked -- or both
    float channel = 0.0f, channel2 = 0.0f;
    __device__ __forceinline__ void begin(float *park)
    {
        slab = PARK ? park + 4u * (threadIdx.x & 63u) : nullptr;
        if (PARK) {
            slab[3] = 0.0f;
            if (!METAL)
                park[256u + (threadIdx.x & 63u)] = 0.0f;
        }
    }
    __device__ __forceinline__ float *second(float *) const { return slab - 4u * (threadIdx.x & 63u) + 256u + (threadIdx.x & 63u); }
    __device__ __forceinline__ V3 read3() const
    {
        asm volatile("" ::: "memory");
        const parked_f3 v = *reinterpret_cast<const parked_f3 *>(slab);
        return mk(v.x, v.y, v.z);
    }
    __device__ __forceinline__ void write3(V3 v) const
    {
        parked_f3 w;
        w.x = v.x;
        w.y = v.y;
        w.z = v.z;
        *reinterpret_cast<parked_f3 *>(slab) = w;
    }
    __device__ __forceinline__ V3 modulation() const { return (PARK && METAL) ? read3() : product; }
    __device__ __forceinline__ void set_modulation(V3 v)
    {
        if (PARK && METAL)
            write3(v);
        else
            product = v;
    }
    __device__ __forceinline__ V3 accumulated() const { return (PARK && !METAL) ? read3() : sum; }
    __device__ __forceinline__ void set_accumulated(V3 v)
    {
        if (PARK && !METAL)
            write3(v);
        else
            sum = v;
    }
    __device__ __forceinline__ float channel_sum() const
    {
        if (!PARK)
            return channel;
        asm volatile("" ::: "memory");
        return slab[3];
    }
    __device__ __forceinline__ void set_channel_sum(float v)
    {
        if (PARK)
            slab[3] = v;
        else
            channel = v;
    }
    __device__ __forceinline__ float channel_sum2() const
    {
        if (!(PARK && !METAL))
            return channel2;
        asm volatile("" ::: "memory");
        return *second(nullptr);
    }
    __device__ __forceinline__ void set_channel_sum2(float v)
    {
        if (PARK && !METAL)
            *second(nullptr) = v;
        else
            channel2 = v;
    }
};

}   // namespace shray
