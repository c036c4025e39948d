// bvh_build.hip -- make_bvh (reference bvh.cpp:288-358, with get_best_split :198-247, sah :107-120, partition :249-286 and
// make_leaf :122-135) on the GPU, behind the C ABI of include/shader_ray_hip.h (shray_bvh_build_device ...).  SURVEY section 8(f)
// rank 3, second half: the first half, the flattener, is flatten.hip, and this file's output -- the tree as pre-order arrays, the
// triangles in post-build order -- is what shray_flatten_device takes.
//
// The reference builds depth-first, one node at a time: bounds of the range, bins over the longest barycentre axis, the SAH
// sweep, an in-place two-ended exchange partition, recurse.  Everything a node's build reads is inside its own triangle
// range, so the nodes of one LEVEL are independent; here a level is built at once, by kernels over all triangles (which know
// their node) and over all nodes of the level:
//   bounds     the vertex box and the barycentre box of every node: min / max are order-independent, so atomics on the floats'
//              order-preserving integer keys give the reference's sequential result (a wave that lies inside one node reduces
//              first and issues one atomic per word)
//   bins       min(40, 2 count) bins over the vertex box along the axis: the same float expression per triangle
//              (bvh.cpp:148-170: floor((b - lo) * bins / (hi - lo)), x86's float -> int conversion), counts and boxes by
//              atomics (a workgroup that lies inside one node gathers in LDS first)
//   SAH sweep  one thread per node, the reference's loop with its operand order (bvh.cpp:107-120, :172-196)
//   partition  the two-ended exchange of bvh.cpp:249-286 swaps the k-th element from the left that belongs right with the k-th
//              from the right that belongs left, k = 1, 2, ... -- a pairing a prefix sum of the "belongs left" flags yields
//              for all elements at once; the swaps themselves touch disjoint pairs.  The order inside each half, and so the
//              order of a leaf's triangles (which decides ties between equal hit distances, fs:333-340), is the reference's
// and the tree is renumbered from creation (breadth-first) order to the pre-order shray_host_export_tree gives.  All float
// expressions keep the host builder's operand order (host/bvh.cpp; -ffp-contract=off, correctly rounded division), so the
// tree, the boxes and the triangle order equal the host's bit for bit (tests/test_gpu_bvh_build.py).  Not reproduced: the order
// in which the sequential min / max meets a +0 and a -0 of the same box plane (the keys put -0 below +0).
#include <hip/hip_runtime.h>
#include <rocprim/device/device_scan.hpp>
#include <rocprim/functional.hpp>

#include <algorithm>
#include <cfloat>
#include <cmath>
#include <climits>
#include <cstdint>
#include <cstring>
#include <memory>
#include <vector>

#include "device_tree_internal.h"
#include "shader_ray_hip.h"

extern "C" int shrayi_fail(int code, const char *message);   // capi.hip: sets shray_last_error()

namespace {

constexpr int kMaxBins = 40;          // bvh.cpp:141
constexpr int kBlock = 256;
constexpr float kBump = .00001f;      // box3d::add(point), vectormath.h:189-195

struct DeviceArray {
    void *p = nullptr;
    ~DeviceArray()
    {
        if (p)
            (void)hipFree(p);
    }
    hipError_t alloc(size_t bytes) { return hipMalloc(&p, bytes ? bytes : 16); }
    template <class T>
    T *as() const { return static_cast<T *>(p); }
};

// order-preserving keys of floats: key(a) < key(b) <=> a < b (and -0 < +0)
__host__ __device__ inline uint32_t float_key(float f)
{
    const uint32_t b = __builtin_bit_cast(uint32_t, f);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__host__ __device__ inline float key_float(uint32_t k)
{
    const uint32_t b = (k & 0x80000000u) ? (k & 0x7fffffffu) : ~k;
    return __builtin_bit_cast(float, b);
}

// float -> int the way x86's cvttss2si does it (host/bvh.cpp: truncate_like_x86): what is not representable is INT_MIN
__device__ inline int truncate_like_x86(float f)
{
    if (!(f >= -2147483648.0f && f < 2147483648.0f))
        return INT_MIN;
    return (int)f;
}

enum : int { NODE_LEAF = 0, NODE_CANDIDATE = 1, NODE_SPLIT = 2, NODE_LARGE_LEAF = 3 };

// per-triangle state, structure of arrays; position p = the triangle's place in the (evolving) build order
struct Triangles {
    int count;
    float *box;        // [6][count]: min x, y, z, max x, y, z
    float *bary;       // [3][count]
    int *original;     // [count]: which input triangle sits at position p
    int *node;         // [count]: index of p's node in the current level's list, -1 once p is inside a leaf
    int *flag;         // [count]: 1 = barycentre below its node's split plane
    int *below;        // [count]: inclusive prefix sum of flag
    int *left_at, *right_at;   // [count]: per node, the k-th misplaced position from the left / from the right
};

// the nodes of the level being built
struct Level {
    int *start, *count, *id;         // the triangle range and the node's number in creation order
    uint32_t *vertex_key, *bary_key; // [n][6] boxes as keys (min x, y, z, max x, y, z)
    int *state, *axis, *bins, *slot; // slot: the candidate's bin storage
    float *lo, *hi, *plane;
    int *mid;                        // triangles below the plane
    int *split, *child_offset;       // 1 for nodes that split; exclusive prefix sum of it
};

struct Tree {   // creation order; 2 count - 1 nodes at most
    int *parent, *negative, *positive, *start, *triangles, *level;
    float *box, *direction;
};

__global__ void gather_pair(const int *a, const int *b, int *out)
{
    out[0] = *a;
    out[1] = *b;
}

__global__ void prepare_triangles(Triangles t, const int *__restrict__ triangle_vertices, const float *__restrict__ vertex_data, int stride)
{
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= t.count)
        return;
    const float *a = vertex_data + (size_t)triangle_vertices[3 * p] * stride;
    const float *b = vertex_data + (size_t)triangle_vertices[3 * p + 1] * stride;
    const float *c = vertex_data + (size_t)triangle_vertices[3 * p + 2] * stride;
    for (int k = 0; k < 3; k++) {
        // indexed_triangle (geometry.h:79-90): box.add(a, b, c) -- each point bumped by 1e-5 --, barycenter = (a + b + c) / 3
        float lo = FLT_MAX, hi = -FLT_MAX;
        lo = fminf(lo, a[k] - kBump); hi = fmaxf(hi, a[k] + kBump);
        lo = fminf(lo, b[k] - kBump); hi = fmaxf(hi, b[k] + kBump);
        lo = fminf(lo, c[k] - kBump); hi = fmaxf(hi, c[k] + kBump);
        t.box[(size_t)k * t.count + p] = lo;
        t.box[(size_t)(3 + k) * t.count + p] = hi;
        t.bary[(size_t)k * t.count + p] = ((a[k] + b[k]) + c[k]) / 3.0f;
    }
    t.original[p] = p;
    t.node[p] = 0;
}

__global__ void clear_level(Level l, int n)
{
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n)
        return;
    for (int k = 0; k < 3; k++) {
        l.vertex_key[6 * j + k] = l.bary_key[6 * j + k] = float_key(FLT_MAX);         // box3d(): boxmin = max float,
        l.vertex_key[6 * j + 3 + k] = l.bary_key[6 * j + 3 + k] = float_key(-FLT_MAX);   // boxmax = -max float
    }
}

__device__ inline uint32_t wave_min_u32(uint32_t v)
{
    for (int s = 32; s >= 1; s >>= 1)
        v = min(v, (uint32_t)__shfl_xor((int)v, s, 64));
    return v;
}
__device__ inline uint32_t wave_max_u32(uint32_t v)
{
    for (int s = 32; s >= 1; s >>= 1)
        v = max(v, (uint32_t)__shfl_xor((int)v, s, 64));
    return v;
}

// vertex_box.add(tris[k].box), bary_box.add(tris[k].barycenter)                    (bvh.cpp:305-311)
__global__ void node_bounds(Triangles t, Level l)
{
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    const int j = p < t.count ? t.node[p] : -1;
    uint32_t key[12];
    for (int k = 0; k < 12; k++)
        key[k] = k % 6 < 3 ? 0xffffffffu : 0u;
    if (j >= 0) {
        for (int k = 0; k < 3; k++) {
            key[k] = float_key(t.box[(size_t)k * t.count + p]);
            key[3 + k] = float_key(t.box[(size_t)(3 + k) * t.count + p]);
            const float b = t.bary[(size_t)k * t.count + p];
            key[6 + k] = float_key(b - kBump);
            key[9 + k] = float_key(b + kBump);
        }
    }
    // a wave whose 64 positions lie in one node (every wave of the upper levels): one atomic per word instead of 64
    const int first = __builtin_amdgcn_readfirstlane(j);
    const bool uniform = __builtin_amdgcn_ballot_w64(j != first) == 0ull && __builtin_amdgcn_ballot_w64(true) == ~0ull;
    if (uniform) {
        if (first < 0)
            return;
        for (int k = 0; k < 12; k++)
            key[k] = k % 6 < 3 ? wave_min_u32(key[k]) : wave_max_u32(key[k]);
        if ((threadIdx.x & 63) != 0)
            return;
    } else if (j < 0)
        return;
    for (int k = 0; k < 6; k++) {
        if (k < 3) {
            atomicMin(&l.vertex_key[6 * j + k], key[k]);
            atomicMin(&l.bary_key[6 * j + k], key[6 + k]);
        } else {
            atomicMax(&l.vertex_key[6 * j + k], key[k]);
            atomicMax(&l.bary_key[6 * j + k], key[6 + k]);
        }
    }
}

struct Options {
    int max_depth, leaf_max;
    float ctrav, cisec;
};

// bins of the candidates: [slot][kMaxBins] counts and [slot][kMaxBins][6] box keys
struct Bins {
    int *n;
    uint32_t *key;
    int *used;      // slots handed out so far (one counter)
};

// leaf test, split axis, the binning's parameters                                  (bvh.cpp:299, :317-326, :148-152)
__global__ void decide_nodes(Level l, int n, int level, Options o, Tree tree, Bins bins)
{
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n)
        return;
    float vbox[6], bbox[6];
    for (int k = 0; k < 6; k++) {
        vbox[k] = key_float(l.vertex_key[6 * j + k]);
        bbox[k] = key_float(l.bary_key[6 * j + k]);
        tree.box[(size_t)6 * l.id[j] + k] = vbox[k];     // a branch keeps its vertex box; a leaf's box is the union of its
    }                                                     // triangles' bumped corners (group.cpp:28-36): the same box
    tree.level[l.id[j]] = level;
    const int count = l.count[j];
    l.split[j] = 0;
    if (level >= o.max_depth || count <= o.leaf_max) {
        l.state[j] = NODE_LEAF;
        return;
    }
    // box3d::dim() = max(0, boxmax - boxmin); the strictly longest axis, x before y before z
    const float sx = fmaxf(0.0f, bbox[3] - bbox[0]), sy = fmaxf(0.0f, bbox[4] - bbox[1]), sz = fmaxf(0.0f, bbox[5] - bbox[2]);
    const int axis = (sx > sy && sx > sz) ? 0 : (sy > sz ? 1 : 2);
    l.axis[j] = axis;
    l.bins[j] = min(kMaxBins, count * 2);
    l.lo[j] = vbox[axis];
    l.hi[j] = vbox[3 + axis];
    l.state[j] = NODE_CANDIDATE;
    const int slot = atomicAdd(bins.used, 1);
    l.slot[j] = slot;
    for (int b = 0; b < kMaxBins; b++) {
        bins.n[(size_t)slot * kMaxBins + b] = 0;
        for (int k = 0; k < 6; k++)
            bins.key[((size_t)slot * kMaxBins + b) * 6 + k] = float_key(k < 3 ? FLT_MAX : -FLT_MAX);
    }
}

// bvh.cpp:153-170: which bin a triangle's barycentre falls into; the bin's count and box
__global__ void __launch_bounds__(kBlock) fill_bins(Triangles t, Level l, Bins bins)
{
    __shared__ int local_n[kMaxBins];
    __shared__ uint32_t local_key[kMaxBins * 6];
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    const int last = min((int)(blockIdx.x * blockDim.x + blockDim.x), t.count) - 1;
    // a workgroup whose positions all lie in one node gathers its bins in LDS first (nodes are contiguous ranges)
    const int j_first = t.node[blockIdx.x * blockDim.x], j_last = t.node[last];
    const bool gathered = j_first == j_last && j_first >= 0 && l.state[j_first] == NODE_CANDIDATE;
    if (gathered) {
        for (int b = threadIdx.x; b < kMaxBins; b += blockDim.x)
            local_n[b] = 0;
        for (int b = threadIdx.x; b < kMaxBins * 6; b += blockDim.x)
            local_key[b] = float_key(b % 6 < 3 ? FLT_MAX : -FLT_MAX);
        __syncthreads();
    }
    const int j = p < t.count ? t.node[p] : -1;
    if (j >= 0 && l.state[j] == NODE_CANDIDATE) {
        const int axis = l.axis[j], nbins = l.bins[j];
        const float lo = l.lo[j], hi = l.hi[j];
        const float scaled = (t.bary[(size_t)axis * t.count + p] - lo) * nbins / (hi - lo);
        const int b = min(nbins - 1, max(0, truncate_like_x86(floorf(scaled))));
        if (gathered) {
            atomicAdd(&local_n[b], 1);
            for (int k = 0; k < 3; k++) {
                atomicMin(&local_key[b * 6 + k], float_key(t.box[(size_t)k * t.count + p]));
                atomicMax(&local_key[b * 6 + 3 + k], float_key(t.box[(size_t)(3 + k) * t.count + p]));
            }
        } else {
            const size_t at = (size_t)l.slot[j] * kMaxBins + b;
            atomicAdd(&bins.n[at], 1);
            for (int k = 0; k < 3; k++) {
                atomicMin(&bins.key[at * 6 + k], float_key(t.box[(size_t)k * t.count + p]));
                atomicMax(&bins.key[at * 6 + 3 + k], float_key(t.box[(size_t)(3 + k) * t.count + p]));
            }
        }
    }
    if (gathered) {
        __syncthreads();
        const size_t base = (size_t)l.slot[j_first] * kMaxBins;
        for (int b = threadIdx.x; b < kMaxBins; b += blockDim.x)
            if (local_n[b]) {
                atomicAdd(&bins.n[base + b], local_n[b]);
                for (int k = 0; k < 3; k++) {
                    atomicMin(&bins.key[(base + b) * 6 + k], local_key[b * 6 + k]);
                    atomicMax(&bins.key[(base + b) * 6 + 3 + k], local_key[b * 6 + 3 + k]);
                }
            }
    }
}

struct Box {
    float lo[3], hi[3];
};
__device__ inline Box empty_box()
{
    Box b;
    for (int k = 0; k < 3; k++) {
        b.lo[k] = FLT_MAX;
        b.hi[k] = -FLT_MAX;
    }
    return b;
}
__device__ inline void box_add(Box &a, const Box &b)      // box3d::add(lo, hi): min(lo, boxmin), max(hi, boxmax)
{
    for (int k = 0; k < 3; k++) {
        a.lo[k] = fminf(b.lo[k], a.lo[k]);
        a.hi[k] = fmaxf(b.hi[k], a.hi[k]);
    }
}
// 2 * (d.x * d.y + d.x * d.z + d.y * d.z) of dim() = max(0, boxmax - boxmin)          (bvh.cpp:101-105)
__device__ inline float half_area_x2(const Box &b)
{
    const float x = fmaxf(0.0f, b.hi[0] - b.lo[0]), y = fmaxf(0.0f, b.hi[1] - b.lo[1]), z = fmaxf(0.0f, b.hi[2] - b.lo[2]);
    return 2 * (x * y + x * z + y * z);
}

// get_best_split's sweep (bvh.cpp:172-196) with sah() (bvh.cpp:107-120), one thread per candidate
__global__ void sweep_splits(Level l, int n, Options o, Bins bins)
{
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n || l.state[j] != NODE_CANDIDATE)
        return;
    const int nbins = l.bins[j], count = l.count[j];
    const size_t base = (size_t)l.slot[j] * kMaxBins;
    auto bin_box = [&](int b) {
        Box x;
        for (int k = 0; k < 3; k++) {
            x.lo[k] = key_float(bins.key[(base + b) * 6 + k]);
            x.hi[k] = key_float(bins.key[(base + b) * 6 + 3 + k]);
        }
        return x;
    };
    // suffix boxes / counts: everything in bins [i, nbins)
    float suffix_area[kMaxBins];
    int suffix_n[kMaxBins];
    {
        Box acc = empty_box();
        int m = 0;
        for (int i = nbins - 1; i >= 0; i--) {
            const Box b = bin_box(i);
            box_add(acc, b);
            m += bins.n[base + i];
            suffix_area[i] = half_area_x2(acc);
            suffix_n[i] = m;
        }
    }
    Box bounds;
    for (int k = 0; k < 3; k++) {
        bounds.lo[k] = key_float(l.vertex_key[6 * j + k]);
        bounds.hi[k] = key_float(l.vertex_key[6 * j + 3 + k]);
    }
    const float area = half_area_x2(bounds);
    const float unsplit = o.ctrav + o.cisec * count;           // the leaf's own cost (bvh.cpp:330)
    const float lo = l.lo[j], hi = l.hi[j];
    float best = unsplit, plane = 0.0f;
    Box prefix = empty_box();
    box_add(prefix, bin_box(0));
    for (int i = 1; i < nbins; i++) {
        const int rn = suffix_n[i], ln = count - rn;
        if (rn != 0 && ln != 0) {
            const float la = half_area_x2(prefix), ra = suffix_area[i];
            const float cost = o.ctrav + o.cisec * (la / area * ln + ra / area * rn);
            if (cost < best) {
                best = cost;
                plane = lo + i * (hi - lo) / nbins;
            }
        }
        box_add(prefix, bin_box(i));
    }
    if (best >= unsplit) {
        l.state[j] = NODE_LARGE_LEAF;      // no split beats the leaf (bvh.cpp:333-338)
        return;
    }
    l.plane[j] = plane;
    l.state[j] = NODE_SPLIT;
}

// partition's predicate (bvh.cpp:262: barycenter[axis] - plane < 0)
__global__ void flag_below(Triangles t, Level l)
{
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= t.count)
        return;
    const int j = t.node[p];
    int f = 0;
    if (j >= 0 && l.state[j] == NODE_SPLIT)
        f = (t.bary[(size_t)l.axis[j] * t.count + p] - l.plane[j] < 0) ? 1 : 0;
    t.flag[p] = f;
}

__global__ void count_below(Triangles t, Level l, int n)
{
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n || l.state[j] != NODE_SPLIT)
        return;
    const int start = l.start[j], end = start + l.count[j];
    const int mid = t.below[end - 1] - (start > 0 ? t.below[start - 1] : 0);
    if (mid <= 0 || mid >= l.count[j]) {
        l.state[j] = NODE_LARGE_LEAF;      // one side empty: the exchange moved nothing (bvh.cpp:344-349)
        return;
    }
    l.mid[j] = mid;
    l.split[j] = 1;
}

// The exchange partition as a pairing: among the first `mid` positions of the node those that belong right, in order from the
// left; among the others those that belong left, in order from the right; the k-th of one list swaps with the k-th of the other.
__global__ void pair_misplaced(Triangles t, Level l)
{
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= t.count)
        return;
    const int j = t.node[p];
    if (j < 0 || l.state[j] != NODE_SPLIT)
        return;
    const int start = l.start[j], end = start + l.count[j], boundary = start + l.mid[j];
    const int before_start = start > 0 ? t.below[start - 1] : 0;
    if (p < boundary && !t.flag[p]) {
        const int k = (p - start) - ((p > 0 ? t.below[p - 1] : 0) - before_start);     // not-below positions in [start, p)
        t.left_at[start + k] = p;
    } else if (p >= boundary && t.flag[p]) {
        const int k = t.below[end - 1] - t.below[p];                                      // below positions in (p, end)
        t.right_at[start + k] = p;
    }
}

__global__ void swap_pairs(Triangles t, Level l)
{
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= t.count)
        return;
    const int j = t.node[p];
    if (j < 0 || l.state[j] != NODE_SPLIT)
        return;
    const int start = l.start[j], boundary = start + l.mid[j];
    if (!(p < boundary && !t.flag[p]))
        return;
    const int before_start = start > 0 ? t.below[start - 1] : 0;
    const int k = (p - start) - ((p > 0 ? t.below[p - 1] : 0) - before_start);
    const int q = t.right_at[start + k];
    // this thread alone touches positions p and q
    for (int c = 0; c < 6; c++) {
        const float x = t.box[(size_t)c * t.count + p];
        t.box[(size_t)c * t.count + p] = t.box[(size_t)c * t.count + q];
        t.box[(size_t)c * t.count + q] = x;
    }
    for (int c = 0; c < 3; c++) {
        const float x = t.bary[(size_t)c * t.count + p];
        t.bary[(size_t)c * t.count + p] = t.bary[(size_t)c * t.count + q];
        t.bary[(size_t)c * t.count + q] = x;
    }
    const int o = t.original[p];
    t.original[p] = t.original[q];
    t.original[q] = o;
}

// the level's nodes into the tree; the next level's list                          (bvh.cpp:351-357, make_leaf :122-135)
__global__ void emit_nodes(Level l, int n, Level next, int next_first_id, Tree tree)
{
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n)
        return;
    const int id = l.id[j];
    if (l.state[j] != NODE_SPLIT) {
        tree.negative[id] = tree.positive[id] = -1;
        tree.start[id] = l.start[j];
        tree.triangles[id] = l.count[j];
        for (int k = 0; k < 3; k++)
            tree.direction[(size_t)3 * id + k] = 0.0f;
        return;
    }
    const int slot = 2 * l.child_offset[j], neg = next_first_id + slot, pos = neg + 1;
    tree.negative[id] = neg;
    tree.positive[id] = pos;
    tree.start[id] = 0;
    tree.triangles[id] = 0;
    for (int k = 0; k < 3; k++)
        tree.direction[(size_t)3 * id + k] = k == l.axis[j] ? 1.0f : 0.0f;
    tree.parent[neg] = tree.parent[pos] = id;
    next.start[slot] = l.start[j];
    next.count[slot] = l.mid[j];
    next.id[slot] = neg;
    next.start[slot + 1] = l.start[j] + l.mid[j];
    next.count[slot + 1] = l.count[j] - l.mid[j];
    next.id[slot + 1] = pos;
}

__global__ void descend(Triangles t, Level l)
{
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= t.count)
        return;
    const int j = t.node[p];
    if (j < 0)
        return;
    if (l.state[j] != NODE_SPLIT) {
        t.node[p] = -1;
        return;
    }
    t.node[p] = 2 * l.child_offset[j] + (p < l.start[j] + l.mid[j] ? 0 : 1);
}

// creation order -> pre-order: subtree sizes bottom-up, pre-order numbers top-down, one level per launch
__global__ void subtree_sizes(Tree tree, int first, int n, int *size)
{
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n)
        return;
    const int id = first + k;
    size[id] = tree.negative[id] < 0 ? 1 : 1 + size[tree.negative[id]] + size[tree.positive[id]];
}
__global__ void preorder_numbers(Tree tree, int first, int n, const int *size, int *number)
{
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n)
        return;
    const int id = first + k;
    if (tree.parent[id] < 0)
        number[id] = 0;
    if (tree.negative[id] >= 0) {
        number[tree.negative[id]] = number[id] + 1;
        number[tree.positive[id]] = number[id] + 1 + size[tree.negative[id]];
    }
}
__global__ void renumber(Tree from, int nodes, const int *number, Tree to)
{
    const int id = blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= nodes)
        return;
    const int me = number[id];
    to.parent[me] = from.parent[id] < 0 ? -1 : number[from.parent[id]];
    to.negative[me] = from.negative[id] < 0 ? -1 : number[from.negative[id]];
    to.positive[me] = from.positive[id] < 0 ? -1 : number[from.positive[id]];
    to.start[me] = from.start[id];
    to.triangles[me] = from.triangles[id];
    to.level[me] = from.level[id];
    for (int k = 0; k < 6; k++)
        to.box[(size_t)6 * me + k] = from.box[(size_t)6 * id + k];
    for (int k = 0; k < 3; k++)
        to.direction[(size_t)3 * me + k] = from.direction[(size_t)3 * id + k];
}
__global__ void reorder_vertices(int count, const int *__restrict__ original, const int *__restrict__ triangle_vertices, int *__restrict__ out)
{
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= count)
        return;
    for (int k = 0; k < 3; k++)
        out[3 * p + k] = triangle_vertices[3 * original[p] + k];
}

inline unsigned int blocks_for(int n) { return (unsigned int)((n + kBlock - 1) / kBlock); }

}   // namespace

struct shray_device_tree {
    int node_count = 0, triangle_count = 0, vertex_count = 0, vertex_stride = 0, leaf_count = 0, max_level = 0, large_leaves = 0;
    // The tree STAYS on the device (round 6: shray_flatten_device_tree and shray_scene_create_from_device read it there): the
    // pre-order arrays, the triangles' vertex indices in post-build order, the build's copy of the vertex data and the order.
    DeviceArray d_parent, d_negative, d_positive, d_start, d_triangles, d_box, d_direction, d_vertices, d_vertex_data, d_order;
    // host copies, made by the first shray_device_tree_download
    bool downloaded = false;
    std::vector<int32_t> parent, negative, positive, start, triangles, order, vertices;
    std::vector<float> box, direction;
    const float *vertex_data = nullptr;
    double seconds = 0;
};

#define BVH_TRY(expr)                                                                                     \
    do {                                                                                                  \
        const hipError_t e_ = (expr);                                                                     \
        if (e_ != hipSuccess) {                                                                           \
            char text_[256];                                                                              \
            snprintf(text_, sizeof text_, "shray_bvh_build_device: %s failed: %s", #expr, hipGetErrorString(e_)); \
            return shrayi_fail(SHRAY_ERR_DEVICE, text_);                                                  \
        }                                                                                                 \
    } while (0)

extern "C" {

int shray_bvh_build_device(const int32_t *triangle_vertices, int32_t triangle_count, const float *vertex_data, int32_t vertex_count,
                           int32_t vertex_stride_floats, const shray_bvh_options *options, shray_device_tree **out_tree)
{
    if (!out_tree)
        return shrayi_fail(SHRAY_ERR_INVALID_ARGUMENT, "shray_bvh_build_device: out_tree is NULL");
    *out_tree = nullptr;
    if (!triangle_vertices || !vertex_data || triangle_count <= 0 || vertex_count <= 0 || vertex_stride_floats < 3)
        return shrayi_fail(SHRAY_ERR_INVALID_ARGUMENT, "shray_bvh_build_device: no triangles, no vertices or a vertex stride below 3 floats");
    if (options && options->struct_size != sizeof(shray_bvh_options))
        return shrayi_fail(SHRAY_ERR_INVALID_ARGUMENT, "shray_bvh_build_device: options->struct_size is not sizeof(shray_bvh_options)");
    if (triangle_count > (1 << 28))
        return shrayi_fail(SHRAY_ERR_INVALID_ARGUMENT, "shray_bvh_build_device: more than 2^28 triangles");
    for (int64_t k = 0; k < (int64_t)3 * triangle_count; k++)
        if (triangle_vertices[k] < 0 || triangle_vertices[k] >= vertex_count)
            return shrayi_fail(SHRAY_ERR_INVALID_ARGUMENT, "shray_bvh_build_device: a triangle names a vertex that does not exist");
    // non-finite coordinates: the host's fmin / fmax ignore a NaN where the device's keyed atomics would take it (a NaN's key lies
    // above +inf's), so the two builds would part without a word -- refused
    for (int64_t v = 0; v < vertex_count; v++)
        for (int k = 0; k < 3; k++)
            if (!std::isfinite(vertex_data[v * vertex_stride_floats + k]))
                return shrayi_fail(SHRAY_ERR_INVALID_ARGUMENT, "shray_bvh_build_device: a vertex coordinate is not finite");
    Options o{30, 10, 1.0f, 4.0f};     // the reference's defaults (bvh.cpp:28-58)
    if (options) {
        o.max_depth = options->max_depth;
        // the reference compares a node's count with BVH_LEAF_MAX as UNSIGNED (bvh.cpp:303): a negative value makes every node a leaf
        o.leaf_max = options->leaf_max < 0 ? INT_MAX - 1 : std::min(options->leaf_max, INT_MAX - 1);
        o.ctrav = options->sah_ctrav;
        o.cisec = options->sah_cisec;
        if (o.max_depth > 4096 || !std::isfinite(o.ctrav) || !std::isfinite(o.cisec))
            return shrayi_fail(SHRAY_ERR_INVALID_ARGUMENT, "shray_bvh_build_device: options out of range (max_depth <= 4096, finite SAH costs)");
    }
    const int T = triangle_count, max_nodes = 2 * T;

    struct Events {     // (destroyed on every way out)
        hipEvent_t began = nullptr, ended = nullptr;
        ~Events()
        {
            if (began)
                (void)hipEventDestroy(began);
            if (ended)
                (void)hipEventDestroy(ended);
        }
    } events;
    BVH_TRY(hipEventCreate(&events.began));
    BVH_TRY(hipEventCreate(&events.ended));
    hipEvent_t &began = events.began, &ended = events.ended;
    DeviceArray d_tv, d_vd, d_box, d_bary, d_original, d_node, d_flag, d_below, d_left, d_right, d_scan_temp, d_pair;
    BVH_TRY(d_pair.alloc(8));
    BVH_TRY(d_tv.alloc((size_t)3 * T * 4));
    BVH_TRY(d_vd.alloc((size_t)vertex_count * vertex_stride_floats * 4));
    BVH_TRY(hipMemcpy(d_tv.p, triangle_vertices, (size_t)3 * T * 4, hipMemcpyHostToDevice));
    BVH_TRY(hipMemcpy(d_vd.p, vertex_data, (size_t)vertex_count * vertex_stride_floats * 4, hipMemcpyHostToDevice));
    BVH_TRY(hipEventRecord(began, nullptr));
    BVH_TRY(d_box.alloc((size_t)6 * T * 4));
    BVH_TRY(d_bary.alloc((size_t)3 * T * 4));
    for (DeviceArray *a : {&d_original, &d_node, &d_flag, &d_below, &d_left, &d_right})
        BVH_TRY(a->alloc((size_t)T * 4));
    Triangles t{T, d_box.as<float>(), d_bary.as<float>(), d_original.as<int>(), d_node.as<int>(), d_flag.as<int>(), d_below.as<int>(),
                d_left.as<int>(), d_right.as<int>()};

    // two level lists (this level's, the next one's); a level has at most T nodes
    DeviceArray lv[2][15];
    Level level[2];
    for (int s = 0; s < 2; s++) {
        const size_t words[15] = {1, 1, 1, 6, 6, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1};
        for (int k = 0; k < 15; k++)
            BVH_TRY(lv[s][k].alloc((size_t)T * words[k] * 4));
        level[s] = Level{lv[s][0].as<int>(), lv[s][1].as<int>(), lv[s][2].as<int>(), lv[s][3].as<uint32_t>(), lv[s][4].as<uint32_t>(),
                         lv[s][5].as<int>(), lv[s][6].as<int>(), lv[s][7].as<int>(), lv[s][8].as<int>(), lv[s][9].as<float>(),
                         lv[s][10].as<float>(), lv[s][11].as<float>(), lv[s][12].as<int>(), lv[s][13].as<int>(), lv[s][14].as<int>()};
    }
    // bins: only nodes of more than leaf_max triangles ask for a slot
    const int slots = T / std::max(1, o.leaf_max + 1) + 2;
    DeviceArray d_bin_n, d_bin_key, d_bin_used;
    BVH_TRY(d_bin_n.alloc((size_t)slots * kMaxBins * 4));
    BVH_TRY(d_bin_key.alloc((size_t)slots * kMaxBins * 6 * 4));
    BVH_TRY(d_bin_used.alloc(4));
    Bins bins{d_bin_n.as<int>(), d_bin_key.as<uint32_t>(), d_bin_used.as<int>()};

    DeviceArray tr[2][8];
    Tree tree[2];
    for (int s = 0; s < 2; s++) {
        const size_t words[8] = {1, 1, 1, 1, 1, 1, 6, 3};
        for (int k = 0; k < 8; k++)
            BVH_TRY(tr[s][k].alloc((size_t)max_nodes * words[k] * 4));
        tree[s] = Tree{tr[s][0].as<int>(), tr[s][1].as<int>(), tr[s][2].as<int>(), tr[s][3].as<int>(), tr[s][4].as<int>(), tr[s][5].as<int>(),
                       tr[s][6].as<float>(), tr[s][7].as<float>()};
    }
    size_t scan_bytes = 0;
    BVH_TRY(rocprim::inclusive_scan(nullptr, scan_bytes, t.flag, t.below, (size_t)T, rocprim::plus<int>(), nullptr));
    size_t scan_bytes_nodes = 0;
    BVH_TRY(rocprim::exclusive_scan(nullptr, scan_bytes_nodes, level[0].split, level[0].child_offset, 0, (size_t)T, rocprim::plus<int>(), nullptr));
    BVH_TRY(d_scan_temp.alloc(std::max(scan_bytes, scan_bytes_nodes)));
    size_t scan_capacity = std::max(scan_bytes, scan_bytes_nodes);

    hipLaunchKernelGGL(prepare_triangles, dim3(blocks_for(T)), dim3(kBlock), 0, nullptr, t, d_tv.as<int>(), d_vd.as<float>(), (int)vertex_stride_floats);
    {
        const int zero = 0, none = -1;
        BVH_TRY(hipMemcpy(level[0].start, &zero, 4, hipMemcpyHostToDevice));
        BVH_TRY(hipMemcpy(level[0].count, &T, 4, hipMemcpyHostToDevice));
        BVH_TRY(hipMemcpy(level[0].id, &zero, 4, hipMemcpyHostToDevice));
        BVH_TRY(hipMemcpy(tree[0].parent, &none, 4, hipMemcpyHostToDevice));
    }
    std::vector<int> level_first, level_nodes;      // creation-order ids of every level
    int n = 1, total = 1, depth = 0, which = 0;
    while (n > 0) {
        Level &l = level[which], &next = level[which ^ 1];
        level_first.push_back(total - n);
        level_nodes.push_back(n);
        BVH_TRY(hipMemsetAsync(bins.used, 0, 4, nullptr));
        hipLaunchKernelGGL(clear_level, dim3(blocks_for(n)), dim3(kBlock), 0, nullptr, l, n);
        hipLaunchKernelGGL(node_bounds, dim3(blocks_for(T)), dim3(kBlock), 0, nullptr, t, l);
        hipLaunchKernelGGL(decide_nodes, dim3(blocks_for(n)), dim3(kBlock), 0, nullptr, l, n, depth, o, tree[0], bins);
        hipLaunchKernelGGL(fill_bins, dim3(blocks_for(T)), dim3(kBlock), 0, nullptr, t, l, bins);
        hipLaunchKernelGGL(sweep_splits, dim3(blocks_for(n)), dim3(kBlock), 0, nullptr, l, n, o, bins);
        hipLaunchKernelGGL(flag_below, dim3(blocks_for(T)), dim3(kBlock), 0, nullptr, t, l);
        size_t bytes = scan_capacity;
        BVH_TRY(rocprim::inclusive_scan(d_scan_temp.p, bytes, t.flag, t.below, (size_t)T, rocprim::plus<int>(), nullptr));
        hipLaunchKernelGGL(count_below, dim3(blocks_for(n)), dim3(kBlock), 0, nullptr, t, l, n);
        bytes = scan_capacity;
        BVH_TRY(rocprim::exclusive_scan(d_scan_temp.p, bytes, l.split, l.child_offset, 0, (size_t)n, rocprim::plus<int>(), nullptr));
        hipLaunchKernelGGL(pair_misplaced, dim3(blocks_for(T)), dim3(kBlock), 0, nullptr, t, l);
        hipLaunchKernelGGL(swap_pairs, dim3(blocks_for(T)), dim3(kBlock), 0, nullptr, t, l);
        hipLaunchKernelGGL(emit_nodes, dim3(blocks_for(n)), dim3(kBlock), 0, nullptr, l, n, next, total, tree[0]);
        hipLaunchKernelGGL(descend, dim3(blocks_for(T)), dim3(kBlock), 0, nullptr, t, l);
        BVH_TRY(hipGetLastError());
        // how many nodes split: the last node's offset + its own flag (one small kernel gathers the two words, one copy fetches them)
        hipLaunchKernelGGL(gather_pair, dim3(1), dim3(1), 0, nullptr, l.child_offset + (n - 1), l.split + (n - 1), d_pair.as<int>());
        int last[2] = {0, 0};
        BVH_TRY(hipMemcpy(last, d_pair.p, 8, hipMemcpyDeviceToHost));
        const int children = 2 * (last[0] + last[1]);
        if (total + children > max_nodes)
            return shrayi_fail(SHRAY_ERR_DEVICE, "shray_bvh_build_device: more nodes than a binary tree over the triangles can have");
        total += children;
        n = children;
        which ^= 1;
        depth++;
        if (depth > 4096)
            return shrayi_fail(SHRAY_ERR_DEVICE, "shray_bvh_build_device: the build does not end");
    }
    // pre-order numbers
    DeviceArray d_size, d_number, d_vertices_out;
    BVH_TRY(d_size.alloc((size_t)total * 4));
    BVH_TRY(d_number.alloc((size_t)total * 4));
    for (int lv_k = (int)level_first.size() - 1; lv_k >= 0; lv_k--)
        hipLaunchKernelGGL(subtree_sizes, dim3(blocks_for(level_nodes[lv_k])), dim3(kBlock), 0, nullptr, tree[0], level_first[lv_k], level_nodes[lv_k],
                           d_size.as<int>());
    for (size_t lv_k = 0; lv_k < level_first.size(); lv_k++)
        hipLaunchKernelGGL(preorder_numbers, dim3(blocks_for(level_nodes[lv_k])), dim3(kBlock), 0, nullptr, tree[0], level_first[lv_k], level_nodes[lv_k],
                           d_size.as<int>(), d_number.as<int>());
    hipLaunchKernelGGL(renumber, dim3(blocks_for(total)), dim3(kBlock), 0, nullptr, tree[0], total, d_number.as<int>(), tree[1]);
    BVH_TRY(d_vertices_out.alloc((size_t)3 * T * 4));
    hipLaunchKernelGGL(reorder_vertices, dim3(blocks_for(T)), dim3(kBlock), 0, nullptr, T, t.original, d_tv.as<int>(), d_vertices_out.as<int>());
    BVH_TRY(hipGetLastError());
    BVH_TRY(hipEventRecord(ended, nullptr));
    BVH_TRY(hipEventSynchronize(ended));
    float ms = 0;
    BVH_TRY(hipEventElapsedTime(&ms, began, ended));

    std::unique_ptr<shray_device_tree> made(new shray_device_tree);
    made->node_count = total;
    made->triangle_count = T;
    made->vertex_count = vertex_count;
    made->vertex_stride = vertex_stride_floats;
    made->vertex_data = vertex_data;
    made->seconds = ms * 1e-3;
    // the statistics need three words per node (print_bvh_stats, bvh.cpp:83-99); everything else is downloaded when asked for
    made->negative.resize(total);
    made->triangles.resize(total);
    std::vector<int32_t> levels(total);
    BVH_TRY(hipMemcpy(made->negative.data(), tree[1].negative, (size_t)total * 4, hipMemcpyDeviceToHost));
    BVH_TRY(hipMemcpy(made->triangles.data(), tree[1].triangles, (size_t)total * 4, hipMemcpyDeviceToHost));
    BVH_TRY(hipMemcpy(levels.data(), tree[1].level, (size_t)total * 4, hipMemcpyDeviceToHost));
    for (int k = 0; k < total; k++) {
        made->max_level = std::max(made->max_level, (int)levels[k]);
        if (made->negative[k] < 0) {
            made->leaf_count++;
            // a leaf above leaf_max triangles that is not at the depth limit: no split beat it (print_bvh_stats, bvh.cpp:83-99)
            if (made->triangles[k] > o.leaf_max && levels[k] < o.max_depth)
                made->large_leaves++;
        }
    }
    // the renumbered tree, the reordered triangles, the vertex data and the order stay: their buffers change owner
    std::swap(made->d_parent.p, tr[1][0].p);
    std::swap(made->d_negative.p, tr[1][1].p);
    std::swap(made->d_positive.p, tr[1][2].p);
    std::swap(made->d_start.p, tr[1][3].p);
    std::swap(made->d_triangles.p, tr[1][4].p);
    std::swap(made->d_box.p, tr[1][6].p);
    std::swap(made->d_direction.p, tr[1][7].p);
    std::swap(made->d_vertices.p, d_vertices_out.p);
    std::swap(made->d_vertex_data.p, d_vd.p);
    std::swap(made->d_order.p, d_original.p);
    *out_tree = made.release();
    return SHRAY_OK;
}

int shray_device_tree_download(shray_device_tree *tree, shray_tree_desc *desc, const int32_t **triangle_order)
{
    if (!tree || !desc)
        return shrayi_fail(SHRAY_ERR_INVALID_ARGUMENT, "shray_device_tree_download: tree or desc is NULL");
    if (!tree->downloaded) {
        const size_t n = (size_t)tree->node_count, T = (size_t)tree->triangle_count;
        tree->parent.resize(n);
        tree->negative.resize(n);
        tree->positive.resize(n);
        tree->start.resize(n);
        tree->triangles.resize(n);
        tree->box.resize(6 * n);
        tree->direction.resize(3 * n);
        tree->order.resize(T);
        tree->vertices.resize(3 * T);
        BVH_TRY(hipMemcpy(tree->parent.data(), tree->d_parent.p, n * 4, hipMemcpyDeviceToHost));
        BVH_TRY(hipMemcpy(tree->negative.data(), tree->d_negative.p, n * 4, hipMemcpyDeviceToHost));
        BVH_TRY(hipMemcpy(tree->positive.data(), tree->d_positive.p, n * 4, hipMemcpyDeviceToHost));
        BVH_TRY(hipMemcpy(tree->start.data(), tree->d_start.p, n * 4, hipMemcpyDeviceToHost));
        BVH_TRY(hipMemcpy(tree->triangles.data(), tree->d_triangles.p, n * 4, hipMemcpyDeviceToHost));
        BVH_TRY(hipMemcpy(tree->box.data(), tree->d_box.p, n * 24, hipMemcpyDeviceToHost));
        BVH_TRY(hipMemcpy(tree->direction.data(), tree->d_direction.p, n * 12, hipMemcpyDeviceToHost));
        BVH_TRY(hipMemcpy(tree->order.data(), tree->d_order.p, T * 4, hipMemcpyDeviceToHost));
        BVH_TRY(hipMemcpy(tree->vertices.data(), tree->d_vertices.p, T * 12, hipMemcpyDeviceToHost));
        tree->downloaded = true;
    }
    memset(desc, 0, sizeof(*desc));
    desc->struct_size = sizeof(shray_tree_desc);
    desc->node_count = tree->node_count;
    desc->node_parent = tree->parent.data();
    desc->node_negative = tree->negative.data();
    desc->node_positive = tree->positive.data();
    desc->node_box = tree->box.data();
    desc->node_direction = tree->direction.data();
    desc->node_start = tree->start.data();
    desc->node_triangles = tree->triangles.data();
    desc->triangle_count = tree->triangle_count;
    desc->triangle_vertices = tree->vertices.data();
    desc->vertex_count = tree->vertex_count;
    desc->vertex_data = tree->vertex_data;
    if (triangle_order)
        *triangle_order = tree->order.data();
    return SHRAY_OK;
}

int shray_device_tree_stats(const shray_device_tree *tree, shray_bvh_stats *stats)
{
    if (!tree || !stats)
        return shrayi_fail(SHRAY_ERR_INVALID_ARGUMENT, "shray_device_tree_stats: tree or stats is NULL");
    stats->node_count = tree->node_count;
    stats->leaf_count = tree->leaf_count;
    stats->max_level = tree->max_level;
    stats->large_leaves = tree->large_leaves;
    stats->device_seconds = tree->seconds;
    return SHRAY_OK;
}

int shray_device_tree_destroy(shray_device_tree *tree)
{
    delete tree;
    return SHRAY_OK;
}

// (internal, device_tree_internal.h)
int shrayi_device_tree_view(const shray_device_tree *tree, ShrayDeviceTreeView *view)
{
    if (!tree || !view)
        return shrayi_fail(SHRAY_ERR_INVALID_ARGUMENT, "device tree is NULL");
    *view = ShrayDeviceTreeView{tree->node_count, tree->triangle_count, tree->vertex_count, tree->vertex_stride, tree->max_level,
                                tree->d_parent.as<int>(), tree->d_negative.as<int>(), tree->d_positive.as<int>(), tree->d_start.as<int>(),
                                tree->d_triangles.as<int>(), tree->d_box.as<float>(), tree->d_direction.as<float>(),
                                tree->d_vertices.as<int>(), tree->d_vertex_data.as<float>()};
    return SHRAY_OK;
}

}   // extern "C"
