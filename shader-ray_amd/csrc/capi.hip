// capi.hip -- implementation of include/shader_ray_hip.h: scene upload,
// validation and repacking, environment upload, render entry points.
// Host code only; the kernels live in kernel_*.hip.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <memory>
#include <string>
#include <vector>

#include "device_types.h"
#include "frame_params_defaults.h"
#include "launch.h"
#include "packed_layout.h"
#include "device_tree_internal.h"
#include "shader_ray_hip.h"

using namespace shray;

namespace {

thread_local std::string g_error;

int fail(int code, const char *fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_error = buf;
    return code;
}

#define HIP_TRY(expr)                                                                          \
    do {                                                                                       \
        hipError_t e_ = (expr);                                                                \
        if (e_ != hipSuccess)                                                                  \
            return fail(e_ == hipErrorOutOfMemory ? SHRAY_ERR_OUT_OF_MEMORY : SHRAY_ERR_DEVICE, \
                        "%s failed: %s", #expr, hipGetErrorString(e_));                        \
    } while (0)

const float kTerminatorF = 16777215.0f;   // raytracer.es.fs:384

// binary32 -> binary16 bits, round to nearest even (GL_RGB16F upload, ray.cpp:474)
__host__ __device__ uint16_t float_to_half_bits(float f)
{
    uint32_t u;
    memcpy(&u, &f, 4);
    const uint16_t sign = (uint16_t)((u >> 16) & 0x8000u);
    uint32_t mag = u & 0x7fffffffu;
    if (mag > 0x7f800000u)
        return sign | 0x7e00u;                      // NaN
    if (mag >= 0x477ff000u)
        return sign | 0x7c00u;                      // overflow -> inf (also inf itself)
    if (mag < 0x33000001u)
        return sign;                                // rounds to zero (<= 2^-25)
    if (mag < 0x38800000u) {                        // subnormal half
        const int shift = 126 - (int)(mag >> 23);   // 14..24
        const uint32_t mant = (mag & 0x7fffffu) | 0x800000u;
        const uint32_t q = mant >> shift;
        const uint32_t rem = mant & ((1u << shift) - 1u);
        const uint32_t halfway = 1u << (shift - 1);
        const uint32_t up = (rem > halfway || (rem == halfway && (q & 1u))) ? 1u : 0u;
        return sign | (uint16_t)(q + up);
    }
    const uint32_t lsb = (mag >> 13) & 1u;
    mag += 0xfffu + lsb;
    return sign | (uint16_t)((mag - 0x38000000u) >> 13);
}

struct DeviceBuffer {
    void *p = nullptr;
    ~DeviceBuffer() { release(); }
    void release()
    {
        if (p)
            (void)hipFree(p);
        p = nullptr;
    }
    hipError_t upload(const void *src, size_t bytes)
    {
        release();
        if (bytes == 0)
            bytes = 16;   // keep a valid pointer for empty scenes
        hipError_t e = hipMalloc(&p, bytes);
        if (e != hipSuccess)
            return e;
        if (src)
            return hipMemcpy(p, src, bytes, hipMemcpyHostToDevice);
        e = hipMemset(p, 0, bytes);
        // the fill runs on the null stream, which non-blocking streams do not wait for: finish it here, before
        // any kernel or copy on another stream can touch the buffer
        return e != hipSuccess ? e : hipStreamSynchronize(nullptr);
    }
    // `bytes` of device memory, uninitialised, or a copy of device memory (shray_scene_create_from_device; null stream)
    hipError_t reserve(size_t bytes)
    {
        release();
        return hipMalloc(&p, bytes ? bytes : 16);
    }
    hipError_t copy_of(const void *device_src, size_t bytes)
    {
        const hipError_t e = reserve(bytes);
        return (e != hipSuccess || !bytes) ? e : hipMemcpyAsync(p, device_src, bytes, hipMemcpyDeviceToDevice, nullptr);
    }
};

}   // namespace

struct shray_scene {
    int device = 0;
    int kernel_id = 0;          // 0 = stack kernel, 1 = literal threaded kernel, 2 = pool kernel (waves merge)
    bool packed_ok = false;     // link tables verified against the packed tree
    int stack_levels = 1;

    DeviceBuffer positions, normals16, normals32, boxmin, boxmax, hitmiss, objects;
    DeviceBuffer packed_nodes, packed_tris, pair_nodes;
    uint32_t max_leaf_count = 0;     // the largest leaf: the pair records keep min(count, 127) (packed_layout.h)
    DeviceBuffer env;
    DeviceBuffer counters;

    // shray_render_batch_device: per-frame FrameViews travel through a small ring of slots
    // (pinned staging -> device); a slot is reused only after the launch that read it has finished
    static constexpr int kBatchSlots = 16;
    FrameView *batch_staging = nullptr;   // pinned host, kBatchSlots * SHRAY_MAX_BATCH
    DeviceBuffer batch_views;             // device, same shape
    hipEvent_t batch_done[kBatchSlots] = {};
    bool batch_pending[kBatchSlots] = {};

    // Dispatch order of the convergent batch kernels: heaviest patches first, learnt from the frames before.
    // A frame's few long-running waves (rays grazing the silhouette, caught between the ears) last as long as a whole
    // frame of average waves; in row-major order many of them start when a launch is almost over, and a launch that is
    // not followed at once by another -- a rank's share of a step on 8 GPUs, the last launches of a short run -- waits
    // for them with the machine empty (profiles/history/r03/dispatch_order_ab.txt: a rank's 20-frame share at N = 8 takes 0.64
    // instead of 0.91 ms).  The waves of the launches a re-sort follows (of every one-frame launch) leave their running
    // time in `cost` (per patch, the longest); every few launches a one-workgroup kernel behind the launch, on its
    // stream, turns the costs into the next permutation (launch_dispatch_order); launches take a permutation up once it
    // is complete (`ready[entry]`, polled: no other stream ever waits for it, and no stream of the library's own competes
    // with the caller's for hardware queues).
    // A ring entry is never rewritten while a launch may still read it: every entry remembers the last launch that was
    // given it (`last_reader`, in the scene's launch numbers) and is written again only when that launch is known to be
    // over -- the launch kBatchSlots launches later has waited for it (launch_stack_views) -- and when it is neither the
    // entry new launches read nor one that is still being written; otherwise the re-sort is skipped (the next due launch
    // tries again).
    struct DispatchOrder {
        static constexpr int kRing = kBatchSlots + 2;
        DeviceBuffer cost, ring;            // 2 n words (costs, the order kernel's copy); kRing * n words
        size_t capacity = 0;                // patches the two buffers were sized for
        uint32_t n = 0;                     // 0: the slot holds no shape
        long long key[8] = {};              // the launch shape the costs belong to
        int current = -1;                   // ring entry new launches read; -1: none yet (identity)
        int written = -1;                   // ring entry the order kernel last wrote (or is writing)
        int pending_first = 0, pending_count = 0;   // entries pending_first .. written (ring order) are enqueued, not known complete
        unsigned long long launches = 0;    // since the shape was set
        hipStream_t written_on = nullptr;   // the stream the pending order kernels were enqueued on: launches enqueued on it
                                            // afterwards are ordered behind them and may read `written` at once
        hipEvent_t ready[kRing] = {};       // recorded behind the order kernel that wrote the entry
        unsigned long long last_reader[kRing] = {};   // scene launch number of the last launch given the entry (0: none)
        unsigned long long last_use = 0;    // (of the scene's dispatch clock: the least recently used shape makes room)
    };
    // a few shapes side by side (a caller that alternates between two frame sizes or tile sets keeps both orders; a new
    // shape beyond that evicts the least recently used one: its buffers are set aside until the launches that may read
    // them are over (`retired`), nothing synchronises)
    static constexpr int kDispatchShapes = 4;
    DispatchOrder dispatch[kDispatchShapes];
    int dispatch_last = -1;                 // the shape of the most recent ordered launch (shray_scene_dispatch_order)
    unsigned long long dispatch_clock = 0;
    unsigned long long launch_seq = 0;      // launches through launch_stack_views, 1-based; launch q uses batch slot (q - 1) % kBatchSlots
    struct RetiredOrder {
        DeviceBuffer cost, ring;
        size_t capacity = 0;
        unsigned long long retired_at = 0;  // the scene's launch number when the shape was evicted
    };
    std::vector<std::unique_ptr<RetiredOrder>> retired;
    // a caller that rotates more shapes than there are slots would evict on every launch: shapes that come back after
    // their eviction are counted (the last kEvictedKeys evicted keys are remembered), and after kThrashLimit of them without
    // a launch that found its shape in place the scene's launches keep row-major order for the next kThrashPause launches
    static constexpr int kEvictedKeys = 8, kThrashLimit = 8;
    static constexpr unsigned long long kThrashPause = 256;
    long long evicted_keys[kEvictedKeys][8] = {};
    int evicted_next = 0;
    int dispatch_returns = 0;               // evicted shapes that came back, since the last launch that found its shape in place
    unsigned long long dispatch_paused_until = 0;   // launch number; ordering is off below it

    // kernel id 4 (wavefront form): path queues, counts and per-sample radiances, grown on demand; one launch of a
    // scene at a time uses them (launches on different streams are ordered by wf_done)
    DeviceBuffer wf_queue0, wf_queue1, wf_counts, wf_radiance;
    size_t wf_paths = 0;
    hipEvent_t wf_done = nullptr;
    bool wf_pending = false;

    // shray_render / shray_render_host_async: the device frame is kept between calls (grown on demand),
    // and the blocking form's readback runs on the scene's own stream
    DeviceBuffer frame;
    size_t frame_bytes = 0;
    hipStream_t readback_stream = nullptr;

    SceneView view{};

    ~shray_scene()
    {
        if (readback_stream)
            (void)hipStreamDestroy(readback_stream);
        if (wf_done)
            (void)hipEventDestroy(wf_done);
        for (DispatchOrder &d : dispatch)
            for (hipEvent_t e : d.ready)
                if (e)
                    (void)hipEventDestroy(e);
        for (int k = 0; k < kBatchSlots; k++)
            if (batch_done[k])
                (void)hipEventDestroy(batch_done[k]);
        if (batch_staging)
            (void)hipHostFree(batch_staging);
    }
};

namespace {

// Validation + reconstruction of the tree from the eight (hit, miss) tables.
// On success fills `nodes` (packed, depth-first) and the depth.
struct TreeBuilder {
    const shray_scene_desc &d;
    uint32_t n, stride, tri_count;
    std::vector<int32_t> neg, pos;      // per reference node; -1 for leaves
    std::vector<uint8_t> axis;
    std::string why;

    explicit TreeBuilder(const shray_scene_desc &desc)
        : d(desc), n((uint32_t)desc.group_count), stride(desc.data_texture_width * (uint32_t)desc.group_data_rows),
          tri_count(desc.vertex_count / 3)
    {
    }

    const float *link(int code, uint32_t node) const { return d.group_hitmiss + 2 * ((size_t)stride * code + node); }

    // every link is END or a node index; every leaf names existing triangles
    bool links_are_safe()
    {
        for (int code = 0; code < 8; code++) {
            for (uint32_t g = 0; g < n; g++) {
                for (int k = 0; k < 2; k++) {
                    const float v = link(code, g)[k];
                    if (v >= kTerminatorF)
                        continue;
                    if (!(v >= 0.0f) || v != floorf(v) || v >= (float)n) {
                        why = "a hit/miss link is neither a node index nor a terminator";
                        return false;
                    }
                }
                const bool leaf = link(code, g)[0] == link(code, g)[1];
                if (leaf) {
                    const float s = d.group_objects[2 * (size_t)g], c = d.group_objects[2 * (size_t)g + 1];
                    if (!(s >= 0.0f) || !(c >= 0.0f) || s != floorf(s) || c != floorf(c) ||
                        (double)s + (double)c > (double)tri_count) {
                        why = "a leaf's (start, count) does not name existing triangles";
                        return false;
                    }
                }
            }
        }
        return true;
    }

    // children + split axis of every branch, from the hit links alone
    bool recover_children()
    {
        neg.assign(n, -1);
        pos.assign(n, -1);
        axis.assign(n, 0);
        for (uint32_t g = 0; g < n; g++) {
            const float h0 = link(0, g)[0];
            bool leaf = h0 == link(0, g)[1];
            for (int code = 1; code < 8; code++)
                if ((link(code, g)[0] == link(code, g)[1]) != leaf)
                    return false;
            if (leaf)
                continue;
            // code 0 (all components <= 0) goes to the positive child first, code 7 to the negative
            const float p = h0, m = link(7, g)[0];
            if (p >= kTerminatorF || m >= kTerminatorF || p == m)
                return false;
            int found = -1;
            for (int k = 0; k < 3 && found < 0; k++) {
                bool ok = true;
                for (int code = 0; code < 8 && ok; code++)
                    ok = link(code, g)[0] == (((code >> k) & 1) ? m : p);
                if (ok)
                    found = k;
            }
            if (found < 0)
                return false;
            axis[g] = (uint8_t)found;
            neg[g] = (int32_t)m;
            pos[g] = (int32_t)p;
        }
        return true;
    }

    // Re-threads the recovered tree for each direction code and compares with
    // the tables given; also checks it is a tree that reaches all n nodes.
    bool tables_match(int *depth_out)
    {
        std::vector<uint32_t> stack;
        int depth = 0;
        for (int code = 0; code < 8; code++) {
            uint32_t visited = 0;
            stack.clear();
            int64_t g = d.tree_root;
            while (g >= 0) {
                if (++visited > n)
                    return false;   // a cycle
                const int64_t next_subtree = stack.empty() ? -1 : (int64_t)stack.back();
                const float expect_miss = next_subtree < 0 ? -1.0f : (float)next_subtree;
                const float got_hit = link(code, (uint32_t)g)[0], got_miss = link(code, (uint32_t)g)[1];
                auto same = [](float got, float expect) { return expect < 0 ? got >= kTerminatorF : got == expect; };
                int64_t go;
                if (neg[g] < 0) {
                    if (!same(got_hit, expect_miss) || !same(got_miss, expect_miss))
                        return false;
                    go = next_subtree;
                    if (!stack.empty())
                        stack.pop_back();
                } else {
                    const bool neg_first = (code >> axis[g]) & 1;
                    const int32_t near_child = neg_first ? neg[g] : pos[g];
                    const int32_t far_child = neg_first ? pos[g] : neg[g];
                    if (got_hit != (float)near_child || !same(got_miss, expect_miss))
                        return false;
                    stack.push_back((uint32_t)far_child);
                    depth = std::max(depth, (int)stack.size());
                    if (stack.size() > 64)
                        return false;   // hitmiss_max_stack_size, world.cpp:228
                    go = near_child;
                }
                g = go;
            }
            if (visited != n)
                return false;
        }
        *depth_out = depth;
        return true;
    }

    // depth-first (negative subtree first) packing
    void pack(std::vector<PackedNode> &nodes, uint32_t *packed_root)
    {
        std::vector<uint32_t> new_index(n, 0);
        std::vector<uint32_t> order;
        order.reserve(n);
        std::vector<char> numbered(n, 0);
        std::vector<uint32_t> todo(1, (uint32_t)d.tree_root);
        while (!todo.empty()) {
            const uint32_t g = todo.back();
            todo.pop_back();
            if (!numbered[g]) {
                new_index[g] = (uint32_t)order.size();
                order.push_back(g);
            }
            if (neg[g] >= 0) {
                todo.push_back((uint32_t)pos[g]);
                todo.push_back((uint32_t)neg[g]);
            }
        }
        nodes.resize(n);
        for (uint32_t k = 0; k < n; k++) {
            const uint32_t g = order[k];
            PackedNode &pn = nodes[k];
            memcpy(pn.lo, d.group_boxmin + 3 * (size_t)g, 12);
            memcpy(pn.hi, d.group_boxmax + 3 * (size_t)g, 12);
            if (neg[g] >= 0) {
                pn.a = ((uint32_t)axis[g] << 30) | new_index[pos[g]];
                pn.b = new_index[neg[g]];
            } else {
                pn.a = (uint32_t)d.group_objects[2 * (size_t)g];
                pn.b = kLeafFlag | (uint32_t)d.group_objects[2 * (size_t)g + 1];
            }
        }
        *packed_root = new_index[(uint32_t)d.tree_root];
    }
};

// 1: multi-sample frames run a pixel's samples in neighbouring lanes (uniform_driver.h); 0: one lane per pixel
// at most 2^5 = 32 lanes per pixel: plaster 64 spp 16.38 / 15.83 / 16.26 ms with up to 64 / 32 / 16 (profiles/sample_lanes_probe.sh)
#ifndef SHRAY_SAMPLE_LANES_LOG2_MAX
#define SHRAY_SAMPLE_LANES_LOG2_MAX 5
#endif

int validate_params(const shray_frame_params *p, int width, int height, int spp)
{
    if (!p)
        return fail(SHRAY_ERR_INVALID_ARGUMENT, "frame params are NULL");
    if (p->struct_size != sizeof(shray_frame_params))
        return fail(SHRAY_ERR_INVALID_ARGUMENT, "shray_frame_params.struct_size is %u, this library expects %zu",
                    p->struct_size, sizeof(shray_frame_params));
    if (width <= 0 || height <= 0 || spp <= 0 || width > 65536 || height > 65536 || spp > (1 << 20))
        return fail(SHRAY_ERR_INVALID_ARGUMENT, "bad frame geometry %dx%d, %d spp", width, height, spp);
    if ((p->which == 3 || p->which == 5) && spp != 1)
        return fail(SHRAY_ERR_INVALID_ARGUMENT, "which = %d is a per-pixel view; spp must be 1", p->which);
    if (p->bounce_count < 0 || p->bounce_count > 64 || p->max_bvh_iterations < 1 || p->max_bvh_iterations > (1 << 24) ||
        p->max_leaf_tests < 0 || p->max_leaf_tests > (1 << 24))
        return fail(SHRAY_ERR_INVALID_ARGUMENT, "shader constants out of range (bounce_count %d, max_bvh_iterations %d, "
                    "max_leaf_tests %d)", p->bounce_count, p->max_bvh_iterations, p->max_leaf_tests);
    return SHRAY_OK;
}

// tiles t < total with phase <= t % stride < phase + count
int64_t owned_tile_count(int64_t total, int64_t stride, int64_t phase, int64_t count)
{
    return total / stride * count + std::min(count, std::max<int64_t>(0, total % stride - phase));
}

int make_frame_view(const shray_frame_params *p, int width, int height, int spp, const shray_tile_set *tiles,
                    FrameView *fr)
{
    memset(fr, 0, sizeof(*fr));
    memcpy(fr->camera_matrix, p->camera_matrix, 64);
    memcpy(fr->camera_normal_matrix, p->camera_normal_matrix, 64);
    memcpy(fr->object_matrix, p->object_matrix, 64);
    memcpy(fr->object_normal_matrix, p->object_normal_matrix, 64);
    memcpy(fr->object_normal_inverse, p->object_normal_inverse, 64);
    fr->image_plane_width = p->image_plane_width;
    fr->aspect = p->aspect;
    memcpy(fr->light_dir, p->light_dir, 12);
    memcpy(fr->specular_color, p->specular_color, 12);
    memcpy(fr->diffuse_color, p->diffuse_color, 12);
    fr->bounce_count = p->bounce_count;
    fr->max_bvh_iterations = p->max_bvh_iterations;
    fr->max_leaf_tests = p->max_leaf_tests;
    fr->cast_shadows = p->cast_shadows;
    fr->tonemap = p->tonemap;
    fr->normals_fp16 = p->normals_fp16;
    fr->which = p->which;
    memcpy(fr->right, p->right, 12);
    memcpy(fr->up, p->up, 12);
    fr->width = width;
    fr->height = height;
    fr->spp = spp;
    // lanes per pixel of a multi-sample frame (uniform_driver.h): the largest power of two <= min(spp, 32), as a
    // block of 2^x by 2^y neighbouring lanes of the wave's 8x8 lane grid
    {
        int log_g = 0;
        while (log_g < SHRAY_SAMPLE_LANES_LOG2_MAX && (2 << log_g) <= spp)
            log_g++;
        fr->sample_log_x = (uint32_t)((log_g + 1) / 2);
        fr->sample_log_y = (uint32_t)(log_g / 2);
    }

    const bool tiled = tiles && tiles->tile_stride > 0;
    if (!tiled) {
        fr->tile_stride = 0;
        fr->patches_x = (width + 15) / 16;
        fr->patches_per_unit = fr->patches_x * ((height + 15) / 16);
        fr->total_patches = (uint32_t)fr->patches_per_unit;
        return SHRAY_OK;
    }
    const int phase_count = tiles->tile_phase_count > 0 ? tiles->tile_phase_count : 1;
    if (tiles->tile_w <= 0 || tiles->tile_h <= 0 || tiles->tile_w % 16 || tiles->tile_h % 16 ||
        tiles->tile_phase < 0 || tiles->tile_phase_count < 0 || tiles->tile_phase + phase_count > tiles->tile_stride)
        return fail(SHRAY_ERR_INVALID_ARGUMENT, "tile set {%d x %d, stride %d, phase %d, count %d}: tile sizes must be positive "
                    "multiples of 16 and 0 <= phase, phase + count <= stride", tiles->tile_w, tiles->tile_h, tiles->tile_stride,
                    tiles->tile_phase, phase_count);
    fr->tile_w = tiles->tile_w;
    fr->tile_h = tiles->tile_h;
    fr->tile_stride = tiles->tile_stride;
    fr->tile_phase = tiles->tile_phase;
    fr->tile_phase_count = phase_count;
    fr->tiles_x = (width + tiles->tile_w - 1) / tiles->tile_w;
    const int tiles_y = (height + tiles->tile_h - 1) / tiles->tile_h;
    fr->owned_tiles = (int32_t)owned_tile_count((int64_t)fr->tiles_x * tiles_y, tiles->tile_stride, tiles->tile_phase, phase_count);
    fr->patches_x = tiles->tile_w / 16;
    fr->patches_per_unit = fr->patches_x * (tiles->tile_h / 16);
    fr->total_patches = (uint32_t)fr->owned_tiles * (uint32_t)fr->patches_per_unit;
    return SHRAY_OK;
}

// Which leaf stage the convergent instances of the stack kernel run.  Dealing a parked ray's triangles to the
// wave's idle lanes shortens divergent waves -- fewer instructions, one memory round trip per leaf instead of up
// to ten -- but the instance holds a second ray's worth of registers: six waves per SIMD instead of eight for the
// spp == 1 gold instance, five instead of six for the diffuse / shadow-ray ones.  Measured on MI355X
// (profiles/history/r02/leaf_stage_ab.txt):
//   * trees larger than an XCD's L2 share (the 1M-triangle scene, 9.4 MB of nodes): rays diverge, the walk is
//     latency-bound, dealing wins by 10-18 %;
//   * one spp == 1 frame per launch (latency): the frame ends with a tail of divergent waves, dealing wins by 11 %;
//   * a cache-resident scene rendered for throughput (several frames per launch, or many samples per pixel):
//     the GPU is full of coherent waves, the extra waves are worth more (3-6 %).
// SHRAY_DISPATCH_ORDER: heaviest patches first in the convergent batch kernels (shray_scene::DispatchOrder);
// the environment variable SHRAY_DISPATCH_ORDER=0 turns it off at run time (A/B, profiles/r03_dispatch_order_ab.sh)
// launches of a shape between two re-sorts (tuning: SHRAY_DISPATCH_PERIOD)
unsigned long long dispatch_period()
{
    static const unsigned long long period = [] {
        const char *e = getenv("SHRAY_DISPATCH_PERIOD");
        const long long v = e ? atoll(e) : 32;
        return (unsigned long long)(v < 1 ? 1 : v);
    }();
    return period;
}
bool dispatch_order_enabled()
{
    static const bool on = [] {
        const char *e = getenv("SHRAY_DISPATCH_ORDER");
        return !(e && e[0] == '0');
    }();
    return on;
}

// A re-sort follows a shape's first three launches, then launches 4, 8, 16 ... up to the period, and every dispatch_period()-th
// from then on.  Round 6 (profiles/r06/dispatch_period_ab.txt): the costs between two re-sorts are the maximum over the launches
// between them and halve at every re-sort, so eight launches between re-sorts made the order follow the last eight views -- on
// an orbit of twenty that is the half the next frames are NOT in -- and cost a 10-40 us kernel per eight frames; thirty-two cover
// the orbit (a lone frame of it 0.400 -> 0.387 ms), the doubling start keeps a new shape's first orders coming as fast as before.
bool dispatch_sorts_after(unsigned long long launch_number)
{
    if (launch_number <= 3 || launch_number % dispatch_period() == 0)
        return true;
    return launch_number < dispatch_period() && (launch_number & (launch_number - 1)) == 0;
}

// costs below (16 - bulk) / 16 of a shape's largest keep their row-major order among themselves (tuning: SHRAY_DISPATCH_BULK)
int dispatch_bulk_class()
{
    static const int bulk = [] {
        const char *e = getenv("SHRAY_DISPATCH_BULK");
        return e ? atoi(e) : 1 << 20;    // every class of its own (measured: profiles/history/r03/dispatch_bulk_ab.txt)
    }();
    return bulk;
}

bool leaf_stage_policy(const shray_scene *scene, int frames_in_launch, int spp)
{
    const bool divergent_scene = (size_t)scene->view.group_count * sizeof(PackedNode) > (2u << 20);
    const bool latency_launch = frames_in_launch == 1 && spp == 1;
    return divergent_scene || latency_launch;
}

// The stack kernel reads its FrameViews from device memory (far fewer scalar registers held, and
// spilled, than with the 480-byte view as a by-value kernel argument): `count` views travel through a
// ring of slots (pinned staging -> device, on `stream`), then one launch renders them all.
// tally / policy_frames: shray_render_counters_timed -- the instance a launch of `policy_frames` frames would run, with
// per-ray work tallies
// Which launches of the stack kernel test both children of a node per turn (variants/pair_traversal.h: inner_stage_pair): those
// of a scene that asked for it, shray_scene_set_kernel(scene, 3).  The pair form issues the same arithmetic and the same
// loads as the one-visit form but about 0.6 of its dependent round trips, for a fatter turn; measured 24-35 % slower on
// every configuration (profiles/EXPERIMENTS.md R3.2), so no launch chooses it by itself.
bool pair_policy(const shray_scene *scene, const FrameView *views, int count)
{
    if (!scene->pair_nodes.p || scene->kernel_id != 3)
        return false;
    // a pair record keeps min(triangle count, 127): exact whenever the leaf cap or the largest leaf stays below that
    for (int k = 0; k < count; k++)
        if (scene->max_leaf_count > 126u && (uint32_t)views[k].max_leaf_tests > 126u)
            return false;
    return true;
}

// A dispatch-order slot starts over for a new launch shape.  Launches of the shape it held may still be running and
// reading its buffers: those are set aside (shray_scene::retired) and handed out again only when every launch that was
// enqueued before the eviction is over -- kBatchSlots launches later (launch_stack_views' slot wait) --, so nothing here
// synchronises with the device.  `seq`: the current launch's number; `stream`: its stream (the new costs are zeroed on it:
// a launch of the shape on ANOTHER stream that reports before the fill has run loses its report, which the next one
// repeats).  A caller that rotates more shapes than there are slots would evict on every launch: shapes that come back
// after their eviction are counted, and after kThrashLimit of them the scene's launches keep row-major order for a while
// (shray_scene::dispatch_paused_until; the slot is then left as it is).
int dispatch_slot_start(shray_scene *scene, shray_scene::DispatchOrder &d, const long long key[8], uint32_t patches,
                        unsigned long long seq, hipStream_t stream)
{
    using Order = shray_scene::DispatchOrder;
    // a shape that was evicted not long ago and is back: the caller rotates more shapes than there are slots
    for (int k = 0; k < shray_scene::kEvictedKeys; k++)
        if (memcmp(scene->evicted_keys[k], key, sizeof(scene->evicted_keys[k])) == 0 && key[7] != 0) {
            if (++scene->dispatch_returns >= shray_scene::kThrashLimit) {
                scene->dispatch_returns = 0;
                scene->dispatch_paused_until = seq + shray_scene::kThrashPause;
                scene->dispatch_last = -1;
                return SHRAY_OK;
            }
            break;
        }
    if (d.n) {
        memcpy(scene->evicted_keys[scene->evicted_next], d.key, sizeof(d.key));
        scene->evicted_next = (scene->evicted_next + 1) % shray_scene::kEvictedKeys;
        d.n = 0;                                   // from here on the slot matches no key, whatever fails below
        std::unique_ptr<shray_scene::RetiredOrder> r(new shray_scene::RetiredOrder);
        std::swap(r->cost.p, d.cost.p);
        std::swap(r->ring.p, d.ring.p);
        r->capacity = d.capacity;
        r->retired_at = seq;
        d.capacity = 0;
        scene->retired.push_back(std::move(r));
    }
    // (a start that failed behind its allocations -- the memset, an event -- left the slot unused, d.n == 0, WITH buffers of the
    // capacity that start asked for: no launch has read them; they serve this start only if they are large enough.  ADVICE round 4)
    if ((d.cost.p || d.ring.p) && d.capacity < patches) {
        d.cost.release();
        d.ring.release();
        d.capacity = 0;
    }
    // buffers: a retired pair that is large enough and no longer read, else new ones
    for (size_t k = 0; k < scene->retired.size() && !d.cost.p; k++) {
        shray_scene::RetiredOrder &r = *scene->retired[k];
        if (r.capacity >= patches && seq >= r.retired_at + shray_scene::kBatchSlots) {
            std::swap(d.cost.p, r.cost.p);
            std::swap(d.ring.p, r.ring.p);
            d.capacity = r.capacity;
            scene->retired.erase(scene->retired.begin() + (long)k);
        }
    }
    if (!d.cost.p || !d.ring.p) {
        d.cost.release();      // (an earlier allocation that failed half-way leaves nothing behind)
        d.ring.release();
        d.capacity = 0;
        // more than a handful set aside: free the ones nobody reads any more (hipFree waits for the device: rare, and only here)
        for (size_t k = 0; scene->retired.size() > 8 && k < scene->retired.size();) {
            if (seq >= scene->retired[k]->retired_at + shray_scene::kBatchSlots)
                scene->retired.erase(scene->retired.begin() + (long)k);
            else
                k++;
        }
        HIP_TRY(hipMalloc(&d.cost.p, (size_t)patches * 4 * 2));              // costs + the order kernel's copy of them
        HIP_TRY(hipMalloc(&d.ring.p, (size_t)patches * 4 * Order::kRing));
        d.capacity = patches;
    }
    HIP_TRY(hipMemsetAsync(d.cost.p, 0, (size_t)patches * 4 * 2, stream));
    for (hipEvent_t &e : d.ready)
        if (!e)
            HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    memcpy(d.key, key, sizeof(d.key));
    d.current = d.written = -1;
    d.pending_first = d.pending_count = 0;
    d.launches = 0;
    d.written_on = nullptr;
    memset(d.last_reader, 0, sizeof(d.last_reader));
    d.n = patches;
    return SHRAY_OK;
}

int launch_stack_views(shray_scene *scene, const FrameView *views, int count, float4 *d_out, size_t frame_stride,
                       hipStream_t stream, DeviceCounters *tally = nullptr, int policy_frames = 0, bool tally_full_walk = false)
{
    if (views[0].total_patches == 0)
        return SHRAY_OK;
    if (!scene->batch_staging) {
        const size_t bytes = sizeof(FrameView) * shray_scene::kBatchSlots * SHRAY_MAX_BATCH;
        HIP_TRY(hipHostMalloc((void **)&scene->batch_staging, bytes, hipHostMallocDefault));
        HIP_TRY(scene->batch_views.upload(nullptr, bytes));
        for (int k = 0; k < shray_scene::kBatchSlots; k++)
            HIP_TRY(hipEventCreateWithFlags(&scene->batch_done[k], hipEventDisableTiming));
    }
    // launch number `seq` uses slot (seq - 1) % kBatchSlots and first waits for the launch that used it before: once this
    // wait has returned, every launch up to seq - kBatchSlots is over (each was waited for by its slot's next user)
    const unsigned long long seq = ++scene->launch_seq;
    const int slot = (int)((seq - 1) % shray_scene::kBatchSlots);
    if (scene->batch_pending[slot])
        HIP_TRY(hipEventSynchronize(scene->batch_done[slot]));   // rarely waits: the launch kBatchSlots launches ago
    FrameView *staged = scene->batch_staging + (size_t)slot * SHRAY_MAX_BATCH;
    FrameView *d_views = (FrameView *)scene->batch_views.p + (size_t)slot * SHRAY_MAX_BATCH;
    memcpy(staged, views, sizeof(FrameView) * (size_t)count);
    // heaviest patches first (shray_scene::DispatchOrder): plain frames of the stack kernel's convergent instances
    bool ordered = false;
    {
        bool plain = scene->kernel_id == 0 || scene->kernel_id == 3;
        for (int k = 0; k < count; k++)
            plain = plain && !(views[k].which == 1 || views[k].which == 2 || views[k].which == 3 || views[k].which == 5);
        // Where it pays (profiles/history/r03/dispatch_order_ab.txt): a launch that is not followed at once by more of the same -- one
        // frame per launch (a lone frame: 0.54 -> 0.45 ms), a tile set (a rank's share of a multi-GPU step: 0.92 -> 0.74 ms for
        // 20 frames at N = 8; within 3 % either way once steps follow each other without a gap).  Several whole frames per
        // launch over several streams (the N = 1 throughput form) run 6 % SLOWER heaviest-first -- the long divergent waves
        // of every launch then crowd the machine together -- and keep their row-major order.
        const bool lone_kind = count == 1 || views[0].tile_stride != 0;
        bool zero_diffuse = true;     // (the instances that read an order exist for zero-diffuse frames: kernel_stack.hip)
        for (int k = 0; k < count; k++)
            zero_diffuse = zero_diffuse && !(views[k].diffuse_color[0] > 0.0f && views[k].diffuse_color[1] > 0.0f && views[k].diffuse_color[2] > 0.0f);
        const bool pairs = plain && pair_policy(scene, views, count);
        // ... and for the instances that deal their leaves: every 1 spp launch, and multi-sample frames of a tree larger than
        // an L2 share (config 4: 3.00 -> 2.74 ms); a cache-resident multi-sample frame (config 5) is 5 % SLOWER re-ordered
        const bool dealing = views[0].spp == 1 || leaf_stage_policy(scene, policy_frames > 0 ? policy_frames : count, views[0].spp);
        if (plain && !tally && lone_kind && zero_diffuse && dealing && !pairs && dispatch_order_enabled() && seq >= scene->dispatch_paused_until) {
            const FrameView &f = views[0];
            const long long key[8] = {f.width, f.height, f.spp, f.tile_w, f.tile_h, f.tile_stride,
                                      ((long long)f.tile_phase << 32) | (unsigned int)f.tile_phase_count, (long long)f.total_patches};
            int which = -1, lru = 0;
            for (int k = 0; k < shray_scene::kDispatchShapes; k++) {
                if (scene->dispatch[k].n && memcmp(key, scene->dispatch[k].key, sizeof(key)) == 0)
                    which = k;
                if (scene->dispatch[k].last_use < scene->dispatch[lru].last_use)
                    lru = k;
            }
            if (which >= 0) {
                scene->dispatch_returns = 0;
            } else {
                // a shape not seen (lately) takes the least recently used slot
                const int rc = dispatch_slot_start(scene, scene->dispatch[lru], key, f.total_patches, seq, stream);
                if (rc)
                    return rc;
                which = seq < scene->dispatch_paused_until ? -1 : lru;   // (-1: too many shapes in rotation, row-major for a while)
            }
            if (which >= 0) {
                scene->dispatch_last = which;
                shray_scene::DispatchOrder &d = scene->dispatch[which];
                d.last_use = ++scene->dispatch_clock;
                // permutations that have become complete, oldest first (the pending order kernels sit on one stream, in order)
                while (d.pending_count > 0 && hipEventQuery(d.ready[d.pending_first]) == hipSuccess) {
                    d.current = d.pending_first;
                    d.pending_first = (d.pending_first + 1) % shray_scene::DispatchOrder::kRing;
                    d.pending_count--;
                }
                const int use = (d.pending_count > 0 && stream == d.written_on) ? d.written : d.current;
                if (use >= 0)
                    d.last_reader[use] = seq;
                // only the launches a re-sort follows report their waves' running times: reporting costs a launch 3 % (a late
                // scalar load of the buffer's address and an atomic per wave, profiles/history/r03/dispatch_mechanism_ab.txt)
                // (the two launches before a re-sort: a loop that alternates long and short steps reports both)
                // A launch of one frame always reports: its views change from launch to launch and the union of the last few is
                // the better predictor (0.474 against 0.480 ms on the orbit).
                const bool reports = count == 1 || dispatch_sorts_after(d.launches + 1) || dispatch_sorts_after(d.launches + 2);
                for (int k = 0; k < count; k++) {
                    staged[k].dispatch_cost = reports ? (uint32_t *)d.cost.p : nullptr;
                    staged[k].dispatch_order = use >= 0 ? (const uint32_t *)d.ring.p + (size_t)use * d.n : nullptr;
                }
                ordered = true;
            }
        }
    }
    HIP_TRY(hipMemcpyAsync(d_views, staged, sizeof(FrameView) * (size_t)count, hipMemcpyHostToDevice, stream));
    bool all_metal = true;
    for (int k = 0; k < count; k++)
        all_metal = all_metal && !(views[k].diffuse_color[0] > 0.0f && views[k].diffuse_color[1] > 0.0f &&
                                   views[k].diffuse_color[2] > 0.0f);
    bool plain_view = true;   // the pool kernel renders which == 0 frames; the shader's debug views stay on the stack kernel
    for (int k = 0; k < count; k++)
        plain_view = plain_view && !(views[k].which == 1 || views[k].which == 2 || views[k].which == 3 || views[k].which == 5);
    if (scene->kernel_id == 4 && plain_view && !tally && count == 1 && views[0].tile_stride == 0) {
        // the wavefront form: one launch per bounce over queues of live paths
        const FrameView &fr = views[0];
        const size_t paths = (size_t)fr.width * fr.height * fr.spp;
        if (scene->wf_paths < paths) {
            HIP_TRY(hipDeviceSynchronize());     // nothing may still be using the old queues
            scene->wf_queue0.release();
            scene->wf_queue1.release();
            scene->wf_radiance.release();
            scene->wf_paths = 0;
            HIP_TRY(hipMalloc(&scene->wf_queue0.p, paths * 64));
            HIP_TRY(hipMalloc(&scene->wf_queue1.p, paths * 64));
            HIP_TRY(hipMalloc(&scene->wf_radiance.p, paths * 16));
            scene->wf_paths = paths;
        }
        if (!scene->wf_counts.p)
            HIP_TRY(scene->wf_counts.upload(nullptr, sizeof(unsigned int) * 80));
        if (!scene->wf_done)
            HIP_TRY(hipEventCreateWithFlags(&scene->wf_done, hipEventDisableTiming));
        if (fr.bounce_count + 2 > 80)
            return fail(SHRAY_ERR_INVALID_ARGUMENT, "bounce_count %d is too large for the wavefront kernel", fr.bounce_count);
        if (scene->wf_pending)
            HIP_TRY(hipStreamWaitEvent(stream, scene->wf_done, 0));
        const hipError_t we = launch_wavefront(scene->view, d_views, fr, all_metal, (PathState *)scene->wf_queue0.p, (PathState *)scene->wf_queue1.p,
                                               (unsigned int *)scene->wf_counts.p, (float4 *)scene->wf_radiance.p, d_out, stream, scene->stack_levels);
        if (we != hipSuccess)
            return fail(SHRAY_ERR_DEVICE, "kernel launch failed: %s", hipGetErrorString(we));
        HIP_TRY(hipEventRecord(scene->wf_done, stream));
        scene->wf_pending = true;
        HIP_TRY(hipEventRecord(scene->batch_done[slot], stream));
        scene->batch_pending[slot] = true;
        return SHRAY_OK;
    }
    const hipError_t e = (scene->kernel_id == 2 && plain_view && !tally)
        ? launch_pool_batch(scene->view, d_views, count, views[0], all_metal, d_out, frame_stride, stream, scene->stack_levels)
        : launch_stack_batch(scene->view, d_views, count, views[0], all_metal, plain_view,
                             leaf_stage_policy(scene, policy_frames > 0 ? policy_frames : count, views[0].spp), d_out, frame_stride, stream,
                             scene->stack_levels, tally, plain_view && pair_policy(scene, views, count),
                             tally_full_walk, ordered);
    if (e != hipSuccess)
        return fail(SHRAY_ERR_DEVICE, "kernel launch failed: %s", hipGetErrorString(e));
    if (ordered) {
        // the next permutation: after each of the first launches of a shape, then every dispatch_period() launches
        using Order = shray_scene::DispatchOrder;
        Order &d = scene->dispatch[scene->dispatch_last];
        d.launches++;
        // (an update that is still pending on ANOTHER stream is not overtaken; on the same stream the kernels queue up)
        if ((d.pending_count == 0 || stream == d.written_on) && dispatch_sorts_after(d.launches)) {
            const int next = (d.written + 1) % Order::kRing;
            // `next` must be free: not the entry launches read, not one still being written, and not read by a launch that
            // may still be running (its last reader is over once kBatchSlots more launches have been enqueued, see above);
            // if it is not, this re-sort is skipped and the next due launch tries again
            const bool ring_full = (d.current >= 0 ? 1 : 0) + d.pending_count >= Order::kRing;
            const bool still_read = d.last_reader[next] != 0 && seq < d.last_reader[next] + shray_scene::kBatchSlots;
            if (!ring_full && !still_read) {
                const hipError_t oe = launch_dispatch_order((uint32_t *)d.cost.p, (uint32_t *)d.ring.p + (size_t)next * d.n, d.n, stream, dispatch_bulk_class());
                if (oe != hipSuccess)
                    return fail(SHRAY_ERR_DEVICE, "dispatch-order kernel launch failed: %s", hipGetErrorString(oe));
                HIP_TRY(hipEventRecord(d.ready[next], stream));
                if (d.pending_count == 0)
                    d.pending_first = next;
                d.pending_count++;
                d.written = next;
                d.written_on = stream;
                d.last_reader[next] = 0;
            }
        }
    }
    // (behind the order kernel, which reads and writes the shape's buffers: "launch seq is over" covers it)
    HIP_TRY(hipEventRecord(scene->batch_done[slot], stream));
    scene->batch_pending[slot] = true;
    return SHRAY_OK;
}

int launch(shray_scene *s, const FrameView &fr_in, float4 *d_out, DeviceCounters *d_counters, hipStream_t stream)
{
    FrameView fr = fr_in;
    if (fr.total_patches == 0)
        return SHRAY_OK;
    hipError_t e;
    if (s->kernel_id != 1 && s->packed_ok && !d_counters)
        return launch_stack_views(s, &fr, 1, d_out, 0, stream);
    const bool view_instance = fr.which == 1 || fr.which == 2 || fr.which == 3 || fr.which == 5;
    if (s->kernel_id == 3 && s->packed_ok && d_counters && !view_instance && pair_policy(s, &fr, 1))
        return launch_stack_views(s, &fr, 1, d_out, 0, stream, d_counters, 0, true);   // the pair traversal's own counting twin
    else if (s->kernel_id == 2 && s->packed_ok &&
             !(fr.which == 1 || fr.which == 2 || fr.which == 3 || fr.which == 5))
        e = launch_pool(s->view, fr, d_out, d_counters, stream, s->stack_levels);
    else if (s->kernel_id != 1 && s->packed_ok)
        e = launch_stack(s->view, fr, d_out, d_counters, stream, s->stack_levels);
    else
        e = launch_threaded(s->view, fr, d_out, d_counters, stream);
    if (e != hipSuccess)
        return fail(SHRAY_ERR_DEVICE, "kernel launch failed: %s", hipGetErrorString(e));
    return SHRAY_OK;
}

}   // namespace

// error reporting for the library's other translation units (flatten.hip)
extern "C" int shrayi_fail(int code, const char *message) { return fail(code, "%s", message); }

extern "C" {

int shray_abi_version(void) { return SHRAY_ABI_VERSION; }

const char *shray_last_error(void) { return g_error.c_str(); }

int shray_device_count(int *count)
{
    if (!count)
        return fail(SHRAY_ERR_INVALID_ARGUMENT, "count is NULL");
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) {
        *count = 0;
        return fail(SHRAY_ERR_NO_DEVICE, "no HIP device is visible");
    }
    *count = n;
    return SHRAY_OK;
}

int shray_set_device(int device_index)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0)
        return fail(SHRAY_ERR_NO_DEVICE, "no HIP device is visible");
    if (device_index < 0 || device_index >= n)
        return fail(SHRAY_ERR_INVALID_ARGUMENT, "device %d out of range [0, %d)", device_index, n);
    HIP_TRY(hipSetDevice(device_index));
    return SHRAY_OK;
}

void shray_frame_params_init(shray_frame_params *params)
{
    if (params)
        shray_frame_params_defaults(params);
}

int shray_scene_create(const shray_scene_desc *desc, shray_scene **out_scene)
{
    if (!desc || !out_scene)
        return fail(SHRAY_ERR_INVALID_ARGUMENT, "desc or out_scene is NULL");
    *out_scene = nullptr;
    if (desc->struct_size != sizeof(shray_scene_desc))
        return fail(SHRAY_ERR_INVALID_ARGUMENT, "shray_scene_desc.struct_size is %u, this library expects %zu",
                    desc->struct_size, sizeof(shray_scene_desc));
    if (desc->data_texture_width == 0 || desc->group_count < 1 || desc->group_data_rows < 1 ||
        desc->tree_root < 0 || desc->tree_root >= desc->group_count || desc->vertex_count % 3 != 0)
        return fail(SHRAY_ERR_INVALID_ARGUMENT, "inconsistent counts (width %u, %d nodes in %d rows, root %d, %u vertices)",
                    desc->data_texture_width, desc->group_count, desc->group_data_rows, desc->tree_root,
                    desc->vertex_count);
    const uint64_t stride = (uint64_t)desc->data_texture_width * (uint64_t)desc->group_data_rows;
    const uint64_t vertex_texels = (uint64_t)desc->data_texture_width * (uint64_t)desc->vertex_data_rows;
    if ((uint64_t)desc->group_count > stride || desc->vertex_count > vertex_texels)
        return fail(SHRAY_ERR_INVALID_ARGUMENT, "counts exceed width * rows");
    if (!desc->group_boxmin || !desc->group_boxmax || !desc->group_hitmiss || !desc->group_objects ||
        (desc->vertex_count && (!desc->vertex_positions || !desc->vertex_normals)))
        return fail(SHRAY_ERR_INVALID_ARGUMENT, "a required array is NULL");
    // The shader carries every index as float32 (raytracer.es.fs:239-245, :384):
    // exact only below 2^24, and node links at or above 16777215 mean "stop".
    if (stride * 8 > 16777216ull || desc->vertex_count > 16777216u)
        return fail(SHRAY_ERR_INDEX_RANGE, "scene too large for float32 indices (%llu link texels, %u vertices; "
                    "limit 2^24)", (unsigned long long)(stride * 8), desc->vertex_count);

    int device = 0;
    if (hipGetDevice(&device) != hipSuccess)
        return fail(SHRAY_ERR_NO_DEVICE, "no HIP device is available (hipGetDevice failed)");

    TreeBuilder tb(*desc);
    if (!tb.links_are_safe())
        return fail(SHRAY_ERR_BAD_TREE, "%s", tb.why.c_str());

    std::unique_ptr<shray_scene> s(new shray_scene);
    s->device = device;
    const size_t nv = desc->vertex_count, ng = (size_t)desc->group_count;

    HIP_TRY(s->positions.upload(desc->vertex_positions, nv * 12));
    HIP_TRY(s->normals32.upload(desc->vertex_normals, nv * 12));
    {
        std::vector<uint16_t> halves(nv * 3);
        for (size_t k = 0; k < halves.size(); k++)
            halves[k] = float_to_half_bits(desc->vertex_normals[k]);
        HIP_TRY(s->normals16.upload(halves.data(), halves.size() * 2));
    }
    HIP_TRY(s->boxmin.upload(desc->group_boxmin, ng * 12));
    HIP_TRY(s->boxmax.upload(desc->group_boxmax, ng * 12));
    HIP_TRY(s->objects.upload(desc->group_objects, ng * 8));
    HIP_TRY(s->hitmiss.upload(desc->group_hitmiss, (size_t)stride * 8 * 8));
    HIP_TRY(s->counters.upload(nullptr, sizeof(DeviceCounters) * kCounterShards));

    // packed layout for the stack kernel, if the tables describe a canonical threaded tree
    int depth = 0;
    if (tb.recover_children() && tb.tables_match(&depth)) {
        std::vector<PackedNode> nodes;
        uint32_t packed_root = 0;
        tb.pack(nodes, &packed_root);
        const size_t nt = nv / 3;
        std::vector<PackedTri> tris(nt + 1);   // a spare record: the leaf cache fetches a leaf in 16-byte chunks (leaf_cache.h)
        for (size_t t = 0; t < nt; t++) {
            const float *v = desc->vertex_positions + 9 * t;
            PackedTri &pt = tris[t];
            memset(&pt, 0, sizeof(pt));
            for (int a = 0; a < 3; a++) {
                pt.v0[a] = v[a];
                pt.e0[a] = v[3 + a] - v[a];        // e0 = v1 - v0, raytracer.es.fs:304
                pt.e1[a] = v[a] - v[6 + a];        // e1 = v0 - v2, raytracer.es.fs:305
            }
        }
        {
            // the device's form (packed_layout.h): one copy per direction octant, holding a box as the planes a ray of that
            // octant enters and leaves it by and a branch's children in the order it visits them, named by byte offset / 8
            // (2^21 nodes at most -- the float32-index check above -- so the eight copies end below 2^29 bytes)
            const size_t n = nodes.size();
            std::vector<PackedNode> copies(n * 8);
            for (uint32_t o = 0; o < 8; o++) {
                PackedNode *c = copies.data() + (size_t)o * n;
                for (size_t k = 0; k < n; k++) {
                    PackedNode pn = nodes[k];
                    for (int axis = 0; axis < 3; axis++)
                        if (!((o >> axis) & 1u))
                            std::swap(pn.lo[axis], pn.hi[axis]);
                    if (!(pn.b & kLeafFlag)) {
                        const uint32_t axis = pn.a >> 30, pos = (pn.a & kChildMask) << (kNodeShift - kNodeNameShift),
                                       neg = pn.b << (kNodeShift - kNodeNameShift);
                        const bool negative_first = (o >> axis) & 1u;     // D[axis] > 0 (a zero component: visit_decision)
                        pn.a = (1u << (kAxisHotShift + axis)) | (negative_first ? neg : pos);
                        pn.b = negative_first ? pos : neg;
                    }
                    // the record's words in the order the visit's packed arithmetic wants them in its register pairs
                    // (packed_layout.h: DeviceNode): { entry.x, entry.y, exit.x, exit.y } { entry.z, exit.z, a, b }
                    DeviceNode dn;
                    dn.entry_xy[0] = pn.lo[0];
                    dn.entry_xy[1] = pn.lo[1];
                    dn.exit_xy[0] = pn.hi[0];
                    dn.exit_xy[1] = pn.hi[1];
                    dn.z[0] = pn.lo[2];
                    dn.z[1] = pn.hi[2];
                    dn.a = pn.a;
                    dn.b = pn.b;
                    memcpy(&c[k], &dn, sizeof(dn));
                }
            }
            HIP_TRY(s->packed_nodes.upload(copies.data(), copies.size() * sizeof(PackedNode)));
            s->view.packed_nodes_bytes = (uint32_t)(n * sizeof(PackedNode));
        }
        HIP_TRY(s->packed_tris.upload(tris.data(), tris.size() * sizeof(PackedTri)));
        s->view.packed_root = packed_root << (kNodeShift - kNodeNameShift);
        // sibling pairs for the pair traversal: the record of an inner node holds both children's boxes and links.
        // A pair link keeps the child index in kPairIndexMask's 22 bits.  The float32-index check above already bounds
        // a scene at 2^21 nodes (8 link tables x stride <= 2^24); a tree that ever got past that keeps no pair records,
        // so pair_policy() answers false and kernel 3 runs the one-visit instances.
        if (nodes.size() <= (size_t)kPairIndexMask + 1) {
            std::vector<PackedPair> pairs(nodes.size());
            memset(pairs.data(), 0, pairs.size() * sizeof(PackedPair));
            uint32_t largest = 0;
            auto link_of = [&](uint32_t child, uint32_t *info) {
                const PackedNode &c = nodes[child];
                if (c.b & kLeafFlag) {
                    const uint32_t count = c.b & ~kLeafFlag;
                    largest = std::max(largest, count);
                    *info = c.a;
                    return child | (std::min(count, kPairCountMask) << kPairCountShift) | kLeafFlag;
                }
                *info = 0;
                return child | ((c.a >> 30) << kPairAxisShift);
            };
            for (size_t k = 0; k < nodes.size(); k++) {
                const PackedNode &pn = nodes[k];
                if (pn.b & kLeafFlag) {
                    largest = std::max(largest, pn.b & ~kLeafFlag);
                    continue;
                }
                const uint32_t pos = pn.a & kChildMask, neg = pn.b;
                PackedPair &pp = pairs[k];
                memcpy(pp.lo0, nodes[neg].lo, 12);
                memcpy(pp.hi0, nodes[neg].hi, 12);
                pp.link0 = link_of(neg, &pp.info0);
                memcpy(pp.lo1, nodes[pos].lo, 12);
                memcpy(pp.hi1, nodes[pos].hi, 12);
                pp.link1 = link_of(pos, &pp.info1);
            }
            HIP_TRY(s->pair_nodes.upload(pairs.data(), pairs.size() * sizeof(PackedPair)));
            s->max_leaf_count = largest;
            uint32_t dummy = 0;
            s->view.pair_root_link = link_of(packed_root, &dummy);
            uint32_t bits = 1;
            while ((1ull << bits) < nodes.size())
                bits++;
            s->view.pair_index_bits = bits;
        }
        // operand-range condition of exact_div.h on the scene's side: every box coordinate
        // is zero or has magnitude in [2^-70, 2^60)
        bool coords_ok = true;
        for (const PackedNode &pn : nodes) {
            for (int k = 0; k < 3 && coords_ok; k++) {
                for (float c : {pn.lo[k], pn.hi[k]}) {
                    const float m = fabsf(c);
                    if (!(c == 0.0f || (m >= 0x1p-70f && m < 0x1p60f)))
                        coords_ok = false;
                }
            }
        }
        s->view.exact_div_ok = coords_ok ? 1u : 0u;
        s->stack_levels = std::max(3, depth);   // at least three: the convergent driver stages a round of samples through levels 0-2
        s->packed_ok = true;
    }

    SceneView &v = s->view;
    v.positions = (const float *)s->positions.p;
    v.normals16 = (const uint16_t *)s->normals16.p;
    v.normals32 = (const float *)s->normals32.p;
    v.boxmin = (const float *)s->boxmin.p;
    v.boxmax = (const float *)s->boxmax.p;
    v.hitmiss = (const float *)s->hitmiss.p;
    v.objects = (const float *)s->objects.p;
    v.table_stride = (uint32_t)stride;
    v.group_count = (uint32_t)ng;
    v.triangle_count = (uint32_t)(nv / 3);
    v.tree_root = (float)desc->tree_root;
    v.packed_nodes = s->packed_nodes.p;
    v.packed_tris = s->packed_tris.p;
    v.pair_nodes = s->pair_nodes.p;
    v.env = nullptr;
    v.env_w = v.env_h = 0;

    *out_scene = s.release();
    return SHRAY_OK;
}

// ---- the same scene from a tree that never left the device ----------------------------------------------------------------
// shray_bvh_build_device -> shray_flatten_device_tree -> here (round 6; SURVEY 8(f) rank 3: scene turnaround,
// world.cpp:46-134, :298-347; bvh.cpp:288-358).  shray_scene_create takes the reference's arrays from the HOST, proves that the
// eight (hit, miss) tables are one threaded binary tree (TreeBuilder) and derives the packed tree, its eight octant copies, the
// packed triangles, the pair records and the fp16 normals on the host.  Here the tree exists as arrays already -- the tables were
// threaded FROM it, by flatten.hip, so there is nothing to recover --, and everything derived from it is computed where it lies:
// one thread per node / triangle / corner.  Packed order = the tree's pre-order (TreeBuilder::pack walks the same way: a node,
// its negative subtree, its positive subtree).  The arrays equal shray_scene_create's bit for bit (tests/test_gpu_scene_device.py
// reads both back).
namespace {

struct SceneFromDeviceFacts {     // what the kernels report back: 16 bytes, one copy
    int depth;                    // deepest ray stack over the eight direction codes (TreeBuilder::tables_match's `depth`)
    uint32_t largest_leaf;
    int coords_out_of_range;      // a box coordinate outside exact_div.h's operand ranges
    int not_canonical;            // a split direction that is not a positive unit axis vector
};

__global__ void sd_pack_nodes(int n, const int *__restrict__ negative, const int *__restrict__ positive, const int *__restrict__ start,
                              const int *__restrict__ triangles, const float *__restrict__ direction, const int *__restrict__ index_of,
                              const float *__restrict__ boxmin, const float *__restrict__ boxmax, PackedNode *__restrict__ nodes,
                              PackedNode *__restrict__ copies, SceneFromDeviceFacts *facts)
{
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n)
        return;
    const size_t me = (size_t)index_of[k];
    PackedNode pn;
    for (int a = 0; a < 3; a++) {
        pn.lo[a] = boxmin[3 * me + a];
        pn.hi[a] = boxmax[3 * me + a];
        const float lo_m = fabsf(pn.lo[a]), hi_m = fabsf(pn.hi[a]);
        if (!(pn.lo[a] == 0.0f || (lo_m >= 0x1p-70f && lo_m < 0x1p60f)) || !(pn.hi[a] == 0.0f || (hi_m >= 0x1p-70f && hi_m < 0x1p60f)))
            facts->coords_out_of_range = 1;
    }
    uint32_t axis = 0;
    if (negative[k] >= 0) {
        // a branch: the split direction is a positive unit axis vector (bvh.cpp:213-228 and bvh_build.hip make no other); the
        // child a ray of direction code 0 visits first is then the positive one (world.cpp:259-265), which is how TreeBuilder
        // tells the children apart
        const float *d = direction + 3 * (size_t)k;
        const int nonzero = (d[0] != 0.0f) + (d[1] != 0.0f) + (d[2] != 0.0f);
        axis = d[0] != 0.0f ? 0u : (d[1] != 0.0f ? 1u : 2u);
        if (nonzero != 1 || !(d[axis] > 0.0f))
            facts->not_canonical = 1;
        pn.a = (axis << 30) | (uint32_t)positive[k];
        pn.b = (uint32_t)negative[k];
    } else {
        pn.a = (uint32_t)start[k];
        pn.b = kLeafFlag | (uint32_t)triangles[k];
        atomicMax(&facts->largest_leaf, (uint32_t)triangles[k]);
    }
    nodes[k] = pn;
    // the eight octant copies (packed_layout.h), exactly as shray_scene_create makes them
    for (uint32_t o = 0; o < 8; o++) {
        PackedNode c = pn;
        for (int a = 0; a < 3; a++)
            if (!((o >> a) & 1u)) {
                const float t = c.lo[a];
                c.lo[a] = c.hi[a];
                c.hi[a] = t;
            }
        if (!(c.b & kLeafFlag)) {
            const uint32_t pos = (c.a & kChildMask) << (kNodeShift - kNodeNameShift), neg = c.b << (kNodeShift - kNodeNameShift);
            const bool negative_first = (o >> axis) & 1u;
            c.a = (1u << (kAxisHotShift + axis)) | (negative_first ? neg : pos);
            c.b = negative_first ? pos : neg;
        }
        DeviceNode dn;
        dn.entry_xy[0] = c.lo[0];
        dn.entry_xy[1] = c.lo[1];
        dn.exit_xy[0] = c.hi[0];
        dn.exit_xy[1] = c.hi[1];
        dn.z[0] = c.lo[2];
        dn.z[1] = c.hi[2];
        dn.a = c.a;
        dn.b = c.b;
        *reinterpret_cast<DeviceNode *>(&copies[(size_t)o * n + k]) = dn;
    }
}

// the deepest stack any ray can ask for: at a branch g a ray of direction code c holds one pending far child for every ancestor
// (g included) whose NEAR child its path went through (TreeBuilder::tables_match counts the same while it re-threads)
__global__ void sd_stack_depth(int n, const int *__restrict__ parent, const int *__restrict__ negative, const PackedNode *__restrict__ nodes,
                               SceneFromDeviceFacts *facts)
{
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= n || negative[g] < 0)
        return;
    int pending[8] = {1, 1, 1, 1, 1, 1, 1, 1};      // g's own far child
    for (int child = g, p = parent[g]; p >= 0; child = p, p = parent[p]) {
        const uint32_t axis = nodes[p].a >> 30;
        const bool child_is_negative = child == negative[p];
        for (int c = 0; c < 8; c++)
            pending[c] += (((c >> axis) & 1) != 0) == child_is_negative ? 1 : 0;    // code bit set: the negative child is the near one
    }
    int deepest = 0;
    for (int c = 0; c < 8; c++)
        deepest = max(deepest, pending[c]);
    atomicMax(&facts->depth, deepest);
}

__global__ void sd_pack_triangles(size_t nt, const float *__restrict__ positions, PackedTri *__restrict__ tris)
{
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t > nt)
        return;
    PackedTri pt;
    memset(&pt, 0, sizeof(pt));
    if (t < nt) {           // (t == nt: the spare record behind the last triangle, zeros)
        const float *v = positions + 9 * t;
        for (int a = 0; a < 3; a++) {
            pt.v0[a] = v[a];
            pt.e0[a] = v[3 + a] - v[a];        // e0 = v1 - v0, raytracer.es.fs:304
            pt.e1[a] = v[a] - v[6 + a];        // e1 = v0 - v2, raytracer.es.fs:305
        }
    }
    tris[t] = pt;
}

__global__ void sd_half_normals(size_t count, const float *__restrict__ normals, uint16_t *__restrict__ halves)
{
    const size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k < count)
        halves[k] = float_to_half_bits(normals[k]);
}

__device__ uint32_t sd_pair_link(const PackedNode *nodes, uint32_t child, uint32_t *info)
{
    const PackedNode &c = nodes[child];
    if (c.b & kLeafFlag) {
        const uint32_t count = c.b & ~kLeafFlag;
        *info = c.a;
        return child | (min(count, kPairCountMask) << kPairCountShift) | kLeafFlag;
    }
    *info = 0;
    return child | ((c.a >> 30) << kPairAxisShift);
}

__global__ void sd_pair_records(int n, const PackedNode *__restrict__ nodes, PackedPair *__restrict__ pairs)
{
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n)
        return;
    PackedPair pp;
    memset(&pp, 0, sizeof(pp));
    const PackedNode pn = nodes[k];
    if (!(pn.b & kLeafFlag)) {
        const uint32_t pos = pn.a & kChildMask, neg = pn.b;
        for (int a = 0; a < 3; a++) {
            pp.lo0[a] = nodes[neg].lo[a];
            pp.hi0[a] = nodes[neg].hi[a];
            pp.lo1[a] = nodes[pos].lo[a];
            pp.hi1[a] = nodes[pos].hi[a];
        }
        pp.link0 = sd_pair_link(nodes, neg, &pp.info0);
        pp.link1 = sd_pair_link(nodes, pos, &pp.info1);
    }
    pairs[k] = pp;
}

}   // namespace

int shray_scene_create_from_device(const shray_device_tree *tree, const shray_device_flat *flat, shray_scene **out_scene)
{
    if (!tree || !flat || !out_scene)
        return fail(SHRAY_ERR_INVALID_ARGUMENT, "tree, flat or out_scene is NULL");
    *out_scene = nullptr;
    ShrayDeviceTreeView t;
    ShrayDeviceFlatView f;
    if (const int rc = shrayi_device_tree_view(tree, &t))
        return rc;
    if (const int rc = shrayi_device_flat_view(flat, &f))
        return rc;
    const shray_scene_desc &desc = f.desc;
    if (desc.group_count != t.node_count || desc.vertex_count != 3u * (uint32_t)t.triangle_count || !f.index_of)
        return fail(SHRAY_ERR_INVALID_ARGUMENT, "the flattened arrays are not those of this tree (%d nodes / %d, %u corners / %d triangles)",
                    desc.group_count, t.node_count, desc.vertex_count, t.triangle_count);
    const uint64_t stride = (uint64_t)desc.data_texture_width * (uint64_t)desc.group_data_rows;
    // the shader's float32 indices (shray_scene_create)
    if (stride * 8 > 16777216ull || desc.vertex_count > 16777216u)
        return fail(SHRAY_ERR_INDEX_RANGE, "scene too large for float32 indices (%llu link texels, %u vertices; limit 2^24)",
                    (unsigned long long)(stride * 8), desc.vertex_count);
    int device = 0;
    if (hipGetDevice(&device) != hipSuccess)
        return fail(SHRAY_ERR_NO_DEVICE, "no HIP device is available (hipGetDevice failed)");

    std::unique_ptr<shray_scene> s(new shray_scene);
    s->device = device;
    const size_t nv = desc.vertex_count, ng = (size_t)desc.group_count, nt = nv / 3;
    const int n = t.node_count, block = 256;

    // the reference's arrays (the literal kernel's inputs): copies on the device
    HIP_TRY(s->positions.copy_of(desc.vertex_positions, nv * 12));
    HIP_TRY(s->normals32.copy_of(desc.vertex_normals, nv * 12));
    HIP_TRY(s->boxmin.copy_of(desc.group_boxmin, ng * 12));
    HIP_TRY(s->boxmax.copy_of(desc.group_boxmax, ng * 12));
    HIP_TRY(s->objects.copy_of(desc.group_objects, ng * 8));
    HIP_TRY(s->hitmiss.copy_of(desc.group_hitmiss, (size_t)stride * 8 * 8));
    HIP_TRY(s->counters.upload(nullptr, sizeof(DeviceCounters) * kCounterShards));
    HIP_TRY(s->normals16.reserve(nv * 3 * 2));
    if (nv)
        hipLaunchKernelGGL(sd_half_normals, dim3((unsigned)((nv * 3 + block - 1) / block)), dim3(block), 0, nullptr, nv * 3,
                           (const float *)s->normals32.p, (uint16_t *)s->normals16.p);

    // the packed tree, its octant copies, the packed triangles, the pair records
    DeviceBuffer d_nodes, d_facts;
    HIP_TRY(d_nodes.reserve((size_t)n * sizeof(PackedNode)));
    HIP_TRY(d_facts.upload(nullptr, sizeof(SceneFromDeviceFacts)));
    HIP_TRY(s->packed_nodes.reserve((size_t)n * 8 * sizeof(PackedNode)));
    HIP_TRY(s->packed_tris.reserve((nt + 1) * sizeof(PackedTri)));
    const dim3 node_grid((unsigned)((n + block - 1) / block));
    hipLaunchKernelGGL(sd_pack_nodes, node_grid, dim3(block), 0, nullptr, n, t.negative, t.positive, t.start, t.triangles, t.direction, f.index_of,
                       desc.group_boxmin, desc.group_boxmax, (PackedNode *)d_nodes.p, (PackedNode *)s->packed_nodes.p,
                       (SceneFromDeviceFacts *)d_facts.p);
    hipLaunchKernelGGL(sd_stack_depth, node_grid, dim3(block), 0, nullptr, n, t.parent, t.negative, (const PackedNode *)d_nodes.p,
                       (SceneFromDeviceFacts *)d_facts.p);
    hipLaunchKernelGGL(sd_pack_triangles, dim3((unsigned)((nt + 1 + block - 1) / block)), dim3(block), 0, nullptr, nt,
                       (const float *)s->positions.p, (PackedTri *)s->packed_tris.p);
    const bool pairs = (size_t)n <= (size_t)kPairIndexMask + 1;
    if (pairs) {
        HIP_TRY(s->pair_nodes.reserve((size_t)n * sizeof(PackedPair)));
        hipLaunchKernelGGL(sd_pair_records, node_grid, dim3(block), 0, nullptr, n, (const PackedNode *)d_nodes.p, (PackedPair *)s->pair_nodes.p);
    }
    HIP_TRY(hipGetLastError());
    SceneFromDeviceFacts facts;
    HIP_TRY(hipMemcpy(&facts, d_facts.p, sizeof(facts), hipMemcpyDeviceToHost));   // (waits for the kernels and copies above)
    if (facts.not_canonical)
        return fail(SHRAY_ERR_BAD_TREE, "a split direction of the device tree is not a positive unit axis vector: take the host path "
                                        "(shray_device_tree_download, shray_scene_create)");
    s->view.packed_nodes_bytes = (uint32_t)((size_t)n * sizeof(PackedNode));
    s->view.packed_root = 0;      // pre-order: the root is record 0
    if (pairs) {
        s->max_leaf_count = facts.largest_leaf;
        // (the root's own link: a leaf root names its triangles, a branch root its split axis -- one record read back)
        PackedNode root;
        HIP_TRY(hipMemcpy(&root, d_nodes.p, sizeof(root), hipMemcpyDeviceToHost));
        s->view.pair_root_link = (root.b & kLeafFlag) ? (0u | (std::min(root.b & ~kLeafFlag, kPairCountMask) << kPairCountShift) | kLeafFlag)
                                                      : (0u | ((root.a >> 30) << kPairAxisShift));
        uint32_t bits = 1;
        while ((1ull << bits) < (size_t)n)
            bits++;
        s->view.pair_index_bits = bits;
    }
    s->view.exact_div_ok = facts.coords_out_of_range ? 0u : 1u;
    s->stack_levels = std::max(3, facts.depth);
    s->packed_ok = true;

    SceneView &v = s->view;
    v.positions = (const float *)s->positions.p;
    v.normals16 = (const uint16_t *)s->normals16.p;
    v.normals32 = (const float *)s->normals32.p;
    v.boxmin = (const float *)s->boxmin.p;
    v.boxmax = (const float *)s->boxmax.p;
    v.hitmiss = (const float *)s->hitmiss.p;
    v.objects = (const float *)s->objects.p;
    v.table_stride = (uint32_t)stride;
    v.group_count = (uint32_t)ng;
    v.triangle_count = (uint32_t)nt;
    v.tree_root = (float)desc.tree_root;
    v.packed_nodes = s->packed_nodes.p;
    v.packed_tris = s->packed_tris.p;
    v.pair_nodes = s->pair_nodes.p;
    v.env = nullptr;
    v.env_w = v.env_h = 0;
    *out_scene = s.release();
    return SHRAY_OK;
}

int shray_scene_derived_sizes(const shray_scene *scene, uint64_t *packed_nodes_bytes, uint64_t *packed_tris_bytes, uint64_t *normals16_bytes,
                              uint64_t *pair_nodes_bytes, int32_t *stack_levels)
{
    if (!scene)
        return fail(SHRAY_ERR_INVALID_ARGUMENT, "scene is NULL");
    const SceneView &v = scene->view;
    if (packed_nodes_bytes)
        *packed_nodes_bytes = scene->packed_ok ? (uint64_t)v.packed_nodes_bytes * 8u : 0u;
    if (packed_tris_bytes)
        *packed_tris_bytes = scene->packed_ok ? (uint64_t)v.triangle_count * sizeof(PackedTri) : 0u;
    if (normals16_bytes)
        *normals16_bytes = (uint64_t)v.triangle_count * 3u * 3u * 2u;
    if (pair_nodes_bytes)
        *pair_nodes_bytes = (scene->packed_ok && scene->pair_nodes.p) ? (uint64_t)(v.packed_nodes_bytes / sizeof(PackedNode)) * sizeof(PackedPair) : 0u;
    if (stack_levels)
        *stack_levels = scene->stack_levels;
    return SHRAY_OK;
}

int shray_scene_derived_download(const shray_scene *scene, void *packed_nodes, void *packed_tris, void *normals16, void *pair_nodes)
{
    if (!scene)
        return fail(SHRAY_ERR_INVALID_ARGUMENT, "scene is NULL");
    uint64_t a = 0, b = 0, c = 0, d = 0;
    shray_scene_derived_sizes(scene, &a, &b, &c, &d, nullptr);
    HIP_TRY(hipSetDevice(scene->device));
    if (packed_nodes && a)
        HIP_TRY(hipMemcpy(packed_nodes, scene->packed_nodes.p, a, hipMemcpyDeviceToHost));
    if (packed_tris && b)
        HIP_TRY(hipMemcpy(packed_tris, scene->packed_tris.p, b, hipMemcpyDeviceToHost));
    if (normals16 && c)
        HIP_TRY(hipMemcpy(normals16, scene->normals16.p, c, hipMemcpyDeviceToHost));
    if (pair_nodes && d)
        HIP_TRY(hipMemcpy(pair_nodes, scene->pair_nodes.p, d, hipMemcpyDeviceToHost));
    return SHRAY_OK;
}

// GL's conversion of a float to 8-bit normalized fixed point and back (GL 3.1 section 2.1.5): clamp to [0, 1],
// c = floor(255 f + 0.5), stored value c / 255; NaN stores 0
static float through_unorm8(float f)
{
    const float clamped = !(f > 0.0f) ? 0.0f : (f > 1.0f ? 1.0f : f);
    return floorf(clamped * 255.0f + 0.5f) / 255.0f;
}

int shray_scene_set_environment(shray_scene *scene, const float *rgb, int width, int height)
{
    return shray_scene_set_environment_storage(scene, rgb, width, height, SHRAY_ENV_FLOAT32);
}

int shray_scene_set_environment_storage(shray_scene *scene, const float *rgb, int width, int height, int storage)
{
    if (!scene || !rgb || width <= 0 || height <= 0 || width > 32768 || height > 32768)
        return fail(SHRAY_ERR_INVALID_ARGUMENT, "bad environment (%p, %d x %d)", (const void *)rgb, width, height);
    if (storage != SHRAY_ENV_FLOAT32 && storage != SHRAY_ENV_UNORM8)
        return fail(SHRAY_ERR_INVALID_ARGUMENT, "environment storage %d (SHRAY_ENV_FLOAT32 or SHRAY_ENV_UNORM8)", storage);
    HIP_TRY(hipSetDevice(scene->device));
    // level 0 followed by its mip chain (2x2 box filter, ((a+b)+(c+d))*0.25, dimensions max(1, n/2)),
    // the pyramid the reference asks GL for with glGenerateMipmap (ray.cpp:509); only the which == 1
    // view reads levels above 0
    std::vector<float> pyramid(rgb, rgb + (size_t)width * height * 3);
    if (storage == SHRAY_ENV_UNORM8)
        for (float &texel : pyramid)
            texel = through_unorm8(texel);
    SceneView &v = scene->view;
    v.mip_levels = 0;
    size_t level_start = 0;
    int w = width, h = height;
    for (;;) {
        v.mip_offset[v.mip_levels] = (uint32_t)level_start;
        v.mip_w[v.mip_levels] = w;
        v.mip_h[v.mip_levels] = h;
        v.mip_levels++;
        if ((w == 1 && h == 1) || v.mip_levels == 16)
            break;
        const int nw = std::max(1, w / 2), nh = std::max(1, h / 2);
        const size_t next_start = pyramid.size();
        pyramid.resize(next_start + (size_t)nw * nh * 3);
        const float *src = pyramid.data() + level_start;
        float *dst = pyramid.data() + next_start;
        for (int j = 0; j < nh; j++)
            for (int i = 0; i < nw; i++) {
                const int i0 = std::min(2 * i, w - 1), i1 = std::min(2 * i + 1, w - 1);
                const int j0 = std::min(2 * j, h - 1), j1 = std::min(2 * j + 1, h - 1);
                for (int c = 0; c < 3; c++) {
                    const float a = src[3 * ((size_t)j0 * w + i0) + c], b = src[3 * ((size_t)j0 * w + i1) + c];
                    const float cc = src[3 * ((size_t)j1 * w + i0) + c], d = src[3 * ((size_t)j1 * w + i1) + c];
                    const float mean = ((a + b) + (cc + d)) * 0.25f;
                    dst[3 * ((size_t)j * nw + i) + c] = storage == SHRAY_ENV_UNORM8 ? through_unorm8(mean) : mean;   // every level is stored
                }
            }
        level_start = next_start;
        w = nw;
        h = nh;
    }
    HIP_TRY(scene->env.upload(pyramid.data(), pyramid.size() * sizeof(float)));
    scene->view.env = (const float *)scene->env.p;
    scene->view.env_w = width;
    scene->view.env_h = height;
    return SHRAY_OK;
}

int shray_scene_destroy(shray_scene *scene)
{
    if (!scene)
        return SHRAY_OK;
    (void)hipSetDevice(scene->device);
    delete scene;
    return SHRAY_OK;
}

int shray_scene_device(const shray_scene *scene, int *device_index)
{
    if (!scene || !device_index)
        return fail(SHRAY_ERR_INVALID_ARGUMENT, "scene or device_index is NULL");
    *device_index = scene->device;
    return SHRAY_OK;
}

int shray_scene_set_kernel(shray_scene *scene, int kernel_id)
{
    if (!scene || kernel_id < 0 || kernel_id > 4)
        return fail(SHRAY_ERR_INVALID_ARGUMENT, "kernel id %d (0 = stack, 1 = threaded, 2 = pool, 3 = stack with pair turns, 4 = wavefront)",
                    kernel_id);
    if (kernel_id != 1 && !scene->packed_ok)
        return fail(SHRAY_ERR_BAD_TREE, "the scene's hit/miss tables are not a canonical threaded tree; only the "
                    "literal threaded kernel (1) can run it");
    scene->kernel_id = kernel_id;
    return SHRAY_OK;
}

int64_t shray_tile_buffer_bytes(int width, int height, const shray_tile_set *tiles)
{
    if (width <= 0 || height <= 0)
        return 0;
    if (!tiles || tiles->tile_stride <= 0)
        return (int64_t)width * height * 16;
    const int phase_count = tiles->tile_phase_count > 0 ? tiles->tile_phase_count : 1;
    if (tiles->tile_w <= 0 || tiles->tile_h <= 0 || tiles->tile_phase < 0 || tiles->tile_phase_count < 0 ||
        tiles->tile_phase + phase_count > tiles->tile_stride)
        return 0;
    const int64_t tx = (width + tiles->tile_w - 1) / tiles->tile_w, ty = (height + tiles->tile_h - 1) / tiles->tile_h;
    return owned_tile_count(tx * ty, tiles->tile_stride, tiles->tile_phase, phase_count) * tiles->tile_w * tiles->tile_h * 16;
}

int shray_render_device(shray_scene *scene, const shray_frame_params *params, int width, int height, int spp,
                        const shray_tile_set *tiles, void *d_rgba_out, void *hip_stream)
{
    if (!scene || !d_rgba_out)
        return fail(SHRAY_ERR_INVALID_ARGUMENT, "scene or output buffer is NULL");
    int rc = validate_params(params, width, height, spp);
    if (rc)
        return rc;
    if (!scene->view.env)
        return fail(SHRAY_ERR_NO_ENVIRONMENT, "no environment set; call shray_scene_set_environment first");
    FrameView fr;
    rc = make_frame_view(params, width, height, spp, tiles, &fr);
    if (rc)
        return rc;
    int current = -1;
    if (hipGetDevice(&current) != hipSuccess || current != scene->device)
        HIP_TRY(hipSetDevice(scene->device));   // the scene's buffers and the stream live on its device
    return launch(scene, fr, (float4 *)d_rgba_out, nullptr, (hipStream_t)hip_stream);
}

int shray_render_batch_device(shray_scene *scene, const shray_frame_params *params, int count, int width, int height,
                              int spp, const shray_tile_set *tiles, void *d_rgba_out, int64_t frame_stride_bytes,
                              void *hip_stream)
{
    if (!scene || !d_rgba_out || !params)
        return fail(SHRAY_ERR_INVALID_ARGUMENT, "scene, params or output buffer is NULL");
    if (count < 1 || count > SHRAY_MAX_BATCH)
        return fail(SHRAY_ERR_INVALID_ARGUMENT, "batch of %d frames (1..%d allowed)", count, SHRAY_MAX_BATCH);
    if (!scene->view.env)
        return fail(SHRAY_ERR_NO_ENVIRONMENT, "no environment set; call shray_scene_set_environment first");
    const int64_t frame_bytes = shray_tile_buffer_bytes(width, height, tiles);
    if (frame_stride_bytes < frame_bytes || frame_stride_bytes % 16 != 0)
        return fail(SHRAY_ERR_INVALID_ARGUMENT, "frame_stride_bytes %lld must be a multiple of 16 and at least the %lld "
                    "bytes of one frame", (long long)frame_stride_bytes, (long long)frame_bytes);
    std::vector<FrameView> views((size_t)count);
    for (int k = 0; k < count; k++) {
        int rc = validate_params(params + k, width, height, spp);
        if (rc)
            return rc;
        rc = make_frame_view(params + k, width, height, spp, tiles, &views[k]);
        if (rc)
            return rc;
        const bool diff0 = views[0].which == 1 || views[0].which == 2, diffk = views[k].which == 1 || views[k].which == 2;
        if (diff0 != diffk)
            return fail(SHRAY_ERR_INVALID_ARGUMENT, "frames %d and 0 of a batch disagree on differential views "
                        "(which = %d vs %d)", k, views[k].which, views[0].which);
    }
    int current = -1;
    if (hipGetDevice(&current) != hipSuccess || current != scene->device)
        HIP_TRY(hipSetDevice(scene->device));
    hipStream_t stream = (hipStream_t)hip_stream;
    char *out = (char *)d_rgba_out;

    // anything but the stack kernel runs as plain consecutive launches
    if (scene->kernel_id == 1 || scene->kernel_id == 4 || !scene->packed_ok) {
        for (int k = 0; k < count; k++) {
            const int rc = launch(scene, views[k], (float4 *)(out + (size_t)k * frame_stride_bytes), nullptr, stream);
            if (rc)
                return rc;
        }
        return SHRAY_OK;
    }
    return launch_stack_views(scene, views.data(), count, (float4 *)d_rgba_out, (size_t)frame_stride_bytes / 16, stream);
}

int shray_assemble_tiles_split_device(const void *d_gathered, int world, int rank0_phases, int other_phases, int frames,
                                      int channels, int64_t rank_stride_bytes, int64_t frame_stride_bytes, int width, int height,
                                      int tile_w, int tile_h, void *d_rgba_out, void *hip_stream)
{
    if (!d_gathered || !d_rgba_out)
        return fail(SHRAY_ERR_INVALID_ARGUMENT, "gathered or output buffer is NULL");
    if (world < 1 || frames < 1 || frames > 65535 || (channels != 3 && channels != 4) || width <= 0 || height <= 0 ||
        width > 65536 || height > 65535 || tile_w <= 0 || tile_h <= 0 || rank0_phases < 1 || other_phases < 1 ||
        rank0_phases > 4096 || other_phases > 4096)
        return fail(SHRAY_ERR_INVALID_ARGUMENT, "bad assemble geometry (world %d, shares %d / %d, %d frames, %d channels, %dx%d, "
                    "tiles %dx%d)", world, rank0_phases, other_phases, frames, channels, width, height, tile_w, tile_h);
    const int64_t tiles = (int64_t)((width + tile_w - 1) / tile_w) * ((height + tile_h - 1) / tile_h);
    const int64_t period = rank0_phases + (int64_t)(world - 1) * other_phases;
    int64_t most = owned_tile_count(tiles, period, 0, rank0_phases);
    for (int r = 1; r < world; r++)
        most = std::max(most, owned_tile_count(tiles, period, rank0_phases + (int64_t)(r - 1) * other_phases, other_phases));
    const int64_t frame_bytes = most * tile_w * tile_h * channels * 4;
    if (frame_stride_bytes < frame_bytes || frame_stride_bytes % 4 || rank_stride_bytes % 4 ||
        rank_stride_bytes < (int64_t)(frames - 1) * frame_stride_bytes + frame_bytes)
        return fail(SHRAY_ERR_INVALID_ARGUMENT, "strides too small: a rank's frame takes %lld bytes (frame stride %lld, "
                    "rank stride %lld)", (long long)frame_bytes, (long long)frame_stride_bytes, (long long)rank_stride_bytes);
    const hipError_t e = launch_assemble_tiles((const float *)d_gathered, (float4 *)d_rgba_out, world, rank0_phases, other_phases,
                                               frames, channels, width, height, tile_w, tile_h, (size_t)rank_stride_bytes / 4,
                                               (size_t)frame_stride_bytes / 4, (hipStream_t)hip_stream);
    if (e != hipSuccess)
        return fail(SHRAY_ERR_DEVICE, "kernel launch failed: %s", hipGetErrorString(e));
    return SHRAY_OK;
}

int shray_assemble_tiles_device(const void *d_gathered, int world, int frames, int channels, int64_t rank_stride_bytes,
                                int64_t frame_stride_bytes, int width, int height, int tile_w, int tile_h,
                                void *d_rgba_out, void *hip_stream)
{
    return shray_assemble_tiles_split_device(d_gathered, world, 1, 1, frames, channels, rank_stride_bytes, frame_stride_bytes, width,
                                             height, tile_w, tile_h, d_rgba_out, hip_stream);
}

namespace {
int ensure_frame(shray_scene *scene, size_t bytes)
{
    if (scene->frame_bytes < bytes) {
        // plain allocation: a fill on the null stream would not be ordered against the kernels that write the
        // frame on other (non-blocking) streams, and every pixel is written by the render anyway
        scene->frame.release();
        scene->frame_bytes = 0;
        HIP_TRY(hipMalloc(&scene->frame.p, bytes));
        scene->frame_bytes = bytes;
    }
    if (!scene->readback_stream)
        HIP_TRY(hipStreamCreateWithFlags(&scene->readback_stream, hipStreamNonBlocking));
    return SHRAY_OK;
}

bool is_pinned_host(const void *p)
{
    hipPointerAttribute_t attr;
    if (hipPointerGetAttributes(&attr, p) != hipSuccess) {
        (void)hipGetLastError();   // pageable memory: not an error for us
        return false;
    }
    return attr.type == hipMemoryTypeHost;
}
}   // namespace

int shray_render(shray_scene *scene, const shray_frame_params *params, int width, int height, int spp,
                 float *rgba_out_host)
{
    if (!scene || !rgba_out_host)
        return fail(SHRAY_ERR_INVALID_ARGUMENT, "scene or output buffer is NULL");
    int rc = validate_params(params, width, height, spp);
    if (rc)
        return rc;
    HIP_TRY(hipSetDevice(scene->device));
    const size_t bytes = (size_t)width * height * 16;
    rc = ensure_frame(scene, bytes);
    if (rc)
        return rc;
    hipStream_t stream = scene->readback_stream;
    rc = shray_render_device(scene, params, width, height, spp, nullptr, scene->frame.p, stream);
    if (rc)
        return rc;
    // pinned destination (shray_pinned_alloc): one DMA straight into the caller's buffer; pageable: the runtime
    // stages the copy through its own pinned buffers -- the same call either way
    HIP_TRY(hipMemcpyAsync(rgba_out_host, scene->frame.p, bytes, hipMemcpyDeviceToHost, stream));
    HIP_TRY(hipStreamSynchronize(stream));
    return SHRAY_OK;
}

int shray_render_host_async(shray_scene *scene, const shray_frame_params *params, int width, int height, int spp,
                            float *rgba_out_pinned, void *hip_stream)
{
    if (!scene || !rgba_out_pinned)
        return fail(SHRAY_ERR_INVALID_ARGUMENT, "scene or output buffer is NULL");
    int rc = validate_params(params, width, height, spp);
    if (rc)
        return rc;
    HIP_TRY(hipSetDevice(scene->device));
    if (!is_pinned_host(rgba_out_pinned))
        return fail(SHRAY_ERR_INVALID_ARGUMENT, "shray_render_host_async needs pinned host memory (shray_pinned_alloc); "
                    "use shray_render for pageable buffers");
    const size_t bytes = (size_t)width * height * 16;
    rc = ensure_frame(scene, bytes);
    if (rc)
        return rc;
    rc = shray_render_device(scene, params, width, height, spp, nullptr, scene->frame.p, hip_stream);
    if (rc)
        return rc;
    HIP_TRY(hipMemcpyAsync(rgba_out_pinned, scene->frame.p, bytes, hipMemcpyDeviceToHost, (hipStream_t)hip_stream));
    return SHRAY_OK;
}

int shray_pinned_alloc(size_t bytes, void **out_ptr)
{
    if (!out_ptr || bytes == 0)
        return fail(SHRAY_ERR_INVALID_ARGUMENT, "shray_pinned_alloc: NULL result pointer or zero size");
    *out_ptr = nullptr;
    HIP_TRY(hipHostMalloc(out_ptr, bytes, hipHostMallocDefault));
    return SHRAY_OK;
}

int shray_pinned_free(void *ptr)
{
    if (ptr)
        HIP_TRY(hipHostFree(ptr));
    return SHRAY_OK;
}

int shray_render_counters(shray_scene *scene, const shray_frame_params *params, int width, int height, int spp,
                          float *rgba_out_host, shray_counters *counters)
{
    if (!scene || !counters)
        return fail(SHRAY_ERR_INVALID_ARGUMENT, "scene or counters is NULL");
    int rc = validate_params(params, width, height, spp);
    if (rc)
        return rc;
    if (!scene->view.env)
        return fail(SHRAY_ERR_NO_ENVIRONMENT, "no environment set; call shray_scene_set_environment first");
    HIP_TRY(hipSetDevice(scene->device));
    FrameView fr;
    rc = make_frame_view(params, width, height, spp, nullptr, &fr);
    if (rc)
        return rc;
    DeviceBuffer frame;
    const size_t bytes = (size_t)width * height * 16;
    HIP_TRY(frame.upload(nullptr, bytes));
    HIP_TRY(hipMemset(scene->counters.p, 0, sizeof(DeviceCounters) * kCounterShards));
    rc = launch(scene, fr, (float4 *)frame.p, (DeviceCounters *)scene->counters.p, nullptr);
    if (rc)
        return rc;
    HIP_TRY(hipDeviceSynchronize());
    DeviceCounters shards[kCounterShards];
    HIP_TRY(hipMemcpy(shards, scene->counters.p, sizeof(shards), hipMemcpyDeviceToHost));
    DeviceCounters dc = {};
    for (const DeviceCounters &sh : shards) {
        dc.node_visits += sh.node_visits;
        dc.leaf_visits += sh.leaf_visits;
        dc.triangle_tests += sh.triangle_tests;
        dc.shaded_hits += sh.shaded_hits;
        dc.env_lookups += sh.env_lookups;
        dc.traversals += sh.traversals;
        dc.bad_hits += sh.bad_hits;
    }
    if (rgba_out_host)
        HIP_TRY(hipMemcpy(rgba_out_host, frame.p, bytes, hipMemcpyDeviceToHost));
    counters->node_visits = dc.node_visits;
    counters->leaf_visits = dc.leaf_visits;
    counters->triangle_tests = dc.triangle_tests;
    counters->shaded_hits = dc.shaded_hits;
    counters->env_lookups = dc.env_lookups;
    counters->traversals = dc.traversals;
    counters->bad_hits = dc.bad_hits;
    counters->samples = (uint64_t)width * height * spp;
    return SHRAY_OK;
}

int shray_render_counters_timed(shray_scene *scene, const shray_frame_params *params, int width, int height, int spp,
                                int frames_per_launch, float *rgba_out_host, shray_counters *counters)
{
    if (!scene || !counters)
        return fail(SHRAY_ERR_INVALID_ARGUMENT, "scene or counters is NULL");
    int rc = validate_params(params, width, height, spp);
    if (rc)
        return rc;
    if (frames_per_launch < 1 || frames_per_launch > SHRAY_MAX_BATCH)
        return fail(SHRAY_ERR_INVALID_ARGUMENT, "frames_per_launch %d (1..%d)", frames_per_launch, SHRAY_MAX_BATCH);
    const bool view = params->which == 1 || params->which == 2 || params->which == 3 || params->which == 5;
    // only the stack kernel's convergent instances have a timed form of their own; everything else is timed as it counts
    if ((scene->kernel_id != 0 && scene->kernel_id != 3 && scene->kernel_id != 4) || !scene->packed_ok || view)
        return shray_render_counters(scene, params, width, height, spp, rgba_out_host, counters);
    if (!scene->view.env)
        return fail(SHRAY_ERR_NO_ENVIRONMENT, "no environment set; call shray_scene_set_environment first");
    HIP_TRY(hipSetDevice(scene->device));
    FrameView fr;
    rc = make_frame_view(params, width, height, spp, nullptr, &fr);
    if (rc)
        return rc;
    DeviceBuffer frame;
    const size_t bytes = (size_t)width * height * 16;
    HIP_TRY(frame.upload(nullptr, bytes));
    HIP_TRY(hipMemset(scene->counters.p, 0, sizeof(DeviceCounters) * kCounterShards));
    HIP_TRY(hipDeviceSynchronize());
    rc = launch_stack_views(scene, &fr, 1, (float4 *)frame.p, 0, nullptr, (DeviceCounters *)scene->counters.p, frames_per_launch);
    if (rc)
        return rc;
    HIP_TRY(hipDeviceSynchronize());
    DeviceCounters shards[kCounterShards];
    HIP_TRY(hipMemcpy(shards, scene->counters.p, sizeof(shards), hipMemcpyDeviceToHost));
    memset(counters, 0, sizeof(*counters));
    for (const DeviceCounters &sh : shards) {
        counters->node_visits += sh.node_visits;
        counters->leaf_visits += sh.leaf_visits;
        counters->triangle_tests += sh.triangle_tests;
        counters->shaded_hits += sh.shaded_hits;
        counters->env_lookups += sh.env_lookups;
        counters->traversals += sh.traversals;
        counters->bad_hits += sh.bad_hits;
    }
    counters->samples = (uint64_t)width * height * spp;
    if (rgba_out_host)
        HIP_TRY(hipMemcpy(rgba_out_host, frame.p, bytes, hipMemcpyDeviceToHost));
    return SHRAY_OK;
}

#ifdef SHRAY_DIAGNOSTICS
// Diagnostic build only (libshray_hip_diag.so, profiles/timeline.py): renders one frame with
// the timed kernel and returns per wave 8 x uint64 {begin, end (100 MHz ticks), xcc<<32|hw_id, 0,
// node-loop iterations, leaf-loop iterations, cycles in the node loop, cycles in the leaf loop}
// (stack kernel);
// `stamps` must hold 64 * ceil(w/16) * ceil(h/16) values.
int shray_debug_timeline(shray_scene *scene, const shray_frame_params *params, int width, int height, int spp,
                         uint64_t *stamps)
{
    int rc = validate_params(params, width, height, spp);
    if (rc)
        return rc;
    FrameView fr;
    rc = make_frame_view(params, width, height, spp, nullptr, &fr);
    if (rc)
        return rc;
    const size_t nstamps = (size_t)fr.total_patches * 64;
    DeviceBuffer frame, dbg;
    HIP_TRY(frame.upload(nullptr, (size_t)width * height * 16));
    HIP_TRY(dbg.upload(nullptr, sizeof(DeviceCounters) * kCounterShards + nstamps * 8));
    shray::g_diag_plain_kernel = true;   // stamp the timed (non-counting) kernel
    // warm caches; SHRAY_DIAG_REPS back-to-back launches (about 2 s of them) let the clock settle under load before the
    // in-kernel clock is read from the last run's stamps, which are the ones returned
    const char *reps_env = getenv("SHRAY_DIAG_REPS");
    const int reps = reps_env ? std::max(1, atoi(reps_env)) : 3;
    for (int rep = 0; rep < reps; rep++) {
        rc = launch(scene, fr, (float4 *)frame.p, (DeviceCounters *)dbg.p, nullptr);
        if (rc)
            return rc;
        HIP_TRY(hipDeviceSynchronize());
    }
    shray::g_diag_plain_kernel = false;
    HIP_TRY(hipMemcpy(stamps, (char *)dbg.p + sizeof(DeviceCounters) * kCounterShards, nstamps * 8, hipMemcpyDeviceToHost));
    return SHRAY_OK;
}
#endif


int shray_scene_dispatch_order(shray_scene *scene, uint32_t *order_out, uint32_t capacity, uint32_t *count_out)
{
    if (!scene || !count_out || (capacity && !order_out))
        return fail(SHRAY_ERR_INVALID_ARGUMENT, "scene, count_out or order_out is NULL");
    HIP_TRY(hipSetDevice(scene->device));
    *count_out = 0;
    if (scene->dispatch_last < 0)
        return SHRAY_OK;
    shray_scene::DispatchOrder &d = scene->dispatch[scene->dispatch_last];
    HIP_TRY(hipDeviceSynchronize());
    if (d.pending_count > 0) {
        d.current = d.written;
        d.pending_count = 0;
    }
    if (d.current < 0 || d.n == 0)
        return SHRAY_OK;
    const uint32_t copy = capacity < d.n ? capacity : d.n;
    if (copy)
        HIP_TRY(hipMemcpy(order_out, (const uint32_t *)d.ring.p + (size_t)d.current * d.n, (size_t)copy * 4, hipMemcpyDeviceToHost));
    *count_out = d.n;
    return SHRAY_OK;
}

int shray_selftest_division(uint64_t pairs, uint64_t seed, uint64_t *mismatches)
{
    if (!mismatches)
        return fail(SHRAY_ERR_INVALID_ARGUMENT, "mismatches is NULL");
    DeviceBuffer count;
    HIP_TRY(count.upload(nullptr, sizeof(unsigned long long)));
    hipError_t e = shray::launch_division_selftest(pairs, seed, (unsigned long long *)count.p, nullptr);
    if (e != hipSuccess)
        return fail(SHRAY_ERR_DEVICE, "self-test launch failed: %s", hipGetErrorString(e));
    HIP_TRY(hipDeviceSynchronize());
    unsigned long long n = 0;
    HIP_TRY(hipMemcpy(&n, count.p, sizeof(n), hipMemcpyDeviceToHost));
    *mismatches = n;
    return SHRAY_OK;
}

int shray_selftest_reciprocal(uint64_t *mismatches)
{
    if (!mismatches)
        return fail(SHRAY_ERR_INVALID_ARGUMENT, "mismatches is NULL");
    DeviceBuffer count;
    HIP_TRY(count.upload(nullptr, sizeof(unsigned long long)));
    hipError_t e = shray::launch_reciprocal_selftest((unsigned long long *)count.p, nullptr);
    if (e != hipSuccess)
        return fail(SHRAY_ERR_DEVICE, "self-test launch failed: %s", hipGetErrorString(e));
    HIP_TRY(hipDeviceSynchronize());
    unsigned long long n = 0;
    HIP_TRY(hipMemcpy(&n, count.p, sizeof(n), hipMemcpyDeviceToHost));
    *mismatches = n;
    return SHRAY_OK;
}

int shray_probe_vector_cache(uint32_t records, uint32_t spread, uint32_t visits_per_lane, uint32_t waves, uint64_t lane_mask, int bytes_per_lane,
                             double *seconds, uint64_t *bytes_loaded)
{
    if (!seconds || !bytes_loaded)
        return fail(SHRAY_ERR_INVALID_ARGUMENT, "seconds or bytes_loaded is NULL");
    auto power_of_two = [](uint32_t v) { return v != 0 && (v & (v - 1)) == 0; };
    const uint32_t runs = spread & 0x80000000u;
    spread &= 0x7fffffffu;
    if (!power_of_two(records) || !power_of_two(spread) || spread > records || records > (1u << 26) || visits_per_lane == 0 ||
        visits_per_lane % 8 != 0 || waves == 0 || waves > (1u << 24) ||
        !(bytes_per_lane == 32 || bytes_per_lane == 16 || bytes_per_lane == 12 || bytes_per_lane == 8 || bytes_per_lane == 4))
        return fail(SHRAY_ERR_INVALID_ARGUMENT, "shray_probe_vector_cache: records and spread are powers of two (spread <= records <= 2^26), "
                    "visits_per_lane a multiple of 8, waves 1..2^24, bytes_per_lane 32, 16, 12, 8 or 4");
    DeviceBuffer table, out;
    HIP_TRY(table.upload(nullptr, (size_t)records * 32));
    HIP_TRY(out.upload(nullptr, (size_t)waves * 64 * 4));
    hipEvent_t a, b;
    HIP_TRY(hipEventCreate(&a));
    HIP_TRY(hipEventCreate(&b));
    float ms = 0.0f;
    hipError_t e = hipSuccess;
    for (int pass = 0; pass < 2 && e == hipSuccess; pass++) {      // the first pass warms the caches and the clock
        (void)hipEventRecord(a, nullptr);
        e = shray::launch_vector_cache_probe((const float4 *)table.p, records, spread | runs, visits_per_lane, waves, (unsigned long long)lane_mask, bytes_per_lane, (float *)out.p, nullptr);
        (void)hipEventRecord(b, nullptr);
        if (e == hipSuccess)
            e = hipEventSynchronize(b);
        if (e == hipSuccess)
            e = hipEventElapsedTime(&ms, a, b);
    }
    (void)hipEventDestroy(a);
    (void)hipEventDestroy(b);
    if (e != hipSuccess)
        return fail(SHRAY_ERR_DEVICE, "vector-cache probe failed: %s", hipGetErrorString(e));
    *seconds = (double)ms * 1e-3;
    *bytes_loaded = (uint64_t)waves * (uint64_t)__builtin_popcountll(lane_mask) * (uint64_t)visits_per_lane * (uint64_t)bytes_per_lane;
    return SHRAY_OK;
}

}   // extern "C"
