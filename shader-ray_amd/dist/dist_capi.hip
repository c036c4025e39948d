// dist_capi.hip -- implementation of include/shader_ray_dist.h (libshray_dist.so): one rank's step of the
// multi-GPU frame loop in C++.  A client of libshray_hip's public C ABI (the tile render, the de-interleave) plus
// one pack kernel and the transports: RCCL (grouped ncclSend / ncclRecv over xGMI), an in-process loopback hub
// (ranks = threads sharing one device: the one-GPU rehearsal) and caller-supplied callbacks.
//
//   hip_stream :  render (one launch, `count` frames) -> pack (RGBA -> wire format; a rank's own tiles go straight
//                 into its gather buffer) -> [event] ............................ [wait] -> de-interleave
//   comm_stream:                                      [wait] -> grouped send / recv -> [event]
//
// The exchange runs on the object's own stream so that the next step's render (another hip_stream, another buffer
// set) overlaps it; all collectives of a rank are issued in step order on that one stream.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <chrono>
#include <condition_variable>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <deque>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

#include "plan.h"
#include "shader_ray_dist.h"

namespace plan = shray_dist_plan_detail;

namespace {

thread_local std::string g_error;

int fail(int code, const char *fmt, ...)
{
    char buf[640];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_error = buf;
    return code;
}

#define HIP_TRY(expr)                                                                              \
    do {                                                                                           \
        hipError_t e_ = (expr);                                                                    \
        if (e_ != hipSuccess)                                                                      \
            return fail(e_ == hipErrorOutOfMemory ? SHRAY_ERR_OUT_OF_MEMORY : SHRAY_ERR_DEVICE,    \
                        "%s failed: %s", #expr, hipGetErrorString(e_));                            \
    } while (0)
#define NCCL_TRY(expr)                                                                             \
    do {                                                                                           \
        ncclResult_t r_ = (expr);                                                                  \
        if (r_ != ncclSuccess)                                                                     \
            return fail(SHRAY_ERR_DEVICE, "%s failed: %s", #expr, ncclGetErrorString(r_));         \
    } while (0)
// a call into libshray_hip: its message is the one to report
#define SHRAY_TRY(expr)                                                                            \
    do {                                                                                           \
        int c_ = (expr);                                                                           \
        if (c_ != SHRAY_OK)                                                                        \
            return fail(c_, "%s", shray_last_error());                                             \
    } while (0)

// ---- pack: RGBA as rendered -> the wire format, frame by frame --------------------------------------------------
// Frame f of the step goes to wire + f * wire_stride, unless this rank assembles it: then its pixels go straight
// into the rank's own row of its gather buffer (slot = the frame's position among the frames it assembles).
template <int C>
__global__ void __launch_bounds__(256) pack_tiles_kernel(const float4 *__restrict__ rendered, size_t render_stride /* float4 */,
                                                         float *__restrict__ wire, size_t wire_stride /* floats */,
                                                         float *__restrict__ own_row, size_t gather_stride /* floats */,
                                                         unsigned int pixels, int rank, int world, int rotate)
{
    const unsigned int i = blockIdx.x * 256u + threadIdx.x;
    const int f = (int)blockIdx.y;
    if (i >= pixels)
        return;
    const bool mine = rotate ? (f % world == rank) : (rank == 0);
    float *dst = mine ? own_row + (size_t)(rotate ? f / world : f) * gather_stride : wire + (size_t)f * wire_stride;
    const float4 v = rendered[(size_t)f * render_stride + i];
    dst += (size_t)i * C;
    dst[0] = v.x;
    dst[1] = v.y;
    dst[2] = v.z;
    if (C == 4)
        dst[3] = v.w;
}

// ---- transports -------------------------------------------------------------------------------------------------
struct Transport {
    virtual ~Transport() {}
    // stream-ordered exchange on `stream`; returns SHRAY_OK or fails with g_error set
    virtual int exchange(char *d_wire, const shray_dist_xfer *sends, int ns, char *d_gather, const shray_dist_xfer *recvs, int nr,
                         hipStream_t stream) = 0;
    virtual bool on_comm_stream() const { return true; }
    // ranks the transport's own communicator reports (RCCL: ncclCommCount), or 0 where there is no such thing
    virtual int communicator_ranks(int *ranks) const
    {
        *ranks = 0;
        return SHRAY_OK;
    }
};

struct RcclTransport : Transport {
    ncclComm_t comm = nullptr;
    int communicator_ranks(int *ranks) const override
    {
        NCCL_TRY(ncclCommCount(comm, ranks));
        return SHRAY_OK;
    }
    ~RcclTransport() override
    {
        if (comm)
            (void)ncclCommDestroy(comm);
    }
    int exchange(char *d_wire, const shray_dist_xfer *sends, int ns, char *d_gather, const shray_dist_xfer *recvs, int nr,
                 hipStream_t stream) override
    {
        if (ns + nr == 0)
            return SHRAY_OK;
        // one group: every peer's transfers progress together, each over its own xGMI link
        NCCL_TRY(ncclGroupStart());
        for (int k = 0; k < nr; k++)
            NCCL_TRY(ncclRecv(d_gather + recvs[k].offset_bytes, (size_t)recvs[k].bytes, ncclUint8, recvs[k].peer, comm, stream));
        for (int k = 0; k < ns; k++)
            NCCL_TRY(ncclSend(d_wire + sends[k].offset_bytes, (size_t)sends[k].bytes, ncclUint8, sends[k].peer, comm, stream));
        NCCL_TRY(ncclGroupEnd());
        return SHRAY_OK;
    }
};

struct CallbackTransport : Transport {
    shray_dist_callbacks cb{};
    int exchange(char *d_wire, const shray_dist_xfer *sends, int ns, char *d_gather, const shray_dist_xfer *recvs, int nr,
                 hipStream_t stream) override
    {
        if (ns + nr == 0)
            return SHRAY_OK;
        const int rc = cb.exchange(cb.user, d_wire, sends, ns, d_gather, recvs, nr, (void *)stream);
        return rc == 0 ? SHRAY_OK : fail(SHRAY_ERR_DEVICE, "the exchange callback failed with code %d", rc);
    }
    bool on_comm_stream() const override { return false; }   // the callee orders itself against the step's stream
};

}   // namespace

// The loopback hub: for every ordered pair of ranks a queue of messages, each staged in a device buffer of the
// hub's own.  A send copies into a free slot and records an event; the matching receive (same position in the
// pair's order) waits for that event on its own stream, copies out and records another, after which the slot is
// free again.  Sends never block on the host, so ranks may issue all their sends before any receive.
struct shray_dist_hub {
    struct Slot {
        void *staging = nullptr;
        size_t capacity = 0, bytes = 0;
        hipEvent_t ready = nullptr, consumed = nullptr;
        bool busy = false, used = false;
    };
    struct Mailbox {
        std::mutex m;
        std::condition_variable cv;
        std::vector<Slot> slots;
        std::deque<int> pending;
    };
    int world;
    int device = -1;
    std::mutex device_mutex;
    std::vector<std::unique_ptr<Mailbox>> boxes;
    explicit shray_dist_hub(int w) : world(w)
    {
        for (int k = 0; k < w * w; k++)
            boxes.emplace_back(new Mailbox);
    }
    Mailbox &box(int src, int dst) { return *boxes[(size_t)src * world + dst]; }
    ~shray_dist_hub()
    {
        if (device >= 0)
            (void)hipSetDevice(device);
        for (auto &b : boxes)
            for (Slot &s : b->slots) {
                if (s.staging)
                    (void)hipFree(s.staging);
                if (s.ready)
                    (void)hipEventDestroy(s.ready);
                if (s.consumed)
                    (void)hipEventDestroy(s.consumed);
            }
    }
};

namespace {

struct LoopbackTransport : Transport {
    shray_dist_hub *hub = nullptr;
    int rank = 0;
    int send_one(const char *src, size_t bytes, int peer, hipStream_t stream)
    {
        shray_dist_hub::Mailbox &mb = hub->box(rank, peer);
        std::lock_guard<std::mutex> lock(mb.m);
        int which = -1;
        for (size_t k = 0; k < mb.slots.size() && which < 0; k++)
            if (!mb.slots[k].busy)
                which = (int)k;
        if (which < 0) {
            mb.slots.emplace_back();
            which = (int)mb.slots.size() - 1;
            HIP_TRY(hipEventCreateWithFlags(&mb.slots[which].ready, hipEventDisableTiming));
            HIP_TRY(hipEventCreateWithFlags(&mb.slots[which].consumed, hipEventDisableTiming));
        }
        shray_dist_hub::Slot &s = mb.slots[which];
        if (s.capacity < bytes) {
            if (s.used)
                HIP_TRY(hipEventSynchronize(s.consumed));   // the last reader of the old buffer has finished
            if (s.staging)
                HIP_TRY(hipFree(s.staging));
            s.staging = nullptr;
            s.capacity = 0;
            HIP_TRY(hipMalloc(&s.staging, bytes));
            s.capacity = bytes;
        } else if (s.used) {
            HIP_TRY(hipStreamWaitEvent(stream, s.consumed, 0));
        }
        HIP_TRY(hipMemcpyAsync(s.staging, src, bytes, hipMemcpyDeviceToDevice, stream));
        HIP_TRY(hipEventRecord(s.ready, stream));
        s.bytes = bytes;
        s.busy = true;
        s.used = true;
        mb.pending.push_back(which);
        mb.cv.notify_all();
        return SHRAY_OK;
    }
    int recv_one(char *dst, size_t bytes, int peer, hipStream_t stream)
    {
        shray_dist_hub::Mailbox &mb = hub->box(peer, rank);
        std::unique_lock<std::mutex> lock(mb.m);
        // a step is collective: the peer's send is on its way unless that rank failed or was never started
        if (!mb.cv.wait_for(lock, std::chrono::seconds(60), [&] { return !mb.pending.empty(); }))
            return fail(SHRAY_ERR_DEVICE, "loopback: rank %d waited 60 s for a transfer from rank %d (did that rank's step fail, or was it "
                        "called with another frame count?)", rank, peer);
        const int which = mb.pending.front();
        mb.pending.pop_front();
        shray_dist_hub::Slot &s = mb.slots[which];
        if (s.bytes != bytes)
            return fail(SHRAY_ERR_DEVICE, "loopback: rank %d expects %zu bytes from rank %d, which sent %zu", rank, bytes, peer, s.bytes);
        HIP_TRY(hipStreamWaitEvent(stream, s.ready, 0));
        HIP_TRY(hipMemcpyAsync(dst, s.staging, bytes, hipMemcpyDeviceToDevice, stream));
        HIP_TRY(hipEventRecord(s.consumed, stream));
        s.busy = false;
        return SHRAY_OK;
    }
    int exchange(char *d_wire, const shray_dist_xfer *sends, int ns, char *d_gather, const shray_dist_xfer *recvs, int nr,
                 hipStream_t stream) override
    {
        for (int k = 0; k < ns; k++) {
            const int rc = send_one(d_wire + sends[k].offset_bytes, (size_t)sends[k].bytes, sends[k].peer, stream);
            if (rc)
                return rc;
        }
        for (int k = 0; k < nr; k++) {
            const int rc = recv_one(d_gather + recvs[k].offset_bytes, (size_t)recvs[k].bytes, recvs[k].peer, stream);
            if (rc)
                return rc;
        }
        return SHRAY_OK;
    }
};

struct BufferSet {
    void *rendered = nullptr, *wire = nullptr, *gather = nullptr, *output = nullptr;
    hipEvent_t packed = nullptr, exchanged = nullptr;
    hipEvent_t finished = nullptr;      // recorded at the end of the last step that used the set, on that step's stream
    hipEvent_t stamp[4] = {nullptr, nullptr, nullptr, nullptr};   // shray_dist_set_timing: start, packed, exchanged, finished
    bool used = false, stamped = false;                            // stamped: the set's most recent step recorded its stamps
};

}   // namespace

struct shray_dist {
    plan::Resolved r{};
    shray_dist_plan plan{};
    shray_scene *scene = nullptr;
    int spp = 1;
    int device = 0;
    std::unique_ptr<Transport> transport;
    hipStream_t comm_stream = nullptr;
    std::vector<BufferSet> sets;
    bool timing = false;
    ~shray_dist()
    {
        (void)hipSetDevice(device);
        if (comm_stream) {
            (void)hipStreamSynchronize(comm_stream);
            (void)hipStreamDestroy(comm_stream);
        }
        for (BufferSet &b : sets) {
            for (void *p : {b.rendered, b.wire, b.gather, b.output})
                if (p)
                    (void)hipFree(p);
            for (hipEvent_t e : b.stamp)
                if (e)
                    (void)hipEventDestroy(e);
            if (b.packed)
                (void)hipEventDestroy(b.packed);
            if (b.exchanged)
                (void)hipEventDestroy(b.exchanged);
            if (b.finished)
                (void)hipEventDestroy(b.finished);
        }
    }
};

extern "C" {

const char *shray_dist_last_error(void) { return g_error.c_str(); }

int shray_dist_balanced_shares(int world, double overhead, int *rank0_phases, int *other_phases)
{
    if (!rank0_phases || !other_phases || world < 1 || world > SHRAY_DIST_MAX_WORLD)
        return fail(SHRAY_ERR_INVALID_ARGUMENT, "shray_dist_balanced_shares: bad arguments (world %d)", world);
    plan::balanced_shares(world, overhead, rank0_phases, other_phases);
    return SHRAY_OK;
}

int shray_dist_make_plan(const shray_dist_config *config, shray_dist_plan *out)
{
    plan::Resolved r;
    if (const char *why = plan::resolve(config, &r))
        return fail(SHRAY_ERR_INVALID_ARGUMENT, "shray_dist_config: %s", why);
    if (!out)
        return fail(SHRAY_ERR_INVALID_ARGUMENT, "plan is NULL");
    plan::make_plan(r, out);
    return SHRAY_OK;
}

int shray_dist_frame_owner(const shray_dist_config *config, int frame)
{
    plan::Resolved r;
    if (const char *why = plan::resolve(config, &r))
        return fail(SHRAY_ERR_INVALID_ARGUMENT, "shray_dist_config: %s", why);
    if (frame < 0)
        return fail(SHRAY_ERR_INVALID_ARGUMENT, "frame %d", frame);
    return plan::frame_owner(r, frame);
}

int shray_dist_step_xfers(const shray_dist_config *config, int count, shray_dist_xfer *sends, int *send_count,
                          shray_dist_xfer *recvs, int *recv_count, int *assembled, int *first_frame, int *frame_step)
{
    plan::Resolved r;
    if (const char *why = plan::resolve(config, &r))
        return fail(SHRAY_ERR_INVALID_ARGUMENT, "shray_dist_config: %s", why);
    if (!sends || !send_count || !recvs || !recv_count || count < 1 || count > r.max_frames)
        return fail(SHRAY_ERR_INVALID_ARGUMENT, "shray_dist_step_xfers: NULL list or count %d outside 1..%d", count, r.max_frames);
    plan::step_xfers(r, count, sends, send_count, recvs, recv_count);
    if (assembled)
        *assembled = plan::assembled_frames(r, r.rank, count);
    if (first_frame)
        *first_frame = r.root_mode == SHRAY_DIST_ROTATE ? r.rank : 0;
    if (frame_step)
        *frame_step = r.root_mode == SHRAY_DIST_ROTATE ? r.world : 1;
    return SHRAY_OK;
}

int shray_dist_unique_id(void *id_out)
{
    static_assert(sizeof(ncclUniqueId) == SHRAY_DIST_UNIQUE_ID_BYTES, "ncclUniqueId is 128 bytes");
    if (!id_out)
        return fail(SHRAY_ERR_INVALID_ARGUMENT, "id_out is NULL");
    ncclUniqueId id;
    NCCL_TRY(ncclGetUniqueId(&id));
    memcpy(id_out, &id, sizeof(id));
    return SHRAY_OK;
}

int shray_dist_hub_create(int world, shray_dist_hub **out_hub)
{
    if (!out_hub || world < 1 || world > SHRAY_DIST_MAX_WORLD)
        return fail(SHRAY_ERR_INVALID_ARGUMENT, "shray_dist_hub_create: bad arguments (world %d)", world);
    *out_hub = new shray_dist_hub(world);
    return SHRAY_OK;
}

int shray_dist_hub_destroy(shray_dist_hub *hub)
{
    delete hub;
    return SHRAY_OK;
}

int shray_dist_create(shray_scene *scene, const shray_dist_config *config, const void *transport_arg, shray_dist **out_dist)
{
    if (!out_dist)
        return fail(SHRAY_ERR_INVALID_ARGUMENT, "out_dist is NULL");
    *out_dist = nullptr;
    plan::Resolved r;
    if (const char *why = plan::resolve(config, &r))
        return fail(SHRAY_ERR_INVALID_ARGUMENT, "shray_dist_config: %s", why);
    if (!scene)
        return fail(SHRAY_ERR_INVALID_ARGUMENT, "scene is NULL");
    if (!transport_arg)
        return fail(SHRAY_ERR_INVALID_ARGUMENT, "transport_arg is NULL (RCCL: the unique id; LOOPBACK: the hub; CALLBACK: the callbacks)");

    std::unique_ptr<shray_dist> d(new shray_dist);
    d->r = r;
    plan::make_plan(r, &d->plan);
    d->scene = scene;
    d->spp = config->spp;
    SHRAY_TRY(shray_scene_device(scene, &d->device));   // the scene's device is this rank's device
    HIP_TRY(hipSetDevice(d->device));

    if (config->transport == SHRAY_DIST_RCCL) {
        std::unique_ptr<RcclTransport> t(new RcclTransport);
        ncclUniqueId id;
        memcpy(&id, transport_arg, sizeof(id));
        // one line per rank on stderr in front of and behind the collective start-up: a run that hangs in it says where
        // (the first contact of several GPUs has never been observed: VERDICT round 4)
        const auto t0 = std::chrono::steady_clock::now();
        fprintf(stderr, "shray_dist_create: rank %d of %d on device %d: ncclCommInitRank ...\n", r.rank, r.world, d->device);
        fflush(stderr);
        NCCL_TRY(ncclCommInitRank(&t->comm, r.world, id, r.rank));
        fprintf(stderr, "shray_dist_create: rank %d of %d: communicator up after %.3f s\n", r.rank, r.world,
                std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count());
        fflush(stderr);
        d->transport = std::move(t);
    } else if (config->transport == SHRAY_DIST_LOOPBACK) {
        shray_dist_hub *hub = (shray_dist_hub *)transport_arg;
        if (hub->world != r.world)
            return fail(SHRAY_ERR_INVALID_ARGUMENT, "the hub was made for %d ranks, the configuration has %d", hub->world, r.world);
        {
            std::lock_guard<std::mutex> lock(hub->device_mutex);
            if (hub->device < 0)
                hub->device = d->device;
            if (hub->device != d->device)
                return fail(SHRAY_ERR_INVALID_ARGUMENT, "loopback ranks share one device: rank %d is on device %d, the hub on %d",
                            r.rank, d->device, hub->device);
        }
        std::unique_ptr<LoopbackTransport> t(new LoopbackTransport);
        t->hub = hub;
        t->rank = r.rank;
        d->transport = std::move(t);
    } else {
        const shray_dist_callbacks *cb = (const shray_dist_callbacks *)transport_arg;
        if (!cb->exchange)
            return fail(SHRAY_ERR_INVALID_ARGUMENT, "shray_dist_callbacks.exchange is NULL");
        std::unique_ptr<CallbackTransport> t(new CallbackTransport);
        t->cb = *cb;
        d->transport = std::move(t);
    }

    HIP_TRY(hipStreamCreateWithFlags(&d->comm_stream, hipStreamNonBlocking));
    d->sets.resize((size_t)r.buffer_sets);
    const shray_dist_plan &p = d->plan;
    const size_t frame_bytes = (size_t)r.width * r.height * 16;
    for (BufferSet &b : d->sets) {
        // plain allocations: every byte that is read later is written first by this step's own kernels or transfers
        // (a rank's wire / gather rows hold exactly its owned tiles; the de-interleave reads only those)
        if (r.world == 1) {
            // a lone rank renders whole frames straight into `output` (shray_dist_step): nothing to pack or gather
            HIP_TRY(hipMalloc(&b.output, frame_bytes * (size_t)r.max_frames));
            HIP_TRY(hipEventCreateWithFlags(&b.packed, hipEventDisableTiming));
            HIP_TRY(hipEventCreateWithFlags(&b.exchanged, hipEventDisableTiming));
            HIP_TRY(hipEventCreateWithFlags(&b.finished, hipEventDisableTiming));
            continue;
        }
        HIP_TRY(hipMalloc(&b.rendered, std::max<size_t>(16, (size_t)p.render_frame_stride_bytes * r.max_frames)));
        HIP_TRY(hipMalloc(&b.wire, std::max<size_t>(16, (size_t)p.wire_frame_stride_bytes * r.max_frames)));
        if (p.max_assembled > 0) {
            HIP_TRY(hipMalloc(&b.gather, std::max<size_t>(16, (size_t)p.gather_rank_stride_bytes * r.world)));
            HIP_TRY(hipMalloc(&b.output, frame_bytes * (size_t)p.max_assembled));
        }
        HIP_TRY(hipEventCreateWithFlags(&b.packed, hipEventDisableTiming));
        HIP_TRY(hipEventCreateWithFlags(&b.exchanged, hipEventDisableTiming));
        HIP_TRY(hipEventCreateWithFlags(&b.finished, hipEventDisableTiming));
    }
    *out_dist = d.release();
    return SHRAY_OK;
}

int shray_dist_destroy(shray_dist *dist)
{
    delete dist;
    return SHRAY_OK;
}

int shray_dist_step(shray_dist *dist, int buffer_set, const shray_frame_params *params, int count, void *hip_stream)
{
    if (!dist || !params)
        return fail(SHRAY_ERR_INVALID_ARGUMENT, "dist or params is NULL");
    const plan::Resolved &r = dist->r;
    const shray_dist_plan &p = dist->plan;
    if (buffer_set < 0 || buffer_set >= (int)dist->sets.size() || count < 1 || count > r.max_frames)
        return fail(SHRAY_ERR_INVALID_ARGUMENT, "buffer set %d of %zu, %d frames of at most %d", buffer_set, dist->sets.size(), count,
                    r.max_frames);
    HIP_TRY(hipSetDevice(dist->device));
    BufferSet &b = dist->sets[(size_t)buffer_set];
    hipStream_t stream = (hipStream_t)hip_stream;
    const bool rotate = r.root_mode == SHRAY_DIST_ROTATE;

    // A buffer set is reused only when the step that used it last is over -- whichever stream that step ran on (a caller
    // with more streams than sets, or one that pairs them differently from step to step, is ordered here; with the same
    // stream as last time the wait is a no-op).  `finished` sits behind that step's de-interleave, which itself waited
    // for its exchange: neither its sends from `wire` nor its receives into `gather` can still be in flight.
    if (b.used)
        HIP_TRY(hipStreamWaitEvent(stream, b.finished, 0));
    b.used = true;
    // shray_dist_set_timing: four stamps per step (timing events, made on first use)
    const bool stamps = dist->timing;
    b.stamped = stamps;
    if (stamps) {
        for (hipEvent_t &e : b.stamp)
            if (!e)
                HIP_TRY(hipEventCreate(&e));
        HIP_TRY(hipEventRecord(b.stamp[0], stream));
    }

    if (r.world == 1) {
        // a lone rank owns every tile and assembles every frame (in both root modes): whole frames, row-major, straight
        // into `output` -- the single-GPU path, no pack, no exchange, no de-interleave
        SHRAY_TRY(shray_render_batch_device(dist->scene, params, count, r.width, r.height, dist->spp, nullptr, b.output,
                                            (int64_t)r.width * r.height * 16, stream));
        if (stamps)
            for (int k = 1; k < 4; k++)
                HIP_TRY(hipEventRecord(b.stamp[k], stream));
        HIP_TRY(hipEventRecord(b.finished, stream));
        return SHRAY_OK;
    }

    // 1. this rank's tiles of all `count` frames, one launch
    if (p.owned_tiles > 0) {
        SHRAY_TRY(shray_render_batch_device(dist->scene, params, count, r.width, r.height, dist->spp, &p.tiles, b.rendered,
                                            p.render_frame_stride_bytes, stream));
        // 2. pack
        const unsigned int pixels = (unsigned int)(p.owned_tiles * r.tile_w * r.tile_h);
        const dim3 grid((pixels + 255u) / 256u, (unsigned)count), block(256);
        float *own_row = b.gather ? (float *)((char *)b.gather + (size_t)r.rank * p.gather_rank_stride_bytes) : nullptr;
        if (p.channels == 3)
            hipLaunchKernelGGL((pack_tiles_kernel<3>), grid, block, 0, stream, (const float4 *)b.rendered,
                               (size_t)p.render_frame_stride_bytes / 16, (float *)b.wire, (size_t)p.wire_frame_stride_bytes / 4, own_row,
                               (size_t)p.gather_frame_stride_bytes / 4, pixels, r.rank, r.world, rotate ? 1 : 0);
        else
            hipLaunchKernelGGL((pack_tiles_kernel<4>), grid, block, 0, stream, (const float4 *)b.rendered,
                               (size_t)p.render_frame_stride_bytes / 16, (float *)b.wire, (size_t)p.wire_frame_stride_bytes / 4, own_row,
                               (size_t)p.gather_frame_stride_bytes / 4, pixels, r.rank, r.world, rotate ? 1 : 0);
        HIP_TRY(hipGetLastError());
    }

    if (stamps)
        HIP_TRY(hipEventRecord(b.stamp[1], stream));

    // 3. exchange
    // a rank sends at most one transfer per frame and receives (frames it assembles) x (world - 1) <= count + world
    shray_dist_xfer sends[SHRAY_MAX_BATCH + SHRAY_DIST_MAX_WORLD], recvs[SHRAY_MAX_BATCH + SHRAY_DIST_MAX_WORLD];
    int ns = 0, nr = 0;
    plan::step_xfers(r, count, sends, &ns, recvs, &nr);
    if (ns + nr > 0) {
        if (dist->transport->on_comm_stream()) {
            HIP_TRY(hipEventRecord(b.packed, stream));
            HIP_TRY(hipStreamWaitEvent(dist->comm_stream, b.packed, 0));
            const int rc = dist->transport->exchange((char *)b.wire, sends, ns, (char *)b.gather, recvs, nr, dist->comm_stream);
            if (rc)
                return rc;
            HIP_TRY(hipEventRecord(b.exchanged, dist->comm_stream));
            HIP_TRY(hipStreamWaitEvent(stream, b.exchanged, 0));
        } else {
            const int rc = dist->transport->exchange((char *)b.wire, sends, ns, (char *)b.gather, recvs, nr, stream);
            if (rc)
                return rc;
        }
    }
    if (stamps)
        HIP_TRY(hipEventRecord(b.stamp[2], stream));   // (behind the wait for the exchange: when hip_stream may go on)

    // 4. de-interleave the frames this rank assembles
    const int assembled = plan::assembled_frames(r, r.rank, count);
    if (assembled > 0)
        SHRAY_TRY(shray_assemble_tiles_split_device(b.gather, r.world, r.c0, r.c1, assembled, p.channels, p.gather_rank_stride_bytes,
                                                    p.gather_frame_stride_bytes, r.width, r.height, r.tile_w, r.tile_h, b.output, stream));
    if (stamps)
        HIP_TRY(hipEventRecord(b.stamp[3], stream));
    HIP_TRY(hipEventRecord(b.finished, stream));
    return SHRAY_OK;
}

int shray_dist_set_timing(shray_dist *dist, int enable)
{
    if (!dist)
        return fail(SHRAY_ERR_INVALID_ARGUMENT, "dist is NULL");
    dist->timing = enable != 0;
    return SHRAY_OK;
}

int shray_dist_step_times(shray_dist *dist, int buffer_set, float *render_ms, float *exchange_ms, float *assemble_ms)
{
    if (!dist || buffer_set < 0 || buffer_set >= (int)dist->sets.size())
        return fail(SHRAY_ERR_INVALID_ARGUMENT, "shray_dist_step_times: bad arguments");
    BufferSet &b = dist->sets[(size_t)buffer_set];
    if (!b.used || !b.stamped)
        return fail(SHRAY_ERR_INVALID_ARGUMENT, "buffer set %d: its most recent step recorded no stamps (shray_dist_set_timing)", buffer_set);
    HIP_TRY(hipSetDevice(dist->device));
    HIP_TRY(hipEventSynchronize(b.stamp[3]));
    float *out[3] = {render_ms, exchange_ms, assemble_ms};
    for (int k = 0; k < 3; k++) {
        float ms = 0.0f;
        HIP_TRY(hipEventElapsedTime(&ms, b.stamp[k], b.stamp[k + 1]));
        if (out[k])
            *out[k] = ms;
    }
    return SHRAY_OK;
}

int shray_dist_world(shray_dist *dist, int *world, int *communicator_ranks)
{
    if (!dist)
        return fail(SHRAY_ERR_INVALID_ARGUMENT, "dist is NULL");
    if (world)
        *world = dist->r.world;
    if (communicator_ranks) {
        HIP_TRY(hipSetDevice(dist->device));
        return dist->transport->communicator_ranks(communicator_ranks);
    }
    return SHRAY_OK;
}

int shray_dist_output(shray_dist *dist, int buffer_set, int count, int *assembled, int *first_frame, int *frame_step, void **d_rgba)
{
    if (!dist || buffer_set < 0 || buffer_set >= (int)dist->sets.size() || count < 1 || count > dist->r.max_frames)
        return fail(SHRAY_ERR_INVALID_ARGUMENT, "shray_dist_output: bad arguments");
    const plan::Resolved &r = dist->r;
    if (assembled)
        *assembled = plan::assembled_frames(r, r.rank, count);
    if (first_frame)
        *first_frame = r.root_mode == SHRAY_DIST_ROTATE ? r.rank : 0;
    if (frame_step)
        *frame_step = r.root_mode == SHRAY_DIST_ROTATE ? r.world : 1;
    if (d_rgba)
        *d_rgba = dist->sets[(size_t)buffer_set].output;
    return SHRAY_OK;
}

int shray_dist_copy_output(shray_dist *dist, int buffer_set, int count, void *d_dst, void *hip_stream)
{
    if (!dist || !d_dst || buffer_set < 0 || buffer_set >= (int)dist->sets.size() || count < 1 || count > dist->r.max_frames)
        return fail(SHRAY_ERR_INVALID_ARGUMENT, "shray_dist_copy_output: bad arguments");
    const int assembled = plan::assembled_frames(dist->r, dist->r.rank, count);
    if (assembled == 0)
        return SHRAY_OK;
    HIP_TRY(hipSetDevice(dist->device));
    BufferSet &b = dist->sets[(size_t)buffer_set];
    hipStream_t stream = (hipStream_t)hip_stream;
    // The copy reads the set's output: it follows the step that wrote it whichever stream that ran on, and the set's NEXT step
    // -- on any stream -- must follow the copy (a lone rank renders straight into `output`, the others de-interleave into
    // it): `finished` moves behind the copy, so shray_dist_step's wait covers it.
    if (b.used)
        HIP_TRY(hipStreamWaitEvent(stream, b.finished, 0));
    HIP_TRY(hipMemcpyAsync(d_dst, b.output, (size_t)assembled * dist->r.width * dist->r.height * 16, hipMemcpyDeviceToDevice, stream));
    if (b.used)
        HIP_TRY(hipEventRecord(b.finished, stream));
    return SHRAY_OK;
}

int shray_dist_copy_to_host(void *host_dst, const void *d_src, int64_t bytes, void *hip_stream)
{
    if (!host_dst || !d_src || bytes < 0)
        return fail(SHRAY_ERR_INVALID_ARGUMENT, "shray_dist_copy_to_host: bad arguments");
    HIP_TRY(hipStreamSynchronize((hipStream_t)hip_stream));
    HIP_TRY(hipMemcpy(host_dst, d_src, (size_t)bytes, hipMemcpyDeviceToHost));
    return SHRAY_OK;
}

int shray_dist_copy_to_device(void *d_dst, const void *host_src, int64_t bytes, void *hip_stream)
{
    if (!d_dst || !host_src || bytes < 0)
        return fail(SHRAY_ERR_INVALID_ARGUMENT, "shray_dist_copy_to_device: bad arguments");
    HIP_TRY(hipStreamSynchronize((hipStream_t)hip_stream));
    HIP_TRY(hipMemcpy(d_dst, host_src, (size_t)bytes, hipMemcpyHostToDevice));
    return SHRAY_OK;
}

}   // extern "C"
