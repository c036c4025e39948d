/*
 * shader_ray_dist.h -- C ABI of the multi-GPU frame loop (libshray_dist.so).
 *
 * The reference renders on one GPU into one framebuffer: DrawFrame (ray.cpp:591-717) called from
 * the frame loop of main (ray.cpp:954, :1096-1131).  This is that loop's body for N GPUs of one
 * node, one rank per GPU (a process, or a thread of one process): every rank holds a replica of
 * the scene (shray_scene, shader_ray_hip.h), renders its interleaved tiles of `count` consecutive
 * frames in one launch, and the packed tile buffers travel over xGMI with grouped
 * ncclSend / ncclRecv (RCCL) to the rank that assembles the frame.  Pixels are independent
 * (raytracer.es.fs:613-682 reads no neighbour) and all samples of a pixel stay on its GPU, so
 * there is no reduction -- one exchange step per `count` frames.
 *
 *   shray_dist_step      <- one turn of the frame loop, ray.cpp:1096-1131, for `count` frames:
 *                           uniform blocks + draws (ray.cpp:648-707) on every rank, then the
 *                           exchange and the de-interleave that stand where the reference has a
 *                           single framebuffer
 *   shray_dist_create    <- (no upstream counterpart: one GPU, one GL context, ray.cpp:954-1000)
 *
 * Root modes (where a frame is assembled):
 *   SHRAY_DIST_ROOT0   every frame on rank 0 (a gather: 7 peers send on their 7 links into rank 0).
 *                      Rank 0 also receives and de-interleaves, so it may own a smaller share of the
 *                      tiles (rank0_phases / other_phases, as in shray_tile_set).
 *   SHRAY_DIST_ROTATE  frame f of a step on rank f % world (an all-to-all of tile buffers: all
 *                      world * (world - 1) directed links carry pixels, each 1 / world of a frame per
 *                      frame; equal shares).  A step of `world` frames leaves one assembled frame on
 *                      every rank.
 *
 * Plain C, plain pointers and sizes.  Errors: SHRAY_OK or a negative shader_ray_hip.h code; the
 * message is shray_dist_last_error()'s (calling thread).  Threading: one shray_dist per rank; calls
 * on one object are not re-entrant; the objects of different ranks are driven concurrently (they
 * must be: a step is collective).
 */
#ifndef SHADER_RAY_DIST_H
#define SHADER_RAY_DIST_H

#include "shader_ray_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

enum { SHRAY_DIST_ROOT0 = 0, SHRAY_DIST_ROTATE = 1 };

enum {
    SHRAY_DIST_RCCL = 0,       /* ncclSend / ncclRecv on a communicator this library creates (one rank per GPU) */
    SHRAY_DIST_LOOPBACK = 1,   /* ranks are threads of ONE process sharing ONE device: device-to-device copies through
                                  an in-process hub.  Rehearsal of the whole step on a one-GPU box (RCCL refuses two
                                  ranks on one device). */
    SHRAY_DIST_CALLBACK = 2    /* the caller moves the bytes (shray_dist_callbacks): MPI, a gloo rehearsal, ... */
};

#define SHRAY_DIST_UNIQUE_ID_BYTES 128   /* sizeof(ncclUniqueId) */
#define SHRAY_DIST_MAX_WORLD 64

typedef struct shray_dist_config {
    uint32_t struct_size;        /* sizeof(shray_dist_config) */
    int32_t rank, world;
    int32_t width, height, spp;  /* the frame every step renders */
    int32_t tile_w, tile_h;      /* multiples of 16; 0 = 32 */
    int32_t max_frames;          /* most frames one step carries, 1..SHRAY_MAX_BATCH */
    int32_t root_mode;           /* SHRAY_DIST_ROOT0 / SHRAY_DIST_ROTATE */
    int32_t rank0_phases;        /* ROOT0 shares (tile phases per period owned by rank 0 / by every other rank);  */
    int32_t other_phases;        /*   0, 0 = shray_dist_balanced_shares(world); ROTATE always uses 1, 1          */
    int32_t rgb_wire;            /* 1: R, G, B travel (alpha is the constant 1, raytracer.es.fs:676): 12 B / pixel */
    int32_t transport;           /* SHRAY_DIST_RCCL / _LOOPBACK / _CALLBACK */
    int32_t buffer_sets;         /* independent sets of step buffers (1..4; 0 = 2): step k may use set k % buffer_sets
                                    while the exchange of step k - 1 is still in flight */
} shray_dist_config;

/* One transfer of a step: `bytes` at `offset_bytes` of this rank's wire buffer (sends) or gather buffer (receives),
 * to / from rank `peer`; `frame` = the frame of the step it belongs to, or -1 when it carries all of them. */
typedef struct shray_dist_xfer {
    int32_t peer;
    int32_t frame;
    int64_t offset_bytes;
    int64_t bytes;
} shray_dist_xfer;

/* Everything a rank's step is made of, computable without a GPU (tests, other hosts' bindings). */
typedef struct shray_dist_plan {
    uint32_t struct_size;             /* sizeof(shray_dist_plan) */
    shray_tile_set tiles;             /* what this rank renders (shray_render_batch_device) */
    int32_t rank0_phases, other_phases;
    int32_t channels;                 /* floats per pixel on the wire: 3 or 4 */
    int32_t max_assembled;            /* most frames this rank assembles in one step */
    int64_t owned_tiles;              /* tiles of one frame this rank renders */
    int64_t max_tiles;                /* the largest share among the ranks: every per-frame stride below is sized for it */
    int64_t render_frame_stride_bytes;/* RGBA as rendered: max_tiles * tile_w * tile_h * 16 */
    int64_t wire_frame_stride_bytes;  /* as sent:          max_tiles * tile_w * tile_h * channels * 4 */
    int64_t gather_rank_stride_bytes; /* gather buffer: [source rank][assembled-frame slot][wire frame] */
    int64_t gather_frame_stride_bytes;
} shray_dist_plan;

/* CALLBACK transport.  exchange() is called once per step on the calling thread, after the rank's render and
 * pack have been enqueued on hip_stream.  It must deliver every send (device memory d_wire + offset) to its peer
 * and complete every receive (into d_gather + offset) before work enqueued on hip_stream AFTER it returns runs
 * (the simplest conforming callee synchronises the stream and moves the bytes before returning).  Returns 0 or
 * non-zero (the step then fails with SHRAY_ERR_DEVICE). */
typedef struct shray_dist_callbacks {
    void *user;
    int (*exchange)(void *user, void *d_wire, const shray_dist_xfer *sends, int send_count,
                    void *d_gather, const shray_dist_xfer *recvs, int recv_count, void *hip_stream);
} shray_dist_callbacks;

typedef struct shray_dist shray_dist;
typedef struct shray_dist_hub shray_dist_hub;   /* the LOOPBACK transport's in-process meeting point */

const char *shray_dist_last_error(void);

/* ---- the plan: host-only, no GPU touched ------------------------------------------------------- */
/* (c0, c1) minimising the slowest rank's time max(c0 / period + overhead, c1 / period), period = c0 + (world - 1) c1,
 * c0 <= c1 <= 8: rank 0's extra work per frame (receive, de-interleave) as `overhead` of one GPU's frame time;
 * overhead < 0 = the measured default 0.06. */
int shray_dist_balanced_shares(int world, double overhead, int *rank0_phases, int *other_phases);
int shray_dist_make_plan(const shray_dist_config *config, shray_dist_plan *plan);
/* The rank that assembles frame `frame` (0-based within a step). */
int shray_dist_frame_owner(const shray_dist_config *config, int frame);
/* The transfers of a step of `count` frames, in the order they are issued (identical on every rank for a pair:
 * rank a's k-th send to b matches b's k-th receive from a).  sends / recvs hold SHRAY_MAX_BATCH + SHRAY_DIST_MAX_WORLD
 * entries at least.  *assembled = how many of the step's frames this rank assembles: frames
 * first_frame, first_frame + frame_step, ... land in gather slots 0, 1, ... */
int shray_dist_step_xfers(const shray_dist_config *config, int count, shray_dist_xfer *sends, int *send_count,
                          shray_dist_xfer *recvs, int *recv_count, int *assembled, int *first_frame, int *frame_step);

/* ---- the step ---------------------------------------------------------------------------------- */
/* RCCL: rank 0 makes an id and hands it to the others (any side channel: a file, MPI, torch.distributed's store). */
int shray_dist_unique_id(void *id_out /* SHRAY_DIST_UNIQUE_ID_BYTES */);
/* LOOPBACK: one hub per group of ranks, created once and passed to every rank's shray_dist_create. */
int shray_dist_hub_create(int world, shray_dist_hub **out_hub);
int shray_dist_hub_destroy(shray_dist_hub *hub);

/* Collective: every rank calls it (concurrently; the RCCL transport blocks until all have).
 * `scene` is this rank's replica on this rank's device, with its environment set; it must outlive the object.
 * transport_arg: RCCL -> the SHRAY_DIST_UNIQUE_ID_BYTES of the id; LOOPBACK -> the shray_dist_hub;
 * CALLBACK -> a shray_dist_callbacks (copied). */
int shray_dist_create(shray_scene *scene, const shray_dist_config *config, const void *transport_arg,
                      shray_dist **out_dist);
int shray_dist_destroy(shray_dist *dist);

/* *world = the configuration's world; *communicator_ranks = how many ranks the transport's own communicator reports
 * -- RCCL: ncclCommCount of the communicator shray_dist_create made (a measurement that claims N GPUs can show that RCCL
 * really joined N of them) --, 0 for the LOOPBACK and CALLBACK transports, which have no communicator.  Either may be NULL. */
int shray_dist_world(shray_dist *dist, int *world, int *communicator_ranks);

/* One step: frames 0..count-1 rendered with params[0..count-1] (this rank's tiles, one launch), packed, exchanged
 * and de-interleaved on the ranks that own them.  Everything is enqueued: the render, pack and de-interleave on
 * hip_stream, the exchange on the object's own communication stream (ordered against hip_stream by events), so two
 * steps on two streams and two buffer sets overlap -- the exchange of one runs under the render of the next.
 * Nothing synchronises with the host (the CALLBACK transport does whatever its callee does).  A buffer set may be driven
 * from any stream: a step first makes hip_stream wait (on the device) for the end of the set's previous step.
 * world == 1: the rank renders whole frames straight into the set's output (no pack, exchange or de-interleave). */
int shray_dist_step(shray_dist *dist, int buffer_set, const shray_frame_params *params, int count, void *hip_stream);

/* Where a step's time goes, for a run that wants to say so (bench.py --gpus N: a sub-linear scaling result then names the
 * stage -- the reference's loop has one stage, DrawFrame, ray.cpp:1096-1131).  enable != 0: every later step also records
 * four timing events -- its start, the end of render + pack (hip_stream), the end of the exchange (the communication stream),
 * the end of the de-interleave.  shray_dist_step_times waits for the set's most recent step and returns the three intervals
 * in milliseconds: render_ms = start -> packed, exchange_ms = packed -> exchanged (what the step waited for the links, net of
 * whatever other steps' work the GPU overlapped with it), assemble_ms = exchanged -> finished.  A lone rank (world 1) has
 * render_ms only.  Either pointer may be NULL.  SHRAY_ERR_INVALID_ARGUMENT if timing was off when that step ran. */
int shray_dist_set_timing(shray_dist *dist, int enable);
int shray_dist_step_times(shray_dist *dist, int buffer_set, float *render_ms, float *exchange_ms, float *assemble_ms);

/* After a step on `buffer_set` (and once hip_stream has reached that point): the frames this rank assembled.
 * *assembled frames, the k-th of them frame *first_frame + k * *frame_step of the step, RGBA float32 row 0 = bottom,
 * at *d_rgba + k * width * height * 16 (device memory owned by the object).  The set's next step overwrites that memory:
 * whoever reads through the raw pointer orders the reads before that step himself (same stream, or an event the step's
 * stream waits for); shray_dist_copy_output does it for its copy. */
int shray_dist_output(shray_dist *dist, int buffer_set, int count, int *assembled, int *first_frame, int *frame_step,
                      void **d_rgba);

/* Enqueues on hip_stream a copy of those assembled frames, back to back, into d_dst (device memory on this
 * rank's device, *assembled * width * height * 16 bytes) -- e.g. a buffer of the application's own.  hip_stream may be any
 * stream: the copy waits (on the device) for the step that wrote the frames, and the set's next step -- on whichever
 * stream -- waits for the copy. */
int shray_dist_copy_output(shray_dist *dist, int buffer_set, int count, void *d_dst, void *hip_stream);

/* For CALLBACK transports written where no HIP binding is at hand (the Python / gloo rehearsal): a blocking copy of
 * `bytes` between host memory and device memory of the calling thread's current device, after hip_stream has drained. */
int shray_dist_copy_to_host(void *host_dst, const void *d_src, int64_t bytes, void *hip_stream);
int shray_dist_copy_to_device(void *d_dst, const void *host_src, int64_t bytes, void *hip_stream);

#ifdef __cplusplus
}
#endif
#endif /* SHADER_RAY_DIST_H */
