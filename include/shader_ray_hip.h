/*
 * shader_ray_hip.h -- C ABI of the MI355X (gfx950) per-pixel tracer.
 *
 * This is the drop-in boundary for shader-ray's hot path.  In the reference
 * the path sits behind three OpenGL call groups; each entry point below
 * names the reference interface it replaces (file:line into the upstream
 * repository):
 *
 *   shray_scene_create      <- texture upload in load_scene_data, ray.cpp:470-497
 *                              (payload = scene_shader_data, world.h:68-93)
 *   shray_scene_set_environment
 *                           <- background texture upload, ray.cpp:499-510
 *                              (payload = float2Dimage, ray.cpp:330-343)
 *   shray_render / shray_render_device
 *                           <- uniform block + glDrawArrays in DrawFrame,
 *                              ray.cpp:648-707, and the glReadPixels readback
 *                              in screenshot, ray.cpp:760
 *   shray_render_batch_device
 *                           <- the frame loop around DrawFrame (benchmark key,
 *                              ray.cpp:1096-1131): several frames per launch
 *   shray_scene_destroy     <- (GL objects are never freed upstream)
 *
 * Plain C, plain pointers and sizes, no C++/torch types.  All matrices are
 * column-major float[16] exactly as the reference hands them to
 * glUniformMatrix4fv(..., GL_FALSE, ...) (vectormath.h:258-272).
 *
 * Ownership: shray_scene_create / shray_scene_set_environment copy what they
 * need to the device; the caller keeps its host arrays.  Output buffers are
 * caller-owned.
 * Errors: every function returns SHRAY_OK (0) or a negative code; a
 * human-readable message for the calling thread's last failure is available
 * from shray_last_error().  Nothing exits or throws across this boundary.
 * Threading: calls on one handle are not re-entrant; different handles may be
 * driven from different threads / processes (one handle per GPU).
 */
#ifndef SHADER_RAY_HIP_H
#define SHADER_RAY_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Bumped whenever a struct below changes layout or an entry point changes meaning; callers compare it with
 * shray_abi_version() before their first call.  1: round 1.  2: shray_tile_set grew tile_phase_count (20 bytes).
 * 3: round 3 (shray_scene_set_environment_storage, shray_render_counters_timed, shray_scene_dispatch_order,
 * shray_selftest_reciprocal, the shray_dist_* companion header).  4: round 4.  Round 5 only ADDED entry points
 * (shray_bvh_build_device, shray_device_tree_*): no bump. */
#define SHRAY_ABI_VERSION 4

enum {
    SHRAY_OK = 0,
    SHRAY_ERR_INVALID_ARGUMENT = -1,
    SHRAY_ERR_NO_DEVICE = -2,       /* HIP runtime reports no usable gfx950 device */
    SHRAY_ERR_DEVICE = -3,          /* a HIP call failed; see shray_last_error() */
    SHRAY_ERR_OUT_OF_MEMORY = -4,
    SHRAY_ERR_INDEX_RANGE = -5,     /* an index does not survive float32 storage (>= 2^24) */
    SHRAY_ERR_BAD_TREE = -6,        /* hit/miss tables are not a well-formed threaded tree */
    SHRAY_ERR_NO_ENVIRONMENT = -7   /* render called before shray_scene_set_environment */
};

/*
 * Flattened scene: the arrays get_shader_data() produces (world.cpp:298-347),
 * taken verbatim.  Every array is float32, padded to data_texture_width *
 * rows texels like the textures they were uploaded as (ray.cpp:470-497).
 *
 *   vertex_positions / vertex_normals : 3 floats per vertex, 3 vertices per
 *       triangle in post-build triangle order (world.cpp:304-317)
 *   group_boxmin / group_boxmax       : 3 floats per node (world.cpp:183-188)
 *   group_hitmiss                     : 8 tables of (hit, miss) float pairs;
 *       table c starts at texel c * data_texture_width * group_data_rows
 *       (world.cpp:343-346, raytracer.es.fs:392); >= 16777215 terminates
 *   group_objects                     : (start, count) per node, 0,0 for
 *       branches (world.cpp:201-208)
 *
 * vertex_colors, group_directions and group_children are uploaded by the
 * reference but never read by the shader; they may be NULL here.
 */
typedef struct shray_scene_desc {
    uint32_t struct_size;          /* sizeof(shray_scene_desc), for ABI checking */
    uint32_t data_texture_width;   /* ray.cpp:326, 2048 upstream */

    uint32_t vertex_count;         /* 3 * triangle count */
    uint32_t vertex_data_rows;
    const float *vertex_positions;
    const float *vertex_normals;
    const float *vertex_colors;    /* optional, unused */

    int32_t group_count;
    int32_t group_data_rows;
    int32_t tree_root;
    const float *group_boxmin;
    const float *group_boxmax;
    const float *group_directions; /* optional, unused */
    const float *group_children;   /* optional, unused */
    const float *group_hitmiss;
    const float *group_objects;
} shray_scene_desc;

/*
 * Per-frame parameter block: the uniforms DrawFrame sets (ray.cpp:648-704)
 * plus the shader's compile-time constants, which are exposed here with the
 * reference values as defaults (see shray_frame_params_init).
 */
typedef struct shray_frame_params {
    uint32_t struct_size;            /* sizeof(shray_frame_params) */

    int32_t which;                   /* ray.cpp:648; 0 = normal rendering; 1 = environment through
                                        textureGrad with the ray differentials (raytracer.es.fs:144-146),
                                        2, 3 = differential debug views (:147-149, :642-650), 5 = 5x5
                                        supersampled reference image (:654-673); others render like 0 */

    float camera_matrix[16];         /* ray.cpp:654 */
    float camera_normal_matrix[16];  /* ray.cpp:655 */
    float object_matrix[16];         /* ray.cpp:656 */
    float object_inverse[16];        /* ray.cpp:657 (unused by the shader) */
    float object_normal_matrix[16];  /* ray.cpp:658 */
    float object_normal_inverse[16]; /* ray.cpp:659 */

    float image_plane_width;         /* ray.cpp:672, 2*tanf(fov/2) */
    float aspect;                    /* ray.cpp:673, height / width */
    float right[3];                  /* ray.cpp:677-679 */
    float up[3];                     /* ray.cpp:681-683 */
    float light_dir[3];              /* ray.cpp:695 */
    float specular_color[3];         /* ray.cpp:699 */
    float diffuse_color[3];          /* ray.cpp:700-704; 0 for metals */

    /* shader constants (raytracer.es.fs:550,381,382,445,525) */
    int32_t bounce_count;            /* 3 */
    int32_t max_bvh_iterations;      /* 400 */
    int32_t max_leaf_tests;          /* 10 */
    int32_t cast_shadows;            /* 1 */
    int32_t tonemap;                 /* 1 (filmic) */
    int32_t normals_fp16;            /* 1: normals were uploaded as GL_RGB16F, ray.cpp:474 */
} shray_frame_params;

/* Interleaved-tile ownership for splitting one frame over several GPUs.
 * The frame is cut into tile_w x tile_h tiles, numbered row-major; this
 * handle renders the tiles with (tile_index % tile_stride) == tile_phase and
 * packs them densely, in increasing tile index, into the output buffer
 * (each tile stored row-major, tile_w*tile_h RGBA texels, edge tiles padded).
 * {0,0,1,0} or NULL = whole frame, plain row-major output. */
typedef struct shray_tile_set {
    int32_t tile_w;
    int32_t tile_h;
    int32_t tile_stride;       /* period: tile t belongs to this set when phase <= t % stride < phase + count */
    int32_t tile_phase;
    int32_t tile_phase_count;  /* consecutive phases owned (0 means 1): ranks may own unequal shares of a frame,
                                  e.g. rank 0, which also receives and de-interleaves, 2 of 23 and its peers 3 */
} shray_tile_set;

/* Work counters of one render, summed over all pixels and samples.  They feed
 * the algorithmic-bytes formula of the roofline report. */
typedef struct shray_counters {
    uint64_t node_visits;      /* group_intersect loop iterations (raytracer.es.fs:395) */
    uint64_t leaf_visits;      /* leaves whose (start,count) was fetched (:263-267) */
    uint64_t triangle_tests;   /* triangle_intersect calls (:416) */
    uint64_t shaded_hits;      /* shade() calls with a valid hit (:505) */
    uint64_t env_lookups;      /* sample_environment calls (:580) */
    uint64_t traversals;       /* group_intersect calls, closest-hit + shadow */
    uint64_t bad_hits;         /* samples that returned the iteration-cap marker (:436-438) */
    uint64_t samples;          /* width * height * spp */
} shray_counters;

typedef struct shray_scene shray_scene;

/* Library / device ------------------------------------------------------- */
int shray_abi_version(void);
const char *shray_last_error(void);
int shray_device_count(int *count);
/* Selects the HIP device used by scenes created afterwards on this thread. */
int shray_set_device(int device_index);

/* Parameters ------------------------------------------------------------- */
/* Fills struct_size, identity matrices and the shader-constant defaults. */
void shray_frame_params_init(shray_frame_params *params);

/* GPU-side flattener ------------------------------------------------------ */
/*
 * get_shader_data (world.cpp:298-347) on the GPU: the host-built BVH goes in as plain arrays, nodes in
 * PRE-order (a node, then its negative subtree, then its positive subtree; root = node 0), and the
 * scene_shader_data arrays come out in device memory, bit-identical to the host flattener's: triangle
 * corners expanded to float3 arrays (world.cpp:303-318), nodes numbered in-order with their boxes, split
 * directions, children and leaf ranges (store_group_data, world.cpp:179-210), and the eight threaded
 * (hit, miss) tables (create_hitmiss, world.cpp:231-288) -- every node finds its own links by walking up
 * its ancestors, so all nodes and all eight direction codes are threaded in parallel.
 * libshray_host's shray_host_export_tree() fills the description from a loaded world.
 */
typedef struct shray_tree_desc {
    uint32_t struct_size;            /* sizeof(shray_tree_desc) */
    int32_t node_count;
    const int32_t *node_parent;      /* -1 for the root */
    const int32_t *node_negative;    /* child indices, -1 for a leaf (group.h:29-30) */
    const int32_t *node_positive;
    const float *node_box;           /* 6 floats per node: boxmin.xyz, boxmax.xyz */
    const float *node_direction;     /* 3 floats per node: the split direction D (group.h:24) */
    const int32_t *node_start;       /* leaves: first triangle and triangle count (group.h:35-36) */
    const int32_t *node_triangles;
    int32_t triangle_count;
    const int32_t *triangle_vertices; /* 3 vertex indices per triangle, in post-build order */
    int32_t vertex_count;
    const float *vertex_data;        /* 9 floats per vertex: position, colour, normal (geometry.h:34-38) */
} shray_tree_desc;

typedef struct shray_device_flat shray_device_flat;   /* the flattened arrays, resident on the device */

int shray_flatten_device(const shray_tree_desc *tree, uint32_t data_texture_width, shray_device_flat **out_flat);
/* *desc gets the counts and DEVICE pointers of the arrays (valid until the object is destroyed). */
int shray_device_flat_describe(const shray_device_flat *flat, shray_scene_desc *desc);
/* Copies the arrays to host memory owned by the object; *desc gets the counts and HOST pointers. */
int shray_device_flat_download(shray_device_flat *flat, shray_scene_desc *desc);
int shray_device_flat_destroy(shray_device_flat *flat);

/* BVH build on the GPU ----------------------------------------------------- */
/* Replaces make_bvh (reference bvh.h:17-21, bvh.cpp:288-358; get_best_split :198-247, sah :107-120, partition :249-286,
 * make_leaf :122-135): the binned-SAH build over a triangle_set's triangles, level by level on the device
 * (csrc/bvh_build.hip).  The tree, its boxes and the post-build ORDER of the triangles are those of the reference's
 * depth-first host build, bit for bit (same float expressions, same exchange partition).
 *   triangle_vertices   3 vertex indices per triangle, in load order (triangle_set::triangles[k].i, triangle-set.h)
 *   vertex_data         vertex_stride_floats floats per vertex, the position first (geometry.h:34-38: 9 packed floats)
 *   options             NULL = the reference's defaults (bvh.cpp:28-58: leaf_max 10, max_depth 30, SAH 1 + 4 n)
 * shray_device_tree_download: the tree as the pre-order arrays of shray_tree_desc -- what shray_host_export_tree gives for a
 * host-built tree and what shray_flatten_device takes; desc->triangle_vertices is in post-build order, desc->vertex_data the
 * caller's pointer; *triangle_order (may be NULL) = which load-order triangle sits at each post-build position.  The arrays
 * stay owned by the object. */
typedef struct shray_bvh_options {
    uint32_t struct_size;            /* sizeof(shray_bvh_options) */
    int32_t max_depth, leaf_max;     /* BVH_MAX_DEPTH, BVH_LEAF_MAX (bvh.cpp:60-79) */
    float sah_ctrav, sah_cisec;      /* SAH_CTRAV, SAH_CISEC */
} shray_bvh_options;
typedef struct shray_bvh_stats {
    int32_t node_count, leaf_count, max_level, large_leaves;   /* print_bvh_stats, bvh.cpp:83-99 */
    double device_seconds;           /* the build on the device, without the upload of its inputs and the download of the tree */
} shray_bvh_stats;
typedef struct shray_device_tree shray_device_tree;
int shray_bvh_build_device(const int32_t *triangle_vertices, int32_t triangle_count, const float *vertex_data, int32_t vertex_count,
                           int32_t vertex_stride_floats, const shray_bvh_options *options, shray_device_tree **out_tree);
int shray_device_tree_download(shray_device_tree *tree, shray_tree_desc *desc, const int32_t **triangle_order);
int shray_device_tree_stats(const shray_device_tree *tree, shray_bvh_stats *stats);
int shray_device_tree_destroy(shray_device_tree *tree);

/* The device-resident scene pipeline (round 6): a tree that shray_bvh_build_device made is flattened and becomes a scene WITHOUT
 * leaving the device -- no download of the tree, no group tree on the host (world.cpp:46-134 builds one; it can still be had on
 * demand: shray_device_tree_download + libshray_host's shray_host_adopt_tree), no re-upload:
 *   shray_flatten_device_tree       get_shader_data (world.cpp:298-347) of the tree where it lies; the tree was built over vertices
 *                                   of geometry.h:34-38's layout (vertex_stride_floats == 9)
 *   shray_scene_create_from_device  what shray_scene_create derives on the host from the flattened arrays -- the packed tree and its
 *                                   eight octant copies, the packed triangles, the pair records, the fp16 normals, the deepest ray
 *                                   stack -- computed on the device from the tree itself; the arrays equal shray_scene_create's bit for
 *                                   bit.  `tree` and `flat` may be destroyed afterwards (the scene holds copies). */
int shray_flatten_device_tree(const shray_device_tree *tree, uint32_t data_texture_width, shray_device_flat **out_flat);
int shray_scene_create_from_device(const shray_device_tree *tree, const shray_device_flat *flat, shray_scene **out_scene);
/* For tests and tools: copies the scene's derived arrays back -- the packed nodes (eight copies x nodes x 32 bytes), the packed
 * triangles (36 bytes each, without the spare record), the fp16 normals -- into caller memory of at least *_bytes bytes (any
 * pointer may be NULL); *stack_levels = the deepest ray stack the scene's kernels provide for.  Sizes: shray_scene_derived_sizes. */
int shray_scene_derived_sizes(const shray_scene *scene, uint64_t *packed_nodes_bytes, uint64_t *packed_tris_bytes, uint64_t *normals16_bytes,
                              uint64_t *pair_nodes_bytes, int32_t *stack_levels);
int shray_scene_derived_download(const shray_scene *scene, void *packed_nodes, void *packed_tris, void *normals16, void *pair_nodes);

/* Scene ------------------------------------------------------------------ */
/* Device memory a scene takes besides the reference's own arrays (which stay resident for the literal kernel): 8 x 32 bytes per
 * node (one repacked copy of the tree per ray-direction octant), 36 bytes per triangle, 64 bytes per node for the pair kernel,
 * normals in fp16 and fp32 -- 12 MB for the 69k-triangle benchmark mesh, 190 MB for a 1M-triangle one. */
int shray_scene_create(const shray_scene_desc *desc, shray_scene **out_scene);
int shray_scene_set_environment(shray_scene *scene, const float *rgb, int width, int height);
/* How the environment is STORED.  The reference uploads its float image with an unsized GL_RGB internal format
 * (ray.cpp:508), which most drivers resolve to 8-bit normalized fixed point: values are clamped to [0, 1] and kept
 * in 1/255 steps, mip levels (ray.cpp:509) included -- an "HDR" environment loses everything above 1.
 * SHRAY_ENV_FLOAT32 (what shray_scene_set_environment uses, and what a literal evaluation of the shader on the
 * float image sees) keeps the floats; SHRAY_ENV_UNORM8 reproduces the common driver behaviour: every texel, and
 * every texel of every mip level, passes through c = floor(255 clamp(f, 0, 1) + 0.5), f = c / 255. */
enum { SHRAY_ENV_FLOAT32 = 0, SHRAY_ENV_UNORM8 = 1 };
int shray_scene_set_environment_storage(shray_scene *scene, const float *rgb, int width, int height, int storage);
int shray_scene_destroy(shray_scene *scene);
/* The HIP device the scene's buffers live on (the device that was current when it was created). */
int shray_scene_device(const shray_scene *scene, int *device_index);

/* Kernel selection: 0 = per-ray LDS stack kernel (default), 1 = literal threaded hit/miss-table traversal
 * over the reference arrays, 2 = pool kernel (a workgroup's waves merge their live rays while they
 * traverse: for divergent scenes), 3 = the stack kernel with both children of a node tested per turn in every
 * launch that can, 4 = the wavefront form (one launch per bounce, the live paths compacted in between; whole frames
 * of the plain view, everything else runs kernel 0's instances).  All of them produce bit-identical frames and,
 * through shray_render_counters, the same work counters. */
int shray_scene_set_kernel(shray_scene *scene, int kernel_id);

/* Render ----------------------------------------------------------------- */
/* Output is RGBA float32, row 0 = bottom row of the image (GL origin,
 * raytracer.vs:56), alpha = 1 (raytracer.es.fs:676). */
int shray_render(shray_scene *scene, const shray_frame_params *params,
                 int width, int height, int spp, float *rgba_out_host);
/* shray_render is the blocking readback form (screenshot, ray.cpp:760): the frame is rendered into a device
 * frame the scene keeps, then copied out on the scene's own stream.  When rgba_out_host is pinned host
 * memory (shray_pinned_alloc, or any hipHostMalloc / hipHostRegister memory) the copy is one DMA at PCIe
 * speed; pageable memory is staged by the HIP runtime.
 *
 * shray_render_host_async is the non-blocking form for frame loops: render + DMA into PINNED host memory
 * are enqueued on hip_stream and the call returns; the frame is complete once the stream reaches that
 * point (hipStreamSynchronize / an event).  One in-flight call per scene (it uses the scene's frame). */
int shray_render_host_async(shray_scene *scene, const shray_frame_params *params,
                            int width, int height, int spp, float *rgba_out_pinned, void *hip_stream);
/* Pinned host memory for the two calls above. */
int shray_pinned_alloc(size_t bytes, void **out_ptr);
int shray_pinned_free(void *ptr);

/* Asynchronous form: d_rgba_out is device memory on the scene's device,
 * hip_stream a hipStream_t (NULL = default stream).  tiles may be NULL.
 * Required size in bytes: shray_tile_buffer_bytes(). */
int shray_render_device(shray_scene *scene, const shray_frame_params *params,
                        int width, int height, int spp,
                        const shray_tile_set *tiles,
                        void *d_rgba_out, void *hip_stream);

int64_t shray_tile_buffer_bytes(int width, int height, const shray_tile_set *tiles);

/* `count` consecutive frames (the DrawFrame loop of ray.cpp:1096-1131, or the frames of a
 * camera animation) in ONE launch: frame k is rendered with params[k] into
 * d_rgba_out + k * frame_stride_bytes, each frame laid out exactly as shray_render_device
 * lays out its single frame.  The few long-running waves of a frame (rays grazing the
 * silhouette) no longer leave the GPU idle: the next frames' work fills in behind them.
 * 1 <= count <= SHRAY_MAX_BATCH; frame_stride_bytes is a multiple of 16 and at least
 * shray_tile_buffer_bytes(); all frames must agree on whether `which` is a
 * differential view (1 or 2) or not. */
#define SHRAY_MAX_BATCH 64
int shray_render_batch_device(shray_scene *scene, const shray_frame_params *params, int count,
                              int width, int height, int spp,
                              const shray_tile_set *tiles,
                              void *d_rgba_out, int64_t frame_stride_bytes, void *hip_stream);

/* Multi-GPU, rank 0: de-interleaves gathered tile buffers into whole frames (the inverse of
 * shray_tile_set ownership).  d_gathered holds, for rank r = 0..world-1 and frame f = 0..frames-1,
 * that rank's packed tiles at d_gathered + r * rank_stride_bytes + f * frame_stride_bytes, pixels
 * as `channels` float32 (4 = RGBA as rendered; 3 = RGB, alpha having been dropped on the wire
 * because it is the constant 1).  d_rgba_out receives `frames` RGBA frames of width * height
 * pixels, back to back.  There is no upstream counterpart: one GPU, one framebuffer. */
int shray_assemble_tiles_device(const void *d_gathered, int world, int frames, int channels,
                                int64_t rank_stride_bytes, int64_t frame_stride_bytes,
                                int width, int height, int tile_w, int tile_h,
                                void *d_rgba_out, void *hip_stream);
/* The same for unequal shares: the tile period is rank0_phases + (world - 1) * other_phases; rank 0 owns the
 * first rank0_phases phases of every period, rank r >= 1 the other_phases phases from rank0_phases +
 * (r - 1) * other_phases (shray_tile_set with that phase and count).  (1, 1) is the even split above. */
int shray_assemble_tiles_split_device(const void *d_gathered, int world, int rank0_phases, int other_phases,
                                      int frames, int channels, int64_t rank_stride_bytes, int64_t frame_stride_bytes,
                                      int width, int height, int tile_w, int tile_h,
                                      void *d_rgba_out, void *hip_stream);

/* Same render with per-ray work counters accumulated (slower; used for the
 * roofline's algorithmic-byte count and for parity of the traversal itself). */
int shray_render_counters(shray_scene *scene, const shray_frame_params *params,
                          int width, int height, int spp,
                          float *rgba_out_host /* may be NULL */,
                          shray_counters *counters);

/* The counters above are those of the reference's traversal: the counting kernels walk every shadow ray to its end
 * (raytracer.es.fs:447-472 does), so that they can be compared with the CPU evaluation node for node.  The kernels
 * that shray_render / shray_render_batch_device actually launch stop a shadow ray at its first hit (the shader only
 * asks whether anything is hit, fs:516-521) and run a pixel's samples in neighbouring lanes; this entry point
 * renders the frame with THAT instance -- the one a launch of frames_per_launch such frames would select -- and
 * returns ITS tallies: equal to the reference's for metals (no shadow rays), smaller for diffuse materials; the
 * image is the same either way.  bad_hits counts primary / bounce samples that returned the iteration-cap marker. */
int shray_render_counters_timed(shray_scene *scene, const shray_frame_params *params,
                                int width, int height, int spp, int frames_per_launch,
                                float *rgba_out_host /* may be NULL */,
                                shray_counters *counters);

/* Dispatch order ------------------------------------------------------------ */
/* The batch kernels of plain (which == 0) frames start a launch's heaviest 16x16 patches first: every wave leaves its
 * running time per patch, and every few launches of one shape (frame size, spp, tile set) the library re-sorts the
 * patches for the launches that follow (csrc/capi.hip: DispatchOrder).  The frames do not depend on it; what does is
 * how long a launch waits for its slowest waves when nothing follows it at once.  This call returns the permutation the
 * next launch of the current shape would read: order_out[k] = the patch rendered by the launch's k-th patch slot.
 * *count_out = its length (0: no shape yet, or the identity is still in use); at most `capacity` entries are written.
 * It waits for the device.  The environment variable SHRAY_DISPATCH_ORDER=0 turns the re-ordering off for a process. */
int shray_scene_dispatch_order(shray_scene *scene, uint32_t *order_out, uint32_t capacity, uint32_t *count_out);

/* Self-test ---------------------------------------------------------------- */
/* Runs the kernel's 5-instruction "divide by a per-ray constant" (csrc/exact_div.h)
 * against true IEEE division on `pairs` pseudo-random operand pairs from the operand
 * ranges it is used in, plus structured hard cases; *mismatches = number of pairs whose
 * results are not bit-identical (must be 0). */
int shray_selftest_division(uint64_t pairs, uint64_t seed, uint64_t *mismatches);
/* The kernels' three-instruction reciprocal (v_rcp_f32 and one Newton step in FMAs; csrc/exact_div.h) against true IEEE
 * division on EVERY float of the domain it is used in (2^-100 <= |x| < 2^100, 3.4e9 values); *mismatches must be 0. */
int shray_selftest_reciprocal(uint64_t *mismatches);

/* Measurement aid (bench.py's roofline): the rate at which the GPU's vector L1 caches serve the access pattern of a node
 * visit -- every lane of a wave reads whole 32-byte records (two 16-byte loads) of a table of `records` records, the 64
 * lanes of one wave-instruction among `spread` consecutive records around a pseudo-random base (powers of two; spread 1 =
 * the whole wave at one record, 64 = every lane elsewhere; with bit 31 set the lanes that share a record are runs of
 * neighbouring lanes instead of a pseudo-random choice), `visits_per_lane` (a multiple of 8) visits per lane in `waves`
 * one-wave workgroups, eight visits in flight per wave.  *seconds = the launch's duration (HIP events, second of two
 * passes), *bytes_loaded = waves * popcount(lane_mask) * visits_per_lane * bytes_per_lane; only the lanes of lane_mask
 * load; bytes_per_lane = 32 (the node's two 16-byte loads) or 16, 12, 8, 4 (one load of that width of the record's first bytes).  The traversal kernels' node and triangle fetches are bound
 * by this rate, not by HBM. */
int shray_probe_vector_cache(uint32_t records, uint32_t spread, uint32_t visits_per_lane, uint32_t waves, uint64_t lane_mask,
                             int bytes_per_lane, double *seconds, uint64_t *bytes_loaded);

#ifdef __cplusplus
}
#endif
#endif /* SHADER_RAY_HIP_H */
