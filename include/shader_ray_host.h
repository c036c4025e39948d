/*
 * shader_ray_host.h -- C view of the host-side loader / BVH / flattener.
 *
 * The host layer itself is C++ and keeps the reference's API (world.h,
 * bvh.h, group.h, triangle-set.h, trisrc-support.h, obj-support.h under
 * shader-ray_amd/host/).  These wrappers exist so that non-C++ callers (the
 * Python test and bench drivers, via ctypes) can reach it; each names the
 * C++ entry point -- and through it the reference interface -- it forwards to.
 */
#ifndef SHADER_RAY_HOST_H
#define SHADER_RAY_HOST_H

#include "shader_ray_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct shray_host_world shray_host_world;

/* Interactive state of the reference's shell (globals at ray.cpp:38-74) with
 * the start-up values of ray.cpp:1077-1088. */
typedef struct shray_host_view {
    float fov;                 /* radians; 40 degrees */
    float zoom;                /* scene_extent / 2 / sinf(fov / 2) */
    float object_rotation[4];  /* angle, axis */
    float object_position[3];
    float light_rotation[4];   /* -20 degrees about (.707, -.707, 0) */
    int32_t which;
    int32_t which_material;    /* index into the materials table, ray.cpp:54-65 */
    int32_t which_diffuse_color; /* ray.cpp:68-73 */
} shray_host_view;

typedef struct shray_host_world_info {
    int32_t triangle_count;
    int32_t independent_vertex_count;
    float scene_center[3];
    float scene_extent;
    int32_t node_count;
    int32_t leaf_count;
    int32_t max_level;
    int32_t large_leaves;
    double parse_seconds;        /* file -> triangle_set (parse, shared vertices, normals), centre + extent */
    double build_seconds;        /* make_bvh */
} shray_host_world_info;

/* load_world() (world.h:64): parse .trisrc / .obj, centre + extent, make_bvh.
 * Returns 0 or -1 (message on stderr, like the reference). */
int shray_host_load_world(const char *filename, shray_host_world **out_world);
void shray_host_free_world(shray_host_world *world);
int shray_host_get_world_info(const shray_host_world *world, shray_host_world_info *info);

/* load_world() in two steps, for a BVH built by shray_bvh_build_device (shader_ray_hip.h) instead of make_bvh:
 *   shray_host_load_triangles   parse, shared vertices, normals, centre + extent -- no BVH yet
 *   shray_host_triangles        what the build takes: 3 vertex indices per triangle in load order, 9 floats per vertex
 *                               (pointers into the world, valid until it is freed or a tree is adopted)
 *   shray_host_adopt_tree       installs the tree (pre-order arrays) and the build's triangle order; build_seconds is recorded
 *                               as the world's BVH time.  -1 if the arrays are not a pre-order binary tree over these triangles. */
int shray_host_load_triangles(const char *filename, shray_host_world **out_world);
int shray_host_triangles(shray_host_world *world, const int32_t **triangle_vertices, int32_t *triangle_count, const float **vertex_data,
                         int32_t *vertex_count);
int shray_host_adopt_tree(shray_host_world *world, const shray_tree_desc *tree, const int32_t *triangle_order, double build_seconds);
/* The build options make_bvh reads from the environment (BVH_MAX_DEPTH, BVH_LEAF_MAX, SAH_CTRAV, SAH_CISEC: bvh.cpp:60-79), as this
 * process's host builder sees them -- what a caller of shray_bvh_build_device passes on so that both builders follow the same
 * environment (struct_size is set). */
int shray_host_bvh_options(shray_bvh_options *options);

/* get_shader_data() (world.h:95).  The arrays named by *desc stay owned by
 * `world` and valid until it is freed.  One flattening is kept per
 * data_texture_width: asking again for the same width returns the same arrays. */
int shray_host_flatten(shray_host_world *world, unsigned int data_texture_width, shray_scene_desc *desc);

/* The BVH as plain pre-order arrays (shray_tree_desc, shader_ray_hip.h) for the GPU-side flattener
 * shray_flatten_device.  The arrays stay owned by `world` until it is freed or exported again. */
int shray_host_export_tree(shray_host_world *world, shray_tree_desc *tree);

/* Start-up view (ray.cpp:1077-1088) and the per-frame block (ray.cpp:648-704). */
int shray_host_default_view(const shray_host_world *world, shray_host_view *view);
int shray_host_frame_params(shray_host_world *world, const shray_host_view *view, int width, int height,
                            shray_frame_params *params);

/* trackball_motion() (ray.cpp:91-98): the rotation {angle, axis} after a mouse drag of (dx, dy) window
 * fractions on top of `previous`; (0, 0) leaves *result untouched, as upstream. */
int shray_host_trackball_motion(const float previous[4], float dx, float dy, float result[4]);

/* load_background() (host/background.h): the reference's background argument (ray.cpp:1002-1075):
 * "r, g, b", "grid", "rrggbb" or a Radiance .hdr file.  *pixels (3 floats per pixel, row 0 =
 * bottom row) is malloc'ed; release it with shray_host_free_background. */
int shray_host_load_background(const char *spec, int *width, int *height, float **pixels);
void shray_host_free_background(float *pixels);

/* Quiet (1) drops the loaders' progress chatter on stderr; errors still print. */
void shray_host_set_quiet(int quiet);

#ifdef __cplusplus
}
#endif
#endif
