// device_types.h -- plain structs passed by value to the gfx950 kernels.
#pragma once

#include <stdint.h>

namespace shray {

// Scene arrays resident in HBM.
//
// "Reference layout" (what shray_scene_create receives, mirrored 1:1; used by
// the literal threaded kernel):
//   positions   3 x f32 per vertex, 3 vertices per triangle      (RGB32F, ray.cpp:471)
//   normals16   3 x f16 per vertex                               (RGB16F, ray.cpp:474)
//   normals32   3 x f32 per vertex (only read when normals_fp16 == 0)
//   boxmin/max  3 x f32 per node                                 (RGB32F, ray.cpp:492-497)
//   hitmiss     8 tables x table_stride x (hit, miss) f32        (RG32F, ray.cpp:484)
//   objects     (start, count) f32 per node                      (RG32F, ray.cpp:480)
//
// "Packed layout" (built once at scene_create, used by the stack kernel; see
// DESIGN.md "Data layout in HBM"):
//   nodes       32 B per node, depth-first order, near/far resolved per ray
//   tris        36 B per triangle: v0, e0 = v1 - v0, e1 = v0 - v2
struct SceneView {
    const float *positions;
    const uint16_t *normals16;
    const float *normals32;
    const float *boxmin;
    const float *boxmax;
    const float *hitmiss;
    const float *objects;
    uint32_t table_stride;   // data_texture_width * group_data_rows
    uint32_t group_count;
    uint32_t triangle_count;
    float tree_root;

    const void *packed_nodes;   // PackedNode[8][group_count]: one copy per direction octant (packed_layout.h)
    uint32_t packed_nodes_bytes;   // of one copy
    const void *packed_tris;    // PackedTri[triangle_count]
    uint32_t packed_root;       // the root's name (packed_layout.h: byte offset / 8)
    uint32_t exact_div_ok;      // every box coordinate is 0 or in [2^-70, 2^60): exact_div.h applies
    const void *pair_nodes;     // PackedPair[group_count] (packed_layout.h), or nullptr: the pair traversal is not available
    uint32_t pair_root_link;    // the root as a pair link: index | axis << 29 | leaf flag
    uint32_t pair_index_bits;   // IB: bits of the largest packed node index; a stack word keeps 29 - IB bits of r0

    const float *env;   // RGB f32, row 0 = t = 0 (straight down); level 0 of the pyramid below
    int32_t env_w, env_h;
    // mip pyramid for the which == 1 view: level k+1 = 2x2 box filter of level k, down to 1x1;
    // level k starts mip_offset[k] floats into env and is mip_w[k] x mip_h[k]
    int32_t mip_levels;
    uint32_t mip_offset[16];
    int32_t mip_w[16], mip_h[16];
};

// Per-launch parameters: the frame block plus frame geometry and tiling.
struct FrameView {
    float camera_matrix[16];
    float camera_normal_matrix[16];
    float object_matrix[16];
    float object_normal_matrix[16];
    float object_normal_inverse[16];
    float image_plane_width, aspect;
    float light_dir[3];
    float specular_color[3];
    float diffuse_color[3];
    int32_t bounce_count, max_bvh_iterations, max_leaf_tests;
    int32_t cast_shadows, tonemap, normals_fp16;
    int32_t which;            // 0 normal; 2, 3, 5: the shader's debug / reference views
    float right[3], up[3];    // one-pixel steps on the image plane (ray differentials)

    int32_t width, height, spp;
    // tiling: tile_stride == 0 means "whole frame, row-major output"
    int32_t tile_w, tile_h, tile_stride, tile_phase, tile_phase_count;   // owned: phase <= tile % stride < phase + count
    int32_t tiles_x;          // tiles per frame row
    int32_t owned_tiles;      // tiles this launch renders
    int32_t patches_x;        // 16x16 patches per row (of the frame, or of one tile)
    int32_t patches_per_unit; // patches per frame / per tile
    uint32_t total_patches;   // grid size in patches
    // multi-sample frames in the convergent batch kernel: a pixel's samples run in 2^sample_log_x x 2^sample_log_y
    // neighbouring lanes of a wave (uniform_driver.h); 0, 0 = one lane per pixel
    uint32_t sample_log_x, sample_log_y;
    // dispatch order of the convergent batch kernels (capi.hip: DispatchOrder): the launch's k-th patch slot renders patch
    // dispatch_order[k] (nullptr = k), and every wave leaves how long it ran in dispatch_cost[patch] (nullptr = not asked)
    const uint32_t *dispatch_order;
    uint32_t *dispatch_cost;
};

// SHRAY_DISPATCH_ORDER (1): heaviest patches first in the convergent batch kernels (capi.hip: DispatchOrder); 0 compiles
// the waves' part of it out (A/B builds)

struct DeviceCounters {
    unsigned long long node_visits, leaf_visits, triangle_tests, shaded_hits, env_lookups, traversals, bad_hits, samples;
};
// The counting kernels add into kCounterShards copies (shard = workgroup index mod
// kCounterShards) so that thousands of waves do not serialise on seven addresses; the host sums.
constexpr int kCounterShards = 64;

}   // namespace shray
