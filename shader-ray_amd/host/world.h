// world.h -- scene container, loader entry point and the flattened
// "shader data" arrays that cross into the GPU path.
//
// Public names and field meanings follow the reference (world.h:28-95) so
// that code written against it keeps compiling:
//   struct camera, struct world, world_ptr, load_world(),
//   struct scene_shader_data, get_shader_data(), trace_image().
#pragma once

#include <memory>
#include <string>

#include "geometry.h"
#include "group.h"
#include "triangle-set.h"
#include "vectormath.h"

struct camera {
    float fov;   // full horizontal view angle, radians
};

struct world {
    int triangle_count;
    triangle_set_ptr triangles;   // traced only through `root`
    group *root;                  // owned

    vec3 scene_center;
    float scene_extent;           // diameter of the bounding sphere about scene_center

    camera cam;
    int xsub, ysub;

    // eye -> world, and the six world <-> object matrices the shader receives (column-major)
    float camera_matrix[16], camera_normal_matrix[16];
    float object_matrix[16], object_inverse[16];
    float object_normal_matrix[16], object_normal_inverse[16];

    // what load_world spent where (the reference prints these, world.cpp:93, :109, :116): file -> triangle_set;
    // centre + extent; make_bvh
    double parse_seconds = 0, extent_seconds = 0, build_seconds = 0;

    world();
    ~world();
    world(const world &) = delete;
    world &operator=(const world &) = delete;
};

typedef std::shared_ptr<world> world_ptr;

// Loads a .trisrc or .obj file, computes centre / extent, builds the BVH.
// Returns nullptr (after a message on stderr) on any failure (world.cpp:46-134).
world_ptr load_world(const std::string &filename);

// load_world in two steps, for a BVH that is built elsewhere (the GPU build of include/shader_ray_hip.h,
// shray_bvh_build_device): load_triangles is load_world without make_bvh (root stays null); adopt_tree installs a tree given as
// pre-order arrays (shray_tree_desc's) and puts the triangles into the build's order (triangle_order[k] = the load-order
// triangle at post-build position k).  False -- and nothing changed -- if the arrays are not a pre-order binary tree over
// exactly these triangles.
world_ptr load_triangles(const std::string &filename);
bool adopt_tree(const world_ptr &w, int node_count, const int *negative, const int *positive, const float *box, const float *direction,
                const int *start, const int *triangles, const int *triangle_order, int triangle_count);

// Declared by the reference (world.h:65) but never defined there.  Here it
// renders the world through the HIP layer (shader_ray_hip.h) with the
// reference's default material and an all-white environment unless one was
// installed with set_trace_environment(); `image` receives width*height RGB8,
// top row first.  Implemented in tools/trace_image.cpp (needs the HIP library).
void trace_image(int width, int height, float aspect, unsigned char *image, const world_ptr Wd, const vec3 &light_dir);
void set_trace_environment(const float *rgb, int width, int height);   // [height][width][3], row 0 = straight down

// The arrays the fragment shader samples, exactly as the reference lays them
// out (world.h:68-93, world.cpp:298-347).  All float32; every array is padded
// to data_texture_width * rows elements (padding is zero here).
struct scene_shader_data {
    // per-triangle vertex attributes: float3 per vertex, three vertices per triangle
    unsigned int vertex_count, vertex_data_rows;
    float *vertex_positions, *vertex_colors, *vertex_normals;

    // per-node arrays, in-order node numbering
    int group_count, group_data_rows, tree_root;
    float *group_boxmin, *group_boxmax;   // float3
    float *group_directions;              // float3, branches only
    float *group_children;                // float2; 2147483648 for leaves
    float *group_hitmiss;                 // 8 tables of float2 (hit, miss); 2147483648 terminates
    float *group_objects;                 // float2 (start, count); 0,0 for branches

    // not upstream: false when a link table could not be threaded (tree deeper than the 64-entry link stack
    // of world.cpp:228, where upstream asserts); such arrays must not be rendered
    bool links_complete = true;

    scene_shader_data();
    ~scene_shader_data();
    scene_shader_data(const scene_shader_data &) = delete;
    scene_shader_data &operator=(const scene_shader_data &) = delete;
};

void get_shader_data(world_ptr w, scene_shader_data &data, unsigned int data_texture_width);
