// frame-params.h -- the per-frame parameter block of the tracer, computed on
// the host exactly as the reference's interactive shell does before each
// draw call:
//   camera / object matrices   ray.cpp:100-140, :162-173
//   light direction            ray.cpp:142-160
//   image plane, right/up      ray.cpp:672-683
//   material table             ray.cpp:48-74, :698-704
//   start-up defaults          ray.cpp:1077-1088 (fov 40 deg, zoom, light)
// The result is the C-ABI struct the HIP layer consumes (shader_ray_hip.h).
#pragma once

#include "shader_ray_hip.h"
#include "world.h"

struct material {
    vec3 specular_color;
    bool metal;
};

// Hoffman's S2010 table, in the reference's order: gold, silver, copper,
// iron, aluminium, plastic/glass (low), plastic (high)   (ray.cpp:54-65).
extern const material materials[7];
extern const int material_count;
// white, reddish, green, blueish (ray.cpp:68-73)
extern const vec3 diffuse_colors[4];
extern const int diffuse_color_count;

void create_camera_matrix(const vec3 &viewpoint, float matrix[16], float normal_matrix[16]);
void create_object_matrix(const vec3 &center, const float rotation[4], const vec3 &position, float matrix[16],
                          float inverse[16], float normal[16], float normal_inverse[16]);
// Rotates (0,0,1) by the axis-angle `light_rotation` (ray.cpp:142-160).
vec3 compute_light_dir(const float light_rotation[4]);
// Fills the six matrices of `w` (ray.cpp:162-173).
void update_view_params(world_ptr w, float zoom, const float object_rotation[4], const vec3 &object_position);

// Mouse-drag trackball of the reference's shell (ray.cpp:76-98): a drag by (dx, dy) in window
// fractions becomes an axis-angle rotation (angle = pi * |drag|, axis = (dy, dx, 0) / |drag|),
// composed onto the previous rotation.  dx = dy = 0 leaves `newrotation` untouched, as upstream.
void drag_to_rotation(float dx, float dy, float rotation[4]);
void trackball_motion(float prevrotation[4], float dx, float dy, float newrotation[4]);

// Interactive state of the reference's shell with its start-up values.
struct view_state {
    float fov;                  // radians
    float zoom;
    float object_rotation[4];   // angle, axis
    vec3 object_position;
    float light_rotation[4];
    int which;
    int which_material;
    int which_diffuse_color;
};
view_state default_view_state(const world_ptr &w);

// Applies `view` to the world's matrices and produces the frame block for a
// width x height frame.
void make_frame_params(world_ptr w, const view_state &view, int width, int height, shray_frame_params *out);
