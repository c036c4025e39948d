// frame_params_defaults.h -- the one place the shader's compile-time
// constants live as defaults (raytracer.es.fs:550 bounce_count, :381
// max_bvh_iterations, :382 max_leaf_tests, :445 cast_shadows, :524-525
// tonemap; ray.cpp:474 normals stored as GL_RGB16F).  Shared by the HIP
// layer (shray_frame_params_init) and the host layer.
#pragma once

#include <string.h>

#include "shader_ray_hip.h"

static inline void shray_frame_params_defaults(shray_frame_params *p)
{
    static const float identity[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
    memset(p, 0, sizeof(*p));
    p->struct_size = (uint32_t)sizeof(*p);
    memcpy(p->camera_matrix, identity, sizeof(identity));
    memcpy(p->camera_normal_matrix, identity, sizeof(identity));
    memcpy(p->object_matrix, identity, sizeof(identity));
    memcpy(p->object_inverse, identity, sizeof(identity));
    memcpy(p->object_normal_matrix, identity, sizeof(identity));
    memcpy(p->object_normal_inverse, identity, sizeof(identity));
    p->image_plane_width = 1.0f;
    p->aspect = 1.0f;
    p->light_dir[2] = 1.0f;
    p->specular_color[0] = p->specular_color[1] = p->specular_color[2] = 1.0f;
    p->bounce_count = 3;
    p->max_bvh_iterations = 400;
    p->max_leaf_tests = 10;
    p->cast_shadows = 1;
    p->tonemap = 1;
    p->normals_fp16 = 1;
}
