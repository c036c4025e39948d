// trace_image.cpp -- the reference declares
//     void trace_image(int width, int height, float aspect, unsigned char *image,
//                      const world_ptr Wd, const vec3& light_dir);        (world.h:65)
// but never defines it.  Here it is the host-side entry into the HIP layer: flatten the
// world (get_shader_data), hand the arrays to shray_scene_create, render one frame with
// the world's current matrices and write 8-bit RGB, top row first.
#include <cmath>
#include <cstdio>
#include <cstring>
#include <vector>

#include "frame-params.h"
#include "frame_params_defaults.h"
#include "shader_ray_hip.h"
#include "world.h"

namespace {
std::vector<float> g_environment = {1.0f, 1.0f, 1.0f};
int g_env_w = 1, g_env_h = 1;
}   // namespace

void set_trace_environment(const float *rgb, int width, int height)
{
    g_environment.assign(rgb, rgb + 3 * (size_t)width * height);
    g_env_w = width;
    g_env_h = height;
}

void trace_image(int width, int height, float aspect, unsigned char *image, const world_ptr Wd, const vec3 &light_dir)
{
    scene_shader_data data;
    get_shader_data(Wd, data, 2048);

    shray_scene_desc desc;
    memset(&desc, 0, sizeof(desc));
    desc.struct_size = sizeof(desc);
    desc.data_texture_width = 2048;
    desc.vertex_count = data.vertex_count;
    desc.vertex_data_rows = data.vertex_data_rows;
    desc.vertex_positions = data.vertex_positions;
    desc.vertex_normals = data.vertex_normals;
    desc.vertex_colors = data.vertex_colors;
    desc.group_count = data.group_count;
    desc.group_data_rows = data.group_data_rows;
    desc.tree_root = data.tree_root;
    desc.group_boxmin = data.group_boxmin;
    desc.group_boxmax = data.group_boxmax;
    desc.group_directions = data.group_directions;
    desc.group_children = data.group_children;
    desc.group_hitmiss = data.group_hitmiss;
    desc.group_objects = data.group_objects;

    shray_scene *scene = nullptr;
    if (shray_scene_create(&desc, &scene) != SHRAY_OK ||
        shray_scene_set_environment(scene, g_environment.data(), g_env_w, g_env_h) != SHRAY_OK) {
        fprintf(stderr, "trace_image: %s\n", shray_last_error());
        shray_scene_destroy(scene);
        return;
    }

    shray_frame_params p;
    shray_frame_params_defaults(&p);
    memcpy(p.camera_matrix, Wd->camera_matrix, 64);
    memcpy(p.camera_normal_matrix, Wd->camera_normal_matrix, 64);
    memcpy(p.object_matrix, Wd->object_matrix, 64);
    memcpy(p.object_inverse, Wd->object_inverse, 64);
    memcpy(p.object_normal_matrix, Wd->object_normal_matrix, 64);
    memcpy(p.object_normal_inverse, Wd->object_normal_inverse, 64);
    p.image_plane_width = 2 * tanf(Wd->cam.fov / 2.0);
    p.aspect = aspect;
    p.light_dir[0] = light_dir.x; p.light_dir[1] = light_dir.y; p.light_dir[2] = light_dir.z;
    p.specular_color[0] = materials[0].specular_color.x;
    p.specular_color[1] = materials[0].specular_color.y;
    p.specular_color[2] = materials[0].specular_color.z;

    std::vector<float> rgba((size_t)width * height * 4);
    if (shray_render(scene, &p, width, height, 1, rgba.data()) != SHRAY_OK) {
        fprintf(stderr, "trace_image: %s\n", shray_last_error());
    } else {
        for (int y = 0; y < height; y++) {
            const float *src = &rgba[(size_t)(height - 1 - y) * width * 4];   // row 0 of the render is the bottom
            unsigned char *dst = image + (size_t)y * width * 3;
            for (int x = 0; x < width; x++)
                for (int c = 0; c < 3; c++) {
                    const float v = src[4 * x + c];
                    dst[3 * x + c] = (unsigned char)(v <= 0 ? 0 : (v >= 1 ? 255 : (int)(v * 255.0f + 0.5f)));
                }
        }
    }
    shray_scene_destroy(scene);
}
