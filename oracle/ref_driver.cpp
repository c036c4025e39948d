// ref_driver.cpp -- TEST INFRASTRUCTURE, not product code.
//
// A small main() that links against the REFERENCE's own host sources,
// compiled where they lie under $(REF) (see oracle/Makefile), to obtain the
// reference's actual outputs for the boundary half of the hot path:
//   load_world()       world.cpp:46      parse + centre/extent + make_bvh
//   get_shader_data()  world.cpp:298     the flattened arrays the shader reads
//   update_view_params / update_light    ray.cpp:100-173 (sliced at build time)
//   start-up defaults                    ray.cpp:1077-1088 (sliced at build time)
// It dumps them as named float32 sections; tests/golden/make_golden.py turns
// the dumps into committed fixtures and tests compare the repo's own host
// layer against them bit for bit.  Nothing from the reference is copied into
// the repository; the binary lands in oracle/_ref/ (git-ignored).
//
// usage: ref_host <scene.trisrc|scene.obj> <out.bin> [width height]
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "world.h"   // the reference's, via -I$(REF)

// symbols of the ray.cpp slice (oracle/Makefile: _ref/ray_params.o)
extern world_ptr gWorld;
extern float zoom;
extern float object_rotation[4];
extern float light_rotation[4];
extern vec3 light_dir;
void ref_startup_defaults();   // body = ray.cpp:1077-1088

namespace {

FILE *g_out;

void section(const char *name, const float *data, uint64_t count)
{
    const uint32_t name_len = (uint32_t)strlen(name);
    fwrite(&name_len, sizeof(name_len), 1, g_out);
    fwrite(name, 1, name_len, g_out);
    fwrite(&count, sizeof(count), 1, g_out);
    if (count)
        fwrite(data, sizeof(float), count, g_out);
}

void scalar_section(const char *name, double v)
{
    const float f = (float)v;
    section(name, &f, 1);
}

}   // namespace

int main(int argc, char **argv)
{
    if (argc < 3) {
        fprintf(stderr, "usage: %s scene out.bin [width height]\n", argv[0]);
        return 2;
    }
    const int width = argc > 4 ? atoi(argv[3]) : 1920;
    const int height = argc > 4 ? atoi(argv[4]) : 1080;

    gWorld = load_world(argv[1]);
    if (!gWorld) {
        fprintf(stderr, "ref_host: load_world failed\n");
        return 1;
    }

    const unsigned int data_texture_width = 2048;   // ray.cpp:326
    scene_shader_data data;
    get_shader_data(gWorld, data, data_texture_width);

    ref_startup_defaults();

    g_out = fopen(argv[2], "wb");
    if (!g_out) {
        perror(argv[2]);
        return 1;
    }
    fwrite("SHRD", 1, 4, g_out);

    scalar_section("triangle_count", gWorld->triangle_count);
    scalar_section("independent_vertices", (double)gWorld->triangles->vertices.size());
    const float center[3] = {gWorld->scene_center.x, gWorld->scene_center.y, gWorld->scene_center.z};
    section("scene_center", center, 3);
    section("scene_extent", &gWorld->scene_extent, 1);

    scalar_section("vertex_count", data.vertex_count);
    scalar_section("vertex_data_rows", data.vertex_data_rows);
    scalar_section("group_count", data.group_count);
    scalar_section("group_data_rows", data.group_data_rows);
    scalar_section("tree_root", data.tree_root);

    // only the populated prefix of each array: the reference leaves the
    // padding up to width*rows uninitialised
    const uint64_t nv = data.vertex_count, ng = (uint64_t)data.group_count;
    section("vertex_positions", data.vertex_positions, 3 * nv);
    section("vertex_normals", data.vertex_normals, 3 * nv);
    section("vertex_colors", data.vertex_colors, 3 * nv);
    section("group_boxmin", data.group_boxmin, 3 * ng);
    section("group_boxmax", data.group_boxmax, 3 * ng);
    section("group_children", data.group_children, 2 * ng);
    section("group_objects", data.group_objects, 2 * ng);
    {
        // group_directions is only written for branches; dump with leaves zeroed
        std::vector<float> dirs(3 * ng, 0.0f);
        for (uint64_t g = 0; g < ng; g++)
            if (data.group_children[2 * g] < 2147483648.0f)
                memcpy(&dirs[3 * g], &data.group_directions[3 * g], 3 * sizeof(float));
        section("group_directions", dirs.data(), 3 * ng);
    }
    const uint64_t table_stride = (uint64_t)data_texture_width * data.group_data_rows;
    for (int code = 0; code < 8; code++) {
        char name[32];
        snprintf(name, sizeof(name), "group_hitmiss_%d", code);
        section(name, data.group_hitmiss + 2 * table_stride * code, 2 * ng);
    }

    // frame parameters: matrices / light / zoom are the reference's own code;
    // the four image-plane values restate ray.cpp:672-683 (GL calls sit between
    // those lines upstream, so they cannot be sliced out and compiled)
    section("fov", &gWorld->cam.fov, 1);
    section("zoom", &zoom, 1);
    section("camera_matrix", gWorld->camera_matrix, 16);
    section("camera_normal_matrix", gWorld->camera_normal_matrix, 16);
    section("object_matrix", gWorld->object_matrix, 16);
    section("object_inverse", gWorld->object_inverse, 16);
    section("object_normal_matrix", gWorld->object_normal_matrix, 16);
    section("object_normal_inverse", gWorld->object_normal_inverse, 16);
    const float light[3] = {light_dir.x, light_dir.y, light_dir.z};
    section("light_dir", light, 3);
    {
        float image_plane_width = 2 * tanf(gWorld->cam.fov / 2.0);
        float aspect = height / (1.0f * width);
        vec4 d(image_plane_width / width, 0, 0, 0.0);
        vec4 right_vector = gWorld->camera_normal_matrix * d;
        d = vec4(0, image_plane_width * aspect / height, 0, 0.0);
        vec4 up_vector = gWorld->camera_normal_matrix * d;
        const float right[3] = {right_vector.x, right_vector.y, right_vector.z};
        const float up[3] = {up_vector.x, up_vector.y, up_vector.z};
        section("image_plane_width", &image_plane_width, 1);
        section("aspect", &aspect, 1);
        section("right", right, 3);
        section("up", up, 3);
    }

    // a second, non-trivial view: rotate the object and the light
    object_rotation[0] = 0.9f;
    object_rotation[1] = 0.26726124f;
    object_rotation[2] = 0.53452248f;
    object_rotation[3] = 0.80178373f;
    light_rotation[0] = 1.1f;
    light_rotation[1] = 0.0f;
    light_rotation[2] = 0.6f;
    light_rotation[3] = 0.8f;
    extern void update_view_params(world_ptr world, float zoom);
    extern void update_light();
    update_view_params(gWorld, zoom * 0.75f);
    update_light();
    section("v2_camera_matrix", gWorld->camera_matrix, 16);
    section("v2_camera_normal_matrix", gWorld->camera_normal_matrix, 16);
    section("v2_object_matrix", gWorld->object_matrix, 16);
    section("v2_object_inverse", gWorld->object_inverse, 16);
    section("v2_object_normal_matrix", gWorld->object_normal_matrix, 16);
    section("v2_object_normal_inverse", gWorld->object_normal_inverse, 16);
    const float light2[3] = {light_dir.x, light_dir.y, light_dir.z};
    section("v2_light_dir", light2, 3);

    fclose(g_out);
    return 0;
}
