// shray_render -- headless counterpart of the reference's `./ray model environment`
// (README.md:11-13, ray.cpp:954-1092): load the model, set up the start-up view, render one
// frame on the GPU and save it the way the 's' key does (color.ppm, ray.cpp:730-787).
//
//   shray_render model.{trisrc,obj} background [-o out.ppm] [-w W -h H] [-m material] [-d diffuse] [-s spp]
//
// background: "r, g, b" floats, "grid", hex "rrggbb" (ray.cpp:1002-1035) or a Radiance .hdr file
// (host/background.cpp; the reference decodes image files through FreeImagePlus).
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "background.h"
#include "frame-params.h"
#include "shader_ray_hip.h"
#include "world.h"

int main(int argc, char **argv)
{
    if (argc < 3) {
        fprintf(stderr, "usage: %s inputfilename backgroundcolorspec [-o out.ppm] [-w W] [-h H] [-m material] [-d diffuse] [-s spp]\n"
                        "background color can be floats as \"r, g, b\", \"grid\", or hex as \"rrggbb\"\n", argv[0]);
        return EXIT_FAILURE;
    }
    int width = 512, height = 512, material = 0, diffuse = 0, spp = 1;
    std::string out = "color.ppm";
    for (int i = 3; i + 1 < argc; i += 2) {
        if (!strcmp(argv[i], "-o")) out = argv[i + 1];
        else if (!strcmp(argv[i], "-w")) width = atoi(argv[i + 1]);
        else if (!strcmp(argv[i], "-h")) height = atoi(argv[i + 1]);
        else if (!strcmp(argv[i], "-m")) material = atoi(argv[i + 1]);
        else if (!strcmp(argv[i], "-d")) diffuse = atoi(argv[i + 1]);
        else if (!strcmp(argv[i], "-s")) spp = atoi(argv[i + 1]);
    }

    world_ptr world = load_world(argv[1]);
    if (!world) {
        fprintf(stderr, "Cannot set up world.\n");
        return EXIT_FAILURE;
    }
    float2Dimage background;
    if (!load_background(argv[2], background))
        return EXIT_FAILURE;
    const std::vector<float> &env = background.pixels;
    const int env_w = background.width, env_h = background.height;

    scene_shader_data data;
    get_shader_data(world, data, 2048);
    shray_scene_desc desc;
    memset(&desc, 0, sizeof(desc));
    desc.struct_size = sizeof(desc);
    desc.data_texture_width = 2048;
    desc.vertex_count = data.vertex_count;
    desc.vertex_data_rows = data.vertex_data_rows;
    desc.vertex_positions = data.vertex_positions;
    desc.vertex_normals = data.vertex_normals;
    desc.group_count = data.group_count;
    desc.group_data_rows = data.group_data_rows;
    desc.tree_root = data.tree_root;
    desc.group_boxmin = data.group_boxmin;
    desc.group_boxmax = data.group_boxmax;
    desc.group_hitmiss = data.group_hitmiss;
    desc.group_objects = data.group_objects;

    shray_scene *scene = nullptr;
    if (shray_scene_create(&desc, &scene) != SHRAY_OK || shray_scene_set_environment(scene, env.data(), env_w, env_h) != SHRAY_OK) {
        fprintf(stderr, "GPU setup failed: %s\n", shray_last_error());
        return EXIT_FAILURE;
    }

    view_state view = default_view_state(world);
    view.which_material = material;
    view.which_diffuse_color = diffuse;
    shray_frame_params params;
    make_frame_params(world, view, width, height, &params);

    std::vector<float> rgba((size_t)width * height * 4);
    const auto then = std::chrono::steady_clock::now();
    if (shray_render(scene, &params, width, height, spp, rgba.data()) != SHRAY_OK) {
        fprintf(stderr, "render failed: %s\n", shray_last_error());
        return EXIT_FAILURE;
    }
    const double seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - then).count();
    fprintf(stderr, "%dx%d, %d spp: %.3f ms including the copy to host\n", width, height, spp, seconds * 1e3);

    FILE *fp = fopen(out.c_str(), "wb");
    if (!fp) {
        fprintf(stderr, "snapshot: couldn't open \"%s\".\n", out.c_str());
        return EXIT_FAILURE;
    }
    fprintf(fp, "P6 %d %d 255\n", width, height);
    std::vector<unsigned char> row((size_t)width * 3);
    for (int y = height - 1; y >= 0; y--) {   // top row first, like the reference's flipped glReadPixels dump
        for (int x = 0; x < width; x++)
            for (int c = 0; c < 3; c++) {
                const float v = rgba[4 * ((size_t)y * width + x) + c];
                row[3 * x + c] = (unsigned char)(v <= 0 ? 0 : (v >= 1 ? 255 : (int)(v * 255.0f + 0.5f)));
            }
        fwrite(row.data(), 3, width, fp);
    }
    fclose(fp);
    shray_scene_destroy(scene);
    return EXIT_SUCCESS;
}
