// shray_render -- headless counterpart of the reference's `./ray model environment`
// (README.md:11-13, ray.cpp:954-1092): load the model, set up the start-up view, render one
// frame on the GPU and save it the way the 's' key does (color.ppm, ray.cpp:730-787).
//
//   shray_render model.{trisrc,obj} background [-o out.ppm] [-w W -h H] [-m material] [-d diffuse] [-s spp]
//                [-n frames]   animate: drag the trackball a little every frame (object for the first
//                              half, light for the second, cycling the material every 25 frames), render
//                              `frames` frames, print the reference's benchmark histogram (the 'B' key,
//                              ray.cpp:1096-1131: 10 buckets of frame time / fps) and save the last frame
//                [-f prefix]   also dump every frame as raw RGBA float32 to <prefix>NNN.rgba (row 0 = bottom)
//
// background: "r, g, b" floats, "grid", hex "rrggbb" (ray.cpp:1002-1035) or a Radiance .hdr file
// (host/background.cpp; the reference decodes image files through FreeImagePlus).
#include <algorithm>
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
    int width = 512, height = 512, material = 0, diffuse = 0, spp = 1, frames = 1;
    std::string out = "color.ppm", dump_prefix;
    for (int i = 3; i + 1 < argc; i += 2) {
        if (!strcmp(argv[i], "-o")) out = argv[i + 1];
        else if (!strcmp(argv[i], "-w")) width = atoi(argv[i + 1]);
        else if (!strcmp(argv[i], "-h")) height = atoi(argv[i + 1]);
        else if (!strcmp(argv[i], "-m")) material = atoi(argv[i + 1]);
        else if (!strcmp(argv[i], "-d")) diffuse = atoi(argv[i + 1]);
        else if (!strcmp(argv[i], "-s")) spp = atoi(argv[i + 1]);
        else if (!strcmp(argv[i], "-n")) frames = atoi(argv[i + 1]);
        else if (!strcmp(argv[i], "-f")) dump_prefix = argv[i + 1];
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

    // the frame lands in pinned host memory: shray_render then reads it back with one DMA at PCIe speed
    float *rgba = nullptr;
    if (shray_pinned_alloc((size_t)width * height * 16, (void **)&rgba) != SHRAY_OK) {
        fprintf(stderr, "pinned allocation failed: %s\n", shray_last_error());
        return EXIT_FAILURE;
    }
    std::vector<float> frame_seconds;
    for (int frame = 0; frame < std::max(frames, 1); frame++) {
        if (frames > 1) {
            // what a user dragging the mouse does (MotionCallback, ray.cpp:879-918): the object for
            // the first half of the run, the light ('l' key) for the second; 'm' every 25 frames
            float *target = (frame < frames / 2) ? view.object_rotation : view.light_rotation;
            trackball_motion(target, 0.011f, 0.004f, target);
            if (frame % 25 == 24)
                view.which_material = (view.which_material + 1) % material_count;
            make_frame_params(world, view, width, height, &params);
        }
        const auto then = std::chrono::steady_clock::now();
        if (shray_render(scene, &params, width, height, spp, rgba) != SHRAY_OK) {
            fprintf(stderr, "render failed: %s\n", shray_last_error());
            return EXIT_FAILURE;
        }
        frame_seconds.push_back(std::chrono::duration<float>(std::chrono::steady_clock::now() - then).count());
        if (!dump_prefix.empty()) {
            char name[1024];
            snprintf(name, sizeof(name), "%s%03d.rgba", dump_prefix.c_str(), frame);
            FILE *df = fopen(name, "wb");
            if (!df || fwrite(rgba, 16, (size_t)width * height, df) != (size_t)width * height) {
                fprintf(stderr, "cannot write %s\n", name);
                return EXIT_FAILURE;
            }
            fclose(df);
        }
    }
    if (frames <= 1) {
        fprintf(stderr, "%dx%d, %d spp: %.3f ms including the copy to host\n", width, height, spp, frame_seconds[0] * 1e3);
    } else {
        // the reference's benchmark print-out (ray.cpp:1116-1131)
        const float lo = *std::min_element(frame_seconds.begin(), frame_seconds.end());
        const float hi = *std::max_element(frame_seconds.begin(), frame_seconds.end());
        printf("%d frames:\n", frames);
        const int buckets = 10;
        for (int b = 0; b < buckets; b++) {
            const float start = lo + (hi - lo) * b / buckets, end = lo + (hi - lo) * (b + 1) / buckets;
            int count = 0;
            for (float d : frame_seconds)
                if (d >= start && (d < end || (b == buckets - 1 && d <= end)))
                    count++;
            printf("%.2f to %.2f ms, %.2f fps : %d\n", start * 1000.0, end * 1000.0, 1 / ((start + end) / 2.0), count);
        }
    }

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
    shray_pinned_free(rgba);
    shray_scene_destroy(scene);
    return EXIT_SUCCESS;
}
