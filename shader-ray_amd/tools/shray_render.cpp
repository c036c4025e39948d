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
//                [-g N]        N GPUs: one thread per GPU, each with a replica of the scene; the frames of the
//                              animation are rendered in steps of N (every rank its interleaved tiles of all N, one
//                              launch), exchanged over xGMI (RCCL) and assembled -- shader_ray_dist.h.
//                [-r mode]     with -g: root0 (every frame assembled on GPU 0) or rotate (frame f on GPU f % N; default)
//                [-t name]     with -g: rccl (default) or loopback (all ranks share GPU 0: rehearsal on a one-GPU box)
//                [-b where]    the BVH: host (make_bvh, bvh.cpp:288-358; default), gpu (shray_bvh_build_device: the same tree,
//                              built on the device -- load_triangles, the device build, adopt_tree) or device (round 6: the build, the
//                              flattening and scene creation all on the device; no group tree on the host, nothing downloaded)
//
// background: "r, g, b" floats, "grid", hex "rrggbb" (ray.cpp:1002-1035) or a Radiance .hdr file
// (host/background.cpp; the reference decodes image files through FreeImagePlus).
#include <hip/hip_runtime_api.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "background.h"
#include "bvh.h"
#include "frame-params.h"
#include "shader_ray_dist.h"
#include "shader_ray_hip.h"
#include "world.h"
#include "host-log.h"

namespace {

// -g N: the frame loop of ray.cpp:1096-1131 on N GPUs.  Every rank (thread) renders its tiles of `world` consecutive
// frames per step; with rotating roots frame f of a step ends up, whole, on rank f % world, which copies it to its
// slot of `frames_out` (pinned host memory, the animation's frames back to back) when `keep_frames` is set.
struct multi_gpu_job {
    int world = 1, width = 0, height = 0, spp = 1, frames = 1;
    int root_mode = SHRAY_DIST_ROTATE, transport = SHRAY_DIST_RCCL;
    const shray_scene_desc *desc = nullptr;
    const float *env = nullptr;
    int env_w = 0, env_h = 0;
    const std::vector<shray_frame_params> *params = nullptr;
    float *frames_out = nullptr;      // frames * width * height * 4 floats, or nullptr: keep only the last frame
    float *last_frame = nullptr;      // width * height * 4 floats
    unsigned char unique_id[SHRAY_DIST_UNIQUE_ID_BYTES];
    shray_dist_hub *hub = nullptr;
    std::vector<std::string> errors;
    std::vector<double> seconds;      // per rank: the loop's wall time
};

constexpr int kSets = 4;   // steps in flight per rank

void run_rank(multi_gpu_job *job, int rank)
{
    auto bail = [&](const char *what, const char *message) {
        char buf[768];
        snprintf(buf, sizeof(buf), "rank %d: %s: %s", rank, what, message);
        job->errors[(size_t)rank] = buf;
    };
    const int device = job->transport == SHRAY_DIST_LOOPBACK ? 0 : rank;
    shray_scene *scene = nullptr;
    shray_dist *dist = nullptr;
    if (shray_set_device(device) != SHRAY_OK || shray_scene_create(job->desc, &scene) != SHRAY_OK ||
        shray_scene_set_environment(scene, job->env, job->env_w, job->env_h) != SHRAY_OK)
        return bail("scene replica", shray_last_error());
    shray_dist_config cfg;
    memset(&cfg, 0, sizeof(cfg));
    cfg.struct_size = sizeof(cfg);
    cfg.rank = rank;
    cfg.world = job->world;
    cfg.width = job->width;
    cfg.height = job->height;
    cfg.spp = job->spp;
    cfg.max_frames = std::min(4 * job->world, SHRAY_MAX_BATCH);   // four frames' worth of rays per launch and rank
    cfg.root_mode = job->root_mode;
    cfg.rgb_wire = 1;
    cfg.transport = job->transport;
    cfg.buffer_sets = kSets;
    const void *arg = job->transport == SHRAY_DIST_LOOPBACK ? (const void *)job->hub : (const void *)job->unique_id;
    if (shray_dist_create(scene, &cfg, arg, &dist) != SHRAY_OK) {
        bail("shray_dist_create", shray_dist_last_error());
        shray_scene_destroy(scene);
        return;
    }
    hipStream_t streams[kSets];
    for (hipStream_t &st : streams)
        if (hipStreamCreateWithFlags(&st, hipStreamNonBlocking) != hipSuccess)
            return bail("hipStreamCreate", "failed");
    const size_t frame_floats = (size_t)job->width * job->height * 4;
    const auto then = std::chrono::steady_clock::now();
    // kSets steps in flight on as many streams and buffer sets: the exchange of one runs under the render of the next
    int step = 0;
    auto collect = [&](int which_step, int first_frame_of_step, int count) {
        const int set = which_step % kSets;
        int assembled = 0, first = 0, stride = 1;
        void *d_rgba = nullptr;
        if (shray_dist_output(dist, set, count, &assembled, &first, &stride, &d_rgba) != SHRAY_OK)
            return bail("shray_dist_output", shray_dist_last_error());
        for (int k = 0; k < assembled; k++) {
            const int frame = first_frame_of_step + first + k * stride;
            float *dst = job->frames_out ? job->frames_out + (size_t)frame * frame_floats : (frame == job->frames - 1 ? job->last_frame : nullptr);
            if (dst && hipMemcpyAsync(dst, (const char *)d_rgba + (size_t)k * frame_floats * 4, frame_floats * 4, hipMemcpyDeviceToHost,
                                      streams[set]) != hipSuccess)
                return bail("hipMemcpyAsync", "frame readback failed");
        }
    };
    int pending_first = -1, pending_count = 0;
    for (int f0 = 0; f0 < job->frames; f0 += cfg.max_frames, step++) {
        const int count = std::min(cfg.max_frames, job->frames - f0);
        if (step >= kSets)
            (void)hipStreamSynchronize(streams[step % kSets]);   // the readbacks of the step that used this buffer set
        if (shray_dist_step(dist, step % kSets, job->params->data() + f0, count, streams[step % kSets]) != SHRAY_OK) {
            bail("shray_dist_step", shray_dist_last_error());
            break;
        }
        if (pending_count)
            collect(step - 1, pending_first, pending_count);
        pending_first = f0;
        pending_count = count;
    }
    if (pending_count && job->errors[(size_t)rank].empty())
        collect(step - 1, pending_first, pending_count);
    for (hipStream_t st : streams)
        (void)hipStreamSynchronize(st);
    job->seconds[(size_t)rank] = std::chrono::duration<double>(std::chrono::steady_clock::now() - then).count();
    for (hipStream_t st : streams)
        (void)hipStreamDestroy(st);
    shray_dist_destroy(dist);
    shray_scene_destroy(scene);
}

}   // namespace

int main(int argc, char **argv)
{
    if (argc < 3) {
        fprintf(stderr, "usage: %s inputfilename backgroundcolorspec [-o out.ppm] [-w W] [-h H] [-m material] [-d diffuse] [-s spp]\n"
                        "background color can be floats as \"r, g, b\", \"grid\", or hex as \"rrggbb\"\n", argv[0]);
        return EXIT_FAILURE;
    }
    int width = 512, height = 512, material = 0, diffuse = 0, spp = 1, frames = 1, gpus = 1;
    std::string out = "color.ppm", dump_prefix, root_mode = "rotate", transport = "rccl", bvh_where = "host";
    for (int i = 3; i + 1 < argc; i += 2) {
        if (!strcmp(argv[i], "-o")) out = argv[i + 1];
        else if (!strcmp(argv[i], "-w")) width = atoi(argv[i + 1]);
        else if (!strcmp(argv[i], "-h")) height = atoi(argv[i + 1]);
        else if (!strcmp(argv[i], "-m")) material = atoi(argv[i + 1]);
        else if (!strcmp(argv[i], "-d")) diffuse = atoi(argv[i + 1]);
        else if (!strcmp(argv[i], "-s")) spp = atoi(argv[i + 1]);
        else if (!strcmp(argv[i], "-n")) frames = atoi(argv[i + 1]);
        else if (!strcmp(argv[i], "-f")) dump_prefix = argv[i + 1];
        else if (!strcmp(argv[i], "-g")) gpus = atoi(argv[i + 1]);
        else if (!strcmp(argv[i], "-r")) root_mode = argv[i + 1];
        else if (!strcmp(argv[i], "-t")) transport = argv[i + 1];
        else if (!strcmp(argv[i], "-b")) bvh_where = argv[i + 1];
    }

    world_ptr world;
    shray_scene *resident = nullptr;      // -b device: the scene, made without the tree ever leaving the device
    double device_pipeline_seconds[3] = {0, 0, 0};
    if (bvh_where == "device") {
        // load_world (world.cpp:46-134) + get_shader_data (world.cpp:298-347) + the texture upload (ray.cpp:470-497) with everything
        // behind the parse on the device: shray_bvh_build_device -> shray_flatten_device_tree -> shray_scene_create_from_device.  The
        // `world` keeps its triangles, centre and extent (what the frame parameters need) and no group tree.
        if (gpus > 1) {
            fprintf(stderr, "-b device renders on one GPU (the ranks of -g N each create their scene from the host arrays)\n");
            return EXIT_FAILURE;
        }
        world = load_triangles(argv[1]);
        if (world) {
            const triangle_set &mesh = *world->triangles;
            std::vector<int32_t> corners(3 * mesh.triangles.size());
            for (size_t t = 0; t < mesh.triangles.size(); t++)
                for (int c = 0; c < 3; c++)
                    corners[3 * t + c] = mesh.triangles[t].i[c];
            const bvh_build_options &bo = bvh_options();
            shray_bvh_options options = {sizeof(shray_bvh_options), bo.max_depth, (int32_t)bo.leaf_max, bo.sah_ctrav, bo.sah_cisec};
            shray_device_tree *built = nullptr;
            shray_device_flat *flat = nullptr;
            auto then = std::chrono::steady_clock::now();
            auto lap = [&then]() {
                const auto now = std::chrono::steady_clock::now();
                const double s = std::chrono::duration<double>(now - then).count();
                then = now;
                return s;
            };
            bool ok = shray_set_device(0) == SHRAY_OK &&
                      shray_bvh_build_device(corners.data(), (int32_t)mesh.triangles.size(), &mesh.vertices[0].v.x, (int32_t)mesh.vertices.size(), 9,
                                             &options, &built) == SHRAY_OK;
            device_pipeline_seconds[0] = lap();
            ok = ok && shray_flatten_device_tree(built, 2048, &flat) == SHRAY_OK;
            device_pipeline_seconds[1] = lap();
            ok = ok && shray_scene_create_from_device(built, flat, &resident) == SHRAY_OK;
            device_pipeline_seconds[2] = lap();
            if (!ok) {
                fprintf(stderr, "The device-resident scene pipeline failed: %s\n", shray_last_error());
                return EXIT_FAILURE;
            }
            shray_bvh_stats stats;
            shray_device_tree_stats(built, &stats);
            world->build_seconds = device_pipeline_seconds[0];
            host_info("BVH: %f seconds (on the GPU: %f; %d nodes, %d leaves, deepest level %d)\n", world->build_seconds, stats.device_seconds,
                      stats.node_count, stats.leaf_count, stats.max_level);
            shray_device_flat_destroy(flat);
            shray_device_tree_destroy(built);
        }
    } else if (bvh_where == "gpu") {
        // load_world (world.cpp:46-134) with make_bvh replaced by the device build
        world = load_triangles(argv[1]);
        if (world) {
            const auto build_began = std::chrono::steady_clock::now();
            const triangle_set &mesh = *world->triangles;
            std::vector<int32_t> corners(3 * mesh.triangles.size());
            for (size_t t = 0; t < mesh.triangles.size(); t++)
                for (int c = 0; c < 3; c++)
                    corners[3 * t + c] = mesh.triangles[t].i[c];
            const bvh_build_options &bo = bvh_options();
            shray_bvh_options options = {sizeof(shray_bvh_options), bo.max_depth, (int32_t)bo.leaf_max, bo.sah_ctrav, bo.sah_cisec};
            shray_device_tree *built = nullptr;
            shray_tree_desc tree;
            const int32_t *order = nullptr;
            if (shray_bvh_build_device(corners.data(), (int32_t)mesh.triangles.size(), &mesh.vertices[0].v.x, (int32_t)mesh.vertices.size(), 9,
                                       &options, &built) != SHRAY_OK ||
                shray_device_tree_download(built, &tree, &order) != SHRAY_OK ||
                !adopt_tree(world, tree.node_count, tree.node_negative, tree.node_positive, tree.node_box, tree.node_direction, tree.node_start,
                            tree.node_triangles, order, tree.triangle_count)) {
                fprintf(stderr, "The BVH build on the GPU failed: %s\n", shray_last_error());
                return EXIT_FAILURE;
            }
            shray_bvh_stats stats;
            shray_device_tree_stats(built, &stats);
            shray_device_tree_destroy(built);
            world->build_seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - build_began).count();
            host_info("BVH: %f seconds (on the GPU: %f; %d nodes, %d leaves, deepest level %d)\n", world->build_seconds, stats.device_seconds,
                      stats.node_count, stats.leaf_count, stats.max_level);
        }
    } else
        world = load_world(argv[1]);
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
    const auto flatten_began = std::chrono::steady_clock::now();
    if (!resident)
        get_shader_data(world, data, 2048);
    const double flatten_seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - flatten_began).count();
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

    view_state view = default_view_state(world);
    view.which_material = material;
    view.which_diffuse_color = diffuse;
    shray_frame_params params;
    make_frame_params(world, view, width, height, &params);
    // the animation: what a user dragging the mouse does (MotionCallback, ray.cpp:879-918): the object for the first
    // half of the run, the light ('l' key) for the second; 'm' every 25 frames
    auto advance_view = [&](int frame) {
        if (frames <= 1)
            return;
        float *target = (frame < frames / 2) ? view.object_rotation : view.light_rotation;
        trackball_motion(target, 0.011f, 0.004f, target);
        if (frame % 25 == 24)
            view.which_material = (view.which_material + 1) % material_count;
        make_frame_params(world, view, width, height, &params);
    };
    auto dump_frame = [&](int frame, const float *pixels) {
        char name[1024];
        snprintf(name, sizeof(name), "%s%03d.rgba", dump_prefix.c_str(), frame);
        FILE *df = fopen(name, "wb");
        if (!df || fwrite(pixels, 16, (size_t)width * height, df) != (size_t)width * height) {
            fprintf(stderr, "cannot write %s\n", name);
            return false;
        }
        fclose(df);
        return true;
    };

    // the frame lands in pinned host memory: the readback is then one DMA at PCIe speed
    float *rgba = nullptr;
    if (shray_pinned_alloc((size_t)width * height * 16, (void **)&rgba) != SHRAY_OK) {
        fprintf(stderr, "pinned allocation failed: %s\n", shray_last_error());
        return EXIT_FAILURE;
    }
    std::vector<float> frame_seconds;
    shray_scene *scene = nullptr;
    if (gpus > 1 || transport == "loopback") {
        multi_gpu_job job;
        job.world = gpus;
        job.width = width;
        job.height = height;
        job.spp = spp;
        job.frames = std::max(frames, 1);
        job.root_mode = root_mode == "root0" ? SHRAY_DIST_ROOT0 : SHRAY_DIST_ROTATE;
        job.transport = transport == "loopback" ? SHRAY_DIST_LOOPBACK : SHRAY_DIST_RCCL;
        job.desc = &desc;
        job.env = env.data();
        job.env_w = env_w;
        job.env_h = env_h;
        std::vector<shray_frame_params> all;
        for (int frame = 0; frame < job.frames; frame++) {
            advance_view(frame);
            all.push_back(params);
        }
        job.params = &all;
        job.last_frame = rgba;
        if (!dump_prefix.empty() &&
            shray_pinned_alloc((size_t)job.frames * width * height * 16, (void **)&job.frames_out) != SHRAY_OK) {
            fprintf(stderr, "pinned allocation failed: %s\n", shray_last_error());
            return EXIT_FAILURE;
        }
        int devices = 0;
        if (shray_device_count(&devices) != SHRAY_OK || (job.transport == SHRAY_DIST_RCCL && devices < gpus)) {
            fprintf(stderr, "-g %d over RCCL needs %d GPUs, %d visible (-t loopback shares GPU 0)\n", gpus, gpus, devices);
            return EXIT_FAILURE;
        }
        if (job.transport == SHRAY_DIST_RCCL ? shray_dist_unique_id(job.unique_id) != SHRAY_OK
                                             : shray_dist_hub_create(gpus, &job.hub) != SHRAY_OK) {
            fprintf(stderr, "transport setup failed: %s\n", shray_dist_last_error());
            return EXIT_FAILURE;
        }
        job.errors.assign((size_t)gpus, std::string());
        job.seconds.assign((size_t)gpus, 0.0);
        std::vector<std::thread> ranks;
        for (int r = 0; r < gpus; r++)
            ranks.emplace_back(run_rank, &job, r);
        for (std::thread &t : ranks)
            t.join();
        if (job.hub)
            shray_dist_hub_destroy(job.hub);
        for (const std::string &e : job.errors)
            if (!e.empty()) {
                fprintf(stderr, "%s\n", e.c_str());
                return EXIT_FAILURE;
            }
        const double slowest = *std::max_element(job.seconds.begin(), job.seconds.end());
        printf("%d frames on %d GPUs (%s, %s): %.3f ms per frame, %.1f Mrays/s\n", job.frames, gpus,
               job.root_mode == SHRAY_DIST_ROOT0 ? "root0" : "rotating roots", job.transport == SHRAY_DIST_RCCL ? "rccl" : "loopback",
               slowest / job.frames * 1e3, (double)width * height * spp * job.frames / slowest / 1e6);
        if (job.frames_out) {
            for (int frame = 0; frame < job.frames; frame++)
                if (!dump_frame(frame, job.frames_out + (size_t)frame * width * height * 4))
                    return EXIT_FAILURE;
            memcpy(rgba, job.frames_out + (size_t)(job.frames - 1) * width * height * 4, (size_t)width * height * 16);
            shray_pinned_free(job.frames_out);
        }
        frames = 0;   // no per-frame histogram: the loop's rate is printed above
    } else {
        const auto create_began = std::chrono::steady_clock::now();
        if (resident)
            scene = resident;
        if ((!resident && shray_scene_create(&desc, &scene) != SHRAY_OK) || shray_scene_set_environment(scene, env.data(), env_w, env_h) != SHRAY_OK) {
            fprintf(stderr, "GPU setup failed: %s\n", shray_last_error());
            return EXIT_FAILURE;
        }
        if (resident)
            fprintf(stderr, "device-resident pipeline: BVH %.4f s, flatten %.4f s, scene %.4f s -- the tree never left the device\n",
                    device_pipeline_seconds[0], device_pipeline_seconds[1], device_pipeline_seconds[2]);
        // scene turnaround, file to resident scene (what the reference prints piecewise: world.cpp:93-116, ray.cpp:470-510)
        fprintf(stderr, "scene turnaround: parse %.3f s, centre + extent %.3f s, BVH %.3f s, flatten %.3f s, validate + repack + upload "
                "(with the environment) %.3f s; %d triangles, %d threads\n", world->parse_seconds, world->extent_seconds,
                world->build_seconds, flatten_seconds,
                std::chrono::duration<double>(std::chrono::steady_clock::now() - create_began).count(), world->triangle_count,
                host_load_threads());
        for (int frame = 0; frame < std::max(frames, 1); frame++) {
            advance_view(frame);
            const auto then = std::chrono::steady_clock::now();
            if (shray_render(scene, &params, width, height, spp, rgba) != SHRAY_OK) {
                fprintf(stderr, "render failed: %s\n", shray_last_error());
                return EXIT_FAILURE;
            }
            frame_seconds.push_back(std::chrono::duration<float>(std::chrono::steady_clock::now() - then).count());
            if (!dump_prefix.empty() && !dump_frame(frame, rgba))
                return EXIT_FAILURE;
        }
    }
    if (frames == 0) {
        // (multi-GPU run: reported above)
    } else if (frames <= 1) {
        fprintf(stderr, "%dx%d, %d spp: %.3f ms including the copy to host\n", width, height, spp, frame_seconds[0] * 1e3);
    } else {
        // the reference's benchmark print-out (ray.cpp:1116-1131)
        const float lo = *std::min_element(frame_seconds.begin(), frame_seconds.end());
        const float hi = *std::max_element(frame_seconds.begin(), frame_seconds.end());
        printf("%d frames:\n", frames);
        const int buckets = 10;
        for (int b = 0; b < buckets; b++) {
            // (the last bucket ends at the slowest frame itself: lo + (hi - lo) need not round back to hi)
            const float start = lo + (hi - lo) * b / buckets, end = b == buckets - 1 ? hi : lo + (hi - lo) * (b + 1) / buckets;
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
