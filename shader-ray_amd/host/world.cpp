// world.cpp -- scene loading and tree flattening.
//
// load_world        : reference world.cpp:46-134
// get_shader_data   : reference world.cpp:136-347.  Layout rules restated:
//   * nodes are numbered in-order: negative subtree, the node, positive subtree
//     (world.cpp:145-177), so the root sits mid-array;
//   * positions / normals / colours are expanded to three float3 per triangle
//     in post-build triangle order (world.cpp:304-317);
//   * for each of the 8 ray-direction sign codes (bit0 = +x, bit1 = +y,
//     bit2 = +z) every node gets a (hit, miss) link pair (world.cpp:231-288):
//     a branch's hit link is its near child (the POSITIVE child when the coded
//     direction has a negative component along the split axis, else the
//     negative child) and its miss link is whatever subtree is next on the
//     traversal stack; a leaf links to that next subtree on both hit and miss;
//     no next subtree = 0x7fffffff, which float32 stores as 2147483648.
// The walk here is iterative and writes the link tables directly.
#include "world.h"

#include <cerrno>
#include <chrono>
#include <thread>
#include <vector>
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <cstring>
#include <vector>

#include "bvh.h"
#include "host-log.h"
#include "obj-support.h"
#include "trisrc-support.h"

int host_load_threads()
{
    static const int threads = [] {
        if (const char *s = getenv("SHRAY_LOAD_THREADS"))
            return std::min(256, std::max(1, atoi(s)));   // (synthesized normals: every thread walks every face)
        return (int)std::max(1u, std::min(std::thread::hardware_concurrency(), 32u));
    }();
    return threads;
}

namespace {

const float kStopLink = (float)0x7fffffffU;   // == 2147483648.0f
const int kDirectionCodes = 8;
const int kLinkStackCapacity = 64;            // world.cpp:228

double seconds_since(std::chrono::steady_clock::time_point then)
{
    return std::chrono::duration<double>(std::chrono::steady_clock::now() - then).count();
}

float *zeroed(size_t n) { return new float[n](); }

// In-order numbering without recursion; returns the node count.
int number_in_order(group *root)
{
    int next = 0;
    std::vector<group *> spine;
    group *g = root;
    while (g || !spine.empty()) {
        while (g) {
            spine.push_back(g);
            g = g->negative;
        }
        g = spine.back();
        spine.pop_back();
        g->my_index = next++;
        g = g->positive;
    }
    return next;
}

template <class F>
void for_each_node(group *root, F &&fn)
{
    std::vector<group *> todo(1, root);
    while (!todo.empty()) {
        group *g = todo.back();
        todo.pop_back();
        fn(g);
        if (g->negative) {
            todo.push_back(g->positive);
            todo.push_back(g->negative);
        }
    }
}

// Threads the tree for one direction code and writes table `code`.
bool thread_direction(group *root, int code, float *table /* float2 per node */)
{
    const float sign[3] = {(code & 1) ? 1.0f : -1.0f, (code & 2) ? 1.0f : -1.0f, (code & 4) ? 1.0f : -1.0f};
    group *stack[kLinkStackCapacity];
    int depth = 0;
    group *g = root;
    while (g) {
        group *next_subtree = depth ? stack[depth - 1] : nullptr;
        group *hit;
        if (g->is_leaf()) {
            hit = next_subtree;
            g->dirhit[code] = g->dirmiss[code] = next_subtree;
            if (depth)
                depth--;
        } else {
            const float along = sign[0] * g->D.x + sign[1] * g->D.y + sign[2] * g->D.z;
            group *near_child = (along < 0) ? g->positive : g->negative;
            group *far_child = (along < 0) ? g->negative : g->positive;
            g->dirhit[code] = near_child;
            g->dirmiss[code] = next_subtree;
            if (depth >= kLinkStackCapacity)
                return false;
            stack[depth++] = far_child;
            hit = near_child;
        }
        float *link = table + 2 * (size_t)g->my_index;
        link[0] = g->dirhit[code] ? (float)g->dirhit[code]->my_index : kStopLink;
        link[1] = g->dirmiss[code] ? (float)g->dirmiss[code]->my_index : kStopLink;
        g = hit;
    }
    return true;
}

}   // namespace

world::world() : triangle_count(0), root(nullptr), scene_center(0.0f), scene_extent(0), xsub(1), ysub(1)
{
    cam.fov = 0;
    mat4_make_identity(camera_matrix);
    mat4_make_identity(camera_normal_matrix);
    mat4_make_identity(object_matrix);
    mat4_make_identity(object_inverse);
    mat4_make_identity(object_normal_matrix);
    mat4_make_identity(object_normal_inverse);
}

world::~world() { delete root; }

world_ptr load_triangles(const std::string &filename)
{
    auto w = std::make_shared<world>();
    w->triangles = std::make_shared<triangle_set>();

    const size_t dot_at = filename.find_last_of('.');
    const std::string extension = (dot_at == std::string::npos) ? filename : filename.substr(dot_at + 1);

    auto then = std::chrono::steady_clock::now();
    if (extension == "trisrc") {
        FILE *fp = fopen(filename.c_str(), "r");
        if (!fp) {
            fprintf(stderr, "Cannot open \"%s\" for input, errno %d\n", filename.c_str(), errno);
            return nullptr;
        }
        const bool ok = ParseTriSrc(fp, w->triangles);
        fclose(fp);
        if (!ok) {
            fprintf(stderr, "Couldn't parse triangles from file.\n");
            return nullptr;
        }
    } else if (extension == "obj") {
        Obj obj;
        errno = 0;
        if (!obj.load_object_from_file(filename)) {
            // the reference prints its "Cannot open" line for every failure of the OBJ reader (world.cpp:83-86); a file that
            // opened and did not parse says so here ("errno 0" for a parse failure)
            if (errno != 0)
                fprintf(stderr, "Cannot open \"%s\" for input, errno %d\n", filename.c_str(), errno);
            else
                fprintf(stderr, "Couldn't parse \"%s\" as an OBJ file.\n", filename.c_str());
            return nullptr;
        }
        if (!obj.fill_triangle_set(w->triangles)) {
            fprintf(stderr, "Couldn't parse triangles from file.\n");
            return nullptr;
        }
    } else {
        fprintf(stderr, "This program doesn't know how to load a file with extension %s\n", extension.c_str());
        return nullptr;
    }
    w->triangles->finish();
    w->parse_seconds = seconds_since(then);
    host_info("Parsing: %f seconds\n", w->parse_seconds);

    triangle_set &mesh = *w->triangles;
    w->triangle_count = (int)mesh.triangles.size();
    host_info("%d triangles, %zu independent vertices\n", w->triangle_count, mesh.vertices.size());

    then = std::chrono::steady_clock::now();
    w->scene_center = mesh.box.center();
    float farthest_squared = 0;
    for (const indexed_triangle &t : mesh.triangles) {
        for (int corner = 0; corner < 3; corner++) {
            const vec3 to_center = w->scene_center - mesh.vertices[t.i[corner]].v;
            farthest_squared = std::max(farthest_squared, dot(to_center, to_center));
        }
    }
    w->scene_extent = sqrtf(farthest_squared) * 2;
    w->extent_seconds = seconds_since(then);
    host_info("Finding scene center and extent: %f seconds\n", w->extent_seconds);
    return w;
}

world_ptr load_world(const std::string &filename)
{
    world_ptr w = load_triangles(filename);
    if (!w)
        return nullptr;
    auto then = std::chrono::steady_clock::now();
    reset_bvh_stats();
    w->root = make_bvh(w->triangles, 0, (unsigned int)w->triangle_count);
    w->build_seconds = seconds_since(then);
    host_info("BVH: %f seconds\n", w->build_seconds);
    if (bvh_options().verbose && !g_host_quiet)
        print_bvh_stats();
    return w;
}

// A tree built elsewhere (the GPU build, shray_bvh_build_device) takes make_bvh's place: the triangles go into the order the
// build left them in, the group tree (group.h:22-40) is made from the pre-order arrays.  What comes out is what make_bvh would
// have left: leaves compute their boxes from their triangles' corners (group.cpp), branches keep the build's vertex box.
bool adopt_tree(const world_ptr &w, int node_count, const int *negative, const int *positive, const float *box, const float *direction,
                const int *start, const int *triangles, const int *triangle_order, int triangle_count)
{
    if (!w || !w->triangles || w->root || node_count <= 0 || triangle_count != (int)w->triangles->triangles.size())
        return false;
    if (!negative || !positive || !box || !direction || !start || !triangles || !triangle_order)
        return false;
    triangle_set &mesh = *w->triangles;
    // every triangle once, every leaf range inside the array, children after their parent (pre-order)
    std::vector<char> seen((size_t)triangle_count, 0);
    for (int k = 0; k < triangle_count; k++) {
        if (triangle_order[k] < 0 || triangle_order[k] >= triangle_count || seen[(size_t)triangle_order[k]])
            return false;
        seen[(size_t)triangle_order[k]] = 1;
    }
    int covered = 0;   // in pre-order the leaves follow each other through the triangle array: no gap, no overlap
    for (int g = 0; g < node_count; g++) {
        const bool leaf = negative[g] < 0;
        if (leaf ? (positive[g] >= 0 || start[g] != covered || triangles[g] < 0 || start[g] + triangles[g] > triangle_count)
                 : (negative[g] != g + 1 || positive[g] <= negative[g] || positive[g] >= node_count))
            return false;
        if (leaf)
            covered += triangles[g];
    }
    if (covered != triangle_count)
        return false;
    // ... and the arrays ARE a pre-order binary tree: a branch's positive child starts right behind its negative child's
    // subtree, and the root's subtree is the whole array -- so that every node has exactly one parent (a node
    // named by two parents would be deleted twice).  Sizes from the back: children come after their parent.
    {
        std::vector<int> size((size_t)node_count, 1);
        for (int g = node_count - 1; g >= 0; g--) {
            if (negative[g] < 0)
                continue;
            const int n = negative[g], p = positive[g];
            if (p != n + size[(size_t)n])
                return false;
            size[(size_t)g] = 1 + size[(size_t)n] + size[(size_t)p];
        }
        if (size[0] != node_count)
            return false;
    }
    std::vector<indexed_triangle> ordered;
    ordered.reserve((size_t)triangle_count);
    for (int k = 0; k < triangle_count; k++)
        ordered.push_back(mesh.triangles[(size_t)triangle_order[k]]);
    mesh.triangles.swap(ordered);
    // children before parents: pre-order backwards
    std::vector<group *> made((size_t)node_count, nullptr);
    for (int g = node_count - 1; g >= 0; g--) {
        if (negative[g] < 0)
            made[(size_t)g] = new group(w->triangles, start[g], (unsigned int)triangles[g]);
        else
            made[(size_t)g] = new group(w->triangles, made[(size_t)negative[g]], made[(size_t)positive[g]],
                                        vec3(direction[3 * g], direction[3 * g + 1], direction[3 * g + 2]),
                                        box3d(vec3(box[6 * g], box[6 * g + 1], box[6 * g + 2]), vec3(box[6 * g + 3], box[6 * g + 4], box[6 * g + 5])));
    }
    w->root = made[0];
    return true;
}

scene_shader_data::scene_shader_data()
    : vertex_count(0), vertex_data_rows(0), vertex_positions(nullptr), vertex_colors(nullptr), vertex_normals(nullptr),
      group_count(0), group_data_rows(0), tree_root(0), group_boxmin(nullptr), group_boxmax(nullptr),
      group_directions(nullptr), group_children(nullptr), group_hitmiss(nullptr), group_objects(nullptr)
{
}

scene_shader_data::~scene_shader_data()
{
    float *owned[] = {vertex_positions, vertex_colors, vertex_normals, group_boxmin, group_boxmax,
                      group_directions, group_children, group_hitmiss, group_objects};
    for (float *p : owned)
        delete[] p;
}

void get_shader_data(world_ptr w, scene_shader_data &data, unsigned int width)
{
    auto then = std::chrono::steady_clock::now();
    const triangle_set &mesh = *w->triangles;

    // ---- per-triangle vertex attributes ----
    data.vertex_count = (unsigned int)mesh.triangles.size() * 3;
    data.vertex_data_rows = (data.vertex_count + width - 1) / width;
    const size_t vertex_texels = (size_t)width * data.vertex_data_rows;
    data.vertex_positions = zeroed(3 * vertex_texels);
    data.vertex_normals = zeroed(3 * vertex_texels);
    data.vertex_colors = zeroed(3 * vertex_texels);
    for (size_t t = 0; t < mesh.triangles.size(); t++) {
        for (unsigned int corner = 0; corner < 3; corner++) {
            const vertex &vtx = mesh.vertices[mesh.triangles[t].i[corner]];
            const unsigned int slot = (unsigned int)(t * 3 + corner);
            vtx.v.store(data.vertex_positions, slot);
            vtx.n.store(data.vertex_normals, slot);
            vtx.c.store(data.vertex_colors, slot);
        }
    }

    // ---- nodes ----
    data.group_count = number_in_order(w->root);
    data.group_data_rows = (int)((data.group_count + width - 1) / width);
    const size_t node_texels = (size_t)width * data.group_data_rows;
    data.group_boxmin = zeroed(3 * node_texels);
    data.group_boxmax = zeroed(3 * node_texels);
    data.group_directions = zeroed(3 * node_texels);
    data.group_children = zeroed(2 * node_texels);
    data.group_objects = zeroed(2 * node_texels);
    data.group_hitmiss = zeroed(kDirectionCodes * 2 * node_texels);
    data.tree_root = w->root->my_index;

    for_each_node(w->root, [&](group *g) {
        const unsigned int me = (unsigned int)g->my_index;
        g->box.boxmin.store(data.group_boxmin, me);
        g->box.boxmax.store(data.group_boxmax, me);
        float *children = data.group_children + 2 * (size_t)me;
        float *objects = data.group_objects + 2 * (size_t)me;
        if (g->is_leaf()) {
            children[0] = children[1] = kStopLink;
            objects[0] = (float)g->start;
            objects[1] = (float)g->count;
        } else {
            g->D.store(data.group_directions, me);
            children[0] = (float)g->negative->my_index;
            children[1] = (float)g->positive->my_index;
        }
    });

    // the eight direction tables are independent walks (each writes only its own code's links and table): one thread each
    bool complete[kDirectionCodes];
    {
        std::vector<std::thread> walkers;
        for (int code = 0; code < kDirectionCodes; code++)
            walkers.emplace_back([&, code] {
                complete[code] = thread_direction(w->root, code, data.group_hitmiss + 2 * node_texels * code);
            });
        for (std::thread &t : walkers)
            t.join();
    }
    for (int code = 0; code < kDirectionCodes; code++) {
        if (!complete[code]) {
            fprintf(stderr, "hitmiss: tree deeper than %d, direction table %d is incomplete\n", kLinkStackCapacity, code);
            data.links_complete = false;
        }
    }
    host_info("hitmiss: %f seconds\n", seconds_since(then));
}
