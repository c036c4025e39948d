// host_capi.cpp -- extern "C" view of the host layer (include/shader_ray_host.h).
#include <chrono>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <utility>
#include <vector>

#include "background.h"
#include "bvh.h"
#include "frame-params.h"
#include "host-log.h"
#include "shader_ray_host.h"
#include "world.h"

bool g_host_quiet = false;

struct shray_host_world {
    world_ptr w;
    // at most one flattening per data_texture_width: asking again for a width reuses it (the tree does not change
    // after load_world), so a caller that re-flattens per frame does not grow memory
    std::vector<std::pair<unsigned int, std::unique_ptr<scene_shader_data>>> flat;
    // shray_host_export_tree
    std::vector<int32_t> tree_parent, tree_negative, tree_positive, tree_start, tree_triangles, tri_vertices;
    std::vector<float> tree_box, tree_direction;
    bvh_build_stats stats;
    double load_seconds = 0;
};

extern "C" {

void shray_host_set_quiet(int quiet) { g_host_quiet = quiet != 0; }

int shray_host_load_world(const char *filename, shray_host_world **out_world)
{
    if (!filename || !out_world)
        return -1;
    *out_world = nullptr;
    const auto then = std::chrono::steady_clock::now();
    world_ptr w = load_world(filename);
    if (!w)
        return -1;
    auto *hw = new shray_host_world;
    hw->w = w;
    hw->stats = bvh_stats();
    hw->load_seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - then).count();
    *out_world = hw;
    return 0;
}

void shray_host_free_world(shray_host_world *world) { delete world; }

int shray_host_load_triangles(const char *filename, shray_host_world **out_world)
{
    if (!filename || !out_world)
        return -1;
    *out_world = nullptr;
    const auto then = std::chrono::steady_clock::now();
    world_ptr w = load_triangles(filename);
    if (!w)
        return -1;
    auto *hw = new shray_host_world;
    hw->w = w;
    hw->load_seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - then).count();
    *out_world = hw;
    return 0;
}

int shray_host_triangles(shray_host_world *world, const int32_t **triangle_vertices, int32_t *triangle_count, const float **vertex_data,
                         int32_t *vertex_count)
{
    if (!world || !world->w || !triangle_vertices || !triangle_count || !vertex_data || !vertex_count)
        return -1;
    const triangle_set &mesh = *world->w->triangles;
    world->tri_vertices.resize(3 * mesh.triangles.size());
    for (size_t t = 0; t < mesh.triangles.size(); t++)
        for (int corner = 0; corner < 3; corner++)
            world->tri_vertices[3 * t + corner] = mesh.triangles[t].i[corner];
    static_assert(sizeof(vertex) == 9 * sizeof(float), "vertex is nine packed floats: position, colour, normal");
    *triangle_vertices = world->tri_vertices.data();
    *triangle_count = (int32_t)mesh.triangles.size();
    *vertex_data = mesh.vertices.empty() ? nullptr : &mesh.vertices[0].v.x;
    *vertex_count = (int32_t)mesh.vertices.size();
    return 0;
}

int shray_host_bvh_options(shray_bvh_options *options)
{
    if (!options)
        return -1;
    const bvh_build_options &o = bvh_options();
    options->struct_size = (uint32_t)sizeof(shray_bvh_options);
    options->max_depth = o.max_depth;
    options->leaf_max = (int32_t)o.leaf_max;
    options->sah_ctrav = o.sah_ctrav;
    options->sah_cisec = o.sah_cisec;
    return 0;
}

int shray_host_adopt_tree(shray_host_world *world, const shray_tree_desc *tree, const int32_t *triangle_order, double build_seconds)
{
    if (!world || !world->w || !tree || !triangle_order || tree->struct_size != sizeof(shray_tree_desc))
        return -1;
    if (!adopt_tree(world->w, tree->node_count, tree->node_negative, tree->node_positive, tree->node_box, tree->node_direction,
                    tree->node_start, tree->node_triangles, triangle_order, tree->triangle_count))
        return -1;
    world->w->build_seconds = build_seconds;
    // the statistics make_bvh keeps (print_bvh_stats, bvh.cpp:83-99), from the tree itself
    bvh_build_stats stats;
    std::vector<std::pair<const group *, int>> todo(1, {world->w->root, 0});
    const bvh_build_options &o = bvh_options();
    while (!todo.empty()) {
        const auto [g, level] = todo.back();
        todo.pop_back();
        stats.node_count++;
        stats.max_level = std::max(stats.max_level, level);
        if (g->is_leaf()) {
            stats.leaf_count++;
            if (g->count > o.leaf_max && level < o.max_depth)
                stats.large_leaves++;
        } else {
            todo.push_back({g->negative, level + 1});
            todo.push_back({g->positive, level + 1});
        }
    }
    world->stats = stats;
    return 0;
}

int shray_host_get_world_info(const shray_host_world *world, shray_host_world_info *info)
{
    if (!world || !info)
        return -1;
    memset(info, 0, sizeof(*info));
    const world_ptr &w = world->w;
    info->triangle_count = w->triangle_count;
    info->independent_vertex_count = (int32_t)w->triangles->vertices.size();
    info->scene_center[0] = w->scene_center.x;
    info->scene_center[1] = w->scene_center.y;
    info->scene_center[2] = w->scene_center.z;
    info->scene_extent = w->scene_extent;
    info->node_count = world->stats.node_count;
    info->leaf_count = world->stats.leaf_count;
    info->max_level = world->stats.max_level;
    info->large_leaves = world->stats.large_leaves;
    info->parse_seconds = w->parse_seconds + w->extent_seconds;   // file -> triangle_set, centre + extent
    info->build_seconds = w->build_seconds;                       // make_bvh
    return 0;
}

int shray_host_flatten(shray_host_world *world, unsigned int data_texture_width, shray_scene_desc *desc)
{
    if (!world || !desc || data_texture_width == 0)
        return -1;
    scene_shader_data *found = nullptr;
    for (auto &entry : world->flat)
        if (entry.first == data_texture_width)
            found = entry.second.get();
    if (!found) {
        std::unique_ptr<scene_shader_data> fresh(new scene_shader_data);
        get_shader_data(world->w, *fresh, data_texture_width);
        if (!fresh->links_complete)
            return -1;   // upstream asserts here (world.cpp:228): never hand out half-threaded link tables
        world->flat.emplace_back(data_texture_width, std::move(fresh));
        found = world->flat.back().second.get();
    }
    scene_shader_data &d = *found;

    memset(desc, 0, sizeof(*desc));
    desc->struct_size = (uint32_t)sizeof(*desc);
    desc->data_texture_width = data_texture_width;
    desc->vertex_count = d.vertex_count;
    desc->vertex_data_rows = d.vertex_data_rows;
    desc->vertex_positions = d.vertex_positions;
    desc->vertex_normals = d.vertex_normals;
    desc->vertex_colors = d.vertex_colors;
    desc->group_count = d.group_count;
    desc->group_data_rows = d.group_data_rows;
    desc->tree_root = d.tree_root;
    desc->group_boxmin = d.group_boxmin;
    desc->group_boxmax = d.group_boxmax;
    desc->group_directions = d.group_directions;
    desc->group_children = d.group_children;
    desc->group_hitmiss = d.group_hitmiss;
    desc->group_objects = d.group_objects;
    return 0;
}

int shray_host_export_tree(shray_host_world *world, shray_tree_desc *tree)
{
    if (!world || !tree || !world->w || !world->w->root)
        return -1;
    // pre-order without recursion: a node, its negative subtree, its positive subtree
    world->tree_parent.clear();
    world->tree_negative.clear();
    world->tree_positive.clear();
    world->tree_start.clear();
    world->tree_triangles.clear();
    world->tree_box.clear();
    world->tree_direction.clear();
    struct pending {
        const group *g;
        int32_t parent;
        bool positive_side;
    };
    std::vector<pending> todo(1, pending{world->w->root, -1, false});
    while (!todo.empty()) {
        const pending p = todo.back();
        todo.pop_back();
        const int32_t me = (int32_t)world->tree_parent.size();
        world->tree_parent.push_back(p.parent);
        world->tree_negative.push_back(-1);
        world->tree_positive.push_back(-1);
        if (p.parent >= 0)
            (p.positive_side ? world->tree_positive : world->tree_negative)[(size_t)p.parent] = me;
        const group *g = p.g;
        const float box[6] = {g->box.boxmin.x, g->box.boxmin.y, g->box.boxmin.z, g->box.boxmax.x, g->box.boxmax.y, g->box.boxmax.z};
        world->tree_box.insert(world->tree_box.end(), box, box + 6);
        const float dir[3] = {g->D.x, g->D.y, g->D.z};
        world->tree_direction.insert(world->tree_direction.end(), dir, dir + 3);
        world->tree_start.push_back(g->is_leaf() ? g->start : 0);
        world->tree_triangles.push_back(g->is_leaf() ? (int32_t)g->count : 0);
        if (!g->is_leaf()) {
            todo.push_back(pending{g->positive, me, true});
            todo.push_back(pending{g->negative, me, false});
        }
    }
    const triangle_set &mesh = *world->w->triangles;
    world->tri_vertices.resize(3 * mesh.triangles.size());
    for (size_t t = 0; t < mesh.triangles.size(); t++)
        for (int corner = 0; corner < 3; corner++)
            world->tri_vertices[3 * t + corner] = mesh.triangles[t].i[corner];
    static_assert(sizeof(vertex) == 9 * sizeof(float), "vertex is nine packed floats: position, colour, normal");

    memset(tree, 0, sizeof(*tree));
    tree->struct_size = (uint32_t)sizeof(*tree);
    tree->node_count = (int32_t)world->tree_parent.size();
    tree->node_parent = world->tree_parent.data();
    tree->node_negative = world->tree_negative.data();
    tree->node_positive = world->tree_positive.data();
    tree->node_box = world->tree_box.data();
    tree->node_direction = world->tree_direction.data();
    tree->node_start = world->tree_start.data();
    tree->node_triangles = world->tree_triangles.data();
    tree->triangle_count = (int32_t)mesh.triangles.size();
    tree->triangle_vertices = world->tri_vertices.data();
    tree->vertex_count = (int32_t)mesh.vertices.size();
    tree->vertex_data = mesh.vertices.empty() ? nullptr : &mesh.vertices[0].v.x;
    return 0;
}

int shray_host_default_view(const shray_host_world *world, shray_host_view *view)
{
    if (!world || !view)
        return -1;
    const view_state s = default_view_state(world->w);
    view->fov = s.fov;
    view->zoom = s.zoom;
    memcpy(view->object_rotation, s.object_rotation, sizeof(view->object_rotation));
    view->object_position[0] = s.object_position.x;
    view->object_position[1] = s.object_position.y;
    view->object_position[2] = s.object_position.z;
    memcpy(view->light_rotation, s.light_rotation, sizeof(view->light_rotation));
    view->which = s.which;
    view->which_material = s.which_material;
    view->which_diffuse_color = s.which_diffuse_color;
    return 0;
}

int shray_host_trackball_motion(const float previous[4], float dx, float dy, float result[4])
{
    if (!previous || !result)
        return -1;
    float prev[4] = {previous[0], previous[1], previous[2], previous[3]};
    trackball_motion(prev, dx, dy, result);
    return 0;
}

int shray_host_frame_params(shray_host_world *world, const shray_host_view *view, int width, int height,
                            shray_frame_params *params)
{
    if (!world || !view || !params || width <= 0 || height <= 0)
        return -1;
    if (view->which_material < 0 || view->which_diffuse_color < 0)
        return -1;
    view_state s;
    s.fov = view->fov;
    s.zoom = view->zoom;
    memcpy(s.object_rotation, view->object_rotation, sizeof(s.object_rotation));
    s.object_position = vec3(view->object_position[0], view->object_position[1], view->object_position[2]);
    memcpy(s.light_rotation, view->light_rotation, sizeof(s.light_rotation));
    s.which = view->which;
    s.which_material = view->which_material;
    s.which_diffuse_color = view->which_diffuse_color;
    make_frame_params(world->w, s, width, height, params);
    return 0;
}

int shray_host_load_background(const char *spec, int *width, int *height, float **pixels)
{
    if (!spec || !width || !height || !pixels)
        return -1;
    float2Dimage image;
    if (!load_background(spec, image))
        return -1;
    float *copy = (float *)malloc(image.pixels.size() * sizeof(float));
    if (!copy)
        return -1;
    memcpy(copy, image.pixels.data(), image.pixels.size() * sizeof(float));
    *width = image.width;
    *height = image.height;
    *pixels = copy;
    return 0;
}

void shray_host_free_background(float *pixels) { free(pixels); }

}   // extern "C"
