// flatten.hip -- get_shader_data (reference world.cpp:298-347) on the GPU, behind the C ABI of
// include/shader_ray_hip.h (shray_flatten_device ...).  SURVEY section 8(f) rank 3.
//
// The host flattener walks the pointer tree nine times (once for the node arrays, once per direction code
// for the threaded links, world.cpp:231-288).  Here the tree arrives as pre-order arrays and every output
// element is computed independently:
//   * a triangle corner copies its vertex (world.cpp:303-318)                        -- one thread per corner
//   * a node finds its IN-ORDER number (negative subtree, self, positive subtree; world.cpp:145-177) from the
//     sizes of the negative subtrees of the ancestors it hangs to the positive side of.  In pre-order the
//     negative child of p is p + 1 and its subtree has positive(p) - p - 1 nodes, so no size array is needed
//   * a node's links for direction code c (world.cpp:231-288): hit = its near child (dot(sign(c), D) < 0 ?
//     positive : negative), miss = the far child of the nearest ancestor it hangs to the NEAR side of, or the
//     terminator 0x7fffffff (stored as float, 2147483648); a leaf's hit equals its miss  -- one thread per
//     (node, code), an O(depth) walk up the parents
// All values are copies or exact small integers converted to float, so the arrays equal the host flattener's
// bit for bit (tests/test_gpu_flatten.py, against the reference-generated fixtures in tests/golden).
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <memory>
#include <vector>

#include "device_tree_internal.h"
#include "shader_ray_hip.h"

extern "C" int shrayi_fail(int code, const char *message);   // capi.hip: sets shray_last_error()

namespace {

const float kStopLinkF = 2147483648.0f;   // (float)0x7fffffff, world.cpp:229

struct DeviceArray {
    void *p = nullptr;
    ~DeviceArray()
    {
        if (p)
            (void)hipFree(p);
    }
    hipError_t zeros(size_t bytes)
    {
        hipError_t e = hipMalloc(&p, bytes ? bytes : 16);
        return e != hipSuccess ? e : hipMemset(p, 0, bytes ? bytes : 16);   // same (null) stream as the kernels below
    }
    hipError_t upload(const void *src, size_t bytes)
    {
        hipError_t e = hipMalloc(&p, bytes ? bytes : 16);
        return (e != hipSuccess || !bytes) ? e : hipMemcpy(p, src, bytes, hipMemcpyHostToDevice);
    }
};

struct TreeView {
    int node_count;
    const int *parent, *negative, *positive, *start, *triangles;
    const float *box, *direction;
};

__global__ void expand_corners(int corners, const int *__restrict__ triangle_vertices, const float *__restrict__ vertex_data,
                               float *__restrict__ positions, float *__restrict__ colors, float *__restrict__ normals)
{
    const int slot = blockIdx.x * blockDim.x + threadIdx.x;
    if (slot >= corners)
        return;
    const float *v = vertex_data + 9 * (size_t)triangle_vertices[slot];
    for (int k = 0; k < 3; k++) {
        positions[3 * (size_t)slot + k] = v[k];
        colors[3 * (size_t)slot + k] = v[3 + k];
        normals[3 * (size_t)slot + k] = v[6 + k];
    }
}

// in-order number of pre-order node g
__device__ int in_order_index(const TreeView &t, int g)
{
    int index = t.negative[g] >= 0 ? t.positive[g] - g - 1 : 0;   // the nodes of its own negative subtree come first
    for (int child = g, p = t.parent[g]; p >= 0; child = p, p = t.parent[p])
        if (child == t.positive[p])
            index += t.positive[p] - p;                           // p's negative subtree and p itself
    return index;
}

__global__ void number_nodes(TreeView t, int *__restrict__ index_of)
{
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g < t.node_count)
        index_of[g] = in_order_index(t, g);
}

// store_group_data, world.cpp:179-210
__global__ void store_nodes(TreeView t, const int *__restrict__ index_of, float *__restrict__ boxmin, float *__restrict__ boxmax,
                            float *__restrict__ directions, float *__restrict__ children, float *__restrict__ objects)
{
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= t.node_count)
        return;
    const size_t me = (size_t)index_of[g];
    for (int k = 0; k < 3; k++) {
        boxmin[3 * me + k] = t.box[6 * (size_t)g + k];
        boxmax[3 * me + k] = t.box[6 * (size_t)g + 3 + k];
    }
    if (t.negative[g] < 0) {
        children[2 * me] = children[2 * me + 1] = kStopLinkF;
        objects[2 * me] = (float)t.start[g];
        objects[2 * me + 1] = (float)t.triangles[g];
    } else {
        for (int k = 0; k < 3; k++)
            directions[3 * me + k] = t.direction[3 * (size_t)g + k];
        children[2 * me] = (float)index_of[t.negative[g]];
        children[2 * me + 1] = (float)index_of[t.positive[g]];
    }
}

__device__ bool positive_is_near(const TreeView &t, int g, int code)
{
    const float sx = (code & 1) ? 1.0f : -1.0f, sy = (code & 2) ? 1.0f : -1.0f, sz = (code & 4) ? 1.0f : -1.0f;
    const float *d = t.direction + 3 * (size_t)g;
    const float along = sx * d[0] + sy * d[1] + sz * d[2];   // world.cpp:259, same operation order
    return along < 0;
}

// create_hitmiss, world.cpp:231-288: table `code` at table_stride * 2 * code floats
__global__ void thread_links(TreeView t, const int *__restrict__ index_of, float *__restrict__ hitmiss, size_t table_floats)
{
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    const int code = blockIdx.y;
    if (g >= t.node_count)
        return;
    int next_subtree = -1;
    for (int child = g, p = t.parent[g]; p >= 0; child = p, p = t.parent[p]) {
        const bool pos_near = positive_is_near(t, p, code);
        const int near_child = pos_near ? t.positive[p] : t.negative[p];
        if (child == near_child) {
            next_subtree = pos_near ? t.negative[p] : t.positive[p];
            break;
        }
    }
    const float miss = next_subtree < 0 ? kStopLinkF : (float)index_of[next_subtree];
    float hit = miss;
    if (t.negative[g] >= 0)
        hit = (float)index_of[positive_is_near(t, g, code) ? t.positive[g] : t.negative[g]];
    float *link = hitmiss + table_floats * (size_t)code + 2 * (size_t)index_of[g];
    link[0] = hit;
    link[1] = miss;
}

}   // namespace

struct shray_device_flat {
    shray_scene_desc desc{};   // device pointers
    DeviceArray positions, normals, colors, boxmin, boxmax, directions, children, hitmiss, objects;
    DeviceArray index_of;      // pre-order node -> its in-order number (kept for shray_scene_create_from_device)
    std::vector<float> host[9];
};

#define FLAT_TRY(expr)                                                                                           \
    do {                                                                                                         \
        hipError_t e_ = (expr);                                                                                  \
        if (e_ != hipSuccess) {                                                                                  \
            char msg_[256];                                                                                      \
            snprintf(msg_, sizeof(msg_), "%s failed: %s", #expr, hipGetErrorString(e_));                         \
            return shrayi_fail(e_ == hipErrorOutOfMemory ? SHRAY_ERR_OUT_OF_MEMORY : SHRAY_ERR_DEVICE, msg_); \
        }                                                                                                        \
    } while (0)

// The flattening proper: the tree and the mesh are on the device already.
static int flatten_on_device(const TreeView &view, const int *d_tri_vertices, const float *d_vertex_data, int tris, uint32_t width,
                             shray_device_flat **out_flat)
{
    const int n = view.node_count;
    std::unique_ptr<shray_device_flat> flat(new shray_device_flat);
    const uint32_t corners = 3u * (uint32_t)tris;
    const uint32_t vertex_rows = (corners + width - 1) / width;
    const int node_rows = (int)(((uint32_t)n + width - 1) / width);
    const size_t vertex_texels = (size_t)width * vertex_rows, node_texels = (size_t)width * node_rows;

    FLAT_TRY(flat->index_of.zeros(sizeof(int) * (size_t)n));
    FLAT_TRY(flat->positions.zeros(sizeof(float) * 3 * vertex_texels));
    FLAT_TRY(flat->normals.zeros(sizeof(float) * 3 * vertex_texels));
    FLAT_TRY(flat->colors.zeros(sizeof(float) * 3 * vertex_texels));
    FLAT_TRY(flat->boxmin.zeros(sizeof(float) * 3 * node_texels));
    FLAT_TRY(flat->boxmax.zeros(sizeof(float) * 3 * node_texels));
    FLAT_TRY(flat->directions.zeros(sizeof(float) * 3 * node_texels));
    FLAT_TRY(flat->children.zeros(sizeof(float) * 2 * node_texels));
    FLAT_TRY(flat->objects.zeros(sizeof(float) * 2 * node_texels));
    FLAT_TRY(flat->hitmiss.zeros(sizeof(float) * 16 * node_texels));

    const int block = 256;
    if (corners)
        hipLaunchKernelGGL(expand_corners, dim3((corners + block - 1) / block), dim3(block), 0, nullptr, (int)corners, d_tri_vertices,
                           d_vertex_data, (float *)flat->positions.p, (float *)flat->colors.p, (float *)flat->normals.p);
    const dim3 node_grid((n + block - 1) / block);
    int *d_index = (int *)flat->index_of.p;
    hipLaunchKernelGGL(number_nodes, node_grid, dim3(block), 0, nullptr, view, d_index);
    hipLaunchKernelGGL(store_nodes, node_grid, dim3(block), 0, nullptr, view, (const int *)d_index, (float *)flat->boxmin.p,
                       (float *)flat->boxmax.p, (float *)flat->directions.p, (float *)flat->children.p, (float *)flat->objects.p);
    hipLaunchKernelGGL(thread_links, dim3(node_grid.x, 8), dim3(block), 0, nullptr, view, (const int *)d_index,
                       (float *)flat->hitmiss.p, 2 * node_texels);
    FLAT_TRY(hipGetLastError());
    int root_index = 0;
    FLAT_TRY(hipMemcpy(&root_index, d_index, sizeof(int), hipMemcpyDeviceToHost));   // also waits for the kernels

    shray_scene_desc &d = flat->desc;
    d.struct_size = (uint32_t)sizeof(d);
    d.data_texture_width = width;
    d.vertex_count = corners;
    d.vertex_data_rows = vertex_rows;
    d.vertex_positions = (const float *)flat->positions.p;
    d.vertex_normals = (const float *)flat->normals.p;
    d.vertex_colors = (const float *)flat->colors.p;
    d.group_count = n;
    d.group_data_rows = node_rows;
    d.tree_root = root_index;
    d.group_boxmin = (const float *)flat->boxmin.p;
    d.group_boxmax = (const float *)flat->boxmax.p;
    d.group_directions = (const float *)flat->directions.p;
    d.group_children = (const float *)flat->children.p;
    d.group_hitmiss = (const float *)flat->hitmiss.p;
    d.group_objects = (const float *)flat->objects.p;
    *out_flat = flat.release();
    return SHRAY_OK;
}

extern "C" {

int shray_flatten_device(const shray_tree_desc *tree, uint32_t width, shray_device_flat **out_flat)
{
    if (!tree || !out_flat || width == 0)
        return shrayi_fail(SHRAY_ERR_INVALID_ARGUMENT, "shray_flatten_device: NULL argument or zero texture width");
    *out_flat = nullptr;
    if (tree->struct_size != sizeof(shray_tree_desc))
        return shrayi_fail(SHRAY_ERR_INVALID_ARGUMENT, "shray_tree_desc.struct_size does not match this library");
    const int n = tree->node_count, tris = tree->triangle_count;
    if (n < 1 || tris < 0 || tree->vertex_count < 0 || !tree->node_parent || !tree->node_negative || !tree->node_positive ||
        !tree->node_box || !tree->node_direction || !tree->node_start || !tree->node_triangles ||
        (tris > 0 && (!tree->triangle_vertices || !tree->vertex_data)))
        return shrayi_fail(SHRAY_ERR_INVALID_ARGUMENT, "shray_tree_desc: missing arrays or bad counts");
    // the kernels walk parent chains and trust child indices: check the pre-order shape on the host first
    if (tree->node_parent[0] != -1)
        return shrayi_fail(SHRAY_ERR_BAD_TREE, "tree: node 0 must be the root");
    for (int g = 0; g < n; g++) {
        const int neg = tree->node_negative[g], pos = tree->node_positive[g], par = tree->node_parent[g];
        const bool leaf = neg < 0;
        if (leaf ? pos >= 0 : (neg != g + 1 || pos <= neg || pos >= n || tree->node_parent[neg] != g || tree->node_parent[pos] != g))
            return shrayi_fail(SHRAY_ERR_BAD_TREE, "tree: nodes are not in pre-order (negative child = node + 1)");
        if (g > 0 && (par < 0 || par >= g))
            return shrayi_fail(SHRAY_ERR_BAD_TREE, "tree: a parent must precede its children");
        if (leaf && (tree->node_start[g] < 0 || tree->node_triangles[g] < 0 ||
                     (int64_t)tree->node_start[g] + tree->node_triangles[g] > tris))
            return shrayi_fail(SHRAY_ERR_BAD_TREE, "tree: a leaf names triangles outside the mesh");
    }
    for (int64_t k = 0; k < 3 * (int64_t)tris; k++)
        if (tree->triangle_vertices[k] < 0 || tree->triangle_vertices[k] >= tree->vertex_count)
            return shrayi_fail(SHRAY_ERR_BAD_TREE, "tree: a triangle names a vertex outside the mesh");

    DeviceArray d_parent, d_negative, d_positive, d_start, d_triangles, d_box, d_direction, d_tri_vertices, d_vertex_data;
    FLAT_TRY(d_parent.upload(tree->node_parent, sizeof(int) * (size_t)n));
    FLAT_TRY(d_negative.upload(tree->node_negative, sizeof(int) * (size_t)n));
    FLAT_TRY(d_positive.upload(tree->node_positive, sizeof(int) * (size_t)n));
    FLAT_TRY(d_start.upload(tree->node_start, sizeof(int) * (size_t)n));
    FLAT_TRY(d_triangles.upload(tree->node_triangles, sizeof(int) * (size_t)n));
    FLAT_TRY(d_box.upload(tree->node_box, sizeof(float) * 6 * (size_t)n));
    FLAT_TRY(d_direction.upload(tree->node_direction, sizeof(float) * 3 * (size_t)n));
    FLAT_TRY(d_tri_vertices.upload(tree->triangle_vertices, sizeof(int) * 3 * (size_t)tris));
    FLAT_TRY(d_vertex_data.upload(tree->vertex_data, sizeof(float) * 9 * (size_t)tree->vertex_count));
    const TreeView view{n, (const int *)d_parent.p, (const int *)d_negative.p, (const int *)d_positive.p, (const int *)d_start.p,
                        (const int *)d_triangles.p, (const float *)d_box.p, (const float *)d_direction.p};
    return flatten_on_device(view, (const int *)d_tri_vertices.p, (const float *)d_vertex_data.p, tris, width, out_flat);
}

/* The same for a tree that IS on the device (shray_bvh_build_device's): nothing is uploaded, nothing validated on the host -- the
 * builder made the pre-order shape the kernels rely on. */
int shray_flatten_device_tree(const shray_device_tree *tree, uint32_t width, shray_device_flat **out_flat)
{
    if (!tree || !out_flat || width == 0)
        return shrayi_fail(SHRAY_ERR_INVALID_ARGUMENT, "shray_flatten_device_tree: NULL argument or zero texture width");
    *out_flat = nullptr;
    ShrayDeviceTreeView t;
    const int rc = shrayi_device_tree_view(tree, &t);
    if (rc)
        return rc;
    if (t.vertex_stride_floats != 9)
        return shrayi_fail(SHRAY_ERR_INVALID_ARGUMENT, "shray_flatten_device_tree: the tree was built over vertices of another layout than "
                                                      "position, colour, normal (9 floats, geometry.h:34-38)");
    const TreeView view{t.node_count, t.parent, t.negative, t.positive, t.start, t.triangles, t.box, t.direction};
    return flatten_on_device(view, t.triangle_vertices, t.vertex_data, t.triangle_count, width, out_flat);
}

int shray_device_flat_describe(const shray_device_flat *flat, shray_scene_desc *desc)
{
    if (!flat || !desc)
        return shrayi_fail(SHRAY_ERR_INVALID_ARGUMENT, "shray_device_flat_describe: NULL argument");
    *desc = flat->desc;
    return SHRAY_OK;
}

int shray_device_flat_download(shray_device_flat *flat, shray_scene_desc *desc)
{
    if (!flat || !desc)
        return shrayi_fail(SHRAY_ERR_INVALID_ARGUMENT, "shray_device_flat_download: NULL argument");
    const shray_scene_desc &d = flat->desc;
    const size_t vt = (size_t)d.data_texture_width * d.vertex_data_rows, nt = (size_t)d.data_texture_width * (size_t)d.group_data_rows;
    const struct {
        const float *device;
        size_t floats;
    } arrays[9] = {{d.vertex_positions, 3 * vt}, {d.vertex_normals, 3 * vt}, {d.vertex_colors, 3 * vt},
                   {d.group_boxmin, 3 * nt},     {d.group_boxmax, 3 * nt},   {d.group_directions, 3 * nt},
                   {d.group_children, 2 * nt},   {d.group_hitmiss, 16 * nt}, {d.group_objects, 2 * nt}};
    for (int k = 0; k < 9; k++) {
        flat->host[k].resize(arrays[k].floats);
        if (arrays[k].floats)
            FLAT_TRY(hipMemcpy(flat->host[k].data(), arrays[k].device, sizeof(float) * arrays[k].floats, hipMemcpyDeviceToHost));
    }
    *desc = d;
    desc->vertex_positions = flat->host[0].data();
    desc->vertex_normals = flat->host[1].data();
    desc->vertex_colors = flat->host[2].data();
    desc->group_boxmin = flat->host[3].data();
    desc->group_boxmax = flat->host[4].data();
    desc->group_directions = flat->host[5].data();
    desc->group_children = flat->host[6].data();
    desc->group_hitmiss = flat->host[7].data();
    desc->group_objects = flat->host[8].data();
    return SHRAY_OK;
}

int shray_device_flat_destroy(shray_device_flat *flat)
{
    delete flat;
    return SHRAY_OK;
}

// (internal, device_tree_internal.h)
int shrayi_device_flat_view(const shray_device_flat *flat, ShrayDeviceFlatView *view)
{
    if (!flat || !view)
        return shrayi_fail(SHRAY_ERR_INVALID_ARGUMENT, "device flat is NULL");
    view->desc = flat->desc;
    view->index_of = (const int *)flat->index_of.p;
    return SHRAY_OK;
}

}   // extern "C"
