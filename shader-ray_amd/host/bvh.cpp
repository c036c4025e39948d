// bvh.cpp -- recursive binned-SAH BVH build.
//
// Behavioural restatement of the reference builder (bvh.cpp:101-358): the
// tree, the per-node boxes and the in-place triangle order must come out
// bit-identical, because the flattened arrays are the data contract of the
// GPU path.  Float expressions therefore keep the reference's operand order:
//   leaf test            bvh.cpp:299      level >= max_depth || count <= leaf_max
//   split axis           bvh.cpp:317-326  strictly-longest barycentre-box axis, x then y then z
//   binning              bvh.cpp:148-170  floor((b - lo) * bins / (hi - lo)) over the VERTEX box
//   SAH                  bvh.cpp:107-120  ctrav + cisec * (Al/A*nl + Ar/A*nr)
//   split plane          bvh.cpp:172-196  lo + i * (hi - lo) / bins
//   partition            bvh.cpp:249-286  Hoare exchange on barycentre < plane
// Compile with -ffp-contract=off, no -ffast-math.
#include "bvh.h"

#include "host-log.h"

#include <cstdio>
#include <cstdlib>
#include <future>
#include <map>

namespace {

bvh_build_options g_options;
bool g_options_from_env = false;
// Everything the build counts.  A sub-tree built on another thread counts into its own tally,
// added to the parent's when the thread is joined.
struct build_tally {
    bvh_build_stats stats;
    std::map<int, int> nodes_per_level;
    std::map<int, int> leaves_per_size;
    int shapes_done = 0;

    void add(const build_tally &other)
    {
        stats.node_count += other.stats.node_count;
        stats.leaf_count += other.stats.leaf_count;
        stats.max_level = std::max(stats.max_level, other.stats.max_level);
        stats.large_leaves += other.stats.large_leaves;
        for (const auto &kv : other.nodes_per_level)
            nodes_per_level[kv.first] += kv.second;
        for (const auto &kv : other.leaves_per_size)
            leaves_per_size[kv.first] += kv.second;
        shapes_done += other.shapes_done;
    }
};
build_tally g_tally;

// Sub-trees over disjoint triangle ranges are independent (the partition works in place inside its
// own range), so the two children of a large node are built concurrently; the tree, the boxes and
// the triangle order are those of the serial build.  SHRAY_BVH_THREADS=0 builds serially.
const unsigned int kParallelMinTriangles = 32768;
bool g_parallel_build = true;

const int kMaxBins = 40;
const int kLeafSizeStatsCap = 64;

void read_env_once()
{
    if (g_options_from_env)
        return;
    g_options_from_env = true;
    if (const char *s = getenv("BVH_MAX_DEPTH")) g_options.max_depth = atoi(s);
    if (const char *s = getenv("BVH_LEAF_MAX")) g_options.leaf_max = atoi(s);
    if (const char *s = getenv("SAH_CTRAV")) g_options.sah_ctrav = atof(s);
    if (const char *s = getenv("SAH_CISEC")) g_options.sah_cisec = atof(s);
    if (const char *s = getenv("SHRAY_BVH_THREADS")) g_parallel_build = atoi(s) != 0;
}

inline float half_area_x2(const vec3 &d) { return 2 * (d.x * d.y + d.x * d.z + d.y * d.z); }

inline float leaf_cost(int n) { return g_options.sah_ctrav + g_options.sah_cisec * n; }

inline float split_cost(float area, const vec3 &ldim, int ln, const vec3 &rdim, int rn)
{
    const float la = half_area_x2(ldim);
    const float ra = half_area_x2(rdim);
    return g_options.sah_ctrav + g_options.sah_cisec * (la / area * ln + ra / area * rn);
}

// float -> int the way x86 cvttss2si does it: anything unrepresentable
// (NaN, +-inf, out of range) becomes INT_MIN.
inline int truncate_like_x86(float f)
{
    if (!(f >= -2147483648.0f && f < 2147483648.0f))
        return std::numeric_limits<int>::min();
    return (int)f;
}

group *emit_leaf(const triangle_set_ptr &mesh, int start, int count, int level, build_tally &tally)
{
    tally.shapes_done += count;
    tally.stats.node_count++;
    tally.stats.leaf_count++;
    tally.stats.max_level = std::max(tally.stats.max_level, level);
    tally.nodes_per_level[level]++;
    tally.leaves_per_size[std::min(count, kLeafSizeStatsCap)]++;
    return new group(mesh, start, (unsigned int)count);
}

// Finds the cheapest bin boundary along `axis`.  Returns the cost (== to_beat
// when nothing is cheaper) and writes the plane coordinate to *plane.
float search_split(const box3d &bounds, int axis, const std::vector<indexed_triangle> &tris, int start, int count,
                   float to_beat, float *plane)
{
    const int bins = std::min(kMaxBins, count * 2);
    const float lo = bounds.boxmin[axis];
    const float hi = bounds.boxmax[axis];

    box3d bin_box[kMaxBins];
    int bin_n[kMaxBins] = {0};
    for (int k = 0; k < count; k++) {
        const indexed_triangle &t = tris[start + k];
        const float scaled = (t.barycenter[axis] - lo) * bins / (hi - lo);
        const int b = std::min(bins - 1, std::max(0, truncate_like_x86(floorf(scaled))));
        bin_box[b].add(t.box);
        bin_n[b]++;
    }

    // suffix boxes / counts: everything in bins [i, bins)
    box3d suffix_box[kMaxBins];
    int suffix_n[kMaxBins];
    {
        box3d acc;
        int n = 0;
        for (int i = bins - 1; i >= 0; i--) {
            acc.add(bin_box[i]);
            n += bin_n[i];
            suffix_box[i] = acc;
            suffix_n[i] = n;
        }
    }

    const float area = half_area_x2(bounds.dim());
    float best = to_beat;
    box3d prefix = box3d().add(bin_box[0]);
    for (int i = 1; i < bins; i++) {
        const int rn = suffix_n[i];
        const int ln = count - rn;
        if (rn != 0 && ln != 0) {
            const float cost = split_cost(area, prefix.dim(), ln, suffix_box[i].dim(), rn);
            if (cost < best) {
                best = cost;
                *plane = lo + i * (hi - lo) / bins;
            }
        }
        prefix.add(bin_box[i]);
    }
    return best;
}

// Two-ended exchange partition: afterwards [start, result) has barycentre <
// plane and [result, start+count) the rest.  The exchange sequence is the
// reference's, so the order inside each half is too.
int split_in_place(std::vector<indexed_triangle> &tris, int start, int count, int axis, float plane)
{
    auto below = [&](int k) { return tris[k].barycenter[axis] - plane < 0; };
    int lo = start - 1;
    int hi = start + count;
    for (;;) {
        do {
            lo++;
        } while (lo < hi && below(lo));
        if (lo >= hi)
            break;
        do {
            hi--;
        } while (lo < hi && !below(hi));
        if (lo >= hi)
            break;
        std::swap(tris[lo], tris[hi]);
    }
    return lo;
}

}   // namespace

bvh_build_options &bvh_options()
{
    read_env_once();
    return g_options;
}

const bvh_build_stats &bvh_stats() { return g_tally.stats; }

void reset_bvh_stats() { g_tally = build_tally(); }

void print_bvh_stats()
{
    fprintf(stderr, "bvh: %d nodes, %d leaves, deepest level %d\n", g_tally.stats.node_count, g_tally.stats.leaf_count,
            g_tally.stats.max_level);
    for (const auto &kv : g_tally.nodes_per_level)
        fprintf(stderr, "  level %2d: %7d nodes\n", kv.first, kv.second);
    for (const auto &kv : g_tally.leaves_per_size)
        fprintf(stderr, "  %s%2d triangles: %7d leaves\n", kv.first == kLeafSizeStatsCap ? ">=" : "  ", kv.first,
                kv.second);
}

namespace {

group *build_subtree(const triangle_set_ptr &mesh, int start, unsigned int count, int level, build_tally &tally)
{
    if (level >= g_options.max_depth || count <= g_options.leaf_max)
        return emit_leaf(mesh, start, (int)count, level, tally);

    std::vector<indexed_triangle> &tris = mesh->triangles;

    box3d vertex_box, bary_box;
    for (unsigned int k = 0; k < count; k++) {
        vertex_box.add(tris[start + k].box);
        bary_box.add(tris[start + k].barycenter);
    }
    const vec3 spread = bary_box.dim();
    const int axis = (spread.x > spread.y && spread.x > spread.z) ? 0 : (spread.y > spread.z ? 1 : 2);

    const float unsplit = leaf_cost((int)count);
    float plane = 0;
    const float best = search_split(vertex_box, axis, tris, start, (int)count, unsplit, &plane);
    if (best >= unsplit) {
        if (g_options.verbose)
            host_info("bvh: no split beats a %u-triangle leaf at level %d\n", count, level);
        tally.stats.large_leaves++;
        return emit_leaf(mesh, start, (int)count, level, tally);
    }

    const int mid = split_in_place(tris, start, (int)count, axis, plane);
    const int below = mid - start;
    const int above = (int)count - below;
    if (below <= 0 || above <= 0) {
        if (g_options.verbose)
            host_info("bvh: split left one side empty, %u-triangle leaf at level %d\n", count, level);
        tally.stats.large_leaves++;
        return emit_leaf(mesh, start, (int)count, level, tally);
    }

    vec3 direction(0.0f);
    (axis == 0 ? direction.x : (axis == 1 ? direction.y : direction.z)) = 1.0f;

    group *neg, *pos;
    if (g_parallel_build && count >= kParallelMinTriangles) {
        build_tally other;
        std::future<group *> negative = std::async(std::launch::async, [&] {
            return build_subtree(mesh, start, (unsigned int)below, level + 1, other);
        });
        pos = build_subtree(mesh, mid, (unsigned int)above, level + 1, tally);
        neg = negative.get();
        tally.add(other);
    } else {
        neg = build_subtree(mesh, start, (unsigned int)below, level + 1, tally);
        pos = build_subtree(mesh, mid, (unsigned int)above, level + 1, tally);
    }
    tally.stats.node_count++;
    tally.stats.max_level = std::max(tally.stats.max_level, level);
    tally.nodes_per_level[level]++;
    return new group(mesh, neg, pos, direction, vertex_box);
}

}   // namespace

group *make_bvh(triangle_set_ptr mesh, int start, unsigned int count, int level)
{
    read_env_once();
    return build_subtree(mesh, start, count, level, g_tally);
}
