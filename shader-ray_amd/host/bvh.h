// bvh.h -- binned-SAH BVH build over a triangle_set (reference: bvh.h:17-21).
#pragma once

#include "group.h"
#include "triangle-set.h"

// Build parameters; defaults are the reference's (bvh.cpp:28-58) and the same
// environment variables override them (bvh.cpp:60-79): BVH_MAX_DEPTH,
// BVH_LEAF_MAX, SAH_CTRAV, SAH_CISEC.
struct bvh_build_options {
    unsigned int leaf_max = 10;
    int max_depth = 30;
    float sah_ctrav = 1;
    float sah_cisec = 4;
    bool verbose = true;   // "Large leaf node" diagnostics on stderr
};
bvh_build_options &bvh_options();

// Statistics of all builds since the last reset (reference: print_bvh_stats, bvh.cpp:83-99).
struct bvh_build_stats {
    int node_count = 0;
    int leaf_count = 0;
    int max_level = 0;
    int large_leaves = 0;   // leaves made because no split beat the leaf cost
};
const bvh_build_stats &bvh_stats();
void reset_bvh_stats();
void print_bvh_stats();

// Recursively builds the subtree over triangles [start, start+count),
// reordering triangles->triangles in place.  Caller owns the result.
group *make_bvh(triangle_set_ptr triangles, int start, unsigned int count, int level = 0);
