// group.h -- one BVH node, as a pointer tree (the reference's node type,
// group.h:22-40, kept so CPU-built trees drop straight in).
//
// A branch has both children and a unit split direction D; a leaf has
// negative == positive == nullptr and names triangles [start, start+count)
// of the shared triangle_set.  dirhit/dirmiss are the threaded-traversal
// links per ray-direction sign code (filled by get_shader_data), my_index the
// node's slot in the flattened arrays.
#pragma once

#include "geometry.h"
#include "triangle-set.h"

struct group {
    vec3 D;
    box3d box;

    group *negative;
    group *positive;
    group *dirhit[8];
    group *dirmiss[8];

    triangle_set_ptr triangles;
    int start;
    unsigned int count;

    int my_index;

    // branch
    group(triangle_set_ptr mesh, group *neg, group *pos, const vec3 &direction, const box3d &bounds);
    // leaf: box is recomputed from the triangles' vertices
    group(triangle_set_ptr mesh, int first, unsigned int n);
    ~group();

    group(const group &) = delete;
    group &operator=(const group &) = delete;

    bool is_leaf() const { return negative == nullptr; }
};
