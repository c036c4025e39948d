// group.cpp -- BVH node construction / teardown (reference: group.cpp:19-47).
#include "group.h"

#include <vector>

namespace {
void clear_links(group *g)
{
    for (int c = 0; c < 8; c++) {
        g->dirhit[c] = nullptr;
        g->dirmiss[c] = nullptr;
    }
}
}   // namespace

group::group(triangle_set_ptr mesh, group *neg, group *pos, const vec3 &direction, const box3d &bounds)
    : D(direction), box(bounds), negative(neg), positive(pos), triangles(mesh), start(0), count(0), my_index(-1)
{
    clear_links(this);
}

group::group(triangle_set_ptr mesh, int first, unsigned int n)
    : D(0.0f), negative(nullptr), positive(nullptr), triangles(mesh), start(first), count(n), my_index(-1)
{
    clear_links(this);
    for (unsigned int k = 0; k < count; k++) {
        const indexed_triangle &it = triangles->triangles[start + k];
        for (int corner = 0; corner < 3; corner++)
            box.add(triangles->vertices[it.i[corner]].v);
    }
}

// Iterative teardown: a 1M-triangle tree is ~290k nodes, no need to recurse.
group::~group()
{
    std::vector<group *> pending;
    if (negative) pending.push_back(negative);
    if (positive) pending.push_back(positive);
    negative = positive = nullptr;
    while (!pending.empty()) {
        group *g = pending.back();
        pending.pop_back();
        if (g->negative) pending.push_back(g->negative);
        if (g->positive) pending.push_back(g->positive);
        g->negative = g->positive = nullptr;
        delete g;
    }
}
