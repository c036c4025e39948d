// geometry.h -- vertex / triangle records shared by the loaders, the BVH
// builder and the flattener.  Same public types as the reference's
// geometry.h:22-91 (range, vertex, triangle, indexed_triangle).
#pragma once

#include "vectormath.h"

struct range {
    float t0, t1;
    range() : t0(-std::numeric_limits<float>::max()), t1(std::numeric_limits<float>::max()) {}
    range(float lo, float hi) : t0(lo), t1(hi) {}
    operator bool() const { return t0 < t1; }
};

struct vertex {
    vec3 v;   // position
    vec3 c;   // colour (linear)
    vec3 n;   // normal
};

struct triangle {
    vec3 v[3];
    vec3 c[3];
    vec3 n[3];
    triangle() {}
    triangle(const vertex &a, const vertex &b, const vertex &d)
    {
        const vertex *src[3] = {&a, &b, &d};
        for (int k = 0; k < 3; k++) {
            v[k] = src[k]->v;
            c[k] = src[k]->c;
            n[k] = src[k]->n;
        }
    }
    triangle(const vec3 pos[3], const vec3 col[3], const vec3 nrm[3])
    {
        for (int k = 0; k < 3; k++) {
            v[k] = pos[k];
            c[k] = col[k];
            n[k] = nrm[k];
        }
    }
};

// A triangle as three indices into triangle_set::vertices, with the two
// quantities the BVH builder bins on: the (inflated) vertex box and the
// barycentre (v0 + v1 + v2) / 3 (geometry.h:79-90).
struct indexed_triangle {
    int i[3];
    box3d box;
    vec3 barycenter;
    indexed_triangle(int i0, int i1, int i2, const vertex &a, const vertex &b, const vertex &d)
    {
        i[0] = i0; i[1] = i1; i[2] = i2;
        box.add(a.v, b.v, d.v);
        barycenter = (a.v + b.v + d.v) / 3.0f;
    }
};
