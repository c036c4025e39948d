// triangle-set.h -- indexed triangle mesh container.
//
// Same public surface as the reference's triangle_set (triangle-set.h:48-102):
// `vertices`, `triangles`, `box`, operator[] / get(), add(), finish(), swap().
// add() de-duplicates vertices that compare equal in position, normal and
// colour and numbers them in first-seen order, like the reference's ordered
// map with VertexComparator (triangle-set.h:26-46); here the lookup is a hash
// on the nine floats (with -0 folded onto +0 so that the equivalence classes
// are the same as under operator<).
#pragma once

#include <cstdint>
#include <memory>
#include <unordered_map>
#include <vector>

#include "geometry.h"

struct triangle_set {
    std::vector<vertex> vertices;
    std::vector<indexed_triangle> triangles;
    box3d box;

    triangle operator[](int t) const
    {
        const indexed_triangle &it = triangles[t];
        return triangle(vertices[it.i[0]], vertices[it.i[1]], vertices[it.i[2]]);
    }
    triangle get(int t) const { return (*this)[t]; }

    // Appends one triangle; returns its index.
    int add(const vertex &a, const vertex &b, const vertex &c)
    {
        const int ia = intern(a);
        const int ib = intern(b);
        const int ic = intern(c);
        triangles.emplace_back(ia, ib, ic, a, b, c);
        box.add(triangles.back().box);
        return (int)triangles.size() - 1;
    }

    // Drops the de-duplication index once loading is over.
    void finish()
    {
        lookup.clear();
        lookup.rehash(0);
    }

    void swap(int a, int b) { std::swap(triangles[a], triangles[b]); }

private:
    struct key {
        uint32_t bits[9];
        bool operator==(const key &o) const { return std::memcmp(bits, o.bits, sizeof(bits)) == 0; }
    };
    struct key_hash {
        size_t operator()(const key &k) const
        {
            uint64_t h = 0x9e3779b97f4a7c15ull;
            for (uint32_t b : k.bits) {
                h ^= b;
                h *= 0x100000001b3ull;
                h ^= h >> 29;
            }
            return (size_t)h;
        }
    };
    std::unordered_map<key, int, key_hash> lookup;

    static uint32_t canonical_bits(float f)
    {
        if (f == 0.0f)
            f = 0.0f;   // -0 and +0 are one vertex
        uint32_t u;
        std::memcpy(&u, &f, sizeof(u));
        return u;
    }

    int intern(const vertex &vtx)
    {
        const float comps[9] = {vtx.v.x, vtx.v.y, vtx.v.z, vtx.n.x, vtx.n.y, vtx.n.z, vtx.c.x, vtx.c.y, vtx.c.z};
        key k;
        for (int j = 0; j < 9; j++)
            k.bits[j] = canonical_bits(comps[j]);
        auto found = lookup.find(k);
        if (found != lookup.end())
            return found->second;
        const int index = (int)vertices.size();
        vertices.push_back(vtx);
        lookup.emplace(k, index);
        return index;
    }
};

typedef std::shared_ptr<triangle_set> triangle_set_ptr;
