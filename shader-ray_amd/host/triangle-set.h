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

#include <algorithm>
#include <cstdint>
#include <cstring>
#include <memory>
#include <thread>
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

    // Appends `count` triangles at once -- corners[3 t + k] = corner k of triangle t -- with the result add() would give
    // called on them one after the other: the same vertex numbering (first seen first), the same triangles, the same box.
    // The loaders hand a whole file over in one call (round 4: the 1M-triangle OBJ spent 0.5 s of its 0.85 s in three
    // million hash look-ups on one core).  How: every corner's key and hash in parallel; the keys are sharded by hash, and
    // each shard's thread walks ALL corners in order interning its own -- so a shard's vertices come out in first-seen
    // order with the corner that introduced each --; the shards' lists are merged by that corner number, which is the
    // global first-seen order; the triangles are then written in parallel.  threads <= 0: hardware_concurrency().
    void add_bulk(const vertex *corners, size_t count, int threads = 0)
    {
        if (count == 0)
            return;
        if (!lookup.empty() || !vertices.empty() || count < 4096) {     // something is interned already: one by one
            for (size_t t = 0; t < count; t++)
                add(corners[3 * t], corners[3 * t + 1], corners[3 * t + 2]);
            return;
        }
        const size_t n = 3 * count;
        unsigned workers = threads > 0 ? (unsigned)threads : std::thread::hardware_concurrency();
        workers = std::max(1u, std::min(workers, 64u));
        unsigned shard_bits = 0;
        while ((1u << shard_bits) < workers && shard_bits < 6)
            shard_bits++;
        const unsigned shards = 1u << shard_bits;
        auto in_parallel = [&](unsigned jobs, auto &&fn) {
            std::vector<std::thread> pool;
            for (unsigned j = 1; j < jobs; j++)
                pool.emplace_back([&fn, j] { fn(j); });
            fn(0u);
            for (std::thread &th : pool)
                th.join();
        };
        // 1. keys and hashes
        std::vector<key> keys(n);
        std::vector<uint64_t> hashes(n);
        in_parallel(workers, [&](unsigned j) {
            const size_t lo = n * j / workers, hi = n * (j + 1) / workers;
            for (size_t c = lo; c < hi; c++) {
                keys[c] = key_of(corners[c]);
                hashes[c] = (uint64_t)key_hash()(keys[c]);
            }
        });
        // 2. per shard: its corners in order
        struct shard_result {
            std::vector<uint32_t> first_corner;     // local vertex -> the corner that introduced it
        };
        std::vector<shard_result> result(shards);
        std::vector<uint32_t> local_of(n);
        const uint64_t shard_mask = shards - 1u;
        in_parallel(shards, [&](unsigned s) {
            std::unordered_map<key, uint32_t, key_hash> seen;
            seen.reserve(n / shards / 2 + 16);
            shard_result &r = result[s];
            for (size_t c = 0; c < n; c++) {
                if (((hashes[c] >> 40) & shard_mask) != s)
                    continue;
                auto found = seen.find(keys[c]);
                if (found == seen.end()) {
                    found = seen.emplace(keys[c], (uint32_t)r.first_corner.size()).first;
                    r.first_corner.push_back((uint32_t)c);
                }
                local_of[c] = found->second;
            }
        });
        // 3. the global numbering: by the corner that introduced the vertex
        struct entry {
            uint32_t first_corner, shard, local;
        };
        std::vector<entry> order;
        for (unsigned s = 0; s < shards; s++)
            for (size_t k = 0; k < result[s].first_corner.size(); k++)
                order.push_back(entry{result[s].first_corner[k], s, (uint32_t)k});
        std::sort(order.begin(), order.end(), [](const entry &a, const entry &b) { return a.first_corner < b.first_corner; });
        std::vector<std::vector<uint32_t>> global_of(shards);
        for (unsigned s = 0; s < shards; s++)
            global_of[s].resize(result[s].first_corner.size());
        vertices.resize(order.size());
        for (size_t g = 0; g < order.size(); g++) {
            global_of[order[g].shard][order[g].local] = (uint32_t)g;
            vertices[g] = corners[order[g].first_corner];
        }
        // 4. the triangles
        triangles.reserve(count);
        triangles.resize(count, indexed_triangle(0, 0, 0, corners[0], corners[1], corners[2]));
        std::vector<box3d> boxes(workers);
        in_parallel(workers, [&](unsigned j) {
            const size_t lo = count * j / workers, hi = count * (j + 1) / workers;
            for (size_t t = lo; t < hi; t++) {
                int index[3];
                for (int k = 0; k < 3; k++) {
                    const size_t c = 3 * t + k;
                    index[k] = (int)global_of[(hashes[c] >> 40) & shard_mask][local_of[c]];
                }
                triangles[t] = indexed_triangle(index[0], index[1], index[2], corners[3 * t], corners[3 * t + 1], corners[3 * t + 2]);
                boxes[j].add(triangles[t].box);
            }
        });
        for (const box3d &b : boxes)
            box.add(b);
        bulk_loaded = true;
    }

    // Drops the de-duplication index once loading is over.
    void finish()
    {
        lookup.clear();
        lookup.rehash(0);
        bulk_loaded = false;
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
    bool bulk_loaded = false;     // add_bulk left `lookup` empty: add() fills it from `vertices` first

    static uint32_t canonical_bits(float f)
    {
        if (f == 0.0f)
            f = 0.0f;   // -0 and +0 are one vertex
        uint32_t u;
        std::memcpy(&u, &f, sizeof(u));
        return u;
    }

    static key key_of(const vertex &vtx)
    {
        const float comps[9] = {vtx.v.x, vtx.v.y, vtx.v.z, vtx.n.x, vtx.n.y, vtx.n.z, vtx.c.x, vtx.c.y, vtx.c.z};
        key k;
        for (int j = 0; j < 9; j++)
            k.bits[j] = canonical_bits(comps[j]);
        return k;
    }

    int intern(const vertex &vtx)
    {
        if (bulk_loaded) {
            for (size_t g = 0; g < vertices.size(); g++)
                lookup.emplace(key_of(vertices[g]), (int)g);
            bulk_loaded = false;
        }
        const key k = key_of(vtx);
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
