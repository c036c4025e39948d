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
#include <system_error>
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
    // million hash look-ups on one core).  How (round 6: no copy of the keys, no node-based map, no sort): every corner's
    // hash in parallel; the corner numbers are partitioned by hash into shards, in corner order; each shard's thread interns
    // its own corners in an open-addressing table whose entries are corner numbers (two corners are one vertex when their
    // nine canonical words agree) and notes for every corner the corner that introduced its vertex; a vertex's global number
    // is the count of introducing corners in front of its own -- a prefix sum, which IS the first-seen order --; the
    // triangles are then written in parallel.  threads <= 0: hardware_concurrency().
    void add_bulk(const vertex *corners, size_t count, int threads = 0)
    {
        if (count == 0)
            return;
        if (!lookup.empty() || !vertices.empty() || count < 4096 || count > 0x50000000u) {     // something is interned already: one by one
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
            unsigned started = 1;
            for (; started < jobs; started++) {
                try {
                    pool.emplace_back([&fn, started] { fn(started); });
                } catch (const std::system_error &) {     // (a thread the system refuses: its job, and the rest, on this one)
                    break;
                }
            }
            fn(0u);
            for (unsigned j = started; j < jobs; j++)
                fn(j);
            for (std::thread &th : pool)
                th.join();
        };
        // 1. hashes; how many corners of each worker's range fall into each shard
        std::unique_ptr<uint64_t[]> hashes(new uint64_t[n]);
        const uint64_t shard_mask = shards - 1u;
        std::vector<uint32_t> tally((size_t)workers * shards, 0u);
        in_parallel(workers, [&](unsigned j) {
            const size_t lo = n * j / workers, hi = n * (j + 1) / workers;
            uint32_t *mine = &tally[(size_t)j * shards];
            for (size_t c = lo; c < hi; c++) {
                const uint64_t h = (uint64_t)key_hash()(key_of(corners[c]));
                hashes[c] = h;
                mine[(h >> 40) & shard_mask]++;
            }
        });
        // 2. the corner numbers by shard, in corner order inside a shard (worker ranges are in corner order)
        std::vector<size_t> shard_begin(shards + 1, 0);
        std::vector<size_t> cursor((size_t)workers * shards);
        {
            size_t run = 0;
            for (unsigned s = 0; s < shards; s++) {
                shard_begin[s] = run;
                for (unsigned j = 0; j < workers; j++) {
                    cursor[(size_t)j * shards + s] = run;
                    run += tally[(size_t)j * shards + s];
                }
            }
            shard_begin[shards] = run;
        }
        std::unique_ptr<uint32_t[]> listed(new uint32_t[n]);
        in_parallel(workers, [&](unsigned j) {
            const size_t lo = n * j / workers, hi = n * (j + 1) / workers;
            size_t *at = &cursor[(size_t)j * shards];
            for (size_t c = lo; c < hi; c++)
                listed[at[(hashes[c] >> 40) & shard_mask]++] = (uint32_t)c;
        });
        // 3. per shard: intern its corners in order; introduced_by[c] = the corner that brought c's vertex in (c itself: a new vertex)
        std::unique_ptr<uint32_t[]> introduced_by(new uint32_t[n]);
        in_parallel(shards, [&](unsigned s) {
            const size_t lo = shard_begin[s], hi = shard_begin[s + 1];
            size_t capacity = 16;
            while (capacity < 2 * (hi - lo))
                capacity <<= 1;
            const uint32_t empty = 0xffffffffu;
            std::vector<uint32_t> table(capacity, empty);
            for (size_t k = lo; k < hi; k++) {
                const uint32_t c = listed[k];
                const key mine = key_of(corners[c]);
                size_t slot = (size_t)hashes[c] & (capacity - 1);
                for (;;) {
                    const uint32_t other = table[slot];
                    if (other == empty) {
                        table[slot] = c;
                        introduced_by[c] = c;
                        break;
                    }
                    if (key_of(corners[other]) == mine) {
                        introduced_by[c] = other;
                        break;
                    }
                    slot = (slot + 1) & (capacity - 1);
                }
            }
        });
        // 4. the global numbering: a new vertex's number = how many corners in front of its own introduced one
        std::unique_ptr<uint32_t[]> number_at(new uint32_t[n]);
        std::vector<size_t> introduced(workers + 1, 0);
        in_parallel(workers, [&](unsigned j) {
            const size_t lo = n * j / workers, hi = n * (j + 1) / workers;
            size_t mine = 0;
            for (size_t c = lo; c < hi; c++)
                mine += introduced_by[c] == (uint32_t)c;
            introduced[j + 1] = mine;
        });
        for (unsigned j = 0; j < workers; j++)
            introduced[j + 1] += introduced[j];
        vertices.resize(introduced[workers]);
        in_parallel(workers, [&](unsigned j) {
            const size_t lo = n * j / workers, hi = n * (j + 1) / workers;
            size_t next = introduced[j];
            for (size_t c = lo; c < hi; c++)
                if (introduced_by[c] == (uint32_t)c) {
                    number_at[c] = (uint32_t)next;
                    vertices[next++] = corners[c];
                }
        });
        // 5. the triangles
        triangles.reserve(count);
        triangles.resize(count, indexed_triangle(0, 0, 0, corners[0], corners[1], corners[2]));
        std::vector<box3d> boxes(workers);
        in_parallel(workers, [&](unsigned j) {
            const size_t lo = count * j / workers, hi = count * (j + 1) / workers;
            for (size_t t = lo; t < hi; t++) {
                int index[3];
                for (int k = 0; k < 3; k++)
                    index[k] = (int)number_at[introduced_by[3 * t + k]];
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
