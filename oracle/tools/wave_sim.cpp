// wave_sim.cpp -- DESIGN-TIME DIAGNOSTIC (test infrastructure): replays the per-ray event traces recorded
// by the CPU oracle (oracle/tools/dump_trace.py) through models of the HIP kernel's wave scheduling, and
// reports how many wave-instructions each candidate policy would execute and with how many lanes active.
// It models instruction COUNTS (the kernel is VALU-issue bound), not time.
//
// A wave = the 64 pixels of an 8x8 tile; a workgroup = a 16x16 patch = 4 waves (kernel_stack.hip).
// A ray = a sequence of traversals; a traversal = a sequence of node visits, each followed by 0..10
// triangle tests (trace byte per visit).  Lanes run traversal k together (trace_common.h: trace_ray).
//
// Build: g++ -O2 -shared -fPIC -o /tmp/wave_sim.so oracle/tools/wave_sim.cpp ; driven by wave_sim.py
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>

namespace {

struct Costs {   // wave-instructions (VALU) per step
    double gen, setup, node, tri, shade, env, stage_switch, loop_iter;
    double deal_setup, deal_round, deal_finish;   // dealt leaf stage
    double compact;                               // per wave per bounce, workgroup compaction
    double event;                                 // async: one event stage (shade + next-traversal set-up, or next pixel)
};

struct Policy {
    int keep_num;         // node loop yields when walkers < alive * keep_num / 64 (floor keep_floor) and someone is parked
    int keep_floor;
    int node_turns;       // visits per evaluation of the exit test
    int deal_max_parked;  // use the dealt leaf stage when parked lanes <= this (0 = never)
    int compact;          // 1 = workgroup-level compaction of surviving rays between traversals
    int deal_group;       // 0 = general dealing (ceil(T/64) rounds), else power-of-two groups
    int async_lanes;      // 1 = every lane runs its own bounce loop; the wave schedules node / leaf / event stages
    int event_min;        // lanes that must wait for an event stage (shade / next ray) before it runs while others still traverse
    int pixels_per_wave;  // async only: 64, 128 or 256 pixels streamed through the wave's 64 lanes
    int epoch_turns;      // > 0: the workgroup's waves meet every epoch_turns node turns and repack their live rays
                          //      into fewer waves whenever they fit (mid-traversal compaction; needs compact = 1)
    int epoch_slack;      // repack only when at least this many lanes would be freed beyond a whole wave
    int interleave;       // 1: wave w of a patch takes the pixels with (x & 1) + 2 (y & 1) == w instead of an 8x8 tile
    int split_heavy;      // > 0: a wave whose stream exceeds this many instructions is replayed as four waves of 16 rays
    int leaf_cap;         // > 0: the plain leaf stage runs at most this many rounds while rays still walk; a parked ray with
                          //      triangles left stays parked and resumes in the next leaf stage
};

struct Ray {
    const uint8_t *p, *end;   // current traversal's visit bytes
};

struct Totals {
    double wave_instr = 0, lane_instr = 0;           // all phases
    double node_w = 0, node_l = 0, leaf_w = 0, leaf_l = 0, other_w = 0, other_l = 0;
    double node_turns = 0, leaf_turns = 0, leaf_stages = 0;
    double node_idle_parked = 0, node_idle_ended = 0, node_idle_empty = 0, leaf_idle_walk = 0, leaf_idle_ended = 0, leaf_idle_done = 0, leaf_idle_empty = 0;
    std::vector<double> wave_stream;                 // per wave: its own instruction stream length
    std::vector<double> group_path;                  // per workgroup: critical path (sum over phases of the slowest wave)
};

// next traversal of a sample: advances `cursor` past the 0xF0 marker; returns false when the sample has no more
inline bool next_traversal(const uint8_t *&cursor, const uint8_t *sample_end, Ray &r)
{
    if (cursor >= sample_end)
        return false;
    // *cursor == 0xF0
    const uint8_t *p = cursor + 1;
    const uint8_t *q = p;
    while (q < sample_end && *q != 0xF0)
        q++;
    r.p = p;
    r.end = q;
    cursor = q;
    return true;
}

// One wave-cooperative traversal of up to 64 rays; returns the wave's instruction count for it.
double traverse(const Costs &c, const Policy &pol, Ray *rays, int n, Totals &t)
{
    enum { WALK, LEAF, ENDED };
    int state[64], tests[64];
    for (int i = 0; i < n; i++) {
        state[i] = rays[i].p < rays[i].end ? WALK : ENDED;
        tests[i] = 0;
    }
    double stream = 0;
    auto count = [&](int s) { int k = 0; for (int i = 0; i < n; i++) k += state[i] == s; return k; };
    for (;;) {
        const int alive = n - count(ENDED);
        if (!alive)
            break;
        stream += c.loop_iter;
        t.other_w += c.loop_iter;
        t.other_l += c.loop_iter * alive;
        const int keep = std::max(pol.keep_floor, (alive * pol.keep_num + 32) >> 6);
        // node loop
        if (count(WALK)) {
            stream += c.stage_switch;
            t.other_w += c.stage_switch;
            t.other_l += c.stage_switch * alive;
        }
        for (;;) {
            if (!count(WALK))
                break;
            for (int turn = 0; turn < pol.node_turns; turn++) {
                const int walkers = count(WALK);
                if (!walkers) {
                    continue;
                }
                stream += c.node;
                t.node_w += c.node;
                t.node_l += c.node * walkers;
                t.node_turns++;
                t.node_idle_parked += c.node * count(LEAF);
                t.node_idle_ended += c.node * count(ENDED);
                t.node_idle_empty += c.node * (64 - n);
                for (int i = 0; i < n; i++) {
                    if (state[i] != WALK)
                        continue;
                    const uint8_t b = *rays[i].p++;
                    if (b) {
                        state[i] = LEAF;
                        tests[i] = b;
                    } else if (rays[i].p >= rays[i].end) {
                        state[i] = ENDED;
                    }
                }
            }
            const int walking = count(WALK);
            if (walking < keep && count(LEAF))
                break;
        }
        // leaf stage
        const int parked = count(LEAF);
        if (parked) {
            int maxc = 0, total = 0;
            for (int i = 0; i < n; i++)
                if (state[i] == LEAF) {
                    maxc = std::max(maxc, tests[i]);
                    total += tests[i];
                }
            t.leaf_stages++;
            double w = 0, l = 0;
            if (pol.deal_max_parked && parked <= pol.deal_max_parked) {
                int rounds;
                if (pol.deal_group == 0) {
                    rounds = (total + 63) / 64;
                } else {
                    int g = 1;
                    while (g * 2 * parked <= 64 && g < 16)
                        g *= 2;
                    rounds = (maxc + g - 1) / g;
                }
                w = c.deal_setup + rounds * c.deal_round + c.deal_finish;
                l = (c.deal_setup + c.deal_finish) * parked + total * c.tri;   // useful work: the tests themselves
                t.leaf_turns += rounds;
            } else {
                const int rounds = (pol.leaf_cap > 0 && count(WALK)) ? std::min(maxc, pol.leaf_cap) : maxc;
                for (int j = 0; j < rounds; j++) {
                    int act = 0;
                    for (int i = 0; i < n; i++)
                        act += state[i] == LEAF && tests[i] > j;
                    w += c.tri;
                    l += c.tri * act;
                    t.leaf_idle_walk += c.tri * count(WALK);
                    t.leaf_idle_ended += c.tri * count(ENDED);
                    t.leaf_idle_done += c.tri * (parked - act);
                    t.leaf_idle_empty += c.tri * (64 - n);
                }
                w += c.stage_switch;
                l += c.stage_switch * parked;
                t.leaf_turns += rounds;
                if (rounds < maxc) {   // rays with triangles left stay parked
                    stream += w;
                    t.leaf_w += w;
                    t.leaf_l += l;
                    for (int i = 0; i < n; i++)
                        if (state[i] == LEAF) {
                            if (tests[i] > rounds)
                                tests[i] -= rounds;
                            else
                                state[i] = rays[i].p >= rays[i].end ? ENDED : WALK;
                        }
                    continue;
                }
            }
            stream += w;
            t.leaf_w += w;
            t.leaf_l += l;
            for (int i = 0; i < n; i++)
                if (state[i] == LEAF)
                    state[i] = rays[i].p >= rays[i].end ? ENDED : WALK;
        }
    }
    return stream;
}

// Lane-independent schedule: each lane owns a list of items (an item = the samples of one pixel); it walks its
// traversals one after another; between two traversals (and between items) it needs an EVENT stage (shading and
// the next ray's set-up).  The wave runs, per iteration, the stage its lanes wait for.
struct LaneWork {
    std::vector<std::pair<const uint8_t *, const uint8_t *>> samples;   // byte ranges, in order
};
double traverse_async(const Costs &c, const Policy &pol, std::vector<LaneWork> &lanes, Totals &t)
{
    enum { WALK, LEAF, EVENT, DONE };
    const int n = (int)lanes.size();
    int state[64], tests[64];
    size_t sample_index[64];
    const uint8_t *cursor[64], *send[64];
    Ray ray[64];
    for (int i = 0; i < n; i++) {
        state[i] = lanes[i].samples.empty() ? DONE : EVENT;   // first event = ray generation
        sample_index[i] = 0;
        cursor[i] = send[i] = nullptr;
        tests[i] = 0;
    }
    auto count = [&](int s) { int k = 0; for (int i = 0; i < n; i++) k += state[i] == s; return k; };
    double stream = 0;
    for (;;) {
        const int alive = n - count(DONE);
        if (!alive)
            break;
        stream += c.loop_iter;
        t.other_w += c.loop_iter;
        t.other_l += c.loop_iter * alive;
        const int keep = std::max(pol.keep_floor, (alive * pol.keep_num + 32) >> 6);
        int walkers = count(WALK);
        const int waiting = count(EVENT);
        if (walkers && (walkers >= keep || (!count(LEAF) && waiting < pol.event_min))) {
            for (;;) {
                for (int turn = 0; turn < pol.node_turns; turn++) {
                    walkers = count(WALK);
                    if (!walkers)
                        continue;
                    stream += c.node;
                    t.node_w += c.node;
                    t.node_l += c.node * walkers;
                    t.node_turns++;
                    t.node_idle_parked += c.node * count(LEAF);
                    t.node_idle_ended += c.node * (count(EVENT) + count(DONE));
                    t.node_idle_empty += c.node * (64 - n);
                    for (int i = 0; i < n; i++) {
                        if (state[i] != WALK)
                            continue;
                        const uint8_t b = *ray[i].p++;
                        if (b) {
                            state[i] = LEAF;
                            tests[i] = b;
                        } else if (ray[i].p >= ray[i].end) {
                            state[i] = EVENT;
                        }
                    }
                }
                walkers = count(WALK);
                if (!walkers || (walkers < keep && (count(LEAF) || count(EVENT) >= pol.event_min)))
                    break;
            }
        }
        const int parked = count(LEAF);
        if (parked) {
            int maxc = 0, total = 0;
            for (int i = 0; i < n; i++)
                if (state[i] == LEAF) {
                    maxc = std::max(maxc, tests[i]);
                    total += tests[i];
                }
            double w = 0, l = 0;
            if (pol.deal_max_parked && parked <= pol.deal_max_parked) {
                const int rounds = (total + 63) / 64;
                w = c.deal_setup + rounds * c.deal_round + c.deal_finish;
                l = (c.deal_setup + c.deal_finish) * parked + total * c.tri;
                t.leaf_turns += rounds;
            } else {
                for (int j = 0; j < maxc; j++) {
                    int act = 0;
                    for (int i = 0; i < n; i++)
                        act += state[i] == LEAF && tests[i] > j;
                    w += c.tri;
                    l += c.tri * act;
                }
                w += c.stage_switch;
                l += c.stage_switch * parked;
                t.leaf_turns += maxc;
            }
            stream += w;
            t.leaf_w += w;
            t.leaf_l += l;
            for (int i = 0; i < n; i++)
                if (state[i] == LEAF)
                    state[i] = ray[i].p >= ray[i].end ? EVENT : WALK;
        }
        const int ev = count(EVENT);
        if (ev && (ev >= pol.event_min || !count(WALK))) {
            stream += c.event;
            t.other_w += c.event;
            t.other_l += c.event * ev;
            t.leaf_stages++;   // reused as: event stages
            for (int i = 0; i < n; i++) {
                if (state[i] != EVENT)
                    continue;
                // next traversal of the current sample, else the next sample, else done
                for (;;) {
                    if (cursor[i] && next_traversal(cursor[i], send[i], ray[i])) {
                        state[i] = WALK;
                        break;
                    }
                    if (sample_index[i] >= lanes[i].samples.size()) {
                        state[i] = DONE;
                        break;
                    }
                    cursor[i] = lanes[i].samples[sample_index[i]].first;
                    send[i] = lanes[i].samples[sample_index[i]].second;
                    sample_index[i]++;
                }
            }
        }
    }
    return stream;
}

// Resumable form of traverse(): per-lane state lives in the caller; runs until `max_node_turns` node turns have
// been executed (checked at stage boundaries) or every lane has ended.
struct WaveRun {
    int n = 0;
    Ray rays[64];
    int state[64], tests[64];   // 0 WALK, 1 LEAF, 2 ENDED
};
double run_epoch(const Costs &c, const Policy &pol, WaveRun &wr, int max_node_turns, Totals &t)
{
    enum { WALK, LEAF, ENDED };
    const int n = wr.n;
    int *state = wr.state, *tests = wr.tests;
    Ray *rays = wr.rays;
    double stream = 0;
    int turns_done = 0;
    auto count = [&](int s) { int k = 0; for (int i = 0; i < n; i++) k += state[i] == s; return k; };
    for (;;) {
        const int alive = n - count(ENDED);
        if (!alive || turns_done >= max_node_turns)
            break;
        stream += c.loop_iter;
        t.other_w += c.loop_iter;
        t.other_l += c.loop_iter * alive;
        const int keep = std::max(pol.keep_floor, (alive * pol.keep_num + 32) >> 6);
        for (;;) {
            if (!count(WALK))
                break;
            for (int turn = 0; turn < pol.node_turns; turn++) {
                const int walkers = count(WALK);
                if (!walkers)
                    continue;
                stream += c.node;
                t.node_w += c.node;
                t.node_l += c.node * walkers;
                t.node_turns++;
                turns_done++;
                t.node_idle_parked += c.node * count(LEAF);
                t.node_idle_ended += c.node * count(ENDED);
                t.node_idle_empty += c.node * (64 - n);
                for (int i = 0; i < n; i++) {
                    if (state[i] != WALK)
                        continue;
                    const uint8_t b = *rays[i].p++;
                    if (b) {
                        state[i] = LEAF;
                        tests[i] = b;
                    } else if (rays[i].p >= rays[i].end) {
                        state[i] = ENDED;
                    }
                }
            }
            const int walking = count(WALK);
            if ((walking < keep && count(LEAF)) || turns_done >= max_node_turns)
                break;
        }
        const int parked = count(LEAF);
        if (parked) {
            int maxc = 0, total = 0;
            for (int i = 0; i < n; i++)
                if (state[i] == LEAF) {
                    maxc = std::max(maxc, tests[i]);
                    total += tests[i];
                }
            t.leaf_stages++;
            double w = 0, l = 0;
            if (pol.deal_max_parked && parked <= pol.deal_max_parked) {
                const int rounds = (total + 63) / 64;
                w = c.deal_setup + rounds * c.deal_round + c.deal_finish;
                l = (c.deal_setup + c.deal_finish) * parked + total * c.tri;
                t.leaf_turns += rounds;
            } else {
                for (int j = 0; j < maxc; j++) {
                    int act = 0;
                    for (int i = 0; i < n; i++)
                        act += state[i] == LEAF && tests[i] > j;
                    w += c.tri;
                    l += c.tri * act;
                }
                w += c.stage_switch;
                l += c.stage_switch * parked;
                t.leaf_turns += maxc;
            }
            stream += w;
            t.leaf_w += w;
            t.leaf_l += l;
            for (int i = 0; i < n; i++)
                if (state[i] == LEAF)
                    state[i] = rays[i].p >= rays[i].end ? ENDED : WALK;
        }
    }
    return stream;
}

}   // namespace

extern "C" int wave_sim(const uint8_t *bytes, const uint64_t *offsets, int W, int H, int spp, const Costs *cp,
                        const Policy *pp, double *out, double *wave_streams, double *group_paths)
{
    const Costs &c = *cp;
    const Policy &pol = *pp;
    Totals t;
    const int patches_x = (W + 15) / 16, patches_y = (H + 15) / 16;
    size_t gi = 0;
    for (int gy = 0; gy < patches_y; gy++) {
        for (int gx = 0; gx < patches_x; gx++, gi++) {
            double wave_stream[4] = {0, 0, 0, 0};
            double group_path = 0;
            if (pol.async_lanes) {
                const int per_wave = pol.pixels_per_wave;   // 64: one 8x8 tile per wave; 256: one wave streams the patch
                const int waves = 256 / per_wave;
                for (int w = 0; w < waves; w++) {
                    std::vector<LaneWork> lanes(64);
                    int n_in = 0;
                    for (int q = 0; q < per_wave; q++) {
                        const int s = w * per_wave + q;          // patch slot, tile-major as below
                        const int tw = s >> 6, l = s & 63;
                        const int x = gx * 16 + (tw & 1) * 8 + (l & 7), y = gy * 16 + (tw >> 1) * 8 + (l >> 3);
                        if (x >= W || y >= H)
                            continue;
                        n_in++;
                        for (int smp = 0; smp < spp; smp++) {
                            const size_t idx = ((size_t)y * W + x) * spp + smp;
                            lanes[q & 63].samples.push_back({bytes + offsets[idx], bytes + offsets[idx + 1]});
                        }
                    }
                    if (!n_in)
                        continue;
                    // env lookups are deferred: per pixel-sample, run together at full occupancy at the end
                    const double tail = c.env * ((n_in + 63) / 64) * spp;
                    t.other_w += tail;
                    t.other_l += c.env * n_in * spp;
                    wave_stream[w] = traverse_async(c, pol, lanes, t) + tail;
                }
                for (int w = 0; w < 4; w++)
                    wave_streams[gi * 4 + w] = wave_stream[w];
                group_paths[gi] = *std::max_element(wave_stream, wave_stream + 4);
                continue;
            }
            // the 256 pixel slots of the patch, wave-major
            int px[256], py[256];
            bool inside[256];
            for (int w = 0; w < 4; w++)
                for (int l = 0; l < 64; l++) {
                    const int s = w * 64 + l;
                    if (pol.interleave) {
                        px[s] = gx * 16 + 2 * (l & 7) + (w & 1);
                        py[s] = gy * 16 + 2 * (l >> 3) + (w >> 1);
                    } else {
                        px[s] = gx * 16 + (w & 1) * 8 + (l & 7);
                        py[s] = gy * 16 + (w >> 1) * 8 + (l >> 3);
                    }
                    inside[s] = px[s] < W && py[s] < H;
                }
            for (int smp = 0; smp < spp; smp++) {
                const uint8_t *cursor[256], *send[256];
                bool live[256];
                int n_inside[4] = {0, 0, 0, 0};
                for (int s = 0; s < 256; s++) {
                    live[s] = inside[s];
                    if (inside[s]) {
                        const size_t idx = ((size_t)py[s] * W + px[s]) * spp + smp;
                        cursor[s] = bytes + offsets[idx];
                        send[s] = bytes + offsets[idx + 1];
                        n_inside[s >> 6]++;
                    }
                }
                double phase_max = 0;
                for (int w = 0; w < 4; w++) {
                    if (!n_inside[w])
                        continue;
                    wave_stream[w] += c.gen;
                    t.other_w += c.gen;
                    t.other_l += c.gen * n_inside[w];
                    phase_max = c.gen;
                }
                group_path += phase_max;
                // slot -> ray mapping; with compaction the surviving rays are repacked before every traversal
                int order[256];
                int n_live = 0;
                for (int s = 0; s < 256; s++)
                    order[s] = s;
                for (int k = 0;; k++) {
                    Ray rays[256];
                    bool has[256];
                    int any = 0;
                    for (int s = 0; s < 256; s++) {
                        has[s] = false;
                        if (live[s]) {
                            has[s] = next_traversal(cursor[s], send[s], rays[s]);
                            live[s] = has[s];
                            any += has[s];
                        }
                    }
                    if (!any)
                        break;
                    // build the waves of this phase
                    std::vector<int> members[4];
                    if (pol.compact && k > 0) {
                        n_live = 0;
                        for (int s = 0; s < 256; s++)
                            if (has[s])
                                members[n_live++ >> 6].push_back(s);
                    } else {
                        for (int s = 0; s < 256; s++)
                            if (has[s])
                                members[s >> 6].push_back(s);
                    }
                    if (pol.epoch_turns > 0) {
                        // epochs: all waves run epoch_turns node turns, meet, and repack when the live rays fit in fewer waves
                        WaveRun wr[4];
                        int used = 0;
                        for (int w = 0; w < 4; w++) {
                            wr[w].n = (int)members[w].size();
                            for (int i = 0; i < wr[w].n; i++) {
                                wr[w].rays[i] = rays[members[w][i]];
                                wr[w].state[i] = 0;
                                wr[w].tests[i] = 0;
                            }
                            if (wr[w].n) {
                                used = w + 1;
                                const double fixed = c.setup + c.shade + (k > 0 ? c.compact : 0);
                                wave_stream[w] += fixed;
                                t.other_w += fixed;
                                t.other_l += fixed * wr[w].n;
                            }
                        }
                        double path = c.setup + c.shade;
                        for (;;) {
                            double emax = 0;
                            int alive_total = 0;
                            for (int w = 0; w < used; w++) {
                                const double e = run_epoch(c, pol, wr[w], pol.epoch_turns, t);
                                wave_stream[w] += e;
                                emax = std::max(emax, e);
                                for (int i = 0; i < wr[w].n; i++)
                                    alive_total += wr[w].state[i] != 2;
                            }
                            path += emax;
                            if (!alive_total)
                                break;
                            int waves_now = 0;
                            for (int w = 0; w < used; w++) {
                                int a = 0;
                                for (int i = 0; i < wr[w].n; i++)
                                    a += wr[w].state[i] != 2;
                                waves_now += a > 0;
                            }
                            // one pairwise merge per epoch: the two waves with the fewest live rays, when they fit in one
                            int alive_w[4] = {0, 0, 0, 0};
                            for (int w = 0; w < used; w++)
                                for (int i = 0; i < wr[w].n; i++)
                                    alive_w[w] += wr[w].state[i] != 2;
                            int a = -1, b = -1;
                            for (int w = 0; w < used; w++) {
                                if (!alive_w[w])
                                    continue;
                                if (a < 0 || alive_w[w] < alive_w[a]) {
                                    b = a;
                                    a = w;
                                } else if (b < 0 || alive_w[w] < alive_w[b]) {
                                    b = w;
                                }
                            }
                            (void)waves_now;
                            if (a >= 0 && b >= 0 && alive_w[a] + alive_w[b] <= 64 - pol.epoch_slack) {
                                // wave a (fewest) gives its rays to wave b; both drop their ended lanes
                                WaveRun merged;
                                for (int src : {b, a})
                                    for (int i = 0; i < wr[src].n; i++)
                                        if (wr[src].state[i] != 2) {
                                            merged.rays[merged.n] = wr[src].rays[i];
                                            merged.state[merged.n] = wr[src].state[i];
                                            merged.tests[merged.n] = wr[src].tests[i];
                                            merged.n++;
                                        }
                                wr[b] = merged;
                                wr[a] = WaveRun();
                                for (int w : {a, b}) {
                                    wave_stream[w] += c.compact;
                                    t.other_w += c.compact;
                                    t.other_l += c.compact * merged.n / 2;
                                }
                                path += c.compact;
                            }
                        }
                        group_path += path;
                        continue;
                    }
                    phase_max = 0;
                    for (int w = 0; w < 4; w++) {
                        double stream = 0;
                        if (pol.compact && k > 0 && n_inside[w]) {   // every wave of the group takes part in the exchange
                            stream += c.compact;
                            t.other_w += c.compact;
                            t.other_l += c.compact * members[w].size();
                        }
                        const int n = (int)members[w].size();
                        if (n) {
                            Ray lane_rays[64];
                            for (int i = 0; i < n; i++)
                                lane_rays[i] = rays[members[w][i]];
                            const double fixed = c.setup + c.shade;
                            stream += fixed;
                            t.other_w += fixed;
                            t.other_l += c.setup * n + c.shade * n;   // (an upper bound: only the lanes that hit shade)
                            stream += traverse(c, pol, lane_rays, n, t);
                        }
                        wave_stream[w] += stream;
                        phase_max = std::max(phase_max, stream);
                    }
                    group_path += phase_max;
                }
                phase_max = 0;
                for (int w = 0; w < 4; w++) {
                    if (!n_inside[w])
                        continue;
                    wave_stream[w] += c.env;
                    t.other_w += c.env;
                    t.other_l += c.env * n_inside[w];
                    phase_max = c.env;
                }
                group_path += phase_max;
            }
            if (pol.split_heavy > 0 && !pol.compact && spp == 1) {
                // replay heavy waves as four waves of 16 rays (quarters of the 8x8 tile, 4x4 pixels each)
                for (int w = 0; w < 4; w++) {
                    if (wave_stream[w] <= pol.split_heavy)
                        continue;
                    double worst = 0;
                    Totals scratch;
                    for (int q = 0; q < 4; q++) {
                        const uint8_t *cursor[16], *send[16];
                        bool live[16];
                        int n = 0;
                        for (int l = 0; l < 64; l++) {
                            const int lx = l & 7, ly = l >> 3;
                            if ((lx >> 2) + 2 * (ly >> 2) != q)
                                continue;
                            const int s2 = w * 64 + l;
                            live[n] = inside[s2];
                            if (inside[s2]) {
                                const size_t idx = ((size_t)py[s2] * W + px[s2]) * spp;
                                cursor[n] = bytes + offsets[idx];
                                send[n] = bytes + offsets[idx + 1];
                            }
                            n++;
                        }
                        double stream = c.gen + c.env;
                        for (;;) {
                            Ray lane_rays[64];
                            int m = 0;
                            for (int i = 0; i < 16; i++)
                                if (live[i]) {
                                    Ray r;
                                    live[i] = next_traversal(cursor[i], send[i], r);
                                    if (live[i])
                                        lane_rays[m++] = r;
                                }
                            if (!m)
                                break;
                            stream += c.setup + c.shade + traverse(c, pol, lane_rays, m, scratch);
                        }
                        worst = std::max(worst, stream);
                        t.other_w += stream;   // the extra waves' instructions (the original wave's stay counted: an upper bound)
                    }
                    wave_stream[w] = worst;
                }
            }
            for (int w = 0; w < 4; w++)
                wave_streams[gi * 4 + w] = wave_stream[w];
            group_paths[gi] = pol.compact ? group_path : *std::max_element(wave_stream, wave_stream + 4);
        }
    }
    t.wave_instr = t.node_w + t.leaf_w + t.other_w;
    t.lane_instr = t.node_l + t.leaf_l + t.other_l;
    out[0] = t.wave_instr;
    out[1] = t.lane_instr;
    out[2] = t.node_w;
    out[3] = t.node_l;
    out[4] = t.leaf_w;
    out[5] = t.leaf_l;
    out[6] = t.other_w;
    out[7] = t.other_l;
    out[8] = t.node_turns;
    out[9] = t.leaf_turns;
    out[10] = t.leaf_stages;
    out[11] = t.node_idle_parked; out[12] = t.node_idle_ended; out[13] = t.node_idle_empty;
    out[14] = t.leaf_idle_walk; out[15] = t.leaf_idle_ended; out[16] = t.leaf_idle_done; out[17] = t.leaf_idle_empty;
    return 0;
}
