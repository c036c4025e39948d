// host-log.h -- progress chatter of the loaders (the reference prints its
// timings and statistics unconditionally on stderr; here it can be muted).
#pragma once

#include <cstdio>
#include <thread>
#include <vector>

extern bool g_host_quiet;

#define host_info(...)                        \
    do {                                      \
        if (!g_host_quiet)                    \
            fprintf(stderr, __VA_ARGS__);     \
    } while (0)

// Threads the loaders use for a file (parsing in chunks, vertex de-duplication, synthesized normals): the machine's, at
// most 32; SHRAY_LOAD_THREADS overrides (1 = everything on the calling thread, the way the reference does it).
int host_load_threads();

// fn(j) for j = 0 .. jobs - 1, job 0 on the calling thread
template <class F>
void host_in_parallel(int jobs, F &&fn);

template <class F>
void host_in_parallel(int jobs, F &&fn)
{
    std::vector<std::thread> pool;
    for (int j = 1; j < jobs; j++)
        pool.emplace_back([&fn, j] { fn(j); });
    fn(0);
    for (std::thread &th : pool)
        th.join();
}
