// host-log.h -- progress chatter of the loaders (the reference prints its
// timings and statistics unconditionally on stderr; here it can be muted).
#pragma once

#include <cstdio>
#include <system_error>
#include <thread>
#include <vector>

extern bool g_host_quiet;

#define host_info(...)                        \
    do {                                      \
        if (!g_host_quiet)                    \
            fprintf(stderr, __VA_ARGS__);     \
    } while (0)

// Threads the loaders use for a file (parsing in chunks, vertex de-duplication, synthesized normals): the machine's, at
// most 32; SHRAY_LOAD_THREADS overrides (1 = everything on the calling thread, the way the reference does it; at most 256).
int host_load_threads();

// fn(j) for j = 0 .. jobs - 1, job 0 on the calling thread
template <class F>
void host_in_parallel(int jobs, F &&fn);

template <class F>
void host_in_parallel(int jobs, F &&fn)
{
    // A thread the system refuses (RLIMIT_NPROC, a cgroup's pids limit: this is a C API loaded into other people's processes)
    // must not end the process -- destroying the joinable threads already started would call std::terminate --: the jobs
    // that got no thread run on the calling thread, the ones that did are joined.
    std::vector<std::thread> pool;
    int started = 1;
    for (; started < jobs; started++) {
        try {
            pool.emplace_back([&fn, started] { fn(started); });
        } catch (const std::system_error &) {
            break;
        }
    }
    fn(0);
    for (int j = started; j < jobs; j++)
        fn(j);
    for (std::thread &th : pool)
        th.join();
}
