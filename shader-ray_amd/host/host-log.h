// host-log.h -- progress chatter of the loaders (the reference prints its
// timings and statistics unconditionally on stderr; here it can be muted).
#pragma once

#include <cstdio>

extern bool g_host_quiet;

#define host_info(...)                        \
    do {                                      \
        if (!g_host_quiet)                    \
            fprintf(stderr, __VA_ARGS__);     \
    } while (0)
