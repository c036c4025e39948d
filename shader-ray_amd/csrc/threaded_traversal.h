// threaded_traversal.h -- the reference's stackless traversal, literally:
// follows the 8 direction-coded (hit, miss) link tables over the reference's
// own SoA arrays (raytracer.es.fs:386-443 group_intersect, :247-270 get_group,
// :200-217 slab test with true divisions, :297-346 triangle_intersect).
// Kept as kernel id 1: the slow, faithful form, and the cross-check for the
// packed stack kernel.
#pragma once

#include "trace_common.h"

namespace shray {

struct ThreadedTraversal {
#ifdef SHRAY_DIAGNOSTICS
    unsigned long long diag_tally[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif
    template <bool COUNT>
    __device__ __forceinline__ void closest(const SceneView &sc, const FrameView &fr, V3 P, V3 D, Hit &hit,
                                            RayCounters &rc)
    {
        if (COUNT)
            rc.traversals++;
        const float code = ((D.x > 0.0f) ? 1.0f : 0.0f) + ((D.y > 0.0f) ? 2.0f : 0.0f) + ((D.z > 0.0f) ? 4.0f : 0.0f);
        const float offset = code * (float)sc.table_stride;   // fs:392
        const float max_leaf = (float)fr.max_leaf_tests;
        float g = sc.tree_root;

        for (int i = 0; i < fr.max_bvh_iterations; i++) {
            if (COUNT)
                rc.node_visits++;
            const unsigned int node = (unsigned int)g;
            const unsigned int link = (unsigned int)(g + offset);
            const float hit_next = sc.hitmiss[2u * link];
            const float miss_next = sc.hitmiss[2u * link + 1u];
            const bool leaf = hit_next == miss_next;
            float start = 0.0f, count = 0.0f;
            if (leaf) {
                start = sc.objects[2u * node];
                count = sc.objects[2u * node + 1u];
                if (COUNT)
                    rc.leaf_visits++;
            }

            // range_intersect_box against [0, 1e8], fs:200-217
            float r0 = 0.0f, r1 = kRangeMax;
            {
                const float lo[3] = {sc.boxmin[3u * node], sc.boxmin[3u * node + 1u], sc.boxmin[3u * node + 2u]};
                const float hi[3] = {sc.boxmax[3u * node], sc.boxmax[3u * node + 1u], sc.boxmax[3u * node + 2u]};
                const float o[3] = {P.x, P.y, P.z};
                const float d[3] = {D.x, D.y, D.z};
#pragma unroll
                for (int a = 0; a < 3; a++) {
                    const float ta = (lo[a] - o[a]) / d[a];
                    const float tb = (hi[a] - o[a]) / d[a];
                    const bool forward = d[a] >= 0.0f;
                    r0 = sel_max(r0, forward ? ta : tb);
                    r1 = sel_min(r1, forward ? tb : ta);
                }
            }

            if (!(r0 >= r1) && (r0 < hit.t)) {
                if (leaf) {
                    for (float j = 0.0f; j < max_leaf; j++) {
                        if (j >= count)
                            break;
                        if (COUNT)
                            rc.triangle_tests++;
                        const float which = start + j;
                        const float *v = sc.positions + 9u * (unsigned int)which;
                        const V3 v0 = mk(v[0], v[1], v[2]), v1 = mk(v[3], v[4], v[5]), v2 = mk(v[6], v[7], v[8]);
                        const V3 e0 = v1 - v0, e1 = v0 - v2;
                        const V3 M = cross3(e1, D);
                        const float det = dot3(e0, M);
                        if (det > -0.0000001f && det < 0.0000001f)
                            continue;
                        const float inv_det = 1.0f / det;
                        const V3 T = P - v0;
                        const V3 Q = cross3(T, e0);
                        const float dist = -dot3(e1, Q) * inv_det;
                        if (dist > hit.t || dist < r0 || dist > r1)
                            continue;
                        const float u = dot3(T, M) * inv_det;
                        if (u < 0.0f || u > 1.0f)
                            continue;
                        const float w = dot3(D, Q) * inv_det;
                        if (w < 0.0f || u + w > 1.0f)
                            continue;
                        hit.which = which;
                        hit.t = dist;
                        hit.bu = u;
                        hit.bv = w;
                    }
                }
                g = hit_next;
            } else {
                g = miss_next;
            }
            if (g >= kTerminator)
                return;
            if (i == fr.max_bvh_iterations - 1)
                hit.t = -1.0f;   // set_bad_hit, fs:436-438
        }
    }
};

}   // namespace shray
