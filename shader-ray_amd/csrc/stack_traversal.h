// stack_traversal.h -- per-ray stack traversal over the packed layout
// (packed_layout.h).  Same visits, same order, same arithmetic as the
// reference's threaded traversal (raytracer.es.fs:386-443); what changes is
// where the data comes from:
//   * one node = two dwordx4 loads (box + links) instead of 3-4 texture fetches
//   * the ray's pending far children live in LDS, laid out [level][thread] so
//     that every lane always addresses its own bank (no conflicts at any mix
//     of depths)
//   * triangle edges are precomputed
//   * the slab test's six true divisions (fs:204-213) become multiplications by the
//     ray's correctly rounded reciprocal plus two FMA refinements that land on the
//     same correctly rounded quotient (exact_div.h); rays or scenes outside the
//     proven operand ranges keep dividing
#pragma once

#include "exact_div.h"
#include "packed_layout.h"
#include "trace_common.h"

namespace shray {

template <int BLOCK>
struct StackTraversal {
    uint32_t *stack;   // LDS, this thread's column: stack[level * BLOCK]

    // group_intersect on an object-space ray.  The slab test needs the six
    // quotients exactly as true division rounds them; when the operand ranges
    // allow it (exact_div.h) every lane of the wave uses the hoisted-reciprocal
    // form, otherwise the whole wave divides.
    template <bool COUNT>
    __device__ __forceinline__ void closest(const SceneView &sc, const FrameView &fr, V3 P, V3 D, Hit &hit,
                                            RayCounters &rc)
    {
        const bool ranges_ok = sc.exact_div_ok && divisor_in_range(D.x) && divisor_in_range(D.y) &&
                               divisor_in_range(D.z) && coordinate_in_range(P.x) && coordinate_in_range(P.y) &&
                               coordinate_in_range(P.z);
        if (__builtin_amdgcn_ballot_w64(!ranges_ok) == 0ull)
            walk<COUNT, true>(sc, fr, P, D, hit, rc);
        else
            walk<COUNT, false>(sc, fr, P, D, hit, rc);
    }

    template <bool COUNT, bool HOISTED>
    __device__ __forceinline__ void walk(const SceneView &sc, const FrameView &fr, V3 P, V3 D, Hit &hit,
                                         RayCounters &rc)
    {
        if (COUNT)
            rc.traversals++;
        const float4 *__restrict__ nodes = reinterpret_cast<const float4 *>(sc.packed_nodes);
        const float4 *__restrict__ tris = reinterpret_cast<const float4 *>(sc.packed_tris);
        // bit k set <=> the negative child is nearer along axis k
        const uint32_t positive_dir = (D.x > 0.0f ? 1u : 0u) | (D.y > 0.0f ? 2u : 0u) | (D.z > 0.0f ? 4u : 0u);
        const uint32_t max_leaf = (uint32_t)fr.max_leaf_tests;
        const bool fx = D.x >= 0.0f, fy = D.y >= 0.0f, fz = D.z >= 0.0f;
        V3 Y = mk(0, 0, 0);
        if (HOISTED)
            Y = mk(1.0f / D.x, 1.0f / D.y, 1.0f / D.z);

        uint32_t node = sc.packed_root;
        int sp = 0;
        for (int i = 0; i < fr.max_bvh_iterations; i++) {
            if (COUNT)
                rc.node_visits++;
            const float4 lo = nodes[2u * node];
            const float4 hi = nodes[2u * node + 1u];
            const uint32_t a = __float_as_uint(lo.w), b = __float_as_uint(hi.w);

            // range_intersect_box against [0, 1e8] (fs:200-217): entry plane is the box's
            // low side when D >= 0, else its high side
            float r0, r1;
            if (HOISTED) {
                const float nx = div_by_constant((fx ? lo.x : hi.x) - P.x, D.x, Y.x);
                const float ny = div_by_constant((fy ? lo.y : hi.y) - P.y, D.y, Y.y);
                const float nz = div_by_constant((fz ? lo.z : hi.z) - P.z, D.z, Y.z);
                const float ux = div_by_constant((fx ? hi.x : lo.x) - P.x, D.x, Y.x);
                const float uy = div_by_constant((fy ? hi.y : lo.y) - P.y, D.y, Y.y);
                const float uz = div_by_constant((fz ? hi.z : lo.z) - P.z, D.z, Y.z);
                // all six are finite here, so the hardware min/max equal GLSL's select forms
                r0 = fmaxf(fmaxf(fmaxf(0.0f, nx), ny), nz);
                r1 = fminf(fminf(fminf(kRangeMax, ux), uy), uz);
            } else {
                r0 = 0.0f;
                r1 = kRangeMax;
                const float tx0 = (lo.x - P.x) / D.x, tx1 = (hi.x - P.x) / D.x;
                r0 = sel_max(r0, fx ? tx0 : tx1);
                r1 = sel_min(r1, fx ? tx1 : tx0);
                const float ty0 = (lo.y - P.y) / D.y, ty1 = (hi.y - P.y) / D.y;
                r0 = sel_max(r0, fy ? ty0 : ty1);
                r1 = sel_min(r1, fy ? ty1 : ty0);
                const float tz0 = (lo.z - P.z) / D.z, tz1 = (hi.z - P.z) / D.z;
                r0 = sel_max(r0, fz ? tz0 : tz1);
                r1 = sel_min(r1, fz ? tz1 : tz0);
            }

            if (COUNT && (b & kLeafFlag))
                rc.leaf_visits++;   // the reference fetches (start, count) before the box test, fs:263-267
            uint32_t next = kNoNode;
            bool descend = false;
            if (!(r0 >= r1) && (r0 < hit.t)) {
                if (b & kLeafFlag) {
                    const uint32_t count = min(b & ~kLeafFlag, max_leaf);
                    for (uint32_t j = 0; j < count; j++) {
                        if (COUNT)
                            rc.triangle_tests++;
                        const uint32_t which = a + j;
                        const float4 q0 = tris[3u * which], q1 = tris[3u * which + 1u], q2 = tris[3u * which + 2u];
                        const V3 v0 = mk(q0.x, q0.y, q0.z), e0 = mk(q0.w, q1.x, q1.y), e1 = mk(q1.z, q1.w, q2.x);
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
                        hit.which = (float)which;
                        hit.t = dist;
                        hit.bu = u;
                        hit.bv = w;
                    }
                } else {
                    const uint32_t axis = a >> 30;
                    const uint32_t pos_child = a & kChildMask, neg_child = b;
                    const bool neg_first = (positive_dir >> axis) & 1u;
                    next = neg_first ? neg_child : pos_child;
                    stack[sp * BLOCK] = neg_first ? pos_child : neg_child;
                    sp++;
                    descend = true;
                }
            }
            if (!descend) {
                if (sp == 0)
                    return;
                sp--;
                next = stack[sp * BLOCK];
            }
            node = next;
            if (i == fr.max_bvh_iterations - 1)
                hit.t = -1.0f;   // set_bad_hit, fs:436-438
        }
    }
};

}   // namespace shray
