// packed_layout.h -- the MI355X-side BVH layout, built once per scene.
//
// The reference keeps boxes, links and leaf ranges in five float textures and
// threads the tree eight times (world.cpp:231-288).  For the GPU the same tree
// is repacked so that one node visit is two 16-byte loads and one triangle
// test is three, and the eight link tables collapse into "push the far child":
//
//   PackedNode (32 B, depth-first order, negative subtree first)
//     lo = { boxmin.xyz, a }      hi = { boxmax.xyz, b }
//     branch: a = (1 << (29 + split_axis)) | positive_child      b = negative_child
//     leaf  : a = first triangle                                  b = 0x80000000 | count
//     A child is named by its BYTE OFFSET in the node array (index * 32: the address of a visit's loads without a shift),
//     the split axis by one of the three top bits (a ray's "positive direction" bits sit there too: which child comes
//     first is one AND and one comparison).  shray_scene_create builds the tree with indices and a two-bit axis
//     (a = axis << 30 | index: what TreeBuilder::pack and the pair records' builder read) and re-encodes it for the device.
//   PackedTri (36 B, same triangle order as the reference arrays; three 12-byte loads per test)
//     { v0.xyz } { e0.xyz } { e1.xyz }   e0 = v1 - v0, e1 = v0 - v2
//     (48-byte records read as three dwordx4 measure 1.5-2 % slower: profiles/r02/leaf_stage_ab.txt)
//
// e0 / e1 are the same single fp32 subtractions triangle_intersect performs
// per test (raytracer.es.fs:304-305), hoisted to scene-creation time.
// Visit order: a ray whose direction component along the split axis is > 0
// descends into the negative child first, otherwise into the positive child
// (world.cpp:259-265 with get_coded_dir, :214-220); the far child is pushed on
// the ray's stack; "miss" = pop.  That reproduces the reference's eight
// threaded orders exactly, which shray_scene_create verifies table by table
// before it lets the stack kernel run.
#pragma once

#include <stdint.h>

namespace shray {

struct PackedNode {
    float lo[3];
    uint32_t a;
    float hi[3];
    uint32_t b;
};
static_assert(sizeof(PackedNode) == 32, "PackedNode must be 32 bytes");

struct PackedTri {
    float v0[3];
    float e0[3];
    float e1[3];
};
static_assert(sizeof(PackedTri) == 36, "PackedTri must be 36 bytes");

// Sibling pairs for the pair traversal (wave_traversal.h: inner_stage_pair): the record of inner node N holds the
// boxes of BOTH its children, so that one memory round trip serves two of the reference's visits.  64 B, indexed like
// PackedNode (only inner nodes' records are read):
//     { neg.boxmin, neg.link } { neg.boxmax, neg.info } { pos.boxmin, pos.link } { pos.boxmax, pos.info }
//   link = child index (bits 21:0) | min(triangle count, 127) << 22 (leaf) | the child's split axis << 29 (branch)
//          | 0x80000000 (leaf);   info = the leaf's first triangle
// A ray's stack word for a pending far child is the low kPairLinkBits' worth of that link squeezed to
// (index | axis << IB | leaf << (IB + 2)), IB = bits of the largest node index, with the child's box entry distance r0
// truncated to the remaining 29 - IB high bits above it (all ones: the box range was empty) -- see lane_pop.
struct PackedPair {
    float lo0[3];
    uint32_t link0;
    float hi0[3];
    uint32_t info0;
    float lo1[3];
    uint32_t link1;
    float hi1[3];
    uint32_t info1;
};
static_assert(sizeof(PackedPair) == 64, "PackedPair must be 64 bytes");
constexpr uint32_t kPairIndexMask = 0x003fffffu;
constexpr uint32_t kPairCountShift = 22, kPairCountMask = 0x7fu, kPairAxisShift = 29;

constexpr uint32_t kLeafFlag = 0x80000000u;
constexpr uint32_t kChildMask = 0x3fffffffu;         // host form: a = axis << 30 | index
constexpr uint32_t kChildOffsetMask = 0x1fffffffu;   // device form: a = 1 << (29 + axis) | byte offset
constexpr uint32_t kAxisHotShift = 29;
constexpr uint32_t kNodeShift = 5;                   // log2(sizeof(PackedNode))
constexpr uint32_t kNoNode = 0xffffffffu;

}   // namespace shray
