// packed_layout.h -- the MI355X-side BVH layout, built once per scene.
//
// The reference keeps boxes, links and leaf ranges in five float textures and
// threads the tree eight times (world.cpp:231-288).  For the GPU the same tree
// is repacked so that one node visit is two 16-byte loads and one triangle
// test is three, and the eight link tables collapse into "push the far child":
//
//   PackedNode (32 B, depth-first order, negative subtree first), as the host builds it:
//     lo = { boxmin.xyz, a }      hi = { boxmax.xyz, b }
//     branch: a = split_axis << 30 | positive_child      b = negative_child       (indices)
//     leaf  : a = first triangle                          b = 0x80000000 | count
//   On the device the array exists EIGHT times, once per direction octant (round 4; bit k of an octant: D[k] >= 0, the
//   predicate range_intersect_box picks a box's entry plane by, fs:204-213).  Copy o holds, node for node at the same offset,
//     lo = { the planes a ray of octant o ENTERS the box by, a' }     hi = { the planes it LEAVES by, b' }
//     (in memory as DeviceNode below: { entry.x, entry.y, exit.x, exit.y } { entry.z, exit.z, a', b' })
//     branch: a' = 1 << (29 + split_axis) | name of the child such a ray visits FIRST      b' = name of the other child
//     leaf  : as above
//   where a node's NAME is its byte offset in a copy / 8 (below 2^23: shray_scene_create admits 2^21 nodes).  A ray reads
//   the copy of its own octant -- lanes of one wave read different copies -- so a visit selects nothing: not the six
//   planes (six v_cndmask), not the child (an AND, a compare, two more v_cndmask); its record's address is
//   (name << 3) + octant * bytes-per-copy, one instruction, the shift dropping the axis bit of a'.  The reference orders the
//   children by D[axis] > 0 (world.cpp:259-265 with get_coded_dir, :214-220); that and D[axis] >= 0 differ for a component
//   that is zero, which the visit corrects on the path its division takes anyway (wave_traversal.h: visit_decision).
//   Eight copies cost memory (bunny-class: 8 x 0.64 MB; the 1M-triangle scene: 8 x 9.3 MB) and bought 2.4 ... 3.4 % on every
//   BASELINE configuration, that one included (profiles/r04/octant_copies_ab.txt).
//   PackedTri (36 B, same triangle order as the reference arrays; three 12-byte loads per test)
//     { v0.xyz } { e0.xyz } { e1.xyz }   e0 = v1 - v0, e1 = v0 - v2
//     (48-byte records read as three dwordx4 measure 1.5-2 % slower: profiles/history/r02/leaf_stage_ab.txt)
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

// A record of an octant copy as it lies in device memory (round 5): the same eight words as PackedNode, ordered so that the two
// 16-byte loads of a visit leave { entry.x, entry.y } { exit.x, exit.y } { entry.z, exit.z } in three aligned register pairs --
// the operands of the visit's six packed fp32 instructions (variants/packed_slab.h: slab_range_fast), where PackedNode's order took
// twelve plain ones.  load_packed_node hands the words back in PackedNode's order to everything else.
struct DeviceNode {
    float entry_xy[2];
    float exit_xy[2];
    float z[2];          // { entry.z, exit.z }
    uint32_t a, b;
};
static_assert(sizeof(DeviceNode) == 32, "DeviceNode must be 32 bytes");

struct PackedTri {
    float v0[3];
    float e0[3];
    float e1[3];
};
static_assert(sizeof(PackedTri) == 36, "PackedTri must be 36 bytes");

// Sibling pairs for the pair traversal (variants/pair_traversal.h: inner_stage_pair): the record of inner node N holds the
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
constexpr uint32_t kChildNameMask = 0x1fffffffu;     // device form: a' = 1 << (29 + axis) | name (byte offset / 8)
constexpr uint32_t kAxisHotShift = 29;
constexpr uint32_t kNodeNameShift = 3;               // name -> byte offset
constexpr uint32_t kNodeShift = 5;                   // log2(sizeof(PackedNode))
constexpr uint32_t kNoNode = 0xffffffffu;

}   // namespace shray
