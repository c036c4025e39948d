// device_tree_internal.h -- what flatten.hip and capi.hip read of the objects bvh_build.hip and flatten.hip keep on the device
// (the device-resident scene pipeline: shray_bvh_build_device -> shray_flatten_device_tree -> shray_scene_create_from_device).
// Internal to libshray_hip.so: plain structs of device pointers, filled by two functions that are not part of the C ABI.
#pragma once

#include "shader_ray_hip.h"

// The tree a shray_device_tree holds, as the pre-order arrays of shray_tree_desc -- on the DEVICE.
struct ShrayDeviceTreeView {
    int node_count, triangle_count, vertex_count, vertex_stride_floats, max_level;
    const int *parent, *negative, *positive, *start, *triangles;
    const float *box, *direction;
    const int *triangle_vertices;   // 3 per triangle, post-build order
    const float *vertex_data;       // vertex_stride_floats per vertex
};

// The flattened arrays a shray_device_flat holds (desc: device pointers), and where every pre-order node went (in-order numbers).
struct ShrayDeviceFlatView {
    shray_scene_desc desc;
    const int *index_of;            // [node_count]: pre-order -> in-order (world.cpp:145-177)
};

extern "C" int shrayi_device_tree_view(const shray_device_tree *tree, ShrayDeviceTreeView *view);
extern "C" int shrayi_device_flat_view(const shray_device_flat *flat, ShrayDeviceFlatView *view);
