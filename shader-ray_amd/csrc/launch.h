// launch.h -- host-side launchers of the gfx950 kernels.
#pragma once

#include <hip/hip_runtime.h>

#include "device_types.h"

namespace shray {

#ifdef SHRAY_DIAGNOSTICS
extern bool g_diag_plain_kernel;   // diagnostic build: run the non-counting kernel even with a counters buffer
#endif

// counters == nullptr selects the plain (timed) kernel, otherwise the counting variant.
hipError_t launch_threaded(const SceneView &sc, const FrameView &fr, float4 *out, DeviceCounters *counters,
                           hipStream_t stream);
// stack_levels = capacity of the per-ray stack (the tree's depth)
hipError_t launch_stack(const SceneView &sc, const FrameView &fr, float4 *out, DeviceCounters *counters,
                        hipStream_t stream, int stack_levels);

// `count` frames in one launch (grid.y = frame): d_frames[k] is frame k's view, written to
// out + k * frame_stride (in float4 units); `first` is frame 0's view (grid shape, differential class)
// all_metal: every frame's diffuse colour is zero (selects the kernel instance without the diffuse branch)
// all_plain: every frame is a plain which == 0 frame (the convergent driver applies); deal: the instances that deal
// leaf triangles to idle lanes (leaf_stage.h) instead of the plain leaf loop
// tally: nullptr for the timed launches; else the same instance with per-ray work tallies added into `tally`
// (kCounterShards copies) -- what the timed form does (shray_render_counters_timed)
hipError_t launch_stack_batch(const SceneView &sc, const FrameView *d_frames, int count, const FrameView &first, bool all_metal,
                              bool all_plain, bool deal, float4 *out, size_t frame_stride, hipStream_t stream, int stack_levels,
                              DeviceCounters *tally = nullptr, bool pair = false, bool tally_full_walk = false, bool ordered = false);
// ordered: the frames carry a dispatch order / cost buffer (capi.hip: DispatchOrder): the instances that read them
// pair: the instances that test both children of a node per turn (variants/pair_traversal.h: inner_stage_pair; the scene needs
// pair_nodes); tally_full_walk: with `tally` and `pair`, the tallies of the reference's walk instead of the timed form's

// kernel id 2 (kernel_pool.hip): the workgroup's rays as a pool whose waves merge during the traversal;
// which == 0 frames only.  Same argument conventions as the stack kernel's launchers.
hipError_t launch_pool(const SceneView &sc, const FrameView &fr, float4 *out, DeviceCounters *counters, hipStream_t stream,
                       int stack_levels);
hipError_t launch_pool_batch(const SceneView &sc, const FrameView *d_frames, int count, const FrameView &first, bool all_metal,
                             float4 *out, size_t frame_stride, hipStream_t stream, int stack_levels);

// kernel id 4 (kernel_wavefront.hip): trace() split by bounce, one launch per bounce with the live paths compacted in
// between; whole plain frames only.  d_view: the frame's view in device memory; queue0 / queue1: width * height * spp
// path records of 64 bytes each; counts: bounce_count + 2 words; radiance: width * height * spp float4 (spp > 1 only)
struct PathState;
hipError_t launch_wavefront(const SceneView &sc, const FrameView *d_view, const FrameView &fr, bool metal, PathState *queue0, PathState *queue1,
                            unsigned int *counts, float4 *radiance, float4 *out, hipStream_t stream, int stack_levels);

// Dispatch order of the convergent batch kernels (kernel_assemble.hip): from the running times the waves left in
// cost[0 .. n) (uniform_driver.h; cost[n .. 2n) is the kernel's scratch) writes the permutation order[0 .. n) -- patches by descending cost in 32 classes of
// cost / max cost, patch order kept inside a class -- and halves every cost, so that
// a patch stays where its last heavy frame put it until newer frames say otherwise.  One workgroup.
// bulk_class: classes from this one on (costs below (16 - bulk_class) / 16 of the largest) count as one.
hipError_t launch_dispatch_order(uint32_t *cost, uint32_t *order, uint32_t n, hipStream_t stream, int bulk_class);

// rank 0's de-interleave (kernel_assemble.hip); strides in floats: rank_stride between ranks' buffers,
// frame_stride between a rank's consecutive frames
// (c0, c1): phases per period owned by rank 0 / by every other rank (1, 1 = even split)
hipError_t launch_assemble_tiles(const float *gathered, float4 *out, int world, int c0, int c1, int frames, int channels,
                                 int width, int height, int tile_w, int tile_h, size_t rank_stride, size_t frame_stride,
                                 hipStream_t stream);

// Compares div_by_constant (exact_div.h) with true division on `pairs` pseudo-random
// operand pairs drawn from the admitted ranges; *mismatches receives the count.
hipError_t launch_division_selftest(unsigned long long pairs, unsigned long long seed,
                                    unsigned long long *mismatches, hipStream_t stream);

// the memory-pipeline probe behind shray_probe_vector_cache (kernel_selftest.hip); records and spread are powers of two
hipError_t launch_vector_cache_probe(const float4 *table, uint32_t records, uint32_t spread, uint32_t visits, uint32_t waves, unsigned long long lanes,
                                     int bytes_per_lane, float *out, hipStream_t stream);

// reciprocal_in_range (exact_div.h) against 1.0f / x on every float of its domain; *mismatches receives the count.
hipError_t launch_reciprocal_selftest(unsigned long long *mismatches, hipStream_t stream);

}   // namespace shray
