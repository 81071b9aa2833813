// lio_scanprep.h -- launch interface of the scan preparation kernels (lio_scanprep.hip).
#pragma once
#include <hip/hip_runtime.h>
#include "lio_common.h"

namespace lio {

struct VoxelGridParams {            // pcl::VoxelGrid state after its first pass
    float inv_leaf;                 // inverse_leaf_size_
    int   min_b[3];                 // min_b_
    int   divb_mul[3];              // divb_mul_
};
// Device-resident working record of one voxel-filter call: nothing of it visits the host before the single
// read-back of `summary` at the end.
struct VoxelGridDev {
    uint32_t mm[6];                 // getMinMax3D as atomicMax targets: ~u(min xyz), u(max xyz), u = order-preserving image; 0 = unset
    VoxelGridParams g;
    uint32_t flags;                 // VOX_EMPTY | VOX_OVERFLOW
    uint32_t n_leaves;              // runs of equal leaf index (the run of dropped points included)
    uint32_t summary[4];            // {n_leaves, 1 if the last run is the dropped one, flags, != 0: a sort / compaction wait ran out}
    // the fast path of time-ordered sweeps (lio_scan_voxel_filter): what the host asks for before it picks the sort ...
    uint32_t probe[2];              // {1: some point is earlier than its predecessor, bits of the leaf index (2^bits > number of leaves)}
    uint32_t arrive;                // workgroups of minmax_kernel that have delivered their extremes (the last one derives `g`)
    // ... and the digit histograms of the compact leaf keys for the LSD sort (lio_gsort.hip), zeroed with the record
    uint32_t hist[4][256];
};
constexpr uint32_t VOX_EMPTY = 1u, VOX_OVERFLOW = 2u;

// feats_down_body straight out of the node's host cloud: one lane per record PULLS x, y, z (the first three floats of a record of
// `stride_bytes`, e.g. pcl::PointXYZINormal's 48) over PCIe from page-locked host memory through its device alias and writes the
// float4 scan buffer -- no host-side pack, no staging copy, no copy-engine latency (scripts/micro/upload_probe.hip: 8.6 k records
// 16 us against 26 us for pack + copy; 100 k records 103-124 us against 90-200 us, with NO host time)
hipError_t launch_scan_pull(const void* src_device_alias, int stride_bytes, int n, float4* dst, hipStream_t s);
// key[i] = order-preserving image of pts[i].w (time offset), order[i] = i
hipError_t launch_time_keys(const float4* pts, int n, uint64_t* keys, uint32_t* order, hipStream_t s);
// parts (optional): [ceil(n / 256)][8] the extremes of every workgroup's compensated points + its time-order flag (lio_scanprep.hip)
hipError_t launch_undistort(const float4* raw, int n, const double* poses, int n_poses, const Pose14& end, float4* out,
                            hipStream_t s, uint32_t* parts = nullptr);
// getMinMax3D and, by the last workgroup to arrive, min_b_ / divb_mul_ / the width of the leaf index (vox->g, flags, probe[1]);
// check_time_order: also sets vox->probe[0] when some point's time key is below its predecessor's
hipError_t launch_minmax_params(const float4* pts, int n, VoxelGridDev* vox, float inv_leaf, bool check_time_order, hipStream_t s);
hipError_t launch_voxel_index(const float4* pts, int n, const VoxelGridDev* vox, bool time_minor, uint64_t* keys, uint32_t* order,
                              hipStream_t s);
// the compact form for clouds already in time order: key = leaf index (< 2^leaf_bits; non-finite points: 2^leaf_bits - 1, beyond
// every leaf), digit histograms of the keys into vox->hist; a stable sort by this key is the sort by (leaf, time, input order)
// parts / n_parts (optional): launch_undistort's records of this cloud -- no launch_minmax_params ran, the kernel folds them and derives
// the grid itself (vox->g, flags, probe as launch_minmax_params leaves them)
hipError_t launch_voxel_index_compact(const float4* pts, int n, VoxelGridDev* vox, uint64_t* keys, hipStream_t s, const uint32_t* parts = nullptr,
                                      uint32_t n_parts = 0, float inv_leaf = 0.0f, bool check_time_order = false);
// centroids of the runs [leaf_start[l], leaf_start[l + 1]) of the leaf-sorted order, written to out[l] (l < cap_out);
// launched for at most max_leaves runs, the run count is read from vox->n_leaves
// zero_for_next_call: another record, cleared here for the next call (no memset launch in front of it);
// compact_keys: the keys are launch_voxel_index_compact's (the run of dropped points carries 2^leaf_bits - 1 instead of a leaf index
// of all ones in the upper half)
hipError_t launch_centroids(const float4* pts, const uint32_t* order, const uint64_t* sorted_keys, const uint32_t* leaf_start,
                            VoxelGridDev* vox, uint32_t n, uint32_t max_leaves, uint32_t cap_out, float4* out, hipStream_t s,
                            bool compact_keys = false, VoxelGridDev* zero_for_next_call = nullptr);

// run heads + centroids of a leaf-sorted cloud with COMPACT keys in one launch (n <= 262144; agg / stamp: the stamped scratch of the
// single-launch compactions, lio_tile.h); vox->n_leaves / summary as launch_run_heads + launch_centroids leave them
hipError_t launch_leaf_centroids(const float4* pts, const uint32_t* order, const uint64_t* sorted_keys, VoxelGridDev* vox, uint32_t n, uint32_t cap_out, float4* out,
                                 VoxelGridDev* zero_for_next_call, unsigned long long* agg, uint32_t stamp, hipStream_t s);

}  // namespace lio
