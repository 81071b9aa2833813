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
};
constexpr uint32_t VOX_EMPTY = 1u, VOX_OVERFLOW = 2u;

// key[i] = order-preserving image of pts[i].w (time offset), order[i] = i
hipError_t launch_time_keys(const float4* pts, int n, uint64_t* keys, uint32_t* order, hipStream_t s);
hipError_t launch_undistort(const float4* raw, int n, const double* poses, int n_poses, const Pose14& end, float4* out,
                            hipStream_t s);
hipError_t launch_minmax(const float4* pts, int n, VoxelGridDev* vox, hipStream_t s);
hipError_t launch_voxel_params(VoxelGridDev* vox, float inv_leaf, hipStream_t s);
hipError_t launch_voxel_index(const float4* pts, int n, const VoxelGridDev* vox, bool time_minor, uint64_t* keys, uint32_t* order,
                              hipStream_t s);
// centroids of the runs [leaf_start[l], leaf_start[l + 1]) of the leaf-sorted order, written to out[l] (l < cap_out);
// launched for at most max_leaves runs, the run count is read from vox->n_leaves
hipError_t launch_centroids(const float4* pts, const uint32_t* order, const uint64_t* sorted_keys, const uint32_t* leaf_start,
                            VoxelGridDev* vox, uint32_t n, uint32_t max_leaves, uint32_t cap_out, float4* out, hipStream_t s);

}  // namespace lio
