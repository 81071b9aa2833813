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

// key[i] = order-preserving image of pts[i].w (time offset), order[i] = i
hipError_t launch_time_keys(const float4* pts, int n, uint64_t* keys, uint32_t* order, hipStream_t s);
hipError_t launch_undistort(const float4* raw, int n, const double* poses, int n_poses, const Pose14& end, float4* out,
                            hipStream_t s);
hipError_t launch_minmax(const float4* pts, int n, int* mm6, hipStream_t s);
float minmax_decode(int ordered);
hipError_t launch_voxel_index(const float4* pts, int n, const VoxelGridParams& g, uint64_t* keys, uint32_t* order, hipStream_t s);
hipError_t launch_centroids(const float4* pts, const uint32_t* order, const uint64_t* leaf_key, const uint32_t* leaf_start,
                            const uint32_t* leaf_count, uint32_t n_leaves, float4* out, hipStream_t s);

}  // namespace lio
