// lio_mapadd.h -- launch interface of the Add_Points kernels (lio_mapadd.hip).
#pragma once
#include <hip/hip_runtime.h>
#include "lio_common.h"

namespace lio {

struct FoldArgs {
    float4*          map_pts;       // tomb-stones are written here
    const HashEntry* table;
    Grid             grid;
    float            ds;            // downsample_size (ikd_Tree.h:197)
    const float4*    new_pts;       // the batch, in the caller's order
    const uint32_t*  order;         // batch indices sorted by voxel key (stable)
    const uint32_t*  group_start;   // per voxel group: first position in `order`
    const uint32_t*  group_count;
    uint32_t         n_groups;
    float4*          out_pts;       // per group: survivor to append (w == 0) or nothing (w == 1)
    uint32_t*        n_added;       // += tmp_counter of ikd_Tree.cpp:448
};
struct IncrArgs {
    const float4*  body;            // feats_down_body
    int            nq, nq_stride;
    Pose14         pose;            // posterior state (state_point, laserMapping.cpp:2381)
    const float4*  nbr;             // Nearest_Points of the last search pass
    const uint8_t* cnt;
    double         filter_size_map_min;
    int            flg_EKF_inited;
    float4*        world;           // feats_down_world at the posterior pose
    uint8_t*       cls;             // 0 skip, 1 PointToAdd, 2 PointNoNeedDownsample
};
hipError_t launch_map_incremental(const IncrArgs& a, hipStream_t s);
hipError_t launch_voxel_keys(const float4* pts, uint32_t n, float ds, uint64_t* vkeys, uint32_t* order, hipStream_t s);
hipError_t launch_fold_groups(const FoldArgs& a, hipStream_t s);

}  // namespace lio
