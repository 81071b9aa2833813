// lio_incr.h -- map_incremental's per-point decision (src/laserMapping.cpp:1448-1489), shared by the stand-alone
// classification kernel (lio_mapadd.hip) and the fused classification + list compaction (lio_compact.hip).
#pragma once
#include "lio_device.h"
#include "lio_mapadd.h"
#include "lio_math.h"

namespace lio {

// World point at the POSTERIOR pose (pointBodyToWorld :1451), then skip / add with down-sampling / add without, from the
// neighbours of the last search pass.  Returns 0 skip, 1 PointToAdd, 2 PointNoNeedDownsample.
__device__ __forceinline__ uint8_t map_incremental_classify_point(const IncrArgs& a, int i, float4& world_pt) {
    const float4 b = a.body[i];
    const PoseQ s = pose_unpack(a.pose);
    float wx, wy, wz;
    body_to_world(s, b.x, b.y, b.z, wx, wy, wz);
    world_pt = make_float4(wx, wy, wz, 0.0f);
    uint8_t cls = 1;                                            // :1487 (no neighbours / EKF not inited)
    const int cnt = a.cnt[i] & ~CNT_UNPROVEN;                   // (the mark of a walked query is not part of the count)
    if (cnt > 0 && a.flg_EKF_inited) {                          // :1453
        const double fs = a.filter_size_map_min;                // a double in the reference (:56)
        const float mx = (float)(floor((double)wx / fs) * fs + 0.5 * fs);   // :1459-1461
        const float my = (float)(floor((double)wy / fs) * fs + 0.5 * fs);
        const float mz = (float)(floor((double)wz / fs) * fs + 0.5 * fs);
        const float dist = sqdist3(wx, wy, wz, mx, my, mz);     // :1462
        // (all five neighbours are asked for together: one memory round trip instead of the chain of dependent loads the
        // early exit of :1476-1480 would make -- the decision is the same, "any of the five nearer than the point")
        float4 nb[K];
#pragma unroll
        for (int k = 0; k < K; ++k) nb[k] = a.nbr[(size_t)k * a.nq_stride + i];
        const float4 n0 = nb[0];
        if ((double)fabsf(n0.x - mx) > 0.5 * fs && (double)fabsf(n0.y - my) > 0.5 * fs && (double)fabsf(n0.z - mz) > 0.5 * fs) {
            cls = 2;                                            // :1465-1469
        } else {
            bool need_add = true;
            if (cnt >= K) {                                     // :1474-1475
#pragma unroll
                for (int k = 0; k < K; ++k)
                    if (sqdist3(nb[k].x, nb[k].y, nb[k].z, mx, my, mz) < dist) need_add = false;   // :1476-1480
            }
            cls = need_add ? 1 : 0;
        }
    }
    return cls;
}

}  // namespace lio
