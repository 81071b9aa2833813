// lio_bbox.h -- bounding box of a batch of points, accumulated by the kernels that produce insert lists (device code).
#pragma once
#include <hip/hip_runtime.h>
#include "lio_mapadd.h"

namespace lio {

// Bounding box of the points a workgroup holds (one per lane, `use` = the lane contributes): a wavefront reduction, then
// six atomics per workgroup on order-preserving unsigned images (lio_mapadd.h).  A non-finite coordinate counts as 0: such a
// point never survives Add_Points(.., true), and as a plain insert it is filed under the cell of 0 (pend_build_kernel).
__device__ __forceinline__ void block_bbox(float4 p, bool use, uint32_t* bbox) {
    __shared__ uint32_t s_bb[4][6];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    uint32_t lo[3], hi[3];
    const float c[3] = {p.x, p.y, p.z};
#pragma unroll
    for (int ax = 0; ax < 3; ++ax) {
        const float v = (__float_as_uint(c[ax]) & 0x7f800000u) != 0x7f800000u ? c[ax] : 0.0f;
        lo[ax] = use ? float_order_image(v) : BBOX_MIN_INIT;
        hi[ax] = use ? float_order_image(v) : BBOX_MAX_INIT;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const uint32_t a = (uint32_t)__shfl_xor((int)lo[ax], o, 64), b = (uint32_t)__shfl_xor((int)hi[ax], o, 64);
            lo[ax] = a < lo[ax] ? a : lo[ax];
            hi[ax] = b > hi[ax] ? b : hi[ax];
        }
    }
    if (lane == 0) {
#pragma unroll
        for (int ax = 0; ax < 3; ++ax) { s_bb[wave][ax] = lo[ax]; s_bb[wave][3 + ax] = hi[ax]; }
    }
    __syncthreads();
    const int nw = (int)(blockDim.x >> 6);
    if (tid < 6) {
        uint32_t v = s_bb[0][tid];
        for (int w = 1; w < nw; ++w) v = tid < 3 ? (s_bb[w][tid] < v ? s_bb[w][tid] : v) : (s_bb[w][tid] > v ? s_bb[w][tid] : v);
        if (tid < 3) { if (v != BBOX_MIN_INIT) atomicMin(bbox + tid, v); }
        else if (v != BBOX_MAX_INIT) atomicMax(bbox + tid, v);
    }
}
}  // namespace lio
