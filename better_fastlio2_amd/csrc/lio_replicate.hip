// lio_replicate.hip -- device side of the replica exchange, gfx950 only.
//
// One rank = one GPU = one scan stream = one full replica of the map.  After a scan step every rank holds the points
// it wants inserted (PointToAdd / PointNoNeedDownsample, src/laserMapping.cpp:1442-1493) and at most a few delete boxes
// (cub_needrm, :1183-1198).  The reference has one map and one stream; here the lists of all ranks travel in ONE
// ncclAllGather (RCCL over xGMI) of fixed-size messages on the library's own stream, and every replica applies them in
// rank order -- so that the replicas stay bit-identical.  The counts travel in the message header and are consumed on
// the device: pack_kernel builds this rank's message from the device-resident lists and counters, compact_kernel turns
// the gathered messages into the two concatenated batches (rank order) and the box list.
//
//   message = [ RepHeader 16 B | boxes 4 x 6 floats (96 B) | pad 16 B ] [ add: cap float4 ] [ nodown: cap float4 ]
#include "lio_kernels.h"
#include "lio_bbox.h"

namespace lio {

__global__ __launch_bounds__(256) void replicate_pack_kernel(RepPackArgs a) {
    const uint32_t n_add = a.counts[0], n_nodown = a.counts[1];
    const uint32_t cap = a.cap;
    const uint32_t m_add = n_add < cap ? n_add : cap, m_nd = n_nodown < cap ? n_nodown : cap;
    float4* msg = a.msg;
    if (blockIdx.x == 0 && threadIdx.x < REP_HEADER_F4) {
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (threadIdx.x == 0) {
            // counts beyond the capacity are flagged: the receiver reports them and the caller repeats the exchange with a
            // larger capacity (nothing is dropped silently)
            const uint32_t flags = (n_add > cap || n_nodown > cap) ? REP_FLAG_OVERFLOW : 0u;
            v = make_float4(__uint_as_float(m_add), __uint_as_float(m_nd), __uint_as_float((uint32_t)a.n_boxes), __uint_as_float(flags));
        } else if (threadIdx.x <= 6) {
            const int f0 = ((int)threadIdx.x - 1) * 4;                    // 24 floats of boxes in float4s 1..6
            v = make_float4(a.boxes[f0], a.boxes[f0 + 1], a.boxes[f0 + 2], a.boxes[f0 + 3]);
        }
        msg[threadIdx.x] = v;
    }
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < cap; i += gridDim.x * blockDim.x) {
        if (i < m_add) msg[REP_HEADER_F4 + i] = a.add[i];
        if (i < m_nd) msg[REP_HEADER_F4 + cap + i] = a.nodown[i];
    }
}

// gathered [world] messages -> add_out / nodown_out (concatenated in rank order), boxes_out, totals
// totals: [0] add points, [1] nodown points, [2] boxes, [3] or of the flags
__global__ __launch_bounds__(256) void replicate_compact_kernel(RepCompactArgs a) {
    const uint32_t cap = a.cap;
    const size_t stride = (size_t)REP_HEADER_F4 + 2 * (size_t)cap;       // float4s per message
    const int r = blockIdx.y;
    const float4 hdr = a.gathered[(size_t)r * stride];
    const uint32_t n_add = __float_as_uint(hdr.x), n_nd = __float_as_uint(hdr.y);
    uint32_t off_add = 0, off_nd = 0, off_box = 0, flags = 0;
    for (int p = 0; p < a.world; ++p) {                                   // world <= a few dozen: every block sums the headers
        const float4 h = a.gathered[(size_t)p * stride];
        if (p < r) { off_add += __float_as_uint(h.x); off_nd += __float_as_uint(h.y); off_box += __float_as_uint(h.z); }
        flags |= __float_as_uint(h.w);
    }
    const float4* src = a.gathered + (size_t)r * stride + REP_HEADER_F4;
    const uint32_t n_max = n_add > n_nd ? n_add : n_nd;
    for (uint32_t i0 = blockIdx.x * blockDim.x; i0 < n_max; i0 += gridDim.x * blockDim.x) {      // (block-uniform: the box reduction has barriers)
        const uint32_t i = i0 + threadIdx.x;
        float4 pa = make_float4(0.f, 0.f, 0.f, 0.f), pn = pa;
        if (i < n_add) { pa = src[i]; a.add_out[off_add + i] = pa; }
        if (i < n_nd) { pn = src[cap + i]; a.nodown_out[off_nd + i] = pn; }
        // the bounding box of everything that was gathered: the frames of the batch's keys (lio_mapadd.h); it reaches the host
        // with the totals
        block_bbox(pa, i < n_add, a.bbox);
        __syncthreads();
        block_bbox(pn, i < n_nd, a.bbox);
        __syncthreads();
    }
    if (blockIdx.x == 0) {
        const uint32_t nb = __float_as_uint(hdr.z);
        const float* bsrc = reinterpret_cast<const float*>(a.gathered + (size_t)r * stride + 1);
        for (uint32_t i = threadIdx.x; i < nb * 6 && off_box * 6 + i < (uint32_t)REP_MAX_BOXES_TOTAL * 6; i += blockDim.x)
            a.boxes_out[off_box * 6 + i] = bsrc[i];
        if (r == a.world - 1 && threadIdx.x == 0) {
            a.totals[0] = off_add + n_add; a.totals[1] = off_nd + n_nd; a.totals[2] = off_box + nb; a.totals[3] = flags;
        }
    }
}

// loop-back test mode (lio_debug_set "replicate_loopback"): the lists and counts of one VIRTUAL rank are parked so that
// several of them can later be packed into the message slots of one gathered buffer (what ncclAllGather delivers)
__global__ __launch_bounds__(256) void replicate_stage_kernel(const uint32_t* counts, const float4* add, const float4* nodown, uint32_t cap,
                                                              float4* stage_add, float4* stage_nodown, uint32_t* stage_counts) {
    const uint32_t n_add = min(counts[0], cap), n_nd = min(counts[1], cap);
    if (blockIdx.x == 0 && threadIdx.x < 2) stage_counts[threadIdx.x] = counts[threadIdx.x];
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < cap; i += gridDim.x * blockDim.x) {
        if (i < n_add) stage_add[i] = add[i];
        if (i < n_nd) stage_nodown[i] = nodown[i];
    }
}
hipError_t launch_replicate_stage(const uint32_t* counts, const float4* add, const float4* nodown, uint32_t cap, float4* stage_add,
                                  float4* stage_nodown, uint32_t* stage_counts, hipStream_t s) {
    uint32_t blocks = (cap + 255u) / 256u;
    if (blocks > 1024u) blocks = 1024u;
    if (blocks == 0u) blocks = 1u;
    hipLaunchKernelGGL(replicate_stage_kernel, dim3(blocks), dim3(256), 0, s, counts, add, nodown, cap, stage_add, stage_nodown, stage_counts);
    return hipGetLastError();
}

hipError_t launch_replicate_pack(const RepPackArgs& a, hipStream_t s) {
    uint32_t blocks = (a.cap + 255u) / 256u;
    if (blocks > 1024u) blocks = 1024u;
    if (blocks == 0u) blocks = 1u;
    hipLaunchKernelGGL(replicate_pack_kernel, dim3(blocks), dim3(256), 0, s, a);
    return hipGetLastError();
}
hipError_t launch_replicate_compact(const RepCompactArgs& a, hipStream_t s) {
    uint32_t blocks = (a.cap + 255u) / 256u;
    if (blocks > 256u) blocks = 256u;
    if (blocks == 0u) blocks = 1u;
    hipLaunchKernelGGL(replicate_compact_kernel, dim3(blocks, a.world), dim3(256), 0, s, a);
    return hipGetLastError();
}

}  // namespace lio
