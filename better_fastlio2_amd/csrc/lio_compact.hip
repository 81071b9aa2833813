// lio_compact.hip -- single-launch, order-preserving stream compaction / run detection / exclusive scan (up to ~260 k
// items) and stable sort (up to 16 k records) for the batch sizes of map maintenance and scan preparation, gfx950 only.
//
// rocPRIM's device-wide select / run_length_encode / exclusive_scan are two to three launches each (an init kernel for
// the look-back state, the sweep, sometimes a fix-up); a map update strings a dozen of them together on batches of a few
// thousand points, where every launch costs more than its work.  Here every primitive is
// ONE launch: a workgroup handles a tile of 256 items, publishes the tile's count(s) as {count | stamp << 32} -- an
// 8-byte store is single-copy atomic, the stamp is this launch's sequence number, so whoever reads the stamp has the
// count (the same idea as the measurement kernel's partial sums, lio_measure.hip) -- and sums the counts of all tiles
// before it, which it polls.  All tiles of a launch are resident together (<= 1024 workgroups of 256 threads) and a
// tile only ever waits for lower-numbered ones; the waits are bounded and a timeout is reported (bit 0 of *err, counts
// set to zero so that nothing downstream runs wild), never silent.  Larger inputs go to rocPRIM (lio_sort.hip).
#include "lio_kernels.h"
#include "lio_mapadd.h"
#include "lio_incr.h"
#include "lio_bbox.h"
#include "lio_tile.h"

namespace lio {

typedef unsigned long long u64;

__global__ __launch_bounds__(256) void bbox_kernel(const float4* pts, uint32_t n, uint32_t* bbox) {
    for (uint32_t i0 = blockIdx.x * 256u; i0 < n; i0 += gridDim.x * 256u) {     // (block-uniform: every lane reaches the barriers)
        const uint32_t i = i0 + threadIdx.x;
        float4 p = make_float4(0.f, 0.f, 0.f, 0.f);
        if (i < n) p = pts[i];
        block_bbox(p, i < n, bbox);
        __syncthreads();
    }
}
hipError_t launch_bbox(const float4* pts, uint32_t n, uint32_t* bbox, hipStream_t s) {
    if (n == 0) return hipSuccess;
    uint32_t b = (n + 255u) / 256u;
    if (b > 512u) b = 512u;
    hipLaunchKernelGGL(bbox_kernel, dim3(b), dim3(256), 0, s, pts, n, bbox);
    return hipGetLastError();
}

// ---- map_incremental's classification (laserMapping.cpp:1448-1489) AND its two order-preserving lists, one launch --
// A tile publishes EIGHT stamped words: its two list counts and the six faces of the bounding box of its list points (order
// images).  Every tile sums the counts of the tiles before it (its write positions); the LAST tile also folds the boxes and
// leaves the totals where the host / the exchange expect them (counts[0..1], bbox[0..5]) and -- round 6 -- as a BatchDesc with
// both key frames for a map update that follows with no host round trip in between.  (Until round 5 every tile added its box
// with six atomicMin / atomicMax on the same cache line: 391 tiles x 6 same-line atomics at ~7 ns each were most of the
// kernel's 16 us at 100 k points.)
__global__ __launch_bounds__(256) void map_incremental_lists_kernel(IncrArgs a, float4* add_out, float4* nodown_out, uint32_t* counts,
                                                                    uint32_t* bbox, u64* agg, uint32_t stamp, uint32_t* err,
                                                                    BatchDesc* desc, float ds, float inv_cell, int kb_max,
                                                                    uint32_t bound_add, uint32_t bound_plain) {
    __shared__ uint32_t s_w1[4], s_w2[4];
    __shared__ uint32_t s_bb[4][6];
    __shared__ uint32_t s_sum[2][4];
    __shared__ uint32_t s_bad;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = (int)blockIdx.x;
    const int i = b * 256 + tid;
    uint8_t cls = 0;
    float4 wp = make_float4(0.f, 0.f, 0.f, 0.f);
    if (i < a.nq) {
        cls = map_incremental_classify_point(a, i, wp);
        a.world[i] = wp;
        a.cls[i] = cls;
    }
    if (tid == 0) s_bad = 0u;
    // the tile's box (of both lists: the frames of the batch's keys, lio_mapadd.h); a non-finite coordinate counts as 0 (lio_bbox.h)
    {
        const float c3[3] = {wp.x, wp.y, wp.z};
        uint32_t lo[3], hi[3];
#pragma unroll
        for (int ax = 0; ax < 3; ++ax) {
            const float v = (__float_as_uint(c3[ax]) & 0x7f800000u) != 0x7f800000u ? c3[ax] : 0.0f;
            lo[ax] = cls != 0 ? float_order_image(v) : BBOX_MIN_INIT;
            hi[ax] = cls != 0 ? float_order_image(v) : BBOX_MAX_INIT;
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                const uint32_t x = (uint32_t)__shfl_xor((int)lo[ax], o, 64), y = (uint32_t)__shfl_xor((int)hi[ax], o, 64);
                lo[ax] = x < lo[ax] ? x : lo[ax];
                hi[ax] = y > hi[ax] ? y : hi[ax];
            }
        }
        if (lane == 0) {
#pragma unroll
            for (int ax = 0; ax < 3; ++ax) { s_bb[wave][ax] = lo[ax]; s_bb[wave][3 + ax] = hi[ax]; }
        }
    }
    uint32_t tot1, tot2;
    const uint32_t r1 = tile_rank(cls == 1, tot1, s_w1);        // (its barrier also completes s_bb and s_bad)
    const uint32_t r2 = tile_rank(cls == 2, tot2, s_w2);
    // publish: words 0..1 the counts, 2..7 the box
    if (tid < 8) {
        uint32_t v;
        if (tid == 0) v = tot1;
        else if (tid == 1) v = tot2;
        else {
            const int f = tid - 2;
            v = s_bb[0][f];
            for (int w = 1; w < 4; ++w) v = f < 3 ? (s_bb[w][f] < v ? s_bb[w][f] : v) : (s_bb[w][f] > v ? s_bb[w][f] : v);
        }
        __hip_atomic_store(agg + (size_t)b * 8 + tid, (u64)v | ((u64)stamp << 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    // the counts of the tiles before this one
    uint32_t acc1 = 0u, acc2 = 0u;
    bool bad = false;
    for (int p = tid; p < b; p += 256) {
        u64 v1, v2;
        unsigned spins = 0;
        do {
            v1 = __hip_atomic_load(agg + (size_t)p * 8 + 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            v2 = __hip_atomic_load(agg + (size_t)p * 8 + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } while (((uint32_t)(v1 >> 32) != stamp || (uint32_t)(v2 >> 32) != stamp) && ++spins < COMPACT_SPIN_LIMIT);
        bad = bad || (uint32_t)(v1 >> 32) != stamp || (uint32_t)(v2 >> 32) != stamp;
        acc1 += (uint32_t)v1; acc2 += (uint32_t)v2;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { acc1 += (uint32_t)__shfl_xor((int)acc1, o, 64); acc2 += (uint32_t)__shfl_xor((int)acc2, o, 64); }
    if (lane == 0) { s_sum[0][wave] = acc1; s_sum[1][wave] = acc2; }
    if (bad) s_bad = 1u;
    __syncthreads();
    const uint32_t before1 = s_sum[0][0] + s_sum[0][1] + s_sum[0][2] + s_sum[0][3];
    const uint32_t before2 = s_sum[1][0] + s_sum[1][1] + s_sum[1][2] + s_sum[1][3];
    bool ok = s_bad == 0u;
    if (ok && cls == 1) add_out[before1 + r1] = wp;
    if (ok && cls == 2) nodown_out[before2 + r2] = wp;
    // a tile whose wait ran out has skipped its writes: EVERY such tile reports (the last tile's own wait may well succeed)
    if (!ok && tid == 0) atomicOr(err, 1u);
    if (b != (int)gridDim.x - 1) return;
    // ---- the last tile: the box of all tiles (their words were published before the counts this tile has just seen were -- but
    // by other lanes: every word carries its own stamp and is polled for it)
    __syncthreads();                                              // (s_bb / s_sum are reused)
    uint32_t bb[6] = {BBOX_MIN_INIT, BBOX_MIN_INIT, BBOX_MIN_INIT, BBOX_MAX_INIT, BBOX_MAX_INIT, BBOX_MAX_INIT};
    for (int p = tid; p <= b; p += 256) {
        u64 v[6];
        unsigned spins = 0;
        bool got;
        do {
            got = true;
#pragma unroll
            for (int f = 0; f < 6; ++f) {
                v[f] = __hip_atomic_load(agg + (size_t)p * 8 + 2 + f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                got = got && (uint32_t)(v[f] >> 32) == stamp;
            }
        } while (!got && ++spins < COMPACT_SPIN_LIMIT);
        if (!got) { s_bad = 1u; continue; }
#pragma unroll
        for (int f = 0; f < 3; ++f) { bb[f] = (uint32_t)v[f] < bb[f] ? (uint32_t)v[f] : bb[f]; bb[3 + f] = (uint32_t)v[3 + f] > bb[3 + f] ? (uint32_t)v[3 + f] : bb[3 + f]; }
    }
#pragma unroll
    for (int f = 0; f < 3; ++f) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const uint32_t x = (uint32_t)__shfl_xor((int)bb[f], o, 64), y = (uint32_t)__shfl_xor((int)bb[3 + f], o, 64);
            bb[f] = x < bb[f] ? x : bb[f];
            bb[3 + f] = y > bb[3 + f] ? y : bb[3 + f];
        }
    }
    if (lane == 0) {
#pragma unroll
        for (int f = 0; f < 6; ++f) s_bb[wave][f] = bb[f];
    }
    __syncthreads();
    if (tid != 0) return;
    ok = s_bad == 0u;
    if (!ok) atomicOr(err, 1u);
#pragma unroll
    for (int f = 0; f < 6; ++f) {
        uint32_t v = s_bb[0][f];
        for (int w = 1; w < 4; ++w) v = f < 3 ? (s_bb[w][f] < v ? s_bb[w][f] : v) : (s_bb[w][f] > v ? s_bb[w][f] : v);
        bb[f] = v;
    }
    const uint32_t n1 = ok ? before1 + tot1 : 0u, n2 = ok ? before2 + tot2 : 0u;
    counts[0] = n1;
    counts[1] = n2;
    // (the accumulators may hold the box of lists made earlier and never consumed: a wider box is harmless, lio_mapadd.h)
#pragma unroll
    for (int f = 0; f < 3; ++f) { if (bb[f] < bbox[f]) bbox[f] = bb[f]; if (bb[3 + f] > bbox[3 + f]) bbox[3 + f] = bb[3 + f]; }
    if (desc != nullptr) {
        BatchDesc d{};
        d.flags = ok ? 0u : 1u;
        if (ok && n1 + n2 != 0u) {
            float blo[3], bhi[3];
#pragma unroll
            for (int f = 0; f < 3; ++f) { blo[f] = float_from_order_image(bbox[f]); bhi[f] = float_from_order_image(bbox[3 + f]); }
            d.vf = voxel_frame_of(blo, bhi, ds);
            d.cf = cell_frame_of(blo, bhi, ds, inv_cell);
            d.kb_v = (uint32_t)frame_bits(d.vf); d.kb_c = (uint32_t)frame_bits(d.cf) + 1u;
            if ((int)d.kb_v > kb_max || (int)d.kb_c > kb_max) d.flags |= 2u;
            if (n1 > bound_add || n2 > bound_plain) d.flags |= 4u;    // longer than the launches behind this one were sized for
        }
        d.n_add = d.flags == 0u ? n1 : 0u;
        d.n_plain = d.flags == 0u ? n2 : 0u;
        *desc = d;
    }
}
hipError_t launch_map_incremental_lists(const IncrArgs& a, float4* add_out, float4* nodown_out, uint32_t* counts, uint32_t* bbox, u64* agg,
                                        uint32_t stamp, uint32_t* err, hipStream_t s, BatchDesc* desc, float ds, float inv_cell, int kb_max,
                                        uint32_t bound_add, uint32_t bound_plain) {
    if (a.nq <= 0) return hipSuccess;
    if ((size_t)((a.nq + 255) / 256) * 8u > COMPACT_AGG_WORDS) return hipErrorInvalidValue;
    hipLaunchKernelGGL(map_incremental_lists_kernel, dim3((a.nq + 255) / 256), dim3(256), 0, s, a, add_out, nodown_out, counts, bbox, agg, stamp, err,
                       desc, ds, inv_cell, kb_max, bound_add, bound_plain);
    return hipGetLastError();
}

// ---- live points (w == 0) of src[0..n) to the front of dst, order kept; *n_out = their number ----------------------
__global__ __launch_bounds__(256) void compact_live_kernel(const float4* src, uint32_t n, float4* dst, uint32_t* n_out, u64* agg, uint32_t stamp, uint32_t* err) {
    __shared__ uint32_t s_w[4];
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    float4 p = make_float4(0.f, 0.f, 0.f, 1.f);
    if (i < n) p = src[i];
    const bool live = i < n && p.w == 0.0f;
    uint32_t tot;
    const uint32_t r = tile_rank(live, tot, s_w);
    const uint32_t mine[1] = {tot};
    uint32_t before[1];
    const bool ok = tile_exclusive_prefix<1>(mine, agg, stamp, before);
    if (ok && live) dst[before[0] + r] = p;
    if (!ok && threadIdx.x == 0) atomicOr(err, 1u);                 // every tile that skipped its writes reports
    if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) *n_out = ok ? before[0] + tot : 0u;
}
hipError_t launch_compact_live(const float4* src, uint32_t n, float4* dst, uint32_t* n_out, u64* agg, uint32_t stamp, uint32_t* err, hipStream_t s) {
    if (n == 0) return hipMemsetAsync(n_out, 0, sizeof(uint32_t), s);
    hipLaunchKernelGGL(compact_live_kernel, dim3((n + 255u) / 256u), dim3(256), 0, s, src, n, dst, n_out, agg, stamp, err);
    return hipGetLastError();
}

// ---- runs of equal keys in a sorted array: unique[r], starts[r] (= exclusive scan of the run lengths), *n_runs ------
__global__ __launch_bounds__(256) void run_heads_kernel(const uint64_t* keys, uint32_t n, int shift, uint64_t* unique_out, uint32_t* starts_out,
                                                        uint32_t* n_runs_out, u64* agg, uint32_t stamp, uint32_t* err, bool drop_filler) {
    __shared__ uint32_t s_w[4];
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    uint64_t k = 0;
    bool head = false;
    if (i < n) {
        k = keys[i];
        head = (i == 0u || (keys[i - 1] >> shift) != (k >> shift)) && !(drop_filler && k == ~0ull);
    }
    uint32_t tot;
    const uint32_t r = tile_rank(head, tot, s_w);
    const uint32_t mine[1] = {tot};
    uint32_t before[1];
    const bool ok = tile_exclusive_prefix<1>(mine, agg, stamp, before);
    if (ok && head) {
        if (unique_out != nullptr) unique_out[before[0] + r] = k;
        starts_out[before[0] + r] = i;
    }
    if (!ok && threadIdx.x == 0) atomicOr(err, 1u);
    if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) *n_runs_out = ok ? before[0] + tot : 0u;
}
hipError_t launch_run_heads(const uint64_t* keys, uint32_t n, int shift, uint32_t* starts_out, uint32_t* n_runs_out, u64* agg, uint32_t stamp,
                            uint32_t* err, hipStream_t s) {
    if (n == 0) return hipMemsetAsync(n_runs_out, 0, sizeof(uint32_t), s);
    hipLaunchKernelGGL(run_heads_kernel, dim3((n + 255u) / 256u), dim3(256), 0, s, keys, n, shift, (uint64_t*)nullptr, starts_out, n_runs_out, agg, stamp, err, false);
    return hipGetLastError();
}

// ---- the replay list of a down-sampled batch: the batch indices whose dirty_pt mark is set, ascending ------------------
__global__ __launch_bounds__(256) void compact_dirty_kernel(BoundaryArgs a, u64* agg, uint32_t stamp, uint32_t* err) {
    __shared__ uint32_t s_w[4];
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    const uint32_t n = a.view.n_dev != nullptr ? min(a.view.n, *a.view.n_dev + a.view.n_add) : a.view.n;   // (beyond it the marks are not initialised)
    const bool live = i < n && a.dirty_pt[i] != 0;
    uint32_t tot;
    const uint32_t r = tile_rank(live, tot, s_w);
    const uint32_t mine[1] = {tot};
    uint32_t before[1];
    const bool ok = tile_exclusive_prefix<1>(mine, agg, stamp, before);
    if (ok && live) a.dlist[before[0] + r] = i;
    if (!ok && threadIdx.x == 0) atomicOr(err, 1u);
    if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) *a.n_dirty = ok ? before[0] + tot : 0u;
}
hipError_t launch_compact_dirty(const BoundaryArgs& a, u64* agg, uint32_t stamp, uint32_t* err, hipStream_t s) {
    if (a.view.n == 0) return hipSuccess;
    hipLaunchKernelGGL(compact_dirty_kernel, dim3((a.view.n + 255u) / 256u), dim3(256), 0, s, a, agg, stamp, err);
    return hipGetLastError();
}

// ---- the pending list of an index update, its (cell, list) keys and their partition histogram, one launch -------------
// tiles 0 .. Tp-1 copy the plain inserts (PointNoNeedDownsample) to the front, the tiles behind them compact the
// survivors (w == 0) of the fold and of the replay behind those, in order
__global__ __launch_bounds__(256) void pend_build_kernel(PendBuildArgs a) {
    __shared__ uint32_t s_w[4];
    __shared__ uint32_t s_hist[256];
    const int tid = threadIdx.x;
    const uint32_t tp = (a.n_plain + 255u) / 256u;              // tiles of the plain list (by the launch bound with desc)
    // (round 6) the counts and the cell frame may still be on the device: n_plain / n_surv_slots are then launch bounds
    uint32_t n_plain = a.n_plain, half = a.n_surv_slots, n_half = a.n_surv_slots;
    if (a.desc != nullptr) { n_plain = min(a.n_plain, a.desc->n_plain); half = a.n_surv_slots / 2u; n_half = min(half, a.desc->n_add); }
    const KeyFrame frame = a.desc != nullptr ? a.desc->cf : a.frame;     // (a local: never assign into the by-value argument struct)
    const int kb = frame_bits(frame) + 1;
    if (a.count_parts != nullptr && blockIdx.x == gridDim.x - 1) {
        // the extra workgroup: the fold's counters, one pair per workgroup of fold_commit_kernel, added up (fixed order)
        uint32_t r = 0, c = 0;
        for (uint32_t i = tid; i < a.n_count_parts; i += 256u) { const uint2 v = a.count_parts[i]; r += v.x; c += v.y; }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { r += (uint32_t)__shfl_xor((int)r, o, 64); c += (uint32_t)__shfl_xor((int)c, o, 64); }
        __shared__ uint32_t s_rc[4][2];
        if ((tid & 63) == 0) { s_rc[tid >> 6][0] = r; s_rc[tid >> 6][1] = c; }
        __syncthreads();
        if (tid < 2) a.fold_counts[tid] = s_rc[0][tid] + s_rc[1][tid] + s_rc[2][tid] + s_rc[3][tid];
        if (a.desc != nullptr && tid == 2) { a.desc_copy[0] = a.desc->n_add; a.desc_copy[1] = a.desc->n_plain; a.desc_copy[2] = a.desc->flags; }
        return;
    }
    const uint32_t n_tiles = gridDim.x - (a.count_parts != nullptr ? 1u : 0u);     // tiles of the list proper
    s_hist[tid] = 0u;
    __syncthreads();
    auto emit = [&](uint32_t pos, float4 p, uint32_t list) {
        // (a non-finite coordinate -- a plain insert may carry one, the reference stores whatever it is given -- is filed under
        // the cell of 0: the bounding box counted it as 0 too)
        auto fin = [](float v) { return (__float_as_uint(v) & 0x7f800000u) != 0x7f800000u ? v : 0.0f; };
        const int cx = cell_coord(fin(p.x), a.grid.inv_cell), cy = cell_coord(fin(p.y), a.grid.inv_cell), cz = cell_coord(fin(p.z), a.grid.inv_cell);
        if (!frame_holds(frame, cx, cy, cz)) atomicOr(a.flags, 2u);            // the host's frame is wrong: reported
        const uint64_t key = (frame_key(frame, cx, cy, cz) << 1) | (uint64_t)list;
        a.pend[pos] = make_float4(p.x, p.y, p.z, 0.0f);
        a.keys[pos] = key;
        a.order[pos] = pos;
        if (a.hist != nullptr) atomicAdd(&s_hist[gsort_digit(key, kb)], 1u);
    };
    if (blockIdx.x < tp) {
        const uint32_t i = blockIdx.x * 256u + (uint32_t)tid;
        if (i < n_plain) emit(i, a.plain[i], 1u);
        if (a.n_surv_slots == 0u && blockIdx.x == 0 && tid == 0) { *a.n_surv = 0u; if (a.n_pend != nullptr) *a.n_pend = n_plain; }
    } else {
        const uint32_t b = blockIdx.x - tp;
        const uint32_t i = b * 256u + (uint32_t)tid;
        float4 p = make_float4(0.f, 0.f, 0.f, 1.f);
        // (with desc: two halves of `half` slots, the first n_half of each are real -- the others were never initialised)
        const bool real = i < a.n_surv_slots && (i < half ? i : i - half) < n_half;
        if (real) p = a.surv[i];
        const bool live = real && p.w == 0.0f;
        uint32_t tot;
        const uint32_t r = tile_rank(live, tot, s_w);
        const uint32_t mine[1] = {tot};
        uint32_t before[1];
        const bool ok = tile_exclusive_prefix<1>(mine, a.agg, a.stamp, before, (int)b);
        if (ok && live) emit(n_plain + before[0] + r, p, 0u);
        if (!ok && tid == 0) atomicOr(a.err, 1u);
        if (blockIdx.x == n_tiles - 1 && tid == 0) {
            *a.n_surv = ok ? before[0] + tot : 0u;
            if (a.n_pend != nullptr) *a.n_pend = ok ? n_plain + before[0] + tot : 0u;
        }
    }
    if (a.hist != nullptr) {
        __syncthreads();
        const uint32_t v = s_hist[tid];
        if (v != 0u) atomicAdd(a.hist + tid, v);
    }
}
hipError_t launch_pend_build(const PendBuildArgs& a, hipStream_t s) {
    const uint32_t tiles = (a.n_plain + 255u) / 256u + (a.n_surv_slots + 255u) / 256u;
    if (tiles == 0u) return hipMemsetAsync(a.n_surv, 0, sizeof(uint32_t), s);
    hipLaunchKernelGGL(pend_build_kernel, dim3(tiles + (a.count_parts != nullptr ? 1u : 0u)), dim3(256), 0, s, a);
    return hipGetLastError();
}

// ---- the rows of a measurement pass for the unmodified callback path: selected points only, scan order, column-major ----
__global__ __launch_bounds__(256) void rows_compact_kernel(RowsCompactArgs a) {
    __shared__ uint32_t s_w[4];
    const int i = blockIdx.x * 256 + threadIdx.x;
    const bool live = i < a.nq && a.sel[i] != 0;
    uint32_t tot;
    const uint32_t r = tile_rank(live, tot, s_w);
    const uint32_t mine[1] = {tot};
    uint32_t before[1];
    const bool ok = tile_exclusive_prefix<1>(mine, a.agg, a.stamp, before);
    if (ok && live) {
        const size_t m = (size_t)*a.m_dev, row = (size_t)before[0] + r;
        if (row < m) {                                                     // (m == 0: the pass failed; nothing is written)
            const double* src = a.rows + (size_t)i * 13;
#pragma unroll
            for (int c = 0; c < 13; ++c) a.dst[(size_t)c * m + row] = src[c];
        }
    }
    if (!ok && threadIdx.x == 0) atomicOr(a.err, 1u);
}
hipError_t launch_rows_compact(const RowsCompactArgs& a, hipStream_t s) {
    if (a.nq <= 0) return hipSuccess;
    hipLaunchKernelGGL(rows_compact_kernel, dim3((a.nq + 255) / 256), dim3(256), 0, s, a);
    return hipGetLastError();
}

// ---- live points (w == 0) of a LARGE array to the front of dst, order kept: the compaction of a full index rebuild --------
// Millions of slots: count per tile of 2048, scan the tile counts (launch_exclusive_scan), scatter -- three launches that
// stream the array twice (rocPRIM's select took 340 us for 5.6 M slots; this is bound by the two reads and one write).
constexpr uint32_t LIVE_TILE = 2048;
__global__ __launch_bounds__(256) void live_tile_count_kernel(const float4* __restrict__ src, uint32_t n, uint32_t* __restrict__ tile_cnt, uint32_t n_tiles) {
    __shared__ uint32_t s_w[4];
    const uint32_t t = blockIdx.x;
    if (t >= n_tiles) { if (threadIdx.x == 0) tile_cnt[t] = 0u; return; }        // the entry behind the last tile: the scan's total
    // (only the tile's total is wanted: neighbouring lanes read neighbouring slots, the eight loads of a lane are independent)
    const uint32_t i0 = t * LIVE_TILE + threadIdx.x;
    float w[8];
#pragma unroll
    for (uint32_t u = 0; u < 8u; ++u) w[u] = i0 + u * 256u < n ? src[i0 + u * 256u].w : 1.0f;
    uint32_t c = 0;
#pragma unroll
    for (uint32_t u = 0; u < 8u; ++u) c += w[u] == 0.0f ? 1u : 0u;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) c += (uint32_t)__shfl_xor((int)c, o, 64);
    if ((threadIdx.x & 63) == 0) s_w[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) tile_cnt[t] = s_w[0] + s_w[1] + s_w[2] + s_w[3];
}
__global__ __launch_bounds__(256) void live_tile_scatter_kernel(const float4* __restrict__ src, uint32_t n, const uint32_t* __restrict__ tile_off,
                                                                float4* __restrict__ dst, uint32_t* __restrict__ n_out, uint32_t n_tiles) {
    __shared__ uint32_t s_w[4];
    const uint32_t t = blockIdx.x, tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const uint32_t i0 = t * LIVE_TILE + tid * 8u;
    float4 p[8];
    uint32_t c = 0;
#pragma unroll
    for (uint32_t u = 0; u < 8u; ++u) {
        p[u] = make_float4(0.f, 0.f, 0.f, 1.f);
        if (i0 + u < n) p[u] = src[i0 + u];
        c += p[u].w == 0.0f ? 1u : 0u;
    }
    uint32_t inc = c;                                            // inclusive scan of the lanes' counts
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t v = (uint32_t)__shfl_up((int)inc, o, 64);
        if (lane >= (uint32_t)o) inc += v;
    }
    if (lane == 63u) s_w[wave] = inc;
    __syncthreads();
    uint32_t pos = tile_off[t] + inc - c;
    for (uint32_t w = 0; w < wave; ++w) pos += s_w[w];
#pragma unroll
    for (uint32_t u = 0; u < 8u; ++u)
        if (p[u].w == 0.0f) dst[pos++] = p[u];
    if (t == 0 && tid == 0) *n_out = tile_off[n_tiles];
}
hipError_t launch_compact_live_large(const float4* src, uint32_t n, float4* dst, uint32_t* n_out, uint32_t* tile_cnt, uint32_t* tile_off,
                                     u64* agg, uint32_t stamp, uint32_t* err, hipStream_t s) {
    if (n == 0) return hipMemsetAsync(n_out, 0, sizeof(uint32_t), s);
    const uint32_t n_tiles = (n + LIVE_TILE - 1u) / LIVE_TILE;
    if (n_tiles + 1u > COMPACT_MAX_ITEMS) return hipErrorInvalidValue;
    hipLaunchKernelGGL(live_tile_count_kernel, dim3(n_tiles + 1u), dim3(256), 0, s, src, n, tile_cnt, n_tiles);
    hipError_t e = launch_exclusive_scan(tile_cnt, n_tiles + 1u, tile_off, agg, stamp, err, s);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(live_tile_scatter_kernel, dim3(n_tiles), dim3(256), 0, s, src, n, tile_off, dst, n_out, n_tiles);
    return hipGetLastError();
}

// ---- exclusive scan of u32[n] ----------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void exclusive_scan_kernel(const uint32_t* in, uint32_t n, uint32_t* out, u64* agg, uint32_t stamp, uint32_t* err) {
    __shared__ uint32_t s_w[4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t i = blockIdx.x * 256u + tid;
    const uint32_t v = i < n ? in[i] : 0u;
    uint32_t inc = v;                                        // inclusive scan inside the wavefront
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t u = (uint32_t)__shfl_up((int)inc, o, 64);
        if (lane >= o) inc += u;
    }
    if (lane == 63) s_w[wave] = inc;
    __syncthreads();
    uint32_t off = 0;
    for (int w = 0; w < wave; ++w) off += s_w[w];
    const uint32_t tot = s_w[0] + s_w[1] + s_w[2] + s_w[3];
    const uint32_t mine[1] = {tot};
    uint32_t before[1];
    const bool ok = tile_exclusive_prefix<1>(mine, agg, stamp, before);
    if (i < n) out[i] = ok ? before[0] + off + inc - v : 0u;
    if (!ok && tid == 0) atomicOr(err, 1u);
}
hipError_t launch_exclusive_scan(const uint32_t* in, uint32_t n, uint32_t* out, u64* agg, uint32_t stamp, uint32_t* err, hipStream_t s) {
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(exclusive_scan_kernel, dim3((n + 255u) / 256u), dim3(256), 0, s, in, n, out, agg, stamp, err);
    return hipGetLastError();
}

// ---- stable sort of (u64 key, u32 value) pairs, n <= SORT_MAX_ITEMS, ONE launch ------------------------------------
// rocPRIM sorts a batch of this size with a block sort and log2(n / block) merge launches (nine launches, ~40 us, for
// the 8 k keys of an insert batch).  At these sizes the whole chip can afford the quadratic answer in one launch with
// no workspace, no hand-off between workgroups and nothing to get wrong: the rank of a record is the number of records
// that sort before it -- smaller key, or equal key and earlier position (= the stable order).  A workgroup takes 64
// records, one per lane; its sixteen wavefronts each count over a sixteenth of the batch, the key of the record under
// comparison is wave-uniform (scalar loads, one 64-bit compare + one add per pair), the counts meet in LDS and the
// record is scattered to its rank.  8.6 k records: 74 M pairs, ~10 us; the cost is n^2, so the library takes over above
// SORT_MAX_ITEMS.  (Tried first and dropped: bitonic tiles in LDS + a multi-way rank search behind a stamped hand-off;
// a 1024-record tile alone took ~30 us -- two wavefronts issuing ~220 instructions per round of the network.)
constexpr uint32_t RANK_WAVES = 16;
// the count of one wavefront's share [j0, j1) for the record (ki, i) of workgroup rows [i0, i0 + 64); T = uint32_t when every key
// of the batch fits 32 bits (compressed keys, lio_mapadd.h: the usual case) -- a 32-bit compare issues at full rate, a 64-bit one
// does not -- the loads stay 8 bytes apart
template <class T>
__device__ __forceinline__ uint32_t rank_count(const uint64_t* __restrict__ keys_in, T ki, uint32_t i, uint32_t i0, uint32_t j0, uint32_t j1) {
    auto key = [&](uint32_t j) -> T { return sizeof(T) == 4 ? (T)reinterpret_cast<const uint32_t*>(keys_in)[2u * j] : (T)keys_in[j]; };
    uint32_t cnt = 0;
    // records before the workgroup's own 64: "not after" = key <=; after them: key <; the 64 themselves: by position
    {
        const uint32_t e = min(j1, i0);
#pragma unroll 16
        for (uint32_t j = j0; j < e; ++j) cnt += key(j) <= ki ? 1u : 0u;
    }
    {
        const uint32_t sj = max(j0, i0), e = min(j1, i0 + 64u);
        for (uint32_t j = sj; j < e; ++j) {
            const T kj = key(j);
            cnt += (kj < ki || (kj == ki && j < i)) ? 1u : 0u;
        }
    }
    {
        const uint32_t sj = max(j0, i0 + 64u);
#pragma unroll 16
        for (uint32_t j = sj; j < j1; ++j) cnt += key(j) < ki ? 1u : 0u;
    }
    return cnt;
}
__global__ __launch_bounds__(1024) void rank_sort_pairs_kernel(const uint64_t* __restrict__ keys_in, const uint32_t* __restrict__ vals_in, uint32_t n,
                                                               uint64_t* __restrict__ keys_out, uint32_t* __restrict__ vals_out,
                                                               const uint32_t* __restrict__ n_dev, uint32_t n_dev_add, int kb, const uint32_t* __restrict__ kb_dev) {
    __shared__ uint32_t s_cnt[RANK_WAVES][64];
    if (n_dev != nullptr) n = min(n, *n_dev + n_dev_add);          // the count is still on the device: n was the launch bound
    if (blockIdx.x * 64u >= n) return;
    if (kb_dev != nullptr) kb = (int)*kb_dev;
    const uint32_t lane = threadIdx.x & 63u, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t i0 = blockIdx.x * 64u, i = i0 + lane;
    const bool in = i < n;
    const uint64_t ki = in ? keys_in[i] : ~0ull;
    const uint32_t per = (n + RANK_WAVES - 1u) / RANK_WAVES;
    const uint32_t j0 = min(n, wave * per), j1 = min(n, j0 + per);
    // (a lane beyond n holds the all-ones key in either width: it counts everything and stores nothing)
    const uint32_t cnt = kb <= 32 ? rank_count<uint32_t>(keys_in, in ? (uint32_t)ki : 0xffffffffu, i, i0, j0, j1) : rank_count<uint64_t>(keys_in, ki, i, i0, j0, j1);
    s_cnt[wave][lane] = cnt;
    __syncthreads();
    if (wave == 0u && in) {
        uint32_t r = 0;
#pragma unroll
        for (int w = 0; w < (int)RANK_WAVES; ++w) r += s_cnt[w][lane];
        keys_out[r] = ki;
        vals_out[r] = vals_in[i];
    }
}
hipError_t launch_sort_pairs_small(const uint64_t* keys_in, const uint32_t* vals_in, uint32_t n, uint64_t* keys_out, uint32_t* vals_out,
                                   hipStream_t s, const uint32_t* n_dev, uint32_t n_dev_add, int kb, const uint32_t* kb_dev) {
    if (n == 0) return hipSuccess;
    if (n > SORT_MAX_ITEMS) return hipErrorInvalidValue;
    hipLaunchKernelGGL(rank_sort_pairs_kernel, dim3((n + 63u) / 64u), dim3(64 * RANK_WAVES), 0, s, keys_in, vals_in, n, keys_out, vals_out, n_dev, n_dev_add, kb, kb_dev);
    return hipGetLastError();
}

}  // namespace lio
