// lio_gsort.hip -- stable sort of (u64 key, u32 value) records for the grouping steps of map maintenance (insert batch by
// down-sample voxel, pending points by grid cell), gfx950 only.
//
// What gets sorted is small and narrow: a few thousand records per scan on one GPU, ~10^5 when eight replicas' lists
// arrive together; the keys are COMPRESSED cell coordinates (lio_mapadd.h: KeyFrame -- coordinates relative to the
// batch's own bounding box, which the host knows from the read-back that also brings the list lengths), 20-30 bits
// where the packed 3 x 21-bit key had 63.  rocPRIM sorts such a batch with seven to ten launches (55-75 us at 10^5
// records, measured); here:
//   n <= GSORT_RANK_MAX   rank by counting, ONE launch (lio_compact.hip: rank_sort_pairs_kernel, round 2)
//   n <= GSORT_MAX        TWO launches:
//     gsort_partition_kernel  stable split by the top 8 key bits into 256 buckets.  A workgroup takes a tile of 2048
//                             records; inside a wavefront equal digits find each other with eight ballots (rank among the
//                             peers = position in the run), wave counters in LDS order the wavefronts, the tile's 256
//                             counts are published as stamped 8-byte words and every tile sums the words of ALL tiles
//                             before it (no chained look-back: nobody waits for a prefix, only for counts that are
//                             published before any waiting starts).  Bucket bases come from a 256-bin histogram the
//                             KEY-GENERATING kernel of the caller accumulated (gsort_hist_* below), so there is no
//                             histogram launch.  The record travels as key << 18 | input position: one 8-byte word.
//     gsort_bucket_kernel     one workgroup per bucket: the bucket (<= 8192 records: 64 KB) is pulled into LDS and
//                             LSD-sorted there over the remaining key bits (<= 8 bits per pass, same ballot ranking,
//                             chunks of 1024 in order), written out as sorted key + value.  Equal keys share their top
//                             bits, so a group never spans buckets.  A bucket that does not fit LDS (every point on one
//                             slab of the top digit) runs the same passes through global memory -- slower, never wrong.
// Stability = input order inside equal keys = batch order, which is what makes the voxel fold equal to the reference's
// one-insert-after-the-other loop.  Waits are bounded and reported (*err |= 1), as in lio_compact.hip.
#include "lio_kernels.h"
#include "lio_mapadd.h"

namespace lio {

typedef unsigned long long u64;
constexpr unsigned GS_SPIN_LIMIT = 1u << 22;
constexpr int GS_IB = GSORT_INDEX_BITS;
constexpr u64 GS_IMASK = (1ull << GS_IB) - 1ull;
constexpr uint32_t GS_TILE = 2048;               // records per partition tile: 8 wavefronts x 4 rounds x 64 lanes
constexpr uint32_t GS_LDS_CAP = 8192;            // records of a bucket that is sorted inside LDS
constexpr uint32_t GS_RANK_CAP = 512;            // ... that is ranked by counting there (no passes)

// lanes of this wavefront (among the valid ones) that hold the same digit
__device__ __forceinline__ u64 gs_match(uint32_t d, bool valid, int bits) {
    u64 peers = __ballot(valid);
    for (int b = 0; b < bits; ++b) {
        const bool bit = ((d >> b) & 1u) != 0u;
        const u64 bb = __ballot(bit);
        peers &= bit ? bb : ~bb;
    }
    return peers;
}
__device__ __forceinline__ void gs_wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
// exclusive scan of 256 values held by threads 0..255 (one each); every thread of the workgroup must call it
__device__ __forceinline__ uint32_t gs_scan256(uint32_t v, uint32_t (&s_wtot)[4]) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    uint32_t inc = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t u = (uint32_t)__shfl_up((int)inc, o, 64);
        if (lane >= o) inc += u;
    }
    if (tid < 256 && lane == 63) s_wtot[wave] = inc;
    __syncthreads();
    uint32_t wb = 0;
    for (int w = 0; w < wave && w < 4; ++w) wb += s_wtot[w];
    return wb + inc - v;
}

__global__ __launch_bounds__(512) void gsort_partition_kernel(GsortArgs a) {
    __shared__ uint32_t s_wcnt[8][256];        // per wavefront: running count of each digit, then the wavefront's base
    __shared__ uint32_t s_base[256];           // first output slot of this tile's records of each digit
    __shared__ uint32_t s_half[2][256];
    __shared__ uint32_t s_wtot[4];
    __shared__ uint32_t s_bad;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t n = a.n_dev != nullptr ? min(a.n, *a.n_dev + a.n_dev_add) : a.n;
    if (blockIdx.x == 0 && tid < 256) a.hist_other[tid] = 0u;             // the histogram of the NEXT sort starts from zero
    for (int x = tid; x < 8 * 256; x += 512) (&s_wcnt[0][0])[x] = 0u;
    if (tid == 0) s_bad = 0u;
    __syncthreads();
    const int kb = a.kb_dev != nullptr ? (int)*a.kb_dev : a.kb;
    const int msd_shift = kb > 8 ? kb - 8 : 0;
    const uint32_t i0 = blockIdx.x * GS_TILE + (uint32_t)wave * 256u;
    u64 rec[4];
    uint32_t off[4], dg[4];
    bool vd[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const uint32_t i = i0 + (uint32_t)k * 64u + (uint32_t)lane;
        vd[k] = i < n;
        const u64 key = vd[k] ? a.keys[i] : 0ull;
        rec[k] = (key << GS_IB) | (u64)i;
        dg[k] = (uint32_t)(key >> msd_shift) & 255u;
        const u64 peers = gs_match(dg[k], vd[k], 8);
        const uint32_t rank = (uint32_t)__popcll(peers & ((1ull << lane) - 1ull));
        const int leader = vd[k] ? __ffsll((long long)peers) - 1 : lane;
        uint32_t old = 0u;
        if (vd[k] && lane == leader) {
            old = s_wcnt[wave][dg[k]];
            s_wcnt[wave][dg[k]] = old + (uint32_t)__popcll(peers);
        }
        gs_wave_sync();
        old = (uint32_t)__shfl((int)old, leader, 64);
        off[k] = old + rank;
    }
    __syncthreads();
    // per digit: bases of the wavefronts inside the tile, the tile's count (published), the counts of the tiles before
    if (tid < 256) {
        uint32_t run = 0u;
#pragma unroll
        for (int w = 0; w < 8; ++w) { const uint32_t t = s_wcnt[w][tid]; s_wcnt[w][tid] = run; run += t; }
        __hip_atomic_store(a.agg + (size_t)blockIdx.x * 256u + (uint32_t)tid, (u64)run | ((u64)a.stamp << 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    {
        const uint32_t d = (uint32_t)tid & 255u, h = (uint32_t)tid >> 8, b = blockIdx.x;
        uint32_t acc = 0u;
        bool bad = false;
        for (uint32_t p0 = h; p0 < b; p0 += 16u) {              // eight words in flight per lane, tiles h, h + 2, ...
            u64 v[8];
            unsigned spins = 0;
            bool ok;
            do {
                ok = true;
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const uint32_t p = p0 + 2u * (uint32_t)u;
                    v[u] = (u64)a.stamp << 32;
                    if (p < b) v[u] = __hip_atomic_load(a.agg + (size_t)p * 256u + d, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    ok = ok && (uint32_t)(v[u] >> 32) == a.stamp;
                }
            } while (!ok && ++spins < GS_SPIN_LIMIT);
            bad = bad || !ok;
#pragma unroll
            for (int u = 0; u < 8; ++u) acc += (uint32_t)v[u];
        }
        s_half[h][d] = acc;
        if (bad) s_bad = 1u;
    }
    const uint32_t hv = tid < 256 ? a.hist[tid] : 0u;
    const uint32_t gbase = gs_scan256(hv, s_wtot);               // (syncs: s_half is complete behind it)
    if (tid < 256) s_base[tid] = gbase + s_half[0][tid] + s_half[1][tid];
    __syncthreads();
    const bool ok = s_bad == 0u;
#pragma unroll
    for (int k = 0; k < 4; ++k)
        if (vd[k] && ok) a.part[s_base[dg[k]] + s_wcnt[wave][dg[k]] + off[k]] = rec[k];
    if (!ok && tid == 0) atomicOr(a.err, 1u);
}

// One LSD pass of a workgroup of 1024 lanes over `cnt` records: src -> dst by the `bits`-wide digit at `shift`, stable.
// src / dst may live in LDS or in global memory (the caller's pointers decide; inlined per call site).
__device__ __forceinline__ void gs_wg_pass(const u64* src, u64* dst, uint32_t cnt, int shift, int bits, uint32_t (&s_wcnt)[16][256],
                                           uint32_t (&s_cur)[256], uint32_t (&s_wtot)[4]) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t m = (1u << bits) - 1u;
    if (tid < 256) s_cur[tid] = 0u;
    __syncthreads();
    for (uint32_t i = tid; i < cnt; i += 1024u) atomicAdd(&s_cur[(uint32_t)(src[i] >> shift) & m], 1u);
    __syncthreads();
    const uint32_t hv = tid < 256 ? s_cur[tid] : 0u;
    const uint32_t ex = gs_scan256(hv, s_wtot);
    if (tid < 256) s_cur[tid] = ex;                                // next output slot of each digit
    __syncthreads();
    for (uint32_t c0 = 0; c0 < cnt; c0 += 1024u) {                 // chunks in order: the pass is stable
        const uint32_t i = c0 + (uint32_t)tid;
        const bool vd = i < cnt;
        const u64 r = vd ? src[i] : 0ull;
        const uint32_t d = (uint32_t)(r >> shift) & m;
        const u64 peers = gs_match(d, vd, bits);
        const uint32_t rank = (uint32_t)__popcll(peers & ((1ull << lane) - 1ull));
        const bool lead = vd && lane == __ffsll((long long)peers) - 1;
        if (lead) s_wcnt[wave][d] = (uint32_t)__popcll(peers);   // (entries of absent digits are zero: every leader clears its own below)
        __syncthreads();
        if (tid < 256) {
            uint32_t run = s_cur[tid];
#pragma unroll
            for (int w = 0; w < 16; ++w) { const uint32_t t = s_wcnt[w][tid]; s_wcnt[w][tid] = run; run += t; }
            s_cur[tid] = run;
        }
        __syncthreads();
        if (vd) dst[s_wcnt[wave][d] + rank] = r;
        __syncthreads();
        // back to "all zero" for the next chunk: every digit this wavefront saw has exactly one leader
        if (lead) s_wcnt[wave][d] = 0u;
        // (bases written for digits a wavefront did NOT hold must go as well)
        if (tid < 256) {
#pragma unroll
            for (int w = 0; w < 16; ++w) s_wcnt[w][tid] = 0u;
        }
        __syncthreads();
    }
}

__global__ __launch_bounds__(1024) void gsort_bucket_kernel(GsortArgs a) {
    __shared__ u64 s_buf[2][GS_LDS_CAP];
    __shared__ uint32_t s_wcnt[16][256];
    __shared__ uint32_t s_cur[256];
    __shared__ uint32_t s_wtot[4];
    __shared__ uint32_t s_range[2];
    const int tid = threadIdx.x;
    const uint32_t hv = tid < 256 ? a.hist[tid] : 0u;
    const uint32_t ex = gs_scan256(hv, s_wtot);
    if (tid == (int)blockIdx.x) { s_range[0] = ex; s_range[1] = hv; }
    for (int x = tid; x < 16 * 256; x += 1024) (&s_wcnt[0][0])[x] = 0u;
    __syncthreads();
    const uint32_t base = s_range[0], cnt = s_range[1];
    if (cnt == 0u) return;
    // what the partition left to sort: the key bits below its digit, 8 bits or fewer per pass
    const int kb = a.kb_dev != nullptr ? (int)*a.kb_dev : a.kb;
    const int rem = kb > 8 ? kb - 8 : 0;
    const int passes = (rem + 7) / 8;
    const int w = passes > 0 ? (rem + passes - 1) / passes : 0;
    const u64* fin;
    if (cnt <= GS_RANK_CAP) {
        // a small bucket (the usual one of a single scan's batch: a few dozen records): rank by counting inside LDS -- the record
        // key << 18 | input position is unique, its order IS the stable key order -- instead of up to three LSD passes of a
        // 1024-lane workgroup with four barriers per chunk
        if (tid < (int)cnt) s_buf[0][tid] = (u64)a.part[base + tid];
        __syncthreads();
        if (tid >= (int)cnt) return;
        const u64 r = s_buf[0][tid];
        uint32_t rank = 0;
        for (uint32_t j = 0; j < cnt; ++j) rank += s_buf[0][j] < r ? 1u : 0u;      // (every lane reads the same word: a broadcast)
        a.keys_out[base + rank] = r >> GS_IB;
        a.vals_out[base + rank] = a.vals[(uint32_t)(r & GS_IMASK)];
        return;
    }
    if (cnt <= GS_LDS_CAP) {
        for (uint32_t i = tid; i < cnt; i += 1024u) s_buf[0][i] = (u64)a.part[base + i];
        __syncthreads();
        int cur = 0;
        for (int p = 0; p < passes; ++p) {
            const int bits = rem - p * w < w ? rem - p * w : w;
            gs_wg_pass(s_buf[cur], s_buf[cur ^ 1], cnt, GS_IB + p * w, bits, s_wcnt, s_cur, s_wtot);
            cur ^= 1;
        }
        fin = s_buf[cur];
        for (uint32_t i = tid; i < cnt; i += 1024u) {
            const u64 r = fin[i];
            a.keys_out[base + i] = r >> GS_IB;
            a.vals_out[base + i] = a.vals[(uint32_t)(r & GS_IMASK)];
        }
        return;
    }
    // a bucket beyond LDS: the same passes through global memory, between this bucket's ranges of `part` and of the
    // (consumed) key array's twin `part2`
    u64* A = reinterpret_cast<u64*>(a.part) + base;
    u64* B = reinterpret_cast<u64*>(a.part2) + base;
    for (int p = 0; p < passes; ++p) {
        const int bits = rem - p * w < w ? rem - p * w : w;
        gs_wg_pass(A, B, cnt, GS_IB + p * w, bits, s_wcnt, s_cur, s_wtot);
        __threadfence();
        __syncthreads();
        u64* t = A; A = B; B = t;
    }
    for (uint32_t i = tid; i < cnt; i += 1024u) {
        const u64 r = A[i];
        a.keys_out[base + i] = r >> GS_IB;
        a.vals_out[base + i] = a.vals[(uint32_t)(r & GS_IMASK)];
    }
}

hipError_t launch_gsort_large(const GsortArgs& a, hipStream_t s) {
    if (a.n == 0) return hipSuccess;
    if (a.n > GSORT_MAX || (a.kb_dev == nullptr && a.kb + GS_IB > 64)) return hipErrorInvalidValue;   // (device-resident key bits: BATCH_KB_MAX)
    hipLaunchKernelGGL(gsort_partition_kernel, dim3((a.n + GS_TILE - 1u) / GS_TILE), dim3(512), 0, s, a);
    hipLaunchKernelGGL(gsort_bucket_kernel, dim3(256), dim3(1024), 0, s, a);
    return hipGetLastError();
}

// ---- LSD form: a stable sort of up to GSORT_MAX records as ceil(kb / 8) stable splits by one 8-bit digit each, least significant
// digit first -- every pass is the partition kernel above with the digit position as a parameter.  For keys whose DISTRIBUTION is
// skewed (the voxel filter's leaf indices: the ground layer of a sweep fills a handful of the 256 top-digit buckets with tens of
// thousands of records each, far beyond what gsort_bucket_kernel sorts inside LDS) the MSD form above degenerates; this one does
// not care.  The caller's key kernel accumulates the histograms of ALL digit positions (hist[pass][256], zero on entry).
struct LsdPassArgs {
    const uint64_t* keys;           // first pass: the keys (< 2^kb); the record becomes key << GSORT_INDEX_BITS | input position
    const uint64_t* in;             // later passes: records
    uint64_t*       out;            // records (every pass but the last)
    uint64_t*       keys_out;       // last pass: sorted keys ...
    uint32_t*       vals_out;       // ... and the input positions in sorted order
    uint32_t        n;
    int             shift;          // digit = (key >> shift) & 255
    const uint32_t* hist;           // [256] of this digit
    unsigned long long* agg;
    uint32_t        stamp;
    uint32_t*       err;
};
template <bool FIRST, bool LAST>
__global__ __launch_bounds__(512) void gsort_lsd_pass_kernel(LsdPassArgs a) {
    __shared__ uint32_t s_wcnt[8][256];
    __shared__ uint32_t s_base[256];
    __shared__ uint32_t s_half[2][256];
    __shared__ uint32_t s_wtot[4];
    __shared__ uint32_t s_bad;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t n = a.n;
    for (int x = tid; x < 8 * 256; x += 512) (&s_wcnt[0][0])[x] = 0u;
    if (tid == 0) s_bad = 0u;
    __syncthreads();
    const uint32_t i0 = blockIdx.x * GS_TILE + (uint32_t)wave * 256u;
    u64 rec[4];
    uint32_t off[4], dg[4];
    bool vd[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const uint32_t i = i0 + (uint32_t)k * 64u + (uint32_t)lane;
        vd[k] = i < n;
        if (FIRST) rec[k] = ((vd[k] ? (u64)a.keys[i] : 0ull) << GS_IB) | (u64)i;
        else rec[k] = vd[k] ? (u64)a.in[i] : 0ull;
        dg[k] = (uint32_t)((rec[k] >> GS_IB) >> a.shift) & 255u;
        const u64 peers = gs_match(dg[k], vd[k], 8);
        const uint32_t rank = (uint32_t)__popcll(peers & ((1ull << lane) - 1ull));
        const int leader = vd[k] ? __ffsll((long long)peers) - 1 : lane;
        uint32_t old = 0u;
        if (vd[k] && lane == leader) {
            old = s_wcnt[wave][dg[k]];
            s_wcnt[wave][dg[k]] = old + (uint32_t)__popcll(peers);
        }
        gs_wave_sync();
        old = (uint32_t)__shfl((int)old, leader, 64);
        off[k] = old + rank;
    }
    __syncthreads();
    if (tid < 256) {
        uint32_t run = 0u;
#pragma unroll
        for (int w = 0; w < 8; ++w) { const uint32_t t = s_wcnt[w][tid]; s_wcnt[w][tid] = run; run += t; }
        __hip_atomic_store(a.agg + (size_t)blockIdx.x * 256u + (uint32_t)tid, (u64)run | ((u64)a.stamp << 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    {
        const uint32_t d = (uint32_t)tid & 255u, h = (uint32_t)tid >> 8, b = blockIdx.x;
        uint32_t acc = 0u;
        bool bad = false;
        for (uint32_t p0 = h; p0 < b; p0 += 32u) {              // sixteen words in flight per lane, tiles h, h + 2, ...
            u64 v[16];
            unsigned spins = 0;
            bool ok;
            do {
                ok = true;
#pragma unroll
                for (int u = 0; u < 16; ++u) {
                    const uint32_t p = p0 + 2u * (uint32_t)u;
                    v[u] = (u64)a.stamp << 32;
                    if (p < b) v[u] = __hip_atomic_load(a.agg + (size_t)p * 256u + d, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    ok = ok && (uint32_t)(v[u] >> 32) == a.stamp;
                }
            } while (!ok && ++spins < GS_SPIN_LIMIT);
            bad = bad || !ok;
#pragma unroll
            for (int u = 0; u < 16; ++u) acc += (uint32_t)v[u];
        }
        s_half[h][d] = acc;
        if (bad) s_bad = 1u;
    }
    const uint32_t hv = tid < 256 ? a.hist[tid] : 0u;
    const uint32_t gbase = gs_scan256(hv, s_wtot);
    if (tid < 256) s_base[tid] = gbase + s_half[0][tid] + s_half[1][tid];
    __syncthreads();
    const bool ok = s_bad == 0u;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        if (!(vd[k] && ok)) continue;
        const uint32_t pos = s_base[dg[k]] + s_wcnt[wave][dg[k]] + off[k];
        if (LAST) { a.keys_out[pos] = rec[k] >> GS_IB; a.vals_out[pos] = (uint32_t)(rec[k] & GS_IMASK); }
        else a.out[pos] = rec[k];
    }
    if (!ok && tid == 0) atomicOr(a.err, 1u);
}

hipError_t launch_gsort_lsd(const GsortLsdArgs& g, hipStream_t s) {
    if (g.n == 0) return hipSuccess;
    const int passes = g.passes > 0 ? g.passes : (g.kb + 7) / 8;
    if (g.n > GSORT_MAX || g.kb + GS_IB > 64 || passes < 1 || passes > 4) return hipErrorInvalidValue;
    const dim3 grid((g.n + GS_TILE - 1u) / GS_TILE), block(512);
    uint64_t* buf[2] = {g.part, g.part2};
    for (int p = 0; p < passes; ++p) {
        LsdPassArgs a{};
        a.keys = g.keys; a.in = p > 0 ? buf[(p - 1) & 1] : nullptr; a.out = buf[p & 1]; a.keys_out = g.keys_out; a.vals_out = g.vals_out;
        a.n = g.n; a.shift = 8 * p; a.hist = g.hist + 256 * p; a.agg = g.agg; a.stamp = g.stamp0 + (uint32_t)p; a.err = g.err;
        const bool first = p == 0, last = p == passes - 1;
        if (first && last)  hipLaunchKernelGGL((gsort_lsd_pass_kernel<true, true>), grid, block, 0, s, a);
        else if (first)     hipLaunchKernelGGL((gsort_lsd_pass_kernel<true, false>), grid, block, 0, s, a);
        else if (last)      hipLaunchKernelGGL((gsort_lsd_pass_kernel<false, true>), grid, block, 0, s, a);
        else                hipLaunchKernelGGL((gsort_lsd_pass_kernel<false, false>), grid, block, 0, s, a);
    }
    return hipGetLastError();
}

// histogram of the partition digit for callers whose key generation cannot accumulate it (fallbacks, tests)
__global__ __launch_bounds__(256) void gsort_hist_kernel(const uint64_t* keys, uint32_t n, const uint32_t* n_dev, uint32_t n_dev_add, int kb, uint32_t* hist) {
    __shared__ uint32_t s_hist[256];
    const uint32_t nr = n_dev != nullptr ? min(n, *n_dev + n_dev_add) : n;
    s_hist[threadIdx.x] = 0u;
    __syncthreads();
    for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < nr; i += gridDim.x * 256u) atomicAdd(&s_hist[gsort_digit(keys[i], kb)], 1u);
    __syncthreads();
    const uint32_t v = s_hist[threadIdx.x];
    if (v != 0u) atomicAdd(hist + threadIdx.x, v);
}
hipError_t launch_gsort_hist(const uint64_t* keys, uint32_t n, const uint32_t* n_dev, uint32_t n_dev_add, int kb, uint32_t* hist, hipStream_t s) {
    if (n == 0) return hipSuccess;
    uint32_t b = (n + 2047u) / 2048u;
    if (b > 256u) b = 256u;
    hipLaunchKernelGGL(gsort_hist_kernel, dim3(b), dim3(256), 0, s, keys, n, n_dev, n_dev_add, kb, hist);
    return hipGetLastError();
}

}  // namespace lio
