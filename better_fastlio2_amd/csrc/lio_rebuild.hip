// lio_rebuild.hip -- full index rebuild by sorting CELL RUNS instead of points (lio_rebuild.h), gfx950 only.
//
// Replaces the point sort of Build / reconstruct (include/ikd-Tree/ikd_Tree.cpp:353-364, 1394-1405: the reference re-balances a
// pointer tree; here the index is an order of the point array).  HBM-bound integer work: two streaming reads of the compacted
// points (run heads: count, emit), a radix sort of the runs (rocPRIM, ~30 key bits), one gather / scatter of the points run by
// run.  Tiles of 2048 points per workgroup, loads coalesced (lane-strided), keys staged in LDS for the neighbour compare.
#include <cstring>
#include <hip/hip_runtime.h>
#include <rocprim/rocprim.hpp>
#include "lio_rebuild.h"

namespace lio {

// keys (packed cell coordinates) of a tile into LDS (s_key[RUN_TILE + 1]: entry 0 = the key of the point in front of the tile, ~0 for
// the first tile -- no packed key has bit 63), the cell coordinates' extrema of this thread's points into lo / hi
__device__ __forceinline__ void tile_keys(const float4* __restrict__ pts, uint32_t n, Grid grid, uint32_t tile, uint64_t* s_key, uint32_t (&lo)[3], uint32_t (&hi)[3]) {
    const uint32_t tid = threadIdx.x, i0 = tile * RUN_TILE;
    float4 p[8];
#pragma unroll
    for (uint32_t u = 0; u < 8u; ++u) {
        const uint32_t i = i0 + u * 256u + tid;
        p[u] = i < n ? pts[i] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    lo[0] = lo[1] = lo[2] = 0xffffffffu; hi[0] = hi[1] = hi[2] = 0u;
#pragma unroll
    for (uint32_t u = 0; u < 8u; ++u) {
        const uint32_t i = i0 + u * 256u + tid;
        const uint32_t c[3] = {(uint32_t)cell_coord(p[u].x, grid.inv_cell), (uint32_t)cell_coord(p[u].y, grid.inv_cell), (uint32_t)cell_coord(p[u].z, grid.inv_cell)};
        s_key[1u + u * 256u + tid] = pack_key(c[0], c[1], c[2]);
        if (i < n) {
#pragma unroll
            for (int ax = 0; ax < 3; ++ax) { lo[ax] = min(lo[ax], c[ax]); hi[ax] = max(hi[ax], c[ax]); }
        }
    }
    if (tid == 0) {
        uint64_t k = ~0ull;
        if (i0 > 0u) {
            const float4 q = pts[i0 - 1u];
            k = pack_key((uint32_t)cell_coord(q.x, grid.inv_cell), (uint32_t)cell_coord(q.y, grid.inv_cell), (uint32_t)cell_coord(q.z, grid.inv_cell));
        }
        s_key[0] = k;
    }
    __syncthreads();
}

__global__ __launch_bounds__(256) void cell_runs_count_kernel(const float4* __restrict__ pts, uint32_t n, Grid grid, uint32_t* __restrict__ tile_cnt,
                                                              uint32_t* __restrict__ tile_box, uint32_t n_tiles) {
    __shared__ uint64_t s_key[RUN_TILE + 1];
    __shared__ uint32_t s_w[4][7];
    const uint32_t t = blockIdx.x, tid = threadIdx.x;
    if (t >= n_tiles) { if (tid == 0) tile_cnt[t] = 0u; return; }
    uint32_t lo[3], hi[3];
    tile_keys(pts, n, grid, t, s_key, lo, hi);
    // heads among this thread's eight CONSECUTIVE points (the emit kernel ranks them in the same arrangement)
    uint32_t c = 0;
#pragma unroll
    for (uint32_t u = 0; u < 8u; ++u) {
        const uint32_t j = tid * 8u + u;
        c += (t * RUN_TILE + j < n && s_key[1u + j] != s_key[j]) ? 1u : 0u;
    }
    uint32_t v[7] = {c, lo[0], lo[1], lo[2], hi[0], hi[1], hi[2]};
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        v[0] += (uint32_t)__shfl_xor((int)v[0], o, 64);
#pragma unroll
        for (int k = 1; k < 4; ++k) v[k] = min(v[k], (uint32_t)__shfl_xor((int)v[k], o, 64));
#pragma unroll
        for (int k = 4; k < 7; ++k) v[k] = max(v[k], (uint32_t)__shfl_xor((int)v[k], o, 64));
    }
    if ((tid & 63u) == 0u) {
#pragma unroll
        for (int k = 0; k < 7; ++k) s_w[tid >> 6][k] = v[k];
    }
    __syncthreads();
    if (tid == 0) tile_cnt[t] = s_w[0][0] + s_w[1][0] + s_w[2][0] + s_w[3][0];
    if (tid >= 1u && tid < 4u) tile_box[(size_t)t * 6u + tid - 1u] = min(min(s_w[0][tid], s_w[1][tid]), min(s_w[2][tid], s_w[3][tid]));
    if (tid >= 4u && tid < 7u) tile_box[(size_t)t * 6u + tid - 1u] = max(max(s_w[0][tid], s_w[1][tid]), max(s_w[2][tid], s_w[3][tid]));
}

__global__ __launch_bounds__(256) void cell_runs_emit_kernel(const float4* __restrict__ pts, uint32_t n, Grid grid, const uint32_t* __restrict__ tile_off,
                                                             uint64_t* __restrict__ run_key, uint32_t* __restrict__ run_start) {
    __shared__ uint64_t s_key[RUN_TILE + 1];
    __shared__ uint32_t s_w[4];
    const uint32_t t = blockIdx.x, tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    uint32_t lo[3], hi[3];
    tile_keys(pts, n, grid, t, s_key, lo, hi);
    uint32_t c = 0, heads = 0;
#pragma unroll
    for (uint32_t u = 0; u < 8u; ++u) {
        const uint32_t j = tid * 8u + u;
        const bool h = t * RUN_TILE + j < n && s_key[1u + j] != s_key[j];
        heads |= h ? 1u << u : 0u;
        c += h ? 1u : 0u;
    }
    uint32_t inc = c;
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
    for (uint32_t u = 0; u < 8u; ++u) {
        if (heads & (1u << u)) {
            const uint32_t j = tid * 8u + u;
            run_key[pos] = s_key[1u + j];
            run_start[pos] = t * RUN_TILE + j;
            ++pos;
        }
    }
}

__global__ __launch_bounds__(1024) void cell_box_publish_kernel(const uint32_t* __restrict__ tile_box, uint32_t n_tiles, const uint32_t* __restrict__ n_runs,
                                                                const uint32_t* __restrict__ check, uint32_t* dst_mapped, uint32_t seq) {
    __shared__ uint32_t s_w[16][6];
    const uint32_t tid = threadIdx.x;
    uint32_t v[6] = {0xffffffffu, 0xffffffffu, 0xffffffffu, 0u, 0u, 0u};
    for (uint32_t t = tid; t < n_tiles; t += 1024u) {
#pragma unroll
        for (int k = 0; k < 3; ++k) { v[k] = min(v[k], tile_box[(size_t)t * 6u + k]); v[3 + k] = max(v[3 + k], tile_box[(size_t)t * 6u + 3 + k]); }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
#pragma unroll
        for (int k = 0; k < 3; ++k) { v[k] = min(v[k], (uint32_t)__shfl_xor((int)v[k], o, 64)); v[3 + k] = max(v[3 + k], (uint32_t)__shfl_xor((int)v[3 + k], o, 64)); }
    }
    if ((tid & 63u) == 0u) {
#pragma unroll
        for (int k = 0; k < 6; ++k) s_w[tid >> 6][k] = v[k];
    }
    __syncthreads();
    // (no system-scope fence: it would write back an L2 full of what the kernels before this one have just produced -- the words go
    // to host-mapped memory as system-scope stores, are acknowledged, then the sequence number follows the same way: cell_apply_kernel)
    if (tid < 6u) {
        uint32_t r = s_w[0][tid];
        for (int w = 1; w < 16; ++w) r = tid < 3u ? min(r, s_w[w][tid]) : max(r, s_w[w][tid]);
        __hip_atomic_store(dst_mapped + 1u + tid, r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    if (tid == 6u) __hip_atomic_store(dst_mapped, *n_runs, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    if (tid == 7u) __hip_atomic_store(dst_mapped + 7, check != nullptr ? *check : 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) __hip_atomic_store(dst_mapped + 32, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// run r: its cell's packed coordinates -> the compressed sort key (KEY = uint32_t when the frame's keys fit: a third less to move per radix
// pass), and the value that rides with it through the sort:
// first point << 32 | number of points (the next run's first point is known by now)
template <class KEY>
__global__ __launch_bounds__(256) void compress_run_keys_kernel(const uint64_t* __restrict__ run_key, const uint32_t* __restrict__ run_start, uint32_t n_runs,
                                                                uint32_t n, CellKeyFrame f, KEY* __restrict__ sort_key, uint64_t* __restrict__ run_val, uint32_t* err) {
    bool bad = false;
    for (uint32_t r = blockIdx.x * 256u + threadIdx.x; r < n_runs; r += gridDim.x * 256u) {
        bool ok;
        const uint64_t pk = run_key[r];
        sort_key[r] = (KEY)cell_key_compress(f, (uint32_t)(pk >> 42) & 0x1fffffu, (uint32_t)(pk >> 21) & 0x1fffffu, (uint32_t)pk & 0x1fffffu, &ok);
        bad = bad || !ok;
        const uint32_t s0 = run_start[r], s1 = r + 1u < n_runs ? run_start[r + 1u] : n;
        run_val[r] = ((uint64_t)s0 << 32) | (uint64_t)(s1 - s0);
    }
    if (__ballot(bad) != 0ull && (threadIdx.x & 63u) == 0u) atomicOr(err, 2u);
}

// One lane per sorted run: its points to their final place.  Adjacent lanes write adjacent destinations; the reads are the
// scattered side (a run is 16 x len contiguous bytes).  Also: the Morton code of the run's cell (for the cell table that
// follows) and, per tree level L, whether this run starts a new cell of that level -- keys differ in a bit of digit >= L.
template <class KEY>
__global__ __launch_bounds__(256) void runs_place_kernel(const float4* __restrict__ src, float4* __restrict__ dst, const KEY* __restrict__ sorted_key,
                                                         const uint64_t* __restrict__ sorted_val, const uint32_t* __restrict__ dst_start, uint32_t n_runs,
                                                         CellKeyFrame f, uint64_t* __restrict__ abs_key, uint32_t* __restrict__ level_parts) {
    __shared__ uint32_t s_lv[TREE_MAX_LEVELS];
    const uint32_t tid = threadIdx.x;
    if (tid < (uint32_t)TREE_MAX_LEVELS) s_lv[tid] = 0u;
    __syncthreads();
    for (uint32_t j0 = blockIdx.x * 256u; j0 < n_runs; j0 += gridDim.x * 256u) {       // (block-uniform bound)
        const uint32_t j = j0 + tid;
        int top = -1;                                             // highest level at which run j starts a new cell
        if (j < n_runs) {
            const uint64_t v = sorted_val[j];
            const uint32_t s0 = (uint32_t)(v >> 32), len = (uint32_t)v;
            const uint32_t d0 = dst_start[j];
            const uint64_t mk = cell_key_expand(f, (uint64_t)sorted_key[j]);
            abs_key[j] = mk;
            if (j == 0u) top = TREE_MAX_LEVELS - 1;
            else {
                const uint64_t x = mk ^ cell_key_expand(f, (uint64_t)sorted_key[j - 1u]);
                if (x != 0ull) top = (63 - __clzll((long long)x)) / 6;
            }
            for (uint32_t k = 0; k < len; ++k) dst[d0 + k] = src[s0 + k];
        }
#pragma unroll
        for (int L = 0; L < TREE_MAX_LEVELS; ++L) {
            const unsigned long long m = __ballot(top >= L);
            if ((tid & 63u) == 0u && m != 0ull) atomicAdd(&s_lv[L], (uint32_t)__popcll(m));
        }
    }
    __syncthreads();
    if (tid < (uint32_t)TREE_MAX_LEVELS) level_parts[(size_t)blockIdx.x * TREE_MAX_LEVELS + tid] = s_lv[tid];
}

__global__ __launch_bounds__(1024) void level_counts_publish_kernel(const uint32_t* __restrict__ level_parts, uint32_t n_parts, const uint32_t* err,
                                                                    uint32_t* dst_mapped, uint32_t seq) {
    __shared__ uint32_t s_w[16][TREE_MAX_LEVELS];
    const uint32_t tid = threadIdx.x;
    uint32_t v[TREE_MAX_LEVELS];
#pragma unroll
    for (int L = 0; L < TREE_MAX_LEVELS; ++L) v[L] = 0u;
    for (uint32_t p = tid; p < n_parts; p += 1024u) {
#pragma unroll
        for (int L = 0; L < TREE_MAX_LEVELS; ++L) v[L] += level_parts[(size_t)p * TREE_MAX_LEVELS + L];
    }
#pragma unroll
    for (int L = 0; L < TREE_MAX_LEVELS; ++L) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v[L] += (uint32_t)__shfl_xor((int)v[L], o, 64);
        if ((tid & 63u) == 0u) s_w[tid >> 6][L] = v[L];
    }
    __syncthreads();
    if (tid < (uint32_t)TREE_MAX_LEVELS) {
        uint32_t r = 0;
        for (int w = 0; w < 16; ++w) r += s_w[w][tid];
        __hip_atomic_store(dst_mapped + tid, r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    if (tid == (uint32_t)TREE_MAX_LEVELS) __hip_atomic_store(dst_mapped + tid, *err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) __hip_atomic_store(dst_mapped + 32, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

static inline uint32_t tiles_of(uint32_t n) { return (n + RUN_TILE - 1u) / RUN_TILE; }

hipError_t launch_cell_runs_count(const float4* pts, uint32_t n, Grid grid, uint32_t* tile_cnt, uint32_t* tile_box, hipStream_t s) {
    const uint32_t nt = tiles_of(n);
    hipLaunchKernelGGL(cell_runs_count_kernel, dim3(nt + 1u), dim3(256), 0, s, pts, n, grid, tile_cnt, tile_box, nt);
    return hipGetLastError();
}
hipError_t launch_cell_runs_emit(const float4* pts, uint32_t n, Grid grid, const uint32_t* tile_off, uint64_t* run_key, uint32_t* run_start, hipStream_t s) {
    const uint32_t nt = tiles_of(n);
    hipLaunchKernelGGL(cell_runs_emit_kernel, dim3(nt), dim3(256), 0, s, pts, n, grid, tile_off, run_key, run_start);
    return hipGetLastError();
}
hipError_t launch_cell_box_publish(const uint32_t* tile_box, uint32_t n_tiles, const uint32_t* n_runs, const uint32_t* check, uint32_t* dst_mapped, uint32_t seq,
                                   hipStream_t s) {
    hipLaunchKernelGGL(cell_box_publish_kernel, dim3(1), dim3(1024), 0, s, tile_box, n_tiles, n_runs, check, dst_mapped, seq);
    return hipGetLastError();
}
hipError_t launch_compress_run_keys(const uint64_t* run_key, const uint32_t* run_start, uint32_t n_runs, uint32_t n, const CellKeyFrame& f, void* sort_key,
                                    uint64_t* run_val, uint32_t* err, hipStream_t s) {
    uint32_t b = (n_runs + 255u) / 256u;
    if (b > 4096u) b = 4096u;
    if (f.bits <= 32) hipLaunchKernelGGL(compress_run_keys_kernel<uint32_t>, dim3(b), dim3(256), 0, s, run_key, run_start, n_runs, n, f, (uint32_t*)sort_key, run_val, err);
    else hipLaunchKernelGGL(compress_run_keys_kernel<uint64_t>, dim3(b), dim3(256), 0, s, run_key, run_start, n_runs, n, f, (uint64_t*)sort_key, run_val, err);
    return hipGetLastError();
}
hipError_t launch_runs_place(const float4* src, float4* dst, const void* sorted_key, const uint64_t* sorted_val, const uint32_t* dst_start,
                             uint32_t n_runs, const CellKeyFrame& f, uint64_t* abs_key, uint32_t* level_parts, hipStream_t s) {
    if (f.bits <= 32)
        hipLaunchKernelGGL(runs_place_kernel<uint32_t>, dim3(PLACE_WORKGROUPS), dim3(256), 0, s, src, dst, (const uint32_t*)sorted_key, sorted_val, dst_start, n_runs, f,
                           abs_key, level_parts);
    else
        hipLaunchKernelGGL(runs_place_kernel<uint64_t>, dim3(PLACE_WORKGROUPS), dim3(256), 0, s, src, dst, (const uint64_t*)sorted_key, sorted_val, dst_start, n_runs, f,
                           abs_key, level_parts);
    return hipGetLastError();
}
hipError_t launch_level_counts_publish(const uint32_t* level_parts, const uint32_t* err, uint32_t* dst_mapped, uint32_t seq, hipStream_t s) {
    hipLaunchKernelGGL(level_counts_publish_kernel, dim3(1), dim3(1024), 0, s, level_parts, PLACE_WORKGROUPS, err, dst_mapped, seq);
    return hipGetLastError();
}

// ---- the library halves: radix sort of the runs, scan of their lengths, runs -> cells --------------------------------------
template <class KEY>
static hipError_t sort_runs_t(void* temp, size_t temp_bytes, const KEY* keys_in, KEY* keys_out, const uint64_t* vals_in, uint64_t* vals_out, uint32_t n_runs, int bits,
                              hipStream_t s) {
    size_t need = 0;
    hipError_t e = rocprim::radix_sort_pairs(nullptr, need, keys_in, keys_out, vals_in, vals_out, (size_t)n_runs, 0u, (unsigned)bits, s);
    if (e != hipSuccess) return e;
    if (need > temp_bytes) return hipErrorOutOfMemory;
    return rocprim::radix_sort_pairs(temp, need, keys_in, keys_out, vals_in, vals_out, (size_t)n_runs, 0u, (unsigned)bits, s);
}
hipError_t sort_runs(void* temp, size_t temp_bytes, const void* keys_in, void* keys_out, const uint64_t* vals_in, uint64_t* vals_out, uint32_t n_runs, int bits,
                     hipStream_t s) {
    if (bits <= 32) return sort_runs_t<uint32_t>(temp, temp_bytes, (const uint32_t*)keys_in, (uint32_t*)keys_out, vals_in, vals_out, n_runs, bits, s);
    return sort_runs_t<uint64_t>(temp, temp_bytes, (const uint64_t*)keys_in, (uint64_t*)keys_out, vals_in, vals_out, n_runs, bits, s);
}
struct RunLength {                            // points of a run: the low half of its value
    __host__ __device__ uint32_t operator()(uint64_t v) const { return (uint32_t)v; }
};
hipError_t scan_run_lengths(void* temp, size_t temp_bytes, const uint64_t* sorted_val, uint32_t n_runs, uint32_t* dst_start, hipStream_t s) {
    auto len = rocprim::make_transform_iterator(sorted_val, RunLength{});
    size_t need = 0;
    hipError_t e = rocprim::exclusive_scan(nullptr, need, len, dst_start, 0u, (size_t)n_runs, rocprim::plus<uint32_t>(), s);
    if (e != hipSuccess) return e;
    if (need > temp_bytes) return hipErrorOutOfMemory;
    return rocprim::exclusive_scan(temp, need, len, dst_start, 0u, (size_t)n_runs, rocprim::plus<uint32_t>(), s);
}
// cells = runs of equal abs_key; cell_val[c] = first point of the cell << 32 | its points: the reduction keeps the first run's start and
// adds the lengths (associative, applied in order) -- the cells' starts need no scan of their own
struct RunPlaced {
    const uint32_t* dst_start;
    const uint64_t* sorted_val;
    __host__ __device__ uint64_t operator()(uint32_t j) const { return ((uint64_t)dst_start[j] << 32) | (uint64_t)(uint32_t)sorted_val[j]; }
};
struct FirstStartSumLength {
    __host__ __device__ uint64_t operator()(uint64_t a, uint64_t b) const { return (a & 0xffffffff00000000ull) | (uint64_t)((uint32_t)a + (uint32_t)b); }
};
hipError_t reduce_runs_to_cells(void* temp, size_t temp_bytes, const uint64_t* abs_key, const uint64_t* sorted_val, const uint32_t* dst_start, uint32_t n_runs,
                                uint64_t* cell_keys, uint64_t* cell_val, uint32_t* n_cells, hipStream_t s) {
    auto val = rocprim::make_transform_iterator(rocprim::counting_iterator<uint32_t>(0u), RunPlaced{dst_start, sorted_val});
    size_t need = 0;
    hipError_t e = rocprim::reduce_by_key(nullptr, need, abs_key, val, (size_t)n_runs, cell_keys, cell_val, n_cells, FirstStartSumLength(),
                                          rocprim::equal_to<uint64_t>(), s);
    if (e != hipSuccess) return e;
    if (need > temp_bytes) return hipErrorOutOfMemory;
    return rocprim::reduce_by_key(temp, need, abs_key, val, (size_t)n_runs, cell_keys, cell_val, n_cells, FirstStartSumLength(), rocprim::equal_to<uint64_t>(), s);
}
// level 1 from the packed cell records: keys >> 6, the points summed (coarsen_level of lio_sort.hip for the packed form)
struct ShiftRight6 { __host__ __device__ uint64_t operator()(uint64_t k) const { return k >> 6; } };
hipError_t coarsen_cells_packed(void* temp, size_t temp_bytes, const uint64_t* cell_keys, const uint64_t* cell_val, size_t n, uint64_t* keys_out, uint32_t* counts_out,
                                uint32_t* n_out, hipStream_t s) {
    auto kin = rocprim::make_transform_iterator(cell_keys, ShiftRight6{});
    auto cin = rocprim::make_transform_iterator(cell_val, RunLength{});
    size_t need = 0;
    hipError_t e = rocprim::reduce_by_key(nullptr, need, kin, cin, n, keys_out, counts_out, n_out, rocprim::plus<uint32_t>(), rocprim::equal_to<uint64_t>(), s);
    if (e != hipSuccess) return e;
    if (need > temp_bytes) return hipErrorOutOfMemory;
    return rocprim::reduce_by_key(temp, need, kin, cin, n, keys_out, counts_out, n_out, rocprim::plus<uint32_t>(), rocprim::equal_to<uint64_t>(), s);
}

}  // namespace lio
