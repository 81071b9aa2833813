// lio_tile.h -- the two device helpers of every single-launch, order-preserving compaction (lio_compact.hip, lio_scanprep.hip): the
// rank of a flagged item inside its 256-item tile, and the exclusive prefix of the tile's count(s) over the tiles before it through
// stamped 8-byte words {count | stamp << 32} (an 8-byte store is single-copy atomic: whoever reads the stamp has the count).  All
// tiles of a launch are resident together (<= 1024 workgroups of 256 threads) and a tile only ever waits for lower-numbered ones;
// the waits are bounded.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace lio {

constexpr unsigned COMPACT_SPIN_LIMIT = 1u << 22;

// exclusive prefix of `mine` over the tiles before this one (K independent streams), the tile's own count published
// first.  All 256 threads call it; returns false when a wait ran out.
template <int K>
__device__ __forceinline__ bool tile_exclusive_prefix(const uint32_t (&mine)[K], unsigned long long* agg, uint32_t stamp, uint32_t (&before)[K], int tile = -1) {
    __shared__ uint32_t s_sum[K][4];
    __shared__ uint32_t s_bad;
    const int tid = threadIdx.x, b = tile >= 0 ? tile : (int)blockIdx.x;
    if (tid == 0) s_bad = 0u;
    __syncthreads();
    if (tid < K) __hip_atomic_store(agg + (size_t)b * K + tid, (unsigned long long)mine[tid] | ((unsigned long long)stamp << 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    uint32_t acc[K];
#pragma unroll
    for (int k = 0; k < K; ++k) acc[k] = 0u;
    bool bad = false;
    for (int p = tid; p < b; p += 256) {
#pragma unroll
        for (int k = 0; k < K; ++k) {
            unsigned long long v;
            unsigned spins = 0;
            do { v = __hip_atomic_load(agg + (size_t)p * K + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
            while ((uint32_t)(v >> 32) != stamp && ++spins < COMPACT_SPIN_LIMIT);
            bad = bad || (uint32_t)(v >> 32) != stamp;
            acc[k] += (uint32_t)v;
        }
    }
#pragma unroll
    for (int k = 0; k < K; ++k) {
        uint32_t v = acc[k];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v += (uint32_t)__shfl_xor((int)v, o, 64);
        if ((tid & 63) == 0) s_sum[k][tid >> 6] = v;
    }
    if (bad) s_bad = 1u;
    __syncthreads();
#pragma unroll
    for (int k = 0; k < K; ++k) before[k] = s_sum[k][0] + s_sum[k][1] + s_sum[k][2] + s_sum[k][3];
    return s_bad == 0u;
}

// position of this thread's item among the flagged items of its 256-item tile (exclusive), and the tile's total
__device__ __forceinline__ uint32_t tile_rank(bool flag, uint32_t& tile_total, uint32_t (&s_wave)[4]) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const unsigned long long m = __ballot(flag);
    const uint32_t in_wave = (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
    if (lane == 0) s_wave[wave] = (uint32_t)__popcll(m);
    __syncthreads();
    uint32_t off = 0;
    for (int w = 0; w < wave; ++w) off += s_wave[w];
    tile_total = s_wave[0] + s_wave[1] + s_wave[2] + s_wave[3];
    return off + in_wave;
}

}  // namespace lio
