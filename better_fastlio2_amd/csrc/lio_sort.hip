// lio_sort.hip -- the library primitives of the map (re)build: radix sort, run-length encode,
// exclusive scan, stream compaction.  These are plain device-wide primitives (rocPRIM, shipped
// with ROCm); the hot path of a scan step never calls them.
#include <cstring>
#include <hip/hip_runtime.h>
#include <rocprim/rocprim.hpp>
#include "lio_kernels.h"

namespace lio {

size_t sort_pairs_temp_bytes(size_t n) {
    size_t bytes = 0;
    (void)rocprim::radix_sort_pairs(nullptr, bytes, (const uint64_t*)nullptr, (uint64_t*)nullptr,
                                    (const uint32_t*)nullptr, (uint32_t*)nullptr, n, 0, 63, (hipStream_t)0);
    return bytes;
}
hipError_t sort_pairs_u64_u32(void* temp, size_t temp_bytes, const uint64_t* keys_in, uint64_t* keys_out,
                              const uint32_t* vals_in, uint32_t* vals_out, size_t n, hipStream_t s) {
    return rocprim::radix_sort_pairs(temp, temp_bytes, keys_in, keys_out, vals_in, vals_out, n, 0, 63, s);
}

size_t rle_temp_bytes(size_t n) {
    size_t bytes = 0;
    (void)rocprim::run_length_encode(nullptr, bytes, (const uint64_t*)nullptr, n, (uint64_t*)nullptr,
                                     (uint32_t*)nullptr, (uint32_t*)nullptr, (hipStream_t)0);
    return bytes;
}
hipError_t sort_keys_u32(void* temp, size_t temp_bytes, const uint32_t* keys_in, uint32_t* keys_out, size_t n, hipStream_t s) {
    return rocprim::radix_sort_keys(temp, temp_bytes, keys_in, keys_out, n, 0, 32, s);
}
hipError_t run_length_encode_u64(void* temp, size_t temp_bytes, const uint64_t* keys, size_t n,
                                 uint64_t* unique_out, uint32_t* counts_out, uint32_t* n_runs_out, hipStream_t s) {
    return rocprim::run_length_encode(temp, temp_bytes, keys, n, unique_out, counts_out, n_runs_out, s);
}

struct ShiftRight {
    int shift;
    __host__ __device__ uint64_t operator()(uint64_t k) const { return k >> shift; }
};
hipError_t run_length_encode_shifted(void* temp, size_t temp_bytes, const uint64_t* keys, size_t n, int shift,
                                     uint64_t* unique_out, uint32_t* counts_out, uint32_t* n_runs_out, hipStream_t s) {
    auto it = rocprim::make_transform_iterator(keys, ShiftRight{shift});
    size_t need = 0;
    hipError_t e = rocprim::run_length_encode(nullptr, need, it, n, unique_out, counts_out, n_runs_out, s);
    if (e != hipSuccess) return e;
    if (need > temp_bytes) return hipErrorOutOfMemory;
    return rocprim::run_length_encode(temp, temp_bytes, it, n, unique_out, counts_out, n_runs_out, s);
}

size_t scan_temp_bytes(size_t n) {
    size_t bytes = 0;
    (void)rocprim::exclusive_scan(nullptr, bytes, (const uint32_t*)nullptr, (uint32_t*)nullptr, 0u, n,
                                  rocprim::plus<uint32_t>(), (hipStream_t)0);
    return bytes;
}
hipError_t exclusive_scan_u32(void* temp, size_t temp_bytes, const uint32_t* in, uint32_t* out, size_t n, hipStream_t s) {
    return rocprim::exclusive_scan(temp, temp_bytes, in, out, 0u, n, rocprim::plus<uint32_t>(), s);
}

struct LivePoint {
    __host__ __device__ bool operator()(const float4& p) const { return p.w == 0.0f; }
};
size_t select_temp_bytes(size_t n) {
    size_t bytes = 0;
    (void)rocprim::select(nullptr, bytes, (const float4*)nullptr, (float4*)nullptr, (uint32_t*)nullptr, n,
                          LivePoint(), (hipStream_t)0);
    return bytes;
}
hipError_t select_live_points(void* temp, size_t temp_bytes, const float4* src, float4* dst, uint32_t* n_out,
                              size_t n, hipStream_t s) {
    return rocprim::select(temp, temp_bytes, src, dst, n_out, n, LivePoint(), s);
}

struct BoxDeletedPoint {
    __device__ bool operator()(const float4& p) const { return p.w == TOMB_BOX; }
};
hipError_t select_box_deleted_points(void* temp, size_t temp_bytes, const float4* src, float4* dst, uint32_t* n_out,
                                     size_t n, hipStream_t s) {
    return rocprim::select(temp, temp_bytes, src, dst, n_out, n, BoxDeletedPoint(), s);
}

// next coarser tree level from the current one: keys >> 6, counts summed per run
hipError_t coarsen_level(void* temp, size_t temp_bytes, const uint64_t* keys_in, const uint32_t* counts_in, size_t n,
                         uint64_t* keys_out, uint32_t* counts_out, uint32_t* n_out, hipStream_t s) {
    auto kin = rocprim::make_transform_iterator(keys_in, ShiftRight{6});
    size_t need = 0;
    hipError_t e = rocprim::reduce_by_key(nullptr, need, kin, counts_in, n, keys_out, counts_out, n_out,
                                          rocprim::plus<uint32_t>(), rocprim::equal_to<uint64_t>(), s);
    if (e != hipSuccess) return e;
    if (need > temp_bytes) return hipErrorOutOfMemory;
    return rocprim::reduce_by_key(temp, temp_bytes, kin, counts_in, n, keys_out, counts_out, n_out,
                                  rocprim::plus<uint32_t>(), rocprim::equal_to<uint64_t>(), s);
}

// merge of two key-sorted (key, value) sequences (stable: ties take the first sequence first)
size_t merge_temp_bytes(size_t n1, size_t n2) {
    size_t bytes = 0;
    (void)rocprim::merge(nullptr, bytes, (const uint64_t*)nullptr, (const uint64_t*)nullptr, (uint64_t*)nullptr,
                         (const uint32_t*)nullptr, (const uint32_t*)nullptr, (uint32_t*)nullptr, n1, n2,
                         rocprim::less<uint64_t>(), (hipStream_t)0);
    return bytes;
}
hipError_t merge_pairs_u64_u32(void* temp, size_t temp_bytes, const uint64_t* k1, const uint64_t* k2, uint64_t* kout,
                               const uint32_t* v1, const uint32_t* v2, uint32_t* vout, size_t n1, size_t n2, hipStream_t s) {
    size_t need = 0;
    hipError_t e = rocprim::merge(nullptr, need, k1, k2, kout, v1, v2, vout, n1, n2, rocprim::less<uint64_t>(), s);
    if (e != hipSuccess) return e;
    if (need > temp_bytes) return hipErrorOutOfMemory;
    return rocprim::merge(temp, temp_bytes, k1, k2, kout, v1, v2, vout, n1, n2, rocprim::less<uint64_t>(), s);
}

// (key, index) pairs whose point is live (pts[index].w == 0), order-preserving
struct LiveIndex {
    const float4* pts;
    __host__ __device__ bool operator()(uint32_t i) const { return pts[i].w == 0.0f; }
};
hipError_t select_live_pairs(void* temp, size_t temp_bytes, const uint64_t* keys_in, const uint32_t* vals_in,
                             const float4* pts, uint64_t* keys_out, uint32_t* vals_out, uint32_t* n_out, size_t n,
                             hipStream_t s) {
    auto flags = rocprim::make_transform_iterator(vals_in, LiveIndex{pts});
    size_t need = 0, need2 = 0;
    hipError_t e = rocprim::select(nullptr, need, keys_in, flags, keys_out, n_out, n, s);
    if (e != hipSuccess) return e;
    e = rocprim::select(nullptr, need2, vals_in, flags, vals_out, n_out, n, s);
    if (e != hipSuccess) return e;
    if (need > temp_bytes || need2 > temp_bytes) return hipErrorOutOfMemory;
    e = rocprim::select(temp, temp_bytes, keys_in, flags, keys_out, n_out, n, s);
    if (e != hipSuccess) return e;
    return rocprim::select(temp, temp_bytes, vals_in, flags, vals_out, n_out, n, s);
}
// merge with an implicit identity value sequence 0..n1-1 for the first input
hipError_t merge_pairs_identity(void* temp, size_t temp_bytes, const uint64_t* k1, const uint64_t* k2, uint64_t* kout,
                                const uint32_t* v2, uint32_t* vout, size_t n1, size_t n2, hipStream_t s) {
    rocprim::counting_iterator<uint32_t> v1(0u);
    size_t need = 0;
    hipError_t e = rocprim::merge(nullptr, need, k1, k2, kout, v1, v2, vout, n1, n2, rocprim::less<uint64_t>(), s);
    if (e != hipSuccess) return e;
    if (need > temp_bytes) return hipErrorOutOfMemory;
    return rocprim::merge(temp, temp_bytes, k1, k2, kout, v1, v2, vout, n1, n2, rocprim::less<uint64_t>(), s);
}

// first positions of the runs of equal MAJOR keys (high 32 bits) in a sorted key array, and their number
struct RunHead {
    const uint64_t* keys;
    __host__ __device__ bool operator()(uint32_t i) const { return i == 0u || (keys[i] >> 32) != (keys[i - 1] >> 32); }
};
hipError_t select_run_heads_u64(void* temp, size_t temp_bytes, const uint64_t* sorted_keys, size_t n, uint32_t* starts_out,
                                uint32_t* n_runs_out, hipStream_t s) {
    rocprim::counting_iterator<uint32_t> idx(0u);
    size_t need = 0;
    hipError_t e = rocprim::select(nullptr, need, idx, starts_out, n_runs_out, n, RunHead{sorted_keys}, s);
    if (e != hipSuccess) return e;
    if (need > temp_bytes) return hipErrorOutOfMemory;
    return rocprim::select(temp, temp_bytes, idx, starts_out, n_runs_out, n, RunHead{sorted_keys}, s);
}

struct ClassIs {
    uint8_t want;
    __host__ __device__ bool operator()(uint8_t c) const { return c == want; }
};
// order-preserving compaction of the points whose class byte equals `want`
hipError_t select_points_by_class(void* temp, size_t temp_bytes, const float4* src, const uint8_t* cls, uint8_t want,
                                  float4* dst, uint32_t* n_out, size_t n, hipStream_t s) {
    auto flags = rocprim::make_transform_iterator(cls, ClassIs{want});
    size_t need = 0;
    hipError_t e = rocprim::select(nullptr, need, src, flags, dst, n_out, n, s);
    if (e != hipSuccess) return e;
    if (need > temp_bytes) return hipErrorOutOfMemory;
    return rocprim::select(temp, temp_bytes, src, flags, dst, n_out, n, s);
}

}  // namespace lio
