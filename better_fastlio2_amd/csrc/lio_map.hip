// lio_map.hip -- construction and mutation kernels of the GPU-resident map index, gfx950 only.
//
// Replaces the structural side of KD_TREE<PointType> (include/ikd-Tree/ikd_Tree.cpp): Build (:353),
// Delete_Point_Boxes (:536) / Delete_by_range (:648-710), flatten (:1326).  The index is rebuilt
// by sorting (no pointer tree, no re-balancing thread): Morton key per point -> radix sort
// (rocPRIM, lio_sort.hip) -> gather -> run-length encode cells -> hash-table insert.
#include "lio_device.h"
#include "lio_kernels.h"
#include "lio_math.h"

namespace lio {

__global__ __launch_bounds__(256) void point_keys_kernel(const float4* pts, uint32_t n, Grid grid,
                                                         uint64_t* mkeys, uint32_t* vals, uint32_t base_index) {
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const float4 p = pts[i];
        const uint32_t cx = (uint32_t)cell_coord(p.x, grid.inv_cell);
        const uint32_t cy = (uint32_t)cell_coord(p.y, grid.inv_cell);
        const uint32_t cz = (uint32_t)cell_coord(p.z, grid.inv_cell);
        mkeys[i] = morton3(cx, cy, cz);
        vals[i] = base_index + i;
    }
}

__global__ __launch_bounds__(256) void gather_points_kernel(const float4* src, const uint32_t* vals, uint32_t n,
                                                            float4* dst) {
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x)
        dst[i] = src[vals[i]];
}

// one lane per occupied cell: claim a slot by CAS on the 64-bit key, then publish {start, count}
__global__ __launch_bounds__(256) void table_insert_kernel(const uint64_t* cell_mkeys, const uint32_t* cell_start,
                                                           const uint32_t* cell_count, uint32_t n_cells,
                                                           HashEntry* table, uint32_t mask) {
    for (uint32_t c = blockIdx.x * blockDim.x + threadIdx.x; c < n_cells; c += gridDim.x * blockDim.x) {
        const uint64_t mk = cell_mkeys[c];
        const uint64_t key = pack_key(compact21(mk >> 2), compact21(mk >> 1), compact21(mk));
        uint32_t slot = hash_key(key) & mask;
        for (;;) {
            unsigned long long* kp = reinterpret_cast<unsigned long long*>(&table[slot]);
            const unsigned long long prev = atomicCAS(kp, (unsigned long long)KEY_EMPTY, (unsigned long long)key);
            if (prev == KEY_EMPTY || prev == key) {
                table[slot].start = cell_start != nullptr ? cell_start[c] : c;          // (nullptr: the entry's rank)
                table[slot].count = cell_count[c];
                break;
            }
            slot = (slot + 1) & mask;
        }
    }
}

__global__ __launch_bounds__(256) void iota_kernel(uint32_t* out, uint32_t n) {
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) out[i] = i;
}

// one lane per occupied level-0 cell: its {start, count} goes to its fixed place in the child block of its
// level-1 cell (block number = the `start` field of the level-1 entry)
__global__ __launch_bounds__(256) void fill_child_blocks_kernel(const uint64_t* cell_mkeys, const uint32_t* cell_start,
                                                                const uint32_t* cell_count, uint32_t n_cells,
                                                                const HashEntry* l1table, uint32_t l1mask, uint2* child) {
    for (uint32_t c = blockIdx.x * blockDim.x + threadIdx.x; c < n_cells; c += gridDim.x * blockDim.x) {
        const uint64_t mk = cell_mkeys[c];
        const uint32_t cx = compact21(mk >> 2), cy = compact21(mk >> 1), cz = compact21(mk);
        uint32_t blk, one;
        table_find(l1table, l1mask, pack_key(cx >> 2, cy >> 2, cz >> 2), blk, one);
        if (one == 0u) continue;                                   // cannot happen: the level-1 cell was derived from this one
        child[(size_t)blk * 64u + (((cx & 3u) << 4) | ((cy & 3u) << 2) | (cz & 3u))] = make_uint2(cell_start[c], cell_count[c]);
    }
}

// the same from the packed cell records of the rebuild by cell runs (first point << 32 | points); lane 0 of the launch also sets the
// number of child blocks in use (*block_bump) and clears the per-level counters of new entries (level_new[TREE_MAX_LEVELS]): two small
// copies less on the stream
__global__ __launch_bounds__(256) void fill_child_blocks_packed_kernel(const uint64_t* cell_mkeys, const uint64_t* cell_val, uint32_t n_cells,
                                                                       const HashEntry* l1table, uint32_t l1mask, uint2* child, uint32_t* block_bump,
                                                                       uint32_t n_blocks, uint32_t* level_new) {
    if (blockIdx.x == 0 && threadIdx.x < (uint32_t)TREE_MAX_LEVELS) level_new[threadIdx.x] = 0u;
    if (blockIdx.x == 0 && threadIdx.x == 0) *block_bump = n_blocks;
    for (uint32_t c = blockIdx.x * blockDim.x + threadIdx.x; c < n_cells; c += gridDim.x * blockDim.x) {
        const uint64_t mk = cell_mkeys[c];
        const uint32_t cx = compact21(mk >> 2), cy = compact21(mk >> 1), cz = compact21(mk);
        uint32_t blk, one;
        table_find(l1table, l1mask, pack_key(cx >> 2, cy >> 2, cz >> 2), blk, one);
        if (one == 0u) continue;
        const uint64_t v = cell_val[c];
        child[(size_t)blk * 64u + (((cx & 3u) << 4) | ((cy & 3u) << 2) | (cz & 3u))] = make_uint2((uint32_t)(v >> 32), (uint32_t)v);
    }
}

// Delete_by_range (ikd_Tree.cpp:648-710): a live point is removed when
// min <= p < max on every axis for any of the boxes.
__global__ __launch_bounds__(256) void delete_boxes_kernel(float4* pts, uint32_t n, const float* boxes6, int nb,
                                                           uint32_t* counter) {
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        float4 p = pts[i];
        if (p.w != 0.0f) continue;
        bool hit = false;
        for (int b = 0; b < nb; ++b) {
            const float* bx = boxes6 + 6 * b;
            if (bx[0] <= p.x && bx[3] > p.x && bx[1] <= p.y && bx[4] > p.y && bx[2] <= p.z && bx[5] > p.z) hit = true;
        }
        if (hit) {
            p.w = TOMB_BOX;
            pts[i] = p;
            atomicAdd(counter, 1u);
        }
    }
}

// box-deleted points that have just been copied out for acquire_removed_points are not reported again
__global__ __launch_bounds__(256) void mark_reported_kernel(float4* pts, uint32_t n) {
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x)
        if (pts[i].w == TOMB_BOX) pts[i].w = TOMB_BOX_REPORTED;
}

__global__ __launch_bounds__(256) void pack_xyz_kernel(const float4* src, uint32_t n, float* dst) {
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const float4 p = src[i];
        dst[3 * (size_t)i + 0] = p.x; dst[3 * (size_t)i + 1] = p.y; dst[3 * (size_t)i + 2] = p.z;
    }
}

// [5][stride] float4 -> the caller-facing layout of Nearest_Points: xyz [n][5][3], sqdist [n][5]
__global__ __launch_bounds__(256) void unpack_neighbors_kernel(const float4* nbr, const uint8_t* cnt, int nq,
                                                               int stride, float* out_xyz, float* out_d, int* out_cnt) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nq) return;
#pragma unroll
    for (int k = 0; k < K; ++k) {
        const float4 p = nbr[(size_t)k * stride + i];
        if (out_xyz != nullptr) {
            float* o = out_xyz + ((size_t)i * K + k) * 3;
            o[0] = p.x; o[1] = p.y; o[2] = p.z;
        }
        if (out_d != nullptr) out_d[(size_t)i * K + k] = p.w;
    }
    if (out_cnt != nullptr) out_cnt[i] = (int)(cnt[i] & ~CNT_UNPROVEN);
}

static inline dim3 grid_for(uint32_t n) {
    uint32_t b = (n + 255u) / 256u;
    if (b > 4096u) b = 4096u;
    if (b == 0u) b = 1u;
    return dim3(b);
}

hipError_t launch_point_keys(const float4* pts, uint32_t n, Grid grid, uint64_t* mkeys, uint32_t* vals, hipStream_t s,
                             uint32_t base_index) {
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(point_keys_kernel, grid_for(n), dim3(256), 0, s, pts, n, grid, mkeys, vals, base_index);
    return hipGetLastError();
}
hipError_t launch_gather_points(const float4* src, const uint32_t* vals, uint32_t n, float4* dst, hipStream_t s) {
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(gather_points_kernel, grid_for(n), dim3(256), 0, s, src, vals, n, dst);
    return hipGetLastError();
}
hipError_t launch_table_insert(const uint64_t* cell_mkeys, const uint32_t* cell_start, const uint32_t* cell_count,
                               uint32_t n_cells, HashEntry* table, uint32_t table_mask, hipStream_t s) {
    if (n_cells == 0) return hipSuccess;
    hipLaunchKernelGGL(table_insert_kernel, grid_for(n_cells), dim3(256), 0, s, cell_mkeys, cell_start, cell_count,
                       n_cells, table, table_mask);
    return hipGetLastError();
}
// Copies n (<= 32) words into host-mapped memory and then publishes a sequence number there with a system-scope
// release: the host polls that word instead of calling into the runtime to wait for the stream.
__global__ void publish_words_kernel(const uint32_t* src, int n, uint32_t* dst_mapped, uint32_t seq) {
    const int t = threadIdx.x;
    if (t < n) dst_mapped[t] = src[t];
    __threadfence_system();
    __syncthreads();
    if (t == 0) __hip_atomic_store(dst_mapped + 32, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
hipError_t launch_publish_words(const uint32_t* src, int n, uint32_t* dst_mapped, uint32_t seq, hipStream_t s) {
    hipLaunchKernelGGL(publish_words_kernel, dim3(1), dim3(64), 0, s, src, n, dst_mapped, seq);
    return hipGetLastError();
}
hipError_t launch_iota(uint32_t* out, uint32_t n, hipStream_t s) {
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(iota_kernel, grid_for(n), dim3(256), 0, s, out, n);
    return hipGetLastError();
}
hipError_t launch_fill_child_blocks(const uint64_t* cell_mkeys, const uint32_t* cell_start, const uint32_t* cell_count,
                                    uint32_t n_cells, const HashEntry* l1table, uint32_t l1mask, uint2* child, hipStream_t s) {
    if (n_cells == 0) return hipSuccess;
    hipLaunchKernelGGL(fill_child_blocks_kernel, grid_for(n_cells), dim3(256), 0, s, cell_mkeys, cell_start, cell_count,
                       n_cells, l1table, l1mask, child);
    return hipGetLastError();
}
hipError_t launch_fill_child_blocks_packed(const uint64_t* cell_mkeys, const uint64_t* cell_val, uint32_t n_cells, const HashEntry* l1table, uint32_t l1mask,
                                           uint2* child, uint32_t* block_bump, uint32_t n_blocks, uint32_t* level_new, hipStream_t s) {
    hipLaunchKernelGGL(fill_child_blocks_packed_kernel, grid_for(n_cells), dim3(256), 0, s, cell_mkeys, cell_val, n_cells, l1table, l1mask, child, block_bump,
                       n_blocks, level_new);
    return hipGetLastError();
}
hipError_t launch_delete_boxes(float4* pts, uint32_t n, const float* boxes6, int nb, uint32_t* counter, hipStream_t s) {
    if (n == 0 || nb <= 0) return hipSuccess;
    hipLaunchKernelGGL(delete_boxes_kernel, grid_for(n), dim3(256), 0, s, pts, n, boxes6, nb, counter);
    return hipGetLastError();
}
hipError_t launch_mark_reported(float4* pts, uint32_t n, hipStream_t s) {
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(mark_reported_kernel, grid_for(n), dim3(256), 0, s, pts, n);
    return hipGetLastError();
}
hipError_t launch_pack_xyz(const float4* src, uint32_t n, float* dst_xyz, hipStream_t s) {
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(pack_xyz_kernel, grid_for(n), dim3(256), 0, s, src, n, dst_xyz);
    return hipGetLastError();
}
hipError_t launch_unpack_neighbors(const float4* nbr, const uint8_t* cnt, int nq, int nq_stride, float* out_xyz,
                                   float* out_d, int* out_cnt, hipStream_t s) {
    if (nq <= 0) return hipSuccess;
    hipLaunchKernelGGL(unpack_neighbors_kernel, dim3((nq + 255) / 256), dim3(256), 0, s, nbr, cnt, nq, nq_stride,
                       out_xyz, out_d, out_cnt);
    return hipGetLastError();
}

}  // namespace lio
