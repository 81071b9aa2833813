// lio_scanprep.hip -- the step BEFORE the measurement path, gfx950 only:
// motion compensation of the raw scan and voxel-grid down-sampling, so that the raw scan is uploaded
// once and feats_down_body never visits the host.
//
//   undistort_kernel   ImuProcess::UndistortPcl's backward pass (src/IMU_Processing.hpp:334-386): each
//                      point (time-sorted, :251) is moved into the LiDAR frame at scan end with the IMU
//                      pose that precedes it: P_c = R_LI^T (R_e^T (R_i (R_LI P + t_LI) + T_ei) - t_LI),
//                      R_i = R_head Exp(gyr_tail, dt) (include/math_tools.h:39-61), fp64.
//   voxel grid         pcl::VoxelGrid<PointType>::applyFilter (PCL is a third-party dependency of the
//                      reference, un-vendored; PCL 1.10 by distro): leaf index from
//                      floor(p * inverse_leaf) - min_b, points grouped by leaf (stable sort: time order
//                      inside a leaf), centroid = fp32 sum in that order / n (CentroidPoint), output in
//                      ascending leaf order.  Only x, y, z are carried (the path reads nothing else).
#include "lio_kernels.h"
#include "lio_math.h"
#include "lio_scanprep.h"
#include "lio_tile.h"

namespace lio {

__device__ __forceinline__ void mat3_mul(const double a[9], const double b[9], double r[9]) {
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) r[i * 3 + j] = (a[i * 3] * b[j] + a[i * 3 + 1] * b[3 + j]) + a[i * 3 + 2] * b[6 + j];
}

__device__ __forceinline__ int float_ordered(float f);
__device__ __forceinline__ uint32_t float_ordered_u(float f);
// parts (optional): [workgroup][8] = getMinMax3D of the workgroup's compensated points as order images (~min x, y, z, max x, y, z;
// 0 = no finite point), [6] = 1 when one of its points is earlier than its predecessor in the sweep -- what minmax_kernel would
// otherwise read the cloud again for (the voxel filter's key kernel folds the workgroups' words, voxel_index_compact_kernel)
__global__ __launch_bounds__(256) void undistort_kernel(const float4* raw, int n, const double* poses, int n_poses,
                                                        Pose14 end, float4* out, uint32_t* parts) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    const bool valid = i < n;
    const float4 p = valid ? raw[i] : make_float4(0.f, 0.f, 0.f, 0.f);
    float4 o = p;
    if (valid) {
    const double t = (double)p.w / double(1000);                       // curvature holds the offset in ms (:351)
    // the IMU interval whose head precedes the point: largest k <= K-2 with poses[k].offset_time < t
    int k = -1;
    for (int j = 0; j + 1 < n_poses; ++j)
        if (poses[j * 22] < t) k = j;
    if (k >= 0) {
        const double* head = poses + (size_t)k * 22;
        const double* tail = head + 22;
        const double dt = t - head[0];                                  // :353
        const double* acc = tail + 1;                                   // acc_imu  << tail->acc   (:346)
        const double* gyr = tail + 4;                                   // angvel   << tail->gyr   (:347)
        const double* vel = head + 7;                                   // vel_imu  << head->vel   (:344)
        const double* pos = head + 10;                                  // pos_imu  << head->pos   (:345)
        const double* Rm = head + 13;                                   // R_imu    << head->rot   (:343)
        // Exp(angvel, dt) (math_tools.h:39-61)
        double E[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
        const double nrm = sqrt((gyr[0] * gyr[0] + gyr[1] * gyr[1]) + gyr[2] * gyr[2]);
        if (nrm > 0.0000001) {
            const double ax = gyr[0] / nrm, ay = gyr[1] / nrm, az = gyr[2] / nrm;
            const double Kx[9] = {0.0, -az, ay, az, 0.0, -ax, -ay, ax, 0.0};
            double KK[9];
            mat3_mul(Kx, Kx, KK);
            const double ang = nrm * dt;
            const double sn = sin(ang), cs = 1.0 - cos(ang);
#pragma unroll
            for (int q = 0; q < 9; ++q) E[q] = (E[q] + sn * Kx[q]) + cs * KK[q];
        }
        double Ri[9];
        mat3_mul(Rm, E, Ri);                                            // R_i = R_imu * Exp(...)  (:355)
        const PoseQ s = pose_unpack(end);                               // imu_state at scan end
        const D3 Pi{(double)p.x, (double)p.y, (double)p.z};
        const D3 Tei{((pos[0] + vel[0] * dt) + ((0.5 * acc[0]) * dt) * dt) - s.pos.x,      // :359
                     ((pos[1] + vel[1] * dt) + ((0.5 * acc[1]) * dt) * dt) - s.pos.y,
                     ((pos[2] + vel[2] * dt) + ((0.5 * acc[2]) * dt) * dt) - s.pos.z};
        const D3 pl = d3_add(q_rotate(s.offR, Pi), s.offT);             // offset_R_L_I * P_i + offset_T_L_I
        const D3 in{((Ri[0] * pl.x + Ri[1] * pl.y) + Ri[2] * pl.z) + Tei.x,
                    ((Ri[3] * pl.x + Ri[4] * pl.y) + Ri[5] * pl.z) + Tei.y,
                    ((Ri[6] * pl.x + Ri[7] * pl.y) + Ri[8] * pl.z) + Tei.z};
        D3 b = q_rotate(q_conj(s.rot), in);                             // imu_state.rot.conjugate() * (...)
        b.x -= s.offT.x; b.y -= s.offT.y; b.z -= s.offT.z;
        const D3 c = q_rotate(q_conj(s.offR), b);                       // :372-374
        o.x = (float)c.x; o.y = (float)c.y; o.z = (float)c.z;
    }
    out[i] = o;
    }
    if (parts == nullptr) return;
    __shared__ uint32_t s_mm[4][7];
    uint32_t m[7] = {0u, 0u, 0u, 0u, 0u, 0u, 0u};
    if (valid) {
        if (i > 0 && float_ordered_u(p.w) < float_ordered_u(raw[i - 1].w)) m[6] = 1u;
        if (isfinite(o.x) && isfinite(o.y) && isfinite(o.z)) {
            const uint32_t v[3] = {float_ordered_u(o.x), float_ordered_u(o.y), float_ordered_u(o.z)};
#pragma unroll
            for (int a = 0; a < 3; ++a) { m[a] = ~v[a]; m[3 + a] = v[a]; }
        }
    }
#pragma unroll
    for (int a = 0; a < 7; ++a) {
#pragma unroll
        for (int sh = 32; sh > 0; sh >>= 1) m[a] = max(m[a], (uint32_t)__shfl_xor((int)m[a], sh, 64));
    }
    if ((threadIdx.x & 63) == 0) {
#pragma unroll
        for (int a = 0; a < 7; ++a) s_mm[threadIdx.x >> 6][a] = m[a];
    }
    __syncthreads();
    if (threadIdx.x < 7) parts[(size_t)blockIdx.x * 8u + threadIdx.x] = max(max(s_mm[0][threadIdx.x], s_mm[1][threadIdx.x]), max(s_mm[2][threadIdx.x], s_mm[3][threadIdx.x]));
}

// ---- voxel grid ------------------------------------------------------------------------------------
__device__ __forceinline__ int float_ordered(float f) {                // monotone float -> int
    const int i = __float_as_int(f);
    return i >= 0 ? i : i ^ 0x7fffffff;
}
__device__ __forceinline__ uint32_t float_ordered_u(float f) { return (uint32_t)float_ordered(f) ^ 0x80000000u; }
__host__ __device__ inline float ordered_float(int i) {
    const int j = i >= 0 ? i : i ^ 0x7fffffff;
    union { int i; float f; } u;
    u.i = j;
    return u.f;
}

// sort key of the time offset (stable radix sort = std::sort(..., time_list) with ties in input order)
__global__ __launch_bounds__(256) void time_keys_kernel(const float4* pts, int n, uint64_t* keys, uint32_t* order) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    keys[i] = (uint64_t)float_ordered_u(pts[i].w);
    order[i] = (uint32_t)i;
}

// getMinMax3D: per-workgroup reduction in LDS, then six atomics per workgroup on unsigned order-preserving images
// (min as the maximum of the complement, so that a zeroed record is the neutral element of all six)
__device__ void voxel_params(VoxelGridDev* vox, const uint32_t mm[6], float inv_leaf);
__global__ __launch_bounds__(256) void minmax_kernel(const float4* pts, int n, VoxelGridDev* vox, float inv_leaf, bool check_time_order) {
    __shared__ uint32_t s_mm[4][6];
    uint32_t m[6] = {0u, 0u, 0u, 0u, 0u, 0u};
    bool earlier = false;                                               // some point lies before its predecessor in time
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const float4 p = pts[i];
        if (check_time_order && i > 0) earlier = earlier || float_ordered_u(p.w) < float_ordered_u(pts[i - 1].w);
        if (!(isfinite(p.x) && isfinite(p.y) && isfinite(p.z))) continue;
        const uint32_t v[3] = {float_ordered_u(p.x), float_ordered_u(p.y), float_ordered_u(p.z)};
#pragma unroll
        for (int a = 0; a < 3; ++a) { m[a] = max(m[a], ~v[a]); m[3 + a] = max(m[3 + a], v[a]); }
    }
#pragma unroll
    for (int a = 0; a < 6; ++a) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) m[a] = max(m[a], (uint32_t)__shfl_xor((int)m[a], o, 64));
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (check_time_order && __any(earlier) && lane == 0) atomicOr(&vox->probe[0], 1u);
    if (lane == 0) {
#pragma unroll
        for (int a = 0; a < 6; ++a) s_mm[wave][a] = m[a];
    }
    __syncthreads();
    if (threadIdx.x < 6) {
        const uint32_t v = max(max(s_mm[0][threadIdx.x], s_mm[1][threadIdx.x]), max(s_mm[2][threadIdx.x], s_mm[3][threadIdx.x]));
        if (v != 0u) atomicMax(&vox->mm[threadIdx.x], v);
    }
    // The last workgroup to arrive derives min_b_ / divb_mul_ from the extremes (a launch of its own before: one wavefront,
    // 4.8 us).  The extremes were combined by device-scope atomics; the ticket is taken behind them (release), read back at
    // device scope (acquire).
    __shared__ uint32_t s_ticket;
    __syncthreads();
    if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        s_ticket = __hip_atomic_fetch_add(&vox->arrive, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    if (s_ticket != gridDim.x - 1u || threadIdx.x != 0) return;
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    uint32_t all[6];
#pragma unroll
    for (int a = 0; a < 6; ++a) all[a] = __hip_atomic_load(&vox->mm[a], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    voxel_params(vox, all, inv_leaf);
}

// min_b_, div_b_, divb_mul_ from the bounding box (voxel_grid.hpp, first pass), on the device: one lane
__device__ void voxel_params_of(const uint32_t mm[6], float inv_leaf, VoxelGridParams& g, uint32_t& flags, uint32_t& bits) {
    g.inv_leaf = inv_leaf;
    flags = 0u;
    if (mm[0] == 0u) flags |= VOX_EMPTY;                                // no finite point
    long long div[3] = {1, 1, 1};
    for (int a = 0; a < 3; ++a) {
        const float lo = ordered_float((int)((~mm[a]) ^ 0x80000000u)), hi = ordered_float((int)(mm[3 + a] ^ 0x80000000u));
        g.min_b[a] = (int)floorf(lo * inv_leaf);
        const int max_b = (int)floorf(hi * inv_leaf);
        div[a] = (long long)max_b - g.min_b[a] + 1;
    }
    if (!(flags & VOX_EMPTY) && div[0] * div[1] * div[2] > (long long)0x7fffffff) flags |= VOX_OVERFLOW;
    g.divb_mul[0] = 1; g.divb_mul[1] = (int)div[0]; g.divb_mul[2] = (int)(div[0] * div[1]);
    // bits of a leaf index: 2^bits > number of leaves, so that 2^bits - 1 lies beyond every leaf (the key of the dropped points)
    bits = 1u;
    if (flags == 0u) {
        const unsigned long long leaves = (unsigned long long)(div[0] * div[1] * div[2]);
        while ((1ull << bits) <= leaves) ++bits;
    }
}
__device__ void voxel_params(VoxelGridDev* vox, const uint32_t mm[6], float inv_leaf) {
    VoxelGridParams g;
    uint32_t flags, bits;
    voxel_params_of(mm, inv_leaf, g, flags, bits);
    vox->g = g;
    vox->flags = flags;
    vox->probe[1] = bits;
}

// key = leaf index (major) | time offset (minor, when the cloud is not in time order yet): one stable sort yields the
// leaves in ascending order with their points in time order, ties in input order -- what the stable sort by leaf of
// a time-sorted cloud yields
__global__ __launch_bounds__(256) void voxel_index_kernel(const float4* pts, int n, const VoxelGridDev* vox, bool time_minor,
                                                          uint64_t* keys, uint32_t* order) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const VoxelGridParams g = vox->g;
    const float4 p = pts[i];
    uint64_t key = ~0ull;                                               // non-finite points sort last and are dropped
    if (isfinite(p.x) && isfinite(p.y) && isfinite(p.z)) {
        // ijk = (int)(floor(p * inverse_leaf) - (float)min_b)          (voxel_grid.hpp, first pass)
        const int i0 = (int)(floorf(p.x * g.inv_leaf) - (float)g.min_b[0]);
        const int i1 = (int)(floorf(p.y * g.inv_leaf) - (float)g.min_b[1]);
        const int i2 = (int)(floorf(p.z * g.inv_leaf) - (float)g.min_b[2]);
        const int idx = i0 * g.divb_mul[0] + i1 * g.divb_mul[1] + i2 * g.divb_mul[2];
        key = ((uint64_t)(uint32_t)idx << 32) | (time_minor ? (uint64_t)float_ordered_u(p.w) : 0ull);
    }
    keys[i] = key;
    order[i] = (uint32_t)i;
}

// The compact form of the key for a cloud that already is in time order: the leaf index alone (stability supplies the rest).
// Every workgroup adds its share to the histograms of the four 8-bit digits of the key (LDS first, one atomic per non-empty bin).
// parts (optional; undistort_kernel's per-workgroup extremes): no minmax_kernel ran -- every workgroup folds the n_parts records
// itself (a few KB out of the L2) and derives the grid; workgroup 0 leaves it in `vox` for the kernels behind this one
__global__ __launch_bounds__(256) void voxel_index_compact_kernel(const float4* pts, int n, VoxelGridDev* vox, uint64_t* keys, const uint32_t* parts,
                                                                  uint32_t n_parts, float inv_leaf, bool check_time_order) {
    __shared__ uint32_t s_hist[4][256];
    __shared__ uint32_t s_mm[4][7];
    __shared__ VoxelGridParams s_g;
    __shared__ uint32_t s_bits;
    for (int x = threadIdx.x; x < 4 * 256; x += 256) (&s_hist[0][0])[x] = 0u;
    if (parts != nullptr) {
        uint32_t m[7] = {0u, 0u, 0u, 0u, 0u, 0u, 0u};
        for (uint32_t q = threadIdx.x; q < n_parts; q += 256u) {
#pragma unroll
            for (int a = 0; a < 7; ++a) m[a] = max(m[a], parts[(size_t)q * 8u + a]);
        }
#pragma unroll
        for (int a = 0; a < 7; ++a) {
#pragma unroll
            for (int sh = 32; sh > 0; sh >>= 1) m[a] = max(m[a], (uint32_t)__shfl_xor((int)m[a], sh, 64));
        }
        if ((threadIdx.x & 63) == 0) {
#pragma unroll
            for (int a = 0; a < 7; ++a) s_mm[threadIdx.x >> 6][a] = m[a];
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            uint32_t all[7];
#pragma unroll
            for (int a = 0; a < 7; ++a) all[a] = max(max(s_mm[0][a], s_mm[1][a]), max(s_mm[2][a], s_mm[3][a]));
            VoxelGridParams g0;
            uint32_t flags, bits0;
            voxel_params_of(all, inv_leaf, g0, flags, bits0);
            s_g = g0; s_bits = bits0;
            if (blockIdx.x == 0) {
                vox->g = g0; vox->flags = flags; vox->probe[1] = bits0;
                if (check_time_order && all[6] != 0u) vox->probe[0] = 1u;
            }
        }
    } else if (threadIdx.x == 0) {
        s_g = vox->g; s_bits = vox->probe[1];
    }
    __syncthreads();
    const VoxelGridParams g = s_g;
    const uint32_t bits = s_bits;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
        const float4 p = pts[i];
        uint32_t key = bits >= 32u ? 0xffffffffu : (1u << bits) - 1u;   // non-finite points sort last and are dropped
        if (isfinite(p.x) && isfinite(p.y) && isfinite(p.z)) {
            const int i0 = (int)(floorf(p.x * g.inv_leaf) - (float)g.min_b[0]);
            const int i1 = (int)(floorf(p.y * g.inv_leaf) - (float)g.min_b[1]);
            const int i2 = (int)(floorf(p.z * g.inv_leaf) - (float)g.min_b[2]);
            key = (uint32_t)(i0 * g.divb_mul[0] + i1 * g.divb_mul[1] + i2 * g.divb_mul[2]);
        }
        keys[i] = (uint64_t)key;
#pragma unroll
        for (int d = 0; d < 4; ++d) atomicAdd(&s_hist[d][(key >> (8 * d)) & 255u], 1u);
    }
    __syncthreads();
    for (int x = threadIdx.x; x < 4 * 256; x += 256) {
        const uint32_t v = (&s_hist[0][0])[x];
        if (v != 0u) atomicAdd(&vox->hist[0][0] + x, v);
    }
}

// lane K of this lane's row of 16, as a DPP operand (row_newbcast:K, gfx90a and later)
template <int K> __device__ __forceinline__ float row_bcast16(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x150 + K, 0xf, 0xf, true));
}
template <int K> __device__ __forceinline__ void centroid_add16(const float4& p, float& sx, float& sy, float& sz) {
    sx += row_bcast16<K>(p.x);
    sy += row_bcast16<K>(p.y);
    sz += row_bcast16<K>(p.z);
    if constexpr (K + 1 < 16) centroid_add16<K + 1>(p, sx, sy, sz);
}

// Sixteen lanes per leaf: the points of the leaf are fetched sixteen at a time (the two dependent loads of every
// point -- order, then the point -- are what a single lane would wait for, one after the other), then added in
// sorted (stable = time) order, one at a time, exactly like CentroidPoint: fp32 sums, divided by the count.
__global__ __launch_bounds__(256) void centroid_kernel(const float4* pts, const uint32_t* order, const uint64_t* sorted_keys,
                                                       const uint32_t* leaf_start, VoxelGridDev* vox, uint32_t n, uint32_t cap_out,
                                                       float4* out, bool compact_keys, VoxelGridDev* zero_next) {
    if (zero_next != nullptr && blockIdx.x == gridDim.x - 1u) {          // (nobody reads that record in this call)
        uint32_t* w = reinterpret_cast<uint32_t*>(zero_next);
        for (uint32_t x = threadIdx.x; x < sizeof(VoxelGridDev) / 4u; x += 256u) w[x] = 0u;
    }
    const uint32_t r = threadIdx.x & 15;
    const uint32_t n_leaves = vox->n_leaves;
    const uint32_t flags = vox->flags;
    // the key of the run of dropped (non-finite) points: all ones in the leaf half, or 2^leaf_bits - 1 in the compact form
    const int key_shift = compact_keys ? 0 : 32;
    const uint64_t drop_key = compact_keys ? (vox->probe[1] >= 32u ? 0xffffffffull : (1ull << vox->probe[1]) - 1ull) : 0xffffffffull;
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        vox->summary[0] = n_leaves;
        vox->summary[1] = (n_leaves > 0 && (sorted_keys[n - 1] >> key_shift) == drop_key) ? 1u : 0u;
        vox->summary[2] = flags;
    }
    if (flags != 0u) return;
    // the grid covers 16 k leaves at a time (it is launched before the leaf count is known to the host)
    for (uint32_t l = (blockIdx.x * 256 + threadIdx.x) >> 4; l < n_leaves; l += gridDim.x * 16u) {
        const uint32_t st = leaf_start[l], en = l + 1 < n_leaves ? leaf_start[l + 1] : n;
        if ((sorted_keys[st] >> key_shift) == drop_key) continue;        // the run of dropped points (always last)
        float sx = 0.0f, sy = 0.0f, sz = 0.0f;
        float4 nxt = make_float4(0.f, 0.f, 0.f, 0.f);
        if (st + r < en) nxt = pts[order[st + r]];
        for (uint32_t j0 = st; j0 < en; j0 += 16) {
            const float4 p = nxt;
            const uint32_t jn = j0 + 16 + r;                            // the next sixteen are in flight while these are added
            if (jn < en) nxt = pts[order[jn]];
            const int cnt = (int)min(16u, en - j0);
            if (cnt == 16) {
                // a full batch: sixteen additions per coordinate, each ONE instruction (v_add_f32 with the DPP row broadcast of
                // lane k as its operand) -- the leaf next to the sensor holds ~2 000 of a sweep's points and its sum, sequential
                // by definition, is the tail of this kernel (same additions, same order: same bits)
                centroid_add16<0>(p, sx, sy, sz);
            } else {
                for (int k = 0; k < cnt; ++k) {
                    sx += __shfl(p.x, k, 16);
                    sy += __shfl(p.y, k, 16);
                    sz += __shfl(p.z, k, 16);
                }
            }
        }
        const float fn = (float)(en - st);
        if (r == 0 && l < cap_out) out[l] = make_float4(sx / fn, sy / fn, sz / fn, 0.0f);
    }
}

// Run heads AND centroids in one launch (time-ordered sweeps, compact keys): a workgroup takes a tile of 256 sorted points, finds the
// leaves that START in it, numbers them through the stamped tile prefix (lio_tile.h) -- out[] is in ascending leaf order -- and gives each
// to sixteen lanes, which follow the run to its end (into later tiles if need be: the end is where the key changes) with the very
// additions of centroid_kernel.  Saves the run-head launch and the round trip of the leaf starts through memory.
__global__ __launch_bounds__(256) void leaf_centroids_kernel(const float4* __restrict__ pts, const uint32_t* __restrict__ order, const uint64_t* __restrict__ sorted_keys,
                                                             VoxelGridDev* vox, uint32_t n, uint32_t cap_out, float4* __restrict__ out, VoxelGridDev* zero_next,
                                                             unsigned long long* agg, uint32_t stamp) {
    __shared__ uint32_t s_w[4];
    __shared__ uint32_t s_head[256];
    const uint32_t tid = threadIdx.x;
    if (zero_next != nullptr && blockIdx.x == gridDim.x - 1u) {          // (nobody reads that record in this call)
        uint32_t* w = reinterpret_cast<uint32_t*>(zero_next);
        for (uint32_t x = tid; x < sizeof(VoxelGridDev) / 4u; x += 256u) w[x] = 0u;
    }
    const uint32_t flags = vox->flags;
    const uint64_t drop_key = vox->probe[1] >= 32u ? 0xffffffffull : (1ull << vox->probe[1]) - 1ull;   // the run of dropped (non-finite) points
    const uint32_t i = blockIdx.x * 256u + tid;
    const uint64_t key = i < n ? sorted_keys[i] : 0ull;
    const bool head = i < n && (i == 0u || sorted_keys[i - 1u] != key);
    uint32_t tot;
    const uint32_t r = tile_rank(head, tot, s_w);
    if (head) s_head[r] = i;
    const uint32_t mine[1] = {tot};
    uint32_t before[1];
    const bool ok = tile_exclusive_prefix<1>(mine, agg, stamp, before);   // (its barriers also publish s_head)
    if (!ok && tid == 0) atomicOr(&vox->summary[3], 1u);
    if (blockIdx.x == gridDim.x - 1u && tid == 0) {
        const uint32_t n_leaves = ok ? before[0] + tot : 0u;
        vox->n_leaves = n_leaves;
        vox->summary[0] = n_leaves;
        vox->summary[1] = (n_leaves > 0u && sorted_keys[n - 1u] == drop_key) ? 1u : 0u;
        vox->summary[2] = flags;
    }
    if (flags != 0u || !ok) return;
    const uint32_t r16 = tid & 15u, lane = tid & 63u;
    for (uint32_t h = tid >> 4; h < tot; h += 16u) {
        const uint32_t st = s_head[h];
        const uint64_t k0 = sorted_keys[st];
        if (k0 == drop_key) continue;                                     // (always the last run)
        float sx = 0.0f, sy = 0.0f, sz = 0.0f;
        uint32_t count = 0u;
        uint32_t j = st + r16;
        bool in = j < n && sorted_keys[j] == k0;
        float4 nxt = make_float4(0.f, 0.f, 0.f, 0.f);
        if (in) nxt = pts[order[j]];
        for (;;) {
            const float4 p = nxt;
            // the points of the run among these sixteen: a prefix (the keys are sorted)
            const int cnt = __popcll((__ballot(in) >> (lane & 48u)) & 0xffffull);
            if (cnt == 16) {                                              // the next sixteen are in flight while these are added
                j += 16u;
                in = j < n && sorted_keys[j] == k0;
                if (in) nxt = pts[order[j]];
                centroid_add16<0>(p, sx, sy, sz);
            } else {
                for (int k = 0; k < cnt; ++k) {
                    sx += __shfl(p.x, k, 16);
                    sy += __shfl(p.y, k, 16);
                    sz += __shfl(p.z, k, 16);
                }
            }
            count += (uint32_t)cnt;
            if (cnt < 16) break;
        }
        const float fn = (float)count;
        const uint32_t l = before[0] + h;
        if (r16 == 0u && l < cap_out) out[l] = make_float4(sx / fn, sy / fn, sz / fn, 0.0f);
    }
}
hipError_t launch_leaf_centroids(const float4* pts, const uint32_t* order, const uint64_t* sorted_keys, VoxelGridDev* vox, uint32_t n, uint32_t cap_out, float4* out,
                                 VoxelGridDev* zero_for_next_call, unsigned long long* agg, uint32_t stamp, hipStream_t s) {
    if (n == 0) return hipSuccess;
    if (n > 262144u) return hipErrorInvalidValue;                         // (1024 tiles: all resident at once)
    hipLaunchKernelGGL(leaf_centroids_kernel, dim3((n + 255u) / 256u), dim3(256), 0, s, pts, order, sorted_keys, vox, n, cap_out, out, zero_for_next_call, agg, stamp);
    return hipGetLastError();
}

hipError_t launch_time_keys(const float4* pts, int n, uint64_t* keys, uint32_t* order, hipStream_t s) {
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(time_keys_kernel, dim3((n + 255) / 256), dim3(256), 0, s, pts, n, keys, order);
    return hipGetLastError();
}
// one lane per host record; records whose x, y, z, pad sit on a 16-byte boundary take one 16-byte read, others three 4-byte reads
template <bool ALIGNED16>
__global__ __launch_bounds__(256) void scan_pull_kernel(const char* __restrict__ src, int stride, int n, float4* __restrict__ dst) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const char* rec = src + (size_t)i * (size_t)stride;
    float4 v;
    if (ALIGNED16) {
        v = *reinterpret_cast<const float4*>(rec);
    } else {
        const float* p = reinterpret_cast<const float*>(rec);
        v.x = p[0]; v.y = p[1]; v.z = p[2];
    }
    v.w = 0.0f;
    dst[i] = v;
}
hipError_t launch_scan_pull(const void* src_device_alias, int stride_bytes, int n, float4* dst, hipStream_t s) {
    if (n <= 0) return hipSuccess;
    const bool aligned = stride_bytes % 16 == 0 && (reinterpret_cast<uintptr_t>(src_device_alias) & 15u) == 0u;
    if (aligned) hipLaunchKernelGGL(scan_pull_kernel<true>, dim3((n + 255) / 256), dim3(256), 0, s, (const char*)src_device_alias, stride_bytes, n, dst);
    else hipLaunchKernelGGL(scan_pull_kernel<false>, dim3((n + 255) / 256), dim3(256), 0, s, (const char*)src_device_alias, stride_bytes, n, dst);
    return hipGetLastError();
}
hipError_t launch_undistort(const float4* raw, int n, const double* poses, int n_poses, const Pose14& end, float4* out,
                            hipStream_t s, uint32_t* parts) {
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(undistort_kernel, dim3((n + 255) / 256), dim3(256), 0, s, raw, n, poses, n_poses, end, out, parts);
    return hipGetLastError();
}
hipError_t launch_minmax_params(const float4* pts, int n, VoxelGridDev* vox, float inv_leaf, bool check_time_order, hipStream_t s) {
    if (n <= 0) return hipSuccess;
    int b = (n + 255) / 256;
    if (b > 64) b = 64;                        // six atomics per workgroup on one cache line: keep them few
    hipLaunchKernelGGL(minmax_kernel, dim3(b), dim3(256), 0, s, pts, n, vox, inv_leaf, check_time_order);
    return hipGetLastError();
}
hipError_t launch_voxel_index_compact(const float4* pts, int n, VoxelGridDev* vox, uint64_t* keys, hipStream_t s, const uint32_t* parts, uint32_t n_parts,
                                      float inv_leaf, bool check_time_order) {
    if (n <= 0) return hipSuccess;
    int b = (n + 1023) / 1024;                 // four points per lane: fewer, fuller histogram flushes
    if (b > 256) b = 256;                      // (64 workgroups, eight points per lane: 8.5 us instead of 7.3)
    hipLaunchKernelGGL(voxel_index_compact_kernel, dim3(b), dim3(256), 0, s, pts, n, vox, keys, parts, n_parts, inv_leaf, check_time_order);
    return hipGetLastError();
}
hipError_t launch_voxel_index(const float4* pts, int n, const VoxelGridDev* vox, bool time_minor, uint64_t* keys, uint32_t* order,
                              hipStream_t s) {
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(voxel_index_kernel, dim3((n + 255) / 256), dim3(256), 0, s, pts, n, vox, time_minor, keys, order);
    return hipGetLastError();
}
hipError_t launch_centroids(const float4* pts, const uint32_t* order, const uint64_t* sorted_keys, const uint32_t* leaf_start,
                            VoxelGridDev* vox, uint32_t n, uint32_t max_leaves, uint32_t cap_out, float4* out, hipStream_t s,
                            bool compact_keys, VoxelGridDev* zero_for_next_call) {
    if (n == 0 || max_leaves == 0) return hipSuccess;
    uint32_t blocks = (uint32_t)(((size_t)max_leaves * 16 + 255) / 256);
    if (blocks > 1024u) blocks = 1024u;
    hipLaunchKernelGGL(centroid_kernel, dim3(blocks), dim3(256), 0, s, pts, order, sorted_keys, leaf_start, vox, n, cap_out, out, compact_keys,
                       zero_for_next_call);
    return hipGetLastError();
}

}  // namespace lio
