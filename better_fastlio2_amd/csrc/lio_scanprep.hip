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

namespace lio {

__device__ __forceinline__ void mat3_mul(const double a[9], const double b[9], double r[9]) {
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) r[i * 3 + j] = (a[i * 3] * b[j] + a[i * 3 + 1] * b[3 + j]) + a[i * 3 + 2] * b[6 + j];
}

__global__ __launch_bounds__(256) void undistort_kernel(const float4* raw, int n, const double* poses, int n_poses,
                                                        Pose14 end, float4* out) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float4 p = raw[i];
    float4 o = p;
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

// ---- voxel grid ------------------------------------------------------------------------------------
__device__ __forceinline__ int float_ordered(float f) {                // monotone float -> int
    const int i = __float_as_int(f);
    return i >= 0 ? i : i ^ 0x7fffffff;
}
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
    keys[i] = (uint64_t)((uint32_t)float_ordered(pts[i].w) ^ 0x80000000u);
    order[i] = (uint32_t)i;
}

// getMinMax3D: six atomics per wavefront on order-preserving integer images of the floats
__global__ __launch_bounds__(256) void minmax_kernel(const float4* pts, int n, int* mm /* min xyz, max xyz */) {
    int lo[3] = {0x7fffffff, 0x7fffffff, 0x7fffffff}, hi[3] = {(int)0x80000000, (int)0x80000000, (int)0x80000000};
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const float4 p = pts[i];
        if (!(isfinite(p.x) && isfinite(p.y) && isfinite(p.z))) continue;
        const int v[3] = {float_ordered(p.x), float_ordered(p.y), float_ordered(p.z)};
#pragma unroll
        for (int a = 0; a < 3; ++a) { lo[a] = min(lo[a], v[a]); hi[a] = max(hi[a], v[a]); }
    }
#pragma unroll
    for (int a = 0; a < 3; ++a) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            lo[a] = min(lo[a], __shfl_xor(lo[a], o, 64));
            hi[a] = max(hi[a], __shfl_xor(hi[a], o, 64));
        }
    }
    if ((threadIdx.x & 63) == 0) {
#pragma unroll
        for (int a = 0; a < 3; ++a) { atomicMin(mm + a, lo[a]); atomicMax(mm + 3 + a, hi[a]); }
    }
}

__global__ __launch_bounds__(256) void voxel_index_kernel(const float4* pts, int n, VoxelGridParams g, uint64_t* keys,
                                                          uint32_t* order) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float4 p = pts[i];
    uint64_t key = ~0ull;                                               // non-finite points sort last and are dropped
    if (isfinite(p.x) && isfinite(p.y) && isfinite(p.z)) {
        // ijk = (int)(floor(p * inverse_leaf) - (float)min_b)          (voxel_grid.hpp, first pass)
        const int i0 = (int)(floorf(p.x * g.inv_leaf) - (float)g.min_b[0]);
        const int i1 = (int)(floorf(p.y * g.inv_leaf) - (float)g.min_b[1]);
        const int i2 = (int)(floorf(p.z * g.inv_leaf) - (float)g.min_b[2]);
        const int idx = i0 * g.divb_mul[0] + i1 * g.divb_mul[1] + i2 * g.divb_mul[2];
        key = (uint64_t)(uint32_t)idx;
    }
    keys[i] = key;
    order[i] = (uint32_t)i;
}

// one lane per leaf: fp32 sums in sorted (stable = time) order, divided by the count (CentroidPoint)
__global__ __launch_bounds__(256) void centroid_kernel(const float4* pts, const uint32_t* order, const uint64_t* leaf_key,
                                                       const uint32_t* leaf_start, const uint32_t* leaf_count, uint32_t n_leaves,
                                                       float4* out) {
    const uint32_t l = blockIdx.x * 256 + threadIdx.x;
    if (l >= n_leaves) return;
    if (leaf_key[l] == ~0ull) return;                                   // the run of dropped points (always last)
    const uint32_t st = leaf_start[l], ct = leaf_count[l];
    float sx = 0.0f, sy = 0.0f, sz = 0.0f;
    for (uint32_t j = 0; j < ct; ++j) {
        const float4 p = pts[order[st + j]];
        sx += p.x; sy += p.y; sz += p.z;
    }
    const float fn = (float)ct;
    out[l] = make_float4(sx / fn, sy / fn, sz / fn, 0.0f);
}

hipError_t launch_time_keys(const float4* pts, int n, uint64_t* keys, uint32_t* order, hipStream_t s) {
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(time_keys_kernel, dim3((n + 255) / 256), dim3(256), 0, s, pts, n, keys, order);
    return hipGetLastError();
}
hipError_t launch_undistort(const float4* raw, int n, const double* poses, int n_poses, const Pose14& end, float4* out,
                            hipStream_t s) {
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(undistort_kernel, dim3((n + 255) / 256), dim3(256), 0, s, raw, n, poses, n_poses, end, out);
    return hipGetLastError();
}
hipError_t launch_minmax(const float4* pts, int n, int* mm6, hipStream_t s) {
    if (n <= 0) return hipSuccess;
    int b = (n + 255) / 256;
    if (b > 1024) b = 1024;
    hipLaunchKernelGGL(minmax_kernel, dim3(b), dim3(256), 0, s, pts, n, mm6);
    return hipGetLastError();
}
hipError_t launch_voxel_index(const float4* pts, int n, const VoxelGridParams& g, uint64_t* keys, uint32_t* order, hipStream_t s) {
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(voxel_index_kernel, dim3((n + 255) / 256), dim3(256), 0, s, pts, n, g, keys, order);
    return hipGetLastError();
}
hipError_t launch_centroids(const float4* pts, const uint32_t* order, const uint64_t* leaf_key, const uint32_t* leaf_start,
                            const uint32_t* leaf_count, uint32_t n_leaves, float4* out, hipStream_t s) {
    if (n_leaves == 0) return hipSuccess;
    hipLaunchKernelGGL(centroid_kernel, dim3((n_leaves + 255) / 256), dim3(256), 0, s, pts, order, leaf_key, leaf_start,
                       leaf_count, n_leaves, out);
    return hipGetLastError();
}
float minmax_decode(int ordered) { return ordered_float(ordered); }

}  // namespace lio
