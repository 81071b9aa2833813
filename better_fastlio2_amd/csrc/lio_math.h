// lio_math.h -- per-point arithmetic of the measurement model, written once for the gfx950 kernels
// (and compilable for the host so that tests/host_math can exercise the very same source on CPU).
//
// All translation units including this header are built with -ffp-contract=off: the reference
// runs on baseline x86-64 (no FMA), and bit-level agreement of the fp32 decisions (neighbour
// gate, plane inlier test, s > 0.9) needs the same un-fused operations in the same order.
// fp32 '/' and sqrtf are correctly rounded under hipcc's default
// -fhip-fp32-correctly-rounded-divide-sqrt.
#pragma once
#include "lio_common.h"

namespace lio {

struct D3 { double x, y, z; };
struct Qd { double x, y, z, w; };            // Eigen::Quaterniond coefficient order

LIO_HD D3 d3_add(D3 a, D3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
LIO_HD D3 d3_cross(D3 a, D3 b) {
    return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}
LIO_HD Qd q_conj(Qd q) { return {-q.x, -q.y, -q.z, q.w}; }

// v' = q v q^-1 in the operation order of Eigen 3.3 QuaternionBase::_transformVector
// (uv = 2 qv x v;  v + w uv + qv x uv), which is what `s.rot * vec` evaluates
// (SO3 derives from Eigen::Quaternion, include/IKFoM_toolkit/mtk/types/SOn.hpp:181).
LIO_HD D3 q_rotate(Qd q, D3 v) {
    D3 qv{q.x, q.y, q.z};
    D3 uv = d3_cross(qv, v);
    uv = d3_add(uv, uv);
    D3 t = d3_cross(qv, uv);
    return {(v.x + q.w * uv.x) + t.x, (v.y + q.w * uv.y) + t.y, (v.z + q.w * uv.z) + t.z};
}

struct PoseQ { D3 pos; Qd rot; Qd offR; D3 offT; };
LIO_HD PoseQ pose_unpack(const Pose14& p) {
    const double* s = p.v;
    return {{s[0], s[1], s[2]}, {s[3], s[4], s[5], s[6]}, {s[7], s[8], s[9], s[10]}, {s[11], s[12], s[13]}};
}

// p_global = s.rot * (s.offset_R_L_I * p_body + s.offset_T_L_I) + s.pos  (laserMapping.cpp:1895),
// fp64, then stored to the fp32 point_world (laserMapping.cpp:1896-1898).
LIO_HD void body_to_world(const PoseQ& s, float bx, float by, float bz, float& wx, float& wy, float& wz) {
    D3 pb{(double)bx, (double)by, (double)bz};
    D3 pi = d3_add(q_rotate(s.offR, pb), s.offT);
    D3 pg = d3_add(q_rotate(s.rot, pi), s.pos);
    wx = (float)pg.x; wy = (float)pg.y; wz = (float)pg.z;
}

// 3x3 rotation matrix of a quaternion in the order of Eigen QuaternionBase::toRotationMatrix
// (needed for `point_be_crossmat * s.offset_R_L_I.conjugate() * C`, laserMapping.cpp:1992).
struct M3d { double m[3][3]; };
LIO_HD M3d q_matrix(Qd q) {
    double tx = 2 * q.x, ty = 2 * q.y, tz = 2 * q.z;
    double twx = tx * q.w, twy = ty * q.w, twz = tz * q.w;
    double txx = tx * q.x, txy = ty * q.x, txz = tz * q.x;
    double tyy = ty * q.y, tyz = tz * q.y, tzz = tz * q.z;
    M3d r;
    r.m[0][0] = 1 - (tyy + tzz); r.m[0][1] = txy - twz;       r.m[0][2] = txz + twy;
    r.m[1][0] = txy + twz;       r.m[1][1] = 1 - (txx + tzz); r.m[1][2] = tyz - twx;
    r.m[2][0] = txz - twy;       r.m[2][1] = tyz + twx;       r.m[2][2] = 1 - (txx + tyy);
    return r;
}
LIO_HD D3 m3_mul_v(const M3d& a, D3 v) {
    return {(a.m[0][0] * v.x + a.m[0][1] * v.y) + a.m[0][2] * v.z,
            (a.m[1][0] * v.x + a.m[1][1] * v.y) + a.m[1][2] * v.z,
            (a.m[2][0] * v.x + a.m[2][1] * v.y) + a.m[2][2] * v.z};
}
// skew(a) * v  (SKEW_SYM_MATRX, include/common_lib.h; product of a 3x3 with a vector)
LIO_HD D3 skew_mul_v(D3 a, D3 v) {
    // rows: [0 -az ay], [az 0 -ax], [-ay ax 0]
    return {(0.0 * v.x + (-a.z) * v.y) + a.y * v.z,
            (a.z * v.x + 0.0 * v.y) + (-a.x) * v.z,
            ((-a.y) * v.x + a.x * v.y) + 0.0 * v.z};
}

// One row of h_x and its h entry (laserMapping.cpp:1970-2002).
//   row = [n, A, B, C],  A = [p_imu]x C,  C = R_wi^T n,  B = [p_body]x R_il^T C (if extrinsic_est_en)
template <bool EXT>
LIO_HD void jacobian_row(const PoseQ& s, float bx, float by, float bz, float nx, float ny, float nz,
                         float pd2, double* row /* 12 (EXT) or 6 */, double& h) {
    D3 pbe{(double)bx, (double)by, (double)bz};
    D3 pthis = d3_add(q_rotate(s.offR, pbe), s.offT);          // :1976
    D3 nv{(double)nx, (double)ny, (double)nz};
    D3 C = q_rotate(q_conj(s.rot), nv);                         // :1985
    D3 A = skew_mul_v(pthis, C);                                // :1986
    row[0] = nv.x; row[1] = nv.y; row[2] = nv.z;
    row[3] = A.x;  row[4] = A.y;  row[5] = A.z;
    if (EXT) {
        // (point_be_crossmat * R(offR^-1)) * C : 3x3 product first, then the vector   :1992
        M3d R = q_matrix(q_conj(s.offR));
        M3d SR;
        const double S[3][3] = {{0.0, -pbe.z, pbe.y}, {pbe.z, 0.0, -pbe.x}, {-pbe.y, pbe.x, 0.0}};
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j)
                SR.m[i][j] = (S[i][0] * R.m[0][j] + S[i][1] * R.m[1][j]) + S[i][2] * R.m[2][j];
        D3 B = m3_mul_v(SR, C);
        row[6] = B.x; row[7] = B.y; row[8] = B.z;
        row[9] = C.x; row[10] = C.y; row[11] = C.z;
    }
    h = -(double)pd2;                                           // :2001
}

// ---------------------------------------------------------------------------------------------
// esti_plane<float> (include/common_lib.h:506-536): solve A n = -1 for the 5x3 matrix of neighbour
// coordinates with Eigen's ColPivHouseholderQR (Eigen is a third-party dependency of the reference,
// un-vendored; Eigen 3.3.7 by distro -- SURVEY.md section 8c), normalise, inlier test.
// fp32 throughout, fully unrolled for the fixed 5x3 shape:
//   column norms -> for k = 0..2: pivot on the largest remaining (down-dated) column norm, swap,
//   Householder reflector on column k below the diagonal, apply it to the trailing columns,
//   down-date the trailing column norms (LAPACK working note 176 rule)
//   -> c = Q^T b -> back-substitute the leading nonzero_pivots x nonzero_pivots triangle -> un-permute.
// ---------------------------------------------------------------------------------------------
LIO_HD bool plane_fit5(const float (&px)[5], const float (&py)[5], const float (&pz)[5], float threshold,
                       float& a, float& b, float& c, float& d) {
    float A[5][3];
#pragma unroll
    for (int j = 0; j < 5; ++j) { A[j][0] = px[j]; A[j][1] = py[j]; A[j][2] = pz[j]; }
    float rhs[5] = {-1.0f, -1.0f, -1.0f, -1.0f, -1.0f};

    const float eps = 1.1920929e-07f;                         // NumTraits<float>::epsilon()
    const float fmin_pos = 1.17549435e-38f;                   // (std::numeric_limits<float>::min)()
    const float downdate_thr = __builtin_sqrtf(eps);

    float nrm_upd[3], nrm_dir[3];
#pragma unroll
    for (int cidx = 0; cidx < 3; ++cidx) {
        float s = 0.0f;
#pragma unroll
        for (int r = 0; r < 5; ++r) s += A[r][cidx] * A[r][cidx];
        nrm_dir[cidx] = __builtin_sqrtf(s);
        nrm_upd[cidx] = nrm_dir[cidx];
    }
    float nmax = nrm_upd[0] > nrm_upd[1] ? nrm_upd[0] : nrm_upd[1];
    nmax = nmax > nrm_upd[2] ? nmax : nrm_upd[2];
    const float t0 = nmax * eps;
    const float thr_helper = (t0 * t0) / 5.0f;

    float tau[3] = {0.0f, 0.0f, 0.0f};
    int transp[3] = {0, 1, 2};
    int nonzero = 3;

#pragma unroll
    for (int k = 0; k < 3; ++k) {
        // pivot: first maximum of the remaining updated norms (all register indices static)
        int big = k;
        float bigv = nrm_upd[k];
#pragma unroll
        for (int j = k + 1; j < 3; ++j) if (nrm_upd[j] > bigv) { bigv = nrm_upd[j]; big = j; }
        const float bigsq = bigv * bigv;
        if (nonzero == 3 && bigsq < thr_helper * (float)(5 - k)) nonzero = k;
        transp[k] = big;
#pragma unroll
        for (int j = k + 1; j < 3; ++j) {
            if (big == j) {
#pragma unroll
                for (int r = 0; r < 5; ++r) { float t = A[r][k]; A[r][k] = A[r][j]; A[r][j] = t; }
                float t = nrm_upd[k]; nrm_upd[k] = nrm_upd[j]; nrm_upd[j] = t;
                t = nrm_dir[k]; nrm_dir[k] = nrm_dir[j]; nrm_dir[j] = t;
            }
        }
        // reflector for A[k:5, k]
        float tail = 0.0f;
#pragma unroll
        for (int r = k + 1; r < 5; ++r) tail += A[r][k] * A[r][k];
        const float c0 = A[k][k];
        float beta, tk;
        if (tail <= fmin_pos) {
            tk = 0.0f; beta = c0;
#pragma unroll
            for (int r = k + 1; r < 5; ++r) A[r][k] = 0.0f;
        } else {
            beta = __builtin_sqrtf(c0 * c0 + tail);
            if (c0 >= 0.0f) beta = -beta;
            const float den = c0 - beta;
#pragma unroll
            for (int r = k + 1; r < 5; ++r) A[r][k] = A[r][k] / den;
            tk = (beta - c0) / beta;
        }
        tau[k] = tk;
        A[k][k] = beta;
        // apply H_k = I - tau v v^T (v = [1; essential]) to the trailing columns
        if (tk != 0.0f) {
#pragma unroll
            for (int j = k + 1; j < 3; ++j) {
                float tmp = 0.0f;
#pragma unroll
                for (int r = k + 1; r < 5; ++r) tmp += A[r][k] * A[r][j];
                tmp += A[k][j];
                A[k][j] -= tk * tmp;
#pragma unroll
                for (int r = k + 1; r < 5; ++r) A[r][j] -= (tk * A[r][k]) * tmp;
            }
        }
        // down-date the trailing column norms
#pragma unroll
        for (int j = k + 1; j < 3; ++j) {
            if (nrm_upd[j] != 0.0f) {
                float t = __builtin_fabsf(A[k][j]) / nrm_upd[j];
                t = (1.0f + t) * (1.0f - t);
                t = t < 0.0f ? 0.0f : t;
                const float ratio = nrm_upd[j] / nrm_dir[j];
                const float t2 = t * (ratio * ratio);
                if (t2 <= downdate_thr) {
                    float s = 0.0f;
#pragma unroll
                    for (int r = k + 1; r < 5; ++r) s += A[r][j] * A[r][j];
                    nrm_dir[j] = __builtin_sqrtf(s);
                    nrm_upd[j] = nrm_dir[j];
                } else {
                    nrm_upd[j] *= __builtin_sqrtf(t);
                }
            }
        }
    }
    // permutation = product of the transpositions, applied left to right
    int perm[3] = {0, 1, 2};
#pragma unroll
    for (int k = 0; k < 3; ++k) {
#pragma unroll
        for (int j = k + 1; j < 3; ++j)
            if (transp[k] == j) { int t = perm[k]; perm[k] = perm[j]; perm[j] = t; }
    }

    float nv[3] = {0.0f, 0.0f, 0.0f};
    if (nonzero > 0) {
        // rhs <- H_{nz-1} ... H_0 rhs
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            if (k < nonzero && tau[k] != 0.0f) {
                float tmp = 0.0f;
#pragma unroll
                for (int r = k + 1; r < 5; ++r) tmp += A[r][k] * rhs[r];
                tmp += rhs[k];
                rhs[k] -= tau[k] * tmp;
#pragma unroll
                for (int r = k + 1; r < 5; ++r) rhs[r] -= (tau[k] * A[r][k]) * tmp;
            }
        }
        // column-oriented back substitution on the leading nonzero x nonzero upper triangle
#pragma unroll
        for (int i = 2; i >= 0; --i) {
            if (i < nonzero) {
                rhs[i] = rhs[i] / A[i][i];
#pragma unroll
                for (int r = 0; r < 3; ++r) if (r < i) rhs[r] -= rhs[i] * A[r][i];
            }
        }
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const float v = i < nonzero ? rhs[i] : 0.0f;
            // nv[perm[i]] = v without dynamic register indexing
            if (perm[i] == 0) nv[0] = v; else if (perm[i] == 1) nv[1] = v; else nv[2] = v;
        }
    }
    const float n = __builtin_sqrtf((nv[0] * nv[0] + nv[1] * nv[1]) + nv[2] * nv[2]);   // common_lib.h:522
    a = nv[0] / n; b = nv[1] / n; c = nv[2] / n;
    d = (float)(1.0 / (double)n);                                                        // `1.0 / n` :526
    bool ok = true;
#pragma unroll
    for (int j = 0; j < 5; ++j) {
        const float r = ((a * px[j] + b * py[j]) + c * pz[j]) + d;                       // :530
        if (__builtin_fabsf(r) > threshold) ok = false;
    }
    return ok;
}

// neighbour gate, plane test, residual and weight test of one scan point
// (laserMapping.cpp:1911-1938).  Returns the new point_selected_surf[i].
LIO_HD bool residual_from_plane(float a, float b, float c, float d, float wx, float wy, float wz, float bx, float by,
                                float bz, float& pd2) {
    pd2 = ((a * wx + b * wy) + c * wz) + d;                                 // :1925
    // p_body.norm() is fp64 (V3D), sqrt of it, 0.9 * fabs(pd2) promoted to double, result to float
    const double bn = __builtin_sqrt(((double)bx * (double)bx + (double)by * (double)by) + (double)bz * (double)bz);
    const float s = (float)(1 - 0.9 * __builtin_fabs((double)pd2) / __builtin_sqrt(bn));   // :1927
    return s > 0.9;                                                         // :1930 (float vs double 0.9)
}
LIO_HD bool point_residual(const float (&px)[5], const float (&py)[5], const float (&pz)[5],
                           float wx, float wy, float wz, float bx, float by, float bz,
                           float& nx, float& ny, float& nz, float& pd2, float* plane_d = nullptr) {
    float a, b, c, d;
    if (!plane_fit5(px, py, pz, 0.1f, a, b, c, d)) return false;           // :1922
    if (plane_d != nullptr) *plane_d = d;
    nx = a; ny = b; nz = c;                                                 // (read by the caller only when selected)
    return residual_from_plane(a, b, c, d, wx, wy, wz, bx, by, bz, pd2);
}

}  // namespace lio
