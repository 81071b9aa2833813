// lio_esekf.cpp -- host-side iterated ESKF around the GPU measurement path (see lio_esekf.hpp).
// Plain C++17, fp64.  Each block cites the lines of the reference it mirrors (paths relative to the
// reference repository; "esekfom.hpp" = include/IKFoM_toolkit/esekfom/esekfom.hpp, "mtk/" =
// include/IKFoM_toolkit/mtk/).
#include "lio_esekf.hpp"

#include <chrono>
#include <cmath>
#include <cstring>
#include <limits>

namespace lio {
namespace {

// ------------------------------------------------------------------------------------------
// tiny fixed-size dense algebra (row-major)
// ------------------------------------------------------------------------------------------
template <int R, int C>
struct Mat {
    double a[R * C];
    double& operator()(int r, int c) { return a[r * C + c]; }
    double operator()(int r, int c) const { return a[r * C + c]; }
    static Mat zero() { Mat m; for (double& v : m.a) v = 0.0; return m; }
    static Mat identity() { Mat m = zero(); for (int i = 0; i < (R < C ? R : C); ++i) m(i, i) = 1.0; return m; }
};
template <int R, int K, int C>
Mat<R, C> operator*(const Mat<R, K>& x, const Mat<K, C>& y) {
    Mat<R, C> z;
    for (int r = 0; r < R; ++r)
        for (int c = 0; c < C; ++c) {
            double s = 0.0;
            for (int k = 0; k < K; ++k) s += x(r, k) * y(k, c);
            z(r, c) = s;
        }
    return z;
}
template <int R, int C>
Mat<C, R> transpose(const Mat<R, C>& x) {
    Mat<C, R> t;
    for (int r = 0; r < R; ++r) for (int c = 0; c < C; ++c) t(c, r) = x(r, c);
    return t;
}
typedef Mat<3, 3> M3;
typedef Mat<3, 1> V3;

V3 v3(double x, double y, double z) { V3 v; v.a[0] = x; v.a[1] = y; v.a[2] = z; return v; }
V3 v3(const Vec3& p) { return v3(p.x, p.y, p.z); }
M3 hat(const V3& v) {                                        // mtk/src/mtkmath.hpp:176-183
    M3 h = M3::zero();
    h(0, 1) = -v.a[2]; h(0, 2) = v.a[1];
    h(1, 0) = v.a[2];  h(1, 2) = -v.a[0];
    h(2, 0) = -v.a[1]; h(2, 1) = v.a[0];
    return h;
}
double dot3(const V3& x, const V3& y) { return x.a[0] * y.a[0] + x.a[1] * y.a[1] + x.a[2] * y.a[2]; }
double norm3(const V3& x) { return std::sqrt(dot3(x, x)); }

// Solve A X = [e_0 .. e_{NRHS-1}] (the first NRHS columns of A^-1) for a fixed-size N x N matrix by
// Gauss-Jordan elimination with partial (row) pivoting on a stack-resident augmented matrix.  The
// reference calls Eigen's inverse() (PartialPivLU for sizes > 4); same pivoting rule, same result
// up to rounding.  Fixed trip counts let the compiler vectorise the row updates; the 23 x 23 case
// is what every pass of the iterated update pays twice.
template <int N, int NRHS>
static inline __attribute__((always_inline)) bool solve_identity_cols_impl(const double* A, double* X /* N x NRHS, row-major */) {
    constexpr int W = N + NRHS;
    double m[N][W];
    for (int r = 0; r < N; ++r) {
        for (int c = 0; c < N; ++c) m[r][c] = A[r * N + c];
        for (int c = 0; c < NRHS; ++c) m[r][N + c] = r == c ? 1.0 : 0.0;
    }
    for (int k = 0; k < N; ++k) {
        int piv = k;
        double big = std::fabs(m[k][k]);
        for (int r = k + 1; r < N; ++r) {
            const double v = std::fabs(m[r][k]);
            if (v > big) { big = v; piv = r; }
        }
        if (big == 0.0) return false;
        if (piv != k)
            for (int c = 0; c < W; ++c) { const double t = m[k][c]; m[k][c] = m[piv][c]; m[piv][c] = t; }
        const double inv = 1.0 / m[k][k];
        for (int c = 0; c < W; ++c) m[k][c] *= inv;
        for (int r = 0; r < N; ++r) {
            if (r == k) continue;
            const double f = m[r][k];
            if (f == 0.0) continue;
            for (int c = 0; c < W; ++c) m[r][c] -= f * m[k][c];
        }
    }
    for (int r = 0; r < N; ++r)
        for (int c = 0; c < NRHS; ++c) X[r * NRHS + c] = m[r][N + c];
    return true;
}

// the two instances a pass needs, multi-versioned so that an AVX2 host runs the wide row updates
#if defined(__x86_64__) && !defined(__HIP_DEVICE_COMPILE__)
#define LIO_HOST_CLONES __attribute__((target_clones("avx2", "default")))
#else
#define LIO_HOST_CLONES
#endif
LIO_HOST_CLONES bool invert23(const double* A, double* Ainv) { return solve_identity_cols_impl<DOF, DOF>(A, Ainv); }
LIO_HOST_CLONES bool invert23_first12(const double* A, double* X) { return solve_identity_cols_impl<DOF, 12>(A, X); }

// general (run-time size) inverse, same method: only the m < 23 dense-gain branch uses it
bool invert(const double* A, double* Ainv, int n) {
    const int w = 2 * n;
    std::vector<double> m((size_t)n * w, 0.0);
    for (int r = 0; r < n; ++r) {
        for (int c = 0; c < n; ++c) m[(size_t)r * w + c] = A[(size_t)r * n + c];
        m[(size_t)r * w + n + r] = 1.0;
    }
    for (int k = 0; k < n; ++k) {
        int piv = k;
        double big = std::fabs(m[(size_t)k * w + k]);
        for (int r = k + 1; r < n; ++r) {
            const double v = std::fabs(m[(size_t)r * w + k]);
            if (v > big) { big = v; piv = r; }
        }
        if (big == 0.0) return false;
        if (piv != k) for (int c = 0; c < w; ++c) std::swap(m[(size_t)k * w + c], m[(size_t)piv * w + c]);
        const double inv = 1.0 / m[(size_t)k * w + k];
        for (int c = 0; c < w; ++c) m[(size_t)k * w + c] *= inv;
        for (int r = 0; r < n; ++r) {
            if (r == k) continue;
            const double f = m[(size_t)r * w + k];
            if (f == 0.0) continue;
            for (int c = 0; c < w; ++c) m[(size_t)r * w + c] -= f * m[(size_t)k * w + c];
        }
    }
    for (int r = 0; r < n; ++r)
        for (int c = 0; c < n; ++c) Ainv[(size_t)r * n + c] = m[(size_t)r * w + n + c];
    return true;
}

// ------------------------------------------------------------------------------------------
// MTK pieces
// ------------------------------------------------------------------------------------------
const double kTol = 1e-11;                                   // MTK::tolerance<double>, mtkmath.hpp:122
const double kGravLen = 98090.0 / 10000.0;                   // S2<double, 98090, 10000, 1>, use-ikfom.hpp:8

// cos(sqrt(x2)) and sin(sqrt(x2))/sqrt(x2)                   mtk/src/mtkmath.hpp:142-174
void cos_sinc_sqrt(double x2, double& c, double& sc) {
    static const double b0 = std::numeric_limits<double>::epsilon();
    static const double b2 = std::sqrt(b0);
    static const double bn = std::sqrt(b2);
    if (x2 >= bn) {
        const double x = std::sqrt(x2);
        c = std::cos(x);
        sc = std::sin(x) / x;
        return;
    }
    static const double inv[] = {1 / 3., 1 / 4., 1 / 5., 1 / 6., 1 / 7., 1 / 8., 1 / 9.};
    double cosi = 1., sinc = 1.;
    double term = -1 / 2. * x2;
    for (int i = 0; i < 3; ++i) {
        cosi += term;
        term *= inv[2 * i];
        sinc += term;
        term *= -inv[2 * i + 1] * x2;
    }
    c = cosi; sc = sinc;
}
// MTK::exp (mtkmath.hpp:250-256) into a quaternion: vec = sinc * scale * v, w = cos
Quat exp_quat(const V3& v, double scale) {
    double c, sc;
    cos_sinc_sqrt(scale * scale * dot3(v, v), c, sc);
    const double mult = sc * scale;
    Quat q;
    q.x = mult * v.a[0]; q.y = mult * v.a[1]; q.z = mult * v.a[2]; q.w = c;
    return q;
}
Quat qmul(const Quat& a, const Quat& b) {                    // Eigen quaternion product
    Quat r;
    r.w = a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z;
    r.x = a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y;
    r.y = a.w * b.y + a.y * b.w + a.z * b.x - a.x * b.z;
    r.z = a.w * b.z + a.z * b.w + a.x * b.y - a.y * b.x;
    return r;
}
Quat qconj(const Quat& q) { Quat r; r.x = -q.x; r.y = -q.y; r.z = -q.z; r.w = q.w; return r; }
M3 qmatrix(const Quat& q) {                                  // Eigen toRotationMatrix
    const double tx = 2 * q.x, ty = 2 * q.y, tz = 2 * q.z;
    const double twx = tx * q.w, twy = ty * q.w, twz = tz * q.w;
    const double txx = tx * q.x, txy = ty * q.x, txz = tz * q.x;
    const double tyy = ty * q.y, tyz = tz * q.y, tzz = tz * q.z;
    M3 r;
    r(0, 0) = 1 - (tyy + tzz); r(0, 1) = txy - twz;       r(0, 2) = txz + twy;
    r(1, 0) = txy + twz;       r(1, 1) = 1 - (txx + tzz); r(1, 2) = tyz - twx;
    r(2, 0) = txz - twy;       r(2, 1) = tyz + twx;       r(2, 2) = 1 - (txx + tyy);
    return r;
}
// SO3::log (mtk/types/SOn.hpp:293-297 -> mtkmath.hpp:268-288, scale 2, plus_minus_periodicity)
V3 so3_log(const Quat& q) {
    V3 v = v3(q.x, q.y, q.z);
    double nv = norm3(v);
    if (nv < kTol) nv = kTol;
    const double s = 2.0 / nv * std::atan(nv / q.w);
    return v3(s * v.a[0], s * v.a[1], s * v.a[2]);
}
// A_matrix (mtkmath.hpp:236-247)
M3 A_matrix(const V3& v) {
    const double sq = v.a[0] * v.a[0] + v.a[1] * v.a[1] + v.a[2] * v.a[2];
    const double n = std::sqrt(sq);
    M3 res = M3::identity();
    if (n < kTol) return res;
    const M3 h = hat(v);
    const M3 hh = h * h;
    const double ca = (1 - std::cos(n)) / sq, cb = (1 - std::sin(n) / n) / sq;
    for (int i = 0; i < 9; ++i) res.a[i] += ca * h.a[i] + cb * hh.a[i];
    return res;
}

// S2 with S2_typ == 1 (mtk/types/S2.hpp:213-231)
Mat<3, 2> S2_Bx(const V3& vec) {
    Mat<3, 2> B = Mat<3, 2>::zero();
    const double L = kGravLen;
    if (vec.a[0] + L > kTol) {
        const double den = L + vec.a[0];
        B(0, 0) = -vec.a[1];                       B(0, 1) = -vec.a[2];
        B(1, 0) = L - vec.a[1] * vec.a[1] / den;   B(1, 1) = -vec.a[2] * vec.a[1] / den;
        B(2, 0) = -vec.a[2] * vec.a[1] / den;      B(2, 1) = L - vec.a[2] * vec.a[2] / den;
        for (double& v : B.a) v /= L;
    } else {
        B(1, 1) = -1;
        B(2, 0) = 1;
    }
    return B;
}
// S2::boxplus (S2.hpp:136-142)
V3 S2_boxplus(const V3& vec, double d0, double d1) {
    Mat<2, 1> d; d.a[0] = d0; d.a[1] = d1;
    const V3 Bu = S2_Bx(vec) * d;
    return qmatrix(exp_quat(Bu, 0.5)) * vec;                 // scalar(scale/2), scale = 1
}
// S2::boxminus (S2.hpp:144-167): res = vec [-] other
void S2_boxminus(const V3& vec, const V3& other, double res[2]) {
    const double v_sin = norm3(hat(vec) * other);
    const double v_cos = dot3(vec, other);
    const double theta = std::atan2(v_sin, v_cos);
    if (v_sin < kTol) {
        if (std::fabs(theta) > kTol) { res[0] = 3.1415926; res[1] = 0; }
        else { res[0] = 0; res[1] = 0; }
        return;
    }
    const Mat<2, 1> r = transpose(S2_Bx(other)) * (hat(other) * vec);
    res[0] = theta / v_sin * r.a[0];
    res[1] = theta / v_sin * r.a[1];
}
// S2_Nx_yy (S2.hpp:259-264)
Mat<2, 3> S2_Nx_yy(const V3& vec) {
    Mat<2, 3> N = transpose(S2_Bx(vec)) * hat(vec);
    for (double& v : N.a) v = 1 / kGravLen / kGravLen * v;
    return N;
}
// S2_Mx (S2.hpp:266-280); scalar(1/2) there is an integer division, i.e. exp(Bu, 0) = identity
Mat<3, 2> S2_Mx(const V3& vec, double d0, double d1) {
    const Mat<3, 2> Bx = S2_Bx(vec);
    Mat<3, 2> res;
    if (std::sqrt(d0 * d0 + d1 * d1) < kTol) {
        res = hat(vec) * Bx;
    } else {
        Mat<2, 1> d; d.a[0] = d0; d.a[1] = d1;
        const V3 Bu = Bx * d;
        const M3 Rexp = qmatrix(exp_quat(Bu, 0.0));
        res = Rexp * hat(vec) * transpose(A_matrix(Bu)) * Bx;
    }
    for (double& v : res.a) v = -v;
    return res;
}

// P rows [idx, idx+k) <- J * rows ; then P cols [idx, idx+k) <- cols * J^T   (esekfom.hpp:1672-1678)
template <int k>
void rows_times(double* P, int idx, const Mat<k, k>& J) {
    for (int c = 0; c < DOF; ++c) {
        double t[k];
        for (int i = 0; i < k; ++i) {
            double s = 0.0;
            for (int j = 0; j < k; ++j) s += J(i, j) * P[(idx + j) * DOF + c];
            t[i] = s;
        }
        for (int i = 0; i < k; ++i) P[(idx + i) * DOF + c] = t[i];
    }
}
template <int k>
void cols_times_T(double* P, int idx, const Mat<k, k>& J) {
    for (int r = 0; r < DOF; ++r) {
        double t[k];
        for (int i = 0; i < k; ++i) {
            double s = 0.0;
            for (int j = 0; j < k; ++j) s += P[r * DOF + idx + j] * J(i, j);
            t[i] = s;
        }
        for (int i = 0; i < k; ++i) P[r * DOF + idx + i] = t[i];
    }
}

const int kSO3Index[2] = {3, 6};                             // x_.SO3_state (build_manifold.hpp:204-212)
const int kS2Index = 21;                                     // x_.S2_state

double now_seconds() {
    return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

}  // namespace

// h_x^T h_x and h_x^T h (esekfom.hpp:1790,1810) from rows the measurement callback returned: h_x is m x 12 COLUMN-major
// (Eigen's default, leading dimension m), h has m entries.  Blocks of 128 rows (13 column slices of 1 KB: L1-resident),
// per block and column pair two 4-lane partial sums -- a plain `t += a[i] * b[i]` is one dependent add per row (strict fp
// order forbids the compiler to split it): 8 ms for 90 k rows, against ~0.5 ms here.  The order of the sums is fixed.
typedef double hth_v4 __attribute__((vector_size(32), aligned(8)));
// column A against columns A .. 12 over `len` rows: one 4-lane accumulator per partner, all of them in registers (the
// trip counts are compile-time constants), column A's lanes loaded once per four rows
template <int A>
static inline __attribute__((always_inline)) void hth_column_block(const double* const col[13], size_t len, double (&acc)[12][13]) {
    constexpr int NB = 13 - A;
    hth_v4 s[NB];
#pragma GCC unroll 13
    for (int k = 0; k < NB; ++k) s[k] = hth_v4{0.0, 0.0, 0.0, 0.0};
    const size_t len4 = len & ~(size_t)3;
    for (size_t i = 0; i < len4; i += 4) {
        const hth_v4 va = *reinterpret_cast<const hth_v4*>(col[A] + i);
#pragma GCC unroll 13
        for (int k = 0; k < NB; ++k) s[k] += va * *reinterpret_cast<const hth_v4*>(col[A + k] + i);
    }
#pragma GCC unroll 13
    for (int k = 0; k < NB; ++k) {
        double t = (s[k][0] + s[k][1]) + (s[k][2] + s[k][3]);
        for (size_t i = len4; i < len; ++i) t += col[A][i] * col[A + k][i];
        acc[A][A + k] += t;
    }
}
LIO_HOST_CLONES void hth_from_rows_colmajor(const double* hx, const double* h, size_t m, double HTH[144], double HTh[12]) {
    double acc[12][13];
    for (int a = 0; a < 12; ++a)
        for (int b = 0; b < 13; ++b) acc[a][b] = 0.0;
    constexpr size_t BLK = 128;
    for (size_t i0 = 0; i0 < m; i0 += BLK) {
        const size_t len = m - i0 < BLK ? m - i0 : BLK;
        const double* col[13];
        for (int b = 0; b < 12; ++b) col[b] = hx + (size_t)b * m + i0;
        col[12] = h + i0;
        hth_column_block<0>(col, len, acc); hth_column_block<1>(col, len, acc); hth_column_block<2>(col, len, acc);
        hth_column_block<3>(col, len, acc); hth_column_block<4>(col, len, acc); hth_column_block<5>(col, len, acc);
        hth_column_block<6>(col, len, acc); hth_column_block<7>(col, len, acc); hth_column_block<8>(col, len, acc);
        hth_column_block<9>(col, len, acc); hth_column_block<10>(col, len, acc); hth_column_block<11>(col, len, acc);
    }
    for (int a = 0; a < 12; ++a) {
        for (int b = a; b < 12; ++b) { HTH[a * 12 + b] = acc[a][b]; HTH[b * 12 + a] = acc[a][b]; }
        HTh[a] = acc[a][12];
    }
}

// ------------------------------------------------------------------------------------------
// quaternion helpers shared with the IMU front end (lio_imu.cpp)
// ------------------------------------------------------------------------------------------
Vec3 quat_rotate(const Quat& q, const Vec3& v) {             // Eigen 3.3 QuaternionBase::_transformVector
    const double ux = 2.0 * (q.y * v.z - q.z * v.y), uy = 2.0 * (q.z * v.x - q.x * v.z), uz = 2.0 * (q.x * v.y - q.y * v.x);
    Vec3 r;
    r.x = v.x + q.w * ux + (q.y * uz - q.z * uy);
    r.y = v.y + q.w * uy + (q.z * ux - q.x * uz);
    r.z = v.z + q.w * uz + (q.x * uy - q.y * ux);
    return r;
}
void quat_to_matrix(const Quat& q, double R[9]) {
    const M3 m = qmatrix(q);
    for (int i = 0; i < 9; ++i) R[i] = m.a[i];
}
Quat quat_from_matrix(const double R[9]) {                   // Eigen 3.3 quaternionbase_assign_impl<Matrix3>
    Quat q;
    double t = R[0] + R[4] + R[8];
    if (t > 0.0) {
        t = std::sqrt(t + 1.0);
        q.w = 0.5 * t;
        t = 0.5 / t;
        q.x = (R[7] - R[5]) * t;
        q.y = (R[2] - R[6]) * t;
        q.z = (R[3] - R[1]) * t;
    } else {
        int i = 0;
        if (R[4] > R[0]) i = 1;
        if (R[8] > R[i * 4]) i = 2;
        const int j = (i + 1) % 3, k = (j + 1) % 3;
        t = std::sqrt(R[i * 4] - R[j * 4] - R[k * 4] + 1.0);
        double v[3];
        v[i] = 0.5 * t;
        t = 0.5 / t;
        q.w = (R[k * 3 + j] - R[j * 3 + k]) * t;
        v[j] = (R[j * 3 + i] + R[i * 3 + j]) * t;
        v[k] = (R[k * 3 + i] + R[i * 3 + k]) * t;
        q.x = v[0]; q.y = v[1]; q.z = v[2];
    }
    return q;
}
Vec3 s2_from_vector(const Vec3& v) {                         // S2(const vect_type&), mtk/types/S2.hpp:124-127
    const double n = std::sqrt(v.x * v.x + v.y * v.y + v.z * v.z);
    Vec3 r{v.x / n, v.y / n, v.z / n};
    r.x *= kGravLen; r.y *= kGravLen; r.z *= kGravLen;
    return r;
}

// process_noise_cov (use-ikfom.hpp:44-52): diag(ng 1e-4, na 1e-4, nbg 1e-5, nba 1e-5), 12 x 12 row-major
void process_noise_cov(double Q[12 * 12]) {
    for (int i = 0; i < 144; ++i) Q[i] = 0.0;
    for (int i = 0; i < 3; ++i) {
        Q[i * 13] = 0.0001; Q[(3 + i) * 13] = 0.0001; Q[(6 + i) * 13] = 0.00001; Q[(9 + i) * 13] = 0.00001;
    }
}

// ------------------------------------------------------------------------------------------
// process model (use-ikfom.hpp:56-99) and esekf::predict (esekfom.hpp:280-383)
// ------------------------------------------------------------------------------------------
// Flattened-state rows (DIM = 24): pos 0, rot 3, offset_R_L_I 6, offset_T_L_I 9, vel 12, bg 15, ba 18, grav 21.
void get_f(const state_ikfom& s, const input_ikfom& in, double f[DIM]) {          // use-ikfom.hpp:56-69
    for (int i = 0; i < DIM; ++i) f[i] = 0.0;
    const Vec3 omega{in.gyro[0] - s.bg.x, in.gyro[1] - s.bg.y, in.gyro[2] - s.bg.z};
    const Vec3 a_inertial = quat_rotate(s.rot, Vec3{in.acc[0] - s.ba.x, in.acc[1] - s.ba.y, in.acc[2] - s.ba.z});
    f[0] = s.vel.x; f[1] = s.vel.y; f[2] = s.vel.z;
    f[3] = omega.x; f[4] = omega.y; f[5] = omega.z;
    f[12] = a_inertial.x + s.grav.x; f[13] = a_inertial.y + s.grav.y; f[14] = a_inertial.z + s.grav.z;
}
void df_dx(const state_ikfom& s, const input_ikfom& in, double F[DIM * DOF]) {    // use-ikfom.hpp:71-88
    for (int i = 0; i < DIM * DOF; ++i) F[i] = 0.0;
    for (int i = 0; i < 3; ++i) F[i * DOF + 12 + i] = 1.0;
    const V3 acc_ = v3(in.acc[0] - s.ba.x, in.acc[1] - s.ba.y, in.acc[2] - s.ba.z);
    const M3 R = qmatrix(s.rot);
    const M3 RA = R * hat(acc_);
    const Mat<3, 2> G = S2_Mx(v3(s.grav), 0.0, 0.0);
    for (int i = 0; i < 3; ++i) {
        for (int j = 0; j < 3; ++j) {
            F[(12 + i) * DOF + 3 + j] = -RA(i, j);
            F[(12 + i) * DOF + 18 + j] = -R(i, j);
        }
        F[(12 + i) * DOF + 21] = G(i, 0);
        F[(12 + i) * DOF + 22] = G(i, 1);
        F[(3 + i) * DOF + 15 + i] = -1.0;
    }
}
void df_dw(const state_ikfom& s, const input_ikfom&, double W[DIM * 12]) {        // use-ikfom.hpp:91-99
    for (int i = 0; i < DIM * 12; ++i) W[i] = 0.0;
    const M3 R = qmatrix(s.rot);
    for (int i = 0; i < 3; ++i) {
        for (int j = 0; j < 3; ++j) W[(12 + i) * 12 + 3 + j] = -R(i, j);
        W[(3 + i) * 12 + i] = -1.0;
        W[(15 + i) * 12 + 6 + i] = 1.0;
        W[(18 + i) * 12 + 9 + i] = 1.0;
    }
}

// P <- F P F^T + W Q W^T for the sparse F (<= 9 entries per row) and W (<= 3) of predict: the products visit the non-zero
// entries only, in ascending index order -- the same sums as the dense loops, the skipped terms being exact zeros.  Both
// products are formed as linear combinations of ROWS (F P: rows of P; (F P) F^T: rows of (F P)^T), so that the inner loops run
// over contiguous memory and the AVX2 clone does four columns at a time; every element still is the same sequence of separate
// multiplications and additions (no contraction: -ffp-contract=off), i.e. the same bits as the scalar loops it replaces.
LIO_HOST_CLONES static void propagate_covariance(const double* F, const double* W, const double* Q, double* P) {
    constexpr int LD = 24;                                     // rows padded to six 4-wide vectors (the 24th lane carries zeros)
    int fnz[DOF][DOF], fcnt[DOF], wnz[DOF][12], wcnt[DOF];
    for (int r = 0; r < DOF; ++r) {
        fcnt[r] = 0; wcnt[r] = 0;
        for (int k = 0; k < DOF; ++k) if (F[r * DOF + k] != 0.0) fnz[r][fcnt[r]++] = k;
        for (int k = 0; k < 12; ++k) if (W[r * 12 + k] != 0.0) wnz[r][wcnt[r]++] = k;
    }
    alignas(32) double Pp[DOF * LD], FP[DOF * LD], WQ[DOF * 12], FPt[LD * LD], WQt[12 * LD], Pt[DOF * LD];
    for (int r = 0; r < DOF; ++r) {
        for (int c = 0; c < DOF; ++c) Pp[r * LD + c] = P[r * DOF + c];
        Pp[r * LD + DOF] = 0.0;
    }
    for (int r = 0; r < DOF; ++r) {
        double* row = FP + r * LD;
        for (int c = 0; c < LD; ++c) row[c] = 0.0;
        for (int i = 0; i < fcnt[r]; ++i) {
            const int k = fnz[r][i];
            const double f = F[r * DOF + k];
            const double* pk = Pp + k * LD;
            for (int c = 0; c < LD; ++c) row[c] += f * pk[c];
        }
        double* wrow = WQ + r * 12;
        for (int c = 0; c < 12; ++c) wrow[c] = 0.0;
        for (int i = 0; i < wcnt[r]; ++i) {
            const int k = wnz[r][i];
            const double f = W[r * 12 + k];
            for (int c = 0; c < 12; ++c) wrow[c] += f * Q[k * 12 + c];
        }
    }
    for (int k = 0; k < LD; ++k) FPt[k * LD + DOF] = 0.0;
    for (int k = 0; k < 12; ++k) WQt[k * LD + DOF] = 0.0;
    for (int r = 0; r < DOF; ++r) {
        for (int k = 0; k < DOF; ++k) FPt[k * LD + r] = FP[r * LD + k];
        for (int k = 0; k < 12; ++k) WQt[k * LD + r] = WQ[r * 12 + k];
    }
    // column c of the result, as a row over r:  sum_k (F P)[r][k] F[c][k]  +  sum_k (W Q)[r][k] W[c][k]
    for (int c = 0; c < DOF; ++c) {
        alignas(32) double a1[LD], a2[LD];
        for (int r = 0; r < LD; ++r) { a1[r] = 0.0; a2[r] = 0.0; }
        for (int i = 0; i < fcnt[c]; ++i) {
            const int k = fnz[c][i];
            const double f = F[c * DOF + k];
            const double* col = FPt + k * LD;
            for (int r = 0; r < LD; ++r) a1[r] += col[r] * f;
        }
        for (int i = 0; i < wcnt[c]; ++i) {
            const int k = wnz[c][i];
            const double f = W[c * 12 + k];
            const double* col = WQt + k * LD;
            for (int r = 0; r < LD; ++r) a2[r] += col[r] * f;
        }
        for (int r = 0; r < LD; ++r) Pt[c * LD + r] = a1[r] + a2[r];
    }
    for (int r = 0; r < DOF; ++r)
        for (int c = 0; c < DOF; ++c) P[r * DOF + c] = Pt[c * LD + r];
}

void esekf::predict(double dt, const double Q[12 * 12], const input_ikfom& in) {
    double f_[DIM], f_x_[DIM * DOF], f_w_[DIM * 12];
    get_f(x_, in, f_);                                       // esekfom.hpp:281-282, 285
    df_dx(x_, in, f_x_);
    df_dw(x_, in, f_w_);
    const state_ikfom x_before = x_;

    // x_.oplus(f_, dt) (esekfom.hpp:288; vect.hpp:124-126, SOn.hpp:242-245, S2.hpp:129-134)
    x_.pos.x += dt * f_[0]; x_.pos.y += dt * f_[1]; x_.pos.z += dt * f_[2];
    x_.rot = qmul(x_.rot, exp_quat(v3(f_[3], f_[4], f_[5]), dt / 2));
    x_.offset_R_L_I = qmul(x_.offset_R_L_I, exp_quat(v3(f_[6], f_[7], f_[8]), dt / 2));
    x_.offset_T_L_I.x += dt * f_[9]; x_.offset_T_L_I.y += dt * f_[10]; x_.offset_T_L_I.z += dt * f_[11];
    x_.vel.x += dt * f_[12]; x_.vel.y += dt * f_[13]; x_.vel.z += dt * f_[14];
    x_.bg.x += dt * f_[15]; x_.bg.y += dt * f_[16]; x_.bg.z += dt * f_[17];
    x_.ba.x += dt * f_[18]; x_.ba.y += dt * f_[19]; x_.ba.z += dt * f_[20];
    {
        const V3 g = qmatrix(exp_quat(v3(f_[21], f_[22], f_[23]), dt / 2)) * v3(x_.grav);
        x_.grav = {g.a[0], g.a[1], g.a[2]};
    }

    double F_x1[DOF * DOF], f_x_final[DOF * DOF], f_w_final[DOF * 12];
    for (int i = 0; i < DOF * DOF; ++i) F_x1[i] = 0.0;
    for (int i = 0; i < DOF; ++i) F_x1[i * DOF + i] = 1.0;
    // vect sub-states: rows copied through (esekfom.hpp:291-303); {idx, dim} pairs, dof 3 each
    static const int vect_idx[5][2] = {{0, 0}, {9, 9}, {12, 12}, {15, 15}, {18, 18}};
    for (const auto& v : vect_idx)
        for (int j = 0; j < 3; ++j) {
            for (int i = 0; i < DOF; ++i) f_x_final[(v[0] + j) * DOF + i] = f_x_[(v[1] + j) * DOF + i];
            for (int i = 0; i < 12; ++i) f_w_final[(v[0] + j) * 12 + i] = f_w_[(v[1] + j) * 12 + i];
        }
    // SO3 sub-states (esekfom.hpp:304-333).  exp(seg, scalar_type(1/2)): 1/2 is an integer division, the
    // scale is 0 and the "rotation" is the identity -- F_x1's diagonal blocks stay I.
    for (int k = 0; k < 2; ++k) {
        const int idx = kSO3Index[k], dim = kSO3Index[k];
        const V3 seg = v3(-1 * f_[dim] * dt, -1 * f_[dim + 1] * dt, -1 * f_[dim + 2] * dt);
        const M3 E = qmatrix(exp_quat(seg, 0.0));
        for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) F_x1[(idx + i) * DOF + idx + j] = E(i, j);
        const M3 A = A_matrix(seg);
        for (int c = 0; c < DOF; ++c)
            for (int i = 0; i < 3; ++i) {
                double acc = 0.0;
                for (int j = 0; j < 3; ++j) acc += A(i, j) * f_x_[(dim + j) * DOF + c];
                f_x_final[(idx + i) * DOF + c] = acc;
            }
        for (int c = 0; c < 12; ++c)
            for (int i = 0; i < 3; ++i) {
                double acc = 0.0;
                for (int j = 0; j < 3; ++j) acc += A(i, j) * f_w_[(dim + j) * 12 + c];
                f_w_final[(idx + i) * 12 + c] = acc;
            }
    }
    // S2 sub-state (esekfom.hpp:336-372)
    {
        const int idx = kS2Index, dim = 21;
        const V3 seg = v3(f_[dim] * dt, f_[dim + 1] * dt, f_[dim + 2] * dt);
        const M3 E = qmatrix(exp_quat(seg, 0.0));
        const Mat<2, 3> Nx = S2_Nx_yy(v3(x_.grav));
        const Mat<3, 2> Mx = S2_Mx(v3(x_before.grav), 0.0, 0.0);
        const Mat<2, 2> D = Nx * E * Mx;
        for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) F_x1[(idx + i) * DOF + idx + j] = D(i, j);
        Mat<2, 3> T = Nx * E * hat(v3(x_before.grav)) * transpose(A_matrix(seg));
        for (double& v : T.a) v = -v;
        for (int c = 0; c < DOF; ++c)
            for (int i = 0; i < 2; ++i) {
                double acc = 0.0;
                for (int j = 0; j < 3; ++j) acc += T(i, j) * f_x_[(dim + j) * DOF + c];
                f_x_final[(idx + i) * DOF + c] = acc;
            }
        for (int c = 0; c < 12; ++c)
            for (int i = 0; i < 2; ++i) {
                double acc = 0.0;
                for (int j = 0; j < 3; ++j) acc += T(i, j) * f_w_[(dim + j) * 12 + c];
                f_w_final[(idx + i) * 12 + c] = acc;
            }
    }
    // F_x1 += f_x_final * dt;  P = F_x1 P F_x1^T + (dt f_w_final) Q (dt f_w_final)^T     (esekfom.hpp:380-381)
    // F_x1 and f_w_final are mostly zeros (<= 9 / <= 3 entries per row): the products below visit the non-zero
    // entries only, in ascending index order -- the same sums as the dense loops, the skipped terms being exact zeros.
    for (int i = 0; i < DOF * DOF; ++i) F_x1[i] += f_x_final[i] * dt;
    for (int i = 0; i < DOF * 12; ++i) f_w_final[i] = dt * f_w_final[i];
    propagate_covariance(F_x1, f_w_final, Q, P_);
}

// ------------------------------------------------------------------------------------------
// state_ikfom
// ------------------------------------------------------------------------------------------
void state_ikfom::boxplus(const double d[DOF]) {
    // vect: += (mtk/types/vect.hpp:117-119); SO3: q <- q * exp(d) (SOn.hpp:233-236); S2 (S2.hpp:136)
    pos.x += d[0]; pos.y += d[1]; pos.z += d[2];
    rot = qmul(rot, exp_quat(v3(d[3], d[4], d[5]), 0.5));
    offset_R_L_I = qmul(offset_R_L_I, exp_quat(v3(d[6], d[7], d[8]), 0.5));
    offset_T_L_I.x += d[9]; offset_T_L_I.y += d[10]; offset_T_L_I.z += d[11];
    vel.x += d[12]; vel.y += d[13]; vel.z += d[14];
    bg.x += d[15]; bg.y += d[16]; bg.z += d[17];
    ba.x += d[18]; ba.y += d[19]; ba.z += d[20];
    const V3 g = S2_boxplus(v3(grav), d[21], d[22]);
    grav.x = g.a[0]; grav.y = g.a[1]; grav.z = g.a[2];
}
void state_ikfom::boxminus(double r[DOF], const state_ikfom& o) const {
    r[0] = pos.x - o.pos.x; r[1] = pos.y - o.pos.y; r[2] = pos.z - o.pos.z;
    V3 t = so3_log(qmul(qconj(o.rot), rot));                 // SOn.hpp:237-239
    r[3] = t.a[0]; r[4] = t.a[1]; r[5] = t.a[2];
    t = so3_log(qmul(qconj(o.offset_R_L_I), offset_R_L_I));
    r[6] = t.a[0]; r[7] = t.a[1]; r[8] = t.a[2];
    r[9] = offset_T_L_I.x - o.offset_T_L_I.x; r[10] = offset_T_L_I.y - o.offset_T_L_I.y; r[11] = offset_T_L_I.z - o.offset_T_L_I.z;
    r[12] = vel.x - o.vel.x; r[13] = vel.y - o.vel.y; r[14] = vel.z - o.vel.z;
    r[15] = bg.x - o.bg.x; r[16] = bg.y - o.bg.y; r[17] = bg.z - o.bg.z;
    r[18] = ba.x - o.ba.x; r[19] = ba.y - o.ba.y; r[20] = ba.z - o.ba.z;
    S2_boxminus(v3(grav), v3(o.grav), r + 21);
}
void state_ikfom::to_array(double x[26]) const {
    x[0] = pos.x; x[1] = pos.y; x[2] = pos.z;
    x[3] = rot.x; x[4] = rot.y; x[5] = rot.z; x[6] = rot.w;
    x[7] = offset_R_L_I.x; x[8] = offset_R_L_I.y; x[9] = offset_R_L_I.z; x[10] = offset_R_L_I.w;
    x[11] = offset_T_L_I.x; x[12] = offset_T_L_I.y; x[13] = offset_T_L_I.z;
    x[14] = vel.x; x[15] = vel.y; x[16] = vel.z;
    x[17] = bg.x; x[18] = bg.y; x[19] = bg.z;
    x[20] = ba.x; x[21] = ba.y; x[22] = ba.z;
    x[23] = grav.x; x[24] = grav.y; x[25] = grav.z;
}
state_ikfom state_ikfom::from_array(const double x[26]) {
    state_ikfom s;
    s.pos = {x[0], x[1], x[2]};
    s.rot.x = x[3]; s.rot.y = x[4]; s.rot.z = x[5]; s.rot.w = x[6];
    s.offset_R_L_I.x = x[7]; s.offset_R_L_I.y = x[8]; s.offset_R_L_I.z = x[9]; s.offset_R_L_I.w = x[10];
    s.offset_T_L_I = {x[11], x[12], x[13]};
    s.vel = {x[14], x[15], x[16]};
    s.bg = {x[17], x[18], x[19]};
    s.ba = {x[20], x[21], x[22]};
    s.grav = {x[23], x[24], x[25]};
    return s;
}
void state_ikfom::pose14(double p[14]) const {
    p[0] = pos.x; p[1] = pos.y; p[2] = pos.z;
    p[3] = rot.x; p[4] = rot.y; p[5] = rot.z; p[6] = rot.w;
    p[7] = offset_R_L_I.x; p[8] = offset_R_L_I.y; p[9] = offset_R_L_I.z; p[10] = offset_R_L_I.w;
    p[11] = offset_T_L_I.x; p[12] = offset_T_L_I.y; p[13] = offset_T_L_I.z;
}

// ------------------------------------------------------------------------------------------
// esekf
// ------------------------------------------------------------------------------------------
esekf::esekf() {
    for (int i = 0; i < DOF * DOF; ++i) P_[i] = 0.0;
    for (int i = 0; i < DOF; ++i) { P_[i * DOF + i] = 1.0; limit_[i] = 0.001; }
}
void esekf::init_dyn_share(measurementModel_dyn_share h, int maximum_iteration, const double limit_vector[DOF]) {
    h_dyn_share_ = std::move(h);                             // esekfom.hpp:244
    maximum_iter_ = maximum_iteration;                       // esekfom.hpp:246
    for (int i = 0; i < DOF; ++i) limit_[i] = limit_vector[i];
}
void esekf::change_P(const double* P) { std::memcpy(P_, P, sizeof(P_)); }

void esekf::update_iterated_dyn_share_modified(double R, double& solve_time) {
    stats_ = update_stats();
    dyn_share_datastruct dyn_share;
    dyn_share.valid = true;
    dyn_share.converge = true;                               // esekfom.hpp:1623-1624
    int t = 0;
    const state_ikfom x_propagated = x_;                     // :1628
    double P_propagated[DOF * DOF];
    std::memcpy(P_propagated, P_, sizeof(P_));               // :1629

    double K_h[DOF];
    double K_x[DOF * DOF];
    double dx_new[DOF];
    for (double& v : dx_new) v = 0.0;                        // :1635

    for (int i = -1; i < maximum_iter_; ++i) {               // :1636
        dyn_share.valid = true;
        const double t_meas = now_seconds();
        if (dyn_share.converge) ++stats_.n_search;
        // Two-phase measurement model (the GPU path): the pass is enqueued, the part of the update that does not
        // depend on its result is computed beside it, then the result is collected.  With a plain callback the same
        // statements simply run after it -- the arithmetic and its order are identical either way.
        if (h_begin_) h_begin_(x_, dyn_share);               // :1639 (first half)
        else h_dyn_share_(x_, dyn_share);

        double solve_overlapped = now_seconds();
        double dx[DOF];
        x_.boxminus(dx, x_propagated);                       // :1655
        double dx_new_w[DOF], P_w[DOF * DOF];
        for (int k = 0; k < DOF; ++k) dx_new_w[k] = dx[k];
        std::memcpy(P_w, P_propagated, sizeof(P_w));         // :1660
        for (int idx : kSO3Index) {                          // :1665-1680
            const M3 J = transpose(A_matrix(v3(dx[idx], dx[idx + 1], dx[idx + 2])));
            const V3 seg = J * v3(dx_new_w[idx], dx_new_w[idx + 1], dx_new_w[idx + 2]);
            dx_new_w[idx] = seg.a[0]; dx_new_w[idx + 1] = seg.a[1]; dx_new_w[idx + 2] = seg.a[2];
            rows_times<3>(P_w, idx, J);
            cols_times_T<3>(P_w, idx, J);
        }
        {                                                    // :1684-1703
            const int idx = kS2Index;
            const Mat<2, 2> J = S2_Nx_yy(v3(x_.grav)) * S2_Mx(v3(x_propagated.grav), dx[idx], dx[idx + 1]);
            const double d0 = J(0, 0) * dx_new_w[idx] + J(0, 1) * dx_new_w[idx + 1];
            const double d1 = J(1, 0) * dx_new_w[idx] + J(1, 1) * dx_new_w[idx + 1];
            dx_new_w[idx] = d0; dx_new_w[idx + 1] = d1;
            rows_times<2>(P_w, idx, J);
            cols_times_T<2>(P_w, idx, J);
        }
        double PR[DOF * DOF], P_temp[DOF * DOF];             // (P_/R).inverse()                    :1788
        for (int k = 0; k < DOF * DOF; ++k) PR[k] = P_w[k] / R;
        invert23(PR, P_temp);
        solve_overlapped = now_seconds() - solve_overlapped;

        if (h_begin_) h_end_(dyn_share);                     // :1639 (second half: wait for the pass)
        stats_.measure_seconds += now_seconds() - t_meas - (h_begin_ ? 0.0 : solve_overlapped);
        ++stats_.n_calls;
        stats_.last_m = dyn_share.m;
        stats_.last_res_mean = dyn_share.res_mean;
        if (!dyn_share.valid) continue;                      // :1641-1644 (nothing of this pass is kept)

        const double solve_start = now_seconds() - solve_overlapped;
        const int dof_Measurement = dyn_share.m;             // :1653
        for (int k = 0; k < DOF; ++k) dx_new[k] = dx_new_w[k];
        std::memcpy(P_, P_w, sizeof(P_));

        if (DOF > dof_Measurement) {
            // fewer rows than states: K = P H^T (H P H^T / R + I)^-1 / R, H zero-padded   :1720-1750
            const int m = dof_Measurement;
            std::vector<double> H((size_t)m * DOF, 0.0);
            for (int r = 0; r < m; ++r)
                for (int c = 0; c < 12; ++c) H[(size_t)r * DOF + c] = dyn_share.h_x[(size_t)r * 12 + c];
            std::vector<double> PHt((size_t)DOF * m), S((size_t)m * m), Sinv((size_t)m * m), Kg((size_t)DOF * m);
            for (int a = 0; a < DOF; ++a)
                for (int r = 0; r < m; ++r) {
                    double s = 0.0;
                    for (int c = 0; c < DOF; ++c) s += P_[a * DOF + c] * H[(size_t)r * DOF + c];
                    PHt[(size_t)a * m + r] = s;
                }
            for (int r = 0; r < m; ++r)
                for (int q = 0; q < m; ++q) {
                    double s = 0.0;
                    for (int c = 0; c < DOF; ++c) s += H[(size_t)r * DOF + c] * PHt[(size_t)c * m + q];
                    S[(size_t)r * m + q] = s / R + (r == q ? 1.0 : 0.0);
                }
            invert(S.data(), Sinv.data(), m);
            for (int a = 0; a < DOF; ++a)
                for (int q = 0; q < m; ++q) {
                    double s = 0.0;
                    for (int r = 0; r < m; ++r) s += PHt[(size_t)a * m + r] * Sinv[(size_t)r * m + q];
                    Kg[(size_t)a * m + q] = s / R;
                }
            for (int a = 0; a < DOF; ++a) {
                double s = 0.0;
                for (int r = 0; r < m; ++r) s += Kg[(size_t)a * m + r] * dyn_share.h[r];
                K_h[a] = s;
                for (int c = 0; c < DOF; ++c) {
                    double v = 0.0;
                    for (int r = 0; r < m; ++r) v += Kg[(size_t)a * m + r] * H[(size_t)r * DOF + c];
                    K_x[a * DOF + c] = v;
                }
            }
        } else {
            // information form                                                      :1788-1815
            double P_inv12[DOF * 12];
            for (int a = 0; a < 12; ++a)
                for (int b = 0; b < 12; ++b) P_temp[a * DOF + b] += dyn_share.HTH[a * 12 + b];   // :1791
            // :1808 P_inv = P_temp.inverse(); only its first 12 columns are used (:1810, :1815)
            invert23_first12(P_temp, P_inv12);
            for (int a = 0; a < DOF; ++a) {                  // K_h = P_inv[:, 0:12] H^T h          :1810
                double s = 0.0;
                for (int c = 0; c < 12; ++c) s += P_inv12[a * 12 + c] * dyn_share.HTh[c];
                K_h[a] = s;
            }
            for (double& v : K_x) v = 0.0;                   // :1814
            for (int a = 0; a < DOF; ++a)                    // K_x[:, 0:12] = P_inv[:, 0:12] HTH   :1815
                for (int b = 0; b < 12; ++b) {
                    double s = 0.0;
                    for (int c = 0; c < 12; ++c) s += P_inv12[a * 12 + c] * dyn_share.HTH[c * 12 + b];
                    K_x[a * DOF + b] = s;
                }
        }

        double dx_[DOF];                                     // dx_ = K_h + (K_x - I) dx_new      :1821
        for (int a = 0; a < DOF; ++a) {
            double s = 0.0;
            for (int c = 0; c < DOF; ++c) s += (K_x[a * DOF + c] - (a == c ? 1.0 : 0.0)) * dx_new[c];
            dx_[a] = K_h[a] + s;
        }
        x_.boxplus(dx_);                                     // :1823
        dyn_share.converge = true;                           // :1824-1832
        for (int k = 0; k < DOF; ++k) {
            if (std::fabs(dx_[k]) > limit_[k]) { dyn_share.converge = false; break; }
        }
        if (dyn_share.converge) ++t;
        if (!t && i == maximum_iter_ - 2) dyn_share.converge = true;   // :1835-1838

        if (t > 1 || i == maximum_iter_ - 1) {               // covariance update           :1841-1935
            double L_[DOF * DOF];
            std::memcpy(L_, P_, sizeof(P_));                 // :1843
            for (int idx : kSO3Index) {                      // :1847-1872
                const M3 J = transpose(A_matrix(v3(dx_[idx], dx_[idx + 1], dx_[idx + 2])));
                for (int c = 0; c < DOF; ++c) {              // L rows <- J * P rows
                    const V3 col = J * v3(P_[idx * DOF + c], P_[(idx + 1) * DOF + c], P_[(idx + 2) * DOF + c]);
                    L_[idx * DOF + c] = col.a[0]; L_[(idx + 1) * DOF + c] = col.a[1]; L_[(idx + 2) * DOF + c] = col.a[2];
                }
                for (int c = 0; c < 12; ++c) {               // K_x rows (first 12 columns)
                    const V3 col = J * v3(K_x[idx * DOF + c], K_x[(idx + 1) * DOF + c], K_x[(idx + 2) * DOF + c]);
                    K_x[idx * DOF + c] = col.a[0]; K_x[(idx + 1) * DOF + c] = col.a[1]; K_x[(idx + 2) * DOF + c] = col.a[2];
                }
                cols_times_T<3>(L_, idx, J);
                cols_times_T<3>(P_, idx, J);
            }
            {                                                // :1874-1907
                const int idx = kS2Index;
                const Mat<2, 2> J = S2_Nx_yy(v3(x_.grav)) * S2_Mx(v3(x_propagated.grav), dx_[idx], dx_[idx + 1]);
                for (int c = 0; c < DOF; ++c) {
                    const double a0 = P_[idx * DOF + c], a1 = P_[(idx + 1) * DOF + c];
                    L_[idx * DOF + c] = J(0, 0) * a0 + J(0, 1) * a1;
                    L_[(idx + 1) * DOF + c] = J(1, 0) * a0 + J(1, 1) * a1;
                }
                for (int c = 0; c < 12; ++c) {
                    const double a0 = K_x[idx * DOF + c], a1 = K_x[(idx + 1) * DOF + c];
                    K_x[idx * DOF + c] = J(0, 0) * a0 + J(0, 1) * a1;
                    K_x[(idx + 1) * DOF + c] = J(1, 0) * a0 + J(1, 1) * a1;
                }
                cols_times_T<2>(L_, idx, J);
                cols_times_T<2>(P_, idx, J);
            }
            // P_ = L_ - K_x[:, 0:12] * P_[0:12, :]                                       :1931
            double Pn[DOF * DOF];
            for (int a = 0; a < DOF; ++a)
                for (int c = 0; c < DOF; ++c) {
                    double s = 0.0;
                    for (int b = 0; b < 12; ++b) s += K_x[a * DOF + b] * P_[b * DOF + c];
                    Pn[a * DOF + c] = L_[a * DOF + c] - s;
                }
            std::memcpy(P_, Pn, sizeof(P_));
            stats_.converged = t > 1;
            const double dt = now_seconds() - solve_start;
            solve_time += dt; stats_.solve_seconds += dt;
            return;
        }
        const double dt = now_seconds() - solve_start;
        solve_time += dt; stats_.solve_seconds += dt;
    }
}

}  // namespace lio
