// TEST INFRASTRUCTURE ONLY -- CPU oracle for the LIO hot path (see lio_oracle.h for the rules
// and the pinning status).  Plain C++17, no third-party headers.  Built by oracle/Makefile with
// -O3 -ffp-contract=off and no -march flag, i.e. the arithmetic the reference gets from a catkin
// Release build on baseline x86-64.
//
// Every function cites the reference lines it restates (paths relative to /root/reference).
#include "lio_oracle.h"

#include <algorithm>
#include <cfloat>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <limits>
#include <unordered_map>
#include <vector>

namespace {

// =====================================================================================
//  small fp64 algebra in the evaluation order of Eigen 3.3.7's fixed-size kernels
// =====================================================================================
struct V3 { double x, y, z; };
struct Q4 { double x, y, z, w; };          // Eigen::Quaterniond coefficient order (x,y,z,w)

inline V3 operator+(V3 a, V3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline V3 operator-(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline V3 operator*(double s, V3 a) { return {s * a.x, s * a.y, s * a.z}; }
inline V3 cross(V3 a, V3 b) {               // Eigen MatrixBase::cross
    return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}
inline double dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline double norm(V3 a) { return std::sqrt(dot(a, a)); }
inline Q4 conj(Q4 q) { return {-q.x, -q.y, -q.z, q.w}; }

// Eigen 3.3.7 QuaternionBase::_transformVector: uv = 2 (qv x v); v + w uv + qv x uv
inline V3 qrot(Q4 q, V3 v) {
    V3 qv{q.x, q.y, q.z};
    V3 uv = cross(qv, v);
    uv = uv + uv;
    V3 t = cross(qv, uv);
    return {(v.x + q.w * uv.x) + t.x, (v.y + q.w * uv.y) + t.y, (v.z + q.w * uv.z) + t.z};
}
// Eigen generic quat_product
inline Q4 qmul(Q4 a, Q4 b) {
    return {a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y,
            a.w * b.y + a.y * b.w + a.z * b.x - a.x * b.z,
            a.w * b.z + a.z * b.w + a.x * b.y - a.y * b.x,
            a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z};
}
struct M3 { double m[3][3]; };
// Eigen QuaternionBase::toRotationMatrix
inline M3 qmat(Q4 q) {
    double tx = 2 * q.x, ty = 2 * q.y, tz = 2 * q.z;
    double twx = tx * q.w, twy = ty * q.w, twz = tz * q.w;
    double txx = tx * q.x, txy = ty * q.x, txz = tz * q.x;
    double tyy = ty * q.y, tyz = tz * q.y, tzz = tz * q.z;
    M3 r;
    r.m[0][0] = 1 - (tyy + tzz); r.m[0][1] = txy - twz;       r.m[0][2] = txz + twy;
    r.m[1][0] = txy + twz;       r.m[1][1] = 1 - (txx + tzz); r.m[1][2] = tyz - twx;
    r.m[2][0] = txz - twy;       r.m[2][1] = tyz + twx;       r.m[2][2] = 1 - (txx + tyy);
    return r;
}
inline M3 hat(V3 v) {                       // mtkmath.hpp:176-183 / SKEW_SYM_MATRX common_lib.h
    M3 r = {{{0, -v.z, v.y}, {v.z, 0, -v.x}, {-v.y, v.x, 0}}};
    return r;
}
inline M3 mmul(const M3& a, const M3& b) {
    M3 r;
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j)
            r.m[i][j] = (a.m[i][0] * b.m[0][j] + a.m[i][1] * b.m[1][j]) + a.m[i][2] * b.m[2][j];
    return r;
}
inline V3 mvec(const M3& a, V3 v) {
    return {(a.m[0][0] * v.x + a.m[0][1] * v.y) + a.m[0][2] * v.z,
            (a.m[1][0] * v.x + a.m[1][1] * v.y) + a.m[1][2] * v.z,
            (a.m[2][0] * v.x + a.m[2][1] * v.y) + a.m[2][2] * v.z};
}
inline M3 mT(const M3& a) {
    M3 r;
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) r.m[i][j] = a.m[j][i];
    return r;
}

struct Pose {                                // the 14-double block of the C ABI
    V3 pos; Q4 rot; Q4 offR; V3 offT;
};
inline Pose pose_from(const double* s) {
    return {{s[0], s[1], s[2]}, {s[3], s[4], s[5], s[6]}, {s[7], s[8], s[9], s[10]},
            {s[11], s[12], s[13]}};
}

// =====================================================================================
//  map live-set model (ikd-Tree visible semantics)
// =====================================================================================
struct P3 { float x, y, z; };

// ikd_Tree.cpp:1374-1378 : fp32, (dx*dx + dy*dy) + dz*dz, no contraction
inline float calc_dist(P3 a, P3 b) {
    float dist = (a.x - b.x) * (a.x - b.x) + (a.y - b.y) * (a.y - b.y) + (a.z - b.z) * (a.z - b.z);
    return dist;
}

// PointType_CMP::operator< (ikd_Tree.h:102-105)
struct Cand { P3 p; float d; };
inline bool cand_less(const Cand& a, const Cand& b) {
    if (std::fabs(a.d - b.d) < 1e-10) return a.p.x < b.p.x;
    return a.d < b.d;
}

// Bounded candidate list with the accept rule of KD_TREE::Search (ikd_Tree.cpp:885-892):
// take while fewer than k, afterwards only if strictly closer than the current worst.
struct TopK {
    int k, n = 0;
    Cand c[16];
    explicit TopK(int k_) : k(k_) {}
    inline float worst() const { return c[n - 1].d; }
    inline void offer(P3 p, float d) {
        if (n >= k && !(d < worst())) return;
        if (n >= k) --n;                                  // q.pop()
        Cand nc{p, d};
        int i = n;
        while (i > 0 && cand_less(nc, c[i - 1])) { c[i] = c[i - 1]; --i; }
        c[i] = nc;
        ++n;
    }
};

struct KdNode {
    int idx; int left, right; uint8_t axis;
    float lo[3], hi[3];
};

struct OMap {
    float ds = 0.2f;                                      // downsample_size ikd_Tree.h:197
    std::vector<P3> pts;
    std::vector<uint8_t> alive;
    int live = 0;
    // voxel buckets (cell = floor(p/ds), fp32 like ikd_Tree.cpp:424-429) for the box search of
    // the down-sampling insert; holds indices of LIVE points only
    std::unordered_map<uint64_t, std::vector<int>> vox;
    // static k-d tree over the live set, rebuilt when dirty
    std::vector<KdNode> nodes;
    std::vector<int> order;
    int root = -1;
    bool dirty = true;

    static inline uint64_t vkey(int ix, int iy, int iz) {
        return ((uint64_t)(uint32_t)(ix + (1 << 20)) << 42) |
               ((uint64_t)(uint32_t)(iy + (1 << 20)) << 21) | (uint64_t)(uint32_t)(iz + (1 << 20));
    }
    inline void cell_of(P3 p, int& ix, int& iy, int& iz) const {
        ix = (int)std::floor(p.x / ds); iy = (int)std::floor(p.y / ds); iz = (int)std::floor(p.z / ds);
    }
    void push(P3 p) {
        int id = (int)pts.size();
        pts.push_back(p); alive.push_back(1); ++live;
        int ix, iy, iz; cell_of(p, ix, iy, iz);
        vox[vkey(ix, iy, iz)].push_back(id);
        dirty = true;
    }
    void kill(int id) {
        if (!alive[id]) return;
        alive[id] = 0; --live;
        int ix, iy, iz; cell_of(pts[id], ix, iy, iz);
        auto it = vox.find(vkey(ix, iy, iz));
        if (it != vox.end()) {
            auto& v = it->second;
            for (size_t i = 0; i < v.size(); ++i) if (v[i] == id) { v[i] = v.back(); v.pop_back(); break; }
            if (v.empty()) vox.erase(it);
        }
        dirty = true;
    }
    void clear() { pts.clear(); alive.clear(); vox.clear(); live = 0; dirty = true; nodes.clear(); root = -1; }

    // ---- BuildTree (ikd_Tree.cpp:574-621): split on the widest axis at the median ----
    int build_rec(int l, int r) {
        if (l > r) return -1;
        int mid = (l + r) >> 1;
        float mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
        for (int i = l; i <= r; ++i) {
            const P3& p = pts[order[i]];
            mn[0] = std::min(mn[0], p.x); mn[1] = std::min(mn[1], p.y); mn[2] = std::min(mn[2], p.z);
            mx[0] = std::max(mx[0], p.x); mx[1] = std::max(mx[1], p.y); mx[2] = std::max(mx[2], p.z);
        }
        int ax = 0;
        float rg[3] = {mx[0] - mn[0], mx[1] - mn[1], mx[2] - mn[2]};
        for (int i = 1; i < 3; ++i) if (rg[i] > rg[ax]) ax = i;
        auto cmp = [&](int a, int b) {
            const P3 &pa = pts[a], &pb = pts[b];
            return ax == 0 ? pa.x < pb.x : ax == 1 ? pa.y < pb.y : pa.z < pb.z;
        };
        std::nth_element(order.begin() + l, order.begin() + mid, order.begin() + r + 1, cmp);
        int me = (int)nodes.size();
        nodes.push_back(KdNode());
        nodes[me].idx = order[mid];
        nodes[me].axis = (uint8_t)ax;
        for (int a = 0; a < 3; ++a) { nodes[me].lo[a] = mn[a]; nodes[me].hi[a] = mx[a]; }
        int L = build_rec(l, mid - 1);
        int R = build_rec(mid + 1, r);
        nodes[me].left = L; nodes[me].right = R;
        return me;
    }
    void ensure_tree() {
        if (!dirty) return;
        order.clear(); nodes.clear();
        for (int i = 0; i < (int)pts.size(); ++i) if (alive[i]) order.push_back(i);
        nodes.reserve(order.size());
        root = build_rec(0, (int)order.size() - 1);
        dirty = false;
    }
    // calc_box_dist (ikd_Tree.cpp:1381-1391)
    inline float box_dist(int n, P3 q) const {
        if (n < 0) return INFINITY;
        const KdNode& nd = nodes[n];
        float m = 0.0f;
        if (q.x < nd.lo[0]) m += (q.x - nd.lo[0]) * (q.x - nd.lo[0]);
        if (q.x > nd.hi[0]) m += (q.x - nd.hi[0]) * (q.x - nd.hi[0]);
        if (q.y < nd.lo[1]) m += (q.y - nd.lo[1]) * (q.y - nd.lo[1]);
        if (q.y > nd.hi[1]) m += (q.y - nd.hi[1]) * (q.y - nd.hi[1]);
        if (q.z < nd.lo[2]) m += (q.z - nd.lo[2]) * (q.z - nd.lo[2]);
        if (q.z > nd.hi[2]) m += (q.z - nd.hi[2]) * (q.z - nd.hi[2]);
        return m;
    }
    // KD_TREE::Search (ikd_Tree.cpp:869-1013) with max_dist = INFINITY, without the locks
    void search(int n, P3 q, TopK& h) const {
        if (n < 0) return;
        const KdNode& nd = nodes[n];
        float dist = calc_dist(q, pts[nd.idx]);
        h.offer(pts[nd.idx], dist);
        float dl = box_dist(nd.left, q), dr = box_dist(nd.right, q);
        if (h.n < h.k || (dl < h.worst() && dr < h.worst())) {
            if (dl <= dr) {
                search(nd.left, q, h);
                if (h.n < h.k || dr < h.worst()) search(nd.right, q, h);
            } else {
                search(nd.right, q, h);
                if (h.n < h.k || dl < h.worst()) search(nd.left, q, h);
            }
        } else {
            if (dl < h.worst()) search(nd.left, q, h);
            if (dr < h.worst()) search(nd.right, q, h);
        }
    }
    void knn(P3 q, int k, TopK& h, int method) {
        if (method == 0) {
            for (size_t i = 0; i < pts.size(); ++i)
                if (alive[i]) h.offer(pts[i], calc_dist(q, pts[i]));
        } else {
            ensure_tree();
            search(root, q, h);
        }
    }
};

// Optional external k-NN provider (knn_method == 2): lets the harness run the restated measurement
// model on top of the REFERENCE's own ikd-Tree (oracle/_ref/libikd_ref.so, ref_ikd_nearest), which
// is what bench.py's cpu_baseline times.
typedef double (*ext_knn_fn)(void* h, const float* q_xyz, int nq, int k, float* out_xyz, float* out_d, int* out_cnt);
ext_knn_fn g_ext_knn = nullptr;
void* g_ext_handle = nullptr;

inline bool same_point(P3 a, P3 b) {                       // ikd_Tree.cpp:1365-1367 (EPSS 1e-6)
    return std::fabs(a.x - b.x) < 1e-6 && std::fabs(a.y - b.y) < 1e-6 && std::fabs(a.z - b.z) < 1e-6;
}

// =====================================================================================
//  esti_plane<float>  (include/common_lib.h:506-536) on Eigen 3.3.7's
//  ColPivHouseholderQR<Matrix<float,5,3>>::computeInPlace + _solve_impl, fp32 throughout
// =====================================================================================
// ---- summation order of the reductions inside Eigen's QR (TEST SWITCH, orc_set_sum_order) ------------------------
// "Parity unpinned" in numbers: every squaredNorm() / dot product of ColPivHouseholderQR runs through Eigen's redux,
// whose association depends on how the reference was built.  Un-vectorised (EIGEN_DONT_VECTORIZE, or a block the
// packet path rejects) it is the sequential sum restated here (mode 0).  Vectorised with SSE2 (x86-64 default,
// CMakeLists.txt sets no -m flags) a 4-float packet is reduced as (a0 + a2) + (a1 + a3) and the scalar tail is added
// afterwards (mode 1); with -msse3 and later the packet is reduced by hadd, (a0 + a1) + (a2 + a3) (mode 2); where the
// packet starts depends on the alignment of the 60-byte matrix on the stack, which no source fixes: mode 3 draws, per
// reduction, a random permutation and a random association (seeded, reproducible) and so bounds every such variant.
// The modes change NOTHING but the order in which the same fp32 terms are added.
static int g_sum_mode = 0;
static uint64_t g_sum_rng = 0x9e3779b97f4a7c15ull;
static inline uint32_t sum_rand() {                        // xorshift64*
    g_sum_rng ^= g_sum_rng >> 12; g_sum_rng ^= g_sum_rng << 25; g_sum_rng ^= g_sum_rng >> 27;
    return (uint32_t)((g_sum_rng * 0x2545F4914F6CDD1Dull) >> 32);
}
static float sum_terms(const float* t, int n) {
    if (n <= 0) return 0.0f;
    if (g_sum_mode == 0 || n < 3) {                        // sequential (two terms have one order)
        float s = t[0];
        for (int i = 1; i < n; ++i) s += t[i];
        return s;
    }
    if (g_sum_mode == 1 || g_sum_mode == 2) {
        if (n == 3) return t[0] + (t[1] + t[2]);           // Eigen's unrolled scalar redux of a fixed 3-vector splits 1 | 2
        float s = g_sum_mode == 1 ? (t[0] + t[2]) + (t[1] + t[3]) : (t[0] + t[1]) + (t[2] + t[3]);
        for (int i = 4; i < n; ++i) s += t[i];
        return s;
    }
    // random permutation, random association: repeatedly add two randomly chosen partial sums
    float v[8];
    for (int i = 0; i < n; ++i) v[i] = t[i];
    int m = n;
    while (m > 1) {
        const int i = (int)(sum_rand() % (uint32_t)m);
        int j = (int)(sum_rand() % (uint32_t)(m - 1));
        if (j >= i) ++j;
        const float sij = v[i] + v[j];
        const int lo = i < j ? i : j, hi = i < j ? j : i;
        v[lo] = sij; v[hi] = v[m - 1];
        --m;
    }
    return v[0];
}

bool esti_plane_f(float pabcd[4], const P3 pt[5], float threshold) {
    const int rows = 5, cols = 3;
    float A[5][3];
    float b[5];
    for (int j = 0; j < rows; ++j) { A[j][0] = pt[j].x; A[j][1] = pt[j].y; A[j][2] = pt[j].z; b[j] = -1.0f; }

    float hc[3] = {0, 0, 0};
    int trans[3] = {0, 1, 2};
    float cnu[3], cnd[3];                                  // m_colNormsUpdated / m_colNormsDirect
    auto colnorm = [&](int c, int r0) {
        float t[5]; int n = 0;
        for (int r = r0; r < rows; ++r) t[n++] = A[r][c] * A[r][c];
        return std::sqrt(sum_terms(t, n));
    };
    for (int k = 0; k < cols; ++k) { cnd[k] = colnorm(k, 0); cnu[k] = cnd[k]; }
    const float eps = std::numeric_limits<float>::epsilon();
    float mx = std::max(cnu[0], std::max(cnu[1], cnu[2]));
    float t0 = mx * eps;
    const float threshold_helper = (t0 * t0) / float(rows);
    const float norm_downdate_threshold = std::sqrt(eps);
    int nonzero_pivots = cols;
    for (int k = 0; k < cols; ++k) {
        int big = k;
        for (int j = k + 1; j < cols; ++j) if (cnu[j] > cnu[big]) big = j;   // maxCoeff: first max
        float big_sq = cnu[big] * cnu[big];
        if (nonzero_pivots == cols && big_sq < threshold_helper * float(rows - k)) nonzero_pivots = k;
        trans[k] = big;
        if (k != big) {
            for (int r = 0; r < rows; ++r) std::swap(A[r][k], A[r][big]);
            std::swap(cnu[k], cnu[big]);
            std::swap(cnd[k], cnd[big]);
        }
        // makeHouseholderInPlace on A[k:,k]
        float tailSq;
        {
            float t[5]; int n = 0;
            for (int r = k + 1; r < rows; ++r) t[n++] = A[r][k] * A[r][k];
            tailSq = sum_terms(t, n);
        }
        float c0 = A[k][k];
        float beta, tau;
        const float tol = std::numeric_limits<float>::min();
        if (tailSq <= tol) {
            tau = 0.0f; beta = c0;
            for (int r = k + 1; r < rows; ++r) A[r][k] = 0.0f;
        } else {
            beta = std::sqrt(c0 * c0 + tailSq);
            if (c0 >= 0.0f) beta = -beta;
            float den = c0 - beta;
            for (int r = k + 1; r < rows; ++r) A[r][k] = A[r][k] / den;
            tau = (beta - c0) / beta;
        }
        hc[k] = tau;
        A[k][k] = beta;
        // applyHouseholderOnTheLeft to A[k:, k+1:]
        if (rows - k == 1) {
            for (int j = k + 1; j < cols; ++j) A[k][j] *= (1.0f - tau);
        } else if (tau != 0.0f) {
            for (int j = k + 1; j < cols; ++j) {
                float t[5]; int n = 0;
                for (int r = k + 1; r < rows; ++r) t[n++] = A[r][k] * A[r][j];
                float tmp = sum_terms(t, n);
                tmp += A[k][j];
                A[k][j] -= tau * tmp;
                for (int r = k + 1; r < rows; ++r) A[r][j] -= (tau * A[r][k]) * tmp;
            }
        }
        // norm down-dating (LAPACK working note 176)
        for (int j = k + 1; j < cols; ++j) {
            if (cnu[j] != 0.0f) {
                float temp = std::fabs(A[k][j]) / cnu[j];
                temp = (1.0f + temp) * (1.0f - temp);
                temp = temp < 0.0f ? 0.0f : temp;
                float r2 = cnu[j] / cnd[j];
                float temp2 = temp * (r2 * r2);
                if (temp2 <= norm_downdate_threshold) {
                    cnd[j] = colnorm(j, k + 1);
                    cnu[j] = cnd[j];
                } else {
                    cnu[j] *= std::sqrt(temp);
                }
            }
        }
    }
    // column permutation from the transpositions
    int perm[3] = {0, 1, 2};
    for (int k = 0; k < cols; ++k) std::swap(perm[k], perm[trans[k]]);

    float nv[3] = {0, 0, 0};
    if (nonzero_pivots > 0) {
        // c = Q^T b : H_0 first
        float c[5];
        for (int r = 0; r < rows; ++r) c[r] = b[r];
        for (int k = 0; k < nonzero_pivots; ++k) {
            float tau = hc[k];
            if (rows - k == 1) { c[k] *= (1.0f - tau); continue; }
            if (tau != 0.0f) {
                float t[5]; int n = 0;
                for (int r = k + 1; r < rows; ++r) t[n++] = A[r][k] * c[r];
                float tmp = sum_terms(t, n);
                tmp += c[k];
                c[k] -= tau * tmp;
                for (int r = k + 1; r < rows; ++r) c[r] -= (tau * A[r][k]) * tmp;
            }
        }
        // upper-triangular solve, column oriented (Eigen ColMajor triangular_solve_vector)
        for (int i = nonzero_pivots - 1; i >= 0; --i) {
            c[i] = c[i] / A[i][i];
            for (int r = 0; r < i; ++r) c[r] -= c[i] * A[r][i];
        }
        for (int i = 0; i < nonzero_pivots; ++i) nv[perm[i]] = c[i];
        for (int i = nonzero_pivots; i < cols; ++i) nv[perm[i]] = 0.0f;
    }
    float n;                                                                 // normvec.norm(), common_lib.h:522
    {
        const float t[3] = {nv[0] * nv[0], nv[1] * nv[1], nv[2] * nv[2]};
        n = std::sqrt(sum_terms(t, 3));
    }
    pabcd[0] = nv[0] / n; pabcd[1] = nv[1] / n; pabcd[2] = nv[2] / n;
    pabcd[3] = (float)(1.0 / (double)n);                                     // `1.0 / n` :526
    for (int j = 0; j < rows; ++j) {
        if (std::fabs(pabcd[0] * pt[j].x + pabcd[1] * pt[j].y + pabcd[2] * pt[j].z + pabcd[3]) > threshold)
            return false;
    }
    return true;
}

// =====================================================================================
//  MTK manifold pieces (double)
// =====================================================================================
const double MTK_TOL = 1e-11;                              // mtkmath.hpp:122

// cos_sinc_sqrt (mtkmath.hpp:142-174)
inline void cos_sinc_sqrt(double x2, double& c, double& s) {
    static const double taylor_0 = std::numeric_limits<double>::epsilon();
    static const double taylor_2 = std::sqrt(taylor_0);
    static const double taylor_n = std::sqrt(taylor_2);
    if (x2 >= taylor_n) {
        double x = std::sqrt(x2);
        c = std::cos(x); s = std::sin(x) / x;
        return;
    }
    static const double inv[] = {1 / 3., 1 / 4., 1 / 5., 1 / 6., 1 / 7., 1 / 8., 1 / 9.};
    double cosi = 1., sinc = 1.;
    double term = -1 / 2. * x2;
    for (int i = 0; i < 3; ++i) {
        cosi += term; term *= inv[2 * i];
        sinc += term; term *= -inv[2 * i + 1] * x2;
    }
    c = cosi; s = sinc;
}
// MTK::exp (mtkmath.hpp:250-256) -> quaternion (vec = mult*v, w = cos)
inline Q4 mtk_exp(V3 v, double scale) {
    double n2 = dot(v, v);
    double c, s;
    cos_sinc_sqrt(scale * scale * n2, c, s);
    double mult = s * scale;
    return {mult * v.x, mult * v.y, mult * v.z, c};
}
// MTK::log (mtkmath.hpp:268-288), plus_minus_periodicity = true, scale = 2 (SOn.hpp:293-297)
inline V3 so3_log(Q4 q) {
    V3 v{q.x, q.y, q.z};
    double nv = norm(v);
    if (nv < MTK_TOL) nv = MTK_TOL;
    double s = 2.0 / nv * std::atan(nv / q.w);
    return s * v;
}
// A_matrix (mtkmath.hpp:236-247)
inline M3 A_matrix(V3 v) {
    M3 I = {{{1, 0, 0}, {0, 1, 0}, {0, 0, 1}}};
    double sq = v.x * v.x + v.y * v.y + v.z * v.z;
    double n = std::sqrt(sq);
    if (n < MTK_TOL) return I;
    M3 h = hat(v), hh = mmul(h, h);
    double a = (1 - std::cos(n)) / sq, b = (1 - std::sin(n) / n) / sq;
    M3 r;
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) r.m[i][j] = I.m[i][j] + a * h.m[i][j] + b * hh.m[i][j];
    return r;
}

const double S2_LEN = 98090.0 / 10000.0;                   // use-ikfom.hpp:8, S2.hpp:104

// S2_Bx for S2_typ == 1 (S2.hpp:213-231)
inline void S2_Bx(V3 vec, double Bx[3][2]) {
    const double L = S2_LEN;
    if (vec.x + L > MTK_TOL) {
        Bx[0][0] = -vec.y;                          Bx[0][1] = -vec.z;
        Bx[1][0] = L - vec.y * vec.y / (L + vec.x); Bx[1][1] = -vec.z * vec.y / (L + vec.x);
        Bx[2][0] = -vec.z * vec.y / (L + vec.x);    Bx[2][1] = L - vec.z * vec.z / (L + vec.x);
        for (int i = 0; i < 3; ++i) for (int j = 0; j < 2; ++j) Bx[i][j] /= L;
    } else {
        for (int i = 0; i < 3; ++i) for (int j = 0; j < 2; ++j) Bx[i][j] = 0;
        Bx[1][1] = -1; Bx[2][0] = 1;
    }
}
// S2::boxplus (S2.hpp:136-142)
inline V3 S2_boxplus(V3 vec, double d0, double d1) {
    double Bx[3][2];
    S2_Bx(vec, Bx);
    V3 Bu{Bx[0][0] * d0 + Bx[0][1] * d1, Bx[1][0] * d0 + Bx[1][1] * d1, Bx[2][0] * d0 + Bx[2][1] * d1};
    Q4 r = mtk_exp(Bu, 0.5);
    return mvec(qmat(r), vec);
}
// S2::boxminus (S2.hpp:144-167): res = this [-] other
inline void S2_boxminus(V3 vec, V3 other, double res[2]) {
    double v_sin = norm(mvec(hat(vec), other));
    double v_cos = dot(vec, other);
    double theta = std::atan2(v_sin, v_cos);
    if (v_sin < MTK_TOL) {
        if (std::fabs(theta) > MTK_TOL) { res[0] = 3.1415926; res[1] = 0; }
        else { res[0] = 0; res[1] = 0; }
    } else {
        double Bx[3][2];
        S2_Bx(other, Bx);
        V3 t = mvec(hat(other), vec);
        double f = theta / v_sin;
        res[0] = f * (Bx[0][0] * t.x + Bx[1][0] * t.y + Bx[2][0] * t.z);
        res[1] = f * (Bx[0][1] * t.x + Bx[1][1] * t.y + Bx[2][1] * t.z);
    }
}
// S2_Nx_yy (S2.hpp:259-264): 1/L/L * Bx^T * hat(vec)  -> 2x3
inline void S2_Nx_yy(V3 vec, double Nx[2][3]) {
    double Bx[3][2];
    S2_Bx(vec, Bx);
    M3 h = hat(vec);
    for (int i = 0; i < 2; ++i)
        for (int j = 0; j < 3; ++j) {
            double s = 0;
            for (int k = 0; k < 3; ++k) s += Bx[k][i] * h.m[k][j];
            Nx[i][j] = 1 / S2_LEN / S2_LEN * s;
        }
}
// S2_Mx (S2.hpp:266-280): note scalar(1/2) == 0 there => exp_delta is the identity
inline void S2_Mx(V3 vec, double d0, double d1, double Mx[3][2]) {
    double Bx[3][2];
    S2_Bx(vec, Bx);
    M3 h = hat(vec);
    if (std::sqrt(d0 * d0 + d1 * d1) < MTK_TOL) {
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 2; ++j) {
                double s = 0;
                for (int k = 0; k < 3; ++k) s += h.m[i][k] * Bx[k][j];
                Mx[i][j] = -s;
            }
    } else {
        V3 Bu{Bx[0][0] * d0 + Bx[0][1] * d1, Bx[1][0] * d0 + Bx[1][1] * d1, Bx[2][0] * d0 + Bx[2][1] * d1};
        Q4 e = mtk_exp(Bu, 0.0);                           // scalar(1/2) -> 0
        M3 Rm = qmat(e);
        M3 At = mT(A_matrix(Bu));
        M3 t = mmul(mmul(Rm, h), At);
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 2; ++j) {
                double s = 0;
                for (int k = 0; k < 3; ++k) s += t.m[i][k] * Bx[k][j];
                Mx[i][j] = -s;
            }
    }
}

// state x[26]: pos 0-2, rot 3-6 (xyzw), offR 7-10, offT 11-13, vel 14-16, bg 17-19, ba 20-22, grav 23-25
struct State {
    V3 pos; Q4 rot; Q4 offR; V3 offT; V3 vel, bg, ba, grav;
};
inline State st_from(const double* x) {
    State s;
    s.pos = {x[0], x[1], x[2]}; s.rot = {x[3], x[4], x[5], x[6]}; s.offR = {x[7], x[8], x[9], x[10]};
    s.offT = {x[11], x[12], x[13]}; s.vel = {x[14], x[15], x[16]}; s.bg = {x[17], x[18], x[19]};
    s.ba = {x[20], x[21], x[22]}; s.grav = {x[23], x[24], x[25]};
    return s;
}
inline void st_to(const State& s, double* x) {
    x[0] = s.pos.x; x[1] = s.pos.y; x[2] = s.pos.z;
    x[3] = s.rot.x; x[4] = s.rot.y; x[5] = s.rot.z; x[6] = s.rot.w;
    x[7] = s.offR.x; x[8] = s.offR.y; x[9] = s.offR.z; x[10] = s.offR.w;
    x[11] = s.offT.x; x[12] = s.offT.y; x[13] = s.offT.z;
    x[14] = s.vel.x; x[15] = s.vel.y; x[16] = s.vel.z;
    x[17] = s.bg.x; x[18] = s.bg.y; x[19] = s.bg.z;
    x[20] = s.ba.x; x[21] = s.ba.y; x[22] = s.ba.z;
    x[23] = s.grav.x; x[24] = s.grav.y; x[25] = s.grav.z;
}
inline void st_pose14(const State& s, double* p) {
    p[0] = s.pos.x; p[1] = s.pos.y; p[2] = s.pos.z;
    p[3] = s.rot.x; p[4] = s.rot.y; p[5] = s.rot.z; p[6] = s.rot.w;
    p[7] = s.offR.x; p[8] = s.offR.y; p[9] = s.offR.z; p[10] = s.offR.w;
    p[11] = s.offT.x; p[12] = s.offT.y; p[13] = s.offT.z;
}
// state boxplus: vect += d (vect.hpp:117-119); SO3 q <- q * exp(d) (SOn.hpp:233-236, :284-288)
inline void st_boxplus(State& s, const double* d) {
    s.pos = s.pos + V3{d[0], d[1], d[2]};
    s.rot = qmul(s.rot, mtk_exp({d[3], d[4], d[5]}, 0.5));
    s.offR = qmul(s.offR, mtk_exp({d[6], d[7], d[8]}, 0.5));
    s.offT = s.offT + V3{d[9], d[10], d[11]};
    s.vel = s.vel + V3{d[12], d[13], d[14]};
    s.bg = s.bg + V3{d[15], d[16], d[17]};
    s.ba = s.ba + V3{d[18], d[19], d[20]};
    s.grav = S2_boxplus(s.grav, d[21], d[22]);
}
// state boxminus: res = a [-] b
inline void st_boxminus(const State& a, const State& b, double* r) {
    V3 t = a.pos - b.pos; r[0] = t.x; r[1] = t.y; r[2] = t.z;
    t = so3_log(qmul(conj(b.rot), a.rot)); r[3] = t.x; r[4] = t.y; r[5] = t.z;
    t = so3_log(qmul(conj(b.offR), a.offR)); r[6] = t.x; r[7] = t.y; r[8] = t.z;
    t = a.offT - b.offT; r[9] = t.x; r[10] = t.y; r[11] = t.z;
    t = a.vel - b.vel; r[12] = t.x; r[13] = t.y; r[14] = t.z;
    t = a.bg - b.bg; r[15] = t.x; r[16] = t.y; r[17] = t.z;
    t = a.ba - b.ba; r[18] = t.x; r[19] = t.y; r[20] = t.z;
    S2_boxminus(a.grav, b.grav, r + 21);
}

// ---- dense helpers (row-major) ----
const int N = 23;
// Eigen's inverse() for a 23x23 goes through PartialPivLU; restated as LU with partial pivoting
// followed by the solve of the identity.
bool inverse_n(const double* Ain, double* inv, int n) {
    std::vector<double> a(Ain, Ain + n * n);
    std::vector<int> piv(n);
    for (int i = 0; i < n; ++i) piv[i] = i;
    for (int k = 0; k < n; ++k) {
        int p = k; double best = std::fabs(a[k * n + k]);
        for (int i = k + 1; i < n; ++i) { double v = std::fabs(a[i * n + k]); if (v > best) { best = v; p = i; } }
        if (best == 0.0) return false;
        if (p != k) { for (int j = 0; j < n; ++j) std::swap(a[k * n + j], a[p * n + j]); std::swap(piv[k], piv[p]); }
        for (int i = k + 1; i < n; ++i) {
            a[i * n + k] /= a[k * n + k];
            double l = a[i * n + k];
            for (int j = k + 1; j < n; ++j) a[i * n + j] -= l * a[k * n + j];
        }
    }
    for (int c = 0; c < n; ++c) {
        std::vector<double> y(n);
        for (int i = 0; i < n; ++i) {
            double s = (piv[i] == c) ? 1.0 : 0.0;
            for (int j = 0; j < i; ++j) s -= a[i * n + j] * y[j];
            y[i] = s;
        }
        for (int i = n - 1; i >= 0; --i) {
            double s = y[i];
            for (int j = i + 1; j < n; ++j) s -= a[i * n + j] * inv[j * n + c];
            inv[i * n + c] = s / a[i * n + i];
        }
    }
    return true;
}

}  // namespace

// =====================================================================================
//  C ABI
// =====================================================================================
extern "C" {

void orc_set_external_knn(void* fn, void* handle) { g_ext_knn = (ext_knn_fn)fn; g_ext_handle = handle; }

void* orc_map_create(float downsample_size) { OMap* m = new OMap; m->ds = downsample_size; return m; }
void orc_map_destroy(void* m) { delete static_cast<OMap*>(m); }

// Build (ikd_Tree.cpp:353-364): replaces the tree by the given cloud, no down-sampling
void orc_map_build(void* mp, const float* xyz, int n) {
    OMap* m = static_cast<OMap*>(mp);
    m->clear();
    m->pts.reserve(n);
    for (int i = 0; i < n; ++i) m->push({xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2]});
}

// Add_Points (ikd_Tree.cpp:414-489), sequential: every insert sees the previous ones
int orc_map_add(void* mp, const float* xyz, int n, int downsample_on) {
    OMap* m = static_cast<OMap*>(mp);
    int counter = 0;
    const float ds = m->ds;
    for (int i = 0; i < n; ++i) {
        P3 p{xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2]};
        if (!downsample_on) { m->push(p); continue; }
        float bmin[3], bmax[3];
        bmin[0] = std::floor(p.x / ds) * ds; bmax[0] = bmin[0] + ds;           // :424-429
        bmin[1] = std::floor(p.y / ds) * ds; bmax[1] = bmin[1] + ds;
        bmin[2] = std::floor(p.z / ds) * ds; bmax[2] = bmin[2] + ds;
        P3 mid;                                                                 // :430-432
        mid.x = (float)(bmin[0] + (bmax[0] - bmin[0]) / 2.0);
        mid.y = (float)(bmin[1] + (bmax[1] - bmin[1]) / 2.0);
        mid.z = (float)(bmin[2] + (bmax[2] - bmin[2]) / 2.0);
        // Search_by_range (:1015-1045): live points with min <= p < max on every axis.  Candidates
        // come from the 27 voxel buckets around the point's own cell (fp32 rounding can put a
        // box member into a neighbouring bucket).
        std::vector<int> in_box;
        int cx, cy, cz; m->cell_of(p, cx, cy, cz);
        for (int dx = -1; dx <= 1; ++dx) for (int dy = -1; dy <= 1; ++dy) for (int dz = -1; dz <= 1; ++dz) {
            auto it = m->vox.find(OMap::vkey(cx + dx, cy + dy, cz + dz));
            if (it == m->vox.end()) continue;
            for (int id : it->second) {
                const P3& s = m->pts[id];
                if (bmin[0] <= s.x && bmax[0] > s.x && bmin[1] <= s.y && bmax[1] > s.y &&
                    bmin[2] <= s.z && bmax[2] > s.z) in_box.push_back(id);
            }
        }
        std::sort(in_box.begin(), in_box.end());            // deterministic visiting order
        float min_dist = calc_dist(p, mid);                 // :435
        P3 result = p;
        for (int id : in_box) {                             // :437-443
            float t = calc_dist(m->pts[id], mid);
            if (t < min_dist) { min_dist = t; result = m->pts[id]; }
        }
        if (in_box.size() > 1 || same_point(p, result)) {   // :445
            for (int id : in_box) m->kill(id);              // Delete_by_range(..., is_downsample)
            m->push(result);                                // Add_by_point
            ++counter;
        }
    }
    return counter;
}

// Delete_Point_Boxes (ikd_Tree.cpp:536-556) / Delete_by_range (:648-710): half-open boxes
int orc_map_delete_boxes(void* mp, const float* mm, int nb) {
    OMap* m = static_cast<OMap*>(mp);
    int counter = 0;
    for (int b = 0; b < nb; ++b) {
        const float* lo = mm + 6 * b; const float* hi = lo + 3;
        for (int i = 0; i < (int)m->pts.size(); ++i) {
            if (!m->alive[i]) continue;
            const P3& s = m->pts[i];
            if (lo[0] <= s.x && hi[0] > s.x && lo[1] <= s.y && hi[1] > s.y && lo[2] <= s.z && hi[2] > s.z) {
                m->kill(i); ++counter;
            }
        }
    }
    return counter;
}
int orc_map_validnum(void* mp) { return static_cast<OMap*>(mp)->live; }
int orc_map_flatten(void* mp, float* out, int cap) {
    OMap* m = static_cast<OMap*>(mp);
    int c = 0;
    for (size_t i = 0; i < m->pts.size(); ++i) {
        if (!m->alive[i]) continue;
        if (c < cap) { out[3 * c] = m->pts[i].x; out[3 * c + 1] = m->pts[i].y; out[3 * c + 2] = m->pts[i].z; }
        ++c;
    }
    return c;
}

// Nearest_Search (ikd_Tree.cpp:367-397)
double orc_map_knn(void* mp, const float* q, int nq, int k, float* out_xyz, float* out_d,
                   int* out_cnt, int method) {
    OMap* m = static_cast<OMap*>(mp);
    if (method != 0) m->ensure_tree();
    auto t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < nq; ++i) {
        TopK h(k);
        m->knn({q[3 * i], q[3 * i + 1], q[3 * i + 2]}, k, h, method);
        out_cnt[i] = h.n;
        for (int j = 0; j < h.n; ++j) {
            out_xyz[(i * k + j) * 3 + 0] = h.c[j].p.x;
            out_xyz[(i * k + j) * 3 + 1] = h.c[j].p.y;
            out_xyz[(i * k + j) * 3 + 2] = h.c[j].p.z;
            out_d[i * k + j] = h.c[j].d;
        }
    }
    return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
}

// test switch: summation order of every reduction inside esti_plane (see sum_terms).  0 = sequential (the order the GPU
// path is bit-equal to), 1 = SSE2 packet order, 2 = SSE3 hadd order, 3 = random permutation + association from `seed`
void orc_set_sum_order(int mode, unsigned long long seed) {
    g_sum_mode = mode;
    g_sum_rng = seed * 0x9e3779b97f4a7c15ull + 0x632be59bd9b4e019ull;
    if (g_sum_rng == 0) g_sum_rng = 1;
}

int orc_esti_plane(const float* p, float threshold, float* pabcd) {
    P3 pt[5];
    for (int j = 0; j < 5; ++j) pt[j] = {p[3 * j], p[3 * j + 1], p[3 * j + 2]};
    return esti_plane_f(pabcd, pt, threshold) ? 1 : 0;
}

// Decision-flip audit (bench.py's parity block): per scan point with five gated neighbours, the distances of the
// fp32 decisions of the path to their thresholds -- margin_plane = min_j | |n.p_j + d| - 0.1 | (common_lib.h:530),
// margin_s = | s - 0.9 | for points whose plane passed (laserMapping.cpp:1927-1930); NaN where the test is not reached.
void orc_decision_margins(const float* nbr_xyz, const float* nbr_d, const int* nbr_cnt, const float* world,
                          const float* body, int n, float* margin_plane, float* margin_s) {
    const int K = 5;
    const float nanv = std::numeric_limits<float>::quiet_NaN();
    for (int i = 0; i < n; ++i) {
        margin_plane[i] = nanv; margin_s[i] = nanv;
        if (nbr_cnt[i] < K || nbr_d[i * K + K - 1] > 5) continue;                        // :1911
        P3 pn[5];
        for (int j = 0; j < K; ++j) pn[j] = {nbr_xyz[(i * K + j) * 3], nbr_xyz[(i * K + j) * 3 + 1], nbr_xyz[(i * K + j) * 3 + 2]};
        float pabcd[4];
        const bool ok = esti_plane_f(pabcd, pn, 0.1f);
        float mp = std::numeric_limits<float>::infinity();
        for (int j = 0; j < K; ++j) {
            const float r = std::fabs(pabcd[0] * pn[j].x + pabcd[1] * pn[j].y + pabcd[2] * pn[j].z + pabcd[3]);
            mp = std::fmin(mp, std::fabs(r - 0.1f));
        }
        margin_plane[i] = mp;
        if (!ok) continue;
        const float pd2 = pabcd[0] * world[3 * i] + pabcd[1] * world[3 * i + 1] + pabcd[2] * world[3 * i + 2] + pabcd[3];
        V3 pb{body[3 * i], body[3 * i + 1], body[3 * i + 2]};
        const float sc = (float)(1 - 0.9 * std::fabs(pd2) / std::sqrt(norm(pb)));
        margin_s[i] = std::fabs(sc - 0.9f);
    }
}

// laserMapping.cpp:1894-1898
void orc_body_to_world(const double* s14, const float* body, int n, float* world) {
    Pose s = pose_from(s14);
    for (int i = 0; i < n; ++i) {
        V3 pb{body[3 * i], body[3 * i + 1], body[3 * i + 2]};
        V3 pg = qrot(s.rot, qrot(s.offR, pb) + s.offT) + s.pos;
        world[3 * i] = (float)pg.x; world[3 * i + 1] = (float)pg.y; world[3 * i + 2] = (float)pg.z;
    }
}

// h_share_model (src/laserMapping.cpp:1876-2004)
int orc_h_share_model(void* map, const double* s14, const float* body, int n, int converge,
                      int extrinsic_est_en, int knn_method, float* nbr_xyz, float* nbr_d,
                      int* nbr_cnt, uint8_t* selected, float* world, float* normvec,
                      float* res_last, double* h_x, double* h, int* m_out, double* res_mean) {
    OMap* m = static_cast<OMap*>(map);
    Pose s = pose_from(s14);
    const int K = 5;                                       // NUM_MATCH_POINTS common_lib.h:149
    if (converge && knn_method == 1) m->ensure_tree();
    for (int i = 0; i < n; ++i) {                          // :1888-1940
        V3 pb{body[3 * i], body[3 * i + 1], body[3 * i + 2]};
        V3 pg = qrot(s.rot, qrot(s.offR, pb) + s.offT) + s.pos;     // :1895
        P3 pw{(float)pg.x, (float)pg.y, (float)pg.z};
        world[3 * i] = pw.x; world[3 * i + 1] = pw.y; world[3 * i + 2] = pw.z;
        if (converge && knn_method == 2 && g_ext_knn) {    // the reference's own Nearest_Search
            const float qq[3] = {pw.x, pw.y, pw.z};
            g_ext_knn(g_ext_handle, qq, 1, K, nbr_xyz + (size_t)i * K * 3, nbr_d + (size_t)i * K, nbr_cnt + i);
            selected[i] = nbr_cnt[i] < K ? 0 : (nbr_d[i * K + K - 1] > 5 ? 0 : 1);
        } else if (converge) {                             // :1906-1912
            TopK hp(K);
            m->knn(pw, K, hp, knn_method);
            nbr_cnt[i] = hp.n;
            for (int j = 0; j < hp.n; ++j) {
                nbr_xyz[(i * K + j) * 3] = hp.c[j].p.x; nbr_xyz[(i * K + j) * 3 + 1] = hp.c[j].p.y;
                nbr_xyz[(i * K + j) * 3 + 2] = hp.c[j].p.z; nbr_d[i * K + j] = hp.c[j].d;
            }
            selected[i] = hp.n < K ? 0 : (nbr_d[i * K + K - 1] > 5 ? 0 : 1);
        }
        if (!selected[i]) continue;                        // :1916
        selected[i] = 0;                                   // :1920
        P3 pn[5];
        for (int j = 0; j < K; ++j) pn[j] = {nbr_xyz[(i * K + j) * 3], nbr_xyz[(i * K + j) * 3 + 1], nbr_xyz[(i * K + j) * 3 + 2]};
        float pabcd[4];
        if (esti_plane_f(pabcd, pn, 0.1f)) {               // :1922
            float pd2 = pabcd[0] * pw.x + pabcd[1] * pw.y + pabcd[2] * pw.z + pabcd[3];   // :1925
            float sc = (float)(1 - 0.9 * std::fabs(pd2) / std::sqrt(norm(pb)));          // :1927
            if (sc > 0.9) {                                // :1930
                selected[i] = 1;
                normvec[4 * i] = pabcd[0]; normvec[4 * i + 1] = pabcd[1]; normvec[4 * i + 2] = pabcd[2];
                normvec[4 * i + 3] = pd2;
                res_last[i] = std::fabs(pd2);
            }
        }
    }
    int mcount = 0;                                        // :1942-1954
    double total_residual = 0.0;
    std::vector<int> src;
    src.reserve(n);
    for (int i = 0; i < n; ++i) {
        if (selected[i]) { src.push_back(i); total_residual += res_last[i]; ++mcount; }
    }
    *m_out = mcount;
    if (mcount < 1) { *res_mean = 0.0; return 0; }         // :1956-1961
    *res_mean = total_residual / mcount;                   // :1963
    for (int r = 0; r < mcount; ++r) {                     // :1970-2002
        int i = src[r];
        V3 pbe{body[3 * i], body[3 * i + 1], body[3 * i + 2]};
        M3 be_cross = hat(pbe);
        V3 pthis = qrot(s.offR, pbe) + s.offT;             // :1976
        M3 pcross = hat(pthis);
        V3 nv{normvec[4 * i], normvec[4 * i + 1], normvec[4 * i + 2]};
        V3 C = qrot(conj(s.rot), nv);                      // :1985
        V3 A = mvec(pcross, C);                            // :1986
        double* row = h_x + (size_t)r * 12;
        row[0] = nv.x; row[1] = nv.y; row[2] = nv.z;
        row[3] = A.x; row[4] = A.y; row[5] = A.z;
        if (extrinsic_est_en) {                            // :1992-1993
            V3 B = mvec(mmul(be_cross, qmat(conj(s.offR))), C);
            row[6] = B.x; row[7] = B.y; row[8] = B.z; row[9] = C.x; row[10] = C.y; row[11] = C.z;
        } else {                                           // :1997
            for (int c = 6; c < 12; ++c) row[c] = 0.0;
        }
        h[r] = -(double)normvec[4 * i + 3];                // :2001
    }
    return 1;
}

void orc_state_boxplus(double* x, const double* dx) { State s = st_from(x); st_boxplus(s, dx); st_to(s, x); }
void orc_state_boxminus(const double* x, const double* y, double* dx) { st_boxminus(st_from(x), st_from(y), dx); }

void orc_hth(const double* h_x, const double* h, int m, double* HTH, double* HTh) {
    for (int a = 0; a < 12; ++a) {
        for (int b = 0; b < 12; ++b) {
            double s = 0;
            for (int r = 0; r < m; ++r) s += h_x[r * 12 + a] * h_x[r * 12 + b];
            HTH[a * 12 + b] = s;
        }
        double s = 0;
        for (int r = 0; r < m; ++r) s += h_x[r * 12 + a] * h[r];
        HTh[a] = s;
    }
}

// update_iterated_dyn_share_modified (esekfom.hpp:1620-1938)
void orc_update_iterated(void* map, double* x26, double* P, const float* body, int n, double R,
                         int max_iter, const double* limit, int extrinsic_est_en, int knn_method,
                         float* nbr_xyz, float* nbr_d, int* nbr_cnt, uint8_t* selected,
                         double* log, int* n_calls) {
    State x = st_from(x26);
    const State x_prop = x;                                // :1628
    std::vector<double> P_prop(P, P + N * N);              // :1629
    std::vector<double> Pm(P, P + N * N);                  // P_
    std::vector<float> world((size_t)n * 3), normvec((size_t)n * 4), res_last(n);
    std::vector<double> h_x((size_t)n * 12), h(n);
    bool converge = true;                                  // :1624
    int t = 0, calls = 0;
    double K_h[N], K_x[N * N];
    double dx_new[N];
    for (int i = 0; i < N; ++i) dx_new[i] = 0;
    const int so3_idx[2] = {3, 6};                         // build_SO3_state (esekfom.hpp:252-254)
    const int s2_idx = 21;

    for (int it = -1; it < max_iter; ++it) {               // :1636
        double pose[14];
        st_pose14(x, pose);
        int m = 0; double res_mean = 0;
        int valid = orc_h_share_model(map, pose, body, n, converge ? 1 : 0, extrinsic_est_en, knn_method,
                                      nbr_xyz, nbr_d, nbr_cnt, selected, world.data(), normvec.data(),
                                      res_last.data(), h_x.data(), h.data(), &m, &res_mean);
        if (log) { log[4 * calls] = converge; log[4 * calls + 1] = valid; log[4 * calls + 2] = m; log[4 * calls + 3] = res_mean; }
        ++calls;
        if (!valid) continue;                              // :1641-1644

        double dx[N];
        st_boxminus(x, x_prop, dx);                        // :1655
        for (int i = 0; i < N; ++i) dx_new[i] = dx[i];
        Pm = P_prop;                                       // :1660

        for (int si = 0; si < 2; ++si) {                   // :1665-1680
            int idx = so3_idx[si];
            M3 J = mT(A_matrix({dx[idx], dx[idx + 1], dx[idx + 2]}));
            V3 seg = mvec(J, {dx_new[idx], dx_new[idx + 1], dx_new[idx + 2]});
            dx_new[idx] = seg.x; dx_new[idx + 1] = seg.y; dx_new[idx + 2] = seg.z;
            for (int c = 0; c < N; ++c) {
                V3 col = mvec(J, {Pm[idx * N + c], Pm[(idx + 1) * N + c], Pm[(idx + 2) * N + c]});
                Pm[idx * N + c] = col.x; Pm[(idx + 1) * N + c] = col.y; Pm[(idx + 2) * N + c] = col.z;
            }
            for (int r = 0; r < N; ++r) {
                V3 row = mvec(J, {Pm[r * N + idx], Pm[r * N + idx + 1], Pm[r * N + idx + 2]});  // row * J^T
                Pm[r * N + idx] = row.x; Pm[r * N + idx + 1] = row.y; Pm[r * N + idx + 2] = row.z;
            }
        }
        {                                                  // :1684-1703
            int idx = s2_idx;
            double Nx[2][3], Mx[3][2], J2[2][2];
            S2_Nx_yy(x.grav, Nx);
            S2_Mx(x_prop.grav, dx[idx], dx[idx + 1], Mx);
            for (int a = 0; a < 2; ++a) for (int b = 0; b < 2; ++b)
                J2[a][b] = (Nx[a][0] * Mx[0][b] + Nx[a][1] * Mx[1][b]) + Nx[a][2] * Mx[2][b];
            double d0 = J2[0][0] * dx_new[idx] + J2[0][1] * dx_new[idx + 1];
            double d1 = J2[1][0] * dx_new[idx] + J2[1][1] * dx_new[idx + 1];
            dx_new[idx] = d0; dx_new[idx + 1] = d1;
            for (int c = 0; c < N; ++c) {
                double a0 = Pm[idx * N + c], a1 = Pm[(idx + 1) * N + c];
                Pm[idx * N + c] = J2[0][0] * a0 + J2[0][1] * a1;
                Pm[(idx + 1) * N + c] = J2[1][0] * a0 + J2[1][1] * a1;
            }
            for (int r = 0; r < N; ++r) {
                double a0 = Pm[r * N + idx], a1 = Pm[r * N + idx + 1];
                Pm[r * N + idx] = J2[0][0] * a0 + J2[0][1] * a1;
                Pm[r * N + idx + 1] = J2[1][0] * a0 + J2[1][1] * a1;
            }
        }

        if (N > m) {                                       // dense-K branch :1720-1750
            // K = P H^T (H P H^T / R + I)^-1 / R, H zero-padded to 23 columns
            std::vector<double> H((size_t)m * N, 0.0);
            for (int r = 0; r < m; ++r) for (int c = 0; c < 12; ++c) H[r * N + c] = h_x[r * 12 + c];
            std::vector<double> PHt((size_t)N * m), S((size_t)m * m), Sinv((size_t)m * m), K((size_t)N * m);
            for (int a = 0; a < N; ++a) for (int r = 0; r < m; ++r) {
                double s = 0; for (int c = 0; c < N; ++c) s += Pm[a * N + c] * H[r * N + c];
                PHt[a * m + r] = s;
            }
            for (int r = 0; r < m; ++r) for (int q = 0; q < m; ++q) {
                double s = 0; for (int c = 0; c < N; ++c) s += H[r * N + c] * PHt[c * m + q];
                S[r * m + q] = s / R + (r == q ? 1.0 : 0.0);
            }
            inverse_n(S.data(), Sinv.data(), m);
            for (int a = 0; a < N; ++a) for (int q = 0; q < m; ++q) {
                double s = 0; for (int r = 0; r < m; ++r) s += PHt[a * m + r] * Sinv[r * m + q];
                K[a * m + q] = s / R;
            }
            for (int a = 0; a < N; ++a) {
                double s = 0; for (int r = 0; r < m; ++r) s += K[a * m + r] * h[r];
                K_h[a] = s;
                for (int c = 0; c < N; ++c) {
                    double v = 0; for (int r = 0; r < m; ++r) v += K[a * m + r] * H[r * N + c];
                    K_x[a * N + c] = v;
                }
            }
        } else {                                           // information form :1788-1815
            std::vector<double> PR(N * N), Ptemp(N * N), Pinv(N * N);
            for (int i = 0; i < N * N; ++i) PR[i] = Pm[i] / R;
            inverse_n(PR.data(), Ptemp.data(), N);         // (P_/R).inverse()
            double HTH[144], HTh[12];
            orc_hth(h_x.data(), h.data(), m, HTH, HTh);    // :1790 (Eigen GEMM; summation order unknowable)
            for (int a = 0; a < 12; ++a) for (int b = 0; b < 12; ++b) Ptemp[a * N + b] += HTH[a * 12 + b];
            inverse_n(Ptemp.data(), Pinv.data(), N);       // :1808
            for (int a = 0; a < N; ++a) {                  // K_h = P_inv[:,0:12] * (h_x^T * h) :1810
                double s = 0; for (int c = 0; c < 12; ++c) s += Pinv[a * N + c] * HTh[c];
                K_h[a] = s;
            }
            for (int i = 0; i < N * N; ++i) K_x[i] = 0;    // :1814-1815
            for (int a = 0; a < N; ++a) for (int b = 0; b < 12; ++b) {
                double s = 0; for (int c = 0; c < 12; ++c) s += Pinv[a * N + c] * HTH[c * 12 + b];
                K_x[a * N + b] = s;
            }
        }
        double dx_[N];                                     // :1821
        for (int a = 0; a < N; ++a) {
            double s = 0;
            for (int c = 0; c < N; ++c) s += (K_x[a * N + c] - (a == c ? 1.0 : 0.0)) * dx_new[c];
            dx_[a] = K_h[a] + s;
        }
        st_boxplus(x, dx_);                                // :1823
        converge = true;                                   // :1824-1832
        for (int a = 0; a < N; ++a) if (std::fabs(dx_[a]) > limit[a]) { converge = false; break; }
        if (converge) ++t;
        if (!t && it == max_iter - 2) converge = true;     // :1835-1838

        if (t > 1 || it == max_iter - 1) {                 // :1841-1935
            std::vector<double> L(Pm);
            for (int si = 0; si < 2; ++si) {
                int idx = so3_idx[si];
                M3 J = mT(A_matrix({dx_[idx], dx_[idx + 1], dx_[idx + 2]}));
                for (int c = 0; c < N; ++c) {
                    V3 col = mvec(J, {Pm[idx * N + c], Pm[(idx + 1) * N + c], Pm[(idx + 2) * N + c]});
                    L[idx * N + c] = col.x; L[(idx + 1) * N + c] = col.y; L[(idx + 2) * N + c] = col.z;
                }
                for (int c = 0; c < 12; ++c) {
                    V3 col = mvec(J, {K_x[idx * N + c], K_x[(idx + 1) * N + c], K_x[(idx + 2) * N + c]});
                    K_x[idx * N + c] = col.x; K_x[(idx + 1) * N + c] = col.y; K_x[(idx + 2) * N + c] = col.z;
                }
                for (int r = 0; r < N; ++r) {
                    V3 row = mvec(J, {L[r * N + idx], L[r * N + idx + 1], L[r * N + idx + 2]});
                    L[r * N + idx] = row.x; L[r * N + idx + 1] = row.y; L[r * N + idx + 2] = row.z;
                    V3 prow = mvec(J, {Pm[r * N + idx], Pm[r * N + idx + 1], Pm[r * N + idx + 2]});
                    Pm[r * N + idx] = prow.x; Pm[r * N + idx + 1] = prow.y; Pm[r * N + idx + 2] = prow.z;
                }
            }
            {
                int idx = s2_idx;
                double Nx[2][3], Mx[3][2], J2[2][2];
                S2_Nx_yy(x.grav, Nx);
                S2_Mx(x_prop.grav, dx_[idx], dx_[idx + 1], Mx);
                for (int a = 0; a < 2; ++a) for (int b = 0; b < 2; ++b)
                    J2[a][b] = (Nx[a][0] * Mx[0][b] + Nx[a][1] * Mx[1][b]) + Nx[a][2] * Mx[2][b];
                for (int c = 0; c < N; ++c) {
                    double a0 = Pm[idx * N + c], a1 = Pm[(idx + 1) * N + c];
                    L[idx * N + c] = J2[0][0] * a0 + J2[0][1] * a1;
                    L[(idx + 1) * N + c] = J2[1][0] * a0 + J2[1][1] * a1;
                }
                for (int c = 0; c < 12; ++c) {
                    double a0 = K_x[idx * N + c], a1 = K_x[(idx + 1) * N + c];
                    K_x[idx * N + c] = J2[0][0] * a0 + J2[0][1] * a1;
                    K_x[(idx + 1) * N + c] = J2[1][0] * a0 + J2[1][1] * a1;
                }
                for (int r = 0; r < N; ++r) {
                    double a0 = L[r * N + idx], a1 = L[r * N + idx + 1];
                    L[r * N + idx] = J2[0][0] * a0 + J2[0][1] * a1;
                    L[r * N + idx + 1] = J2[1][0] * a0 + J2[1][1] * a1;
                    double b0 = Pm[r * N + idx], b1 = Pm[r * N + idx + 1];
                    Pm[r * N + idx] = J2[0][0] * b0 + J2[0][1] * b1;
                    Pm[r * N + idx + 1] = J2[1][0] * b0 + J2[1][1] * b1;
                }
            }
            // P_ = L_ - K_x[:,0:12] * P_[0:12,:]   :1931
            std::vector<double> Pout(N * N);
            for (int a = 0; a < N; ++a) for (int c = 0; c < N; ++c) {
                double s = 0; for (int b = 0; b < 12; ++b) s += K_x[a * N + b] * Pm[b * N + c];
                Pout[a * N + c] = L[a * N + c] - s;
            }
            Pm = Pout;
            break;
        }
    }
    st_to(x, x26);
    std::memcpy(P, Pm.data(), sizeof(double) * N * N);
    if (n_calls) *n_calls = calls;
}

// map_incremental (src/laserMapping.cpp:1440-1496), classification part
void orc_map_incremental_classify(const float* world, int n, const float* nbr_xyz, const int* nbr_cnt,
                                  double fs, int flg_EKF_inited, uint8_t* cls) {   // fs is a double: laserMapping.cpp:56
    const int K = 5;
    for (int i = 0; i < n; ++i) {
        P3 pw{world[3 * i], world[3 * i + 1], world[3 * i + 2]};
        if (nbr_cnt[i] > 0 && flg_EKF_inited) {            // :1453
            P3 mid;                                        // :1459-1461 (double arithmetic, stored fp32)
            mid.x = (float)(std::floor(pw.x / fs) * fs + 0.5 * fs);
            mid.y = (float)(std::floor(pw.y / fs) * fs + 0.5 * fs);
            mid.z = (float)(std::floor(pw.z / fs) * fs + 0.5 * fs);
            float dist = calc_dist(pw, mid);               // :1462 (common_lib.h:495-498, same fp32 form)
            P3 n0{nbr_xyz[(i * K) * 3], nbr_xyz[(i * K) * 3 + 1], nbr_xyz[(i * K) * 3 + 2]};
            if (std::fabs(n0.x - mid.x) > 0.5 * fs && std::fabs(n0.y - mid.y) > 0.5 * fs &&
                std::fabs(n0.z - mid.z) > 0.5 * fs) {      // :1465
                cls[i] = 2; continue;
            }
            bool need_add = true;
            for (int j = 0; j < K; ++j) {                  // :1472-1481
                if (nbr_cnt[i] < K) break;
                P3 pj{nbr_xyz[(i * K + j) * 3], nbr_xyz[(i * K + j) * 3 + 1], nbr_xyz[(i * K + j) * 3 + 2]};
                if (calc_dist(pj, mid) < dist) { need_add = false; break; }
            }
            cls[i] = need_add ? 1 : 0;
        } else {
            cls[i] = 1;                                    // :1487
        }
    }
}

// ImuProcess::UndistortPcl, point side (src/IMU_Processing.hpp:250-251 sort by time, :334-386 backward
// compensation).  xyzt = n x (x, y, z, curvature = time offset in ms); poses = K x Pose6D (22 doubles:
// offset_time, acc, gyr, vel, pos, rot row-major; msg/Pose6D.msg); end14 = imu_state after propagation.
// Output: the time-sorted (stable), compensated cloud.  The inner loop assigns a point to the interval
// [head, tail] with the largest head.offset_time < t (the reference walks points and poses backwards);
// the reference's re-processing of the very first point when it lies beyond the second pose (the
// `break` at :384 leaves it_pcl at begin) is not reproduced.
void orc_undistort(const float* xyzt, int n, const double* poses, int K, const double* end14, float* out_xyz,
                   float* out_t) {
    std::vector<int> order(n);
    for (int i = 0; i < n; ++i) order[i] = i;
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return xyzt[4 * a + 3] < xyzt[4 * b + 3]; });
    Pose s = pose_from(end14);
    for (int o = 0; o < n; ++o) {
        const float* p = xyzt + 4 * order[o];
        float rx = p[0], ry = p[1], rz = p[2];
        const double t = p[3] / double(1000);
        int k = -1;
        for (int j = 0; j + 1 < K; ++j) if (poses[22 * j] < t) k = j;
        if (k >= 0) {
            const double* head = poses + 22 * k; const double* tail = head + 22;
            const double dt = t - head[0];
            V3 acc{tail[1], tail[2], tail[3]}, gyr{tail[4], tail[5], tail[6]};
            V3 vel{head[7], head[8], head[9]}, pos{head[10], head[11], head[12]};
            M3 Rm; for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) Rm.m[i][j] = head[13 + 3 * i + j];
            // Exp(ang_vel, dt)  include/math_tools.h:39-61
            M3 E = {{{1, 0, 0}, {0, 1, 0}, {0, 0, 1}}};
            const double nrm = norm(gyr);
            if (nrm > 0.0000001) {
                V3 ax{gyr.x / nrm, gyr.y / nrm, gyr.z / nrm};
                M3 Kx = hat(ax), KK = mmul(Kx, Kx);
                const double ang = nrm * dt;
                const double sn = std::sin(ang), cs = 1.0 - std::cos(ang);
                for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) E.m[i][j] = (E.m[i][j] + sn * Kx.m[i][j]) + cs * KK.m[i][j];
            }
            M3 Ri = mmul(Rm, E);                                                           // :355
            V3 Pi{p[0], p[1], p[2]};
            V3 Tei{((pos.x + vel.x * dt) + ((0.5 * acc.x) * dt) * dt) - s.pos.x,              // :359
                   ((pos.y + vel.y * dt) + ((0.5 * acc.y) * dt) * dt) - s.pos.y,
                   ((pos.z + vel.z * dt) + ((0.5 * acc.z) * dt) * dt) - s.pos.z};
            V3 in = mvec(Ri, qrot(s.offR, Pi) + s.offT) + Tei;
            V3 b = qrot(conj(s.rot), in) - s.offT;
            V3 c = qrot(conj(s.offR), b);                                                  // :372-374
            rx = (float)c.x; ry = (float)c.y; rz = (float)c.z;
        }
        out_xyz[3 * o] = rx; out_xyz[3 * o + 1] = ry; out_xyz[3 * o + 2] = rz;
        if (out_t) out_t[o] = p[3];
    }
}

// pcl::VoxelGrid<PointType>::applyFilter (PCL 1.10 filters/impl/voxel_grid.hpp; PCL is a third-party,
// un-vendored dependency of the reference: laserMapping.cpp:112, 2135, 2322-2323), xyz only:
// getMinMax3D -> min_b/div_b -> leaf index per point -> sort by leaf (PCL: std::sort on the index, i.e.
// unspecified order among equal leaves; here stable = input order) -> centroid (CentroidPoint: fp32 sums
// in that order, divided by the count) -> output in ascending leaf order.  Returns the number of leaves
// (or -1 for the "leaf size too small" overflow, where PCL returns the input unfiltered).
int orc_voxel_grid(const float* xyz, int n, float leaf, float* out_xyz, int cap) {
    const float inv = 1.0f / leaf;
    float mn[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, mx[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
    int finite = 0;
    for (int i = 0; i < n; ++i) {
        const float* p = xyz + 3 * i;
        if (!std::isfinite(p[0]) || !std::isfinite(p[1]) || !std::isfinite(p[2])) continue;
        for (int a = 0; a < 3; ++a) { mn[a] = std::min(mn[a], p[a]); mx[a] = std::max(mx[a], p[a]); }
        ++finite;
    }
    if (!finite) return 0;
    int min_b[3], max_b[3]; long long div[3];
    for (int a = 0; a < 3; ++a) {
        min_b[a] = static_cast<int>(std::floor(mn[a] * inv));
        max_b[a] = static_cast<int>(std::floor(mx[a] * inv));
        div[a] = (long long)max_b[a] - min_b[a] + 1;
    }
    if (div[0] * div[1] * div[2] > (long long)std::numeric_limits<int>::max()) return -1;
    const int mul[3] = {1, (int)div[0], (int)(div[0] * div[1])};
    std::vector<std::pair<unsigned, int>> iv;
    iv.reserve(n);
    for (int i = 0; i < n; ++i) {
        const float* p = xyz + 3 * i;
        if (!std::isfinite(p[0]) || !std::isfinite(p[1]) || !std::isfinite(p[2])) continue;
        const int i0 = static_cast<int>(std::floor(p[0] * inv) - static_cast<float>(min_b[0]));
        const int i1 = static_cast<int>(std::floor(p[1] * inv) - static_cast<float>(min_b[1]));
        const int i2 = static_cast<int>(std::floor(p[2] * inv) - static_cast<float>(min_b[2]));
        iv.emplace_back(static_cast<unsigned>(i0 * mul[0] + i1 * mul[1] + i2 * mul[2]), i);
    }
    std::stable_sort(iv.begin(), iv.end(), [](const std::pair<unsigned, int>& a, const std::pair<unsigned, int>& b) { return a.first < b.first; });
    int nout = 0;
    size_t i = 0;
    while (i < iv.size()) {
        size_t j = i;
        float sx = 0.0f, sy = 0.0f, sz = 0.0f;
        while (j < iv.size() && iv[j].first == iv[i].first) {
            const float* p = xyz + 3 * iv[j].second;
            sx += p[0]; sy += p[1]; sz += p[2];
            ++j;
        }
        const float fn = static_cast<float>(j - i);
        if (nout < cap) { out_xyz[3 * nout] = sx / fn; out_xyz[3 * nout + 1] = sy / fn; out_xyz[3 * nout + 2] = sz / fn; }
        ++nout;
        i = j;
    }
    return nout;
}

// lasermap_fov_segment (src/laserMapping.cpp:1136-1200).  state7 = LocalMap_Points min xyz, max xyz,
// Localmap_Initialized (as float 0/1).  Writes up to 3 boxes (cub_needrm) and returns their number; the
// caller applies them with orc_map_delete_boxes.
int orc_fov_segment(float* state7, const double* pos_lid, double cube_len, float DET_RANGE, float* boxes18) {
    const float MOV_THRESHOLD = 1.5f;
    float* vmin = state7; float* vmax = state7 + 3;
    if (state7[6] == 0.0f) {
        for (int i = 0; i < 3; i++) {
            vmin[i] = pos_lid[i] - cube_len / 2.0;
            vmax[i] = pos_lid[i] + cube_len / 2.0;
        }
        state7[6] = 1.0f;
        return 0;
    }
    float dist_to_map_edge[3][2];
    bool need_move = false;
    for (int i = 0; i < 3; i++) {
        dist_to_map_edge[i][0] = std::fabs(pos_lid[i] - vmin[i]);
        dist_to_map_edge[i][1] = std::fabs(pos_lid[i] - vmax[i]);
        if (dist_to_map_edge[i][0] <= MOV_THRESHOLD * DET_RANGE || dist_to_map_edge[i][1] <= MOV_THRESHOLD * DET_RANGE) need_move = true;
    }
    if (!need_move) return 0;
    float nmin[3] = {vmin[0], vmin[1], vmin[2]}, nmax[3] = {vmax[0], vmax[1], vmax[2]};
    float mov_dist = std::max((cube_len - 2.0 * MOV_THRESHOLD * DET_RANGE) * 0.5 * 0.9, double(DET_RANGE * (MOV_THRESHOLD - 1)));
    int nb = 0;
    for (int i = 0; i < 3; i++) {
        float tmin[3] = {vmin[0], vmin[1], vmin[2]}, tmax[3] = {vmax[0], vmax[1], vmax[2]};
        if (dist_to_map_edge[i][0] <= MOV_THRESHOLD * DET_RANGE) {
            nmax[i] -= mov_dist; nmin[i] -= mov_dist;
            tmin[i] = vmax[i] - mov_dist;
        } else if (dist_to_map_edge[i][1] <= MOV_THRESHOLD * DET_RANGE) {
            nmax[i] += mov_dist; nmin[i] += mov_dist;
            tmax[i] = vmin[i] + mov_dist;
        } else continue;
        for (int k = 0; k < 3; k++) { boxes18[6 * nb + k] = tmin[k]; boxes18[6 * nb + 3 + k] = tmax[k]; }
        ++nb;
    }
    for (int i = 0; i < 3; i++) { vmin[i] = nmin[i]; vmax[i] = nmax[i]; }
    return nb;
}

}  // extern "C"
