// TEST INFRASTRUCTURE ONLY (tests/).  Host build (g++) of the arithmetic the GPU kernels execute:
// it includes the very same headers the kernels include (csrc/lio_math.h, csrc/lio_common.h) and
// the host-side filter (csrc/lio_esekf.cpp), and exposes them to the CPU test-suite so that they
// can be checked against the oracle without a GPU.  Nothing in the product links this file.
#include <chrono>
#include <cstdint>
#include <cstring>

#include "../../better_fastlio2_amd/csrc/lio_math.h"
#include "../../better_fastlio2_amd/csrc/lio_esekf.hpp"
#include "../../include/lio_hip.h"

using namespace lio;

extern "C" {

int hh_plane_fit(const float* pts5x3, float thr, float* pabcd) {
    float px[5], py[5], pz[5];
    for (int j = 0; j < 5; ++j) { px[j] = pts5x3[3 * j]; py[j] = pts5x3[3 * j + 1]; pz[j] = pts5x3[3 * j + 2]; }
    float a, b, c, d;
    const bool ok = plane_fit5(px, py, pz, thr, a, b, c, d);
    pabcd[0] = a; pabcd[1] = b; pabcd[2] = c; pabcd[3] = d;
    return ok ? 1 : 0;
}

void hh_body_to_world(const double* state14, const float* body, int n, float* world) {
    Pose14 p; std::memcpy(p.v, state14, sizeof(p.v));
    const PoseQ s = pose_unpack(p);
    for (int i = 0; i < n; ++i)
        body_to_world(s, body[3 * i], body[3 * i + 1], body[3 * i + 2], world[3 * i], world[3 * i + 1], world[3 * i + 2]);
}

// one scan point through gate-less plane fit + residual + Jacobian row; returns selected flag
int hh_point(const double* state14, const float* body3, const float* world3, const float* nbr5x3, int ext,
             float* normvec4, double* row12, double* h) {
    Pose14 p; std::memcpy(p.v, state14, sizeof(p.v));
    const PoseQ s = pose_unpack(p);
    float px[5], py[5], pz[5];
    for (int j = 0; j < 5; ++j) { px[j] = nbr5x3[3 * j]; py[j] = nbr5x3[3 * j + 1]; pz[j] = nbr5x3[3 * j + 2]; }
    float nx = 0, ny = 0, nz = 0, pd2 = 0;
    const bool sel = point_residual(px, py, pz, world3[0], world3[1], world3[2], body3[0], body3[1], body3[2], nx, ny, nz, pd2);
    for (int c = 0; c < 12; ++c) row12[c] = 0.0;
    *h = 0.0;
    if (sel) {
        normvec4[0] = nx; normvec4[1] = ny; normvec4[2] = nz; normvec4[3] = pd2;
        if (ext) jacobian_row<true>(s, body3[0], body3[1], body3[2], nx, ny, nz, pd2, row12, *h);
        else jacobian_row<false>(s, body3[0], body3[1], body3[2], nx, ny, nz, pd2, row12, *h);
    }
    return sel ? 1 : 0;
}

void hh_keys(const float* xyz, int n, float inv_cell, uint64_t* packed, uint64_t* morton, uint32_t* hash) {
    for (int i = 0; i < n; ++i) {
        const uint32_t cx = (uint32_t)cell_coord(xyz[3 * i], inv_cell), cy = (uint32_t)cell_coord(xyz[3 * i + 1], inv_cell),
                       cz = (uint32_t)cell_coord(xyz[3 * i + 2], inv_cell);
        packed[i] = pack_key(cx, cy, cz);
        morton[i] = morton3(cx, cy, cz);
        hash[i] = hash_key(packed[i]);
    }
}
void hh_morton_decode(const uint64_t* morton, int n, uint32_t* xyz) {
    for (int i = 0; i < n; ++i) {
        xyz[3 * i] = compact21(morton[i] >> 2); xyz[3 * i + 1] = compact21(morton[i] >> 1); xyz[3 * i + 2] = compact21(morton[i]);
    }
}
float hh_sqdist(const float* a, const float* b) { return sqdist3(a[0], a[1], a[2], b[0], b[1], b[2]); }

// the host filter driven by a caller-supplied measurement callback (the CPU tests plug the oracle's
// h_share_model in here; the GPU tests use lio_eskf_update_iterated_dyn_share_modified instead)
typedef int (*hh_measure_cb)(const double* pose14, int converge, int* m, double* res_mean, double* HTH144,
                             double* HTh12, double* h_x, double* h, int cap_rows);

int hh_eskf_update(double* x26, double* P, double R, int max_iter, const double* limit23, hh_measure_cb cb,
                   int cap_rows, int* n_calls) {
    esekf kf;
    kf.change_x(state_ikfom::from_array(x26));
    kf.change_P(P);
    int calls = 0;
    std::vector<double> hx((size_t)cap_rows * 12), hv((size_t)cap_rows);
    kf.init_dyn_share([&](state_ikfom& s, dyn_share_datastruct& d) {
        double pose[14];
        s.pose14(pose);
        int m = 0; double rm = 0;
        const int valid = cb(pose, d.converge ? 1 : 0, &m, &rm, d.HTH, d.HTh, hx.data(), hv.data(), cap_rows);
        ++calls;
        d.valid = valid != 0; d.m = m; d.res_mean = rm;
        if (d.valid && m < DOF) {
            d.h_x.assign(hx.begin(), hx.begin() + (size_t)m * 12);
            d.h.assign(hv.begin(), hv.begin() + m);
        }
    }, max_iter, limit23);
    double st = 0;
    kf.update_iterated_dyn_share_modified(R, st);
    kf.get_x().to_array(x26);
    std::memcpy(P, kf.get_P(), sizeof(double) * DOF * DOF);
    if (n_calls) *n_calls = calls;
    return 0;
}

void hh_boxplus(double* x26, const double* dx23) {
    state_ikfom s = state_ikfom::from_array(x26);
    s.boxplus(dx23);
    s.to_array(x26);
}
void hh_boxminus(const double* x26, const double* y26, double* dx23) {
    state_ikfom::from_array(x26).boxminus(dx23, state_ikfom::from_array(y26));
}

// h_x^T h_x / h_x^T h from column-major rows (the callback path's filter side), and its time for m rows
double hh_hth_rows(const double* hx, const double* h, int m, double* HTH, double* HTh, int reps) {
    const auto t0 = std::chrono::steady_clock::now();
    for (int r = 0; r < (reps > 0 ? reps : 1); ++r) lio::hth_from_rows_colmajor(hx, h, (size_t)m, HTH, HTh);
    return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() / (reps > 0 ? reps : 1);
}
}  // extern "C"

// micro-benchmark of the host-side algebra of one scan step (no GPU, synthetic H^T H): seconds per update
extern "C" double hh_eskf_bench(int iters, int max_iter) {
    using namespace lio;
    double HTH[144], HTh[12];
    for (int i = 0; i < 12; ++i) {
        for (int j = 0; j < 12; ++j) HTH[i * 12 + j] = (i == j ? 5000.0 + 100.0 * i : 10.0 / (1 + i + j));
        HTh[i] = 1e-3 * (i - 6);
    }
    esekf kf;
    double lim[DOF];
    for (double& v : lim) v = 0.001;
    kf.init_dyn_share([&](state_ikfom&, dyn_share_datastruct& d) {
        d.valid = true; d.m = 5000; d.res_mean = 0.01;
        std::memcpy(d.HTH, HTH, sizeof(HTH)); std::memcpy(d.HTh, HTh, sizeof(HTh));
    }, max_iter, lim);
    double P0[DOF * DOF] = {0};
    for (int i = 0; i < DOF; ++i) P0[i * DOF + i] = i < 6 ? 1.0 : 1e-4;
    double st = 0.0, total = 0.0;
    for (int it = 0; it < iters; ++it) {
        kf.change_x(state_ikfom());
        kf.change_P(P0);
        st = 0.0;
        kf.update_iterated_dyn_share_modified(0.001, st);
        total += st;
    }
    return total / iters;
}
