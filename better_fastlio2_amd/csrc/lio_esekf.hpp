// lio_esekf.hpp -- host-side iterated error-state Kalman filter on the 23-DOF LIO manifold.
//
// Mirrors the pieces of the reference's IKFoM instantiation that surround the measurement path,
// with the same names and call shapes, so that host code written against the reference reads the
// same (paths relative to the reference repository):
//   state_ikfom                          include/use-ikfom.hpp:21-30 (MTK_BUILD_MANIFOLD)
//   esekfom::dyn_share_datastruct        include/IKFoM_toolkit/esekfom/esekfom.hpp:79-89
//   esekfom::esekf::init_dyn_share       esekfom.hpp:239-258
//   esekf::update_iterated_dyn_share_modified   esekfom.hpp:1620-1938
//   get_x / get_P / change_x / change_P  esekfom.hpp:1940-1961
//   esekf::predict                       esekfom.hpp:280-383 with the process model of use-ikfom.hpp:44-99
// The reference builds these from Eigen + Boost.PP (neither ships with it nor with this image);
// here they are plain C++17 with fixed-size arrays.
//
// The measurement callback has the reference's shape
//     void h(state_ikfom& s, dyn_share_datastruct& d)          (esekfom.hpp:130)
// and, being the consumer of the GPU path, receives the reduced information form
// (d.HTH = h_x^T h_x, d.HTh = h_x^T h, d.m rows) instead of the m x 12 matrix; d.h_x / d.h are
// filled only when m < 23, where the reference switches to the dense gain (esekfom.hpp:1720-1750).
#pragma once
#include <cstddef>
#include <functional>
#include <vector>

namespace lio {

constexpr int DOF = 23;                       // state_ikfom::DOF
constexpr int DIM = 24;                       // state_ikfom::DIM (two quaternions, S2 stored as 3-vector)

struct Vec3 { double x = 0, y = 0, z = 0; };
struct Quat { double x = 0, y = 0, z = 0, w = 1; };   // Eigen coefficient order

// DOF layout: pos 0, rot 3, offset_R_L_I 6, offset_T_L_I 9, vel 12, bg 15, ba 18, grav 21 (S2)
struct state_ikfom {
    Vec3 pos;
    Quat rot;
    Quat offset_R_L_I;
    Vec3 offset_T_L_I;
    Vec3 vel, bg, ba;
    Vec3 grav{0.0, 0.0, -9.809};               // S2 of length 98090/10000 (use-ikfom.hpp:8)

    void boxplus(const double dx[DOF]);                         // x <- x [+] dx
    void boxminus(double res[DOF], const state_ikfom& other) const;   // res = x [-] other
    void to_array(double x26[26]) const;
    static state_ikfom from_array(const double x26[26]);
    void pose14(double p[14]) const;           // the block handed to lio_measure
};

struct dyn_share_datastruct {
    bool valid = true;
    bool converge = true;
    int m = 0;                                 // rows of h_x
    double HTH[144];                           // h_x^T h_x, row-major 12x12
    double HTh[12];                            // h_x^T h
    std::vector<double> h_x;                   // m x 12 row-major, only when m < DOF
    std::vector<double> h;                     // m
    double res_mean = 0.0;
};

typedef std::function<void(state_ikfom&, dyn_share_datastruct&)> measurementModel_dyn_share;

struct input_ikfom { double acc[3] = {0, 0, 0}, gyro[3] = {0, 0, 0}; };   // use-ikfom.hpp:32-35

// process model, use-ikfom.hpp:44-99 (row-major; flattened-state rows: pos 0, rot 3, offset_R_L_I 6,
// offset_T_L_I 9, vel 12, bg 15, ba 18, grav 21)
void process_noise_cov(double Q[12 * 12]);
void get_f(const state_ikfom& s, const input_ikfom& in, double f[DIM]);
void df_dx(const state_ikfom& s, const input_ikfom& in, double F[DIM * DOF]);
void df_dw(const state_ikfom& s, const input_ikfom& in, double W[DIM * 12]);

// Eigen-equivalent quaternion helpers used by the IMU front end
Vec3 quat_rotate(const Quat& q, const Vec3& v);             // q * v
void quat_to_matrix(const Quat& q, double R[9]);            // toRotationMatrix, row-major
Quat quat_from_matrix(const double R[9]);                   // Quaternion(Matrix3)
Vec3 s2_from_vector(const Vec3& v);                         // S2(vec): direction of v, length 9.809

// h_x^T h_x (12 x 12 row-major) and h_x^T h from m rows, h_x column-major with leading dimension m (esekfom.hpp:1790,1810)
void hth_from_rows_colmajor(const double* hx, const double* h, size_t m, double HTH[144], double HTh[12]);

struct update_stats {
    int n_calls = 0, n_search = 0, last_m = 0;
    bool converged = false;
    double last_res_mean = 0.0, solve_seconds = 0.0, measure_seconds = 0.0;
};

class esekf {
public:
    esekf();
    void init_dyn_share(measurementModel_dyn_share h, int maximum_iteration, const double limit_vector[DOF]);
    void set_measurement_model(measurementModel_dyn_share h) { h_dyn_share_ = std::move(h); h_begin_ = nullptr; h_end_ = nullptr; }
    // Two-phase form for an asynchronous measurement model: begin(s, d) enqueues the pass for state s (reads
    // d.converge), end(d) waits for it and fills d.  Between the two the update computes what does not depend on the
    // result (x [-] x_prop, the Jacobian-corrected P, its first inversion): same arithmetic, overlapped with the pass.
    void set_measurement_model_async(std::function<void(state_ikfom&, const dyn_share_datastruct&)> begin,
                                     std::function<void(dyn_share_datastruct&)> end) {
        h_begin_ = std::move(begin); h_end_ = std::move(end);
    }
    // R = LASER_POINT_COV (laserMapping.cpp:14, :2380)
    void update_iterated_dyn_share_modified(double R, double& solve_time);
    // IMU propagation over dt with input i_in and process noise Q (12 x 12 row-major), esekfom.hpp:280-383
    void predict(double dt, const double Q[12 * 12], const input_ikfom& i_in);
    const state_ikfom& get_x() const { return x_; }
    const double* get_P() const { return P_; }             // 23 x 23 row-major
    void change_x(const state_ikfom& s) { x_ = s; }
    void change_P(const double* P);
    const update_stats& stats() const { return stats_; }

private:
    state_ikfom x_;
    double P_[DOF * DOF];
    double limit_[DOF];
    int maximum_iter_ = 0;
    measurementModel_dyn_share h_dyn_share_;
    std::function<void(state_ikfom&, const dyn_share_datastruct&)> h_begin_;
    std::function<void(dyn_share_datastruct&)> h_end_;
    update_stats stats_;
};

}  // namespace lio
