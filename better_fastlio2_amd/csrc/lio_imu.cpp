// lio_imu.cpp -- IMU front end (see lio_imu.hpp).  Host C++17, fp64; line numbers cite src/IMU_Processing.hpp
// of the reference repository unless another file is named.
#include "lio_imu.hpp"

#include <cmath>

namespace lio {
namespace {

const int MAX_INI_COUNT = 10;                 // IMU_Processing.hpp:4
const double G_m_s2 = 9.81;                   // include/common_lib.h:143

double norm3(const double v[3]) { return std::sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]); }

// set_pose6d (include/common_lib.h:446-460)
lio_pose6d set_pose6d(double t, const double a[3], const double g[3], const Vec3& v, const Vec3& p, const Quat& rot) {
    lio_pose6d k;
    k.offset_time = t;
    for (int i = 0; i < 3; ++i) { k.acc[i] = a[i]; k.gyr[i] = g[i]; }
    k.vel[0] = v.x; k.vel[1] = v.y; k.vel[2] = v.z;
    k.pos[0] = p.x; k.pos[1] = p.y; k.pos[2] = p.z;
    quat_to_matrix(rot, k.rot);
    return k;
}

}  // namespace

ImuProcess::ImuProcess() {                                   // :57-72
    init_iter_num = 1;
    process_noise_cov(Q);
    for (int i = 0; i < 3; ++i) {
        cov_acc[i] = 0.1; cov_gyr[i] = 0.1;
        cov_bias_gyr[i] = 0.0001; cov_bias_acc[i] = 0.0001;
        cov_acc_scale[i] = 0.1; cov_gyr_scale[i] = 0.1;      // the node always sets these (laserMapping.cpp:2143-2144)
        mean_gyr[i] = 0.0; angvel_last[i] = 0.0; acc_s_last[i] = 0.0;
        Lidar_T_wrt_IMU[i] = 0.0;
    }
    mean_acc[0] = 0.0; mean_acc[1] = 0.0; mean_acc[2] = -1.0;
    for (int i = 0; i < 9; ++i) Lidar_R_wrt_IMU[i] = (i % 4 == 0) ? 1.0 : 0.0;
    last_imu_ = ImuSample();
}

void ImuProcess::Reset() {                                   // :91-103
    mean_acc[0] = 0.0; mean_acc[1] = 0.0; mean_acc[2] = -1.0;
    for (int i = 0; i < 3; ++i) { mean_gyr[i] = 0.0; angvel_last[i] = 0.0; }
    imu_need_init_ = true;
    init_iter_num = 1;
    IMUpose.clear();
    last_imu_ = ImuSample();
}

void ImuProcess::set_extrinsic(const double T[3], const double R[9]) {
    for (int i = 0; i < 3; ++i) Lidar_T_wrt_IMU[i] = T[i];
    for (int i = 0; i < 9; ++i) Lidar_R_wrt_IMU[i] = R[i];
}
void ImuProcess::set_gyr_cov(const double s[3]) { for (int i = 0; i < 3; ++i) cov_gyr_scale[i] = s[i]; }
void ImuProcess::set_acc_cov(const double s[3]) { for (int i = 0; i < 3; ++i) cov_acc_scale[i] = s[i]; }
void ImuProcess::set_gyr_bias_cov(const double b[3]) { for (int i = 0; i < 3; ++i) cov_bias_gyr[i] = b[i]; }
void ImuProcess::set_acc_bias_cov(const double b[3]) { for (int i = 0; i < 3; ++i) cov_bias_acc[i] = b[i]; }

// Static initialisation: running mean / variance of the samples, gravity from the mean specific force,
// gyroscope bias from the mean rate, extrinsics and the initial covariance (:174-231).
void ImuProcess::IMU_init(const MeasureGroup& meas, esekf& kf_state, int& N) {
    if (b_first_frame_) {                                    // :183-194
        Reset();
        N = 1;
        b_first_frame_ = false;
        for (int i = 0; i < 3; ++i) { mean_acc[i] = meas.imu.front().acc[i]; mean_gyr[i] = meas.imu.front().gyr[i]; }
        first_lidar_time = meas.lidar_beg_time;
    }
    for (const ImuSample& imu : meas.imu) {                  // :197-213
        for (int i = 0; i < 3; ++i) {
            const double cur_acc = imu.acc[i], cur_gyr = imu.gyr[i];
            mean_acc[i] += (cur_acc - mean_acc[i]) / N;
            mean_gyr[i] += (cur_gyr - mean_gyr[i]) / N;
            cov_acc[i] = cov_acc[i] * (N - 1.0) / N + (cur_acc - mean_acc[i]) * (cur_acc - mean_acc[i]) * (N - 1.0) / (N * N);
            cov_gyr[i] = cov_gyr[i] * (N - 1.0) / N + (cur_gyr - mean_gyr[i]) * (cur_gyr - mean_gyr[i]) * (N - 1.0) / (N * N);
        }
        N++;
    }
    state_ikfom init_state = kf_state.get_x();               // :215-223
    const double na = norm3(mean_acc);
    init_state.grav = s2_from_vector(Vec3{-mean_acc[0] / na * G_m_s2, -mean_acc[1] / na * G_m_s2, -mean_acc[2] / na * G_m_s2});
    init_state.bg = Vec3{mean_gyr[0], mean_gyr[1], mean_gyr[2]};
    init_state.offset_T_L_I = Vec3{Lidar_T_wrt_IMU[0], Lidar_T_wrt_IMU[1], Lidar_T_wrt_IMU[2]};
    init_state.offset_R_L_I = quat_from_matrix(Lidar_R_wrt_IMU);
    kf_state.change_x(init_state);

    double init_P[DOF * DOF];                                // :225-232
    for (int i = 0; i < DOF * DOF; ++i) init_P[i] = 0.0;
    for (int i = 0; i < DOF; ++i) init_P[i * DOF + i] = 1.0;
    for (int i = 6; i < 9; ++i) init_P[i * DOF + i] = 0.00001;
    for (int i = 9; i < 12; ++i) init_P[i * DOF + i] = 0.00001;
    for (int i = 15; i < 18; ++i) init_P[i * DOF + i] = 0.0001;
    for (int i = 18; i < 21; ++i) init_P[i * DOF + i] = 0.001;
    for (int i = 21; i < 23; ++i) init_P[i * DOF + i] = 0.00001;
    kf_state.change_P(init_P);
    last_imu_ = meas.imu.back();
}

// Forward propagation through the IMU samples of the scan (:237-332), then the point compensation on the GPU.
int ImuProcess::UndistortPcl(const MeasureGroup& meas, esekf& kf_state, lio_ctx* ctx) {
    // the raw scan starts its way to the GPU now; the forward propagation below runs while it travels
    if (ctx) {
        const int rc = lio_scan_upload_raw(ctx, meas.lidar, meas.n_points, meas.stride_bytes, meas.time_offset_bytes);
        if (rc != LIO_OK) return rc;
    }
    std::vector<ImuSample> v_imu;                            // :240-241
    v_imu.reserve(meas.imu.size() + 1);
    v_imu.push_back(last_imu_);
    v_imu.insert(v_imu.end(), meas.imu.begin(), meas.imu.end());
    const double imu_end_time = v_imu.back().stamp;
    const double pcl_beg_time = meas.lidar_beg_time;
    const double pcl_end_time = meas.lidar_end_time;

    state_ikfom imu_state = kf_state.get_x();                // :253-257
    IMUpose.clear();
    IMUpose.push_back(set_pose6d(0.0, acc_s_last, angvel_last, imu_state.vel, imu_state.pos, imu_state.rot));

    double dt = 0;
    input_ikfom in;
    for (size_t k = 0; k + 1 < v_imu.size(); ++k) {          // :268-321
        const ImuSample& head = v_imu[k];
        const ImuSample& tail = v_imu[k + 1];
        if (tail.stamp < last_lidar_end_time_) continue;
        double angvel_avr[3], acc_avr[3];
        for (int i = 0; i < 3; ++i) {
            angvel_avr[i] = 0.5 * (head.gyr[i] + tail.gyr[i]);
            acc_avr[i] = 0.5 * (head.acc[i] + tail.acc[i]);
        }
        const double na = norm3(mean_acc);
        for (int i = 0; i < 3; ++i) acc_avr[i] = acc_avr[i] * G_m_s2 / na;        // :287
        if (head.stamp < last_lidar_end_time_) dt = tail.stamp - last_lidar_end_time_;
        else dt = tail.stamp - head.stamp;
        for (int i = 0; i < 3; ++i) {
            in.acc[i] = acc_avr[i];
            in.gyro[i] = angvel_avr[i];
            Q[i * 13] = cov_gyr[i];                          // :303-306
            Q[(3 + i) * 13] = cov_acc[i];
            Q[(6 + i) * 13] = cov_bias_gyr[i];
            Q[(9 + i) * 13] = cov_bias_acc[i];
        }
        kf_state.predict(dt, Q, in);                         // :309

        imu_state = kf_state.get_x();                        // :311-317
        angvel_last[0] = angvel_avr[0] - imu_state.bg.x; angvel_last[1] = angvel_avr[1] - imu_state.bg.y; angvel_last[2] = angvel_avr[2] - imu_state.bg.z;
        const Vec3 a = quat_rotate(imu_state.rot, Vec3{acc_avr[0] - imu_state.ba.x, acc_avr[1] - imu_state.ba.y, acc_avr[2] - imu_state.ba.z});
        acc_s_last[0] = a.x; acc_s_last[1] = a.y; acc_s_last[2] = a.z;
        acc_s_last[0] += imu_state.grav.x; acc_s_last[1] += imu_state.grav.y; acc_s_last[2] += imu_state.grav.z;

        const double offs_t = tail.stamp - pcl_beg_time;     // :319-321
        IMUpose.push_back(set_pose6d(offs_t, acc_s_last, angvel_last, imu_state.vel, imu_state.pos, imu_state.rot));
    }

    const double note = pcl_end_time > imu_end_time ? 1.0 : -1.0;                 // :324-326
    dt = note * (pcl_end_time - imu_end_time);
    kf_state.predict(dt, Q, in);

    imu_state = kf_state.get_x();                            // :328-330
    last_imu_ = meas.imu.back();
    last_lidar_end_time_ = pcl_end_time;

    if (!ctx) return LIO_OK;
    double end14[14];                                        // :334-386 on the device
    imu_state.pose14(end14);
    return lio_scan_compensate(ctx, IMUpose.data(), (int)IMUpose.size(), end14);
}

int ImuProcess::Process(const MeasureGroup& meas, esekf& kf_state, lio_ctx* ctx, int* undistorted) {   // :392-441
    if (undistorted) *undistorted = 0;
    if (meas.imu.empty()) return LIO_OK;
    if (imu_need_init_) {
        IMU_init(meas, kf_state, init_iter_num);
        imu_need_init_ = true;
        last_imu_ = meas.imu.back();
        if (init_iter_num > MAX_INI_COUNT) {
            const double na = norm3(mean_acc);
            for (int i = 0; i < 3; ++i) cov_acc[i] *= std::pow(G_m_s2 / na, 2);
            imu_need_init_ = false;
            for (int i = 0; i < 3; ++i) { cov_acc[i] = cov_acc_scale[i]; cov_gyr[i] = cov_gyr_scale[i]; }
        }
        return LIO_OK;
    }
    const int rc = UndistortPcl(meas, kf_state, ctx);
    if (rc == LIO_OK && ctx && undistorted) *undistorted = 1;
    return rc;
}

}  // namespace lio
