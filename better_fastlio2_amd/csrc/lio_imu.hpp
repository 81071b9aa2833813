// lio_imu.hpp -- host-side IMU front end: the step BEFORE the measurement path.
//
// Mirrors the reference's ImuProcess (src/IMU_Processing.hpp:30-441) with the same method names:
// static initialisation (IMU_init, :174-231), forward propagation of the filter through the IMU samples of one
// scan (UndistortPcl, :237-332) and the hand-over of the propagated poses to the point compensation, which runs
// on the GPU (lio_scan_undistort = UndistortPcl's point loop, :334-386).  The state and the covariance stay on
// the host (lio::esekf); only the raw scan, the pose list (a few hundred bytes) and the end state cross PCIe.
//
// ROS message types are replaced by plain structs holding exactly the fields the reference reads
// (sensor_msgs::Imu: header.stamp, linear_acceleration, angular_velocity; MeasureGroup: common_lib.h:324-335).
#pragma once
#include <vector>

#include "lio_esekf.hpp"
#include "lio_hip.h"

namespace lio {

struct ImuSample {                 // sensor_msgs::Imu
    double stamp = 0.0;            // header.stamp.toSec()
    double acc[3] = {0, 0, 0};     // linear_acceleration
    double gyr[3] = {0, 0, 0};     // angular_velocity
};

struct MeasureGroup {              // common_lib.h:324-335
    double lidar_beg_time = 0.0;
    double lidar_end_time = 0.0;
    const float* lidar = nullptr;  // raw points of the scan, records of stride_bytes; x,y,z at offset 0
    int n_points = 0;
    int stride_bytes = 16;
    int time_offset_bytes = 12;    // float, milliseconds after lidar_beg_time (`curvature`, byte 36 of PointXYZINormal)
    std::vector<ImuSample> imu;
};

class ImuProcess {
public:
    ImuProcess();
    void Reset();                                                      // IMU_Processing.hpp:91-103
    void set_extrinsic(const double T[3], const double R[9]);          // :131-135 (R row-major)
    void set_gyr_cov(const double s[3]);                               // :141-144
    void set_acc_cov(const double s[3]);                               // :150-153
    void set_gyr_bias_cov(const double b[3]);                          // :159-162
    void set_acc_bias_cov(const double b[3]);                          // :168-171
    // Process (:392-441).  ctx == nullptr: propagate the filter only (no point work, no GPU needed).
    // Returns LIO_OK; *undistorted = 1 when feats_undistort was produced on ctx (0 while initialising).
    int Process(const MeasureGroup& meas, esekf& kf_state, lio_ctx* ctx, int* undistorted);

    const std::vector<lio_pose6d>& poses() const { return IMUpose; }   // of the last UndistortPcl
    bool need_init() const { return imu_need_init_; }
    int init_iter() const { return init_iter_num; }
    double last_lidar_end_time() const { return last_lidar_end_time_; }

    double cov_acc[3], cov_gyr[3], cov_acc_scale[3], cov_gyr_scale[3], cov_bias_gyr[3], cov_bias_acc[3];
    double first_lidar_time = 0.0;
    double mean_acc[3], mean_gyr[3];

private:
    void IMU_init(const MeasureGroup& meas, esekf& kf_state, int& N);
    int UndistortPcl(const MeasureGroup& meas, esekf& kf_state, lio_ctx* ctx);

    double Q[12 * 12];
    ImuSample last_imu_;
    std::vector<lio_pose6d> IMUpose;
    double Lidar_R_wrt_IMU[9], Lidar_T_wrt_IMU[3];
    double angvel_last[3], acc_s_last[3];
    double last_lidar_end_time_ = 0.0;
    int init_iter_num = 1;
    bool b_first_frame_ = true;
    bool imu_need_init_ = true;
};

}  // namespace lio
