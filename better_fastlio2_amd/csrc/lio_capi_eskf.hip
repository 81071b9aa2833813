// lio_capi_eskf.hip -- the C API of the host-side filter and IMU front end (include/lio_hip.h: lio_eskf_*, lio_imu_*, lio_state_*):
// thin bindings of csrc/lio_esekf.{hpp,cpp} (update_iterated_dyn_share_modified, esekfom.hpp:1620-1938; predict :280-383) and
// csrc/lio_imu.{hpp,cpp} (src/IMU_Processing.hpp) to the context's measurement passes.  Host code only.
#include "lio_ctx.h"

extern "C" {
// ---- host-side ESKF -----------------------------------------------------------------------------
struct lio_eskf {
    lio::esekf kf;
};

lio_eskf* lio_eskf_create(void) { return new lio_eskf(); }
void lio_eskf_destroy(lio_eskf* kf) { delete kf; }
int lio_eskf_init_dyn_share(lio_eskf* kf, int maximum_iter, const double limit23[23]) {
    if (!kf || maximum_iter < 0 || !limit23) return LIO_ERR_INVALID_ARG;
    kf->kf.init_dyn_share(nullptr, maximum_iter, limit23);
    return LIO_OK;
}
int lio_eskf_change_x(lio_eskf* kf, const double x26[26]) {
    if (!kf || !x26) return LIO_ERR_INVALID_ARG;
    kf->kf.change_x(lio::state_ikfom::from_array(x26));
    return LIO_OK;
}
int lio_eskf_change_P(lio_eskf* kf, const double P[23 * 23]) {
    if (!kf || !P) return LIO_ERR_INVALID_ARG;
    kf->kf.change_P(P);
    return LIO_OK;
}
int lio_eskf_get_x(lio_eskf* kf, double x26[26]) {
    if (!kf || !x26) return LIO_ERR_INVALID_ARG;
    kf->kf.get_x().to_array(x26);
    return LIO_OK;
}
int lio_eskf_get_P(lio_eskf* kf, double P[23 * 23]) {
    if (!kf || !P) return LIO_ERR_INVALID_ARG;
    std::memcpy(P, kf->kf.get_P(), sizeof(double) * 23 * 23);
    return LIO_OK;
}
int lio_eskf_update_iterated_dyn_share_modified(lio_eskf* kf, lio_ctx* ctx, double R, int extrinsic_est_en,
                                                lio_update_log* log) {
    if (!kf || !ctx || !(R > 0.0)) return LIO_ERR_INVALID_ARG;
    int status = LIO_OK;
    // (the converge flag of every call -- recorded by lio_measure_begin -- predicts the NEXT update's passes)
    if (!ctx->pattern_cur.empty()) { ctx->pattern_prev.swap(ctx->pattern_cur); ctx->pattern_cur.clear(); }
    // the reference's h_share_model, bound to the GPU path (laserMapping.cpp:1876 <-> lio_measure), in two halves so
    // that the filter's result-independent algebra runs while the GPU works on the pass
    kf->kf.set_measurement_model_async(
        [&](lio::state_ikfom& s, const lio::dyn_share_datastruct& d) {
            if (status != LIO_OK) return;
            double pose[14];
            s.pose14(pose);
            // (lio_measure_begin also puts the passes after this one onto the stream, each behind a gate, as the previous update ran them)
            const int rc = lio_measure_begin(ctx, pose, d.converge ? 1 : 0, extrinsic_est_en);
            if (rc != LIO_OK) { status = rc; return; }
        },
        [&](lio::dyn_share_datastruct& d) {
            if (status != LIO_OK) { d.valid = false; d.m = 0; return; }
            lio_out o;
            const int rc = lio_measure_end(ctx, &o);
            if (rc != LIO_OK) { status = rc; d.valid = false; d.m = 0; return; }
            d.valid = o.valid != 0;
            d.m = o.m;
            d.res_mean = o.res_mean;
            std::memcpy(d.HTH, o.HTH, sizeof(d.HTH));
            std::memcpy(d.HTh, o.HTh, sizeof(d.HTh));
            if (d.valid && o.m < lio::DOF) {                    // dense-gain branch needs the rows
                d.h_x.assign((size_t)o.m * 12, 0.0);
                d.h.assign((size_t)o.m, 0.0);
                const int rc2 = lio_measure_rows(ctx, d.h_x.data(), d.h.data(), o.m, 0);
                if (rc2 != LIO_OK) { status = rc2; d.valid = false; }
            }
        });
    double solve_time = 0.0;
    kf->kf.update_iterated_dyn_share_modified(R, solve_time);
    gate_call_off_this_scan(ctx);                // the update ended earlier than the previous one: nothing of THIS scan stays armed
    ctx->pattern_prev.swap(ctx->pattern_cur);
    ctx->pattern_cur.clear();
    if (log) {
        const lio::update_stats& st = kf->kf.stats();
        log->n_calls = st.n_calls; log->n_search = st.n_search; log->converged = st.converged ? 1 : 0;
        log->last_m = st.last_m; log->last_res_mean = st.last_res_mean;
        log->solve_seconds = st.solve_seconds; log->measure_seconds = st.measure_seconds;
    }
    return status;
}
// esekf with a caller-supplied h_dyn_share (esekfom.hpp:239-258): the callback answers with rows, the products
// h_x^T h_x / h_x^T h are formed here (esekfom.hpp:1790,1810) -- column blocks that stay in the L1 / L2 of one core
int lio_eskf_update_iterated_dyn_share_callback(lio_eskf* kf, double R, lio_h_dyn_share_fn h, void* user, lio_update_log* log) {
    if (!kf || !h || !(R > 0.0)) return LIO_ERR_INVALID_ARG;
    int status = LIO_OK;
    kf->kf.set_measurement_model([&](lio::state_ikfom& s, lio::dyn_share_datastruct& d) {
        double x26[26];
        s.to_array(x26);
        lio_dyn_share ds{};
        ds.converge = d.converge ? 1 : 0;
        ds.valid = 1;
        h(user, x26, &ds);
        d.valid = ds.valid != 0;
        d.m = ds.m;
        d.res_mean = 0.0;
        if (!d.valid) return;
        // what h_share_model guarantees (laserMapping.cpp:1956-1961: valid = false when effct_feat_num < 1) is checked, not trusted
        if (ds.m < 1) { d.valid = false; d.m = 0; return; }
        if (ds.h_x != nullptr && ds.h == nullptr) { status = LIO_ERR_INVALID_ARG; d.valid = false; d.m = 0; return; }
        if (ds.h_x == nullptr) {
            std::memcpy(d.HTH, ds.HTH, sizeof(d.HTH));
            std::memcpy(d.HTh, ds.HTh, sizeof(d.HTh));
        } else {
            lio::hth_from_rows_colmajor(ds.h_x, ds.h, (size_t)ds.m, d.HTH, d.HTh);
        }
        if (ds.m < lio::DOF && ds.h_x != nullptr) {            // dense-gain branch (esekfom.hpp:1720-1750): rows, row-major
            d.h_x.assign((size_t)ds.m * 12, 0.0);
            d.h.assign((size_t)ds.m, 0.0);
            for (int r = 0; r < ds.m; ++r) {
                for (int cidx = 0; cidx < 12; ++cidx) d.h_x[(size_t)r * 12 + cidx] = ds.h_x[(size_t)cidx * ds.m + r];
                d.h[r] = ds.h[r];
            }
        } else if (ds.m < lio::DOF) {
            d.valid = false;                                    // the reduced form cannot feed the dense gain
        }
    });
    double solve_time = 0.0;
    kf->kf.update_iterated_dyn_share_modified(R, solve_time);
    if (log) {
        const lio::update_stats& st = kf->kf.stats();
        log->n_calls = st.n_calls; log->n_search = st.n_search; log->converged = st.converged ? 1 : 0;
        log->last_m = st.last_m; log->last_res_mean = st.last_res_mean;
        log->solve_seconds = st.solve_seconds; log->measure_seconds = st.measure_seconds;
    }
    return status;
}
int lio_eskf_scan_step_device(lio_eskf* kf, lio_ctx* ctx, const double x26[26], const double P[23 * 23],
                              const void* d_body_f4, int n, double R, int extrinsic_est_en, lio_update_log* log) {
    if (!kf || !ctx) return LIO_ERR_INVALID_ARG;
    int rc = LIO_OK;
    if (x26 != nullptr) rc = lio_eskf_change_x(kf, x26);
    if (rc == LIO_OK && P != nullptr) rc = lio_eskf_change_P(kf, P);
    if (rc == LIO_OK && d_body_f4 != nullptr) rc = lio_scan_set_device(ctx, d_body_f4, n);
    if (rc == LIO_OK) rc = lio_eskf_update_iterated_dyn_share_modified(kf, ctx, R, extrinsic_est_en, log);
    return rc;
}
int lio_eskf_scan_step_pipelined(lio_eskf* kf, lio_ctx* ctx, const double x26[26], const double P[23 * 23],
                                 const void* next_host_f4, int n_next, double R, int extrinsic_est_en, lio_update_log* log) {
    if (!kf || !ctx) return LIO_ERR_INVALID_ARG;
    ctx->stamp('P');
    int rc = lio_scan_swap(ctx);                                   // the scan uploaded during the previous step
    ctx->stamp('w');
    if (rc == LIO_OK && next_host_f4 != nullptr) rc = lio_scan_upload_async(ctx, next_host_f4, n_next);   // travels during this one
    if (rc == LIO_OK && x26 != nullptr) rc = lio_eskf_change_x(kf, x26);
    if (rc == LIO_OK && P != nullptr) rc = lio_eskf_change_P(kf, P);
    // (the next call of this function is what follows on the stream: the first search of the scan just handed over may be put
    // there behind this scan's last passes -- arm_predicted)
    ctx->arm_across_scans = next_host_f4 != nullptr && n_next > 0;
    if (rc == LIO_OK) rc = lio_eskf_update_iterated_dyn_share_modified(kf, ctx, R, extrinsic_est_en, log);
    ctx->arm_across_scans = false;
    if (rc != LIO_OK) enter(ctx);
    ctx->stamp('U');
    return rc;
}
int lio_state_boxplus(double x26[26], const double dx23[23]) {
    if (!x26 || !dx23) return LIO_ERR_INVALID_ARG;
    lio::state_ikfom s = lio::state_ikfom::from_array(x26);
    s.boxplus(dx23);
    s.to_array(x26);
    return LIO_OK;
}
int lio_eskf_predict(lio_eskf* kf, double dt, const double Q[12 * 12], const double acc[3], const double gyro[3]) {
    if (!kf || !Q || !acc || !gyro) return LIO_ERR_INVALID_ARG;
    lio::input_ikfom in;
    for (int i = 0; i < 3; ++i) { in.acc[i] = acc[i]; in.gyro[i] = gyro[i]; }
    kf->kf.predict(dt, Q, in);
    return LIO_OK;
}
int lio_process_noise_cov(double Q[12 * 12]) {
    if (!Q) return LIO_ERR_INVALID_ARG;
    lio::process_noise_cov(Q);
    return LIO_OK;
}

// ---- host-side IMU front end ----------------------------------------------------------------------
struct lio_imu {
    lio::ImuProcess p;
};
lio_imu* lio_imu_create(void) { return new lio_imu(); }
void lio_imu_destroy(lio_imu* imu) { delete imu; }
int lio_imu_reset(lio_imu* imu) {
    if (!imu) return LIO_ERR_INVALID_ARG;
    imu->p.Reset();
    return LIO_OK;
}
int lio_imu_set_extrinsic(lio_imu* imu, const double T[3], const double R[9]) {
    if (!imu || !T || !R) return LIO_ERR_INVALID_ARG;
    imu->p.set_extrinsic(T, R);
    return LIO_OK;
}
int lio_imu_set_cov(lio_imu* imu, const double gyr_cov[3], const double acc_cov[3], const double b_gyr_cov[3],
                    const double b_acc_cov[3]) {
    if (!imu) return LIO_ERR_INVALID_ARG;
    if (gyr_cov) imu->p.set_gyr_cov(gyr_cov);
    if (acc_cov) imu->p.set_acc_cov(acc_cov);
    if (b_gyr_cov) imu->p.set_gyr_bias_cov(b_gyr_cov);
    if (b_acc_cov) imu->p.set_acc_bias_cov(b_acc_cov);
    return LIO_OK;
}
int lio_imu_process(lio_imu* imu, lio_eskf* kf, lio_ctx* ctx, double lidar_beg_time, double lidar_end_time,
                    const float* raw_pts, int n, int stride_bytes, int time_offset_bytes,
                    const lio_imu_sample* samples, int n_samples, int* undistorted) {
    if (!imu || !kf || n_samples < 0 || (n_samples > 0 && !samples)) return LIO_ERR_INVALID_ARG;
    if (ctx && (n < 0 || (n > 0 && !raw_pts))) return LIO_ERR_INVALID_ARG;
    lio::MeasureGroup meas;
    meas.lidar_beg_time = lidar_beg_time;
    meas.lidar_end_time = lidar_end_time;
    meas.lidar = raw_pts;
    meas.n_points = n;
    meas.stride_bytes = stride_bytes;
    meas.time_offset_bytes = time_offset_bytes;
    meas.imu.resize((size_t)n_samples);
    for (int i = 0; i < n_samples; ++i) {
        meas.imu[i].stamp = samples[i].stamp;
        for (int k = 0; k < 3; ++k) { meas.imu[i].acc[k] = samples[i].acc[k]; meas.imu[i].gyr[k] = samples[i].gyr[k]; }
    }
    int und = 0;
    const int rc = imu->p.Process(meas, kf->kf, ctx, &und);
    if (undistorted) *undistorted = und;
    // a sweep prefetched for this call (lio_scan_prefetch_raw) that Process did not upload -- it returns before UndistortPcl while the
    // IMU initialises -- must not answer for a later sweep in the same driver buffer
    if (ctx && !und) ctx->raw_prefetched = false;
    return rc;
}
int lio_imu_get_poses(lio_imu* imu, lio_pose6d* out, int cap, int* n) {
    if (!imu || !n) return LIO_ERR_INVALID_ARG;
    const std::vector<lio_pose6d>& v = imu->p.poses();
    *n = (int)v.size();
    if (out) {
        if (cap < (int)v.size()) return LIO_ERR_CAPACITY;
        for (size_t i = 0; i < v.size(); ++i) out[i] = v[i];
    }
    return LIO_OK;
}
int lio_imu_get_info(lio_imu* imu, lio_imu_info* info) {
    if (!imu || !info) return LIO_ERR_INVALID_ARG;
    info->need_init = imu->p.need_init() ? 1 : 0;
    info->init_iter_num = imu->p.init_iter();
    for (int i = 0; i < 3; ++i) {
        info->mean_acc[i] = imu->p.mean_acc[i]; info->mean_gyr[i] = imu->p.mean_gyr[i];
        info->cov_acc[i] = imu->p.cov_acc[i]; info->cov_gyr[i] = imu->p.cov_gyr[i];
    }
    info->last_lidar_end_time = imu->p.last_lidar_end_time();
    return LIO_OK;
}

int lio_state_boxminus(const double x26[26], const double y26[26], double dx23[23]) {
    if (!x26 || !y26 || !dx23) return LIO_ERR_INVALID_ARG;
    lio::state_ikfom::from_array(x26).boxminus(dx23, lio::state_ikfom::from_array(y26));
    return LIO_OK;
}

}  // extern "C"
