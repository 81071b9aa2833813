// lio_demo.cpp -- the C ABI of include/lio_hip.h driven from plain C++ (no Python, no torch): the shape
// of the reference's per-scan loop body (src/laserMapping.cpp:2320-2401) on a small synthetic scene.
//
//   g++ -std=c++17 -O2 -Iinclude examples/lio_demo.cpp -o lio_demo \
//       -Lbetter_fastlio2_amd/lib -llio_hip -Wl,-rpath,$PWD/better_fastlio2_amd/lib
//   ./lio_demo            (needs an MI355X; prints LIO_ERR_NO_DEVICE and exits 2 otherwise)
#include <lio_hip.h>

#include <cmath>
#include <cstdio>
#include <random>
#include <vector>

int main() {
    std::printf("%s\n", lio_version());
    lio_config cfg{};
    cfg.device = 0; cfg.filter_size_map_min = 0.2f; cfg.cell_size = 0.0f;
    cfg.max_scan_points = 20000; cfg.max_map_points = 400000; cfg.lanes_per_query = 0;
    lio_ctx* ctx = nullptr;
    int rc = lio_create(&cfg, &ctx);
    if (rc != LIO_OK) { std::printf("lio_create failed: %d (no gfx950 device?)\n", rc); return 2; }

    // a 60 m x 60 m ground patch with two walls, one point per 0.2 m voxel, jittered
    std::mt19937 rng(7);
    std::uniform_real_distribution<float> jit(-0.08f, 0.08f), zn(-0.02f, 0.02f);
    std::vector<float> map;
    for (int i = 0; i < 300; ++i)
        for (int j = 0; j < 300; ++j) { map.push_back(-30.f + 0.2f * i + jit(rng)); map.push_back(-30.f + 0.2f * j + jit(rng)); map.push_back(zn(rng)); }
    for (int i = 0; i < 200; ++i)
        for (int k = 0; k < 15; ++k) { map.push_back(8.0f + zn(rng)); map.push_back(-20.f + 0.2f * i + jit(rng)); map.push_back(0.1f + 0.2f * k + jit(rng)); }
    for (int i = 0; i < 200; ++i)
        for (int k = 0; k < 15; ++k) { map.push_back(-20.f + 0.2f * i + jit(rng)); map.push_back(10.0f + zn(rng)); map.push_back(0.1f + 0.2f * k + jit(rng)); }
    rc = lio_map_build(ctx, map.data(), (long long)map.size() / 3, 12);
    if (rc != LIO_OK) { std::printf("lio_map_build: %s\n", lio_last_error(ctx)); return 1; }
    lio_map_info info;
    lio_map_get_info(ctx, &info);
    std::printf("map: %lld points, %lld cells of %.2f m\n", info.n_live, info.n_cells, info.cell_size);

    // a scan: the same surfaces seen from a LiDAR 1.7 m above the origin (body frame = world - (0,0,1.7))
    std::vector<float> scan;
    std::uniform_real_distribution<float> u(-25.f, 25.f);
    for (int i = 0; i < 12000; ++i) { scan.push_back(u(rng)); scan.push_back(u(rng)); scan.push_back(-1.7f + zn(rng)); }
    for (int i = 0; i < 3000; ++i) { scan.push_back(8.0f + zn(rng)); scan.push_back(0.6f * u(rng)); scan.push_back(-1.6f + 0.11f * (i % 25)); }
    for (int i = 0; i < 3000; ++i) { scan.push_back(0.6f * u(rng)); scan.push_back(10.0f + zn(rng)); scan.push_back(-1.6f + 0.11f * (i % 25)); }
    rc = lio_scan_set(ctx, scan.data(), (int)scan.size() / 3, 12);
    if (rc != LIO_OK) { std::printf("lio_scan_set: %s\n", lio_last_error(ctx)); return 1; }

    // prior: 8 cm / 0.3 deg off; identity extrinsics
    double x[26] = {0.06, -0.05, 1.72, 0, 0, std::sin(0.0026), std::cos(0.0026), 0, 0, 0, 1, 0, 0, 0,
                    0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, -9.809};
    double P[23 * 23] = {0};
    for (int i = 0; i < 23; ++i) P[i * 23 + i] = 1.0;
    double lim[23];
    for (double& v : lim) v = 0.001;
    lio_eskf* kf = lio_eskf_create();
    lio_eskf_init_dyn_share(kf, 3, lim);
    lio_eskf_change_x(kf, x);
    lio_eskf_change_P(kf, P);
    lio_update_log log{};
    rc = lio_eskf_update_iterated_dyn_share_modified(kf, ctx, 0.001, 0, &log);
    if (rc != LIO_OK) { std::printf("update: %s\n", lio_last_error(ctx)); return 1; }
    lio_eskf_get_x(kf, x);
    std::printf("update: %d passes (%d searches), m = %d, res_mean = %.4f\n", log.n_calls, log.n_search, log.last_m, log.last_res_mean);
    std::printf("posterior pos = %.4f %.4f %.4f (truth 0 0 1.7)\n", x[0], x[1], x[2]);

    int n_add = 0, n_nodown = 0, counter = 0;
    rc = lio_map_incremental(ctx, x, 0.2, 1, 1, nullptr, &n_add, nullptr, &n_nodown, &counter);
    std::printf("map_incremental: PointToAdd %d, NoNeedDownsample %d, Add_Points counter %d, map now %lld\n", n_add, n_nodown, counter,
                lio_map_validnum(ctx));
    bool ok = rc == LIO_OK && std::fabs(x[0]) < 0.02 && std::fabs(x[1]) < 0.02 && std::fabs(x[2] - 1.7) < 0.02 && log.last_m > 5000;

    // ---- a moving stream through the IMU front end (laserMapping.cpp:2256-2390): the sensor rests for one scan
    // (IMU_init), then accelerates along x and yaws; every scan = lio_imu_process (forward propagation on the host,
    // point compensation on the GPU) -> voxel filter -> iterated update -> map_incremental.
    lio_imu* imu = lio_imu_create();
    const double T0[3] = {0, 0, 0}, R0[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, c1[3] = {0.1, 0.1, 0.1}, c2[3] = {1e-4, 1e-4, 1e-4};
    lio_imu_set_extrinsic(imu, T0, R0);
    lio_imu_set_cov(imu, c1, c1, c2, c2);
    lio_eskf_destroy(kf);
    kf = lio_eskf_create();
    lio_eskf_init_dyn_share(kf, 3, lim);
    const double ax = 0.6, wz = 0.25, t_move = 0.105;           // m/s^2 and rad/s once the sensor moves (t >= t_move)
    auto truth = [&](double t, double pos[3], double& yaw) {
        const double tm = t > t_move ? t - t_move : 0.0;
        pos[0] = 0.5 * ax * tm * tm; pos[1] = 0.0; pos[2] = 1.7;
        yaw = wz * tm;
    };
    std::uniform_real_distribution<float> ut(0.0f, 99.0f);
    double err = 0.0;
    for (int k = 0; k < 7; ++k) {
        const double beg = 0.1 * k, end = beg + 0.099;
        std::vector<lio_imu_sample> samples;                   // 200 Hz
        for (double t = beg; t < end + 0.004; t += 0.005) {
            double pos[3], yaw; truth(t, pos, yaw);
            lio_imu_sample sm{};
            sm.stamp = t;
            const double awx = t >= t_move ? ax : 0.0;         // world acceleration; specific force = R^T (a - g)
            sm.acc[0] = std::cos(yaw) * awx; sm.acc[1] = -std::sin(yaw) * awx; sm.acc[2] = 9.81;
            sm.gyr[2] = t >= t_move ? wz : 0.0;
            samples.push_back(sm);
        }
        std::vector<float> raw;                                // x, y, z, time offset [ms]; taken while moving
        auto see = [&](float wx, float wy, float wzz) {
            const float tms = ut(rng);
            double pos[3], yaw; truth(beg + tms / 1000.0, pos, yaw);
            const double dx = wx - pos[0], dy = wy - pos[1], dz = wzz - pos[2];
            raw.push_back((float)(std::cos(yaw) * dx + std::sin(yaw) * dy));
            raw.push_back((float)(-std::sin(yaw) * dx + std::cos(yaw) * dy));
            raw.push_back((float)dz);
            raw.push_back(tms);
        };
        for (int i = 0; i < 12000; ++i) see(u(rng), u(rng), zn(rng));
        for (int i = 0; i < 3000; ++i) see(8.0f + zn(rng), 0.6f * u(rng), 0.1f + 0.11f * (i % 25));
        for (int i = 0; i < 3000; ++i) see(0.6f * u(rng), 10.0f + zn(rng), 0.1f + 0.11f * (i % 25));
        int ready = 0;
        rc = lio_imu_process(imu, kf, ctx, beg, end, raw.data(), (int)raw.size() / 4, 16, 12, samples.data(), (int)samples.size(), &ready);
        if (rc != LIO_OK) { std::printf("lio_imu_process: %s\n", lio_last_error(ctx)); return 1; }
        if (!ready) {                                          // initialised: place the filter in the map frame
            lio_eskf_get_x(kf, x);
            x[2] = 1.7;
            lio_eskf_change_x(kf, x);
            std::printf("scan %d: IMU initialised, gravity = %.4f %.4f %.4f\n", k, x[23], x[24], x[25]);
            continue;
        }
        int n_down = 0;
        rc = lio_scan_voxel_filter(ctx, 0.5f, &n_down);
        if (rc == LIO_OK) rc = lio_eskf_update_iterated_dyn_share_modified(kf, ctx, 0.001, 0, &log);
        if (rc != LIO_OK) { std::printf("scan step: %s\n", lio_last_error(ctx)); return 1; }
        lio_eskf_get_x(kf, x);
        rc = lio_map_incremental(ctx, x, 0.2, 1, 1, nullptr, &n_add, nullptr, &n_nodown, &counter);
        double pos[3], yaw; truth(end, pos, yaw);
        const double e = std::sqrt((x[0] - pos[0]) * (x[0] - pos[0]) + (x[1] - pos[1]) * (x[1] - pos[1]) + (x[2] - pos[2]) * (x[2] - pos[2]));
        const double eyaw = std::fabs(2.0 * std::atan2(x[5], x[6]) - yaw);
        err = e > err ? e : err;
        std::printf("scan %d: %d raw -> %d points, %d passes, m = %d, pos err %.4f m, yaw err %.5f rad, map %lld\n", k, (int)raw.size() / 4,
                    n_down, log.n_calls, log.last_m, e, eyaw, lio_map_validnum(ctx));
        ok = ok && rc == LIO_OK && e < 0.03 && eyaw < 0.003 && log.last_m > 1000;
    }
    lio_imu_destroy(imu);
    lio_destroy(ctx);
    std::printf(ok ? "DEMO OK\n" : "DEMO FAILED\n");
    return ok ? 0 : 1;
}
