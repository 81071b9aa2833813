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
    const bool ok = rc == LIO_OK && std::fabs(x[0]) < 0.02 && std::fabs(x[1]) < 0.02 && std::fabs(x[2] - 1.7) < 0.02 && log.last_m > 5000;
    lio_eskf_destroy(kf);
    lio_destroy(ctx);
    std::printf(ok ? "DEMO OK\n" : "DEMO FAILED\n");
    return ok ? 0 : 1;
}
