"""The node's whole loop body on one MI355X, stage by stage (laserMapping.cpp:2256-2401): ImuProcess::Process
(host propagation + point compensation on the GPU) -> voxel filter -> iterated update -> map_incremental ->
lasermap_fov_segment, on a KITTI-64-like raw sweep (120 k points, 10 Hz, 200 Hz IMU) against the 5 M-point map.
Not the bench metric (that is the update alone on BASELINE's N_q = 100 k); it shows what a drop-in node pays per scan.

    python scripts/loop_body_step.py [n_scans]
"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from better_fastlio2_amd import capi, synth

n_scans = int(sys.argv[1]) if len(sys.argv) > 1 else 24
nm, n_raw = 5_000_000, 120_000
world = synth.make_world(nm, 0.2)
ctx = capi.LioContext(0.2, max_scan_points=n_raw, max_map_points=nm + 2_000_000)
ctx.map_build(world.points)
rng = np.random.default_rng(5)
scans = []
for k in range(6):
    sc = synth.make_scan(world, n_raw, seed=synth.SEED + 10 + k, pos=(3.0 + 0.8 * k, -2.0 + 0.1 * k, 0.0), rpy=(0.004, -0.006, 0.5 + 0.02 * k))
    t_ms = np.sort(rng.uniform(0, 99.0, n_raw)).astype(np.float32)
    # the sweep where a sensor driver would put it: x, y, z, t records in page-locked memory (read in place by the copy engine)
    scans.append((sc, capi.pinned_points(np.concatenate([sc.body, t_ms[:, None]], 1).astype(np.float32))))
def qmat(q):
    x, y, z, w = q
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                     [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])
imu = capi.ImuProcess(); imu.set_extrinsic(scans[0][0].off_T, qmat(np.asarray(scans[0][0].off_R, np.float64))); imu.set_cov(0.1, 0.1, 1e-4, 1e-4)
kf = capi.Esekf(3)
def samples(t0, t1):
    ts = np.arange(t0, t1, 0.005)
    return np.stack([np.concatenate([[t], [0, 0, 9.81], [0, 0, 0]]) for t in ts])
imu.process(kf, ctx, 0.0, 0.1, samples(0.0, 0.1), scans[0][1])            # static initialisation
stages = {k: [] for k in ("imu+undistort", "voxel", "update", "map_incremental", "fov_segment", "total")}
nq_list = []
for it in range(n_scans + 4):
    sc, raw = scans[it % len(scans)]
    pos, rot = synth.perturb_state(sc, seed=100 + it)
    x = kf.get_x(); x[0:3] = pos; x[3:7] = rot; x[14:17] = 0.0; kf.change_x(x); kf.change_P(synth.init_cov())
    beg = 0.1 * (it + 1)
    smp = samples(beg, beg + 0.1)
    ctx.synchronize()
    t0 = time.perf_counter()
    ready = imu.process(kf, ctx, beg, beg + 0.099, smp, raw); ctx.synchronize()
    t1 = time.perf_counter()
    nq = ctx.scan_voxel_filter(0.5); ctx.synchronize()
    t2 = time.perf_counter()
    # the stationary IMU stream moved nothing: put the prior back at the scan's pose (the scans are independent poses)
    x = kf.get_x(); x[0:3] = pos; x[3:7] = rot; x[14:17] = 0.0; kf.change_x(x)
    t2b = time.perf_counter()
    kf.update_iterated_dyn_share_modified(ctx, 0.001, False)
    t3 = time.perf_counter()
    xs = kf.get_x()
    ctx.map_incremental(xs[:14], 0.2, True, apply=True, want_lists=False); ctx.synchronize()
    t4 = time.perf_counter()
    ctx.fov_segment(xs[0:3], 1000.0, 100.0); ctx.synchronize()
    t5 = time.perf_counter()
    if it >= 4:
        for k, v in zip(stages, (t1 - t0, t2 - t1, t3 - t2b, t4 - t3, t5 - t4, (t5 - t0) - (t2b - t2))):
            stages[k].append(v * 1e6)
        nq_list.append(nq)
        assert ready
print("raw %d pts -> feats_down %d pts (voxel 0.5 m); map %d -> %d points" % (n_raw, int(np.median(nq_list)), nm, ctx.map_validnum()))
for k, v in stages.items():
    print("  %-16s median %8.1f us   (p90 %8.1f)" % (k, np.median(v), np.percentile(v, 90)))
print("LOOP BODY: %.0f scans/s on one GPU (median of %d scans)" % (1e6 / np.median(stages["total"]), len(nq_list)))
