"""The node's whole loop body on one MI355X (laserMapping.cpp:2256-2401): ImuProcess::Process (host propagation + point
compensation on the GPU) -> voxel filter -> iterated update -> map_incremental -> lasermap_fov_segment, on a KITTI-64-like raw
sweep (120 k points, 10 Hz, 200 Hz IMU) against the 5 M-point map.  Not the bench metric (that is the update alone on BASELINE's
N_q = 100 k); it shows what a drop-in node pays per scan.  Three loops:
  staged      a synchronisation behind every stage (stage attribution; convenience wrappers)
  serial      the same order as a node that keeps going runs it: no synchronisation between the stages, arguments converted once
  overlapped  sweep k + 1 prefetched by the copy engine while scan k is processed (lio_scan_prefetch_raw: the moment a driver has
              the sweep) and compensated + voxel-filtered on a stream of its own beside map_incremental / lasermap_fov_segment of
              scan k (lio_scan_prep_begin / _end)

    python scripts/loop_body_step.py [n_scans]

bench.py imports run() for the `loop_body` block of its default line.
"""
import ctypes as C
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def _qmat(q):
    x, y, z, w = q
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                     [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])


def _samples(t0, t1):
    ts = np.arange(t0, t1, 0.005)
    return np.stack([np.concatenate([[t], [0, 0, 9.81], [0, 0, 0]]) for t in ts])


def run(ctx, capi, synth, world, n_scans=24, n_raw=120_000, staged=True):
    """The three loops on `ctx` (its map is `world`, freshly built; it grows by the loops' inserts).  Returns a dict."""
    L = ctx.L
    rng = np.random.default_rng(5)
    scans = []
    for k in range(6):
        sc = synth.make_scan(world, n_raw, seed=synth.SEED + 10 + k, pos=(3.0 + 0.8 * k, -2.0 + 0.1 * k, 0.0), rpy=(0.004, -0.006, 0.5 + 0.02 * k))
        t_ms = np.sort(rng.uniform(0, 99.0, n_raw)).astype(np.float32)
        # the sweep where a sensor driver would put it: x, y, z, t records in page-locked memory (read in place by the copy engine)
        scans.append((sc, capi.pinned_points(np.concatenate([sc.body, t_ms[:, None]], 1).astype(np.float32))))
    imu = capi.ImuProcess()
    imu.set_extrinsic(scans[0][0].off_T, _qmat(np.asarray(scans[0][0].off_R, np.float64)))
    imu.set_cov(0.1, 0.1, 1e-4, 1e-4)
    kf = capi.Esekf(3)
    imu.process(kf, ctx, 0.0, 0.1, _samples(0.0, 0.1), scans[0][1])            # static initialisation
    out = {"raw_points": n_raw, "map_points_before": int(ctx.map_validnum())}
    if staged:
        stages = {k: [] for k in ("imu+undistort", "voxel", "update", "map_incremental", "fov_segment", "total")}
        nq_list = []
        for it in range(n_scans + 4):
            sc, raw = scans[it % len(scans)]
            pos, rot = synth.perturb_state(sc, seed=100 + it)
            x = kf.get_x(); x[0:3] = pos; x[3:7] = rot; x[14:17] = 0.0; kf.change_x(x); kf.change_P(synth.init_cov())
            beg = 0.1 * (it + 1)
            smp = _samples(beg, beg + 0.1)
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
        out["staged_us"] = {k: {"median": float(np.median(v)), "p90": float(np.percentile(v, 90))} for k, v in stages.items()}
        out["feats_down"] = int(np.median(nq_list))

    # ---- no synchronisation between the stages, arguments converted once (the reference's host is C++) ------------------------------
    na, nb, cnt, nbx, ndl, und, nout = (C.c_int(0) for _ in range(7))
    boxes = np.zeros((3, 6), np.float32)
    x26 = np.zeros(26)
    x26_p = x26.ctypes.data_as(C.c_void_p)
    ulog = capi.lio_update_log()
    P0 = np.ascontiguousarray(synth.init_cov(), np.float64)
    priors = []
    for it in range(len(scans)):
        pos, rot = synth.perturb_state(scans[it][0], seed=100 + it)
        priors.append((np.asarray(pos, np.float64), np.asarray(rot, np.float64)))
    smp_all = [np.ascontiguousarray(_samples(0.1 * (it + 1), 0.1 * (it + 1) + 0.1), np.float64) for it in range(n_scans + 8)]
    voxel_d, leaf = C.c_double(0.2), C.c_float(0.5)
    pos3 = np.zeros(3)
    n_total = n_scans + 4

    def err():
        return L.lio_last_error(ctx.h)

    def imu_process(it):
        raw = scans[it % len(scans)][1]
        beg = 0.1 * (it + 1)
        smp = smp_all[it]
        rc = imu.L.lio_imu_process(imu.h, kf.h, ctx.h, beg, beg + 0.099, raw.ctypes.data_as(C.c_void_p), len(raw), 16, 12,
                                   smp.ctypes.data_as(C.c_void_p), len(smp), C.byref(und))
        assert rc == 0 and und.value, (rc, err())

    def put_prior(it):
        # the stationary IMU stream moved nothing: the prior back at the scan's pose (the scans are independent poses)
        L.lio_eskf_get_x(kf.h, x26_p)
        x26[0:3], x26[3:7] = priors[it % len(scans)]
        x26[14:17] = 0.0
        L.lio_eskf_change_x(kf.h, x26_p)
        L.lio_eskf_change_P(kf.h, P0.ctypes.data_as(C.c_void_p))

    def update_and_map(it, overlapped):
        rc = L.lio_eskf_update_iterated_dyn_share_modified(kf.h, ctx.h, C.c_double(0.001), 0, C.byref(ulog))
        assert rc == 0, err()
        L.lio_eskf_get_x(kf.h, x26_p)
        post = x26.copy()
        if overlapped and it + 1 < n_total:
            # scan k + 1's preparation goes onto its own stream now (its sweep was prefetched during scan k's update) ...
            rc = L.lio_scan_prep_begin(ctx.h); assert rc == 0, err()
            imu_process(it + 1)
            rc = L.lio_scan_voxel_filter(ctx.h, leaf, C.byref(nout)); assert rc == 0, err()
        # ... and scan k's map maintenance runs beside it
        rc = L.lio_map_incremental(ctx.h, post.ctypes.data_as(C.c_void_p), voxel_d, 1, 1, None, C.byref(na), None, C.byref(nb), C.byref(cnt))
        assert rc == 0, err()
        pos3[:] = post[0:3]
        rc = L.lio_fov_segment(ctx.h, pos3.ctypes.data_as(C.c_void_p), C.c_double(1000.0), C.c_float(100.0), boxes.ctypes.data_as(C.c_void_p), C.byref(nbx), C.byref(ndl))
        assert rc == 0, err()

    for overlapped in (False, True):
        lat = []
        ctx.synchronize()
        if overlapped:
            L.lio_scan_prefetch_raw(ctx.h, scans[0][1].ctypes.data_as(C.c_void_p), n_raw, 16, 12)
            L.lio_scan_prep_begin(ctx.h); imu_process(0); L.lio_scan_voxel_filter(ctx.h, leaf, C.byref(nout))
        hits0, ov0 = ctx.debug_get("raw_prefetch_hits"), ctx.debug_get("prep_overlapped")
        t_prev = time.perf_counter()
        for it in range(n_total):
            if overlapped:
                rc = L.lio_scan_prep_end(ctx.h, C.byref(nout)); assert rc == 0, err()
                if it + 1 < n_total:
                    nxt = scans[(it + 1) % len(scans)][1]
                    L.lio_scan_prefetch_raw(ctx.h, nxt.ctypes.data_as(C.c_void_p), n_raw, 16, 12)    # sweep k + 1 "arrives" while scan k is processed
            else:
                imu_process(it)
                rc = L.lio_scan_voxel_filter(ctx.h, leaf, C.byref(nout)); assert rc == 0, err()
            put_prior(it)
            update_and_map(it, overlapped)
            t = time.perf_counter()
            lat.append(t - t_prev); t_prev = t
        ctx.synchronize()
        v = np.asarray(lat[4:]) * 1e6
        key = "overlapped" if overlapped else "serial"
        worst = np.argsort(v)[-4:][::-1]
        out[key] = {"median_us": float(np.median(v)), "p90_us": float(np.percentile(v, 90)), "scans_per_s": float(1e6 / np.median(v)), "scans": int(len(v)),
                    "feats_down": int(nout.value), "slowest_scans": [{"scan": int(i) + 4, "us": float(v[i])} for i in worst]}
        if overlapped:
            out[key].update({"prefetched_sweeps": int(ctx.debug_get("raw_prefetch_hits") - hits0),
                             "prefetch_hits_per_scan": float(ctx.debug_get("raw_prefetch_hits") - hits0) / n_total,
                             "preparations_filtered_beside_the_map_update": int(ctx.debug_get("prep_overlapped") - ov0)})
    out["map_points_after"] = int(ctx.map_validnum())
    return out


if __name__ == "__main__":
    from better_fastlio2_amd import capi, synth
    n_scans = int(sys.argv[1]) if len(sys.argv) > 1 else 24
    nm, n_raw = 5_000_000, 120_000
    world = synth.make_world(nm, 0.2)
    ctx = capi.LioContext(0.2, max_scan_points=n_raw, max_map_points=nm + 2_000_000)
    ctx.map_build(world.points)
    r = run(ctx, capi, synth, world, n_scans, n_raw)
    print("raw %d pts -> feats_down %d pts (voxel 0.5 m); map %d -> %d points" % (n_raw, r["feats_down"], r["map_points_before"], r["map_points_after"]))
    print("SERIAL loop, a synchronisation behind every stage (stage attribution):")
    for k, v in r["staged_us"].items():
        print("  %-16s median %8.1f us   (p90 %8.1f)" % (k, v["median"], v["p90"]))
    print("LOOP BODY: %.0f scans/s on one GPU (median of %d scans)" % (1e6 / r["staged_us"]["total"]["median"], n_scans))
    for key, name in (("serial", "SERIAL    "), ("overlapped", "OVERLAPPED")):
        v = r[key]
        extra = "  (prefetched sweeps %d, preparations filtered beside the map update %d)" % (v["prefetched_sweeps"], v["preparations_filtered_beside_the_map_update"]) if key == "overlapped" else ""
        print("%s loop, no synchronisation between the stages: median %.1f us per scan (p90 %.1f) = %.0f scans/s, feats_down %d%s"
              % (name, v["median_us"], v["p90_us"], v["scans_per_s"], v["feats_down"], extra))
