"""Per-call wall times of a node loop (resident scan step + lio_map_incremental) over many steps: the slow calls, with the counters
around them.  Environment: LIO_FAST_UPDATE, LIO_PUBLISH_IN_KERNEL select the path."""
import os, sys, time, ctypes as C
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from better_fastlio2_amd import capi, synth
nm, nq = 5_000_000, 100_000
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
world = synth.make_world(nm, 0.2)
ctx = capi.LioContext(0.2, max_scan_points=nq, max_map_points=nm + 6_000_000)
ctx.map_build(world.points)
L = ctx.L
kf = capi.Esekf(3)
dev = torch.device("cuda", 0)
scans, xs = [], []
for k in range(8):
    sc = synth.make_scan(world, nq, seed=synth.SEED + 1 + 17 * k, pos=(3.1 + 0.9 * k, -2.2 + 0.4 * k, 0.0), rpy=(0.004, -0.006, 0.52 + 0.05 * k))
    ps, rt = synth.perturb_state(sc, seed=synth.SEED + 2 + 7 * k)
    b4 = np.zeros((nq, 4), np.float32); b4[:, :3] = sc.body
    scans.append(torch.from_numpy(b4).to(dev))
    xs.append(np.ascontiguousarray(capi.make_state26(ps, rt, sc.off_R, sc.off_T), np.float64))
P0 = np.ascontiguousarray(synth.init_cov(), np.float64)
ulog = capi.lio_update_log()
x_post = np.zeros(26); xp = x_post.ctypes.data_as(C.c_void_p)
na, nb, cnt = C.c_int(0), C.c_int(0), C.c_int(0)
keys = ("gate_released", "gate_called_off", "gate_reenqueued", "fast_updates", "fast_update_overflows")
def counters():
    d = {}
    for k in keys:
        try:
            d[k] = ctx.debug_get(k)
        except capi.LioError:
            d[k] = 0
    d["rebuilds"] = ctx.map_info()["n_full_rebuilds"]
    return d
t_step, t_map = [], []
slow = []
c0 = counters()
for it in range(steps):
    k = it % 8
    t0 = time.perf_counter()
    rc = L.lio_eskf_scan_step_device(kf.h, ctx.h, xs[k].ctypes.data_as(C.c_void_p), P0.ctypes.data_as(C.c_void_p), C.c_void_p(scans[k].data_ptr()), nq, 0.001, 0, C.byref(ulog))
    assert rc == 0
    t1 = time.perf_counter()
    L.lio_eskf_get_x(kf.h, xp)
    rc = L.lio_map_incremental(ctx.h, xp, C.c_double(0.2), 1, 1, None, C.byref(na), None, C.byref(nb), C.byref(cnt))
    assert rc == 0, L.lio_last_error(ctx.h)
    t2 = time.perf_counter()
    t_step.append(t1 - t0); t_map.append(t2 - t1)
    if it > 8 and (t1 - t0 > 1e-3 or t2 - t1 > 1e-3):
        c1 = counters()
        info = ctx.map_info()
        extra = {k: c1[k] - c0[k] for k in c1}
        extra.update({"n_calls": ulog.n_calls, "n_search": ulog.n_search, "measure_ms": round(ulog.measure_seconds * 1e3, 3), "solve_ms": round(ulog.solve_seconds * 1e3, 3),
                      "n_points": info["n_points"], "n_live": info["n_live"], "n_cells": info["n_cells"], "n_retired": info["n_retired"]})
        slow.append((it, (t1 - t0) * 1e3, (t2 - t1) * 1e3, extra))
    if it == 8:
        c0 = counters()
ts, tm = np.asarray(t_step[8:]) * 1e6, np.asarray(t_map[8:]) * 1e6
print("FAST_UPDATE=%s PUBLISH_IN_KERNEL=%s: %d steps | scan step p50 %.1f p99 %.1f max %.1f us | map update p50 %.1f p99 %.1f max %.1f us" % (
    os.environ.get("LIO_FAST_UPDATE", "1"), os.environ.get("LIO_PUBLISH_IN_KERNEL", "1"), len(ts), np.median(ts), np.percentile(ts, 99), ts.max(), np.median(tm), np.percentile(tm, 99), tm.max()))
for s in slow[:12]:
    print("   slow call at step %d: scan step %.2f ms, map update %.2f ms, counters since step 8: %s" % s)
print("   counters at the end:", counters())
