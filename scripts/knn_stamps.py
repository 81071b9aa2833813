"""The rungs of knn27_kernel's dependent chain, per workgroup (why a 24 k-query launch takes 12 us):
lio_debug_set("knn_stamps") runs the STAMPED instantiation of the kernel (same code, a full wait for memory in front of the stamps
that close a memory phase) and lio_debug_knn_stamps returns eight wall-clock stamps per workgroup.  Printed per config: when the
workgroups START (the launch ramp), how long every rung takes (median / p90 over workgroups), when the last workgroup ends.
Also the floor of a scan step: the same pipelined entry point on 256-point scans (six launches + four mail round trips, no work).

    python scripts/knn_stamps.py [c3] [c2] [node]
"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from better_fastlio2_amd import capi, synth

CFG = {"c3": (2_000_000, 24_000, "hap"), "c2": (5_000_000, 100_000, "kitti64"), "node": (5_000_000, 8_600, "kitti64")}
RUNGS = ["entry -> query known (transform wavefront + barrier)", "-> directory records arrived (8 hash probes, 27 cell records)",
         "-> bucket list compacted, prefix form in LDS", "-> candidates walked (16-byte gathers, top-5 ladders)",
         "-> merged over the lanes + proven", "-> results stored"]


def ladder(ctx, pose, label):
    n = np.zeros(1, np.int32)
    ctx.L.lio_debug_knn_stamps(ctx.h, None, 0, n.ctypes.data)
    wgs = int(n[0])
    buf = np.zeros((wgs, 8), np.uint64)
    ctx._chk(ctx.L.lio_debug_knn_stamps(ctx.h, buf.ctypes.data, buf.size, n.ctypes.data), "lio_debug_knn_stamps")
    t = buf[:, :7].astype(np.int64)
    t0 = t[:, 0].min()
    us = (t - t0) / 100.0
    return us


def run(name):
    nm, nq, kind = CFG[name]
    world = synth.make_world(nm, 0.2)
    scan = synth.make_scan(world, nq, kind=kind)
    pos, rot = synth.perturb_state(scan)
    st = synth.state_vector(pos, rot, scan.off_R, scan.off_T)
    st2 = st.copy(); st2[:3] += [0.05, -0.04, 0.01]
    ext = scan.extrinsic_est_en
    ctx = capi.LioContext(0.2, max_scan_points=nq, max_map_points=nm + 1000, lanes_per_query=8)
    ctx.map_build(world.points)
    ctx.scan_set(scan.body)
    for _ in range(3):
        ctx.measure(st, True, ext)
    # the kernel as the product runs it, event-stamped (no stamps inside)
    ctx.profile_enable(only=[capi.PROF_KNN, capi.PROF_KNN_HINT]); ctx.profile_reset()
    for _ in range(16):
        ctx.scan_set(scan.body); ctx.measure(st, True, ext); ctx.measure(st2, True, ext)
    prof = ctx.profile(); ctx.profile_enable(False)
    plain = prof["knn27"]["total_ms"] / max(prof["knn27"]["launches"], 1) * 1e3
    plain_h = prof["knn27_hint"]["total_ms"] / max(prof["knn27_hint"]["launches"], 1) * 1e3
    ctx.debug_set("knn_stamps", 1)
    first, hint = [], []
    for _ in range(12):
        ctx.scan_set(scan.body)
        ctx.measure(st, True, ext); first.append(ladder(ctx, st, "first"))
        ctx.measure(st2, True, ext); hint.append(ladder(ctx, st2, "bounded"))
    ctx.debug_set("knn_stamps", 0)
    print(f"== {name}: {kind} scan, {nq} queries ({first[0].shape[0]} workgroups of 32 queries, 8 lanes per query) vs {nm}-point map; "
          f"product kernel by dispatch-stamped events: first search {plain:.2f} us, bounded re-search {plain_h:.2f} us")
    for label, runs in (("first search", first[2:]), ("bounded re-search", hint[2:])):
        start = np.stack([r[:, 0] for r in runs])                 # [run][wg]
        end = np.stack([r[:, 6] for r in runs])
        span = np.median(end.max(axis=1))
        print(f"  {label} (stamped instantiation, median of {len(runs)} launches): first workgroup entry -> last store {span:.2f} us")
        print(f"    workgroup START after the first one: p50 {np.median(np.percentile(start, 50, axis=1)):.2f}  p90 {np.median(np.percentile(start, 90, axis=1)):.2f}  last {np.median(start.max(axis=1)):.2f} us")
        dur = np.stack([r[:, 6] - r[:, 0] for r in runs])
        print(f"    one workgroup, entry -> stored: p50 {np.median(np.percentile(dur, 50, axis=1)):.2f}  p90 {np.median(np.percentile(dur, 90, axis=1)):.2f}  max {np.median(dur.max(axis=1)):.2f} us")
        for k, rung in enumerate(RUNGS):
            d = np.stack([r[:, k + 1] - r[:, k] for r in runs])
            print(f"      {np.median(np.percentile(d, 50, axis=1)):6.2f} us (p90 {np.median(np.percentile(d, 90, axis=1)):5.2f})  {rung}")
    ctx.close()


def floor():
    # six launches and four mail round trips with nothing to do: the pipelined entry point on 256-point scans
    world = synth.make_world(200_000, 0.2)
    ctx = capi.LioContext(0.2, max_scan_points=4096, max_map_points=300_000)
    ctx.map_build(world.points)
    P0 = synth.init_cov()
    kf = capi.Esekf(3)
    scans = []
    for k in range(4):
        sc = synth.make_scan(world, 256, seed=synth.SEED + 40 + k, pos=(0.5 * k, 0.1 * k, 0.0))
        ps, rt = synth.perturb_state(sc, seed=synth.SEED + 50 + k)
        scans.append((capi.pinned_points(sc.body), capi.make_state26(ps, rt, sc.off_R, sc.off_T)))
    ctx.scan_upload_async(scans[0][0])
    logs = []
    for i in range(64):
        logs.append(capi.scan_step_pipelined(kf, ctx, scans[i % 4][1], P0, scans[(i + 1) % 4][0]))
    ctx.synchronize(); ctx.scan_upload_async(scans[0][0])
    import ctypes as C
    L = ctx.L
    x_p = [np.ascontiguousarray(s[1], np.float64) for s in scans]
    P_c = np.ascontiguousarray(P0, np.float64)
    ulog = capi.lio_update_log()
    n = 2000
    t0 = time.perf_counter()
    for i in range(n):
        L.lio_eskf_scan_step_pipelined(kf.h, ctx.h, x_p[i % 4].ctypes.data_as(C.c_void_p), P_c.ctypes.data_as(C.c_void_p),
                                       C.c_void_p(scans[(i + 1) % 4][0].ctypes.data), 256, 0.001, 0, C.byref(ulog))
    dt = (time.perf_counter() - t0) / n * 1e6
    ctx.scan_swap(); ctx.synchronize()
    print(f"== step floor: 256-point scans through lio_eskf_scan_step_pipelined, {logs[-1]['n_calls']} passes ({logs[-1]['n_search']} searches) per step: {dt:.1f} us per step")
    ctx.close()


if __name__ == "__main__":
    which = [a for a in sys.argv[1:] if a in CFG] or ["c3", "c2"]
    for w in which:
        run(w)
    floor()
