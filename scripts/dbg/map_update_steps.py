"""Per-call wall time of lio_map_incremental over the first steps of a node loop (C2 map, 100 k-point scans): where the slow calls are."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from better_fastlio2_amd import capi, synth
nm, nq = 5_000_000, 100_000
world = synth.make_world(nm, 0.2)
for mode in (1, 0):
    ctx = capi.LioContext(0.2, max_scan_points=nq, max_map_points=nm + 2_000_000)
    ctx.debug_set("fast_update", mode)
    ctx.map_build(world.points)
    kf = capi.Esekf(3)
    scans = []
    for k in range(8):
        sc = synth.make_scan(world, nq, seed=synth.SEED + 1 + 17 * k, pos=(3.1 + 0.9 * k, -2.2 + 0.4 * k, 0.0), rpy=(0.004, -0.006, 0.52 + 0.05 * k))
        ps, rt = synth.perturb_state(sc, seed=synth.SEED + 2 + 7 * k)
        scans.append((sc, capi.make_state26(ps, rt, sc.off_R, sc.off_T)))
    ts = []
    for it in range(40):
        sc, x0 = scans[it % 8]
        ctx.scan_set(sc.body)
        kf.change_x(x0); kf.change_P(synth.init_cov())
        kf.update_iterated_dyn_share_modified(ctx, 0.001, False)
        ctx.synchronize()
        t0 = time.perf_counter()
        r = ctx.map_incremental(kf.get_x()[:14], 0.2, True, apply=True, want_lists=False)
        ts.append((time.perf_counter() - t0) * 1e6)
    info = ctx.map_info()
    print("fast_update", mode, "calls us:", " ".join("%.0f" % t for t in ts))
    print("   fast", ctx.debug_get("fast_updates"), "overflows", ctx.debug_get("fast_update_overflows"), "rebuilds", info["n_full_rebuilds"], "updates", info["n_incremental_updates"])
    ctx.close()
