"""The first search of a scan is ~5 us slower once the map has had incremental updates (idle_probe.py: it is the map's state,
not the idle gap).  What in the state?  knn27 first-search times (event-stamped) and candidates per query (census)
    0  on the freshly built map          1  after 14 scans with map_incremental (no further updates while timing)
    2  after a forced full rebuild       3  in a NEW context built from the flattened points of that map
python scripts/dbg/post_update_probe.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from better_fastlio2_amd import capi, synth

nm, nq = 5_000_000, 100_000
world = synth.make_world(nm, 0.2)
scans = []
for k in range(4):
    sc = synth.make_scan(world, nq, seed=synth.SEED + 1 + 17 * k, pos=(3.1 + 0.9 * k, -2.2 + 0.4 * k, 0.0), rpy=(0.004, -0.006, 0.52 + 0.05 * k))
    ps, rt = synth.perturb_state(sc, seed=synth.SEED + 2 + 7 * k)
    scans.append((sc, synth.state_vector(ps, rt, sc.off_R, sc.off_T)))


def timing(ctx, tag):
    ctx.census_enable(True)
    cand = []
    for sc, st in scans:
        ctx.scan_set(sc.body)
        cand.append(ctx.measure(st, True, False)["n_candidates"] / nq)
    ctx.census_enable(False)
    ctx.profile_enable(only=[capi.PROF_KNN, capi.PROF_KNN_HINT, capi.PROF_RESIDUAL]); ctx.profile_reset()
    for it in range(12):
        sc, st = scans[it % 4]
        ctx.scan_set(sc.body)
        ctx.measure(st, True, False)
        st2 = st.copy(); st2[:3] += [0.03, -0.05, 0.01]
        ctx.measure(st2, True, False)
    p = ctx.profile(); ctx.profile_enable(False)
    info = ctx.map_info()
    print("%-46s first %.2f us  hinted %.2f us  search-pass measurement kernel %.2f us   C(q) mean %.1f   live %d slots %d retired %d rebuilds %d" % (
        tag, p["knn27"]["total_ms"] / p["knn27"]["launches"] * 1e3, p["knn27_hint"]["total_ms"] / p["knn27_hint"]["launches"] * 1e3,
        p["residual"]["total_ms"] / p["residual"]["launches"] * 1e3, np.mean(cand), info["n_live"], info["n_points"], info["n_retired"], info["n_full_rebuilds"]), flush=True)


ctx = capi.LioContext(0.2, max_scan_points=nq, max_map_points=nm + 2_000_000)
ctx.map_build(world.points)
timing(ctx, "0 fresh map")
for it in range(14):
    sc, st = scans[it % 4]
    ctx.scan_set(sc.body)
    ctx.measure(st, True, False)
    ctx.map_incremental(st, 0.2)
timing(ctx, "1 after 14 incremental updates")
ctx.debug_set("retired_budget", 1)
sc, st = scans[2]
ctx.scan_set(sc.body); ctx.measure(st, True, False); ctx.map_incremental(st, 0.2)
ctx.debug_set("retired_budget", 0)
timing(ctx, "2 after a forced full rebuild")
pts = ctx.map_flatten()
ctx2 = capi.LioContext(0.2, max_scan_points=nq, max_map_points=nm + 2_000_000)
ctx2.map_build(pts)
timing(ctx2, "3 new context, map_build(flatten())")
rng = np.random.default_rng(3)
ctx2.map_build(pts[rng.permutation(len(pts))])
timing(ctx2, "4 the same points, shuffled before the build")
