"""Does the k-NN slow down as re-written buckets pile up at the tail of the point array (retired-slot budget = 2 x live)?
A drive of `steps` scans, 1 m apart, each followed by map_incremental; every `every` steps the average first-search
k-NN time of the interval, then ONE forced full rebuild and the same scans' searches again on the re-sorted map."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from better_fastlio2_amd import capi, synth
steps, every, nq, nm = int(sys.argv[1]) if len(sys.argv) > 1 else 160, 20, 40000, 5_000_000
world = synth.make_world(nm, 0.2)
ctx = capi.LioContext(0.2, max_scan_points=nq, max_map_points=nm + 4_000_000)
ctx.map_build(world.points)
P0 = synth.init_cov()
kf = capi.Esekf(3)
x0 = -0.4 * world.ground_half
ctx.profile_enable(only=[capi.PROF_KNN, capi.PROF_KNN_HINT])
last = []
for k in range(steps):
    sc = synth.make_scan(world, nq, seed=synth.SEED + 900 + k, pos=(x0 + 1.0 * k, 3.0, 0.0), rpy=(0.003, -0.004, 0.3))
    ps, rt = synth.perturb_state(sc, seed=synth.SEED + 950 + k)
    x = capi.make_state26(ps, rt, sc.off_R, sc.off_T)
    buf = capi.scan_to_device(sc.body)
    capi.scan_step_device(kf, ctx, x, P0, buf.ptr, nq, 0.001, False)
    mi = ctx.map_incremental(kf.get_x()[:14], 0.2, True)
    last.append((x, buf, mi["n_add"] + mi["n_nodown"]))
    last = last[-every:]
    if (k + 1) % every == 0:
        p = ctx.profile(); ctx.profile_reset()
        info = ctx.map_info()
        print("step %4d  knn first %.2f us  hint %.2f us  inserted/step %.0f  retired %d live %d slots %d rebuilds %d" % (
            k + 1, p["knn27"]["total_ms"] / max(p["knn27"]["launches"], 1) * 1e3,
            p["knn27_hint"]["total_ms"] / max(p["knn27_hint"]["launches"], 1) * 1e3,
            np.mean([l[2] for l in last]), info["n_retired"], info["n_live"], info["n_points"], info["n_full_rebuilds"]), flush=True)

def replay(tag):
    ctx.profile_reset()
    for x, buf, _ in last:
        capi.scan_step_device(kf, ctx, x, P0, buf.ptr, nq, 0.001, False)
    p = ctx.profile()
    print(tag, "knn first %.2f us  hint %.2f us" % (p["knn27"]["total_ms"] / p["knn27"]["launches"] * 1e3,
                                                   p["knn27_hint"]["total_ms"] / p["knn27_hint"]["launches"] * 1e3), flush=True)
replay("replay of the last %d scans on the fragmented array:" % every)
ctx.debug_set("retired_budget", 1)
x, buf, _ = last[-1]
capi.scan_step_device(kf, ctx, x, P0, buf.ptr, nq, 0.001, False)
ctx.map_incremental(kf.get_x()[:14], 0.2, True)
ctx.debug_set("retired_budget", 0)
print("forced rebuild:", ctx.map_info()["n_full_rebuilds"], "retired", ctx.map_info()["n_retired"])
replay("replay on the re-sorted array:                 ")
