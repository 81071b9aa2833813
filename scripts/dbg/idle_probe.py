"""Why does the first search pass after a map update run ~5 us longer per kernel than the same pass in a loop without
map updates?  Event-stamped times of knn27_kernel / the search pass's measurement kernel, C2 sizes, with
    A  nothing between two scan steps                      B  the host sleeps 300 us between them (an idle GPU)
    C  a map update between them (the node loop)           D  a map update, then a small throw-away search (8 k queries)
    E  ~150 us of small one-workgroup launches between them (a busy queue, an idle chip: the shape of a map update)
python scripts/dbg/idle_probe.py"""
import sys, os, time
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
small = np.ascontiguousarray(scans[0][0].body[:8000])
ctx = capi.LioContext(0.2, max_scan_points=nq, max_map_points=nm + 2_000_000)
ctx.map_build(world.points)
few = np.ascontiguousarray(world.points[:64] + np.float32(0.01))


def between(mode, st):
    if mode == "B":
        ctx.synchronize(); time.sleep(0.0003)
    elif mode in ("C", "D"):
        ctx.map_incremental(st, 0.2)
        if mode == "D":
            ctx.nearest_search(small)
    elif mode == "E":
        for _ in range(12):
            ctx.nearest_search(few)


for rep in range(2):
    for mode in "ABCDE":
        if mode in "CD":                       # same map for every mode
            ctx.map_build(world.points)
        res = []
        for it in range(14):
            sc, st = scans[it % 4]
            ctx.scan_set(sc.body)
            timed = it >= 2
            ctx.profile_enable(only=[capi.PROF_KNN, capi.PROF_RESIDUAL]) if timed else ctx.profile_enable(False)
            ctx.profile_reset()
            ctx.measure(st, True, False)
            if timed:
                p = ctx.profile()
                res.append((p["knn27"]["total_ms"] * 1e3, p["residual"]["total_ms"] * 1e3))
            ctx.profile_enable(False)
            ctx.measure(st, False, False)
            between(mode, st)
        r = np.array(res)
        print("mode %s  knn27 first search: median %.2f us (min %.2f)   measurement kernel of the search pass: median %.2f us (min %.2f)"
              % (mode, np.median(r[:, 0]), r[:, 0].min(), np.median(r[:, 1]), r[:, 1].min()), flush=True)
