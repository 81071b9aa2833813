"""A/B of knn27_kernel's candidate walk (KnnArgs.walk: 0 = plain, 1 / 4 = prefix-form list with that many candidates per
lane in flight) x lanes per query, per-launch event times (first search of a scan / bounded re-search), over scan sizes:
    python scripts/dbg/knn_walk_ab.py [c2|c3]
c2: 5 M-point map, kitti64 scans of 100 k / 50 k / 24 k / 8.6 k / 3 k points; c3: 2 M-point map, 24 k-point HAP scans."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from better_fastlio2_amd import capi, synth

which = sys.argv[1] if len(sys.argv) > 1 else "c2"
LANES = [int(v) for v in os.environ.get("AB_LANES", "8,16").split(",")]
WALKS = [int(v) for v in os.environ.get("AB_WALKS", "0,1,4").split(",")]
if which == "c3":
    nm, sizes, kind = 2_000_000, [24_000], "hap"
else:
    nm, sizes, kind = 5_000_000, [int(v) for v in os.environ.get("AB_SIZES", "100000,50000,24000,8600,3000").split(",")], "kitti64"
world = synth.make_world(nm, 0.2)
ctx = capi.LioContext(0.2, max_scan_points=max(sizes), max_map_points=nm + 1000)
ctx.map_build(world.points)
for nq in sizes:
    scans = []
    for k in range(4):
        sc = synth.make_scan(world, nq, kind=kind, seed=synth.SEED + 1 + 17 * k, pos=(3.1 + 0.9 * k, -2.2 + 0.4 * k, 0.0), rpy=(0.004, -0.006, 0.52 + 0.05 * k))
        ps, rt = synth.perturb_state(sc, seed=synth.SEED + 2 + 7 * k)
        # first search at the prior (0.1 m / 0.5 deg off), bounded re-search at the pose the update converges to (~ the truth)
        scans.append((sc, synth.state_vector(ps, rt, sc.off_R, sc.off_T), synth.state_vector(sc.pos, sc.rot, sc.off_R, sc.off_T)))
    best = {}
    for rep in range(2):
        for lanes in LANES:
            for walk in WALKS:
                ctx.debug_set("lanes_per_query", lanes)
                ctx.debug_set("knn_walk", walk)
                ctx.profile_enable(only=[capi.PROF_KNN, capi.PROF_KNN_HINT]); ctx.profile_reset()
                for it in range(12):
                    sc, st, st2 = scans[it % 4]
                    ctx.scan_set(sc.body)
                    ctx.measure(st, True, False)
                    ctx.measure(st2, True, False)
                p = ctx.profile()
                f = p["knn27"]["total_ms"] / p["knn27"]["launches"] * 1e3
                h = p["knn27_hint"]["total_ms"] / p["knn27_hint"]["launches"] * 1e3
                key = (lanes, walk)
                best[key] = (min(best.get(key, (1e9, 1e9))[0], f), min(best.get(key, (1e9, 1e9))[1], h))
    for (lanes, walk), (f, h) in sorted(best.items()):
        print("%s nq %6d  lanes %2d  walk %d   first %6.2f us   hinted %6.2f us" % (which, nq, lanes, walk, f, h), flush=True)
ctx.debug_set("lanes_per_query", 0); ctx.debug_set("knn_walk", -1)
