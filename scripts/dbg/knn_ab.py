"""A/B of knn27_kernel experiment switches on one box: per-launch event times over a drive of scans."""
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
ctx = capi.LioContext(0.2, max_scan_points=nq, max_map_points=nm + 1000)
ctx.map_build(world.points)
for rep in range(3):
    for tune in (0, 1):
        ctx.debug_set("knn_tune", tune)
        ctx.profile_enable(only=[capi.PROF_KNN, capi.PROF_KNN_HINT]); ctx.profile_reset()
        for it in range(12):
            sc, st = scans[it % 4]
            ctx.scan_set(sc.body)
            ctx.measure(st, True, False)
            st2 = st.copy(); st2[:3] += [0.03, -0.05, 0.01]
            ctx.measure(st2, True, False)
        p = ctx.profile()
        print("tune", tune, "first %.2f us  hinted %.2f us" % (p["knn27"]["total_ms"] / p["knn27"]["launches"] * 1e3, p["knn27_hint"]["total_ms"] / p["knn27_hint"]["launches"] * 1e3))
