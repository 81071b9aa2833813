"""Time of ONE full index rebuild of the C2 map (5 M points, a few incremental updates behind it): a small Add_Points call with the
retired-slot budget at 1 (the update + the rebuild it triggers) against the same call under the default budget (the update alone);
median of N.  LIO_REBUILD_BY_CELLS=0 in the environment = the 63-bit point sort of rounds 1-5.
    python scripts/rebuild_ab.py [n_map] [repeats] [batch]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from better_fastlio2_amd import capi, synth

n_map = int(sys.argv[1]) if len(sys.argv) > 1 else 5_000_000
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 15
nb = int(sys.argv[3]) if len(sys.argv) > 3 else 2000
w = synth.make_world(n_map)
ctx = capi.LioContext(0.2, max_scan_points=100_000, max_map_points=int(n_map * 1.2) + 200_000)
ctx.map_build(w.points)
rng = np.random.default_rng(7)
lo, hi = w.points.min(0), w.points.max(0)

def batch():
    c = np.array([rng.uniform(lo[0] + 20, hi[0] - 20), rng.uniform(lo[1] + 20, hi[1] - 20), 0.5])
    return (c + rng.uniform(-15, 15, (nb, 3)) * np.array([1, 1, 0.03])).astype(np.float32)

for _ in range(5):                                    # buckets at the tail, retired slots
    ctx.map_add(batch(), True)
t_upd, t_reb = [], []
for r in range(reps):
    b = batch()
    ctx.synchronize(); t0 = time.perf_counter(); ctx.map_add(b, True); ctx.synchronize(); t_upd.append(time.perf_counter() - t0)
    b = batch()
    ctx.debug_set("retired_budget", 1)
    i0 = ctx.map_info()["n_full_rebuilds"]
    ctx.synchronize(); t0 = time.perf_counter(); ctx.map_add(b, True); ctx.synchronize(); t_reb.append(time.perf_counter() - t0)
    assert ctx.map_info()["n_full_rebuilds"] == i0 + 1
    ctx.debug_set("retired_budget", 0)
    for _ in range(2):
        ctx.map_add(batch(), True)
u, f = np.median(t_upd) * 1e3, np.median(t_reb) * 1e3
info = ctx.map_info()
print("by_cells=%s n_map=%d live=%d cells=%d update %.3f ms, update + rebuild %.3f ms -> rebuild %.3f ms (min %.3f)" % (
    os.environ.get("LIO_REBUILD_BY_CELLS", "1"), n_map, info["n_live"], info["n_cells"], u, f, f - u, (np.min(t_reb) - np.min(t_upd)) * 1e3))
