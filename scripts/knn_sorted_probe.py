"""What does grouping the queries of a scan by grid cell buy the k-NN kernel AS IT IS?  (Round 6, item 3 of the round-5 review: a
query-tile k-NN needs the queries binned by world cell.)  The scan of C2 / C3 is searched in arrival order and again with its points
sorted on the HOST by the cell key of their world position at the search pose -- binning for free -- through the same entry point;
the first search is event-stamped by the dispatch.  Same neighbours (checked), so the difference is what locality alone is worth; the
instructions a tile kernel would share are not touched.  Also printed: queries per occupied cell (what a tile could share).
    python scripts/knn_sorted_probe.py [c2 c3]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from better_fastlio2_amd import capi, synth

CONFIGS = {"c2": (5_000_000, 100_000, 0.2, "kitti64"), "c3": (2_000_000, 24_000, 0.2, "hap")}


def body_to_world(st, body):
    x, y, z, w = st[3:7]
    R = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                  [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                  [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])
    x, y, z, w = st[7:11]
    Ro = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                   [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                   [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])
    return (R @ (Ro @ body.T.astype(np.float64) + st[11:14, None]) + st[0:3, None]).T


for cfg in (sys.argv[1:] or ["c2", "c3"]):
    nm, nq, voxel, kind = CONFIGS[cfg]
    world = synth.make_world(nm, voxel)
    ctx = capi.LioContext(voxel, max_scan_points=nq, max_map_points=nm + 100_000)
    ctx.map_build(world.points)
    info = ctx.map_info()
    cell = info["cell_size"]
    res = {}
    for k in range(4):
        sc = synth.make_scan(world, nq, kind=kind, seed=synth.SEED + 1 + 17 * k, pos=(3.1 + 0.9 * k, -2.2 + 0.4 * k, 0.0), rpy=(0.004, -0.006, 0.52 + 0.05 * k))
        ps, rt = synth.perturb_state(sc, seed=synth.SEED + 2 + 7 * k)
        st = synth.state_vector(ps, rt, sc.off_R, sc.off_T)
        wq = body_to_world(st, sc.body)
        c = np.floor(wq / cell).astype(np.int64) + (1 << 20)
        key = (c[:, 0] << 42) | (c[:, 1] << 21) | c[:, 2]
        order = np.argsort(key, kind="stable")
        u, cnt = np.unique(key, return_counts=True)
        if k == 0:
            print("%s: %d queries in %d occupied cells (%.1f per cell, median %d; %.0f %% of the queries share their cell with >= 7 others)"
                  % (cfg, nq, len(u), nq / len(u), int(np.median(cnt)), 100.0 * cnt[cnt >= 8].sum() / nq))
        for name, body in (("arrival order", sc.body), ("sorted by cell ", np.ascontiguousarray(sc.body[order]))):
            ctx.scan_set(body)
            ctx.measure(st, True, sc.extrinsic_est_en)             # warm
            nx, nd, nc = ctx.neighbors()
            ctx.profile_enable(only=[capi.PROF_KNN])
            ctx.profile_reset()
            for _ in range(24):
                ctx.scan_set(body)                                   # (a fresh scan: no hint, a first search every time)
                ctx.measure(st, True, sc.extrinsic_est_en)
            p = ctx.profile()
            ctx.profile_enable(False)
            res.setdefault(name, []).append(p["knn27"]["total_ms"] / max(p["knn27"]["launches"], 1) * 1e3)
            if name.startswith("sorted"):
                assert np.array_equal(nd.view(np.uint32), res["_d"][order].view(np.uint32)), "the sorted scan found other neighbours"
            else:
                res["_d"] = nd
    for name in ("arrival order", "sorted by cell "):
        print("  %s first search %.2f us (scans: %s)" % (name, float(np.mean(res[name])), " ".join("%.2f" % v for v in res[name])))
    ctx.close()
