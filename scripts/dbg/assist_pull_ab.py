"""In-process A/B of the assisted pull (lio_debug_set "assist_pull") on the drop-in callback chain (examples/callback_bench.cpp, the
two-line-patch form): the drive of bench.py at C2 (or C3), one context, the shares in turn, `reps` alternations.
    python scripts/dbg/assist_pull_ab.py [c2|c3] [reps] [steps] shares...      (0 = off, 1 = by the running estimate, 2..90 = per cent)"""
import sys, os, time, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import bench
from better_fastlio2_amd import capi, synth

cfg = sys.argv[1] if len(sys.argv) > 1 else "c2"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 200
shares = [int(v) for v in sys.argv[4:]] or [0, 1, 50, 80]
nm, nq, voxel, kind, max_iter = bench.CONFIGS[cfg]
world = synth.make_world(nm, voxel)
ctx = capi.LioContext(voxel, max_scan_points=nq, max_map_points=nm + 1_000_000)
ctx.map_build(world.points)
P0 = np.ascontiguousarray(synth.init_cov(), np.float64)
B = capi.callback_bench()
B.cbb_setup(ctx.h, nq)
priors, ids = [], []
px, py, yaw = 3.1, -2.2, 0.52
n_rot = int(os.environ.get("ROTATE", "8"))
for k in range(n_rot):
    pose = ((px + 0.9 * k, py + 0.4 * k, 0.0), (0.004, -0.006, yaw + 0.05 * k))
    sc = synth.make_scan(world, nq, kind=kind, seed=synth.SEED + 1 + 17 * k, pos=pose[0], rpy=pose[1])
    ps, rt = synth.perturb_state(sc, seed=synth.SEED + 2 + 7 * k)
    priors.append(np.ascontiguousarray(capi.make_state26(ps, rt, sc.off_R, sc.off_T), np.float64))
    b = np.ascontiguousarray(sc.body, np.float32)
    ids.append(B.cbb_add_scan(b.ctypes.data_as(C.c_void_p), len(b), 3))
ext = int(sc.extrinsic_est_en)
kf = capi.Esekf(max_iter)
ulog = capi.lio_update_log()
tick = [0]

def step():
    k = tick[0] % n_rot
    tick[0] += 1
    rc = B.cbb_step(kf.h, ids[k], priors[k].ctypes.data_as(C.c_void_p), P0.ctypes.data_as(C.c_void_p), 0.001, ext, 1, C.byref(ulog))
    assert rc == 0, ctx.L.lio_last_error(ctx.h)

chunk = int(os.environ.get("CHUNK", "0"))
ctx.scan_pack_threads(int(os.environ.get("THREADS", "1")))
ctx.debug_set("assist_pull_chunk", chunk)
ref = None
res = {v: [] for v in shares}
for r in range(reps):
    for v in shares:
        ctx.debug_set("assist_pull", v)
        for _ in range(16):
            step()
        tick[0] = 0
        step(); x = kf.get_x().copy()
        if ref is None: ref = x
        assert np.array_equal(x, ref), "posterior of scan 0 changed with the share"
        ctx.synchronize(); t0 = time.perf_counter()
        for _ in range(steps):
            step()
        ctx.synchronize(); dt = time.perf_counter() - t0
        res[v].append((dt / steps * 1e6, ctx.debug_get("last_pack_front"), ctx.debug_get("last_pack_ns") / 1e3, ctx.debug_get("pack_helper_late"), ctx.debug_get("assist_share_permille") / 10.0))
for v in shares:
    print("%s rotate %d chunk %d threads %s assist_pull %2d: step us %s | scans/s (best) %.0f | last front %d records packed in %.1f us (helper late on %d scans so far); steered share %.0f %%" % (
        cfg, n_rot, chunk, os.environ.get('THREADS', '1'), v, " ".join("%.1f" % m[0] for m in res[v]), 1e6 / min(m[0] for m in res[v]), res[v][-1][1], res[v][-1][2], res[v][-1][3], res[v][-1][4]), flush=True)
