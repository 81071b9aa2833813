"""Where does knn27_kernel's time go?  The same map and kernel on four query sets of 100 k points:
the scan as it comes (ring order), the scan shuffled (no spatial coherence between neighbouring lanes), the scan sorted
by grid cell (maximal coherence), and 100 k copies of one point (every lane of a wavefront touches the same lines).
If the kernel were bound by instruction issue the four would take the same time; the spread is what the memory pipeline
(lines per wavefront-wide load) costs.

    python scripts/knn_coherence_probe.py
"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from better_fastlio2_amd import capi, synth

nm, nq = 5_000_000, 100_000
world = synth.make_world(nm, 0.2)
scan = synth.make_scan(world, nq, seed=synth.SEED + 1, pos=(3.1, -2.2, 0.0), rpy=(0.004, -0.006, 0.52))
pos, rot = synth.perturb_state(scan, seed=synth.SEED + 2)
st = capi.make_state26(pos, rot, scan.off_R, scan.off_T)[:14]
ctx = capi.LioContext(0.2, max_scan_points=nq, max_map_points=nm + 1024)
ctx.map_build(world.points)
rng = np.random.default_rng(1)
body = np.ascontiguousarray(scan.body)
# world-frame cell of every scan point (for the cell-sorted variant): use the library's own transform
ctx.scan_set(body); ctx.measure(st, True, False)
wpts = ctx.scan_state()[0]
variants = {"scan order": body, "shuffled": body[rng.permutation(nq)]}
if wpts is not None:
    cell = np.floor(wpts / 0.5).astype(np.int64)
    key = (cell[:, 0] << 42) + (cell[:, 1] << 21) + cell[:, 2]
    variants["sorted by cell"] = body[np.argsort(key, kind="stable")]
variants["one point x 100 k"] = np.repeat(body[nq // 2][None, :], nq, 0)
for name, b in variants.items():
    ctx.scan_set(np.ascontiguousarray(b))
    for _ in range(3):
        ctx.scan_set(np.ascontiguousarray(b)); ctx.measure(st, True, False)
    ctx.profile_enable(True); ctx.profile_reset()
    for _ in range(10):
        ctx.scan_set(np.ascontiguousarray(b))          # a fresh scan: no bound from previous neighbours
        g = ctx.measure(st, True, False)
    p = ctx.profile(); ctx.profile_enable(False)
    print("%-20s knn27 %6.2f us   tree %6.2f us   residual %6.2f us   (m = %d)" % (
        name, p["knn27"]["total_ms"] / p["knn27"]["launches"] * 1e3, p["knn_tree"]["total_ms"] / max(p["knn_tree"]["launches"], 1) * 1e3,
        p["residual"]["total_ms"] / p["residual"]["launches"] * 1e3, g["m"]), flush=True)
