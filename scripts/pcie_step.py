"""Scan step with the scan handed over from host memory every step (lio_scan_set: packing + 1.6 MB H2D)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from better_fastlio2_amd import capi, synth
nm, nq = 5_000_000, 100_000
world = synth.make_world(nm, 0.2); scan = synth.make_scan(world, nq, seed=synth.SEED + 1, pos=(3.1, -2.2, 0.0), rpy=(0.004, -0.006, 0.52))
pos, rot = synth.perturb_state(scan, seed=synth.SEED + 2)
x0 = capi.make_state26(pos, rot, scan.off_R, scan.off_T); P0 = synth.init_cov()
ctx = capi.LioContext(0.2, max_scan_points=nq, max_map_points=nm + 1024); ctx.map_build(world.points)
kf = capi.Esekf(3)
body = np.ascontiguousarray(scan.body)
def step():
    kf.change_x(x0); kf.change_P(P0); ctx.scan_set(body)
    return kf.update_iterated_dyn_share_modified(ctx, 0.001, False)
for _ in range(20): step()
N = 200; t0 = time.perf_counter()
for _ in range(N): step()
dt = (time.perf_counter() - t0) / N
print("PCIe-inclusive scan step: %.1f us => %.0f scans/s" % (dt * 1e6, 1 / dt))
