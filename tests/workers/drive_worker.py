"""A stand-alone process that owns ONE context on the GPU and drives pipelined scan steps (lio_eskf_scan_step_pipelined: passes armed
ahead of their pose, the next scan's first search armed across the scan boundary, uploads on the copy stream) plus a map update
every step -- what tests/test_gpu_robustness.py runs two of at once, stops with SIGSTOP, or starts under HIP_LAUNCH_BLOCKING /
GPU_MAX_HW_QUEUES.  Not a test by itself.

    python drive_worker.py <repo root> <out.npz> <n_steps> <nq> <progress file or -> [go file or -]

Writes out.npz: posterior of every step (x, 26 doubles), per-step wall times, counters of the pass control.  The progress file
receives "0" when the set-up is done and then the number of steps done (rewritten every 16 steps); with a go file the loop starts
when that file appears (a parent starts several workers' loops at the same moment)."""
import os
import sys
import time

import numpy as np

root, out, n_steps, nq, progress = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), sys.argv[5]
go_file = sys.argv[6] if len(sys.argv) > 6 else "-"
seed_off = 0
sys.path.insert(0, root)
from better_fastlio2_amd import capi, synth  # noqa: E402

w = synth.make_world(200_000, 0.2)
ctx = capi.LioContext(0.2, max_scan_points=nq, max_map_points=900_000)
ctx.map_build(w.points)
P0 = np.ascontiguousarray(synth.init_cov(), np.float64)
scans, priors, pinned = [], [], []
for k in range(6):
    sc = synth.make_scan(w, nq, seed=synth.SEED + 700 + k + seed_off, pos=(0.7 * k, 0.2 * k, 0.0), rpy=(0.002, -0.003, 0.03 * k))
    ps, rt = synth.perturb_state(sc, seed=synth.SEED + 800 + k + seed_off)
    scans.append(sc)
    priors.append(np.ascontiguousarray(capi.make_state26(ps, rt, sc.off_R, sc.off_T), np.float64))
    pinned.append(capi.pinned_points(sc.body))
kf = capi.Esekf(3)
xs = np.zeros((n_steps, 26))
lat = np.zeros(n_steps)
def report(n):
    if progress != "-":
        with open(progress + ".tmp", "w") as fh:
            fh.write(str(n))
        os.replace(progress + ".tmp", progress)


report(0)
while go_file != "-" and not os.path.exists(go_file):
    time.sleep(0.001)
ctx.scan_upload_async(pinned[0])
t_prev = time.perf_counter()
for i in range(n_steps):
    k = i % 6
    capi.scan_step_pipelined(kf, ctx, priors[k], P0, pinned[(k + 1) % 6], 0.001, False)
    xs[i] = kf.get_x()
    if i % 3 == 2:
        ctx.map_incremental(xs[i][:14], 0.2, True, apply=True, want_lists=False)     # (calls the armed next-scan search off: both paths get exercised)
    t = time.perf_counter()
    lat[i] = t - t_prev
    t_prev = t
    if i % 16 == 15:
        report(i + 1)
ctx.scan_swap()
ctx.synchronize()
counters = {k: ctx.debug_get(k) for k in ("gate_released", "gate_called_off", "gate_reenqueued", "next_scan_armed", "next_scan_released", "gate_in_kernel")}
np.savez(out, xs=xs, lat=lat, valid=ctx.map_validnum(), **{k: np.int64(v) for k, v in counters.items()})
ctx.close()
