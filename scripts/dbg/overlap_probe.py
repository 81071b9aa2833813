"""Does work on a SECOND stream hide inside a scan step?  (the question behind a plan/commit split of the replicated map update:
the read-only half of Add_Points -- keys, sort, fold plan: ~70 us of mid-occupancy kernels for 8 ranks' lists -- enqueued on a side
stream at the start of the next scan step.)  Per step: K small torch kernels on a low-priority side stream, then the resident scan
step on the library's stream; step wall time and the side work's completion time against the plain step."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import ctypes as C
import numpy as np
import torch
from better_fastlio2_amd import capi, synth
nm, nq = 5_000_000, 100_000
world = synth.make_world(nm, 0.2)
ctx = capi.LioContext(0.2, max_scan_points=nq, max_map_points=nm + 1000)
ctx.map_build(world.points)
dev = torch.device("cuda", 0)
scans, priors, d = [], [], []
for k in range(8):
    sc = synth.make_scan(world, nq, seed=synth.SEED + 1 + 17 * k, pos=(3.1 + 0.9 * k, -2.2 + 0.4 * k, 0.0), rpy=(0.004, -0.006, 0.52 + 0.05 * k))
    ps, rt = synth.perturb_state(sc, seed=synth.SEED + 2 + 7 * k)
    priors.append(np.ascontiguousarray(capi.make_state26(ps, rt, sc.off_R, sc.off_T), np.float64))
    b4 = np.zeros((nq, 4), np.float32); b4[:, :3] = sc.body
    d.append(torch.from_numpy(b4).to(dev))
kf = capi.Esekf(3)
P0 = np.ascontiguousarray(synth.init_cov(), np.float64)
ulog = capi.lio_update_log()
L = ctx.L
side = torch.cuda.Stream(device=dev, priority=0)
N_SIDE = int(os.environ.get("SIDE_ELEMS", str(16_000_000)))
x = torch.zeros(N_SIDE, device=dev)               # a side kernel long enough (~20 us) that the host gets ahead of the side stream
y = torch.zeros(N_SIDE, device=dev)


def step(k):
    rc = L.lio_eskf_scan_step_device(kf.h, ctx.h, priors[k % 8].ctypes.data_as(C.c_void_p), P0.ctypes.data_as(C.c_void_p), C.c_void_p(d[k % 8].data_ptr()), nq, 0.001, 0, C.byref(ulog))
    assert rc == 0


def side_work(n_kernels):
    with torch.cuda.stream(side):
        for _ in range(n_kernels):
            torch.sin(x, out=y)                   # ~4-6 us each at half a million lanes: launch-floor sized, like the plan's kernels


for n_side in (0, 2, 4, 8):
    for k in range(16):
        step(k)
    torch.cuda.synchronize()
    ts, tside = [], []
    for k in range(160):
        ev = None
        if n_side:
            side_work(n_side)
            ev = torch.cuda.Event(); ev.record(side)
        t0 = time.perf_counter()                  # (the side launches are issued: their host time is not the question here)
        step(k)
        t1 = time.perf_counter()
        if ev is not None:
            ev.synchronize()
        t2 = time.perf_counter()
        ts.append(t1 - t0); tside.append(t2 - t0)
    ts, tside = np.array(ts) * 1e6, np.array(tside) * 1e6
    # the side kernels alone
    torch.cuda.synchronize()
    t0 = time.perf_counter(); side_work(max(n_side, 1) * 20); side.synchronize(); alone = (time.perf_counter() - t0) / 20 * 1e6
    print(f"{n_side:3d} side kernels per step (alone: {alone if n_side else 0:6.1f} us): scan step median {np.median(ts):6.1f} us (p90 {np.percentile(ts, 90):6.1f}), step + side work done {np.median(tside):6.1f} us")
