#!/usr/bin/env python3
"""bench.py -- scans/sec (+ mean kNN us/point) of the LIO measurement path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config c2|c3|c5]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W          (N > 1: one rank per GPU)

A "step" is one scan step of the reference (everything inside kf.update_iterated_dyn_share_modified,
src/laserMapping.cpp:2380): up to NUM_MAX_ITERATIONS+1 passes of the measurement model (k-NN +
plane fit + residual + Jacobian + H^T H reduction on the GPU, 23x23 algebra on the host), on a
synthetic KITTI-64-like 100 k-point scan against a 5 M-point map (BASELINE.json configs[1];
SURVEY.md section 8d generators, seed 20240926).  Scan and map are resident in HBM before the timed
region.  N > 1: every rank owns one GPU, one replica of the map and its own scan stream (weak
scaling, no data-path collective inside a scan step -- SURVEY.md section 8e).

Prints ONE JSON line (rank 0).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

CONFIGS = {
    # name: (N_m, N_q, voxel, scan kind, NUM_MAX_ITERATIONS)
    "c2": (5_000_000, 100_000, 0.2, "kitti64", 3),
    "c3": (2_000_000, 24_000, 0.2, "hap", 3),
    "c5": (50_000_000, 1_000_000, 0.1, "wide", 3),
}
HBM_PEAK_GBS = 8000.0            # MI355X HBM3E spec peak (MI355X_MICROARCH.md)


def knn_algorithmic_bytes(nq: int, cand_sum: int) -> float:
    """per query 16 (query in) + 216 (27 cell headers x 8 B) + 16 C(q)
    (candidate float4 gathers) + 40 (5 x (idx + sqdist) out)."""
    return nq * (16.0 + 216.0 + 40.0) + 16.0 * cand_sum


def dist_setup(n_gpus: int):
    """(rank, world, local_rank, dist or None).  Re-launches itself under torchrun when asked for
    N > 1 from a plain `python bench.py --gpus N`."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if n_gpus > 1 and world == 1:
        port = 29500 + (os.getpid() % 2000)
        os.execvp(sys.executable, [sys.executable, "-m", "torch.distributed.run", "--nnodes=1",
                                   f"--nproc-per-node={n_gpus}", "--master-addr", "127.0.0.1",
                                   "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:])
    if world == 1 and "RANK" not in os.environ:
        return 0, 1, 0, None
    import torch
    import torch.distributed as dist
    local = int(os.environ.get("LOCAL_RANK", "0"))
    # every rank drives its GPU from one polling host thread: give the ranks disjoint core sets so that two
    # of them never share a core
    try:
        cores = sorted(os.sched_getaffinity(0))
        per = len(cores) // max(1, int(os.environ.get("LOCAL_WORLD_SIZE", str(world))))
        if per >= 1:
            os.sched_setaffinity(0, set(cores[local * per:(local + 1) * per]))
    except (AttributeError, OSError):
        pass
    torch.cuda.set_device(local)
    # RCCL prints a version banner on stdout when the communicator is created; create it now, with
    # stdout pointed at stderr, so that rank 0's stdout carries nothing but the JSON line
    sys.stdout.flush()
    saved = os.dup(1)
    os.dup2(2, 1)
    try:
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local))
        warm = torch.zeros(1, device=torch.device("cuda", local))
        dist.all_reduce(warm)
        torch.cuda.synchronize(local)
    finally:
        import ctypes
        ctypes.CDLL(None).fflush(None)
        os.dup2(saved, 1)
        os.close(saved)
    return dist.get_rank(), dist.get_world_size(), local, dist


def timed_steps(step_fn, steps: int, warmup: int, dist, device):
    """W untimed steps, then exactly K steps bracketed by barrier + synchronize; MAX over ranks."""
    import torch
    for _ in range(warmup):
        step_fn()
    torch.cuda.synchronize(device)
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize(device)
    t0 = time.perf_counter()
    for _ in range(steps):
        step_fn()
    torch.cuda.synchronize(device)
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize(device)
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    return elapsed


def cpu_baseline(world_pts, scan, x0, P0, max_iter: int, ext: bool, sample_every: int):
    """The reference CPU path timed on this host: the reference's own ikd-Tree (oracle/_ref, compiled
    verbatim) for Nearest_Search + the oracle restatement of h_share_model / IKFoM update, one thread,
    on every `sample_every`-th point of the scan against the full map."""
    from oracle import pyoracle as po
    body = np.ascontiguousarray(scan.body[::sample_every])
    om = po.OracleMap(0.2)
    use_ref = po.ref_available()
    t_build0 = time.perf_counter()
    if use_ref:
        tree = po.RefIkdTree(0.2)
        tree.build(world_pts)
        po.use_reference_knn(tree)
        method = 2
    else:
        om.build(world_pts)
        method = 1
    t_build = time.perf_counter() - t_build0
    ts = []
    for _ in range(5):
        t0 = time.perf_counter()
        xo, Po, arrs, log = po.update_iterated(om, x0, P0, body, 0.001, max_iter, None, ext, knn_method=method)
        ts.append(time.perf_counter() - t0)
    # mean kNN cost per point on the same sample
    st = np.concatenate([x0[0:14]])
    wq = po.body_to_world(st, body)
    if use_ref:
        _, _, _, secs = tree.knn(wq)
        po.use_reference_knn(None)
        del tree
    else:
        _, _, _, secs = om.knn(wq, 5, 1)
    t_step = float(np.median(ts)) * sample_every
    return {"value": 1.0 / t_step, "unit": "scans/s", "cores": 1, "kind": "port",
            "knn": "reference ikd-Tree (oracle/_ref, verbatim)" if use_ref else "oracle kd-tree restatement",
            "knn_us_per_point": secs / len(wq) * 1e6,
            "sample": (f"the whole scan ({len(body)} points)" if sample_every == 1 else
                       f"every {sample_every}th point of the scan ({len(body)} of {len(scan.body)}), scaled x{sample_every},")
                      + f" vs the full {len(world_pts)}-point map, {len(log)} passes, median of 5; tree build {t_build:.1f}s untimed",
            "passes": int(len(log)), "posterior": xo}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--config", default="c2", choices=sorted(CONFIGS))
    ap.add_argument("--lanes", type=int, default=0, help="kNN lanes per query (0 = auto)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--map-update", action="store_true",
                    help="also time map maintenance per step: map_incremental on every rank, all-gather of the insert "
                         "lists (RCCL over xGMI) and their application on every replica (BASELINE.json configs[3])")
    a = ap.parse_args()

    rank, world_size, local, dist = dist_setup(a.gpus)
    import torch
    from better_fastlio2_amd import capi, synth
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the measurement path has no CPU fallback")
    device = torch.device("cuda", local)

    nm, nq, voxel, kind, max_iter = CONFIGS[a.config]
    world = synth.make_world(nm, voxel)                        # identical replica on every rank
    # independent scan streams: every rank looks at the world from its own pose
    rng = np.random.default_rng(1000 + rank)
    px, py = (3.1, -2.2) if rank == 0 else tuple(rng.uniform(-0.3, 0.3, 2) * world.ground_half)
    yaw = 0.52 if rank == 0 else float(rng.uniform(-3.1, 3.1))
    scan = synth.make_scan(world, nq, kind=kind, seed=synth.SEED + 1 + rank, pos=(px, py, 0.0), rpy=(0.004, -0.006, yaw))
    pos, rot = synth.perturb_state(scan, seed=synth.SEED + 2 + rank)
    x0 = capi.make_state26(pos, rot, scan.off_R, scan.off_T)
    P0 = synth.init_cov()
    ext = scan.extrinsic_est_en

    ctx = capi.LioContext(voxel, max_scan_points=nq, max_map_points=nm + (8 * nq * max(world_size, 1) * 32 if a.map_update else 1024),
                          device=local, lanes_per_query=a.lanes)
    ctx.map_build(world.points)
    body4 = np.zeros((nq, 4), np.float32)
    body4[:, :3] = scan.body
    d_body = torch.from_numpy(body4).to(device)                # scan resident in HBM
    kf = capi.Esekf(max_iter)

    # map-maintenance mode: a short drive (4 poses per rank) so that successive steps really add points
    drive = []
    if a.map_update:
        for k in range(4):
            sc = synth.make_scan(world, nq, kind=kind, seed=synth.SEED + 101 + 17 * rank + k,
                                 pos=(px + 0.9 * k, py + 0.4 * k, 0.0), rpy=(0.004, -0.006, yaw + 0.05 * k))
            ps, rt = synth.perturb_state(sc, dpos=0.05, drot_deg=0.3, seed=synth.SEED + 7 + k)
            b4 = np.zeros((nq, 4), np.float32)
            b4[:, :3] = sc.body
            drive.append((torch.from_numpy(b4).to(device), capi.make_state26(ps, rt, sc.off_R, sc.off_T)))
    step_no = [0]
    maint = {"n_add": 0, "n_nodown": 0, "seconds": 0.0}

    # The reference's host is C++; in the loop below the C entry points are called with arguments converted once
    # (the numpy -> ctypes conversions of the convenience wrappers cost microseconds per call, a scan step is ~160).
    import ctypes as C
    L = ctx.L
    x0_c = np.ascontiguousarray(x0, np.float64); P0_c = np.ascontiguousarray(P0, np.float64)
    x0_p, P0_p = x0_c.ctypes.data_as(C.c_void_p), P0_c.ctypes.data_as(C.c_void_p)
    d_body_p = C.c_void_p(d_body.data_ptr())
    ulog = capi.lio_update_log()
    ulog_ref = C.byref(ulog)
    ext_i = int(ext)

    def step():
        if a.map_update:
            d_b, xs = drive[step_no[0] % len(drive)]
            step_no[0] += 1
            kf.change_x(xs)
            kf.change_P(P0)
            ctx.scan_set_device(d_b.data_ptr(), nq)
            lg = kf.update_iterated_dyn_share_modified(ctx, 0.001, ext)
        else:
            rc = L.lio_eskf_scan_step_device(kf.h, ctx.h, x0_p, P0_p, d_body_p, nq, 0.001, ext_i, ulog_ref)
            if rc != 0:
                raise capi.LioError(f"scan step failed: {capi.ERRORS.get(rc, rc)}: {L.lio_last_error(ctx.h).decode()}")
            lg = ulog
        if a.map_update:
            t0 = time.perf_counter()
            post = kf.get_x()[:14]
            if dist is None:
                r = ctx.map_incremental(post, voxel, True, apply=True)
            else:
                # every rank contributes its lists; every replica applies all of them, rank 0 first.
                # The lists never leave HBM: device pointers -> RCCL all-gather -> device pointers.
                from better_fastlio2_amd import replicate
                pa, na, pb, nb = ctx.map_incremental_device(post, voxel, True)
                A, _ = replicate.exchange_inserts_device(pa, na, nq, device)
                B, _ = replicate.exchange_inserts_device(pb, nb, nq, device)
                torch.cuda.synchronize(device)                 # torch's stream -> the library's stream
                ctx.map_add_pair_device(A.data_ptr(), A.shape[0], B.data_ptr(), B.shape[0])
                r = {"n_add": na, "n_nodown": nb}
            maint["n_add"] += r["n_add"]; maint["n_nodown"] += r["n_nodown"]
            maint["seconds"] += time.perf_counter() - t0
        return lg

    # untimed: candidate census for the byte model, parity of the posterior with the truth
    ctx.census_enable(True)
    ctx.scan_set_device(d_body.data_ptr(), nq)
    g = ctx.measure(x0[:14], True, ext)
    ctx.census_enable(False)
    cand_sum, unresolved = g["n_candidates"], g["n_unresolved"]
    log = step()
    log = log if isinstance(log, dict) else {k: getattr(log, k) for k, _ in capi.lio_update_log._fields_}
    x_gpu = kf.get_x().copy()                                  # posterior of the benchmarked step (parity gate below)
    pos_err = float(np.linalg.norm(x_gpu[:3] - scan.pos))

    # untimed: per-family kernel times (a HIP event pair on every launch)
    ctx.profile_enable(True)
    ctx.profile_reset()
    for _ in range(10):
        step()
    kernels_us = {k: (v["total_ms"] / max(v["launches"], 1)) * 1e3 for k, v in ctx.profile().items() if v["launches"]}
    # timed region: HIP events only on the dominant kernel (knn27), on the library's stream
    # The event pair costs host time on the launch path (the dispatch waits for it), so only every SAMPLE-th step
    # of the timed region carries it: the kernel time is still measured live, over the timed region, on the
    # library's stream -- without taxing the other steps.
    SAMPLE = int(os.environ.get("LIO_BENCH_EVENT_EVERY", "8"))
    tick = [0]

    def timed_step():
        if SAMPLE > 0 and tick[0] % SAMPLE == 0:
            ctx.profile_enable(only=[capi.PROF_KNN])
        elif SAMPLE <= 0 or tick[0] % SAMPLE == 1:
            ctx.profile_enable(False)
        tick[0] += 1
        return step()

    ctx.profile_enable(False)
    ctx.profile_reset()
    elapsed = timed_steps(timed_step, a.steps, a.warmup, dist, device)
    prof = ctx.profile()
    ctx.profile_enable(False)
    # the warm-up launches are in the accumulators too; averages are per launch either way
    knn_ms = prof["knn27"]["total_ms"] / max(prof["knn27"]["launches"], 1)

    if rank != 0:
        if dist is not None:
            dist.destroy_process_group()
        return
    total_scans = world_size * a.steps
    bytes_knn = knn_algorithmic_bytes(nq, cand_sum)
    achieved = bytes_knn / (knn_ms * 1e-3) / 1e9 if knn_ms > 0 else 0.0
    traffic = None
    tpath = os.path.join(ROOT, "profiles", "knn_traffic.json")
    if os.path.exists(tpath):
        try:
            traffic = json.load(open(tpath)).get(a.config, {}).get("hbm_bytes_per_launch")
        except Exception:
            traffic = None
    out = {
        "metric": "scans/sec + mean kNN us/point, KITTI-64 100k-pt scan vs 5M-pt map",
        "value": total_scans / elapsed, "unit": "scans/s", "n_gpus": world_size, "steps": a.steps, "warmup": a.warmup,
        "ms_per_step": elapsed / a.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"{a.config}: {kind} scan N_q={nq} vs map N_m={nm} (voxel {voxel} m), k=5, "
                               f"NUM_MAX_ITERATIONS={max_iter}, extrinsic_est_en={ext}",
                   "passes_per_step": log["n_calls"], "search_passes_per_step": log["n_search"],
                   "parallelism": f"{world_size} independent scan stream(s), replicated map",
                   "lanes_per_query": a.lanes or "auto", "C_mean": cand_sum / nq, "unresolved_queries": unresolved,
                   "pos_err_to_truth_m": pos_err,
                   "map_update": ({"per_step_ms": maint["seconds"] / max(step_no[0], 1) * 1e3,
                                   "inserted_per_step": (maint["n_add"] + maint["n_nodown"]) / max(step_no[0], 1),
                                   "exchange": "all_gather_into_tensor (RCCL)" if dist is not None else "none (1 rank)",
                                   "incremental_index_updates": ctx.map_info()["n_incremental_updates"],
                                   "full_index_rebuilds": ctx.map_info()["n_full_rebuilds"]}
                                  if a.map_update else None)},
        "knn_us_per_point": knn_ms * 1e3 / nq,
        "kernels_us": kernels_us,
        "roofline": {"bound": "hbm", "kernel": "knn27_kernel", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                     "algorithmic_bytes_per_launch": bytes_knn, "avg_launch_us": knn_ms * 1e3},
    }
    if world_size == 1 and not a.no_cpu_baseline:
        # the reference's ikd-Tree printf()s from its re-balancing thread: keep stdout to the ONE JSON line
        sys.stdout.flush()
        saved = os.dup(1)
        os.dup2(2, 1)
        try:
            # the whole scan when a CPU scan step stays around a second (C2: 0.4 s), a sample of it otherwise
            cb = cpu_baseline(world.points, scan, x0, P0, max_iter, ext, sample_every=1 if nq <= 200_000 else 8)
            import gc
            gc.collect()                                       # ~RefIkdTree joins the thread (prints again)
        finally:
            sys.stdout.flush()
            import ctypes
            ctypes.CDLL(None).fflush(None)                     # the tree's printf()s sit in libc's buffer
            os.dup2(saved, 1)
            os.close(saved)
        # parity gate of this very measurement: the CPU path ran the same scan step from the
        # same prior -- its posterior against the GPU's (bar: 1e-4 m / 1e-5 rad)
        x_cpu = cb.pop("posterior")
        if cb["sample"].startswith("the whole scan") and not a.map_update:
            q1, q2 = x_gpu[3:7], x_cpu[3:7]
            out["parity"] = {"pos_diff_m": float(np.linalg.norm(x_gpu[:3] - x_cpu[:3])),
                             "rot_diff_rad": float(4.0 * np.arcsin(min(1.0, 0.5 * min(np.linalg.norm(q1 - q2), np.linalg.norm(q1 + q2))))),
                             "state_max_abs_diff": float(np.max(np.abs(x_gpu - x_cpu))),
                             "against": "oracle h_share_model + IKFoM update on the reference ikd-Tree, same prior, same scan"}
        out["cpu_baseline"] = cb
    print(json.dumps(out), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
