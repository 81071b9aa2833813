"""Differential fuzzing of the scan preparation (lio_scan_undistort / lio_scan_voxel_filter / lio_scan_get_cloud)
against the oracle restatements: random cloud sizes and extents, leaf sizes, time stamps with ties, non-finite
points, pose lists of random length, both call orders (cloud read before / after the filter); half of the cases prepared
between lio_scan_prep_begin / _end (prefetched sweep, own stream) beside an insert batch on the main stream.

    python tests/fuzz/fuzz_scanprep.py [n_cases] [first_seed]
"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from better_fastlio2_amd import capi
from oracle import pyoracle as po


def bits_equal(a, b):
    return a.shape == b.shape and np.array_equal(np.ascontiguousarray(a).view(np.uint32), np.ascontiguousarray(b).view(np.uint32))


def poses_for(rng, n_poses, dt):
    P = np.zeros((n_poses, 22))
    w = rng.normal(0, 0.3, 3); acc = rng.normal(0, 0.5, 3); v0 = rng.normal(0, 2.0, 3)
    for k in range(n_poses):
        t = k * dt
        ang = np.linalg.norm(w) * t; ax = w / np.linalg.norm(w)
        Kx = np.array([[0, -ax[2], ax[1]], [ax[2], 0, -ax[0]], [-ax[1], ax[0], 0]])
        R = np.eye(3) + np.sin(ang) * Kx + (1 - np.cos(ang)) * Kx @ Kx
        P[k, 0] = t - (0.001 if k == 1 and rng.random() < 0.3 else 0.0)
        P[k, 1:4] = acc + rng.normal(0, 0.05, 3); P[k, 4:7] = w + rng.normal(0, 0.01, 3)
        P[k, 7:10] = v0 + acc * t; P[k, 10:13] = v0 * t + 0.5 * acc * t * t
        P[k, 13:22] = R.ravel()
    return P


def run_case(seed, ctx):
    global n_overlapped
    rng = np.random.default_rng(seed)
    n = int(rng.choice([1, 2, 15, 16, 17, 255, 1000, 4096, 20_000, 60_000]))
    extent = float(rng.choice([2.0, 20.0, 80.0]))
    xyz = (rng.normal(0, 1, (n, 3)) * [extent, extent, extent * 0.1] + rng.uniform(-50, 50, 3)).astype(np.float32)
    if rng.random() < 0.5:                                     # a dense clump: leaves with hundreds of points
        m = min(n, int(rng.integers(1, 800)))
        xyz[:m] = (xyz[0] + rng.normal(0, 0.05, (m, 3))).astype(np.float32)
    t_ms = rng.uniform(0, 100.0, n).astype(np.float32)
    if rng.random() < 0.5:
        t_ms = np.round(t_ms)                                  # many ties
    if rng.random() < 0.3:
        t_ms[: max(1, n // 10)] = 0.0
    if rng.random() < 0.45:
        t_ms = np.sort(t_ms)                                   # a sweep in firing order: the compact-key (leaf index only) sort of the filter
        if rng.random() < 0.25 and n > 4:
            i = int(rng.integers(1, n - 1))
            t_ms[i - 1], t_ms[i] = t_ms[i] + np.float32(1.0), t_ms[i - 1]   # ... but for one pair (the prediction fails: redone)
    raw = np.concatenate([xyz, t_ms[:, None]], 1).astype(np.float32)
    if n > 20 and rng.random() < 0.3:
        bad = rng.choice(n, 3, replace=False)
        raw[bad[0], 0] = np.nan; raw[bad[1], 1] = np.inf; raw[bad[2], 2] = -np.inf
    n_poses = int(rng.choice([2, 3, 12, 25]))
    poses = poses_for(rng, n_poses, 0.1 / (n_poses - 1) * float(rng.uniform(0.8, 1.1)))
    q = rng.normal(size=4); q /= np.linalg.norm(q)
    qo = np.array([0.01, -0.02, 0.03, 1.0]); qo /= np.linalg.norm(qo)
    end = np.concatenate([poses[-1, 10:13] + rng.normal(0, 0.01, 3), q, qo, rng.normal(0, 0.1, 3)])
    leaf = float(rng.choice([0.1, 0.2, 0.5, 1.0, 3.0]))
    read_first = bool(rng.integers(0, 2))
    # round 5: half of the cases take the OVERLAPPED route -- the sweep prefetched by the copy engine (when it lies in page-locked
    # memory), compensated and filtered between lio_scan_prep_begin / _end while an insert batch runs its own stamped scans and sorts
    # on the main stream -- and must come out exactly like the serial calls
    overlapped = rng.random() < 0.5
    und_o, ts = po.undistort(raw, poses, end)
    und_g = None
    try:
        if overlapped:
            src = capi.pinned_points(raw) if rng.random() < 0.7 else raw
            ctx.scan_prefetch_raw(src) if src is not raw else None
            ctx.scan_prep_begin()
            ctx.scan_undistort(src, poses, end)
            assert ctx.scan_voxel_filter(leaf) == -1, seed
            batch = (rng.uniform(-40, 40, (int(rng.integers(1, 6000)), 3)) * [1, 1, 0.05]).astype(np.float32)
            ctx.map_add(batch, bool(rng.integers(0, 2)))
            n_overlapped += 1
            m = ctx.scan_prep_end()
        else:
            ctx.scan_undistort(raw, poses, end)
            und_g = ctx.scan_cloud(0) if read_first else None
            m = ctx.scan_voxel_filter(leaf)
    except capi.LioError as e:
        # PCL's index overflow check, or more leaves than the scan buffer holds: the oracle must agree on the reason
        assert "LIO_ERR_CAPACITY" in str(e), (seed, str(e))
        try:
            ov = po.voxel_grid(und_o, leaf)
        except OverflowError:
            return "refused (index overflow)"
        assert len(ov) > ctx_cap, (seed, str(e), len(ov))
        return "refused (scan buffer)"
    if und_g is None:
        und_g = ctx.scan_cloud(0)
    assert und_g.shape == und_o.shape, (seed, "shape")
    fin = np.isfinite(und_o).all(1) & np.isfinite(und_g).all(1)
    assert np.array_equal(np.isfinite(und_o).all(1), np.isfinite(und_g).all(1)), (seed, "finite mask")
    scale = max(1.0, float(np.abs(und_o[fin]).max())) if fin.any() else 1.0
    assert (not fin.any()) or np.abs(und_g[fin] - und_o[fin]).max() < 2e-6 * scale + 1e-5, (seed, "undistort", np.abs(und_g[fin] - und_o[fin]).max())
    down_o = po.voxel_grid(und_g, leaf)                        # the filter is exact given the same input cloud
    down_g = ctx.scan_cloud(1)
    assert m == len(down_o) == len(down_g), (seed, "leaf count", m, len(down_o))
    assert bits_equal(down_g, down_o), (seed, "centroids")
    return "ok"


ctx_cap = 70_000
n_overlapped = 0                                 # cases that went through lio_scan_prep_begin / _end
if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 50
    s0 = int(sys.argv[2]) if len(sys.argv) > 2 else 30000
    ctx = capi.LioContext(0.2, max_scan_points=ctx_cap, max_map_points=2_000_000)
    ctx.map_build((np.random.default_rng(1).uniform(-40, 40, (40_000, 3)) * [1, 1, 0.05]).astype(np.float32))   # (the overlapped cases insert into it)
    res = {}
    for seed in range(s0, s0 + n):
        r = run_case(seed, ctx)
        res[r] = res.get(r, 0) + 1
    print("FUZZ SCANPREP OK: %d cases %s; compact-key attempts %d, redone %d; %d cases through lio_scan_prep_begin / _end (%d sweeps from the prefetch buffer, %d filtered beside the insert batch)"
          % (n, res, ctx.debug_get("voxel_fast_runs"), ctx.debug_get("voxel_redone"), n_overlapped, ctx.debug_get("raw_prefetch_hits"), ctx.debug_get("prep_overlapped")), flush=True)
