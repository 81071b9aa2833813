"""Differential fuzzing of the measurement path (transform, k-NN, gate, plane fit, residual, Jacobian, H^T H reduction,
iterated update) against the CPU path: oracle h_share_model / IKFoM update with Nearest_Search of the reference
ikd-Tree.  Random small worlds (planes + clutter, at offsets up to kilometres), random scans (on the surfaces, noisy,
outliers, far points), random priors, both extrinsic modes, random scan sizes (reduction edges).

    python tests/fuzz/fuzz_measure.py [n_cases] [first_seed]
"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from better_fastlio2_amd import capi, synth
from oracle import pyoracle as po


def bits_equal(a, b):
    return a.shape == b.shape and np.array_equal(np.ascontiguousarray(a).view(np.uint32), np.ascontiguousarray(b).view(np.uint32))


def planes(rng, n, centre, extent, ds):
    pts = []
    for _ in range(rng.integers(3, 7)):
        o = centre + rng.uniform(-extent, extent, 3) * [1, 1, 0.15]
        u, v = rng.normal(size=3), rng.normal(size=3)
        u /= np.linalg.norm(u); v -= u * (u @ v); v /= np.linalg.norm(v)
        ab = rng.uniform(-extent, extent, (n // 4, 2))
        pts.append(o + ab[:, :1] * u + ab[:, 1:] * v + rng.normal(0, 0.01, (n // 4, 3)))
    p = np.concatenate(pts).astype(np.float32)
    key = np.floor(p / ds).astype(np.int64)                       # one point per down-sample voxel, like a real map
    _, first = np.unique(key, axis=0, return_index=True)
    return np.ascontiguousarray(p[np.sort(first)])


def quat_mul(a, b):
    x1, y1, z1, w1 = a; x2, y2, z2, w2 = b
    return np.array([w1 * x2 + x1 * w2 + y1 * z2 - z1 * y2, w1 * y2 - x1 * z2 + y1 * w2 + z1 * x2,
                     w1 * z2 + x1 * y2 - y1 * x2 + z1 * w2, w1 * w2 - x1 * x2 - y1 * y2 - z1 * z2])


def small_rot(rng, ang):
    ax = rng.normal(size=3); ax /= np.linalg.norm(ax)
    return np.concatenate([ax * np.sin(ang / 2), [np.cos(ang / 2)]])


def rotate(q, v):
    x, y, z, w = q
    R = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                  [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                  [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])
    return v @ R.T


TIES = [0]            # search passes of this run that met an exact tie among ranks 1..6


def run_case(seed):
    rng = np.random.default_rng(seed)
    ds = float(rng.choice([0.1, 0.2, 0.4]))
    centre = rng.uniform(-200, 200, 3) * [1, 1, 0.05]
    if seed % 4 == 0:
        centre = centre + rng.choice([-1, 1], 3) * [1500.0, 900.0, 40.0]
    world = planes(rng, int(rng.integers(20_000, 120_000)), centre, float(rng.uniform(8, 30)), ds)
    # true pose and extrinsics
    rot = small_rot(rng, rng.uniform(0, 3.1)); pos = centre + rng.uniform(-3, 3, 3)
    offR = small_rot(rng, rng.uniform(0, 0.3)); offT = rng.uniform(-0.3, 0.3, 3)
    # scan: world points seen from the pose (body = offR^T (rot^T (w - pos) - offT)), noise, outliers, far points
    nq = int(rng.choice([1, 5, 255, 256, 257, 1000, 4097, 9000]))
    wsel = world[rng.choice(len(world), nq)] + rng.normal(0, 0.02, (nq, 3))
    out = rng.random(nq) < 0.05
    wsel[out] += rng.normal(0, 3.0, (out.sum(), 3))
    conj = lambda q: np.array([-q[0], -q[1], -q[2], q[3]])
    body = rotate(conj(offR), rotate(conj(rot), wsel - pos) - offT).astype(np.float32)
    ext = bool(rng.integers(0, 2))
    lanes = int(rng.choice([0, 0, 4, 8, 16, 32, 64]))
    cells = [c for c in (0.125, 0.25, 0.5, 1.0, 2.0, 4.0) if c >= ds]          # power of two, not below the voxel
    cell = float(rng.choice([0.0, 0.0] + cells))                               # 0 = auto
    ctx = capi.LioContext(ds, max_scan_points=max(nq, 16), max_map_points=len(world) + 1000,
                          lanes_per_query=lanes, cell_size=cell)
    ctx.map_build(world); ctx.scan_set(body)
    om = po.OracleMap(ds); om.build(world)
    ref = po.RefIkdTree(ds); ref.build(world); po.use_reference_knn(ref)
    try:
        arr = po.ScanArrays(nq)
        st = np.concatenate([pos + rng.normal(0, 0.05, 3), quat_mul(rot, small_rot(rng, rng.uniform(0, 0.01))), offR, offT])
        tied = False
        for search in (True, False, True):
            st = st.copy(); st[:3] += rng.normal(0, 0.01, 3)
            valid, hx, h, m, rm = po.h_share_model(om, st, body, arr, search, ext, knn_method=2)
            g = ctx.measure(st, search, ext)
            assert g["m"] == m and g["valid"] == valid, (seed, "m", g["m"], m)
            wp, sel, nv = ctx.scan_state()
            assert bits_equal(wp, arr.world), (seed, "world")
            assert np.array_equal(sel, arr.selected), (seed, "selection")
            if m > 0:
                HTH, HTh = po.hth(hx, h)
                # Exactly equidistant map points (equal fp32 squared distances among ranks 1..6): which of them the reference returns,
                # and in which order, follows from its tree's traversal (strict `<`, ikd_Tree.cpp:887) -- here the map index decides.
                # Same distances, a different order of the rows of the 5 x 3 system (or another member of the tie): the plane of THAT
                # point moves in its last bits.  Such a scan is held to 1e-5 instead of 1e-9 (seed 42230: one tie between ranks 4 and
                # 5 in 9 000 queries, H^T H 2.6e-8 apart) and counted.
                if search:
                    _, d6, _, _ = om.knn(arr.world, 6, 1)
                    tied = bool(((np.diff(d6, axis=1) == 0) & np.isfinite(d6[:, 1:])).any())
                    TIES[0] += int(tied)
                tol = 1e-5 if tied else 1e-9
                assert np.abs(g["HTH"] - HTH).max() <= tol * max(np.abs(HTH).max(), 1e-300), (seed, "HTH")
                assert np.abs(g["HTh"] - HTh).max() <= (1e-3 if tied else 1e-9) * max(np.abs(HTh).max(), 1e-300), (seed, "HTh")
                assert abs(g["res_mean"] - rm) <= (1e-7 if tied else 1e-12), (seed, "res_mean")
            if search:
                nx, nd, nc = ctx.neighbors()
                assert np.array_equal(nc, arr.nbr_cnt) and bits_equal(nd, arr.nbr_d), (seed, "neighbour distances")
        # iterated update from a perturbed prior
        x0 = capi.make_state26(pos + rng.normal(0, 0.04, 3), quat_mul(rot, small_rot(rng, 0.004)), offR, offT)
        xo, Po, arrs, log = po.update_iterated(om, x0, synth.init_cov(), body, 0.001, 3, None, ext, knn_method=2)
        kf = capi.Esekf(3); kf.change_x(x0); kf.change_P(synth.init_cov())
        kf.update_iterated_dyn_share_modified(ctx, 0.001, ext)
        xg = kf.get_x()
        dx = xg - xo                                               # bar: 1e-4 m / 1e-5 rad (see fuzz_loop.py)
        assert np.abs(dx[:3]).max() < 1e-4 and 2.0 * np.abs(dx[3:7]).max() < 1e-5 and np.abs(dx[7:]).max() < 1e-4, (seed, "posterior", dx)
    finally:
        po.use_reference_knn(None)
        ctx.close()


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    s0 = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
    for seed in range(s0, s0 + n):
        run_case(seed)
    print("FUZZ MEASURE OK: %d cases (%d search passes with an exact distance tie among ranks 1..6: held to 1e-5)" % (n, TIES[0]))
