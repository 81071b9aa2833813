"""Second opinions for the parts of the oracle that cannot be pinned on the reference (they need
Eigen, which is neither vendored nor installed -- SURVEY.md section 8c): scipy / numpy in higher
precision as the independent check, with explicit tolerances."""
import numpy as np
import scipy.linalg as sl

from conftest import bits_equal


def _planar_sets(n, seed=1):
    rng = np.random.default_rng(seed)
    out = []
    for _ in range(n):
        nrm = rng.normal(size=3); nrm /= np.linalg.norm(nrm)
        base = rng.uniform(-60, 60, 3)
        u = np.cross(nrm, [1, 0, 0.3]); u /= np.linalg.norm(u)
        v = np.cross(nrm, u)
        pts = base + rng.uniform(-0.3, 0.3, (5, 1)) * u + rng.uniform(-0.3, 0.3, (5, 1)) * v \
            + rng.normal(0, rng.choice([0.004, 0.02, 0.06]), (5, 1)) * nrm
        out.append(pts.astype(np.float32))
    return out


def test_esti_plane_against_scipy_qr(oracle):
    """common_lib.h:506-536: least-squares plane through 5 points, fp32 column-pivoted QR.  The fp64
    least-squares solution is the truth; LAPACK's own fp32 pivoted QR (sgeqp3 via scipy) sets the
    error scale an fp32 Householder QR is allowed on these ill-conditioned systems (|p| ~ 60 m,
    spread ~ 0.3 m)."""
    err_o, err_l = [], []
    for pts in _planar_sets(3000):
        ok, p = oracle.esti_plane(pts, 0.1)
        A = pts.astype(np.float64)
        sol, *_ = sl.lstsq(A, -np.ones(5), lapack_driver="gelsy")
        n = np.linalg.norm(sol)
        ref = np.concatenate([sol / n, [1 / n]])
        Q, R, P = sl.qr(pts, mode="economic", pivoting=True)                 # float32 throughout
        y = sl.solve_triangular(R, Q.T @ (-np.ones(5, np.float32)))
        s32 = np.zeros(3, np.float32); s32[P] = y
        n32 = np.linalg.norm(s32)
        lap = np.concatenate([s32 / n32, [1 / n32]]).astype(np.float64)
        scale = max(1.0, abs(ref[3]))
        err_o.append(np.abs(p.astype(np.float64) - ref).max() / scale)
        err_l.append(np.abs(lap - ref).max() / scale)
        resid = np.abs(A @ ref[:3] + ref[3])
        if np.abs(resid - 0.1).min() > 1e-3:          # away from the inlier threshold
            assert ok == bool((resid <= 0.1).all())
    err_o, err_l = np.array(err_o), np.array(err_l)
    assert np.median(err_o) < 2e-4
    assert err_o.max() < 4 * err_l.max() + 1e-5
    assert np.mean(err_o) < 3 * np.mean(err_l) + 1e-6


def test_world_transform(oracle, small_case):
    from better_fastlio2_amd import synth
    sc = small_case["scan"]
    st = small_case["state14"]
    w = oracle.body_to_world(st, sc.body[:2000])
    R_wi = synth.quat_to_rot(st[3:7]); R_il = synth.quat_to_rot(st[7:11])
    ref = (R_wi @ (R_il @ sc.body[:2000].astype(np.float64).T + st[11:14, None]) + st[0:3, None]).T
    assert np.abs(w - ref).max() < 2e-5               # fp32 storage of ~60 m coordinates


def test_manifold_boxplus_boxminus_roundtrip(oracle):
    rng = np.random.default_rng(2)
    from better_fastlio2_amd import capi
    for _ in range(200):
        q = rng.normal(size=4); q /= np.linalg.norm(q)
        q2 = rng.normal(size=4); q2 /= np.linalg.norm(q2)
        x = capi.make_state26(rng.normal(size=3), q, q2, rng.normal(size=3), rng.normal(size=3),
                              rng.normal(size=3) * 0.01, rng.normal(size=3) * 0.01, (0.3, 0.2, -9.8009))
        x[23:26] *= 9.809 / np.linalg.norm(x[23:26])
        dx = rng.normal(size=23) * 0.05
        y = oracle.state_boxplus(x, dx)
        back = oracle.state_boxminus(y, x)
        assert np.abs(back - dx).max() < 1e-9
        assert abs(np.linalg.norm(y[23:26]) - 9.809) < 1e-9
        assert abs(np.linalg.norm(y[3:7]) - 1) < 1e-12


def test_iterated_update_reduces_pose_error(oracle, small_case):
    """End-to-end sanity of the restated filter: three iterations pull a 10 cm / 0.5 deg perturbation
    back to the centimetre level, and the information-form gain equals the textbook one."""
    from better_fastlio2_amd import capi, synth
    w, sc = small_case["world"], small_case["scan"]
    om = oracle.OracleMap(0.2); om.build(w.points)
    x0 = capi.make_state26(small_case["pos"], small_case["rot"], sc.off_R, sc.off_T)
    x1, P1, arr, log = oracle.update_iterated(om, x0, synth.init_cov(), sc.body)
    assert np.linalg.norm(x0[:3] - sc.pos) > 0.09
    assert np.linalg.norm(x1[:3] - sc.pos) < 0.02
    assert len(log) == 4 and log[0][0] == 1          # first pass always searches (esekfom.hpp:1624)
    assert np.allclose(P1, P1.T, atol=1e-12)
    assert np.linalg.eigvalsh(0.5 * (P1 + P1.T)).min() > -1e-12
    assert P1[0, 0] < 1e-4                            # position covariance collapsed from 1.0


def test_hth_matches_numpy(oracle):
    rng = np.random.default_rng(3)
    hx = rng.normal(size=(500, 12)); h = rng.normal(size=500)
    HTH, HTh = oracle.hth(hx, h)
    assert np.allclose(HTH, hx.T @ hx, rtol=1e-12, atol=1e-12)
    assert np.allclose(HTh, hx.T @ h, rtol=1e-12, atol=1e-12)


def test_map_incremental_classification_is_total(oracle, small_case):
    """map_incremental (laserMapping.cpp:1440-1496): every point is skip / add / add-no-downsample."""
    w, sc = small_case["world"], small_case["scan"]
    om = oracle.OracleMap(0.2); om.build(w.points)
    wq = oracle.body_to_world(small_case["state14"], sc.body[:3000])
    x, d, c, _ = om.knn(wq)
    cls = oracle.map_incremental_classify(wq, x, c, 0.2)
    assert set(np.unique(cls)) <= {0, 1, 2}
    assert (cls == 0).sum() > 0 and (cls == 1).sum() > 0


def test_eigen_summation_order_sensitivity(oracle, synth, small_case):
    """"Parity unpinned", quantified: the oracle adds the terms of every squaredNorm / dot product of
    esti_plane's QR sequentially; the reference's Eigen build adds them in a packet order that depends on its flags
    (common_lib.h:520, CMakeLists.txt sets none -> SSE2 (a0+a2)+(a1+a3), with -msse3 (a0+a1)+(a2+a3)) and on stack
    alignment.  Re-running one scan step under those orders and under random orders moves thousands of planes by an ulp
    or more -- and the pose by < 1e-6 m / 1e-7 rad, three orders inside the 1e-4 m / 1e-5 rad bar."""
    from better_fastlio2_amd import capi
    sc = small_case["scan"]
    x0 = capi.make_state26(small_case["pos"], small_case["rot"], sc.off_R, sc.off_T)
    om = oracle.OracleMap(0.2)
    om.build(small_case["world"].points)
    # the default order is untouched by the switch: same bits before and after a detour through another mode
    ok0, p0 = oracle.esti_plane(small_case["world"].points[:5])
    oracle.set_sum_order("random", 7)
    oracle.set_sum_order("sequential")
    ok1, p1 = oracle.esti_plane(small_case["world"].points[:5])
    assert ok0 == ok1 and np.array_equal(p0.view(np.uint32), p1.view(np.uint32))
    r = oracle.eigen_order_sensitivity(om, x0, synth.init_cov(), sc.body)
    variants = [k for k in r if isinstance(r[k], dict)]
    assert len(variants) == 5
    for k in variants:
        v = r[k]
        assert v["planes_with_any_bit_changed"] > 100, "the switch must really change the arithmetic"
        assert v["accept_mask_flips"] <= 5
        assert v["max_abs_dpd2_m"] < 1e-3
        assert v["posterior_dpos_m"] < 1e-6 and v["posterior_drot_rad"] < 1e-7


def _golden_plane_cases(n=4000):
    """The 4 000 neighbourhoods of tests/golden/make_model_golden.py (the generator only: the module itself binds the product)."""
    import os
    src = open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "make_model_golden.py")).read()
    ns = {"np": np}
    exec(src[src.index("def plane_cases"):src.index("def main")], ns)
    return ns["plane_cases"](np.random.default_rng(20240926), n)


def test_esti_plane_against_the_exact_referee(oracle):
    """"Parity unpinned" tightened as far as this image allows: the oracle's esti_plane restates Eigen's
    ColPivHouseholderQR from a reading of Eigen; oracle/referee.py pins it on the mathematics Eigen implements instead.  For each
    of the 4 000 golden neighbourhoods (noisy planes at map-like coordinates, near-collinear and near-coincident sets) the 5 x 3
    least-squares problem of common_lib.h:511-520 is solved EXACTLY (rationals), and the oracle's (n^, d) must be
      * the exact least-squares solution of a problem within 16 u32 ||A||_F of the given one (backward error after Walden, Karlson
        and Sun -- what any backward-stable fp32 QR delivers, whatever its pivoting or summation order),
      * hence within the first-order perturbation bound on the five signed distances the inlier test reads,
      * with the SAME accept decision (common_lib.h:528-534) wherever every exact margin to the 0.1 m threshold exceeds that bound.
    The referee has teeth: an fp32 solve of the normal equations (not backward stable at kappa ~ 1e3..1e4) fails it by orders of
    magnitude, and so does a plane with the sign convention of d reversed."""
    from oracle import referee
    cases = _golden_plane_cases()
    planes = np.zeros((len(cases), 4), np.float32)
    oks = np.zeros(len(cases), bool)
    worst_be, worst_pd, n_dec, n_und = 0.0, 0.0, 0, 0
    for i, p in enumerate(cases):
        oks[i], planes[i] = oracle.esti_plane(p, 0.1)
        r = referee.referee_case(p, planes[i], oks[i], c_bound=16.0)
        assert not r["singular"]
        worst_be = max(worst_be, r["backward_error_over_u"])
        worst_pd = max(worst_pd, r["pd_max_diff"] / r["pd_bound"])
        if r["decision_equal"] is None:
            n_und += 1
        else:
            n_dec += 1
            assert r["decision_equal"], (i, r)
    assert worst_be <= 16.0 and worst_pd <= 1.0
    assert n_dec >= 0.95 * len(cases) and n_und < 0.05 * len(cases)
    # the batched fp64 route bench.py uses on every plane of a scan gives the same verdicts on these cases
    b = referee.referee_batch_fp64(cases, planes, oks, c_bound=16.0)
    assert b["audited"] == len(cases) and abs(b["max_backward_error_over_u"] - worst_be) < 1e-3 * worst_be
    assert b["decisions_equal"] and b["decidable"] == n_dec and b["pd_diff_over_bound_max"] <= 1.0
    # teeth 1: normal equations in fp32
    bad = np.zeros_like(planes)
    for i, p in enumerate(cases[:400]):
        A = p.astype(np.float32)
        x = np.linalg.solve((A.T @ A).astype(np.float32), (-A.sum(0)).astype(np.float32)).astype(np.float32)
        nrm = np.float32(np.linalg.norm(x))
        bad[i] = np.concatenate([x / nrm, [np.float32(1) / nrm]])
    bb = referee.referee_batch_fp64(cases[:400:5], bad[:400:5], oks[:400:5], c_bound=16.0)         # the well-conditioned kind
    assert bb["max_backward_error_over_u"] > 100.0
    # teeth 2: d with the other sign convention (n^.p - d)
    flipped = planes.copy(); flipped[:, 3] *= -1
    bf = referee.referee_batch_fp64(cases[:400], flipped[:400], oks[:400], c_bound=16.0)
    assert bf["max_backward_error_over_u"] > 1e4
