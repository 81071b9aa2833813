"""Differential fuzzing of the PASS CONTROL of round 4 -- passes armed ahead of their pose (gate kernel / mail read inside the pass's
first kernel), the first search of the next scan armed across the scan boundary, the pipelined upload -- against the plain sequence
(scan handed over, one pass after the other, nothing armed) on a twin context: scans of random sizes (consecutive scans of different
sizes), update lengths of 1..4 iterations (every prediction of the previous update is wrong somewhere), extrinsic estimation on/off,
random calls between two scans (map_incremental, a box search, an extra pass, rows, a wait, nothing), random switches of the armed
forms, forced mispredictions.  Compared after every scan: posterior, covariance, update log (bits); after the calls their results;
at the end the live points of both maps.

    python tests/fuzz/fuzz_pipeline.py [n_cases] [first_seed]
"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from better_fastlio2_amd import capi, synth

SIZES = (700, 1500, 3000, 5000, 8000, 12000, 20000)


def run_case(seed):
    rng = np.random.default_rng(seed)
    w = synth.make_world(int(rng.choice([120_000, 200_000])), 0.2, seed=seed)
    n_scans = int(rng.integers(8, 15))
    kind = str(rng.choice(["kitti64", "hap"]))
    P0 = synth.init_cov()
    scans, priors, iters, ops, sw = [], [], [], [], []
    for k in range(n_scans):
        nq = int(rng.choice(SIZES))
        sc = synth.make_scan(w, nq, kind=kind, seed=seed * 131 + k, pos=(0.6 * k - 2.0, 0.3 * k, 0.0), rpy=(0.002, -0.003, 0.04 * k + 0.1))
        ps, rt = synth.perturb_state(sc, dpos=float(rng.choice([0.02, 0.1, 0.25])), drot_deg=float(rng.choice([0.1, 0.5, 1.0])), seed=seed * 17 + k)
        scans.append(sc); priors.append(capi.make_state26(ps, rt, sc.off_R, sc.off_T))
        iters.append(int(rng.integers(1, 5)))
        ops.append(str(rng.choice(["none"] * 8 + ["map_incremental", "map_incremental", "nearest_search", "extra_pass", "rows", "synchronize", "fov"])))
        # switches of the armed forms (no effect on any result): gate_in_kernel 0 / 1 / 3 / 7, pre_enqueue, arm_next_scan, gate_mispredict
        # (every lio_debug_set calls off what is armed: one switch on a quarter of the scans, so that most armed passes get released)
        one = [-1, -1, -1, -1]
        if rng.random() < 0.25:
            j = int(rng.integers(0, 4))
            one[j] = int(rng.choice([0, 1, 3, 7])) if j == 0 else int(rng.integers(0, 2))
        sw.append(tuple(one))
    ext = bool(scans[0].extrinsic_est_en)
    cap = max(len(s.body) for s in scans)
    qs = (w.points[:: max(1, len(w.points) // 400)][:400] + np.float32(0.013)).astype(np.float32)

    def run(piped):
        ctx = capi.LioContext(0.2, max_scan_points=cap, max_map_points=600_000)
        ctx.map_build(w.points)
        if not piped:
            ctx.debug_set("pre_enqueue", 0); ctx.debug_set("arm_next_scan", 0)
        pinned = [capi.pinned_points(s.body) for s in scans]
        out = []
        if piped:
            ctx.scan_upload_async(pinned[0])
        for k in range(n_scans):
            kf = capi.Esekf(iters[k])
            if piped:
                g, pe, an, mp = sw[k]
                if g >= 0: ctx.debug_set("gate_in_kernel", g)
                if pe >= 0: ctx.debug_set("pre_enqueue", pe)
                if an >= 0: ctx.debug_set("arm_next_scan", an)
                if mp >= 0: ctx.debug_set("gate_mispredict", mp)
                log = capi.scan_step_pipelined(kf, ctx, priors[k], P0, pinned[k + 1] if k + 1 < n_scans else None, 0.001, ext)
            else:
                buf = capi.scan_to_device(scans[k].body)
                log = capi.scan_step_device(kf, ctx, priors[k], P0, buf.ptr, len(scans[k].body), 0.001, ext)
            x = kf.get_x()
            out.append(("x", x.copy())); out.append(("P", kf.get_P().copy()))
            out.append(("log", np.array([log["n_calls"], log["n_search"], log["last_m"], log["converged"]], np.int64)))
            op = ops[k]
            if op == "map_incremental":
                mi = ctx.map_incremental(x[:14], 0.2, True)
                out.append(("mi", np.array([mi["n_add"], mi["n_nodown"], ctx.map_validnum()], np.int64)))
            elif op == "nearest_search":
                gx, gd, gc = ctx.nearest_search(qs)
                out.append(("ns_x", gx)); out.append(("ns_d", gd)); out.append(("ns_c", gc))
            elif op == "extra_pass":
                g2 = ctx.measure(x[:14], bool(k & 1), ext)
                out.append(("pass", np.concatenate([g2["HTH"].ravel(), g2["HTh"], [g2["m"], g2["res_mean"]]])))
            elif op == "rows":
                g2 = ctx.measure(x[:14], False, ext)
                if g2["m"] > 0:
                    hx, h = ctx.measure_rows(g2["m"])
                    out.append(("rows", np.concatenate([hx.ravel(), h])))
            elif op == "synchronize":
                ctx.synchronize()
            elif op == "fov":
                r = ctx.fov_segment(x[:3], 200.0, 60.0)
                out.append(("fov", np.array([ctx.map_validnum()], np.int64)))
        flat = ctx.map_flatten()
        out.append(("flat", flat[np.lexsort(flat.T[::-1])]))
        stats = (ctx.debug_get("next_scan_armed"), ctx.debug_get("next_scan_released"), ctx.debug_get("gate_released"), ctx.debug_get("gate_called_off"))
        ctx.close()
        return out, stats

    plain, _ = run(False)
    piped, stats = run(True)
    assert len(plain) == len(piped), (len(plain), len(piped))
    for (na, a), (nb, b) in zip(plain, piped):
        assert na == nb, (na, nb)
        a = np.ascontiguousarray(a); b = np.ascontiguousarray(b)
        if a.shape != b.shape or a.dtype != b.dtype or not np.array_equal(a.view(np.uint8), b.view(np.uint8)):
            raise AssertionError(f"seed {seed}: '{na}' differs between the plain and the pipelined run (scans {[len(s.body) for s in scans]}, iters {iters}, ops {ops}, switches {sw})")
    return stats


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    first = int(sys.argv[2]) if len(sys.argv) > 2 else 61000
    tot = np.zeros(4, np.int64)
    for s in range(first, first + n):
        tot += np.array(run_case(s), np.int64)
    print("FUZZ PIPELINE OK: %d cases; next-scan searches armed %d (released %d), passes released by the host's word %d, called off %d" % (n, *tot))
