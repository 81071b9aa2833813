"""First-contact diagnostics on a real MI355X: parity of every stage against the oracle at a small
size, then kernel timings at the benchmark size.  Prints; exits non-zero on a parity failure.

    python tests/fuzz/gpu_probe.py [--nm 200000 --nq 20000] [--big]
"""
from __future__ import annotations

import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))

from better_fastlio2_amd import capi, synth  # noqa: E402
from oracle import pyoracle as po  # noqa: E402


def bits_equal(a, b):
    return np.array_equal(np.ascontiguousarray(a).view(np.uint32), np.ascontiguousarray(b).view(np.uint32))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--nm", type=int, default=200000)
    ap.add_argument("--nq", type=int, default=20000)
    ap.add_argument("--big", action="store_true")
    ap.add_argument("--lanes", type=int, default=0)
    a = ap.parse_args()
    fails = 0

    world = synth.make_world(a.nm, 0.2)
    scan = synth.make_scan(world, a.nq)
    pos, rot = synth.perturb_state(scan)
    st = synth.state_vector(pos, rot, scan.off_R, scan.off_T)
    ctx = capi.LioContext(0.2, max_scan_points=max(a.nq, 1000), max_map_points=a.nm + 100000, lanes_per_query=a.lanes)
    t0 = time.time(); ctx.map_build(world.points); print("map_build s", time.time() - t0, ctx.map_info())

    om = po.OracleMap(0.2); om.build(world.points)
    # --- raw nearest search, including hard queries: far away, map edge, sparse
    rng = np.random.default_rng(3)
    wq = po.body_to_world(st, scan.body)
    hard = np.concatenate([
        wq[:3000],
        rng.uniform(-1.2 * world.ground_half, 1.2 * world.ground_half, (1500, 3)).astype(np.float32) * np.array([1, 1, 0.05], np.float32),
        rng.uniform(-300, 300, (300, 3)).astype(np.float32),
        np.array([[1e4, -2e4, 50.0], [0, 0, 500.0]], np.float32)])
    gx, gd, gc = ctx.nearest_search(hard)
    ox, od, oc, _ = om.knn(hard, 5, 1)
    ok = bits_equal(gx, ox) and bits_equal(gd, od) and np.array_equal(gc, oc)
    print("nearest_search parity:", ok)
    if not ok:
        fails += 1
        bad = np.where((gd.view(np.uint32) != od.view(np.uint32)).any(1) | (gc != oc))[0]
        print("  bad queries", len(bad), bad[:10])
        for i in bad[:3]:
            print("  q", hard[i], "\n  gpu", gd[i], gc[i], "\n  orc", od[i], oc[i])
    prof_ctx = ctx
    for ext in (False, True):
        arr = po.ScanArrays(a.nq)
        ctx.scan_set(scan.body)
        for pi, (search, pose_st) in enumerate([(True, st), (False, st + np.concatenate([[0.01, -0.02, 0.005], np.zeros(11)]))]):
            valid, hx, h, m, rm = po.h_share_model(om, pose_st, scan.body, arr, search, ext)
            g = ctx.measure(pose_st, search, ext)
            HTH, HTh = po.hth(hx, h)
            w, sel, nv = ctx.scan_state()
            ok_sel = np.array_equal(sel, arr.selected)
            ok_w = bits_equal(w, arr.world)
            selm = arr.selected.astype(bool)
            ok_nv = bits_equal(nv[selm], arr.normvec[selm])
            rel = np.abs(g["HTH"] - HTH).max() / max(np.abs(HTH).max(), 1e-30)
            relh = np.abs(g["HTh"] - HTh).max() / max(np.abs(HTh).max(), 1e-30)
            print(f"measure ext={ext} pass={pi} search={search}: m {g['m']} vs {m}, sel {ok_sel}, world {ok_w}, normvec {ok_nv}, "
                  f"HTH rel {rel:.2e}, HTh rel {relh:.2e}, res_mean {g['res_mean']:.9g} vs {rm:.9g}, unresolved {g['n_unresolved']}, C {g['n_candidates'] / a.nq:.1f}")
            if not (ok_sel and ok_w and ok_nv and g["m"] == m and rel < 1e-9 and relh < 1e-9):
                fails += 1
            if search:
                nx, nd, nc = ctx.neighbors()
                okn = bits_equal(nx, arr.nbr_xyz) and bits_equal(nd, arr.nbr_d) and np.array_equal(nc, arr.nbr_cnt)
                print("   neighbours parity:", okn)
                fails += 0 if okn else 1
            ghx, gh = ctx.measure_rows(g["m"])
            okr = np.array_equal(ghx, hx) and np.array_equal(gh, h)
            print("   rows bit-equal:", okr, "" if okr else f"max diff {np.abs(ghx - hx).max():.3e}")
            fails += 0 if okr else 1
    # --- full iterated update
    for ext in (False, True):
        x0 = capi.make_state26(pos, rot, scan.off_R, scan.off_T)
        P0 = synth.init_cov()
        xo, Po, arrs, log = po.update_iterated(om, x0, P0, scan.body, 0.001, 3, None, ext)
        kf = capi.Esekf(3)
        kf.change_x(x0); kf.change_P(P0)
        ctx.scan_set(scan.body)
        lg = kf.update_iterated_dyn_share_modified(ctx, 0.001, ext)
        xg, Pg = kf.get_x(), kf.get_P()
        dpos = np.abs(xg[:3] - xo[:3]).max()
        dq = np.abs(xg[3:7] - xo[3:7]).max()
        dP = np.abs(Pg - Po).max() / np.abs(Po).max()
        err_true = np.linalg.norm(xg[:3] - scan.pos)
        print(f"update ext={ext}: calls {lg['n_calls']} (oracle {len(log)}), |dpos| {dpos:.3e}, |dquat| {dq:.3e}, relP {dP:.3e}, "
              f"pos err to truth {err_true:.4f} (start {np.linalg.norm(pos - scan.pos):.4f}), solve {lg['solve_seconds'] * 1e3:.3f} ms, measure {lg['measure_seconds'] * 1e3:.3f} ms")
        if not (dpos < 1e-4 and dq < 1e-5 and lg["n_calls"] == len(log)):
            fails += 1

    if a.big:
        del ctx
        nm, nq = 5_000_000, 100_000
        t0 = time.time(); world = synth.make_world(nm, 0.2); print("world gen s", time.time() - t0)
        scan = synth.make_scan(world, nq)
        pos, rot = synth.perturb_state(scan)
        st = synth.state_vector(pos, rot, scan.off_R, scan.off_T)
        for lanes in (8,):
            ctx = capi.LioContext(0.2, max_scan_points=nq, max_map_points=nm + 200000, lanes_per_query=lanes)
            t0 = time.time(); ctx.map_build(world.points); tb = time.time() - t0
            ctx.scan_set(scan.body)
            ctx.profile_enable(True)
            for _ in range(3):
                g = ctx.measure(st, True, False)
            ctx.profile_reset()
            t0 = time.time()
            for _ in range(20):
                g = ctx.measure(st, True, False)
            tw = (time.time() - t0) / 20
            pr = ctx.profile()
            print(f"lanes {lanes}: build {tb:.2f}s, measure(search) wall {tw * 1e6:.1f} us, m {g['m']}, unresolved {g['n_unresolved']}, Cmean {g['n_candidates'] / nq:.2f}, "
                  + ", ".join(f"{k} {v['total_ms'] / max(v['launches'], 1) * 1e3:.1f}us" for k, v in pr.items()))
            ctx.profile_reset()
            t0 = time.time()
            for _ in range(20):
                g = ctx.measure(st, False, False)
            tw = (time.time() - t0) / 20
            pr = ctx.profile()
            print(f"           measure(no search) wall {tw * 1e6:.1f} us, " + ", ".join(f"{k} {v['total_ms'] / max(v['launches'], 1) * 1e3:.1f}us" for k, v in pr.items()))
            kf = capi.Esekf(3)
            x0 = capi.make_state26(pos, rot, scan.off_R, scan.off_T)
            ctx.profile_enable(False)
            d_body = capi.scan_to_device(scan.body)
            for mode in ("host scan_set", "device-resident scan"):
                ts, sol, mea = [], [], []
                for _ in range(30):
                    kf.change_x(x0); kf.change_P(synth.init_cov())
                    t0 = time.perf_counter()
                    if mode.startswith("host"):
                        ctx.scan_set(scan.body)
                    else:
                        ctx.scan_set_device(d_body.ptr, nq)
                    lg = kf.update_iterated_dyn_share_modified(ctx, 0.001, False)
                    ts.append(time.perf_counter() - t0); sol.append(lg["solve_seconds"]); mea.append(lg["measure_seconds"])
                print(f"           scan step [{mode}] {np.median(ts) * 1e6:.1f} us ({lg['n_calls']} passes, {lg['n_search']} searches) -> {1 / np.median(ts):.1f} scans/s; "
                      f"host solve {np.median(sol) * 1e6:.1f} us, inside lio_measure {np.median(mea) * 1e6:.1f} us; final pos err {np.linalg.norm(kf.get_x()[:3] - scan.pos):.4f}")
            del ctx
    print("FAILS", fails)
    return 1 if fails else 0


if __name__ == "__main__":
    sys.exit(main())
