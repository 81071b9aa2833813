"""The zero-patch drop-in path: the rows h_x / h a pass stages on the device for the unmodified
update_iterated_dyn_share_modified (lio_measure_want_rows + lio_measure_rows), and the whole callback chain
(examples/callback_bench.cpp: include/lio_hip_adaptor.hpp behind the reference's callback signature, a filter that forms
h_x^T h_x from the rows, esekfom.hpp:1790,1810) against the library's own overlapped update."""
import ctypes as C

import numpy as np
import pytest

from conftest import bits_equal

pytestmark = pytest.mark.gpu


def test_staged_rows_equal_the_reevaluated_ones(gpu_capi, small_case, hap_case):
    """Rows left by the pass itself (compacted on the device, column-major; written straight into pinned host memory or
    moved by the copy engine) == rows of the re-evaluation pass (round 2's lio_measure_rows), bit for bit -- search passes
    (with queries the tree walk settles) and passes that re-use the neighbours, both extrinsic modes."""
    for case, ext in ((small_case, False), (hap_case, True)):
        w, sc, st = case["world"], case["scan"], case["state14"]
        ctx = gpu_capi.LioContext(0.2, max_scan_points=len(sc.body), max_map_points=400_000)
        ctx.map_build(w.points)
        ctx.scan_set(sc.body)
        st2 = st.copy(); st2[:3] += [0.01, -0.02, 0.005]
        ref = []
        for state, search in ((st, True), (st2, False), (st2, True)):
            g = ctx.measure(state, search, ext)
            ref.append((g, ) + ctx.measure_rows(g["m"]))
        ctx.measure_want_rows(True)
        for direct in (0, 1):
            ctx.debug_set("rows_direct", direct)
            ctx.scan_set(sc.body)
            for (state, search), (g0, hx0, h0) in zip(((st, True), (st2, False), (st2, True)), ref):
                g = ctx.measure(state, search, ext)
                assert g["m"] == g0["m"] and np.array_equal(g["HTH"], g0["HTH"])
                hx, h = ctx.measure_rows(g["m"])
                assert bits_equal(hx, hx0) and bits_equal(h, h0)
                hxc, hc = ctx.measure_rows(g["m"], col_major=True)
                assert bits_equal(np.ascontiguousarray(hxc), hx0) and bits_equal(hc, h0)
        ctx.close()


@pytest.mark.parametrize("patched", [0, 1])
def test_callback_chain_equals_the_library_update(gpu_capi, synth, small_case, patched):
    """kf.update_iterated_dyn_share_modified -> h_share_model(state&, dyn_share&) through the adaptor header, rows back, H^T H
    formed by the filter on the host: same iterations, same posterior (to the summation order of H^T H) as the library's
    own update; `patched` = the callback hands over the reduced form instead of rows."""
    w, sc = small_case["world"], small_case["scan"]
    B = gpu_capi.callback_bench()
    ctx = gpu_capi.LioContext(0.2, max_scan_points=len(sc.body), max_map_points=400_000)
    ctx.map_build(w.points)
    x0 = gpu_capi.make_state26(small_case["pos"], small_case["rot"], sc.off_R, sc.off_T)
    P0 = np.ascontiguousarray(synth.init_cov(), np.float64)
    kf = gpu_capi.Esekf(3)
    kf.change_x(x0); kf.change_P(P0)
    ctx.scan_set(sc.body)
    log = kf.update_iterated_dyn_share_modified(ctx, 0.001, False)
    x_lib, P_lib = kf.get_x().copy(), kf.get_P().copy()
    assert B.cbb_setup(ctx.h, len(sc.body)) == 0
    body = np.ascontiguousarray(sc.body, np.float32)
    k = B.cbb_add_scan(body.ctypes.data_as(C.c_void_p), len(body), 3)
    kf2 = gpu_capi.Esekf(3)
    ulog = gpu_capi.lio_update_log()
    x0c = np.ascontiguousarray(x0, np.float64)
    rc = B.cbb_step(kf2.h, k, x0c.ctypes.data_as(C.c_void_p), P0.ctypes.data_as(C.c_void_p), 0.001, 0, patched, C.byref(ulog))
    assert rc == 0
    assert ulog.n_calls == log["n_calls"] and ulog.n_search == log["n_search"] and B.cbb_last_m() == log["last_m"]
    x_cb, P_cb = kf2.get_x(), kf2.get_P()
    if patched:
        assert np.array_equal(x_cb, x_lib) and np.array_equal(P_cb, P_lib)      # the very same numbers went in
    else:
        assert np.abs(x_cb - x_lib).max() < 1e-11 and np.abs(P_cb - P_lib).max() < 1e-12
    ctx.measure_want_rows(False)
    ctx.close()


def test_kitti_bin_step_script():
    """scripts/kitti_bin_step.py on a sweep in KITTI's .bin format (a synthetic one: no data set ships here): read, no-pose
    lio_scan_undistort, lio_scan_voxel_filter(0.5), the scan step; neighbours against the reference tree."""
    import json, os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "scripts", "kitti_bin_step.py"), "--synthetic", "--check"],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    out = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert out["scan_points"] == 120000 and 5000 < out["feats_down_size"] < 60000
    assert out["effct_feat_num"] > out["feats_down_size"] // 2 and out["pos_err_to_truth_m"] < 0.02
    if "neighbors_vs_reference" in out:
        assert out["neighbors_vs_reference"]["sqdist_mismatches"] == 0 and out["neighbors_vs_reference"]["count_mismatches"] == 0
