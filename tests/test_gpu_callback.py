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


def _callback_case(gpu_capi, synth, small_case, n_scans=3):
    w, sc = small_case["world"], small_case["scan"]
    scans, priors = [], []
    for k in range(n_scans):
        s2 = sc if k == 0 else synth.make_scan(w, len(sc.body), seed=synth.SEED + 500 + k, pos=(0.5 * k, -0.2 * k, 0.0), rpy=(0.002, -0.003, 0.03 * k))
        ps, rt = (small_case["pos"], small_case["rot"]) if k == 0 else synth.perturb_state(s2, seed=synth.SEED + 600 + k)
        scans.append(s2)
        priors.append(np.ascontiguousarray(gpu_capi.make_state26(ps, rt, s2.off_R, s2.off_T), np.float64))
    return w, scans, priors


# (patched, page-locked node buffers, one-argument scan_changed, rows through the copy engine)
@pytest.mark.parametrize("patched,pin,legacy,rows_dma", [(0, 1, 0, 0), (1, 1, 0, 0), (0, 0, 0, 0), (1, 0, 1, 0), (0, 1, 0, 1), (0, 0, 1, 1)])
def test_callback_chain_equals_the_library_update(gpu_capi, synth, small_case, patched, pin, legacy, rows_dma):
    """kf.update_iterated_dyn_share_modified -> h_share_model(state&, dyn_share&) through the adaptor header, rows back, H^T H
    formed by the filter on the host: same iterations, same posterior (to the summation order of H^T H) as the library's
    own update; `patched` = the callback hands over the reduced form instead of rows.  Three scans in a row, so that the later
    ones run on passes lio_measure enqueued ahead of their pose (the previous scan's converge pattern) and on a cloud a kernel
    pulled out of the page-locked node buffer (`pin`) the moment lio::scan_changed(ikdtree, G) saw it; `legacy` = the
    one-argument scan_changed (the first callback uploads); `rows_dma` = rows copied straight into page-locked h_x / h."""
    w, scans, priors = _callback_case(gpu_capi, synth, small_case)
    B = gpu_capi.callback_bench()
    n = len(scans[0].body)
    ctx = gpu_capi.LioContext(0.2, max_scan_points=n, max_map_points=400_000)
    ctx.map_build(w.points)
    P0 = np.ascontiguousarray(synth.init_cov(), np.float64)
    ref = []
    for sc, x0 in zip(scans, priors):
        kf = gpu_capi.Esekf(3)
        kf.change_x(x0); kf.change_P(P0)
        ctx.scan_set(sc.body)
        log = kf.update_iterated_dyn_share_modified(ctx, 0.001, False)
        ref.append((kf.get_x().copy(), kf.get_P().copy(), log))
    assert B.cbb_setup(ctx.h, n) == 0
    B.cbb_option(0, pin); B.cbb_option(1, 0); B.cbb_option(2, legacy); B.cbb_option(3, 4096 if rows_dma else (2 << 20))
    if rows_dma:
        ctx.debug_set("rows_direct", 0)            # rows through device memory + the copy engine, as for C2's 9.6 MB per pass
    ids = []
    for sc in scans:
        body = np.ascontiguousarray(sc.body, np.float32)
        ids.append(B.cbb_add_scan(body.ctypes.data_as(C.c_void_p), len(body), 3))
    released0, pulled0, dma0 = ctx.debug_get("gate_released"), ctx.debug_get("scan_pulled"), ctx.debug_get("rows_direct_dma")
    for k, (x_lib, P_lib, log) in enumerate(ref):
        kf2 = gpu_capi.Esekf(3)
        ulog = gpu_capi.lio_update_log()
        rc = B.cbb_step(kf2.h, ids[k], priors[k].ctypes.data_as(C.c_void_p), P0.ctypes.data_as(C.c_void_p), 0.001, 0, patched, C.byref(ulog))
        assert rc == 0 and B.cbb_last_status() == 0
        assert ulog.n_calls == log["n_calls"] and ulog.n_search == log["n_search"] and B.cbb_last_m() == log["last_m"]
        x_cb, P_cb = kf2.get_x(), kf2.get_P()
        if patched:
            assert np.array_equal(x_cb, x_lib) and np.array_equal(P_cb, P_lib)      # the very same numbers went in
        else:
            assert np.abs(x_cb - x_lib).max() < 1e-11 and np.abs(P_cb - P_lib).max() < 1e-12
    if patched:
        assert ctx.debug_get("gate_released") - released0 >= 2      # scans 2 and 3: passes that were on the stream before their pose
    assert (ctx.debug_get("scan_pulled") - pulled0 == len(scans)) == bool(pin)
    if not patched:
        assert (ctx.debug_get("rows_direct_dma") - dma0 > 0) == bool(pin and rows_dma)
    ctx.debug_set("rows_direct", -1)
    ctx.measure_want_rows(False)
    B.cbb_option(0, 1); B.cbb_option(2, 0); B.cbb_option(3, 2 << 20)
    ctx.close()


def test_a_failed_pass_costs_the_pass_not_the_node(gpu_capi, synth, small_case):
    """The reference's callback has no error channel: trouble = ekfom_data.valid = false + a warning, and the update skips the
    pass (laserMapping.cpp:1956-1961, esekfom.hpp:1641-1644).  The adaptor's default does the same with every LIO_ERR_* (here a
    bounded wait of the reduction made to run out): the step returns, the state is the prior, the NEXT scan works; with
    throw_on_error the same failure surfaces as an exception (callback_bench turns it into its status)."""
    w, scans, priors = _callback_case(gpu_capi, synth, small_case, 2)
    B = gpu_capi.callback_bench()
    n = len(scans[0].body)
    ctx = gpu_capi.LioContext(0.2, max_scan_points=n, max_map_points=400_000)
    ctx.map_build(w.points)
    P0 = np.ascontiguousarray(synth.init_cov(), np.float64)
    assert B.cbb_setup(ctx.h, n) == 0
    ids = [B.cbb_add_scan(np.ascontiguousarray(sc.body, np.float32).ctypes.data_as(C.c_void_p), n, 3) for sc in scans]
    for patched in (0, 1):
        for throwing in (0, 1):
            B.cbb_option(1, throwing)
            ctx.debug_set("force_reduction_timeout", 1)
            kf = gpu_capi.Esekf(3)
            ulog = gpu_capi.lio_update_log()
            rc = B.cbb_step(kf.h, ids[0], priors[0].ctypes.data_as(C.c_void_p), P0.ctypes.data_as(C.c_void_p), 0.001, 0, patched, C.byref(ulog))
            assert B.cbb_last_status() == gpu_capi.LIO_ERR_HIP
            if throwing:
                assert rc == gpu_capi.LIO_ERR_HIP
            else:
                assert rc == 0
                assert np.array_equal(kf.get_x(), priors[0]) and np.array_equal(kf.get_P(), P0)     # every pass invalid: nothing moved
            ctx.debug_set("force_reduction_timeout", 0)
            B.cbb_option(1, 0)
            rc = B.cbb_step(kf.h, ids[1], priors[1].ctypes.data_as(C.c_void_p), P0.ctypes.data_as(C.c_void_p), 0.001, 0, patched, C.byref(ulog))
            assert rc == 0 and B.cbb_last_status() == 0 and B.cbb_last_m() > n // 2
            assert np.linalg.norm(kf.get_x()[:3] - scans[1].pos) < 0.02
    ctx.measure_want_rows(False)
    ctx.close()


@pytest.mark.parametrize("stride_floats", [3, 4, 12])
def test_scan_set_async_equals_scan_set(gpu_capi, small_case, stride_floats):
    """lio_scan_set_async: records pulled by a kernel out of page-locked memory (lio_host_register / lio_host_alloc), or packed
    and copied without a wait from pageable memory -- the passes behind it see exactly the scan lio_scan_set would have
    uploaded (strides of 12, 16 and pcl::PointXYZINormal's 48 bytes)."""
    w, sc, st = small_case["world"], small_case["scan"], small_case["state14"]
    n = len(sc.body)
    ctx = gpu_capi.LioContext(0.2, max_scan_points=n, max_map_points=400_000)
    ctx.map_build(w.points)
    ctx.scan_set(sc.body)
    g0 = ctx.measure(st, True, False)
    nx0, nd0, nc0 = ctx.neighbors()
    rec = np.full((n, stride_floats), 7.0, np.float32)
    rec[:, :3] = sc.body
    L = ctx.L
    for pinned, fused in ((False, 1), (True, 1), (True, 0)):
        ctx.debug_set("fused_pull", fused)          # 1: the first search pulls its own queries; 0: a pull kernel in front of it
        buf = rec.copy()
        if pinned:
            assert L.lio_host_register(buf.ctypes.data_as(C.c_void_p), buf.nbytes) == 0
        before = ctx.debug_get("scan_pulled")
        rc = L.lio_scan_set_async(ctx.h, buf.ctypes.data_as(C.c_void_p), n, 4 * stride_floats)
        assert rc == 0, L.lio_last_error(ctx.h)
        g = ctx.measure(st, True, False)
        assert ctx.debug_get("scan_pulled") - before == (1 if pinned else 0)
        nx, nd, nc = ctx.neighbors()
        assert g["m"] == g0["m"] and np.array_equal(g["HTH"], g0["HTH"]) and np.array_equal(g["HTh"], g0["HTh"])
        assert bits_equal(nx, nx0) and bits_equal(nd, nd0) and np.array_equal(nc, nc0)
        assert bits_equal(ctx.scan_cloud(1), np.ascontiguousarray(sc.body, np.float32))     # the scan buffer every later reader sees
        # a scan handed over and never searched is pulled by whoever asks for it first
        buf[:, :3] += 1.0
        assert L.lio_scan_set_async(ctx.h, buf.ctypes.data_as(C.c_void_p), n, 4 * stride_floats) == 0
        assert bits_equal(ctx.scan_cloud(1), np.ascontiguousarray(sc.body + 1.0, np.float32))
        buf[:, :3] -= 1.0
        if pinned:
            ctx.synchronize()
            assert L.lio_host_unregister(buf.ctypes.data_as(C.c_void_p)) == 0
    ctx.close()


@pytest.mark.parametrize("assist", [0, 60])
def test_a_stale_registration_is_caught_behind_the_pull(gpu_capi, small_case, assist):
    """a cloud freed and re-allocated at the same address keeps its OLD pages registered -- the GPU's pull then reads
    them, not the scan.  The record of the pass that pulled carries the first and the last point the GPU saw (DeviceOut::probe);
    lio_measure_end compares them with the host's records, uploads the scan again through the staging buffer on a difference and
    repeats the pass ("stale_pulls").  Emulated with the test hook "pull_alias_shift": the kernel reads three records further on."""
    w, sc, st = small_case["world"], small_case["scan"], small_case["state14"]
    n = len(sc.body) - 8
    ctx = gpu_capi.LioContext(0.2, max_scan_points=n + 8, max_map_points=400_000)
    ctx.map_build(w.points)
    ctx.scan_set(sc.body[:n])
    g0 = ctx.measure(st, True, False)
    nx0, nd0, nc0 = ctx.neighbors()
    rec = np.full((n + 8, 12), 7.0, np.float32)
    rec[:, :3] = sc.body
    L = ctx.L
    assert L.lio_host_register(rec.ctypes.data_as(C.c_void_p), rec.nbytes) == 0
    ctx.debug_set("assist_pull", assist)
    for shift, stale in ((0, 0), (3, 1), (0, 0)):
        ctx.debug_set("pull_alias_shift", shift)
        before = ctx.debug_get("stale_pulls")
        assert L.lio_scan_set_async(ctx.h, rec.ctypes.data_as(C.c_void_p), n, 48) == 0
        g = ctx.measure(st, True, False)                  # the pull, the check, (the repeat)
        assert ctx.debug_get("stale_pulls") - before == stale
        nx, nd, nc = ctx.neighbors()
        assert g["m"] == g0["m"] and np.array_equal(g["HTH"], g0["HTH"]) and np.array_equal(g["HTh"], g0["HTh"])
        assert bits_equal(nx, nx0) and bits_equal(nd, nd0) and np.array_equal(nc, nc0)
        assert bits_equal(ctx.scan_cloud(1), np.ascontiguousarray(sc.body[:n], np.float32))
        g2 = ctx.measure(st, False, False)                # and the scan goes on as usual
        assert g2["m"] == g0["m"]
    ctx.debug_set("pull_alias_shift", 0); ctx.debug_set("assist_pull", 1)
    ctx.synchronize()
    assert L.lio_host_unregister(rec.ctypes.data_as(C.c_void_p)) == 0
    ctx.close()


@pytest.mark.parametrize("stride_floats", [8, 12])
def test_assisted_pull_equals_scan_set(gpu_capi, synth, small_case, stride_floats):
    """The first search of a scan handed over as page-locked host records, with the host packing the FRONT of the cloud beside
    it (assisted pull, lio_pull.hip; alone or with the library's optional second packing thread): workgroups of the back pull raw records, workgroups of the front wait for the host's
    progress word and read 16-byte records -- or give up on a stalled host and read raw.  Every share (off, by the running
    estimate, 25 / 50 / 90 per cent) and a host that sleeps 2 ms before it packs leave exactly the scan, the neighbours and the
    sums of lio_scan_set."""
    w = small_case["world"]
    n = 60_000
    sc = synth.make_scan(w, n, seed=synth.SEED + 4242)
    ps, rt = synth.perturb_state(sc, seed=synth.SEED + 4243)
    st = gpu_capi.make_state26(ps, rt, sc.off_R, sc.off_T)[:14]
    ctx = gpu_capi.LioContext(0.2, max_scan_points=n, max_map_points=400_000)
    if not ctx.debug_get("gate_mail_in_vram"):
        ctx.close(); pytest.skip("no large BAR: the progress word would be polled over PCIe (assisted pull stays off)")
    ctx.map_build(w.points)
    ctx.scan_set(sc.body)
    g0 = ctx.measure(st, True, False)
    nx0, nd0, nc0 = ctx.neighbors()
    rec = np.full((n, stride_floats), 7.0, np.float32)
    rec[:, :3] = sc.body
    L = ctx.L
    buf = rec.copy()
    assert L.lio_host_register(buf.ctypes.data_as(C.c_void_p), buf.nbytes) == 0
    for lanes, threads in ((0, 1), (16, 1), (0, 2), (0, 1)):
        ctx.debug_set("lanes_per_query", lanes)
        ctx.scan_pack_threads(threads)                  # 2: a second packing thread inside the library takes the second half of the front
        for share, stall in ((0, 0), (1, 0), (25, 0), (50, 0), (90, 0), (50, 2000), (1, 0)):
            ctx.debug_set("assist_pull", share); ctx.debug_set("assist_pull_stall_us", stall)
            before = ctx.debug_get("assisted_pulls")
            assert L.lio_scan_set_async(ctx.h, buf.ctypes.data_as(C.c_void_p), n, 4 * stride_floats) == 0, L.lio_last_error(ctx.h)
            g = ctx.measure(st, True, False)
            assert ctx.debug_get("assisted_pulls") - before == (0 if share == 0 else 1)      # (also right behind the stalled scan: one slow scan does not end the help)
            if share >= 2:
                assert ctx.debug_get("last_pack_front") == (share * n // 100) // 64 * 64
            nx, nd, nc = ctx.neighbors()
            assert g["m"] == g0["m"] and np.array_equal(g["HTH"], g0["HTH"]) and np.array_equal(g["HTh"], g0["HTh"])
            assert bits_equal(nx, nx0) and bits_equal(nd, nd0) and np.array_equal(nc, nc0)
            assert bits_equal(ctx.scan_cloud(1), np.ascontiguousarray(sc.body, np.float32))
            g = ctx.measure(st, False, False)                       # (a pass on the scan buffer the search left)
            assert g["m"] > n // 2
    ctx.debug_set("assist_pull_stall_us", 0)
    ctx.synchronize()
    assert L.lio_host_unregister(buf.ctypes.data_as(C.c_void_p)) == 0
    ctx.close()


def test_assisted_pull_survives_the_nodes_call_order(gpu_capi, synth):
    """The node calls ikdtree.validnum() and ikdtree.size() between lio::scan_changed and the update (laserMapping.cpp:2343-2344).
    Both read host counters: the scan pull still owed by scan_changed must stay owed -- the scan's first search pulls it, with the
    calling thread packing the front (assisted pull) -- and passes armed across calls must stay armed.  (Round 4's lio_map_size /
    lio_map_validnum went through enter_map() and flushed the pull with a stand-alone kernel: the assisted pull never ran in the
    node's own order.)  callback_bench's cbb_step issues the two calls where the node does."""
    import ctypes as C
    nq = 60_000                                   # 48-byte records: 2.9 MB, 50 us of PCIe time -- a cloud worth the help
    w = synth.make_world(400_000, 0.2)
    B = gpu_capi.callback_bench()
    ctx = gpu_capi.LioContext(0.2, max_scan_points=nq, max_map_points=600_000)
    if not ctx.debug_get("gate_mail_in_vram"):
        ctx.close(); pytest.skip("no large BAR: the assisted pull stays off")
    ctx.map_build(w.points)
    P0 = np.ascontiguousarray(synth.init_cov(), np.float64)
    scans, priors = [], []
    for k in range(3):
        s2 = synth.make_scan(w, nq, seed=synth.SEED + 900 + k, pos=(0.6 * k, 0.2 * k, 0.0), rpy=(0.002, -0.003, 0.04 * k))
        ps, rt = synth.perturb_state(s2, seed=synth.SEED + 950 + k)
        scans.append(s2); priors.append(np.ascontiguousarray(gpu_capi.make_state26(ps, rt, s2.off_R, s2.off_T), np.float64))
    ref = []
    for sc, x0 in zip(scans, priors):
        kf = gpu_capi.Esekf(3)
        kf.change_x(x0); kf.change_P(P0)
        ctx.scan_set(sc.body)
        log = kf.update_iterated_dyn_share_modified(ctx, 0.001, False)
        ref.append((kf.get_x().copy(), kf.get_P().copy(), log))
    assert B.cbb_setup(ctx.h, nq) == 0
    B.cbb_option(0, 1); B.cbb_option(1, 0); B.cbb_option(2, 0)
    ids = [B.cbb_add_scan(np.ascontiguousarray(sc.body, np.float32).ctypes.data_as(C.c_void_p), nq, 3) for sc in scans]
    assisted0, pulled0 = ctx.debug_get("assisted_pulls"), ctx.debug_get("scan_pulled")
    for rep in range(2):
        for k, (x_lib, P_lib, log) in enumerate(ref):
            kf2 = gpu_capi.Esekf(3)
            ulog = gpu_capi.lio_update_log()
            rc = B.cbb_step(kf2.h, ids[k], priors[k].ctypes.data_as(C.c_void_p), P0.ctypes.data_as(C.c_void_p), 0.001, 0, 1, C.byref(ulog))
            assert rc == 0 and B.cbb_last_status() == 0 and B.cbb_last_validnum() == len(w.points)
            assert ulog.n_calls == log["n_calls"] and B.cbb_last_m() == log["last_m"]
            assert np.array_equal(kf2.get_x(), x_lib) and np.array_equal(kf2.get_P(), P_lib)
    assert ctx.debug_get("scan_pulled") - pulled0 == 6
    assert ctx.debug_get("assisted_pulls") - assisted0 == 6           # every scan's first search pulled its own cloud, host packing beside it
    ctx.measure_want_rows(False)
    ctx.close()


def test_assisted_pull_fuzz_seed():
    """150 scans of tests/fuzz/fuzz_assist.py: clouds of random sizes and strides, share steered / fixed / off, one or two packing
    threads switched at random, a host that stalls -- every scan's first pass equal to lio_scan_set's on a twin context, to the bit."""
    import importlib.util
    import os
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "fuzz", "fuzz_assist.py")
    spec = importlib.util.spec_from_file_location("fuzz_assist", path)
    fz = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(fz)
    fz.main(150, 71500)


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
