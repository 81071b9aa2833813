"""Parity of the HIP path (through the C ABI) with the oracle on identical seeded inputs, the edge
cases of the domain, and size-independent properties at the benchmark's full size."""
import numpy as np
import pytest

from conftest import bits_equal

pytestmark = pytest.mark.gpu


def _hard_queries(case, oracle, n_scan=4000, seed=5):
    rng = np.random.default_rng(seed)
    wq = oracle.body_to_world(case["state14"], case["scan"].body)[:n_scan]
    half = case["world"].ground_half
    return np.ascontiguousarray(np.concatenate([
        wq,
        (rng.uniform(-1.3 * half, 1.3 * half, (2000, 3)) * [1, 1, 0.03]).astype(np.float32),   # map edge and beyond
        (rng.uniform(-half, half, (500, 3)) * [1, 1, 0.1] + [0, 0, 6]).astype(np.float32),      # above everything
        rng.uniform(-300, 300, (300, 3)).astype(np.float32),                                    # far away
        # exact hits and near-hits: squared distance 0 / subnormal -- the k-NN kernel carries (distance bits,
        # index) keys as doubles, and these keys are DENORMAL doubles (a flush would lose the index)
        case["world"].points[::97][:1500].copy(),
        (case["world"].points[5::131][:500] + np.float32(3e-23)).astype(np.float32),
        np.array([[1e4, -2e4, 50.0], [0, 0, 900.0], [-3e5, 1e5, 0]], np.float32)]))


@pytest.fixture(scope="module")
def ctx_small(gpu_capi, small_case):
    c = gpu_capi.LioContext(0.2, max_scan_points=32768, max_map_points=400_000)
    c.map_build(small_case["world"].points)
    yield c
    c.close()


@pytest.mark.parametrize("lanes", [4, 8, 16, 32, 64])
def test_nearest_search_bit_exact(gpu_capi, oracle, small_case, lanes):
    """Neighbour points, squared distances and counts identical to the (reference-pinned) oracle,
    for in-map queries (27-cell kernel) and edge/far queries (exact tree walk)."""
    ctx = gpu_capi.LioContext(0.2, max_scan_points=1024, max_map_points=400_000, lanes_per_query=lanes)
    ctx.map_build(small_case["world"].points)
    q = _hard_queries(small_case, oracle)
    om = oracle.OracleMap(0.2); om.build(small_case["world"].points)
    ox, od, oc, _ = om.knn(q)
    gx, gd, gc = ctx.nearest_search(q)
    assert np.array_equal(gc, oc)
    assert bits_equal(gd, od)                                   # every squared distance, every query
    # Which of several EXACTLY equidistant points the reference returns depends on its traversal
    # order (strict `<` at ikd_Tree.cpp:887), so points are compared where ranks 1..6 are tie-free
    # (all in-map queries; far queries collide in fp32: ulp(d^2) ~ 8 at 10 km) ...
    x6, d6, c6, _ = om.knn(q, 6, 1)
    tie_free = (np.diff(d6, axis=1) > 0).all(1)
    assert tie_free[:4000].all() and tie_free.mean() > 0.95 and (od[:, 0] == 0).sum() >= 1500
    assert bits_equal(gx[tie_free], ox[tie_free])
    # ... and on tied rows every returned point must be a map point at exactly the reported distance
    mp = {p.tobytes() for p in small_case["world"].points}
    for i in np.where(~tie_free)[0]:
        for k in range(5):
            assert gx[i, k].tobytes() in mp
            dd = q[i] - gx[i, k]
            assert np.float32(np.float32(dd[0] * dd[0] + dd[1] * dd[1]) + dd[2] * dd[2]) == gd[i, k]
    # stride-48 input (pcl::PointXYZINormal records, common_lib.h:161) gives the same answer
    rec = np.zeros((len(q), 12), np.float32); rec[:, :3] = q; rec[:, 3] = 1.0; rec[:, 8] = 7.0
    gx2, gd2, gc2 = ctx.nearest_search(rec)
    assert bits_equal(gx2, gx) and bits_equal(gd2, gd)
    ctx.close()


@pytest.mark.parametrize("lanes", [4, 8, 16])
def test_candidate_walk_variants_are_bit_identical(gpu_capi, oracle, small_case, lanes):
    """knn27_kernel's candidate walk in prefix form with one / two / four candidates per lane in flight (KnnArgs.walk, lio_knn.hip)
    sees the same candidates as the plain walk: neighbours, distances, counts, the unproven list (hence the walk's results)
    and the sums of a search pass and of the bounded re-search at a moved pose are the same bits -- and the oracle's."""
    q = _hard_queries(small_case, oracle)
    om = oracle.OracleMap(0.2); om.build(small_case["world"].points)
    ox, od, oc, _ = om.knn(q)
    ident = np.array([0, 0, 0, 0, 0, 0, 1, 0, 0, 0, 1, 0, 0, 0], np.float64)
    moved = ident.copy(); moved[:3] = [0.04, -0.03, 0.02]
    ctx = gpu_capi.LioContext(0.2, max_scan_points=len(q), max_map_points=400_000, lanes_per_query=lanes)
    ctx.map_build(small_case["world"].points)
    res = {}
    for walk in (0, 1, 2, 4):
        ctx.debug_set("knn_walk", walk)
        gx, gd, gc = ctx.nearest_search(q)
        ctx.scan_set(q)
        a = ctx.measure(ident, True, False)
        n1 = ctx.neighbors()
        b = ctx.measure(moved, True, False)          # bounded by the previous neighbours (HINT variant of the kernel)
        n2 = ctx.neighbors()
        res[walk] = (gx, gd, gc, a, n1, b, n2)
    ctx.close()
    assert np.array_equal(res[0][2], oc) and bits_equal(res[0][1], od)
    for walk in (1, 2, 4):
        r, z = res[walk], res[0]
        assert bits_equal(r[0], z[0]) and bits_equal(r[1], z[1]) and np.array_equal(r[2], z[2])
        for k in (3, 5):
            assert r[k]["m"] == z[k]["m"] and r[k]["n_unresolved"] == z[k]["n_unresolved"]
            assert np.array_equal(r[k]["HTH"], z[k]["HTH"]) and np.array_equal(r[k]["HTh"], z[k]["HTh"])
        for k in (4, 6):
            assert bits_equal(r[k][0], z[k][0]) and bits_equal(r[k][1], z[k][1]) and np.array_equal(r[k][2], z[k][2])


@pytest.mark.parametrize("lanes", [4, 8, 16, 32, 64])
def test_second_chance_inside_the_knn_kernel_is_bit_identical(gpu_capi, oracle, small_case, lanes):
    """Round 4 (opt-in, knn_tune bit 1; measured and not adopted): a group of knn27_kernel whose 3x3x3 result stays unproven is finished by its wavefront inside the 5x5x5 block
    (shell_second_chance, lio_knn.hip) instead of being listed for the tree walk.  Same cells, same candidates, same keys, same
    proof as phase A of the walk: neighbours, distances and counts of the stand-alone search, of a search pass and of the bounded
    re-search at a moved pose are the bits the kernel without the second chance + the walk produce -- and the
    oracle's; the list shrinks to what lies beyond 5x5x5 cells; the sums agree to rounding (a settled query is summed by its
    owner workgroup instead of by its walker)."""
    q = _hard_queries(small_case, oracle)
    om = oracle.OracleMap(0.2); om.build(small_case["world"].points)
    ox, od, oc, _ = om.knn(q)
    ident = np.array([0, 0, 0, 0, 0, 0, 1, 0, 0, 0, 1, 0, 0, 0], np.float64)
    moved = ident.copy(); moved[:3] = [0.04, -0.03, 0.02]
    ctx = gpu_capi.LioContext(0.2, max_scan_points=len(q), max_map_points=400_000, lanes_per_query=lanes)
    ctx.map_build(small_case["world"].points)
    res = {}
    for tune in (0, 2):
        ctx.debug_set("knn_tune", tune)
        gx, gd, gc = ctx.nearest_search(q)
        ctx.scan_set(q)
        a = ctx.measure(ident, True, False)
        n1 = ctx.neighbors()
        b = ctx.measure(moved, True, False)
        n2 = ctx.neighbors()
        w, sel, nv = ctx.scan_state()
        res[tune] = (gx, gd, gc, a, n1, b, n2, sel, nv)
    ctx.close()
    off, on = res[0], res[2]
    assert np.array_equal(on[2], oc) and bits_equal(on[1], od)
    assert bits_equal(on[0], off[0]) and bits_equal(on[1], off[1]) and np.array_equal(on[2], off[2])
    for k in (4, 6):
        assert bits_equal(on[k][0], off[k][0]) and bits_equal(on[k][1], off[k][1]) and np.array_equal(on[k][2], off[k][2])
    assert np.array_equal(on[7], off[7]) and bits_equal(on[8], off[8])
    for k in (3, 5):
        assert on[k]["m"] == off[k]["m"] and 0 < on[k]["n_unresolved"] < off[k]["n_unresolved"]
        assert np.abs(on[k]["HTH"] - off[k]["HTH"]).max() <= 1e-12 * np.abs(off[k]["HTH"]).max()
        assert np.abs(on[k]["HTh"] - off[k]["HTh"]).max() <= 1e-12 * np.abs(off[k]["HTh"]).max()


@pytest.mark.parametrize("ext,second_chance", [(False, 0), (True, 0), (False, 1), (True, 1)])
def test_tree_walk_inside_the_measurement_kernel(gpu_capi, oracle, small_case, ext, second_chance):
    """A search pass walks the queries the 27-cell kernel could not prove INSIDE the measurement kernel (walker
    workgroups in front of the model workgroups; the owners of such queries wait for them).  With a scan that is
    mostly edge / far / above-the-map points (thousands of walks, clustered and scattered) the pass must equal the
    same pass with the stand-alone walk kernel bit for bit, and the neighbours must equal the oracle's.  second_chance = 0:
    knn27_kernel lists every unproven query (the default); 1 (knn_tune bit 1, opt-in): it settles what the 5x5x5 block settles
    itself and lists only the rest -- far fewer walks, the same bits."""
    q = _hard_queries(small_case, oracle)
    ident = np.array([0, 0, 0, 0, 0, 0, 1, 0, 0, 0, 1, 0, 0, 0], np.float64)      # world frame == body frame
    om = oracle.OracleMap(0.2); om.build(small_case["world"].points)
    ox, od, oc, _ = om.knn(q)
    res = {}
    for separate in (0, 1):
        ctx = gpu_capi.LioContext(0.2, max_scan_points=len(q), max_map_points=400_000)
        ctx.map_build(small_case["world"].points)
        ctx.debug_set("knn_tune", 2 if second_chance else 0)
        ctx.debug_set("separate_tree_walk", separate)
        ctx.scan_set(q)
        g = ctx.measure(ident, True, ext)
        g2 = ctx.measure(ident, True, ext)          # the bounded re-search walks its unproven queries the same way
        world, sel, nv = ctx.scan_state()
        nx, nd, nc = ctx.neighbors()
        res[separate] = (g, g2, sel, nv, nx, nd, nc)
        ctx.close()
    a, b = res[0], res[1]
    assert a[0]["n_unresolved"] == b[0]["n_unresolved"] > (300 if second_chance else 1000)
    for k in (0, 1):
        assert a[k]["m"] == b[k]["m"] and np.array_equal(a[k]["HTH"], b[k]["HTH"]) and np.array_equal(a[k]["HTh"], b[k]["HTh"])
    assert np.array_equal(a[0]["HTH"], a[1]["HTH"])
    assert np.array_equal(a[2], b[2]) and bits_equal(a[3], b[3])
    assert bits_equal(a[4], b[4]) and bits_equal(a[5], b[5]) and np.array_equal(a[6], b[6])
    assert np.array_equal(a[6], oc) and bits_equal(a[5], od)
    x6, d6, c6, _ = om.knn(q, 6, 1)
    tie_free = (np.diff(d6, axis=1) > 0).all(1)
    assert bits_equal(a[4][tie_free], ox[tie_free])


@pytest.mark.parametrize("ext,second_chance", [(False, 0), (True, 0), (False, 1), (True, 1)])
def test_walk_changes_a_result_with_few_listed_queries(gpu_capi, oracle, small_case, ext, second_chance):
    """Few listed queries (<= FIX_MAX): the walker workgroup of a listed query walks, evaluates the model on the exact
    neighbours and publishes the query's term of every sum; the last workgroup folds those terms in query order.  Queries
    0.8 m above the ground are such queries, and their walks DO change the result (the 3x3x3 block sees only the ground
    points with z >= 0).  Outputs, accept flags, rows and sums must be the oracle's on the exact neighbours
    (laserMapping.cpp:1909-1938, ikd_Tree.cpp:367-397).  second_chance = 1 (knn_tune bit 1, opt-in): the same queries are finished
    inside knn27_kernel by the 5x5x5 second chance of their wavefront -- nothing is listed, the owners evaluate them like any
    other point, and every output is the same."""
    rng = np.random.default_rng(11)
    half = small_case["world"].ground_half
    wq = oracle.body_to_world(small_case["state14"], small_case["scan"].body)[:3000]
    lifted = (rng.uniform(-0.5 * half, 0.5 * half, (120, 3)) * [1, 1, 0] + [0, 0, 0.8]).astype(np.float32)
    q = np.ascontiguousarray(np.concatenate([wq[:1500], lifted[:60], wq[1500:], lifted[60:]]))
    ident = np.array([0, 0, 0, 0, 0, 0, 1, 0, 0, 0, 1, 0, 0, 0], np.float64)      # world frame == body frame
    om = oracle.OracleMap(0.2); om.build(small_case["world"].points)
    arr = oracle.ScanArrays(len(q))
    valid, hx, h, m, rm = oracle.h_share_model(om, ident, q, arr, True, ext)
    HTH, HTh = oracle.hth(hx, h)
    res = {}
    for separate in (0, 1):
        ctx = gpu_capi.LioContext(0.2, max_scan_points=len(q), max_map_points=400_000)
        ctx.map_build(small_case["world"].points)
        ctx.debug_set("knn_tune", 2 if second_chance else 0)
        ctx.debug_set("separate_tree_walk", separate)
        ctx.measure_want_rows(True)                     # the rows of THIS pass, staged by it (the walkers write their queries' rows)
        ctx.scan_set(q)
        g = ctx.measure(ident, True, ext)
        world, sel, nv = ctx.scan_state()
        nx, nd, nc = ctx.neighbors()
        ghx, gh = ctx.measure_rows(g["m"])
        res[separate] = (g, sel, nv, nx, ghx, gh)
        ctx.close()
    g, sel, nv, nx, ghx, gh = res[0]
    lifted_idx = np.r_[1500:1560, 3060:3120]
    if second_chance:
        assert g["n_unresolved"] <= 2                                  # the lifted queries are settled inside knn27_kernel (5x5x5 block)
    else:
        assert 100 <= g["n_unresolved"] <= 256                        # few listed queries: the lifted ones (and a handful of the scan's)
    assert (nx[lifted_idx, :, 2] < 0).any(axis=1).sum() > 60           # ... whose exact neighbours lie outside the 3x3x3 block (z < 0)
    assert g["m"] == m and np.array_equal(sel, arr.selected)
    sb = sel.astype(bool)
    assert bits_equal(nv[sb], arr.normvec[sb])
    assert np.abs(g["HTH"] - HTH).max() <= 1e-9 * np.abs(HTH).max() and np.abs(g["HTh"] - HTh).max() <= 1e-9 * np.abs(HTh).max()
    assert abs(g["res_mean"] - rm) <= 1e-12
    assert bits_equal(ghx, hx) and bits_equal(gh, h)            # the callback's rows: the walked queries' final rows
    # the stand-alone walk (every model row on the exact neighbours from the start): same flags, same outputs, sums within rounding
    g1, sel1, nv1 = res[1][0], res[1][1], res[1][2]
    assert g1["m"] == g["m"] and np.array_equal(sel1, sel) and bits_equal(nv1, nv)
    assert np.abs(g1["HTH"] - g["HTH"]).max() <= 1e-12 * np.abs(HTH).max()


@pytest.mark.parametrize("case_name,ext", [("small_case", False), ("small_case", True), ("hap_case", True)])
def test_measurement_pass_parity(gpu_capi, oracle, request, case_name, ext):
    """h_share_model: search pass, then a no-search pass at a moved pose (reuses neighbours and the
    previous accept flags, laserMapping.cpp:1906-1917)."""
    case = request.getfixturevalue(case_name)
    w, sc, st = case["world"], case["scan"], case["state14"]
    ctx = gpu_capi.LioContext(0.2, max_scan_points=len(sc.body), max_map_points=400_000)
    ctx.map_build(w.points)
    ctx.scan_set(sc.body)
    om = oracle.OracleMap(0.2); om.build(w.points)
    arr = oracle.ScanArrays(len(sc.body))
    st2 = st.copy(); st2[:3] += [0.012, -0.02, 0.006]
    for search, pose in ((True, st), (False, st2), (True, st2)):
        valid, hx, h, m, rm = oracle.h_share_model(om, pose, sc.body, arr, search, ext)
        g = ctx.measure(pose, search, ext)
        assert g["m"] == m and g["valid"] == valid and m > 1000
        world, sel, nv = ctx.scan_state()
        assert bits_equal(world, arr.world)
        assert np.array_equal(sel, arr.selected)                 # validity mask identical
        s = sel.astype(bool)
        assert bits_equal(nv[s], arr.normvec[s])
        HTH, HTh = oracle.hth(hx, h)
        assert np.abs(g["HTH"] - HTH).max() <= 1e-9 * np.abs(HTH).max()
        assert np.abs(g["HTh"] - HTh).max() <= 1e-9 * np.abs(HTh).max()
        assert np.allclose(g["HTH"], g["HTH"].T, rtol=0, atol=0)
        assert abs(g["res_mean"] - rm) <= 1e-12
        ghx, gh = ctx.measure_rows(m)
        assert np.array_equal(ghx, hx) and np.array_equal(gh, h)  # rows bit-equal, scan order
        if search:
            nx, nd, nc = ctx.neighbors()
            assert np.array_equal(nc, arr.nbr_cnt) and bits_equal(nd, arr.nbr_d) and bits_equal(nx, arr.nbr_xyz)
    ctx.close()


def test_scan_step_call_equals_the_four_calls(gpu_capi, synth, small_case):
    """lio_eskf_scan_step_device == change_x + change_P + lio_scan_set_device + update_iterated_dyn_share_modified."""
    w, sc = small_case["world"], small_case["scan"]
    ctx = gpu_capi.LioContext(0.2, max_scan_points=len(sc.body), max_map_points=400_000)
    ctx.map_build(w.points)
    x0 = gpu_capi.make_state26(small_case["pos"], small_case["rot"], sc.off_R, sc.off_T)
    buf = gpu_capi.scan_to_device(sc.body)
    kf1 = gpu_capi.Esekf(3); kf1.change_x(x0); kf1.change_P(synth.init_cov())
    ctx.scan_set_device(buf.ptr, len(sc.body))
    l1 = kf1.update_iterated_dyn_share_modified(ctx, 0.001, False)
    kf2 = gpu_capi.Esekf(3)
    l2 = gpu_capi.scan_step_device(kf2, ctx, x0, synth.init_cov(), buf.ptr, len(sc.body), 0.001, False)
    assert np.array_equal(kf1.get_x(), kf2.get_x()) and np.array_equal(kf1.get_P(), kf2.get_P())
    assert (l1["n_calls"], l1["n_search"], l1["last_m"]) == (l2["n_calls"], l2["n_search"], l2["last_m"])
    ctx.close()


def test_passes_enqueued_ahead_of_their_pose(gpu_capi, synth, small_case):
    """Inside the iterated update the next pass goes onto the stream while the GPU works on the current one, behind a gate
    kernel that waits for the pose (lio_pass.hip, gate_arm): the posterior, the covariance and the call log must be what the
    plain sequence gives, bit for bit -- when every prediction holds, when every prediction is wrong (each armed pass is called
    off, its kernels stand down), when the update runs fewer passes than the previous one (a pass stays armed at the end and is
    called off), and the stream must be free for map maintenance right after (esekfom.hpp:1636-1938, laserMapping.cpp:2380)."""
    w, sc = small_case["world"], small_case["scan"]
    P0 = synth.init_cov()
    scans, priors = [], []
    for k in range(4):
        s2 = synth.make_scan(w, len(sc.body), seed=synth.SEED + 300 + k, pos=(0.6 * k, -0.3 * k, 0.0), rpy=(0.002, -0.003, 0.04 * k))
        ps, rt = synth.perturb_state(s2, seed=synth.SEED + 400 + k)
        scans.append(s2); priors.append(gpu_capi.make_state26(ps, rt, s2.off_R, s2.off_T))
    out = {}
    # (the wait for the host's word: inside the first kernel of the armed pass where the mail slots are device memory -- the default
    # on a large-BAR box -- or in a gate kernel in front of it; "gate_in_kernel" 0 forces the latter, 1 / 2 = re-use / search passes only)
    for mode in ("plain", "ahead", "wrong", "ahead_gate_kernel", "wrong_gate_kernel", "ahead_in_kernel_reuse_only", "ahead_in_kernel_search_only"):
        ctx = gpu_capi.LioContext(0.2, max_scan_points=len(sc.body), max_map_points=400_000)
        ctx.map_build(w.points)
        if mode == "plain":
            ctx.debug_set("pre_enqueue", 0)
        if mode.startswith("wrong"):
            ctx.debug_set("gate_mispredict", 1)
        if mode.endswith("gate_kernel"):
            ctx.debug_set("gate_in_kernel", 0)
        if mode.endswith("reuse_only"):
            ctx.debug_set("gate_in_kernel", 1)
        if mode.endswith("search_only"):
            ctx.debug_set("gate_in_kernel", 2)
        res = []
        for k, iters in enumerate((3, 3, 1, 3)):               # the third update is shorter than the one before it
            kf = gpu_capi.Esekf(iters)
            buf = gpu_capi.scan_to_device(scans[k].body)
            log = gpu_capi.scan_step_device(kf, ctx, priors[k], P0, buf.ptr, len(scans[k].body), 0.001, False)
            res.append((kf.get_x(), kf.get_P(), (log["n_calls"], log["n_search"], log["last_m"], log["converged"])))
            mi = ctx.map_incremental(kf.get_x()[:14], 0.2, True)          # the stream is free: nothing waits behind a gate
            res.append((np.array([mi["n_add"], mi["n_nodown"]]), np.array([ctx.map_validnum()]), ()))
        out[mode] = (res, ctx.debug_get("gate_released"), ctx.debug_get("gate_called_off"), ctx.debug_get("gate_in_kernel"), ctx.debug_get("gate_mail_in_vram"))
        ctx.close()
    for mode in out:
        for a, b in zip(out["plain"][0], out[mode][0]):
            assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and a[2] == b[2], mode
        if mode.startswith("ahead"):
            assert out[mode][1] == out["ahead"][1] and out[mode][2] == out["ahead"][2], mode
    assert out["ahead_gate_kernel"][3] == 0 and out["wrong_gate_kernel"][3] == 0
    if out["ahead"][4]:
        assert out["ahead"][3] >= out["ahead"][1] and 0 < out["ahead_in_kernel_reuse_only"][3] < out["ahead"][3] and 0 < out["ahead_in_kernel_search_only"][3] < out["ahead"][3]
    assert out["plain"][1] == 0 and out["plain"][2] == 0
    assert out["ahead"][1] >= 3 and out["ahead"][2] >= 1          # passes that ran on a pre-enqueued launch; the one left armed by the short update
    assert out["wrong"][2] >= 3                                       # (a flipped prediction is right where this update differs from the last)


@pytest.mark.parametrize("in_kernel", [3, 0])
def test_a_pass_whose_gate_gave_up_is_put_on_the_stream_again(gpu_capi, small_case, in_kernel):
    """A pass armed ahead of its pose sits behind a gate kernel that gives up after two seconds (a queue must never hang on a host
    that does not come back).  A host that DOES come back late -- a debugger, SIGSTOP, a VM pause -- finds its pass stood down:
    lio_measure_end then sees an idle stream without a record and enqueues the pass directly, once ("gate_reenqueued"), instead
    of failing the iterated update.  Same numbers as the pass that was never armed."""
    import time
    w, sc, st = small_case["world"], small_case["scan"], small_case["state14"]
    st2 = st.copy(); st2[:3] += [0.01, -0.02, 0.005]
    ctx = gpu_capi.LioContext(0.2, max_scan_points=len(sc.body), max_map_points=400_000)
    ctx.map_build(w.points)
    ctx.debug_set("pre_enqueue", 0)
    ctx.scan_set(sc.body)
    ref = [ctx.measure(st, True, False), ctx.measure(st2, False, False)]
    ctx.debug_set("pre_enqueue", 1)
    ctx.debug_set("gate_in_kernel", in_kernel)            # (3: the armed pass waits inside its own first kernel; 0: behind a gate kernel)
    ctx.scan_set(sc.body)                                 # the pattern of the scan above (search, re-use) predicts this one's
    a = ctx.measure(st, True, False)                      # ... arms the re-use pass behind a gate
    assert ctx.debug_get("gate_reenqueued") == 0
    time.sleep(2.4)                                       # the gate gives up
    b = ctx.measure(st2, False, False)
    assert ctx.debug_get("gate_reenqueued") == 1
    for got, want in ((a, ref[0]), (b, ref[1])):
        assert got["m"] == want["m"] and np.array_equal(got["HTH"], want["HTH"]) and np.array_equal(got["HTh"], want["HTh"])
    c = ctx.measure(st2, True, False)                     # and the context goes on as if nothing had happened
    assert c["m"] > 1000
    ctx.close()


@pytest.mark.parametrize("ext", [False, True])
def test_reduction_across_workgroup_counts(gpu_capi, oracle, small_case, ext):
    """The cross-workgroup reduction (groups of 16 workgroups folded by designated reducers, then the group sums)
    for scan sizes that hit its edges: one point, one workgroup, a group of exactly one workgroup behind a full
    group (4097 points = 17 workgroups), partial groups, and the same context reused across sizes (stale stamps of
    larger passes must never be read)."""
    w, sc, st = small_case["world"], small_case["scan"], small_case["state14"]
    ctx = gpu_capi.LioContext(0.2, max_scan_points=len(sc.body), max_map_points=400_000)
    ctx.map_build(w.points)
    om = oracle.OracleMap(0.2); om.build(w.points)
    for n in (len(sc.body), 1, 7, 255, 256, 257, 4096, 4097, 4352, 8193, 12_345, 3):
        body = np.ascontiguousarray(sc.body[:n])
        ctx.scan_set(body)
        arr = oracle.ScanArrays(n)
        for search in (True, False):
            valid, hx, h, m, rm = oracle.h_share_model(om, st, body, arr, search, ext)
            g = ctx.measure(st, search, ext)
            assert g["m"] == m and g["valid"] == valid
            if m == 0:
                continue
            HTH, HTh = oracle.hth(hx, h)
            assert np.abs(g["HTH"] - HTH).max() <= 1e-9 * max(np.abs(HTH).max(), 1e-300)
            assert np.abs(g["HTh"] - HTh).max() <= 1e-9 * max(np.abs(HTh).max(), 1e-300)
            assert abs(g["res_mean"] - rm) <= 1e-12
    ctx.close()


@pytest.mark.parametrize("case_name,ext", [("small_case", False), ("hap_case", True)])
def test_iterated_update_pose_parity(gpu_capi, oracle, synth, request, case_name, ext):
    """update_iterated_dyn_share_modified on the GPU path vs the CPU path: pose within
    1e-4 m / 1e-5 rad (north_star), same number of passes, same final neighbours."""
    case = request.getfixturevalue(case_name)
    w, sc = case["world"], case["scan"]
    ctx = gpu_capi.LioContext(0.2, max_scan_points=len(sc.body), max_map_points=400_000)
    ctx.map_build(w.points)
    om = oracle.OracleMap(0.2); om.build(w.points)
    x0 = gpu_capi.make_state26(case["pos"], case["rot"], sc.off_R, sc.off_T)
    P0 = synth.init_cov()
    xo, Po, arrs, log = oracle.update_iterated(om, x0, P0, sc.body, 0.001, 3, None, ext)
    kf = gpu_capi.Esekf(3)
    kf.change_x(x0); kf.change_P(P0)
    ctx.scan_set(sc.body)
    lg = kf.update_iterated_dyn_share_modified(ctx, 0.001, ext)
    xg, Pg = kf.get_x(), kf.get_P()
    assert lg["n_calls"] == len(log)
    assert np.abs(xg[:3] - xo[:3]).max() < 1e-4
    assert 2 * np.abs(xg[3:7] - xo[3:7]).max() < 1e-5 and 2 * np.abs(xg[7:11] - xo[7:11]).max() < 1e-5
    assert np.abs(xg - xo).max() < 1e-8                          # in practice ~1e-12
    assert np.abs(Pg - Po).max() <= 1e-6 * np.abs(Po).max()
    # the update moves the estimate towards the truth (how far depends on the geometry: the narrow
    # HAP field of view with extrinsic estimation on is weaker than the 360-degree scan)
    assert np.linalg.norm(xg[:3] - sc.pos) < 0.6 * np.linalg.norm(x0[:3] - sc.pos)
    nx, nd, nc = ctx.neighbors()
    assert bits_equal(nx, arrs.nbr_xyz) and np.array_equal(nc, arrs.nbr_cnt)
    ctx.close()


def test_dense_gain_branch_small_scan(gpu_capi, oracle, synth, small_case):
    """m < 23 rows: the reference switches to the dense Kalman gain (esekfom.hpp:1720-1750) and needs
    h_x itself -> lio_measure_rows path."""
    w, sc = small_case["world"], small_case["scan"]
    body = np.ascontiguousarray(sc.body[:15])
    ctx = gpu_capi.LioContext(0.2, max_scan_points=64, max_map_points=400_000)
    ctx.map_build(w.points)
    om = oracle.OracleMap(0.2); om.build(w.points)
    x0 = gpu_capi.make_state26(small_case["pos"], small_case["rot"], sc.off_R, sc.off_T)
    xo, Po, arrs, log = oracle.update_iterated(om, x0, synth.init_cov(), body)
    kf = gpu_capi.Esekf(3); kf.change_x(x0); kf.change_P(synth.init_cov())
    ctx.scan_set(body)
    lg = kf.update_iterated_dyn_share_modified(ctx, 0.001, False)
    assert 0 < lg["last_m"] < 23
    assert np.abs(kf.get_x() - xo).max() < 1e-9 and np.abs(kf.get_P() - Po).max() < 1e-9
    ctx.close()


def test_set_downsample_param(gpu_capi, oracle, small_case):
    """KD_TREE::set_downsample_param (laserMapping.cpp:2332): a context created for 0.5 m and then set to 0.2 m before
    Build behaves like one created for 0.2 m (grid cell included); changed after Build only the Add_Points box changes."""
    if not oracle.ref_available():
        pytest.skip("reference ikd-Tree not built")
    w = small_case["world"]
    rng = np.random.default_rng(3)
    base = np.ascontiguousarray(w.points[::2])
    ctx = gpu_capi.LioContext(0.5, max_scan_points=1024, max_map_points=400_000)
    ctx.set_downsample_param(0.2)
    ctx.map_build(base)
    assert ctx.map_info()["cell_size"] == 0.5 and abs(ctx.map_info()["downsample_size"] - 0.2) < 1e-7
    ref = oracle.RefIkdTree(0.2); ref.build(base)
    batch = (w.points[1::2][:6000] + rng.uniform(-0.02, 0.02, (6000, 3))).astype(np.float32)
    assert ctx.map_add(batch, True) == ref.add(batch, True)
    assert ctx.map_validnum() == ref.validnum()
    ctx.set_downsample_param(0.4)                       # coarser down-sampling from now on, same grid
    ref2 = oracle.RefIkdTree(0.4); ref2.build(ctx.map_flatten())
    batch2 = (w.points[1::2][6000:12000] + rng.uniform(-0.02, 0.02, (6000, 3))).astype(np.float32)
    assert ctx.map_add(batch2, True) == ref2.add(batch2, True)
    assert ctx.map_validnum() == ref2.validnum() and ctx.map_info()["cell_size"] == 0.5
    with pytest.raises(gpu_capi.LioError, match="LIO_ERR_INVALID_ARG"):
        ctx.set_downsample_param(0.8)                   # a box larger than the cell of the built grid is refused
    ctx.close()


def test_downsample_boundary_points_fuzz_seeds(gpu_capi, oracle):
    """Add_Points(downsample) next to voxel boundaries, where the reference's fp32 boxes misbehave: a point outside
    its own box (seed 1000: two identical points both survive), a map point inside the boxes of two neighbouring
    voxels that both receive points (seed 5025).  Cases of tests/fuzz/fuzz_map.py (random worlds at kilometre
    coordinates, random mutation sequences) replayed against the reference ikd-Tree."""
    if not oracle.ref_available():
        pytest.skip("reference ikd-Tree not built")
    import importlib.util
    import os
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "fuzz", "fuzz_map.py")
    spec = importlib.util.spec_from_file_location("fuzz_map", path)
    fz = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(fz)
    for seed in (1000, 5025, 5000, 5005, 5010, 5015):
        fz.run_case(seed)


def test_measurement_fuzz_seeds(gpu_capi, oracle):
    """A few cases of tests/fuzz/fuzz_measure.py (random worlds up to kilometre offsets, random scans with outliers, both
    extrinsic modes, scan sizes on the reduction's edges) against the CPU path on the reference ikd-Tree."""
    if not oracle.ref_available():
        pytest.skip("reference ikd-Tree not built")
    import importlib.util
    import os
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "fuzz", "fuzz_measure.py")
    spec = importlib.util.spec_from_file_location("fuzz_measure", path)
    fz = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(fz)
    for seed in (2000, 2004, 2008, 2013, 2021):
        fz.run_case(seed)


def test_node_loop_fuzz_seeds(gpu_capi, oracle):
    """A few cases of tests/fuzz/fuzz_loop.py: the per-scan loop (fov segment, iterated update, map_incremental, both
    Add_Points) at kilometre offsets with the REFERENCE ikd-Tree as the CPU side's map."""
    if not oracle.ref_available():
        pytest.skip("reference ikd-Tree not built")
    import importlib.util
    import os
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "fuzz", "fuzz_loop.py")
    spec = importlib.util.spec_from_file_location("fuzz_loop", path)
    fz = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(fz)
    for seed in (7001, 7002, 7004):
        fz.run_case(seed)


def test_scan_preparation_fuzz_seeds(gpu_capi, oracle):
    """Forty cases of tests/fuzz/fuzz_scanprep.py: raw clouds from 1 to 60 k points with time ties, clumps (leaves with
    hundreds of points), non-finite points, pose lists of 2..25 entries, leaf sizes from 0.1 to 3 m, the cloud read
    before or after the filter -- compensation vs the restatement, voxel grid bit-exact, PCL's overflow refusal; half of the cases
    prepared between lio_scan_prep_begin / _end beside an insert batch on the main stream (round 5)."""
    import importlib.util
    import os
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "fuzz", "fuzz_scanprep.py")
    spec = importlib.util.spec_from_file_location("fuzz_scanprep", path)
    fz = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(fz)
    ctx = gpu_capi.LioContext(0.2, max_scan_points=fz.ctx_cap, max_map_points=2_000_000)
    ctx.map_build((np.random.default_rng(1).uniform(-40, 40, (40_000, 3)) * [1, 1, 0.05]).astype(np.float32))   # (the overlapped cases insert into it)
    seen = set()
    for seed in range(30000, 30040):
        seen.add(fz.run_case(seed, ctx))
    assert "ok" in seen
    assert fz.n_overlapped >= 10 and ctx.debug_get("prep_overlapped") >= 1      # (round 5: half of the cases go through lio_scan_prep_begin / _end)
    ctx.close()


def test_pass_control_fuzz_seeds(gpu_capi):
    """Eight cases of tests/fuzz/fuzz_pipeline.py: the pipelined scan step with everything armed (passes ahead of their pose,
    the next scan's first search across the scan boundary) against the plain sequence on a twin context -- scans of changing
    sizes, updates of 1..4 iterations, calls between two scans, switches of the armed forms, forced mispredictions: equal bits
    after every scan, equal maps at the end."""
    import importlib.util
    import os
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "fuzz", "fuzz_pipeline.py")
    spec = importlib.util.spec_from_file_location("fuzz_pipeline", path)
    fz = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(fz)
    tot = np.zeros(4, np.int64)
    for seed in range(62000, 62008):
        tot += np.array(fz.run_case(seed), np.int64)
    assert tot[0] > 0 and tot[1] > 0 and tot[2] > 0 and tot[3] > 0      # armed, released, released by word, called off: all exercised


def test_map_mutation_parity(gpu_capi, oracle, small_case):
    """Add_Points (down-sampled: one point per voxel, nearest the centre; plain), Delete_Point_Boxes,
    validnum/size/flatten: same return values and live set as the oracle, k-NN parity afterwards."""
    w, sc = small_case["world"], small_case["scan"]
    rng = np.random.default_rng(21)
    base = w.points[:80000]
    ctx = gpu_capi.LioContext(0.2, max_scan_points=1024, max_map_points=400_000)
    ctx.map_build(base)
    om = oracle.OracleMap(0.2); om.build(base)
    wq = oracle.body_to_world(small_case["state14"], sc.body)
    key = lambda p: p[np.lexsort((p[:, 2], p[:, 1], p[:, 0]))]
    for rnd in range(3):
        new = np.ascontiguousarray(wq[rnd * 5000:(rnd + 1) * 5000] + rng.normal(0, 0.04, (5000, 3)).astype(np.float32))
        # repeated points and several points per voxel inside one batch (order dependence)
        new[100:200] = new[0:100]
        new[200:300] = new[0:100] + np.float32(0.01)
        assert ctx.map_add(new, True) == om.add(new, True)
        plain = np.ascontiguousarray(rng.uniform(-35, 35, (300, 3)).astype(np.float32))
        assert ctx.map_add(plain, False) == om.add(plain, False) == 0
        boxes = np.array([[-12 + 3 * rnd, -10, -1, 4, 2 + rnd, 3], [3, 3, -1, 9, 9 + rnd, 2.5]], np.float32)
        assert ctx.map_delete_boxes(boxes) == om.delete_boxes(boxes)
        assert ctx.map_validnum() == om.validnum()
        assert bits_equal(key(ctx.map_flatten()), key(om.flatten()))
        q = np.ascontiguousarray(wq[15000 + rnd * 1000: 16000 + rnd * 1000])
        gx, gd, gc = ctx.nearest_search(q)
        ox, od, oc, _ = om.knn(q)
        assert np.array_equal(gc, oc) and bits_equal(gd, od) and bits_equal(gx, ox)
    assert ctx.map_size() >= ctx.map_validnum()
    ctx.close()


@pytest.mark.parametrize("seed", [3, 4])
def test_add_points_dense_voxels(gpu_capi, oracle, small_case, seed):
    """Voxel groups of hundreds of points in ONE batch (an un-down-sampled scan on a near wall; eight replicas' lists): the
    fold over such a group runs 64 points at a time (prefix-min, lio_mapadd.hip) and must leave what the reference's
    one-insert-after-the-other rule leaves (ikd_Tree.cpp:422-449): same survivor, same tmp_counter -- with exact
    duplicates, points equal to map points, equal distances to the voxel centre (mirror images: the later one wins),
    voxels the map already holds zero / one / several points of, and group sizes on both sides of the 16-point and
    64-point edges."""
    w = small_case["world"]
    rng = np.random.default_rng(seed)
    base = np.ascontiguousarray(w.points[:60000])
    ctx = gpu_capi.LioContext(0.2, max_scan_points=1024, max_map_points=400_000)
    ctx.map_build(base)
    ref = oracle.RefIkdTree(0.2) if oracle.ref_available() else None
    om = oracle.OracleMap(0.2); om.build(base)
    if ref is not None:
        ref.build(base)
    key = lambda p: p[np.lexsort((p[:, 2], p[:, 1], p[:, 0]))]
    for rnd in range(3):
        parts = []
        sizes = [1, 2, 15, 16, 17, 63, 64, 65, 130, 400, 900] + list(rng.integers(1, 40, 30))
        for n in sizes:
            if rng.random() < 0.6:
                anchor = base[rng.integers(0, len(base))]            # a voxel the map knows
            else:
                anchor = rng.uniform(-20, 20, 3).astype(np.float32) + np.float32([0, 0, 60])   # an empty one, far above the map
            lo = np.floor(anchor / np.float32(0.2)) * np.float32(0.2)
            pts = (lo + rng.uniform(0.01, 0.19, (n, 3))).astype(np.float32)
            if n >= 4:
                pts[n // 2] = pts[0]                                  # exact duplicates inside the group
                mid = lo + np.float32(0.1)
                pts[n - 1] = (2 * mid - pts[1]).astype(np.float32)    # (nearly) the same distance to the centre, later in the batch
            if n >= 16 and rng.random() < 0.5:
                pts[3] = anchor                                       # a point equal to a map point
            parts.append(pts)
        batch = np.concatenate(parts)
        batch = np.ascontiguousarray(batch[rng.permutation(len(batch))])     # group members scattered over the batch
        c_gpu = ctx.map_add(batch, True)
        assert c_gpu == om.add(batch, True)
        if ref is not None:
            assert c_gpu == ref.add(batch, True)
            assert ctx.map_validnum() == ref.validnum()
            ref.quiesce(50)
            assert bits_equal(key(ctx.map_flatten()), key(ref.flatten()))
        assert bits_equal(key(ctx.map_flatten()), key(om.flatten()))
    ctx.close()


@pytest.mark.parametrize("shape", ["wide", "slab", "clumps"])
def test_add_points_large_batches(gpu_capi, oracle, small_case, shape):
    """Batches beyond the one-launch rank sort (lio_gsort.hip: partition by the top key digit + LDS bucket sorts): a wide batch
    (every bucket small), a thin slab along the axis of the top digit (ONE bucket holds everything: the bucket that does not
    fit LDS) and dense clumps (buckets on both sides of the LDS capacity, large voxel groups).  Down-sampled adds, plain adds
    and the pair, against the reference tree: Add_Points' counter, validnum, the live set, searches afterwards."""
    w = small_case["world"]
    rng = np.random.default_rng(11)
    base = np.ascontiguousarray(w.points[:60000])
    ctx = gpu_capi.LioContext(0.2, max_scan_points=1024, max_map_points=600_000)
    ctx.map_build(base)
    om = oracle.OracleMap(0.2); om.build(base)
    ref = oracle.RefIkdTree(0.2) if oracle.ref_available() else None
    if ref is not None:
        ref.build(base)
    key = lambda p: p[np.lexsort((p[:, 2], p[:, 1], p[:, 0]))]

    def batch(n):
        if shape == "wide":
            b = rng.uniform(-40, 40, (n, 3)); b[:, 2] = rng.uniform(-1, 3, n)
        elif shape == "slab":
            b = np.stack([rng.uniform(5.0, 5.3, n), rng.uniform(-60, 60, n), rng.uniform(-2, 10, n)], 1)
        else:
            centres = rng.uniform(-30, 30, (6, 3)); centres[:, 2] = rng.uniform(0, 2, 6)
            b = centres[rng.integers(0, 6, n)] + rng.normal(0, 0.6, (n, 3))
        return np.ascontiguousarray(b.astype(np.float32))

    for rnd, n in enumerate((30000, 9000, 70000)):
        add, plain = batch(n), batch(n // 3)
        if rnd == 0:
            assert ctx.map_add(add, True) == om.add(add, True)
            assert ctx.map_add(plain, False) == om.add(plain, False) == 0
        else:
            cnt = ctx.map_add_pair(add, plain)
            assert cnt == om.add(add, True)
            om.add(plain, False)
        if ref is not None:
            ref.add(add, True); ref.add(plain, False)
            assert ref.validnum() == ctx.map_validnum()
        assert ctx.map_validnum() == om.validnum()
        assert bits_equal(key(ctx.map_flatten()), key(om.flatten()))
        q = batch(2000)
        gx, gd, gc = ctx.nearest_search(q)
        ox, od, oc, _ = om.knn(q)
        assert np.array_equal(gc, oc) and bits_equal(gd, od)
    if ref is not None:
        ref.quiesce(100)
        assert bits_equal(key(ctx.map_flatten()), key(ref.flatten()))
    ctx.close()


def test_stamp_wrap_around(gpu_capi, synth, small_case):
    """The cross-workgroup hand-offs are stamped with 32-bit sequence numbers (partial sums of the measurement kernel, tile
    counts of the single-launch scans and of the group sort).  When a number goes round, words stamped 2^32 launches ago must
    not pass for fresh ones -- the scratch starts over.  Driven across the wrap with the sequence numbers set just below it:
    a large scan first (its workgroups leave stamps a smaller scan's grid never overwrites), then smaller scans and map
    updates across the boundary; every result equals the one of a context that never wrapped."""
    w = small_case["world"]
    big = synth.make_scan(w, 20_000)
    small = [synth.make_scan(w, 3_000 + 500 * k, seed=synth.SEED + 60 + k, pos=(2.0 + k, -1.0, 0.0), rpy=(0.0, 0.0, 0.3 + 0.1 * k)) for k in range(4)]
    res = []
    for wrap in (False, True):
        ctx = gpu_capi.LioContext(0.2, max_scan_points=20_000, max_map_points=400_000)
        ctx.map_build(np.ascontiguousarray(w.points[::2]))
        if wrap:
            ctx.debug_set("sequence_numbers", 2 ** 32 - 3)
        out = []
        st = synth.state_vector(*synth.perturb_state(big), big.off_R, big.off_T)
        ctx.scan_set(big.body)
        out.append(ctx.measure(st, True, False)["HTH"].copy())
        for sc in small:
            st = synth.state_vector(*synth.perturb_state(sc), sc.off_R, sc.off_T)
            ctx.scan_set(sc.body)
            g1 = ctx.measure(st, True, False)
            g2 = ctx.measure(st, False, False)
            r = ctx.map_incremental(st, 0.2, True, apply=True)
            out.append((g1["m"], g1["HTH"].copy(), g2["HTH"].copy(), r["n_add"], r["n_nodown"], r["add_counter"], ctx.map_validnum()))
        out.append(ctx.map_flatten().copy())
        res.append(out)
        ctx.close()
    a, b = res
    assert np.array_equal(a[0], b[0]) and bits_equal(a[-1], b[-1])
    for x, y in zip(a[1:-1], b[1:-1]):
        assert x[0] == y[0] and np.array_equal(x[1], y[1]) and np.array_equal(x[2], y[2]) and x[3:] == y[3:]


def test_edge_cases(gpu_capi, oracle, small_case):
    sc, st = small_case["scan"], small_case["state14"]
    ctx = gpu_capi.LioContext(0.2, max_scan_points=1000, max_map_points=10_000)
    # empty map: no neighbours, measurement invalid ("No Effective Points", laserMapping.cpp:1956)
    q = np.ascontiguousarray(sc.body[:10])
    x, d, c = ctx.nearest_search(q)
    assert (c == 0).all() and np.isinf(d).all() and np.isnan(x).all()
    ctx.scan_set(sc.body[:100])
    g = ctx.measure(st, True, False)
    assert g["m"] == 0 and not g["valid"]
    # fewer than 5 map points
    pts = np.array([[0, 0, 0], [1, 0, 0], [0, 2, 0]], np.float32)
    ctx.map_build(pts)
    om = oracle.OracleMap(0.2); om.build(pts)
    qq = np.array([[0.1, 0.1, 0.1], [10, 10, 10]], np.float32)
    x, d, c = ctx.nearest_search(qq)
    ox, od, oc, _ = om.knn(qq)
    assert np.array_equal(c, oc) and bits_equal(d, od) and bits_equal(x[:, :3], ox[:, :3])
    g = ctx.measure(st, True, False)
    assert g["m"] == 0 and not g["valid"]
    # empty scan
    ctx.scan_set(np.zeros((0, 3), np.float32))
    assert not ctx.measure(st, True, False)["valid"]
    # a pass without a preceding search of the same scan is refused, as is over-capacity input
    ctx.scan_set(sc.body[:100])
    with pytest.raises(gpu_capi.LioError, match="LIO_ERR_NO_SCAN"):
        ctx.measure(st, False, False)
    with pytest.raises(gpu_capi.LioError, match="LIO_ERR_CAPACITY"):
        ctx.scan_set(sc.body[:2000])
    with pytest.raises(gpu_capi.LioError, match="LIO_ERR_CAPACITY"):
        ctx.map_build(np.zeros((20_000, 3), np.float32))
    # ragged size (not a multiple of the 256-point workgroup) and duplicate map points
    dup = np.repeat(np.array([[1, 1, 1], [1.05, 1, 1], [1, 1.05, 1]], np.float32), 3, 0)
    ctx.map_build(dup)
    om = oracle.OracleMap(0.2); om.build(dup)
    x, d, c = ctx.nearest_search(np.array([[1.01, 1.0, 1.02]], np.float32))
    ox, od, oc, _ = om.knn(np.array([[1.01, 1.0, 1.02]], np.float32))
    assert np.array_equal(c, oc) and bits_equal(d, od)       # ties in distance: same multiset of distances
    ctx.close()


def test_run_to_run_determinism(gpu_capi, ctx_small, small_case):
    sc, st = small_case["scan"], small_case["state14"]
    ctx_small.scan_set(sc.body)
    a = ctx_small.measure(st, True, True)
    b = ctx_small.measure(st, True, True)
    assert np.array_equal(a["HTH"], b["HTH"]) and np.array_equal(a["HTh"], b["HTh"]) and a["res_mean"] == b["res_mean"]


def test_full_size_properties(gpu_capi, synth):
    """BASELINE config 2 (100 k-point scan vs 5 M-point map): properties that need no oracle run.
    - every reported sqdist is the fp32 distance of the reported point, ascending, and the gate,
    - a brute-force check of 200 random queries against all 5 M points,
    - the candidate census C_mean of the byte model is in the surveyed range,
    - two runs give identical bits, and the update converges to the true pose."""
    nm, nq = 5_000_000, 100_000
    world = synth.make_world(nm, 0.2)
    scan = synth.make_scan(world, nq)
    pos, rot = synth.perturb_state(scan)
    st = synth.state_vector(pos, rot, scan.off_R, scan.off_T)
    ctx = gpu_capi.LioContext(0.2, max_scan_points=nq, max_map_points=nm + 1000)
    ctx.map_build(world.points)
    assert ctx.map_validnum() == nm
    ctx.census_enable(True)
    ctx.scan_set(scan.body)
    g = ctx.measure(st, True, False)
    ctx.census_enable(False)
    assert 40 < g["n_candidates"] / nq < 80
    assert g["m"] > 0.8 * nq
    world_pts, sel, nv = ctx.scan_state()
    nx, nd, nc = ctx.neighbors()
    assert (nc == 5).all()
    diff = nx.astype(np.float32) - world_pts[:, None, :]
    d = (diff[..., 0] * diff[..., 0] + diff[..., 1] * diff[..., 1]) + diff[..., 2] * diff[..., 2]
    assert bits_equal(d.astype(np.float32), nd)
    assert (np.diff(nd, axis=1) >= 0).all()
    rng = np.random.default_rng(0)
    mp = world.points
    for i in rng.choice(nq, 200, replace=False):
        dd = mp - world_pts[i]
        dist = (dd[:, 0] * dd[:, 0] + dd[:, 1] * dd[:, 1]) + dd[:, 2] * dd[:, 2]
        best = np.sort(dist)[:5]
        assert bits_equal(best.astype(np.float32), nd[i])
    g2 = ctx.measure(st, True, False)
    assert np.array_equal(g["HTH"], g2["HTH"]) and g["m"] == g2["m"]
    kf = gpu_capi.Esekf(3)
    kf.change_x(gpu_capi.make_state26(pos, rot, scan.off_R, scan.off_T)); kf.change_P(synth.init_cov())
    ctx.scan_set(scan.body)
    lg = kf.update_iterated_dyn_share_modified(ctx, 0.001, False)
    assert lg["n_calls"] <= 4
    assert np.linalg.norm(kf.get_x()[:3] - scan.pos) < 0.02
    ctx.close()


def test_map_incremental_sequence(gpu_capi, oracle, synth, small_case):
    """Three consecutive scans through the reference's loop body: iterated update, then
    map_incremental (laserMapping.cpp:1440-1496) with Add_Points(down-sampled) + Add_Points(plain).
    Lists, counters and the live set must equal the CPU path after every scan; the map starts from a
    THINNED world so that the scans really add points."""
    w = small_case["world"]
    base = np.ascontiguousarray(w.points[::3])                     # a sparser initial map
    ctx = gpu_capi.LioContext(0.2, max_scan_points=20_000, max_map_points=400_000)
    ctx.map_build(base)
    om = oracle.OracleMap(0.2); om.build(base)
    key = lambda p: p[np.lexsort((p[:, 2], p[:, 1], p[:, 0]))]
    for k, (px, py, yaw) in enumerate([(3.1, -2.2, 0.52), (4.0, -1.6, 0.60), (5.2, -1.1, 0.71)]):
        sc = synth.make_scan(w, 12_000, seed=synth.SEED + 40 + k, pos=(px, py, 0.0), rpy=(0.003, -0.004, yaw))
        pos, rot = synth.perturb_state(sc, dpos=0.05, drot_deg=0.3, seed=synth.SEED + 50 + k)
        x0 = gpu_capi.make_state26(pos, rot, sc.off_R, sc.off_T)
        # CPU path
        xo, Po, arrs, log = oracle.update_iterated(om, x0, synth.init_cov(), sc.body)
        world_post = oracle.body_to_world(xo[:14], sc.body)
        cls = oracle.map_incremental_classify(world_post, arrs.nbr_xyz, arrs.nbr_cnt, 0.2, True)
        to_add, nodown = world_post[cls == 1], world_post[cls == 2]
        # GPU path
        kf = gpu_capi.Esekf(3); kf.change_x(x0); kf.change_P(synth.init_cov())
        ctx.scan_set(sc.body)
        kf.update_iterated_dyn_share_modified(ctx, 0.001, False)
        xg = kf.get_x()
        assert np.abs(xg - xo).max() < 1e-8
        # same posterior bits are needed for bit-equal world points: feed the oracle's (the two agree to ~1e-12)
        r = ctx.map_incremental(xo[:14], 0.2, True, apply=True, want_lists=True)
        assert r["n_add"] == len(to_add) and r["n_nodown"] == len(nodown)
        assert bits_equal(r["add"], to_add) and bits_equal(r["nodown"], nodown)
        c1 = om.add(np.ascontiguousarray(to_add), True)
        om.add(np.ascontiguousarray(nodown), False)
        assert r["add_counter"] == c1
        assert ctx.map_validnum() == om.validnum()
        assert bits_equal(key(ctx.map_flatten()), key(om.flatten()))
        assert r["n_add"] > 500                                  # the thinned map does get filled in
    ctx.close()


def test_incremental_index_stress(gpu_capi, oracle, small_case):
    """The O(batch) index update (buckets re-written at the tail, new cells inserted into the table and
    the tree levels) against the reference ikd-Tree over a long mutation sequence: small batches (incremental
    path), a large one (retired-slot budget -> full rebuild), box deletes in between, territory the map never
    had (new cells, then cells outside the root of the tree).  After every step: validnum, the live set, and
    Nearest_Search for near AND far queries (the far ones take the tree walk through the new cells)."""
    if not oracle.ref_available():
        pytest.skip("reference ikd-Tree not built")
    w = small_case["world"]
    rng = np.random.default_rng(4242)
    base = np.ascontiguousarray(w.points[::2])
    ctx = gpu_capi.LioContext(0.2, max_scan_points=4096, max_map_points=600_000)
    ctx.map_build(base)
    ref = oracle.RefIkdTree(0.2); ref.build(base)
    key = lambda p: p[np.lexsort((p[:, 2], p[:, 1], p[:, 0]))]
    lo, hi = base.min(0), base.max(0)

    def queries(extra):
        near = base[rng.choice(len(base), 600, replace=False)] + rng.normal(0, 0.05, (600, 3))
        far = rng.uniform(lo - 40.0, hi + 40.0, (300, 3))
        far[:, 2] = rng.uniform(-3.0, 25.0, 300)
        return np.concatenate([near, far, extra]).astype(np.float32)

    def check(extra=np.zeros((0, 3), np.float32)):
        assert ctx.map_validnum() == ref.validnum()
        q = queries(extra)
        gx, gd, gc = ctx.nearest_search(q)
        rx, rd, rc, _ = ref.knn(q, 5)
        assert np.array_equal(gc, rc)
        assert bits_equal(gd, rd)                       # distances always; points where the 5th/6th do not tie
        rx6, rd6, rc6, _ = ref.knn(q, 6)
        tie_free = np.array([len(np.unique(rd6[i, :rc6[i]])) == rc6[i] for i in range(len(q))])
        assert tie_free.mean() > 0.95 and bits_equal(gx[tie_free], rx[tie_free])

    info0 = ctx.map_info()
    # 1. small down-sampled batches: jittered copies of map points and points in the gaps of the thinned map
    for k in range(6):
        src = w.points[1::2][rng.choice(len(w.points) // 2, 2500, replace=False)]
        batch = (src + rng.uniform(-0.02, 0.02, src.shape)).astype(np.float32)
        assert ctx.map_add(batch, True) == ref.add(batch, True)
        if k % 2 == 1:
            c0 = np.array([rng.uniform(lo[0] + 5, hi[0] - 5), rng.uniform(lo[1] + 5, hi[1] - 5), 1.0])
            box = np.concatenate([c0 - 2.5, c0 + 2.5]).astype(np.float32)[None]
            assert ctx.map_delete_boxes(box) == ref.delete_boxes(box)
        ref.quiesce(100)
        check()
    info1 = ctx.map_info()
    assert info1["n_incremental_updates"] - info0["n_incremental_updates"] == 6
    assert info1["n_full_rebuilds"] == info0["n_full_rebuilds"] and info1["n_retired"] > 0
    # 2. plain (no down-sampling) inserts into territory next to the map: new cells, new tree nodes
    strip = rng.uniform([hi[0] + 1.0, lo[1], 0.0], [hi[0] + 30.0, hi[1], 6.0], (4000, 3)).astype(np.float32)
    assert ctx.map_add(strip, False) == ref.add(strip, False) == 0
    ref.quiesce(100)
    check(strip[:200] + np.float32(0.01))
    info2 = ctx.map_info()
    assert info2["n_cells"] > info1["n_cells"]
    # 3. a batch large enough to exceed the retired-slot budget (here: the budget of round 2, a map this small never reaches
    #    the default of half its live points): full rebuild, same answers
    ctx.debug_set("retired_budget", 65536)
    big = (w.points[1::2] + rng.uniform(-0.02, 0.02, w.points[1::2].shape)).astype(np.float32)
    assert ctx.map_add(big, True) == ref.add(big, True)
    ref.quiesce(300)
    info3 = ctx.map_info()
    assert info3["n_full_rebuilds"] == info2["n_full_rebuilds"] + 1 and info3["n_retired"] == 0
    check()
    # 4. points kilometres away (this map straddles the origin, so its root node already spans the whole grid)
    away = (rng.uniform(-3.0, 3.0, (500, 3)) + np.array([4000.0, -2500.0, 10.0])).astype(np.float32)
    assert ctx.map_add(away, False) == ref.add(away, False) == 0
    ref.quiesce(100)
    check(away[:100] + np.float32(0.02))
    assert bits_equal(key(ctx.map_flatten()), key(ref.flatten()))
    ctx.close()


def test_rebuild_by_cell_runs_equals_the_point_sort(gpu_capi, small_case, monkeypatch):
    """Build / reconstruct / the full rebuild of a mutated map through the sort of CELL RUNS (lio_rebuild.hip: the default) against the
    63-bit point sort it replaces (LIO_REBUILD_BY_CELLS=0): the same point array slot for slot (flatten is order-preserving), the same
    cell and level counts, the same neighbours for near and far queries -- after a Build of points in arbitrary order (every point
    its own run), after a forced rebuild behind incremental updates and box deletes (buckets at the tail, tomb-stones inside
    buckets, pending points), and with a far-away cluster in the map (a key frame with many more bits)."""
    w = small_case["world"]
    rng = np.random.default_rng(606)
    base = np.ascontiguousarray(w.points[::2])
    ctxs = []
    # the default; the point sort; cell runs with a host-side live count that is off by one (the rebuild restarts from the device's count)
    for by_cells, miscount in (("1", "0"), ("0", "0"), ("1", "1")):
        monkeypatch.setenv("LIO_REBUILD_BY_CELLS", by_cells)
        monkeypatch.setenv("LIO_TEST_LIVE_MISCOUNT", miscount)
        ctxs.append(gpu_capi.LioContext(0.2, max_scan_points=4096, max_map_points=600_000))
    monkeypatch.delenv("LIO_REBUILD_BY_CELLS")
    monkeypatch.delenv("LIO_TEST_LIVE_MISCOUNT")
    lo, hi = base.min(0), base.max(0)
    q = np.concatenate([base[rng.choice(len(base), 800, replace=False)] + rng.normal(0, 0.05, (800, 3)),
                        rng.uniform(lo - 40.0, hi + 40.0, (400, 3))]).astype(np.float32)

    def same(extra=None):
        a = ctxs[0]
        ia, fa = a.map_info(), a.map_flatten()
        qq = q if extra is None else np.concatenate([q, extra]).astype(np.float32)
        ax, ad, ac = a.nearest_search(qq)
        for b in ctxs[1:]:
            ib = b.map_info()
            for k in ("n_points", "n_live", "n_cells", "n_retired", "n_full_rebuilds"):
                assert ia[k] == ib[k], k
            assert bits_equal(fa, b.map_flatten())
            bx, bd, bc = b.nearest_search(qq)
            assert np.array_equal(ac, bc) and bits_equal(ad, bd) and bits_equal(ax, bx)

    for c in ctxs:
        c.map_build(base)
    same()
    shuffled = base[np.random.default_rng(1).permutation(len(base))]
    for c in ctxs:
        c.map_reconstruct(shuffled)
    same()
    for k in range(4):
        src = w.points[1::2][rng.choice(len(w.points) // 2, 3000, replace=False)]
        batch = (src + rng.uniform(-0.02, 0.02, src.shape)).astype(np.float32)
        c0 = np.array([rng.uniform(lo[0] + 5, hi[0] - 5), rng.uniform(lo[1] + 5, hi[1] - 5), 1.0])
        box = np.concatenate([c0 - 3.0, c0 + 3.0]).astype(np.float32)[None]
        for c in ctxs:
            c.map_add(batch, True)
            c.map_delete_boxes(box)
    info = [c.map_info() for c in ctxs]
    assert info[0]["n_retired"] > 0 and info[0]["n_incremental_updates"] > 0
    for c in ctxs:
        c.debug_set("retired_budget", 1024)
    strip = rng.uniform([hi[0] + 1.0, lo[1], 0.0], [hi[0] + 20.0, hi[1], 6.0], (3000, 3)).astype(np.float32)
    for c in ctxs:
        c.map_add(strip, False)                                   # over the budget: the full rebuild takes the pending strip in
    assert ctxs[0].map_info()["n_full_rebuilds"] == info[0]["n_full_rebuilds"] + 1 and ctxs[0].map_info()["n_retired"] == 0
    same(strip[:200] + np.float32(0.01))
    for c in ctxs:
        c.debug_set("retired_budget", 0)
    away = (rng.uniform(-3.0, 3.0, (500, 3)) + np.array([4000.0, -2500.0, 10.0])).astype(np.float32)
    for c in ctxs:
        c.map_add(away, False)
        c.map_reconstruct(c.map_flatten())
    same(away[:100] + np.float32(0.02))
    # a sparse map three kilometres wide: tens of thousands of level-2 cells, roots at level 6 (a frame of few parents at a high level,
    # every tree level sized from the counts of the copy kernel); far queries walk the tree through every level
    wide = (rng.uniform(-1500.0, 1500.0, (60_000, 3)) * [1, 1, 0.004]).astype(np.float32)
    for c in ctxs:
        c.map_build(wide)
        c.map_reconstruct(c.map_flatten())                        # (in order now: the rebuild by cell runs)
    q = np.concatenate([wide[:600] + np.float32(0.03), (rng.uniform(-1700.0, 1700.0, (600, 3)) * [1, 1, 0.01])]).astype(np.float32)
    same()
    for c in ctxs:
        c.close()


def test_incremental_insert_outside_tree_root(gpu_capi, oracle, small_case):
    """A map inside one octant has a small root node; a new cell outside it cannot be hung into the tree
    levels, so the update falls back to a full rebuild with a higher root -- and far queries (tree walk)
    still find the new points."""
    if not oracle.ref_available():
        pytest.skip("reference ikd-Tree not built")
    rng = np.random.default_rng(99)
    base = (small_case["world"].points[::4] + np.array([1500.0, 1200.0, 300.0])).astype(np.float32)
    ctx = gpu_capi.LioContext(0.2, max_scan_points=1024, max_map_points=200_000)
    ctx.map_build(base)
    ref = oracle.RefIkdTree(0.2); ref.build(base)
    near = (base[:2000] + rng.uniform(-0.05, 0.05, (2000, 3))).astype(np.float32)
    assert ctx.map_add(near, True) == ref.add(near, True)
    i1 = ctx.map_info()
    assert i1["n_incremental_updates"] == 1 and i1["n_full_rebuilds"] == 1
    away = (rng.uniform(-3.0, 3.0, (400, 3)) + np.array([9000.0, 7000.0, 350.0])).astype(np.float32)
    assert ctx.map_add(away, False) == ref.add(away, False) == 0
    ref.quiesce(100)
    assert ctx.map_info()["n_full_rebuilds"] == 2
    q = np.concatenate([away[:150] + np.float32(0.02), base[:150] + np.float32(0.03),
                        rng.uniform([1400, 1100, 250], [9100, 7100, 400], (200, 3))]).astype(np.float32)
    gx, gd, gc = ctx.nearest_search(q)
    rx, rd, rc, _ = ref.knn(q, 5)
    assert np.array_equal(gc, rc) and bits_equal(gd, rd)
    assert ctx.map_validnum() == ref.validnum()
    ctx.close()


def test_map_incremental_before_first_search(gpu_capi, small_case):
    """No neighbour lists yet (EKF not initialised / first scans): every point is a PointToAdd (:1487)."""
    sc, st = small_case["scan"], small_case["state14"]
    ctx = gpu_capi.LioContext(0.2, max_scan_points=4096, max_map_points=50_000)
    ctx.scan_set(sc.body[:3000])
    r = ctx.map_incremental(st, 0.2, False, apply=True)
    assert r["n_add"] == 3000 and r["n_nodown"] == 0
    assert 0 < ctx.map_validnum() <= 3000
    ctx.close()


def test_replicated_apply_from_device_lists(gpu_capi, oracle, synth, small_case):
    """The multi-GPU map-maintenance path on one GPU: map_incremental's lists stay in HBM, pass through
    replicate.exchange_inserts_device (identity without a process group; RCCL all-gather with one) and
    are applied with lio_map_add_pair_device -- same live set as applying map_incremental directly."""
    import torch
    from better_fastlio2_amd import replicate
    w, sc = small_case["world"], small_case["scan"]
    base = np.ascontiguousarray(w.points[::3])
    key = lambda p: p[np.lexsort((p[:, 2], p[:, 1], p[:, 0]))]
    ctxs = [gpu_capi.LioContext(0.2, max_scan_points=20_000, max_map_points=400_000) for _ in range(2)]
    x0 = gpu_capi.make_state26(small_case["pos"], small_case["rot"], sc.off_R, sc.off_T)
    posts = []
    for c in ctxs:
        c.map_build(base)
        kf = gpu_capi.Esekf(3); kf.change_x(x0); kf.change_P(synth.init_cov())
        c.scan_set(sc.body)
        kf.update_iterated_dyn_share_modified(c, 0.001, False)
        posts.append(kf.get_x()[:14])
    assert np.array_equal(posts[0], posts[1])                     # same inputs -> same bits on both replicas
    r = ctxs[0].map_incremental(posts[0], 0.2, True, apply=True)
    pa, na, pb, nb = ctxs[1].map_incremental_device(posts[1], 0.2, True)
    assert (na, nb) == (r["n_add"], r["n_nodown"])
    dev = torch.device("cuda", 0)
    A, ca = replicate.exchange_inserts_device(pa, na, 20_000, dev)
    B, cb = replicate.exchange_inserts_device(pb, nb, 20_000, dev)
    torch.cuda.synchronize()
    assert ca == [na] and cb == [nb] and A.shape == (na, 4)
    cnt = ctxs[1].map_add_pair_device(A.data_ptr(), na, B.data_ptr(), nb)
    assert cnt == r["add_counter"]
    assert ctxs[0].map_validnum() == ctxs[1].map_validnum()
    assert bits_equal(key(ctxs[0].map_flatten()), key(ctxs[1].map_flatten()))
    for c in ctxs:
        c.close()


def test_fov_segment_sliding_cube(gpu_capi, oracle, small_case):
    """lasermap_fov_segment (laserMapping.cpp:1136-1200): the local-map cube follows the LiDAR and the
    slabs left behind are deleted -- same boxes, same delete counts, same live set as the CPU path."""
    w = small_case["world"]
    ctx = gpu_capi.LioContext(0.2, max_scan_points=1024, max_map_points=400_000)
    ctx.map_build(w.points)
    om = oracle.OracleMap(0.2); om.build(w.points)
    fov = oracle.FovSegment()
    key = lambda p: p[np.lexsort((p[:, 2], p[:, 1], p[:, 0]))]
    cube_len, det_range = 60.0, 15.0               # 1.5 * 15 = 22.5 m from a face triggers a move of 6.75 m
    moved = 0
    for k in range(12):
        pos = np.array([-20.0 + 5.5 * k, 3.0 - 2.0 * k, 1.7])
        boxes_o = fov.step(pos, cube_len, det_range)
        boxes_g, nd_g = ctx.fov_segment(pos, cube_len, det_range)
        assert boxes_g.shape == boxes_o.shape and bits_equal(boxes_g, boxes_o)
        nd_o = om.delete_boxes(boxes_o) if len(boxes_o) else 0
        assert nd_g == nd_o
        moved += len(boxes_o)
        assert ctx.map_validnum() == om.validnum()
    assert moved >= 3
    assert bits_equal(key(ctx.map_flatten()), key(om.flatten()))
    ctx.close()


def test_acquire_removed_points(gpu_capi, oracle, small_case):
    """KD_TREE::acquire_removed_points (ikd_Tree.cpp:559-571): box-deleted points are handed out exactly once,
    by the first call after the deletion; points replaced inside a down-sample voxel are not reported
    (ikd_Tree.cpp:1339).  The reference reports a point only when a rebuild happens to flatten its sub-tree,
    so its list must be a subset of ours; ours is the full set of box-deleted points."""
    w = small_case["world"]
    rng = np.random.default_rng(77)
    ctx = gpu_capi.LioContext(0.2, max_scan_points=1024, max_map_points=400_000)
    ctx.map_build(w.points)
    key = lambda p: p[np.lexsort((p[:, 2], p[:, 1], p[:, 0]))]
    assert len(ctx.acquire_removed_points()) == 0
    boxes = np.array([[-12.0, -9.0, -1.0, -4.0, 2.0, 2.5], [5.0, 1.0, -0.5, 9.5, 6.0, 4.0]], np.float32)
    inside = np.zeros(len(w.points), bool)
    for b in boxes:
        inside |= np.all((w.points >= b[:3]) & (w.points < b[3:]), axis=1)
    assert inside.sum() > 500
    half = int(inside.sum())
    assert ctx.map_delete_boxes(boxes[:1]) + ctx.map_delete_boxes(boxes[1:]) == half
    first = ctx.acquire_removed_points()                       # handed out while still tomb-stones in the array
    in0 = np.all((w.points >= boxes[0, :3]) & (w.points < boxes[0, 3:]), axis=1)
    assert 0 < in0.sum() < half and len(first) == half
    # an insert batch with down-sampling: replaces points inside voxels (never reported)
    batch = (w.points[~inside][rng.choice((~inside).sum(), 3000, replace=False)] +
             rng.uniform(-0.03, 0.03, (3000, 3))).astype(np.float32)
    ctx.map_add(batch, True)
    got = first
    assert bits_equal(key(got), key(w.points[inside]))
    assert len(ctx.acquire_removed_points()) == 0              # handed out once, the down-sample replacements never
    # a later delete is reported by the next call only
    box3 = np.array([[-3.0, -3.0, -1.0, 3.0, 3.0, 3.0]], np.float32)
    nd = ctx.map_delete_boxes(box3)
    assert nd > 100 and len(ctx.acquire_removed_points()) == nd
    if oracle.ref_available():
        ref = oracle.RefIkdTree(0.2)
        ref.build(w.points)
        assert ref.delete_boxes(boxes) == int(inside.sum())
        ref.add(batch, True)
        ref.quiesce(300)
        rr = ref.acquire_removed()
        ours = {tuple(p) for p in got.tolist()}
        assert all(tuple(p) in ours for p in rr.tolist())
        assert ref.delete_boxes(box3) == nd
        assert ref.validnum() == ctx.map_validnum()
    ctx.close()


@pytest.mark.parametrize("lanes", [0, 4, 8, 16, 32, 64])
@pytest.mark.parametrize("shift", [2e-4, 5e-3, 0.03, 0.4])
def test_repeated_search_stays_exact(gpu_capi, oracle, small_case, shift, lanes):
    """Second and third searches of the same scan (esekfom.hpp:1824-1838) are bounded by the previous neighbours
    (knn27_kernel<.., HINT>): Nearest_Points must equal a fresh search at the new pose -- for pose changes from the
    filter's convergence scale (0.2 mm) to far beyond it (0.4 m, where the old neighbours bound almost nothing), and
    for every lanes-per-query width (with 4 lanes one lane has to re-measure two of the five old neighbours)."""
    w, sc, st = small_case["world"], small_case["scan"], small_case["state14"]
    ctx = gpu_capi.LioContext(0.2, max_scan_points=20_000, max_map_points=400_000, lanes_per_query=lanes)
    ctx.map_build(w.points)
    om = oracle.OracleMap(0.2); om.build(w.points)
    ctx.scan_set(sc.body)
    ctx.measure(st, True, False)
    rng = np.random.default_rng(int(shift * 1e6))
    for rep in range(2):
        st2 = np.array(st, dtype=np.float64)
        st2[0:3] += rng.normal(0, 1, 3) * shift / np.sqrt(3)
        ang = shift * 0.2
        dq = np.array([ang / 2 * 0.6, -ang / 2 * 0.3, ang / 2 * 0.74, 1.0]); dq /= np.linalg.norm(dq)
        x, y, z, wq = st2[3:7]
        st2[3:7] = [wq * dq[0] + x * dq[3] + y * dq[2] - z * dq[1], wq * dq[1] - x * dq[2] + y * dq[3] + z * dq[0],
                    wq * dq[2] + x * dq[1] - y * dq[0] + z * dq[3], wq * dq[3] - x * dq[0] - y * dq[1] - z * dq[2]]
        g = ctx.measure(st2, True, False)
        gx, gd, gc = ctx.neighbors()
        wq2 = oracle.body_to_world(st2, sc.body)
        ox, od, oc, _ = om.knn(wq2)
        assert np.array_equal(gc, oc)
        assert bits_equal(gd, od)
        x6, d6, c6, _ = om.knn(wq2, 6, 1)
        tie_free = (np.diff(d6, axis=1) > 0).all(1)
        assert tie_free.mean() > 0.99 and bits_equal(gx[tie_free], ox[tie_free])
        arr = oracle.ScanArrays(len(sc.body))
        valid, hx, h, m, rm = oracle.h_share_model(om, st2, sc.body, arr, True, False)
        assert g["m"] == m
        st = st2
    ctx.close()


def test_search_after_map_mutation_ignores_stale_bound(gpu_capi, oracle, small_case):
    """The repeated-search bound (previous neighbours) must be dropped when the map changes between two
    searches of the same scan: delete a slab that holds many of the previous neighbours, search again."""
    w, sc, st = small_case["world"], small_case["scan"], small_case["state14"]
    ctx = gpu_capi.LioContext(0.2, max_scan_points=20_000, max_map_points=400_000)
    ctx.map_build(w.points)
    om = oracle.OracleMap(0.2); om.build(w.points)
    ctx.scan_set(sc.body)
    arr = oracle.ScanArrays(len(sc.body))
    oracle.h_share_model(om, st, sc.body, arr, True, False)
    ctx.measure(st, True, False)
    box = np.array([[-6.0, -8.0, -1.0, 9.0, 4.0, 3.0]], np.float32)       # around the sensor: most neighbours live here
    assert ctx.map_delete_boxes(box) == om.delete_boxes(box) > 1000
    valid, hx, h, m, rm = oracle.h_share_model(om, st, sc.body, arr, True, False)
    g = ctx.measure(st, True, False)
    assert g["m"] == m
    nx, nd, nc = ctx.neighbors()
    assert np.array_equal(nc, arr.nbr_cnt) and bits_equal(nd, arr.nbr_d) and bits_equal(nx, arr.nbr_xyz)
    # ... and with an unchanged map the bounded search gives the same bits as the unbounded one
    st2 = st.copy(); st2[:3] += [0.02, -0.015, 0.004]
    oracle.h_share_model(om, st2, sc.body, arr, True, False)
    ctx.measure(st2, True, False)
    nx, nd, nc = ctx.neighbors()
    assert np.array_equal(nc, arr.nbr_cnt) and bits_equal(nd, arr.nbr_d) and bits_equal(nx, arr.nbr_xyz)
    ctx.close()


def _imu_poses(n_poses=14, dt=0.0085):
    """A smooth IMUpose list (Pose6D records) for the undistortion tests: gentle turn + acceleration."""
    poses = np.zeros((n_poses, 22))
    w = np.array([0.02, -0.015, 0.35]); acc = np.array([0.4, 0.1, 0.05]); v0 = np.array([2.0, 0.3, 0.0])
    for k in range(n_poses):
        t = k * dt
        ang = np.linalg.norm(w) * t
        ax = w / np.linalg.norm(w)
        Kx = np.array([[0, -ax[2], ax[1]], [ax[2], 0, -ax[0]], [-ax[1], ax[0], 0]])
        R = np.eye(3) + np.sin(ang) * Kx + (1 - np.cos(ang)) * Kx @ Kx
        poses[k, 0] = t
        poses[k, 1:4] = acc + [0, 0, 0.01 * k]; poses[k, 4:7] = w + [0.001 * k, 0, 0]
        poses[k, 7:10] = v0 + acc * t; poses[k, 10:13] = v0 * t + 0.5 * acc * t * t
        poses[k, 13:22] = R.ravel()
    return poses


def test_imu_front_end_feeds_the_scan_step(gpu_capi, oracle, synth, small_case):
    """The whole step before the path, as the node runs it (laserMapping.cpp:2256, 2322-2325, 2380):
    ImuProcess::Process -- forward propagation on the host, point compensation on the device -- then the voxel
    filter and the iterated update, against the restatements (oracle/imu_model.py for the IMU side)."""
    from oracle import imu_model as im
    w, sc = small_case["world"], small_case["scan"]
    rng = np.random.default_rng(21)
    n = len(sc.body)
    ctx = gpu_capi.LioContext(0.2, max_scan_points=20_000, max_map_points=400_000)
    ctx.map_build(w.points)
    om = oracle.OracleMap(0.2); om.build(w.points)
    Rm = im.qmat(np.asarray(sc.off_R, np.float64))
    p = gpu_capi.ImuProcess(); p.set_extrinsic(sc.off_T, Rm); p.set_cov(0.1, 0.1, 0.0001, 0.0001)
    o = im.ImuProcess(); o.Lidar_T, o.Lidar_R = np.array(sc.off_T, np.float64), np.array(Rm, np.float64)
    kf = gpu_capi.Esekf(3)
    xo, Po = kf.get_x(), kf.get_P()

    def imu_rows(t0, t1, moving):
        rows = []
        for t in np.arange(t0, t1, 0.005):
            acc = np.array([0.0, 0.0, 9.79]) + rng.normal(0, 0.01, 3)
            gyr = np.array([0.001, -0.002, 0.0005]) + rng.normal(0, 0.001, 3)
            if moving:
                acc = acc + [0.8 * np.sin(9 * t), 0.5, 0.1]
                gyr = gyr + [0.03, -0.02, 0.3]
            rows.append(np.concatenate([[t], acc, gyr]))
        return np.array(rows)

    # scan 0: static initialisation (no cloud is produced)
    t_ms = np.sort(rng.uniform(0, 100.0, n)).astype(np.float32)
    raw = np.concatenate([sc.body, t_ms[:, None]], 1).astype(np.float32)
    imu0 = imu_rows(10.0, 10.1, False)
    assert p.process(kf, ctx, 10.0, 10.1, imu0, raw) is False
    xo, Po, ready = o.process(imu0, 10.0, 10.1, xo, Po)
    assert not ready and not p.info()["need_init"]
    # put the initialised filter at the scan's pose (the node starts at the origin; the map here does not)
    x = kf.get_x(); x[0:7] = small_case["state14"][0:7]; kf.change_x(x); xo[0:7] = x[0:7]
    # scan 1: the sensor turns and accelerates while the scan is taken
    rng.shuffle(raw)                                             # Process must time-sort the points itself
    imu1 = imu_rows(10.1 - 0.002, 10.2 + 0.003, True)
    assert p.process(kf, ctx, 10.1, 10.1 + float(raw[:, 3].max()) / 1000.0, imu1, raw) is True
    xo, Po, ready = o.process(imu1, 10.1, 10.1 + float(raw[:, 3].max()) / 1000.0, xo, Po)
    assert ready
    assert np.abs(kf.get_x() - xo).max() < 1e-11 and np.abs(kf.get_P() - Po).max() < 1e-11
    poses = np.array(o.poses)
    assert np.abs(p.poses() - poses).max() < 1e-11
    und_g = ctx.scan_cloud(0)
    und_o, ts = oracle.undistort(raw, poses, np.concatenate([xo[0:14]]))
    assert und_g.shape == und_o.shape == (n, 3)
    assert np.abs(und_g - und_o).max() < 1e-5 and (und_g.view(np.uint32) == und_o.view(np.uint32)).mean() > 0.999
    assert np.abs(und_g - raw[np.argsort(raw[:, 3], kind="stable"), :3]).max() > 0.02      # the motion was compensated
    # down-sample and update from the propagated state and covariance
    m = ctx.scan_voxel_filter(0.5)
    down_o = oracle.voxel_grid(und_g, 0.5)
    assert m == len(down_o) and bits_equal(ctx.scan_cloud(1), down_o)
    log = kf.update_iterated_dyn_share_modified(ctx, 0.001, False)
    xu, Pu, arrs, olog = oracle.update_iterated(om, xo, Po, down_o)
    assert log["n_calls"] == len(olog) and log["last_m"] > 100
    assert np.abs(kf.get_x()[0:3] - xu[0:3]).max() < 1e-4 and np.abs(kf.get_x()[3:7] - xu[3:7]).max() < 0.5e-5
    assert np.abs(kf.get_P() - Pu).max() < 1e-6 * np.abs(Pu).max()
    ctx.close()


def test_scan_undistort_and_voxel_filter(gpu_capi, oracle, synth, small_case):
    """The step before the path: UndistortPcl's point side and pcl::VoxelGrid on the
    device vs the oracle restatement (both "parity unpinned": Eigen / PCL are un-vendored).  Then the
    down-sampled scan drives a scan step exactly like a scan handed over with lio_scan_set."""
    w, sc = small_case["world"], small_case["scan"]
    rng = np.random.default_rng(12)
    n = len(sc.body)
    t_ms = rng.uniform(0, 100.0, n).astype(np.float32)
    t_ms[:5] = 0.0                                              # points at the scan start are left untouched (:351)
    t_ms[5:9] = t_ms[9]                                         # equal time stamps: stable order
    raw = np.concatenate([sc.body, t_ms[:, None]], 1).astype(np.float32)
    poses = _imu_poses()
    end = synth.state_vector(poses[-1, 10:13] + [0.01, 0, 0], synth._quat_from_rot(poses[-1, 13:22].reshape(3, 3)), sc.off_R, sc.off_T)
    ctx = gpu_capi.LioContext(0.2, max_scan_points=20_000, max_map_points=400_000)
    ctx.map_build(w.points)
    ctx.scan_undistort(raw, poses, end)
    und_g = ctx.scan_cloud(0)
    # the same sweep from page-locked memory (read in place by the copy engine, no staging copy): same bits
    ctx.scan_undistort(gpu_capi.pinned_points(raw), poses, end)
    assert bits_equal(ctx.scan_cloud(0), und_g)
    und_o, ts = oracle.undistort(raw, poses, end)
    assert und_g.shape == und_o.shape == (n, 3)
    # fp64 arithmetic with device sin/cos vs glibc: sub-ulp differences may flip an fp32 rounding
    assert np.abs(und_g - und_o).max() < 1e-5
    assert (und_g.view(np.uint32) == und_o.view(np.uint32)).mean() > 0.999
    assert np.abs(und_g - sc.body[np.argsort(t_ms, kind="stable")]).max() > 0.05      # it did move points
    # voxel grid: bit-exact given the same input cloud
    for leaf in (0.5, 0.2):
        ctx.scan_undistort(np.concatenate([und_o, ts[:, None]], 1).astype(np.float32))   # upload only
        m = ctx.scan_voxel_filter(leaf)
        down_o = oracle.voxel_grid(und_o, leaf)
        down_g = ctx.scan_cloud(1)
        assert m == len(down_o) == len(down_g) and m < n
        assert bits_equal(down_g, down_o)
    # the filtered cloud is the scan: a measurement pass on it equals lio_scan_set of the same points
    st = small_case["state14"]
    g1 = ctx.measure(st, True, False)
    ctx.scan_set(down_o)
    g2 = ctx.measure(st, True, False)
    assert g1["m"] == g2["m"] > 100 and np.array_equal(g1["HTH"], g2["HTH"])
    # non-finite points are dropped, an absurd leaf size is refused like PCL's overflow check
    bad = np.concatenate([und_o[:100], [[np.nan, 0, 0], [np.inf, 1, 1]]]).astype(np.float32)
    ctx.scan_undistort(np.concatenate([bad, np.zeros((len(bad), 1), np.float32)], 1))
    assert ctx.scan_voxel_filter(0.5) == len(oracle.voxel_grid(bad, 0.5))
    with pytest.raises(gpu_capi.LioError, match="LIO_ERR_CAPACITY"):
        ctx.scan_voxel_filter(1e-4)
    ctx.close()


def test_node_loop_composition(gpu_capi, oracle, synth, small_case):
    """The reference's main loop body (laserMapping.cpp:2320-2401), five scans of a short drive:
    lasermap_fov_segment -> VoxelGrid down-sampling -> iterated update -> map_incremental, every stage on the
    GPU path vs the CPU path, state carried from scan to scan (posterior -> next prior, map -> next map)."""
    w = small_case["world"]
    base = np.ascontiguousarray(w.points[::2])
    ctx = gpu_capi.LioContext(0.2, max_scan_points=20_000, max_map_points=400_000)
    ctx.map_build(base)
    om = oracle.OracleMap(0.2); om.build(base)
    fov = oracle.FovSegment()
    key = lambda p: p[np.lexsort((p[:, 2], p[:, 1], p[:, 0]))]
    P_g = P_o = synth.init_cov()
    for k in range(5):
        sc = synth.make_scan(w, 15_000, seed=synth.SEED + 70 + k, pos=(2.0 + 1.1 * k, -2.0 + 0.3 * k, 0.0), rpy=(0.002, -0.003, 0.5 + 0.04 * k))
        # the prior of every scan = truth (+) a small error, as a good IMU prediction would give (IMU
        # propagation is the host's job and not under test); covariance and map are carried over
        pos, rot = synth.perturb_state(sc, dpos=0.04, drot_deg=0.2, seed=synth.SEED + 80 + k)
        x_g = x_o = gpu_capi.make_state26(pos, rot, sc.off_R, sc.off_T)
        pos_lid = x_o[:3]
        # lasermap_fov_segment
        bo = fov.step(pos_lid, 70.0, 20.0)
        bg, nd = ctx.fov_segment(pos_lid, 70.0, 20.0)
        assert bits_equal(bg, bo) and nd == (om.delete_boxes(bo) if len(bo) else 0)
        # VoxelGrid down-sampling of the (already compensated) scan
        raw = np.concatenate([sc.body, np.zeros((len(sc.body), 1), np.float32)], 1)
        ctx.scan_undistort(raw)
        m = ctx.scan_voxel_filter(0.3)
        down = oracle.voxel_grid(sc.body, 0.3)
        assert m == len(down) and bits_equal(ctx.scan_cloud(1), down)
        # iterated update
        xo, Po, arrs, log = oracle.update_iterated(om, x_o, P_o, down)
        kf = gpu_capi.Esekf(3); kf.change_x(x_g); kf.change_P(P_g)
        lg = kf.update_iterated_dyn_share_modified(ctx, 0.001, False)
        xg, Pg = kf.get_x(), kf.get_P()
        assert lg["n_calls"] == len(log)
        assert np.abs(xg[:3] - xo[:3]).max() < 1e-4 and 2 * np.abs(xg[3:7] - xo[3:7]).max() < 1e-5
        assert np.abs(xg - xo).max() < 1e-6                      # rounding-level (different 23x23 elimination orders)
        assert np.abs(Pg - Po).max() <= 1e-6 * np.abs(Po).max()
        assert np.linalg.norm(xg[:3] - sc.pos) < 0.06            # sanity only (tight carried-over covariance)
        # map_incremental at the CPU path's posterior (bit-equal world points need identical pose bits)
        wp = oracle.body_to_world(xo[:14], down)
        cls = oracle.map_incremental_classify(wp, arrs.nbr_xyz, arrs.nbr_cnt, 0.2, True)
        r = ctx.map_incremental(xo[:14], 0.2, True, apply=True)
        c1 = om.add(np.ascontiguousarray(wp[cls == 1]), True)
        om.add(np.ascontiguousarray(wp[cls == 2]), False)
        assert (r["n_add"], r["n_nodown"], r["add_counter"]) == (int((cls == 1).sum()), int((cls == 2).sum()), c1)
        assert ctx.map_validnum() == om.validnum()
        # both paths continue from the same prior, so that every scan is an independent bit-level comparison
        P_g, P_o = Po, Po
    assert bits_equal(key(ctx.map_flatten()), key(om.flatten()))
    ctx.close()


def test_contexts_release_their_memory(gpu_capi, small_case):
    """lio_destroy gives back everything a context took, including what grew on demand (map-add staging,
    removed-point lists, the nearest-search batch buffers, scan preparation): device memory free before
    == after a dozen create / use / destroy cycles."""
    import ctypes as C
    gpu_capi.load()
    hip = C.CDLL("libamdhip64.so.7")
    hip.hipMemGetInfo.argtypes = [C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]

    def free_bytes():
        f, t = C.c_size_t(0), C.c_size_t(0)
        assert hip.hipMemGetInfo(C.byref(f), C.byref(t)) == 0
        return f.value

    w, sc, st = small_case["world"], small_case["scan"], small_case["state14"]

    def cycle():
        ctx = gpu_capi.LioContext(0.2, max_scan_points=20_000, max_map_points=400_000)
        ctx.map_build(w.points[::2])
        ctx.map_add(w.points[1::2][:30_000], True)
        ctx.scan_set(sc.body)
        ctx.measure(st, True, False)
        ctx.map_incremental(st, 0.2, True, apply=True, want_lists=False)
        ctx.nearest_search(w.points[:1000])
        ctx.map_delete_boxes(np.array([[0, 0, -1, 5, 5, 3]], np.float32))
        ctx.acquire_removed_points()
        raw = np.concatenate([sc.body, np.zeros((len(sc.body), 1), np.float32)], 1)
        ctx.scan_undistort(raw)
        ctx.scan_voxel_filter(0.5)
        ctx.close()

    cycle()                                   # first use pays for runtime-internal pools (code objects, rocPRIM)
    before = free_bytes()
    for _ in range(12):
        cycle()
    after = free_bytes()
    assert before - after < (8 << 20), (before, after)      # nothing of the ~100 MB per context stays behind


@pytest.mark.parametrize("variant", ["armed", "armed_wait_in_kernel", "not_armed", "map_update_between_scans"])
def test_first_search_of_the_next_scan_is_armed_across_the_scan_boundary(gpu_capi, synth, small_case, variant):
    """lio_eskf_scan_step_pipelined puts the first search of the NEXT scan (already uploaded) onto the stream behind the last
    passes of the current one (gate_arm_next_scan, lio_pass.hip): lio_scan_swap keeps it armed, the scan's first lio_measure_begin
    hands it its pose.  Bit-identical to the plain sequence -- also when the previous update mispredicts where this one ends (the
    third update is shorter, the fourth longer than the one before), when anything else uses the stream between two scans (the
    armed search is called off and launched directly) and with the switch off."""
    w, sc = small_case["world"], small_case["scan"]
    P0 = synth.init_cov()
    scans, priors = [], []
    for k in range(6):
        s2 = synth.make_scan(w, len(sc.body), seed=synth.SEED + 700 + k, pos=(0.5 * k, 0.2 * k, 0.0), rpy=(0.001, -0.002, 0.03 * k))
        ps, rt = synth.perturb_state(s2, seed=synth.SEED + 800 + k)
        scans.append(s2); priors.append(gpu_capi.make_state26(ps, rt, s2.off_R, s2.off_T))
    iters = (3, 3, 1, 3, 3, 3)
    between = variant == "map_update_between_scans"

    def run(pipelined):
        ctx = gpu_capi.LioContext(0.2, max_scan_points=len(sc.body), max_map_points=400_000)
        ctx.map_build(w.points)
        if variant == "not_armed":
            ctx.debug_set("arm_next_scan", 0)
        if variant == "armed_wait_in_kernel":
            ctx.debug_set("gate_in_kernel", 7)
        pinned = [gpu_capi.pinned_points(s.body) for s in scans]
        res = []
        if pipelined:
            ctx.scan_upload_async(pinned[0])
        for k in range(len(scans)):
            kf = gpu_capi.Esekf(iters[k])
            if pipelined:
                log = gpu_capi.scan_step_pipelined(kf, ctx, priors[k], P0, pinned[k + 1] if k + 1 < len(scans) else None, 0.001, False)
            else:
                buf = gpu_capi.scan_to_device(scans[k].body)
                log = gpu_capi.scan_step_device(kf, ctx, priors[k], P0, buf.ptr, len(scans[k].body), 0.001, False)
            res.append((kf.get_x(), kf.get_P(), (log["n_calls"], log["n_search"], log["last_m"], log["converged"])))
            if between:
                mi = ctx.map_incremental(kf.get_x()[:14], 0.2, True)
                res.append((np.array([mi["n_add"], mi["n_nodown"]]), np.array([ctx.map_validnum()]), ()))
        counters = (ctx.debug_get("next_scan_armed"), ctx.debug_get("next_scan_released"), ctx.debug_get("gate_called_off"))
        ctx.synchronize()                                   # (nothing left waiting: returns at once)
        ctx.close()
        return res, counters

    plain, c0 = run(False)
    piped, c1 = run(True)
    assert c0[0] == 0 and c0[1] == 0
    for a, b in zip(plain, piped):
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and a[2] == b[2]
    if variant == "not_armed":
        assert c1[0] == 0 and c1[1] == 0
    elif between:
        assert c1[0] >= 3 and c1[1] == 0                    # armed, called off by map_incremental every time
    else:
        assert c1[0] >= 3 and 2 <= c1[1] <= c1[0], c1        # (a mispredicted boundary: armed too early, called off, launched directly)


@pytest.mark.gpu
@pytest.mark.parametrize("nq", [6_000, 20_000])
def test_map_update_without_the_host_round_trip_equals_the_one_with_it(gpu_capi, synth, small_case, nq):
    """Round 6: lio_map_incremental(apply = 1, no lists wanted) launches the whole map update -- lists, both Add_Points, the index
    update -- without reading the list lengths and the bounding box back in the middle (list lengths and key frames stay on the
    device, BatchDesc).  Twin contexts, one with lio_debug_set("fast_update", 0): same counts, same map (flattened, in order, bitwise),
    same statistics after every scan of a short drive (the twin's path is the one test_map_incremental_sequence and the fuzz seeds
    hold against the reference tree: laserMapping.cpp:1440-1496, ikd_Tree.cpp:414-489).  6 000 points: the one-launch rank sort;
    20 000: the group sort, whose key bits now come from the device too."""
    w = small_case["world"]
    ctxs = [gpu_capi.LioContext(0.2, max_scan_points=nq, max_map_points=500_000) for _ in range(2)]
    ctxs[1].debug_set("fast_update", 0)
    for c in ctxs:
        c.map_build(w.points)
    for k in range(5):
        sc = synth.make_scan(w, nq, seed=synth.SEED + 300 + k, pos=(1.0 + 0.7 * k, -0.5 + 0.3 * k, 0.0), rpy=(0.002, -0.003, 0.3 + 0.04 * k))
        st = synth.state_vector(sc.pos, sc.rot, sc.off_R, sc.off_T)
        res = []
        for c in ctxs:
            c.scan_set(sc.body)
            c.measure(st, True, False)
            res.append(c.map_incremental(st, 0.2, True, apply=True, want_lists=False))
        assert res[0] == res[1], (k, res)
        assert res[0]["n_add"] + res[0]["n_nodown"] > 0
        f0, f1 = ctxs[0].map_flatten(), ctxs[1].map_flatten()
        assert bits_equal(f0, f1), k
        i0, i1 = ctxs[0].map_info(), ctxs[1].map_info()
        for key in ("n_points", "n_live", "n_cells", "n_retired", "n_incremental_updates", "n_full_rebuilds", "n_boundary_points"):
            assert i0[key] == i1[key], (k, key, i0[key], i1[key])
    # (the first update of a context has no previous lists to size its launches by: it takes the round trip)
    assert ctxs[0].debug_get("fast_updates") + ctxs[0].debug_get("fast_update_overflows") == 4 and ctxs[0].debug_get("fast_updates") >= 2
    assert ctxs[1].debug_get("fast_updates") == 0
    # the same scan again, no search in between (its neighbours are the old ones: most voxels are taken now) ...
    res = [c.map_incremental(st, 0.2, True, apply=True, want_lists=False) for c in ctxs]
    assert res[0] == res[1]
    assert bits_equal(ctxs[0].map_flatten(), ctxs[1].map_flatten())
    for c in ctxs:
        c.close()
    # A list longer than the launches were sized for: fresh twins, a scan of 150 points first (its lists are what the next update is
    # sized by), then a full scan from elsewhere -- the list kernel says so on the device, every kernel behind it does nothing, the
    # lists go through the round trip
    ctxs = [gpu_capi.LioContext(0.2, max_scan_points=nq, max_map_points=500_000) for _ in range(2)]
    ctxs[1].debug_set("fast_update", 0)
    sc = synth.make_scan(w, nq, seed=synth.SEED + 399, pos=(-9.0, 7.0, 0.0), rpy=(0.0, 0.0, -2.0))
    st = synth.state_vector(sc.pos, sc.rot, sc.off_R, sc.off_T)
    for c in ctxs:
        c.map_build(w.points)
    for body in (np.ascontiguousarray(sc.body[:150]), sc.body):
        res = []
        for c in ctxs:
            c.scan_set(body)
            c.measure(st, True, False)
            res.append(c.map_incremental(st, 0.2, True, apply=True, want_lists=False))
        assert res[0] == res[1]
        assert bits_equal(ctxs[0].map_flatten(), ctxs[1].map_flatten())
    assert res[0]["n_add"] + res[0]["n_nodown"] > 1024, res[0]
    assert ctxs[0].debug_get("fast_update_overflows") == 1 and ctxs[0].debug_get("fast_updates") == 0
    i0, i1 = ctxs[0].map_info(), ctxs[1].map_info()
    assert i0["n_live"] == i1["n_live"] and i0["n_points"] == i1["n_points"] and i0["n_incremental_updates"] == i1["n_incremental_updates"]
    for c in ctxs:
        c.close()
