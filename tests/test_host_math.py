"""The arithmetic the GPU kernels execute (csrc/lio_math.h, csrc/lio_common.h) and the host-side
filter (csrc/lio_esekf.cpp), compiled for the host by tests/csrc/host_harness.cpp and compared
with the oracle -- bit for bit where the oracle is fp-deterministic."""
import ctypes as C
import os
import subprocess

import numpy as np

from conftest import bits_equal

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
from test_oracle_math import _planar_sets


def test_plane_fit_bit_exact(oracle, host_harness):
    sets = _planar_sets(5000, seed=9)
    # degenerate inputs: collinear, duplicated, tiny, huge offsets
    sets.append(np.array([[i, 2 * i, 3 * i] for i in range(5)], np.float32))
    sets.append(np.tile(np.array([[1.5, -2.0, 0.25]], np.float32), (5, 1)))
    sets.append((np.random.default_rng(0).normal(size=(5, 3)) * 1e-3).astype(np.float32))
    sets.append((np.random.default_rng(1).normal(size=(5, 3)) + 5e3).astype(np.float32))
    for pts in sets:
        ok1, p1 = oracle.esti_plane(pts, 0.1)
        p2 = np.zeros(4, np.float32)
        ok2 = host_harness.hh_plane_fit(np.ascontiguousarray(pts), 0.1, p2)
        assert ok1 == bool(ok2)
        assert bits_equal(p1, p2) or (np.isnan(p1).any() and np.isnan(p2).any())


def test_world_transform_bit_exact(oracle, host_harness, small_case):
    sc = small_case["scan"]
    a = oracle.body_to_world(small_case["state14"], sc.body)
    b = np.zeros_like(a)
    host_harness.hh_body_to_world(small_case["state14"], sc.body, len(sc.body), b)
    assert bits_equal(a, b)


def test_rows_and_selection_bit_exact(oracle, host_harness, hap_case):
    """Per-point gate + plane + residual + Jacobian row == oracle h_share_model, both extrinsic modes."""
    w, sc, st = hap_case["world"], hap_case["scan"], hap_case["state14"]
    om = oracle.OracleMap(0.2); om.build(w.points)
    body = sc.body[:3000]
    for ext in (False, True):
        arr = oracle.ScanArrays(len(body))
        valid, hx, h, m, rm = oracle.h_share_model(om, st, body, arr, True, ext)
        rows, hs, sel = [], [], []
        for i in range(len(body)):
            if arr.nbr_cnt[i] < 5 or arr.nbr_d[i, 4] > 5:
                sel.append(0); continue
            nv = np.zeros(4, np.float32); row = np.zeros(12); hh = np.zeros(1)
            r = host_harness.hh_point(st, body[i].copy(), arr.world[i].copy(), np.ascontiguousarray(arr.nbr_xyz[i]),
                                      int(ext), nv, row, hh)
            sel.append(r)
            if r:
                rows.append(row); hs.append(hh[0])
                assert bits_equal(nv, arr.normvec[i])
        assert np.array_equal(np.array(sel, np.uint8), arr.selected)
        assert m == len(rows) and m > 500
        assert np.array_equal(np.array(rows), hx) and np.array_equal(np.array(hs), h)
        if ext:
            assert np.abs(hx[:, 6:]).max() > 0
        else:
            assert np.abs(hx[:, 6:]).max() == 0


def test_cell_keys_morton_roundtrip(host_harness):
    rng = np.random.default_rng(4)
    xyz = np.concatenate([rng.uniform(-500, 500, (5000, 3)), [[0, 0, 0], [-0.0, 0.5, -0.5], [524287.9, -524288, 3]]]).astype(np.float32)
    n = len(xyz)
    packed = np.zeros(n, np.uint64); mort = np.zeros(n, np.uint64); hsh = np.zeros(n, np.uint32)
    host_harness.hh_keys(np.ascontiguousarray(xyz), n, 2.0, packed, mort, hsh)
    dec = np.zeros((n, 3), np.uint32)
    host_harness.hh_morton_decode(mort, n, dec)
    cells = np.floor(xyz.astype(np.float64) * 2.0).astype(np.int64) + (1 << 20)
    cells = np.clip(cells, 0, (1 << 21) - 1)
    assert np.array_equal(dec.astype(np.int64), cells)
    assert np.array_equal(packed, (cells[:, 0].astype(np.uint64) << np.uint64(42)) | (cells[:, 1].astype(np.uint64) << np.uint64(21)) | cells[:, 2].astype(np.uint64))
    # Morton order == octree order: sorting by morton groups every 2x2x2 block contiguously
    order = np.argsort(mort, kind="stable")
    parents = (mort[order] >> np.uint64(3))
    assert (np.diff(parents.astype(np.int64)) >= 0).all()


def test_host_filter_matches_oracle(oracle, host_harness, small_case):
    """lio::esekf::update_iterated_dyn_share_modified driven by the oracle's h_share_model through a
    callback == the oracle's own iterated update (same m, same neighbours) to 1e-9."""
    from better_fastlio2_amd import capi, synth
    w, sc = small_case["world"], small_case["scan"]
    om = oracle.OracleMap(0.2); om.build(w.points)
    for ext, nq in ((False, 20000), (True, 4000), (False, 15)):      # 15 rows: dense-gain branch (m < 23)
        body = np.ascontiguousarray(sc.body[:nq])
        x0 = capi.make_state26(small_case["pos"], small_case["rot"], sc.off_R, sc.off_T)
        P0 = synth.init_cov()
        xo, Po, arrs, log = oracle.update_iterated(om, x0, P0, body, 0.001, 3, None, ext)
        arr = oracle.ScanArrays(nq)
        CB = C.CFUNCTYPE(C.c_int, C.POINTER(C.c_double), C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_double),
                         C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_int)

        def cb(pose, converge, m_out, rm_out, HTH, HTh, hx_out, h_out, cap):
            st = np.array([pose[i] for i in range(14)])
            valid, hx, h, m, rm = oracle.h_share_model(om, st, body, arr, bool(converge), ext)
            m_out[0] = m; rm_out[0] = rm
            A, b = oracle.hth(hx, h) if m else (np.zeros((12, 12)), np.zeros(12))
            for i in range(144): HTH[i] = A.flat[i]
            for i in range(12): HTh[i] = b[i]
            if m < 23:
                for i in range(m * 12): hx_out[i] = hx.flat[i]
                for i in range(m): h_out[i] = h[i]
            return int(valid)

        x = x0.copy(); P = P0.copy(); nc = C.c_int(0)
        host_harness.hh_eskf_update.argtypes = [np.ctypeslib.ndpointer(np.float64), np.ctypeslib.ndpointer(np.float64),
                                                C.c_double, C.c_int, np.ctypeslib.ndpointer(np.float64), CB, C.c_int,
                                                C.POINTER(C.c_int)]
        host_harness.hh_eskf_update(x, P, 0.001, 3, np.full(23, 0.001), CB(cb), 64, C.byref(nc))
        assert nc.value == len(log)
        # the two 23x23 inversions per pass use Gauss-Jordan here and LU in the oracle (both partial
        # pivoting): agreement is at the rounding level of cond(P/R) ~ 1e5, far inside 1e-4 m / 1e-5 rad
        assert np.abs(x - xo).max() < 1e-9
        assert np.abs(P - Po).max() < 1e-9 * max(1.0, np.abs(Po).max())


def test_boxplus_boxminus_match_oracle(oracle, host_harness):
    rng = np.random.default_rng(6)
    from better_fastlio2_amd import capi
    for _ in range(100):
        q = rng.normal(size=4); q /= np.linalg.norm(q)
        q2 = rng.normal(size=4); q2 /= np.linalg.norm(q2)
        x = capi.make_state26(rng.normal(size=3), q, q2, rng.normal(size=3), grav=(-9.7, 1.2, 0.8))
        x[23:26] *= 9.809 / np.linalg.norm(x[23:26])
        dx = rng.normal(size=23) * 0.1
        a = oracle.state_boxplus(x, dx)
        b = x.copy(); host_harness.hh_boxplus(b, dx)
        assert np.array_equal(a, b)
        r1 = oracle.state_boxminus(a, x)
        r2 = np.zeros(23); host_harness.hh_boxminus(a, x, r2)
        assert np.array_equal(r1, r2)


def test_hth_from_callback_rows(host_harness):
    """The filter side of the zero-patch callback path: h_x^T h_x and h_x^T h from m x 12 column-major rows
    (esekfom.hpp:1790,1810), blocked 4-lane partial sums -- against numpy, at sizes around the block and vector edges."""
    rng = np.random.default_rng(4)
    for m in (1, 7, 8, 127, 128, 129, 1000, 90000):
        hx = np.asfortranarray(rng.normal(0, 1, (m, 12)))
        h = rng.normal(0, 0.05, m)
        HTH = np.zeros(144); HTh = np.zeros(12)
        secs = host_harness.hh_hth_rows(np.ascontiguousarray(hx.T).reshape(-1), h, m, HTH, HTh, 3 if m > 10000 else 1)
        ref = hx.T @ hx
        assert np.allclose(HTH.reshape(12, 12), ref, rtol=1e-12, atol=1e-12 * max(m, 1))
        assert np.allclose(HTh, hx.T @ h, rtol=1e-12, atol=1e-12 * max(m, 1))
        assert np.array_equal(HTH.reshape(12, 12), HTH.reshape(12, 12).T)
    assert secs < 0.01                                          # 90 k rows: a fraction of a millisecond on one core, not 8 ms


def test_batch_key_layout_roundtrip_and_frame_independence(tmp_path):
    """csrc/lio_mapadd.h: the key of the map update's group sorts (a digit of middle coordinate bits on top) decodes back, keeps the
    all-ones key for the spare coordinates, and orders a batch the same way whatever box its frame was derived from."""
    import shutil
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not found")
    exe = str(tmp_path / "key_layout_check")
    subprocess.check_call([hipcc, "-x", "hip", "--offload-arch=gfx950", "-I", os.path.join(ROOT, "better_fastlio2_amd", "csrc"), "-I", os.path.join(ROOT, "include"),
                           "-o", exe, os.path.join(ROOT, "tests", "csrc", "key_layout_check.cpp")])
    assert subprocess.check_output([exe]).decode().strip() == "ok"


def test_cell_key_frame_keeps_the_morton_order(tmp_path):
    """csrc/lio_rebuild.h: the compressed cell keys of the rebuild by cell runs ([rank of the coarse parent | low Morton bits]) sort like
    the absolute Morton codes, expand back to them, fit the bits the frame declares and refuse cells outside the frame."""
    import shutil
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not found")
    exe = str(tmp_path / "cell_frame_check")
    subprocess.check_call([hipcc, "-x", "hip", "--offload-arch=gfx950", "-I", os.path.join(ROOT, "better_fastlio2_amd", "csrc"), "-I", os.path.join(ROOT, "include"),
                           "-o", exe, os.path.join(ROOT, "tests", "csrc", "cell_frame_check.cpp")])
    assert subprocess.check_output([exe]).decode().strip() == "ok 400 frames"
