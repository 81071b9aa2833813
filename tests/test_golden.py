"""Golden vectors (tests/golden/, made by tests/golden/make_golden.py from the reference's own
ikd-Tree): the oracle must reproduce them on any host, the GPU path on the MI355X."""
import os

import numpy as np
import pytest

from conftest import bits_equal

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(scope="module")
def knn_ref():
    return np.load(os.path.join(G, "knn_ref.npz"))


@pytest.fixture(scope="module")
def step():
    return np.load(os.path.join(G, "step_oracle.npz"))


def _sorted(p):
    return p[np.lexsort((p[:, 2], p[:, 1], p[:, 0]))]


# ------------------------------------------------------------------ oracle (CPU) vs the reference's bits
def test_oracle_knn_reproduces_reference_golden(oracle, knn_ref):
    om = oracle.OracleMap(0.2); om.build(knn_ref["map"])
    for method in (0, 1):
        x, d, c, _ = om.knn(knn_ref["queries"], 5, method)
        assert np.array_equal(c, knn_ref["nbr_cnt"])
        assert bits_equal(d, knn_ref["nbr_d"]) and bits_equal(x, knn_ref["nbr_xyz"])
    assert om.add(knn_ref["add_pts"], True) == int(knn_ref["n_add"])
    assert om.add(knn_ref["add_plain"], False) == int(knn_ref["n_add_plain"])
    assert om.delete_boxes(knn_ref["boxes"]) == int(knn_ref["n_del"])
    assert bits_equal(_sorted(om.flatten()), knn_ref["live_sorted"])
    x, d, c, _ = om.knn(knn_ref["queries"])
    assert np.array_equal(c, knn_ref["nbr_cnt_after"])
    assert bits_equal(d, knn_ref["nbr_d_after"]) and bits_equal(x, knn_ref["nbr_xyz_after"])


def test_oracle_step_reproduces_golden(oracle, step):
    om = oracle.OracleMap(0.2); om.build(step["map"])
    for ext, k in ((False, "noext"), (True, "ext")):
        arr = oracle.ScanArrays(len(step["body"]))
        valid, hx, h, m, rm = oracle.h_share_model(om, step["state14"], step["body"], arr, True, ext)
        assert m == int(step[f"{k}_m"]) and np.array_equal(arr.selected, step[f"{k}_sel"])
        assert np.array_equal(hx[:64], step[f"{k}_hx_head"]) and np.array_equal(h[:64], step[f"{k}_h_head"])
        x1, P1, arrs, log = oracle.update_iterated(om, step["x0"], step["P0"], step["body"], 0.001, 3, None, ext)
        assert np.array_equal(x1, step[f"{k}_x_post"]) and np.array_equal(P1, step[f"{k}_P_post"])


# ------------------------------------------------------------------ GPU path vs the same files
@pytest.mark.gpu
def test_gpu_knn_reproduces_reference_golden(gpu_capi, knn_ref):
    for lanes in (4, 8, 16, 32, 64):
        ctx = gpu_capi.LioContext(0.2, max_scan_points=4096, max_map_points=100_000, lanes_per_query=lanes)
        ctx.map_build(knn_ref["map"])
        x, d, c = ctx.nearest_search(knn_ref["queries"])
        assert np.array_equal(c, knn_ref["nbr_cnt"]), lanes
        assert bits_equal(d, knn_ref["nbr_d"]) and bits_equal(x, knn_ref["nbr_xyz"]), lanes
        ctx.close()
    ctx = gpu_capi.LioContext(0.2, max_scan_points=4096, max_map_points=100_000)
    ctx.map_build(knn_ref["map"])
    assert ctx.map_add(knn_ref["add_pts"], True) == int(knn_ref["n_add"])
    assert ctx.map_add(knn_ref["add_plain"], False) == int(knn_ref["n_add_plain"])
    assert ctx.map_delete_boxes(knn_ref["boxes"]) == int(knn_ref["n_del"])
    assert ctx.map_validnum() == len(knn_ref["live_sorted"])
    assert bits_equal(_sorted(ctx.map_flatten()), knn_ref["live_sorted"])
    x, d, c = ctx.nearest_search(knn_ref["queries"])
    assert np.array_equal(c, knn_ref["nbr_cnt_after"])
    assert bits_equal(d, knn_ref["nbr_d_after"]) and bits_equal(x, knn_ref["nbr_xyz_after"])


@pytest.mark.gpu
def test_gpu_step_reproduces_golden(gpu_capi, step):
    ctx = gpu_capi.LioContext(0.2, max_scan_points=4096, max_map_points=100_000)
    ctx.map_build(step["map"])
    for ext, k in ((False, "noext"), (True, "ext")):
        ctx.scan_set(step["body"])
        g = ctx.measure(step["state14"], True, ext)
        assert g["m"] == int(step[f"{k}_m"])
        world, sel, nv = ctx.scan_state()
        assert np.array_equal(sel, step[f"{k}_sel"]) and bits_equal(world, step[f"{k}_world"])
        s = sel.astype(bool)
        assert bits_equal(nv[s], step[f"{k}_normvec"][s])
        assert np.allclose(g["HTH"], step[f"{k}_HTH"], rtol=1e-9, atol=1e-9 * np.abs(step[f"{k}_HTH"]).max())
        assert np.allclose(g["HTh"], step[f"{k}_HTh"], rtol=1e-9, atol=1e-9 * np.abs(step[f"{k}_HTh"]).max())
        assert abs(g["res_mean"] - float(step[f"{k}_res_mean"])) < 1e-12
        hx, h = ctx.measure_rows(g["m"])
        assert np.array_equal(hx[:64], step[f"{k}_hx_head"]) and np.array_equal(h[:64], step[f"{k}_h_head"])
        kf = gpu_capi.Esekf(3)
        kf.change_x(step["x0"]); kf.change_P(step["P0"])
        ctx.scan_set(step["body"])
        lg = kf.update_iterated_dyn_share_modified(ctx, 0.001, ext)
        x1 = kf.get_x()
        assert lg["n_calls"] == len(step[f"{k}_log"])
        # north_star tolerance: pose within 1e-4 m / 1e-5 rad of the CPU path
        assert np.abs(x1[:3] - step[f"{k}_x_post"][:3]).max() < 1e-4
        assert np.abs(x1[3:11] - step[f"{k}_x_post"][3:11]).max() < 0.5e-5     # |dq| = angle / 2
        nx, nd, nc = ctx.neighbors()
        assert np.array_equal(nc, step[f"{k}_last_nbr_cnt"]) and bits_equal(nx, step[f"{k}_last_nbr_xyz"])
