"""Pins the oracle's map / k-NN restatement against the REFERENCE itself: the verbatim
include/ikd-Tree/ikd_Tree.cpp compiled into oracle/_ref/libikd_ref.so (oracle/Makefile)."""
import numpy as np
import pytest

from conftest import bits_equal


@pytest.fixture(scope="module")
def ref(oracle):
    if not oracle.ref_available():
        pytest.skip("oracle/_ref/libikd_ref.so not built (needs /root/reference once)")
    return oracle


def _queries(small_case, oracle, n=4000, seed=5):
    rng = np.random.default_rng(seed)
    sc = small_case["scan"]
    wq = oracle.body_to_world(small_case["state14"], sc.body)[:n]
    half = small_case["world"].ground_half
    extra = np.concatenate([
        (rng.uniform(-1.3 * half, 1.3 * half, (1500, 3)) * [1, 1, 0.03]).astype(np.float32),   # edge / outside
        rng.uniform(-200, 200, (300, 3)).astype(np.float32),                                    # far away
        np.array([[5e3, -7e3, 40.0]], np.float32)])
    return np.ascontiguousarray(np.concatenate([wq, extra]))


def test_knn_bit_exact_vs_reference_ikdtree(ref, small_case):
    w = small_case["world"]
    q = _queries(small_case, ref)
    om = ref.OracleMap(0.2); om.build(w.points)
    rt = ref.RefIkdTree(0.2); rt.build(w.points)
    rx, rd, rc, _ = rt.knn(q)
    for method in (0, 1):
        ox, od, oc, _ = om.knn(q if method else q[:400], 5, method)
        n = len(ox)
        assert np.array_equal(oc, rc[:n])
        assert bits_equal(od, rd[:n])
        assert bits_equal(ox, rx[:n])
    # the generators are tie-free at rank 5/6: every 5th distance is strictly below the 6th
    x6, d6, c6, _ = om.knn(q[:2000], 6, 1)
    assert (d6[:, 4] < d6[:, 5]).all()


def test_fewer_points_than_k(ref):
    pts = np.array([[0, 0, 0], [1, 0, 0], [0, 2, 0]], np.float32)
    q = np.array([[0.1, 0.1, 0.1], [10, 10, 10]], np.float32)
    om = ref.OracleMap(0.2); om.build(pts)
    rt = ref.RefIkdTree(0.2); rt.build(pts)
    ox, od, oc, _ = om.knn(q)
    rx, rd, rc, _ = rt.knn(q)
    assert np.array_equal(oc, rc) and (oc == 3).all()
    assert bits_equal(od[:, :3], rd[:, :3]) and bits_equal(ox[:, :3], rx[:, :3])


def test_add_points_and_delete_boxes_live_set(ref, small_case):
    """Add_Points (down-sampled and plain) and Delete_Point_Boxes: same return values, same live set,
    same neighbours afterwards."""
    w = small_case["world"]
    rng = np.random.default_rng(11)
    base = w.points[:60000]
    om = ref.OracleMap(0.2); om.build(base)
    rt = ref.RefIkdTree(0.2); rt.build(base)
    sc = small_case["scan"]
    wq = ref.body_to_world(small_case["state14"], sc.body)
    new = np.ascontiguousarray(wq[:8000] + rng.normal(0, 0.03, (8000, 3)).astype(np.float32))
    a1, a2 = om.add(new, True), rt.add(new, True)
    assert a1 == a2
    plain = np.ascontiguousarray(rng.uniform(-30, 30, (500, 3)).astype(np.float32))
    assert om.add(plain, False) == rt.add(plain, False)
    boxes = np.array([[-10, -10, -1, 5, 3, 2], [2, 2, -1, 12, 9, 4]], np.float32)
    assert om.delete_boxes(boxes) == rt.delete_boxes(boxes)
    rt.quiesce(300)
    assert om.validnum() == rt.validnum()
    fo, fr = om.flatten(), rt.flatten()
    key = lambda p: np.lexsort((p[:, 2], p[:, 1], p[:, 0]))
    assert bits_equal(fo[key(fo)], fr[key(fr)])
    q = np.ascontiguousarray(wq[8000:11000])
    ox, od, oc, _ = om.knn(q)
    rx, rd, rc, _ = rt.knn(q)
    assert np.array_equal(oc, rc) and bits_equal(od, rd) and bits_equal(ox, rx)


def test_full_step_on_reference_tree_equals_restated_tree(ref, small_case):
    """h_share_model + iterated update: k-NN through the reference's Nearest_Search (knn_method 2)
    gives the same posterior as through the restated tree."""
    from better_fastlio2_amd import capi, synth
    w, sc = small_case["world"], small_case["scan"]
    om = ref.OracleMap(0.2); om.build(w.points)
    rt = ref.RefIkdTree(0.2); rt.build(w.points)
    x0 = capi.make_state26(small_case["pos"], small_case["rot"], sc.off_R, sc.off_T)
    body = sc.body[:5000]
    a = ref.update_iterated(om, x0, synth.init_cov(), body, knn_method=1)
    ref.use_reference_knn(rt)
    try:
        b = ref.update_iterated(om, x0, synth.init_cov(), body, knn_method=2)
    finally:
        ref.use_reference_knn(None)
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
    assert bits_equal(a[2].nbr_xyz, b[2].nbr_xyz)
