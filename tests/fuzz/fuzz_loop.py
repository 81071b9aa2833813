"""Differential fuzzing of the node's per-scan loop (iterated update -> map_incremental -> Add_Points x2 ->
lasermap_fov_segment) at kilometre coordinates against the CPU path with the REFERENCE ikd-Tree as the map: random
offsets, thinning of the initial map, scan poses along a drive, lanes per query, extrinsic estimation on/off.  After every scan: posterior, the two
insert lists (bit-equal), Add_Points counter, validnum, fov boxes and delete counts.

    python tests/fuzz/fuzz_loop.py [n_cases] [first_seed]
"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from better_fastlio2_amd import capi, synth
from oracle import pyoracle as po


def bits_equal(a, b):
    return a.shape == b.shape and np.array_equal(np.ascontiguousarray(a).view(np.uint32), np.ascontiguousarray(b).view(np.uint32))


FAST = [0, 0]


def run_case(seed):
    rng = np.random.default_rng(seed)
    w = synth.make_world(int(rng.choice([120_000, 200_000])), 0.2, seed=seed)
    off = (rng.choice([-1, 1], 3) * rng.uniform([200, 200, 5], [4000, 4000, 60])).astype(np.float64)
    if seed % 3 == 0:
        off[:] = 0.0
    pts = (w.points.astype(np.float64) + off).astype(np.float32)
    ds = 0.2
    base = np.ascontiguousarray(pts[:: int(rng.integers(2, 5))])
    nq = int(rng.choice([3000, 8000, 12000]))
    lanes = int(rng.choice([0, 0, 0, 4, 8, 16, 32, 64]))
    ext = bool(rng.integers(0, 2))
    cell = float(rng.choice([0.0, 0.0, 0.0, 0.25, 0.5, 1.0, 2.0]))              # 0 = auto (0.5 m for the 0.2 m voxel)
    ctx = capi.LioContext(ds, max_scan_points=nq, max_map_points=600_000, lanes_per_query=lanes, cell_size=cell)
    ctx.map_build(base)
    # round 6: a TWIN context that calls map_incremental the way the node does (no lists wanted): from its second scan on the whole
    # map update runs without a host round trip (lio_update.hip: BatchDesc, launches sized by the previous lists; every case ends up
    # overflowing or not by its own inserts).  Same inputs => the same counts and, point for point in order, the same map.
    twin = capi.LioContext(ds, max_scan_points=nq, max_map_points=600_000, lanes_per_query=lanes, cell_size=cell)
    twin.map_build(base)
    ref = po.RefIkdTree(ds); ref.build(base); po.use_reference_knn(ref)
    om = po.OracleMap(ds)                                        # only carries parameters: k-NN goes to the reference tree
    fov = po.FovSegment()
    try:
        px, py, yaw = float(rng.uniform(-5, 5)), float(rng.uniform(-5, 5)), float(rng.uniform(-3, 3))
        for k in range(int(rng.integers(2, 5))):
            px += float(rng.uniform(0.3, 1.5)); py += float(rng.uniform(-0.6, 0.6)); yaw += float(rng.uniform(-0.1, 0.1))
            sc = synth.make_scan(w, nq, seed=seed * 31 + k, pos=(px, py, 0.0), rpy=(0.003, -0.004, yaw))
            pos, rot = synth.perturb_state(sc, dpos=0.04, drot_deg=0.25, seed=seed * 17 + k)
            x0 = capi.make_state26(np.asarray(pos) + off, rot, sc.off_R, sc.off_T)
            # lasermap_fov_segment first (laserMapping.cpp:2317), on both sides
            lid = np.asarray(sc.pos, np.float64) + off
            boxes_o = fov.step(lid, 60.0, 12.0)
            boxes_g, nd_g = ctx.fov_segment(lid, 60.0, 12.0)
            boxes_t, nd_t = twin.fov_segment(lid, 60.0, 12.0)
            assert bits_equal(boxes_t, boxes_g) and nd_t == nd_g, (seed, k, "twin: fov")
            assert boxes_g.shape == boxes_o.shape and bits_equal(boxes_g, boxes_o), (seed, k, "fov boxes")
            nd_o = ref.delete_boxes(boxes_o) if len(boxes_o) else 0
            assert nd_g == nd_o, (seed, k, "fov delete count", nd_g, nd_o)
            # CPU path
            xo, Po, arrs, log = po.update_iterated(om, x0, synth.init_cov(), sc.body, extrinsic_est_en=ext, knn_method=2)
            world_post = po.body_to_world(xo[:14], sc.body)
            cls = po.map_incremental_classify(world_post, arrs.nbr_xyz, arrs.nbr_cnt, ds, True)
            to_add, nodown = world_post[cls == 1], world_post[cls == 2]
            # GPU path
            kf = capi.Esekf(3); kf.change_x(x0); kf.change_P(synth.init_cov())
            ctx.scan_set(sc.body)
            kf.update_iterated_dyn_share_modified(ctx, 0.001, ext)
            # the bar of the path: 1e-4 m / 1e-5 rad.  Usually the two agree to 1e-12; when a point sits
            # exactly on a decision (s > 0.9, the 0.1 m plane test) the 1e-15 difference of the summed H^T H can flip it
            # in a later pass, which moves the posterior by ~1e-5 m
            dx = kf.get_x() - xo
            assert np.abs(dx[:3]).max() < 1e-4 and 2.0 * np.abs(dx[3:7]).max() < 1e-5 and np.abs(dx[7:]).max() < 1e-4, (seed, k, "posterior", dx)
            r = ctx.map_incremental(xo[:14], ds, True, apply=True, want_lists=True)
            assert r["n_add"] == len(to_add) and r["n_nodown"] == len(nodown), (seed, k, "list sizes")
            assert bits_equal(r["add"], to_add) and bits_equal(r["nodown"], nodown), (seed, k, "lists")
            c1 = ref.add(np.ascontiguousarray(to_add), True) if len(to_add) else 0
            if len(nodown): ref.add(np.ascontiguousarray(nodown), False)
            ref.quiesce(30)
            assert r["add_counter"] == c1, (seed, k, "Add_Points counter", r["add_counter"], c1)
            assert ctx.map_validnum() == ref.validnum(), (seed, k, "validnum", ctx.map_validnum(), ref.validnum())
            kt = capi.Esekf(3); kt.change_x(x0); kt.change_P(synth.init_cov())
            twin.scan_set(sc.body)
            kt.update_iterated_dyn_share_modified(twin, 0.001, ext)
            assert bits_equal(kt.get_x(), kf.get_x()), (seed, k, "twin: posterior")
            rt = twin.map_incremental(xo[:14], ds, True, apply=True, want_lists=False)
            assert (rt["n_add"], rt["n_nodown"], rt["add_counter"]) == (r["n_add"], r["n_nodown"], r["add_counter"]), (seed, k, "twin: counts", rt, r["n_add"], r["n_nodown"], r["add_counter"])
            assert twin.map_validnum() == ctx.map_validnum(), (seed, k, "twin: validnum")
            assert bits_equal(twin.map_flatten(), ctx.map_flatten()), (seed, k, "twin: the maps differ")
    finally:
        po.use_reference_knn(None)
        FAST[0] += twin.debug_get("fast_updates"); FAST[1] += twin.debug_get("fast_update_overflows")
        twin.close()
        ctx.close()


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    s0 = int(sys.argv[2]) if len(sys.argv) > 2 else 7000
    for seed in range(s0, s0 + n):
        run_case(seed)
    print("FUZZ LOOP OK: %d cases; the twin's map updates without a host round trip: %d (+ %d redone after a list outgrew its launches)" % (n, FAST[0], FAST[1]), flush=True)
