"""Generates tests/golden/*.npz.  Run once in the build container, where /root/reference exists:

    python tests/golden/make_golden.py

knn_ref.npz   : outputs of the REFERENCE's own ikd-Tree (include/ikd-Tree/ikd_Tree.cpp compiled
                verbatim into oracle/_ref/libikd_ref.so) -- Build, Nearest_Search, Add_Points,
                Delete_Point_Boxes on a seeded 30 k-point world.  These are the reference's bits.
step_oracle.npz : the oracle's full scan step (h_share_model passes + iterated update) on the same
                world; k-NN inside it went through the reference tree (knn_method 2).  The model part
                is "parity unpinned" (needs Eigen) -- this file freezes the
                restatement so that any drift of oracle or GPU path is caught.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from better_fastlio2_amd import capi, synth  # noqa: E402
from oracle import pyoracle as po  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def main():
    assert po.ref_available(), "needs oracle/_ref/libikd_ref.so (i.e. /root/reference)"
    world = synth.make_world(30_000, 0.2, seed=synth.SEED + 100)
    scan = synth.make_scan(world, 3_000, seed=synth.SEED + 101, pos=(1.0, -0.5, 0.0), rpy=(0.0, 0.01, 0.3))
    pos, rot = synth.perturb_state(scan, seed=synth.SEED + 102)
    st = synth.state_vector(pos, rot, scan.off_R, scan.off_T)
    rng = np.random.default_rng(77)
    wq = po.body_to_world(st, scan.body)
    q = np.ascontiguousarray(np.concatenate([
        wq[:600],
        (rng.uniform(-1.4, 1.4, (200, 3)) * [world.ground_half, world.ground_half, 1.0]).astype(np.float32),
        rng.uniform(-80, 80, (60, 3)).astype(np.float32)]))
    rt = po.RefIkdTree(0.2)
    rt.build(world.points)
    x0, d0, c0, _ = rt.knn(q)
    new = np.ascontiguousarray(wq[600:2600] + rng.normal(0, 0.03, (2000, 3)).astype(np.float32))
    n_add = rt.add(new, True)
    plain = np.ascontiguousarray(rng.uniform(-8, 8, (100, 3)).astype(np.float32))
    n_add_plain = rt.add(plain, False)
    boxes = np.array([[-6, -6, -1, 2, 1, 3], [1, 1, -1, 7, 6, 2]], np.float32)
    n_del = rt.delete_boxes(boxes)
    rt.quiesce(300)
    live = rt.flatten()
    live = live[np.lexsort((live[:, 2], live[:, 1], live[:, 0]))]
    x1, d1, c1, _ = rt.knn(q)
    np.savez_compressed(os.path.join(HERE, "knn_ref.npz"), map=world.points, queries=q, nbr_xyz=x0, nbr_d=d0, nbr_cnt=c0,
                        add_pts=new, n_add=n_add, add_plain=plain, n_add_plain=n_add_plain, boxes=boxes, n_del=n_del,
                        live_sorted=live, nbr_xyz_after=x1, nbr_d_after=d1, nbr_cnt_after=c1)

    rt2 = po.RefIkdTree(0.2)
    rt2.build(world.points)
    om = po.OracleMap(0.2)
    po.use_reference_knn(rt2)
    out = {}
    for ext in (False, True):
        x26 = capi.make_state26(pos, rot, scan.off_R, scan.off_T)
        arr = po.ScanArrays(len(scan.body))
        valid, hx, h, m, rm = po.h_share_model(om, st, scan.body, arr, True, ext, knn_method=2)
        HTH, HTh = po.hth(hx, h)
        x1, P1, arrs, log = po.update_iterated(om, x26, synth.init_cov(), scan.body, 0.001, 3, None, ext, knn_method=2)
        k = "ext" if ext else "noext"
        out.update({f"{k}_m": m, f"{k}_res_mean": rm, f"{k}_HTH": HTH, f"{k}_HTh": HTh, f"{k}_sel": arr.selected.copy(),
                    f"{k}_normvec": arr.normvec.copy(), f"{k}_world": arr.world.copy(), f"{k}_hx_head": hx[:64].copy(),
                    f"{k}_h_head": h[:64].copy(), f"{k}_x_post": x1, f"{k}_P_post": P1, f"{k}_log": log,
                    f"{k}_last_nbr_xyz": arrs.nbr_xyz.copy(), f"{k}_last_nbr_cnt": arrs.nbr_cnt.copy()})
    po.use_reference_knn(None)
    np.savez_compressed(os.path.join(HERE, "step_oracle.npz"), map=world.points, body=scan.body, state14=st,
                        x0=capi.make_state26(pos, rot, scan.off_R, scan.off_T), P0=synth.init_cov(), true_pos=scan.pos,
                        true_rot=scan.rot, **out)
    for f in ("knn_ref.npz", "step_oracle.npz"):
        print(f, os.path.getsize(os.path.join(HERE, f)) // 1024, "KiB")


if __name__ == "__main__":
    main()
