#!/usr/bin/env python3
"""One scan step on a KITTI velodyne .bin sweep (BASELINE.json configs[0]; include/common_lib.h:594-614 readBin):

    python scripts/kitti_bin_step.py SCAN.bin [--map MAP.bin ...] [--leaf 0.5] [--voxel 0.2]

SCAN.bin is read, handed to lio_scan_undistort without IMU poses (a .bin sweep carries no per-point time), down-sampled
with lio_scan_voxel_filter(leaf) and registered against a map built from the --map sweeps (voxel-filtered at `voxel`),
or, without any, against the synthetic world of the bench.  Prints the posterior, the number of effective points, the
kernel times and -- when the reference ikd-Tree has been built (oracle/_ref) -- the neighbour parity of the scan and the
number of rank-1..6 distance ties.  Needs an MI355X.
"""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("scan", nargs="?", default=None)
    ap.add_argument("--synthetic", action="store_true",
                    help="no file at hand: write a synthetic 120 k-point KITTI-64-like sweep of the bench's world as a .bin first, then run on it")
    ap.add_argument("--map", nargs="*", default=[])
    ap.add_argument("--leaf", type=float, default=0.5)
    ap.add_argument("--voxel", type=float, default=0.2)
    ap.add_argument("--check", action="store_true", help="compare the neighbours with the reference ikd-Tree (oracle/_ref)")
    a = ap.parse_args()
    from better_fastlio2_amd import capi, kitti, synth
    truth = None
    if a.synthetic:
        import tempfile
        world = synth.make_world(2_000_000, a.voxel)
        sc = synth.make_scan(world, 120_000)
        a.scan = os.path.join(tempfile.mkdtemp(), "000000.bin")
        kitti.write_bin(a.scan, sc.body, np.linspace(0.0, 1.0, len(sc.body), dtype=np.float32))
        truth = sc
    if a.scan is None or not os.path.exists(a.scan):
        raise SystemExit(f"no such file: {a.scan} (no KITTI data ships with this repository; --synthetic writes a stand-in)")
    sweep = kitti.read_bin(a.scan)
    if a.map:
        pts = np.concatenate([kitti.read_bin(p)[:, :3] for p in a.map], 0)
        key = np.unique(np.floor(pts / a.voxel).astype(np.int64), axis=0, return_index=True)[1]      # one point per voxel
        map_pts = np.ascontiguousarray(pts[np.sort(key)])
        prior_pos, prior_rot = np.zeros(3), np.array([0.0, 0.0, 0.0, 1.0])
    else:
        if truth is None:
            world = synth.make_world(2_000_000, a.voxel)
        map_pts = world.points
        prior_pos, prior_rot = np.array([3.1, -2.2, 0.0]), np.array([0.0, 0.0, 0.257, 0.966])
        if truth is not None:
            prior_pos, prior_rot = synth.perturb_state(truth)
    ctx = capi.LioContext(a.voxel, max_scan_points=max(len(sweep), 1024), max_map_points=len(map_pts) + 1_000_000)
    ctx.map_build(map_pts)
    raw = np.ascontiguousarray(np.concatenate([sweep[:, :3], np.zeros((len(sweep), 1), np.float32)], 1))
    ctx.scan_undistort(raw)                                       # no poses: the sweep is taken as it is
    n_down = ctx.scan_voxel_filter(a.leaf)
    # LiDAR -> IMU extrinsics of config/velodyne64_kitti_dataset.yaml (the synthetic sweeps use the same)
    x0 = capi.make_state26(prior_pos, prior_rot, synth._quat_from_rot(synth.KITTI_EXTRINSIC_R), synth.KITTI_EXTRINSIC_T)
    check = None
    if a.check:
        # neighbours and world points of ONE search pass at the prior pose (the update below searches again elsewhere)
        ctx.measure(x0[:14], True, False)
        check = (ctx.neighbors(), ctx.scan_state()[0])
    kf = capi.Esekf(3)
    kf.change_x(x0); kf.change_P(synth.init_cov())
    ctx.profile_enable(True)
    log = kf.update_iterated_dyn_share_modified(ctx, 0.001, False)
    out = {"scan_points": int(len(sweep)), "feats_down_size": int(n_down), "map_points": int(len(map_pts)), "passes": log["n_calls"],
           "effct_feat_num": log["last_m"], "posterior_pos": kf.get_x()[:3].tolist(), "posterior_rot_xyzw": kf.get_x()[3:7].tolist(),
           "kernels_us": {k: v["total_ms"] / max(v["launches"], 1) * 1e3 for k, v in ctx.profile().items() if v["launches"]}}
    if a.check:
        from oracle import pyoracle as po
        if po.ref_available():
            (nx, nd, nc), wq = check
            tree = po.RefIkdTree(a.voxel); tree.build(map_pts)
            rx, rd, rc, _ = tree.knn(wq, 6)
            out["neighbors_vs_reference"] = {"sqdist_mismatches": int((nd.view(np.uint32) != rd[:, :5].view(np.uint32)).any(1).sum()),
                                             "count_mismatches": int((nc != np.minimum(rc, 5)).sum()),
                                             "rows_with_a_tie_in_ranks_1_6": int((np.diff(rd, axis=1) <= 0).any(1).sum())}
    if truth is not None:
        out["pos_err_to_truth_m"] = float(np.linalg.norm(kf.get_x()[:3] - truth.pos))
    print(json.dumps(out))


if __name__ == "__main__":
    main()
