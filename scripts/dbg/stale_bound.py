import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from better_fastlio2_amd import capi, synth
from oracle import pyoracle as oracle
world = synth.make_world(200_000, 0.2); sc = synth.make_scan(world, 20_000)
pos, rot = synth.perturb_state(sc); st = synth.state_vector(pos, rot, sc.off_R, sc.off_T)
for separate in (1, 0):
    ctx = capi.LioContext(0.2, max_scan_points=20_000, max_map_points=400_000)
    ctx.debug_set("separate_tree_walk", separate)
    ctx.map_build(world.points)
    om = oracle.OracleMap(0.2); om.build(world.points)
    ctx.scan_set(sc.body)
    arr = oracle.ScanArrays(len(sc.body))
    oracle.h_share_model(om, st, sc.body, arr, True, False)
    ctx.measure(st, True, False)
    box = np.array([[-6.0, -8.0, -1.0, 9.0, 4.0, 3.0]], np.float32)
    print("deleted", ctx.map_delete_boxes(box), om.delete_boxes(box))
    valid, hx, h, m, rm = oracle.h_share_model(om, st, sc.body, arr, True, False)
    g = ctx.measure(st, True, False)
    nx, nd, nc = ctx.neighbors()
    bad_c = np.where(nc != arr.nbr_cnt)[0]
    bad_d = np.where((nd.view(np.uint32) != arr.nbr_d.view(np.uint32)).any(1))[0]
    print("separate", separate, "m", g["m"], m, "unres", g["n_unresolved"], "cnt mismatches", len(bad_c), "dist mismatches", len(bad_d))
    if len(bad_d):
        i = bad_d[0]
        print(" first bad", i, nc[i], arr.nbr_cnt[i], nd[i], arr.nbr_d[i])
        print(" bad idx range", bad_d[:10], bad_d[-5:], "blocks", np.unique(bad_d // 256)[:20])
    ctx.close()
