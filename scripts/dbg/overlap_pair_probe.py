"""Upper bound of overlapping a search pass's measurement kernel with its k-NN kernel (lio_debug_overlap_pair_probe, lio_ctx.hip):
k-NN alone / measurement kernel alone / both on one stream / both at once on two streams, host microseconds per iteration.
    python scripts/dbg/overlap_pair_probe.py [c2|c3] [iters]"""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import bench
from better_fastlio2_amd import capi, synth
cfg = sys.argv[1] if len(sys.argv) > 1 else "c2"
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 200
nm, nq, voxel, kind, max_iter = bench.CONFIGS[cfg]
world = synth.make_world(nm, voxel)
scan = synth.make_scan(world, nq, kind=kind)
pos, rot = synth.perturb_state(scan)
st = np.ascontiguousarray(synth.state_vector(pos, rot, scan.off_R, scan.off_T)[:14], np.float64)
ctx = capi.LioContext(voxel, max_scan_points=nq, max_map_points=nm + 1000)
ctx.map_build(world.points); ctx.scan_set(scan.body)
g = ctx.measure(st, True, scan.extrinsic_est_en)
out = np.zeros(4, np.float64)
f = ctx.L.lio_debug_overlap_pair_probe
f.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]; f.restype = C.c_int
for rep in range(3):
    rc = f(ctx.h, st.ctypes.data, int(scan.extrinsic_est_en), iters, out.ctypes.data)
    assert rc == 0, ctx.L.lio_last_error(ctx.h)
    print("%s rep %d: k-NN alone %.1f us | measurement kernel (no walk) alone %.1f | one stream %.1f | two streams at once %.1f  (launch + wait included in each)" % (cfg, rep, *out), flush=True)
g2 = ctx.measure(st, True, scan.extrinsic_est_en)
assert g2["m"] == g["m"] and np.array_equal(g2["HTH"], g["HTH"]), "the context did not come back clean"
print("context clean after the probe: m = %d" % g2["m"])
