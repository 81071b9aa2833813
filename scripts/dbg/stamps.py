import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from better_fastlio2_amd import capi, synth
nm, nq = 5_000_000, 100_000
world = synth.make_world(nm, 0.2); scan = synth.make_scan(world, nq)
pos, rot = synth.perturb_state(scan); st = synth.state_vector(pos, rot, scan.off_R, scan.off_T)
ctx = capi.LioContext(0.2, max_scan_points=nq, max_map_points=nm + 1000)
ctx.map_build(world.points); ctx.scan_set(scan.body)
ctx.debug_set("stamps", 1)
st2 = st.copy(); st2[:3] += [0.01, -0.02, 0.005]
buf = np.zeros(8, np.uint64)
for search, pose in [(True, st), (False, st2), (True, st2), (False, st), (True, st), (True, st2), (False, st)] * 2:
    g = ctx.measure(pose, search, False)
    ctx.L.lio_debug_stamps(ctx.h, buf.ctypes.data)
    d = (buf[1:].astype(np.int64) - np.int64(buf[0])) / 100.0
    d = np.where(buf[1:] == 0, np.nan, d)
    print("search" if search else "reuse ", "unres", g["n_unresolved"], "us since last WG start: partial %.2f grp %.2f grid %.2f | listed terms folded %.2f (-) %.2f sums final %.2f | written %.2f" % tuple(d))
