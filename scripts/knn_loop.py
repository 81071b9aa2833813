"""Runs search passes (first search, then a bounded re-search at a moved pose, per round) of one BASELINE config for rocprofv3
captures:  python scripts/knn_loop.py [lanes] [rounds] [c2|c3|c5]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from better_fastlio2_amd import capi, synth
import bench
cfg = sys.argv[3] if len(sys.argv) > 3 else "c2"
nm, nq, voxel, kind, _ = bench.CONFIGS[cfg]
world = synth.make_world(nm, voxel); scan = synth.make_scan(world, nq, kind=kind)
pos, rot = synth.perturb_state(scan); st = synth.state_vector(pos, rot, scan.off_R, scan.off_T)
st2 = st.copy(); st2[:3] += [0.03, -0.05, 0.01]
ctx = capi.LioContext(voxel, max_scan_points=nq, max_map_points=nm + 1000, lanes_per_query=int(sys.argv[1]) if len(sys.argv) > 1 else 0)
ctx.map_build(world.points)
for _ in range(int(sys.argv[2]) if len(sys.argv) > 2 else 10):
    ctx.scan_set(scan.body)
    g = ctx.measure(st, True, scan.extrinsic_est_en)
    g = ctx.measure(st2, True, scan.extrinsic_est_en)
print("m", g["m"], "unresolved", g["n_unresolved"])
