"""Runs the C2 search pass a few times (for rocprofv3 --pmc / --kernel-trace captures)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from better_fastlio2_amd import capi, synth
nm, nq = 5_000_000, 100_000
world = synth.make_world(nm, 0.2); scan = synth.make_scan(world, nq)
pos, rot = synth.perturb_state(scan); st = synth.state_vector(pos, rot, scan.off_R, scan.off_T)
ctx = capi.LioContext(0.2, max_scan_points=nq, max_map_points=nm + 1000, lanes_per_query=int(sys.argv[1]) if len(sys.argv) > 1 else 0)
ctx.map_build(world.points); ctx.scan_set(scan.body)
for _ in range(int(sys.argv[2]) if len(sys.argv) > 2 else 10):
    g = ctx.measure(st, True, False)
print("m", g["m"], "unresolved", g["n_unresolved"])
