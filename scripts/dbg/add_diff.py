"""Debug: one down-sampled Add_Points against the oracle -- which points are extra / missing?"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from better_fastlio2_amd import capi, synth
from oracle import pyoracle as po
w = synth.make_world(200_000, 0.2)
base = np.ascontiguousarray(w.points[:80000])
rng = np.random.default_rng(21)
new = np.ascontiguousarray(base[rng.integers(0, len(base), 5000)] + rng.normal(0, 0.04, (5000, 3)).astype(np.float32))
ctx = capi.LioContext(0.2, max_scan_points=1024, max_map_points=400_000)
ctx.map_build(base)
om = po.OracleMap(0.2); om.build(base)
cg, co = ctx.map_add(new, True), om.add(new, True)
print("counter gpu/oracle", cg, co, "validnum", ctx.map_validnum(), om.validnum())
fg, fo = ctx.map_flatten(), om.flatten()
key = lambda p: set(map(bytes, np.ascontiguousarray(p).view(np.uint8).reshape(len(p), 12)))
sg, so = key(fg), key(fo)
extra, missing = sg - so, so - sg
print("flat sizes", len(fg), len(fo), "extra", len(extra), "missing", len(missing))
sb, sn = key(base), key(new)
print("extra that are base points:", len(extra & sb), " extra that are new points:", len(extra & sn), " other:", len(extra - sb - sn))
print("missing that are base points:", len(missing & sb), " missing that are new points:", len(missing & sn))
ex = np.array([np.frombuffer(b, np.float32) for b in list(extra - sb - sn)[:5]])
print("sample other extras:", ex)
print(ctx.map_info())
