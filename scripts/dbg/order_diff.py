"""Debug: library_sort vs default path of tests/test_gpu_compact.py -- where do the flattened maps differ?"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import numpy as np
from better_fastlio2_amd import capi, synth
import test_gpu_compact as T
w = synth.make_world(200_000, 0.2)
n_raw = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
b, fb = T._drive(capi, synth, w, n_raw, "library_sort", 3)
a, fa = T._drive(capi, synth, w, n_raw, None, 3)
print("per-scan tuples equal:", [x[0] == y[0] and x[3:] == y[3:] for x, y in zip(a, b)])
print([x[3:] for x in a])
same_set = np.array_equal(np.sort(fa.view(np.uint32).reshape(-1, 3).astype(np.uint64) @ np.array([1, 1 << 21, 1 << 42], np.uint64)),
                          np.sort(fb.view(np.uint32).reshape(-1, 3).astype(np.uint64) @ np.array([1, 1 << 21, 1 << 42], np.uint64)))
print("same set (hash):", same_set, fa.shape, fb.shape)
d = np.nonzero((fa.view(np.uint32) != fb.view(np.uint32)).any(1))[0]
print("differing rows:", len(d), "first", d[:10], "last", d[-5:] if len(d) else None)
if len(d):
    i = d[0]
    print("custom :", fa[i - 2:i + 6])
    print("library:", fb[i - 2:i + 6])
    ka = np.floor(fa[i:i + 6] / 0.5).astype(int); kb = np.floor(fb[i:i + 6] / 0.5).astype(int)
    print("cells custom :", ka.tolist()); print("cells library:", kb.tolist())
