"""Differential fuzzing of the assisted pull (the host packs the front of a page-locked cloud while the scan's first search pulls the
back; optional second packing thread): clouds of random sizes (below and above the sizes that get help) and record strides, the
share steered or fixed, one or two packing threads switched at random, a host that stalls now and then -- against lio_scan_set on
a twin context: the scan buffer, the neighbours and the sums of every scan's first pass must be equal to the bit.

    python tests/fuzz/fuzz_assist.py [n_scans] [seed]
"""
import os, sys, ctypes as C
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from better_fastlio2_amd import capi, synth


def bits_equal(a, b):
    a = np.ascontiguousarray(a); b = np.ascontiguousarray(b)
    return a.shape == b.shape and a.dtype == b.dtype and np.array_equal(a.view(np.uint8), b.view(np.uint8))


def main(n_scans, seed):
    rng = np.random.default_rng(seed)
    w = synth.make_world(300_000, 0.2, seed=seed)
    cap = 131072
    big = synth.make_scan(w, cap, seed=seed + 1)
    ps, rt = synth.perturb_state(big, seed=seed + 2)
    st = capi.make_state26(ps, rt, big.off_R, big.off_T)[:14]
    ctx = capi.LioContext(0.2, max_scan_points=cap, max_map_points=400_000)
    twin = capi.LioContext(0.2, max_scan_points=cap, max_map_points=400_000)
    if not ctx.debug_get("gate_mail_in_vram"):
        print("FUZZ ASSIST SKIPPED: no large BAR (assisted pull stays off)"); return
    ctx.map_build(w.points); twin.map_build(w.points)
    L = ctx.L
    bufs = {}
    for sf in (8, 12):
        b = np.full((cap, sf), 3.0, np.float32)
        assert L.lio_host_register(b.ctypes.data_as(C.c_void_p), b.nbytes) == 0
        bufs[sf] = b
    assisted = late = 0
    threads = 1
    for k in range(n_scans):
        n = int(rng.choice([9000, 20000, 33000, 50000, 64000, 80000, 100000, 120000, cap]))
        sf = int(rng.choice([8, 12]))
        idx = rng.permutation(cap)[:n]
        body = np.ascontiguousarray(big.body[idx] + rng.normal(0, 0.01, (n, 3)).astype(np.float32))
        if rng.random() < 0.1:
            threads = 3 - threads
            ctx.scan_pack_threads(threads)
        ctx.debug_set("assist_pull", int(rng.choice([1, 1, 1, 0, 30, 60, 92])))
        ctx.debug_set("assist_pull_stall_us", int(rng.choice([0] * 12 + [150, 700])))
        buf = bufs[sf]
        buf[:n, :3] = body
        a0 = ctx.debug_get("assisted_pulls")
        assert L.lio_scan_set_async(ctx.h, buf.ctypes.data_as(C.c_void_p), n, 4 * sf) == 0, L.lio_last_error(ctx.h)
        g = ctx.measure(st, True, False)
        assisted += ctx.debug_get("assisted_pulls") - a0
        twin.scan_set(body)
        g0 = twin.measure(st, True, False)
        nx, nd, nc = ctx.neighbors(); nx0, nd0, nc0 = twin.neighbors()
        ok = (g["m"] == g0["m"] and bits_equal(g["HTH"], g0["HTH"]) and bits_equal(g["HTh"], g0["HTh"]) and bits_equal(nx, nx0)
              and bits_equal(nd, nd0) and bits_equal(nc, nc0) and bits_equal(ctx.scan_cloud(1), twin.scan_cloud(1)))
        if not ok:
            raise AssertionError(f"seed {seed} scan {k}: n {n}, stride {4 * sf}, threads {threads}: the assisted pull changed a result")
        if k % 7 == 0:
            g = ctx.measure(st, False, False); g0 = twin.measure(st, False, False)
            assert g["m"] == g0["m"] and bits_equal(g["HTH"], g0["HTH"])
    late = ctx.debug_get("pack_helper_late")
    ctx.debug_set("assist_pull_stall_us", 0); ctx.synchronize()
    for b in bufs.values():
        assert L.lio_host_unregister(b.ctypes.data_as(C.c_void_p)) == 0
    ctx.close(); twin.close()
    print("FUZZ ASSIST OK: %d scans, %d with the host's help, the second thread late on %d" % (n_scans, assisted, late))


if __name__ == "__main__":
    main(int(sys.argv[1]) if len(sys.argv) > 1 else 300, int(sys.argv[2]) if len(sys.argv) > 2 else 71000)
