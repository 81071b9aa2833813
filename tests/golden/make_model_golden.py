"""Generates tests/golden/model_ref.npz: outputs of the REFERENCE's own model half -- esti_plane
(include/common_lib.h:506-536), h_share_model (src/laserMapping.cpp:1876-2004) and
esekf::update_iterated_dyn_share_modified (include/IKFoM_toolkit/esekfom/esekfom.hpp:1620-1938) -- compiled against the
real Eigen by `make -C oracle ref_model EIGEN=...` (oracle/ref_model_capi.cpp).  Needs Eigen + Boost headers and the
reference checkout; none of the three exists in the build image, so there this script exits with a message and the
model half stays "parity unpinned".  Where they exist:

    make -C oracle ref_model EIGEN=/usr/include/eigen3
    python tests/golden/make_model_golden.py          # writes tests/golden/model_ref.npz (commit it)

tests/test_golden.py::test_*_reproduce_the_reference_model then compares the oracle restatement (any host) and the GPU
path (-m gpu) with these vectors.
"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from better_fastlio2_amd import capi, synth  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
SO = os.path.join(ROOT, "oracle", "_ref", "libref_model.so")


def plane_cases(rng, n=4000):
    """5-point neighbourhoods: noisy planes of every orientation at map-like coordinates (decided far from the 0.1 m
    threshold and right at it), near-collinear and near-coincident sets (rank decisions of the pivoted QR)."""
    out = []
    for i in range(n):
        c = rng.uniform(-200, 200, 3)
        nrm = rng.normal(size=3); nrm /= np.linalg.norm(nrm)
        u = np.cross(nrm, rng.normal(size=3)); u /= np.linalg.norm(u)
        v = np.cross(nrm, u)
        kind = i % 5
        spread = 0.3 if kind < 3 else 0.05
        pts = c + np.outer(rng.uniform(-spread, spread, 5), u) + np.outer(rng.uniform(-spread, spread, 5), v)
        noise = [0.005, 0.03, 0.07, 0.0, 0.0][kind]
        pts += np.outer(rng.normal(0, noise, 5), nrm)
        if kind == 3:
            pts = c + np.outer(rng.uniform(-0.3, 0.3, 5), u) + rng.normal(0, 1e-4, (5, 3))       # near-collinear
        if kind == 4:
            pts = c + rng.normal(0, 1e-3, (5, 3))                                                 # near-coincident
        out.append(pts)
    return np.ascontiguousarray(np.array(out, np.float32))


def main():
    if not os.path.exists(SO):
        print("oracle/_ref/libref_model.so is not built (needs Eigen + Boost headers + the reference checkout: "
              "`make -C oracle ref_model EIGEN=/path/to/eigen3`).  The model half stays 'parity unpinned'.")
        return 2
    L = C.CDLL(SO)
    f32 = np.ctypeslib.ndpointer(np.float32, flags="C"); f64 = np.ctypeslib.ndpointer(np.float64, flags="C")
    i32 = np.ctypeslib.ndpointer(np.int32, flags="C"); u8 = np.ctypeslib.ndpointer(np.uint8, flags="C")
    L.refm_esti_plane.argtypes = [f32, C.c_float, f32]; L.refm_esti_plane.restype = C.c_int
    L.refm_boxplus.argtypes = [f64, f64]; L.refm_boxminus.argtypes = [f64, f64, f64]
    L.refm_map_build.argtypes = [f32, C.c_int, C.c_float]; L.refm_scan_set.argtypes = [f32, C.c_int]
    L.refm_h_share_model.argtypes = [f64, C.c_int, C.c_int, f64, f64, C.POINTER(C.c_int), C.POINTER(C.c_double), f32, f32, u8, f32, i32]
    L.refm_h_share_model.restype = C.c_int
    L.refm_update_iterated.argtypes = [f64, f64, C.c_int, C.c_int, C.c_double]
    L.refm_eigen_version.restype = C.c_char_p
    rng = np.random.default_rng(20240926)

    planes_in = plane_cases(rng)
    planes_out = np.zeros((len(planes_in), 4), np.float32)
    planes_ok = np.zeros(len(planes_in), np.uint8)
    for i, p in enumerate(planes_in):
        planes_ok[i] = L.refm_esti_plane(np.ascontiguousarray(p), 0.1, planes_out[i])

    # manifold operations on random states / increments
    xs = np.zeros((64, 26)); dxs = rng.normal(0, 0.05, (64, 23)); xplus = np.zeros((64, 26)); xminus = np.zeros((64, 23))
    for i in range(64):
        q1 = rng.normal(size=4); q1 /= np.linalg.norm(q1); q2 = rng.normal(size=4); q2 /= np.linalg.norm(q2)
        g = rng.normal(size=3); g *= 9.809 / np.linalg.norm(g)
        xs[i] = np.concatenate([rng.normal(0, 10, 3), q1, q2, rng.normal(0, 0.3, 3), rng.normal(0, 1, 3), rng.normal(0, 0.01, 6), g])
        xplus[i] = xs[i]
        L.refm_boxplus(xplus[i], np.ascontiguousarray(dxs[i]))
        L.refm_boxminus(np.ascontiguousarray(xplus[i]), np.ascontiguousarray(xs[i]), xminus[i])

    # one scan step per extrinsic mode on a seeded world (same generators as tests/golden/make_golden.py)
    world = synth.make_world(30_000, 0.2, seed=synth.SEED + 100)
    scan = synth.make_scan(world, 3_000, seed=synth.SEED + 101, pos=(1.0, -0.5, 0.0), rpy=(0.0, 0.01, 0.3))
    pos, rot = synth.perturb_state(scan, seed=synth.SEED + 102)
    x0 = capi.make_state26(pos, rot, scan.off_R, scan.off_T)
    P0 = synth.init_cov()
    n = len(scan.body)
    L.refm_map_build(np.ascontiguousarray(world.points), len(world.points), 0.2)
    out = {"eigen_version": np.array(L.refm_eigen_version().decode()), "planes_in": planes_in, "planes_out": planes_out, "planes_ok": planes_ok,
           "xs": xs, "dxs": dxs, "xplus": xplus, "xminus": xminus, "map": world.points, "body": scan.body, "x0": x0, "P0": P0}
    for ext, k in ((0, "noext"), (1, "ext")):
        L.refm_scan_set(np.ascontiguousarray(scan.body), n)
        hx = np.zeros((n, 12)); h = np.zeros(n); m = C.c_int(0); rm = C.c_double(0)
        wpts = np.zeros((n, 3), np.float32); nv = np.zeros((n, 4), np.float32); sel = np.zeros(n, np.uint8)
        nx = np.full((n, 5, 3), np.nan, np.float32); nc = np.zeros(n, np.int32)
        valid = L.refm_h_share_model(np.ascontiguousarray(x0), 1, ext, hx, h, C.byref(m), C.byref(rm), wpts, nv, sel, nx, nc)
        out.update({f"{k}_valid": valid, f"{k}_m": m.value, f"{k}_res_mean": rm.value, f"{k}_hx": hx[:m.value].copy(), f"{k}_h": h[:m.value].copy(),
                    f"{k}_world": wpts, f"{k}_normvec": nv, f"{k}_sel": sel, f"{k}_nbr_xyz": nx, f"{k}_nbr_cnt": nc})
        L.refm_scan_set(np.ascontiguousarray(scan.body), n)
        x1 = np.ascontiguousarray(x0.copy()); P1 = np.ascontiguousarray(P0.copy())
        L.refm_update_iterated(x1, P1, 3, ext, 0.001)
        out.update({f"{k}_x_post": x1, f"{k}_P_post": P1})
    np.savez_compressed(os.path.join(HERE, "model_ref.npz"), **out)
    print("wrote tests/golden/model_ref.npz (Eigen %s): %d planes (%d accepted), m = %d / %d" %
          (out["eigen_version"], len(planes_in), int(planes_ok.sum()), out["noext_m"], out["ext_m"]))
    return 0


if __name__ == "__main__":
    sys.exit(main())
