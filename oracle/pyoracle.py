"""TEST INFRASTRUCTURE ONLY -- ctypes access to the CPU oracle (oracle/liblio_oracle.so) and,
when it has been built, to the reference's own ikd-Tree (oracle/_ref/libikd_ref.so).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
The product package (better_fastlio2_amd/) must never do so.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ORACLE_SO = os.path.join(HERE, "liblio_oracle.so")
REF_SO = os.path.join(HERE, "_ref", "libikd_ref.so")

_f32p = np.ctypeslib.ndpointer(np.float32, flags="C_CONTIGUOUS")
_f64p = np.ctypeslib.ndpointer(np.float64, flags="C_CONTIGUOUS")
_i32p = np.ctypeslib.ndpointer(np.int32, flags="C_CONTIGUOUS")
_u8p = np.ctypeslib.ndpointer(np.uint8, flags="C_CONTIGUOUS")


def build(force: bool = False) -> None:
    """Compile the oracle (and oracle/_ref when /root/reference is present)."""
    if force or not os.path.exists(ORACLE_SO) or \
            os.path.getmtime(ORACLE_SO) < os.path.getmtime(os.path.join(HERE, "lio_oracle.cpp")):
        subprocess.check_call(["make", "-C", HERE, "liblio_oracle.so"], stdout=subprocess.DEVNULL)
    if os.path.exists("/root/reference/include/ikd-Tree/ikd_Tree.cpp") and \
            (force or not os.path.exists(REF_SO)):
        subprocess.check_call(["make", "-C", HERE, "ref"], stdout=subprocess.DEVNULL)


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(ORACLE_SO)
        L.orc_map_create.restype = C.c_void_p
        L.orc_map_create.argtypes = [C.c_float]
        L.orc_map_destroy.argtypes = [C.c_void_p]
        L.orc_map_build.argtypes = [C.c_void_p, _f32p, C.c_int]
        L.orc_map_add.restype = C.c_int
        L.orc_map_add.argtypes = [C.c_void_p, _f32p, C.c_int, C.c_int]
        L.orc_map_delete_boxes.restype = C.c_int
        L.orc_map_delete_boxes.argtypes = [C.c_void_p, _f32p, C.c_int]
        L.orc_map_validnum.restype = C.c_int
        L.orc_map_validnum.argtypes = [C.c_void_p]
        L.orc_map_flatten.restype = C.c_int
        L.orc_map_flatten.argtypes = [C.c_void_p, _f32p, C.c_int]
        L.orc_map_knn.restype = C.c_double
        L.orc_map_knn.argtypes = [C.c_void_p, _f32p, C.c_int, C.c_int, _f32p, _f32p, _i32p, C.c_int]
        L.orc_set_sum_order.argtypes = [C.c_int, C.c_ulonglong]
        L.orc_esti_plane.restype = C.c_int
        L.orc_esti_plane.argtypes = [_f32p, C.c_float, _f32p]
        L.orc_body_to_world.argtypes = [_f64p, _f32p, C.c_int, _f32p]
        L.orc_decision_margins.argtypes = [_f32p, _f32p, _i32p, _f32p, _f32p, C.c_int, _f32p, _f32p]
        L.orc_h_share_model.restype = C.c_int
        L.orc_h_share_model.argtypes = [C.c_void_p, _f64p, _f32p, C.c_int, C.c_int, C.c_int, C.c_int,
                                        _f32p, _f32p, _i32p, _u8p, _f32p, _f32p, _f32p, _f64p, _f64p,
                                        C.POINTER(C.c_int), C.POINTER(C.c_double)]
        L.orc_state_boxplus.argtypes = [_f64p, _f64p]
        L.orc_state_boxminus.argtypes = [_f64p, _f64p, _f64p]
        L.orc_update_iterated.argtypes = [C.c_void_p, _f64p, _f64p, _f32p, C.c_int, C.c_double, C.c_int,
                                          _f64p, C.c_int, C.c_int, _f32p, _f32p, _i32p, _u8p, _f64p,
                                          C.POINTER(C.c_int)]
        L.orc_hth.argtypes = [_f64p, _f64p, C.c_int, _f64p, _f64p]
        L.orc_map_incremental_classify.argtypes = [_f32p, C.c_int, _f32p, _i32p, C.c_double, C.c_int, _u8p]
        L.orc_set_external_knn.argtypes = [C.c_void_p, C.c_void_p]
        L.orc_undistort.argtypes = [_f32p, C.c_int, _f64p, C.c_int, _f64p, _f32p, _f32p]
        L.orc_voxel_grid.restype = C.c_int
        L.orc_voxel_grid.argtypes = [_f32p, C.c_int, C.c_float, _f32p, C.c_int]
        L.orc_fov_segment.restype = C.c_int
        L.orc_fov_segment.argtypes = [_f32p, _f64p, C.c_double, C.c_float, _f32p]
        _lib = L
    return _lib


def _c32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


class OracleMap:
    """Restated ikd-Tree live set (orc_map_*)."""

    def __init__(self, downsample_size: float = 0.2):
        self.L = lib()
        self.h = self.L.orc_map_create(downsample_size)

    def __del__(self):
        try:
            self.L.orc_map_destroy(self.h)
        except Exception:
            pass

    def build(self, xyz):
        xyz = _c32(xyz)
        self.L.orc_map_build(self.h, xyz, len(xyz))

    def add(self, xyz, downsample_on: bool) -> int:
        xyz = _c32(xyz)
        return self.L.orc_map_add(self.h, xyz, len(xyz), int(downsample_on))

    def delete_boxes(self, boxes) -> int:
        boxes = _c32(boxes).reshape(-1, 6)
        return self.L.orc_map_delete_boxes(self.h, boxes, len(boxes))

    def validnum(self) -> int:
        return self.L.orc_map_validnum(self.h)

    def flatten(self):
        n = self.validnum()
        out = np.zeros((max(n, 1), 3), np.float32)
        self.L.orc_map_flatten(self.h, out, n)
        return out[:n]

    def knn(self, q, k: int = 5, method: int = 1):
        q = _c32(q)
        nq = len(q)
        xyz = np.full((nq, k, 3), np.nan, np.float32)
        d = np.full((nq, k), np.inf, np.float32)
        cnt = np.zeros(nq, np.int32)
        secs = self.L.orc_map_knn(self.h, q, nq, k, xyz, d, cnt, method)
        return xyz, d, cnt, secs


def esti_plane(pts5, threshold: float = 0.1):
    out = np.zeros(4, np.float32)
    ok = lib().orc_esti_plane(_c32(pts5).reshape(5, 3), threshold, out)
    return bool(ok), out


SUM_ORDERS = {"sequential": 0, "eigen_sse2_packet": 1, "eigen_sse3_hadd": 2, "random": 3}


def set_sum_order(mode="sequential", seed: int = 0):
    """Order of the fp32 reductions inside esti_plane's QR (orc_set_sum_order); "sequential" is the default."""
    lib().orc_set_sum_order(SUM_ORDERS[mode] if isinstance(mode, str) else int(mode), int(seed))


def eigen_order_sensitivity(omap, x26, P, body, max_iter: int = 3, extrinsic_est_en: bool = False, knn_method: int = 1,
                            random_seeds=(1, 2, 3)):
    """What "parity unpinned" can cost on one scan step: the model is re-run with every reduction of esti_plane's QR
    added in Eigen's two packet orders and in random orders (same neighbours, same prior); returned per variant, against
    the sequential oracle: accept-mask flips / max |d n^| / max |d pd2| of the FIRST pass (identical pose, so nothing but
    the order differs), and the posterior's position / rotation difference after the whole iterated update."""
    body = _c32(body)
    variants = [("eigen_sse2_packet", 0), ("eigen_sse3_hadd", 0)] + [("random", s) for s in random_seeds]

    def run():
        arr = ScanArrays(len(body))
        h_share_model(omap, np.ascontiguousarray(x26[:14], np.float64), body, arr, True, extrinsic_est_en, knn_method)
        x, _, _, _ = update_iterated(omap, x26, P, body, 0.001, max_iter, None, extrinsic_est_en, knn_method)
        return arr, x

    set_sum_order("sequential")
    a0, x0 = run()
    out = {}
    try:
        for name, seed in variants:
            set_sum_order(name, seed)
            a, x = run()
            both = (a.selected != 0) & (a0.selected != 0)
            dn = np.abs(a.normvec[both, :3] - a0.normvec[both, :3])
            dp = np.abs(a.normvec[both, 3] - a0.normvec[both, 3])
            q1, q2 = x[3:7], x0[3:7]
            out[name if name != "random" else f"random_seed{seed}"] = {
                "accept_mask_flips": int((a.selected != a0.selected).sum()),
                "max_abs_dn": float(dn.max()) if dn.size else 0.0,
                "max_abs_dpd2_m": float(dp.max()) if dp.size else 0.0,
                "planes_with_any_bit_changed": int((a.normvec[both].view(np.uint32) != a0.normvec[both].view(np.uint32)).any(axis=1).sum()),
                "posterior_dpos_m": float(np.linalg.norm(x[:3] - x0[:3])),
                "posterior_drot_rad": float(4.0 * np.arcsin(min(1.0, 0.5 * min(np.linalg.norm(q1 - q2), np.linalg.norm(q1 + q2))))),
            }
    finally:
        set_sum_order("sequential")
    out["points"] = int(len(body))
    out["selected_sequential"] = int((a0.selected != 0).sum())
    return out


def decision_margins(nbr_xyz, nbr_d, nbr_cnt, world, body):
    """(margin_plane, margin_s): distance of each point's 0.1 m inlier test / s > 0.9 test to its threshold."""
    n = len(world)
    mp = np.zeros(n, np.float32)
    ms = np.zeros(n, np.float32)
    lib().orc_decision_margins(_c32(nbr_xyz), _c32(nbr_d), np.ascontiguousarray(nbr_cnt, np.int32), _c32(world), _c32(body),
                               n, mp, ms)
    return mp, ms


def body_to_world(state14, body):
    body = _c32(body)
    out = np.zeros_like(body)
    lib().orc_body_to_world(np.ascontiguousarray(state14, np.float64), body, len(body), out)
    return out


class ScanArrays:
    """The reference's per-scan globals (Nearest_Points, point_selected_surf, ...)."""

    def __init__(self, n: int):
        self.nbr_xyz = np.full((n, 5, 3), np.nan, np.float32)
        self.nbr_d = np.full((n, 5), np.inf, np.float32)
        self.nbr_cnt = np.zeros(n, np.int32)
        self.selected = np.ones(n, np.uint8)       # laserMapping.cpp:2131
        self.world = np.zeros((n, 3), np.float32)
        self.normvec = np.zeros((n, 4), np.float32)
        self.res_last = np.zeros(n, np.float32)


def h_share_model(omap: OracleMap, state14, body, arrays: ScanArrays, converge: bool,
                  extrinsic_est_en: bool, knn_method: int = 1):
    body = _c32(body)
    n = len(body)
    h_x = np.zeros((n, 12), np.float64)
    h = np.zeros(n, np.float64)
    m = C.c_int(0)
    rm = C.c_double(0)
    valid = lib().orc_h_share_model(omap.h, np.ascontiguousarray(state14, np.float64), body, n,
                                    int(converge), int(extrinsic_est_en), knn_method,
                                    arrays.nbr_xyz, arrays.nbr_d, arrays.nbr_cnt, arrays.selected,
                                    arrays.world, arrays.normvec, arrays.res_last, h_x, h,
                                    C.byref(m), C.byref(rm))
    return bool(valid), h_x[:m.value].copy(), h[:m.value].copy(), m.value, rm.value


def hth(h_x, h):
    HTH = np.zeros((12, 12))
    HTh = np.zeros(12)
    lib().orc_hth(np.ascontiguousarray(h_x), np.ascontiguousarray(h), len(h), HTH, HTh)
    return HTH, HTh


def update_iterated(omap: OracleMap, x26, P, body, R: float = 0.001, max_iter: int = 3,
                    limit=None, extrinsic_est_en: bool = False, knn_method: int = 1):
    body = _c32(body)
    n = len(body)
    x = np.ascontiguousarray(x26, np.float64).copy()
    Pm = np.ascontiguousarray(P, np.float64).copy()
    lim = np.full(23, 0.001) if limit is None else np.ascontiguousarray(limit, np.float64)
    arrays = ScanArrays(n)
    log = np.zeros((max_iter + 2, 4))
    nc = C.c_int(0)
    lib().orc_update_iterated(omap.h, x, Pm, body, n, R, max_iter, lim, int(extrinsic_est_en),
                              knn_method, arrays.nbr_xyz, arrays.nbr_d, arrays.nbr_cnt,
                              arrays.selected, log, C.byref(nc))
    return x, Pm, arrays, log[:nc.value]


def state_boxplus(x26, dx23):
    x = np.ascontiguousarray(x26, np.float64).copy()
    lib().orc_state_boxplus(x, np.ascontiguousarray(dx23, np.float64))
    return x


def state_boxminus(x26, y26):
    out = np.zeros(23)
    lib().orc_state_boxminus(np.ascontiguousarray(x26, np.float64),
                             np.ascontiguousarray(y26, np.float64), out)
    return out


def map_incremental_classify(world, nbr_xyz, nbr_cnt, filter_size_map_min: float = 0.2,
                             flg_EKF_inited: bool = True):
    world = _c32(world)
    cls = np.zeros(len(world), np.uint8)
    lib().orc_map_incremental_classify(world, len(world), _c32(nbr_xyz), np.ascontiguousarray(nbr_cnt, np.int32),
                                       filter_size_map_min, int(flg_EKF_inited), cls)
    return cls


# ---------------------------------------------------------------------------------------------
#  the reference's own ikd-Tree (oracle/_ref)
# ---------------------------------------------------------------------------------------------
_ref = None


def ref_available() -> bool:
    if not os.path.exists(REF_SO):
        try:
            build()
        except Exception:
            return False
    return os.path.exists(REF_SO)


def ref_lib():
    global _ref
    if _ref is None:
        if not ref_available():
            raise RuntimeError("oracle/_ref/libikd_ref.so is not built (needs /root/reference)")
        L = C.CDLL(REF_SO)
        L.ref_ikd_create.restype = C.c_void_p
        L.ref_ikd_create.argtypes = [C.c_float, C.c_float, C.c_float]
        L.ref_ikd_destroy.argtypes = [C.c_void_p]
        L.ref_ikd_set_downsample.argtypes = [C.c_void_p, C.c_float]
        L.ref_ikd_build.argtypes = [C.c_void_p, _f32p, C.c_int]
        L.ref_ikd_reconstruct.argtypes = [C.c_void_p, _f32p, C.c_int]
        L.ref_ikd_size.restype = C.c_int
        L.ref_ikd_size.argtypes = [C.c_void_p]
        L.ref_ikd_validnum.restype = C.c_int
        L.ref_ikd_validnum.argtypes = [C.c_void_p]
        L.ref_ikd_nearest.restype = C.c_double
        L.ref_ikd_nearest.argtypes = [C.c_void_p, _f32p, C.c_int, C.c_int, _f32p, _f32p, _i32p]
        L.ref_ikd_add_points.restype = C.c_int
        L.ref_ikd_add_points.argtypes = [C.c_void_p, _f32p, C.c_int, C.c_int]
        L.ref_ikd_delete_boxes.restype = C.c_int
        L.ref_ikd_delete_boxes.argtypes = [C.c_void_p, _f32p, C.c_int]
        L.ref_ikd_flatten.restype = C.c_int
        L.ref_ikd_flatten.argtypes = [C.c_void_p, _f32p, C.c_int]
        L.ref_ikd_acquire_removed.restype = C.c_int
        L.ref_ikd_acquire_removed.argtypes = [C.c_void_p, _f32p, C.c_int]
        L.ref_ikd_quiesce.argtypes = [C.c_int]
        _ref = L
    return _ref


class RefIkdTree:
    """KD_TREE<pcl::PointXYZINormal> of the reference, compiled verbatim."""

    def __init__(self, downsample_size: float = 0.2, delete_param: float = 0.3, balance_param: float = 0.6):
        # global `KD_TREE<PointType> ikdtree;` (laserMapping.cpp:116) -> defaults 0.5 / 0.6 / 0.2
        self.L = ref_lib()
        self.h = self.L.ref_ikd_create(0.5, 0.6, downsample_size)

    def __del__(self):
        try:
            self.L.ref_ikd_destroy(self.h)
        except Exception:
            pass

    def build(self, xyz):
        xyz = _c32(xyz)
        self.L.ref_ikd_build(self.h, xyz, len(xyz))

    def reconstruct(self, xyz):
        xyz = _c32(xyz)
        self.L.ref_ikd_reconstruct(self.h, xyz, len(xyz))

    def add(self, xyz, downsample_on: bool) -> int:
        xyz = _c32(xyz)
        return self.L.ref_ikd_add_points(self.h, xyz, len(xyz), int(downsample_on))

    def delete_boxes(self, boxes) -> int:
        boxes = _c32(boxes).reshape(-1, 6)
        return self.L.ref_ikd_delete_boxes(self.h, boxes, len(boxes))

    def validnum(self) -> int:
        return self.L.ref_ikd_validnum(self.h)

    def size(self) -> int:
        return self.L.ref_ikd_size(self.h)

    def flatten(self):
        n = max(self.size(), 1)
        out = np.zeros((n, 3), np.float32)
        c = self.L.ref_ikd_flatten(self.h, out, n)
        return out[:c]

    def acquire_removed(self, cap: int = 1 << 22):
        out = np.zeros((cap, 3), np.float32)
        c = self.L.ref_ikd_acquire_removed(self.h, out, cap)
        return out[:min(c, cap)]

    def knn(self, q, k: int = 5):
        q = _c32(q)
        nq = len(q)
        xyz = np.full((nq, k, 3), np.nan, np.float32)
        d = np.full((nq, k), np.inf, np.float32)
        cnt = np.zeros(nq, np.int32)
        secs = self.L.ref_ikd_nearest(self.h, q, nq, k, xyz, d, cnt)
        return xyz, d, cnt, secs

    def quiesce(self, ms: int = 200):
        self.L.ref_ikd_quiesce(ms)


def undistort(xyzt, poses22, end14):
    """UndistortPcl's point side: returns (time-sorted compensated xyz [n,3], sorted times [n])."""
    xyzt = _c32(xyzt).reshape(-1, 4)
    p = np.ascontiguousarray(poses22, np.float64).reshape(-1, 22)
    out = np.zeros((len(xyzt), 3), np.float32)
    tt = np.zeros(len(xyzt), np.float32)
    lib().orc_undistort(xyzt, len(xyzt), p, len(p), np.ascontiguousarray(end14, np.float64), out, tt)
    return out, tt


def voxel_grid(xyz, leaf: float):
    """pcl::VoxelGrid centroid filter (xyz only)."""
    xyz = _c32(xyz).reshape(-1, 3)
    out = np.zeros((max(len(xyz), 1), 3), np.float32)
    n = lib().orc_voxel_grid(xyz, len(xyz), leaf, out, len(xyz))
    if n < 0:
        raise OverflowError("leaf size too small for the extent of the cloud")
    return out[:n].copy()


class FovSegment:
    """lasermap_fov_segment state machine (LocalMap_Points / Localmap_Initialized)."""

    def __init__(self):
        self.state = np.zeros(7, np.float32)

    def step(self, pos_lid, cube_len: float, det_range: float):
        boxes = np.zeros((3, 6), np.float32)
        nb = lib().orc_fov_segment(self.state, np.ascontiguousarray(pos_lid, np.float64), cube_len, det_range, boxes)
        return boxes[:nb].copy()


def use_reference_knn(tree: "RefIkdTree | None"):
    """knn_method=2 of h_share_model / update_iterated then calls the reference's Nearest_Search."""
    if tree is None:
        lib().orc_set_external_knn(None, None)
    else:
        fn = C.cast(ref_lib().ref_ikd_nearest, C.c_void_p)
        lib().orc_set_external_knn(fn, tree.h)
