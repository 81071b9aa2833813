"""ctypes binding of liblio_hip.so (include/lio_hip.h) plus thin Python mirrors of the reference's
host-side objects for the LIO measurement path:

    MapIndex      <-> KD_TREE<PointType> ikdtree        (include/ikd-Tree/ikd_Tree.h:225-249)
    LioContext.measure(...)  <-> h_share_model(s, ekfom_data)   (src/laserMapping.cpp:1876-2004)
    Esekf         <-> esekfom::esekf<state_ikfom, 12, input_ikfom>  (esekfom.hpp)

There is NO CPU fallback: importing works anywhere (so that the CPU test-suite can check the ABI),
but creating a LioContext needs a gfx950 GPU and raises LioError otherwise.
"""
from __future__ import annotations

import ctypes as C
import weakref
import importlib.util
import os
import sys

import numpy as np

from . import build as _build

LIO_OK = 0
LIO_ERR_INVALID_ARG, LIO_ERR_NO_DEVICE, LIO_ERR_HIP, LIO_ERR_CAPACITY, LIO_ERR_NO_SCAN, LIO_ERR_NO_MAP = -1, -2, -3, -4, -5, -6
ERRORS = {-1: "LIO_ERR_INVALID_ARG", -2: "LIO_ERR_NO_DEVICE", -3: "LIO_ERR_HIP", -4: "LIO_ERR_CAPACITY",
          -5: "LIO_ERR_NO_SCAN", -6: "LIO_ERR_NO_MAP"}
PROF_KNN, PROF_FALLBACK, PROF_RESIDUAL, PROF_REDUCE, PROF_KNN_HINT = 0, 1, 2, 3, 4
PROF_NAMES = {PROF_KNN: "knn27", PROF_FALLBACK: "knn_tree", PROF_RESIDUAL: "residual", PROF_REDUCE: "reduce_final",
              PROF_KNN_HINT: "knn27_hint"}


class LioError(RuntimeError):
    pass


class lio_config(C.Structure):
    _fields_ = [("device", C.c_int), ("filter_size_map_min", C.c_float), ("cell_size", C.c_float),
                ("max_scan_points", C.c_int), ("max_map_points", C.c_longlong), ("lanes_per_query", C.c_int)]


class lio_out(C.Structure):
    _fields_ = [("m", C.c_int), ("valid", C.c_int), ("n_unresolved", C.c_int), ("reserved", C.c_int),
                ("res_mean", C.c_double), ("HTH", C.c_double * 144), ("HTh", C.c_double * 12),
                ("n_candidates", C.c_ulonglong)]


class lio_map_info(C.Structure):
    _fields_ = [("n_points", C.c_longlong), ("n_live", C.c_longlong), ("n_cells", C.c_longlong),
                ("table_slots", C.c_longlong), ("cell_size", C.c_float), ("downsample_size", C.c_float),
                ("bytes_resident", C.c_longlong), ("n_retired", C.c_longlong), ("n_incremental_updates", C.c_longlong),
                ("n_full_rebuilds", C.c_longlong), ("n_boundary_points", C.c_longlong)]


class lio_update_log(C.Structure):
    _fields_ = [("n_calls", C.c_int), ("n_search", C.c_int), ("converged", C.c_int), ("last_m", C.c_int),
                ("last_res_mean", C.c_double), ("solve_seconds", C.c_double), ("measure_seconds", C.c_double)]


class lio_imu_info(C.Structure):
    _fields_ = [("need_init", C.c_int), ("init_iter_num", C.c_int), ("mean_acc", C.c_double * 3), ("mean_gyr", C.c_double * 3),
                ("cov_acc", C.c_double * 3), ("cov_gyr", C.c_double * 3), ("last_lidar_end_time", C.c_double)]


# every symbol include/lio_hip.h declares (checked by tests/test_abi.py)
EXPORTS = [
    "lio_create", "lio_destroy", "lio_last_error", "lio_version", "lio_device_count", "lio_stream", "lio_synchronize",
    "lio_map_build", "lio_map_build_device", "lio_map_add", "lio_map_add_pair", "lio_map_add_pair_device", "lio_map_incremental", "lio_map_incremental_device", "lio_map_delete_boxes", "lio_replicate_unique_id", "lio_replicate_init", "lio_replicate_finalize", "lio_replicate_queue_boxes", "lio_replicate_step", "lio_replicate_defer", "lio_fov_segment", "lio_map_reconstruct",
    "lio_map_set_downsample_param", "lio_map_size", "lio_map_validnum", "lio_map_flatten", "lio_map_acquire_removed_points", "lio_map_get_info", "lio_map_nearest_search",
    "lio_scan_set", "lio_scan_set_device", "lio_scan_set_async", "lio_scan_pack_threads", "lio_host_register", "lio_host_unregister", "lio_scan_upload_async", "lio_scan_swap", "lio_host_alloc", "lio_host_free", "lio_scan_undistort", "lio_scan_upload_raw", "lio_scan_compensate", "lio_scan_voxel_filter", "lio_scan_prefetch_raw", "lio_scan_prep_begin", "lio_scan_prep_end", "lio_scan_get_cloud", "lio_measure", "lio_measure_begin", "lio_measure_end", "lio_measure_rows", "lio_measure_want_rows", "lio_eskf_update_iterated_dyn_share_callback", "lio_get_neighbors",
    "lio_get_scan_state", "lio_scan_size", "lio_profile_enable", "lio_census_enable", "lio_profile_reset", "lio_profile_get", "lio_debug_set", "lio_debug_get", "lio_debug_stamps", "lio_debug_knn_stamps",
    "lio_eskf_create", "lio_eskf_destroy", "lio_eskf_init_dyn_share", "lio_eskf_change_x", "lio_eskf_change_P",
    "lio_eskf_get_x", "lio_eskf_get_P", "lio_eskf_update_iterated_dyn_share_modified", "lio_eskf_scan_step_device", "lio_eskf_scan_step_pipelined", "lio_state_boxplus",
    "lio_state_boxminus", "lio_eskf_predict", "lio_process_noise_cov",
    "lio_imu_create", "lio_imu_destroy", "lio_imu_reset", "lio_imu_set_extrinsic", "lio_imu_set_cov", "lio_imu_process",
    "lio_imu_get_poses", "lio_imu_get_info",
]

_lib = None
_vp = C.c_void_p


def library_path() -> str:
    return _build.LIB


def load(build_if_missing: bool = True):
    """Load liblio_hip.so (building it with hipcc when it is missing).  Fails loudly otherwise."""
    global _lib
    if _lib is not None:
        return _lib
    # PyTorch-ROCm wheels bundle their own libamdhip64.so (same SONAME as /opt/rocm's); two HIP runtimes
    # in one process fight over the device ("No HIP GPUs are available" for whichever initialises
    # second).  Map torch's copy first -- without importing torch, which can take minutes on a cold
    # box -- so that liblio_hip.so and a later `import torch` share one runtime whatever the import order.
    if "torch" not in sys.modules:
        try:
            spec = importlib.util.find_spec("torch")
            cand = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so") if spec else None
            if cand and os.path.exists(cand):
                C.CDLL(cand, mode=C.RTLD_GLOBAL)
        except Exception:
            pass
    # LIO_HIP_LIBRARY: another build of the library (A/B of two builds on ONE box: the only comparison that holds, profiles/README.md)
    path = os.environ.get("LIO_HIP_LIBRARY") or library_path()
    if not os.path.exists(path):
        if not build_if_missing:
            raise LioError(f"{path} is missing: run `python -m better_fastlio2_amd.build` (no CPU fallback exists)")
        _build.build()
    L = C.CDLL(path)
    L.lio_version.restype = C.c_char_p
    L.lio_last_error.restype = C.c_char_p
    L.lio_last_error.argtypes = [_vp]
    L.lio_create.argtypes = [C.POINTER(lio_config), C.POINTER(_vp)]
    L.lio_destroy.argtypes = [_vp]
    L.lio_destroy.restype = None
    L.lio_stream.argtypes = [_vp]
    L.lio_stream.restype = _vp
    L.lio_synchronize.argtypes = [_vp]
    L.lio_map_build.argtypes = [_vp, _vp, C.c_longlong, C.c_int]
    L.lio_map_build_device.argtypes = [_vp, _vp, C.c_longlong]
    L.lio_map_add.argtypes = [_vp, _vp, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int)]
    L.lio_map_delete_boxes.argtypes = [_vp, _vp, C.c_int, C.POINTER(C.c_int)]
    L.lio_replicate_unique_id.argtypes = [_vp]
    L.lio_replicate_init.argtypes = [_vp, _vp, C.c_int, C.c_int, C.c_int]
    L.lio_replicate_finalize.argtypes = [_vp]
    L.lio_replicate_defer.argtypes = [_vp, C.c_int]
    L.lio_replicate_finalize.restype = None
    L.lio_replicate_queue_boxes.argtypes = [_vp, _vp, C.c_int]
    L.lio_replicate_step.argtypes = [_vp, _vp, C.c_double, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]
    L.lio_map_add_pair.argtypes = [_vp, _vp, C.c_int, _vp, C.c_int, C.c_int, C.POINTER(C.c_int)]
    L.lio_map_incremental_device.argtypes = [_vp, _vp, C.c_double, C.c_int, C.POINTER(_vp), C.POINTER(C.c_int), C.POINTER(_vp), C.POINTER(C.c_int)]
    L.lio_map_add_pair_device.argtypes = [_vp, _vp, C.c_int, _vp, C.c_int, C.POINTER(C.c_int)]
    L.lio_map_incremental.argtypes = [_vp, _vp, C.c_double, C.c_int, C.c_int, _vp, C.POINTER(C.c_int), _vp, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    L.lio_fov_segment.argtypes = [_vp, _vp, C.c_double, C.c_float, _vp, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    L.lio_map_reconstruct.argtypes = [_vp, _vp, C.c_longlong, C.c_int]
    L.lio_map_size.argtypes = [_vp]
    L.lio_map_size.restype = C.c_longlong
    L.lio_map_validnum.argtypes = [_vp]
    L.lio_map_validnum.restype = C.c_longlong
    L.lio_map_flatten.argtypes = [_vp, _vp, C.c_longlong, C.POINTER(C.c_longlong)]
    L.lio_map_set_downsample_param.argtypes = [_vp, C.c_float]
    L.lio_map_acquire_removed_points.argtypes = [_vp, _vp, C.c_longlong, C.POINTER(C.c_longlong)]
    L.lio_map_get_info.argtypes = [_vp, C.POINTER(lio_map_info)]
    L.lio_map_nearest_search.argtypes = [_vp, _vp, C.c_int, C.c_int, _vp, _vp, _vp]
    L.lio_scan_set.argtypes = [_vp, _vp, C.c_int, C.c_int]
    L.lio_scan_set_device.argtypes = [_vp, _vp, C.c_int]
    L.lio_scan_set_async.argtypes = [_vp, _vp, C.c_int, C.c_int]
    L.lio_scan_pack_threads.argtypes = [_vp, C.c_int]
    L.lio_host_register.argtypes = [_vp, C.c_size_t]
    L.lio_host_unregister.argtypes = [_vp]
    L.lio_scan_upload_async.argtypes = [_vp, _vp, C.c_int]
    L.lio_scan_swap.argtypes = [_vp]
    L.lio_host_alloc.argtypes = [C.c_size_t]
    L.lio_host_alloc.restype = _vp
    L.lio_measure_want_rows.argtypes = [_vp, C.c_int]
    L.lio_host_free.argtypes = [_vp]
    L.lio_host_free.restype = None
    L.lio_scan_undistort.argtypes = [_vp, _vp, C.c_int, C.c_int, C.c_int, _vp, C.c_int, _vp]
    L.lio_scan_upload_raw.argtypes = [_vp, _vp, C.c_int, C.c_int, C.c_int]
    L.lio_scan_compensate.argtypes = [_vp, _vp, C.c_int, _vp]
    L.lio_scan_voxel_filter.argtypes = [_vp, C.c_float, C.POINTER(C.c_int)]
    L.lio_scan_prefetch_raw.argtypes = [_vp, _vp, C.c_int, C.c_int, C.c_int]
    L.lio_scan_prep_begin.argtypes = [_vp]
    L.lio_scan_prep_end.argtypes = [_vp, C.POINTER(C.c_int)]
    L.lio_scan_get_cloud.argtypes = [_vp, C.c_int, _vp, C.c_int, C.POINTER(C.c_int)]
    L.lio_measure.argtypes = [_vp, _vp, C.c_int, C.c_int, C.POINTER(lio_out)]
    L.lio_measure_rows.argtypes = [_vp, _vp, _vp, C.c_int, C.c_int]
    L.lio_measure_begin.argtypes = [_vp, _vp, C.c_int, C.c_int]
    L.lio_measure_end.argtypes = [_vp, C.POINTER(lio_out)]
    L.lio_get_neighbors.argtypes = [_vp, _vp, _vp, _vp]
    L.lio_get_scan_state.argtypes = [_vp, _vp, _vp, _vp]
    L.lio_scan_size.argtypes = [_vp]
    L.lio_profile_enable.argtypes = [_vp, C.c_int]
    L.lio_profile_reset.argtypes = [_vp]
    L.lio_census_enable.argtypes = [_vp, C.c_int]
    L.lio_debug_set.argtypes = [_vp, C.c_char_p, C.c_longlong]
    L.lio_debug_get.argtypes = [_vp, C.c_char_p, C.POINTER(C.c_longlong)]
    L.lio_debug_stamps.argtypes = [_vp, _vp]
    L.lio_debug_knn_stamps.argtypes = [_vp, _vp, C.c_longlong, _vp]
    L.lio_profile_get.argtypes = [_vp, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_longlong)]
    L.lio_eskf_create.restype = _vp
    L.lio_eskf_destroy.argtypes = [_vp]
    L.lio_eskf_destroy.restype = None
    L.lio_eskf_init_dyn_share.argtypes = [_vp, C.c_int, _vp]
    L.lio_eskf_change_x.argtypes = [_vp, _vp]
    L.lio_eskf_change_P.argtypes = [_vp, _vp]
    L.lio_eskf_get_x.argtypes = [_vp, _vp]
    L.lio_eskf_get_P.argtypes = [_vp, _vp]
    L.lio_eskf_update_iterated_dyn_share_modified.argtypes = [_vp, _vp, C.c_double, C.c_int, C.POINTER(lio_update_log)]
    L.lio_eskf_scan_step_device.argtypes = [_vp, _vp, _vp, _vp, _vp, C.c_int, C.c_double, C.c_int, C.POINTER(lio_update_log)]
    L.lio_eskf_scan_step_pipelined.argtypes = [_vp, _vp, _vp, _vp, _vp, C.c_int, C.c_double, C.c_int, C.POINTER(lio_update_log)]
    L.lio_state_boxplus.argtypes = [_vp, _vp]
    L.lio_state_boxminus.argtypes = [_vp, _vp, _vp]
    L.lio_eskf_predict.argtypes = [_vp, C.c_double, _vp, _vp, _vp]
    L.lio_process_noise_cov.argtypes = [_vp]
    L.lio_imu_create.restype = _vp
    L.lio_imu_destroy.argtypes = [_vp]
    L.lio_imu_destroy.restype = None
    L.lio_imu_reset.argtypes = [_vp]
    L.lio_imu_set_extrinsic.argtypes = [_vp, _vp, _vp]
    L.lio_imu_set_cov.argtypes = [_vp, _vp, _vp, _vp, _vp]
    L.lio_imu_process.argtypes = [_vp, _vp, _vp, C.c_double, C.c_double, _vp, C.c_int, C.c_int, C.c_int, _vp, C.c_int,
                                  C.POINTER(C.c_int)]
    L.lio_imu_get_poses.argtypes = [_vp, _vp, C.c_int, C.POINTER(C.c_int)]
    L.lio_imu_get_info.argtypes = [_vp, C.POINTER(lio_imu_info)]
    _lib = L
    return L


def pinned_points(xyz) -> np.ndarray:
    """[n, 4] float32 copy of xyz in pinned host memory (lio_host_alloc): the buffer a sensor driver would fill.
    The memory lives as long as the returned array (and views of it).  Lifetime rule of the C interface: memory handed to
    lio_scan_upload_async / lio_scan_upload_raw / lio_scan_set_async is read AFTER those calls return (until lio_scan_swap +
    the next pass / a call that waits for the stream); LioContext.scan_upload_async, scan_undistort and scan_set_async hold a
    reference for that long, so passing a temporary is safe through them -- a caller of the raw ctypes entry points keeps the
    array alive itself."""
    L = load()
    xyz = np.ascontiguousarray(xyz, np.float32)
    n = len(xyz)
    p = L.lio_host_alloc(max(n, 1) * 16)
    if not p:
        raise LioError("lio_host_alloc failed")
    buf = (C.c_float * (max(n, 1) * 4)).from_address(p)
    # the ctypes array is the base object of every view numpy hands out: when the last view dies it dies, and the
    # page-locked memory goes back (a per-scan caller would otherwise leak it for the life of the process)
    weakref.finalize(buf, L.lio_host_free, p)
    arr = np.frombuffer(buf, np.float32).reshape(-1, 4)[:n]
    arr[:] = 0.0
    arr[:, :xyz.shape[1]] = xyz[:, :4]            # [n, 4] input: x, y, z, t records (a raw sweep)
    return arr


_cbb = None


def callback_bench():
    """ctypes handle of lib/libcallback_bench.so (examples/callback_bench.cpp): the zero-patch callback chain."""
    global _cbb
    if _cbb is None:
        load()
        B = C.CDLL(_build.build_callback_bench())
        B.cbb_setup.argtypes = [_vp, C.c_int]
        B.cbb_add_scan.argtypes = [_vp, C.c_int, C.c_int]
        B.cbb_step.argtypes = [_vp, C.c_int, _vp, _vp, C.c_double, C.c_int, C.c_int, _vp]
        B.cbb_last_m.restype = C.c_int
        B.cbb_last_status.restype = C.c_int
        B.cbb_last_validnum.restype = C.c_int
        B.cbb_option.argtypes = [C.c_int, C.c_int]
        B.cbb_option.restype = None
        B.cbb_take_model_seconds.restype = C.c_double
        _cbb = B
    return _cbb


def replicate_unique_id() -> bytes:
    """ncclGetUniqueId through the library (rank 0); hand the bytes to every rank's replicate_init."""
    L = load()
    buf = (C.c_char * 128)()
    rc = L.lio_replicate_unique_id(buf)
    if rc != LIO_OK:
        raise LioError("lio_replicate_unique_id failed (librccl not loadable?)")
    return bytes(buf)


def _ptr(a: np.ndarray):
    return a.ctypes.data_as(_vp)


def _points(a) -> np.ndarray:
    a = np.ascontiguousarray(a, dtype=np.float32)
    if a.ndim != 2 or a.shape[1] < 3:
        raise ValueError("points must be [n, >=3] float32")
    return a


def _stride(a: np.ndarray) -> int:
    """Record stride in bytes (numpy reports (0, 0) strides for empty arrays)."""
    return a.shape[1] * a.itemsize


class LioContext:
    """One GPU-resident map + scan + measurement-model context (lio_ctx)."""

    def __init__(self, filter_size_map_min: float = 0.2, max_scan_points: int = 100000,
                 max_map_points: int = 6_000_000, device: int = 0, cell_size: float = 0.0,
                 lanes_per_query: int = 0):
        self.L = load()
        cfg = lio_config(device, filter_size_map_min, cell_size, max_scan_points, max_map_points, lanes_per_query)
        h = _vp()
        rc = self.L.lio_create(C.byref(cfg), C.byref(h))
        if rc != LIO_OK:
            raise LioError(f"lio_create failed: {ERRORS.get(rc, rc)} (a gfx950 GPU is required; there is no CPU path)")
        self.h = h
        self.max_scan_points = max_scan_points

    def close(self):
        if getattr(self, "h", None):
            self.L.lio_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc: int, what: str):
        if rc != LIO_OK:
            raise LioError(f"{what}: {ERRORS.get(rc, rc)}: {self.L.lio_last_error(self.h).decode()}")

    # ---- map (KD_TREE facade) --------------------------------------------------------------------
    def map_build(self, pts):
        p = _points(pts)
        self._chk(self.L.lio_map_build(self.h, _ptr(p), len(p), _stride(p)), "lio_map_build")

    def map_reconstruct(self, pts):
        """KD_TREE::reconstruct (ikd_Tree.cpp:1394-1405): the map becomes exactly this cloud."""
        p = _points(pts)
        self._chk(self.L.lio_map_reconstruct(self.h, _ptr(p), len(p), _stride(p)), "lio_map_reconstruct")

    def map_build_device(self, d_ptr: int, n: int):
        self._chk(self.L.lio_map_build_device(self.h, _vp(d_ptr), n), "lio_map_build_device")

    def map_add(self, pts, downsample_on: bool) -> int:
        p = _points(pts)
        n_added = C.c_int(0)
        self._chk(self.L.lio_map_add(self.h, _ptr(p), len(p), _stride(p), int(downsample_on), C.byref(n_added)),
                  "lio_map_add")
        return n_added.value

    def map_add_pair(self, add, nodown) -> int:
        """Add_Points(add, True); Add_Points(nodown, False) with one index rebuild."""
        a = np.ascontiguousarray(add, np.float32).reshape(-1, 3)
        b = np.ascontiguousarray(nodown, np.float32).reshape(-1, 3)
        cnt = C.c_int(0)
        self._chk(self.L.lio_map_add_pair(self.h, _ptr(a) if len(a) else None, len(a), _ptr(b) if len(b) else None, len(b),
                                          12, C.byref(cnt)), "lio_map_add_pair")
        return cnt.value

    def map_incremental_device(self, state14, filter_size_map_min: float = 0.2, flg_EKF_inited: bool = True):
        """map_incremental's two lists left in HBM: (ptr_add, n_add, ptr_nodown, n_nodown), float4 records."""
        s = np.ascontiguousarray(state14, dtype=np.float64)
        pa, pb, na, nb = _vp(), _vp(), C.c_int(0), C.c_int(0)
        self._chk(self.L.lio_map_incremental_device(self.h, _ptr(s), float(filter_size_map_min), int(flg_EKF_inited),
                                                    C.byref(pa), C.byref(na), C.byref(pb), C.byref(nb)), "lio_map_incremental_device")
        return pa.value or 0, na.value, pb.value or 0, nb.value

    def map_add_pair_device(self, ptr_add: int, n_add: int, ptr_nodown: int, n_nodown: int) -> int:
        cnt = C.c_int(0)
        self._chk(self.L.lio_map_add_pair_device(self.h, _vp(ptr_add), n_add, _vp(ptr_nodown), n_nodown, C.byref(cnt)),
                  "lio_map_add_pair_device")
        return cnt.value

    def map_incremental(self, state14, filter_size_map_min: float = 0.2, flg_EKF_inited: bool = True,
                        apply: bool = True, want_lists: bool = False) -> dict:
        """map_incremental (laserMapping.cpp:1440-1496) on the current scan and its last neighbours."""
        s = np.ascontiguousarray(state14, dtype=np.float64)
        n = self.L.lio_scan_size(self.h)
        a = np.zeros((max(n, 1), 3), np.float32) if want_lists else None
        b = np.zeros((max(n, 1), 3), np.float32) if want_lists else None
        na, nb, cnt = C.c_int(0), C.c_int(0), C.c_int(0)
        self._chk(self.L.lio_map_incremental(self.h, _ptr(s), float(filter_size_map_min), int(flg_EKF_inited), int(apply),
                                             _ptr(a) if want_lists else None, C.byref(na), _ptr(b) if want_lists else None,
                                             C.byref(nb), C.byref(cnt)), "lio_map_incremental")
        out = {"n_add": na.value, "n_nodown": nb.value, "add_counter": cnt.value}
        if want_lists:
            out["add"] = a[:na.value].copy()
            out["nodown"] = b[:nb.value].copy()
        return out

    # ---- replicated maps (one rank per GPU) ---------------------------------------------------------
    def replicate_init(self, unique_id: bytes, rank: int, world: int, list_capacity: int = 0):
        """lio_replicate_init: unique_id = the 128 bytes of replicate_unique_id() made by rank 0 (ignored for world 1)."""
        buf = (C.c_char * 128).from_buffer_copy(unique_id.ljust(128, b"\0")[:128]) if unique_id else None
        self._chk(self.L.lio_replicate_init(self.h, buf, rank, world, list_capacity), "lio_replicate_init")

    def replicate_defer(self, on: bool = True):
        """lio_replicate_defer: the gathered lists of exchange k enter the map at the start of exchange k + 1; the read-only half
        of their Add_Points runs on a side stream beside scan step k + 1."""
        self._chk(self.L.lio_replicate_defer(self.h, int(on)), "lio_replicate_defer")

    def replicate_queue_boxes(self, boxes):
        b = np.ascontiguousarray(boxes, dtype=np.float32).reshape(-1, 6)
        self._chk(self.L.lio_replicate_queue_boxes(self.h, _ptr(b), len(b)), "lio_replicate_queue_boxes")

    def replicate_step(self, state14, filter_size_map_min: float = 0.2, flg_EKF_inited: bool = True) -> dict:
        s = np.ascontiguousarray(state14, dtype=np.float64)
        na, nb, nd, cnt = C.c_int(0), C.c_int(0), C.c_int(0), C.c_int(0)
        self._chk(self.L.lio_replicate_step(self.h, _ptr(s), float(filter_size_map_min), int(flg_EKF_inited), C.byref(na),
                                            C.byref(nb), C.byref(nd), C.byref(cnt)), "lio_replicate_step")
        return {"n_add": na.value, "n_nodown": nb.value, "n_deleted": nd.value, "add_counter": cnt.value}

    def map_delete_boxes(self, boxes) -> int:
        b = np.ascontiguousarray(boxes, dtype=np.float32).reshape(-1, 6)
        n = C.c_int(0)
        self._chk(self.L.lio_map_delete_boxes(self.h, _ptr(b), len(b), C.byref(n)), "lio_map_delete_boxes")
        return n.value

    def fov_segment(self, pos_lid, cube_len: float, det_range: float):
        """lasermap_fov_segment (laserMapping.cpp:1136-1200): returns (boxes [nb, 6], n_deleted)."""
        p = np.ascontiguousarray(pos_lid, np.float64)
        boxes = np.zeros((3, 6), np.float32)
        nb, nd = C.c_int(0), C.c_int(0)
        self._chk(self.L.lio_fov_segment(self.h, _ptr(p), float(cube_len), float(det_range), _ptr(boxes), C.byref(nb), C.byref(nd)),
                  "lio_fov_segment")
        return boxes[:nb.value].copy(), nd.value

    def set_downsample_param(self, downsample_size: float):
        self._chk(self.L.lio_map_set_downsample_param(self.h, float(downsample_size)), "lio_map_set_downsample_param")

    def map_size(self) -> int:
        return self.L.lio_map_size(self.h)

    def map_validnum(self) -> int:
        return self.L.lio_map_validnum(self.h)

    def map_flatten(self) -> np.ndarray:
        n = self.map_validnum()
        out = np.zeros((max(n, 1), 3), np.float32)
        got = C.c_longlong(0)
        self._chk(self.L.lio_map_flatten(self.h, _ptr(out), n, C.byref(got)), "lio_map_flatten")
        return out[:min(n, got.value)]

    def acquire_removed_points(self) -> np.ndarray:
        """KD_TREE::acquire_removed_points (ikd_Tree.cpp:559-571): box-deleted points reclaimed since the last call."""
        got = C.c_longlong(0)
        self._chk(self.L.lio_map_acquire_removed_points(self.h, None, 0, C.byref(got)), "lio_map_acquire_removed_points")
        out = np.zeros((max(got.value, 1), 3), np.float32)
        self._chk(self.L.lio_map_acquire_removed_points(self.h, _ptr(out), got.value, C.byref(got)), "lio_map_acquire_removed_points")
        return out[:got.value]

    def map_info(self) -> dict:
        info = lio_map_info()
        self._chk(self.L.lio_map_get_info(self.h, C.byref(info)), "lio_map_get_info")
        return {k: getattr(info, k) for k, _ in lio_map_info._fields_}

    def nearest_search(self, queries):
        q = _points(queries)
        n = len(q)
        xyz = np.full((n, 5, 3), np.nan, np.float32)
        d = np.full((n, 5), np.inf, np.float32)
        cnt = np.zeros(n, np.int32)
        self._chk(self.L.lio_map_nearest_search(self.h, _ptr(q), n, _stride(q), _ptr(xyz), _ptr(d), _ptr(cnt)),
                  "lio_map_nearest_search")
        return xyz, d, cnt

    # ---- scan + measurement model ------------------------------------------------------------------
    def scan_set(self, body):
        b = _points(body)
        self._chk(self.L.lio_scan_set(self.h, _ptr(b), len(b), _stride(b)), "lio_scan_set")

    def scan_set_async(self, body):
        """lio_scan_set without the wait.  Page-locked records (pinned_points, lio_host_register) are pulled by a kernel where
        they lie: the context keeps a reference to the array until the next scan replaces it."""
        b = _points(body)
        self._chk(self.L.lio_scan_set_async(self.h, _ptr(b), len(b), _stride(b)), "lio_scan_set_async")
        self._scan_keep = b

    def scan_pack_threads(self, n: int):
        """lio_scan_pack_threads: 2 = a second packing thread inside the library for the assisted pull, 1 = none (default)."""
        self._chk(self.L.lio_scan_pack_threads(self.h, int(n)), "lio_scan_pack_threads")

    def scan_undistort(self, raw_xyzt, imu_poses=None, end_state14=None):
        """UndistortPcl's point side: raw_xyzt [n, 4] = x, y, z, time offset in ms; imu_poses [K, 22]
        (offset_time, acc3, gyr3, vel3, pos3, rot9); end_state14 = propagated state at scan end.
        x, y, z, t records in page-locked memory (pinned_points) are read by the copy engine IN PLACE after this call has
        returned: the context holds a reference to the array until the next raw scan replaces it (the C rule: untouched until a
        call that waits for the stream has returned -- lio_scan_voxel_filter, lio_synchronize)."""
        r = np.ascontiguousarray(raw_xyzt, np.float32)
        assert r.ndim == 2 and r.shape[1] == 4
        self._raw_keep = r
        if imu_poses is None:
            self._chk(self.L.lio_scan_undistort(self.h, _ptr(r), len(r), 16, 12, None, 0, None), "lio_scan_undistort")
            return
        p = np.ascontiguousarray(imu_poses, np.float64).reshape(-1, 22)
        e = np.ascontiguousarray(end_state14, np.float64)
        self._chk(self.L.lio_scan_undistort(self.h, _ptr(r), len(r), 16, 12, _ptr(p), len(p), _ptr(e)), "lio_scan_undistort")

    def scan_voxel_filter(self, leaf_size: float) -> int:
        n = C.c_int(0)
        self._chk(self.L.lio_scan_voxel_filter(self.h, float(leaf_size), C.byref(n)), "lio_scan_voxel_filter")
        return n.value

    def scan_prefetch_raw(self, pinned_xyzt: np.ndarray):
        """A sweep of x, y, z, t records in page-locked memory (pinned_points) starts its way to the GPU on the copy stream; the
        scan_undistort / ImuProcess.process call that names the same array takes it where it is.  Keep the array alive until then."""
        assert pinned_xyzt.dtype == np.float32 and pinned_xyzt.ndim == 2 and pinned_xyzt.shape[1] == 4 and pinned_xyzt.flags.c_contiguous
        self._chk(self.L.lio_scan_prefetch_raw(self.h, _ptr(pinned_xyzt), len(pinned_xyzt), 16, 12), "lio_scan_prefetch_raw")
        self._prefetch_keep = pinned_xyzt

    def scan_prep_begin(self):
        """From here to scan_prep_end the raw-scan calls (scan_undistort, ImuProcess.process, scan_voxel_filter) only enqueue, on a
        stream of their own; map calls of the current scan go on meanwhile."""
        self._chk(self.L.lio_scan_prep_begin(self.h), "lio_scan_prep_begin")

    def scan_prep_end(self) -> int:
        n = C.c_int(0)
        self._chk(self.L.lio_scan_prep_end(self.h, C.byref(n)), "lio_scan_prep_end")
        return n.value

    def scan_cloud(self, which: int) -> np.ndarray:
        n = C.c_int(0)
        self._chk(self.L.lio_scan_get_cloud(self.h, which, None, 0, C.byref(n)), "lio_scan_get_cloud")
        out = np.zeros((max(n.value, 1), 3), np.float32)
        self._chk(self.L.lio_scan_get_cloud(self.h, which, _ptr(out), n.value, C.byref(n)), "lio_scan_get_cloud")
        return out[:n.value]

    def scan_set_device(self, d_ptr: int, n: int):
        self._chk(self.L.lio_scan_set_device(self.h, _vp(d_ptr), n), "lio_scan_set_device")

    def scan_upload_async(self, pinned_f4: np.ndarray):
        """Start the host-to-device copy of the NEXT scan ([n, 4] float32 in pinned memory, see pinned_points)."""
        assert pinned_f4.dtype == np.float32 and pinned_f4.ndim == 2 and pinned_f4.shape[1] == 4 and pinned_f4.flags.c_contiguous
        self._chk(self.L.lio_scan_upload_async(self.h, _ptr(pinned_f4), len(pinned_f4)), "lio_scan_upload_async")
        # The C side only stores the pointer: the copy is issued by the next measurement pass or by lio_scan_swap and may still
        # be in flight when lio_scan_swap returns.  Hold the array (a temporary from pinned_points would otherwise give its
        # page-locked memory back under the copy engine) until the upload AFTER the next one has been handed over.
        self._upload_keep = (getattr(self, "_upload_keep", (None, None))[1], pinned_f4)

    def scan_swap(self):
        """The scan uploaded with scan_upload_async becomes the current scan."""
        self._chk(self.L.lio_scan_swap(self.h), "lio_scan_swap")

    def measure(self, state14, do_search: bool, extrinsic_est_en: bool) -> dict:
        s = np.ascontiguousarray(state14, dtype=np.float64)
        assert s.shape == (14,)
        o = lio_out()
        self._chk(self.L.lio_measure(self.h, _ptr(s), int(do_search), int(extrinsic_est_en), C.byref(o)), "lio_measure")
        return {"m": o.m, "valid": bool(o.valid), "n_unresolved": o.n_unresolved, "res_mean": o.res_mean,
                "HTH": np.array(o.HTH, dtype=np.float64).reshape(12, 12), "HTh": np.array(o.HTh, dtype=np.float64),
                "n_candidates": int(o.n_candidates)}

    def measure_rows(self, m: int, col_major: bool = False):
        h_x = np.zeros((12, max(m, 1)) if col_major else (max(m, 1), 12), np.float64)
        h = np.zeros(max(m, 1), np.float64)
        self._chk(self.L.lio_measure_rows(self.h, _ptr(h_x), _ptr(h), m, int(col_major)), "lio_measure_rows")
        if col_major:                                           # Eigen's layout, leading dimension m: as a [m, 12] view
            return h_x.reshape(-1)[:12 * m].reshape(12, m).T, h[:m]
        return h_x[:m], h[:m]

    def measure_want_rows(self, on: bool = True):
        """lio_measure_want_rows: the following passes stage h_x / h compacted on the device (the callback path)."""
        self._chk(self.L.lio_measure_want_rows(self.h, int(on)), "lio_measure_want_rows")

    def neighbors(self):
        n = self.L.lio_scan_size(self.h)
        xyz = np.full((n, 5, 3), np.nan, np.float32)
        d = np.full((n, 5), np.inf, np.float32)
        cnt = np.zeros(n, np.int32)
        self._chk(self.L.lio_get_neighbors(self.h, _ptr(xyz), _ptr(d), _ptr(cnt)), "lio_get_neighbors")
        return xyz, d, cnt

    def scan_state(self):
        n = self.L.lio_scan_size(self.h)
        world = np.zeros((n, 3), np.float32)
        sel = np.zeros(n, np.uint8)
        nv = np.zeros((n, 4), np.float32)
        self._chk(self.L.lio_get_scan_state(self.h, _ptr(world), _ptr(sel), _ptr(nv)), "lio_get_scan_state")
        return world, sel, nv

    # ---- timing ------------------------------------------------------------------------------------
    def profile_enable(self, on=True, only=None):
        """Time kernel families with HIP events.  only=[PROF_KNN, ...] restricts it to those families."""
        if only is not None:
            mask = 0x100
            for w in only:
                mask |= 1 << w
            self.L.lio_profile_enable(self.h, mask)
        else:
            self.L.lio_profile_enable(self.h, int(bool(on)))

    def census_enable(self, on: bool = True):
        """Make search passes report n_candidates (sum of C(q)); off in timed runs."""
        self.L.lio_census_enable(self.h, int(on))

    def debug_set(self, key: str, value: int):
        """Test hooks of the library (lio_debug_set)."""
        self._chk(self.L.lio_debug_set(self.h, key.encode(), int(value)), "lio_debug_set")

    def debug_get(self, key):
        v = C.c_longlong(0)
        self._chk(self.L.lio_debug_get(self.h, key.encode(), C.byref(v)), "lio_debug_get")
        return int(v.value)

    def profile_reset(self):
        self.L.lio_profile_reset(self.h)

    def profile(self) -> dict:
        out = {}
        for w, name in PROF_NAMES.items():
            ms, n = C.c_double(0), C.c_longlong(0)
            self.L.lio_profile_get(self.h, w, C.byref(ms), C.byref(n))
            out[name] = {"total_ms": ms.value, "launches": n.value}
        return out

    def synchronize(self):
        self._chk(self.L.lio_synchronize(self.h), "lio_synchronize")


class Esekf:
    """Host-side iterated ESKF (lio_eskf_*): mirror of esekfom::esekf around the GPU path."""

    def __init__(self, maximum_iter: int = 3, limit=None):
        self.L = load()
        self.h = _vp(self.L.lio_eskf_create())
        self.maximum_iter = maximum_iter
        lim = np.full(23, 0.001) if limit is None else np.ascontiguousarray(limit, np.float64)   # laserMapping.cpp:2148
        rc = self.L.lio_eskf_init_dyn_share(self.h, maximum_iter, _ptr(lim))
        if rc != LIO_OK:
            raise LioError("lio_eskf_init_dyn_share failed")

    def __del__(self):
        try:
            if self.h:
                self.L.lio_eskf_destroy(self.h)
                self.h = None
        except Exception:
            pass

    def change_x(self, x26):
        x = np.ascontiguousarray(x26, np.float64)
        assert x.shape == (26,)
        self.L.lio_eskf_change_x(self.h, _ptr(x))

    def change_P(self, P):
        p = np.ascontiguousarray(P, np.float64)
        assert p.shape == (23, 23)
        self.L.lio_eskf_change_P(self.h, _ptr(p))

    def get_x(self) -> np.ndarray:
        x = np.zeros(26)
        self.L.lio_eskf_get_x(self.h, _ptr(x))
        return x

    def get_P(self) -> np.ndarray:
        p = np.zeros((23, 23))
        self.L.lio_eskf_get_P(self.h, _ptr(p))
        return p

    def predict(self, dt: float, Q, acc, gyro):
        """esekf::predict (esekfom.hpp:280-383): propagate x and P over dt with one IMU input."""
        q = np.ascontiguousarray(Q, np.float64)
        a = np.ascontiguousarray(acc, np.float64)
        g = np.ascontiguousarray(gyro, np.float64)
        assert q.shape == (12, 12) and a.shape == (3,) and g.shape == (3,)
        rc = self.L.lio_eskf_predict(self.h, float(dt), _ptr(q), _ptr(a), _ptr(g))
        if rc != LIO_OK:
            raise LioError("lio_eskf_predict failed")

    def update_iterated_dyn_share_modified(self, ctx: LioContext, R: float = 0.001, extrinsic_est_en: bool = False) -> dict:
        log = lio_update_log()
        rc = self.L.lio_eskf_update_iterated_dyn_share_modified(self.h, ctx.h, R, int(extrinsic_est_en), C.byref(log))
        if rc != LIO_OK:
            raise LioError(f"update_iterated_dyn_share_modified: {ERRORS.get(rc, rc)}: {ctx.L.lio_last_error(ctx.h).decode()}")
        return {k: getattr(log, k) for k, _ in lio_update_log._fields_}


class ImuProcess:
    """Host-side IMU front end (lio_imu_*): mirror of the reference's ImuProcess (src/IMU_Processing.hpp).
    IMU samples are rows [stamp, ax, ay, az, gx, gy, gz]."""

    def __init__(self):
        self.L = load()
        self.h = _vp(self.L.lio_imu_create())

    def __del__(self):
        try:
            if self.h:
                self.L.lio_imu_destroy(self.h)
                self.h = None
        except Exception:
            pass

    def reset(self):
        self.L.lio_imu_reset(self.h)

    def set_extrinsic(self, T, R):
        t = np.ascontiguousarray(T, np.float64).reshape(3)
        r = np.ascontiguousarray(R, np.float64).reshape(9)
        self.L.lio_imu_set_extrinsic(self.h, _ptr(t), _ptr(r))

    def set_cov(self, gyr_cov=None, acc_cov=None, b_gyr_cov=None, b_acc_cov=None):
        a = [None if v is None else np.ascontiguousarray(np.broadcast_to(np.asarray(v, np.float64), (3,))) for v in
             (gyr_cov, acc_cov, b_gyr_cov, b_acc_cov)]
        self.L.lio_imu_set_cov(self.h, *[None if v is None else _ptr(v) for v in a])

    def process(self, kf: "Esekf", ctx, lidar_beg_time: float, lidar_end_time: float, imu, raw_xyzt=None) -> bool:
        """ImuProcess::Process.  ctx=None: propagate only.  Returns True when ctx holds feats_undistort."""
        m = np.ascontiguousarray(imu, np.float64).reshape(-1, 7)
        r = None if raw_xyzt is None else np.ascontiguousarray(raw_xyzt, np.float32).reshape(-1, 4)
        if ctx is not None and r is not None:
            ctx._raw_keep = r                      # page-locked records are read in place after the call (see scan_undistort)
        und = C.c_int(0)
        rc = self.L.lio_imu_process(self.h, kf.h, None if ctx is None else ctx.h, lidar_beg_time, lidar_end_time,
                                    None if r is None else _ptr(r), 0 if r is None else len(r), 16, 12,
                                    _ptr(m) if len(m) else None, len(m), C.byref(und))
        if rc != LIO_OK:
            msg = ctx.L.lio_last_error(ctx.h).decode() if ctx is not None else ""
            raise LioError(f"lio_imu_process: {ERRORS.get(rc, rc)}: {msg}")
        return bool(und.value)

    def poses(self) -> np.ndarray:
        """IMUpose as rows of 22 doubles: offset_time, acc3, gyr3, vel3, pos3, rot9."""
        n = C.c_int(0)
        self.L.lio_imu_get_poses(self.h, None, 0, C.byref(n))
        out = np.zeros((n.value, 22))
        if n.value:
            self.L.lio_imu_get_poses(self.h, _ptr(out), n.value, C.byref(n))
        return out

    def info(self) -> dict:
        i = lio_imu_info()
        self.L.lio_imu_get_info(self.h, C.byref(i))
        return {"need_init": bool(i.need_init), "init_iter_num": i.init_iter_num, "mean_acc": np.array(i.mean_acc[:]),
                "mean_gyr": np.array(i.mean_gyr[:]), "cov_acc": np.array(i.cov_acc[:]), "cov_gyr": np.array(i.cov_gyr[:]),
                "last_lidar_end_time": i.last_lidar_end_time}


def process_noise_cov() -> np.ndarray:
    q = np.zeros((12, 12))
    load().lio_process_noise_cov(_ptr(q))
    return q


def scan_step_device(kf: "Esekf", ctx: "LioContext", x26, P, d_body_ptr: int, n: int, R: float = 0.001,
                     extrinsic_est_en: bool = False) -> dict:
    """lio_eskf_scan_step_device: prior in, device-resident scan in, iterated update -- one call."""
    x = None if x26 is None else np.ascontiguousarray(x26, np.float64)
    p = None if P is None else np.ascontiguousarray(P, np.float64)
    log = lio_update_log()
    rc = kf.L.lio_eskf_scan_step_device(kf.h, ctx.h, None if x is None else _ptr(x), None if p is None else _ptr(p),
                                        _vp(d_body_ptr), n, R, int(extrinsic_est_en), C.byref(log))
    if rc != LIO_OK:
        raise LioError(f"lio_eskf_scan_step_device: {ERRORS.get(rc, rc)}: {ctx.L.lio_last_error(ctx.h).decode()}")
    return {k: getattr(log, k) for k, _ in lio_update_log._fields_}


def scan_step_pipelined(kf: "Esekf", ctx: "LioContext", x26, P, next_pinned_f4, R: float = 0.001, extrinsic_est_en: bool = False) -> dict:
    """lio_eskf_scan_step_pipelined: the scan uploaded during the previous call becomes current, the copy of `next_pinned_f4`
    ([n, 4] float32 from pinned_points, or None) travels during this one.  The caller keeps `next_pinned_f4` alive until the
    call after the next has returned."""
    x = None if x26 is None else np.ascontiguousarray(x26, np.float64)
    p = None if P is None else np.ascontiguousarray(P, np.float64)
    log = lio_update_log()
    rc = kf.L.lio_eskf_scan_step_pipelined(kf.h, ctx.h, None if x is None else _ptr(x), None if p is None else _ptr(p),
                                           None if next_pinned_f4 is None else _ptr(next_pinned_f4),
                                           0 if next_pinned_f4 is None else len(next_pinned_f4), R, int(extrinsic_est_en), C.byref(log))
    if rc != LIO_OK:
        raise LioError(f"lio_eskf_scan_step_pipelined: {ERRORS.get(rc, rc)}: {ctx.L.lio_last_error(ctx.h).decode()}")
    return {k: getattr(log, k) for k, _ in lio_update_log._fields_}


def make_state26(pos, rot, off_R, off_T, vel=(0, 0, 0), bg=(0, 0, 0), ba=(0, 0, 0), grav=(0, 0, -9.809)) -> np.ndarray:
    """x26 = pos3, rot xyzw, offR xyzw, offT3, vel3, bg3, ba3, grav3 (use-ikfom.hpp:21-30)."""
    return np.concatenate([pos, rot, off_R, off_T, vel, bg, ba, grav]).astype(np.float64)


class DeviceBuffer:
    """A plain hipMalloc'ed buffer filled from a numpy array (torch-free way to keep a scan resident
    in HBM for lio_scan_set_device / lio_map_build_device)."""

    def __init__(self, host: np.ndarray):
        load()
        self.hip = C.CDLL("libamdhip64.so.7")          # SONAME of the runtime that is already mapped
        self.hip.hipMalloc.argtypes = [C.POINTER(_vp), C.c_size_t]
        self.hip.hipMemcpy.argtypes = [_vp, _vp, C.c_size_t, C.c_int]
        self.hip.hipFree.argtypes = [_vp]
        host = np.ascontiguousarray(host)
        self.nbytes = host.nbytes
        p = _vp()
        if self.hip.hipMalloc(C.byref(p), max(self.nbytes, 16)) != 0:
            raise LioError("hipMalloc failed")
        self.ptr = p.value
        if self.nbytes and self.hip.hipMemcpy(_vp(self.ptr), host.ctypes.data_as(_vp), self.nbytes, 1) != 0:   # H2D
            raise LioError("hipMemcpy failed")

    def __del__(self):
        try:
            if getattr(self, "ptr", None):
                self.hip.hipFree(_vp(self.ptr))
                self.ptr = None
        except Exception:
            pass


def scan_to_device(body_xyz: np.ndarray) -> "DeviceBuffer":
    """float4 records (x, y, z, 0) of a body-frame scan, resident in HBM."""
    b4 = np.zeros((len(body_xyz), 4), np.float32)
    b4[:, :3] = body_xyz
    return DeviceBuffer(b4)
