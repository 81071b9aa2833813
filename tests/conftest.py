"""Shared fixtures.  `-m "not gpu"` runs on any host; `-m gpu` needs one MI355X and calls the
product only through the C ABI of liblio_hip.so (better_fastlio2_amd.capi)."""
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    from oracle import pyoracle
    pyoracle.build()
    return pyoracle


@pytest.fixture(scope="session")
def synth():
    from better_fastlio2_amd import synth as s
    return s


@pytest.fixture(scope="session")
def small_case(synth):
    """200 k-point world, 20 k-point KITTI-64-like scan, perturbed pose."""
    world = synth.make_world(200_000, 0.2)
    scan = synth.make_scan(world, 20_000)
    pos, rot = synth.perturb_state(scan)
    return {"world": world, "scan": scan, "pos": pos, "rot": rot,
            "state14": synth.state_vector(pos, rot, scan.off_R, scan.off_T)}


@pytest.fixture(scope="session")
def hap_case(synth):
    """Livox-HAP-like narrow field of view, extrinsic estimation on (config 3, scaled down)."""
    world = synth.make_world(150_000, 0.2, seed=synth.SEED + 7)
    scan = synth.make_scan(world, 6_000, kind="hap", pos=(-4.0, 6.5, 0.0), rpy=(0.01, 0.02, -1.1))
    pos, rot = synth.perturb_state(scan, seed=synth.SEED + 9)
    return {"world": world, "scan": scan, "pos": pos, "rot": rot,
            "state14": synth.state_vector(pos, rot, scan.off_R, scan.off_T)}


@pytest.fixture(scope="session")
def host_harness():
    """tests/csrc/host_harness.cpp: g++ build of the kernels' shared arithmetic + the host filter."""
    import ctypes as C
    src = os.path.join(ROOT, "tests", "csrc", "host_harness.cpp")
    so = os.path.join(ROOT, "tests", "csrc", "libhost_harness.so")
    deps = [src] + [os.path.join(ROOT, "better_fastlio2_amd", "csrc", f)
                    for f in ("lio_math.h", "lio_common.h", "lio_esekf.cpp", "lio_esekf.hpp")]
    if not os.path.exists(so) or os.path.getmtime(so) < max(os.path.getmtime(d) for d in deps):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-ffp-contract=off", "-shared", "-o", so, src,
                               os.path.join(ROOT, "better_fastlio2_amd", "csrc", "lio_esekf.cpp")])
    H = C.CDLL(so)
    f32 = np.ctypeslib.ndpointer(np.float32, flags="C")
    f64 = np.ctypeslib.ndpointer(np.float64, flags="C")
    u64 = np.ctypeslib.ndpointer(np.uint64, flags="C")
    u32 = np.ctypeslib.ndpointer(np.uint32, flags="C")
    H.hh_plane_fit.argtypes = [f32, C.c_float, f32]
    H.hh_plane_fit.restype = C.c_int
    H.hh_body_to_world.argtypes = [f64, f32, C.c_int, f32]
    H.hh_point.argtypes = [f64, f32, f32, f32, C.c_int, f32, f64, f64]
    H.hh_point.restype = C.c_int
    H.hh_keys.argtypes = [f32, C.c_int, C.c_float, u64, u64, u32]
    H.hh_morton_decode.argtypes = [u64, C.c_int, u32]
    H.hh_sqdist.argtypes = [f32, f32]
    H.hh_sqdist.restype = C.c_float
    H.hh_boxplus.argtypes = [f64, f64]
    H.hh_boxminus.argtypes = [f64, f64, f64]
    H.hh_hth_rows.argtypes = [f64, f64, C.c_int, f64, f64, C.c_int]
    H.hh_hth_rows.restype = C.c_double
    return H


def bits_equal(a, b) -> bool:
    a = np.ascontiguousarray(a)
    b = np.ascontiguousarray(b)
    if a.shape != b.shape or a.dtype != b.dtype:
        return False
    return np.array_equal(a.view(np.uint8), b.view(np.uint8))


@pytest.fixture(scope="session")
def gpu_capi():
    """The product binding; GPU tests fail (not skip) when the library or the device is missing."""
    from better_fastlio2_amd import capi
    capi.load()
    return capi
