"""The C-ABI boundary: liblio_hip.so builds/loads without a GPU and exports every symbol that
include/lio_hip.h declares; argument checking and the no-CPU-fallback rule."""
import ctypes as C
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    text = open(os.path.join(ROOT, "include", "lio_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(lio_[A-Za-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol(gpu_capi):
    L = gpu_capi.load()
    syms = header_symbols()
    assert len(syms) >= 35
    missing = [s for s in syms if not hasattr(L, s)]
    assert not missing, f"declared in include/lio_hip.h but not exported: {missing}"
    # and the binding's list is the same set
    assert sorted(gpu_capi.EXPORTS) == syms


def test_version_and_struct_layout(gpu_capi):
    L = gpu_capi.load()
    assert b"gfx950" in L.lio_version()
    assert C.sizeof(gpu_capi.lio_out) == 4 * 4 + 8 + 144 * 8 + 12 * 8 + 8
    assert C.sizeof(gpu_capi.lio_config) == 32


def test_invalid_arguments_are_rejected(gpu_capi):
    L = gpu_capi.load()
    h = C.c_void_p()
    assert L.lio_create(None, C.byref(h)) == -1
    bad = gpu_capi.lio_config(0, -1.0, 0.0, 1000, 1000, 0)
    rc = L.lio_create(C.byref(bad), C.byref(h))
    assert rc in (-1, -2)          # invalid arg, or no device on a CPU-only host (checked first)
    assert not h.value
    x = np.zeros(26); dx = np.zeros(23)
    assert L.lio_state_boxplus(None, dx.ctypes.data_as(C.c_void_p)) == -1
    assert L.lio_eskf_init_dyn_share(None, 3, None) == -1


def test_no_cpu_fallback(gpu_capi):
    """Without a GPU the product refuses to run (LIO_ERR_NO_DEVICE); it never computes on the host."""
    if os.path.exists("/dev/kfd"):
        pytest.skip("GPU present: the refusal path is exercised on CPU-only hosts")
    with pytest.raises(gpu_capi.LioError, match="LIO_ERR_NO_DEVICE"):
        gpu_capi.LioContext()


def test_product_does_not_touch_the_oracle():
    """Nothing under better_fastlio2_amd/, include/ may reference oracle/ (parity rule)."""
    for base in ("better_fastlio2_amd", "include"):
        for dp, _, files in os.walk(os.path.join(ROOT, base)):
            for f in files:
                if f.endswith((".py", ".h", ".hpp", ".cpp", ".hip")):
                    text = open(os.path.join(dp, f), errors="ignore").read()
                    assert "pyoracle" not in text and "lio_oracle" not in text and "libikd_ref" not in text, f


def test_kitti_bin_reader(tmp_path):
    """KITTI velodyne .bin sweeps (common_lib.h:594-614 readBin): x, y, z, intensity fp32 records; the reference's loop
    appends one all-zero point after the failed read at end of file."""
    import numpy as np
    from better_fastlio2_amd import kitti
    rng = np.random.default_rng(2)
    xyz = rng.normal(0, 20, (1000, 3)).astype(np.float32)
    inten = rng.uniform(0, 1, 1000).astype(np.float32)
    p = str(tmp_path / "000000.bin")
    kitti.write_bin(p, xyz, inten)
    got = kitti.read_bin(p)
    assert got.shape == (1000, 4) and np.array_equal(got[:, :3], xyz) and np.array_equal(got[:, 3], inten)
    quirk = kitti.read_bin(p, reference_eof_quirk=True)
    assert quirk.shape == (1001, 4) and not quirk[-1].any()
    with open(p, "ab") as fh:
        fh.write(b"\x00\x01\x02")                                # a truncated record is dropped
    assert kitti.read_bin(p).shape == (1000, 4)
