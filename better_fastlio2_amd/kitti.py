"""KITTI velodyne `.bin` sweeps (x, y, z, intensity as little-endian fp32 records) for BASELINE.json configs[0]
("KITTI seq00"): the reader the reference uses is readBin (include/common_lib.h:594-614).

No data set ships with this repository (no network in the build image): `scripts/kitti_bin_step.py` runs the day a file
is supplied and is exercised in the tests on a synthetic sweep written in the same format.
"""
from __future__ import annotations

import numpy as np


def read_bin(path: str, reference_eof_quirk: bool = False) -> np.ndarray:
    """[n, 4] float32: x, y, z, intensity.  A trailing partial record is dropped.
    reference_eof_quirk: readBin loops `while (input.good() && !input.eof())` and pushes the point after a FAILED read, so
    the cloud it builds ends with one extra all-zero point (common_lib.h:602-612); True reproduces that."""
    raw = np.fromfile(path, dtype="<f4")
    pts = raw[: (raw.size // 4) * 4].reshape(-1, 4).astype(np.float32, copy=False)
    if reference_eof_quirk:
        pts = np.concatenate([pts, np.zeros((1, 4), np.float32)], 0)
    return np.ascontiguousarray(pts)


def write_bin(path: str, xyz: np.ndarray, intensity: np.ndarray | None = None) -> None:
    """The inverse (tests, synthetic sweeps)."""
    xyz = np.asarray(xyz, np.float32)
    out = np.zeros((len(xyz), 4), "<f4")
    out[:, :3] = xyz[:, :3]
    if intensity is not None:
        out[:, 3] = intensity
    out.tofile(path)
