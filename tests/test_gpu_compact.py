"""The single-launch sort / compaction kernels of map maintenance and scan preparation (csrc/lio_compact.hip) against
the library primitives they replace (rocPRIM, lio_debug_set("library_sort", 1)): same bits, same order, at one tile,
a few tiles and the largest raw sweep the single launch takes."""
import numpy as np
import pytest

from conftest import bits_equal

pytestmark = pytest.mark.gpu


def _drive(gpu_capi, synth, world, n_raw, mode, seed, leaf=None, sorted_times=False, fast_runs=None):
    rng = np.random.default_rng(seed)
    ctx = gpu_capi.LioContext(0.2, max_scan_points=131072, max_map_points=600_000)
    if mode:
        ctx.debug_set(mode, 1)
    ctx.map_build(world.points)
    half = world.ground_half
    out = []
    poses = np.zeros((12, 22))                                       # a straight, slow drive: time-keyed sweeps
    poses[:, 0] = np.arange(12) * 0.01
    poses[:, 7] = 1.0
    poses[:, 10] = poses[:, 0] * 1.0
    poses[:, 13] = poses[:, 17] = poses[:, 21] = 1.0
    st = None
    for k in range(3):
        # a sweep with many points per leaf, repeated time stamps (stable order) and a few non-finite points
        xyz = (rng.uniform(-0.6 * half, 0.6 * half, (n_raw, 3)) * [1, 1, 0.02]).astype(np.float32)
        xyz[::7] = xyz[1::7][: len(xyz[::7])] + np.float32(0.01)
        t = np.floor(rng.uniform(0, 100.0, n_raw) * 4).astype(np.float32) / 4
        if sorted_times and k != 2:
            t = np.sort(t)                                            # a driver that delivers its points in firing order
        elif sorted_times:
            t = np.sort(t); t[n_raw // 2], t[n_raw // 2 + 1] = t[n_raw // 2 + 1] + 1, t[n_raw // 2]   # ... but for one pair
        if n_raw > 100:
            xyz[5] = [np.nan, 0, 0]
        raw = np.concatenate([xyz, t[:, None]], 1).astype(np.float32)
        if k == 1:
            ctx.scan_undistort(raw)                                   # already compensated: leaf index is the whole key
        else:
            ctx.scan_undistort(raw, poses, synth.state_vector(poses[-1, 10:13], [0, 0, 0, 1], [0, 0, 0, 1.0], np.zeros(3)))
        m = ctx.scan_voxel_filter(leaf if leaf else (0.5 if n_raw > 20000 else 0.1))
        down = ctx.scan_cloud(1).copy()
        und = ctx.scan_cloud(0).copy()                                # time order (a second sort of the sweep)
        st = synth.state_vector([0.05 * k, 0.02, 0.0], [0, 0, 0, 1], np.array([0, 0, 0, 1.0]), np.zeros(3))
        g = ctx.measure(st, True, False)
        r = ctx.map_incremental(st, 0.2, True, apply=True)
        out.append((m, down, und, g["m"], r["n_add"], r["n_nodown"], r["add_counter"], ctx.map_validnum()))
    flat = ctx.map_flatten().copy()
    if fast_runs is not None:
        fast_runs.append((ctx.debug_get("voxel_fast_runs"), ctx.debug_get("voxel_redone")))
    ctx.close()
    return out, flat


def _same(a, fa, b, fb):
    for x, y in zip(a, b):
        assert x[0] == y[0] and x[3:] == y[3:]
        assert bits_equal(x[1], y[1]) and bits_equal(x[2], y[2])
    assert fa.shape == fb.shape and bits_equal(fa, fb)               # the maps, point for point IN ORDER


@pytest.mark.parametrize("n_raw", [60, 1024, 1025, 3000, 9000, 16384, 131072])
def test_single_launch_sort_and_compaction_equal_the_library(gpu_capi, synth, small_case, n_raw):
    """default (rank sort, single-launch compactions) == rocPRIM everywhere"""
    b, fb = _drive(gpu_capi, synth, small_case["world"], n_raw, "library_sort", 3)
    a, fa = _drive(gpu_capi, synth, small_case["world"], n_raw, None, 3)
    _same(a, fa, b, fb)
    assert b[0][0] > 0 and b[-1][4] + b[-1][5] > 0                   # something was filtered and inserted


@pytest.mark.parametrize("n_raw", [16385, 40000, 131072])
def test_time_ordered_sweeps_take_the_compact_key_sort(gpu_capi, synth, small_case, n_raw):
    """A sweep whose points come in time order is sorted by the leaf index alone (stable LSD splits, lio_gsort.hip): the same
    filtered cloud, bit for bit and in order, as the 64-bit (leaf | time) library sort -- with repeated time stamps (ties keep
    the input order), non-finite points (dropped) and many points per leaf.  The host predicts "in order" from the previous sweep
    and learns the truth with the result: the sweep with one out-of-order pair is filtered twice (compact keys, then the general
    path), and so is a sweep whose leaf index needs more 8-bit splits than the previous one's.  pcl::VoxelGrid semantics as
    restated in oracle/ (DESIGN.md section 3.5)."""
    runs = []
    b, fb = _drive(gpu_capi, synth, small_case["world"], n_raw, "library_sort", 5, sorted_times=True)
    a, fa = _drive(gpu_capi, synth, small_case["world"], n_raw, None, 5, sorted_times=True, fast_runs=runs)
    _same(a, fa, b, fb)
    assert runs == [(3, 1)]                                          # three sweeps tried with compact keys; the last one (a swapped pair) redone
    c, fc = _drive(gpu_capi, synth, small_case["world"], n_raw, "library_sort", 5, sorted_times=True, leaf=0.1)
    d, fd = _drive(gpu_capi, synth, small_case["world"], n_raw, None, 5, sorted_times=True, leaf=0.1)
    _same(d, fd, c, fc)                                              # a finer grid: more bits in the leaf index (more passes)
