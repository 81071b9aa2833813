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


@pytest.mark.parametrize("n_raw", [9000, 40000, 131072])
def test_overlapped_scan_preparation_equals_the_serial_loop(gpu_capi, synth, small_case, n_raw):
    """lio_scan_prefetch_raw + lio_scan_prep_begin / _end: sweep k + 1 travels on the copy stream, is compensated and voxel-filtered on
    a stream of its own (own stamped scratch, own mail box, centroids into the back scan buffer) WHILE map_incremental of scan k runs
    on the main stream -- the same filtered clouds, measurement passes, insert counts and final map, bit for bit, as the serial loop
    (laserMapping.cpp:2256-2401 in its own order).  Sweeps in time order (compact-key LSD sort on the preparation stream), one with a
    swapped pair (the prediction fails: the filter is redone by lio_scan_prep_end), tiny sweeps (general path: filtered at the end)."""
    world = small_case["world"]
    half = world.ground_half
    rng = np.random.default_rng(77)
    poses = np.zeros((12, 22))
    poses[:, 0] = np.arange(12) * 0.01
    poses[:, 7] = 1.0
    poses[:, 10] = poses[:, 0] * 1.0
    poses[:, 13] = poses[:, 17] = poses[:, 21] = 1.0
    end = synth.state_vector(poses[-1, 10:13], [0, 0, 0, 1], [0, 0, 0, 1.0], np.zeros(3))
    sweeps = []
    for k in range(5):
        xyz = (rng.uniform(-0.6 * half, 0.6 * half, (n_raw, 3)) * [1, 1, 0.02]).astype(np.float32)
        t = np.sort(np.floor(rng.uniform(0, 100.0, n_raw) * 4).astype(np.float32) / 4)
        if k == 3:
            t[n_raw // 2], t[n_raw // 2 + 1] = t[n_raw // 2 + 1] + 1, t[n_raw // 2]      # one out-of-order pair
        sweeps.append(gpu_capi.pinned_points(np.concatenate([xyz, t[:, None]], 1).astype(np.float32)) if k != 2
                      else np.concatenate([xyz, t[:, None]], 1).astype(np.float32))            # (one sweep in pageable memory: no prefetch)
    leaf = 0.5 if n_raw > 20000 else 0.1
    states = [synth.state_vector([0.05 * k, 0.02, 0.0], [0, 0, 0, 1], np.array([0, 0, 0, 1.0]), np.zeros(3)) for k in range(5)]

    def serial():
        ctx = gpu_capi.LioContext(0.2, max_scan_points=131072, max_map_points=600_000)
        ctx.map_build(world.points)
        out = []
        for k in range(5):
            ctx.scan_undistort(sweeps[k], poses, end)
            m = ctx.scan_voxel_filter(leaf)
            down = ctx.scan_cloud(1).copy()
            g = ctx.measure(states[k], True, False)
            r = ctx.map_incremental(states[k], 0.2, True, apply=True)
            out.append((m, down, g["m"], g["HTH"].copy(), r["n_add"], r["n_nodown"], r["add_counter"], ctx.map_validnum()))
        flat = ctx.map_flatten().copy()
        ctx.close()
        return out, flat

    def overlapped():
        ctx = gpu_capi.LioContext(0.2, max_scan_points=131072, max_map_points=600_000)
        ctx.map_build(world.points)
        out = []
        ctx.scan_prefetch_raw(sweeps[0])
        ctx.scan_prep_begin()
        ctx.scan_undistort(sweeps[0], poses, end)
        assert ctx.scan_voxel_filter(leaf) == -1
        m = ctx.scan_prep_end()
        for k in range(5):
            down = ctx.scan_cloud(1).copy()
            g = ctx.measure(states[k], True, False)
            if k + 1 < 5:
                ctx.scan_prefetch_raw(sweeps[k + 1])                            # (sweep 2 lives in pageable memory: the hint does nothing)
                ctx.scan_prep_begin()
                ctx.scan_undistort(sweeps[k + 1], poses, end)
                assert ctx.scan_voxel_filter(leaf) == -1
            r = ctx.map_incremental(states[k], 0.2, True, apply=True)          # ... beside the preparation of the next sweep
            out.append((m, down, g["m"], g["HTH"].copy(), r["n_add"], r["n_nodown"], r["add_counter"], ctx.map_validnum()))
            if k + 1 < 5:
                m = ctx.scan_prep_end()
        flat = ctx.map_flatten().copy()
        stats = (ctx.debug_get("raw_prefetch_hits"), ctx.debug_get("prep_overlapped"), ctx.debug_get("voxel_redone"))
        ctx.close()
        return out, flat, stats

    a, fa = serial()
    b, fb, stats = overlapped()
    for k, (x, y) in enumerate(zip(a, b)):
        assert x[0] == y[0] and x[2] == y[2] and x[4:] == y[4:], k
        assert bits_equal(x[1], y[1]) and np.array_equal(x[3], y[3]), k
    assert fa.shape == fb.shape and bits_equal(fa, fb)
    assert stats[0] == 4                                             # every page-locked sweep was taken from the prefetch buffer
    if n_raw > 16384:
        assert stats[1] >= 3 and stats[2] >= 1                       # filtered beside the map update; the swapped pair was redone at the end
    else:
        assert stats[1] == 0                                         # tiny sweeps: the general path, filtered by lio_scan_prep_end


def test_scan_preparation_calls_out_of_order(gpu_capi, synth, small_case):
    """lio_scan_prep_begin / _end out of order, the back scan buffer already promised to lio_scan_upload_async, a preparation that only
    uploads, a first sweep taken from the prefetch buffer on a context that has never staged anything: errors are reported, nothing hangs,
    the context goes on."""
    world = small_case["world"]
    ctx = gpu_capi.LioContext(0.2, max_scan_points=65536, max_map_points=400_000)
    dev = gpu_capi.scan_to_device(world.points[:100_000])
    ctx.map_build_device(dev.ptr, 100_000)                             # (no host upload: the staging buffer does not exist yet)
    rng = np.random.default_rng(3)
    xyz = (rng.uniform(-20, 20, (30_000, 3)) * [1, 1, 0.02]).astype(np.float32)
    t = np.sort(rng.uniform(0, 100.0, 30_000)).astype(np.float32)
    sweep = gpu_capi.pinned_points(np.concatenate([xyz, t[:, None]], 1))
    poses = np.zeros((12, 22))
    poses[:, 0] = np.arange(12) * 0.01
    poses[:, 7] = 1.0
    poses[:, 10] = poses[:, 0] * 1.0
    poses[:, 13] = poses[:, 17] = poses[:, 21] = 1.0
    end = synth.state_vector(poses[-1, 10:13], [0, 0, 0, 1], [0, 0, 0, 1.0], np.zeros(3))
    with pytest.raises(gpu_capi.LioError, match="no preparation is running"):
        ctx.scan_prep_end()
    ctx.scan_prefetch_raw(sweep)
    ctx.scan_prep_begin()
    with pytest.raises(gpu_capi.LioError, match="already running"):
        ctx.scan_prep_begin()
    with pytest.raises(gpu_capi.LioError, match="running preparation"):
        ctx.scan_upload_async(gpu_capi.pinned_points(xyz[:100]))
    ctx.scan_undistort(sweep, poses, end)                              # from the prefetch buffer
    assert ctx.scan_voxel_filter(0.5) == -1
    with pytest.raises(gpu_capi.LioError, match="one filter per"):
        ctx.scan_voxel_filter(0.5)
    n = ctx.scan_prep_end()
    assert ctx.debug_get("raw_prefetch_hits") == 1 and 1000 < n < 30_000
    down = ctx.scan_cloud(1).copy()
    ctx.scan_undistort(sweep, poses, end)                              # the same sweep, serial calls
    assert ctx.scan_voxel_filter(0.5) == n and bits_equal(ctx.scan_cloud(1), down)
    ctx.scan_prep_begin()                                              # a preparation without a filter: only the upload + compensation run there
    ctx.scan_undistort(sweep, poses, end)
    assert ctx.scan_prep_end() == 0
    assert ctx.scan_voxel_filter(0.5) == n and bits_equal(ctx.scan_cloud(1), down)
    ctx.scan_upload_async(gpu_capi.pinned_points(xyz[:100]))
    with pytest.raises(gpu_capi.LioError, match="lio_scan_upload_async"):
        ctx.scan_prep_begin()
    ctx.scan_swap()
    ctx.close()
