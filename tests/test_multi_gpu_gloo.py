"""The N > 1 path on CPU: world_size-2 `gloo` process groups exercise (a) the insert-list exchange
that keeps map replicas identical and (b) bench.py's barrier + max-over-ranks timing."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class _OracleReplica:
    """Stands in for a GPU replica in the CPU test: same map_add contract (Add_Points semantics)."""

    def __init__(self, base):
        from oracle import pyoracle as po
        self.m = po.OracleMap(0.2)
        self.m.build(base)

    def map_add(self, pts, downsample_on):
        return self.m.add(pts, downsample_on)

    def map_delete_boxes(self, boxes):
        return self.m.delete_boxes(boxes)


def _worker(rank, world, port, tmpdir):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from better_fastlio2_amd import replicate, synth
        world_pts = synth.make_world(20_000, 0.2, seed=4242).points          # identical replica everywhere
        rng = np.random.default_rng(100 + rank)                              # different scan stream per rank
        n = 300 + 150 * rank
        mine = (world_pts[rng.choice(len(world_pts), n, replace=False)] + rng.normal(0, 0.05, (n, 3))).astype(np.float32)
        lists = replicate.exchange_inserts(mine, capacity=1000)
        assert len(lists) == world and np.array_equal(lists[rank], mine)
        assert [len(x) for x in lists] == [300 + 150 * r for r in range(world)]
        boxes = replicate.exchange_boxes(np.array([[rank, 0, 0, rank + 1, 1, 1]], np.float32))
        assert boxes.shape == (world, 6) and boxes[1, 0] == 1.0
        rep = _OracleReplica(world_pts)
        added = replicate.apply_in_rank_order(rep, lists)
        live = rep.m.flatten()
        live = live[np.lexsort((live[:, 2], live[:, 1], live[:, 0]))]
        np.save(os.path.join(tmpdir, f"live_{rank}.npy"), live)
        np.save(os.path.join(tmpdir, f"added_{rank}.npy"), np.array([added]))

        # bench.py's timing contract: barrier, K steps, barrier, MAX over ranks
        import bench
        import time

        def step():
            time.sleep(0.002 * (rank + 1))                                   # rank 1 is the slow one

        class _NoCuda:                                                       # CPU stand-in for torch.cuda.synchronize
            pass
        torch.cuda.synchronize = lambda *a, **k: None
        el = bench.timed_steps(step, steps=5, warmup=1, dist=dist, device=torch.device("cpu"))
        np.save(os.path.join(tmpdir, f"el_{rank}.npy"), np.array([el]))
    finally:
        dist.destroy_process_group()


def test_replicas_stay_identical_and_timing_is_max_over_ranks(tmp_path):
    world = 2
    port = 29600 + (os.getpid() % 300)
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    a, b = np.load(tmp_path / "live_0.npy"), np.load(tmp_path / "live_1.npy")
    assert a.shape == b.shape and np.array_equal(a.view(np.uint32), b.view(np.uint32))     # bit-identical replicas
    assert np.load(tmp_path / "added_0.npy")[0] == np.load(tmp_path / "added_1.npy")[0] > 0
    e0, e1 = np.load(tmp_path / "el_0.npy")[0], np.load(tmp_path / "el_1.npy")[0]
    assert e0 == e1 and e0 >= 5 * 0.004                                                      # the slow rank's time


def test_exchange_without_process_group_is_identity():
    from better_fastlio2_amd import replicate
    pts = np.random.default_rng(0).normal(size=(10, 3)).astype(np.float32)
    out = replicate.exchange_inserts(pts, capacity=16)
    assert len(out) == 1 and np.array_equal(out[0], pts)


def _stream_lists(world_pts, rank, k):
    """What rank `rank` wants inserted after its scan step k (a function of rank and step only: every process can replay it)."""
    rng = np.random.default_rng(1000 * k + rank)
    n = 200 + 50 * rank + 30 * k
    add = (world_pts[rng.choice(len(world_pts), n, replace=False)] + rng.normal(0, 0.05, (n, 3))).astype(np.float32)
    nodown = (world_pts[rng.choice(len(world_pts), 40, replace=False)] + rng.normal(0, 0.3, (40, 3))).astype(np.float32)
    boxes = np.array([[-3.0 + rank, -3.0, -1.0, -1.0 + rank, 3.0, 2.0]], np.float32) if (k == 1 and rank == 1) else np.zeros((0, 6), np.float32)
    return add, nodown, boxes


def _deferred_worker(rank, world, port, tmpdir):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from better_fastlio2_amd import replicate, synth
        world_pts = synth.make_world(20_000, 0.2, seed=4242).points
        rep = _OracleReplica(world_pts)
        ex = replicate.DeferredExchange(capacity=1000)
        sizes = []
        for k in range(3):
            # (scan step k would search `rep` here: the batch gathered by exchange k - 1 is NOT in it yet)
            sizes.append(len(rep.m.flatten()))
            add, nodown, boxes = _stream_lists(world_pts, rank, k)
            ex.step(rep, add, nodown, boxes)
        ex.flush(rep)
        live = rep.m.flatten()
        np.save(os.path.join(tmpdir, f"dlive_{rank}.npy"), live[np.lexsort((live[:, 2], live[:, 1], live[:, 0]))])
        np.save(os.path.join(tmpdir, f"dsizes_{rank}.npy"), np.array(sizes))
    finally:
        dist.destroy_process_group()


def test_deferred_schedule_keeps_replicas_identical(tmp_path):
    """lio_replicate_defer's ORDER on two gloo ranks (replicate.DeferredExchange): the batch gathered by exchange k enters at the
    start of exchange k + 1, boxes first, rank order inside each list.  Replicas bit-identical, the map a scan step searches lags
    one exchange, and the end state equals a single-process replay of the same calls."""
    world = 2
    port = 29900 + (os.getpid() % 300)
    mp.spawn(_deferred_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    a, b = np.load(tmp_path / "dlive_0.npy"), np.load(tmp_path / "dlive_1.npy")
    assert a.shape == b.shape and np.array_equal(a.view(np.uint32), b.view(np.uint32))
    s0, s1 = np.load(tmp_path / "dsizes_0.npy"), np.load(tmp_path / "dsizes_1.npy")
    assert np.array_equal(s0, s1) and s0[0] == s0[1] and s0[2] != s0[1]        # exchange 0's lists are in the map at step 2, not at step 1
    sys.path.insert(0, ROOT)
    from better_fastlio2_amd import synth
    world_pts = synth.make_world(20_000, 0.2, seed=4242).points
    ref = _OracleReplica(world_pts)
    for k in range(3):
        per_rank = [_stream_lists(world_pts, r, k) for r in range(world)]
        boxes = np.concatenate([p[2] for p in per_rank])
        if len(boxes):
            ref.map_delete_boxes(boxes)
        for r in range(world):
            ref.map_add(per_rank[r][0], True)
        for r in range(world):
            ref.map_add(per_rank[r][1], False)
    live = ref.m.flatten()
    live = live[np.lexsort((live[:, 2], live[:, 1], live[:, 0]))]
    assert np.array_equal(a.view(np.uint32), live.view(np.uint32))
