"""The replica exchange behind the C ABI (lio_replicate_*): map_incremental lists and delete
boxes of every rank in one ncclAllGather on the library's stream, applied in rank order on every replica.

 - one rank, no communicator and one rank THROUGH RCCL (dlopen'd librccl, ncclCommInitRank with one rank): the exchange
   path must equal lio_map_incremental / lio_map_delete_boxes on a twin context and the reference ikd-Tree;
 - the overflow path (a message too small for the lists) repeats the exchange and gives the same map;
 - two ranks on two GPUs (skipped on a one-GPU box): replicas bit-identical and equal to the reference tree fed the
   rank-ordered lists.
"""
import os
import subprocess
import sys
import tempfile

import numpy as np
import pytest

from conftest import bits_equal

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _key(p):
    return p[np.lexsort((p[:, 2], p[:, 1], p[:, 0]))]


def _drive(synth, world, n, k, seed_off=0):
    sc = synth.make_scan(world, n, seed=synth.SEED + 31 + 17 * k + seed_off, pos=(2.0 + 0.9 * k, -1.5 + 0.4 * k, 0.0),
                         rpy=(0.004, -0.006, 0.3 + 0.05 * k))
    ps, rt = synth.perturb_state(sc, dpos=0.05, drot_deg=0.3, seed=synth.SEED + 5 + k)
    return sc, ps, rt


@pytest.mark.parametrize("mode", ["no_comm", "rccl_one_rank", "overflow"])
def test_exchange_equals_local_map_incremental(gpu_capi, oracle, synth, small_case, mode):
    w = small_case["world"]
    thin = np.ascontiguousarray(w.points[::2])            # a thinned map: scans really add points
    nq = 8000
    a = gpu_capi.LioContext(0.2, max_scan_points=nq, max_map_points=400_000)     # exchange path
    b = gpu_capi.LioContext(0.2, max_scan_points=nq, max_map_points=400_000)     # plain path
    a.map_build(thin); b.map_build(thin)
    ref = oracle.RefIkdTree(0.2) if oracle.ref_available() else None
    if ref is not None:
        ref.build(thin)
    uid = gpu_capi.replicate_unique_id() if mode != "no_comm" else b""
    a.replicate_init(uid, 0, 1, 64 if mode == "overflow" else 0)
    kfa, kfb = gpu_capi.Esekf(3), gpu_capi.Esekf(3)
    for k in range(4):
        sc, ps, rt = _drive(synth, w, nq, k)
        x0 = gpu_capi.make_state26(ps, rt, sc.off_R, sc.off_T)
        for ctx, kf in ((a, kfa), (b, kfb)):
            kf.change_x(x0); kf.change_P(synth.init_cov())
            ctx.scan_set(sc.body)
            kf.update_iterated_dyn_share_modified(ctx, 0.001, False)
        assert np.array_equal(kfa.get_x(), kfb.get_x())
        post = kfa.get_x()[:14]
        boxes = np.array([[-30.0 + 2 * k, -30.0, -1.0, -27.0 + 2 * k, 30.0, 5.0]], np.float32) if k % 2 == 1 else np.zeros((0, 6), np.float32)
        if len(boxes):
            a.replicate_queue_boxes(boxes)
        ra = a.replicate_step(post, 0.2, True)
        nd_b = b.map_delete_boxes(boxes) if len(boxes) else 0
        rb = b.map_incremental(post, 0.2, True, apply=True, want_lists=True)
        assert (ra["n_add"], ra["n_nodown"], ra["add_counter"], ra["n_deleted"]) == (rb["n_add"], rb["n_nodown"], rb["add_counter"], nd_b)
        assert ra["n_add"] + ra["n_nodown"] > 100
        assert a.map_validnum() == b.map_validnum()
        assert bits_equal(_key(a.map_flatten()), _key(b.map_flatten()))
        if ref is not None:
            if len(boxes):
                assert ref.delete_boxes(boxes) == nd_b
            assert ref.add(rb["add"], True) == rb["add_counter"]
            ref.add(rb["nodown"], False)
            assert ref.validnum() == a.map_validnum()
            ref.quiesce(50)
            assert bits_equal(_key(a.map_flatten()), _key(ref.flatten()))
    a.close(); b.close()


@pytest.mark.parametrize("W", [2, 4, 8])
def test_loopback_world(gpu_capi, oracle, synth, small_case, W):
    """The C-ABI exchange with world > 1 on ONE GPU (lio_debug_set "replicate_loopback"): W virtual ranks, each with its own
    scan stream, park their lists; the last one packs every rank's message into its slot of the gathered buffer (the
    place of ncclAllGather) and runs what a real rank runs -- replicate_pack_kernel per rank, replicate_compact_kernel's
    cross-rank offsets, the overflow repeat, the rank-ordered Delete_Point_Boxes + Add_Points pair.  A delete box comes
    from a middle rank, exactly one rank overflows its message in round 0.  Checked after every round against a twin
    context that applies the rank-ordered lists through lio_map_add_pair and against the reference ikd-Tree
    (ikd_Tree.cpp:414-489; lists laserMapping.cpp:1442-1493; boxes :1183-1198)."""
    w = small_case["world"]
    thin = np.ascontiguousarray(w.points[::2])
    nq_small, nq_big = 1500, 6000
    big_rank = W // 2                                        # the rank whose lists overflow a message in round 0
    box_rank = max(0, W // 2 - 1) if W > 2 else 1
    a = gpu_capi.LioContext(0.2, max_scan_points=nq_big, max_map_points=400_000)     # loop-back exchange
    b = gpu_capi.LioContext(0.2, max_scan_points=nq_big, max_map_points=400_000)     # twin: plain calls
    a.map_build(thin); b.map_build(thin)
    ref = oracle.RefIkdTree(0.2) if oracle.ref_available() else None
    if ref is not None:
        ref.build(thin)
    a.debug_set("replicate_loopback", 1)
    kfa, kfb = gpu_capi.Esekf(3), gpu_capi.Esekf(3)
    for k in range(3):
        posts, lists, boxes_all = [], [], []
        # every virtual rank's scan step on the twin first: its own lists (apply = False: the map only changes at the exchange)
        for r in range(W):
            n = nq_big if r == big_rank else nq_small
            sc = synth.make_scan(w, n, seed=synth.SEED + 31 + 17 * k + 1000 * r, pos=(2.0 + 0.9 * k + 5.0 * r, -1.5 + 0.4 * k - 3.0 * r, 0.0),
                                 rpy=(0.004, -0.006, 0.3 + 0.05 * k + 0.7 * r))
            ps, rt = synth.perturb_state(sc, dpos=0.05, drot_deg=0.3, seed=synth.SEED + 5 + k + r)
            x0 = gpu_capi.make_state26(ps, rt, sc.off_R, sc.off_T)
            kfb.change_x(x0); kfb.change_P(synth.init_cov())
            b.scan_set(sc.body)
            kfb.update_iterated_dyn_share_modified(b, 0.001, False)
            post = kfb.get_x()[:14].copy()
            posts.append((sc, x0, post))
            lists.append(b.map_incremental(post, 0.2, True, apply=False, want_lists=True))
        if k == 0:
            # a message capacity only the big rank's lists exceed
            sizes = sorted(max(l["n_add"], l["n_nodown"]) for l in lists)
            biggest = max(lists[big_rank]["n_add"], lists[big_rank]["n_nodown"])
            assert biggest == sizes[-1] and sizes[-1] > sizes[-2] + 1
            a.replicate_init(b"", W - 1, W, (sizes[-1] + sizes[-2]) // 2)
        for r in range(W):
            sc, x0, post = posts[r]
            kfa.change_x(x0); kfa.change_P(synth.init_cov())
            a.scan_set(sc.body)
            kfa.update_iterated_dyn_share_modified(a, 0.001, False)
            assert np.array_equal(kfa.get_x()[:14], post)
            a.debug_set("replicate_loopback_rank", r)
            if r == box_rank and k >= 1:
                bx = np.array([[-30.0 + 2 * k, -30.0, -1.0, -27.0 + 2 * k, 30.0, 5.0]], np.float32)
                a.replicate_queue_boxes(bx)
                boxes_all.append(bx)
            ra = a.replicate_step(post, 0.2, True)
            if r + 1 < W:
                assert (ra["n_add"], ra["n_nodown"]) == (0, 0)           # parked, nothing exchanged yet
        add = np.concatenate([l["add"] for l in lists]); nodown = np.concatenate([l["nodown"] for l in lists])
        assert (ra["n_add"], ra["n_nodown"]) == (len(add), len(nodown))
        nd_b = b.map_delete_boxes(np.concatenate(boxes_all)) if boxes_all else 0
        cnt_b = b.map_add_pair(add, nodown)
        assert (ra["n_deleted"], ra["add_counter"]) == (nd_b, cnt_b)
        assert a.map_validnum() == b.map_validnum()
        assert bits_equal(_key(a.map_flatten()), _key(b.map_flatten()))
        if ref is not None:
            if boxes_all:
                assert ref.delete_boxes(np.concatenate(boxes_all)) == nd_b
            assert ref.add(add, True) == cnt_b
            ref.add(nodown, False)
            assert ref.validnum() == a.map_validnum()
            ref.quiesce(50)
            assert bits_equal(_key(a.map_flatten()), _key(ref.flatten()))
    a.close(); b.close()


def test_deferred_schedule_through_rccl(gpu_capi, oracle, synth, small_case):
    """The deferred schedule with the REAL collective on the library's stream (one rank: ncclCommInitRank with world 1, the
    all-gather really runs -- what `bench.py --gpus N` does on every rank): scan step k + 1 searches the map without the lists of
    exchange k, which enter at the start of exchange k + 1 with their plan made on the side stream.  Twin context on the same
    schedule through plain calls; posteriors equal at every step, maps equal at the end (and equal to the reference tree)."""
    w = small_case["world"]
    thin = np.ascontiguousarray(w.points[::2])
    nq = 8000
    a = gpu_capi.LioContext(0.2, max_scan_points=nq, max_map_points=400_000)
    b = gpu_capi.LioContext(0.2, max_scan_points=nq, max_map_points=400_000)
    a.map_build(thin); b.map_build(thin)
    ref = oracle.RefIkdTree(0.2) if oracle.ref_available() else None
    if ref is not None:
        ref.build(thin)
    a.replicate_init(gpu_capi.replicate_unique_id(), 0, 1, 0)
    a.replicate_defer(True)
    kfa, kfb = gpu_capi.Esekf(3), gpu_capi.Esekf(3)
    held = None
    for k in range(4):
        sc, ps, rt = _drive(synth, w, nq, k)
        x0 = gpu_capi.make_state26(ps, rt, sc.off_R, sc.off_T)
        for ctx, kf in ((a, kfa), (b, kfb)):
            kf.change_x(x0); kf.change_P(synth.init_cov())
            ctx.scan_set(sc.body)
            kf.update_iterated_dyn_share_modified(ctx, 0.001, False)
        assert np.array_equal(kfa.get_x(), kfb.get_x()), k            # both searched the map without the previous exchange's lists
        post = kfa.get_x()[:14]
        ra = a.replicate_step(post, 0.2, True)
        mine = b.map_incremental(post, 0.2, True, apply=False, want_lists=True)
        assert (ra["n_add"], ra["n_nodown"]) == (mine["n_add"], mine["n_nodown"])
        if held is not None:
            cnt = b.map_add_pair(held["add"], held["nodown"])
            assert ra["add_counter"] == cnt
            if ref is not None:
                assert ref.add(held["add"], True) == cnt
                ref.add(held["nodown"], False)
        held = mine
    assert a.debug_get("plans_on_side_stream") == 3 and a.debug_get("deferred_settled") == 3
    assert a.debug_get("settled_beside_exchange") == 3                 # (round 5: every planned batch was committed on the side stream, beside the next exchange)
    cnt = b.map_add_pair(held["add"], held["nodown"])
    if ref is not None:
        ref.add(held["add"], True); ref.add(held["nodown"], False)
    assert a.map_validnum() == b.map_validnum()                        # (any map call lets the held batch in)
    assert bits_equal(_key(a.map_flatten()), _key(b.map_flatten()))
    if ref is not None:
        ref.quiesce(50)
        assert bits_equal(_key(a.map_flatten()), _key(ref.flatten()))
    a.close(); b.close()


@pytest.mark.parametrize("W,tiny_messages", [(2, False), (8, False), (4, True)])
def test_deferred_schedule_in_a_loopback_world(gpu_capi, oracle, synth, small_case, W, tiny_messages):
    """lio_replicate_defer: the lists and boxes gathered by exchange k enter the map at the START of exchange k + 1; scan step
    k + 1 searches the map without them while the read-only half of their Add_Points (keys, sort, the speculative fold) runs on
    a side stream, and only the half that writes stays between two scan steps.  Held against a twin context driven by plain
    calls on the SAME schedule -- every rank's scan step, then the previous round's Delete_Point_Boxes + Add_Points pair -- so
    the posterior of every scan step (it depends on every neighbour the search found) says the two maps were the same map
    when they were searched; the flattened maps are compared at the end (any map call lets a batch that is held back enter
    first) with each other and with the reference ikd-Tree fed the same sequence of calls (ikd_Tree.cpp:414-489, 536-556).  A
    delete box travels in round 1 (its batch is planned after the deletion, on the main stream), one rank overflows its
    message in round 0; the other batches must have been planned on the side stream.  tiny_messages: a message holds 64 points,
    so EVERY round overflows and repeats its exchange at full capacity -- in rounds >= 1 behind a first round that has already let
    the held batch into the map: the repeat must pack the lists this rank made BEFORE that (the documented schedule, the twin's)."""
    w = small_case["world"]
    thin = np.ascontiguousarray(w.points[::2])
    nq_small, nq_big = 1500, 6000
    big_rank = W // 2
    box_rank = max(0, W // 2 - 1) if W > 2 else 1
    a = gpu_capi.LioContext(0.2, max_scan_points=nq_big, max_map_points=400_000)     # loop-back exchange, deferred schedule
    b = gpu_capi.LioContext(0.2, max_scan_points=nq_big, max_map_points=400_000)     # twin: plain calls on the same schedule
    a.map_build(thin); b.map_build(thin)
    ref = oracle.RefIkdTree(0.2) if oracle.ref_available() else None
    if ref is not None:
        ref.build(thin)
    a.debug_set("replicate_loopback", 1)
    kfa, kfb = gpu_capi.Esekf(3), gpu_capi.Esekf(3)
    held = None                                              # the twin's batch of the previous round: (boxes, add, nodown)
    rounds = 4

    def apply_held():
        boxes, add, nodown = held
        nd = b.map_delete_boxes(boxes) if len(boxes) else 0
        cnt = b.map_add_pair(add, nodown)
        if ref is not None:
            if len(boxes):
                assert ref.delete_boxes(boxes) == nd
            assert ref.add(add, True) == cnt
            ref.add(nodown, False)
        return nd, cnt

    for k in range(rounds):
        posts, lists, boxes_all = [], [], []
        for r in range(W):                                   # the twin: every rank's scan step on the map WITHOUT the held batch
            n = nq_big if r == big_rank else nq_small
            sc = synth.make_scan(w, n, seed=synth.SEED + 31 + 17 * k + 1000 * r, pos=(2.0 + 0.9 * k + 5.0 * r, -1.5 + 0.4 * k - 3.0 * r, 0.0),
                                 rpy=(0.004, -0.006, 0.3 + 0.05 * k + 0.7 * r))
            ps, rt = synth.perturb_state(sc, dpos=0.05, drot_deg=0.3, seed=synth.SEED + 5 + k + r)
            x0 = gpu_capi.make_state26(ps, rt, sc.off_R, sc.off_T)
            kfb.change_x(x0); kfb.change_P(synth.init_cov())
            b.scan_set(sc.body)
            kfb.update_iterated_dyn_share_modified(b, 0.001, False)
            post = kfb.get_x()[:14].copy()
            posts.append((sc, x0, post))
            lists.append(b.map_incremental(post, 0.2, True, apply=False, want_lists=True))
        if k == 0:
            sizes = sorted(max(l["n_add"], l["n_nodown"]) for l in lists)
            a.replicate_init(b"", W - 1, W, 64 if tiny_messages else (sizes[-1] + sizes[-2]) // 2)
            a.replicate_defer(True)
        if tiny_messages:
            assert max(max(l["n_add"], l["n_nodown"]) for l in lists) > 64, k       # this round overflows
        for r in range(W):
            sc, x0, post = posts[r]
            a.debug_set("replicate_loopback_rank", r)       # (before the scan step: the LAST rank's step is the one the plan runs beside)
            kfa.change_x(x0); kfa.change_P(synth.init_cov())
            a.scan_set(sc.body)
            kfa.update_iterated_dyn_share_modified(a, 0.001, False)
            assert np.array_equal(kfa.get_x()[:14], post), (k, r)      # searched the same map
            if r == box_rank and k == 1:
                bx = np.array([[-30.0 + 2 * k, -30.0, -1.0, -27.0 + 2 * k, 30.0, 5.0]], np.float32)
                a.replicate_queue_boxes(bx)
                boxes_all.append(bx)
            ra = a.replicate_step(post, 0.2, True)
        add = np.concatenate([l["add"] for l in lists]); nodown = np.concatenate([l["nodown"] for l in lists])
        assert (ra["n_add"], ra["n_nodown"]) == (len(add), len(nodown))
        if held is not None:                                 # the previous round's batch entered both maps during this exchange
            nd_b, cnt_b = apply_held()
            assert (ra["n_deleted"], ra["add_counter"]) == (nd_b, cnt_b)
        else:
            assert (ra["n_deleted"], ra["add_counter"]) == (0, 0)
        held = (np.concatenate(boxes_all) if boxes_all else np.zeros((0, 6), np.float32), add, nodown)
    assert a.debug_get("deferred_settled") == rounds - 1
    assert a.debug_get("plans_on_side_stream") == rounds - 2    # every settled batch but the one with the delete box
    assert a.debug_get("settled_beside_exchange") == rounds - 2  # ... and those were committed on the side stream, beside the exchange that followed
    # any map call lets the last batch in
    apply_held()
    assert a.map_validnum() == b.map_validnum()
    assert a.debug_get("deferred_settled") == rounds
    assert bits_equal(_key(a.map_flatten()), _key(b.map_flatten()))
    if ref is not None:
        assert ref.validnum() == a.map_validnum()
        ref.quiesce(50)
        assert bits_equal(_key(a.map_flatten()), _key(ref.flatten()))
    a.close(); b.close()


_WORKER = r"""
import os, sys, numpy as np
sys.path.insert(0, sys.argv[1])
rank, world, idfile, outdir = int(sys.argv[2]), int(sys.argv[3]), sys.argv[4], sys.argv[5]
from better_fastlio2_amd import capi, synth
uid = open(idfile, "rb").read()
w = synth.make_world(200_000, 0.2)
thin = np.ascontiguousarray(w.points[::2])
nq = 8000
ctx = capi.LioContext(0.2, max_scan_points=nq, max_map_points=600_000, device=rank)
ctx.map_build(thin)
ctx.replicate_init(uid, rank, world, 0)
kf = capi.Esekf(3)
lists = []
for k in range(3):
    sc = synth.make_scan(w, nq, seed=synth.SEED + 31 + 17 * k + 1000 * rank, pos=(2.0 + 0.9 * k + 6.0 * rank, -1.5 + 0.4 * k, 0.0),
                         rpy=(0.004, -0.006, 0.3 + 0.05 * k + 0.7 * rank))
    ps, rt = synth.perturb_state(sc, dpos=0.05, drot_deg=0.3, seed=synth.SEED + 5 + k + rank)
    kf.change_x(capi.make_state26(ps, rt, sc.off_R, sc.off_T)); kf.change_P(synth.init_cov())
    ctx.scan_set(sc.body)
    kf.update_iterated_dyn_share_modified(ctx, 0.001, False)
    post = kf.get_x()[:14]
    mine = ctx.map_incremental(post, 0.2, True, apply=False, want_lists=True)      # this rank's lists, for the reference replay
    if rank == 1 and k == 1:
        ctx.replicate_queue_boxes(np.array([[-30.0, -30.0, -1.0, -27.0, 30.0, 5.0]], np.float32))
    r = ctx.replicate_step(post, 0.2, True)
    lists.append((mine["add"], mine["nodown"], r))
np.savez(os.path.join(outdir, f"rank{rank}.npz"), flat=ctx.map_flatten(), valid=ctx.map_validnum(),
         **{f"add{k}": l[0] for k, l in enumerate(lists)}, **{f"nodown{k}": l[1] for k, l in enumerate(lists)},
         totals=np.array([[l[2]["n_add"], l[2]["n_nodown"], l[2]["n_deleted"]] for l in lists]))
ctx.close()
"""


def test_two_ranks_two_gpus(gpu_capi, oracle, synth):
    if gpu_capi.load().lio_device_count() < 2:
        pytest.skip("needs two GPUs (the driver's multi-GPU node); one rank through RCCL is covered above")
    uid = gpu_capi.replicate_unique_id()
    with tempfile.TemporaryDirectory() as td:
        idfile = os.path.join(td, "id.bin")
        open(idfile, "wb").write(uid)
        script = os.path.join(td, "worker.py")
        open(script, "w").write(_WORKER)
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs = [subprocess.Popen([sys.executable, script, ROOT, str(r), "2", idfile, td], env=env) for r in range(2)]
        for p in procs:
            assert p.wait(timeout=600) == 0
        r0, r1 = np.load(os.path.join(td, "rank0.npz")), np.load(os.path.join(td, "rank1.npz"))
    assert int(r0["valid"]) == int(r1["valid"]) and np.array_equal(r0["totals"], r1["totals"])
    assert bits_equal(_key(r0["flat"]), _key(r1["flat"]))                       # replicas bit-identical
    if oracle.ref_available():
        w = synth.make_world(200_000, 0.2)
        ref = oracle.RefIkdTree(0.2)
        ref.build(np.ascontiguousarray(w.points[::2]))
        for k in range(3):                                                       # boxes, then rank-ordered lists
            if k == 1:
                ref.delete_boxes(np.array([[-30.0, -30.0, -1.0, -27.0, 30.0, 5.0]], np.float32))
            ref.add(np.concatenate([r0[f"add{k}"], r1[f"add{k}"]]), True)
            ref.add(np.concatenate([r0[f"nodown{k}"], r1[f"nodown{k}"]]), False)
        ref.quiesce(100)
        assert ref.validnum() == int(r0["valid"])
        assert bits_equal(_key(r0["flat"]), _key(ref.flatten()))
