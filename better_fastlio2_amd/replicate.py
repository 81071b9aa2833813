"""Keeping map replicas identical across GPUs.

Every rank owns one GPU, one scan stream and one full replica of the map.  After a scan step a rank
holds the points it wants inserted (`PointToAdd` / `PointNoNeedDownsample`, src/laserMapping.cpp
:1442-1493) and at most a few delete boxes (`cub_needrm`, :1183-1198).  One all-gather of
fixed-capacity buffers moves everybody's list to everybody (backend "nccl" = RCCL over xGMI on
MI355X; "gloo" in the CPU tests); every replica then applies the lists in RANK ORDER through the same
sequential-equivalent `Add_Points` rule, so the replicas stay bit-identical.  Messages are small
(<= N_q float4 = 1.6 MB per rank), i.e. latency-bound: no ring all-reduce, no bucketing.
"""
from __future__ import annotations

import numpy as np
import torch
import torch.distributed as dist


def exchange_inserts(points: np.ndarray, capacity: int, device: "torch.device | str" = "cpu", group=None):
    """All-gather variable-length insert lists.  points: [n, 3] float32 with n <= capacity.
    Returns a list (one entry per rank, rank order) of [n_r, 3] float32 arrays."""
    if not dist.is_initialized():
        return [np.ascontiguousarray(points, np.float32)]
    world = dist.get_world_size(group)
    n = len(points)
    if n > capacity:
        raise ValueError(f"{n} points exceed the exchange capacity {capacity}")
    # one buffer per rank: row 0 = [count, 0, 0, 0], rows 1.. = x, y, z, 0  (float4 records)
    buf = torch.zeros((capacity + 1, 4), dtype=torch.float32)
    buf[0, 0] = float(n)
    if n:
        buf[1:n + 1, :3] = torch.from_numpy(np.ascontiguousarray(points, np.float32))
    buf = buf.to(device)
    out = torch.empty((world * (capacity + 1), 4), dtype=torch.float32, device=device)   # concatenated along dim 0
    dist.all_gather_into_tensor(out, buf, group=group)
    out = out.cpu().numpy().reshape(world, capacity + 1, 4)
    res = []
    for r in range(world):
        k = int(out[r, 0, 0])
        res.append(np.ascontiguousarray(out[r, 1:k + 1, :3]))
    return res


class _DeviceView:
    """Wraps a raw device pointer so that torch can alias it (``__cuda_array_interface__``, read-only use)."""

    def __init__(self, ptr: int, n: int):
        self.__cuda_array_interface__ = {"shape": (n, 4), "typestr": "<f4", "data": (int(ptr), False), "version": 2}


def exchange_inserts_device(ptr: int, n: int, capacity: int, device, group=None):
    """Device-resident form: this rank's list is n float4 records at device pointer `ptr`
    (lio_map_incremental_device).  Returns (tensor [total, 4] float32 on `device`, rank order; counts).
    One all-gather of the counts and one of fixed-capacity buffers -- xGMI moves <= capacity*16 B per
    rank, nothing goes through the host."""
    if n > capacity:
        raise ValueError(f"{n} points exceed the exchange capacity {capacity}")
    mine = torch.as_tensor(_DeviceView(ptr, n), device=device) if n else torch.empty((0, 4), dtype=torch.float32, device=device)
    if not dist.is_initialized():
        return mine.clone(), [n]
    world = dist.get_world_size(group)
    buf = torch.zeros((capacity, 4), dtype=torch.float32, device=device)
    if n:
        buf[:n] = mine
    cnt = torch.tensor([n], dtype=torch.int32, device=device)
    counts = torch.empty(world, dtype=torch.int32, device=device)
    dist.all_gather_into_tensor(counts, cnt, group=group)
    out = torch.empty((world * capacity, 4), dtype=torch.float32, device=device)
    dist.all_gather_into_tensor(out, buf, group=group)
    cs = [int(v) for v in counts.cpu().tolist()]
    parts = [out[r * capacity: r * capacity + cs[r]] for r in range(world)]
    return torch.cat(parts, 0).contiguous(), cs


def exchange_boxes(boxes: np.ndarray, max_boxes: int = 4, device: "torch.device | str" = "cpu", group=None):
    """All-gather delete boxes ([nb, 6] float32, nb <= max_boxes); returns them concatenated in rank order."""
    boxes = np.ascontiguousarray(boxes, np.float32).reshape(-1, 6)
    if not dist.is_initialized():
        return boxes
    world = dist.get_world_size(group)
    buf = torch.zeros((max_boxes + 1, 6), dtype=torch.float32)
    buf[0, 0] = float(len(boxes))
    if len(boxes):
        buf[1:len(boxes) + 1] = torch.from_numpy(boxes)
    buf = buf.to(device)
    out = torch.empty((world * (max_boxes + 1), 6), dtype=torch.float32, device=device)
    dist.all_gather_into_tensor(out, buf, group=group)
    out = out.cpu().numpy().reshape(world, max_boxes + 1, 6)
    parts = [out[r, 1:int(out[r, 0, 0]) + 1] for r in range(world)]
    return np.ascontiguousarray(np.concatenate(parts, 0)) if parts else boxes


def apply_in_rank_order(map_like, insert_lists, downsample_on: bool = True) -> int:
    """Apply every rank's insert list, rank 0 first (map_like: LioContext or anything with
    map_add(points, downsample_on)).  Returns the summed Add_Points counters."""
    total = 0
    for pts in insert_lists:
        if len(pts):
            total += map_like.map_add(pts, downsample_on)
    return total


class DeferredExchange:
    """The DEFERRED schedule of the exchange (lio_replicate_defer) as a protocol twin: the lists and boxes
    gathered by exchange k enter every replica at the START of exchange k + 1 -- scan step k + 1 searches the map without them, on
    every rank alike -- in the order every replica shares: Delete_Point_Boxes(all boxes), Add_Points(all PointToAdd, true),
    Add_Points(all PointNoNeedDownsample, false), rank order inside each.  (In liblio_hip.so the read-only half of the held batch's
    Add_Points runs on a side stream beside scan step k + 1; here only the ORDER is modelled -- which is what keeps the replicas
    bit-identical.)  map_like: anything with map_add(points, downsample_on) and, when boxes are used, map_delete_boxes(boxes)."""

    def __init__(self, capacity: int, device: "torch.device | str" = "cpu", group=None):
        self.capacity, self.device, self.group = capacity, device, group
        self.held = None                       # (boxes [nb, 6], add lists per rank, nodown lists per rank)

    def _apply(self, map_like) -> int:
        if self.held is None:
            return 0
        boxes, adds, nodowns = self.held
        self.held = None
        if len(boxes):
            map_like.map_delete_boxes(boxes)
        counter = apply_in_rank_order(map_like, adds, True)
        apply_in_rank_order(map_like, nodowns, False)
        return counter

    def step(self, map_like, add_pts, nodown_pts, boxes=None) -> int:
        """One exchange: gather this step's lists, let the PREVIOUS step's batch into the map, hold the new one.
        Returns Add_Points' counter of the batch that entered."""
        adds = exchange_inserts(add_pts, self.capacity, self.device, self.group)
        nodowns = exchange_inserts(nodown_pts, self.capacity, self.device, self.group)
        bx = exchange_boxes(np.zeros((0, 6), np.float32) if boxes is None else boxes, device=self.device, group=self.group)
        counter = self._apply(map_like)
        self.held = (bx, adds, nodowns)
        return counter

    def flush(self, map_like) -> int:
        """Any other use of the map lets a held batch in first (lio_hip.h)."""
        return self._apply(map_like)
