"""Multi-GPU sharding of the scene-processing path (one process per GPU).

Instances are sharded by contiguous range: rank g owns global candidate indices
[g*n, (g+1)*n) and the transform subtrees they hang from, so transform, cull and
the local sort need no communication. The only exchange step is after the
frame:

1. all-gather of every rank's per-view compaction states and transparent count
   (a few hundred bytes);
2. all-gather of the packed visible lists (RCCL over xGMI; `nccl` backend);
   because rank order is index order, the merged list of a (view, bucket) is the
   concatenation of the ranks' ranges - built by one segmented-copy kernel from
   a descriptor table (`plan_visible_merge`);
3. all-gather of the sorted draw-key runs, merged by the G-way rank merge kernel.

The reference is single-GPU; nothing here has a counterpart in it. The
collectives are plain torch.distributed calls, so the same exchange runs on
`gloo` with CPU tensors in the tests (the two device kernels are then served by
a numpy-backed source, see tests/test_parallel_cpu.py).
"""
from __future__ import annotations

import numpy as np

from . import _native as N


def shard_range(total: int, world: int, rank: int):
    """Contiguous range [begin, end) of `total` items owned by `rank`."""
    base, rem = divmod(total, world)
    begin = rank * base + min(rank, rem)
    return begin, begin + base + (1 if rank < rem else 0)


def plan_visible_merge(ranges: np.ndarray, partition_capacity: int = 0):
    """Descriptor table for merging per-rank visible lists.

    ranges: (G, V, 4, 2) uint32 execution ranges {offset, count} of every rank.
    Each rank's send buffer holds its views back to back (view v's list starts at
    the sum of the earlier views' visible counts). Returns
      desc           (G*V*4, 3) uint64 rows {src offset in the gathered buffer,
                     dst offset in the merged buffer, count}, given a per-rank
                     stride added by the caller (src offsets here are inside the
                     rank's own packed buffer),
      merged_ranges  (V, 4, 2) uint64 {offset, count} of the merged lists (offsets
                     inside view v's merged list),
      view_offsets   (V+1,) uint64 start of view v in the merged buffer,
      overflow       (V,) counts dropped by the per-bucket capacity.
    """
    ranges = np.asarray(ranges, dtype=np.uint64)
    g, v = ranges.shape[0], ranges.shape[1]
    counts = ranges[..., 1]                         # (G, V, 4)
    per_rank_view = counts.sum(axis=2)              # (G, V)
    rank_view_start = np.zeros((g, v), np.uint64)   # start of view v inside rank's buffer
    rank_view_start[:, 1:] = np.cumsum(per_rank_view, axis=1)[:, :-1]
    merged_counts = counts.sum(axis=0)              # (V, 4)
    overflow = np.zeros(v, np.uint64)
    if partition_capacity:
        cap = np.uint64(partition_capacity)
        clipped = np.minimum(merged_counts, cap)
        overflow = (merged_counts - clipped).sum(axis=1)
        merged_counts = clipped
    merged_ranges = np.zeros((v, 4, 2), np.uint64)
    merged_ranges[..., 1] = merged_counts
    merged_ranges[:, 1:, 0] = np.cumsum(merged_counts, axis=1)[:, :-1]
    view_totals = merged_counts.sum(axis=1)
    view_offsets = np.zeros(v + 1, np.uint64)
    view_offsets[1:] = np.cumsum(view_totals)
    desc = np.zeros((g * v * 4, 3), np.uint64)
    k = 0
    for vi in range(v):
        for b in range(4):
            written = np.uint64(0)
            for r in range(g):
                cnt = counts[r, vi, b]
                room = merged_counts[vi, b] - written
                take = min(cnt, room)
                desc[k] = (rank_view_start[r, vi] + ranges[r, vi, b, 0],
                           view_offsets[vi] + merged_ranges[vi, b, 0] + written, take)
                written += take
                k += 1
    return desc, merged_ranges, view_offsets, overflow


def apply_descriptors(gathered: np.ndarray, rank_stride: int, desc: np.ndarray, g: int,
                      total: int) -> np.ndarray:
    """numpy statement of the segmented copy (used by the CPU tests)."""
    out = np.zeros(total, gathered.dtype)
    flat = gathered.reshape(-1)
    k = 0
    views_buckets = desc.shape[0] // g
    for vb in range(views_buckets):
        for r in range(g):
            src, dst, cnt = (int(x) for x in desc[k])
            out[dst:dst + cnt] = flat[r * rank_stride + src: r * rank_stride + src + cnt]
            k += 1
    return out


class _DevicePointer:
    """Zero-copy view of library-owned HBM for torch (`__cuda_array_interface__`)."""

    def __init__(self, ptr: int, count: int, typestr: str):
        self.__cuda_array_interface__ = {
            "shape": (count,), "typestr": typestr, "data": (ptr, False), "version": 2}


def wrap_device(ptr: int, count: int, dtype):
    import torch
    typestr = {torch.int32: "<i4", torch.int64: "<i8", torch.uint8: "|u1"}[dtype]
    if count == 0:
        return torch.empty(0, dtype=dtype, device="cuda")
    return torch.as_tensor(_DevicePointer(ptr, count, typestr), device="cuda")


class HipShardSource:
    """What ShardExchange needs from one rank's scene, served from the HIP
    library's device buffers (zero-copy torch views) and its two merge kernels."""

    def __init__(self, scene):
        import torch
        self.torch = torch
        self.scene = scene
        self.device = torch.device("cuda", torch.cuda.current_device())

    def frame_info(self):
        return self.scene.frame_info()  # waits for the frame; scalar results on the host

    def begin(self):
        self.out = self.scene.device_outputs()

    def visible(self, view: int, count: int):
        return wrap_device(self.out.visible + 4 * view * self.out.visible_stride, count,
                           self.torch.int32)

    def keys(self, count: int):
        return wrap_device(self.out.transparent_keys, count, self.torch.int64)

    def concat_ranges(self, gathered, merged, desc: np.ndarray):
        self.scene.concat_ranges(gathered.data_ptr(), merged.data_ptr(), desc)

    def merge_runs(self, gathered, run_stride: int, lengths: np.ndarray, merged):
        ptrs = [gathered.data_ptr() + 8 * r * run_stride for r in range(lengths.shape[0])]
        self.scene.merge_runs(ptrs, lengths, merged.data_ptr())


class ShardExchange:
    """Per-frame exchange of one rank's results with the other ranks.

    `source` provides this rank's frame results and the two merge primitives
    (HipShardSource on a GPU; the CPU tests pass a numpy-backed stand-in and run
    the same exchange over `gloo`). After run(): vis_merged holds, view after
    view, the merged visible list of every (view, bucket) (merged_ranges /
    view_offsets describe it) and key_merged[:transparent_total] the merged
    descending draw keys - identical on every rank.
    """

    def __init__(self, source, world: int, rank: int, views: int, group=None, device=None):
        import torch
        self.torch = torch
        from .scene import HipScene
        if isinstance(source, HipScene):
            source = HipShardSource(source)
        self.source = source
        self.world, self.rank, self.views, self.group = world, rank, views, group
        self.device = device if device is not None else source.device
        self.hdr_len = views * 8 + 2
        self.hdr_send = torch.zeros(self.hdr_len, dtype=torch.int64, device=self.device)
        self.hdr_recv = torch.zeros(world * self.hdr_len, dtype=torch.int64, device=self.device)
        self.vis_cap = 0
        self.key_cap = 0
        self.vis_send = self.vis_recv = self.vis_merged = None
        self.key_send = self.key_recv = self.key_merged = None
        self.merged_ranges = None
        self.view_offsets = None
        self.transparent_total = 0

    def _grow(self, vis_need: int, key_need: int, merged_vis: int, merged_keys: int):
        # every rank sees the same gathered header, so all ranks grow alike and the
        # all-gather buffers keep equal sizes
        torch = self.torch
        if vis_need > self.vis_cap or self.vis_send is None:
            self.vis_cap = max(int(vis_need * 1.25) + 1024, 1024)
            self.vis_send = torch.zeros(self.vis_cap, dtype=torch.int32, device=self.device)
            self.vis_recv = torch.zeros(self.world * self.vis_cap, dtype=torch.int32, device=self.device)
        if key_need > self.key_cap or self.key_send is None:
            self.key_cap = max(int(key_need * 1.25) + 1024, 1024)
            self.key_send = torch.zeros(self.key_cap, dtype=torch.int64, device=self.device)
            self.key_recv = torch.zeros(self.world * self.key_cap, dtype=torch.int64, device=self.device)
        if self.vis_merged is None or self.vis_merged.numel() < merged_vis:
            self.vis_merged = torch.zeros(int(merged_vis * 1.25) + 1024, dtype=torch.int32,
                                          device=self.device)
        if self.key_merged is None or self.key_merged.numel() < merged_keys:
            self.key_merged = torch.zeros(int(merged_keys * 1.25) + 1024, dtype=torch.int64,
                                          device=self.device)

    def run(self):
        torch, dist = self.torch, self.torch.distributed
        src, v, g = self.source, self.views, self.world
        info = src.frame_info()
        hdr = np.zeros(self.hdr_len, np.int64)
        for vi, st in enumerate(info.views):
            hdr[vi * 8: vi * 8 + 8] = np.asarray(st.execution_ranges).reshape(-1)
        hdr[v * 8] = info.transparent_draw_count
        self.hdr_send.copy_(torch.from_numpy(hdr))
        dist.all_gather_into_tensor(self.hdr_recv, self.hdr_send, group=self.group)
        all_hdr = self.hdr_recv.cpu().numpy().reshape(g, self.hdr_len)
        ranges = all_hdr[:, : v * 8].reshape(g, v, 4, 2).astype(np.uint64)
        t_all = all_hdr[:, v * 8].astype(np.uint64)
        vis_counts = ranges[..., 1].sum(axis=(1, 2))
        desc, merged_ranges, view_offsets, _ = plan_visible_merge(ranges)
        self._grow(int(vis_counts.max()), int(t_all.max()), int(view_offsets[-1]), int(t_all.sum()))

        src.begin()
        # pack this rank's views back to back
        pos = 0
        for vi, st in enumerate(info.views):
            cnt = int(st.visible_count)
            if cnt:
                self.vis_send[pos: pos + cnt].copy_(src.visible(vi, cnt), non_blocking=True)
            pos += cnt
        dist.all_gather_into_tensor(self.vis_recv, self.vis_send, group=self.group)
        t_local = int(info.transparent_draw_count)
        if t_local:
            self.key_send[:t_local].copy_(src.keys(t_local), non_blocking=True)
        dist.all_gather_into_tensor(self.key_recv, self.key_send, group=self.group)

        # merged visible lists: one segmented copy; descriptor k of a (view, bucket)
        # belongs to rank k % g, whose packed list starts at rank * vis_cap
        desc_abs = desc.copy()
        desc_abs[:, 0] += (np.arange(desc.shape[0], dtype=np.uint64) % np.uint64(g)) * np.uint64(self.vis_cap)
        src.concat_ranges(self.vis_recv, self.vis_merged, desc_abs)
        # merged draw keys: G-way merge of the descending runs
        src.merge_runs(self.key_recv, self.key_cap, t_all, self.key_merged)
        self.merged_ranges, self.view_offsets = merged_ranges, view_offsets
        self.transparent_total = int(t_all.sum())
        return self
