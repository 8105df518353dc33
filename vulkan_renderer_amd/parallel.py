"""Multi-GPU sharding of the scene-processing path (one process per GPU).

Instances are sharded by contiguous range: rank g owns global candidate indices
[g*n, (g+1)*n) and the transform subtrees they hang from, so transform, cull and
the local sort need no communication. The only exchange step is after the
frame, and it is enqueued without a host round trip:

1. every rank packs one block at fixed capacities: a header (the execution
   ranges of every view, the draw count), its visible lists back to back and
   its sorted draw-key run (`vkrhip_exchange_pack`);
2. ONE all-gather of the blocks (RCCL over xGMI; `nccl` backend);
3. a planning kernel reads the gathered headers on the device and the merged
   lists follow: rank order is index order, so the merged list of a (view,
   bucket) is the concatenation of the ranks' ranges (one segmented-copy
   kernel), and the draw keys are a G-way rank merge (`vkrhip_exchange_merge`).

The host looks at the result words only in finish(): a rank whose lists did not
fit the capacities is flagged there, every rank sees the same flags and grows
its buffers alike, and the exchange of that frame is repeated.

Collective and merge run on a stream of their own and the buffers exist twice,
so the next frame starts as soon as the pack kernel has copied this frame's
results out of the scene: the all-gather over xGMI and the merge kernels of
frame k overlap the transform and cull of frame k+1.

The reference is single-GPU; nothing here has a counterpart in it. The
collective is a plain torch.distributed call, so the same exchange runs on
`gloo` with CPU tensors in the tests (pack and merge are then served by a
numpy-backed source, see tests/test_parallel_cpu.py).
"""
from __future__ import annotations

import contextlib

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


# ---- block layout (mirrors include/vkr_hip_scene.h, vkrhip_exchange_*) ------
def _round_up4(x: int) -> int:
    return (int(x) + 3) & ~3


def exchange_header_words(views: int) -> int:
    return _round_up4(views * 8 + 4)


def exchange_block_words(views: int, vis_cap: int, key_cap: int) -> int:
    return exchange_header_words(views) + _round_up4(vis_cap) + _round_up4(key_cap * 2)


RESULT_FIXED_WORDS = 8
FLAG_VISIBLE_TRUNCATED, FLAG_KEYS_TRUNCATED, FLAG_HEADER_MISMATCH = 1, 2, 4


def exchange_result_words(views: int) -> int:
    return RESULT_FIXED_WORDS + views + 1 + views * 8


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
    """One rank's scene as ShardExchange sees it. Pack is enqueued on the stream
    the scene itself runs on (wrapped for torch as an external stream); the
    collective and the merge go to a second stream, ordered after the pack by an
    event - no host wait anywhere."""

    def __init__(self, scene):
        import torch
        self.torch = torch
        self.scene = scene
        self.device = torch.device("cuda", torch.cuda.current_device())
        if scene.stream_handle() == 0:
            scene.set_stream(None)
        self.stream = torch.cuda.ExternalStream(scene.stream_handle(), device=self.device)
        self.exchange_stream = torch.cuda.Stream(device=self.device)

    def scene_scope(self):
        return self.torch.cuda.stream(self.stream)

    def exchange_scope(self):
        return self.torch.cuda.stream(self.exchange_stream)

    def frame_serial(self) -> int:
        return self.scene.frame_serial

    def pack(self, block, vis_cap: int, key_cap: int):
        self.scene.exchange_pack(block.data_ptr(), vis_cap, key_cap)

    def merge(self, gathered, world: int, vis_cap: int, key_cap: int, vis_merged, key_merged, result):
        self.scene.exchange_merge(gathered.data_ptr(), world, vis_cap, key_cap, vis_merged.data_ptr(),
                                  key_merged.data_ptr(), result.data_ptr(),
                                  self.exchange_stream.cuda_stream)

    def packed_event(self):
        """Event after everything enqueued on the scene's stream so far."""
        ev = self.torch.cuda.Event()
        ev.record(self.stream)
        return ev

    def exchanged_event(self):
        ev = self.torch.cuda.Event()
        ev.record(self.exchange_stream)
        return ev

    def scene_wait(self, event):
        self.stream.wait_event(event)

    def exchange_wait(self, event):
        self.exchange_stream.wait_event(event)

    def synchronize(self):
        self.stream.synchronize()
        self.exchange_stream.synchronize()


class ExchangeOverflow(RuntimeError):
    """A rank's lists did not fit the exchange blocks and the frame can no longer
    be repeated: the scene has been advanced since the run() that is being
    finished, so a second pack would carry the newer frame's lists under the
    older frame's name. Carries the capacities that would have fitted."""

    def __init__(self, need_visible: int, need_keys: int):
        super().__init__(f"exchange blocks overflowed (need {need_visible} visible indices, "
                         f"{need_keys} keys per rank) after the scene advanced: call finish() before "
                         "the next frame() until the blocks are sized")
        self.need_visible, self.need_keys = need_visible, need_keys


class _Buffers:
    __slots__ = ("send", "recv", "vis_merged", "key_merged", "result", "done")


class ShardExchange:
    """Per-frame exchange of one rank's results with the other ranks.

    `source` provides pack() and merge() for this rank's frame (HipShardSource on
    a GPU; the CPU tests pass a numpy-backed stand-in and run the same exchange
    over `gloo`). run() only enqueues; finish() waits for the latest run,
    validates the capacities (growing them and repeating the exchange when a rank
    overflowed) and fills merged_ranges / view_offsets / visible_total /
    transparent_total. After finish(): vis_merged holds, view after view, the
    merged visible list of every (view, bucket) and key_merged[:transparent_total]
    the merged descending draw keys - identical on every rank.

    Every rank must call run() and finish() in the same order: the capacities
    change only in finish(), from values every rank reads out of the same
    gathered headers, so the all-gather sizes always agree.

    Overlapped use (frame k+1 enqueued before finish() of frame k) needs blocks
    that already fit: a repeated exchange packs what the scene holds NOW, so
    finish() refuses to repeat once the scene has advanced past the frame whose
    run() it is finishing. Size the blocks with a finish() right after the first
    run() (or pass capacities that cover the worst case).
    """

    def __init__(self, source, world: int, rank: int, views: int, group=None, device=None,
                 vis_cap: int = 4096, key_cap: int = 1024):
        import torch
        self.torch = torch
        from .scene import HipScene
        if isinstance(source, HipScene):
            source = HipShardSource(source)
        self.source = source
        self.world, self.rank, self.views, self.group = world, rank, views, group
        self.device = device if device is not None else source.device
        self.overlapped = hasattr(source, "exchange_scope")
        self._scene_scope = source.scene_scope if self.overlapped else contextlib.nullcontext
        self._exchange_scope = source.exchange_scope if self.overlapped else contextlib.nullcontext
        self.regrown = 0
        self.merged_ranges = None
        self.view_offsets = None
        self.visible_total = 0
        self.transparent_total = 0
        self.vis_merged = self.key_merged = None
        self._turn = 0
        self._alloc(vis_cap, key_cap)

    def _alloc(self, vis_cap: int, key_cap: int):
        torch = self.torch
        self.vis_cap, self.key_cap = _round_up4(max(vis_cap, 4)), _round_up4(max(key_cap, 4))
        words = exchange_block_words(self.views, self.vis_cap, self.key_cap)
        self._sets = []
        with self._exchange_scope():
            for _ in range(2 if self.overlapped else 1):
                b = _Buffers()
                b.send = torch.zeros(words, dtype=torch.int32, device=self.device)
                b.recv = torch.zeros(self.world * words, dtype=torch.int32, device=self.device)
                b.vis_merged = torch.zeros(self.world * self.vis_cap, dtype=torch.int32, device=self.device)
                b.key_merged = torch.zeros(self.world * self.key_cap, dtype=torch.int64, device=self.device)
                b.result = torch.zeros(exchange_result_words(self.views), dtype=torch.int64,
                                       device=self.device)
                b.done = None
                self._sets.append(b)
        if self.overlapped:
            self.source.synchronize()  # the fills above ran on the exchange stream
        # gloo moves host memory: device blocks are staged through pinned buffers here
        # (its own staging of device tensors stalled after a few frames without a host
        # wait in between)
        dist = torch.distributed
        self._host_staged = (self._sets[0].send.is_cuda and dist.is_available() and dist.is_initialized()
                             and dist.get_backend(self.group) == "gloo")
        if self._host_staged:
            self._host_send = torch.zeros(words, dtype=torch.int32).pin_memory()
            self._host_recv = torch.zeros(self.world * words, dtype=torch.int32).pin_memory()
        self._last = None

    def run(self):
        """Enqueue pack, all-gather and merge of the frame that was just enqueued."""
        dist = self.torch.distributed
        src = self.source
        b = self._sets[self._turn % len(self._sets)]
        self._turn += 1
        self._run_serial = src.frame_serial() if hasattr(src, "frame_serial") else None
        if self.overlapped and b.done is not None:
            src.scene_wait(b.done)  # the exchange that used this set two frames ago
        with self._scene_scope():
            src.pack(b.send, self.vis_cap, self.key_cap)
        if self.overlapped:
            src.exchange_wait(src.packed_event())
        with self._exchange_scope():
            if self.world == 1 and not (dist.is_available() and dist.is_initialized()):
                b.recv.copy_(b.send, non_blocking=True)
            elif self._host_staged:
                self._host_send.copy_(b.send, non_blocking=True)
                self.torch.cuda.current_stream().synchronize()
                dist.all_gather_into_tensor(self._host_recv, self._host_send, group=self.group)
                b.recv.copy_(self._host_recv, non_blocking=True)
            else:
                dist.all_gather_into_tensor(b.recv, b.send, group=self.group)
            src.merge(b.recv, self.world, self.vis_cap, self.key_cap, b.vis_merged, b.key_merged,
                      b.result)
        if self.overlapped:
            b.done = src.exchanged_event()
        self._last = b
        return self

    def finish(self):
        """Wait for the latest exchange, repeat it with larger blocks if a rank overflowed."""
        v = self.views
        if self._last is None:
            raise RuntimeError("finish() without run()")
        while True:
            with self._exchange_scope():
                res = self._last.result.cpu().numpy().view(np.uint64)  # waits for the exchange stream
            flags = int(res[0])
            if flags & FLAG_HEADER_MISMATCH:
                raise RuntimeError("exchange blocks were packed for different view counts")
            if not flags & (FLAG_VISIBLE_TRUNCATED | FLAG_KEYS_TRUNCATED):
                break
            if self._run_serial is not None and self.source.frame_serial() != self._run_serial:
                raise ExchangeOverflow(int(res[1]), int(res[2]))
            vis_cap, key_cap = self.vis_cap, self.key_cap
            if int(res[1]) > vis_cap:
                vis_cap = int(int(res[1]) * 1.25) + 1024
            if int(res[2]) > key_cap:
                key_cap = int(int(res[2]) * 1.25) + 1024
            if self.overlapped:
                self.source.synchronize()
            self._alloc(vis_cap, key_cap)
            self.regrown += 1
            self.run()
        self.vis_merged, self.key_merged = self._last.vis_merged, self._last.key_merged
        self.visible_total = int(res[3])
        self.transparent_total = int(res[4])
        self.view_offsets = res[RESULT_FIXED_WORDS: RESULT_FIXED_WORDS + v + 1].copy()
        self.merged_ranges = res[RESULT_FIXED_WORDS + v + 1:].reshape(v, 4, 2).copy()
        return self


class RcclExchange:
    """The same exchange run entirely inside the HIP library: pack, ncclAllGather
    and merge are enqueued by one C call per frame (vkrhip_exchange_run) on the
    library's two streams, so the host spends a few launches per frame instead of
    a Python collective. torch.distributed is only used once, to hand rank 0's
    RCCL unique id to the other ranks. Same interface and the same capacity
    protocol as ShardExchange; world 1 needs no process group (the merge reads
    the rank's own block)."""

    def __init__(self, scene, world: int, rank: int, views: int, group=None, vis_cap: int = 4096,
                 key_cap: int = 1024):
        import torch
        self.torch = torch
        self.scene, self.world, self.rank, self.views = scene, world, rank, views
        if world > 1:
            dist = torch.distributed
            uid = torch.zeros(128, dtype=torch.uint8)
            if rank == 0:
                uid = torch.from_numpy(scene.comm_unique_id().copy())
            backend = dist.get_backend(group)
            if backend == "nccl":
                uid = uid.cuda()
            dist.broadcast(uid, src=0, group=group)
            scene.comm_init(uid.cpu().numpy(), world, rank)
        self.regrown = 0
        self.merged_ranges = None
        self.view_offsets = None
        self.visible_total = 0
        self.transparent_total = 0
        self.vis_merged = self.key_merged = None
        self._configure(vis_cap, key_cap)

    def _configure(self, vis_cap: int, key_cap: int):
        self.vis_cap, self.key_cap = _round_up4(max(vis_cap, 4)), _round_up4(max(key_cap, 4))
        self.scene.exchange_configure(self.world, self.vis_cap, self.key_cap)

    def run(self):
        self.scene.exchange_run()
        self._run_serial = self.scene.frame_serial
        return self

    def finish(self):
        v = self.views
        while True:
            res, vis_ptr, key_ptr = self.scene.exchange_finish(v)
            flags = int(res[0])
            if flags & FLAG_HEADER_MISMATCH:
                raise RuntimeError("exchange blocks were packed for different view counts")
            if not flags & (FLAG_VISIBLE_TRUNCATED | FLAG_KEYS_TRUNCATED):
                break
            if self.scene.frame_serial != self._run_serial:
                raise ExchangeOverflow(int(res[1]), int(res[2]))
            vis_cap, key_cap = self.vis_cap, self.key_cap
            if int(res[1]) > vis_cap:
                vis_cap = int(int(res[1]) * 1.25) + 1024
            if int(res[2]) > key_cap:
                key_cap = int(int(res[2]) * 1.25) + 1024
            self._configure(vis_cap, key_cap)
            self.regrown += 1
            self.run()
        self.vis_merged = wrap_device(vis_ptr, self.world * self.vis_cap, self.torch.int32)
        self.key_merged = wrap_device(key_ptr, self.world * self.key_cap, self.torch.int64)
        self.visible_total = int(res[3])
        self.transparent_total = int(res[4])
        self.view_offsets = res[RESULT_FIXED_WORDS: RESULT_FIXED_WORDS + v + 1].copy()
        self.merged_ranges = res[RESULT_FIXED_WORDS + v + 1:].reshape(v, 4, 2).copy()
        return self
