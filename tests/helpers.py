"""Shared drivers for the parity tests: run one scene through the HIP path (via
the C ABI) and through the CPU oracle, and compare every output bit for bit."""
from __future__ import annotations

import numpy as np

from conftest import assert_bits_equal

DIRTY_LOCAL, DIRTY_WORLD, WORLD_UPDATED = 0x01, 0x02, 0x08


def view_set(vs, extra_vp=None):
    """(planes, required, camera_view, projection) for a synth.ViewSet using the
    library's host-side frustum extraction."""
    from vulkan_renderer_amd import build_view_set, frustum_from_matrix
    planes, required = build_view_set(vs.view, vs.projection, vs.light_view_proj)
    if extra_vp is not None and len(extra_vp):
        ep = np.stack([frustum_from_matrix(m) for m in extra_vp])
        planes = np.concatenate([planes, ep])
        required = np.concatenate([required, np.full(len(extra_vp), 0x4, np.uint32)])
    return planes, required


def oracle_frame(oracle, sc, planes, required, view, projection, partition_capacity=0,
                 tie_eps=0.0, index_base=0, world=None):
    """Whole path on the CPU oracle. Returns a dict of every output."""
    n = sc.n
    if world is None:
        local, world, flags, dirty = oracle.scene_update(
            sc.parent, pos=sc.pos, quat=sc.quat, scale=sc.scale,
            flags=np.full(n, DIRTY_LOCAL, np.uint8))
    else:
        local, flags, dirty = None, None, np.zeros(0, np.uint32)
    pay = oracle.build_world_payload(world, sc.node, sc.ident, sc.matbits, sc.bounds, view,
                                     projection, source_index_base=index_base)
    cls, states, vis, ties = oracle.multiview_cull(
        pay["gpu_candidates"], planes, required, partition_capacity=partition_capacity,
        index_base=index_base, tie_eps=tie_eps)
    return dict(local=local, world=world, flags=flags, dirty=dirty, payload=pay, cls=cls,
                states=states, visible=vis, ties=ties)


def hip_scene_for(sc, views, **kw):
    from vulkan_renderer_amd import HipScene
    n_cand = sc.ident.shape[0]
    hs = HipScene(max_nodes=max(sc.n, 1), max_candidates=max(n_cand, 1), max_views=max(views, 1),
                  **kw)
    hs._index_base = kw.get("source_index_base", 0)
    hs.set_hierarchy(sc.parent, sc.level_offsets)
    hs.upload_trs(sc.pos, sc.quat, sc.scale)
    hs.set_candidates(sc.node, sc.ident, sc.matbits, sc.bounds)
    return hs


def compare_frame(hs, ref, sc, check_transform=True, partition_capacity=0):
    """Compares the last HIP frame of `hs` with the oracle outputs in `ref`."""
    info = hs.frame_info()
    pay = ref["payload"]
    n_cand = sc.ident.shape[0]
    if check_transform:
        assert_bits_equal(hs.read_locals(), ref["local"], "local matrices")
        assert_bits_equal(hs.read_world(), ref["world"], "world matrices")
        assert (hs.read_node_flags() == ref["flags"]).all(), "node flags"
        assert info.nodes_updated == ref["dirty"].shape[0]
    # stats and counts (application.h:745-749,761-762)
    assert info.objects_tested == int(pay["stats"][0])
    assert info.objects_culled_camera == int(pay["stats"][1])
    assert info.objects_without_bounds == int(pay["stats"][2])
    assert info.gpu_candidate_count == int(pay["counts"][0])
    assert info.gpu_camera_opaque_candidate_count == int(pay["counts"][1])
    assert info.transmission_gpu_candidate_count == int(pay["counts"][2])
    assert info.transparent_draw_count == int(pay["counts"][3])
    # per-view compaction state and visible lists
    states = ref["states"]
    assert len(info.views) == states.shape[0]
    for v, st in enumerate(info.views):
        assert (st.execution_ranges == states[v]["execution_ranges"]).all(), f"view {v} ranges"
        assert (st.bucket_counts == states[v]["bucket_counts"]).all(), f"view {v} counts"
        assert (st.bucket_cursors == states[v]["bucket_cursors"]).all(), f"view {v} cursors"
        assert st.visible_count == int(states[v]["visible_count"]), f"view {v} visible_count"
        assert st.overflow_count == int(states[v]["overflow_count"]), f"view {v} overflow"
        assert st.near_plane_ties == int(ref["ties"][v]), f"view {v} ties"
        got = hs.read_visible(v, st.visible_count)
        assert (got == ref["visible"][v, : st.visible_count]).all(), f"view {v} visible list"
    # transparent visibility bytes, sorted draws, instances
    if n_cand:
        assert (hs.read_transparent_visible() == pay["transparent_visible"]).all()
    t = info.transparent_draw_count
    draws, inst, keys, src = hs.read_transparent(t)
    assert (keys == pay["sort_keys"]).all(), "sorted keys"
    assert (src.astype(np.uint64) + np.uint64(hs_index_base(hs)) ==
            pay["sorted_source"].astype(np.uint64)).all(), "sorted source rows"
    assert draws.tobytes() == pay["draws"].tobytes(), "VkrDrawItem rows"
    assert inst.tobytes() == pay["instances"].tobytes(), "VkrInstanceDataGPU rows"
    return info


def clamp_to_list_capacity(states, visible, capacity):
    """What a per-view list capacity below the survivor count leaves of the oracle's
    (list-unlimited) result: every bucket keeps what still fits behind the buckets
    before it, the rest is counted as overflow. The reference sizes the list for
    every candidate (deferred.slang:400-403), so this knob is the library's own."""
    states = states.copy()
    out = np.full_like(visible, 0xFFFFFFFF)
    for v in range(states.shape[0]):
        cursor = 0
        for b in range(4):
            off, cnt = (int(x) for x in states[v]["execution_ranges"][b])
            kept = min(cnt, max(capacity - cursor, 0))
            out[v, cursor: cursor + kept] = visible[v, off: off + kept]
            states[v]["execution_ranges"][b] = (cursor, kept)
            states[v]["overflow_count"] += cnt - kept
            cursor += kept
        states[v]["visible_count"] = cursor
    return states, out


def hs_index_base(hs):
    return getattr(hs, "_index_base", 0)


# ---- the multi-rank exchange restated in numpy (checker for the exchange kernels,
# and the stand-in source that lets the exchange run over gloo on CPU) ----------
def numpy_exchange_pack(states, visible, keys, views, vis_cap, key_cap):
    """One rank's exchange block as include/vkr_hip_scene.h lays it out."""
    from vulkan_renderer_amd.parallel import exchange_block_words, exchange_header_words
    vis_cap4, header = (vis_cap + 3) & ~3, exchange_header_words(views)
    block = np.zeros(exchange_block_words(views, vis_cap, key_cap), np.uint32)
    pos = 0
    for v in range(views):
        block[v * 8: v * 8 + 8] = np.asarray(states[v]["execution_ranges"], np.uint32).reshape(-1)
        cnt = int(states[v]["visible_count"])
        room = max(0, min(cnt, vis_cap - pos))
        block[header + pos: header + pos + room] = visible[v][:room]
        pos += cnt
    block[views * 8: views * 8 + 4] = (len(keys), pos, views, 0)
    kept = min(len(keys), key_cap)
    key_words = block[header + vis_cap4: header + vis_cap4 + 2 * key_cap].view(np.uint64)
    key_words[:kept] = np.asarray(keys[:kept], np.uint64)
    return block


def numpy_exchange_merge(oracle, gathered, world, views, vis_cap, key_cap):
    """(flags, max_rank_visible, max_rank_keys, view_offsets, merged_ranges,
    merged visible list, merged keys) from the gathered blocks."""
    from vulkan_renderer_amd.parallel import (apply_descriptors, exchange_block_words,
                                              exchange_header_words, plan_visible_merge)
    words, header = exchange_block_words(views, vis_cap, key_cap), exchange_header_words(views)
    vis_cap4 = (vis_cap + 3) & ~3
    blocks = np.asarray(gathered, np.uint32).reshape(world, words)
    ranges = blocks[:, : views * 8].reshape(world, views, 4, 2).astype(np.uint64)
    draws = blocks[:, views * 8].astype(np.uint64)
    rank_visible = ranges[..., 1].sum(axis=(1, 2))
    flags = (1 if (rank_visible > vis_cap).any() else 0) | (2 if (draws > key_cap).any() else 0)
    desc, merged_ranges, view_offsets, _ = plan_visible_merge(ranges)
    lists = np.ascontiguousarray(blocks[:, header: header + vis_cap4])
    merged = None
    runs = None
    if not flags:
        merged = apply_descriptors(lists, vis_cap4, desc, world, int(view_offsets[-1]))
        runs = [blocks[r, header + vis_cap4: header + vis_cap4 + 2 * key_cap].view(np.uint64)[: int(draws[r])]
                for r in range(world)]
        runs = oracle.merge_runs_desc(runs)
    return (flags, int(rank_visible.max()), int(draws.max()), view_offsets, merged_ranges, merged, runs)


class OracleShardSource:
    """One rank's frame results from the CPU oracle, in the shape ShardExchange
    asks for: pack() and merge() on CPU tensors."""

    def __init__(self, oracle, frame, views):
        import torch
        self.oracle, self.frame, self.views = oracle, frame, views
        self.device = torch.device("cpu")

    def pack(self, block, vis_cap, key_cap):
        f = self.frame
        packed = numpy_exchange_pack(f["states"], f["visible"], f["payload"]["sort_keys"], self.views,
                                     vis_cap, key_cap)
        block.numpy().view(np.uint32)[:] = packed

    def merge(self, gathered, world, vis_cap, key_cap, vis_merged, key_merged, result):
        flags, max_vis, max_keys, view_offsets, merged_ranges, merged, keys = numpy_exchange_merge(
            self.oracle, gathered.numpy().view(np.uint32), world, self.views, vis_cap, key_cap)
        res = result.numpy().view(np.uint64)
        res[:] = 0
        res[0], res[1], res[2] = flags, max_vis, max_keys
        res[3] = int(view_offsets[-1])
        res[8: 8 + self.views + 1] = view_offsets
        res[8 + self.views + 1:] = merged_ranges.reshape(-1)
        if not flags:
            res[4] = keys.shape[0]
            vis_merged.numpy().view(np.uint32)[: merged.shape[0]] = merged
            key_merged.numpy().view(np.uint64)[: keys.shape[0]] = keys


# ---- synthetic depth pyramid (tests of the HZB occlusion branch) ----------------
def depth_pyramid(width, height, walls, background=1.0):
    """Mip chain (farthest depth per texel, 2x2 max reduction, extent >> mip) of a
    depth image that is `background` except for axis-aligned rectangles
    (x0, y0, x1, y1, depth) in texels."""
    img = np.full((height, width), background, np.float32)
    for x0, y0, x1, y1, d in walls:
        img[y0:y1, x0:x1] = np.float32(d)
    mips = [img]
    while mips[-1].shape[0] > 1 or mips[-1].shape[1] > 1:
        src = mips[-1]
        h, w = max(src.shape[0] >> 1, 1), max(src.shape[1] >> 1, 1)
        dst = np.empty((h, w), np.float32)
        for y in range(h):
            for x in range(w):
                dst[y, x] = src[min(2 * y, src.shape[0] - 1): 2 * y + 2, min(2 * x, src.shape[1] - 1): 2 * x + 2].max()
        mips.append(dst)
    return mips
