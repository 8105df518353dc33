"""The multi-rank exchange on CPU: two `gloo` processes, each owning half of a
scene, run the same ShardExchange code the GPU ranks run (block pack, one
all-gather, merge, capacity growth in finish()). Each rank's frame results come
from the CPU oracle on its shard and the block layout is restated in numpy
(helpers.py); the merged result must equal the oracle's result on the whole
scene."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shard_range_covers_everything():
    from vulkan_renderer_amd.parallel import shard_range
    for total in (0, 1, 7, 1024, 1000003):
        for world in (1, 2, 3, 8):
            edges = [shard_range(total, world, r) for r in range(world)]
            assert edges[0][0] == 0 and edges[-1][1] == total
            for a, b in zip(edges, edges[1:]):
                assert a[1] == b[0]
            sizes = [e - b for b, e in edges]
            assert max(sizes) - min(sizes) <= 1


def test_plan_visible_merge_matches_concatenation():
    from vulkan_renderer_amd.parallel import apply_descriptors, plan_visible_merge
    rng = np.random.default_rng(5)
    g, v = 3, 4
    counts = rng.integers(0, 50, (g, v, 4)).astype(np.uint64)
    ranges = np.zeros((g, v, 4, 2), np.uint64)
    ranges[..., 1] = counts
    ranges[..., 1:, 0] = np.cumsum(counts, axis=2)[..., :-1]
    stride = int(counts.sum(axis=(1, 2)).max()) + 5
    gathered = np.zeros((g, stride), np.uint32)
    lists = {}
    for r in range(g):
        pos = 0
        for vi in range(v):
            for b in range(4):
                c = int(counts[r, vi, b])
                vals = rng.integers(0, 1 << 30, c).astype(np.uint32)
                gathered[r, pos:pos + c] = vals
                lists[(r, vi, b)] = vals
                pos += c
    for cap in (0, 60):
        desc, merged_ranges, view_offsets, overflow = plan_visible_merge(ranges, cap)
        out = apply_descriptors(gathered, stride, desc, g, int(view_offsets[-1]))
        for vi in range(v):
            for b in range(4):
                expect = np.concatenate([lists[(r, vi, b)] for r in range(g)])
                total = expect.shape[0]
                if cap:
                    expect = expect[:cap]
                off = int(view_offsets[vi] + merged_ranges[vi, b, 0])
                cnt = int(merged_ranges[vi, b, 1])
                assert cnt == expect.shape[0]
                assert (out[off:off + cnt] == expect).all()
                if cap:
                    assert total - cnt >= 0
            if cap:
                dropped = sum(max(0, int(counts[:, vi, b].sum()) - cap) for b in range(4))
                assert int(overflow[vi]) == dropped


def _worker(rank, world, port, n_per_rank, result_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch
    import torch.distributed as dist
    from helpers import OracleShardSource, oracle_frame, view_set
    from oracle import pyoracle
    from vulkan_renderer_amd import synth
    from vulkan_renderer_amd.parallel import ShardExchange
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        oracle = pyoracle.Oracle()
        vs = synth.camera_views(cascades=2)
        planes, required = view_set(vs)
        sc = synth.flat_scene(n_per_rank, seed=9, first=rank * n_per_rank, blend_fraction=0.3)
        sc.pos[:, :3] *= np.float32(0.3)  # denser: every rank contributes to every bucket
        frame = oracle_frame(oracle, sc, planes, required, vs.view, vs.projection,
                             index_base=rank * n_per_rank)
        # small starting capacities: the first finish() has to grow and repeat
        ex = ShardExchange(OracleShardSource(oracle, frame, planes.shape[0]), world, rank,
                           planes.shape[0], device=torch.device("cpu"), vis_cap=64, key_cap=16)
        ex.run().finish()
        regrown = ex.regrown
        ex.run().finish()  # second frame reuses the grown buffers
        assert ex.regrown == regrown >= 1
        np.savez(os.path.join(result_dir, f"rank{rank}.npz"),
                 vis=ex.vis_merged.numpy()[: int(ex.view_offsets[-1])].view(np.uint32),
                 keys=ex.key_merged.numpy().view(np.uint64)[: ex.transparent_total],
                 merged_ranges=ex.merged_ranges, view_offsets=ex.view_offsets)
    finally:
        dist.destroy_process_group()


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_two_rank_exchange_equals_single_process(oracle, tmp_path):
    import torch.multiprocessing as mp
    from helpers import oracle_frame, view_set
    from vulkan_renderer_amd import synth
    world, n_per_rank = 2, 3000
    mp.spawn(_worker, args=(world, _free_port(), n_per_rank, str(tmp_path)), nprocs=world, join=True)

    vs = synth.camera_views(cascades=2)
    planes, required = view_set(vs)
    sc = synth.flat_scene(world * n_per_rank, seed=9, blend_fraction=0.3)
    sc.pos[:, :3] *= np.float32(0.3)
    whole = oracle_frame(oracle, sc, planes, required, vs.view, vs.projection)
    results = [np.load(tmp_path / f"rank{r}.npz") for r in range(world)]
    for name in ("vis", "keys", "merged_ranges", "view_offsets"):
        assert (results[0][name] == results[1][name]).all(), f"ranks disagree on {name}"
    r0 = results[0]
    assert r0["keys"].shape[0] > 50
    assert (r0["keys"] == whole["payload"]["sort_keys"]).all(), "merged draw keys"
    for v in range(planes.shape[0]):
        st = whole["states"][v]
        assert (r0["merged_ranges"][v] == st["execution_ranges"]).all(), f"view {v} ranges"
        count = int(st["visible_count"])
        assert count > 20
        off = int(r0["view_offsets"][v])
        assert (r0["vis"][off: off + count] == whole["visible"][v, :count]).all(), f"view {v} list"


def test_sharded_hierarchy_union_equals_the_whole_scene(oracle):
    """A depth-8 hierarchy cut into whole subtrees (vkrhip_shard_hierarchy, top level
    replicated): every shard's oracle frame, put together, is the frame of the
    whole scene - world matrices of every node, per-view visible sets and the
    merged draw order. Rows are numbered shard after shard (the rows of the
    replicated nodes belong to shard 0), which is the row order of the whole scene."""
    from helpers import oracle_frame, view_set
    from vulkan_renderer_amd import synth
    from vulkan_renderer_amd.scene import shard_hierarchy, shard_local_hierarchy
    sc = synth.deep_scene(60000, depth=8, seed=5, fanout=4, roots=6, blend_fraction=0.2)
    vs = synth.camera_views(cascades=2, eye=(0.0, 0.0, 400.0), target=(0.0, 0.0, 0.0), far=1500.0,
                            fov_deg=50.0)
    planes, required = view_set(vs)
    views = planes.shape[0]
    shards = 3
    ranges = shard_hierarchy(sc.parent, sc.level_offsets, shards, replicated_levels=1)
    levels = sc.level_offsets.shape[0] - 1
    # the ranges tile every level below the replicated one, in order
    for l in range(1, levels):
        assert int(ranges[0, l, 0]) == int(sc.level_offsets[l])
        assert int(ranges[-1, l, 1]) == int(sc.level_offsets[l + 1])
        assert (ranges[1:, l, 0] == ranges[:-1, l, 1]).all()
    sizes = (ranges[:, :, 1] - ranges[:, :, 0]).sum(axis=1).astype(np.int64)
    assert sizes.max() < 1.3 * sizes.min(), "about equal shards"

    world_whole = np.zeros((sc.n, 16), np.float32)
    frames, row_node, bases = [], [], []
    base = 0
    for s in range(shards):
        nodes, local_parent, offsets = shard_local_hierarchy(sc.parent, ranges[s])
        # rows: shard 0 draws every node it holds, the others only what they own
        owned = np.ones(nodes.shape[0], bool) if s == 0 else nodes >= sc.level_offsets[1]
        rows = np.flatnonzero(owned).astype(np.uint32)
        local = synth.Scene(local_parent, offsets, sc.pos[nodes], sc.quat[nodes], sc.scale[nodes],
                            sc.ident[nodes[rows]], sc.matbits[nodes[rows]], sc.bounds[nodes[rows]],
                            node=rows)
        f = oracle_frame(oracle, local, planes, required, vs.view, vs.projection, index_base=base)
        world_whole[nodes] = f["world"]
        frames.append(f)
        row_node.append(nodes[rows])
        bases.append(base)
        base += rows.shape[0]
    row_node = np.concatenate(row_node)
    assert row_node.shape[0] == sc.n and np.unique(row_node).shape[0] == sc.n

    whole = synth.Scene(sc.parent, sc.level_offsets, sc.pos, sc.quat, sc.scale, sc.ident[row_node],
                        sc.matbits[row_node], sc.bounds[row_node], node=row_node.astype(np.uint32))
    ref = oracle_frame(oracle, whole, planes, required, vs.view, vs.projection)
    assert world_whole.view(np.uint32).tobytes() == ref["world"].view(np.uint32).tobytes()
    # visible lists: per (view, bucket) the shards' ranges one after another
    for v in range(views):
        st = ref["states"][v]
        for b in range(4):
            off, cnt = (int(x) for x in st["execution_ranges"][b])
            parts = []
            for f in frames:
                o, c = (int(x) for x in f["states"][v]["execution_ranges"][b])
                parts.append(f["visible"][v, o: o + c])
            assert (np.concatenate(parts) == ref["visible"][v, off: off + cnt]).all(), (v, b)
    # draw order: the merge of the shards' descending key runs
    merged = oracle.merge_runs_desc([f["payload"]["sort_keys"] for f in frames])
    assert merged.shape[0] > 100 and (merged == ref["payload"]["sort_keys"]).all()


def test_shard_hierarchy_refuses_other_numberings():
    from vulkan_renderer_amd import _native as N
    from vulkan_renderer_amd.scene import shard_hierarchy
    parent = np.array([0xFFFFFFFF, 0xFFFFFFFF, 1, 0, 1], np.uint32)   # parents do not ascend
    with pytest.raises(N.VkrHipError) as e:
        shard_hierarchy(parent, np.array([0, 2, 5], np.uint64), 2)
    assert e.value.code == N.ERR_STATE
    # level offsets that do not ascend, or run past the node count, are refused
    # before anything is indexed by them
    parent = np.array([0xFFFFFFFF, 0xFFFFFFFF, 0, 0, 1], np.uint32)
    for offsets in ([0, 9, 5], [0, 4, 2, 5], [0, 1 << 40, 5]):
        with pytest.raises(N.VkrHipError) as e:
            shard_hierarchy(parent, np.array(offsets, np.uint64), 2)
        assert e.value.code == N.ERR_ARG


def test_sharded_deep_scene_is_the_whole_scene_cut_at_the_roots():
    """synth.sharded_deep_scene (what bench.py gives every rank of a multi-GPU
    hierarchy run): the shards partition the nodes of the whole scene, carry its
    node data, and their local parents name the same nodes."""
    from vulkan_renderer_amd import synth
    whole = synth.deep_scene(4 * 20000, depth=6, seed=9, fanout=5, roots=8)
    seen = np.zeros(whole.n, np.int32)
    base = 0
    for r in range(4):
        sh = synth.sharded_deep_scene(20000, 4, r, depth=6, seed=9, fanout=5, roots=8)
        g = sh.meta["global_nodes"].astype(np.int64)
        seen[g] += 1
        assert sh.meta["index_base"] == base and sh.meta["total"] == whole.n
        base += sh.n
        assert 0.8 * 20000 < sh.n < 1.2 * 20000
        for name in ("pos", "quat", "scale", "ident", "matbits", "bounds"):
            assert (getattr(sh, name) == getattr(whole, name)[g]).all(), name
        has = sh.parent != 0xFFFFFFFF
        assert (has == (whole.parent[g] != 0xFFFFFFFF)).all()
        assert (g[sh.parent[has]] == whole.parent[g][has]).all()
        assert int(sh.level_offsets[-1]) == sh.n
    assert (seen == 1).all()
