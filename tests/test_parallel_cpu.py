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
