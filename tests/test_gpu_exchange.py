"""The device side of the multi-GPU exchange on one GPU: the G-way draw-key merge
and the segmented copy that builds merged visible lists, fed (a) with synthetic
runs and (b) with two shard contexts whose results are concatenated by hand in
place of the all-gather; plus the real ShardExchange over RCCL with one rank."""
import os
import socket

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("g", [1, 2, 3, 8])
def test_merge_runs_kernel(oracle, g):
    import torch
    from vulkan_renderer_amd import HipScene
    rng = np.random.default_rng(40 + g)
    lengths = rng.integers(0, 5000, g)
    lengths[rng.integers(0, g)] = 0 if g > 1 else 300          # an empty run among them
    total = int(lengths.sum())
    keys = rng.permutation(np.arange(1, total * 3 + 1, dtype=np.uint64))[:total]
    keys = (keys << np.uint64(20)) | np.arange(total, dtype=np.uint64)   # unique
    runs, pos = [], 0
    for n in lengths:
        runs.append(np.sort(keys[pos:pos + n])[::-1].copy())
        pos += n
    expect = oracle.merge_runs_desc(runs)
    with HipScene(max_nodes=1, max_candidates=1, max_views=1) as hs:
        dev = [torch.from_numpy(r.view(np.int64)).cuda() for r in runs]
        out = torch.zeros(max(total, 1), dtype=torch.int64, device="cuda")
        hs.set_stream(torch.cuda.current_stream().cuda_stream)
        hs.merge_runs([d.data_ptr() for d in dev], np.asarray(lengths, np.uint64), out.data_ptr())
        hs.sync()
        got = out.cpu().numpy().view(np.uint64)[:total]
    assert (got == expect).all()


def test_two_shards_merge_to_the_whole_scene(oracle):
    import torch
    from helpers import oracle_frame, view_set
    from vulkan_renderer_amd import HipScene, synth
    from vulkan_renderer_amd.parallel import plan_visible_merge, wrap_device
    world, n = 2, 40000
    vs = synth.camera_views(cascades=2)
    planes, required = view_set(vs)
    v = planes.shape[0]
    shards, infos, vis_bufs, key_bufs = [], [], [], []
    stream = torch.cuda.current_stream().cuda_stream
    for rank in range(world):
        sc = synth.flat_scene(n, seed=9, first=rank * n, blend_fraction=0.3)
        sc.pos[:, :3] *= np.float32(0.3)
        hs = HipScene(max_nodes=n, max_candidates=n, max_views=v, source_index_base=rank * n)
        hs.set_stream(stream)
        hs.set_hierarchy(None, sc.level_offsets)
        hs.upload_trs(sc.pos, sc.quat, sc.scale)
        hs.set_candidates(None, sc.ident, sc.matbits, sc.bounds)
        hs.set_views(planes, required, vs.view)
        hs.frame()
        info = hs.frame_info()
        out = hs.device_outputs()
        packed = [wrap_device(out.visible + 4 * vi * out.visible_stride, st.visible_count, torch.int32)
                  for vi, st in enumerate(info.views)]
        vis_bufs.append(torch.cat(packed))
        key_bufs.append(wrap_device(out.transparent_keys, info.transparent_draw_count, torch.int64).clone())
        shards.append(hs), infos.append(info)
    # what the all-gather would deliver: every rank's packed list at a fixed stride
    stride = max(b.numel() for b in vis_bufs) + 7
    gathered = torch.zeros(world * stride, dtype=torch.int32, device="cuda")
    for r, b in enumerate(vis_bufs):
        gathered[r * stride: r * stride + b.numel()] = b
    ranges = np.stack([np.stack([st.execution_ranges for st in info.views]) for info in infos]).astype(np.uint64)
    desc, merged_ranges, view_offsets, _ = plan_visible_merge(ranges)
    desc[:, 0] += (np.arange(desc.shape[0], dtype=np.uint64) % np.uint64(world)) * np.uint64(stride)
    merged = torch.zeros(int(view_offsets[-1]) + 1, dtype=torch.int32, device="cuda")
    shards[0].concat_ranges(gathered.data_ptr(), merged.data_ptr(), desc)
    t_all = np.array([k.numel() for k in key_bufs], np.uint64)
    keys_merged = torch.zeros(int(t_all.sum()) + 1, dtype=torch.int64, device="cuda")
    shards[0].merge_runs([k.data_ptr() for k in key_bufs], t_all, keys_merged.data_ptr())
    shards[0].sync()

    sc = synth.flat_scene(world * n, seed=9, blend_fraction=0.3)
    sc.pos[:, :3] *= np.float32(0.3)
    whole = oracle_frame(oracle, sc, planes, required, vs.view, vs.projection)
    got_keys = keys_merged.cpu().numpy().view(np.uint64)[: int(t_all.sum())]
    assert got_keys.shape[0] > 100
    assert (got_keys == whole["payload"]["sort_keys"]).all()
    got_vis = merged.cpu().numpy().view(np.uint32)
    for vi in range(v):
        st = whole["states"][vi]
        assert (merged_ranges[vi] == st["execution_ranges"]).all()
        count = int(st["visible_count"])
        off = int(view_offsets[vi])
        assert (got_vis[off: off + count] == whole["visible"][vi, :count]).all(), f"view {vi}"
    for hs in shards:
        hs.close()


def test_exchange_single_rank_over_rccl(oracle):
    """One rank, backend nccl (RCCL): the real HipShardSource plumbing end to end."""
    import torch
    import torch.distributed as dist
    from helpers import oracle_frame, view_set
    from vulkan_renderer_amd import HipScene, synth
    from vulkan_renderer_amd.parallel import ShardExchange
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(_free_port())
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        n = 30000
        vs = synth.camera_views(cascades=1)
        planes, required = view_set(vs)
        sc = synth.flat_scene(n, seed=4, blend_fraction=0.3)
        sc.pos[:, :3] *= np.float32(0.3)
        ref = oracle_frame(oracle, sc, planes, required, vs.view, vs.projection)
        with HipScene(max_nodes=n, max_candidates=n, max_views=planes.shape[0]) as hs:
            hs.set_stream(torch.cuda.current_stream().cuda_stream)
            hs.set_hierarchy(None, sc.level_offsets)
            hs.upload_trs(sc.pos, sc.quat, sc.scale)
            hs.set_candidates(None, sc.ident, sc.matbits, sc.bounds)
            hs.set_views(planes, required, vs.view)
            ex = ShardExchange(hs, 1, 0, planes.shape[0])
            for _ in range(2):
                hs.mark_all(0x02)
                hs.frame()
                ex.run()
            torch.cuda.synchronize()
            keys = ex.key_merged.cpu().numpy().view(np.uint64)[: ex.transparent_total]
            assert (keys == ref["payload"]["sort_keys"]).all()
            vis = ex.vis_merged.cpu().numpy().view(np.uint32)
            for vi in range(planes.shape[0]):
                count = int(ref["states"][vi]["visible_count"])
                off = int(ex.view_offsets[vi])
                assert (vis[off: off + count] == ref["visible"][vi, :count]).all()
    finally:
        dist.destroy_process_group()
