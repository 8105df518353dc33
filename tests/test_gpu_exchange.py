"""The device side of the multi-GPU exchange on one GPU: the G-way draw-key merge
and the segmented copy that builds merged visible lists, fed (a) with synthetic
runs and (b) with two shard contexts whose results are concatenated by hand in
place of the all-gather; the device-planned exchange (pack, plan, merge) checked
against its numpy restatement and the whole-scene oracle, including blocks that
are too small; plus the real ShardExchange over RCCL with one rank."""
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
        torch.cuda.synchronize()   # torch's default stream is not adopted: order the fill by hand
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
    t_all = np.array([k.numel() for k in key_bufs], np.uint64)
    keys_merged = torch.zeros(int(t_all.sum()) + 1, dtype=torch.int64, device="cuda")
    torch.cuda.synchronize()       # the library's stream is its own: order torch's fills by hand
    shards[0].concat_ranges(gathered.data_ptr(), merged.data_ptr(), desc)
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


def _shard_scene(rank, n, seed=9, spread=0.3):
    from vulkan_renderer_amd import synth
    sc = synth.flat_scene(n, seed=seed, first=rank * n, blend_fraction=0.3)
    sc.pos[:, :3] *= np.float32(spread)
    return sc


# the last case is the shape of an eight-GPU run of the benchmark: 1M instances per shard
@pytest.mark.parametrize("world,cascades,n", [(2, 2, 30000), (3, 0, 30000), (8, 0, 1 << 20)])
def test_device_planned_exchange(oracle, world, cascades, n):
    _planned_exchange(oracle, world, cascades, n)


@pytest.mark.parametrize("seed", range(int(__import__("os").environ.get("VKR_RANDOM_SEEDS", "6"))))
def test_random_device_planned_exchange(oracle, seed):
    """Random shard counts (1 to 8), view counts, shard sizes and densities."""
    rng = np.random.default_rng(77000 + seed)
    _planned_exchange(oracle, int(rng.integers(1, 9)), int(rng.integers(0, 5)),
                      int(rng.choice([700, 1024, 5000, 33000])), seed=200 + seed,
                      spread=float(rng.choice([0.1, 0.3, 0.6])), need_some=False)


def _planned_exchange(oracle, world, cascades, n, seed=9, spread=0.3, need_some=True):
    import torch
    from helpers import numpy_exchange_merge, numpy_exchange_pack, oracle_frame, view_set
    from vulkan_renderer_amd import HipScene, synth
    from vulkan_renderer_amd.parallel import exchange_block_words, exchange_result_words
    vs = synth.camera_views(cascades=cascades)
    planes, required = view_set(vs)
    v = planes.shape[0]
    stream = torch.cuda.current_stream().cuda_stream
    shards, refs = [], []
    for rank in range(world):
        sc = _shard_scene(rank, n, seed, spread)
        hs = HipScene(max_nodes=n, max_candidates=n, max_views=v, source_index_base=rank * n)
        hs.set_stream(stream)
        hs.set_hierarchy(None, sc.level_offsets)
        hs.upload_trs(sc.pos, sc.quat, sc.scale)
        hs.set_candidates(None, sc.ident, sc.matbits, sc.bounds)
        hs.set_views(planes, required, vs.view)
        hs.frame()
        shards.append(hs)
        refs.append(oracle_frame(oracle, sc, planes, required, vs.view, vs.projection, index_base=rank * n))
    need_vis = max(int(sum(int(st["visible_count"]) for st in ref["states"])) for ref in refs)
    need_keys = max(ref["payload"]["sort_keys"].shape[0] for ref in refs)
    assert not need_some or (need_vis > 200 and need_keys > 50)

    sc = synth.flat_scene(world * n, seed=seed, blend_fraction=0.3)
    sc.pos[:, :3] *= np.float32(spread)
    whole = oracle_frame(oracle, sc, planes, required, vs.view, vs.projection)

    GUARD = 0x5a5a5a5a
    for vis_cap, key_cap in ((need_vis + 13, need_keys + 3), (need_vis, need_keys),
                             (need_vis // 2, need_keys + 3), (need_vis + 13, need_keys // 3)):
        words = exchange_block_words(v, vis_cap, key_cap)
        assert words == shards[0].exchange_block_words(v, vis_cap, key_cap)
        assert exchange_result_words(v) == shards[0].exchange_result_words(v)
        gathered = torch.full((world * words + 4,), GUARD, dtype=torch.int32, device="cuda")
        merged_vis = torch.full((world * vis_cap + 4,), GUARD, dtype=torch.int32, device="cuda")
        merged_keys = torch.full((world * key_cap + 4,), GUARD, dtype=torch.int64, device="cuda")
        result = torch.zeros(exchange_result_words(v) + 1, dtype=torch.int64, device="cuda")
        result[-1] = GUARD
        # every context enqueues on a stream of its own here (torch's default stream is
        # not adopted): the fills, the shards' packs and the merge are ordered by hand,
        # as the collective orders them between ranks
        torch.cuda.synchronize()
        for rank, hs in enumerate(shards):
            hs.exchange_pack(gathered.data_ptr() + 4 * rank * words, vis_cap, key_cap)
        torch.cuda.synchronize()
        shards[0].exchange_merge(gathered.data_ptr(), world, vis_cap, key_cap, merged_vis.data_ptr(),
                                 merged_keys.data_ptr(), result.data_ptr())
        torch.cuda.synchronize()
        got_blocks = gathered.cpu().numpy().view(np.uint32)
        assert (got_blocks[world * words:] == GUARD).all()
        # every block equals the numpy restatement of the layout, byte for byte
        # (words a short list leaves unused keep the fill value)
        for rank, ref in enumerate(refs):
            expect = numpy_exchange_pack(ref["states"], ref["visible"], ref["payload"]["sort_keys"], v,
                                         vis_cap, key_cap)
            got = got_blocks[rank * words: (rank + 1) * words]
            written = got != GUARD
            assert (got[written] == expect[written]).all(), f"rank {rank} block"
            assert (expect[~written] == 0).all(), f"rank {rank}: a word that should be written is not"
        flags, max_vis, max_keys, view_offsets, merged_ranges, exp_vis, exp_keys = numpy_exchange_merge(
            oracle, np.where(got_blocks[: world * words] == GUARD, 0, got_blocks[: world * words]),
            world, v, vis_cap, key_cap)
        res = result.cpu().numpy().view(np.uint64)
        assert int(res[-1]) == GUARD
        assert int(res[0]) == flags == ((1 if vis_cap < need_vis else 0) | (2 if key_cap < need_keys else 0))
        assert int(res[1]) == max_vis == need_vis and int(res[2]) == max_keys == need_keys
        assert (merged_vis.cpu().numpy().view(np.uint32)[world * vis_cap:] == GUARD).all()
        assert (merged_keys.cpu().numpy()[world * key_cap:] == GUARD).all()
        if flags:
            continue
        assert (res[8: 8 + v + 1] == view_offsets).all()
        assert (res[8 + v + 1: 8 + v + 1 + v * 8].reshape(v, 4, 2) == merged_ranges).all()
        total_keys = int(res[4])
        assert int(res[3]) == int(view_offsets[-1]) and total_keys == exp_keys.shape[0]
        got_vis = merged_vis.cpu().numpy().view(np.uint32)
        assert (got_vis[: exp_vis.shape[0]] == exp_vis).all()
        got_keys = merged_keys.cpu().numpy().view(np.uint64)[:total_keys]
        assert (got_keys == exp_keys).all()
        # and both equal the single-process result on the whole scene
        assert (got_keys == whole["payload"]["sort_keys"]).all()
        for vi in range(v):
            st = whole["states"][vi]
            assert (merged_ranges[vi] == st["execution_ranges"]).all()
            count, off = int(st["visible_count"]), int(view_offsets[vi])
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
            ex = ShardExchange(hs, 1, 0, planes.shape[0], vis_cap=256, key_cap=64)
            for _ in range(3):
                hs.mark_all(0x02)
                hs.frame()
                ex.run()
            ex.finish()
            assert ex.regrown == 1
            for _ in range(2):
                hs.mark_all(0x02)
                hs.frame()
                ex.run().finish()
            assert ex.regrown == 1
            keys = ex.key_merged.cpu().numpy().view(np.uint64)[: ex.transparent_total]
            assert (keys == ref["payload"]["sort_keys"]).all()
            vis = ex.vis_merged.cpu().numpy().view(np.uint32)
            for vi in range(planes.shape[0]):
                count = int(ref["states"][vi]["visible_count"])
                off = int(ex.view_offsets[vi])
                assert (vis[off: off + count] == ref["visible"][vi, :count]).all()
    finally:
        dist.destroy_process_group()


def _gloo_worker(rank, world, port, n, result_dir):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    sys.path.insert(0, os.path.join(root, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch
    import torch.distributed as dist
    from helpers import view_set
    from vulkan_renderer_amd import HipScene, synth
    from vulkan_renderer_amd.parallel import ShardExchange
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        vs = synth.camera_views(cascades=2)
        planes, required = view_set(vs)
        sc = _shard_scene(rank, n)
        with HipScene(max_nodes=n, max_candidates=n, max_views=planes.shape[0],
                      source_index_base=rank * n) as hs:
            hs.set_hierarchy(None, sc.level_offsets)
            hs.upload_trs(sc.pos, sc.quat, sc.scale)
            hs.set_candidates(None, sc.ident, sc.matbits, sc.bounds)
            hs.set_views(planes, required, vs.view)
            ex = ShardExchange(hs, world, rank, planes.shape[0], vis_cap=128, key_cap=32)
            for _ in range(3):
                hs.mark_all(0x02)
                hs.frame()
                ex.run()
            ex.finish()
            np.savez(os.path.join(result_dir, f"rank{rank}.npz"),
                     vis=ex.vis_merged.cpu().numpy().view(np.uint32)[: ex.visible_total],
                     keys=ex.key_merged.cpu().numpy().view(np.uint64)[: ex.transparent_total],
                     merged_ranges=ex.merged_ranges, view_offsets=ex.view_offsets,
                     regrown=np.array([ex.regrown]))
    finally:
        dist.destroy_process_group()


def test_two_processes_one_gpu_over_gloo(oracle, tmp_path):
    """Two ranks as two processes (both on this GPU, collective over gloo): the
    whole per-rank flow of a multi-GPU run except the RCCL transport."""
    import torch.multiprocessing as mp
    from helpers import oracle_frame, view_set
    from vulkan_renderer_amd import synth
    world, n = 2, 20000
    mp.spawn(_gloo_worker, args=(world, _free_port(), n, str(tmp_path)), nprocs=world, join=True)
    vs = synth.camera_views(cascades=2)
    planes, required = view_set(vs)
    sc = synth.flat_scene(world * n, seed=9, blend_fraction=0.3)
    sc.pos[:, :3] *= np.float32(0.3)
    whole = oracle_frame(oracle, sc, planes, required, vs.view, vs.projection)
    results = [np.load(tmp_path / f"rank{r}.npz") for r in range(world)]
    for name in ("vis", "keys", "merged_ranges", "view_offsets", "regrown"):
        assert (results[0][name] == results[1][name]).all(), f"ranks disagree on {name}"
    r0 = results[0]
    assert int(r0["regrown"][0]) >= 1
    assert (r0["keys"] == whole["payload"]["sort_keys"]).all()
    for v in range(planes.shape[0]):
        st = whole["states"][v]
        assert (r0["merged_ranges"][v] == st["execution_ranges"]).all()
        count, off = int(st["visible_count"]), int(r0["view_offsets"][v])
        assert (r0["vis"][off: off + count] == whole["visible"][v, :count]).all()


@pytest.mark.parametrize("tail", ["0", "2"])
def test_exchange_in_a_process_full_of_streams(tail):
    """The process shape that stalled a context while its exchange stream had the
    highest priority: eight more streams of default priority and two of high
    priority alive in the process (what PyTorch's stream pools leave behind), the
    frame's sort - or its whole tail - on the second stream, thirteen frames with
    an exchange each. A fresh process that ends by its own watchdog (exit code 3)
    if it stalls; tools/exchange_hang_probe.py."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, VKRHIP_OVERLAP_SORT="2", VKRHIP_OVERLAP_TAIL=tail, PROBE_STREAMS="8",
               PROBE_PRIORITY_STREAMS="2")
    run = subprocess.run([sys.executable, os.path.join(root, "tools", "exchange_hang_probe.py"), "20"],
                         env=env, capture_output=True, text=True, timeout=90)
    assert run.returncode == 0 and "COMPLETED 13 frames" in run.stdout, run.stdout + run.stderr


@pytest.mark.parametrize("streams", ["default", "sort", "tail"])
@pytest.mark.parametrize("with_comm", [False, True])
def test_native_exchange_single_rank(oracle, with_comm, streams, monkeypatch):
    """The exchange inside the library (pack, RCCL all-gather or device copy,
    merge on the exchange stream, two buffer sets): five frames in flight without a
    host wait, growth from too-small blocks, result equal to the oracle. "sort":
    the frame's sort and emission on the second stream (VKRHIP_OVERLAP_SORT=2),
    "tail": scan and gather as well (VKRHIP_OVERLAP_TAIL=2) - the pack then follows
    them there and three streams are in play, which large scenes do by default."""
    import torch
    if streams != "default":
        monkeypatch.setenv("VKRHIP_OVERLAP_SORT", "2")
        monkeypatch.setenv("VKRHIP_OVERLAP_TAIL", "2" if streams == "tail" else "0")
    from helpers import oracle_frame, view_set
    from vulkan_renderer_amd import HipScene, synth
    from vulkan_renderer_amd.parallel import RcclExchange
    n = 30000
    vs = synth.camera_views(cascades=1)
    planes, required = view_set(vs)
    sc = synth.flat_scene(n, seed=4, blend_fraction=0.3)
    sc.pos[:, :3] *= np.float32(0.3)
    ref = oracle_frame(oracle, sc, planes, required, vs.view, vs.projection)
    with HipScene(max_nodes=n, max_candidates=n, max_views=planes.shape[0]) as hs:
        hs.set_hierarchy(None, sc.level_offsets)
        hs.upload_trs(sc.pos, sc.quat, sc.scale)
        hs.set_candidates(None, sc.ident, sc.matbits, sc.bounds)
        hs.set_views(planes, required, vs.view)
        if with_comm:
            hs.comm_init(hs.comm_unique_id(), 1, 0)      # a one-rank RCCL communicator
        ex = RcclExchange(hs, 1, 0, planes.shape[0], vis_cap=256, key_cap=64)
        for _ in range(3):
            hs.mark_all(0x02)
            hs.frame()
            ex.run()
        ex.finish()
        assert ex.regrown == 1
        for _ in range(5):
            hs.mark_all(0x02)
            hs.frame()
            ex.run()
        ex.finish()
        assert ex.regrown == 1
        keys = ex.key_merged.cpu().numpy().view(np.uint64)[: ex.transparent_total]
        assert keys.shape[0] > 50 and (keys == ref["payload"]["sort_keys"]).all()
        vis = ex.vis_merged.cpu().numpy().view(np.uint32)
        for vi in range(planes.shape[0]):
            st = ref["states"][vi]
            assert (ex.merged_ranges[vi] == st["execution_ranges"]).all()
            count, off = int(st["visible_count"]), int(ex.view_offsets[vi])
            assert (vis[off: off + count] == ref["visible"][vi, :count]).all()


def test_exchange_refuses_a_publication(oracle):
    """With a publication the visible lists hold rank-local packed row indices,
    which cannot be concatenated across ranks: the exchange says so."""
    import torch
    from helpers import view_set
    from test_gpu_pack_encode import tables
    from vulkan_renderer_amd import HipScene, _native as N, synth
    sc = synth.flat_scene(2000, seed=3)
    vs = synth.camera_views()
    planes, required = view_set(vs)
    geom, sub, mat = tables(sc, np.random.default_rng(1))
    with HipScene(max_nodes=sc.n, max_candidates=sc.n, max_views=1) as hs:
        hs.set_hierarchy(None, sc.level_offsets)
        hs.upload_trs(sc.pos, sc.quat, sc.scale)
        hs.set_candidates(None, sc.ident, sc.matbits, sc.bounds)
        hs.set_views(planes, required, vs.view)
        hs.set_publication(geom, sub, mat)
        hs.frame()
        block = torch.zeros(hs.exchange_block_words(1, 4096, 512), dtype=torch.int32, device="cuda")
        with pytest.raises(N.VkrHipError) as e:
            hs.exchange_pack(block.data_ptr(), 4096, 512)
        assert e.value.code == N.ERR_STATE


def test_overflow_after_the_scene_advanced_is_not_repeated(oracle):
    """finish() repeats an exchange whose blocks were too small by packing again;
    once a newer frame has been enqueued that would carry the newer frame's lists,
    so it raises with the sizes that would have fitted instead."""
    from helpers import view_set
    from vulkan_renderer_amd import HipScene, synth
    from vulkan_renderer_amd.parallel import ExchangeOverflow, RcclExchange
    n = 20000
    vs = synth.camera_views()
    planes, required = view_set(vs)
    sc = synth.flat_scene(n, seed=4, blend_fraction=0.3)
    sc.pos[:, :3] *= np.float32(0.3)
    with HipScene(max_nodes=n, max_candidates=n, max_views=1) as hs:
        hs.set_hierarchy(None, sc.level_offsets)
        hs.upload_trs(sc.pos, sc.quat, sc.scale)
        hs.set_candidates(None, sc.ident, sc.matbits, sc.bounds)
        hs.set_views(planes, required, vs.view)
        ex = RcclExchange(hs, 1, 0, 1, vis_cap=64, key_cap=16)
        hs.frame()
        ex.run()
        hs.mark_all(0x02)
        hs.frame()          # frame k+1 is enqueued before frame k's exchange is finished
        with pytest.raises(ExchangeOverflow) as e:
            ex.finish()
        assert e.value.need_visible > 64 and e.value.need_keys > 16
        ex.run()            # the newer frame's own exchange sizes the blocks
        ex.finish()
        assert ex.regrown == 1


@pytest.mark.parametrize("workload,n", [("deep16m", 200000), ("flat32m", 150000)])
def test_bench_two_ranks_one_gpu(oracle, tmp_path, workload, n):
    """bench.py --gpus 2 as two fresh processes on this GPU (collective over gloo):
    the line rank 0 prints, and the merged lists and draw keys it leaves behind
    against the CPU oracle on the UNSHARDED scene. A hierarchy's rows are numbered
    shard after shard, so its lists are compared as sets per (view, bucket) after
    the renumbering; a flat scene keeps its numbering and is compared as it is."""
    import json
    import os
    import subprocess
    import sys
    from helpers import oracle_frame, view_set
    from vulkan_renderer_amd import synth
    import bench
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    world, port, dump = 2, _free_port(), str(tmp_path / "merged.npz")
    procs = []
    for rank in range(world):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen(
            [sys.executable, os.path.join(root, "bench.py"), "--gpus", str(world), "--backend", "gloo",
             "--workload", workload, "--instances", str(n), "--steps", "3", "--warmup", "2",
             "--no-cpu-baseline", "--dump-merged", dump],
            env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=240) for p in procs]
    assert all(p.returncode == 0 for p in procs), [o[1][-2000:] for o in outs]
    line = json.loads([l for l in outs[0][0].splitlines() if l.startswith("{")][-1])
    spec = bench.WORKLOADS[workload]
    vs = synth.camera_views(cascades=spec["cascades"])
    planes, required = view_set(vs)
    if spec["depth"] > 1:
        whole = synth.deep_scene(n * world, depth=spec["depth"], seed=2)
        new_id = np.zeros(whole.n, np.uint64)
        total = 0
        for r in range(world):
            shard = synth.sharded_deep_scene(n, world, r, depth=spec["depth"], seed=2)
            nodes = shard.meta["global_nodes"].astype(np.int64)
            new_id[nodes] = shard.meta["index_base"] + np.arange(nodes.shape[0], dtype=np.uint64)
            total = shard.meta["total"]
    else:
        whole = synth.flat_scene(n * world, seed=1)
        new_id = np.arange(whole.n, dtype=np.uint64)
        total = whole.n
    assert line["n_gpus"] == world and line["config"]["instances_total"] == total
    assert line["config"]["ranks_seen"]["torch_distributed"] == world
    assert line["ms_per_step_ranks"]["max"] >= line["ms_per_step_ranks"]["min"] > 0
    assert line["roofline"]["frac"] > 0 and line["exchange"]["bytes_per_rank"] > 0
    ref = oracle_frame(oracle, whole, planes, required, vs.view, vs.projection)
    got = np.load(dump)
    for v in range(planes.shape[0]):
        st = ref["states"][v]
        assert (got["merged_ranges"][v][:, 1] == st["execution_ranges"][:, 1]).all(), f"view {v} counts"
        for b in range(4):
            off, cnt = (int(x) for x in st["execution_ranges"][b])
            want = new_id[ref["visible"][v, off: off + cnt].astype(np.int64)]
            goff = int(got["view_offsets"][v]) + int(got["merged_ranges"][v][b][0])
            have = got["vis"][goff: goff + cnt].astype(np.uint64)
            assert (np.diff(have.astype(np.int64)) > 0).all(), f"view {v} bucket {b} not ascending"
            assert (np.sort(want) == have).all(), f"view {v} bucket {b}"
    keys = ref["payload"]["sort_keys"]
    want = (keys & np.uint64(0xFFFFFFFF00000000)) | (new_id[(keys & np.uint64(0xFFFFFFFF)).astype(np.int64) - 1]
                                                      + np.uint64(1))
    assert got["keys"].shape == want.shape and (np.diff(got["keys"].astype(np.float64)) <= 0).all()
    assert (np.sort(got["keys"]) == np.sort(want)).all(), "merged draw keys"
