"""vkrhip_exchange_merge_static - the exchange merge for callers that cannot
allocate - against vkrhip_exchange_merge on the same gathered blocks: merged
visible lists, merged draw keys and result words must be identical, on two calls
in a row, the second with a smaller input (fewer ranks, fewer views, shorter
lists), so that what the first call left in the shared planning tables would
show. Limits beyond the fixed tables are refused with a clear error."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

GUARD = 0x5A5A5A5A


def _gathered_blocks(world, cascades, n, seed):
    """`world` shard contexts run one frame each and pack their blocks side by
    side, as the all-gather would leave them. Returns (contexts, gathered tensor,
    views, vis_cap, key_cap)."""
    import torch
    from helpers import view_set
    from vulkan_renderer_amd import HipScene, synth
    vs = synth.camera_views(cascades=cascades)
    planes, required = view_set(vs)
    views = planes.shape[0]
    shards = []
    for rank in range(world):
        sc = synth.flat_scene(n, seed=seed, first=rank * n, blend_fraction=0.3)
        sc.pos[:, :3] *= np.float32(0.3)
        hs = HipScene(max_nodes=n, max_candidates=n, max_views=views, source_index_base=rank * n)
        hs.set_hierarchy(None, sc.level_offsets)
        hs.upload_trs(sc.pos, sc.quat, sc.scale)
        hs.set_candidates(None, sc.ident, sc.matbits, sc.bounds)
        hs.set_views(planes, required, vs.view)
        hs.frame()
        shards.append(hs)
    infos = [hs.frame_info() for hs in shards]
    vis_cap = max(sum(st.visible_count for st in info.views) for info in infos) + 5
    key_cap = max(info.transparent_draw_count for info in infos) + 3
    assert vis_cap > 100 and key_cap > 20
    words = shards[0].exchange_block_words(views, vis_cap, key_cap)
    gathered = torch.zeros(world * words, dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    for rank, hs in enumerate(shards):
        hs.exchange_pack(gathered.data_ptr() + 4 * rank * words, vis_cap, key_cap)
        hs.sync()
    return shards, gathered, views, vis_cap, key_cap


def _merge_both(shards, gathered, views, vis_cap, key_cap):
    import torch
    from vulkan_renderer_amd import _native as N
    lib = N.load()
    world = len(shards)
    res_words = shards[0].exchange_result_words(views)
    out = []
    for static in (False, True):
        vis = torch.full((world * vis_cap + 4,), GUARD, dtype=torch.int32, device="cuda")
        keys = torch.full((world * key_cap + 4,), GUARD, dtype=torch.int64, device="cuda")
        res = torch.full((res_words + 1,), GUARD, dtype=torch.int64, device="cuda")
        torch.cuda.synchronize()
        if static:
            stream = torch.cuda.current_stream().cuda_stream
            rc = lib.vkrhip_exchange_merge_static(torch.cuda.current_device(), views,
                                                  gathered.data_ptr(), world, vis_cap, key_cap,
                                                  vis.data_ptr(), keys.data_ptr(), res.data_ptr(),
                                                  stream)
            assert rc == N.OK, lib.vkrhip_last_error(None)
        else:
            shards[0].exchange_merge(gathered.data_ptr(), world, vis_cap, key_cap, vis.data_ptr(),
                                     keys.data_ptr(), res.data_ptr())
            shards[0].sync()
        torch.cuda.synchronize()
        out.append((vis.cpu().numpy(), keys.cpu().numpy(), res.cpu().numpy()))
    (vis_a, keys_a, res_a), (vis_b, keys_b, res_b) = out
    assert int(res_a[0]) == 0, "the blocks were sized to fit"
    assert res_a[3] > 100 and res_a[4] > 20, "a merge worth comparing"
    assert (res_a == res_b).all(), "result words"
    assert (vis_a == vis_b).all(), "merged visible lists"
    assert (keys_a == keys_b).all(), "merged draw keys"
    assert res_b[-1] == GUARD and (vis_b[world * vis_cap:] == GUARD).all()
    assert (keys_b[world * key_cap:] == GUARD).all()
    return int(res_a[3]), int(res_a[4])


def test_static_merge_matches_twice_the_second_call_smaller():
    big = _gathered_blocks(world=3, cascades=3, n=30000, seed=11)
    small = _gathered_blocks(world=2, cascades=0, n=4000, seed=12)
    try:
        vis_big, keys_big = _merge_both(*big)
        vis_small, keys_small = _merge_both(*small)
        assert vis_small < vis_big and keys_small < keys_big
        _merge_both(*big)  # and back: nothing of the small call lingers either
    finally:
        for hs in big[0] + small[0]:
            hs.close()


def test_static_merge_allocates_nothing_and_is_capturable():
    """What the entry point is for: after vkrhip_static_init the call leaves the
    device's free memory as it found it, and it can be made while its stream is
    being captured into a HIP graph (where an allocation would break the capture);
    the replayed graph writes the same outputs as the eager call."""
    import torch
    from vulkan_renderer_amd import _native as N
    lib = N.load()
    dev = torch.cuda.current_device()
    assert lib.vkrhip_static_init(dev) == N.OK
    shards, gathered, views, vis_cap, key_cap = _gathered_blocks(world=2, cascades=1, n=20000, seed=13)
    try:
        world = len(shards)
        res_words = shards[0].exchange_result_words(views)
        vis = torch.full((world * vis_cap,), GUARD, dtype=torch.int32, device="cuda")
        keys = torch.full((world * key_cap,), GUARD, dtype=torch.int64, device="cuda")
        res = torch.full((res_words,), GUARD, dtype=torch.int64, device="cuda")

        def call(stream):
            rc = lib.vkrhip_exchange_merge_static(dev, views, gathered.data_ptr(), world, vis_cap, key_cap,
                                                  vis.data_ptr(), keys.data_ptr(), res.data_ptr(), stream)
            assert rc == N.OK, lib.vkrhip_last_error(None)

        torch.cuda.synchronize()
        free_before = torch.cuda.mem_get_info(dev)[0]
        call(torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        assert torch.cuda.mem_get_info(dev)[0] == free_before, "device memory was allocated"
        eager = (vis.cpu().numpy().copy(), keys.cpu().numpy().copy(), res.cpu().numpy().copy())
        assert int(eager[2][0]) == 0 and eager[2][3] > 100

        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            call(torch.cuda.current_stream().cuda_stream)  # the capturing stream
        for t in (vis, keys, res):
            t.fill_(GUARD)
        torch.cuda.synchronize()
        assert (vis.cpu().numpy() == np.int32(GUARD)).all()
        graph.replay()
        torch.cuda.synchronize()
        assert (vis.cpu().numpy() == eager[0]).all(), "merged visible lists from the replayed graph"
        assert (keys.cpu().numpy() == eager[1]).all(), "merged draw keys from the replayed graph"
        assert (res.cpu().numpy() == eager[2]).all(), "result words from the replayed graph"
    finally:
        for hs in shards:
            hs.close()


def test_static_merge_limits():
    import torch
    from vulkan_renderer_amd import _native as N
    lib = N.load()
    assert lib.vkrhip_static_init(torch.cuda.current_device()) == N.OK
    buf = torch.zeros(1024, dtype=torch.int64, device="cuda")
    p = buf.data_ptr()
    dev = torch.cuda.current_device()
    rc = lib.vkrhip_exchange_merge_static(dev, 1, p, 9, 4, 4, p, p, p, None)
    assert rc == N.ERR_CAPACITY and b"world size 9" in lib.vkrhip_last_error(None)
    assert lib.vkrhip_exchange_merge_static(dev, 65, p, 1, 4, 4, p, p, p, None) == N.ERR_CAPACITY
    assert lib.vkrhip_exchange_merge_static(dev, 0, p, 1, 4, 4, p, p, p, None) == N.ERR_CAPACITY
    assert lib.vkrhip_exchange_merge_static(dev, 1, None, 1, 4, 4, p, p, p, None) == N.ERR_ARG
    assert lib.vkrhip_exchange_merge_static(4096, 1, p, 1, 4, 4, p, p, p, None) == N.ERR_ARG
