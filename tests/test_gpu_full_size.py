"""The benchmark configurations at their full single-GPU sizes, every output
compared with the CPU oracle bit for bit (the oracle needs about half a minute
for the largest of them), plus the properties that hold at any size."""
import numpy as np
import pytest

from helpers import DIRTY_WORLD, compare_frame, hip_scene_for, oracle_frame, view_set

pytestmark = pytest.mark.gpu

STAGE_ALL, STAGE_DIRTY_LIST = 0x7, 0x100


def _synth():
    from vulkan_renderer_amd import synth
    return synth


def _check_lists(hs, info, n, base=0):
    """Size-independent properties of the per-view lists and the sorted draws
    (`base` = the context's source index base: the lists hold global numbers)."""
    _, cf = hs.read_candidate_prep()
    for v, st in enumerate(info.views):
        vis = hs.read_visible(v, st.visible_count)
        assert st.visible_count == int(st.bucket_counts.sum())
        for b in range(4):
            off, cnt = (int(x) for x in st.execution_ranges[b])
            seg = vis[off:off + cnt].astype(np.int64) - base
            assert (np.diff(seg) > 0).all(), f"view {v} bucket {b} not ascending"
            assert (((cf[seg] >> 4) & 0xF) == b).all(), f"view {v} bucket {b} holds other buckets"
    t = info.transparent_draw_count
    draws, _, keys, src = hs.read_transparent(t)
    assert (keys[:-1] > keys[1:]).all(), "keys strictly descending"
    tv = hs.read_transparent_visible()
    assert t == int(tv.sum()) and (np.sort(src) == np.flatnonzero(tv)).all()
    assert (draws["first_instance"] == np.arange(t)).all()
    assert ((keys & np.uint64(0xFFFFFFFF)) == src.astype(np.uint64) + np.uint64(base + 1)).all()


def test_config2_one_million_flat(oracle):
    """1M instances, flat hierarchy, one view."""
    synth = _synth()
    sc = synth.flat_scene(1 << 20, seed=1)
    vs = synth.camera_views()
    planes, required = view_set(vs)
    ref = oracle_frame(oracle, sc, planes, required, vs.view, vs.projection)
    with hip_scene_for(sc, 1) as hs:
        hs.set_views(planes, required, vs.view)
        hs.frame()
        info = compare_frame(hs, ref, sc)
        assert info.views[0].visible_count > 10000 and info.transparent_draw_count > 1000
        _check_lists(hs, info, sc.n)
        # the steady state of the benchmark: every node marked moved, same result
        hs.mark_all(DIRTY_WORLD)
        hs.frame()
        compare_frame(hs, ref, sc, check_transform=False)
        assert hs.frame_info().nodes_updated == sc.n


def test_config3_sixteen_million_depth8_five_views(oracle):
    """16M instances, depth-8 hierarchy, camera + 4 shadow cascades."""
    synth = _synth()
    sc = synth.deep_scene(1 << 24, depth=8, seed=3)
    assert len(sc.level_offsets) == 9 and sc.n == 1 << 24
    vs = synth.camera_views(cascades=4)
    planes, required = view_set(vs)
    assert planes.shape[0] == 5
    ref = oracle_frame(oracle, sc, planes, required, vs.view, vs.projection)
    with hip_scene_for(sc, 5) as hs:
        hs.set_views(planes, required, vs.view)
        hs.frame()
        info = compare_frame(hs, ref, sc)
        assert info.transparent_draw_count > 8192  # the multi-pass sort
        assert min(st.visible_count for st in info.views) > 10000
        _check_lists(hs, info, sc.n)
        hs.mark_all(DIRTY_WORLD)
        hs.frame()
        compare_frame(hs, ref, sc, check_transform=False)
        # a tenth of the subtrees below level 2 moves (marked on the device), with the
        # render-dirty list: the moved set against the oracle's, frames back to back
        lo = sc.level_offsets
        first, count = int(lo[2]) + 7, int(lo[3] - lo[2]) // 10
        flags = hs.read_node_flags()
        hs.mark_nodes(first, count, DIRTY_WORLD)
        hs.frame(STAGE_ALL | STAGE_DIRTY_LIST)
        flags[first: first + count] |= DIRTY_WORLD
        _, world, flags, dirty = oracle.scene_update(sc.parent, local=ref["local"], flags=flags,
                                                     world=ref["world"])
        got = hs.read_dirty_list()
        assert 0 < dirty.shape[0] < sc.n and got.shape == dirty.shape and (got == dirty).all()
        assert (hs.read_node_flags() == flags).all()
        compare_frame(hs, ref, sc, check_transform=False)  # nothing changed place: same lists and draws


def test_config5_sixteen_million_sixty_four_views(oracle):
    """The 64-view probe configuration at the benchmark's per-GPU size (16M instances;
    VKR_CONFIG5_INSTANCES sets another size). The oracle needs about half a minute."""
    import os
    synth = _synth()
    sc = synth.flat_scene(int(os.environ.get("VKR_CONFIG5_INSTANCES", 1 << 24)), seed=5)
    vps, _ = synth.ring_views(63)
    vs = synth.camera_views()
    planes, required = view_set(vs, extra_vp=vps)
    assert planes.shape[0] == 64
    ref = oracle_frame(oracle, sc, planes, required, vs.view, vs.projection)
    with hip_scene_for(sc, 64) as hs:
        hs.set_views(planes, required, vs.view)
        hs.frame()
        info = compare_frame(hs, ref, sc)
        assert sum(st.visible_count for st in info.views) > 100000
        _check_lists(hs, info, sc.n)


def test_config4_shard_thirty_two_million_flat(oracle):
    """One GPU's shard of the 256M-instance configuration: 32M instances, flat, one
    view, numbered as rank 3 of 8 numbers them."""
    import os
    synth = _synth()
    n = int(os.environ.get("VKR_CONFIG4_INSTANCES", 1 << 25))   # 1 << 27: the largest run of bench.py
    sc = synth.flat_scene(n, seed=1, first=3 * n)
    vs = synth.camera_views()
    planes, required = view_set(vs)
    ref = oracle_frame(oracle, sc, planes, required, vs.view, vs.projection, index_base=3 * n)
    with hip_scene_for(sc, 1, source_index_base=3 * n) as hs:
        hs.set_views(planes, required, vs.view)
        hs.frame()
        info = compare_frame(hs, ref, sc)
        assert info.transparent_draw_count > 8192 and info.views[0].visible_count > 500000
        _check_lists(hs, info, sc.n, base=3 * n)


def _two_level_scene(roots, parent_of_child, seed=11):
    """`roots` roots and one child per entry of parent_of_child (ascending root
    indices): a level large enough for the streaming launches."""
    synth = _synth()
    children = parent_of_child.shape[0]
    n = roots + children
    parent = np.full(n, 0xFFFFFFFF, np.uint32)
    parent[roots:] = parent_of_child.astype(np.uint32)
    offsets = np.array([0, roots, n], np.uint64)
    index = np.arange(n, dtype=np.uint64)
    pos_range = np.where(index < roots, np.float32(400.0), np.float32(40.0)).astype(np.float32)
    pos, quat, scale = synth.make_trs(seed, index, pos_range, (0.9, 1.1))
    ident, matbits, bounds = synth.make_candidates(seed, index)
    return synth.Scene(parent, offsets, pos, quat, scale, ident, matbits, bounds, meta={"kind": "deep", "n": n})


@pytest.mark.parametrize("shape", ["stepping", "childless_parents", "repeated_runs"])
def test_large_level_parent_forms(oracle, shape):
    """A level of 3.4M nodes (too large to join the merged launch of the small
    levels) that begins in the middle of a 64-node chunk and of a tile. 'stepping': every parent is the predecessor's or the next one (read as a
    first parent and step masks); 'childless_parents': some roots have no child,
    so the parents skip (the level keeps its index per node); 'repeated_runs':
    stepping with runs longer than a chunk and single-child parents mixed."""
    roots, children = 300_001, 3_400_000
    j = np.arange(children, dtype=np.uint64)
    if shape == "stepping":
        par = (j * np.uint64(roots)) // np.uint64(children)
    elif shape == "childless_parents":
        par = (j * np.uint64(roots)) // np.uint64(children)
        par = np.where(par % np.uint64(7) == 3, par + np.uint64(1), par)  # every seventh root stays childless
        par = np.minimum(par, np.uint64(roots - 1))
        par = np.maximum.accumulate(par)
        assert (np.diff(par.astype(np.int64)) > 1).any()
    else:
        rng = np.random.default_rng(5)
        run = rng.choice(np.array([1, 1, 2, 9, 150, 200], np.int64), size=roots)
        par = np.repeat(np.arange(roots, dtype=np.uint64), run)[:children]
        assert par.shape[0] == children and (np.diff(par.astype(np.int64)) <= 1).all()
    sc = _two_level_scene(roots, par)
    synth = _synth()
    vs = synth.camera_views(cascades=1)
    planes, required = view_set(vs)
    ref = oracle_frame(oracle, sc, planes, required, vs.view, vs.projection)
    with hip_scene_for(sc, planes.shape[0]) as hs:
        hs.set_views(planes, required, vs.view)
        hs.frame()
        compare_frame(hs, ref, sc)
        hs.mark_all(DIRTY_WORLD)
        hs.frame()
        compare_frame(hs, ref, sc, check_transform=False)
        assert hs.frame_info().nodes_updated == sc.n


@pytest.mark.parametrize("n", [524288, 1000000])
def test_hierarchy_whose_last_level_has_a_low_fan_out(oracle, n):
    """Depth-8 hierarchies whose last level is smaller than (524 288 nodes) or less than
    three times (1M nodes) the level above it: blocks of the merged top-level launch
    would have more ancestors than its table holds, so the plan is rebuilt with fewer
    levels or with shorter blocks."""
    synth = _synth()
    sc = synth.deep_scene(n, depth=8, seed=9)
    sizes = np.diff(sc.level_offsets.astype(np.int64))
    assert sizes[-1] < 3 * sizes[-2]
    vs = synth.camera_views(cascades=2)
    planes, required = view_set(vs)
    ref = oracle_frame(oracle, sc, planes, required, vs.view, vs.projection)
    with hip_scene_for(sc, planes.shape[0]) as hs:
        hs.set_views(planes, required, vs.view)
        hs.frame()
        compare_frame(hs, ref, sc)
        hs.mark_all(DIRTY_WORLD)
        hs.frame()
        compare_frame(hs, ref, sc, check_transform=False)
        assert hs.frame_info().nodes_updated == sc.n
