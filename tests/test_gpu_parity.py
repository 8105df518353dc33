"""HIP path (through the C ABI) against the CPU oracle on the same seeded
inputs. Integer/index outputs and float outputs are compared bit for bit; the
stated tolerance for world matrices (1e-6 relative) is therefore met with zero
slack."""
import numpy as np
import pytest

from conftest import assert_bits_equal
from helpers import (DIRTY_LOCAL, DIRTY_WORLD, WORLD_UPDATED, compare_frame, hip_scene_for,
                     oracle_frame, view_set)

pytestmark = pytest.mark.gpu

STAGE_TRANSFORM, STAGE_CULL, STAGE_TRANSPARENT, STAGE_CANDIDATES, STAGE_BOUNDS = 1, 2, 4, 8, 16


def _synth():
    from vulkan_renderer_amd import synth
    return synth


@pytest.mark.parametrize("n", [1, 63, 64, 255, 256, 257, 1023, 1024, 1025, 5000, 70001])
def test_flat_single_view(oracle, n, staged_mode):
    synth = _synth()
    sc = synth.flat_scene(n, seed=1, blend_fraction=0.3, no_bounds_fraction=0.02)
    vs = synth.camera_views(far=1200.0)
    planes, required = view_set(vs)
    ref = oracle_frame(oracle, sc, planes, required, vs.view, vs.projection)
    with hip_scene_for(sc, 1) as hs:
        hs.set_views(planes, required, vs.view)
        hs.frame()
        compare_frame(hs, ref, sc)


def test_deep_hierarchy_five_views(oracle, staged_mode, tail_mode):
    synth = _synth()
    sc = synth.deep_scene(40000, depth=8, seed=2, fanout=4, roots=5, blend_fraction=0.2,
                          transmissive_fraction=0.05, no_bounds_fraction=0.01)
    vs = synth.camera_views(cascades=4, far=600.0)
    planes, required = view_set(vs)
    assert planes.shape[0] == 5
    ref = oracle_frame(oracle, sc, planes, required, vs.view, vs.projection, tie_eps=1e-5)
    with hip_scene_for(sc, 5, tie_epsilon=1e-5) as hs:
        hs.set_views(planes, required, vs.view)
        hs.frame(STAGE_TRANSFORM | STAGE_CULL | STAGE_TRANSPARENT | STAGE_CANDIDATES)
        info = compare_frame(hs, ref, sc)
        assert sum(v.visible_count for v in info.views) > 1000
        # both candidate streams (application.h:641-645)
        assert hs.read_candidates().tobytes() == ref["payload"]["gpu_candidates"].tobytes()
        expect = ref["payload"]["transmission_candidates"]
        assert 1000 < expect.shape[0] == info.transmission_gpu_candidate_count
        got = hs.read_transmission_candidates(info.transmission_gpu_candidate_count)
        assert got.tobytes() == expect.tobytes(), "transmission_gpu_candidates rows"


def test_sixty_four_views(oracle):
    synth = _synth()
    sc = synth.flat_scene(20000, seed=7, blend_fraction=0.1)
    vps, (view, proj) = synth.ring_views(63)
    vs = synth.camera_views(far=800.0)
    planes, required = view_set(vs, extra_vp=vps)
    assert planes.shape[0] == 64
    ref = oracle_frame(oracle, sc, planes, required, vs.view, vs.projection)
    with hip_scene_for(sc, 64) as hs:
        hs.set_views(planes, required, vs.view)
        hs.frame()
        compare_frame(hs, ref, sc)


def test_partition_capacity_overflow(oracle, staged_mode):
    synth = _synth()
    sc = synth.flat_scene(30000, seed=3, blend_fraction=0.05)
    vs = synth.camera_views(far=3000.0, fov_deg=100.0)
    planes, required = view_set(vs)
    ref = oracle_frame(oracle, sc, planes, required, vs.view, vs.projection, partition_capacity=500)
    assert int(ref["states"][0]["overflow_count"]) > 0
    with hip_scene_for(sc, 1, partition_capacity=500) as hs:
        hs.set_views(planes, required, vs.view)
        hs.frame()
        compare_frame(hs, ref, sc)


def test_node_mapping_and_submeshes(oracle):
    """Several candidate rows (submeshes) per transform, in arbitrary node order."""
    synth = _synth()
    sc = synth.deep_scene(3000, depth=5, seed=4, fanout=6, roots=3)
    rng = np.random.default_rng(5)
    n_cand = 8000
    node = rng.integers(0, sc.n, n_cand).astype(np.uint32)
    ident, matbits, bounds = synth.make_candidates(9, np.arange(n_cand, dtype=np.uint64),
                                                   blend_fraction=0.25)
    sc.node, sc.ident, sc.matbits, sc.bounds = node, ident, matbits, bounds
    vs = synth.camera_views(cascades=2, far=900.0)
    planes, required = view_set(vs)
    ref = oracle_frame(oracle, sc, planes, required, vs.view, vs.projection)
    with hip_scene_for(sc, 3) as hs:
        hs.set_views(planes, required, vs.view)
        hs.frame(STAGE_TRANSFORM | STAGE_CULL | STAGE_TRANSPARENT | STAGE_CANDIDATES | STAGE_BOUNDS)
        compare_frame(hs, ref, sc)
        rows = hs.read_candidates()
        assert rows.tobytes() == ref["payload"]["gpu_candidates"].tobytes(), "VkrWorldDrawCandidate rows"
        wb = hs.read_world_bounds()
        exp = np.stack([oracle.instance_world_bounds(ref["world"][node[i]], bounds[i, :3],
                                                     ref["payload"]["gpu_candidates"][i]["sphere"][3])
                        for i in range(0, n_cand, 7)])
        assert_bits_equal(wb[::7], exp, "instance world bounds")


def test_source_index_base(oracle, staged_mode):
    synth = _synth()
    sc = synth.flat_scene(9000, seed=6, first=123456, blend_fraction=0.4)
    vs = synth.camera_views(far=1500.0)
    planes, required = view_set(vs)
    ref = oracle_frame(oracle, sc, planes, required, vs.view, vs.projection, index_base=123456)
    with hip_scene_for(sc, 1, source_index_base=123456) as hs:
        hs.set_views(planes, required, vs.view)
        hs.frame()
        compare_frame(hs, ref, sc)


def test_dirty_propagation(oracle, staged_mode, tail_mode):
    """Only moved nodes and their descendants are recomputed
    (vkr_scene_system.c:1310-1321)."""
    synth = _synth()
    sc = synth.deep_scene(6000, depth=6, seed=8, fanout=5, roots=4)
    n = sc.n
    with hip_scene_for(sc, 1) as hs:
        hs.frame(STAGE_TRANSFORM)
        info = hs.frame_info()
        assert info.nodes_updated == n
        local0, world0, flags0, _ = oracle.scene_update(
            sc.parent, pos=sc.pos, quat=sc.quat, scale=sc.scale, flags=np.full(n, DIRTY_LOCAL, np.uint8))
        assert_bits_equal(hs.read_world(), world0, "first frame")
        # second frame with nothing dirty: nothing moves, WORLD_UPDATED clears
        hs.frame(STAGE_TRANSFORM)
        assert hs.frame_info().nodes_updated == 0
        assert (hs.read_node_flags() == 0).all()
        # move one level-1 node: its subtree follows
        lvl1 = int(sc.level_offsets[1]) + 1
        pos = sc.pos.copy()
        pos[lvl1, :3] += np.float32(3.5)
        hs.upload_trs(pos[lvl1:lvl1 + 1], sc.quat[lvl1:lvl1 + 1], sc.scale[lvl1:lvl1 + 1], first=lvl1)
        hs.frame(STAGE_TRANSFORM)
        flags1 = np.zeros(n, np.uint8)
        flags1[lvl1] = DIRTY_LOCAL
        local1, world1, flags_out, dirty = oracle.scene_update(
            sc.parent, local=local0, world=world0, flags=flags1, pos=pos, quat=sc.quat, scale=sc.scale)
        assert 1 < dirty.shape[0] < n
        assert hs.frame_info().nodes_updated == dirty.shape[0]
        assert_bits_equal(hs.read_world(), world1, "after partial update")
        assert_bits_equal(hs.read_locals(), local1, "locals after partial update")
        assert (hs.read_node_flags() == flags_out).all()
        # bulk "everything moved" marker
        hs.mark_all(DIRTY_WORLD)
        hs.frame(STAGE_TRANSFORM)
        assert hs.frame_info().nodes_updated == n
        assert_bits_equal(hs.read_world(), world1, "after mark_all")


def test_mark_nodes_moves_whole_subtrees(oracle):
    """vkrhip_mark_nodes: DIRTY_WORLD set on the device for a run of nodes of one
    level; their descendants follow by deferred propagation, everything else stays
    clean (flags, world matrices and the length of the render-dirty list against
    the oracle with the same flags set on the host). Two frames in a row, the second
    with another run, without a host wait in between."""
    synth = _synth()
    sc = synth.deep_scene(60000, depth=6, seed=31, fanout=5, roots=6)
    vs = synth.camera_views(cascades=1)
    planes, required = view_set(vs)
    with hip_scene_for(sc, planes.shape[0]) as hs:
        hs.set_views(planes, required, vs.view)
        hs.frame()
        ref = oracle_frame(oracle, sc, planes, required, vs.view, vs.projection)
        compare_frame(hs, ref, sc)
        local, world, flags = ref["local"], ref["world"], ref["flags"]
        lo = sc.level_offsets
        for first, count in ((int(lo[2]) + 3, 17), (int(lo[1]) + 1, 2)):
            hs.mark_nodes(first, count, DIRTY_WORLD)
            hs.frame()
            flags = flags.copy()
            flags[first: first + count] |= DIRTY_WORLD
            _, world, flags, dirty = oracle.scene_update(sc.parent, local=local, flags=flags, world=world)
        info = hs.frame_info()
        assert info.nodes_updated == dirty.shape[0] and 0 < dirty.shape[0] < sc.n
        assert_bits_equal(hs.read_world(), world, "world matrices")
        assert (hs.read_node_flags() == flags).all()
        assert hs.lib.vkrhip_mark_nodes(hs.ctx, sc.n, 1 << 40, DIRTY_WORLD) != 0  # beyond the capacity


def test_uploaded_locals_and_unordered_hierarchy(oracle):
    """Arbitrary parent array -> levelize -> propagate, against the oracle's BFS."""
    from vulkan_renderer_amd import HipScene, levelize
    rng = np.random.default_rng(21)
    n = 5000
    parent = np.full(n, 0xFFFFFFFF, np.uint32)
    for i in range(1, n):
        if rng.uniform() > 0.01:
            parent[i] = rng.integers(0, i)
    perm = rng.permutation(n)          # scramble storage order
    inv = np.argsort(perm)
    parent_s = np.full(n, 0xFFFFFFFF, np.uint32)
    for new, old in enumerate(perm):
        p = parent[old]
        parent_s[new] = 0xFFFFFFFF if p == 0xFFFFFFFF else inv[p]
    local_s = rng.uniform(-1.5, 1.5, (n, 16)).astype(np.float32)
    topo, _, levels = oracle.build_topo_order(parent_s)
    _, world_s, _, _ = oracle.scene_update(parent_s, local=local_s, flags=np.full(n, DIRTY_WORLD, np.uint8),
                                           topo=topo)
    order, new_parent, offsets = levelize(parent_s)
    assert (order == topo).all() and offsets.shape[0] - 1 == levels
    with HipScene(max_nodes=n, max_candidates=1) as hs:
        hs.set_hierarchy(new_parent, offsets)
        hs.upload_locals(local_s[order])
        hs.frame(STAGE_TRANSFORM)
        assert_bits_equal(hs.read_world(), world_s[order], "levelized world matrices")


def test_edge_values(oracle, staged_mode):
    """NaN / zero-scale / behind-camera / equal-depth rows."""
    synth = _synth()
    n = 4096
    sc = synth.flat_scene(n, seed=10, blend_fraction=1.0)
    sc.pos[:, 2] = -np.abs(sc.pos[:, 2])       # in front of the camera
    sc.pos[:1500, :3] = sc.pos[0, :3]           # many identical depths (ties broken by index)
    sc.scale[100:140, :3] = 0.0                 # zero scale: radius 0
    sc.pos[200:210, 0] = np.nan                 # NaN centre: every compare false => visible
    sc.pos[300:320, 2] = 50.0                   # behind the camera
    sc.scale[300:320, :3] = 400.0               # ... but huge: visible with depth clamped to 0
    sc.bounds[400:420, 8:11] = np.inf           # infinite extent
    sc.matbits[500:520] &= ~np.uint32(0x10)     # bounds invalid: always visible
    vs = synth.camera_views(far=2000.0)
    planes, required = view_set(vs)
    ref = oracle_frame(oracle, sc, planes, required, vs.view, vs.projection)
    keys = ref["payload"]["sort_keys"]
    assert ((keys >> np.uint64(32)) == 0).sum() >= 10      # clamped depths present
    with hip_scene_for(sc, 1) as hs:
        hs.set_views(planes, required, vs.view)
        hs.frame()
        compare_frame(hs, ref, sc)


def test_empty_scene_candidates(oracle):
    from vulkan_renderer_amd import HipScene
    synth = _synth()
    vs = synth.camera_views()
    planes, required = view_set(vs)
    sc = synth.flat_scene(10, seed=1)
    with HipScene(max_nodes=10, max_candidates=4) as hs:
        hs.set_hierarchy(sc.parent, sc.level_offsets)
        hs.upload_trs(sc.pos, sc.quat, sc.scale)
        hs.set_candidates(None, np.zeros((0, 8), np.uint32), np.zeros(0, np.uint32),
                          np.zeros((0, 12), np.float32))
        hs.set_views(planes, required, vs.view)
        hs.frame()
        info = hs.frame_info()
        assert info.objects_tested == 0 and info.transparent_draw_count == 0
        assert info.views[0].visible_count == 0 and info.nodes_updated == 10


def test_error_conventions():
    from vulkan_renderer_amd import HipScene
    from vulkan_renderer_amd._native import ERR_ARG, ERR_CAPACITY, ERR_STATE, VkrHipError
    synth = _synth()
    with HipScene(max_nodes=8, max_candidates=8) as hs:
        with pytest.raises(VkrHipError) as e:
            hs.frame()
        assert e.value.code == ERR_STATE
        sc = synth.flat_scene(9)
        with pytest.raises(VkrHipError) as e:
            hs.set_hierarchy(sc.parent, sc.level_offsets)
        assert e.value.code == ERR_CAPACITY
        bad_parent = np.array([0xFFFFFFFF, 5, 0], np.uint32)
        with pytest.raises(VkrHipError) as e:
            hs.set_hierarchy(bad_parent, np.array([0, 1, 3], np.uint64))
        assert e.value.code == ERR_ARG


def test_large_properties():
    """Full-size run (1M, the bench workload): size-independent properties."""
    synth = _synth()
    n = 1 << 20
    sc = synth.flat_scene(n, seed=1)
    vs = synth.camera_views()
    planes, required = view_set(vs)
    with hip_scene_for(sc, 1) as hs:
        hs.set_views(planes, required, vs.view)
        hs.frame()
        info = hs.frame_info()
        st = info.views[0]
        vis = hs.read_visible(0, st.visible_count)
        # bucket-major, strictly ascending inside each bucket, bucket matches material
        sphere, cf = hs.read_candidate_prep()
        for b in range(4):
            off, cnt = st.execution_ranges[b]
            seg = vis[off:off + cnt]
            assert (np.diff(seg.astype(np.int64)) > 0).all()
            assert (((cf[seg] >> 4) & 0xF) == b).all()
        assert st.visible_count == st.bucket_counts.sum()
        t = info.transparent_draw_count
        draws, inst, keys, src = hs.read_transparent(t)
        assert (np.diff(keys.astype(np.float64)) <= 0).all() and np.unique(keys).shape[0] == t
        assert (keys[:-1] > keys[1:]).all()
        tv = hs.read_transparent_visible()
        assert t == int(tv.sum()) and (np.sort(src) == np.flatnonzero(tv)).all()
        assert (draws["first_instance"] == np.arange(t)).all()
        assert ((keys & np.uint64(0xFFFFFFFF)) == src.astype(np.uint64) + np.uint64(1)).all()
        # idempotence: same frame again gives the same lists
        hs.frame()
        info2 = hs.frame_info()
        assert info2.views[0].visible_count == st.visible_count
        assert (hs.read_visible(0, st.visible_count) == vis).all()
        assert (hs.read_transparent(t)[2] == keys).all()


def test_large_transparent_sort(oracle):
    """More visible blend rows than the single-workgroup sort holds: the
    multi-pass histogram/scan/scatter path."""
    synth = _synth()
    sc = synth.flat_scene(150000, seed=12, blend_fraction=1.0)
    sc.pos[::3, :3] = sc.pos[0, :3]            # a third of the rows share one depth
    vs = synth.camera_views(far=4000.0, fov_deg=110.0)
    planes, required = view_set(vs)
    ref = oracle_frame(oracle, sc, planes, required, vs.view, vs.projection)
    assert ref["payload"]["draws"].shape[0] > 3 * 8192
    with hip_scene_for(sc, 1) as hs:
        hs.set_views(planes, required, vs.view)
        hs.frame()
        compare_frame(hs, ref, sc)


def test_rank_sort_above_eight_thousand(oracle):
    """Between 8 192 and about 16 000 visible blend rows the rows are still ranked by
    counting (one launch) once the host has seen the count: the first frame sorts them
    with the multi-pass kernels, the second by ranking, both like the oracle."""
    synth = _synth()
    sc = synth.flat_scene(180000, seed=21, blend_fraction=1.0)
    sc.pos[::5, :3] = sc.pos[1, :3]            # a fifth of the rows share one depth
    vs = synth.camera_views(far=4000.0, fov_deg=70.0)
    planes, required = view_set(vs)
    ref = oracle_frame(oracle, sc, planes, required, vs.view, vs.projection)
    assert 9000 < ref["payload"]["draws"].shape[0] < 15000
    with hip_scene_for(sc, 1) as hs:
        hs.set_views(planes, required, vs.view)
        hs.frame()
        compare_frame(hs, ref, sc)
        hs.mark_all(DIRTY_WORLD)
        hs.frame()
        compare_frame(hs, ref, sc, check_transform=False)


def test_separate_stages(oracle):
    """Transform in one frame call, cull + sort in another: the stand-alone
    classify kernel with the identity candidate->node mapping."""
    synth = _synth()
    sc = synth.deep_scene(30000, depth=6, seed=14, fanout=5, roots=6, blend_fraction=0.3)
    vs = synth.camera_views(cascades=3, far=900.0)
    planes, required = view_set(vs)
    ref = oracle_frame(oracle, sc, planes, required, vs.view, vs.projection)
    with hip_scene_for(sc, 4) as hs:
        hs.set_views(planes, required, vs.view)
        hs.frame(STAGE_TRANSFORM)
        assert hs.frame_info().nodes_updated == sc.n
        hs.frame(STAGE_CULL | STAGE_TRANSPARENT)
        compare_frame(hs, ref, sc, check_transform=False)
        assert_bits_equal(hs.read_world(), ref["world"], "world matrices")


def test_sort_path_misprediction(oracle):
    """The host picks the sort path from an earlier frame's transparent count.
    A frame with few visible blend rows followed by one with many makes the
    single-workgroup sort handle a count beyond its LDS capacity (streaming
    through global memory); the frame after that uses the multi-pass path."""
    synth = _synth()
    sc = synth.flat_scene(120000, seed=15, blend_fraction=1.0)
    narrow = synth.camera_views(far=30.0, fov_deg=20.0)
    wide = synth.camera_views(far=4000.0, fov_deg=110.0)
    p_n, r_n = view_set(narrow)
    p_w, r_w = view_set(wide)
    ref_n = oracle_frame(oracle, sc, p_n, r_n, narrow.view, narrow.projection)
    ref_w = oracle_frame(oracle, sc, p_w, r_w, wide.view, wide.projection)
    assert ref_n["payload"]["draws"].shape[0] < 2000
    assert ref_w["payload"]["draws"].shape[0] > 20000
    with hip_scene_for(sc, 1) as hs:
        hs.set_views(p_n, r_n, narrow.view)
        hs.frame()
        compare_frame(hs, ref_n, sc)                 # count small, small path
        hs.set_views(p_w, r_w, wide.view)
        hs.mark_all(DIRTY_WORLD)
        hs.frame()                                   # predicted small, actually large
        compare_frame(hs, ref_w, sc, check_transform=False)
        hs.frame()                                   # predicted large: multi-pass kernels
        compare_frame(hs, ref_w, sc, check_transform=False)
        hs.set_views(p_n, r_n, narrow.view)
        hs.frame()                                   # predicted large, actually small
        compare_frame(hs, ref_n, sc, check_transform=False)


def test_rows_must_name_nodes_of_the_scene(oracle):
    """A row that names a node beyond the bound hierarchy is refused when it is
    bound, and a hierarchy that shrinks under bound rows fails the frame."""
    from vulkan_renderer_amd import _native as N
    synth = _synth()
    sc = synth.deep_scene(600, depth=4, seed=4, fanout=5, roots=3)
    vs = synth.camera_views()
    planes, required = view_set(vs)
    n_cand = 900
    ident, matbits, bounds = synth.make_candidates(3, np.arange(n_cand, dtype=np.uint64))
    node = (np.arange(n_cand) % sc.n).astype(np.uint32)
    sc.node, sc.ident, sc.matbits, sc.bounds = node, ident, matbits, bounds
    from vulkan_renderer_amd import HipScene
    with HipScene(max_nodes=2000, max_candidates=n_cand, max_views=1) as hs:
        hs.set_hierarchy(sc.parent, sc.level_offsets)
        hs.upload_trs(sc.pos, sc.quat, sc.scale)
        bad = node.copy()
        bad[17] = sc.n  # inside max_nodes, outside the hierarchy
        with pytest.raises(N.VkrHipError) as e:
            hs.set_candidates(bad, ident, matbits, bounds)
        assert e.value.code == N.ERR_ARG
        hs.set_candidates(node, ident, matbits, bounds)
        hs.set_views(planes, required, vs.view)
        hs.frame()
        ref = oracle_frame(oracle, sc, planes, required, vs.view, vs.projection)
        compare_frame(hs, ref, sc)
        # the scene shrinks to its first two levels: the rows still name deeper nodes
        small = int(sc.level_offsets[2])
        hs.set_hierarchy(sc.parent[:small], sc.level_offsets[:3])
        with pytest.raises(N.VkrHipError) as e:
            hs.frame()
        assert e.value.code == N.ERR_STATE
        hs.set_candidates(node % small, ident, matbits, bounds)
        hs.frame()


def test_render_dirty_list(oracle):
    """STAGE_DIRTY_LIST: the list scene_mark_render_dirty builds while vkr_scene_update
    walks the topological order - the updated nodes, ascending - against the
    oracle's: all nodes on the first frame, moved subtrees afterwards, nothing when
    nothing moved; with the fused kernel and with the transform alone."""
    from vulkan_renderer_amd import _native as N
    synth = _synth()
    sc = synth.deep_scene(70000, depth=6, seed=17, fanout=6, roots=7)
    vs = synth.camera_views(cascades=1)
    planes, required = view_set(vs)
    rng = np.random.default_rng(5)
    with hip_scene_for(sc, planes.shape[0]) as hs:
        hs.set_views(planes, required, vs.view)
        hs.frame(N.STAGE_ALL | N.STAGE_DIRTY_LIST)
        local, world, flags, dirty = oracle.scene_update(
            sc.parent, pos=sc.pos, quat=sc.quat, scale=sc.scale, flags=np.full(sc.n, DIRTY_LOCAL, np.uint8))
        assert (hs.read_dirty_list() == dirty).all() and dirty.shape[0] == sc.n
        for frame, stages in enumerate((N.STAGE_ALL, N.STAGE_TRANSFORM, N.STAGE_ALL)):
            marks = np.zeros(sc.n, np.uint8)
            if frame < 2:
                marks[rng.integers(0, sc.n, 40)] = DIRTY_WORLD
                marks[int(sc.level_offsets[2]) + 5] = DIRTY_WORLD
            hs.upload_node_flags(flags | marks)
            hs.frame(stages | N.STAGE_DIRTY_LIST)
            _, world, flags, dirty = oracle.scene_update(sc.parent, local=local, flags=flags | marks, world=world)
            got = hs.read_dirty_list()
            assert got.shape == dirty.shape and (got == dirty).all(), f"frame {frame}"
            assert hs.frame_info().nodes_updated == dirty.shape[0]
            assert (frame < 2) == (dirty.shape[0] > 0)
        hs.frame(N.STAGE_ALL)
        with pytest.raises(N.VkrHipError) as e:
            hs.read_dirty_list()
        assert e.value.code == N.ERR_STATE


def test_hierarchy_shrinks_and_grows_back(oracle):
    """The flags of nodes outside the bound hierarchy read the same whichever of the
    two flag copies is current: after the scene shrinks and an odd number of frames
    has run, the nodes beyond it still show what the last frame that covered them
    left (WORLD_UPDATED), and when the scene grows back they are not recomputed
    from stale dirty bits - matrices uploaded for them meanwhile stay."""
    synth = _synth()
    sc = synth.deep_scene(3000, depth=4, seed=12, fanout=6, roots=5)
    n, small = sc.n, int(sc.level_offsets[2])
    with hip_scene_for(sc, 1) as hs:
        hs.frame(STAGE_TRANSFORM)
        assert (hs.read_node_flags() == WORLD_UPDATED).all()
        hs.set_hierarchy(sc.parent[:small], sc.level_offsets[:3])
        hs.frame(STAGE_TRANSFORM)  # one frame: the other flag copy is current now
        flags = hs.read_node_flags(0, n)
        assert (flags[:small] == 0).all() and (flags[small:n] == WORLD_UPDATED).all()
        placed = hs.read_world(small, n - small)
        placed[:, 12:15] += np.float32(7.0)
        hs.upload_worlds(placed, first=small)
        hs.set_hierarchy(sc.parent, sc.level_offsets)
        hs.frame(STAGE_TRANSFORM)
        assert hs.frame_info().nodes_updated == 0
        assert (hs.read_node_flags() == 0).all()
        assert_bits_equal(hs.read_world()[small:], placed, "matrices uploaded while the scene was small")


@pytest.mark.parametrize("top_levels", ["1", "0"])
@pytest.mark.parametrize("dirty", ["all", "subset"])
def test_merged_top_levels_on_and_off(oracle, monkeypatch, top_levels, dirty):
    """The small top levels of a hierarchy in one launch (every block recomputes
    its nodes' ancestors) and as a launch per level (VKRHIP_TOP_LEVELS=0): both
    equal the oracle over three frames, with every node moved and with only a
    few subtrees moved (deferred dirty propagation through recomputed ancestors,
    the two copies of the node flags swapping every frame)."""
    synth = _synth()
    monkeypatch.setenv("VKRHIP_TOP_LEVELS", top_levels)
    sc = synth.deep_scene(90000, depth=8, seed=13, fanout=5, roots=7, blend_fraction=0.2)
    vs = synth.camera_views(cascades=2, far=700.0)
    planes, required = view_set(vs)
    rng = np.random.default_rng(3)
    with hip_scene_for(sc, 3) as hs:
        hs.set_views(planes, required, vs.view)
        hs.frame()
        compare_frame(hs, oracle_frame(oracle, sc, planes, required, vs.view, vs.projection), sc)
        world = hs.read_world()
        local = hs.read_locals()
        for frame in range(3):
            if dirty == "all":
                flags = np.full(sc.n, DIRTY_WORLD, np.uint8)
            else:  # a handful of nodes near the roots and a few leaves
                flags = np.zeros(sc.n, np.uint8)
                flags[rng.integers(0, 200, 5)] = DIRTY_WORLD
                flags[rng.integers(0, sc.n, 40)] = DIRTY_WORLD
            # moved nodes get a new local matrix (a small translation)
            moved = np.flatnonzero(flags)
            local = local.copy()
            local[moved, 12] += np.float32(0.25 * (frame + 1))
            hs.upload_locals(local)
            hs.upload_node_flags(flags)
            _, want_world, want_flags, dirty_list = oracle.scene_update(sc.parent, local=local, flags=flags,
                                                                         world=world)
            hs.frame()
            ref = oracle_frame(oracle, sc, planes, required, vs.view, vs.projection, world=want_world)
            compare_frame(hs, ref, sc, check_transform=False)
            assert_bits_equal(hs.read_world(), want_world, f"world matrices, frame {frame}")
            assert (hs.read_node_flags() == want_flags).all(), f"node flags, frame {frame}"
            assert hs.frame_info().nodes_updated == dirty_list.shape[0] > 0
            world = want_world


@pytest.mark.parametrize("balance", ["1", "0"])
def test_merged_top_levels_balanced(oracle, monkeypatch, balance):
    """The merged launch sized for one resident set of blocks: the merged levels
    here make more 256-node blocks than a 256-CU device holds at once, so the
    widest of them gets longer blocks (two rounds of a tile each). Three frames,
    every node moved and then a few subtrees, then a transform-only frame;
    VKRHIP_TOP_BALANCE=0 is the plan with 256-node blocks throughout."""
    synth = _synth()
    monkeypatch.setenv("VKRHIP_TOP_BALANCE", balance)
    sc = synth.deep_scene(1500000, depth=4, seed=18, fanout=200, roots=8, blend_fraction=0.05)
    vs = synth.camera_views(cascades=2, far=700.0)
    planes, required = view_set(vs)
    rng = np.random.default_rng(4)
    with hip_scene_for(sc, 3) as hs:
        hs.set_views(planes, required, vs.view)
        hs.frame()
        compare_frame(hs, oracle_frame(oracle, sc, planes, required, vs.view, vs.projection), sc)
        world = hs.read_world()
        local = hs.read_locals()
        offsets = np.asarray(sc.level_offsets)
        for frame in range(3):
            if frame == 0:
                flags = np.full(sc.n, DIRTY_WORLD, np.uint8)
            else:  # a few nodes of every level, the end of the third level among them
                flags = np.zeros(sc.n, np.uint8)
                flags[rng.integers(0, int(offsets[2]), 4)] = DIRTY_WORLD
                flags[rng.integers(int(offsets[3]) - 1500, int(offsets[3]), 30)] = DIRTY_WORLD
                flags[rng.integers(0, sc.n, 60)] = DIRTY_WORLD
            moved = np.flatnonzero(flags)
            local = local.copy()
            local[moved, 13] += np.float32(0.5 * (frame + 1))
            hs.upload_locals(local)
            hs.upload_node_flags(flags)
            _, want_world, want_flags, dirty_list = oracle.scene_update(sc.parent, local=local, flags=flags,
                                                                         world=world)
            hs.frame()
            ref = oracle_frame(oracle, sc, planes, required, vs.view, vs.projection, world=want_world)
            compare_frame(hs, ref, sc, check_transform=False)
            assert_bits_equal(hs.read_world(), want_world, f"world matrices, frame {frame}")
            assert (hs.read_node_flags() == want_flags).all(), f"node flags, frame {frame}"
            assert hs.frame_info().nodes_updated == dirty_list.shape[0] > 0
            world = want_world
        # transform only: the same plan without the cull
        flags = np.full(sc.n, DIRTY_WORLD, np.uint8)
        local = local.copy()
        local[:, 12] -= np.float32(0.75)
        hs.upload_locals(local)
        hs.upload_node_flags(flags)
        _, want_world, _, _ = oracle.scene_update(sc.parent, local=local, flags=flags, world=world)
        hs.frame(STAGE_TRANSFORM)
        assert_bits_equal(hs.read_world(), want_world, "world matrices, transform only")


@pytest.mark.parametrize("streaming", ["1", "0"])
def test_streaming_accesses_forced(oracle, monkeypatch, streaming):
    """Frames larger than the last-level cache run the level launches and the
    candidate rows with streaming loads and stores (other instantiations of the
    kernels). VKRHIP_STREAMING=1 forces them on a small scene: a first frame that
    rebuilds every local matrix from its TRS, both candidate streams, a partial
    TRS change, a frame with nothing moved, a few subtrees moved through their
    local matrices, and a transform-only frame - every output against the oracle."""
    synth = _synth()
    monkeypatch.setenv("VKRHIP_STREAMING", streaming)
    sc = synth.deep_scene(150000, depth=6, seed=31, fanout=6, roots=5, blend_fraction=0.2,
                          transmissive_fraction=0.05, no_bounds_fraction=0.01)
    n = sc.n
    vs = synth.camera_views(cascades=2, far=700.0)
    planes, required = view_set(vs)
    rng = np.random.default_rng(9)
    stages = STAGE_TRANSFORM | STAGE_CULL | STAGE_TRANSPARENT | STAGE_CANDIDATES
    with hip_scene_for(sc, 3) as hs:
        hs.set_views(planes, required, vs.view)
        hs.frame(stages)
        ref = oracle_frame(oracle, sc, planes, required, vs.view, vs.projection)
        info = compare_frame(hs, ref, sc)
        assert hs.read_candidates().tobytes() == ref["payload"]["gpu_candidates"].tobytes()
        expect = ref["payload"]["transmission_candidates"]
        got = hs.read_transmission_candidates(info.transmission_gpu_candidate_count)
        assert got.tobytes() == expect.tobytes(), "transmission_gpu_candidates rows"
        local, world = ref["local"], ref["world"]
        # some nodes get a new position (pass 1 for them, pass 2 for their subtrees)
        moved = np.unique(np.concatenate([rng.integers(0, 40, 3), rng.integers(0, n, 200)]))
        pos = sc.pos.copy()
        pos[moved, :3] += np.float32(1.25)
        for i in moved:
            hs.upload_trs(pos[i:i + 1], sc.quat[i:i + 1], sc.scale[i:i + 1], first=int(i))
        flags = np.zeros(n, np.uint8)
        flags[moved] = DIRTY_LOCAL
        local, world, flags_out, dirty = oracle.scene_update(sc.parent, local=local, world=world, flags=flags,
                                                             pos=pos, quat=sc.quat, scale=sc.scale)
        hs.frame(stages)
        assert hs.frame_info().nodes_updated == dirty.shape[0]
        assert_bits_equal(hs.read_locals(), local, "locals after the partial change")
        assert (hs.read_node_flags() == flags_out).all()
        ref = oracle_frame(oracle, sc, planes, required, vs.view, vs.projection, world=world)
        compare_frame(hs, ref, sc, check_transform=False)
        assert_bits_equal(hs.read_world(), world, "world matrices after the partial change")
        assert hs.read_candidates().tobytes() == ref["payload"]["gpu_candidates"].tobytes()
        # nothing moved
        hs.frame(stages)
        assert hs.frame_info().nodes_updated == 0
        compare_frame(hs, ref, sc, check_transform=False)
        assert_bits_equal(hs.read_world(), world, "world matrices, nothing moved")
        # subtrees moved through their local matrices
        flags = np.zeros(n, np.uint8)
        flags[rng.integers(0, 200, 5)] = DIRTY_WORLD
        flags[rng.integers(0, n, 50)] = DIRTY_WORLD
        moved = np.flatnonzero(flags)
        local = local.copy()
        local[moved, 12] -= np.float32(0.75)
        hs.upload_locals(local)
        hs.upload_node_flags(flags)
        _, world, flags_out, dirty = oracle.scene_update(sc.parent, local=local, flags=flags, world=world)
        hs.frame(stages)
        ref = oracle_frame(oracle, sc, planes, required, vs.view, vs.projection, world=world)
        compare_frame(hs, ref, sc, check_transform=False)
        assert_bits_equal(hs.read_world(), world, "world matrices, subtrees moved")
        assert (hs.read_node_flags() == flags_out).all()
        assert hs.frame_info().nodes_updated == dirty.shape[0] > 0
        # transform only
        hs.mark_all(DIRTY_WORLD)
        hs.frame(STAGE_TRANSFORM)
        assert hs.frame_info().nodes_updated == n
        assert_bits_equal(hs.read_world(), world, "world matrices, transform only")


def test_page_locked_host_buffers(oracle):
    """vkrhip_host_alloc: the read copies land in page-locked memory of the
    library exactly as they land in ordinary memory; zero bytes and a null
    result pointer are argument errors, freeing null is not."""
    import ctypes as C
    from vulkan_renderer_amd import _native as N
    synth = _synth()
    sc = synth.deep_scene(30000, depth=5, seed=12, fanout=6, roots=4, blend_fraction=0.3)
    vs = synth.camera_views(cascades=1, far=600.0)
    planes, required = view_set(vs)
    with hip_scene_for(sc, 2) as hs:
        hs.set_views(planes, required, vs.view)
        hs.frame(STAGE_TRANSFORM | STAGE_CULL | STAGE_TRANSPARENT | STAGE_CANDIDATES)
        info = hs.frame_info()
        t = info.transparent_draw_count
        assert t > 100
        rows = hs.host_buffer(sc.n * 144)
        draws, inst = hs.host_buffer(t * 48), hs.host_buffer(t * 80)
        hs._ok(hs.lib.vkrhip_read_candidates(hs.ctx, 0, sc.n, N.ptr(rows)))
        hs._ok(hs.lib.vkrhip_read_transparent(hs.ctx, t, N.ptr(draws), N.ptr(inst), None, None))
        assert rows.tobytes() == hs.read_candidates().tobytes()
        want_draws, want_inst, _, _ = hs.read_transparent(t)
        assert draws.tobytes() == want_draws.tobytes() and inst.tobytes() == want_inst.tobytes()
        ref = oracle_frame(oracle, sc, planes, required, vs.view, vs.projection)
        assert rows.tobytes() == ref["payload"]["gpu_candidates"].tobytes()
        del rows, draws, inst
        hs.host_buffers_free()
        p = C.c_void_p()
        assert hs.lib.vkrhip_host_alloc(hs.ctx, 0, C.byref(p)) == N.ERR_ARG and not p.value
        assert hs.lib.vkrhip_host_alloc(hs.ctx, 64, None) == N.ERR_ARG
        assert hs.lib.vkrhip_host_free(hs.ctx, None) == 0


@pytest.mark.parametrize("overlap", ["0", "2", "tail"])
@pytest.mark.parametrize("blend_fraction", [0.02, 0.6])
def test_transparent_sort_on_second_stream(oracle, monkeypatch, overlap, blend_fraction):
    """VKRHIP_OVERLAP_SORT=2: the transparent sort and emission of every frame run on
    a second stream beside the next frame's transform (by default only large
    hierarchies do that); =0 keeps them on the frame's stream; "tail"
    (VKRHIP_OVERLAP_TAIL=2): scan and gather go there as well, the next frame starts
    right behind the classify and rewrites the world matrices while the previous
    frame's lists and draws are still being made. Frames are enqueued
    in pairs with different dirty sets and nothing read in between, so the second
    frame's transform overwrites world matrices while the first frame's rows may
    still be in flight; then a caller's kernel-side read is imitated with join() +
    the device pointers. Small (rank kernel) and large (radix passes)
    transparent counts."""
    import torch
    from vulkan_renderer_amd.parallel import wrap_device
    synth = _synth()
    monkeypatch.setenv("VKRHIP_OVERLAP_SORT", "2" if overlap == "tail" else overlap)
    monkeypatch.setenv("VKRHIP_OVERLAP_TAIL", "2" if overlap == "tail" else "0")
    sc = synth.deep_scene(260000, depth=5, seed=21, fanout=9, roots=11, blend_fraction=blend_fraction)
    vs = synth.camera_views(cascades=2, far=900.0)
    planes, required = view_set(vs)
    with hip_scene_for(sc, 3) as hs:
        hs.set_views(planes, required, vs.view)
        hs.frame()
        compare_frame(hs, oracle_frame(oracle, sc, planes, required, vs.view, vs.projection), sc)
        world = hs.read_world()
        local = hs.read_locals()
        out = hs.device_outputs()  # (the pointers stay as they are from frame to frame)
        consumer = torch.cuda.ExternalStream(hs.stream_handle())
        rng = np.random.default_rng(8)
        for pair in range(3):
            for half in range(2):
                # the first frame of a pair moves every node, the second a few subtrees
                # and leaves (a different dirty set: most world matrices stay as they are)
                moved = np.arange(sc.n) if half == 0 else np.unique(np.concatenate(
                    [rng.integers(0, 300, 6), rng.integers(0, sc.n, 500)]))
                flags = np.zeros(sc.n, np.uint8)
                flags[moved] = DIRTY_WORLD
                local = local.copy()
                local[moved, 12] += np.float32(0.5 * (pair + 1) + 0.125 * half)
                local[moved, 14] -= np.float32(0.25 * (half + 1))
                hs.upload_locals(local)
                hs.upload_node_flags(flags)
                hs.frame()
                _, world, _, _ = oracle.scene_update(sc.parent, local=local, flags=flags, world=world)
                if pair == 2 and half == 0:
                    # a frame without a cull in between (nothing moved): the frame after
                    # it follows a frame that left nothing on the second stream
                    hs.frame(STAGE_TRANSFORM)
            ref = oracle_frame(oracle, sc, planes, required, vs.view, vs.projection, world=world)
            if pair == 1:
                # a consumer on the context's stream: join, then copy the device array there
                hs.join()
                t = int(ref["payload"]["counts"][3])
                with torch.cuda.stream(consumer):
                    keys = wrap_device(out.transparent_keys, t, torch.int64).clone()
                consumer.synchronize()
                assert (keys.cpu().numpy().view(np.uint64) == ref["payload"]["sort_keys"]).all()
            compare_frame(hs, ref, sc, check_transform=False)
            assert_bits_equal(hs.read_world(), world, f"world matrices, pair {pair}")
