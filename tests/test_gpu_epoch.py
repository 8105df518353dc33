"""The candidate epoch hash (vkr_vk_candidate_epoch, vkr_vulkan_draws.c:9-44)
computed on the device by the split FNV-1a passes, against the byte-serial
oracle run over the candidate rows the same frame emitted."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _epoch_case(oracle, sc, views=1):
    from helpers import hip_scene_for, view_set
    from vulkan_renderer_amd import _native as N
    from vulkan_renderer_amd import synth
    vs = synth.camera_views(cascades=views - 1)
    planes, required = view_set(vs)
    with hip_scene_for(sc, planes.shape[0]) as hs:
        hs.set_views(planes, required, vs.view)
        hs.frame(N.STAGE_ALL | N.STAGE_CANDIDATES)
        rows = hs.read_candidates()
        got_hash, got_count = hs.candidate_epoch()
        assert got_count == int((rows["flags"] & 2 != 0).sum())
        assert got_hash == oracle.candidate_epoch(rows), "epoch hash"
        return got_hash, got_count, hs, rows


@pytest.mark.parametrize("n", [1, 63, 64, 65, 1000, 70001])
def test_epoch_flat(oracle, n):
    from vulkan_renderer_amd import synth
    sc = synth.flat_scene(n, seed=20 + n % 7, blend_fraction=0.2)
    h, count, _, _ = _epoch_case(oracle, sc)
    if n >= 63:
        assert 0 < count < n


def test_epoch_hierarchy_and_node_mapping(oracle):
    """Several candidate rows per transform, in arbitrary node order, below a
    depth-5 hierarchy."""
    from vulkan_renderer_amd import synth
    sc = synth.deep_scene(30000, depth=5, seed=3, fanout=6, roots=3)
    rng = np.random.default_rng(8)
    n_cand = 45001
    node = rng.integers(0, sc.n, n_cand).astype(np.uint32)
    ident, matbits, bounds = synth.make_candidates(9, np.arange(n_cand, dtype=np.uint64),
                                                   blend_fraction=0.25)
    sc.node, sc.ident, sc.matbits, sc.bounds = node, ident, matbits, bounds
    h, count, _, _ = _epoch_case(oracle, sc, views=3)
    assert 0 < count < n_cand


def test_epoch_no_occluders_and_empty(oracle):
    from oracle import pyoracle as P
    from vulkan_renderer_amd import synth
    sc = synth.flat_scene(500, seed=2, blend_fraction=1.0)   # every row blends: none is an occluder
    h, count, _, rows = _epoch_case(oracle, sc)
    assert count == 0 and h == oracle.candidate_epoch(np.zeros(0, P.CANDIDATE_DT))


def test_epoch_follows_the_frame(oracle):
    """Moving one occluder changes the hash; the next frame's hash matches the
    oracle again (world matrices are hashed)."""
    from helpers import hip_scene_for, view_set
    from vulkan_renderer_amd import _native as N
    from vulkan_renderer_amd import synth
    sc = synth.flat_scene(5000, seed=31, blend_fraction=0.1)
    vs = synth.camera_views(cascades=0)
    planes, required = view_set(vs)
    with hip_scene_for(sc, 1) as hs:
        hs.set_views(planes, required, vs.view)
        hs.frame(N.STAGE_ALL | N.STAGE_CANDIDATES)
        before, _ = hs.candidate_epoch()
        pos = sc.pos.copy()
        pos[1234, 0] += np.float32(0.25)
        hs.upload_trs(pos, sc.quat, sc.scale)
        hs.frame(N.STAGE_ALL | N.STAGE_CANDIDATES)
        rows = hs.read_candidates()
        after, _ = hs.candidate_epoch()
        assert after == oracle.candidate_epoch(rows)
        if rows["flags"][1234] & 2:
            assert after != before
