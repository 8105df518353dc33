"""Config 1: the reference's bundled scene files through the whole path.

tests/golden/config1/ holds copies of the reference's scene and material data
files (assets/scenes/fixtures/deferred_state_matrix.scene.json,
assets/scenes/default.scene.json, tests/fixtures/rendering/p20_*.mt). The CPU
part checks the reader and the oracle; the GPU part compares the HIP path with
the oracle bit for bit on the same scene."""
import os

import numpy as np
import pytest

from conftest import GOLDEN, assert_bits_equal

CONFIG1 = os.path.join(GOLDEN, "config1")


def camera():
    """Camera of the headless config: eye (0,2,6) looking at (0,1.4,-1), 60 deg, 16:9."""
    from vulkan_renderer_amd import synth
    return synth.camera_views(fov_deg=60.0, aspect=16.0 / 9.0, near=0.1, far=1000.0, cascades=4,
                              eye=(0.0, 2.0, 6.0), target=(0.0, 1.4, -1.0))


def prepared(name):
    from vulkan_renderer_amd import levelize
    from vulkan_renderer_amd.scene_json import load_scene
    ls = load_scene(os.path.join(CONFIG1, name))
    order, new_parent, offsets = levelize(ls.parent)
    new_id = np.empty_like(order)
    new_id[order] = np.arange(order.shape[0], dtype=np.uint32)
    return ls, order, new_parent, offsets, new_id


def test_state_matrix_scene_reads_as_the_reference_describes_it():
    ls, order, new_parent, offsets, _ = prepared("deferred_state_matrix.scene.json")
    assert len(ls.names) == 11 and ls.names[0] == "Root"
    assert (ls.parent[1:] == 0).all() and ls.parent[0] == 0xFFFFFFFF
    assert offsets.tolist() == [0, 1, 11]
    by_name = {ls.names[e]: int(b) for e, b in zip(ls.cand_entity, ls.matbits)}
    assert len(by_name) == 9                       # backdrop + 4 opaque/cutout + 4 transmission
    assert by_name["OpaqueSingle"] == 0x10 and by_name["OpaqueDouble"] == 0x18
    assert by_name["CutoutSingle"] == 0x11 and by_name["CutoutDouble"] == 0x19
    assert by_name["TransmissionOpaqueSingle"] == 0x14 and by_name["TransmissionCutoutDouble"] == 0x1D
    assert not ls.notes


def oracle_scene(oracle, ls, order, new_parent, new_id, vs, planes, required):
    pos, quat, scale = ls.pos[order], ls.quat[order], ls.scale[order]
    local, world, flags, dirty = oracle.scene_update(new_parent, pos=pos, quat=quat, scale=scale,
                                                     flags=np.full(order.shape[0], 0x01, np.uint8))
    node = new_id[ls.cand_entity]
    pay = oracle.build_world_payload(world, node, ls.ident, ls.matbits, ls.bounds, vs.view,
                                     vs.projection)
    cls, states, vis, ties = oracle.multiview_cull(pay["gpu_candidates"], planes, required)
    return dict(local=local, world=world, flags=flags, dirty=dirty, payload=pay, cls=cls,
                states=states, visible=vis, ties=ties), node, (pos, quat, scale)


@pytest.mark.parametrize("name", ["deferred_state_matrix.scene.json", "default.scene.json"])
def test_oracle_runs_the_bundled_scenes(oracle, name):
    from helpers import view_set
    ls, order, new_parent, offsets, new_id = prepared(name)
    vs = camera()
    planes, required = view_set(vs)
    ref, node, _ = oracle_scene(oracle, ls, order, new_parent, new_id, vs, planes, required)
    pay = ref["payload"]
    assert int(pay["counts"][0]) == ls.ident.shape[0]
    if name.startswith("deferred"):
        # 4 transmissive rows go to the transmission stream and never to view 0
        assert int(pay["counts"][2]) == 4
        assert int(pay["counts"][1]) == 5
        assert int(ref["states"][0]["visible_count"]) == 5      # everything faces the camera
        assert ref["states"][0]["bucket_counts"].tolist() == [2, 1, 1, 1]
        # children inherit the identity root: world == local
        assert_bits_equal(ref["world"], ref["local"], "flat children")


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["deferred_state_matrix.scene.json", "default.scene.json"])
def test_hip_path_matches_oracle_on_the_bundled_scenes(oracle, name):
    from helpers import compare_frame, view_set
    from vulkan_renderer_amd import HipScene
    ls, order, new_parent, offsets, new_id = prepared(name)
    vs = camera()
    planes, required = view_set(vs)
    ref, node, (pos, quat, scale) = oracle_scene(oracle, ls, order, new_parent, new_id, vs, planes,
                                                 required)

    class Shape:
        n = order.shape[0]
        ident = ls.ident

    with HipScene(max_nodes=Shape.n, max_candidates=ls.ident.shape[0], max_views=planes.shape[0]) as hs:
        hs.set_hierarchy(new_parent, offsets)
        hs.upload_trs(pos, quat, scale)
        hs.set_candidates(node, ls.ident, ls.matbits, ls.bounds)
        hs.set_views(planes, required, vs.view)
        hs.frame()
        compare_frame(hs, ref, Shape)
