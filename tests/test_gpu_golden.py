"""The HIP path (through the C ABI) against the committed golden vectors, which
were produced by the reference's own compiled code (oracle/gen_golden.py). No
oracle in between: world matrices, per-view decisions, bucket ranges, visible
lists, world bounds, visibility bytes and the sorted draw rows are compared
bit for bit with the fixture arrays."""
import os

import numpy as np
import pytest

from conftest import GOLDEN, assert_bits_equal

pytestmark = pytest.mark.gpu

STAGE_TRANSFORM, STAGE_CULL, STAGE_TRANSPARENT, STAGE_WORLD_BOUNDS = 1, 2, 4, 0x10


def load(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)


def test_hierarchy_world_matrices():
    from vulkan_renderer_amd import HipScene
    g = load("hierarchy")
    n = g["parent"].shape[0]
    with HipScene(max_nodes=n, max_candidates=1, max_views=1) as hs:
        hs.set_hierarchy(g["parent"], g["level_offsets"])
        # from TRS: pass 1 builds the locals, pass 2 the worlds
        hs.upload_trs(g["pos"], g["quat"], g["scale"])
        hs.frame(STAGE_TRANSFORM)
        assert_bits_equal(hs.read_locals(), g["local"], "local matrices from TRS")
        assert_bits_equal(hs.read_world(), g["world"], "world matrices from TRS")
        assert (hs.read_node_flags() == 0x08).all()
        # from cached locals with every node marked moved
        hs.upload_worlds(np.zeros((n, 16), np.float32))
        hs.mark_all(0x02)
        hs.frame(STAGE_TRANSFORM)
        assert_bits_equal(hs.read_world(), g["world"], "world matrices from cached locals")
        assert hs.frame_info().nodes_updated == n


@pytest.mark.parametrize("fused", [True, False])
def test_cull_and_transparent(fused):
    from vulkan_renderer_amd import HipScene, build_view_set
    g = load("cull")
    n = g["ident"].shape[0]
    planes, required = build_view_set(g["camera_view"], g["camera_projection"], g["light_view_proj"])
    assert_bits_equal(planes.reshape(planes.shape[0], -1), g["planes"], "host-side plane extraction")
    assert (required == g["required"]).all()
    with HipScene(max_nodes=n, max_candidates=n, max_views=planes.shape[0]) as hs:
        hs.set_hierarchy(None, np.array([0, n], np.uint64))
        hs.upload_trs(g["pos"], g["quat"], g["scale"])
        hs.set_candidates(None, g["ident"], g["matbits"], g["bounds"])
        hs.set_views(planes, required, g["camera_view"])
        if fused:
            hs.frame(STAGE_TRANSFORM | STAGE_CULL | STAGE_TRANSPARENT | STAGE_WORLD_BOUNDS)
        else:
            hs.frame(STAGE_TRANSFORM)
            hs.frame(STAGE_CULL | STAGE_TRANSPARENT | STAGE_WORLD_BOUNDS)
        assert_bits_equal(hs.read_world(), g["world"], "world matrices")
        info = hs.frame_info()
        for v, st in enumerate(info.views):
            assert (st.bucket_counts == g["bucket_counts"][v]).all(), f"view {v} bucket counts"
            offsets = np.concatenate([[0], np.cumsum(g["bucket_counts"][v])[:-1]])
            assert (st.execution_ranges[:, 0] == offsets).all(), f"view {v} range offsets"
            assert (st.execution_ranges[:, 1] == g["bucket_counts"][v]).all()
            count = int(g["bucket_counts"][v].sum())
            assert st.visible_count == count
            assert (hs.read_visible(v, count) == g["visible"][v, :count]).all(), f"view {v} list"
        assert_bits_equal(hs.read_world_bounds(), g["world_bounds"], "world bounds")
        sphere, cflags = hs.read_candidate_prep()
        assert_bits_equal(sphere, g["sphere"], "local spheres")
        assert ((cflags & 0x7) == g["flags"]).all() and (((cflags >> 4) & 0xF) == g["state_bucket"]).all()
        assert (hs.read_transparent_visible() == g["transparent_visible"]).all()
        t = int(g["transparent_visible"].sum())
        assert info.transparent_draw_count == t
        draws, inst, keys, src = hs.read_transparent(t)
        assert draws.tobytes() == g["draws"].tobytes(), "VkrDrawItem rows"
        assert inst.tobytes() == g["instances"].tobytes(), "VkrInstanceDataGPU rows"
        assert keys.tolist() == sorted(g["unsorted_keys"].tolist(), reverse=True)


def test_edge_case_spheres_and_keys():
    """Spheres exactly on a plane, NaN / infinite centres and radii, depth
    clamping and a zero-scale model, each as a one-candidate frame."""
    from vulkan_renderer_amd import HipScene
    g = load("edge_cases")
    cases = g["sphere_cases"]
    n = cases.shape[0]
    planes = g["planes"].reshape(1, 6, 4)
    # identity model, local sphere = the case: world sphere == local sphere
    world = np.tile(np.eye(4, dtype=np.float32).reshape(16), (n, 1))
    bounds = np.zeros((n, 12), np.float32)
    ident = np.zeros((n, 8), np.uint32)
    ident[:, 7] = 1 + np.arange(n)
    with HipScene(max_nodes=n, max_candidates=n, max_views=1) as hs:
        hs.set_hierarchy(None, np.array([0, n], np.uint64))
        hs.upload_locals(world)
        hs.upload_worlds(world)
        hs.upload_node_flags(np.zeros(n, np.uint8))
        # half-extent (r,0,0) gives local radius r exactly; centre carries the case
        bounds[:, 0:3] = cases[:, 0:3]
        bounds[:, 4:7] = cases[:, 0:3]
        bounds[:, 8:11] = cases[:, 0:3]
        finite = np.isfinite(cases[:, 3])
        bounds[finite, 4] -= cases[finite, 3]
        bounds[finite, 8] += cases[finite, 3]
        usable = finite & np.isfinite(cases[:, 0:3]).all(axis=1) & (np.abs(cases[:, 0:3]).max(axis=1) < 1e6)
        matbits = np.full(n, 0x10, np.uint32)  # opaque, bounds valid
        hs.set_candidates(None, ident, matbits, bounds)
        hs.set_views(planes, np.array([2], np.uint32), g["camera_view"])
        hs.frame(STAGE_CULL)
        sphere, _ = hs.read_candidate_prep()
        st = hs.frame_info().views[0]
        vis = set(hs.read_visible(0, st.visible_count).tolist())
        checked = 0
        for i in range(n):
            if usable[i] and sphere[i, 3] == cases[i, 3]:  # the AABB reproduced the radius exactly
                assert (i in vis) == bool(cases[i, 4]), cases[i]
                checked += 1
        assert checked >= 5
    # sort keys and spheres of the special models
    center = np.array([0.25, -0.5, 0.125], np.float32)
    m = g["models"].shape[0]
    b = np.zeros((m, 12), np.float32)
    b[:, 0:3] = center
    b[:, 4:7] = [-1, -2, -3]
    b[:, 8:11] = [1, 2, 3]
    idn = np.zeros((m, 8), np.uint32)
    with HipScene(max_nodes=m, max_candidates=m, max_views=1, source_index_base=76) as hs:
        hs.set_hierarchy(None, np.array([0, m], np.uint64))
        hs.upload_worlds(g["models"])
        hs.upload_node_flags(np.zeros(m, np.uint8))
        hs.set_candidates(None, idn, np.full(m, 0x2, np.uint32), b)  # blend, bounds invalid => visible
        hs.set_views(planes, np.array([2], np.uint32), g["camera_view"])
        hs.frame(STAGE_CULL | STAGE_TRANSPARENT | STAGE_WORLD_BOUNDS)
        info = hs.frame_info()
        assert info.transparent_draw_count == m
        _, _, keys, src = hs.read_transparent(m)
        depth_bits = {int(s): int(k >> np.uint64(32)) for k, s in zip(keys, src)}
        for k in range(m):
            assert depth_bits[k] == int(g["keys"][k] >> np.uint64(32)), f"model {k} depth bits"
        got, want = hs.read_world_bounds(), g["spheres"]
        nan = np.isnan(want)
        assert (np.isnan(got) == nan).all(), "NaN placement (payload bits are not compared)"
        assert_bits_equal(np.where(nan, 0, got), np.where(nan, 0, want),
                          "world spheres of special models")
