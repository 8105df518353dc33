"""The CPU oracle against the committed golden vectors (tests/golden/).

The vectors were written by oracle/gen_golden.py from the reference's own
compiled code, so these tests pin the oracle bit for bit on machines that have
no reference checkout. reference_kats.json additionally restates the
known-answer cases of the reference's own unit tests.
"""
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN, assert_bits_equal


def load(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)


@pytest.fixture(scope="module")
def kats():
    with open(os.path.join(GOLDEN, "reference_kats.json")) as fh:
        return json.load(fh)


@pytest.mark.parametrize("regime", ["unit", "wide", "affine"])
def test_leaf_math(oracle, regime):
    g = load("leaf_math")
    a, b, v = g[f"{regime}_a"], g[f"{regime}_b"], g[f"{regime}_v"]
    n = a.shape[0]
    assert_bits_equal(np.stack([oracle.mat4_mul(a[i], b[i]) for i in range(n)]).reshape(n, 16),
                      g[f"{regime}_mul"], "mat4_mul")
    assert_bits_equal(np.stack([oracle.mat4_mul_vec3(a[i], v[i, :3]) for i in range(n)])[:, :3],
                      g[f"{regime}_mul_vec3"][:, :3], "mat4_mul_vec3")
    assert_bits_equal(np.stack([oracle.mat4_mul_vec4(a[i], v[i]) for i in range(n)]),
                      g[f"{regime}_mul_vec4"], "mat4_mul_vec4")
    assert_bits_equal([oracle.vec3_dot(v[i, :3], b[i, :3]) for i in range(n)], g[f"{regime}_dot"],
                      "vec3_dot")
    assert_bits_equal([oracle.vec3_length(v[i, :3]) for i in range(n)], g[f"{regime}_length"],
                      "vec3_length")


def test_trs_and_projections(oracle):
    g = load("leaf_math")
    n = g["trs_pos"].shape[0]
    assert_bits_equal(np.stack([oracle.quat_to_mat4(g["trs_quat"][i]) for i in range(n)]).reshape(n, 16),
                      g["trs_rotation"], "quat_to_mat4")
    assert_bits_equal(np.stack([oracle.local_from_trs(g["trs_pos"][i], g["trs_quat"][i],
                                                      g["trs_scale"][i]) for i in range(n)]).reshape(n, 16),
                      g["trs_local"], "local_from_trs")
    assert_bits_equal(np.stack([oracle.perspective(*g["persp_args"][i]) for i in range(n)]).reshape(n, 16),
                      g["persp"], "perspective")
    assert_bits_equal(np.stack([oracle.ortho_zo_yinv(*g["ortho_args"][i]) for i in range(n)]).reshape(n, 16),
                      g["ortho"], "ortho")


def test_frustum(oracle):
    g = load("frustum")
    for k in range(g["view"].shape[0]):
        planes = oracle.frustum_from_view_projection(g["view"][k], g["projection"][k])
        assert_bits_equal(planes.reshape(-1), g["planes_view_projection"][k], "planes(view, proj)")
        assert_bits_equal(oracle.frustum_from_matrix(g["view_projection"][k]).reshape(-1),
                          g["planes_matrix"][k], "planes(matrix)")
        for j in range(g["spheres"].shape[1]):
            s = g["spheres"][k, j]
            assert oracle.frustum_test_sphere(planes, s[:3], s[3]) == bool(g["visible"][k, j])
    assert_bits_equal(oracle.frustum_from_matrix(g["degenerate_matrix"]).reshape(-1),
                      g["degenerate_planes"], "degenerate planes stay unnormalised")


def test_hierarchy(oracle):
    g = load("hierarchy")
    parent = g["parent"]
    n = parent.shape[0]
    topo, _, levels = oracle.build_topo_order(parent)
    assert levels == g["level_offsets"].shape[0] - 1
    assert (topo == np.arange(n)).all()
    # from TRS (pass 1 + pass 2) and from cached locals (pass 2 only)
    lo, wo, fl, dirty = oracle.scene_update(parent, pos=g["pos"], quat=g["quat"], scale=g["scale"],
                                            flags=np.full(n, 0x01, np.uint8))
    assert_bits_equal(lo, g["local"], "local matrices")
    assert_bits_equal(wo, g["world"], "world matrices")
    assert (fl == 0x08).all() and dirty.shape[0] == n
    _, wo2, _, _ = oracle.scene_update(parent, local=g["local"], flags=np.full(n, 0x02, np.uint8))
    assert_bits_equal(wo2, g["world"], "world matrices from cached locals")


def test_cull_and_transparent(oracle):
    from oracle import pyoracle
    g = load("cull")
    n = g["ident"].shape[0]
    pay = oracle.build_world_payload(g["world"], None, g["ident"], g["matbits"], g["bounds"],
                                     g["camera_view"], g["camera_projection"])
    cand = pay["gpu_candidates"]
    assert (cand["state_bucket"] == g["state_bucket"]).all()
    assert (cand["flags"] == g["flags"]).all()
    assert_bits_equal(cand["sphere"], g["sphere"], "local spheres")
    assert (pay["transparent_visible"] == g["transparent_visible"]).all()
    assert pay["draws"].tobytes() == g["draws"].tobytes(), "VkrDrawItem rows"
    assert pay["instances"].tobytes() == g["instances"].tobytes(), "VkrInstanceDataGPU rows"
    assert sorted(pay["sort_keys"].tolist(), reverse=True) == pay["sort_keys"].tolist()
    assert sorted(g["unsorted_keys"].tolist(), reverse=True) == pay["sort_keys"].tolist()

    planes = oracle_planes(oracle, g)
    assert_bits_equal(planes, g["planes"], "view planes")
    cls, states, vis, _ = oracle.multiview_cull(cand, planes, g["required"])
    for v in range(planes.shape[0]):
        passed = cls[v] != pyoracle.CULLED
        assert (passed == g["passed"][v].astype(bool)).all(), f"view {v} decisions"
        assert (states[v]["bucket_counts"] == g["bucket_counts"][v]).all(), f"view {v} counts"
        count = int(states[v]["visible_count"])
        assert count == int(g["bucket_counts"][v].sum())
        assert (vis[v, :count] == g["visible"][v, :count]).all(), f"view {v} visible list"
    for i in range(0, n, 7):
        assert_bits_equal(oracle.instance_world_bounds(g["world"][i], g["sphere"][i, :3],
                                                       g["sphere"][i, 3]), g["world_bounds"][i],
                          "world bounds")


def oracle_planes(oracle, g):
    planes = [oracle.frustum_from_view_projection(g["camera_view"], g["camera_projection"]).reshape(-1)]
    planes += [oracle.frustum_from_matrix(m).reshape(-1) for m in g["light_view_proj"]]
    return np.stack(planes)


def test_edge_cases(oracle):
    g = load("edge_cases")
    for row in g["sphere_cases"]:
        assert oracle.frustum_test_sphere(g["planes"], row[:3], row[3]) == bool(row[4]), row
    center = np.array([0.25, -0.5, 0.125], np.float32)
    for k, m in enumerate(g["models"]):
        key, depth = oracle.transparent_key(g["camera_view"], m, center, 77)
        assert key == int(g["keys"][k])
        assert_bits_equal([depth], [g["depths"][k]], "depth")
        c, r = oracle.submesh_sphere(m, center, np.array([-1, -2, -3], np.float32),
                                     np.array([1, 2, 3], np.float32))
        assert_bits_equal([*c, r], g["spheres"][k], "sphere")


# ---- the reference's own known-answer tests --------------------------------------
def test_kat_grid_never_rejects_visible_points(oracle, kats):
    k = kats["grid"]
    view, proj = np.array(k["view"], np.float32), np.array(k["projection"], np.float32)
    planes = oracle.frustum_from_view_projection(view, proj)
    vp = oracle.mat4_mul(proj, view)
    lo, hi, step = k["range"]
    seen = 0
    for x in range(lo, hi + 1, step):
        for y in range(lo, hi + 1, step):
            for z in range(lo, hi + 1, step):
                p = np.array([x, y, z, 1], np.float32)
                c = oracle.mat4_mul_vec4(vp, p)
                inside = (c[3] > 0 and 0 <= c[2] <= c[3] and -c[3] <= c[0] <= c[3]
                          and -c[3] <= c[1] <= c[3])
                if inside:
                    seen += 1
                    assert oracle.frustum_test_sphere(planes, p[:3], np.float32(k["radius"]))
    assert seen > 0


def test_kat_ortho_depth(oracle, kats):
    k = kats["ortho_depth"]
    planes = oracle.frustum_from_view_projection(np.array(k["view"], np.float32),
                                                 np.array(k["projection"], np.float32))
    for p, expect in zip(k["points"], k["visible"]):
        assert oracle.frustum_test_sphere(planes, np.array(p, np.float32), np.float32(0)) == expect


def test_kat_sort_and_emit(oracle, kats):
    from oracle import pyoracle
    k = kats["sort_emit"]
    # three visible blend rows whose keys are forced through identity matrices
    # placed at depths that give the KAT's key order 10 < 20 < 30
    n = 3
    world = np.tile(np.eye(4, dtype=np.float32).reshape(16), (n, 1))
    depth_of = {10: 1.0, 20: 2.0, 30: 3.0}
    for i, key in enumerate(k["sort_key"]):
        world[i, 14] = -depth_of[key]
    ident = np.zeros((n, 8), np.uint32)
    for i in range(n):
        ident[i] = [*k["mesh"][i], *k["geometry"][i], 0, *k["material"][i], k["object_id"][i]]
    matbits = np.full(n, 2, np.uint32)  # blend, bounds invalid => visible
    bounds = np.zeros((n, 12), np.float32)
    view = np.eye(4, dtype=np.float32).reshape(16)
    pay = oracle.build_world_payload(world, None, ident, matbits, bounds, view, view)
    draws = pay["draws"]
    assert draws.shape[0] == 3
    assert [int(m[0]) for m in draws["material"]] == k["sorted_material_id"]
    assert (draws["instance_count"] == 1).all()
    assert draws["first_instance"].tolist() == [0, 1, 2]
    assert sorted(draws["sort_key"].tolist(), reverse=True) == draws["sort_key"].tolist()
    expect_ids = [oracle.lib.vko_picking_encode_scene_id(o) for o in k["sorted_object_id"]]
    assert pay["instances"]["object_id"].tolist() == expect_ids
    assert pyoracle.DRAW_DT.itemsize == 48 and pyoracle.INSTANCE_DT.itemsize == 80


def test_kat_scaled_sphere(oracle, kats):
    k = kats["scaled_sphere"]
    model = np.eye(4, dtype=np.float32)
    for a in range(3):
        model[a, a] = k["scale"][a]
    model = model.reshape(16)
    model[12:15] = k["translation"]
    c, r = oracle.submesh_sphere(model, np.array(k["center"], np.float32),
                                 np.array(k["min"], np.float32), np.array(k["max"], np.float32))
    assert np.abs(c - np.array(k["expected_center"], np.float32)).max() < k["tolerance"]
    assert abs(r - np.sqrt(np.float32(k["expected_radius_sq"]))) < k["tolerance"]


def test_kat_routing_and_buckets(oracle, kats):
    k = kats["routing"]
    b = k["buckets"]
    sb = oracle.lib.vko_state_bucket
    assert sb(0, 0) == b["opaque_back"] and sb(0, 1) == b["opaque_double"]
    assert sb(1, 0) == b["cutout_back"] and sb(1, 1) == b["cutout_double"]


def test_kat_parent_child(oracle, kats):
    k = kats["parent_child"]
    parent = np.array([0xFFFFFFFF, 0], np.uint32)
    pos = np.array([[*k["parent_pos"], 0], [*k["child_pos"], 0]], np.float32)
    quat = np.array([k["parent_quat"], k["child_quat"]], np.float32)
    scale = np.ones((2, 4), np.float32)
    _, world, _, _ = oracle.scene_update(parent, pos=pos, quat=quat, scale=scale,
                                         flags=np.full(2, 0x01, np.uint8))
    assert np.abs(world[1] - np.array(k["expected_world"], np.float32)).max() < k["tolerance"]
    assert_bits_equal(world[1], np.array(k["expected_world"], np.float32), "child world")


def test_effective_visibility_equals_the_parent_chain_walk(oracle):
    """The library's parents-first pass (vkrhip_effective_visibility) against the
    oracle's literal walk of scene_entity_is_visible, on random hierarchies with
    every combination of {no component, hidden, visible+inherit, visible+own}."""
    from vulkan_renderer_amd import effective_visibility, levelize, synth
    rng = np.random.default_rng(77)
    for n, depth in ((1, 1), (50, 3), (4000, 6), (4000, 12)):
        sizes = synth.level_sizes(n, depth, fanout=3, roots=2)
        parent, _ = synth.build_parents(sizes)
        n = parent.shape[0]
        spec = rng.choice(np.array([0, 1, 3, 5, 7, 2, 6], np.uint8), n,
                          p=[0.4, 0.05, 0.1, 0.05, 0.3, 0.05, 0.05])
        got = effective_visibility(parent, spec)
        assert (got == oracle.entity_is_visible(parent, spec)).all()
        assert 0 < got.sum() < n or n == 1
        # the same in a shuffled numbering, brought parents-first by levelize
        perm = rng.permutation(n).astype(np.uint32)
        inv = np.empty(n, np.uint32)
        inv[perm] = np.arange(n, dtype=np.uint32)
        parent_s = np.where(parent[perm] == 0xFFFFFFFF, 0xFFFFFFFF, inv[np.minimum(parent[perm], n - 1)]).astype(np.uint32)
        spec_s = spec[perm]
        expect_s = oracle.entity_is_visible(parent_s, spec_s)
        order, new_parent, _ = levelize(parent_s)
        got_s = np.empty(n, np.uint8)
        got_s[order] = effective_visibility(new_parent, spec_s[order])
        assert (got_s == expect_s).all()


def test_hzb_occlusion_hand_cases(oracle):
    """vko_hzb_occluded on cases that can be checked by hand (deferred.slang:240-303)."""
    from helpers import depth_pyramid
    from vulkan_renderer_amd import mat4_mul, synth
    vs = synth.camera_views(cascades=0)          # at the origin, looking down -Z, near 0.1, far 400
    vp = mat4_mul(vs.projection, vs.view)
    w, h = 64, 64
    # wall over the left half at the depth of z = -10: ndc z = far/(far-near) * (1 - near/10)
    wall = np.float32(400.0 / 399.9 * (1.0 - 0.1 / 10.0))
    oracle.set_hzb(depth_pyramid(w, h, [(0, 0, 32, 64, wall)]), w, h, vp, 0.0)
    try:
        assert oracle.hzb_occluded([-20.0, 0.0, -50.0], 1.0)        # left, behind the wall
        assert not oracle.hzb_occluded([-3.0, 0.0, -5.0], 0.5)      # left, in front of it
        assert not oracle.hzb_occluded([20.0, 0.0, -50.0], 1.0)     # right half: background
        assert not oracle.hzb_occluded([0.0, 0.0, -50.0], 2.0)      # straddles the wall's edge
        assert not oracle.hzb_occluded([0.0, 0.0, 5.0], 1.0)        # behind the camera: w <= 0
        assert not oracle.hzb_occluded([-1.0, 0.0, -0.5], 1.0)      # cube reaches behind the near plane
        assert oracle.hzb_occluded([-20.0, 0.0, -50.0], float("nan"))  # max(NaN, 0) = 0: a point
        assert oracle.hzb_occluded([-20.0, 0.0, -50.0], -5.0)       # negative radius clamps to 0
    finally:
        oracle.set_hzb(None)


def test_hzb_build_matches_a_numpy_statement(oracle):
    """vko_build_hzb against the 2x2 maximum written with numpy slices (non-negative
    depths, so the shader's starting value 0 does not show), plus its two special
    rules: level 0 is a clamped copy, and nothing below 0 survives a reduction."""
    from helpers import depth_pyramid
    rng = np.random.default_rng(3)
    for w, h in ((64, 32), (37, 21), (1, 9)):
        depth = rng.uniform(0.0, 1.0, (h, w)).astype(np.float32)
        expect = depth_pyramid(w, h, [])          # structure only; fill level 0 with the image
        expect[0] = depth
        for m in range(1, len(expect)):
            src = expect[m - 1]
            hh, ww = expect[m].shape
            for y in range(hh):
                for x in range(ww):
                    expect[m][y, x] = src[min(2 * y, src.shape[0] - 1): 2 * y + 2,
                                          min(2 * x, src.shape[1] - 1): 2 * x + 2].max()
        got = oracle.build_hzb(depth, w, h, len(expect))
        for m, level in enumerate(expect):
            assert got[m].tobytes() == level.tobytes(), f"{w}x{h} level {m}"
    neg = np.full((4, 4), -1.0, np.float32)
    got = oracle.build_hzb(neg, 6, 6, 2)
    assert got[0].shape == (6, 6) and (got[0] == -1.0).all()      # copy, clamped at the source edge
    assert (got[1] == 0.0).all()                                   # max(0, ...) from then on


def test_bench_cpu_baseline_legs():
    """The benchmark's CPU legs (one core, and the same loop on several cores at
    once) on a small flat scene: both report a rate and say what was sampled."""
    import bench
    from vulkan_renderer_amd import build_view_set, synth
    sc = synth.flat_scene(4096, seed=1)
    vs = synth.camera_views()
    planes, required = build_view_set(vs.view, vs.projection, vs.light_view_proj)
    out = bench.cpu_baseline(sc, planes, required, vs, 0.3)
    assert out["cores"] == 1 and out["kind"] == "port" and out["value"] > 0
    many = out["all_cores"]
    assert many["cores"] >= 1 and many["value"] > 0 and "not in the reference" in many["sample"]
