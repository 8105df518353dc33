"""The CPU oracle against the reference's own code (oracle/_ref), bit for bit.

Runs only where the reference checkout was available to build oracle/_ref;
elsewhere the committed golden vectors (test_oracle_golden.py) carry the same
evidence.
"""
import os

import numpy as np
import pytest

from conftest import assert_bits_equal


def _rand_matrices(rng, n, regime):
    if regime == "unit":
        return rng.uniform(-1, 1, (n, 16)).astype(np.float32)
    if regime == "wide":
        return (rng.uniform(-1, 1, (n, 16)) * 10.0 ** rng.uniform(-6, 6, (n, 16))).astype(np.float32)
    m = rng.uniform(-2, 2, (n, 16)).astype(np.float32)  # affine-like
    m[:, 3] = m[:, 7] = m[:, 11] = 0.0
    m[:, 15] = 1.0
    m[:, 12:15] *= 300.0
    return m


@pytest.mark.parametrize("regime", ["unit", "wide", "affine"])
def test_leaf_math(oracle, reference, regime):
    rng = np.random.default_rng(11)
    n = 4000
    a = _rand_matrices(rng, n, regime)
    b = _rand_matrices(rng, n, regime)
    v = (rng.uniform(-1, 1, (n, 4)) * (1.0 if regime == "unit" else 50.0)).astype(np.float32)
    for i in range(n):
        assert_bits_equal(oracle.mat4_mul(a[i], b[i]), reference.mat4_mul(a[i], b[i]), "mat4_mul")
        assert_bits_equal(oracle.mat4_mul_vec3(a[i], v[i, :3]), reference.mat4_mul_vec3(a[i], v[i, :3]),
                          "mat4_mul_vec3")
        assert_bits_equal(oracle.mat4_mul_vec4(a[i], v[i]), reference.mat4_mul_vec4(a[i], v[i]),
                          "mat4_mul_vec4")
        assert_bits_equal([oracle.vec3_dot(v[i, :3], b[i, :3])], [reference.vec3_dot(v[i, :3], b[i, :3])],
                          "vec3_dot")
        assert_bits_equal([oracle.vec3_length(v[i, :3])], [reference.vec3_length(v[i, :3])],
                          "vec3_length")


def test_trs(oracle, reference):
    rng = np.random.default_rng(12)
    for _ in range(3000):
        pos = rng.uniform(-500, 500, 3).astype(np.float32)
        q = rng.uniform(-1, 1, 4)
        q = (q / np.linalg.norm(q)).astype(np.float32)
        s = rng.uniform(0.5, 2, 3).astype(np.float32)
        assert_bits_equal(oracle.quat_to_mat4(q), reference.quat_to_mat4(q), "quat_to_mat4")
        assert_bits_equal(oracle.local_from_trs(pos, q, s), reference.local_from_trs(pos, q, s),
                          "local_from_trs")


def test_projection_builders(oracle, reference):
    rng = np.random.default_rng(13)
    for _ in range(500):
        fov = np.float32(rng.uniform(0.3, 2.5))
        aspect = np.float32(rng.uniform(0.5, 2.5))
        near = np.float32(rng.uniform(0.01, 1))
        far = np.float32(rng.uniform(10, 5000))
        assert_bits_equal(oracle.perspective(fov, aspect, near, far),
                          reference.perspective(fov, aspect, near, far), "perspective")
        l, r = np.float32(rng.uniform(-100, -1)), np.float32(rng.uniform(1, 100))
        b, t = np.float32(rng.uniform(-100, -1)), np.float32(rng.uniform(1, 100))
        assert_bits_equal(oracle.ortho_zo_yinv(l, r, b, t, near, far),
                          reference.ortho_zo_yinv(l, r, b, t, near, far), "ortho")


def _random_view_proj(rng, reference):
    eye = rng.uniform(-300, 300, 3).astype(np.float32)
    target = (eye + rng.uniform(-1, 1, 3)).astype(np.float32)
    view = reference.look_at(eye, target, np.array([0, 1, 0], np.float32))
    if rng.uniform() < 0.5:
        proj = reference.perspective(np.float32(rng.uniform(0.4, 2.0)), np.float32(rng.uniform(0.5, 2)),
                                     np.float32(0.1), np.float32(rng.uniform(50, 1000)))
    else:
        e = np.float32(rng.uniform(5, 200))
        proj = reference.ortho_zo_yinv(-e, e, -e, e, np.float32(0.1), np.float32(rng.uniform(50, 800)))
    return view, proj


def test_frustum(oracle, reference):
    rng = np.random.default_rng(14)
    for _ in range(400):
        view, proj = _random_view_proj(rng, reference)
        planes_ref = reference.frustum_from_view_projection(view, proj)
        planes = oracle.frustum_from_view_projection(view, proj)
        assert_bits_equal(planes, planes_ref, "frustum_from_view_projection")
        vp = reference.mat4_mul(proj, view)
        assert_bits_equal(oracle.frustum_from_matrix(vp), reference.frustum_from_matrix(vp),
                          "frustum_from_matrix")
        for _ in range(40):
            c = rng.uniform(-400, 400, 3).astype(np.float32)
            r = np.float32(rng.uniform(0, 30))
            assert oracle.frustum_test_sphere(planes, c, r) == reference.frustum_test_sphere(planes, c, r)


def test_degenerate_plane_is_not_normalised(oracle, reference):
    vp = np.zeros(16, np.float32)
    vp[15] = 1.0
    assert_bits_equal(oracle.frustum_from_matrix(vp), reference.frustum_from_matrix(vp), "degenerate")


def test_submesh_sphere_and_keys(oracle, reference):
    rng = np.random.default_rng(15)
    view = reference.look_at(np.array([3, 2, 6], np.float32), np.array([0, 1, -1], np.float32),
                             np.array([0, 1, 0], np.float32))
    for i in range(3000):
        pos = rng.uniform(-300, 300, 3).astype(np.float32)
        q = rng.uniform(-1, 1, 4)
        q = (q / np.linalg.norm(q)).astype(np.float32)
        s = rng.uniform(0.2, 4, 3).astype(np.float32)
        model = reference.local_from_trs(pos, q, s)
        center = rng.uniform(-1, 1, 3).astype(np.float32)
        half = rng.uniform(0.05, 2, 3).astype(np.float32)
        mn, mx = center - half, center + half
        c0, r0 = oracle.submesh_sphere(model, center, mn, mx)
        c1, r1 = reference.submesh_sphere(model, center, mn, mx)
        assert_bits_equal(c0, c1, "sphere center")
        assert_bits_equal([r0], [r1], "sphere radius")
        k0, d0 = oracle.transparent_key(view, model, center, i + 1)
        k1, d1 = reference.transparent_key(view, model, center, i + 1)
        assert k0 == k1
        assert_bits_equal([d0], [d1], "depth")
        # the multi-view classify formula (local radius precomputed)
        lr = reference.half_extent_length(mn, mx)
        wb = oracle.instance_world_bounds(model, center, lr)
        _, cr = reference.sphere_visible(model, np.array([*center, lr], np.float32),
                                         np.zeros(24, np.float32))
        assert_bits_equal(wb, cr, "world bounds")


def test_integer_helpers(oracle, reference):
    for alpha in (0, 1, 2):
        for ds in (0, 1):
            assert oracle.lib.vko_state_bucket(alpha, ds) == reference.lib.ref_state_bucket(alpha, ds)
    for rid in (0, 1, 2, 77, 0x1FFFFFFD, 0x1FFFFFFE, 0x1FFFFFFF, 0x20000000, 0xFFFFFFFF):
        expect = reference.lib.ref_picking_encode_scene(rid) if rid else 0
        assert oracle.lib.vko_picking_encode_scene_id(rid) == expect


def test_layout(reference):
    lay = reference.layout()
    assert list(lay[:28]) == [64, 16, 144, 8, 16, 20, 32, 112, 128, 132, 48, 20, 28, 32, 40, 80, 64,
                              112, 96, 80, 48, 32, 192, 262144, 65536, 1, 2, 4]


def test_payload_layout(reference):
    """vkrhip_world_payload / vkrhip_visibility_stats are byte for byte the reference's
    VkrWorldPassPayload / VkrVisibilityStats (application.h:696-750) as its own header
    compiles on this box, and the candidate capacity behind the CAP+1 convention is
    the reference's VKR_GPU_DRAW_CANDIDATE_CAPACITY."""
    import ctypes as C
    from vulkan_renderer_amd import _native as N
    lay = [int(x) for x in reference.payload_layout()]
    P, S = N.WorldPayload, N.VisibilityStats
    fields = ("gpu_candidates", "gpu_candidate_count", "gpu_camera_opaque_candidate_count",
              "gpu_shadow_candidate_count", "transmission_gpu_candidates",
              "transmission_gpu_candidate_count", "transparent_draws", "transparent_draw_count",
              "instances", "instance_count", "text_draws", "text_draw_count")
    assert lay[0] == C.sizeof(P) == 88
    assert lay[1:13] == [getattr(P, f).offset for f in fields]
    assert lay[13] == C.sizeof(S) == 12
    assert lay[14:17] == [S.objects_tested.offset, S.objects_culled_camera.offset,
                          S.objects_without_bounds.offset]
    assert lay[17] == 262144
    header = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                               "include", "vkr_hip_scene.h")).read()
    assert "#define VKRHIP_DRAW_CANDIDATE_CAPACITY 262144u" in header


def test_propagation(oracle, reference):
    from vulkan_renderer_amd import synth
    sizes = synth.level_sizes(3000, 6, fanout=4, roots=3)
    parent, offsets = synth.build_parents(sizes)
    n = parent.shape[0]
    rng = np.random.default_rng(16)
    local = _rand_matrices(rng, n, "affine")
    topo, level, levels = oracle.build_topo_order(parent)
    assert levels == len(sizes)
    assert (topo == np.arange(n)).all()  # level-ordered input is already BFS order
    world_ref = reference.propagate(topo, parent, local)
    flags = np.full(n, 0x02, np.uint8)
    _, world, fl, dirty = oracle.scene_update(parent, local=local, flags=flags)
    assert_bits_equal(world, world_ref, "world matrices")
    assert (fl == 0x08).all() and dirty.shape[0] == n


def test_sort_and_emit(oracle, reference):
    from oracle import pyoracle
    from vulkan_renderer_amd import synth
    sc = synth.flat_scene(5000, seed=5, blend_fraction=0.5)
    lo, world, _, _ = oracle.scene_update(
        sc.parent, pos=sc.pos, quat=sc.quat, scale=sc.scale, flags=np.full(sc.n, 0x01, np.uint8))
    vs = synth.camera_views()
    out = oracle.build_world_payload(world, None, sc.ident, sc.matbits, sc.bounds, vs.view,
                                     vs.projection)
    t = out["draws"].shape[0]
    assert t > 20
    # rebuild the unsorted transparent candidates and let the reference sort + emit them
    vis = np.flatnonzero(out["transparent_visible"])
    cand = np.zeros(vis.shape[0], pyoracle.TRANSPARENT_DT)
    for k, i in enumerate(vis):
        c = out["gpu_candidates"][i]
        cand[k]["model"] = c["model"]
        cand[k]["mesh"] = c["mesh"]
        cand[k]["geometry"] = c["geometry"]
        cand[k]["material"] = c["material"]
        cand[k]["submesh_index"] = c["submesh_index"]
        cand[k]["object_id"] = c["object_id"]
        cand[k]["sort_key"] = reference.transparent_key(vs.view, c["model"], sc.bounds[i, :3], i + 1)[0]
    draws, inst = reference.transparent_sort_emit(cand)
    for name in ("mesh", "geometry", "submesh_index", "material", "instance_count", "first_instance",
                 "sort_key"):
        assert (draws[name] == out["draws"][name]).all(), name
    assert (inst["object_id"] == out["instances"]["object_id"]).all()
    assert_bits_equal(inst["model"], out["instances"]["model"], "instance models")
