#!/usr/bin/env python3
"""Generates tests/golden/*.npz from the reference's own compiled code.

TEST INFRASTRUCTURE. Runs only where oracle/_ref/libvkr_ref.so could be built
(the reference checkout is present); the fixtures it writes travel with the
repository so that the CPU oracle - and through it the HIP path - stays pinned
on machines that have no reference checkout.

Every floating-point value in the fixtures is produced by a reference function
(or a composition of reference functions in oracle/ref_shim.c). Integer
bookkeeping that the reference only has in shader form (bucket counts, the
ascending visible lists) is derived here from the reference's per-row decisions.
Inputs come from vulkan_renderer_amd.synth (counter-based RNG, seeds below).

    python oracle/gen_golden.py            # rewrites tests/golden/
"""
from __future__ import annotations

import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import pyoracle  # noqa: E402
from vulkan_renderer_amd import synth  # noqa: E402

GOLDEN = os.path.join(ROOT, "tests", "golden")


def rand_matrices(rng, n, regime):
    if regime == "unit":
        return rng.uniform(-1, 1, (n, 16)).astype(np.float32)
    if regime == "wide":
        return (rng.uniform(-1, 1, (n, 16)) * 10.0 ** rng.uniform(-6, 6, (n, 16))).astype(np.float32)
    m = rng.uniform(-2, 2, (n, 16)).astype(np.float32)  # affine-like
    m[:, 3] = m[:, 7] = m[:, 11] = 0.0
    m[:, 15] = 1.0
    m[:, 12:15] *= 300.0
    return m


def gen_leaf_math(ref):
    rng = np.random.default_rng(101)
    out = {}
    n = 256
    for regime in ("unit", "wide", "affine"):
        a = rand_matrices(rng, n, regime)
        b = rand_matrices(rng, n, regime)
        v = (rng.uniform(-1, 1, (n, 4)) * (1.0 if regime == "unit" else 50.0)).astype(np.float32)
        out[f"{regime}_a"], out[f"{regime}_b"], out[f"{regime}_v"] = a, b, v
        out[f"{regime}_mul"] = np.stack([ref.mat4_mul(a[i], b[i]) for i in range(n)])
        out[f"{regime}_mul_vec3"] = np.stack([ref.mat4_mul_vec3(a[i], v[i, :3]) for i in range(n)])
        out[f"{regime}_mul_vec4"] = np.stack([ref.mat4_mul_vec4(a[i], v[i]) for i in range(n)])
        out[f"{regime}_dot"] = np.array([ref.vec3_dot(v[i, :3], b[i, :3]) for i in range(n)], np.float32)
        out[f"{regime}_length"] = np.array([ref.vec3_length(v[i, :3]) for i in range(n)], np.float32)
    pos = rng.uniform(-500, 500, (n, 3)).astype(np.float32)
    q = rng.uniform(-1, 1, (n, 4))
    q = (q / np.linalg.norm(q, axis=1, keepdims=True)).astype(np.float32)
    s = rng.uniform(0.5, 2, (n, 3)).astype(np.float32)
    out["trs_pos"], out["trs_quat"], out["trs_scale"] = pos, q, s
    out["trs_rotation"] = np.stack([ref.quat_to_mat4(q[i]) for i in range(n)])
    out["trs_local"] = np.stack([ref.local_from_trs(pos[i], q[i], s[i]) for i in range(n)])
    persp = np.stack([rng.uniform(0.3, 2.5, n), rng.uniform(0.5, 2.5, n), rng.uniform(0.01, 1, n),
                      rng.uniform(10, 5000, n)], axis=1).astype(np.float32)
    ortho = np.stack([rng.uniform(-100, -1, n), rng.uniform(1, 100, n), rng.uniform(-100, -1, n),
                      rng.uniform(1, 100, n), rng.uniform(0.01, 1, n), rng.uniform(10, 5000, n)],
                     axis=1).astype(np.float32)
    out["persp_args"], out["ortho_args"] = persp, ortho
    out["persp"] = np.stack([ref.perspective(*persp[i]) for i in range(n)])
    out["ortho"] = np.stack([ref.ortho_zo_yinv(*ortho[i]) for i in range(n)])
    return out


def gen_frustum(ref):
    rng = np.random.default_rng(102)
    nv, ns = 64, 48
    views, projs, planes_vp, vps, planes_m = [], [], [], [], []
    spheres = np.zeros((nv, ns, 4), np.float32)
    visible = np.zeros((nv, ns), np.uint8)
    for k in range(nv):
        eye = rng.uniform(-300, 300, 3).astype(np.float32)
        target = (eye + rng.uniform(-1, 1, 3)).astype(np.float32)
        view = ref.look_at(eye, target, np.array([0, 1, 0], np.float32))
        if k % 2 == 0:
            proj = ref.perspective(np.float32(rng.uniform(0.4, 2.0)), np.float32(rng.uniform(0.5, 2)),
                                   np.float32(0.1), np.float32(rng.uniform(50, 1000)))
        else:
            e = np.float32(rng.uniform(5, 200))
            proj = ref.ortho_zo_yinv(-e, e, -e, e, np.float32(0.1), np.float32(rng.uniform(50, 800)))
        pl = ref.frustum_from_view_projection(view, proj)
        vp = ref.mat4_mul(proj, view)
        views.append(view), projs.append(proj), planes_vp.append(pl), vps.append(vp)
        planes_m.append(ref.frustum_from_matrix(vp))
        for j in range(ns):
            c = (eye + rng.uniform(-400, 400, 3)).astype(np.float32)
            r = np.float32(rng.uniform(0, 30))
            spheres[k, j] = [*c, r]
            visible[k, j] = ref.frustum_test_sphere(pl, c, r)
    degenerate = np.zeros(16, np.float32)
    degenerate[15] = 1.0
    return dict(view=np.stack(views), projection=np.stack(projs), planes_view_projection=np.stack(planes_vp),
                view_projection=np.stack(vps), planes_matrix=np.stack(planes_m), spheres=spheres,
                visible=visible, degenerate_matrix=degenerate,
                degenerate_planes=ref.frustum_from_matrix(degenerate))


def gen_hierarchy(ref):
    sizes = synth.level_sizes(4096, 8, fanout=3, roots=2)
    parent, offsets = synth.build_parents(sizes)
    n = parent.shape[0]
    index = np.arange(n, dtype=np.uint64)
    pos, quat, scale = synth.make_trs(7, index, 20.0, (0.8, 1.25))
    local = np.stack([ref.local_from_trs(pos[i, :3], quat[i], scale[i, :3]) for i in range(n)])
    topo = np.arange(n, dtype=np.uint32)  # level order is the reference's BFS order
    world = ref.propagate(topo, parent, local)
    return dict(parent=parent, level_offsets=np.asarray(offsets, np.uint64), pos=pos, quat=quat,
                scale=scale, local=local, world=world)


def material_fields(ref, matbits):
    """(state_bucket, flags, blend, transmissive) per row from the reference's
    routing helpers (vkr_visibility.h:12-39) and the flag rule of
    application_emit_world_source (application.h:632-638)."""
    n = matbits.shape[0]
    bucket = np.zeros(n, np.uint32)
    flags = np.zeros(n, np.uint32)
    blend = np.zeros(n, bool)
    for i, m in enumerate(matbits):
        alpha, transmissive, double_sided, bounds_valid = m & 3, bool(m & 4), bool(m & 8), bool(m & 16)
        routing = np.zeros(2, np.uint32)
        ref.lib.ref_alpha_routing(int(alpha), routing)
        blend[i] = bool(routing[0])
        bucket[i] = ref.lib.ref_state_bucket(int(alpha), int(double_sided))
        f = 4  # SHADOW_CASTER
        if bounds_valid:
            f |= 1
        if not transmissive and not blend[i]:
            f |= 2
        flags[i] = f
    return bucket, flags, blend


def gen_cull(ref):
    n = 4096
    sc = synth.flat_scene(n, seed=3, blend_fraction=0.25, no_bounds_fraction=0.03)
    # the default camera sees ~9 % of a +-500 cube; pull the instances in so that
    # every bucket of every view is populated at this size
    pos = sc.pos.copy()
    pos[:, :3] *= np.float32(0.25)
    world = np.stack([ref.local_from_trs(pos[i, :3], sc.quat[i], sc.scale[i, :3]) for i in range(n)])
    vs = synth.camera_views(cascades=4)
    views = [ref.mat4_mul(vs.projection, vs.view)] + [m for m in vs.light_view_proj]
    planes = np.stack([ref.frustum_from_view_projection(vs.view, vs.projection)] +
                      [ref.frustum_from_matrix(m) for m in vs.light_view_proj])
    required = np.array([2] + [4] * 4, np.uint32)
    bucket, flags, blend = material_fields(ref, sc.matbits)
    center, mn, mx = sc.bounds[:, 0:3], sc.bounds[:, 4:7], sc.bounds[:, 8:11]
    sphere = np.zeros((n, 4), np.float32)
    sphere[:, :3] = center
    sphere[:, 3] = [ref.half_extent_length(mn[i], mx[i]) for i in range(n)]
    nviews = planes.shape[0]
    passed = np.zeros((nviews, n), np.uint8)
    world_bounds = np.stack([ref.sphere_visible(world[i], sphere[i], planes[0])[1] for i in range(n)])
    for v in range(nviews):
        for i in range(n):
            ok = (flags[i] & required[v]) == required[v] and bucket[i] < 4
            if ok and (flags[i] & 1):
                ok = ref.sphere_visible(world[i], sphere[i], planes[v])[0]
            passed[v, i] = ok
    # prefix / encode bookkeeping (deferred.slang:354-420) on the reference's decisions
    counts = np.zeros((nviews, 4), np.uint32)
    visible = np.full((nviews, n), 0xFFFFFFFF, np.uint32)
    for v in range(nviews):
        cursor = 0
        for b in range(4):
            rows = np.flatnonzero(passed[v].astype(bool) & (bucket == b)).astype(np.uint32)
            counts[v, b] = rows.shape[0]
            visible[v, cursor:cursor + rows.shape[0]] = rows
            cursor += rows.shape[0]
    # transparent rows: visibility byte, keys, sorted draws (application.h:790-803,940-961)
    tvis = np.zeros(n, np.uint8)
    for i in range(n):
        if blend[i] and not (sc.matbits[i] & 4):
            if sc.matbits[i] & 16:
                c, r = ref.submesh_sphere(world[i], center[i], mn[i], mx[i])
                tvis[i] = ref.frustum_test_sphere(planes[0], c, r)
            else:
                tvis[i] = 1
    rows = np.flatnonzero(tvis)
    cand = np.zeros(rows.shape[0], pyoracle.TRANSPARENT_DT)
    for k, i in enumerate(rows):
        cand[k]["model"] = world[i]
        cand[k]["mesh"] = sc.ident[i, 0:2]
        cand[k]["geometry"] = sc.ident[i, 2:4]
        cand[k]["submesh_index"] = sc.ident[i, 4]
        cand[k]["material"] = sc.ident[i, 5:7]
        rid = int(sc.ident[i, 7])
        cand[k]["object_id"] = ref.lib.ref_picking_encode_scene(rid) if rid else 0
        cand[k]["sort_key"] = ref.transparent_key(vs.view, world[i], center[i], int(i) + 1)[0]
    draws, inst = ref.transparent_sort_emit(cand)
    return dict(pos=pos, quat=sc.quat, scale=sc.scale, ident=sc.ident, matbits=sc.matbits,
                bounds=sc.bounds, world=world, camera_view=vs.view, camera_projection=vs.projection,
                light_view_proj=vs.light_view_proj, view_projection=np.stack(views), planes=planes,
                required=required, state_bucket=bucket, flags=flags, sphere=sphere,
                world_bounds=world_bounds, passed=passed, bucket_counts=counts, visible=visible,
                transparent_visible=tvis, unsorted_keys=cand["sort_key"].copy(),
                draws=np.frombuffer(draws.tobytes(), np.uint8),
                instances=np.frombuffer(inst.tobytes(), np.uint8))


def gen_edge_cases(ref):
    """Boundary behaviour: a sphere exactly on a plane, NaN and infinite inputs,
    depth clamping, a zero-scale model."""
    vs = synth.camera_views()
    planes = ref.frustum_from_view_projection(vs.view, vs.projection)
    ident = np.eye(4, dtype=np.float32).reshape(16)
    cases = []
    # near plane of the default camera is z = -0.1: sphere centre at distance r behind it
    for c, r in [((0, 0, 0.9), 1.0), ((0, 0, 0.9000001), 1.0), ((0, 0, 0.8999999), 1.0),
                 ((0, 0, -0.1), 0.0), ((0, 0, -400.0), 0.0), ((0, 0, -400.5), 0.5),
                 ((np.nan, 0, -5), 1.0), ((0, 0, -5), np.nan), ((np.inf, 0, -5), 1.0),
                 ((0, 0, -5), np.inf), ((0, 0, 5), np.inf), ((1e30, 1e30, -1e30), 1e30)]:
        c = np.array(c, np.float32)
        cases.append([*c, np.float32(r), ref.frustum_test_sphere(planes, c, np.float32(r))])
    models = [ident.copy() for _ in range(5)]
    models[1][0] = models[1][5] = models[1][10] = 0.0          # zero scale
    models[2][12:15] = [0, 0, 10]                               # behind the camera: depth clamps to 0
    models[3][12:15] = [0, 0, -10]
    models[4][0] = np.nan
    keys, depths, spheres = [], [], []
    for m in models:
        k, d = ref.transparent_key(vs.view, m, np.array([0.25, -0.5, 0.125], np.float32), 77)
        keys.append(k), depths.append(d)
        c, r = ref.submesh_sphere(m, np.array([0.25, -0.5, 0.125], np.float32),
                                  np.array([-1, -2, -3], np.float32), np.array([1, 2, 3], np.float32))
        spheres.append([*c, r])
    return dict(planes=planes, camera_view=vs.view, sphere_cases=np.array(cases, np.float32),
                models=np.stack(models), keys=np.array(keys, np.uint64),
                depths=np.array(depths, np.float32), spheres=np.array(spheres, np.float32))


def gen_reference_kats(ref):
    """The known-answer cases of the reference's own unit tests, re-expressed as
    data (inputs, plus the matrices its builders give for them) with the answers
    those tests assert."""
    up = np.array([0, 1, 0], np.float32)
    zero = np.zeros(3, np.float32)
    fwd = np.array([0, 0, -1], np.float32)
    kats = {}
    # tests/src/visibility_test.c:193-213
    view = ref.look_at(zero, fwd, up)
    proj = ref.perspective(ref.to_radians(np.float32(60.0)), np.float32(1.0), np.float32(0.1),
                           np.float32(100.0))
    kats["grid"] = {"source": "tests/src/visibility_test.c:193-213",
                    "view": view.tolist(), "projection": proj.tolist(),
                    "range": [-60, 60, 5], "radius": 0.0,
                    "rule": "every grid point inside the clip volume must pass the sphere test"}
    # tests/src/visibility_test.c:215-225
    oproj = ref.ortho_zo_yinv(*[np.float32(x) for x in (-10, 10, -10, 10, 0.1, 100)])
    kats["ortho_depth"] = {"source": "tests/src/visibility_test.c:215-225",
                           "view": view.tolist(), "projection": oproj.tolist(),
                           "points": [[0, 0, -0.11], [0, 0, -99.9], [0, 0, 1.0], [0, 0, -101.0]],
                           "visible": [True, True, False, False]}
    # tests/src/visibility_test.c:227-258
    kats["sort_emit"] = {"source": "tests/src/visibility_test.c:227-258",
                         "mesh": [[1, 1], [2, 1], [3, 1]], "geometry": [[11, 1], [12, 1], [13, 1]],
                         "material": [[21, 1], [22, 1], [23, 1]], "object_id": [31, 32, 33],
                         "sort_key": [10, 30, 20], "sorted_keys": [30, 20, 10],
                         "sorted_object_id": [32, 33, 31], "sorted_material_id": [22, 23, 21]}
    # tests/src/visibility_test.c:260-279
    kats["scaled_sphere"] = {"source": "tests/src/visibility_test.c:260-279",
                             "scale": [2, 3, 4], "translation": [7, -2, 5], "center": [1, 2, 3],
                             "min": [-1, -2, -3], "max": [1, 2, 3],
                             "expected_center": [9, 4, 17], "expected_radius_sq": 14 * 16,
                             "tolerance": 1e-4}
    # tests/src/visibility_test.c:156-181
    kats["routing"] = {"source": "tests/src/visibility_test.c:156-181",
                       "alpha_modes": {"opaque": 0, "cutout": 1, "blend": 2},
                       "world_transparent": [False, False, True],
                       "shadow_alpha_tested": [False, True, False],
                       "buckets": {"opaque_back": 0, "opaque_double": 1, "cutout_back": 2,
                                   "cutout_double": 3}}
    # tests/src/transform_test.c:118-147: parent (2,0,0) rotated 90 deg about Y, child (0,1,0)
    # rotated 45 deg about X; quaternion = (axis*sin(a/2), cos(a/2))
    def axis_angle(axis, angle):
        s = np.float32(np.sin(np.float32(angle) * np.float32(0.5)))
        c = np.float32(np.cos(np.float32(angle) * np.float32(0.5)))
        return np.array([axis[0] * s, axis[1] * s, axis[2] * s, c], np.float32)
    pq = axis_angle((0, 1, 0), np.pi / 2)
    cq = axis_angle((1, 0, 0), np.pi / 4)
    one = np.ones(3, np.float32)
    pl = ref.local_from_trs(np.array([2, 0, 0], np.float32), pq, one)
    cl = ref.local_from_trs(np.array([0, 1, 0], np.float32), cq, one)
    kats["parent_child"] = {"source": "tests/src/transform_test.c:118-147",
                            "parent_pos": [2, 0, 0], "parent_quat": pq.tolist(),
                            "child_pos": [0, 1, 0], "child_quat": cq.tolist(),
                            "expected_world": ref.mat4_mul(pl, cl).tolist(), "tolerance": 1e-4}
    return kats


def main() -> int:
    pyoracle.build()
    if not pyoracle.have_reference():
        print("oracle/_ref/libvkr_ref.so is not built: the reference checkout is needed", file=sys.stderr)
        return 1
    ref = pyoracle.Reference()
    os.makedirs(GOLDEN, exist_ok=True)
    for name, fn in [("leaf_math", gen_leaf_math), ("frustum", gen_frustum),
                     ("hierarchy", gen_hierarchy), ("cull", gen_cull), ("edge_cases", gen_edge_cases)]:
        data = fn(ref)
        path = os.path.join(GOLDEN, name + ".npz")
        np.savez_compressed(path, **data)
        print(f"{path}: {os.path.getsize(path) / 1024:.0f} KiB, {len(data)} arrays")
    with open(os.path.join(GOLDEN, "reference_kats.json"), "w") as fh:
        json.dump(gen_reference_kats(ref), fh, indent=1)
    return 0


if __name__ == "__main__":
    sys.exit(main())
