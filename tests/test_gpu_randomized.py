"""Seeded random scenes run for several frames through the whole HIP path and
through the CPU oracle: random sizes (tile edges included), depths, view
counts, partition capacities, and per frame a random set of moved nodes (TRS
changes, DIRTY_WORLD marks, or nothing). Every output is compared bit for bit
after every frame."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

DIRTY_LOCAL, DIRTY_WORLD = 0x01, 0x02


def _frame_reference(oracle, sc, state, planes, required, vs, partition_capacity):
    """Oracle frame from the carried (local, world, flags) state."""
    local, world, flags, dirty = oracle.scene_update(
        sc.parent, local=state["local"], world=state["world"], flags=state["flags"],
        pos=state["pos"], quat=sc.quat, scale=sc.scale)
    pay = oracle.build_world_payload(world, sc.node, sc.ident, sc.matbits, sc.bounds, vs.view,
                                     vs.projection)
    cls, states, vis, ties = oracle.multiview_cull(pay["gpu_candidates"], planes, required,
                                                   partition_capacity=partition_capacity)
    state.update(local=local, world=world, flags=flags)
    return dict(local=local, world=world, flags=flags, dirty=dirty, payload=pay, cls=cls, states=states,
                visible=vis, ties=ties)


# VKR_RANDOM_SEEDS widens the sweep for a soak run and VKR_RANDOM_SCALE multiplies the
# scene sizes; seeds from 12 on also draw probe views (up to 64 views in all)
@pytest.mark.parametrize("seed", range(int(os.environ.get("VKR_RANDOM_SEEDS", "16"))))
def test_random_scene_over_frames(oracle, seed, staged_mode, tail_mode):
    from helpers import compare_frame, hip_scene_for, view_set
    from vulkan_renderer_amd import synth
    rng = np.random.default_rng(1000 + seed)
    depth = int(rng.integers(1, 7))
    n = int(rng.choice([1, 63, 255, 256, 257, 1023, 1024, 1025, 3000, 9000, 20000])) * int(os.environ.get("VKR_RANDOM_SCALE", "1"))
    if depth == 1:
        sc = synth.flat_scene(n, seed=seed, blend_fraction=0.25, transmissive_fraction=0.05,
                              no_bounds_fraction=0.03)
    else:
        sc = synth.deep_scene(max(n, 40), depth=depth, seed=seed, fanout=int(rng.integers(2, 7)),
                              roots=int(rng.integers(1, 5)), blend_fraction=0.25,
                              transmissive_fraction=0.05, no_bounds_fraction=0.03)
    sc.pos[:, :3] *= np.float32(rng.choice([0.2, 0.5, 1.0]))   # denser scenes: more survivors
    n = sc.n
    cascades = int(rng.integers(0, 5))
    vs = synth.camera_views(cascades=cascades, far=float(rng.choice([300.0, 900.0])))
    probes = int(rng.choice([0, 0, 7, 20, 59])) if seed >= 12 else 0
    planes, required = view_set(vs, extra_vp=synth.ring_views(probes)[0] if probes else None)
    capacity = int(rng.choice([0, 0, 64, 1000]))
    state = dict(local=None, world=None, flags=np.full(n, DIRTY_LOCAL, np.uint8), pos=sc.pos.copy())
    with hip_scene_for(sc, planes.shape[0], partition_capacity=capacity) as hs:
        hs.set_views(planes, required, vs.view)
        for frame in range(4):
            if frame:
                kind = int(rng.integers(0, 4))
                flags = state["flags"].copy()      # WORLD_UPDATED of the last frame is still set
                if kind == 0:                        # some TRS change
                    moved = np.flatnonzero(rng.uniform(size=n) < 0.03)
                    state["pos"][moved, :3] += rng.uniform(-2, 2, (moved.shape[0], 3)).astype(np.float32)
                    for i in moved:
                        hs.upload_trs(state["pos"][i:i + 1], sc.quat[i:i + 1], sc.scale[i:i + 1], first=int(i))
                    flags[moved] |= DIRTY_LOCAL
                elif kind == 1:                      # world marks on a random subset
                    marked = rng.uniform(size=n) < 0.1
                    add = np.where(marked, DIRTY_WORLD, 0).astype(np.uint8)
                    got = hs.read_node_flags()
                    hs.upload_node_flags(got | add)
                    flags |= add
                elif kind == 2:                      # everything moved
                    hs.mark_all(DIRTY_WORLD)
                    flags |= DIRTY_WORLD
                state["flags"] = flags
            ref = _frame_reference(oracle, sc, state, planes, required, vs, capacity)
            hs.frame()
            compare_frame(hs, ref, sc, partition_capacity=capacity)


STAGE_TRANSFORM, STAGE_CULL, STAGE_TRANSPARENT, STAGE_CANDIDATES, STAGE_BOUNDS = 1, 2, 4, 8, 16


@pytest.mark.parametrize("seed", range(int(os.environ.get("VKR_RANDOM_SEEDS", "16"))))
def test_random_feature_mix(oracle, seed, staged_mode):
    """Random combinations of what selects a kernel variant: identity or arbitrary
    candidate->node mapping, tie counting on or off, world bounds written or not,
    the candidate rows emitted or not, one frame call or the stages split over two,
    a source index base, small and large transparent counts."""
    from conftest import assert_bits_equal
    from helpers import compare_frame, hip_scene_for, oracle_frame, view_set
    from vulkan_renderer_amd import synth
    rng = np.random.default_rng(50000 + seed)
    depth = int(rng.integers(1, 6))
    n = int(rng.choice([200, 1024, 1500, 5000, 12000, 40000])) * int(os.environ.get("VKR_RANDOM_SCALE", "1"))
    blend = float(rng.choice([0.1, 0.5, 1.0]))
    if depth == 1:
        sc = synth.flat_scene(n, seed=seed, blend_fraction=blend, no_bounds_fraction=0.02)
    else:
        sc = synth.deep_scene(n, depth=depth, seed=seed, fanout=int(rng.integers(2, 7)),
                              roots=int(rng.integers(1, 5)), blend_fraction=blend, no_bounds_fraction=0.02)
    sc.pos[:, :3] *= np.float32(rng.choice([0.1, 0.5, 1.0]))
    if rng.uniform() < 0.4:   # several rows per transform, arbitrary order
        n_cand = int(rng.integers(1, 3 * sc.n))
        sc.node = rng.integers(0, sc.n, n_cand).astype(np.uint32)
        sc.ident, sc.matbits, sc.bounds = synth.make_candidates(
            seed + 7, np.arange(n_cand, dtype=np.uint64), blend_fraction=blend)
    n_cand = sc.ident.shape[0]
    tie_eps = float(rng.choice([0.0, 0.0, 1e-5, 1e-2]))
    base = int(rng.choice([0, 0, 1 << 20]))
    vs = synth.camera_views(cascades=int(rng.integers(0, 5)), far=float(rng.choice([300.0, 2000.0])),
                            fov_deg=float(rng.choice([60.0, 110.0])))
    planes, required = view_set(vs)
    ref = oracle_frame(oracle, sc, planes, required, vs.view, vs.projection, tie_eps=tie_eps,
                       index_base=base)
    extra = (STAGE_CANDIDATES if rng.uniform() < 0.5 else 0) | (STAGE_BOUNDS if rng.uniform() < 0.5 else 0)
    split = rng.uniform() < 0.3
    with hip_scene_for(sc, planes.shape[0], tie_epsilon=tie_eps, source_index_base=base) as hs:
        hs.set_views(planes, required, vs.view)
        if split:
            hs.frame(STAGE_TRANSFORM)
            assert hs.frame_info().nodes_updated == sc.n
            hs.frame(STAGE_CULL | STAGE_TRANSPARENT | extra)
            compare_frame(hs, ref, sc, check_transform=False)
            assert_bits_equal(hs.read_world(), ref["world"], "world matrices")
        else:
            hs.frame(STAGE_TRANSFORM | STAGE_CULL | STAGE_TRANSPARENT | extra)
            compare_frame(hs, ref, sc)
        if extra & STAGE_CANDIDATES:
            assert hs.read_candidates().tobytes() == ref["payload"]["gpu_candidates"].tobytes()
        if extra & STAGE_BOUNDS:
            wb = hs.read_world_bounds()
            node = sc.node if sc.node is not None else np.arange(n_cand, dtype=np.uint32)
            pick = rng.integers(0, n_cand, 40)
            exp = np.stack([oracle.instance_world_bounds(ref["world"][node[i]], sc.bounds[i, :3],
                                                         ref["payload"]["gpu_candidates"][i]["sphere"][3])
                            for i in pick])
            assert_bits_equal(wb[pick], exp, "instance world bounds")
        # a second frame with everything marked moved gives the same lists
        hs.mark_all(DIRTY_WORLD)
        hs.frame(STAGE_TRANSFORM | STAGE_CULL | STAGE_TRANSPARENT)
        compare_frame(hs, ref, sc, check_transform=False)
