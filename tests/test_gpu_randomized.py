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


# VKR_RANDOM_SEEDS widens the sweep for a soak run; seeds from 12 on also draw
# probe views (up to 64 views in all)
@pytest.mark.parametrize("seed", range(int(os.environ.get("VKR_RANDOM_SEEDS", "16"))))
def test_random_scene_over_frames(oracle, seed):
    from helpers import compare_frame, hip_scene_for, view_set
    from vulkan_renderer_amd import synth
    rng = np.random.default_rng(1000 + seed)
    depth = int(rng.integers(1, 7))
    n = int(rng.choice([1, 63, 255, 256, 257, 1023, 1024, 1025, 3000, 9000, 20000]))
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
