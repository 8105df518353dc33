"""Backend pack (vkr_vk_pack_gpu_candidates) and encode outputs (visible rows,
indirect commands) of the HIP path against the CPU oracle, byte for byte: all
rows published, a publication boundary that omits rows (the visible lists then
hold packed indices), and a submesh outside its geometry."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

STAGES = 0x1 | 0x2 | 0x4 | 0x8 | 0x20 | 0x40  # transform, cull, transparent, candidates, pack, encode


def tables(sc, rng, drop_geometry=0.0, drop_material=0.0, stale_generation=0.0, bad_submesh=0):
    """Publication tables covering every handle the scene's rows name."""
    from vulkan_renderer_amd.scene import GEOMETRY_DT, MATERIAL_DT, SUBMESH_DT
    n_geom = int(sc.ident[:, 2].max())
    n_mat = int(sc.ident[:, 5].max())
    geom = np.zeros(n_geom, GEOMETRY_DT)
    geom["generation"] = 1
    geom["resolvable"] = rng.uniform(size=n_geom) >= drop_geometry
    geom["generation"][rng.uniform(size=n_geom) < stale_generation] = 7
    geom["first_index"] = rng.integers(0, 1 << 20, n_geom)
    geom["submesh_count"] = 4
    geom["submesh_offset"] = np.arange(n_geom) * 4
    if bad_submesh:
        geom["submesh_count"][:bad_submesh] = 2      # rows naming submesh 2 or 3 become invalid
    sub = np.zeros(n_geom * 4, SUBMESH_DT)
    sub["first_index"] = rng.integers(0, 1 << 16, n_geom * 4)
    sub["index_count"] = rng.integers(3, 3000, n_geom * 4)
    sub["vertex_offset"] = rng.integers(-1000, 100000, n_geom * 4)
    mat = np.zeros(n_mat, MATERIAL_DT)
    mat["generation"] = 1
    mat["live"] = rng.uniform(size=n_mat) >= drop_material
    mat["slot_index"] = rng.permutation(n_mat)
    return geom, sub, mat


def run_case(oracle, n, partition, scene_seed=6, cascades=2, spread=0.3, list_capacity=0, **drops):
    from helpers import clamp_to_list_capacity, hip_scene_for, view_set
    from oracle import pyoracle
    from vulkan_renderer_amd import synth
    rng = np.random.default_rng(n + partition + 7919 * (scene_seed - 6))
    sc = synth.flat_scene(n, seed=scene_seed, blend_fraction=0.2)
    sc.pos[:, :3] *= np.float32(spread)
    vs = synth.camera_views(cascades=cascades)
    planes, required = view_set(vs)
    views = planes.shape[0]
    geom, sub, mat = tables(sc, rng, **drops)

    _, world, _, _ = oracle.scene_update(sc.parent, pos=sc.pos, quat=sc.quat, scale=sc.scale,
                                         flags=np.full(n, 0x01, np.uint8))
    pay = oracle.build_world_payload(world, None, sc.ident, sc.matbits, sc.bounds, vs.view,
                                     vs.projection)
    ok, rows, inst, src, counts = oracle.pack_gpu_candidates(pay["gpu_candidates"], geom, sub, mat)
    # the reference culls the packed rows: same candidates, packed numbering
    packed_candidates = pay["gpu_candidates"][src]
    _, states, vis, _ = oracle.multiview_cull(packed_candidates, planes, required,
                                              partition_capacity=partition)

    if list_capacity:
        states, vis = clamp_to_list_capacity(states, vis, list_capacity)

    with hip_scene_for(sc, views, partition_capacity=partition, visible_capacity=list_capacity) as hs:
        hs.set_views(planes, required, vs.view)
        hs.set_publication(geom, sub, mat)
        hs.frame(STAGES)
        info = hs.frame_info()
        assert [info.packed_count, info.unpublished_geometry_count, info.unpublished_material_count,
                info.invalid_submesh_count] == counts.tolist()
        assert (info.invalid_submesh_count == 0) == ok
        k = info.packed_count
        got_rows, got_inst = hs.read_packed(k)
        assert got_rows.tobytes() == rows.tobytes(), "VkrGpuCandidateDrawRow rows"
        assert got_inst.tobytes() == inst.tobytes(), "packed VkrInstanceDataGPU rows"
        # the source rows themselves are untouched by the publication state
        assert hs.read_candidates().tobytes() == pay["gpu_candidates"].tobytes()
        for v, st in enumerate(info.views):
            assert (st.execution_ranges == states[v]["execution_ranges"]).all(), f"view {v} ranges"
            assert (st.bucket_counts == states[v]["bucket_counts"]).all()
            assert st.overflow_count == int(states[v]["overflow_count"])
            count = int(states[v]["visible_count"])
            assert st.visible_count == count
            assert (hs.read_visible(v, count) == vis[v, :count]).all(), f"view {v} packed indices"
            want = oracle.encode_visible_rows(rows, vis[v], states[v])
            assert hs.read_visible_rows(v, count).tobytes() == want.tobytes(), f"view {v} visible rows"
            if partition:
                cmds = oracle.encode_commands(rows, vis[v], states[v], v, partition, views)
                for b in range(4):
                    written = int(states[v]["execution_ranges"][b][1])
                    first = (v * 4 + b) * partition
                    assert (hs.read_commands(v, b, written).tobytes() ==
                            cmds[first: first + written].tobytes()), f"view {v} bucket {b} commands"
        # the transparent list does not go through the pack
        t = info.transparent_draw_count
        draws, _, _, _ = hs.read_transparent(t)
        assert draws.tobytes() == pay["draws"].tobytes()
        return info, counts


def test_everything_published(oracle):
    info, counts = run_case(oracle, 6000, 4096)
    assert counts[0] == 6000 and info.views[0].visible_count > 50


def test_publication_boundary_omits_rows(oracle, staged_mode):
    info, counts = run_case(oracle, 9000, 8192, drop_geometry=0.15, drop_material=0.1,
                            stale_generation=0.05)
    assert counts[1] > 100 and counts[2] > 100 and counts[0] < 9000


def test_partition_overflow_and_no_commands(oracle, staged_mode):
    info, _ = run_case(oracle, 5000, 64, drop_geometry=0.05)
    assert any(st.overflow_count for st in info.views)
    run_case(oracle, 3000, 0, drop_material=0.2)       # unlimited partitions: no command buffer


@pytest.mark.parametrize("list_capacity", [333, 1, 128, 1000])
def test_list_capacity_below_the_survivors(oracle, list_capacity, staged_mode):
    """A per-view list shorter than the survivors (and not a multiple of the encode
    block): the ranges, counts, visible rows and commands all end at the capacity
    and no view reads or writes its neighbour's stretch."""
    info, _ = run_case(oracle, 20000, 4096, cascades=3, list_capacity=list_capacity)
    assert max(st.visible_count for st in info.views) == list_capacity
    assert any(st.overflow_count for st in info.views)
    run_case(oracle, 6000, 0, cascades=1, list_capacity=list_capacity, drop_geometry=0.1)


def test_invalid_submesh_is_reported(oracle):
    info, counts = run_case(oracle, 4000, 4096, bad_submesh=40)
    assert counts[3] > 0 and info.invalid_submesh_count == counts[3]


@pytest.mark.parametrize("seed", range(int(__import__("os").environ.get("VKR_RANDOM_SEEDS", "8"))))
def test_random_publications(oracle, seed, staged_mode):
    """Random sizes, partition capacities, view counts and publication gaps."""
    rng = np.random.default_rng(31000 + seed)
    run_case(oracle, int(rng.choice([1, 64, 1025, 3000, 20000])), int(rng.choice([0, 16, 700, 1 << 15])),
             scene_seed=100 + seed, cascades=int(rng.integers(0, 5)), spread=float(rng.choice([0.1, 0.4, 1.0])),
             drop_geometry=float(rng.choice([0.0, 0.1, 0.9])), drop_material=float(rng.choice([0.0, 0.2])),
             stale_generation=float(rng.choice([0.0, 0.05])), bad_submesh=int(rng.choice([0, 0, 3])))


def test_stages_need_a_publication(oracle):
    from helpers import hip_scene_for, view_set
    from vulkan_renderer_amd import _native as N, synth
    sc = synth.flat_scene(100, seed=1)
    vs = synth.camera_views()
    planes, required = view_set(vs)
    with hip_scene_for(sc, 1) as hs:
        hs.set_views(planes, required, vs.view)
        with pytest.raises(N.VkrHipError) as e:
            hs.frame(0x7 | 0x20)
        assert e.value.code == N.ERR_STATE
