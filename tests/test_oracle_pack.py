"""Hand-checked cases of the oracle's backend pack and encode restatement
(vkr_vk_pack_gpu_candidates, vkr_vulkan_draws.c:321-428; vk_gpu_draw_encode,
deferred.slang:379-420). The reference's Vulkan backend cannot be built here, so
these follow the source text; parity of this step is unpinned beyond that."""
import numpy as np


def test_pack_resolution_rules(oracle):
    from oracle import pyoracle as P
    c = np.zeros(6, P.CANDIDATE_DT)
    c["geometry"] = [[1, 1], [2, 1], [3, 1], [1, 1], [0, 1], [2, 9]]
    c["material"] = [[1, 1], [1, 1], [2, 1], [2, 2], [1, 1], [1, 1]]
    c["submesh_index"] = [0, 1, 0, 0, 0, 0]
    c["state_bucket"] = [0, 1, 2, 3, 0, 1]
    c["flags"] = [7, 6, 5, 4, 3, 2]
    c["object_id"] = 10 + np.arange(6)
    g = np.zeros(3, P.GEOMETRY_DT)
    g["generation"] = 1
    g["resolvable"] = [1, 1, 0]          # geometry 3 still uploading
    g["submesh_count"] = [1, 2, 1]
    g["submesh_offset"] = [0, 1, 3]
    g["first_index"] = [100, 200, 300]
    sm = np.zeros(4, P.SUBMESH_DT)
    sm["index_count"] = [6, 12, 18, 24]
    sm["first_index"] = [0, 0, 36, 0]
    sm["vertex_offset"] = [0, 5, -7, 0]
    m = np.zeros(2, P.MATERIAL_DT)
    m["generation"] = 1
    m["live"] = 1
    m["slot_index"] = [7, 9]
    ok, rows, inst, src, counts = oracle.pack_gpu_candidates(c, g, sm, m)
    # row 2: geometry not resolvable; row 3: material generation differs; row 4: null
    # geometry handle; row 5: geometry generation differs
    assert ok and counts.tolist() == [2, 3, 1, 0]
    assert src.tolist() == [0, 1]
    assert rows["geometry_index"].tolist() == [0, 1]
    assert rows["material_index"].tolist() == [7, 7]
    assert rows["instance_index"].tolist() == [0, 1]
    assert rows["first_index"].tolist() == [100, 236]
    assert rows["index_count"].tolist() == [6, 18]
    assert rows["vertex_offset"].tolist() == [0, -7]
    assert rows["state_bucket"].tolist() == [0, 1] and rows["flags"].tolist() == [7, 6]
    assert inst["object_id"].tolist() == [10, 11]
    # a submesh outside its geometry rejects the frame
    c["submesh_index"][0] = 1
    ok, _, _, _, counts = oracle.pack_gpu_candidates(c, g, sm, m)
    assert not ok and counts[3] == 1


def test_encode_rows_and_commands(oracle):
    from oracle import pyoracle as P
    rows = np.zeros(5, P.GPU_ROW_DT)
    rows["geometry_index"] = np.arange(5)
    rows["index_count"] = 3 * (1 + np.arange(5))
    rows["first_index"] = 100 * np.arange(5)
    rows["vertex_offset"] = -np.arange(5)
    rows["state_bucket"] = [0, 2, 0, 1, 2]
    state = np.zeros(1, P.STATE_DT)[0]
    state["execution_ranges"] = [[0, 2], [2, 1], [3, 2], [5, 0]]
    state["visible_count"] = 5
    visible = np.array([0, 2, 3, 1, 4], np.uint32)
    out = oracle.encode_visible_rows(rows, visible, state)
    assert out["geometry_index"].tolist() == [0, 2, 3, 1, 4]
    assert out["state_bucket"].tolist() == [0, 0, 1, 2, 2]
    cmds = oracle.encode_commands(rows, visible, state, 1, 8, 2)
    first = (1 * 4 + 2) * 8
    assert cmds["index_count"][first: first + 2].tolist() == [6, 15]
    assert cmds["first_index"][first: first + 2].tolist() == [100, 400]
    assert cmds["vertex_offset"][first: first + 2].tolist() == [-1, -4]
    assert (cmds["instance_count"][first: first + 2] == 1).all()
    assert cmds["index_count"][:8].tolist() == [0] * 8      # view 0 untouched
