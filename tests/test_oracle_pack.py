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


# ---- candidate epoch (vkr_vk_candidate_epoch, vkr_vulkan_draws.c:9-44) ----------
FNV_STANDARD_BASIS = 0xcbf29ce484222325
# vkr_vulkan_draws.c:21 starts from 1469598103934665603, the standard basis
# 14695981039346656037 without its last digit; parity means following the reference
REFERENCE_BASIS = 1469598103934665603


def _fnv1a64(data: bytes, h: int = FNV_STANDARD_BASIS) -> int:
    for b in data:
        h = ((h ^ b) * 0x100000001b3) & 0xffffffffffffffff
    return h


def python_candidate_epoch(cand) -> int:
    """The epoch hash stated a second time, in plain Python over the row bytes:
    words 0-6 (mesh, geometry, submesh_index, material), 8-23 (instance.model),
    28-31 (local_bounding_sphere), 32-33 (state_bucket, flags) of each 144-byte
    row that has CAMERA_OPAQUE (0x2) in its flags, then the row count."""
    rows = np.ascontiguousarray(cand).view(np.uint8).reshape(-1, 144)
    h, occluders = REFERENCE_BASIS, 0
    for row in rows:
        flags = int(row[132:136].view(np.uint32)[0])
        if not flags & 0x2:
            continue
        occluders += 1
        h = _fnv1a64(row[0:28].tobytes() + row[32:96].tobytes() + row[112:136].tobytes(), h)
    return _fnv1a64(int(occluders).to_bytes(4, "little"), h)


def test_epoch_hash_known_answers(oracle):
    from oracle import pyoracle as P
    # published FNV-1a 64 vectors pin the byte step
    assert _fnv1a64(b"") == 0xcbf29ce484222325
    assert _fnv1a64(b"a") == 0xaf63dc4c8601ec8c
    assert _fnv1a64(b"foobar") == 0x85944171f73967e8
    # no candidates: only the four count bytes
    assert oracle.candidate_epoch(np.zeros(0, P.CANDIDATE_DT)) == _fnv1a64(b"\0\0\0\0", REFERENCE_BASIS)
    rng = np.random.default_rng(12)
    c = np.zeros(37, P.CANDIDATE_DT)
    raw = c.view(np.uint8).reshape(-1, 144)
    raw[:] = rng.integers(0, 256, raw.shape, dtype=np.uint8)   # padding bytes included
    c["flags"] = rng.integers(0, 8, 37)
    assert 0 < int((c["flags"] & 2 != 0).sum()) < 37
    assert oracle.candidate_epoch(c) == python_candidate_epoch(c)
    # a field that is not hashed does not change it, a hashed one does
    base = oracle.candidate_epoch(c)
    d = c.copy()
    d["object_id"] += 1
    assert oracle.candidate_epoch(d) == base
    first = int(np.flatnonzero(c["flags"] & 2)[0])
    d = c.copy()
    d["model"][first, 5] += np.float32(1.0)
    assert oracle.candidate_epoch(d) != base
