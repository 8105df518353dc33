"""The glTF node reader (vkrhip_gltf_read_nodes: "nodes" and "scenes" of a
document -> per-node arrays) against Python's json module as an independent
parser, on the two glTF files the reference's tests hold
(tests/golden/gltf/{punctual_lights,specgloss_factor_parity}.gltf) and on
nested.gltf; then the scene those arrays describe through the oracle and, on a
GPU, through the HIP path."""
import json
import os

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
FILES = ["punctual_lights.gltf", "specgloss_factor_parity.gltf", "nested.gltf"]
NO_PARENT = 0xFFFFFFFF


def path_of(name):
    return os.path.join(HERE, "golden", "gltf", name)


def expected_nodes(doc, scene=None):
    """What the arrays must hold, from json.load (numbers -> float32 once)."""
    nodes = doc.get("nodes", [])
    n = len(nodes)
    parent = np.full(n, NO_PARENT, np.uint32)
    pos, scale = np.zeros((n, 4), np.float32), np.zeros((n, 4), np.float32)
    quat = np.zeros((n, 4), np.float32)
    mesh = np.full(n, -1, np.int32)
    has_matrix = np.zeros(n, bool)
    matrix = np.tile(np.eye(4, dtype=np.float32).reshape(16), (n, 1))
    for i, nd in enumerate(nodes):
        for c in nd.get("children", []):
            parent[c] = i
        pos[i, :3] = np.array(nd.get("translation", [0, 0, 0]), np.float64).astype(np.float32)
        quat[i] = np.array(nd.get("rotation", [0, 0, 0, 1]), np.float64).astype(np.float32)
        scale[i, :3] = np.array(nd.get("scale", [1, 1, 1]), np.float64).astype(np.float32)
        mesh[i] = nd.get("mesh", -1)
        if "matrix" in nd:
            has_matrix[i] = True
            matrix[i] = np.array(nd["matrix"], np.float64).astype(np.float32)
    reached = np.ones(n, bool)
    chosen = 0
    if doc.get("scenes"):
        chosen = scene if scene is not None else doc.get("scene", 0)
        reached[:] = False
        stack = list(doc["scenes"][chosen].get("nodes", []))
        while stack:
            k = stack.pop()
            reached[k] = True
            stack.extend(nodes[k].get("children", []))
    return dict(parent=parent, pos=pos, quat=quat, scale=scale, mesh=mesh, has_matrix=has_matrix,
                matrix=matrix, in_scene=reached, scene=chosen)


@pytest.mark.parametrize("name", FILES)
def test_reader_against_json_module(name):
    from vulkan_renderer_amd import gltf
    text = open(path_of(name), "rb").read()
    doc = json.loads(text)
    scenes = [None] + list(range(len(doc.get("scenes", []))))
    for scene in scenes:
        got = gltf.read_nodes(text, scene)
        want = expected_nodes(doc, scene)
        assert got.scene == want["scene"] and got.scene_count == len(doc.get("scenes", []))
        for key in ("parent", "mesh", "has_matrix", "in_scene"):
            assert (getattr(got, key) == want[key]).all(), key
        for key in ("pos", "quat", "scale", "matrix"):
            assert getattr(got, key).tobytes() == want[key].tobytes(), key
    if name == "nested.gltf":
        g = gltf.read_nodes(text)
        assert g.in_scene.tolist() == [True] * 7 + [False, False]
        assert gltf.read_nodes(text, 1).in_scene.tolist() == [False] * 7 + [True, False]


def test_reference_fixture_contents():
    """What the reference's two files hold (so a changed copy would be noticed)."""
    from vulkan_renderer_amd import gltf
    lights = gltf.read_nodes(open(path_of("punctual_lights.gltf"), "rb").read())
    assert lights.parent.tolist() == [NO_PARENT] * 3 and lights.mesh.tolist() == [-1, -1, -1]
    assert lights.pos[:, :3].tolist() == [[1.0, 2.0, 3.0], [-2.0, 1.0, 0.5], [0.0, 0.0, 0.0]]
    spec = gltf.read_nodes(open(path_of("specgloss_factor_parity.gltf"), "rb").read())
    assert spec.mesh.tolist() == [0, 1] and spec.in_scene.all()
    assert spec.pos[:, 0].tolist() == [np.float32(-1.15), np.float32(1.15)]


@pytest.mark.parametrize("text", [
    b"", b"{", b'{"nodes": [{"children": [1]}]}', b'{"nodes": [{"children": [0]}]}',
    b'{"nodes": [{"children": [1]}, {"children": [0]}]}',                      # a cycle
    b'{"nodes": [{"children": [2]}, {"children": [2]}, {}]}',                  # two parents
    b'{"nodes": [{"translation": [1, 2]}]}', b'{"nodes": [{"mesh": -1}]}',
    b'{"nodes": [{}], "scenes": [{"nodes": [0]}], "scene": 3}',
    b'{"nodes": [{"children": [1]}, {}], "scenes": [{"nodes": [1]}]}',         # a scene lists a non-root
    b'{"nodes": [{"name": "bad \\q escape"}]}', b'{"nodes": [{}]} trailing', b'{"nodes": [{"scale": [01, 1, 1]}]}',
    b"[" * 200 + b"]" * 200,
])
def test_malformed_documents_are_refused(text):
    from vulkan_renderer_amd import _native as N
    from vulkan_renderer_amd import gltf
    with pytest.raises(N.VkrHipError) as e:
        gltf.read_nodes(text)
    assert e.value.code == N.ERR_ARG


def test_glb_container(tmp_path):
    import struct
    from vulkan_renderer_amd import gltf
    text = open(path_of("nested.gltf"), "rb").read()
    text += b" " * (-len(text) % 4)
    blob = struct.pack("<4sII", b"glTF", 2, 12 + 8 + len(text)) + struct.pack("<I4s", len(text), b"JSON") + text
    p = tmp_path / "nested.glb"
    p.write_bytes(blob)
    a, b = gltf.load_nodes(str(p)), gltf.load_nodes(path_of("nested.gltf"))
    assert (a.file_index == b.file_index).all() and a.pos.tobytes() == b.pos.tobytes()


def _scene_of(name, oracle):
    """The scene's arrays for the frame: level-ordered TRS nodes; one row per
    primitive of every mesh node (as the reference draws one submesh per primitive),
    with the bounds of its POSITION accessor where the document gives them - a
    mesh without primitives, or a primitive without accessor bounds, gets a unit
    cube; nodes given as a matrix get that matrix as their local."""
    from vulkan_renderer_amd import gltf, synth
    gs = gltf.load_nodes(path_of(name))
    mb = gltf.read_mesh_bounds(gltf.document_text(path_of(name)))
    n = gs.parent.shape[0]
    unit = np.zeros(12, np.float32)
    unit[4:7], unit[8:11] = -0.5, 0.5
    rows, row_bounds = [], []
    for node in np.flatnonzero(gs.mesh >= 0):
        prims = np.flatnonzero(mb.mesh == gs.mesh[node])
        for p in prims:
            rows.append(node)
            row_bounds.append(mb.bounds[p] if mb.has_bounds[p] else unit)
        if prims.shape[0] == 0:
            rows.append(node)
            row_bounds.append(unit)
    rows = np.asarray(rows, np.uint32)
    ident, matbits, bounds = synth.make_candidates(3, np.arange(rows.shape[0], dtype=np.uint64))
    if rows.shape[0]:
        bounds[:] = np.stack(row_bounds)
    local = np.stack([oracle.local_from_trs(gs.pos[i], gs.quat[i], gs.scale[i]) for i in range(n)]) \
        if n else np.zeros((0, 16), np.float32)
    local[gs.has_matrix] = gs.matrix[gs.has_matrix]
    return gs, rows, ident, matbits, bounds, local


def test_mesh_bounds_of_the_reference_fixture():
    """specgloss_factor_parity.gltf (the reference's own test file): both meshes draw
    one primitive whose POSITION accessor spans [-1,-1,0]..[1,1,0] - and that is what
    the vertex positions in the file's embedded buffer span (the reference takes the
    minimum and maximum over the vertices, mesh_loader.c:1159-1179)."""
    import base64
    import json
    from vulkan_renderer_amd import gltf
    text = gltf.document_text(path_of("specgloss_factor_parity.gltf"))
    mb = gltf.read_mesh_bounds(text)
    assert mb.mesh_count == 2 and mb.mesh.tolist() == [0, 1] and mb.primitive.tolist() == [0, 0]
    assert mb.material.tolist() == [0, 1] and mb.has_bounds.all()
    for p in range(2):
        assert mb.bounds[p, 0:3].tolist() == [0.0, 0.0, 0.0]       # center
        assert mb.bounds[p, 4:7].tolist() == [-1.0, -1.0, 0.0]     # min_extents
        assert mb.bounds[p, 8:11].tolist() == [1.0, 1.0, 0.0]      # max_extents
    doc = json.loads(text)
    blob = base64.b64decode(doc["buffers"][0]["uri"].split(",", 1)[1])
    view = doc["bufferViews"][doc["accessors"][0]["bufferView"]]
    positions = np.frombuffer(blob, np.float32, count=4 * 3, offset=view["byteOffset"]).reshape(4, 3)
    assert positions.min(axis=0).tolist() == mb.bounds[0, 4:7].tolist()
    assert positions.max(axis=0).tolist() == mb.bounds[0, 8:11].tolist()


def test_mesh_bounds_reader_cases():
    """Meshes with two, one and no primitives; an accessor without min / max; a
    scalar accessor's one-component min / max is not a bound; errors."""
    from vulkan_renderer_amd import _native as N, gltf
    mb = gltf.read_mesh_bounds(gltf.document_text(path_of("nested.gltf")))
    assert mb.mesh_count == 3 and mb.mesh.tolist() == [0, 0, 1] and mb.primitive.tolist() == [0, 1, 0]
    assert mb.material.tolist() == [3, -1, 0] and mb.has_bounds.tolist() == [True, True, False]
    f = np.float32
    assert mb.bounds[0, 4:7].tolist() == [f(-0.5), f(-1.25), f(0.125)] and mb.bounds[0, 8:11].tolist() == [f(2.5), f(0.75), f(3)]
    assert mb.bounds[0, 0:3].tolist() == [(f(-0.5) + f(2.5)) * f(0.5), (f(-1.25) + f(0.75)) * f(0.5),
                                          (f(0.125) + f(3)) * f(0.5)]
    assert mb.bounds[1, 4:7].tolist() == [f(-0.1), f(-0.2), f(-0.3)] and mb.bounds[1, 8:11].tolist() == [f(0.1), f(0.2), f(0.3)]
    assert not mb.bounds[2].any() and not mb.bounds[:, 3].any()
    assert gltf.read_mesh_bounds(gltf.document_text(path_of("punctual_lights.gltf"))).mesh.shape[0] == 0
    for bad in (b'{"meshes": [{"primitives": [{"attributes": {"POSITION": 4}}]}], "accessors": []}',
                b'{"meshes": [', b'[]'):
        with pytest.raises(N.VkrHipError) as e:
            gltf.read_mesh_bounds(bad)
        assert e.value.code == N.ERR_ARG


@pytest.mark.parametrize("name", FILES)
def test_oracle_runs_the_gltf_scenes(oracle, name):
    gs, rows, ident, matbits, bounds, local = _scene_of(name, oracle)
    n = gs.parent.shape[0]
    assert int(gs.level_offsets[-1]) == n
    _, world, flags, dirty = oracle.scene_update(gs.parent, local=local, flags=np.full(n, 0x02, np.uint8))
    assert dirty.shape[0] == n
    if name == "nested.gltf":
        assert n == 7 and rows.shape[0] == 5 and gs.level_offsets.tolist() == [0, 2, 5, 6, 7]
        # node 4 sits under 3 under 1 under 0: its world position, by hand in float64
        k = int(np.flatnonzero(gs.file_index == 4)[0])
        r = np.array([[0, 0, 1], [0, 1, 0], [-1, 0, 0]], np.float64)   # 90 degrees about +Y
        p = np.array([1.5, -2.25, 3.0]) + r @ (2.0 * np.array([-1e-3, 1e2, 0.000123456789]))
        assert np.allclose(world[k, 12:15], p, rtol=1e-6, atol=1e-5)
    else:
        assert (gs.parent == NO_PARENT).all()
        assert world.tobytes() == local.tobytes()


@pytest.mark.gpu
@pytest.mark.parametrize("name", FILES)
def test_hip_path_matches_oracle_on_the_gltf_scenes(oracle, name):
    from helpers import compare_frame, oracle_frame, view_set
    from vulkan_renderer_amd import HipScene, synth
    gs, rows, ident, matbits, bounds, local = _scene_of(name, oracle)
    n = gs.parent.shape[0]
    vs = synth.camera_views(cascades=2, far=300.0, eye=(0.0, 2.0, 14.0), target=(0.0, 0.0, 0.0))
    planes, required = view_set(vs)

    class Shape:
        pass

    sc = Shape()
    sc.n, sc.parent, sc.node, sc.ident, sc.matbits, sc.bounds = n, gs.parent, rows, ident, matbits, bounds
    _, world, _, _ = oracle.scene_update(gs.parent, local=local, flags=np.full(n, 0x02, np.uint8))
    ref = oracle_frame(oracle, sc, planes, required, vs.view, vs.projection, world=world)
    with HipScene(max_nodes=n, max_candidates=max(rows.shape[0], 1), max_views=planes.shape[0]) as hs:
        hs.set_hierarchy(gs.parent, gs.level_offsets)
        hs.upload_locals(local)                      # (matrix nodes have no TRS)
        hs.set_candidates(rows, ident, matbits, bounds)
        hs.set_views(planes, required, vs.view)
        hs.mark_all(0x02)
        hs.frame()
        compare_frame(hs, ref, sc, check_transform=False)
        assert hs.read_world().tobytes() == world.tobytes()
        if rows.shape[0]:
            assert int(ref["states"][0]["visible_count"]) >= 1, "the camera sees the scene"
