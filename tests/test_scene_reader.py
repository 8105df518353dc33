"""vkrhip_scene_read_entities (host code behind the C ABI) against Python's json
on the reference's own scene files and on documents made up for the corner
cases: absent and null members, escapes, sizing call, errors."""
import ctypes as C
import glob
import json
import os

import numpy as np
import pytest

from vulkan_renderer_amd import _native as N
from vulkan_renderer_amd import scene_json as S

HERE = os.path.dirname(os.path.abspath(__file__))
NO_PARENT = 0xFFFFFFFF


def expected(doc):
    ents = doc["entities"]
    n = len(ents)
    parent = np.full(n, NO_PARENT, np.uint32)
    pos = np.zeros((n, 4), np.float32)
    quat = np.tile(np.array([0, 0, 0, 1], np.float32), (n, 1))
    scale = np.tile(np.array([1, 1, 1, 0], np.float32), (n, 1))
    kind = np.zeros(n, np.uint8)
    dims = np.ones((n, 3), np.float32)
    paths = [""] * n
    for i, e in enumerate(ents):
        if e.get("parent") is not None:
            parent[i] = e["parent"]
        t = e.get("transform") or {}
        pos[i, :3] = t.get("pos", [0, 0, 0])
        quat[i] = t.get("rot", [0, 0, 0, 1])
        scale[i, :3] = t.get("scale", [1, 1, 1])
        shape, mesh = e.get("shape"), e.get("mesh")
        if shape is not None:
            kind[i] = 1
            dims[i] = shape.get("dimensions", [1, 1, 1])
            paths[i] = (shape.get("material") or {}).get("path", "")
        elif mesh is not None:
            kind[i] = 2
    return parent, pos, quat, scale, kind, dims, paths


def check(text: bytes):
    got = S.read_entities(text)
    doc = json.loads(text)
    parent, pos, quat, scale, kind, dims, paths = expected(doc)
    assert got["version"] == doc.get("version", 0)
    assert (got["parent"] == parent).all()
    for name, want in (("pos", pos), ("quat", quat), ("scale", scale), ("dimensions", dims)):
        assert got[name].view(np.uint32).tobytes() == want.view(np.uint32).tobytes(), name
    assert (got["kind"] == kind).all()
    assert got["renderables"] == int((kind != 0).sum())
    for i, (offset, length) in enumerate(got["material_path"]):
        assert text[int(offset): int(offset) + int(length)].decode() == paths[i]
    return got


@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(HERE, "golden", "config1", "*.scene.json"))))
def test_reference_scene_files(path):
    with open(path, "rb") as fh:
        got = check(fh.read())
    assert got["parent"].shape[0] >= 8


def test_corner_cases():
    doc = {"version": 2, "name": "corner \u00e9", "entities": [
        {"name": "root"},
        {"name": "null members", "parent": None, "transform": None, "shape": None, "mesh": None},
        {"name": "child", "parent": 0, "transform": {"pos": [1.5, -2e-3, 3], "extra": {"deep": [1, [2, {"x": None}]]}}},
        {"name": "rot and scale", "parent": 2, "transform": {"rot": [0.1, 0.2, 0.3, 0.9], "scale": [2, 3, 4]}},
        {"name": "shape without material", "parent": 1, "shape": {"type": "cube", "dimensions": [2, 4, 0.5]}},
        {"name": "shape, null material", "shape": {"type": "cube", "material": None}},
        {"name": "shape and mesh", "shape": {"material": {"name": "m", "path": "a/b/c.mt"}}, "mesh": {"path": "x.obj"}},
        {"name": "mesh", "parent": 6, "mesh": {"path": "models/falcon.obj"}},
        {"name": "numbers", "transform": {"pos": [1e10, 0.1, -0.0], "scale": [1E-3, 16777217, 0.30000000000000004]}},
    ], "environment": {"skybox": [1, 2, 3]}}
    got = check(json.dumps(doc).encode())
    assert got["kind"].tolist() == [0, 0, 0, 0, 1, 1, 1, 2, 0]
    # pretty-printed, with the members in another order
    check(json.dumps(doc, indent=3, sort_keys=True).encode())
    # no entities at all
    assert S.read_entities(b'{"version": 2, "entities": []}')["parent"].shape[0] == 0
    assert S.read_entities(b'{"version": 2}')["parent"].shape[0] == 0


def test_sizing_call_and_errors():
    lib = N.load()
    text = json.dumps({"version": 2, "entities": [{}, {"parent": 0}, {"parent": 1, "mesh": {}}]}).encode()
    info = N.SceneEntities()
    rc = lib.vkrhip_scene_read_entities(text, len(text), 0, C.byref(info), None, None, None, None, None, None, None)
    assert rc == N.ERR_CAPACITY and (info.version, info.entity_count, info.renderable_count) == (2, 3, 1)
    parent = np.zeros(3, np.uint32)
    rc = lib.vkrhip_scene_read_entities(text, len(text), 3, C.byref(info), N.ptr(parent), None, None, None, None,
                                        None, None)
    assert rc == 0 and parent.tolist() == [NO_PARENT, 0, 1]
    assert lib.vkrhip_scene_read_entities(None, 0, 0, C.byref(info), None, None, None, None, None, None, None) \
        == N.ERR_ARG
    for bad in (b"", b"[]", b'{"entities": [{"parent": 5}]}', b'{"entities": [{"parent": 0}]}',
                b'{"entities": [{"parent": 1}, {"parent": 0}]}', b'{"entities": [{"parent": -1}]}',
                b'{"entities": [{"parent": 0.5}, {}]}', b'{"entities": [{"transform": {"pos": [1, 2]}}]}',
                b'{"entities": [{}] trailing', b'{"entities": [{"shape": {"material": {"path": 3}}}]}',
                b'{"version": -1, "entities": []}'):
        with pytest.raises(ValueError):
            S.read_entities(bad)
