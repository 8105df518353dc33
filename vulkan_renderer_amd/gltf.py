"""glTF 2.0 node hierarchies as scene arrays: a thin wrapper over the library's
reader (vkrhip_gltf_read_nodes, csrc/vkrhip_gltf.cpp), which takes "nodes" and
"scenes" of a document - what the reference obtains through cgltf
(vendor/cgltf.h) - and returns per-node parents, TRS / matrices and mesh
indices. `load_nodes` keeps the nodes the chosen scene reaches, renumbers them
in file order and levelizes the result for vkrhip_set_hierarchy.
"""
from __future__ import annotations

import ctypes as C
import struct
from dataclasses import dataclass

import numpy as np

from . import _native as N

NO_PARENT = 0xFFFFFFFF


def document_text(path: str) -> bytes:
    """The JSON text of a .gltf file, or the JSON chunk of a .glb container."""
    data = open(path, "rb").read()
    if data[:4] == b"glTF":
        _, _, length = struct.unpack_from("<4sII", data, 0)
        chunk_length, chunk_type = struct.unpack_from("<I4s", data, 12)
        if chunk_type != b"JSON" or 20 + chunk_length > length:
            raise ValueError("not a glTF binary container with a leading JSON chunk")
        return data[20: 20 + chunk_length]
    return data


@dataclass
class GltfNodes:
    """Per-node arrays in FILE order (every node of the document)."""
    parent: np.ndarray      # (n,) u32, NO_PARENT for roots
    in_scene: np.ndarray    # (n,) bool
    pos: np.ndarray         # (n,4) f32
    quat: np.ndarray        # (n,4) f32 x,y,z,w
    scale: np.ndarray       # (n,4) f32
    mesh: np.ndarray        # (n,) i32, -1 = none
    has_matrix: np.ndarray  # (n,) bool
    matrix: np.ndarray      # (n,16) f32 column-major
    scene: int
    scene_count: int


def read_nodes(text: bytes, scene: int | None = None) -> GltfNodes:
    lib = N.load()
    info = N.GltfNodes()
    want = NO_PARENT if scene is None else scene
    rc = lib.vkrhip_gltf_read_nodes(text, len(text), want, 0, C.byref(info), *([None] * 8))
    if rc not in (N.OK, N.ERR_CAPACITY):
        raise N.VkrHipError(rc, "vkrhip_gltf_read_nodes: not a glTF document this reader accepts")
    n = info.node_count
    parent = np.zeros(n, np.uint32)
    in_scene, has_matrix = np.zeros(n, np.uint8), np.zeros(n, np.uint8)
    pos, quat, scale = (np.zeros((n, 4), np.float32) for _ in range(3))
    mesh = np.zeros(n, np.int32)
    matrix = np.zeros((n, 16), np.float32)
    rc = lib.vkrhip_gltf_read_nodes(text, len(text), want, n, C.byref(info), N.ptr(parent), N.ptr(in_scene),
                                    N.ptr(pos), N.ptr(quat), N.ptr(scale), N.ptr(mesh), N.ptr(has_matrix),
                                    N.ptr(matrix))
    if rc != N.OK:
        raise N.VkrHipError(rc, "vkrhip_gltf_read_nodes")
    return GltfNodes(parent, in_scene.astype(bool), pos, quat, scale, mesh, has_matrix.astype(bool), matrix,
                     int(info.scene), int(info.scene_count))


@dataclass
class GltfMeshBounds:
    """Per-primitive arrays in FILE order (mesh after mesh, primitive after primitive)."""
    mesh: np.ndarray        # (p,) u32 index into "meshes"
    primitive: np.ndarray   # (p,) u32 index into that mesh's primitives
    material: np.ndarray    # (p,) i32, -1 = none
    has_bounds: np.ndarray  # (p,) bool: the POSITION accessor carries min and max
    bounds: np.ndarray      # (p,12) f32: center, min_extents, max_extents as padded Vec3
    mesh_count: int


def read_mesh_bounds(text: bytes) -> GltfMeshBounds:
    """vkrhip_gltf_read_mesh_bounds: the submesh bounds the reference derives from a
    primitive's vertex positions, taken from the POSITION accessor's min / max."""
    lib = N.load()
    info = N.GltfMeshes()
    rc = lib.vkrhip_gltf_read_mesh_bounds(text, len(text), 0, C.byref(info), *([None] * 5))
    if rc not in (N.OK, N.ERR_CAPACITY):
        raise N.VkrHipError(rc, "vkrhip_gltf_read_mesh_bounds: not a glTF document this reader accepts")
    p = info.primitive_count
    mesh, primitive = np.zeros(p, np.uint32), np.zeros(p, np.uint32)
    material = np.zeros(p, np.int32)
    has_bounds = np.zeros(p, np.uint8)
    bounds = np.zeros((p, 12), np.float32)
    rc = lib.vkrhip_gltf_read_mesh_bounds(text, len(text), p, C.byref(info), N.ptr(mesh), N.ptr(primitive),
                                          N.ptr(material), N.ptr(has_bounds), N.ptr(bounds))
    if rc != N.OK:
        raise N.VkrHipError(rc, "vkrhip_gltf_read_mesh_bounds")
    return GltfMeshBounds(mesh, primitive, material, has_bounds.astype(bool), bounds, int(info.mesh_count))


@dataclass
class GltfScene:
    """The chosen scene's nodes, level-ordered for vkrhip_set_hierarchy."""
    file_index: np.ndarray     # (m,) glTF node index of every scene node, in level order
    parent: np.ndarray         # (m,) u32 level-ordered parents
    level_offsets: np.ndarray  # (levels + 1,) u64
    pos: np.ndarray
    quat: np.ndarray
    scale: np.ndarray
    mesh: np.ndarray           # (m,) i32
    has_matrix: np.ndarray
    matrix: np.ndarray


def load_nodes(path: str, scene: int | None = None) -> GltfScene:
    from .scene import levelize
    g = read_nodes(document_text(path), scene)
    keep = np.flatnonzero(g.in_scene)
    new_id = np.full(g.parent.shape[0], NO_PARENT, np.uint32)
    new_id[keep] = np.arange(keep.shape[0], dtype=np.uint32)
    par = g.parent[keep]
    compact = np.where(par == NO_PARENT, NO_PARENT, new_id[np.minimum(par, g.parent.shape[0] - 1)]).astype(np.uint32)
    order, level_parent, offsets = levelize(compact)
    src = keep[order]
    return GltfScene(src, level_parent, offsets, g.pos[src], g.quat[src], g.scale[src], g.mesh[src],
                     g.has_matrix[src], g.matrix[src])
