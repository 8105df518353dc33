"""Reader for the reference's scene files (`assets/scenes/*.scene.json`, version 2)
restricted to what the scene-processing path consumes: the entity hierarchy,
TRS transforms and one draw candidate per renderable entity.

The entity arrays come from the library's reader (vkrhip_scene_read_entities,
C ABI); this module adds the material files and the candidate rows. Follows
scene_loader.c (lib/src/renderer/resources/loaders/scene_loader.c:1230-1460):
`parent` is an index into `entities` or null, `transform` = {pos, rot (x,y,z,w),
scale}. A `shape` of type cube gets the bounds the geometry system gives a cube
(centre 0, extents +-dimensions/2; lib/src/renderer/systems/vkr_geometry_system.c:588-590);
a `mesh` entity gets a unit cube as a placeholder because mesh files are not
read here. Material files (`*.mt`, key=value lines) supply alpha_mode,
double_sided and transmission_factor (> 0 means transmissive,
lib/src/renderer/systems/vkr_material_system.c:82-98).
"""
from __future__ import annotations

import json
import os
from dataclasses import dataclass, field

import numpy as np

from . import _native as N

ALPHA = {"opaque": N.ALPHA_OPAQUE, "cutout": N.ALPHA_CUTOUT, "mask": N.ALPHA_CUTOUT,
         "blend": N.ALPHA_BLEND}


def read_material(path: str) -> dict:
    out = {"alpha_mode": "opaque", "double_sided": False, "transmission_factor": 0.0}
    with open(path) as fh:
        for line in fh:
            line = line.split("#", 1)[0].strip()
            if "=" not in line:
                continue
            key, value = (x.strip() for x in line.split("=", 1))
            key = key.lower()
            if key == "alpha_mode":
                out["alpha_mode"] = value.lower()
            elif key == "double_sided":
                out["double_sided"] = value.lower() in ("true", "1", "yes")
            elif key == "transmission_factor":
                out["transmission_factor"] = float(value)
    return out


def material_bits(mat: dict) -> int:
    bits = ALPHA.get(mat["alpha_mode"], N.ALPHA_OPAQUE) | N.MAT_BOUNDS_VALID
    if mat["double_sided"]:
        bits |= N.MAT_DOUBLE_SIDED
    if mat["transmission_factor"] > 0.0:
        bits |= N.MAT_TRANSMISSIVE
    return bits


@dataclass
class LoadedScene:
    names: list
    parent: np.ndarray        # (n,) u32 in file order, NO_PARENT for roots
    pos: np.ndarray           # (n,4) f32
    quat: np.ndarray
    scale: np.ndarray
    cand_entity: np.ndarray   # (m,) u32 entity index of each candidate row (file order)
    ident: np.ndarray         # (m,8) u32
    matbits: np.ndarray       # (m,) u32
    bounds: np.ndarray        # (m,12) f32 centre / min / max
    notes: list = field(default_factory=list)


def read_entities(text: bytes) -> dict:
    """The entity arrays of a scene document through the library
    (vkrhip_scene_read_entities): parent, TRS, what an entity draws, a shape's
    dimensions and material path. Raises ValueError for what the library rejects."""
    import ctypes as C
    lib = N.load()
    info = N.SceneEntities()
    rc = lib.vkrhip_scene_read_entities(text, len(text), 0, C.byref(info), None, None, None, None, None,
                                        None, None)
    if rc not in (0, N.ERR_CAPACITY):
        raise ValueError("not a scene document the library accepts")
    n = info.entity_count
    out = {"version": info.version, "parent": np.zeros(n, np.uint32), "pos": np.zeros((n, 4), np.float32),
           "quat": np.zeros((n, 4), np.float32), "scale": np.zeros((n, 4), np.float32),
           "kind": np.zeros(n, np.uint8), "dimensions": np.zeros((n, 3), np.float32),
           "material_path": np.zeros((n, 2), np.uint64), "renderables": info.renderable_count}
    if n:
        rc = lib.vkrhip_scene_read_entities(text, len(text), n, C.byref(info), N.ptr(out["parent"]),
                                            N.ptr(out["pos"]), N.ptr(out["quat"]), N.ptr(out["scale"]),
                                            N.ptr(out["kind"]), N.ptr(out["dimensions"]),
                                            N.ptr(out["material_path"]))
        if rc != 0:
            raise ValueError("not a scene document the library accepts")
    return out


def load_scene(path: str, material_dir: str | None = None) -> LoadedScene:
    """Parses one scene file. Materials are looked up by file name in
    `material_dir` (default: the scene file's directory); a missing material
    file means an opaque single-sided material and is recorded in `notes`."""
    with open(path, "rb") as fh:
        text = fh.read()
    ent = read_entities(text)  # the library's reader (C ABI)
    if ent["version"] != 2:
        raise ValueError(f"{path}: scene version {ent['version']} is not supported")
    material_dir = material_dir or os.path.dirname(os.path.abspath(path))
    n = ent["parent"].shape[0]
    parent, pos, quat, scale = ent["parent"], ent["pos"], ent["quat"], ent["scale"]
    # names are only used in the notes below
    names = [e.get("name", f"entity{i}") for i, e in enumerate(json.loads(text)["entities"])]
    rows, notes = [], []
    for i in range(n):
        kind = int(ent["kind"][i])
        if kind == 0:
            continue
        offset, length = (int(x) for x in ent["material_path"][i])
        mat_path = text[offset: offset + length].decode() if length else ""
        if kind == 1:
            dims = ent["dimensions"][i]
        else:
            dims = np.ones(3, np.float32)
            notes.append(f"{names[i]}: mesh file not read, unit-cube bounds")
        mpath = os.path.join(material_dir, os.path.basename(mat_path)) if mat_path else ""
        if mpath and os.path.isfile(mpath):
            m = read_material(mpath)
        else:
            m = {"alpha_mode": "opaque", "double_sided": False, "transmission_factor": 0.0}
            if mat_path:
                notes.append(f"{names[i]}: material {mat_path} not found, opaque assumed")
        rows.append((i, dims * np.float32(0.5), material_bits(m)))
    m_rows = len(rows)
    ident = np.zeros((m_rows, 8), np.uint32)
    matbits = np.zeros(m_rows, np.uint32)
    bounds = np.zeros((m_rows, 12), np.float32)
    cand_entity = np.zeros(m_rows, np.uint32)
    for k, (i, half, bits) in enumerate(rows):
        cand_entity[k] = i
        ident[k] = [k + 1, 1, k + 1, 1, 0, k + 1, 1, i + 1]   # handles by row, render_id by entity
        matbits[k] = bits
        bounds[k, 4:7] = -half
        bounds[k, 8:11] = half
    return LoadedScene(names, parent, pos, quat, scale, cand_entity, ident, matbits, bounds, notes)
