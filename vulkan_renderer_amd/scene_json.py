"""Reader for the reference's scene files (`assets/scenes/*.scene.json`, version 2)
restricted to what the scene-processing path consumes: the entity hierarchy,
TRS transforms and one draw candidate per renderable entity.

Follows scene_loader.c (lib/src/renderer/resources/loaders/scene_loader.c:1230-1460):
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


def load_scene(path: str, material_dir: str | None = None) -> LoadedScene:
    """Parses one scene file. Materials are looked up by file name in
    `material_dir` (default: the scene file's directory); a missing material
    file means an opaque single-sided material and is recorded in `notes`."""
    with open(path) as fh:
        doc = json.load(fh)
    if doc.get("version") != 2:
        raise ValueError(f"{path}: scene version {doc.get('version')} is not supported")
    material_dir = material_dir or os.path.dirname(os.path.abspath(path))
    ents = doc["entities"]
    n = len(ents)
    parent = np.full(n, N.NO_PARENT, np.uint32)
    pos = np.zeros((n, 4), np.float32)
    quat = np.zeros((n, 4), np.float32)
    quat[:, 3] = 1.0
    scale = np.ones((n, 4), np.float32)
    scale[:, 3] = 0.0
    rows, notes, names = [], [], []
    for i, e in enumerate(ents):
        names.append(e.get("name", f"entity{i}"))
        if e.get("parent") is not None:
            p = int(e["parent"])
            if not 0 <= p < n or p == i:
                raise ValueError(f"{path}: entity {i} has parent {p}")
            parent[i] = p
        t = e.get("transform", {})
        pos[i, :3] = t.get("pos", [0, 0, 0])
        quat[i] = t.get("rot", [0, 0, 0, 1])
        scale[i, :3] = t.get("scale", [1, 1, 1])
        if "shape" in e:
            dims = np.asarray(e["shape"].get("dimensions", [1, 1, 1]), np.float32)
            mat = e["shape"].get("material", {})
        elif "mesh" in e:
            dims = np.ones(3, np.float32)
            mat = {}
            notes.append(f"{names[-1]}: mesh file not read, unit-cube bounds")
        else:
            continue
        mpath = os.path.join(material_dir, os.path.basename(mat.get("path", ""))) if mat else ""
        if mpath and os.path.isfile(mpath):
            m = read_material(mpath)
        else:
            m = {"alpha_mode": "opaque", "double_sided": False, "transmission_factor": 0.0}
            if mat:
                notes.append(f"{names[-1]}: material {mat.get('path')} not found, opaque assumed")
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
