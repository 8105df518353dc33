"""Synthetic instance hierarchies, candidate rows and view sets.

Everything is a pure function of (seed, index): a counter-based generator
(splitmix64 of the element index) so any contiguous range can be produced
independently - which is what sharding across GPUs and bounded CPU samples
need - and the same bytes reach the GPU path and the CPU oracle.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field

import numpy as np

NO_PARENT = 0xFFFFFFFF
ALPHA_OPAQUE, ALPHA_CUTOUT, ALPHA_BLEND = 0, 1, 2
MAT_TRANSMISSIVE, MAT_DOUBLE_SIDED, MAT_BOUNDS_VALID = 0x4, 0x8, 0x10


def _splitmix64(x: np.ndarray) -> np.ndarray:
    x = (x + np.uint64(0x9E3779B97F4A7C15)).astype(np.uint64)
    z = x
    z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
    z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
    return z ^ (z >> np.uint64(31))


def uniform01(seed: int, stream: int, index: np.ndarray) -> np.ndarray:
    """24-bit uniform floats in [0, 1), exact in float32."""
    with np.errstate(over="ignore"):
        key = _splitmix64(np.uint64(seed) * np.uint64(0x632BE59BD9B4E019) + np.uint64(stream))
        h = _splitmix64(index.astype(np.uint64) ^ key)
    return ((h >> np.uint64(40)).astype(np.float32)) * np.float32(2.0 ** -24)


def uniform(seed, stream, index, lo, hi):
    u = uniform01(seed, stream, index)
    return (np.float32(lo) + u * np.float32(hi - lo)).astype(np.float32)


# ---- hierarchy --------------------------------------------------------------
def level_sizes(n: int, depth: int, fanout: int = 8, roots: int = 8) -> list:
    """Geometric level sizes, the last level taking what is left of n."""
    if depth <= 1:
        return [n]
    sizes = []
    s = roots
    for _ in range(depth - 1):
        sizes.append(s)
        s *= fanout
    used = sum(sizes)
    if used >= n:
        # shrink: keep as many full levels as fit and put the remainder last
        sizes2, tot = [], 0
        for s in sizes:
            if tot + s >= n:
                break
            sizes2.append(s)
            tot += s
        sizes2.append(n - tot)
        return sizes2
    sizes.append(n - used)
    return sizes


def build_parents(sizes: list):
    """Level-ordered parent array: children of a parent are contiguous and
    parents ascend inside a level (the reference's BFS topo order)."""
    n = int(sum(sizes))
    parent = np.full(n, NO_PARENT, np.uint32)
    offsets = np.zeros(len(sizes) + 1, np.uint64)
    for l, s in enumerate(sizes):
        offsets[l + 1] = offsets[l] + np.uint64(s)
    for l in range(1, len(sizes)):
        s, ps = sizes[l], sizes[l - 1]
        j = np.arange(s, dtype=np.uint64)
        parent[int(offsets[l]): int(offsets[l + 1])] = (
            offsets[l - 1] + (j * np.uint64(ps)) // np.uint64(s)).astype(np.uint32)
    return parent, offsets


@dataclass
class Scene:
    parent: np.ndarray          # (n,) u32, level ordered
    level_offsets: np.ndarray   # (levels+1,) u64
    pos: np.ndarray             # (n,4) f32
    quat: np.ndarray            # (n,4) f32 x,y,z,w
    scale: np.ndarray           # (n,4) f32
    ident: np.ndarray           # (n,8) u32
    matbits: np.ndarray         # (n,) u32
    bounds: np.ndarray          # (n,12) f32
    node: np.ndarray | None = None
    meta: dict = field(default_factory=dict)

    @property
    def n(self) -> int:
        return self.parent.shape[0]


def make_trs(seed: int, index: np.ndarray, pos_range, scale_range):
    """position uniform in +-pos_range (per element), rotation a normalised
    uniform quaternion, scale uniform in scale_range."""
    n = index.shape[0]
    pos = np.zeros((n, 4), np.float32)
    quat = np.zeros((n, 4), np.float32)
    scale = np.zeros((n, 4), np.float32)
    pr = np.asarray(pos_range, np.float32)
    for c in range(3):
        pos[:, c] = (uniform01(seed, 10 + c, index) * np.float32(2.0) - np.float32(1.0)) * pr
        scale[:, c] = uniform(seed, 20 + c, index, scale_range[0], scale_range[1])
    q = np.stack([uniform(seed, 30 + c, index, -1.0, 1.0) for c in range(4)], axis=1)
    norm = np.sqrt((q.astype(np.float64) ** 2).sum(axis=1))
    norm[norm < 1e-12] = 1.0
    quat[:] = (q / norm[:, None]).astype(np.float32)
    return pos, quat, scale


def make_candidates(seed: int, index: np.ndarray, blend_fraction=0.10, cutout_fraction=0.10,
                    double_sided_fraction=0.10, transmissive_fraction=0.0,
                    no_bounds_fraction=0.0):
    """One candidate row per index: identity handles, material bits, local AABB."""
    n = index.shape[0]
    idx32 = index.astype(np.uint32)
    ident = np.zeros((n, 8), np.uint32)
    ident[:, 0] = idx32 + np.uint32(1)                 # mesh.id = instance slot + 1
    ident[:, 1] = (idx32 >> np.uint32(20)) + np.uint32(1)  # mesh.generation
    ident[:, 2] = (idx32 % np.uint32(1024)) + np.uint32(1)  # geometry.id
    ident[:, 3] = np.uint32(1)
    ident[:, 4] = idx32 % np.uint32(4)                 # submesh_index
    ident[:, 5] = (idx32 % np.uint32(257)) + np.uint32(1)   # material.id
    ident[:, 6] = np.uint32(1)
    ident[:, 7] = idx32 + np.uint32(1)                 # render_id
    u = uniform01(seed, 40, index)
    alpha = np.zeros(n, np.uint32)
    alpha[u < np.float32(blend_fraction)] = ALPHA_BLEND
    alpha[(u >= np.float32(blend_fraction)) &
          (u < np.float32(blend_fraction + cutout_fraction))] = ALPHA_CUTOUT
    matbits = alpha.copy()
    matbits[uniform01(seed, 41, index) < np.float32(double_sided_fraction)] |= MAT_DOUBLE_SIDED
    matbits[uniform01(seed, 42, index) < np.float32(transmissive_fraction)] |= MAT_TRANSMISSIVE
    valid = uniform01(seed, 43, index) >= np.float32(no_bounds_fraction)
    matbits[valid] |= MAT_BOUNDS_VALID
    bounds = np.zeros((n, 12), np.float32)
    for c in range(3):
        center = uniform(seed, 50 + c, index, -0.5, 0.5)
        half = uniform(seed, 60 + c, index, 0.1, 1.0)
        bounds[:, c] = center
        bounds[:, 4 + c] = center - half
        bounds[:, 8 + c] = center + half
    return ident, matbits, bounds


def flat_scene(n: int, seed: int = 1, first: int = 0, **cand_kw) -> Scene:
    """n root nodes (no hierarchy): position +-500, scale [0.5, 2]."""
    index = np.arange(first, first + n, dtype=np.uint64)
    pos, quat, scale = make_trs(seed, index, 500.0, (0.5, 2.0))
    ident, matbits, bounds = make_candidates(seed, index, **cand_kw)
    parent = np.full(n, NO_PARENT, np.uint32)
    offsets = np.array([0, n], np.uint64)
    return Scene(parent, offsets, pos, quat, scale, ident, matbits, bounds,
                 meta={"kind": "flat", "n": n, "seed": seed, "first": first})


def deep_scene(n: int, depth: int = 8, seed: int = 2, fanout: int = 8, roots: int = 8,
               **cand_kw) -> Scene:
    """Depth-`depth` hierarchy: `roots` roots, fan-out `fanout`, the last level
    holding the bulk. Local offsets halve per level (+-250, +-125, ...), so world
    positions cover roughly +-500; per-level scale in [0.9, 1.1]."""
    sizes = level_sizes(n, depth, fanout, roots)
    parent, offsets = build_parents(sizes)
    n = parent.shape[0]
    index = np.arange(n, dtype=np.uint64)
    level = np.zeros(n, np.int32)
    for l in range(len(sizes)):
        level[int(offsets[l]): int(offsets[l + 1])] = l
    pos_range = (np.float32(250.0) * np.float32(0.5) ** level.astype(np.float32)).astype(np.float32)
    pos, quat, scale = make_trs(seed, index, pos_range, (0.9, 1.1))
    ident, matbits, bounds = make_candidates(seed, index, **cand_kw)
    return Scene(parent, offsets, pos, quat, scale, ident, matbits, bounds,
                 meta={"kind": "deep", "n": n, "seed": seed, "sizes": sizes})


def sharded_deep_scene(n_per_shard: int, shards: int, shard: int, depth: int = 8, seed: int = 2,
                       fanout: int = 8, roots: int = 8, **cand_kw) -> Scene:
    """Shard `shard` of deep_scene(n_per_shard * shards): the hierarchy is cut into
    whole trees of about equal size (vkrhip_shard_hierarchy at the root level, so
    `roots` >= `shards`) and only the shard's nodes are generated. Node and row
    data are functions of the GLOBAL node index, so the shards together are the
    whole scene; meta["index_base"] is the number of rows in the shards before
    this one (rows are numbered shard after shard)."""
    from .scene import shard_hierarchy, shard_local_hierarchy
    if roots < shards:
        raise ValueError("a cut at the root level needs at least one root per shard")
    sizes = level_sizes(n_per_shard * shards, depth, fanout, roots)
    parent, offsets = build_parents(sizes)
    ranges = shard_hierarchy(parent, offsets, shards, replicated_levels=0)
    nodes, local_parent, local_offsets = shard_local_hierarchy(parent, ranges[shard])
    del parent
    level = np.zeros(nodes.shape[0], np.int32)
    for l in range(len(sizes)):
        level[int(local_offsets[l]): int(local_offsets[l + 1])] = l
    pos_range = (np.float32(250.0) * np.float32(0.5) ** level.astype(np.float32)).astype(np.float32)
    pos, quat, scale = make_trs(seed, nodes, pos_range, (0.9, 1.1))
    ident, matbits, bounds = make_candidates(seed, nodes, **cand_kw)
    shard_sizes = (ranges[:, :, 1] - ranges[:, :, 0]).sum(axis=1)
    return Scene(local_parent, local_offsets, pos, quat, scale, ident, matbits, bounds,
                 meta={"kind": "deep", "n": int(nodes.shape[0]), "seed": seed, "sizes": sizes,
                       "index_base": int(shard_sizes[:shard].sum()), "total": int(shard_sizes.sum()),
                       "global_nodes": nodes})


# ---- views ------------------------------------------------------------------
def look_at(eye, center, up) -> np.ndarray:
    """Right-handed view matrix (camera looks down -Z), column-major float32.
    View matrices are inputs of the path; this need not match the reference's
    rsqrt-based normalisation bit for bit."""
    eye, center, up = (np.asarray(a, np.float64) for a in (eye, center, up))
    f = center - eye
    f /= np.linalg.norm(f)
    s = np.cross(f, up)
    s /= np.linalg.norm(s)
    u = np.cross(s, f)
    m = np.zeros(16, np.float64)
    m[0], m[1], m[2] = s[0], u[0], -f[0]
    m[4], m[5], m[6] = s[1], u[1], -f[1]
    m[8], m[9], m[10] = s[2], u[2], -f[2]
    m[12], m[13], m[14] = -s.dot(eye), -u.dot(eye), f.dot(eye)
    m[15] = 1.0
    return m.astype(np.float32)


def perspective(fov_deg, aspect, near, far) -> np.ndarray:
    """Vulkan [0,1] depth, Y inverted: the shape of mat4_perspective
    (lib/src/math/mat.h:405-416)."""
    f = 1.0 / math.tan(math.radians(fov_deg) * 0.5)
    m = np.zeros(16, np.float64)
    m[0] = f / aspect
    m[5] = -f
    m[10] = far / (near - far)
    m[11] = -1.0
    m[14] = (far * near) / (near - far)
    return m.astype(np.float32)


def ortho_zo_yinv(l, r, b, t, n, f) -> np.ndarray:
    """The shape of mat4_ortho_zo_yinv (lib/src/math/mat.h:365-375)."""
    m = np.zeros(16, np.float64)
    m[0] = 2.0 / (r - l)
    m[5] = -2.0 / (t - b)
    m[10] = -1.0 / (f - n)
    m[12] = -((r + l) / (r - l))
    m[13] = (t + b) / (t - b)
    m[14] = -(n / (f - n))
    m[15] = 1.0
    return m.astype(np.float32)


def cascade_splits(near, far, count, lam):
    """Practical split scheme (lambda blends logarithmic and uniform)."""
    out = []
    for i in range(1, count + 1):
        p = i / count
        log = near * (far / near) ** p
        uni = near + (far - near) * p
        out.append(lam * log + (1.0 - lam) * uni)
    return out


@dataclass
class ViewSet:
    view: np.ndarray            # camera view (16,)
    projection: np.ndarray      # camera projection (16,)
    light_view_proj: np.ndarray  # (cascades,16)
    extra_view_proj: np.ndarray  # (extra,16) additional perspective views


def camera_views(fov_deg=60.0, aspect=1.0, near=0.1, far=400.0, cascades=0, eye=(0.0, 0.0, 0.0),
                 target=(0.0, 0.0, -1.0), shadow_far=200.0, lam=0.80,
                 light_dir=(-0.57735, -0.57735, -0.57735)) -> ViewSet:
    """Camera at `eye` looking at `target` plus `cascades` orthographic shadow
    cascades fitted to bounding spheres of the camera-frustum slices (4 cascades,
    lambda 0.80, 200 m is the reference's VKR_SHADOW_CONFIG_HIGH,
    lib/src/renderer/systems/vkr_shadow_system.h:194-217)."""
    view = look_at(eye, target, (0.0, 1.0, 0.0))
    proj = perspective(fov_deg, aspect, near, far)
    lvps = []
    if cascades:
        from .scene import mat4_mul
        eye_v = np.asarray(eye, np.float64)
        fwd = np.asarray(target, np.float64) - eye_v
        fwd /= np.linalg.norm(fwd)
        ld = np.asarray(light_dir, np.float64)
        ld /= np.linalg.norm(ld)
        splits = [near] + cascade_splits(near, shadow_far, cascades, lam)
        tan_h = math.tan(math.radians(fov_deg) * 0.5)
        for i in range(cascades):
            d0, d1 = splits[i], splits[i + 1]
            center = eye_v + fwd * (0.5 * (d0 + d1))
            half_diag = d1 * tan_h * math.sqrt(1.0 + aspect * aspect)
            radius = math.sqrt((0.5 * (d1 - d0)) ** 2 + half_diag ** 2)
            pad = 100.0
            leye = center - ld * (radius + pad)
            lview = look_at(leye, center, (0.0, 1.0, 0.0))
            lproj = ortho_zo_yinv(-radius, radius, -radius, radius, 0.1, 2.0 * radius + 2.0 * pad)
            lvps.append(mat4_mul(lproj, lview))
    lvp = np.stack(lvps).astype(np.float32) if lvps else np.zeros((0, 16), np.float32)
    return ViewSet(view, proj, lvp, np.zeros((0, 16), np.float32))


def ring_views(count: int, radius=300.0, fov_deg=60.0, aspect=1.0, near=0.1, far=400.0):
    """`count` perspective views on a ring looking at the origin (probe /
    multi-viewport culling). Returns (count,16) view-projection matrices and the
    first view's (view, projection)."""
    from .scene import mat4_mul
    vps, first = [], None
    proj = perspective(fov_deg, aspect, near, far)
    for k in range(count):
        a = 2.0 * math.pi * k / count
        eye = (radius * math.cos(a), 20.0, radius * math.sin(a))
        view = look_at(eye, (0.0, 0.0, 0.0), (0.0, 1.0, 0.0))
        if first is None:
            first = (view, proj)
        vps.append(mat4_mul(proj, view))
    return np.stack(vps).astype(np.float32), first
