"""ctypes bindings for the CPU oracle (oracle/libvkr_oracle.so) and, when it was
built, the compiled reference (oracle/_ref/libvkr_ref.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg. The shipped GPU path never imports this module.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ORACLE_SO = os.path.join(HERE, "libvkr_oracle.so")
REF_SHADOW_SO = os.path.join(HERE, "_ref", "libvkr_ref_shadow.so")
REF_SO = os.path.join(HERE, "_ref", "libvkr_ref.so")

f32p = np.ctypeslib.ndpointer(dtype=np.float32, flags="C_CONTIGUOUS")
u32p = np.ctypeslib.ndpointer(dtype=np.uint32, flags="C_CONTIGUOUS")
u8p = np.ctypeslib.ndpointer(dtype=np.uint8, flags="C_CONTIGUOUS")
u64p = np.ctypeslib.ndpointer(dtype=np.uint64, flags="C_CONTIGUOUS")
vp = C.c_void_p

NO_PARENT = 0xFFFFFFFF
CULLED = 0xFFFFFFFF

# numpy views of the reference's row layouts (sizes asserted in vkr_oracle.c)
CANDIDATE_DT = np.dtype(
    {
        "names": ["mesh", "geometry", "submesh_index", "material", "model", "object_id",
                  "reserved", "sphere", "state_bucket", "flags"],
        "formats": ["2<u4", "2<u4", "<u4", "2<u4", "16<f4", "<u4", "3<u4", "4<f4", "<u4", "<u4"],
        "offsets": [0, 8, 16, 20, 32, 96, 100, 112, 128, 132],
        "itemsize": 144,
    }
)
DRAW_DT = np.dtype(
    {
        "names": ["mesh", "geometry", "submesh_index", "material", "instance_count",
                  "first_instance", "sort_key"],
        "formats": ["2<u4", "2<u4", "<u4", "2<u4", "<u4", "<u4", "<u8"],
        "offsets": [0, 8, 16, 20, 28, 32, 40],
        "itemsize": 48,
    }
)
INSTANCE_DT = np.dtype(
    {
        "names": ["model", "object_id", "reserved"],
        "formats": ["16<f4", "<u4", "3<u4"],
        "offsets": [0, 64, 68],
        "itemsize": 80,
    }
)
STATE_DT = np.dtype(
    {
        "names": ["execution_ranges", "bucket_counts", "bucket_cursors", "visible_count",
                  "overflow_count", "resolve_invalid_count", "occlusion_culled_count"],
        "formats": ["(4,2)<u4", "4<u4", "4<u4", "<u4", "<u4", "<u4", "<u4"],
        "offsets": [0, 32, 48, 64, 68, 72, 76],
        "itemsize": 80,
    }
)
TRANSPARENT_DT = np.dtype(
    {
        "names": ["model", "mesh", "geometry", "material", "submesh_index", "object_id",
                  "sort_key"],
        "formats": ["16<f4", "2<u4", "2<u4", "2<u4", "<u4", "<u4", "<u8"],
        "offsets": [0, 64, 72, 80, 88, 92, 96],
        "itemsize": 112,
    }
)
GPU_ROW_DT = np.dtype(
    {
        "names": ["geometry_index", "material_index", "instance_index", "first_index",
                  "index_count", "vertex_offset", "state_bucket", "flags", "sphere"],
        "formats": ["<u4", "<u4", "<u4", "<u4", "<u4", "<i4", "<u4", "<u4", "4<f4"],
        "offsets": [0, 4, 8, 12, 16, 20, 24, 28, 32],
        "itemsize": 48,
    }
)
VISIBLE_ROW_DT = np.dtype(
    {
        "names": ["geometry_index", "material_index", "instance_index", "first_index",
                  "index_count", "vertex_offset", "state_bucket", "flags"],
        "formats": ["<u4", "<u4", "<u4", "<u4", "<u4", "<i4", "<u4", "<u4"],
        "offsets": [0, 4, 8, 12, 16, 20, 24, 28],
        "itemsize": 32,
    }
)
GEOMETRY_DT = np.dtype(
    {
        "names": ["generation", "resolvable", "first_index", "submesh_count", "submesh_offset"],
        "formats": ["<u4"] * 5,
        "offsets": [0, 4, 8, 12, 16],
        "itemsize": 32,
    }
)
SUBMESH_DT = np.dtype(
    {
        "names": ["first_index", "index_count", "vertex_offset"],
        "formats": ["<u4", "<u4", "<i4"],
        "offsets": [0, 4, 8],
        "itemsize": 16,
    }
)
MATERIAL_DT = np.dtype(
    {
        "names": ["generation", "live", "slot_index"],
        "formats": ["<u4"] * 3,
        "offsets": [0, 4, 8],
        "itemsize": 16,
    }
)
COMMAND_DT = np.dtype(
    {
        "names": ["index_count", "instance_count", "first_index", "vertex_offset", "first_instance"],
        "formats": ["<u4", "<u4", "<u4", "<i4", "<u4"],
        "offsets": [0, 4, 8, 12, 16],
        "itemsize": 20,
    }
)


def build(force: bool = False) -> None:
    """Compile the oracle (and the reference shim when /root/reference exists)."""
    sources = [os.path.join(HERE, f) for f in ("vkr_oracle.c", "vkr_oracle_shadow.c", "vkr_oracle.h")]
    if force or not os.path.exists(ORACLE_SO) or (
        os.path.getmtime(ORACLE_SO) < max(os.path.getmtime(f) for f in sources)
    ):
        subprocess.check_call(["make", "-C", HERE, os.path.join(HERE, "libvkr_oracle.so")])
    if os.path.isdir("/root/reference/lib/src") and (
            force or not os.path.exists(REF_SO) or not os.path.exists(REF_SHADOW_SO)):
        subprocess.check_call(["make", "-C", HERE, "ref"])


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _u32(a):
    return np.ascontiguousarray(a, dtype=np.uint32)


class Hzb(C.Structure):
    _fields_ = [("mips", C.c_void_p), ("width", C.c_uint32), ("height", C.c_uint32),
                ("mip_count", C.c_uint32), ("depth_epsilon", C.c_float),
                ("view_projection", C.c_float * 16)]


class Oracle:
    """Array-shaped front end of libvkr_oracle.so."""

    def __init__(self, path: str = ORACLE_SO):
        if not os.path.exists(path):
            build()
        self.lib = L = C.CDLL(path)
        L.vko_mat4_mul.argtypes = [f32p, f32p, f32p]
        L.vko_mat4_mul_vec3.argtypes = [f32p, f32p, f32p]
        L.vko_mat4_mul_vec4.argtypes = [f32p, f32p, f32p]
        L.vko_vec3_dot.argtypes = [f32p, f32p]
        L.vko_vec3_dot.restype = C.c_float
        L.vko_vec3_length.argtypes = [f32p]
        L.vko_vec3_length.restype = C.c_float
        L.vko_quat_to_mat4.argtypes = [f32p, f32p]
        L.vko_mat4_translate.argtypes = [f32p, f32p]
        L.vko_mat4_scale.argtypes = [f32p, f32p]
        L.vko_local_from_trs.argtypes = [f32p, f32p, f32p, f32p]
        L.vko_mat4_perspective.argtypes = [C.c_float] * 4 + [f32p]
        L.vko_mat4_ortho_zo_yinv.argtypes = [C.c_float] * 6 + [f32p]
        L.vko_frustum_from_matrix.argtypes = [f32p, f32p]
        L.vko_frustum_from_view_projection.argtypes = [f32p, f32p, f32p]
        L.vko_frustum_test_sphere.argtypes = [f32p, f32p, C.c_float]
        L.vko_frustum_test_sphere.restype = C.c_int
        L.vko_submesh_sphere.argtypes = [f32p, f32p, f32p, f32p, f32p, f32p]
        L.vko_state_bucket.argtypes = [C.c_uint32, C.c_uint32]
        L.vko_state_bucket.restype = C.c_uint32
        L.vko_picking_encode_scene_id.argtypes = [C.c_uint32]
        L.vko_picking_encode_scene_id.restype = C.c_uint32
        L.vko_transparent_depth.argtypes = [f32p, f32p, f32p]
        L.vko_transparent_depth.restype = C.c_float
        L.vko_pack_sort_key.argtypes = [C.c_float, C.c_uint32]
        L.vko_pack_sort_key.restype = C.c_uint64
        L.vko_scene_update.argtypes = [vp, C.c_uint64, u32p, C.c_uint64, vp]
        L.vko_scene_update.restype = C.c_uint64
        L.vko_build_topo_order.argtypes = [u32p, C.c_uint64, u32p, u32p]
        L.vko_build_topo_order.restype = C.c_uint32
        L.vko_transforms_fill.argtypes = [vp, C.c_uint64] + [vp] * 7
        L.vko_transforms_extract.argtypes = [vp, C.c_uint64, vp, vp, vp]
        L.vko_instance_world_bounds.argtypes = [f32p, f32p, C.c_float, f32p]
        L.vko_build_world_payload.argtypes = [
            f32p, vp, u32p, u32p, f32p, C.c_uint64, f32p, f32p, C.c_uint64, C.c_uint64,
            vp, vp, vp, vp, vp, vp, vp, vp, vp]
        L.vko_build_world_payload.restype = C.c_uint32
        L.vko_multiview_cull.argtypes = [
            vp, C.c_uint64, f32p, u32p, C.c_uint32, C.c_uint32, C.c_uint64, C.c_uint64,
            C.c_float, vp, vp, vp, vp]
        L.vko_multiview_cull_flat.argtypes = [
            f32p, vp, f32p, u32p, u32p, C.c_uint64, f32p, u32p, C.c_uint32, C.c_uint32,
            C.c_uint64, C.c_uint64, C.c_float, vp, vp, vp, vp]
        L.vko_encode_visible_rows.argtypes = [vp, u32p, vp, vp]
        L.vko_entity_is_visible.argtypes = [vp, vp, C.c_uint64, C.c_uint64]
        L.vko_entity_is_visible.restype = C.c_int
        L.vko_build_hzb.argtypes = [vp, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, vp]
        L.vko_build_hzb.restype = None
        L.vko_set_hzb.argtypes = [vp]
        L.vko_set_hzb.restype = None
        L.vko_hzb_occluded.argtypes = [vp, vp, C.c_float]
        L.vko_hzb_occluded.restype = C.c_int
        L.vko_candidate_epoch.argtypes = [vp, C.c_uint32]
        L.vko_candidate_epoch.restype = C.c_uint64
        L.vko_merge_runs_desc.argtypes = [vp, u64p, C.c_uint32, u64p]
        L.vko_baseline_create.argtypes = [
            C.c_uint64, u32p, f32p, C.c_uint64, vp, u32p, u32p, f32p, C.c_uint32, f32p,
            u32p, f32p, f32p]
        L.vko_baseline_create.restype = vp
        L.vko_baseline_frame.argtypes = [vp]
        L.vko_baseline_frame.restype = C.c_uint64
        L.vko_baseline_destroy.argtypes = [vp]

    # ---- leaf math ------------------------------------------------------
    def mat4_mul(self, a, b):
        out = np.empty(16, np.float32)
        self.lib.vko_mat4_mul(_f32(a).reshape(16), _f32(b).reshape(16), out)
        return out

    def mat4_mul_vec3(self, m, v):
        out = np.empty(4, np.float32)
        self.lib.vko_mat4_mul_vec3(_f32(m).reshape(16), _f32(v), out)
        return out

    def mat4_mul_vec4(self, m, v):
        out = np.empty(4, np.float32)
        self.lib.vko_mat4_mul_vec4(_f32(m).reshape(16), _f32(v), out)
        return out

    def vec3_dot(self, a, b):
        return np.float32(self.lib.vko_vec3_dot(_f32(a), _f32(b)))

    def vec3_length(self, v):
        return np.float32(self.lib.vko_vec3_length(_f32(v)))

    def quat_to_mat4(self, q):
        out = np.empty(16, np.float32)
        self.lib.vko_quat_to_mat4(_f32(q), out)
        return out

    def local_from_trs(self, pos, quat, scale):
        out = np.empty(16, np.float32)
        self.lib.vko_local_from_trs(_f32(pos), _f32(quat), _f32(scale), out)
        return out

    def perspective(self, fov, aspect, near, far):
        out = np.empty(16, np.float32)
        self.lib.vko_mat4_perspective(fov, aspect, near, far, out)
        return out

    def ortho_zo_yinv(self, l, r, b, t, n, f):
        out = np.empty(16, np.float32)
        self.lib.vko_mat4_ortho_zo_yinv(l, r, b, t, n, f, out)
        return out

    def frustum_from_matrix(self, vp_):
        out = np.empty(24, np.float32)
        self.lib.vko_frustum_from_matrix(_f32(vp_).reshape(16), out)
        return out

    def frustum_from_view_projection(self, view, proj):
        out = np.empty(24, np.float32)
        self.lib.vko_frustum_from_view_projection(_f32(view).reshape(16), _f32(proj).reshape(16), out)
        return out

    def frustum_test_sphere(self, planes, center, radius):
        return bool(self.lib.vko_frustum_test_sphere(_f32(planes).reshape(24), _f32(center), radius))

    def submesh_sphere(self, model, center, mn, mx):
        c = np.empty(3, np.float32)
        r = np.empty(1, np.float32)
        self.lib.vko_submesh_sphere(_f32(model).reshape(16), _f32(center), _f32(mn), _f32(mx), c, r)
        return c, r[0]

    def transparent_key(self, view, model, center, tie_breaker):
        d = self.lib.vko_transparent_depth(_f32(view).reshape(16), _f32(model).reshape(16), _f32(center))
        return int(self.lib.vko_pack_sort_key(d, tie_breaker)), np.float32(d)

    def instance_world_bounds(self, model, center, radius):
        out = np.empty(4, np.float32)
        self.lib.vko_instance_world_bounds(_f32(model).reshape(16), _f32(center), radius, out)
        return out

    # ---- scene update ---------------------------------------------------
    def build_topo_order(self, parent):
        parent = _u32(parent)
        n = parent.shape[0]
        topo = np.empty(n, np.uint32)
        level = np.empty(n, np.uint32)
        levels = self.lib.vko_build_topo_order(parent, n, topo, level) if n else 0
        return topo, level, int(levels)

    def scene_update(self, parent, local=None, world=None, flags=None, pos=None, quat=None,
                     scale=None, topo=None):
        """Runs vkr_scene_update once. Returns (local, world, flags, dirty_list)."""
        parent = _u32(parent)
        n = parent.shape[0]
        rec = np.zeros(n * 208, np.uint8)

        def ptr(a, dt):
            if a is None:
                return None
            a = np.ascontiguousarray(a, dtype=dt)
            keep.append(a)
            return a.ctypes.data

        keep = []
        self.lib.vko_transforms_fill(
            rec.ctypes.data, n, parent.ctypes.data, ptr(pos, np.float32), ptr(quat, np.float32),
            ptr(scale, np.float32), ptr(local, np.float32), ptr(world, np.float32),
            ptr(flags, np.uint8))
        if topo is None:
            topo, _, _ = self.build_topo_order(parent)
        topo = _u32(topo)
        dirty = np.empty(max(n, 1), np.uint32)
        nd = self.lib.vko_scene_update(rec.ctypes.data, n, topo, topo.shape[0], dirty.ctypes.data)
        lo = np.empty((n, 16), np.float32)
        wo = np.empty((n, 16), np.float32)
        fo = np.empty(n, np.uint8)
        self.lib.vko_transforms_extract(rec.ctypes.data, n, lo.ctypes.data, wo.ctypes.data,
                                        fo.ctypes.data)
        return lo, wo, fo, dirty[:nd].copy()

    # ---- payload --------------------------------------------------------
    def build_world_payload(self, world, node, ident, matbits, bounds, view, proj, capacity=0,
                            source_index_base=0):
        world = _f32(world).reshape(-1, 16)
        ident = _u32(ident).reshape(-1, 8)
        n = ident.shape[0]
        matbits = _u32(matbits)
        bounds = _f32(bounds).reshape(-1, 12)
        node_arr = None if node is None else _u32(node)
        cand = np.zeros(max(n, 1), CANDIDATE_DT)
        trans = np.zeros(max(n, 1), CANDIDATE_DT)
        tvis = np.zeros(max(n, 1), np.uint8)
        draws = np.zeros(max(n, 1), DRAW_DT)
        inst = np.zeros(max(n, 1), INSTANCE_DT)
        keys = np.zeros(max(n, 1), np.uint64)
        src = np.zeros(max(n, 1), np.uint32)
        stats = np.zeros(3, np.uint32)
        counts = np.zeros(4, np.uint32)
        t = self.lib.vko_build_world_payload(
            world, None if node_arr is None else node_arr.ctypes.data, ident, matbits, bounds, n,
            _f32(view).reshape(16), _f32(proj).reshape(16), capacity, source_index_base,
            cand.ctypes.data, trans.ctypes.data, tvis.ctypes.data, draws.ctypes.data,
            inst.ctypes.data, keys.ctypes.data, src.ctypes.data, stats.ctypes.data,
            counts.ctypes.data)
        overflow = capacity and n > capacity
        return {
            "gpu_candidates": cand[: 0 if overflow else n],
            "transmission_candidates": trans[: int(counts[2])],
            "transparent_visible": tvis[:n],
            "draws": draws[:t],
            "instances": inst[:t],
            "sort_keys": keys[:t],
            "sorted_source": src[:t],
            "stats": stats,
            "counts": counts,
        }

    def multiview_cull(self, candidates, planes, required_flags, partition_capacity=0,
                       index_base=0, tie_eps=0.0):
        n = candidates.shape[0]
        planes = _f32(planes).reshape(-1, 24)
        v = planes.shape[0]
        cls = np.empty((v, max(n, 1)), np.uint32)
        states = np.zeros(v, STATE_DT)
        vis = np.full((v, max(n, 1)), 0xFFFFFFFF, np.uint32)
        ties = np.zeros(v, np.uint64)
        c = np.ascontiguousarray(candidates)
        self.lib.vko_multiview_cull(
            c.ctypes.data, n, planes, _u32(required_flags), v, partition_capacity, max(n, 1),
            index_base, tie_eps, cls.ctypes.data, states.ctypes.data, vis.ctypes.data,
            ties.ctypes.data)
        return cls[:, :n], states, vis[:, :n], ties

    def multiview_cull_flat(self, world, node, sphere4, flags, bucket, planes, required_flags,
                            partition_capacity=0, index_base=0, tie_eps=0.0):
        sphere4 = _f32(sphere4).reshape(-1, 4)
        n = sphere4.shape[0]
        planes = _f32(planes).reshape(-1, 24)
        v = planes.shape[0]
        node_arr = None if node is None else _u32(node)
        cls = np.empty((v, max(n, 1)), np.uint32)
        states = np.zeros(v, STATE_DT)
        vis = np.full((v, max(n, 1)), 0xFFFFFFFF, np.uint32)
        ties = np.zeros(v, np.uint64)
        self.lib.vko_multiview_cull_flat(
            _f32(world).reshape(-1), None if node_arr is None else node_arr.ctypes.data, sphere4,
            _u32(flags), _u32(bucket), n, planes, _u32(required_flags), v, partition_capacity,
            max(n, 1), index_base, tie_eps, cls.ctypes.data, states.ctypes.data,
            vis.ctypes.data, ties.ctypes.data)
        return cls[:, :n], states, vis[:, :n], ties

    def encode_visible_rows(self, rows, visible, state):
        rows = np.ascontiguousarray(rows)
        out = np.zeros(int(state["visible_count"]), VISIBLE_ROW_DT)
        st = np.ascontiguousarray(state)
        self.lib.vko_encode_visible_rows(rows.ctypes.data, _u32(visible), st.ctypes.data,
                                         out.ctypes.data)
        return out

    def pack_gpu_candidates(self, candidates, geometries, submeshes, materials):
        """vkr_vk_pack_gpu_candidates. Returns (ok, rows48, instances80, packed_source, counts4)."""
        c = np.ascontiguousarray(candidates)
        g = np.ascontiguousarray(geometries, dtype=GEOMETRY_DT)
        sm = np.ascontiguousarray(submeshes, dtype=SUBMESH_DT)
        m = np.ascontiguousarray(materials, dtype=MATERIAL_DT)
        n = c.shape[0]
        rows = np.zeros(max(n, 1), GPU_ROW_DT)
        inst = np.zeros(max(n, 1), INSTANCE_DT)
        src = np.zeros(max(n, 1), np.uint32)
        counts = np.zeros(4, np.uint32)
        self.lib.vko_pack_gpu_candidates.restype = C.c_int
        ok = self.lib.vko_pack_gpu_candidates(
            C.c_void_p(c.ctypes.data), C.c_uint32(n), C.c_void_p(g.ctypes.data),
            C.c_uint32(g.shape[0]), C.c_void_p(sm.ctypes.data), C.c_void_p(m.ctypes.data),
            C.c_uint32(m.shape[0]), C.c_void_p(rows.ctypes.data), C.c_void_p(inst.ctypes.data),
            C.c_void_p(src.ctypes.data), C.c_void_p(counts.ctypes.data))
        k = int(counts[0])
        return bool(ok), rows[:k], inst[:k], src[:k], counts

    def encode_commands(self, rows, visible, state, view, partition_capacity, views):
        """Indirect commands of one view into a (views*4*partition,) command array."""
        rows = np.ascontiguousarray(rows)
        out = np.zeros(views * 4 * partition_capacity, COMMAND_DT)
        st = np.ascontiguousarray(state)
        vis = _u32(visible)
        self.lib.vko_encode_commands(
            C.c_void_p(rows.ctypes.data), C.c_void_p(vis.ctypes.data), C.c_void_p(st.ctypes.data),
            C.c_uint32(view), C.c_uint32(partition_capacity), C.c_void_p(out.ctypes.data))
        return out

    def entity_is_visible(self, parent, spec):
        """vko_entity_is_visible for every node."""
        parent = np.ascontiguousarray(parent, dtype=np.uint32)
        spec = np.ascontiguousarray(spec, dtype=np.uint8)
        n = parent.shape[0]
        return np.array([self.lib.vko_entity_is_visible(parent.ctypes.data, spec.ctypes.data, n, i)
                         for i in range(n)], np.uint8)

    def set_hzb(self, mips, width=0, height=0, view_projection=None, depth_epsilon=0.0):
        """Depth pyramid for view 0 of the following multiview_cull* calls (None: off)."""
        if mips is None:
            self._hzb = None
            self.lib.vko_set_hzb(None)
            return
        flat = np.ascontiguousarray(np.concatenate([np.asarray(m, np.float32).reshape(-1) for m in mips]))
        h = Hzb()
        h.mips = flat.ctypes.data
        h.width, h.height, h.mip_count = width, height, len(mips)
        h.depth_epsilon = depth_epsilon
        vp = np.asarray(view_projection, np.float32).reshape(16)
        for k in range(16):
            h.view_projection[k] = float(vp[k])
        self._hzb = (h, flat)  # keeps the memory alive
        self.lib.vko_set_hzb(C.byref(h))

    def build_hzb(self, depth, width, height, mip_count):
        """List of levels built from a depth image (vk_hzb_build)."""
        depth = np.ascontiguousarray(depth, dtype=np.float32)
        extents = [(max(height >> m, 1), max(width >> m, 1)) for m in range(mip_count)]
        flat = np.zeros(sum(h * w for h, w in extents), np.float32)
        self.lib.vko_build_hzb(depth.ctypes.data, depth.shape[1], depth.shape[0], width, height,
                               mip_count, flat.ctypes.data)
        out, pos = [], 0
        for h, w in extents:
            out.append(flat[pos: pos + h * w].reshape(h, w).copy())
            pos += h * w
        return out

    def hzb_occluded(self, center3, radius):
        c = np.ascontiguousarray(center3, dtype=np.float32)
        return bool(self.lib.vko_hzb_occluded(C.byref(self._hzb[0]), c.ctypes.data, float(radius)))

    def candidate_epoch(self, candidates):
        c = np.ascontiguousarray(candidates)
        return int(self.lib.vko_candidate_epoch(c.ctypes.data, c.shape[0]))

    def merge_runs_desc(self, runs):
        runs = [np.ascontiguousarray(r, dtype=np.uint64) for r in runs]
        ptrs = (C.c_void_p * len(runs))(*[r.ctypes.data for r in runs])
        lengths = np.array([r.shape[0] for r in runs], np.uint64)
        out = np.empty(int(lengths.sum()), np.uint64)
        self.lib.vko_merge_runs_desc(C.cast(ptrs, vp), lengths, len(runs), out)
        return out

    # ---- baseline -------------------------------------------------------
    def baseline_create(self, parent, local, node, ident, matbits, bounds, planes,
                        required_flags, view, proj):
        parent = _u32(parent)
        ident = _u32(ident).reshape(-1, 8)
        planes = _f32(planes).reshape(-1, 24)
        node_arr = None if node is None else _u32(node)
        h = self.lib.vko_baseline_create(
            parent.shape[0], parent, _f32(local).reshape(-1), ident.shape[0],
            None if node_arr is None else node_arr.ctypes.data, ident, _u32(matbits),
            _f32(bounds).reshape(-1), planes.shape[0], planes, _u32(required_flags),
            _f32(view).reshape(16), _f32(proj).reshape(16))
        if not h:
            raise RuntimeError("vko_baseline_create failed")
        return h

    def baseline_frame(self, h):
        return int(self.lib.vko_baseline_frame(h))

    def baseline_destroy(self, h):
        self.lib.vko_baseline_destroy(h)


class Reference:
    """Front end of oracle/_ref/libvkr_ref.so (the reference's own code)."""

    def __init__(self, path: str = REF_SO):
        if not os.path.exists(path):
            raise FileNotFoundError(path)
        self.lib = L = C.CDLL(path)
        L.ref_layout.argtypes = [u32p]
        L.ref_payload_layout.argtypes = [u32p]
        L.ref_mat4_mul.argtypes = [f32p, f32p, f32p]
        L.ref_mat4_mul_vec3.argtypes = [f32p, f32p, f32p]
        L.ref_mat4_mul_vec4.argtypes = [f32p, f32p, f32p]
        L.ref_vec3_dot.argtypes = [f32p, f32p]
        L.ref_vec3_dot.restype = C.c_float
        L.ref_vec3_length.argtypes = [f32p]
        L.ref_vec3_length.restype = C.c_float
        L.ref_quat_to_mat4.argtypes = [f32p, f32p]
        L.ref_mat4_translate.argtypes = [f32p, f32p]
        L.ref_mat4_scale.argtypes = [f32p, f32p]
        L.ref_mat4_perspective.argtypes = [C.c_float] * 4 + [f32p]
        L.ref_mat4_ortho_zo_yinv.argtypes = [C.c_float] * 6 + [f32p]
        L.ref_mat4_look_at.argtypes = [f32p, f32p, f32p, f32p]
        L.ref_to_radians.argtypes = [C.c_float]
        L.ref_to_radians.restype = C.c_float
        L.ref_frustum_from_matrix.argtypes = [f32p, f32p]
        L.ref_frustum_from_view_projection.argtypes = [f32p, f32p, f32p]
        L.ref_frustum_test_sphere.argtypes = [f32p, f32p, C.c_float]
        L.ref_frustum_test_sphere.restype = C.c_int
        L.ref_submesh_sphere.argtypes = [f32p, f32p, f32p, f32p, f32p, f32p]
        L.ref_state_bucket.argtypes = [C.c_uint32, C.c_uint32]
        L.ref_state_bucket.restype = C.c_uint32
        L.ref_alpha_routing.argtypes = [C.c_uint32, u32p]
        L.ref_picking_encode_scene.argtypes = [C.c_uint32]
        L.ref_picking_encode_scene.restype = C.c_uint32
        L.ref_transparent_sort_emit.argtypes = [vp, C.c_uint32, vp, vp]
        L.ref_transparent_sort_emit.restype = C.c_uint32
        L.refc_local_from_trs.argtypes = [f32p, f32p, f32p, f32p]
        L.refc_propagate.argtypes = [u32p, u32p, C.c_uint64, f32p, f32p]
        L.refc_transparent_key.argtypes = [f32p, f32p, f32p, C.c_uint32, vp]
        L.refc_transparent_key.restype = C.c_uint64
        L.refc_half_extent_length.argtypes = [f32p, f32p]
        L.refc_half_extent_length.restype = C.c_float
        L.refc_sphere_visible.argtypes = [f32p, f32p, f32p, f32p]
        L.refc_sphere_visible.restype = C.c_int

    def layout(self):
        out = np.zeros(32, np.uint32)
        self.lib.ref_layout(out)
        return out

    def payload_layout(self):
        """sizeof/offsetof of VkrWorldPassPayload and VkrVisibilityStats as the
        reference's header (application.h:696-750) compiles here, then
        VKR_GPU_DRAW_CANDIDATE_CAPACITY."""
        out = np.zeros(18, np.uint32)
        self.lib.ref_payload_layout(out)
        return out

    def mat4_mul(self, a, b):
        out = np.empty(16, np.float32)
        self.lib.ref_mat4_mul(_f32(a).reshape(16), _f32(b).reshape(16), out)
        return out

    def mat4_mul_vec3(self, m, v):
        out = np.empty(4, np.float32)
        self.lib.ref_mat4_mul_vec3(_f32(m).reshape(16), _f32(v), out)
        return out

    def mat4_mul_vec4(self, m, v):
        out = np.empty(4, np.float32)
        self.lib.ref_mat4_mul_vec4(_f32(m).reshape(16), _f32(v), out)
        return out

    def vec3_dot(self, a, b):
        return np.float32(self.lib.ref_vec3_dot(_f32(a), _f32(b)))

    def vec3_length(self, v):
        return np.float32(self.lib.ref_vec3_length(_f32(v)))

    def quat_to_mat4(self, q):
        out = np.empty(16, np.float32)
        self.lib.ref_quat_to_mat4(_f32(q), out)
        return out

    def local_from_trs(self, pos, quat, scale):
        out = np.empty(16, np.float32)
        self.lib.refc_local_from_trs(_f32(pos), _f32(quat), _f32(scale), out)
        return out

    def perspective(self, fov, aspect, near, far):
        out = np.empty(16, np.float32)
        self.lib.ref_mat4_perspective(fov, aspect, near, far, out)
        return out

    def ortho_zo_yinv(self, l, r, b, t, n, f):
        out = np.empty(16, np.float32)
        self.lib.ref_mat4_ortho_zo_yinv(l, r, b, t, n, f, out)
        return out

    def look_at(self, eye, center, up):
        out = np.empty(16, np.float32)
        self.lib.ref_mat4_look_at(_f32(eye), _f32(center), _f32(up), out)
        return out

    def to_radians(self, deg):
        return np.float32(self.lib.ref_to_radians(deg))

    def frustum_from_matrix(self, vp_):
        out = np.empty(24, np.float32)
        self.lib.ref_frustum_from_matrix(_f32(vp_).reshape(16), out)
        return out

    def frustum_from_view_projection(self, view, proj):
        out = np.empty(24, np.float32)
        self.lib.ref_frustum_from_view_projection(_f32(view).reshape(16), _f32(proj).reshape(16), out)
        return out

    def frustum_test_sphere(self, planes, center, radius):
        return bool(self.lib.ref_frustum_test_sphere(_f32(planes).reshape(24), _f32(center), radius))

    def submesh_sphere(self, model, center, mn, mx):
        c = np.empty(3, np.float32)
        r = np.empty(1, np.float32)
        self.lib.ref_submesh_sphere(_f32(model).reshape(16), _f32(center), _f32(mn), _f32(mx), c, r)
        return c, r[0]

    def propagate(self, topo, parent, local):
        local = _f32(local).reshape(-1, 16)
        world = np.zeros_like(local)
        self.lib.refc_propagate(_u32(topo), _u32(parent), local.shape[0], local.reshape(-1),
                                world.reshape(-1))
        return world

    def transparent_key(self, view, model, center, tie_breaker):
        d = C.c_float(0)
        k = self.lib.refc_transparent_key(_f32(view).reshape(16), _f32(model).reshape(16),
                                          _f32(center), tie_breaker, C.addressof(d))
        return int(k), np.float32(d.value)

    def half_extent_length(self, mn, mx):
        return np.float32(self.lib.refc_half_extent_length(_f32(mn), _f32(mx)))

    def sphere_visible(self, model, sphere4, planes):
        out = np.empty(4, np.float32)
        v = self.lib.refc_sphere_visible(_f32(model).reshape(16), _f32(sphere4), _f32(planes).reshape(24), out)
        return bool(v), out

    def transparent_sort_emit(self, candidates):
        c = np.ascontiguousarray(candidates).copy()
        n = c.shape[0]
        draws = np.zeros(max(n, 1), DRAW_DT)
        inst = np.zeros(max(n, 1), INSTANCE_DT)
        self.lib.ref_transparent_sort_emit(c.ctypes.data, n, draws.ctypes.data, inst.ctypes.data)
        return draws[:n], inst[:n]


def have_reference() -> bool:
    return os.path.exists(REF_SO)


# ---- cascade fitting ---------------------------------------------------------
SHADOW_CASCADES_MAX = 8


class ShadowFit(C.Structure):
    _fields_ = [(n, C.c_float) for n in ("center_x", "center_y", "extent", "min_z", "max_z",
                                         "world_units_per_texel")]

    def tuple(self):
        return tuple(np.float32(getattr(self, n)) for n, _ in self._fields_)


class ShadowConfig(C.Structure):
    _fields_ = [("cascade_count", C.c_uint32), ("shadow_map_size", C.c_uint32),
                ("cascade_split_lambda", C.c_float), ("max_shadow_distance", C.c_float),
                ("cascade_guard_band_texels", C.c_float),
                ("cascade_guard_band_texels_per", C.c_float * 8),
                ("z_extension_factor", C.c_float),
                ("cascade_z_extension_factor_per", C.c_float * 8),
                ("use_constant_cascade_size", C.c_uint32), ("anchor_snap_texels", C.c_float),
                ("stabilize_cascades", C.c_uint32), ("use_scene_bounds", C.c_uint32),
                ("scene_min", C.c_float * 3), ("scene_max", C.c_float * 3)]


class ShadowCamera(C.Structure):
    _fields_ = [("type", C.c_uint32), ("position", C.c_float * 3), ("forward", C.c_float * 3),
                ("right", C.c_float * 3), ("up", C.c_float * 3), ("near_clip", C.c_float),
                ("far_clip", C.c_float), ("zoom", C.c_float), ("window_width", C.c_uint32),
                ("window_height", C.c_uint32), ("left_clip", C.c_float), ("right_clip", C.c_float),
                ("bottom_clip", C.c_float), ("top_clip", C.c_float)]


class ShadowCascade(C.Structure):
    _fields_ = [("view_projection", C.c_float * 16), ("split_far", C.c_float),
                ("world_units_per_texel", C.c_float), ("light_space_origin", C.c_float * 2),
                ("light_space_depth_span", C.c_float), ("bounds_center", C.c_float * 3),
                ("bounds_radius", C.c_float)]


class ShadowSystem(C.Structure):
    _fields_ = [("config", ShadowConfig), ("cascades", ShadowCascade * 8),
                ("cascade_splits", C.c_float * 9), ("light_direction", C.c_float * 3),
                ("light_enabled", C.c_uint32), ("history_fits", ShadowFit * 8),
                ("history_light_direction", C.c_float * 3), ("history_cascade_count", C.c_uint32),
                ("history_shadow_map_size", C.c_uint32),
                ("history_projection_convention", C.c_uint32), ("history_valid", C.c_uint32),
                ("history_enable_generation", C.c_uint64), ("enable_generation", C.c_uint64),
                ("initialized", C.c_uint32), ("reserved", C.c_uint32)]


def shadow_config_high() -> ShadowConfig:
    """VKR_SHADOW_CONFIG_HIGH (= the default), vkr_shadow_system.h:194-217."""
    c = ShadowConfig()
    c.cascade_count, c.shadow_map_size = 4, 2048
    c.cascade_split_lambda, c.max_shadow_distance = 0.80, 200.0
    c.cascade_guard_band_texels = 32.0
    c.cascade_guard_band_texels_per[:4] = [16.0, 24.0, 32.0, 48.0]
    c.z_extension_factor = 4.0
    c.cascade_z_extension_factor_per[:4] = [2.0, 3.0, 4.0, 5.0]
    c.use_constant_cascade_size, c.anchor_snap_texels, c.stabilize_cascades = 1, 16.0, 1
    c.use_scene_bounds = 0
    c.scene_min[:] = [-20.0, -20.0, -20.0]
    c.scene_max[:] = [20.0, 20.0, 20.0]
    return c


def shadow_camera(position, forward, right, up, near_clip=0.1, far_clip=500.0, zoom=70.0,
                  window=(1280, 720), ortho=None) -> ShadowCamera:
    cam = ShadowCamera()
    cam.type = 1 if ortho is not None else 0
    cam.position[:] = [float(x) for x in position]
    cam.forward[:] = [float(x) for x in forward]
    cam.right[:] = [float(x) for x in right]
    cam.up[:] = [float(x) for x in up]
    cam.near_clip, cam.far_clip, cam.zoom = near_clip, far_clip, zoom
    cam.window_width, cam.window_height = window
    if ortho is not None:
        cam.left_clip, cam.right_clip, cam.bottom_clip, cam.top_clip = ortho
    return cam


class _ShadowApi:
    """The cascade-fit entry points, which the oracle (vko_ prefix) and the
    reference shim (ref_ prefix) export with the same signatures."""

    def _bind_shadow(self, L, prefix):
        f32p, fitp = np.ctypeslib.ndpointer(np.float32, flags="C"), C.POINTER(ShadowFit)
        cfgp, camp = C.POINTER(ShadowConfig), C.POINTER(ShadowCamera)
        g = lambda name: getattr(L, prefix + name)  # noqa: E731
        g("shadow_cascade_splits").argtypes = [cfgp, C.c_float, C.c_float, C.c_float, f32p]
        g("shadow_frustum_corners").argtypes = [camp, C.c_float, C.c_float, f32p]
        g("shadow_light_view").argtypes = [camp, cfgp, f32p, C.c_uint32, f32p]
        g("shadow_fit_relevant_caster_z").argtypes = [f32p, f32p, f32p] + [C.c_float] * 4 + [f32p, f32p]
        g("shadow_fit_relevant_caster_z").restype = C.c_int
        g("shadow_quantize_extent_up").argtypes = [C.c_float, C.c_uint32]
        g("shadow_quantize_extent_up").restype = C.c_float
        g("shadow_apply_fit_hysteresis").argtypes = [fitp, fitp, fitp]
        g("shadow_cascade_matrix").argtypes = [f32p, f32p, C.c_uint32, C.c_int, C.c_float, C.c_int, cfgp,
                                               C.c_float, fitp, f32p, fitp, f32p]
        g("mat4_inverse_rigid").argtypes = [f32p, f32p]
        self._sh = g

    def cascade_splits(self, cfg, near_clip, far_clip, lam):
        out = np.zeros(9, np.float32)
        self._sh("shadow_cascade_splits")(C.byref(cfg), near_clip, far_clip, lam, out)
        return out[: cfg.cascade_count + 1] if cfg.cascade_count else out[:2]

    def frustum_corners(self, cam, near_split, far_split):
        out = np.zeros(24, np.float32)
        self._sh("shadow_frustum_corners")(C.byref(cam), near_split, far_split, out)
        return out.reshape(8, 3)

    def light_view(self, cam, cfg, light_direction, shadow_map_size):
        out = np.zeros(16, np.float32)
        self._sh("shadow_light_view")(C.byref(cam), C.byref(cfg), _f32(light_direction), shadow_map_size, out)
        return out

    def fit_relevant_caster_z(self, light_view, bounds_min, bounds_max, left, right, bottom, top):
        lo, hi = np.zeros(1, np.float32), np.zeros(1, np.float32)
        ok = self._sh("shadow_fit_relevant_caster_z")(_f32(light_view).reshape(16), _f32(bounds_min),
                                                      _f32(bounds_max), left, right, bottom, top, lo, hi)
        return bool(ok), lo[0], hi[0]

    def quantize_extent_up(self, extent, shadow_map_size):
        return np.float32(self._sh("shadow_quantize_extent_up")(extent, shadow_map_size))

    def apply_fit_hysteresis(self, previous, raw):
        out = ShadowFit()
        self._sh("shadow_apply_fit_hysteresis")(C.byref(previous), C.byref(raw), C.byref(out))
        return out

    def cascade_matrix(self, light_view, corners, shadow_map_size, stabilize, guard_band_texels,
                       use_constant_cascade_size, cfg, z_extension_factor, previous_fit=None):
        vp, origin, fit = np.zeros(16, np.float32), np.zeros(2, np.float32), ShadowFit()
        self._sh("shadow_cascade_matrix")(
            _f32(light_view).reshape(16), _f32(corners).reshape(24), shadow_map_size, int(stabilize),
            guard_band_texels, int(use_constant_cascade_size), C.byref(cfg), z_extension_factor,
            None if previous_fit is None else C.byref(previous_fit), vp, C.byref(fit), origin)
        return vp, fit, origin

    def inverse_rigid(self, m):
        out = np.zeros(16, np.float32)
        self._sh("mat4_inverse_rigid")(_f32(m).reshape(16), out)
        return out


class OracleShadow(_ShadowApi):
    """The cascade fit of the CPU oracle (vkr_oracle_shadow.c)."""

    def __init__(self, path: str = ORACLE_SO):
        self.lib = L = C.CDLL(path)
        self._bind_shadow(L, "vko_")
        sysp = C.POINTER(ShadowSystem)
        L.vko_shadow_system_init.argtypes = [sysp, C.POINTER(ShadowConfig)]
        L.vko_shadow_system_init.restype = C.c_int
        L.vko_shadow_system_invalidate_fit_history.argtypes = [sysp]
        L.vko_shadow_system_update.argtypes = [sysp, C.POINTER(ShadowCamera), C.c_int,
                                               np.ctypeslib.ndpointer(np.float32, flags="C")]

    def system(self, cfg):
        s = ShadowSystem()
        assert self.lib.vko_shadow_system_init(C.byref(s), C.byref(cfg))
        return s

    def update(self, s, cam, light_enabled, light_direction):
        self.lib.vko_shadow_system_update(C.byref(s), C.byref(cam), int(light_enabled), _f32(light_direction))
        return s

    def invalidate(self, s):
        self.lib.vko_shadow_system_invalidate_fit_history(C.byref(s))


class ReferenceShadow(_ShadowApi):
    """The reference's own cascade fit (oracle/_ref/libvkr_ref_shadow.so). Loaded
    with lazy binding: the reference's file names a window function that only its
    macOS and Windows builds define, and no call made here reaches it."""

    def __init__(self, path: str = REF_SHADOW_SO):
        if not os.path.exists(path):
            raise FileNotFoundError(path)
        self.lib = L = C.CDLL(path, mode=os.RTLD_LAZY)
        self._bind_shadow(L, "ref_")
        f32p = np.ctypeslib.ndpointer(np.float32, flags="C")
        L.ref_shadow_default_config.argtypes = [C.POINTER(ShadowConfig)]
        L.ref_vec3_normalize.argtypes = [f32p, f32p]
        L.ref_shadow_system_create.argtypes = [C.POINTER(ShadowConfig)]
        L.ref_shadow_system_create.restype = C.c_void_p
        L.ref_shadow_system_destroy.argtypes = [C.c_void_p]
        L.ref_shadow_system_invalidate.argtypes = [C.c_void_p]
        L.ref_shadow_system_set_stabilize.argtypes = [C.c_void_p, C.c_int]
        L.ref_shadow_system_update.argtypes = [C.c_void_p, C.POINTER(ShadowCamera), C.c_int, f32p]
        L.ref_shadow_system_read.argtypes = [C.c_void_p, C.POINTER(ShadowSystem)]
        L.ref_shadow_system_poke_history.argtypes = [C.c_void_p, C.c_uint32, C.POINTER(ShadowFit)]

    def default_config(self):
        c = ShadowConfig()
        self.lib.ref_shadow_default_config(C.byref(c))
        return c

    def normalize(self, v):
        out = np.zeros(3, np.float32)
        self.lib.ref_vec3_normalize(_f32(v), out)
        return out

    def system(self, cfg):
        h = self.lib.ref_shadow_system_create(C.byref(cfg))
        assert h
        return h

    def destroy(self, h):
        self.lib.ref_shadow_system_destroy(h)

    def update(self, h, cam, light_enabled, light_direction):
        self.lib.ref_shadow_system_update(h, C.byref(cam), int(light_enabled), _f32(light_direction))

    def read(self, h):
        s = ShadowSystem()
        self.lib.ref_shadow_system_read(h, C.byref(s))
        return s

    def invalidate(self, h):
        self.lib.ref_shadow_system_invalidate(h)

    def set_stabilize(self, h, on):
        self.lib.ref_shadow_system_set_stabilize(h, int(on))

    def poke_history(self, h, cascade, fit):
        self.lib.ref_shadow_system_poke_history(h, cascade, C.byref(fit))


def have_reference_shadow() -> bool:
    return os.path.exists(REF_SHADOW_SO)
