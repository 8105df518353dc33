"""Host-side front end of the gfx950 scene-processing library.

Mirrors the reference interfaces of the path it stands behind:

* ``HipScene.set_hierarchy / upload_trs / upload_locals`` - the SceneTransform
  columns that ``vkr_scene_update`` walks
  (lib/src/renderer/systems/vkr_scene_system.c:1254-1337);
* ``HipScene.set_candidates`` - the (instance, submesh) rows that
  ``application_build_world_payload`` enumerates (lib/src/application.h:730-806);
* ``HipScene.set_views`` - camera + cascades as the backend builds them
  (lib/src/renderer/vulkan/vkr_vulkan_deferred.c:242-295);
* ``HipScene.frame`` - one pass of transform -> cull -> sort on the device.

Every call goes through the C ABI in include/vkr_hip_scene.h; there is no CPU
fallback.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass

import numpy as np

from . import _native as N

# numpy views of the reference's row layouts
DRAW_DT = np.dtype({
    "names": ["mesh", "geometry", "submesh_index", "material", "instance_count",
              "first_instance", "sort_key"],
    "formats": ["2<u4", "2<u4", "<u4", "2<u4", "<u4", "<u4", "<u8"],
    "offsets": [0, 8, 16, 20, 28, 32, 40], "itemsize": 48})       # VkrDrawItem
INSTANCE_DT = np.dtype({
    "names": ["model", "object_id", "reserved"], "formats": ["16<f4", "<u4", "3<u4"],
    "offsets": [0, 64, 68], "itemsize": 80})                      # VkrInstanceDataGPU
GPU_ROW_DT = np.dtype({  # VkrGpuCandidateDrawRow, lib/src/renderer/vkr_gpu_abi.h:70-80
    "names": ["geometry_index", "material_index", "instance_index", "first_index", "index_count",
              "vertex_offset", "state_bucket", "flags", "sphere"],
    "formats": ["<u4", "<u4", "<u4", "<u4", "<u4", "<i4", "<u4", "<u4", "4<f4"],
    "offsets": [0, 4, 8, 12, 16, 20, 24, 28, 32], "itemsize": 48})
VISIBLE_ROW_DT = np.dtype({  # VkrGpuVisibleDrawRow, lib/src/renderer/vkr_gpu_abi.h:83-92
    "names": ["geometry_index", "material_index", "instance_index", "first_index", "index_count",
              "vertex_offset", "state_bucket", "flags"],
    "formats": ["<u4", "<u4", "<u4", "<u4", "<u4", "<i4", "<u4", "<u4"],
    "offsets": [0, 4, 8, 12, 16, 20, 24, 28], "itemsize": 32})
COMMAND_DT = np.dtype({  # VkrVkIndirectCommand, deferred.slang:21-28
    "names": ["index_count", "instance_count", "first_index", "vertex_offset", "first_instance"],
    "formats": ["<u4", "<u4", "<u4", "<i4", "<u4"], "offsets": [0, 4, 8, 12, 16], "itemsize": 20})
GEOMETRY_DT = np.dtype({  # vkrhip_geometry_entry
    "names": ["generation", "resolvable", "first_index", "submesh_count", "submesh_offset"],
    "formats": ["<u4"] * 5, "offsets": [0, 4, 8, 12, 16], "itemsize": 32})
SUBMESH_DT = np.dtype({  # vkrhip_submesh_range
    "names": ["first_index", "index_count", "vertex_offset"],
    "formats": ["<u4", "<u4", "<i4"], "offsets": [0, 4, 8], "itemsize": 16})
MATERIAL_DT = np.dtype({  # vkrhip_material_entry
    "names": ["generation", "live", "slot_index"],
    "formats": ["<u4"] * 3, "offsets": [0, 4, 8], "itemsize": 16})
CANDIDATE_DT = np.dtype({
    "names": ["mesh", "geometry", "submesh_index", "material", "model", "object_id",
              "reserved", "sphere", "state_bucket", "flags"],
    "formats": ["2<u4", "2<u4", "<u4", "2<u4", "16<f4", "<u4", "3<u4", "4<f4", "<u4", "<u4"],
    "offsets": [0, 8, 16, 20, 32, 96, 100, 112, 128, 132], "itemsize": 144})  # VkrWorldDrawCandidate


@dataclass
class ViewState:
    """VkrGpuDrawCompactionState of one view (lib/src/renderer/vkr_gpu_abi.h:95-103)."""
    execution_ranges: np.ndarray  # (4, 2) {offset, count}
    bucket_counts: np.ndarray
    bucket_cursors: np.ndarray
    visible_count: int
    overflow_count: int
    near_plane_ties: int
    occlusion_culled_count: int = 0


@dataclass
class FrameResult:
    objects_tested: int
    objects_culled_camera: int
    objects_without_bounds: int
    gpu_candidate_count: int
    gpu_camera_opaque_candidate_count: int
    transmission_gpu_candidate_count: int
    transparent_draw_count: int
    transparent_overflow: int
    nodes_updated: int
    views: list
    packed_count: int = 0
    unpublished_geometry_count: int = 0
    unpublished_material_count: int = 0
    invalid_submesh_count: int = 0


def _check(lib, ctx, rc: int) -> None:
    if rc != N.OK:
        msg = lib.vkrhip_last_error(ctx)
        raise N.VkrHipError(rc, msg.decode() if msg else "")


class HipScene:
    """One GPU's HBM-resident scene and its per-frame processing path."""

    def __init__(self, max_nodes: int, max_candidates: int, max_views: int = 1, device: int = 0,
                 partition_capacity: int = 0, visible_capacity: int = 0,
                 transparent_capacity: int = 0, source_index_base: int = 0,
                 tie_epsilon: float = 0.0):
        self.lib = N.load()
        cfg = N.Config()
        cfg.struct_size = C.sizeof(N.Config)
        cfg.device = device
        cfg.max_nodes = max_nodes
        cfg.max_candidates = max_candidates
        cfg.max_views = max_views
        cfg.partition_capacity = partition_capacity
        cfg.visible_capacity = visible_capacity
        cfg.transparent_capacity = transparent_capacity
        cfg.source_index_base = source_index_base
        cfg.tie_epsilon = tie_epsilon
        cfg.flags = 0
        self.ctx = C.c_void_p()
        rc = self.lib.vkrhip_create(C.byref(cfg), C.byref(self.ctx))
        if rc != N.OK:
            msg = self.lib.vkrhip_last_error(None)
            raise N.VkrHipError(rc, msg.decode() if msg else "")
        self.max_nodes = max_nodes
        self.max_candidates = max_candidates
        self.max_views = max_views
        self.n_nodes = 0
        self.n_candidates = 0
        self.views = 0
        self.visible_capacity = visible_capacity or max_candidates
        self.frame_serial = 0  # frames enqueued so far (the exchange checks it is repeating the right one)

    # ---- lifetime -------------------------------------------------------
    def close(self) -> None:
        if getattr(self, "ctx", None) is not None and self.ctx:
            for address in getattr(self, "_host_buffers", []):
                self.lib.vkrhip_host_free(self.ctx, C.c_void_p(address))
            self._host_buffers = []
            self.lib.vkrhip_destroy(self.ctx)
            self.ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def _ok(self, rc: int) -> None:
        _check(self.lib, self.ctx, rc)

    def set_stream(self, hip_stream: int | None) -> None:
        self._ok(self.lib.vkrhip_set_stream(self.ctx, hip_stream))

    def stream_handle(self) -> int:
        """hipStream_t the context enqueues on (its own stream by default)."""
        return int(self.lib.vkrhip_get_stream(self.ctx) or 0)

    def join(self) -> None:
        """Orders the context's stream behind the frame's transparent sort and
        emission when they ran on the second stream (vkrhip_join)."""
        self._ok(self.lib.vkrhip_join(self.ctx))

    def sync(self) -> None:
        self._ok(self.lib.vkrhip_sync(self.ctx))

    # ---- scene state ----------------------------------------------------
    def set_hierarchy(self, parent, level_offsets) -> None:
        level_offsets = np.ascontiguousarray(level_offsets, dtype=np.uint64)
        levels = level_offsets.shape[0] - 1
        n = int(level_offsets[-1]) if levels >= 0 and level_offsets.shape[0] else 0
        p = None if parent is None else N.as_u32(parent)
        self._ok(self.lib.vkrhip_set_hierarchy(self.ctx, N.ptr(p), n, N.ptr(level_offsets),
                                               max(levels, 0)))
        self.n_nodes = n

    def upload_trs(self, pos4, quat4, scale4, first: int = 0) -> None:
        pos4, quat4, scale4 = (N.as_f32(a).reshape(-1, 4) for a in (pos4, quat4, scale4))
        self._ok(self.lib.vkrhip_upload_trs(self.ctx, first, pos4.shape[0], N.ptr(pos4),
                                            N.ptr(quat4), N.ptr(scale4)))

    def upload_locals(self, local16, first: int = 0) -> None:
        local16 = N.as_f32(local16).reshape(-1, 16)
        self._ok(self.lib.vkrhip_upload_locals(self.ctx, first, local16.shape[0], N.ptr(local16)))

    def upload_worlds(self, world16, first: int = 0) -> None:
        world16 = N.as_f32(world16).reshape(-1, 16)
        self._ok(self.lib.vkrhip_upload_worlds(self.ctx, first, world16.shape[0], N.ptr(world16)))

    def upload_node_flags(self, flags, first: int = 0) -> None:
        flags = np.ascontiguousarray(flags, dtype=np.uint8)
        self._ok(self.lib.vkrhip_upload_node_flags(self.ctx, first, flags.shape[0], N.ptr(flags)))

    def mark_all(self, flags: int = N.DIRTY_WORLD) -> None:
        self._ok(self.lib.vkrhip_mark_all(self.ctx, flags))

    def mark_nodes(self, first: int, count: int, flags: int = N.DIRTY_WORLD) -> None:
        """ORs `flags` into nodes [first, first + count) on the device, in stream order."""
        self._ok(self.lib.vkrhip_mark_nodes(self.ctx, first, count, flags))

    # ---- candidates / views --------------------------------------------
    def set_candidates(self, node, ident, matbits, bounds) -> None:
        ident = N.as_u32(ident).reshape(-1, 8)
        n = ident.shape[0]
        matbits = N.as_u32(matbits)
        bounds = N.as_f32(bounds).reshape(-1, 12)
        if matbits.shape[0] != n or bounds.shape[0] != n:
            raise ValueError("candidate arrays disagree on the row count")
        nd = None if node is None else N.as_u32(node)
        if nd is not None and nd.shape[0] != n:
            raise ValueError("node array disagrees on the row count")
        self._ok(self.lib.vkrhip_set_candidates(self.ctx, n, N.ptr(nd), N.ptr(ident),
                                                N.ptr(matbits), N.ptr(bounds)))
        self.n_candidates = n

    def set_views(self, planes, required_flags, camera_view) -> None:
        planes = N.as_f32(planes).reshape(-1, 24)
        required = N.as_u32(required_flags)
        view = N.as_f32(camera_view).reshape(16)
        if required.shape[0] != planes.shape[0]:
            raise ValueError("planes and required_flags disagree on the view count")
        self._ok(self.lib.vkrhip_set_views(self.ctx, planes.shape[0], N.ptr(planes), N.ptr(required),
                                           N.ptr(view)))
        self.views = planes.shape[0]

    def set_publication(self, geometries, submeshes, materials) -> None:
        """Backend publication tables (structured arrays of GEOMETRY_DT, SUBMESH_DT,
        MATERIAL_DT): entry k describes handle id k + 1. Enables STAGE_PACK /
        STAGE_ENCODE; the visible lists hold packed row indices from then on."""
        g = np.ascontiguousarray(geometries, dtype=GEOMETRY_DT)
        sm = np.ascontiguousarray(submeshes, dtype=SUBMESH_DT)
        m = np.ascontiguousarray(materials, dtype=MATERIAL_DT)
        self._ok(self.lib.vkrhip_set_publication(self.ctx, N.ptr(g), g.shape[0], N.ptr(sm),
                                                 sm.shape[0], N.ptr(m), m.shape[0]))

    # ---- the frame ------------------------------------------------------
    def frame(self, stages: int = N.STAGE_ALL) -> None:
        self._ok(self.lib.vkrhip_frame(self.ctx, stages))
        self.frame_serial += 1

    def frame_info(self) -> FrameResult:
        info = N.FrameInfo()
        self._ok(self.lib.vkrhip_get_frame_info(self.ctx, C.byref(info)))
        views = []
        for v in range(info.view_count):
            st = info.views[v]
            views.append(ViewState(
                execution_ranges=np.array([[st.execution_ranges[b][0], st.execution_ranges[b][1]]
                                           for b in range(4)], np.uint32),
                bucket_counts=np.array(list(st.bucket_counts), np.uint32),
                bucket_cursors=np.array(list(st.bucket_cursors), np.uint32),
                visible_count=int(st.visible_count), overflow_count=int(st.overflow_count),
                near_plane_ties=int(info.near_plane_ties[v]),
                occlusion_culled_count=int(st.occlusion_culled_count)))
        return FrameResult(
            objects_tested=info.stats.objects_tested,
            objects_culled_camera=info.stats.objects_culled_camera,
            objects_without_bounds=info.stats.objects_without_bounds,
            gpu_candidate_count=info.gpu_candidate_count,
            gpu_camera_opaque_candidate_count=info.gpu_camera_opaque_candidate_count,
            transmission_gpu_candidate_count=info.transmission_gpu_candidate_count,
            transparent_draw_count=info.transparent_draw_count,
            transparent_overflow=info.transparent_overflow,
            nodes_updated=info.nodes_updated, views=views, packed_count=info.packed_count,
            unpublished_geometry_count=info.unpublished_geometry_count,
            unpublished_material_count=info.unpublished_material_count,
            invalid_submesh_count=info.invalid_submesh_count)

    def device_outputs(self) -> N.DeviceOutputs:
        out = N.DeviceOutputs()
        self._ok(self.lib.vkrhip_get_device_outputs(self.ctx, C.byref(out)))
        return out

    # ---- read-back ------------------------------------------------------
    def read_world(self, first: int = 0, count: int | None = None) -> np.ndarray:
        count = self.n_nodes - first if count is None else count
        out = np.empty((count, 16), np.float32)
        self._ok(self.lib.vkrhip_read_world(self.ctx, first, count, N.ptr(out)))
        return out

    def read_locals(self, first: int = 0, count: int | None = None) -> np.ndarray:
        count = self.n_nodes - first if count is None else count
        out = np.empty((count, 16), np.float32)
        self._ok(self.lib.vkrhip_read_locals(self.ctx, first, count, N.ptr(out)))
        return out

    def read_node_flags(self, first: int = 0, count: int | None = None) -> np.ndarray:
        count = self.n_nodes - first if count is None else count
        out = np.empty(count, np.uint8)
        self._ok(self.lib.vkrhip_read_node_flags(self.ctx, first, count, N.ptr(out)))
        return out

    def read_dirty_list(self) -> np.ndarray:
        """Nodes the last frame's update rewrote, in topological order (STAGE_DIRTY_LIST)."""
        count = C.c_uint64(0)
        out = np.empty(self.n_nodes, np.uint32)
        self._ok(self.lib.vkrhip_read_dirty_list(self.ctx, out.shape[0], N.ptr(out), C.byref(count)))
        return out[: int(count.value)].copy()

    def read_visible(self, view: int, count: int) -> np.ndarray:
        out = np.empty(count, np.uint32)
        self._ok(self.lib.vkrhip_read_visible(self.ctx, view, count, N.ptr(out)))
        return out

    def read_transparent_visible(self) -> np.ndarray:
        out = np.empty(self.n_candidates, np.uint8)
        self._ok(self.lib.vkrhip_read_transparent_visible(self.ctx, 0, self.n_candidates, N.ptr(out)))
        return out

    def read_transparent(self, count: int):
        draws = np.zeros(count, DRAW_DT)
        inst = np.zeros(count, INSTANCE_DT)
        keys = np.zeros(count, np.uint64)
        src = np.zeros(count, np.uint32)
        self._ok(self.lib.vkrhip_read_transparent(self.ctx, count, N.ptr(draws), N.ptr(inst),
                                                  N.ptr(keys), N.ptr(src)))
        return draws, inst, keys, src

    def host_buffer(self, nbytes: int) -> np.ndarray:
        """Page-locked host bytes (vkrhip_host_alloc) as a uint8 array, for the
        destinations of read copies; released by host_buffer_free or close()."""
        p = C.c_void_p()
        self._ok(self.lib.vkrhip_host_alloc(self.ctx, max(int(nbytes), 1), C.byref(p)))
        self._host_buffers = getattr(self, "_host_buffers", [])
        self._host_buffers.append(p.value)
        return np.ctypeslib.as_array((C.c_uint8 * max(int(nbytes), 1)).from_address(p.value))

    def host_buffers_free(self) -> None:
        """Releases every host_buffer() of this scene (their arrays become invalid)."""
        for address in getattr(self, "_host_buffers", []):
            self._ok(self.lib.vkrhip_host_free(self.ctx, C.c_void_p(address)))
        self._host_buffers = []

    def read_candidates(self) -> np.ndarray:
        out = np.zeros(self.n_candidates, CANDIDATE_DT)
        self._ok(self.lib.vkrhip_read_candidates(self.ctx, 0, self.n_candidates, N.ptr(out)))
        return out

    def read_transmission_candidates(self, count: int) -> np.ndarray:
        out = np.zeros(count, CANDIDATE_DT)
        if count:
            self._ok(self.lib.vkrhip_read_transmission_candidates(self.ctx, 0, count, N.ptr(out)))
        return out

    def read_packed(self, count: int):
        rows = np.zeros(count, GPU_ROW_DT)
        inst = np.zeros(count, INSTANCE_DT)
        self._ok(self.lib.vkrhip_read_packed(self.ctx, count, N.ptr(rows), N.ptr(inst)))
        return rows, inst

    def read_visible_rows(self, view: int, count: int) -> np.ndarray:
        out = np.zeros(count, VISIBLE_ROW_DT)
        if count:
            self._ok(self.lib.vkrhip_read_visible_rows(self.ctx, view, count, N.ptr(out)))
        return out

    def read_commands(self, view: int, bucket: int, count: int) -> np.ndarray:
        out = np.zeros(count, COMMAND_DT)
        if count:
            self._ok(self.lib.vkrhip_read_commands(self.ctx, view, bucket, count, N.ptr(out)))
        return out

    def read_world_bounds(self) -> np.ndarray:
        out = np.empty((self.n_candidates, 4), np.float32)
        self._ok(self.lib.vkrhip_read_world_bounds(self.ctx, 0, self.n_candidates, N.ptr(out)))
        return out

    def caster_bounds(self):
        """(min (3,), max (3,)) of the box around the shadow casters' world spheres
        (the last frame ran with STAGE_CASTER_BOUNDS); min > max when there is none."""
        lo, hi = np.empty(3, np.float32), np.empty(3, np.float32)
        self._ok(self.lib.vkrhip_get_caster_bounds(self.ctx, N.ptr(lo), N.ptr(hi)))
        return lo, hi

    def read_candidate_prep(self):
        sphere = np.empty((self.n_candidates, 4), np.float32)
        flags = np.empty(self.n_candidates, np.uint32)
        self._ok(self.lib.vkrhip_read_candidate_prep(self.ctx, 0, self.n_candidates, N.ptr(sphere),
                                                     N.ptr(flags)))
        return sphere, flags

    # ---- timing ---------------------------------------------------------
    def set_profiling(self, stages=None) -> None:
        """stages: None/False = off, True = all, or an iterable of timer names."""
        if not stages:
            mask = 0
        elif stages is True:
            mask = (1 << len(N.TIMER_NAMES)) - 1
        else:
            mask = 0
            for name in stages:
                mask |= 1 << N.TIMER_NAMES.index(name)
        self._ok(self.lib.vkrhip_set_profiling(self.ctx, mask))

    def timers(self, reset: bool = False) -> dict:
        ms = np.zeros(len(N.TIMER_NAMES), np.float64)
        n = np.zeros(len(N.TIMER_NAMES), np.uint64)
        self._ok(self.lib.vkrhip_get_timers(self.ctx, N.ptr(ms), N.ptr(n), int(reset)))
        return {name: (float(ms[i]), int(n[i])) for i, name in enumerate(N.TIMER_NAMES)}

    def merge_runs(self, run_ptrs, lengths, out_ptr: int) -> None:
        g = len(run_ptrs)
        ptrs = (C.c_void_p * g)(*run_ptrs)
        lens = np.ascontiguousarray(lengths, dtype=np.uint64)
        self._ok(self.lib.vkrhip_merge_runs(self.ctx, g, C.cast(ptrs, C.c_void_p), N.ptr(lens),
                                            out_ptr))


    def concat_ranges(self, src_ptr: int, dst_ptr: int, desc) -> None:
        desc = np.ascontiguousarray(desc, dtype=np.uint64).reshape(-1, 3)
        self._ok(self.lib.vkrhip_concat_ranges(self.ctx, src_ptr, dst_ptr, N.ptr(desc),
                                               desc.shape[0]))


    # ---- the exchange inside the library, over RCCL --------------------------
    def comm_unique_id(self) -> np.ndarray:
        out = np.zeros(128, np.uint8)
        rc = self.lib.vkrhip_comm_unique_id(N.ptr(out))
        if rc != N.OK:
            msg = self.lib.vkrhip_last_error(None)
            raise N.VkrHipError(rc, msg.decode() if msg else "")
        return out

    def comm_init(self, unique_id, world: int, rank: int) -> None:
        uid = np.ascontiguousarray(unique_id, dtype=np.uint8)
        assert uid.shape == (128,)
        self._ok(self.lib.vkrhip_comm_init(self.ctx, N.ptr(uid), world, rank))

    def comm_count(self) -> int:
        """Ranks of the context's RCCL communicator as RCCL reports them."""
        out = C.c_uint32(0)
        self._ok(self.lib.vkrhip_comm_count(self.ctx, C.byref(out)))
        return int(out.value)

    def exchange_configure(self, world: int, vis_cap: int, key_cap: int) -> None:
        self._ok(self.lib.vkrhip_exchange_configure(self.ctx, world, vis_cap, key_cap))

    def exchange_run(self) -> None:
        self._ok(self.lib.vkrhip_exchange_run(self.ctx))

    def exchange_finish(self, views: int):
        """(result words, merged visible device pointer, merged keys device pointer)."""
        res = np.zeros(self.exchange_result_words(views), np.uint64)
        vis, keys = C.c_void_p(), C.c_void_p()
        self._ok(self.lib.vkrhip_exchange_finish(self.ctx, N.ptr(res), C.byref(vis), C.byref(keys)))
        return res, int(vis.value or 0), int(keys.value or 0)

    def set_hzb(self, mips, width: int = 0, height: int = 0, view_projection=None,
                depth_epsilon: float = 0.0) -> None:
        """Depth pyramid of the camera view (deferred.slang:240-303): `mips` is a list
        of 2-D float32 arrays, mip 0 first; None switches the occlusion test off."""
        if mips is None:
            self._ok(self.lib.vkrhip_set_hzb(self.ctx, None, 0, 0, 0, None, 0.0))
            return
        flat = np.ascontiguousarray(np.concatenate([N.as_f32(m).reshape(-1) for m in mips]))
        vp = N.as_f32(view_projection).reshape(16)
        self._ok(self.lib.vkrhip_set_hzb(self.ctx, N.ptr(flat), width, height, len(mips), N.ptr(vp),
                                         depth_epsilon))

    def build_hzb(self, depth, width: int, height: int, mip_count: int, view_projection,
                  depth_epsilon: float = 0.0) -> None:
        """Pyramid from a depth image (vk_hzb_build, deferred.slang:970-998), then as set_hzb."""
        depth = N.as_f32(depth)
        assert depth.ndim == 2
        vp = N.as_f32(view_projection).reshape(16)
        self._ok(self.lib.vkrhip_build_hzb(self.ctx, N.ptr(depth), depth.shape[1], depth.shape[0], width,
                                           height, mip_count, N.ptr(vp), depth_epsilon))
        self._hzb_extent = (width, height)

    def read_hzb(self, mip: int) -> np.ndarray:
        w, h = (max(e >> mip, 1) for e in self._hzb_extent)
        out = np.empty((h, w), np.float32)
        self._ok(self.lib.vkrhip_read_hzb(self.ctx, mip, N.ptr(out)))
        return out

    def candidate_epoch(self):
        """(hash, occluder_count) of vkr_vk_candidate_epoch
        (lib/src/renderer/vulkan/vkr_vulkan_draws.c:19-44) for the current candidates
        and world matrices, computed on the device."""
        h, k = C.c_uint64(0), C.c_uint32(0)
        self._ok(self.lib.vkrhip_candidate_epoch(self.ctx, C.byref(h), C.byref(k)))
        return int(h.value), int(k.value)

    # ---- multi-GPU exchange planned on the device --------------------------
    def exchange_block_words(self, views: int, vis_cap: int, key_cap: int) -> int:
        return int(self.lib.vkrhip_exchange_block_words(views, vis_cap, key_cap))

    def exchange_result_words(self, views: int) -> int:
        return int(self.lib.vkrhip_exchange_result_words(views))

    def exchange_pack(self, block_ptr: int, vis_cap: int, key_cap: int) -> None:
        self._ok(self.lib.vkrhip_exchange_pack(self.ctx, block_ptr, vis_cap, key_cap))

    def exchange_merge(self, gathered_ptr: int, world: int, vis_cap: int, key_cap: int,
                       merged_visible_ptr: int, merged_keys_ptr: int, result_ptr: int,
                       stream: int | None = None) -> None:
        self._ok(self.lib.vkrhip_exchange_merge(self.ctx, gathered_ptr, world, vis_cap, key_cap,
                                                merged_visible_ptr, merged_keys_ptr, result_ptr,
                                                stream))


# ---- host-side leaf mirrors ---------------------------------------------
def frustum_from_matrix(view_projection) -> np.ndarray:
    """vkr_frustum_from_matrix (lib/src/math/vkr_frustum.c:44-66)."""
    lib = N.load()
    vp = N.as_f32(view_projection).reshape(16)
    out = np.empty(24, np.float32)
    lib.vkrhip_frustum_from_matrix(N.ptr(vp), N.ptr(out))
    return out


def frustum_from_view_projection(view, projection) -> np.ndarray:
    """vkr_frustum_from_view_projection (lib/src/math/vkr_frustum.c:33-38)."""
    lib = N.load()
    v = N.as_f32(view).reshape(16)
    p = N.as_f32(projection).reshape(16)
    out = np.empty(24, np.float32)
    lib.vkrhip_frustum_from_view_projection(N.ptr(v), N.ptr(p), N.ptr(out))
    return out


def mat4_mul(a, b) -> np.ndarray:
    """mat4_mul (lib/src/math/mat.h:1145-1163)."""
    lib = N.load()
    a = N.as_f32(a).reshape(16)
    b = N.as_f32(b).reshape(16)
    out = np.empty(16, np.float32)
    lib.vkrhip_mat4_mul(N.ptr(a), N.ptr(b), N.ptr(out))
    return out


def build_view_set(view, projection, light_view_proj=None):
    """Camera + cascade plane sets and required flags
    (lib/src/renderer/vulkan/vkr_vulkan_deferred.c:257-273,292-294)."""
    lib = N.load()
    v = N.as_f32(view).reshape(16)
    p = N.as_f32(projection).reshape(16)
    lvp = np.zeros((0, 16), np.float32) if light_view_proj is None else \
        N.as_f32(light_view_proj).reshape(-1, 16)
    cascades = lvp.shape[0]
    planes = np.empty((1 + cascades, 24), np.float32)
    required = np.empty(1 + cascades, np.uint32)
    lib.vkrhip_build_view_set(N.ptr(v), N.ptr(p), N.ptr(lvp) if cascades else None, cascades,
                              N.ptr(planes), N.ptr(required))
    return planes, required


def levelize(parent):
    """BFS relabelling of an arbitrary parent array
    (scene_rebuild_topo_order, lib/src/renderer/systems/vkr_scene_system.c:777-843).
    Returns (order, new_parent, level_offsets)."""
    lib = N.load()
    parent = N.as_u32(parent)
    n = parent.shape[0]
    order = np.empty(n, np.uint32)
    new_parent = np.empty(n, np.uint32)
    offsets = np.zeros(n + 2, np.uint64)
    levels = lib.vkrhip_levelize(N.ptr(parent), n, N.ptr(order), N.ptr(new_parent), N.ptr(offsets))
    if n and not levels:
        raise MemoryError("vkrhip_levelize failed")
    return order, new_parent, offsets[: levels + 1].copy()


def shard_hierarchy(parent, level_offsets, shards: int, replicated_levels: int = 1) -> np.ndarray:
    """Per-shard node ranges of a BFS-ordered hierarchy cut into whole subtrees
    (vkrhip_shard_hierarchy): array (shards, levels, 2) of [begin, end)."""
    lib = N.load()
    parent = N.as_u32(parent)
    level_offsets = np.ascontiguousarray(level_offsets, dtype=np.uint64)
    levels = level_offsets.shape[0] - 1
    out = np.zeros((shards, levels, 2), np.uint64)
    rc = lib.vkrhip_shard_hierarchy(N.ptr(parent), parent.shape[0], N.ptr(level_offsets), levels,
                                    shards, replicated_levels, N.ptr(out))
    if rc != N.OK:
        raise N.VkrHipError(rc, "the hierarchy is not in BFS order with ascending parents"
                            if rc == N.ERR_STATE else "bad argument")
    return out


def shard_nodes(ranges_of_shard) -> np.ndarray:
    """Global node indices of one shard (its ranges level by level) - the shard's
    local numbering is the position in this array."""
    return np.concatenate([np.arange(int(b), int(e), dtype=np.uint64) for b, e in ranges_of_shard])


def shard_local_hierarchy(parent, ranges_of_shard):
    """(global node index of every local node, local parent array, local level
    offsets) of one shard."""
    parent = N.as_u32(parent)
    nodes = shard_nodes(ranges_of_shard)
    sizes = np.array([int(e) - int(b) for b, e in ranges_of_shard], np.uint64)
    offsets = np.concatenate([[0], np.cumsum(sizes)]).astype(np.uint64)
    local_parent = np.full(nodes.shape[0], 0xFFFFFFFF, np.uint32)
    for l in range(1, len(sizes)):
        lb, le = int(offsets[l]), int(offsets[l + 1])
        if le == lb:
            continue
        gp = parent[nodes[lb:le]].astype(np.int64)
        pb = int(ranges_of_shard[l - 1][0])
        local_parent[lb:le] = (gp - pb + int(offsets[l - 1])).astype(np.uint32)
    return nodes, local_parent, offsets


def effective_visibility(parent, spec) -> np.ndarray:
    """scene_entity_is_visible (lib/src/renderer/systems/vkr_scene_system.c:1935-1977)
    for every node of a parents-first hierarchy. spec: bit 0 has a SceneVisibility,
    bit 1 visible, bit 2 inherit_parent. Returns 0/1 bytes."""
    lib = N.load()
    parent = N.as_u32(parent)
    spec = np.ascontiguousarray(spec, dtype=np.uint8)
    out = np.empty(parent.shape[0], np.uint8)
    lib.vkrhip_effective_visibility(N.ptr(parent), N.ptr(spec), parent.shape[0], N.ptr(out))
    return out
