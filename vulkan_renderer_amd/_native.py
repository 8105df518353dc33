"""ctypes binding of libvkrhip.so (the C ABI in include/vkr_hip_scene.h).

There is no CPU fallback: a missing library is an error, raised at load time.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(PKG_DIR, "libvkrhip.so")

MAX_VIEWS = 64
NO_PARENT = 0xFFFFFFFF

OK = 0
ERR_ARG, ERR_HIP, ERR_CAPACITY, ERR_STATE = -1, -2, -3, -4

DIRTY_LOCAL, DIRTY_WORLD, WORLD_UPDATED = 0x01, 0x02, 0x08
CAND_BOUNDS_VALID, CAND_CAMERA_OPAQUE, CAND_SHADOW_CASTER = 0x1, 0x2, 0x4
MAT_ALPHA_MASK, MAT_TRANSMISSIVE, MAT_DOUBLE_SIDED, MAT_BOUNDS_VALID = 0x3, 0x4, 0x8, 0x10
ALPHA_OPAQUE, ALPHA_CUTOUT, ALPHA_BLEND = 0, 1, 2

STAGE_TRANSFORM, STAGE_CULL, STAGE_TRANSPARENT = 0x1, 0x2, 0x4
STAGE_CANDIDATES, STAGE_WORLD_BOUNDS = 0x8, 0x10
STAGE_PACK, STAGE_ENCODE = 0x20, 0x40
STAGE_CASTER_BOUNDS = 0x80
STAGE_DIRTY_LIST = 0x100
STAGE_ALL = 0x7

TIMER_NAMES = ["transform", "classify", "fused", "scan", "scatter", "sort", "emit",
               "candidates", "frame", "exchange_pack", "exchange_gather", "exchange_merge"]

EXPORTED_SYMBOLS = [
    "vkrhip_create", "vkrhip_destroy", "vkrhip_last_error", "vkrhip_set_stream", "vkrhip_get_stream", "vkrhip_join", "vkrhip_sync",
    "vkrhip_set_hierarchy", "vkrhip_upload_trs", "vkrhip_upload_locals", "vkrhip_upload_worlds",
    "vkrhip_upload_node_flags", "vkrhip_mark_all", "vkrhip_mark_nodes", "vkrhip_set_candidates", "vkrhip_set_views",
    "vkrhip_frame", "vkrhip_get_frame_info", "vkrhip_get_device_outputs", "vkrhip_read_world",
    "vkrhip_read_locals", "vkrhip_read_node_flags", "vkrhip_read_visible",
    "vkrhip_read_transparent_visible", "vkrhip_read_transparent", "vkrhip_read_candidates",
    "vkrhip_read_world_bounds", "vkrhip_read_dirty_list", "vkrhip_read_candidate_prep", "vkrhip_set_publication",
    "vkrhip_read_packed", "vkrhip_read_visible_rows", "vkrhip_read_commands", "vkrhip_set_profiling",
    "vkrhip_get_timers", "vkrhip_merge_runs", "vkrhip_concat_ranges", "vkrhip_frustum_from_matrix",
    "vkrhip_exchange_block_words", "vkrhip_exchange_result_words", "vkrhip_exchange_pack",
    "vkrhip_exchange_merge", "vkrhip_comm_unique_id", "vkrhip_comm_init", "vkrhip_comm_count", "vkrhip_exchange_configure",
    "vkrhip_exchange_run", "vkrhip_exchange_finish", "vkrhip_debug_streams", "vkrhip_candidate_epoch", "vkrhip_set_hzb", "vkrhip_build_hzb", "vkrhip_read_hzb", "vkrhip_read_transmission_candidates",
    "vkrhip_frustum_from_view_projection", "vkrhip_mat4_mul", "vkrhip_build_view_set",
    "vkrhip_levelize", "vkrhip_effective_visibility",
    "vkrhip_static_init", "vkrhip_exchange_merge_static",
    "vkrhip_fill_world_payload", "vkrhip_shard_hierarchy",
    "vkrhip_shadow_config_high", "vkrhip_shadow_system_init",
    "vkrhip_shadow_system_invalidate_fit_history", "vkrhip_shadow_system_update",
    "vkrhip_shadow_fit_cascade", "vkrhip_shadow_fit_relevant_caster_z",
    "vkrhip_shadow_quantize_extent_up", "vkrhip_shadow_apply_fit_hysteresis",
    "vkrhip_shadow_build_view_set", "vkrhip_get_caster_bounds", "vkrhip_gltf_read_nodes", "vkrhip_gltf_read_mesh_bounds",
    "vkrhip_host_alloc", "vkrhip_host_free", "vkrhip_scene_read_entities",
]


class Config(C.Structure):
    _fields_ = [
        ("struct_size", C.c_uint32),
        ("device", C.c_int32),
        ("max_nodes", C.c_uint64),
        ("max_candidates", C.c_uint64),
        ("max_views", C.c_uint32),
        ("partition_capacity", C.c_uint32),
        ("visible_capacity", C.c_uint64),
        ("transparent_capacity", C.c_uint64),
        ("source_index_base", C.c_uint64),
        ("tie_epsilon", C.c_float),
        ("flags", C.c_uint32),
    ]


class ShadowFit(C.Structure):  # vkrhip_shadow_fit
    _fields_ = [(n, C.c_float) for n in ("center_x", "center_y", "extent", "min_z", "max_z",
                                         "world_units_per_texel")]

    def tuple(self):
        import numpy as np
        return tuple(np.float32(getattr(self, n)) for n, _ in self._fields_)


class ShadowConfig(C.Structure):  # vkrhip_shadow_config
    _fields_ = [("cascade_count", C.c_uint32), ("shadow_map_size", C.c_uint32),
                ("cascade_split_lambda", C.c_float), ("max_shadow_distance", C.c_float),
                ("cascade_guard_band_texels", C.c_float),
                ("cascade_guard_band_texels_per", C.c_float * 8),
                ("z_extension_factor", C.c_float),
                ("cascade_z_extension_factor_per", C.c_float * 8),
                ("use_constant_cascade_size", C.c_uint32), ("anchor_snap_texels", C.c_float),
                ("stabilize_cascades", C.c_uint32), ("use_scene_bounds", C.c_uint32),
                ("scene_min", C.c_float * 3), ("scene_max", C.c_float * 3)]


class ShadowCamera(C.Structure):  # vkrhip_shadow_camera
    _fields_ = [("type", C.c_uint32), ("position", C.c_float * 3), ("forward", C.c_float * 3),
                ("right", C.c_float * 3), ("up", C.c_float * 3), ("near_clip", C.c_float),
                ("far_clip", C.c_float), ("zoom", C.c_float), ("window_width", C.c_uint32),
                ("window_height", C.c_uint32), ("left_clip", C.c_float), ("right_clip", C.c_float),
                ("bottom_clip", C.c_float), ("top_clip", C.c_float)]


class ShadowCascade(C.Structure):  # vkrhip_shadow_cascade
    _fields_ = [("view_projection", C.c_float * 16), ("split_far", C.c_float),
                ("world_units_per_texel", C.c_float), ("light_space_origin", C.c_float * 2),
                ("light_space_depth_span", C.c_float), ("bounds_center", C.c_float * 3),
                ("bounds_radius", C.c_float)]


class ShadowSystem(C.Structure):  # vkrhip_shadow_system
    _fields_ = [("config", ShadowConfig), ("cascades", ShadowCascade * 8),
                ("cascade_splits", C.c_float * 9), ("light_direction", C.c_float * 3),
                ("light_enabled", C.c_uint32), ("history_fits", ShadowFit * 8),
                ("history_light_direction", C.c_float * 3), ("history_cascade_count", C.c_uint32),
                ("history_shadow_map_size", C.c_uint32),
                ("history_projection_convention", C.c_uint32), ("history_valid", C.c_uint32),
                ("history_enable_generation", C.c_uint64), ("enable_generation", C.c_uint64),
                ("initialized", C.c_uint32), ("reserved", C.c_uint32)]


class GltfNodes(C.Structure):  # vkrhip_gltf_nodes
    _fields_ = [("node_count", C.c_uint32), ("scene_count", C.c_uint32), ("scene", C.c_uint32),
                ("scene_nodes", C.c_uint32)]


class GltfMeshes(C.Structure):  # vkrhip_gltf_meshes
    _fields_ = [("mesh_count", C.c_uint32), ("primitive_count", C.c_uint32), ("accessor_count", C.c_uint32),
                ("bounded_count", C.c_uint32)]


class SceneEntities(C.Structure):
    _fields_ = [("version", C.c_uint32), ("entity_count", C.c_uint32), ("renderable_count", C.c_uint32)]


class VisibilityStats(C.Structure):
    _fields_ = [("objects_tested", C.c_uint32), ("objects_culled_camera", C.c_uint32),
                ("objects_without_bounds", C.c_uint32)]


class CompactionState(C.Structure):
    _fields_ = [
        ("execution_ranges", (C.c_uint32 * 2) * 4),
        ("bucket_counts", C.c_uint32 * 4),
        ("bucket_cursors", C.c_uint32 * 4),
        ("visible_count", C.c_uint32),
        ("overflow_count", C.c_uint32),
        ("resolve_invalid_count", C.c_uint32),
        ("occlusion_culled_count", C.c_uint32),
    ]


class FrameInfo(C.Structure):
    _fields_ = [
        ("stats", VisibilityStats),
        ("gpu_candidate_count", C.c_uint32),
        ("gpu_camera_opaque_candidate_count", C.c_uint32),
        ("transmission_gpu_candidate_count", C.c_uint32),
        ("transparent_draw_count", C.c_uint32),
        ("transparent_overflow", C.c_uint32),
        ("nodes_updated", C.c_uint32),
        ("view_count", C.c_uint32),
        ("reserved", C.c_uint32),
        ("packed_count", C.c_uint32),
        ("unpublished_geometry_count", C.c_uint32),
        ("unpublished_material_count", C.c_uint32),
        ("invalid_submesh_count", C.c_uint32),
        ("near_plane_ties", C.c_uint64 * MAX_VIEWS),
        ("views", CompactionState * MAX_VIEWS),
    ]


class WorldPayload(C.Structure):
    """vkrhip_world_payload = VkrWorldPassPayload (lib/src/renderer/vkr_render_packet.h:162-179)."""
    _fields_ = [
        ("gpu_candidates", C.c_void_p),
        ("gpu_candidate_count", C.c_uint32),
        ("gpu_camera_opaque_candidate_count", C.c_uint32),
        ("gpu_shadow_candidate_count", C.c_uint32),
        ("transmission_gpu_candidates", C.c_void_p),
        ("transmission_gpu_candidate_count", C.c_uint32),
        ("transparent_draws", C.c_void_p),
        ("transparent_draw_count", C.c_uint32),
        ("instances", C.c_void_p),
        ("instance_count", C.c_uint32),
        ("text_draws", C.c_void_p),
        ("text_draw_count", C.c_uint32),
    ]


class DeviceOutputs(C.Structure):
    _fields_ = [
        ("world", C.c_void_p),
        ("node_flags", C.c_void_p),
        ("visible", C.c_void_p),
        ("visible_stride", C.c_uint64),
        ("transparent_visible", C.c_void_p),
        ("transparent_draws", C.c_void_p),
        ("transparent_instances", C.c_void_p),
        ("transparent_keys", C.c_void_p),
        ("transparent_source", C.c_void_p),
        ("gpu_candidates", C.c_void_p),
        ("world_bounds", C.c_void_p),
        ("device_counts", C.c_void_p),
        ("packed_rows", C.c_void_p),
        ("packed_instances", C.c_void_p),
        ("visible_rows", C.c_void_p),
        ("commands", C.c_void_p),
        ("transmission_gpu_candidates", C.c_void_p),
        ("dirty_nodes", C.c_void_p),
        ("dirty_node_count", C.c_void_p),
    ]


class VkrHipError(RuntimeError):
    def __init__(self, code: int, message: str):
        super().__init__(f"vkrhip error {code}: {message}")
        self.code = code


_lib = None


def load(path: str = LIB_PATH) -> C.CDLL:
    """Loads libvkrhip.so and declares every entry point. Raises if it is absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(path):
        raise ImportError(
            f"{path} is missing: build it with `python -m vulkan_renderer_amd.build` "
            "(hipcc --offload-arch=gfx950). There is no CPU fallback.")
    L = C.CDLL(path)
    vp, u64, u32 = C.c_void_p, C.c_uint64, C.c_uint32
    L.vkrhip_create.argtypes = [C.POINTER(Config), C.POINTER(vp)]
    L.vkrhip_destroy.argtypes = [vp]
    L.vkrhip_destroy.restype = None
    L.vkrhip_last_error.argtypes = [vp]
    L.vkrhip_last_error.restype = C.c_char_p
    L.vkrhip_set_stream.argtypes = [vp, vp]
    L.vkrhip_get_stream.argtypes = [vp]
    L.vkrhip_join.argtypes = [vp]
    L.vkrhip_get_stream.restype = vp
    L.vkrhip_sync.argtypes = [vp]
    L.vkrhip_set_hierarchy.argtypes = [vp, vp, u64, vp, u32]
    L.vkrhip_upload_trs.argtypes = [vp, u64, u64, vp, vp, vp]
    L.vkrhip_upload_locals.argtypes = [vp, u64, u64, vp]
    L.vkrhip_upload_worlds.argtypes = [vp, u64, u64, vp]
    L.vkrhip_upload_node_flags.argtypes = [vp, u64, u64, vp]
    L.vkrhip_mark_all.argtypes = [vp, C.c_uint8]
    L.vkrhip_set_candidates.argtypes = [vp, u64, vp, vp, vp, vp]
    L.vkrhip_set_views.argtypes = [vp, u32, vp, vp, vp]
    L.vkrhip_frame.argtypes = [vp, u32]
    L.vkrhip_get_frame_info.argtypes = [vp, C.POINTER(FrameInfo)]
    L.vkrhip_get_device_outputs.argtypes = [vp, C.POINTER(DeviceOutputs)]
    L.vkrhip_read_world.argtypes = [vp, u64, u64, vp]
    L.vkrhip_read_locals.argtypes = [vp, u64, u64, vp]
    L.vkrhip_read_node_flags.argtypes = [vp, u64, u64, vp]
    L.vkrhip_read_visible.argtypes = [vp, u32, u64, vp]
    L.vkrhip_read_transparent_visible.argtypes = [vp, u64, u64, vp]
    L.vkrhip_read_transparent.argtypes = [vp, u64, vp, vp, vp, vp]
    L.vkrhip_read_candidates.argtypes = [vp, u64, u64, vp]
    L.vkrhip_scene_read_entities.argtypes = [C.c_char_p, u64, u32, C.POINTER(SceneEntities), vp, vp, vp, vp,
                                             vp, vp, vp]
    L.vkrhip_host_alloc.argtypes = [vp, u64, C.POINTER(C.c_void_p)]
    L.vkrhip_host_free.argtypes = [vp, vp]
    L.vkrhip_read_world_bounds.argtypes = [vp, u64, u64, vp]
    L.vkrhip_read_candidate_prep.argtypes = [vp, u64, u64, vp, vp]
    L.vkrhip_set_publication.argtypes = [vp, vp, u32, vp, u32, vp, u32]
    L.vkrhip_read_packed.argtypes = [vp, u64, vp, vp]
    L.vkrhip_read_visible_rows.argtypes = [vp, u32, u64, vp]
    L.vkrhip_read_commands.argtypes = [vp, u32, u32, u64, vp]
    L.vkrhip_set_profiling.argtypes = [vp, u32]
    L.vkrhip_get_timers.argtypes = [vp, vp, vp, C.c_int]
    L.vkrhip_merge_runs.argtypes = [vp, u32, vp, vp, vp]
    L.vkrhip_concat_ranges.argtypes = [vp, vp, vp, vp, u32]
    L.vkrhip_comm_unique_id.argtypes = [vp]
    L.vkrhip_comm_init.argtypes = [vp, vp, u32, u32]
    L.vkrhip_comm_count.argtypes = [vp, vp]
    L.vkrhip_read_dirty_list.argtypes = [vp, u64, vp, vp]
    L.vkrhip_mark_nodes.argtypes = [vp, u64, u64, C.c_uint8]
    L.vkrhip_debug_streams.argtypes = [vp, vp, u32]
    L.vkrhip_exchange_configure.argtypes = [vp, u32, u32, u32]
    L.vkrhip_exchange_run.argtypes = [vp]
    L.vkrhip_exchange_finish.argtypes = [vp, vp, vp, vp]
    L.vkrhip_set_hzb.argtypes = [vp, vp, u32, u32, u32, vp, C.c_float]
    L.vkrhip_build_hzb.argtypes = [vp, vp, u32, u32, u32, u32, u32, vp, C.c_float]
    L.vkrhip_read_hzb.argtypes = [vp, u32, vp]
    L.vkrhip_candidate_epoch.argtypes = [vp, vp, vp]
    L.vkrhip_read_transmission_candidates.argtypes = [vp, u64, u64, vp]
    L.vkrhip_exchange_block_words.argtypes = [u32, u32, u32]
    L.vkrhip_exchange_block_words.restype = u64
    L.vkrhip_exchange_result_words.argtypes = [u32]
    L.vkrhip_exchange_result_words.restype = u64
    L.vkrhip_exchange_pack.argtypes = [vp, vp, u32, u32]
    L.vkrhip_exchange_merge.argtypes = [vp, vp, u32, u32, u32, vp, vp, vp, vp]
    L.vkrhip_fill_world_payload.argtypes = [vp, C.POINTER(WorldPayload), C.POINTER(VisibilityStats)]
    L.vkrhip_shard_hierarchy.argtypes = [vp, u64, vp, u32, u32, u32, vp]
    fitp, cfgp = C.POINTER(ShadowFit), C.POINTER(ShadowConfig)
    L.vkrhip_shadow_config_high.argtypes = [cfgp]
    L.vkrhip_shadow_config_high.restype = None
    L.vkrhip_shadow_system_init.argtypes = [C.POINTER(ShadowSystem), cfgp]
    L.vkrhip_shadow_system_invalidate_fit_history.argtypes = [C.POINTER(ShadowSystem)]
    L.vkrhip_shadow_system_invalidate_fit_history.restype = None
    L.vkrhip_shadow_system_update.argtypes = [C.POINTER(ShadowSystem), C.POINTER(ShadowCamera), u32, vp]
    L.vkrhip_shadow_fit_cascade.argtypes = [vp, vp, u32, C.c_int, C.c_float, C.c_int, cfgp, C.c_float,
                                            fitp, vp, fitp, vp]
    L.vkrhip_shadow_fit_cascade.restype = None
    L.vkrhip_shadow_fit_relevant_caster_z.argtypes = [vp, vp, vp] + [C.c_float] * 4 + [vp, vp]
    L.vkrhip_shadow_quantize_extent_up.argtypes = [C.c_float, u32]
    L.vkrhip_shadow_quantize_extent_up.restype = C.c_float
    L.vkrhip_shadow_apply_fit_hysteresis.argtypes = [fitp, fitp, fitp]
    L.vkrhip_shadow_apply_fit_hysteresis.restype = None
    L.vkrhip_shadow_build_view_set.argtypes = [C.POINTER(ShadowSystem), vp, vp, vp, vp, vp]
    L.vkrhip_get_caster_bounds.argtypes = [vp, vp, vp]
    L.vkrhip_gltf_read_nodes.argtypes = [C.c_char_p, u64, u32, u32, C.POINTER(GltfNodes)] + [vp] * 8
    L.vkrhip_gltf_read_mesh_bounds.argtypes = [C.c_char_p, u64, u32, C.POINTER(GltfMeshes)] + [vp] * 5
    L.vkrhip_static_init.argtypes = [C.c_int]
    L.vkrhip_exchange_merge_static.argtypes = [C.c_int, u32, vp, u32, u32, u32, vp, vp, vp, vp]
    L.vkrhip_frustum_from_matrix.argtypes = [vp, vp]
    L.vkrhip_frustum_from_matrix.restype = None
    L.vkrhip_frustum_from_view_projection.argtypes = [vp, vp, vp]
    L.vkrhip_frustum_from_view_projection.restype = None
    L.vkrhip_mat4_mul.argtypes = [vp, vp, vp]
    L.vkrhip_mat4_mul.restype = None
    L.vkrhip_build_view_set.argtypes = [vp, vp, vp, u32, vp, vp]
    L.vkrhip_build_view_set.restype = None
    L.vkrhip_levelize.argtypes = [vp, u64, vp, vp, vp]
    L.vkrhip_levelize.restype = u32
    L.vkrhip_effective_visibility.argtypes = [vp, vp, u64, vp]
    L.vkrhip_effective_visibility.restype = None
    _lib = L
    return L


def ptr(a: np.ndarray | None):
    return None if a is None else a.ctypes.data


def as_f32(a, shape=None):
    a = np.ascontiguousarray(a, dtype=np.float32)
    return a if shape is None else a.reshape(shape)


def as_u32(a, shape=None):
    a = np.ascontiguousarray(a, dtype=np.uint32)
    return a if shape is None else a.reshape(shape)
