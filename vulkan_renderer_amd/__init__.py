"""MI355X-native scene processing for vulkan_renderer's per-frame path:
world-transform propagation -> multi-view frustum cull -> draw-key sort.

The compute path is hand-written HIP for gfx950 behind the C ABI declared in
include/vkr_hip_scene.h (libvkrhip.so, built in-tree by vulkan_renderer_amd.build).
"""
from . import _native  # noqa: F401
from .scene import (CANDIDATE_DT, DRAW_DT, INSTANCE_DT, FrameResult, HipScene, ViewState,  # noqa: F401
                    build_view_set, frustum_from_matrix, frustum_from_view_projection, levelize,
                    effective_visibility,
                    mat4_mul, shard_hierarchy, shard_local_hierarchy, shard_nodes)

__all__ = ["HipScene", "FrameResult", "ViewState", "build_view_set", "frustum_from_matrix",
           "frustum_from_view_projection", "levelize", "effective_visibility", "mat4_mul", "shard_hierarchy",
           "shard_local_hierarchy", "shard_nodes", "DRAW_DT", "INSTANCE_DT",
           "CANDIDATE_DT"]
