"""Cascade fitting: the host step that produces the shadow views of the cull.

Mirror of the reference's shadow system for this path
(lib/src/renderer/systems/vkr_shadow_system.{h,c}): `ShadowSystem` wraps the
library's vkrhip_shadow_system_* entry points (include/vkr_hip_scene.h), which
restate vkr_shadow_system_init / _update / _invalidate_fit_history; the module
functions wrap the reference's public leaf functions of the same names.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _native as N


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def config_high() -> N.ShadowConfig:
    """VKR_SHADOW_CONFIG_HIGH, the reference's default (vkr_shadow_system.h:194-217)."""
    cfg = N.ShadowConfig()
    N.load().vkrhip_shadow_config_high(C.byref(cfg))
    return cfg


def camera(position, forward, right, up, near_clip=0.1, far_clip=500.0, zoom=70.0,
           window=(1280, 720), ortho=None) -> N.ShadowCamera:
    """The camera fields the fit reads (VkrCamera): basis vectors, clip planes,
    vertical field of view in degrees, window size for the aspect ratio;
    `ortho` = (left, right, bottom, top) makes it orthographic."""
    cam = N.ShadowCamera()
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


class ShadowSystem:
    """vkr_shadow_system_init / _update / _invalidate_fit_history."""

    def __init__(self, config: N.ShadowConfig | None = None):
        self.lib = N.load()
        self.state = N.ShadowSystem()
        rc = self.lib.vkrhip_shadow_system_init(C.byref(self.state),
                                                None if config is None else C.byref(config))
        if rc != N.OK:
            raise N.VkrHipError(rc, "vkrhip_shadow_system_init")

    @property
    def config(self) -> N.ShadowConfig:
        return self.state.config

    def update(self, cam: N.ShadowCamera, light_enabled: bool, light_direction) -> "ShadowSystem":
        d = _f32(light_direction)
        rc = self.lib.vkrhip_shadow_system_update(C.byref(self.state), C.byref(cam),
                                                  1 if light_enabled else 0, N.ptr(d))
        if rc != N.OK:
            raise N.VkrHipError(rc, "vkrhip_shadow_system_update")
        return self

    def invalidate_fit_history(self):
        self.lib.vkrhip_shadow_system_invalidate_fit_history(C.byref(self.state))

    @property
    def cascade_count(self) -> int:
        return int(self.state.config.cascade_count) if self.state.light_enabled else 0

    def view_projections(self) -> np.ndarray:
        """(cascades, 16) light view-projection matrices of the last update."""
        n = int(self.state.config.cascade_count)
        return np.array([self.state.cascades[i].view_projection[:] for i in range(n)],
                        np.float32).reshape(n, 16)

    def split_far(self) -> np.ndarray:
        n = int(self.state.config.cascade_count)
        return np.array([self.state.cascades[i].split_far for i in range(n)], np.float32)

    def view_set(self, view, projection):
        """(planes (views, 24), required_flags (views,)) for vkrhip_set_views: the
        camera first, then the fitted cascades."""
        planes = np.zeros((1 + N.ShadowSystem.cascades.size // C.sizeof(N.ShadowCascade), 24), np.float32)
        required = np.zeros(planes.shape[0], np.uint32)
        views = C.c_uint32(0)
        v, p = _f32(view).reshape(16), _f32(projection).reshape(16)
        rc = self.lib.vkrhip_shadow_build_view_set(C.byref(self.state), N.ptr(v), N.ptr(p),
                                                   N.ptr(planes), N.ptr(required), C.byref(views))
        if rc != N.OK:
            raise N.VkrHipError(rc, "vkrhip_shadow_build_view_set")
        return planes[: views.value].copy(), required[: views.value].copy()


def fit_cascade(light_view, corners, shadow_map_size, stabilize, guard_band_texels,
                use_constant_cascade_size, config, z_extension_factor, previous_fit=None):
    """vkr_shadow_compute_cascade_matrix: (view_projection, fit, light_space_origin)."""
    vp, origin, fit = np.zeros(16, np.float32), np.zeros(2, np.float32), N.ShadowFit()
    lv, co = _f32(light_view).reshape(16), _f32(corners).reshape(24)
    N.load().vkrhip_shadow_fit_cascade(
        N.ptr(lv), N.ptr(co), shadow_map_size, int(stabilize), guard_band_texels,
        int(use_constant_cascade_size), C.byref(config), z_extension_factor,
        None if previous_fit is None else C.byref(previous_fit), N.ptr(vp), C.byref(fit), N.ptr(origin))
    return vp, fit, origin


def fit_relevant_caster_z(light_view, bounds_min, bounds_max, left, right, bottom, top):
    """vkr_shadow_fit_relevant_caster_z: (overlaps, min_z, max_z)."""
    lo, hi = np.zeros(1, np.float32), np.zeros(1, np.float32)
    lv, mn, mx = _f32(light_view).reshape(16), _f32(bounds_min), _f32(bounds_max)
    ok = N.load().vkrhip_shadow_fit_relevant_caster_z(N.ptr(lv), N.ptr(mn), N.ptr(mx), left, right,
                                                      bottom, top, N.ptr(lo), N.ptr(hi))
    return bool(ok), lo[0], hi[0]


def quantize_extent_up(extent: float, shadow_map_size: int) -> np.float32:
    """vkr_shadow_quantize_extent_up."""
    return np.float32(N.load().vkrhip_shadow_quantize_extent_up(extent, shadow_map_size))


def apply_fit_hysteresis(previous: N.ShadowFit, raw: N.ShadowFit) -> N.ShadowFit:
    """vkr_shadow_apply_fit_hysteresis."""
    out = N.ShadowFit()
    N.load().vkrhip_shadow_apply_fit_hysteresis(C.byref(previous), C.byref(raw), C.byref(out))
    return out
