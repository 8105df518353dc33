/*
 * ref_shadow_shim.c - C-ABI wrappers around the REFERENCE's own cascade fitting,
 * compiled from /root/reference where it lies (see oracle/Makefile). TEST
 * INFRASTRUCTURE ONLY; built into oracle/_ref/libvkr_ref_shadow.so, which is
 * never committed. The reference keeps most of the fit's steps file-local, so
 * this file includes the reference's translation unit instead of linking it;
 * no reference source text is copied here.
 *
 * The structs that cross this boundary are the oracle's (vkr_oracle.h): plain
 * floats and integers, converted to the reference's types field by field.
 */
#include "renderer/systems/vkr_shadow_system.c"

#include "renderer/systems/vkr_camera.h"
#include "vkr_oracle.h"

#include <string.h>

static Vec3 load3(const float *v) { return vec3_new(v[0], v[1], v[2]); }
static void store3(float *o, Vec3 v) {
  o[0] = v.x;
  o[1] = v.y;
  o[2] = v.z;
}
static Mat4 load16(const float *m) {
  Mat4 r;
  memcpy(&r, m, sizeof(r));
  return r;
}

static VkrCamera to_camera(const vko_shadow_camera *c) {
  VkrCamera cam;
  memset(&cam, 0, sizeof(cam));
  cam.type = c->type == VKO_CAMERA_ORTHOGRAPHIC ? VKR_CAMERA_TYPE_ORTHOGRAPHIC : VKR_CAMERA_TYPE_PERSPECTIVE;
  cam.position = load3(c->position);
  cam.forward = load3(c->forward);
  cam.right = load3(c->right);
  cam.up = load3(c->up);
  cam.near_clip = c->near_clip;
  cam.far_clip = c->far_clip;
  cam.zoom = c->zoom;
  cam.cached_window_width = c->window_width;
  cam.cached_window_height = c->window_height;
  cam.left_clip = c->left_clip;
  cam.right_clip = c->right_clip;
  cam.bottom_clip = c->bottom_clip;
  cam.top_clip = c->top_clip;
  cam.window = NULL;
  return cam;
}

static VkrShadowConfig to_config(const vko_shadow_config *c) {
  VkrShadowConfig cfg = VKR_SHADOW_CONFIG_DEFAULT;
  cfg.cascade_count = c->cascade_count;
  cfg.shadow_map_size = c->shadow_map_size;
  cfg.cascade_split_lambda = c->cascade_split_lambda;
  cfg.max_shadow_distance = c->max_shadow_distance;
  cfg.cascade_guard_band_texels = c->cascade_guard_band_texels;
  cfg.z_extension_factor = c->z_extension_factor;
  for (int i = 0; i < VKR_SHADOW_CASCADE_COUNT_MAX; ++i) {
    cfg.cascade_guard_band_texels_per[i] = c->cascade_guard_band_texels_per[i];
    cfg.cascade_z_extension_factor_per[i] = c->cascade_z_extension_factor_per[i];
  }
  cfg.use_constant_cascade_size = c->use_constant_cascade_size ? true_v : false_v;
  cfg.anchor_snap_texels = c->anchor_snap_texels;
  cfg.stabilize_cascades = c->stabilize_cascades ? true_v : false_v;
  cfg.scene_bounds.use_scene_bounds = c->use_scene_bounds ? true_v : false_v;
  cfg.scene_bounds.min = load3(c->scene_min);
  cfg.scene_bounds.max = load3(c->scene_max);
  return cfg;
}

static VkrShadowFit to_fit(const vko_shadow_fit *f) {
  VkrShadowFit r = {f->center_x, f->center_y, f->extent, f->min_z, f->max_z, f->world_units_per_texel};
  return r;
}
static void from_fit(vko_shadow_fit *o, const VkrShadowFit *f) {
  o->center_x = f->center_x;
  o->center_y = f->center_y;
  o->extent = f->extent;
  o->min_z = f->min_z;
  o->max_z = f->max_z;
  o->world_units_per_texel = f->world_units_per_texel;
}

/* the default preset, as the reference's header spells it */
void ref_shadow_default_config(vko_shadow_config *out) {
  const VkrShadowConfig d = VKR_SHADOW_CONFIG_DEFAULT;
  memset(out, 0, sizeof(*out));
  out->cascade_count = d.cascade_count;
  out->shadow_map_size = d.shadow_map_size;
  out->cascade_split_lambda = d.cascade_split_lambda;
  out->max_shadow_distance = d.max_shadow_distance;
  out->cascade_guard_band_texels = d.cascade_guard_band_texels;
  out->z_extension_factor = d.z_extension_factor;
  for (int i = 0; i < VKR_SHADOW_CASCADE_COUNT_MAX; ++i) {
    out->cascade_guard_band_texels_per[i] = d.cascade_guard_band_texels_per[i];
    out->cascade_z_extension_factor_per[i] = d.cascade_z_extension_factor_per[i];
  }
  out->use_constant_cascade_size = d.use_constant_cascade_size;
  out->anchor_snap_texels = d.anchor_snap_texels;
  out->stabilize_cascades = d.stabilize_cascades;
  out->use_scene_bounds = d.scene_bounds.use_scene_bounds;
  store3(out->scene_min, d.scene_bounds.min);
  store3(out->scene_max, d.scene_bounds.max);
}

void ref_vec3_normalize(const float *v, float *out) { store3(out, vec3_normalize(load3(v))); }

void ref_mat4_inverse_rigid(const float *m, float *out) {
  const Mat4 r = mat4_inverse_rigid(load16(m));
  memcpy(out, &r, sizeof(r));
}

void ref_shadow_cascade_splits(const vko_shadow_config *cfg, float near_clip, float far_clip, float lambda,
                               float *splits) {
  VkrShadowSystem system;
  memset(&system, 0, sizeof(system));
  system.config = to_config(cfg);
  vkr_shadow_compute_cascade_splits(&system, near_clip, far_clip, lambda);
  memcpy(splits, system.cascade_splits, sizeof(system.cascade_splits));
}

void ref_shadow_frustum_corners(const vko_shadow_camera *cam, float near_split, float far_split,
                                float *corners24) {
  const VkrCamera camera = to_camera(cam);
  Vec3 corners[8];
  memset(corners, 0, sizeof(corners));
  vkr_shadow_compute_frustum_corners(&camera, near_split, far_split, corners);
  for (int i = 0; i < 8; ++i) store3(corners24 + 3 * i, corners[i]);
}

void ref_shadow_light_view(const vko_shadow_camera *cam, const vko_shadow_config *cfg,
                           const float *light_direction, uint32_t shadow_map_size, float *out) {
  const VkrCamera camera = to_camera(cam);
  const VkrShadowConfig config = to_config(cfg);
  const Mat4 view = vkr_shadow_compute_light_view(&camera, &config.scene_bounds, load3(light_direction),
                                                  config.max_shadow_distance, shadow_map_size,
                                                  config.anchor_snap_texels);
  memcpy(out, &view, sizeof(view));
}

int ref_shadow_fit_relevant_caster_z(const float *light_view, const float *bounds_min, const float *bounds_max,
                                     float left, float right, float bottom, float top, float *out_min_z,
                                     float *out_max_z) {
  const Mat4 view = load16(light_view);
  VkrShadowSceneBounds bounds;
  bounds.min = load3(bounds_min);
  bounds.max = load3(bounds_max);
  bounds.use_scene_bounds = true_v;
  return vkr_shadow_fit_relevant_caster_z(&view, &bounds, left, right, bottom, top, out_min_z, out_max_z) ? 1 : 0;
}

float ref_shadow_quantize_extent_up(float extent, uint32_t shadow_map_size) {
  return vkr_shadow_quantize_extent_up(extent, shadow_map_size);
}

void ref_shadow_apply_fit_hysteresis(const vko_shadow_fit *previous, const vko_shadow_fit *raw,
                                     vko_shadow_fit *out) {
  const VkrShadowFit p = to_fit(previous), r = to_fit(raw);
  const VkrShadowFit f = vkr_shadow_apply_fit_hysteresis(&p, &r);
  from_fit(out, &f);
}

void ref_shadow_cascade_matrix(const float *light_view, const float *corners24, uint32_t shadow_map_size,
                               int stabilize, float guard_band_texels, int use_constant_cascade_size,
                               const vko_shadow_config *cfg, float z_extension_factor,
                               const vko_shadow_fit *previous_fit, float *out_view_projection,
                               vko_shadow_fit *out_fit, float *out_light_space_origin) {
  const Mat4 view = load16(light_view);
  const VkrShadowConfig config = to_config(cfg);
  Vec3 corners[8];
  for (int i = 0; i < 8; ++i) corners[i] = load3(corners24 + 3 * i);
  VkrShadowFit previous;
  if (previous_fit) previous = to_fit(previous_fit);
  Mat4 vp;
  VkrShadowFit fit;
  Vec2 origin;
  vkr_shadow_compute_cascade_matrix(&view, corners, shadow_map_size, stabilize ? true_v : false_v,
                                    guard_band_texels, use_constant_cascade_size ? true_v : false_v,
                                    &config.scene_bounds, z_extension_factor,
                                    previous_fit ? &previous : NULL, &vp, &fit, &origin);
  memcpy(out_view_projection, &vp, sizeof(vp));
  from_fit(out_fit, &fit);
  out_light_space_origin[0] = origin.x;
  out_light_space_origin[1] = origin.y;
}

/* ---- the whole system, driven like tests/src/shadow_system_test.c drives it ---- */
static int g_frontend_placeholder; /* init only rejects a null frontend (shadow_system_test.c:21-27) */

void *ref_shadow_system_create(const vko_shadow_config *cfg) {
  VkrShadowSystem *system = (VkrShadowSystem *)calloc(1, sizeof(VkrShadowSystem));
  const VkrShadowConfig config = to_config(cfg);
  if (!system || !vkr_shadow_system_init(system, (struct s_RendererFrontend *)&g_frontend_placeholder, &config)) {
    free(system);
    return NULL;
  }
  return system;
}
void ref_shadow_system_destroy(void *handle) { free(handle); }
void ref_shadow_system_invalidate(void *handle) {
  vkr_shadow_system_invalidate_fit_history((VkrShadowSystem *)handle);
}
void ref_shadow_system_set_stabilize(void *handle, int on) {
  ((VkrShadowSystem *)handle)->config.stabilize_cascades = on ? true_v : false_v;
}
void ref_shadow_system_update(void *handle, const vko_shadow_camera *cam, int light_enabled,
                              const float *light_direction) {
  const VkrCamera camera = to_camera(cam);
  vkr_shadow_system_update((VkrShadowSystem *)handle, &camera, light_enabled ? true_v : false_v,
                           load3(light_direction));
}
/* state after an update, in the oracle's layout */
void ref_shadow_system_read(const void *handle, vko_shadow_system *out) {
  const VkrShadowSystem *s = (const VkrShadowSystem *)handle;
  memset(out, 0, sizeof(*out));
  out->config.cascade_count = s->config.cascade_count;
  out->config.shadow_map_size = s->config.shadow_map_size;
  out->config.stabilize_cascades = s->config.stabilize_cascades;
  for (int i = 0; i < VKR_SHADOW_CASCADE_COUNT_MAX; ++i) {
    const VkrCascadeData *c = &s->cascades[i];
    vko_shadow_cascade *o = &out->cascades[i];
    memcpy(o->view_projection, &c->view_projection, sizeof(o->view_projection));
    o->split_far = c->split_far;
    o->world_units_per_texel = c->world_units_per_texel;
    o->light_space_origin[0] = c->light_space_origin.x;
    o->light_space_origin[1] = c->light_space_origin.y;
    o->light_space_depth_span = c->light_space_depth_span;
    store3(o->bounds_center, c->bounds_center);
    o->bounds_radius = c->bounds_radius;
    from_fit(&out->history_fits[i], &s->fit_history.cascades[i]);
  }
  memcpy(out->cascade_splits, s->cascade_splits, sizeof(out->cascade_splits));
  store3(out->light_direction, s->light_direction);
  out->light_enabled = s->light_enabled;
  store3(out->history_light_direction, s->fit_history.light_direction);
  out->history_cascade_count = s->fit_history.cascade_count;
  out->history_shadow_map_size = s->fit_history.shadow_map_size;
  out->history_projection_convention = s->fit_history.projection_convention;
  out->history_valid = s->fit_history.valid;
  out->history_enable_generation = s->fit_history.enable_generation;
  out->enable_generation = s->enable_generation;
  out->initialized = s->initialized;
}
void ref_shadow_system_poke_history(void *handle, uint32_t cascade, const vko_shadow_fit *fit) {
  ((VkrShadowSystem *)handle)->fit_history.cascades[cascade] = to_fit(fit);
}
