/*
 * vkr_oracle_shadow.c - CPU restatement of the reference's cascade fitting
 * (lib/src/renderer/systems/vkr_shadow_system.c), the step that produces the
 * shadow views the cull consumes. TEST INFRASTRUCTURE ONLY, like vkr_oracle.c.
 *
 * One deliberate difference: the reference normalises vectors with its SIMD
 * reciprocal square root, which on x86 is the hardware estimate plus one
 * Newton step (lib/src/math/vkr_simd.h:839-851) and on other targets something
 * else again, so its own results differ from CPU to CPU. This file uses the
 * reference's portable form, v * (1 / sqrtf(|v|^2)) (vkr_simd.h:1112-1116).
 * Everything downstream of the normalised vectors - corners, light-space
 * bounds, the caster-depth fit, quantisation, hysteresis, the projection - is
 * evaluated operation for operation like the reference and compares bit for
 * bit when it is given the same light view and corners.
 *
 * Compile with -ffp-contract=off.
 */
#include "vkr_oracle.h"

#include <math.h>
#include <string.h>

#define SH_FLOAT_EPSILON 1.192092896e-07f /* lib/src/math/vkr_math.h:154 */
#define SH_FLOAT_MAX 3.402823466e+38f     /* lib/src/math/vkr_math.h:160 */
#define SH_PI 3.14159265358979323846f     /* lib/src/math/vkr_math.h:72 */

static float sh_min(float a, float b) { return (a < b) ? a : b; } /* vkr_math.h:196-198 */
static float sh_max(float a, float b) { return (a > b) ? a : b; } /* vkr_math.h:206-208 */

static void v3_set(float *o, float x, float y, float z) {
  o[0] = x;
  o[1] = y;
  o[2] = z;
}
static void v3_add(const float *a, const float *b, float *o) { v3_set(o, a[0] + b[0], a[1] + b[1], a[2] + b[2]); }
static void v3_sub(const float *a, const float *b, float *o) { v3_set(o, a[0] - b[0], a[1] - b[1], a[2] - b[2]); }
static void v3_scale(const float *a, float s, float *o) { v3_set(o, a[0] * s, a[1] * s, a[2] * s); }
/* vec.h:644-655: (a.yzx * b.zxy) - (a.zxy * b.yzx), products rounded separately */
static void v3_cross(const float *a, const float *b, float *o) {
  const float x = a[1] * b[2] - a[2] * b[1];
  const float y = a[2] * b[0] - a[0] * b[2];
  const float z = a[0] * b[1] - a[1] * b[0];
  v3_set(o, x, y, z);
}
/* vec.h:686-695 with the portable reciprocal square root (see the file comment) */
static void v3_normalize(const float *v, float *o) {
  const float len_sq = vko_vec3_dot(v, v);
  if (len_sq > SH_FLOAT_EPSILON * SH_FLOAT_EPSILON) {
    const float inv = 1.0f / sqrtf(len_sq);
    v3_set(o, v[0] * inv, v[1] * inv, v[2] * inv);
  } else {
    v3_set(o, 0.0f, 0.0f, 0.0f);
  }
}
static void xform_point(const float *m, const float *p, float *out4) {
  const float v[4] = {p[0], p[1], p[2], 1.0f};
  vko_mat4_mul_vec4(m, v, out4);
}

/* mat.h:447-457 */
void vko_mat4_look_at(const float *eye, const float *center, const float *up, float *out) {
  float d[3], f[3], c[3], s[3], u[3];
  v3_sub(center, eye, d);
  v3_normalize(d, f);
  v3_cross(f, up, c);
  v3_normalize(c, s);
  v3_cross(s, f, u);
  out[0] = s[0];
  out[1] = u[0];
  out[2] = -f[0];
  out[3] = 0.0f;
  out[4] = s[1];
  out[5] = u[1];
  out[6] = -f[1];
  out[7] = 0.0f;
  out[8] = s[2];
  out[9] = u[2];
  out[10] = -f[2];
  out[11] = 0.0f;
  out[12] = -vko_vec3_dot(s, eye);
  out[13] = -vko_vec3_dot(u, eye);
  out[14] = vko_vec3_dot(f, eye);
  out[15] = 1.0f;
}

/* mat.h:1717-1755 */
void vko_mat4_inverse_rigid(const float *m, float *out) {
  float r[16];
  r[0] = m[0];
  r[4] = m[1];
  r[8] = m[2];
  r[1] = m[4];
  r[5] = m[5];
  r[9] = m[6];
  r[2] = m[8];
  r[6] = m[9];
  r[10] = m[10];
  r[3] = r[7] = r[11] = 0.0f;
  const float t[3] = {m[12], m[13], m[14]};
  const float row0[3] = {r[0], r[4], r[8]};
  const float row1[3] = {r[1], r[5], r[9]};
  const float row2[3] = {r[2], r[6], r[10]};
  r[12] = -vko_vec3_dot(row0, t);
  r[13] = -vko_vec3_dot(row1, t);
  r[14] = -vko_vec3_dot(row2, t);
  r[15] = 1.0f;
  memcpy(out, r, sizeof(r));
}

/* vkr_shadow_system.c:12-37 */
void vko_shadow_cascade_splits(const vko_shadow_config *cfg, float near_clip, float far_clip,
                               float lambda, float *splits) {
  const uint32_t count = cfg->cascade_count;
  if (count == 0) {
    splits[0] = near_clip;
    splits[1] = far_clip;
    return;
  }
  float far_for_shadows = far_clip;
  if (cfg->max_shadow_distance > 0.0f) far_for_shadows = sh_min(far_for_shadows, cfg->max_shadow_distance);
  far_for_shadows = sh_max(far_for_shadows, near_clip + 0.001f);
  for (uint32_t i = 0; i <= count; ++i) {
    const float p = (float)i / (float)count;
    const float log_split = near_clip * powf(far_for_shadows / near_clip, p);
    const float linear_split = near_clip + (far_for_shadows - near_clip) * p;
    splits[i] = lambda * log_split + (1.0f - lambda) * linear_split;
  }
}

/* vkr_shadow_system.c:39-72 */
static int sh_any_per_cascade(const float *values, uint32_t count) {
  for (uint32_t i = 0; i < count; ++i)
    if (values[i] > 0.0f) return 1;
  return 0;
}
static float sh_profile_lerp(float start, float end, uint32_t cascade, uint32_t count) {
  if (count <= 1) return start;
  const float t = (float)cascade / (float)(count - 1);
  return start + (end - start) * t;
}

/* vkr_shadow_system.c:74-153; corners: near TL/TR/BR/BL, far TL/TR/BR/BL */
void vko_shadow_frustum_corners(const vko_shadow_camera *cam, float near_split, float far_split,
                                float *corners24) {
  float forward[3], right[3], up[3];
  memcpy(forward, cam->forward, sizeof(forward));
  if (vko_vec3_length(forward) < 0.001f) v3_set(forward, 0.0f, 0.0f, -1.0f);
  v3_normalize(forward, forward);
  memcpy(right, cam->right, sizeof(right));
  if (vko_vec3_length(right) < 0.001f) v3_set(right, 1.0f, 0.0f, 0.0f);
  v3_normalize(right, right);
  memcpy(up, cam->up, sizeof(up));
  if (vko_vec3_length(up) < 0.001f) v3_set(up, 0.0f, 1.0f, 0.0f);
  v3_normalize(up, up);

  const float near_d = sh_max(near_split, 0.0f);
  const float far_d = sh_max(far_split, near_d);
  float near_half_w = 0.0f, near_half_h = 0.0f, far_half_w = 0.0f, far_half_h = 0.0f;
  if (cam->type == VKO_CAMERA_PERSPECTIVE) {
    float aspect = 1.0f;
    if (cam->window_width > 0 && cam->window_height > 0)
      aspect = (float)cam->window_width / (float)cam->window_height;
    const float fov = cam->zoom * SH_PI / 180.0f; /* vkr_math.h:123,176-178 */
    const float tan_half_fov = tanf(fov * 0.5f);
    near_half_h = near_d * tan_half_fov;
    near_half_w = near_half_h * aspect;
    far_half_h = far_d * tan_half_fov;
    far_half_w = far_half_h * aspect;
  } else if (cam->type == VKO_CAMERA_ORTHOGRAPHIC) {
    near_half_w = 0.5f * (cam->right_clip - cam->left_clip);
    near_half_h = 0.5f * (cam->top_clip - cam->bottom_clip);
    far_half_w = near_half_w;
    far_half_h = near_half_h;
  } else {
    return;
  }
  float tmp[3], near_center[3], far_center[3], near_right[3], near_up[3], far_right[3], far_up[3];
  v3_scale(forward, near_d, tmp);
  v3_add(cam->position, tmp, near_center);
  v3_scale(forward, far_d, tmp);
  v3_add(cam->position, tmp, far_center);
  v3_scale(right, near_half_w, near_right);
  v3_scale(up, near_half_h, near_up);
  v3_scale(right, far_half_w, far_right);
  v3_scale(up, far_half_h, far_up);
  float *c = corners24;
  v3_sub(near_center, near_right, tmp);
  v3_add(tmp, near_up, c + 0);
  v3_add(near_center, near_right, tmp);
  v3_add(tmp, near_up, c + 3);
  v3_add(near_center, near_right, tmp);
  v3_sub(tmp, near_up, c + 6);
  v3_sub(near_center, near_right, tmp);
  v3_sub(tmp, near_up, c + 9);
  v3_sub(far_center, far_right, tmp);
  v3_add(tmp, far_up, c + 12);
  v3_add(far_center, far_right, tmp);
  v3_add(tmp, far_up, c + 15);
  v3_add(far_center, far_right, tmp);
  v3_sub(tmp, far_up, c + 18);
  v3_sub(far_center, far_right, tmp);
  v3_sub(tmp, far_up, c + 21);
}

/* vkr_shadow_system.c:155-219 */
void vko_shadow_light_view(const vko_shadow_camera *cam, const vko_shadow_config *cfg,
                           const float *light_direction, uint32_t shadow_map_size, float *out) {
  float dir[3];
  memcpy(dir, light_direction, sizeof(dir));
  if (vko_vec3_length(dir) < 0.001f) v3_set(dir, 0.0f, -1.0f, 0.0f);
  v3_normalize(dir, dir);
  float up_ref[3];
  if (fabsf(dir[1]) > 0.99f)
    v3_set(up_ref, 0.0f, 0.0f, 1.0f);
  else
    v3_set(up_ref, 0.0f, 1.0f, 0.0f);
  float right[3], up[3], tmp[3];
  v3_cross(up_ref, dir, tmp);
  v3_normalize(tmp, right);
  v3_cross(dir, right, up);

  float anchor[3] = {0.0f, 0.0f, 0.0f};
  float radius = 1.0f;
  if (cfg->use_scene_bounds) {
    float sum[3], half_extent[3];
    v3_add(cfg->scene_min, cfg->scene_max, sum);
    v3_scale(sum, 0.5f, anchor);
    v3_sub(cfg->scene_max, cfg->scene_min, tmp);
    v3_scale(tmp, 0.5f, half_extent);
    radius = vko_vec3_length(half_extent);
  } else {
    memcpy(anchor, cam->position, sizeof(anchor));
    float far_for_shadows = cam->far_clip;
    if (cfg->max_shadow_distance > 0.0f) far_for_shadows = sh_min(far_for_shadows, cfg->max_shadow_distance);
    far_for_shadows = sh_max(far_for_shadows, cam->near_clip + 0.001f);
    float corners[24];
    memset(corners, 0, sizeof(corners));
    vko_shadow_frustum_corners(cam, cam->near_clip, far_for_shadows, corners);
    radius = 0.0f;
    for (int i = 0; i < 8; ++i) {
      float diff[3];
      v3_sub(corners + 3 * i, anchor, diff);
      radius = sh_max(radius, vko_vec3_length(diff));
    }
  }
  if (radius < 0.001f) radius = 0.001f;

  if (shadow_map_size > 0 && cfg->anchor_snap_texels > 0.0f) {
    const float texel_size = (radius * 2.0f) / (float)shadow_map_size;
    if (texel_size > 0.0f) {
      const float snap = sh_max(texel_size * cfg->anchor_snap_texels, 0.001f);
      float anchor_x = vko_vec3_dot(anchor, right);
      float anchor_y = vko_vec3_dot(anchor, up);
      const float anchor_z = vko_vec3_dot(anchor, dir);
      anchor_x = floorf(anchor_x / snap) * snap;
      anchor_y = floorf(anchor_y / snap) * snap;
      float a[3], b[3], c[3], ab[3];
      v3_scale(right, anchor_x, a);
      v3_scale(up, anchor_y, b);
      v3_scale(dir, anchor_z, c);
      v3_add(a, b, ab);
      v3_add(ab, c, anchor);
    }
  }
  const float light_distance = sh_max(radius * 2.0f, 1.0f);
  float light_pos[3];
  v3_scale(dir, light_distance, tmp);
  v3_sub(anchor, tmp, light_pos);
  vko_mat4_look_at(light_pos, anchor, up, out);
}

static void sh_include_z(float z, int *found, float *min_z, float *max_z) { /* :227-232 */
  *min_z = *found ? sh_min(*min_z, z) : z;
  *max_z = *found ? sh_max(*max_z, z) : z;
  *found = 1;
}

/* vkr_shadow_system.c:234-346 */
int vko_shadow_fit_relevant_caster_z(const float *light_view, const float *bounds_min,
                                     const float *bounds_max, float left, float right, float bottom,
                                     float top, float *out_min_z, float *out_max_z) {
  if (bounds_min[0] > bounds_max[0] || bounds_min[1] > bounds_max[1] || bounds_min[2] > bounds_max[2] ||
      left > right || bottom > top)
    return 0;
  float corners[8][3];
  int found = 0;
  float min_z = 0.0f, max_z = 0.0f;
  for (uint32_t i = 0; i < 8u; ++i) {
    const float corner[3] = {(i & 1u) ? bounds_max[0] : bounds_min[0], (i & 2u) ? bounds_max[1] : bounds_min[1],
                             (i & 4u) ? bounds_max[2] : bounds_min[2]};
    float lc[4];
    xform_point(light_view, corner, lc);
    v3_set(corners[i], lc[0], lc[1], lc[2]);
    if (lc[0] >= left && lc[0] <= right && lc[1] >= bottom && lc[1] <= top)
      sh_include_z(lc[2], &found, &min_z, &max_z);
  }
  static const uint8_t edges[12][2] = {{0, 1}, {2, 3}, {4, 5}, {6, 7}, {0, 2}, {1, 3},
                                       {4, 6}, {5, 7}, {0, 4}, {1, 5}, {2, 6}, {3, 7}};
  const float x_planes[2] = {left, right};
  const float y_planes[2] = {bottom, top};
  for (uint32_t e = 0; e < 12u; ++e) {
    const float *a = corners[edges[e][0]];
    const float *b = corners[edges[e][1]];
    float delta[3];
    v3_sub(b, a, delta);
    if (fabsf(delta[0]) > 1e-6f) {
      for (uint32_t p = 0; p < 2u; ++p) {
        const float t = (x_planes[p] - a[0]) / delta[0];
        const float y = a[1] + delta[1] * t;
        if (t >= 0.0f && t <= 1.0f && y >= bottom && y <= top)
          sh_include_z(a[2] + delta[2] * t, &found, &min_z, &max_z);
      }
    }
    if (fabsf(delta[1]) > 1e-6f) {
      for (uint32_t p = 0; p < 2u; ++p) {
        const float t = (y_planes[p] - a[1]) / delta[1];
        const float x = a[0] + delta[0] * t;
        if (t >= 0.0f && t <= 1.0f && x >= left && x <= right)
          sh_include_z(a[2] + delta[2] * t, &found, &min_z, &max_z);
      }
    }
  }
  /* the light-space Z lines through the rectangle's corners against the box (slab test) */
  float inverse_view[16];
  vko_mat4_inverse_rigid(light_view, inverse_view);
  for (uint32_t ci = 0; ci < 4u; ++ci) {
    const float o4[4] = {(ci & 1u) ? right : left, (ci & 2u) ? top : bottom, 0.0f, 1.0f};
    const float d4[4] = {0.0f, 0.0f, 1.0f, 0.0f};
    float origin[4], direction[4];
    vko_mat4_mul_vec4(inverse_view, o4, origin);
    vko_mat4_mul_vec4(inverse_view, d4, direction);
    float t_min = -SH_FLOAT_MAX, t_max = SH_FLOAT_MAX;
    int intersects = 1;
    for (uint32_t axis = 0; axis < 3u; ++axis) {
      if (fabsf(direction[axis]) <= 1e-6f) {
        if (origin[axis] < bounds_min[axis] || origin[axis] > bounds_max[axis]) {
          intersects = 0;
          break;
        }
        continue;
      }
      float t0 = (bounds_min[axis] - origin[axis]) / direction[axis];
      float t1 = (bounds_max[axis] - origin[axis]) / direction[axis];
      if (t0 > t1) {
        const float swap = t0;
        t0 = t1;
        t1 = swap;
      }
      t_min = sh_max(t_min, t0);
      t_max = sh_min(t_max, t1);
      if (t_min > t_max) {
        intersects = 0;
        break;
      }
    }
    if (intersects) {
      sh_include_z(t_min, &found, &min_z, &max_z);
      sh_include_z(t_max, &found, &min_z, &max_z);
    }
  }
  if (!found) return 0;
  *out_min_z = min_z;
  *out_max_z = max_z;
  return 1;
}

/* vkr_shadow_system.c:348-360 */
float vko_shadow_quantize_extent_up(float extent, uint32_t shadow_map_size) {
  int exponent = 0;
  (void)frexpf(extent, &exponent);
  const float bracket = ldexpf(1.0f, exponent);
  const float quantum = bracket / (float)shadow_map_size;
  return ceilf(extent / quantum) * quantum;
}

/* vkr_shadow_system.c:362-396 */
void vko_shadow_apply_fit_hysteresis(const vko_shadow_fit *previous, const vko_shadow_fit *raw,
                                     vko_shadow_fit *out) {
  vko_shadow_fit fit = *raw;
  const float raw_texel = raw->world_units_per_texel;
  if (raw->extent < previous->extent && (previous->extent - raw->extent) < 2.0f * raw_texel) {
    fit.extent = previous->extent;
    fit.world_units_per_texel = previous->world_units_per_texel;
  }
  const float texel = fit.world_units_per_texel;
  if (fabsf(raw->center_x - previous->center_x) <= texel) fit.center_x = previous->center_x;
  if (fabsf(raw->center_y - previous->center_y) <= texel) fit.center_y = previous->center_y;
  if (raw->min_z > previous->min_z && (raw->min_z - previous->min_z) < 2.0f * texel) fit.min_z = previous->min_z;
  if (raw->max_z < previous->max_z && (previous->max_z - raw->max_z) < 2.0f * texel) fit.max_z = previous->max_z;
  *out = fit;
}

/* vkr_shadow_system.c:419-589 */
void vko_shadow_cascade_matrix(const float *light_view, const float *corners24,
                               uint32_t shadow_map_size, int stabilize, float guard_band_texels,
                               int use_constant_cascade_size, const vko_shadow_config *cfg,
                               float z_extension_factor, const vko_shadow_fit *previous_fit,
                               float *out_view_projection, vko_shadow_fit *out_fit,
                               float *out_light_space_origin) {
  float center[3] = {0.0f, 0.0f, 0.0f};
  for (int i = 0; i < 8; ++i) v3_add(center, corners24 + 3 * i, center);
  v3_scale(center, 1.0f / 8.0f, center);
  float radius_sq = 0.0f;
  for (int i = 0; i < 8; ++i) {
    float diff[3];
    v3_sub(corners24 + 3 * i, center, diff);
    radius_sq = sh_max(radius_sq, vko_vec3_dot(diff, diff));
  }
  float radius = sqrtf(radius_sq);
  if (radius < 0.001f) radius = 0.001f;
  if (stabilize) radius = ceilf(radius * 16.0f) / 16.0f;

  float min_x = SH_FLOAT_MAX, max_x = -SH_FLOAT_MAX, min_y = SH_FLOAT_MAX, max_y = -SH_FLOAT_MAX;
  float min_z = SH_FLOAT_MAX, max_z = -SH_FLOAT_MAX;
  for (int i = 0; i < 8; ++i) {
    float ls[4];
    xform_point(light_view, corners24 + 3 * i, ls);
    min_x = sh_min(min_x, ls[0]);
    max_x = sh_max(max_x, ls[0]);
    min_y = sh_min(min_y, ls[1]);
    max_y = sh_max(max_y, ls[1]);
    min_z = sh_min(min_z, ls[2]);
    max_z = sh_max(max_z, ls[2]);
  }
  const int scene = cfg && cfg->use_scene_bounds;
  if (!scene && z_extension_factor > 0.0f) {
    const float z_ext = radius * z_extension_factor;
    min_z -= z_ext;
    max_z += z_ext;
  }
  const float extent_x = max_x - min_x;
  const float extent_y = max_y - min_y;
  float center_x = (min_x + max_x) * 0.5f;
  float center_y = (min_y + max_y) * 0.5f;
  float extent = sh_max(extent_x, extent_y);
  if (use_constant_cascade_size) {
    float cls[4];
    xform_point(light_view, center, cls);
    center_x = cls[0];
    center_y = cls[1];
    extent = radius * 2.0f;
  }
  if (extent < 0.001f) extent = 0.001f;
  float texel_size = (shadow_map_size > 0) ? (extent / (float)shadow_map_size) : 0.0f;
  if (texel_size < 0.000001f) texel_size = 0.000001f;
  float guard_texels = sh_max(guard_band_texels, 0.0f);
  if (stabilize && shadow_map_size > 0) guard_texels += 1.0f;
  extent += 2.0f * texel_size * guard_texels;
  if (stabilize) extent = vko_shadow_quantize_extent_up(extent, shadow_map_size);
  texel_size = (shadow_map_size > 0) ? (extent / (float)shadow_map_size) : texel_size;
  if (texel_size < 0.000001f) texel_size = 0.000001f;
  const float half = extent * 0.5f;
  if (stabilize && shadow_map_size > 0) {
    float snap_x = center_x - half;
    float snap_y = center_y - half;
    snap_x = floorf(snap_x / texel_size) * texel_size;
    snap_y = floorf(snap_y / texel_size) * texel_size;
    center_x = snap_x + half;
    center_y = snap_y + half;
  }
  if (scene) {
    float caster_min_z = 0.0f, caster_max_z = 0.0f;
    if (vko_shadow_fit_relevant_caster_z(light_view, cfg->scene_min, cfg->scene_max, center_x - half,
                                         center_x + half, center_y - half, center_y + half,
                                         &caster_min_z, &caster_max_z)) {
      min_z = sh_min(min_z, caster_min_z);
      max_z = sh_max(max_z, caster_max_z);
    }
  }
  const float z_range = max_z - min_z;
  const float z_pad = sh_max(0.5f, z_range * 0.05f);
  min_z -= z_pad;
  max_z += z_pad;

  vko_shadow_fit fit;
  fit.center_x = center_x;
  fit.center_y = center_y;
  fit.extent = extent;
  fit.min_z = min_z;
  fit.max_z = max_z;
  fit.world_units_per_texel = texel_size;
  if (stabilize && previous_fit) vko_shadow_apply_fit_hysteresis(previous_fit, &fit, &fit);

  const float final_half = fit.extent * 0.5f;
  const float left = fit.center_x - final_half;
  const float right = fit.center_x + final_half;
  const float bottom = fit.center_y - final_half;
  const float top = fit.center_y + final_half;
  *out_fit = fit;
  /* vkr_shadow_light_space_origin_from_view, :221-225 */
  out_light_space_origin[0] = light_view[12] - left;
  out_light_space_origin[1] = bottom - light_view[13];

  float near_clip = -fit.max_z;
  float far_clip = -fit.min_z;
  if (near_clip < 0.0f) near_clip = 0.0f;
  if (far_clip <= near_clip + 0.001f) far_clip = near_clip + 0.001f;
  float projection[16];
  vko_mat4_ortho_zo_yinv(left, right, bottom, top, near_clip, far_clip, projection);
  vko_mat4_mul(projection, light_view, out_view_projection);
}

/* vkr_shadow_system.c:591-645 */
int vko_shadow_system_init(vko_shadow_system *s, const vko_shadow_config *config) {
  if (!s || !config) return 0;
  memset(s, 0, sizeof(*s));
  s->config = *config;
  vko_shadow_config *c = &s->config;
  if (c->cascade_count == 0) c->cascade_count = 1;
  if (c->cascade_count > VKO_SHADOW_CASCADES_MAX) c->cascade_count = VKO_SHADOW_CASCADES_MAX;
  if (c->shadow_map_size == 0) c->shadow_map_size = 4096; /* VKR_SHADOW_MAP_SIZE_DEFAULT */
  c->cascade_split_lambda =
      (c->cascade_split_lambda < 0.0f) ? 0.0f : (c->cascade_split_lambda > 1.0f) ? 1.0f : c->cascade_split_lambda;
  if (c->cascade_guard_band_texels < 0.0f) c->cascade_guard_band_texels = 0.0f;
  if (c->z_extension_factor < 0.0f) c->z_extension_factor = 0.0f;
  if (!(c->anchor_snap_texels >= 0.0f)) c->anchor_snap_texels = 0.0f;
  for (uint32_t i = 0; i < VKO_SHADOW_CASCADES_MAX; ++i) {
    if (!(c->cascade_guard_band_texels_per[i] >= 0.0f)) c->cascade_guard_band_texels_per[i] = 0.0f;
    if (!(c->cascade_z_extension_factor_per[i] >= 0.0f)) c->cascade_z_extension_factor_per[i] = 0.0f;
  }
  s->enable_generation = 1u;
  s->history_valid = 0;
  s->initialized = 1;
  return 1;
}

void vko_shadow_system_invalidate_fit_history(vko_shadow_system *s) { /* :413-417 */
  if (s) s->history_valid = 0;
}

/* vkr_shadow_system.c:654-777 */
void vko_shadow_system_update(vko_shadow_system *s, const vko_shadow_camera *cam, int light_enabled,
                              const float *light_direction) {
  if (!s || !s->initialized || !cam) return;
  const int was_enabled = (int)s->light_enabled;
  s->light_enabled = light_enabled ? 1u : 0u;
  memcpy(s->light_direction, light_direction, sizeof(s->light_direction));
  if (!light_enabled) {
    if (was_enabled) s->enable_generation++;
    for (uint32_t i = 0; i < s->config.cascade_count; ++i) {
      vko_shadow_cascade *c = &s->cascades[i];
      memset(c, 0, sizeof(*c));
      c->view_projection[0] = c->view_projection[5] = c->view_projection[10] = c->view_projection[15] = 1.0f;
    }
    return;
  }
  vko_shadow_cascade_splits(&s->config, cam->near_clip, cam->far_clip, s->config.cascade_split_lambda,
                            s->cascade_splits);
  const uint32_t shadow_map_size = s->config.shadow_map_size ? s->config.shadow_map_size : 4096u;
  float light_view[16];
  vko_shadow_light_view(cam, &s->config, light_direction, shadow_map_size, light_view);

  const uint32_t cascade_count = s->config.cascade_count;
  const int has_guard_per = sh_any_per_cascade(s->config.cascade_guard_band_texels_per, cascade_count);
  const int has_z_per = sh_any_per_cascade(s->config.cascade_z_extension_factor_per, cascade_count);
  const uint32_t projection_convention = 0u;
  const int history_usable =
      s->config.stabilize_cascades && s->history_valid && s->history_cascade_count == cascade_count &&
      s->history_shadow_map_size == shadow_map_size &&
      s->history_projection_convention == projection_convention &&
      s->history_enable_generation == s->enable_generation &&
      s->history_light_direction[0] == light_direction[0] &&
      s->history_light_direction[1] == light_direction[1] &&
      s->history_light_direction[2] == light_direction[2];

  for (uint32_t i = 0; i < cascade_count; ++i) {
    const float split_near = s->cascade_splits[i];
    const float split_far = s->cascade_splits[i + 1];
    float corners[24];
    memset(corners, 0, sizeof(corners));
    vko_shadow_frustum_corners(cam, split_near, split_far, corners);

    float guard_band = s->config.cascade_guard_band_texels;
    if (has_guard_per) {
      const float per = s->config.cascade_guard_band_texels_per[i];
      if (per > 0.0f) guard_band = per;
    } else {
      guard_band = guard_band * sh_profile_lerp(0.5f, 1.25f, i, cascade_count);
    }
    float z_extension = s->config.z_extension_factor;
    if (has_z_per) {
      const float per = s->config.cascade_z_extension_factor_per[i];
      if (per > 0.0f) z_extension = per;
    } else {
      z_extension = z_extension * sh_profile_lerp(0.6f, 1.2f, i, cascade_count);
    }
    vko_shadow_fit fit;
    memset(&fit, 0, sizeof(fit));
    vko_shadow_cascade *c = &s->cascades[i];
    vko_shadow_cascade_matrix(light_view, corners, shadow_map_size, (int)s->config.stabilize_cascades,
                              guard_band, (int)s->config.use_constant_cascade_size, &s->config,
                              z_extension, history_usable ? &s->history_fits[i] : NULL,
                              c->view_projection, &fit, c->light_space_origin);
    s->history_fits[i] = fit;
    c->world_units_per_texel = fit.world_units_per_texel;
    c->light_space_depth_span = fit.max_z - fit.min_z;

    float cascade_center[3] = {0.0f, 0.0f, 0.0f};
    for (int k = 0; k < 8; ++k) v3_add(cascade_center, corners + 3 * k, cascade_center);
    v3_scale(cascade_center, 1.0f / 8.0f, cascade_center);
    float max_radius_sq = 0.0f;
    for (int k = 0; k < 8; ++k) {
      float diff[3];
      v3_sub(corners + 3 * k, cascade_center, diff);
      const float dist_sq = vko_vec3_dot(diff, diff);
      if (dist_sq > max_radius_sq) max_radius_sq = dist_sq;
    }
    memcpy(c->bounds_center, cascade_center, sizeof(c->bounds_center));
    c->bounds_radius = sqrtf(max_radius_sq);
    c->split_far = split_far;
  }
  memcpy(s->history_light_direction, light_direction, sizeof(s->history_light_direction));
  s->history_cascade_count = cascade_count;
  s->history_shadow_map_size = shadow_map_size;
  s->history_projection_convention = projection_convention;
  s->history_enable_generation = s->enable_generation;
  s->history_valid = s->config.stabilize_cascades ? 1u : 0u;
}
