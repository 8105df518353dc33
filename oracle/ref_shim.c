/*
 * ref_shim.c - thin C-ABI wrappers around the REFERENCE's own code, compiled
 * from /root/reference where it lies (see oracle/Makefile). TEST
 * INFRASTRUCTURE ONLY; built into oracle/_ref/libvkr_ref.so, which is never
 * committed and is used to (a) validate oracle/vkr_oracle.c and (b) generate
 * the golden fixtures under tests/golden/.
 *
 * ref_*  : direct call of one reference function.
 * refc_* : composition of reference leaf functions in the order a reference
 *          caller that cannot be linked here applies them (the caller is cited
 *          at each one).
 *
 * No reference source text is copied: this file only includes the reference
 * headers and calls into them.
 */
#include "defines.h"
#include "math/mat.h"
#include "math/vec.h"
#include "math/vkr_frustum.h"
#include "math/vkr_math.h"
#include "math/vkr_quat.h"
#include "renderer/systems/vkr_picking_ids.h"
#include "renderer/vkr_gpu_abi.h"
#include "renderer/vkr_render_packet.h"
#include "renderer/vkr_visibility.h"

#include <stdlib.h>
#include <string.h>

static Mat4 load_mat4(const float *m) {
  Mat4 r;
  memcpy(&r, m, sizeof(r));
  return r;
}
static void store_mat4(float *out, Mat4 m) { memcpy(out, &m, sizeof(m)); }
static void store_frustum(float *planes24, const VkrFrustum *f) {
  for (int i = 0; i < VKR_FRUSTUM_PLANE_COUNT; ++i) {
    planes24[4 * i + 0] = f->planes[i].normal.x;
    planes24[4 * i + 1] = f->planes[i].normal.y;
    planes24[4 * i + 2] = f->planes[i].normal.z;
    planes24[4 * i + 3] = f->planes[i].d;
  }
}
static VkrFrustum load_frustum(const float *planes24) {
  VkrFrustum f;
  memset(&f, 0, sizeof(f));
  for (int i = 0; i < VKR_FRUSTUM_PLANE_COUNT; ++i) {
    f.planes[i].normal = vec3_new(planes24[4 * i], planes24[4 * i + 1], planes24[4 * i + 2]);
    f.planes[i].d = planes24[4 * i + 3];
  }
  return f;
}

/* ---- struct layout probes ------------------------------------------------ */
/* VkrWorldPassPayload (vkr_render_packet.h:162-179) and VkrVisibilityStats
 * (vkr_visibility.h:42-46): what vkrhip_fill_world_payload writes. */
void ref_payload_layout(uint32_t *out) {
  uint32_t i = 0;
  out[i++] = sizeof(VkrWorldPassPayload);
  out[i++] = offsetof(VkrWorldPassPayload, gpu_candidates);
  out[i++] = offsetof(VkrWorldPassPayload, gpu_candidate_count);
  out[i++] = offsetof(VkrWorldPassPayload, gpu_camera_opaque_candidate_count);
  out[i++] = offsetof(VkrWorldPassPayload, gpu_shadow_candidate_count);
  out[i++] = offsetof(VkrWorldPassPayload, transmission_gpu_candidates);
  out[i++] = offsetof(VkrWorldPassPayload, transmission_gpu_candidate_count);
  out[i++] = offsetof(VkrWorldPassPayload, transparent_draws);
  out[i++] = offsetof(VkrWorldPassPayload, transparent_draw_count);
  out[i++] = offsetof(VkrWorldPassPayload, instances);
  out[i++] = offsetof(VkrWorldPassPayload, instance_count);
  out[i++] = offsetof(VkrWorldPassPayload, text_draws);
  out[i++] = offsetof(VkrWorldPassPayload, text_draw_count);
  out[i++] = sizeof(VkrVisibilityStats);
  out[i++] = offsetof(VkrVisibilityStats, objects_tested);
  out[i++] = offsetof(VkrVisibilityStats, objects_culled_camera);
  out[i++] = offsetof(VkrVisibilityStats, objects_without_bounds);
  out[i++] = VKR_GPU_DRAW_CANDIDATE_CAPACITY;
}

void ref_layout(uint32_t *out) {
  uint32_t i = 0;
  out[i++] = sizeof(Mat4);
  out[i++] = sizeof(Vec3);
  out[i++] = sizeof(VkrWorldDrawCandidate);
  out[i++] = offsetof(VkrWorldDrawCandidate, geometry);
  out[i++] = offsetof(VkrWorldDrawCandidate, submesh_index);
  out[i++] = offsetof(VkrWorldDrawCandidate, material);
  out[i++] = offsetof(VkrWorldDrawCandidate, instance);
  out[i++] = offsetof(VkrWorldDrawCandidate, local_bounding_sphere);
  out[i++] = offsetof(VkrWorldDrawCandidate, state_bucket);
  out[i++] = offsetof(VkrWorldDrawCandidate, flags);
  out[i++] = sizeof(VkrDrawItem);
  out[i++] = offsetof(VkrDrawItem, material);
  out[i++] = offsetof(VkrDrawItem, instance_count);
  out[i++] = offsetof(VkrDrawItem, first_instance);
  out[i++] = offsetof(VkrDrawItem, sort_key);
  out[i++] = sizeof(VkrInstanceDataGPU);
  out[i++] = offsetof(VkrInstanceDataGPU, object_id);
  out[i++] = sizeof(VkrTransparentDrawCandidate);
  out[i++] = offsetof(VkrTransparentDrawCandidate, sort_key);
  out[i++] = sizeof(VkrGpuDrawCompactionState);
  out[i++] = sizeof(VkrGpuCandidateDrawRow);
  out[i++] = sizeof(VkrGpuVisibleDrawRow);
  out[i++] = sizeof(VkrFrustum);
  out[i++] = VKR_GPU_DRAW_CANDIDATE_CAPACITY;
  out[i++] = VKR_INSTANCE_BUFFER_MAX_INSTANCES;
  out[i++] = VKR_WORLD_DRAW_CANDIDATE_BOUNDS_VALID;
  out[i++] = VKR_WORLD_DRAW_CANDIDATE_CAMERA_OPAQUE;
  out[i++] = VKR_WORLD_DRAW_CANDIDATE_SHADOW_CASTER;
}

/* ---- leaf math ----------------------------------------------------------- */
void ref_mat4_mul(const float *a, const float *b, float *out) {
  store_mat4(out, mat4_mul(load_mat4(a), load_mat4(b)));
}
void ref_mat4_mul_vec3(const float *m, const float *v, float *out4) {
  Vec3 r = mat4_mul_vec3(load_mat4(m), vec3_new(v[0], v[1], v[2]));
  memcpy(out4, &r, 16);
}
void ref_mat4_mul_vec4(const float *m, const float *v, float *out4) {
  Vec4 r = mat4_mul_vec4(load_mat4(m), vec4_new(v[0], v[1], v[2], v[3]));
  memcpy(out4, &r, 16);
}
float ref_vec3_dot(const float *a, const float *b) {
  return vec3_dot(vec3_new(a[0], a[1], a[2]), vec3_new(b[0], b[1], b[2]));
}
float ref_vec3_length(const float *v) { return vec3_length(vec3_new(v[0], v[1], v[2])); }
void ref_quat_to_mat4(const float *q, float *out) {
  VkrQuat quat;
  quat.x = q[0];
  quat.y = q[1];
  quat.z = q[2];
  quat.w = q[3];
  store_mat4(out, vkr_quat_to_mat4(quat));
}
void ref_mat4_translate(const float *v, float *out) {
  store_mat4(out, mat4_translate(vec3_new(v[0], v[1], v[2])));
}
void ref_mat4_scale(const float *v, float *out) {
  store_mat4(out, mat4_scale(vec3_new(v[0], v[1], v[2])));
}
void ref_mat4_perspective(float fov, float aspect, float n, float f, float *out) {
  store_mat4(out, mat4_perspective(fov, aspect, n, f));
}
void ref_mat4_ortho_zo_yinv(float l, float r, float b, float t, float n, float f, float *out) {
  store_mat4(out, mat4_ortho_zo_yinv(l, r, b, t, n, f));
}
void ref_mat4_look_at(const float *eye, const float *center, const float *up, float *out) {
  store_mat4(out, mat4_look_at(vec3_new(eye[0], eye[1], eye[2]),
                               vec3_new(center[0], center[1], center[2]),
                               vec3_new(up[0], up[1], up[2])));
}
float ref_to_radians(float deg) { return vkr_to_radians(deg); }

/* ---- frustum ------------------------------------------------------------- */
void ref_frustum_from_matrix(const float *vp, float *planes24) {
  VkrFrustum f = vkr_frustum_from_matrix(load_mat4(vp));
  store_frustum(planes24, &f);
}
void ref_frustum_from_view_projection(const float *view, const float *proj, float *planes24) {
  VkrFrustum f = vkr_frustum_from_view_projection(load_mat4(view), load_mat4(proj));
  store_frustum(planes24, &f);
}
int ref_frustum_test_sphere(const float *planes24, const float *c, float radius) {
  VkrFrustum f = load_frustum(planes24);
  return vkr_frustum_test_sphere(&f, vec3_new(c[0], c[1], c[2]), radius) ? 1 : 0;
}

/* ---- visibility helpers -------------------------------------------------- */
void ref_submesh_sphere(const float *model, const float *center, const float *mn,
                        const float *mx, float *out_center3, float *out_radius) {
  Vec3 c;
  float32_t r = 0.0f;
  vkr_visibility_submesh_sphere(load_mat4(model), vec3_new(center[0], center[1], center[2]),
                                vec3_new(mn[0], mn[1], mn[2]), vec3_new(mx[0], mx[1], mx[2]),
                                &c, &r);
  out_center3[0] = c.x;
  out_center3[1] = c.y;
  out_center3[2] = c.z;
  *out_radius = r;
}
uint32_t ref_state_bucket(uint32_t alpha_mode, uint32_t double_sided) {
  return (uint32_t)vkr_world_draw_state_bucket((VkrMaterialAlphaMode)alpha_mode,
                                               double_sided ? true_v : false_v);
}
void ref_alpha_routing(uint32_t alpha_mode, uint32_t *out2) {
  VkrDrawAlphaRouting r = vkr_draw_alpha_routing((VkrMaterialAlphaMode)alpha_mode);
  out2[0] = r.world_transparent ? 1u : 0u;
  out2[1] = r.shadow_alpha_tested ? 1u : 0u;
}
uint32_t ref_picking_encode_scene(uint32_t render_id) {
  return vkr_picking_encode_id(VKR_PICKING_ID_KIND_SCENE, render_id);
}

/* qsort + emit over VkrTransparentDrawCandidate rows the caller lays out in
 * the reference's struct (112 B). Writes VkrDrawItem (48 B) and
 * VkrInstanceDataGPU (80 B) rows. */
uint32_t ref_transparent_sort_emit(void *candidates112, uint32_t count, void *draws48,
                                   void *instances80) {
  if (count > 1u)
    qsort(candidates112, count, sizeof(VkrTransparentDrawCandidate),
          vkr_transparent_draw_depth_compare);
  return vkr_transparent_draw_emit((const VkrTransparentDrawCandidate *)candidates112, count,
                                   (VkrDrawItem *)draws48, (VkrInstanceDataGPU *)instances80);
}

/* ---- compositions of reference leaf functions ---------------------------- */

/* scene_compute_local_matrix, vkr_scene_system.c:555-561 */
void refc_local_from_trs(const float *pos, const float *quat, const float *scale, float *out) {
  VkrQuat q;
  q.x = quat[0];
  q.y = quat[1];
  q.z = quat[2];
  q.w = quat[3];
  Mat4 t = mat4_translate(vec3_new(pos[0], pos[1], pos[2]));
  Mat4 r = vkr_quat_to_mat4(q);
  Mat4 s = mat4_scale(vec3_new(scale[0], scale[1], scale[2]));
  store_mat4(out, mat4_mul(mat4_mul(t, r), s));
}

/* vkr_scene_update pass 2, vkr_scene_system.c:1293-1336, all nodes dirty:
 * world = parent ? mat4_mul(parent.world, local) : local, parents first. */
void refc_propagate(const uint32_t *topo_order, const uint32_t *parent, uint64_t count,
                    const float *local16, float *world16) {
  for (uint64_t k = 0; k < count; ++k) {
    const uint32_t i = topo_order[k];
    if (parent[i] != 0xffffffffu) {
      Mat4 w = mat4_mul(load_mat4(world16 + 16 * (uint64_t)parent[i]),
                        load_mat4(local16 + 16 * (uint64_t)i));
      store_mat4(world16 + 16 * (uint64_t)i, w);
    } else {
      memcpy(world16 + 16 * (uint64_t)i, local16 + 16 * (uint64_t)i, 64);
    }
  }
}

/* application_transparent_depth + application_pack_transparent_sort_key,
 * application.h:573-587 */
uint64_t refc_transparent_key(const float *view, const float *model, const float *center,
                              uint32_t tie_breaker, float *out_depth) {
  Vec3 wc = mat4_mul_vec3(load_mat4(model), vec3_new(center[0], center[1], center[2]));
  Vec4 vp = mat4_mul_vec4(load_mat4(view), vec4_new(wc.x, wc.y, wc.z, 1.0f));
  float32_t depth = -vp.z;
  depth = depth > 0.0f ? depth : 0.0f;
  if (out_depth)
    *out_depth = depth;
  uint32_t bits = 0;
  memcpy(&bits, &depth, sizeof(bits));
  return ((uint64_t)bits << 32) | (uint64_t)tie_breaker;
}

/* local bounding sphere radius as application_emit_world_source computes it,
 * application.h:620-629 */
float refc_half_extent_length(const float *mn, const float *mx) {
  const Vec3 half = vec3_scale(
      vec3_sub(vec3_new(mx[0], mx[1], mx[2]), vec3_new(mn[0], mn[1], mn[2])), 0.5f);
  return vec3_length(half);
}

/* classify sphere test as the CPU helpers evaluate the shader's formula
 * (deferred.slang:325-335 == vkr_visibility.c:37-50 + vkr_frustum.c:68-84):
 * centre = mat4_mul_vec3(model, sphere.xyz), radius = sphere.w * max basis length */
int refc_sphere_visible(const float *model, const float *sphere4, const float *planes24,
                        float *out_center_radius4) {
  Mat4 m = load_mat4(model);
  Vec3 c = mat4_mul_vec3(m, vec3_new(sphere4[0], sphere4[1], sphere4[2]));
  Vec3 col0 = vec3_new(m.m00, m.m10, m.m20);
  Vec3 col1 = vec3_new(m.m01, m.m11, m.m21);
  Vec3 col2 = vec3_new(m.m02, m.m12, m.m22);
  const float32_t max_scale =
      vkr_max_f32(vkr_max_f32(vec3_length(col0), vec3_length(col1)), vec3_length(col2));
  const float32_t radius = sphere4[3] * max_scale;
  if (out_center_radius4) {
    out_center_radius4[0] = c.x;
    out_center_radius4[1] = c.y;
    out_center_radius4[2] = c.z;
    out_center_radius4[3] = radius;
  }
  VkrFrustum f = load_frustum(planes24);
  return vkr_frustum_test_sphere(&f, c, radius) ? 1 : 0;
}
