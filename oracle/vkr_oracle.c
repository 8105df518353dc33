/*
 * vkr_oracle.c - CPU restatement of the reference's scene-processing path.
 * TEST INFRASTRUCTURE ONLY (parity checker + timed CPU baseline); see
 * vkr_oracle.h for how it is pinned against the reference.
 *
 * Compile with -ffp-contract=off. Every floating-point expression below is
 * written in the association the reference's x86 path evaluates it in; fmaf
 * appears only where the reference calls vkr_simd_fma_f32x4.
 */
#include "vkr_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

_Static_assert(sizeof(vko_instance_data) == 80, "VkrInstanceDataGPU is 80 bytes");
_Static_assert(sizeof(vko_world_candidate) == 144, "VkrWorldDrawCandidate is 144 bytes");
_Static_assert(offsetof(vko_world_candidate, instance) == 32, "instance @32");
_Static_assert(offsetof(vko_world_candidate, local_bounding_sphere) == 112, "sphere @112");
_Static_assert(offsetof(vko_world_candidate, state_bucket) == 128, "bucket @128");
_Static_assert(offsetof(vko_world_candidate, flags) == 132, "flags @132");
_Static_assert(sizeof(vko_draw_item) == 48, "VkrDrawItem is 48 bytes");
_Static_assert(offsetof(vko_draw_item, first_instance) == 32, "first_instance @32");
_Static_assert(offsetof(vko_draw_item, sort_key) == 40, "sort_key @40");
_Static_assert(sizeof(vko_compaction_state) == 80, "VkrGpuDrawCompactionState is 80 bytes");
_Static_assert(sizeof(vko_gpu_candidate_row) == 48, "VkrGpuCandidateDrawRow is 48 bytes");
_Static_assert(sizeof(vko_gpu_visible_row) == 32, "VkrGpuVisibleDrawRow is 32 bytes");
_Static_assert(sizeof(vko_scene_transform) == 208, "SceneTransform is 208 bytes");
_Static_assert(offsetof(vko_scene_transform, parent) == 48, "parent @48");
_Static_assert(offsetof(vko_scene_transform, local) == 64, "local @64");
_Static_assert(offsetof(vko_scene_transform, world) == 128, "world @128");
_Static_assert(offsetof(vko_scene_transform, flags) == 192, "flags @192");

#define VKO_FLOAT_EPSILON 1.192092896e-07f /* lib/src/math/vkr_math.h:154 */

/* ======================================================================== */
/* Leaf math                                                                 */
/* ======================================================================== */

/* mat.h:1145-1163. Per result column i and row r:
 *   t = A.c0[r]*b.x; t = fma(A.c1[r], b.y, t); t = fma(A.c2[r], b.z, t);
 *   C.ci[r] = fma(A.c3[r], b.w, t)
 * (vkr_simd_fma_f32x4(a,b,c) = a + b*c fused, vkr_simd.h:867-873). */
void vko_mat4_mul(const float *a, const float *b, float *out) {
  float tmp[16];
  for (int i = 0; i < 4; ++i) {
    const float x = b[4 * i + 0], y = b[4 * i + 1], z = b[4 * i + 2], w = b[4 * i + 3];
    for (int r = 0; r < 4; ++r) {
      float t = a[r] * x;
      t = fmaf(a[4 + r], y, t);
      t = fmaf(a[8 + r], z, t);
      tmp[4 * i + r] = fmaf(a[12 + r], w, t);
    }
  }
  memcpy(out, tmp, sizeof(tmp));
}

/* mat.h:1295-1300: ((c0*x + c1*y) + c2*z) + c3*1.0f, all four lanes. */
void vko_mat4_mul_vec3(const float *m, const float *v, float *out4) {
  const float x = v[0], y = v[1], z = v[2];
  for (int r = 0; r < 4; ++r) {
    const float p0 = m[r] * x;
    const float p1 = m[4 + r] * y;
    const float p2 = m[8 + r] * z;
    const float p3 = m[12 + r] * 1.0f;
    out4[r] = ((p0 + p1) + p2) + p3;
  }
}

/* mat.h:1260-1264: (c0*x + c1*y) + (c2*z + c3*w). */
void vko_mat4_mul_vec4(const float *m, const float *v, float *out4) {
  const float x = v[0], y = v[1], z = v[2], w = v[3];
  for (int r = 0; r < 4; ++r) {
    const float p0 = m[r] * x;
    const float p1 = m[4 + r] * y;
    const float p2 = m[8 + r] * z;
    const float p3 = m[12 + r] * w;
    out4[r] = (p0 + p1) + (p2 + p3);
  }
}

/* vkr_simd.h:914-917: _mm_dp_ps(a, b, 0x71) = (x*x' + y*y') + (z*z' + 0). */
float vko_vec3_dot(const float *a, const float *b) {
  const float p0 = a[0] * b[0];
  const float p1 = a[1] * b[1];
  const float p2 = a[2] * b[2];
  const float zero = 0.0f;
  return (p0 + p1) + (p2 + zero);
}

/* vec.h:663-675 */
float vko_vec3_length(const float *v) { return sqrtf(vko_vec3_dot(v, v)); }

/* vkr_math.h:206-208 */
static inline float vko_max_f32(float a, float b) { return (a > b) ? a : b; }

/* mat.h:1538-1560 */
void vko_quat_to_mat4(const float *q, float *out) {
  const float x = q[0], y = q[1], z = q[2], w = q[3];
  const float xx = x * x, yy = y * y, zz = z * z;
  const float xy = x * y, xz = x * z, yz = y * z;
  const float wx = w * x, wy = w * y, wz = w * z;
  out[0] = 1.0f - 2.0f * (yy + zz);
  out[1] = 2.0f * (xy + wz);
  out[2] = 2.0f * (xz - wy);
  out[3] = 0.0f;
  out[4] = 2.0f * (xy - wz);
  out[5] = 1.0f - 2.0f * (xx + zz);
  out[6] = 2.0f * (yz + wx);
  out[7] = 0.0f;
  out[8] = 2.0f * (xz + wy);
  out[9] = 2.0f * (yz - wx);
  out[10] = 1.0f - 2.0f * (xx + yy);
  out[11] = 0.0f;
  out[12] = 0.0f;
  out[13] = 0.0f;
  out[14] = 0.0f;
  out[15] = 1.0f;
}

/* mat.h:479-482 */
void vko_mat4_translate(const float *v, float *out) {
  memset(out, 0, 16 * sizeof(float));
  out[0] = out[5] = out[10] = out[15] = 1.0f;
  out[12] = v[0];
  out[13] = v[1];
  out[14] = v[2];
}

/* mat.h:507-510 */
void vko_mat4_scale(const float *v, float *out) {
  memset(out, 0, 16 * sizeof(float));
  out[0] = v[0];
  out[5] = v[1];
  out[10] = v[2];
  out[15] = 1.0f;
}

/* vkr_scene_system.c:555-561: mat4_mul(mat4_mul(T, R), S) */
void vko_local_from_trs(const float *pos, const float *quat, const float *scale, float *out) {
  float t[16], r[16], s[16], tr[16];
  vko_mat4_translate(pos, t);
  vko_quat_to_mat4(quat, r);
  vko_mat4_scale(scale, s);
  vko_mat4_mul(t, r, tr);
  vko_mat4_mul(tr, s, out);
}

/* mat.h:405-416 */
void vko_mat4_perspective(float fov, float aspect, float near_clip, float far_clip,
                          float *out) {
  const float f = 1.0f / tanf(fov * 0.5f);
  memset(out, 0, 16 * sizeof(float));
  out[0] = f / aspect;
  out[5] = -f;
  out[10] = far_clip / (near_clip - far_clip);
  out[11] = -1.0f;
  out[14] = (far_clip * near_clip) / (near_clip - far_clip);
}

/* mat.h:365-375 */
void vko_mat4_ortho_zo_yinv(float left, float right, float bottom, float top, float near_clip,
                            float far_clip, float *out) {
  const float rl = (right - left);
  const float tb = (top - bottom);
  const float fn = (far_clip - near_clip);
  memset(out, 0, 16 * sizeof(float));
  out[0] = 2.0f / rl;
  out[5] = -2.0f / tb;
  out[10] = -1.0f / fn;
  out[12] = -((right + left) / rl);
  out[13] = +((top + bottom) / tb);
  out[14] = -(near_clip / fn);
  out[15] = 1.0f;
}

/* ======================================================================== */
/* Frustum                                                                   */
/* ======================================================================== */

/* vkr_frustum.c:12-31: normalise (n, d) when |n| > epsilon. vec3_scale
 * multiplies by inv_len per component (vec.h:620-622). */
static void vko_plane_from_vec4(float x, float y, float z, float w, float *plane) {
  float n[3] = {x, y, z};
  float d = w;
  const float len = vko_vec3_length(n);
  if (len > VKO_FLOAT_EPSILON) {
    const float inv_len = 1.0f / len;
    n[0] = n[0] * inv_len;
    n[1] = n[1] * inv_len;
    n[2] = n[2] * inv_len;
    d *= inv_len;
  }
  plane[0] = n[0];
  plane[1] = n[1];
  plane[2] = n[2];
  plane[3] = d;
}

/* vkr_frustum.c:44-66; rows via mat4_row (mat.h:927-930). */
void vko_frustum_from_matrix(const float *vp, float *planes) {
  float r[4][4];
  for (int row = 0; row < 4; ++row)
    for (int c = 0; c < 4; ++c)
      r[row][c] = vp[row + 4 * c];
  /* LEFT = r3 + r0, RIGHT = r3 - r0, BOTTOM = r3 + r1, TOP = r3 - r1,
   * NEAR = r2 (Vulkan 0 <= z <= w), FAR = r3 - r2 */
  vko_plane_from_vec4(r[3][0] + r[0][0], r[3][1] + r[0][1], r[3][2] + r[0][2],
                      r[3][3] + r[0][3], planes + 0);
  vko_plane_from_vec4(r[3][0] - r[0][0], r[3][1] - r[0][1], r[3][2] - r[0][2],
                      r[3][3] - r[0][3], planes + 4);
  vko_plane_from_vec4(r[3][0] + r[1][0], r[3][1] + r[1][1], r[3][2] + r[1][2],
                      r[3][3] + r[1][3], planes + 8);
  vko_plane_from_vec4(r[3][0] - r[1][0], r[3][1] - r[1][1], r[3][2] - r[1][2],
                      r[3][3] - r[1][3], planes + 12);
  vko_plane_from_vec4(r[2][0], r[2][1], r[2][2], r[2][3], planes + 16);
  vko_plane_from_vec4(r[3][0] - r[2][0], r[3][1] - r[2][1], r[3][2] - r[2][2],
                      r[3][3] - r[2][3], planes + 20);
}

/* vkr_frustum.c:33-38 */
void vko_frustum_from_view_projection(const float *view, const float *projection,
                                      float *planes) {
  float vp[16];
  vko_mat4_mul(projection, view, vp);
  vko_frustum_from_matrix(vp, planes);
}

/* vkr_frustum.c:68-84. NaN distance or radius compares false => visible. */
int vko_frustum_test_sphere(const float *planes, const float *center, float radius) {
  for (uint32_t i = 0; i < VKO_PLANE_COUNT; ++i) {
    const float *p = planes + 4 * i;
    const float dist = vko_vec3_dot(p, center) + p[3];
    if (dist < -radius)
      return 0;
  }
  return 1;
}

/* ======================================================================== */
/* Visibility helpers                                                        */
/* ======================================================================== */

static float vko_max_basis_length(const float *model) {
  const float l0 = vko_vec3_length(model + 0);
  const float l1 = vko_vec3_length(model + 4);
  const float l2 = vko_vec3_length(model + 8);
  return vko_max_f32(vko_max_f32(l0, l1), l2);
}

/* half-extent length: vec3_length(vec3_scale(vec3_sub(max, min), 0.5f))
 * (vkr_visibility.c:48, application.h:620-621,629) */
static float vko_half_extent_length(const float *mn, const float *mx) {
  float half[3];
  for (int i = 0; i < 3; ++i)
    half[i] = (mx[i] - mn[i]) * 0.5f;
  return vko_vec3_length(half);
}

/* vkr_visibility.c:37-50 */
void vko_submesh_sphere(const float *model, const float *center, const float *mn,
                        const float *mx, float *out_center, float *out_radius) {
  float c4[4];
  vko_mat4_mul_vec3(model, center, c4);
  out_center[0] = c4[0];
  out_center[1] = c4[1];
  out_center[2] = c4[2];
  const float max_scale = vko_max_basis_length(model);
  *out_radius = vko_half_extent_length(mn, mx) * max_scale;
}

/* vkr_visibility.h:29-39 */
uint32_t vko_state_bucket(uint32_t alpha_mode, uint32_t double_sided) {
  const int cutout = alpha_mode == VKO_ALPHA_CUTOUT;
  if (cutout)
    return double_sided ? 3u : 2u;
  return double_sided ? 1u : 0u;
}

/* vkr_picking_ids.h:35-42 with kind SCENE = 0; application.h:904-907 maps
 * render_id 0 to object id 0. */
uint32_t vko_picking_encode_scene_id(uint32_t render_id) {
  if (!render_id)
    return 0u;
  const uint32_t value_mask = (1u << 29) - 1u;
  if (render_id > value_mask - 1u)
    return 0u;
  return (0u << 29) | (render_id + 1u);
}

/* application.h:573-580 */
float vko_transparent_depth(const float *view, const float *model, const float *local_center) {
  float wc[4], vp[4];
  vko_mat4_mul_vec3(model, local_center, wc);
  const float p[4] = {wc[0], wc[1], wc[2], 1.0f};
  vko_mat4_mul_vec4(view, p, vp);
  const float depth = -vp[2];
  return depth > 0.0f ? depth : 0.0f;
}

/* application.h:582-587 */
uint64_t vko_pack_sort_key(float distance, uint32_t tie_breaker) {
  uint32_t bits = 0;
  memcpy(&bits, &distance, sizeof(bits));
  return ((uint64_t)bits << 32) | (uint64_t)tie_breaker;
}

/* vkr_mesh_manager.c:315-337 */
void vko_instance_world_bounds(const float *model, const float *local_center,
                               float local_radius, float *out4) {
  float c4[4];
  vko_mat4_mul_vec3(model, local_center, c4);
  out4[0] = c4[0];
  out4[1] = c4[1];
  out4[2] = c4[2];
  out4[3] = local_radius * vko_max_basis_length(model);
}

/* ======================================================================== */
/* Scene update                                                              */
/* ======================================================================== */

/* vkr_scene_system.c:580-604 (pass 1) and :1293-1336 (pass 2). */
uint64_t vko_scene_update(vko_scene_transform *t, uint64_t count, const uint32_t *topo_order,
                          uint64_t topo_count, uint32_t *dirty_out) {
  for (uint64_t i = 0; i < count; ++i) {
    vko_scene_transform *x = &t[i];
    x->flags &= (uint8_t)~VKO_WORLD_UPDATED;
    if (x->flags & VKO_DIRTY_LOCAL) {
      vko_local_from_trs(x->position, x->rotation, x->scale, x->local);
      x->flags &= (uint8_t)~VKO_DIRTY_LOCAL;
      x->flags |= VKO_DIRTY_WORLD;
    }
  }
  uint64_t dirty = 0;
  for (uint64_t i = 0; i < topo_count; ++i) {
    vko_scene_transform *x = &t[topo_order[i]];
    vko_scene_transform *parent = NULL;
    if (x->parent != (uint64_t)VKO_NO_PARENT) {
      parent = &t[x->parent];
      if (parent->flags & VKO_WORLD_UPDATED)
        x->flags |= VKO_DIRTY_WORLD;
    }
    if (!(x->flags & VKO_DIRTY_WORLD))
      continue;
    if (parent)
      vko_mat4_mul(parent->world, x->local, x->world);
    else
      memcpy(x->world, x->local, sizeof(x->world));
    x->flags &= (uint8_t)~VKO_DIRTY_WORLD;
    x->flags |= VKO_WORLD_UPDATED;
    if (dirty_out)
      dirty_out[dirty] = topo_order[i];
    dirty++;
  }
  return dirty;
}

/* BFS in the shape of vkr_scene_system.c:777-843: roots in storage order,
 * then each dequeued node's children in storage order. */
uint32_t vko_build_topo_order(const uint32_t *parent, uint64_t count, uint32_t *topo_order,
                              uint32_t *level_of) {
  if (!count)
    return 0;
  /* child lists via counting sort on parent */
  uint64_t *first = calloc(count + 1, sizeof(uint64_t));
  uint32_t *children = malloc(count * sizeof(uint32_t));
  uint8_t *visited = calloc(count, 1);
  for (uint64_t i = 0; i < count; ++i)
    if (parent[i] != VKO_NO_PARENT && parent[i] < count)
      first[parent[i] + 1]++;
  for (uint64_t i = 0; i < count; ++i)
    first[i + 1] += first[i];
  uint64_t *cursor = malloc(count * sizeof(uint64_t));
  memcpy(cursor, first, count * sizeof(uint64_t));
  for (uint64_t i = 0; i < count; ++i)
    if (parent[i] != VKO_NO_PARENT && parent[i] < count)
      children[cursor[parent[i]]++] = (uint32_t)i;
  uint64_t head = 0, tail = 0;
  uint32_t levels = 0;
  for (uint64_t i = 0; i < count; ++i) {
    if (parent[i] == VKO_NO_PARENT || parent[i] >= count) {
      topo_order[tail++] = (uint32_t)i;
      level_of[i] = 0;
      visited[i] = 1;
      levels = 1;
    }
  }
  while (head < tail) {
    const uint32_t e = topo_order[head++];
    for (uint64_t c = first[e]; c < first[e + 1]; ++c) {
      const uint32_t child = children[c];
      if (visited[child])
        continue;
      visited[child] = 1;
      level_of[child] = level_of[e] + 1;
      if (level_of[child] + 1 > levels)
        levels = level_of[child] + 1;
      topo_order[tail++] = child;
    }
  }
  if (!levels)
    levels = 1;
  for (uint64_t i = 0; i < count; ++i) {
    if (!visited[i]) {
      topo_order[tail++] = (uint32_t)i;
      level_of[i] = levels - 1;
    }
  }
  free(first);
  free(children);
  free(visited);
  free(cursor);
  return levels;
}

void vko_transforms_fill(vko_scene_transform *t, uint64_t count, const uint32_t *parent,
                         const float *pos4, const float *quat4, const float *scale4,
                         const float *local16, const float *world16, const uint8_t *flags) {
  memset(t, 0, count * sizeof(*t));
  for (uint64_t i = 0; i < count; ++i) {
    t[i].parent = parent ? (uint64_t)parent[i] : (uint64_t)VKO_NO_PARENT;
    if (pos4)
      memcpy(t[i].position, pos4 + 4 * i, 16);
    if (quat4)
      memcpy(t[i].rotation, quat4 + 4 * i, 16);
    if (scale4)
      memcpy(t[i].scale, scale4 + 4 * i, 16);
    if (local16)
      memcpy(t[i].local, local16 + 16 * i, 64);
    if (world16)
      memcpy(t[i].world, world16 + 16 * i, 64);
    if (flags)
      t[i].flags = flags[i];
  }
}

void vko_transforms_extract(const vko_scene_transform *t, uint64_t count, float *local16,
                            float *world16, uint8_t *flags) {
  for (uint64_t i = 0; i < count; ++i) {
    if (local16)
      memcpy(local16 + 16 * i, t[i].local, 64);
    if (world16)
      memcpy(world16 + 16 * i, t[i].world, 64);
    if (flags)
      flags[i] = t[i].flags;
  }
}

/* ======================================================================== */
/* Payload builder                                                           */
/* ======================================================================== */

/* VkrTransparentDrawCandidate, vkr_visibility.h:49-57 (112 B, sort_key @96) */
typedef struct vko_transparent_candidate {
  float model[16];
  vko_handle mesh;
  vko_handle geometry;
  vko_handle material;
  uint32_t submesh_index;
  uint32_t object_id;
  uint64_t sort_key;
  uint64_t pad;
} vko_transparent_candidate;
_Static_assert(sizeof(vko_transparent_candidate) == 112, "VkrTransparentDrawCandidate 112 B");
_Static_assert(offsetof(vko_transparent_candidate, sort_key) == 96, "sort_key @96");

/* vkr_visibility.c:5-13: descending by key */
static int vko_transparent_compare(const void *lhs, const void *rhs) {
  const vko_transparent_candidate *a = lhs;
  const vko_transparent_candidate *b = rhs;
  if (a->sort_key > b->sort_key)
    return -1;
  if (a->sort_key < b->sort_key)
    return 1;
  return 0;
}

typedef struct vko_source_bits {
  int world_transparent; /* vkr_visibility.h:18-26 */
  int shadow_alpha_tested;
  int transmissive;
  int double_sided;
  int bounds_valid;
} vko_source_bits;

static vko_source_bits vko_decode_matbits(uint32_t m) {
  vko_source_bits s;
  const uint32_t alpha = m & VKO_MAT_ALPHA_MASK;
  s.world_transparent = alpha == VKO_ALPHA_BLEND;
  s.shadow_alpha_tested = alpha == VKO_ALPHA_CUTOUT;
  s.transmissive = (m & VKO_MAT_TRANSMISSIVE) != 0;
  s.double_sided = (m & VKO_MAT_DOUBLE_SIDED) != 0;
  s.bounds_valid = (m & VKO_MAT_BOUNDS_VALID) != 0;
  return s;
}

static void vko_fill_candidate(vko_world_candidate *c, const float *model, const uint32_t *id,
                               const vko_source_bits *s, const float *b) {
  /* application.h:617-640 */
  memset(c, 0, sizeof(*c));
  c->mesh.id = id[0];
  c->mesh.generation = id[1];
  c->geometry.id = id[2];
  c->geometry.generation = id[3];
  c->submesh_index = id[4];
  c->material.id = id[5];
  c->material.generation = id[6];
  memcpy(c->instance.model, model, 64);
  c->instance.object_id = vko_picking_encode_scene_id(id[7]);
  c->local_bounding_sphere[0] = b[0];
  c->local_bounding_sphere[1] = b[1];
  c->local_bounding_sphere[2] = b[2];
  c->local_bounding_sphere[3] = vko_half_extent_length(b + 4, b + 8);
  c->state_bucket =
      vko_state_bucket(s->shadow_alpha_tested ? VKO_ALPHA_CUTOUT : VKO_ALPHA_OPAQUE,
                       (uint32_t)s->double_sided);
  c->flags = (s->bounds_valid ? VKO_CAND_BOUNDS_VALID : 0u) |
             (!s->transmissive && !s->world_transparent ? VKO_CAND_CAMERA_OPAQUE : 0u) |
             VKO_CAND_SHADOW_CASTER;
}

uint32_t vko_build_world_payload(const float *world16, const uint32_t *node,
                                 const uint32_t *ident, const uint32_t *matbits,
                                 const float *bounds, uint64_t n, const float *view,
                                 const float *projection, uint64_t capacity,
                                 uint64_t source_index_base,
                                 vko_world_candidate *gpu_candidates,
                                 vko_world_candidate *transmission_candidates,
                                 uint8_t *transparent_visible, vko_draw_item *draws,
                                 vko_instance_data *instances, uint64_t *sort_keys,
                                 uint32_t *sorted_source, vko_visibility_stats *out_stats,
                                 uint32_t *counts) {
  float camera_planes[24];
  vko_frustum_from_view_projection(view, projection, camera_planes); /* :676-677 */
  vko_visibility_stats stats = {0, 0, 0};
  uint32_t local_counts[4] = {0, 0, 0, 0};

  /* traversal 1 (:684-711): count, capacity gate */
  if (capacity && n > capacity) {
    local_counts[0] = (uint32_t)capacity + 1u;
    stats.objects_tested = (uint32_t)capacity + 1u;
    if (out_stats)
      *out_stats = stats;
    if (counts)
      memcpy(counts, local_counts, sizeof(local_counts));
    return 0;
  }
  local_counts[0] = (uint32_t)n;

  uint8_t *visible_local = transparent_visible;
  if (!visible_local)
    visible_local = malloc(n ? n : 1);
  memset(visible_local, 0, n);

  /* traversal 2 (:768-806): stats + blend-row camera cull */
  uint32_t camera_opaque = 0, transmission = 0, transparent = 0;
  for (uint64_t i = 0; i < n; ++i) {
    const vko_source_bits s = vko_decode_matbits(matbits[i]);
    const float *model = world16 + 16 * (uint64_t)(node ? node[i] : i);
    const float *b = bounds + 12 * i;
    stats.objects_tested++;
    stats.objects_without_bounds += s.bounds_valid ? 0u : 1u;
    camera_opaque += !s.transmissive && !s.world_transparent ? 1u : 0u;
    transmission += s.transmissive ? 1u : 0u;
    if (!s.transmissive && s.world_transparent) {
      int visible = 1;
      if (s.bounds_valid) {
        float center[3];
        float radius = 0.0f;
        vko_submesh_sphere(model, b, b + 4, b + 8, center, &radius);
        visible = vko_frustum_test_sphere(camera_planes, center, radius);
      }
      visible_local[i] = (uint8_t)visible;
      transparent += visible ? 1u : 0u;
      stats.objects_culled_camera += visible ? 0u : 1u;
    }
  }
  local_counts[1] = camera_opaque;
  local_counts[2] = transmission;
  local_counts[3] = transparent;

  /* traversal 3 (:853-938): emit */
  vko_transparent_candidate *tc =
      transparent ? malloc((uint64_t)transparent * sizeof(*tc)) : NULL;
  uint32_t ti = 0, xi = 0;
  for (uint64_t i = 0; i < n; ++i) {
    const vko_source_bits s = vko_decode_matbits(matbits[i]);
    const float *model = world16 + 16 * (uint64_t)(node ? node[i] : i);
    const float *b = bounds + 12 * i;
    const uint32_t *id = ident + 8 * i;
    vko_world_candidate c;
    vko_fill_candidate(&c, model, id, &s, b);
    if (gpu_candidates)
      gpu_candidates[i] = c;
    if (s.transmissive) {
      if (transmission_candidates)
        transmission_candidates[xi] = c;
      xi++;
    }
    if (!s.transmissive && s.world_transparent && visible_local[i]) {
      const float depth = vko_transparent_depth(view, model, b); /* :647-648 */
      vko_transparent_candidate *t = &tc[ti++];
      memset(t, 0, sizeof(*t));
      memcpy(t->model, model, 64);
      t->mesh = c.mesh;
      t->geometry = c.geometry;
      t->material = c.material;
      t->submesh_index = c.submesh_index;
      t->object_id = c.instance.object_id;
      t->sort_key =
          vko_pack_sort_key(depth, (uint32_t)(source_index_base + i) + 1u); /* :657-658 */
    }
  }

  /* :940-944 */
  if (transparent > 1u)
    qsort(tc, transparent, sizeof(*tc), vko_transparent_compare);
  for (uint32_t i = 0; i < transparent; ++i) { /* vkr_visibility.c:15-35 */
    if (draws) {
      memset(&draws[i], 0, sizeof(draws[i]));
      draws[i].mesh = tc[i].mesh;
      draws[i].geometry = tc[i].geometry;
      draws[i].submesh_index = tc[i].submesh_index;
      draws[i].material = tc[i].material;
      draws[i].instance_count = 1u;
      draws[i].first_instance = i;
      draws[i].sort_key = tc[i].sort_key;
    }
    if (instances) {
      memset(&instances[i], 0, sizeof(instances[i]));
      memcpy(instances[i].model, tc[i].model, 64);
      instances[i].object_id = tc[i].object_id;
    }
    if (sort_keys)
      sort_keys[i] = tc[i].sort_key;
    if (sorted_source)
      sorted_source[i] = (uint32_t)(tc[i].sort_key & 0xffffffffu) - 1u;
  }
  free(tc);
  if (!transparent_visible)
    free(visible_local);
  if (out_stats)
    *out_stats = stats;
  if (counts)
    memcpy(counts, local_counts, sizeof(local_counts));
  return transparent;
}

/* ======================================================================== */
/* Multi-view cull + compaction                                              */
/* ======================================================================== */

/* deferred.slang:240-303 */
static const vko_hzb *g_hzb = NULL;
void vko_set_hzb(const vko_hzb *hzb) { g_hzb = hzb; }

static uint32_t vko_min_u32(uint32_t a, uint32_t b) { return a < b ? a : b; }

/* deferred.slang:970-998 */
void vko_build_hzb(const float *depth, uint32_t sw, uint32_t sh, uint32_t width, uint32_t height,
                   uint32_t mip_count, float *mips) {
  const float *src = depth;
  float *dst = mips;
  for (uint32_t m = 0; m < mip_count; ++m) {
    const uint32_t w = (width >> m) ? (width >> m) : 1u, h = (height >> m) ? (height >> m) : 1u;
    for (uint32_t y = 0; y < h; ++y)
      for (uint32_t x = 0; x < w; ++x) {
        float out;
        if (m == 0) { /* source_is_depth == 2: a clamped copy */
          out = src[(uint64_t)vko_min_u32(y, sh - 1u) * sw + vko_min_u32(x, sw - 1u)];
        } else {
          out = 0.0f;
          for (uint32_t dy = 0; dy < 2u; ++dy)
            for (uint32_t dx = 0; dx < 2u; ++dx) {
              const uint32_t px = vko_min_u32(x * 2u + dx, sw - 1u);
              const uint32_t py = vko_min_u32(y * 2u + dy, sh - 1u);
              out = fmaxf(out, src[(uint64_t)py * sw + px]);
            }
        }
        dst[(uint64_t)y * w + x] = out;
      }
    src = dst;
    sw = w;
    sh = h;
    dst += (uint64_t)w * h;
  }
}

static float vko_clamp01(float x) { return fminf(fmaxf(x, 0.0f), 1.0f); }

int vko_hzb_occluded(const vko_hzb *h, const float *center, float world_radius) {
  if (!h || !h->mips || h->mip_count == 0u || h->width == 0u || h->height == 0u)
    return 0;
  const float radius = fmaxf(world_radius, 0.0f);
  float ndc_min[2] = {1e30f, 1e30f}, ndc_max[2] = {-1e30f, -1e30f};
  float nearest_depth = 1.0f;
  for (uint32_t corner = 0; corner < 8u; ++corner) {
    const float p[4] = {center[0] + ((corner & 1u) ? 1.0f : -1.0f) * radius,
                        center[1] + ((corner & 2u) ? 1.0f : -1.0f) * radius,
                        center[2] + ((corner & 4u) ? 1.0f : -1.0f) * radius, 1.0f};
    float clip[4];
    vko_mat4_mul_vec4(h->view_projection, p, clip);
    if (!isfinite(clip[0]) || !isfinite(clip[1]) || !isfinite(clip[2]) || !isfinite(clip[3]) ||
        clip[3] <= 1e-6f)
      return 0;
    const float nx = clip[0] / clip[3], ny = clip[1] / clip[3], nz = clip[2] / clip[3];
    ndc_min[0] = fminf(ndc_min[0], nx);
    ndc_min[1] = fminf(ndc_min[1], ny);
    ndc_max[0] = fmaxf(ndc_max[0], nx);
    ndc_max[1] = fmaxf(ndc_max[1], ny);
    nearest_depth = fminf(nearest_depth, nz);
  }
  if (nearest_depth <= 0.0f || ndc_min[0] > 1.0f || ndc_min[1] > 1.0f || ndc_max[0] < -1.0f ||
      ndc_max[1] < -1.0f)
    return 0;
  const float uv_min[2] = {vko_clamp01(ndc_min[0] * 0.5f + 0.5f), vko_clamp01(-ndc_max[1] * 0.5f + 0.5f)};
  const float uv_max[2] = {vko_clamp01(ndc_max[0] * 0.5f + 0.5f), vko_clamp01(-ndc_min[1] * 0.5f + 0.5f)};
  const uint32_t extent[2] = {h->width, h->height};
  uint32_t base_min[2], base_max[2];
  for (int k = 0; k < 2; ++k) {
    base_min[k] = vko_min_u32((uint32_t)floorf(uv_min[k] * (float)extent[k]), extent[k] - 1u);
    base_max[k] = vko_min_u32((uint32_t)floorf(uv_max[k] * (float)extent[k]), extent[k] - 1u);
  }
  const uint32_t span_x = base_max[0] - base_min[0] + 1u, span_y = base_max[1] - base_min[1] + 1u;
  const uint32_t max_span = span_x > span_y ? span_x : span_y;
  uint32_t mip = 0;
  if (max_span > 1u) /* floor(log2(float(max_span))): exact as a bit scan below 2^24 */
    mip = 31u - (uint32_t)__builtin_clz(max_span);
  mip = vko_min_u32(mip, h->mip_count - 1u);
  uint64_t offset = 0;
  for (uint32_t m = 0; m < mip; ++m) {
    const uint32_t w = (h->width >> m) ? (h->width >> m) : 1u, ht = (h->height >> m) ? (h->height >> m) : 1u;
    offset += (uint64_t)w * ht;
  }
  const uint32_t mw = (h->width >> mip) ? (h->width >> mip) : 1u;
  const uint32_t mh = (h->height >> mip) ? (h->height >> mip) : 1u;
  const uint32_t x0 = vko_min_u32(base_min[0] >> mip, mw - 1u), x1 = vko_min_u32(base_max[0] >> mip, mw - 1u);
  const uint32_t y0 = vko_min_u32(base_min[1] >> mip, mh - 1u), y1 = vko_min_u32(base_max[1] >> mip, mh - 1u);
  for (uint32_t y = y0; y <= y1; ++y)
    for (uint32_t x = x0; x <= x1; ++x) {
      const float occluder_depth = h->mips[offset + (uint64_t)y * mw + x];
      if (nearest_depth <= occluder_depth + h->depth_epsilon)
        return 0;
    }
  return 1;
}

static void vko_multiview_core(const float *const *models, const float *sphere4,
                               uint64_t sphere_stride, const uint32_t *flags,
                               uint64_t flags_stride, const uint32_t *bucket,
                               uint64_t bucket_stride, uint64_t n, const float *planes,
                               const uint32_t *required_flags, uint32_t views,
                               uint32_t partition_capacity, uint64_t visible_stride,
                               uint64_t index_base, float tie_eps, uint32_t *classification,
                               vko_compaction_state *states, uint32_t *visible,
                               uint64_t *near_plane_ties, const float *world16,
                               const uint32_t *node) {
  const uint32_t cap = partition_capacity ? partition_capacity : 0xffffffffu;
  uint32_t *cls_row = classification ? NULL : malloc((n ? n : 1) * sizeof(uint32_t));
  for (uint32_t v = 0; v < views; ++v) {
    uint32_t *cls = classification ? classification + (uint64_t)v * n : cls_row;
    vko_compaction_state st;
    memset(&st, 0, sizeof(st));
    uint64_t ties = 0;
    const float *vp = planes + 24 * (uint64_t)v;
    /* classify: deferred.slang:305-352 (HZB branch excluded) */
    for (uint64_t i = 0; i < n; ++i) {
      const uint32_t f = flags[i * flags_stride];
      const uint32_t b = bucket[i * bucket_stride];
      const uint32_t req = required_flags[v];
      if (b >= VKO_BUCKET_COUNT || (f & req) != req) {
        cls[i] = VKO_CULLED;
        continue;
      }
      if (f & VKO_CAND_BOUNDS_VALID) {
        const float *model =
            models ? models[i] : world16 + 16 * (uint64_t)(node ? node[i] : i);
        const float *ls = sphere4 + i * sphere_stride;
        float c4[4];
        vko_mat4_mul_vec3(model, ls, c4);
        const float radius = ls[3] * vko_max_basis_length(model);
        int vis = vko_frustum_test_sphere(vp, c4, radius);
        if (near_plane_ties && tie_eps > 0.0f) {
          int tie = 0;
          for (uint32_t p = 0; p < VKO_PLANE_COUNT; ++p) {
            const float *pl = vp + 4 * p;
            const float dist = vko_vec3_dot(pl, c4) + pl[3];
            const float mag = fabsf(pl[3]) > 1.0f ? fabsf(pl[3]) : 1.0f;
            if (fabsf(dist + radius) <= tie_eps * mag)
              tie = 1;
          }
          ties += (uint64_t)tie;
        }
        if (!vis) {
          cls[i] = VKO_CULLED;
          continue;
        }
        /* deferred.slang:340-347: view 0 only, rows with valid bounds only */
        if (v == 0u && g_hzb && vko_hzb_occluded(g_hzb, c4, radius)) {
          cls[i] = VKO_CULLED;
          st.occlusion_culled_count++;
          continue;
        }
      }
      cls[i] = b;
      st.bucket_counts[b]++;
    }
    /* prefix: deferred.slang:354-377 */
    uint32_t cursor = 0, overflow = 0;
    for (uint32_t b = 0; b < VKO_BUCKET_COUNT; ++b) {
      const uint32_t count = st.bucket_counts[b];
      const uint32_t written = count < cap ? count : cap;
      st.execution_ranges[b][0] = cursor;
      st.execution_ranges[b][1] = written;
      st.bucket_cursors[b] = 0;
      cursor += written;
      overflow += count - written;
    }
    st.visible_count = cursor;
    st.overflow_count = overflow;
    /* encode: deferred.slang:379-420, candidate order inside a bucket */
    for (uint64_t i = 0; i < n; ++i) {
      const uint32_t b = cls[i];
      if (b >= VKO_BUCKET_COUNT)
        continue;
      const uint32_t local_index = st.bucket_cursors[b]++;
      if (local_index >= cap)
        continue;
      const uint64_t slot = (uint64_t)st.execution_ranges[b][0] + local_index;
      if (visible && slot < visible_stride)
        visible[(uint64_t)v * visible_stride + slot] = (uint32_t)(index_base + i);
    }
    if (states)
      states[v] = st;
    if (near_plane_ties)
      near_plane_ties[v] = ties;
  }
  free(cls_row);
}

void vko_multiview_cull(const vko_world_candidate *c, uint64_t n, const float *planes,
                        const uint32_t *required_flags, uint32_t views,
                        uint32_t partition_capacity, uint64_t visible_stride,
                        uint64_t index_base, float tie_eps, uint32_t *classification,
                        vko_compaction_state *states, uint32_t *visible,
                        uint64_t *near_plane_ties) {
  const float **models = malloc((n ? n : 1) * sizeof(float *));
  for (uint64_t i = 0; i < n; ++i)
    models[i] = c[i].instance.model;
  const uint64_t stride_f = sizeof(*c) / sizeof(float);
  const uint64_t stride_u = sizeof(*c) / sizeof(uint32_t);
  vko_multiview_core(models, n ? c[0].local_bounding_sphere : NULL, stride_f,
                     n ? &c[0].flags : NULL, stride_u, n ? &c[0].state_bucket : NULL, stride_u,
                     n, planes, required_flags, views, partition_capacity, visible_stride,
                     index_base, tie_eps, classification, states, visible, near_plane_ties,
                     NULL, NULL);
  free(models);
}

void vko_multiview_cull_flat(const float *world16, const uint32_t *node, const float *sphere4,
                             const uint32_t *flags, const uint32_t *bucket, uint64_t n,
                             const float *planes, const uint32_t *required_flags,
                             uint32_t views, uint32_t partition_capacity,
                             uint64_t visible_stride, uint64_t index_base, float tie_eps,
                             uint32_t *classification, vko_compaction_state *states,
                             uint32_t *visible, uint64_t *near_plane_ties) {
  vko_multiview_core(NULL, sphere4, 4, flags, 1, bucket, 1, n, planes, required_flags, views,
                     partition_capacity, visible_stride, index_base, tie_eps, classification,
                     states, visible, near_plane_ties, world16, node);
}

int vko_pack_gpu_candidates(const vko_world_candidate *source, uint32_t count,
                            const vko_geometry_entry *geometries, uint32_t geometry_capacity,
                            const vko_submesh_range *submeshes,
                            const vko_material_entry *materials, uint32_t material_capacity,
                            vko_gpu_candidate_row *rows, vko_instance_data *instances,
                            uint32_t *packed_source, uint32_t *counts4) {
  uint32_t packed = 0, no_geometry = 0, no_material = 0, bad_submesh = 0;
  for (uint32_t i = 0; i < count; ++i) {
    const vko_world_candidate *c = &source[i];
    /* vkr_vk_resolve_geometry / _material, vkr_vulkan_draws.c:321-342 */
    const vko_geometry_entry *g = NULL;
    if (c->geometry.id != 0u && c->geometry.id <= geometry_capacity) {
      const vko_geometry_entry *e = &geometries[c->geometry.id - 1u];
      if (e->resolvable && e->generation == c->geometry.generation)
        g = e;
    }
    const vko_material_entry *m = NULL;
    if (c->material.id != 0u && c->material.id <= material_capacity) {
      const vko_material_entry *e = &materials[c->material.id - 1u];
      if (e->live && e->generation == c->material.generation)
        m = e;
    }
    if (!g) {
      no_geometry++;
      continue;
    }
    if (!m) {
      no_material++;
      continue;
    }
    if (c->submesh_index >= g->submesh_count) {
      bad_submesh++;
      continue;
    }
    const vko_submesh_range *sub = &submeshes[g->submesh_offset + c->submesh_index];
    vko_gpu_candidate_row *r = &rows[packed];
    r->geometry_index = c->geometry.id - 1u;
    r->material_index = m->slot_index;
    r->instance_index = packed;
    r->first_index = g->first_index + sub->first_index;
    r->index_count = sub->index_count;
    r->vertex_offset = sub->vertex_offset;
    r->state_bucket = c->state_bucket;
    r->flags = c->flags;
    memcpy(r->local_bounding_sphere, c->local_bounding_sphere, 16);
    instances[packed] = c->instance;
    if (packed_source)
      packed_source[packed] = i;
    packed++;
  }
  counts4[0] = packed;
  counts4[1] = no_geometry;
  counts4[2] = no_material;
  counts4[3] = bad_submesh;
  return bad_submesh ? 0 : 1;
}

void vko_encode_commands(const vko_gpu_candidate_row *rows, const uint32_t *visible,
                         const vko_compaction_state *state, uint32_t view,
                         uint32_t partition_capacity, vko_indirect_command *commands) {
  for (uint32_t b = 0; b < VKO_BUCKET_COUNT; ++b) {
    const uint32_t off = state->execution_ranges[b][0];
    const uint32_t cnt = state->execution_ranges[b][1];
    for (uint32_t k = 0; k < cnt; ++k) {
      const vko_gpu_candidate_row *c = &rows[visible[off + k]];
      vko_indirect_command *o =
          &commands[((uint64_t)view * VKO_BUCKET_COUNT + b) * partition_capacity + k];
      o->index_count = c->index_count;
      o->instance_count = 1u;
      o->first_index = c->first_index;
      o->vertex_offset = c->vertex_offset;
      o->first_instance = 0u;
    }
  }
}

void vko_encode_visible_rows(const vko_gpu_candidate_row *rows, const uint32_t *visible,
                             const vko_compaction_state *state, vko_gpu_visible_row *out) {
  for (uint32_t b = 0; b < VKO_BUCKET_COUNT; ++b) {
    const uint32_t off = state->execution_ranges[b][0];
    const uint32_t cnt = state->execution_ranges[b][1];
    for (uint32_t k = 0; k < cnt; ++k) {
      const vko_gpu_candidate_row *c = &rows[visible[off + k]];
      vko_gpu_visible_row *o = &out[off + k];
      o->geometry_index = c->geometry_index;
      o->material_index = c->material_index;
      o->instance_index = c->instance_index;
      o->first_index = c->first_index;
      o->index_count = c->index_count;
      o->vertex_offset = c->vertex_offset;
      o->state_bucket = b;
      o->flags = c->flags;
    }
  }
}

/* vkr_scene_system.c:1935-1977 */
int vko_entity_is_visible(const uint32_t *parent, const uint8_t *spec, uint64_t n, uint64_t node) {
  uint64_t depth = 0;
  uint64_t current = node;
  while (current < n && depth < n) { /* max_depth = directory capacity */
    const uint8_t sp = spec[current];
    if (sp & 1u) {
      if (!(sp & 2u))
        return 0;
      if (!(sp & 4u))
        return 1;
    }
    const uint32_t p = parent[current];
    if (p == VKO_NO_PARENT)
      break;
    current = p;
    depth++;
  }
  return 1;
}

/* vkr_vulkan_draws.c:9-44 */
static uint64_t vko_hash_bytes(uint64_t hash, const void *bytes, uint64_t size) {
  const uint8_t *data = bytes;
  for (uint64_t i = 0; i < size; ++i) {
    hash ^= data[i];
    hash *= 1099511628211ull;
  }
  return hash;
}

uint64_t vko_candidate_epoch(const vko_world_candidate *candidates, uint32_t count) {
  uint64_t hash = 1469598103934665603ull;
  uint32_t occluders = 0;
  for (uint32_t i = 0; i < count; ++i) {
    const vko_world_candidate *c = &candidates[i];
    if ((c->flags & VKO_CAND_CAMERA_OPAQUE) == 0u)
      continue;
    occluders++;
    hash = vko_hash_bytes(hash, &c->mesh.id, 4);
    hash = vko_hash_bytes(hash, &c->mesh.generation, 4);
    hash = vko_hash_bytes(hash, &c->geometry.id, 4);
    hash = vko_hash_bytes(hash, &c->geometry.generation, 4);
    hash = vko_hash_bytes(hash, &c->submesh_index, 4);
    hash = vko_hash_bytes(hash, &c->material.id, 4);
    hash = vko_hash_bytes(hash, &c->material.generation, 4);
    hash = vko_hash_bytes(hash, c->instance.model, 64);
    hash = vko_hash_bytes(hash, c->local_bounding_sphere, 16);
    hash = vko_hash_bytes(hash, &c->state_bucket, 4);
    hash = vko_hash_bytes(hash, &c->flags, 4);
  }
  return vko_hash_bytes(hash, &occluders, 4);
}

void vko_merge_runs_desc(const uint64_t *const *runs, const uint64_t *lengths, uint32_t g,
                         uint64_t *out) {
  uint64_t pos[64];
  if (g > 64)
    g = 64;
  memset(pos, 0, sizeof(pos));
  uint64_t o = 0;
  for (;;) {
    int best = -1;
    uint64_t best_key = 0;
    for (uint32_t r = 0; r < g; ++r) {
      if (pos[r] >= lengths[r])
        continue;
      const uint64_t k = runs[r][pos[r]];
      if (best < 0 || k > best_key) {
        best = (int)r;
        best_key = k;
      }
    }
    if (best < 0)
      break;
    out[o++] = best_key;
    pos[best]++;
  }
}

/* ======================================================================== */
/* Whole-frame single-thread baseline                                        */
/* ======================================================================== */

struct vko_baseline_scene {
  uint64_t n_nodes, n_cand;
  uint32_t views;
  vko_scene_transform *transforms;
  uint32_t *topo;
  uint32_t *dirty;
  float *world16; /* bridge-synced instance models (VkrMeshInstance.model) */
  uint32_t *node;
  uint32_t *ident;
  uint32_t *matbits;
  float *bounds;
  float planes[64 * 24];
  uint32_t required[64];
  float view[16], projection[16];
  vko_world_candidate *cand;
  vko_world_candidate *trans;
  uint8_t *tvis;
  vko_draw_item *draws;
  vko_instance_data *inst;
  uint32_t *cls;
  uint32_t *visible;
  vko_compaction_state states[64];
};

static void *vko_dup(const void *p, size_t bytes) {
  void *q = malloc(bytes ? bytes : 1);
  if (p)
    memcpy(q, p, bytes);
  return q;
}

vko_baseline_scene *vko_baseline_create(uint64_t n_nodes, const uint32_t *parent,
                                        const float *local16, uint64_t n_cand,
                                        const uint32_t *node, const uint32_t *ident,
                                        const uint32_t *matbits, const float *bounds,
                                        uint32_t views, const float *planes,
                                        const uint32_t *required_flags, const float *view,
                                        const float *projection) {
  if (views > 64)
    return NULL;
  vko_baseline_scene *s = calloc(1, sizeof(*s));
  s->n_nodes = n_nodes;
  s->n_cand = n_cand;
  s->views = views;
  s->transforms = malloc((n_nodes ? n_nodes : 1) * sizeof(vko_scene_transform));
  vko_transforms_fill(s->transforms, n_nodes, parent, NULL, NULL, NULL, local16, NULL, NULL);
  s->topo = malloc((n_nodes ? n_nodes : 1) * sizeof(uint32_t));
  uint32_t *level = malloc((n_nodes ? n_nodes : 1) * sizeof(uint32_t));
  vko_build_topo_order(parent, n_nodes, s->topo, level);
  free(level);
  s->dirty = malloc((n_nodes ? n_nodes : 1) * sizeof(uint32_t));
  s->world16 = malloc((n_nodes ? n_nodes : 1) * 64);
  s->node = node ? vko_dup(node, n_cand * 4) : NULL;
  s->ident = vko_dup(ident, n_cand * 32);
  s->matbits = vko_dup(matbits, n_cand * 4);
  s->bounds = vko_dup(bounds, n_cand * 48);
  memcpy(s->planes, planes, (size_t)views * 24 * sizeof(float));
  memcpy(s->required, required_flags, (size_t)views * sizeof(uint32_t));
  memcpy(s->view, view, 64);
  memcpy(s->projection, projection, 64);
  s->cand = malloc((n_cand ? n_cand : 1) * sizeof(vko_world_candidate));
  s->trans = malloc((n_cand ? n_cand : 1) * sizeof(vko_world_candidate));
  s->tvis = malloc(n_cand ? n_cand : 1);
  s->draws = malloc((n_cand ? n_cand : 1) * sizeof(vko_draw_item));
  s->inst = malloc((n_cand ? n_cand : 1) * sizeof(vko_instance_data));
  s->cls = malloc((n_cand ? n_cand : 1) * (uint64_t)views * sizeof(uint32_t));
  s->visible = malloc((n_cand ? n_cand : 1) * (uint64_t)views * sizeof(uint32_t));
  return s;
}

uint64_t vko_baseline_frame(vko_baseline_scene *s) {
  /* every node moved this frame */
  for (uint64_t i = 0; i < s->n_nodes; ++i)
    s->transforms[i].flags |= VKO_DIRTY_WORLD;
  const uint64_t dirty =
      vko_scene_update(s->transforms, s->n_nodes, s->topo, s->n_nodes, s->dirty);
  /* render bridge sync (vkr_scene_system.c:2123-2169): inst.model = world */
  for (uint64_t k = 0; k < dirty; ++k) {
    const uint32_t e = s->dirty[k];
    memcpy(s->world16 + 16 * (uint64_t)e, s->transforms[e].world, 64);
  }
  uint32_t counts[4];
  vko_visibility_stats stats;
  const uint32_t t = vko_build_world_payload(
      s->world16, s->node, s->ident, s->matbits, s->bounds, s->n_cand, s->view,
      s->projection, 0, 0, s->cand, s->trans, s->tvis, s->draws, s->inst, NULL, NULL, &stats,
      counts);
  vko_multiview_cull(s->cand, s->n_cand, s->planes, s->required, s->views, 0, s->n_cand, 0,
                     0.0f, s->cls, s->states, s->visible, NULL);
  uint64_t sum = dirty + t + stats.objects_culled_camera;
  for (uint32_t v = 0; v < s->views; ++v)
    sum += s->states[v].visible_count;
  if (t)
    sum ^= s->draws[t - 1].sort_key;
  return sum;
}

void vko_baseline_destroy(vko_baseline_scene *s) {
  if (!s)
    return;
  free(s->transforms);
  free(s->topo);
  free(s->dirty);
  free(s->world16);
  free(s->node);
  free(s->ident);
  free(s->matbits);
  free(s->bounds);
  free(s->cand);
  free(s->trans);
  free(s->tvis);
  free(s->draws);
  free(s->inst);
  free(s->cls);
  free(s->visible);
  free(s);
}
