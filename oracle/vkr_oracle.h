/*
 * vkr_oracle.h - CPU restatement of the reference's per-frame scene-processing
 * path (transform propagation -> sphere frustum cull -> draw keys, sort, emit).
 *
 * TEST INFRASTRUCTURE ONLY. This is the parity checker and the timed CPU
 * baseline. Nothing in the shipped GPU path may call into it.
 *
 * Parity pinning: every function below is checked (tests/test_oracle_*.py)
 * against (a) the reference's own known-answer tests re-expressed as fixtures
 * and (b) golden vectors produced by the reference's unmodified math /
 * frustum / visibility translation units compiled into oracle/_ref (see
 * oracle/Makefile, oracle/gen_golden.py). The callers the reference cannot be
 * linked for here (vkr_scene_update, application_build_world_payload, the
 * Slang classify/prefix/encode shaders) are restated from their source text;
 * their goldens are produced by driving the compiled reference leaf functions
 * in the reference's loop order.
 *
 * All matrices are column-major float[16] (element = m[col*4 + row]), as the
 * reference's Mat4 (lib/src/math/mat.h:214-222). Vec3 arguments are float[3];
 * arrays of Vec3 use the reference's padded 16-byte stride where noted.
 *
 * Build with -ffp-contract=off: the reference fuses multiply-add only where it
 * calls vkr_simd_fma_f32x4 (lib/src/math/vkr_simd.h:867-873).
 */
#ifndef VKR_ORACLE_H
#define VKR_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- scene transform flags: lib/src/renderer/systems/vkr_scene_system.h:60-65 */
#define VKO_DIRTY_LOCAL 0x01u
#define VKO_DIRTY_WORLD 0x02u
#define VKO_DIRTY_HIERARCHY 0x04u
#define VKO_WORLD_UPDATED 0x08u

/* ---- candidate flags: lib/src/renderer/vkr_render_packet.h:113-118 */
#define VKO_CAND_BOUNDS_VALID 0x1u
#define VKO_CAND_CAMERA_OPAQUE 0x2u
#define VKO_CAND_SHADOW_CASTER 0x4u

/* ---- material alpha mode: lib/src/renderer/resources/vkr_resources.h:129-133 */
#define VKO_ALPHA_OPAQUE 0u
#define VKO_ALPHA_CUTOUT 1u
#define VKO_ALPHA_BLEND 2u

/* material bit word used by the candidate source rows (ours, not the
 * reference's): bits 0-1 alpha mode, bit 2 transmissive, bit 3 double sided,
 * bit 4 bounds valid. */
#define VKO_MAT_ALPHA_MASK 0x3u
#define VKO_MAT_TRANSMISSIVE 0x4u
#define VKO_MAT_DOUBLE_SIDED 0x8u
#define VKO_MAT_BOUNDS_VALID 0x10u

#define VKO_NO_PARENT 0xffffffffu
#define VKO_CULLED 0xffffffffu
#define VKO_BUCKET_COUNT 4u
#define VKO_PLANE_COUNT 6u

/* reference capacities: lib/src/renderer/vkr_gpu_abi.h:7-9 */
#define VKO_REF_CANDIDATE_CAPACITY 262144u
#define VKO_REF_INSTANCE_MAX 65536u

/* Byte layouts of the reference's packet rows (sizes/offsets probed from the
 * reference headers; asserted again in tests against oracle/_ref). */
typedef struct vko_handle {
  uint32_t id;
  uint32_t generation;
} vko_handle;

/* VkrInstanceDataGPU, lib/src/renderer/vkr_gpu_abi.h:34-38 (80 B) */
typedef struct vko_instance_data {
  float model[16];
  uint32_t object_id;
  uint32_t reserved[3];
} vko_instance_data;

/* VkrWorldDrawCandidate, lib/src/renderer/vkr_render_packet.h:126-135 (144 B) */
typedef struct vko_world_candidate {
  vko_handle mesh;        /* 0 */
  vko_handle geometry;    /* 8 */
  uint32_t submesh_index; /* 16 */
  vko_handle material;    /* 20 */
  uint32_t pad0;          /* 28 */
  vko_instance_data instance;     /* 32 */
  float local_bounding_sphere[4]; /* 112 */
  uint32_t state_bucket;          /* 128 */
  uint32_t flags;                 /* 132 */
  uint32_t pad1[2];               /* 136 */
} vko_world_candidate;

/* VkrDrawItem, lib/src/renderer/vkr_render_packet.h:100-111 (48 B) */
typedef struct vko_draw_item {
  vko_handle mesh;         /* 0 */
  vko_handle geometry;     /* 8 */
  uint32_t submesh_index;  /* 16 */
  vko_handle material;     /* 20 */
  uint32_t instance_count; /* 28 */
  uint32_t first_instance; /* 32 */
  uint32_t pad0;           /* 36 */
  uint64_t sort_key;       /* 40 */
} vko_draw_item;

/* VkrGpuDrawCompactionState, lib/src/renderer/vkr_gpu_abi.h:95-103 (80 B) */
typedef struct vko_compaction_state {
  uint32_t execution_ranges[4][2];
  uint32_t bucket_counts[4];
  uint32_t bucket_cursors[4];
  uint32_t visible_count;
  uint32_t overflow_count;
  uint32_t resolve_invalid_count;
  uint32_t occlusion_culled_count;
} vko_compaction_state;

/* VkrGpuCandidateDrawRow, lib/src/renderer/vkr_gpu_abi.h:70-80 (48 B) */
typedef struct vko_gpu_candidate_row {
  uint32_t geometry_index;
  uint32_t material_index;
  uint32_t instance_index;
  uint32_t first_index;
  uint32_t index_count;
  int32_t vertex_offset;
  uint32_t state_bucket;
  uint32_t flags;
  float local_bounding_sphere[4];
} vko_gpu_candidate_row;

/* VkrGpuVisibleDrawRow, lib/src/renderer/vkr_gpu_abi.h:83-92 (32 B) */
typedef struct vko_gpu_visible_row {
  uint32_t geometry_index;
  uint32_t material_index;
  uint32_t instance_index;
  uint32_t first_index;
  uint32_t index_count;
  int32_t vertex_offset;
  uint32_t state_bucket;
  uint32_t flags;
} vko_gpu_visible_row;

/* VkrVisibilityStats, lib/src/renderer/vkr_visibility.h:42-46 */
typedef struct vko_visibility_stats {
  uint32_t objects_tested;
  uint32_t objects_culled_camera;
  uint32_t objects_without_bounds;
} vko_visibility_stats;

/* SceneTransform, lib/src/renderer/systems/vkr_scene_system.h:79-89 (208 B).
 * `parent` is the index of the parent record (the reference stores a 64-bit
 * entity id and resolves it through the ECS directory). */
typedef struct vko_scene_transform {
  float position[4]; /* 0 */
  float rotation[4]; /* 16  (x,y,z,w) */
  float scale[4];    /* 32 */
  uint64_t parent;   /* 48  VKO_NO_PARENT (zero-extended) = root */
  uint64_t pad0;     /* 56 */
  float local[16];   /* 64 */
  float world[16];   /* 128 */
  uint8_t flags;     /* 192 */
  uint8_t pad1[15];
} vko_scene_transform;

/* ------------------------------------------------------------------------
 * Leaf math (lib/src/math/mat.h, vec.h, vkr_simd.h x86 path)
 * ---------------------------------------------------------------------- */
void vko_mat4_mul(const float *a, const float *b, float *out);            /* mat.h:1145-1163 */
void vko_mat4_mul_vec3(const float *m, const float *v3, float *out4);     /* mat.h:1295-1300 */
void vko_mat4_mul_vec4(const float *m, const float *v4, float *out4);     /* mat.h:1260-1264 */
float vko_vec3_dot(const float *a, const float *b);                       /* vkr_simd.h:914-917 */
float vko_vec3_length(const float *v);                                    /* vec.h:663-675 */
void vko_quat_to_mat4(const float *q_xyzw, float *out);                   /* mat.h:1538-1560 */
void vko_mat4_translate(const float *v3, float *out);                     /* mat.h:479-482 */
void vko_mat4_scale(const float *v3, float *out);                         /* mat.h:507-510 */
void vko_local_from_trs(const float *pos3, const float *quat_xyzw,
                        const float *scale3, float *out);                 /* vkr_scene_system.c:555-561 */
void vko_mat4_perspective(float fov, float aspect, float near_clip,
                          float far_clip, float *out);                    /* mat.h:405-416 */
void vko_mat4_ortho_zo_yinv(float l, float r, float b, float t, float n,
                            float f, float *out);                         /* mat.h:365-375 */

/* ------------------------------------------------------------------------
 * Frustum (lib/src/math/vkr_frustum.c). planes = 6 x (nx,ny,nz,d), order
 * LEFT,RIGHT,BOTTOM,TOP,NEAR,FAR.
 * ---------------------------------------------------------------------- */
void vko_frustum_from_matrix(const float *view_projection, float *planes24);        /* :44-66 */
void vko_frustum_from_view_projection(const float *view, const float *projection,
                                      float *planes24);                             /* :33-38 */
int vko_frustum_test_sphere(const float *planes24, const float *center3,
                            float radius);                                          /* :68-84 */

/* ------------------------------------------------------------------------
 * Visibility helpers (lib/src/renderer/vkr_visibility.{h,c})
 * ---------------------------------------------------------------------- */
void vko_submesh_sphere(const float *model, const float *center3, const float *min3,
                        const float *max3, float *out_center3, float *out_radius);  /* .c:37-50 */
uint32_t vko_state_bucket(uint32_t alpha_mode, uint32_t double_sided);              /* .h:29-39 */
uint32_t vko_picking_encode_scene_id(uint32_t render_id); /* systems/vkr_picking_ids.h:35-42 + application.h:904-907 */
float vko_transparent_depth(const float *view, const float *model,
                            const float *local_center3);                            /* application.h:573-580 */
uint64_t vko_pack_sort_key(float distance, uint32_t tie_breaker);                   /* application.h:582-587 */

/* ------------------------------------------------------------------------
 * Scene update (vkr_scene_update, vkr_scene_system.c:580-604,1254-1337).
 * `topo_order` lists record indices parents-first (the reference's BFS order,
 * :783-833). `dirty_out` receives the render-dirty list (record indices, in
 * visit order); returns its length.
 * ---------------------------------------------------------------------- */
uint64_t vko_scene_update(vko_scene_transform *transforms, uint64_t count,
                          const uint32_t *topo_order, uint64_t topo_count,
                          uint32_t *dirty_out);

/* BFS from roots in ascending index order; children visited in ascending
 * index order. Writes topo_order[count] and level_of[count]; returns number
 * of levels. Cycles (never reached) are appended at the end with level = levels-1
 * like vkr_scene_system.c:835-843 appends unvisited entities. */
uint32_t vko_build_topo_order(const uint32_t *parent, uint64_t count,
                              uint32_t *topo_order, uint32_t *level_of);

/* Convenience for array-shaped callers: fills AoS records from flat arrays
 * (position/scale with the reference's padded 4-float Vec3 stride). Any input
 * pointer may be NULL (field left zero). */
void vko_transforms_fill(vko_scene_transform *t, uint64_t count, const uint32_t *parent,
                         const float *pos4, const float *quat4, const float *scale4,
                         const float *local16, const float *world16, const uint8_t *flags);
void vko_transforms_extract(const vko_scene_transform *t, uint64_t count, float *local16,
                            float *world16, uint8_t *flags);

/* World bounds of an instance (vkr_mesh_manager.c:315-337): out4 = (cx,cy,cz,r). */
void vko_instance_world_bounds(const float *model, const float *local_center3,
                               float local_radius, float *out4);

/* ------------------------------------------------------------------------
 * Payload builder (application_build_world_payload, application.h:671-961)
 * over candidate source rows. One source row = one (instance, submesh) pair in
 * the reference's enumeration order.
 *   node[i]      index of the transform record whose world matrix is the model
 *   ident[i*8..] mesh.id, mesh.generation, geometry.id, geometry.generation,
 *                submesh_index, material.id, material.generation, render_id
 *   matbits[i]   VKO_MAT_* word
 *   bounds[i*12] center(xyz,pad) min(xyz,pad) max(xyz,pad)
 * Outputs (caller-allocated, any may be NULL to skip):
 *   gpu_candidates[n], transmission_candidates[<=n], transparent_visible[n],
 *   draws[<=n], instances[<=n], counts[4] = {gpu_candidate_count,
 *   camera_opaque_count, transmission_count, transparent_draw_count}.
 * `capacity` = 0 disables the reference's 262144-row cap; otherwise on
 * overflow counts[0] = capacity+1 and nothing else is written
 * (application.h:703-711). Returns transparent_draw_count.
 * ---------------------------------------------------------------------- */
uint32_t vko_build_world_payload(const float *world16, const uint32_t *node,
                                 const uint32_t *ident, const uint32_t *matbits,
                                 const float *bounds, uint64_t n, const float *view,
                                 const float *projection, uint64_t capacity,
                                 uint64_t source_index_base,
                                 vko_world_candidate *gpu_candidates,
                                 vko_world_candidate *transmission_candidates,
                                 uint8_t *transparent_visible, vko_draw_item *draws,
                                 vko_instance_data *instances, uint64_t *sort_keys,
                                 uint32_t *sorted_source, vko_visibility_stats *stats,
                                 uint32_t *counts);

/* ------------------------------------------------------------------------
 * Multi-view cull + compaction semantics (deferred.slang:225-238,305-420; the HZB
 * branch only while vko_set_hzb holds a pyramid; host side
 * vkr_vulkan_deferred.c:242-295).
 *   planes: V x 24 floats; required_flags[v]
 *   classification[v*n + i] = bucket or VKO_CULLED
 *   states[v]; visible[v*visible_stride + k] = candidate index, bucket-major,
 *   ascending candidate index inside a bucket (one of the orders the
 *   reference's atomics may produce).
 *   partition_capacity = 0 means unlimited.
 *   tie_eps > 0: near_plane_ties[v] counts candidates that passed the flag gate
 *   and have |dist + radius| <= tie_eps * max(1, |d|) on some plane.
 * ---------------------------------------------------------------------- */
void vko_multiview_cull(const vko_world_candidate *candidates, uint64_t n,
                        const float *planes, const uint32_t *required_flags, uint32_t views,
                        uint32_t partition_capacity, uint64_t visible_stride,
                        uint64_t index_base, float tie_eps, uint32_t *classification,
                        vko_compaction_state *states, uint32_t *visible,
                        uint64_t *near_plane_ties);

/* HZB occlusion of view 0: vkr_vk_candidate_occluded, deferred.slang:240-303. The
 * shader leaves the order of its float operations to the compiler; this
 * restatement fixes one (every * and + rounded on its own, the matrix products in
 * the order of mat4_mul_vec4, IEEE division, floor(log2) of the integer span as a
 * bit scan). Nothing in the reference pins the bits of this step.
 *   mips: mip 0 first, each row-major with extent max(extent >> mip, 1);
 *   view_projection: column-major, the camera's (view_projections[0]).
 * vko_set_hzb(NULL) switches the test off; while set, vko_multiview_cull* apply
 * it to view 0 and count states[0].occlusion_culled_count. */
typedef struct vko_hzb {
  const float *mips;
  uint32_t width, height, mip_count;
  float depth_epsilon;
  float view_projection[16];
} vko_hzb;
/* vk_hzb_build, deferred.slang:970-998: all levels into `mips` (level 0 first). */
void vko_build_hzb(const float *depth, uint32_t source_width, uint32_t source_height,
                   uint32_t width, uint32_t height, uint32_t mip_count, float *mips);
void vko_set_hzb(const vko_hzb *hzb);
int vko_hzb_occluded(const vko_hzb *hzb, const float *world_center3, float world_radius);

/* Same, from flat arrays (no 144-byte rows): sphere[i*4] = local bounding
 * sphere, flags[i], bucket[i], model taken from world16[node[i]]. */
void vko_multiview_cull_flat(const float *world16, const uint32_t *node,
                             const float *sphere4, const uint32_t *flags,
                             const uint32_t *bucket, uint64_t n, const float *planes,
                             const uint32_t *required_flags, uint32_t views,
                             uint32_t partition_capacity, uint64_t visible_stride,
                             uint64_t index_base, float tie_eps, uint32_t *classification,
                             vko_compaction_state *states, uint32_t *visible,
                             uint64_t *near_plane_ties);

/* Encode (deferred.slang:379-420) in the canonical order. */
/* Backend publication tables flattened to POD (VkrVulkanPublishedGeometry /
 * VkrVulkanSubmeshRange / VkrVulkanPublishedMaterial,
 * lib/src/renderer/vulkan/vkr_vulkan_internal.h:838-870). */
typedef struct vko_geometry_entry {
  uint32_t generation;     /* handle.generation of the published record */
  uint32_t resolvable;     /* live && pending_initialization_count == 0 */
  uint32_t first_index;    /* gpu_row.first_index */
  uint32_t submesh_count;
  uint32_t submesh_offset; /* first entry in the submesh array */
  uint32_t pad[3];
} vko_geometry_entry;
typedef struct vko_submesh_range {
  uint32_t first_index;
  uint32_t index_count;
  int32_t vertex_offset;
  uint32_t pad;
} vko_submesh_range;
typedef struct vko_material_entry {
  uint32_t generation;
  uint32_t live;
  uint32_t slot_index;
  uint32_t pad;
} vko_material_entry;
/* VkrVkIndirectCommand, deferred.slang:21-28 (20 B) */
typedef struct vko_indirect_command {
  uint32_t index_count;
  uint32_t instance_count;
  uint32_t first_index;
  int32_t vertex_offset;
  uint32_t first_instance;
} vko_indirect_command;

/* vkr_vk_pack_gpu_candidates, lib/src/renderer/vulkan/vkr_vulkan_draws.c:344-428.
 * counts = {packed, unpublished geometry, unpublished material, invalid submesh};
 * packed_source[k] = candidate index of packed row k. Returns 0 when the
 * reference would reject the frame (a submesh outside its geometry). */
int vko_pack_gpu_candidates(const vko_world_candidate *source, uint32_t count,
                            const vko_geometry_entry *geometries, uint32_t geometry_capacity,
                            const vko_submesh_range *submeshes,
                            const vko_material_entry *materials, uint32_t material_capacity,
                            vko_gpu_candidate_row *rows, vko_instance_data *instances,
                            uint32_t *packed_source, uint32_t *counts4);
/* the indirect commands of vk_gpu_draw_encode, deferred.slang:411-419 */
void vko_encode_commands(const vko_gpu_candidate_row *rows, const uint32_t *visible,
                         const vko_compaction_state *state, uint32_t view,
                         uint32_t partition_capacity, vko_indirect_command *commands);
void vko_encode_visible_rows(const vko_gpu_candidate_row *rows, const uint32_t *visible,
                             const vko_compaction_state *state, vko_gpu_visible_row *out);

/* scene_entity_is_visible (vkr_scene_system.c:1935-1977): the walk up the parent
 * chain from `node`, for a hierarchy in any numbering. spec[i]: bit 0 = has a
 * SceneVisibility, bit 1 = visible, bit 2 = inherit_parent. */
int vko_entity_is_visible(const uint32_t *parent, const uint8_t *spec, uint64_t n, uint64_t node);

/* Backend pack epoch hash (vkr_vulkan_draws.c:9-44). */
uint64_t vko_candidate_epoch(const vko_world_candidate *candidates, uint32_t count);

/* G-way merge of descending-sorted key runs (multi-GPU draw-key merge). */
void vko_merge_runs_desc(const uint64_t *const *runs, const uint64_t *lengths, uint32_t g,
                         uint64_t *out);

/* ------------------------------------------------------------------------
 * Whole-frame CPU baseline in the reference's loop structure: AoS 208-byte
 * transforms, topo-ordered propagation, three-traversal payload build with
 * qsort, multi-view classify/prefix/encode over 144-byte rows. Single thread.
 * Returns a checksum so the work cannot be optimised away.
 * ---------------------------------------------------------------------- */
typedef struct vko_baseline_scene vko_baseline_scene;
vko_baseline_scene *vko_baseline_create(uint64_t n_nodes, const uint32_t *parent,
                                        const float *local16, uint64_t n_cand,
                                        const uint32_t *node, const uint32_t *ident,
                                        const uint32_t *matbits, const float *bounds,
                                        uint32_t views, const float *planes,
                                        const uint32_t *required_flags, const float *view,
                                        const float *projection);
uint64_t vko_baseline_frame(vko_baseline_scene *s);
void vko_baseline_destroy(vko_baseline_scene *s);

/* ------------------------------------------------------------------------
 * Cascade fitting (vkr_shadow_system.c), vkr_oracle_shadow.c. The step in front
 * of the cull for the shadow views: camera + light direction + configuration
 * -> one light view-projection matrix per cascade.
 * ---------------------------------------------------------------------- */
#define VKO_SHADOW_CASCADES_MAX 8 /* VKR_SHADOW_CASCADE_COUNT_MAX */
enum { VKO_CAMERA_PERSPECTIVE = 0, VKO_CAMERA_ORTHOGRAPHIC = 1 };

typedef struct vko_shadow_fit { /* VkrShadowFit, vkr_shadow_system.h:32-39 */
  float center_x, center_y, extent, min_z, max_z, world_units_per_texel;
} vko_shadow_fit;

typedef struct vko_shadow_config { /* the fields of VkrShadowConfig (:151-177) the fit reads */
  uint32_t cascade_count;
  uint32_t shadow_map_size;
  float cascade_split_lambda;
  float max_shadow_distance;
  float cascade_guard_band_texels;
  float cascade_guard_band_texels_per[VKO_SHADOW_CASCADES_MAX];
  float z_extension_factor;
  float cascade_z_extension_factor_per[VKO_SHADOW_CASCADES_MAX];
  uint32_t use_constant_cascade_size;
  float anchor_snap_texels;
  uint32_t stabilize_cascades;
  uint32_t use_scene_bounds; /* VkrShadowSceneBounds */
  float scene_min[3];
  float scene_max[3];
} vko_shadow_config;

typedef struct vko_shadow_camera { /* what the fit reads of VkrCamera */
  uint32_t type;
  float position[3], forward[3], right[3], up[3];
  float near_clip, far_clip, zoom;
  uint32_t window_width, window_height; /* cached_window_width / _height */
  float left_clip, right_clip, bottom_clip, top_clip;
} vko_shadow_camera;

typedef struct vko_shadow_cascade { /* VkrCascadeData, :74-83 */
  float view_projection[16];
  float split_far;
  float world_units_per_texel;
  float light_space_origin[2];
  float light_space_depth_span;
  float bounds_center[3];
  float bounds_radius;
} vko_shadow_cascade;

typedef struct vko_shadow_system { /* VkrShadowSystem + VkrShadowFitHistory */
  vko_shadow_config config;
  vko_shadow_cascade cascades[VKO_SHADOW_CASCADES_MAX];
  float cascade_splits[VKO_SHADOW_CASCADES_MAX + 1];
  float light_direction[3];
  uint32_t light_enabled;
  vko_shadow_fit history_fits[VKO_SHADOW_CASCADES_MAX];
  float history_light_direction[3];
  uint32_t history_cascade_count;
  uint32_t history_shadow_map_size;
  uint32_t history_projection_convention;
  uint32_t history_valid;
  uint64_t history_enable_generation;
  uint64_t enable_generation;
  uint32_t initialized;
  uint32_t reserved;
} vko_shadow_system;

void vko_mat4_look_at(const float *eye, const float *center, const float *up, float *out);
void vko_mat4_inverse_rigid(const float *m, float *out);
void vko_shadow_cascade_splits(const vko_shadow_config *cfg, float near_clip, float far_clip,
                               float lambda, float *splits);
void vko_shadow_frustum_corners(const vko_shadow_camera *cam, float near_split, float far_split,
                                float *corners24);
void vko_shadow_light_view(const vko_shadow_camera *cam, const vko_shadow_config *cfg,
                           const float *light_direction, uint32_t shadow_map_size, float *out);
int vko_shadow_fit_relevant_caster_z(const float *light_view, const float *bounds_min,
                                     const float *bounds_max, float left, float right, float bottom,
                                     float top, float *out_min_z, float *out_max_z);
float vko_shadow_quantize_extent_up(float extent, uint32_t shadow_map_size);
void vko_shadow_apply_fit_hysteresis(const vko_shadow_fit *previous, const vko_shadow_fit *raw,
                                     vko_shadow_fit *out);
void vko_shadow_cascade_matrix(const float *light_view, const float *corners24,
                               uint32_t shadow_map_size, int stabilize, float guard_band_texels,
                               int use_constant_cascade_size, const vko_shadow_config *cfg,
                               float z_extension_factor, const vko_shadow_fit *previous_fit,
                               float *out_view_projection, vko_shadow_fit *out_fit,
                               float *out_light_space_origin);
int vko_shadow_system_init(vko_shadow_system *s, const vko_shadow_config *config);
void vko_shadow_system_invalidate_fit_history(vko_shadow_system *s);
void vko_shadow_system_update(vko_shadow_system *s, const vko_shadow_camera *cam, int light_enabled,
                              const float *light_direction);

#ifdef __cplusplus
}
#endif
#endif
