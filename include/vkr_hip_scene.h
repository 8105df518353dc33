/*
 * vkr_hip_scene.h - C ABI of the MI355X (gfx950) scene-processing library.
 *
 * One context owns the HBM-resident scene (hierarchy, local matrices, candidate
 * rows) of one GPU and runs the per-frame path
 *     transform propagation -> multi-view sphere frustum cull + ordered
 *     compaction -> transparent key generation, radix sort, draw emission
 * entirely on the device. Plain pointers and sizes only; every entry point
 * returns VKRHIP_OK (0) or a negative error and never aborts.
 *
 * All matrices are column-major float[16] (element = m[col*4 + row]), the
 * reference's Mat4 (lib/src/math/mat.h:214-222). "Vec3" arrays use the
 * reference's padded 16-byte stride (lib/src/math/vec.h:210-216).
 *
 * Threading: one context per host thread, not re-entrant (the reference runs
 * this path on its render thread only, lib/src/application.h:283).
 * Ownership: output buffers belong to the context and stay valid until the
 * next vkrhip_frame() on it, matching the frame-scoped borrow of
 * VkrWorldPassPayload arrays (lib/src/renderer/vkr_render_packet.h:121-125).
 *
 * File:line citations are relative to the reference checkout.
 */
#ifndef VKR_HIP_SCENE_H
#define VKR_HIP_SCENE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VKRHIP_OK 0
#define VKRHIP_ERR_ARG (-1)      /* null/invalid argument or out-of-range size */
#define VKRHIP_ERR_HIP (-2)      /* a HIP runtime call failed (see vkrhip_last_error) */
#define VKRHIP_ERR_CAPACITY (-3) /* exceeds a capacity fixed at vkrhip_create */
#define VKRHIP_ERR_STATE (-4)    /* call order violated (e.g. frame before set_views) */

#define VKRHIP_MAX_VIEWS 64u
#define VKRHIP_NO_PARENT 0xffffffffu
#define VKRHIP_PLANE_COUNT 6u  /* LEFT,RIGHT,BOTTOM,TOP,NEAR,FAR: lib/src/math/vkr_frustum.h */
#define VKRHIP_BUCKET_COUNT 4u /* lib/src/renderer/vkr_gpu_abi.h:14-20 */

/* scene transform flags, lib/src/renderer/systems/vkr_scene_system.h:60-65 */
#define VKRHIP_DIRTY_LOCAL 0x01u
#define VKRHIP_DIRTY_WORLD 0x02u
#define VKRHIP_WORLD_UPDATED 0x08u

/* candidate flags, lib/src/renderer/vkr_render_packet.h:113-118 */
#define VKRHIP_CAND_BOUNDS_VALID 0x1u
#define VKRHIP_CAND_CAMERA_OPAQUE 0x2u
#define VKRHIP_CAND_SHADOW_CASTER 0x4u

/* per-candidate material word (input of vkrhip_set_candidates): bits 0-1 alpha
 * mode (0 opaque, 1 cutout, 2 blend: lib/src/renderer/resources/vkr_resources.h:129-133),
 * bit 2 transmissive, bit 3 double sided, bit 4 bounds valid. */
#define VKRHIP_MAT_ALPHA_MASK 0x3u
#define VKRHIP_MAT_TRANSMISSIVE 0x4u
#define VKRHIP_MAT_DOUBLE_SIDED 0x8u
#define VKRHIP_MAT_BOUNDS_VALID 0x10u

/* vkrhip_frame stage mask */
#define VKRHIP_STAGE_TRANSFORM 0x1u   /* vkr_scene_update, vkr_scene_system.c:1254-1337 */
#define VKRHIP_STAGE_CULL 0x2u        /* classify/prefix/encode, deferred.slang:305-420 + application.h:730-806 */
#define VKRHIP_STAGE_TRANSPARENT 0x4u /* key, sort, emit: application.h:573-587,940-944 */
#define VKRHIP_STAGE_CANDIDATES 0x8u  /* 144-byte VkrWorldDrawCandidate rows, application.h:617-662 */
#define VKRHIP_STAGE_WORLD_BOUNDS 0x10u /* instance world spheres, vkr_mesh_manager.c:315-337 */
#define VKRHIP_STAGE_PACK 0x20u   /* vkr_vk_pack_gpu_candidates, vkr_vulkan_draws.c:344-428 */
#define VKRHIP_STAGE_ENCODE 0x40u /* visible rows + indirect commands, deferred.slang:379-420 */
#define VKRHIP_STAGE_CASTER_BOUNDS 0x80u /* box of the shadow casters, for VkrShadowSceneBounds */
#define VKRHIP_STAGE_DIRTY_LIST 0x100u /* scene->render_dirty_list, vkr_scene_system.c:458-479,1335 */
#define VKRHIP_STAGE_ALL 0x7u

typedef struct vkrhip_ctx vkrhip_ctx;

typedef struct vkrhip_config {
  uint32_t struct_size;         /* sizeof(vkrhip_config) */
  int32_t device;               /* HIP device ordinal */
  uint64_t max_nodes;           /* scene transform capacity */
  uint64_t max_candidates;      /* instance x submesh source-row capacity */
  uint32_t max_views;           /* 1..VKRHIP_MAX_VIEWS */
  uint32_t partition_capacity;  /* per-view per-bucket cap as command_partition_capacity
                                   (vkr_vulkan_deferred.c:6-7,289-290); 0 = unlimited */
  uint64_t visible_capacity;    /* per-view visible list capacity; 0 = max_candidates */
  uint64_t transparent_capacity;/* sorted blend draw capacity; 0 = max_candidates */
  uint64_t source_index_base;   /* global index of candidate 0 of this shard */
  float tie_epsilon;            /* >0: count spheres within eps*max(1,|d|) of a plane */
  uint32_t flags;               /* reserved, 0 */
} vkrhip_config;

/* VkrVisibilityStats, lib/src/renderer/vkr_visibility.h:42-46 */
typedef struct vkrhip_visibility_stats {
  uint32_t objects_tested;
  uint32_t objects_culled_camera;
  uint32_t objects_without_bounds;
} vkrhip_visibility_stats;

/* VkrGpuDrawCompactionState, lib/src/renderer/vkr_gpu_abi.h:95-103 (80 bytes) */
typedef struct vkrhip_compaction_state {
  uint32_t execution_ranges[4][2]; /* {offset, count} per bucket */
  uint32_t bucket_counts[4];
  uint32_t bucket_cursors[4];
  uint32_t visible_count;
  uint32_t overflow_count;
  uint32_t resolve_invalid_count;
  uint32_t occlusion_culled_count;
} vkrhip_compaction_state;

/* Per-frame scalar results (host copy, valid after vkrhip_get_frame_info). */
typedef struct vkrhip_frame_info {
  vkrhip_visibility_stats stats;
  uint32_t gpu_candidate_count;               /* VkrWorldPassPayload.gpu_candidate_count */
  uint32_t gpu_camera_opaque_candidate_count; /* application.h:747-748 */
  uint32_t transmission_gpu_candidate_count;  /* application.h:749 */
  uint32_t transparent_draw_count;            /* application.h:761, clamped to capacity */
  uint32_t transparent_overflow;              /* visible blend rows beyond transparent_capacity */
  uint32_t nodes_updated;                     /* length of the render-dirty list, vkr_scene_system.c:1335 */
  uint32_t view_count;
  uint32_t reserved;
  /* vkr_vk_pack_gpu_candidates counters (zero without a publication): rows packed,
   * omitted for an unpublished geometry / material, rejected for a submesh
   * outside its geometry (the reference fails the frame when that is non-zero) */
  uint32_t packed_count;
  uint32_t unpublished_geometry_count;
  uint32_t unpublished_material_count;
  uint32_t invalid_submesh_count;
  uint64_t near_plane_ties[VKRHIP_MAX_VIEWS];
  vkrhip_compaction_state views[VKRHIP_MAX_VIEWS];
} vkrhip_frame_info;

/* Device-resident outputs of the last frame (HBM pointers on config.device). */
typedef struct vkrhip_device_outputs {
  const float *world;               /* max_nodes x 16: SceneTransform.world */
  const uint8_t *node_flags;        /* max_nodes: SceneTransform.flags after the update */
  const uint32_t *visible;          /* view v at visible + v*visible_stride: candidate
                                       indices (+source_index_base), bucket-major, ascending
                                       inside a bucket; ranges in vkrhip_frame_info.views[v] */
  uint64_t visible_stride;
  const uint8_t *transparent_visible; /* n: application.h:714-723,760 */
  const void *transparent_draws;      /* T x VkrDrawItem (48 B), vkr_render_packet.h:100-111 */
  const void *transparent_instances;  /* T x VkrInstanceDataGPU (80 B), vkr_gpu_abi.h:34-38 */
  const uint64_t *transparent_keys;   /* T sort keys, descending (application.h:582-587) */
  const uint32_t *transparent_source; /* T local candidate indices in draw order */
  const void *gpu_candidates;         /* n x VkrWorldDrawCandidate (144 B) when STAGE_CANDIDATES ran */
  const float *world_bounds;          /* n x (cx,cy,cz,r) when STAGE_WORLD_BOUNDS ran */
  const uint32_t *device_counts;      /* device copy of the scalar counters */
  const void *packed_rows;            /* packed_count x VkrGpuCandidateDrawRow (48 B) when STAGE_PACK ran */
  const void *packed_instances;       /* packed_count x VkrInstanceDataGPU (80 B) */
  const void *visible_rows;           /* view v at v*visible_stride: VkrGpuVisibleDrawRow (32 B)
                                         per visible slot when STAGE_ENCODE ran */
  const void *commands;               /* (view*4+bucket)*partition_capacity + local: 20-byte
                                         indirect commands; null when partition_capacity is 0 */
  const void *transmission_gpu_candidates; /* transmission_gpu_candidate_count rows (144 B): the
                                         rows of the transmissive sources once more, in source
                                         order (application.h:643-645); null when there are none */
  const uint32_t *dirty_nodes;        /* when STAGE_DIRTY_LIST ran: the nodes the update rewrote, in
                                         topological order (ascending index), the list
                                         scene_mark_render_dirty builds (vkr_scene_system.c:458-479) */
  const uint32_t *dirty_node_count;   /* ... and its length (one device word) */
} vkrhip_device_outputs;

/* Backend publication state, flattened (VkrVulkanPublishedGeometry,
 * VkrVulkanSubmeshRange, VkrVulkanPublishedMaterial:
 * lib/src/renderer/vulkan/vkr_vulkan_internal.h:838-870). Entry k describes
 * handle id k + 1. */
typedef struct vkrhip_geometry_entry {
  uint32_t generation;     /* handle.generation of the published record */
  uint32_t resolvable;     /* live && pending_initialization_count == 0 */
  uint32_t first_index;    /* gpu_row.first_index */
  uint32_t submesh_count;
  uint32_t submesh_offset; /* first entry of this geometry in the submesh array */
  uint32_t reserved[3];
} vkrhip_geometry_entry;
typedef struct vkrhip_submesh_range {
  uint32_t first_index;
  uint32_t index_count;
  int32_t vertex_offset;
  uint32_t reserved;
} vkrhip_submesh_range;
typedef struct vkrhip_material_entry {
  uint32_t generation;
  uint32_t live;
  uint32_t slot_index;
  uint32_t reserved;
} vkrhip_material_entry;

/* ---- lifetime --------------------------------------------------------- */
int vkrhip_create(const vkrhip_config *config, vkrhip_ctx **out_ctx);
void vkrhip_destroy(vkrhip_ctx *ctx);
/* Message of the last failure on this context (or of vkrhip_create when ctx is NULL). */
const char *vkrhip_last_error(const vkrhip_ctx *ctx);
/* Run all work of this context on an existing hipStream_t (NULL restores the
 * context's own stream). */
int vkrhip_set_stream(vkrhip_ctx *ctx, void *hip_stream);
/* The hipStream_t the context currently enqueues on (its own stream unless
 * vkrhip_set_stream changed it), for callers that want to order their own work
 * (copies, collectives) with the context's. */
void *vkrhip_get_stream(const vkrhip_ctx *ctx);
/* On its own stream a context with a large hierarchy sorts and emits a frame's
 * transparent rows on a second stream, beside the next frame's transform. Every
 * entry point that reads, copies or hands out results orders the context's stream
 * behind that work by itself; a caller that enqueues kernels of its own on
 * vkrhip_get_stream() to read the transparent arrays calls vkrhip_join after
 * vkrhip_frame first (one stream-wait, the host does not block). No counterpart in
 * the reference, whose frame is one host thread. */
int vkrhip_join(vkrhip_ctx *ctx);
/* Waits for everything the context has enqueued (vkr_renderer_submit_packet
 * returns only after the payload was consumed, lib/src/renderer/vkr_render_packet.h:121-125). */
int vkrhip_sync(vkrhip_ctx *ctx);

/* ---- scene state: replaces the SceneTransform column walked by
 * vkr_scene_update (lib/src/renderer/systems/vkr_scene_system.c:1254-1337) ---
 * Nodes are stored in the reference's BFS topo order (vkr_scene_system.c:783-833):
 * level l occupies [level_offsets[l], level_offsets[l+1]) and a node's parent
 * lies in the level above. vkrhip_levelize() produces this order. */
int vkrhip_set_hierarchy(vkrhip_ctx *ctx, const uint32_t *parent, uint64_t n_nodes,
                         const uint64_t *level_offsets, uint32_t levels);
/* position/rotation(x,y,z,w)/scale, 4 floats each; marks the nodes DIRTY_LOCAL
 * like the transform setters do. */
int vkrhip_upload_trs(vkrhip_ctx *ctx, uint64_t first, uint64_t count, const float *pos4,
                      const float *quat4, const float *scale4);
/* Cached local matrices (SceneTransform.local); marks the nodes DIRTY_WORLD. */
int vkrhip_upload_locals(vkrhip_ctx *ctx, uint64_t first, uint64_t count, const float *local16);
/* World matrices computed elsewhere (SceneTransform.world, vkr_scene_system.h:60-65), for
 * frames that run the cull stages without the transform stage; marks nothing. */
int vkrhip_upload_worlds(vkrhip_ctx *ctx, uint64_t first, uint64_t count, const float *world16);
int vkrhip_upload_node_flags(vkrhip_ctx *ctx, uint64_t first, uint64_t count,
                             const uint8_t *flags);
/* OR `flags` into every node (e.g. VKRHIP_DIRTY_WORLD: everything moved). */
int vkrhip_mark_all(vkrhip_ctx *ctx, uint8_t flags);
/* OR `flags` into nodes [first, first + count), on the device and in stream order
 * (no host wait): what the reference's transform setters do for one transform
 * (vkr_scene_system.h:70-72), for a run of nodes - e.g. the roots of the subtrees
 * that moved this frame; their descendants follow by the deferred propagation of
 * vkr_scene_update (vkr_scene_system.c:1310-1316). */
int vkrhip_mark_nodes(vkrhip_ctx *ctx, uint64_t first, uint64_t count, uint8_t flags);

/* ---- candidate source rows: the (instance, submesh) enumeration of
 * application_build_world_payload (lib/src/application.h:730-806) -----------
 *   node[i]        transform whose world matrix is the row's model; NULL = i
 *   ident[i*8..]   mesh.id, mesh.generation, geometry.id, geometry.generation,
 *                  submesh_index, material.id, material.generation, render_id
 *   matbits[i]     VKRHIP_MAT_* word
 *   bounds[i*12..] center, min_extents, max_extents (3 padded Vec3,
 *                  VkrMeshAssetSubmesh @80..127, vkr_resources.h:292-294) */
int vkrhip_set_candidates(vkrhip_ctx *ctx, uint64_t n, const uint32_t *node,
                          const uint32_t *ident, const uint32_t *matbits, const float *bounds);

/* ---- views: camera + cascades as the backend builds them
 * (lib/src/renderer/vulkan/vkr_vulkan_deferred.c:242-295) --------------------
 *   planes          V x 6 x (nx,ny,nz,d), from vkrhip_frustum_from_* or the
 *                   reference's vkr_frustum_from_* (lib/src/math/vkr_frustum.c:33-66)
 *   required_flags  per view; view 0 = CAMERA_OPAQUE, others SHADOW_CASTER in the reference
 *   camera_view     the camera view matrix (globals.view) for transparent depth */
int vkrhip_set_views(vkrhip_ctx *ctx, uint32_t views, const float *planes,
                     const uint32_t *required_flags, const float *camera_view);

/* ---- backend publication: the tables vkr_vk_pack_gpu_candidates resolves
 * handles against (lib/src/renderer/vulkan/vkr_vulkan_draws.c:321-342). Call
 * after vkrhip_set_candidates and again whenever the tables or the candidates
 * change. From then on rows that do not resolve are in no GPU view and the
 * visible lists hold PACKED row indices (the cull of the reference runs on the
 * packed rows); source_index_base does not apply to them. */
int vkrhip_set_publication(vkrhip_ctx *ctx, const vkrhip_geometry_entry *geometries,
                           uint32_t geometry_capacity, const vkrhip_submesh_range *submeshes,
                           uint32_t submesh_count, const vkrhip_material_entry *materials,
                           uint32_t material_capacity);

/* ---- the frame ------------------------------------------------------------ */
/* Enqueues the selected stages on the context's stream and returns without
 * waiting. */
int vkrhip_frame(vkrhip_ctx *ctx, uint32_t stages);
/* Waits for the last frame and copies its scalar results to the host. */
int vkrhip_get_frame_info(vkrhip_ctx *ctx, vkrhip_frame_info *out);
int vkrhip_get_device_outputs(vkrhip_ctx *ctx, vkrhip_device_outputs *out);

/* ---- the payload of the reference's world pass -------------------------------
 * VkrWorldPassPayload (lib/src/renderer/vkr_render_packet.h:162-179, 88 bytes)
 * and VkrVisibilityStats (lib/src/renderer/vkr_visibility.h:42-46) as
 * application_build_world_payload leaves them (lib/src/application.h:946-960):
 * vkrhip_fill_world_payload completes the last frame - which must have run
 * VKRHIP_STAGE_CULL | VKRHIP_STAGE_TRANSPARENT | VKRHIP_STAGE_CANDIDATES - and
 * fills both. The arrays the payload points to are host memory (pinned) owned by
 * the context, valid until the next vkrhip_frame, like the reference's
 * frame-scoped scratch allocations (vkr_render_packet.h:121-125). The frame's
 * counts reach the host through a pinned block the device writes, not through
 * copies of their own.
 * More source rows than VKRHIP_DRAW_CANDIDATE_CAPACITY is not a failure, as in
 * the reference (application.h:703-711): the payload is empty except for
 * gpu_candidate_count = capacity + 1, objects_tested likewise, and packet
 * validation rejects the frame. stats may be NULL. */
#define VKRHIP_DRAW_CANDIDATE_CAPACITY 262144u /* VKR_GPU_DRAW_CANDIDATE_CAPACITY, vkr_gpu_abi.h:9 */
typedef struct vkrhip_world_payload {
  const void *gpu_candidates; /* gpu_candidate_count x VkrWorldDrawCandidate (144 B) */
  uint32_t gpu_candidate_count;
  uint32_t gpu_camera_opaque_candidate_count;
  uint32_t gpu_shadow_candidate_count;
  const void *transmission_gpu_candidates;
  uint32_t transmission_gpu_candidate_count;
  const void *transparent_draws; /* transparent_draw_count x VkrDrawItem (48 B) */
  uint32_t transparent_draw_count;
  const void *instances; /* instance_count x VkrInstanceDataGPU (80 B) */
  uint32_t instance_count;
  const void *text_draws; /* not produced by this path: NULL, 0 */
  uint32_t text_draw_count;
} vkrhip_world_payload;
int vkrhip_fill_world_payload(vkrhip_ctx *ctx, vkrhip_world_payload *out_payload,
                              vkrhip_visibility_stats *out_stats);

/* Page-locked host memory for the destinations of the vkrhip_read_* copies below
 * (the arrays of the VkrWorldPassPayload a host-side consumer reads, which the
 * reference takes from its frame scratch allocator, application.h:1415-1419): a
 * copy into it runs at the PCIe rate, about four times that of pageable memory
 * (2.4 GB of candidate rows: 44 ms against 174 ms). The memory belongs to the
 * caller until vkrhip_host_free; any host pointer still works as a destination. */
int vkrhip_host_alloc(vkrhip_ctx *ctx, uint64_t bytes, void **out_pointer);
int vkrhip_host_free(vkrhip_ctx *ctx, void *pointer);

/* Synchronous device-to-host copies of outputs (after the frame completes). */
int vkrhip_read_world(vkrhip_ctx *ctx, uint64_t first, uint64_t count, float *out16);
int vkrhip_read_locals(vkrhip_ctx *ctx, uint64_t first, uint64_t count, float *out16);
int vkrhip_read_node_flags(vkrhip_ctx *ctx, uint64_t first, uint64_t count, uint8_t *out);
int vkrhip_read_visible(vkrhip_ctx *ctx, uint32_t view, uint64_t count, uint32_t *out);
int vkrhip_read_transparent_visible(vkrhip_ctx *ctx, uint64_t first, uint64_t count,
                                    uint8_t *out);
int vkrhip_read_transparent(vkrhip_ctx *ctx, uint64_t count, void *draws48, void *instances80,
                            uint64_t *keys, uint32_t *source);
int vkrhip_read_candidates(vkrhip_ctx *ctx, uint64_t first, uint64_t count, void *rows144);
/* rows of VkrWorldPassPayload.transmission_gpu_candidates (vkr_render_packet.h:162-179) */
int vkrhip_read_transmission_candidates(vkrhip_ctx *ctx, uint64_t first, uint64_t count,
                                        void *rows144);
int vkrhip_read_packed(vkrhip_ctx *ctx, uint64_t count, void *rows48, void *instances80);
int vkrhip_read_visible_rows(vkrhip_ctx *ctx, uint32_t view, uint64_t count, void *rows32);
/* `count` commands of (view, bucket), starting at local index 0 */
int vkrhip_read_commands(vkrhip_ctx *ctx, uint32_t view, uint32_t bucket, uint64_t count,
                         void *commands20);
int vkrhip_read_world_bounds(vkrhip_ctx *ctx, uint64_t first, uint64_t count, float *out4);
/* The render-dirty list of the last frame (STAGE_DIRTY_LIST): every node whose world
 * matrix vkr_scene_update rewrote, in the order the reference appends them
 * (scene_mark_render_dirty, vkr_scene_system.c:458-479, called at :1335 while the
 * topological order is walked; the render bridge consumes the list at :2123-2124).
 * *out_count is the length of the list (= vkrhip_frame_info::nodes_updated of a frame
 * with STAGE_TRANSFORM); at most `capacity` entries are copied. */
int vkrhip_read_dirty_list(vkrhip_ctx *ctx, uint64_t capacity, uint32_t *out_nodes, uint64_t *out_count);
int vkrhip_read_candidate_prep(vkrhip_ctx *ctx, uint64_t first, uint64_t count,
                               float *sphere4, uint32_t *flags_bucket);

/* ---- per-stage device timing (HIP events on the context's stream) ---------- */
#define VKRHIP_TIMER_TRANSFORM 0 /* stand-alone level launches */
#define VKRHIP_TIMER_CLASSIFY 1  /* stand-alone classify */
#define VKRHIP_TIMER_FUSED 2     /* transform + classify in one launch per level */
#define VKRHIP_TIMER_SCAN 3
#define VKRHIP_TIMER_SCATTER 4
#define VKRHIP_TIMER_SORT 5
#define VKRHIP_TIMER_EMIT 6
#define VKRHIP_TIMER_CANDIDATES 7
#define VKRHIP_TIMER_FRAME 8
#define VKRHIP_TIMER_EXCHANGE_PACK 9    /* vkrhip_exchange_run: the pack kernel */
#define VKRHIP_TIMER_EXCHANGE_GATHER 10 /* ... the all-gather (with a communicator) */
#define VKRHIP_TIMER_EXCHANGE_MERGE 11  /* ... plan + merge kernels */
#define VKRHIP_TIMER_COUNT 12
/* timer_mask: bit VKRHIP_TIMER_x set = bracket that stage with HIP events; 0 = off. */
int vkrhip_set_profiling(vkrhip_ctx *ctx, uint32_t timer_mask);
/* Accumulated milliseconds and sample counts since the last reset; waits for
 * outstanding work. */
int vkrhip_get_timers(vkrhip_ctx *ctx, double *ms_out, uint64_t *samples_out, int reset);

/* ---- multi-GPU draw-key merge (no reference counterpart: the reference is
 * single-GPU; this is the exchange step after an all-gather of per-GPU runs) --
 * runs_keys: `g` device arrays of sort keys, each sorted descending; lengths on
 * the host. Writes the merged descending keys (the low key word names the
 * source row, so no payload travels). */
int vkrhip_merge_runs(vkrhip_ctx *ctx, uint32_t g, const uint64_t *const *runs_keys,
                      const uint64_t *lengths, uint64_t *out_keys_dev);

/* Segmented device copy of 32-bit elements: descriptor d = {src offset, dst
 * offset, count}; builds the merged per-(view, bucket) visible lists out of the
 * all-gathered per-GPU lists (rank order is index order). No reference
 * counterpart. */
int vkrhip_concat_ranges(vkrhip_ctx *ctx, const uint32_t *src_dev, uint32_t *dst_dev,
                         const uint64_t *desc, uint32_t n_desc);

/* ---- occlusion of the camera view against a depth pyramid (the HZB branch of
 * vk_gpu_draw_classify: vkr_vk_candidate_occluded,
 * lib/src/renderer/shaders/vulkan/slang/world/deferred.slang:240-303,340-347;
 * host gates lib/src/renderer/vulkan/vkr_vulkan_deferred.c:298-366) ------------
 * mips: mip_count levels, mip 0 first, each row-major with extent
 * max(extent >> mip, 1), one float (farthest depth) per texel; view_projection:
 * the camera's, column-major (root.view_projections[0]). While a pyramid is set,
 * rows of view 0 that pass the frustum and have valid bounds are tested against
 * it; occluded rows leave view 0 and are counted in
 * views[0].occlusion_culled_count. NULL mips or mip_count 0 switch the test off
 * (root.hzb_enabled == 0). Waits for the context's stream; the data is copied. */
int vkrhip_set_hzb(vkrhip_ctx *ctx, const float *mips, uint32_t width, uint32_t height,
                   uint32_t mip_count, const float *view_projection, float depth_epsilon);

/* The pyramid built on the device from a depth image (vk_hzb_build,
 * deferred.slang:970-998): level 0 is texel min(xy, source extent - 1) of the
 * source_width x source_height depth image at width x height; every further
 * level is the maximum, starting from 0, of the 2x2 texels of the level above
 * (clamped at its edge), extent max(extent >> level, 1). Then as vkrhip_set_hzb.
 * depth is host memory. vkrhip_read_hzb copies one level back. */
int vkrhip_build_hzb(vkrhip_ctx *ctx, const float *depth, uint32_t source_width,
                     uint32_t source_height, uint32_t width, uint32_t height, uint32_t mip_count,
                     const float *view_projection, float depth_epsilon);
int vkrhip_read_hzb(vkrhip_ctx *ctx, uint32_t mip, float *out);

/* ---- candidate epoch (replaces vkr_vk_candidate_epoch,
 * lib/src/renderer/vulkan/vkr_vulkan_draws.c:9-44) -----------------------------
 * 64-bit FNV-1a over mesh, geometry, submesh_index, material, instance.model,
 * local_bounding_sphere, state_bucket and flags of every CAMERA_OPAQUE source
 * candidate in row order (with the current world matrices), closed with the four
 * bytes of the occluder count: the key of the backend's occlusion history.
 * Computed on the device from the candidate inputs, no candidate rows needed;
 * waits for the result. */
int vkrhip_candidate_epoch(vkrhip_ctx *ctx, uint64_t *out_hash, uint32_t *out_occluders);

/* ---- multi-GPU exchange planned on the device (no reference counterpart) ------
 * The per-frame exchange of one rank with the others without a host round trip:
 *   vkrhip_exchange_pack   -> this rank's block in `block_dev`
 *   (all-gather of the blocks with the caller's collective library, e.g. RCCL)
 *   vkrhip_exchange_merge  -> merged visible lists, merged draw keys, result words
 * All launches return at once; pack goes to the context's stream.
 *
 * Block, in 32-bit words (vkrhip_exchange_block_words gives the total, a
 * multiple of 4): execution_ranges[view][bucket] = {offset, count} for every
 * view; {draw count, visible total, views, 0}; padding to a multiple of 4;
 * vis_cap (rounded up to 4) visible candidate indices, the views' lists back to
 * back; key_cap 64-bit draw keys in descending order. Lists longer than the
 * capacities are cut; the header keeps the true counts and the merge flags it.
 *
 * Result, in 64-bit words (vkrhip_exchange_result_words): [0] flags (0 = merged
 * outputs complete; 1 = some rank's visible lists exceeded vis_cap, 2 = some
 * rank's keys exceeded key_cap, 4 = a block was packed for another view count),
 * [1] largest visible total of a rank, [2] largest draw count of a rank, [3]
 * merged visible total, [4] merged draw count, [5..7] zero, [8..8+views] start of
 * every view in the merged list (+ the total), then views*4 {offset inside the
 * view, count} pairs. merged_visible_dev holds world*vis_cap words and
 * merged_keys_dev world*key_cap keys. */
uint64_t vkrhip_exchange_block_words(uint32_t views, uint32_t vis_cap, uint32_t key_cap);
uint64_t vkrhip_exchange_result_words(uint32_t views);
int vkrhip_exchange_pack(vkrhip_ctx *ctx, uint32_t *block_dev, uint32_t vis_cap, uint32_t key_cap);
/* hip_stream: the hipStream_t the merge is enqueued on; NULL = the context's
 * stream. A caller that runs collective and merge on a stream of their own lets
 * the next frame start right after vkrhip_exchange_pack. Merges of one context
 * must not overlap each other (they share the planning tables). */
int vkrhip_exchange_merge(vkrhip_ctx *ctx, const uint32_t *gathered_dev, uint32_t world,
                          uint32_t vis_cap, uint32_t key_cap, uint32_t *merged_visible_dev,
                          uint64_t *merged_keys_dev, uint64_t *result_dev, void *hip_stream);

/* ---- host-side mirrors of reference leaf functions (exact op order) -------- */
/* vkr_frustum_from_matrix, lib/src/math/vkr_frustum.c:44-66 */
void vkrhip_frustum_from_matrix(const float *view_projection, float *planes24);
/* vkr_frustum_from_view_projection, lib/src/math/vkr_frustum.c:33-38 */
void vkrhip_frustum_from_view_projection(const float *view, const float *projection,
                                         float *planes24);
/* mat4_mul, lib/src/math/mat.h:1145-1163 */
void vkrhip_mat4_mul(const float *a, const float *b, float *out);
/* View set of the reference backend (vkr_vulkan_deferred.c:257-273): view 0 from
 * (view, projection), views 1..cascades from light_view_proj[]. Writes
 * (1+cascades) x 24 planes and required flags. */
void vkrhip_build_view_set(const float *view, const float *projection,
                           const float *light_view_proj, uint32_t cascades, float *planes,
                           uint32_t *required_flags);
/* BFS relabelling in the shape of scene_rebuild_topo_order
 * (vkr_scene_system.c:777-843): order[k] = original index of the node stored at
 * k; new_parent in the new numbering; level_offsets[levels+1]. Returns levels
 * (0 on allocation failure). level_offsets must hold n+1 entries at most
 * (levels <= n). */
uint32_t vkrhip_levelize(const uint32_t *parent, uint64_t n, uint32_t *order,
                         uint32_t *new_parent, uint64_t *level_offsets);

/* ---- the whole exchange inside the library, over RCCL (no reference
 * counterpart). One communicator per context; RCCL is bound at run time.
 *   rank 0:  vkrhip_comm_unique_id(id)      -> send the 128 bytes to every rank
 *   all:     vkrhip_comm_init(ctx, id, world, rank)          (collective)
 *            vkrhip_exchange_configure(ctx, world, vis_cap, key_cap)
 *   per frame, after vkrhip_frame:  vkrhip_exchange_run(ctx)
 *   when the merged result is needed: vkrhip_exchange_finish(ctx, ...)
 * vkrhip_exchange_run enqueues the pack on the context's stream and the
 * all-gather and the merge on a second stream, into one of two buffer sets: the
 * next vkrhip_frame may be enqueued at once and overlaps them. Without a
 * communicator (world 1) the merge reads the packed block itself. vkrhip_exchange_finish
 * waits for the latest run and copies its result words (layout above,
 * vkrhip_exchange_result_words(views) of them); when result[0] has flag 1 or 2 a
 * rank did not fit: every rank sees the same flags and sizes, so all call
 * vkrhip_exchange_configure with larger capacities and run again. The merged
 * lists stay valid until the run after the next one. */
int vkrhip_comm_unique_id(uint8_t *out_id128);
int vkrhip_comm_init(vkrhip_ctx *ctx, const uint8_t *id128, uint32_t world, uint32_t rank);
/* Ranks the context's communicator spans, as RCCL itself reports them (ncclCommCount). */
int vkrhip_comm_count(vkrhip_ctx *ctx, uint32_t *out_ranks);
int vkrhip_exchange_configure(vkrhip_ctx *ctx, uint32_t world, uint32_t vis_cap, uint32_t key_cap);
int vkrhip_exchange_run(vkrhip_ctx *ctx);
int vkrhip_exchange_finish(vkrhip_ctx *ctx, uint64_t *result_out,
                           const uint32_t **merged_visible_dev, const uint64_t **merged_keys_dev);
/* Diagnosis of a stalled context (no reference counterpart), callable from another
 * host thread while the owner is blocked in a wait: 1 = busy / not reached, 0 =
 * idle / reached, -1 = does not exist, for out[0..2] the context's stream, its
 * second stream and the exchange stream, out[3] the event behind the last classify
 * launch, out[4..5] the events behind the second stream's batches, out[6..9] the
 * packed / merged events of the two exchange buffer sets. Waits for nothing. */
int vkrhip_debug_streams(vkrhip_ctx *ctx, int32_t *out, uint32_t n);

/* ---- sharding a hierarchy over GPUs (no reference counterpart: the reference
 * runs on one GPU) -----------------------------------------------------------
 * Cuts a hierarchy in BFS order (the order vkrhip_levelize produces: parents
 * ascend with their children, every parent one level up) into `shards` pieces
 * that need no communication to propagate transforms: levels [0,
 * replicated_levels) - a handful of nodes - go to every shard whole; the nodes of
 * level `replicated_levels` are dealt out in contiguous runs of about equal
 * subtree weight, and every deeper level follows its parents, so a shard owns
 * whole subtrees. out_ranges[(s * levels + l) * 2 + {0, 1}] = [begin, end) of
 * level l on shard s, in the numbering of the input. A shard's scene is the
 * concatenation of its ranges level by level (parents renumbered accordingly);
 * which shard draws the rows of a replicated node is the caller's choice.
 * Returns VKRHIP_ERR_STATE when the numbering is not of that shape. Host side. */
int vkrhip_shard_hierarchy(const uint32_t *parent, uint64_t n, const uint64_t *level_offsets,
                           uint32_t levels, uint32_t shards, uint32_t replicated_levels,
                           uint64_t *out_ranges);

/* ---- the exchange merge without an allocation -------------------------------
 * No counterpart in the reference. For callers that cannot allocate device
 * memory where they call: code under HIP graph capture, tight serving loops,
 * harnesses that forbid hipMalloc. vkrhip_exchange_merge_static is
 * vkrhip_exchange_merge without a context: same inputs, same outputs (merged
 * visible lists, merged draw keys, result words as laid out above), but the
 * planning tables its two kernels share live in a device array inside the
 * library (VKRHIP_STATIC_SCRATCH_BYTES on every GPU the library is used on, held
 * until the process ends) instead of buffers a context allocates. It calls no
 * hipMalloc / hipFree, here or in anything it calls. Limits: world <=
 * VKRHIP_STATIC_MAX_WORLD ranks and views <= VKRHIP_MAX_VIEWS; beyond them it
 * returns VKRHIP_ERR_CAPACITY (message: vkrhip_last_error(NULL)).
 *   device       HIP device the pointers live on; made current by the call
 *   views        view count the blocks were packed for
 *   hip_stream   stream the two kernels are enqueued on (NULL: the null stream)
 * All calls on one GPU share that GPU's tables: calls that overlap on one GPU
 * (different streams or threads) race; calls on one stream are ordered.
 * vkrhip_static_init(device) looks the array's address up ahead of time (e.g.
 * before a graph capture starts); otherwise the first call on a device does. */
#define VKRHIP_STATIC_MAX_WORLD 8u
#define VKRHIP_STATIC_SCRATCH_BYTES 56352u
int vkrhip_static_init(int device);
int vkrhip_exchange_merge_static(int device, uint32_t views, const uint32_t *gathered_dev,
                                 uint32_t world, uint32_t vis_cap, uint32_t key_cap,
                                 uint32_t *merged_visible_dev, uint64_t *merged_keys_dev,
                                 uint64_t *result_dev, void *hip_stream);

/* scene_entity_is_visible (vkr_scene_system.c:1935-1977) for every node of a
 * hierarchy stored parents-first (the order vkrhip_levelize produces): walking up
 * from a node, the first SceneVisibility with visible == 0 hides it, the first
 * one with inherit_parent == 0 ends the walk as visible, nodes without the
 * component pass the question to their parent, the top of the chain is visible.
 * spec[i]: bit 0 = the node has a SceneVisibility, bit 1 = visible, bit 2 =
 * inherit_parent. out_visible[i] = 0/1. Host side, like the reference (its
 * result decides which rows application_build_world_payload enumerates,
 * application.h:688,734,772-774, i.e. the numbering of source rows: the
 * caller re-runs vkrhip_set_candidates with the rows of the visible nodes
 * after vkr_scene_set_visibility, which the reference answers with a full
 * render-bridge sync as well, vkr_scene_system.c:1634-1655). */
void vkrhip_effective_visibility(const uint32_t *parent, const uint8_t *spec, uint64_t n,
                                 uint8_t *out_visible);

/* The box around the world spheres of every row that casts shadows (flag
 * VKRHIP_CAND_SHADOW_CASTER, valid bounds), accumulated by the cull of the last
 * frame when it ran with VKRHIP_STAGE_CASTER_BOUNDS: per axis the smallest
 * centre - radius and the largest centre + radius, each rounded once (so the
 * result does not depend on the order of accumulation). It is what the
 * reference's VkrShadowSceneBounds asks the application for
 * (vkr_shadow_system.h:85-99: the caster AABB vkr_shadow_fit_relevant_caster_z
 * clips every cascade's depth range against); copy it into
 * vkrhip_shadow_config::scene_min / scene_max with use_scene_bounds = 1. Waits
 * for the frame. Without any caster the box is empty (min > max). */
int vkrhip_get_caster_bounds(vkrhip_ctx *ctx, float *out_min3, float *out_max3);

/* ---- cascade fitting: the shadow views of the cull --------------------------
 * The reference's shadow system turns the camera, the light direction and a
 * configuration into one light view-projection matrix per cascade every frame
 * (vkr_shadow_system_update, lib/src/renderer/systems/vkr_shadow_system.c:654-777);
 * application_draw_frame extracts a plane set from each and the cull classifies
 * shadow casters against them. These entry points are that step, on the host
 * (O(cascades) float work), with the reference's names and meaning:
 *   vkrhip_shadow_system_init / _update / _invalidate_fit_history
 *       vkr_shadow_system_init (:591-645), vkr_shadow_system_update (:654-777),
 *       vkr_shadow_system_invalidate_fit_history (:413-417)
 *   vkrhip_shadow_fit_cascade          vkr_shadow_compute_cascade_matrix (:419-589)
 *   vkrhip_shadow_fit_relevant_caster_z  vkr_shadow_fit_relevant_caster_z (:234-346)
 *   vkrhip_shadow_quantize_extent_up   vkr_shadow_quantize_extent_up (:348-360)
 *   vkrhip_shadow_apply_fit_hysteresis vkr_shadow_apply_fit_hysteresis (:362-396)
 * Vector normalisation uses the reference's portable formula (vkr_simd.h:1112-
 * 1116), not a CPU's reciprocal-square-root estimate; given the same light view
 * and slice corners every later step is bit for bit the reference's. */
#define VKRHIP_SHADOW_CASCADES_MAX 8u /* VKR_SHADOW_CASCADE_COUNT_MAX */
enum { VKRHIP_CAMERA_PERSPECTIVE = 0, VKRHIP_CAMERA_ORTHOGRAPHIC = 1 };

typedef struct vkrhip_shadow_fit { /* VkrShadowFit, vkr_shadow_system.h:32-39 */
  float center_x, center_y, extent, min_z, max_z, world_units_per_texel;
} vkrhip_shadow_fit;

typedef struct vkrhip_shadow_config { /* the fields of VkrShadowConfig (:151-177) the fit reads */
  uint32_t cascade_count;
  uint32_t shadow_map_size;
  float cascade_split_lambda;
  float max_shadow_distance;
  float cascade_guard_band_texels;
  float cascade_guard_band_texels_per[VKRHIP_SHADOW_CASCADES_MAX];
  float z_extension_factor;
  float cascade_z_extension_factor_per[VKRHIP_SHADOW_CASCADES_MAX];
  uint32_t use_constant_cascade_size;
  float anchor_snap_texels;
  uint32_t stabilize_cascades;
  uint32_t use_scene_bounds; /* VkrShadowSceneBounds, :95-99 */
  float scene_min[3];
  float scene_max[3];
} vkrhip_shadow_config;

typedef struct vkrhip_shadow_camera { /* what the fit reads of VkrCamera */
  uint32_t type; /* VKRHIP_CAMERA_* */
  float position[3], forward[3], right[3], up[3];
  float near_clip, far_clip;
  float zoom;                           /* vertical field of view in degrees */
  uint32_t window_width, window_height; /* aspect ratio (cached_window_width / _height) */
  float left_clip, right_clip, bottom_clip, top_clip; /* orthographic cameras */
} vkrhip_shadow_camera;

typedef struct vkrhip_shadow_cascade { /* VkrCascadeData, :74-83 */
  float view_projection[16];
  float split_far;
  float world_units_per_texel;
  float light_space_origin[2];
  float light_space_depth_span;
  float bounds_center[3];
  float bounds_radius;
} vkrhip_shadow_cascade;

typedef struct vkrhip_shadow_system { /* VkrShadowSystem with its VkrShadowFitHistory, :52-60,314-328 */
  vkrhip_shadow_config config;
  vkrhip_shadow_cascade cascades[VKRHIP_SHADOW_CASCADES_MAX];
  float cascade_splits[VKRHIP_SHADOW_CASCADES_MAX + 1];
  float light_direction[3];
  uint32_t light_enabled;
  vkrhip_shadow_fit history_fits[VKRHIP_SHADOW_CASCADES_MAX];
  float history_light_direction[3];
  uint32_t history_cascade_count;
  uint32_t history_shadow_map_size;
  uint32_t history_projection_convention;
  uint32_t history_valid;
  uint64_t history_enable_generation;
  uint64_t enable_generation;
  uint32_t initialized;
  uint32_t reserved;
} vkrhip_shadow_system;

/* VKR_SHADOW_CONFIG_HIGH, the reference's default (vkr_shadow_system.h:194-217,255). */
void vkrhip_shadow_config_high(vkrhip_shadow_config *out);
/* config NULL: the default preset. Clamps the configuration like the reference. */
int vkrhip_shadow_system_init(vkrhip_shadow_system *s, const vkrhip_shadow_config *config);
void vkrhip_shadow_system_invalidate_fit_history(vkrhip_shadow_system *s);
/* Fills s->cascades[0 .. config.cascade_count) and s->cascade_splits. With
 * light_enabled == 0 the cascades are reset (identity matrices) and the fit
 * history of the enabled interval is abandoned, like the reference. */
int vkrhip_shadow_system_update(vkrhip_shadow_system *s, const vkrhip_shadow_camera *camera,
                                uint32_t light_enabled, const float *light_direction);
/* One cascade from a light view matrix and the eight corners of the camera
 * frustum slice (near TL/TR/BR/BL, far TL/TR/BR/BL, 3 floats each). config
 * supplies the scene bounds; previous_fit NULL = no hysteresis. */
void vkrhip_shadow_fit_cascade(const float *light_view, const float *corners24,
                               uint32_t shadow_map_size, int stabilize, float guard_band_texels,
                               int use_constant_cascade_size, const vkrhip_shadow_config *config,
                               float z_extension_factor, const vkrhip_shadow_fit *previous_fit,
                               float *out_view_projection, vkrhip_shadow_fit *out_fit,
                               float *out_light_space_origin);
/* Returns 0 when the box and the rectangle's column do not overlap. */
int vkrhip_shadow_fit_relevant_caster_z(const float *light_view, const float *bounds_min,
                                        const float *bounds_max, float left, float right,
                                        float bottom, float top, float *out_min_z, float *out_max_z);
float vkrhip_shadow_quantize_extent_up(float extent, uint32_t shadow_map_size);
void vkrhip_shadow_apply_fit_hysteresis(const vkrhip_shadow_fit *previous, const vkrhip_shadow_fit *raw,
                                        vkrhip_shadow_fit *out);
/* The view set of a frame from the camera matrices and the fitted cascades
 * (vkrhip_build_view_set over s->cascades): planes[(1 + cascades) * 24],
 * required_flags[1 + cascades], *views = 1 + cascades (1 while the light is off). */
int vkrhip_shadow_build_view_set(const vkrhip_shadow_system *s, const float *view,
                                 const float *projection, float *planes, uint32_t *required_flags,
                                 uint32_t *views);

/* ---- glTF node hierarchy -----------------------------------------------------
 * "nodes" and "scenes" of a glTF 2.0 document (the JSON text of a .gltf file or
 * of a .glb's JSON chunk) as per-node arrays in file order: what the reference
 * gets from cgltf (vendor/cgltf.h: cgltf_node parent / children / translation /
 * rotation / scale / matrix / mesh, cgltf_scene nodes) when it imports such a
 * file, read by a JSON walker of the library's own. Host code.
 *   scene      index into "scenes"; 0xffffffff = the document's "scene", else 0
 *   capacity   room in the arrays; with fewer, info is filled and
 *              VKRHIP_ERR_CAPACITY returned (call once with 0 to size them)
 *   parent     glTF index of the node's parent, 0xffffffff for a root
 *   in_scene   1 when the chosen scene reaches the node (every node when the
 *              document has no scenes)
 *   pos4 / quat4 (x, y, z, w) / scale4   the node's TRS, glTF defaults where absent
 *   mesh       index into "meshes", -1 for none
 *   has_matrix / matrix16   a node given as a column-major matrix instead
 * Any array may be NULL. Numbers are rounded once (decimal text -> double ->
 * float). VKRHIP_ERR_ARG: not JSON, an index out of range, a node with two
 * parents, a cycle, a scene that lists a non-root. The parent array goes to
 * vkrhip_levelize as it is (restricted to the scene's nodes by the caller). */
typedef struct vkrhip_gltf_nodes {
  uint32_t node_count;  /* "nodes" of the document */
  uint32_t scene_count; /* "scenes" of the document */
  uint32_t scene;       /* the scene that was read */
  uint32_t scene_nodes; /* nodes it reaches */
} vkrhip_gltf_nodes;
int vkrhip_gltf_read_nodes(const char *json, uint64_t length, uint32_t scene, uint32_t capacity,
                           vkrhip_gltf_nodes *info, uint32_t *parent, uint8_t *in_scene, float *pos4,
                           float *quat4, float *scale4, int32_t *mesh, uint8_t *has_matrix,
                           float *matrix16);

/* The bounds of every mesh primitive of a glTF document, in file order (mesh after
 * mesh, primitive after primitive): center / min_extents / max_extents as the
 * reference stores them per submesh (VkrMeshAssetSubmesh,
 * lib/src/renderer/resources/vkr_resources.h:292-294), from the "min" / "max" of the
 * primitive's POSITION accessor (cgltf_accessor.min / .max, vendor/cgltf.h) - the
 * values the reference's importer arrives at by a minimum / maximum over the vertex
 * positions (vkr_mesh_loader_compute_bounds,
 * lib/src/renderer/resources/loaders/mesh_loader.c:1159-1179; centre = (min + max) *
 * 0.5 there and here). Host code, the JSON walker of vkrhip_gltf_read_nodes.
 *   capacity   room in the arrays; with fewer, info is filled and
 *              VKRHIP_ERR_CAPACITY returned (call once with 0 to size them)
 *   mesh / primitive   index into "meshes" / into that mesh's "primitives"
 *   material   index into "materials", -1 for none
 *   has_bounds 1 when the POSITION accessor carries a three-component min and max
 *   bounds12   three padded Vec3 per primitive - the layout vkrhip_set_candidates
 *              takes: center, min_extents, max_extents; zero without bounds
 * Any array may be NULL. VKRHIP_ERR_ARG: not JSON, or an accessor index out of range. */
typedef struct vkrhip_gltf_meshes {
  uint32_t mesh_count;      /* "meshes" of the document */
  uint32_t primitive_count; /* primitives of all meshes */
  uint32_t accessor_count;  /* "accessors" of the document */
  uint32_t bounded_count;   /* primitives with bounds */
} vkrhip_gltf_meshes;
int vkrhip_gltf_read_mesh_bounds(const char *json, uint64_t length, uint32_t capacity,
                                 vkrhip_gltf_meshes *info, uint32_t *mesh, uint32_t *primitive,
                                 int32_t *material, uint8_t *has_bounds, float *bounds12);

/* ---- scene files ---------------------------------------------------------------
 * "entities" of one of the reference's scene files (assets/scenes/ *.scene.json,
 * version 2) as per-entity arrays in file order: what scene_json_parse_parent /
 * _transform / _mesh / _shape take from an entity
 * (lib/src/renderer/resources/loaders/scene_loader.c:1243-1310,1311-1330,1417-1475),
 * restricted to what the scene path consumes. Host code, the JSON walker of the
 * glTF reader.
 *   capacity     room in the arrays; with fewer, info is filled and
 *                VKRHIP_ERR_CAPACITY returned (call once with 0 to size them)
 *   parent       index of the parent entity, 0xffffffff for none ("parent" absent or null)
 *   pos4 / quat4 (x, y, z, w) / scale4   "transform": "pos", "rot", "scale";
 *                0 / identity / 1 where absent
 *   kind         0 nothing to draw, 1 a "shape", 2 a "mesh" (a shape wins)
 *   dimensions3  "dimensions" of a shape ((1, 1, 1) where absent): a cube's bounds are
 *                +-dimensions / 2 (lib/src/renderer/systems/vkr_geometry_system.c:588-590)
 *   material_path  two words per entity: offset and length, in bytes of `json`, of
 *                the text of a shape's "material" "path" (between the quotes, escapes
 *                not decoded); 0, 0 for none
 * Any array may be NULL. A member given as null counts as absent, as in the
 * reference. VKRHIP_ERR_ARG: not JSON, a parent index out of range or a cycle. */
typedef struct vkrhip_scene_entities {
  uint32_t version;          /* "version" of the document (the reference reads 2) */
  uint32_t entity_count;     /* "entities" */
  uint32_t renderable_count; /* entities with a shape or a mesh */
} vkrhip_scene_entities;
int vkrhip_scene_read_entities(const char *json, uint64_t length, uint32_t capacity,
                               vkrhip_scene_entities *info, uint32_t *parent, float *pos4, float *quat4,
                               float *scale4, uint8_t *kind, float *dimensions3,
                               uint64_t *material_path);

#ifdef __cplusplus
}
#endif
#endif
