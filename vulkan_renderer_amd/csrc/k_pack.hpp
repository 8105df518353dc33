// k_pack.hpp - the backend's candidate pack and the encode outputs of the
// multi-view cull.
//
// vkr_vk_pack_gpu_candidates (lib/src/renderer/vulkan/vkr_vulkan_draws.c:344-428)
// turns every source row whose geometry and material are published into a
// 48-byte VkrGpuCandidateDrawRow (lib/src/renderer/vkr_gpu_abi.h:70-80) plus its
// 80-byte VkrInstanceDataGPU; rows that do not resolve are omitted and the rest
// close ranks (instance_index = packed index). vk_gpu_draw_encode
// (lib/src/renderer/shaders/vulkan/slang/world/deferred.slang:379-420) writes a
// 32-byte VkrGpuVisibleDrawRow and a 20-byte indirect command per surviving
// (row, view).
//
// Resolution depends only on the handles and the publication tables, so it is
// done once per publication change (k_pack_resolve, k_pack_index); the per-frame
// work is two streaming kernels: k_emit_packed writes the rows (the instance
// carries this frame's world matrix) and k_encode_visible expands the compacted
// visible lists. Unpublished rows carry CF_UNPUBLISHED in their candidate word,
// which keeps them out of every view, and the visible lists then hold packed
// indices, exactly as if the cull had run on the packed rows.
#pragma once
#include "k_cull.hpp"
#include "k_sort.hpp"

namespace vkr {

// POD flattening of VkrVulkanPublishedGeometry / VkrVulkanSubmeshRange /
// VkrVulkanPublishedMaterial (lib/src/renderer/vulkan/vkr_vulkan_internal.h:838-870)
struct GeometryEntry {
  uint32_t generation;
  uint32_t resolvable;  // live && pending_initialization_count == 0
  uint32_t first_index;
  uint32_t submesh_count;
  uint32_t submesh_offset;
  uint32_t pad[3];
};
struct SubmeshRange {
  uint32_t first_index;
  uint32_t index_count;
  int32_t vertex_offset;
  uint32_t pad;
};
struct MaterialEntry {
  uint32_t generation;
  uint32_t live;
  uint32_t slot_index;
  uint32_t pad;
};
static_assert(sizeof(GeometryEntry) == 32 && sizeof(SubmeshRange) == 16 && sizeof(MaterialEntry) == 16,
              "publication table layouts");

enum : uint8_t { PACK_OK = 0, PACK_NO_GEOMETRY = 1, PACK_NO_MATERIAL = 2, PACK_BAD_SUBMESH = 3 };

struct PackResolveArgs {
  const uint4 *ident;  // 2 x uint4 per candidate
  const GeometryEntry *geometries;
  const SubmeshRange *submeshes;
  const MaterialEntry *materials;
  uint32_t *cflags;
  uint16_t *cflags16;  // (the copy the fused kernel reads)
  uint4 *pack_a;     // geometry_index, material_index, first_index, index_count
  int32_t *pack_vo;  // vertex_offset
  uint8_t *status;
  uint64_t n;
  uint32_t geometry_capacity;
  uint32_t material_capacity;
};

// vkr_vk_resolve_geometry / vkr_vk_resolve_material and the row fields,
// vkr_vulkan_draws.c:321-342,370-400
__global__ __launch_bounds__(256) void k_pack_resolve(PackResolveArgs a) {
  const uint64_t i = (uint64_t)blockIdx.x * 256u + threadIdx.x;
  if (i >= a.n) return;
  const uint4 id0 = a.ident[i * 2u];       // mesh.id, mesh.generation, geometry.id, geometry.generation
  const uint4 id1 = a.ident[i * 2u + 1u];  // submesh_index, material.id, material.generation, render_id
  uint8_t status = PACK_OK;
  uint4 row = make_uint4(0u, 0u, 0u, 0u);
  int32_t vertex_offset = 0;
  bool geometry_ok = false, material_ok = false;
  GeometryEntry g{};
  if (id0.z != 0u && id0.z <= a.geometry_capacity) {
    g = a.geometries[id0.z - 1u];
    geometry_ok = g.resolvable != 0u && g.generation == id0.w;
  }
  MaterialEntry m{};
  if (id1.y != 0u && id1.y <= a.material_capacity) {
    m = a.materials[id1.y - 1u];
    material_ok = m.live != 0u && m.generation == id1.z;
  }
  if (!geometry_ok) {
    status = PACK_NO_GEOMETRY;
  } else if (!material_ok) {
    status = PACK_NO_MATERIAL;
  } else if (id1.x >= g.submesh_count) {
    status = PACK_BAD_SUBMESH;
  } else {
    const SubmeshRange sub = a.submeshes[g.submesh_offset + id1.x];
    row = make_uint4(id0.z - 1u, m.slot_index, g.first_index + sub.first_index, sub.index_count);
    vertex_offset = sub.vertex_offset;
  }
  a.status[i] = status;
  a.pack_a[i] = row;
  a.pack_vo[i] = vertex_offset;
  const uint32_t cf = a.cflags[i] & ~CF_UNPUBLISHED;
  a.cflags[i] = status == PACK_OK ? cf : (cf | CF_UNPUBLISHED);
  a.cflags16[i] = (uint16_t)(status == PACK_OK ? cf : (cf | CF_UNPUBLISHED));
}

// Packed index of every candidate (exclusive prefix over the rows that
// resolve) and the four pack counters. One block; runs when the publication
// tables change, not per frame.
__global__ __launch_bounds__(SCAN_BLOCK) void k_pack_index(const uint8_t *status,
                                                           uint32_t *packed_index, uint64_t n,
                                                           uint32_t *counts4) {
  __shared__ uint32_t s_wave[SCAN_BLOCK / 64];
  __shared__ uint32_t s_other[4];
  if (threadIdx.x < 4) s_other[threadIdx.x] = 0u;
  __syncthreads();
  uint32_t carry = 0u;
  uint32_t mine[4] = {0u, 0u, 0u, 0u};
  for (uint64_t base = 0; base < n; base += SCAN_BLOCK) {
    const uint64_t i = base + threadIdx.x;
    const uint8_t st = i < n ? status[i] : (uint8_t)0xff;
    if (st < 4) mine[st] += 1u;
    uint32_t total;
    const uint32_t ex = block_exclusive_scan(st == PACK_OK ? 1u : 0u, s_wave, &total);
    if (i < n) packed_index[i] = carry + ex;
    carry += total;
  }
  for (int k = 1; k < 4; ++k)
    if (mine[k]) atomicAdd(&s_other[k], mine[k]);
  __syncthreads();
  if (threadIdx.x == 0) {
    counts4[0] = carry;
    counts4[1] = s_other[1];
    counts4[2] = s_other[2];
    counts4[3] = s_other[3];
  }
}

struct PackEmitArgs {
  const uint8_t *status;
  const uint32_t *packed_index;
  const uint4 *pack_a;
  const int32_t *pack_vo;
  const uint32_t *cflags;
  const float4 *sphere;
  const uint4 *ident;
  const float4 *world;
  const uint32_t *node;
  uint4 *rows;        // 3 x uint4 per packed row: VkrGpuCandidateDrawRow
  float4 *instances;  // 5 x float4 per packed row: VkrInstanceDataGPU
  uint64_t n;
};

// Eight lanes per source row: lanes 0-2 write the three 16-byte pieces of the
// 48-byte row, lanes 3-7 the five of the instance.
__global__ __launch_bounds__(256) void k_emit_packed(PackEmitArgs a) {
  const uint64_t i = (uint64_t)blockIdx.x * 32u + (threadIdx.x >> 3);
  const uint32_t piece = threadIdx.x & 7u;
  if (i >= a.n || a.status[i] != PACK_OK) return;
  const uint32_t p = a.packed_index[i];
  if (piece < 3u) {
    uint4 o;
    if (piece == 0u) {
      const uint4 r = a.pack_a[i];
      o = make_uint4(r.x, r.y, p, r.z);  // geometry_index, material_index, instance_index, first_index
    } else if (piece == 1u) {
      const uint32_t cf = a.cflags[i];
      o = make_uint4(a.pack_a[i].w, (uint32_t)a.pack_vo[i], (cf >> CF_BUCKET_SHIFT) & 0xfu,
                     cf & CF_REF_MASK);  // index_count, vertex_offset, state_bucket, flags
    } else {
      const float4 s = a.sphere[i];
      o = make_uint4(__float_as_uint(s.x), __float_as_uint(s.y), __float_as_uint(s.z),
                     __float_as_uint(s.w));
    }
    a.rows[(uint64_t)p * 3u + piece] = o;
  } else {
    const uint32_t c = piece - 3u;
    float4 o;
    if (c < 4u) {
      const uint64_t nd = a.node ? a.node[i] : i;
      o = a.world[nd * 4u + c];
    } else {
      const uint4 id1 = a.ident[i * 2u + 1u];
      o = make_float4(__uint_as_float(encode_scene_object_id(id1.w)), 0.f, 0.f, 0.f);
    }
    a.instances[(uint64_t)p * 5u + c] = o;
  }
}

struct EncodeArgs {
  const uint32_t *visible;  // [view][visible_stride] packed row indices
  const CompactionState *states;
  const uint4 *rows;        // packed rows, 3 x uint4 each
  uint4 *visible_rows;      // [view][visible_stride] x 2 uint4: VkrGpuVisibleDrawRow
  uint32_t *commands;       // [(view*4+bucket)*partition + local] x 5 words, may be null
  uint64_t visible_stride;
  uint32_t views;
  uint32_t partition_capacity;
};

// Two lanes per visible slot (the two 16-byte halves of the 32-byte row); the
// lane of the second half also writes the indirect command. gridDim.y = views.
__global__ __launch_bounds__(256) void k_encode_visible(EncodeArgs a) {
  const uint32_t view = blockIdx.y;
  const CompactionState st = a.states[view];
  const uint64_t slot = ((uint64_t)blockIdx.x * 256u + threadIdx.x) >> 1;
  const uint32_t half = threadIdx.x & 1u;
  if (slot >= st.visible_count) return;
  // (the bucket's start is tracked beside the bucket: indexing the copied state
  // with a run-time bucket would turn it into an array in memory)
  uint32_t bucket = 0u, bucket_start = st.execution_ranges[0][0];
#pragma unroll
  for (uint32_t b = 1; b < 4u; ++b)
    if (st.execution_ranges[b][1] != 0u && slot >= st.execution_ranges[b][0]) {
      bucket = b;
      bucket_start = st.execution_ranges[b][0];
    }
  const uint32_t p = a.visible[(uint64_t)view * a.visible_stride + slot];
  const uint4 r0 = a.rows[(uint64_t)p * 3u];       // geometry, material, instance, first_index
  const uint4 r1 = a.rows[(uint64_t)p * 3u + 1u];  // index_count, vertex_offset, bucket, flags
  uint4 *out = a.visible_rows + ((uint64_t)view * a.visible_stride + slot) * 2u;
  if (half == 0u) {
    out[0] = r0;
  } else {
    out[1] = make_uint4(r1.x, r1.y, bucket, r1.w);
    if (a.commands != nullptr) {
      const uint32_t local = (uint32_t)slot - bucket_start;
      uint32_t *cmd = a.commands +
                      (((uint64_t)view * 4u + bucket) * a.partition_capacity + local) * 5u;
      cmd[0] = r1.x;  // index_count
      cmd[1] = 1u;    // instance_count
      cmd[2] = r0.w;  // first_index
      cmd[3] = r1.y;  // vertex_offset
      cmd[4] = 0u;    // first_instance
    }
  }
}

}  // namespace vkr
