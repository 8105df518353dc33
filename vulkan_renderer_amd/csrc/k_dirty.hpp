// k_dirty.hpp - the render-dirty list of a frame: the nodes whose world matrix the
// transform update rewrote, in topological order.
//
// vkr_scene_update hands every entity it updates to scene_mark_render_dirty
// (lib/src/renderer/systems/vkr_scene_system.c:458-479, called at :1335), which
// appends it to scene->render_dirty_list; the render bridge walks that list
// (:2123-2124). The update runs over the topological order, and the nodes here
// are stored in that order, so the list is the ascending sequence of node
// indices whose flags carry WORLD_UPDATED after the frame - an ordered
// compaction of one byte per node: count per 1024-node block, one scan over the
// block counts, ordered write.
#pragma once
#include "k_cull.hpp"
#include "k_transform.hpp"

namespace vkr {

constexpr uint32_t DIRTY_BLOCK = 256;
constexpr uint32_t DIRTY_NODES = DIRTY_BLOCK * 4u;  // four flag bytes per thread

struct DirtyArgs {
  const uint8_t *flags;    // the nodes' flags after the frame (padded to a multiple of 16 bytes)
  uint32_t *block_counts;  // [blocks]
  uint32_t *block_offsets; // [blocks] exclusive prefix (k_dirty_scan)
  uint32_t *list;          // out: node indices, ascending
  uint32_t *total;         // out: length of the list
  uint64_t n;
  uint32_t blocks;
};

// Bit k of the result: flag byte k of the word carries WORLD_UPDATED and its node
// lies inside the scene.
__device__ __forceinline__ uint32_t dirty_bits_of(const DirtyArgs &a, uint64_t first_node) {
  if (first_node >= a.n) return 0u;
  const uint32_t w = *reinterpret_cast<const uint32_t *>(a.flags + first_node);
  uint32_t bits = 0u;
#pragma unroll
  for (uint32_t k = 0; k < 4u; ++k)
    if (((w >> (8u * k)) & F_WORLD_UPDATED) && first_node + k < a.n) bits |= 1u << k;
  return bits;
}

__global__ __launch_bounds__(DIRTY_BLOCK) void k_dirty_count(DirtyArgs a) {
  __shared__ uint32_t s_wave[DIRTY_BLOCK / 64];
  const uint64_t node = ((uint64_t)blockIdx.x * DIRTY_BLOCK + threadIdx.x) * 4u;
  const uint32_t mine = (uint32_t)__popc(dirty_bits_of(a, node));
  const uint32_t wave_total = (uint32_t)__builtin_amdgcn_readlane((int)wave_inclusive_scan(mine), 63);
  if (lane_id() == 0) s_wave[threadIdx.x >> 6] = wave_total;
  __syncthreads();
  if (threadIdx.x == 0) a.block_counts[blockIdx.x] = s_wave[0] + s_wave[1] + s_wave[2] + s_wave[3];
}

// Exclusive prefix of the block counts by one block, 1024 counts per sweep.
__global__ __launch_bounds__(SCAN_BLOCK) void k_dirty_scan(DirtyArgs a) {
  __shared__ uint32_t s_wave[SCAN_BLOCK / 64];
  uint32_t carry = 0u;
  for (uint32_t base = 0; base < a.blocks; base += SCAN_BLOCK) {  // block-uniform
    const uint32_t i = base + threadIdx.x;
    const uint32_t v = i < a.blocks ? a.block_counts[i] : 0u;
    uint32_t total;
    const uint32_t before = block_exclusive_scan(v, s_wave, &total);
    if (i < a.blocks) a.block_offsets[i] = carry + before;
    carry += total;
  }
  if (threadIdx.x == 0) *a.total = carry;
}

__global__ __launch_bounds__(DIRTY_BLOCK) void k_dirty_write(DirtyArgs a) {
  __shared__ uint32_t s_wave[DIRTY_BLOCK / 64];
  const uint64_t node = ((uint64_t)blockIdx.x * DIRTY_BLOCK + threadIdx.x) * 4u;
  const uint32_t bits = dirty_bits_of(a, node);
  const uint32_t mine = (uint32_t)__popc(bits);
  const uint32_t upto = wave_inclusive_scan(mine);
  if (lane_id() == 63u) s_wave[threadIdx.x >> 6] = upto;
  __syncthreads();
  uint32_t pos = a.block_offsets[blockIdx.x] + upto - mine;
  for (uint32_t w = 0; w < (threadIdx.x >> 6); ++w) pos += s_wave[w];
#pragma unroll
  for (uint32_t k = 0; k < 4u; ++k)
    if (bits & (1u << k)) a.list[pos++] = (uint32_t)(node + k);
}

}  // namespace vkr
