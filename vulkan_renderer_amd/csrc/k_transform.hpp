// k_transform.hpp - arguments and helpers of the depth-levelled world-transform
// propagation for gfx950.
//
// The two passes of vkr_scene_update
// (lib/src/renderer/systems/vkr_scene_system.c:580-604 and :1293-1336) run in
// k_propagate_classify (k_fused.hpp), with or without its cull phase. Nodes are
// stored in the reference's BFS topo order, so one launch per hierarchy level
// sees finished parents. Data layout in HBM is one array per attribute (local[],
// world[], parent[], flags[], TRS), each Mat4 as 4 consecutive float4 columns; 4
// lanes cooperate on a node, lane j owning column j, so every wave load/store
// instruction touches 1 KiB of consecutive bytes.
#pragma once
#include "device_math.hpp"

namespace vkr {

constexpr uint8_t F_DIRTY_LOCAL = 0x01;
constexpr uint8_t F_DIRTY_WORLD = 0x02;
constexpr uint8_t F_WORLD_UPDATED = 0x08;
constexpr uint32_t NO_PARENT = 0xffffffffu;
constexpr uint32_t TF_STAT_SHARDS = 64;  // = STAT_SHARDS
constexpr uint32_t TF_CNT_STRIDE = 8;    // = CNT_COUNT; slot 0 = CNT_NODES_UPDATED

struct TransformArgs {
  const uint32_t *parent;  // global array, indexed by node
  const float4 *pos;       // TRS, may be null (locals supplied directly)
  const float4 *quat;
  const float4 *scale;
  float4 *local;  // 4 float4 per node
  float4 *world;
  const uint8_t *flags_in;  // the nodes' flags before the frame
  uint8_t *flags_out;       // ... and after it (another buffer: the host swaps the two)
  uint32_t *counters;  // sharded frame counters; blocks add to NODES_UPDATED of their shard
  uint64_t begin;  // level range [begin, end)
  uint64_t end;
  uint32_t force;  // flag bits OR-ed into every node first (bulk "everything moved")
};

__global__ __launch_bounds__(256) void k_or_flags(uint8_t *flags, uint64_t n, uint8_t bits) {
  // 16 flag bytes per thread
  const uint64_t i = ((uint64_t)blockIdx.x * 256u + threadIdx.x) * 16u;
  if (i + 16u <= n) {
    uint4 v = *reinterpret_cast<uint4 *>(flags + i);
    const uint32_t b = bits * 0x01010101u;
    v.x |= b;
    v.y |= b;
    v.z |= b;
    v.w |= b;
    *reinterpret_cast<uint4 *>(flags + i) = v;
  } else {
    for (uint64_t k = i; k < n; ++k) flags[k] |= bits;
  }
}

}  // namespace vkr
