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
  // "Stepping" levels (every node's parent is its predecessor's or the one after
  // it, as in a breadth-first order with every parent of the level above having a
  // child): the parents of 64 consecutive nodes are a first parent and a 64-bit
  // mask - bit k: node k + 1 of the chunk hangs from the next parent - 16 bytes per
  // chunk instead of 256. Entry of chunk g = (global node index / 64): words
  // 4 g .. 4 g + 2 of parent_steps (first parent, mask low, mask high); the pointer
  // is offset per level, and bits below the level's first node in a chunk are 0.
  // Null: the level reads parent[].
  const uint32_t *parent_steps;
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

// ORs `bits` into the flag byte of nodes [first, first + count) in both copies of
// the flags (the transform setters of the reference mark single transforms dirty,
// vkr_scene_system.h:70-72; this is the same for a run of nodes, on the device, in
// stream order).
__global__ __launch_bounds__(256) void k_or_flags(uint8_t *flags_a, uint8_t *flags_b, uint64_t first,
                                                  uint64_t count, uint8_t bits) {
  const uint64_t i = (uint64_t)blockIdx.x * 256u + threadIdx.x;
  if (i >= count) return;
  flags_a[first + i] |= bits;
  flags_b[first + i] |= bits;
}

// Ordering between the context's two streams without a wait packet. A stream-wait
// between two level launches costs the device 8.5 us even when the event was
// recorded long ago (the queue drains, the packet is evaluated, the next launch
// starts cold); two tiny kernels cost a third of that: the batch on the second
// stream ends with k_side_signal, which publishes the batch's serial number, and
// the context's stream runs k_side_gate - one wave that returns once the serial
// has arrived, normally at its first look - ahead of the launch that must come
// behind the batch. Kernel boundaries on either side make the batch's writes
// visible to that launch as they do between any two launches.
__global__ __launch_bounds__(64) void k_side_signal(uint64_t *flag, uint64_t serial) {
  if (threadIdx.x == 0) __hip_atomic_store(flag, serial, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
// (A gate that has looked 2^22 times - seconds - gives up, so that the grid always
// drains, and leaves a mark in *stalled that vkrhip_get_frame_info reports.)
__global__ __launch_bounds__(64) void k_side_gate(const uint64_t *flag, uint64_t serial, uint64_t *stalled) {
  if (threadIdx.x != 0) return;
  for (uint32_t looks = 0; looks < (1u << 22); ++looks) {
    if (__hip_atomic_load(flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) >= serial) return;
    __builtin_amdgcn_s_sleep(8);
  }
  __hip_atomic_store(stalled, serial, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

}  // namespace vkr
