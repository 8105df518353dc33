// k_transform.hpp - depth-levelled world-transform propagation for gfx950.
//
// Replaces the two passes of vkr_scene_update
// (lib/src/renderer/systems/vkr_scene_system.c:580-604 and :1293-1336). Nodes
// are stored in the reference's BFS topo order, so one launch per hierarchy
// level sees finished parents. Data layout in HBM is one array per attribute
// (local[], world[], parent[], flags[], TRS), each Mat4 as 4 consecutive float4
// columns; 4 lanes cooperate on a node, lane j owning column j, so every wave
// load/store instruction touches 1 KiB of consecutive bytes.
//
// Parent world matrices of a 256-node tile are staged once in LDS when the
// tile's parents form a window of at most 256 nodes (always true for the BFS
// order, where siblings are contiguous and parents ascend); otherwise lanes
// fetch their parent through the cache hierarchy.
#pragma once
#include "device_math.hpp"

namespace vkr {

constexpr int TF_BLOCK = 256;   // threads
constexpr int TF_TILE = 256;    // nodes per block (4 rounds of 64 nodes)
constexpr int TF_WINDOW = 256;  // LDS-staged parent window (nodes)

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
  uint8_t *flags;
  uint32_t *counters;  // sharded frame counters; blocks add to NODES_UPDATED of their shard
  uint64_t begin;  // level range [begin, end)
  uint64_t end;
  uint32_t force;  // flag bits OR-ed into every node first (bulk "everything moved")
};

template <bool HAS_PARENT>
__global__ __launch_bounds__(TF_BLOCK) void k_propagate_level(TransformArgs a) {
  __shared__ float4 s_pworld[HAS_PARENT ? TF_WINDOW * 4 : 1];
  __shared__ uint8_t s_pflags[HAS_PARENT ? TF_WINDOW : 1];
  __shared__ uint32_t s_par[HAS_PARENT ? TF_TILE : 1];
  __shared__ uint32_t s_min, s_max, s_updated;

  const uint32_t t = threadIdx.x;
  const uint64_t tile_base = a.begin + (uint64_t)blockIdx.x * TF_TILE;
  if (t == 0) {
    s_min = 0xffffffffu;
    s_max = 0u;
    s_updated = 0u;
  }
  __syncthreads();

  bool window = false;
  uint32_t pmin = 0;
  if (HAS_PARENT) {
    const uint64_t i = tile_base + t;
    uint32_t p = NO_PARENT;
    if (i < a.end) p = a.parent[i];
    s_par[t] = p;
    if (p != NO_PARENT) {
      atomicMin(&s_min, p);
      atomicMax(&s_max, p);
    }
    __syncthreads();
    pmin = s_min;
    const uint32_t pmax = s_max;
    window = pmin <= pmax && (pmax - pmin) < (uint32_t)TF_WINDOW;
    if (window) {
      const uint32_t span = pmax - pmin + 1u;
      const float4 *src = a.world + (uint64_t)pmin * 4u;
      for (uint32_t k = t; k < span * 4u; k += TF_BLOCK) s_pworld[k] = src[k];
      for (uint32_t k = t; k < span; k += TF_BLOCK) s_pflags[k] = a.flags[(uint64_t)pmin + k];
    }
    __syncthreads();
  }

  // The four sub-rounds are written as separate passes so that every load of
  // the tile is in flight before the first result is consumed.
  const uint32_t col = t & 3u;
  uint8_t f[4];
  bool valid[4];
  uint32_t par[4];
  float4 lcol[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const uint32_t slot = (uint32_t)r * 64u + (t >> 2);
    const uint64_t node = tile_base + slot;
    valid[r] = node < a.end;
    f[r] = valid[r] ? (uint8_t)(a.flags[node] | a.force) : (uint8_t)0;
    par[r] = NO_PARENT;
    if (HAS_PARENT) par[r] = s_par[slot];
  }
  // pass 1 flag handling (vkr_scene_system.c:592-603) and the parent-updated
  // test of pass 2 (:1310-1316)
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    f[r] &= (uint8_t)~F_WORLD_UPDATED;
    if (HAS_PARENT && valid[r] && par[r] != NO_PARENT) {
      const uint8_t pf = window ? s_pflags[par[r] - pmin] : a.flags[par[r]];
      if (pf & F_WORLD_UPDATED) f[r] |= F_DIRTY_WORLD;
    }
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const uint64_t node = tile_base + (uint32_t)r * 64u + (t >> 2);
    lcol[r] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (valid[r]) {
      if (f[r] & F_DIRTY_LOCAL) {
        if (a.pos != nullptr) {
          lcol[r] = local_col_from_trs(a.pos[node], a.quat[node], a.scale[node], col);
          a.local[node * 4u + col] = lcol[r];
        } else {
          lcol[r] = a.local[node * 4u + col];
        }
        f[r] = (uint8_t)((f[r] & ~F_DIRTY_LOCAL) | F_DIRTY_WORLD);
      } else if (f[r] & F_DIRTY_WORLD) {
        lcol[r] = a.local[node * 4u + col];
      }
    }
  }
  // pass 2 proper, vkr_scene_system.c:1319-1336
  uint32_t updated_here = 0;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const uint64_t node = tile_base + (uint32_t)r * 64u + (t >> 2);
    bool did_update = false;
    if (valid[r] && (f[r] & F_DIRTY_WORLD)) {
      float4 wcol = lcol[r];
      if (HAS_PARENT && par[r] != NO_PARENT) {
        float4 a0, a1, a2, a3;
        if (window) {
          const float4 *pw = s_pworld + (par[r] - pmin) * 4u;
          a0 = pw[0];
          a1 = pw[1];
          a2 = pw[2];
          a3 = pw[3];
        } else {
          const float4 *pw = a.world + (uint64_t)par[r] * 4u;
          a0 = pw[0];
          a1 = pw[1];
          a2 = pw[2];
          a3 = pw[3];
        }
        wcol = mat4_mul_col(a0, a1, a2, a3, lcol[r]);
      }
      a.world[node * 4u + col] = wcol;
      f[r] = (uint8_t)((f[r] & ~F_DIRTY_WORLD) | F_WORLD_UPDATED);
      did_update = true;
    }
    if (valid[r] && col == 0) a.flags[node] = f[r];
    const unsigned long long m = __ballot(did_update && col == 0);
    if ((t & 63u) == 0) updated_here += (uint32_t)__popcll(m);
  }
  if ((t & 63u) == 0 && updated_here) atomicAdd(&s_updated, updated_here);
  __syncthreads();
  if (t == 0 && s_updated)
    atomicAdd(a.counters + (blockIdx.x % TF_STAT_SHARDS) * TF_CNT_STRIDE + 0u, s_updated);
}

// scene_compute_local_matrix over a node range, without touching flags or
// world matrices (vkr_scene_system.c:555-561). Used at upload time when the
// caller wants cached locals materialised immediately.
__global__ __launch_bounds__(256) void k_local_from_trs(const float4 *pos, const float4 *quat,
                                                        const float4 *scale, float4 *local,
                                                        uint64_t begin, uint64_t end) {
  const uint64_t lane = (uint64_t)blockIdx.x * 256u + threadIdx.x;
  const uint64_t node = begin + (lane >> 2);
  const uint32_t col = threadIdx.x & 3u;
  if (node >= end) return;
  local[node * 4u + col] = local_col_from_trs(pos[node], quat[node], scale[node], col);
}

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
