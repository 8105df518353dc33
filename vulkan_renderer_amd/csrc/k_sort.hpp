// k_sort.hpp - transparent draw keys, stable LSD radix sort in LDS, draw emission.
//
// Key: ((uint64)bits(depth) << 32) | (source_index + 1), depth clamped >= 0
// (lib/src/application.h:573-587), ordered DESCENDING by the reference's
// comparator (lib/src/renderer/vkr_visibility.c:5-13) and qsort
// (application.h:940-942). Keys are unique, so any correct sort reproduces the
// reference's order.
//
// The visible blend rows arrive in ascending source index. A STABLE ascending
// sort on key bits [begin_bit, 64) followed by reading the result backwards
// therefore yields descending keys with larger source index first among equal
// depths - exactly descending order on the full 64-bit key - while only the 32
// depth bits have to be moved through the radix passes. A full 64-bit sort
// (begin_bit = 0) is the same code with eight passes.
//
// Per 8-bit pass: per-tile digit histogram -> exclusive scan over
// (digit, tile) -> stable scatter with wave-ballot ranking (64-wide match on
// the digit, per-wave digit counters in LDS). The element count lives in
// device memory, so no host round trip sits between cull and sort.
#pragma once
#include "device_math.hpp"
#include "k_cull.hpp"

namespace vkr {

constexpr int SORT_BLOCK = 256;
constexpr int SORT_ITEMS = 8;                          // keys per lane
constexpr int SORT_TILE = SORT_BLOCK * SORT_ITEMS;     // 2048 keys per block
constexpr int SORT_WAVE_KEYS = 64 * SORT_ITEMS;        // 512 consecutive keys per wave
constexpr int SORT_RADIX = 256;

struct KeygenArgs {
  const uint32_t *tlist;     // visible blend rows, ascending local candidate index
  const uint32_t *counters;  // CNT_TRANSPARENT
  const float4 *world;
  const uint32_t *node;
  const float4 *sphere;
  const float4 *view;  // camera view matrix columns
  uint64_t *keys;
  uint32_t *vals;
  uint64_t index_base;
};

// application_transparent_depth + application_pack_transparent_sort_key,
// application.h:573-587
__global__ __launch_bounds__(256) void k_keygen(KeygenArgs a) {
  const uint32_t count = a.counters[CNT_TRANSPARENT];
  const uint32_t stride = gridDim.x * 256u;
  Mat4 view;
  view.c[0] = a.view[0];
  view.c[1] = a.view[1];
  view.c[2] = a.view[2];
  view.c[3] = a.view[3];
  for (uint32_t k = blockIdx.x * 256u + threadIdx.x; k < count; k += stride) {
    const uint32_t i = a.tlist[k];
    const uint64_t nd = a.node ? a.node[i] : i;
    Mat4 m;
    m.c[0] = a.world[nd * 4u + 0];
    m.c[1] = a.world[nd * 4u + 1];
    m.c[2] = a.world[nd * 4u + 2];
    m.c[3] = a.world[nd * 4u + 3];
    const float4 ls = a.sphere[i];
    const float3 wc = mat4_mul_point(m, ls.x, ls.y, ls.z);
    const float vz = mat4_mul_vec4_z(view, wc.x, wc.y, wc.z, 1.0f);
    float depth = -vz;
    depth = depth > 0.0f ? depth : 0.0f;
    const uint32_t tie = (uint32_t)(a.index_base + i) + 1u;
    a.keys[k] = ((uint64_t)__float_as_uint(depth) << 32) | (uint64_t)tie;
    a.vals[k] = i;
  }
}

struct SortArgs {
  const uint64_t *keys_in;
  const uint32_t *vals_in;
  uint64_t *keys_out;
  uint32_t *vals_out;
  uint32_t *hist;            // [digit][tiles_stride]
  const uint32_t *counters;  // CNT_TRANSPARENT
  uint32_t tiles_stride;
  uint32_t shift;
};

__global__ __launch_bounds__(SORT_BLOCK) void k_radix_hist(SortArgs a) {
  __shared__ uint32_t s_hist[SORT_RADIX];
  const uint32_t count = a.counters[CNT_TRANSPARENT];
  const uint32_t tiles = (count + SORT_TILE - 1) / SORT_TILE;
  const uint32_t t = threadIdx.x;
  for (uint32_t tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
    s_hist[t] = 0u;
    __syncthreads();
    const uint32_t base = tile * SORT_TILE;
#pragma unroll
    for (int k = 0; k < SORT_ITEMS; ++k) {
      const uint32_t idx = base + (uint32_t)k * SORT_BLOCK + t;
      if (idx < count) atomicAdd(&s_hist[(uint32_t)(a.keys_in[idx] >> a.shift) & 0xffu], 1u);
    }
    __syncthreads();
    a.hist[t * a.tiles_stride + tile] = s_hist[t];
    __syncthreads();
  }
}

// Exclusive scan over hist[digit][0..tiles) in digit-major order, one block.
__global__ __launch_bounds__(SCAN_BLOCK) void k_radix_scan(SortArgs a) {
  __shared__ uint32_t s_wave[SCAN_BLOCK / 64];
  const uint32_t count = a.counters[CNT_TRANSPARENT];
  const uint32_t tiles = (count + SORT_TILE - 1) / SORT_TILE;
  if (tiles == 0) return;
  const uint64_t total_entries = (uint64_t)SORT_RADIX * tiles;
  uint32_t carry = 0;
  for (uint64_t base = 0; base < total_entries; base += SCAN_BLOCK) {
    const uint64_t e = base + threadIdx.x;
    uint32_t v = 0;
    uint64_t addr = 0;
    if (e < total_entries) {
      const uint32_t digit = (uint32_t)(e / tiles);
      const uint32_t tile = (uint32_t)(e - (uint64_t)digit * tiles);
      addr = (uint64_t)digit * a.tiles_stride + tile;
      v = a.hist[addr];
    }
    uint32_t total;
    const uint32_t ex = block_exclusive_scan(v, s_wave, &total);
    if (e < total_entries) a.hist[addr] = carry + ex;
    carry += total;
  }
}

// Stable scatter. Wave w of the block owns keys [w*512, w*512+512) of the tile
// and walks them in 8 steps of 64 consecutive keys, so (wave, step, lane) order
// is memory order.
__global__ __launch_bounds__(SORT_BLOCK) void k_radix_scatter(SortArgs a) {
  __shared__ uint32_t s_whist[4][SORT_RADIX];  // per-wave running digit counts
  __shared__ uint32_t s_base[SORT_RADIX];      // global base of (digit, tile)
  const uint32_t count = a.counters[CNT_TRANSPARENT];
  const uint32_t tiles = (count + SORT_TILE - 1) / SORT_TILE;
  const uint32_t t = threadIdx.x;
  const uint32_t wave = t >> 6;
  const uint32_t lane = t & 63u;
  const unsigned long long lt_mask = (1ull << lane) - 1ull;
  volatile uint32_t *whist = s_whist[wave];

  for (uint32_t tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
    __syncthreads();  // previous tile's readers are done
    for (int w = 0; w < 4; ++w) s_whist[w][t] = 0u;
    s_base[t] = a.hist[t * a.tiles_stride + tile];
    __syncthreads();

    uint64_t key[SORT_ITEMS];
    uint32_t val[SORT_ITEMS];
    uint32_t rank[SORT_ITEMS];
    const uint32_t wbase = tile * SORT_TILE + wave * SORT_WAVE_KEYS;
#pragma unroll
    for (int k = 0; k < SORT_ITEMS; ++k) {
      const uint32_t idx = wbase + (uint32_t)k * 64u + lane;
      const bool valid = idx < count;
      key[k] = valid ? a.keys_in[idx] : 0ull;
      val[k] = valid ? a.vals_in[idx] : 0u;
    }
#pragma unroll
    for (int k = 0; k < SORT_ITEMS; ++k) {
      const uint32_t idx = wbase + (uint32_t)k * 64u + lane;
      const bool valid = idx < count;
      const uint32_t digit = (uint32_t)(key[k] >> a.shift) & 0xffu;
      // lanes holding the same digit (invalid lanes match nobody)
      unsigned long long peers = __ballot(valid);
#pragma unroll
      for (int b = 0; b < 8; ++b) {
        const bool bit = (digit >> b) & 1u;
        const unsigned long long m = __ballot(bit);
        peers &= bit ? m : ~m;
      }
      const uint32_t before = whist[digit];
      rank[k] = before + (uint32_t)__popcll(peers & lt_mask);
      __builtin_amdgcn_wave_barrier();
      if (valid && (peers & lt_mask) == 0ull) whist[digit] = before + (uint32_t)__popcll(peers);
      __builtin_amdgcn_wave_barrier();
    }
    __syncthreads();
    // exclusive prefix of the per-wave digit totals across waves, plus the
    // global base of this tile's digit run
    {
      uint32_t run = s_base[t];
#pragma unroll
      for (int w = 0; w < 4; ++w) {
        const uint32_t c = s_whist[w][t];
        s_whist[w][t] = run;
        run += c;
      }
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < SORT_ITEMS; ++k) {
      const uint32_t idx = wbase + (uint32_t)k * 64u + lane;
      if (idx < count) {
        const uint32_t digit = (uint32_t)(key[k] >> a.shift) & 0xffu;
        const uint32_t pos = s_whist[wave][digit] + rank[k];
        a.keys_out[pos] = key[k];
        a.vals_out[pos] = val[k];
      }
    }
  }
}

struct EmitArgs {
  const uint64_t *keys;      // ascending-sorted
  const uint32_t *vals;      // local candidate index per key
  const uint32_t *counters;  // CNT_TRANSPARENT
  const float4 *world;
  const uint32_t *node;
  const uint4 *ident;  // 2 x uint4 per candidate
  uint4 *draws;        // 3 x uint4 per row: VkrDrawItem
  float4 *instances;   // 5 x float4 per row: VkrInstanceDataGPU
  uint64_t *out_keys;  // descending
  uint32_t *out_source;
};

// vkr_picking_encode_id(SCENE, render_id), systems/vkr_picking_ids.h:35-42;
// render_id 0 maps to object id 0 (application.h:904-907).
__host__ __device__ __forceinline__ uint32_t encode_scene_object_id(uint32_t render_id) {
  if (render_id == 0u) return 0u;
  if (render_id > 0x1ffffffeu) return 0u;
  return render_id + 1u;
}

// vkr_transparent_draw_emit, lib/src/renderer/vkr_visibility.c:15-35. Eight
// lanes per output row: lanes 0-2 write the three 16-byte pieces of the
// VkrDrawItem, lanes 3-7 the five of the VkrInstanceDataGPU, so each output
// array is written in consecutive 16-byte pieces.
__global__ __launch_bounds__(256) void k_emit_transparent(EmitArgs a) {
  const uint32_t count = a.counters[CNT_TRANSPARENT];
  const uint32_t rows_per_block = 256u / 8u;
  const uint32_t piece = threadIdx.x & 7u;
  for (uint32_t row = blockIdx.x * rows_per_block + (threadIdx.x >> 3); row < count;
       row += gridDim.x * rows_per_block) {
    const uint32_t slot = count - 1u - row;  // read the ascending result backwards
    const uint64_t key = a.keys[slot];
    const uint32_t i = a.vals[slot];
    if (piece < 3u) {
      uint4 o;
      if (piece == 0u) {
        o = a.ident[(uint64_t)i * 2u];  // mesh.id, mesh.generation, geometry.id, geometry.generation
      } else if (piece == 1u) {
        const uint4 id1 = a.ident[(uint64_t)i * 2u + 1u];
        o = make_uint4(id1.x, id1.y, id1.z, 1u);  // submesh_index, material, instance_count
      } else {
        o = make_uint4(row, 0u, (uint32_t)key, (uint32_t)(key >> 32));  // first_instance, pad, key
        a.out_keys[row] = key;
        a.out_source[row] = i;
      }
      a.draws[(uint64_t)row * 3u + piece] = o;
    } else {
      const uint32_t c = piece - 3u;
      float4 o;
      if (c < 4u) {
        const uint64_t nd = a.node ? a.node[i] : i;
        o = a.world[nd * 4u + c];
      } else {
        const uint4 id1 = a.ident[(uint64_t)i * 2u + 1u];
        o = make_float4(__uint_as_float(encode_scene_object_id(id1.w)), 0.f, 0.f, 0.f);
      }
      a.instances[(uint64_t)row * 5u + c] = o;
    }
  }
}

struct CandidateEmitArgs {
  const float4 *world;
  const uint32_t *node;
  const uint4 *ident;
  const float4 *sphere;
  const uint32_t *cflags;
  uint4 *rows;  // 9 x uint4 per candidate: VkrWorldDrawCandidate (144 B)
  uint64_t n;
};

// application_emit_world_source, application.h:617-641: one 144-byte
// VkrWorldDrawCandidate per source row; 9 sixteen-byte pieces written by 9 of
// every 16 lanes... kept simple: 16 lanes per row, pieces 0-8 active.
__global__ __launch_bounds__(256) void k_emit_candidates(CandidateEmitArgs a) {
  const uint64_t row = (uint64_t)blockIdx.x * 16u + (threadIdx.x >> 4);
  const uint32_t piece = threadIdx.x & 15u;
  if (row >= a.n || piece >= 9u) return;
  const uint4 id0 = a.ident[row * 2u];
  const uint4 id1 = a.ident[row * 2u + 1u];
  uint4 o;
  if (piece == 0u) {
    o = id0;  // mesh, geometry
  } else if (piece == 1u) {
    o = make_uint4(id1.x, id1.y, id1.z, 0u);  // submesh_index, material, pad
  } else if (piece < 6u) {
    const uint64_t nd = a.node ? a.node[row] : row;
    const float4 c = a.world[nd * 4u + (piece - 2u)];
    o = make_uint4(__float_as_uint(c.x), __float_as_uint(c.y), __float_as_uint(c.z),
                   __float_as_uint(c.w));
  } else if (piece == 6u) {
    o = make_uint4(encode_scene_object_id(id1.w), 0u, 0u, 0u);
  } else if (piece == 7u) {
    const float4 s = a.sphere[row];
    o = make_uint4(__float_as_uint(s.x), __float_as_uint(s.y), __float_as_uint(s.z),
                   __float_as_uint(s.w));
  } else {
    const uint32_t cf = a.cflags[row];
    o = make_uint4((cf >> CF_BUCKET_SHIFT) & 0xfu, cf & CF_REF_MASK, 0u, 0u);
  }
  a.rows[row * 9u + piece] = o;
}

}  // namespace vkr
