// k_epoch.hpp - the candidate epoch hash on the device.
//
// vkr_vk_candidate_epoch (lib/src/renderer/vulkan/vkr_vulkan_draws.c:9-44) is a
// 64-bit FNV-1a over eleven fields (116 bytes) of every candidate that carries
// CAMERA_OPAQUE, in candidate order, closed with the 4 bytes of the occluder
// count. FNV-1a is written as a byte-serial chain,  h' = (h ^ b) * P  mod 2^64,
// but it splits into parallel parts:
//
//  * XOR with a byte only touches the low 8 bits, and the low 8 bits of h' depend
//    only on the low 8 bits of h (P is odd, carries go up). With l = h & 0xff:
//      h ^ b = h + e,  e = (l ^ b) - l            (a small signed integer)
//      h'    = h * P + e * P                      (affine in h once e is known)
//      l'    = ((l ^ b) * 0xB3) & 0xff            (0xB3 = P & 0xff)
//  * so over m bytes  h_m = h_0 * P^m + C,  C = ((..(e_0 P + e_1) P ..) + e_{m-1}) P,
//    and C of a candidate is known once the low byte entering it is known;
//    candidates then combine like affine maps: (A1, C1) then (A2, C2) =
//    (A1 A2, C1 A2 + C2) - an ordered tree reduction.
//  * the low byte itself is a 256-state automaton. It splits once more: the low
//    nibble n' = ((n ^ b_lo) * 3) & 15 is closed in itself (16 states, kept as a
//    16 x 4-bit map in one 64-bit word and stepped for all 16 states at once
//    with nibble-wise arithmetic), and with the low nibble's path known the
//    high nibble is  H' = (((H ^ b_hi) * 3) + carry) & 15  - 16 states again.
//
// Five passes, every one parallel over candidates (one lane per candidate, one
// wave per 64 candidates):
//   k_epoch_low      low-nibble map of every candidate, composed per wave
//   k_epoch_scan     nibble entering every wave (single block)
//   k_epoch_high     high-nibble map of every candidate given its low-nibble path
//   k_epoch_scan     nibble entering every wave
//   k_epoch_affine   (A, C) of every candidate, reduced per wave
//   k_epoch_final    ordered reduction of the wave pairs, the count bytes
#pragma once
#include "k_cull.hpp"

namespace vkr {

// the reference's starting value (vkr_vulkan_draws.c:21): the standard 64-bit
// basis 14695981039346656037 without its last digit; kept as the reference has it
constexpr uint64_t FNV_OFFSET = 1469598103934665603ull;
constexpr uint64_t FNV_PRIME = 1099511628211ull;
constexpr int EPOCH_WORDS = 29;  // 116 hashed bytes per candidate
constexpr uint64_t NIB_ONES = 0x1111111111111111ull;
constexpr uint64_t NIB_IDENTITY = 0xFEDCBA9876543210ull;  // nibble i holds i

constexpr uint64_t fnv_prime_power(int bytes) {
  uint64_t r = 1;
  for (int i = 0; i < bytes; ++i) r *= FNV_PRIME;
  return r;
}
constexpr uint64_t FNV_PRIME_CANDIDATE = fnv_prime_power(EPOCH_WORDS * 4);

struct EpochArgs {
  const uint4 *ident;  // 2 x uint4 per candidate
  const float4 *world;
  const uint32_t *node;  // null: candidate i uses node i
  const float4 *sphere;
  const uint32_t *cflags;
  uint64_t n;
  uint64_t waves;            // ceil(n / 64)
  uint64_t *maps;            // per candidate: the nibble map of the current pass
  uint64_t *maps_low;        // per candidate: kept low-nibble maps
  uint64_t *wave_maps;       // per wave: composition of its candidates' maps
  uint8_t *enter_low;        // per wave: low nibble of the hash entering it
  uint8_t *enter_high;       // per wave
  uint32_t *wave_occluders;  // per wave
  uint64_t *pairs;           // per wave {A, C}
  uint64_t *result;          // {hash, occluder count}
};

// The hashed words of candidate i in order: mesh.id, mesh.generation,
// geometry.id, geometry.generation, submesh_index, material.id,
// material.generation, instance.model (16), local_bounding_sphere (4),
// state_bucket, flags. Returns whether the candidate takes part.
__device__ __forceinline__ bool epoch_load(const EpochArgs &a, uint64_t i, uint32_t (&w)[EPOCH_WORDS]) {
  const uint4 id0 = a.ident[i * 2u];
  const uint4 id1 = a.ident[i * 2u + 1u];
  w[0] = id0.x, w[1] = id0.y, w[2] = id0.z, w[3] = id0.w;
  w[4] = id1.x, w[5] = id1.y, w[6] = id1.z;
  const uint64_t nd = a.node ? a.node[i] : i;
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    const float4 col = a.world[nd * 4u + c];
    w[7 + c * 4] = __float_as_uint(col.x);
    w[8 + c * 4] = __float_as_uint(col.y);
    w[9 + c * 4] = __float_as_uint(col.z);
    w[10 + c * 4] = __float_as_uint(col.w);
  }
  const float4 s = a.sphere[i];
  w[23] = __float_as_uint(s.x), w[24] = __float_as_uint(s.y);
  w[25] = __float_as_uint(s.z), w[26] = __float_as_uint(s.w);
  const uint32_t cf = a.cflags[i];
  w[27] = (cf >> CF_BUCKET_SHIFT) & 0xfu;
  w[28] = cf & CF_REF_MASK;
  return (cf & CF_CAMERA_OPAQUE) != 0u;
}

// nibble-wise a + b (mod 16 in every nibble, no carry between nibbles)
__device__ __forceinline__ uint64_t nib_add(uint64_t a, uint64_t b) {
  const uint64_t low3 = 0x7777777777777777ull;
  return ((a & low3) + (b & low3)) ^ ((a ^ b) & ~low3);
}
__device__ __forceinline__ uint64_t nib_mul3(uint64_t x) {
  return nib_add(x, (x << 1) & 0xEEEEEEEEEEEEEEEEull);
}
__device__ __forceinline__ uint64_t nib_splat(uint32_t v) {  // v < 16 in every nibble
  const uint32_t r = v * 0x11111111u;
  return ((uint64_t)r << 32) | r;
}
__device__ __forceinline__ uint32_t nib_at(uint64_t map, uint32_t i) {
  return (uint32_t)(map >> (4u * i)) & 0xfu;
}
// first f, then g
__device__ __forceinline__ uint64_t nib_compose(uint64_t f, uint64_t g) {
  uint64_t r = 0;
#pragma unroll
  for (uint32_t i = 0; i < 16u; ++i) r |= (uint64_t)nib_at(g, nib_at(f, i)) << (4u * i);
  return r;
}

__device__ __forceinline__ uint64_t read_lane64(uint64_t v, int lane) {
  const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)v, lane);
  const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(v >> 32), lane);
  return ((uint64_t)hi << 32) | lo;
}

// Composition of the maps of lanes 0..63 in lane order (same value in every
// lane): lanes 0..15 each carry one start state through the 64 maps.
__device__ __forceinline__ uint64_t wave_compose(uint64_t m) {
  const uint32_t lane = lane_id();
  uint32_t v = lane & 15u;
#pragma unroll
  for (int j = 0; j < 64; ++j) v = nib_at(read_lane64(m, j), v);
  uint64_t part = lane < 16u ? (uint64_t)v << (4u * lane) : 0ull;
#pragma unroll
  for (int d = 1; d < 16; d <<= 1) {
    const uint32_t lo = __shfl_xor((uint32_t)part, d, 64);
    const uint32_t hi = __shfl_xor((uint32_t)(part >> 32), d, 64);
    part |= ((uint64_t)hi << 32) | lo;
  }
  return read_lane64(part, 0);
}

// Lane j gets `enter` carried through the maps of lanes 0..j-1.
__device__ __forceinline__ uint32_t wave_enter(uint64_t m, uint32_t enter) {
  const uint32_t lane = lane_id();
  uint32_t v = enter, mine = 0u;
#pragma unroll
  for (int j = 0; j < 64; ++j) {
    if (lane == (uint32_t)j) mine = v;
    v = nib_at(read_lane64(m, j), v);  // uniform
  }
  return mine;
}

// Pass 1: the low-nibble map of every candidate.
__global__ __launch_bounds__(256) void k_epoch_low(EpochArgs a) {
  const uint64_t i = (uint64_t)blockIdx.x * 256u + threadIdx.x;
  const uint64_t wave = i >> 6;
  if (wave >= a.waves) return;  // whole waves only
  uint64_t map = NIB_IDENTITY;
  bool takes_part = false;
  if (i < a.n) {
    uint32_t w[EPOCH_WORDS];
    takes_part = epoch_load(a, i, w);
    if (takes_part) {
#pragma unroll
      for (int k = 0; k < EPOCH_WORDS; ++k)
#pragma unroll
        for (int b = 0; b < 4; ++b) map = nib_mul3(map ^ nib_splat((w[k] >> (8 * b)) & 0xfu));
    }
    a.maps_low[i] = map;
  }
  const uint64_t all = wave_compose(map);
  const uint32_t occluders = (uint32_t)__popcll(__ballot(takes_part));
  if (lane_id() == 0) {
    a.wave_maps[wave] = all;
    a.wave_occluders[wave] = occluders;
  }
}

// The nibble entering every wave: `start` carried through the wave maps in
// order. One block of SCAN_BLOCK threads; each thread owns a run of waves.
__global__ __launch_bounds__(SCAN_BLOCK) void k_epoch_scan(const uint64_t *wave_maps, uint64_t waves,
                                                           uint32_t start, uint8_t *enter) {
  __shared__ uint64_t s_map[2][SCAN_BLOCK];
  const uint32_t t = threadIdx.x;
  const uint64_t per = (waves + SCAN_BLOCK - 1u) / SCAN_BLOCK;
  const uint64_t lo = min((uint64_t)t * per, waves), hi = min(lo + per, waves);
  uint64_t run = NIB_IDENTITY;
  for (uint64_t w = lo; w < hi; ++w) run = nib_compose(run, wave_maps[w]);
  // inclusive scan of the run maps (composition is associative, order kept)
  int cur = 0;
  s_map[0][t] = run;
  __syncthreads();
  for (uint32_t d = 1; d < SCAN_BLOCK; d <<= 1) {
    uint64_t v = s_map[cur][t];
    if (t >= d) v = nib_compose(s_map[cur][t - d], v);
    s_map[cur ^ 1][t] = v;
    cur ^= 1;
    __syncthreads();
  }
  uint32_t v = t == 0 ? start : nib_at(s_map[cur][t - 1u], start);
  for (uint64_t w = lo; w < hi; ++w) {
    enter[w] = (uint8_t)v;
    v = nib_at(wave_maps[w], v);
  }
}

// Pass 2: the high-nibble map of every candidate, given the low nibble that
// enters it (the low nibble's path through the candidate is replayed).
__global__ __launch_bounds__(256) void k_epoch_high(EpochArgs a) {
  const uint64_t i = (uint64_t)blockIdx.x * 256u + threadIdx.x;
  const uint64_t wave = i >> 6;
  if (wave >= a.waves) return;
  const bool inside = i < a.n;
  const uint64_t low_map = inside ? a.maps_low[i] : NIB_IDENTITY;
  uint32_t n = wave_enter(low_map, a.enter_low[wave]);
  uint64_t map = NIB_IDENTITY;
  if (inside) {
    uint32_t w[EPOCH_WORDS];
    if (epoch_load(a, i, w)) {
#pragma unroll
      for (int k = 0; k < EPOCH_WORDS; ++k)
#pragma unroll
        for (int b = 0; b < 4; ++b) {
          const uint32_t byte = (w[k] >> (8 * b)) & 0xffu;
          const uint32_t xl = n ^ (byte & 0xfu);
          const uint32_t carry = ((xl * 0xB3u) >> 4) & 0xfu;
          n = (xl * 3u) & 0xfu;
          map = nib_add(nib_mul3(map ^ nib_splat(byte >> 4)), nib_splat(carry));
        }
    }
    a.maps[i] = map;
  }
  const uint64_t all = wave_compose(map);
  if (lane_id() == 0) a.wave_maps[wave] = all;
}

__device__ __forceinline__ uint64_t shfl_down64(uint64_t v, int d) {
  const uint32_t lo = __shfl_down((uint32_t)v, d, 64);
  const uint32_t hi = __shfl_down((uint32_t)(v >> 32), d, 64);
  return ((uint64_t)hi << 32) | lo;
}

// Pass 3: the affine pair of every candidate, reduced in order per wave.
__global__ __launch_bounds__(256) void k_epoch_affine(EpochArgs a) {
  const uint64_t i = (uint64_t)blockIdx.x * 256u + threadIdx.x;
  const uint64_t wave = i >> 6;
  if (wave >= a.waves) return;
  const bool inside = i < a.n;
  const uint32_t lo = wave_enter(inside ? a.maps_low[i] : NIB_IDENTITY, a.enter_low[wave]);
  const uint32_t hi = wave_enter(inside ? a.maps[i] : NIB_IDENTITY, a.enter_high[wave]);
  uint64_t mul = 1ull, add = 0ull;
  if (inside) {
    uint32_t w[EPOCH_WORDS];
    if (epoch_load(a, i, w)) {
      uint32_t l = lo | (hi << 4);
#pragma unroll
      for (int k = 0; k < EPOCH_WORDS; ++k)
#pragma unroll
        for (int b = 0; b < 4; ++b) {
          const uint32_t x = l ^ ((w[k] >> (8 * b)) & 0xffu);
          add = (add + (uint64_t)(int64_t)((int32_t)x - (int32_t)l)) * FNV_PRIME;
          l = (x * 0xB3u) & 0xffu;
        }
      mul = FNV_PRIME_CANDIDATE;
    }
  }
  const uint32_t lane = lane_id();
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const uint64_t right_mul = shfl_down64(mul, d);
    const uint64_t right_add = shfl_down64(add, d);
    if ((lane & (2u * d - 1u)) == 0u) {  // left operand is the lower lanes
      add = add * right_mul + right_add;
      mul *= right_mul;
    }
  }
  if (lane == 0) {
    a.pairs[wave * 2u] = mul;
    a.pairs[wave * 2u + 1u] = add;
  }
}

// Ordered reduction of the wave pairs and the closing count bytes.
__global__ __launch_bounds__(SCAN_BLOCK) void k_epoch_final(EpochArgs a) {
  __shared__ uint64_t s_mul[SCAN_BLOCK], s_add[SCAN_BLOCK];
  __shared__ uint32_t s_count[SCAN_BLOCK];
  const uint32_t t = threadIdx.x;
  const uint64_t per = (a.waves + SCAN_BLOCK - 1u) / SCAN_BLOCK;
  const uint64_t lo = min((uint64_t)t * per, a.waves), hi = min(lo + per, a.waves);
  uint64_t mul = 1ull, add = 0ull;
  uint32_t count = 0u;
  for (uint64_t w = lo; w < hi; ++w) {
    const uint64_t m = a.pairs[w * 2u], c = a.pairs[w * 2u + 1u];
    add = add * m + c;
    mul *= m;
    count += a.wave_occluders[w];
  }
  s_mul[t] = mul, s_add[t] = add, s_count[t] = count;
  __syncthreads();
  for (uint32_t d = 1; d < SCAN_BLOCK; d <<= 1) {
    if ((t & (2u * d - 1u)) == 0u) {
      s_add[t] = s_add[t] * s_mul[t + d] + s_add[t + d];
      s_mul[t] *= s_mul[t + d];
      s_count[t] += s_count[t + d];
    }
    __syncthreads();
  }
  if (t == 0) {
    uint64_t h = FNV_OFFSET * s_mul[0] + s_add[0];
    const uint32_t occluders = s_count[0];
    for (int b = 0; b < 4; ++b) h = (h ^ ((occluders >> (8 * b)) & 0xffu)) * FNV_PRIME;
    a.result[0] = h;
    a.result[1] = occluders;
  }
}

}  // namespace vkr
