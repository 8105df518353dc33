// k_cull.hpp - multi-view sphere frustum cull with deterministic, order
// preserving compaction for gfx950.
//
// Semantics: the camera + shadow-cascade classify / prefix / encode passes
// (lib/src/renderer/shaders/vulkan/slang/world/deferred.slang:225-238,305-420,
// HZB branch excluded) evaluated with the CPU helpers' arithmetic
// (lib/src/renderer/vkr_visibility.c:37-50, lib/src/math/vkr_frustum.c:68-84),
// plus the ordinary-blend camera cull, depth key and counters of
// application_build_world_payload (lib/src/application.h:573-587,730-806).
//
// Three launches:
//   classify    one thread per candidate; the world matrix (64 B), local sphere
//               (16 B) and flags (4 B) are read once for ALL views (planes are
//               wave-uniform scalar loads). Writes one visibility bit per view,
//               the blend-row visibility byte and sort key, and adds per-tile
//               survivor counts per (view, bucket) row. It exists stand-alone
//               (k_classify) and fused behind the transform of the same node
//               (k_fused.hpp).
//   k_scan      exclusive scan of the tile counts of every row (one block per
//               row) and the row totals.
//   k_scatter   4-bucket prefix of vk_gpu_draw_prefix per view from the row
//               totals; re-reads the bit masks (not the matrices) and writes candidate
//               indices at tile base + wave-ballot rank. Order inside a bucket
//               is ascending candidate index, one of the orders the
//               reference's atomics can produce.
#pragma once
#include "device_math.hpp"

namespace vkr {

constexpr int CULL_BLOCK = 256;
constexpr int CULL_ROUNDS = 4;
constexpr int CULL_TILE = CULL_BLOCK * CULL_ROUNDS;  // candidates per count tile
constexpr int CULL_MAX_VIEWS = 64;
constexpr int CULL_MAX_ROWS = CULL_MAX_VIEWS * 4 + 1;  // (view,bucket) rows + transparent row
constexpr int MAT_LDS_STRIDE = 5;  // float4 per staged matrix (80 B: conflict-free b128 reads)

// prepared candidate word
constexpr uint32_t CF_BOUNDS_VALID = 0x1u;    // VKR_WORLD_DRAW_CANDIDATE_BOUNDS_VALID
constexpr uint32_t CF_CAMERA_OPAQUE = 0x2u;   // VKR_WORLD_DRAW_CANDIDATE_CAMERA_OPAQUE
constexpr uint32_t CF_SHADOW_CASTER = 0x4u;   // VKR_WORLD_DRAW_CANDIDATE_SHADOW_CASTER
constexpr uint32_t CF_REF_MASK = 0x7u;
constexpr uint32_t CF_BUCKET_SHIFT = 4u;      // bits 4..7 state bucket
constexpr uint32_t CF_BLEND = 0x100u;         // world_transparent
constexpr uint32_t CF_TRANSMISSIVE = 0x200u;

// device counters (uint32)
enum : int {
  CNT_NODES_UPDATED = 0,
  CNT_WITHOUT_BOUNDS = 1,
  CNT_CULLED_CAMERA = 2,
  CNT_CAMERA_OPAQUE = 3,
  CNT_TRANSMISSION = 4,
  CNT_TRANSPARENT = 5,        // clamped to capacity
  CNT_TRANSPARENT_TOTAL = 6,  // before the clamp
  CNT_COUNT = 8
};
// Counters that every block adds to are kept in STAT_SHARDS copies (block b uses
// copy b % STAT_SHARDS) and summed on the host: thousands of device-scope
// atomics on one word serialise at ~90 per microsecond.
constexpr uint32_t STAT_SHARDS = 64;
constexpr uint32_t TIE_SHARDS = 16;
__device__ __forceinline__ uint32_t stat_shard() { return (blockIdx.x % STAT_SHARDS) * CNT_COUNT; }

struct CompactionState {  // VkrGpuDrawCompactionState, vkr_gpu_abi.h:95-103
  uint32_t execution_ranges[4][2];
  uint32_t bucket_counts[4];
  uint32_t bucket_cursors[4];
  uint32_t visible_count;
  uint32_t overflow_count;
  uint32_t resolve_invalid_count;
  uint32_t occlusion_culled_count;
};
static_assert(sizeof(CompactionState) == 80, "VkrGpuDrawCompactionState is 80 bytes");

// ---------------------------------------------------------------------------
// Candidate preparation (static per scene): application_emit_world_source's
// flags, state bucket and local sphere (application.h:620-640;
// vkr_visibility.h:18-39).
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_prepare_candidates(const uint32_t *__restrict__ matbits,
                                                            const float4 *__restrict__ bounds,
                                                            float4 *__restrict__ sphere,
                                                            uint32_t *__restrict__ cflags,
                                                            uint64_t n) {
  const uint64_t i = (uint64_t)blockIdx.x * 256u + threadIdx.x;
  if (i >= n) return;
  const uint32_t m = matbits[i];
  const uint32_t alpha = m & 0x3u;
  const bool blend = alpha == 2u;
  const bool cutout = alpha == 1u;
  const bool transmissive = (m & 0x4u) != 0;
  const bool double_sided = (m & 0x8u) != 0;
  const bool bounds_valid = (m & 0x10u) != 0;
  const uint32_t bucket = cutout ? (double_sided ? 3u : 2u) : (double_sided ? 1u : 0u);
  uint32_t f = (bounds_valid ? CF_BOUNDS_VALID : 0u) |
               ((!transmissive && !blend) ? CF_CAMERA_OPAQUE : 0u) | CF_SHADOW_CASTER;
  f |= bucket << CF_BUCKET_SHIFT;
  if (blend) f |= CF_BLEND;
  if (transmissive) f |= CF_TRANSMISSIVE;
  const float4 c = bounds[i * 3u + 0];
  const float4 mn = bounds[i * 3u + 1];
  const float4 mx = bounds[i * 3u + 2];
  const float hx = (mx.x - mn.x) * 0.5f;
  const float hy = (mx.y - mn.y) * 0.5f;
  const float hz = (mx.z - mn.z) * 0.5f;
  sphere[i] = make_float4(c.x, c.y, c.z, length3(hx, hy, hz));
  cflags[i] = f;
}

struct CullArgs {
  const float4 *world;     // node matrices, 4 float4 each
  const uint32_t *node;    // candidate -> node, null = identity
  const float4 *sphere;    // local sphere per candidate
  const uint32_t *cflags;  // prepared word per candidate
  const float4 *planes;    // views x 6
  const float *tie_eps;    // views x 6: eps * max(1,|d|)
  const uint32_t *required;
  const float4 *view;      // camera view matrix columns
  uint32_t *masks;         // [word][mask_stride]
  uint8_t *tvis;           // transparent_visible
  uint64_t *tkey;          // sort key of visible blend rows (sparse, per candidate)
  float4 *world_bounds;    // optional (cx,cy,cz,r) per candidate
  uint32_t *tile_counts;   // [row][tiles_stride], zeroed before the frame
  uint32_t *counters;        // [STAT_SHARDS][CNT_COUNT]
  unsigned long long *ties;  // [TIE_SHARDS][CULL_MAX_VIEWS]
  uint64_t n;
  uint64_t mask_stride;
  uint64_t tiles_stride;
  uint64_t index_base;
  uint32_t views;
};

__device__ __forceinline__ uint32_t lane_id() { return threadIdx.x & 63u; }

// LDS accumulators of one classify block.
struct CullShared {
  uint32_t cnt[CULL_MAX_ROWS];
  uint32_t stat[4];
  uint32_t ties[CULL_MAX_VIEWS];
};

__device__ __forceinline__ void cull_shared_clear(CullShared &s, uint32_t views) {
  const uint32_t t = threadIdx.x;
  for (uint32_t r = t; r < views * 4u + 1u; r += blockDim.x) s.cnt[r] = 0u;
  if (t < 4) s.stat[t] = 0u;
  for (uint32_t r = t; r < views; r += blockDim.x) s.ties[r] = 0u;
}

// Classifies one candidate against every view. Must be called by all threads of
// the wave (it votes). `valid` lanes own candidate i with model matrix m.
template <bool TIES, bool BOUNDS_OUT>
__device__ __forceinline__ void classify_candidate(const CullArgs &a, CullShared &s, uint64_t i,
                                                   bool valid, const Mat4 &m, float4 ls,
                                                   uint32_t cf) {
  const bool leader = lane_id() == 0;
  // world sphere: vkr_visibility.c:40-49 / deferred.slang:325-333
  const float3 c = mat4_mul_point(m, ls.x, ls.y, ls.z);
  const float radius = ls.w * max_basis_length(m);
  if (BOUNDS_OUT && valid) a.world_bounds[i] = make_float4(c.x, c.y, c.z, radius);

  const uint32_t bucket = (cf >> CF_BUCKET_SHIFT) & 0xfu;
  const bool bounds_valid = (cf & CF_BOUNDS_VALID) != 0;

  // ordinary blend rows: application.h:790-803 (camera frustum = view 0 planes)
  const bool blend_row = valid && (cf & CF_BLEND) && !(cf & CF_TRANSMISSIVE);
  bool tv = false;
  const uniform_ptr<float> planes = as_uniform(a.planes);
  const uniform_ptr<float> tie_eps = as_uniform(a.tie_eps);
  const uniform_ptr<uint32_t> required = as_uniform(a.required);
  if (blend_row) tv = bounds_valid ? sphere_visible(planes, c.x, c.y, c.z, radius) : true;
  if (valid) a.tvis[i] = tv ? 1u : 0u;
  if (tv) {
    // application_transparent_depth + application_pack_transparent_sort_key,
    // application.h:573-587. The world centre is the one computed above; only
    // the z row of the view matrix takes part.
    const uniform_ptr<float> vm = as_uniform(a.view);
    Mat4 view;
    view.c[0] = make_float4(0.f, 0.f, vm[2], 0.f);
    view.c[1] = make_float4(0.f, 0.f, vm[6], 0.f);
    view.c[2] = make_float4(0.f, 0.f, vm[10], 0.f);
    view.c[3] = make_float4(0.f, 0.f, vm[14], 0.f);
    const float vz = mat4_mul_vec4_z(view, c.x, c.y, c.z, 1.0f);
    float depth = -vz;
    depth = depth > 0.0f ? depth : 0.0f;
    const uint32_t tie_breaker = (uint32_t)(a.index_base + i) + 1u;
    a.tkey[i] = ((uint64_t)__float_as_uint(depth) << 32) | (uint64_t)tie_breaker;
  }
  {
    const unsigned long long mt = __ballot(tv);
    const unsigned long long mc = __ballot(blend_row && !tv);
    const unsigned long long mb = __ballot(valid && !bounds_valid);
    const unsigned long long mo = __ballot(valid && (cf & CF_CAMERA_OPAQUE));
    const unsigned long long mx = __ballot(valid && (cf & CF_TRANSMISSIVE));
    if (leader) {
      if (mt) atomicAdd(&s.cnt[a.views * 4u], (uint32_t)__popcll(mt));
      if (mb) atomicAdd(&s.stat[0], (uint32_t)__popcll(mb));
      if (mc) atomicAdd(&s.stat[1], (uint32_t)__popcll(mc));
      if (mo) atomicAdd(&s.stat[2], (uint32_t)__popcll(mo));
      if (mx) atomicAdd(&s.stat[3], (uint32_t)__popcll(mx));
    }
  }

  // views: deferred.slang:316-351
  const uint32_t words = (a.views + 31u) >> 5;
  unsigned long long in_bucket[4];  // lanes of the wave per state bucket (wave-uniform)
#pragma unroll
  for (uint32_t q = 0; q < 4u; ++q) in_bucket[q] = __ballot(bucket == q);
  for (uint32_t w = 0; w < words; ++w) {
    uint32_t bits = 0u;
    const uint32_t vend = min(a.views, (w + 1u) * 32u);
    for (uint32_t v = w * 32u; v < vend; ++v) {
      const uint32_t req = required[v];
      bool pass = valid && bucket < 4u && (cf & req) == req;
      bool tie = false;
      if (pass && bounds_valid) {
        const uniform_ptr<float> vp = planes + v * 24u;
        pass = sphere_visible(vp, c.x, c.y, c.z, radius);
        if (TIES) {
#pragma unroll
          for (int p = 0; p < 6; ++p) {
            const float dist =
                dot3(vp[p * 4 + 0], vp[p * 4 + 1], vp[p * 4 + 2], c.x, c.y, c.z) + vp[p * 4 + 3];
            tie = tie || (fabsf(dist + radius) <= tie_eps[v * 6u + p]);
          }
        }
      }
      if (TIES) {
        const unsigned long long mtie = __ballot(tie);
        if (leader && mtie) atomicAdd(&s.ties[v], (uint32_t)__popcll(mtie));
      }
      if (pass) bits |= 1u << (v & 31u);
      const unsigned long long mp = __ballot(pass);
      if (leader && mp) {
#pragma unroll
        for (uint32_t q = 0; q < 4u; ++q) {
          const uint32_t nq = (uint32_t)__popcll(mp & in_bucket[q]);
          if (nq) atomicAdd(&s.cnt[v * 4u + q], nq);
        }
      }
    }
    if (valid) a.masks[(uint64_t)w * a.mask_stride + i] = bits;
  }
}

// Adds the block's accumulators to the global tile counts / counters. All
// candidates of the block must lie in count tile `tile`.
template <bool TIES>
__device__ __forceinline__ void cull_shared_flush(const CullArgs &a, CullShared &s,
                                                  uint64_t tile) {
  const uint32_t t = threadIdx.x;
  const uint32_t rows = a.views * 4u + 1u;
  __syncthreads();
#ifndef VKR_EXP_NO_TILE_ATOMICS
  for (uint32_t r = t; r < rows; r += blockDim.x)
    if (s.cnt[r]) atomicAdd(&a.tile_counts[(uint64_t)r * a.tiles_stride + tile], s.cnt[r]);
#endif
  if (t == 0) {
    uint32_t *shard = a.counters + stat_shard();
    if (s.stat[0]) atomicAdd(&shard[CNT_WITHOUT_BOUNDS], s.stat[0]);
    if (s.stat[1]) atomicAdd(&shard[CNT_CULLED_CAMERA], s.stat[1]);
    if (s.stat[2]) atomicAdd(&shard[CNT_CAMERA_OPAQUE], s.stat[2]);
    if (s.stat[3]) atomicAdd(&shard[CNT_TRANSMISSION], s.stat[3]);
  }
  if (TIES)
    for (uint32_t r = t; r < a.views; r += blockDim.x)
      if (s.ties[r])
        atomicAdd(&a.ties[(blockIdx.x % TIE_SHARDS) * CULL_MAX_VIEWS + r],
                  (unsigned long long)s.ties[r]);
}

// Stand-alone classify: one block per 256 candidates. With the identity
// candidate->node mapping the matrices arrive as coalesced 16-byte loads staged
// through LDS (thread t then reads its own four columns); otherwise each thread
// gathers its node's matrix.
template <bool IDENTITY, bool TIES, bool BOUNDS_OUT>
__global__ __launch_bounds__(CULL_BLOCK) void k_classify(CullArgs a) {
  __shared__ float4 s_mat[IDENTITY ? CULL_BLOCK * MAT_LDS_STRIDE : 1];
  __shared__ CullShared s;
  const uint32_t t = threadIdx.x;
  cull_shared_clear(s, a.views);

  const uint64_t base = (uint64_t)blockIdx.x * CULL_BLOCK;
  const uint64_t i = base + t;
  const bool valid = i < a.n;
  float4 ls = make_float4(0.f, 0.f, 0.f, 0.f);
  uint32_t cf = 0xf0u;  // invalid bucket: fails every gate
  if (valid) {
    ls = a.sphere[i];
    cf = a.cflags[i];
  }
  Mat4 m;
  if (IDENTITY) {
    const uint64_t total4 = a.n * 4u;
    float4 v[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const uint64_t g = base * 4u + t + 256u * (uint32_t)k;
      v[k] = g < total4 ? a.world[g] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const uint32_t l = t + 256u * (uint32_t)k;
      s_mat[(l >> 2) * MAT_LDS_STRIDE + (l & 3u)] = v[k];
    }
    __syncthreads();
    const float4 *p = s_mat + t * MAT_LDS_STRIDE;
    m.c[0] = p[0];
    m.c[1] = p[1];
    m.c[2] = p[2];
    m.c[3] = p[3];
  } else {
    __syncthreads();  // s cleared
    if (valid) {
      const float4 *p = a.world + (uint64_t)a.node[i] * 4u;
      m.c[0] = p[0];
      m.c[1] = p[1];
      m.c[2] = p[2];
      m.c[3] = p[3];
    } else {
      m.c[0] = m.c[1] = m.c[2] = m.c[3] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
  }
  classify_candidate<TIES, BOUNDS_OUT>(a, s, i, valid, m, ls, cf);
  cull_shared_flush<TIES>(a, s, base / CULL_TILE);
}

// ---------------------------------------------------------------------------
// Block-wide exclusive scan helpers (1024 threads, 16 waves).
// ---------------------------------------------------------------------------
constexpr int SCAN_BLOCK = 1024;

__device__ __forceinline__ uint32_t wave_inclusive_scan(uint32_t v) {
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const uint32_t u = __shfl_up(v, d, 64);
    if ((int)lane_id() >= d) v += u;
  }
  return v;
}

// Returns the exclusive prefix of `v` over the block and the block total.
__device__ __forceinline__ uint32_t block_exclusive_scan(uint32_t v, uint32_t *s_wave,
                                                         uint32_t *total) {
  const uint32_t w = threadIdx.x >> 6;
  const uint32_t nw = blockDim.x >> 6;
  const uint32_t inc = wave_inclusive_scan(v);
  __syncthreads();  // s_wave reuse
  if (lane_id() == 63) s_wave[w] = inc;
  __syncthreads();
  uint32_t wave_base = 0, sum = 0;
  for (uint32_t k = 0; k < nw; ++k) {
    const uint32_t x = s_wave[k];
    if (k < w) wave_base += x;
    sum += x;
  }
  *total = sum;
  return wave_base + inc - v;
}

// Four independent scans at once (the four bucket rows of a view).
__device__ __forceinline__ uint4 block_exclusive_scan4(uint4 v, uint4 *s_wave4, uint4 *total) {
  const uint32_t w = threadIdx.x >> 6;
  const uint32_t nw = blockDim.x >> 6;
  uint4 inc = v;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const uint32_t ux = __shfl_up(inc.x, d, 64);
    const uint32_t uy = __shfl_up(inc.y, d, 64);
    const uint32_t uz = __shfl_up(inc.z, d, 64);
    const uint32_t uw = __shfl_up(inc.w, d, 64);
    if ((int)lane_id() >= d) {
      inc.x += ux;
      inc.y += uy;
      inc.z += uz;
      inc.w += uw;
    }
  }
  __syncthreads();
  if (lane_id() == 63) s_wave4[w] = inc;
  __syncthreads();
  uint4 base = make_uint4(0, 0, 0, 0), sum = make_uint4(0, 0, 0, 0);
  for (uint32_t k = 0; k < nw; ++k) {
    const uint4 x = s_wave4[k];
    if (k < w) {
      base.x += x.x;
      base.y += x.y;
      base.z += x.z;
      base.w += x.w;
    }
    sum.x += x.x;
    sum.y += x.y;
    sum.z += x.z;
    sum.w += x.w;
  }
  *total = sum;
  return make_uint4(base.x + inc.x - v.x, base.y + inc.y - v.y, base.z + inc.z - v.z,
                    base.w + inc.w - v.w);
}

struct ScanArgs {
  uint32_t *tile_counts;  // [row][tiles_stride], scanned in place
  uint32_t *row_totals;   // [rows]: survivors per (view, bucket) row and of the transparent row
  uint32_t *counters;
  uint64_t tiles;  // tiles in use this frame
  uint64_t tiles_stride;
  uint32_t views;
  uint64_t transparent_capacity;
  uint32_t *host_hint;   // pinned host word: last frame's transparent count (may be null)
  uint32_t *clear;       // the NEXT frame's counter block, zeroed here (may be null)
  uint32_t clear_words;
};

__device__ __forceinline__ uint32_t sum4(uint4 v) { return v.x + v.y + v.z + v.w; }

// Exclusive scan of the tile counts, one 1024-thread block per row. A wave owns a
// contiguous run of the row and reads it 64 x 16 bytes at a time (coalesced);
// sweep 1 only totals each wave's run, one barrier combines the 16 totals, and
// sweep 2 re-reads the run (from cache), scans it 64 uint4 at a time with wave
// shuffles and writes the exclusive prefix in place: two memory round trips for
// any tile count. Rows are padded with zeros to whole uint4s.
__global__ __launch_bounds__(SCAN_BLOCK) void k_scan(ScanArgs a) {
  __shared__ uint32_t s_wave[SCAN_BLOCK / 64];
  const uint32_t r = blockIdx.x;
  uint4 *row = reinterpret_cast<uint4 *>(a.tile_counts + (uint64_t)r * a.tiles_stride);
  const uint64_t quads = (a.tiles + 3u) / 4u;
  const uint32_t w = threadIdx.x >> 6;
  const uint32_t lane = lane_id();
  const uint64_t wave_quads = ((quads + SCAN_BLOCK / 64 - 1) / (SCAN_BLOCK / 64) + 63u) & ~(uint64_t)63u;
  const uint64_t qb = min((uint64_t)w * wave_quads, quads);
  const uint64_t qe = min(qb + wave_quads, quads);

  uint32_t mine = 0u;
  for (uint64_t q = qb + lane; q < qe; q += 64u) mine += sum4(row[q]);
  const uint32_t wave_total = (uint32_t)__shfl((int)wave_inclusive_scan(mine), 63, 64);
  if (lane == 0) s_wave[w] = wave_total;
  __syncthreads();
  uint32_t carry = 0u, total = 0u;
  for (uint32_t k = 0; k < SCAN_BLOCK / 64; ++k) {
    const uint32_t x = s_wave[k];
    if (k < w) carry += x;
    total += x;
  }
  for (uint64_t q0 = qb; q0 < qe; q0 += 64u) {  // wave-uniform trip count
    const uint64_t q = q0 + lane;
    const bool valid = q < qe;
    const uint4 v = valid ? row[q] : make_uint4(0u, 0u, 0u, 0u);
    const uint32_t s = sum4(v);
    const uint32_t incl = wave_inclusive_scan(s);
    uint4 o;
    o.x = carry + incl - s;
    o.y = o.x + v.x;
    o.z = o.y + v.y;
    o.w = o.z + v.z;
    // the padding of the last uint4 stays zero: it may be a real tile next frame
    if (q * 4u + 1u >= a.tiles) o.y = 0u;
    if (q * 4u + 2u >= a.tiles) o.z = 0u;
    if (q * 4u + 3u >= a.tiles) o.w = 0u;
    if (valid) row[q] = o;
    carry += (uint32_t)__shfl((int)incl, 63, 64);
  }
  if (r == 0 && a.clear != nullptr)
    for (uint32_t k = threadIdx.x; k < a.clear_words; k += SCAN_BLOCK) a.clear[k] = 0u;
  if (threadIdx.x == 0) {
    a.row_totals[r] = total;
    if (r == a.views * 4u) {  // the transparent row
      const uint64_t cap = a.transparent_capacity;
      a.counters[CNT_TRANSPARENT_TOTAL] = total;
      a.counters[CNT_TRANSPARENT] = (uint64_t)total < cap ? total : (uint32_t)cap;
      // lets the host pick the sort path of later frames without waiting
      if (a.host_hint) *reinterpret_cast<volatile uint32_t *>(a.host_hint) = total + 1u;
    }
  }
}

// vk_gpu_draw_prefix (deferred.slang:354-377): the four bucket totals of a view
// become execution ranges clamped to the partition capacity.
__device__ __forceinline__ CompactionState prefix_state(const uint32_t *totals4,
                                                        uint32_t partition_capacity) {
  CompactionState st;
  uint32_t cursor = 0, overflow = 0;
  const uint32_t cap = partition_capacity ? partition_capacity : 0xffffffffu;
  for (uint32_t b = 0; b < 4u; ++b) {
    const uint32_t count = totals4[b];
    const uint32_t written = count < cap ? count : cap;
    st.execution_ranges[b][0] = cursor;
    st.execution_ranges[b][1] = written;
    st.bucket_counts[b] = count;
    // every surviving thread of vk_gpu_draw_encode bumps the cursor once
    st.bucket_cursors[b] = count;
    cursor += written;
    overflow += count - written;
  }
  st.visible_count = cursor;
  st.overflow_count = overflow;
  st.resolve_invalid_count = 0;
  st.occlusion_culled_count = 0;
  return st;
}

struct ScatterArgs {
  const uint32_t *masks;
  const uint32_t *cflags;
  const uint8_t *tvis;
  const uint64_t *tkey;
  uint32_t *tile_offsets;        // scanned tile counts; zeroed again once consumed
  const uint32_t *row_totals;    // from k_scan
  CompactionState *states;       // written by block 0
  uint32_t *visible;  // [view][visible_stride]
  uint64_t *keys;     // compacted sort keys of the visible blend rows
  uint64_t n;
  uint64_t mask_stride;
  uint64_t tiles_stride;
  uint64_t visible_stride;
  uint64_t transparent_capacity;
  uint64_t index_base;
  uint32_t views;
  uint32_t partition_capacity;
};

__global__ __launch_bounds__(CULL_BLOCK) void k_scatter(ScatterArgs a) {
  // survivors of (round, wave) per row, turned into bucket-local start indices
  __shared__ uint32_t s_cnt[CULL_ROUNDS * 4][CULL_MAX_ROWS];
  __shared__ uint32_t s_range[CULL_MAX_ROWS];  // execution_ranges[b].x of the row's view

  const uint32_t t = threadIdx.x;
  const uint32_t wave = t >> 6;
  const uint32_t lane = lane_id();
  const unsigned long long lt_mask = (1ull << lane) - 1ull;
  const uint32_t rows = a.views * 4u + 1u;
  const uint64_t tile = blockIdx.x;
  const uint32_t cap = a.partition_capacity ? a.partition_capacity : 0xffffffffu;
  const uint32_t words = (a.views + 31u) >> 5;

  // All inputs are requested up front with unconditional loads from clamped
  // indices (a select on a loaded value would become a branch around the load
  // that waits for it); out-of-range lanes are blanked with integer masks.
  const uint64_t last = a.n - 1u;  // n > 0: the grid has a block per tile
  uint32_t cf_r[CULL_ROUNDS];
  bool tv_r[CULL_ROUNDS];
  uint32_t bits0_r[CULL_ROUNDS];
  uint32_t tv_raw[CULL_ROUNDS];
#pragma unroll
  for (int round = 0; round < CULL_ROUNDS; ++round) {
    const uint64_t i = min(tile * CULL_TILE + (uint64_t)round * CULL_BLOCK + t, last);
    cf_r[round] = a.cflags[i];
    tv_raw[round] = a.tvis[i];
    bits0_r[round] = a.masks[i];
  }
  // Sixteen lanes per row turn the row's (round, wave) counts into start
  // indices; the tile's scanned offset and the totals of the buckets before the
  // row's bucket are requested now for the first 16 rows (more rows: later).
  constexpr uint32_t SLOTS = CULL_ROUNDS * 4;
  const uint32_t slot = t % SLOTS;
  const uint32_t first_row = min(t / SLOTS, rows - 1u);
  const uint32_t first_offset = a.tile_offsets[(uint64_t)first_row * a.tiles_stride + tile];
  uint32_t first_before[3];
#pragma unroll
  for (uint32_t b = 0; b < 3u; ++b)
    first_before[b] = min(a.row_totals[min((first_row & ~3u) + b, rows - 1u)], cap);
#pragma unroll
  for (int round = 0; round < CULL_ROUNDS; ++round) {
    const uint64_t i = tile * CULL_TILE + (uint64_t)round * CULL_BLOCK + t;
    const uint32_t vm = i < a.n ? 0xffffffffu : 0u;
    cf_r[round] = (cf_r[round] & vm) | (0xf0u & ~vm);
    tv_r[round] = (tv_raw[round] & vm) != 0u;
    bits0_r[round] &= vm;
  }

  // phase 1: per-(round, wave) survivor counts of every row
#pragma unroll
  for (int round = 0; round < CULL_ROUNDS; ++round) {
    const uint64_t i = tile * CULL_TILE + (uint64_t)round * CULL_BLOCK + t;
    const bool valid = i < a.n;
    const uint32_t bucket = (cf_r[round] >> CF_BUCKET_SHIFT) & 0xfu;
    uint32_t *cnt = s_cnt[round * 4 + (int)wave];
    unsigned long long in_bucket[4];  // lanes of the wave per state bucket
#pragma unroll
    for (uint32_t q = 0; q < 4u; ++q) in_bucket[q] = __ballot(bucket == q);
    for (uint32_t w = 0; w < words; ++w) {
      const uint32_t bits =
          w == 0 ? bits0_r[round] : (valid ? a.masks[(uint64_t)w * a.mask_stride + i] : 0u);
      const uint32_t vend = min(a.views, (w + 1u) * 32u);
      for (uint32_t v = w * 32u; v < vend; ++v) {
        const unsigned long long mp = __ballot((bits >> (v & 31u)) & 1u);
        if (lane == 0) {
#pragma unroll
          for (uint32_t q = 0; q < 4u; ++q)
            cnt[v * 4u + q] = (uint32_t)__popcll(mp & in_bucket[q]);
        }
      }
    }
    const unsigned long long mt = __ballot(tv_r[round]);
    if (lane == 0) cnt[a.views * 4u] = (uint32_t)__popcll(mt);
  }
  __syncthreads();
  // exclusive prefix over the 16 (round, wave) slots in candidate order, on top
  // of the tile's scanned offset: a 16-lane shuffle scan per row
  for (uint32_t r0 = 0; r0 < rows; r0 += CULL_BLOCK / SLOTS) {
    const uint32_t r = r0 + t / SLOTS;
    const bool live = r < rows;
    const uint32_t rc = live ? r : rows - 1u;
    const uint32_t c = live ? s_cnt[slot][rc] : 0u;
    uint32_t incl = c;
#pragma unroll
    for (uint32_t d = 1; d < SLOTS; d <<= 1) {
      const uint32_t u = __shfl_up(incl, d, SLOTS);
      if (slot >= d) incl += u;
    }
    uint32_t offset = first_offset;
    uint32_t before[3] = {first_before[0], first_before[1], first_before[2]};
    if (r0 != 0u) {
      offset = a.tile_offsets[(uint64_t)rc * a.tiles_stride + tile];
#pragma unroll
      for (uint32_t b = 0; b < 3u; ++b) before[b] = min(a.row_totals[min((rc & ~3u) + b, rows - 1u)], cap);
    }
    if (live) {
      s_cnt[slot][r] = offset + incl - c;
      if (slot == 0u) {
        a.tile_offsets[(uint64_t)r * a.tiles_stride + tile] = 0u;  // ready for the next frame
        // start of the row's bucket inside its view
        uint32_t range = 0u;
        if (r < a.views * 4u) {
#pragma unroll
          for (uint32_t b = 0; b < 3u; ++b) range += b < (r & 3u) ? before[b] : 0u;
        }
        s_range[r] = range;
      }
    }
  }
  if (blockIdx.x == 0)
    for (uint32_t v = t; v < a.views; v += CULL_BLOCK)
      a.states[v] = prefix_state(a.row_totals + v * 4u, a.partition_capacity);
  __syncthreads();

  // phase 2: positions
#pragma unroll
  for (int round = 0; round < CULL_ROUNDS; ++round) {
    const uint64_t i = tile * CULL_TILE + (uint64_t)round * CULL_BLOCK + t;
    const bool valid = i < a.n;
    const uint32_t bucket = (cf_r[round] >> CF_BUCKET_SHIFT) & 0xfu;
    const uint32_t *start = s_cnt[round * 4 + (int)wave];
    unsigned long long below_same_bucket = 0ull;  // lower lanes holding this lane's bucket
#pragma unroll
    for (uint32_t q = 0; q < 4u; ++q) {
      const unsigned long long mb = __ballot(bucket == q);
      if (bucket == q) below_same_bucket = mb & lt_mask;
    }
    for (uint32_t w = 0; w < words; ++w) {
      const uint32_t bits =
          w == 0 ? bits0_r[round] : (valid ? a.masks[(uint64_t)w * a.mask_stride + i] : 0u);
      const uint32_t vend = min(a.views, (w + 1u) * 32u);
      for (uint32_t v = w * 32u; v < vend; ++v) {
        const bool pass = (bits >> (v & 31u)) & 1u;
        const uint32_t row = v * 4u + (bucket & 3u);
        // survivors of the same bucket in the lanes below this one
        const unsigned long long mp = __ballot(pass);
        const uint32_t rank = (uint32_t)__popcll(mp & below_same_bucket);
        if (pass) {
          const uint32_t local = start[row] + rank;
          if (local < cap) {
            const uint64_t slot = (uint64_t)s_range[row] + local;
            if (slot < a.visible_stride)
              a.visible[(uint64_t)v * a.visible_stride + slot] = (uint32_t)(a.index_base + i);
          }
        }
      }
    }
    const bool tv = tv_r[round];
    const unsigned long long mt = __ballot(tv);
    if (tv) {
      const uint64_t pos = (uint64_t)start[a.views * 4u] + (uint32_t)__popcll(mt & lt_mask);
      if (pos < a.transparent_capacity) a.keys[pos] = a.tkey[i];
    }
  }
}

}  // namespace vkr
