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
//               wave-uniform scalar loads). A block works on one SEGMENT of at
//               most 1024 consecutive candidates. It keeps the per-view verdict
//               bits in LDS, and when the segment is done it packs the
//               survivors of every (view, bucket) row - ascending candidate
//               index, buckets back to back - into the segment's own stretch
//               of a staging list and stores the row counts. Only survivors
//               are written, nothing per candidate except the blend-row
//               visibility byte. It exists stand-alone (k_classify) and fused
//               behind the transform of the same node (k_fused.hpp).
//   k_scan      exclusive scan of the segment counts of every row (one block
//               per row) and the row totals.
//   k_gather    4-bucket prefix of vk_gpu_draw_prefix per view from the row
//               totals; one wave per segment moves the segment's staged runs to
//               their final offsets (traffic proportional to the survivors).
//               Order inside a bucket is ascending candidate index, one of the
//               orders the reference's atomics can produce.
#pragma once
#include "device_math.hpp"

namespace vkr {
constexpr int CULL_BLOCK = 256;
constexpr int CULL_ROUNDS = 4;
constexpr int CULL_TILE = CULL_BLOCK * CULL_ROUNDS;  // candidates per segment (at most)
constexpr int CULL_SLOTS = CULL_ROUNDS * (CULL_BLOCK / 64);  // (round, wave) pairs of a segment
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
constexpr uint32_t CF_UNPUBLISHED = 0x400u;   // omitted by the backend pack: in no GPU view

// device counters (uint32)
enum : int {
  CNT_NODES_UPDATED = 0,
  CNT_WITHOUT_BOUNDS = 1,
  CNT_CULLED_CAMERA = 2,
  CNT_CAMERA_OPAQUE = 3,
  CNT_TRANSMISSION = 4,
  CNT_TRANSPARENT = 5,        // clamped to capacity
  CNT_TRANSPARENT_TOTAL = 6,  // before the clamp
  CNT_OCCLUDED = 7,           // view 0 rows rejected by the depth pyramid
  CNT_COUNT = 8
};
// Counters that every block adds to are kept in STAT_SHARDS copies (block b uses
// copy b % STAT_SHARDS) and summed on the host: thousands of device-scope
// atomics on one word serialise at ~90 per microsecond.
constexpr uint32_t STAT_SHARDS = 64;
constexpr uint32_t TIE_SHARDS = 16;
__device__ __forceinline__ uint32_t stat_shard() { return (blockIdx.x % STAT_SHARDS) * CNT_COUNT; }

// The frame's counters, summed over their shards, to pinned host memory: word 0
// takes the frame serial last, so a host that sees the serial sees the totals.
__global__ __launch_bounds__(64) void k_publish_counts(const uint32_t *counters, uint32_t *host_block,
                                                       uint32_t serial) {
  const uint32_t t = threadIdx.x;
  if (t < (uint32_t)CNT_COUNT) {
    uint32_t sum = 0u;
    for (uint32_t sh = 0; sh < STAT_SHARDS; ++sh) sum += counters[sh * CNT_COUNT + t];
    host_block[1u + t] = sum;
  }
  __threadfence_system();
  __builtin_amdgcn_wave_barrier();
  if (t == 0) {
    __threadfence_system();
    *reinterpret_cast<volatile uint32_t *>(host_block) = serial;
  }
}

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
  const float4 *planes_paired;  // views x 6, planes 2j and 2j+1 interleaved per component
  const float *tie_eps;    // views x 6: eps * max(1,|d|)
  const uint32_t *required;
  const float4 *view;      // camera view matrix columns
  uint8_t *tvis;           // transparent_visible
  uint64_t *tkey;          // sort key of visible blend rows (sparse, per candidate)
  float4 *world_bounds;    // optional (cx,cy,cz,r) per candidate
  uint32_t *stage;         // [views + 1][stage_stride] segment-local survivor lists (candidate
                           // indices); list `views` holds the visible blend rows
  uint32_t *seg_counts;    // [row][seg_stride] survivors of the row in the segment
  uint32_t *seg_first;     // [seg_stride] first candidate of the segment
  uint32_t *counters;        // [STAT_SHARDS][CNT_COUNT]
  unsigned long long *ties;  // [TIE_SHARDS][CULL_MAX_VIEWS]
  uint64_t n;
  uint64_t stage_stride;
  uint64_t seg_stride;
  uint64_t index_base;
  uint32_t views;
  // depth pyramid of view 0 (k_classify<.., HZB = true> only)
  const float *hzb;            // mip 0 first, row-major, extent of mip m = max(extent >> m, 1)
  const uint32_t *hzb_offset;  // first texel of every mip
  const float4 *hzb_view_projection;  // the camera's, four columns
  uint32_t hzb_width, hzb_height, hzb_mips;
  float hzb_epsilon;
};

__device__ __forceinline__ uint32_t lane_id() { return threadIdx.x & 63u; }

// LDS working set of one classify block, carved out of dynamic shared memory
// (its size depends on the view count). A slot is one (round, wave) pair, i.e.
// 64 consecutive candidates:
//   mask  [CULL_SLOTS][views]   lanes of the slot that survive the view (ballot)
//   inb   [CULL_SLOTS][4]       lanes of the slot per state bucket
//   tmask [CULL_SLOTS]          lanes of the slot that are visible blend rows
//   cnt   [CULL_SLOTS][rows]    (16-bit) filled when the segment ends: start position of the
//                               slot's survivors of a row inside the segment
//   ties  [views], stat [5]
struct CullShared {
  unsigned long long *mask;
  unsigned long long *inb;
  unsigned long long *tmask;
  uint16_t *cnt;  // positions inside a segment: at most CULL_TILE
  uint32_t *ties;
  uint32_t *stat;
  uint32_t rows;
  uint32_t views;
};

__host__ __device__ inline size_t cull_shared_words(uint32_t views) {
  const uint32_t rows = views * 4u + 1u;
  static_assert(CULL_SLOTS % 2 == 0 && CULL_TILE < 65536, "16-bit positions, two to a word");
  return 2u * ((size_t)CULL_SLOTS * views + CULL_SLOTS * 4u + CULL_SLOTS) + (size_t)(CULL_SLOTS / 2) * rows +
         views + 5u;
}
__host__ __device__ inline size_t cull_shared_bytes(uint32_t views) {
  return sizeof(uint32_t) * cull_shared_words(views);
}

__device__ __forceinline__ CullShared cull_shared_init(uint32_t views) {
  extern __shared__ unsigned long long cull_lds64[];  // 8-byte aligned
  uint32_t *cull_lds = reinterpret_cast<uint32_t *>(cull_lds64);
  CullShared s;
  s.rows = views * 4u + 1u;
  s.views = views;
  s.mask = cull_lds64;
  s.inb = s.mask + CULL_SLOTS * views;
  s.tmask = s.inb + CULL_SLOTS * 4u;
  s.cnt = reinterpret_cast<uint16_t *>(s.tmask + CULL_SLOTS);
  s.ties = reinterpret_cast<uint32_t *>(s.cnt + CULL_SLOTS * s.rows);
  s.stat = s.ties + views;
  // everything starts at zero: slots a block does not run must read as "nobody survives"
  const uint32_t total = (uint32_t)cull_shared_words(views);
  for (uint32_t k = threadIdx.x; k < total; k += blockDim.x) cull_lds[k] = 0u;
  return s;
}

// vkr_vk_candidate_occluded, deferred.slang:240-303: the screen rectangle of the
// world sphere's bounding cube picks a pyramid level where it spans at most a
// few texels; the row is occluded when its nearest depth lies behind every texel
// it touches. The shader leaves the order of its float operations open; this is
// the order the CPU checker uses (separately rounded * and +, products summed as
// (c0 x + c1 y) + (c2 z + c3 w), IEEE division, floor(log2) of the integer span
// as a bit scan).
__device__ __forceinline__ bool hzb_occluded(const CullArgs &a, float cx, float cy, float cz,
                                             float world_radius) {
  const float radius = fmaxf(world_radius, 0.0f);
  const uniform_ptr<float> vp = as_uniform(a.hzb_view_projection);
  float min_x = 1e30f, min_y = 1e30f, max_x = -1e30f, max_y = -1e30f, nearest = 1.0f;
  bool usable = true;
#pragma unroll
  for (uint32_t corner = 0; corner < 8u; ++corner) {
    const float x = cx + ((corner & 1u) ? 1.0f : -1.0f) * radius;
    const float y = cy + ((corner & 2u) ? 1.0f : -1.0f) * radius;
    const float z = cz + ((corner & 4u) ? 1.0f : -1.0f) * radius;
    float clip[4];
#pragma unroll
    for (int r = 0; r < 4; ++r)
      clip[r] = (vp[r] * x + vp[4 + r] * y) + (vp[8 + r] * z + vp[12 + r] * 1.0f);
    usable = usable && isfinite(clip[0]) && isfinite(clip[1]) && isfinite(clip[2]) &&
             isfinite(clip[3]) && !(clip[3] <= 1e-6f);
    const float nx = clip[0] / clip[3], ny = clip[1] / clip[3], nz = clip[2] / clip[3];
    min_x = fminf(min_x, nx);
    min_y = fminf(min_y, ny);
    max_x = fmaxf(max_x, nx);
    max_y = fmaxf(max_y, ny);
    nearest = fminf(nearest, nz);
  }
  // a corner that cannot be projected ends the test in the shader; later corners
  // then no longer matter, so their (possibly non-finite) results are dropped here
  if (!usable) return false;
  if (nearest <= 0.0f || min_x > 1.0f || min_y > 1.0f || max_x < -1.0f || max_y < -1.0f) return false;
  const float u0 = fminf(fmaxf(min_x * 0.5f + 0.5f, 0.0f), 1.0f);
  const float v0 = fminf(fmaxf(-max_y * 0.5f + 0.5f, 0.0f), 1.0f);
  const float u1 = fminf(fmaxf(max_x * 0.5f + 0.5f, 0.0f), 1.0f);
  const float v1 = fminf(fmaxf(-min_y * 0.5f + 0.5f, 0.0f), 1.0f);
  const uint32_t bx0 = min((uint32_t)floorf(u0 * (float)a.hzb_width), a.hzb_width - 1u);
  const uint32_t by0 = min((uint32_t)floorf(v0 * (float)a.hzb_height), a.hzb_height - 1u);
  const uint32_t bx1 = min((uint32_t)floorf(u1 * (float)a.hzb_width), a.hzb_width - 1u);
  const uint32_t by1 = min((uint32_t)floorf(v1 * (float)a.hzb_height), a.hzb_height - 1u);
  const uint32_t span = max(bx1 - bx0 + 1u, by1 - by0 + 1u);
  uint32_t mip = span > 1u ? 31u - (uint32_t)__clz((int)span) : 0u;
  mip = min(mip, a.hzb_mips - 1u);
  const uint32_t mw = max(a.hzb_width >> mip, 1u), mh = max(a.hzb_height >> mip, 1u);
  const uint32_t x0 = min(bx0 >> mip, mw - 1u), x1 = min(bx1 >> mip, mw - 1u);
  const uint32_t y0 = min(by0 >> mip, mh - 1u), y1 = min(by1 >> mip, mh - 1u);
  const float *level = a.hzb + a.hzb_offset[mip];
  for (uint32_t y = y0; y <= y1; ++y)
    for (uint32_t x = x0; x <= x1; ++x)
      if (nearest <= level[(uint64_t)y * mw + x] + a.hzb_epsilon) return false;
  return true;
}

// vk_hzb_build, deferred.slang:970-998. Level 0 copies the depth image (source
// texel min(xy, source extent - 1)); every further level takes the maximum,
// starting from 0, of the 2x2 source texels min(2 xy + {0,1}, source extent - 1).
struct HzbBuildArgs {
  const float *source;
  float *destination;
  uint32_t source_width, source_height;
  uint32_t width, height;
  uint32_t copy;  // level 0
};
__global__ __launch_bounds__(256) void k_hzb_build(HzbBuildArgs a) {
  const uint32_t x = blockIdx.x * 16u + (threadIdx.x & 15u);
  const uint32_t y = blockIdx.y * 16u + (threadIdx.x >> 4);
  if (x >= a.width || y >= a.height) return;
  float out;
  if (a.copy) {
    out = a.source[(uint64_t)min(y, a.source_height - 1u) * a.source_width + min(x, a.source_width - 1u)];
  } else {
    out = 0.0f;
#pragma unroll
    for (uint32_t dy = 0; dy < 2u; ++dy)
#pragma unroll
      for (uint32_t dx = 0; dx < 2u; ++dx) {
        const uint32_t sx = min(x * 2u + dx, a.source_width - 1u);
        const uint32_t sy = min(y * 2u + dy, a.source_height - 1u);
        out = fmaxf(out, a.source[(uint64_t)sy * a.source_width + sx]);
      }
  }
  a.destination[(uint64_t)y * a.width + x] = out;
}

// Classifies one candidate against every view. Must be called by all threads of
// the wave (it votes). `valid` lanes own candidate i with model matrix m;
// `round` is the candidate's 256-wide round inside its segment. Returns the
// lane's state bucket (bits 0-3) and blend-row visibility (bit 4).
template <bool TIES, bool BOUNDS_OUT, bool HZB = false>
__device__ __forceinline__ uint32_t classify_candidate(const CullArgs &a, CullShared &s,
                                                       uint32_t round, uint64_t i, bool valid,
                                                       const Mat4 &m, float4 ls, uint32_t cf) {
  const bool leader = lane_id() == 0;
  const uint32_t slot = round * (CULL_BLOCK / 64u) + (threadIdx.x >> 6);
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
  const uniform_ptr<float> paired = as_uniform(a.planes_paired);
  const uniform_ptr<float> tie_eps = as_uniform(a.tie_eps);
  const uniform_ptr<uint32_t> required = as_uniform(a.required);
  if (blend_row) tv = bounds_valid ? sphere_visible(paired, c.x, c.y, c.z, radius) : true;
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
      s.tmask[slot] = mt;
      if (mb) atomicAdd(&s.stat[0], (uint32_t)__popcll(mb));
      if (mc) atomicAdd(&s.stat[1], (uint32_t)__popcll(mc));
      if (mo) atomicAdd(&s.stat[2], (uint32_t)__popcll(mo));
      if (mx) atomicAdd(&s.stat[3], (uint32_t)__popcll(mx));
    }
  }

  // views: deferred.slang:316-351
#pragma unroll
  for (uint32_t q = 0; q < 4u; ++q) {
    const unsigned long long in_bucket = __ballot(bucket == q);
    if (leader) s.inb[slot * 4u + q] = in_bucket;
  }
  const bool gate = valid && bucket < 4u && !(cf & CF_UNPUBLISHED);
  // the view-independent part of the outcome as wave masks: what is left per view
  // are two compares whose results are masks already
  const unsigned long long gate_mask = __ballot(gate);
  const unsigned long long unbounded_mask = __ballot(!bounds_valid);
  for (uint32_t v = 0; v < a.views; ++v) {
    const uint32_t req = required[v];
    // evaluated for the whole wave without a branch: the plane loads then issue
    // together with the gate word, ahead of any wait
    const bool inside = sphere_visible(paired + v * 24u, c.x, c.y, c.z, radius);
    if (!TIES && !HZB) {
      const unsigned long long mp =
          gate_mask & __ballot((cf & req) == req) & (__ballot(inside) | unbounded_mask);
      if (leader) s.mask[slot * a.views + v] = mp;
      continue;
    }
    bool pass = gate && (cf & req) == req;
    bool tie = false;
    if (!TIES) pass = pass && (inside || !bounds_valid);
    if (TIES && pass && bounds_valid) {
      const uniform_ptr<float> vp = planes + v * 24u;
      pass = inside;
      {
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
    if (HZB && v == 0u) {  // deferred.slang:340-347
      const bool occluded = pass && bounds_valid && hzb_occluded(a, c.x, c.y, c.z, radius);
      pass = pass && !occluded;
      const unsigned long long mocc = __ballot(occluded);
      if (leader && mocc) atomicAdd(&s.stat[4], (uint32_t)__popcll(mocc));
    }
    const unsigned long long mp = __ballot(pass);
    if (leader) s.mask[slot * a.views + v] = mp;
  }
  return (bucket & 0xfu) | (tv ? 0x10u : 0u);
}

// Ends a segment: turns the slot counts into start positions, stores the row
// counts, packs the survivors into the segment's stretch of the staging lists
// and flushes the statistics. `packed` holds classify_candidate's return value
// of the thread's candidate in round r in bits [8r, 8r+5); candidate (round r,
// thread t) is segment-relative index `rel0 + 256 r + t`, the segment's staging
// stretch starts at candidate `first`, and rel0 = (tile start) - first may be
// "negative" (wraps): only survivors' indices are formed.
template <bool TIES>
__device__ __forceinline__ void segment_finish(const CullArgs &a, CullShared &s, uint64_t seg,
                                               uint64_t first, uint32_t tile_to_first,
                                               uint32_t packed) {
  const uint32_t t = threadIdx.x;
  const uint32_t lane = lane_id();
  const uint32_t wave = t >> 6;
  const unsigned long long lt_mask = (1ull << lane) - 1ull;
  __syncthreads();
  // Sixteen lanes per row: the slot's survivors of the row (its view's mask cut
  // to the row's bucket), then an exclusive prefix over the (round, wave) slots
  // in candidate order. A wave holds the four bucket rows of one view, so the
  // starts of buckets 1-3 behind bucket 0 come from three shuffles.
  for (uint32_t r0 = 0; r0 < s.rows; r0 += CULL_BLOCK / CULL_SLOTS) {
    const uint32_t slot = t % CULL_SLOTS;
    const uint32_t r = r0 + t / CULL_SLOTS;
    const bool live = r < s.rows;
    uint32_t c = 0u;
    if (live)
      c = r < a.views * 4u
              ? (uint32_t)__popcll(s.mask[slot * a.views + (r >> 2)] & s.inb[slot * 4u + (r & 3u)])
              : (uint32_t)__popcll(s.tmask[slot]);
    uint32_t incl = c;
#pragma unroll
    for (uint32_t d = 1; d < CULL_SLOTS; d <<= 1) {
      const uint32_t u = __shfl_up(incl, d, CULL_SLOTS);
      if (slot >= d) incl += u;
    }
    const uint32_t t0 = __shfl(incl, 15, 64), t1 = __shfl(incl, 31, 64), t2 = __shfl(incl, 47, 64);
    const uint32_t b = lane >> 4;  // bucket of the row inside its view
    const uint32_t bucket_start = (b > 0u ? t0 : 0u) + (b > 1u ? t1 : 0u) + (b > 2u ? t2 : 0u);
    if (live) {
      s.cnt[slot * s.rows + r] = (uint16_t)(bucket_start + incl - c);
      if (slot == CULL_SLOTS - 1u) a.seg_counts[(uint64_t)r * a.seg_stride + seg] = incl;
    }
  }
  if (t == 0) a.seg_first[seg] = (uint32_t)first;
  __syncthreads();

  // View by view, the four rounds of the thread side by side: their mask reads
  // are independent, so a view costs one LDS round trip, not four.
  unsigned long long below[CULL_ROUNDS];  // lower lanes of the slot holding this lane's bucket
  const uint16_t *start[CULL_ROUNDS];
  uint32_t rel[CULL_ROUNDS];
#pragma unroll
  for (uint32_t round = 0; round < (uint32_t)CULL_ROUNDS; ++round) {
    const uint32_t bucket = (packed >> (8u * round)) & 0xfu;
    const uint32_t slot = round * (CULL_BLOCK / 64u) + wave;
    rel[round] = round * CULL_BLOCK + t - tile_to_first;  // candidate index - first
    below[round] = bucket < 4u ? s.inb[slot * 4u + bucket] & lt_mask : 0ull;
    start[round] = s.cnt + slot * s.rows + (bucket & 3u);
  }
  for (uint32_t v = 0; v < a.views; ++v) {
    unsigned long long mp[CULL_ROUNDS];
#pragma unroll
    for (uint32_t round = 0; round < (uint32_t)CULL_ROUNDS; ++round)
      mp[round] = s.mask[(round * (CULL_BLOCK / 64u) + wave) * a.views + v];  // the same for the whole wave
    uint32_t *list = a.stage + (uint64_t)v * a.stage_stride + first;
#pragma unroll
    for (uint32_t round = 0; round < (uint32_t)CULL_ROUNDS; ++round) {
      if (mp[round] == 0ull) continue;  // wave-uniform
      if ((mp[round] >> lane) & 1ull) {
        const uint32_t pos = start[round][v * 4u] + (uint32_t)__popcll(mp[round] & below[round]);
        list[pos] = (uint32_t)first + rel[round];
      }
    }
  }
#pragma unroll
  for (uint32_t round = 0; round < (uint32_t)CULL_ROUNDS; ++round) {
    if ((packed >> (8u * round)) & 0x10u) {  // a visible blend row
      const uint32_t slot = round * (CULL_BLOCK / 64u) + wave;
      const unsigned long long mt = s.tmask[slot];
      const uint32_t pos = s.cnt[slot * s.rows + a.views * 4u] + (uint32_t)__popcll(mt & lt_mask);
      a.stage[(uint64_t)a.views * a.stage_stride + first + pos] = (uint32_t)first + rel[round];
    }
  }
  if (t == 0) {
    uint32_t *shard = a.counters + stat_shard();
    if (s.stat[0]) atomicAdd(&shard[CNT_WITHOUT_BOUNDS], s.stat[0]);
    if (s.stat[1]) atomicAdd(&shard[CNT_CULLED_CAMERA], s.stat[1]);
    if (s.stat[2]) atomicAdd(&shard[CNT_CAMERA_OPAQUE], s.stat[2]);
    if (s.stat[3]) atomicAdd(&shard[CNT_TRANSMISSION], s.stat[3]);
    if (s.stat[4]) atomicAdd(&shard[CNT_OCCLUDED], s.stat[4]);
  }
  if (TIES)
    for (uint32_t r = t; r < a.views; r += blockDim.x)
      if (s.ties[r])
        atomicAdd(&a.ties[(blockIdx.x % TIE_SHARDS) * CULL_MAX_VIEWS + r],
                  (unsigned long long)s.ties[r]);
}

// A segment without candidates still owns a column of the count table.
__device__ __forceinline__ void segment_empty(const CullArgs &a, uint64_t seg) {
  for (uint32_t r = threadIdx.x; r < a.views * 4u + 1u; r += blockDim.x)
    a.seg_counts[(uint64_t)r * a.seg_stride + seg] = 0u;
}

// Stand-alone classify: one block per segment of 1024 candidates, four rounds of
// 256. With the identity candidate->node mapping the matrices arrive as
// coalesced 16-byte loads staged through LDS (thread t then reads its own four
// columns); otherwise each thread gathers its node's matrix.
template <bool IDENTITY, bool TIES, bool BOUNDS_OUT, bool HZB = false>
__global__ __launch_bounds__(CULL_BLOCK) void k_classify(CullArgs a) {
  __shared__ float4 s_mat[IDENTITY ? CULL_BLOCK * MAT_LDS_STRIDE : 1];
  CullShared s = cull_shared_init(a.views);
  const uint32_t t = threadIdx.x;
  const uint64_t seg = blockIdx.x;
  const uint64_t first = seg * CULL_TILE;
  const uint64_t last = a.n - 1u;  // n > 0: the grid has a block per segment
  uint32_t packed = 0u;
  __syncthreads();  // accumulators cleared

  // all rounds' small inputs are requested up front (clamped, unconditional)
  float4 ls_r[CULL_ROUNDS];
  uint32_t cf_r[CULL_ROUNDS];
#pragma unroll
  for (int round = 0; round < CULL_ROUNDS; ++round) {
    const uint64_t i = min(first + (uint64_t)round * CULL_BLOCK + t, last);
    ls_r[round] = a.sphere[i];
    cf_r[round] = a.cflags[i];
  }
#pragma unroll
  for (int round = 0; round < CULL_ROUNDS; ++round) {
    const uint64_t base = first + (uint64_t)round * CULL_BLOCK;
    if (base >= a.n) break;  // block-uniform
    const uint64_t i = base + t;
    const bool valid = i < a.n;
    const uint32_t vm = valid ? 0xffffffffu : 0u;
    const uint32_t cf = (cf_r[round] & vm) | (0xf0u & ~vm);  // invalid bucket: fails every gate
    Mat4 m;
    if (IDENTITY) {
      const uint64_t last4 = a.n * 4u - 1u;
      float4 v[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) v[k] = a.world[min(base * 4u + t + 256u * (uint32_t)k, last4)];
      __syncthreads();  // the previous round's readers of s_mat are done
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
      const float4 *p = a.world + (uint64_t)a.node[min(i, last)] * 4u;
      m.c[0] = p[0];
      m.c[1] = p[1];
      m.c[2] = p[2];
      m.c[3] = p[3];
    }
    packed |= classify_candidate<TIES, BOUNDS_OUT, HZB>(a, s, (uint32_t)round, i, valid, m,
                                                        ls_r[round], cf)
              << (8 * round);
  }
  segment_finish<TIES>(a, s, seg, first, 0u, packed);
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
  const uint32_t *seg_counts;  // [row][seg_stride]
  uint32_t *seg_offsets;       // [row][seg_stride] exclusive prefix of the counts
  uint32_t *row_totals;        // [rows]: survivors per (view, bucket) row and of the transparent row
  uint32_t *counters;
  uint64_t segments;  // segments in use this frame
  uint64_t seg_stride;
  uint32_t views;
  uint64_t transparent_capacity;
  uint32_t *host_hint;   // pinned host word: last frame's transparent count (may be null)
  uint32_t *clear;       // the NEXT frame's counter block, zeroed here (may be null)
  uint32_t clear_words;
};

__device__ __forceinline__ uint32_t sum4(uint4 v) { return v.x + v.y + v.z + v.w; }

// One uint4 of a count row; entries at or beyond `segments` read as zero (the
// rows are padded to whole uint4s and the padding is never written).
__device__ __forceinline__ uint4 load_counts4(const uint4 *row, uint64_t q, uint64_t segments) {
  uint4 v = row[q];
  if (q * 4u + 1u >= segments) v.y = 0u;
  if (q * 4u + 2u >= segments) v.z = 0u;
  if (q * 4u + 3u >= segments) v.w = 0u;
  return v;
}

// Exclusive scan of the segment counts, one 1024-thread block per row. A wave
// owns a contiguous run of the row and reads it 64 x 16 bytes at a time
// (coalesced); sweep 1 only totals each wave's run, one barrier combines the 16
// totals, and sweep 2 re-reads the run (from cache), scans it 64 uint4 at a time
// with wave shuffles and writes the exclusive prefix: two memory round trips for
// any segment count.
__global__ __launch_bounds__(SCAN_BLOCK) void k_scan(ScanArgs a) {
  __shared__ uint32_t s_wave[SCAN_BLOCK / 64];
  const uint32_t r = blockIdx.x;
  const uint4 *row = reinterpret_cast<const uint4 *>(a.seg_counts + (uint64_t)r * a.seg_stride);
  uint4 *out = reinterpret_cast<uint4 *>(a.seg_offsets + (uint64_t)r * a.seg_stride);
  const uint64_t quads = (a.segments + 3u) / 4u;
  const uint32_t w = threadIdx.x >> 6;
  const uint32_t lane = lane_id();
  const uint64_t wave_quads = ((quads + SCAN_BLOCK / 64 - 1) / (SCAN_BLOCK / 64) + 63u) & ~(uint64_t)63u;
  const uint64_t qb = min((uint64_t)w * wave_quads, quads);
  const uint64_t qe = min(qb + wave_quads, quads);

  uint32_t mine = 0u;
  for (uint64_t q = qb + lane; q < qe; q += 64u) mine += sum4(load_counts4(row, q, a.segments));
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
    const uint4 v = valid ? load_counts4(row, q, a.segments) : make_uint4(0u, 0u, 0u, 0u);
    const uint32_t s = sum4(v);
    const uint32_t incl = wave_inclusive_scan(s);
    uint4 o;
    o.x = carry + incl - s;
    o.y = o.x + v.x;
    o.z = o.y + v.y;
    o.w = o.z + v.z;
    if (valid) out[q] = o;
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
// become execution ranges clamped to the partition capacity. The view's list
// has a capacity of its own here (the reference sizes it for every candidate,
// so it cannot run out there): a bucket keeps what still fits behind the
// buckets before it, and what does not fit is counted in overflow_count like
// the rows beyond a partition.
__host__ __device__ __forceinline__ uint32_t bucket_room(uint32_t count, uint32_t partition_capacity,
                                                         uint64_t list_capacity, uint32_t cursor) {
  const uint32_t cap = partition_capacity ? partition_capacity : 0xffffffffu;
  const uint32_t written = count < cap ? count : cap;
  const uint64_t room = list_capacity > cursor ? list_capacity - cursor : 0u;
  return (uint64_t)written < room ? written : (uint32_t)room;
}

__device__ __forceinline__ CompactionState prefix_state(const uint32_t *totals4,
                                                        uint32_t partition_capacity,
                                                        uint64_t list_capacity) {
  CompactionState st;
  uint32_t cursor = 0, overflow = 0;
  for (uint32_t b = 0; b < 4u; ++b) {
    const uint32_t count = totals4[b];
    const uint32_t written = bucket_room(count, partition_capacity, list_capacity, cursor);
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

struct GatherArgs {
  const uint32_t *stage;        // [views + 1][stage_stride]
  const uint32_t *seg_offsets;  // [row][seg_stride], from k_scan
  const uint32_t *seg_first;
  const uint32_t *row_totals;
  const uint64_t *tkey;
  const uint32_t *remap;    // candidate -> packed row index when a publication is set, else null
  CompactionState *states;  // written by block 0
  uint32_t *visible;        // [view][visible_stride]
  uint64_t *keys;           // compacted sort keys of the visible blend rows
  uint64_t segments;
  uint64_t stage_stride;
  uint64_t seg_stride;
  uint64_t visible_stride;
  uint64_t transparent_capacity;
  uint64_t index_base;
  uint32_t views;
  uint32_t partition_capacity;
};

constexpr int GATHER_BLOCK = 256;  // four waves, one segment each
constexpr int GATHER_UNROLL = 4;

__host__ __device__ inline size_t gather_shared_bytes(uint32_t views) {
  return sizeof(uint32_t) * 3u * (size_t)(views * 4u + 1u) * (GATHER_BLOCK / 64);
}

// One wave per segment. The lanes first work out, for every (view, bucket) row,
// how many staged survivors the segment has (next offset - own offset), where
// the run sits in the staging list (buckets of a view back to back) and where it
// goes (bucket start + scanned offset, clamped to the partition capacity); then
// all runs are copied as one flat list of entries. The visible blend rows take
// their sort keys along.
__global__ __launch_bounds__(GATHER_BLOCK) void k_gather(GatherArgs a) {
  extern __shared__ uint32_t gather_lds[];
  const uint32_t t = threadIdx.x;
  const uint32_t lane = lane_id();
  const uint32_t wave = t >> 6;
  const uint32_t rows = a.views * 4u + 1u;
  const uint32_t vrows = a.views * 4u;
  if (blockIdx.x == 0)
    for (uint32_t v = t; v < a.views; v += GATHER_BLOCK)
      a.states[v] = prefix_state(a.row_totals + v * 4u, a.partition_capacity, a.visible_stride);

  const uint64_t seg = (uint64_t)blockIdx.x * (GATHER_BLOCK / 64) + wave;
  if (seg >= a.segments) return;  // wave-uniform; no block-wide barrier below
  uint32_t *e_start = gather_lds + wave * 3u * rows;  // first flat entry of the row
  uint32_t *e_src = e_start + rows;                   // staging position of the row's run
  uint32_t *e_dst = e_src + rows;                     // output slot of the row's run
  const uint32_t first = a.seg_first[seg];
  const bool last_seg = seg + 1u == a.segments;

  uint32_t flat = 0u;       // running number of entries (wave-uniform)
  uint32_t t_count = 0u, t_offset = 0u;
  for (uint32_t r0 = 0; r0 < rows; r0 += 64u) {
    const uint32_t r = r0 + lane;
    const bool live = r < rows;
    const uint32_t rc = live ? r : rows - 1u;
    const uint32_t off = a.seg_offsets[(uint64_t)rc * a.seg_stride + seg];
    const uint32_t nxt = last_seg ? a.row_totals[rc] : a.seg_offsets[(uint64_t)rc * a.seg_stride + seg + 1u];
    uint32_t cnt = live ? nxt - off : 0u;
    // staging: the buckets of a view lie back to back (four lanes per view)
    uint32_t before = 0u;
#pragma unroll
    for (uint32_t d = 1; d < 4u; ++d) {
      const uint32_t u = __shfl_up(cnt, d, 4);
      if ((lane & 3u) >= d) before += u;
    }
    // output: start of the bucket inside the view's list (as prefix_state lays the
    // buckets out), then the scanned offset
    uint32_t range = 0u, kept = 0u;
    if (live && r < vrows) {
      for (uint32_t b = 0; b < (r & 3u); ++b)
        range += bucket_room(a.row_totals[(r & ~3u) + b], a.partition_capacity, a.visible_stride, range);
      kept = bucket_room(a.row_totals[r], a.partition_capacity, a.visible_stride, range);
    }
    if (r == vrows) {  // the transparent row: copied with its keys below
      t_count = cnt;
      t_offset = off;
      cnt = 0u;
    } else if (live) {
      // entries beyond the partition capacity or the list capacity are dropped
      cnt = min(cnt, kept > off ? kept - off : 0u);
    }
    const uint32_t incl = wave_inclusive_scan(cnt);
    if (live) {
      e_start[r] = flat + incl - cnt;
      e_src[r] = first + before;
      e_dst[r] = range + off;
    }
    flat += (uint32_t)__shfl((int)incl, 63, 64);
  }
  __builtin_amdgcn_wave_barrier();  // the row table is read by other lanes of the wave
  t_count = (uint32_t)__shfl((int)t_count, (int)(vrows & 63u), 64);
  t_offset = (uint32_t)__shfl((int)t_offset, (int)(vrows & 63u), 64);

  // flat copy: entry e belongs to the last row whose first entry is <= e. Four
  // entries per lane and step, so that four staging reads are in flight before
  // the first store needs its value.
  for (uint32_t e0 = lane; e0 < flat; e0 += 64u * GATHER_UNROLL) {
    uint32_t idx[GATHER_UNROLL];
    uint64_t out[GATHER_UNROLL];
#pragma unroll
    for (int u = 0; u < GATHER_UNROLL; ++u) {
      const uint32_t e = min(e0 + 64u * (uint32_t)u, flat - 1u);  // clamped: the load is unconditional
      uint32_t lo = 0u, hi = vrows;  // rows [lo, hi) still possible; e_start is non-decreasing
      while (hi - lo > 1u) {
        const uint32_t mid = (lo + hi) >> 1;
        if (e_start[mid] <= e) lo = mid; else hi = mid;
      }
      const uint32_t k = e - e_start[lo];
      const uint32_t v = lo >> 2;
      idx[u] = a.stage[(uint64_t)v * a.stage_stride + e_src[lo] + k];
      out[u] = (uint64_t)v * a.visible_stride + e_dst[lo] + k;
    }
    if (a.remap) {
#pragma unroll
      for (int u = 0; u < GATHER_UNROLL; ++u) idx[u] = a.remap[idx[u]];
    } else {
#pragma unroll
      for (int u = 0; u < GATHER_UNROLL; ++u) idx[u] = (uint32_t)(a.index_base + idx[u]);
    }
#pragma unroll
    for (int u = 0; u < GATHER_UNROLL; ++u)
      if (e0 + 64u * (uint32_t)u < flat) a.visible[out[u]] = idx[u];
  }
  for (uint32_t k = lane; k < t_count; k += 64u) {
    const uint64_t pos = (uint64_t)t_offset + k;
    if (pos < a.transparent_capacity)
      a.keys[pos] = a.tkey[a.stage[(uint64_t)a.views * a.stage_stride + first + k]];
  }
}

}  // namespace vkr
