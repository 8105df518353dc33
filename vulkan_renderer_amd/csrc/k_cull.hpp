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
// Up to this many views a classify block expands its segment's survivors itself
// (segment_finish_staged) instead of leaving its masks to the gather kernel.
constexpr uint32_t STAGED_MAX_VIEWS = 8;

// prepared candidate word
constexpr uint32_t CF_BOUNDS_VALID = 0x1u;    // VKR_WORLD_DRAW_CANDIDATE_BOUNDS_VALID
constexpr uint32_t CF_CAMERA_OPAQUE = 0x2u;   // VKR_WORLD_DRAW_CANDIDATE_CAMERA_OPAQUE
constexpr uint32_t CF_SHADOW_CASTER = 0x4u;   // VKR_WORLD_DRAW_CANDIDATE_SHADOW_CASTER
constexpr uint32_t CF_REF_MASK = 0x7u;
constexpr uint32_t CF_BUCKET_SHIFT = 4u;      // bits 4..7 state bucket
constexpr uint32_t CF_BLEND = 0x100u;         // world_transparent
constexpr uint32_t CF_TRANSMISSIVE = 0x200u;
constexpr uint32_t CF_UNPUBLISHED = 0x400u;   // omitted by the backend pack: in no GPU view
static_assert((CF_REF_MASK | (0xfu << CF_BUCKET_SHIFT) | CF_BLEND | CF_TRANSMISSIVE | CF_UNPUBLISHED) <= 0xffffu,
              "the candidate word has a 16-bit copy");

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
                                                            uint16_t *__restrict__ cflags16,
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
  cflags16[i] = (uint16_t)f;
}

// Floats as unsigned keys in the same order (every finite value maps above 0), so
// that a box can be accumulated with integer atomic maxima from zeroed memory:
// words 0-2 hold the largest ~key of the low corner (i.e. the smallest value),
// words 3-5 the largest key of the high corner. A word still 0 means "nothing".
__host__ __device__ __forceinline__ uint32_t order_key(float f) {
  const uint32_t b = __builtin_bit_cast(uint32_t, f);
  return b ^ ((b >> 31) ? 0xffffffffu : 0x80000000u);
}
__host__ __device__ __forceinline__ float order_key_value(uint32_t k) {
  return __builtin_bit_cast(float, k ^ ((k >> 31) ? 0x80000000u : 0xffffffffu));
}
__device__ __forceinline__ uint32_t wave_max_u32(uint32_t v) {
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) v = max(v, (uint32_t)__shfl_xor((int)v, d, 64));
  return v;
}

struct CullArgs {
  const float4 *world;     // node matrices, 4 float4 each
  const uint32_t *node;    // candidate -> node, null = identity
  const float4 *sphere;    // local sphere per candidate
  const uint32_t *cflags;  // prepared word per candidate
  const uint16_t *cflags16;  // the same word in 16 bits (every CF_ bit is below 0x800): what the fused
                             //   kernel reads, 2 bytes per candidate less
  const float4 *planes;    // views x 6
  const float4 *planes_paired;  // views x 6, planes 2j and 2j+1 interleaved per component
  const float *tie_eps;    // views x 6: eps * max(1,|d|)
  const uint32_t *required;
  const float4 *view;      // camera view matrix columns
  uint8_t *tvis;           // transparent_visible
  uint64_t *tkey;          // sort key of visible blend rows (sparse, per candidate)
  float4 *tworld_sparse;   // optional: the world matrix of every visible blend row, under its
                           // candidate index (see classify_candidate)
  float4 *world_bounds;    // optional (cx,cy,cz,r) per candidate
  unsigned long long *seg_masks;  // [segment][seg_record_words(views)] survivor masks (see CullShared)
  uint32_t *seg_counts;    // [row][seg_stride] survivors of the row in the segment
  uint32_t *seg_first;     // [seg_stride] candidate of the segment's slot 0, lane 0
  uint32_t *counters;        // [STAT_SHARDS][CNT_COUNT]
  unsigned long long *ties;  // [TIE_SHARDS][CULL_MAX_VIEWS]
  uint32_t *caster_bounds;   // optional: [6] box of the shadow casters' world spheres (see order_key)
  uint64_t n;
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

// Element `index` of an array whose base is the same for the whole wave, addressed
// as that base (a scalar register pair) plus a 32-bit byte offset per lane: no
// 64-bit address arithmetic and no 64-bit address registers per lane. The caller
// guarantees that index * sizeof(T) fits 32 bits.
template <typename T>
__device__ __forceinline__ const T *at32(const T *uniform_base, uint32_t index) {
  return reinterpret_cast<const T *>(reinterpret_cast<const char *>(uniform_base) +
                                     index * (uint32_t)sizeof(T));
}
template <typename T>
__device__ __forceinline__ T *at32(T *uniform_base, uint32_t index) {
  return reinterpret_cast<T *>(reinterpret_cast<char *>(uniform_base) + index * (uint32_t)sizeof(T));
}

// Inclusive prefix sum over the 64 lanes with data-parallel-primitive moves (no
// LDS round trips): four shifted adds scan each row of 16 lanes, then lane 15 of
// rows 0 and 2 is added to rows 1 and 3, then lane 31 to rows 2 and 3. A lane
// whose source lies outside its row, or whose row is masked, receives 0.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ uint32_t dpp_or_zero(uint32_t v) {
  return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, ROW_MASK, 0xf, false);
}
__device__ __forceinline__ uint32_t wave_inclusive_scan(uint32_t v) {
  v += dpp_or_zero<0x111, 0xf>(v);  // row_shr:1
  v += dpp_or_zero<0x112, 0xf>(v);  // row_shr:2
  v += dpp_or_zero<0x114, 0xf>(v);  // row_shr:4
  v += dpp_or_zero<0x118, 0xf>(v);  // row_shr:8
  v += dpp_or_zero<0x142, 0xa>(v);  // row_bcast:15 into rows 1 and 3
  v += dpp_or_zero<0x143, 0xc>(v);  // row_bcast:31 into rows 2 and 3
  return v;
}

// Sixteen bytes that only pass through registers on their way from global memory
// to LDS. As the native vector type the copies are one load and one store; as
// float4 (a struct) the compiler chains block copies through a stack slot.
typedef float f32x4 __attribute__((ext_vector_type(4)));

// STREAM: a frame whose arrays do not fit the last-level cache (256 MB) reads
// every input once and writes every output once, so keeping any of it in a
// cache only evicts what IS read twice (the parents' matrices); such frames
// mark their loads and stores as streaming. A scene that fits the cache is found
// there by the next frame and must not use the hint (1M nodes: 30 -> 32.5 us).
template <bool STREAM, typename T>
__device__ __forceinline__ T load_once(const T *p) {
  if constexpr (STREAM) return __builtin_nontemporal_load(p);
  return *p;
}
template <bool STREAM, typename T>
__device__ __forceinline__ void store_once(T *p, T v) {
  if constexpr (STREAM) __builtin_nontemporal_store(v, p);
  else *p = v;
}


// LDS working set of one classify block, carved out of dynamic shared memory
// (its size depends on the view count). A slot is one (round, wave) pair, i.e.
// 64 consecutive candidates:
//   mask  [views][CULL_SLOTS]   lanes of the slot that survive the view (ballot)
//   inb   [CULL_SLOTS][4]       lanes of the slot per state bucket
//   tmask [CULL_SLOTS]          lanes of the slot that are visible blend rows
//   ties  [views], stat [5], box [6]
// mask, inb and tmask, in this order and without gaps, are the segment's record:
// it goes to global memory as it is when the segment ends, and the gather kernel
// turns the masks into the ordered lists once every segment's counts are scanned.
struct CullShared {
  unsigned long long *mask;
  unsigned long long *inb;
  unsigned long long *tmask;
  uint32_t *ties;
  uint32_t *stat;
  uint32_t *box;  // the block's share of CullArgs::caster_bounds
  uint16_t *pair;  // [views][64] (views <= STAGED_MAX_VIEWS): see segment_finish_staged
  uint32_t rows;
  uint32_t views;
};

// 64-bit words of a segment's record
__host__ __device__ inline uint32_t seg_record_words(uint32_t views) { return CULL_SLOTS * (views + 5u); }

__host__ __device__ inline size_t cull_shared_words(uint32_t views) {
  return 2u * (size_t)seg_record_words(views) + views + 5u + 6u +
         (views <= STAGED_MAX_VIEWS ? views * 32u : 0u);  // 64 16-bit pair bases per view
}
// A block may hold the working set more than once (a block that walks several
// segments alternates between two sets); a set starts on an 8-byte boundary.
__host__ __device__ inline size_t cull_shared_set_words(uint32_t views) {
  return (cull_shared_words(views) + 1u) & ~(size_t)1u;
}
__host__ __device__ inline size_t cull_shared_bytes(uint32_t views, uint32_t sets = 1) {
  return sizeof(uint32_t) * cull_shared_set_words(views) * sets;
}

// Set `which` of the block's working sets (no clearing).
__device__ __forceinline__ CullShared cull_shared_set(uint32_t views, uint32_t which) {
  extern __shared__ unsigned long long cull_lds64[];  // 8-byte aligned
  CullShared s;
  s.rows = views * 4u + 1u;
  s.views = views;
  s.mask = cull_lds64 + which * (cull_shared_set_words(views) / 2u);
  s.inb = s.mask + CULL_SLOTS * views;
  s.tmask = s.inb + CULL_SLOTS * 4u;
  s.ties = reinterpret_cast<uint32_t *>(s.tmask + CULL_SLOTS);
  s.stat = s.ties + views;
  s.box = s.stat + 5;
  s.pair = reinterpret_cast<uint16_t *>(s.box + 6);
  return s;
}

// The slots of a set that the calling wave writes (slot = round * 4 + wave), set
// back to "nobody survives": what a wave does when it enters a further segment
// (rounds the segment does not run must read as empty). views + 5 <= 64.
__device__ __forceinline__ void cull_clear_wave_slots(const CullShared &s) {
  const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
#pragma unroll
  for (uint32_t r = 0; r < (uint32_t)CULL_ROUNDS; ++r) {
    const uint32_t slot = r * (CULL_BLOCK / 64u) + wave;
    if (lane < s.views)
      s.mask[lane * CULL_SLOTS + slot] = 0ull;
    else if (lane < s.views + 4u)
      s.inb[slot * 4u + (lane - s.views)] = 0ull;
    else if (lane == s.views + 4u)
      s.tmask[slot] = 0ull;
  }
}

__device__ __forceinline__ CullShared cull_shared_init(uint32_t views, uint32_t sets = 1) {
  extern __shared__ unsigned long long cull_lds64[];  // 8-byte aligned
  uint32_t *cull_lds = reinterpret_cast<uint32_t *>(cull_lds64);
  CullShared s;
  s.rows = views * 4u + 1u;
  s.views = views;
  s.mask = cull_lds64;
  s.inb = s.mask + CULL_SLOTS * views;
  s.tmask = s.inb + CULL_SLOTS * 4u;
  s.ties = reinterpret_cast<uint32_t *>(s.tmask + CULL_SLOTS);
  s.stat = s.ties + views;
  s.box = s.stat + 5;
  s.pair = reinterpret_cast<uint16_t *>(s.box + 6);
  // everything starts at zero: slots a block does not run must read as "nobody survives"
  const uint32_t total = (uint32_t)(cull_shared_set_words(views) * sets);
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
  if (a.caster_bounds != nullptr) {  // wave-uniform
    // the box around the world spheres of the rows that cast shadows
    const bool casts = valid && bounds_valid && (cf & CF_SHADOW_CASTER) && !(cf & CF_UNPUBLISHED);
    const float centre[3] = {c.x, c.y, c.z};
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const uint32_t lo = wave_max_u32(casts ? ~order_key(centre[k] - radius) : 0u);
      const uint32_t hi = wave_max_u32(casts ? order_key(centre[k] + radius) : 0u);
      if (leader && lo) atomicMax(&s.box[k], lo);
      if (leader && hi) atomicMax(&s.box[3 + k], hi);
    }
  }

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
    // When the frame's compaction, sort and emission run beside the NEXT frame's
    // transform, the emission can no longer read the row's matrix from world[]: it
    // is set aside here, under the candidate's own index (a visible blend row is
    // rare - well under one row in a hundred - so the copy is a few scattered
    // stores, and no place in a compact array has to be known yet).
    if (a.tworld_sparse != nullptr) {  // wave-uniform
      float4 *aside = a.tworld_sparse + i * 4u;
      aside[0] = m.c[0];
      aside[1] = m.c[1];
      aside[2] = m.c[2];
      aside[3] = m.c[3];
    }
  }
  {
    const unsigned long long mt = __ballot(tv);
    const unsigned long long mc = __ballot(blend_row && !tv);
    const unsigned long long mb = __ballot(valid && !bounds_valid);
    const unsigned long long mo = __ballot(valid && (cf & CF_CAMERA_OPAQUE));
    const unsigned long long mx = __ballot(valid && (cf & CF_TRANSMISSIVE));
    // views: deferred.slang:316-351. The slot's lanes per bucket leave with the
    // other words of the slot under ONE branch (a branch per word otherwise).
    const unsigned long long b0 = __ballot(bucket == 0u), b1 = __ballot(bucket == 1u);
    const unsigned long long b2 = __ballot(bucket == 2u), b3 = __ballot(bucket == 3u);
    if (leader) {
      unsigned long long *in_bucket = s.inb + slot * 4u;
      in_bucket[0] = b0;
      in_bucket[1] = b1;
      in_bucket[2] = b2;
      in_bucket[3] = b3;
      s.tmask[slot] = mt;
      if (mb) atomicAdd(&s.stat[0], (uint32_t)__popcll(mb));
      if (mc) atomicAdd(&s.stat[1], (uint32_t)__popcll(mc));
      if (mo) atomicAdd(&s.stat[2], (uint32_t)__popcll(mo));
      if (mx) atomicAdd(&s.stat[3], (uint32_t)__popcll(mx));
    }
  }

  const bool gate = valid && bucket < 4u && !(cf & CF_UNPUBLISHED);
  // the view-independent part of the outcome as wave masks: what is left per view
  // are two compares whose results are masks already
  const unsigned long long gate_mask = __ballot(gate);
  const unsigned long long unbounded_mask = __ballot(!bounds_valid);
  // (views mostly ask for the same flags as the view before them - every shadow
  // and probe view for SHADOW_CASTER -, so the lanes that pass the gate and carry
  // a view's flags are only worked out when the word changes)
  uint32_t flags_of = ~0u;  // the word carried_mask belongs to (no view asks for every bit)
  unsigned long long carried_mask = 0ull;
  // The survivors' ballot of view v is parked in lane v of a register pair (two
  // lane writes, no branch) and the views' ballots go to the table with one store
  // behind the loop, lane v writing view v's (views <= 64).
  uint32_t ballots_lo = 0u, ballots_hi = 0u;
  // v_writelane_b32 takes one scalar register besides the lane number, which
  // therefore travels in M0. M0 is set right here and nothing else in these
  // kernels reads it (the LDS instructions of this chip do not), so the
  // compiler's remark that it does not track the register is without effect.
  // One hazard the compiler does not cover inside an asm statement:
  // v_writelane_b32 right behind a VECTOR instruction that wrote its data
  // register (v_cmp) reads the old value on this chip (measured: wrong in 99.9 %
  // of cases without wait states, never with four); a register written by a
  // scalar instruction needs none. So the value is always made by a scalar
  // instruction INSIDE the statement: the last AND of the fast path's masks
  // (park_and), or a copy (park).
  // (A select on lane == v instead costs three vector instructions per view:
  // 64 views 1036 us against 1010 us.)
  auto park_and = [&](uint32_t v, unsigned long long x, unsigned long long y) {
    uint32_t made_lo, made_hi;
    asm("s_and_b32 %2, %4, %6\n\ts_and_b32 %3, %5, %7\n\ts_mov_b32 m0, %8\n\t"
        "v_writelane_b32 %0, %2, m0\n\tv_writelane_b32 %1, %3, m0"
        : "+v"(ballots_lo), "+v"(ballots_hi), "=&s"(made_lo), "=&s"(made_hi)
        : "s"((uint32_t)x), "s"((uint32_t)(x >> 32)), "s"((uint32_t)y), "s"((uint32_t)(y >> 32)), "s"(v)
        : "m0", "scc");
  };
  auto park = [&](uint32_t v, unsigned long long ballot) {
    uint32_t copy_lo, copy_hi;
    asm("s_mov_b32 %2, %4\n\ts_mov_b32 %3, %5\n\ts_mov_b32 m0, %6\n\t"
        "v_writelane_b32 %0, %2, m0\n\tv_writelane_b32 %1, %3, m0"
        : "+v"(ballots_lo), "+v"(ballots_hi), "=&s"(copy_lo), "=&s"(copy_hi)
        : "s"((uint32_t)ballot), "s"((uint32_t)(ballot >> 32)), "s"(v)
        : "m0");
  };
  static_assert(CULL_MAX_VIEWS <= 64, "one lane per view");
  for (uint32_t v = 0; v < a.views; ++v) {
    const uint32_t req = required[v];
    // evaluated for the whole wave without a branch: the plane loads then issue
    // together with the gate word, ahead of any wait
    const bool inside = sphere_visible(paired + v * 24u, c.x, c.y, c.z, radius);
    if (!TIES && !HZB) {
      if (req != flags_of) {  // wave-uniform
        carried_mask = gate_mask & __ballot((cf & req) == req);
        flags_of = req;
      }
      park_and(v, carried_mask, __ballot(inside) | unbounded_mask);
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
    park(v, __ballot(pass));
  }
  if (lane_id() < a.views)
    s.mask[lane_id() * CULL_SLOTS + slot] = ((unsigned long long)ballots_hi << 32) | ballots_lo;
  return (bucket & 0xfu) | (tv ? 0x10u : 0u);
}

// The block's statistics, tie counts and caster box go to their global accumulators.
// (Every LDS accumulator is set back to zero by the thread that flushes it: a block
// that walks several segments uses the set again two segments later.)
template <bool TIES>
__device__ __forceinline__ void segment_flush_statistics(const CullArgs &a, CullShared &s) {
  const uint32_t t = threadIdx.x;
  if (t == 0) {
    uint32_t *shard = a.counters + stat_shard();
    const int which[5] = {CNT_WITHOUT_BOUNDS, CNT_CULLED_CAMERA, CNT_CAMERA_OPAQUE, CNT_TRANSMISSION,
                          CNT_OCCLUDED};
#pragma unroll
    for (int k = 0; k < 5; ++k) {
      const uint32_t v = s.stat[k];
      if (v) {
        atomicAdd(&shard[which[k]], v);
        s.stat[k] = 0u;
      }
    }
  }
  if (TIES)
    for (uint32_t r = t; r < a.views; r += blockDim.x) {
      const uint32_t v = s.ties[r];
      if (v) {
        atomicAdd(&a.ties[(blockIdx.x % TIE_SHARDS) * CULL_MAX_VIEWS + r], (unsigned long long)v);
        s.ties[r] = 0u;
      }
    }
  if (a.caster_bounds != nullptr && t < 6u) {
    const uint32_t v = s.box[t];
    if (v) {
      atomicMax(&a.caster_bounds[t], v);
      s.box[t] = 0u;
    }
  }
}

// Ends a segment: stores the survivor counts of every row (one column of the
// count table), the segment's record and the statistics. Slot k, lane j of the
// record is candidate `origin + 64 k + j` (`origin` may lie before the level's
// first candidate: such lanes survive nowhere).
template <bool TIES>
__device__ __forceinline__ void segment_finish(const CullArgs &a, CullShared &s, uint64_t seg,
                                               uint32_t origin) {
  const uint32_t t = threadIdx.x;
  __syncthreads();
  // sixteen lanes per row, one per slot: the slot's survivors of the row (its
  // view's mask cut to the row's bucket), summed over the slots
  static_assert(CULL_SLOTS == 16, "row reduction below");
  for (uint32_t r0 = 0; r0 < s.rows; r0 += CULL_BLOCK / CULL_SLOTS) {
    const uint32_t slot = t % CULL_SLOTS;
    const uint32_t r = r0 + t / CULL_SLOTS;
    const bool live = r < s.rows;
    uint32_t c = 0u;
    if (live)
      c = r < a.views * 4u
              ? (uint32_t)__popcll(s.mask[(r >> 2) * CULL_SLOTS + slot] & s.inb[slot * 4u + (r & 3u)])
              : (uint32_t)__popcll(s.tmask[slot]);
#pragma unroll
    for (uint32_t d = CULL_SLOTS / 2; d != 0u; d >>= 1) c += __shfl_xor(c, d, CULL_SLOTS);
    if (live && slot == 0u) a.seg_counts[(uint64_t)r * a.seg_stride + seg] = c;
  }
  {
    const uint32_t words = seg_record_words(a.views);
    unsigned long long *record = a.seg_masks + seg * words;
    for (uint32_t k = t; k < words; k += CULL_BLOCK) record[k] = s.mask[k];
  }
  segment_flush_statistics<TIES>(a, s);
  if (threadIdx.x == 0) a.seg_first[seg] = origin;
}

// A segment without candidates still owns a column of the count table and a
// record (nobody survives).
__device__ __forceinline__ void segment_empty(const CullArgs &a, uint64_t seg) {
  for (uint32_t r = threadIdx.x; r < a.views * 4u + 1u; r += blockDim.x)
    a.seg_counts[(uint64_t)r * a.seg_stride + seg] = 0u;
  const uint32_t words = seg_record_words(a.views);
  unsigned long long *record = a.seg_masks + seg * words;
  for (uint32_t k = threadIdx.x; k < words; k += blockDim.x) record[k] = 0ull;
  if (threadIdx.x == 0) a.seg_first[seg] = 0u;
}

// ---------------------------------------------------------------------------
// Staged compaction (views <= STAGED_MAX_VIEWS). Instead of leaving its masks to
// the gather kernel, the classify block expands them itself when its segment
// ends, from LDS, while the other blocks of the CU keep the memory system busy:
// for every view the segment's survivors go, as 16-bit positions inside the
// segment, to the segment's own stretch of a staging list - bucket after bucket,
// ascending candidate inside a bucket, i.e. in the order of the final list. What
// is left for the launch behind the scan (k_place) is to move runs: where a
// bucket of a view starts depends on the survivors of ALL segments in the buckets
// before it (vk_gpu_draw_prefix, deferred.slang:354-377), so no block can know
// a final position while the classify launch is still running.
//
// In this mode CullArgs::seg_masks is one allocation of
//   [seg_stride][CULL_SLOTS]           64-bit masks of the segments' visible blend rows
//   [views][seg_stride][CULL_TILE]     16-bit staging lists
// ---------------------------------------------------------------------------
__host__ __device__ inline size_t staged_table_words(uint64_t seg_stride, uint32_t views) {
  return (size_t)seg_stride * CULL_SLOTS + ((size_t)views * seg_stride * CULL_TILE * sizeof(uint16_t) + 7u) / 8u;
}
__host__ __device__ inline uint16_t *staged_lists(unsigned long long *seg_masks, uint64_t seg_stride) {
  return reinterpret_cast<uint16_t *>(seg_masks + seg_stride * CULL_SLOTS);
}
__host__ __device__ inline const uint16_t *staged_lists(const unsigned long long *seg_masks, uint64_t seg_stride) {
  return reinterpret_cast<const uint16_t *>(seg_masks + seg_stride * CULL_SLOTS);
}

// A value the whole wave holds alike (read from LDS at a wave-uniform address),
// as a scalar.
__device__ __forceinline__ unsigned long long uniform64(unsigned long long v) {
  const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)v);
  const uint32_t hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(v >> 32));
  return ((unsigned long long)hi << 32) | lo;
}

__device__ __forceinline__ void segment_finish_staged(const CullArgs &a, CullShared &s, uint64_t seg,
                                                      uint32_t origin) {
  static_assert(CULL_SLOTS == 16 && CULL_BLOCK == 256, "lane layouts below");
  const uint32_t t = threadIdx.x;
  const uint32_t lane = t & 63u, wave = t >> 6;
  __syncthreads();  // every wave's ballots are in the tables
  // Wave w takes views w, w + 4, ...: the 64 lanes stand for the view's (bucket,
  // slot) pairs, bucket-major - the order of the list - so one wave scan of the
  // pairs' survivor counts gives every pair its place inside the segment's stretch,
  // and the sums over a bucket's sixteen pairs are the segment's row counts.
  for (uint32_t v = wave; v < a.views; v += CULL_BLOCK / 64u) {  // wave-uniform
    const uint32_t slot = lane & 15u, bucket = lane >> 4;
    const uint32_t count = (uint32_t)__popcll(s.mask[v * CULL_SLOTS + slot] & s.inb[slot * 4u + bucket]);
    const uint32_t upto = wave_inclusive_scan(count);
    s.pair[v * 64u + lane] = (uint16_t)(upto - count);
    const uint32_t before = (uint32_t)__shfl_up((int)upto, 16, 64);  // the buckets before this one
    if (slot == 15u)
      a.seg_counts[(uint64_t)(v * 4u + bucket) * a.seg_stride + seg] = upto - (bucket ? before : 0u);
  }
  if (wave == (a.views & 3u)) {  // the transparent row; the blend-row masks are what k_place still reads
    const unsigned long long blend = lane < (uint32_t)CULL_SLOTS ? s.tmask[lane] : 0ull;
    const uint32_t rows = wave_inclusive_scan((uint32_t)__popcll(blend));
    if (lane < (uint32_t)CULL_SLOTS) a.seg_masks[seg * CULL_SLOTS + lane] = blend;
    if (lane == 63u) {
      a.seg_counts[(uint64_t)(a.views * 4u) * a.seg_stride + seg] = rows;
      a.seg_first[seg] = origin;
    }
  }
  __syncthreads();  // pair bases
  // Every wave expands the slots it classified (slot = round * 4 + wave), one lane
  // per candidate: a survivor's place is its pair's base plus the survivors of the
  // same bucket below it in the slot. Everything a slot needs from LDS is requested
  // in one go (clamped indices instead of branches), so that a slot costs one LDS
  // round trip and not one per view.
  uint16_t *lists = staged_lists(a.seg_masks, a.seg_stride) + seg * CULL_TILE;
  const uint64_t list_stride = a.seg_stride * CULL_TILE;
  const uint32_t bit_lo = lane < 32u ? 1u << lane : 0u;
  const uint32_t bit_hi = lane < 32u ? 0u : 1u << (lane - 32u);
  const uint32_t last_view = a.views - 1u;
#pragma unroll
  for (uint32_t r = 0; r < (uint32_t)CULL_ROUNDS; ++r) {
    const uint32_t slot = r * (CULL_BLOCK / 64u) + wave;
    unsigned long long m[STAGED_MAX_VIEWS];
#pragma unroll
    for (uint32_t v = 0; v < STAGED_MAX_VIEWS; ++v) m[v] = s.mask[min(v, last_view) * CULL_SLOTS + slot];
    const unsigned long long b0 = uniform64(s.inb[slot * 4u + 0u]), b1 = uniform64(s.inb[slot * 4u + 1u]);
    const unsigned long long b2 = uniform64(s.inb[slot * 4u + 2u]), b3 = uniform64(s.inb[slot * 4u + 3u]);
    if ((b0 | b1 | b2 | b3) == 0ull) continue;  // wave-uniform: a round the segment did not run
    auto mine = [&](unsigned long long x) { return (((uint32_t)x & bit_lo) | ((uint32_t)(x >> 32) & bit_hi)) != 0u; };
    // (a lane in none of the four buckets survives nowhere; what it picks here is never used)
    const bool in1 = mine(b1), in2 = mine(b2), in3 = mine(b3);
    const uint32_t bucket = in1 ? 1u : (in2 ? 2u : (in3 ? 3u : 0u));
    const unsigned long long same = in1 ? b1 : (in2 ? b2 : (in3 ? b3 : b0));
    const uint32_t same_lo = (uint32_t)same, same_hi = (uint32_t)(same >> 32);
    uint32_t base[STAGED_MAX_VIEWS];
#pragma unroll
    for (uint32_t v = 0; v < STAGED_MAX_VIEWS; ++v) base[v] = s.pair[min(v, last_view) * 64u + bucket * 16u + slot];
    const uint32_t position = slot * 64u + lane;  // of the candidate inside the segment
#pragma unroll
    for (uint32_t v = 0; v < STAGED_MAX_VIEWS; ++v) {
      if (v >= a.views) break;  // wave-uniform
      const unsigned long long mv = uniform64(m[v]);
      if (mv != 0ull && mine(mv)) {
        const uint32_t below = __builtin_amdgcn_mbcnt_hi((uint32_t)(mv >> 32) & same_hi,
                                                         __builtin_amdgcn_mbcnt_lo((uint32_t)mv & same_lo, 0u));
        *at32(lists + v * list_stride, base[v] + below) = (uint16_t)position;
      }
    }
  }
  segment_flush_statistics<false>(a, s);
}

__device__ __forceinline__ void segment_empty_staged(const CullArgs &a, uint64_t seg) {
  for (uint32_t r = threadIdx.x; r < a.views * 4u + 1u; r += blockDim.x)
    a.seg_counts[(uint64_t)r * a.seg_stride + seg] = 0u;
  if (threadIdx.x < (uint32_t)CULL_SLOTS) a.seg_masks[seg * CULL_SLOTS + threadIdx.x] = 0ull;
  if (threadIdx.x == 0) a.seg_first[seg] = 0u;
}

// Stand-alone classify: one block per segment of 1024 candidates, four rounds of
// 256. With the identity candidate->node mapping the matrices arrive as
// coalesced 16-byte loads staged through LDS (thread t then reads its own four
// columns); otherwise each thread gathers its node's matrix.
// STREAM: streaming loads for what is read once (see load_once); the matrices
// gathered through a candidate -> node mapping are shared by a node's rows and
// keep the default policy.
template <bool IDENTITY, bool TIES, bool BOUNDS_OUT, bool HZB = false, bool STREAM = false>
__global__ __launch_bounds__(CULL_BLOCK) void k_classify(CullArgs a) {
  __shared__ float4 s_mat[IDENTITY ? CULL_BLOCK * MAT_LDS_STRIDE : 1];
  CullShared s = cull_shared_init(a.views);
  const uint32_t t = threadIdx.x;
  const uint64_t seg = blockIdx.x;
  const uint64_t first = seg * CULL_TILE;
  const uint64_t last = a.n - 1u;  // n > 0: the grid has a block per segment
  __syncthreads();  // accumulators cleared

  // all rounds' small inputs are requested up front (clamped, unconditional)
  float4 ls_r[CULL_ROUNDS];
  uint32_t cf_r[CULL_ROUNDS];
#pragma unroll
  for (int round = 0; round < CULL_ROUNDS; ++round) {
    const uint64_t i = min(first + (uint64_t)round * CULL_BLOCK + t, last);
    const f32x4 ls = load_once<STREAM>(reinterpret_cast<const f32x4 *>(a.sphere) + i);
    ls_r[round] = make_float4(ls.x, ls.y, ls.z, ls.w);
    cf_r[round] = load_once<STREAM>(a.cflags + i);
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
      // (named values: as an array the four loads are kept in scratch memory
      // between the load and the LDS store)
      const f32x4 *wm = reinterpret_cast<const f32x4 *>(a.world);
      const f32x4 u0 = load_once<STREAM>(wm + min(base * 4u + t, last4));
      const f32x4 u1 = load_once<STREAM>(wm + min(base * 4u + t + 256u, last4));
      const f32x4 u2 = load_once<STREAM>(wm + min(base * 4u + t + 512u, last4));
      const f32x4 u3 = load_once<STREAM>(wm + min(base * 4u + t + 768u, last4));
      const float4 v0 = make_float4(u0.x, u0.y, u0.z, u0.w), v1 = make_float4(u1.x, u1.y, u1.z, u1.w);
      const float4 v2 = make_float4(u2.x, u2.y, u2.z, u2.w), v3 = make_float4(u3.x, u3.y, u3.z, u3.w);
      __syncthreads();  // the previous round's readers of s_mat are done
      s_mat[(t >> 2) * MAT_LDS_STRIDE + (t & 3u)] = v0;
      s_mat[((t + 256u) >> 2) * MAT_LDS_STRIDE + (t & 3u)] = v1;
      s_mat[((t + 512u) >> 2) * MAT_LDS_STRIDE + (t & 3u)] = v2;
      s_mat[((t + 768u) >> 2) * MAT_LDS_STRIDE + (t & 3u)] = v3;
      __syncthreads();
      const float4 *p = s_mat + t * MAT_LDS_STRIDE;
      m.c[0] = p[0];
      m.c[1] = p[1];
      m.c[2] = p[2];
      m.c[3] = p[3];
    } else {
      const float4 *p = a.world + (uint64_t)load_once<STREAM>(a.node + min(i, last)) * 4u;
      m.c[0] = p[0];
      m.c[1] = p[1];
      m.c[2] = p[2];
      m.c[3] = p[3];
    }
    classify_candidate<TIES, BOUNDS_OUT, HZB>(a, s, (uint32_t)round, i, valid, m, ls_r[round], cf);
  }
  segment_finish<TIES>(a, s, seg, (uint32_t)first);
}

// ---------------------------------------------------------------------------
// Block-wide exclusive scan helpers (1024 threads, 16 waves).
// ---------------------------------------------------------------------------
constexpr int SCAN_BLOCK = 1024;

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
// THREADS: 1024, or 256 for short rows - a block of four waves of this kernel's 46
// registers finds room beside the next frame's classify blocks on every SIMD, a
// block of sixteen does not (see tail_on_second_stream in vkrhip.hip).
template <int THREADS>
__global__ __launch_bounds__(THREADS) void k_scan(ScanArgs a) {
  __shared__ uint32_t s_wave[THREADS / 64];
  const uint32_t r = blockIdx.x;
  const uint4 *row = reinterpret_cast<const uint4 *>(a.seg_counts + (uint64_t)r * a.seg_stride);
  uint4 *out = reinterpret_cast<uint4 *>(a.seg_offsets + (uint64_t)r * a.seg_stride);
  const uint64_t quads = (a.segments + 3u) / 4u;
  const uint32_t w = threadIdx.x >> 6;
  const uint32_t lane = lane_id();
  const uint64_t wave_quads = ((quads + THREADS / 64 - 1) / (THREADS / 64) + 63u) & ~(uint64_t)63u;
  const uint64_t qb = min((uint64_t)w * wave_quads, quads);
  const uint64_t qe = min(qb + wave_quads, quads);

  uint32_t mine = 0u;
  for (uint64_t q = qb + lane; q < qe; q += 64u) mine += sum4(load_counts4(row, q, a.segments));
  const uint32_t wave_total = (uint32_t)__shfl((int)wave_inclusive_scan(mine), 63, 64);
  if (lane == 0) s_wave[w] = wave_total;
  __syncthreads();
  uint32_t carry = 0u, total = 0u;
  for (uint32_t k = 0; k < THREADS / 64; ++k) {
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
    for (uint32_t k = threadIdx.x; k < a.clear_words; k += THREADS) a.clear[k] = 0u;
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
  const unsigned long long *seg_masks;  // [segment][seg_record_words(views)]; k_place: see segment_finish_staged
  const uint32_t *seg_counts;   // [row][seg_stride] (k_place only)
  const uint32_t *seg_offsets;  // [row][seg_stride], from k_scan
  const uint32_t *seg_first;
  const uint32_t *row_totals;
  const uint64_t *tkey;
  const uint32_t *remap;    // candidate -> packed row index when a publication is set, else null
  CompactionState *states;  // written by block 0
  uint32_t *visible;        // [view][visible_stride]
  uint64_t *keys;           // compact sort keys of the visible blend rows: depth bits << 32 | place
  uint32_t *tsrc;           // the candidate of every visible blend row, by place; null: the
                            //   compact key carries the candidate itself instead of the place
  const float4 *world;      // with tworld: the rows' world matrices are set aside by place
  const uint32_t *node;     //   (candidate -> node, null = identity)
  float4 *tworld;           // null: the emission reads world[] in place
  uint64_t segments;
  uint64_t seg_stride;
  uint64_t visible_stride;
  uint64_t transparent_capacity;
  uint64_t index_base;
  uint64_t candidates;      // n > 0
  uint32_t views;
  uint32_t partition_capacity;
  // k_gather: waves per segment. 1: a wave walks all views of its segment, four per
  // request; `views`: one wave per (segment, view) - scenes of few segments, where the
  // launch is as long as one wave's chain of views (256 segments, five views: 27 us).
  uint32_t view_parts;
};

constexpr int GATHER_BLOCK = 256;  // four waves, one segment each
constexpr uint32_t GATHER_SPARSE_LIMIT = 384;  // survivors of a (segment, view) written bit by bit

__device__ __forceinline__ unsigned long long shfl64(unsigned long long v, uint32_t src) {
  const uint32_t lo = (uint32_t)__shfl((int)(uint32_t)v, (int)src, 64);
  const uint32_t hi = (uint32_t)__shfl((int)(uint32_t)(v >> 32), (int)src, 64);
  return ((unsigned long long)hi << 32) | lo;
}
__device__ __forceinline__ unsigned long long readlane64(unsigned long long v, uint32_t uniform_src) {
  const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)v, (int)uniform_src);
  const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(v >> 32), (int)uniform_src);
  return ((unsigned long long)hi << 32) | lo;
}

// The visible blend rows of SEGS consecutive segments (one wave), one lane per ROW:
// row j is the j-th set bit of the segments' slot masks (`blend`: lane 16 k + s
// holds slot s of segment k; candidate order, which is the order of the places),
// so that everything a row needs - its key and, when the sort runs on a second
// stream, its world matrix - is requested in one round for all of them.
// `origin_of(k)`: the candidate of segment k's slot 0, lane 0; `first_of(k)`: the
// place of segment k's first row.
template <uint32_t SEGS, typename OriginOf, typename FirstOf>
__device__ __forceinline__ void gather_blend_rows(const GatherArgs &a, unsigned long long blend,
                                                  OriginOf origin_of, FirstOf first_of) {
  static_assert(SEGS == 1u || SEGS == 2u || SEGS == 4u, "one lane per slot mask");
  constexpr uint32_t MASKS = SEGS * (uint32_t)CULL_SLOTS;
  const uint32_t lane = lane_id();
  const uint32_t last_candidate = (uint32_t)(a.candidates - 1u);
  const uint32_t in_slot = (uint32_t)__popcll(blend);
  const uint32_t upto = wave_inclusive_scan(in_slot);  // lanes 0..MASKS-1: rows of masks 0..k
  const uint32_t rows = (uint32_t)__builtin_amdgcn_readlane((int)upto, 63);
  for (uint32_t j0 = 0; j0 < rows; j0 += 64u) {  // wave-uniform; one trip unless > 64 rows
    const uint32_t j = j0 + lane;
    // the mask of row j: how many masks end at or before it (their row counts ascend)
    uint32_t k = 0u;
#pragma unroll
    for (uint32_t step = MASKS / 2u; step != 0u; step >>= 1) {
      const uint32_t ends = (uint32_t)__shfl((int)upto, (int)(k + step - 1u), 64);
      if (ends <= j) k += step;
    }
    const unsigned long long m = shfl64(blend, k);
    uint32_t r = j - (uint32_t)__shfl((int)(upto - in_slot), (int)k, 64);  // row j is the r-th set bit of m
    // the r-th set bit, by halving
    uint32_t w = (uint32_t)m, bit = 0u;
    uint32_t c = (uint32_t)__popc(w);
    if (r >= c) { r -= c; w = (uint32_t)(m >> 32); bit = 32u; }
    c = (uint32_t)__popc(w & 0xffffu);
    if (r >= c) { r -= c; w >>= 16; bit += 16u; }
    c = (uint32_t)__popc(w & 0xffu);
    if (r >= c) { r -= c; w >>= 8; bit += 8u; }
    c = (uint32_t)__popc(w & 0xfu);
    if (r >= c) { r -= c; w >>= 4; bit += 4u; }
    c = (uint32_t)__popc(w & 0x3u);
    if (r >= c) { r -= c; w >>= 2; bit += 2u; }
    if (r >= (w & 1u)) bit += 1u;
    // rows of the segments before this row's own
    const uint32_t before = (uint32_t)__shfl((int)(upto - in_slot), (int)(k & ~((uint32_t)CULL_SLOTS - 1u)), 64);
    const uint64_t pos = (uint64_t)first_of(k / (uint32_t)CULL_SLOTS) + (j - before);
    if (j < rows && pos < a.transparent_capacity) {
      const uint32_t candidate =
          min(origin_of(k / (uint32_t)CULL_SLOTS) + (k % (uint32_t)CULL_SLOTS) * 64u + bit, last_candidate);
      const uint64_t key = a.tkey[candidate];
      if (a.tworld) {
        const uint64_t nd = a.node ? a.node[candidate] : candidate;
        // (named values, not an array: an array here is given LDS space addressed
        // through the dispatch packet, a read of host memory in every wave)
        const float4 c0 = a.world[nd * 4u], c1 = a.world[nd * 4u + 1u];
        const float4 c2 = a.world[nd * 4u + 2u], c3 = a.world[nd * 4u + 3u];
        a.tworld[pos * 4u] = c0;
        a.tworld[pos * 4u + 1u] = c1;
        a.tworld[pos * 4u + 2u] = c2;
        a.tworld[pos * 4u + 3u] = c3;
      }
      // (either low word ascends with the candidate, which is all the sort asks of it)
      if (a.tsrc) {
        a.keys[pos] = (key & 0xffffffff00000000ull) | pos;
        a.tsrc[pos] = candidate;
      } else {
        a.keys[pos] = (key & 0xffffffff00000000ull) | candidate;
      }
    }
  }
}

// One wave per segment: turns the segment's survivor masks into its stretch of
// every view's visible list. For one view the lanes stand for the 64 (bucket,
// slot) pairs, bucket-major - the order of the list - so a wave scan of the
// pairs' survivor counts gives every pair its position behind the segment's
// scanned offset. Few survivors (the usual case) are then written pair by pair,
// each lane walking the set bits of its own mask; many are written slot by slot
// with one lane per candidate, 64 consecutive entries per store. Everything a
// step needs is requested a step ahead.
__global__ __launch_bounds__(GATHER_BLOCK) void k_gather(GatherArgs a) {
  static_assert(CULL_SLOTS == 16, "lane layouts below");
  const uint32_t t = threadIdx.x;
  const uint32_t lane = lane_id();
  const uint32_t wave = t >> 6;
  const uint32_t vrows = a.views * 4u;
  if (blockIdx.x == 0)
    for (uint32_t v = t; v < a.views; v += GATHER_BLOCK)
      a.states[v] = prefix_state(a.row_totals + v * 4u, a.partition_capacity, a.visible_stride);

  const uint32_t parts = a.view_parts;  // >= 1
  const uint64_t wave_index = (uint64_t)blockIdx.x * (GATHER_BLOCK / 64) + wave;
  const uint64_t seg = wave_index / parts;
  const uint32_t part = (uint32_t)(wave_index % parts);
  if (seg >= a.segments) return;  // wave-uniform; no block-wide barrier below
  const unsigned long long *record = a.seg_masks + seg * seg_record_words(a.views);
  // (the same address, known to be the same for the whole wave: scalar loads)
  const unsigned long long *uniform_record =
      a.seg_masks + (((uint64_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(seg >> 32)) << 32) |
                     (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)seg)) *
                        seg_record_words(a.views);
  const uint32_t slot = lane & 15u, bucket = lane >> 4;

  // What the wave needs for four views, requested in one go: lane -> (view v0 +
  // lane / 16, slot lane % 16) for the masks; lane j < 16 -> row 4 v0 + j for the
  // row's scanned offset and the four bucket totals of the row's view.
  struct Chunk {
    unsigned long long masks;
    uint32_t row, offset, total[4];
  };
  auto request = [&](uint32_t v0) {
    Chunk c;
    c.masks = record[min(v0 + bucket, a.views - 1u) * CULL_SLOTS + slot];
    c.row = min(v0 * 4u + (lane & 15u), vrows - 1u);
    c.offset = a.seg_offsets[(uint64_t)c.row * a.seg_stride + seg];
#pragma unroll
    for (uint32_t b = 0; b < 4u; ++b) c.total[b] = a.row_totals[(c.row & ~3u) + b];
    return c;
  };
  const uint32_t view_step = parts > 1u ? parts : 4u;
  Chunk cur = request(parts > 1u ? part : 0u);
  const uint32_t origin = a.seg_first[seg];
  const unsigned long long in_bucket = record[CULL_SLOTS * a.views + slot * 4u + bucket];
  // (a set bit always stands for a candidate; the clamp keeps the loads inside the
  // arrays whatever a record holds)
  const uint32_t last_candidate = (uint32_t)(a.candidates - 1u);
  auto source_row = [&](uint32_t candidate) -> uint32_t {
    return a.remap ? a.remap[min(candidate, last_candidate)] : (uint32_t)(a.index_base + candidate);
  };

  if (part == 0u)  // wave-uniform
    gather_blend_rows<1>(a, lane < (uint32_t)CULL_SLOTS ? record[CULL_SLOTS * (a.views + 4u) + lane] : 0ull,
                         [&](uint32_t) { return origin; },
                         [&](uint32_t) { return a.seg_offsets[(uint64_t)vrows * a.seg_stride + seg]; });

  for (uint32_t v0 = parts > 1u ? part : 0u; v0 < a.views; v0 += view_step) {
    const Chunk c = cur;
    if (v0 + view_step < a.views) cur = request(v0 + view_step);  // in flight while this chunk is written
    // where the row's bucket starts inside its view's list and how much of it is
    // kept (as prefix_state lays the buckets out)
    uint32_t row_start = 0u, row_kept = 0u;
#pragma unroll
    for (uint32_t b = 0; b < 4u; ++b) {
      const uint32_t room = bucket_room(c.total[b], a.partition_capacity, a.visible_stride, row_start);
      if (b == (c.row & 3u)) row_kept = room;
      if (b < (c.row & 3u)) row_start += room;
    }

    const uint32_t in_chunk = parts > 1u ? 1u : min(4u, a.views - v0);
    for (uint32_t j = 0; j < in_chunk; ++j) {  // wave-uniform
      const unsigned long long mine = shfl64(c.masks, j * 16u + slot) & in_bucket;
      const uint32_t count = (uint32_t)__popcll(mine);
      const uint32_t incl = wave_inclusive_scan(count);
      const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
      if (total == 0u) continue;
      const uint32_t excl = incl - count;
      const uint32_t bucket_first = (uint32_t)__shfl((int)excl, (int)(lane & 48u), 64);
      // position inside the bucket (before the clamp), start of the bucket, entries kept
      uint32_t p = (uint32_t)__shfl((int)c.offset, (int)(j * 4u + bucket), 64) + (excl - bucket_first);
      const uint32_t start = (uint32_t)__shfl((int)row_start, (int)(j * 4u + bucket), 64);
      const uint32_t kept = (uint32_t)__shfl((int)row_kept, (int)(j * 4u + bucket), 64);
      uint32_t *list = a.visible + (uint64_t)(v0 + j) * a.visible_stride;
      if (total <= GATHER_SPARSE_LIMIT) {  // wave-uniform
        unsigned long long left = mine;
        while (__ballot(left != 0ull) != 0ull) {
          if (left != 0ull) {
            const uint32_t bit = (uint32_t)__ffsll((long long)left) - 1u;
            left &= left - 1ull;
            if (p < kept) list[start + p] = source_row(origin + slot * 64u + bit);
            ++p;
          }
        }
      } else {
        // one lane per candidate, slot by slot; the slot's masks are the same for the
        // whole wave (scalar values), so a lane's place among its bucket's survivors
        // is a masked bit count
        uint32_t base[4], limit[4];
#pragma unroll
        for (uint32_t b = 0; b < 4u; ++b) {
          const uint32_t row_lane = j * 4u + b;
          const uint32_t first = (uint32_t)__builtin_amdgcn_readlane((int)row_start, (int)row_lane);
          base[b] = first + (uint32_t)__builtin_amdgcn_readlane((int)c.offset, (int)row_lane);
          limit[b] = first + (uint32_t)__builtin_amdgcn_readlane((int)row_kept, (int)row_lane);
        }
        // (constant address space: nothing in this kernel writes the records)
        const uniform_ptr<unsigned long long> slot_buckets = as_uniform(uniform_record + CULL_SLOTS * a.views);
#pragma unroll 2
        for (uint32_t k = 0; k < (uint32_t)CULL_SLOTS; ++k) {
          const unsigned long long m = readlane64(c.masks, j * 16u + k);
          if (m == 0ull) continue;  // wave-uniform
          uint32_t pos = 0u, end = 0u;
#pragma unroll
          for (uint32_t b = 0; b < 4u; ++b) {
            const unsigned long long mb = m & slot_buckets[k * 4u + b];
            const uint32_t below = __builtin_amdgcn_mbcnt_hi(
                (uint32_t)(mb >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mb, 0u));
            if ((mb >> lane) & 1ull) {
              pos = base[b] + below;
              end = limit[b];
            }
            base[b] += (uint32_t)__popcll(mb);
          }
          if (pos < end) list[pos] = source_row(origin + k * 64u + lane);
        }
      }
    }
  }
}

constexpr uint32_t GATHER_MANY_VIEWS = 8;

// k_gather for many views (more than GATHER_MANY_VIEWS): four views per step. Lane
// 16 j + s stands for slot s of the step's view j, and per state bucket a scan
// inside the 16-lane rows gives every (view, bucket, slot) its position behind the
// segment's scanned offset, for four views at the price of one. Few survivors are
// then written pair by pair, each lane walking the set bits of its own four bucket
// masks one after the other; a view with many goes slot by slot as in k_gather.
// With 64 views over 16M rows the launch takes 205 us where k_gather takes 289 us;
// with one or five views it loses (a view's 64 pairs on 64 lanes walk their bits in
// fewer steps than on 16 lanes: 16M-node hierarchy, five views, frame 565 -> 572
// us), hence two kernels.
__global__ __launch_bounds__(GATHER_BLOCK) __attribute__((amdgpu_waves_per_eu(8, 8))) void k_gather_many(GatherArgs a) {
  static_assert(CULL_SLOTS == 16, "lane layouts below");
  const uint32_t t = threadIdx.x;
  const uint32_t lane = lane_id();
  const uint32_t wave = t >> 6;
  const uint32_t vrows = a.views * 4u;
  if (blockIdx.x == 0)
    for (uint32_t v = t; v < a.views; v += GATHER_BLOCK)
      a.states[v] = prefix_state(a.row_totals + v * 4u, a.partition_capacity, a.visible_stride);

  const uint64_t seg = (uint64_t)blockIdx.x * (GATHER_BLOCK / 64) + wave;
  if (seg >= a.segments) return;  // wave-uniform; no block-wide barrier below
  const unsigned long long *record = a.seg_masks + seg * seg_record_words(a.views);
  // (the same address, known to be the same for the whole wave: scalar loads)
  const unsigned long long *uniform_record =
      a.seg_masks + (((uint64_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(seg >> 32)) << 32) |
                     (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)seg)) *
                        seg_record_words(a.views);
  const uint32_t slot = lane & 15u, bucket = lane >> 4;

  // What the wave needs for four views, requested in one go: lane -> (view v0 +
  // lane / 16, slot lane % 16) for the masks; lane j < 16 -> row 4 v0 + j for the
  // row's scanned offset and the four bucket totals of the row's view.
  struct Chunk {
    unsigned long long masks;
    uint32_t row, offset, total[4];
  };
  auto request = [&](uint32_t v0) {
    Chunk c;
    c.masks = record[min(v0 + bucket, a.views - 1u) * CULL_SLOTS + slot];
    c.row = min(v0 * 4u + (lane & 15u), vrows - 1u);
    c.offset = a.seg_offsets[(uint64_t)c.row * a.seg_stride + seg];
#pragma unroll
    for (uint32_t b = 0; b < 4u; ++b) c.total[b] = a.row_totals[(c.row & ~3u) + b];
    return c;
  };
  Chunk cur = request(0u);
  const uint32_t origin = a.seg_first[seg];
  // the four state-bucket masks of this lane's slot (lanes 16 j + s all hold slot s)
  const unsigned long long *slot_in_bucket = record + CULL_SLOTS * a.views + slot * 4u;
  const unsigned long long inb0 = slot_in_bucket[0], inb1 = slot_in_bucket[1];
  const unsigned long long inb2 = slot_in_bucket[2], inb3 = slot_in_bucket[3];
  // (a set bit always stands for a candidate; the clamp keeps the loads inside the
  // arrays whatever a record holds)
  const uint32_t last_candidate = (uint32_t)(a.candidates - 1u);
  auto source_row = [&](uint32_t candidate) -> uint32_t {
    return a.remap ? a.remap[min(candidate, last_candidate)] : (uint32_t)(a.index_base + candidate);
  };

  gather_blend_rows<1>(a, lane < (uint32_t)CULL_SLOTS ? record[CULL_SLOTS * (a.views + 4u) + lane] : 0ull,
                       [&](uint32_t) { return origin; },
                       [&](uint32_t) { return a.seg_offsets[(uint64_t)vrows * a.seg_stride + seg]; });

  for (uint32_t v0 = 0; v0 < a.views; v0 += 4u) {
    const Chunk c = cur;
    if (v0 + 4u < a.views) cur = request(v0 + 4u);  // in flight while this chunk is written
    // where the row's bucket starts inside its view's list and how much of it is
    // kept (as prefix_state lays the buckets out)
    uint32_t row_start = 0u, row_kept = 0u;
#pragma unroll
    for (uint32_t b = 0; b < 4u; ++b) {
      const uint32_t room = bucket_room(c.total[b], a.partition_capacity, a.visible_stride, row_start);
      if (b == (c.row & 3u)) row_kept = room;
      if (b < (c.row & 3u)) row_start += room;
    }

    // Four views at a time: lane 16 j + s stands for (view v0 + j, slot s). Per state
    // bucket the slots' survivor counts are scanned inside the 16-lane row (four
    // shifted adds), which places every (bucket, slot) pair of all four views at once;
    // a view with few survivors (the usual case) is then written pair by pair, each
    // lane walking the set bits of its own masks, bucket after bucket.
    const uint32_t j = bucket;  // the lane's view inside the chunk
    const unsigned long long mine = v0 + j < a.views ? c.masks : 0ull;
    const unsigned long long mb[4] = {mine & inb0, mine & inb1, mine & inb2, mine & inb3};
    uint32_t excl[4], view_total = 0u;
#pragma unroll
    for (uint32_t b = 0; b < 4u; ++b) {
      const uint32_t count = (uint32_t)__popcll(mb[b]);
      uint32_t incl = count;
      incl += dpp_or_zero<0x111, 0xf>(incl);  // row_shr:1
      incl += dpp_or_zero<0x112, 0xf>(incl);  // row_shr:2
      incl += dpp_or_zero<0x114, 0xf>(incl);  // row_shr:4
      incl += dpp_or_zero<0x118, 0xf>(incl);  // row_shr:8
      excl[b] = incl - count;
      view_total += (uint32_t)__shfl((int)incl, (int)(lane | 15u), 64);
    }
    const bool dense = view_total > GATHER_SPARSE_LIMIT;  // the same for the 16 lanes of a view
    const unsigned long long dense_lanes = __ballot(dense);
    {
      uint32_t *list = a.visible + (uint64_t)(v0 + j) * a.visible_stride;
#pragma unroll
      for (uint32_t b = 0; b < 4u; ++b) {
        const uint32_t row_lane = j * 4u + b;
        // position inside the bucket (before the clamp), start of the bucket, entries kept
        uint32_t p = (uint32_t)__shfl((int)c.offset, (int)row_lane, 64) + excl[b];
        const uint32_t start = (uint32_t)__shfl((int)row_start, (int)row_lane, 64);
        const uint32_t kept = (uint32_t)__shfl((int)row_kept, (int)row_lane, 64);
        unsigned long long left = dense ? 0ull : mb[b];
        while (__ballot(left != 0ull) != 0ull) {
          if (left != 0ull) {
            const uint32_t bit = (uint32_t)__ffsll((long long)left) - 1u;
            left &= left - 1ull;
            if (p < kept) list[start + p] = source_row(origin + slot * 64u + bit);
            ++p;
          }
        }
      }
    }

    // views with many survivors: one lane per candidate, slot by slot; the slot's
    // masks are the same for the whole wave (scalar values), so a lane's place among
    // its bucket's survivors is a masked bit count, 64 consecutive entries per store
    for (uint32_t jd = 0; jd < 4u; ++jd) {  // wave-uniform
      if (!((dense_lanes >> (16u * jd)) & 1ull)) continue;
      uint32_t *list = a.visible + (uint64_t)(v0 + jd) * a.visible_stride;
      uint32_t base[4], limit[4];
#pragma unroll
      for (uint32_t b = 0; b < 4u; ++b) {
        const uint32_t row_lane = jd * 4u + b;
        const uint32_t first = (uint32_t)__builtin_amdgcn_readlane((int)row_start, (int)row_lane);
        base[b] = first + (uint32_t)__builtin_amdgcn_readlane((int)c.offset, (int)row_lane);
        limit[b] = first + (uint32_t)__builtin_amdgcn_readlane((int)row_kept, (int)row_lane);
      }
      // (constant address space: nothing in this kernel writes the records)
      const uniform_ptr<unsigned long long> slot_buckets = as_uniform(uniform_record + CULL_SLOTS * a.views);
#pragma unroll 2
      for (uint32_t k = 0; k < (uint32_t)CULL_SLOTS; ++k) {
        const unsigned long long m = readlane64(c.masks, jd * 16u + k);
        if (m == 0ull) continue;  // wave-uniform
        uint32_t pos = 0u, end = 0u;
#pragma unroll
        for (uint32_t b = 0; b < 4u; ++b) {
          const unsigned long long mbk = m & slot_buckets[k * 4u + b];
          const uint32_t below = __builtin_amdgcn_mbcnt_hi(
              (uint32_t)(mbk >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mbk, 0u));
          if ((mbk >> lane) & 1ull) {
            pos = base[b] + below;
            end = limit[b];
          }
          base[b] += (uint32_t)__popcll(mbk);
        }
        if (pos < end) list[pos] = source_row(origin + k * 64u + lane);
      }
    }
  }
}

// What k_place knows about one (segment, view): the entry indices at which the
// stretch's buckets end, and per bucket the distance from an entry's index to its
// place in the view's list and the end of what the list keeps of the bucket.
struct PlaceRun {
  uint32_t *list;
  const uint32_t *remap;
  uint32_t entries, end0, end1, end2;
  uint32_t shift0, shift1, shift2, shift3;
  uint32_t stop0, stop1, stop2, stop3;
  uint32_t first_row;  // source row of the segment's position 0 without a publication
  uint32_t origin, last_candidate;
};
// Entry e of the stretch (the low 16 bits of `position`: the candidate's place
// inside its segment) goes to its bucket's run in the view's list.
// (The run is taken BY VALUE: selects between fields reached through a reference are
// turned into a load from a selected address, which keeps the whole struct - and
// with it everything it was filled from - in scratch memory.)
__device__ __forceinline__ void place_entry(const PlaceRun r, uint32_t e, uint32_t position) {
  const bool in0 = e < r.end0, in1 = e < r.end1, in2 = e < r.end2;
  const uint32_t at = e + (in0 ? r.shift0 : (in1 ? r.shift1 : (in2 ? r.shift2 : r.shift3)));
  const uint32_t stop = in0 ? r.stop0 : (in1 ? r.stop1 : (in2 ? r.stop2 : r.stop3));
  if (e < r.entries && at < stop) {
    position &= 0xffffu;
    r.list[at] = r.remap ? r.remap[min(r.origin + position, r.last_candidate)] : r.first_row + position;
  }
}

// The launch behind the scan when the classify blocks staged their survivors
// themselves (segment_finish_staged): moves every segment's staged runs - per view
// four, one per state bucket, back to back - to where the scanned offsets and the
// bucket totals of the whole frame put them, one lane per entry, and turns the
// 16-bit positions into source rows on the way. A wave takes PLACE_SEGS
// consecutive segments and requests everything that does not depend on a loaded
// value - the tables of all of them and the first 64 staged entries of every
// (segment, view) - in one go: the kernel is bound by the memory round trips a
// wave makes one after the other (three here, whatever the scene), not by bytes
// or instructions. views <= STAGED_MAX_VIEWS, so one lane per count row holds
// everything the wave needs of a segment's tables.
constexpr uint32_t PLACE_SEGS = 4;
__global__ __launch_bounds__(GATHER_BLOCK) void k_place(GatherArgs a) {
  static_assert(STAGED_MAX_VIEWS * 4u + 1u <= 64u, "one lane per count row");
  static_assert(PLACE_SEGS * CULL_SLOTS <= 64u, "one lane per blend-row mask");
  // the first 64 staged entries of every (segment, view) of the block's waves
  __shared__ uint16_t s_head[GATHER_BLOCK / 64][PLACE_SEGS][STAGED_MAX_VIEWS][64];
  const uint32_t t = threadIdx.x;
  const uint32_t lane = lane_id();
  const uint32_t wave = t >> 6;
  const uint32_t vrows = a.views * 4u;
  if (blockIdx.x == 0)
    for (uint32_t v = t; v < a.views; v += GATHER_BLOCK)
      a.states[v] = prefix_state(a.row_totals + v * 4u, a.partition_capacity, a.visible_stride);

  // Wave w of W takes segments w, w + W, w + 2 W, ...: neighbouring segments are
  // neighbours in space and tend to be dense in a view together, and a dense
  // segment is ten times the work of a sparse one.
  const uint64_t waves = (a.segments + PLACE_SEGS - 1u) / PLACE_SEGS;
  const uint64_t seg0 = (uint64_t)blockIdx.x * (GATHER_BLOCK / 64) + wave;
  if (seg0 >= waves) return;  // wave-uniform; no block-wide barrier below
  // segments of this wave (the last ones lack where the count is no multiple)
  const uint32_t held = (uint32_t)((a.segments - seg0 + waves - 1u) / waves);
  // lane j: count row j (lanes beyond the rows repeat the transparent row)
  const uint32_t row = min(lane, vrows);
  const uint16_t *lists = staged_lists(a.seg_masks, a.seg_stride);
  const uint64_t list_stride = a.seg_stride * CULL_TILE;
  const uint32_t last_view = a.views - 1u;
  uint32_t offset[PLACE_SEGS], count[PLACE_SEGS], origin[PLACE_SEGS];
  {
    uint32_t head[PLACE_SEGS][STAGED_MAX_VIEWS];
#pragma unroll
    for (uint32_t k = 0; k < PLACE_SEGS; ++k) {
      const uint64_t sk = seg0 + min(k, held - 1u) * waves;
      offset[k] = a.seg_offsets[(uint64_t)row * a.seg_stride + sk];
      count[k] = a.seg_counts[(uint64_t)row * a.seg_stride + sk];
      origin[k] = a.seg_first[sk];
#pragma unroll
      for (uint32_t v = 0; v < STAGED_MAX_VIEWS; ++v)
        head[k][v] = *at32(lists + min(v, last_view) * list_stride + sk * CULL_TILE, lane);
    }
#pragma unroll
    for (uint32_t k = 0; k < PLACE_SEGS; ++k)
#pragma unroll
      for (uint32_t v = 0; v < STAGED_MAX_VIEWS; ++v) s_head[wave][k][v][lane] = (uint16_t)head[k][v];
  }
  const unsigned long long blend =
      lane / (uint32_t)CULL_SLOTS < held
          ? a.seg_masks[(seg0 + lane / (uint32_t)CULL_SLOTS * waves) * CULL_SLOTS + lane % (uint32_t)CULL_SLOTS]
          : 0ull;
  const uint32_t view_row = min(row, vrows - 1u) & ~3u;
  const uint32_t total0 = a.row_totals[view_row], total1 = a.row_totals[view_row + 1u];
  const uint32_t total2 = a.row_totals[view_row + 2u], total3 = a.row_totals[view_row + 3u];
  const uint32_t last_candidate = (uint32_t)(a.candidates - 1u);

  {
    // (separate values, captured by value: a select over the elements of an array is
    // turned into an indexed load, which puts the array into scratch memory)
    const uint32_t o0 = origin[0], o1 = origin[1], o2 = origin[2], o3 = origin[3];
    const uint32_t f0 = (uint32_t)__builtin_amdgcn_readlane((int)offset[0], 63);
    const uint32_t f1 = (uint32_t)__builtin_amdgcn_readlane((int)offset[1], 63);
    const uint32_t f2 = (uint32_t)__builtin_amdgcn_readlane((int)offset[2], 63);
    const uint32_t f3 = (uint32_t)__builtin_amdgcn_readlane((int)offset[3], 63);
    gather_blend_rows<PLACE_SEGS>(
        a, blend, [o0, o1, o2, o3](uint32_t k) { return k == 0u ? o0 : (k == 1u ? o1 : (k == 2u ? o2 : o3)); },
        [f0, f1, f2, f3](uint32_t k) { return k == 0u ? f0 : (k == 1u ? f1 : (k == 2u ? f2 : f3)); });
  }

  // where the row's bucket starts inside its view's list and how much of it is
  // kept (as prefix_state lays the buckets out)
  uint32_t row_start = 0u, row_kept = 0u;
  {
    const uint32_t b = row & 3u;
    const uint32_t room0 = bucket_room(total0, a.partition_capacity, a.visible_stride, 0u);
    const uint32_t room1 = bucket_room(total1, a.partition_capacity, a.visible_stride, room0);
    const uint32_t room2 = bucket_room(total2, a.partition_capacity, a.visible_stride, room0 + room1);
    const uint32_t room3 = bucket_room(total3, a.partition_capacity, a.visible_stride, room0 + room1 + room2);
    row_kept = b == 0u ? room0 : (b == 1u ? room1 : (b == 2u ? room2 : room3));
    row_start = b == 0u ? 0u : (b == 1u ? room0 : (b == 2u ? room0 + room1 : room0 + room1 + room2));
  }
  __builtin_amdgcn_wave_barrier();  // s_head is only read by the wave that wrote it
  for (uint32_t v = 0; v < a.views; ++v) {  // wave-uniform
    const uint16_t *staged = lists + v * list_stride;
    uint32_t *list = a.visible + (uint64_t)v * a.visible_stride;
    const uint32_t r = v * 4u;
    const uint32_t s0 = (uint32_t)__builtin_amdgcn_readlane((int)row_start, (int)r);
    const uint32_t s1 = (uint32_t)__builtin_amdgcn_readlane((int)row_start, (int)(r + 1u));
    const uint32_t s2 = (uint32_t)__builtin_amdgcn_readlane((int)row_start, (int)(r + 2u));
    const uint32_t s3 = (uint32_t)__builtin_amdgcn_readlane((int)row_start, (int)(r + 3u));
    const uint32_t k0 = (uint32_t)__builtin_amdgcn_readlane((int)row_kept, (int)r);
    const uint32_t k1 = (uint32_t)__builtin_amdgcn_readlane((int)row_kept, (int)(r + 1u));
    const uint32_t k2 = (uint32_t)__builtin_amdgcn_readlane((int)row_kept, (int)(r + 2u));
    const uint32_t k3 = (uint32_t)__builtin_amdgcn_readlane((int)row_kept, (int)(r + 3u));
    // The view's runs of all segments of the wave. What lies beyond the first 64
    // entries of a stretch is requested for ALL of them before any entry is placed,
    // 64 consecutive entries per load (lane = entry, so that the stores behind them
    // fall on consecutive words): one more round trip per view, not per stretch.
    PlaceRun run[PLACE_SEGS];
    uint32_t more[PLACE_SEGS][CULL_TILE / 64 - 1];
#pragma unroll
    for (uint32_t k = 0; k < PLACE_SEGS; ++k) {
      const uint32_t c0 = (uint32_t)__builtin_amdgcn_readlane((int)count[k], (int)r);
      const uint32_t c1 = (uint32_t)__builtin_amdgcn_readlane((int)count[k], (int)(r + 1u));
      const uint32_t c2 = (uint32_t)__builtin_amdgcn_readlane((int)count[k], (int)(r + 2u));
      const uint32_t c3 = (uint32_t)__builtin_amdgcn_readlane((int)count[k], (int)(r + 3u));
      PlaceRun &n = run[k];
      n.list = list;
      n.remap = a.remap;
      n.entries = k < held ? c0 + c1 + c2 + c3 : 0u;  // <= CULL_TILE
      n.end0 = c0, n.end1 = c0 + c1, n.end2 = c0 + c1 + c2;
      // per bucket: (its start in the view's list + the segment's scanned offset in it)
      // minus the entry index at which the bucket starts in the stretch, and the end
      // of what the list keeps of the bucket
      n.shift0 = s0 + (uint32_t)__builtin_amdgcn_readlane((int)offset[k], (int)r);
      n.shift1 = s1 + (uint32_t)__builtin_amdgcn_readlane((int)offset[k], (int)(r + 1u)) - n.end0;
      n.shift2 = s2 + (uint32_t)__builtin_amdgcn_readlane((int)offset[k], (int)(r + 2u)) - n.end1;
      n.shift3 = s3 + (uint32_t)__builtin_amdgcn_readlane((int)offset[k], (int)(r + 3u)) - n.end2;
      n.stop0 = s0 + k0, n.stop1 = s1 + k1, n.stop2 = s2 + k2, n.stop3 = s3 + k3;
      n.first_row = (uint32_t)a.index_base + origin[k];
      n.origin = origin[k];
      n.last_candidate = last_candidate;
#pragma unroll
      for (uint32_t j = 1; j < (uint32_t)CULL_TILE / 64u; ++j) {
        more[k][j - 1u] = 0u;
        if (n.entries > 64u * j)  // wave-uniform
          more[k][j - 1u] = *at32(staged + (seg0 + min(k, held - 1u) * waves) * CULL_TILE, 64u * j + lane);
      }
    }
#pragma unroll
    for (uint32_t k = 0; k < PLACE_SEGS; ++k) {
      if (run[k].entries == 0u) continue;  // wave-uniform
      place_entry(run[k], lane, s_head[wave][k][v][lane]);
#pragma unroll
      for (uint32_t j = 1; j < (uint32_t)CULL_TILE / 64u; ++j) {
        if (run[k].entries <= 64u * j) break;  // wave-uniform
        place_entry(run[k], 64u * j + lane, more[k][j - 1u]);
      }
    }
  }
}

}  // namespace vkr
