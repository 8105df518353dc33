// k_exchange.hpp - device side of the multi-GPU exchange.
//
// The reference runs on one GPU; nothing here has a counterpart in it. Instances
// are sharded by contiguous range, so after every rank has run its own frame the
// whole-scene result is: per (view, bucket) the ranks' visible ranges one after
// another in rank order, and the G-way merge of the ranks' descending draw-key
// runs. The exchange is three launches around ONE all-gather and never needs the
// host to know a count:
//
//   k_exchange_pack   this rank's block: header (the execution ranges of every
//                     view, the draw count), the visible lists of all views back
//                     to back, the sorted draw keys - at fixed capacities;
//   (all-gather of the blocks, done by the caller's collective library)
//   k_exchange_plan   one block reads the G headers and writes the copy
//                     descriptors, the run table and the merged ranges, and flags
//                     a rank whose lists did not fit its block;
//   k_exchange_apply  the segmented copy of the visible lists and the rank merge
//                     of the key runs, one launch.
//
// Block layout in 32-bit words, H = exchange_header_words(views):
//   [0, views*8)            execution_ranges[view][bucket] = {offset, count}
//   [views*8, views*8+4)    draw count, visible total, views, 0
//   [H, H + vis_words)      visible candidate indices, views back to back
//   [H + vis_words, ...)    key_cap 64-bit draw keys, descending
#pragma once
#include "k_cull.hpp"

namespace vkr {

__host__ __device__ inline uint64_t round_up4(uint64_t x) { return (x + 3u) & ~(uint64_t)3u; }
__host__ __device__ inline uint64_t exchange_header_words(uint32_t views) {
  return round_up4((uint64_t)views * 8u + 4u);
}
__host__ __device__ inline uint64_t exchange_block_words(uint32_t views, uint64_t vis_cap,
                                                         uint64_t key_cap) {
  return exchange_header_words(views) + round_up4(vis_cap) + round_up4(key_cap * 2u);
}

// result of k_exchange_plan, 64-bit words
enum : uint32_t {
  XR_FLAGS = 0,             // XF_* bits; 0 = the merged outputs are complete
  XR_MAX_RANK_VISIBLE = 1,  // largest visible total of a rank (what vis_cap must hold)
  XR_MAX_RANK_KEYS = 2,     // largest draw count of a rank (what key_cap must hold)
  XR_VISIBLE_TOTAL = 3,
  XR_TRANSPARENT_TOTAL = 4,
  XR_VIEW_OFFSETS = 8       // views + 1 entries, then views*4 {offset, count} pairs
};
enum : uint64_t { XF_VISIBLE_TRUNCATED = 1, XF_KEYS_TRUNCATED = 2, XF_HEADER_MISMATCH = 4 };
__host__ __device__ inline uint64_t exchange_result_words(uint32_t views) {
  return XR_VIEW_OFFSETS + (uint64_t)views + 1u + (uint64_t)views * 8u;
}

struct ExchangePackArgs {
  const CompactionState *states;
  const uint32_t *visible;   // [view][visible_stride]
  const uint64_t *keys;      // sorted draw keys
  const uint32_t *counters;  // this frame's counter block (CNT_TRANSPARENT)
  uint32_t *block;
  uint64_t visible_stride;
  uint32_t views;
  uint32_t vis_cap;
  uint32_t key_cap;
  uint32_t have_keys;  // the frame ran the transparent stage
};

// gridDim.y = views + 1: row v < views copies view v's list, the last row the keys.
__global__ __launch_bounds__(256) void k_exchange_pack(ExchangePackArgs a) {
  const uint32_t row = blockIdx.y;
  const uint64_t header = exchange_header_words(a.views);
  const uint64_t step = (uint64_t)gridDim.x * 256u;
  const uint64_t first = (uint64_t)blockIdx.x * 256u + threadIdx.x;
  if (row < a.views) {
    uint64_t start = 0;  // uniform: scalar loads
    for (uint32_t u = 0; u < row; ++u) start += a.states[u].visible_count;
    const uint64_t count = a.states[row].visible_count;
    if (blockIdx.x == 0 && threadIdx.x < 8u)
      a.block[row * 8u + threadIdx.x] = a.states[row].execution_ranges[threadIdx.x >> 1][threadIdx.x & 1u];
    const uint32_t *src = a.visible + (uint64_t)row * a.visible_stride;
    // (what of the list fits the block; eight loads in flight per thread)
    const uint64_t n = min(count, (uint64_t)a.vis_cap - min(start, (uint64_t)a.vis_cap));
    uint32_t *dst = a.block + header + start;
    uint64_t i = first;
    for (; i + 7u * step < n; i += 8u * step) {
      uint32_t v[8];
#pragma unroll
      for (uint32_t u = 0; u < 8u; ++u) v[u] = src[i + u * step];
#pragma unroll
      for (uint32_t u = 0; u < 8u; ++u) dst[i + u * step] = v[u];
    }
    for (; i < n; i += step) dst[i] = src[i];
    return;
  }
  const uint32_t draws = a.have_keys ? a.counters[CNT_TRANSPARENT] : 0u;
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    uint64_t total = 0;
    for (uint32_t u = 0; u < a.views; ++u) total += a.states[u].visible_count;
    uint32_t *tail = a.block + (uint64_t)a.views * 8u;
    tail[0] = draws;
    tail[1] = (uint32_t)min(total, (uint64_t)0xffffffffu);
    tail[2] = a.views;
    tail[3] = 0u;
  }
  uint64_t *dst = reinterpret_cast<uint64_t *>(a.block + header + round_up4(a.vis_cap));
  const uint64_t n = min((uint64_t)draws, (uint64_t)a.key_cap);
  for (uint64_t i = first; i < n; i += step) dst[i] = a.keys[i];
}

struct ExchangePlanArgs {
  const uint32_t *gathered;  // world blocks
  uint64_t *desc;            // [views*4][world] x {src word, dst word, count}
  uint64_t *rank_view_start; // [world][views] scratch
  uint64_t *prefix;          // [views*4 + 1] scratch
  uint64_t *runs;            // [0, 64): word offset of every rank's key run, [64, 128): its length
  uint64_t *result;          // exchange_result_words(views)
  uint32_t world;
  uint32_t views;
  uint32_t vis_cap;
  uint32_t key_cap;
};

__device__ __forceinline__ uint64_t wave_inclusive_scan64(uint64_t v) {
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const uint32_t lo = __shfl_up((uint32_t)v, d, 64);
    const uint32_t hi = __shfl_up((uint32_t)(v >> 32), d, 64);
    if ((int)lane_id() >= d) v += ((uint64_t)hi << 32) | lo;
  }
  return v;
}

// One block of SCAN_BLOCK threads.
__global__ __launch_bounds__(SCAN_BLOCK) void k_exchange_plan(ExchangePlanArgs a) {
  __shared__ uint64_t s_wave[SCAN_BLOCK / 64];
  __shared__ unsigned long long s_flags, s_max_visible, s_max_keys, s_keys_total;
  const uint32_t t = threadIdx.x;
  const uint64_t header = exchange_header_words(a.views);
  const uint64_t block_words = exchange_block_words(a.views, a.vis_cap, a.key_cap);
  const uint64_t merged_cap = (uint64_t)a.world * a.vis_cap;
  if (t == 0) {
    s_flags = 0ull;
    s_max_visible = 0ull;
    s_max_keys = 0ull;
    s_keys_total = 0ull;
  }
  __syncthreads();
  // per rank: where each view starts inside the rank's packed list, its totals
  for (uint32_t r = t; r < a.world; r += SCAN_BLOCK) {
    const uint32_t *hdr = a.gathered + (uint64_t)r * block_words;
    uint64_t start = 0;
    for (uint32_t v = 0; v < a.views; ++v) {
      a.rank_view_start[(uint64_t)r * a.views + v] = start;
      // the list of a view holds its four ranges back to back
      start += (uint64_t)hdr[v * 8u + 1u] + hdr[v * 8u + 3u] + hdr[v * 8u + 5u] + hdr[v * 8u + 7u];
    }
    const uint32_t *tail = hdr + (uint64_t)a.views * 8u;
    const uint64_t draws = tail[0];
    unsigned long long flags = 0ull;
    if (start > a.vis_cap) flags |= XF_VISIBLE_TRUNCATED;
    if (draws > a.key_cap) flags |= XF_KEYS_TRUNCATED;
    if (tail[2] != a.views) flags |= XF_HEADER_MISMATCH;
    if (flags) atomicOr(&s_flags, flags);
    atomicMax(&s_max_visible, (unsigned long long)start);
    atomicMax(&s_max_keys, (unsigned long long)draws);
    const uint64_t len = min(draws, (uint64_t)a.key_cap);
    atomicAdd(&s_keys_total, (unsigned long long)len);
    a.runs[r] = (uint64_t)r * block_words + header + round_up4(a.vis_cap);
    a.runs[64u + r] = len;
  }
  __syncthreads();
  // merged count of every (view, bucket) and its exclusive prefix: views follow
  // each other and so do the buckets of a view, so one flat prefix places all
  const uint32_t pairs = a.views * 4u;
  uint64_t carry = 0;
  for (uint32_t base = 0; base < pairs; base += SCAN_BLOCK) {
    const uint32_t k = base + t;
    uint64_t merged = 0;
    if (k < pairs)
      for (uint32_t r = 0; r < a.world; ++r) merged += a.gathered[(uint64_t)r * block_words + k * 2u + 1u];
    const uint64_t inc = wave_inclusive_scan64(merged);
    __syncthreads();  // s_wave reuse
    if (lane_id() == 63) s_wave[t >> 6] = inc;
    __syncthreads();
    uint64_t wave_base = 0, sum = 0;
    for (uint32_t w = 0; w < SCAN_BLOCK / 64; ++w) {
      const uint64_t x = s_wave[w];
      if (w < (t >> 6)) wave_base += x;
      sum += x;
    }
    if (k < pairs) a.prefix[k] = carry + wave_base + inc - merged;
    carry += sum;
  }
  if (t == 0) a.prefix[pairs] = carry;
  __syncthreads();  // prefix[] is read back by other threads of this block
  for (uint32_t k = t; k < pairs; k += SCAN_BLOCK) {
    const uint32_t v = k >> 2;
    const uint64_t at = a.prefix[k];
    const uint64_t view_at = a.prefix[k & ~3u];
    uint64_t *range = a.result + XR_VIEW_OFFSETS + a.views + 1u + (uint64_t)k * 2u;
    range[0] = at - view_at;
    range[1] = a.prefix[k + 1u] - at;
    if ((k & 3u) == 0u) a.result[XR_VIEW_OFFSETS + v] = at;
    uint64_t written = 0;
    for (uint32_t r = 0; r < a.world; ++r) {
      const uint32_t *hdr = a.gathered + (uint64_t)r * block_words;
      const uint64_t in_rank = a.rank_view_start[(uint64_t)r * a.views + v] + hdr[k * 2u];
      uint64_t take = hdr[k * 2u + 1u];
      // a truncated block (flagged above) must still not be read past its end,
      // nor the merged buffer written past its own
      take = min(take, a.vis_cap - min(in_rank, (uint64_t)a.vis_cap));
      const uint64_t dst = at + written;
      take = min(take, merged_cap - min(dst, merged_cap));
      uint64_t *d = a.desc + ((uint64_t)k * a.world + r) * 3u;
      d[0] = (uint64_t)r * block_words + header + in_rank;
      d[1] = dst;
      d[2] = take;
      written += hdr[k * 2u + 1u];
    }
  }
  if (t == 0) {
    a.result[XR_VIEW_OFFSETS + a.views] = carry;
    a.result[XR_FLAGS] = s_flags;
    a.result[XR_MAX_RANK_VISIBLE] = s_max_visible;
    a.result[XR_MAX_RANK_KEYS] = s_max_keys;
    a.result[XR_VISIBLE_TOTAL] = carry;
    a.result[XR_TRANSPARENT_TOTAL] = s_keys_total;
    a.result[5] = a.result[6] = a.result[7] = 0ull;
  }
}

// Rank-based G-way merge of the gathered descending key runs (unique keys): the
// output slot of a key is its index in its own run plus, for every other run,
// the number of keys greater than it. The run table comes from k_exchange_plan.
//
// A block takes 256 consecutive keys of ONE run (blocks are numbered run after
// run; a run of n keys has ceil(n / 256) of them). Keys descend inside a run, so
// in any other run the ranks of the block's keys lie between the rank of its
// first key and the rank of its last: two full binary searches per other run -
// all of them at once, one thread each - bracket the block's keys, the bracket
// (about as many keys as the block has when the runs are alike) is copied to LDS
// and every thread finishes its search there. A key then costs about 8 loads from
// the gathered blocks instead of 17 per other run, and a block's chain of
// dependent loads is 17 long instead of 17 x (runs - 1): 8 runs of 121 K keys
// merge in a third of the time. A bracket that does not fit LDS is searched where it lies.
constexpr uint32_t XMERGE_LDS_KEYS = 1024;
__device__ __forceinline__ uint64_t keys_greater(const uint64_t *run, uint64_t lo, uint64_t hi, uint64_t key) {
  while (lo < hi) {  // first index whose key is <= `key` in a descending run
    const uint64_t mid = (lo + hi) >> 1;
    if (run[mid] > key)
      lo = mid + 1;
    else
      hi = mid;
  }
  return lo;
}
__device__ __forceinline__ void exchange_merge_keys_block(const uint32_t *gathered,
                                                          const uint64_t *runs, uint32_t g,
                                                          uint64_t *out, uint64_t block) {
  __shared__ uint64_t s_keys[XMERGE_LDS_KEYS];
  __shared__ uint64_t s_lo[64], s_hi[64];
  const uint32_t t = threadIdx.x;
  // the block's run and its chunk of it (block-uniform)
  uint32_t q = 0;
  uint64_t chunk = block;
  while (q < g) {
    const uint64_t blocks = (runs[64u + q] + 255u) / 256u;
    if (chunk < blocks) break;
    chunk -= blocks;
    ++q;
  }
  if (q >= g) return;  // past the last run
  const uint64_t *own = reinterpret_cast<const uint64_t *>(gathered + runs[q]);
  const uint64_t len = runs[64u + q];
  const uint64_t j0 = chunk * 256u, j1 = min(j0 + 256u, len);  // j0 < j1
  const uint64_t j = j0 + t;
  const bool live = j < j1;
  const uint64_t key = own[live ? j : j1 - 1u];
  // thread 2 r brackets run r from above (the block's first key), 2 r + 1 from below (its last)
  if (t < 2u * g) {
    const uint32_t r = t >> 1;
    if (r != q) {
      const uint64_t *run = reinterpret_cast<const uint64_t *>(gathered + runs[r]);
      const uint64_t at = keys_greater(run, 0, runs[64u + r], own[(t & 1u) ? j1 - 1u : j0]);
      if (t & 1u) s_hi[r] = at;
      else s_lo[r] = at;
    }
  }
  __syncthreads();
  uint64_t pos = j;
  for (uint32_t r = 0; r < g; ++r) {  // block-uniform
    if (r == q) continue;
    const uint64_t *run = reinterpret_cast<const uint64_t *>(gathered + runs[r]);
    const uint64_t lo = s_lo[r], hi = s_hi[r];
    if (hi - lo <= XMERGE_LDS_KEYS) {
      for (uint64_t k = t; k < hi - lo; k += 256u) s_keys[k] = run[lo + k];
      __syncthreads();
      pos += lo + keys_greater(s_keys, 0, hi - lo, key);
      __syncthreads();
    } else {
      pos += keys_greater(run, lo, hi, key);
    }
  }
  if (live) out[pos] = key;
}

// The segmented copy of the visible lists and the key merge in one launch (both
// only depend on the plan): blocks [0, copy_blocks) are a (descriptor, row) of
// the copy, the rest merge keys.
__global__ __launch_bounds__(256) void k_exchange_apply(const uint32_t *gathered, const uint64_t *desc,
                                                        uint32_t descriptors, uint32_t rows,
                                                        uint32_t *merged_visible,
                                                        const uint64_t *runs, uint32_t g,
                                                        uint64_t *merged_keys) {
  const uint32_t copy_blocks = descriptors * rows;
  if (blockIdx.x < copy_blocks) {
    const uint32_t d = blockIdx.x / rows, row = blockIdx.x % rows;
    const uint64_t src = desc[d * 3u + 0], dst = desc[d * 3u + 1], n = desc[d * 3u + 2];
    // (eight loads in flight per thread: a copy that waits for every word before it
    // asks for the next moves 432 MB of lists - eight ranks - at a third of the rate)
    const uint64_t stride = (uint64_t)rows * 256u;
    uint64_t k = (uint64_t)row * 256u + threadIdx.x;
    for (; k + 7u * stride < n; k += 8u * stride) {
      uint32_t v[8];
#pragma unroll
      for (uint32_t u = 0; u < 8u; ++u) v[u] = __builtin_nontemporal_load(gathered + src + k + u * stride);
#pragma unroll
      for (uint32_t u = 0; u < 8u; ++u) merged_visible[dst + k + u * stride] = v[u];
    }
    for (; k < n; k += stride) merged_visible[dst + k] = gathered[src + k];
    return;
  }
  exchange_merge_keys_block(gathered, runs, g, merged_keys, blockIdx.x - copy_blocks);
}

}  // namespace vkr
