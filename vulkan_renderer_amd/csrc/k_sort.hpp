// k_sort.hpp - ordering of the transparent draw keys and draw emission.
//
// Key: ((uint64)bits(depth) << 32) | (source_index + 1), depth clamped >= 0
// (lib/src/application.h:573-587), ordered DESCENDING by the reference's
// comparator (lib/src/renderer/vkr_visibility.c:5-13) and qsort
// (application.h:940-942). Keys are unique, so any correct sort reproduces the
// reference's order. The low word names the source row, so no payload travels
// with the key.
//
// Small counts (<= RANK_LIMIT) are ordered by counting, in the same launch that
// emits the rows (k_rank_emit). Larger counts go through a stable LSD radix
// sort: the visible blend rows arrive in ascending source index, so a STABLE
// ascending sort on key bits [32, 64) followed by reading the result backwards
// yields descending keys with larger source index first among equal depths -
// exactly descending order on the full 64-bit key - while only the 32 depth
// bits have to be moved through the radix passes: one histogram launch for all
// four digit positions, then one launch per 8-bit pass (decoupled look-back
// between the tiles of a pass, see k_sort_pass).
// All kernels read the count on the device (no host round trip between cull
// and sort). Ranking inside a wave is a
// 64-wide match on the digit built from eight ballots, with per-wave running
// digit counters in LDS.
#pragma once
#include "device_math.hpp"
#include "k_cull.hpp"

namespace vkr {

constexpr int SORT_BLOCK = 256;
constexpr int SORT_ITEMS = 8;                       // keys per lane
constexpr int SORT_TILE = SORT_BLOCK * SORT_ITEMS;  // 2048 keys per block
constexpr int SORT_WAVE_KEYS = 64 * SORT_ITEMS;     // 512 consecutive keys per wave
constexpr int SORT_RADIX = 256;

constexpr int SMALL_THREADS = 512;
constexpr int SMALL_WAVES = SMALL_THREADS / 64;
constexpr int SMALL_KEYS = 8192;
constexpr int SMALL_STEPS = SMALL_KEYS / SMALL_THREADS;  // 64-key steps per wave
constexpr int SMALL_CHUNK = 4;                           // steps ranked together

// Lanes of the wave whose `digit` equals this lane's (restricted to `valid` lanes).
__device__ __forceinline__ unsigned long long match_digit(uint32_t digit, bool valid) {
  unsigned long long peers = __ballot(valid);
#pragma unroll
  for (int b = 0; b < 8; ++b) {
    const bool bit = (digit >> b) & 1u;
    const unsigned long long m = __ballot(bit);
    peers &= bit ? m : ~m;
  }
  return peers;
}

// Stable ranks of one wave's keys for one digit position. The wave owns `STEPS`
// groups of 64 consecutive keys (key[k] of lane l is element k*64 + l); on
// return rank[k] is the number of the wave's keys with the same digit that come
// before it, and whist[d] has grown by the wave's count of digit d. whist must
// only be touched by this wave while this runs.
//
// The running per-digit count is kept by LDS atomics issued by the first lane of
// every group of equal digits. They return in issue order and nothing consumes
// their results until every step has issued, so the LDS latency is paid once
// per tile, not once per step.
template <int STEPS>
__device__ __forceinline__ void wave_rank(const uint64_t (&key)[STEPS], const bool (&valid)[STEPS],
                                          uint32_t steps, uint32_t shift, uint32_t *whist,
                                          uint32_t (&rank)[STEPS]) {
  const uint32_t lane = threadIdx.x & 63u;
  const unsigned long long lt_mask = (1ull << lane) - 1ull;
  uint32_t before[STEPS];
  uint32_t leader[STEPS];
#pragma unroll
  for (int k = 0; k < STEPS; ++k) {
    before[k] = 0u;
    leader[k] = 0u;
    rank[k] = 0u;
    if ((uint32_t)k < steps) {  // wave-uniform
      const uint32_t digit = (uint32_t)(key[k] >> shift) & 0xffu;
      const unsigned long long peers = match_digit(digit, valid[k]);
      const unsigned long long lower = peers & lt_mask;
      rank[k] = (uint32_t)__popcll(lower);
      leader[k] = peers ? (uint32_t)__ffsll((long long)peers) - 1u : 0u;
      if (valid[k] && lower == 0ull) before[k] = atomicAdd(&whist[digit], (uint32_t)__popcll(peers));
    }
  }
#pragma unroll
  for (int k = 0; k < STEPS; ++k)
    if ((uint32_t)k < steps) rank[k] += (uint32_t)__shfl((int)before[k], (int)leader[k], 64);
}


// LDS of the single-workgroup sort (the keys stay in global memory).
struct SmallSortShared {
  uint16_t rank[SMALL_KEYS];  // rank of each key of the current tile inside its wave's digit run
  uint32_t whist[SMALL_WAVES][SORT_RADIX];
  uint32_t base[SORT_RADIX];
  uint32_t wave[SMALL_THREADS / 64];
  uint32_t skip;
};

// One stable pass of a whole array by ONE workgroup: digit histogram ->
// exclusive scan -> tiles of SMALL_KEYS keys in order, each ranked by wave_rank
// and scattered at (running digit base + wave prefix + rank). Returns false
// (and writes nothing) when every key has the same digit.
__device__ __forceinline__ bool block_radix_pass(const uint64_t *src, uint64_t *dst,
                                                 uint32_t count, uint32_t shift,
                                                 SmallSortShared &s) {
  const uint32_t t = threadIdx.x;
  const uint32_t wave = t >> 6;
  const uint32_t lane = t & 63u;
  // A single tile's digit totals fall out of the ranking below; only a longer
  // array needs the histogram up front.
  const bool single = count <= SMALL_KEYS;
  __syncthreads();  // src is complete, the previous pass is done with s
  if (t < SORT_RADIX) s.base[t] = 0u;
  if (t == 0) s.skip = 0u;
  if (!single) {
    __syncthreads();
    // wave-aggregated: one LDS atomic per group of equal digits, not per key
    // (the high bytes of a float key take very few values)
    for (uint32_t base = 0; base < count; base += SMALL_THREADS) {
      const uint32_t idx = base + t;
      const bool valid = idx < count;
      const uint32_t digit = valid ? (uint32_t)(src[idx] >> shift) & 0xffu : 0u;
      const unsigned long long peers = match_digit(digit, valid);
      if (valid && (peers & ((1ull << lane) - 1ull)) == 0ull)
        atomicAdd(&s.base[digit], (uint32_t)__popcll(peers));
    }
    __syncthreads();
    const uint32_t tot = t < SORT_RADIX ? s.base[t] : 0u;
    if (t < SORT_RADIX && tot == count) s.skip = 1u;
    uint32_t total;
    const uint32_t ex = block_exclusive_scan(tot, s.wave, &total);
    __syncthreads();
    if (t < SORT_RADIX) s.base[t] = ex;
    if (s.skip) return false;  // uniform: read after the barrier above
  }

  for (uint32_t tile_base = 0; tile_base < count; tile_base += SMALL_KEYS) {
    __syncthreads();  // s.base updated, previous tile's whist readers done
    for (uint32_t d = t; d < SMALL_WAVES * SORT_RADIX; d += SMALL_THREADS) (&s.whist[0][0])[d] = 0u;
    __syncthreads();
    const uint32_t n_tile = min((uint32_t)SMALL_KEYS, count - tile_base);
    const uint32_t steps = (n_tile + SMALL_THREADS - 1) / SMALL_THREADS;  // per wave, <= SMALL_STEPS
    const uint32_t wbase = tile_base + wave * steps * 64u;
    const uint32_t tile_end = tile_base + n_tile;
    // ranks, four steps at a time (keeps the register footprint small; the ranks
    // wait in LDS for the scatter)
#pragma unroll 1
    for (uint32_t k0 = 0; k0 < steps; k0 += SMALL_CHUNK) {
      uint64_t key[SMALL_CHUNK];
      bool valid[SMALL_CHUNK];
      uint32_t rank[SMALL_CHUNK];
#pragma unroll
      for (int k = 0; k < SMALL_CHUNK; ++k) {
        const uint32_t idx = wbase + (k0 + (uint32_t)k) * 64u + lane;
        valid[k] = k0 + (uint32_t)k < steps && idx < tile_end;
        key[k] = valid[k] ? src[idx] : 0ull;
      }
      wave_rank<SMALL_CHUNK>(key, valid, steps - k0, shift, s.whist[wave], rank);
#pragma unroll
      for (int k = 0; k < SMALL_CHUNK; ++k) {
        const uint32_t idx = wbase + (k0 + (uint32_t)k) * 64u + lane;
        if (valid[k]) s.rank[idx - tile_base] = (uint16_t)rank[k];
      }
    }
    __syncthreads();
    if (single) {
      uint32_t tot = 0;
      if (t < SORT_RADIX) {
#pragma unroll
        for (int w = 0; w < SMALL_WAVES; ++w) tot += s.whist[w][t];
        if (tot == count) s.skip = 1u;  // every key has this digit: identity permutation
      }
      uint32_t total;
      const uint32_t ex = block_exclusive_scan(tot, s.wave, &total);
      if (t < SORT_RADIX) s.base[t] = ex;
      __syncthreads();
      if (s.skip) return false;  // uniform
    }
    if (t < SORT_RADIX) {
      uint32_t run = s.base[t];
#pragma unroll
      for (int w = 0; w < SMALL_WAVES; ++w) {
        const uint32_t c = s.whist[w][t];
        s.whist[w][t] = run;
        run += c;
      }
      s.base[t] = run;  // running base of the next tile
    }
    __syncthreads();
#pragma unroll 4
    for (uint32_t k = 0; k < steps; ++k) {
      const uint32_t idx = wbase + k * 64u + lane;
      if (idx < tile_end) {
        const uint64_t key = src[idx];
        const uint32_t digit = (uint32_t)(key >> shift) & 0xffu;
        dst[s.whist[wave][digit] + s.rank[idx - tile_base]] = key;
      }
    }
  }
  return true;
}

// ---------------------------------------------------------------------------
// Large counts: four stable 8-bit passes, ONE launch each, behind one histogram
// launch that reads the keys once for all four digit positions.
//
// A pass must know, for every (tile, digit), how many keys with that digit lie
// in the tiles before it. Instead of a histogram and a scan launch per pass, a
// block publishes its tile's digit counts as soon as it has ranked the tile and
// then looks back over the tiles before it until it meets one whose running
// total is known ("decoupled look-back"). Tiles are handed out by a ticket
// counter, so a block only ever waits for tiles taken earlier, i.e. for blocks
// that are already running: every wait ends. A status word is
//   frame serial << 32 | state << 30 | count      (state 1: the tile's own count,
//                                                  state 2: running total up to and
//                                                  including the tile)
// and needs no clearing between frames (a stale serial reads as "not there yet").
// The global digit histograms, the ticket counters and the stall flag live in
// the frame's zeroed counter block.
// ---------------------------------------------------------------------------
constexpr int SORT_PASSES = 4;
constexpr uint32_t SORT_STATE_WORDS = SORT_PASSES * SORT_RADIX + SORT_PASSES + 4;  // see SortState
constexpr uint32_t SORT_SPIN_LIMIT = 1u << 24;  // look-back polls before a block gives up

struct SortState {  // zero at the start of a frame
  uint32_t hist[SORT_PASSES][SORT_RADIX];  // keys per digit, per pass
  uint32_t ticket[SORT_PASSES];            // next tile of the pass
  uint32_t stalled;                        // a look-back gave up (never expected): 1 | pass << 8 | digit << 16
  uint32_t stalled_tile;                   // ... the tile that gave up,
  uint32_t stalled_on;                     // ... and the tile whose status it was waiting for
  uint32_t reserved;
};
static_assert(sizeof(SortState) == SORT_STATE_WORDS * sizeof(uint32_t), "SortState layout");

struct SortArgs {
  const uint64_t *keys_in;
  uint64_t *keys_out;
  SortState *state;
  unsigned long long *status;  // [SORT_PASSES][tiles_stride][SORT_RADIX]
  const uint32_t *counters;    // CNT_TRANSPARENT
  uint32_t tiles_stride;
  uint32_t pass;               // 0..3: key bits [32 + 8 pass, 40 + 8 pass)
  uint32_t serial;             // frame serial of the status words (never 0)
  uint32_t capacity;           // elements of keys_in / keys_out (> 0)
  uint32_t descending;         // this pass writes its output back to front
};

// Digit histograms of all four passes in one read of the keys. The keys of a
// block's first tile are requested together with the count (from clamped
// positions), not after it.
__global__ __launch_bounds__(SORT_BLOCK) void k_sort_hist(SortArgs a) {
  __shared__ uint32_t s_hist[SORT_PASSES][SORT_RADIX];
  const uint32_t t = threadIdx.x;
  const uint32_t last = a.capacity - 1u;
  uint64_t key[SORT_ITEMS];
#pragma unroll
  for (int k = 0; k < SORT_ITEMS; ++k)
    key[k] = a.keys_in[min(blockIdx.x * SORT_TILE + (uint32_t)k * SORT_BLOCK + t, last)];
  const uint32_t count = a.counters[CNT_TRANSPARENT];
  const uint32_t tiles = (count + SORT_TILE - 1) / SORT_TILE;
  if (blockIdx.x >= tiles) return;  // block-uniform
#pragma unroll
  for (int p = 0; p < SORT_PASSES; ++p) s_hist[p][t] = 0u;
  __syncthreads();
  for (uint32_t tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
    const uint32_t base = tile * SORT_TILE;
#pragma unroll
    for (int k = 0; k < SORT_ITEMS; ++k) {
      const uint32_t idx = base + (uint32_t)k * SORT_BLOCK + t;
      if (tile != blockIdx.x) key[k] = a.keys_in[min(idx, last)];  // block-uniform
      if (idx < count) {
        const uint32_t hi = (uint32_t)(key[k] >> 32);
#pragma unroll
        for (int p = 0; p < SORT_PASSES; ++p) atomicAdd(&s_hist[p][(hi >> (8 * p)) & 0xffu], 1u);
      }
    }
  }
  __syncthreads();
#pragma unroll
  for (int p = 0; p < SORT_PASSES; ++p) {
    const uint32_t c = s_hist[p][t];
    if (c) atomicAdd(&a.state->hist[p][t], c);
  }
}

__device__ __forceinline__ unsigned long long sort_status(uint32_t serial, uint32_t state,
                                                          uint32_t count) {
  return ((unsigned long long)serial << 32) | ((unsigned long long)state << 30) | count;
}

// One stable pass. Wave w of the block owns keys [w*512, w*512+512) of its tile
// and walks them in 8 steps of 64 consecutive keys, so (wave, step, lane) order
// is memory order.
//
// Every tile, the first one of a block included, is handed out by the ticket
// counter: a tile's look-back only waits for lower tiles, i.e. for tickets taken
// earlier, i.e. for blocks that are already running - whatever share of the grid
// the device holds at once (beside the next frame's transform, on the second
// stream, that is often a handful of blocks, and the blocks of a grid are not
// promised to start in index order). The keys a block requests before it knows
// its tile are those of tile blockIdx.x, which is the tile it gets when the
// blocks start in order; otherwise it requests its tile's keys again.
//
// At most 128 registers (four waves per SIMD): beside the next frame's large level
// launch - four waves of 128 registers per SIMD, the whole register file - a block
// of this kernel is placed as soon as one block of that grid retires. With 204
// registers it needed two of them to retire on the same SIMD at once, which does
// not happen before the grid has no block left to place: a pass then sat through
// the whole launch (400 us for 12 us of work).
__global__ __launch_bounds__(SORT_BLOCK) __attribute__((amdgpu_waves_per_eu(4, 4))) void k_sort_pass(SortArgs a) {
  __shared__ uint32_t s_whist[4][SORT_RADIX];  // per-wave digit counts, then bases
  __shared__ uint32_t s_wave[SORT_BLOCK / 64];
  __shared__ uint32_t s_tile;
  const uint32_t t = threadIdx.x;
  const uint32_t wave = t >> 6;
  const uint32_t lane = t & 63u;
  const uint32_t shift = 32u + 8u * a.pass;
  const uint32_t last = a.capacity - 1u;
  uint64_t key[SORT_ITEMS];
  {
    const uint32_t wbase = blockIdx.x * SORT_TILE + wave * SORT_WAVE_KEYS;
#pragma unroll
    for (int k = 0; k < SORT_ITEMS; ++k) key[k] = a.keys_in[min(wbase + (uint32_t)k * 64u + lane, last)];
  }
  if (t == 0) s_tile = atomicAdd(&a.state->ticket[a.pass], 1u);
  const uint32_t my_total = a.state->hist[a.pass][t];
  const uint32_t count = a.counters[CNT_TRANSPARENT];
  const uint32_t tiles = (count + SORT_TILE - 1) / SORT_TILE;
  for (int w = 0; w < 4; ++w) s_whist[w][t] = 0u;
  __syncthreads();
  const uint32_t first_tile = s_tile;
  if (first_tile >= tiles) return;  // block-uniform: the grid covers an estimate of the count
  // keys with smaller digits come first: exclusive scan of the digit totals
  uint32_t all_keys;
  const uint32_t digit_base = block_exclusive_scan(my_total, s_wave, &all_keys);
  unsigned long long *status = a.status + (size_t)a.pass * a.tiles_stride * SORT_RADIX;

  bool requested = first_tile == blockIdx.x;  // key[] holds the tile's keys already
  for (uint32_t tile = first_tile;;) {
    bool valid[SORT_ITEMS];
    uint32_t rank[SORT_ITEMS];
    const uint32_t wbase = tile * SORT_TILE + wave * SORT_WAVE_KEYS;
#pragma unroll
    for (int k = 0; k < SORT_ITEMS; ++k) {
      const uint32_t idx = wbase + (uint32_t)k * 64u + lane;
      valid[k] = idx < count;
      if (!requested) key[k] = a.keys_in[min(idx, last)];  // block-uniform
      if (!valid[k]) key[k] = 0ull;
    }
    wave_rank<SORT_ITEMS>(key, valid, SORT_ITEMS, shift, s_whist[wave], rank);
    __syncthreads();
    // thread t = digit t: the tile's count, published at once for the tiles behind
    uint32_t c[4], mine = 0u;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      c[w] = s_whist[w][t];
      mine += c[w];
    }
    unsigned long long *own = status + (size_t)tile * SORT_RADIX + t;
    __hip_atomic_store(own, sort_status(a.serial, tile == 0u ? 2u : 1u, mine), __ATOMIC_RELAXED,
                       __HIP_MEMORY_SCOPE_AGENT);
    // look back: keys of this digit in the tiles before this one
    uint32_t before = 0u;
    if (tile != 0u) {
      // twelve tiles per step, nearest first: their status words are independent
      // loads, so a step costs one memory round trip, not twelve (thirty-two per
      // step measured no faster and cost 80 registers more)
      constexpr uint32_t LOOK = 12;
      uint32_t polls = 0u;
      bool done = false;
      for (uint32_t hi = tile; hi != 0u && !done;) {
        const uint32_t n = min(hi, LOOK);
        for (;;) {
          unsigned long long v[LOOK];
#pragma unroll
          for (uint32_t j = 0; j < LOOK; ++j) {
            const uint32_t p = hi - 1u - min(j, n - 1u);  // (slots past n repeat the last tile)
            v[j] = __hip_atomic_load(status + (size_t)p * SORT_RADIX + t, __ATOMIC_RELAXED,
                                     __HIP_MEMORY_SCOPE_AGENT);
          }
          // nearest first: add counts until a running total (nothing further back is
          // needed, and it may not be there yet) or a tile that has not published
          uint32_t sum = 0u;
          bool blocked = false, total = false;
#pragma unroll
          for (uint32_t j = 0; j < LOOK; ++j) {
            if (j < n && !blocked && !total) {
              const uint32_t state = (uint32_t)(v[j] >> 30) & 3u;
              if ((uint32_t)(v[j] >> 32) != a.serial || state == 0u) {
                blocked = true;
              } else {
                sum += (uint32_t)v[j] & 0x3fffffffu;
                total = state == 2u;
              }
            }
          }
          if (!blocked) {
            before += sum;
            done = total;
            break;
          }
          if (++polls > SORT_SPIN_LIMIT) break;
          __builtin_amdgcn_s_sleep(1);
        }
        if (polls > SORT_SPIN_LIMIT) {  // never expected: leave a mark that names the place and get out
          a.state->stalled = 1u | (a.pass << 8) | (t << 16);
          a.state->stalled_tile = tile;
          a.state->stalled_on = hi - 1u;
          break;
        }
        hi -= n;
      }
      __hip_atomic_store(own, sort_status(a.serial, 2u, before + mine), __ATOMIC_RELAXED,
                         __HIP_MEMORY_SCOPE_AGENT);
    }
    // exclusive prefix of the per-wave counts, behind the digit's global base
    {
      uint32_t run = digit_base + before;
#pragma unroll
      for (int w = 0; w < 4; ++w) {
        s_whist[w][t] = run;
        run += c[w];
      }
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < SORT_ITEMS; ++k) {
      if (valid[k]) {
        const uint32_t digit = (uint32_t)(key[k] >> shift) & 0xffu;
        const uint32_t pos = s_whist[wave][digit] + rank[k];
        a.keys_out[a.descending ? all_keys - 1u - pos : pos] = key[k];
      }
    }
    if (tiles <= gridDim.x) break;  // block-uniform: every block takes one ticket before any takes a second
    requested = false;
    __syncthreads();                // the tile's readers are done with the tables
    if (t == 0) s_tile = atomicAdd(&a.state->ticket[a.pass], 1u);
    for (int w = 0; w < 4; ++w) s_whist[w][t] = 0u;
    __syncthreads();
    tile = s_tile;
    if (tile >= tiles) break;  // block-uniform
  }
}

// The keys that are sorted are "compact" keys: depth bits << 32 | the row's place
// p among the frame's visible blend rows (candidate order, so the low word
// orders ties like the source index does). Everything the emission needs of a
// row was set aside under p when the frame's lists were gathered: its candidate
// (tsrc) and its world matrix (tworld). The kernels here therefore read nothing
// that the next frame's transform overwrites, and run beside it on a second
// stream.
struct EmitArgs {
  const uint64_t *keys;      // sorted compact keys: ascending, or descending when keys_descending
  const uint32_t *counters;  // CNT_TRANSPARENT
  const float4 *tworld;      // 4 x float4 per visible blend row
  const uint32_t *tsrc;      // its candidate; null: the key's low word IS the candidate
  const float4 *world;       // when tworld is null (single stream): the matrices are read in place,
  const uint32_t *node;      //   through candidate -> node (null = identity)
  const uint4 *ident;  // 2 x uint4 per candidate
  uint4 *draws;        // 3 x uint4 per row: VkrDrawItem
  float4 *instances;   // 5 x float4 per row: VkrInstanceDataGPU
  uint64_t *out_keys;  // descending
  uint32_t *out_source;
  uint64_t index_base;
  uint32_t capacity;         // elements of keys (> 0)
  uint32_t keys_descending;  // k_emit_transparent only
};

// vkr_picking_encode_id(SCENE, render_id), systems/vkr_picking_ids.h:35-42;
// render_id 0 maps to object id 0 (application.h:904-907).
__host__ __device__ __forceinline__ uint32_t encode_scene_object_id(uint32_t render_id) {
  if (render_id == 0u) return 0u;
  if (render_id > 0x1ffffffeu) return 0u;
  return render_id + 1u;
}

// Output rows of vkr_transparent_draw_emit (lib/src/renderer/vkr_visibility.c:15-35)
// are written as 16-byte pieces: pieces 0-2 are the VkrDrawItem, 3-7 the
// VkrInstanceDataGPU; eight lanes take one row so that each output array is
// written in consecutive 16-byte pieces. Every piece is made from ONE 16-byte load
// whose address is computed, not branched to (the identity words of the row's
// candidate, or a column of its world matrix), so a caller can request the pieces
// of several rows back to back and before it knows the output rows.
//
// The candidate of the row with this key: the low word of a compact key is the
// row's place, under which the gather left the candidate (tsrc), or - when the
// rows' matrices were set aside under the candidate index - the candidate itself.
__device__ __forceinline__ uint32_t emit_candidate(const EmitArgs &a, uint64_t key) {
  const uint32_t p = (uint32_t)key;
  return a.tsrc ? a.tsrc[p] : p;
}

// Where the 16 bytes come from that piece `piece` of candidate i's row is made of.
__device__ __forceinline__ const uint4 *emit_piece_source(const EmitArgs &a, uint64_t key, uint32_t i,
                                                          uint64_t nd, uint32_t piece) {
  const bool column = piece >= 3u && piece < 7u;
  const float4 *matrix = a.tworld ? a.tworld + (uint64_t)(uint32_t)key * 4u : a.world + nd * 4u;
  const uint4 *words = a.ident + (uint64_t)i * 2u + ((piece == 1u || piece == 7u) ? 1u : 0u);
  return column ? reinterpret_cast<const uint4 *>(matrix + (piece - 3u)) : words;
}

// The part of the piece that does not depend on the output row, from the loaded 16 bytes.
__device__ __forceinline__ uint4 emit_piece_finish(const EmitArgs &a, uint64_t key, uint32_t i,
                                                   uint32_t piece, uint4 raw) {
  // piece 0: mesh.id/.generation, geometry.id/.generation; 3-6: the matrix columns
  if (piece == 1u) return make_uint4(raw.x, raw.y, raw.z, 1u);  // submesh_index, material, instance_count
  // the sort key as the reference packs it: the low word is source_index + 1
  // (application.h:573-587,657-658)
  if (piece == 2u) return make_uint4(0u, 0u, (uint32_t)(a.index_base + i) + 1u, (uint32_t)(key >> 32));
  if (piece == 7u) return make_uint4(encode_scene_object_id(raw.w), 0u, 0u, 0u);
  return raw;
}

__device__ __forceinline__ uint4 emit_piece_gather(const EmitArgs &a, uint64_t key, uint32_t piece) {
  const uint32_t i = emit_candidate(a, key);
  const uint64_t nd = (!a.tworld && a.node) ? a.node[i] : i;
  return emit_piece_finish(a, key, i, piece, *emit_piece_source(a, key, i, nd, piece));
}

__device__ __forceinline__ void emit_piece_store(const EmitArgs &a, uint32_t row, uint64_t key,
                                                 uint32_t piece, uint4 o) {
  if (piece < 3u) {
    if (piece == 2u) {
      o.x = row;  // first_instance
      a.out_keys[row] = ((uint64_t)o.w << 32) | o.z;
      a.out_source[row] = (uint32_t)((uint64_t)(o.z - 1u) - a.index_base);
    }
    a.draws[(uint64_t)row * 3u + piece] = o;
  } else {
    a.instances[(uint64_t)row * 5u + (piece - 3u)] =
        make_float4(__uint_as_float(o.x), __uint_as_float(o.y), __uint_as_float(o.z),
                    __uint_as_float(o.w));
  }
}

__device__ __forceinline__ void emit_transparent_piece(const EmitArgs &a, uint32_t row,
                                                       uint64_t key, uint32_t piece) {
  emit_piece_store(a, row, key, piece, emit_piece_gather(a, key, piece));
}

// Emission from the sorted key array: read backwards when it is ascending; when
// the last sort pass left it descending, a row's key does not depend on the
// count and is requested together with it. A block takes EMIT_BATCH x 32
// consecutive rows, eight lanes per row, and every step of a row's chain of
// dependent loads (key -> candidate -> node -> 16 bytes) is requested for all of a
// thread's rows before the first is used: the launch runs beside the next frame's
// level launches, where a round trip takes several microseconds and a resident
// block holds a place that a block of those launches would fill.
constexpr uint32_t EMIT_BATCH = 8;
constexpr uint32_t EMIT_BLOCK_ROWS = 32u * EMIT_BATCH;
__global__ __launch_bounds__(256) void k_emit_transparent(EmitArgs a) {
  const uint32_t piece = threadIdx.x & 7u;
  const uint32_t group = threadIdx.x >> 3;
  const uint32_t last_key = a.capacity - 1u;
  uint32_t base = blockIdx.x * EMIT_BLOCK_ROWS;
  uint64_t key[EMIT_BATCH];
#pragma unroll
  for (uint32_t k = 0; k < EMIT_BATCH; ++k) key[k] = a.keys[min(base + k * 32u + group, last_key)];
  const uint32_t count = min(a.counters[CNT_TRANSPARENT], a.capacity);
  for (; base < count; base += gridDim.x * EMIT_BLOCK_ROWS) {  // block-uniform
    uint32_t row[EMIT_BATCH], cand[EMIT_BATCH];
    uint64_t nd[EMIT_BATCH];
    uint4 raw[EMIT_BATCH];
    bool live[EMIT_BATCH];
#pragma unroll
    for (uint32_t k = 0; k < EMIT_BATCH; ++k) {
      row[k] = base + k * 32u + group;
      live[k] = row[k] < count;
    }
    if (!a.keys_descending || base != blockIdx.x * EMIT_BLOCK_ROWS) {
#pragma unroll
      for (uint32_t k = 0; k < EMIT_BATCH; ++k) {
        const uint32_t r = min(row[k], count - 1u);
        key[k] = a.keys[a.keys_descending ? r : count - 1u - r];
      }
    }
    // rows past the count: their keys are whatever the array holds; they load from
    // place 0 and store nothing
#pragma unroll
    for (uint32_t k = 0; k < EMIT_BATCH; ++k) {
      if (!live[k]) key[k] = 0ull;
      cand[k] = (uint32_t)key[k];
    }
    if (a.tsrc) {
#pragma unroll
      for (uint32_t k = 0; k < EMIT_BATCH; ++k) cand[k] = a.tsrc[(uint32_t)key[k]];
    }
#pragma unroll
    for (uint32_t k = 0; k < EMIT_BATCH; ++k) nd[k] = cand[k];
    if (!a.tworld && a.node) {
#pragma unroll
      for (uint32_t k = 0; k < EMIT_BATCH; ++k) nd[k] = a.node[cand[k]];
    }
#pragma unroll
    for (uint32_t k = 0; k < EMIT_BATCH; ++k) raw[k] = *emit_piece_source(a, key[k], cand[k], nd[k], piece);
#pragma unroll
    for (uint32_t k = 0; k < EMIT_BATCH; ++k)
      if (live[k]) emit_piece_store(a, row[k], key[k], piece, emit_piece_finish(a, key[k], cand[k], piece, raw[k]));
  }
}

// Sort + emission in one launch for small counts. Sort keys are unique (the low
// word is the source row), so the output row of a key is simply the number of
// keys greater than it. A block takes RANK_BLOCK_KEYS keys; its eight waves
// split the key array between them and sweep their stretch past the block's
// keys, 64 keys per load, one compare + ballot population count per (key,
// load); the partial counts meet in LDS and one wave writes the eight 16-byte
// pieces of the eight rows. The sweep reads the array straight from L2 (every
// wave load is 512 consecutive bytes) - no copy of the whole array per block, no
// barrier before the sweep. count^2 / 64 compares: 1M wave-steps at 8192 keys,
// spread over every CU, against four passes of barriers and LDS atomics on one
// CU for a radix sort.
//
// Above RANK_LIMIT keys the quadratic sweep is not started: block 0 alone sorts
// by streaming radix passes through global memory and emits every row. That
// only happens when the host launched this kernel on a stale small estimate of
// the count; it is correct, but runs at one CU's bandwidth.
constexpr int RANK_BLOCK_KEYS = 8;
constexpr int RANK_LOADS = 4;  // sweep loads in flight per wave
// (ranking is quadratic, the radix path five dependent launches: measured with the tail
// beside the next frame, 7 K keys 102 -> 67 us per frame and 14.6 K keys 132 -> 113 us by
// ranking, 21 K keys 161 -> 168 us - the paths cross at about 16 K keys)
constexpr uint32_t RANK_LIMIT = 20480;

__global__ __launch_bounds__(SMALL_THREADS) void k_rank_emit(EmitArgs a, uint64_t *keys,
                                                             uint64_t *scratch) {
  __shared__ SmallSortShared s;  // the fallback's staging
  __shared__ uint32_t s_part[SMALL_WAVES][RANK_BLOCK_KEYS];
  const uint32_t count = a.counters[CNT_TRANSPARENT];
  if (count == 0) return;
  const uint32_t t = threadIdx.x;
  if (count > RANK_LIMIT) {
    if (blockIdx.x != 0) return;
    uint64_t *src = keys, *dst = scratch;
    for (uint32_t shift = 32u; shift < 64u; shift += 8u) {
      if (block_radix_pass(src, dst, count, shift, s)) {
        uint64_t *tmp = src;
        src = dst;
        dst = tmp;
      }
    }
    __syncthreads();
    for (uint32_t row = t >> 3; row < count; row += SMALL_THREADS / 8u)
      emit_transparent_piece(a, row, src[count - 1u - row], t & 7u);
    return;
  }

  const uint32_t lane = t & 63u;
  const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(t >> 6));
  // the wave's stretch of the array: an eighth of it, in whole 64-key loads
  const uint32_t per_wave = ((count + 64u * SMALL_WAVES - 1u) / (64u * SMALL_WAVES)) * 64u;
  const uint32_t lo = wave * per_wave;
  const uint32_t hi = min(lo + per_wave, count);
  const uniform_ptr<uint64_t> block_keys = as_uniform(keys);  // nothing writes keys[] on this path
  const uint32_t groups = (count + RANK_BLOCK_KEYS - 1u) / RANK_BLOCK_KEYS;
  for (uint32_t g = blockIdx.x; g < groups; g += gridDim.x) {
    // the block's keys, scalar loads; slots past the end repeat the last key
    // (they are not emitted)
    uint64_t mine[RANK_BLOCK_KEYS];
    uint32_t greater[RANK_BLOCK_KEYS];
#pragma unroll
    for (int k = 0; k < RANK_BLOCK_KEYS; ++k) {
      mine[k] = block_keys[min(g * RANK_BLOCK_KEYS + (uint32_t)k, count - 1u)];
      greater[k] = 0u;
    }
    // wave 0 emits (a wave per eight keys): lane = 8 * row slot + piece; the piece is gathered
    // while the sweep works out where the row goes
    const uint32_t slot = wave * 8u + (lane >> 3);
    uint64_t own_key = 0ull;
    uint4 own_piece = make_uint4(0u, 0u, 0u, 0u);
    if (wave < RANK_BLOCK_KEYS / 8) {
      own_key = keys[min(g * RANK_BLOCK_KEYS + slot, count - 1u)];
      own_piece = emit_piece_gather(a, own_key, lane & 7u);
    }
    for (uint32_t base = lo; base < hi; base += 64u * RANK_LOADS) {
      uint64_t other[RANK_LOADS];
      uint64_t inside[RANK_LOADS];
#pragma unroll
      for (int u = 0; u < RANK_LOADS; ++u) {
        const uint32_t i = base + (uint32_t)u * 64u + lane;
        other[u] = keys[min(i, count - 1u)];
        inside[u] = __ballot(i < hi);
      }
#pragma unroll
      for (int u = 0; u < RANK_LOADS; ++u)
#pragma unroll
        for (int k = 0; k < RANK_BLOCK_KEYS; ++k)
          greater[k] += (uint32_t)__popcll(__ballot(other[u] > mine[k]) & inside[u]);
    }
    if (lane == 0) {
#pragma unroll
      for (int k = 0; k < RANK_BLOCK_KEYS; ++k) s_part[wave][k] = greater[k];
    }
    __syncthreads();
    if (wave < RANK_BLOCK_KEYS / 8) {
      uint32_t row = 0u;
#pragma unroll
      for (int w = 0; w < SMALL_WAVES; ++w) row += s_part[w][slot];
      if (g * RANK_BLOCK_KEYS + slot < count) emit_piece_store(a, row, own_key, lane & 7u, own_piece);
    }
    __syncthreads();  // s_part is reused by the next group
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
  const uint32_t *list;  // null: output row r is source row r; else source row list[r]
};

// application_emit_world_source, application.h:617-641: one 144-byte
// VkrWorldDrawCandidate per source row, nine 16-byte pieces: 0 mesh + geometry,
// 1 submesh / material, 2-5 the world matrix, 6 the object id, 7 the local
// sphere, 8 bucket + flags. A block takes EMIT_ROWS consecutive output rows and
// assembles them in LDS: every thread requests its eight 16-byte input pieces
// (two of the identity words, four matrix columns, sphere, candidate word) at
// once - consecutive lanes read consecutive pieces of each input array, so with
// output row = source row = node every wave load covers 1 KiB of consecutive
// bytes - and drops them where they belong in the rows; behind one barrier the
// block copies the rows out as they lie, nine pieces per thread, every wave store
// 1 KiB of consecutive bytes.
// A 16-byte load, marked as streaming when STREAM (read once, not worth a place
// in a cache: for arrays larger than the last-level cache; smaller ones are found
// there and lose by the hint).
template <bool STREAM>
__device__ __forceinline__ float4 load_streaming(const float4 *p) {
  if constexpr (!STREAM) return *p;
  typedef float f32x4_t __attribute__((ext_vector_type(4)));
  const f32x4_t v = __builtin_nontemporal_load(reinterpret_cast<const f32x4_t *>(p));
  return make_float4(v.x, v.y, v.z, v.w);
}
template <bool STREAM>
__device__ __forceinline__ uint4 load_streaming(const uint4 *p) {
  if constexpr (!STREAM) return *p;
  typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
  const u32x4_t v = __builtin_nontemporal_load(reinterpret_cast<const u32x4_t *>(p));
  return make_uint4(v.x, v.y, v.z, v.w);
}

constexpr uint32_t EMIT_ROWS = 256;
constexpr uint32_t EMIT_PIECES = 9;

template <bool STREAM>
__global__ __launch_bounds__(EMIT_ROWS) void k_emit_candidates(CandidateEmitArgs a) {
  __shared__ uint4 s_rows[EMIT_ROWS * EMIT_PIECES];  // 36 KiB: four blocks per CU
  const uint32_t t = threadIdx.x;
  const uint64_t first = (uint64_t)blockIdx.x * EMIT_ROWS;  // < a.n (the grid is sized so)
  const uint32_t rows_here = (uint32_t)min((uint64_t)EMIT_ROWS, a.n - first);
  // Source row of the block's output row r (rows past the end repeat the last one
  // and are not copied out); the transmission stream repeats the rows of the
  // transmissive sources (application.h:643-645).
  auto source = [&](uint32_t r) -> uint64_t {
    const uint64_t out_row = first + min(r, rows_here - 1u);
    return a.list ? (uint64_t)a.list[out_row] : out_row;
  };
  const uint64_t own = source(t);
  uint4 id[2];
  float4 col[4];
#pragma unroll
  for (uint32_t j = 0; j < 2u; ++j) {  // piece q & 1 of the identity words of row q >> 1
    const uint32_t q = t + j * EMIT_ROWS;
    id[j] = load_streaming<STREAM>(a.ident + source(q >> 1) * 2u + (q & 1u));
  }
#pragma unroll
  for (uint32_t j = 0; j < 4u; ++j) {  // column q & 3 of the matrix of row q >> 2
    const uint32_t q = t + j * EMIT_ROWS;
    const uint64_t row = source(q >> 2);
    const uint64_t nd = a.node ? (uint64_t)a.node[row] : row;
    col[j] = load_streaming<STREAM>(a.world + nd * 4u + (q & 3u));
  }
  const float4 sph = load_streaming<STREAM>(a.sphere + own);
  const uint32_t cf = STREAM ? __builtin_nontemporal_load(a.cflags + own) : a.cflags[own];
#pragma unroll
  for (uint32_t j = 0; j < 2u; ++j) {
    const uint32_t q = t + j * EMIT_ROWS;
    uint4 *row = s_rows + (q >> 1) * EMIT_PIECES;
    if (q & 1u) {
      row[1] = make_uint4(id[j].x, id[j].y, id[j].z, 0u);  // submesh_index, material, pad
      row[6] = make_uint4(encode_scene_object_id(id[j].w), 0u, 0u, 0u);
    } else {
      row[0] = id[j];  // mesh, geometry
    }
  }
#pragma unroll
  for (uint32_t j = 0; j < 4u; ++j) {
    const uint32_t q = t + j * EMIT_ROWS;
    s_rows[(q >> 2) * EMIT_PIECES + 2u + (q & 3u)] =
        make_uint4(__float_as_uint(col[j].x), __float_as_uint(col[j].y), __float_as_uint(col[j].z),
                   __float_as_uint(col[j].w));
  }
  s_rows[t * EMIT_PIECES + 7u] = make_uint4(__float_as_uint(sph.x), __float_as_uint(sph.y),
                                            __float_as_uint(sph.z), __float_as_uint(sph.w));
  s_rows[t * EMIT_PIECES + 8u] = make_uint4((cf >> CF_BUCKET_SHIFT) & 0xfu, cf & CF_REF_MASK, 0u, 0u);
  __syncthreads();
  typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
  u32x4 *out4 = reinterpret_cast<u32x4 *>(a.rows + first * EMIT_PIECES);
  const u32x4 *s_rows4 = reinterpret_cast<const u32x4 *>(s_rows);
  const uint32_t pieces_here = rows_here * EMIT_PIECES;
#pragma unroll
  for (uint32_t k = 0; k < EMIT_PIECES; ++k) {
    const uint32_t i = t + k * EMIT_ROWS;
    if (i < pieces_here) {
      if constexpr (STREAM) __builtin_nontemporal_store(s_rows4[i], out4 + i);
      else out4[i] = s_rows4[i];
    }
  }
}

}  // namespace vkr
