// k_fused.hpp - transform propagation fused with the multi-view cull.
//
// When candidate i is drawn with node i's matrix (one instance per scene node,
// the layout of the large synthetic scenes) the world matrix a level kernel has
// just produced is classified before it leaves the CU: it is written to HBM once
// (world[] is an output) and never re-read by a separate cull launch. Per node
// the HBM traffic is local 64 B + flags 1 B in, world 64 B + flags 1 B out,
// sphere 16 B + candidate word 4 B in, visibility byte out (+ parent index 4 B
// below the root level) and 4 B per surviving (node, view).
//
// A block owns one 1024-node count tile (aligned to 1024 in the global node
// index) and walks it in four rounds of 256 nodes (small levels: half or a
// quarter of a tile). The matrices of round r+1 and the flags / cull inputs of
// round r+2 are requested before round r is consumed, so a block always has
// more than a round's worth of HBM requests in flight while it computes.
//
// Thread mapping inside a round:
//   transform phase  4 lanes per node, lane j owns column j, four sub-rounds of
//                    64 nodes: every wave load/store covers 1 KiB of consecutive
//                    bytes;
//   hand-over        the world columns go to LDS with an 80-byte node stride
//                    (conflict-free 16-byte reads);
//   cull phase       one thread per node reads its four columns back and runs
//                    classify_candidate().
// Parent world matrices of a round are staged in LDS when the round's parents
// form a window of at most FUSED_WINDOW nodes (siblings are contiguous and
// parents ascend in the BFS node order, so 256 nodes of fan-out >= 2 qualify);
// otherwise lanes read their parent through the cache hierarchy.
#pragma once
#include "k_cull.hpp"
#include "k_transform.hpp"

namespace vkr {

constexpr int FUSED_BLOCK = 256;
constexpr int FUSED_ROUNDS = CULL_TILE / FUSED_BLOCK;  // 4
constexpr int FUSED_WINDOW = 128;                      // LDS-staged parent window (nodes)

struct FusedArgs {
  TransformArgs t;
  CullArgs c;         // c.n = candidates; candidate i <-> node i
  uint32_t rounds;    // rounds of a tile one block walks: 4, 2 or 1 (more, shorter blocks)
  uint32_t seg_base;  // cull segment of block 0 of this launch (segments follow launch order)
};

// min and max of `p` over the lanes of the wave that hold a parent index
__device__ __forceinline__ void wave_min_max(uint32_t p, uint32_t &mn, uint32_t &mx) {
  mn = p;  // NO_PARENT = 0xffffffff never wins the min
  mx = p == NO_PARENT ? 0u : p;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const uint32_t on = (uint32_t)__shfl_xor((int)mn, d, 64);
    const uint32_t ox = (uint32_t)__shfl_xor((int)mx, d, 64);
    mn = on < mn ? on : mn;
    mx = ox > mx ? ox : mx;
  }
}

// What one thread loads for one round, in two dependent steps: the head (flags,
// parent index, cull inputs) depends on nothing; the matrix columns are picked
// by the flags of the head. Both steps are unconditional loads from clamped
// indices whose results are masked with integer arithmetic afterwards (a
// select on a loaded value would be turned back into a branch around the load,
// and every such branch waits for its own load), so all loads of a step are
// issued back to back and the rounds can be software-pipelined. Positions are
// 32-bit offsets into the block's 1024-node tile; [lo, hi) is the part of the
// tile that belongs to the level.
struct FusedHead {
  uint32_t raw[4];  // node flag bytes as loaded
  uint32_t fp;      // after fused_resolve_head: the four pass-1 flag bytes packed (byte r =
                    // sub-round r), 0 outside the level
  __device__ __forceinline__ uint32_t f(int r) const { return (fp >> (8 * r)) & 0xffu; }
  uint32_t p;     // parent of node (round base + thread)
  float4 ls;      // local sphere of candidate (round base + thread)
  uint32_t cf;
};

__device__ __forceinline__ uint32_t clamp_rel(uint32_t rel, uint32_t lo, uint32_t hi) {
  return min(max(rel, lo), hi - 1u);
}
__device__ __forceinline__ uint32_t in_range_mask(uint32_t rel, uint32_t lo, uint32_t hi) {
  return (rel >= lo && rel < hi) ? 0xffffffffu : 0u;
}

template <bool HAS_PARENT>
__device__ __forceinline__ void fused_issue_head(const FusedArgs &a, uint64_t tile_base,
                                                 uint32_t lo, uint32_t hi, uint32_t rbase,
                                                 FusedHead &h) {
  const uint32_t t = threadIdx.x;
  const uint64_t i = tile_base + clamp_rel(rbase + t, lo, hi);
  h.p = NO_PARENT;
  if (HAS_PARENT) h.p = a.t.parent[i];
  const uint64_t ci = min(i, a.c.n - 1u);
  h.ls = a.c.sphere[ci];
  h.cf = a.c.cflags[ci];
#pragma unroll
  for (int r = 0; r < 4; ++r)
    h.raw[r] = a.t.flags[tile_base + clamp_rel(rbase + (uint32_t)r * 64u + (t >> 2), lo, hi)];
}

// Waits for the head's loads. Applies the bulk dirty bits and pass 1's clearing
// of WORLD_UPDATED (vkr_scene_system.c:596) and blanks lanes outside the level.
__device__ __forceinline__ void fused_resolve_head(const FusedArgs &a, uint64_t tile_base,
                                                   uint32_t lo, uint32_t hi, uint32_t rbase,
                                                   FusedHead &h) {
  const uint32_t t = threadIdx.x;
  const uint32_t rel = rbase + t;
  uint32_t m = in_range_mask(rel, lo, hi);
  h.p |= ~m;  // NO_PARENT outside the level
  if (tile_base + rel >= a.c.n) m = 0u;
  h.cf = (h.cf & m) | (0xf0u & ~m);  // invalid bucket: fails every gate
  h.fp = 0u;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const uint32_t mr = in_range_mask(rbase + (uint32_t)r * 64u + (t >> 2), lo, hi);
    h.fp |= ((((h.raw[r] | a.t.force) & ~(uint32_t)F_WORLD_UPDATED) & mr) & 0xffu) << (8 * r);
  }
}

struct FusedBody {
  float4 m[4];  // per sub-round: the local column when the node is dirty, else the stored world column
};

// Needs a resolved head. The node's own flags pick the matrix to fetch: dirty ->
// the local column; clean -> the stored world column (the cull phase needs it
// unless the parent turns out to have moved, which is handled when the parents
// are known).
__device__ __forceinline__ void fused_issue_body(const FusedArgs &a, uint64_t tile_base,
                                                 uint32_t lo, uint32_t hi, uint32_t rbase,
                                                 FusedHead &h, FusedBody &b) {
  const uint32_t t = threadIdx.x;
  const uint32_t col = t & 3u;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const uint64_t node = tile_base + clamp_rel(rbase + (uint32_t)r * 64u + (t >> 2), lo, hi);
    const uint32_t f = h.f(r);
    const bool hl = (f & (F_DIRTY_LOCAL | F_DIRTY_WORLD)) != 0;
    const float4 *src = hl ? a.t.local : a.t.world;
    // a local that is about to be rebuilt from TRS is not fetched: those lanes
    // all read one fixed element instead (a single cached request), which keeps
    // the load unconditional
    const bool rebuilt = a.t.pos != nullptr && (f & F_DIRTY_LOCAL) != 0;
    b.m[r] = src[rebuilt ? a.t.begin * 4u : node * 4u + col];
  }
  // pass 1, vkr_scene_system.c:598-602: a changed TRS rebuilds the local matrix
  if (a.t.pos != nullptr) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      if (h.f(r) & F_DIRTY_LOCAL) {
        const uint64_t node = tile_base + rbase + (uint32_t)r * 64u + (t >> 2);
        b.m[r] = local_col_from_trs(a.t.pos[node], a.t.quat[node], a.t.scale[node], col);
        a.t.local[node * 4u + col] = b.m[r];
      }
    }
  }
  // DIRTY_LOCAL becomes DIRTY_WORLD (all four bytes at once); from here on a set
  // DIRTY_WORLD bit also says that b.m holds the local column
  h.fp = (h.fp | ((h.fp & 0x01010101u * F_DIRTY_LOCAL) << 1)) & ~(0x01010101u * F_DIRTY_LOCAL);
}

template <bool HAS_PARENT, bool TIES, bool BOUNDS_OUT>
__global__ __launch_bounds__(FUSED_BLOCK) void k_propagate_classify(FusedArgs a) {
  __shared__ float4 s_mat[FUSED_BLOCK * MAT_LDS_STRIDE];
  __shared__ float4 s_pworld[HAS_PARENT ? FUSED_WINDOW * 4 : 1];
  __shared__ uint8_t s_pflags[HAS_PARENT ? FUSED_WINDOW : 1];
  __shared__ uint32_t s_par[HAS_PARENT ? FUSED_BLOCK : 1];
  __shared__ uint32_t s_wmin[4], s_wmax[4], s_updated;
  static_assert(F_DIRTY_LOCAL == 1 && F_DIRTY_WORLD == 2, "flag arithmetic above");

  const uint32_t t = threadIdx.x;
  const uint32_t col = t & 3u;
  // Small levels are cut into more blocks of fewer rounds: a lone block per CU
  // cannot hide its own load latency.
  const uint32_t per_tile = (uint32_t)FUSED_ROUNDS / a.rounds;
  const uint64_t tile = a.t.begin / CULL_TILE + blockIdx.x / per_tile;
  const uint64_t tile_base = tile * CULL_TILE;
  const uint32_t first_round = (blockIdx.x % per_tile) * a.rounds;
  // the block's part of the level, as offsets into the tile
  uint32_t lo = a.t.begin > tile_base ? (uint32_t)(a.t.begin - tile_base) : 0u;
  uint32_t hi = (uint32_t)min((uint64_t)CULL_TILE, a.t.end - tile_base);
  lo = max(lo, first_round * FUSED_BLOCK);
  hi = min(hi, (first_round + a.rounds) * FUSED_BLOCK);
  const uint64_t seg = (uint64_t)a.seg_base + blockIdx.x;  // block order is candidate order
  if (lo >= hi) {  // block-uniform
    segment_empty(a.c, seg);
    return;
  }
  if (t == 0) s_updated = 0u;
  CullShared s = cull_shared_init(a.c.views);
  uint32_t packed = 0u;  // per round: state bucket and blend-row visibility of this thread's candidate

  // rounds of the tile that overlap the level (block-uniform)
  auto active = [&](int r) {
    return r < FUSED_ROUNDS && (uint32_t)r * FUSED_BLOCK < hi && (uint32_t)(r + 1) * FUSED_BLOCK > lo;
  };
  // Three-deep software pipeline: while round r is computed, the matrices of
  // round r+1 and the head of round r+2 are in flight.
  FusedHead head_next, head_after;
  FusedBody body_next;
  if (active(0)) fused_issue_head<HAS_PARENT>(a, tile_base, lo, hi, 0u, head_next);
  if (active(1)) fused_issue_head<HAS_PARENT>(a, tile_base, lo, hi, FUSED_BLOCK, head_after);
  if (active(0)) {
    fused_resolve_head(a, tile_base, lo, hi, 0u, head_next);
    fused_issue_body(a, tile_base, lo, hi, 0u, head_next, body_next);
  }
  uint32_t updated_here = 0;

#pragma unroll
  for (int round = 0; round < FUSED_ROUNDS; ++round) {
    const uint32_t rbase = (uint32_t)round * FUSED_BLOCK;
    const uint64_t base = tile_base + rbase;
    FusedHead hd = head_next;
    FusedBody cur = body_next;
    head_next = head_after;
    if (active(round + 1)) {
      fused_resolve_head(a, tile_base, lo, hi, rbase + FUSED_BLOCK, head_next);
      fused_issue_body(a, tile_base, lo, hi, rbase + FUSED_BLOCK, head_next, body_next);
    }
    if (active(round + 2))
      fused_issue_head<HAS_PARENT>(a, tile_base, lo, hi, rbase + 2 * FUSED_BLOCK, head_after);
    if (!active(round)) continue;

    // ---- parent window -------------------------------------------------------
    bool window = false;
    uint32_t pmin = 0;
    if (HAS_PARENT) {
      s_par[t] = hd.p;
      uint32_t wmn, wmx;
      wave_min_max(hd.p, wmn, wmx);
      if ((t & 63u) == 0) {
        s_wmin[t >> 6] = wmn;
        s_wmax[t >> 6] = wmx;
      }
      __syncthreads();  // also: the previous round's readers of s_pworld are done
      pmin = min(min(s_wmin[0], s_wmin[1]), min(s_wmin[2], s_wmin[3]));
      const uint32_t pmax = max(max(s_wmax[0], s_wmax[1]), max(s_wmax[2], s_wmax[3]));
      window = pmin != NO_PARENT && (pmax - pmin) < (uint32_t)FUSED_WINDOW;
      if (window) {
        const uint32_t span = pmax - pmin + 1u;
        const float4 *src = a.t.world + (uint64_t)pmin * 4u;
        for (uint32_t k = t; k < span * 4u; k += FUSED_BLOCK) s_pworld[k] = src[k];
        for (uint32_t k = t; k < span; k += FUSED_BLOCK)
          s_pflags[k] = a.t.flags[(uint64_t)pmin + k];
      }
    }
    __syncthreads();  // window staged; accumulators cleared; s_mat free again

    // ---- transform phase -------------------------------------------------------
    uint32_t par[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const uint32_t slot = (uint32_t)r * 64u + (t >> 2);
      const uint64_t node = base + slot;
      par[r] = NO_PARENT;
      if (HAS_PARENT) par[r] = s_par[slot];
      // deferred dirty propagation, vkr_scene_system.c:1310-1316
      if (HAS_PARENT && par[r] != NO_PARENT) {
        const uint8_t pf = window ? s_pflags[par[r] - pmin] : a.t.flags[par[r]];
        if (pf & F_WORLD_UPDATED) {
          if (!(hd.f(r) & F_DIRTY_WORLD)) cur.m[r] = a.t.local[node * 4u + col];  // parent moved
          hd.fp |= (uint32_t)F_DIRTY_WORLD << (8 * r);
        }
      }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const uint32_t slot = (uint32_t)r * 64u + (t >> 2);
      const uint64_t node = base + slot;
      const bool valid = rbase + slot >= lo && rbase + slot < hi;
      float4 wcol = cur.m[r];
      bool did_update = false;
      uint32_t fr = hd.f(r);
      if (fr & F_DIRTY_WORLD) {  // pass 2, vkr_scene_system.c:1319-1336
        if (HAS_PARENT && par[r] != NO_PARENT) {
          float4 a0, a1, a2, a3;
          if (window) {
            const float4 *pw = s_pworld + (par[r] - pmin) * 4u;
            a0 = pw[0];
            a1 = pw[1];
            a2 = pw[2];
            a3 = pw[3];
          } else {
            const float4 *pw = a.t.world + (uint64_t)par[r] * 4u;
            a0 = pw[0];
            a1 = pw[1];
            a2 = pw[2];
            a3 = pw[3];
          }
          wcol = mat4_mul_col(a0, a1, a2, a3, cur.m[r]);
        }
        a.t.world[node * 4u + col] = wcol;
        fr = (fr & ~(uint32_t)F_DIRTY_WORLD) | F_WORLD_UPDATED;
        did_update = true;
      }
      if (valid && col == 0) a.t.flags[node] = (uint8_t)fr;
      s_mat[slot * MAT_LDS_STRIDE + col] = wcol;
      updated_here += (did_update && col == 0) ? 1u : 0u;  // per lane; summed when the block ends
    }
    __syncthreads();

    // ---- cull phase ------------------------------------------------------------
    Mat4 m;
    {
      const float4 *p = s_mat + t * MAT_LDS_STRIDE;
      m.c[0] = p[0];
      m.c[1] = p[1];
      m.c[2] = p[2];
      m.c[3] = p[3];
    }
    const bool cvalid = rbase + t >= lo && rbase + t < hi && base + t < a.c.n;
    packed |= classify_candidate<TIES, BOUNDS_OUT>(a.c, s, (uint32_t)round, base + t, cvalid, m,
                                                   hd.ls, hd.cf)
              << (8 * round);
  }
  {
    const uint32_t wave_updates = wave_inclusive_scan(updated_here);
    if ((t & 63u) == 63u && wave_updates) atomicAdd(&s_updated, wave_updates);
  }
  segment_finish<TIES>(a.c, s, seg, tile_base + lo, lo, packed);
  if (t == 0 && s_updated) atomicAdd(a.t.counters + stat_shard() + CNT_NODES_UPDATED, s_updated);
}

}  // namespace vkr
