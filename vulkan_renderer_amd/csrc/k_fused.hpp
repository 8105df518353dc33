// k_fused.hpp - transform propagation fused with the multi-view cull.
//
// When candidate i is drawn with node i's matrix (one instance per scene node,
// the layout of the large synthetic scenes) the world matrix a level kernel has
// just produced is classified before it leaves the CU: it is written to HBM once
// (world[] is an output) and never re-read by a separate cull launch. Per node
// the HBM traffic is local 64 B + flags 1 B in, world 64 B + flags 1 B out,
// sphere 16 B + candidate word 4 B in, visibility byte out (+ parent index 4 B
// below the root level) and one bit per (node, view) of survivor masks.
//
// A block owns one 1024-node count tile (aligned to 1024 in the global node
// index) and walks it in four rounds of 256 nodes (small levels: half or a
// quarter of a tile). The matrices of round r+1 and the flags / cull inputs of
// round r+2 are requested before round r is consumed, so a block always has
// more than a round's worth of HBM requests in flight while it computes.
//
// Two lane layouts meet in every round, and a wave only ever works on its own 64
// consecutive nodes, so it never waits for another wave:
//   memory layout   "column lanes": lane L of a wave moves column L & 3 of node
//                   16 r + (L >> 2) of the wave in step r = 0..3, so every wave
//                   load / store of matrices covers 1 KiB of consecutive bytes;
//   compute layout  thread t owns node t of the round: flags, the 4x4 product
//                   with the parent's world matrix and classify_candidate() run
//                   once per node, with the whole matrix in the thread's
//                   registers.
// The wave's stretch of LDS (80-byte node stride: conflict-free 16-byte
// accesses in both layouts) transposes between the two: columns in after the
// loads, columns out before the stores. Below the root level the parents of the
// wave's nodes arrive as one prefetched window (FusedParent).
// The only block-wide barriers are the one after the LDS tables are cleared and
// the one before the segment's masks and counts are stored.
#pragma once
#include "k_cull.hpp"
#include "k_transform.hpp"

namespace vkr {

constexpr int FUSED_BLOCK = 256;
constexpr int FUSED_ROUNDS = CULL_TILE / FUSED_BLOCK;  // 4
constexpr uint32_t F_HAS_PARENT = 0x80u;  // marker in the parent flag byte (not a scene flag)

// One block of the merged top levels (TOP = true): its tile, and where its
// ancestors sit in the flattened ancestor tables. The first levels of a
// hierarchy are small; instead of a launch per level, whose duration is one
// block's chain of dependent loads, ONE launch covers them, and every block
// first recomputes the world matrices of its nodes' ancestors itself - from the
// same inputs with the same operations in the same order as the blocks that own
// those nodes, hence bit for bit the same values - so no block waits for
// another. Ancestors are listed root side first; entry e has the node index
// anc_node[e] and anc_info[e] = the slot of its parent among the block's entries
// (0xffff: a root) | its step in the chain << 16. Everything here is a function
// of the hierarchy alone and is built when the hierarchy is bound.
constexpr uint32_t TOP_MAX_ANCESTORS = 256;  // one thread per ancestor
struct TopDesc {
  uint32_t tile_base;  // first node of the block's 1024-node tile
  uint32_t lo_hi;      // the block's part of the tile: lo | hi << 16
  uint32_t anc_off;    // first entry of the block in anc_node / anc_info
  uint32_t anc_count;
  uint32_t par_first;  // the parents of the block's nodes are the entries
  uint32_t par_slot;   //   [par_slot, ...) = nodes [par_first, ...)
  uint32_t steps;      // chain steps (levels that hold ancestors)
  uint32_t reserved;
};

struct FusedArgs {
  TransformArgs t;
  CullArgs c;         // c.n = candidates; candidate i <-> node i
  uint32_t rounds;    // rounds of a tile one block walks: 4, 2 or 1 (more, shorter blocks)
  uint32_t seg_base;  // cull segment of block 0 of this launch (segments follow launch order)
  // merged top levels only
  const TopDesc *top;
  const uint32_t *top_anc_node;
  const uint32_t *top_anc_info;
  uint32_t top_lds_offset;  // bytes of dynamic LDS ahead of the ancestor table (the cull tables)
  uint32_t top_anc_cap;     // entries the ancestor table is sized for
};

// Node a lane moves a column of in step r of a round, relative to the round's
// first node (memory layout): the wave's 64 nodes in four groups of 16.
__device__ __forceinline__ uint32_t sub_slot(int r) {
  return (threadIdx.x & ~63u) + (uint32_t)r * 16u + ((threadIdx.x & 63u) >> 2);
}


// What one thread loads for one round, in two dependent steps: the head (flags,
// parent index, cull inputs of its own node) depends on nothing; the matrix
// columns are picked by the flags of the head. Both steps are unconditional
// loads from clamped indices whose results are masked with integer arithmetic
// afterwards (a select on a loaded value would be turned back into a branch
// around the load, and every such branch waits for its own load), so all loads
// of a step are issued back to back and the rounds can be software-pipelined.
// Positions are 32-bit offsets into the block's 1024-node tile; [lo, hi) is the
// part of the tile that belongs to the level.
struct FusedHead {
  uint32_t f;    // flag byte of node (round base + thread); after fused_resolve_head the
                 // pass-1 flags, 0 outside the level
  uint32_t p;    // parent of that node
  f32x4 ls;      // local sphere and candidate word of candidate (round base + thread): two
  uint32_t cf;   // rounds ahead like the flags (with the matrices, a round later, the frame
                 // of a 16M-node hierarchy takes 3 % longer)
  bool rebuild;  // after fused_issue_body: the node's local matrix is rebuilt from its TRS
};

__device__ __forceinline__ uint32_t clamp_rel(uint32_t rel, uint32_t lo, uint32_t hi) {
  return min(max(rel, lo), hi - 1u);
}
__device__ __forceinline__ uint32_t in_range_mask(uint32_t rel, uint32_t lo, uint32_t hi) {
  return (rel >= lo && rel < hi) ? 0xffffffffu : 0u;
}

template <bool HAS_PARENT, bool CULL, bool STREAM, bool STEPS = false>
__device__ __forceinline__ void fused_issue_head(const FusedArgs &a, uint64_t tile_base,
                                                 uint32_t lo, uint32_t hi, uint32_t rbase,
                                                 FusedHead &h) {
  const uint32_t rel = clamp_rel(rbase + threadIdx.x, lo, hi);  // < 1024: every offset fits 32 bits
  h.f = load_once<STREAM>(at32(a.t.flags_in + tile_base, rel));
  h.p = NO_PARENT;
  if (HAS_PARENT && STEPS) {
    // the wave's chunk (clamped into the level's part of the tile), lane L word L & 3 of its entry
    const uint32_t chunk = min(max(rbase + (threadIdx.x & ~63u), lo & ~63u), (hi - 1u) & ~63u) >> 6;
    h.p = *at32(a.t.parent_steps + (tile_base >> 6) * 4u, chunk * 4u + (threadIdx.x & 3u));
  } else if (HAS_PARENT) {
    h.p = load_once<STREAM>(at32(a.t.parent + tile_base, rel));
  }
  h.ls = f32x4{0.f, 0.f, 0.f, 0.f};
  h.cf = 0u;
  if (CULL) {
    // (candidate i <-> node i; the tile's first node is a candidate, or the block has none to classify)
    const uint64_t cbase = min(tile_base, a.c.n - 1u);
    const uint32_t crel = (uint32_t)min((uint64_t)rel, a.c.n - 1u - cbase);
    h.ls = load_once<STREAM>(at32(reinterpret_cast<const f32x4 *>(a.c.sphere) + cbase, crel));
    h.cf = load_once<STREAM>(at32(a.c.cflags16 + cbase, crel));
  }
}

// Waits for the head's loads. Applies the bulk dirty bits and pass 1's clearing
// of WORLD_UPDATED (vkr_scene_system.c:596) and blanks lanes outside the level.
template <bool STEPS = false>
__device__ __forceinline__ void fused_resolve_head(const FusedArgs &a, uint64_t tile_base,
                                                   uint32_t lo, uint32_t hi, uint32_t rbase,
                                                   FusedHead &h) {
  const uint32_t rel = rbase + threadIdx.x;
  uint32_t m = in_range_mask(rel, lo, hi);
  if (STEPS) {  // first parent of the chunk + the steps up to this lane
    const uint32_t first = (uint32_t)__builtin_amdgcn_readlane((int)h.p, 0);
    const uint32_t steps_lo = (uint32_t)__builtin_amdgcn_readlane((int)h.p, 1);
    const uint32_t steps_hi = (uint32_t)__builtin_amdgcn_readlane((int)h.p, 2);
    h.p = first + __builtin_amdgcn_mbcnt_hi(steps_hi, __builtin_amdgcn_mbcnt_lo(steps_lo, 0u));
  }
  h.f = ((h.f | a.t.force) & ~(uint32_t)F_WORLD_UPDATED) & m & 0xffu;
  h.p |= ~m;  // NO_PARENT outside the level
}

struct FusedBody {
  // memory layout, per step: the local column when the node is dirty, the stored
  // world column when it is clean, and one piece of the node's TRS when the local
  // matrix has to be rebuilt (column lane 0 position, 1 rotation, 2 scale)
  f32x4 m0, m1, m2, m3;
  template <int R>
  __device__ __forceinline__ f32x4 &at() {
    if constexpr (R == 0) return m0;
    if constexpr (R == 1) return m1;
    if constexpr (R == 2) return m2;
    return m3;
  }
};

// Needs a resolved head. A node's own flags pick what to fetch for it: dirty ->
// the local matrix; clean -> the stored world matrix (the cull phase needs it
// unless the parent turns out to have moved, which is handled when the round is
// computed); TRS changed (pass 1, vkr_scene_system.c:598-602) -> position,
// rotation and scale, from which the local matrix is rebuilt when the round is
// computed. The column lanes learn the flags of the node they fetch for from
// the lane that owns it (a wave shuffle). One unconditional load per step.
template <int R, bool STREAM>
__device__ __forceinline__ void fused_issue_column(const FusedArgs &a, uint64_t tile_base,
                                                   uint32_t lo, uint32_t hi, uint32_t rbase,
                                                   const FusedHead &h, FusedBody &b) {
  const uint32_t t = threadIdx.x;
  const uint32_t col = t & 3u;
  const bool have_trs = a.t.pos != nullptr;
  const float4 *piece = col == 1u ? a.t.quat : (col == 2u ? a.t.scale : a.t.pos);
  const uint64_t node = tile_base + clamp_rel(rbase + sub_slot(R), lo, hi);
  const uint32_t f = (uint32_t)__shfl((int)h.f, R * 16 + (int)((t & 63u) >> 2), 64);
  const bool hl = (f & (F_DIRTY_LOCAL | F_DIRTY_WORLD)) != 0;
  const bool rebuilt = have_trs && (f & F_DIRTY_LOCAL) != 0;
  const float4 *src = rebuilt ? piece + node : (hl ? a.t.local : a.t.world) + node * 4u + col;
  b.at<R>() = load_once<STREAM>(reinterpret_cast<const f32x4 *>(src));
}

template <bool CULL, bool STREAM>
__device__ __forceinline__ void fused_issue_body(const FusedArgs &a, uint64_t tile_base,
                                                 uint32_t lo, uint32_t hi, uint32_t rbase,
                                                 FusedHead &h, FusedBody &b) {
  fused_issue_column<0, STREAM>(a, tile_base, lo, hi, rbase, h, b);
  fused_issue_column<1, STREAM>(a, tile_base, lo, hi, rbase, h, b);
  fused_issue_column<2, STREAM>(a, tile_base, lo, hi, rbase, h, b);
  fused_issue_column<3, STREAM>(a, tile_base, lo, hi, rbase, h, b);
  const bool have_trs = a.t.pos != nullptr;
  // DIRTY_LOCAL becomes DIRTY_WORLD; from here on a set DIRTY_WORLD bit also says
  // that the fetched matrix is the local one or, where h.rebuild says so, the TRS
  h.rebuild = have_trs && (h.f & F_DIRTY_LOCAL) != 0;
  h.f = (h.f | ((h.f & F_DIRTY_LOCAL) << 1)) & ~(uint32_t)F_DIRTY_LOCAL;
}

// The parents of a wave's 64 nodes. Siblings are contiguous in the BFS node
// order and parents ascend with their children, so the 64 parents of a wave are
// normally a handful of consecutive nodes: when they all lie in a window of 16
// nodes [pmin, pmin + 16) the wave fetches that window with ONE load per lane
// (lane L: column L & 3 of node pmin + (L >> 2); 1 KiB of consecutive bytes) and
// the window's 16 flag bytes with another, a round ahead of their use, like the
// matrices of the round. The columns reach the threads that need them through
// LDS, the flags by a wave shuffle. A wave whose parents are spread wider
// (ok == false) falls back to a gather per thread when the round is computed.
struct FusedParent {
  float4 pm;      // window column of this lane
  uint32_t pf;    // flags of node pmin + (lane & 15)
  uint32_t pmin;  // wave-uniform
  bool ok;        // wave-uniform
};

// Largest value of the wave (a scalar), by the moves of wave_inclusive_scan.
__device__ __forceinline__ uint32_t wave_max(uint32_t v) {
  v = max(v, dpp_or_zero<0x111, 0xf>(v));  // row_shr:1
  v = max(v, dpp_or_zero<0x112, 0xf>(v));  // row_shr:2
  v = max(v, dpp_or_zero<0x114, 0xf>(v));  // row_shr:4
  v = max(v, dpp_or_zero<0x118, 0xf>(v));  // row_shr:8
  v = max(v, dpp_or_zero<0x142, 0xa>(v));  // row_bcast:15 into rows 1 and 3
  v = max(v, dpp_or_zero<0x143, 0xc>(v));  // row_bcast:31 into rows 2 and 3
  return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}

// Needs a resolved head (h.p is NO_PARENT for roots and outside the level).
__device__ __forceinline__ void fused_issue_parent(const FusedArgs &a, const FusedHead &h,
                                                   FusedParent &par) {
  // The window starts at the parent of the wave's first node that has one: in
  // BFS order that is the smallest. Whether every other parent falls inside it
  // is one compare and a vote (no reduction across the wave).
  const unsigned long long have = __ballot(h.p != NO_PARENT);
  const int first = have ? __ffsll((long long)have) - 1 : 0;
  const uint32_t p0 = (uint32_t)__builtin_amdgcn_readlane((int)h.p, first);
  par.pmin = have ? p0 : 0u;
  const unsigned long long inside = __ballot(h.p - par.pmin < 16u);
  par.ok = (inside | ~have) == ~0ull;
  // The window is cut behind the wave's last parent (it also stays inside the
  // arrays that way): the lanes beyond it repeat that parent's columns, which
  // costs no memory traffic - with a fan-out of 8 the wave's 64 nodes hang from 8
  // or 9 parents, and a full window of 16 would fetch twice what is needed.
  const uint32_t last = wave_max(h.p != NO_PARENT ? h.p : 0u);
  const uint32_t lane = threadIdx.x & 63u;
  par.pm = a.t.world[(uint64_t)min(par.pmin + (lane >> 2), last) * 4u + (lane & 3u)];
  par.pf = a.t.flags_out[min(par.pmin + (lane & 15u), last)];  // as the parents' launch left them
}

// Moves one matrix per thread (compute layout) out to global memory in the
// memory layout, through the wave's stretch of LDS: every wave store covers
// 1 KiB of consecutive bytes. `want` marks the wave's threads whose matrix is
// to be stored (wave-uniform value, bit = lane).
template <bool STREAM>
__device__ __forceinline__ void store_matrices(float4 *s_mat, float4 *dst, uint64_t round_base,
                                               float4 c0, float4 c1, float4 c2, float4 c3,
                                               unsigned long long want) {
  const uint32_t t = threadIdx.x;
  const uint32_t col = t & 3u;
  float4 *own = s_mat + t * MAT_LDS_STRIDE;
  own[0] = c0;
  own[1] = c1;
  own[2] = c2;
  own[3] = c3;
  __builtin_amdgcn_wave_barrier();  // a wave's LDS accesses stay in order
  // step r is step 0 plus a constant: one address register per array, the rest
  // goes into the instructions' offset fields
  const uint32_t slot0 = sub_slot(0);
  const float4 *in0 = s_mat + slot0 * MAT_LDS_STRIDE + col;
  float4 *out0 = at32(dst + round_base * 4u, slot0 * 4u + col);
  const uint32_t lane_bit = (t & 63u) >> 2;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const float4 v = in0[r * 16 * MAT_LDS_STRIDE];
    const uint32_t chunk = (uint32_t)(want >> (16 * r)) & 0xffffu;  // scalar
    if ((chunk >> lane_bit) & 1u)
      store_once<STREAM>(reinterpret_cast<f32x4 *>(out0 + r * 16 * 4), f32x4{v.x, v.y, v.z, v.w});
  }
  __builtin_amdgcn_wave_barrier();
}

// The kernel arguments arrive as 16-dword scalar loads, and the register
// allocator treats every such group as ONE value: when it runs short of scalar
// registers it parks the whole group in lanes of a vector register and brings
// all sixteen dwords back for every use of one pointer. Passing each pointer
// through an empty asm statement gives it a register pair (and a live range) of
// its own, so the groups die right after the loads.
// (The statement hides where the value came from, and with it the fact that it is
// a pointer to global memory; the address space is stated again afterwards, or
// every access through it would become a flat instruction, which counts on both
// memory counters and defeats the partial waits of the software pipeline.)
template <typename T>
__device__ __forceinline__ void own_sgprs(T *&p) {
  uint64_t v = (uint64_t)p;
  asm volatile("" : "+s"(v));
  p = (T *)(T __attribute__((address_space(1))) *)v;
}
__device__ __forceinline__ void own_sgprs(uint64_t &v) { asm volatile("" : "+s"(v)); }

template <bool CULL, bool STEPS>
__device__ __forceinline__ void split_argument_groups(FusedArgs &a) {
  if (STEPS) own_sgprs(a.t.parent_steps);
  else own_sgprs(a.t.parent);
  own_sgprs(a.t.pos);
  own_sgprs(a.t.quat);
  own_sgprs(a.t.scale);
  own_sgprs(a.t.local);
  own_sgprs(a.t.world);
  own_sgprs(a.t.flags_in);
  own_sgprs(a.t.flags_out);
  own_sgprs(a.t.counters);
  own_sgprs(a.t.begin);
  own_sgprs(a.t.end);
  if (CULL) {
    own_sgprs(a.c.sphere);
    own_sgprs(a.c.cflags16);
    own_sgprs(a.c.planes);
    own_sgprs(a.c.planes_paired);
    own_sgprs(a.c.tie_eps);
    own_sgprs(a.c.required);
    own_sgprs(a.c.view);
    own_sgprs(a.c.tvis);
    own_sgprs(a.c.tkey);
    own_sgprs(a.c.tworld_sparse);
    own_sgprs(a.c.world_bounds);
    own_sgprs(a.c.seg_masks);
    own_sgprs(a.c.seg_counts);
    own_sgprs(a.c.seg_first);
    own_sgprs(a.c.counters);
    own_sgprs(a.c.ties);
    own_sgprs(a.c.caster_bounds);
    own_sgprs(a.c.n);
    own_sgprs(a.c.seg_stride);
    own_sgprs(a.c.index_base);
  }
}

// Waves per SIMD. Four below the root level: a level of 1M nodes is exactly one
// resident set of blocks (4 per CU), and the register budget of 128 lets the
// compiler keep the three pipeline stages apart. The root-level variant has no
// parent window and no matrix product to hold and fits five waves, which pays
// when many views make the classification the longer part of a round; without
// the cull phase it needs 71 registers and fits six (8M roots: 197 -> 193 us).
// ROOM: the root-level cull variant at four waves as well. A frame whose tail runs
// beside the NEXT frame's launch needs that launch to leave room on every SIMD for a
// wave of the tail's kernels (up to 124 registers): five waves of 91 registers
// leave 57, four leave 148.
constexpr int fused_waves(bool has_parent, bool cull, bool room = false) {
  return has_parent ? 4 : (cull ? (room ? 4 : 5) : 6);
}
// STAGED: the block expands its segment's survivors itself (segment_finish_staged).
// KEEP (streaming variants): the launch's nodes have children in a later launch of
// the frame, which reads their world matrices again as its parents': those are
// stored without the streaming hint and a part of them is still in the last-level
// cache when the children ask (16M-node hierarchy, 2M parents of the last level:
// 557.7 -> 553.6 us; a compile-time variant, because a run-time choice of the store
// inside store_matrices costs the kernel 85 us of its schedule). The children
// walking their tiles from the last to the first - the parents written last asked
// for first - gives the gain away again (558.7 us).
// STEPS: the level's parents come as a first parent and a step mask per 64 nodes
// (TransformArgs::parent_steps) instead of an index per node: 4 bytes per node less
// to read.
template <bool HAS_PARENT, bool TIES, bool BOUNDS_OUT, bool CULL = true, bool TOP = false,
          bool STREAM = false, bool STAGED = false, bool ROOM = false, bool KEEP = false,
          bool STEPS = false>
__global__ __launch_bounds__(FUSED_BLOCK)
    __attribute__((amdgpu_waves_per_eu(fused_waves(HAS_PARENT, CULL, ROOM),
                                       fused_waves(HAS_PARENT, CULL, ROOM)))) void
k_propagate_classify(FusedArgs a) {
  __shared__ float4 s_mat[FUSED_BLOCK * MAT_LDS_STRIDE];  // wave w uses slots [64 w, 64 w + 64)
  __shared__ float4 s_par[HAS_PARENT && !TOP ? FUSED_BLOCK : 1];  // wave w: its parent window, 64 columns
  static_assert(!TOP || HAS_PARENT, "the merged top levels hold nodes with parents");
  static_assert(!STAGED || (CULL && !TIES), "staged compaction: cull variants without tie counting");
  __shared__ uint32_t s_updated;
  static_assert(F_DIRTY_LOCAL == 1 && F_DIRTY_WORLD == 2, "flag arithmetic above");
  static_assert((F_DIRTY_LOCAL | F_DIRTY_WORLD | F_WORLD_UPDATED) < F_HAS_PARENT, "marker bit is free");

  // (only where it pays: the root-level and tie-counting variants compile to more
  // register traffic with separate values than with the groups)
  if (HAS_PARENT && !TIES) split_argument_groups<CULL, STEPS>(a);
  const uint32_t t = threadIdx.x;
  const uint32_t col = t & 3u;
  uint64_t tile_base;
  uint32_t lo, hi;  // the block's part of the level, as offsets into the tile
  TopDesc top{};
  if (TOP) {
    top = a.top[blockIdx.x];  // block-uniform: scalar loads
    tile_base = top.tile_base;
    lo = top.lo_hi & 0xffffu;
    hi = top.lo_hi >> 16;
  } else {
    // Small levels are cut into more blocks of fewer rounds: a lone block per CU
    // cannot hide its own load latency.
    const uint32_t per_tile = (uint32_t)FUSED_ROUNDS / a.rounds;
    const uint64_t tile = a.t.begin / CULL_TILE + blockIdx.x / per_tile;
    tile_base = tile * CULL_TILE;
    const uint32_t first_round = (blockIdx.x % per_tile) * a.rounds;
    lo = a.t.begin > tile_base ? (uint32_t)(a.t.begin - tile_base) : 0u;
    hi = (uint32_t)min((uint64_t)CULL_TILE, a.t.end - tile_base);
    lo = max(lo, first_round * FUSED_BLOCK);
    hi = min(hi, (first_round + a.rounds) * FUSED_BLOCK);
  }
  const uint64_t seg = (uint64_t)a.seg_base + blockIdx.x;  // block order is candidate order
  if (lo >= hi) {  // block-uniform
    if (CULL) {
      if (STAGED) segment_empty_staged(a.c, seg);
      else segment_empty(a.c, seg);
    }
    return;
  }
  if (t == 0) s_updated = 0u;
  CullShared s{};
  if (CULL) s = cull_shared_init(a.c.views);

  // rounds that overlap the level (block-uniform)
  auto active = [&](int r) {
    return r < FUSED_ROUNDS && (uint32_t)r * FUSED_BLOCK < hi && (uint32_t)(r + 1) * FUSED_BLOCK > lo;
  };
  // Three-deep software pipeline: while round r is computed, the matrices of
  // round r+1 and the head of round r+2 are in flight.
  FusedHead head_next, head_after;
  FusedBody body_next;
  FusedParent par_next;
  if (active(0)) fused_issue_head<HAS_PARENT, CULL, STREAM, STEPS>(a, tile_base, lo, hi, 0u, head_next);
  if (active(1)) fused_issue_head<HAS_PARENT, CULL, STREAM, STEPS>(a, tile_base, lo, hi, FUSED_BLOCK, head_after);
  // merged top levels: thread e < anc_count looks after ancestor entry e
  float4 *anc_world = nullptr;   // [entry][4]: the ancestors' world matrices
  uint32_t *anc_moved = nullptr;  // [entry]: the ancestor's world matrix changes this frame
  uint32_t anc_node = 0u, anc_info = 0u, anc_f = 0u;
  const bool anc_live = TOP && t < top.anc_count;
  if (TOP) {
    extern __shared__ unsigned long long fused_dynamic_lds[];  // the cull tables come first
    anc_world = reinterpret_cast<float4 *>(reinterpret_cast<char *>(fused_dynamic_lds) + a.top_lds_offset);
    anc_moved = reinterpret_cast<uint32_t *>(anc_world + (size_t)a.top_anc_cap * 4u);
    const uint32_t e = top.anc_off + min(t, max(top.anc_count, 1u) - 1u);
    anc_node = a.top_anc_node[e];
    anc_info = a.top_anc_info[e];
    anc_f = a.t.flags_in[anc_live ? anc_node : 0u];
  }
  if (active(0)) {
    fused_resolve_head<STEPS>(a, tile_base, lo, hi, 0u, head_next);
    if (HAS_PARENT && !TOP) fused_issue_parent(a, head_next, par_next);
    fused_issue_body<CULL, STREAM>(a, tile_base, lo, hi, 0u, head_next, body_next);
  }
  uint32_t updated_here = 0;
  if (TOP) {
    // The ancestor's matrix, picked by its flags like a node's own: position /
    // rotation / scale (in x0-x2) when its TRS changed, the local matrix when it
    // is dirty, the stored world matrix when it is clean.
    anc_f = anc_live ? ((anc_f | a.t.force) & ~(uint32_t)F_WORLD_UPDATED) & 0xffu : 0u;
    const bool have_trs = a.t.pos != nullptr;
    const bool rebuild = have_trs && (anc_f & F_DIRTY_LOCAL) != 0;
    const bool dirty = (anc_f & (F_DIRTY_LOCAL | F_DIRTY_WORLD)) != 0;
    const uint64_t g = anc_live ? anc_node : 0u;
    const float4 *m = (dirty ? a.t.local : a.t.world) + g * 4u;
    float4 x0 = *(rebuild ? a.t.pos + g : m);
    float4 x1 = *(rebuild ? a.t.quat + g : m + 1);
    float4 x2 = *(rebuild ? a.t.scale + g : m + 2);
    float4 x3 = m[3];
    if (rebuild) {  // scene_compute_local_matrix (the owner of the node stores it)
      const Mat4 rebuilt_local = local_from_trs(x0, x1, x2);
      x0 = rebuilt_local.c[0];
      x1 = rebuilt_local.c[1];
      x2 = rebuilt_local.c[2];
      x3 = rebuilt_local.c[3];
    }
    bool moved = dirty;
    const uint32_t pslot = anc_info & 0xffffu;
    const uint32_t step = anc_info >> 16;
    __syncthreads();  // LDS tables cleared
    // the chain, root side first: vkr_scene_update pass 2 for the ancestors
    for (uint32_t s2 = 0; s2 < top.steps; ++s2) {  // block-uniform
      if (anc_live && step == s2) {
        if (pslot != 0xffffu) {
          if (anc_moved[pslot] != 0u && !moved) {  // deferred dirty propagation
            const float4 *l = a.t.local + g * 4u;
            x0 = l[0];
            x1 = l[1];
            x2 = l[2];
            x3 = l[3];
            moved = true;
          }
          if (moved) {
            const float4 *w = anc_world + pslot * 4u;
            const float4 p0 = w[0], p1 = w[1], p2 = w[2], p3 = w[3];
            const float4 y0 = mat4_mul_col(p0, p1, p2, p3, x0);
            const float4 y1 = mat4_mul_col(p0, p1, p2, p3, x1);
            const float4 y2 = mat4_mul_col(p0, p1, p2, p3, x2);
            const float4 y3 = mat4_mul_col(p0, p1, p2, p3, x3);
            x0 = y0;
            x1 = y1;
            x2 = y2;
            x3 = y3;
          }
        }
        float4 *w = anc_world + t * 4u;
        w[0] = x0;
        w[1] = x1;
        w[2] = x2;
        w[3] = x3;
        anc_moved[t] = moved ? 1u : 0u;
      }
      __syncthreads();
    }
  } else {
    __syncthreads();  // LDS tables cleared
  }

#pragma unroll
  for (int round = 0; round < FUSED_ROUNDS; ++round) {
    const uint32_t rbase = (uint32_t)round * FUSED_BLOCK;
    const uint64_t base = tile_base + rbase;
    FusedHead hd = head_next;
    FusedBody cur = body_next;
    FusedParent par = par_next;
    head_next = head_after;
    if (active(round + 1)) {
      fused_resolve_head<STEPS>(a, tile_base, lo, hi, rbase + FUSED_BLOCK, head_next);
      // ahead of the matrices: loads return in issue order and the window is needed first
      if (HAS_PARENT && !TOP) fused_issue_parent(a, head_next, par_next);
      fused_issue_body<CULL, STREAM>(a, tile_base, lo, hi, rbase + FUSED_BLOCK, head_next, body_next);
    }
    if (active(round + 2))
      fused_issue_head<HAS_PARENT, CULL, STREAM, STEPS>(a, tile_base, lo, hi, rbase + 2u * FUSED_BLOCK, head_after);
    if (!active(round)) continue;

    // ---- columns in: memory layout -> compute layout ----------------------------
    {
      f32x4 *in0 = reinterpret_cast<f32x4 *>(s_mat + sub_slot(0) * MAT_LDS_STRIDE + col);
      in0[0 * 16 * MAT_LDS_STRIDE] = cur.m0;
      in0[1 * 16 * MAT_LDS_STRIDE] = cur.m1;
      in0[2 * 16 * MAT_LDS_STRIDE] = cur.m2;
      in0[3 * 16 * MAT_LDS_STRIDE] = cur.m3;
    }
    if (HAS_PARENT && !TOP) {
      if (par.ok) s_par[t] = par.pm;
    }
    __builtin_amdgcn_wave_barrier();  // a wave's LDS accesses stay in order
    // the node's local matrix when it is dirty, its stored world matrix when it is
    // clean, position / rotation / scale in l0-l2 when hd.rebuild
    float4 l0, l1, l2, l3;
    {
      const float4 *own = s_mat + t * MAT_LDS_STRIDE;
      l0 = own[0];
      l1 = own[1];
      l2 = own[2];
      l3 = own[3];
    }
    __builtin_amdgcn_wave_barrier();  // read before anything below reuses the slots

    // ---- transform, one thread per node ------------------------------------------
    const uint64_t node = base + t;
    const bool valid = rbase + t >= lo && rbase + t < hi;
    uint32_t fr = hd.f;
    // pass 1: scene_compute_local_matrix, vkr_scene_system.c:555-561,598-602
    if (hd.rebuild) {
      const Mat4 rebuilt_local = local_from_trs(l0, l1, l2);
      l0 = rebuilt_local.c[0];
      l1 = rebuilt_local.c[1];
      l2 = rebuilt_local.c[2];
      l3 = rebuilt_local.c[3];
    }
    {
      const unsigned long long rebuilt = __ballot(hd.rebuild);  // outside the level: never set
      if (rebuilt != 0ull) store_matrices<STREAM>(s_mat, a.t.local, base, l0, l1, l2, l3, rebuilt);  // wave-uniform
    }
    // Parents were finished by the previous level's launch, or recomputed by this
    // block (merged top levels).
    uint32_t pf = 0u;  // the parent's flags | F_HAS_PARENT, 0 for a root
    uint32_t pk = 0u;  // the parent's place in the wave's window / the block's ancestor table
    if (TOP) {
      if (hd.p != NO_PARENT) {
        pk = top.par_slot + (hd.p - top.par_first);
        pf = F_HAS_PARENT | (anc_moved[pk] != 0u ? (uint32_t)F_WORLD_UPDATED : 0u);
      }
    } else if (HAS_PARENT) {
      const uint32_t pm = hd.p != NO_PARENT ? 0xffffffffu : 0u;
      uint32_t pflag;
      if (par.ok) {  // wave-uniform
        pk = (hd.p - par.pmin) & 15u;
        pflag = (uint32_t)__shfl((int)par.pf, (int)pk, 64);
      } else {
        pflag = a.t.flags_out[hd.p & pm];  // roots read node 0 and drop the result
      }
      pf = ((pflag & 0x7fu) | F_HAS_PARENT) & pm & 0xffu;
    }
    // deferred dirty propagation, vkr_scene_system.c:1310-1316
    if (HAS_PARENT && valid && (pf & F_WORLD_UPDATED) && !(fr & F_DIRTY_WORLD)) {
      const float4 *l = a.t.local + node * 4u;  // the parent moved: this node moves too
      l0 = l[0];
      l1 = l[1];
      l2 = l[2];
      l3 = l[3];
      fr |= F_DIRTY_WORLD;
    }
    bool did_update = false;
    if (fr & F_DIRTY_WORLD) {  // pass 2, vkr_scene_system.c:1319-1336
      did_update = true;
      fr = (fr & ~(uint32_t)F_DIRTY_WORLD) | F_WORLD_UPDATED;
    }
    float4 w0 = l0, w1 = l1, w2 = l2, w3 = l3;
    if (HAS_PARENT && did_update && (pf & F_HAS_PARENT)) {  // mat4_mul(parent.world, local)
      // one column of the parent's matrix at a time (see mat4_mul_step); a loop per
      // source, so that each load knows its address space
      auto times_local = [&](const float4 *parent_world) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const float4 pcol = parent_world[k];
          mat4_mul_step(k, pcol, l0, w0);
          mat4_mul_step(k, pcol, l1, w1);
          mat4_mul_step(k, pcol, l2, w2);
          mat4_mul_step(k, pcol, l3, w3);
        }
      };
      if (TOP) {
        times_local(anc_world + pk * 4u);
      } else if (par.ok) {  // wave-uniform
        times_local(s_par + (t & ~63u) + pk * 4u);
      } else {
        times_local(a.t.world + (uint64_t)hd.p * 4u);
      }
    }
    if (valid) *at32(a.t.flags_out + base, t) = (uint8_t)fr;
    {
      const unsigned long long moved = __ballot(did_update);
      if (moved != 0ull) store_matrices<STREAM && !KEEP>(s_mat, a.t.world, base, w0, w1, w2, w3, moved);  // wave-uniform
    }
    updated_here += did_update ? 1u : 0u;  // per lane; summed when the block ends
    if (!CULL) continue;  // transform only

    // ---- cull phase ------------------------------------------------------------
    const bool cvalid = valid && base + t < a.c.n;
    const uint32_t cf = cvalid ? hd.cf : 0xf0u;  // invalid bucket: fails every gate
    Mat4 world;
    world.c[0] = w0;
    world.c[1] = w1;
    world.c[2] = w2;
    world.c[3] = w3;
    classify_candidate<TIES, BOUNDS_OUT>(a.c, s, (uint32_t)round, base + t, cvalid, world,
                                         make_float4(hd.ls.x, hd.ls.y, hd.ls.z, hd.ls.w), cf);
  }
  if (CULL) {
    if (STAGED) segment_finish_staged(a.c, s, seg, (uint32_t)tile_base);
    else segment_finish<TIES>(a.c, s, seg, (uint32_t)tile_base);
  }
  {
    const uint32_t wave_updates = wave_inclusive_scan(updated_here);
    if ((t & 63u) == 63u && wave_updates) atomicAdd(&s_updated, wave_updates);
  }
  __syncthreads();  // s_updated is complete
  if (t == 0 && s_updated) atomicAdd(a.t.counters + stat_shard() + CNT_NODES_UPDATED, s_updated);
}

}  // namespace vkr
