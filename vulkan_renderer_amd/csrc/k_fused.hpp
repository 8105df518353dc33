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
// Inside a round every wave works on its own 64 consecutive nodes and never
// waits for another wave:
//   transform phase  4 lanes per node, lane j owns column j, four sub-rounds of
//                    16 nodes: every wave load/store covers 1 KiB of consecutive
//                    bytes. Below the root level the four lanes of a node fetch
//                    one column each of the parent's world matrix (siblings are
//                    contiguous in the BFS node order, so the 16 nodes of a
//                    sub-round share a couple of parents and the requests merge)
//                    and trade columns with quad DPP moves; the parent index
//                    comes from the lane that loaded it by a wave shuffle;
//   hand-over        the world columns go to the wave's stretch of LDS with an
//                    80-byte node stride (conflict-free 16-byte reads);
//   cull phase       one thread per node reads its four columns back and runs
//                    classify_candidate().
// The only block-wide barriers are the one after the LDS tables are cleared and
// those of the end-of-segment packing.
#pragma once
#include "k_cull.hpp"
#include "k_transform.hpp"

namespace vkr {

constexpr int FUSED_BLOCK = 256;
constexpr int FUSED_ROUNDS = CULL_TILE / FUSED_BLOCK;  // 4
constexpr uint32_t F_HAS_PARENT = 0x80u;  // marker in the prefetched parent flag byte (not a scene flag)

struct FusedArgs {
  TransformArgs t;
  CullArgs c;         // c.n = candidates; candidate i <-> node i
  uint32_t rounds;    // rounds of a tile one block walks: 4, 2 or 1 (more, shorter blocks)
  uint32_t seg_base;  // cull segment of block 0 of this launch (segments follow launch order)
};

// Value of lane (quad base + K) for every lane of the quad (quad_perm DPP move).
template <int K>
__device__ __forceinline__ float quad_bcast(float v) {
  constexpr int ctrl = K | (K << 2) | (K << 4) | (K << 6);
  return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), ctrl, 0xf, 0xf, true));
}
template <int K>
__device__ __forceinline__ float4 quad_bcast4(float4 v) {
  return make_float4(quad_bcast<K>(v.x), quad_bcast<K>(v.y), quad_bcast<K>(v.z), quad_bcast<K>(v.w));
}

// Node a lane works on in sub-round r of a round, relative to the round's first
// node: the wave's 64 nodes in four groups of 16, four lanes per node.
__device__ __forceinline__ uint32_t sub_slot(int r) {
  return (threadIdx.x & ~63u) + (uint32_t)r * 16u + ((threadIdx.x & 63u) >> 2);
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
  uint32_t raw[4];  // node flag bytes as loaded (sub-round mapping)
  uint32_t fp;      // after fused_resolve_head: the four pass-1 flag bytes packed (byte r =
                    // sub-round r), 0 outside the level
  uint32_t rebuild; // after fused_issue_body: bit r = sub-round r rebuilds its local matrix from TRS
  uint32_t p;       // parent of node (round base + thread)
  float4 ls;        // local sphere of candidate (round base + thread)
  uint32_t cf;
  __device__ __forceinline__ uint32_t f(int r) const { return (fp >> (8 * r)) & 0xffu; }
};

__device__ __forceinline__ uint32_t clamp_rel(uint32_t rel, uint32_t lo, uint32_t hi) {
  return min(max(rel, lo), hi - 1u);
}
__device__ __forceinline__ uint32_t in_range_mask(uint32_t rel, uint32_t lo, uint32_t hi) {
  return (rel >= lo && rel < hi) ? 0xffffffffu : 0u;
}

template <bool HAS_PARENT, bool CULL>
__device__ __forceinline__ void fused_issue_head(const FusedArgs &a, uint64_t tile_base,
                                                 uint32_t lo, uint32_t hi, uint32_t rbase,
                                                 FusedHead &h) {
  const uint32_t t = threadIdx.x;
  const uint64_t i = tile_base + clamp_rel(rbase + t, lo, hi);
  h.p = NO_PARENT;
  if (HAS_PARENT) h.p = a.t.parent[i];
  h.ls = make_float4(0.f, 0.f, 0.f, 0.f);
  h.cf = 0u;
  if (CULL) {
    const uint64_t ci = min(i, a.c.n - 1u);
    h.ls = a.c.sphere[ci];
    h.cf = a.c.cflags[ci];
  }
#pragma unroll
  for (int r = 0; r < 4; ++r)
    h.raw[r] = a.t.flags[tile_base + clamp_rel(rbase + sub_slot(r), lo, hi)];
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
    const uint32_t mr = in_range_mask(rbase + sub_slot(r), lo, hi);
    h.fp |= ((((h.raw[r] | a.t.force) & ~(uint32_t)F_WORLD_UPDATED) & mr) & 0xffu) << (8 * r);
  }
}

struct FusedBody {
  // per sub-round: the local column when the node is dirty, the stored world
  // column when it is clean, and one piece of the node's TRS when the local
  // matrix has to be rebuilt (lane 0 position, 1 rotation, 2 scale)
  float4 m[4];
};

// Needs a resolved head. The node's own flags pick what to fetch: dirty -> the
// local column; clean -> the stored world column (the cull phase needs it
// unless the parent turns out to have moved, which is handled when the round is
// computed); TRS changed (pass 1, vkr_scene_system.c:598-602) -> the four lanes
// of the node fetch position, rotation and scale, one 16-byte piece each, and
// the local matrix is rebuilt from them when the round is computed. Either way
// it is one unconditional load per sub-round.
__device__ __forceinline__ void fused_issue_body(const FusedArgs &a, uint64_t tile_base,
                                                 uint32_t lo, uint32_t hi, uint32_t rbase,
                                                 FusedHead &h, FusedBody &b) {
  const uint32_t t = threadIdx.x;
  const uint32_t col = t & 3u;
  const bool have_trs = a.t.pos != nullptr;
  const float4 *piece = col == 1u ? a.t.quat : (col == 2u ? a.t.scale : a.t.pos);
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const uint64_t node = tile_base + clamp_rel(rbase + sub_slot(r), lo, hi);
    const uint32_t f = h.f(r);
    const bool hl = (f & (F_DIRTY_LOCAL | F_DIRTY_WORLD)) != 0;
    const bool rebuilt = have_trs && (f & F_DIRTY_LOCAL) != 0;
    const float4 *src = rebuilt ? piece + node : (hl ? a.t.local : a.t.world) + node * 4u + col;
    b.m[r] = *src;
  }
  // DIRTY_LOCAL becomes DIRTY_WORLD (all four bytes at once); from here on a set
  // DIRTY_WORLD bit also says that b.m holds the local column or, where
  // h.rebuild says so, the piece it is rebuilt from
  const uint32_t dl = h.fp & (0x01010101u * F_DIRTY_LOCAL);
  h.rebuild = have_trs ? ((dl | (dl >> 7) | (dl >> 14) | (dl >> 21)) & 0xfu) : 0u;
  h.fp = (h.fp | (dl << 1)) & ~(0x01010101u * F_DIRTY_LOCAL);
}

// The parents of a wave's 64 nodes. Siblings are contiguous in the BFS node
// order and parents ascend with their children, so the 64 parents of a wave are
// normally a handful of consecutive nodes: when they all lie in a window of 16
// nodes [pmin, pmin + 16) the wave fetches that window with ONE load per lane
// (lane L: column L & 3 of node pmin + (L >> 2); 1 KiB of consecutive bytes) and
// the window's 16 flag bytes with another, a round ahead of their use, like the
// matrices of the round. The columns reach the lanes that need them through
// LDS, the flags by a wave shuffle. A wave whose parents are spread wider
// (ok == false) falls back to one gather per sub-round when the round is
// computed.
struct FusedParent {
  float4 pm;      // window column of this lane
  uint32_t pf;    // flags of node pmin + (lane & 15)
  uint32_t pmin;  // wave-uniform
  bool ok;        // wave-uniform
};

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
  // parents sit before the level, so the window is cut there (and stays inside the arrays)
  const uint32_t last = a.t.begin ? (uint32_t)a.t.begin - 1u : 0u;
  const uint32_t lane = threadIdx.x & 63u;
  par.pm = a.t.world[(uint64_t)min(par.pmin + (lane >> 2), last) * 4u + (lane & 3u)];
  par.pf = a.t.flags[min(par.pmin + (lane & 15u), last)];
}

// The local column of a node whose four lanes hold position, rotation and scale
// (lanes 0, 1, 2 of the quad): scene_compute_local_matrix,
// vkr_scene_system.c:555-561.
__device__ __forceinline__ float4 local_col_from_quad(float4 piece, uint32_t col) {
  const float4 pos = quad_bcast4<0>(piece);
  const float4 quat = quad_bcast4<1>(piece);
  const float4 scale = quad_bcast4<2>(piece);
  return local_col_from_trs(pos, quat, scale, col);
}

// The kernel arguments arrive as 16-dword scalar loads, and the register
// allocator treats every such group as ONE value: when it runs short of scalar
// registers it parks the whole group in lanes of a vector register and brings
// all sixteen dwords back for every use of one pointer. Passing each pointer
// through an empty asm statement gives it a register pair (and a live range) of
// its own, so the groups die right after the loads.
template <typename T>
__device__ __forceinline__ void own_sgprs(T *&p) {
  asm volatile("" : "+s"(p));
}
__device__ __forceinline__ void own_sgprs(uint64_t &v) { asm volatile("" : "+s"(v)); }

template <bool CULL>
__device__ __forceinline__ void split_argument_groups(FusedArgs &a) {
  own_sgprs(a.t.parent);
  own_sgprs(a.t.pos);
  own_sgprs(a.t.quat);
  own_sgprs(a.t.scale);
  own_sgprs(a.t.local);
  own_sgprs(a.t.world);
  own_sgprs(a.t.flags);
  own_sgprs(a.t.counters);
  own_sgprs(a.t.begin);
  own_sgprs(a.t.end);
  if (CULL) {
    own_sgprs(a.c.sphere);
    own_sgprs(a.c.cflags);
    own_sgprs(a.c.planes);
    own_sgprs(a.c.planes_paired);
    own_sgprs(a.c.tie_eps);
    own_sgprs(a.c.required);
    own_sgprs(a.c.view);
    own_sgprs(a.c.tvis);
    own_sgprs(a.c.tkey);
    own_sgprs(a.c.world_bounds);
    own_sgprs(a.c.stage);
    own_sgprs(a.c.seg_counts);
    own_sgprs(a.c.seg_first);
    own_sgprs(a.c.counters);
    own_sgprs(a.c.ties);
    own_sgprs(a.c.n);
    own_sgprs(a.c.stage_stride);
    own_sgprs(a.c.seg_stride);
    own_sgprs(a.c.index_base);
  }
}

template <bool HAS_PARENT, bool TIES, bool BOUNDS_OUT, bool CULL = true>
// Four waves per SIMD: a level of 1M nodes is exactly one resident set of blocks
// (4 per CU), and the register budget of 128 lets the compiler keep the three
// pipeline stages apart.
__global__ __launch_bounds__(FUSED_BLOCK) __attribute__((amdgpu_waves_per_eu(4, 4))) void
k_propagate_classify(FusedArgs a) {
  __shared__ float4 s_mat[CULL ? FUSED_BLOCK * MAT_LDS_STRIDE : 1];  // wave w uses slots [64 w, 64 w + 64)
  __shared__ float4 s_par[HAS_PARENT ? FUSED_BLOCK : 1];  // wave w: its parent window, 64 columns
  __shared__ uint32_t s_updated;
  static_assert(F_DIRTY_LOCAL == 1 && F_DIRTY_WORLD == 2, "flag arithmetic above");
  static_assert((F_DIRTY_LOCAL | F_DIRTY_WORLD | F_WORLD_UPDATED) < F_HAS_PARENT, "marker bit is free");

  // (only where it pays: the root-level and tie-counting variants compile to more
  // register traffic with separate values than with the groups)
  if (HAS_PARENT && !TIES) split_argument_groups<CULL>(a);
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
    if (CULL) segment_empty(a.c, seg);
    return;
  }
  if (t == 0) s_updated = 0u;
  CullShared s{};
  if (CULL) s = cull_shared_init(a.c.views);
  uint32_t packed = 0u;  // per round: state bucket and blend-row visibility of this thread's candidate

  // rounds of the tile that overlap the level (block-uniform)
  auto active = [&](int r) {
    return r < FUSED_ROUNDS && (uint32_t)r * FUSED_BLOCK < hi && (uint32_t)(r + 1) * FUSED_BLOCK > lo;
  };
  // Three-deep software pipeline: while round r is computed, the matrices of
  // round r+1 and the head of round r+2 are in flight.
  FusedHead head_next, head_after;
  FusedBody body_next;
  FusedParent par_next;
  if (active(0)) fused_issue_head<HAS_PARENT, CULL>(a, tile_base, lo, hi, 0u, head_next);
  if (active(1)) fused_issue_head<HAS_PARENT, CULL>(a, tile_base, lo, hi, FUSED_BLOCK, head_after);
  if (active(0)) {
    fused_resolve_head(a, tile_base, lo, hi, 0u, head_next);
    if (HAS_PARENT) fused_issue_parent(a, head_next, par_next);
    fused_issue_body(a, tile_base, lo, hi, 0u, head_next, body_next);
  }
  uint32_t updated_here = 0;
  __syncthreads();  // LDS tables cleared

#pragma unroll
  for (int round = 0; round < FUSED_ROUNDS; ++round) {
    const uint32_t rbase = (uint32_t)round * FUSED_BLOCK;
    const uint64_t base = tile_base + rbase;
    FusedHead hd = head_next;
    FusedBody cur = body_next;
    FusedParent par = par_next;
    head_next = head_after;
    if (active(round + 1)) {
      fused_resolve_head(a, tile_base, lo, hi, rbase + FUSED_BLOCK, head_next);
      // ahead of the matrices: loads return in issue order and the window is needed first
      if (HAS_PARENT) fused_issue_parent(a, head_next, par_next);
      fused_issue_body(a, tile_base, lo, hi, rbase + FUSED_BLOCK, head_next, body_next);
    }
    if (active(round + 2))
      fused_issue_head<HAS_PARENT, CULL>(a, tile_base, lo, hi, rbase + 2 * FUSED_BLOCK, head_after);
    if (!active(round)) continue;

    // ---- transform phase -------------------------------------------------------
    // Parents were finished by the previous level's launch. The wave's parent
    // window arrived with the round's matrices; it is spread over the wave's
    // stretch of LDS so that the four lanes of a node can pick one column each of
    // their parent's world matrix.
    if (HAS_PARENT) {
      if (par.ok) s_par[t] = par.pm;
      __builtin_amdgcn_wave_barrier();  // a wave's LDS accesses stay in order
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const uint32_t slot = sub_slot(r);
      const uint64_t node = base + slot;
      const bool valid = rbase + slot >= lo && rbase + slot < hi;
      uint32_t fr = hd.f(r);
      float4 lcol = cur.m[r];
      if (hd.rebuild & (1u << r)) {  // the same for the four lanes of the node
        lcol = local_col_from_quad(lcol, col);
        if (valid) a.t.local[node * 4u + col] = lcol;
      }
      float4 pcr = make_float4(0.f, 0.f, 0.f, 0.f);
      uint32_t pf = 0u;  // the parent's flags | F_HAS_PARENT, 0 for a root
      if (HAS_PARENT) {
        // the parent index sits in the lane that loaded it for the cull mapping
        const uint32_t par_node = (uint32_t)__shfl((int)hd.p, r * 16 + (int)((t & 63u) >> 2), 64);
        const uint32_t pm = par_node != NO_PARENT ? 0xffffffffu : 0u;
        uint32_t pflag;
        if (par.ok) {  // wave-uniform
          const uint32_t k = (par_node - par.pmin) & 15u;
          pcr = s_par[(t & ~63u) + k * 4u + col];
          pflag = (uint32_t)__shfl((int)par.pf, (int)k, 64);
        } else {
          const uint64_t pidx = par_node & pm;  // roots read node 0 and drop the result
          pcr = a.t.world[pidx * 4u + col];
          pflag = a.t.flags[pidx];
        }
        pf = ((pflag & 0x7fu) | F_HAS_PARENT) & pm & 0xffu;
      }
      // deferred dirty propagation, vkr_scene_system.c:1310-1316
      if (HAS_PARENT && valid && (pf & F_WORLD_UPDATED) && !(fr & F_DIRTY_WORLD)) {
        lcol = a.t.local[node * 4u + col];  // the parent moved: this node moves too
        fr |= F_DIRTY_WORLD;
      }
      float4 wcol = lcol;
      bool did_update = false;
      if (fr & F_DIRTY_WORLD) {  // pass 2, vkr_scene_system.c:1319-1336
        did_update = true;
        fr = (fr & ~(uint32_t)F_DIRTY_WORLD) | F_WORLD_UPDATED;
      }
      if (HAS_PARENT) {
        // all four lanes of the node take part in the column exchange
        const float4 a0 = quad_bcast4<0>(pcr);
        const float4 a1 = quad_bcast4<1>(pcr);
        const float4 a2 = quad_bcast4<2>(pcr);
        const float4 a3 = quad_bcast4<3>(pcr);
        if (did_update && (pf & F_HAS_PARENT)) wcol = mat4_mul_col(a0, a1, a2, a3, lcol);
      }
      if (did_update) a.t.world[node * 4u + col] = wcol;
      if (valid && col == 0) a.t.flags[node] = (uint8_t)fr;
      if (CULL) s_mat[slot * MAT_LDS_STRIDE + col] = wcol;
      updated_here += (did_update && col == 0) ? 1u : 0u;  // per lane; summed when the block ends
    }
    if (!CULL) continue;  // transform only
    __builtin_amdgcn_wave_barrier();  // the wave's own columns: LDS keeps a wave's accesses in order

    // ---- cull phase ------------------------------------------------------------
    Mat4 m;
    {
      const float4 *p = s_mat + t * MAT_LDS_STRIDE;
      m.c[0] = p[0];
      m.c[1] = p[1];
      m.c[2] = p[2];
      m.c[3] = p[3];
    }
    __builtin_amdgcn_wave_barrier();  // read before the next round overwrites the slots
    const bool cvalid = rbase + t >= lo && rbase + t < hi && base + t < a.c.n;
    packed |= classify_candidate<TIES, BOUNDS_OUT>(a.c, s, (uint32_t)round, base + t, cvalid, m,
                                                   hd.ls, hd.cf)
              << (8 * round);
  }
  {
    const uint32_t wave_updates = wave_inclusive_scan(updated_here);
    if ((t & 63u) == 63u && wave_updates) atomicAdd(&s_updated, wave_updates);
  }
  if (CULL)
    segment_finish<TIES>(a.c, s, seg, tile_base + lo, lo, packed);
  else
    __syncthreads();  // s_updated is complete
  if (t == 0 && s_updated) atomicAdd(a.t.counters + stat_shard() + CNT_NODES_UPDATED, s_updated);
}

}  // namespace vkr
