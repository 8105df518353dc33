// k_fused.hpp - transform propagation fused with the multi-view cull.
//
// When candidate i is drawn with node i's matrix (one instance per scene node,
// the layout of the large synthetic scenes) the world matrix a level kernel has
// just produced is classified before it leaves the CU: it is written to HBM once
// (world[] is an output) and never re-read by a separate cull launch. Per node
// the HBM traffic is local 64 B + flags 1 B in, world 64 B + flags 1 B out,
// sphere 16 B + candidate word 4 B in, view-mask word(s) + visibility byte out.
//
// Thread mapping inside a block of 256 nodes:
//   transform phase  4 lanes per node, lane j owns column j, four sub-rounds of
//                    64 nodes; all loads of the four sub-rounds are issued
//                    before the first use;
//   hand-over        the world columns go to LDS with an 80-byte node stride
//                    (conflict-free 16-byte reads);
//   cull phase       one thread per node reads its four columns back and runs
//                    classify_candidate().
// Blocks are aligned to multiples of 256 in the global node index so that every
// block falls into exactly one 1024-candidate count tile.
#pragma once
#include "k_cull.hpp"
#include "k_transform.hpp"

namespace vkr {

constexpr int FUSED_BLOCK = 256;
constexpr int FUSED_WINDOW = 128;  // LDS-staged parent window (nodes)

struct FusedArgs {
  TransformArgs t;
  CullArgs c;  // c.n = candidates; candidate i <-> node i
};

// min and max of `p` over the lanes of the wave that hold a parent index
__device__ __forceinline__ void wave_min_max(uint32_t p, uint32_t &mn, uint32_t &mx) {
  mn = p;                              // NO_PARENT = 0xffffffff never wins the min
  mx = p == NO_PARENT ? 0u : p;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const uint32_t on = (uint32_t)__shfl_xor((int)mn, d, 64);
    const uint32_t ox = (uint32_t)__shfl_xor((int)mx, d, 64);
    mn = on < mn ? on : mn;
    mx = ox > mx ? ox : mx;
  }
}

template <bool HAS_PARENT, bool TIES, bool BOUNDS_OUT>
__global__ __launch_bounds__(FUSED_BLOCK) void k_propagate_classify(FusedArgs a) {
  __shared__ float4 s_mat[FUSED_BLOCK * MAT_LDS_STRIDE];
  __shared__ float4 s_pworld[HAS_PARENT ? FUSED_WINDOW * 4 : 1];
  __shared__ uint8_t s_pflags[HAS_PARENT ? FUSED_WINDOW : 1];
  __shared__ uint32_t s_par[HAS_PARENT ? FUSED_BLOCK : 1];
  __shared__ CullShared s;
  __shared__ uint32_t s_wmin[4], s_wmax[4], s_updated;

  const uint32_t t = threadIdx.x;
  const uint32_t col = t & 3u;
  const uint64_t first_block = a.t.begin / FUSED_BLOCK;
  const uint64_t tile_base = (first_block + blockIdx.x) * FUSED_BLOCK;
  if (t == 0) s_updated = 0u;
  cull_shared_clear(s, a.c.views);

  // ---- every load that depends on nothing is requested first -----------------
  uint32_t p = NO_PARENT;
  if (HAS_PARENT) {
    const uint64_t i = tile_base + t;
    if (i >= a.t.begin && i < a.t.end) p = a.t.parent[i];
  }
  uint8_t f[4];
  bool valid[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const uint64_t node = tile_base + (uint32_t)r * 64u + (t >> 2);
    valid[r] = node >= a.t.begin && node < a.t.end;
    f[r] = valid[r] ? (uint8_t)(a.t.flags[node] | a.t.force) : (uint8_t)0;
  }
  // cull-phase inputs of this thread's node, used after the hand-over
  const uint64_t ci = tile_base + t;
  const bool cvalid = ci >= a.t.begin && ci < a.t.end && ci < a.c.n;
  float4 ls = make_float4(0.f, 0.f, 0.f, 0.f);
  uint32_t cf = 0xf0u;
  if (cvalid) {
    ls = a.c.sphere[ci];
    cf = a.c.cflags[ci];
  }

  // ---- matrix loads that the node's own flags decide ---------------------------
  // own flags dirty -> the local column; clean -> the stored world column (the
  // cull phase needs it unless the parent turns out to have moved, see below)
  float4 lcol[4], wold[4];
  bool have_local[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const uint64_t node = tile_base + (uint32_t)r * 64u + (t >> 2);
    f[r] &= (uint8_t)~F_WORLD_UPDATED;  // pass 1, vkr_scene_system.c:596
    lcol[r] = make_float4(0.f, 0.f, 0.f, 0.f);
    wold[r] = make_float4(0.f, 0.f, 0.f, 0.f);
    have_local[r] = false;
    if (valid[r]) {
      if (f[r] & F_DIRTY_LOCAL) {  // pass 1, vkr_scene_system.c:598-602
        if (a.t.pos != nullptr) {
          lcol[r] = local_col_from_trs(a.t.pos[node], a.t.quat[node], a.t.scale[node], col);
          a.t.local[node * 4u + col] = lcol[r];
        } else {
          lcol[r] = a.t.local[node * 4u + col];
        }
        have_local[r] = true;
        f[r] = (uint8_t)((f[r] & ~F_DIRTY_LOCAL) | F_DIRTY_WORLD);
      } else if (f[r] & F_DIRTY_WORLD) {
        lcol[r] = a.t.local[node * 4u + col];
        have_local[r] = true;
      } else {
        wold[r] = a.t.world[node * 4u + col];
      }
    }
  }

  // ---- parent window ---------------------------------------------------------------
  bool window = false;
  uint32_t pmin = 0;
  if (HAS_PARENT) {
    s_par[t] = p;
    uint32_t wmn, wmx;
    wave_min_max(p, wmn, wmx);
    if ((t & 63u) == 0) {
      s_wmin[t >> 6] = wmn;
      s_wmax[t >> 6] = wmx;
    }
    __syncthreads();
    pmin = min(min(s_wmin[0], s_wmin[1]), min(s_wmin[2], s_wmin[3]));
    const uint32_t pmax = max(max(s_wmax[0], s_wmax[1]), max(s_wmax[2], s_wmax[3]));
    window = pmin != NO_PARENT && (pmax - pmin) < (uint32_t)FUSED_WINDOW;
    if (window) {
      const uint32_t span = pmax - pmin + 1u;
      const float4 *src = a.t.world + (uint64_t)pmin * 4u;
      for (uint32_t k = t; k < span * 4u; k += FUSED_BLOCK) s_pworld[k] = src[k];
      for (uint32_t k = t; k < span; k += FUSED_BLOCK) s_pflags[k] = a.t.flags[(uint64_t)pmin + k];
    }
  }
  __syncthreads();  // window staged, accumulators cleared

  // ---- transform phase ----------------------------------------------------------------
  uint32_t par[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const uint32_t slot = (uint32_t)r * 64u + (t >> 2);
    const uint64_t node = tile_base + slot;
    par[r] = NO_PARENT;
    if (HAS_PARENT) par[r] = s_par[slot];
    // deferred dirty propagation, vkr_scene_system.c:1310-1316
    if (HAS_PARENT && valid[r] && par[r] != NO_PARENT) {
      const uint8_t pf = window ? s_pflags[par[r] - pmin] : a.t.flags[par[r]];
      if (pf & F_WORLD_UPDATED) {
        f[r] |= F_DIRTY_WORLD;
        if (!have_local[r]) lcol[r] = a.t.local[node * 4u + col];  // the parent moved after all
      }
    }
  }
  uint32_t updated_here = 0;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const uint32_t slot = (uint32_t)r * 64u + (t >> 2);
    const uint64_t node = tile_base + slot;
    float4 wcol = wold[r];
    bool did_update = false;
    if (valid[r] && (f[r] & F_DIRTY_WORLD)) {  // pass 2, vkr_scene_system.c:1319-1336
      wcol = lcol[r];
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
        wcol = mat4_mul_col(a0, a1, a2, a3, lcol[r]);
      }
      a.t.world[node * 4u + col] = wcol;
      f[r] = (uint8_t)((f[r] & ~F_DIRTY_WORLD) | F_WORLD_UPDATED);
      did_update = true;
    }
    if (valid[r] && col == 0) a.t.flags[node] = f[r];
    s_mat[slot * MAT_LDS_STRIDE + col] = wcol;
    const unsigned long long m = __ballot(did_update && col == 0);
    if ((t & 63u) == 0) updated_here += (uint32_t)__popcll(m);
  }
  if ((t & 63u) == 0 && updated_here) atomicAdd(&s_updated, updated_here);
  __syncthreads();

  // ---- cull phase -----------------------------------------------------------
  Mat4 m;
  {
    const float4 *p = s_mat + t * MAT_LDS_STRIDE;
    m.c[0] = p[0];
    m.c[1] = p[1];
    m.c[2] = p[2];
    m.c[3] = p[3];
  }
  classify_candidate<TIES, BOUNDS_OUT>(a.c, s, ci, cvalid, m, ls, cf);
  cull_shared_flush<TIES>(a.c, s, tile_base / CULL_TILE);
  if (t == 0 && s_updated) atomicAdd(a.t.counters + stat_shard() + CNT_NODES_UPDATED, s_updated);
}

}  // namespace vkr
