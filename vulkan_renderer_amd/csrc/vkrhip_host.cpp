// vkrhip_host.cpp - host-side mirrors of the reference leaf functions that stay
// on the CPU: frustum plane extraction for the view set and the BFS relabelling
// of the hierarchy. Plain C++, no device code. Built with -ffp-contract=off so
// multiply-add is fused only where written (the reference fuses only inside
// mat4_mul, lib/src/math/mat.h:1145-1163).
#include "../../include/vkr_hip_scene.h"

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <vector>

#pragma clang fp contract(off)

namespace {

constexpr float kFloatEpsilon = 1.192092896e-07f;  // lib/src/math/vkr_math.h:154

// vkr_simd_dot3_f32x4, lib/src/math/vkr_simd.h:914-917
inline float dot3(const float *a, const float *b) {
  const float p0 = a[0] * b[0];
  const float p1 = a[1] * b[1];
  const float p2 = a[2] * b[2];
  const float zero = 0.0f;
  return (p0 + p1) + (p2 + zero);
}

// vkr_plane_normalize / vkr_plane_from_vec4, lib/src/math/vkr_frustum.c:12-31
inline void plane_from_vec4(float x, float y, float z, float w, float *plane) {
  float n[3] = {x, y, z};
  float d = w;
  const float len = std::sqrt(dot3(n, n));
  if (len > kFloatEpsilon) {
    const float inv_len = 1.0f / len;
    n[0] = n[0] * inv_len;
    n[1] = n[1] * inv_len;
    n[2] = n[2] * inv_len;
    d *= inv_len;
  }
  plane[0] = n[0];
  plane[1] = n[1];
  plane[2] = n[2];
  plane[3] = d;
}

}  // namespace

extern "C" {

void vkrhip_mat4_mul(const float *a, const float *b, float *out) {
  float tmp[16];
  for (int i = 0; i < 4; ++i) {
    const float x = b[4 * i + 0], y = b[4 * i + 1], z = b[4 * i + 2], w = b[4 * i + 3];
    for (int r = 0; r < 4; ++r) {
      float t = a[r] * x;
      t = __builtin_fmaf(a[4 + r], y, t);
      t = __builtin_fmaf(a[8 + r], z, t);
      tmp[4 * i + r] = __builtin_fmaf(a[12 + r], w, t);
    }
  }
  std::memcpy(out, tmp, sizeof(tmp));
}

void vkrhip_frustum_from_matrix(const float *vp, float *planes) {
  // rows via mat4_row (lib/src/math/mat.h:927-930)
  float r[4][4];
  for (int row = 0; row < 4; ++row)
    for (int c = 0; c < 4; ++c) r[row][c] = vp[row + 4 * c];
  plane_from_vec4(r[3][0] + r[0][0], r[3][1] + r[0][1], r[3][2] + r[0][2], r[3][3] + r[0][3],
                  planes + 0);  // LEFT
  plane_from_vec4(r[3][0] - r[0][0], r[3][1] - r[0][1], r[3][2] - r[0][2], r[3][3] - r[0][3],
                  planes + 4);  // RIGHT
  plane_from_vec4(r[3][0] + r[1][0], r[3][1] + r[1][1], r[3][2] + r[1][2], r[3][3] + r[1][3],
                  planes + 8);  // BOTTOM
  plane_from_vec4(r[3][0] - r[1][0], r[3][1] - r[1][1], r[3][2] - r[1][2], r[3][3] - r[1][3],
                  planes + 12);  // TOP
  plane_from_vec4(r[2][0], r[2][1], r[2][2], r[2][3], planes + 16);  // NEAR (0 <= z)
  plane_from_vec4(r[3][0] - r[2][0], r[3][1] - r[2][1], r[3][2] - r[2][2], r[3][3] - r[2][3],
                  planes + 20);  // FAR
}

void vkrhip_frustum_from_view_projection(const float *view, const float *projection,
                                         float *planes) {
  float vp[16];
  vkrhip_mat4_mul(projection, view, vp);
  vkrhip_frustum_from_matrix(vp, planes);
}

void vkrhip_build_view_set(const float *view, const float *projection,
                           const float *light_view_proj, uint32_t cascades, float *planes,
                           uint32_t *required_flags) {
  vkrhip_frustum_from_view_projection(view, projection, planes);
  required_flags[0] = VKRHIP_CAND_CAMERA_OPAQUE;
  for (uint32_t i = 0; i < cascades; ++i) {
    vkrhip_frustum_from_matrix(light_view_proj + 16 * (size_t)i, planes + 24 * (size_t)(i + 1));
    required_flags[i + 1] = VKRHIP_CAND_SHADOW_CASTER;
  }
}

void vkrhip_effective_visibility(const uint32_t *parent, const uint8_t *spec, uint64_t n,
                                 uint8_t *out_visible) {
  // parents come first, so the answer of the parent is already there
  for (uint64_t i = 0; i < n; ++i) {
    const bool top = parent[i] == VKRHIP_NO_PARENT || parent[i] >= i;
    const uint8_t above = top ? (uint8_t)1 : out_visible[parent[i]];
    const uint8_t sp = spec[i];
    uint8_t v = above;
    if (sp & 1u) v = !(sp & 2u) ? (uint8_t)0 : ((sp & 4u) ? above : (uint8_t)1);
    out_visible[i] = v;
  }
}

uint32_t vkrhip_levelize(const uint32_t *parent, uint64_t n, uint32_t *order,
                         uint32_t *new_parent, uint64_t *level_offsets) {
  if (n == 0) return 0;
  std::vector<uint64_t> first;
  std::vector<uint32_t> children, new_index, level_of;
  std::vector<uint8_t> visited;
  try {
    first.assign(n + 1, 0);
    children.resize(n);
    new_index.resize(n);
    level_of.assign(n, 0);
    visited.assign(n, 0);
  } catch (...) {
    return 0;
  }
  auto is_root = [&](uint64_t i) { return parent[i] == VKRHIP_NO_PARENT || parent[i] >= n; };
  for (uint64_t i = 0; i < n; ++i)
    if (!is_root(i)) first[parent[i] + 1]++;
  for (uint64_t i = 0; i < n; ++i) first[i + 1] += first[i];
  {
    std::vector<uint64_t> cursor(first.begin(), first.end() - 1);
    for (uint64_t i = 0; i < n; ++i)
      if (!is_root(i)) children[cursor[parent[i]]++] = (uint32_t)i;
  }
  uint64_t head = 0, tail = 0;
  uint32_t levels = 1;
  for (uint64_t i = 0; i < n; ++i) {
    if (is_root(i)) {
      order[tail++] = (uint32_t)i;
      visited[i] = 1;
    }
  }
  while (head < tail) {
    const uint32_t e = order[head++];
    for (uint64_t c = first[e]; c < first[e + 1]; ++c) {
      const uint32_t child = children[c];
      if (visited[child]) continue;
      visited[child] = 1;
      level_of[child] = level_of[e] + 1;
      if (level_of[child] + 1 > levels) levels = level_of[child] + 1;
      order[tail++] = child;
    }
  }
  // members of cycles are never reached; the reference appends them after the
  // BFS (vkr_scene_system.c:835-843). They become extra roots of the last level
  // here so that no node reads a parent of its own level.
  const uint64_t reached = tail;
  for (uint64_t i = 0; i < n; ++i) {
    if (!visited[i]) {
      order[tail++] = (uint32_t)i;
      level_of[i] = levels - 1;
    }
  }
  for (uint64_t k = 0; k < n; ++k) new_index[order[k]] = (uint32_t)k;
  for (uint64_t k = 0; k < n; ++k) {
    const uint32_t src = order[k];
    if (k >= reached || is_root(src))
      new_parent[k] = VKRHIP_NO_PARENT;
    else
      new_parent[k] = new_index[parent[src]];
  }
  // BFS order is level order, so level boundaries are where level_of steps up
  uint32_t l = 0;
  level_offsets[0] = 0;
  for (uint64_t k = 0; k < reached; ++k) {
    while (level_of[order[k]] > l) level_offsets[++l] = k;
  }
  while (l + 1 < levels) level_offsets[++l] = reached;
  level_offsets[levels] = n;
  return levels;
}

int vkrhip_shard_hierarchy(const uint32_t *parent, uint64_t n, const uint64_t *level_offsets,
                           uint32_t levels, uint32_t shards, uint32_t replicated_levels,
                           uint64_t *out_ranges) {
  if (!level_offsets || !out_ranges || levels == 0 || shards == 0) return VKRHIP_ERR_ARG;
  if (level_offsets[0] != 0 || level_offsets[levels] != n) return VKRHIP_ERR_ARG;
  // (ascending, hence all within [0, n]: everything below indexes by them)
  for (uint32_t l = 0; l < levels; ++l)
    if (level_offsets[l] > level_offsets[l + 1]) return VKRHIP_ERR_ARG;
  if (levels > 1 && !parent) return VKRHIP_ERR_ARG;
  const uint32_t cut = replicated_levels < levels ? replicated_levels : levels;
  // subtrees are contiguous per level only when parents ascend with their children
  for (uint32_t l = 1; l < levels; ++l) {
    uint32_t last = 0;
    for (uint64_t i = level_offsets[l]; i < level_offsets[l + 1]; ++i) {
      const uint32_t p = parent[i];
      if (p == VKRHIP_NO_PARENT) return VKRHIP_ERR_STATE;  // a root below the top level
      if (p < level_offsets[l - 1] || p >= level_offsets[l] || p < last) return VKRHIP_ERR_STATE;
      last = p;
    }
  }
  auto range = [&](uint32_t s, uint32_t l) { return out_ranges + ((size_t)s * levels + l) * 2u; };
  for (uint32_t s = 0; s < shards; ++s)
    for (uint32_t l = 0; l < cut; ++l) {
      range(s, l)[0] = level_offsets[l];
      range(s, l)[1] = level_offsets[l + 1];
    }
  if (cut == levels) return VKRHIP_OK;
  // weight of every node of the cut level = nodes of its subtree (bottom-up counts)
  std::vector<uint64_t> weight;
  try {
    weight.assign(n - level_offsets[cut], 1);
  } catch (...) {
    return VKRHIP_ERR_CAPACITY;
  }
  const uint64_t base = level_offsets[cut];
  for (uint32_t l = levels - 1; l > cut; --l)
    for (uint64_t i = level_offsets[l]; i < level_offsets[l + 1]; ++i)
      weight[parent[i] - base] += weight[i - base];
  const uint64_t cb = level_offsets[cut], ce = level_offsets[cut + 1];
  uint64_t total = 0;
  for (uint64_t i = cb; i < ce; ++i) total += weight[i - base];
  // boundaries in the cut level: shard s ends where the running weight passes (s+1)/shards
  uint64_t run = 0, at = cb;
  for (uint32_t s = 0; s < shards; ++s) {
    range(s, cut)[0] = at;
    const uint64_t goal = (total * (uint64_t)(s + 1) + shards - 1) / shards;
    // a subtree goes to the shard in which its middle falls; the last shard takes the rest
    while (at < ce && (s + 1 == shards || run + weight[at - base] / 2 < goal)) {
      run += weight[at - base];
      ++at;
    }
    range(s, cut)[1] = at;
  }
  // below the cut: the children of a contiguous range are a contiguous range
  for (uint32_t l = cut + 1; l < levels; ++l) {
    const uint32_t *lp = parent + level_offsets[l];
    const uint64_t count = level_offsets[l + 1] - level_offsets[l];
    for (uint32_t s = 0; s < shards; ++s) {
      const uint64_t pb = range(s, l - 1)[0], pe = range(s, l - 1)[1];
      range(s, l)[0] = level_offsets[l] + (uint64_t)(std::lower_bound(lp, lp + count, pb,
          [](uint32_t a, uint64_t b) { return (uint64_t)a < b; }) - lp);
      range(s, l)[1] = level_offsets[l] + (uint64_t)(std::lower_bound(lp, lp + count, pe,
          [](uint32_t a, uint64_t b) { return (uint64_t)a < b; }) - lp);
    }
  }
  return VKRHIP_OK;
}

}  // extern "C"
