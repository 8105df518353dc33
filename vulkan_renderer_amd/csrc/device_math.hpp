// device_math.hpp - bit-exact restatement of the reference's leaf math for
// gfx950. Every expression keeps the association the reference's x86 path
// evaluates (lib/src/math/mat.h, vec.h, vkr_simd.h); multiply-add is fused only
// where the reference calls its FMA intrinsic. The translation unit is built
// with -ffp-contract=off and the pragma below repeats that for this header.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#pragma clang fp contract(off)

namespace vkr {

struct Mat4 {
  float4 c[4];  // columns
};

__host__ __device__ __forceinline__ float fma_rn(float a, float b, float c) {
  return __builtin_fmaf(a, b, c);
}

// Correctly rounded square root. On the device this relies on
// -fhip-fp32-correctly-rounded-divide-sqrt (hipcc's default, passed explicitly
// by the build): __builtin_sqrtf then expands to the IEEE sequence, whereas
// HIP's __fsqrt_rn maps to the 1-ulp native v_sqrt_f32.
__host__ __device__ __forceinline__ float sqrt_rn(float x) { return __builtin_sqrtf(x); }

// vkr_simd_dot3_f32x4, lib/src/math/vkr_simd.h:914-917: (x*x' + y*y') + (z*z' + 0)
__host__ __device__ __forceinline__ float dot3(float ax, float ay, float az, float bx, float by,
                                               float bz) {
  const float p0 = ax * bx;
  const float p1 = ay * by;
  const float p2 = az * bz;
  const float zero = 0.0f;
  return (p0 + p1) + (p2 + zero);
}

// vec3_length, lib/src/math/vec.h:663-675
__host__ __device__ __forceinline__ float length3(float x, float y, float z) {
  return sqrt_rn(dot3(x, y, z, x, y, z));
}

// vkr_max_f32, lib/src/math/vkr_math.h:206-208
__host__ __device__ __forceinline__ float max_sel(float a, float b) { return (a > b) ? a : b; }

// one result column of mat4_mul, lib/src/math/mat.h:1145-1163:
// t = A.c0*b.x; t = fma(A.c1,b.y,t); t = fma(A.c2,b.z,t); out = fma(A.c3,b.w,t)
__host__ __device__ __forceinline__ float4 mat4_mul_col(const float4 a0, const float4 a1,
                                                        const float4 a2, const float4 a3,
                                                        const float4 b) {
  float4 t;
  t.x = a0.x * b.x;
  t.y = a0.y * b.x;
  t.z = a0.z * b.x;
  t.w = a0.w * b.x;
  t.x = fma_rn(a1.x, b.y, t.x);
  t.y = fma_rn(a1.y, b.y, t.y);
  t.z = fma_rn(a1.z, b.y, t.z);
  t.w = fma_rn(a1.w, b.y, t.w);
  t.x = fma_rn(a2.x, b.z, t.x);
  t.y = fma_rn(a2.y, b.z, t.y);
  t.z = fma_rn(a2.z, b.z, t.z);
  t.w = fma_rn(a2.w, b.z, t.w);
  t.x = fma_rn(a3.x, b.w, t.x);
  t.y = fma_rn(a3.y, b.w, t.y);
  t.z = fma_rn(a3.z, b.w, t.z);
  t.w = fma_rn(a3.w, b.w, t.w);
  return t;
}

// mat4_mul with the left matrix taken one column at a time (k = 0..3): after step
// k every result element holds the first k + 1 terms of the reference's chain
// (one rounded multiply, then fused multiply-adds in k order), so the result is
// the same bit for bit, and only one column of the left matrix has to be in
// registers at a time.
__host__ __device__ __forceinline__ void mat4_mul_step(int k, const float4 ak, const float4 b,
                                                       float4 &acc) {
  const float s = k == 0 ? b.x : (k == 1 ? b.y : (k == 2 ? b.z : b.w));
  if (k == 0) {
    acc.x = ak.x * s;
    acc.y = ak.y * s;
    acc.z = ak.z * s;
    acc.w = ak.w * s;
  } else {
    acc.x = fma_rn(ak.x, s, acc.x);
    acc.y = fma_rn(ak.y, s, acc.y);
    acc.z = fma_rn(ak.z, s, acc.z);
    acc.w = fma_rn(ak.w, s, acc.w);
  }
}

__host__ __device__ __forceinline__ Mat4 mat4_mul(const Mat4 &a, const Mat4 &b) {
  Mat4 r;  // (no loop: a column index that is not a constant would put the matrix in memory)
  r.c[0] = mat4_mul_col(a.c[0], a.c[1], a.c[2], a.c[3], b.c[0]);
  r.c[1] = mat4_mul_col(a.c[0], a.c[1], a.c[2], a.c[3], b.c[1]);
  r.c[2] = mat4_mul_col(a.c[0], a.c[1], a.c[2], a.c[3], b.c[2]);
  r.c[3] = mat4_mul_col(a.c[0], a.c[1], a.c[2], a.c[3], b.c[3]);
  return r;
}

// mat4_mul_vec3 (xyz lanes), lib/src/math/mat.h:1295-1300:
// ((c0*x + c1*y) + c2*z) + c3*1.0f
__host__ __device__ __forceinline__ float3 mat4_mul_point(const Mat4 &m, float x, float y,
                                                          float z) {
  float3 r;
  r.x = ((m.c[0].x * x + m.c[1].x * y) + m.c[2].x * z) + m.c[3].x * 1.0f;
  r.y = ((m.c[0].y * x + m.c[1].y * y) + m.c[2].y * z) + m.c[3].y * 1.0f;
  r.z = ((m.c[0].z * x + m.c[1].z * y) + m.c[2].z * z) + m.c[3].z * 1.0f;
  return r;
}

// z lane of mat4_mul_vec4, lib/src/math/mat.h:1260-1264: (c0*x + c1*y) + (c2*z + c3*w)
__host__ __device__ __forceinline__ float mat4_mul_vec4_z(const Mat4 &m, float x, float y,
                                                          float z, float w) {
  return (m.c[0].z * x + m.c[1].z * y) + (m.c[2].z * z + m.c[3].z * w);
}

// max(max(|col0|, |col1|), |col2|), lib/src/renderer/vkr_visibility.c:42-46.
// The reference takes three square roots and then the maxima; here the maxima
// are taken on the squared lengths and one root follows. The result has the
// same bits: a correctly rounded square root is monotonic, so every `a > b`
// selection picks an operand whose root is the value the reference selects
// (roots that round to the same float are the same float; a NaN operand makes
// both forms fall through to the other operand; zeros of either sign compare
// equal in both forms).
__host__ __device__ __forceinline__ float max_basis_length(const Mat4 &m) {
  const float d0 = dot3(m.c[0].x, m.c[0].y, m.c[0].z, m.c[0].x, m.c[0].y, m.c[0].z);
  const float d1 = dot3(m.c[1].x, m.c[1].y, m.c[1].z, m.c[1].x, m.c[1].y, m.c[1].z);
  const float d2 = dot3(m.c[2].x, m.c[2].y, m.c[2].z, m.c[2].x, m.c[2].y, m.c[2].z);
  return sqrt_rn(max_sel(max_sel(d0, d1), d2));
}

// vkr_quat_to_mat4, lib/src/math/mat.h:1538-1560
__host__ __device__ __forceinline__ Mat4 quat_to_mat4(float4 q) {
  const float x = q.x, y = q.y, z = q.z, w = q.w;
  const float xx = x * x, yy = y * y, zz = z * z;
  const float xy = x * y, xz = x * z, yz = y * z;
  const float wx = w * x, wy = w * y, wz = w * z;
  Mat4 m;
  m.c[0] = make_float4(1.0f - 2.0f * (yy + zz), 2.0f * (xy + wz), 2.0f * (xz - wy), 0.0f);
  m.c[1] = make_float4(2.0f * (xy - wz), 1.0f - 2.0f * (xx + zz), 2.0f * (yz + wx), 0.0f);
  m.c[2] = make_float4(2.0f * (xz + wy), 2.0f * (yz - wx), 1.0f - 2.0f * (xx + yy), 0.0f);
  m.c[3] = make_float4(0.0f, 0.0f, 0.0f, 1.0f);
  return m;
}

// scene_compute_local_matrix, lib/src/renderer/systems/vkr_scene_system.c:555-561:
// mat4_mul(mat4_mul(translate(p), quat_to_mat4(q)), scale(s)). The generic
// product is evaluated in full so that non-finite inputs propagate exactly as
// in the reference (0*inf terms included).
__host__ __device__ __forceinline__ Mat4 local_from_trs(float4 p, float4 q, float4 s) {
  Mat4 t;
  t.c[0] = make_float4(1.0f, 0.0f, 0.0f, 0.0f);
  t.c[1] = make_float4(0.0f, 1.0f, 0.0f, 0.0f);
  t.c[2] = make_float4(0.0f, 0.0f, 1.0f, 0.0f);
  t.c[3] = make_float4(p.x, p.y, p.z, 1.0f);
  Mat4 sc;
  sc.c[0] = make_float4(s.x, 0.0f, 0.0f, 0.0f);
  sc.c[1] = make_float4(0.0f, s.y, 0.0f, 0.0f);
  sc.c[2] = make_float4(0.0f, 0.0f, s.z, 0.0f);
  sc.c[3] = make_float4(0.0f, 0.0f, 0.0f, 1.0f);
  const Mat4 tr = mat4_mul(t, quat_to_mat4(q));
  return mat4_mul(tr, sc);
}

// Column `col` of local_from_trs without materialising the other three: column
// j of (T*R)*S is mat4_mul_col(T*R, S.c[j]) and S.c[j] is built from scalars.
__host__ __device__ __forceinline__ float4 local_col_from_trs(float4 p, float4 q, float4 s,
                                                              uint32_t col) {
  Mat4 t;
  t.c[0] = make_float4(1.0f, 0.0f, 0.0f, 0.0f);
  t.c[1] = make_float4(0.0f, 1.0f, 0.0f, 0.0f);
  t.c[2] = make_float4(0.0f, 0.0f, 1.0f, 0.0f);
  t.c[3] = make_float4(p.x, p.y, p.z, 1.0f);
  const Mat4 r = quat_to_mat4(q);
  const float4 tr0 = mat4_mul_col(t.c[0], t.c[1], t.c[2], t.c[3], r.c[0]);
  const float4 tr1 = mat4_mul_col(t.c[0], t.c[1], t.c[2], t.c[3], r.c[1]);
  const float4 tr2 = mat4_mul_col(t.c[0], t.c[1], t.c[2], t.c[3], r.c[2]);
  const float4 tr3 = mat4_mul_col(t.c[0], t.c[1], t.c[2], t.c[3], r.c[3]);
  const float4 b = make_float4(col == 0u ? s.x : 0.0f, col == 1u ? s.y : 0.0f,
                               col == 2u ? s.z : 0.0f, col == 3u ? 1.0f : 0.0f);
  return mat4_mul_col(tr0, tr1, tr2, tr3, b);
}

// Per-launch tables that every lane reads at the same index (frustum planes,
// view gates, the camera matrix) are read through the constant address space:
// the compiler may then assume no store of the launch aliases them and fetches
// them with scalar loads into SGPRs instead of one vector load per lane.
template <typename T>
using uniform_ptr = const T __attribute__((address_space(4))) *;
template <typename T>
__device__ __forceinline__ uniform_ptr<T> as_uniform(const T *p) {
  return (uniform_ptr<T>)(uintptr_t)p;
}
__device__ __forceinline__ uniform_ptr<float> as_uniform(const float4 *p) {
  return (uniform_ptr<float>)(uintptr_t)p;
}

// vkr_frustum_test_sphere, lib/src/math/vkr_frustum.c:68-84, without the early
// return (the result is the conjunction either way). planes = 6 x (nx,ny,nz,d).
// A NaN distance or radius compares false and keeps the sphere visible.
//
// Two planes are evaluated side by side on two-wide vectors, which the
// compiler maps to the packed fp32 instructions (v_pk_mul_f32, v_pk_add_f32):
// every lane of a packed operation is the same separately rounded IEEE
// operation as its scalar form. The "+ 0" of the reference's dot product is
// left out: adding +0 can only turn a -0 sum into +0, and the distance is only
// compared against -radius, where the sign of a zero makes no difference.
// `paired` holds the six planes two at a time, interleaved per component:
// (nx, nx', ny, ny', nz, nz', d, d') for planes (0,1), (2,3), (4,5), so that
// every packed operand is an aligned pair of consecutive scalars.
typedef float float2v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ bool sphere_visible(uniform_ptr<float> paired, float cx, float cy,
                                               float cz, float radius) {
  const float neg_r = -radius;
  const float2v x = {cx, cx}, y = {cy, cy}, z = {cz, cz};
  float2v dist[3];
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    const float2v nx = {paired[j * 8 + 0], paired[j * 8 + 1]};
    const float2v ny = {paired[j * 8 + 2], paired[j * 8 + 3]};
    const float2v nz = {paired[j * 8 + 4], paired[j * 8 + 5]};
    const float2v d = {paired[j * 8 + 6], paired[j * 8 + 7]};
    dist[j] = ((nx * x + ny * y) + nz * z) + d;
  }
  // "some plane has dist < -radius" is "the smallest distance is < -radius": the
  // minimum of IEEE minNum skips NaN distances exactly as their comparisons would
  // have come out false, and a NaN radius makes the one comparison false as it
  // made all six. Two three-way minima, one two-way, one comparison.
  const float nearest = fminf(fminf(fminf(dist[0].x, dist[0].y), dist[1].x),
                              fminf(fminf(dist[1].y, dist[2].x), dist[2].y));
  return !(nearest < neg_r);
}

}  // namespace vkr
