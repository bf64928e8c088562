// pack.h - element functions of the three weight layouts, shared by the per-tensor packing kernels (conv2d.hip, winograd.hip,
// winograd_f4.hip) and the multi-tensor launch of the training path (pack.hip).
#pragma once
#include "common.h"

namespace edvr {

typedef float pk_f32x4 __attribute__((ext_vector_type(4)));
typedef float pk_f32x2 __attribute__((ext_vector_type(2)));

// w (co, ci, k, k) [or the data-gradient kernel w'(ci, co) flipped when transpose_flip] at (o, c, tap t); 0 in the padding
__device__ __forceinline__ float pack_src(const float *__restrict__ w, int co, int ci, int kk, int o, int c, int t, int transpose_flip) {
  if (o >= co || c >= ci) return 0.f;
  return transpose_flip ? w[((int64_t)c * co + o) * kk + (kk - 1 - t)] : w[((int64_t)o * ci + c) * kk + t];
}

// direct layout [cip][kk][cop], element i
__device__ __forceinline__ void pack_direct_elem(const float *__restrict__ w, float *__restrict__ wpk, int64_t i, int co, int ci, int kk, int cop,
                                                 int transpose_flip) {
  const int o = (int)(i % cop), t = (int)((i / cop) % kk), c = (int)(i / ((int64_t)cop * kk));
  wpk[i] = pack_src(w, co, ci, kk, o, c, t, transpose_flip);
}

// F(2x2,3x3): U[c][xi][o] = (G g G^T)[xi] for element i = c * cop + o of [cip][16][cop]
__device__ __forceinline__ void pack_u2_elem(const float *__restrict__ w, float *__restrict__ U, int64_t i, int co, int ci, int cop, int transpose_flip) {
  const int o = (int)(i % cop), c = (int)(i / cop);
  float g[9];
#pragma unroll
  for (int t = 0; t < 9; ++t) g[t] = pack_src(w, co, ci, 9, o, c, t, transpose_flip);
  float tmp[12];
#pragma unroll
  for (int jx = 0; jx < 3; ++jx) {  // G g
    tmp[0 * 3 + jx] = g[0 * 3 + jx];
    tmp[1 * 3 + jx] = 0.5f * (g[0 * 3 + jx] + g[1 * 3 + jx] + g[2 * 3 + jx]);
    tmp[2 * 3 + jx] = 0.5f * (g[0 * 3 + jx] - g[1 * 3 + jx] + g[2 * 3 + jx]);
    tmp[3 * 3 + jx] = g[2 * 3 + jx];
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) {  // (G g) G^T
    float *dst = U + ((int64_t)c * 16 + r * 4) * cop + o;
    dst[0 * (int64_t)cop] = tmp[r * 3 + 0];
    dst[1 * (int64_t)cop] = 0.5f * (tmp[r * 3 + 0] + tmp[r * 3 + 1] + tmp[r * 3 + 2]);
    dst[2 * (int64_t)cop] = 0.5f * (tmp[r * 3 + 0] - tmp[r * 3 + 1] + tmp[r * 3 + 2]);
    dst[3 * (int64_t)cop] = tmp[r * 3 + 2];
  }
}

// F(4x4,3x3): U = G g G^T in the MFMA operand order of winograd_f4.hip ([co block 64][channel pair][row 6][co half 2][lane x 4 | lane x 2]),
// element i = c * cop + o of [cip][cop]
__device__ __forceinline__ void pack_f4_elem(const float *__restrict__ w, float *__restrict__ U, int64_t i, int co, int ci, int cop, int cip,
                                             int transpose_flip) {
  const int o = (int)(i % cop), c = (int)(i / cop);
  float g[9];
#pragma unroll
  for (int t = 0; t < 9; ++t) g[t] = pack_src(w, co, ci, 9, o, c, t, transpose_flip);
  // rows of G: (1/4, 0, 0), (-1/6, -1/6, -1/6), (-1/6, 1/6, -1/6), (1/24, 1/12, 1/6), (1/24, -1/12, 1/6), (0, 0, 1)
  auto G6 = [](float g0, float g1, float g2, float *o6) {
    const float e = (g0 + g2) * (-1.f / 6.f), f = g1 * (-1.f / 6.f);
    const float p = g0 * (1.f / 24.f) + g2 * (1.f / 6.f), q2 = g1 * (1.f / 12.f);
    o6[0] = g0 * 0.25f;
    o6[1] = e + f;
    o6[2] = e - f;
    o6[3] = p + q2;
    o6[4] = p - q2;
    o6[5] = g2;
  };
  float tmp[6][3];  // G g
#pragma unroll
  for (int jx = 0; jx < 3; ++jx) {
    float col[6];
    G6(g[0 * 3 + jx], g[1 * 3 + jx], g[2 * 3 + jx], col);
#pragma unroll
    for (int r = 0; r < 6; ++r) tmp[r][jx] = col[r];
  }
  const int64_t blk0 = ((int64_t)(o >> 6) * (cip >> 1) + (c >> 1)) * 12 + ((o >> 5) & 1);
  const int ln = (c & 1) * 32 + (o & 31);
#pragma unroll
  for (int r = 0; r < 6; ++r) {  // (G g) G^T
    float u[6];
    G6(tmp[r][0], tmp[r][1], tmp[r][2], u);
    float *blk = U + (blk0 + 2 * r) * 384;
    *reinterpret_cast<pk_f32x4 *>(blk + ln * 4) = pk_f32x4{u[0], u[1], u[2], u[3]};
    *reinterpret_cast<pk_f32x2 *>(blk + 256 + ln * 2) = pk_f32x2{u[4], u[5]};
  }
}

// ---- split-operand F(4x4) weights (winograd_f4s.hip): U * s_U as one dword (f16 hi | f16 lo << 16) per element, behind a 16-dword
// header: [0] = s_U, [1] = 1 / s_U (floats), [2], [3] = max |w| bits gathered by the multi-tensor path (pack.hip), alternating per call
typedef _Float16 pk_f16x2 __attribute__((ext_vector_type(2)));

// x * s = hi + lo in f16 (s a power of two): v_fma_mixlo_f16 + v_fma_mixhi_f16 - the residual x s - hi is exact in the fma
__device__ __forceinline__ unsigned split_f16x2(float x, float s) {
  const float xs = x * s;
  const _Float16 hi = (_Float16)xs;
  const _Float16 lo = (_Float16)(xs - (float)hi);
  return __builtin_bit_cast(unsigned, pk_f16x2{hi, lo});
}

// biased exponent field of s_U = 2^e with max|w| s_U in [2^14, 2^15) (|G g G^T| <= max|w|): m = f 2^k, f in [1, 2) -> e = 14 - k
__device__ __forceinline__ unsigned f4s_weight_scale_field(unsigned amax_bits) {
  const int be = (int)((amax_bits >> 23) & 255u);
  return (unsigned)min(max(127 + 14 - (be - 127), 7), 200);
}

// element i = c * cop + o of [cip][cop]: U = G g G^T of the 3x3 kernel, scaled, split, in the operand order above
__device__ __forceinline__ void pack_f4s_elem(const float *__restrict__ w, unsigned *__restrict__ U, int64_t i, int co, int ci, int cop, int cip,
                                              int transpose_flip, float s_u) {
  const int o = (int)(i % cop), c = (int)(i / cop);
  float g[9];
#pragma unroll
  for (int t = 0; t < 9; ++t) g[t] = pack_src(w, co, ci, 9, o, c, t, transpose_flip);
  auto G6 = [](float g0, float g1, float g2, float *o6) {  // rows of G: (1/4,0,0), (-1/6,-1/6,-1/6), (-1/6,1/6,-1/6), (1/24,1/12,1/6), (1/24,-1/12,1/6), (0,0,1)
    const float e = (g0 + g2) * (-1.f / 6.f), f = g1 * (-1.f / 6.f);
    const float p = g0 * (1.f / 24.f) + g2 * (1.f / 6.f), q2 = g1 * (1.f / 12.f);
    o6[0] = g0 * 0.25f;
    o6[1] = e + f;
    o6[2] = e - f;
    o6[3] = p + q2;
    o6[4] = p - q2;
    o6[5] = g2;
  };
  float tmp[6][3];  // G g
#pragma unroll
  for (int jx = 0; jx < 3; ++jx) {
    float col[6];
    G6(g[0 * 3 + jx], g[1 * 3 + jx], g[2 * 3 + jx], col);
#pragma unroll
    for (int r = 0; r < 6; ++r) tmp[r][jx] = col[r];
  }
  const int64_t blk0 = (((int64_t)(o >> 6) * (cip >> 3) + (c >> 3)) * 6) * 2 + ((o >> 5) & 1);  // + 2 r -> (co block, chunk, row r, co half)
  const int ln = ((c >> 2) & 1) * 32 + (o & 31);
#pragma unroll
  for (int r = 0; r < 6; ++r) {  // (G g) G^T
    float u[6];
    G6(tmp[r][0], tmp[r][1], tmp[r][2], u);
    unsigned *blk = U + 16 + (blk0 + 2 * r) * (6 * 256) + ln * 4 + (c & 3);
#pragma unroll
    for (int cc = 0; cc < 6; ++cc) blk[cc * 256] = split_f16x2(u[cc], s_u);
  }
}


}  // namespace edvr
