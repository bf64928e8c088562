// conv_small.hip - 3x3 / stride-1 convolution with at most 4 output channels on the vector ALUs (gfx950).
//
// EDVR's last layer (conv_last, 64 -> 3 channels at the full 720x1280 output resolution, edvr_arch.py:353,412) wastes a matrix
// core: the smallest MFMA tile is 32 output channels wide, so the direct kernel spends 29/32 of its MFMAs on padding (1.19 ms,
// 10 TF/s of useful work).  1728 multiply-adds per pixel are nothing for the VALUs: each thread owns two horizontally adjacent
// pixels x all (<= 4) output channels, the input halo tile of 8 channels sits in LDS (every LDS value feeds 3 taps x 2 pixels
// x co FMAs), the weights are wave-uniform and come through the scalar cache.  HBM-bound: reads 4 Ci B, writes 4 Co B per pixel.
#include "common.h"

namespace edvr {

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

struct SmallCoArgs {
  edvr_conv2d_desc d;
  int ci, cop, tiles_x;
};

__global__ __launch_bounds__(256) void conv3x3_smallco_kernel(const SmallCoArgs a) {
  constexpr int TH = 8, TW = 64, CKS = 8, IH = TH + 2, IW = TW + 2, RS = 68;  // RS even: the 64-bit LDS reads stay aligned
  __shared__ __attribute__((aligned(16))) float xs[CKS * IH * RS];
  const edvr_conv2d_desc &d = a.d;
  const int tid = threadIdx.x, ty = tid >> 5, tx = tid & 31;
  int tile, unused, img;
  xcd_block_index(tile, unused, img);
  const int ty0 = (tile / a.tiles_x) * TH, tx0 = (tile % a.tiles_x) * TW;
  const int hw = d.h * d.w;
  const float *x1 = d.x1 + (int64_t)img * d.x1_img_stride;
  const float *x2 = nullptr;
  if (d.x2) {
    const int i2 = d.x2_div > 0 ? (img / d.x2_div) * d.x2_mul + d.x2_add : img;
    x2 = d.x2 + (int64_t)i2 * d.x2_img_stride;
  }
  f32x2 acc[4];
#pragma unroll
  for (int o = 0; o < 4; ++o) acc[o] = f32x2{0.f, 0.f};

  for (int c0 = 0; c0 < a.ci; c0 += CKS) {
    // stage the halo tile of 8 channels (zero outside the image and past the last channel): lanes along x
    for (int i = tid; i < CKS * IH * IW; i += 256) {
      const int ch = i / (IH * IW), rem = i - ch * (IH * IW), r = rem / IW, col = rem - r * IW;
      const int c = c0 + ch, gy = ty0 - 1 + r, gx = tx0 - 1 + col;
      float v = 0.f;
      if (c < a.ci && gy >= 0 && gy < d.h && gx >= 0 && gx < d.w)
        v = (c < d.c1 ? x1 + (int64_t)c * hw : x2 + (int64_t)(c - d.c1) * hw)[gy * d.w + gx];
      xs[(ch * IH + r) * RS + col] = v;
    }
    __syncthreads();
#pragma unroll
    for (int ch = 0; ch < CKS; ++ch) {
      const float *wrow = d.wpk + (int64_t)(c0 + ch) * 9 * a.cop;  // packed direct layout [ci_pad16][9][cop]: wave-uniform
#pragma unroll
      for (int r = 0; r < 3; ++r) {
        const float *row = xs + (ch * IH + ty + r) * RS + 2 * tx;
        const f32x2 v01 = *reinterpret_cast<const f32x2 *>(row), v23 = *reinterpret_cast<const f32x2 *>(row + 2);
        const float v[4] = {v01[0], v01[1], v23[0], v23[1]};
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
          const f32x4 w = *reinterpret_cast<const f32x4 *>(wrow + (r * 3 + kx) * a.cop);
          const f32x2 xv = f32x2{v[kx], v[kx + 1]};
#pragma unroll
          for (int o = 0; o < 4; ++o) acc[o] += w[o] * xv;  // v_pk_fma_f32: both pixels at once
        }
      }
    }
    __syncthreads();
  }

  const int oy = ty0 + ty, ox = tx0 + 2 * tx;
  if (oy >= d.h) return;
  const float slope = d.act == EDVR_ACT_LRELU ? 0.1f : (d.act == EDVR_ACT_RELU ? 0.f : 1.f);
  float *y = d.y + (int64_t)img * d.y_img_stride;
  const float *r1 = d.res1 ? d.res1 + (int64_t)img * d.res1_img_stride : nullptr;
  const float *r2 = d.res2 ? d.res2 + (int64_t)img * d.res2_img_stride : nullptr;
#pragma unroll
  for (int o = 0; o < 4; ++o) {
    if (o >= d.co) break;
    const float b = d.bias ? d.bias[o] : 0.f;
#pragma unroll
    for (int px = 0; px < 2; ++px) {
      if (ox + px >= d.w) continue;
      float v = acc[o][px] + b;
      if (d.act == EDVR_ACT_SIGMOID) v = o >= d.act_from ? __builtin_amdgcn_rcpf(1.f + __expf(-v)) : v;
      else if (o >= d.act_from) v = fmaxf(v, slope * v);
      const int off = o * hw + oy * d.w + ox + px;
      if (r1) v += r1[off];
      if (r2) v += r2[off];
      y[off] = v;
    }
  }
}

bool conv_small_eligible(const edvr_conv2d_desc &d) {
  return d.ks == 3 && d.stride == 1 && d.co <= 4 && d.out_mode == EDVR_OUT_NCHW && d.algo == EDVR_CONV_AUTO;
}

int conv_small_launch(const edvr_conv2d_desc &d, hipStream_t stream) {
  SmallCoArgs a;
  a.d = d;
  a.ci = d.c1 + d.c2;
  a.cop = (d.co + 31) / 32 * 32;
  a.tiles_x = cdiv(d.w, 64);
  hipLaunchKernelGGL(conv3x3_smallco_kernel, dim3(a.tiles_x * cdiv(d.h, 8), 1, d.n), dim3(256), 0, stream, a);
  return check_launch("conv3x3_smallco_kernel");
}

}  // namespace edvr
