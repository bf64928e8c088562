// conv_small.hip - 3x3 / stride-1 convolution with at most 4 output channels on the vector ALUs (gfx950).
//
// EDVR's last layer (conv_last, 64 -> 3 channels at the full 720x1280 output resolution, edvr_arch.py:353,412) wastes a matrix
// core: the smallest MFMA tile is 32 output channels wide, so the direct kernel spends 29/32 of its MFMAs on padding (1.19 ms,
// 10 TF/s of useful work).  1728 multiply-adds per pixel are nothing for the VALUs: each thread owns two horizontally adjacent
// pixels x all (<= 4) output channels, the input halo tile of 8 channels sits in LDS (every LDS value feeds 3 taps x 2 pixels
// x co FMAs), the weights of the chunk sit in LDS too (broadcast reads).  HBM-bound: reads 4 Ci B, writes 4 Co B per pixel.
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
  // the chunk's weights, one 16-byte record per (channel, tap): read back as LDS broadcasts.  (Through the scalar cache every
  // s_load shares lgkmcnt with the LDS reads and is waited with lgkmcnt(0): PMC showed the waves 78 % of the time in s_waitcnt.)
  __shared__ f32x4 wl[CKS * 9];
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
  // the halo positions this thread stages (the same for every channel): 660 positions / 256 threads = up to 3, resolved once
  // per tile - with the index arithmetic inside the channel loop the staging cost twice the FMAs
  constexpr int NPOS = (IH * IW + 255) / 256;
  int g_off[NPOS], l_off[NPOS];  // element offset inside a channel plane (-1: zero padding / no position), offset inside an LDS plane
#pragma unroll
  for (int k = 0; k < NPOS; ++k) {
    const int p = tid + k * 256, r = p / IW, col = p - r * IW;
    const int gy = ty0 - 1 + r, gx = tx0 - 1 + col;
    l_off[k] = p < IH * IW ? r * RS + col : -1;
    g_off[k] = (p < IH * IW && gy >= 0 && gy < d.h && gx >= 0 && gx < d.w) ? gy * d.w + gx : -1;
  }

  // The halo tile of 8 channels (zero outside the image and past the last channel; lanes along x) and the chunk's weights are
  // fetched into registers one chunk AHEAD: the loads of chunk k + 1 are in flight during the FMAs of chunk k (round 4; fetched
  // in front of the FMAs, every chunk began with a full memory round trip that only other workgroups could cover: 0.25 of the HBM rate)
  float pre[CKS][NPOS];
  f32x4 wpre = f32x4{0.f, 0.f, 0.f, 0.f};
  auto fetch = [&](int c0) {
#pragma unroll
    for (int ch = 0; ch < CKS; ++ch) {
      const int c = c0 + ch;
      const float *plane = c < d.c1 ? x1 + (int64_t)c * hw : x2 + (int64_t)(c - d.c1) * hw;  // wave-uniform
#pragma unroll
      for (int k = 0; k < NPOS; ++k) pre[ch][k] = (c < a.ci && g_off[k] >= 0) ? plane[g_off[k]] : 0.f;
    }
    if (tid < CKS * 9)  // packed direct layout [ci_pad16][9][cop]: channels past ci are zero rows there
      wpre = *reinterpret_cast<const f32x4 *>(d.wpk + ((int64_t)(c0 + tid / 9) * 9 + tid % 9) * a.cop);
  };
  fetch(0);
  for (int c0 = 0; c0 < a.ci; c0 += CKS) {
#pragma unroll
    for (int ch = 0; ch < CKS; ++ch)
#pragma unroll
      for (int k = 0; k < NPOS; ++k)
        if (l_off[k] >= 0) xs[ch * IH * RS + l_off[k]] = pre[ch][k];
    if (tid < CKS * 9) wl[tid] = wpre;
    __syncthreads();
    if (c0 + CKS < a.ci) fetch(c0 + CKS);
#pragma unroll
    for (int ch = 0; ch < CKS; ++ch) {
#pragma unroll
      for (int r = 0; r < 3; ++r) {
        const float *row = xs + (ch * IH + ty + r) * RS + 2 * tx;
        const f32x2 v01 = *reinterpret_cast<const f32x2 *>(row), v23 = *reinterpret_cast<const f32x2 *>(row + 2);
        const float v[4] = {v01[0], v01[1], v23[0], v23[1]};
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
          const f32x4 w = wl[ch * 9 + r * 3 + kx];
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


// ------------------------------------------------------------------------------------------------------------------------
// Weight gradient of the same layers (co <= 4): dW[co][ci][tap] = sum_{n, pixel} dz[co][pixel] * x[ci][pixel + tap].  The direct
// MFMA kernel pads co to 32 (2.6 ms for conv_last at 32 x 64 x 256 x 256, 1.4 % of the training step).  Here a thread owns one input
// channel and one row of a 4 x 32 pixel tile and keeps its co x 9 partial sums in registers while its workgroup walks a range of
// tiles (split-K over tiles, partials reduced by reduce_partials_launch); x tile + halo in LDS with an odd channel stride, the
// dz values are LDS broadcasts.
struct SmallCoWgradArgs {
  const float *x, *dz;
  float *ws;  // [splits][co][ci][9]
  int ci, co, n, h, w;
  int64_t x_img_stride, dz_img_stride;
  int tiles_x, tiles_y, tiles, splits;
};

__global__ __launch_bounds__(256, 2) void wgrad3x3_smallco_kernel(const SmallCoWgradArgs a) {
  constexpr int TH = 4, TW = 32, IH = TH + 2, IW = TW + 2, CHS = IH * IW + 1;  // 205: odd channel stride, lanes = channels
  __shared__ float xs[64 * CHS];
  __shared__ float zs[4 * TH * TW];
  const int tid = threadIdx.x, cl = tid & 63, q = tid >> 6;  // channel of the 64-block, tile row
  const int ci0 = blockIdx.y * 64, split = blockIdx.x;
  const int t_begin = (int)((int64_t)a.tiles * split / a.splits), t_end = (int)((int64_t)a.tiles * (split + 1) / a.splits);
  const int hw = a.h * a.w;
  float acc[4][9];
#pragma unroll
  for (int o = 0; o < 4; ++o)
#pragma unroll
    for (int t = 0; t < 9; ++t) acc[o][t] = 0.f;

  // Thread p < 204 owns halo position p of ALL 64 channels: the position (and its validity) is resolved once per tile, the channel
  // loop is 64 independent loads at a constant stride (rows coalesced along p).  The loads of tile k + 1 are issued into registers
  // BEFORE the arithmetic of tile k (round 4: staged in front of the arithmetic, eight at a time, their latency - eight round trips
  // per tile with three workgroups per CU - was the kernel: 0.06 of the HBM rate; an earlier flat (channel, position) walk with two
  // integer divisions per element the same).
  float pre[64], zpre[2];
  auto fetch = [&](int tile) {
    const int img = tile / (a.tiles_x * a.tiles_y), tr = tile - img * (a.tiles_x * a.tiles_y);
    const int ty0 = (tr / a.tiles_x) * TH, tx0 = (tr % a.tiles_x) * TW;
    const float *xi = a.x + (int64_t)img * a.x_img_stride;
    const float *zi = a.dz + (int64_t)img * a.dz_img_stride;
    {
      // buffer loads: one wave-uniform resource over this block's channel planes of the image + ONE 32-bit lane offset, the channel
      // in the scalar offset (64-bit pointers per load: 128 address registers, one wave per SIMD); positions outside the image and
      // the lanes beyond the halo carry the out-of-range offset, channels past the last one lie beyond num_records: all read 0
      const int r = tid / IW, col = tid - r * IW;
      const int gy = ty0 - 1 + r, gx = tx0 - 1 + col;
      const bool inside = tid < IH * IW && gy >= 0 && gy < a.h && gx >= 0 && gx < a.w;
      const int nch = min(64, a.ci - ci0);
      const uint64_t pv = reinterpret_cast<uint64_t>(xi + (int64_t)ci0 * hw);
      const uint64_t pu = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(pv >> 32)) << 32) |
                          (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)pv);
      const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void *>(pu), (short)0, nch * hw * 4, 0x00020000);
      const int vo = inside ? (gy * a.w + gx) * 4 : (int)0x80000000;
#pragma unroll
      for (int ch = 0; ch < 64; ++ch) pre[ch] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, vo, ch * hw * 4, 0));
    }
#pragma unroll
    for (int k = 0; k < 2; ++k) {  // 4 x TH x TW = 512 dz values: two per thread
      const int i = tid + 256 * k;
      const int o = i / (TH * TW), rem = i - o * (TH * TW), r = rem / TW, col = rem - r * TW;
      const int gy = ty0 + r, gx = tx0 + col;
      zpre[k] = (o < a.co && gy < a.h && gx < a.w) ? zi[(int64_t)o * hw + gy * a.w + gx] : 0.f;
    }
  };
  if (t_begin < t_end) fetch(t_begin);
  for (int tile = t_begin; tile < t_end; ++tile) {
    __syncthreads();  // previous tile fully consumed
    if (tid < IH * IW) {
#pragma unroll
      for (int ch = 0; ch < 64; ++ch) xs[tid + ch * CHS] = pre[ch];
    }
    zs[tid] = zpre[0];
    zs[tid + 256] = zpre[1];
    __syncthreads();
    if (tile + 1 < t_end) fetch(tile + 1);  // in flight during the arithmetic below
    // this thread: channel cl, output row q of the tile; slide along x with a 3-column window of its three input rows
    const float *xr = xs + cl * CHS + q * IW;
    float w0[3], w1[3], w2[3];  // columns px-1, px, px+1 of rows q, q+1, q+2 (halo coordinates)
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      w0[r] = xr[r * IW + 0];
      w1[r] = xr[r * IW + 1];
    }
#pragma unroll 4
    for (int px = 0; px < TW; ++px) {
#pragma unroll
      for (int r = 0; r < 3; ++r) w2[r] = xr[r * IW + px + 2];
#pragma unroll
      for (int o = 0; o < 4; ++o) {
        const float z = zs[(o * TH + q) * TW + px];  // wave-uniform address: LDS broadcast
#pragma unroll
        for (int r = 0; r < 3; ++r) {
          acc[o][r * 3 + 0] += z * w0[r];
          acc[o][r * 3 + 1] += z * w1[r];
          acc[o][r * 3 + 2] += z * w2[r];
        }
      }
#pragma unroll
      for (int r = 0; r < 3; ++r) {
        w0[r] = w1[r];
        w1[r] = w2[r];
      }
    }
  }
  // reduce the four tile rows (one wave each) through LDS, then one partial per (split, co, ci, tap)
  __syncthreads();
  float *red = xs;  // 4 x 64 x 36 floats
#pragma unroll
  for (int o = 0; o < 4; ++o)
#pragma unroll
    for (int t = 0; t < 9; ++t) red[(q * 64 + cl) * 37 + o * 9 + t] = acc[o][t];
  __syncthreads();
  const int ci_total = a.ci;
  for (int i = tid; i < 64 * 36; i += 256) {
    const int ch = i / 36, k = i - ch * 36, o = k / 9, t = k - o * 9;
    const int c = ci0 + ch;
    if (c < ci_total && o < a.co) {
      const float s = (red[(0 * 64 + ch) * 37 + k] + red[(1 * 64 + ch) * 37 + k]) + (red[(2 * 64 + ch) * 37 + k] + red[(3 * 64 + ch) * 37 + k]);
      a.ws[((int64_t)split * a.co + o) * ci_total * 9 + (int64_t)c * 9 + t] = s;
    }
  }
}

bool wgrad_small_plan(int n, int c1, int c2, int h, int w, int co, int ks, int stride, int *splits) {
  if (ks != 3 || stride != 1 || co > 4 || c2 != 0) return false;
  if ((int64_t)64 * h * w * 4 >= ((int64_t)1 << 31)) return false;  // 32-bit buffer offsets over a block of 64 channel planes
  const int tiles = n * cdiv(h, 4) * cdiv(w, 32);
  *splits = std::max(1, std::min(tiles, 512 / cdiv(c1, 64)));
  return true;
}

size_t wgrad_small_ws_bytes(int co, int ci, int splits) { return (size_t)splits * co * ci * 9 * sizeof(float); }

int wgrad_small_launch(const float *x, const float *dz, float *ws, int ci, int co, int n, int h, int w, int64_t x_img_stride,
                       int64_t dz_img_stride, int splits, hipStream_t stream) {
  SmallCoWgradArgs a;
  a.x = x; a.dz = dz; a.ws = ws;
  a.ci = ci; a.co = co; a.n = n; a.h = h; a.w = w;
  a.x_img_stride = x_img_stride; a.dz_img_stride = dz_img_stride;
  a.tiles_x = cdiv(w, 32);
  a.tiles_y = cdiv(h, 4);
  a.tiles = n * a.tiles_x * a.tiles_y;
  a.splits = splits;
  hipLaunchKernelGGL(wgrad3x3_smallco_kernel, dim3(splits, cdiv(ci, 64)), dim3(256), 0, stream, a);
  return check_launch("wgrad3x3_smallco_kernel");
}

}  // namespace edvr
