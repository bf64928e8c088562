// dcn_fused.hip - fused modulated deformable convolution forward for the EDVR signature (gfx950).
//
// Same arithmetic as modulated_deformable_im2col_gpu_kernel + addmm_ of the reference
// (basicsr/models/ops/dcn/src/deform_conv_cuda_kernel.cu:570-633, deform_conv_cuda.cpp:550-555) for
// kernel 3x3, stride 1, pad 1, dilation 1, groups 1 - the only configuration EDVR uses
// (edvr_arch.py:47-52,61-66) - but WITHOUT the column buffer: the bilinear sample is computed in
// registers and fed to the matrix core directly.
//
//   v_mfma_f32_32x32x2_f32:  D[co, pixel] += W[co, (c, tap)] * col[(c, tap), pixel]
//   B operand: lane l holds col[k = l>>5][pixel = l&31]; lanes 0-31 sample channel c, lanes 32-63
//   channel c+1 of the SAME 32 pixels at the SAME tap - both channels belong to one deformable group, so
//   the sampling position, the four bilinear weights (mask folded in) and the LDS address are computed
//   once per (pixel, group, tap) and kept in registers for all channels of the group.
//
// Per workgroup (256 threads, 4 waves): a 4 x 32 pixel output tile x up to 128 output channels.
// Per chunk of 8 input channels: the input halo tile (tile + 1 + R pixels on every side, zero outside the
// image so the reference's per-corner bounds test .cu:481-491 falls out of the data) and the weight slab
// (72 rows x 128) are staged in LDS through the same register-prefetch pipeline as conv2d.hip.
// A tap whose 2x2 cell leaves the staged halo (|offset| > R) takes a wave-uniform slow path that gathers
// from global memory with the full bounds logic - correctness never depends on R.
#include "common.h"

namespace edvr {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

struct DcnFusedArgs {
  const float *x, *offset, *mask, *wpk, *bias;
  float *y;
  int B, C, H, W, Co, dg, cop, act, co_start, tiles_x, tiles_y;
  int64_t off_bs, msk_bs;
};

// bilinear sample of one channel plane straight from global memory (slow path), full reference semantics
__device__ __noinline__ float dcn_sample_global(const float *plane, float h, float w, float m, int H, int W) {
  if (!(h > -1.f && w > -1.f && h < (float)H && w < (float)W)) return 0.f;
  const float fh = floorf(h), fw = floorf(w);
  const int h0 = (int)fh, w0 = (int)fw, h1 = h0 + 1, w1 = w0 + 1;
  const float lh = h - fh, lw = w - fw, hh = 1.f - lh, hw = 1.f - lw;
  float v = 0.f;
  if (h0 >= 0 && w0 >= 0) v += hh * hw * plane[h0 * W + w0];
  if (h0 >= 0 && w1 <= W - 1) v += hh * lw * plane[h0 * W + w1];
  if (h1 <= H - 1 && w0 >= 0) v += lh * hw * plane[h1 * W + w0];
  if (h1 <= H - 1 && w1 <= W - 1) v += lh * lw * plane[h1 * W + w1];
  return v * m;
}

template <int MT, int R>
__global__ __launch_bounds__(256, 2) void dcn_fused_fwd_kernel(const DcnFusedArgs a) {
  constexpr int TH = 4, TW = 32, KK = 9, CK = 8;
  constexpr int IH = TH + 2 + 2 * R, IW = TW + 2 + 2 * R;  // R = 3: 12 x 40 halo tile; R = 7: 20 x 48
  constexpr int RS = IW, CHS = IH * RS;
  constexpr int MB = 32 * MT, WROWS = CK * KK;
  constexpr int XS_ELEMS = (CK * CHS + 3) / 4 * 4;
  __shared__ __attribute__((aligned(16))) float smem[XS_ELEMS + WROWS * MB];
  float *xs = smem;
  float *wsm = smem + XS_ELEMS;

  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, half = lane >> 5, j = lane & 31;
  int tile, blk_y, img;
  xcd_block_index(tile, blk_y, img);  // neighbouring tiles share one XCD's L2 (common.h)
  const int ty0 = (tile / a.tiles_x) * TH, tx0 = (tile % a.tiles_x) * TW;
  const int co_blk = a.co_start + blk_y * MB;
  const int P = a.H * a.W, cpg = a.C / a.dg;
  const float *x = a.x + (int64_t)img * a.C * P;
  const float *off_b = a.offset + (int64_t)img * a.off_bs;
  const float *msk_b = a.mask + (int64_t)img * a.msk_bs;

  // this lane's output pixel
  const int oy = ty0 + wave, ox = tx0 + j;
  const bool pix_ok = oy < a.H && ox < a.W;
  const int p = pix_ok ? oy * a.W + ox : 0;
  const int hy0 = ty0 - 1 - R, wx0 = tx0 - 1 - R;  // image coordinates of LDS (0, 0)

  f32x16 acc[MT];
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[m][r] = 0.f;

  // ---- staging pipeline state (see conv2d.hip)
  constexpr int NXK = (CHS + 255) / 256;
  constexpr int V4_PER_ROW = MB / 4, ROWS_PER_PASS = 256 / V4_PER_ROW;
  constexpr int NW = (WROWS + ROWS_PER_PASS - 1) / ROWS_PER_PASS;
  int xoff[NXK];
#pragma unroll
  for (int k = 0; k < NXK; ++k) {
    const int q = tid + k * 256;
    const int iy = q / RS, ix = q - iy * RS;
    const int gy = hy0 + iy, gx = wx0 + ix;
    xoff[k] = (q < CHS && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W) ? gy * a.W + gx : -1;
  }
  const int wrow0 = tid / V4_PER_ROW, wc4 = tid - wrow0 * V4_PER_ROW;
  const bool w_active = wrow0 < ROWS_PER_PASS;
  const int woff0 = w_active ? wrow0 * a.cop + wc4 * 4 : 0;
  float xr[CK * NXK];
  f32x4 wr[NW];
  auto prefetch = [&](int c0) {
    const float *wsrc = a.wpk + (int64_t)c0 * KK * a.cop + co_blk + woff0;
#pragma unroll
    for (int i = 0; i < NW; ++i) {
      const bool ok = (i + 1) * ROWS_PER_PASS <= WROWS || wrow0 + i * ROWS_PER_PASS < WROWS;
      wr[i] = *reinterpret_cast<const f32x4 *>(wsrc + (ok ? i * ROWS_PER_PASS * a.cop : 0));
    }
#pragma unroll
    for (int ch = 0; ch < CK; ++ch) {
      const float *src = x + (int64_t)(c0 + ch) * P;
#pragma unroll
      for (int k = 0; k < NXK; ++k) {
        const bool ok = xoff[k] >= 0;
        const float v = src[ok ? xoff[k] : 0];
        xr[ch * NXK + k] = ok ? v : 0.f;
      }
    }
  };
  auto commit = [&]() {
#pragma unroll
    for (int i = 0; i < NW; ++i) {
      const int row = wrow0 + i * ROWS_PER_PASS;
      if (w_active && ((i + 1) * ROWS_PER_PASS <= WROWS || row < WROWS)) *reinterpret_cast<f32x4 *>(wsm + row * MB + wc4 * 4) = wr[i];
    }
#pragma unroll
    for (int ch = 0; ch < CK; ++ch)
#pragma unroll
      for (int k = 0; k < NXK; ++k)
        if ((k + 1) * 256 <= CHS || tid + k * 256 < CHS) xs[ch * CHS + tid + k * 256] = xr[ch * NXK + k];
  };

  // ---- per-(pixel, group, tap) sampling state, refreshed when the chunk enters a new deformable group
  float w00[KK], w01[KK], w10[KK], w11[KK];
  float hs[KK], wsx[KK];  // sampling position (slow path only)
  int taddr[KK];
  unsigned slow = 0;  // bit t: the 2x2 cell of tap t is not fully inside the staged halo
  auto load_taps = [&](int g) {
#pragma unroll
    for (int t = 0; t < KK; ++t) {
      const float dy = off_b[(int64_t)(g * 18 + 2 * t) * P + p];
      const float dx = off_b[(int64_t)(g * 18 + 2 * t + 1) * P + p];
      const float m = msk_b[(int64_t)(g * 9 + t) * P + p];
      const float h = (float)(oy - 1 + t / 3) + dy, w = (float)(ox - 1 + t % 3) + dx;
      const bool valid = pix_ok && h > -1.f && w > -1.f && h < (float)a.H && w < (float)a.W;  // .cu:618
      const float fh = floorf(h), fw = floorf(w);
      const float lh = h - fh, lw = w - fw;
      const float mm = valid ? m : 0.f;
      hs[t] = h;
      wsx[t] = w;
      w00[t] = (1.f - lh) * (1.f - lw) * mm;
      w01[t] = (1.f - lh) * lw * mm;
      w10[t] = lh * (1.f - lw) * mm;
      w11[t] = lh * lw * mm;
      const int ry = (int)fh - hy0, rx = (int)fw - wx0;
      const bool inside = ry >= 0 && ry <= IH - 2 && rx >= 0 && rx <= IW - 2;
      taddr[t] = half * CHS + (inside ? ry * RS + rx : 0);
      if (valid && !inside) slow |= 1u << t; else slow &= ~(1u << t);
    }
  };

  const int abase = half * KK * MB + j;
  prefetch(0);
  commit();
  load_taps(0);
  __syncthreads();
  for (int c0 = 0; c0 < a.C; c0 += CK) {
    const bool more = (c0 + CK) < a.C;
    if (more) prefetch(c0 + CK);
#pragma unroll 1
    for (int cp = 0; cp < CK / 2; ++cp) {
#pragma unroll
      for (int t = 0; t < KK; ++t) {
        float av[MT];
#pragma unroll
        for (int m = 0; m < MT; ++m) av[m] = wsm[abase + (2 * cp * KK + t) * MB + m * 32];
        const float *cell = xs + taddr[t] + 2 * cp * CHS;
        float bv = w00[t] * cell[0] + w01[t] * cell[1] + w10[t] * cell[RS] + w11[t] * cell[RS + 1];
        if (__any(slow >> t & 1u)) {  // wave-uniform: some lane's cell left the halo -> gather from global
          if (slow >> t & 1u) {
            const float m = msk_b[(int64_t)((c0 / cpg) * 9 + t) * P + p];
            bv = dcn_sample_global(x + (int64_t)(c0 + 2 * cp + half) * P, hs[t], wsx[t], m, a.H, a.W);
          }
        }
#pragma unroll
        for (int m = 0; m < MT; ++m) acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[m], bv, acc[m], 0, 0, 0);
      }
    }
    if (more) {
      __syncthreads();
      commit();
      if ((c0 + CK) % cpg == 0) load_taps((c0 + CK) / cpg);
      __syncthreads();
    }
  }

  // ---- epilogue: bias, activation, store
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int co = co_blk + m * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
      float v = acc[m][r];
      if (a.bias) v += a.bias[co < a.Co ? co : a.Co - 1];
      if (a.act == EDVR_ACT_LRELU) v = v > 0.f ? v : 0.1f * v;
      else if (a.act == EDVR_ACT_RELU) v = fmaxf(v, 0.f);
      else if (a.act == EDVR_ACT_SIGMOID) v = __builtin_amdgcn_rcpf(1.f + __expf(-v));
      if (pix_ok && co < a.Co) a.y[((int64_t)img * a.Co + co) * P + p] = v;
    }
}

bool dcn_fused_supported(int C, int Co, int kh, int kw, int stride, int pad, int dil, int groups, int dg) {
  return kh == 3 && kw == 3 && stride == 1 && pad == 1 && dil == 1 && groups == 1 && C % dg == 0 && (C / dg) % 8 == 0 && Co > 0;
}

template <int MT>
static int launch_fused(DcnFusedArgs a, int co_start, int co_blocks, int halo, hipStream_t stream) {
  a.co_start = co_start;
  dim3 grid(a.tiles_x * a.tiles_y, co_blocks, a.B);
  if (halo > 3) hipLaunchKernelGGL((dcn_fused_fwd_kernel<MT, 7>), grid, dim3(256), 0, stream, a);
  else hipLaunchKernelGGL((dcn_fused_fwd_kernel<MT, 3>), grid, dim3(256), 0, stream, a);
  return check_launch("dcn_fused_fwd_kernel");
}

int dcn_fused_forward(const float *x, const float *offset, const float *mask, const float *wpk, const float *bias, float *y, int B, int C,
                      int H, int W, int Co, int dg, int64_t off_bs, int64_t msk_bs, int act, int halo, hipStream_t stream) {
  EDVR_REQUIRE(B <= 65535, "dcn_fused: batch %d exceeds grid.z", B);
  DcnFusedArgs a;
  a.x = x; a.offset = offset; a.mask = mask; a.wpk = wpk; a.bias = bias; a.y = y;
  a.B = B; a.C = C; a.H = H; a.W = W; a.Co = Co; a.dg = dg; a.act = act;
  a.cop = (Co + 31) / 32 * 32;
  a.off_bs = off_bs; a.msk_bs = msk_bs;
  a.tiles_x = cdiv(W, 32);
  a.tiles_y = cdiv(H, 4);
  a.co_start = 0;
  const int full = Co / 128, rem_tiles = cdiv(Co - full * 128, 32);
  int rc = EDVR_OK;
  if (full > 0) rc = launch_fused<4>(a, 0, full, halo, stream);
  if (rc || rem_tiles == 0) return rc;
  switch (rem_tiles) {
    case 1: return launch_fused<1>(a, full * 128, 1, halo, stream);
    case 2: return launch_fused<2>(a, full * 128, 1, halo, stream);
    case 3: return launch_fused<3>(a, full * 128, 1, halo, stream);
    default: return launch_fused<4>(a, full * 128, 1, halo, stream);
  }
}

}  // namespace edvr
