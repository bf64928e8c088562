// conv2d.hip - fp32 implicit-GEMM convolution on the CDNA4 matrix cores (gfx950).
//
// Replaces the at::conv2d/cuDNN calls under every nn.Conv2d of the reference's
// basicsr/models/archs/edvr_arch.py (:37-66,139-155,230-244,322-353) together
// with the elementwise ops that follow them (LeakyReLU(0.1) :70, ReLU + identity
// arch_util.py:92-95, torch.cat :90,95-96,101-102,113, PixelShuffle(2) :351,
// sigmoid on the mask third of conv_offset arch_util.py:245-247).
//
// Mapping to the hardware
//   GEMM view:  D[co, pixel] = sum_{ci,tap} W[co, ci, tap] * X[ci, pixel + tap]
//   v_mfma_f32_32x32x2_f32 (exact fp32, 64 FLOP/clk/SIMD):
//     A operand = weights  : lane l holds W[co = l&31][k = l>>5]
//     B operand = pixels   : lane l holds X[k = l>>5][pixel = l&31]
//     one k-step = one tap of TWO input channels (lanes 0-31: ci, lanes 32-63: ci+1)
//   A 256-thread workgroup (4 waves, one per SIMD) owns an output tile of
//   TH x TW pixels (8x32 or 16x16) for up to 128 output channels; each wave owns
//   64 pixels x all channels = MT x 2 accumulator tiles of 32x32 (AGPRs).
//   The input halo tile of CK channels is staged in LDS once per chunk and every
//   B operand is a single ds_read_b32 at base + compile-time offset; A operands
//   stream from the packed weight array ([ci][tap][co], co fastest -> each
//   half-wave reads one 128-B line) through L1/L2 straight into VGPRs.
//   Epilogue (bias, activation, residuals, pixel-shuffle) is applied on the
//   accumulators, so none of those ops costs an HBM pass.
#include "common.h"

namespace edvr {

typedef float f32x16 __attribute__((ext_vector_type(16)));

struct ConvArgs {
  edvr_conv2d_desc d;
  int ci, cop, ho, wo, tiles_x, tiles_y;
};

__device__ __forceinline__ float apply_act(float v, int act) {
  switch (act) {
    case EDVR_ACT_RELU: return v > 0.f ? v : 0.f;
    case EDVR_ACT_LRELU: return v > 0.f ? v : 0.1f * v;
    case EDVR_ACT_SIGMOID: return 1.f / (1.f + __expf(-v));
    default: return v;
  }
}

template <int KS, int STRIDE>
struct ConvGeom {
  static constexpr int CK = (STRIDE == 2) ? 8 : 16;  // input channels staged per chunk
};

template <int KS, int STRIDE, int MT, int SW>
__global__ __launch_bounds__(256) void conv2d_mfma_kernel(const ConvArgs a) {
  constexpr int SH = 32 / SW;        // rows of one 32-pixel subtile
  constexpr int NSUB = 2;            // subtiles per wave
  constexpr int TW = SW;             // output tile width
  constexpr int TH = 4 * NSUB * SH;  // output tile height (4 waves)
  constexpr int IW = (TW - 1) * STRIDE + KS;
  constexpr int IH = (TH - 1) * STRIDE + KS;
  constexpr int RS = IW;       // LDS row stride (floats)
  constexpr int CHS = IH * RS; // LDS channel stride
  constexpr int CK = ConvGeom<KS, STRIDE>::CK;
  constexpr int KK = KS * KS;
  constexpr int PAD = KS / 2;

  __shared__ float xs[CK * CHS];

  const edvr_conv2d_desc &d = a.d;
  const int tid = threadIdx.x;
  const int wave = tid >> 6, lane = tid & 63, half = lane >> 5, j = lane & 31;
  const int tile = blockIdx.x;
  const int ty0 = (tile / a.tiles_x) * TH, tx0 = (tile % a.tiles_x) * TW;
  const int co_blk = blockIdx.y * (32 * MT);
  const int img = blockIdx.z;
  int mt_count = (d.co - co_blk + 31) / 32;
  if (mt_count > MT) mt_count = MT;

  const float *x1 = d.x1 + (int64_t)img * d.x1_img_stride;
  const float *x2 = nullptr;
  if (d.x2) {
    const int i2 = d.x2_div > 0 ? (img / d.x2_div) * d.x2_mul + d.x2_add : img;
    x2 = d.x2 + (int64_t)i2 * d.x2_img_stride;
  }
  const int hw = d.h * d.w;

  // per-lane LDS read bases (floats) of the two subtiles
  int bbase[NSUB];
#pragma unroll
  for (int s = 0; s < NSUB; ++s) {
    const int r = (wave * NSUB + s) * SH + j / SW, c = j % SW;
    bbase[s] = half * CHS + (r * STRIDE) * RS + c * STRIDE;
  }

  f32x16 acc[MT][NSUB];
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int s = 0; s < NSUB; ++s)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[m][s][r] = 0.f;

  const int iy0 = ty0 * STRIDE - PAD, ix0 = tx0 * STRIDE - PAD;

  for (int c0 = 0; c0 < a.ci; c0 += CK) {
    __syncthreads();
    // ---- stage CK channels of the input halo tile (zero outside the image / past ci)
    for (int e = tid; e < CK * CHS; e += 256) {
      const int ch = e / CHS, rem = e - ch * CHS;
      const int iy = rem / RS, ix = rem - iy * RS;
      const int gy = iy0 + iy, gx = ix0 + ix, c = c0 + ch;
      float v = 0.f;
      if (c < a.ci && gy >= 0 && gy < d.h && gx >= 0 && gx < d.w) {
        const float *src = (c < d.c1) ? (x1 + (int64_t)c * hw) : (x2 + (int64_t)(c - d.c1) * hw);
        v = src[gy * d.w + gx];
      }
      xs[e] = v;
    }
    __syncthreads();
    // ---- MFMA over (channel pair, tap)
    const float *wp = d.wpk + ((int64_t)(c0 + half) * KK) * a.cop + co_blk + j;
#pragma unroll 2
    for (int cp = 0; cp < CK / 2; ++cp) {
#pragma unroll
      for (int t = 0; t < KK; ++t) {
        const int kh = t / KS, kw = t % KS;
        float av[MT], bv[NSUB];
#pragma unroll
        for (int m = 0; m < MT; ++m) av[m] = (m < mt_count) ? wp[((int64_t)(2 * cp) * KK + t) * a.cop + m * 32] : 0.f;
#pragma unroll
        for (int s = 0; s < NSUB; ++s) bv[s] = xs[bbase[s] + 2 * cp * CHS + kh * RS + kw];
#pragma unroll
        for (int m = 0; m < MT; ++m) {
          if (m < mt_count) {
#pragma unroll
            for (int s = 0; s < NSUB; ++s) acc[m][s] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[m], bv[s], acc[m][s], 0, 0, 0);
          }
        }
      }
    }
  }

  // ---- epilogue: bias, activation, residuals, store
  const int64_t plane = (int64_t)a.ho * a.wo;
  float *y = d.y + (int64_t)img * d.y_img_stride;
  const float *r1 = d.res1 ? d.res1 + (int64_t)img * d.res1_img_stride : nullptr;
  const float *r2 = d.res2 ? d.res2 + (int64_t)img * d.res2_img_stride : nullptr;
#pragma unroll
  for (int s = 0; s < NSUB; ++s) {
    const int oy = ty0 + (wave * NSUB + s) * SH + j / SW, ox = tx0 + j % SW;
    const bool pix_ok = (oy < a.ho) && (ox < a.wo);
#pragma unroll
    for (int m = 0; m < MT; ++m) {
      if (m < mt_count) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int co = co_blk + m * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
          if (pix_ok && co < d.co) {
            float v = acc[m][s][r];
            if (d.bias) v += d.bias[co];
            if (co >= d.act_from) v = apply_act(v, d.act);
            const int64_t o = (int64_t)co * plane + (int64_t)oy * a.wo + ox;
            if (r1) v += r1[o];
            if (r2) v += r2[o];
            if (d.out_mode == EDVR_OUT_PIXEL_SHUFFLE2) {
              const int oc = co >> 2, sy = (co >> 1) & 1, sx = co & 1;
              y[(int64_t)oc * plane * 4 + (int64_t)(2 * oy + sy) * (2 * a.wo) + 2 * ox + sx] = v;
            } else {
              y[o] = v;
            }
          }
        }
      }
    }
  }
}

__global__ void pack_weight_kernel(const float *__restrict__ w, float *__restrict__ wpk, int co, int ci, int kk, int cop,
                                   int cip, int transpose_flip) {
  const int64_t total = (int64_t)cip * kk * cop;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int o = (int)(i % cop);
    const int t = (int)((i / cop) % kk);
    const int c = (int)(i / ((int64_t)cop * kk));
    float v = 0.f;
    if (o < co && c < ci) v = transpose_flip ? w[((int64_t)c * co + o) * kk + (kk - 1 - t)] : w[((int64_t)o * ci + c) * kk + t];
    wpk[i] = v;
  }
}

static inline int round_up(int v, int m) { return (v + m - 1) / m * m; }

template <int KS, int STRIDE, int MT, int SW>
static int launch_one(const ConvArgs &a, hipStream_t stream) {
  constexpr int SH = 32 / SW, TH = 8 * SH, TW = SW;
  ConvArgs b = a;
  b.tiles_x = cdiv(a.wo, TW);
  b.tiles_y = cdiv(a.ho, TH);
  dim3 grid(b.tiles_x * b.tiles_y, cdiv(a.d.co, 32 * MT), a.d.n);
  hipLaunchKernelGGL((conv2d_mfma_kernel<KS, STRIDE, MT, SW>), grid, dim3(256), 0, stream, b);
  return check_launch("conv2d_mfma_kernel");
}

template <int KS, int STRIDE, int MT>
static int launch_sw(const ConvArgs &a, hipStream_t stream) {
  // pick the tile geometry (8x32 or 16x16) that wastes fewer lanes on this output size
  const int64_t w32 = (int64_t)cdiv(a.ho, 8) * 8 * cdiv(a.wo, 32) * 32;
  const int64_t w16 = (int64_t)cdiv(a.ho, 16) * 16 * cdiv(a.wo, 16) * 16;
  return (w16 < w32) ? launch_one<KS, STRIDE, MT, 16>(a, stream) : launch_one<KS, STRIDE, MT, 32>(a, stream);
}

template <int KS, int STRIDE>
static int launch_mt(const ConvArgs &a, hipStream_t stream) {
  if (a.d.co <= 32) return launch_sw<KS, STRIDE, 1>(a, stream);
  if (a.d.co <= 64) return launch_sw<KS, STRIDE, 2>(a, stream);
  return launch_sw<KS, STRIDE, 4>(a, stream);
}

int conv2d_launch(const edvr_conv2d_desc &d, hipStream_t stream) {
  EDVR_REQUIRE(d.x1 && d.wpk && d.y, "conv2d: null x1/wpk/y");
  EDVR_REQUIRE(d.n > 0 && d.h > 0 && d.w > 0 && d.c1 > 0 && d.co > 0, "conv2d: bad sizes n=%d h=%d w=%d c1=%d co=%d", d.n,
               d.h, d.w, d.c1, d.co);
  EDVR_REQUIRE((d.x2 != nullptr) == (d.c2 > 0), "conv2d: x2/c2 mismatch");
  EDVR_REQUIRE(d.n <= 65535, "conv2d: n=%d exceeds grid.z", d.n);
  if (!((d.ks == 3 && (d.stride == 1 || d.stride == 2)) || (d.ks == 1 && d.stride == 1))) {
    set_error("conv2d: unsupported ks=%d stride=%d", d.ks, d.stride);
    return EDVR_ERR_UNSUPPORTED;
  }
  if (d.out_mode == EDVR_OUT_PIXEL_SHUFFLE2) EDVR_REQUIRE(d.co % 4 == 0 && !d.res1 && !d.res2, "conv2d: pixel-shuffle needs co%%4==0 and no residual");
  ConvArgs a;
  a.d = d;
  a.ci = d.c1 + d.c2;
  a.cop = round_up(d.co, 32);
  const int pad = d.ks / 2;
  a.ho = (d.h + 2 * pad - d.ks) / d.stride + 1;
  a.wo = (d.w + 2 * pad - d.ks) / d.stride + 1;
  a.tiles_x = a.tiles_y = 0;
  if (d.ks == 3 && d.stride == 1) return launch_mt<3, 1>(a, stream);
  if (d.ks == 3 && d.stride == 2) return launch_mt<3, 2>(a, stream);
  return launch_mt<1, 1>(a, stream);
}

}  // namespace edvr

extern "C" {

size_t edvr_conv2d_packed_weight_elems(int co, int ci, int ks) {
  return (size_t)edvr::round_up(ci, 16) * ks * ks * edvr::round_up(co, 32);
}

int edvr_conv2d_pack_weight_f32(const float *w, float *wpk, int co, int ci, int ks, int transpose_flip,
                                edvr_stream_t stream) {
  EDVR_REQUIRE(w && wpk && co > 0 && ci > 0 && ks > 0, "pack_weight: bad arguments");
  const int cop = edvr::round_up(co, 32), cip = edvr::round_up(ci, 16), kk = ks * ks;
  const int64_t total = (int64_t)cip * kk * cop;
  const int blocks = (int)std::min<int64_t>(edvr::cdiv64(total, 256), 4096);
  hipLaunchKernelGGL(edvr::pack_weight_kernel, dim3(blocks), dim3(256), 0, edvr::as_stream(stream), w, wpk, co, ci, kk,
                     cop, cip, transpose_flip);
  return edvr::check_launch("pack_weight_kernel");
}

int edvr_conv2d_f32(const edvr_conv2d_desc *d, edvr_stream_t stream) {
  EDVR_REQUIRE(d != nullptr, "conv2d: null descriptor");
  return edvr::conv2d_launch(*d, edvr::as_stream(stream));
}

}  // extern "C"
