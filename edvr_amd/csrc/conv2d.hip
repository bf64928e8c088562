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
typedef float f32x4 __attribute__((ext_vector_type(4)));

struct ConvArgs {
  edvr_conv2d_desc d;
  int ci, cop, ho, wo, tiles_x, tiles_y;
  int co_start;  // first output channel of this launch (tail launches cover the last partial 128-block)
};

__device__ __forceinline__ float sigmoid_fast(float v) { return __builtin_amdgcn_rcpf(1.f + __expf(-v)); }

template <int KS>
struct ConvChunk {
  static constexpr int CK = (KS == 1) ? 32 : 8;  // input channels staged per chunk (K per chunk = CK * KS * KS)
};

#ifndef EDVR_CONV_MINWAVES
#define EDVR_CONV_MINWAVES 2
#endif

// NS = 32-pixel sub-tiles per wave.  Two halve the weight reads per MFMA, one halves the tile (twice the workgroups, half the
// accumulators: more waves in flight).  Measured (direct kernel alone, TF/s, NS = 2 -> 1): 3x3 stride 2 61 -> 111, 1x1 41-50 ->
// 71-82, 3x3 on 45x80 / 90x160 images 62 / 101 -> 92 / 124, 128 -> 512 channels 112 -> 126, 128 -> 128 at 180x320 109 -> 115;
// only the 64-channel layers (two 32-channel tiles per workgroup) lose 3 %.  So: one sub-tile from three channel tiles up.
template <int KS, int STRIDE, int MT, int SW, int NS>
__global__ __launch_bounds__(256, EDVR_CONV_MINWAVES) void conv2d_mfma_kernel(const ConvArgs a) {
  // (measured: giving the 1x1/MT=4 instantiation the 512-register budget instead of spilling is SLOWER, 27 vs 41 TF/s:
  //  that shape is latency-bound and wants the occupancy)
  constexpr int SH = 32 / SW;        // rows of one 32-pixel subtile
  constexpr int NSUB = NS;           // subtiles per wave
  constexpr int TW = SW;             // output tile width
  constexpr int TH = 4 * NSUB * SH;  // output tile height (4 waves)
  constexpr int IW = (TW - 1) * STRIDE + KS;
  constexpr int IH = (TH - 1) * STRIDE + KS;
  constexpr int RS = IW;        // LDS row stride (floats)
  constexpr int CHS = IH * RS;  // LDS channel stride
  constexpr int CK = ConvChunk<KS>::CK;
  constexpr int KK = KS * KS;
  constexpr int PAD = KS / 2;
  constexpr int MB = 32 * MT;         // output channels of this workgroup
  constexpr int WROWS = CK * KK;      // weight rows per chunk, each MB floats
  constexpr int XS_ELEMS = (CK * CHS + 3) / 4 * 4;

  __shared__ __attribute__((aligned(16))) float smem[XS_ELEMS + WROWS * MB];
  float *xs = smem;
  float *wsm = smem + XS_ELEMS;

  const edvr_conv2d_desc &d = a.d;
  const int tid = threadIdx.x;
  const int wave = tid >> 6, lane = tid & 63, half = lane >> 5, j = lane & 31;
  int tile = blockIdx.x, blk_y = blockIdx.y, img = blockIdx.z;
  if constexpr (KS > 1) xcd_block_index(tile, blk_y, img);  // neighbouring tiles share one XCD's L2 (common.h); 1x1 has no halo
  const int ty0 = (tile / a.tiles_x) * TH, tx0 = (tile % a.tiles_x) * TW;
  const int co_blk = a.co_start + blk_y * MB;

  const float *x1 = d.x1 + (int64_t)img * d.x1_img_stride;
  const float *x2 = nullptr;
  if (d.x2) {
    const int i2 = d.x2_div > 0 ? (img / d.x2_div) * d.x2_mul + d.x2_add : img;
    x2 = d.x2 + (int64_t)i2 * d.x2_img_stride;
  }
  const int hw = d.h * d.w;

  // per-lane LDS read bases (floats): B operand (pixels) of the two subtiles, A operand (weights)
  int bbase[NSUB];
#pragma unroll
  for (int s = 0; s < NSUB; ++s) {
    const int r = (wave * NSUB + s) * SH + j / SW, c = j % SW;
    bbase[s] = half * CHS + (r * STRIDE) * RS + c * STRIDE;
  }
  const int abase = half * KK * MB + j;

  f32x16 acc[MT][NSUB];
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int s = 0; s < NSUB; ++s)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[m][s][r] = 0.f;

  const int iy0 = ty0 * STRIDE - PAD, ix0 = tx0 * STRIDE - PAD;

  // ---- software pipeline: the global loads of chunk c+1 are issued into registers BEFORE the MFMA
  //      block of chunk c and land in LDS after it, so HBM/L2 latency hides under ~18k MFMA cycles.
  // Every load is unconditional (clamped address, value selected afterwards): a predicated load would
  // put each one in its own basic block and serialise the stream.
  constexpr int NXK = (CHS + 255) / 256;          // plane positions per thread (channel index is compile-time)
  constexpr int V4_PER_ROW = MB / 4;              // float4 per weight row
  constexpr int ROWS_PER_PASS = 256 / V4_PER_ROW; // weight rows covered by the 256 threads in one pass
  constexpr int NW = (WROWS + ROWS_PER_PASS - 1) / ROWS_PER_PASS;
  int xoff[NXK];  // offset inside one channel plane, -1 where the halo leaves the image
#pragma unroll
  for (int k = 0; k < NXK; ++k) {
    const int p = tid + k * 256;
    const int iy = p / RS, ix = p - iy * RS;
    const int gy = iy0 + iy, gx = ix0 + ix;
    xoff[k] = (p < CHS && gy >= 0 && gy < d.h && gx >= 0 && gx < d.w) ? gy * d.w + gx : -1;
  }
  const int wrow0 = tid / V4_PER_ROW, wc4 = tid - wrow0 * V4_PER_ROW;
  const bool w_active = wrow0 < ROWS_PER_PASS;
  const int woff0 = w_active ? wrow0 * a.cop + wc4 * 4 : 0;
  float xr[CK * NXK];
  f32x4 wr[NW];
  auto prefetch = [&](int c0) {
    const float *wsrc = d.wpk + (int64_t)c0 * KK * a.cop + co_blk + woff0;
#pragma unroll
    for (int i = 0; i < NW; ++i) {
      const bool ok = (i + 1) * ROWS_PER_PASS <= WROWS || wrow0 + i * ROWS_PER_PASS < WROWS;
      wr[i] = *reinterpret_cast<const f32x4 *>(wsrc + (ok ? i * ROWS_PER_PASS * a.cop : 0));
    }
#pragma unroll
    for (int ch = 0; ch < CK; ++ch) {
      const int c = c0 + ch;
      const bool cok = c < a.ci;
      const int cc = cok ? c : 0;
      const float *src = (cc < d.c1) ? (x1 + (int64_t)cc * hw) : (x2 + (int64_t)(cc - d.c1) * hw);
#pragma unroll
      for (int k = 0; k < NXK; ++k) {
        const bool ok = cok && xoff[k] >= 0;
        const float v = src[ok ? xoff[k] : 0];
        xr[ch * NXK + k] = ok ? v : 0.f;
      }
    }
  };
  auto commit = [&]() {
#pragma unroll
    for (int i = 0; i < NW; ++i) {
      const int row = wrow0 + i * ROWS_PER_PASS;
      if (w_active && ((i + 1) * ROWS_PER_PASS <= WROWS || row < WROWS))
        *reinterpret_cast<f32x4 *>(wsm + row * MB + wc4 * 4) = wr[i];
    }
#pragma unroll
    for (int ch = 0; ch < CK; ++ch)
#pragma unroll
      for (int k = 0; k < NXK; ++k)
        if ((k + 1) * 256 <= CHS || tid + k * 256 < CHS) xs[ch * CHS + tid + k * 256] = xr[ch * NXK + k];
  };

  prefetch(0);
  commit();
  __syncthreads();
  for (int c0 = 0; c0 < a.ci; c0 += CK) {
    const bool more = (c0 + CK) < a.ci;
    if (more) prefetch(c0 + CK);
    // ---- MFMA over (channel pair, tap): branch-free, every operand one ds_read_b32 at base + immediate
#pragma unroll 1
    for (int cp = 0; cp < CK / 2; ++cp) {
#pragma unroll
      for (int t = 0; t < KK; ++t) {
        const int kh = t / KS, kw = t % KS;
        float av[MT], bv[NSUB];
#pragma unroll
        for (int m = 0; m < MT; ++m) av[m] = wsm[abase + (2 * cp * KK + t) * MB + m * 32];
#pragma unroll
        for (int s = 0; s < NSUB; ++s) bv[s] = xs[bbase[s] + 2 * cp * CHS + kh * RS + kw];
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
          for (int s = 0; s < NSUB; ++s) acc[m][s] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[m], bv[s], acc[m][s], 0, 0, 0);
      }
    }
    if (more) {
      __syncthreads();  // every wave is done reading this chunk
      commit();
      __syncthreads();
    }
  }

  // ---- epilogue on the accumulators: bias, activation (uniform switch hoisted), residuals, store
  if (d.bias) {
    // unconditional clamped loads (a predicated load would serialise: one basic block + wait per element)
    float bvals[MT * 16];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int co = co_blk + m * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
        bvals[m * 16 + r] = d.bias[co < d.co ? co : d.co - 1];
      }
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int r = 0; r < 16; ++r)
#pragma unroll
        for (int s = 0; s < NSUB; ++s) acc[m][s][r] += bvals[m * 16 + r];
  }
  if (d.act != EDVR_ACT_NONE) {
    const int rel_from = d.act_from - co_blk - 4 * half;  // activation applies where (m*32 + row(r)) >= rel_from
#define EDVR_ACT_LOOP(EXPR)                                             \
  _Pragma("unroll") for (int m = 0; m < MT; ++m)                        \
  _Pragma("unroll") for (int r = 0; r < 16; ++r) {                      \
    if (m * 32 + (r & 3) + 8 * (r >> 2) >= rel_from) {                  \
      _Pragma("unroll") for (int s = 0; s < NSUB; ++s) {                \
        const float v = acc[m][s][r];                                   \
        acc[m][s][r] = (EXPR);                                          \
      }                                                                 \
    }                                                                   \
  }
    if (d.act == EDVR_ACT_LRELU) {
      EDVR_ACT_LOOP(v > 0.f ? v : 0.1f * v)
    } else if (d.act == EDVR_ACT_RELU) {
      EDVR_ACT_LOOP(fmaxf(v, 0.f))
    } else {
      EDVR_ACT_LOOP(sigmoid_fast(v))
    }
#undef EDVR_ACT_LOOP
  }
  // per-image offsets fit 32 bits (co * ho * wo < 2^31); uniform conditions are hoisted out of the store loops
  const int plane = a.ho * a.wo;
  float *y = d.y + (int64_t)img * d.y_img_stride;
  const float *r1 = d.res1 ? d.res1 + (int64_t)img * d.res1_img_stride : nullptr;
  const float *r2 = d.res2 ? d.res2 + (int64_t)img * d.res2_img_stride : nullptr;
  const int co_lane = co_blk + 4 * half;
#define EDVR_STORE_LOOP(BODY)                                                      \
  _Pragma("unroll") for (int s = 0; s < NSUB; ++s) {                               \
    const int oy = ty0 + (wave * NSUB + s) * SH + j / SW, ox = tx0 + j % SW;       \
    if (oy < a.ho && ox < a.wo) {                                                  \
      const int pix = oy * a.wo + ox;                                              \
      _Pragma("unroll") for (int m = 0; m < MT; ++m) {                             \
        _Pragma("unroll") for (int r = 0; r < 16; ++r) {                           \
          const int co = co_lane + m * 32 + (r & 3) + 8 * (r >> 2);                \
          if (co < d.co) {                                                         \
            float v = acc[m][s][r];                                                \
            const int o = co * plane + pix;                                        \
            BODY                                                                   \
          }                                                                        \
        }                                                                          \
      }                                                                            \
    }                                                                              \
  }
  const float ys = d.y_scale == 0.f ? 1.f : d.y_scale;
  if (ys != 1.f || d.gate) {  // generic variant: scale, gate (activation backward of a data-gradient conv), residuals; NCHW only
    const float *gt = d.gate ? d.gate + (int64_t)img * d.gate_img_stride : nullptr;
    EDVR_STORE_LOOP({
      v *= ys;
      if (gt) v = gt[o] > 0.f ? v : d.gate_slope * v;
      if (r1) v += r1[o];
      if (r2) v += r2[o];
      y[o] = v;
    })
  } else if (d.out_mode == EDVR_OUT_PIXEL_SHUFFLE2) {
    EDVR_STORE_LOOP({
      (void)o;
      y[(co >> 2) * plane * 4 + (2 * oy + ((co >> 1) & 1)) * (2 * a.wo) + 2 * ox + (co & 1)] = v;
    })
  } else if (r1 && r2) {
    EDVR_STORE_LOOP({ y[o] = v + r1[o] + r2[o]; })
  } else if (r1) {
    EDVR_STORE_LOOP({ y[o] = v + r1[o]; })
  } else {
    EDVR_STORE_LOOP({ y[o] = v; })
  }
#undef EDVR_STORE_LOOP
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
static inline size_t direct_packed_elems(int co, int ci, int ks) {
  return (size_t)round_up(ci, ks == 1 ? 32 : 16) * ks * ks * round_up(co, 32);
}

template <int KS, int STRIDE, int MT, int SW>
static int launch_one(const ConvArgs &a, int co_start, int co_blocks, hipStream_t stream) {
  constexpr int NS = MT >= 3 ? 1 : 2;
  constexpr int SH = 32 / SW, TH = 4 * NS * SH, TW = SW;
  ConvArgs b = a;
  b.tiles_x = cdiv(a.wo, TW);
  b.tiles_y = cdiv(a.ho, TH);
  b.co_start = co_start;
  dim3 grid(b.tiles_x * b.tiles_y, co_blocks, a.d.n);
  hipLaunchKernelGGL((conv2d_mfma_kernel<KS, STRIDE, MT, SW, NS>), grid, dim3(256), 0, stream, b);
  return check_launch("conv2d_mfma_kernel");
}

static inline bool use_sw16(int ho, int wo, int ns = 2) {
  // pick the tile geometry (4ns x 32 or 8ns x 16) that wastes fewer lanes on this output size
  const int64_t w32 = (int64_t)cdiv(ho, 4 * ns) * 4 * ns * cdiv(wo, 32) * 32;
  const int64_t w16 = (int64_t)cdiv(ho, 8 * ns) * 8 * ns * cdiv(wo, 16) * 16;
  return w16 < w32;
}

template <int KS, int STRIDE, int MT>
static int launch_sw(const ConvArgs &a, int co_start, int co_blocks, hipStream_t stream) {
  return use_sw16(a.ho, a.wo, MT >= 3 ? 1 : 2) ? launch_one<KS, STRIDE, MT, 16>(a, co_start, co_blocks, stream)
                                               : launch_one<KS, STRIDE, MT, 32>(a, co_start, co_blocks, stream);
}

template <int KS, int STRIDE>
static int launch_mt(const ConvArgs &a, hipStream_t stream) {
  // full 128-channel blocks with MT = 4, then one exact-size tail launch (no masked MFMA work)
  const int full = a.d.co / 128, rem_tiles = cdiv(a.d.co - full * 128, 32);
  int rc = EDVR_OK;
  if (full > 0) rc = launch_sw<KS, STRIDE, 4>(a, 0, full, stream);
  if (rc || rem_tiles == 0) return rc;
  switch (rem_tiles) {
    case 1: return launch_sw<KS, STRIDE, 1>(a, full * 128, 1, stream);
    case 2: return launch_sw<KS, STRIDE, 2>(a, full * 128, 1, stream);
    case 3: return launch_sw<KS, STRIDE, 3>(a, full * 128, 1, stream);
    default: return launch_sw<KS, STRIDE, 4>(a, full * 128, 1, stream);  // 97..127 channels left
  }
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
  a.co_start = 0;
  const bool scaled = d.y_scale != 0.f && d.y_scale != 1.f;
  if ((d.gate || scaled) && (d.ks != 3 || d.out_mode != EDVR_OUT_NCHW)) {
    set_error("conv2d: gate / y_scale need a 3x3 kernel and the NCHW output mode");
    return EDVR_ERR_UNSUPPORTED;
  }
  if (d.gate && d.act == EDVR_ACT_SIGMOID) {
    set_error("conv2d: gate together with a sigmoid epilogue is not supported");
    return EDVR_ERR_UNSUPPORTED;
  }
  if (d.abs_sum && (conv_small_eligible(d) || !(winograd_f4_eligible(d) || winograd_f4s_eligible(d)) || d.out_mode != EDVR_OUT_NCHW)) {
    set_error("conv2d: abs_sum is an epilogue of the F(4x4) Winograd kernel's NCHW store only (ask edvr_conv2d_abs_sum_supported)");
    return EDVR_ERR_UNSUPPORTED;
  }
  if (d.y_amax && (conv_small_eligible(d) || !(winograd_f4s_eligible(d) || conv1x1_split_eligible(d)))) {
    set_error("conv2d: y_amax is an epilogue of the split-operand kernels only (ask edvr_conv2d_y_amax_supported)");
    return EDVR_ERR_UNSUPPORTED;
  }
  if (!d.gate && !scaled && conv_small_eligible(d)) return conv_small_launch(d, stream);
  if (winograd_f4s_eligible(d)) return winograd_f4s_launch(d, stream);
  if (winograd_f4_eligible(d)) return winograd_f4_launch(d, stream);
  if (winograd_eligible(d)) {
    const float *U = d.wpk + direct_packed_elems(d.co, a.ci, 3);
    return winograd_launch(d, U, round_up(d.co, 64), stream);
  }
  if (d.ks == 3 && d.stride == 1) return launch_mt<3, 1>(a, stream);
  if (d.ks == 3 && d.stride == 2) return launch_mt<3, 2>(a, stream);
  if (conv1x1_split_eligible(d)) return conv1x1_split_launch(d, stream);
  if (conv1x1_eligible(d)) return conv1x1_launch(d, stream);
  return launch_mt<1, 1>(a, stream);
}

}  // namespace edvr

extern "C" {

size_t edvr_conv2d_packed_weight_elems(int co, int ci, int ks) {
  // direct layout [ci_pad][ks*ks][co_pad32]; for 3x3 kernels followed by the Winograd-transformed weights [ci_pad][16][co_pad64]
  return edvr::direct_packed_elems(co, ci, ks) + (ks == 3 ? (size_t)edvr::round_up(ci, 16) * 16 * edvr::round_up(co, 64) : 0);
}

int edvr_conv2d_pack_weight_f32(const float *w, float *wpk, int co, int ci, int ks, int transpose_flip,
                                edvr_stream_t stream) {
  EDVR_REQUIRE(w && wpk && co > 0 && ci > 0 && ks > 0, "pack_weight: bad arguments");
  const int cop = edvr::round_up(co, 32), cip = edvr::round_up(ci, ks == 1 ? 32 : 16), kk = ks * ks;
  const int64_t total = (int64_t)cip * kk * cop;
  const int blocks = (int)std::min<int64_t>(edvr::cdiv64(total, 256), 4096);
  if (ks == 3)  // one launch writes both the direct layout and the Winograd-transformed weights
    return edvr::winograd_pack(w, wpk + edvr::direct_packed_elems(co, ci, 3), co, ci, edvr::round_up(co, 64), cip, transpose_flip, wpk, cop,
                               edvr::as_stream(stream));
  hipLaunchKernelGGL(edvr::pack_weight_kernel, dim3(blocks), dim3(256), 0, edvr::as_stream(stream), w, wpk, co, ci, kk,
                     cop, cip, transpose_flip);
  return edvr::check_launch("pack_weight_kernel");
}

int edvr_conv2d_kernel_name(const edvr_conv2d_desc *d, char *buf, size_t buf_len) {
  EDVR_REQUIRE(d && buf && buf_len > 0, "kernel_name: bad arguments");
  const int pad = d->ks / 2;
  const int ho = (d->h + 2 * pad - d->ks) / d->stride + 1, wo = (d->w + 2 * pad - d->ks) / d->stride + 1;
  if (edvr::conv_small_eligible(*d)) {
    snprintf(buf, buf_len, "conv3x3_smallco_kernel");
    return EDVR_OK;
  }
  if (edvr::winograd_f4s_eligible(*d)) {
    snprintf(buf, buf_len, "conv3x3_winograd_f4s_kernel");
    return EDVR_OK;
  }
  if (edvr::winograd_f4_eligible(*d)) {
    snprintf(buf, buf_len, "conv3x3_winograd_f4_kernel");
    return EDVR_OK;
  }
  if (edvr::winograd_eligible(*d)) {
    snprintf(buf, buf_len, "conv3x3_winograd_kernel");
    return EDVR_OK;
  }
  if (edvr::conv1x1_split_eligible(*d)) {
    snprintf(buf, buf_len, "conv1x1_split_kernel");
    return EDVR_OK;
  }
  if (edvr::conv1x1_eligible(*d)) {
    snprintf(buf, buf_len, "conv1x1_stream_kernel");
    return EDVR_OK;
  }
  const int mt = d->co >= 128 ? 4 : edvr::cdiv(d->co, 32);  // the launch carrying most of the work
  const int ns = mt >= 3 ? 1 : 2;
  snprintf(buf, buf_len, "conv2d_mfma_kernel<%d, %d, %d, %d, %d>", d->ks, d->stride, mt, edvr::use_sw16(ho, wo, ns) ? 16 : 32, ns);
  return EDVR_OK;
}

int edvr_conv2d_executed_flops(const edvr_conv2d_desc *d, double *flops) {
  EDVR_REQUIRE(d && flops, "executed_flops: bad arguments");
  const int pad = d->ks / 2;
  const double ho = (d->h + 2 * pad - d->ks) / d->stride + 1, wo = (d->w + 2 * pad - d->ks) / d->stride + 1;
  if (!edvr::conv_small_eligible(*d) && edvr::winograd_f4s_eligible(*d)) *flops = edvr::winograd_f4s_executed_flops(*d);
  else if (!edvr::conv_small_eligible(*d) && edvr::winograd_f4_eligible(*d)) *flops = edvr::winograd_f4_executed_flops(*d);
  else if (!edvr::conv_small_eligible(*d) && edvr::winograd_eligible(*d)) *flops = edvr::winograd_executed_flops(*d);
  else if (edvr::conv1x1_split_eligible(*d)) *flops = 8.0 * d->n * ho * wo * d->co * (d->c1 + d->c2);  // four f16 products per fp32 one
  else *flops = 2.0 * d->n * ho * wo * d->co * (d->c1 + d->c2) * d->ks * d->ks;  // direct algorithm (tile padding not counted)
  return EDVR_OK;
}

int edvr_conv2d_abs_sum_supported(const edvr_conv2d_desc *d) {
  if (!d) return 0;
  return (!edvr::conv_small_eligible(*d) && (edvr::winograd_f4_eligible(*d) || edvr::winograd_f4s_eligible(*d)) && d->out_mode == EDVR_OUT_NCHW) ? 1 : 0;  // (the PixelShuffle store has no such sum)
}

int edvr_conv2d_y_amax_supported(const edvr_conv2d_desc *d) {
  if (!d) return 0;
  return (!edvr::conv_small_eligible(*d) && (edvr::winograd_f4s_eligible(*d) || edvr::conv1x1_split_eligible(*d))) ? 1 : 0;
}

int edvr_conv2d_gate_supported(const edvr_conv2d_desc *d) {
  if (!d) return 0;
  edvr_conv2d_desc q = *d;
  if (!q.gate) q.gate = reinterpret_cast<const float *>(&q);  // any non-null pointer (callers probe with x1 unset): only the eligibility rules are evaluated
  return edvr::winograd_eligible(q) ? 1 : 0;  // (the direct kernel takes a gate too - correct, slower; this asks for the fused-fast path)
}

int edvr_conv2d_f32(const edvr_conv2d_desc *d, edvr_stream_t stream) {
  EDVR_REQUIRE(d != nullptr, "conv2d: null descriptor");
  return edvr::conv2d_launch(*d, edvr::as_stream(stream));
}

}  // extern "C"
