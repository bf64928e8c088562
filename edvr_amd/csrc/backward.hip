// backward.hip - gradients of the HBM-bound glue of the EDVR hot path + the training loss (gfx950).
//
// These replace the ATen backward kernels autograd would run for the reference graph
// (basicsr/models/archs/edvr_arch.py) during SRModel.optimize_parameters (sr_model.py:88-112):
// Upsample(x2 bilinear), MaxPool2d/AvgPool2d(3,2,1), the TSA temporal attention (:171-184), the TSA
// output combine (:210-213), PixelShuffle(2), and CharbonnierLoss (losses/losses.py:23-25, reduction sum).
// All are written in GATHER form (one thread per gradient element, no atomics) so they are
// deterministic, with lanes along the contiguous pixel axis.
#include "common.h"

namespace edvr {

__device__ __forceinline__ float sigmoidf_(float v) { return 1.f / (1.f + __expf(-v)); }
static inline unsigned grid_for(int64_t total) { return (unsigned)std::max<int64_t>(1, std::min<int64_t>(cdiv64(total, 256), 256 * 16)); }

// y (n, 4c, h, w) <- x (n, c, 2h, 2w): inverse of PixelShuffle(2) (gradient of the fused upconv epilogue)
__global__ __launch_bounds__(256) void pixel_unshuffle2_kernel(const float *__restrict__ x, float *__restrict__ y, int64_t total, int c4,
                                                               int h, int w) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int ox = (int)(i % w);
    const int oy = (int)((i / w) % h);
    const int co = (int)((i / ((int64_t)w * h)) % c4);
    const int64_t n = i / ((int64_t)w * h * c4);
    const int oc = co >> 2, sy = (co >> 1) & 1, sx = co & 1;
    y[i] = x[((n * (c4 >> 2) + oc) * (2 * h) + 2 * oy + sy) * (int64_t)(2 * w) + 2 * ox + sx];
  }
}

// The same with the activation backward folded in (gradient of `PixelShuffle(act(conv))`, the two up-convolutions of the tail): dz =
// unshuffle(dy * act'(y)), y = the shuffled activation output.  One thread = one 2 x 2 block of the shuffled tensor = the same
// pixel of four channel planes: two 8-byte loads per input, four plane-coalesced stores (the plain kernel above gathers every
// second float of a row per thread; as two launches the gradient made two round trips over the largest tensors of the step).
__global__ __launch_bounds__(256) void pixel_unshuffle2_act_bwd_kernel(const float *__restrict__ dy, const float *__restrict__ y,
                                                                       float *__restrict__ dz, int64_t blocks, int h, int w, int act) {
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < blocks; i += (int64_t)gridDim.x * 256) {
    const int ox = (int)(i % w);
    const int oy = (int)((i / w) % h);
    const int64_t pl = i / ((int64_t)w * h);  // n * c + oc
    const int64_t src = (pl * (2 * h) + 2 * oy) * (int64_t)(2 * w) + 2 * ox;
    float g[4], v[4] = {1.f, 1.f, 1.f, 1.f};
#pragma unroll
    for (int sy = 0; sy < 2; ++sy) {
      const f32x2 a = *reinterpret_cast<const f32x2 *>(dy + src + (int64_t)sy * 2 * w);
      g[2 * sy] = a[0];
      g[2 * sy + 1] = a[1];
      if (y) {
        const f32x2 b = *reinterpret_cast<const f32x2 *>(y + src + (int64_t)sy * 2 * w);
        v[2 * sy] = b[0];
        v[2 * sy + 1] = b[1];
      }
    }
    float *dst = dz + (pl * 4 * h + oy) * (int64_t)w + ox;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      float r = g[k];
      if (act == EDVR_ACT_RELU) r = v[k] > 0.f ? r : 0.f;
      else if (act == EDVR_ACT_LRELU) r = v[k] > 0.f ? r : 0.1f * r;
      else if (act == EDVR_ACT_SIGMOID) r = r * v[k] * (1.f - v[k]);
      dst[(int64_t)k * h * w] = r;
    }
  }
}

// z (n, c, H, W) with z[2oy, 2ox] = dz[oy, ox], zero elsewhere: stride-2 data gradient = stride-1 conv of z
__global__ __launch_bounds__(256) void zero_stuff2_kernel(const float *__restrict__ dz, float *__restrict__ z, int64_t total, int H, int W,
                                                          int ho, int wo) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int x = (int)(i % W);
    const int y = (int)((i / W) % H);
    const int64_t pl = i / ((int64_t)W * H);
    float v = 0.f;
    if (!(x & 1) && !(y & 1) && (y >> 1) < ho && (x >> 1) < wo) v = dz[(pl * ho + (y >> 1)) * wo + (x >> 1)];
    z[i] = v;
  }
}

// dst[b, center, i] += sum_t src[b, t, i]  (gradient of a reference frame broadcast over the t frames of its clip)
__global__ __launch_bounds__(256) void frame_reduce_add_kernel(const float *__restrict__ src, float *__restrict__ dst, int64_t total, int t,
                                                               int center, int64_t chw) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int64_t b = i / chw, e = i - b * chw;
    float s = 0.f;
    for (int k = 0; k < t; ++k) s += src[(b * t + k) * chw + e];
    dst[(b * t + center) * chw + e] += s;
  }
}

__device__ __forceinline__ void src_index2(int dst, int in, int &i0, int &i1, float &l) {
  float s = ((float)dst + 0.5f) * 0.5f - 0.5f;
  if (s < 0.f) s = 0.f;
  i0 = (int)s;
  i1 = i0 + ((i0 < in - 1) ? 1 : 0);
  l = s - (float)i0;
}

// dx (nc, h, w) <- dy (nc, 2h, 2w) of y = scale * bilinear_x2(x)
__global__ __launch_bounds__(256) void upsample2x_bwd_kernel(const float *__restrict__ dy, float *__restrict__ dx, int64_t total, int h, int w,
                                                             float scale) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int ix = (int)(i % w);
    const int iy = (int)((i / w) % h);
    const int64_t pl = i / ((int64_t)w * h);
    const float *g = dy + pl * (4 * (int64_t)h * w);
    float s = 0.f;
#pragma unroll
    for (int a = -1; a <= 2; ++a) {
      const int oy = 2 * iy + a;
      if (oy < 0 || oy >= 2 * h) continue;
      int y0, y1;
      float ly;
      src_index2(oy, h, y0, y1, ly);
      const float wy = (y0 == iy ? 1.f - ly : 0.f) + (y1 == iy ? ly : 0.f);
      if (wy == 0.f) continue;
#pragma unroll
      for (int b = -1; b <= 2; ++b) {
        const int ox = 2 * ix + b;
        if (ox < 0 || ox >= 2 * w) continue;
        int x0, x1;
        float lx;
        src_index2(ox, w, x0, x1, lx);
        const float wx = (x0 == ix ? 1.f - lx : 0.f) + (x1 == ix ? lx : 0.f);
        s += wy * wx * g[(int64_t)oy * (2 * w) + ox];
      }
    }
    dx[i] = s * scale;
  }
}

// The same for even widths and 16-byte aligned rows: the adjoint of the x2 bilinear map has the fixed taps (1/4, 3/4, 3/4, 1/4) per
// axis on output rows / columns 2i - 1 .. 2i + 2 (at the borders the clamped source index folds the outer tap into its neighbour:
// 0, 1, 3/4, 1/4 at i = 0 and 1/4, 3/4, 1, 0 at the last index).  A thread owns input columns (2j, 2j + 1) of one row: four 16-byte
// loads of dy (columns 4j .. 4j + 3 of its four rows), the columns 4j - 1 / 4j + 4 come from the neighbouring lanes - 4 loads for 2
// results where the kernel above issues 32 scalar ones behind its generic index logic (0.17 of the HBM rate in the training step).
__global__ __launch_bounds__(256) void upsample2x_bwd_wide_kernel(const float *__restrict__ dy, float *__restrict__ dx, int nc, int h, int w, float scale) {
  typedef float f32x4 __attribute__((ext_vector_type(4)));
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  const int wq = w >> 1, lane = threadIdx.x & 63, wo = 2 * w, ho = 2 * h;
  const int64_t total = (int64_t)nc * h * wq;
  const int64_t rounded = (total + 255) / 256 * 256;
  for (int64_t base = (int64_t)blockIdx.x * 256; base < rounded; base += (int64_t)gridDim.x * 256) {
    const int64_t idx_raw = base + threadIdx.x;
    const bool active = idx_raw < total;
    const int64_t idx = active ? idx_raw : total - 1;  // (inactive lanes still take part in the lane exchanges)
    const int j = (int)(idx % wq);
    const int iy = (int)((idx / wq) % h);
    const int64_t pl = idx / ((int64_t)wq * h);
    const float *g = dy + pl * ((int64_t)ho * wo);
    // row taps: output rows 2 iy - 1 .. 2 iy + 2
    const float wy[4] = {iy == 0 ? 0.f : 0.25f, iy == 0 ? 1.f : 0.75f, iy == h - 1 ? 1.f : 0.75f, iy == h - 1 ? 0.f : 0.25f};
    // column taps of input column 2j (output columns 4j - 1 .. 4j + 2) and 2j + 1 (4j + 1 .. 4j + 4)
    const bool first = j == 0, last = j == wq - 1;
    const float a0 = first ? 0.f : 0.25f, a1 = first ? 1.f : 0.75f, b2 = last ? 1.f : 0.75f, b3 = last ? 0.f : 0.25f;
    float s0 = 0.f, s1 = 0.f;
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      const int oy = min(max(2 * iy - 1 + a, 0), ho - 1);  // (clamped rows carry the tap 0)
      const f32x4 m = *reinterpret_cast<const f32x4 *>(g + (int64_t)oy * wo + 4 * j);
      const float from_left = __shfl_up(m[3], 1, 64), from_right = __shfl_down(m[0], 1, 64);
      float l = first ? 0.f : from_left, r = last ? 0.f : from_right;
      if (lane == 0 && !first) l = g[(int64_t)oy * wo + 4 * j - 1];
      if (lane == 63 && !last) r = g[(int64_t)oy * wo + 4 * j + 4];
      const float h0 = __builtin_fmaf(a0, l, __builtin_fmaf(a1, m[0], __builtin_fmaf(0.75f, m[1], 0.25f * m[2])));
      const float h1 = __builtin_fmaf(0.25f, m[1], __builtin_fmaf(0.75f, m[2], __builtin_fmaf(b2, m[3], b3 * r)));
      s0 = __builtin_fmaf(wy[a], h0, s0);
      s1 = __builtin_fmaf(wy[a], h1, s1);
    }
    if (active) *reinterpret_cast<f32x2 *>(dx + (pl * h + iy) * (int64_t)w + 2 * j) = f32x2{s0 * scale, s1 * scale};
  }
}

// dx (n, c, h, w) <- dy (n, 2c, ho, wo) of y = cat(maxpool(x), avgpool(x)), 3x3 / s2 / p1
__global__ __launch_bounds__(256) void pool_maxavg_bwd_kernel(const float *__restrict__ x, const float *__restrict__ dy, float *__restrict__ dx,
                                                              int64_t total, int c, int h, int w, int ho, int wo) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int ix = (int)(i % w);
    const int iy = (int)((i / w) % h);
    const int ch = (int)((i / ((int64_t)w * h)) % c);
    const int64_t n = i / ((int64_t)w * h * c);
    const float *src = x + (n * c + ch) * (int64_t)h * w;
    const float *gmax = dy + (n * 2 * c + ch) * (int64_t)ho * wo;
    const float *gavg = dy + (n * 2 * c + c + ch) * (int64_t)ho * wo;
    float s = 0.f;
    const int oy_lo = max(0, (iy) / 2), oy_hi = min(ho - 1, (iy + 1) / 2);
    const int ox_lo = max(0, (ix) / 2), ox_hi = min(wo - 1, (ix + 1) / 2);
    for (int oy = oy_lo; oy <= oy_hi; ++oy)
      for (int ox = ox_lo; ox <= ox_hi; ++ox) {
        // window rows 2oy-1..2oy+1, cols 2ox-1..2ox+1 contains (iy, ix) by construction
        s += gavg[oy * wo + ox] * (1.f / 9.f);
        float mx = -INFINITY;
        int arg = -1;
        for (int dyy = 0; dyy < 3; ++dyy) {
          const int yy = 2 * oy - 1 + dyy;
          if (yy < 0 || yy >= h) continue;
          for (int dxx = 0; dxx < 3; ++dxx) {
            const int xx = 2 * ox - 1 + dxx;
            if (xx < 0 || xx >= w) continue;
            const float v = src[yy * w + xx];
            if (v > mx || arg < 0) {  // first maximum in row-major order wins, like ATen
              mx = v;
              arg = yy * w + xx;
            }
          }
        }
        if (arg == iy * w + ix) s += gmax[oy * wo + ox];
      }
    dx[i] = s;
  }
}

// TSA temporal attention backward.  One thread per (b, pixel); T <= 16.
__global__ __launch_bounds__(256) void tsa_temporal_bwd_kernel(const float *__restrict__ emb, const float *__restrict__ emb_ref,
                                                               const float *__restrict__ aligned, const float *__restrict__ dout,
                                                               float *__restrict__ d_emb, float *__restrict__ d_ref, float *__restrict__ d_al,
                                                               int b, int t, int c, int hw) {
  const int64_t total = (int64_t)b * hw;
  for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
    const int p = (int)(idx % hw);
    const int bi = (int)(idx / hw);
    const float *r = emb_ref + (int64_t)bi * c * hw + p;
    float ds[16];
    for (int ti = 0; ti < t; ++ti) {
      const int64_t base = ((int64_t)(bi * t + ti) * c) * hw + p;
      float dot = 0.f, dp = 0.f;
      for (int ch = 0; ch < c; ++ch) {
        dot += emb[base + (int64_t)ch * hw] * r[(int64_t)ch * hw];
        dp += dout[base + (int64_t)ch * hw] * aligned[base + (int64_t)ch * hw];
      }
      const float pr = sigmoidf_(dot);
      ds[ti] = dp * pr * (1.f - pr);
      for (int ch = 0; ch < c; ++ch) {
        d_al[base + (int64_t)ch * hw] = dout[base + (int64_t)ch * hw] * pr;
        d_emb[base + (int64_t)ch * hw] = ds[ti] * r[(int64_t)ch * hw];
      }
    }
    for (int ch = 0; ch < c; ++ch) {
      float s = 0.f;
      for (int ti = 0; ti < t; ++ti) s += ds[ti] * emb[((int64_t)(bi * t + ti) * c + ch) * hw + p];
      d_ref[((int64_t)bi * c + ch) * hw + p] = s;
    }
  }
}

// The same in two fully parallel passes (the kernel above runs ONE thread per (clip, pixel) through t x c dependent-address loads:
// 131 072 threads on the training shape, 0.21 of the HBM rate).  Pass 1, one thread per (clip, frame, pixel): the two channel sums,
// ds = <dout, aligned> p (1 - p) into a small scratch plane, d_aligned = dout * p and d_emb = ds * ref.  Pass 2, one thread per
// (clip, channel, pixel): d_ref = sum_t ds[t] * emb[t].  Same arithmetic per element, same summation order over channels / frames.
__global__ __launch_bounds__(256) void tsa_temporal_bwd_frames_kernel(const float *__restrict__ emb, const float *__restrict__ emb_ref,
                                                                      const float *__restrict__ aligned, const float *__restrict__ dout,
                                                                      float *__restrict__ d_emb, float *__restrict__ d_al, float *__restrict__ ds_ws,
                                                                      int64_t total, int t, int c, int hw) {
  for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
    const int p = (int)(idx % hw);
    const int64_t bt = idx / hw;  // clip * t + frame
    const int64_t bi = bt / t;
    const float *r = emb_ref + bi * c * hw + p;
    const int64_t base = bt * c * hw + p;
    float dot = 0.f, dp = 0.f;
#pragma unroll 8
    for (int ch = 0; ch < c; ++ch) {
      dot += emb[base + (int64_t)ch * hw] * r[(int64_t)ch * hw];
      dp += dout[base + (int64_t)ch * hw] * aligned[base + (int64_t)ch * hw];
    }
    const float pr = sigmoidf_(dot);
    const float ds = dp * pr * (1.f - pr);
    ds_ws[idx] = ds;
#pragma unroll 8
    for (int ch = 0; ch < c; ++ch) {
      d_al[base + (int64_t)ch * hw] = dout[base + (int64_t)ch * hw] * pr;
      d_emb[base + (int64_t)ch * hw] = ds * r[(int64_t)ch * hw];
    }
  }
}

__global__ __launch_bounds__(256) void tsa_temporal_bwd_ref_kernel(const float *__restrict__ emb, const float *__restrict__ ds_ws, float *__restrict__ d_ref,
                                                                   int64_t total, int t, int c, int hw) {
  for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
    const int p = (int)(idx % hw);
    const int ch = (int)((idx / hw) % c);
    const int64_t bi = idx / ((int64_t)hw * c);
    float s = 0.f;
    for (int ti = 0; ti < t; ++ti) s += ds_ws[(bi * t + ti) * hw + p] * emb[((bi * t + ti) * c + ch) * (int64_t)hw + p];
    d_ref[idx] = s;
  }
}

// y = feat * sigmoid(attn) * 2 + add  ->  dfeat, dattn (dadd = dy)
__global__ __launch_bounds__(256) void tsa_combine_bwd_kernel(const float *__restrict__ feat, const float *__restrict__ attn,
                                                              const float *__restrict__ dy, float *__restrict__ dfeat, float *__restrict__ dattn,
                                                              int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const float s = sigmoidf_(attn[i]), g = dy[i];
    dfeat[i] = g * s * 2.f;
    dattn[i] = g * feat[i] * 2.f * s * (1.f - s);
  }
}

// loss += sum sqrt((p - t)^2 + eps);  dpred = gscale * (p - t) / sqrt(...)   (one pass, forward + backward)
__global__ __launch_bounds__(256) void charbonnier_kernel(const float *__restrict__ pred, const float *__restrict__ target, float *__restrict__ loss,
                                                          float *__restrict__ dpred, int64_t n, float eps, float gscale) {
  float s = 0.f;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const float d = pred[i] - target[i];
    const float r = sqrtf(d * d + eps);
    s += r;
    if (dpred) dpred[i] = gscale * d / r;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o, 64);
  __shared__ float red[4];
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) unsafeAtomicAdd(loss, red[0] + red[1] + red[2] + red[3]);
}

}  // namespace edvr

extern "C" {

int edvr_pixel_unshuffle2_f32(const float *x, float *y, int n, int c, int h, int w, edvr_stream_t stream) {
  using namespace edvr;
  EDVR_REQUIRE(x && y && n > 0 && c > 0 && h > 0 && w > 0, "pixel_unshuffle2: bad arguments");
  const int64_t total = (int64_t)n * 4 * c * h * w;
  hipLaunchKernelGGL(pixel_unshuffle2_kernel, dim3(grid_for(total)), dim3(256), 0, as_stream(stream), x, y, total, 4 * c, h, w);
  return check_launch("pixel_unshuffle2_kernel");
}

int edvr_pixel_unshuffle2_act_bwd_f32(const float *dy, const float *y, float *dz, int n, int c, int h, int w, int act, edvr_stream_t stream) {
  using namespace edvr;
  EDVR_REQUIRE(dy && dz && n > 0 && c > 0 && h > 0 && w > 0, "pixel_unshuffle2_act_bwd: bad arguments");
  EDVR_REQUIRE(y || act == EDVR_ACT_NONE, "pixel_unshuffle2_act_bwd: an activation needs its output");
  EDVR_REQUIRE((reinterpret_cast<uintptr_t>(dy) & 7) == 0 && (reinterpret_cast<uintptr_t>(y) & 7) == 0, "pixel_unshuffle2_act_bwd: 8-byte aligned tensors");
  const int64_t blocks = (int64_t)n * c * h * w;
  hipLaunchKernelGGL(pixel_unshuffle2_act_bwd_kernel, dim3(grid_for(blocks)), dim3(256), 0, as_stream(stream), dy, act == EDVR_ACT_NONE ? nullptr : y, dz,
                     blocks, h, w, act);
  return check_launch("pixel_unshuffle2_act_bwd_kernel");
}

int edvr_zero_stuff2_f32(const float *dz, float *z, int nc, int H, int W, int ho, int wo, edvr_stream_t stream) {
  using namespace edvr;
  EDVR_REQUIRE(dz && z && nc > 0 && H > 0 && W > 0 && ho > 0 && wo > 0, "zero_stuff2: bad arguments");
  const int64_t total = (int64_t)nc * H * W;
  hipLaunchKernelGGL(zero_stuff2_kernel, dim3(grid_for(total)), dim3(256), 0, as_stream(stream), dz, z, total, H, W, ho, wo);
  return check_launch("zero_stuff2_kernel");
}

int edvr_frame_reduce_add_f32(const float *src, float *dst, int b, int t, int center, int64_t chw, edvr_stream_t stream) {
  using namespace edvr;
  EDVR_REQUIRE(src && dst && b > 0 && t > 0 && center >= 0 && center < t && chw > 0, "frame_reduce_add: bad arguments");
  const int64_t total = (int64_t)b * chw;
  hipLaunchKernelGGL(frame_reduce_add_kernel, dim3(grid_for(total)), dim3(256), 0, as_stream(stream), src, dst, total, t, center, chw);
  return check_launch("frame_reduce_add_kernel");
}

int edvr_upsample2x_bwd_f32(const float *dy, float *dx, int nc, int h, int w, float scale, edvr_stream_t stream) {
  using namespace edvr;
  EDVR_REQUIRE(dy && dx && nc > 0 && h > 0 && w > 0, "upsample2x_bwd: bad arguments");
  const int64_t total = (int64_t)nc * h * w;
  if ((w & 1) == 0 && w >= 4 && (((int64_t)h * w) & 1) == 0 && (reinterpret_cast<uintptr_t>(dy) & 15) == 0 && (reinterpret_cast<uintptr_t>(dx) & 7) == 0) {
    hipLaunchKernelGGL(upsample2x_bwd_wide_kernel, dim3(grid_for(total / 2)), dim3(256), 0, as_stream(stream), dy, dx, nc, h, w, scale);
    return check_launch("upsample2x_bwd_wide_kernel");
  }
  hipLaunchKernelGGL(upsample2x_bwd_kernel, dim3(grid_for(total)), dim3(256), 0, as_stream(stream), dy, dx, total, h, w, scale);
  return check_launch("upsample2x_bwd_kernel");
}

int edvr_pool_maxavg_3x3s2_bwd_f32(const float *x, const float *dy, float *dx, int n, int c, int h, int w, edvr_stream_t stream) {
  using namespace edvr;
  EDVR_REQUIRE(x && dy && dx && n > 0 && c > 0 && h > 0 && w > 0, "pool_maxavg_bwd: bad arguments");
  const int64_t total = (int64_t)n * c * h * w;
  hipLaunchKernelGGL(pool_maxavg_bwd_kernel, dim3(grid_for(total)), dim3(256), 0, as_stream(stream), x, dy, dx, total, c, h, w, (h - 1) / 2 + 1,
                     (w - 1) / 2 + 1);
  return check_launch("pool_maxavg_bwd_kernel");
}

int edvr_tsa_temporal_bwd_f32(const float *emb, const float *emb_ref, const float *aligned, const float *dout, float *d_emb, float *d_emb_ref,
                              float *d_aligned, int b, int t, int c, int hw, float *ws, edvr_stream_t stream) {
  using namespace edvr;
  EDVR_REQUIRE(emb && emb_ref && aligned && dout && d_emb && d_emb_ref && d_aligned && b > 0 && t > 0 && t <= 16 && c > 0 && hw > 0,
               "tsa_temporal_bwd: bad arguments (t <= 16)");
  if (ws) {
    const int64_t frames = (int64_t)b * t * hw, refs = (int64_t)b * c * hw;
    hipLaunchKernelGGL(tsa_temporal_bwd_frames_kernel, dim3(grid_for(frames)), dim3(256), 0, as_stream(stream), emb, emb_ref, aligned, dout, d_emb,
                       d_aligned, ws, frames, t, c, hw);
    hipLaunchKernelGGL(tsa_temporal_bwd_ref_kernel, dim3(grid_for(refs)), dim3(256), 0, as_stream(stream), emb, ws, d_emb_ref, refs, t, c, hw);
    return check_launch("tsa_temporal_bwd_frames_kernel");
  }
  hipLaunchKernelGGL(tsa_temporal_bwd_kernel, dim3(grid_for((int64_t)b * hw)), dim3(256), 0, as_stream(stream), emb, emb_ref, aligned, dout, d_emb,
                     d_emb_ref, d_aligned, b, t, c, hw);
  return check_launch("tsa_temporal_bwd_kernel");
}

int edvr_tsa_combine_bwd_f32(const float *feat, const float *attn, const float *dy, float *dfeat, float *dattn, int64_t numel,
                             edvr_stream_t stream) {
  using namespace edvr;
  EDVR_REQUIRE(feat && attn && dy && dfeat && dattn && numel > 0, "tsa_combine_bwd: bad arguments");
  hipLaunchKernelGGL(tsa_combine_bwd_kernel, dim3(grid_for(numel)), dim3(256), 0, as_stream(stream), feat, attn, dy, dfeat, dattn, numel);
  return check_launch("tsa_combine_bwd_kernel");
}

int edvr_charbonnier_f32(const float *pred, const float *target, float *loss, float *dpred, int64_t numel, float eps, float grad_scale,
                         edvr_stream_t stream) {
  using namespace edvr;
  EDVR_REQUIRE(pred && target && loss && numel > 0, "charbonnier: bad arguments");
  if (hipMemsetAsync(loss, 0, sizeof(float), as_stream(stream)) != hipSuccess) {
    set_error("charbonnier: hipMemsetAsync failed");
    return EDVR_ERR_LAUNCH;
  }
  hipLaunchKernelGGL(charbonnier_kernel, dim3(std::min<unsigned>(grid_for(numel), 1024u)), dim3(256), 0, as_stream(stream), pred, target, loss,
                     dpred, numel, eps, grad_scale);
  return check_launch("charbonnier_kernel");
}

}  // extern "C"
