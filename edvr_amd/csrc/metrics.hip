// metrics.hip - PSNR of the network output on the device for gfx950.
//
// The reference's validation (basicsr/models/video_base_model.py:19-119) copies every output frame to the host, converts it with
// tensor2img (basicsr/utils/img_util.py:36-98: clamp to [0,1], x255, round, uint8) and runs calculate_psnr in NumPy
// (basicsr/metrics/psnr_ssim.py:7-51).  This kernel computes, per image, the sum of squared differences of the two
// clamp-round-uint8 images directly from the fp32 tensors: integer arithmetic, so the RGB result is exact and only n doubles
// leave the device.  test_y_channel follows to_y_channel / bgr2ycbcr(y_only) (metric_util.py:34-47, matlab_functions.py:207-240)
// on float32 values, accumulated in double.
#include <cmath>

#include "common.h"

namespace edvr {

__device__ __forceinline__ float to_u8(float v) { return rintf(fminf(fmaxf(v, 0.f), 1.f) * 255.f); }  // np.round: half to even

// Y of one pixel as to_y_channel computes it: BGR image of the reference = channels (2, 1, 0) of the RGB tensor; float32 / 255,
// dot with the BT.601 row in double, + 16, / 255 -> float32, x 255 in float32.  NOT inlined: the two images must run the very
// same instruction sequence, or identical images stop giving a difference of exactly zero (PSNR = inf in the reference).
__device__ __noinline__ float y_of_pixel(const float *__restrict__ p, int64_t o, int64_t hw) {
  const float r = __fdiv_rn(to_u8(p[o]), 255.f), g = __fdiv_rn(to_u8(p[hw + o]), 255.f), bl = __fdiv_rn(to_u8(p[2 * hw + o]), 255.f);
  const double d = (double)bl * 24.966 + (double)g * 128.553 + (double)r * 65.481 + 16.0;
  return __fmul_rn((float)(d / 255.0), 255.f);
}

// partial[img][block] = sum over this block's pixels (inside the crop) of squared differences; channels = 3 (RGB tensors) or 1
__global__ __launch_bounds__(256) void psnr_sse_kernel(const float *__restrict__ a, const float *__restrict__ b, double *__restrict__ partial,
                                                       int c, int h, int w, int64_t a_stride, int64_t b_stride, int crop, int y_channel) {
  const int img = blockIdx.y, blocks = gridDim.x;
  const int ch = h - 2 * crop, cw = w - 2 * crop;
  const int64_t npx = (int64_t)ch * cw, hw = (int64_t)h * w;
  const float *pa = a + (int64_t)img * a_stride, *pb = b + (int64_t)img * b_stride;
  unsigned long long isum = 0;
  double dsum = 0.0;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < npx; i += (int64_t)blocks * 256) {
    const int y = (int)(i / cw) + crop, x = (int)(i % cw) + crop;
    const int64_t o = (int64_t)y * w + x;
    if (y_channel && c == 3) {
      float ya[2] = {y_of_pixel(pa, o, hw), y_of_pixel(pb, o, hw)};
      const float df = __fsub_rn(ya[0], ya[1]);  // (float32 difference and square, as NumPy computes them on float32 arrays)
      dsum += (double)__fmul_rn(df, df);
    } else {
      for (int k = 0; k < c; ++k) {
        const int d = (int)to_u8(pa[k * hw + o]) - (int)to_u8(pb[k * hw + o]);
        isum += (unsigned)(d * d);
      }
    }
  }
  double s = dsum + (double)isum;  // isum < 2^53: exact
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o, 64);
  __shared__ double red[4];
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) partial[(int64_t)img * blocks + blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}


// ------------------------------------------------------------------------------------------------ SSIM
// calculate_ssim (basicsr/metrics/psnr_ssim.py:54-141): per channel, 11x11 Gaussian window (sigma 1.5, cv2.getGaussianKernel)
// correlated over the [0, 255] images in float64, valid region only ([5:-5, 5:-5]), mean of the SSIM map, then the mean over
// channels.  One 256-thread workgroup = a 16x16 tile of the valid region of one channel of one image: the 26x26 input window
// of both images goes to LDS as doubles (exact integers for RGB, the float32 Y value for test_y_channel), the separable
// Gaussian runs as a horizontal pass into LDS (5 moments: a, b, a^2, b^2, ab) and a vertical pass in registers.
struct SsimWindow {
  double g[11];  // travels in the kernel-argument block (scalar registers): no constant-memory upload, no global state
};

__global__ __launch_bounds__(256) void ssim_kernel(const float *__restrict__ a, const float *__restrict__ b, double *__restrict__ partial, int c,
                                                   int h, int w, int64_t a_stride, int64_t b_stride, int crop, int y_channel, int tiles_x,
                                                   const SsimWindow win) {
  constexpr int T = 16, R = 5, IN = T + 2 * R;  // 26
  __shared__ double va[IN][IN + 1], vb[IN][IN + 1];
  __shared__ double hs[5][IN][T + 1];
  const int tile = blockIdx.x, ch = blockIdx.y, img = blockIdx.z, tid = threadIdx.x;
  const int ty0 = (tile / tiles_x) * T, tx0 = (tile % tiles_x) * T;  // origin in the valid region = top-left of the input window in the cropped image
  const int chh = h - 2 * crop, cww = w - 2 * crop, vh = chh - 2 * R, vw = cww - 2 * R;
  const int64_t hw = (int64_t)h * w;
  const float *pa = a + (int64_t)img * a_stride, *pb = b + (int64_t)img * b_stride;
  for (int e = tid; e < IN * IN; e += 256) {
    const int r = e / IN, q = e % IN;
    const int y = min(ty0 + r, chh - 1) + crop, x = min(tx0 + q, cww - 1) + crop;  // clamped: only feeds masked outputs
    const int64_t o = (int64_t)y * w + x;
    if (y_channel && c == 3) {
      va[r][q] = (double)y_of_pixel(pa, o, hw);
      vb[r][q] = (double)y_of_pixel(pb, o, hw);
    } else {
      va[r][q] = (double)to_u8(pa[ch * hw + o]);
      vb[r][q] = (double)to_u8(pb[ch * hw + o]);
    }
  }
  __syncthreads();
  for (int e = tid; e < IN * T; e += 256) {
    const int r = e / T, q = e % T;
    double s[5] = {0, 0, 0, 0, 0};
#pragma unroll
    for (int k = 0; k < 11; ++k) {
      const double g = win.g[k], x = va[r][q + k], y = vb[r][q + k];
      s[0] += g * x;
      s[1] += g * y;
      s[2] += g * (x * x);
      s[3] += g * (y * y);
      s[4] += g * (x * y);
    }
#pragma unroll
    for (int m = 0; m < 5; ++m) hs[m][r][q] = s[m];
  }
  __syncthreads();
  const int r = tid / T, q = tid % T;
  double s[5] = {0, 0, 0, 0, 0};
#pragma unroll
  for (int k = 0; k < 11; ++k) {
    const double g = win.g[k];
#pragma unroll
    for (int m = 0; m < 5; ++m) s[m] += g * hs[m][r + k][q];
  }
  const double C1 = (0.01 * 255) * (0.01 * 255), C2 = (0.03 * 255) * (0.03 * 255);
  const double mu1_sq = s[0] * s[0], mu2_sq = s[1] * s[1], mu12 = s[0] * s[1];
  const double sig1 = s[2] - mu1_sq, sig2 = s[3] - mu2_sq, sig12 = s[4] - mu12;
  double v = ((2 * mu12 + C1) * (2 * sig12 + C2)) / ((mu1_sq + mu2_sq + C1) * (sig1 + sig2 + C2));
  if (ty0 + r >= vh || tx0 + q >= vw) v = 0.0;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
  __shared__ double red[4];
  if ((tid & 63) == 0) red[tid >> 6] = v;
  __syncthreads();
  if (tid == 0) partial[((int64_t)img * gridDim.y + ch) * gridDim.x + tile] = (red[0] + red[1]) + (red[2] + red[3]);
}

}  // namespace edvr

extern "C" {

int edvr_psnr_sse_f32(const float *a, const float *b, double *partial, int n, int c, int h, int w, int64_t a_img_stride,
                      int64_t b_img_stride, int crop_border, int y_channel, int blocks, edvr_stream_t stream) {
  using namespace edvr;
  EDVR_REQUIRE(a && b && partial && n > 0 && (c == 1 || c == 3) && h > 0 && w > 0 && blocks > 0, "psnr_sse: bad arguments");
  EDVR_REQUIRE(crop_border >= 0 && 2 * crop_border < h && 2 * crop_border < w, "psnr_sse: crop_border %d too large for %dx%d", crop_border, h, w);
  hipLaunchKernelGGL(psnr_sse_kernel, dim3(blocks, n), dim3(256), 0, as_stream(stream), a, b, partial, c, h, w,
                     a_img_stride ? a_img_stride : (int64_t)c * h * w, b_img_stride ? b_img_stride : (int64_t)c * h * w, crop_border, y_channel);
  return check_launch("psnr_sse_kernel");
}

size_t edvr_ssim_partials(int h, int w, int crop_border) {
  const int vh = h - 2 * crop_border - 10, vw = w - 2 * crop_border - 10;
  return (vh <= 0 || vw <= 0) ? 0 : (size_t)edvr::cdiv(vh, 16) * edvr::cdiv(vw, 16);
}

int edvr_ssim_f32(const float *a, const float *b, double *partial, int n, int c, int h, int w, int64_t a_img_stride, int64_t b_img_stride,
                  int crop_border, int y_channel, edvr_stream_t stream) {
  using namespace edvr;
  EDVR_REQUIRE(a && b && partial && n > 0 && (c == 1 || c == 3) && h > 0 && w > 0 && crop_border >= 0, "ssim: bad arguments");
  const int vh = h - 2 * crop_border - 10, vw = w - 2 * crop_border - 10;
  EDVR_REQUIRE(vh > 0 && vw > 0, "ssim: %dx%d with crop_border %d leaves no 11x11 window", h, w, crop_border);
  SsimWindow win;  // cv2.getGaussianKernel(11, 1.5): exp(-(i - 5)^2 / (2 sigma^2)), normalised to sum 1, float64
  double sum = 0.0;
  for (int i = 0; i < 11; ++i) sum += win.g[i] = std::exp(-0.5 / (1.5 * 1.5) * (i - 5) * (i - 5));
  for (int i = 0; i < 11; ++i) win.g[i] *= 1.0 / sum;
  const int tiles_x = cdiv(vw, 16), tiles = tiles_x * cdiv(vh, 16), chans = (y_channel && c == 3) ? 1 : c;
  hipLaunchKernelGGL(ssim_kernel, dim3(tiles, chans, n), dim3(256), 0, as_stream(stream), a, b, partial, c, h, w,
                     a_img_stride ? a_img_stride : (int64_t)c * h * w, b_img_stride ? b_img_stride : (int64_t)c * h * w, crop_border, y_channel, tiles_x, win);
  return check_launch("ssim_kernel");
}

}  // extern "C"
