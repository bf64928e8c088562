// metrics.hip - PSNR of the network output on the device for gfx950.
//
// The reference's validation (basicsr/models/video_base_model.py:19-119) copies every output frame to the host, converts it with
// tensor2img (basicsr/utils/img_util.py:36-98: clamp to [0,1], x255, round, uint8) and runs calculate_psnr in NumPy
// (basicsr/metrics/psnr_ssim.py:7-51).  This kernel computes, per image, the sum of squared differences of the two
// clamp-round-uint8 images directly from the fp32 tensors: integer arithmetic, so the RGB result is exact and only n doubles
// leave the device.  test_y_channel follows to_y_channel / bgr2ycbcr(y_only) (metric_util.py:34-47, matlab_functions.py:207-240)
// on float32 values, accumulated in double.
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

}  // extern "C"
