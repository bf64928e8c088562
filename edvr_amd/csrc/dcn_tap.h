// dcn_tap.h - one bilinear tap of the deformable sampling grid with the reference's bounds rules
// (deform_conv_cuda_kernel.cu:481-491 corners outside the image contribute 0; :618 the tap counts iff -1 < h < H, -1 < w < W).
#pragma once

namespace edvr {

struct Tap {
  float w00, w01, w10, w11;  // bilinear corner weights, 0 where the corner is outside the image
  int o00, o01, o10, o11;    // clamped element offsets inside one channel plane
  float lh, lw;
  bool ok00, ok01, ok10, ok11;  // corner inside the image AND tap valid
  int h0, w0;                   // unclamped integer coordinates of corner 00
};

__device__ __forceinline__ Tap resolve_tap(float h, float w, int H, int W) {
  Tap t;
  const bool valid = (h > -1.f) && (w > -1.f) && (h < (float)H) && (w < (float)W);
  const float fh = floorf(h), fw = floorf(w);
  const int h0 = (int)fh, w0 = (int)fw, h1 = h0 + 1, w1 = w0 + 1;
  t.h0 = h0;
  t.w0 = w0;
  t.lh = h - fh;
  t.lw = w - fw;
  const float hh = 1.f - t.lh, hw = 1.f - t.lw;
  const bool r0 = valid && h0 >= 0, r1 = valid && h1 <= H - 1;
  const bool c0 = w0 >= 0, c1 = w1 <= W - 1;
  t.ok00 = r0 && c0;
  t.ok01 = r0 && c1;
  t.ok10 = r1 && c0;
  t.ok11 = r1 && c1;
  t.w00 = t.ok00 ? hh * hw : 0.f;
  t.w01 = t.ok01 ? hh * t.lw : 0.f;
  t.w10 = t.ok10 ? t.lh * hw : 0.f;
  t.w11 = t.ok11 ? t.lh * t.lw : 0.f;
  const int ch0 = min(max(h0, 0), H - 1), ch1 = min(max(h1, 0), H - 1);
  const int cw0 = min(max(w0, 0), W - 1), cw1 = min(max(w1, 0), W - 1);
  t.o00 = ch0 * W + cw0;
  t.o01 = ch0 * W + cw1;
  t.o10 = ch1 * W + cw0;
  t.o11 = ch1 * W + cw1;
  return t;
}

// Neighbour exchange inside a wave (DPP wave shifts, no LDS): lane i reads lane i+1 / lane i-1; lanes without a neighbour get 0.
__device__ __forceinline__ int lane_next_i(int v) { return __builtin_amdgcn_update_dpp(0, v, 0x130 /* wave_shl:1 */, 0xf, 0xf, false); }
__device__ __forceinline__ int lane_prev_i(int v) { return __builtin_amdgcn_update_dpp(0, v, 0x138 /* wave_shr:1 */, 0xf, 0xf, false); }
__device__ __forceinline__ float lane_prev_f(float v) { return __int_as_float(lane_prev_i(__float_as_int(v))); }
__device__ __forceinline__ float lane_next_f(float v) { return __int_as_float(lane_next_i(__float_as_int(v))); }

}  // namespace edvr
