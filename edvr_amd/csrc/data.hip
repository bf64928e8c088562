// data.hip - the last step of the input pipeline on the device: decoded uint8 HWC patches -> the float32
// (clip, frame, channel, y, x) tensors the network consumes, with the per-clip augmentation applied on the way.
//
// Reference: imfrombytes(float32=True) divides by 255 on the host (basicsr/utils/img_util.py:101-123), paired_random_crop /
// augment permute float arrays per image (basicsr/data/transforms.py:25-151), img2tensor swaps BGR->RGB and goes HWC->CHW
// (img_util.py:9-33), the DataLoader collates and CUDAPrefetcher copies 4 bytes per sample over PCIe (prefetch_dataloader.py:
// 84-126).  Everything after the decode is a permutation plus one exact division, so it commutes: the host here only crops
// bytes, 1 byte per sample crosses PCIe, and this kernel does flip / transpose / channel swap / HWC->CHW / division in one pass.
//
// Byte work, HBM-bound (3 B in, 12 B out per pixel).  A workgroup owns a 32x32 OUTPUT tile; its source tile (the transposed /
// mirrored one under augmentation) is read row-contiguously into LDS whatever the flags are, so reads and writes stay coalesced
// for all 8 augmentation states.
#include <algorithm>

#include "common.h"

namespace edvr {

typedef float f32x4 __attribute__((ext_vector_type(4)));

struct FramesArgs {
  const uint8_t *src;  // [n][h][w][3]
  float *dst;          // [n][3][ho][wo]; a transposed clip has (ho, wo) = (w, h), and the host admits it only when h == w
  int n, frames_per_clip, h, w, swap_rb, clip0;
  uint8_t flags[256];  // per clip, relative to clip0: EDVR_AUG_HFLIP | EDVR_AUG_VFLIP | EDVR_AUG_ROT90
};

// byte / 255 in float32, correctly rounded (== numpy's float32 division for all 256 inputs, checked exhaustively by
// tests/test_gpu_data.py::test_division_is_numpy_division): q = u * fl(1/255), one Newton correction with the exact remainder.
// 3 VALU instructions instead of the ~10 of the IEEE division sequence - the kernel converts 3 values per 15 bytes of traffic.
__device__ __forceinline__ float div255(unsigned u) {
  const float r = 1.f / 255.f, f = (float)u;
  const float q = __fmul_rn(f, r);
  return __fmaf_rn(__fmaf_rn(-q, 255.f, f), r, q);
}

// FAST: w % 4 == 0 (every REDS / Vimeo size and every training patch) - each 32-pixel row segment of a tile is then 96 bytes at
// a 4-byte aligned address: dword loads, dword LDS traffic, one 16-byte store per channel for 4 pixels of a row.  The generic
// instantiation moves single bytes.
template <bool FAST>
__global__ __launch_bounds__(256) void frames_u8_to_f32_kernel(const FramesArgs a) {
  constexpr int T = 32, ROWW = T * 3 / 4 + 1, ROW = ROWW * 4;  // 96 payload bytes per LDS row + one dword of padding
  __shared__ uint32_t tile32[T][ROWW];
  uint8_t(*tile)[ROW] = reinterpret_cast<uint8_t(*)[ROW]>(tile32);
  const int img = blockIdx.z;
  const int fl = a.flags[img / a.frames_per_clip];
  const bool hflip = fl & EDVR_AUG_HFLIP, vflip = fl & EDVR_AUG_VFLIP, rot = fl & EDVR_AUG_ROT90;
  const int h = a.h, w = a.w;
  const int ho = rot ? w : h, wo = rot ? h : w;
  const int oy0 = blockIdx.y * T, ox0 = blockIdx.x * T;
  // augment(): hflip, then vflip, then transpose.  Output (oy, ox) = mirrored image at (y, x) = rot ? (ox, oy) : (oy, ox)
  const int y0 = rot ? ox0 : oy0, x0 = rot ? oy0 : ox0;
  const int sy_base = vflip ? h - T - y0 : y0, sx_base = hflip ? w - T - x0 : x0;  // LDS (r, q) <- source (sy_base + r, sx_base + q)
  const int64_t first = (int64_t)(a.clip0 * a.frames_per_clip) + img;
  const uint8_t *src = a.src + first * h * w * 3;
  float *dst = a.dst + first * 3 * h * w;
  const int tid = threadIdx.x;
  if (FAST) {
    const uint32_t *src32 = reinterpret_cast<const uint32_t *>(src);
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const int e = tid + k * 256, r = e / 24, dq = e - r * 24;  // 24 dwords per row segment
      const int sy = sy_base + r, bx = sx_base * 3 + dq * 4;     // byte position in the source row: a multiple of 4 (w % 4 == 0)
      if (sy >= 0 && sy < h && bx >= 0 && bx < w * 3) tile32[r][dq] = src32[((int64_t)sy * w * 3 + bx) >> 2];
    }
  } else {
    for (int e = tid; e < T * T * 3; e += 256) {
      const int r = e / (T * 3), bq = e - r * (T * 3);
      const int sy = sy_base + r, sx = sx_base + bq / 3;
      if (sy >= 0 && sy < h && sx >= 0 && sx < w) tile[r][bq] = src[((int64_t)sy * w + sx_base) * 3 + bq];
    }
  }
  __syncthreads();
  if (FAST) {  // thread = 4 pixels of one output row (wo % 4 == 0 as well: a transposed clip is square)
    const int oy = oy0 + (tid >> 3), oxq = (tid & 7) * 4, ox = ox0 + oxq;
    if (oy >= ho || ox >= wo) return;
    uint32_t px[4];  // px[i] = bytes of output pixel ox + i, source channel order
    if (!rot) {
      const int r = vflip ? T - 1 - (tid >> 3) : (tid >> 3);
      const int d0 = hflip ? 21 - 3 * (tid & 7) : 3 * (tid & 7);  // the 12 bytes of LDS pixels [q0, q0 + 4), q0 = hflip ? 28 - oxq : oxq
      const uint32_t w0 = tile32[r][d0], w1 = tile32[r][d0 + 1], w2 = tile32[r][d0 + 2];
      const uint32_t p0 = w0 & 0xffffffu, p1 = (w0 >> 24) | ((w1 & 0xffffu) << 8), p2 = (w1 >> 16) | ((w2 & 0xffu) << 16), p3 = w2 >> 8;
      px[0] = hflip ? p3 : p0, px[1] = hflip ? p2 : p1, px[2] = hflip ? p1 : p2, px[3] = hflip ? p0 : p3;
    } else {
      const int q = hflip ? T - 1 - (tid >> 3) : (tid >> 3);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int r = vflip ? T - 1 - (oxq + i) : oxq + i;
        px[i] = tile[r][q * 3] | (tile[r][q * 3 + 1] << 8) | (tile[r][q * 3 + 2] << 16);
      }
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const int sh = 8 * (a.swap_rb ? 2 - c : c);
      f32x4 v;
#pragma unroll
      for (int i = 0; i < 4; ++i) v[i] = div255((px[i] >> sh) & 0xffu);
      *reinterpret_cast<f32x4 *>(dst + ((int64_t)c * ho + oy) * wo + ox) = v;
    }
  } else {
    const int ox = ox0 + (tid & 31);
    if (ox >= wo) return;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int oy = oy0 + (tid >> 5) + 8 * k;
      if (oy >= ho) break;
      const int dy = (rot ? ox : oy) - y0, dx = (rot ? oy : ox) - x0;
      const int r = vflip ? T - 1 - dy : dy, q = hflip ? T - 1 - dx : dx;
#pragma unroll
      for (int c = 0; c < 3; ++c) dst[((int64_t)c * ho + oy) * wo + ox] = div255(tile[r][q * 3 + (a.swap_rb ? 2 - c : c)]);
    }
  }
}

}  // namespace edvr

extern "C" int edvr_frames_u8_to_f32(const uint8_t *src, float *dst, int n_clips, int frames_per_clip, int h, int w,
                                     const uint8_t *clip_flags, int swap_rb, edvr_stream_t stream) {
  using namespace edvr;
  if (!src || !dst || n_clips < 0 || frames_per_clip <= 0 || h <= 0 || w <= 0) {
    set_error("edvr_frames_u8_to_f32: bad argument");
    return EDVR_ERR_ARG;
  }
  bool any_rot = false;
  for (int i = 0; clip_flags && i < n_clips; ++i) any_rot |= (clip_flags[i] & EDVR_AUG_ROT90) != 0;
  if (any_rot && h != w) {  // a transposed clip in a batch of (h, w) clips cannot be collated (nor can the reference's)
    set_error("edvr_frames_u8_to_f32: EDVR_AUG_ROT90 needs square patches (%d x %d)", h, w);
    return EDVR_ERR_ARG;
  }
  for (int c0 = 0; c0 < n_clips; c0 += 256) {
    FramesArgs a;
    a.src = src, a.dst = dst, a.frames_per_clip = frames_per_clip, a.h = h, a.w = w, a.swap_rb = swap_rb, a.clip0 = c0;
    const int nc = std::min(256, n_clips - c0);
    a.n = nc * frames_per_clip;
    for (int i = 0; i < 256; ++i) a.flags[i] = (clip_flags && i < nc) ? clip_flags[c0 + i] : 0;
    if (a.n > 65535) {
      set_error("edvr_frames_u8_to_f32: more than 65535 frames in 256 clips");
      return EDVR_ERR_ARG;
    }
    const dim3 grid(cdiv(w, 32), cdiv(h, 32), a.n);
    if (w % 4 == 0 && reinterpret_cast<uintptr_t>(src) % 4 == 0 && reinterpret_cast<uintptr_t>(dst) % 16 == 0)
      hipLaunchKernelGGL(frames_u8_to_f32_kernel<true>, grid, dim3(256), 0, (hipStream_t)stream, a);
    else
      hipLaunchKernelGGL(frames_u8_to_f32_kernel<false>, grid, dim3(256), 0, (hipStream_t)stream, a);
  }
  return check_launch("frames_u8_to_f32_kernel");
}
