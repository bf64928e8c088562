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

struct FramesArgs {
  const uint8_t *src;  // [n][h][w][3]
  float *dst;          // [n][3][ho][wo]; a transposed clip has (ho, wo) = (w, h), and the host admits it only when h == w
  int n, frames_per_clip, h, w, swap_rb, clip0;
  uint8_t flags[256];  // per clip, relative to clip0: EDVR_AUG_HFLIP | EDVR_AUG_VFLIP | EDVR_AUG_ROT90
};

__global__ __launch_bounds__(256) void frames_u8_to_f32_kernel(const FramesArgs a) {
  constexpr int T = 32, ROW = T * 3 + 4;
  __shared__ uint8_t tile[T][ROW];
  const int img = blockIdx.z;
  const int fl = a.flags[img / a.frames_per_clip];
  const bool hflip = fl & EDVR_AUG_HFLIP, vflip = fl & EDVR_AUG_VFLIP, rot = fl & EDVR_AUG_ROT90;
  const int h = a.h, w = a.w;
  const int ho = rot ? w : h, wo = rot ? h : w;
  const int oy0 = blockIdx.y * T, ox0 = blockIdx.x * T;
  // augment(): hflip, then vflip, then transpose.  Output (oy, ox) = mirrored image at (y, x) = rot ? (ox, oy) : (oy, ox)
  const int y0 = rot ? ox0 : oy0, x0 = rot ? oy0 : ox0;
  const int sy_base = vflip ? h - T - y0 : y0, sx_base = hflip ? w - T - x0 : x0;  // LDS (r, q) <- source (sy_base + r, sx_base + q)
  const uint8_t *src = a.src + ((int64_t)(a.clip0 * a.frames_per_clip) + img) * h * w * 3;
  for (int e = threadIdx.x; e < T * T * 3; e += 256) {
    const int r = e / (T * 3), b = e - r * (T * 3);
    const int sy = sy_base + r, sx = sx_base + b / 3;
    if (sy >= 0 && sy < h && sx >= 0 && sx < w) tile[r][b] = src[((int64_t)sy * w + sx_base) * 3 + b];
  }
  __syncthreads();
  float *dst = a.dst + ((int64_t)(a.clip0 * a.frames_per_clip) + img) * 3 * h * w;
  const int ox = ox0 + (threadIdx.x & 31);
  if (ox >= wo) return;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int oy = oy0 + (threadIdx.x >> 5) + 8 * k;
    if (oy >= ho) break;
    const int dy = (rot ? ox : oy) - y0, dx = (rot ? oy : ox) - x0;
    const int r = vflip ? T - 1 - dy : dy, q = hflip ? T - 1 - dx : dx;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const uint8_t u = tile[r][q * 3 + (a.swap_rb ? 2 - c : c)];
      dst[((int64_t)c * ho + oy) * wo + ox] = __fdiv_rn((float)u, 255.f);  // numpy's float32 / 255. : IEEE division, exact parity
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
    hipLaunchKernelGGL(frames_u8_to_f32_kernel, dim3(cdiv(w, 32), cdiv(h, 32), a.n), dim3(256), 0, (hipStream_t)stream, a);
  }
  return check_launch("frames_u8_to_f32_kernel");
}
