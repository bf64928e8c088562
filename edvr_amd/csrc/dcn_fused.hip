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
#include <type_traits>

#include "common.h"

namespace edvr {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

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
  constexpr int TH = 4, TW = 32, KK = 9, CK = 8, WCK = 4;  // x tile: chunks of 8 input channels; weight slab: sub-chunks of 4
  constexpr int IH = TH + 2 + 2 * R, IW = TW + 2 + 2 * R;  // R = 3: 12 x 40 halo tile; R = 7: 20 x 48
  constexpr int RS = IW, CHS = IH * RS;
  constexpr int MB = 32 * MT, WROWS = WCK * KK;            // 36 weight rows (channel, tap) of MB floats per sub-chunk
  constexpr int XS_ELEMS = (CK * CHS + 3) / 4 * 4, WS_ELEMS = WROWS * MB;
  // LDS: the x halo tile and the weight slab are BOTH double-buffered (R = 3, MT = 4: 2 x 15 KB + 2 x 18 KB = 66 KB, two
  // workgroups per CU) and BOTH filled by LDS-DMA (global_load_lds_dwordx4 / buffer_load_dword ... lds): no staging registers,
  // no ds_write pass.  (Round 1 staged the 36 KB weight slab of a chunk through 36 registers per thread; at 256 VGPRs the
  // compiler spilled exactly those, so the prefetch went through scratch.  Removing that - and the other changes of round 2
  // noted below - freed 23 VGPRs and all scratch but did NOT move the kernel's speed: DESIGN.md section 8 has the measurements.)
  // (the R = 7 halo with >= 96 output channels would need 89-98 KB that way: there the x tile keeps ONE buffer and is committed
  //  at the chunk boundary between two barriers, as in round 1)
  constexpr bool XDB = (2 * XS_ELEMS + 2 * WS_ELEMS + 3 * 9 * 128) * 4 <= 80 * 1024;
  constexpr int TP_ELEMS = 3 * KK * TH * TW;  // 27 offset / mask values of the NEXT deformable group for the 128 pixels: 13.5 KB
  __shared__ __attribute__((aligned(16))) float smem[(XDB ? 2 : 1) * XS_ELEMS + 2 * WS_ELEMS + TP_ELEMS];
  float *xs0 = smem, *ws0 = smem + (XDB ? 2 : 1) * XS_ELEMS, *tp = ws0 + 2 * WS_ELEMS;

  const int tid = threadIdx.x, lane = tid & 63, half = lane >> 5, j = lane & 31;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
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

  // ---- x halo tile -> LDS, asynchronously too (buffer_load_dword ... lds): lane q of the workgroup owns tile positions q and
  //      q + 256; a position outside the image carries the offset 0x80000000, which fails the buffer range check and lands in
  //      LDS as 0 - the zero padding the reference's per-corner bounds test (.cu:481-491) needs.  No staging registers at all.
  constexpr int NXK = (CHS + 255) / 256;
  constexpr int RSRC_FLAGS = 0x00020000;
  int xoff[NXK];
#pragma unroll
  for (int k = 0; k < NXK; ++k) {
    const int q = tid + k * 256;
    const int iy = q / RS, ix = q - iy * RS;
    const int gy = hy0 + iy, gx = wx0 + ix;
    xoff[k] = (gy >= 0 && gy < a.H && gx >= 0 && gx < a.W) ? (gy * a.W + gx) * 4 : (int)0x80000000;
  }
  const __amdgpu_buffer_rsrc_t x_rsrc = [&]() {
    const uint64_t pv = reinterpret_cast<uint64_t>(x);
    const uint64_t pu = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(pv >> 32)) << 32) |
                        (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)pv);
    return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void *>(pu), (short)0, a.C * P * 4, RSRC_FLAGS);
  }();
  auto dma_x = [&](float *dst, int c0) {
    typedef __attribute__((address_space(3))) void lvoid;
#pragma unroll
    for (int ch = 0; ch < CK; ++ch)
#pragma unroll
      for (int k = 0; k < NXK; ++k)
        if ((k + 1) * 256 <= CHS || tid + k * 256 < CHS)  // (the last pass covers part of the workgroup: those lanes are masked off)
          __builtin_amdgcn_raw_ptr_buffer_load_lds(x_rsrc, (lvoid *)(dst + ch * CHS + k * 256 + wave * 64), 4, xoff[k], (c0 + ch) * P * 4, 0, 0);
  };
  // ---- weight sub-chunk (channels cb .. cb + 3, all taps) -> LDS buffer `dst`, asynchronously.  One wave instruction moves
  //      64 x 16 B to a CONTIGUOUS 1 KB of LDS (wave-uniform base + lane x 16); the slab is dense row-major [row][MB], so lane q
  //      of instruction i carries float4 number i * 64 + q of the slab: row (i * 64 + q) / (MB / 4) of the packed weights.
  auto dma_w = [&](float *dst, int cb) {
    constexpr int Q = MB / 4, TOTAL = WROWS * Q;
    const float *src = a.wpk + (int64_t)cb * KK * a.cop + co_blk;
    for (int i = wave; i * 64 < TOTAL; i += 4) {
      const int q = i * 64 + lane;
      if (q < TOTAL) {
        const int row = q / Q, c4 = q - row * Q;
        typedef __attribute__((address_space(1))) void gvoid;
        typedef __attribute__((address_space(3))) void lvoid;
        __builtin_amdgcn_global_load_lds((gvoid *)(src + (int64_t)row * a.cop + c4 * 4), (lvoid *)(dst + i * 256), 16, 0, 0);
      }
    }
  };

  // ---- per-(pixel, group, tap) sampling state, refreshed when the chunk enters a new deformable group
  float w00[KK], w01[KK], w10[KK], w11[KK];
  int taddr[KK];  // (the slow path re-reads its offsets from global memory: 18 registers of sampling positions kept for a rare
                  //  branch were spilling state of the hot loop)
  unsigned slow = 0;      // bit t: the 2x2 cell of tap t is not fully inside the staged halo (this lane)
  unsigned slow_any = 0;  // the same for ANY lane of the wave, in a scalar register: the per-tap test in the hot loop is then one
                          // s_bitcmp + branch instead of a v_cmp / ballot / exec-mask sequence, and a whole half chunk can take the
                          // branch-free copy of the loop
  // The 27 offset / mask values of a group come from global memory; consumed right away, their latency (1-2 us) sat exposed
  // between two chunks every 16 channels (measured on the 64-channel layer, where registers allowed a register prefetch: +17 %).
  // They are fetched by LDS-DMA one whole group ahead - requested right after a group's first barrier, read from LDS when the
  // group ends - into `tp` [value 27][wave 4][pixel 32]; both half-waves of a pixel read the same word (broadcast).
  const int tp_voff = pix_ok ? p * 4 : (int)0x80000000;
  auto request_taps = [&](int g) {
    typedef __attribute__((address_space(3))) void lvoid;
    auto rsrc_of = [&](const float *ptr, int bytes) {  // (built here: eight scalar registers live for 27 instructions, not for the kernel)
      const uint64_t pv = reinterpret_cast<uint64_t>(ptr);
      const uint64_t pu = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(pv >> 32)) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)pv);
      return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void *>(pu), (short)0, bytes, RSRC_FLAGS);
    };
    const __amdgpu_buffer_rsrc_t off_rsrc = rsrc_of(off_b, a.dg * 18 * P * 4), msk_rsrc = rsrc_of(msk_b, a.dg * 9 * P * 4);
    if (half == 0) {  // lanes 0-31: one word per pixel
#pragma unroll
      for (int t = 0; t < KK; ++t) {
        __builtin_amdgcn_raw_ptr_buffer_load_lds(off_rsrc, (lvoid *)(tp + ((3 * t + 0) * 4 + wave) * 32), 4, tp_voff, (g * 18 + 2 * t) * P * 4, 0, 0);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(off_rsrc, (lvoid *)(tp + ((3 * t + 1) * 4 + wave) * 32), 4, tp_voff, (g * 18 + 2 * t + 1) * P * 4, 0, 0);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(msk_rsrc, (lvoid *)(tp + ((3 * t + 2) * 4 + wave) * 32), 4, tp_voff, (g * 9 + t) * P * 4, 0, 0);
      }
    }
  };
  auto finish_taps = [&]() {
#pragma unroll
    for (int t = 0; t < KK; ++t) {
      const float dy = tp[((3 * t + 0) * 4 + wave) * 32 + j], dx = tp[((3 * t + 1) * 4 + wave) * 32 + j], m = tp[((3 * t + 2) * 4 + wave) * 32 + j];
      const float h = (float)(oy - 1 + t / 3) + dy, w = (float)(ox - 1 + t % 3) + dx;
      const bool valid = pix_ok && h > -1.f && w > -1.f && h < (float)a.H && w < (float)a.W;  // .cu:618
      const float fh = floorf(h), fw = floorf(w);
      const float lh = h - fh, lw = w - fw;
      const float mm = valid ? m : 0.f;
      w00[t] = (1.f - lh) * (1.f - lw) * mm;
      w01[t] = (1.f - lh) * lw * mm;
      w10[t] = lh * (1.f - lw) * mm;
      w11[t] = lh * lw * mm;
      const int ry = (int)fh - hy0, rx = (int)fw - wx0;
      const bool inside = ry >= 0 && ry <= IH - 2 && rx >= 0 && rx <= IW - 2;
      taddr[t] = half * CHS + (inside ? ry * RS + rx : 0);
      if (valid && !inside) slow |= 1u << t; else slow &= ~(1u << t);
    }
    unsigned any = 0;
#pragma unroll
    for (int t = 0; t < KK; ++t) any |= __any(slow >> t & 1u) ? 1u << t : 0u;
    slow_any = __builtin_amdgcn_readfirstlane(any);
  };

  const int abase = half * KK * MB + j * MT;
  // One (channel pair, tap) step = 4 LDS gathers + one weight read -> 7 VALU -> MT MFMAs, 18 steps per half chunk.
  // SLOW = false: no tap of this wave's pixels leaves the halo for the current deformable group (slow_any == 0, decided once per
  // group): straight-line code, no exec-mask test between the MFMAs, and a three-stage software pipeline the scheduler is told
  // to interleave - while the MT MFMAs of step s issue (one every 64 cycles), the LDS reads of step s + 2 and the bilinear
  // arithmetic of step s + 1 go into the issue slots between them.  Measured: 84 TF/s at sigma(offset) = 0.3 px against 80 for
  // round 1's read-then-wait order; the ablation ceiling of this work split (no arithmetic at all) is 94.
  auto run_half = [&](auto SLOWT, const float *xs, const float *wsb, int c0, int cp0) {
    constexpr bool SLOW = decltype(SLOWT)::value;
    constexpr int NT = 2 * KK;  // 18 (channel pair, tap) steps
    float cv[3][4], av[3][MT];
    auto issue = [&](int step, float (&c)[4], float (&aw)[MT]) {
      const int q = step / KK, t = step % KK;  // q = channel pair inside this half (0 / 1)
      const float *cell = xs + taddr[t] + 2 * (cp0 + q) * CHS;
      c[0] = cell[0];
      c[1] = cell[1];
      c[2] = cell[RS];
      c[3] = cell[RS + 1];
      // weight slab rows are stored [j][m] (dcn_fused_pack_kernel): the MT weights of this lane's 32-channel tiles are adjacent,
      // one ds_read_b128 (MT = 4) / b64 (MT = 2) instead of MT ds_read_b32
      const float *ap = wsb + abase + (2 * q * KK + t) * MB;
      if constexpr (MT == 4) {
        const f32x4 v = *reinterpret_cast<const f32x4 *>(ap);
        aw[0] = v[0]; aw[1] = v[1]; aw[2] = v[2]; aw[3] = v[3];
      } else if constexpr (MT == 2) {
        const f32x2 v = *reinterpret_cast<const f32x2 *>(ap);
        aw[0] = v[0]; aw[1] = v[1];
      } else {
#pragma unroll
        for (int m = 0; m < MT; ++m) aw[m] = ap[m];
      }
    };
    auto sample = [&](int step, const float (&c)[4]) -> float {
      const int q = step / KK, t = step % KK;
      float bv = w00[t] * c[0] + w01[t] * c[1] + w10[t] * c[2] + w11[t] * c[3];
      if constexpr (SLOW) {
        if (slow_any >> t & 1u) {  // wave-uniform (SGPR): some lane's cell left the halo -> gather from global
          if (slow >> t & 1u) {
            const int g = c0 / cpg;
            const float m = msk_b[(int64_t)(g * 9 + t) * P + p];
            const float hsp = (float)(oy - 1 + t / 3) + off_b[(int64_t)(g * 18 + 2 * t) * P + p];
            const float wsp = (float)(ox - 1 + t % 3) + off_b[(int64_t)(g * 18 + 2 * t + 1) * P + p];
            bv = dcn_sample_global(x + (int64_t)(c0 + 2 * (cp0 + q) + half) * P, hsp, wsp, m, a.H, a.W);
          }
        }
      }
      return bv;
    };
    issue(0, cv[0], av[0]);
    issue(1, cv[1], av[1]);
    float bv = sample(0, cv[0]);
#pragma unroll
    for (int st = 0; st < NT; ++st) {
      const int cur = st % 3, n1 = (st + 1) % 3, n2 = (st + 2) % 3;
      if (st + 2 < NT) issue(st + 2, cv[n2], av[n2]);
      float bvn = 0.f;
      if (st + 1 < NT) bvn = sample(st + 1, cv[n1]);
#pragma unroll
      for (int m = 0; m < MT; ++m) acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[cur][m], bv, acc[m], 0, 0, 0);
      bv = bvn;
      if constexpr (!SLOW) {  // issue order inside this step: DS reads, then MFMAs with the VALU work spread between them
        __builtin_amdgcn_sched_group_barrier(0x100, 3, 0);  // DS read
#pragma unroll
        for (int m = 0; m < MT; ++m) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);  // one MFMA
          __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);  // up to three VALU
        }
      }
    }
  };
  // every wave waits for ITS OWN outstanding loads (the LDS-DMA pieces it issued, its x registers) before the barrier: past
  // the barrier everything that was in flight has landed
  auto settle = [&]() {
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __syncthreads();
  };

  // ---- schedule (chunk = 8 channels = weight sub-chunks A, B of 4):
  //   boundary : [settle]  A(c), x(c) valid; nobody reads B or the other x buffer any more -> DMA B(c), DMA x(c+8);  MFMAs on A(c)
  //   middle   : [settle]  B(c) valid; nobody reads A any more                             -> DMA A(c+8);            MFMAs on B(c)
  dma_w(ws0, 0);
  dma_x(xs0, 0);
  request_taps(0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (each wave reads back only what it requested itself)
  finish_taps();
  for (int c0 = 0; c0 < a.C; c0 += CK) {
    const bool more = (c0 + CK) < a.C;
    const int xb = XDB ? (c0 / CK) & 1 : 0;
    const float *xs = xs0 + xb * XS_ELEMS;
    settle();
    if (!XDB && c0 > 0) {  // single x buffer (big halo, many output channels): the tile is fetched here, latency exposed
      dma_x(xs0, c0);
      settle();
    }
    dma_w(ws0 + WS_ELEMS, c0 + WCK);
    if (XDB && more) dma_x(xs0 + (xb ^ 1) * XS_ELEMS, c0 + CK);
    if (c0 % cpg == 0 && c0 + cpg < a.C) request_taps(c0 / cpg + 1);  // first chunk of a group: the next group's 27 values
    if (slow_any) run_half(std::true_type{}, xs, ws0, c0, 0); else run_half(std::false_type{}, xs, ws0, c0, 0);
    settle();
    if (more) dma_w(ws0, c0 + CK);
    if (slow_any) run_half(std::true_type{}, xs, ws0 + WS_ELEMS, c0, 2); else run_half(std::false_type{}, xs, ws0 + WS_ELEMS, c0, 2);
    if (more && (c0 + CK) % cpg == 0) finish_taps();  // (requested a group ago; landed before the middle barrier of its first chunk)
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

// Weight layout of the fused kernel: wpk[c][tap][co'] with the output channels of each launch block (128 = 4 tiles of 32, or the
// 1-3 tiles of the tail launch) reordered [j][m]: co = block_start + m * 32 + j  ->  position block_start + j * MT + m.
__global__ void dcn_fused_pack_kernel(const float *__restrict__ w, float *__restrict__ wpk, int Co, int C, int cop) {
  const int64_t total = (int64_t)C * 9 * cop;
  const int full = Co / 128, rem_tiles = (Co - full * 128 + 31) / 32;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int pos = (int)(i % cop), t = (int)((i / cop) % 9), c = (int)(i / ((int64_t)cop * 9));
    const int blk = pos / 128 < full ? pos / 128 : full, start = blk * 128, mt = blk < full ? 4 : rem_tiles;
    const int r = pos - start, j = r / mt, m = r - j * mt, co = start + m * 32 + j;
    wpk[i] = (mt > 0 && j < 32 && co < Co) ? w[((int64_t)co * C + c) * 9 + t] : 0.f;
  }
}

int dcn_fused_pack(const float *weight, float *wpk, int Co, int C, hipStream_t stream) {
  const int cop = (Co + 31) / 32 * 32;
  const int64_t total = (int64_t)C * 9 * cop;
  hipLaunchKernelGGL(dcn_fused_pack_kernel, dim3((unsigned)std::min<int64_t>(cdiv64(total, 256), 2048)), dim3(256), 0, stream, weight, wpk, Co, C, cop);
  return check_launch("dcn_fused_pack_kernel");
}

bool dcn_fused_supported(int C, int Co, int H, int W, int kh, int kw, int stride, int pad, int dil, int groups, int dg) {
  if (!(kh == 3 && kw == 3 && stride == 1 && pad == 1 && dil == 1 && groups == 1 && C % dg == 0 && (C / dg) % 8 == 0 && Co > 0)) return false;
  // x, the offsets and the masks of one image are addressed through 32-bit buffer offsets (range-checked resources: a byte count
  // of 2^31 or more would wrap and gather wrong data silently); larger images take the generic column-buffer path (64-bit pointers)
  const int64_t P = (int64_t)H * W, lim = (int64_t)1 << 31;
  return (int64_t)C * P * 4 < lim && (int64_t)dg * 18 * P * 4 < lim && (int64_t)C * 9 * ((Co + 31) / 32 * 32) * 4 < lim;
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
