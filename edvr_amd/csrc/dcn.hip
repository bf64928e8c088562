// dcn.hip - modulated deformable convolution (DCNv2) forward / backward for gfx950.
//
// Replaces (paths relative to the reference's basicsr/models/ops/dcn/src/):
//   modulated_deform_conv_cuda_forward   deform_conv_cuda.cpp:490-569
//   modulated_deform_conv_cuda_backward  deform_conv_cuda.cpp:571-685
//   modulated_deformable_im2col / col2im / col2im_coord kernels
//                                        deform_conv_cuda_kernel.cu:570-767
// Semantics reproduced exactly (see oracle/dcnv2_oracle_impl.inc for the CPU
// restatement these kernels are tested against):
//   * offset channel = g*2K + 2k + {0: dy, 1: dx}, mask channel = g*K + k, column row = c*K + k
//   * a tap contributes iff -1 < h < H and -1 < w < W (.cu:618); every bilinear corner
//     outside [0,H-1]x[0,W-1] contributes 0 (.cu:481-491)
//   * d/d(offset) is the derivative inside the cell [floor p, floor p + 1) (.cu:526-568)
//
// Differences in HOW (MI355X-first):
//   * the whole batch is processed by one launch per stage (the reference loops over the
//     batch on the host: 2*B launches + an at::zeros per call, cpp:532-543);
//   * one thread owns one (image, deformable group, tap, pixel): offsets/mask are read and the
//     bilinear cell is resolved ONCE and reused for the C/dg channels of the group (the reference
//     re-reads them per channel), lanes run along the pixel axis so every column/offset access
//     is a coalesced 256-B line;
//   * forward: the EDVR signature (3x3, stride 1, pad 1, groups 1) never builds columns - dcn_fused.hip samples straight into
//     the MFMA B operand; other signatures build columns here and run the column x weight product on the fp32 MFMA conv
//     kernel of conv2d.hip (bias fused);
//   * backward: one fused kernel produces dOffset, dMask, dX and rewrites the dY-columns buffer in place with the forward
//     columns needed by dW (the reference runs three kernels and a 25-iteration window scan per column entry, .cu:677-691).
//     dX is scattered either with device atomics or through a per-tile LDS window (scatter_hint), right-hand bilinear
//     corners merged into the neighbouring lane by a DPP shift first.  dcol = W^T dY is a 1x1 convolution of dY (conv1x1.hip),
//     dW = sum dY col^T the split-K MFMA GEMM below (deterministic sum over images and pixels); DCNv1 entry points at the end
//     of the file reuse all of it.
#include <cstdlib>

#include "common.h"
#include "dcn_tap.h"

namespace edvr {

typedef float f32x16 __attribute__((ext_vector_type(16)));

__global__ void fill_kernel(float *__restrict__ p, float v, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) p[i] = v;
}

// In a smooth offset field the right-hand corners (01, 11) of pixel p are the left-hand corners (00, 10) of pixel p+1,
// i.e. of the next lane.  merge_right() decides, once per (pixel, tap), whether this lane hands its 01 / 11 contributions
// to the next lane (which adds them to its own 00 / 10 before its atomic): half the atomics where the field is smooth.
struct Merge {
  bool give01, give11;  // this lane's 01 / 11 go to lane+1
  bool take00, take10;  // lane-1 gives its 01 / 11 to this lane's 00 / 10
};
__device__ __forceinline__ Merge merge_right(const Tap &t, bool same_plane_as_next) {
  Merge mg;
  const int lane = __lane_id();
  const int n_o00 = lane_next_i(t.o00), n_o10 = lane_next_i(t.o10);
  const int n_ok = lane_next_i((t.ok00 ? 1 : 0) | (t.ok10 ? 2 : 0));
  const bool has_next = lane < 63 && same_plane_as_next;
  mg.give01 = has_next && t.ok01 && (n_ok & 1) && n_o00 == t.o01;
  mg.give11 = has_next && t.ok11 && (n_ok & 2) && n_o10 == t.o11;
  const int p_give = lane_prev_i((mg.give01 ? 1 : 0) | (mg.give11 ? 2 : 0));
  mg.take00 = (p_give & 1) != 0;
  mg.take10 = (p_give & 2) != 0;
  return mg;
}

// ---------------------------------------------------------------------------------------------
// forward gather: col[img, c*K + k, p] = mask * bilinear(x[img, c], p + tap + offset)
__global__ __launch_bounds__(256) void dcn_im2col_kernel(const float *__restrict__ x, const float *__restrict__ offset,
                                                         const float *__restrict__ mask, float *__restrict__ col,
                                                         const DcnShape s) {
  const int K = s.kh * s.kw, P = s.Ho * s.Wo, cpg = s.C / s.dg;
  const int64_t total = (int64_t)s.B * s.dg * K * P;
  for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
    const int p = (int)(idx % P);
    const int k = (int)((idx / P) % K);
    const int g = (int)((idx / ((int64_t)P * K)) % s.dg);
    const int b = (int)(idx / ((int64_t)P * K * s.dg));
    const int ho = p / s.Wo, wo = p - ho * s.Wo;
    const int i = k / s.kw, j = k - i * s.kw;
    const float *off_b = offset + (int64_t)b * s.off_bs + (int64_t)(g * 2 * K + 2 * k) * P + p;
    const float dy = off_b[0], dx = off_b[P];
    const float m = mask[(int64_t)b * s.msk_bs + (int64_t)(g * K + k) * P + p];
    const Tap t = resolve_tap((float)(ho * s.stride - s.pad + i * s.dil) + dy, (float)(wo * s.stride_w - s.pad_w + j * s.dil_w) + dx, s.H, s.W);
    const float *xp = x + ((int64_t)b * s.C + (int64_t)g * cpg) * s.H * s.W;
    float *cp = col + ((int64_t)b * s.C * K + (int64_t)(g * cpg) * K + k) * P + p;
    const int64_t plane = (int64_t)s.H * s.W;
    for (int cc = 0; cc < cpg; ++cc) {
      const float v = t.w00 * xp[t.o00] + t.w01 * xp[t.o01] + t.w10 * xp[t.o10] + t.w11 * xp[t.o11];
      *cp = v * m;
      xp += plane;
      cp += (int64_t)K * P;
    }
  }
}

// ---------------------------------------------------------------------------------------------
// backward: per (img, g, k, p) reduce over the group's channels.
//   dcol (in)  : W^T dY, layout (img, c*K + k, p);  rewritten in place with the forward column
//   dmask, doffset written; dx accumulated with fp32 atomics (dx pre-zeroed by the driver)
template <bool WITH_DX>
__global__ __launch_bounds__(256) void dcn_bwd_coord_kernel(const float *__restrict__ x, const float *__restrict__ offset,
                                                            const float *__restrict__ mask, float *__restrict__ dcol,
                                                            float *__restrict__ dx, float *__restrict__ doffset,
                                                            float *__restrict__ dmask, const DcnShape s) {
  const int K = s.kh * s.kw, P = s.Ho * s.Wo, cpg = s.C / s.dg;
  const int64_t total = (int64_t)s.B * s.dg * K * P;
  for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
    const int p = (int)(idx % P);
    const int k = (int)((idx / P) % K);
    const int g = (int)((idx / ((int64_t)P * K)) % s.dg);
    const int b = (int)(idx / ((int64_t)P * K * s.dg));
    const int ho = p / s.Wo, wo = p - ho * s.Wo;
    const int i = k / s.kw, j = k - i * s.kw;
    const float *off_b = offset + (int64_t)b * s.off_bs + (int64_t)(g * 2 * K + 2 * k) * P + p;
    const float dy = off_b[0], dxo = off_b[P];
    const float m = mask[(int64_t)b * s.msk_bs + (int64_t)(g * K + k) * P + p];
    const Tap t = resolve_tap((float)(ho * s.stride - s.pad + i * s.dil) + dy, (float)(wo * s.stride_w - s.pad_w + j * s.dil_w) + dxo, s.H, s.W);
    // d(bilinear)/dh and /dw weights per corner (zero where the corner is outside or the tap invalid)
    const float hh = 1.f - t.lh, hw = 1.f - t.lw;
    const bool ok00 = t.ok00, ok01 = t.ok01, ok10 = t.ok10, ok11 = t.ok11;
    const float gy00 = ok00 ? -hw : 0.f, gy01 = ok01 ? -t.lw : 0.f, gy10 = ok10 ? hw : 0.f, gy11 = ok11 ? t.lw : 0.f;
    const float gx00 = ok00 ? -hh : 0.f, gx01 = ok01 ? hh : 0.f, gx10 = ok10 ? -t.lh : 0.f, gx11 = ok11 ? t.lh : 0.f;

    const int64_t plane = (int64_t)s.H * s.W;
    const float *xp = x + ((int64_t)b * s.C + (int64_t)g * cpg) * plane;
    float *gp = WITH_DX ? dx + ((int64_t)b * s.C + (int64_t)g * cpg) * plane : nullptr;
    float *cp = dcol + ((int64_t)b * s.C * K + (int64_t)(g * cpg) * K + k) * P + p;
    float s_m = 0.f, s_y = 0.f, s_x = 0.f;
    // lane+1 is pixel p+1 of the same (image, group, tap) unless this lane is the last pixel of the plane
    // (all 64 lanes of a wave are live here or the wave is the grid's tail: idx + 1 < total covers it)
    Merge mg = {};
    if (WITH_DX) mg = merge_right(t, p + 1 < P && idx + 1 < total);
    if (!WITH_DX) {
      // no scatter: channels in batches of 4 with all loads first (the column is rewritten in place, so the compiler cannot
      // hoist the loads of channel c + 1 above the store of channel c: one dependent memory round trip per channel otherwise)
      for (int cc0 = 0; cc0 < cpg; cc0 += 4) {
        float dc[4], a00[4], a01[4], a10[4], a11[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int cu = cc0 + u < cpg ? u : 0;
          dc[u] = cp[(int64_t)cu * K * P];
          a00[u] = xp[cu * plane + t.o00];
          a01[u] = xp[cu * plane + t.o01];
          a10[u] = xp[cu * plane + t.o10];
          a11[u] = xp[cu * plane + t.o11];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          if (cc0 + u < cpg) {
            const float val = t.w00 * a00[u] + t.w01 * a01[u] + t.w10 * a10[u] + t.w11 * a11[u];
            s_m += dc[u] * val;
            s_y += dc[u] * (gy00 * a00[u] + gy01 * a01[u] + gy10 * a10[u] + gy11 * a11[u]);
            s_x += dc[u] * (gx00 * a00[u] + gx01 * a01[u] + gx10 * a10[u] + gx11 * a11[u]);
            cp[(int64_t)u * K * P] = val * m;
          }
        }
        xp += 4 * plane;
        cp += (int64_t)4 * K * P;
      }
    }
    for (int cc = 0; WITH_DX && cc < cpg; ++cc) {
      const float dc = *cp;
      const float a00 = xp[t.o00], a01 = xp[t.o01], a10 = xp[t.o10], a11 = xp[t.o11];
      const float val = t.w00 * a00 + t.w01 * a01 + t.w10 * a10 + t.w11 * a11;
      s_m += dc * val;
      s_y += dc * (gy00 * a00 + gy01 * a01 + gy10 * a10 + gy11 * a11);
      s_x += dc * (gx00 * a00 + gx01 * a01 + gx10 * a10 + gx11 * a11);
      if (WITH_DX) {  // (false: dX comes from dcn_bwd_dx_strip_kernel)
        const float tt = dc * m;
        const float v01 = t.w01 * tt, v11 = t.w11 * tt;
        const float l01 = lane_prev_f(v01), l11 = lane_prev_f(v11);  // uniform control flow: every lane executes the shifts
        const float v00 = t.w00 * tt + (mg.take00 ? l01 : 0.f), v10 = t.w10 * tt + (mg.take10 ? l11 : 0.f);
        if (ok00 || mg.take00) unsafeAtomicAdd(gp + t.o00, v00);
        if (ok01 && !mg.give01) unsafeAtomicAdd(gp + t.o01, v01);
        if (ok10 || mg.take10) unsafeAtomicAdd(gp + t.o10, v10);
        if (ok11 && !mg.give11) unsafeAtomicAdd(gp + t.o11, v11);
      }
      *cp = val * m;  // forward column, consumed by the dW GEMM
      xp += plane;
      gp += plane;
      cp += (int64_t)K * P;
    }
    dmask[(int64_t)b * s.dmsk_bs + (int64_t)(g * K + k) * P + p] = s_m;
    float *dob = doffset + (int64_t)b * s.doff_bs + (int64_t)(g * 2 * K + 2 * k) * P + p;
    dob[0] = s_y * m;
    dob[P] = s_x * m;
  }
}

// ---------------------------------------------------------------------------------------------
// dX of the backward WITHOUT scatter atomics, for the EDVR signature (3x3, stride 1, pad 1, dil 1).  The scatter above costs 2-4 atomics per (pixel, tap, channel) - 1.7 G per launch on the 64x64 training layer, the
// L2 atomic rate - although every dX element is the sum of only ~36 contributions from its 5x5 neighbourhood.  Here ONE wave
// owns a 64-column strip of whole channel planes (lane = column; one strip at training-patch widths) and walks the rows top to bottom:
//   * a tap whose offset is sub-pixel (floor(offset) in {-1, 0}) lands on a 2x2 block inside the 3x3 cells around its regular
//     position; its bilinear weights factor into 3 row x 3 column weights (2 non-zero each, selected once per (pixel, tap)), so
//     the 9 taps of a pixel accumulate into a 5x5 register patch with STATIC indices: out[i + a][j + b] += dc * ry[a] * cx[b];
//   * the patch is folded onto the owner lanes with DPP wave shifts (cell x + s comes from lane x - s): 4 shifts per row;
//   * the five patch rows go into a register ring acc[5]; after row y the ring slot of row y - 2 is complete and leaves with one
//     uncontended atomic per element (84 M per launch instead of 1.7 G; atomic because the rare path below may hit any cell);
//   * taps with larger offsets take the reference's route: four device atomics per channel, per lane, branch-divergent.
//   * wider images: the <= 2 patch columns that cross a strip edge go to their cells directly (4 lanes of 64).
// No inter-wave communication: all contributions to a strip of a channel plane are produced by the wave that owns it.
__global__ __launch_bounds__(256, 4) void dcn_bwd_dx_strip_kernel(const float *__restrict__ offset, const float *__restrict__ mask,
                                                               const float *__restrict__ dcol, float *__restrict__ dx,
                                                               const DcnShape s) {
  // CQ channels share one evaluation of the tap weights (which is 2/3 of the instructions of a one-channel pass).  Their row
  // rings live in LDS (slot = row % 5, one private column per lane: plain read-add-write, no conflicts, no atomics) so that
  // the channel loop can stay a real loop: unrolled with the rings in registers, hipcc interleaves the 25-register patches of
  // all channels and spills 208 (CQ = 2) to 628 (CQ = 4) bytes per lane.  99 VGPRs, 5 waves per SIMD.
  constexpr int K = 9, CQ = 4;
  __shared__ float ring[4][CQ][5][64];
  constexpr int RSRC_FLAGS = 0x00020000, OOB = (int)0x80000000;
  const int g = blockIdx.x % s.dg, strip = blockIdx.x / s.dg, strips = gridDim.x / s.dg, b = blockIdx.y;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int H = s.H, W = s.W, P = H * W, cpg = s.C / s.dg;
  const int x = strip * 64 + lane;  // image column of this lane; images wider than 64 columns are cut into strips
  const bool live = x < W;
  // Loads go through buffer resources (wave-uniform base in SGPRs + 32-bit per-lane offset + scalar plane offset): no 64-bit
  // address arithmetic per lane, and out-of-range reads return 0 - dead lanes (x >= W) and channels past the group's last
  // need no branches.
  auto rsrc_of = [&](const float *ptr, int bytes) {
    const uint64_t pv = reinterpret_cast<uint64_t>(ptr);
    const uint64_t pu = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(pv >> 32)) << 32) |
                        (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)pv);
    return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void *>(pu), (short)0, bytes, RSRC_FLAGS);
  };
  auto ld = [&](__amdgpu_buffer_rsrc_t r, int voff, int soff) {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, 0));
  };
  const float *off_g = offset + (int64_t)b * s.off_bs + (int64_t)(g * 2 * K) * P;
  const float *msk_g = mask + (int64_t)b * s.msk_bs + (int64_t)(g * K) * P;
  const __amdgpu_buffer_rsrc_t r_off = rsrc_of(off_g, 2 * K * P * 4), r_msk = rsrc_of(msk_g, K * P * 4);
  const int plane_b = P * 4;
  for (int cq = wave; cq * CQ < cpg; cq += 4) {
    const int c0 = g * cpg + cq * CQ;  // first channel of this pass
    const int nch = min(CQ, cpg - cq * CQ);
    const float *dc_base = dcol + ((int64_t)b * s.C + c0) * K * P;
    const __amdgpu_buffer_rsrc_t r_dc = rsrc_of(dc_base, nch * K * P * 4);
    float *dx_base = dx + ((int64_t)b * s.C + c0) * P;
    float(*acc)[5][64] = ring[wave];  // acc[c][row % 5][lane]: rows y - 2 .. y + 2 of channel c0 + c while row y is processed
#pragma unroll
    for (int c = 0; c < CQ; ++c)
#pragma unroll
      for (int r = 0; r < 5; ++r) acc[c][r][lane] = 0.f;
    for (int y = 0; y < H; ++y) {
      const int p = y * W + x;
      const int voff = live ? p * 4 : OOB;
      // ---- per (pixel, tap): mask-scaled row weights and column weights on the 3x3 cells around the regular tap position
      float rym[K][3], cx[K][3];
      unsigned far = 0;  // taps outside the sub-pixel window: slow path
#pragma unroll
      for (int k = 0; k < K; ++k) {
        const int i = k / 3, j = k - 3 * i;
        const float dyo = ld(r_off, voff, (2 * k) * plane_b), dxo = ld(r_off, voff, (2 * k + 1) * plane_b);
        const float m = ld(r_msk, voff, k * plane_b);  // 0 for dead lanes: all their weights vanish
        const float ph = (float)(y - 1 + i) + dyo, pw = (float)(x - 1 + j) + dxo;
        const bool valid = (ph > -1.f) && (pw > -1.f) && (ph < (float)H) && (pw < (float)W);
        const float fh = floorf(ph), fw = floorf(pw);
        const int h0 = (int)fh, w0 = (int)fw;
        const float lh = ph - fh, lw = pw - fw;
        const int fy = h0 - (y - 1 + i), fx = w0 - (x - 1 + j);  // -1 or 0 for a sub-pixel offset
        const bool near = (fy == -1 || fy == 0) && (fx == -1 || fx == 0);
        const float wt = (valid && near && h0 >= 0) ? m * (1.f - lh) : 0.f, wb = (valid && near && h0 + 1 <= H - 1) ? m * lh : 0.f;
        const float wl = w0 >= 0 ? 1.f - lw : 0.f, wr = w0 + 1 <= W - 1 ? lw : 0.f;
        rym[k][0] = fy == -1 ? wt : 0.f;
        rym[k][1] = fy == -1 ? wb : wt;
        rym[k][2] = fy == -1 ? 0.f : wb;
        cx[k][0] = fx == -1 ? wl : 0.f;
        cx[k][1] = fx == -1 ? wr : wl;
        cx[k][2] = fx == -1 ? 0.f : wr;
        if (valid && !near && m != 0.f) far |= 1u << k;
      }
      int slot[5];  // ring slot of row y + r - 2
#pragma unroll
      for (int r = 0; r < 5; ++r) slot[r] = (y + r + 3) % 5;
#pragma unroll 1
      for (int c = 0; c < nch; ++c) {
        float dc[K];
#pragma unroll
        for (int k = 0; k < K; ++k) dc[k] = ld(r_dc, voff, (c * K + k) * plane_b);  // 0 past the group's last channel
        float out[5][5];
#pragma unroll
        for (int r = 0; r < 5; ++r)
#pragma unroll
          for (int q = 0; q < 5; ++q) out[r][q] = 0.f;
#pragma unroll
        for (int k = 0; k < K; ++k) {
          const int i = k / 3, j = k - 3 * i;
#pragma unroll
          for (int a = 0; a < 3; ++a) {
            const float ra = dc[k] * rym[k][a];
#pragma unroll
            for (int q = 0; q < 3; ++q) out[i + a][j + q] += ra * cx[k][q];
          }
        }
#pragma unroll
        for (int r = 0; r < 5; ++r)  // out[r][q]: contribution of pixel x to cell x + q - 2 -> owner lanes
          acc[c][slot[r]][lane] += out[r][2] + lane_prev_f(out[r][3] + lane_prev_f(out[r][4])) + lane_next_f(out[r][1] + lane_next_f(out[r][0]));
        if (strips > 1 && live && (lane < 2 || lane > 61)) {
          // strip edges: patch columns that belong to a neighbouring strip fall off the wave in the shifts above; they go to
          // their cells directly (lanes 0, 1, 62, 63 only: <= 3 cells x 5 rows each)
          float *gp = dx_base + (int64_t)c * P;
#pragma unroll
          for (int q = 0; q < 5; ++q) {
            if (q == 2) continue;
            const int tl = lane + q - 2, tx = x + q - 2;
            if ((tl < 0 || tl > 63) && tx >= 0 && tx < W) {
#pragma unroll
              for (int r = 0; r < 5; ++r) {
                const int ty = y + r - 2;
                if (ty >= 0 && ty < H && out[r][q] != 0.f) unsafeAtomicAdd(gp + ty * W + tx, out[r][q]);
              }
            }
          }
        }
        if (far) {  // rare: the reference's scatter for the taps that left the window
          float *gp = dx_base + (int64_t)c * P;
          for (unsigned rest = far; rest; rest &= rest - 1) {
            const int k = __builtin_ctz(rest), i = k / 3, j = k - 3 * i;
            const float dyo = off_g[(int64_t)(2 * k) * P + p], dxo = off_g[(int64_t)(2 * k + 1) * P + p];
            const Tap t = resolve_tap((float)(y - 1 + i) + dyo, (float)(x - 1 + j) + dxo, H, W);
            const float tt = dc_base[((int64_t)c * K + k) * P + p] * msk_g[(int64_t)k * P + p];
            if (t.ok00) unsafeAtomicAdd(gp + t.o00, t.w00 * tt);
            if (t.ok01) unsafeAtomicAdd(gp + t.o01, t.w01 * tt);
            if (t.ok10) unsafeAtomicAdd(gp + t.o10, t.w10 * tt);
            if (t.ok11) unsafeAtomicAdd(gp + t.o11, t.w11 * tt);
          }
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      // ---- row y - 2 is complete: it leaves, and its slot becomes row y + 3
#pragma unroll 1
      for (int c = 0; c < nch; ++c) {
        const float v = acc[c][slot[0]][lane];
        acc[c][slot[0]][lane] = 0.f;
        if (live && y >= 2) unsafeAtomicAdd(dx_base + (int64_t)c * P + (y - 2) * W + x, v);
      }
    }
    // ---- rows H - 2 and H - 1; the slots of rows >= H hold zeros (corners below the image have zero weight)
    for (int c = 0; c < nch; ++c)
#pragma unroll
      for (int r = 0; r < 2; ++r) {
        const int row = H - 2 + r;
        if (live && row >= 0) unsafeAtomicAdd(dx_base + (int64_t)c * P + row * W + x, acc[c][(row + 5) % 5][lane]);
      }
  }
}

// Float accumulation into LDS.  The hardware's ds_add_f32 runs at 0.4 lane-operations per cycle and CU on gfx950 - 203 G/s device-wide, where
// ds_add_u32 does 9137 G/s and a plain read-add-write 4877 (scripts/micro/lds_atomics.hip, profiles/r6/micro_lds_atomics.log): 5.4 of the
// 12.0 ms of the LDS-window backward on the EDVR-L training layer were this one instruction.  A compare-and-swap loop on the integer pipe
// does the same addition at 4.7 lane-operations per cycle and CU (2.6 when lane pairs share a cell): 12x faster; in the kernel 12.0 -> 8.8 ms
// per call (the path without any scatter: 6.5).  Batching the sixteen loops of a channel batch (all reads, all swaps, retries) was
// measured too: 9.5 ms - more registers, no gain.  (Bit patterns are compared, so a NaN cannot spin the loop.)
__device__ __forceinline__ void lds_add_f32(float *p, float v) {
  unsigned *q = reinterpret_cast<unsigned *>(p);
  unsigned old = *q, assumed;
  do {
    assumed = old;
    old = atomicCAS(q, assumed, __builtin_bit_cast(unsigned, __builtin_bit_cast(float, assumed) + v));
  } while (old != assumed);
}

// ---------------------------------------------------------------------------------------------
// backward, tile version for the EDVR signature (3x3, stride 1, pad 1, dil 1, <= 16 channels per deformable group):
// same arithmetic as dcn_bwd_coord_kernel, but dX is accumulated in LDS.  The grid-stride kernel above issues four
// device-scope fp32 atomics per (pixel, tap, channel) - 9 GB of fabric write traffic per launch by the PMC counters, and the
// reason it ran at ~5600 cycles per (tap, channel) iteration.  Here a workgroup owns (image, group, 8 x 32 output pixels):
// its corner contributions go to a (8 + 2 + 2R) x (32 + 2 + 2R) x 16-channel LDS window (lds_add_f32 above), and the window is
// flushed once with one global atomic per touched element (windows of neighbouring tiles overlap): ~14x fewer global
// atomics.  Taps that leave the window (|offset| > R) fall back to the global atomic, per corner.
template <int R>
__global__ __launch_bounds__(256) void dcn_bwd_coord_tile_kernel(const float *__restrict__ x, const float *__restrict__ offset,
                                                                 const float *__restrict__ mask, float *__restrict__ dcol,
                                                                 float *__restrict__ dx, float *__restrict__ doffset,
                                                                 float *__restrict__ dmask, const DcnShape s, int tiles_x) {
  constexpr int TH = 8, TW = 32, LH = TH + 2 + 2 * R, LW = TW + 2 + 2 * R, LWP = LW + 1, MAXC = 16, K = 9;
  __shared__ float win[MAXC * LH * LWP];
  const int tid = threadIdx.x;
  int tile, g, b;
  xcd_block_index(tile, g, b);  // neighbouring tiles (overlapping windows, shared x lines) on one XCD
  const int ty0 = (tile / tiles_x) * TH, tx0 = (tile % tiles_x) * TW;
  const int wy0 = ty0 - 1 - R, wx0 = tx0 - 1 - R;  // image coordinates of window element (0, 0)
  const int P = s.Ho * s.Wo, cpg = s.C / s.dg;
  for (int i = tid; i < MAXC * LH * LWP; i += 256) win[i] = 0.f;
  __syncthreads();

  const int ho = ty0 + (tid >> 5), wo = tx0 + (tid & 31);
  const bool live = ho < s.Ho && wo < s.Wo;
  const int p = live ? ho * s.Wo + wo : 0;
  const int64_t plane = (int64_t)s.H * s.W;
  const float *xg = x + ((int64_t)b * s.C + (int64_t)g * cpg) * plane;
  float *gg = dx + ((int64_t)b * s.C + (int64_t)g * cpg) * plane;
  if (live) {
    for (int k = 0; k < K; ++k) {
      const int i = k / 3, j = k - 3 * i;
      const float *off_b = offset + (int64_t)b * s.off_bs + (int64_t)(g * 2 * K + 2 * k) * P + p;
      const float dy = off_b[0], dxo = off_b[P];
      const float m = mask[(int64_t)b * s.msk_bs + (int64_t)(g * K + k) * P + p];
      const Tap t = resolve_tap((float)(ho - 1 + i) + dy, (float)(wo - 1 + j) + dxo, s.H, s.W);
      const float hh = 1.f - t.lh, hw = 1.f - t.lw;
      const bool ok00 = t.ok00, ok01 = t.ok01, ok10 = t.ok10, ok11 = t.ok11;
      const float gy00 = ok00 ? -hw : 0.f, gy01 = ok01 ? -t.lw : 0.f, gy10 = ok10 ? hw : 0.f, gy11 = ok11 ? t.lw : 0.f;
      const float gx00 = ok00 ? -hh : 0.f, gx01 = ok01 ? hh : 0.f, gx10 = ok10 ? -t.lh : 0.f, gx11 = ok11 ? t.lh : 0.f;
      const int ly = t.h0 - wy0, lx = t.w0 - wx0;
      const bool inwin = ly >= 0 && ly + 1 < LH && lx >= 0 && lx + 1 < LW;
      float *w00 = win + ly * LWP + lx;  // only dereferenced when inwin
      const float *xp = xg;
      float *gp = gg;
      float *cp = dcol + ((int64_t)b * s.C * K + (int64_t)(g * cpg) * K + k) * P + p;
      float s_m = 0.f, s_y = 0.f, s_x = 0.f;
      const Merge mg = merge_right(t, true);  // lane+1 = next pixel of the row (addresses only match inside one plane)
      // Channels in batches of 4 with all loads first: the column is rewritten in place, so the compiler cannot hoist the
      // loads of channel c+1 above the store of channel c and every iteration paid a full dependent memory round trip.
      for (int cc0 = 0; cc0 < cpg; cc0 += 4) {
        float dc[4], a00[4], a01[4], a10[4], a11[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int cu = cc0 + u < cpg ? u : 0;  // (cpg is a multiple of 4 for every EDVR layer; clamp keeps the loads in range)
          dc[u] = cp[(int64_t)cu * K * P];
          a00[u] = xp[cu * plane + t.o00];
          a01[u] = xp[cu * plane + t.o01];
          a10[u] = xp[cu * plane + t.o10];
          a11[u] = xp[cu * plane + t.o11];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          if (cc0 + u < cpg) {
            const float val = t.w00 * a00[u] + t.w01 * a01[u] + t.w10 * a10[u] + t.w11 * a11[u];
            s_m += dc[u] * val;
            s_y += dc[u] * (gy00 * a00[u] + gy01 * a01[u] + gy10 * a10[u] + gy11 * a11[u]);
            s_x += dc[u] * (gx00 * a00[u] + gx01 * a01[u] + gx10 * a10[u] + gx11 * a11[u]);
            const float tt = dc[u] * m;
            const float v01 = t.w01 * tt, v11 = t.w11 * tt;
            const float l01 = lane_prev_f(v01), l11 = lane_prev_f(v11);
            const float v00 = t.w00 * tt + (mg.take00 ? l01 : 0.f), v10 = t.w10 * tt + (mg.take10 ? l11 : 0.f);
            if (inwin) {
              float *wc = w00 + (cc0 + u) * (LH * LWP);
              if (ok00) lds_add_f32(wc, v00);
              if (ok01 && !mg.give01) lds_add_f32(wc + 1, v01);
              if (ok10) lds_add_f32(wc + LWP, v10);
              if (ok11 && !mg.give11) lds_add_f32(wc + LWP + 1, v11);
            } else {
              float *gq = gp + u * plane;
              if (ok00) unsafeAtomicAdd(gq + t.o00, v00);
              if (ok01 && !mg.give01) unsafeAtomicAdd(gq + t.o01, v01);
              if (ok10) unsafeAtomicAdd(gq + t.o10, v10);
              if (ok11 && !mg.give11) unsafeAtomicAdd(gq + t.o11, v11);
            }
            cp[(int64_t)u * K * P] = val * m;  // forward column, consumed by the dW GEMM
          }
        }
        xp += 4 * plane;
        gp += 4 * plane;
        cp += (int64_t)4 * K * P;
      }
      dmask[(int64_t)b * s.dmsk_bs + (int64_t)(g * K + k) * P + p] = s_m;
      float *dob = doffset + (int64_t)b * s.doff_bs + (int64_t)(g * 2 * K + 2 * k) * P + p;
      dob[0] = s_y * m;
      dob[P] = s_x * m;
    }
  }
  __syncthreads();
  // flush the window: one global atomic per touched element inside the image
  for (int i = tid; i < cpg * LH * LW; i += 256) {
    const int cc = i / (LH * LW), rem = i - cc * (LH * LW), ly = rem / LW, lx = rem - ly * LW;
    const float v = win[cc * (LH * LWP) + ly * LWP + lx];
    const int gy = wy0 + ly, gx = wx0 + lx;
    if (v != 0.f && gy >= 0 && gy < s.H && gx >= 0 && gx < s.W) unsafeAtomicAdd(gg + (int64_t)cc * plane + gy * s.W + gx, v);
  }
}

// ---------------------------------------------------------------------------------------------
// C[M,N] (+)= sum_batches A_b[M,K] * B_b[N,K]^T, K contiguous in both (the pixel axis): dW = sum_{b,p} dY col^T of the backward
// (deform_conv_cuda.cpp:664-672).  fp32 MFMA 32x32x2; a 256-thread workgroup owns a 128 x 128 tile of C (4 waves as 2 x 2, each
// 64 x 64 = four accumulator tiles, so every LDS operand read feeds two MFMAs), K staged in chunks of 32 through a
// double-buffered LDS pair with odd row strides (conflict-free for the float4 -> 4 x b32 staging writes and for the operand
// reads), global loads of chunk c + 1 issued into registers before the 64 MFMAs of chunk c, one barrier per chunk.
// Deterministic split-K over (image, pixel chunk) into ws[split][M][N], summed by reduce_partials_launch.
// (Round 1 ran this product on rocBLAS; its own first instance had one accumulator tile per wave and a barrier pair per 16 MFMAs.)
constexpr int GK = 32;
struct GemmNT {
  const float *A, *B;
  float *ws;
  int M, N, nb;
  int64_t K, lda, ldb, a_bs, b_bs;
  int chunks_per_batch, total_chunks, splits;
};

template <bool VEC>  // VEC: every row start is 16-byte aligned and K % 4 == 0 (float4 loads), else scalar loads
__global__ __launch_bounds__(256, 2) void gemm_nt_kernel(const GemmNT g) {
  constexpr int BM = 128, BN = 128, LDK = GK + 1;
  __shared__ float as[2][BM * LDK];
  __shared__ float bs[2][BN * LDK];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, half = lane >> 5, j = lane & 31;
  const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN, split = blockIdx.z;
  const int wm = (wave & 1) * 64, wn = (wave >> 1) * 64;
  const int c_begin = (int)((int64_t)g.total_chunks * split / g.splits);
  const int c_end = (int)((int64_t)g.total_chunks * (split + 1) / g.splits);
  f32x16 acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
  // staging role: 4 consecutive k of row (tid >> 3) + 32 i, i = 0..3, of both operands
  const int k4 = (tid & 7) * 4, row0 = tid >> 3;
  float ar[4][4], br[4][4];
  auto load = [&](int c) {
    const int b = c / g.chunks_per_batch;
    const int64_t k = (int64_t)(c - b * g.chunks_per_batch) * GK + k4;
    const float *Ab = g.A + (int64_t)b * g.a_bs, *Bb = g.B + (int64_t)b * g.b_bs;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int row = row0 + 32 * i;
      const bool aok = m0 + row < g.M, bok = n0 + row < g.N;
      const float *pa = Ab + (int64_t)(aok ? m0 + row : 0) * g.lda, *pb = Bb + (int64_t)(bok ? n0 + row : 0) * g.ldb;
      if (VEC) {
        const bool kok = k < g.K;  // K % 4 == 0: the four elements are valid together
        const float4 va = *reinterpret_cast<const float4 *>(pa + (kok ? k : 0)), vb = *reinterpret_cast<const float4 *>(pb + (kok ? k : 0));
        ar[i][0] = (aok && kok) ? va.x : 0.f; ar[i][1] = (aok && kok) ? va.y : 0.f; ar[i][2] = (aok && kok) ? va.z : 0.f; ar[i][3] = (aok && kok) ? va.w : 0.f;
        br[i][0] = (bok && kok) ? vb.x : 0.f; br[i][1] = (bok && kok) ? vb.y : 0.f; br[i][2] = (bok && kok) ? vb.z : 0.f; br[i][3] = (bok && kok) ? vb.w : 0.f;
      } else {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const bool kok = k + q < g.K;
          const float va = pa[kok ? k + q : 0], vb = pb[kok ? k + q : 0];
          ar[i][q] = (aok && kok) ? va : 0.f;
          br[i][q] = (bok && kok) ? vb : 0.f;
        }
      }
    }
  };
  auto commit = [&](int buf) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        as[buf][(row0 + 32 * i) * LDK + k4 + q] = ar[i][q];
        bs[buf][(row0 + 32 * i) * LDK + k4 + q] = br[i][q];
      }
  };
  if (c_begin < c_end) {
    load(c_begin);
    commit(0);
  }
  __syncthreads();
  for (int c = c_begin; c < c_end; ++c) {
    const int buf = (c - c_begin) & 1;
    const bool more = c + 1 < c_end;
    if (more) load(c + 1);
    const float *pa = as[buf] + (wm + j) * LDK + half, *pb = bs[buf] + (wn + j) * LDK + half;
#pragma unroll
    for (int kk = 0; kk < GK; kk += 2) {
      const float a0 = pa[kk], a1 = pa[32 * LDK + kk], b0 = pb[kk], b1 = pb[32 * LDK + kk];
      acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
      acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
      acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
      acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
    }
    if (more) commit(buf ^ 1);  // the idle buffer: last read one iteration ago, a barrier since
    __syncthreads();
  }
  float *out = g.ws + (int64_t)split * g.M * g.N;
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = m0 + wm + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * half, n = n0 + wn + b * 32 + j;
        if (m < g.M && n < g.N) out[(int64_t)m * g.N + n] = acc[a][b][r];
      }
}

static int gemm_splits(int M, int N, int64_t total_chunks) {
  const int tiles = cdiv(M, 128) * cdiv(N, 128);
  int s = 512 / tiles;  // two workgroups per CU, ONE round: rounding up (57 x 9 = 513 workgroups) left one workgroup for a second round
  if (s > total_chunks) s = (int)total_chunks;
  if (s < 1) s = 1;
  if (s > 256) s = 256;
  return s;
}

int gemm_nt_batched(const float *A, const float *B, float *C, int M, int N, int64_t K, int64_t lda, int64_t ldb, int nb,
                           int64_t a_bs, int64_t b_bs, bool accumulate, float *ws, hipStream_t stream) {
  GemmNT g;
  g.A = A; g.B = B; g.ws = ws; g.M = M; g.N = N; g.nb = nb; g.K = K; g.lda = lda; g.ldb = ldb; g.a_bs = a_bs; g.b_bs = b_bs;
  g.chunks_per_batch = (int)cdiv64(K, GK);
  g.total_chunks = g.chunks_per_batch * nb;
  g.splits = gemm_splits(M, N, g.total_chunks);
  dim3 grid(cdiv(N, 128), cdiv(M, 128), g.splits);
  const bool vec = ((K | lda | ldb | a_bs | b_bs) & 3) == 0 && ((reinterpret_cast<uintptr_t>(A) | reinterpret_cast<uintptr_t>(B)) & 15) == 0;
  if (vec) hipLaunchKernelGGL(gemm_nt_kernel<true>, grid, dim3(256), 0, stream, g);
  else hipLaunchKernelGGL(gemm_nt_kernel<false>, grid, dim3(256), 0, stream, g);
  int rc = check_launch("gemm_nt_kernel");
  if (rc) return rc;
  return reduce_partials_launch(ws, C, (int64_t)M * N, g.splits, accumulate ? 1 : 0, stream);
}

size_t gemm_nt_ws_elems_b(int M, int N, int64_t K, int nb) {
  const int64_t chunks = cdiv64(K, GK) * nb;
  return (size_t)gemm_splits(M, N, chunks) * M * N;
}

size_t gemm_nt_ws_elems(int M, int N, int64_t K) { return gemm_nt_ws_elems_b(M, N, K, 1); }

int gemm_nt_launch(const float *A, const float *B, float *C, int M, int N, int64_t K, int64_t lda, int64_t ldb, bool accumulate,
                   float *ws, hipStream_t stream) {
  return gemm_nt_batched(A, B, C, M, N, K, lda, ldb, 1, 0, 0, accumulate, ws, stream);
}

// ---------------------------------------------------------------------------------------------
static inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

int dcn_fill_shape(DcnShape &s, int B, int C, int H, int W, int Co, int kh, int kw, int stride, int pad, int dil, int groups,
                   int dg, int64_t off_bs, int64_t msk_bs) {
  // stride / pad / dil may carry an (h, w) pair: EDVR_HW(h, w) = h | (w + 1) << 16 (include/edvr_amd.h); a plain value means h == w
  auto hw_pair = [](int v, int &h, int &w) {
    h = v & 0xffff;
    w = (v >> 16) ? (v >> 16) - 1 : h;
  };
  int stride_w, pad_w, dil_w;
  EDVR_REQUIRE(stride > 0 && pad >= 0 && dil > 0, "dcnv2: non-positive stride / dilation or negative padding");
  hw_pair(stride, stride, stride_w);
  hw_pair(pad, pad, pad_w);
  hw_pair(dil, dil, dil_w);
  EDVR_REQUIRE(B > 0 && C > 0 && H > 0 && W > 0 && Co > 0 && kh > 0 && kw > 0 && stride > 0 && stride_w > 0 && dil > 0 && dil_w > 0 && groups > 0 && dg > 0,
               "dcnv2: non-positive size");
  EDVR_REQUIRE(C % groups == 0 && Co % groups == 0, "dcnv2: channels (%d -> %d) not divisible by groups %d", C, Co, groups);
  EDVR_REQUIRE(C % dg == 0, "dcnv2: channels %d not divisible by deformable_groups %d", C, dg);
  s.B = B; s.C = C; s.H = H; s.W = W; s.Co = Co; s.kh = kh; s.kw = kw; s.stride = stride; s.pad = pad; s.dil = dil;
  s.stride_w = stride_w; s.pad_w = pad_w; s.dil_w = dil_w;
  s.groups = groups; s.dg = dg;
  s.Ho = (H + 2 * pad - (dil * (kh - 1) + 1)) / stride + 1;
  s.Wo = (W + 2 * pad_w - (dil_w * (kw - 1) + 1)) / stride_w + 1;
  EDVR_REQUIRE(s.Ho > 0 && s.Wo > 0, "dcnv2: convolution input is too small (output would be %dx%d)", s.Ho, s.Wo);
  const int64_t P = (int64_t)s.Ho * s.Wo, K = kh * kw;
  s.off_bs = off_bs ? off_bs : (int64_t)dg * 2 * K * P;
  s.msk_bs = msk_bs < 0 ? 0 : (msk_bs ? msk_bs : (int64_t)dg * K * P);  // < 0: one mask plane set broadcast over the batch (DCNv1)
  s.doff_bs = (int64_t)dg * 2 * K * P;
  s.dmsk_bs = (int64_t)dg * K * P;
  return EDVR_OK;
}

static bool use_fused(const DcnShape &s) {
  const char *e = getenv("EDVR_DCN_FUSED");  // "0": force the generic column-buffer path (A/B, tests)
  if (e && e[0] == '0') return false;
  if (s.stride != s.stride_w || s.pad != s.pad_w || s.dil != s.dil_w) return false;
  return dcn_fused_supported(s.C, s.Co, s.H, s.W, s.kh, s.kw, s.stride, s.pad, s.dil, s.groups, s.dg);
}

struct FwdWs { size_t col, wpk, total; };
static FwdWs fwd_ws(const DcnShape &s) {

  const size_t K = (size_t)s.kh * s.kw, P = (size_t)s.Ho * s.Wo;
  FwdWs w;
  w.col = 0;
  size_t off = align_up((size_t)s.B * s.C * K * P * 4, 256);
  w.wpk = off;
  size_t pk = edvr_conv2d_packed_weight_elems(s.Co / s.groups, (int)((s.C / s.groups) * K), 1) * 4 * s.groups;
  if (use_fused(s)) pk = std::max(pk, edvr_conv2d_packed_weight_elems(s.Co, s.C, 3) * 4);
  off += align_up(pk, 256);
  w.total = off;
  return w;
}

struct BwdWs { size_t col, wpk, gemm, total; };
static BwdWs bwd_ws(const DcnShape &s) {
  const size_t K = (size_t)s.kh * s.kw, P = (size_t)s.Ho * s.Wo;
  const int cig = s.C / s.groups, cog = s.Co / s.groups;
  BwdWs w;
  w.col = 0;
  size_t off = align_up((size_t)s.B * s.C * K * P * 4, 256);
  w.wpk = off;
  off += align_up(std::max(edvr_conv2d_packed_weight_elems((int)(cig * K), cog, 1) * s.groups, dcn_bwd_fused_wbk_elems(s.dg)) * 4, 256);
  w.gemm = off;
  off += align_up(std::max(gemm_nt_ws_elems_b(cog, (int)(cig * K), (int64_t)P, s.B), (size_t)s.B * s.Co * cig * K) * 4, 256);
  w.total = off;
  return w;
}

}  // namespace edvr

extern "C" {

size_t edvr_dcnv2_fwd_ws_bytes(int B, int C, int H, int W, int Co, int kh, int kw, int stride, int pad, int dil, int groups,
                               int dg) {
  edvr::DcnShape s;
  if (edvr::dcn_fill_shape(s, B, C, H, W, Co, kh, kw, stride, pad, dil, groups, dg, 0, 0)) return 0;
  return edvr::fwd_ws(s).total;
}

size_t edvr_dcnv2_bwd_ws_bytes(int B, int C, int H, int W, int Co, int kh, int kw, int stride, int pad, int dil, int groups,
                               int dg) {
  edvr::DcnShape s;
  if (edvr::dcn_fill_shape(s, B, C, H, W, Co, kh, kw, stride, pad, dil, groups, dg, 0, 0)) return 0;
  return edvr::bwd_ws(s).total;
}

static int dcnv2_fwd_impl(const float *x, const float *offset, const float *mask, const float *weight, const float *bias,
                          float *y, int B, int C, int H, int W, int Co, int kh, int kw, int stride, int pad, int dil, int groups,
                          int dg, int64_t offset_bstride, int64_t mask_bstride, int act, int halo_hint, void *ws, size_t ws_bytes,
                          const float *xm_amax, edvr_stream_t stream_) {
  using namespace edvr;
  EDVR_REQUIRE(x && offset && mask && weight && y, "dcnv2_fwd: null pointer");
  DcnShape s;
  int rc = dcn_fill_shape(s, B, C, H, W, Co, kh, kw, stride, pad, dil, groups, dg, offset_bstride, mask_bstride);
  if (rc) return rc;
  const FwdWs wsz = fwd_ws(s);
  if (!ws || ws_bytes < wsz.total) {
    set_error("dcnv2_fwd: workspace %zu < required %zu", ws_bytes, wsz.total);
    return EDVR_ERR_WORKSPACE;
  }
  hipStream_t stream = as_stream(stream_);
  const int K = kh * kw;
  const int64_t P = (int64_t)s.Ho * s.Wo;
  float *col = reinterpret_cast<float *>(static_cast<char *>(ws) + wsz.col);
  float *wpk = reinterpret_cast<float *>(static_cast<char *>(ws) + wsz.wpk);
  if (use_fused(s) && halo_hint == EDVR_DCN_HALO_TAPWIN && xm_amax && dcn_tapwin_split_enabled() &&
      dcn_tapwin_supported(C, Co, H, W, kh, kw, s.stride, s.pad, s.dil, groups, dg) && (reinterpret_cast<uintptr_t>(x) & 15) == 0)
    // the tap-window kernel with split fp32 operands on the f16 matrix pipe (dcn_tapwin_s.hip): its own weight layout in the same workspace
    return dcn_tapwin_split_forward(x, offset, mask, weight, reinterpret_cast<unsigned *>(wpk), bias, y, B, C, H, W, Co, dg, s.off_bs, s.msk_bs, act,
                                    xm_amax, stream);
  if (use_fused(s) && halo_hint >= 0) {
    rc = dcn_fused_pack(weight, wpk, Co, C, stream);
    if (rc) return rc;
    // EDVR_DCN_HALO_TAPWIN: the window follows each tap's displacement (dcn_tapwin.hip) - needs 16-byte aligned rows; layers it
    // does not take (odd widths, other group sizes, unaligned views) run the zero-centred halo kernel, R = 7
    if (halo_hint == EDVR_DCN_HALO_TAPWIN) {
      if (dcn_tapwin_supported(C, Co, H, W, kh, kw, s.stride, s.pad, s.dil, groups, dg) && (reinterpret_cast<uintptr_t>(x) & 15) == 0)
        return dcn_tapwin_forward(x, offset, mask, wpk, bias, y, B, C, H, W, Co, dg, s.off_bs, s.msk_bs, act, stream);
      halo_hint = 7;
    }
    return dcn_fused_forward(x, offset, mask, wpk, bias, y, B, C, H, W, Co, dg, s.off_bs, s.msk_bs, act, halo_hint, stream);
  }
  const int cig = C / groups, cog = Co / groups;
  const size_t wpk_g = edvr_conv2d_packed_weight_elems(cog, cig * K, 1);
  for (int g = 0; g < groups; ++g) {
    rc = edvr_conv2d_pack_weight_f32(weight + (size_t)g * cog * cig * K, wpk + g * wpk_g, cog, cig * K, 1, 0, stream_);
    if (rc) return rc;
  }
  const int64_t total = (int64_t)B * dg * K * P;
  hipLaunchKernelGGL(dcn_im2col_kernel, dim3((unsigned)std::min<int64_t>(cdiv64(total, 256), 1 << 20)), dim3(256), 0, stream, x,
                     offset, mask, col, s);
  rc = check_launch("dcn_im2col_kernel");
  if (rc) return rc;
  for (int g = 0; g < groups; ++g) {
    edvr_conv2d_desc d = {};
    d.x1 = col + (int64_t)g * cig * K * P;
    d.c1 = cig * K;
    d.x1_img_stride = (int64_t)C * K * P;
    d.n = B; d.h = s.Ho; d.w = s.Wo;
    d.wpk = wpk + g * wpk_g;
    d.bias = bias ? bias + g * cog : nullptr;
    d.co = cog; d.ks = 1; d.stride = 1;
    d.act = act;
    d.y = y + (int64_t)g * cog * P;
    d.y_img_stride = (int64_t)Co * P;
    d.out_mode = EDVR_OUT_NCHW;
    rc = conv2d_launch(d, stream);
    if (rc) return rc;
  }
  return EDVR_OK;
}

int edvr_dcnv2_fwd_f32(const float *x, const float *offset, const float *mask, const float *weight, const float *bias,
                       float *y, int B, int C, int H, int W, int Co, int kh, int kw, int stride, int pad, int dil, int groups,
                       int dg, int64_t offset_bstride, int64_t mask_bstride, int act, int halo_hint, void *ws, size_t ws_bytes,
                       edvr_stream_t stream) {
  return dcnv2_fwd_impl(x, offset, mask, weight, bias, y, B, C, H, W, Co, kh, kw, stride, pad, dil, groups, dg, offset_bstride, mask_bstride, act,
                        halo_hint, ws, ws_bytes, nullptr, stream);
}

int edvr_dcnv2_fwd_split_f32(const float *x, const float *offset, const float *mask, const float *weight, const float *bias,
                             float *y, int B, int C, int H, int W, int Co, int kh, int kw, int stride, int pad, int dil, int groups,
                             int dg, int64_t offset_bstride, int64_t mask_bstride, int act, int halo_hint, void *ws, size_t ws_bytes,
                             const float *xm_amax, edvr_stream_t stream) {
  EDVR_REQUIRE(xm_amax != nullptr, "dcnv2_fwd_split: null magnitude bound");
  return dcnv2_fwd_impl(x, offset, mask, weight, bias, y, B, C, H, W, Co, kh, kw, stride, pad, dil, groups, dg, offset_bstride, mask_bstride, act,
                        halo_hint, ws, ws_bytes, xm_amax, stream);
}

int edvr_dcnv2_fwd_split_applies(const float *x, int B, int C, int H, int W, int Co, int kh, int kw, int stride, int pad, int dil, int groups,
                                 int dg, int halo_hint) {
  using namespace edvr;
  DcnShape s;
  if (dcn_fill_shape(s, B, C, H, W, Co, kh, kw, stride, pad, dil, groups, dg, 0, 0)) return 0;
  return (use_fused(s) && halo_hint == EDVR_DCN_HALO_TAPWIN && dcn_tapwin_split_enabled() &&
          dcn_tapwin_supported(C, Co, H, W, kh, kw, s.stride, s.pad, s.dil, groups, dg) && (reinterpret_cast<uintptr_t>(x) & 15) == 0) ? 1 : 0;
}

int edvr_dcnv2_fwd_kernel_name(const float *x, int B, int C, int H, int W, int Co, int kh, int kw, int stride, int pad, int dil, int groups,
                               int dg, int halo_hint, char *buf, size_t buf_len) {
  using namespace edvr;
  EDVR_REQUIRE(buf && buf_len > 0, "dcnv2_fwd_kernel_name: null buffer");
  DcnShape s;
  int rc = dcn_fill_shape(s, B, C, H, W, Co, kh, kw, stride, pad, dil, groups, dg, 0, 0);
  if (rc) return rc;
  const char *name = "dcn_im2col_kernel";
  if (use_fused(s) && halo_hint >= 0) {
    const bool tapwin = halo_hint == EDVR_DCN_HALO_TAPWIN && dcn_tapwin_supported(C, Co, H, W, kh, kw, s.stride, s.pad, s.dil, groups, dg) &&
                        (reinterpret_cast<uintptr_t>(x) & 15) == 0;
    name = tapwin ? "dcn_tapwin_fwd_kernel" : "dcn_fused_fwd_kernel";
  }
  snprintf(buf, buf_len, "%s", name);
  return EDVR_OK;
}

// xm_amax / dy_amax (both or neither): device pointers to ONE float >= max |x| * max(1, max |mask|) and >= max |dy| - the dW product
// then runs in its split-operand form (gemm_nt_s.hip)
static int dcnv2_bwd_impl(const float *x, const float *offset, const float *mask, const float *weight, const float *dy, float *dx,
                          float *doffset, float *dmask, float *dweight, float *dbias, int B, int C, int H, int W, int Co, int kh,
                          int kw, int stride, int pad, int dil, int groups, int dg, int64_t offset_bstride, int64_t mask_bstride,
                          int64_t doffset_bstride, int64_t dmask_bstride, int scatter_hint, void *ws, size_t ws_bytes,
                          const float *xm_amax, const float *dy_amax, edvr_stream_t stream_) {
  using namespace edvr;
  EDVR_REQUIRE(x && offset && mask && weight && dy && dx && doffset && dmask && dweight, "dcnv2_bwd: null pointer");
  DcnShape s;
  int rc = dcn_fill_shape(s, B, C, H, W, Co, kh, kw, stride, pad, dil, groups, dg, offset_bstride, mask_bstride);
  if (rc) return rc;
  if (doffset_bstride) s.doff_bs = doffset_bstride;
  if (dmask_bstride) s.dmsk_bs = dmask_bstride < 0 ? 0 : dmask_bstride;  // < 0: every image writes the same (discarded) planes
  const BwdWs wsz = bwd_ws(s);
  if (!ws || ws_bytes < wsz.total) {
    set_error("dcnv2_bwd: workspace %zu < required %zu", ws_bytes, wsz.total);
    return EDVR_ERR_WORKSPACE;
  }
  hipStream_t stream = as_stream(stream_);
  const int K = kh * kw;
  const int64_t P = (int64_t)s.Ho * s.Wo;
  float *col = reinterpret_cast<float *>(static_cast<char *>(ws) + wsz.col);
  float *wpk = reinterpret_cast<float *>(static_cast<char *>(ws) + wsz.wpk);
  float *gws = reinterpret_cast<float *>(static_cast<char *>(ws) + wsz.gemm);
  const int cig = C / groups, cog = Co / groups;
  if (scatter_hint == EDVR_DCN_SCATTER_STRIP && dcn_bwd_fused_supported(s)) {
    // 1 + 2 in one kernel: the dcol slice of a (group, tap pair) is consumed in the registers the matrix core left it in.  Its dX
    // route is built for sub-pixel offsets, the same layers the strip hint marks (6.4 vs 8.6 ms at sigma 0.3, 13.2 vs 16.6 at 0.5)
    if (hipMemsetAsync(dx, 0, (size_t)B * C * H * W * sizeof(float), stream) != hipSuccess) {
      set_error("dcnv2_bwd: hipMemsetAsync failed");
      return EDVR_ERR_LAUNCH;
    }
    rc = dcn_bwd_fused_launch(s, x, offset, mask, weight, dy, wpk, col, dx, doffset, dmask, stream);
    if (rc) return rc;
  } else {
    // 1. dcol[b, g] (cig*K x P) = W[g]^T (cig*K x cog) dY[b, g] (cog x P): a 1x1 convolution of dY with cog -> cig*K channels, on the
    //    streaming-GEMM kernel of conv1x1.hip (B operand = dY straight from global memory, W slab in LDS)
    const size_t wpk_g = edvr_conv2d_packed_weight_elems(cig * K, cog, 1);
    for (int g = 0; g < groups; ++g) {
      const float *wg = weight + (size_t)g * cog * cig * K;
      const float *wuse = wg;
      if ((cog % 32) != 0 || ((cig * K) % 32) != 0) {  // W (cog x cig*K, row-major) is already the packed layout when aligned
        rc = edvr_conv2d_pack_weight_f32(wg, wpk + g * wpk_g, cig * K, cog, 1, 1, stream_);
        if (rc) return rc;
        wuse = wpk + g * wpk_g;
      }
      edvr_conv2d_desc d = {};
      d.x1 = dy + (int64_t)g * cog * P;
      d.c1 = cog;
      d.x1_img_stride = (int64_t)Co * P;
      d.n = B; d.h = s.Ho; d.w = s.Wo;
      d.wpk = wuse;
      d.co = cig * K; d.ks = 1; d.stride = 1;
      d.y = col + (int64_t)g * cig * K * P;
      d.y_img_stride = (int64_t)C * K * P;
      rc = conv2d_launch(d, stream);
      if (rc) return rc;
    }
    // 2. dOffset, dMask, dX (+ forward columns in place)
    if (hipMemsetAsync(dx, 0, (size_t)B * C * H * W * sizeof(float), stream) != hipSuccess) {
      set_error("dcnv2_bwd: hipMemsetAsync failed");
      return EDVR_ERR_LAUNCH;
    }
    static const bool use_tile = []() {
      const char *e = getenv("EDVR_DCN_BWD_TILE");  // "0": never use the LDS-window kernel (A/B)
      return !(e && e[0] == '0');
    }();
    const bool edvr_sig = kh == 3 && kw == 3 && s.stride == 1 && s.pad == 1 && s.dil == 1 && s.stride_w == 1 && s.pad_w == 1 && s.dil_w == 1;
    if (scatter_hint == EDVR_DCN_SCATTER_STRIP && edvr_sig && dg * cdiv(W, 64) <= 65535) {
      // dX by the register-ring kernel (reads dcol before the next kernel rewrites it), then dOffset / dMask / columns without dX
      hipLaunchKernelGGL(dcn_bwd_dx_strip_kernel, dim3(dg * cdiv(W, 64), B), dim3(256), 0, stream, offset, mask, col, dx, s);
      const int64_t total = (int64_t)B * dg * K * P;
      hipLaunchKernelGGL(dcn_bwd_coord_kernel<false>, dim3((unsigned)std::min<int64_t>(cdiv64(total, 256), 1 << 20)), dim3(256), 0, stream,
                         x, offset, mask, col, static_cast<float *>(nullptr), doffset, dmask, s);
      rc = check_launch("dcn_bwd_dx_strip_kernel + dcn_bwd_coord_kernel");
    } else if (use_tile && scatter_hint != EDVR_DCN_SCATTER_DEVICE && scatter_hint != EDVR_DCN_SCATTER_STRIP && edvr_sig && C / dg <= 16) {
      const int tiles_x = cdiv(s.Wo, 32), tiles_y = cdiv(s.Ho, 8);
      // window margin beyond the tile's 3x3 footprint (a corner outside the window is a device atomic): 3 px (42 KB of LDS, three workgroups
      // per CU) or, EDVR_DCN_SCATTER_LDS_WIDE, 6 px (66 KB, two per CU) for fields whose taps sit ~4 px out and more (15.0 -> 11.0 ms per
      // call at sigma 6 px per tap, 9.7 vs 9.8 at sigma 4, 7.9 vs 9.9 at sigma 2; 8 px = one workgroup per CU: 15-17 ms everywhere:
      // profiles/r6/dcn_bwd_margin.log)
      if (scatter_hint == EDVR_DCN_SCATTER_LDS_WIDE)
        hipLaunchKernelGGL((dcn_bwd_coord_tile_kernel<6>), dim3(tiles_x * tiles_y, dg, B), dim3(256), 0, stream, x, offset, mask, col, dx, doffset, dmask, s, tiles_x);
      else
        hipLaunchKernelGGL((dcn_bwd_coord_tile_kernel<3>), dim3(tiles_x * tiles_y, dg, B), dim3(256), 0, stream, x, offset, mask, col, dx, doffset, dmask, s, tiles_x);
      rc = check_launch("dcn_bwd_coord_tile_kernel");
    } else {
      const int64_t total = (int64_t)B * dg * K * P;
      hipLaunchKernelGGL(dcn_bwd_coord_kernel<true>, dim3((unsigned)std::min<int64_t>(cdiv64(total, 256), 1 << 20)), dim3(256), 0, stream, x,
                         offset, mask, col, dx, doffset, dmask, s);
      rc = check_launch("dcn_bwd_coord_kernel");
    }
    if (rc) return rc;
  }
  // 3. dW[g] = sum_{b,p} dY[b, g] col[b, g]^T ; db = sum dY
  for (int g = 0; g < groups; ++g) {
    rc = (xm_amax && dy_amax && gemm_nt_split_enabled())
             ? gemm_nt_split_batched(dy + (int64_t)g * cog * P, col + (int64_t)g * cig * K * P, dweight + (size_t)g * cog * cig * K, cog, cig * K,
                                     P, P, P, B, (int64_t)Co * P, (int64_t)C * K * P, false, gws, dy_amax, xm_amax, stream)
             : gemm_nt_batched(dy + (int64_t)g * cog * P, col + (int64_t)g * cig * K * P, dweight + (size_t)g * cog * cig * K, cog, cig * K, P,
                               P, P, B, (int64_t)Co * P, (int64_t)C * K * P, false, gws, stream);
    if (rc) return rc;
  }
  if (dbias) {  // db = sum over (image, pixel) of dY: the conv layers' bias-gradient kernel (gws is free again here)
    rc = edvr_channel_sum_f32(dy, dbias, B, Co, P, (int64_t)Co * P, gws, (size_t)64 * Co * sizeof(float), stream_);
    if (rc) return rc;
  }
  return EDVR_OK;
}

int edvr_dcnv2_bwd_f32(const float *x, const float *offset, const float *mask, const float *weight, const float *dy, float *dx,
                       float *doffset, float *dmask, float *dweight, float *dbias, int B, int C, int H, int W, int Co, int kh,
                       int kw, int stride, int pad, int dil, int groups, int dg, int64_t offset_bstride, int64_t mask_bstride,
                       int64_t doffset_bstride, int64_t dmask_bstride, int scatter_hint, void *ws, size_t ws_bytes,
                       edvr_stream_t stream) {
  return dcnv2_bwd_impl(x, offset, mask, weight, dy, dx, doffset, dmask, dweight, dbias, B, C, H, W, Co, kh, kw, stride, pad, dil, groups, dg,
                        offset_bstride, mask_bstride, doffset_bstride, dmask_bstride, scatter_hint, ws, ws_bytes, nullptr, nullptr, stream);
}

int edvr_dcnv2_bwd_split_f32(const float *x, const float *offset, const float *mask, const float *weight, const float *dy, float *dx,
                             float *doffset, float *dmask, float *dweight, float *dbias, int B, int C, int H, int W, int Co, int kh,
                             int kw, int stride, int pad, int dil, int groups, int dg, int64_t offset_bstride, int64_t mask_bstride,
                             int64_t doffset_bstride, int64_t dmask_bstride, int scatter_hint, void *ws, size_t ws_bytes,
                             const float *xm_amax, const float *dy_amax, edvr_stream_t stream) {
  EDVR_REQUIRE(xm_amax && dy_amax, "dcnv2_bwd_split: null magnitude bound");
  return dcnv2_bwd_impl(x, offset, mask, weight, dy, dx, doffset, dmask, dweight, dbias, B, C, H, W, Co, kh, kw, stride, pad, dil, groups, dg,
                        offset_bstride, mask_bstride, doffset_bstride, dmask_bstride, scatter_hint, ws, ws_bytes, xm_amax, dy_amax, stream);
}

int edvr_dcnv2_bwd_split_applies(void) { return edvr::gemm_nt_split_enabled() ? 1 : 0; }


// ------------------------------------------------------------------------------------------------ DCNv1 (DeformConv)
// deform_conv_forward / deform_conv_backward_input / deform_conv_backward_parameters of the reference
// (deform_conv_ext.cpp:51-104, deform_conv_cuda.cpp:152-488): the same gather as DCNv2 without the modulation mask and without
// bias (deformable_im2col, .cu:190-250, is modulated_deformable_im2col with mask = 1).  Runs the DCNv2 kernels on ONE
// all-ones mask plane set that is broadcast over the batch (mask image stride 0); d(mask) lands in a scratch plane set.
// The reference's im2col_step batching is a workspace-size knob of its column buffer and has no counterpart here.
size_t edvr_dcnv1_fwd_ws_bytes(int B, int C, int H, int W, int Co, int kh, int kw, int stride, int pad, int dil, int groups, int dg) {
  edvr::DcnShape s;
  if (edvr::dcn_fill_shape(s, B, C, H, W, Co, kh, kw, stride, pad, dil, groups, dg, 0, 0)) return 0;
  return edvr::fwd_ws(s).total + edvr::align_up((size_t)dg * kh * kw * s.Ho * s.Wo * 4, 256);
}

int edvr_dcnv1_fwd_f32(const float *x, const float *offset, const float *weight, float *y, int B, int C, int H, int W, int Co, int kh,
                       int kw, int stride, int pad, int dil, int groups, int dg, int64_t offset_bstride, int halo_hint, void *ws,
                       size_t ws_bytes, edvr_stream_t stream_) {
  using namespace edvr;
  const size_t v2 = edvr_dcnv2_fwd_ws_bytes(B, C, H, W, Co, kh, kw, stride, pad, dil, groups, dg);
  const size_t need = edvr_dcnv1_fwd_ws_bytes(B, C, H, W, Co, kh, kw, stride, pad, dil, groups, dg);
  EDVR_REQUIRE(v2 && need, "dcnv1: bad shape");
  if (!ws || ws_bytes < need) {
    set_error("dcnv1_fwd: workspace %zu < required %zu", ws_bytes, need);
    return EDVR_ERR_WORKSPACE;
  }
  float *ones = reinterpret_cast<float *>(static_cast<char *>(ws) + v2);
  const int64_t n1 = (int64_t)(need - v2) / 4;
  hipLaunchKernelGGL(fill_kernel, dim3((unsigned)std::min<int64_t>(cdiv64(n1, 256), 4096)), dim3(256), 0, as_stream(stream_), ones, 1.f, n1);
  int rc = check_launch("fill_kernel");
  if (rc) return rc;
  return edvr_dcnv2_fwd_f32(x, offset, ones, weight, nullptr, y, B, C, H, W, Co, kh, kw, stride, pad, dil, groups, dg, offset_bstride, -1,
                            EDVR_ACT_NONE, halo_hint, ws, v2, stream_);
}

size_t edvr_dcnv1_bwd_ws_bytes(int B, int C, int H, int W, int Co, int kh, int kw, int stride, int pad, int dil, int groups, int dg) {
  edvr::DcnShape s;
  if (edvr::dcn_fill_shape(s, B, C, H, W, Co, kh, kw, stride, pad, dil, groups, dg, 0, 0)) return 0;
  return edvr::bwd_ws(s).total + 2 * edvr::align_up((size_t)dg * kh * kw * s.Ho * s.Wo * 4, 256);
}

int edvr_dcnv1_bwd_f32(const float *x, const float *offset, const float *weight, const float *dy, float *dx, float *doffset,
                       float *dweight, int B, int C, int H, int W, int Co, int kh, int kw, int stride, int pad, int dil, int groups,
                       int dg, int64_t offset_bstride, int64_t doffset_bstride, int scatter_hint, void *ws, size_t ws_bytes,
                       edvr_stream_t stream_) {
  using namespace edvr;
  const size_t v2 = edvr_dcnv2_bwd_ws_bytes(B, C, H, W, Co, kh, kw, stride, pad, dil, groups, dg);
  const size_t need = edvr_dcnv1_bwd_ws_bytes(B, C, H, W, Co, kh, kw, stride, pad, dil, groups, dg);
  EDVR_REQUIRE(v2 && need, "dcnv1: bad shape");
  if (!ws || ws_bytes < need) {
    set_error("dcnv1_bwd: workspace %zu < required %zu", ws_bytes, need);
    return EDVR_ERR_WORKSPACE;
  }
  const size_t plane = (need - v2) / 2;
  float *ones = reinterpret_cast<float *>(static_cast<char *>(ws) + v2);
  float *dmask_scratch = reinterpret_cast<float *>(static_cast<char *>(ws) + v2 + plane);
  const int64_t n1 = (int64_t)plane / 4;
  hipLaunchKernelGGL(fill_kernel, dim3((unsigned)std::min<int64_t>(cdiv64(n1, 256), 4096)), dim3(256), 0, as_stream(stream_), ones, 1.f, n1);
  int rc = check_launch("fill_kernel");
  if (rc) return rc;
  return edvr_dcnv2_bwd_f32(x, offset, ones, weight, dy, dx, doffset, dmask_scratch, dweight, nullptr, B, C, H, W, Co, kh, kw, stride, pad,
                            dil, groups, dg, offset_bstride, -1, doffset_bstride, -1, scatter_hint, ws, v2, stream_);
}

}  // extern "C"
