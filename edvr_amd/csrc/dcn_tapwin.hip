// dcn_tapwin.hip - fused modulated deformable convolution forward whose cost does not depend on the SIZE of the offsets (gfx950).
//
// Same arithmetic as modulated_deformable_im2col_gpu_kernel + addmm_ of the reference
// (basicsr/models/ops/dcn/src/deform_conv_cuda_kernel.cu:570-633, deform_conv_cuda.cpp:550-555) for the EDVR signature
// (3x3, stride 1, pad 1, dilation 1, groups 1; edvr_arch.py:47-52,61-66), without the column buffer, like dcn_fused.hip.
// The reference's kernel costs the same at any offset: every column entry is four global loads.  dcn_fused.hip stages ONE
// zero-centred halo tile per 8 input channels and is fast only while |offset| stays inside it (R = 3 / 7); a trained EDVR
// predicts multi-pixel, spatially smooth displacements - a different one for each of the dg x 9 taps.  Here the staged window
// follows the displacement:
//
//   walk (deformable group g) -> (tap t) -> (channel pair):  one step = one (g, t) = 16 channels x MT x 32 output channels.
//   Per step the input window of the group's channels is fetched by LDS-DMA around  tile + regular tap position + s(g, t),
//   s = the rounded mid-range of that tap's offsets over the tile (a small pre-pass per workgroup): 12 rows x 40 columns per
//   channel (dense, 480 floats) in 16-byte pieces, i.e. +-2 px of slack around the 8 x 32 pixel tile in both directions after the shift (the
//   columns start on a multiple of 4 so that every piece is one aligned dwordx4).  Pieces outside the image carry an
//   out-of-range buffer offset and arrive as zeros: the reference's per-corner bounds test (.cu:481-491) falls out of the data.
//   A lane whose 2x2 cell still leaves the window (an object boundary inside the tile, a rough field) takes a wave-uniform
//   slow path through global memory with the full bounds logic - correctness never depends on the window.
//
//   v_mfma_f32_32x32x2_f32:  D[co, pixel] += W[co, (c, t)] * col[(c, t), pixel];  lanes 0-31 / 32-63 sample channels c / c + 1 of
//   the same 32 pixels.  The tap's sampling state (4 bilinear weights x mask + one LDS address) lives for ONE step only - 5
//   registers per pixel instead of the 45 dcn_fused.hip keeps for the 9 taps of a group - so a wave owns TWO pixel rows
//   (sub-tiles): every weight operand read from LDS feeds 2 x MT MFMAs instead of MT.
//
// Per workgroup (256 threads, 4 waves, 2 workgroups per CU): 8 x 32 output pixels x up to 128 output channels.
// LDS: 2 x 30 KB windows + 2 x 8 KB weight slabs (both double-buffered, both by LDS-DMA: no staging registers) + the shifts.
// One barrier per step (64 MFMAs per wave); nothing waits on vector memory between the DMA issue at the top of a step and the
// last MFMA of the step (no scratch, tap values consumed after the MFMAs).
#include <type_traits>

#include "common.h"
#include "dcn_tap.h"

namespace edvr {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

namespace {
constexpr int TWN_TH = 8, TWN_TW = 32, TWN_RY = 2, TWN_RX = 2;
constexpr int TWN_IH = TWN_TH + 2 * TWN_RY;               // 12 rows: fl(dy) - s in [-RY, RY - 1], plus the cell's lower row
constexpr int TWN_IW = (TWN_TW + 2 * TWN_RX + 3 + 3) / 4 * 4;  // 40 columns: 36 needed + up to 3 lost to the 16-byte alignment
constexpr int TWN_PPR = TWN_IW / 4;                       // 10 16-byte pieces per row
constexpr int TWN_HR = TWN_IH / 2;                        // one DMA instruction moves 6 rows x 10 pieces (lanes 0-59) of one channel
constexpr int TWN_HB = TWN_HR * TWN_IW;                   // ... to floats [0, 240) / [240, 480) of the channel: the rows stay dense
constexpr int TWN_CHS = TWN_IH * TWN_IW;                  // 480 floats per channel = 32 (mod 64): the two half-waves of a gather (channels
                                                          // c / c + 1, same position) fall on disjoint halves of the 64 LDS banks
static_assert(TWN_HR * TWN_PPR <= 64 && TWN_IH == 2 * TWN_HR && TWN_CHS % 64 == 32, "window halves must fit one wave instruction");
constexpr int TWN_MAX_DG = 8;   // (the shift table and the tap staging area share the last 3.6 KB of the workgroup's 80 KB)
constexpr int TWN_OOB = (int)0x80000000;
constexpr int TWN_RSRC_FLAGS = 0x00020000;
}  // namespace

struct DcnTapwinArgs {
  const float *x, *offset, *mask, *wpk, *bias;
  float *y;
  int B, C, H, W, Co, dg, cop, act, co_start, tiles_x, tiles_y;
  int64_t off_bs, msk_bs;
};

typedef int i32x4 __attribute__((ext_vector_type(4)));

// LDS-DMA from inline assembly: lane l of the wave delivers its 16 bytes to LDS byte address `lds` + 16 l.  The compiler treats the
// builtin form as a store to LDS that any later LDS read may alias and puts `s_waitcnt vmcnt(0)` in front of the first ds_read
// after it - i.e. the fetch of step k + 1 would be WAITED FOR at the top of step k instead of overlapping its 64 MFMAs (that wait is
// in dcn_fused.hip's code, at the top of every half chunk).  Issued this way the request is invisible to the wait-count pass; the
// kernel waits for it itself (`s_waitcnt vmcnt(0)` after the MFMAs of the step) and the barrier orders it against the readers.
// (m0 = LDS base; one wait state between the scalar write of m0 and the instruction that uses it)
__device__ __forceinline__ void twn_dma16(i32x4 rsrc, unsigned lds, int voff, int soff) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" ::"s"(lds), "v"(voff), "s"(rsrc), "s"(soff) : "memory", "m0");
}

__device__ __forceinline__ void twn_dma4(i32x4 rsrc, unsigned lds, int voff, int soff) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dword %1, %2, %3 offen lds" ::"s"(lds), "v"(voff), "s"(rsrc), "s"(soff) : "memory", "m0");
}

__device__ __forceinline__ i32x4 twn_rsrc4(const void *ptr, int bytes) {
  const uint64_t pv = reinterpret_cast<uint64_t>(ptr);
  i32x4 r;
  r[0] = __builtin_amdgcn_readfirstlane((int)(uint32_t)pv);
  r[1] = __builtin_amdgcn_readfirstlane((int)(pv >> 32)) & 0xffff;
  r[2] = bytes;
  r[3] = TWN_RSRC_FLAGS;
  return r;
}

__device__ __forceinline__ __amdgpu_buffer_rsrc_t twn_rsrc(const void *ptr, int bytes) {
  const uint64_t pv = reinterpret_cast<uint64_t>(ptr);
  const uint64_t pu = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(pv >> 32)) << 32) |
                      (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)pv);
  return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void *>(pu), (short)0, bytes, TWN_RSRC_FLAGS);
}

template <int MT, int CPG>
__global__ __launch_bounds__(256, 2) void dcn_tapwin_fwd_kernel(const DcnTapwinArgs a) {
  constexpr int TH = TWN_TH, TW = TWN_TW, IH = TWN_IH, IW = TWN_IW, CHS = TWN_CHS, KK = 9, OOB = TWN_OOB;
  constexpr int MB = 32 * MT, NQ = CPG / 2;
  constexpr int XW = CPG * CHS;            // floats of one window buffer (16 channels: 7680 = 30 KB)
  constexpr int WS = CPG * MB;             // floats of one weight slab (16 x 128: 8 KB)
  constexpr int CPW = CPG / 4;             // channels whose window this wave fetches (2 instructions each)
  static_assert(CPG % 4 == 0, "four waves share the channels of a group");
  __shared__ __attribute__((aligned(16))) float xw[2 * XW];
  __shared__ __attribute__((aligned(16))) float wsl[2 * WS];
  __shared__ int shifts[TWN_MAX_DG * 18];
  __shared__ float tpl[4 * 3 * 64];  // offsets / mask of the NEXT step's tap: [wave][dy | dx | mask][pixel row 2][column 32], wave-private

  const int tid = threadIdx.x, lane = tid & 63, half = lane >> 5, j = lane & 31;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  int tile, blk_y, img;
  xcd_block_index(tile, blk_y, img);  // neighbouring tiles share one XCD's L2 (common.h)
  tile = __builtin_amdgcn_readfirstlane(tile);
  blk_y = __builtin_amdgcn_readfirstlane(blk_y);
  img = __builtin_amdgcn_readfirstlane(img);
  const int ty0 = (tile / a.tiles_x) * TH, tx0 = (tile % a.tiles_x) * TW;
  const int co_blk = a.co_start + blk_y * MB;
  const int P = a.H * a.W;
  const float *x_img = a.x + (int64_t)img * a.C * P;
  const float *off_b = a.offset + (int64_t)img * a.off_bs;
  const float *msk_b = a.mask + (int64_t)img * a.msk_bs;

  // ---- pre-pass: the window shift of every (group, tap, dy | dx) = rounded mid-range of that offset plane over 2 x 4 sample
  //      pixels of the tile (a smooth field varies little inside 8 x 32 pixels; whatever the samples miss goes the slow way)
  for (int v = tid; v < a.dg * 18; v += 256) {
    const float *pl = off_b + (int64_t)v * P;  // plane g * 18 + 2 t + {0: dy, 1: dx} = v
    float mn = 3.0e38f, mx = -3.0e38f;
#pragma unroll
    for (int r = 1; r < TH; r += 4) {
      const int yy = min(ty0 + r, a.H - 1);
#pragma unroll
      for (int c = 4; c < TW; c += 8) {
        const float f = pl[yy * a.W + min(tx0 + c, a.W - 1)];
        mn = fminf(mn, f);
        mx = fmaxf(mx, f);
      }
    }
    const float mid = fminf(fmaxf(0.5f * (mn + mx), -16384.f), 16384.f);
    shifts[v] = (int)floorf(mid + 0.5f);
  }

  // ---- this lane's two output pixels (sub-tile s = row 2 wave + s of the tile, column j)
  const int oy0 = ty0 + 2 * wave, ox = tx0 + j;
  const bool ok0 = oy0 < a.H && ox < a.W, ok1 = oy0 + 1 < a.H && ox < a.W;
  const int p0 = oy0 * a.W + ox;
  const int tvo[2] = {ok0 ? p0 * 4 : OOB, ok1 ? (p0 + a.W) * 4 : OOB};

  f32x16 acc[2][MT];
#pragma unroll
  for (int s = 0; s < 2; ++s)
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[s][m][r] = 0.f;

  // ---- window DMA: wave w fetches channels CPW w .. CPW w + CPW - 1 of the group, two instructions per channel (rows 0-5 / 6-11);
  //      lane l < 60 carries piece (row l / 10, columns 4 (l % 10) ..), the other four lanes are masked off (LDS-DMA writes active
  //      lanes only).  Per lane: ONE shift-independent byte offset; per step: one offset per half, out-of-range where the piece
  //      lies outside the image (W % 4 == 0: a piece is inside or outside as a whole).
  const int d_r = lane / TWN_PPR, d_c = 4 * (lane - d_r * TWN_PPR);
  const int rel0 = (d_r * a.W + d_c) * 4;
  const __amdgpu_buffer_rsrc_t x_rsrc = twn_rsrc(x_img, a.C * P * 4);  // (fix-up pass)
  const i32x4 x_rsrc4 = twn_rsrc4(x_img, a.C * P * 4), w_rsrc4 = twn_rsrc4(a.wpk, a.C * KK * a.cop * 4);
  typedef __attribute__((address_space(3))) void lvoid;
  const unsigned xw_lds = (unsigned)(size_t)(lvoid *)xw, ws_lds = (unsigned)(size_t)(lvoid *)wsl;
  const __amdgpu_buffer_rsrc_t off_rsrc = twn_rsrc(off_b, a.dg * 18 * P * 4), msk_rsrc = twn_rsrc(msk_b, a.dg * 9 * P * 4);
  auto dma_x = [&](int buf, int g, int wy0, int wx0) {
    const int sh = (wy0 * a.W + wx0) * 4;
    const bool colok = (unsigned)(wx0 + d_c) < (unsigned)a.W;
    const int vo0 = (colok && (unsigned)(wy0 + d_r) < (unsigned)a.H) ? rel0 + sh : OOB;
    const int vo1 = (colok && (unsigned)(wy0 + d_r + TWN_HR) < (unsigned)a.H) ? rel0 + sh + TWN_HR * a.W * 4 : OOB;
    if (lane < TWN_HR * TWN_PPR) {
#pragma unroll
      for (int cc = 0; cc < CPW; ++cc) {
        const int ch = wave * CPW + cc;
        const unsigned dst = xw_lds + (buf * XW + ch * CHS) * 4;
        twn_dma16(x_rsrc4, dst, vo0, (g * CPG + ch) * P * 4);
        twn_dma16(x_rsrc4, dst + TWN_HB * 4, vo1, (g * CPG + ch) * P * 4);
      }
    }
  };
  // ---- weight slab of (g, t): rows (channel 16 g + cc, tap t) of the packed weights, MB floats each, dense [cc][MB] in LDS:
  //      piece q = 64 i + lane of the slab is (row q / Q, 16-byte column q % Q); wave w issues instructions i = w, w + 4, ...
  constexpr int WQ = MB / 4, WTOTAL = CPG * WQ, WNI = WTOTAL / 64, WNK = (WNI + 3) / 4;
  static_assert(WTOTAL % 64 == 0, "weight slab must fill whole wave instructions");
  int wvo[WNK];
#pragma unroll
  for (int k = 0; k < WNK; ++k) {
    const int q = (wave + 4 * k) * 64 + lane, row = q / WQ, c4 = q - row * WQ;
    wvo[k] = (row * KK * a.cop + c4 * 4) * 4;
  }
  auto dma_w = [&](int buf, int g, int t) {
    const int so = ((g * CPG * KK + t) * a.cop + co_blk) * 4;
#pragma unroll
    for (int k = 0; k < WNK; ++k)
      if (wave + 4 * k < WNI) twn_dma16(w_rsrc4, ws_lds + (buf * WS + (wave + 4 * k) * 256) * 4, wvo[k], so);
  };
  // window origin of step (g, t): rows  ty0 - 1 + ti + sy - RY ..,  columns from the multiple of 4 at or below  tx0 - 1 + tj + sx - RX
  auto origin = [&](int g, int t, int &wy0, int &wx0) {
    const int sy = __builtin_amdgcn_readfirstlane(shifts[(g * KK + t) * 2]);
    const int sx = __builtin_amdgcn_readfirstlane(shifts[(g * KK + t) * 2 + 1]);
    const int ti = t / 3, tj = t - 3 * ti;
    wy0 = ty0 - 1 + ti + sy - TWN_RY;
    wx0 = (tx0 - 1 + tj + sx - TWN_RX) & ~3;
  };
  // offsets / mask of a tap for this wave's 64 pixels -> the wave's private staging rows, by LDS-DMA (lane = (pixel row, column)),
  // requested TWO steps ahead; a lane reads the six values of its two pixels (both half-waves read the same words: broadcast)
  const i32x4 off_rsrc4 = twn_rsrc4(off_b, a.dg * 18 * P * 4), msk_rsrc4 = twn_rsrc4(msk_b, a.dg * 9 * P * 4);
  const unsigned tpl_lds = (unsigned)(size_t)(lvoid *)tpl + wave * (3 * 64 * 4);
  const int tvd = half ? tvo[1] : tvo[0];  // (here the lane's upper bit selects the pixel ROW, not the channel parity)
  auto dma_taps = [&](int g, int t) {
    twn_dma4(off_rsrc4, tpl_lds, tvd, (g * 18 + 2 * t) * P * 4);
    twn_dma4(off_rsrc4, tpl_lds + 256, tvd, (g * 18 + 2 * t + 1) * P * 4);
    twn_dma4(msk_rsrc4, tpl_lds + 512, tvd, (g * 9 + t) * P * 4);
  };
  auto read_taps = [&](float (&d)[6]) {
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int v = 0; v < 3; ++v) d[3 * s + v] = tpl[wave * 192 + v * 64 + s * 32 + j];
  };

  // ---- per-step sampling state of the two pixels
  struct St {
    float bw[2][4];
    int addr[2];
    unsigned slow;      // bit s: the 2x2 cell of sub-tile s's pixel is valid but not inside the staged window
    unsigned slow_any;  // any lane of the wave (scalar)
  };
  // (pure register arithmetic, no memory access; split per sub-tile so that the halves can sit in different stages of the step before)
  auto state_half = [&](St &st, int s, int t, int wy0, int wx0, const float (&d)[6]) {
    const int ti = t / 3, tj = t - 3 * ti;
    const bool pix_ok = s ? ok1 : ok0;
    const float h = (float)(oy0 + s - 1 + ti) + d[3 * s], w = (float)(ox - 1 + tj) + d[3 * s + 1];
    const bool valid = pix_ok && h > -1.f && w > -1.f && h < (float)a.H && w < (float)a.W;  // .cu:618
    const float fh = floorf(h), fw = floorf(w);
    const float lh = h - fh, lw = w - fw, m = d[3 * s + 2];
    const int ry = (int)fh - wy0, rx = (int)fw - wx0;
    const bool inside = valid && ry >= 0 && ry <= IH - 2 && rx >= 0 && rx <= IW - 2;
    const float mm = inside ? m : 0.f;  // (a valid tap outside the window: zero here, added by the fix-up pass)
    const float hm = (1.f - lh) * mm, lm = lh * mm, hw = 1.f - lw;
    st.bw[s][0] = hm * hw;
    st.bw[s][1] = hm * lw;
    st.bw[s][2] = lm * hw;
    st.bw[s][3] = lm * lw;
    st.addr[s] = half * CHS + (inside ? ry * IW + rx : 0);
    st.slow = (s ? st.slow : 0u) | ((valid && !inside) ? 1u << s : 0u);
  };
  auto state_any = [&](St &st) { st.slow_any = __builtin_amdgcn_readfirstlane(__any(st.slow != 0) ? 1 : 0); };
  auto make_state = [&](St &st, int t, int wy0, int wx0, const float (&d)[6]) {
    state_half(st, 0, t, wy0, wx0, d);
    state_half(st, 1, t, wy0, wx0, d);
    state_any(st);
  };

  // ---- one step: NQ channel pairs x 2 sub-tiles x MT MFMAs.  Operands of pair q + 2 are requested while pair q multiplies.
  auto run_step = [&](auto SLOWT, const St &cur, const float *xb, const float *wb, int g, int t, St &nxt, int tn, int ny0, int nx0,
                      const float (&tnv)[6]) {
    constexpr bool SLOW = decltype(SLOWT)::value;
    const float (&bw)[2][4] = cur.bw;
    const unsigned slow = cur.slow;
    const unsigned cbase[2] = {(unsigned)(size_t)(lvoid *)xb + (unsigned)cur.addr[0] * 4u, (unsigned)(size_t)(lvoid *)xb + (unsigned)cur.addr[1] * 4u};
    auto issue = [&](int q, float (&c)[2][4], float (&aw)[MT]) {
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        // (the pair's element offset is made opaque to the optimiser: it would otherwise fold the row offset into a SECOND base
        //  register per pair - the byte offset of the lower row exceeds ds_read2's 8-bit field only together with the pair's 3840 q -
        //  instead of `offset0:40 offset1:41` on one base)
        typedef __attribute__((address_space(3))) const float lds_cf;
        unsigned co = cbase[s] + q * (2 * CHS * 4);  // LDS byte address of the cell in this pair's first channel
        asm volatile("" : "+v"(co));
        lds_cf *cell = (lds_cf *)(size_t)co;
        c[s][0] = cell[0];
        c[s][1] = cell[1];
        c[s][2] = cell[IW];
        c[s][3] = cell[IW + 1];
      }
      const float *ap = wb + (2 * q + half) * MB + j * MT;  // slab rows are stored [j][m] (dcn_fused_pack_kernel): MT adjacent weights
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
    // four dependent FMAs per sample, written out: left to `-ffp-contract`, hipcc forms the four products first (v_mul / v_pk_mul) and
    // adds them up - 7 vector instructions instead of 4, and the vector instruction count next to the MFMAs is what this loop costs
    auto sample = [&](int s, const float (&c)[4]) -> float {
      return __builtin_fmaf(bw[s][3], c[3], __builtin_fmaf(bw[s][2], c[2], __builtin_fmaf(bw[s][1], c[1], bw[s][0] * c[0])));
    };
    if constexpr (SLOW) {
      // Fix-up pass after the regular one (in which the slow lanes carried zero weights): the lanes whose cell lies outside the
      // window gather their four corners from global memory with the full bounds logic (dcn_tap.h), every other lane contributes
      // 0.
      const int ti = t / 3, tj = t - 3 * ti;
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        if (!__any(slow >> s & 1u)) continue;
        const bool on = slow >> s & 1u;
        const int pv = on ? (p0 + s * a.W) * 4 : OOB;
        const float m = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(msk_rsrc, pv, (g * 9 + t) * P * 4, 0));
        const float hsp = (float)(oy0 + s - 1 + ti) + __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(off_rsrc, pv, (g * 18 + 2 * t) * P * 4, 0));
        const float wsp = (float)(ox - 1 + tj) + __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(off_rsrc, pv, (g * 18 + 2 * t + 1) * P * 4, 0));
        const Tap tq = resolve_tap(hsp, wsp, a.H, a.W);
        const float s0 = on ? tq.w00 * m : 0.f, s1 = on ? tq.w01 * m : 0.f, s2 = on ? tq.w10 * m : 0.f, s3 = on ? tq.w11 * m : 0.f;
        const int hp = half * P;
        const int o0 = on ? (tq.o00 + hp) * 4 : OOB, o1 = on ? (tq.o01 + hp) * 4 : OOB, o2 = on ? (tq.o10 + hp) * 4 : OOB, o3 = on ? (tq.o11 + hp) * 4 : OOB;
        // all 4 NQ corner gathers of the sub-tile in flight at once, then the products (round 6: it was a rolled loop of NQ dependent
        // gathers - fields that vary inside a tile, bench.py's `motion` leg, run this pass on most steps; dcn_tapwin_s.hip likewise)
        float gv[NQ][4];
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
          const int so = (g * CPG + 2 * q) * P * 4;
          gv[q][0] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(x_rsrc, o0, so, 0));
          gv[q][1] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(x_rsrc, o1, so, 0));
          gv[q][2] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(x_rsrc, o2, so, 0));
          gv[q][3] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(x_rsrc, o3, so, 0));
        }
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
          const float bv = s0 * gv[q][0] + s1 * gv[q][1] + s2 * gv[q][2] + s3 * gv[q][3];
          const float *ap = wb + (2 * q + half) * MB + j * MT;
#pragma unroll
          for (int m2 = 0; m2 < MT; ++m2) acc[s][m2] = __builtin_amdgcn_mfma_f32_32x32x2f32(ap[m2], bv, acc[s][m2], 0, 0, 0);
        }
      }
    } else {
      // Three-stage software pipeline over the channel pairs: stage q issues the LDS reads of pair q + 2, the bilinear arithmetic
      // of pair q + 1 and the 2 MT MFMAs of pair q.  (Left free, hipcc sinks every LDS read to just before its first use - `ds_read ...
      // s_waitcnt lgkmcnt(0) ... 7 VALU ... 4 MFMAs` per group; measured on four schedules, from that one to DS / MFMA / VALU groups
      // pinned instruction by instruction: 100.1 - 100.9 TF/s all of them - with two waves per SIMD one wave's gaps are the other's
      // issue slots.  profiles/r4/tapwin_sched_ab.log)
      float cv[3][2][4], av[3][MT];
      issue(0, cv[0], av[0]);
      if (NQ > 1) issue(1, cv[1], av[1]);
      float b0 = sample(0, cv[0][0]), b1 = sample(1, cv[0][1]);
#pragma unroll
      for (int q = 0; q < NQ; ++q) {
        const int cq = q % 3, n1 = (q + 1) % 3, n2 = (q + 2) % 3;
        if (q + 2 < NQ) issue(q + 2, cv[n2], av[n2]);
        float nb0 = 0.f, nb1 = 0.f;
        if (q + 1 < NQ) {
          nb0 = sample(0, cv[n1][0]);
          nb1 = sample(1, cv[n1][1]);
        }
#pragma unroll
        for (int m = 0; m < MT; ++m) acc[0][m] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[cq][m], b0, acc[0][m], 0, 0, 0);
#pragma unroll
        for (int m = 0; m < MT; ++m) acc[1][m] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[cq][m], b1, acc[1][m], 0, 0, 0);
        b0 = nb0;
        b1 = nb1;
        // the NEXT step's sampling state, from tap values that are already in registers: ~70 vector instructions with no memory
        // access and no dependence on this step, one sub-tile in each of the first two stages - under the MFMAs of the SIMD's other
        // wave they cost next to nothing; after the last MFMA of the step (first version) they were 9 % of the kernel
        // (profiles/r4/tapwin_ablations.log).  The empty asm statements DEFINE the values here: hipcc otherwise sinks the
        // arithmetic into the block after the fix-up branch, where they are first used.
        if (q < 2) {
          state_half(nxt, q, tn, ny0, nx0, tnv);
#pragma unroll
          for (int i = 0; i < 4; ++i) asm volatile("" : "+v"(nxt.bw[q][i]));
          asm volatile("" : "+v"(nxt.addr[q]), "+v"(nxt.slow));
        }
        if (q == 2) {
          state_any(nxt);
          asm volatile("" : "+s"(nxt.slow_any));
        }
        __builtin_amdgcn_sched_barrier(0);  // stage boundary: nothing moves across (finer pinning - DS / MFMA / VALU groups - measured: no difference)
      }
    }
  };

  // ---- schedule.  Step k computes on buffers k & 1.  At its top (after the barrier: nobody reads buffers (k + 1) & 1 any more, every
  //      wave's pieces of step k have landed - each wave waited for its own before the barrier) the tap values of step k + 1 are
  //      read from the wave's staging rows into registers, the DMA of step k + 1 (window, weights) and of the tap values of step k + 2
  //      is requested, and the state of step k + 1 is computed BETWEEN the MFMAs of step k.  One wait (vmcnt(0)) after the MFMAs.
  __syncthreads();  // shifts
  const int n_steps = a.dg * KK;
  St cur, nxt;
  float tnv[6];
  int o1y, o1x;
  origin(0, 0, o1y, o1x);
  dma_x(0, 0, o1y, o1x);
  dma_w(0, 0, 0);
  dma_taps(0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  read_taps(tnv);
  make_state(cur, 0, o1y, o1x, tnv);
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // (the staging rows are read before they are requested again)
  int g = 0, t = 0, g1 = 0, t1 = 1;  // (g1, t1) = step k + 1
  if (n_steps > 1) {
    origin(g1, t1, o1y, o1x);
    dma_taps(g1, t1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  for (int k = 0; k < n_steps; ++k) {
    __syncthreads();
    const int g2 = (t1 == KK - 1) ? g1 + 1 : g1, t2 = (t1 == KK - 1) ? 0 : t1 + 1;  // step k + 2
    int o2y = 0, o2x = 0;
    if (k + 1 < n_steps) {
      read_taps(tnv);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      dma_x((k + 1) & 1, g1, o1y, o1x);
      dma_w((k + 1) & 1, g1, t1);
      if (k + 2 < n_steps) {
        origin(g2, t2, o2y, o2x);
        dma_taps(g2, t2);
      }
    }
    const float *xb = xw + (k & 1) * XW, *wb = wsl + (k & 1) * WS;
    run_step(std::false_type{}, cur, xb, wb, g, t, nxt, t1, o1y, o1x, tnv);
    if (cur.slow_any) run_step(std::true_type{}, cur, xb, wb, g, t, nxt, t1, o1y, o1x, tnv);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's DMA pieces of step k + 1 and tap values of step k + 2
    cur = nxt;
    g = g1; t = t1; g1 = g2; t1 = t2; o1y = o2y; o1x = o2x;
  }

  // ---- epilogue: bias, activation, store.  Buffer instructions (wave-uniform resource + 32-bit lane offset + scalar channel offset):
  //      no 64-bit address arithmetic next to 128 live accumulators; dead lanes / channels past Co carry the out-of-range offset.
  const __amdgpu_buffer_rsrc_t y_rsrc = twn_rsrc(a.y + (int64_t)img * a.Co * P, a.Co * P * 4);
  const __amdgpu_buffer_rsrc_t b_rsrc = twn_rsrc(a.bias ? a.bias : a.x, a.bias ? a.Co * 4 : 0);
#pragma unroll
  for (int m = 0; m < MT; ++m) {
#pragma unroll
    for (int r4 = 0; r4 < 4; ++r4) {
      const int cob = co_blk + m * 32 + 8 * r4;  // this lane's rows: cob + 4 half + (0..3)
      float bias4[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const bool cok = cob + 4 * half + r < a.Co;
        bias4[r] = a.bias ? __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(b_rsrc, cok ? (4 * half + r) * 4 : OOB, cob * 4, 0)) : 0.f;
      }
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        const bool pix_ok = s ? ok1 : ok0;
        const int pv = (p0 + s * a.W + 4 * half * P) * 4;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float v = acc[s][m][4 * r4 + r] + bias4[r];
          if (a.act == EDVR_ACT_LRELU) v = v > 0.f ? v : 0.1f * v;
          else if (a.act == EDVR_ACT_RELU) v = fmaxf(v, 0.f);
          else if (a.act == EDVR_ACT_SIGMOID) v = __builtin_amdgcn_rcpf(1.f + __expf(-v));
          const bool cok = cob + 4 * half + r < a.Co;
          __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), y_rsrc, (pix_ok && cok) ? pv : OOB, (cob + r) * P * 4, 0);
        }
      }
    }
  }
}

bool dcn_tapwin_supported(int C, int Co, int H, int W, int kh, int kw, int stride, int pad, int dil, int groups, int dg) {
  if (!dcn_fused_supported(C, Co, H, W, kh, kw, stride, pad, dil, groups, dg)) return false;  // signature + 32-bit buffer-offset limits
  const int cpg = C / dg;
  return (cpg == 16 || cpg == 8) && dg <= TWN_MAX_DG && W % 4 == 0 && W >= 32 &&  // 16-byte aligned window rows
         (int64_t)Co * H * W * 4 < ((int64_t)1 << 31);                              // y through 32-bit buffer offsets
}

template <int MT>
static int launch_tapwin(DcnTapwinArgs a, int co_start, int co_blocks, hipStream_t stream) {
  a.co_start = co_start;
  dim3 grid(a.tiles_x * a.tiles_y, co_blocks, a.B);
  if (a.C / a.dg == 16) hipLaunchKernelGGL((dcn_tapwin_fwd_kernel<MT, 16>), grid, dim3(256), 0, stream, a);
  else hipLaunchKernelGGL((dcn_tapwin_fwd_kernel<MT, 8>), grid, dim3(256), 0, stream, a);
  return check_launch("dcn_tapwin_fwd_kernel");
}

// wpk: the layout of dcn_fused_pack (dcn_fused.hip): [c][tap][co'] with the output channels of a launch block reordered [j][m]
int dcn_tapwin_forward(const float *x, const float *offset, const float *mask, const float *wpk, const float *bias, float *y, int B, int C,
                       int H, int W, int Co, int dg, int64_t off_bs, int64_t msk_bs, int act, hipStream_t stream) {
  EDVR_REQUIRE(B <= 65535, "dcn_tapwin: batch %d exceeds grid.z", B);
  EDVR_REQUIRE((reinterpret_cast<uintptr_t>(x) & 15) == 0, "dcn_tapwin: x must be 16-byte aligned");
  DcnTapwinArgs a;
  a.x = x; a.offset = offset; a.mask = mask; a.wpk = wpk; a.bias = bias; a.y = y;
  a.B = B; a.C = C; a.H = H; a.W = W; a.Co = Co; a.dg = dg; a.act = act;
  a.cop = (Co + 31) / 32 * 32;
  a.off_bs = off_bs; a.msk_bs = msk_bs;
  a.tiles_x = cdiv(W, TWN_TW);
  a.tiles_y = cdiv(H, TWN_TH);
  a.co_start = 0;
  const int full = Co / 128, rem_tiles = cdiv(Co - full * 128, 32);
  int rc = EDVR_OK;
  if (full > 0) rc = launch_tapwin<4>(a, 0, full, stream);
  if (rc || rem_tiles == 0) return rc;
  switch (rem_tiles) {
    case 1: return launch_tapwin<1>(a, full * 128, 1, stream);
    case 2: return launch_tapwin<2>(a, full * 128, 1, stream);
    case 3: return launch_tapwin<3>(a, full * 128, 1, stream);
    default: return launch_tapwin<4>(a, full * 128, 1, stream);
  }
}

}  // namespace edvr
