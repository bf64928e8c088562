// dcn_tapwin_s.hip - the tap-window DCNv2 forward (dcn_tapwin.hip) with SPLIT fp32 operands on the f16 matrix pipe (gfx950).
//
// Same walk, same windows, same fix-up path as dcn_tapwin.hip: (deformable group g) -> (tap t), one step = 16 (8) channels x 2 pixel
// rows x MT x 32 output channels, the window of each (g, t) fetched by LDS-DMA around the tap's displacement.  What changes is the
// arithmetic of D[co, pixel] += W[co, (c, t)] * col[(c, t), pixel] (deform_conv_cuda.cpp:550-555): both operands travel as f16
// (hi, lo) pairs (winograd_f4s.hip) and all four cross products are accumulated in fp32 by v_mfma_f32_32x32x16_f16.
//   * weights: packed once per call as [channel quad][tap][co'][4 channels] dwords (hi | lo << 16) of w * s_W behind a 64-byte header
//     (s_W from max |w|): a lane's A operand - four channels x (hi, lo) of its output channel - is ONE ds_read_b128 of the slab;
//   * columns: the lane samples the FOUR channels 8 kg + 2 i + half of a K-group for its pixel (bilinear weights x mask x s_X, four
//     FMAs per sample as before), splits them (2 instructions each) -> the B operand; the second MFMA of a pair takes B rotated by
//     16 bits ((lo, hi): the cross terms).  s_X from `xm_amax`, an upper bound of max |x| * max(1, max |mask|).  (Interleaved, not
//     4 half + i: the half-waves of a gather are then ONE channel = 480 floats = 32 banks apart and use disjoint halves of the 64
//     LDS banks; four channels apart (0 mod 64) every cell read was a two-way conflict - 54 % of the LDS-active cycles.)
// Per step and wave 32 MFMAs of 32 cycles where the fp32 kernel issues 64 of 64: a quarter of the matrix-pipe time.
#include <type_traits>

#include "common.h"
#include <cstdlib>

#include "dcn_tap.h"
#include "pack.h"

namespace edvr {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

namespace {
constexpr int TWS_TH = 8, TWS_TW = 32, TWS_RY = 2, TWS_RX = 2;
constexpr int TWS_IH = TWS_TH + 2 * TWS_RY;               // 12 rows: fl(dy) - s in [-RY, RY - 1], plus the cell's lower row
constexpr int TWS_IW = (TWS_TW + 2 * TWS_RX + 3 + 3) / 4 * 4;  // 40 columns: 36 needed + up to 3 lost to the 16-byte alignment
constexpr int TWS_PPR = TWS_IW / 4;                       // 10 16-byte pieces per row
constexpr int TWS_HR = TWS_IH / 2;                        // one DMA instruction moves 6 rows x 10 pieces (lanes 0-59) of one channel
constexpr int TWS_HB = TWS_HR * TWS_IW;                   // ... to floats [0, 240) / [240, 480) of the channel: the rows stay dense
constexpr int TWS_CHS = TWS_IH * TWS_IW;                  // 480 floats per channel = 32 (mod 64): the two half-waves of a gather (channels
                                                          // c / c + 1, same position) fall on disjoint halves of the 64 LDS banks
static_assert(TWS_HR * TWS_PPR <= 64 && TWS_IH == 2 * TWS_HR && TWS_CHS % 64 == 32, "window halves must fit one wave instruction");
constexpr int TWS_MAX_DG = 8;   // (the shift table and the tap staging area share the last 3.6 KB of the workgroup's 80 KB)
constexpr int TWS_OOB = (int)0x80000000;
constexpr int TWS_RSRC_FLAGS = 0x00020000;
}  // namespace

struct DcnTapwinSArgs {
  const float *xm_amax;  // device: >= max |x| * max(1, max |mask|)
  const float *x, *offset, *mask;
  const unsigned *wpk;   // header (16 dwords: s_W, 1 / s_W) + [channel quad][tap][co'][4] dwords
  const float *bias;
  float *y;
  int B, C, H, W, Co, dg, cop, act, co_start, tiles_x, tiles_y;
  int64_t off_bs, msk_bs;
};

typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2t __attribute__((ext_vector_type(2)));

// LDS-DMA from inline assembly: lane l of the wave delivers its 16 bytes to LDS byte address `lds` + 16 l.  The compiler treats the
// builtin form as a store to LDS that any later LDS read may alias and puts `s_waitcnt vmcnt(0)` in front of the first ds_read
// after it - i.e. the fetch of step k + 1 would be WAITED FOR at the top of step k instead of overlapping its 64 MFMAs (that wait is
// in dcn_fused.hip's code, at the top of every half chunk).  Issued this way the request is invisible to the wait-count pass; the
// kernel waits for it itself (`s_waitcnt vmcnt(0)` after the MFMAs of the step) and the barrier orders it against the readers.
// (m0 = LDS base; one wait state between the scalar write of m0 and the instruction that uses it)
static __device__ __forceinline__ void tws_dma16(i32x4 rsrc, unsigned lds, int voff, int soff) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" ::"s"(lds), "v"(voff), "s"(rsrc), "s"(soff) : "memory", "m0");
}

static __device__ __forceinline__ void tws_dma4(i32x4 rsrc, unsigned lds, int voff, int soff) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dword %1, %2, %3 offen lds" ::"s"(lds), "v"(voff), "s"(rsrc), "s"(soff) : "memory", "m0");
}

static __device__ __forceinline__ i32x4 tws_rsrc4(const void *ptr, int bytes) {
  const uint64_t pv = reinterpret_cast<uint64_t>(ptr);
  i32x4 r;
  r[0] = __builtin_amdgcn_readfirstlane((int)(uint32_t)pv);
  r[1] = __builtin_amdgcn_readfirstlane((int)(pv >> 32)) & 0xffff;
  r[2] = bytes;
  r[3] = TWS_RSRC_FLAGS;
  return r;
}

static __device__ __forceinline__ __amdgpu_buffer_rsrc_t tws_rsrc(const void *ptr, int bytes) {
  const uint64_t pv = reinterpret_cast<uint64_t>(ptr);
  const uint64_t pu = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(pv >> 32)) << 32) |
                      (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)pv);
  return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void *>(pu), (short)0, bytes, TWS_RSRC_FLAGS);
}

// v = hi + lo in f16 (the operand scale is already in v): v_fma_mixlo_f16 + v_fma_mixhi_f16, four values with the independent halves first
static __device__ __forceinline__ void tws_split4(const float (&x)[4], unsigned (&o)[4]) {
  asm volatile(
      "v_fma_mixlo_f16 %0, %4, 1.0, 0\n\tv_fma_mixlo_f16 %1, %5, 1.0, 0\n\tv_fma_mixlo_f16 %2, %6, 1.0, 0\n\tv_fma_mixlo_f16 %3, %7, 1.0, 0\n\t"
      "v_fma_mixhi_f16 %0, %4, 1.0, -%0 op_sel_hi:[0,0,1]\n\tv_fma_mixhi_f16 %1, %5, 1.0, -%1 op_sel_hi:[0,0,1]\n\t"
      "v_fma_mixhi_f16 %2, %6, 1.0, -%2 op_sel_hi:[0,0,1]\n\tv_fma_mixhi_f16 %3, %7, 1.0, -%3 op_sel_hi:[0,0,1]"
      : "=&v"(o[0]), "=&v"(o[1]), "=&v"(o[2]), "=&v"(o[3])
      : "v"(x[0]), "v"(x[1]), "v"(x[2]), "v"(x[3]));
}

template <int MT, int CPG>
__global__ __launch_bounds__(256, 2) void dcn_tapwin_split_fwd_kernel(const DcnTapwinSArgs a) {
  constexpr int TH = TWS_TH, TW = TWS_TW, IH = TWS_IH, IW = TWS_IW, CHS = TWS_CHS, KK = 9, OOB = TWS_OOB;
  constexpr int MB = 32 * MT, NQ = CPG / 2, KG = CPG / 8;
  constexpr int XW = CPG * CHS;            // floats of one window buffer (16 channels: 7680 = 30 KB)
  constexpr int WS = CPG * MB;             // floats of one weight slab (16 x 128: 8 KB)
  constexpr int CPW = CPG / 4;             // channels whose window this wave fetches (2 instructions each)
  static_assert(CPG % 4 == 0, "four waves share the channels of a group");
  __shared__ __attribute__((aligned(16))) float xw[2 * XW];
  __shared__ __attribute__((aligned(16))) unsigned wsl[2 * WS];  // [channel quad CPG / 4][co' MB][4 channels], dword = (hi | lo << 16)
  __shared__ int shifts[TWS_MAX_DG * 18];
  __shared__ float tpl[4 * 3 * 64];  // offsets / mask of the NEXT step's tap: [wave][dy | dx | mask][pixel row 2][column 32], wave-private

  const int tid = threadIdx.x, lane = tid & 63, half = lane >> 5, j = lane & 31;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // s_X = 2^e with amax * 2^e < 2^15 (amax = m 2^k, m in [1, 2) -> e = 14 - k); s_W comes with the packed weights
  const unsigned amax_bits = (unsigned)__builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, *a.xm_amax));
  const float s_x = __builtin_bit_cast(float, (unsigned)min(max(127 + 14 - ((int)((amax_bits >> 23) & 255u) - 127), 7), 220) << 23);
  const float unscale = (1.f / s_x) * __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane((int)a.wpk[1]));
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
  const int d_r = lane / TWS_PPR, d_c = 4 * (lane - d_r * TWS_PPR);
  const int rel0 = (d_r * a.W + d_c) * 4;
  const __amdgpu_buffer_rsrc_t x_rsrc = tws_rsrc(x_img, a.C * P * 4);  // (fix-up pass)
  const i32x4 x_rsrc4 = tws_rsrc4(x_img, a.C * P * 4), w_rsrc4 = tws_rsrc4(a.wpk, 64 + a.C * KK * a.cop * 4);
  typedef __attribute__((address_space(3))) void lvoid;
  const unsigned xw_lds = (unsigned)(size_t)(lvoid *)xw, ws_lds = (unsigned)(size_t)(lvoid *)wsl;
  const __amdgpu_buffer_rsrc_t off_rsrc = tws_rsrc(off_b, a.dg * 18 * P * 4), msk_rsrc = tws_rsrc(msk_b, a.dg * 9 * P * 4);
  auto dma_x = [&](int buf, int g, int wy0, int wx0) {
    const int sh = (wy0 * a.W + wx0) * 4;
    const bool colok = (unsigned)(wx0 + d_c) < (unsigned)a.W;
    const int vo0 = (colok && (unsigned)(wy0 + d_r) < (unsigned)a.H) ? rel0 + sh : OOB;
    const int vo1 = (colok && (unsigned)(wy0 + d_r + TWS_HR) < (unsigned)a.H) ? rel0 + sh + TWS_HR * a.W * 4 : OOB;
    if (lane < TWS_HR * TWS_PPR) {
#pragma unroll
      for (int cc = 0; cc < CPW; ++cc) {
        const int ch = wave * CPW + cc;
        const unsigned dst = xw_lds + (buf * XW + ch * CHS) * 4;
        tws_dma16(x_rsrc4, dst, vo0, (g * CPG + ch) * P * 4);
        tws_dma16(x_rsrc4, dst + TWS_HB * 4, vo1, (g * CPG + ch) * P * 4);
      }
    }
  };
  // ---- weight slab of (g, t): rows (channel quad (CPG / 4) g + r, tap t) of the packed weights, MB x 4 dwords each, dense [r][MB][4] in
  //      LDS: piece q = 64 i + lane of the slab is (row q / MB, 16-byte column q % MB); wave w issues instructions i = w, w + 4, ...
  constexpr int WTOTAL = (CPG / 4) * MB, WNI = WTOTAL / 64, WNK = (WNI + 3) / 4;
  static_assert(WTOTAL % 64 == 0, "weight slab must fill whole wave instructions");
  int wvo[WNK];
#pragma unroll
  for (int k = 0; k < WNK; ++k) {
    const int q = (wave + 4 * k) * 64 + lane, row = q / MB, c16 = q - row * MB;
    wvo[k] = (row * KK * a.cop + c16) * 16;
  }
  auto dma_w = [&](int buf, int g, int t) {
    const int so = 64 + ((g * (CPG / 4) * KK + t) * a.cop + co_blk) * 16;
#pragma unroll
    for (int k = 0; k < WNK; ++k)
      if (wave + 4 * k < WNI) tws_dma16(w_rsrc4, ws_lds + (buf * WS + (wave + 4 * k) * 256) * 4, wvo[k], so);
  };
  // window origin of step (g, t): rows  ty0 - 1 + ti + sy - RY ..,  columns from the multiple of 4 at or below  tx0 - 1 + tj + sx - RX
  auto origin = [&](int g, int t, int &wy0, int &wx0) {
    const int sy = __builtin_amdgcn_readfirstlane(shifts[(g * KK + t) * 2]);
    const int sx = __builtin_amdgcn_readfirstlane(shifts[(g * KK + t) * 2 + 1]);
    const int ti = t / 3, tj = t - 3 * ti;
    wy0 = ty0 - 1 + ti + sy - TWS_RY;
    wx0 = (tx0 - 1 + tj + sx - TWS_RX) & ~3;
  };
  // offsets / mask of a tap for this wave's 64 pixels -> the wave's private staging rows, by LDS-DMA (lane = (pixel row, column)),
  // requested TWO steps ahead; a lane reads the six values of its two pixels (both half-waves read the same words: broadcast)
  const i32x4 off_rsrc4 = tws_rsrc4(off_b, a.dg * 18 * P * 4), msk_rsrc4 = tws_rsrc4(msk_b, a.dg * 9 * P * 4);
  const unsigned tpl_lds = (unsigned)(size_t)(lvoid *)tpl + wave * (3 * 64 * 4);
  const int tvd = half ? tvo[1] : tvo[0];  // (here the lane's upper bit selects the pixel ROW, not the channel parity)
  auto dma_taps = [&](int g, int t) {
    tws_dma4(off_rsrc4, tpl_lds, tvd, (g * 18 + 2 * t) * P * 4);
    tws_dma4(off_rsrc4, tpl_lds + 256, tvd, (g * 18 + 2 * t + 1) * P * 4);
    tws_dma4(msk_rsrc4, tpl_lds + 512, tvd, (g * 9 + t) * P * 4);
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
    const float mm = inside ? m * s_x : 0.f;  // (a valid tap outside the window: zero here, added by the fix-up pass); the operand scale rides on the mask
    const float hm = (1.f - lh) * mm, lm = lh * mm, hw = 1.f - lw;
    st.bw[s][0] = hm * hw;
    st.bw[s][1] = hm * lw;
    st.bw[s][2] = lm * hw;
    st.bw[s][3] = lm * lw;
    st.addr[s] = half * CHS + (inside ? ry * IW + rx : 0);  // this lane samples channels 8 kg + 2 (0..3) + half
    st.slow = (s ? st.slow : 0u) | ((valid && !inside) ? 1u << s : 0u);
  };
  auto state_any = [&](St &st) { st.slow_any = __builtin_amdgcn_readfirstlane(__any(st.slow != 0) ? 1 : 0); };
  auto make_state = [&](St &st, int t, int wy0, int wx0, const float (&d)[6]) {
    state_half(st, 0, t, wy0, wx0, d);
    state_half(st, 1, t, wy0, wx0, d);
    state_any(st);
  };

  // ---- one step: KG K-groups of 8 channels x 2 sub-tiles x MT x 2 MFMAs.
  auto run_step = [&](auto SLOWT, const St &cur, const float *xb, const unsigned *wb, int g, int t, St &nxt, int tn, int ny0, int nx0,
                      const float (&tnv)[6]) {
    constexpr bool SLOW = decltype(SLOWT)::value;
    const float (&bw)[2][4] = cur.bw;
    const unsigned slow = cur.slow;
    const unsigned cbase[2] = {(unsigned)(size_t)(lvoid *)xb + (unsigned)cur.addr[0] * 4u, (unsigned)(size_t)(lvoid *)xb + (unsigned)cur.addr[1] * 4u};
    auto issue_cells = [&](int kg, float (&c)[2][4][4]) {  // the 2x2 cells of the lane's four channels of K-group kg, both sub-tiles
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        typedef __attribute__((address_space(3))) const float lds_cf;
        unsigned co = cbase[s];
        asm volatile("" : "+v"(co));  // (one base register per sub-tile: the channel offsets stay immediates)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          lds_cf *cell = (lds_cf *)(size_t)(co + (kg * 8 + 2 * i) * (CHS * 4));
          c[s][i][0] = cell[0];
          c[s][i][1] = cell[1];
          c[s][i][2] = cell[IW];
          c[s][i][3] = cell[IW + 1];
        }
      }
    };
    // four dependent FMAs per sample, written out (dcn_tapwin.hip)
    auto sample = [&](int s, const float (&c)[4]) -> float {
      return __builtin_fmaf(bw[s][3], c[3], __builtin_fmaf(bw[s][2], c[2], __builtin_fmaf(bw[s][1], c[1], bw[s][0] * c[0])));
    };
    if constexpr (SLOW) {
      // Fix-up pass after the regular one (in which the slow lanes carried zero weights): the lanes whose cell lies outside the
      // window gather their corners from global memory with the full bounds logic (dcn_tap.h), every other lane contributes 0.
      // The SAME arithmetic as the regular pass - four channels per lane and K-group, split, two f16 MFMAs per output block - with
      // all 16 KG corner loads of a sub-tile in flight at once: one memory latency + 4 KG MT matrix instructions of 32 cycles per
      // sub-tile (it was a rolled loop of CPG / 2 dependent gathers on the fp32 matrix instruction: ~5x a regular step, which
      // fields that vary inside a tile - object boundaries, bench.py's `motion` leg - pay on most steps).
      const int ti = t / 3, tj = t - 3 * ti;
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        if (!__any(slow >> s & 1u)) continue;
        const bool on = slow >> s & 1u;
        const int pv = on ? (p0 + s * a.W) * 4 : OOB;
        const float m = s_x * __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(msk_rsrc, pv, (g * 9 + t) * P * 4, 0));
        const float hsp = (float)(oy0 + s - 1 + ti) + __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(off_rsrc, pv, (g * 18 + 2 * t) * P * 4, 0));
        const float wsp = (float)(ox - 1 + tj) + __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(off_rsrc, pv, (g * 18 + 2 * t + 1) * P * 4, 0));
        const Tap tq = resolve_tap(hsp, wsp, a.H, a.W);
        const float sw[4] = {on ? tq.w00 * m : 0.f, on ? tq.w01 * m : 0.f, on ? tq.w10 * m : 0.f, on ? tq.w11 * m : 0.f};
        const int hp = half * P;
        const int ov[4] = {on ? (tq.o00 + hp) * 4 : OOB, on ? (tq.o01 + hp) * 4 : OOB, on ? (tq.o10 + hp) * 4 : OOB, on ? (tq.o11 + hp) * 4 : OOB};
        float gv[KG][4][4];  // [K-group][channel 8 kg + 2 i + half][corner]
#pragma unroll
        for (int kg = 0; kg < KG; ++kg)
#pragma unroll
          for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int c = 0; c < 4; ++c)
              gv[kg][i][c] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(x_rsrc, ov[c], (g * CPG + 8 * kg + 2 * i) * P * 4, 0));
#pragma unroll
        for (int kg = 0; kg < KG; ++kg) {
          i32x4 aw[MT];
#pragma unroll
          for (int m2 = 0; m2 < MT; ++m2) aw[m2] = *reinterpret_cast<const i32x4 *>(wb + ((kg * 2 + half) * MB + m2 * 32 + j) * 4);
          float v[4];
#pragma unroll
          for (int i = 0; i < 4; ++i)
            v[i] = __builtin_fmaf(sw[3], gv[kg][i][3], __builtin_fmaf(sw[2], gv[kg][i][2], __builtin_fmaf(sw[1], gv[kg][i][1], sw[0] * gv[kg][i][0])));
          unsigned pk[4];
          tws_split4(v, pk);
          i32x4 bq = i32x4{(int)pk[0], (int)pk[1], (int)pk[2], (int)pk[3]};
#pragma unroll
          for (int m2 = 0; m2 < MT; ++m2)
            acc[s][m2] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, aw[m2]), __builtin_bit_cast(f16x8, bq), acc[s][m2], 0, 0, 0);
          asm volatile("v_alignbit_b32 %0, %0, %0, 16\n\tv_alignbit_b32 %1, %1, %1, 16\n\tv_alignbit_b32 %2, %2, %2, 16\n\tv_alignbit_b32 %3, %3, %3, 16"
                       : "+v"(bq[0]), "+v"(bq[1]), "+v"(bq[2]), "+v"(bq[3]));
#pragma unroll
          for (int m2 = 0; m2 < MT; ++m2)
            acc[s][m2] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, aw[m2]), __builtin_bit_cast(f16x8, bq), acc[s][m2], 0, 0, 0);
        }
      }
    } else {
      // Per K-group: the A operands of the group (MT ds_read_b128), the 2 x 4 samples from the cells read during the previous group's
      // MFMAs, their split; the next group's cells are requested before this group's 4 MT MFMAs and land under them.
      float cv[2][4][4];
      issue_cells(0, cv);
#pragma unroll
      for (int kg = 0; kg < KG; ++kg) {
        i32x4 aw[MT];
#pragma unroll
        for (int m = 0; m < MT; ++m) aw[m] = *reinterpret_cast<const i32x4 *>(wb + ((kg * 2 + half) * MB + m * 32 + j) * 4);  // lane stride 16 bytes: conflict-free
        i32x4 bq[2];
#pragma unroll
        for (int s = 0; s < 2; ++s) {
          const float v[4] = {sample(s, cv[s][0]), sample(s, cv[s][1]), sample(s, cv[s][2]), sample(s, cv[s][3])};
          unsigned pk[4];
          tws_split4(v, pk);
          bq[s] = i32x4{(int)pk[0], (int)pk[1], (int)pk[2], (int)pk[3]};
        }
        if (kg + 1 < KG) issue_cells(kg + 1, cv);
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
          for (int m = 0; m < MT; ++m)
            acc[s][m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, aw[m]), __builtin_bit_cast(f16x8, bq[s]), acc[s][m], 0, 0, 0);
        // the NEXT step's sampling state from tap values already in registers (dcn_tapwin.hip): one sub-tile per K-group, between the
        // two MFMA batches; the empty asm statements DEFINE the values here
        if (kg == 0 || KG == 1) {
          state_half(nxt, 0, tn, ny0, nx0, tnv);
#pragma unroll
          for (int i = 0; i < 4; ++i) asm volatile("" : "+v"(nxt.bw[0][i]));
          asm volatile("" : "+v"(nxt.addr[0]), "+v"(nxt.slow));
        }
        if (kg == KG - 1) {
          state_half(nxt, 1, tn, ny0, nx0, tnv);
#pragma unroll
          for (int i = 0; i < 4; ++i) asm volatile("" : "+v"(nxt.bw[1][i]));
          asm volatile("" : "+v"(nxt.addr[1]), "+v"(nxt.slow));
          state_any(nxt);
          asm volatile("" : "+s"(nxt.slow_any));
        }
        // (hi, lo) -> (lo, hi) in place: the two cross terms
#pragma unroll
        for (int s = 0; s < 2; ++s)
          asm volatile("v_alignbit_b32 %0, %0, %0, 16\n\tv_alignbit_b32 %1, %1, %1, 16\n\tv_alignbit_b32 %2, %2, %2, 16\n\tv_alignbit_b32 %3, %3, %3, 16"
                       : "+v"(bq[s][0]), "+v"(bq[s][1]), "+v"(bq[s][2]), "+v"(bq[s][3]));
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
          for (int m = 0; m < MT; ++m)
            acc[s][m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, aw[m]), __builtin_bit_cast(f16x8, bq[s]), acc[s][m], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
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
    const float *xb = xw + (k & 1) * XW;
    const unsigned *wb = wsl + (k & 1) * WS;
    run_step(std::false_type{}, cur, xb, wb, g, t, nxt, t1, o1y, o1x, tnv);
    if (cur.slow_any) run_step(std::true_type{}, cur, xb, wb, g, t, nxt, t1, o1y, o1x, tnv);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's DMA pieces of step k + 1 and tap values of step k + 2
    cur = nxt;
    g = g1; t = t1; g1 = g2; t1 = t2; o1y = o2y; o1x = o2x;
  }

  // ---- epilogue: bias, activation, store.  Buffer instructions (wave-uniform resource + 32-bit lane offset + scalar channel offset):
  //      no 64-bit address arithmetic next to 128 live accumulators; dead lanes / channels past Co carry the out-of-range offset.
  const __amdgpu_buffer_rsrc_t y_rsrc = tws_rsrc(a.y + (int64_t)img * a.Co * P, a.Co * P * 4);
  const __amdgpu_buffer_rsrc_t b_rsrc = tws_rsrc(a.bias ? a.bias : a.x, a.bias ? a.Co * 4 : 0);
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
          float v = __builtin_fmaf(acc[s][m][4 * r4 + r], unscale, bias4[r]);
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

// header[0] = s_W = 2^e with max|w| s_W in [2^14, 2^15), header[1] = 1 / s_W
__global__ __launch_bounds__(1024) void dcn_tapwin_split_scale_kernel(const float *__restrict__ w, unsigned *__restrict__ wpk, int64_t total) {
  __shared__ float red[16];
  float m = 0.f;
  for (int64_t i = threadIdx.x; i < total; i += 1024) m = fmaxf(m, fabsf(w[i]));
#pragma unroll
  for (int sh = 32; sh > 0; sh >>= 1) m = fmaxf(m, __shfl_xor(m, sh));
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int i = 1; i < 16; ++i) m = fmaxf(m, red[i]);
    const unsigned field = f4s_weight_scale_field(__builtin_bit_cast(unsigned, m));
    wpk[0] = field << 23;
    wpk[1] = (254u - field) << 23;
    for (int i = 2; i < 16; ++i) wpk[i] = 0u;
  }
}

// [channel quad][tap][co][4 channels] (the fp32 kernel's slab has the output channels of a launch block reordered [j][m] for its
// scalar operand reads; here a lane's A operand is one ds_read_b128 and that order put neighbouring lanes 64 bytes apart: a 4-way
// bank conflict on every operand read, 54 % of the LDS-active cycles - plain channel order is conflict-free)
__global__ void dcn_tapwin_split_pack_kernel(const float *__restrict__ w, unsigned *__restrict__ wpk, int Co, int C, int cop) {
  const float s_w = __builtin_bit_cast(float, wpk[0]);
  const int64_t total = (int64_t)(C / 4) * 9 * cop;
  const int full = Co / 128, rem_tiles = (Co - full * 128 + 31) / 32;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int pos = (int)(i % cop), t = (int)((i / cop) % 9), c4 = (int)(i / ((int64_t)cop * 9));
    const int blk = pos / 128 < full ? pos / 128 : full, start = blk * 128, mt = blk < full ? 4 : rem_tiles;
    const int r = pos - start, co = pos;  // output channels in their own order inside a launch block: [m][j], a lane's operand 16 bytes from its neighbour's
    const bool live = mt > 0 && r < 32 * mt && co < Co;
#pragma unroll
    for (int q = 0; q < 4; ++q) wpk[16 + i * 4 + q] = live ? split_f16x2(w[((int64_t)co * C + 8 * (c4 >> 1) + 2 * q + (c4 & 1)) * 9 + t], s_w) : 0u;  // quad 2 kg + half, slot q = channel 8 kg + 2 q + half
  }
}

template <int MT>
static int launch_tapwin_s(DcnTapwinSArgs a, int co_start, int co_blocks, hipStream_t stream) {
  a.co_start = co_start;
  dim3 grid(a.tiles_x * a.tiles_y, co_blocks, a.B);
  if (a.C / a.dg == 16) hipLaunchKernelGGL((dcn_tapwin_split_fwd_kernel<MT, 16>), grid, dim3(256), 0, stream, a);
  else hipLaunchKernelGGL((dcn_tapwin_split_fwd_kernel<MT, 8>), grid, dim3(256), 0, stream, a);
  return check_launch("dcn_tapwin_split_fwd_kernel");
}

bool dcn_tapwin_split_enabled() {
  static const bool on = []() {
    const char *e = getenv("EDVR_DCN_SPLIT");  // "0": the fp32 tap-window kernel instead
    return !(e && e[0] == '0');
  }();
  return on;
}

// wpk: workspace of 16 + C * 9 * cop dwords, filled here (scale pre-pass + packing: two small launches per call, like dcn_fused_pack)
int dcn_tapwin_split_forward(const float *x, const float *offset, const float *mask, const float *weight, unsigned *wpk, const float *bias, float *y,
                             int B, int C, int H, int W, int Co, int dg, int64_t off_bs, int64_t msk_bs, int act, const float *xm_amax,
                             hipStream_t stream) {
  EDVR_REQUIRE(B <= 65535, "dcn_tapwin_split: batch %d exceeds grid.z", B);
  EDVR_REQUIRE((reinterpret_cast<uintptr_t>(x) & 15) == 0 && (reinterpret_cast<uintptr_t>(wpk) & 15) == 0, "dcn_tapwin_split: x / wpk must be 16-byte aligned");
  const int cop = (Co + 31) / 32 * 32;
  hipLaunchKernelGGL(dcn_tapwin_split_scale_kernel, dim3(1), dim3(1024), 0, stream, weight, wpk, (int64_t)Co * C * 9);
  const int64_t total = (int64_t)(C / 4) * 9 * cop;
  hipLaunchKernelGGL(dcn_tapwin_split_pack_kernel, dim3((unsigned)std::min<int64_t>(cdiv64(total, 256), 2048)), dim3(256), 0, stream, weight, wpk, Co, C, cop);
  DcnTapwinSArgs a;
  a.xm_amax = xm_amax;
  a.x = x; a.offset = offset; a.mask = mask; a.wpk = wpk; a.bias = bias; a.y = y;
  a.B = B; a.C = C; a.H = H; a.W = W; a.Co = Co; a.dg = dg; a.act = act;
  a.cop = cop;
  a.off_bs = off_bs; a.msk_bs = msk_bs;
  a.tiles_x = cdiv(W, TWS_TW);
  a.tiles_y = cdiv(H, TWS_TH);
  a.co_start = 0;
  const int full = Co / 128, rem_tiles = cdiv(Co - full * 128, 32);
  int rc = EDVR_OK;
  if (full > 0) rc = launch_tapwin_s<4>(a, 0, full, stream);
  if (rc || rem_tiles == 0) return rc;
  switch (rem_tiles) {
    case 1: return launch_tapwin_s<1>(a, full * 128, 1, stream);
    case 2: return launch_tapwin_s<2>(a, full * 128, 1, stream);
    case 3: return launch_tapwin_s<3>(a, full * 128, 1, stream);
    default: return launch_tapwin_s<4>(a, full * 128, 1, stream);
  }
}

}  // namespace edvr
