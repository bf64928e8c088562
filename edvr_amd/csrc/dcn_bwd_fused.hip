// dcn_bwd_fused.hip - DCNv2 backward for the EDVR signature WITHOUT the dcol round trip (gfx950).
//
// The reference (deform_conv_cuda.cpp:623-657) and the staged path of dcn.hip compute dcol = W^T dY into a column-sized buffer
// (9 C P floats per image: 3 GB on the 160 x 128 x 64 x 64 training layer), then read it back twice - once for dX (col2im,
// .cu:636-694), once for dOffset / dMask (col2im_coord, .cu:696-767).  Here a workgroup owns (image, 8 x 32 output pixels) and
// walks the deformable groups; per (group, tap pair) it
//   1. runs the 32-row slice of the GEMM on the matrix core: D[row, pixel] = sum_co W^T[row, co] dY[co, pixel] with the dY tile
//      REGISTER-RESIDENT as the B operand (64 registers per lane: the whole 128 x 32 tile of the wave's pixel row, loaded once
//      per workgroup) and the W^T slice streamed through a double-buffered LDS slab by LDS-DMA.  The 32 rows are
//      (tap of the pair, 16 channels of the group) ordered so that lane (pixel, half) ends up holding the 16 channel values of
//      ITS tap for ITS pixel in its 16 accumulator registers: no cross-lane traffic between the GEMM and its consumer;
//   2. consumes those 16 values in registers: the group's x window sits in LDS (LDS-DMA, zero outside the image by the buffer
//      range check), each channel costs 4 LDS reads for the bilinear cell, the arithmetic of modulated_deformable_col2im_coord
//      (dOffset, dMask), one coalesced store of the forward column value the dW GEMM needs, and the dX scatter.
// dX without atomics (LDS float atomics run at ~4 cycles per LANE on this chip: a first version of this kernel with ds_add_f32
// took 18 ms where the arithmetic needs 3): for a sub-pixel offset (floor in {-1, 0} on both axes: every tap of a fresh or
// lightly trained conv_offset) the 2 x 2 bilinear cell lies inside the STATIC 3 x 3 block around the tap's regular position and
// the four weights factor into 3 row x 3 column weights (2 non-zero each), as in dcn_bwd_dx_strip_kernel.  Lane (pixel j, tap)
// then adds to cells (row oy + ti - 1 + a, column ox + tj - 1 + b) of a WAVE-PRIVATE accumulator (5 rows x 36 columns x 16
// channels in LDS) with plain read-add-write: in one instruction all lanes of a half-wave hit different columns, and the two
// half-waves hit different rows - the tap pairs (0,3) (1,4) (2,5) share tj and differ by one in ti; the row they have in common
// is exchanged between the half-waves (one ds_bpermute per channel) so that each half owns two rows.  The pair (6,7) shares ti:
// its half-waves take turns.  Column passes b = -1, 0, +1 are separate instructions (in-order LDS), which resolves the overlap
// between neighbouring lanes.  Per group the eight private accumulators are summed and flushed with one global atomic per
// touched element.  Any other valid tap (|offset| >= 1 somewhere) takes a per-lane path through global memory with the full
// bounds logic and device atomics: the kernel is meant for EDVR_DCN_SCATTER_STRIP layers, correct for all.
#include <cstdlib>
#include <type_traits>

#include "common.h"
#include "dcn_tap.h"

namespace edvr {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {
#ifndef BF_TH_ROWS
#define BF_TH_ROWS 8
#endif
constexpr int BF_TH = BF_TH_ROWS, BF_NT = 64 * BF_TH, BF_TW = 32, BF_IH = BF_TH + 6, BF_IW = BF_TW + 6, BF_CHS = BF_IH * BF_IW;  // x window: the tile + 3 on every side
constexpr int BF_CPG = 16, BF_NS = 64, BF_LS = BF_NS + 4, BF_SLAB = 32 * 2 * BF_LS, BF_TP = 5;  // slab: [row 32][co parity 2][co pair 64 + 4 pad]
constexpr int BF_PR = 5, BF_PC = BF_TW + 4, BF_PCH = BF_PR * BF_PC, BF_PW = BF_CPG * BF_PCH;  // private dX rows: [wave][channel][5][36]
// tap of (step tp, half-wave): pairs with equal tj and ti = 0 / 1 first, then (6, 7), then 8 alone (9 = no tap)
__host__ __device__ constexpr int bf_tap(int tp, int hf) { return hf == 0 ? (tp < 3 ? tp : (tp == 3 ? 6 : 8)) : (tp < 3 ? 3 + tp : (tp == 3 ? 7 : 9)); }
}  // namespace

typedef int bf_i32x4 __attribute__((ext_vector_type(4)));

// LDS-DMA from inline assembly (see dcn_tapwin.hip): the compiler puts `s_waitcnt vmcnt(0)` in front of the first LDS read that
// follows the builtin form - here that was the first W^T operand read of EVERY step, right after the request for the next step's
// slab, i.e. the fetch was waited for instead of overlapping the 64 MFMAs.  Issued this way the wait-count pass does not see the
// request; the kernel waits itself (`s_waitcnt vmcnt(0)` after the MFMAs), the barriers order it against the readers.
__device__ __forceinline__ void bf_dma16(bf_i32x4 rsrc, unsigned lds, int voff, int soff) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" ::"s"(lds), "v"(voff), "s"(rsrc), "s"(soff) : "memory", "m0");
}
__device__ __forceinline__ void bf_dma4(bf_i32x4 rsrc, unsigned lds, int voff, int soff) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dword %1, %2, %3 offen lds" ::"s"(lds), "v"(voff), "s"(rsrc), "s"(soff) : "memory", "m0");
}
__device__ __forceinline__ bf_i32x4 bf_rsrc4(const void *ptr, int bytes) {
  const uint64_t pv = reinterpret_cast<uint64_t>(ptr);
  bf_i32x4 r;
  r[0] = __builtin_amdgcn_readfirstlane((int)(uint32_t)pv);
  r[1] = __builtin_amdgcn_readfirstlane((int)(pv >> 32)) & 0xffff;
  r[2] = bytes;
  r[3] = 0x00020000;
  return r;
}

struct DcnBwdFusedArgs {
  const float *x, *offset, *mask, *wbk, *dy;
  float *col, *dx, *doffset, *dmask;
  int B, C, H, W, Co, dg, tiles_x, tiles_y;
  int64_t off_bs, msk_bs, doff_bs, dmsk_bs;
};

// W (Co, C, 3, 3) -> wbk[g][tp][row m][co parity][co pair s (68: 4 zero pads)] with row m = (tap bf_tap(tp, (m >> 2) & 1), channel
// 16 g + (m & 3) + 4 (m >> 3)): the order in which v_mfma_f32_32x32x2_f32 deals output rows to the two half-waves.
__global__ void dcn_bwd_fused_pack_kernel(const float *__restrict__ w, float *__restrict__ wbk, int Co, int C, int dg) {
  const int64_t total = (int64_t)dg * BF_TP * BF_SLAB;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int s = (int)(i % BF_LS), hf = (int)((i / BF_LS) & 1), m = (int)((i / (2 * BF_LS)) % 32);
    const int tp = (int)((i / BF_SLAB) % BF_TP), g = (int)(i / ((int64_t)BF_SLAB * BF_TP));
    const int t = bf_tap(tp, (m >> 2) & 1), c = g * BF_CPG + (m & 3) + 4 * (m >> 3), co = 2 * s + hf;
    wbk[i] = (s < BF_NS && t < 9 && co < Co) ? w[((int64_t)co * C + c) * 9 + t] : 0.f;
  }
}

__global__ __launch_bounds__(BF_NT) void dcn_bwd_fused_kernel(const DcnBwdFusedArgs a) {
  constexpr int TH = BF_TH, TW = BF_TW, IH = BF_IH, IW = BF_IW, CHS = BF_CHS, CPG = BF_CPG, NS = BF_NS, LS = BF_LS, SLAB = BF_SLAB;
  constexpr int PC = BF_PC, PCH = BF_PCH, PW = BF_PW;
#ifndef BF_CG
#define BF_CG 2
#endif
  constexpr int CG = BF_CG;  // channels whose accumulator updates share one LDS round trip per column pass (2: 6.23 ms, 4: 6.37, 8: 6.98 - registers)
  constexpr int RSRC_FLAGS = 0x00020000;
  constexpr int OOB = (int)0x80000000;
  typedef __attribute__((address_space(3))) void lvoid;
  typedef __attribute__((address_space(1))) void gvoid;
  // three separate objects: the compiler may then move x-window reads across private-accumulator writes (151 KB in all)
  __shared__ __attribute__((aligned(16))) float xs[CPG * CHS];
  __shared__ __attribute__((aligned(16))) float priv[TH * PW];
  __shared__ __attribute__((aligned(16))) float wsl[2 * SLAB];

  const int tid = threadIdx.x, lane = tid & 63, half = lane >> 5, j = lane & 31;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  int tile, unused, img;
  xcd_block_index(tile, unused, img);
  const int ty0 = (tile / a.tiles_x) * TH, tx0 = (tile % a.tiles_x) * TW;
  const int P = a.H * a.W;
  const int oy = ty0 + wave, ox = tx0 + j;
  const bool pix_ok = oy < a.H && ox < a.W;
  const int p = pix_ok ? oy * a.W + ox : 0;
  const int fy0 = ty0 - 2, fx0 = tx0 - 2;  // image coordinates of element (row 0 of wave 0, column 0) of the private dX rows
  const int wy0 = ty0 - 3, wx0 = tx0 - 3;  // image coordinates of element (0, 0) of the x window
  const float *x_img = a.x + (int64_t)img * a.C * P;
  float *dx_img = a.dx + (int64_t)img * a.C * P;
  const float *off_b = a.offset + (int64_t)img * a.off_bs;
  const float *msk_b = a.mask + (int64_t)img * a.msk_bs;

  // Every global access of the hot path is a buffer instruction: wave-uniform resource + 32-bit lane offset + scalar offset (no
  // 64-bit address arithmetic in vector registers), dead lanes carry an out-of-range offset (loads return 0, stores are dropped:
  // no branches around them).
  auto rsrc_of = [&](const void *ptr, int bytes) {
    const uint64_t pv = reinterpret_cast<uint64_t>(ptr);
    const uint64_t pu = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(pv >> 32)) << 32) |
                        (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)pv);
    return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void *>(pu), (short)0, bytes, RSRC_FLAGS);
  };
  const __amdgpu_buffer_rsrc_t x_rsrc = rsrc_of(x_img, a.C * P * 4);
  const __amdgpu_buffer_rsrc_t col_rsrc = rsrc_of(a.col + (int64_t)img * a.C * 9 * P, a.C * 9 * P * 4);
  const __amdgpu_buffer_rsrc_t off_rsrc = rsrc_of(off_b, a.dg * 18 * P * 4), msk_rsrc = rsrc_of(msk_b, a.dg * 9 * P * 4);
  const __amdgpu_buffer_rsrc_t doff_rsrc = rsrc_of(a.doffset + (int64_t)img * a.doff_bs, a.dg * 18 * P * 4);
  const __amdgpu_buffer_rsrc_t dmsk_rsrc = rsrc_of(a.dmask + (int64_t)img * a.dmsk_bs, a.dg * 9 * P * 4);
  auto bload = [&](const __amdgpu_buffer_rsrc_t &r, int voff, int soff) {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, 0));
  };
  auto bstore = [&](float v, const __amdgpu_buffer_rsrc_t &r, int voff, int soff) {
    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), r, voff, soff, 0);
  };

  // ---- dY tile of this wave's pixel row: dyr[s] = dY[co = 2 s + half][pixel j] (the MFMA B operand of k-step s); rows past Co carry
  //      the out-of-range LANE offset (not left to the range check of lane + scalar offset) and read 0.  BF_DY_STREAM: re-read (L2) at the top of every step instead of register-resident, so
  //      that the consumer phase has the 64 registers for its own loads in flight.
  const __amdgpu_buffer_rsrc_t dy_rsrc = rsrc_of(a.dy + (int64_t)img * a.Co * P, a.Co * P * 4);
  const int dy_voff = pix_ok ? (half * P + p) * 4 : OOB;
  float dyr[NS];
  auto load_dy = [&]() {
#pragma unroll
    for (int s = 0; s < NS; ++s) dyr[s] = bload(dy_rsrc, 2 * s + half < a.Co ? dy_voff : OOB, 2 * s * P * 4);
  };
  load_dy();

  // ---- x window of one deformable group -> LDS (as dcn_fused.hip: positions outside the image fail the range check -> 0)
  constexpr int NT = BF_NT, NXK = (CHS + NT - 1) / NT;
  int xoff[NXK];
#pragma unroll
  for (int k = 0; k < NXK; ++k) {
    const int q = tid + k * NT;
    const int iy = q / IW, ix = q - iy * IW;
    const int gy = wy0 + iy, gx = wx0 + ix;
    xoff[k] = (gy >= 0 && gy < a.H && gx >= 0 && gx < a.W) ? (gy * a.W + gx) * 4 : OOB;
  }
  const bf_i32x4 x_rsrc4 = bf_rsrc4(x_img, a.C * P * 4);
  const bf_i32x4 w_rsrc4 = bf_rsrc4(a.wbk, a.dg * BF_TP * SLAB * 4);
  const unsigned xs_lds = (unsigned)(size_t)(lvoid *)xs, wsl_lds = (unsigned)(size_t)(lvoid *)wsl;
  auto dma_x = [&](int g) {
#pragma unroll
    for (int k = 0; k < NXK; ++k)
      if (tid + k * NT < CHS) {  // the lanes past the end are masked off (LDS-DMA writes active lanes only)
#pragma unroll
        for (int ch = 0; ch < CPG; ++ch) bf_dma4(x_rsrc4, xs_lds + (ch * CHS + k * NT + wave * 64) * 4, xoff[k], (g * CPG + ch) * P * 4);
      }
  };
  // ---- W^T slab of (group, step) i -> LDS buffer: a linear copy of 17 KB in 16-byte pieces
  auto dma_w = [&](int buf, int i) {
    constexpr int TOTAL = SLAB / 4;  // float4 pieces
    for (int q0 = wave * 64; q0 < TOTAL; q0 += NT)
      if (q0 + lane < TOTAL) bf_dma16(w_rsrc4, wsl_lds + (buf * SLAB + q0 * 4) * 4, (q0 + lane) * 16, i * SLAB * 4);
  };

  for (int i = tid; i < TH * PW; i += NT) priv[i] = 0.f;
  dma_w(0, 0);
  dma_x(0);
  // offsets / mask of the lane's tap of a step: requested at the top of the step, used after its MFMAs
  auto tap_voff = [&](int tp, int &v1, int &v2) {  // lane offsets of (tap plane, pixel) in 9-plane and 18-plane groups
    const int t = half ? bf_tap(tp, 1) : bf_tap(tp, 0);
    const bool on = pix_ok && t < 9;
    v1 = on ? (t * P + p) * 4 : OOB;
    v2 = on ? (2 * t * P + p) * 4 : OOB;
  };
  auto fetch_tap = [&](int g, int tp, float &fh, float &fw, float &fm) {
    int v1, v2;
    tap_voff(tp, v1, v2);
    fh = bload(off_rsrc, v2, g * 18 * P * 4);
    fw = bload(off_rsrc, v2, (g * 18 + 1) * P * 4);
    fm = bload(msk_rsrc, v1, g * 9 * P * 4);
  };
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // slab 0 (this wave's pieces) before the first barrier

  const int n_steps = a.dg * BF_TP;
  const int abase = (j * 2 + half) * LS;
  float *pw = priv + wave * PW;  // this wave's accumulator rows: [channel][row oy - 2 .. oy + 2][column tx0 - 2 .. tx0 + 33]

  auto step = [&](auto TPC, int g) {
    constexpr int TP = decltype(TPC)::value;
    const int i = g * BF_TP + TP;
    // LDS-only barrier: slab i is in LDS (every wave waited for its own pieces after the MFMAs of the step before), nobody reads
    // slab i - 1 or updates accumulator rows of the step before any more
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    const float *slab = wsl + (i & 1) * SLAB;
    if (i + 1 < n_steps) dma_w((i + 1) & 1, i + 1);
    float o_h, o_w, o_m;  // offsets / mask of the lane's tap: requested here, landed by the end of the MFMAs
    fetch_tap(g, TP, o_h, o_w, o_m);

    // ---- 1. dcol rows of (g, the two taps of the step) for this wave's 32 pixels: 64 k-steps over the output channels
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
    for (int s = 0; s < NS; s += 4) {
      const f32x4 av = *reinterpret_cast<const f32x4 *>(slab + abase + s);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[0], dyr[s + 0], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[1], dyr[s + 1], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[2], dyr[s + 2], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[3], dyr[s + 3], acc, 0, 0, 0);
    }

    // everything this wave has in flight - its pieces of slab i + 1, its taps, at a group's first step its pieces of the x
    // window - was requested before the 64 MFMAs: waiting HERE, and not at the barrier, keeps the column stores of the consumer
    // below (whose completion nothing depends on) out of every wait
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if constexpr (TP == 0) __syncthreads();  // the other waves' pieces of the x window

    // ---- 2. consume: lane (pixel j, half) holds dcol[channel r][its tap][pixel] in acc[r]
    const int t = half ? bf_tap(TP, 1) : bf_tap(TP, 0);
    const bool act = pix_ok && t < 9;
    const int ti = t / 3, tj = t - 3 * ti;
    const float h = (float)(oy - 1 + ti) + o_h, w = (float)(ox - 1 + tj) + o_w;
    const bool valid = act && h > -1.f && w > -1.f && h < (float)a.H && w < (float)a.W;  // .cu:618
    const float fh = floorf(h), fw = floorf(w);
    const float lh = h - fh, lw = w - fw, hh = 1.f - lh, hw = 1.f - lw;
    const int fhi = (int)fh, fwi = (int)fw;
    const int fy = fhi - (oy - 1 + ti), fx = fwi - (ox - 1 + tj);  // floor of the offsets
    const int wr = fhi - wy0, wc = fwi - wx0;
    const bool xw = valid && wr >= 0 && wr <= IH - 2 && wc >= 0 && wc <= IW - 2;  // the bilinear cell lies inside the x window
    const bool sub = valid && (unsigned)(fy + 1) <= 1u && (unsigned)(fx + 1) <= 1u;  // sub-pixel offset (always inside the window)
    const bool mid = xw && !sub, slow = valid && !xw;
    const int xaddr = xw ? wr * IW + wc : 0;
    const float u00 = xw ? hh * hw : 0.f, u01 = xw ? hh * lw : 0.f, u10 = xw ? lh * hw : 0.f, u11 = xw ? lh * lw : 0.f;
    // row / column weights of the static 3 x 3 block around the regular tap position (index -1, 0, +1)
    const float rym = (sub && fy < 0) ? hh : 0.f, ry0 = sub ? (fy < 0 ? lh : hh) : 0.f, ryp = (sub && fy == 0) ? lh : 0.f;
    const float cx[3] = {(sub && fx < 0) ? hw : 0.f, sub ? (fx < 0 ? lw : hw) : 0.f, (sub && fx == 0) ? lw : 0.f};
    const int pcol = j + tj + 1;  // accumulator column of the block's centre
    float s_m = 0.f, s_y = 0.f, s_x = 0.f;
    int v1, v2;
    tap_voff(TP, v1, v2);

    // merged steps (0-2): taps (ti = 0, tj) and (ti = 1, tj).  The half-wave of ti = 0 owns accumulator rows 0, 1 (block rows
    // -1, 0), the other one rows 3, 2 (block rows +1, 0); block row +1 of the first (accumulator row 2) and block row -1 of the
    // second (accumulator row 1) are handed over: k0 = the own outer row, k1 = the own centre row which also takes the gift.
    const float wk0 = half ? ryp : rym, wk1 = ry0, wgift = half ? rym : ryp;
    float pcx[3];
    if constexpr (TP < 3) {
#pragma unroll
      for (int b = 0; b < 3; ++b) pcx[b] = __shfl_xor(cx[b], 32, 64);  // the partner's column weights (same tj: same columns)
    }
    float *pk0 = pw + (half ? 3 : 0) * PC + pcol - 1, *pk1 = pw + (half ? 2 : 1) * PC + pcol - 1;  // column pass b at [b]
    float *pun = pw + ti * PC + pcol - 1;  // unmerged steps: block row a (0..2 = -1..+1) at [a * PC], rows ti .. ti + 2

#pragma unroll
    for (int c4 = 0; c4 < CPG; c4 += CG) {
      float tt[CG];
#pragma unroll
      for (int u = 0; u < CG; ++u) {
        const int c = c4 + u;
        const float *cell = xs + c * CHS + xaddr;
        const float a00 = cell[0], a01 = cell[1], a10 = cell[IW], a11 = cell[IW + 1];
        const float d = xw ? acc[c] : 0.f;
        const float val = u00 * a00 + u01 * a01 + u10 * a10 + u11 * a11;
        s_m += d * val;
        s_y += d * (hw * (a10 - a00) + lw * (a11 - a01));
        s_x += d * (hh * (a01 - a00) + lh * (a11 - a10));
        tt[u] = d * o_m;
        bstore(val * o_m, col_rsrc, v1, (g * CPG + c) * 9 * P * 4);  // forward column (row c * 9 + t), consumed by the dW GEMM
      }
      if constexpr (TP < 3) {
        float r0[CG], r1[CG], gift[CG];
#pragma unroll
        for (int u = 0; u < CG; ++u) {
          r0[u] = tt[u] * wk0;
          r1[u] = tt[u] * wk1;
          gift[u] = __shfl_xor(tt[u] * wgift, 32, 64);
        }
#pragma unroll
        for (int b = 0; b < 3; ++b) {  // column passes: separate instructions, in order (neighbouring lanes overlap across passes)
          float o0[CG], o1[CG];
#pragma unroll
          for (int u = 0; u < CG; ++u) {
            o0[u] = pk0[(c4 + u) * PCH + b];
            o1[u] = pk1[(c4 + u) * PCH + b];
          }
#pragma unroll
          for (int u = 0; u < CG; ++u) {
            pk0[(c4 + u) * PCH + b] = o0[u] + r0[u] * cx[b];
            pk1[(c4 + u) * PCH + b] = o1[u] + r1[u] * cx[b] + gift[u] * pcx[b];
          }
          // the next pass of THIS lane reads what its neighbour lane wrote in this one: invisible to the compiler's per-thread
          // alias analysis, so the order is pinned here (the hardware executes a wave's LDS instructions in order)
          asm volatile("" ::: "memory");
        }
      } else {
        // taps 6, 7 (same row, neighbouring columns: the half-waves take turns), tap 8 (first half-wave only)
#pragma unroll
        for (int hs = 0; hs < (TP == 3 ? 2 : 1); ++hs) {
          if (half == hs) {
#pragma unroll
            for (int b = 0; b < 3; ++b) {
              float o[CG][3];
#pragma unroll
              for (int u = 0; u < CG; ++u)
#pragma unroll
                for (int r = 0; r < 3; ++r) o[u][r] = pun[(c4 + u) * PCH + r * PC + b];
#pragma unroll
              for (int u = 0; u < CG; ++u) {
                const float tc = tt[u] * cx[b];
                pun[(c4 + u) * PCH + 0 * PC + b] = o[u][0] + tc * rym;
                pun[(c4 + u) * PCH + 1 * PC + b] = o[u][1] + tc * ry0;
                pun[(c4 + u) * PCH + 2 * PC + b] = o[u][2] + tc * ryp;
              }
              asm volatile("" ::: "memory");  // (as above)
            }
          }
          asm volatile("" ::: "memory");  // ... and the second half-wave's turn reads what the first one wrote
        }
      }
    }
    if (__any(mid)) {  // inside the window but not sub-pixel: dX by device atomics (four per channel, nothing waits for them)
      if (mid) {
        const bool r0 = fhi >= 0, r1 = fhi + 1 <= a.H - 1, c0 = fwi >= 0, c1 = fwi + 1 <= a.W - 1;  // corners inside the image (.cu:481-491)
        float *gp = dx_img + (int64_t)g * CPG * P + fhi * a.W + fwi;
#pragma unroll
        for (int c = 0; c < CPG; ++c) {
          const float tt = acc[c] * o_m;
          if (r0 && c0) unsafeAtomicAdd(gp, u00 * tt);
          if (r0 && c1) unsafeAtomicAdd(gp + 1, u01 * tt);
          if (r1 && c0) unsafeAtomicAdd(gp + a.W, u10 * tt);
          if (r1 && c1) unsafeAtomicAdd(gp + a.W + 1, u11 * tt);
          gp += P;
        }
      }
    }
    if (__any(slow)) {  // the cell left the window: global memory, full bounds logic, device atomics
      if (slow) {
        const Tap tq = resolve_tap(h, w, a.H, a.W);
        const float gy00 = tq.ok00 ? -hw : 0.f, gy01 = tq.ok01 ? -lw : 0.f, gy10 = tq.ok10 ? hw : 0.f, gy11 = tq.ok11 ? lw : 0.f;
        const float gx00 = tq.ok00 ? -hh : 0.f, gx01 = tq.ok01 ? hh : 0.f, gx10 = tq.ok10 ? -lh : 0.f, gx11 = tq.ok11 ? lh : 0.f;
        const float *xp = x_img + (int64_t)g * CPG * P;
        float *gp = dx_img + (int64_t)g * CPG * P;
#pragma unroll 1
        for (int c = 0; c < CPG; ++c) {
          const float a00 = xp[tq.o00], a01 = xp[tq.o01], a10 = xp[tq.o10], a11 = xp[tq.o11];
          float d = 0.f;  // acc[c], c not a constant in this rolled loop (the path is cold: code size over speed)
#pragma unroll
          for (int r = 0; r < CPG; ++r) d = r == c ? acc[r] : d;
          const float val = tq.w00 * a00 + tq.w01 * a01 + tq.w10 * a10 + tq.w11 * a11;
          s_m += d * val;
          s_y += d * (gy00 * a00 + gy01 * a01 + gy10 * a10 + gy11 * a11);
          s_x += d * (gx00 * a00 + gx01 * a01 + gx10 * a10 + gx11 * a11);
          const float tt = d * o_m;
          if (tq.ok00) unsafeAtomicAdd(gp + tq.o00, tq.w00 * tt);
          if (tq.ok01) unsafeAtomicAdd(gp + tq.o01, tq.w01 * tt);
          if (tq.ok10) unsafeAtomicAdd(gp + tq.o10, tq.w10 * tt);
          if (tq.ok11) unsafeAtomicAdd(gp + tq.o11, tq.w11 * tt);
          bstore(val * o_m, col_rsrc, v1, (g * CPG + c) * 9 * P * 4);
          xp += P;
          gp += P;
        }
      }
    }
    bstore(s_m, dmsk_rsrc, v1, g * 9 * P * 4);
    bstore(s_y * o_m, doff_rsrc, v2, g * 18 * P * 4);
    bstore(s_x * o_m, doff_rsrc, v2, (g * 18 + 1) * P * 4);
  };

  for (int g = 0; g < a.dg; ++g) {
    step(std::integral_constant<int, 0>{}, g);
    step(std::integral_constant<int, 1>{}, g);
    step(std::integral_constant<int, 2>{}, g);
    step(std::integral_constant<int, 3>{}, g);
    step(std::integral_constant<int, 4>{}, g);
    // group done: sum the eight private accumulators over the rows they share (image row fy0 + R is row R - w of wave w), flush
    // with one global atomic per touched element inside the image, clear them, fetch the next x window
    __syncthreads();
    float *gg = dx_img + (int64_t)g * CPG * P;
    for (int q = tid; q < CPG * (TH + 4) * PC; q += NT) {
      const int cc = q / ((TH + 4) * PC), rem = q - cc * ((TH + 4) * PC), R = rem / PC, col = rem - R * PC;
      float v = 0.f;
#pragma unroll
      for (int r = 0; r < BF_PR; ++r) {
        const int wv = R - r;
        if (wv >= 0 && wv < TH) {
          float *e = priv + wv * PW + cc * PCH + r * PC + col;
          v += *e;
          *e = 0.f;
        }
      }
      const int gy = fy0 + R, gx = fx0 + col;
      if (v != 0.f && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W) unsafeAtomicAdd(gg + (int64_t)cc * P + gy * a.W + gx, v);
    }
    if (g + 1 < a.dg) dma_x(g + 1);
  }
}

bool dcn_bwd_fused_supported(const DcnShape &s) {
  static const bool enabled = []() {
    const char *e = getenv("EDVR_DCN_BWD_FUSED");  // "0": the staged path (dcol GEMM + strip / window kernels), for A/B
    return !(e && e[0] == '0');
  }();
  if (!enabled) return false;
  if (!(s.kh == 3 && s.kw == 3 && s.stride == 1 && s.pad == 1 && s.dil == 1 && s.stride_w == 1 && s.pad_w == 1 && s.dil_w == 1)) return false;
  if (s.groups != 1 || s.C != s.dg * BF_CPG || s.Co > 2 * BF_NS || s.B > 65535) return false;
  if (s.W < BF_TW) return false;  // half-empty waves: the staged path is faster (0.74 vs 0.98 ms on the 160 x 128 x 16 x 16 layer)
  return (int64_t)s.C * 9 * s.H * s.W * 4 < ((int64_t)1 << 31);  // 32-bit buffer offsets (the column rows of one image are the largest extent)
}

size_t dcn_bwd_fused_wbk_elems(int dg) { return (size_t)dg * BF_TP * BF_SLAB; }

int dcn_bwd_fused_launch(const DcnShape &s, const float *x, const float *offset, const float *mask, const float *weight, const float *dy,
                         float *wbk, float *col, float *dx, float *doffset, float *dmask, hipStream_t stream) {
  const int64_t total = (int64_t)dcn_bwd_fused_wbk_elems(s.dg);
  hipLaunchKernelGGL(dcn_bwd_fused_pack_kernel, dim3((unsigned)cdiv64(total, 256)), dim3(256), 0, stream, weight, wbk, s.Co, s.C, s.dg);
  DcnBwdFusedArgs a;
  a.x = x; a.offset = offset; a.mask = mask; a.wbk = wbk; a.dy = dy;
  a.col = col; a.dx = dx; a.doffset = doffset; a.dmask = dmask;
  a.B = s.B; a.C = s.C; a.H = s.H; a.W = s.W; a.Co = s.Co; a.dg = s.dg;
  a.tiles_x = cdiv(s.W, BF_TW);
  a.tiles_y = cdiv(s.H, BF_TH);
  a.off_bs = s.off_bs; a.msk_bs = s.msk_bs; a.doff_bs = s.doff_bs; a.dmsk_bs = s.dmsk_bs;
  hipLaunchKernelGGL(dcn_bwd_fused_kernel, dim3(a.tiles_x * a.tiles_y, 1, s.B), dim3(BF_NT), 0, stream, a);
  return check_launch("dcn_bwd_fused_kernel");
}

}  // namespace edvr
