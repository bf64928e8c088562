// wgrad_direct_s.hip - 3x3 / stride-1 weight gradient as a DIRECT pixel-axis GEMM with split fp32 operands on the f16 matrix pipe (gfx950).
//
//   dW[co][ci][ky][kx] = sum over images and pixels of dY[co][y][x] * X[ci][y + ky - 1][x + kx - 1]
//   (the reduction cuDNN's backward-filter does for the 3x3 convs of basicsr/models/archs/arch_util.py:67-95, edvr_arch.py:346-353).
//
// winograd_wgrad_s.hip runs this in the F(2x2) Winograd domain: 2.25x fewer multiplications, but both operands are DATA there, so
// every chunk pays two transforms + 16 splits per tile and channel (~35 vector instructions per pixel and channel pair side) - on
// split operands the matrix pipe was 22 % busy and the staging streams set the time.  With the f16 pipe 4x cheaper the
// trade flips: here nothing is transformed.  A value is split ONCE (2 instructions: x s = hi + lo, winograd_f4s.hip) and written to
// LDS; the nine taps are nine reads of the same rows at shifted addresses.  Per pixel and channel ~3 vector instructions instead
// of ~35; 2.25x the matrix work, which is what the pipe has room for.
//
// Work: (split, 64 output channels, 64 input channels) per 512-thread workgroup; a split walks a contiguous range of rows of
// 32-pixel column strips, one row = one step (K = 32 pixels):
//   * k-slots (2 i, 2 i + 1) of an operand register hold (hi, lo) of ONE pixel: a v_mfma_f32_32x32x16_f16 covers 8 pixels; the second
//     MFMA of a pair takes A rotated by 16 bits ((lo, hi): the cross terms) - A = dY is rotated once per 8 pixels and reused by all taps;
//   * LDS: dY rows as [pixel quad 8][co 64][4 pixels] dwords (hi | lo << 16); X rows in a ring of four, each row THREE times -
//     copy kx holds x[4 q + kx - 1 .. + 3] in quad q, so the operand of every tap (ky, kx) is ONE aligned, conflict-free
//     ds_read_b128 (an operand is four consecutive, even-aligned registers: a one-pixel shift cannot be a register offset);
//   * wave = (32 x 32 block of the 64 x 64, tap group): waves 0-3 accumulate taps 0-4, waves 4-7 taps 5-8 (siblings share a SIMD:
//     9 taps x 4 x 2 MFMAs of 32 cycles per SIMD and step), 80 accumulator registers, no exchange at the end;
//   * staging, one (channel, pixel quad) of each tensor per thread and step: a 16-byte load + the two neighbours of the X quad,
//     10 splits, 4 ds_write_b128; the loads of row r + 1 are issued before the MFMAs of the step and written at the next step.
// Row r of a strip is staged at iteration r and multiplied (as the centre row of dY row r - 2) at iteration r; three iterations per
// strip segment only stage.  Partials [split][co][ci][9] (+ bias-gradient partials) in the format of winograd_wgrad.hip: the same
// reduction kernel sums them.
#include <cstdlib>
#include <type_traits>

#include "common.h"

namespace edvr {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

namespace {
// x * s = hi + lo in f16 (s a power of two), packed (hi | lo << 16)
__device__ __forceinline__ unsigned wds_split(float x, float s) {
  unsigned o;
  asm volatile("v_fma_mixlo_f16 %0, %1, %2, 0\n\tv_fma_mixhi_f16 %0, %1, %2, -%0 op_sel_hi:[0,0,1]" : "=&v"(o) : "v"(x), "s"(s));
  return o;
}
// 2^e with amax * 2^e < 2^15: amax = m 2^k, m in [1, 2) -> e = 14 - k  (zero / tiny bounds stop at 2^88, non-finite ones give 2^-120)
__device__ __forceinline__ float wds_scale(float amax) {
  const int be = (int)((__builtin_bit_cast(unsigned, amax) >> 23) & 255u);
  return __builtin_bit_cast(float, (unsigned)min(max(127 + 14 - (be - 127), 7), 215) << 23);
}
// The rows in flight live in FIXED registers v216 .. v255, outside the compiler's reach (the kernel is limited to 216 registers):
// four sets of ten - X columns 4 q .. 4 q + 3 (B .. B + 3), dY columns (B + 4 .. B + 7), X columns 4 q - 1 and 4 q + 4 (B + 8, B + 9) -
// requested four steps before their use.  With loads the compiler can see (builtins) its wait-count pass cannot follow registers
// through the unrolled walk and puts s_waitcnt vmcnt(0) in front of every use: one row in flight, ~2 us of loaded-memory latency per
// ~1.2 us step, and the load time ADDED to the multiplying time (0.27 + 0.60 ms on the 160 x 128 x 64 x 64 layer); loads into
// compiler-allocated registers from inline assembly were copied (v_mov) while in flight.  Here a request is invisible to the compiler
// and nothing but these statements touches the registers; the kernel counts itself: every iteration requests exactly four loads,
// so `s_waitcnt vmcnt(12)` = the set requested four iterations ago has landed.
constexpr int WDS_ROWREG = 216;
template <int R>
__device__ __forceinline__ void wds_load16(i32x4 rsrc, int voff) {
  asm volatile("buffer_load_dwordx4 v[%2:%3], %0, %1, 0 offen" ::"v"(voff), "s"(rsrc), "n"(R), "n"(R + 3) : "memory");
}
template <int R>
__device__ __forceinline__ void wds_load4(i32x4 rsrc, int voff) {
  asm volatile("buffer_load_dword v[%2], %0, %1, 0 offen" ::"v"(voff), "s"(rsrc), "n"(R) : "memory");
}
template <int R>
__device__ __forceinline__ unsigned wds_split_reg(float s) {  // wds_split of fixed register R
  unsigned o;
  asm volatile("v_fma_mixlo_f16 %0, v[%2], %1, 0\n\tv_fma_mixhi_f16 %0, v[%2], %1, -%0 op_sel_hi:[0,0,1]" : "=&v"(o) : "s"(s), "n"(R));
  return o;
}
template <int R>
__device__ __forceinline__ float wds_sum4_reg(float acc) {  // acc + v[R] + v[R + 1] + v[R + 2] + v[R + 3]
  asm volatile("v_add_f32 %0, %0, v[%1]\n\tv_add_f32 %0, %0, v[%2]\n\tv_add_f32 %0, %0, v[%3]\n\tv_add_f32 %0, %0, v[%4]"
               : "+v"(acc)
               : "n"(R), "n"(R + 1), "n"(R + 2), "n"(R + 3));
  return acc;
}
constexpr int WDS_SLAB = 8 * 64 * 4;        // dwords of one row of one tensor: [quad 8][channel 64][4]
constexpr int WDS_XROW = 3 * WDS_SLAB;      // the three shifted copies of an X row
}  // namespace

#ifdef WDS_CLOCK
__device__ unsigned long long wds_clock_cycles;  // measurement build: the longest workgroup's s_memtime span
#endif

struct WgradDirectSArgs {
  const float *x_amax, *dz_amax;
  const float *x1, *x2, *dz;
  float *ws;  // [splits][co][ci][9], then (want_db) [splits][co]
  int want_db;
  int c1, c2, n, h, w, co;
  int64_t x1_img_stride, x2_img_stride, dz_img_stride;
  int x2_div, x2_mul, x2_add;
  int cw, total_rows, splits, ci_blocks, co_blocks;  // 32-pixel strips per image row; n * cw * h
};

__global__ __launch_bounds__(512, 1) __attribute__((amdgpu_num_vgpr(WDS_ROWREG))) void conv3x3_wgrad_direct_split_kernel(const WgradDirectSArgs a) {
  asm volatile("" ::: "v255");  // (the register allocation of a wave covers the fixed row registers)
#ifdef WDS_CLOCK
  unsigned long long wds_t0;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(wds_t0)::"memory");
#endif
  constexpr int RSRC_FLAGS = 0x00020000;
  constexpr int OOB = (int)0x80000000;
  __shared__ __attribute__((aligned(16))) unsigned smem[4 * WDS_XROW + 4 * WDS_SLAB];  // X ring (96 KB) + dY ring (32 KB)
  unsigned *const Xs = smem, *const Zs = smem + 4 * WDS_XROW;
  const float s_x = wds_scale(__builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, *a.x_amax))));
  const float s_z = wds_scale(__builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, *a.dz_amax))));

  const int tid = threadIdx.x, lane = tid & 63, half = lane >> 5, j = lane & 31;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int quadrant = wave & 3, wm = quadrant >> 1, wn = quadrant & 1, ph = wave >> 2;
  const int hw = a.h * a.w, ci_total = a.c1 + a.c2;
  const int blocks = a.ci_blocks * a.co_blocks;
  const int lg = xcd_remap((int)blockIdx.x, (int)gridDim.x);
  const int split = __builtin_amdgcn_readfirstlane(lg / blocks);
  const int blk = __builtin_amdgcn_readfirstlane(lg % blocks);
  const int co_blk = __builtin_amdgcn_readfirstlane((blk / a.ci_blocks) * 64);
  const int ci_blk = __builtin_amdgcn_readfirstlane((blk % a.ci_blocks) * 64);
  // rows [R0, R1) of the flattened (image, strip, row) axis
  const int R0 = __builtin_amdgcn_readfirstlane((int)((int64_t)a.total_rows * split / a.splits));
  const int R1 = __builtin_amdgcn_readfirstlane((int)((int64_t)a.total_rows * (split + 1) / a.splits));
  const float unscale = (1.f / s_x) * (1.f / s_z);
  float *const out = a.ws + (int64_t)split * a.co * ci_total * 9;

  // ---- staging role: pixel quad q = lane >> 3 of the strip row, channel chl = 8 wave + (lane & 7) of the block (eight lanes of a
  //      16-byte LDS write differ in the channel: 128 contiguous bytes)
  const int q = lane >> 3, chl = wave * 8 + (lane & 7);
  const bool use_x2 = a.c2 > 0 && ci_blk >= a.c1;  // a block never straddles x1 / x2 (c1 % 64 == 0, host check)
  const int ci_s = ci_blk + chl, co_s = co_blk + chl;
  const bool valid_ci = ci_s < (use_x2 || a.c2 == 0 ? ci_total : a.c1), valid_co = co_s < a.co;
  const int ci_in = use_x2 ? ci_s - a.c1 : ci_s;
  const int x_ch = ci_in * hw * 4, z_ch = co_s * hw * 4;  // byte offsets of the channel planes
  const int w_off = (q * 64 + chl) * 4;                   // this thread's 16 bytes of a slab (dwords)

  auto exact_rsrc = [&](const float *p, int bytes) {  // base = the image, num_records = its bytes: anything past the end reads 0
    const uint64_t pv = reinterpret_cast<uint64_t>(p);
    i32x4 r;
    r[0] = __builtin_amdgcn_readfirstlane((int)(uint32_t)pv);
    r[1] = __builtin_amdgcn_readfirstlane((int)(pv >> 32)) & 0xffff;
    r[2] = bytes;
    r[3] = RSRC_FLAGS;
    return r;
  };
  const int x_img_bytes = (use_x2 ? a.c2 : a.c1) * hw * 4, z_img_bytes = a.co * hw * 4;
  i32x4 x_rsrc = exact_rsrc(a.x1, 0), z_rsrc = x_rsrc;
  int rsrc_img = -1;

  // ---- rows in flight.  A cursor = (column = image * cw + strip, segment [ya, yb) of its dY rows, row r); a segment takes the rows
  //      r = ya - 1 .. yb + 1 (X rows ya - 1 .. yb are staged; dY row r - 2 is multiplied when row r is committed).  The LOAD cursor
  //      runs four rows ahead of the commit cursor (four fixed register sets, above).
  struct Cur {
    int col, ya, yb, r;
  };
  auto cur_valid = [&](const Cur &c) { return c.yb > c.ya; };
  auto cur_next = [&](Cur &c) {
    if (++c.r > c.yb + 1) {
      ++c.col;
      c.ya = 0;
      c.yb = min(a.h, R1 - c.col * a.h);  // (<= 0 past the split's last row: invalid)
      c.r = -1;
    }
  };
  // (always four loads - past the split's last row with out-of-range offsets, which return zeros without touching memory: the
  // kernel's own wait count relies on it)
  auto issue_loads = [&](auto SET, const Cur &c, bool live) {
    constexpr int B = WDS_ROWREG + 10 * decltype(SET)::value;
    const int img = live ? c.col / a.cw : max(rsrc_img, 0), x0 = (c.col - img * a.cw) * 32 + 4 * q;
    if (img != rsrc_img) {  // wave-uniform
      rsrc_img = img;
      const float *xi;
      if (use_x2) {
        const int i2 = a.x2_div > 0 ? (img / a.x2_div) * a.x2_mul + a.x2_add : img;
        xi = a.x2 + (int64_t)i2 * a.x2_img_stride;
      } else {
        xi = a.x1 + (int64_t)img * a.x1_img_stride;
      }
      x_rsrc = exact_rsrc(xi, x_img_bytes);
      z_rsrc = exact_rsrc(a.dz + (int64_t)img * a.dz_img_stride, z_img_bytes);
    }
    const bool row_x = live && (unsigned)c.r < (unsigned)a.h && c.r <= c.yb;  // rows above / below the image are the zero padding
    const bool row_z = live && c.r >= c.ya && c.r < c.yb;                      // dY rows outside the segment belong to another split
    const int po = (c.r * a.w + x0) * 4;
    const bool in_w = x0 < a.w;                                        // w % 4 == 0: a quad is inside or outside as a whole
    wds_load16<B>(x_rsrc, (row_x && in_w && valid_ci) ? x_ch + po : OOB);
    wds_load16<B + 4>(z_rsrc, (row_z && in_w && valid_co) ? z_ch + po : OOB);
    wds_load4<B + 8>(x_rsrc, (row_x && valid_ci && x0 > 0 && x0 - 1 < a.w) ? x_ch + po - 4 : OOB);
    wds_load4<B + 9>(x_rsrc, (row_x && valid_ci && x0 + 4 < a.w) ? x_ch + po + 16 : OOB);
  };
  float bsum = 0.f;
  // Staging of one row in EIGHT pieces (registers -> split -> ring slot r & 3, then the set is re-requested): an iteration that
  // multiplies issues one piece behind each of its first eight MFMAs - a piece is 4-12 vector instructions, an MFMA keeps the
  // pipe for 32 cycles - instead of all of it in front of them (with every wave staging at once and then every wave multiplying
  // the two phases added up: 0.27 + 0.60 ms on the 160 x 128 x 64 x 64 layer).
  unsigned sx[6], sz[4];
  auto stage_piece = [&](auto SET, int k, int r, bool more, Cur &lc) {
    constexpr int B = WDS_ROWREG + 10 * decltype(SET)::value;
    if (k == 0) {
      asm volatile("s_waitcnt vmcnt(12)" ::: "memory");  // loads return in order: all that is outstanding now are the 12 of the three newer sets
      sx[0] = wds_split_reg<B + 8>(s_x); sx[1] = wds_split_reg<B + 0>(s_x); sx[2] = wds_split_reg<B + 1>(s_x);
    } else if (k == 1) {
      sx[3] = wds_split_reg<B + 2>(s_x); sx[4] = wds_split_reg<B + 3>(s_x); sx[5] = wds_split_reg<B + 9>(s_x);
    } else if (k >= 2 && k <= 4) {  // copy kx = k - 2: x[4 q + kx - 1 .. + 3]
      unsigned *xd = Xs + (r & 3) * WDS_XROW + (k - 2) * WDS_SLAB + w_off;
      *reinterpret_cast<i32x4 *>(xd) = i32x4{(int)sx[k - 2], (int)sx[k - 1], (int)sx[k], (int)sx[k + 1]};
    } else if (k == 5) {
      bsum = wds_sum4_reg<B + 4>(bsum);  // (rows outside the segment were loaded as zeros)
      sz[0] = wds_split_reg<B + 4>(s_z); sz[1] = wds_split_reg<B + 5>(s_z); sz[2] = wds_split_reg<B + 6>(s_z); sz[3] = wds_split_reg<B + 7>(s_z);
    } else if (k == 6) {
      *reinterpret_cast<i32x4 *>(Zs + (r & 3) * WDS_SLAB + w_off) = i32x4{(int)sz[0], (int)sz[1], (int)sz[2], (int)sz[3]};
    } else if (k == 7) {
      issue_loads(SET, lc, more);
      if (more) cur_next(lc);
    }
  };

  // operand addresses: quad 2 s + half of the row slab, this wave's 32 channels
  const unsigned a_lane = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned *)(Zs + (half * 64 + wm * 32 + j) * 4);
  const unsigned b_lane = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned *)(Xs + (half * 64 + wn * 32 + j) * 4);
  typedef __attribute__((address_space(3))) const i32x4 lds_q;

  // The whole walk once per tap group (a wave-uniform choice made ONCE: inside the loop the two groups' accumulator sets met at every
  // join and were copied register by register).
  auto run = [&](auto PHT) {
    constexpr int PH = decltype(PHT)::value, T0 = PH ? 5 : 0, NT = PH ? 4 : 5;
    f32x16 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    if (R0 < R1) {
      Cur cc, lc;  // commit / load cursors
      cc.col = R0 / a.h; cc.ya = R0 - cc.col * a.h; cc.yb = min(a.h, R1 - cc.col * a.h); cc.r = cc.ya - 1;
      lc = cc;
      // FOUR register sets: a row is requested four iterations before it is committed
      using S0 = std::integral_constant<int, 0>;
      using S1 = std::integral_constant<int, 1>;
      using S2 = std::integral_constant<int, 2>;
      using S3 = std::integral_constant<int, 3>;
      auto prime = [&](auto SET) {
        const bool live = cur_valid(lc);
        issue_loads(SET, lc, live);
        if (live) cur_next(lc);
      };
      prime(S0{}); prime(S1{}); prime(S2{}); prime(S3{});
      // one iteration on register set P: multiply dY row r - 2 and, behind its first MFMAs, wait for the set's row, commit it and
      // re-request the set four rows ahead
      auto iteration = [&](auto P) {
        const bool more = cur_valid(lc);
        const int y = cc.r - 2;  // X rows y - 1, y, y + 1 = ring slots (r - 3 .. r - 1) & 3 are complete
        if (y >= cc.ya && y < cc.yb) {
          const unsigned za = a_lane + ((unsigned)(y & 3) * WDS_SLAB) * 4u;
          unsigned xb[3];
#pragma unroll
          for (int ky = 0; ky < 3; ++ky) xb[ky] = b_lane + ((unsigned)((y - 1 + ky) & 3) * WDS_XROW) * 4u;
          // operands of k-step s (8 pixels): A = dY, B[t] = X of tap T0 + t; the reads of step s + 1 are issued before the MFMAs of s
          i32x4 A[2], B[2][NT];
          auto fetch = [&](int buf, int s) {
            A[buf] = *(lds_q *)(size_t)(za + s * (2 * 64 * 16));
#pragma unroll
            for (int t = 0; t < NT; ++t) {
              const int ky = (T0 + t) / 3, kx = (T0 + t) - 3 * ky;
              B[buf][t] = *(lds_q *)(size_t)(xb[ky] + kx * (WDS_SLAB * 4) + s * (2 * 64 * 16));
            }
          };
          fetch(0, 0);
#pragma unroll
          for (int s = 0; s < 4; ++s) {
            const int cur = s & 1;
            if (s + 1 < 4) fetch(cur ^ 1, s + 1);
            i32x4 Ar;
#pragma unroll
            for (int i = 0; i < 4; ++i) Ar[i] = __builtin_amdgcn_alignbit(A[cur][i], A[cur][i], 16);  // (hi, lo) -> (lo, hi): the cross terms
#pragma unroll
            for (int t = 0; t < 2 * NT; ++t) {
              const int tt = t < NT ? t : t - NT;
              acc[tt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, t < NT ? A[cur] : Ar), __builtin_bit_cast(f16x8, B[cur][tt]), acc[tt], 0, 0, 0);
              if (s == 0 && t < 8) {  // one staging piece behind each of the first eight MFMAs (2 NT >= 8)
                __builtin_amdgcn_sched_barrier(0);
                stage_piece(P, t, cc.r, more, lc);
                __builtin_amdgcn_sched_barrier(0);
              }
            }
          }
        } else {
#pragma unroll
          for (int k = 0; k < 8; ++k) stage_piece(P, k, cc.r, more, lc);
        }
        cur_next(cc);
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");  // LDS only: the loads in flight target registers
      };
      // rows of this split + three staging-only iterations per column segment, rounded up to the unrolling (the iterations past the
      // last row request nothing, commit zeros into ring slots nobody reads any more and multiply nothing)
      const int n_seg = (R1 - 1) / a.h - R0 / a.h + 1;
      const int n_iter = (R1 - R0) + 3 * n_seg;
#pragma unroll 1
      for (int it = 0; it < n_iter; it += 4) {
        iteration(S0{});
        iteration(S1{});
        iteration(S2{});
        iteration(S3{});
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (the requests past the last row: nothing stays in flight)
    }
    // ---- partial dW of this split: the wave's taps of its 32 x 32 block (C layout: column j = ci, rows (r & 3) + 8 (r >> 2) + 4 half = co)
    const int ci_o = ci_blk + wn * 32 + j;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int co_o = co_blk + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
      if (co_o < a.co && ci_o < ci_total) {
        float *dst = out + ((int64_t)co_o * ci_total + ci_o) * 9 + T0;
#pragma unroll
        for (int t = 0; t < NT; ++t) dst[t] = acc[t][r] * unscale;
      }
    }
  };
  if (ph == 0) run(std::integral_constant<int, 0>{});
  else run(std::integral_constant<int, 1>{});

#ifdef WDS_CLOCK
  {
    unsigned long long t1;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
    if (threadIdx.x == 0) atomicMax(&wds_clock_cycles, t1 - wds_t0);
  }
#endif
  // ---- bias gradient partial of this split: the eight quads of a channel sit in lanes (lane & 7) + 8 q
  if (a.want_db && ci_blk == 0) {
    float s = bsum;
    s += __shfl_xor(s, 8, 64);
    s += __shfl_xor(s, 16, 64);
    s += __shfl_xor(s, 32, 64);
    if (q == 0 && valid_co) (a.ws + (int64_t)a.splits * a.co * ci_total * 9)[(int64_t)split * a.co + co_s] = s;
  }
}

// OPT-IN (EDVR_WGRAD_DIRECT_SPLIT=1).  Measured on MI355X (profiles/r6/wgrad_direct_*.log): bit-for-bit the accuracy class of the
// Winograd-domain split kernel (3e-7 vs fp64), but 0.77-0.85 ms against 0.66-0.75 ms on the 160 x 128 x 64 x 64 layer.  Its staging
// is as cheap as designed (no measurable cost once interleaved) and its rows are four steps ahead - what it pays for is the 2.25x
// matrix work: under this kernel the core clock is 1.45-1.57 GHz (s_memtime against wall time; 2.0 GHz in a bare MFMA loop, 1.72 GHz under
// the split F(4x4) kernel), i.e. the f16 matrix pipe is power-limited to ~1.6 PFLOP/s, and the 2304 matrix cycles per step run at
// 66 % occupancy of that.  Trading vector instructions for matrix instructions does not pay on this part; kept as a measured alternative.
bool wgrad_direct_split_enabled() {
  static const bool on = []() {
    const char *e = getenv("EDVR_WGRAD_DIRECT_SPLIT");  // "1": this kernel instead of the Winograd-domain split kernel
    return e && e[0] == '1';
  }();
  return on;
}

// what the kernel needs beyond winograd_wgrad_plan(): 16-byte rows
bool wgrad_direct_split_supported(const float *x1, const float *x2, const float *dz, int h, int w, int64_t x1_img_stride, int64_t x2_img_stride,
                                  int64_t dz_img_stride) {
  auto aligned = [](const void *p, int64_t img_stride) { return !p || ((reinterpret_cast<uintptr_t>(p) & 15) == 0 && (img_stride & 3) == 0); };
  return (w & 3) == 0 && h >= 1 && aligned(x1, x1_img_stride) && aligned(x2, x2_img_stride) && aligned(dz, dz_img_stride);
}

int wgrad_direct_split_launch(const float *x1, const float *x2, const float *dz, float *ws, int c1, int c2, int n, int h, int w, int co,
                              int64_t x1_img_stride, int64_t x2_img_stride, int x2_div, int x2_mul, int x2_add, int64_t dz_img_stride,
                              int splits, int want_db, const float *x_amax, const float *dz_amax, hipStream_t stream) {
  WgradDirectSArgs a;
  a.x_amax = x_amax; a.dz_amax = dz_amax;
  a.want_db = want_db;
  a.x1 = x1; a.x2 = x2; a.dz = dz; a.ws = ws;
  a.c1 = c1; a.c2 = c2; a.n = n; a.h = h; a.w = w; a.co = co;
  a.x1_img_stride = x1_img_stride; a.x2_img_stride = x2_img_stride; a.dz_img_stride = dz_img_stride;
  a.x2_div = x2_div; a.x2_mul = x2_mul; a.x2_add = x2_add;
  a.cw = cdiv(w, 32);
  a.total_rows = n * a.cw * h;
  a.splits = splits;  // (<= total_rows: a split without rows would leave its partial unwritten - the caller clamps)
  a.ci_blocks = cdiv(c1 + c2, 64);
  a.co_blocks = cdiv(co, 64);
  hipLaunchKernelGGL(conv3x3_wgrad_direct_split_kernel, dim3(splits * a.ci_blocks * a.co_blocks), dim3(512), 0, stream, a);
  return check_launch("conv3x3_wgrad_direct_split_kernel");
}

}  // namespace edvr

#ifdef WDS_CLOCK
extern "C" int edvr_wds_clock_read(unsigned long long *host) {
  int rc = (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(edvr::wds_clock_cycles), sizeof(unsigned long long));
  static unsigned long long zero = 0;
  if (rc == 0) rc = (int)hipMemcpyToSymbol(HIP_SYMBOL(edvr::wds_clock_cycles), &zero, sizeof(zero));
  return rc;
}
#endif
