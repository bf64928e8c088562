// winograd.hip - 3x3 / stride-1 convolution as Winograd F(2x2, 3x3) on the fp32 matrix cores (gfx950).
//
// The direct kernel (conv2d.hip) is MFMA-bound at ~75-85 % of the fp32 peak, so the remaining lever in exact
// fp32 arithmetic is to issue fewer MFMAs: F(2x2,3x3) needs 16 multiplies per 2x2 output tile and channel
// pair instead of 36 (2.25x fewer), the same algorithm cuDNN selects for these layers of the reference under
// torch.backends.cudnn.benchmark = True (basicsr/train.py:132).  Everything is fused in one launch:
//
//   V = B^T d B   input transform of each 4x4 patch, done in registers when the chunk is staged
//   M[xi] = sum_ci U[xi][co, ci] * V[xi][ci, tile]      16 independent GEMMs on v_mfma_f32_32x32x2_f32
//   Y = A^T M A   output transform + bias / activation / residuals / PixelShuffle in the epilogue
//   U = G g G^T   precomputed per weight version by edvr_conv2d_pack_weight_f32 (layout [ci][xi][co_pad])
//
// Work split: a 256-thread workgroup (4 waves, one per SIMD, full 512-register budget) owns 64 output
// channels x 64 tiles (4 x 16 tiles = the same 8 x 32 pixel tile as the direct kernel).  Wave w owns the four
// transform positions xi = 4w..4w+3 for all 64 x 64 outputs: 4 x (2 co-tiles x 2 tile-groups) accumulator
// tiles = 256 registers.  Per chunk of 8 input channels: U slab (8 x 16 x 64) and V slab (8 x 16 x 64) in LDS
// (64 KB), operands are single ds_read_b32 at base + immediate, global loads of chunk c+1 are issued before the
// MFMA block of chunk c (register-prefetch pipeline, all loads unconditional).  The 16 positions of one output
// live in 4 different waves, so the output transform goes through LDS in four 64-KB pieces.
#include <cstdlib>
#include <type_traits>

#include "common.h"

namespace edvr {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

struct WinoArgs {
  edvr_conv2d_desc d;
  const float *U;  // [ci_pad][16][cop]
  int ci, cop, tiles_x, tiles_y, items;
};

// MFMA with the accumulator PINNED in the accumulator file ("+a"): with 256 accumulator registers per wave hipcc
// otherwise copies them between AGPRs and VGPRs around every loop iteration (216 v_accvgpr_write + 289
// v_accvgpr_read per 64 MFMAs, plus scratch spills - measured 4x slower than the direct kernel).  Operands come
// straight from ds_read (the compiler waits lgkmcnt for asm inputs); s_nop 1 covers a VALU-written operand.
__device__ __forceinline__ void mfma_acc(f32x16 &acc, float a, float b) {
  asm volatile("s_nop 1\n\tv_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(b));
}

__device__ __forceinline__ float wino_act(float v, int act) {
  if (act == EDVR_ACT_LRELU) return v > 0.f ? v : 0.1f * v;
  if (act == EDVR_ACT_RELU) return fmaxf(v, 0.f);
  if (act == EDVR_ACT_SIGMOID) return __builtin_amdgcn_rcpf(1.f + __expf(-v));
  return v;
}

__global__ __launch_bounds__(256, 1) void conv3x3_winograd_kernel(const WinoArgs a) {
  constexpr int CK = 8, COB = 64, TY = 4, TX = 16, TILES = TY * TX;  // 64 tiles = 8 x 32 output pixels
  constexpr int SLAB = CK * 16 * 64;                                 // floats per LDS slab (U or V)
  constexpr int SMEM = 4 * SLAB;  // two (U, V) slab pairs = 128 KB
  __shared__ __attribute__((aligned(16))) float smem[SMEM];

  const edvr_conv2d_desc &d = a.d;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, half = lane >> 5, j = lane & 31;
  const int hw = d.h * d.w;
  // Persistent workgroups: with 128 KB of LDS only one workgroup fits a CU, so a one-tile-per-workgroup grid pays the
  // full dispatch / teardown turnaround per tile with nothing to hide it (measured ~50k cycles per tile).  Each
  // workgroup instead walks items = (image, spatial tile, 64-channel block), channel block fastest so the co-blocks of
  // one tile run back to back and share its input through L2.
  // ---- per-item state (mutable: the NEXT item is set up, and its first loads issued, before the epilogue of the
  //      current one stores its outputs - vmcnt is in-order, so loads issued after the stores would wait for them)
  const int co_blocks = (d.co + 63) / 64;
  const int u_row0 = tid >> 4, u_c4 = tid & 15;
  const int p_tile = tid & 63, p_ty = p_tile >> 4, p_tx = p_tile & 15, p_ch = (tid >> 6) * 2;
  int co_blk = 0, img = 0, ty0 = 0, tx0 = 0;
  const float *x1 = d.x1, *x2 = d.x1, *u_src0 = a.U;
  // per-thread patch geometry of the item: 16 element offsets inside a channel plane (clamped to 0 where the patch
  // leaves the image) and 16 multipliers (1 inside, 0 outside): the loop then needs no address or mask math
  int p_off[16];
  float p_mul[16];
  auto setup = [&](int item) {
    co_blk = (item % co_blocks) * 64;
    const int tile_blk = (item / co_blocks) % (a.tiles_x * a.tiles_y);
    img = item / (co_blocks * a.tiles_x * a.tiles_y);
    ty0 = (tile_blk / a.tiles_x) * (2 * TY);  // output-pixel origin
    tx0 = (tile_blk % a.tiles_x) * (2 * TX);
    x1 = d.x1 + (int64_t)img * d.x1_img_stride;
    x2 = x1;
    if (d.x2) {
      const int i2 = d.x2_div > 0 ? (img / d.x2_div) * d.x2_mul + d.x2_add : img;
      x2 = d.x2 + (int64_t)i2 * d.x2_img_stride;
    }
    u_src0 = a.U + (int64_t)u_row0 * a.cop + co_blk + u_c4 * 4;
    const int gy0 = ty0 + 2 * p_ty - 1, gx0 = tx0 + 2 * p_tx - 1;  // top-left of the 4x4 patch (pad 1)
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const bool ok = gy0 + r >= 0 && gy0 + r < d.h && gx0 + c >= 0 && gx0 + c < d.w;
#ifdef WINO_EXP_ONEADDR
        p_off[r * 4 + c] = 0;  /* ablation only: every patch load hits one cached line */
#else
        p_off[r * 4 + c] = ok ? (gy0 + r) * d.w + gx0 + c : 0;
#endif
        p_mul[r * 4 + c] = ok ? 1.f : 0.f;
      }
  };
  const int wm = wave >> 1, wn = wave & 1;  // wave -> one (co tile, tile group) quadrant, ALL 16 transform positions
  const int p_ch_u = __builtin_amdgcn_readfirstlane(p_ch);  // wave-uniform: channel bases stay in SGPRs
  f32x16 acc[16];  // [xi]: 256 accumulator registers, pinned in the AGPR file by mfma_acc

  // ---- 2-deep software pipeline, one barrier per chunk:
  //   iteration k (parity P = k & 1):  MFMA block on LDS pair P (chunk k)
  //                                    || transform + commit of register set 1-P (chunk k+1) into LDS pair 1-P
  //                                    || global loads of chunk k+2 into register set P (just freed)
  // The commit is sliced over the 16 MFMA groups so its VALU / ds_write work issues in the shadow of the MFMAs
  // (one wave per SIMD: nothing else would hide it).  Chunk indices past the end are clamped (harmless re-staging).
  f32x4 ur[2][8];
  float pr[2][2][16];
  const int c_last = ((a.ci - 1) / CK) * CK;
  // Pointers of the chunk being loaded (hoisted out of the 16 slices): U slab row base and the two channel planes
  const float *ld_u = nullptr, *ld_p[2] = {nullptr, nullptr};
  auto load_begin = [&](int c0) {
    c0 = c0 <= c_last ? c0 : c_last;
    ld_u = u_src0 + (int64_t)c0 * 16 * a.cop;
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int c = c0 + p_ch_u + k;
      const int cc = c < a.ci ? c : a.ci - 1;  // channels past ci meet all-zero U rows: any finite data works
      ld_p[k] = (cc < d.c1) ? (x1 + (int64_t)cc * hw) : (x2 + (int64_t)(cc - d.c1) * hw);
    }
  };
  // slice g (0..15) of loading into register set S: U vector g (g < 8) and two patch elements (raw; the 0/1 border
  // mask is applied at transform time)
  auto load_slice = [&](auto SET, int g) {
    constexpr int S = decltype(SET)::value;
#ifdef WINO_EXP_NOULOAD
    if (g < 8) ur[S][g] = f32x4{1.f, 2.f, 3.f, 4.f};  /* ablation only */
#else
    if (g < 8) ur[S][g] = *reinterpret_cast<const f32x4 *>(ld_u + (int64_t)g * 16 * a.cop);
#endif
    const int k = g >> 3, e0 = (g & 7) * 2;
    pr[S][k][e0] = ld_p[k][p_off[e0]];
    pr[S][k][e0 + 1] = ld_p[k][p_off[e0 + 1]];
  };
  auto load_set = [&](auto SET, int c0) {
    load_begin(c0);
#pragma unroll
    for (int g = 0; g < 16; ++g) load_slice(SET, g);
  };
  // Commit schedule over the 16 MFMA groups (balanced, <= 12 VALU + 3 LDS writes per group):
  //   g 0-3 : column g of B^T d for patch 0            g 4-7 : column g-4 for patch 1, v[0..7] of patch 0
  //   g 8-11: v[8..15] of patch 0, v[0..7] of patch 1  g 12-15: v[8..15] of patch 1        U vector g for g < 8
  float tt[2][16];  // B^T d of the two patches being committed
  auto transform_col = [&](auto SET, int k, int c) {
    constexpr int S = decltype(SET)::value;
    const float d0 = pr[S][k][0 * 4 + c] * p_mul[0 * 4 + c], d1 = pr[S][k][1 * 4 + c] * p_mul[1 * 4 + c];
    const float d2 = pr[S][k][2 * 4 + c] * p_mul[2 * 4 + c], d3 = pr[S][k][3 * 4 + c] * p_mul[3 * 4 + c];
    tt[k][0 * 4 + c] = d0 - d2;
    tt[k][1 * 4 + c] = d1 + d2;
    tt[k][2 * 4 + c] = d2 - d1;
    tt[k][3 * 4 + c] = d1 - d3;
  };
  // element xi = r*4 + c of (B^T d) B from row r of tt[k]
  auto v_elem = [&](int k, int xi) {
    const int r = xi >> 2, c = xi & 3;
    const float *t = tt[k] + r * 4;
    return c == 0 ? t[0] - t[2] : c == 1 ? t[1] + t[2] : c == 2 ? t[2] - t[1] : t[1] - t[3];
  };
  auto commit_slice = [&](auto SET, int dst, int g) {
    constexpr int S = decltype(SET)::value;
    float *Us = smem + dst * 2 * SLAB, *Vs = Us + SLAB;
    if (g < 8) *reinterpret_cast<f32x4 *>(Us + (tid + g * 256) * 4) = ur[S][g];
    if (g < 4) transform_col(SET, 0, g);
    else if (g < 8) transform_col(SET, 1, g - 4);
    if (g >= 4 && g < 12) {  // patch 0: two values per group
      const int x0 = (g - 4) * 2;
      Vs[(p_ch * 16 + x0) * 64 + p_tile] = v_elem(0, x0);
      Vs[(p_ch * 16 + x0 + 1) * 64 + p_tile] = v_elem(0, x0 + 1);
    }
    if (g >= 8) {  // patch 1
      const int x0 = (g - 8) * 2;
      Vs[((p_ch + 1) * 16 + x0) * 64 + p_tile] = v_elem(1, x0);
      Vs[((p_ch + 1) * 16 + x0 + 1) * 64 + p_tile] = v_elem(1, x0 + 1);
    }
  };
  const int abase = half * 16 * 64 + wm * 32 + j;  // A operand (U): channel `half` of the pair, this wave's co tile
  const int bbase = half * 16 * 64 + wn * 32 + j;  // B operand (V): this wave's tile group
  auto iteration = [&](auto PAR, int c0) {
    constexpr int P = decltype(PAR)::value;
    using Other = std::integral_constant<int, 1 - P>;
    const float *Us = smem + P * 2 * SLAB, *Vs = Us + SLAB;
    // 16 groups (channel pair cp = g >> 2, positions xi = 4*(g & 3) .. +3) of 4 MFMAs; operands of group g+1 are fetched
    // before the MFMAs of group g; slice g of the loads (chunk k+2) and of the commit (chunk k+1) issue in their shadow
    load_begin(c0 + 2 * CK);  // chunk k+2 -> the register set freed by the previous iteration
    float av[2][4], bv[2][4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      av[0][i] = Us[abase + i * 64];
      bv[0][i] = Vs[bbase + i * 64];
    }
#pragma unroll
    for (int g = 0; g < 16; ++g) {
      const int cur = g & 1, nxt = cur ^ 1;
      if (g + 1 < 16) {  // (fetching two groups ahead instead of one measured no gain)
        const int cpn = (g + 1) >> 2, x0n = ((g + 1) & 3) * 4;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          av[nxt][i] = Us[abase + (2 * cpn * 16 + x0n + i) * 64];
          bv[nxt][i] = Vs[bbase + (2 * cpn * 16 + x0n + i) * 64];
        }
      }
      const int x0 = (g & 3) * 4;
#pragma unroll
      for (int i = 0; i < 4; ++i) mfma_acc(acc[x0 + i], av[cur][i], bv[cur][i]);
      load_slice(PAR, g);
      commit_slice(Other{}, 1 - P, g);
    }
    // LDS-only barrier: __syncthreads() would also wait vmcnt(0), i.e. for the chunk k+2 loads issued a few cycles ago
    // (full memory latency exposed every chunk).  Those loads target registers and need no cross-wave ordering.
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
  };
  using S0 = std::integral_constant<int, 0>;
  using S1 = std::integral_constant<int, 1>;
  // Persistent workgroups: with 128 KB of LDS only one workgroup fits a CU; each walks items = (image, spatial tile,
  // 64-channel block), channel block fastest so the co-blocks of one tile run back to back and share its input via L2.
  // XCD-aware walk: workgroup b runs on XCD b % 8, each XCD with its own 4 MB L2.  Handing every XCD one CONTIGUOUS
  // range of items keeps the workgroups that share input lines (the co-blocks of a tile, the x/y-neighbouring tiles whose
  // 128-byte lines and halo rows overlap) on one L2 at the same time; a plain `item = b + k*grid` walk spreads them over
  // all 8 L2s and every one of them re-fetches the lines from the fabric (measured 7.2x the output bytes, now ~1.6x).
  const int n_xcd = gridDim.x < 8 ? 1 : 8;
  const int xcd = n_xcd == 1 ? 0 : (int)blockIdx.x % 8, xcd_rank = n_xcd == 1 ? (int)blockIdx.x : (int)blockIdx.x / 8;
  const int xcd_wgs = n_xcd == 1 ? (int)gridDim.x : ((int)gridDim.x - xcd + 7) / 8;
  const int chunk = (a.items + n_xcd - 1) / n_xcd;
  const int item_end = min(a.items, (xcd + 1) * chunk);
  const int item_first = xcd * chunk + xcd_rank;
  if (item_first >= item_end) return;
  setup(item_first);
  load_set(S0{}, 0);
  for (int item = item_first; item < item_end; item += xcd_wgs) {
#pragma unroll
    for (int xi = 0; xi < 16; ++xi)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[xi][r] = 0.f;
#pragma unroll
    for (int g = 0; g < 16; ++g) commit_slice(S0{}, 0, g);
    load_set(S1{}, CK);
    __syncthreads();
    // ci is a multiple of 2*CK (checked by winograd_eligible), so the loop body is branch-free: with loop-carried
    // accumulators in the AGPR file any control flow inside the loop makes hipcc copy all 256 of them to VGPRs and back.
#pragma unroll 1
    for (int c0 = 0; c0 < a.ci; c0 += 2 * CK) {
      iteration(S0{}, c0);
      iteration(S1{}, c0 + CK);
    }
    // geometry of the item being finished; then set up the NEXT item and issue its first loads ahead of the stores
    const int e_img = img, e_ty0 = ty0, e_tx0 = tx0, e_co_blk = co_blk;
    {
      const int next = item + xcd_wgs;
      setup(next < item_end ? next : item);
      load_set(S0{}, 0);
    }

  asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");  // MFMA results -> v_accvgpr_read: hipcc does not pad around inline asm
  // ---- output transform Y = A^T M A, in registers: lane (half, j) holds tile wn*32 + j and 16 output channels
  //      co_blk + wm*32 + (r&3) + 8*(r>>2) + 4*half, each with its 16 positions acc[xi][r]
  const int plane = d.h * d.w;  // stride 1, pad 1: output size == input size
  float *y = d.y + (int64_t)e_img * d.y_img_stride;
  const float *r1 = d.res1 ? d.res1 + (int64_t)e_img * d.res1_img_stride : nullptr;
  const float *r2 = d.res2 ? d.res2 + (int64_t)e_img * d.res2_img_stride : nullptr;
  const int tile = wn * 32 + j, tyy = tile >> 4, txx = tile & 15;
  const int oy = e_ty0 + 2 * tyy, ox = e_tx0 + 2 * txx;
  const int co_lane = e_co_blk + wm * 32 + 4 * half;
  // Every uniform condition is resolved ONCE (compile-time variants below): evaluated per output element they become
  // ~1400 scalar branches per workgroup and made this epilogue cost as much as six chunks of the main loop.
  const float slope = d.act == EDVR_ACT_LRELU ? 0.1f : (d.act == EDVR_ACT_RELU ? 0.f : 1.f);  // none/relu/lrelu = max(v, slope*v)
  const bool interior = e_ty0 + 2 * TY <= d.h && e_tx0 + 2 * TX <= d.w && e_co_blk + 64 <= d.co && (d.w & 1) == 0;
  auto emit = [&](auto HAS_RES, auto SHUFFLE, auto SIGMOID, auto INTERIOR) {
    constexpr bool RES = decltype(HAS_RES)::value, SHF = decltype(SHUFFLE)::value, SIG = decltype(SIGMOID)::value,
                   INT = decltype(INTERIOR)::value;
#pragma unroll
    for (int rh = 0; rh < 2; ++rh) {  // two batches of 8 channels: bias / residual loads are issued ahead of their use
      float bias_r[8], rr[8][2][2];
#pragma unroll
      for (int ri = 0; ri < 8; ++ri) {
        const int r = rh * 8 + ri, co = co_lane + (r & 3) + 8 * (r >> 2);
        const int cc = INT ? co : (co < d.co ? co : d.co - 1);
        bias_r[ri] = d.bias ? d.bias[cc] : 0.f;
        if (RES) {
#pragma unroll
          for (int yy = 0; yy < 2; ++yy)
#pragma unroll
            for (int xx = 0; xx < 2; ++xx) {
              const bool ok = INT || (co < d.co && oy + yy < d.h && ox + xx < d.w);
              const int off = ok ? cc * plane + (oy + yy) * d.w + ox + xx : 0;
              float v = r1[off];
              if (r2) v += r2[off];
              rr[ri][yy][xx] = ok ? v : 0.f;
            }
        }
      }
#pragma unroll
      for (int ri = 0; ri < 8; ++ri) {
        const int r = rh * 8 + ri, co = co_lane + (r & 3) + 8 * (r >> 2);
        float s0[4], s1[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {  // A^T M
          s0[c] = acc[0 * 4 + c][r] + acc[1 * 4 + c][r] + acc[2 * 4 + c][r];
          s1[c] = acc[1 * 4 + c][r] - acc[2 * 4 + c][r] - acc[3 * 4 + c][r];
        }
        float o[2][2];
        o[0][0] = s0[0] + s0[1] + s0[2];
        o[0][1] = s0[1] - s0[2] - s0[3];
        o[1][0] = s1[0] + s1[1] + s1[2];
        o[1][1] = s1[1] - s1[2] - s1[3];
        const float sl = (co >= d.act_from) ? slope : 1.f;  // per-lane select, no branch
#pragma unroll
        for (int yy = 0; yy < 2; ++yy)
#pragma unroll
          for (int xx = 0; xx < 2; ++xx) {
            float v = o[yy][xx] + bias_r[ri];
            if (SIG) v = (co >= d.act_from) ? __builtin_amdgcn_rcpf(1.f + __expf(-v)) : v;
            else v = fmaxf(v, sl * v);
            if (RES) v += rr[ri][yy][xx];
            o[yy][xx] = v;
          }
        if (SHF) {
#pragma unroll
          for (int yy = 0; yy < 2; ++yy)
#pragma unroll
            for (int xx = 0; xx < 2; ++xx)
              if (INT || (co < d.co && oy + yy < d.h && ox + xx < d.w))
                y[(co >> 2) * plane * 4 + (2 * (oy + yy) + ((co >> 1) & 1)) * (2 * d.w) + 2 * (ox + xx) + (co & 1)] = o[yy][xx];
        } else if (INT) {
#pragma unroll
          for (int yy = 0; yy < 2; ++yy)  // 16 lanes x 8 B = one 128-B line per row
            *reinterpret_cast<f32x2 *>(y + co * plane + (oy + yy) * d.w + ox) = f32x2{o[yy][0], o[yy][1]};
        } else {
#pragma unroll
          for (int yy = 0; yy < 2; ++yy)
#pragma unroll
            for (int xx = 0; xx < 2; ++xx)
              if (co < d.co && oy + yy < d.h && ox + xx < d.w) y[co * plane + (oy + yy) * d.w + ox + xx] = o[yy][xx];
        }
      }
    }
  };
  using T = std::true_type;
  using F = std::false_type;
  if (d.act == EDVR_ACT_SIGMOID) {
    emit(F{}, F{}, T{}, F{});
  } else if (d.out_mode == EDVR_OUT_PIXEL_SHUFFLE2) {
    if (interior) emit(F{}, T{}, F{}, T{}); else emit(F{}, T{}, F{}, F{});
  } else if (r1) {
    if (interior) emit(T{}, F{}, F{}, T{}); else emit(T{}, F{}, F{}, F{});
  } else {
    if (interior) emit(F{}, F{}, F{}, T{}); else emit(F{}, F{}, F{}, F{});
  }
  }  // persistent item loop
}

// U[ci][xi][cop] = (G g G^T)[xi] of the 3x3 kernel g = w[co][ci] (or the data-gradient kernel when transpose_flip)
__global__ void winograd_weight_kernel(const float *__restrict__ w, float *__restrict__ U, int co, int ci, int cop, int cip,
                                       int transpose_flip) {
  const int64_t total = (int64_t)cip * cop;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int o = (int)(i % cop), c = (int)(i / cop);
    float g[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      float v = 0.f;
      if (o < co && c < ci) v = transpose_flip ? w[((int64_t)c * co + o) * 9 + (8 - t)] : w[((int64_t)o * ci + c) * 9 + t];
      g[t] = v;
    }
    float tmp[12];
#pragma unroll
    for (int jx = 0; jx < 3; ++jx) {  // G g
      tmp[0 * 3 + jx] = g[0 * 3 + jx];
      tmp[1 * 3 + jx] = 0.5f * (g[0 * 3 + jx] + g[1 * 3 + jx] + g[2 * 3 + jx]);
      tmp[2 * 3 + jx] = 0.5f * (g[0 * 3 + jx] - g[1 * 3 + jx] + g[2 * 3 + jx]);
      tmp[3 * 3 + jx] = g[2 * 3 + jx];
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {  // (G g) G^T
      float *dst = U + ((int64_t)c * 16 + r * 4) * cop + o;
      dst[0 * (int64_t)cop] = tmp[r * 3 + 0];
      dst[1 * (int64_t)cop] = 0.5f * (tmp[r * 3 + 0] + tmp[r * 3 + 1] + tmp[r * 3 + 2]);
      dst[2 * (int64_t)cop] = 0.5f * (tmp[r * 3 + 0] - tmp[r * 3 + 1] + tmp[r * 3 + 2]);
      dst[3 * (int64_t)cop] = tmp[r * 3 + 2];
    }
  }
}

bool winograd_eligible(const edvr_conv2d_desc &d) {
  static const bool enabled = []() {
    const char *e = getenv("EDVR_CONV_WINOGRAD");  // "0": always use the direct kernel (A/B, fallback)
    return !(e && e[0] == '0');
  }();
  // the epilogue is specialised for: plain | residual(s) | pixel-shuffle | sigmoid - other combinations use the direct kernel
  const bool has_res = d.res1 || d.res2;
  if ((d.res2 && !d.res1) || (d.act == EDVR_ACT_SIGMOID && (has_res || d.out_mode != EDVR_OUT_NCHW)) ||
      (d.out_mode != EDVR_OUT_NCHW && has_res))
    return false;
  if (d.algo == EDVR_CONV_DIRECT) return false;
  const bool applicable = d.ks == 3 && d.stride == 1 && (d.c1 + d.c2) % 16 == 0;
  if (d.algo == EDVR_CONV_WINOGRAD) return applicable;  // explicit request: any size the kernel can do
  return enabled && applicable && d.co >= 48 && d.w > 16 && d.h >= 4;  // auto: only where it beats the direct kernel
}

int winograd_launch(const edvr_conv2d_desc &d, const float *U, int cop, hipStream_t stream) {
  WinoArgs a;
  a.d = d;
  a.U = U;
  a.ci = d.c1 + d.c2;
  a.cop = cop;
  a.tiles_x = cdiv(d.w, 32);
  a.tiles_y = cdiv(d.h, 8);
  a.items = a.tiles_x * a.tiles_y * cdiv(d.co, 64) * d.n;
  static const int n_cu = []() {
    int dev = 0, n = 256;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) n = prop.multiProcessorCount;
    return n;
  }();
  hipLaunchKernelGGL(conv3x3_winograd_kernel, dim3(std::min(a.items, n_cu)), dim3(256), 0, stream, a);
  return check_launch("conv3x3_winograd_kernel");
}

int winograd_pack(const float *w, float *U, int co, int ci, int cop, int cip, int transpose_flip, hipStream_t stream) {
  const int64_t total = (int64_t)cip * cop;
  hipLaunchKernelGGL(winograd_weight_kernel, dim3((unsigned)std::min<int64_t>(cdiv64(total, 256), 4096)), dim3(256), 0, stream, w, U, co, ci,
                     cop, cip, transpose_flip);
  return check_launch("winograd_weight_kernel");
}

}  // namespace edvr
