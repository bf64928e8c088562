// winograd4.hip - the Winograd F(2x2, 3x3) convolution of winograd.hip as FOUR-wave workgroups, two per CU (gfx950).
// EXPERIMENT, opt-in (EDVR_WINOGRAD_4WAVE=1): correct (same tests as winograd.hip).
//
// winograd.hip runs one 8-wave workgroup per CU: both waves of a SIMD belong to it and meet at the same barrier, so while an
// item's epilogue (output transform, sibling exchange, 64 KB of stores) and the pipeline refill run - 5.6 of 39 us per item at
// 128 input channels - the matrix pipe of the whole CU idles, and every barrier stall idles it too.  Here a workgroup has 4
// waves (wave = (ph, 32-channel half of the 64-channel block), 8 accumulator tiles = the same 256-register budget) and owns
// 64 output channels x 32 tiles (4 x 32 output pixels); TWO of them share a CU, one wave of each per SIMD, and nothing ties
// them together: one workgroup's epilogue, barrier waits and refill are the other's MFMA time.
//   * U (transformed weights): with this split every element of U is an MFMA A operand of exactly ONE wave of the workgroup
//     (its channel half, its two transform rows), so staging U through LDS is pure overhead.  The weights are packed in
//     operand order instead ([co block][channel pair][row][co half][lane][4 positions], winograd_weight_kernel) and each lane
//     fetches its next four A operands with one coalesced 16-byte buffer load (1 KB per wave) straight into a rotating set of
//     four register quads, half a chunk ahead of the MFMAs that consume them.  No LDS traffic, no barrier for U.
//   * V (transformed input) keeps the rotating-register pipeline of winograd.hip - thread = (channel of the 8-channel chunk,
//     tile of 32): loads of chunk k + 2, transform + commit of chunk k + 1 in the shadow of the MFMAs of chunk k - in a
//     double-buffered 2 x 16 KB slab, one barrier per chunk (LDS traffic only; the loads in flight target registers).
//   * The sibling exchange of the output transform has its own 32 KB, so the epilogue needs one barrier.
// The second dispatch wave of workgroups starts a good epilogue later (one-off s_sleep): with equal work per item a phase
// difference between the two workgroups of a CU persists, so their epilogues do not coincide.
//
// Earlier forms of this file, measured (n = 20, 128 -> 128, 180x320, bias + LeakyReLU; 8-wave kernel 1.72-1.75 ms):
//   U by LDS-DMA (global_load_lds_dwordx4, no staging registers)        1.95 ms   one CU's LDS-DMA path delivers ~25 GB/s
//   ... with the U fetch removed (wrong results; bound of the structure) 1.45 ms   the overlap works: 235 alg. TF/s = 0.66 executed
//   U through rotating registers into a double-buffered LDS slab        1.81 ms   twice the U staging per MFMA of the 8-wave kernel
// Same arithmetic and the same epilogue variants as winograd.hip (results are bit-identical).
#include <cstdlib>
#include <type_traits>

#include "common.h"

namespace edvr {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

struct Wino4Args {
  edvr_conv2d_desc d;
  const float *U;  // operand order: [co block 64][channel pair][row 4][co half 2][lane 64][4]
  int ci, ci_real, cop, tiles_x, tiles_y, items;  // ci: rounded up to 16 (U has all-zero rows there), ci_real = c1 + c2
  float ys, ys_gs;  // y_scale (0 -> 1) and y_scale * gate_slope, resolved on the host
};

__device__ __forceinline__ void mfma_acc4(f32x16 &acc, float a, float b) {
  acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
}

template <bool PAIR, bool GATE = false>
__global__ __launch_bounds__(256, 2) void conv3x3_winograd4_kernel(const Wino4Args a) {
  constexpr int CK = 8, TY = 2, TX = 16;  // V chunk channels; 32 tiles = 4 x 32 output pixels
  constexpr int VSLAB = CK * 16 * 32;      // floats per V stage (16 KB)
  __shared__ __attribute__((aligned(16))) float smem[2 * VSLAB + 4 * 2048];  // two V stages + the exchange area = 64 KB
  float *const Vs0 = smem, *const Vs1 = smem + VSLAB, *const Xs = smem + 2 * VSLAB;

  const edvr_conv2d_desc &d = a.d;
  const int tid = threadIdx.x, lane = tid & 63, half = lane >> 5, j = lane & 31;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave & 1, ph = wave >> 1;
  const int hw = d.h * d.w;
  const int co_blocks = (d.co + 63) / 64;
  const int p_ty = j >> 4, p_tx = j & 15;  // V staging: channel 2 wave + half of the chunk, tile j

  // ---- geometry of the item being LOADED (the pipeline loads one item ahead of the MFMAs at item boundaries)
  int co_blk = 0, img = 0, ty0 = 0, tx0 = 0;
  const float *x1 = d.x1, *x2 = d.x1;
  constexpr int RSRC_FLAGS = 0x00020000;  // raw buffer, 32-bit data format (gfx9 family)
  const int plane_bytes = hw * 4;
  auto uniform_rsrc = [&](const float *p, int bytes) {
    const uint64_t pv = reinterpret_cast<uint64_t>(p);
    const uint64_t pu = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(pv >> 32)) << 32) |
                        (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)pv);
    return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void *>(pu), (short)0, bytes, RSRC_FLAGS);
  };
  int p_off[16];
  auto setup = [&](int item) {
    co_blk = __builtin_amdgcn_readfirstlane((item % co_blocks) * 64);
    const int tile_blk = __builtin_amdgcn_readfirstlane((item / co_blocks) % (a.tiles_x * a.tiles_y));
    img = __builtin_amdgcn_readfirstlane(item / (co_blocks * a.tiles_x * a.tiles_y));
    ty0 = __builtin_amdgcn_readfirstlane((tile_blk / a.tiles_x) * (2 * TY));  // output-pixel origin
    tx0 = __builtin_amdgcn_readfirstlane((tile_blk % a.tiles_x) * (2 * TX));
    x1 = d.x1 + (int64_t)img * d.x1_img_stride;
    x2 = x1;
    if (d.x2) {
      const int i2 = d.x2_div > 0 ? (img / d.x2_div) * d.x2_mul + d.x2_add : img;
      x2 = d.x2 + (int64_t)i2 * d.x2_img_stride;
    }
    const int gy0 = ty0 + 2 * p_ty - 1, gx0 = tx0 + 2 * p_tx - 1;  // top-left of the 4x4 patch (pad 1)
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const bool ok = gy0 + r >= 0 && gy0 + r < d.h && gx0 + c >= 0 && gx0 + c < d.w;
        // the two half-waves stage two consecutive channel planes through ONE wave-uniform resource: + one plane for the upper half
        p_off[r * 4 + c] = ok ? ((gy0 + r) * d.w + gx0 + c) * 4 + half * plane_bytes : (int)0x80000000;
      }
  };

  f32x16 acc[8];  // tiles 0-3 = transform row row_lo, 4-7 = row_hi (below)
  float pr[16];   // raw patch of (channel 2 wave + half, tile j) of the chunk being staged
  float tt[16];   // B^T d
  f32x4 aq[4];    // A operands (U) of four MFMA groups: group g of a chunk uses aq[g & 3], reloaded for group g + 4 right after

  __amdgpu_buffer_rsrc_t ld_rsrc = uniform_rsrc(a.U, 0);
  auto load_begin = [&](int c0) {
    const int c = c0 + 2 * wave;  // even; c1 is even (host check), so the pair never straddles x1 / x2
    const float *pl = (c < d.c1) ? (x1 + (int64_t)c * hw) : (x2 + (int64_t)(c - d.c1) * hw);
    const int nvalid = a.ci_real - c;  // channels of the 16-padding: empty (or one-plane) buffer, their loads return 0
    ld_rsrc = uniform_rsrc(pl, nvalid >= 2 ? 2 * plane_bytes : (nvalid == 1 ? plane_bytes : 0));
  };
  auto load_col = [&](int c) {
    if (PAIR && c == 1) return;  // loaded together with column 2
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      if (PAIR && c == 2) {
        const f32x2 v = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(ld_rsrc, p_off[r * 4 + 1], 0, 0));
        pr[r * 4 + 1] = v[0];
        pr[r * 4 + 2] = v[1];
      } else {
        pr[r * 4 + c] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(ld_rsrc, p_off[r * 4 + c], 0, 0));
      }
    }
  };
  auto transform_col = [&](int c) {
    const float d0 = pr[0 * 4 + c], d1 = pr[1 * 4 + c], d2 = pr[2 * 4 + c], d3 = pr[3 * 4 + c];
    tt[0 * 4 + c] = d0 - d2;
    tt[1 * 4 + c] = d1 + d2;
    tt[2 * 4 + c] = d2 - d1;
    tt[3 * 4 + c] = d1 - d3;
  };
  auto commit_v_row = [&](float *Vs, int r) {  // positions xi = 4r .. 4r+3 of (B^T d) B; slab [channel 8][xi 16][tile 32]
    const float *t = tt + r * 4;
    float *dst = Vs + ((2 * wave + half) * 16 + r * 4) * 32 + j;
    dst[0 * 32] = t[0] - t[2];
    dst[1 * 32] = t[1] + t[2];
    dst[2 * 32] = t[2] - t[1];
    dst[3 * 32] = t[1] - t[3];
  };
  const int row_lo = ph ? 2 : 1, row_hi = ph ? 3 : 0;
  // A operands of group g = (channel pair g >> 1, row set g & 1) of the chunk at channel c0, output-channel block coblk:
  // one 1 KB block of the packed weights per wave, lane l reads 16 bytes at 16 l (positions 4 row .. 4 row + 3)
  const int u_bytes = a.ci * 16 * a.cop * 4, np = a.ci >> 1;
  const int u_voff = lane * 16;
  const __amdgpu_buffer_rsrc_t u_rsrc = uniform_rsrc(a.U, u_bytes);
  auto load_a = [&](int q, int c0, int g, int coblk) {
    const int row = (g & 1) ? row_hi : row_lo;
    const int soff = (((((coblk >> 6) * np + (c0 >> 1) + (g >> 1)) * 4 + row) * 2 + wm) * 1024);
    aq[q] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(u_rsrc, u_voff, soff, 0));
  };

  const int bbase[2] = {half * 16 * 32 + row_lo * 4 * 32 + j, half * 16 * 32 + row_hi * 4 * 32 + j};                      // B (V): the 32 tiles
  // One 8-channel chunk (V stage P): 8 groups (channel pair g >> 1, accumulator tiles 4 (g & 1) .. +3) of 4 MFMAs.  (cur_c,
  // cur_co): channel base / output-channel block of THIS chunk - its groups 4..7 are fetched while groups 0..3 run; (nxt_c,
  // nxt_co): the next chunk, whose groups 0..3 are fetched while groups 4..7 run.
  auto iteration = [&](auto PAR, int c_load, auto LOAD, int cur_c, int cur_co, int nxt_c, int nxt_co) {
    constexpr int P = decltype(PAR)::value;
    constexpr bool LD = decltype(LOAD)::value;  // false: the patch reloads are issued by the caller (end of an item)
    const float *Vs = P ? Vs1 : Vs0;
    float *Vd = P ? Vs0 : Vs1;
    // this chunk's V stage is complete; nobody reads the other stage any more
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    load_begin(c_load);
    float bv[2][4];
#pragma unroll
    for (int i = 0; i < 4; ++i) bv[0][i] = Vs[bbase[0] + i * 32];
#pragma unroll
    for (int g = 0; g < 8; ++g) {
      const int cur = g & 1, nxt = cur ^ 1;
      if (g + 1 < 8) {
        const int cpn = (g + 1) >> 1, hn = (g + 1) & 1;
#pragma unroll
        for (int i = 0; i < 4; ++i) bv[nxt][i] = Vs[bbase[hn] + (2 * cpn * 16 + i) * 32];
      }
      const int x0 = (g & 1) * 4;
#pragma unroll
      for (int i = 0; i < 4; ++i) mfma_acc4(acc[x0 + i], aq[g & 3][i], bv[cur][i]);
      // half a chunk ahead (a full chunk ahead - 8 quads - measured no faster: 1.66 vs 1.64 ms, and spills in the epilogue)
      if (g < 4) {
        load_a(g, cur_c, g + 4, cur_co);
        transform_col(g);
        if (LD) load_col(g);
      } else {
        load_a(g - 4, nxt_c, g - 4, nxt_co);
        commit_v_row(Vd, g - 4);
      }
      __builtin_amdgcn_sched_barrier(0);  // pin the slice schedule (winograd.hip)
    }
  };
  using S0 = std::integral_constant<int, 0>;
  using S1 = std::integral_constant<int, 1>;

  // Persistent workgroups, XCD-aware walk (winograd.hip): every XCD gets one contiguous range of items.
  const int n_xcd = gridDim.x < 8 ? 1 : 8;
  const int xcd = n_xcd == 1 ? 0 : (int)blockIdx.x % 8, xcd_rank = n_xcd == 1 ? (int)blockIdx.x : (int)blockIdx.x / 8;
  const int xcd_wgs = n_xcd == 1 ? (int)gridDim.x : ((int)gridDim.x - xcd + 7) / 8;
  const int chunk = (a.items + n_xcd - 1) / n_xcd;
  const int item_end = min(a.items, (xcd + 1) * chunk);
  const int item_first = xcd * chunk + xcd_rank;
  if (item_first >= item_end) return;
  // second dispatch wave (the workgroups that share a CU with an earlier one): start a good epilogue later
  if (2 * xcd_rank >= xcd_wgs && xcd_wgs > 1) {
#pragma unroll 1
    for (int i = 0; i < 3; ++i) __builtin_amdgcn_s_sleep(127);
  }

  // ---- prologue of the first item: U_A(0) on its way, V chunk 0 -> registers -> stage 0, chunk 1 -> registers
  setup(item_first);
#pragma unroll
  for (int g = 0; g < 4; ++g) load_a(g, 0, g, co_blk);
  load_begin(0);
#pragma unroll
  for (int c = 0; c < 4; ++c) load_col(c);
#pragma unroll
  for (int c = 0; c < 4; ++c) transform_col(c);
#pragma unroll
  for (int r = 0; r < 4; ++r) commit_v_row(Vs0, r);
  load_begin(CK);
#pragma unroll
  for (int c = 0; c < 4; ++c) load_col(c);
#pragma unroll
  for (int xi = 0; xi < 8; ++xi)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[xi][r] = 0.f;

  for (int item = item_first; item < item_end; item += xcd_wgs) {
    // a.ci is a multiple of 2*CK (rounded up by the host): chunk k (parity k & 1) loads chunk k+2
    const int e_img = img, e_ty0 = ty0, e_tx0 = tx0, e_co_blk = co_blk;
#pragma unroll 1
    for (int c0 = 2 * CK; c0 < a.ci; c0 += 2 * CK) {
      iteration(S0{}, c0, std::true_type{}, c0 - 2 * CK, e_co_blk, c0 - CK, e_co_blk);
      iteration(S1{}, c0 + CK, std::true_type{}, c0 - CK, e_co_blk, c0, e_co_blk);
    }
    // last two chunks of this item: their patch loads already belong to the NEXT item (or re-stage this one after the last)
    {
      const int next = item + xcd_wgs;
      setup(next < item_end ? next : item);
    }
    iteration(S0{}, 0, std::true_type{}, a.ci - 2 * CK, e_co_blk, a.ci - CK, e_co_blk);
    // The last chunk commits V chunk 0 of the next item and fetches its first A operands, but leaves the patch registers EMPTY
    // (the output transform needs them); chunk 1's patches are loaded after the exchange, under the stores.
    iteration(S1{}, CK, std::false_type{}, a.ci - CK, e_co_blk, 0, co_blk);

    // ---- output transform Y = A^T M A.  Lane (half, j) holds tile j and 16 output channels
    //      co_blk + wm*32 + (r&3) + 8*(r>>2) + 4*half, for rows 2ph, 2ph+1 of M: acc[rr*4 + c][r].
    //      t[row][jx] = (M A)[row][jx]; Y[0][jx] = t0 + t1 + t2, Y[1][jx] = t1 - t2 - t3.  Wave ph finishes output row ph:
    //      ph = 0 keeps t0 + t1 and needs t2 from its sibling, ph = 1 keeps -(t2 + t3) and needs t1.
    const int plane = d.h * d.w;  // stride 1, pad 1: output size == input size
    float *y = d.y + (int64_t)e_img * d.y_img_stride;
    // `gate` (data gradient through a ReLU / LeakyReLU: y *= gate > 0 ? 1 : gate_slope) rides on the residual machinery:
    // same tile, same prefetch, a select instead of an add (winograd_eligible rejects gate together with residuals).  GATE is its
    // own instantiation of the kernel, so the ungated one keeps exactly the code it had without the feature.  The gated one
    // deliberately keeps the whole run-time dispatch below (d.gate is tested, not assumed): with only its own two epilogue
    // variants left the compiler hoists their common part above the branch and spills 60-80 bytes per lane.
    const bool gated = GATE && d.gate != nullptr;
    const float *r1 = gated ? d.gate + (int64_t)e_img * d.gate_img_stride : (d.res1 ? d.res1 + (int64_t)e_img * d.res1_img_stride : nullptr);
    const float *r2 = (!gated && d.res2) ? d.res2 + (int64_t)e_img * d.res2_img_stride : nullptr;
    const int tile = j, tyy = tile >> 4, txx = tile & 15;  // 32 tiles = 2 x 16
    const int oy = e_ty0 + 2 * tyy + ph, ox = e_tx0 + 2 * txx;
    const int co_lane = e_co_blk + wm * 32 + 4 * half;
    // Every uniform condition is resolved ONCE (compile-time variants below): evaluated per output element they become
    // ~1400 scalar branches per workgroup and made this epilogue cost as much as six chunks of the main loop.
    const float slope = d.act == EDVR_ACT_LRELU ? 0.1f : (d.act == EDVR_ACT_RELU ? 0.f : 1.f);  // none/relu/lrelu = max(v, slope*v)
    const bool interior = e_ty0 + 2 * TY <= d.h && e_tx0 + 2 * TX <= d.w && e_co_blk + 64 <= d.co && (d.w & 1) == 0;
    // sibling exchange [wave][r][lane][2] = 8 KB per wave in its own area: the previous item's values were read many barriers ago
    float *xsend = Xs + wave * 2048;
    const float *xrecv = Xs + (wave ^ 2) * 2048;
    // ---- common to all epilogue variants (kept OUT of the specialised lambdas: hoisted above their dispatch by the
    //      compiler, the sums were spilled to scratch across the multi-way branch, ~70 scratch accesses per item)
    // Row pass (M A)[row][.] = (m0 + m1 + m2, m1 - m2 - m3) of the row to send (tiles 0-3), one accumulator tile at a
    // time: hipcc moves a tile out of the accumulator file as a whole 16-register tuple, so walking channel by channel
    // (8 tiles live at once) needs 128 VGPRs and spilled ~150 registers to scratch per item.
#ifdef WINO_EXP_NOEPI  /* ablation only: no output transform / exchange / stores; accumulators kept alive without instructions */
#pragma unroll
    for (int xi = 0; xi < 8; ++xi) asm volatile("" ::"v"(acc[xi]));
    load_begin(CK);
#pragma unroll
    for (int c = 0; c < 4; ++c) load_col(c);
#else
    // bias now, residuals right after the row pass (when the accumulators are dead): their latency hides behind the
    // transform and the exchange instead of being exposed once per batch of stores
    float bias_r[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) bias_r[r] = 0.f;
    if (d.bias) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int co = co_lane + (r & 3) + 8 * (r >> 2);
        bias_r[r] = d.bias[co < d.co ? co : d.co - 1];
      }
    }
    float sum[16][2];
#define WINO_ROWPASS(BASE, FIRST)                                                   \
  _Pragma("unroll") for (int r = 0; r < 16; ++r) {                                  \
    const float v = acc[BASE + 0][r];                                               \
    sum[r][0] = FIRST ? v : sum[r][0] + v;                                          \
  }                                                                                 \
  __builtin_amdgcn_sched_barrier(0);                                                \
  _Pragma("unroll") for (int r = 0; r < 16; ++r) {                                  \
    const float v = acc[BASE + 1][r];                                               \
    sum[r][0] += v;                                                                 \
    sum[r][1] = FIRST ? v : sum[r][1] + v;                                          \
  }                                                                                 \
  __builtin_amdgcn_sched_barrier(0);                                                \
  _Pragma("unroll") for (int r = 0; r < 16; ++r) {                                  \
    const float v = acc[BASE + 2][r];                                               \
    sum[r][0] += v;                                                                 \
    sum[r][1] -= v;                                                                 \
  }                                                                                 \
  __builtin_amdgcn_sched_barrier(0);                                                \
  _Pragma("unroll") for (int r = 0; r < 16; ++r) sum[r][1] -= acc[BASE + 3][r];     \
  __builtin_amdgcn_sched_barrier(0);
    WINO_ROWPASS(0, true)
#pragma unroll
    for (int r = 0; r < 16; ++r) {  // (scalar LDS writes: 64-bit pairs made hipcc spill the sums to form register tuples)
#ifdef WINO_EXP_NOXCHG
      if (sum[r][0] == 12345.f)  /* ablation only */
#endif
      {
        xsend[(2 * r) * 64 + lane] = sum[r][0];
        xsend[(2 * r + 1) * 64 + lane] = sum[r][1];
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    WINO_ROWPASS(4, false)  // + the row this wave keeps: sum = t1 + t0 (ph 0) or t2 + t3 (ph 1)
#undef WINO_ROWPASS
    const bool res_fast = r1 && interior && d.out_mode == EDVR_OUT_NCHW && d.act != EDVR_ACT_SIGMOID;
    f32x2 rr_all[16];  // first half here, second half at the start of the store phase (all 16 pairs did not fit: spills)
    auto load_res = [&](int r_lo) {
#pragma unroll
      for (int r = r_lo; r < r_lo + 8; ++r) {
        const int off = (co_lane + (r & 3) + 8 * (r >> 2)) * plane + oy * d.w + ox;
        rr_all[r] = *reinterpret_cast<const f32x2 *>(r1 + off);
      }
      if (r2) {
#pragma unroll
        for (int r = r_lo; r < r_lo + 8; ++r) {
          const int off = (co_lane + (r & 3) + 8 * (r >> 2)) * plane + oy * d.w + ox;
          rr_all[r] += *reinterpret_cast<const f32x2 *>(r2 + off);
        }
      }
    };
    if (res_fast) load_res(0);
    const float sgn = ph ? -1.f : 1.f;  // Y[0] = (t0 + t1) + t2,  Y[1] = -(t2 + t3) + t1
    // y_scale rides on instructions the residual / gate variants already issue (add -> fma, gate select picks between two
    // constants): bit-identical results for y_scale = 1, no cost; the other variants do not take a scale (winograd_eligible)
    const float ys = a.ys, ys_gs = a.ys_gs;
#ifndef WINO_EXP_NOXCHG
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
#endif
    float mine[16][2];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
#ifdef WINO_EXP_NOXCHG
      mine[r][0] = sgn * sum[r][0];
      mine[r][1] = sgn * sum[r][1];
#else
      mine[r][0] = sgn * sum[r][0] + xrecv[(2 * r) * 64 + lane];
      mine[r][1] = sgn * sum[r][1] + xrecv[(2 * r + 1) * 64 + lane];
#endif
    }
    __builtin_amdgcn_sched_barrier(0);
    load_begin(CK);  // chunk 1 of the next item (geometry already switched)
#pragma unroll
    for (int c = 0; c < 4; ++c) load_col(c);
    auto emit = [&](auto HAS_RES, auto SHUFFLE, auto SIGMOID, auto INTERIOR) {
      constexpr int RES = decltype(HAS_RES)::value;  // 0: none, 1: add residual(s), 2: gate
      constexpr bool SHF = decltype(SHUFFLE)::value, SIG = decltype(SIGMOID)::value, INT = decltype(INTERIOR)::value;
      if (RES && INT) load_res(8);
#pragma unroll
      for (int rb = 0; rb < 16; rb += 4) {  // batches of 4 channels: bias / residual loads issued ahead of their use
        f32x2 rr[4];
#pragma unroll
        for (int ri = 0; ri < 4; ++ri) {
          const int r = rb + ri, co = co_lane + (r & 3) + 8 * (r >> 2);
          const int cc = INT ? co : (co < d.co ? co : d.co - 1);
          if (RES) {
            if (INT) {
              rr[ri] = rr_all[r];  // prefetched above (res_fast)
            } else {
#pragma unroll
              for (int xx = 0; xx < 2; ++xx) {
                const bool ok = co < d.co && oy < d.h && ox + xx < d.w;
                const int off = ok ? cc * plane + oy * d.w + ox + xx : 0;
                float v = r1[off];
                if (r2) v += r2[off];
                rr[ri][xx] = ok ? v : 0.f;
              }
            }
          }
        }
#pragma unroll
        for (int ri = 0; ri < 4; ++ri) {
          const int r = rb + ri, co = co_lane + (r & 3) + 8 * (r >> 2);
          const float sl = (co >= d.act_from) ? slope : 1.f;  // per-lane select, no branch
          float o[2];
#pragma unroll
          for (int xx = 0; xx < 2; ++xx) {
            float v = mine[r][xx] + bias_r[r];
            if (SIG) v = (co >= d.act_from) ? __builtin_amdgcn_rcpf(1.f + __expf(-v)) : v;
            else v = fmaxf(v, sl * v);
            if (RES == 1) v = __builtin_fmaf(v, ys, rr[ri][xx]);
            if (RES == 2) v *= rr[ri][xx] > 0.f ? ys : ys_gs;
            o[xx] = v;
          }
          if (SHF) {
#pragma unroll
            for (int xx = 0; xx < 2; ++xx)
              if (INT || (co < d.co && oy < d.h && ox + xx < d.w))
                y[(co >> 2) * plane * 4 + (2 * oy + ((co >> 1) & 1)) * (2 * d.w) + 2 * (ox + xx) + (co & 1)] = o[xx];
          } else if (INT) {  // 16 lanes x 8 B = one 128-B line per row
#ifdef WINO_EXP_NOSTORE
            if (o[0] == 12345.f)  /* ablation only */
#endif
            *reinterpret_cast<f32x2 *>(y + co * plane + oy * d.w + ox) = f32x2{o[0], o[1]};
          } else {
#pragma unroll
            for (int xx = 0; xx < 2; ++xx)
              if (co < d.co && oy < d.h && ox + xx < d.w) y[co * plane + oy * d.w + ox + xx] = o[xx];
          }
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    };
    using T = std::true_type;
    using F = std::false_type;
    if (d.act == EDVR_ACT_SIGMOID) {
      emit(F{}, F{}, T{}, F{});
    } else if (d.out_mode == EDVR_OUT_PIXEL_SHUFFLE2) {
      if (interior) emit(F{}, T{}, F{}, T{}); else emit(F{}, T{}, F{}, F{});
    } else if (gated) {
      using G2 = std::integral_constant<int, 2>;
      if (interior) emit(G2{}, F{}, F{}, T{}); else emit(G2{}, F{}, F{}, F{});
    } else if (r1) {
      if (interior) emit(T{}, F{}, F{}, T{}); else emit(T{}, F{}, F{}, F{});
    } else {
      if (interior) emit(F{}, F{}, F{}, T{}); else emit(F{}, F{}, F{}, F{});
    }
#endif  // WINO_EXP_NOEPI
#pragma unroll
    for (int xi = 0; xi < 8; ++xi)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[xi][r] = 0.f;
  }  // persistent item loop
}

bool winograd4_enabled() {
  static const bool on = []() {
    const char *e = getenv("EDVR_WINOGRAD_4WAVE");  // "1": this kernel instead of the 8-wave one of winograd.hip (A/B).  Default off:
    return e && e[0] == '1';                        // measured at parity (-3 % on 20-image launches, +3 % on 4-image ones), see the header
  }();
  return on;
}

// Same eligibility as winograd.hip plus: an even c1 when a second input is concatenated (a staging wave covers two consecutive
// channels with one buffer resource) and two channel planes below 2 GB.
bool winograd4_supported(const edvr_conv2d_desc &d) {
  return winograd4_enabled() && (d.c2 == 0 || (d.c1 & 1) == 0) && (int64_t)d.h * d.w * 8 < ((int64_t)1 << 31);
}

int winograd4_launch(const edvr_conv2d_desc &d, const float *U, int cop, hipStream_t stream) {
  Wino4Args a;
  a.d = d;
  a.U = U;
  a.ci_real = d.c1 + d.c2;
  a.ci = (a.ci_real + 15) / 16 * 16;
  a.cop = cop;
  a.ys = d.y_scale == 0.f ? 1.f : d.y_scale;
  a.ys_gs = a.ys * d.gate_slope;
  a.tiles_x = cdiv(d.w, 32);
  a.tiles_y = cdiv(d.h, 4);
  a.items = a.tiles_x * a.tiles_y * cdiv(d.co, 64) * d.n;
  static const int n_cu = []() {
    int dev = 0, n = 256;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) n = prop.multiProcessorCount;
    return n;
  }();
  auto aligned8 = [](const float *p, int64_t img_stride) { return (reinterpret_cast<uintptr_t>(p) & 7) == 0 && (img_stride & 1) == 0; };
  const bool pair = (d.w & 1) == 0 && ((int64_t)d.h * d.w & 1) == 0 && aligned8(d.x1, d.x1_img_stride) && (!d.x2 || aligned8(d.x2, d.x2_img_stride));
  const dim3 grid(std::min(a.items, 2 * n_cu));
  if (d.gate) {
    if (pair) hipLaunchKernelGGL((conv3x3_winograd4_kernel<true, true>), grid, dim3(256), 0, stream, a);
    else hipLaunchKernelGGL((conv3x3_winograd4_kernel<false, true>), grid, dim3(256), 0, stream, a);
  } else if (pair) hipLaunchKernelGGL((conv3x3_winograd4_kernel<true, false>), grid, dim3(256), 0, stream, a);
  else hipLaunchKernelGGL((conv3x3_winograd4_kernel<false, false>), grid, dim3(256), 0, stream, a);
  return check_launch("conv3x3_winograd4_kernel");
}

}  // namespace edvr
