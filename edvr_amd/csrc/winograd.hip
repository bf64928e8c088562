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
// Work split: a persistent 512-thread workgroup (8 waves, TWO per SIMD, 256 registers each) owns 64 output channels
// x 64 tiles (4 x 16 tiles = 8 x 32 output pixels) per item.  Wave w = (ph, quadrant): the quadrant is one 32-channel x
// 32-tile MFMA tile, ph selects transform rows {2ph, 2ph+1} = 8 of the 16 positions: 8 accumulator tiles = 128 AGPRs.
// Two waves per SIMD is the point of the split: with one 256-accumulator wave per SIMD (round-1 first version, 44 % of
// the MFMA peak) every staging stall of the single in-order wave idles the matrix core; here the sibling wave issues.
// Per chunk of 8 input channels: U slab (8 x 16 x 64) and V slab (8 x 16 x 64) in LDS, double-buffered (128 KB);
// operands are single ds_read_b32 at base + immediate.  Staging is a rotating-register pipeline: the 16 patch elements
// and 4 U vectors a thread holds are consumed (transformed / written to LDS) and immediately reloaded with the data of
// the chunk after next, so every load has one full iteration of latency budget with a single register set.  The
// pipeline runs ACROSS items: the last two iterations of an item already stage chunks 0 and 1 of the next one.
// The output transform is split the same way: each wave reduces its two rows, the siblings swap one half through LDS
// (the idle buffer pair) and each finishes one output row of every 2x2 tile.
#include <cstdlib>
#include <type_traits>

#include "common.h"
#include "pack.h"

namespace edvr {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

struct WinoArgs {
  edvr_conv2d_desc d;
  const float *U;  // [ci_pad][16][cop]
  int ci, ci_real, cop, tiles_x, tiles_y, items;  // ci: rounded up to 16 (U has all-zero rows there), ci_real = c1 + c2
  float ys, ys_gs;  // y_scale (0 -> 1) and y_scale * gate_slope, resolved on the host: kernel arguments live in SGPRs, a select or
                    // a product computed in the kernel would occupy vector registers this kernel does not have
};

// Accumulators are plain vector values: with 128 of them per wave (two waves per SIMD share the unified 512-register file,
// 256 each) hipcc keeps them in place across the loop.  (The 4-wave version of this kernel had 256 accumulator registers
// per wave and needed inline asm with "+a" constraints to stop the compiler from copying them between the accumulator and
// vector files every iteration.)
__device__ __forceinline__ void mfma_acc(f32x16 &acc, float a, float b) {
  acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
}

// PAIR: even image width and 8-byte aligned planes - columns 1 and 2 of every patch row are then an aligned pair, valid or
// padded together, and come from ONE 64-bit load (12 instead of 16 gathers per chunk and thread).
template <bool PAIR, bool GATE = false>
__global__ __launch_bounds__(512, 1) void conv3x3_winograd_kernel(const WinoArgs a) {
  constexpr int CK = 8, TY = 4, TX = 16;  // 64 tiles = 8 x 32 output pixels
  constexpr int SLAB = CK * 16 * 64;      // floats per LDS slab (U or V)
  constexpr int SMEM = 4 * SLAB;          // two (U, V) slab pairs = 128 KB
  __shared__ __attribute__((aligned(16))) float smem[SMEM];

  const edvr_conv2d_desc &d = a.d;
  const int tid = threadIdx.x, lane = tid & 63, half = lane >> 5, j = lane & 31;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int quad = wave & 3, wm = quad >> 1, wn = quad & 1, ph = wave >> 2;
  const int hw = d.h * d.w;
  const int co_blocks = (d.co + 63) / 64;
  const int u_row0 = tid >> 4, u_c4 = tid & 15;  // U staging: slab rows u_row0 + 32 g, 4 floats at column 4 u_c4
  const int p_ty = lane >> 4, p_tx = lane & 15;  // V staging: channel `wave` of the chunk, tile `lane`

  // ---- geometry of the item being LOADED (the pipeline loads one item ahead of the MFMAs at item boundaries)
  int co_blk = 0, img = 0, ty0 = 0, tx0 = 0;
  const float *x1 = d.x1, *x2 = d.x1;
  // Buffer addressing (uniform 128-bit resource in SGPRs + one 32-bit VGPR byte offset per lane) instead of 64-bit flat
  // pointers: no per-load address arithmetic, 16 offset registers instead of 32, and the hardware range check supplies
  // the zero padding: a patch element outside the image gets offset 0x80000000 >= num_records and loads as 0.
  constexpr int RSRC_FLAGS = 0x00020000;  // raw buffer, 32-bit data format (gfx9 family)
  const int plane_bytes = hw * 4;
  auto uniform_rsrc = [&](const float *p, int bytes) {
    // readfirstlane: the pointers ARE wave-uniform, but unless the compiler can prove it (and keeps the resource in SGPRs)
    // every buffer load is wrapped in a waterfall loop - a branch per load inside the MFMA loop
    const uint64_t pv = reinterpret_cast<uint64_t>(p);
    const uint64_t pu = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(pv >> 32)) << 32) |
                        (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)pv);
    return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void *>(pu), (short)0, bytes, RSRC_FLAGS);
  };
  const int u_bytes = a.ci * 16 * a.cop * 4;
  const int u_voff = (u_row0 * a.cop + u_c4 * 4) * 4;
  int p_off[16];
  auto setup = [&](int item) {
    // (integer division runs on the VALU: readfirstlane moves the wave-uniform results back to SGPRs, otherwise every use
    //  as a scalar operand - buffer soffset, resource base - costs a waterfall loop)
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
        p_off[r * 4 + c] = ok ? ((gy0 + r) * d.w + gx0 + c) * 4 : (int)0x80000000;
      }
  };

  f32x16 acc[8];  // 128 accumulator registers: tiles 0-3 = transform row row_lo, 4-7 = row_hi (below)
  float pr[16];   // raw patch of (channel `wave`, tile `lane`) of the chunk being staged
  f32x4 ur[4];    // its U vectors
  float tt[16];   // B^T d

  // ---- loads (chunk channel base c0): the channel plane of this wave is a wave-uniform buffer resource
  __amdgpu_buffer_rsrc_t ld_rsrc = uniform_rsrc(a.U, u_bytes), u_rsrc = ld_rsrc;
  int ld_u_soff = 0;
  auto load_begin = [&](int c0) {
    const int c = c0 + wave;
    const float *pl = (c < d.c1) ? (x1 + (int64_t)c * hw) : (x2 + (int64_t)(c - d.c1) * hw);
    ld_rsrc = uniform_rsrc(pl, c < a.ci_real ? plane_bytes : 0);  // channels of the 16-padding: empty buffer, every load returns 0
    u_rsrc = uniform_rsrc(a.U, u_bytes);
    ld_u_soff = (c0 * 16 * a.cop + co_blk) * 4;
  };
  auto load_col = [&](int c) {
    if (PAIR && c == 1) return;  // loaded together with column 2, once both have been consumed
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
  auto load_u = [&](int g) {
    ur[g] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(u_rsrc, u_voff, ld_u_soff + g * 32 * a.cop * 4, 0));
  };
  // ---- transform + commit
  auto transform_col = [&](int c) {
    const float d0 = pr[0 * 4 + c], d1 = pr[1 * 4 + c], d2 = pr[2 * 4 + c], d3 = pr[3 * 4 + c];
    tt[0 * 4 + c] = d0 - d2;
    tt[1 * 4 + c] = d1 + d2;
    tt[2 * 4 + c] = d2 - d1;
    tt[3 * 4 + c] = d1 - d3;
  };
  auto commit_v_row = [&](float *Vs, int r) {  // positions xi = 4r .. 4r+3 of (B^T d) B
    const float *t = tt + r * 4;
    float *dst = Vs + (wave * 16 + r * 4) * 64 + lane;
    dst[0 * 64] = t[0] - t[2];
    dst[1 * 64] = t[1] + t[2];
    dst[2 * 64] = t[2] - t[1];
    dst[3 * 64] = t[1] - t[3];
  };
  auto commit_u = [&](float *Us, int g) { *reinterpret_cast<f32x4 *>(Us + (tid + g * 512) * 4) = ur[g]; };

  // Accumulator tiles 0-3 hold the transform row this wave SENDS to its sibling in the epilogue (row 1 for ph = 0, row 2 for
  // ph = 1), tiles 4-7 the other one (row 0 / row 3): the epilogue is then the same straight-line code for both kinds of
  // wave and only ever holds one running sum.  The row is just an LDS operand offset.
  const int row_lo = ph ? 2 : 1, row_hi = ph ? 3 : 0;
  const int abase[2] = {half * 16 * 64 + row_lo * 4 * 64 + wm * 32 + j, half * 16 * 64 + row_hi * 4 * 64 + wm * 32 + j};  // A (U): co tile
  const int bbase[2] = {half * 16 * 64 + row_lo * 4 * 64 + wn * 32 + j, half * 16 * 64 + row_hi * 4 * 64 + wn * 32 + j};  // B (V): tile group
  // One chunk: 8 groups (channel pair cp = g >> 1, accumulator tiles 4 (g & 1) .. +3) of 4 MFMAs on LDS pair P; operands of
  // group g+1 are fetched before the MFMAs of group g.  In their shadow the chunk held in registers (k+1) is transformed
  // and written to pair 1-P, and every register is reloaded with the chunk after it (channel base c_load) right after
  // its last use.  Branch-free: any control flow here makes hipcc shuffle the accumulators through VGPRs.
  auto iteration = [&](auto PAR, int c_load, auto LOAD) {
    constexpr int P = decltype(PAR)::value;
    constexpr bool LD = decltype(LOAD)::value;  // false: the reloads are issued by the caller (end of an item, see below)
    const float *Us = smem + P * 2 * SLAB, *Vs = Us + SLAB;
    float *Ud = smem + (1 - P) * 2 * SLAB, *Vd = Ud + SLAB;
    load_begin(c_load);
    float av[2][4], bv[2][4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      av[0][i] = Us[abase[0] + i * 64];
      bv[0][i] = Vs[bbase[0] + i * 64];
    }
#pragma unroll
    for (int g = 0; g < 8; ++g) {
      const int cur = g & 1, nxt = cur ^ 1;
      if (g + 1 < 8) {
        const int cpn = (g + 1) >> 1, hn = (g + 1) & 1;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          av[nxt][i] = Us[abase[hn] + (2 * cpn * 16 + i) * 64];
          bv[nxt][i] = Vs[bbase[hn] + (2 * cpn * 16 + i) * 64];
        }
      }
      const int x0 = (g & 1) * 4;
#pragma unroll
      for (int i = 0; i < 4; ++i) mfma_acc(acc[x0 + i], av[cur][i], bv[cur][i]);
      if (g < 4) {
        transform_col(g);
        if (LD) load_col(g);
      } else {
        commit_v_row(Vd, g - 4);
        commit_u(Ud, g - 4);
        if (LD) load_u(g - 4);
      }
      // pin the slice schedule: left free, the scheduler hoists all column transforms to the top of the iteration, where they
      // wait for (nearly) every load issued in the previous one (vmcnt 4 instead of 16) - measured 1.51 -> 1.85 ms
      __builtin_amdgcn_sched_barrier(0);
    }
    // LDS-only barrier: __syncthreads() would also wait vmcnt(0), i.e. for the loads issued a few cycles ago (full memory
    // latency exposed every chunk).  Those loads target registers and need no cross-wave ordering.
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
  };
  using S0 = std::integral_constant<int, 0>;
  using S1 = std::integral_constant<int, 1>;

  // Persistent workgroups, XCD-aware walk: workgroup b runs on XCD b % 8, each XCD with its own 4 MB L2.  Handing every XCD
  // one CONTIGUOUS range of items (channel block fastest, then tiles along x) keeps the workgroups that share input lines
  // - the co-blocks of a tile, neighbouring tiles whose 128-byte lines and halo rows overlap - on one L2 at the same time; a
  // plain `item = b + k*grid` walk spreads them over all 8 L2s (measured 7.2x the output bytes fetched; 2.3x with this).
  const int n_xcd = gridDim.x < 8 ? 1 : 8;
  const int xcd = n_xcd == 1 ? 0 : (int)blockIdx.x % 8, xcd_rank = n_xcd == 1 ? (int)blockIdx.x : (int)blockIdx.x / 8;
  const int xcd_wgs = n_xcd == 1 ? (int)gridDim.x : ((int)gridDim.x - xcd + 7) / 8;
  const int chunk = (a.items + n_xcd - 1) / n_xcd;
  const int item_end = min(a.items, (xcd + 1) * chunk);
  const int item_first = xcd * chunk + xcd_rank;
  if (item_first >= item_end) return;

  // ---- prologue of the first item: chunk 0 -> registers -> LDS pair 0, chunk 1 -> registers
  setup(item_first);
  load_begin(0);
#pragma unroll
  for (int c = 0; c < 4; ++c) load_col(c);
#pragma unroll
  for (int g = 0; g < 4; ++g) load_u(g);
#pragma unroll
  for (int c = 0; c < 4; ++c) transform_col(c);
#pragma unroll
  for (int r = 0; r < 4; ++r) commit_v_row(smem + SLAB, r);
#pragma unroll
  for (int g = 0; g < 4; ++g) commit_u(smem, g);
  load_begin(CK);
#pragma unroll
  for (int c = 0; c < 4; ++c) load_col(c);
#pragma unroll
  for (int g = 0; g < 4; ++g) load_u(g);
#pragma unroll
  for (int xi = 0; xi < 8; ++xi)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[xi][r] = 0.f;
  __syncthreads();

  for (int item = item_first; item < item_end; item += xcd_wgs) {
    // a.ci is a multiple of 2*CK (rounded up by the host): chunk k (parity k & 1) loads chunk k+2
#pragma unroll 1
    for (int c0 = 2 * CK; c0 < a.ci; c0 += 2 * CK) {
      iteration(S0{}, c0, std::true_type{});
      iteration(S1{}, c0 + CK, std::true_type{});
    }
    // last two chunks of this item: they stage chunks 0 and 1 of the NEXT item (or re-stage this one after the last)
    const int e_img = img, e_ty0 = ty0, e_tx0 = tx0, e_co_blk = co_blk;
    {
      const int next = item + xcd_wgs;
      setup(next < item_end ? next : item);
    }
    iteration(S0{}, 0, std::true_type{});  // transforms the last chunk of this item, loads with the new geometry
    // The last chunk commits chunk 0 of the next item but leaves the staging registers EMPTY: the output transform below
    // needs them (all 128 accumulators are still live there); chunk 1 is loaded after the exchange, under the stores.
    iteration(S1{}, CK, std::false_type{});

    // ---- output transform Y = A^T M A.  Lane (half, j) holds tile wn*32 + j and 16 output channels
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
    const int tile = wn * 32 + j, tyy = tile >> 4, txx = tile & 15;
    const int oy = e_ty0 + 2 * tyy + ph, ox = e_tx0 + 2 * txx;
    const int co_lane = e_co_blk + wm * 32 + 4 * half;
    // Every uniform condition is resolved ONCE (compile-time variants below): evaluated per output element they become
    // ~1400 scalar branches per workgroup and made this epilogue cost as much as six chunks of the main loop.
    const float slope = d.act == EDVR_ACT_LRELU ? 0.1f : (d.act == EDVR_ACT_RELU ? 0.f : 1.f);  // none/relu/lrelu = max(v, slope*v)
    const bool interior = e_ty0 + 2 * TY <= d.h && e_tx0 + 2 * TX <= d.w && e_co_blk + 64 <= d.co && (d.w & 1) == 0;
    float *xsend = smem + 2 * SLAB + wave * 2048;        // exchange through the idle pair 1: [wave][r][lane][2] = 64 KB
    const float *xrecv = smem + 2 * SLAB + (wave ^ 4) * 2048;
    // ---- common to all epilogue variants (kept OUT of the specialised lambdas: hoisted above their dispatch by the
    //      compiler, the sums were spilled to scratch across the multi-way branch, ~70 scratch accesses per item)
    // Row pass (M A)[row][.] = (m0 + m1 + m2, m1 - m2 - m3) of the row to send (tiles 0-3), one accumulator tile at a
    // time: hipcc moves a tile out of the accumulator file as a whole 16-register tuple, so walking channel by channel
    // (8 tiles live at once) needs 128 VGPRs and spilled ~150 registers to scratch per item.
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
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    float mine[16][2];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      mine[r][0] = sgn * sum[r][0] + xrecv[(2 * r) * 64 + lane];
      mine[r][1] = sgn * sum[r][1] + xrecv[(2 * r + 1) * 64 + lane];
    }
    __builtin_amdgcn_sched_barrier(0);
    // second barrier: pair 1 is overwritten by the first iteration of the next item
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    load_begin(CK);  // chunk 1 of the next item (geometry already switched)
#pragma unroll
    for (int c = 0; c < 4; ++c) load_col(c);
#pragma unroll
    for (int g = 0; g < 4; ++g) load_u(g);
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
#pragma unroll
    for (int xi = 0; xi < 8; ++xi)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[xi][r] = 0.f;
  }  // persistent item loop
}

// U[ci][xi][cop] = (G g G^T)[xi] of the 3x3 kernel g = w[co][ci] (or the data-gradient kernel when transpose_flip)
// When `wpk` is given the same launch also writes the direct kernel's layout [cip][9][cop32] (training repacks every weight
// each step: one launch per layer and orientation instead of two).
__global__ void winograd_weight_kernel(const float *__restrict__ w, float *__restrict__ U, int co, int ci, int cop, int cip,
                                       int transpose_flip, float *__restrict__ wpk, int cop32) {
  const int64_t total = (int64_t)cip * cop;
  if (wpk) {
    const int64_t dtotal = (int64_t)cip * 9 * cop32;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < dtotal; i += (int64_t)gridDim.x * blockDim.x)
      pack_direct_elem(w, wpk, i, co, ci, 9, cop32, transpose_flip);  // pack.h
  }
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x)
    pack_u2_elem(w, U, i, co, ci, cop, transpose_flip);
}

bool winograd_eligible(const edvr_conv2d_desc &d) {
  static const bool enabled = []() {
    const char *e = getenv("EDVR_CONV_WINOGRAD");  // "0": always use the direct kernel (A/B, fallback)
    return !(e && e[0] == '0');
  }();
  // the epilogue is specialised for: plain | residual(s) | pixel-shuffle | sigmoid - other combinations use the direct kernel
  const bool has_res = d.res1 || d.res2;
  if (d.gate && (has_res || d.act == EDVR_ACT_SIGMOID || d.out_mode != EDVR_OUT_NCHW)) return false;
  if ((d.res2 && !d.res1) || (d.act == EDVR_ACT_SIGMOID && (has_res || d.out_mode != EDVR_OUT_NCHW)) ||
      (d.out_mode != EDVR_OUT_NCHW && has_res))
    return false;
  if (d.algo == EDVR_CONV_DIRECT) return false;
  if (d.y_scale != 0.f && d.y_scale != 1.f && !d.res1 && !d.gate) return false;  // the scale lives in the residual / gate epilogues
  const bool applicable = d.ks == 3 && d.stride == 1;  // any channel count: the loop runs over ci rounded up to 16
  if (d.algo == EDVR_CONV_WINOGRAD || d.algo == EDVR_CONV_WINOGRAD_F4 || d.algo == EDVR_CONV_WINOGRAD_F4S) return applicable;  // explicit request: any size the kernel can do
  return enabled && applicable && d.co >= 48 && d.c1 + d.c2 >= 32 && d.w > 16 && d.h >= 4;  // auto: only where it beats the direct kernel
}

// flops the matrix cores execute for `d`, padding included: an item is 64 output channels x 64 tiles (8 x 32 pixels) x 16
// positions, its k loop runs over the input channels rounded up to 16
double winograd_executed_flops(const edvr_conv2d_desc &d) {
  const double items = (double)cdiv(d.w, 32) * cdiv(d.h, 8) * cdiv(d.co, 64) * d.n;
  return items * ((d.c1 + d.c2 + 15) / 16 * 16) * (64.0 * 64.0 * 16.0 * 2.0);
}

int winograd_launch(const edvr_conv2d_desc &d, const float *U, int cop, hipStream_t stream) {
  WinoArgs a;
  a.d = d;
  a.U = U;
  a.ci_real = d.c1 + d.c2;
  a.ci = (a.ci_real + 15) / 16 * 16;
  a.cop = cop;
  a.ys = d.y_scale == 0.f ? 1.f : d.y_scale;
  a.ys_gs = a.ys * d.gate_slope;
  a.tiles_x = cdiv(d.w, 32);
  a.tiles_y = cdiv(d.h, 8);
  a.items = a.tiles_x * a.tiles_y * cdiv(d.co, 64) * d.n;
  static const int n_cu = []() {
    int dev = 0, n = 256;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) n = prop.multiProcessorCount;
    return n;
  }();
  auto aligned8 = [](const float *p, int64_t img_stride) { return (reinterpret_cast<uintptr_t>(p) & 7) == 0 && (img_stride & 1) == 0; };
  const bool pair = (d.w & 1) == 0 && aligned8(d.x1, d.x1_img_stride) && (!d.x2 || aligned8(d.x2, d.x2_img_stride));
  const dim3 grid(std::min(a.items, n_cu));
  if (d.gate) {
    if (pair) hipLaunchKernelGGL((conv3x3_winograd_kernel<true, true>), grid, dim3(512), 0, stream, a);
    else hipLaunchKernelGGL((conv3x3_winograd_kernel<false, true>), grid, dim3(512), 0, stream, a);
  } else if (pair) hipLaunchKernelGGL((conv3x3_winograd_kernel<true, false>), grid, dim3(512), 0, stream, a);
  else hipLaunchKernelGGL((conv3x3_winograd_kernel<false, false>), grid, dim3(512), 0, stream, a);
  return check_launch("conv3x3_winograd_kernel");
}

int winograd_pack(const float *w, float *U, int co, int ci, int cop, int cip, int transpose_flip, float *wpk_direct, int cop32,
                  hipStream_t stream) {
  const int64_t total = std::max((int64_t)cip * cop, wpk_direct ? (int64_t)cip * 9 * cop32 : (int64_t)0);
  hipLaunchKernelGGL(winograd_weight_kernel, dim3((unsigned)std::min<int64_t>(cdiv64(total, 256), 4096)), dim3(256), 0, stream, w, U, co, ci,
                     cop, cip, transpose_flip, wpk_direct, cop32);
  return check_launch("winograd_weight_kernel");
}

}  // namespace edvr
