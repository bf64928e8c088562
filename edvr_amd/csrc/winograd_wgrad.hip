// winograd_wgrad.hip - weight gradient of the 3x3 / stride-1 convolution in the Winograd F(2x2, 3x3) domain (gfx950).
//
// The direct weight-gradient kernel (wgrad.hip) issues 36 multiplies per 2x2 output tile and channel pair and runs at
// ~57 % of the fp32 MFMA peak; it was the largest kernel of the training step (35 %).  With Y = A^T [U (.) V] A the
// gradient of the transformed weights is a plain GEMM over tiles, 16 multiplies per tile and channel pair (2.25x fewer):
//
//   V  = B^T d B            input transform of each 4x4 patch         (as in the forward, winograd.hip)
//   Z  = A dY A^T           transform of the 2x2 output-gradient tile (4x4)
//   dU[xi][co, ci] = sum over tiles of Z[xi][co, tile] * V[xi][ci, tile]      16 GEMMs on v_mfma_f32_32x32x2_f32, k = tile
//   dW = G^T dU G           3x3 from 4x4, linear: applied per workgroup before the split-K partial is written
//
// Same machine mapping as the forward: 512-thread workgroups, two waves per SIMD, wave = (ph, 32 x 32 quadrant of the
// 64 co x 64 ci block), ph = transform rows {2ph, 2ph+1} = 8 accumulator tiles; per chunk of 8 tiles (one k-chunk) the Z
// and V slabs are staged in LDS (double-buffered, 132 KB) by a rotating-register pipeline; all global reads are buffer
// loads whose hardware range check supplies the zero padding.  The tile axis is split over workgroups (deterministic
// split-K); the two ph waves of a quadrant combine their rows through LDS, one G-transformed partial per split is written and
// wgrad_reduce_kernel sums `splits` of them.
#include <cstdlib>
#include <type_traits>

#include "common.h"

namespace edvr {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

struct WinoWgradArgs {
  const float *x1, *x2, *dz;
  float *ws;  // [splits][co][ci][9], then (want_db) [splits][co] bias-gradient partials
  int want_db;
  int c1, c2, n, h, w, co;
  int64_t x1_img_stride, x2_img_stride, dz_img_stride;
  int x2_div, x2_mul, x2_add;
  int cpr, th, total_chunks, splits, ci_blocks, co_blocks;  // chunks (8 tiles) per tile row, tile rows, n * th * cpr
};

__global__ __launch_bounds__(512, 1) void conv3x3_winograd_wgrad_kernel(const WinoWgradArgs a) {
  constexpr int CS = 20;       // floats per (tile, channel) row of a slab: its 16 positions + 4 - with a row stride of 20 the eight lanes
                               // of a 16-byte LDS access cover all 32 banks (0, 20, 8, 28, 16, 4, 24, 12), reads and writes alike
  constexpr int TS = 64 * CS;  // floats per tile: [tile][channel][xi]: the four positions a group of MFMAs consumes are ONE ds_read_b128
                               // and a transformed row is ONE ds_write_b128 (first layout: [tile][xi][channel], 4-byte accesses - 96
                               // LDS instructions per 32 MFMAs and wave, now 24)
  constexpr int SLAB = 8 * TS, SMEM = 4 * SLAB;  // (Z, V) x 2 stages = 160 KB: all of the CU's LDS
  constexpr int RSRC_FLAGS = 0x00020000;
  constexpr int OOB = (int)0x80000000;  // >= num_records: the load returns 0 without touching memory
  __shared__ __attribute__((aligned(16))) float smem[SMEM];

  const int tid = threadIdx.x, lane = tid & 63, half = lane >> 5, j = lane & 31;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int quad = wave & 3, wm = quad >> 1, wn = quad & 1, ph = wave >> 2;
  const int hw = a.h * a.w, ci_total = a.c1 + a.c2;
  // (split, co block, ci block) of this workgroup, XCD-aware: the blocks of one split read the same tiles (common.h)
  const int blocks = a.ci_blocks * a.co_blocks;
  const int lg = xcd_remap((int)blockIdx.x, (int)gridDim.x);
  const int split = __builtin_amdgcn_readfirstlane(lg / blocks);
  const int blk = __builtin_amdgcn_readfirstlane(lg % blocks);
  const int co_blk = __builtin_amdgcn_readfirstlane((blk / a.ci_blocks) * 64);
  const int ci_blk = __builtin_amdgcn_readfirstlane((blk % a.ci_blocks) * 64);
  // chunk range of this split, in pairs (the loop body handles two chunks); chunks >= total_chunks are fully masked
  const int pairs_total = (a.total_chunks + 1) / 2;
  const int q0 = __builtin_amdgcn_readfirstlane(2 * (int)((int64_t)pairs_total * split / a.splits));
  const int q1 = __builtin_amdgcn_readfirstlane(2 * (int)((int64_t)pairs_total * (split + 1) / a.splits));
  if (q0 >= q1) return;  // (never with splits <= pairs_total; the partial of this split would stay unwritten)

  // ---- staging role of this thread: channel chl = 8 wave + (lane & 7) of the block, tile t = lane >> 3 of the chunk (channel-fast:
  //      the eight lanes of a 16-byte LDS write then differ in the channel - conflict-free, see CS)
  const int t = lane >> 3, chl = wave * 8 + (lane & 7);
  const bool use_x2 = a.c2 > 0 && ci_blk >= a.c1;  // blocks never straddle x1 / x2 (c1 % 64 == 0, checked by the host)
  const int ci_s = ci_blk + chl, co_s = co_blk + chl;
  const bool valid_ci = ci_s < (use_x2 || a.c2 == 0 ? ci_total : a.c1), valid_co = co_s < a.co;
  const int ci_in = use_x2 ? ci_s - a.c1 : ci_s;
  int rel[4];  // byte offset of patch row r (its column 0) of tile t from the chunk's window origin (top-left halo pixel)
#pragma unroll
  for (int r = 0; r < 4; ++r) rel[r] = (ci_in * hw + r * a.w + 2 * t) * 4;
  const int dz_rel = (co_s * hw + 2 * t) * 4;

  // ---- geometry of the chunk being LOADED (wave-uniform; advanced branch-free once per iteration)
  int q = q0;
  int xc, ty, img;
  {
    const int unit = q0 / a.cpr;
    xc = __builtin_amdgcn_readfirstlane(q0 % a.cpr);
    ty = __builtin_amdgcn_readfirstlane(unit % a.th);
    img = __builtin_amdgcn_readfirstlane(unit / a.th);
  }
  auto uniform_rsrc = [&](const float *p) {
    const uint64_t pv = reinterpret_cast<uint64_t>(p);
    const uint64_t pu = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(pv >> 32)) << 32) |
                        (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)pv);
    return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void *>(pu), (short)0, 0x7fffffff, RSRC_FLAGS);
  };
  // One 16-byte load per patch row (4 vector-memory instructions per chunk, each touching 8-16 cache lines; the first version of
  // this kernel issued b32 + b64 + b32 per row).  A row of the first tile of an image row starts one element LEFT of the image
  // row - for the first row of the tensor that is outside the allocation - so the loads go through a resource that covers exactly
  // this block's image (base = its first element, num_records = its bytes): a dword past its end is range-checked to zero
  // without touching memory (multi-dword buffer loads are checked per component).  The resources change with the IMAGE only (a
  // scalar branch once per 64 chunks on the 64 x 64 training layers); the chunk's position inside the image is a byte offset
  // added to the lane offsets (rebuilding both resources per chunk was most of the 11 % the chunk geometry cost).
  __amdgpu_buffer_rsrc_t xrow_rsrc = uniform_rsrc(a.x1), z_rsrc = xrow_rsrc;
  bool rowok[4], colok[4], tile_ok;
  int win_off = 0;  // byte offset of the chunk's window origin (halo pixel (2 ty - 1, 16 xc - 1)) from the image's first element: may be negative
  int z_off = 0;    // byte offset of the chunk's first dY pixel (2 ty, 16 xc) from the image's first element
  int rsrc_img = -1;
  const int x_img_bytes = (use_x2 ? a.c2 : a.c1) * hw * 4, z_img_bytes = a.co * hw * 4;
  auto exact_rsrc = [&](const float *p, int bytes) {
    const uint64_t pv = reinterpret_cast<uint64_t>(p);
    const uint64_t pu = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(pv >> 32)) << 32) |
                        (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)pv);
    return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void *>(pu), (short)0, bytes, RSRC_FLAGS);
  };
  auto geometry = [&]() {  // resources, offsets and validity masks of chunk q = (img, ty, xc)
    const bool in_range = q < a.total_chunks;
    if (img != rsrc_img) {  // wave-uniform
      rsrc_img = img;
      const float *xi;
      if (use_x2) {
        const int i2 = a.x2_div > 0 ? (img / a.x2_div) * a.x2_mul + a.x2_add : img;
        xi = a.x2 + (int64_t)i2 * a.x2_img_stride;
      } else {
        xi = a.x1 + (int64_t)img * a.x1_img_stride;
      }
      xrow_rsrc = exact_rsrc(xi, x_img_bytes);
      z_rsrc = exact_rsrc(a.dz + (int64_t)img * a.dz_img_stride, z_img_bytes);
    }
    win_off = ((2 * ty - 1) * a.w + 16 * xc - 1) * 4;
    z_off = (2 * ty * a.w + 16 * xc) * 4;
    const int gx = 16 * xc + 2 * t - 1;
#pragma unroll
    for (int r = 0; r < 4; ++r) rowok[r] = in_range && (unsigned)(2 * ty - 1 + r) < (unsigned)a.h;
#pragma unroll
    for (int c = 0; c < 4; ++c) colok[c] = valid_ci && (unsigned)(gx + c) < (unsigned)a.w;
    tile_ok = in_range && valid_co && (8 * xc + t) * 2 < a.w;
  };
  auto advance = [&]() {
    ++q;
    ++xc;
    const bool wrap_x = xc == a.cpr;
    xc = wrap_x ? 0 : xc;
    ty += wrap_x ? 1 : 0;
    const bool wrap_y = ty == a.th;
    ty = wrap_y ? 0 : ty;
    img += wrap_y ? 1 : 0;
  };

  f32x16 acc[8];  // [xi - 8 ph]
  float pr[16];   // raw patch (channel chl, tile t) of the chunk being staged
  float tt[16];   // B^T d
  f32x2 dy[2][2]; // [chunk parity][row] 2x2 output-gradient tile of (channel chl, tile t)
  float bsum = 0.f;  // bias gradient: sum of this thread's dY tiles (the values are in registers anyway)
  // Column c of the patch for chunk q (the geometry() state).  Columns 1 and 2 of a row are an 8-byte aligned pair (even x,
  // even w and h), both valid or both invalid: one 64-bit load, issued once both columns have been consumed (c == 2).
  // Row r of the patch for chunk q (the geometry() state).  The lanes whose column 0 lies left of the image (first tile of an image
  // row) load columns 1..4 instead and move them up one place: their offset stays >= 0 (a negative one would have to rely on how
  // the range check wraps); at the right border and at the end of the image the dwords past num_records come back as 0.
  auto load_row = [&](int r) {
    const bool shl = !colok[0];
    const int off = rel[r] + win_off + (shl ? 4 : 0);
    const f32x4 v = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xrow_rsrc, (rowok[r] && colok[1]) ? off : OOB, 0, 0));
    pr[r * 4 + 0] = shl ? 0.f : v[0];
    pr[r * 4 + 1] = shl ? v[0] : v[1];
    pr[r * 4 + 2] = shl ? v[1] : v[2];
    pr[r * 4 + 3] = colok[3] ? (shl ? v[2] : v[3]) : 0.f;
  };
  auto load_dy = [&](auto SET) {
    constexpr int S = decltype(SET)::value;
    dy[S][0] = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(z_rsrc, tile_ok ? dz_rel + z_off : OOB, 0, 0));
    dy[S][1] = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(z_rsrc, tile_ok ? dz_rel + z_off + a.w * 4 : OOB, 0, 0));
  };
  auto transform_row = [&](int r) {  // (d B) of patch row r; commit_v_row applies B^T down the columns.  Row-wise first, so that a
                                     // row's registers are free - and re-requested - right after its own pass
    const float d0 = pr[r * 4 + 0], d1 = pr[r * 4 + 1], d2 = pr[r * 4 + 2], d3 = pr[r * 4 + 3];
    tt[r * 4 + 0] = d0 - d2;
    tt[r * 4 + 1] = d1 + d2;
    tt[r * 4 + 2] = d2 - d1;
    tt[r * 4 + 3] = d1 - d3;
  };
  auto commit_v_row = [&](float *Vs, int r) {  // positions xi = 4r .. 4r+3 of B^T (d B): B^T = [[1,0,-1,0],[0,1,1,0],[0,-1,1,0],[0,1,0,-1]]
    float *dst = Vs + t * TS + chl * CS + r * 4;
    const float *ra = tt + (r == 0 ? 0 : r == 1 ? 1 : r == 2 ? 2 : 1) * 4, *rb = tt + (r == 0 ? 2 : r == 1 ? 2 : r == 2 ? 1 : 3) * 4;
    {
      f32x4 v;
#pragma unroll
      for (int jx = 0; jx < 4; ++jx) v[jx] = r == 1 ? ra[jx] + rb[jx] : ra[jx] - rb[jx];
      *reinterpret_cast<f32x4 *>(dst) = v;
    }
  };
  float bvalid = 1.f;  // 0 once the chunk being committed lies beyond this split's range (its dY must not be counted)
  auto commit_z_row = [&](float *Zs, auto SET, int r) {  // row r of A dY A^T, A = [[1,0],[1,1],[1,-1],[0,-1]]
    constexpr int S = decltype(SET)::value;
    const float p = dy[S][0][0], qq = dy[S][0][1], u = dy[S][1][0], v = dy[S][1][1];
    if (r == 0) bsum += bvalid * ((p + qq) + (u + v));
    // rows of A dY: (p, qq), (p + u, qq + v), (p - u, qq - v), (-u, -v)
    const float e = r == 0 ? p : r == 1 ? p + u : r == 2 ? p - u : -u;
    const float f = r == 0 ? qq : r == 1 ? qq + v : r == 2 ? qq - v : -v;
    f32x4 zv;
    zv[0] = e;
    zv[1] = e + f;
    zv[2] = e - f;
    zv[3] = -f;
    *reinterpret_cast<f32x4 *>(Zs + t * TS + chl * CS + r * 4) = zv;
  };

  const int abase = half * TS + (wm * 32 + j) * CS + ph * 8;  // A operand (Z): tile `half` of the pair, this wave's co tile
  const int bbase = half * TS + (wn * 32 + j) * CS + ph * 8;  // B operand (V): this wave's ci tile
  // One chunk (parity P): 8 groups (tile pair s = g >> 1, positions 8 ph + 4 (g & 1) .. +3) of 4 MFMAs on LDS stage P; the
  // chunk held in registers (k+1) is transformed into stage 1-P and every register is reloaded with chunk k+2 right after
  // its last use.  Branch-free.
  auto iteration = [&](auto PAR) {
    constexpr int P = decltype(PAR)::value;
    using Other = std::integral_constant<int, 1 - P>;
    const float *Zs = smem + P * 2 * SLAB, *Vs = Zs + SLAB;
    float *Zd = smem + (1 - P) * 2 * SLAB, *Vd = Zd + SLAB;
    f32x4 av[2], bv[2];
    av[0] = *reinterpret_cast<const f32x4 *>(Zs + abase);
    bv[0] = *reinterpret_cast<const f32x4 *>(Vs + bbase);
#pragma unroll
    for (int g = 0; g < 8; ++g) {
      const int cur = g & 1, nxt = cur ^ 1;
      if (g + 1 < 8) {
        const int sn = (g + 1) >> 1, x0n = ((g + 1) & 1) * 4;
        av[nxt] = *reinterpret_cast<const f32x4 *>(Zs + abase + 2 * sn * TS + x0n);
        bv[nxt] = *reinterpret_cast<const f32x4 *>(Vs + bbase + 2 * sn * TS + x0n);
      }
      const int x0 = (g & 1) * 4;
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[x0 + i] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[cur][i], bv[cur][i], acc[x0 + i], 0, 0, 0);
      if (g == 0) load_dy(PAR);  // set P held chunk k, consumed an iteration ago
      if (g < 4) {
        transform_row(g);
        load_row(g);
      } else {
        commit_v_row(Vd, g - 4);
        commit_z_row(Zd, Other{}, g - 4);
      }
      __builtin_amdgcn_sched_barrier(0);  // pin the slice schedule (see winograd.hip)
    }
    // geometry of the NEXT iteration's loads, here rather than at the top of the iteration: the scalar chain (pointer
    // arithmetic, resource words, validity masks) then runs under the MFMAs of the last group instead of in front of an idle
    // matrix pipe right after the barrier (ablation: 8 % of the kernel)
    advance();
    geometry();
    // LDS-only barrier: the loads just issued target registers and need no cross-wave ordering
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
  };
  using S0 = std::integral_constant<int, 0>;
  using S1 = std::integral_constant<int, 1>;

  // ---- prologue: chunk q0 -> registers -> LDS stage 0, chunk q0 + 1 -> registers
  geometry();
  load_dy(S0{});
#pragma unroll
  for (int c = 0; c < 4; ++c) load_row(c);
#pragma unroll
  for (int c = 0; c < 4; ++c) transform_row(c);
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    commit_v_row(smem + SLAB, r);
    commit_z_row(smem, S0{}, r);
  }
  advance();
  geometry();
  load_dy(S1{});
#pragma unroll
  for (int c = 0; c < 4; ++c) load_row(c);
  advance();
  geometry();  // chunk q0 + 2: loaded by the first iteration (each iteration prepares the next one's geometry at its end)
#pragma unroll
  for (int xi = 0; xi < 8; ++xi)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[xi][r] = 0.f;
  __syncthreads();

#pragma unroll 1
  for (int k = q0; k < q1; k += 2) {
    iteration(S0{});                     // commits chunk k + 1 (always inside the range: q1 - q0 is even)
    bvalid = k + 2 < q1 ? 1.f : 0.f;     // the second one commits chunk k + 2
    iteration(S1{});
  }

  // ---- bias gradient partial of this split: sum over the 8 tiles a channel's lanes hold (lanes 8 t + channel), written by the
  //      workgroups of input-channel block 0 only (every ci block saw the same dY)
  if (a.want_db && ci_blk == 0) {
    float s = bsum;
    s += __shfl_xor(s, 8, 64);
    s += __shfl_xor(s, 16, 64);
    s += __shfl_xor(s, 32, 64);
    if (t == 0 && valid_co) (a.ws + (int64_t)a.splits * a.co * ci_total * 9)[(int64_t)split * a.co + co_s] = s;
  }

  // ---- epilogue: dW = G^T dU G of this split.  Row pass t[rr][jx] = (dU G)[2 ph + rr][jx] in every wave; the ph = 1
  //      wave of a quadrant hands its two rows to its ph = 0 sibling through LDS (the slabs are dead after the last
  //      barrier: 4 waves x 64 lanes x 96 floats = 96 KB, lane-contiguous so the exchange is conflict-free), and the
  //      ph = 0 wave applies G^T over all four rows and writes ONE partial per split; wgrad_reduce_kernel sums `splits`
  //      of them.  G = [[1,0,0],[.5,.5,.5],[.5,-.5,.5],[0,0,1]]
  float tr[16][2][3];
#pragma unroll
  for (int r = 0; r < 16; ++r)
#pragma unroll
    for (int rr = 0; rr < 2; ++rr) {
      const float u0 = acc[rr * 4 + 0][r], u1 = acc[rr * 4 + 1][r], u2 = acc[rr * 4 + 2][r], u3 = acc[rr * 4 + 3][r];
      tr[r][rr][0] = u0 + 0.5f * (u1 + u2);
      tr[r][rr][1] = 0.5f * (u1 - u2);
      tr[r][rr][2] = 0.5f * (u1 + u2) + u3;
    }
  float *xch = smem + quad * (96 * 64) + lane;  // [quad][r * 6 + rr * 3 + jx][lane]
  if (ph) {
#pragma unroll
    for (int r = 0; r < 16; ++r)
#pragma unroll
      for (int q = 0; q < 6; ++q) xch[(r * 6 + q) * 64] = tr[r][q / 3][q % 3];
  }
  __syncthreads();
  if (ph) return;
  float *out = a.ws + (int64_t)split * a.co * ci_total * 9;
  const int ci_o = ci_blk + wn * 32 + j;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int co_o = co_blk + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
    float t2[3], t3[3];
#pragma unroll
    for (int jx = 0; jx < 3; ++jx) {
      t2[jx] = xch[(r * 6 + jx) * 64];
      t3[jx] = xch[(r * 6 + 3 + jx) * 64];
    }
    if (co_o < a.co && ci_o < ci_total) {
      float *dst = out + ((int64_t)co_o * ci_total + ci_o) * 9;
#pragma unroll
      for (int jx = 0; jx < 3; ++jx) {
        const float m = 0.5f * (tr[r][1][jx] + t2[jx]), d = 0.5f * (tr[r][1][jx] - t2[jx]);
        dst[0 + jx] = tr[r][0][jx] + m;
        dst[3 + jx] = d;
        dst[6 + jx] = m + t3[jx];
      }
    }
  }
}

static int wino_wgrad_cus() {
  static const int n_cu = []() {
    int dev = 0, n = 256;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) n = prop.multiProcessorCount;
    return n;
  }();
  return n_cu;
}

static int g_wgrad_algo = EDVR_CONV_AUTO;
int winograd_wgrad_get_algo() { return g_wgrad_algo; }
int winograd_wgrad_set_algo(int algo) {
  const int prev = g_wgrad_algo;
  g_wgrad_algo = algo;
  return prev;
}

// Plan of the Winograd weight gradient: false if the layer is not eligible (the direct kernel handles it).
bool winograd_wgrad_plan(int n, int c1, int c2, int h, int w, int co, int ks, int stride, int *splits) {
  static const bool enabled = []() {
    const char *e = getenv("EDVR_WGRAD_WINOGRAD");  // "0": always use the direct kernel (A/B, fallback)
    return !(e && e[0] == '0');
  }();
  const int ci = c1 + c2;
  const bool forced = g_wgrad_algo == EDVR_CONV_WINOGRAD;
  if (g_wgrad_algo == EDVR_CONV_DIRECT || (!enabled && !forced)) return false;
  if (ks != 3 || stride != 1 || (h & 1) || (w & 1)) return false;
  if (!forced && (co < 48 || ci < 48)) return false;  // mostly-empty 64 x 64 blocks: the direct kernel's narrow tiles win
  if (c2 > 0 && (c1 % 64) != 0) return false;                      // a 64-channel block must not straddle the two inputs
  if ((int64_t)std::max(std::max(c1, c2), co) * h * w * 4 >= (int64_t)1 << 31) return false;  // 32-bit buffer offsets
  const int cpr = cdiv(w / 2, 8), total_chunks = n * (h / 2) * cpr, pairs = (total_chunks + 1) / 2;
  const int blocks = cdiv(ci, 64) * cdiv(co, 64);
  int s = std::max(1, wino_wgrad_cus() / blocks);
  if (s > pairs) s = pairs;
  if (!forced && pairs < 4 * s && pairs < 64) return false;  // too little work per workgroup to amortise the pipeline fill
  *splits = s;
  return true;
}

size_t winograd_wgrad_ws_bytes(int co, int ci, int splits) {
  return ((size_t)splits * co * ci * 9 + (size_t)splits * co) * sizeof(float);  // dW partials + bias-gradient partials
}

int winograd_wgrad_launch(const float *x1, const float *x2, const float *dz, float *ws, int c1, int c2, int n, int h, int w, int co,
                          int64_t x1_img_stride, int64_t x2_img_stride, int x2_div, int x2_mul, int x2_add, int64_t dz_img_stride,
                          int splits, int want_db, hipStream_t stream) {
  WinoWgradArgs a;
  a.want_db = want_db;
  a.x1 = x1; a.x2 = x2; a.dz = dz; a.ws = ws;
  a.c1 = c1; a.c2 = c2; a.n = n; a.h = h; a.w = w; a.co = co;
  a.x1_img_stride = x1_img_stride; a.x2_img_stride = x2_img_stride; a.dz_img_stride = dz_img_stride;
  a.x2_div = x2_div; a.x2_mul = x2_mul; a.x2_add = x2_add;
  a.cpr = cdiv(w / 2, 8);
  a.th = h / 2;
  a.total_chunks = n * a.th * a.cpr;
  a.splits = splits;
  a.ci_blocks = cdiv(c1 + c2, 64);
  a.co_blocks = cdiv(co, 64);
  hipLaunchKernelGGL(conv3x3_winograd_wgrad_kernel, dim3(splits * a.ci_blocks * a.co_blocks), dim3(512), 0, stream, a);
  return check_launch("conv3x3_winograd_wgrad_kernel");
}

}  // namespace edvr
