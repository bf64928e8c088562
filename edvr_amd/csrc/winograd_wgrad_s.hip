// winograd_wgrad_s.hip - the Winograd-domain weight gradient (winograd_wgrad.hip) with SPLIT fp32 operands on the f16 matrix pipe.
//
// Same algorithm, same work split, same partial format as winograd_wgrad.hip:
//   V = B^T d B, Z = A dY A^T per 2x2 tile;  dU[xi][co, ci] = sum over tiles of Z[xi][co, tile] V[xi][ci, tile];  dW = G^T dU G.
// Here both operands of the tile-axis GEMM travel as f16 (hi, lo) pairs (winograd_f4s.hip: x s = hi + lo, 22 significant bits, s a
// power of two from a bound of the tensor's magnitude) and every product is the sum of all four cross terms, accumulated in fp32 by
// v_mfma_f32_32x32x16_f16: k-slots (2 i, 2 i + 1) of an operand register hold (hi, lo) of ONE tile, so one MFMA covers the 8 tiles of a
// chunk, and the second MFMA of a position takes the A registers rotated by 16 bits ((lo, hi): the two cross terms).  Per chunk and
// wave 16 MFMAs of 32 cycles where the fp32 kernel issues 32 of 64: a quarter of the matrix-pipe time, and the staging arithmetic
// (transforms + 2 instructions per split) overlaps with it.
//   |V| <= 4 max|x|, |Z| <= 4 max|dY| (rows of B^T and A have absolute sums <= 2): s_V, s_Z from `x_amax` / `dz_amax`, any upper bounds
//   of the two tensors' magnitudes on the device; 1 / (s_V s_Z) leaves in the epilogue's G^T . G pass.
// LDS: a slab is [position 16][tile quad 2][channel 64][tile 4] dwords (32 KB; Z and V, two stages = 128 KB): an operand is ONE
// conflict-free ds_read_b128 (the 4 tiles x (hi, lo) of a lane's channel), a staged value one ds_write_b32.
#include <cstdlib>
#include <type_traits>

#include "common.h"

namespace edvr {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2s __attribute__((ext_vector_type(2)));

// x * s = hi + lo in f16 (s a power of two): v_fma_mixlo_f16 + v_fma_mixhi_f16, four values at once with the four independent first
// halves in front of the dependent second ones
__device__ __forceinline__ void split4_f16x2(const float (&x)[4], float s, unsigned (&o)[4]) {
  asm volatile(
      "v_fma_mixlo_f16 %0, %4, %8, 0\n\tv_fma_mixlo_f16 %1, %5, %8, 0\n\tv_fma_mixlo_f16 %2, %6, %8, 0\n\tv_fma_mixlo_f16 %3, %7, %8, 0\n\t"
      "v_fma_mixhi_f16 %0, %4, %8, -%0 op_sel_hi:[0,0,1]\n\tv_fma_mixhi_f16 %1, %5, %8, -%1 op_sel_hi:[0,0,1]\n\t"
      "v_fma_mixhi_f16 %2, %6, %8, -%2 op_sel_hi:[0,0,1]\n\tv_fma_mixhi_f16 %3, %7, %8, -%3 op_sel_hi:[0,0,1]"
      : "=&v"(o[0]), "=&v"(o[1]), "=&v"(o[2]), "=&v"(o[3])
      : "v"(x[0]), "v"(x[1]), "v"(x[2]), "v"(x[3]), "s"(s));
}
// 2^e with 4 * amax * 2^e < 2^15: amax = m 2^k, m in [1, 2) -> e = 12 - k
__device__ __forceinline__ float wgs_scale(float amax) {
  const int be = (int)((__builtin_bit_cast(unsigned, amax) >> 23) & 255u);
  return __builtin_bit_cast(float, (unsigned)min(max(127 + 12 - (be - 127), 7), 215) << 23);
}

struct WinoWgradSArgs {
  const float *x_amax, *dz_amax;
  const float *x1, *x2, *dz;
  float *ws;  // [splits][co][ci][9], then (want_db) [splits][co] bias-gradient partials
  int want_db;
  int c1, c2, n, h, w, co;
  int64_t x1_img_stride, x2_img_stride, dz_img_stride;
  int x2_div, x2_mul, x2_add;
  int cpr, th, total_chunks, splits, ci_blocks, co_blocks;  // chunks (8 tiles) per tile row, tile rows, n * th * cpr
};

__global__ __launch_bounds__(512, 1) void conv3x3_winograd_wgrad_split_kernel(const WinoWgradSArgs a) {
  constexpr int PS = 2 * 64 * 4;                   // dwords per position of a slab: [tile quad 2][channel 64][tile 4]
  constexpr int SLAB = 16 * PS, SMEM = 4 * SLAB;   // (Z, V) x 2 stages = 128 KB
  constexpr int RSRC_FLAGS = 0x00020000;
  constexpr int OOB = (int)0x80000000;  // >= num_records: the load returns 0 without touching memory
  __shared__ __attribute__((aligned(16))) unsigned smem[SMEM];
  const float s_v = wgs_scale(__builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, *a.x_amax))));
  const float s_z = wgs_scale(__builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, *a.dz_amax))));

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
  const int w_off = ((t >> 2) * 64 + chl) * 4 + (t & 3);  // this thread's dword of position 0 of a slab
  auto commit_v_row = [&](unsigned *Vs, int r) {  // positions xi = 4r .. 4r+3 of B^T (d B): B^T = [[1,0,-1,0],[0,1,1,0],[0,-1,1,0],[0,1,0,-1]]
    const float *ra = tt + (r == 0 ? 0 : r == 1 ? 1 : r == 2 ? 2 : 1) * 4, *rb = tt + (r == 0 ? 2 : r == 1 ? 2 : r == 2 ? 1 : 3) * 4;
    float v[4];
#pragma unroll
    for (int jx = 0; jx < 4; ++jx) v[jx] = r == 1 ? ra[jx] + rb[jx] : ra[jx] - rb[jx];
    unsigned pk[4];
    split4_f16x2(v, s_v, pk);
#pragma unroll
    for (int jx = 0; jx < 4; ++jx) Vs[(4 * r + jx) * PS + w_off] = pk[jx];
  };
  float bvalid = 1.f;  // 0 once the chunk being committed lies beyond this split's range (its dY must not be counted)
  auto commit_z_row = [&](unsigned *Zs, auto SET, int r) {  // row r of A dY A^T, A = [[1,0],[1,1],[1,-1],[0,-1]]
    constexpr int S = decltype(SET)::value;
    const float p = dy[S][0][0], qq = dy[S][0][1], u = dy[S][1][0], v = dy[S][1][1];
    if (r == 0) bsum += bvalid * ((p + qq) + (u + v));
    // rows of A dY: (p, qq), (p + u, qq + v), (p - u, qq - v), (-u, -v)
    const float e = r == 0 ? p : r == 1 ? p + u : r == 2 ? p - u : -u;
    const float f = r == 0 ? qq : r == 1 ? qq + v : r == 2 ? qq - v : -v;
    const float zv[4] = {e, e + f, e - f, -f};
    unsigned pk[4];
    split4_f16x2(zv, s_z, pk);
#pragma unroll
    for (int jx = 0; jx < 4; ++jx) Zs[(4 * r + jx) * PS + w_off] = pk[jx];
  };

  // operands of position xi = 8 ph + p: the 4 tiles (hi, lo) of tile quad `half` of this lane's channel - one 16-byte read each
  const int abase = ((8 * ph * 2 + half) * 64 + wm * 32 + j) * 4;  // A (Z): this wave's co tile
  const int bbase = ((8 * ph * 2 + half) * 64 + wn * 32 + j) * 4;  // B (V): this wave's ci tile
  // One chunk (parity P): 8 positions of 2 MFMAs on LDS stage P; the chunk held in registers (k+1) is transformed into stage 1-P and
  // every register is reloaded with chunk k+2 right after its last use.  Branch-free.
  auto iteration = [&](auto PAR) {
    constexpr int P = decltype(PAR)::value;
    using Other = std::integral_constant<int, 1 - P>;
    const unsigned *Zs = smem + P * 2 * SLAB, *Vs = Zs + SLAB;
    unsigned *Zd = smem + (1 - P) * 2 * SLAB, *Vd = Zd + SLAB;
    i32x4 av[2], bv[2];
    av[0] = *reinterpret_cast<const i32x4 *>(Zs + abase);
    bv[0] = *reinterpret_cast<const i32x4 *>(Vs + bbase);
#pragma unroll
    for (int g = 0; g < 8; ++g) {
      const int cur = g & 1, nxt = cur ^ 1;
      if (g + 1 < 8) {
        av[nxt] = *reinterpret_cast<const i32x4 *>(Zs + abase + (g + 1) * PS);
        bv[nxt] = *reinterpret_cast<const i32x4 *>(Vs + bbase + (g + 1) * PS);
      }
      const f16x8 B = __builtin_bit_cast(f16x8, bv[cur]);
      acc[g] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, av[cur]), B, acc[g], 0, 0, 0);
      i32x4 ar;
#pragma unroll
      for (int i = 0; i < 4; ++i) ar[i] = __builtin_amdgcn_alignbit(av[cur][i], av[cur][i], 16);  // (hi, lo) -> (lo, hi): the cross terms
      if (g == 0) load_dy(PAR);  // set P held chunk k, consumed an iteration ago
      if (g < 4) {
        transform_row(g);
        load_row(g);
      } else {
        commit_v_row(Vd, g - 4);
        commit_z_row(Zd, Other{}, g - 4);
      }
      acc[g] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, ar), B, acc[g], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);  // pin the slice schedule (see winograd.hip)
    }
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
  const float unscale = (1.f / s_v) * (1.f / s_z);  // exact powers of two
  float tr[16][2][3];
#pragma unroll
  for (int r = 0; r < 16; ++r)
#pragma unroll
    for (int rr = 0; rr < 2; ++rr) {
      const float u0 = acc[rr * 4 + 0][r] * unscale, u1 = acc[rr * 4 + 1][r] * unscale, u2 = acc[rr * 4 + 2][r] * unscale, u3 = acc[rr * 4 + 3][r] * unscale;
      tr[r][rr][0] = u0 + 0.5f * (u1 + u2);
      tr[r][rr][1] = 0.5f * (u1 - u2);
      tr[r][rr][2] = 0.5f * (u1 + u2) + u3;
    }
  float *xch = reinterpret_cast<float *>(smem) + quad * (96 * 64) + lane;  // [quad][r * 6 + rr * 3 + jx][lane]
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

bool winograd_wgrad_split_enabled() {
  static const bool on = []() {
    const char *e = getenv("EDVR_WGRAD_SPLIT");  // "0": the fp32 Winograd-domain kernel instead
    return !(e && e[0] == '0');
  }();
  return on;
}

int winograd_wgrad_split_launch(const float *x1, const float *x2, const float *dz, float *ws, int c1, int c2, int n, int h, int w, int co,
                                int64_t x1_img_stride, int64_t x2_img_stride, int x2_div, int x2_mul, int x2_add, int64_t dz_img_stride,
                                int splits, int want_db, const float *x_amax, const float *dz_amax, hipStream_t stream) {
  WinoWgradSArgs a;
  a.x_amax = x_amax; a.dz_amax = dz_amax;
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
  hipLaunchKernelGGL(conv3x3_winograd_wgrad_split_kernel, dim3(splits * a.ci_blocks * a.co_blocks), dim3(512), 0, stream, a);
  return check_launch("conv3x3_winograd_wgrad_split_kernel");
}

}  // namespace edvr
