// winograd_f4s.hip - 3x3 / stride-1 convolution as Winograd F(4x4, 3x3) with SPLIT fp32 operands on the f16 matrix pipe (gfx950).
//
// winograd_f4.hip multiplies in exact fp32 on v_mfma_f32_32x32x2_f32, which runs at 1/16 of the f16 rate and shares its datapath
// with the vector ALUs (every vector instruction of the staging waves costs matrix-pipe time there).  Here every
// fp32 operand is carried as TWO f16 numbers, x * s = hi + lo with hi = f16(x * s), lo = f16(x * s - hi) (s a power of two that
// puts the tensor's largest magnitude just under the f16 range: 22 significant bits, the fp32 product's own rounding level), and
// a product is the sum of all FOUR cross terms, accumulated in fp32 by v_mfma_f32_32x32x16_f16:
//
//   k-slots (2 i, 2 i + 1) of an operand register i hold (hi, lo) of ONE element, so one 16-slot MFMA covers 8 channels:
//     MFMA 1:  A = (Uhi, Ulo), B = (Vhi, Vlo)   ->  Uhi Vhi + Ulo Vlo
//     MFMA 2:  A = (Ulo, Uhi)  (the same registers rotated by 16 bits, one v_alignbit_b32 each)   ->  Ulo Vhi + Uhi Vlo
//   = 2 MFMAs of 32 cycles per 8 channels where the fp32 kernel issues 4 of 64: 4x less matrix-pipe time, and the staging
//   waves' vector work overlaps with it.  Errors: |x s - hi - lo| <= 2^-22 |x s| per operand (fp32: 2^-24), nothing dropped in the
//   product; measured against fp64 the layer error is at or below the fp32 F(4x4) kernel's (tests/test_gpu_conv_f4s.py).
//
// Scales (all exact powers of two, undone by ONE multiplier in the bias fma of the epilogue):
//   V: |B^T d B| <= 100 max|d|; s_V = 2^e with 100 max|d| s_V < 65504, from `x_amax` - ANY upper bound of max|x1|, |x2| on the
//      device (edvr_amax_f32, or the producing kernel's own statistic).  A bound 2^k too large costs nothing until elements fall
//      below 2^-18 of it; below that they keep an ABSOLUTE accuracy of 2^-41 of the bound (f16 subnormals are exact in the MFMA).
//   U: |G g G^T| <= max|g|; s_U from the weights' own maximum, taken by the packing kernels (header of the packed buffer).
//
// Everything else is winograd_f4.hip's structure: wave-specialised 1024-thread workgroups, one per CU, persistent over items of
// 64 output channels x 32 tiles; 4 staging waves (LDS-DMA rows -> packed-fp32 B^T d B -> split -> V slab), 12 multiplying waves
// = (32-channel half, transform row), U straight from global memory in operand order; row pass in the multiplying waves,
// column pass + epilogue + stores in the staging waves.  The V slab holds one dword (hi | lo << 16) where the fp32 kernel holds
// a float: same LDS budget, same addresses.
#include <cstdlib>
#include <type_traits>

#include "common.h"
#include "pack.h"

// experiment switch: 1 = the input transform on plain fp32 instructions (v_fma_f32 / v_add_f32 overlap with the f16 matrix pipe of the
// other waves where v_pk_*_f32 do not - profiles/r5/micro_mfma16_prices.log; -3.5 % on the trunk layer), 0 = packed fp32
#ifndef F4S_PLAIN
#define F4S_PLAIN 1
#endif

namespace edvr {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ float max_raw_s(float a, float b) {  // v_max_f32 without fmaxf()'s canonicalising pre-pass
  float o;
  asm("v_max_f32 %0, %1, %2" : "=v"(o) : "v"(a), "v"(b));
  return o;
}

// the same for six values at once, the six independent v_fma_mixlo_f16 before the six v_fma_mixhi_f16 that depend on them (left to
// itself hipcc issues every pair back to back: a lone wave then waits out the dependent latency 36 times per patch)
__device__ __forceinline__ void split6_f16x2(const float (&x)[6], float s, unsigned (&o)[6]) {
  asm volatile(
      "v_fma_mixlo_f16 %0, %6, %12, 0\n\tv_fma_mixlo_f16 %1, %7, %12, 0\n\tv_fma_mixlo_f16 %2, %8, %12, 0\n\t"
      "v_fma_mixlo_f16 %3, %9, %12, 0\n\tv_fma_mixlo_f16 %4, %10, %12, 0\n\tv_fma_mixlo_f16 %5, %11, %12, 0\n\t"
      "v_fma_mixhi_f16 %0, %6, %12, -%0 op_sel_hi:[0,0,1]\n\tv_fma_mixhi_f16 %1, %7, %12, -%1 op_sel_hi:[0,0,1]\n\t"
      "v_fma_mixhi_f16 %2, %8, %12, -%2 op_sel_hi:[0,0,1]\n\tv_fma_mixhi_f16 %3, %9, %12, -%3 op_sel_hi:[0,0,1]\n\t"
      "v_fma_mixhi_f16 %4, %10, %12, -%4 op_sel_hi:[0,0,1]\n\tv_fma_mixhi_f16 %5, %11, %12, -%5 op_sel_hi:[0,0,1]"
      : "=&v"(o[0]), "=&v"(o[1]), "=&v"(o[2]), "=&v"(o[3]), "=&v"(o[4]), "=&v"(o[5])
      : "v"(x[0]), "v"(x[1]), "v"(x[2]), "v"(x[3]), "v"(x[4]), "v"(x[5]), "s"(s));
}

struct WinoF4SArgs {
  edvr_conv2d_desc d;
  const unsigned *U;  // header (16 dwords: s_U, 1 / s_U) + [co block 64][chunk of 8 channels][row 6][co half 2][position 6][lane 64][4 dwords]
  int ci, ci_real, cop, tiles_x, tiles_y, items;
  float ys, ys_gs;
};

#define F4S_LDS_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")

#ifdef F4S_PROF /* measurement build (scripts/exp/f4s_flags.py prof "-DF4S_PROF"): per wave, cycles (s_memtime) spent waiting at the chunk
                   barriers [0] and the epilogue barriers [1], in the chunk loops [2] and the epilogues [3], and (staging waves) waiting
                   for the LDS-DMA [4]; everything stays in scalar registers (the multiplying waves have no vector register to
                   spare) and is added to f4s_prof[wave][.] once, at the end.  Two s_memtime per barrier: ~3 % on the launch. */
__device__ unsigned long long f4s_prof[16 * 8];
#define F4S_NOW(v) asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(v)::"memory")
#define F4S_BARRIER_AT(slot)                                        \
  do {                                                              \
    unsigned long long a_, b_;                                      \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");              \
    F4S_NOW(a_);                                                    \
    asm volatile("s_barrier" ::: "memory");                         \
    F4S_NOW(b_);                                                    \
    f4s_pc[slot] += (unsigned)(b_ - a_);                            \
  } while (0)
#define F4S_SPAN_BEGIN() F4S_NOW(f4s_t0)
#define F4S_SPAN_END(slot)                                          \
  do {                                                              \
    unsigned long long b_;                                          \
    F4S_NOW(b_);                                                    \
    f4s_pc[slot] += (unsigned)(b_ - f4s_t0);                        \
    f4s_t0 = b_;                                                    \
  } while (0)
#define F4S_WAIT_DMA()                                              \
  do {                                                              \
    unsigned long long a_, b_;                                      \
    F4S_NOW(a_);                                                    \
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                \
    F4S_NOW(b_);                                                    \
    f4s_pc[4] += (unsigned)(b_ - a_);                               \
  } while (0)
#define F4S_PROF_FLUSH()                                                                                         \
  do {                                                                                                          \
    if ((threadIdx.x & 63) == 0)                                                                                \
      for (int i_ = 0; i_ < 8; ++i_) atomicAdd(&f4s_prof[(threadIdx.x >> 6) * 8 + i_], (unsigned long long)f4s_pc[i_]); \
  } while (0)
#else
#define F4S_BARRIER_AT(slot) F4S_LDS_BARRIER()
#define F4S_SPAN_BEGIN()
#define F4S_SPAN_END(slot)
#define F4S_WAIT_DMA() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")
#define F4S_PROF_FLUSH()
#endif
#if defined(F4S_PROF) && F4S_PROF >= 2
#define F4S_POS_T0() unsigned long long pa_, pb_, pc_; F4S_NOW(pa_)
#define F4S_POS_T1()                                                \
  do {                                                              \
    F4S_NOW(pb_);                                                   \
    asm volatile("s_waitcnt vmcnt(5)" ::: "memory");                \
    F4S_NOW(pc_);                                                   \
    f4s_pc[5] += (unsigned)(pb_ - pa_);                             \
    f4s_pc[6] += (unsigned)(pc_ - pb_);                             \
  } while (0)
#else
#define F4S_POS_T0()
#define F4S_POS_T1()
#endif
#define F4S_BARRIER() F4S_BARRIER_AT(7)

// s_V = 2^e, the largest power of two with 100 * amax * s_V < 65504 (|B^T d B| <= 100 max|d|): amax = m 2^k, m in [1, 2) -> e = 8 - k.
// Zero / tiny bounds stop at 2^100 (inputs below 2^-92 lose relative accuracy gradually; the unscaling factor stays a normal number), infinities and NaNs give a harmless 2^-120.
__device__ __forceinline__ float f4s_input_scale(float amax) {
  const int be = (int)((__builtin_bit_cast(unsigned, amax) >> 23) & 255u);  // biased exponent of the bound
  const int field = min(max(127 + 8 - (be - 127), 7), 227);
  return __builtin_bit_cast(float, (unsigned)field << 23);
}

template <int TXL>
__global__ __launch_bounds__(1024, 1) void conv3x3_winograd_f4s_kernel(const WinoF4SArgs a) {
  constexpr int TX = 1 << TXL, TY = 32 / TX;               // tiles per block row / rows
  constexpr int BW = 4 * TX, BH = 4 * TY;                  // output pixels of a block
  constexpr int RROWS = BH + 2, RPIECES = TX + 2;          // raw input rows / 16-byte pieces per row (columns tx0 - 4 .. tx0 + BW + 3)
  static_assert(RROWS * RPIECES == 180, "a wave's region holds 2 x 180 pieces either way");
  constexpr int CK = 8;
  constexpr int VSLAB = CK * 36 * 32;      // dwords per V stage (36 KB): [channel 8][position 36][tile 32], dword = (hi | lo << 16)
  constexpr int XSZ = 2 * 6 * 8 * 32 * 4;  // exchange area (2 x 24 KB): [phase parity][row 6][channel 8][tile 32][4]
  constexpr int RWAVE = 6 * 64 * 4 + 4;    // raw-input region of one staging wave: 6 DMA instructions x 64 lanes x 16 B, + one-dword shift
  constexpr int UCHUNK = 6 * 2 * 6 * 1024; // bytes of U per (co block, 8-channel chunk)
  __shared__ __attribute__((aligned(16))) float smem[2 * VSLAB + XSZ + 4 * RWAVE + 64];  // 144 KB
  float *const Xs = smem + 2 * VSLAB;
  float *const bias_s = smem + 2 * VSLAB + XSZ + 4 * RWAVE;

  const edvr_conv2d_desc &d = a.d;
  const int tid = threadIdx.x, lane = tid & 63, half = lane >> 5, j = lane & 31;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int hw = d.h * d.w, plane_bytes = hw * 4;
  const int co_blocks = (d.co + 63) / 64;
  const int n_chunks = a.ci / CK;
  constexpr int RSRC_FLAGS = 0x00020000;  // raw buffer, 32-bit data format (gfx9 family)
  auto uniform_rsrc = [&](const void *p, int bytes) {
    const uint64_t pv = reinterpret_cast<uint64_t>(p);
    const uint64_t pu = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(pv >> 32)) << 32) |
                        (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)pv);
    return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void *>(pu), (short)0, bytes, RSRC_FLAGS);
  };

  // Persistent workgroups, XCD-aware walk: every XCD gets one contiguous range of items.
  const int n_xcd = gridDim.x < 8 ? 1 : 8;
  const int xcd = n_xcd == 1 ? 0 : (int)blockIdx.x % 8, xcd_rank = n_xcd == 1 ? (int)blockIdx.x : (int)blockIdx.x / 8;
  const int xcd_wgs = n_xcd == 1 ? (int)gridDim.x : ((int)gridDim.x - xcd + 7) / 8;
  const int span = (a.items + n_xcd - 1) / n_xcd;
  const int item_end = min(a.items, (xcd + 1) * span);
  const int item_first = xcd * span + xcd_rank;
  if (item_first >= item_end) return;
#ifdef F4S_PROF
  unsigned f4s_pc[8] = {0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u};
  unsigned long long f4s_t0 = 0;
#endif
  auto decode = [&](int item, int &co_blk, int &img, int &ty0, int &tx0) {
    co_blk = __builtin_amdgcn_readfirstlane((item % co_blocks) * 64);
    const int tile_blk = __builtin_amdgcn_readfirstlane((item / co_blocks) % (a.tiles_x * a.tiles_y));
    img = __builtin_amdgcn_readfirstlane(item / (co_blocks * a.tiles_x * a.tiles_y));
    ty0 = __builtin_amdgcn_readfirstlane((tile_blk / a.tiles_x) * BH);
    tx0 = __builtin_amdgcn_readfirstlane((tile_blk % a.tiles_x) * BW);
  };

  if (wave < 4) {
    // =========================================================================================== staging waves
    __builtin_amdgcn_s_setprio(3);
    const float s_v = f4s_input_scale(__builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, *d.x_amax))));
    const float s_u_inv = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane((int)a.U[1]));
    const float unscale = s_u_inv * (1.f / s_v);  // M = M' / (s_U s_V): an exact power of two
    const int p_ty = j >> TXL, p_tx = j & (TX - 1);  // tile j of the TY x TX; channel 2 wave + half of the chunk
    // Raw input of this wave's two channels for one chunk: [channel 2][row BH + 2][(TX + 2) x 16 B] = image rows ty0 - 1 .. ty0 + BH,
    // columns tx0 - 4 .. tx0 + BW + 3, fetched by LDS-DMA (lane l of instruction i delivers 16-byte piece 64 i + l; pieces outside
    // the image get an out-of-range offset and arrive as zeros = the padding).  The region is PRIVATE to the wave: read the patches
    // of chunk k + 1, then request chunk k + 2 into the same place.
    typedef __attribute__((address_space(3))) void lvoid;
    constexpr int OOB = (int)0x80000000;
    float *const Rw = smem + 2 * VSLAB + XSZ + wave * RWAVE;
    int dma_off[6];
    const float *x1 = d.x1, *x2 = d.x1;
    int l_item = item_first, l_k = 0;  // load cursor: (item, chunk) the NEXT request belongs to
    auto setup = [&](int item) {
      int cb, img, ty0, tx0;
      decode(item, cb, img, ty0, tx0);
      x1 = d.x1 + (int64_t)img * d.x1_img_stride;
      x2 = x1;
      if (d.x2) {
        const int i2 = d.x2_div > 0 ? (img / d.x2_div) * d.x2_mul + d.x2_add : img;
        x2 = d.x2 + (int64_t)i2 * d.x2_img_stride;
      }
#pragma unroll
      for (int i = 0; i < 6; ++i) {
        const int q = i * 64 + lane, ch = q / 180, rem = q - ch * 180, row = rem / RPIECES, cx = rem - row * RPIECES;
        const int gy = ty0 - 1 + row, gx = tx0 - 4 + 4 * cx;  // w % 4 == 0: a piece is inside or outside the row as a whole
        dma_off[i] = (q < 360 && gy >= 0 && gy < d.h && gx >= 0 && gx < d.w) ? (ch * hw + gy * d.w + gx) * 4 : OOB;
      }
    };
    __amdgpu_buffer_rsrc_t ld_rsrc = uniform_rsrc(d.x1, 0);
    auto load_begin = [&](int c0) {
      const int c = c0 + 2 * wave;  // even; c1 is even when there is an x2 (host check): the pair never straddles x1 / x2
      const float *pl = (c < d.c1) ? (x1 + (int64_t)c * hw) : (x2 + (int64_t)(c - d.c1) * hw);
      const int nvalid = a.ci_real - c;  // channels of the padding: empty (or one-plane) buffer, their loads return 0
      ld_rsrc = uniform_rsrc(pl, nvalid >= 2 ? 2 * plane_bytes : (nvalid == 1 ? plane_bytes : 0));
    };
    auto dma_issue = [&]() {
#pragma unroll
      for (int i = 0; i < 6; ++i) __builtin_amdgcn_raw_ptr_buffer_load_lds(ld_rsrc, (lvoid *)(Rw + 1 + i * 256), 16, dma_off[i], 0, 0, 0);
    };
    // Packed-fp32 input transform (winograd_f4.hip): B^T d on pairs of adjacent columns, (.) B on one row held as the same pairs.
    f32x2 pp[6][3];  // raw patch: row r, columns (2 cp, 2 cp + 1)
    f32x2 tp[6][3];  // B^T d, same pairing
    const float *patch = Rw + ((half * RROWS + 4 * p_ty) * RPIECES + p_tx) * 4 + 4;  // patch row r, column c: patch[r * 4 RPIECES + c]
    auto read_patch = [&]() {
      F4S_WAIT_DMA();  // s_waitcnt vmcnt(0): the requested chunk is in the region
#pragma unroll
      for (int r = 0; r < 6; ++r) {
        const f32x4 m = *reinterpret_cast<const f32x4 *>(patch + r * (4 * RPIECES));
        pp[r][0] = f32x2{m[0], m[1]};
        pp[r][1] = f32x2{m[2], m[3]};
        pp[r][2] = *reinterpret_cast<const f32x2 *>(patch + r * (4 * RPIECES) + 4);
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // ... and in registers: the region may be overwritten
    };
#if F4S_PLAIN
    // experiment: the same arithmetic on plain fp32 instructions (v_fma_f32 / v_add_f32 overlap with the f16 matrix pipe of the other
    // waves where v_pk_*_f32 do not - profiles/r5/micro_mfma16_prices.log), built with -fno-slp-vectorize
    float tq[6][6];
    auto transform_cols = [&](int cp) {
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        const float d0 = pp[0][cp][e], d1 = pp[1][cp][e], d2 = pp[2][cp][e], d3 = pp[3][cp][e], d4 = pp[4][cp][e], d5 = pp[5][cp][e];
        const float p_ = d4 - 4.f * d2, q_ = d3 - 4.f * d1, r_ = d4 - d2, s_ = d3 - d1;
        tq[0][2 * cp + e] = 4.f * d0 + (d4 - 5.f * d2);
        tq[1][2 * cp + e] = p_ + q_;
        tq[2][2 * cp + e] = p_ - q_;
        tq[3][2 * cp + e] = r_ + 2.f * s_;
        tq[4][2 * cp + e] = r_ - 2.f * s_;
        tq[5][2 * cp + e] = 4.f * d1 + (d5 - 5.f * d3);
      }
    };
    auto commit_row = [&](unsigned *Vd, int r) {
      const float d0 = tq[r][0], d1 = tq[r][1], d2 = tq[r][2], d3 = tq[r][3], d4 = tq[r][4], d5 = tq[r][5];
      const float p_ = d4 - 4.f * d2, q_ = d3 - 4.f * d1, r_ = d4 - d2, s_ = d3 - d1;
      const float t[6] = {4.f * d0 + (d4 - 5.f * d2), p_ + q_, p_ - q_, r_ + 2.f * s_, r_ - 2.f * s_, 4.f * d1 + (d5 - 5.f * d3)};
      unsigned *dst = Vd + ((2 * wave + half) * 36 + r * 6) * 32 + j;
      unsigned pk[6];
      split6_f16x2(t, s_v, pk);
#pragma unroll
      for (int c = 0; c < 6; ++c) dst[c * 32] = pk[c];
    };
#else
    auto transform_cols = [&](int cp) {  // 1-D input transform B^T (Lavin & Gray), 12 packed operations
      const f32x2 d0 = pp[0][cp], d1 = pp[1][cp], d2 = pp[2][cp], d3 = pp[3][cp], d4 = pp[4][cp], d5 = pp[5][cp];
      const f32x2 p_ = d4 - 4.f * d2, q_ = d3 - 4.f * d1, r_ = d4 - d2, s_ = d3 - d1;
      tp[0][cp] = 4.f * d0 + (d4 - 5.f * d2);
      tp[1][cp] = p_ + q_;
      tp[2][cp] = p_ - q_;
      tp[3][cp] = r_ + 2.f * s_;
      tp[4][cp] = r_ - 2.f * s_;
      tp[5][cp] = 4.f * d1 + (d5 - 5.f * d3);
    };
    auto commit_row = [&](unsigned *Vd, int r) {  // positions (r, 0..5) of (B^T d) B, each split into (hi, lo)
      const f32x2 P0 = tp[r][0], P1 = tp[r][1], P2 = tp[r][2];
      const f32x2 lo1 = __builtin_shufflevector(P1, P1, 0, 0), hi1 = __builtin_shufflevector(P1, P1, 1, 1);  // d2, d3
      const f32x2 lo2 = __builtin_shufflevector(P2, P2, 0, 0), hi0 = __builtin_shufflevector(P0, P0, 1, 1);  // d4, d1
      const f32x2 t05 = 4.f * P0 + (P2 - 5.f * P1);
      const f32x2 pr_ = lo2 + f32x2{-4.f, -1.f} * lo1;
      const f32x2 qs_ = hi1 + f32x2{-4.f, -1.f} * hi0;
      const f32x2 t13 = pr_ + f32x2{1.f, 2.f} * qs_;
      const f32x2 t24 = pr_ - f32x2{1.f, 2.f} * qs_;
      unsigned *dst = Vd + ((2 * wave + half) * 36 + r * 6) * 32 + j;
      const float t[6] = {t05[0], t13[0], t24[0], t13[1], t24[1], t05[1]};
      unsigned pk[6];
      split6_f16x2(t, s_v, pk);
#pragma unroll
      for (int c = 0; c < 6; ++c) dst[c * 32] = pk[c];
    };
#endif
    auto advance = [&]() {  // the load cursor moves one chunk; at an item boundary the geometry switches
      if (++l_k == n_chunks) {
        l_k = 0;
        const int nx = l_item + xcd_wgs;
        l_item = nx < item_end ? nx : l_item;  // past the last item: re-stage it (never consumed)
        setup(l_item);
      }
    };
    unsigned *const Vst = reinterpret_cast<unsigned *>(smem);

    // ---- prologue: chunk 0 -> region -> registers -> stage 0; chunk 1 requested
    setup(item_first);
    load_begin(0);
    dma_issue();
    read_patch();
    advance();
    load_begin(l_k * CK);
    dma_issue();
#pragma unroll
    for (int cp = 0; cp < 3; ++cp) transform_cols(cp);
#pragma unroll
    for (int r = 0; r < 6; ++r) commit_row(Vst, r);
    F4S_BARRIER();

    int par = 0;  // stage the multiplying waves read during the current step
    float asum = 0.f;  // abs_sum epilogue (edvr_conv2d_desc.abs_sum), as in winograd_f4.hip
    int asum_img = -1;
    auto asum_flush = [&]() {
      float s = asum;
#pragma unroll
      for (int sh = 32; sh > 0; sh >>= 1) s += __shfl_xor(s, sh);
      if (lane == 0 && asum_img >= 0 && s != 0.f) atomicAdd(d.abs_sum + asum_img, s);
      asum = 0.f;
    };
    // y_amax epilogue: max |y| of everything this thread stores (rows below the image included: still a bound), kept as a BIT PATTERN
    // (non-negative floats order as unsigned integers, and every NaN orders above +inf): a non-finite output is STICKY where
    // v_max_f32 would drop a NaN - the host's overflow guard reads these slots (ops.split_guard_submit).
    unsigned amx = 0u;
    auto abits = [](float v) { return __builtin_bit_cast(unsigned, v) & 0x7fffffffu; };
    // max |.| of one 4x4 output tile.  Y[0][0], Y[0][3], Y[3][0], Y[3][3] together depend on all 36 positions of M (A^T rows 0 and 3
    // cover columns 0..4 and 1..5), and 0 * (inf or NaN) = NaN: four fmas see an overflow anywhere in the tile's products
    auto amax16 = [&](const f32x4 (&Y)[4]) {
      float m = fmaxf(fmaxf(fabsf(Y[0][0]), fabsf(Y[0][1])), fmaxf(fabsf(Y[0][2]), fabsf(Y[0][3])));
#pragma unroll
      for (int i = 1; i < 4; ++i) m = fmaxf(fmaxf(fmaxf(m, fabsf(Y[i][0])), fmaxf(fabsf(Y[i][1]), fabsf(Y[i][2]))), fabsf(Y[i][3]));
      const float chk = __builtin_fmaf(Y[0][0], 0.f, __builtin_fmaf(Y[0][3], 0.f, __builtin_fmaf(Y[3][0], 0.f, Y[3][3] * 0.f)));
      amx = max(max(amx, __builtin_bit_cast(unsigned, m)), abits(chk));
    };
    F4S_SPAN_BEGIN();
    for (int item = item_first; item < item_end; item += xcd_wgs) {
      int e_co_blk, e_img, e_ty0, e_tx0;
      decode(item, e_co_blk, e_img, e_ty0, e_tx0);
      if (d.abs_sum && e_img != asum_img) {
        asum_flush();
        asum_img = e_img;
      }
#pragma unroll 1
      for (int k = 0; k < n_chunks; ++k) {
        unsigned *Vd = Vst + (par ^ 1) * VSLAB;
        read_patch();
        advance();
        load_begin(l_k * CK);
        dma_issue();
#pragma unroll
        for (int cp = 0; cp < 3; ++cp) transform_cols(cp);
#pragma unroll
        for (int r = 0; r < 6; ++r) commit_row(Vd, r);
        F4S_BARRIER_AT(0);
        par ^= 1;
      }
      F4S_SPAN_END(2);

      // ---- column pass Y = A^T T + epilogue + stores: 8 phases of 8 output channels, one (channel, tile) per thread and phase.
      const int plane = hw;
      float *y = d.y + (int64_t)e_img * d.y_img_stride;
      const float *r1 = d.res1 ? d.res1 + (int64_t)e_img * d.res1_img_stride : nullptr;
      const float *r2 = d.res2 ? d.res2 + (int64_t)e_img * d.res2_img_stride : nullptr;
      const float *gt = d.gate ? d.gate + (int64_t)e_img * d.gate_img_stride : nullptr;
      const float *rq = gt ? gt : r1;  // the tensor read per output element (gate and residuals exclude each other)
      const float slope = d.act == EDVR_ACT_LRELU ? 0.1f : (d.act == EDVR_ACT_RELU ? 0.f : 1.f);  // none/relu/lrelu = max(v, slope*v)
      const bool sig = d.act == EDVR_ACT_SIGMOID, shuffle = d.out_mode == EDVR_OUT_PIXEL_SHUFFLE2;
      const bool vec = e_tx0 + BW <= d.w;  // the block is inside the image in x (w % 4 == 0): 16-byte rows, only the ROW is tested
      const int cl8 = (wave * 64 + lane) >> 5, tile = lane & 31;
      const int oy = e_ty0 + 4 * (tile >> TXL), ox = e_tx0 + 4 * (tile & (TX - 1));
      const int co_t = e_co_blk + (cl8 >> 2) * 32 + 4 * ((cl8 >> 1) & 1) + 8 * (cl8 & 1);  // + (p & 3) + 16 (p >> 2) in phase p
      const int pix = oy * d.w + ox;
      const int rows_in = d.h - oy;  // rows of this lane's tile inside the image (>= 4: all of them)
      if (wave == 0) bias_s[lane] = (d.bias && e_co_blk + lane < d.co) ? d.bias[e_co_blk + lane] : 0.f;  // read after the first phase barrier
      auto column_pass = [&](auto VEC, auto SHUF) {
        constexpr bool V = decltype(VEC)::value;     // whole 16-byte rows inside the image in x: no per-element tests
        constexpr bool SHF = decltype(SHUF)::value;  // V && PixelShuffle(2)
        f32x4 Yprev[4];
        f32x4 rr[4];
        auto co_of = [&](int p) { return co_t + (p & 3) + 16 * (p >> 2); };
        auto prefetch = [&](int p) {  // (V) rows oy .. oy + 3 of the residual(s) / gate of channel co_of(p)
          const int co = min(co_of(p), d.co - 1);
          if (V && !SHF && rq) {
            const float *q1 = rq + (int64_t)co * plane + pix;
#pragma unroll
            for (int i = 0; i < 4; ++i) rr[i] = i < rows_in ? *reinterpret_cast<const f32x4 *>(q1 + i * d.w) : f32x4{0.f, 0.f, 0.f, 0.f};
            if (r2) {
              const float *q2 = r2 + (int64_t)co * plane + pix;
#pragma unroll
              for (int i = 0; i < 4; ++i)
                if (i < rows_in) rr[i] += *reinterpret_cast<const f32x4 *>(q2 + i * d.w);
            }
          }
        };
        prefetch(0);
#pragma unroll 1
        for (int p = 0; p < 8; ++p) {
          F4S_BARRIER_AT(1);  // T of this phase is in its half of the exchange area
          const float *Xb = Xs + (p & 1) * (XSZ / 2) + (cl8 * 32 + tile) * 4;
          f32x4 T[6];
#pragma unroll
          for (int r = 0; r < 6; ++r) T[r] = *reinterpret_cast<const f32x4 *>(Xb + r * (8 * 32 * 4));
          const int co = co_of(p);
          const float b = bias_s[co - e_co_blk];
          const float sl = co >= d.act_from ? slope : 1.f;
          f32x4 Y[4];
          // A^T along the rows (6 -> 4) on PACKED fp32 pairs of columns (the matrix pipe is idle during the output pass, and the lone
          // staging wave of a SIMD pays per instruction, not per lane-operation); the operand scales leave in the bias fma
#pragma unroll
          for (int jp = 0; jp < 2; ++jp) {
            auto col2 = [&](int r) { return f32x2{T[r][2 * jp], T[r][2 * jp + 1]}; };
            const f32x2 t0 = col2(0), t1 = col2(1), t2 = col2(2), t3 = col2(3), t4 = col2(4), t5 = col2(5);
            const f32x2 s1 = t1 + t2, d1 = t1 - t2, s2 = t3 + t4, d2 = t3 - t4;
            const f32x2 us = f32x2{unscale, unscale}, bb = f32x2{b, b};
            f32x2 y0 = (t0 + s1 + s2) * us + bb;
            f32x2 y1 = (2.f * d2 + d1) * us + bb;
            f32x2 y2 = (4.f * s2 + s1) * us + bb;
            f32x2 y3 = (8.f * d2 + d1 + t5) * us + bb;
            if (!sig) {  // none / relu / lrelu = max(v, slope * v): one packed multiply + two v_max_f32 per pair
              const f32x2 sl2 = f32x2{sl, sl};
              const f32x2 z0 = y0 * sl2, z1 = y1 * sl2, z2 = y2 * sl2, z3 = y3 * sl2;
              y0 = f32x2{max_raw_s(y0[0], z0[0]), max_raw_s(y0[1], z0[1])};
              y1 = f32x2{max_raw_s(y1[0], z1[0]), max_raw_s(y1[1], z1[1])};
              y2 = f32x2{max_raw_s(y2[0], z2[0]), max_raw_s(y2[1], z2[1])};
              y3 = f32x2{max_raw_s(y3[0], z3[0]), max_raw_s(y3[1], z3[1])};
            }
            Y[0][2 * jp] = y0[0]; Y[0][2 * jp + 1] = y0[1];
            Y[1][2 * jp] = y1[0]; Y[1][2 * jp + 1] = y1[1];
            Y[2][2 * jp] = y2[0]; Y[2][2 * jp + 1] = y2[1];
            Y[3][2 * jp] = y3[0]; Y[3][2 * jp + 1] = y3[1];
          }
          if (sig && co >= d.act_from) {
            if (d.y_amax) amx = max(amx, abits(__builtin_fmaf(Y[0][0], 0.f, __builtin_fmaf(Y[0][3], 0.f, __builtin_fmaf(Y[3][0], 0.f, Y[3][3] * 0.f)))));  // (a sigmoid maps +-inf to finite values)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
              for (int jj = 0; jj < 4; ++jj) Y[i][jj] = __builtin_amdgcn_rcpf(1.f + __expf(-Y[i][jj]));
          }
          if (SHF) {
            prefetch(min(p + 1, 7));
            if (d.y_amax) amax16(Y);
            if ((p & 1) == 0) {
#pragma unroll
              for (int i = 0; i < 4; ++i) Yprev[i] = Y[i];
            } else if (co < d.co) {  // co odd; co - 1 is in Yprev.  Output plane co >> 2, row 2 y + ((co >> 1) & 1), columns 2 x + (co & 1)
              float *q = y + (int64_t)(co >> 2) * plane * 4 + (2 * oy + ((co >> 1) & 1)) * (2 * d.w) + 2 * ox;
#pragma unroll
              for (int i = 0; i < 4; ++i)
                if (i < rows_in) {
                  *reinterpret_cast<f32x4 *>(q + i * 4 * d.w) = f32x4{Yprev[i][0], Y[i][0], Yprev[i][1], Y[i][1]};
                  *reinterpret_cast<f32x4 *>(q + i * 4 * d.w + 4) = f32x4{Yprev[i][2], Y[i][2], Yprev[i][3], Y[i][3]};
                }
            }
          } else if (V) {
            if (d.abs_sum && co < d.abs_sum_channels) {
#pragma unroll
              for (int i = 0; i < 4; ++i)
                if (i < rows_in) asum += (fabsf(Y[i][0]) + fabsf(Y[i][1])) + (fabsf(Y[i][2]) + fabsf(Y[i][3]));
            }
            if (gt) {
#pragma unroll
              for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) Y[i][jj] *= rr[i][jj] > 0.f ? a.ys : a.ys_gs;
            } else if (r1) {
#pragma unroll
              for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) Y[i][jj] = __builtin_fmaf(Y[i][jj], a.ys, rr[i][jj]);
            }
            prefetch(min(p + 1, 7));
            if (d.y_amax) amax16(Y);
            if (co < d.co) {
              float *q = y + (int64_t)co * plane + pix;
#pragma unroll
              for (int i = 0; i < 4; ++i)
                if (i < rows_in) *reinterpret_cast<f32x4 *>(q + i * d.w) = Y[i];
            }
          } else if (co < d.co) {
#pragma unroll 1
            for (int i = 0; i < 4; ++i) {
              if (oy + i >= d.h) break;
              const int64_t off = (int64_t)co * plane + pix + i * d.w;
#pragma unroll
              for (int jj = 0; jj < 4; ++jj) {
                if (ox + jj < d.w) {
                  float o = Y[i][jj];
                  if (d.abs_sum && co < d.abs_sum_channels) asum += fabsf(o);
                  if (gt) o *= gt[off + jj] > 0.f ? a.ys : a.ys_gs;
                  else if (r1) o = __builtin_fmaf(o, a.ys, r1[off + jj] + (r2 ? r2[off + jj] : 0.f));
                  amx = max(amx, abits(o));
                  if (shuffle)
                    y[(int64_t)(co >> 2) * plane * 4 + (2 * (oy + i) + ((co >> 1) & 1)) * (2 * d.w) + 2 * (ox + jj) + (co & 1)] = o;
                  else
                    y[off + jj] = o;
                }
              }
            }
            prefetch(min(p + 1, 7));
          } else {
            prefetch(min(p + 1, 7));
          }
        }
      };
      if (vec && shuffle) column_pass(std::true_type{}, std::true_type{});
      else if (vec) column_pass(std::true_type{}, std::false_type{});
      else column_pass(std::false_type{}, std::false_type{});
      F4S_SPAN_END(3);
    }
    if (d.abs_sum) asum_flush();
    if (d.y_amax) {
#pragma unroll
      for (int sh = 32; sh > 0; sh >>= 1) amx = max(amx, (unsigned)__shfl_xor((int)amx, sh));
      if (lane == 0 && amx > 0u) atomicMax(reinterpret_cast<unsigned *>(d.y_amax), amx);  // non-negative floats order as integers, NaNs above them
    }
    F4S_PROF_FLUSH();
  } else {
    // =========================================================================================== multiplying waves
    const int q = wave - 4, wm = q & 1, row = q >> 1;
    const __amdgpu_buffer_rsrc_t u_rsrc = uniform_rsrc(a.U, 64 + a.cop * a.ci * 36 * 4);
    const int voff = lane * 16;
    f32x16 acc[6];
    i32x4 A[6];  // A operand of position (row, c): 4 channels x (hi, lo); a set is re-requested right after its use, one chunk ahead
    int u_base = 0, u_next = 0;  // byte offsets of (co block, chunk 0, row, wm) of this item and of the next
    auto item_base = [&](int co_blk) { return 64 + ((((co_blk >> 6) * n_chunks) * 6 + row) * 2 + wm) * (6 * 1024); };
    auto load_a = [&](int c, int soff) { A[c] = __builtin_bit_cast(i32x4, __builtin_amdgcn_raw_buffer_load_b128(u_rsrc, voff, soff + c * 1024, 0)); };
    int co_blk, img_, ty_, tx_;
    decode(item_first, co_blk, img_, ty_, tx_);
    u_base = item_base(co_blk);
#pragma unroll
    for (int c = 0; c < 6; ++c) load_a(c, u_base);
#pragma unroll
    for (int c = 0; c < 6; ++c)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;
    F4S_BARRIER();

    int par = 0;
    const unsigned *const Vst = reinterpret_cast<const unsigned *>(smem);
    // B operand: channels 4 half .. + 3, position (row, c), tile j of the current stage (LDS byte address; smem is the first allocation)
    unsigned v_addr = (unsigned)(size_t)(__attribute__((address_space(3))) const unsigned *)(Vst + 4 * half * 36 * 32 + row * 6 * 32 + j);
    int v_step = __builtin_amdgcn_readfirstlane(VSLAB * 4);
    F4S_SPAN_BEGIN();
    for (int item = item_first; item < item_end; item += xcd_wgs) {
      {
        const int nx = item + xcd_wgs;
        decode(nx < item_end ? nx : item, co_blk, img_, ty_, tx_);
        u_next = item_base(co_blk);
      }
#pragma unroll 1
      for (int k = 0; k < n_chunks; ++k) {
        const int soff_nxt = k + 1 < n_chunks ? u_base + (k + 1) * UCHUNK : u_next;  // (at the end: warms the next item's first chunk; re-requested below)
#pragma unroll
        for (int c = 0; c < 6; ++c) {
          // B operand: four dwords 36 x 32 apart.  Written as four ds_read_b32 with immediate offsets from ONE address register
          // (hipcc pairs them into ds_read2st64_b32, whose 64-dword offset unit needs a second base for the odd positions - and
          // 96 accumulators + 24 registers of A + 4 of B leave exactly four); single-buffered, the two other waves of the SIMD
          // cover the LDS latency.
          int b0, b1, b2, b3;
          F4S_POS_T0();
          asm volatile("ds_read_b32 %0, %4 offset:%5\n\tds_read_b32 %1, %4 offset:%6\n\tds_read_b32 %2, %4 offset:%7\n\tds_read_b32 %3, %4 offset:%8\n\t"
                       "s_waitcnt lgkmcnt(0)"
                       : "=&v"(b0), "=&v"(b1), "=&v"(b2), "=&v"(b3)
                       : "v"(v_addr), "n"(c * 128), "n"(36 * 128 + c * 128), "n"(2 * 36 * 128 + c * 128), "n"(3 * 36 * 128 + c * 128)
                       : "memory");
          F4S_POS_T1();  // (second-level measurement build only: B wait [5], then the wait for this position's A [6])
          const f16x8 B = __builtin_bit_cast(f16x8, i32x4{b0, b1, b2, b3});
          acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, A[c]), B, acc[c], 0, 0, 0);
          // (hi, lo) -> (lo, hi) IN PLACE (no second register set), then the set is re-requested for the next chunk
          asm volatile("v_alignbit_b32 %0, %0, %0, 16\n\tv_alignbit_b32 %1, %1, %1, 16\n\tv_alignbit_b32 %2, %2, %2, 16\n\tv_alignbit_b32 %3, %3, %3, 16"
                       : "+v"(A[c][0]), "+v"(A[c][1]), "+v"(A[c][2]), "+v"(A[c][3]));
          acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, A[c]), B, acc[c], 0, 0, 0);
          __builtin_amdgcn_sched_barrier(0);
          load_a(c, soff_nxt);  // the same position of the next chunk: a whole chunk step to arrive
          __builtin_amdgcn_sched_barrier(0);
        }
        v_addr += v_step;  // the other stage
        v_step = -v_step;
        F4S_BARRIER_AT(0);
        par ^= 1;
      }
      u_base = u_next;
      F4S_SPAN_END(2);
      // ---- row pass T = M A (6 -> 4) and hand-over to the staging waves: 8 phases of two accumulator registers.  The store address
      //      is derived HERE from an opaque copy of the lane index: hoisted out of the item loop it stays live across the chunk loop,
      //      whose 96 + 24 + 4 registers leave no room for it (it was spilled and reloaded once per item)
      int lane_e = lane;
      asm volatile("" : "+v"(lane_e));
      float *const Xw = Xs + ((row * 8 + wm * 4 + (lane_e >> 5) * 2) * 32 + (lane_e & 31)) * 4;
#pragma unroll
      for (int p = 0; p < 8; ++p) {
        float *Xb = Xw + (p & 1) * (XSZ / 2);
#pragma unroll
        for (int rr = 0; rr < 2; ++rr) {
          const int r = (p & 3) + 4 * (2 * (p >> 2) + rr);
          const float m0 = acc[0][r], m1 = acc[1][r], m2 = acc[2][r], m3 = acc[3][r], m4 = acc[4][r], m5 = acc[5][r];
          const float s1 = m1 + m2, d1 = m1 - m2, s2 = m3 + m4, d2 = m3 - m4;
          f32x4 T;
          T[0] = m0 + s1 + s2;
          T[1] = __builtin_fmaf(2.f, d2, d1);
          T[2] = __builtin_fmaf(4.f, s2, s1);
          T[3] = __builtin_fmaf(8.f, d2, d1) + m5;
          *reinterpret_cast<f32x4 *>(Xb + rr * 32 * 4) = T;
        }
        F4S_BARRIER_AT(1);
      }
      // the next item's first chunk of U: requested here, not from the last chunk step - 24 registers live across the row pass do not fit
#pragma unroll
      for (int c = 0; c < 6; ++c) load_a(c, u_base);
#pragma unroll
      for (int c = 0; c < 6; ++c)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;
      F4S_SPAN_END(3);
    }
    F4S_PROF_FLUSH();
  }
}

// ------------------------------------------------------------------------------------------------ weights
// header[0] = s_U = 2^e with max|g| s_U in [2^14, 2^15) (|G g G^T| <= max|g|), header[1] = 1 / s_U
__global__ __launch_bounds__(1024) void winograd_f4s_weight_scale_kernel(const float *__restrict__ w, unsigned *__restrict__ U, int64_t total) {
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
    U[0] = field << 23;
    U[1] = (254u - field) << 23;
    for (int i = 2; i < 16; ++i) U[i] = 0u;
  }
}

__global__ void winograd_f4s_weight_kernel(const float *__restrict__ w, unsigned *__restrict__ U, int co, int ci, int cop, int cip, int transpose_flip) {
  const int64_t total = (int64_t)cip * cop;
  const float s_u = __builtin_bit_cast(float, U[0]);
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x)
    pack_f4s_elem(w, U, i, co, ci, cop, cip, transpose_flip, s_u);
}

// out[0] = max(out[0], max |x|) over n images of `per_img` contiguous elements, as BIT PATTERNS: non-negative floats order as unsigned
// integers and every NaN orders above +inf, so a non-finite element is sticky (v_max_f32 would drop a NaN) - a consumer's scale
// then falls back to a harmless one (f4s_input_scale) and the host's overflow guard sees the slot.  16-byte loads, four in flight
// per thread; contiguous batches are folded into one long image by the launcher.
__global__ __launch_bounds__(256) void amax_kernel(const float *__restrict__ x, unsigned *__restrict__ out, int n, int64_t per_img, int64_t img_stride) {
  unsigned m = 0u;
  auto ab = [](float v) { return __builtin_bit_cast(unsigned, v) & 0x7fffffffu; };
  auto ab4 = [&](const f32x4 &v) { return max(max(ab(v[0]), ab(v[1])), max(ab(v[2]), ab(v[3]))); };
  const int64_t quads = per_img >> 2;
  const int64_t step = (int64_t)gridDim.x * 256;
  for (int img = blockIdx.y; img < n; img += gridDim.y) {
    const float *p = x + (int64_t)img * img_stride;
    if ((reinterpret_cast<uintptr_t>(p) & 15) == 0) {
      const f32x4 *q = reinterpret_cast<const f32x4 *>(p);
      int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
      for (; i + 3 * step < quads; i += 4 * step) {
        const f32x4 a = q[i], b = q[i + step], c = q[i + 2 * step], e = q[i + 3 * step];
        m = max(m, max(max(ab4(a), ab4(b)), max(ab4(c), ab4(e))));
      }
      for (; i < quads; i += step) m = max(m, ab4(q[i]));
      for (int64_t t = quads * 4 + (int64_t)blockIdx.x * 256 + threadIdx.x; t < per_img; t += step) m = max(m, ab(p[t]));
    } else {
      for (int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; t < per_img; t += step) m = max(m, ab(p[t]));
    }
  }
#pragma unroll
  for (int sh = 32; sh > 0; sh >>= 1) m = max(m, (unsigned)__shfl_xor((int)m, sh));
  if ((threadIdx.x & 63) == 0 && m > 0u) atomicMax(out, m);
}

bool winograd_f4s_enabled() {
  static const bool on = []() {
    const char *e = getenv("EDVR_WINOGRAD_F4S");   // "0": never (the fp32 F(4x4) kernel instead)
    const char *w = getenv("EDVR_CONV_WINOGRAD");  // "0": always the direct kernel
    return !(e && e[0] == '0') && !(w && w[0] == '0');
  }();
  return on;
}

// the kernel itself: what winograd_f4.hip takes, with the split weights and a bound of the input's magnitude
bool winograd_f4s_supported(const edvr_conv2d_desc &d) {
  if (!d.wpk_f4s || !d.x_amax || d.ks != 3 || d.stride != 1) return false;
  const bool has_res = d.res1 || d.res2;
  if (d.gate && (has_res || d.act == EDVR_ACT_SIGMOID || d.out_mode != EDVR_OUT_NCHW)) return false;
  if ((d.res2 && !d.res1) || (d.out_mode != EDVR_OUT_NCHW && has_res)) return false;
  if (d.y_scale != 0.f && d.y_scale != 1.f && !d.res1 && !d.gate) return false;
  if (d.c2 > 0 && (d.c1 & 1)) return false;
  if ((int64_t)((d.co + 63) / 64 * 64) * ((d.c1 + d.c2 + 7) / 8 * 8) * 144 + 64 >= ((int64_t)1 << 31)) return false;
  if ((int64_t)d.h * d.w * 8 >= ((int64_t)1 << 31)) return false;
  if (d.w & 3) return false;
  auto aligned = [](const void *p, int64_t img_stride, int a) { return !p || ((reinterpret_cast<uintptr_t>(p) & (a - 1)) == 0 && (img_stride * 4 & (a - 1)) == 0); };
  if (!aligned(d.x1, d.x1_img_stride, 16) || !aligned(d.x2, d.x2_img_stride, 16)) return false;
  if (!aligned(d.y, d.y_img_stride, 16) || !aligned(d.res1, d.res1_img_stride, 16) || !aligned(d.res2, d.res2_img_stride, 16) ||
      !aligned(d.gate, d.gate_img_stride, 16) || !aligned(d.wpk_f4s, 0, 16))
    return false;
  return true;
}

bool winograd_f4s_eligible(const edvr_conv2d_desc &d) {
  if (!winograd_f4s_supported(d)) return false;
  if (d.algo == EDVR_CONV_WINOGRAD_F4S) return true;  // explicit request: any size the kernel can do
  if (d.algo != EDVR_CONV_AUTO || !winograd_f4s_enabled()) return false;
  return d.co >= 48 && d.c1 + d.c2 >= 32 && d.w >= 32 && d.h >= 8;
}

static int f4s_geometry(const edvr_conv2d_desc &d, int &tiles_x, int &tiles_y, bool &tx8) {
  const int64_t pad16 = (int64_t)cdiv(d.w, 64) * 64 * cdiv(d.h, 8) * 8, pad8 = (int64_t)cdiv(d.w, 32) * 32 * cdiv(d.h, 16) * 16;
  tx8 = pad8 < pad16;
  tiles_x = cdiv(d.w, tx8 ? 32 : 64);
  tiles_y = cdiv(d.h, tx8 ? 16 : 8);
  return tiles_x * tiles_y * cdiv(d.co, 64) * d.n;
}

// flops the f16 matrix pipe executes for `d`, padding included: per item and 8-channel chunk 144 v_mfma_f32_32x32x16_f16 of 32768 flops
double winograd_f4s_executed_flops(const edvr_conv2d_desc &d) {
  int tx, ty;
  bool tx8;
  const double items = f4s_geometry(d, tx, ty, tx8);
  return items * ((d.c1 + d.c2 + 7) / 8) * 144.0 * 32768.0;
}

int winograd_f4s_launch(const edvr_conv2d_desc &d, hipStream_t stream) {
  WinoF4SArgs a;
  a.d = d;
  a.U = reinterpret_cast<const unsigned *>(d.wpk_f4s);
  a.ci_real = d.c1 + d.c2;
  a.ci = (a.ci_real + 7) / 8 * 8;
  a.cop = (d.co + 63) / 64 * 64;
  a.ys = d.y_scale == 0.f ? 1.f : d.y_scale;
  a.ys_gs = a.ys * d.gate_slope;
  bool tx8;
  a.items = f4s_geometry(d, a.tiles_x, a.tiles_y, tx8);
  static const int n_cu = []() {
    int dev = 0, n = 256;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) n = prop.multiProcessorCount;
    return n;
  }();
  const dim3 grid(std::min(a.items, n_cu));
  if (tx8) hipLaunchKernelGGL(conv3x3_winograd_f4s_kernel<3>, grid, dim3(1024), 0, stream, a);
  else hipLaunchKernelGGL(conv3x3_winograd_f4s_kernel<4>, grid, dim3(1024), 0, stream, a);
  return check_launch("conv3x3_winograd_f4s_kernel");
}

}  // namespace edvr

extern "C" {

#ifdef F4S_PROF
int edvr_f4s_prof_read(unsigned long long *host, int reset) {  // host[16 * 8]
  int rc = (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(edvr::f4s_prof), sizeof(unsigned long long) * 128);
  if (rc == 0 && reset) {
    static unsigned long long zero[128];
    rc = (int)hipMemcpyToSymbol(HIP_SYMBOL(edvr::f4s_prof), zero, sizeof(zero));
  }
  return rc;
}
#endif

size_t edvr_conv2d_packed_weight_f4s_elems(int co, int ci) { return 16 + (size_t)((co + 63) / 64 * 64) * ((ci + 7) / 8 * 8) * 36; }

int edvr_conv2d_pack_weight_f4s_f32(const float *w, void *wpk_f4s, int co, int ci, int transpose_flip, edvr_stream_t stream) {
  EDVR_REQUIRE(w && wpk_f4s && co > 0 && ci > 0, "pack_weight_f4s: bad arguments");
  const int cop = (co + 63) / 64 * 64, cip = (ci + 7) / 8 * 8;
  const int64_t total = (int64_t)cip * cop;
  unsigned *U = reinterpret_cast<unsigned *>(wpk_f4s);
  hipLaunchKernelGGL(edvr::winograd_f4s_weight_scale_kernel, dim3(1), dim3(1024), 0, edvr::as_stream(stream), w, U, (int64_t)co * ci * 9);
  hipLaunchKernelGGL(edvr::winograd_f4s_weight_kernel, dim3((unsigned)std::min<int64_t>(edvr::cdiv64(total, 256), 4096)), dim3(256), 0,
                     edvr::as_stream(stream), w, U, co, ci, cop, cip, transpose_flip);
  return edvr::check_launch("winograd_f4s_weight_kernel");
}

int edvr_amax_f32(const float *x, float *amax, int n, int64_t per_img, int64_t img_stride, edvr_stream_t stream) {
  EDVR_REQUIRE(x && amax && n > 0 && per_img > 0, "amax: bad arguments");
  if (img_stride == per_img || n == 1) {  // one long array
    per_img *= n;
    n = 1;
  }
  const int bx = (int)std::min<int64_t>(edvr::cdiv64(per_img, 256 * 16), 2048);
  const int by = std::min(n, std::max(1, 2048 / bx));
  hipLaunchKernelGGL(edvr::amax_kernel, dim3(bx, by), dim3(256), 0, edvr::as_stream(stream), x, reinterpret_cast<unsigned *>(amax), n, per_img,
                     img_stride);
  return edvr::check_launch("amax_kernel");
}

}  // extern "C"
