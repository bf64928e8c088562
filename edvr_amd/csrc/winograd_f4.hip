// winograd_f4.hip - 3x3 / stride-1 convolution as Winograd F(4x4, 3x3) on the fp32 matrix cores (gfx950).
//
// F(2x2,3x3) (winograd.hip) issues 16 multiplies per 2x2 outputs and channel pair = 4 per output; F(4x4,3x3) issues 36 per 4x4
// = 2.25 per output: 1.78x fewer MFMAs again.  The price is arithmetic error - the transform matrices hold 4, 5, 8 where
// F(2x2) has 1 - measured 1.4e-6 of the output scale at 128 input channels in fp32 (F(2x2): 2e-7), far inside the 1e-3 dB
// PSNR bound the path is specified with; the training path uses it for forward and data-gradient convs as well (the gradient
// parity tests hold at their 1e-5 bounds), the weight gradient stays in the F(2x2) domain (winograd_wgrad.hip).
//
//   V = B^T d B      6x6 input patches (stride 4), transformed in registers by the PRODUCER waves, -> LDS
//   M[xi] = sum_ci U[xi][co, ci] * V[xi][ci, tile]      36 GEMMs on v_mfma_f32_32x32x2_f32, CONSUMER waves
//   Y = A^T M A      row pass (6 -> 4) in the consumers' registers, column pass by the producers after an LDS exchange
//   U = G g G^T      packed once per weight version in MFMA operand order (winograd_f4_pack)
//
// Work split: wave-specialised 1024-thread workgroups, one per CU, persistent over items of 64 output channels x 32 tiles
// (2 x 16 tiles = 8 x 64 output pixels, or 4 x 8 = 16 x 32 where that pads the image less: template parameter TXL):
//   * 12 consumer waves = (32-channel half wm, transform row r): 6 accumulator tiles (32 co x 32 tiles, positions (r, 0..5)) =
//     96 registers, three per SIMD.  Their loop is ds_read (B operand, shared by the two wm waves) + MFMA only; the A operands
//     (U) belong to exactly one wave each, so they come straight from global memory - one 16-byte and one 8-byte buffer load per
//     six MFMAs, packed so that a wave reads 1.5 KB contiguous - two k-steps ahead (a third set does not fit 128 registers).
//     No staging instruction sits on a wave that issues MFMAs.
//   * 4 producer waves (one per SIMD): the raw input rows of a wave's two channels arrive by LDS-DMA in a 6 KB region private
//     to the wave (requested one chunk ahead); thread = (channel of the 8-channel chunk, tile) reads its 6x6 patch from there,
//     applies B^T . B and writes the 36 positions to the double-buffered V slab (2 x 36 KB).  They also own everything that
//     touches the output:
//     after the consumers' row pass (T = M A, written to a double-buffered 2 x 24 KB exchange area in eight 8-channel phases)
//     they finish Y = A^T T, apply bias / activation / residuals / gate / PixelShuffle and store 16-byte rows.
//   One barrier per chunk (LDS only) + 8 per item.
//
// Measured: 0.62 of the F(2x2) kernel's time on the same layers, 0.59 of the fp32 MFMA peak in executed flops.
// The staging waves arrive last at every barrier (scripts/f4_trace.py, -DF4_EXP_TRACE) and the input traffic delays the
// consumers' weight loads: with the input fetch forced out of range (ablation builds, scripts/build_variant.sh) the kernel runs 22 % faster.
#include <cstdlib>
#include <type_traits>

#include "common.h"
#include "pack.h"

namespace edvr {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

// v_max_f32 as it is: fmaxf() compiles to a canonicalising v_max_f32 v, v, v in front of the real one (IEEE sNaN quieting), a third
// instruction per output element of the activation on waves whose instruction count is what the epilogue costs
__device__ __forceinline__ float max_raw(float a, float b) {
  float o;
  asm("v_max_f32 %0, %1, %2" : "=v"(o) : "v"(a), "v"(b));
  return o;
}

struct WinoF4Args {
  edvr_conv2d_desc d;
  const float *U;  // [co block 64][channel pair][row 6][co half 2][lane 64 x 4 | lane 64 x 2]
  int ci, ci_real, cop, tiles_x, tiles_y, items;  // ci: rounded up to 8 (U has all-zero rows there), ci_real = c1 + c2
  float ys, ys_gs;                                // y_scale (0 -> 1) and y_scale * gate_slope
};

#define F4_EPI_PHASES 8
#define F4_LDS_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")

#define F4_BARRIER_T() F4_LDS_BARRIER()
#define F4_PSTAMP(i)

// TXL: log2 of the tiles per block row.  4: blocks of 2 x 16 tiles = 8 x 64 output pixels; 3: 4 x 8 tiles = 16 x 32 pixels, for
// images whose width wastes much of a 64-pixel block (160 wide: 17 %, 80 wide: 37 %) and for 32-wide pyramid levels.
template <int TXL>
__global__ __launch_bounds__(1024, 1) void conv3x3_winograd_f4_kernel(const WinoF4Args a) {
  constexpr int TX = 1 << TXL, TY = 32 / TX;               // tiles per block row / rows
  constexpr int BW = 4 * TX, BH = 4 * TY;                  // output pixels of a block
  constexpr int RROWS = BH + 2, RPIECES = TX + 2;          // raw input rows / 16-byte pieces per row (columns tx0 - 4 .. tx0 + BW + 3)
  static_assert(RROWS * RPIECES == 180, "a wave's region holds 2 x 180 pieces either way");
  constexpr int CK = 8;
  constexpr int VSLAB = CK * 36 * 32;   // floats per V stage (36 KB): [channel 8][position 36][tile 32]
  constexpr int XSZ = 2 * 6 * 8 * 32 * 4;  // exchange area (2 x 24 KB): [phase parity][row 6][channel 8][tile 32][4]
  constexpr int RWAVE = 6 * 64 * 4 + 4;     // raw-input region of one producer wave: 6 DMA instructions x 64 lanes x 16 B, + one-dword shift (below)
  __shared__ __attribute__((aligned(16))) float smem[2 * VSLAB + XSZ + 4 * RWAVE + 64];  // 144 KB
  float *const Xs = smem + 2 * VSLAB;
  float *const bias_s = smem + 2 * VSLAB + XSZ + 4 * RWAVE;  // the 64 biases of the item's channel block (below)

  const edvr_conv2d_desc &d = a.d;
  const int tid = threadIdx.x, lane = tid & 63, half = lane >> 5, j = lane & 31;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int hw = d.h * d.w, plane_bytes = hw * 4;
  const int co_blocks = (d.co + 63) / 64;
  const int n_chunks = a.ci / CK;
  constexpr int RSRC_FLAGS = 0x00020000;  // raw buffer, 32-bit data format (gfx9 family)
  auto uniform_rsrc = [&](const float *p, int bytes) {
    const uint64_t pv = reinterpret_cast<uint64_t>(p);
    const uint64_t pu = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(pv >> 32)) << 32) |
                        (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)pv);
    return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void *>(pu), (short)0, bytes, RSRC_FLAGS);
  };

  // Persistent workgroups, XCD-aware walk (winograd.hip): every XCD gets one contiguous range of items.
  const int n_xcd = gridDim.x < 8 ? 1 : 8;
  const int xcd = n_xcd == 1 ? 0 : (int)blockIdx.x % 8, xcd_rank = n_xcd == 1 ? (int)blockIdx.x : (int)blockIdx.x / 8;
  const int xcd_wgs = n_xcd == 1 ? (int)gridDim.x : ((int)gridDim.x - xcd + 7) / 8;
  const int span = (a.items + n_xcd - 1) / n_xcd;
  const int item_end = min(a.items, (xcd + 1) * span);
  const int item_first = xcd * span + xcd_rank;
  if (item_first >= item_end) return;
  auto decode = [&](int item, int &co_blk, int &img, int &ty0, int &tx0) {
    co_blk = __builtin_amdgcn_readfirstlane((item % co_blocks) * 64);
    const int tile_blk = __builtin_amdgcn_readfirstlane((item / co_blocks) % (a.tiles_x * a.tiles_y));
    img = __builtin_amdgcn_readfirstlane(item / (co_blocks * a.tiles_x * a.tiles_y));
    ty0 = __builtin_amdgcn_readfirstlane((tile_blk / a.tiles_x) * BH);  // output-pixel origin of the block
    tx0 = __builtin_amdgcn_readfirstlane((tile_blk % a.tiles_x) * BW);
  };

  if (wave < 4) {
    // =========================================================================================== producers
    // Next to three waves that issue MFMAs back to back this wave advances slowly (~160 instructions in ~6000 cycles per chunk)
    // and is the last at every barrier.  The wave priority makes no measurable difference in either direction (raised here,
    // raised on the consumers instead, default everywhere: same time); kept raised.
    __builtin_amdgcn_s_setprio(3);
    const int p_ty = j >> TXL, p_tx = j & (TX - 1);  // tile j of the TY x TX; channel 2 wave + half of the chunk
    // Raw input of this wave's two channels for one chunk: [channel 2][row BH + 2][(TX + 2) x 16 B] = image rows ty0 - 1 .. ty0 + BH,
    // columns tx0 - 4 .. tx0 + BW + 3 (10 x 18 or 18 x 10 pieces), fetched by LDS-DMA (buffer_load_dwordx4 ... lds: lane l of instruction i delivers 16-byte piece
    // 64 i + l; pieces outside the image get an out-of-range offset and arrive as zeros = the padding).  The region is PRIVATE
    // to the wave - it alone reads the patches of these two channels - so no cross-wave ordering is needed: read the patches of
    // chunk k + 1, then request chunk k + 2 into the same place; it has the rest of the step to arrive.
    // (First version: every thread gathered its 6x6 patch with 24 buffer loads - 16 quad accesses per instruction in the
    // texture addresser, 4160 L1 accesses per chunk and CU, address path 50 % busy: 1.24 ms where the MFMAs alone need 0.65.)
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
    // The input transform runs on PACKED fp32 math (v_pk_fma_f32 / v_pk_add_f32: two lanes of work per instruction): vector
    // instructions of the staging waves take issue slots from the MFMAs of the same SIMD, so their count is what matters
    // (72 packed operations per patch and chunk; the scalar form had 144 + 57 moves).
    //   B^T d    along the rows, on pairs of adjacent columns (2 cp, 2 cp + 1): the 12-operation scheme on both at once;
    //   (.) B    along the columns of one row held as the same pairs P0 = (d0, d1), P1 = (d2, d3), P2 = (d4, d5): the outputs
    //            pair up so that every operand is one register pair with a lo/hi selection (op_sel) - no repacking:
    //            (t0, t5) = 4 P0 - 5 P1 + P2;  (p, r) = d4 + (-4, -1) d2;  (q, s) = d3 + (-4, -1) d1;
    //            (t1, t3) = (p, r) + (1, 2) (q, s);  (t2, t4) = (p, r) - (1, 2) (q, s).
    f32x2 pp[6][3];  // raw patch: row r, columns (2 cp, 2 cp + 1)
    f32x2 tp[6][3];  // B^T d, same pairing
    // The DMA writes the pieces ONE DWORD into the region: column tx0 - 4 + k of a row lands at dword k + 1, so the patch of tile
    // p_tx (columns 4 p_tx + 3 .. + 8) starts at the 16-byte aligned dword 4 p_tx + 4: one b128 + one b64 read per row, already
    // paired the way the packed transform wants them (no moves).
    const float *patch = Rw + ((half * RROWS + 4 * p_ty) * RPIECES + p_tx) * 4 + 4;  // patch row r, column c: patch[r * 4 RPIECES + c]
    auto read_patch = [&]() {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the requested chunk is in the region
      F4_PSTAMP(0);
#pragma unroll
      for (int r = 0; r < 6; ++r) {
        const f32x4 m = *reinterpret_cast<const f32x4 *>(patch + r * (4 * RPIECES));
        pp[r][0] = f32x2{m[0], m[1]};
        pp[r][1] = f32x2{m[2], m[3]};
        pp[r][2] = *reinterpret_cast<const f32x2 *>(patch + r * (4 * RPIECES) + 4);
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // ... and in registers: the region may be overwritten
      F4_PSTAMP(1);
    };
    auto transform_cols = [&](int cp) {  // 1-D input transform B^T (Lavin & Gray), 12 operations (a * b + c contracts to an fma)
      const f32x2 d0 = pp[0][cp], d1 = pp[1][cp], d2 = pp[2][cp], d3 = pp[3][cp], d4 = pp[4][cp], d5 = pp[5][cp];
      const f32x2 p_ = d4 - 4.f * d2, q_ = d3 - 4.f * d1, r_ = d4 - d2, s_ = d3 - d1;
      tp[0][cp] = 4.f * d0 + (d4 - 5.f * d2);
      tp[1][cp] = p_ + q_;
      tp[2][cp] = p_ - q_;
      tp[3][cp] = r_ + 2.f * s_;
      tp[4][cp] = r_ - 2.f * s_;
      tp[5][cp] = 4.f * d1 + (d5 - 5.f * d3);
    };
    auto commit_row = [&](float *Vd, int r) {  // positions (r, 0..5) of (B^T d) B
      const f32x2 P0 = tp[r][0], P1 = tp[r][1], P2 = tp[r][2];
      const f32x2 lo1 = __builtin_shufflevector(P1, P1, 0, 0), hi1 = __builtin_shufflevector(P1, P1, 1, 1);  // d2, d3
      const f32x2 lo2 = __builtin_shufflevector(P2, P2, 0, 0), hi0 = __builtin_shufflevector(P0, P0, 1, 1);  // d4, d1
      const f32x2 t05 = 4.f * P0 + (P2 - 5.f * P1);
      const f32x2 pr_ = lo2 + f32x2{-4.f, -1.f} * lo1;
      const f32x2 qs_ = hi1 + f32x2{-4.f, -1.f} * hi0;
      const f32x2 t13 = pr_ + f32x2{1.f, 2.f} * qs_;
      const f32x2 t24 = pr_ - f32x2{1.f, 2.f} * qs_;
      float *dst = Vd + ((2 * wave + half) * 36 + r * 6) * 32 + j;
      dst[0 * 32] = t05[0];
      dst[1 * 32] = t13[0];
      dst[2 * 32] = t24[0];
      dst[3 * 32] = t13[1];
      dst[4 * 32] = t24[1];
      dst[5 * 32] = t05[1];
    };
    auto advance = [&]() {  // the load cursor moves one chunk; at an item boundary the geometry switches
      if (++l_k == n_chunks) {
        l_k = 0;
        const int nx = l_item + xcd_wgs;
        l_item = nx < item_end ? nx : l_item;  // past the last item: re-stage it (never consumed)
        setup(l_item);
      }
    };

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
    for (int r = 0; r < 6; ++r) commit_row(smem, r);
    F4_BARRIER_T();

    int par = 0;  // stage the consumers read during the current step
    // abs_sum epilogue (edvr_conv2d_desc.abs_sum): every staging thread sums |y| of the elements it stores, over the items of one
    // image; when the walk enters another image (a workgroup's items cover one or two) and at the end, the wave folds its 64 sums
    // and adds one value to abs_sum[image] - a few thousand atomics per launch.
    float asum = 0.f;
    int asum_img = -1;
    auto asum_flush = [&]() {
      float s = asum;
#pragma unroll
      for (int sh = 32; sh > 0; sh >>= 1) s += __shfl_xor(s, sh);
      if (lane == 0 && asum_img >= 0 && s != 0.f) atomicAdd(d.abs_sum + asum_img, s);
      asum = 0.f;
    };
    for (int item = item_first; item < item_end; item += xcd_wgs) {
      int e_co_blk, e_img, e_ty0, e_tx0;
      decode(item, e_co_blk, e_img, e_ty0, e_tx0);
      if (d.abs_sum && e_img != asum_img) {
        asum_flush();
        asum_img = e_img;
      }
#pragma unroll 1
      for (int k = 0; k < n_chunks; ++k) {
        // the region holds chunk k + 1 (chunk 0 of the next item at the end): patches -> registers, request chunk k + 2,
        // transform into the idle stage
        float *Vd = smem + (par ^ 1) * VSLAB;
        read_patch();
        advance();
        load_begin(l_k * CK);
        dma_issue();
        F4_PSTAMP(2);
#pragma unroll
        for (int cp = 0; cp < 3; ++cp) transform_cols(cp);
#pragma unroll
        for (int r = 0; r < 6; ++r) commit_row(Vd, r);
        F4_PSTAMP(3);
        F4_BARRIER_T();
        par ^= 1;
      }

      // ---- column pass Y = A^T T + epilogue + stores: 8 phases of 8 output channels, one (channel, tile) per thread and phase.
      //      The exchange area is double-buffered: the consumers write phase p + 1 while phase p is read here (one barrier per
      //      phase).  Bias and residual / gate rows of phase p + 1 are requested as soon as those of phase p are consumed.
      const int plane = hw;
      float *y = d.y + (int64_t)e_img * d.y_img_stride;
      const float *r1 = d.res1 ? d.res1 + (int64_t)e_img * d.res1_img_stride : nullptr;
      const float *r2 = d.res2 ? d.res2 + (int64_t)e_img * d.res2_img_stride : nullptr;
      const float *gt = d.gate ? d.gate + (int64_t)e_img * d.gate_img_stride : nullptr;
      const float *rq = gt ? gt : r1;  // the tensor read per output element (gate and residuals exclude each other)
      const float slope = d.act == EDVR_ACT_LRELU ? 0.1f : (d.act == EDVR_ACT_RELU ? 0.f : 1.f);  // none/relu/lrelu = max(v, slope*v)
      const bool sig = d.act == EDVR_ACT_SIGMOID, shuffle = d.out_mode == EDVR_OUT_PIXEL_SHUFFLE2;
      const bool vec = e_tx0 + BW <= d.w;  // the block is inside the image in x (w % 4 == 0): 16-byte rows, only the ROW is tested
      const int cl8 = (wave * 64 + lane) >> 5, tile = lane & 31;  // (thread index within the four staging waves)
      const int oy = e_ty0 + 4 * (tile >> TXL), ox = e_tx0 + 4 * (tile & (TX - 1));
      const int co_t = e_co_blk + (cl8 >> 2) * 32 + 4 * ((cl8 >> 1) & 1) + 8 * (cl8 & 1);  // + (p & 3) + 16 (p >> 2) in phase p
      const int pix = oy * d.w + ox;
      const int rows_in = d.h - oy;  // rows of this lane's tile inside the image (>= 4: all of them)
      // The biases of the item's 64 channels go through LDS: a vector-memory load inside the phase loop would have to be waited
      // for with vmcnt(0) - the counter is in issue order and counts stores - i.e. every phase would also wait for the store
      // acknowledgements of its predecessor.  Layers without residual / gate then have no vector-memory load in the loop at all.
      if (wave == 0) bias_s[lane] = (d.bias && e_co_blk + lane < d.co) ? d.bias[e_co_blk + lane] : 0.f;  // read after the first phase barrier
      auto column_pass = [&](auto VEC, auto SHUF) {
        constexpr bool V = decltype(VEC)::value;     // whole 16-byte rows inside the image in x: no per-element tests
        constexpr bool SHF = decltype(SHUF)::value;  // V && PixelShuffle(2): channels 2 q, 2 q + 1 (consecutive phases of this thread)
                                                     // interleave along x - two 16-byte stores per row and channel pair
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
        for (int p = 0; p < (F4_EPI_PHASES); ++p) {
          F4_BARRIER_T();  // T of this phase is in its half of the exchange area
          const float *Xb = Xs + (p & 1) * (XSZ / 2) + (cl8 * 32 + tile) * 4;
          f32x4 T[6];
#pragma unroll
          for (int r = 0; r < 6; ++r) T[r] = *reinterpret_cast<const f32x4 *>(Xb + r * (8 * 32 * 4));
          const int co = co_of(p);
          const float b = bias_s[co - e_co_blk];
          const float sl = co >= d.act_from ? slope : 1.f;
          f32x4 Y[4];
#pragma unroll
          for (int jj = 0; jj < 4; ++jj) {  // A^T along the rows (6 -> 4), + bias
            const float s1 = T[1][jj] + T[2][jj], d1 = T[1][jj] - T[2][jj], s2 = T[3][jj] + T[4][jj], d2 = T[3][jj] - T[4][jj];
            Y[0][jj] = T[0][jj] + s1 + s2 + b;
            Y[1][jj] = __builtin_fmaf(2.f, d2, d1) + b;
            Y[2][jj] = __builtin_fmaf(4.f, s2, s1) + b;
            Y[3][jj] = __builtin_fmaf(8.f, d2, d1) + T[5][jj] + b;
          }
          if (sig) {
            if (co >= d.act_from) {
#pragma unroll
              for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) Y[i][jj] = __builtin_amdgcn_rcpf(1.f + __expf(-Y[i][jj]));
            }
          } else {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
              for (int jj = 0; jj < 4; ++jj) Y[i][jj] = max_raw(Y[i][jj], sl * Y[i][jj]);
          }
          // The bias / residual / gate rows of the NEXT phase are requested BEFORE this phase's stores: the memory counter is
          // in issue order, so their use then only waits for operations issued up to them (vmcnt(4): the four stores below stay
          // in flight) - requested after the stores, every phase waited for its predecessor's store acknowledgements.
          if (SHF) {
            prefetch(min(p + 1, 7));
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
            if (d.abs_sum && co < d.abs_sum_channels) {  // (channels below act_from: Y is the conv output + bias here)
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
    }
    if (d.abs_sum) asum_flush();
  } else {
    // =========================================================================================== consumers
    const int q = wave - 4, wm = q & 1, row = q >> 1;
    const int np = a.ci >> 1;
    const __amdgpu_buffer_rsrc_t u_rsrc = uniform_rsrc(a.U, a.cop * a.ci * 36 * 4);
    const int voff4 = lane * 16, voff2 = 1024 + lane * 8;
    f32x16 acc[6];
    f32x4 a4[2];
    f32x2 a2[2];
    const int n_steps = 4 * n_chunks;  // k-steps (channel pairs) of an item
    int u_base = 0;  // byte offset of (co block, channel pair 0, row, wm); a k-step further on is 12 blocks of 1536 bytes
    auto set_item = [&](int co_blk) { u_base = (((co_blk >> 6) * np * 6 + row) * 2 + wm) * 1536; };
    auto load_a = [&](int set, int t) {  // A operands of k-step t (channel pair t): positions (row, 0..3) and (row, 4..5)
      const int soff = u_base + t * (12 * 1536);
      a4[set] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(u_rsrc, voff4, soff, 0));
      a2[set] = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(u_rsrc, voff2, soff, 0));
    };
    auto load_a4 = [&](int set, int t) {
      a4[set] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(u_rsrc, voff4, u_base + t * (12 * 1536), 0));
    };
    auto load_a2 = [&](int set, int t) {
      a2[set] = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(u_rsrc, voff2, u_base + t * (12 * 1536), 0));
    };
    int co_blk, img_, ty_, tx_;
    decode(item_first, co_blk, img_, ty_, tx_);
    set_item(co_blk);
    load_a(0, 0);
    load_a(1, 1);
#pragma unroll
    for (int c = 0; c < 6; ++c)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;
    F4_BARRIER_T();

    int par = 0;
    const int b_lane = half * 36 * 32 + row * 6 * 32 + j;  // B operand: channel 2 cp + half, position (row, c), tile j
    for (int item = item_first; item < item_end; item += xcd_wgs) {
#pragma unroll 1
      for (int k = 0; k < n_chunks; ++k) {
        const float *Vs = smem + par * VSLAB + b_lane;
        // 8 groups of 3 MFMAs (k-step cp = channel pair, positions (row, 0..2) / (row, 3..5)); the B operands of group g + 1 are
        // fetched before the MFMAs of group g, the A registers of a k-step are re-requested (two k-steps ahead) right after
        // its last MFMA.  The schedule is pinned: left free, hipcc sinks every load to just before its first use.
        float bv[2][3];
#pragma unroll
        for (int c = 0; c < 3; ++c) bv[0][c] = Vs[c * 32];
#pragma unroll
        for (int g = 0; g < 8; ++g) {
          const int cp = g >> 1, hi = g & 1, set = cp & 1, cur = g & 1, nxt = cur ^ 1;
          if (g + 1 < 8) {
            const int cpn = (g + 1) >> 1, hn = (g + 1) & 1;
#pragma unroll
            for (int c = 0; c < 3; ++c) bv[nxt][c] = Vs[(2 * cpn * 36 + 3 * hn + c) * 32];
          }
          if (!hi) {
#pragma unroll
            for (int c = 0; c < 3; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[set][c], bv[cur][c], acc[c], 0, 0, 0);
          } else {
            acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[set][3], bv[cur][0], acc[3], 0, 0, 0);
            const int t_next = min(4 * k + cp + 2, n_steps - 1);  // past the end of the item: a redundant reload of its last k-step
            load_a4(set, t_next);  // the 16-byte part is free two MFMAs before the 8-byte part: requested that much earlier
            acc[4] = __builtin_amdgcn_mfma_f32_32x32x2f32(a2[set][0], bv[cur][1], acc[4], 0, 0, 0);
            acc[5] = __builtin_amdgcn_mfma_f32_32x32x2f32(a2[set][1], bv[cur][2], acc[5], 0, 0, 0);
            load_a2(set, t_next);
          }
          __builtin_amdgcn_sched_barrier(0);
        }
        F4_BARRIER_T();
        par ^= 1;
      }
      {  // A operands of the next item's first two k-steps: in flight during the row pass
        const int nx = item + xcd_wgs;
        decode(nx < item_end ? nx : item, co_blk, img_, ty_, tx_);
        set_item(co_blk);
        load_a(0, 0);
        load_a(1, 1);
      }
      // ---- row pass T = M A (6 -> 4) and hand-over to the producers: 8 phases of two accumulator registers (8 output channels
      //      over the two channel halves), alternating halves of the exchange area
#pragma unroll
      for (int p = 0; p < (F4_EPI_PHASES); ++p) {
        float *Xb = Xs + (p & 1) * (XSZ / 2);
#pragma unroll
        for (int rr = 0; rr < 2; ++rr) {
          const int r = (p & 3) + 4 * (2 * (p >> 2) + rr);  // consecutive phases of a staging thread are consecutive channels
          const float m0 = acc[0][r], m1 = acc[1][r], m2 = acc[2][r], m3 = acc[3][r], m4 = acc[4][r], m5 = acc[5][r];
          const float s1 = m1 + m2, d1 = m1 - m2, s2 = m3 + m4, d2 = m3 - m4;
          f32x4 T;
          T[0] = m0 + s1 + s2;
          T[1] = __builtin_fmaf(2.f, d2, d1);
          T[2] = __builtin_fmaf(4.f, s2, s1);
          T[3] = __builtin_fmaf(8.f, d2, d1) + m5;
          const int cl8 = wm * 4 + half * 2 + rr;
          *reinterpret_cast<f32x4 *>(Xb + ((row * 8 + cl8) * 32 + j) * 4) = T;
        }
        F4_BARRIER_T();  // phase p written (and phase p - 1 read: its half may be overwritten next)
      }
#pragma unroll
      for (int c = 0; c < 6; ++c)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;
    }
  }
}

// U[xi] = (G g G^T)[xi] of the 3x3 kernel g = w[co][ci] (data-gradient kernel when transpose_flip) in the operand order above.
__global__ void winograd_f4_weight_kernel(const float *__restrict__ w, float *__restrict__ U, int co, int ci, int cop, int cip,
                                          int transpose_flip) {
  const int64_t total = (int64_t)cip * cop;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x)
    pack_f4_elem(w, U, i, co, ci, cop, cip, transpose_flip);  // pack.h
}

bool winograd_f4_enabled() {
  static const bool on = []() {
    const char *e = getenv("EDVR_WINOGRAD_F4");    // "0": never (F(2x2) / direct everywhere)
    const char *w = getenv("EDVR_CONV_WINOGRAD");  // "0": always the direct kernel (winograd.hip) - this kernel follows it
    return !(e && e[0] == '0') && !(w && w[0] == '0');
  }();
  return on;
}

// The kernel itself: any 3x3 / stride-1 conv whose epilogue the F(2x2) kernel also takes, with a packed-F4 weight buffer.
bool winograd_f4_supported(const edvr_conv2d_desc &d) {
  if (!d.wpk_f4 || d.ks != 3 || d.stride != 1) return false;
  const bool has_res = d.res1 || d.res2;
  if (d.gate && (has_res || d.act == EDVR_ACT_SIGMOID || d.out_mode != EDVR_OUT_NCHW)) return false;
  if ((d.res2 && !d.res1) || (d.out_mode != EDVR_OUT_NCHW && has_res)) return false;
  if (d.y_scale != 0.f && d.y_scale != 1.f && !d.res1 && !d.gate) return false;  // the scale lives in the residual / gate epilogues
  if (d.c2 > 0 && (d.c1 & 1)) return false;                                      // a staging wave covers two consecutive channels
  if ((int64_t)((d.co + 63) / 64 * 64) * ((d.c1 + d.c2 + 7) / 8 * 8) * 144 >= ((int64_t)1 << 31)) return false;  // the packed weights: 32-bit offsets
  if ((int64_t)d.h * d.w * 8 >= ((int64_t)1 << 31)) return false;                // two planes in 32-bit buffer offsets
  if (d.w & 3) return false;                                                     // input rows are fetched as aligned 16-byte pieces
  auto aligned = [](const float *p, int64_t img_stride, int a) { return !p || ((reinterpret_cast<uintptr_t>(p) & (a - 1)) == 0 && (img_stride * 4 & (a - 1)) == 0); };
  if (!aligned(d.x1, d.x1_img_stride, 16) || !aligned(d.x2, d.x2_img_stride, 16)) return false;
  if (!aligned(d.y, d.y_img_stride, 16) || !aligned(d.res1, d.res1_img_stride, 16) || !aligned(d.res2, d.res2_img_stride, 16) ||
      !aligned(d.gate, d.gate_img_stride, 16))
    return false;
  return true;
}

bool winograd_f4_eligible(const edvr_conv2d_desc &d) {
  if (!winograd_f4_supported(d)) return false;
  if (d.algo == EDVR_CONV_WINOGRAD_F4 || d.algo == EDVR_CONV_WINOGRAD_F4S) return true;  // explicit request (F4S falls back here): any size the kernel can do
  if (d.algo != EDVR_CONV_AUTO || !winograd_f4_enabled()) return false;
  return d.co >= 48 && d.c1 + d.c2 >= 32 && d.w >= 32 && d.h >= 8;  // auto: only where it beats F(2x2)
}

// block shape: the one that pads the image less (8 x 64 output pixels on ties); returns the number of items (64 co x 32 tiles)
static int f4_geometry(const edvr_conv2d_desc &d, int &tiles_x, int &tiles_y, bool &tx8) {
  const int64_t pad16 = (int64_t)cdiv(d.w, 64) * 64 * cdiv(d.h, 8) * 8, pad8 = (int64_t)cdiv(d.w, 32) * 32 * cdiv(d.h, 16) * 16;
  tx8 = pad8 < pad16;
  tiles_x = cdiv(d.w, tx8 ? 32 : 64);
  tiles_y = cdiv(d.h, tx8 ? 16 : 8);
  return tiles_x * tiles_y * cdiv(d.co, 64) * d.n;
}

// flops the matrix cores execute for `d`, padding included: per item and 8-channel chunk 288 v_mfma_f32_32x32x2_f32 of 4096 flops
double winograd_f4_executed_flops(const edvr_conv2d_desc &d) {
  int tx, ty;
  bool tx8;
  const double items = f4_geometry(d, tx, ty, tx8);
  return items * ((d.c1 + d.c2 + 7) / 8) * 288.0 * 4096.0;
}

int winograd_f4_launch(const edvr_conv2d_desc &d, hipStream_t stream) {
  WinoF4Args a;
  a.d = d;
  a.U = d.wpk_f4;
  a.ci_real = d.c1 + d.c2;
  a.ci = (a.ci_real + 7) / 8 * 8;
  a.cop = (d.co + 63) / 64 * 64;
  a.ys = d.y_scale == 0.f ? 1.f : d.y_scale;
  a.ys_gs = a.ys * d.gate_slope;
  bool tx8;
  a.items = f4_geometry(d, a.tiles_x, a.tiles_y, tx8);
  static const int n_cu = []() {
    int dev = 0, n = 256;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) n = prop.multiProcessorCount;
    return n;
  }();
  const dim3 grid(std::min(a.items, n_cu));
  if (tx8) hipLaunchKernelGGL(conv3x3_winograd_f4_kernel<3>, grid, dim3(1024), 0, stream, a);
  else hipLaunchKernelGGL(conv3x3_winograd_f4_kernel<4>, grid, dim3(1024), 0, stream, a);
  return check_launch("conv3x3_winograd_f4_kernel");
}

int winograd_f4_pack(const float *w, float *U, int co, int ci, int transpose_flip, hipStream_t stream) {
  const int cop = (co + 63) / 64 * 64, cip = (ci + 7) / 8 * 8;
  const int64_t total = (int64_t)cip * cop;
  hipLaunchKernelGGL(winograd_f4_weight_kernel, dim3((unsigned)std::min<int64_t>(cdiv64(total, 256), 4096)), dim3(256), 0, stream, w, U, co, ci,
                     cop, cip, transpose_flip);
  return check_launch("winograd_f4_weight_kernel");
}

}  // namespace edvr

extern "C" {


size_t edvr_conv2d_packed_weight_f4_elems(int co, int ci) { return (size_t)((co + 63) / 64 * 64) * ((ci + 7) / 8 * 8) * 36; }

int edvr_conv2d_pack_weight_f4_f32(const float *w, float *wpk_f4, int co, int ci, int transpose_flip, edvr_stream_t stream) {
  EDVR_REQUIRE(w && wpk_f4 && co > 0 && ci > 0, "pack_weight_f4: bad arguments");
  return edvr::winograd_f4_pack(w, wpk_f4, co, ci, transpose_flip, edvr::as_stream(stream));
}

}  // extern "C"
