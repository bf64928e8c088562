// conv1x1_s.hip - the streaming 1x1 convolution (conv1x1.hip: TSA fusion's feat_fusion, edvr_arch.py:190-244) with SPLIT fp32
// operands on the f16 matrix pipe (gfx950).
//
// Same structure as conv1x1_stream_kernel: the B operand (activations) comes straight from global memory into a rotating register
// set, the A operand (weights) from a double-buffered LDS slab, a wave owns 32 pixels x MT x 32 output channels.  What changes is
// the arithmetic (winograd_f4s.hip): every operand travels as ONE dword (f16 hi | f16 lo << 16) of x * s, s a power of two -
//   * weights: packed once per parameter version as [channel quad][co'][4 channels] dwords of w * s_W behind a 64-byte header
//     (s_W from max |w|): a lane's A operand - four channels x (hi, lo) of its output channel - is ONE ds_read_b128 of the slab;
//   * activations: lane (half, j) loads the FOUR channels 8 s + 4 half + i of k-step s for its pixel (coalesced 128-byte rows, as
//     before), splits them (two instructions each; s_X from `x_amax`, an upper bound of max |x|) -> the B operand; the second
//     MFMA of a pair takes B rotated by 16 bits ((lo, hi): the cross terms).
// Per 8 channels and wave 2 MT v_mfma_f32_32x32x16_f16 of 32 cycles where the fp32 kernel issues 4 MT v_mfma_f32_32x32x2_f32 of 64.
// 1 / (s_W s_X) leaves in the bias fma of the epilogue; `y_amax` (optional) receives max |y| for the next layer's bound.
#include <cstdlib>

#include "common.h"
#include "pack.h"

namespace edvr {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef int c1s_i32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 c1s_f16x8 __attribute__((ext_vector_type(8)));

namespace {
__device__ __forceinline__ void c1s_split4(const float (&x)[4], float s, unsigned (&o)[4]) {
  asm volatile(
      "v_fma_mixlo_f16 %0, %4, %8, 0\n\tv_fma_mixlo_f16 %1, %5, %8, 0\n\tv_fma_mixlo_f16 %2, %6, %8, 0\n\tv_fma_mixlo_f16 %3, %7, %8, 0\n\t"
      "v_fma_mixhi_f16 %0, %4, %8, -%0 op_sel_hi:[0,0,1]\n\tv_fma_mixhi_f16 %1, %5, %8, -%1 op_sel_hi:[0,0,1]\n\t"
      "v_fma_mixhi_f16 %2, %6, %8, -%2 op_sel_hi:[0,0,1]\n\tv_fma_mixhi_f16 %3, %7, %8, -%3 op_sel_hi:[0,0,1]"
      : "=&v"(o[0]), "=&v"(o[1]), "=&v"(o[2]), "=&v"(o[3])
      : "v"(x[0]), "v"(x[1]), "v"(x[2]), "v"(x[3]), "s"(s));
}
// 2^e with amax * 2^e < 2^15: amax = m 2^k, m in [1, 2) -> e = 14 - k (clamped: a zero / tiny / huge bound stays a normal number)
__device__ __forceinline__ float c1s_scale(float amax) {
  const int be = (int)((__builtin_bit_cast(unsigned, amax) >> 23) & 255u);
  return __builtin_bit_cast(float, (unsigned)min(max(127 + 14 - (be - 127), 7), 215) << 23);
}
}  // namespace

struct Conv1x1SArgs {
  edvr_conv2d_desc d;
  const unsigned *wq;          // header (16 dwords: s_W, 1 / s_W) + [channel quad cip / 4][cop][4] dwords
  int ci, cop, co_start;       // ci rounded up to the 64-channel slab
  int seg_shift;               // channels are addressed in segments of 1 << seg_shift planes (conv1x1.hip)
};

template <int MT>
__global__ __launch_bounds__(256, 2) void conv1x1_split_kernel(const Conv1x1SArgs a) {
  constexpr int CK = 64, MB = 32 * MT, DEPTH = 8, NW = CK * MB / 4 / 256;  // channels per weight slab, co per block, k-steps (x 4 loads) in flight
  constexpr int RSRC_FLAGS = 0x00020000;
  constexpr int OOB = (int)0x80000000;
  __shared__ __attribute__((aligned(16))) unsigned wsm[2][CK * MB];
  const edvr_conv2d_desc &d = a.d;
  const float s_x = c1s_scale(__builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, *d.x_amax))));
  const float inv_sw = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane((int)a.wq[1]));
  const int tid = threadIdx.x, lane = tid & 63, half = lane >> 5, j = lane & 31;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int hw = d.h * d.w;
  const int img = blockIdx.z, co_blk = a.co_start + blockIdx.y * MB;
  const int p = (blockIdx.x * 4 + wave) * 32 + j;  // this lane's pixel
  const bool p_ok = p < hw;

  auto uniform_ptr = [&](const float *ptr) {
    const uint64_t pv = reinterpret_cast<uint64_t>(ptr);
    return ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(pv >> 32)) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)pv);
  };
  const uint64_t b1 = uniform_ptr(d.x1 + (int64_t)img * d.x1_img_stride);
  uint64_t b2 = b1;
  if (d.x2) {
    const int i2 = d.x2_div > 0 ? (img / d.x2_div) * d.x2_mul + d.x2_add : img;
    b2 = uniform_ptr(d.x2 + (int64_t)i2 * d.x2_img_stride);
  }
  const int voff = p_ok ? (4 * half * hw + p) * 4 : OOB;  // channel quad `half` of the eight; out-of-range pixels read as 0
  // k-step s = channels 8 s .. 8 s + 7 of cat(x1, x2); c1 is a multiple of 8 (checked by the host), so a step never straddles the
  // inputs.  Segmented resources as in conv1x1.hip (an image may exceed the 2 GB a 32-bit buffer offset reaches).
  const int real_steps = (d.c1 + d.c2) / 8;
  const int seg_mask = (1 << a.seg_shift) - 1;
  const int64_t plane_bytes = (int64_t)hw * 4;
  auto load_b = [&](int s, float (&v)[4]) {  // branch-free: base / channel offset / validity are scalar selects
    const int c = 8 * s;
    const bool first = c < d.c1;
    const int cc = first ? c : c - d.c1;
    const uint64_t base = (first ? b1 : b2) + (uint64_t)((int64_t)(cc & ~seg_mask) * plane_bytes);
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void *>(base), (short)0, 0x7fffffff, RSRC_FLAGS);
    const bool ok = s < real_steps;
#pragma unroll
    for (int i = 0; i < 4; ++i)
      v[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, ok ? voff : OOB, ok ? ((cc & seg_mask) + i) * hw * 4 : 0, 0));
  };

  // weight slab staging: packed layout [channel quad][cop][4] -> slab [quad 16][MB][4]; thread copies NW 16-byte pieces per slab
  const int quads_total = (d.c1 + d.c2 + 3) / 4;
  c1s_i32x4 wr[NW];
  auto w_load = [&](int c0) {
#pragma unroll
    for (int i = 0; i < NW; ++i) {
      const int e = tid + i * 256, q = e / MB, col = e - q * MB;
      const bool ok = c0 / 4 + q < quads_total;
      const c1s_i32x4 v = *reinterpret_cast<const c1s_i32x4 *>(a.wq + 16 + ((int64_t)(ok ? c0 / 4 + q : 0) * a.cop + co_blk + col) * 4);
      wr[i] = ok ? v : c1s_i32x4{0, 0, 0, 0};
    }
  };
  auto w_commit = [&](int buf) {
#pragma unroll
    for (int i = 0; i < NW; ++i) *reinterpret_cast<c1s_i32x4 *>(&wsm[buf][(tid + i * 256) * 4]) = wr[i];
  };

  f32x16 acc[MT];
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[m][r] = 0.f;
  float bq[DEPTH][4];
#pragma unroll
  for (int k = 0; k < DEPTH; ++k) load_b(k, bq[k]);
  w_load(0);
  w_commit(0);
  __syncthreads();

  const int chunks = a.ci / CK;
  for (int ch = 0; ch < chunks; ++ch) {
    const int buf = ch & 1;
    const bool more = ch + 1 < chunks;
    if (more) w_load((ch + 1) * CK);
    const unsigned *ws = wsm[buf] + (half * MB + j) * 4;
#pragma unroll
    for (int s = 0; s < CK / 8; ++s) {
      c1s_i32x4 av[MT];
#pragma unroll
      for (int m = 0; m < MT; ++m) av[m] = *reinterpret_cast<const c1s_i32x4 *>(ws + ((2 * s) * MB + m * 32) * 4);
      unsigned pk[4];
      c1s_split4(bq[s % DEPTH], s_x, pk);
      load_b(ch * (CK / 8) + s + DEPTH, bq[s % DEPTH]);  // same registers, DEPTH k-steps ahead
      c1s_i32x4 b, br;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        b[q] = (int)pk[q];
        br[q] = (int)__builtin_amdgcn_alignbit(pk[q], pk[q], 16);
      }
#pragma unroll
      for (int m = 0; m < MT; ++m) acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(c1s_f16x8, av[m]), __builtin_bit_cast(c1s_f16x8, b), acc[m], 0, 0, 0);
#pragma unroll
      for (int m = 0; m < MT; ++m) acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(c1s_f16x8, av[m]), __builtin_bit_cast(c1s_f16x8, br), acc[m], 0, 0, 0);
      if (s == CK / 16 && more) w_commit(buf ^ 1);  // next slab -> the idle buffer, mid-chunk
    }
    // LDS-only barrier (the x loads in flight target registers)
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
  }

  // ---- epilogue: lane (half, j) holds pixel p and channels co_blk + m*32 + (r&3) + 8*(r>>2) + 4*half
  const float unscale = inv_sw / s_x;
  const float slope = d.act == EDVR_ACT_LRELU ? 0.1f : (d.act == EDVR_ACT_RELU ? 0.f : 1.f);
  const bool sig = d.act == EDVR_ACT_SIGMOID;
  float *y = d.y + (int64_t)img * d.y_img_stride;
  const float *q1 = d.res1 ? d.res1 + (int64_t)img * d.res1_img_stride : nullptr;
  const float *q2 = d.res2 ? d.res2 + (int64_t)img * d.res2_img_stride : nullptr;
  unsigned vmax = 0u;  // max |y| as a bit pattern: non-negative floats order as integers, NaNs above +inf (sticky for the host's overflow guard)
  if (p_ok) {
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int co = co_blk + m * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
        if (co < d.co) {
          float v = __builtin_fmaf(acc[m][r], unscale, d.bias ? d.bias[co] : 0.f);
          if (co >= d.act_from) v = sig ? __builtin_amdgcn_rcpf(1.f + __expf(-v)) : fmaxf(v, slope * v);
          const int64_t off = (int64_t)co * hw + p;
          if (q1) v += q1[off];
          if (q2) v += q2[off];
          y[off] = v;
          vmax = max(vmax, __builtin_bit_cast(unsigned, v) & 0x7fffffffu);
        }
      }
  }
  if (d.y_amax) {  // max |y| for the next layer's bound: one atomic per wave (non-negative floats order as their bit patterns)
#pragma unroll
    for (int sh = 32; sh > 0; sh >>= 1) vmax = max(vmax, (unsigned)__shfl_xor((int)vmax, sh));
    if (lane == 0) atomicMax(reinterpret_cast<unsigned *>(d.y_amax), vmax);
  }
}

// header[0] = s_W = 2^e with max|w| s_W in [2^14, 2^15), header[1] = 1 / s_W
__global__ __launch_bounds__(1024) void conv1x1_split_scale_kernel(const float *__restrict__ w, unsigned *__restrict__ wq, int64_t total) {
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
    wq[0] = field << 23;
    wq[1] = (254u - field) << 23;
    for (int i = 2; i < 16; ++i) wq[i] = 0u;
  }
}

// w (co, ci) -> [channel quad][cop][4 channels] dwords (hi | lo << 16) of w * s_W, zero beyond co / ci
__global__ void conv1x1_split_pack_kernel(const float *__restrict__ w, unsigned *__restrict__ wq, int co, int ci, int cop, int quads) {
  const float s_w = __builtin_bit_cast(float, wq[0]);
  const int64_t total = (int64_t)quads * cop;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int o = (int)(i % cop), q = (int)(i / cop);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int c = 4 * q + e;
      wq[16 + i * 4 + e] = (o < co && c < ci) ? split_f16x2(w[(int64_t)o * ci + c], s_w) : 0u;
    }
  }
}

static bool conv1x1_split_enabled() {
  static const bool on = []() {
    const char *e = getenv("EDVR_CONV1X1_SPLIT");  // "0": the fp32 streaming kernel instead
    return !(e && e[0] == '0');
  }();
  return on;
}

bool conv1x1_split_eligible(const edvr_conv2d_desc &d) {
  if (!conv1x1_split_enabled() || !d.wpk_f4s || !d.x_amax || d.algo == EDVR_CONV_DIRECT) return false;
  if (!conv1x1_eligible(d)) return false;                   // the fp32 streaming kernel's own rules (1x1, stride 1, NCHW, >= 320 channels ...)
  if ((d.c1 & 7) || (d.c2 & 7)) return false;               // a k-step of 8 channels must not straddle x1 / x2
  return (int64_t)d.h * d.w * 32 < ((int64_t)1 << 31);      // 8 channel planes inside one 32-bit buffer offset range
}

int conv1x1_split_launch(const edvr_conv2d_desc &d, hipStream_t stream) {
  EDVR_REQUIRE((reinterpret_cast<uintptr_t>(d.wpk_f4s) & 15) == 0, "conv1x1_split: packed weights must be 16-byte aligned");
  Conv1x1SArgs a;
  a.d = d;
  a.wq = reinterpret_cast<const unsigned *>(d.wpk_f4s);
  a.ci = ((d.c1 + d.c2 + 63) / 64) * 64;  // k-steps past the real channels load no x (zero B operand) and zero weights
  a.cop = (d.co + 127) / 128 * 128;        // (whole 128-channel blocks in the packed buffer: the slab copy never leaves it)
  const int hw = d.h * d.w;
  a.seg_shift = 30;
  while (((int64_t)1 << a.seg_shift) * hw * 4 >= ((int64_t)1 << 31)) --a.seg_shift;
  if (a.seg_shift < 3) a.seg_shift = 3;    // (conv1x1_split_eligible: eight planes always fit)
  const int full = d.co / 128, rem_tiles = cdiv(d.co - full * 128, 32);
  if (full > 0) {
    a.co_start = 0;
    hipLaunchKernelGGL((conv1x1_split_kernel<4>), dim3(cdiv(hw, 128), full, d.n), dim3(256), 0, stream, a);
  }
  if (rem_tiles > 0) {
    a.co_start = full * 128;
    const dim3 grid(cdiv(hw, 128), 1, d.n);
    if (rem_tiles == 1) hipLaunchKernelGGL((conv1x1_split_kernel<1>), grid, dim3(256), 0, stream, a);
    else if (rem_tiles == 2) hipLaunchKernelGGL((conv1x1_split_kernel<2>), grid, dim3(256), 0, stream, a);
    else hipLaunchKernelGGL((conv1x1_split_kernel<3>), grid, dim3(256), 0, stream, a);
  }
  return check_launch("conv1x1_split_kernel");
}

}  // namespace edvr

extern "C" {

size_t edvr_conv2d_packed_weight_1x1s_elems(int co, int ci) {
  if (co <= 0 || ci <= 0) return 0;
  return 16 + (size_t)((ci + 63) / 64 * 16) * ((co + 127) / 128 * 128) * 4;
}

int edvr_conv2d_pack_weight_1x1s_f32(const float *w, void *wpk, int co, int ci, edvr_stream_t stream_) {
  using namespace edvr;
  EDVR_REQUIRE(w && wpk && co > 0 && ci > 0, "pack_weight_1x1s: bad arguments");
  EDVR_REQUIRE((reinterpret_cast<uintptr_t>(wpk) & 15) == 0, "pack_weight_1x1s: wpk must be 16-byte aligned");
  hipStream_t stream = as_stream(stream_);
  unsigned *wq = static_cast<unsigned *>(wpk);
  const int cop = (co + 127) / 128 * 128, quads = (ci + 63) / 64 * 16;
  hipLaunchKernelGGL(conv1x1_split_scale_kernel, dim3(1), dim3(1024), 0, stream, w, wq, (int64_t)co * ci);
  const int64_t total = (int64_t)quads * cop;
  hipLaunchKernelGGL(conv1x1_split_pack_kernel, dim3((unsigned)std::min<int64_t>(cdiv64(total, 256), 2048)), dim3(256), 0, stream, w, wq, co, ci, cop, quads);
  return check_launch("conv1x1_split_pack_kernel");
}

}  // extern "C"
