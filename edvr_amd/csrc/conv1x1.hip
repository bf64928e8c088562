// conv1x1.hip - 1x1 convolution (TSA fusion's feat_fusion / spatial attention convs, edvr_arch.py:190-244) as a streaming GEMM
// on the fp32 matrix cores (gfx950).
//
// A 1x1 conv has no spatial reuse, so staging the input through LDS (what conv2d.hip does for every kernel size) only adds a
// store + a load per element: the 1x1 instance of that kernel ran at 41-50 TF/s.  Here the B operand of v_mfma_f32_32x32x2_f32
// comes STRAIGHT from global memory - lane (half, j) needs x[channel 2s + half][pixel p0 + j], i.e. two coalesced 128-byte rows
// per load, addressed as buffer_load(resource = image, voffset = per-lane constant, soffset = channel offset in an SGPR): no
// address arithmetic at all - into a rotating set of DEPTH registers that keeps DEPTH k-steps of loads in flight.  Each loaded
// value feeds 4 MFMAs (the 4 x 32 output channels a wave owns), whose A operands (weights) come from a double-buffered LDS slab.
// One LDS read per MFMA instead of two, no LDS writes for activations.
#include <cstdlib>

#include "common.h"

namespace edvr {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

struct Conv1x1Args {
  edvr_conv2d_desc d;
  int ci, cip, cop, co_start;  // ci rounded up to the 64-channel slab, cip = rows of the packed weight buffer
  int seg_shift;               // channels are addressed in segments of 1 << seg_shift planes (below)
};

template <int MT>
__global__ __launch_bounds__(256, 2) void conv1x1_stream_kernel(const Conv1x1Args a) {
  constexpr int CK = 64, MB = 32 * MT, DEPTH = 16, NW = CK * MB / 4 / 256;  // channels per weight slab, co per block, loads in flight (8: 64 TF/s, 16: 75-101, 32: 73-99)
  constexpr int RSRC_FLAGS = 0x00020000;
  constexpr int OOB = (int)0x80000000;
  __shared__ __attribute__((aligned(16))) float wsm[2][CK * MB];
  const edvr_conv2d_desc &d = a.d;
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
  const int voff = p_ok ? (half * hw + p) * 4 : OOB;  // channel `half` of the pair; out-of-range pixels read as 0
  // k-step s = channel pair (2s, 2s + 1) of cat(x1, x2); c1 is even (checked by the host), so a pair never straddles the inputs.
  // Buffer offsets are 32 bits with the range check at 2^31 (OOB above), an image of 640 channels x 720 x 1280 is 2.4 GB: the
  // resource is re-based per SEGMENT of (1 << seg_shift) channel planes (the host picks the largest segment that fits; one segment
  // for every image below 2 GB), the channel inside the segment goes to the scalar offset.  All of it is scalar arithmetic on
  // the unrolled k-step index - a handful of SALU instructions next to 4 MFMAs.
  const int real_steps = (d.c1 + d.c2) / 2;
  const int seg_mask = (1 << a.seg_shift) - 1;
  const int64_t plane_bytes = (int64_t)hw * 4;
  auto load_b = [&](int s) -> float {  // branch-free: base / channel offset / validity are scalar selects
    const int c = 2 * s;
    const bool first = c < d.c1;
    const int cc = first ? c : c - d.c1;
    const uint64_t base = (first ? b1 : b2) + (uint64_t)((int64_t)(cc & ~seg_mask) * plane_bytes);
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void *>(base), (short)0, 0x7fffffff, RSRC_FLAGS);
    const int soff = (cc & seg_mask) * hw * 4;
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, s < real_steps ? voff : OOB, s < real_steps ? soff : 0, 0));
  };

  // weight slab staging: packed 1x1 layout [ci_pad32][cop] is already [channel][co]; thread copies NW float4 per slab
  f32x4 wr[NW];
  auto w_load = [&](int c0) {
#pragma unroll
    for (int i = 0; i < NW; ++i) {
      const int e = (tid + i * 256) * 4, row = e / MB, col = e - row * MB;
      const bool ok = c0 + row < a.cip;  // the packed buffer has round_up(ci, 32) rows; a 64-row slab may reach past them
      const f32x4 v = *reinterpret_cast<const f32x4 *>(d.wpk + (int64_t)(ok ? c0 + row : 0) * a.cop + co_blk + col);
      wr[i] = ok ? v : f32x4{0.f, 0.f, 0.f, 0.f};
    }
  };
  auto w_commit = [&](int buf) {
#pragma unroll
    for (int i = 0; i < NW; ++i) *reinterpret_cast<f32x4 *>(&wsm[buf][(tid + i * 256) * 4]) = wr[i];
  };

  f32x16 acc[MT];
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[m][r] = 0.f;
  float bq[DEPTH];
#pragma unroll
  for (int k = 0; k < DEPTH; ++k) bq[k] = load_b(k);
  w_load(0);
  w_commit(0);
  __syncthreads();

  const int chunks = a.ci / CK;
  for (int ch = 0; ch < chunks; ++ch) {
    const int buf = ch & 1;
    const bool more = ch + 1 < chunks;
    if (more) w_load((ch + 1) * CK);
    const float *ws = wsm[buf] + half * MB + j;
#pragma unroll
    for (int s = 0; s < CK / 2; ++s) {
      float av[MT];
#pragma unroll
      for (int m = 0; m < MT; ++m) av[m] = ws[(2 * s) * MB + m * 32];
      const float b = bq[s % DEPTH];
#pragma unroll
      for (int m = 0; m < MT; ++m) acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[m], b, acc[m], 0, 0, 0);
      bq[s % DEPTH] = load_b(ch * (CK / 2) + s + DEPTH);  // same register, DEPTH k-steps ahead
      if (s == CK / 4 && more) w_commit(buf ^ 1);         // next slab -> the idle buffer, mid-chunk
    }
    // LDS-only barrier (the x loads in flight target registers)
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
  }

  // ---- epilogue: lane (half, j) holds pixel p and channels co_blk + m*32 + (r&3) + 8*(r>>2) + 4*half
  if (!p_ok) return;
  const float slope = d.act == EDVR_ACT_LRELU ? 0.1f : (d.act == EDVR_ACT_RELU ? 0.f : 1.f);
  const bool sig = d.act == EDVR_ACT_SIGMOID;
  float *y = d.y + (int64_t)img * d.y_img_stride;
  const float *q1 = d.res1 ? d.res1 + (int64_t)img * d.res1_img_stride : nullptr;
  const float *q2 = d.res2 ? d.res2 + (int64_t)img * d.res2_img_stride : nullptr;
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int co = co_blk + m * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
      if (co < d.co) {
        float v = acc[m][r] + (d.bias ? d.bias[co] : 0.f);
        if (co >= d.act_from) v = sig ? __builtin_amdgcn_rcpf(1.f + __expf(-v)) : fmaxf(v, slope * v);
        const int64_t off = (int64_t)co * hw + p;
        if (q1) v += q1[off];
        if (q2) v += q2[off];
        y[off] = v;
      }
    }
}

bool conv1x1_eligible(const edvr_conv2d_desc &d) {
  static const bool enabled = []() {
    const char *e = getenv("EDVR_CONV1X1_STREAM");  // "0": the LDS-staged direct kernel (A/B)
    return !(e && e[0] == '0');
  }();
  if (!enabled || d.ks != 1 || d.stride != 1 || d.out_mode != EDVR_OUT_NCHW || d.algo == EDVR_CONV_DIRECT) return false;
  if ((d.c1 & 1) || (d.c2 & 1)) return false;                                      // channel pairs must not straddle x1 / x2
  // 32-bit buffer offsets inside a channel segment (conv1x1_launch): a channel PAIR must stay below 2 GB (h * w < 2^28); a deep K:
  // at 128 input channels the one-sub-tile direct kernel (3 waves per SIMD) is faster - 82.6 vs 71.7 TF/s on the 128 -> 1152
  // `dcol` product of the DCN backward - while this kernel wins from 640 channels up (84 vs 71, 104 vs 82)
  return (int64_t)d.h * d.w * 8 < ((int64_t)1 << 31) && d.co >= 32 && d.c1 + d.c2 >= 320;
}

int conv1x1_launch(const edvr_conv2d_desc &d, hipStream_t stream) {
  Conv1x1Args a;
  a.d = d;
  a.ci = ((d.c1 + d.c2 + 63) / 64) * 64;  // k-steps past the real channels load no x (zero B operand) and zero weights
  a.cip = ((d.c1 + d.c2 + 31) / 32) * 32;
  a.cop = (d.co + 31) / 32 * 32;
  const int hw = d.h * d.w;
  a.seg_shift = 30;  // largest power-of-two channel segment whose planes fit 32-bit offsets (>= 1: a pair always does, conv1x1_eligible)
  while (((int64_t)1 << a.seg_shift) * hw * 4 >= ((int64_t)1 << 31)) --a.seg_shift;
  const int full = d.co / 128, rem_tiles = cdiv(d.co - full * 128, 32);
  if (full > 0) {
    a.co_start = 0;
    hipLaunchKernelGGL((conv1x1_stream_kernel<4>), dim3(cdiv(hw, 128), full, d.n), dim3(256), 0, stream, a);
  }
  if (rem_tiles > 0) {
    a.co_start = full * 128;
    const dim3 grid(cdiv(hw, 128), 1, d.n);
    if (rem_tiles == 1) hipLaunchKernelGGL((conv1x1_stream_kernel<1>), grid, dim3(256), 0, stream, a);
    else if (rem_tiles == 2) hipLaunchKernelGGL((conv1x1_stream_kernel<2>), grid, dim3(256), 0, stream, a);
    else hipLaunchKernelGGL((conv1x1_stream_kernel<3>), grid, dim3(256), 0, stream, a);
  }
  return check_launch("conv1x1_stream_kernel");
}

}  // namespace edvr
