// gemm_nt_s.hip - C[M,N] (+)= sum_batches A_b[M,K] B_b[N,K]^T (gemm_nt_kernel of dcn.hip) with SPLIT fp32 operands on the f16
// matrix pipe (gfx950).
//
// The product behind dW = sum_{b,p} dY col^T of the DCN backward (deform_conv_cuda.cpp:664-672) and behind the weight gradient of
// the 1x1 convolutions: K = the pixel axis, contiguous in both operands.  Same tiling as the fp32 kernel (a 256-thread workgroup
// owns 128 x 128 of C, 4 waves as 2 x 2, K in chunks of 32 through a double-buffered LDS pair, global loads of chunk c + 1 in flight
// under the MFMAs of chunk c, deterministic split-K into ws[split][M][N]); what changes is the arithmetic: an element travels as
// ONE dword (f16 hi | f16 lo << 16) of x * s, s a power of two from a magnitude bound of its operand (winograd_f4s.hip), made when
// the chunk is committed to LDS (two instructions per element, once per workgroup).  v_mfma_f32_32x32x16_f16 takes 8 k-slots per
// lane = (hi, lo) of FOUR pixels: one ds_read_b128 per operand; the first MFMA of a pair accumulates a_hi b_hi + a_lo b_lo, the
// second, with B rotated by 16 bits, the cross terms.  Per chunk and wave 32 MFMAs of 32 cycles where the fp32 kernel issues 64 of
// 64.  The row stride of 36 dwords puts the 16 lanes of one 16-byte pass on 16 different bank quads.  Partials leave unscaled
// (1 / (s_A s_B) is a power of two: exact).
#include "common.h"
#include <cstdlib>

namespace edvr {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef int gs_i32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 gs_f16x8 __attribute__((ext_vector_type(8)));

namespace {
constexpr int GSK = 32, GS_LD = GSK + 4;

// x * s = hi + lo in f16: four values, the independent first halves in front of the dependent second ones (winograd_wgrad_s.hip)
__device__ __forceinline__ void gs_split4(const float (&x)[4], float s, unsigned (&o)[4]) {
  asm volatile(
      "v_fma_mixlo_f16 %0, %4, %8, 0\n\tv_fma_mixlo_f16 %1, %5, %8, 0\n\tv_fma_mixlo_f16 %2, %6, %8, 0\n\tv_fma_mixlo_f16 %3, %7, %8, 0\n\t"
      "v_fma_mixhi_f16 %0, %4, %8, -%0 op_sel_hi:[0,0,1]\n\tv_fma_mixhi_f16 %1, %5, %8, -%1 op_sel_hi:[0,0,1]\n\t"
      "v_fma_mixhi_f16 %2, %6, %8, -%2 op_sel_hi:[0,0,1]\n\tv_fma_mixhi_f16 %3, %7, %8, -%3 op_sel_hi:[0,0,1]"
      : "=&v"(o[0]), "=&v"(o[1]), "=&v"(o[2]), "=&v"(o[3])
      : "v"(x[0]), "v"(x[1]), "v"(x[2]), "v"(x[3]), "s"(s));
}
// 2^e with amax * 2^e < 2^15: amax = m 2^k, m in [1, 2) -> e = 14 - k (clamped: a zero / tiny / huge bound stays a normal number)
__device__ __forceinline__ float gs_scale(float amax) {
  const int be = (int)((__builtin_bit_cast(unsigned, amax) >> 23) & 255u);
  return __builtin_bit_cast(float, (unsigned)min(max(127 + 14 - (be - 127), 7), 215) << 23);
}
}  // namespace

struct GemmNTS {
  const float *A, *B, *a_amax, *b_amax;
  float *ws;
  int M, N, nb;
  int64_t K, lda, ldb, a_bs, b_bs;
  int chunks_per_batch, total_chunks, splits;
};

template <bool VEC>  // VEC: every row start is 16-byte aligned and K % 4 == 0 (float4 loads), else scalar loads
__global__ __launch_bounds__(256, 2) void gemm_nt_split_kernel(const GemmNTS g) {
  constexpr int BM = 128, BN = 128, LD = GS_LD;
  __shared__ __attribute__((aligned(16))) unsigned as[2][BM * LD];
  __shared__ __attribute__((aligned(16))) unsigned bs[2][BN * LD];
  const float s_a = gs_scale(__builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, *g.a_amax))));
  const float s_b = gs_scale(__builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, *g.b_amax))));
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, half = lane >> 5, j = lane & 31;
  const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN, split = blockIdx.z;
  const int wm = (wave & 1) * 64, wn = (wave >> 1) * 64;
  const int c_begin = (int)((int64_t)g.total_chunks * split / g.splits);
  const int c_end = (int)((int64_t)g.total_chunks * (split + 1) / g.splits);
  f32x16 acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
  // staging role: 4 consecutive k of row (tid >> 3) + 32 i, i = 0..3, of both operands
  const int k4 = (tid & 7) * 4, row0 = tid >> 3;
  float ar[4][4], br[4][4];
  auto load = [&](int c) {
    const int b = c / g.chunks_per_batch;
    const int64_t k = (int64_t)(c - b * g.chunks_per_batch) * GSK + k4;
    const float *Ab = g.A + (int64_t)b * g.a_bs, *Bb = g.B + (int64_t)b * g.b_bs;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int row = row0 + 32 * i;
      const bool aok = m0 + row < g.M, bok = n0 + row < g.N;
      const float *pa = Ab + (int64_t)(aok ? m0 + row : 0) * g.lda, *pb = Bb + (int64_t)(bok ? n0 + row : 0) * g.ldb;
      if (VEC) {
        const bool kok = k < g.K;  // K % 4 == 0: the four elements are valid together
        const float4 va = *reinterpret_cast<const float4 *>(pa + (kok ? k : 0)), vb = *reinterpret_cast<const float4 *>(pb + (kok ? k : 0));
        ar[i][0] = (aok && kok) ? va.x : 0.f; ar[i][1] = (aok && kok) ? va.y : 0.f; ar[i][2] = (aok && kok) ? va.z : 0.f; ar[i][3] = (aok && kok) ? va.w : 0.f;
        br[i][0] = (bok && kok) ? vb.x : 0.f; br[i][1] = (bok && kok) ? vb.y : 0.f; br[i][2] = (bok && kok) ? vb.z : 0.f; br[i][3] = (bok && kok) ? vb.w : 0.f;
      } else {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const bool kok = k + q < g.K;
          const float va = pa[kok ? k + q : 0], vb = pb[kok ? k + q : 0];
          ar[i][q] = (aok && kok) ? va : 0.f;
          br[i][q] = (bok && kok) ? vb : 0.f;
        }
      }
    }
  };
  auto commit = [&](int buf) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      unsigned pa[4], pb[4];
      gs_split4(ar[i], s_a, pa);
      gs_split4(br[i], s_b, pb);
      *reinterpret_cast<gs_i32x4 *>(&as[buf][(row0 + 32 * i) * LD + k4]) = gs_i32x4{(int)pa[0], (int)pa[1], (int)pa[2], (int)pa[3]};
      *reinterpret_cast<gs_i32x4 *>(&bs[buf][(row0 + 32 * i) * LD + k4]) = gs_i32x4{(int)pb[0], (int)pb[1], (int)pb[2], (int)pb[3]};
    }
  };
  if (c_begin < c_end) {
    load(c_begin);
    commit(0);
  }
  __syncthreads();
  for (int c = c_begin; c < c_end; ++c) {
    const int buf = (c - c_begin) & 1;
    const bool more = c + 1 < c_end;
    if (more) load(c + 1);
    const unsigned *pa = as[buf] + (wm + j) * LD + 4 * half, *pb = bs[buf] + (wn + j) * LD + 4 * half;
#pragma unroll
    for (int kk = 0; kk < GSK; kk += 8) {  // 8 pixels per MFMA: lanes 0-31 carry pixels kk .. kk + 3, lanes 32-63 pixels kk + 4 .. kk + 7
      const gs_i32x4 a0 = *reinterpret_cast<const gs_i32x4 *>(pa + kk), a1 = *reinterpret_cast<const gs_i32x4 *>(pa + 32 * LD + kk);
      const gs_i32x4 b0 = *reinterpret_cast<const gs_i32x4 *>(pb + kk), b1 = *reinterpret_cast<const gs_i32x4 *>(pb + 32 * LD + kk);
      gs_i32x4 b0r, b1r;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        b0r[q] = (int)__builtin_amdgcn_alignbit((unsigned)b0[q], (unsigned)b0[q], 16);
        b1r[q] = (int)__builtin_amdgcn_alignbit((unsigned)b1[q], (unsigned)b1[q], 16);
      }
      const gs_f16x8 A0 = __builtin_bit_cast(gs_f16x8, a0), A1 = __builtin_bit_cast(gs_f16x8, a1);
      acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A0, __builtin_bit_cast(gs_f16x8, b0), acc[0][0], 0, 0, 0);
      acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A0, __builtin_bit_cast(gs_f16x8, b1), acc[0][1], 0, 0, 0);
      acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A1, __builtin_bit_cast(gs_f16x8, b0), acc[1][0], 0, 0, 0);
      acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A1, __builtin_bit_cast(gs_f16x8, b1), acc[1][1], 0, 0, 0);
      acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A0, __builtin_bit_cast(gs_f16x8, b0r), acc[0][0], 0, 0, 0);
      acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A0, __builtin_bit_cast(gs_f16x8, b1r), acc[0][1], 0, 0, 0);
      acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A1, __builtin_bit_cast(gs_f16x8, b0r), acc[1][0], 0, 0, 0);
      acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A1, __builtin_bit_cast(gs_f16x8, b1r), acc[1][1], 0, 0, 0);
    }
    if (more) commit(buf ^ 1);  // the idle buffer: last read one iteration ago, a barrier since
    __syncthreads();
  }
  const float unscale = 1.f / (s_a * s_b);
  float *out = g.ws + (int64_t)split * g.M * g.N;
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = m0 + wm + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * half, n = n0 + wn + b * 32 + j;
        if (m < g.M && n < g.N) out[(int64_t)m * g.N + n] = acc[a][b][r] * unscale;
      }
}

bool gemm_nt_split_enabled() {
  static const bool on = []() {
    const char *e = getenv("EDVR_GEMM_SPLIT");  // "0": the fp32 kernel (A/B, tests)
    return !(e && e[0] == '0');
  }();
  return on;
}

// Same contract (splits, workspace, reduction) as gemm_nt_batched; a_amax / b_amax: device pointers to ONE float >= max |A|, max |B|.
int gemm_nt_split_batched(const float *A, const float *B, float *C, int M, int N, int64_t K, int64_t lda, int64_t ldb, int nb,
                          int64_t a_bs, int64_t b_bs, bool accumulate, float *ws, const float *a_amax, const float *b_amax,
                          hipStream_t stream) {
  GemmNTS g;
  g.A = A; g.B = B; g.a_amax = a_amax; g.b_amax = b_amax; g.ws = ws; g.M = M; g.N = N; g.nb = nb; g.K = K; g.lda = lda; g.ldb = ldb;
  g.a_bs = a_bs; g.b_bs = b_bs;
  g.chunks_per_batch = (int)cdiv64(K, GSK);
  g.total_chunks = g.chunks_per_batch * nb;
  g.splits = (int)(gemm_nt_ws_elems_b(M, N, K, nb) / ((size_t)M * N));  // the fp32 kernel's split count (and workspace)
  dim3 grid(cdiv(N, 128), cdiv(M, 128), g.splits);
  const bool vec = ((K | lda | ldb | a_bs | b_bs) & 3) == 0 && ((reinterpret_cast<uintptr_t>(A) | reinterpret_cast<uintptr_t>(B)) & 15) == 0;
  if (vec) hipLaunchKernelGGL(gemm_nt_split_kernel<true>, grid, dim3(256), 0, stream, g);
  else hipLaunchKernelGGL(gemm_nt_split_kernel<false>, grid, dim3(256), 0, stream, g);
  int rc = check_launch("gemm_nt_split_kernel");
  if (rc) return rc;
  return reduce_partials_launch(ws, C, (int64_t)M * N, g.splits, accumulate ? 1 : 0, stream);
}

}  // namespace edvr
