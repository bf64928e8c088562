// dcn_any.hip - DCNv2 forward / backward in float64 and float16: the other two legs of the reference's dtype dispatch
// (AT_DISPATCH_FLOATING_TYPES_AND_HALF, basicsr/models/ops/dcn/src/deform_conv_cuda_kernel.cu:781,811,843 -> double, float, half).
//
// EDVR itself runs in fp32 (dcn.hip / dcn_fused.hip: the measured path).  A drop-in `ops/dcn` also serves callers that hand it
// float64 tensors (torch.autograd.gradcheck) or float16 ones; those take this file: the reference's algorithm restated as three
// plain kernels per direction, any kernel size / stride / padding / dilation (EDVR_HW pairs) / groups / deformable groups,
// templated on the storage type T and the arithmetic type CT:
//   double : CT = double throughout (bilinear weights, products, sums, dX atomics) - what the reference computes for scalar_t = double;
//   half   : CT = float; tensors are float16 in memory, every intermediate (columns, sums, the dX accumulator) is float32 and is
//            rounded to float16 once on the way out.  (The reference computes its bilinear weights in half as well; its results
//            differ from the exact ones by more than these do.  Tests hold both against the fp64 oracle at float16 resolution.)
//   forward : dcn_any_im2col (columns, CT) -> gemm_any (y = W col + b)
//   backward: gemm_any (dcol = W^T dY) -> dcn_any_coord (d offset, d mask, dX by atomics, columns rebuilt in place)
//             -> gemm_any (dW = sum_b dY col^T) -> dcn_any_bias
// Correctness-first kernels (16 x 16 LDS tiles on the vector ALUs, no MFMA): these dtypes are not on the path bench.py measures.
#include <hip/hip_fp16.h>

#include <algorithm>
#include <type_traits>

#include "common.h"

namespace edvr {

template <typename T>
struct Compute {
  typedef float type;
};
template <>
struct Compute<double> {
  typedef double type;
};

template <typename T>
__device__ __forceinline__ typename Compute<T>::type ld(const T *p) {
  return (typename Compute<T>::type)(*p);
}
template <>
__device__ __forceinline__ float ld<__half>(const __half *p) {
  return __half2float(*p);
}
template <typename T, typename CT>
__device__ __forceinline__ void st(T *p, CT v) {
  *p = (T)v;
}
template <>
__device__ __forceinline__ void st<__half, float>(__half *p, float v) {
  *p = __float2half(v);
}

template <typename CT>
struct TapT {
  CT w00, w01, w10, w11;  // bilinear corner weights, 0 where the corner is outside the image or the tap invalid (.cu:467-497)
  CT lh, lw;
  int o00, o01, o10, o11;  // clamped element offsets inside one channel plane
  bool ok00, ok01, ok10, ok11;
};

// Sampling position (h, w) of one tap -> corners.  Same validity rule as the fp32 path (dcn.hip resolve_tap; reference .cu:618:
// h > -1, w > -1, h < H, w < W) and the same cell selection by floor(), so the one-sided derivative at integer positions is the
// reference's (.cu:526-568).
template <typename CT>
__device__ __forceinline__ TapT<CT> resolve_tap_t(CT h, CT w, int H, int W) {
  TapT<CT> t;
  const bool valid = (h > (CT)-1) && (w > (CT)-1) && (h < (CT)H) && (w < (CT)W);
  const CT fh = floor(h), fw = floor(w);
  const int h0 = (int)fh, w0 = (int)fw, h1 = h0 + 1, w1 = w0 + 1;
  t.lh = h - fh;
  t.lw = w - fw;
  const CT hh = (CT)1 - t.lh, hw = (CT)1 - t.lw;
  const bool r0 = valid && h0 >= 0, r1 = valid && h1 <= H - 1;
  const bool c0 = w0 >= 0, c1 = w1 <= W - 1;
  t.ok00 = r0 && c0;
  t.ok01 = r0 && c1;
  t.ok10 = r1 && c0;
  t.ok11 = r1 && c1;
  t.w00 = t.ok00 ? hh * hw : (CT)0;
  t.w01 = t.ok01 ? hh * t.lw : (CT)0;
  t.w10 = t.ok10 ? t.lh * hw : (CT)0;
  t.w11 = t.ok11 ? t.lh * t.lw : (CT)0;
  const int ch0 = min(max(h0, 0), H - 1), ch1 = min(max(h1, 0), H - 1);
  const int cw0 = min(max(w0, 0), W - 1), cw1 = min(max(w1, 0), W - 1);
  t.o00 = ch0 * W + cw0;
  t.o01 = ch0 * W + cw1;
  t.o10 = ch1 * W + cw0;
  t.o11 = ch1 * W + cw1;
  return t;
}

// decode (image, deformable group, tap, pixel) and resolve the tap; shared by the forward gather and the coordinate kernel
template <typename T, typename CT>
__device__ __forceinline__ TapT<CT> tap_of(const T *offset, const T *mask, const DcnShape &s, int64_t idx, int &b, int &g, int &k, int &p, CT &m) {
  const int K = s.kh * s.kw, P = s.Ho * s.Wo;
  p = (int)(idx % P);
  k = (int)((idx / P) % K);
  g = (int)((idx / ((int64_t)P * K)) % s.dg);
  b = (int)(idx / ((int64_t)P * K * s.dg));
  const int ho = p / s.Wo, wo = p - ho * s.Wo;
  const int i = k / s.kw, j = k - i * s.kw;
  const T *off_b = offset + (int64_t)b * s.off_bs + (int64_t)(g * 2 * K + 2 * k) * P + p;
  const CT dy = ld(off_b), dx = ld(off_b + P);
  m = ld(mask + (int64_t)b * s.msk_bs + (int64_t)(g * K + k) * P + p);
  return resolve_tap_t<CT>((CT)(ho * s.stride - s.pad + i * s.dil) + dy, (CT)(wo * s.stride_w - s.pad_w + j * s.dil_w) + dx, s.H, s.W);
}

// col[img, c * K + k, p] = mask * bilinear(x[img, c], p + tap + offset)    (.cu:570-633)
template <typename T, typename CT>
__global__ __launch_bounds__(256) void dcn_any_im2col_kernel(const T *__restrict__ x, const T *__restrict__ offset, const T *__restrict__ mask,
                                                             CT *__restrict__ col, const DcnShape s) {
  const int K = s.kh * s.kw, P = s.Ho * s.Wo, cpg = s.C / s.dg;
  const int64_t total = (int64_t)s.B * s.dg * K * P, plane = (int64_t)s.H * s.W;
  for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
    int b, g, k, p;
    CT m;
    const TapT<CT> t = tap_of<T, CT>(offset, mask, s, idx, b, g, k, p, m);
    const T *xp = x + ((int64_t)b * s.C + (int64_t)g * cpg) * plane;
    CT *cp = col + ((int64_t)b * s.C * K + (int64_t)(g * cpg) * K + k) * P + p;
    for (int cc = 0; cc < cpg; ++cc) {
      const CT v = t.w00 * ld(xp + t.o00) + t.w01 * ld(xp + t.o01) + t.w10 * ld(xp + t.o10) + t.w11 * ld(xp + t.o11);
      *cp = v * m;
      xp += plane;
      cp += (int64_t)K * P;
    }
  }
}

// backward, per (img, g, k, p), reduced over the group's channels (.cu:635-767): d(mask), d(offset) written; dX accumulated with
// atomics into a CT buffer (pre-zeroed); dcol (W^T dY) is replaced in place by the forward column the dW product needs.
template <typename T, typename CT>
__global__ __launch_bounds__(256) void dcn_any_coord_kernel(const T *__restrict__ x, const T *__restrict__ offset, const T *__restrict__ mask,
                                                            CT *__restrict__ dcol, CT *__restrict__ dx, T *__restrict__ doffset,
                                                            T *__restrict__ dmask, const DcnShape s) {
  const int K = s.kh * s.kw, P = s.Ho * s.Wo, cpg = s.C / s.dg;
  const int64_t total = (int64_t)s.B * s.dg * K * P, plane = (int64_t)s.H * s.W;
  for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
    int b, g, k, p;
    CT m;
    const TapT<CT> t = tap_of<T, CT>(offset, mask, s, idx, b, g, k, p, m);
    const CT hh = (CT)1 - t.lh, hw = (CT)1 - t.lw, z = (CT)0;
    const CT gy00 = t.ok00 ? -hw : z, gy01 = t.ok01 ? -t.lw : z, gy10 = t.ok10 ? hw : z, gy11 = t.ok11 ? t.lw : z;
    const CT gx00 = t.ok00 ? -hh : z, gx01 = t.ok01 ? hh : z, gx10 = t.ok10 ? -t.lh : z, gx11 = t.ok11 ? t.lh : z;
    const T *xp = x + ((int64_t)b * s.C + (int64_t)g * cpg) * plane;
    CT *gp = dx + ((int64_t)b * s.C + (int64_t)g * cpg) * plane;
    CT *cp = dcol + ((int64_t)b * s.C * K + (int64_t)(g * cpg) * K + k) * P + p;
    CT s_m = z, s_y = z, s_x = z;
    for (int cc = 0; cc < cpg; ++cc) {
      const CT dc = *cp;
      const CT a00 = ld(xp + t.o00), a01 = ld(xp + t.o01), a10 = ld(xp + t.o10), a11 = ld(xp + t.o11);
      const CT val = t.w00 * a00 + t.w01 * a01 + t.w10 * a10 + t.w11 * a11;
      s_m += dc * val;
      s_y += dc * (gy00 * a00 + gy01 * a01 + gy10 * a10 + gy11 * a11);
      s_x += dc * (gx00 * a00 + gx01 * a01 + gx10 * a10 + gx11 * a11);
      const CT tt = dc * m;
      if (t.ok00) atomicAdd(gp + t.o00, t.w00 * tt);
      if (t.ok01) atomicAdd(gp + t.o01, t.w01 * tt);
      if (t.ok10) atomicAdd(gp + t.o10, t.w10 * tt);
      if (t.ok11) atomicAdd(gp + t.o11, t.w11 * tt);
      *cp = val * m;
      xp += plane;
      gp += plane;
      cp += (int64_t)K * P;
    }
    st(dmask + (int64_t)b * s.dmsk_bs + (int64_t)(g * K + k) * P + p, s_m);
    T *dob = doffset + (int64_t)b * s.doff_bs + (int64_t)(g * 2 * K + 2 * k) * P + p;
    st(dob, s_y * m);
    st(dob + P, s_x * m);
  }
}

// C[z][m][n] (CT or T) = sum_k opA(A)[m][k] * opB(B)[k][n] (+ bias[m]), batched over blockIdx.z with element strides; the k axis
// may itself run over `kb` batches (dW: sum over the images).  16 x 16 output tile per workgroup, operands through LDS.
//   TA: A[m * lda + k], else A[k * lda + m];  TB: B[n * ldb + k], else B[k * ldb + n]
template <typename TA_, typename TB_, typename TC_, typename CT, bool TRANS_A, bool TRANS_B>
__global__ __launch_bounds__(256) void gemm_any_kernel(const TA_ *__restrict__ A, const TB_ *__restrict__ Bm, TC_ *__restrict__ Cm,
                                                       const TC_ *__restrict__ bias, int M, int N, int Kd, int64_t lda, int64_t ldb,
                                                       int64_t ldc, int64_t a_zs, int64_t b_zs, int64_t c_zs, int kb, int64_t a_ks,
                                                       int64_t b_ks) {
  __shared__ CT As[16][17], Bs[16][17];
  const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
  const int m0 = blockIdx.y * 16, n0 = blockIdx.x * 16, zb = blockIdx.z;
  CT acc = (CT)0;
  for (int kbi = 0; kbi < kb; ++kbi) {
    const TA_ *Ab = A + zb * a_zs + kbi * a_ks;
    const TB_ *Bb = Bm + zb * b_zs + kbi * b_ks;
    for (int k0 = 0; k0 < Kd; k0 += 16) {
      {  // A tile: As[mi][ki], B tile: Bs[ki][ni]; loaded so that the contiguous axis of each operand runs along tx
        const int mi = TRANS_A ? ty : tx, ki = TRANS_A ? tx : ty;  // TRANS_A: k contiguous
        const int m = m0 + mi, k = k0 + ki;
        CT v = (CT)0;
        if (m < M && k < Kd) v = (CT)ld(TRANS_A ? Ab + (int64_t)m * lda + k : Ab + (int64_t)k * lda + m);
        As[mi][ki] = v;
      }
      {
        const int ni = TRANS_B ? ty : tx, ki = TRANS_B ? tx : ty;  // TRANS_B: k contiguous
        const int n = n0 + ni, k = k0 + ki;
        CT v = (CT)0;
        if (n < N && k < Kd) v = (CT)ld(TRANS_B ? Bb + (int64_t)n * ldb + k : Bb + (int64_t)k * ldb + n);
        Bs[ki][ni] = v;
      }
      __syncthreads();
#pragma unroll
      for (int kk = 0; kk < 16; ++kk) acc += As[ty][kk] * Bs[kk][tx];
      __syncthreads();
    }
  }
  const int m = m0 + ty, n = n0 + tx;
  if (m < M && n < N) {
    if (bias) acc += (CT)ld(bias + m);
    st(Cm + zb * c_zs + (int64_t)m * ldc + n, acc);
  }
}

// db[co] = sum_{b, p} dy[b, co, p]: one workgroup per channel
template <typename T, typename CT>
__global__ __launch_bounds__(256) void dcn_any_bias_kernel(const T *__restrict__ dy, T *__restrict__ db, int B, int Co, int64_t P) {
  __shared__ CT red[256];
  const int co = blockIdx.x;
  CT sum = (CT)0;
  for (int b = 0; b < B; ++b)
    for (int64_t p = threadIdx.x; p < P; p += 256) sum += (CT)ld(dy + ((int64_t)b * Co + co) * P + p);
  red[threadIdx.x] = sum;
  __syncthreads();
  for (int st_ = 128; st_ > 0; st_ >>= 1) {
    if ((int)threadIdx.x < st_) red[threadIdx.x] += red[threadIdx.x + st_];
    __syncthreads();
  }
  if (threadIdx.x == 0) st(db + co, red[0]);
}

template <typename T, typename CT>
__global__ void convert_kernel(const CT *__restrict__ src, T *__restrict__ dst, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) st(dst + i, src[i]);
}

static inline size_t up256(size_t v) { return (v + 255) / 256 * 256; }
static inline unsigned grid_for(int64_t n) { return (unsigned)std::min<int64_t>(cdiv64(n, 256), 1 << 20); }

template <typename T>
size_t any_ws_bytes(const DcnShape &s) {
  typedef typename Compute<T>::type CT;
  const size_t K = (size_t)s.kh * s.kw, P = (size_t)s.Ho * s.Wo;
  return up256((size_t)s.B * s.C * K * P * sizeof(CT)) + up256((size_t)s.B * s.C * s.H * s.W * sizeof(CT));  // columns + dX accumulator
}

template <typename T>
int any_forward(const T *x, const T *offset, const T *mask, const T *weight, const T *bias, T *y, const DcnShape &s, void *ws,
                hipStream_t stream) {
  typedef typename Compute<T>::type CT;
  const int K = s.kh * s.kw, P = s.Ho * s.Wo, cig = s.C / s.groups, cog = s.Co / s.groups;
  CT *col = static_cast<CT *>(ws);
  hipLaunchKernelGGL((dcn_any_im2col_kernel<T, CT>), dim3(grid_for((int64_t)s.B * s.dg * K * P)), dim3(256), 0, stream, x, offset, mask, col, s);
  for (int g = 0; g < s.groups; ++g)  // y[b, g] = W[g] col[b, g] + bias[g]     (deform_conv_cuda.cpp:545-560)
    hipLaunchKernelGGL((gemm_any_kernel<T, CT, T, CT, true, false>), dim3(cdiv(P, 16), cdiv(cog, 16), s.B), dim3(256), 0, stream,
                       weight + (size_t)g * cog * cig * K, col + (size_t)g * cig * K * P, y + (size_t)g * cog * P,
                       bias ? bias + (size_t)g * cog : nullptr, cog, P, cig * K, (int64_t)cig * K, (int64_t)P, (int64_t)P, (int64_t)0,
                       (int64_t)s.C * K * P, (int64_t)s.Co * P, 1, (int64_t)0, (int64_t)0);
  return check_launch("dcn_any forward");
}

template <typename T>
int any_backward(const T *x, const T *offset, const T *mask, const T *weight, const T *dy, T *dx, T *doffset, T *dmask, T *dweight, T *dbias,
                 const DcnShape &s, void *ws, hipStream_t stream) {
  typedef typename Compute<T>::type CT;
  const int K = s.kh * s.kw, P = s.Ho * s.Wo, cig = s.C / s.groups, cog = s.Co / s.groups;
  const size_t n_dx = (size_t)s.B * s.C * s.H * s.W;
  CT *col = static_cast<CT *>(ws);
  CT *dxa = reinterpret_cast<CT *>(static_cast<char *>(ws) + up256((size_t)s.B * s.C * K * P * sizeof(CT)));
  CT *dx_acc = std::is_same<T, CT>::value ? reinterpret_cast<CT *>(dx) : dxa;
  if (hipMemsetAsync(dx_acc, 0, n_dx * sizeof(CT), stream) != hipSuccess) {
    set_error("dcn_any backward: hipMemsetAsync failed");
    return EDVR_ERR_LAUNCH;
  }
  for (int g = 0; g < s.groups; ++g)  // dcol[b, g] = W[g]^T dY[b, g]     (deform_conv_cuda.cpp:623-632)
    hipLaunchKernelGGL((gemm_any_kernel<T, T, CT, CT, false, false>), dim3(cdiv(P, 16), cdiv(cig * K, 16), s.B), dim3(256), 0, stream,
                       weight + (size_t)g * cog * cig * K, dy + (size_t)g * cog * P, col + (size_t)g * cig * K * P,
                       static_cast<const CT *>(nullptr), cig * K, P, cog, (int64_t)cig * K, (int64_t)P, (int64_t)P, (int64_t)0,
                       (int64_t)s.Co * P, (int64_t)s.C * K * P, 1, (int64_t)0, (int64_t)0);
  hipLaunchKernelGGL((dcn_any_coord_kernel<T, CT>), dim3(grid_for((int64_t)s.B * s.dg * K * P)), dim3(256), 0, stream, x, offset, mask, col, dx_acc,
                     doffset, dmask, s);
  if (!std::is_same<T, CT>::value)
    hipLaunchKernelGGL((convert_kernel<T, CT>), dim3(grid_for((int64_t)n_dx)), dim3(256), 0, stream, dx_acc, dx, (int64_t)n_dx);
  for (int g = 0; g < s.groups; ++g)  // dW[g] = sum_b dY[b, g] col[b, g]^T     (deform_conv_cuda.cpp:659-672)
    hipLaunchKernelGGL((gemm_any_kernel<T, CT, T, CT, true, true>), dim3(cdiv(cig * K, 16), cdiv(cog, 16), 1), dim3(256), 0, stream,
                       dy + (size_t)g * cog * P, col + (size_t)g * cig * K * P, dweight + (size_t)g * cog * cig * K,
                       static_cast<const T *>(nullptr), cog, cig * K, P, (int64_t)P, (int64_t)P, (int64_t)cig * K, (int64_t)0, (int64_t)0,
                       (int64_t)0, s.B, (int64_t)s.Co * P, (int64_t)s.C * K * P);
  if (dbias) hipLaunchKernelGGL((dcn_any_bias_kernel<T, CT>), dim3(s.Co), dim3(256), 0, stream, dy, dbias, s.B, s.Co, (int64_t)P);
  return check_launch("dcn_any backward");
}

}  // namespace edvr

extern "C" {

size_t edvr_dcnv2_any_ws_bytes(int dtype, int B, int C, int H, int W, int Co, int kh, int kw, int stride, int pad, int dil, int groups,
                               int dg) {
  edvr::DcnShape s;
  if (edvr::dcn_fill_shape(s, B, C, H, W, Co, kh, kw, stride, pad, dil, groups, dg, 0, 0)) return 0;
  if (dtype == EDVR_DTYPE_F64) return edvr::any_ws_bytes<double>(s);
  if (dtype == EDVR_DTYPE_F16) return edvr::any_ws_bytes<__half>(s);
  return std::max(edvr_dcnv2_fwd_ws_bytes(B, C, H, W, Co, kh, kw, stride, pad, dil, groups, dg),
                  edvr_dcnv2_bwd_ws_bytes(B, C, H, W, Co, kh, kw, stride, pad, dil, groups, dg));
}

int edvr_dcnv2_fwd_any(int dtype, const void *x, const void *offset, const void *mask, const void *weight, const void *bias, void *y, int B,
                       int C, int H, int W, int Co, int kh, int kw, int stride, int pad, int dil, int groups, int dg,
                       int64_t offset_bstride, int64_t mask_bstride, void *ws, size_t ws_bytes, edvr_stream_t stream) {
  using namespace edvr;
  if (dtype == EDVR_DTYPE_F32)
    return edvr_dcnv2_fwd_f32(static_cast<const float *>(x), static_cast<const float *>(offset), static_cast<const float *>(mask),
                              static_cast<const float *>(weight), static_cast<const float *>(bias), static_cast<float *>(y), B, C, H, W, Co,
                              kh, kw, stride, pad, dil, groups, dg, offset_bstride, mask_bstride, EDVR_ACT_NONE, 0, ws, ws_bytes, stream);
  EDVR_REQUIRE(dtype == EDVR_DTYPE_F64 || dtype == EDVR_DTYPE_F16, "dcnv2_fwd_any: unknown dtype %d", dtype);
  EDVR_REQUIRE(x && offset && mask && weight && y, "dcnv2_fwd_any: null pointer");
  DcnShape s;
  int rc = dcn_fill_shape(s, B, C, H, W, Co, kh, kw, stride, pad, dil, groups, dg, offset_bstride, mask_bstride);
  if (rc) return rc;
  if (ws_bytes < edvr_dcnv2_any_ws_bytes(dtype, B, C, H, W, Co, kh, kw, stride, pad, dil, groups, dg) || !ws) {
    set_error("dcnv2_fwd_any: workspace too small");
    return EDVR_ERR_WORKSPACE;
  }
  if (dtype == EDVR_DTYPE_F64)
    return any_forward<double>(static_cast<const double *>(x), static_cast<const double *>(offset), static_cast<const double *>(mask),
                               static_cast<const double *>(weight), static_cast<const double *>(bias), static_cast<double *>(y), s, ws,
                               as_stream(stream));
  return any_forward<__half>(static_cast<const __half *>(x), static_cast<const __half *>(offset), static_cast<const __half *>(mask),
                             static_cast<const __half *>(weight), static_cast<const __half *>(bias), static_cast<__half *>(y), s, ws,
                             as_stream(stream));
}

int edvr_dcnv2_bwd_any(int dtype, const void *x, const void *offset, const void *mask, const void *weight, const void *dy, void *dx,
                       void *doffset, void *dmask, void *dweight, void *dbias, int B, int C, int H, int W, int Co, int kh, int kw, int stride,
                       int pad, int dil, int groups, int dg, int64_t offset_bstride, int64_t mask_bstride, int64_t doffset_bstride,
                       int64_t dmask_bstride, void *ws, size_t ws_bytes, edvr_stream_t stream) {
  using namespace edvr;
  if (dtype == EDVR_DTYPE_F32)
    return edvr_dcnv2_bwd_f32(static_cast<const float *>(x), static_cast<const float *>(offset), static_cast<const float *>(mask),
                              static_cast<const float *>(weight), static_cast<const float *>(dy), static_cast<float *>(dx),
                              static_cast<float *>(doffset), static_cast<float *>(dmask), static_cast<float *>(dweight),
                              static_cast<float *>(dbias), B, C, H, W, Co, kh, kw, stride, pad, dil, groups, dg, offset_bstride, mask_bstride,
                              doffset_bstride, dmask_bstride, EDVR_DCN_SCATTER_AUTO, ws, ws_bytes, stream);
  EDVR_REQUIRE(dtype == EDVR_DTYPE_F64 || dtype == EDVR_DTYPE_F16, "dcnv2_bwd_any: unknown dtype %d", dtype);
  EDVR_REQUIRE(x && offset && mask && weight && dy && dx && doffset && dmask && dweight, "dcnv2_bwd_any: null pointer");
  DcnShape s;
  int rc = dcn_fill_shape(s, B, C, H, W, Co, kh, kw, stride, pad, dil, groups, dg, offset_bstride, mask_bstride);
  if (rc) return rc;
  if (doffset_bstride) s.doff_bs = doffset_bstride;
  if (dmask_bstride) s.dmsk_bs = dmask_bstride;
  if (ws_bytes < edvr_dcnv2_any_ws_bytes(dtype, B, C, H, W, Co, kh, kw, stride, pad, dil, groups, dg) || !ws) {
    set_error("dcnv2_bwd_any: workspace too small");
    return EDVR_ERR_WORKSPACE;
  }
  if (dtype == EDVR_DTYPE_F64)
    return any_backward<double>(static_cast<const double *>(x), static_cast<const double *>(offset), static_cast<const double *>(mask),
                                static_cast<const double *>(weight), static_cast<const double *>(dy), static_cast<double *>(dx),
                                static_cast<double *>(doffset), static_cast<double *>(dmask), static_cast<double *>(dweight),
                                static_cast<double *>(dbias), s, ws, as_stream(stream));
  return any_backward<__half>(static_cast<const __half *>(x), static_cast<const __half *>(offset), static_cast<const __half *>(mask),
                              static_cast<const __half *>(weight), static_cast<const __half *>(dy), static_cast<__half *>(dx),
                              static_cast<__half *>(doffset), static_cast<__half *>(dmask), static_cast<__half *>(dweight),
                              static_cast<__half *>(dbias), s, ws, as_stream(stream));
}

}  // extern "C"
