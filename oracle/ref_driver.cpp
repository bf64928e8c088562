// TEST INFRASTRUCTURE ONLY.  Builds oracle/_ref/libdcn_ref.so (git-ignored).
//
// This translation unit #includes the reference's own CUDA kernel file
//   basicsr/models/ops/dcn/src/deform_conv_cuda_kernel.cu        (REF_CU)
// from where it lies under /root/reference, compiled as serial C++ through the
// shim headers in oracle/shim/ (see oracle/shim/ATen/ATen.h).  The arithmetic of
// the deformable gather / scatter / coordinate-gradient kernels executed here
// is therefore the reference's own code, line for line.
//
// What is restated here (ours): only the HOST drivers of
//   basicsr/models/ops/dcn/src/deform_conv_cuda.cpp:490-569  (modulated forward)
//   basicsr/models/ops/dcn/src/deform_conv_cuda.cpp:571-685  (modulated backward)
//   basicsr/models/ops/dcn/src/deform_conv_cuda.cpp:152-488  (DCNv1 forward / backward_input / backward_parameters)
// whose at::addmm_ calls become plain triple loops (they need ATen, which the
// shim does not provide).  Per-sample loop, zeroed columns, group split and
// the final bias add follow those lines.
//
// Nothing in the product (edvr_amd/) may load this library.
#include REF_CU

#include <cstring>
#include <vector>

namespace {

template <typename T>
void gemm_nn_acc(const T *A, const T *B, T *C, int M, int K, int N) {
  // C[M,N] += A[M,K] * B[K,N]
  for (int m = 0; m < M; ++m)
    for (int k = 0; k < K; ++k) {
      const T a = A[(size_t)m * K + k];
      const T *b = B + (size_t)k * N;
      T *c = C + (size_t)m * N;
      for (int n = 0; n < N; ++n) c[n] += a * b[n];
    }
}

template <typename T>
void gemm_tn_set(const T *A, const T *B, T *C, int M, int K, int N) {
  // C[M,N] = A[K,M]^T * B[K,N]
  for (size_t i = 0; i < (size_t)M * N; ++i) C[i] = 0;
  for (int k = 0; k < K; ++k)
    for (int m = 0; m < M; ++m) {
      const T a = A[(size_t)k * M + m];
      const T *b = B + (size_t)k * N;
      T *c = C + (size_t)m * N;
      for (int n = 0; n < N; ++n) c[n] += a * b[n];
    }
}

template <typename T>
void gemm_nt_acc(const T *A, const T *B, T *C, int M, int K, int N) {
  // C[M,N] += A[M,K] * B[N,K]^T
  for (int m = 0; m < M; ++m)
    for (int n = 0; n < N; ++n) {
      T s = 0;
      for (int k = 0; k < K; ++k) s += A[(size_t)m * K + k] * B[(size_t)n * K + k];
      C[(size_t)m * N + n] += s;
    }
}

// deform_conv_cuda.cpp:490-569
template <typename T>
int mdcn_forward(const T *input, const T *weight, const T *bias, const T *offset, const T *mask, T *output,
                 int batch, int channels, int height, int width, int channels_out, int kh, int kw,
                 int stride, int pad, int dil, int group, int dg) {
  const int Ho = (height + 2 * pad - (dil * (kh - 1) + 1)) / stride + 1;
  const int Wo = (width + 2 * pad - (dil * (kw - 1) + 1)) / stride + 1;
  const int K = kh * kw, P = Ho * Wo;
  if (channels % group || channels_out % group || channels % dg) return -1;
  std::vector<T> columns((size_t)channels * K * P, (T)0);
  for (size_t i = 0; i < (size_t)batch * channels_out * P; ++i) output[i] = 0;  // :530
  const int cig = channels / group, cog = channels_out / group;
  for (int b = 0; b < batch; ++b) {
    const int n = channels * 1 * Ho * Wo;
    modulated_deformable_im2col_gpu_kernel<T>(
        n, input + (size_t)b * channels * height * width, offset + (size_t)b * dg * 2 * K * P,
        mask + (size_t)b * dg * K * P, height, width, kh, kw, pad, pad, stride, stride, dil, dil,
        channels / dg, 1, channels, dg, Ho, Wo, columns.data());  // :539-543
    for (int g = 0; g < group; ++g)                               // :550-555
      gemm_nn_acc(weight + (size_t)g * cog * cig * K, columns.data() + (size_t)g * cig * K * P,
                  output + ((size_t)b * channels_out + (size_t)g * cog) * P, cog, cig * K, P);
  }
  if (bias)  // :566-568
    for (int b = 0; b < batch; ++b)
      for (int o = 0; o < channels_out; ++o)
        for (int p = 0; p < P; ++p) output[((size_t)b * channels_out + o) * P + p] += bias[o];
  return 0;
}

// deform_conv_cuda.cpp:571-685.  grad_* must be zero-filled by the caller, as
// deform_conv.py:154-158 does.
template <typename T>
int mdcn_backward(const T *input, const T *weight, const T *offset, const T *mask, const T *grad_output,
                  T *grad_input, T *grad_weight, T *grad_bias, T *grad_offset, T *grad_mask, int batch,
                  int channels, int height, int width, int channels_out, int kh, int kw, int stride, int pad,
                  int dil, int group, int dg) {
  const int Ho = (height + 2 * pad - (dil * (kh - 1) + 1)) / stride + 1;
  const int Wo = (width + 2 * pad - (dil * (kw - 1) + 1)) / stride + 1;
  const int K = kh * kw, P = Ho * Wo;
  if (channels % group || channels_out % group || channels % dg) return -1;
  std::vector<T> columns((size_t)channels * K * P, (T)0);
  const int cig = channels / group, cog = channels_out / group;
  for (int b = 0; b < batch; ++b) {
    const T *in_b = input + (size_t)b * channels * height * width;
    const T *off_b = offset + (size_t)b * dg * 2 * K * P;
    const T *msk_b = mask + (size_t)b * dg * K * P;
    const T *go_b = grad_output + (size_t)b * channels_out * P;
    for (int g = 0; g < group; ++g)  // :621-626  columns[g] = W[g]^T * dY[b][g]
      gemm_tn_set(weight + (size_t)g * cog * cig * K, go_b + (size_t)g * cog * P,
                  columns.data() + (size_t)g * cig * K * P, cig * K, cog, P);
    modulated_deformable_col2im_coord_gpu_kernel<T>(  // :634-638
        1 * Ho * Wo * 2 * K * dg, columns.data(), in_b, off_b, msk_b, channels, height, width, kh, kw, pad,
        pad, stride, stride, dil, dil, channels * K / dg, 1, 2 * K * dg, dg, Ho, Wo,
        grad_offset + (size_t)b * dg * 2 * K * P, grad_mask + (size_t)b * dg * K * P);
    modulated_deformable_col2im_gpu_kernel<T>(  // :640-643
        channels * K * 1 * Ho * Wo, columns.data(), off_b, msk_b, channels, height, width, kh, kw, pad, pad,
        stride, stride, dil, dil, channels / dg, 1, dg, Ho, Wo,
        grad_input + (size_t)b * channels * height * width);
    modulated_deformable_im2col_gpu_kernel<T>(  // :647-650
        channels * 1 * Ho * Wo, in_b, off_b, msk_b, height, width, kh, kw, pad, pad, stride, stride, dil, dil,
        channels / dg, 1, channels, dg, Ho, Wo, columns.data());
    for (int g = 0; g < group; ++g) {  // :658-672
      gemm_nt_acc(go_b + (size_t)g * cog * P, columns.data() + (size_t)g * cig * K * P,
                  grad_weight + (size_t)g * cog * cig * K, cog, P, cig * K);
      if (grad_bias)
        for (int o = 0; o < cog; ++o) {
          T s = 0;
          for (int p = 0; p < P; ++p) s += go_b[((size_t)g * cog + o) * P + p];
          grad_bias[g * cog + o] += s;
        }
    }
  }
  return 0;
}

// ---- DCNv1 (DeformConv): deform_conv_cuda.cpp:152-243 (forward), :245-372 (backward_input), :374-488 (backward_parameters),
//      restated per sample (im2col_step = 1, so parallel_imgs = 1 and the column layout has batch_size 1).
template <typename T>
int dcn1_forward(const T *input, const T *weight, const T *offset, T *output, int batch, int channels, int height, int width,
                 int channels_out, int kh, int kw, int sh, int sw, int ph, int pw, int dh, int dw, int group, int dg) {
  const int Ho = (height + 2 * ph - (dh * (kh - 1) + 1)) / sh + 1;  // deform_conv_cuda.cpp:186-189
  const int Wo = (width + 2 * pw - (dw * (kw - 1) + 1)) / sw + 1;
  const int K = kh * kw, P = Ho * Wo;
  if (channels % group || channels_out % group || channels % dg) return -1;
  std::vector<T> columns((size_t)channels * K * P, (T)0);
  for (size_t i = 0; i < (size_t)batch * channels_out * P; ++i) output[i] = 0;
  const int cig = channels / group, cog = channels_out / group;
  for (int b = 0; b < batch; ++b) {
    deformable_im2col_gpu_kernel<T>(channels * Ho * Wo * 1, input + (size_t)b * channels * height * width,
                                    offset + (size_t)b * dg * 2 * K * P, height, width, kh, kw, ph, pw, sh, sw, dh, dw,
                                    channels / dg, 1, channels, dg, Ho, Wo, columns.data());  // .cu:252-276 launcher
    for (int g = 0; g < group; ++g)                                                             // .cpp:214-222
      gemm_nn_acc(weight + (size_t)g * cog * cig * K, columns.data() + (size_t)g * cig * K * P,
                  output + ((size_t)b * channels_out + (size_t)g * cog) * P, cog, cig * K, P);
  }
  return 0;
}

// grad_* must be zero-filled by the caller (deform_conv.py:71-76 allocates zeros)
template <typename T>
int dcn1_backward(const T *input, const T *weight, const T *offset, const T *grad_output, T *grad_input, T *grad_weight,
                  T *grad_offset, int batch, int channels, int height, int width, int channels_out, int kh, int kw, int sh, int sw,
                  int ph, int pw, int dh, int dw, int group, int dg) {
  const int Ho = (height + 2 * ph - (dh * (kh - 1) + 1)) / sh + 1;
  const int Wo = (width + 2 * pw - (dw * (kw - 1) + 1)) / sw + 1;
  const int K = kh * kw, P = Ho * Wo;
  if (channels % group || channels_out % group || channels % dg) return -1;
  std::vector<T> columns((size_t)channels * K * P, (T)0);
  const int cig = channels / group, cog = channels_out / group;
  for (int b = 0; b < batch; ++b) {
    const T *in_b = input + (size_t)b * channels * height * width;
    const T *off_b = offset + (size_t)b * dg * 2 * K * P;
    const T *go_b = grad_output + (size_t)b * channels_out * P;
    for (int g = 0; g < group; ++g)  // .cpp:325-334 columns[g] = W[g]^T dY[b][g]
      gemm_tn_set(weight + (size_t)g * cog * cig * K, go_b + (size_t)g * cog * P, columns.data() + (size_t)g * cig * K * P, cig * K,
                  cog, P);
    deformable_col2im_coord_gpu_kernel<T>(Ho * Wo * 2 * K * dg * 1, columns.data(), in_b, off_b, channels, height, width, kh, kw,
                                          ph, pw, sh, sw, dh, dw, channels * K / dg, 1, 2 * K * dg, dg, Ho, Wo,
                                          grad_offset + (size_t)b * dg * 2 * K * P);  // .cpp:341-344
    deformable_col2im_gpu_kernel<T>(channels * K * Ho * Wo * 1, columns.data(), off_b, channels, height, width, kh, kw, ph, pw,
                                    sh, sw, dh, dw, channels / dg, 1, dg, Ho, Wo,
                                    grad_input + (size_t)b * channels * height * width);  // .cpp:346-348
    deformable_im2col_gpu_kernel<T>(channels * Ho * Wo * 1, in_b, off_b, height, width, kh, kw, ph, pw, sh, sw, dh, dw,
                                    channels / dg, 1, channels, dg, Ho, Wo, columns.data());  // .cpp:445-447
    for (int g = 0; g < group; ++g)  // .cpp:460-468 (scale = 1)
      gemm_nt_acc(go_b + (size_t)g * cog * P, columns.data() + (size_t)g * cig * K * P, grad_weight + (size_t)g * cog * cig * K, cog, P,
                  cig * K);
  }
  return 0;
}

}  // namespace

extern "C" {

int ref_dcn1_forward_f64(const double *input, const double *weight, const double *offset, double *output, int batch, int channels,
                         int height, int width, int channels_out, int kh, int kw, int stride, int pad, int dil, int group, int dg) {
  return dcn1_forward<double>(input, weight, offset, output, batch, channels, height, width, channels_out, kh, kw, stride, stride, pad,
                              pad, dil, dil, group, dg);
}
// rectangular stride / padding / dilation (deform_conv.py:33-36 takes pairs; the kernels take h and w values separately)
int ref_dcn1_forward_rect_f64(const double *input, const double *weight, const double *offset, double *output, int batch, int channels,
                              int height, int width, int channels_out, int kh, int kw, int sh, int sw, int ph, int pw, int dh, int dw,
                              int group, int dg) {
  return dcn1_forward<double>(input, weight, offset, output, batch, channels, height, width, channels_out, kh, kw, sh, sw, ph, pw, dh,
                              dw, group, dg);
}
int ref_dcn1_backward_rect_f64(const double *input, const double *weight, const double *offset, const double *grad_output,
                               double *grad_input, double *grad_weight, double *grad_offset, int batch, int channels, int height,
                               int width, int channels_out, int kh, int kw, int sh, int sw, int ph, int pw, int dh, int dw, int group,
                               int dg) {
  return dcn1_backward<double>(input, weight, offset, grad_output, grad_input, grad_weight, grad_offset, batch, channels, height,
                               width, channels_out, kh, kw, sh, sw, ph, pw, dh, dw, group, dg);
}
int ref_dcn1_backward_f64(const double *input, const double *weight, const double *offset, const double *grad_output,
                          double *grad_input, double *grad_weight, double *grad_offset, int batch, int channels, int height,
                          int width, int channels_out, int kh, int kw, int stride, int pad, int dil, int group, int dg) {
  return dcn1_backward<double>(input, weight, offset, grad_output, grad_input, grad_weight, grad_offset, batch, channels, height,
                               width, channels_out, kh, kw, stride, stride, pad, pad, dil, dil, group, dg);
}


int ref_mdcn_forward_f64(const double *input, const double *weight, const double *bias, const double *offset,
                         const double *mask, double *output, int batch, int channels, int height, int width,
                         int channels_out, int kh, int kw, int stride, int pad, int dil, int group, int dg) {
  return mdcn_forward<double>(input, weight, bias, offset, mask, output, batch, channels, height, width,
                              channels_out, kh, kw, stride, pad, dil, group, dg);
}
int ref_mdcn_forward_f32(const float *input, const float *weight, const float *bias, const float *offset,
                         const float *mask, float *output, int batch, int channels, int height, int width,
                         int channels_out, int kh, int kw, int stride, int pad, int dil, int group, int dg) {
  return mdcn_forward<float>(input, weight, bias, offset, mask, output, batch, channels, height, width,
                             channels_out, kh, kw, stride, pad, dil, group, dg);
}
int ref_mdcn_backward_f64(const double *input, const double *weight, const double *offset, const double *mask,
                          const double *grad_output, double *grad_input, double *grad_weight,
                          double *grad_bias, double *grad_offset, double *grad_mask, int batch, int channels,
                          int height, int width, int channels_out, int kh, int kw, int stride, int pad,
                          int dil, int group, int dg) {
  return mdcn_backward<double>(input, weight, offset, mask, grad_output, grad_input, grad_weight, grad_bias,
                               grad_offset, grad_mask, batch, channels, height, width, channels_out, kh, kw,
                               stride, pad, dil, group, dg);
}
int ref_mdcn_backward_f32(const float *input, const float *weight, const float *offset, const float *mask,
                          const float *grad_output, float *grad_input, float *grad_weight, float *grad_bias,
                          float *grad_offset, float *grad_mask, int batch, int channels, int height, int width,
                          int channels_out, int kh, int kw, int stride, int pad, int dil, int group, int dg) {
  return mdcn_backward<float>(input, weight, offset, mask, grad_output, grad_input, grad_weight, grad_bias,
                              grad_offset, grad_mask, batch, channels, height, width, channels_out, kh, kw,
                              stride, pad, dil, group, dg);
}

}  // extern "C"
