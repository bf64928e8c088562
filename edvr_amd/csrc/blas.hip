// blas.hip - the two PLAIN GEMMs of the DCNv2 backward on rocBLAS (fp32 sgemm, exact fp32 MFMA on gfx950).
//
// dcol[b] = W^T dY[b] and dW = sum_b dY[b] col[b]^T have no fusion opportunity left (the column buffer is consumed / produced
// by the coordinate kernel), so they are library GEMMs: measured 100-110 TF/s on the EDVR shapes vs 48 (1x1 instance of the
// direct conv kernel) and 55 TF/s (gemm_nt_kernel).  Replaces the two addmm_ calls of the reference's backward
// (deform_conv_cuda.cpp:627-632 columns = W^T grad_output, :664-672 grad_weight += grad_output columns^T).
#include <rocblas/rocblas.h>

#include "common.h"

namespace edvr {

static rocblas_handle blas_handle(hipStream_t stream) {
  static thread_local rocblas_handle h = nullptr;
  if (!h) {
    if (rocblas_create_handle(&h) != rocblas_status_success) {
      h = nullptr;
      return nullptr;
    }
    rocblas_set_pointer_mode(h, rocblas_pointer_mode_host);
  }
  if (rocblas_set_stream(h, stream) != rocblas_status_success) return nullptr;
  return h;
}

// Row-major C_b[M x N] = op(A_b) op(B_b) over `batch` matrices (strides in elements), via the column-major identity
// C^T = op(B)^T op(A)^T.  A is M x K (lda) or, if a_trans, stored K x M; B is K x N (ldb) or, if b_trans, stored N x K.
int blas_gemm_rowmajor(const float *A, const float *B, float *C, int M, int N, int K, bool a_trans, bool b_trans, int64_t lda,
                       int64_t ldb, int64_t ldc, int64_t stride_a, int64_t stride_b, int64_t stride_c, int batch, hipStream_t stream) {
  rocblas_handle h = blas_handle(stream);
  if (!h) {
    set_error("rocBLAS: cannot create handle / set stream");
    return EDVR_ERR_LAUNCH;
  }
  const float one = 1.f, zero = 0.f;
  // column-major view: C^T (N x M, ld ldc) = op(B)^T (N x K) * op(A)^T (K x M); a row-major X (r x c, ld) is the column-major
  // X^T (c x r, ld), so the stored B is already "op(B)^T" unless it needs the opposite transpose.
  const rocblas_status st = rocblas_sgemm_strided_batched(
      h, b_trans ? rocblas_operation_transpose : rocblas_operation_none, a_trans ? rocblas_operation_transpose : rocblas_operation_none,
      N, M, K, &one, B, (rocblas_int)ldb, stride_b, A, (rocblas_int)lda, stride_a, &zero, C, (rocblas_int)ldc, stride_c, batch);
  if (st != rocblas_status_success) {
    set_error("rocblas_sgemm_strided_batched: %s", rocblas_status_to_string(st));
    return EDVR_ERR_LAUNCH;
  }
  return EDVR_OK;
}

}  // namespace edvr
