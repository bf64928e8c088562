// Internal helpers shared by the HIP translation units of libedvr_amd.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>

#include "../../include/edvr_amd.h"

namespace edvr {

void set_error(const char *fmt, ...);

inline int check_launch(const char *what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    set_error("%s: %s", what, hipGetErrorString(e));
    return EDVR_ERR_LAUNCH;
  }
  return EDVR_OK;
}

inline hipStream_t as_stream(edvr_stream_t s) { return reinterpret_cast<hipStream_t>(s); }

inline int cdiv(int a, int b) { return (a + b - 1) / b; }
inline int64_t cdiv64(int64_t a, int64_t b) { return (a + b - 1) / b; }

// internal C++ entry used by dcn.hip for its GEMMs (same as the C ABI, no re-validation of wpk)
int conv2d_launch(const edvr_conv2d_desc &d, hipStream_t stream);

// C[M,N] (+)= A[M,K] * B[N,K]^T with K contiguous in both (pixel axis); deterministic split-K.
// ws must hold splits*M*N floats.  Used for dW of DCN (and conv wgrad on explicit columns).
size_t gemm_nt_ws_elems(int M, int N, int64_t K);
int gemm_nt_launch(const float *A, const float *B, float *C, int M, int N, int64_t K, int64_t lda, int64_t ldb,
                   bool accumulate, float *ws, hipStream_t stream);
// the same summed over `nb` (A, B) pairs a_bs / b_bs elements apart (images): dW of the DCNv2 backward and of the 1x1 convs
size_t gemm_nt_ws_elems_b(int M, int N, int64_t K, int nb);
int gemm_nt_batched(const float *A, const float *B, float *C, int M, int N, int64_t K, int64_t lda, int64_t ldb, int nb, int64_t a_bs,
                    int64_t b_bs, bool accumulate, float *ws, hipStream_t stream);
// gemm_nt_s.hip: the same product with split fp32 operands on the f16 matrix pipe (same splits / workspace); a_amax / b_amax = device
// pointers to one float >= max |A| / max |B|.  gemm_nt_split_enabled: EDVR_GEMM_SPLIT != 0
bool gemm_nt_split_enabled();
int gemm_nt_split_batched(const float *A, const float *B, float *C, int M, int N, int64_t K, int64_t lda, int64_t ldb, int nb, int64_t a_bs,
                          int64_t b_bs, bool accumulate, float *ws, const float *a_amax, const float *b_amax, hipStream_t stream);

// winograd.hip: F(2x2,3x3) path of the 3x3 / stride-1 convolution.  U (transformed weights, [ci_pad][16][round_up(co,64)])
// follows the direct packed layout inside the buffer edvr_conv2d_pack_weight_f32 fills.
bool winograd_eligible(const edvr_conv2d_desc &d);
int winograd_launch(const edvr_conv2d_desc &d, const float *U, int cop64, hipStream_t stream);
double winograd_executed_flops(const edvr_conv2d_desc &d);
int winograd_pack(const float *w, float *U, int co, int ci, int cop64, int cip, int transpose_flip, float *wpk_direct, int cop32,
                  hipStream_t stream);  // wpk_direct != nullptr: also writes the direct layout [cip][9][cop32] in the same launch

// winograd_f4.hip: F(4x4,3x3), wave-specialised workgroups; needs edvr_conv2d_desc.wpk_f4 (inference path)
bool winograd_f4_eligible(const edvr_conv2d_desc &d);
int winograd_f4_launch(const edvr_conv2d_desc &d, hipStream_t stream);
double winograd_f4_executed_flops(const edvr_conv2d_desc &d);

// winograd_f4s.hip: the same algorithm with split fp32 operands on the f16 matrix pipe; needs wpk_f4s and x_amax
bool winograd_f4s_eligible(const edvr_conv2d_desc &d);
int winograd_f4s_launch(const edvr_conv2d_desc &d, hipStream_t stream);
double winograd_f4s_executed_flops(const edvr_conv2d_desc &d);

// conv_small.hip: 3x3 / stride-1 conv with <= 4 output channels on the vector ALUs (EDVR's conv_last)
bool conv_small_eligible(const edvr_conv2d_desc &d);
int conv_small_launch(const edvr_conv2d_desc &d, hipStream_t stream);
bool wgrad_small_plan(int n, int c1, int c2, int h, int w, int co, int ks, int stride, int *splits);
size_t wgrad_small_ws_bytes(int co, int ci, int splits);
int wgrad_small_launch(const float *x, const float *dz, float *ws, int ci, int co, int n, int h, int w, int64_t x_img_stride,
                       int64_t dz_img_stride, int splits, hipStream_t stream);

// conv1x1.hip: 1x1 conv as a streaming GEMM (B operand straight from global memory)
bool conv1x1_eligible(const edvr_conv2d_desc &d);
int conv1x1_launch(const edvr_conv2d_desc &d, hipStream_t stream);
// conv1x1_s.hip: the same kernel with split fp32 operands on the f16 matrix pipe (needs d.wpk_f4s = edvr_conv2d_pack_weight_1x1s_f32's buffer and d.x_amax)
bool conv1x1_split_eligible(const edvr_conv2d_desc &d);
int conv1x1_split_launch(const edvr_conv2d_desc &d, hipStream_t stream);

// wgrad.hip: out[i] (+)= sum_k ws[k * total + i]
int reduce_partials_launch(const float *ws, float *out, int64_t total, int parts, int accumulate, hipStream_t stream,
                           const float *ws2 = nullptr, float *out2 = nullptr, int total2 = 0, int parts2 = 0);  // second, small array in the same launch

// winograd_wgrad.hip: weight gradient of the 3x3 / stride-1 conv in the Winograd domain.  plan() says whether the layer is
// eligible and how many split-K partial pairs it writes; the partials ([2*splits][co][ci][9]) are summed by wgrad_reduce_kernel.
bool winograd_wgrad_plan(int n, int c1, int c2, int h, int w, int co, int ks, int stride, int *splits);
size_t winograd_wgrad_ws_bytes(int co, int ci, int splits);
int winograd_wgrad_set_algo(int algo);
int winograd_wgrad_get_algo();
int winograd_wgrad_launch(const float *x1, const float *x2, const float *dz, float *ws, int c1, int c2, int n, int h, int w, int co,
                          int64_t x1_img_stride, int64_t x2_img_stride, int x2_div, int x2_mul, int x2_add, int64_t dz_img_stride,
                          int splits, int want_db, hipStream_t stream);  // want_db: also [splits][co] partial sums of dz after the dW partials

// winograd_wgrad_s.hip: the same kernel with split fp32 operands on the f16 matrix pipe (needs bounds of both tensors' magnitudes)
bool winograd_wgrad_split_enabled();
int winograd_wgrad_split_launch(const float *x1, const float *x2, const float *dz, float *ws, int c1, int c2, int n, int h, int w, int co,
                                int64_t x1_img_stride, int64_t x2_img_stride, int x2_div, int x2_mul, int x2_add, int64_t dz_img_stride,
                                int splits, int want_db, const float *x_amax, const float *dz_amax, hipStream_t stream);

// wgrad_direct_s.hip: the 3x3 / stride-1 weight gradient as a direct pixel-axis GEMM on split operands (no transforms: the staging
// streams of the Winograd-domain form are what bounded it); same partial format, same reduction
bool wgrad_direct_split_enabled();
bool wgrad_direct_split_supported(const float *x1, const float *x2, const float *dz, int h, int w, int64_t x1_img_stride, int64_t x2_img_stride,
                                  int64_t dz_img_stride);
int wgrad_direct_split_launch(const float *x1, const float *x2, const float *dz, float *ws, int c1, int c2, int n, int h, int w, int co,
                              int64_t x1_img_stride, int64_t x2_img_stride, int x2_div, int x2_mul, int x2_add, int64_t dz_img_stride,
                              int splits, int want_db, const float *x_amax, const float *dz_amax, hipStream_t stream);

// dcn.hip: geometry of one DCN call, shared with dcn_any.hip
struct DcnShape {
  int B, C, H, W, Co, kh, kw, stride, pad, dil, groups, dg, Ho, Wo;  // stride / pad / dil: along h
  int stride_w, pad_w, dil_w;                                        // along w (EDVR_HW pairs of the C ABI; equal to the h values otherwise)
  int64_t off_bs, msk_bs;    // image strides of offset / mask (inputs)
  int64_t doff_bs, dmsk_bs;  // image strides of doffset / dmask (backward outputs)
};
// validates the sizes, decodes EDVR_HW pairs, computes Ho / Wo and the default image strides (0 = contiguous)
int dcn_fill_shape(DcnShape &s, int B, int C, int H, int W, int Co, int kh, int kw, int stride, int pad, int dil, int groups, int dg,
                   int64_t off_bs, int64_t msk_bs);

// dcn_fused.hip: column-buffer-free DCNv2 forward for the EDVR signature (3x3, stride 1, pad 1, dil 1, groups 1)
bool dcn_fused_supported(int C, int Co, int H, int W, int kh, int kw, int stride, int pad, int dil, int groups, int dg);  // incl. the 32-bit buffer-offset limits
int dcn_fused_pack(const float *weight, float *wpk, int Co, int C, hipStream_t stream);  // (Co, C, 3, 3) -> the fused kernel's layout, C * 9 * round_up(Co, 32) floats
int dcn_fused_forward(const float *x, const float *offset, const float *mask, const float *wpk, const float *bias, float *y, int B, int C,
                      int H, int W, int Co, int dg, int64_t off_bs, int64_t msk_bs, int act, int halo, hipStream_t stream);

// dcn_tapwin.hip: the same operator with the staged window following each (group, tap)'s displacement: cost independent of the
// offset magnitude for spatially smooth fields.  Takes the weights in dcn_fused_pack's layout.
bool dcn_tapwin_supported(int C, int Co, int H, int W, int kh, int kw, int stride, int pad, int dil, int groups, int dg);
int dcn_tapwin_forward(const float *x, const float *offset, const float *mask, const float *wpk, const float *bias, float *y, int B, int C,
                       int H, int W, int Co, int dg, int64_t off_bs, int64_t msk_bs, int act, hipStream_t stream);

// dcn_tapwin_s.hip: the tap-window kernel with split fp32 operands on the f16 matrix pipe; packs its own weights into `wpk`
bool dcn_tapwin_split_enabled();
int dcn_tapwin_split_forward(const float *x, const float *offset, const float *mask, const float *weight, unsigned *wpk, const float *bias, float *y,
                             int B, int C, int H, int W, int Co, int dg, int64_t off_bs, int64_t msk_bs, int act, const float *xm_amax,
                             hipStream_t stream);

// dcn_bwd_fused.hip: DCNv2 backward (dX, dOffset, dMask, forward columns) for the EDVR signature without the dcol buffer
bool dcn_bwd_fused_supported(const DcnShape &s);
size_t dcn_bwd_fused_wbk_elems(int dg);  // floats of the re-ordered W^T the kernel streams (workspace)
int dcn_bwd_fused_launch(const DcnShape &s, const float *x, const float *offset, const float *mask, const float *weight, const float *dy,
                         float *wbk, float *col, float *dx, float *doffset, float *dmask, hipStream_t stream);

// XCD-aware workgroup order.  The dispatcher deals consecutive workgroups round-robin to the 8 XCDs (workgroup L runs on
// XCD L % 8), each with its own 4 MB L2, so spatially adjacent tiles - which share 128-byte lines and halo rows - land on
// eight different L2s and each re-fetches the shared lines from the fabric.  xcd_remap gives XCD x one contiguous range of
// the logical tile order instead (a bijection on [0, total)), so the tiles in flight on one L2 are neighbours.
__device__ inline int xcd_remap(int linear, int total) {
  const int q = total >> 3, rem = total & 7, xcd = linear & 7;
  return xcd * q + (xcd < rem ? xcd : rem) + (linear >> 3);
}
// (tile, y-block, image) of this workgroup of a dim3(tiles, yblocks, images) grid under the XCD-aware order, tile fastest
__device__ inline void xcd_block_index(int &bx, int &by, int &bz) {
  const int gx = gridDim.x, gy = gridDim.y;
  const int lin = blockIdx.x + gx * (blockIdx.y + gy * blockIdx.z);
  const int lg = xcd_remap(lin, gx * gy * (int)gridDim.z);
  bx = lg % gx;
  by = (lg / gx) % gy;
  bz = lg / (gx * gy);
}

}  // namespace edvr

#define EDVR_REQUIRE(cond, ...)       \
  do {                                \
    if (!(cond)) {                    \
      ::edvr::set_error(__VA_ARGS__); \
      return EDVR_ERR_ARG;            \
    }                                 \
  } while (0)
