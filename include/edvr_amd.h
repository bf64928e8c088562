/* edvr_amd.h - C ABI of libedvr_amd.so: the MI355X (gfx950) native EDVR hot path.
 *
 * Drop-in boundary.  These entry points are what the reference's FFI for this
 * path binds; each cites the reference interface it replaces (paths relative to
 * xinntao/EDVR = BasicSR v1.2.0):
 *
 *   edvr_dcnv2_fwd_f32  <- modulated_deform_conv_forward
 *                          basicsr/models/ops/dcn/src/deform_conv_ext.cpp:106-124
 *                          (driver deform_conv_cuda.cpp:490-569, kernel .cu:570-633)
 *   edvr_dcnv2_bwd_f32  <- modulated_deform_conv_backward
 *                          basicsr/models/ops/dcn/src/deform_conv_ext.cpp:126-147
 *                          (driver deform_conv_cuda.cpp:571-685, kernels .cu:635-767)
 *   edvr_conv2d_*       <- the at::conv2d / cuDNN calls under every nn.Conv2d of
 *                          basicsr/models/archs/edvr_arch.py (:37-66,139-155,230-244,322-353)
 *                          with the LeakyReLU/ReLU/residual/PixelShuffle/cat that follow them
 *                          (arch_util.py:92-95, edvr_arch.py:70,157,351,410-411) fused in.
 *   edvr_tsa_temporal_* <- TSAFusion.forward temporal attention, edvr_arch.py:171-184
 *   edvr_pool_*, edvr_upsample2x_*, edvr_tsa_combine_*, edvr_upsample4x_add_*
 *                       <- MaxPool2d/AvgPool2d(3,2,1), nn.Upsample(x2 bilinear) and the
 *                          elementwise tail of TSAFusion / EDVR.forward
 *                          (edvr_arch.py:144-145,158-159,190-213,414-419)
 *
 * Contract (all entry points):
 *   - every pointer is a DEVICE pointer to contiguous fp32 unless stated; NCHW;
 *   - the caller allocates everything (outputs, gradients, workspace), as the
 *     reference's Python does (deform_conv.py:138-140,154-158);
 *   - asynchronous on `stream` (a hipStream_t; NULL = default stream), no host
 *     synchronisation, no allocation, no global state: re-entrant across streams;
 *   - returns 0 or a negative EDVR_ERR_* code; edvr_last_error() gives the
 *     message for the calling thread.  Kernel launch failures are returned, not
 *     printf'd (the reference only prints: deform_conv_cuda_kernel.cu:794-798).
 *   - there is NO CPU implementation behind this ABI.
 */
#ifndef EDVR_AMD_H
#define EDVR_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define EDVR_OK 0
#define EDVR_ERR_ARG (-1)          /* bad shape / null pointer / mismatched sizes */
#define EDVR_ERR_UNSUPPORTED (-2)  /* parameter combination not implemented */
#define EDVR_ERR_LAUNCH (-3)       /* HIP launch error */
#define EDVR_ERR_WORKSPACE (-4)    /* workspace too small */

typedef void *edvr_stream_t; /* hipStream_t */

const char *edvr_version(void);
const char *edvr_last_error(void);
/* Device-capability probe: 0 if the current HIP device is gfx950, else EDVR_ERR_UNSUPPORTED. */
int edvr_check_device(void);

/* ------------------------------------------------------------------ activations */
#define EDVR_ACT_NONE 0
#define EDVR_ACT_RELU 1
#define EDVR_ACT_LRELU 2 /* negative slope 0.1 (edvr_arch.py:70,157,248,356) */
#define EDVR_ACT_SIGMOID 3

#define EDVR_DCN_HALO_TAPWIN 16 /* halo_hint of edvr_dcnv2_fwd_f32 / edvr_dcnv1_fwd_f32: per-tap shifted windows (below) */
#define EDVR_DCN_SCATTER_AUTO 0
#define EDVR_DCN_SCATTER_DEVICE 1
#define EDVR_DCN_SCATTER_LDS 2
#define EDVR_DCN_SCATTER_STRIP 3
#define EDVR_DCN_SCATTER_LDS_WIDE 4
#define EDVR_CONV_AUTO 0
#define EDVR_CONV_DIRECT 1
#define EDVR_CONV_WINOGRAD 2
#define EDVR_CONV_WINOGRAD_F4 3 /* F(4x4,3x3): needs edvr_conv2d_desc.wpk_f4; falls back like EDVR_CONV_WINOGRAD where it does not apply */
#define EDVR_CONV_WINOGRAD_F4S 4 /* F(4x4,3x3) with split fp32 operands on the f16 matrix pipe: needs wpk_f4s and x_amax; falls back to
                                  * EDVR_CONV_WINOGRAD_F4's chain where it does not apply */

#define EDVR_OUT_NCHW 0
#define EDVR_OUT_PIXEL_SHUFFLE2 1 /* y[n, co/4, 2h+(co%4)/2, 2w+co%2]  (nn.PixelShuffle(2)) */

/* ------------------------------------------------------------------ conv2d (fp32 MFMA implicit GEMM)
 * y = act(conv(cat(x1, x2), W) + bias) + res1 + res2, kernel ks in {1,3}, pad = ks/2,
 * stride in {1,2}, dilation 1, groups 1.  Weights come pre-packed by
 * edvr_conv2d_pack_weight_f32 (layout [ci_pad][ks*ks][co_pad], co fastest; for 3x3 kernels followed by the
 * Winograd-transformed weights G g G^T as [ci_pad][16][co_pad64]).  3x3 / stride-1 layers with >= 48 output channels
 * and w > 16 run as Winograd F(2x2,3x3) on the fp32 MFMA (2.25x fewer multiplies, fp32 throughout; set the
 * environment variable EDVR_CONV_WINOGRAD=0 to force the direct kernel); with `wpk_f4` present, layers with w >= 32 and
 * w % 4 == 0 run as Winograd F(4x4,3x3) (2.25 multiplies per output and channel pair where the direct algorithm needs 9 and
 * F(2x2) 4; EDVR_WINOGRAD_F4=0 switches it off). */
typedef struct edvr_conv2d_desc {
  const float *x1;        /* (n, c1, h, w) */
  const float *x2;        /* optional second input, concatenated after x1 on the channel axis */
  int c1, c2;             /* c2 = 0 when x2 == NULL */
  int64_t x1_img_stride;  /* elements between consecutive images of x1 (>= c1*h*w) */
  int64_t x2_img_stride;
  int x2_div, x2_mul, x2_add; /* image map of x2: i2 = (i / x2_div) * x2_mul + x2_add; x2_div = 0 -> i2 = i */
  int n, h, w;
  const float *wpk;       /* packed weights */
  const float *bias;      /* (co) or NULL */
  int co, ks, stride;
  int act;                /* EDVR_ACT_*; applied to output channels >= act_from only */
  int act_from;
  const float *res1;      /* optional residuals, same shape as the NCHW output (n, co, ho, wo) */
  const float *res2;
  int64_t res1_img_stride, res2_img_stride;
  float *y;
  int64_t y_img_stride;   /* elements between images of y */
  int out_mode;           /* EDVR_OUT_* */
  int algo;               /* EDVR_CONV_AUTO | EDVR_CONV_DIRECT | EDVR_CONV_WINOGRAD (F(2x2); falls back to direct where not applicable) |
                           * EDVR_CONV_WINOGRAD_F4 (needs wpk_f4; falls back to F(2x2), then direct) */
  const float *gate;      /* optional (n, co, ho, wo): y *= gate > 0 ? 1 : gate_slope, applied after bias / act.  This is the
                           * backward of a ReLU (slope 0) / LeakyReLU (0.1) fused into the data-gradient conv that produces its
                           * input gradient (gate = the activation's forward output).  3x3 kernels, NCHW output, no sigmoid
                           * (EDVR_ERR_UNSUPPORTED otherwise).  The Winograd kernels take it when there are no residuals (their
                           * fused-fast epilogue; edvr_conv2d_gate_supported asks for exactly that), every other 3x3 case - with
                           * residuals, stride 2, small layers - runs in the direct kernel's generic store path. */
  int64_t gate_img_stride;
  float gate_slope;
  float y_scale;          /* y = y_scale * act(conv + bias) [gated] + res1 + res2; 0 means 1.  ResidualBlockNoBN's res_scale
                           * (arch_util.py:95).  3x3 kernels with the NCHW output only (EDVR_ERR_UNSUPPORTED for 1x1 / PixelShuffle); a scaled
                           * conv with <= 4 output channels runs on the MFMA kernel instead of the small-co VALU kernel. */
  const float *wpk_f4;    /* optional: the same weights packed by edvr_conv2d_pack_weight_f4_f32.  Its presence ALLOWS the
                           * F(4x4,3x3) Winograd kernel (csrc/winograd_f4.hip: 2.25 instead of 4 multiplies per output; fp32
                           * rounding error ~1e-6 of the output scale instead of ~2e-7) under EDVR_CONV_AUTO where that kernel is the
                           * fastest; NULL keeps the F(2x2) / direct choice.  edvr_amd's inference AND training paths (forward and
                           * data-gradient convs; the weight gradient stays in the F(2x2) domain) pass it by default. */
  float *abs_sum;         /* optional (n): abs_sum[i] += sum |y[i, c, :, :]| over the output channels c < abs_sum_channels, added with
                           * atomics in the epilogue (the caller zeroes the array; summation order is not deterministic).  This is the
                           * statistic behind DCNv2Pack's "Offset abs mean is ..., larger than 50" check (arch_util.py:248-253) taken
                           * where conv_offset's output is still in registers instead of re-reading it (edvr_abs_sum_f32).  Only the
                           * F(4x4) kernel has this epilogue, in its NCHW store only (EDVR_OUT_PIXEL_SHUFFLE2 has none): ask
                           * edvr_conv2d_abs_sum_supported; EDVR_ERR_UNSUPPORTED otherwise.  The sum is taken over conv + bias (+ the
                           * activation of channels >= act_from), BEFORE gate, y_scale and the residuals are applied. */
  int abs_sum_channels;
  const void *wpk_f4s;    /* optional: the weights packed by edvr_conv2d_pack_weight_f4s_f32 (csrc/winograd_f4s.hip).  Together with
                           * `x_amax` it ALLOWS the split-operand F(4x4,3x3) kernel: every fp32 operand travels as two f16 numbers
                           * (hi + lo, 22 significant bits, after a power-of-two scaling), all four cross products are accumulated in
                           * fp32 by v_mfma_f32_32x32x16_f16 - the fp32 kernel's arithmetic to within its own rounding level at 1/4 of
                           * its matrix-pipe time.  EDVR_WINOGRAD_F4S=0 switches it off under EDVR_CONV_AUTO. */
  const float *x_amax;    /* device pointer to ONE float >= max |x1|, |x2| (any upper bound: edvr_amax_f32, or a statistic the
                           * producer of x already has).  Read by the split-operand kernel only, to place the transformed input in
                           * the f16 range; a bound that is too SMALL overflows to infinities, one that is 2^k too large costs
                           * accuracy only for elements below 2^-18 of it.  Two consequences a caller should know: operands carry 22
                           * significant bits (fp32: 24), and ONE bound covers the whole launch - the low bits of an image's output
                           * depend on which other images share the launch (within the kernel's 3e-5 tolerance; per-image calls with
                           * per-image bounds remove the coupling). */
  float *y_amax;          /* optional, split-operand kernel only (EDVR_ERR_UNSUPPORTED elsewhere: ask edvr_conv2d_y_amax_supported): y_amax[0] =
                           * max(y_amax[0], max |y|) over everything the launch stores (residuals / gate applied), with one atomic per
                           * wave at the end - the `x_amax` of the next conv of a chain for free.  The caller zeroes it (or folds bounds).
                           * The maximum is kept as a BIT PATTERN (non-negative floats order as integers, every NaN above +inf): a
                           * non-finite output is sticky in the slot - an overflow sentinel the host can read (edvr_amd/ops.py split_guard_*). */
} edvr_conv2d_desc;

size_t edvr_conv2d_packed_weight_elems(int co, int ci, int ks);
/* w: (co, ci, ks, ks) -> wpk.  transpose_flip != 0 packs the data-gradient kernel instead:
 * w'(ci, co, ks, ks) with w'[c][o][i][j] = w[o][c][ks-1-i][ks-1-j] (then `co`/`ci` refer to w'). */
int edvr_conv2d_pack_weight_f32(const float *w, float *wpk, int co, int ci, int ks, int transpose_flip,
                                edvr_stream_t stream);
/* F(4x4,3x3) weights of a 3x3 conv: U = G g G^T in MFMA operand order, edvr_conv2d_packed_weight_f4_elems(co, ci) floats.
 * Replaces: the algorithm / filter-transform choice cuDNN makes for the reference's 3x3 nn.Conv2d layers under
 * torch.backends.cudnn.benchmark = True (basicsr/train.py:132; edvr_arch.py:24-71,190-244,322-352, arch_util.py:86-95).
 * transpose_flip as in edvr_conv2d_pack_weight_f32 (data-gradient kernel). */
size_t edvr_conv2d_packed_weight_f4_elems(int co, int ci);
int edvr_conv2d_pack_weight_f4_f32(const float *w, float *wpk_f4, int co, int ci, int transpose_flip, edvr_stream_t stream);
/* The same U for the split-operand kernel (EDVR_CONV_WINOGRAD_F4S): a 64-byte header (s_U = the power of two that puts max |w| in
 * [2^14, 2^15), and 1 / s_U, taken from the weights by a one-workgroup pre-pass on the same stream) followed by one dword
 * (f16 hi | f16 lo << 16) of U * s_U per element in that kernel's operand order.  edvr_conv2d_packed_weight_f4s_elems(co, ci) dwords,
 * 16-byte aligned.  Replaces the same cuDNN choice as edvr_conv2d_pack_weight_f4_f32. */
size_t edvr_conv2d_packed_weight_f4s_elems(int co, int ci);
int edvr_conv2d_pack_weight_f4s_f32(const float *w, void *wpk_f4s, int co, int ci, int transpose_flip, edvr_stream_t stream);
/* The split-operand packing of a 1x1 conv's weights (csrc/conv1x1_s.hip): the same 64-byte header followed by [channel quad][co padded
 * to 128][4 channels] dwords (f16 hi | f16 lo << 16) of w * s_W.  Passed as edvr_conv2d_desc.wpk_f4s of a ks == 1 launch (with x_amax) it
 * allows the split form of the streaming 1x1 kernel (>= 320 input channels, c1 and c2 multiples of 8; EDVR_CONV1X1_SPLIT=0 switches
 * it off): the fp32 kernel's result to within fp32 rounding.  Replaces: the cuDNN call under TSAFusion.feat_fusion's nn.Conv2d
 * (edvr_arch.py:190-193,229).  edvr_conv2d_packed_weight_1x1s_elems(co, ci) dwords, 16-byte aligned. */
size_t edvr_conv2d_packed_weight_1x1s_elems(int co, int ci);
int edvr_conv2d_pack_weight_1x1s_f32(const float *w, void *wpk_1x1s, int co, int ci, edvr_stream_t stream);
/* amax[0] = max(amax[0], max |x|) over n images of per_img contiguous floats, img_stride elements apart (the caller zeroes amax
 * before the first call; several tensors may be folded into one bound).  Feeds edvr_conv2d_desc.x_amax. */
int edvr_amax_f32(const float *x, float *amax, int n, int64_t per_img, int64_t img_stride, edvr_stream_t stream);
/* Many weights in ONE launch (the training path repacks every conv weight after each optimizer step: ~480 tiny launches per
 * iteration otherwise).  `jobs`: DEVICE array of n_jobs records of edvr_pack_job_bytes() = 64 bytes { const float *w; float *wpk;
 * float *wpk_f4; int32 co, ci, ks, transpose_flip; int32 first_block, n_blocks; void *wpk_f4s; 8 bytes padding }: wpk / wpk_f4 /
 * wpk_f4s as the three functions above fill them (any may be NULL; the 16 header dwords of a wpk_f4s buffer must be zero before
 * its FIRST use here), transpose_flip as above; job j owns the workgroups [first_block, first_block +
 * n_blocks) of the launch (ascending, contiguous from 0; any n_blocks >= 1 - 256 elements per workgroup and pass), total_blocks =
 * their sum.  Results are bit-identical to the per-tensor functions.  split_parity: -1 when no job has a wpk_f4s; else a counter the
 * caller increments per call (its low bit selects which header slot collects max |w| this time: one extra launch, no memset). */
size_t edvr_pack_job_bytes(void);
int edvr_conv2d_pack_weights_multi(const void *jobs, int n_jobs, int total_blocks, int split_parity, edvr_stream_t stream);
int edvr_conv2d_f32(const edvr_conv2d_desc *d, edvr_stream_t stream);
/* 1 if `d` with a `gate` (and no residuals) would run in a Winograd kernel's fused epilogue (that kernel applies under d->algo,
 * the sizes and the EDVR_CONV_WINOGRAD environment switch), else 0 (edvr_conv2d_f32 still accepts the gate on the direct
 * kernel - correct, slower).  Callers that fuse an activation backward into a data-gradient conv ask first and keep the
 * two-launch form otherwise.  Pointers of `d` need not be set. */
int edvr_conv2d_gate_supported(const edvr_conv2d_desc *d);
/* 1 if edvr_conv2d_f32 would run `d` on the split-operand kernel, whose epilogue takes `y_amax`, else 0. */
int edvr_conv2d_y_amax_supported(const edvr_conv2d_desc *d);
/* 1 if edvr_conv2d_f32 would run `d` on a kernel whose epilogue takes `abs_sum` (the F(4x4) Winograd kernel), else 0. */
int edvr_conv2d_abs_sum_supported(const edvr_conv2d_desc *d);
/* Name of the kernel template instantiation edvr_conv2d_f32 would launch for `d` (as rocprofv3 prints it),
 * written to buf; returns 0 or EDVR_ERR_*.  Measurement aid only. */
int edvr_conv2d_kernel_name(const edvr_conv2d_desc *d, char *buf, size_t buf_len);
/* Flops the fp32 matrix cores EXECUTE for that launch: for the two Winograd kernels the number of v_mfma_f32_32x32x2_f32 issued
 * x 4096, tile and channel padding included (what rocprofv3's SQ_INSTS_VALU_MFMA_MOPS_F32 counts); for every other kernel the
 * algorithmic 2 * n * ho * wo * co * ci * ks^2.  Measurement aid only (bench.py's roofline fraction). */
int edvr_conv2d_executed_flops(const edvr_conv2d_desc *d, double *flops);

/* stride / pad / dil arguments of the DCN entry points below: a plain value n means (h, w) = (n, n); EDVR_HW(h, w) passes a
 * rectangular pair in one int (h <= 65535 in the low half, w + 1 in the high half).  The reference's DCNv1 Python takes pairs
 * (deform_conv.py:33-36,56-58 -> kW, kH, dW, dH, padW, padH, dilationW, dilationH of deform_conv_ext.cpp:51-104); its DCNv2 Python
 * only ever passes equal values (deform_conv.py:141-146).  Rectangular geometries run on the generic column-buffer kernels. */
#define EDVR_HW(h, w) ((int)(h) | (((int)(w) + 1) << 16))

/* ------------------------------------------------------------------ DCNv2 (modulated deformable conv)
 * x (B,C,H,W); offset (B, dg*2*kh*kw, Ho, Wo) channel = g*2K + 2k + {0:dy,1:dx};
 * mask (B, dg*kh*kw, Ho, Wo); weight (Co, C/groups, kh, kw); bias (Co) or NULL; y (B,Co,Ho,Wo).
 * offset_bstride / mask_bstride: elements between consecutive images of offset / mask
 * (0 = contiguous), so channel-sliced views of one conv_offset output can be passed without a copy.
 * act: EDVR_ACT_* applied to y in the GEMM epilogue (EDVR_ACT_NONE = the reference op; PCDAlignment
 * follows two of its four DCNs with LeakyReLU, edvr_arch.py:103-104,116).
 * halo_hint: performance hint only: every class computes the same operator, but their summation orders (and the tap-window class's
 * explicit fma chains) differ, so results agree to fp32 rounding (<= 2e-5 of the output scale, tests/test_gpu_dcn.py), not bit for bit.  The
 * Python layer picks the class from statistics of earlier calls that arrive asynchronously; EDVR_DCN_HINT_WAIT=1 makes that choice (and so
 * the bits) reproducible at the price of one host synchronisation per forward.  The EDVR signature (3x3, stride 1, pad 1,
 * dil 1, groups 1, (C/dg) % 8 == 0) runs a fused kernel that stages an input halo of R pixels around each tile in LDS
 * and falls back to global gathers for taps that leave it: 0 or 3 -> R = 3 (|offset| mostly < 3), 7 -> R = 7,
 * -1 -> skip the fused kernel (generic column-buffer path; always used for other signatures).
 * EDVR_DCN_HALO_TAPWIN (16) -> csrc/dcn_tapwin.hip: the walk is (group, tap, channel pair) and the staged window of every
 * (group, tap) is centred on that tap's displacement over the tile, so the cost does not depend on the offset MAGNITUDE as long as
 * the field is spatially smooth (+-2 px inside an 8 x 32 pixel tile) - what a trained conv_offset produces, and what the
 * reference's gather costs at any offset (.cu:570-633).  Needs W % 4 == 0, W >= 32, 8 or 16 channels per deformable group,
 * dg <= 8 and a 16-byte aligned x; otherwise the R = 7 kernel runs. */
size_t edvr_dcnv2_fwd_ws_bytes(int B, int C, int H, int W, int Co, int kh, int kw, int stride, int pad, int dil,
                               int groups, int dg);
int edvr_dcnv2_fwd_f32(const float *x, const float *offset, const float *mask, const float *weight,
                       const float *bias, float *y, int B, int C, int H, int W, int Co, int kh, int kw, int stride,
                       int pad, int dil, int groups, int dg, int64_t offset_bstride, int64_t mask_bstride, int act,
                       int halo_hint, void *ws, size_t ws_bytes, edvr_stream_t stream);

/* edvr_dcnv2_fwd_f32 with split fp32 operands on the f16 matrix pipe where the tap-window kernel applies (halo_hint EDVR_DCN_HALO_TAPWIN and
 * its constraints; csrc/dcn_tapwin_s.hip: weights and sampled columns as f16 (hi, lo) pairs, all four cross products, fp32 accumulation -
 * the fp32 kernel's result to within fp32 rounding at a quarter of its matrix-pipe time); identical to edvr_dcnv2_fwd_f32 everywhere
 * else.  xm_amax: device pointer to ONE float >= max |x| * max(1, max |mask|) (an upper bound; DCNv2Pack's masks are sigmoid outputs, so a
 * bound of |x| does).  edvr_dcnv2_fwd_split_applies: 1 if that kernel would run.  EDVR_DCN_SPLIT=0 switches it off.  Replaces the same
 * reference code as edvr_dcnv2_fwd_f32 (deform_conv_cuda.cpp:490-569, deform_conv_cuda_kernel.cu:570-633). */
int edvr_dcnv2_fwd_split_f32(const float *x, const float *offset, const float *mask, const float *weight, const float *bias,
                             float *y, int B, int C, int H, int W, int Co, int kh, int kw, int stride, int pad, int dil, int groups,
                             int dg, int64_t offset_bstride, int64_t mask_bstride, int act, int halo_hint, void *ws, size_t ws_bytes,
                             const float *xm_amax, edvr_stream_t stream);
int edvr_dcnv2_fwd_split_applies(const float *x, int B, int C, int H, int W, int Co, int kh, int kw, int stride, int pad, int dil, int groups,
                                 int dg, int halo_hint);
/* Name of the kernel edvr_dcnv2_fwd_f32 would launch for these arguments (as rocprofv3 prints it): dcn_tapwin_fwd_kernel,
 * dcn_fused_fwd_kernel or dcn_im2col_kernel (+ the conv kernel of its GEMM).  Measurement aid only. */
int edvr_dcnv2_fwd_kernel_name(const float *x, int B, int C, int H, int W, int Co, int kh, int kw, int stride, int pad, int dil,
                               int groups, int dg, int halo_hint, char *buf, size_t buf_len);

size_t edvr_dcnv2_bwd_ws_bytes(int B, int C, int H, int W, int Co, int kh, int kw, int stride, int pad, int dil,
                               int groups, int dg);
/* Gradients are OVERWRITTEN (no pre-zeroing needed).  dbias may be NULL.  dx accumulates by fp32
 * atomics (as the reference's col2im does, .cu:688), so its summation order is not deterministic. */
int edvr_dcnv2_bwd_f32(const float *x, const float *offset, const float *mask, const float *weight, const float *dy,
                       float *dx, float *doffset, float *dmask, float *dweight, float *dbias, int B, int C, int H,
                       int W, int Co, int kh, int kw, int stride, int pad, int dil, int groups, int dg,
                       int64_t offset_bstride, int64_t mask_bstride, int64_t doffset_bstride, int64_t dmask_bstride,
                       int scatter_hint, void *ws, size_t ws_bytes, edvr_stream_t stream);
/* The same with magnitude bounds: xm_amax / dy_amax = device pointers to ONE float >= max |x| * max(1, max |mask|) and >= max |dy|
 * (upper BOUNDS, as edvr_conv2d_desc.x_amax).  dW = sum dY col^T (deform_conv_cuda.cpp:664-672) then runs with both operands as f16
 * (hi, lo) pairs on the f16 matrix pipe (csrc/gemm_nt_s.hip): the fp32 result to within fp32 rounding.  Everything else is
 * edvr_dcnv2_bwd_f32.  edvr_dcnv2_bwd_split_applies: 1 unless EDVR_GEMM_SPLIT=0. */
int edvr_dcnv2_bwd_split_f32(const float *x, const float *offset, const float *mask, const float *weight, const float *dy,
                             float *dx, float *doffset, float *dmask, float *dweight, float *dbias, int B, int C, int H,
                             int W, int Co, int kh, int kw, int stride, int pad, int dil, int groups, int dg,
                             int64_t offset_bstride, int64_t mask_bstride, int64_t doffset_bstride, int64_t dmask_bstride,
                             int scatter_hint, void *ws, size_t ws_bytes, const float *xm_amax, const float *dy_amax,
                             edvr_stream_t stream);
int edvr_dcnv2_bwd_split_applies(void);
/* scatter_hint (performance only, like halo_hint of the forward; results are the same up to the summation order of dx):
 * how the four corner contributions per (pixel, tap, channel) are accumulated into dx.
 *   EDVR_DCN_SCATTER_DEVICE (1): fp32 device atomics straight to dx, as the reference's col2im (.cu:688).  Fastest when the
 *       offset field is smooth (neighbouring pixels hit neighbouring addresses: 6 ms on the EDVR-L training layer), collapses
 *       when it is not (89 ms with white-noise offsets of 1 px).
 *   EDVR_DCN_SCATTER_LDS (2): per-tile LDS window flushed with one device atomic per touched element: 7 ms /
 *       14 ms on the same two cases (round 6: the window's float additions are compare-and-swap loops - the hardware's ds_add_f32 runs
 *       at 0.4 lane-operations per cycle and CU on gfx950 -: 12.0 -> 8.8 ms per call on a trained-like field).  3x3, stride 1, pad 1,
 *       dil 1, <= 16 channels per deformable group; else DEVICE is used.  The window reaches 3 px beyond the tile's 3x3 footprint;
 *       EDVR_DCN_SCATTER_LDS_WIDE (4): 6 px, for fields whose taps sit ~4 px out and more (a corner outside the window is a device atomic).
 *   EDVR_DCN_SCATTER_STRIP (3): no scatter at all for sub-pixel offsets - one wave owns whole channel planes, folds the 9 taps of a
 *       pixel into a 5x5 register patch, the patch onto its owner lanes with DPP wave shifts and the rows into a register ring;
 *       one uncontended atomic per dx element.  Taps with |offset| >= 1 fall back to device atomics one by one, so this is the
 *       choice for fresh / lightly trained offset convs.  3x3, stride 1, pad 1, dil 1 (any size; one wave per 64-column strip); else DEVICE.
 *       With 16 channels per deformable group, Co <= 128 and W >= 32 this hint selects the kernel that needs no dcol buffer at
 *       all (csrc/dcn_bwd_fused.hip: W^T dY slices consumed in the registers the matrix core leaves them in; dx through
 *       wave-private LDS rows for sub-pixel taps, device atomics for the others): 6.4 vs 8.6 ms on the EDVR-L training layer.
 *   EDVR_DCN_SCATTER_AUTO (0): LDS where applicable.
 * doffset_bstride / dmask_bstride (0 = contiguous): image strides of the two gradient outputs, so both can be
 * written straight into channel slices of one (B, 3*dg*K, Ho, Wo) buffer = the gradient of conv_offset's output. */

/* The same operator in float64 / float16: the other legs of the reference's AT_DISPATCH_FLOATING_TYPES_AND_HALF
 * (deform_conv_cuda_kernel.cu:781,811,843).  All tensors of one call have the type `dtype` names; EDVR_DTYPE_F32 forwards to the
 * entry points above (no activation, default hints).  float64 computes in float64 throughout (torch.autograd.gradcheck users);
 * float16 keeps tensors in float16 and every intermediate in float32.  Plain VALU kernels (csrc/dcn_any.hip): any geometry, not
 * the measured path.  Gradients are overwritten; dbias may be NULL.  One workspace size serves both directions. */
#define EDVR_DTYPE_F32 0
#define EDVR_DTYPE_F64 1
#define EDVR_DTYPE_F16 2
size_t edvr_dcnv2_any_ws_bytes(int dtype, int B, int C, int H, int W, int Co, int kh, int kw, int stride, int pad, int dil, int groups,
                               int dg);
int edvr_dcnv2_fwd_any(int dtype, const void *x, const void *offset, const void *mask, const void *weight, const void *bias, void *y, int B,
                       int C, int H, int W, int Co, int kh, int kw, int stride, int pad, int dil, int groups, int dg,
                       int64_t offset_bstride, int64_t mask_bstride, void *ws, size_t ws_bytes, edvr_stream_t stream);
int edvr_dcnv2_bwd_any(int dtype, const void *x, const void *offset, const void *mask, const void *weight, const void *dy, void *dx,
                       void *doffset, void *dmask, void *dweight, void *dbias, int B, int C, int H, int W, int Co, int kh, int kw, int stride,
                       int pad, int dil, int groups, int dg, int64_t offset_bstride, int64_t mask_bstride, int64_t doffset_bstride,
                       int64_t dmask_bstride, void *ws, size_t ws_bytes, edvr_stream_t stream);

/* DCNv1 (DeformConv / DeformConvPack: no mask, no bias) <- deform_conv_forward, deform_conv_backward_input,
 * deform_conv_backward_parameters, basicsr/models/ops/dcn/src/deform_conv_ext.cpp:51-104 (drivers deform_conv_cuda.cpp:152-488,
 * kernels .cu:190-465).  Same offset layout and gather as DCNv2; gradients are overwritten; dx by fp32 atomics.
 * The reference's im2col_step only sizes its column buffer and has no counterpart. */
size_t edvr_dcnv1_fwd_ws_bytes(int B, int C, int H, int W, int Co, int kh, int kw, int stride, int pad, int dil,
                               int groups, int dg);
int edvr_dcnv1_fwd_f32(const float *x, const float *offset, const float *weight, float *y, int B, int C, int H, int W,
                       int Co, int kh, int kw, int stride, int pad, int dil, int groups, int dg, int64_t offset_bstride,
                       int halo_hint, void *ws, size_t ws_bytes, edvr_stream_t stream);
size_t edvr_dcnv1_bwd_ws_bytes(int B, int C, int H, int W, int Co, int kh, int kw, int stride, int pad, int dil,
                               int groups, int dg);
int edvr_dcnv1_bwd_f32(const float *x, const float *offset, const float *weight, const float *dy, float *dx,
                       float *doffset, float *dweight, int B, int C, int H, int W, int Co, int kh, int kw, int stride,
                       int pad, int dil, int groups, int dg, int64_t offset_bstride, int64_t doffset_bstride,
                       int scatter_hint, void *ws, size_t ws_bytes, edvr_stream_t stream);

/* ------------------------------------------------------------------ TSA / PCD glue kernels (HBM-bound) */
/* Temporal attention (edvr_arch.py:171-184): prob[b,t,p] = sigmoid(sum_c emb[b,t,c,p]*emb_ref[b,c,p]);
 * out[b,t,c,p] = aligned[b,t,c,p] * prob[b,t,p].  prob_out (b,t,hw) may be NULL. */
int edvr_tsa_temporal_f32(const float *emb, const float *emb_ref, const float *aligned, float *out, float *prob_out,
                          int b, int t, int c, int hw, edvr_stream_t stream);
/* MaxPool2d(3,2,1) and AvgPool2d(3,2,1, count_include_pad) in one pass; y (n, 2c, ho, wo) = cat(max, avg). */
int edvr_pool_maxavg_3x3s2_f32(const float *x, float *y, int n, int c, int h, int w, edvr_stream_t stream);
/* nn.Upsample(scale_factor=2, bilinear, align_corners=False); y = scale * up(x). */
int edvr_upsample2x_f32(const float *x, float *y, int nc, int h, int w, float scale, edvr_stream_t stream);
/* y = feat * sigmoid(attn) * 2 + attn_add  (edvr_arch.py:210-213). */
int edvr_tsa_combine_f32(const float *feat, const float *attn, const float *attn_add, float *y, int64_t numel,
                         edvr_stream_t stream);
/* y += bilinear x4 (align_corners=False) of base (n, c, h, w); y is (n, c, 4h, 4w)  (edvr_arch.py:417-419). */
int edvr_upsample4x_add_f32(const float *base, float *y, int nc, int h, int w, edvr_stream_t stream);
/* y = a + b */
int edvr_add_f32(const float *a, const float *b, float *y, int64_t numel, edvr_stream_t stream);
/* dz = dy * act'(.) expressed through the activation OUTPUT y (relu: y>0; lrelu: y>0 ? 1 : 0.1;
 * sigmoid: y(1-y)); channels < act_from of an (n, c, hw) tensor pass through unchanged. */
int edvr_act_bwd_f32(const float *dy, const float *y, const float *res1, const float *res2, float *dz, int n, int c,
                     int64_t hw, int act, int act_from, edvr_stream_t stream);
/* res1/res2 (nullable): the fused conv computed y = act(z) + res1 + res2; they are subtracted to recover act(z). */
/* out[i] = sum |x[i, :per_img]| for each of n images (image stride img_stride) - feeds the
 * "offset abs mean > 50" warning of arch_util.py:248-253 without a per-call host sync. */
int edvr_abs_sum_f32(const float *x, float *out, int n, int64_t per_img, int64_t img_stride, edvr_stream_t stream);
/* The same sums in out[0 .. n) plus, in out[n .. 2n), sum |x[i, c, r, col] - x[i, c, r, col + 1]| over the three neighbour pairs inside
 * every aligned group of four columns of the w-wide rows (3 of every 4 horizontal pairs; per_img % w == 0) - the ROUGHNESS of an
 * offset field: together with the sums it tells which DCN kernel suits the layer (a smooth field of any magnitude runs on the per-tap
 * windows, EDVR_DCN_HALO_TAPWIN).  (The F(4x4) conv epilogue only takes the plain sums, edvr_conv2d_desc.abs_sum: one more
 * accumulator in its staging waves spills - 128 registers - and costs the kernel 4 % on EVERY layer, measured.)
 * Rows that are not whole 16-byte groups (w % 4 != 0 or unaligned views): out[n .. 2n) = -1 (unknown). */
int edvr_abs_stats_f32(const float *x, float *out, int n, int64_t per_img, int w, int64_t img_stride, edvr_stream_t stream);

/* ------------------------------------------------------------------ training: gradients of the fused launches
 * (what autograd runs under SRModel.optimize_parameters, basicsr/models/sr_model.py:88-112) */
/* dW (co, c1+c2, ks, ks) (+)= sum_{n,pixel} dz[n,co,pixel] * cat(x1,x2)[n,ci,pixel*stride+tap-pad]; fp32 MFMA implicit GEMM,
 * deterministic split-K through `ws`.  x2 image map as in edvr_conv2d_desc. */
size_t edvr_conv2d_wgrad_ws_bytes(int n, int ci, int h, int w, int co, int ks, int stride);
int edvr_conv2d_wgrad_f32(const float *x1, const float *x2, const float *dz, float *dw, int c1, int c2, int n, int h, int w,
                          int co, int ks, int stride, int64_t x1_img_stride, int64_t x2_img_stride, int x2_div, int x2_mul,
                          int x2_add, int64_t dz_img_stride, int accumulate, float *dbias, void *ws, size_t ws_bytes,
                          edvr_stream_t stream);
/* The same with split fp32 operands on the f16 matrix pipe where the Winograd-domain kernel applies (both GEMM operands as f16 (hi, lo)
 * pairs, all four cross products, fp32 accumulation - the fp32 kernel's result to within its own rounding level): the Winograd-domain
 * form csrc/winograd_wgrad_s.hip; identical to edvr_conv2d_wgrad_f32 everywhere else.  Opt-in alternative (EDVR_WGRAD_DIRECT_SPLIT=1,
 * rows of 16-byte aligned tensors with w % 4 == 0; edvr_conv2d_wgrad_split_is_direct() says whether it would run): the DIRECT pixel-axis
 * GEMM csrc/wgrad_direct_s.hip - nothing is transformed, a value is split once and the nine taps are nine reads of the same LDS rows;
 * same accuracy, 10-15 % slower on MI355X (2.25x the matrix work meets the power-limited matrix clock, see that file).  x_amax / dz_amax: device pointers to
 * ONE float each, upper bounds of max |x1|, |x2| and of max |dz| (edvr_amax_f32 or a producer's statistic; too small = infinities).
 * edvr_conv2d_wgrad_split_applies: 1 if that kernel would run for this layer (callers skip computing the bounds otherwise).
 * EDVR_WGRAD_SPLIT=0 switches it off. */
int edvr_conv2d_wgrad_split_f32(const float *x1, const float *x2, const float *dz, float *dw, int c1, int c2, int n, int h, int w,
                                int co, int ks, int stride, int64_t x1_img_stride, int64_t x2_img_stride, int x2_div, int x2_mul,
                                int x2_add, int64_t dz_img_stride, int accumulate, float *dbias, void *ws, size_t ws_bytes,
                                const float *x_amax, const float *dz_amax, edvr_stream_t stream);
int edvr_conv2d_wgrad_split_applies(int n, int c1, int c2, int h, int w, int co, int ks, int stride);
int edvr_conv2d_wgrad_split_is_direct(int h, int w);
/* Name of the kernel edvr_conv2d_wgrad_f32 would launch for this layer (as rocprofv3 prints it).  Measurement aid only. */
int edvr_conv2d_wgrad_kernel_name(int n, int c1, int c2, int h, int w, int co, int ks, int stride, char *buf, size_t buf_len);
/* dbias (nullable): also db[co] = sum_{n,pixel} dz (the bias gradient).  The Winograd-domain kernel holds every dz value in
 * registers already and adds it to its two launches; the direct kernel runs edvr_channel_sum_f32 afterwards. */

/* Algorithm request for edvr_conv2d_wgrad_f32 (process-wide; tests and benchmarks use it to run both kernels on the same
 * layer): EDVR_CONV_AUTO (default: Winograd-domain kernel where eligible and profitable), EDVR_CONV_DIRECT, or
 * EDVR_CONV_WINOGRAD (wherever structurally possible: 3x3, stride 1, even h and w, c1 % 64 == 0 when x2 is given).
 * Returns the previous setting.  No reference counterpart (cuDNN picks its backward-filter algorithm internally). */
int edvr_conv2d_wgrad_algo(int algo);
/* out[c] = sum_{n,p} x[n,c,p]  (bias gradient); img_stride 0 = contiguous; ws: >= 64*c floats of scratch for the
 * deterministic two-stage reduction (NULL = single-stage, slower) */
int edvr_channel_sum_f32(const float *x, float *out, int n, int c, int64_t hw, int64_t img_stride, void *ws, size_t ws_bytes,
                         edvr_stream_t stream);
/* inverse of PixelShuffle(2): x (n, c, 2h, 2w) -> y (n, 4c, h, w) */
int edvr_pixel_unshuffle2_f32(const float *x, float *y, int n, int c, int h, int w, edvr_stream_t stream);
/* the same on dy * act'(y): dz (n, 4c, h, w) = unshuffle(dy (n, c, 2h, 2w) gated by the activation output y of the same shape) - the
 * gradient of PixelShuffle(act(conv(.))) w.r.t. the conv output in one pass (edvr_arch.py:403-404); y may be NULL with EDVR_ACT_NONE */
int edvr_pixel_unshuffle2_act_bwd_f32(const float *dy, const float *y, float *dz, int n, int c, int h, int w, int act, edvr_stream_t stream);
/* z (nc, H, W): z[2oy,2ox] = dz[oy,ox] (dz is (nc, ho, wo)), 0 elsewhere - the stride-2 data gradient is the
 * stride-1 transposed-kernel conv of z */
int edvr_zero_stuff2_f32(const float *dz, float *z, int nc, int H, int W, int ho, int wo, edvr_stream_t stream);
/* dst[b, center, :] += sum_t src[b, t, :]   (src, dst: (b, t, chw)) */
int edvr_frame_reduce_add_f32(const float *src, float *dst, int b, int t, int center, int64_t chw, edvr_stream_t stream);
int edvr_upsample2x_bwd_f32(const float *dy, float *dx, int nc, int h, int w, float scale, edvr_stream_t stream);
int edvr_pool_maxavg_3x3s2_bwd_f32(const float *x, const float *dy, float *dx, int n, int c, int h, int w, edvr_stream_t stream);
/* ws: b*t*hw floats of scratch (two fully parallel passes); NULL = the one-pass kernel, one thread per (clip, pixel) */
int edvr_tsa_temporal_bwd_f32(const float *emb, const float *emb_ref, const float *aligned, const float *dout, float *d_emb,
                              float *d_emb_ref, float *d_aligned, int b, int t, int c, int hw, float *ws, edvr_stream_t stream);
int edvr_tsa_combine_bwd_f32(const float *feat, const float *attn, const float *dy, float *dfeat, float *dattn, int64_t numel,
                             edvr_stream_t stream);
/* CharbonnierLoss, reduction 'sum' (basicsr/models/losses/losses.py:23-25): *loss = sum sqrt((p-t)^2 + eps) and, if dpred != NULL,
 * dpred = grad_scale * (p-t)/sqrt((p-t)^2+eps) in the same pass. */
int edvr_charbonnier_f32(const float *pred, const float *target, float *loss, float *dpred, int64_t numel, float eps,
                         float grad_scale, edvr_stream_t stream);

/* Validation PSNR on the device <- tensor2img (basicsr/utils/img_util.py:36-98) + calculate_psnr (basicsr/metrics/psnr_ssim.py:7-51),
 * which the reference runs per frame on the host after a D2H copy (video_base_model.py:60-98).  a, b: (n, c, h, w) fp32 in RGB
 * channel order, c = 3 or 1.  partial[img * blocks + k] receives the k-th partial sum of squared differences of the two
 * clamp[0,1]-x255-round uint8 images inside the crop (exact integers in double; sum them per image, divide by the element
 * count, PSNR = 20 log10(255 / sqrt(mse))).  y_channel != 0: differences of the Y channel as to_y_channel computes it
 * (metric_util.py:34-47; float32, not rounded), one value per pixel. */
int edvr_psnr_sse_f32(const float *a, const float *b, double *partial, int n, int c, int h, int w, int64_t a_img_stride,
                      int64_t b_img_stride, int crop_border, int y_channel, int blocks, edvr_stream_t stream);

/* Validation SSIM on the device <- tensor2img + calculate_ssim / _ssim (basicsr/metrics/psnr_ssim.py:54-141): per channel, 11x11
 * Gaussian window (sigma 1.5) over the clamp-round-uint8 images in float64, valid region only, mean of the SSIM map.  a, b as
 * for edvr_psnr_sse_f32.  partial[(img * channels + ch) * tiles + k], tiles = edvr_ssim_partials(h, w, crop_border), channels =
 * 1 when y_channel (and c == 3) else c, receives the k-th partial SUM of the SSIM map: add them, divide by
 * (h - 2 crop - 10) * (w - 2 crop - 10), average over channels.  Images without room for one window -> EDVR_ERR_ARG. */
size_t edvr_ssim_partials(int h, int w, int crop_border);
int edvr_ssim_f32(const float *a, const float *b, double *partial, int n, int c, int h, int w, int64_t a_img_stride,
                  int64_t b_img_stride, int crop_border, int y_channel, edvr_stream_t stream);

/* Input pipeline, device side <- imfrombytes(float32=True) (basicsr/utils/img_util.py:101-123: uint8 -> float32 / 255.), augment
 * (basicsr/data/transforms.py:84-151: hflip, then vflip, then transpose, the same state for every image of a clip), img2tensor
 * (img_util.py:9-33: BGR->RGB, HWC->CHW), default collate + CUDAPrefetcher's H2D copy (prefetch_dataloader.py:84-126).
 * src: DEVICE uint8 (n_clips, frames_per_clip, h, w, 3) - decoded and cropped on the host, 1 byte per sample over PCIe;
 * dst: DEVICE float32 (n_clips, frames_per_clip, 3, h', w'), (h', w') = (w, h) under EDVR_AUG_ROT90;
 * clip_flags: HOST array of n_clips bytes (OR of EDVR_AUG_*), or NULL for none (validation: read_img_seq, data_util.py:11-33);
 * it travels in the kernel-argument block, no device allocation or copy.  ROT90 with h != w -> EDVR_ERR_ARG.
 * swap_rb != 0: channel order of src is reversed on the way (src BGR as cv2 decodes -> RGB). */
#define EDVR_AUG_HFLIP 1
#define EDVR_AUG_VFLIP 2
#define EDVR_AUG_ROT90 4
int edvr_frames_u8_to_f32(const uint8_t *src, float *dst, int n_clips, int frames_per_clip, int h, int w,
                          const uint8_t *clip_flags, int swap_rb, edvr_stream_t stream);

/* Multi-tensor Adam step <- torch.optim.Adam.step() as the reference builds it (basicsr/models/edvr_model.py:21-53, parameter
 * groups with dcn_lr_mul; stepped in sr_model.py:112).  `chunk_table` is a DEVICE array of n_chunks records of
 * edvr_adam_chunk_bytes() = 64 bytes: { float *p; const float *g; float *m; float *v; int32 n (<= 65536 elements of one tensor);
 * float lr; float weight_decay; float 1/(1-beta1^t); float 1/sqrt(1-beta2^t); 12 bytes padding }.  One launch updates every
 * tensor of every group; arithmetic of torch.optim.Adam (amsgrad = False, maximize = False). */
size_t edvr_adam_chunk_bytes(void);
int edvr_adam_multi_f32(const void *chunk_table, int n_chunks, float beta1, float beta2, float eps, edvr_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* EDVR_AMD_H */
