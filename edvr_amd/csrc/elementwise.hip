// elementwise.hip - the HBM-bound glue of the EDVR hot path (gfx950).
//
// Replaces stock ATen kernels the reference calls around its convolutions
// (basicsr/models/archs/edvr_arch.py):
//   TSA temporal attention      :171-184  (T reductions + cat + sigmoid + expand().contiguous() + mul)
//   MaxPool2d/AvgPool2d(3,2,1)  :144-145,192-194,199-201  (+ the torch.cat of both)
//   nn.Upsample(x2, bilinear)   :68-69,109-110,158-159,204,208
//   feat*sigmoid(attn)*2+add    :210-213
//   F.interpolate(x4) + add     :417-419
// All of them are bandwidth-bound: lanes run along the contiguous pixel axis, 16 B per lane
// where the shape allows, one pass over each operand.
#include "common.h"

namespace edvr {

__device__ __forceinline__ float sigmoidf(float v) { return 1.f / (1.f + __expf(-v)); }

// ---- TSA temporal attention.  VEC pixels per thread (float4 when hw % 4 == 0).
template <int VEC>
__global__ __launch_bounds__(256) void tsa_temporal_kernel(const float *__restrict__ emb, const float *__restrict__ emb_ref,
                                                           const float *__restrict__ aligned, float *__restrict__ out,
                                                           float *__restrict__ prob_out, int b, int t, int c, int hw) {
  typedef float vec_t __attribute__((ext_vector_type(VEC)));
  const int hwv = hw / VEC;
  const int64_t total = (int64_t)b * t * hwv;
  for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
    const int pv = (int)(idx % hwv);
    const int ti = (int)((idx / hwv) % t);
    const int bi = (int)(idx / ((int64_t)hwv * t));
    const vec_t *e = reinterpret_cast<const vec_t *>(emb + ((int64_t)(bi * t + ti) * c) * hw) + pv;
    const vec_t *r = reinterpret_cast<const vec_t *>(emb_ref + ((int64_t)bi * c) * hw) + pv;
    vec_t dot = 0.f;
    for (int ch = 0; ch < c; ++ch) dot += e[(int64_t)ch * hwv] * r[(int64_t)ch * hwv];
    vec_t pr;
#pragma unroll
    for (int v = 0; v < VEC; ++v) pr[v] = sigmoidf(dot[v]);
    if (prob_out) reinterpret_cast<vec_t *>(prob_out + (int64_t)(bi * t + ti) * hw)[pv] = pr;
    const vec_t *a = reinterpret_cast<const vec_t *>(aligned + ((int64_t)(bi * t + ti) * c) * hw) + pv;
    vec_t *o = reinterpret_cast<vec_t *>(out + ((int64_t)(bi * t + ti) * c) * hw) + pv;
    for (int ch = 0; ch < c; ++ch) o[(int64_t)ch * hwv] = a[(int64_t)ch * hwv] * pr;
  }
}

// ---- fused 3x3/s2/p1 max + avg pooling -> cat(max, avg)
__global__ __launch_bounds__(256) void pool_maxavg_kernel(const float *__restrict__ x, float *__restrict__ y, int n, int c, int h, int w,
                                                          int ho, int wo) {
  const int64_t total = (int64_t)n * c * ho * wo;
  for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
    const int ox = (int)(idx % wo);
    const int oy = (int)((idx / wo) % ho);
    const int ch = (int)((idx / ((int64_t)wo * ho)) % c);
    const int ni = (int)(idx / ((int64_t)wo * ho * c));
    const float *src = x + ((int64_t)ni * c + ch) * h * w;
    float mx = -INFINITY, sm = 0.f;
#pragma unroll
    for (int dy = 0; dy < 3; ++dy) {
      const int iy = oy * 2 - 1 + dy;
      if (iy < 0 || iy >= h) continue;
#pragma unroll
      for (int dx = 0; dx < 3; ++dx) {
        const int ix = ox * 2 - 1 + dx;
        if (ix < 0 || ix >= w) continue;
        const float v = src[iy * w + ix];
        mx = fmaxf(mx, v);
        sm += v;
      }
    }
    const int64_t plane = (int64_t)ho * wo;
    float *dst = y + ((int64_t)ni * 2 * c) * plane + (int64_t)oy * wo + ox;
    dst[(int64_t)ch * plane] = mx;
    dst[(int64_t)(c + ch) * plane] = sm * (1.f / 9.f);  // count_include_pad=True: always / 9
  }
}

// ---- bilinear upsampling, align_corners=False: src = (dst + 0.5) / S - 0.5, clamped at 0
template <int S>
__device__ __forceinline__ void src_index(int dst, int in, int &i0, int &i1, float &l) {
  float s = ((float)dst + 0.5f) * (1.f / S) - 0.5f;
  if (s < 0.f) s = 0.f;
  i0 = (int)s;
  i1 = i0 + ((i0 < in - 1) ? 1 : 0);
  l = s - (float)i0;
}

// One bilinear sample with the rounding order written out (three fused multiply-adds on two products): every upsampling kernel
// below calls this, so that their results are bit-identical whichever one a shape / alignment selects - left to -ffp-contract the
// compiler picks which product of `a * b + c * d` goes into the fma kernel by kernel.
__device__ __forceinline__ float bilerp(float v00, float v01, float v10, float v11, float lx, float ly) {
  const float top = __builtin_fmaf(lx, v01, (1.f - lx) * v00);
  const float bot = __builtin_fmaf(lx, v11, (1.f - lx) * v10);
  return __builtin_fmaf(ly, bot, (1.f - ly) * top);
}

template <int S, bool ADD>
__global__ __launch_bounds__(256) void upsample_kernel(const float *__restrict__ x, float *__restrict__ y, int nc, int h, int w,
                                                       float scale) {
  const int ho = h * S, wo = w * S;
  const int64_t total = (int64_t)nc * ho * wo;
  for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
    const int ox = (int)(idx % wo);
    const int oy = (int)((idx / wo) % ho);
    const int64_t pl = idx / ((int64_t)wo * ho);
    int y0, y1, x0, x1;
    float ly, lx;
    src_index<S>(oy, h, y0, y1, ly);
    src_index<S>(ox, w, x0, x1, lx);
    const float *src = x + pl * h * w;
    const float v = bilerp(src[y0 * w + x0], src[y0 * w + x1], src[y1 * w + x0], src[y1 * w + x1], lx, ly);
    if (ADD)
      y[idx] += v;
    else
      y[idx] = v * scale;
  }
}

// x2 fast path (even input width, 8-byte aligned planes): one thread = input columns (2j, 2j + 1) of input row i -> the 2 x 4
// output block at (2i, 4j).  Twelve loads (rows i-1..i+1, columns 2j-1..2j+2, clamped) feed eight outputs that leave as two
// 16-byte stores; a wave writes two full 1 KB output rows segments.  The arithmetic per output element is the expression of
// upsample_kernel above with the same (i0, i1, l), so the results are bit-identical to it.  (The generic kernel - one scalar
// store and four gathers per output - ran at 1.4 TB/s on the PCD pyramid's offset / feature maps.)
__global__ __launch_bounds__(256) void upsample2x_block_kernel(const float *__restrict__ x, float *__restrict__ y, int nc, int h, int w,
                                                               float scale) {
  typedef float f32x4 __attribute__((ext_vector_type(4)));
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  const int wh = w >> 1, wo = 2 * w;
  const int64_t total = (int64_t)nc * h * wh;
  for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
    const int j = (int)(idx % wh);
    const int i = (int)((idx / wh) % h);
    const int64_t pl = idx / ((int64_t)wh * h);
    const float *src = x + pl * h * w;
    const int r0 = max(i - 1, 0), r2 = min(i + 1, h - 1);
    const int c0 = max(2 * j - 1, 0), c3 = min(2 * j + 2, w - 1);
    float p[3][4];  // p[a][b] = src[row i - 1 + a (clamped)][col 2j - 1 + b (clamped)]
    const int rows[3] = {r0, i, r2};
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      const f32x2 mid = *reinterpret_cast<const f32x2 *>(src + rows[a] * w + 2 * j);
      p[a][0] = src[rows[a] * w + c0];
      p[a][1] = mid[0];
      p[a][2] = mid[1];
      p[a][3] = src[rows[a] * w + c3];
    }
    float *dst = y + pl * (4 * (int64_t)h * w) + (int64_t)(2 * i) * wo + 4 * j;
    // src_index<2>: output 2i -> rows (i - 1, i), l = 0.75 (row 0: rows (0, 1), l = 0); output 2i + 1 -> rows (i, i + 1), l = 0.25
    // (i + 1 clamped: p[2] then repeats p[1]); the same along x.  Only the first output row / column has a border variant, so
    // the patch entries are picked with a handful of selects instead of a general 12-way choice per output.
    const bool top = i == 0, left = j == 0;
#pragma unroll
    for (int dy = 0; dy < 2; ++dy) {
      const float ly = dy ? 0.25f : (top ? 0.f : 0.75f);
      float ra[4], rb[4];  // the two source rows of this output row
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        ra[b] = (dy || top) ? p[1][b] : p[0][b];
        rb[b] = (dy || top) ? p[2][b] : p[1][b];
      }
      f32x4 o;
#pragma unroll
      for (int dx = 0; dx < 4; ++dx) {
        const float lx = dx == 0 ? (left ? 0.f : 0.75f) : (dx == 1 ? 0.25f : (dx == 2 ? 0.75f : 0.25f));
        float v00, v01, v10, v11;
        if (dx == 0) {
          v00 = left ? ra[1] : ra[0]; v01 = left ? ra[2] : ra[1];
          v10 = left ? rb[1] : rb[0]; v11 = left ? rb[2] : rb[1];
        } else if (dx == 3) {
          v00 = ra[2]; v01 = ra[3]; v10 = rb[2]; v11 = rb[3];
        } else {
          v00 = ra[1]; v01 = ra[2]; v10 = rb[1]; v11 = rb[2];
        }
        o[dx] = bilerp(v00, v01, v10, v11, lx, ly) * scale;
      }
      *reinterpret_cast<f32x4 *>(dst + (int64_t)dy * wo) = o;
    }
  }
}

// The same for rows that are whole 16-byte groups (w % 4 == 0): a thread owns FOUR source columns of one source row - three 16-byte
// loads (rows i - 1, i, i + 1), the columns left and right of its group come from the neighbouring lanes (they hold the adjacent
// groups of the same row; the first / last lane of a wave and the row ends load or clamp them) - and writes 2 x 8 outputs as four
// 16-byte stores: 0.75 loads per store instead of 4.5.  Same expression per output as above: bit-identical.
__global__ __launch_bounds__(256) void upsample2x_wide_kernel(const float *__restrict__ x, float *__restrict__ y, int nc, int h, int w, float scale) {
  typedef float f32x4 __attribute__((ext_vector_type(4)));
  const int wq = w >> 2, wo = 2 * w, lane = threadIdx.x & 63;
  const int64_t total = (int64_t)nc * h * wq;
  const int64_t rounded = (total + 255) / 256 * 256;
  for (int64_t base = (int64_t)blockIdx.x * 256; base < rounded; base += (int64_t)gridDim.x * 256) {
    const int64_t idx_raw = base + threadIdx.x;
    const bool active = idx_raw < total;
    const int64_t idx = active ? idx_raw : total - 1;  // (inactive lanes still take part in the lane exchanges)
    const int j = (int)(idx % wq);
    const int i = (int)((idx / wq) % h);
    const int64_t pl = idx / ((int64_t)wq * h);
    const float *src = x + pl * h * w;
    const int rows[3] = {max(i - 1, 0), i, min(i + 1, h - 1)};
    float p[3][6];  // p[a][b] = src[row i - 1 + a (clamped)][col 4j - 1 + b (clamped)]
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      const f32x4 m = *reinterpret_cast<const f32x4 *>(src + rows[a] * w + 4 * j);
      const float from_left = __shfl_up(m[3], 1, 64), from_right = __shfl_down(m[0], 1, 64);
      float l = j == 0 ? m[0] : from_left, r = j == wq - 1 ? m[3] : from_right;
      if (lane == 0 && j > 0) l = src[rows[a] * w + 4 * j - 1];
      if (lane == 63 && j < wq - 1) r = src[rows[a] * w + 4 * j + 4];
      p[a][0] = l; p[a][1] = m[0]; p[a][2] = m[1]; p[a][3] = m[2]; p[a][4] = m[3]; p[a][5] = r;
    }
    if (!active) continue;
    float *dst = y + pl * (4 * (int64_t)h * w) + (int64_t)(2 * i) * wo + 8 * j;
    const bool top = i == 0, left = j == 0;
#pragma unroll
    for (int dy = 0; dy < 2; ++dy) {
      const float ly = dy ? 0.25f : (top ? 0.f : 0.75f);
      float ra[6], rb[6];  // the two source rows of this output row
#pragma unroll
      for (int b = 0; b < 6; ++b) {
        ra[b] = (dy || top) ? p[1][b] : p[0][b];
        rb[b] = (dy || top) ? p[2][b] : p[1][b];
      }
#pragma unroll
      for (int g = 0; g < 2; ++g) {
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int k = 2 * g + (e >> 1);  // source column 4j + k; even outputs pair it with its left neighbour, odd ones with its right
          const bool first = left && k == 0 && (e & 1) == 0;  // output column 0: the border variant of src_index (l = 0 on columns 0, 1)
          const float lx = (e & 1) ? 0.25f : (first ? 0.f : 0.75f);
          const int b0 = (e & 1) ? k + 1 : (first ? 1 : k);
          const float v00 = first ? ra[1] : ra[b0], v01 = first ? ra[2] : ra[b0 + 1];
          const float v10 = first ? rb[1] : rb[b0], v11 = first ? rb[2] : rb[b0 + 1];
          o[e] = bilerp(v00, v01, v10, v11, lx, ly) * scale;
        }
        *reinterpret_cast<f32x4 *>(dst + (int64_t)dy * wo + 4 * g) = o;
      }
    }
  }
}

__global__ __launch_bounds__(256) void tsa_combine_kernel(const float *__restrict__ feat, const float *__restrict__ attn,
                                                          const float *__restrict__ attn_add, float *__restrict__ y, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256)
    y[i] = feat[i] * sigmoidf(attn[i]) * 2.f + attn_add[i];
}

__global__ __launch_bounds__(256) void add_kernel(const float *__restrict__ a, const float *__restrict__ b, float *__restrict__ y, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) y[i] = a[i] + b[i];
}

__global__ __launch_bounds__(256) void act_bwd_kernel(const float *__restrict__ dy, const float *__restrict__ y, const float *__restrict__ res1,
                                                      const float *__restrict__ res2, float *__restrict__ dz, int64_t total, int c,
                                                      int64_t hw, int act, int act_from) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int ch = (int)((i / hw) % c);
    float g = dy[i];
    if (ch >= act_from) {
      float v = y[i];
      if (res1) v -= res1[i];  // y = act(z) + res: recover act(z)
      if (res2) v -= res2[i];
      if (act == EDVR_ACT_RELU) g = v > 0.f ? g : 0.f;
      else if (act == EDVR_ACT_LRELU) g = v > 0.f ? g : 0.1f * g;
      else if (act == EDVR_ACT_SIGMOID) g = g * v * (1.f - v);
    }
    dz[i] = g;
  }
}

// out[img] += sum |x|; rough (optional, vec4 only) [img] += sum |x[i] - x[i + 1]| over the three neighbour pairs inside every aligned
// group of four elements (rows are multiples of 4 wide there: 3 of every 4 horizontal neighbour pairs, never across a row end)
template <bool ROUGH>
__global__ __launch_bounds__(256) void abs_sum_kernel(const float *__restrict__ x, float *__restrict__ out, float *__restrict__ rough,
                                                      int64_t per_img, int64_t img_stride, int vec4) {
  const int img = blockIdx.y;
  const float *src = x + (int64_t)img * img_stride;
  float s = 0.f, r = 0.f;
  auto diff4 = [](const float4 &a) { return (fabsf(a.x - a.y) + fabsf(a.y - a.z)) + fabsf(a.z - a.w); };
  if (vec4) {  // 16-byte loads, two in flight per thread (the scalar loop ran at 2.4 TB/s)
    const float4 *s4 = reinterpret_cast<const float4 *>(src);
    const int64_t n4 = per_img >> 2, step = (int64_t)gridDim.x * 256;
    float s2 = 0.f;
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    for (; i + step < n4; i += 2 * step) {
      const float4 a = s4[i], b = s4[i + step];
      s += (fabsf(a.x) + fabsf(a.y)) + (fabsf(a.z) + fabsf(a.w));
      s2 += (fabsf(b.x) + fabsf(b.y)) + (fabsf(b.z) + fabsf(b.w));
      if (ROUGH) r += diff4(a) + diff4(b);
    }
    if (i < n4) {
      const float4 a = s4[i];
      s += (fabsf(a.x) + fabsf(a.y)) + (fabsf(a.z) + fabsf(a.w));
      if (ROUGH) r += diff4(a);
    }
    s += s2;
  } else {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < per_img; i += (int64_t)gridDim.x * 256) s += fabsf(src[i]);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    s += __shfl_down(s, o, 64);
    if (ROUGH) r += __shfl_down(r, o, 64);
  }
  __shared__ float red[8];
  if ((threadIdx.x & 63) == 0) {
    red[threadIdx.x >> 6] = s;
    red[4 + (threadIdx.x >> 6)] = r;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    unsafeAtomicAdd(out + img, red[0] + red[1] + red[2] + red[3]);
    if (ROUGH) unsafeAtomicAdd(rough + img, red[4] + red[5] + red[6] + red[7]);
  }
}

__global__ void fill_f32_kernel(float *p, float v, int n) {
  for (int i = threadIdx.x; i < n; i += blockDim.x) p[i] = v;
}

static inline unsigned grid_for(int64_t total) { return (unsigned)std::max<int64_t>(1, std::min<int64_t>(cdiv64(total, 256), 256 * 16)); }

}  // namespace edvr

extern "C" {

int edvr_tsa_temporal_f32(const float *emb, const float *emb_ref, const float *aligned, float *out, float *prob_out, int b, int t,
                          int c, int hw, edvr_stream_t stream) {
  using namespace edvr;
  EDVR_REQUIRE(emb && emb_ref && aligned && out && b > 0 && t > 0 && c > 0 && hw > 0, "tsa_temporal: bad arguments");
  if (hw % 4 == 0) {
    hipLaunchKernelGGL(tsa_temporal_kernel<4>, dim3(grid_for((int64_t)b * t * (hw / 4))), dim3(256), 0, as_stream(stream), emb, emb_ref,
                       aligned, out, prob_out, b, t, c, hw);
  } else {
    hipLaunchKernelGGL(tsa_temporal_kernel<1>, dim3(grid_for((int64_t)b * t * hw)), dim3(256), 0, as_stream(stream), emb, emb_ref,
                       aligned, out, prob_out, b, t, c, hw);
  }
  return check_launch("tsa_temporal_kernel");
}

int edvr_pool_maxavg_3x3s2_f32(const float *x, float *y, int n, int c, int h, int w, edvr_stream_t stream) {
  using namespace edvr;
  EDVR_REQUIRE(x && y && n > 0 && c > 0 && h > 0 && w > 0, "pool_maxavg: bad arguments");
  const int ho = (h - 1) / 2 + 1, wo = (w - 1) / 2 + 1;
  hipLaunchKernelGGL(pool_maxavg_kernel, dim3(grid_for((int64_t)n * c * ho * wo)), dim3(256), 0, as_stream(stream), x, y, n, c, h, w, ho,
                     wo);
  return check_launch("pool_maxavg_kernel");
}

int edvr_upsample2x_f32(const float *x, float *y, int nc, int h, int w, float scale, edvr_stream_t stream) {
  using namespace edvr;
  EDVR_REQUIRE(x && y && nc > 0 && h > 0 && w > 0, "upsample2x: bad arguments");
  if ((w & 3) == 0 && w >= 8 && (reinterpret_cast<uintptr_t>(x) & 15) == 0 && (reinterpret_cast<uintptr_t>(y) & 15) == 0) {
    hipLaunchKernelGGL(upsample2x_wide_kernel, dim3(grid_for((int64_t)nc * h * (w / 4))), dim3(256), 0, as_stream(stream), x, y, nc, h, w, scale);
    return check_launch("upsample2x_wide_kernel");
  }
  if ((w & 1) == 0 && (((int64_t)h * w) & 1) == 0 && (reinterpret_cast<uintptr_t>(x) & 7) == 0 && (reinterpret_cast<uintptr_t>(y) & 15) == 0) {
    hipLaunchKernelGGL(upsample2x_block_kernel, dim3(grid_for((int64_t)nc * h * (w / 2))), dim3(256), 0, as_stream(stream), x, y, nc, h, w, scale);
    return check_launch("upsample2x_block_kernel");
  }
  hipLaunchKernelGGL((upsample_kernel<2, false>), dim3(grid_for((int64_t)nc * h * w * 4)), dim3(256), 0, as_stream(stream), x, y, nc, h, w,
                     scale);
  return check_launch("upsample_kernel<2>");
}

int edvr_upsample4x_add_f32(const float *base, float *y, int nc, int h, int w, edvr_stream_t stream) {
  using namespace edvr;
  EDVR_REQUIRE(base && y && nc > 0 && h > 0 && w > 0, "upsample4x_add: bad arguments");
  hipLaunchKernelGGL((upsample_kernel<4, true>), dim3(grid_for((int64_t)nc * h * w * 16)), dim3(256), 0, as_stream(stream), base, y, nc, h,
                     w, 1.f);
  return check_launch("upsample_kernel<4>");
}

int edvr_tsa_combine_f32(const float *feat, const float *attn, const float *attn_add, float *y, int64_t numel, edvr_stream_t stream) {
  using namespace edvr;
  EDVR_REQUIRE(feat && attn && attn_add && y && numel > 0, "tsa_combine: bad arguments");
  hipLaunchKernelGGL(tsa_combine_kernel, dim3(grid_for(numel)), dim3(256), 0, as_stream(stream), feat, attn, attn_add, y, numel);
  return check_launch("tsa_combine_kernel");
}

int edvr_add_f32(const float *a, const float *b, float *y, int64_t numel, edvr_stream_t stream) {
  using namespace edvr;
  EDVR_REQUIRE(a && b && y && numel > 0, "add: bad arguments");
  hipLaunchKernelGGL(add_kernel, dim3(grid_for(numel)), dim3(256), 0, as_stream(stream), a, b, y, numel);
  return check_launch("add_kernel");
}

int edvr_act_bwd_f32(const float *dy, const float *y, const float *res1, const float *res2, float *dz, int n, int c, int64_t hw, int act,
                     int act_from, edvr_stream_t stream) {
  using namespace edvr;
  EDVR_REQUIRE(dy && y && dz && n > 0 && c > 0 && hw > 0, "act_bwd: bad arguments");
  const int64_t total = (int64_t)n * c * hw;
  hipLaunchKernelGGL(act_bwd_kernel, dim3(grid_for(total)), dim3(256), 0, as_stream(stream), dy, y, res1, res2, dz, total, c, hw, act, act_from);
  return check_launch("act_bwd_kernel");
}

int edvr_abs_sum_f32(const float *x, float *out, int n, int64_t per_img, int64_t img_stride, edvr_stream_t stream) {
  using namespace edvr;
  EDVR_REQUIRE(x && out && n > 0 && per_img > 0, "abs_sum: bad arguments");
  if (hipMemsetAsync(out, 0, sizeof(float) * n, as_stream(stream)) != hipSuccess) {
    set_error("abs_sum: hipMemsetAsync failed");
    return EDVR_ERR_LAUNCH;
  }
  const unsigned gx = (unsigned)std::max<int64_t>(1, std::min<int64_t>(cdiv64(per_img, 256 * 8), 512));
  const int vec4 = (per_img & 3) == 0 && (img_stride & 3) == 0 && (reinterpret_cast<uintptr_t>(x) & 15) == 0;
  hipLaunchKernelGGL(abs_sum_kernel<false>, dim3(gx, n), dim3(256), 0, as_stream(stream), x, out, static_cast<float *>(nullptr), per_img, img_stride, vec4);
  return check_launch("abs_sum_kernel");
}

int edvr_abs_stats_f32(const float *x, float *out, int n, int64_t per_img, int w, int64_t img_stride, edvr_stream_t stream) {
  using namespace edvr;
  EDVR_REQUIRE(x && out && n > 0 && per_img > 0 && w > 0 && per_img % w == 0, "abs_stats: bad arguments");
  if (hipMemsetAsync(out, 0, sizeof(float) * 2 * n, as_stream(stream)) != hipSuccess) {
    set_error("abs_stats: hipMemsetAsync failed");
    return EDVR_ERR_LAUNCH;
  }
  const unsigned gx = (unsigned)std::max<int64_t>(1, std::min<int64_t>(cdiv64(per_img, 256 * 8), 512));
  const int vec4 = (w & 3) == 0 && (img_stride & 3) == 0 && (reinterpret_cast<uintptr_t>(x) & 15) == 0;
  if (vec4) {
    hipLaunchKernelGGL(abs_sum_kernel<true>, dim3(gx, n), dim3(256), 0, as_stream(stream), x, out, out + n, per_img, img_stride, 1);
  } else {  // rows that are not whole 16-byte groups: the sums only, the differences are reported as unknown (-1)
    hipLaunchKernelGGL(abs_sum_kernel<false>, dim3(gx, n), dim3(256), 0, as_stream(stream), x, out, static_cast<float *>(nullptr), per_img, img_stride, 0);
    hipLaunchKernelGGL(fill_f32_kernel, dim3(1), dim3(256), 0, as_stream(stream), out + n, -1.f, n);
  }
  return check_launch("abs_sum_kernel");
}

}  // extern "C"
