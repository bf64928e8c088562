// optim.hip - multi-tensor Adam step for gfx950.
//
// Replaces the torch.optim.Adam(...).step() of the reference's training loop (basicsr/models/edvr_model.py:47-49 builds it with
// the `dcn_lr_mul` parameter groups, sr_model.py:112 steps it).  EDVR-L has 20.6 M parameters in ~470 tensors: the stock
// foreach implementation walks them in ~10 elementwise passes; here ONE launch updates every tensor of every group, reading
// p, g, m, v once and writing p, m, v once (578 MB, HBM-bound).  Arithmetic = torch.optim.Adam (no amsgrad, no maximize):
//   g' = g + wd p;  m = b1 m + (1-b1) g';  v = b2 v + (1-b2) g'^2;  p -= lr / (1 - b1^t) * m / (sqrt(v) / sqrt(1 - b2^t) + eps)
#include "common.h"

namespace edvr {

// one entry per chunk of <= ADAM_CHUNK elements of one tensor; the table lives in device memory (uploaded by the host side)
struct AdamChunk {
  float *p;
  const float *g;
  float *m;
  float *v;
  int32_t n;      // elements in this chunk
  float lr;       // learning rate of the tensor's parameter group
  float wd;       // weight decay of the group
  float step_c1;  // 1 / (1 - b1^t) of the tensor's step count
  float rsq_c2;   // 1 / sqrt(1 - b2^t)
  float pad_[3];
};
static_assert(sizeof(AdamChunk) == 64, "AdamChunk is 64 bytes (the host packs it with this layout)");

__global__ __launch_bounds__(256) void adam_multi_kernel(const AdamChunk *__restrict__ table, float beta1, float beta2, float eps) {
  const AdamChunk c = table[blockIdx.x];
  for (int i = threadIdx.x; i < c.n; i += 256) {
    const float p = c.p[i];
    const float g = c.g[i] + c.wd * p;
    const float m0 = c.m[i];
    const float m = m0 + (1.f - beta1) * (g - m0);  // exp_avg.lerp_(grad, 1 - beta1), as torch.optim.Adam writes it
    const float v = beta2 * c.v[i] + (1.f - beta2) * g * g;
    c.m[i] = m;
    c.v[i] = v;
    c.p[i] = p - (c.lr * c.step_c1) * m / (sqrtf(v) * c.rsq_c2 + eps);
  }
}

}  // namespace edvr

extern "C" {

size_t edvr_adam_chunk_bytes(void) { return sizeof(edvr::AdamChunk); }

int edvr_adam_multi_f32(const void *chunk_table, int n_chunks, float beta1, float beta2, float eps, edvr_stream_t stream) {
  using namespace edvr;
  EDVR_REQUIRE(chunk_table && n_chunks > 0, "adam_multi: empty table");
  hipLaunchKernelGGL(adam_multi_kernel, dim3(n_chunks), dim3(256), 0, as_stream(stream), static_cast<const AdamChunk *>(chunk_table), beta1,
                     beta2, eps);
  return check_launch("adam_multi_kernel");
}

}  // extern "C"
