// Standalone micro-benchmark (not part of the library): what the chunk boundary of a lockstep workgroup costs - all MFMA waves of a
// CU pass a barrier and then need an LDS operand before their first MFMA (the structure of conv3x3_winograd_wgrad_kernel and of the
// F(4x4) consumers).  16 waves per workgroup (4 per SIMD), 24 (or 48 / 96) back-to-back v_mfma_f32_32x32x2_f32 per wave and chunk.
//   hipcc --offload-arch=gfx950 -O3 scripts/micro/mfma_boundary.hip -o scripts/micro/mfma_boundary
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// MODE 0: no barrier at all; 1: barrier per chunk; 2: barrier + one dependent ds_read_b128 (operand of the first MFMA) after it;
// 3: like 2, and a ds_write_b128 + lgkmcnt(0) in front of the barrier (the commit of the next stage)
template <int MODE, int NM>
__global__ __launch_bounds__(1024, 1) void k(float *out, int chunks) {
  __shared__ __attribute__((aligned(16))) float lds[16 * 64 * 4];
  const int lane = threadIdx.x & 63;
  f32x4 *my = reinterpret_cast<f32x4 *>(lds) + threadIdx.x;
  *my = f32x4{1.f, 1.f, 1.f, 1.f};
  __syncthreads();
  f32x16 acc[6];
#pragma unroll
  for (int t = 0; t < 6; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
  float a = 1.f + lane * 1e-3f;
  const float b = 1.f - lane * 1e-3f;
  for (int ch = 0; ch < chunks; ++ch) {
    if (MODE >= 2) {
      const f32x4 v = *reinterpret_cast<volatile f32x4 *>(my);
      a = v[0] + lane * 1e-3f;  // the first MFMA depends on the LDS read
    }
#pragma unroll
    for (int g = 0; g < NM / 6; ++g)
#pragma unroll
      for (int t = 0; t < 6; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[t], 0, 0, 0);
    if (MODE == 3) *reinterpret_cast<volatile f32x4 *>(my) = f32x4{1.f, a, b, 1.f};
    if (MODE >= 1) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
  }
  float s = 0.f;
#pragma unroll
  for (int t = 0; t < 6; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) s += acc[t][r];
  if (s == 12345.678f) out[threadIdx.x] = s;
}

template <typename F>
float timed(F launch) {
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  launch();
  CHECK(hipDeviceSynchronize());
  CHECK(hipEventRecord(e0));
  for (int i = 0; i < 3; ++i) launch();
  CHECK(hipEventRecord(e1));
  CHECK(hipEventSynchronize(e1));
  float ms;
  CHECK(hipEventElapsedTime(&ms, e0, e1));
  return ms / 3;
}

int main() {
  float *out;
  CHECK(hipMalloc(&out, 4096));
#define RUN(MODE, NM, CH, label)                                                                              \
  {                                                                                                          \
    const float ms = timed([&] { hipLaunchKernelGGL((k<MODE, NM>), dim3(256), dim3(1024), 0, 0, out, CH); }); \
    printf("%-86s %8.1f ns per chunk = %5.1f ns per MFMA of a SIMD\n", label, ms * 1e6 / (CH), ms * 1e6 / (CH) / (4.0 * NM)); \
  }
  RUN(0, 24, 2000, "warm-up (discard)");
  RUN(0, 24, 2000, "4 waves per SIMD x 24 MFMAs per chunk, no barrier");
  RUN(1, 24, 2000, "4 x 24, barrier per chunk");
  RUN(2, 24, 2000, "4 x 24, barrier + dependent ds_read_b128 after it");
  RUN(3, 24, 2000, "4 x 24, ds_write_b128 + barrier + dependent ds_read_b128");
  RUN(3, 12, 4000, "4 x 12 (half-size chunks), ds_write_b128 + barrier + dependent ds_read_b128");
  RUN(3, 48, 1000, "4 x 48 (double-size chunks), ds_write_b128 + barrier + dependent ds_read_b128");
  return 0;
}
