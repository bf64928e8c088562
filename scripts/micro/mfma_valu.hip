// Standalone micro-benchmark (not part of the library): what vector work costs NEXT TO fp32 MFMA waves on one SIMD, as a
// function of WHERE it runs - on a dedicated wave (the F(4x4) kernel's staging waves) or inside the waves that issue the MFMAs.
// Workgroup = 16 waves (4 per SIMD) like conv3x3_winograd_f4_kernel: waves 0-3 "staging", waves 4-15 "consumers" (24 back-to-back
// v_mfma_f32_32x32x2_f32 per chunk on 6 independent accumulator tiles), one barrier per chunk.
//   hipcc --offload-arch=gfx950 -O3 scripts/micro/mfma_valu.hip -o scripts/micro/mfma_valu && scripts/micro/mfma_valu
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

// MODE 0: consumers only (staging waves just keep the barrier)            -> the MFMA time
// MODE 1: staging waves run a DEPENDENT chain of NV packed FMAs per chunk  -> what the F(4x4) kernel does (transform arithmetic)
// MODE 2: staging waves run NV packed FMAs per chunk as 8 independent chains
// MODE 3: no staging work; every consumer runs NV / 3 packed FMAs per chunk (same total per SIMD), placed between its MFMAs
// MODE 4: like 1, plus NL LDS writes + NL LDS reads per chunk on the staging waves
// MODE 5: like 3, plus the LDS traffic of mode 4 split over the consumers
template <int MODE, int NV, int NL>
__global__ __launch_bounds__(1024, 1) void k(float *out, int chunks) {
  __shared__ float lds[16 * 64 * 8];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  float *my = lds + (wave * 64 + lane) * 8;
  f32x2 c = f32x2{1.0001f, 0.9999f}, dd = f32x2{1e-3f, -1e-3f};
  if (wave < 4) {
    f32x2 x[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) x[i] = f32x2{(float)lane + i, (float)i};
    for (int ch = 0; ch < chunks; ++ch) {
      if (MODE == 1 || MODE == 4) {
#pragma unroll
        for (int i = 0; i < NV; ++i) x[0] = x[0] * c + dd;
      }
      if (MODE == 2) {
#pragma unroll
        for (int i = 0; i < NV; ++i) x[i & 7] = x[i & 7] * c + dd;
      }
      if (MODE == 4) {
#pragma unroll
        for (int i = 0; i < NL; ++i) my[i & 7] = x[0][0] + i;
#pragma unroll
        for (int i = 0; i < NL; ++i) x[1][i & 1] += my[(i + 1) & 7];
      }
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += x[i][0] + x[i][1];
    if (s == 12345.678f) out[threadIdx.x] = s;
  } else {
    f32x16 acc[6];
#pragma unroll
    for (int t = 0; t < 6; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    float a = 1.f + lane * 1e-3f, b = 1.f - lane * 1e-3f;
    f32x2 x[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) x[i] = f32x2{(float)lane + i, (float)i};
    for (int ch = 0; ch < chunks; ++ch) {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
#pragma unroll
        for (int t = 0; t < 6; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[t], 0, 0, 0);
        if (MODE == 3 || MODE == 5) {  // a quarter of this wave's share after each group of 6 MFMAs: independent of them
#pragma unroll
          for (int i = 0; i < NV / 12; ++i) x[0] = x[0] * c + dd;
        }
        if (MODE == 5) {
#pragma unroll
          for (int i = 0; i < NL / 12; ++i) my[i & 7] = x[0][0] + i;
#pragma unroll
          for (int i = 0; i < NL / 12; ++i) x[1][i & 1] += my[(i + 1) & 7];
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    }
    float s = 0.f;
#pragma unroll
    for (int t = 0; t < 6; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) s += acc[t][r];
#pragma unroll
    for (int i = 0; i < 4; ++i) s += x[i][0] + x[i][1];
    if (s == 12345.678f) out[threadIdx.x] = s;
  }
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
  const int chunks = 2000, wgs = 256;
  const double mfma_cycles = 72.0 * 64.0;  // per SIMD and chunk: 3 consumers x 24 MFMAs x 64 cycles
#define RUN(MODE, NV, NL, label)                                                                                      \
  {                                                                                                                  \
    const float ms = timed([&] { hipLaunchKernelGGL((k<MODE, NV, NL>), dim3(wgs), dim3(1024), 0, 0, out, chunks); }); \
    printf("%-78s %7.3f ms  %6.0f ns per chunk (MFMA pipe alone at 2.4 GHz: %.0f ns)\n", label, ms, ms * 1e6 / chunks, mfma_cycles / 2.4); \
  }
  RUN(0, 0, 0, "consumers only");
  RUN(1, 72, 0, "staging wave: 72 dependent packed FMAs per chunk");
  RUN(1, 144, 0, "staging wave: 144 dependent packed FMAs per chunk");
  RUN(1, 288, 0, "staging wave: 288 dependent packed FMAs per chunk");
  RUN(2, 144, 0, "staging wave: 144 packed FMAs per chunk, 8 independent chains");
  RUN(2, 288, 0, "staging wave: 288 packed FMAs per chunk, 8 independent chains");
  RUN(3, 144, 0, "in the consumers: 3 x 48 packed FMAs per chunk between their MFMAs");
  RUN(3, 288, 0, "in the consumers: 3 x 96 packed FMAs per chunk between their MFMAs");
  RUN(4, 144, 24, "staging wave: 144 dependent packed FMAs + 24 LDS writes + 24 LDS reads");
  RUN(5, 144, 24, "in the consumers: the same, split three ways");
  return 0;
}
