// Standalone micro-benchmark (not part of the library): the layout of conv3x3_winograd_f4s_kernel - 16 waves per CU, waves 4-15 issue
// v_mfma_f32_32x32x16_f16 (12 per 8-channel chunk and wave = 36 per SIMD = 1152 matrix-pipe cycles), waves 0-3 issue N instructions of one
// kind per chunk, one barrier per chunk.  Question: does a vector instruction of the fourth wave still cost matrix-pipe
// time when the matrix instructions are f16 (on the fp32 MFMA it does: 5-7.6 cycles each, profiles/r4/micro_mfma_prices.log)?  And what
// does the staging wave's own stream cost when the matrix pipe is NOT the bottleneck?
//   hipcc --offload-arch=gfx950 -O3 scripts/micro/mfma16_prices.hip -o scripts/micro/mfma16_prices
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

enum { NONE, FMA8, PKFMA4, MIX, ALIGNBIT, DSW32, DSR128, MIXED };
// CONS: 0 = bare MFMAs; 1 = + the real loop's 2 v_alignbit per MFMA and 2 ds_read2_b32 per 2 MFMAs

template <int KIND, int N, int CONS, int NMFMA>
__global__ __launch_bounds__(1024, 1) void k(float *out, int chunks) {
  __shared__ __attribute__((aligned(16))) float lds[16 * 64 * 8];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  float *my = lds + (wave * 64 + lane) * 8;
  const unsigned my_a = (unsigned)(size_t)(__attribute__((address_space(3))) float *)my;
  if (wave < 4) {
    float x = lane, c = 1.0001f, d = 1e-3f;
    f32x2 pc = f32x2{1.0001f, 0.9999f}, pd = f32x2{1e-3f, -1e-3f};
    f32x2 q[4] = {f32x2{x, 1.f}, f32x2{x, 2.f}, f32x2{x, 3.f}, f32x2{x, 4.f}};
    float y8[8] = {x, x + 1, x + 2, x + 3, x + 4, x + 5, x + 6, x + 7};
    unsigned pk[8] = {0, 1, 2, 3, 4, 5, 6, 7};
    f32x4 v4 = f32x4{x, x, x, x};
    for (int ch = 0; ch < chunks; ++ch) {
#pragma unroll
      for (int i = 0; i < N; ++i) {
        if (KIND == FMA8) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(y8[i & 7]) : "v"(c), "v"(d));
        if (KIND == PKFMA4) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(q[i & 3]) : "v"(pc), "v"(pd));
        if (KIND == MIX) {  // the split: one mixlo + one mixhi per value (N counts instructions)
          if (i & 1) asm volatile("v_fma_mixhi_f16 %0, %1, %2, -%0 op_sel_hi:[0,0,1]" : "+v"(pk[(i >> 1) & 7]) : "v"(y8[(i >> 1) & 7]), "v"(c));
          else asm volatile("v_fma_mixlo_f16 %0, %1, %2, 0" : "=v"(pk[(i >> 1) & 7]) : "v"(y8[(i >> 1) & 7]), "v"(c));
        }
        if (KIND == ALIGNBIT) asm volatile("v_alignbit_b32 %0, %0, %0, 16" : "+v"(pk[i & 7]));
        if (KIND == DSW32) asm volatile("ds_write_b32 %0, %1" ::"v"(my_a), "v"(x) : "memory");
        if (KIND == DSR128) asm volatile("ds_read_b128 %0, %1" : "=v"(v4) : "v"(my_a) : "memory");
        if (KIND == MIXED) {  // the staging wave's real mix per patch, N = 192: 72 packed fp32 + 72 mix + 36 ds_write_b32 + 12 ds_read_b128 (interleaved 6:6:3:1)
          const int r = i % 16;
          if (r < 6) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(q[i & 3]) : "v"(pc), "v"(pd));
          else if (r < 12) {
            if (i & 1) asm volatile("v_fma_mixhi_f16 %0, %1, %2, -%0 op_sel_hi:[0,0,1]" : "+v"(pk[(i >> 1) & 7]) : "v"(y8[(i >> 1) & 7]), "v"(c));
            else asm volatile("v_fma_mixlo_f16 %0, %1, %2, 0" : "=v"(pk[(i >> 1) & 7]) : "v"(y8[(i >> 1) & 7]), "v"(c));
          } else if (r < 15) asm volatile("ds_write_b32 %0, %1" ::"v"(my_a), "v"(pk[i & 7]) : "memory");
          else asm volatile("ds_read_b128 %0, %1" : "=v"(v4) : "v"(my_a) : "memory");
        }
      }
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    }
    float s = x + v4[0] + v4[3];
#pragma unroll
    for (int i = 0; i < 4; ++i) s += q[i][0] + q[i][1];
#pragma unroll
    for (int i = 0; i < 8; ++i) s += y8[i] + pk[i];
    if (s == 12345.678f) out[threadIdx.x] = s;
  } else {
    f32x16 acc[6];
#pragma unroll
    for (int t = 0; t < 6; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    i32x4 A = {lane, lane + 1, lane + 2, lane + 3}, B = {lane, 2 * lane, 3 * lane, 4 * lane};
    for (int ch = 0; ch < chunks; ++ch) {
#pragma unroll
      for (int g = 0; g < NMFMA / 12; ++g) {
#pragma unroll
        for (int t = 0; t < 6; ++t) {
          if (CONS == 1) {
            long long b01, b23;
            asm volatile("ds_read2_b32 %0, %1 offset1:1" : "=v"(b01) : "v"(my_a) : "memory");
            asm volatile("ds_read2_b32 %0, %1 offset0:2 offset1:3" : "=v"(b23) : "v"(my_a) : "memory");
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(b01), "+v"(b23)::"memory");
            B[0] = (int)b01; B[1] = (int)(b01 >> 32); B[2] = (int)b23; B[3] = (int)(b23 >> 32);
          }
          acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, A), __builtin_bit_cast(f16x8, B), acc[t], 0, 0, 0);
          if (CONS == 1) {
#pragma unroll
            for (int i = 0; i < 4; ++i) A[i] = __builtin_amdgcn_alignbit(A[i], A[i], 16);
          }
          acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, A), __builtin_bit_cast(f16x8, B), acc[t], 0, 0, 0);
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    }
    float s = 0.f;
#pragma unroll
    for (int t = 0; t < 6; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) s += acc[t][r];
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
  const int chunks = 3000, wgs = 256;
  float base = 0;
#define RUN(KIND, N, CONS, NM, label)                                                                                         \
  {                                                                                                                          \
    const float ms = timed([&] { hipLaunchKernelGGL((k<KIND, N, CONS, NM>), dim3(wgs), dim3(1024), 0, 0, out, chunks); });    \
    const float ns = ms * 1e6f / chunks;                                                                                     \
    if (KIND == NONE) base = ns;                                                                                             \
    printf("%-64s %7.0f ns per chunk", label, ns);                                                                            \
    if (KIND != NONE && N > 0) printf("   %+6.2f ns per instruction vs the last baseline", (ns - base) / N);                  \
    printf("\n");                                                                                                            \
  }
  RUN(NONE, 0, 0, 12, "warm-up (discard)");
  RUN(NONE, 0, 0, 12, "multiplying waves only: 36 f16 MFMAs per SIMD and chunk");
  RUN(FMA8, 144, 0, 12, "+ 144 v_fma_f32 (8 chains) on the staging wave");
  RUN(PKFMA4, 144, 0, 12, "+ 144 v_pk_fma_f32 (4 chains)");
  RUN(PKFMA4, 72, 0, 12, "+ 72 v_pk_fma_f32 (4 chains)");
  RUN(MIX, 144, 0, 12, "+ 144 v_fma_mixlo/mixhi_f16 (72 splits)");
  RUN(ALIGNBIT, 144, 0, 12, "+ 144 v_alignbit_b32");
  RUN(DSW32, 36, 0, 12, "+ 36 ds_write_b32");
  RUN(DSR128, 12, 0, 12, "+ 12 ds_read_b128");
  RUN(MIXED, 192, 0, 12, "+ the staging mix: 72 pk + 72 mix + 36 ds_write + 12 ds_read");
  RUN(NONE, 0, 1, 12, "multiplying waves with their real loop (2 ds_read2 + 4 v_alignbit per MFMA pair)");
  RUN(MIXED, 192, 1, 12, "real loop + the staging mix");
  RUN(NONE, 0, 0, 0, "no MFMAs at all (barrier loop only)");
  RUN(MIXED, 192, 0, 0, "staging mix alone (no MFMAs): the lone wave's own rate");
  RUN(FMA8, 144, 0, 0, "144 v_fma_f32 alone");
  RUN(PKFMA4, 144, 0, 0, "144 v_pk_fma_f32 alone");
  RUN(MIX, 144, 0, 0, "144 mix alone");
  RUN(NONE, 0, 0, 24, "72 f16 MFMAs per SIMD and chunk (the pipe at twice the work)");
  RUN(MIXED, 192, 0, 24, "72 MFMAs + the staging mix");
  return 0;
}
