// Standalone micro-benchmark (not part of the library): the price, in MFMA-pipe time, of ONE instruction of each kind issued by a
// staging wave that shares its SIMD with three waves of back-to-back v_mfma_f32_32x32x2_f32 (the layout of conv3x3_winograd_f4_kernel),
// and whether accumulators in AGPRs change it.   hipcc --offload-arch=gfx950 -O3 scripts/micro/mfma_prices.hip -o scripts/micro/mfma_prices
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

enum { NONE, FMA, PKFMA, PKFMA_IND, MOV, IADD, SALU, DSW32, DSW128, DSR128, DSR32, VMEM_LD, EXP, FMA_IND8, DPP, DSADD, VMEM_LAG, DMA16, VMEM_ST };

template <int KIND, int N, bool AGPR>
__global__ __launch_bounds__(1024, 1) void k(float *out, const float *in, int chunks) {
  __shared__ __attribute__((aligned(16))) float lds[16 * 64 * 8];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  if (wave < 4) {
    float *my = lds + (wave * 64 + lane) * 8;
    float x = lane, c = 1.0001f, d = 1e-3f;
    f32x2 p = f32x2{(float)lane, 1.f}, pc = f32x2{1.0001f, 0.9999f}, pd = f32x2{1e-3f, -1e-3f};
    f32x2 q[4] = {p, p + 1.f, p + 2.f, p + 3.f};
    int iv = lane, sv = __builtin_amdgcn_readfirstlane(wave);
    f32x4 v4 = f32x4{x, x, x, x};
    float y8[8] = {x, x + 1, x + 2, x + 3, x + 4, x + 5, x + 6, x + 7};
    typedef int i32x4 __attribute__((ext_vector_type(4)));
    i32x4 rs;
    {
      const unsigned long long pv = (unsigned long long)in;
      rs[0] = __builtin_amdgcn_readfirstlane((int)(unsigned)pv);
      rs[1] = __builtin_amdgcn_readfirstlane((int)(pv >> 32)) & 0xffff;
      rs[2] = 16384;
      rs[3] = 0x00020000;
    }
    const unsigned ldsb = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(size_t)(__attribute__((address_space(3))) float *)(lds + wave * 64 * 8));
    for (int ch = 0; ch < chunks; ++ch) {
#pragma unroll
      for (int i = 0; i < N; ++i) {
        if (KIND == FMA) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x) : "v"(c), "v"(d));
        if (KIND == PKFMA) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p) : "v"(pc), "v"(pd));
        if (KIND == PKFMA_IND) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(q[i & 3]) : "v"(pc), "v"(pd));
        if (KIND == MOV) asm volatile("v_mov_b32 %0, %1" : "=v"(x) : "v"(c));
        if (KIND == IADD) asm volatile("v_add_u32 %0, %0, %1" : "+v"(iv) : "v"(lane));
        if (KIND == SALU) asm volatile("s_add_u32 %0, %0, 1" : "+s"(sv));
        if (KIND == EXP) asm volatile("v_exp_f32 %0, %0" : "+v"(x));
        if (KIND == DSW32) asm volatile("ds_write_b32 %0, %1" ::"v"((unsigned)(size_t)(__attribute__((address_space(3))) float *)my), "v"(x) : "memory");
        if (KIND == DSW128) asm volatile("ds_write_b128 %0, %1" ::"v"((unsigned)(size_t)(__attribute__((address_space(3))) float *)my), "v"(v4) : "memory");
        if (KIND == DSR128) asm volatile("ds_read_b128 %0, %1" : "=v"(v4) : "v"((unsigned)(size_t)(__attribute__((address_space(3))) float *)my) : "memory");
        if (KIND == DSR32) asm volatile("ds_read_b32 %0, %1" : "=v"(x) : "v"((unsigned)(size_t)(__attribute__((address_space(3))) float *)my) : "memory");
        if (KIND == VMEM_LD) x += in[(ch * N + i) * 64 % 4096 + lane];
        if (KIND == FMA_IND8) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(y8[i & 7]) : "v"(c), "v"(d));
        if (KIND == DPP) asm volatile("v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf" : "=v"(x) : "v"(c));
        if (KIND == DSADD) asm volatile("ds_add_f32 %0, %1" ::"v"((unsigned)(size_t)(__attribute__((address_space(3))) float *)my), "v"(c) : "memory");
        if (KIND == VMEM_LAG) asm volatile("buffer_load_dword %0, %1, %2, 0 offen" : "=v"(y8[i & 7]) : "v"(lane * 4), "s"(rs) : "memory");
        if (KIND == DMA16) asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds" ::"s"(ldsb), "v"(lane * 16), "s"(rs) : "memory", "m0");
        if (KIND == VMEM_ST) asm volatile("buffer_store_dword %0, %1, %2, 0 offen" ::"v"(c), "v"(lane * 4 + 8192), "s"(rs) : "memory");
      }
      if (KIND == VMEM_LAG || KIND == DMA16 || KIND == VMEM_ST) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(N) : "memory");  // one chunk of slack
      else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    }
    float s = x + p[0] + p[1] + iv + sv + v4[0] + v4[3];
#pragma unroll
    for (int i = 0; i < 4; ++i) s += q[i][0] + q[i][1];
#pragma unroll
    for (int i = 0; i < 8; ++i) s += y8[i];
    if (s == 12345.678f) out[threadIdx.x] = s;
  } else {
    f32x16 acc[6];
#pragma unroll
    for (int t = 0; t < 6; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    const float a = 1.f + lane * 1e-3f, b = 1.f - lane * 1e-3f;
    for (int ch = 0; ch < chunks; ++ch) {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
#pragma unroll
        for (int t = 0; t < 6; ++t) {
          if (AGPR) asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+a"(acc[t]) : "v"(a), "v"(b));
          else acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[t], 0, 0, 0);
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
  float *out, *in;
  CHECK(hipMalloc(&out, 4096));
  CHECK(hipMalloc(&in, 4096 * 4 + 256));
  CHECK(hipMemset(in, 0, 4096 * 4 + 256));
  const int chunks = 1500, wgs = 256;
  float base[2] = {0, 0};
#define RUN(KIND, N, AG, label)                                                                                                \
  {                                                                                                                           \
    const float ms = timed([&] { hipLaunchKernelGGL((k<KIND, N, AG>), dim3(wgs), dim3(1024), 0, 0, out, in, chunks); });       \
    const float ns = ms * 1e6f / chunks;                                                                                      \
    if (KIND == NONE) base[AG] = ns;                                                                                          \
    printf("%-9s %-44s %7.0f ns per chunk", AG ? "acc=AGPR" : "acc=VGPR", label, ns);                                          \
    if (KIND != NONE) printf("   +%5.2f ns per instruction", (ns - base[AG]) / N);                                             \
    printf("\n");                                                                                                             \
  }
  RUN(NONE, 0, false, "warm-up (discard)");
  RUN(NONE, 0, false, "consumers only (72 MFMAs per SIMD and chunk)");
  RUN(FMA, 144, false, "144 v_fma_f32, dependent");
  RUN(FMA_IND8, 144, false, "144 v_fma_f32, 8 chains");
  RUN(PKFMA_IND, 144, false, "144 v_pk_fma_f32, 4 chains");
  RUN(DPP, 144, false, "144 v_mov_b32_dpp row_shr:1");
  RUN(DSADD, 48, false, "48 ds_add_f32 (private addresses)");
  RUN(VMEM_LAG, 24, false, "24 buffer_load_dword, waited one chunk later");
  RUN(VMEM_ST, 24, false, "24 buffer_store_dword, waited one chunk later");
  RUN(DMA16, 6, false, "6 buffer_load_dwordx4 ... lds, waited one chunk later");
  RUN(DMA16, 24, false, "24 buffer_load_dwordx4 ... lds, waited one chunk later");
  RUN(NONE, 0, false, "consumers only, again");
  return 0;
}
