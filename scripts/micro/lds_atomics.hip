// Standalone micro-benchmark: rate of LDS float / integer atomics and of plain read-add-write, per CU, conflict-free addresses
// (lane-consecutive dwords) - what the DCN backward's LDS-window dX accumulation pays per contribution.
//   hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics scripts/micro/lds_atomics.hip -o scripts/micro/lds_atomics
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
enum { ADD_F32, ADD_U32, RMW, ADD_F32_RTN, ADD_F32_X4, CAS_LOOP, ADD_U64, CAS_LOOP_COLLIDE };
template <int KIND>
__global__ __launch_bounds__(256) void k(float *out, int iters) {
  __shared__ float win[16 * 16 * 41];
  for (int i = threadIdx.x; i < 16 * 16 * 41; i += 256) win[i] = 0.f;
  __syncthreads();
  const int row = threadIdx.x >> 5, col = threadIdx.x & 31;
  float *p = win + (row + 3) * 41 + col + 3;
  const unsigned a = (unsigned)(size_t)(__attribute__((address_space(3))) float *)p;
  float v = (float)threadIdx.x * 1e-3f;
  float acc = 0.f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int c = 0; c < 16; ++c) {
      const unsigned ac = a + c * (16 * 41 * 4);
      if (KIND == ADD_F32) asm volatile("ds_add_f32 %0, %1" ::"v"(ac), "v"(v) : "memory");
      if (KIND == ADD_F32_X4) asm volatile("ds_add_f32 %0, %1\n\tds_add_f32 %0, %1 offset:4\n\tds_add_f32 %0, %1 offset:164\n\tds_add_f32 %0, %1 offset:168" ::"v"(ac), "v"(v) : "memory");
      if (KIND == ADD_U32) asm volatile("ds_add_u32 %0, %1" ::"v"(ac), "v"(__builtin_bit_cast(unsigned, v)) : "memory");
      if (KIND == ADD_F32_RTN) { float r; asm volatile("ds_add_rtn_f32 %0, %1, %2\n\ts_waitcnt lgkmcnt(0)" : "=v"(r) : "v"(ac), "v"(v) : "memory"); acc += r; }
      if (KIND == CAS_LOOP || KIND == CAS_LOOP_COLLIDE) {  // float add as a compare-and-swap loop on the integer pipe
        unsigned *q = reinterpret_cast<unsigned *>(p + c * (16 * 41)) - (KIND == CAS_LOOP_COLLIDE ? (threadIdx.x & 1) : 0);  // (COLLIDE: lane pairs share a cell)
        unsigned old = *q, assumed;
        do {
          assumed = old;
          old = atomicCAS(q, assumed, __builtin_bit_cast(unsigned, __builtin_bit_cast(float, assumed) + v));
        } while (old != assumed);
      }
      if (KIND == ADD_U64) asm volatile("ds_add_u64 %0, %1" ::"v"(ac & ~7u), "v"((unsigned long long)__builtin_bit_cast(unsigned, v)) : "memory");
      if (KIND == RMW) { float r; asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(r) : "v"(ac) : "memory"); r += v; asm volatile("ds_write_b32 %0, %1" ::"v"(ac), "v"(r) : "memory"); }
    }
  }
  __syncthreads();
  if (win[threadIdx.x] + acc == 12345.678f) out[0] = 1.f;
}
template <typename F>
float timed(F launch) {
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  launch(); CHECK(hipDeviceSynchronize());
  CHECK(hipEventRecord(e0));
  for (int i = 0; i < 3; ++i) launch();
  CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
  float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
  return ms / 3;
}
int main() {
  float *out; CHECK(hipMalloc(&out, 64));
  const int iters = 2000, wgs = 256 * 3;
#define RUN(K, mult, label) { const float ms = timed([&] { hipLaunchKernelGGL((k<K>), dim3(wgs), dim3(256), 0, 0, out, iters); }); \
    const double ops = (double)wgs * 256 * iters * 16 * mult; printf("%-44s %8.1f G lane-ops/s  (%.2f lane-ops per cycle and CU at 2.0 GHz)\n", label, ops / ms / 1e6, ops / ms / 1e6 / 256 / 2.0); }
  RUN(ADD_F32, 1, "warm-up");
  RUN(ADD_F32, 1, "ds_add_f32 (no return)");
  RUN(ADD_F32_X4, 4, "ds_add_f32 x4 (the four corners)");
  RUN(ADD_U32, 1, "ds_add_u32 (no return)");
  RUN(ADD_F32_RTN, 1, "ds_add_rtn_f32 + wait");
  RUN(RMW, 1, "ds_read_b32, add, ds_write_b32");
  RUN(CAS_LOOP, 1, "float add as a CAS loop (atomicCAS on LDS)");
  RUN(CAS_LOOP_COLLIDE, 1, "... with lane pairs sharing a cell");
  RUN(ADD_U64, 1, "ds_add_u64 (no return)");
  return 0;
}
