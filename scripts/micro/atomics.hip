// Standalone micro-benchmark (not part of the library): throughput of global float atomic adds on MI355X as a function of
// the access pattern and the scope - the number that decides whether a per-tap dX flush ("tap-window backward") is affordable.
//   hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics scripts/micro/atomics.hip -o /tmp/atomics && /tmp/atomics
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

// mode 0: contiguous (lane i of wave w -> element base + i: one 256-byte run per wave), every element touched `reps` times by different workgroups
// mode 1: the same runs, but every workgroup owns its own region (no two workgroups touch the same line)
// mode 2: scattered (a hashed element per lane)
template <int SCOPE>
__global__ void atomics_kernel(float *buf, size_t n, int mode, int rounds) {
  const size_t tid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t total = (size_t)gridDim.x * blockDim.x;
  for (int r = 0; r < rounds; ++r) {
    size_t idx;
    if (mode == 0) idx = (tid + (size_t)r * 4099 * 64) % n;                       // runs shared between workgroups over the rounds
    else if (mode == 1) idx = ((size_t)blockIdx.x * 4096 + (threadIdx.x + (size_t)r * 256) % 4096) % n;  // private 16 KB region per workgroup
    else idx = ((tid * 2654435761u) ^ ((size_t)r * 40503u)) % n;
    if (SCOPE == 0) unsafeAtomicAdd(buf + idx, 1.0f);
    else if (SCOPE == 1) __hip_atomic_fetch_add(buf + idx, 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else __hip_atomic_fetch_add(buf + idx, 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  }
  (void)total;
}

__global__ void plain_kernel(float *buf, size_t n, int rounds) {  // the same traffic as mode 1 with plain read-add-write (private regions: no race)
  for (int r = 0; r < rounds; ++r) {
    const size_t idx = ((size_t)blockIdx.x * 4096 + (threadIdx.x + (size_t)r * 256) % 4096) % n;
    buf[idx] += 1.0f;
  }
}

template <typename F>
float timed(F launch) {
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  launch();
  CHECK(hipDeviceSynchronize());
  CHECK(hipEventRecord(e0));
  for (int i = 0; i < 5; ++i) launch();
  CHECK(hipEventRecord(e1));
  CHECK(hipEventSynchronize(e1));
  float ms;
  CHECK(hipEventElapsedTime(&ms, e0, e1));
  return ms / 5;
}

int main() {
  const size_t n = (size_t)84 << 20;  // 84 M floats = 336 MB: the dX of a 160 x 128 x 64 x 64 layer
  float *buf;
  CHECK(hipMalloc(&buf, n * 4));
  CHECK(hipMemset(buf, 0, n * 4));
  const int blocks = 16384, rounds = 64;
  const double ops = (double)blocks * 256 * rounds;
  const char *modes[3] = {"contiguous runs, shared between workgroups", "contiguous runs, private 16 KB region per workgroup", "scattered"};
  for (int mode = 0; mode < 3; ++mode) {
    const float a = timed([&] { hipLaunchKernelGGL(atomics_kernel<0>, dim3(blocks), dim3(256), 0, 0, buf, n, mode, rounds); });
    const float b = timed([&] { hipLaunchKernelGGL(atomics_kernel<1>, dim3(blocks), dim3(256), 0, 0, buf, n, mode, rounds); });
    const float c = timed([&] { hipLaunchKernelGGL(atomics_kernel<2>, dim3(blocks), dim3(256), 0, 0, buf, n, mode, rounds); });
    printf("%-52s unsafeAtomicAdd %7.1f G/s | agent scope %7.1f G/s | workgroup scope %7.1f G/s\n", modes[mode], ops / a / 1e6, ops / b / 1e6, ops / c / 1e6);
  }
  const float p = timed([&] { hipLaunchKernelGGL(plain_kernel, dim3(blocks), dim3(256), 0, 0, buf, n, rounds); });
  printf("%-52s plain read-add-write %7.1f G/s\n", modes[1], ops / p / 1e6);
  return 0;
}
