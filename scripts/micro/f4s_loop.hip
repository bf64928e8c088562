// Standalone micro-benchmark (not part of the library): the MULTIPLYING loop of conv3x3_winograd_f4s_kernel in isolation - 16 waves per CU,
// waves 0-3 only take part in the barrier, waves 4-15 = (transform row, 32-channel half) run 6 positions x 2 v_mfma_f32_32x32x16_f16 per
// 8-channel chunk (36 per SIMD = 1152 matrix-pipe cycles) in one of several loop forms, one barrier per chunk.  profiles/r5/
// micro_mfma16_prices.log measured 568 ns per chunk for the bare MFMAs and 2010 ns for the kernel's loop: which ingredient costs what?
//   hipcc --offload-arch=gfx950 -O3 scripts/micro/f4s_loop.hip -o scripts/micro/f4s_loop
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

enum {
  BARE,        // 2 MFMAs per position, operands constant
  REAL,        // the kernel's loop: 4 ds_read_b32 + wait, MFMA, 4 v_alignbit in place, MFMA
  NO_ROT,      // REAL without the v_alignbit
  NO_READ,     // REAL without the LDS reads
  ROT_COPY,    // REAL with the rotation into a SECOND register set (no write-after-read on the first MFMA's operand)
  DBUF,        // REAL with the B operand double-buffered: position c + 1 is requested before the MFMAs of c
  B128,        // one conflict-free ds_read_b128 per position instead of four ds_read_b32, single-buffered
  B128_DBUF,   // ... double-buffered
  PAIRS,       // two positions at a time: reads of c, c + 1; M1(c), M1(c + 1), rotations, M2(c), M2(c + 1)
  PAIRS_B128,  // ... with ds_read_b128
  ROT_FIRST,   // both A forms ready BEFORE the MFMAs: rotation of position c + 1 issued between the MFMAs of c (out of place, 2 sets)
  PRIO,        // REAL with s_setprio 1 around the MFMA pair
};

template <int V, bool GL, int ST = 0>
__global__ __launch_bounds__(1024, 1) void k(float *out, const i32x4 *U, int chunks) {
  __shared__ __attribute__((aligned(16))) unsigned lds[2 * 8 * 36 * 32];
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63, half = lane >> 5, j = lane & 31;
  for (int i = threadIdx.x; i < 2 * 8 * 36 * 32; i += 1024) lds[i] = 0x3c003c00u;
  __syncthreads();
  if (wave < 4) {
    // ST: 0 = idle; otherwise the staging wave's instruction mix per chunk: 12 ds_read_b128, 144 plain fp32 (8 chains), 72 v_fma_mix (36 splits),
    // and the 36 split values written as 1: 36 ds_write_b32, 2: 18 ds_write_b64, 3: 9 ds_write_b128, 4: not at all
    __shared__ __attribute__((aligned(16))) float sl[4 * 64 * 4];
    const unsigned my_a = (unsigned)(size_t)(__attribute__((address_space(3))) float *)(sl + (wave * 64 + lane) * 4);
    float x = lane, cc = 1.0001f, dd = 1e-3f;
    float y8[8] = {x, x + 1, x + 2, x + 3, x + 4, x + 5, x + 6, x + 7};
    unsigned pk[8] = {0, 1, 2, 3, 4, 5, 6, 7};
    f32x4 v4 = f32x4{x, x, x, x};
    __builtin_amdgcn_s_setprio(3);
    for (int ch = 0; ch < chunks; ++ch) {
      if (ST != 0) {
#pragma unroll
        for (int i = 0; i < 12; ++i) asm volatile("ds_read_b128 %0, %1" : "=v"(v4) : "v"(my_a) : "memory");
#pragma unroll
        for (int i = 0; i < 144; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(y8[i & 7]) : "v"(cc), "v"(dd));
#pragma unroll
        for (int g = 0; g < 9; ++g) {  // four splits (8 instructions, lo first), then their stores
#pragma unroll
          for (int i = 0; i < 4; ++i) asm volatile("v_fma_mixlo_f16 %0, %1, %2, 0" : "=v"(pk[i]) : "v"(y8[i]), "v"(cc));
#pragma unroll
          for (int i = 0; i < 4; ++i) asm volatile("v_fma_mixhi_f16 %0, %1, %2, -%0 op_sel_hi:[0,0,1]" : "+v"(pk[i]) : "v"(y8[i]), "v"(cc));
          if (ST == 1) {
#pragma unroll
            for (int i = 0; i < 4; ++i) asm volatile("ds_write_b32 %0, %1" ::"v"(my_a), "v"(pk[i]) : "memory");
          } else if (ST == 2) {
            asm volatile("ds_write_b64 %0, %1" ::"v"(my_a), "v"(((unsigned long long)pk[1] << 32) | pk[0]) : "memory");
            asm volatile("ds_write_b64 %0, %1 offset:8" ::"v"(my_a), "v"(((unsigned long long)pk[3] << 32) | pk[2]) : "memory");
          } else if (ST == 3) {
            asm volatile("ds_write_b128 %0, %1" ::"v"(my_a), "v"(i32x4{(int)pk[0], (int)pk[1], (int)pk[2], (int)pk[3]}) : "memory");
          }
        }
      }
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    }
    float r = x + v4[0] + v4[3];
#pragma unroll
    for (int i = 0; i < 8; ++i) r += y8[i] + pk[i];
    if (r == 12345.678f) out[1] = r;
    return;
  }
  const int q = wave - 4, wm = q & 1, row = q >> 1;
  f32x16 acc[6];
#pragma unroll
  for (int t = 0; t < 6; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
  constexpr bool kGload = GL;
  constexpr bool kB128 = V == B128 || V == B128_DBUF || V == PAIRS_B128;
  // the forms that need four more registers (a second B or a rotated copy) run with FIVE sets of A (position 5 shares set 0: timing only)
  constexpr int NA = (V == ROT_COPY || V == ROT_FIRST || V == DBUF || V == B128_DBUF || V == PAIRS || V == PAIRS_B128) ? 5 : 6;
  i32x4 A[NA];
#pragma unroll
  for (int c = 0; c < NA; ++c) {
    A[c] = i32x4{lane + c, lane + 1, lane + 2, lane + 3};
    asm volatile("" : "+v"(A[c]));  // opaque: six independent register sets, as in the kernel
  }
  // the kernel's addresses: [channel 8][position 36][tile 32] dwords; or (B128) [position 36][lane 64][4]
  unsigned v_addr = kB128 ? (unsigned)(size_t)(__attribute__((address_space(3))) unsigned *)(lds + (row * 6 * 64 + lane) * 4)
                          : (unsigned)(size_t)(__attribute__((address_space(3))) unsigned *)(lds + 4 * half * 36 * 32 + row * 6 * 32 + j);
  int v_step = 8 * 36 * 32 * 4;
  const int voff = lane * 16;
  const int u_wave = (row * 2 + wm) * 6 * 1024;
  auto rot = [](i32x4 &a) {
    asm volatile("v_alignbit_b32 %0, %0, %0, 16\n\tv_alignbit_b32 %1, %1, %1, 16\n\tv_alignbit_b32 %2, %2, %2, 16\n\tv_alignbit_b32 %3, %3, %3, 16"
                 : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]));
  };
  auto rot_to = [](i32x4 &o, const i32x4 &a) {
    asm volatile("v_alignbit_b32 %0, %4, %4, 16\n\tv_alignbit_b32 %1, %5, %5, 16\n\tv_alignbit_b32 %2, %6, %6, 16\n\tv_alignbit_b32 %3, %7, %7, 16"
                 : "=&v"(o[0]), "=&v"(o[1]), "=&v"(o[2]), "=&v"(o[3])
                 : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]));
  };
#define MFMA(ACC, AA, BB) ACC = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, AA), __builtin_bit_cast(f16x8, BB), ACC, 0, 0, 0)
#define READ4(bv, c)                                                                                                                    \
  asm volatile("ds_read_b32 %0, %4 offset:%5\n\tds_read_b32 %1, %4 offset:%6\n\tds_read_b32 %2, %4 offset:%7\n\tds_read_b32 %3, %4 offset:%8" \
               : "=&v"(bv[0]), "=&v"(bv[1]), "=&v"(bv[2]), "=&v"(bv[3])                                                                  \
               : "v"(v_addr), "n"((c) * 128), "n"(36 * 128 + (c) * 128), "n"(2 * 36 * 128 + (c) * 128), "n"(3 * 36 * 128 + (c) * 128)   \
               : "memory")
#define READ128(bv, c) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=&v"(bv) : "v"(v_addr), "n"((c) * 1024) : "memory")
#define READB(bv, c)            \
  do {                          \
    if (kB128) READ128(bv, c);  \
    else READ4(bv, c);          \
  } while (0)
#define RELOAD(c)                                                                                                  \
  do {                                                                                                             \
    if (kGload) {                                                                                                  \
      __builtin_amdgcn_sched_barrier(0);                                                                           \
      A[(c) % NA] = __builtin_bit_cast(i32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, soff + (c) * 1024, 0));   \
      __builtin_amdgcn_sched_barrier(0);                                                                           \
    }                                                                                                              \
  } while (0)
#define WAIT(n, bv) asm volatile("s_waitcnt lgkmcnt(" #n ")" : "+v"(bv)::"memory")
  const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)U, (short)0, 2 * 16 * 12 * 6 * 1024, 0x00020000);
  i32x4 B = {lane, 2 * lane, 3 * lane, 4 * lane}, B2 = {lane, lane, lane, lane};
  for (int ch = 0; ch < chunks; ++ch) {
    const int soff = ((ch + 1) & 15) * (12 * 6 * 1024) + u_wave;
    if (V == DBUF || V == B128_DBUF) READB(B, 0);
    i32x4 R;
    if (V == ROT_FIRST) rot_to(R, A[0]);
#pragma unroll
    for (int c = 0; c < 6; ++c) {
      if (V == BARE) {
        MFMA(acc[c], A[c % NA], B);
        MFMA(acc[c], A[c % NA], B);
      } else if (V == REAL || V == NO_ROT || V == NO_READ || V == B128 || V == PRIO) {
        if (V != NO_READ) {
          READB(B, c);
          WAIT(0, B);
        }
        if (V == PRIO) __builtin_amdgcn_s_setprio(1);
        MFMA(acc[c], A[c % NA], B);
        if (V != NO_ROT) rot(A[c % NA]);
        MFMA(acc[c], A[c % NA], B);
        if (V == PRIO) __builtin_amdgcn_s_setprio(0);
        RELOAD(c);
      } else if (V == ROT_COPY) {
        READB(B, c);
        WAIT(0, B);
        i32x4 R2;
        MFMA(acc[c], A[c % NA], B);
        rot_to(R2, A[c % NA]);
        MFMA(acc[c], R2, B);
        RELOAD(c);
      } else if (V == ROT_FIRST) {
        READB(B, c);
        WAIT(0, B);
        MFMA(acc[c], A[c % NA], B);
        MFMA(acc[c], R, B);
        RELOAD(c);
        if (c < 5) rot_to(R, A[(c + 1) % NA]);
      } else if (V == DBUF || V == B128_DBUF) {
        i32x4 &cur = (c & 1) ? B2 : B, &nxt = (c & 1) ? B : B2;
        if (c < 5) {
          if (c == 0) READB(nxt, 1);
          if (c == 1) READB(nxt, 2);
          if (c == 2) READB(nxt, 3);
          if (c == 3) READB(nxt, 4);
          if (c == 4) READB(nxt, 5);
          if (kB128) asm volatile("s_waitcnt lgkmcnt(1)" : "+v"(cur)::"memory");
          else asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(cur)::"memory");
        } else {
          WAIT(0, cur);
        }
        MFMA(acc[c], A[c % NA], cur);
        rot(A[c % NA]);
        MFMA(acc[c], A[c % NA], cur);
        RELOAD(c);
      } else if (V == PAIRS || V == PAIRS_B128) {
        if ((c & 1) == 0) {
          if (c == 0) { READB(B, 0); READB(B2, 1); }
          if (c == 2) { READB(B, 2); READB(B2, 3); }
          if (c == 4) { READB(B, 4); READB(B2, 5); }
          asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(B), "+v"(B2)::"memory");
          MFMA(acc[c], A[c % NA], B);
          MFMA(acc[c + 1], A[(c + 1) % NA], B2);
          rot(A[c % NA]);
          rot(A[(c + 1) % NA]);
          MFMA(acc[c], A[c % NA], B);
          MFMA(acc[c + 1], A[(c + 1) % NA], B2);
          RELOAD(c);
          RELOAD(c + 1);
        }
      }
    }
    v_addr += v_step;
    v_step = -v_step;
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
  }
  float s = 0.f;
#pragma unroll
  for (int t = 0; t < 6; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) s += acc[t][r];
  if (s == 12345.678f) out[0] = s;
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
  i32x4 *U;
  CHECK(hipMalloc(&out, 4096));
  CHECK(hipMalloc(&U, 2 * 16 * 12 * 6 * 1024));
  CHECK(hipMemset(U, 0x3c, 2 * 16 * 12 * 6 * 1024));
  const int chunks = 3000, wgs = 256;
#define RUN(V, GL, label) RUNS(V, GL, 0, label)
#define RUNS(V, GL, ST, label)                                                                                         \
  {                                                                                                                    \
    const float ms = timed([&] { hipLaunchKernelGGL((k<V, GL, ST>), dim3(wgs), dim3(1024), 0, 0, out, U, chunks); });   \
    printf("%-118s %7.0f ns per chunk\n", label, ms * 1e6f / chunks);                                                  \
  }
  RUN(BARE, false, "warm-up (discard)");
  RUN(BARE, false, "bare: 12 MFMAs per wave and chunk, constant operands");
  RUN(REAL, false, "the kernel's loop without the global loads: 4 ds_read_b32 + wait, MFMA, 4 v_alignbit in place, MFMA");
  RUN(REAL, true, "the kernel's loop: ... + A re-requested from global memory (L2-resident 2.4 MB) after its use");
  RUN(NO_ROT, true, "... without the v_alignbit");
  RUN(NO_READ, true, "... without the LDS reads");
  RUN(ROT_COPY, true, "... rotation into a second register set (after the first MFMA)");
  RUN(ROT_FIRST, true, "... rotation of the NEXT position's operand issued after the MFMA pair (both forms ready before the pair)");
  RUN(PRIO, true, "... s_setprio 1 around the MFMA pair");
  RUN(DBUF, true, "... B double-buffered (position c + 1 requested before the MFMAs of c)");
  RUN(B128, true, "... one ds_read_b128 per position");
  RUN(B128_DBUF, true, "... one ds_read_b128 per position, double-buffered");
  RUN(PAIRS, true, "... two positions at a time (M1 M1 rot rot M2 M2)");
  RUN(PAIRS_B128, true, "... two positions at a time, ds_read_b128");
  RUNS(BARE, false, 1, "bare MFMAs + the staging wave's mix (12 ds_read_b128, 144 fp32, 36 splits, 36 ds_write_b32)");
  RUNS(BARE, false, 2, "bare MFMAs + the staging mix with 18 ds_write_b64");
  RUNS(BARE, false, 3, "bare MFMAs + the staging mix with 9 ds_write_b128");
  RUNS(BARE, false, 4, "bare MFMAs + the staging mix without the stores");
  RUNS(REAL, true, 1, "the kernel's loop + the staging mix (36 ds_write_b32)");
  RUNS(REAL, true, 3, "the kernel's loop + the staging mix (9 ds_write_b128)");
  RUNS(REAL, true, 4, "the kernel's loop + the staging mix without the stores");
  RUNS(B128_DBUF, true, 1, "ds_read_b128 double-buffered + the staging mix (36 ds_write_b32)");
  RUNS(B128_DBUF, true, 3, "ds_read_b128 double-buffered + the staging mix (9 ds_write_b128)");
  RUNS(PAIRS_B128, true, 1, "two positions at a time, ds_read_b128 + the staging mix (36 ds_write_b32)");
  return 0;
}
