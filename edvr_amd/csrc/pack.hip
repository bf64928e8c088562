// pack.hip - every stale conv weight of a training iteration packed in ONE launch.
//
// The optimizer rewrites all ~235 conv weights of EDVR-L each iteration, and each 3x3 / stride-1 layer needs four packed variants
// (forward and data-gradient orientation x {direct + F(2x2) layout, F(4x4) layout}): ~480 launches of ~5 us per iteration, 1.7 %
// of the step, each far too small to fill the GPU.  edvr_conv2d_pack_weights_multi takes a device table of jobs instead
// (the pointers do not change between iterations - parameters are updated in place, the packed buffers are reused - so the host
// builds and uploads it once) and gives every job the workgroups its size asks for.
#include "common.h"
#include "pack.h"

namespace edvr {

struct PackJob {       // 64 bytes, mirrored by edvr_amd/ops.py
  const float *w;      // (co, ci, ks, ks), or (ci, co, ks, ks) read transposed + flipped when transpose_flip (then co / ci name the packed side)
  float *wpk;          // direct layout (+ F(2x2) layout behind it for 3x3 kernels), edvr_conv2d_packed_weight_elems floats; may be NULL
  float *wpk_f4;       // F(4x4) layout, edvr_conv2d_packed_weight_f4_elems floats; may be NULL
  int co, ci, ks, transpose_flip;
  int first_block, n_blocks;  // this job's workgroups of the launch: [first_block, first_block + n_blocks)
  unsigned *wpk_f4s;   // split-operand F(4x4) layout (header + dwords), edvr_conv2d_packed_weight_f4s_elems dwords; may be NULL
  int pad[2];
};
static_assert(sizeof(PackJob) == 64, "PackJob layout");

static inline __host__ __device__ int rup(int v, int m) { return (v + m - 1) / m * m; }

// the job of this workgroup: last job whose first_block <= blockIdx.x (binary search on a wave-uniform index)
__device__ __forceinline__ int job_of_block(const PackJob *__restrict__ jobs, int n_jobs) {
  int lo = 0, hi = n_jobs - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (jobs[mid].first_block <= (int)blockIdx.x) lo = mid; else hi = mid - 1;
  }
  return lo;
}

// max |w| of every job that has a split-operand layout, into header slot 2 + parity of its buffer (atomics on the bit pattern).  The
// packing kernel of the same call reads that slot and zeroes the OTHER one, which the next call (parity flipped) accumulates into:
// no memset launch; a buffer that skips calls may find an old maximum in its slot - still a bound of the weights' magnitude only
// if it is >= the current one, which atomicMax guarantees.
__global__ __launch_bounds__(256) void weights_amax_multi_kernel(const PackJob *__restrict__ jobs, int n_jobs, int parity) {
  const PackJob jb = jobs[job_of_block(jobs, n_jobs)];
  if (!jb.wpk_f4s || jb.ks != 3) return;
  const int64_t total = (int64_t)jb.co * jb.ci * 9, stride = (int64_t)jb.n_blocks * 256;
  float m = 0.f;
  for (int64_t i = (int64_t)((int)blockIdx.x - jb.first_block) * 256 + threadIdx.x; i < total; i += stride) m = fmaxf(m, fabsf(jb.w[i]));
#pragma unroll
  for (int sh = 32; sh > 0; sh >>= 1) m = fmaxf(m, __shfl_xor(m, sh));
  if ((threadIdx.x & 63) == 0 && m > 0.f) atomicMax(jb.wpk_f4s + 2 + parity, __builtin_bit_cast(unsigned, m));
}

__global__ __launch_bounds__(256) void pack_weights_multi_kernel(const PackJob *__restrict__ jobs, int n_jobs, int parity) {
  const PackJob jb = jobs[job_of_block(jobs, n_jobs)];
  const int64_t stride = (int64_t)jb.n_blocks * 256, i0 = (int64_t)((int)blockIdx.x - jb.first_block) * 256 + threadIdx.x;
  const int kk = jb.ks * jb.ks;
  if (jb.wpk) {
    const int cop = rup(jb.co, 32), cip = rup(jb.ci, jb.ks == 1 ? 32 : 16);
    const int64_t total = (int64_t)cip * kk * cop;
    for (int64_t i = i0; i < total; i += stride) pack_direct_elem(jb.w, jb.wpk, i, jb.co, jb.ci, kk, cop, jb.transpose_flip);
    if (jb.ks == 3) {
      const int cop64 = rup(jb.co, 64);
      float *U = jb.wpk + total;
      const int64_t utotal = (int64_t)cip * cop64;
      for (int64_t i = i0; i < utotal; i += stride) pack_u2_elem(jb.w, U, i, jb.co, jb.ci, cop64, jb.transpose_flip);
    }
  }
  if (jb.wpk_f4 && jb.ks == 3) {
    const int cop64 = rup(jb.co, 64), cip8 = rup(jb.ci, 8);
    const int64_t ftotal = (int64_t)cip8 * cop64;
    for (int64_t i = i0; i < ftotal; i += stride) pack_f4_elem(jb.w, jb.wpk_f4, i, jb.co, jb.ci, cop64, cip8, jb.transpose_flip);
  }
  if (jb.wpk_f4s && jb.ks == 3) {
    const int cop64 = rup(jb.co, 64), cip8 = rup(jb.ci, 8);
    const int64_t ftotal = (int64_t)cip8 * cop64;
    const unsigned field = f4s_weight_scale_field(jb.wpk_f4s[2 + parity]);  // (weights_amax_multi_kernel of this call)
    const float s_u = __builtin_bit_cast(float, field << 23);
    if (i0 == 0) {
      jb.wpk_f4s[0] = field << 23;
      jb.wpk_f4s[1] = (254u - field) << 23;
      jb.wpk_f4s[2 + (parity ^ 1)] = 0u;
    }
    for (int64_t i = i0; i < ftotal; i += stride) pack_f4s_elem(jb.w, jb.wpk_f4s, i, jb.co, jb.ci, cop64, cip8, jb.transpose_flip, s_u);
  }
}

}  // namespace edvr

extern "C" {

size_t edvr_pack_job_bytes(void) { return sizeof(edvr::PackJob); }

int edvr_conv2d_pack_weights_multi(const void *jobs, int n_jobs, int total_blocks, int split_parity, edvr_stream_t stream) {
  EDVR_REQUIRE(jobs && n_jobs > 0 && total_blocks > 0, "pack_weights_multi: bad arguments");
  if (split_parity >= 0)  // some job has a split-operand layout: its scale needs max |w| first
    hipLaunchKernelGGL(edvr::weights_amax_multi_kernel, dim3(total_blocks), dim3(256), 0, edvr::as_stream(stream),
                       static_cast<const edvr::PackJob *>(jobs), n_jobs, split_parity & 1);
  hipLaunchKernelGGL(edvr::pack_weights_multi_kernel, dim3(total_blocks), dim3(256), 0, edvr::as_stream(stream),
                     static_cast<const edvr::PackJob *>(jobs), n_jobs, split_parity < 0 ? 0 : (split_parity & 1));
  return edvr::check_launch("pack_weights_multi_kernel");
}

}  // extern "C"
