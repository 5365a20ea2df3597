#include <hip/hip_runtime.h>
#include <cstdio>
// throughput of no-return LDS f64 atomic adds: 64 lanes, consecutive doubles (conflict-free) / 12-lane groups of 3
template <int MODE>
__global__ __launch_bounds__(1024) void k(double* out, int iters) {
  extern __shared__ double s[];
  const int tid = threadIdx.x;
  for (int i = tid; i < 8192; i += blockDim.x) s[i] = 0.0;
  __syncthreads();
  double v = 1.0 + tid;
  int base = (tid * 37) & 8191;
  for (int it = 0; it < iters; ++it) {
    if (MODE == 0) {
      __hip_atomic_fetch_add(&s[(base + 64 * it) & 8191], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    } else if (MODE == 1) {
      s[(base + 64 * it) & 8191] = v;      // plain store for comparison
    } else {
      const int lane = tid & 63;
      const int a = ((tid >> 6) * 577 + it * 36 * 5 + (lane / 12) * 36 + (lane % 12) * 3) & 8191;
      if (lane < 60) {
        __hip_atomic_fetch_add(&s[a], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        __hip_atomic_fetch_add(&s[(a + 1) & 8191], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        __hip_atomic_fetch_add(&s[(a + 2) & 8191], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      }
    }
    v += 1.0;
  }
  __syncthreads();
  if (tid < 64) out[blockIdx.x * 64 + tid] = s[tid];
}
int main() {
  double* out; hipMalloc(&out, 256 * 64 * 8 * 4);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  const int iters = 4096;
  for (int mode = 0; mode < 3; ++mode) {
    for (int rep = 0; rep < 2; ++rep) {
      hipEventRecord(a);
      if (mode == 0) k<0><<<256, 1024, 65536>>>(out, iters);
      if (mode == 1) k<1><<<256, 1024, 65536>>>(out, iters);
      if (mode == 2) k<2><<<256, 1024, 65536>>>(out, iters);
      hipEventRecord(b); hipEventSynchronize(b);
      float ms; hipEventElapsedTime(&ms, a, b);
      const double wave_ops = 16.0 * iters * (mode == 2 ? 3 : 1);
      if (rep) printf("mode %d: %.3f ms, %.1f ns per wave-instruction per CU -> %.1f cycles @2.4GHz\n", mode, ms, ms * 1e6 / wave_ops, ms * 1e6 / wave_ops * 2.4);
    }
  }
  return 0;
}
