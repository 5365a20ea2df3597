// f64 matrix-core and vector rates on gfx950, measured (the local guide has no FP64 peak):
//   v_mfma_f64_16x16x4_f64 (2048 flop / wave-instruction), v_mfma_f64_4x4x4_4b_f64 (512), v_fma_f64 (128).
// Independent accumulators, back-to-back issue; 1, 2 and 4 waves per SIMD.  Prints cycles per instruction per SIMD
// (from s_memtime) and the chip-wide TFLOP/s that rate gives (from wall time, so DVFS is included).
// Build: hipcc -O3 --offload-arch=gfx950 mfma_f64_rate.hip -o mfma_f64_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double double4_t __attribute__((ext_vector_type(4)));

template <int OP>
__global__ __launch_bounds__(1024) void k(double* out, int iters, unsigned long long* cyc) {
  const double a = 1.0 + 1e-9 * threadIdx.x, b = 1.0 - 1e-9 * threadIdx.x;
  double4_t c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
  double s0 = a, s1 = b, s2 = a, s3 = b, s4 = a, s5 = b, s6 = a, s7 = b;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < iters; ++i) {
    if (OP == 0) {
      c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);
      c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c1, 0, 0, 0);
      c2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c2, 0, 0, 0);
      c3 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c3, 0, 0, 0);
    } else if (OP == 1) {
      s0 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, s0, 0, 0, 0);
      s1 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, s1, 0, 0, 0);
      s2 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, s2, 0, 0, 0);
      s3 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, s3, 0, 0, 0);
    } else {
      s0 = __builtin_fma(s0, a, b); s1 = __builtin_fma(s1, a, b); s2 = __builtin_fma(s2, a, b); s3 = __builtin_fma(s3, a, b);
      s4 = __builtin_fma(s4, a, b); s5 = __builtin_fma(s5, a, b); s6 = __builtin_fma(s6, a, b); s7 = __builtin_fma(s7, a, b);
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  out[blockIdx.x * blockDim.x + threadIdx.x] = c0[0] + c1[1] + c2[2] + c3[3] + s0 + s1 + s2 + s3 + s4 + s5 + s6 + s7;
  if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

template <int OP>
void run(const char* name, double flop_per_instr, int instr_per_iter, int threads) {
  hipDeviceProp_t p;
  hipGetDeviceProperties(&p, 0);
  const int blocks = p.multiProcessorCount, iters = 20000;
  double* out;
  unsigned long long* cyc;
  hipMalloc(&out, sizeof(double) * blocks * threads);
  hipMalloc(&cyc, 8);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int w = 0; w < 3; ++w) k<OP><<<blocks, threads>>>(out, iters, cyc);
  hipEventRecord(e0);
  for (int w = 0; w < 5; ++w) k<OP><<<blocks, threads>>>(out, iters, cyc);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  unsigned long long c;
  hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
  const double waves_per_simd = threads / 256.0;
  const double n_instr = (double)iters * instr_per_iter;                  // per wave
  const double tf = 5.0 * blocks * (threads / 64.0) * n_instr * flop_per_instr / (ms * 1e-3) / 1e12;
  printf("%-28s waves/SIMD %.0f  cycles per instr per SIMD %.2f  chip %.1f TFLOP/s (%d CUs)\n", name, waves_per_simd,
         (double)c / (n_instr * waves_per_simd), tf, blocks);
  hipFree(out); hipFree(cyc);
}

int main() {
  for (int t : {256, 512, 1024}) {
    run<0>("v_mfma_f64_16x16x4_f64", 2048, 4, t);
    run<1>("v_mfma_f64_4x4x4_4b_f64", 512, 4, t);
    run<2>("v_fma_f64", 128, 8, t);
  }
  return 0;
}
