// precision of v_rsq_f64 / v_rcp_f64 and of one/two Newton steps on gfx950 (max relative error over a sweep)
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <vector>
__global__ void k(const double* d, double* o, int n) {
  int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n) return;
  double x = d[t];
  double r0 = __builtin_amdgcn_rsq(x);
  double r1 = r0 * (1.5 - 0.5 * x * r0 * r0);
  double r2 = r1 * (1.5 - 0.5 * x * r1 * r1);
  double c0 = __builtin_amdgcn_rcp(x);
  double c1 = c0 * (2.0 - x * c0);
  o[5 * t] = r0; o[5 * t + 1] = r1; o[5 * t + 2] = r2; o[5 * t + 3] = c0; o[5 * t + 4] = c1;
}
int main() {
  const int n = 1 << 20;
  std::vector<double> h(n), o(5 * n);
  unsigned long long s = 88172645463325252ull;
  for (int i = 0; i < n; ++i) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; h[i] = std::exp(((s >> 11) * (1.0 / 9007199254740992.0) - 0.5) * 60.0); }
  double *d, *od;
  hipMalloc(&d, n * 8); hipMalloc(&od, 5 * n * 8);
  hipMemcpy(d, h.data(), n * 8, hipMemcpyHostToDevice);
  k<<<n / 256, 256>>>(d, od, n);
  hipMemcpy(o.data(), od, 5 * n * 8, hipMemcpyDeviceToHost);
  double e[5] = {0, 0, 0, 0, 0};
  for (int i = 0; i < n; ++i) {
    long double x = h[i];
    long double rs = 1.0L / sqrtl(x), rc = 1.0L / x;
    for (int j = 0; j < 3; ++j) e[j] = fmax(e[j], (double)fabsl((o[5 * i + j] - rs) / rs));
    for (int j = 3; j < 5; ++j) e[j] = fmax(e[j], (double)fabsl((o[5 * i + j] - rc) / rc));
  }
  printf("rsq raw %.3e  newton1 %.3e  newton2 %.3e | rcp raw %.3e newton1 %.3e  (eps=%.3e)\n", e[0], e[1], e[2], e[3], e[4], 2.22e-16);
  return 0;
}
