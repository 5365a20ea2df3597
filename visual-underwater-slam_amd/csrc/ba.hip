// ba.hip -- stereo bundle-adjustment kernels for gfx950 (MI355X), fp64.
//
// What gtsam.LevenbergMarquardtOptimizer(graph, values, params).optimize() (reference batch.py:337)
// spends its time on for a graph of GenericStereoFactor3D factors (batch.py:300-305) with
// PriorFactorPose3 gauge priors (batch.py:281): per-factor residual/Jacobian evaluation, damped
// normal equations, landmark Schur complement, reduced camera solve, back-substitution, retraction
// and error evaluation.  The Levenberg-Marquardt control flow stays on the host (ba.py).
//
// Layout (include/vus.h): observations in L-order (point-major) and P-order (pose-major) with both
// permutations precomputed; the per-observation 6x3 product W in L-ORDER (round 4; rounds 1-3: P-order), so that a
// landmark's rows of an 8-pose tile are consecutive: the tile-pair Schur kernel reads runs of up to 8 rows, the
// linearisation writes and the back-substitution reads W as one stream (Y = W Vinv is formed on the fly).
//
// Kernel map.  Reductions are fixed-order wave / LDS / DPP sums everywhere except the cooperative
// back-substitution of the band solve, which adds its partial products into the right-hand side with f64
// atomics: two solves of the same system agree to ~1e-13 relative, not bitwise (the landmark-sharded
// solver broadcasts rank 0's dp for that reason, dist.py).
//   lin_points   wave / point      r, H1, H2 -> W (L-order: streamed), V, gl, error partial
//   lin_poses    workgroup / pose  r, H1 (recomputed, never stored) -> Hpp, gp
//   priors       one lane          PriorFactorPose3 information / gradient / error
//   vinv (ymul)  thread / point, thread / obs   (V + lambda I)^-1  (Y = W Vinv only as an optional output)
//   schur_tiles  S = Hpp + lambda I - sum Y W^T and gs = gp - sum Y gl as a block-sparse GEMM on v_mfma_f64_16x16x4:
//                persistent workgroups, one 8 x 8-pose tile pair at a time, the landmarks seen from both tiles side by
//                side along K (vus_ba_tiles, built by pack.hip)
//   chol_panel (panel 0) / chol_trsm + chol_syrk (two launches per 8-pose panel: MFMA block substitution of the
//                window's row tiles, then the SYRK tiles on v_mfma_f64_16x16x4_f64, tile (0,0) goes on to factor
//                the next panel; chol_trsm_update = both fused in one launch, used for a single system) /
//                diag_invert + chol_backsolve (diagonal panels inverted in place, then the cooperative,
//                flag-ordered sweep) / split_* (two-sided elimination: both ends of the band at once)
//   backsub      wave / point      dl = -Vinv (gl + sum W^T dp)
//   retract, eval_points, error_points, reduce_partials
#include <atomic>
#include <cerrno>
#include <cstdio>
#include <cstdlib>
#include <memory>
#include <mutex>
#include <vector>
#include "vus_common.h"
#include "band_index.h"

namespace {

// Tuning knobs (include/vus.h, vus_ba_set_tuning): the environment is read ONCE, when the library is loaded.
struct Knobs {
  std::atomic<int> band_mode{-1};
  std::atomic<int> cb_max_wg{0};
  std::atomic<int> last_mode{-1};      // how the last factorisation of this process was issued (0..3): diagnostics / bench
  std::atomic<int> win_pad{1};         // VUS_WIN_PAD=0: no padding blocks beside the critical workgroups (A/B timing)
  std::atomic<int> win_fault{0};       // VUS_TUNE_WIN_FAULT (tests): one window workgroup exits at once, as if it had never become resident
  static void env_knob(const char* name, long lo, long hi, std::atomic<int>& knob) {
    const char* e = getenv(name);
    if (!e) return;
    char* end = nullptr;
    errno = 0;
    const long v = strtol(e, &end, 10);
    if (end == e || *end != '\0' || errno != 0 || v < lo || v > hi) {
      fprintf(stderr, "libvus_hip: ignoring %s=\"%s\" (an integer in [%ld, %ld] is expected)\n", name, e, lo, hi);
      return;
    }
    knob = (int)v;
  }
  Knobs() {
    // same bounds as vus_ba_set_tuning; anything else is reported and leaves the default (a typo such as
    // VUS_BAND_MODE=auto used to read as 0 = the slowest mode, silently)
    env_knob("VUS_BAND_MODE", -1, 3, band_mode);
    env_knob("VUS_CB_MAX_WG", 0, 1 << 20, cb_max_wg);
    env_knob("VUS_WIN_PAD", 0, 1, win_pad);
  }
};
Knobs g_knobs;

typedef double double4_t __attribute__((ext_vector_type(4)));
typedef double d2a_t __attribute__((ext_vector_type(2), aligned(16)));

// the wave's index in its workgroup as a SCALAR: branches on it are scalar branches (derived from threadIdx.x alone the
// compiler treats it as divergent and wraps every wave-specialised region in exec-mask saves and restores)
__device__ __forceinline__ int wave_index() { return __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)); }

// compute units of the current device (cached per device)
int device_cu_count() {
  static std::mutex mu;
  static int cached[64] = {};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 0;
  std::lock_guard<std::mutex> lock(mu);
  if (!cached[dev] && hipDeviceGetAttribute(&cached[dev], hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) cached[dev] = 0;
  return cached[dev];
}

constexpr double kEps = 2.220446049250313e-16;
constexpr double kPi = 3.14159265358979323846;

// ---------------------------------------------------------------------------------------------
// Lie-group helpers (gtsam Pose3 / Rot3 conventions; mirrored independently by the CPU oracle)
__device__ void so3_expmap(const double* w, double* R) {
  const double th2 = w[0] * w[0] + w[1] * w[1] + w[2] * w[2];
  const double Wx[9] = {0, -w[2], w[1], w[2], 0, -w[0], -w[1], w[0], 0};
  if (th2 <= kEps) {
#pragma unroll
    for (int i = 0; i < 9; ++i) R[i] = Wx[i] + (i % 4 == 0 ? 1.0 : 0.0);
    return;
  }
  const double th = sqrt(th2);
  const double s = sin(th) / th;
  const double sh = sin(0.5 * th);
  const double c = 2.0 * sh * sh / th2;
#pragma unroll
  for (int r = 0; r < 3; ++r)
#pragma unroll
    for (int cc = 0; cc < 3; ++cc) {
      double ww = 0;
#pragma unroll
      for (int k = 0; k < 3; ++k) ww += Wx[3 * r + k] * Wx[3 * k + cc];
      R[3 * r + cc] = (r == cc ? 1.0 : 0.0) + s * Wx[3 * r + cc] + c * ww;
    }
}

__device__ void so3_logmap(const double* R, double* w) {
  const double tr = R[0] + R[4] + R[8];
  if (tr + 1.0 < 1e-10) {
    if (fabs(R[8] + 1.0) > 1e-5) {
      double k = kPi / sqrt(2.0 + 2.0 * R[8]);
      w[0] = k * R[2]; w[1] = k * R[5]; w[2] = k * (1.0 + R[8]);
    } else if (fabs(R[4] + 1.0) > 1e-5) {
      double k = kPi / sqrt(2.0 + 2.0 * R[4]);
      w[0] = k * R[1]; w[1] = k * (1.0 + R[4]); w[2] = k * R[7];
    } else {
      double k = kPi / sqrt(2.0 + 2.0 * R[0]);
      w[0] = k * (1.0 + R[0]); w[1] = k * R[3]; w[2] = k * R[6];
    }
    return;
  }
  double mag;
  const double tr3 = tr - 3.0;
  if (tr3 < -1e-7) {
    double th = acos((tr - 1.0) / 2.0);
    mag = th / (2.0 * sin(th));
  } else {
    mag = 0.5 - tr3 / 12.0;
  }
  w[0] = mag * (R[7] - R[5]);
  w[1] = mag * (R[2] - R[6]);
  w[2] = mag * (R[3] - R[1]);
}

// out = T * Exp(xi)
__device__ void pose_retract(const double* T, const double* xi, double* out) {
  double Re[9], te[3];
  so3_expmap(xi, Re);
  const double* w = xi;
  const double* v = xi + 3;
  const double th2 = w[0] * w[0] + w[1] * w[1] + w[2] * w[2];
  if (th2 > kEps) {
    double wv = w[0] * v[0] + w[1] * v[1] + w[2] * v[2];
    double c[3] = {w[1] * v[2] - w[2] * v[1], w[2] * v[0] - w[0] * v[2], w[0] * v[1] - w[1] * v[0]};
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      double Rc = Re[3 * r] * c[0] + Re[3 * r + 1] * c[1] + Re[3 * r + 2] * c[2];
      te[r] = (c[r] - Rc + w[r] * wv) / th2;
    }
  } else {
    te[0] = v[0]; te[1] = v[1]; te[2] = v[2];
  }
#pragma unroll
  for (int r = 0; r < 3; ++r) {
#pragma unroll
    for (int c = 0; c < 3; ++c)
      out[3 * r + c] = T[3 * r] * Re[c] + T[3 * r + 1] * Re[3 + c] + T[3 * r + 2] * Re[6 + c];
    out[9 + r] = T[9 + r] + (T[3 * r] * te[0] + T[3 * r + 1] * te[1] + T[3 * r + 2] * te[2]);
  }
}

// xi = Logmap(T^-1 * T2)
__device__ void pose_local(const double* T, const double* T2, double* xi) {
  double R[9], t[3];
#pragma unroll
  for (int r = 0; r < 3; ++r) {
#pragma unroll
    for (int c = 0; c < 3; ++c) R[3 * r + c] = T[r] * T2[c] + T[3 + r] * T2[3 + c] + T[6 + r] * T2[6 + c];
    t[r] = T[r] * (T2[9] - T[9]) + T[3 + r] * (T2[10] - T[10]) + T[6 + r] * (T2[11] - T[11]);
  }
  double w[3];
  so3_logmap(R, w);
  const double th = sqrt(w[0] * w[0] + w[1] * w[1] + w[2] * w[2]);
  xi[0] = w[0]; xi[1] = w[1]; xi[2] = w[2];
  if (th < 1e-10) {
    xi[3] = t[0]; xi[4] = t[1]; xi[5] = t[2];
    return;
  }
  const double k[3] = {w[0] / th, w[1] / th, w[2] / th};
  const double WT[3] = {k[1] * t[2] - k[2] * t[1], k[2] * t[0] - k[0] * t[2], k[0] * t[1] - k[1] * t[0]};
  const double WWT[3] = {k[1] * WT[2] - k[2] * WT[1], k[2] * WT[0] - k[0] * WT[2], k[0] * WT[1] - k[1] * WT[0]};
  const double tn = tan(0.5 * th);
#pragma unroll
  for (int r = 0; r < 3; ++r) xi[3 + r] = t[r] - (0.5 * th) * WT[r] + (1.0 - th / (2.0 * tn)) * WWT[r];
}

struct Calib {
  double fx, fy, cx, cy, b, w;
};

__device__ __forceinline__ Calib load_calib(const double* K, double inv_sigma) {
  return Calib{K[0], K[1], K[3], K[4], K[5], inv_sigma};
}

// GenericStereoFactor3D: whitened residual and (optionally) Jacobians.
// gtsam StereoCamera::project2: q = R^T (p - t); z <= 0 -> cheirality: residual 2 fx, zero Jacobians.
template <bool WITH_H1, bool WITH_H2>
__device__ __forceinline__ void stereo_factor(const double* __restrict__ T, const double* __restrict__ p,
                                              const double* __restrict__ m, const Calib& K, double* r,
                                              double* H1, double* H2) {
  const double d0 = p[0] - T[9], d1 = p[1] - T[10], d2 = p[2] - T[11];
  const double x = T[0] * d0 + T[3] * d1 + T[6] * d2;
  const double y = T[1] * d0 + T[4] * d1 + T[7] * d2;
  const double z = T[2] * d0 + T[5] * d1 + T[8] * d2;
  if (z <= 0.0) {
    r[0] = r[1] = r[2] = 2.0 * K.fx * K.w;
    if (WITH_H1)
#pragma unroll
      for (int k = 0; k < 18; ++k) H1[k] = 0.0;
    if (WITH_H2)
#pragma unroll
      for (int k = 0; k < 9; ++k) H2[k] = 0.0;
    return;
  }
  const double d = 1.0 / z;
  r[0] = (K.cx + d * K.fx * x - m[0]) * K.w;
  r[1] = (K.cx + d * K.fx * (x - K.b) - m[1]) * K.w;
  r[2] = (K.cy + d * K.fy * y - m[2]) * K.w;
  if (!WITH_H1 && !WITH_H2) return;
  const double J[9] = {K.w * d * K.fx, 0.0, -K.w * d * d * K.fx * x,
                       K.w * d * K.fx, 0.0, -K.w * d * d * K.fx * (x - K.b),
                       0.0, K.w * d * K.fy, -K.w * d * d * K.fy * y};
  if (WITH_H2)
#pragma unroll
    for (int rr = 0; rr < 3; ++rr)
#pragma unroll
      for (int c = 0; c < 3; ++c)
        H2[3 * rr + c] = J[3 * rr] * T[3 * c] + J[3 * rr + 1] * T[3 * c + 1] + J[3 * rr + 2] * T[3 * c + 2];
  if (WITH_H1) {
    const double Q[9] = {0, -z, y, z, 0, -x, -y, x, 0};
#pragma unroll
    for (int rr = 0; rr < 3; ++rr)
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        H1[6 * rr + c] = J[3 * rr] * Q[c] + J[3 * rr + 1] * Q[3 + c] + J[3 * rr + 2] * Q[6 + c];
        H1[6 * rr + 3 + c] = -J[3 * rr + c];
      }
  }
}

__host__ __device__ __forceinline__ int pose_stride(const vus_ba_problem& P) { return P.pose_stride > 1 ? P.pose_stride : 1; }

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

__device__ __forceinline__ void load12(const double* __restrict__ src, double* dst) {
#pragma unroll
  for (int k = 0; k < 12; ++k) dst[k] = src[k];
}

__device__ __forceinline__ double sym3(const double* v, int r, int c) {
  // upper-triangle storage xx,xy,xz,yy,yz,zz
  const int lo = r < c ? r : c, hi = r < c ? c : r;
  return v[lo == 0 ? hi : (lo == 1 ? 2 + hi : 5)];
}

// ---------------------------------------------------------------------------------------------
// linearisation
__global__ __launch_bounds__(256) void lin_points_kernel(vus_ba_problem P, const double* __restrict__ poses,
                                                         const double* __restrict__ points,
                                                         double* __restrict__ W, double* __restrict__ V,
                                                         double* __restrict__ gl, double* __restrict__ err_part) {
  const int lane = threadIdx.x & 63;
  const int j = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (j >= P.n_points) return;
  const Calib K = load_calib(P.K, P.inv_sigma);
  const int a0 = P.point_ptr[j], a1 = P.point_ptr[j + 1];
  const double p[3] = {points[3 * j], points[3 * j + 1], points[3 * j + 2]};
  double v[6] = {0, 0, 0, 0, 0, 0}, g[3] = {0, 0, 0}, e = 0;
  for (int a = a0 + lane; a < a1; a += 64) {
    double T[12], r[3], H1[18], H2[9];
    load12(poses + 12 * (size_t)P.obs_pose[a], T);
    const double m[3] = {P.meas[3 * (size_t)a], P.meas[3 * (size_t)a + 1], P.meas[3 * (size_t)a + 2]};
    stereo_factor<true, true>(T, p, m, K, r, H1, H2);
    e += 0.5 * (r[0] * r[0] + r[1] * r[1] + r[2] * r[2]);
    double* Wa = W + 18 * (size_t)a;
#pragma unroll
    for (int rr = 0; rr < 6; ++rr)
#pragma unroll
      for (int c = 0; c < 3; ++c) Wa[3 * rr + c] = H1[rr] * H2[c] + H1[6 + rr] * H2[3 + c] + H1[12 + rr] * H2[6 + c];
    int u = 0;
#pragma unroll
    for (int rr = 0; rr < 3; ++rr)
#pragma unroll
      for (int c = rr; c < 3; ++c, ++u) v[u] += H2[rr] * H2[c] + H2[3 + rr] * H2[3 + c] + H2[6 + rr] * H2[6 + c];
#pragma unroll
    for (int c = 0; c < 3; ++c) g[c] += H2[c] * r[0] + H2[3 + c] * r[1] + H2[6 + c] * r[2];
  }
#pragma unroll
  for (int k = 0; k < 6; ++k) v[k] = wave_sum(v[k]);
#pragma unroll
  for (int k = 0; k < 3; ++k) g[k] = wave_sum(g[k]);
  e = wave_sum(e);
  if (lane == 0) {
#pragma unroll
    for (int k = 0; k < 6; ++k) V[6 * (size_t)j + k] = v[k];
#pragma unroll
    for (int k = 0; k < 3; ++k) gl[3 * (size_t)j + k] = g[k];
    err_part[j] = e;
  }
}

__global__ __launch_bounds__(256) void lin_poses_kernel(vus_ba_problem P, const double* __restrict__ poses,
                                                        const double* __restrict__ points,
                                                        double* __restrict__ Hpp, double* __restrict__ gp) {
  __shared__ double s_part[4][27];
  const int i = blockIdx.x;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const Calib K = load_calib(P.K, P.inv_sigma);
  double T[12];
  load12(poses + 12 * (size_t)i, T);
  double acc[27];
#pragma unroll
  for (int k = 0; k < 27; ++k) acc[k] = 0;
  const int s0 = P.pose_ptr[i], s1 = P.pose_ptr[i + 1];
  for (int s = s0 + (int)threadIdx.x; s < s1; s += 256) {
    const int a = P.pobs_lidx[s];
    const int j = P.obs_point[a];
    const double p[3] = {points[3 * (size_t)j], points[3 * (size_t)j + 1], points[3 * (size_t)j + 2]};
    const double m[3] = {P.meas[3 * (size_t)a], P.meas[3 * (size_t)a + 1], P.meas[3 * (size_t)a + 2]};
    double r[3], H1[18];
    stereo_factor<true, false>(T, p, m, K, r, H1, nullptr);
    int u = 0;
#pragma unroll
    for (int rr = 0; rr < 6; ++rr)
#pragma unroll
      for (int c = rr; c < 6; ++c, ++u) acc[u] += H1[rr] * H1[c] + H1[6 + rr] * H1[6 + c] + H1[12 + rr] * H1[12 + c];
#pragma unroll
    for (int rr = 0; rr < 6; ++rr) acc[21 + rr] += H1[rr] * r[0] + H1[6 + rr] * r[1] + H1[12 + rr] * r[2];
  }
#pragma unroll
  for (int k = 0; k < 27; ++k) acc[k] = wave_sum(acc[k]);
  if (lane == 0)
#pragma unroll
    for (int k = 0; k < 27; ++k) s_part[wave][k] = acc[k];
  __syncthreads();
  if (threadIdx.x < 36) {
    const int rr = threadIdx.x / 6, c = threadIdx.x % 6;
    const int lo = rr < c ? rr : c, hi = rr < c ? c : rr;
    const int u = lo * 6 - lo * (lo - 1) / 2 + (hi - lo);  // index in the row-wise upper triangle
    Hpp[36 * (size_t)i + threadIdx.x] = ((s_part[0][u] + s_part[1][u]) + s_part[2][u]) + s_part[3][u];
  } else if (threadIdx.x < 42) {
    const int u = 21 + threadIdx.x - 36;
    gp[6 * (size_t)i + threadIdx.x - 36] = ((s_part[0][u] + s_part[1][u]) + s_part[2][u]) + s_part[3][u];
  }
}

// PriorFactorPose3 (gtsam PriorFactor::evaluateError: e = -Local(x, prior), H = I), one lane, in order.
// mode 0: add information/gradient into Hpp/gp and write the error to err_out[0]
// mode 1: error only;  mode 2: linearised error 0.5|r + w dp|^2 at OLD poses to err_out[0]
__global__ void priors_kernel(vus_ba_problem P, const double* __restrict__ poses, const double* __restrict__ dp,
                              double* __restrict__ Hpp, double* __restrict__ gp, double* __restrict__ err_out,
                              int mode) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  double e = 0;
  for (int q = 0; q < P.n_priors; ++q) {
    const int i = P.prior_pose[q];
    double xi[6];
    pose_local(poses + 12 * (size_t)i, P.prior_T + 12 * (size_t)q, xi);
    for (int k = 0; k < 6; ++k) {
      const double w = P.prior_w[6 * (size_t)q + k];
      double r = -xi[k] * w;
      if (mode == 0) {
        Hpp[36 * (size_t)i + 7 * k] += w * w;
        gp[6 * (size_t)i + k] += w * r;
      }
      if (mode == 2) r += w * dp[6 * (size_t)pose_stride(P) * i + k];
      e += 0.5 * r * r;
    }
  }
  err_out[0] = e;
}

// out[0] = sum of part[0..n) in a fixed order (one workgroup)
__global__ __launch_bounds__(1024) void reduce_partials_kernel(const double* __restrict__ part, int n,
                                                               double* __restrict__ out) {
  __shared__ double s[1024];
  double acc = 0;
  for (int k = threadIdx.x; k < n; k += 1024) acc += part[k];
  s[threadIdx.x] = acc;
  __syncthreads();
  for (int o = 512; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) s[threadIdx.x] += s[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) out[0] = s[0];
}

// ---------------------------------------------------------------------------------------------
// damped landmark elimination
__global__ void vinv_kernel(int n_points, double lambda, const double* __restrict__ V, double* __restrict__ Vinv) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n_points) return;
  const double a = V[6 * (size_t)j] + lambda, b = V[6 * (size_t)j + 1], c = V[6 * (size_t)j + 2];
  const double d = V[6 * (size_t)j + 3] + lambda, e = V[6 * (size_t)j + 4], f = V[6 * (size_t)j + 5] + lambda;
  const double c00 = d * f - e * e, c01 = c * e - b * f, c02 = b * e - c * d;
  const double id = 1.0 / (a * c00 + b * c01 + c * c02);
  double* o = Vinv + 6 * (size_t)j;
  o[0] = c00 * id; o[1] = c01 * id; o[2] = c02 * id;
  o[3] = (a * f - c * c) * id; o[4] = (b * c - a * e) * id; o[5] = (a * d - b * b) * id;
}

__global__ void ymul_kernel(vus_ba_problem P, const double* __restrict__ W, const double* __restrict__ Vinv,
                            double* __restrict__ Y) {
  const int s = blockIdx.x * blockDim.x + threadIdx.x;      // L-order row
  if (s >= P.n_obs) return;
  const int j = P.obs_point[s];
  double vi[6];
#pragma unroll
  for (int k = 0; k < 6; ++k) vi[k] = Vinv[6 * (size_t)j + k];
  const double* Ws = W + 18 * (size_t)s;
  double* Ys = Y + 18 * (size_t)s;
#pragma unroll
  for (int rr = 0; rr < 6; ++rr) {
    const double w0 = Ws[3 * rr], w1 = Ws[3 * rr + 1], w2 = Ws[3 * rr + 2];
    Ys[3 * rr + 0] = w0 * vi[0] + w1 * vi[1] + w2 * vi[2];
    Ys[3 * rr + 1] = w0 * vi[1] + w1 * vi[3] + w2 * vi[4];
    Ys[3 * rr + 2] = w0 * vi[2] + w1 * vi[4] + w2 * vi[5];
  }
}

// ---------------------------------------------------------------------------------------------
// The landmark elimination on the matrix cores: S(I,K) = init - sum_j A_j B_j^T per 8 x 8-pose tile pair (include/vus.h,
// vus_ba_tiles), the sum over the landmarks being the K dimension of a GEMM.  A workgroup of four waves owns one unit at
// a time (persistent, units handed out largest first through a global counter).  A wave takes the chunks c = wave,
// wave + 4, ... of the unit's landmark list, ST_CH landmarks each: it stages A = [W_ij Vinv_j] and B = [W_kj] of the chunk
// K-MAJOR into its own LDS panels (panel[k][row], row = 6 * pose + component: an MFMA operand fetch is 16 consecutive
// doubles per k), zero rows where a pose does not see the landmark, and runs the 3 x 3 output tiles of 16 x 16 over the
// chunk's K = 3 ST_CH columns.  Software-pipelined inside the wave: while the matrix cores work on chunk c, the W rows of
// chunk c + 1 and the entry of chunk c + 2 are in flight (two dependent loads deep: entry -> rows) -- with 77 KB of LDS
// per workgroup in panels only two waves per SIMD are resident and nothing else would hide a memory round trip.  No
// workgroup barrier inside a unit: the panels are wave-private, LDS operations of one wave complete in order.  The four
// partial tiles meet in LDS at the end (fixed order: no result depends on scheduling, bit-identical from run to run) and
// leave as 16-byte vectors in the band's address order.  W is in L-order: a landmark's rows of a tile are consecutive.
//
// Measured at configs[2] (profiles/r04_summary.md; 2000 keyframes, 50,238 landmarks, 2.0 M factors, 1.82 M entries): the
// per-pair vector kernel of rounds 1-3 1.36 ms (2.6 ms once a pose has more than 1000 observations) + 0.11 ms for the
// right-hand side; this kernel 0.65 ms including it.  Steps on the way (same box A/B, tools/schur_ab.py): W gathered
// through obs_ppos 1.66 -> W in L-order 1.23 -> pipelined 0.86 -> one entry and Vinv per lane in registers instead of LDS
// tables, operands of the next K step requested before the products of this one, conflict-free panel writes 0.77 ->
// right-hand side fused 0.69.  One queue of units per XCD in natural order (neighbouring tile rows together on one L2)
// raised the L2 hit rate from 16 % to 42 % and was SLOWER (0.79): the kernel is not memory-bound -- with every row read
// from a cache-resident 4096-row window it takes 0.85 instead of 0.89 ms -- but a unit is up to 1638 entries = 40 % of the
// kernel's duration for its workgroup, so the largest-first order matters more than locality.
constexpr int ST_CH = 8;                       // landmarks per chunk
constexpr int ST_KC = 3 * ST_CH;               // K columns per chunk: 6 steps of v_mfma_f64_16x16x4
constexpr int ST_LD = 50;                      // panel stride in doubles (48 rows + 2)
constexpr int ST_PANEL = ST_KC * ST_LD;
constexpr int ST_WAVES = 4;
constexpr int ST_WAVE_DOUBLES = 2 * ST_PANEL + ST_KC + 8;                 // the two panels, gl by K row (+ padding to 16 bytes)
static_assert(2 * ST_PANEL >= 48 * 48, "a wave's panels also hold its partial 48 x 48 tile");
static_assert(ST_KC % 4 == 0, "whole MFMA K steps");

__device__ __forceinline__ void wave_lds_fence() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

__global__ __launch_bounds__(64 * ST_WAVES, 2) void schur_tiles_kernel(vus_ba_tiles T, int n_poses, int ps, int band_nodes,
                                                                        double lambda, const double* __restrict__ W,
                                                                        const double* __restrict__ Vinv,
                                                                        const double* __restrict__ Hpp,
                                                                        const double* __restrict__ gl, const double* __restrict__ gp,
                                                                        double* __restrict__ Sband, double* __restrict__ gs,
                                                                        int* __restrict__ counter) {
  extern __shared__ __attribute__((aligned(16))) double st_smem[];
  __shared__ int s_unit;
  const int tid = threadIdx.x, lane = tid & 63, wave = wave_index();
  double* __restrict__ PA = st_smem + wave * ST_WAVE_DOUBLES;
  double* __restrict__ PB = PA + ST_PANEL;
  double* __restrict__ gtab = PB + ST_PANEL;      // [ST_KC]: gl of the chunk's landmarks by K row (tile pairs (I, I) only)
  const int4* __restrict__ entries = reinterpret_cast<const int4*>(T.entries);
  const int dt1 = T.n_units / T.n_tiles;
  const int arow = lane & 15, kq = lane >> 4;
  // lane = 8 * landmark of the chunk + sub: the lane's six tasks per side are rows qr = 8 rnd + sub of ITS landmark, so one
  // entry and one Vinv per lane and chunk, straight from memory into registers (no LDS tables, no dependent LDS reads in
  // the staging pass); the eight lanes of a landmark read 192 contiguous bytes per round.
  const int tl = lane >> 3, sub = lane & 7, krow = (tl >> 1) + 4 * (tl & 1);
  int tq[6], tr[6];
#pragma unroll
  for (int rnd = 0; rnd < 6; ++rnd) {
    tq[rnd] = (8 * rnd + sub) / 6;
    tr[rnd] = (8 * rnd + sub) - 6 * tq[rnd];
  }
  while (true) {
    __syncthreads();                 // the previous unit's partial tiles have been consumed, s_unit has been read
    if (tid == 0) s_unit = atomicAdd(counter, 1);
    __syncthreads();
    const int ui = s_unit;
    if (ui >= T.n_units) break;
    const int u = T.order[ui];
    const int I = u / dt1, d = u - I * dt1, K = I - d;
    if (K < 0) continue;
    const bool diag = d == 0;
    const int e0 = T.unit_ptr[u], e1 = T.unit_ptr[u + 1];
    const int n_chunks = (e1 - e0 + ST_CH - 1) / ST_CH;
    double4_t acc[3][3];
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
      for (int q = 0; q < 3; ++q) acc[t][q] = double4_t{0.0, 0.0, 0.0, 0.0};
    // pipeline registers
    int4 en;               // the lane's entry of the chunk whose rows are requested next
    double wa[6][3], wb[6][3], vi[6], gv = 0.0;
    unsigned pres = 0;     // bit rnd / 6 + rnd: the task's pose sees the landmark (side A / B)
    // (nothing may touch `en` between its load and load_rows: a use would make the compiler wait for it at once, and with
    // it -- the memory counter is in-order -- for every row load issued just before: the whole prefetch would be lost)
    auto load_entries = [&](int c) {
      const int e = e0 + ST_CH * c + tl;
      en = entries[e < e1 ? e : e0];
    };
    auto load_rows = [&](int c) {     // from en (arrived): the W rows of chunk c and the landmark's Vinv
      pres = 0;
      const bool valid = e0 + ST_CH * c + tl < e1;
      const int ma = valid ? en.w & 0xFF : 0, mb = valid ? (en.w >> 8) & 0xFF : 0;
#pragma unroll
      for (int rnd = 0; rnd < 6; ++rnd) {
        const int bit = 1 << tq[rnd], below = bit - 1;
        const bool pa = ma & bit, pb = mb & bit;
        pres |= (pa ? 1u : 0u) << rnd | (pb ? 1u : 0u) << (6 + rnd);
        const double* ra = W + 18 * (size_t)(en.x + (pa ? __popc(ma & below) : 0)) + 3 * tr[rnd];
        const double* rb = W + 18 * (size_t)(en.y + (pb ? __popc(mb & below) : 0)) + 3 * tr[rnd];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
          wa[rnd][k] = ra[k];
          wb[rnd][k] = rb[k];
        }
      }
      const double* vsrc = Vinv + 6 * (size_t)en.z;
#pragma unroll
      for (int k = 0; k < 6; ++k) vi[k] = vsrc[k];
      if (diag) gv = valid && sub < 3 ? gl[3 * (size_t)en.z + sub] : 0.0;      // the reduced right-hand side rides along
    };
    if (wave < n_chunks) {
      load_entries(wave);
      load_rows(wave);
      if (wave + ST_WAVES < n_chunks) load_entries(wave + ST_WAVES);
    }
    for (int c = wave; c < n_chunks; c += ST_WAVES) {
      // chunk c: rows (arrived during the previous chunk's products) -> panels
      if (diag && sub < 3) gtab[8 * sub + krow] = gv;
#pragma unroll
      for (int rnd = 0; rnd < 6; ++rnd) {
        const bool pa = (pres >> rnd) & 1, pb = (pres >> (6 + rnd)) & 1;
        const double a0 = pa ? wa[rnd][0] : 0.0, a1 = pa ? wa[rnd][1] : 0.0, a2 = pa ? wa[rnd][2] : 0.0;
        // K row of (landmark tl, column c) = 8 c + krow: which K index holds what is free as long as both panels agree.
        // ds_write_b64 is served in groups of 16 consecutive lanes = two landmarks: with a stride of 50 doubles their
        // rows must lie 4 apart to fall on disjoint banks (4 * 100 dwords = 16 mod 32)
        double* da = PA + krow * ST_LD + 8 * rnd + sub;
        da[0] = a0 * vi[0] + a1 * vi[1] + a2 * vi[2];
        da[8 * ST_LD] = a0 * vi[1] + a1 * vi[3] + a2 * vi[4];
        da[16 * ST_LD] = a0 * vi[2] + a1 * vi[4] + a2 * vi[5];
        double* db = PB + krow * ST_LD + 8 * rnd + sub;
        db[0] = pb ? wb[rnd][0] : 0.0;
        db[8 * ST_LD] = pb ? wb[rnd][1] : 0.0;
        db[16 * ST_LD] = pb ? wb[rnd][2] : 0.0;
      }
      // chunk c + 1: its entries have arrived -> request its rows; chunk c + 2: request its entries
      if (c + ST_WAVES < n_chunks) {
        load_rows(c + ST_WAVES);
        if (c + 2 * ST_WAVES < n_chunks) load_entries(c + 2 * ST_WAVES);
      }
      wave_lds_fence();
      {
        double a[2][3], b[2][3];     // the operands of step s2 + 1 are requested before the products of step s2 are issued
#pragma unroll
        for (int t = 0; t < 3; ++t) {
          a[0][t] = PA[kq * ST_LD + 16 * t + arow];
          b[0][t] = PB[kq * ST_LD + 16 * t + arow];
        }
#pragma unroll
        for (int s2 = 0; s2 < ST_KC / 4; ++s2) {
          const int cur = s2 & 1, nxt = cur ^ 1;
          if (s2 + 1 < ST_KC / 4) {
#pragma unroll
            for (int t = 0; t < 3; ++t) {
              a[nxt][t] = PA[(4 * (s2 + 1) + kq) * ST_LD + 16 * t + arow];
              b[nxt][t] = PB[(4 * (s2 + 1) + kq) * ST_LD + 16 * t + arow];
            }
          }
#pragma unroll
          for (int t = 0; t < 3; ++t)
#pragma unroll
            for (int q = 0; q <= t; ++q) acc[t][q] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[cur][t], b[cur][q], acc[t][q], 0, 0, 0);
          // Of the tile pair (I, I) the 16 x 16 tiles above the diagonal are not stored: their accumulators take
          // Y gl instead -- a B operand whose column 0 is gl along K and whose other columns are zero -- so column 0 of
          // the three tiles is sum_j Y_ij Vinv_j... = what gs subtracts, rows 0-15, 16-31, 32-47
          const double bg = diag && arow == 0 ? gtab[4 * s2 + kq] : 0.0;
          acc[0][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[cur][0], diag ? bg : b[cur][1], acc[0][1], 0, 0, 0);
          acc[0][2] = __builtin_amdgcn_mfma_f64_16x16x4f64(diag ? a[cur][1] : a[cur][0], diag ? bg : b[cur][2], acc[0][2], 0, 0, 0);
          acc[1][2] = __builtin_amdgcn_mfma_f64_16x16x4f64(diag ? a[cur][2] : a[cur][1], diag ? bg : b[cur][2], acc[1][2], 0, 0, 0);
        }
      }
      wave_lds_fence();
    }
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
      for (int q = 0; q < 3; ++q)
#pragma unroll
        for (int r = 0; r < 4; ++r) PA[(16 * t + kq + 4 * r) * 48 + 16 * q + arow] = acc[t][q][r];
    __syncthreads();
    for (int v = tid; v < 8 * 144; v += 64 * ST_WAVES) {
      const int ii = v / 144, w = v - 144 * ii, o = w / 18, e = 2 * (w - 18 * o), kk = 7 - o;
      const int i = 8 * I + ii, k = 8 * K + kk;
      if (i >= n_poses || k > i || ps * (i - k) > band_nodes) continue;
      const int rr = e / 6, cc = e - 6 * rr;
      // a diagonal block is written whole: its elements above the diagonal come from their mirror images (of the tile pair
      // (I, I) only the 16 x 16 tiles on and below the diagonal are computed)
      const bool dblk = i == k;
      const int at0 = dblk && cc > rr ? (6 * ii + cc) * 48 + 6 * kk + rr : (6 * ii + rr) * 48 + 6 * kk + cc;
      const int at1 = dblk && cc + 1 > rr ? (6 * ii + cc + 1) * 48 + 6 * kk + rr : (6 * ii + rr) * 48 + 6 * kk + cc + 1;
      double s0 = 0.0, s1 = 0.0;
#pragma unroll
      for (int wv = 0; wv < ST_WAVES; ++wv) {
        s0 += st_smem[wv * ST_WAVE_DOUBLES + at0];
        s1 += st_smem[wv * ST_WAVE_DOUBLES + at1];
      }
      d2a_t out = d2a_t{-s0, -s1};
      if (i == k) {
        out.x += Hpp[36 * (size_t)i + e] + (rr == cc ? lambda : 0.0);
        out.y += Hpp[36 * (size_t)i + e + 1] + (rr == cc + 1 ? lambda : 0.0);
      }
      *reinterpret_cast<d2a_t*>(Sband + 36 * ((size_t)(ps * i) * (band_nodes + 1) + (size_t)ps * (i - k)) + e) = out;
    }
    if (diag && tid < 48 && 8 * I + tid / 6 < n_poses) {      // gs rows of the tile's poses: column 0 of the three spare tiles
      const int m = tid, t = m >> 4, mm = m & 15;
      const int at = t == 0 ? mm * 48 + 16 : (t == 1 ? mm * 48 + 32 : (16 + mm) * 48 + 32);
      double sg = 0.0;
#pragma unroll
      for (int wv = 0; wv < ST_WAVES; ++wv) sg += st_smem[wv * ST_WAVE_DOUBLES + at];
      const int i = 8 * I + m / 6;
      gs[6 * (size_t)(ps * i) + (m - 6 * (m / 6))] = gp[6 * (size_t)i + (m - 6 * (m / 6))] - sg;
    }
  }
}

// ---------------------------------------------------------------------------------------------
// block-band Cholesky, right-looking, panels of PB poses (NB = 6 PB scalar columns).
// Sband entry (i, s) is the 6x6 block (i, i - s), s in [0, band].
//
// A panel step = the panel's 48x48 diagonal block factored (panel_factor: four waves, lane = row, 6-column block
// steps, the right-hand sides ride along as extra rows), the window's rows solved against it (block forward
// substitution on the matrix cores, stage_and_solve) and the window updated (SYRK tiles of 48x48 on
// v_mfma_f64_16x16x4_f64).  How the steps are issued -- one fused launch per panel, a TRSM + SYRK launch pair, or the
// persistent window kernel -- is decided in factor_launches().
constexpr int PB = 8;
constexpr int NB = 6 * PB;
constexpr int LDD = NB + 1;

__device__ __forceinline__ double* blk_ptr(double* Sb, int band, int i, int k) {
  return Sb + 36 * ((size_t)i * (band + 1) + (i - k));
}
__device__ __forceinline__ const double* blk_ptr(const double* Sb, int band, int i, int k) {
  return Sb + 36 * ((size_t)i * (band + 1) + (i - k));
}


template <int CTRL>
__device__ __forceinline__ double dpp_f64(double v) {      // v of the lane the DPP control selects
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xF, 0xF, true);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xF, 0xF, true);
  return __hiloint2double(hi, lo);
}

__device__ __forceinline__ double bcast_lane(double v, int src_lane) {   // src_lane wave-uniform
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), src_lane);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), src_lane);
  return __hiloint2double(hi, lo);
}

// v_rsq_f64 delivers ~24 bits (measured 5.2e-8, tools/ubench/rsq_prec.hip); one third-order step
// r (1 + h/2 + 3h^2/8), h = 1 - d r^2, brings it to 1 ulp in five dependent operations.
__device__ __forceinline__ double rsqrt_newton(double d) {
  const double r = __builtin_amdgcn_rsq(d);
  const double h = __builtin_fma(-(d * r), r, 1.0);
  const double p = __builtin_fma(0.375, h, 0.5);
  return __builtin_fma(r * h, p, r);
}

// Four waves, lane R of every wave = row R of the 48x48 block (rows nb.. = the right-hand sides riding
// along); wave w keeps the 6-column blocks kb = w and w + 4 of its rows in registers.  Block step s: the
// owner wave pulls the 6x6 diagonal block into SGPRs (readlane), factors it in registers, solves its rows
// against it and publishes them through a ring of three LDS panels; one barrier per block step, and the
// owner of block s+1 updates that block first and factors it while the other waves finish step s.
// The 36 doubles of the solved 6x6 block (a wave-uniform LDS address: 18 broadcast ds_read_b128) are all requested
// before the first multiply-add: left to itself the compiler issues them one by one with a full s_waitcnt behind each
// (18 dependent LDS round trips per block, ~1.5 k cycles on the panel's critical chain).
__device__ __forceinline__ void panel_update(double (&blk)[6], const double (&xr)[6], const double* __restrict__ xs) {
  // two halves of nine loads: all 18 at once cost the window kernel its last free registers (scratch spills)
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    double w[18];
#pragma unroll
    for (int i = 0; i < 18; ++i) w[i] = xs[18 * h + i];
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int k = 0; k < 6; ++k)
#pragma unroll
      for (int c = 0; c < 3; ++c) blk[3 * h + c] = __builtin_fma(-xr[k], w[6 * c + k], blk[3 * h + c]);
    if (h == 0) __builtin_amdgcn_sched_barrier(0);      // the second half may mix with what follows (the pivot chain's idle slots)
  }
}

// -DVUS_TIMING: s_memtime marks of ONE panel step of the window kernel's critical workgroup (system 0, panel 41), kept in
// a device array and printed once after the last step, so that the marks cost thread 0 a scalar load and a store and
// nothing else (a printf inside the loop costs the whole kernel registers and shifts every number)
#ifdef VUS_TIMING
__device__ unsigned long long g_wtm[32];      // [0, 16): window kernel; [16, 32): back-substitution
__shared__ unsigned long long s_wtm[16];       // marks go to LDS (a global store per mark would stall thread 0's wave at the next reuse of its registers)
#define VUS_WM(k) do { if (vus_wm_on && threadIdx.x == 0) s_wtm[k] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define VUS_WM(k)
#endif
// FROM_LDS: the 48x48 block (row stride LDD) and the right-hand-side rows (row stride NB) are handed over in
// LDS by the workgroup that has just produced them (fused launch) instead of being re-read from memory;
// only the first lds_poses poses of the panel were touched by that update (bands narrower than a panel),
// the rest still comes from memory.
// PUBLISH (persistent window kernel): nothing is written to memory here; lds_out / lds_rhs_out receive the factor (row
// stride LDD, zeros above the diagonal) and the solved right-hand-side rows (row stride NB), and the caller stores both
// with agent-scope (sc1) stores, because workgroups of the SAME launch read them; lds_inv receives 1 / L_cc of the
// panel's 6 pb columns.
template <bool FROM_LDS, bool PUBLISH = false>
__device__ __forceinline__ void panel_factor(double* Sb, int n_poses, int band, int k0, double* yv, size_t ystride,
                                             int n_rhs, int* __restrict__ status, double (*s_x)[64 * 6],
                                             int& s_bad, const double* lds_tile = nullptr,
                                             const double* lds_rhs = nullptr, int lds_poses = 0,
                                             double* lds_out = nullptr, double* lds_rhs_out = nullptr,
                                             double* lds_inv = nullptr) {
  [[maybe_unused]] const bool vus_wm_on = PUBLISH && k0 == 8 * 41 && blockIdx.x == 0;
  const int lane = threadIdx.x & 63, wave = wave_index();
  const int pb = min(PB, n_poses - k0);
  const int nb = 6 * pb;
  const int R = lane;                 // rows 0..nb-1: block rows; rows nb..nb+n_rhs-1: the right-hand sides
  const int ii = R / 6;
  const bool is_rhs = R >= nb && R < nb + n_rhs;
  double* yrow = yv + (size_t)(is_rhs ? R - nb : 0) * ystride + 6 * (size_t)k0;
  if (threadIdx.x == 0) s_bad = 0x7FFFFFFF;
  double row[2][6];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int kb = wave + 4 * j;
#pragma unroll
    for (int c = 0; c < 6; ++c) row[j][c] = 0.0;
    // PUBLISH: the whole panel comes from LDS (lds_poses = pb).  Kept apart at compile time: a pointer SELECTED between
    // LDS and memory is a generic pointer and its loads FLAT loads (slower than ds_read, and they wait on both counters).
    if constexpr (PUBLISH) {
      if (bandidx::panel_row_ok(band, nb, R, kb)) {
        const double* src = lds_tile + R * LDD + 6 * kb;
#pragma unroll
        for (int c = 0; c < 6; ++c) row[j][c] = src[c];
      }
      if (is_rhs && kb < pb) {
        const double* src = lds_rhs + (R - nb) * NB + 6 * kb;
#pragma unroll
        for (int c = 0; c < 6; ++c) row[j][c] = src[c];
      }
    } else {
      typedef const __attribute__((address_space(3))) double* lds_cptr;      // keeps the two sources' loads apart
      if (bandidx::panel_row_ok(band, nb, R, kb)) {
        if (FROM_LDS && ii < lds_poses) {
          lds_cptr src = (lds_cptr)(lds_tile + R * LDD + 6 * kb);
#pragma unroll
          for (int c = 0; c < 6; ++c) row[j][c] = src[c];
        } else {      // the offset into the band (64-bit arithmetic) only where the row really comes from memory
          const double* src = Sb + bandidx::panel_row(band, k0, nb, R, kb);
#pragma unroll
          for (int c = 0; c < 6; ++c) row[j][c] = src[c];
        }
      }
      if (is_rhs && kb < pb) {
        if (FROM_LDS && kb < lds_poses) {
          lds_cptr src = (lds_cptr)(lds_rhs + (R - nb) * NB + 6 * kb);
#pragma unroll
          for (int c = 0; c < 6; ++c) row[j][c] = src[c];
        } else {
#pragma unroll
          for (int c = 0; c < 6; ++c) row[j][c] = yrow[6 * kb + c];
        }
      }
    }
  }
  __syncthreads();
  VUS_WM(1);
#pragma unroll
  for (int s = 0; s < PB; ++s) {
    if (s < pb) {   // uniform
      double xp[6];
      if (wave == (s & 3)) {
        double (&a)[6] = row[s >> 2];
        if (s > 0) {   // look-ahead: bring this block up to date with step s-1 before factoring it
#pragma unroll
          for (int k = 0; k < 6; ++k) xp[k] = s_x[(s - 1) % 3][6 * lane + k];
          panel_update(a, xp, &s_x[(s - 1) % 3][36 * s]);
        }
        double D[6][6], inv[6];
#pragma unroll
        for (int r = 0; r < 6; ++r)
#pragma unroll
          for (int c = 0; c <= r; ++c) D[r][c] = bcast_lane(a[c], 6 * s + r);
        // A non-positive pivot is NOTED (first one of the block) and reported after the block; the arithmetic goes on
        // with it (NaN from there on: the caller discards a solve whose status is set).  Testing every pivot before
        // its square root put a compare, two selects and a branch into each of the 48 links of the panel's chain.
        int first_bad = 6;
#pragma unroll
        for (int c = 0; c < 6; ++c) {
          const double d = D[c][c];
          if (!(d > 0.0) && first_bad == 6) first_bad = c;
          const double rs = rsqrt_newton(d);
          inv[c] = rs;
#pragma unroll
          for (int r = c + 1; r < 6; ++r) D[r][c] *= rs;
#pragma unroll
          for (int r = c + 1; r < 6; ++r)
#pragma unroll
            for (int c2 = c + 1; c2 <= r; ++c2) D[r][c2] -= D[r][c] * D[c2][c];
        }
        // x = a * L66^-T (rows of the diagonal block reproduce L66 in their lower part)
#pragma unroll
        for (int c = 0; c < 6; ++c) {
          double acc = a[c];
#pragma unroll
          for (int k = 0; k < c; ++k) acc -= a[k] * D[c][k];
          a[c] = acc * inv[c];
        }
#pragma unroll
        for (int k = 0; k < 6; ++k) s_x[s % 3][6 * lane + k] = a[k];
        if (first_bad < 6 && lane == 0) atomicMin(&s_bad, 6 * k0 + 6 * s + first_bad + 1);
        if (PUBLISH && lds_inv != nullptr && lane == 0) {      // 1 / L_cc, for the caller's inverse blocks
#pragma unroll
          for (int c = 0; c < 6; ++c) lds_inv[6 * s + c] = inv[c];
        }
      }
      __syncthreads();
      // the step s-1 update of the owner's other block was deferred behind the factorisation and the
      // barrier (the ring of three buffers keeps step s-1 readable until the barrier of step s+1)
      if (wave == (s & 3) && s > 0 && s < 4 && s + 4 < pb) panel_update(row[1], xp, &s_x[(s - 1) % 3][36 * (s + 4)]);
      {
        double xr[6];
#pragma unroll
        for (int k = 0; k < 6; ++k) xr[k] = s_x[s % 3][6 * lane + k];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const int kb = wave + 4 * j;
          // block s+1 is updated by its owner at the top of the next step
          if (4 * j + 3 > s && kb > s + 1 && kb < pb && !(wave == ((s + 1) & 3) && s + 1 < pb))
            panel_update(row[j], xr, &s_x[s % 3][36 * kb]);
        }
      }
    }
  }
  VUS_WM(2);
  __syncthreads();
  if (threadIdx.x == 0 && s_bad != 0x7FFFFFFF && status[0] == 0) status[0] = s_bad;
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int kb = wave + 4 * j;
    const bool stored = bandidx::panel_row_ok(band, nb, R, kb);
    if (!PUBLISH && stored) {          // PUBLISH: the caller stores the tile from lds_out, whole cache lines at a time
      double* dst = Sb + bandidx::panel_row(band, k0, nb, R, kb);
#pragma unroll
      for (int c = 0; c < 6; ++c) dst[c] = (6 * kb + c <= R) ? row[j][c] : 0.0;   // strict upper part of the diagonal blocks = 0
    }
    if (PUBLISH && lds_out != nullptr && R < NB) {
#pragma unroll
      for (int c = 0; c < 6; ++c) lds_out[R * LDD + 6 * kb + c] = (stored && 6 * kb + c <= R) ? row[j][c] : 0.0;
    }
    if (!PUBLISH && is_rhs && kb < pb) {     // PUBLISH: the caller stores them from lds_rhs_out (a store from `row` here
#pragma unroll                             // would hold the registers the LDS copies below rewrite: a store round trip each)
      for (int c = 0; c < 6; ++c) yrow[6 * kb + c] = row[j][c];
    }
    if (PUBLISH && lds_rhs_out != nullptr && R >= nb && R < nb + n_rhs) {
#pragma unroll
      for (int c = 0; c < 6; ++c) lds_rhs_out[(R - nb) * NB + 6 * kb + c] = kb < pb ? row[j][c] : 0.0;
    }
  }
  VUS_WM(3);
}

// One block-band system handed to the factorisation kernels: storage, right-hand sides [n_rhs, 6 n] (solved in
// place), status word, flag area of the cooperative sweep, number of poses.  A launch serves one system, or two
// of identical geometry interleaved block by block (the two halves of the two-sided solve, see band_solve_split).
struct BandSys {
  double* Sb;
  double* y;
  int* status;
  int* F;
  int n;
  double* win_pub = nullptr;      // scratch of the persistent window kernel (window_doubles()); null: launches only
  int* win_F = nullptr;
};
struct BandSet {
  BandSys s[2];
  int count;
};

__global__ __launch_bounds__(256) void chol_panel_kernel(BandSet S, int band, int k0, int n_rhs) {
  __shared__ __attribute__((aligned(16))) double s_x[3][64 * 6];
  __shared__ int s_bad;
  const BandSys B = S.s[blockIdx.x];
  panel_factor<false>(B.Sb, B.n, band, k0, B.y, 6 * (size_t)B.n, n_rhs, B.status, s_x, s_bad);
}

// Rows below the panel, fused with the trailing update.  X = A_rows,panel * L_D^-T is what a TRSM
// kernel would write back before the SYRK A_ij -= X_i X_j^T; here every workgroup of the update
// solves the two 48-row tiles it needs itself (the panel columns of Sband stay untouched while the
// launch runs, so there is no ordering between workgroups to respect), and the in-place write-back of
// X -- only the final back-substitution reads it -- rides along as extra workgroups of the NEXT
// panel's launch.  One launch per panel instead of two, and no 1-wave-per-CU substitution kernel.
//
// Update tiles: one workgroup per UT x UT tile of the lower triangle, (UT/16)^2 MFMA tiles of 16x16
// shared by 4 waves, K = 48 = 12 steps of v_mfma_f64_16x16x4_f64.  The right-hand sides ride along:
// y_i -= X_i y_panel (done by the diagonal tiles).
constexpr int BS_RHS_MAX = 8;      // right-hand sides a solve can carry (= BS_MAX_RHS)
constexpr int UT = 48;            // scalar rows per tile
constexpr int UMT = UT / 16;      // MFMA tiles per side
constexpr int UQ = (UMT * UMT + 3) / 4;   // MFMA tiles per wave
constexpr int UTP = UT / 6;       // poses per tile
constexpr int ULD = NB + 1;       // LDS row stride (doubles)
constexpr int MLD = 17;           // LDS row stride of the 16x16 inverse blocks

#ifdef VUS_TIMING
__device__ unsigned long long g_tm[8];
#define VUS_TMARK(n) do { if (threadIdx.x == 0 && blockIdx.x == 0) g_tm[n] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define VUS_TMARK(n)
#endif

// Inverse blocks of the panel's 48x48 factor L_D held in sL (row stride LDD, zeros above the diagonal; sInv = 1 / diagonal,
// 1 for the rows of a short last panel): M_b = (16x16 diagonal block b)^-1 -> sM, G_b = -M_b * L_D[row block b][columns <
// 16 b] written over L_D's blocks (1,0), (2,0), (2,1) in sL.  Called by all 256 threads behind a barrier; ends in one.
// Workgroup barrier that orders LDS traffic only: __syncthreads() also waits for every outstanding vector-memory
// operation (s_waitcnt vmcnt(0)), i.e. for the write-through stores and the prefetching loads the persistent window
// kernel keeps in flight on purpose.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
template <bool LDS_ONLY>
__device__ __forceinline__ void wg_barrier() {
  if (LDS_ONLY) lds_barrier();
  else __syncthreads();
}

template <bool LDS_ONLY = false>
__device__ __forceinline__ void block_inverses(double* __restrict__ sL, double* __restrict__ sM,
                                               const double* __restrict__ sInv) {
  const int tid = threadIdx.x, lane = tid & 63, wave = wave_index();
  // M_b = (16x16 diagonal block b)^-1 from its 8x8 quadrants:  [A 0; C B]^-1 = [A^-1 0; -B^-1 C A^-1  B^-1]
  if (tid < NB) {   // column n of the inverse of 8x8 diagonal block h by forward substitution (rows past nb: identity)
    const int h = tid >> 3, n = tid & 7;
    const double* Ld = sL + (8 * h) * LDD + 8 * h;
    // every LDS operand requested before the first use (the compiler would otherwise wait for each in turn: 14 dependent
    // LDS round trips on the critical workgroup's chain)
    double l[8][8], iv[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      iv[r] = sInv[8 * h + r];
#pragma unroll
      for (int k = 0; k < r; ++k) l[r][k] = Ld[r * LDD + k];
    }
    __builtin_amdgcn_sched_barrier(0);
    double m[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      double acc = (r == n) ? 1.0 : 0.0;
#pragma unroll
      for (int k = 0; k < r; ++k) acc -= l[r][k] * m[k];
      m[r] = acc * iv[r];
    }
    double* Mq = sM + 16 * MLD * (h >> 1) + (8 * MLD + 8) * (h & 1);
#pragma unroll
    for (int r = 0; r < 8; ++r) Mq[MLD * r + n] = m[r];
    if (h & 1) {   // upper-right quadrant of M_b
#pragma unroll
      for (int r = 0; r < 8; ++r) sM[16 * MLD * (h >> 1) + MLD * r + 8 + n] = 0.0;
    }
  }
  wg_barrier<LDS_ONLY>();
  {
    const int b3 = tid >> 6, r = (tid >> 3) & 7, cq = tid & 7;   // threads < 192: element (r, cq) of quadrant C of block b3
    const bool act = tid < 192;
    double* Mb = sM + 16 * MLD * (act ? b3 : 0);
    double t = 0.0;
    if (act) {   // T = C A^-1
      const double* Lc = sL + (16 * b3 + 8 + r) * LDD + 16 * b3;
      double lc[8], ma[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        lc[k] = Lc[k];
        ma[k] = Mb[MLD * k + cq];
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int k = 0; k < 8; ++k) t += lc[k] * ma[k];
      Mb[MLD * (8 + r) + cq] = t;     // parked in the quadrant it will leave
    }
    wg_barrier<LDS_ONLY>();
    if (act) {   // -B^-1 T
      double u = 0.0, mb[8], tk[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        mb[k] = Mb[MLD * (8 + r) + 8 + k];
        tk[k] = Mb[MLD * (8 + k) + cq];
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int k = 0; k < 8; ++k) u -= mb[k] * tk[k];
      t = u;
    }
    wg_barrier<LDS_ONLY>();
    if (act) Mb[MLD * (8 + r) + cq] = t;
  }
  wg_barrier<LDS_ONLY>();
  VUS_TMARK(2);
  const int arow = lane & 15, kq = lane >> 4;
  if (wave < 3) {   // G tiles (b, kt) = (1,0), (2,0), (2,1):  -M_b * L_D[16b.., 16kt..], in place
    const int b = wave == 0 ? 1 : 2, kt = wave == 2 ? 1 : 0;
    const double* pm = sM + 16 * MLD * b + MLD * arow + kq;                // A operand: M_b[arow][4s + kq]
    const double* pl = sL + (16 * b + kq) * LDD + 16 * kt + arow;     // B operand: L[16b + 4s + kq][16kt + arow]
    double av[4], bv[4];
#pragma unroll
    for (int s2 = 0; s2 < 4; ++s2) {
      av[s2] = pm[4 * s2];
      bv[s2] = pl[4 * s2 * LDD];
    }
    __builtin_amdgcn_sched_barrier(0);
    double4_t acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int s2 = 0; s2 < 4; ++s2) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av[s2], bv[s2], acc, 0, 0, 0);
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
#pragma unroll
    for (int r = 0; r < 4; ++r) sL[(16 * b + kq + 4 * r) * LDD + 16 * kt + arow] = -acc[r];
  }
  wg_barrier<LDS_ONLY>();
  VUS_TMARK(3);
}

// X = A * L_D^-T for n_tiles (1 or 2) 48-row tiles staged row-major in Xa / Xb (row stride ULD), in place: block forward
// substitution with the inverse blocks of block_inverses(),
//   X_b = [X_0 .. X_b-1] * G_b^T + A_b * M_b^T   (4 b + 4 steps of v_mfma_f64_16x16x4_f64 per 16-row tile).
// A 16-row tile belongs to one wave from start to end, so the three block steps need no workgroup barrier.  Ends in one.
template <bool LDS_ONLY = false>
__device__ __forceinline__ void solve_rows(int n_tiles, double* __restrict__ Xa, double* __restrict__ Xb,
                                           const double* __restrict__ sL, const double* __restrict__ sM) {
  const int tid = threadIdx.x, lane = tid & 63, wave = wave_index();
  const int arow = lane & 15, kq = lane >> 4;
  // 16-row tiles: n_tiles * 3 of them, wave w takes tiles w and w + 4
  const int n16 = 3 * n_tiles;
  double* Xt0 = (wave < 3 ? Xa + 16 * wave * ULD : Xb);
  double* Xt1 = Xb + 16 * (wave + 1) * ULD;          // tiles 4, 5 = rows 16.., 32.. of the second tile
  const bool two = wave + 4 < n16;
  const bool one = wave < n16;
  if (one) {   // wave-uniform
    const double* pa0 = Xt0 + arow * ULD + kq;
    const double* pa1 = Xt1 + arow * ULD + kq;
#pragma unroll
    for (int b = 0; b < 3; ++b) {
      const double* gb = sL + (16 * b + arow) * LDD + kq;
      const double* mb = sM + 16 * MLD * b + MLD * arow + kq;
      double bv[12], av0[12], av1[12];
#pragma unroll
      for (int s2 = 0; s2 < 12; ++s2) {
        if (s2 >= 4 * b + 4) continue;
        bv[s2] = s2 < 4 * b ? gb[4 * s2] : mb[4 * (s2 - 4 * b)];
        av0[s2] = pa0[4 * s2];
        av1[s2] = two ? pa1[4 * s2] : 0.0;
      }
      __builtin_amdgcn_sched_barrier(0);
      double4_t acc0 = {0.0, 0.0, 0.0, 0.0}, acc1 = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int s2 = 0; s2 < 12; ++s2) {
        if (s2 >= 4 * b + 4) continue;
        acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(av0[s2], bv[s2], acc0, 0, 0, 0);
        if (two) acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(av1[s2], bv[s2], acc1, 0, 0, 0);
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      // D layout: col = lane & 15, row = (lane >> 4) + 4 * reg
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        Xt0[(kq + 4 * r) * ULD + 16 * b + arow] = acc0[r];
        if (two) Xt1[(kq + 4 * r) * ULD + 16 * b + arow] = acc1[r];
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    }
  }
  wg_barrier<LDS_ONLY>();
  VUS_TMARK(4);
}

// Stage the panel's diagonal block and n_tiles (1 or 2) 48-row tiles of the panel columns in LDS (rows
// past the window or left of the band are zero), then solve the tiles in place: X = A * L_D^-T.
// Block forward substitution on the matrix cores, 16 columns at a time: with M_b = (16x16 diagonal
// block b of L_D)^-1 and G_b = -M_b * L_D[row block b][columns < 16b] (written over L_D in LDS),
//   X_b = [X_0 .. X_b-1] * G_b^T + A_b * M_b^T
// is 4b + 4 steps of v_mfma_f64_16x16x4_f64 per 16-row tile.  A 16-row tile belongs to one wave from
// start to end, so the three block steps need no workgroup barrier.
__device__ __forceinline__ void stage_and_solve(const double* __restrict__ Sb, int band, int k0, int pb, int i_last,
                                                int pose0_a, int pose0_b, int n_tiles, double* __restrict__ Xa,
                                                double* __restrict__ Xb, double* __restrict__ sL,
                                                double* __restrict__ sM, double* __restrict__ sInv) {
  const int tid = threadIdx.x;
  {
    // items = (tile, scalar row lr, panel pose kk): 6 contiguous doubles each; tile 0 = the diagonal block.
    // Every global load is in flight before the first LDS store.
    constexpr int ITEMS = UT * PB;                 // per tile
    constexpr int XU = (3 * ITEMS + 255) / 256;
    d2a_t ld[XU][3];
#pragma unroll
    for (int u = 0; u < XU; ++u) {
      const int item = tid + 256 * u;
      const long long o_item = bandidx::stage_item(band, k0, pb, i_last, pose0_a, pose0_b, n_tiles, item);
      const bool have = o_item >= 0;
      const double* src = have ? Sb + o_item : Sb;
      const d2a_t* s2 = reinterpret_cast<const d2a_t*>(src);
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        ld[u][c] = s2[have ? c : 0];
        if (!have) ld[u][c] = d2a_t{0.0, 0.0};
      }
    }
#pragma unroll
    for (int u = 0; u < XU; ++u) {
      const int item = tid + 256 * u;
      const int tile = (item >= ITEMS) + (item >= 2 * ITEMS);
      const int e = item - ITEMS * tile;
      const int lr = e >> 3, kk = e & 7;
      if (item < (1 + n_tiles) * ITEMS) {
        double v[6] = {ld[u][0].x, ld[u][0].y, ld[u][1].x, ld[u][1].y, ld[u][2].x, ld[u][2].y};
        double* dst;
        if (tile == 0) {
          const int ii = lr / 6, rr = lr - 6 * ii;
          dst = sL + lr * LDD + 6 * kk;
          if (kk == ii) {   // strict upper part of the diagonal 6x6 block is not part of L
            double d = 1.0;
#pragma unroll
            for (int c = 0; c < 6; ++c) {
              if (c == rr) d = v[c];
              if (c > rr) v[c] = 0.0;
            }
            sInv[lr] = (ii < pb) ? 1.0 / d : 1.0;
          }
        } else {
          dst = (tile == 1 ? Xa : Xb) + lr * ULD + 6 * kk;
        }
#pragma unroll
        for (int c = 0; c < 6; ++c) dst[c] = v[c];
      }
    }
  }
  __syncthreads();
  VUS_TMARK(1);
  block_inverses(sL, sM, sInv);
  solve_rows(n_tiles, Xa, Xb, sL, sM);
}

__global__ __launch_bounds__(256) void chol_trsm_update_kernel(BandSet S, int band, int k0, int n_update, int k0_prev,
                                                               int n_rhs, int factor_next) {
  __shared__ __attribute__((aligned(16))) double Xi[UT * ULD];
  __shared__ double Xj[UT * ULD];
  __shared__ double sL[NB * LDD];
  __shared__ double sM[3 * 16 * MLD];
  __shared__ double sInv[NB];
  __shared__ int s_bad;
  VUS_TMARK(0);
  const int tid = threadIdx.x, lane = tid & 63, wave = wave_index();
  // two systems: even blocks serve system 0, odd blocks system 1, so that both critical workgroups (bid 0) are
  // among the first blocks dispatched
  const int sysi = S.count == 2 ? (int)(blockIdx.x & 1) : 0;
  const int bid = S.count == 2 ? (int)(blockIdx.x >> 1) : (int)blockIdx.x;
  double* __restrict__ Sb = S.s[sysi].Sb;
  double* __restrict__ yv = S.s[sysi].y;
  int* __restrict__ status = S.s[sysi].status;
  const int n_poses = S.s[sysi].n;
  const size_t ystride = 6 * (size_t)n_poses;
  if (bid >= n_update) {
    // write-back of the previous panel's rows: tile t of its window
    const int t = bid - n_update;
    const int pbp = min(PB, n_poses - k0_prev);
    const int i_first = k0_prev + pbp;
    const int i_last = min(n_poses - 1, k0_prev + pbp - 1 + band);
    const int p0 = i_first + t * UTP;
    stage_and_solve(Sb, band, k0_prev, pbp, i_last, p0, p0, 1, Xi, Xi, sL, sM, sInv);
    // stored TRANSPOSED ([column][row] inside each 6x6 block): the back-substitution, the only reader,
    // walks these blocks by column
    for (int e = tid; e < UTP * PB * 6; e += 256) {
      const int ii = e / (6 * PB), rem = e - 6 * PB * ii;
      const int kk = rem / 6, c = rem - 6 * kk;
      const long long o_blk = bandidx::solved_item(band, k0_prev, pbp, i_last, p0, e);
      if (o_blk >= 0) {
        double* b = Sb + o_blk;
#pragma unroll
        for (int r = 0; r < 6; ++r) b[r] = Xi[(6 * ii + r) * ULD + 6 * kk + c];
      }
    }
    return;
  }
  const int pb = min(PB, n_poses - k0);
  const int nb = 6 * pb;
  const int i_first = k0 + pb;
  const int i_last = min(n_poses - 1, k0 + pb - 1 + band);
  // tile (ti, tj), tj <= ti, from the linear block index
  int ti = (int)((sqrtf(8.0f * (float)bid + 1.0f) - 1.0f) * 0.5f);
  while ((ti + 1) * (ti + 2) / 2 <= bid) ++ti;
  while (ti * (ti + 1) / 2 > bid) --ti;
  const int tj = bid - ti * (ti + 1) / 2;
  const int pi0 = i_first + ti * UTP, pj0 = i_first + tj * UTP;   // first pose of the tile rows / columns
  const double* Xjj = (ti == tj) ? Xi : Xj;
  const int arow = lane & 15, kq = lane >> 4;
  // this wave's MFMA tiles: accumulate X_i X_j^T - A and store the negation, so the old values enter
  // as the C operand (their loads are in flight during the solve) instead of a read-modify-write tail
  double4_t acc[UQ];
  bool ok[UQ][4];
  long long off[UQ][4];
#pragma unroll
  for (int q = 0; q < UQ; ++q) {
    const int t = wave + 4 * q;
    const int a = t / UMT, b = t - UMT * a;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const long long o_el = t < UMT * UMT ? bandidx::tile_scalar(band, i_last, pi0, pj0, 16 * a + kq + 4 * r, 16 * b + arow) : -1;
      ok[q][r] = o_el >= 0;
      off[q][r] = ok[q][r] ? o_el : 0;
      acc[q][r] = -Sb[off[q][r]];
    }
  }
  if (ti == tj)   // the panel's solved right-hand sides, for the y update below (Xj is unused by a diagonal tile)
    for (int e = tid; e < NB * n_rhs; e += 256) {
      const int q = e / NB, c = e - NB * q;
      Xj[e] = c < nb ? yv[(size_t)q * ystride + 6 * (size_t)k0 + c] : 0.0;
    }
  stage_and_solve(Sb, band, k0, pb, i_last, pi0, pj0, ti == tj ? 1 : 2, Xi, Xj, sL, sM, sInv);
#pragma unroll
  for (int q = 0; q < UQ; ++q) {
    const int t = wave + 4 * q;
    const int a = t / UMT, b = t - UMT * a;
    if (t >= UMT * UMT || (ti == tj && b > a)) continue;   // strictly upper tiles of a diagonal workgroup
    const double* pa = Xi + (16 * a + arow) * ULD + kq;
    const double* pbm = Xjj + (16 * b + arow) * ULD + kq;
#pragma unroll
    for (int s = 0; s < NB / 4; ++s) acc[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(pa[4 * s], pbm[4 * s], acc[q], 0, 0, 0);
  }
  VUS_TMARK(5);
  // C/D layout (f64): col = lane & 15, row = (lane >> 4) + 4 * reg
#pragma unroll
  for (int q = 0; q < UQ; ++q) {
    const int t = wave + 4 * q;
    const int a = t / UMT, b = t - UMT * a;
    if (t >= UMT * UMT || (ti == tj && b > a)) continue;
#pragma unroll
    for (int r = 0; r < 4; ++r)
      if (ok[q][r]) Sb[off[q][r]] = -acc[q][r];
    if (bid == 0 && factor_next) {   // tile (0,0) also leaves its result in LDS (sL is free) for the panel factorisation below
      const int Cc = 16 * b + arow;
#pragma unroll
      for (int r = 0; r < 4; ++r) sL[(16 * a + kq + 4 * r) * LDD + Cc] = -acc[q][r];
    }
  }
  VUS_TMARK(6);
  if (ti == tj && tid < UT) {
    const int i = pi0 + tid / 6;
    if (i <= i_last)
      for (int q = 0; q < n_rhs; ++q) {
        double* yq = yv + (size_t)q * ystride;
        double acc = 0.0;
#pragma unroll 8
        for (int c = 0; c < NB; ++c) acc += Xi[tid * ULD + c] * Xj[NB * q + c];
        const double ynew = yq[6 * (size_t)i + (tid % 6)] - acc;
        yq[6 * (size_t)i + (tid % 6)] = ynew;
        if (bid == 0 && factor_next) Xj[NB * BS_RHS_MAX + NB * q + tid] = ynew;   // the next panel's right-hand-side rows
      }
  }
  VUS_TMARK(7);
  if (bid == 0 && factor_next) {
    // Tile (0,0) is the next panel's diagonal block, complete once this workgroup has stored it: factor it
    // here instead of in a launch of its own (the other ~400 workgroups of this launch take as long anyway).
    __syncthreads();                   // the LDS copies are complete; the X tiles in LDS are dead
    panel_factor<true>(Sb, n_poses, band, i_first, yv, ystride, n_rhs, status,
                       reinterpret_cast<double(*)[64 * 6]>(Xi), s_bad, sL, Xj + NB * BS_RHS_MAX,
                       min(PB, i_last - i_first + 1));
#ifdef VUS_TIMING
    if (threadIdx.x == 0 && k0 == 800) {
      const unsigned long long t8 = __builtin_amdgcn_s_memtime();
      printf("TM %llu %llu %llu %llu %llu %llu %llu %llu\n", g_tm[1] - g_tm[0], g_tm[2] - g_tm[1], g_tm[3] - g_tm[2],
             g_tm[4] - g_tm[3], g_tm[5] - g_tm[4], g_tm[6] - g_tm[5], g_tm[7] - g_tm[6], t8 - g_tm[7]);
    }
#endif
  }
}

// ---- the same panel step as TWO launches (used when two systems share the launches, factor_launches) ----------
// chol_trsm_update_kernel makes every update tile solve the two row tiles it needs: with ~28 row tiles that is 28x
// redundant work and 63 KB of LDS per workgroup (two workgroups per CU).  One system per launch hides that behind
// tile (0,0)'s longer dependent chain; two systems per launch do not (868 workgroups on 512 slots = two rounds).
// Here the rows are solved ONCE per panel by a small launch (one workgroup per 48-row tile: X = A L_D^-T on the
// matrix cores, written back in place already transposed for the back-substitution, right-hand sides updated), and
// the update launch only stages two solved tiles and runs the SYRK: 37 KB of LDS, four workgroups per CU, every
// tile of both systems resident in one round.
__global__ __launch_bounds__(256) void chol_trsm_kernel(BandSet S, int band, int k0, int n_rhs) {
  __shared__ __attribute__((aligned(16))) double Xi[UT * ULD];
  __shared__ double sL[NB * LDD];
  __shared__ double sM[3 * 16 * MLD];
  __shared__ double sInv[NB];
  __shared__ double s_y[BS_RHS_MAX][NB];
  const int tid = threadIdx.x;
  const int sysi = S.count == 2 ? (int)(blockIdx.x & 1) : 0;
  const int t = S.count == 2 ? (int)(blockIdx.x >> 1) : (int)blockIdx.x;
  double* __restrict__ Sb = S.s[sysi].Sb;
  double* __restrict__ yv = S.s[sysi].y;
  const int n_poses = S.s[sysi].n;
  const size_t ystride = 6 * (size_t)n_poses;
  const int pb = min(PB, n_poses - k0);
  const int nb = 6 * pb;
  const int i_first = k0 + pb;
  const int i_last = min(n_poses - 1, k0 + pb - 1 + band);
  const int p0 = i_first + t * UTP;
  for (int e = tid; e < NB * n_rhs; e += 256) {      // the panel's solved right-hand sides
    const int q = e / NB, c = e - NB * q;
    s_y[q][c] = c < nb ? yv[(size_t)q * ystride + 6 * (size_t)k0 + c] : 0.0;
  }
  stage_and_solve(Sb, band, k0, pb, i_last, p0, p0, 1, Xi, Xi, sL, sM, sInv);
  // X back in place, TRANSPOSED inside each 6x6 block ([column][row]): the SYRK launch and the back-substitution
  // both read these blocks by column
  for (int e = tid; e < UTP * PB * 6; e += 256) {
    const int ii = e / (6 * PB), rem = e - 6 * PB * ii;
    const int kk = rem / 6, c = rem - 6 * kk;
    const long long o_blk = bandidx::solved_item(band, k0, pb, i_last, p0, e);
    if (o_blk >= 0) {
      double* b = Sb + o_blk;
#pragma unroll
      for (int r = 0; r < 6; ++r) b[r] = Xi[(6 * ii + r) * ULD + 6 * kk + c];
    }
  }
  // y_i -= X_i y_panel: four threads per row, twelve columns each, summed over the 4-lane group
  {
    const int row = tid >> 2, part = tid & 3;
    const int i = p0 + row / 6;
    for (int q = 0; q < n_rhs; ++q) {
      double acc = 0.0;
      if (row < UT) {
#pragma unroll
        for (int c2 = 0; c2 < NB / 4; ++c2) acc += Xi[row * ULD + 12 * part + c2] * s_y[q][12 * part + c2];
      }
      acc += __shfl_xor(acc, 1);
      acc += __shfl_xor(acc, 2);
      if (row < UT && part == 0 && i <= i_last) yv[(size_t)q * ystride + 6 * (size_t)i + (row % 6)] -= acc;
    }
  }
}

// Stage one 48-row tile of SOLVED rows (transposed 6x6 blocks in memory) into LDS, row-major with stride ULD.
__device__ __forceinline__ void stage_solved_tile(const double* __restrict__ Sb, int band, int k0, int pb, int i_last,
                                                  int pose0, double* __restrict__ X, int tid, int first, int step) {
  for (int item = first; item < UTP * PB * 6; item += step) {      // (pose ii, panel pose kk, column c): 6 rows
    const int ii = item / (6 * PB), rem = item - 6 * PB * ii;
    const int kk = rem / 6, c = rem - 6 * kk;
    const long long o_blk = bandidx::solved_item(band, k0, pb, i_last, pose0, item);
    const bool have = o_blk >= 0;
    const d2a_t* src = reinterpret_cast<const d2a_t*>(have ? Sb + o_blk : Sb);
    d2a_t v0 = src[0], v1 = src[have ? 1 : 0], v2 = src[have ? 2 : 0];
    if (!have) v0 = v1 = v2 = d2a_t{0.0, 0.0};
    double* dst = X + (6 * ii) * ULD + 6 * kk + c;
    dst[0] = v0.x; dst[ULD] = v0.y; dst[2 * ULD] = v1.x; dst[3 * ULD] = v1.y; dst[4 * ULD] = v2.x; dst[5 * ULD] = v2.y;
  }
}

__global__ __launch_bounds__(256, 4) void chol_syrk_kernel(BandSet S, int band, int k0, int n_rhs, int factor_next) {
  __shared__ __attribute__((aligned(16))) double Xi[UT * ULD];
  __shared__ double Xj[UT * ULD];
  __shared__ int s_bad;
  const int tid = threadIdx.x, lane = tid & 63, wave = wave_index();
  const int sysi = S.count == 2 ? (int)(blockIdx.x & 1) : 0;
  const int bid = S.count == 2 ? (int)(blockIdx.x >> 1) : (int)blockIdx.x;
  double* __restrict__ Sb = S.s[sysi].Sb;
  double* __restrict__ yv = S.s[sysi].y;
  int* __restrict__ status = S.s[sysi].status;
  const int n_poses = S.s[sysi].n;
  const size_t ystride = 6 * (size_t)n_poses;
  const int pb = min(PB, n_poses - k0);
  const int i_first = k0 + pb;
  const int i_last = min(n_poses - 1, k0 + pb - 1 + band);
  int ti = (int)((sqrtf(8.0f * (float)bid + 1.0f) - 1.0f) * 0.5f);
  while ((ti + 1) * (ti + 2) / 2 <= bid) ++ti;
  while (ti * (ti + 1) / 2 > bid) --ti;
  const int tj = bid - ti * (ti + 1) / 2;
  const int pi0 = i_first + ti * UTP, pj0 = i_first + tj * UTP;
  const double* Xjj = (ti == tj) ? Xi : Xj;
  const int arow = lane & 15, kq = lane >> 4;
  // The 48x48 tile in memory: for pose row i the eight blocks (i, pj0 .. pj0+7) are CONTIGUOUS (block (i, j) sits at
  // slot i - j of row i), 288 doubles starting at block (i, pj0 + 7).  The old values are fetched as 16-byte vectors
  // (in flight during the staging and the products), the products go through LDS, and the tile leaves as 16-byte
  // vectors again: the MFMA accumulator layout (one column per lane, rows 4 apart) never touches memory.
  constexpr int CV = (UTP * 8 * 18 + 255) / 256;       // 16-byte vectors of the tile per thread (1152 / 256 -> 5)
  d2a_t oldv[CV];
  unsigned vmask[CV];                                  // bit 0/1: element 0/1 of the vector is part of the band's lower part
#pragma unroll
  for (int u = 0; u < CV; ++u) {
    unsigned m;
    const long long o_vec = bandidx::tile_vec(band, i_last, pi0, pj0, tid + 256 * u, m);
    vmask[u] = m;
    oldv[u] = d2a_t{0.0, 0.0};
    if (m) oldv[u] = *reinterpret_cast<const d2a_t*>(Sb + o_vec);
  }
  if (ti == tj) {
    stage_solved_tile(Sb, band, k0, pb, i_last, pi0, Xi, tid, tid, 256);
  } else {       // two tiles: half of the workgroup each
    stage_solved_tile(Sb, band, k0, pb, i_last, tid < 128 ? pi0 : pj0, tid < 128 ? Xi : Xj, tid, tid & 127, 128);
  }
  __syncthreads();
  double4_t acc[UQ];
#pragma unroll
  for (int q = 0; q < UQ; ++q) {
    acc[q] = double4_t{0.0, 0.0, 0.0, 0.0};
    const int t = wave + 4 * q;
    const int a = t / UMT, b = t - UMT * a;
    if (t >= UMT * UMT || (ti == tj && b > a)) continue;
    const double* pa = Xi + (16 * a + arow) * ULD + kq;
    const double* pbm = Xjj + (16 * b + arow) * ULD + kq;
#pragma unroll
    for (int s2 = 0; s2 < NB / 4; ++s2) acc[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(pa[4 * s2], pbm[4 * s2], acc[q], 0, 0, 0);
  }
  __syncthreads();                     // every wave has read its X fragments: both LDS tiles are free
  // products -> LDS (Xj), row-major with stride LDD; C/D layout (f64): col = lane & 15, row = (lane >> 4) + 4 * reg
#pragma unroll
  for (int q = 0; q < UQ; ++q) {
    const int t = wave + 4 * q;
    const int a = t / UMT, b = t - UMT * a;
    if (t >= UMT * UMT || (ti == tj && b > a)) continue;
#pragma unroll
    for (int r = 0; r < 4; ++r) Xj[(16 * a + kq + 4 * r) * LDD + 16 * b + arow] = acc[q][r];
  }
  __syncthreads();
  const bool crit = bid == 0 && factor_next;
  // Two passes: all new values first, then the stores back to back.  In one loop every store was preceded by an
  // s_waitcnt vmcnt(0) (for the old value it combines), which on this in-order counter also waits for the PREVIOUS
  // store to be acknowledged: five serialised round trips per thread, on the critical tile too.  The explicit wait
  // tells the compiler that the old values (loaded under per-vector conditions long ago) have all arrived.
  __builtin_amdgcn_s_waitcnt(0x0F70);    // vmcnt(0)
#pragma unroll
  for (int u = 0; u < CV; ++u) {
    if (!vmask[u]) continue;
    const int v = tid + 256 * u;
    const int ii = v / 144, w = v - 144 * ii;
    const int o = w / 18, e = 2 * (w - 18 * o);
    const int rr = e / 6, c = e - 6 * rr;
    double* d = Xj + (6 * ii + rr) * LDD + 6 * (7 - o) + c;
    d2a_t nv = oldv[u];
    if (vmask[u] & 1u) nv.x -= d[0];
    if (vmask[u] & 2u) nv.y -= d[1];
    oldv[u] = nv;
    if (crit) { d[0] = nv.x; d[1] = nv.y; }      // tile (0,0): the updated block stays in LDS for the factorisation
  }
#pragma unroll
  for (int u = 0; u < CV; ++u) {
    if (!vmask[u]) continue;
    unsigned m;
    const long long o_vec = bandidx::tile_vec(band, i_last, pi0, pj0, tid + 256 * u, m);
    *reinterpret_cast<d2a_t*>(Sb + o_vec) = oldv[u];
  }
  if (crit) {
    // Tile (0,0) is the next panel's diagonal block, complete once this workgroup has stored it: factor it here.
    // Its right-hand-side rows (already updated by the TRSM launch) go behind the factorisation's ring in Xi.
    double* s_rhs = Xi + 3 * 64 * 6;   // Xi holds 2352 doubles, the ring of three panels 1152, the rows <= 384
    for (int e = tid; e < NB * n_rhs; e += 256) {
      const int q = e / NB, c = e - NB * q;
      const int i = i_first + c / 6;
      s_rhs[e] = i <= i_last ? yv[(size_t)q * ystride + 6 * (size_t)i + (c % 6)] : 0.0;
    }
    __syncthreads();
    panel_factor<true>(Sb, n_poses, band, i_first, yv, ystride, n_rhs, status, reinterpret_cast<double(*)[64 * 6]>(Xi), s_bad,
                       Xj, s_rhs, min(PB, i_last - i_first + 1));
  }
}

// x = L^-T y in place (yv), from the last panel to the first.
//
// One compute unit cannot stream the factor fast enough (a single CU sustains ~35 GB/s from HBM, the
// factor of a 2000-pose / band-224 problem is 130 MB), so the sweep is spread over one workgroup per
// 8-pose row group of the band:
//   workgroup 0 (the solver) owns the sequential part: per panel p (last to first) it takes y_p once the
//     contributions to it have arrived (awaited and loaded one panel ahead by its eighth wave), multiplies it
//     with the inverted 48x48 diagonal block (lane = row, the lane's column of L_pp^-1 in registers; bands
//     narrower than a panel: substitution with the pre-scaled column of L_pp instead), publishes x_p, and
//     computes the contribution of x_p to the panel right above itself;
//   workgroup g >= 1 waits for x_p, multiplies it with the blocks L(panel p, panel p - g - 1)^T --
//     transposed on write-back, so an output reads one contiguous 48-byte block row -- and adds the
//     result to y with f64 atomics.  Its operands are loaded one panel ahead of the x it waits for.
// Flags (agent-scope atomics, see below): F[0] = panels solved, F[1] = abort, F[2 + g] = panels done by
// workgroup g.  Every wait is bounded: a wait that expires raises the abort flag, all loops drain and
// status = -1.
constexpr int BS_MAX_RHS = 8;
constexpr int CB_THREADS = 512;          // 8 solver waves at most; threads < 8 * 48 = (panel row kk, output 6a + c)
constexpr int CB_SPIN_LIMIT = 1 << 22;
#ifndef VUS_CB_MAX_WG
#define VUS_CB_MAX_WG 1024
#endif
constexpr int CB_MAX_WG = VUS_CB_MAX_WG;    // solver + helpers

// Inter-workgroup protocol (MI355X_MICROARCH.md, "Valid forms": 8-byte agent-scope atomics on BOTH sides):
// every word that crosses workgroups (x, y, the flags) is written and read ONLY with agent-scope atomic
// stores / loads / adds, which are performed at the coherence point (sc1: they bypass the reader's L1 and
// are never left dirty in a non-coherent cache); a producer drains its own vector-memory operations
// (s_waitcnt vmcnt(0)), the workgroup meets at a barrier, THEN one lane raises the flag; a consumer polls the
// flag with relaxed sc1 loads from one lane, the workgroup meets at a barrier, and only then are the words
// loaded (again sc1).  The factor L itself was written by earlier KERNELS and is read with plain loads.
// No __threadfence(): its L2 write-back / invalidate costs ~10 us per panel here and orders nothing more.
__device__ __forceinline__ int cb_load(const int* f) { return __hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void cb_drain() {   // every vector-memory operation of this wave has completed
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __builtin_amdgcn_s_waitcnt(0);
}

// true once *f >= need; false after an abort (raised here when the wait expires)
__device__ __forceinline__ bool cb_wait(const int* f, int need, int* abort_flag, int limit = CB_SPIN_LIMIT) {
  for (int it = 0; it < limit; ++it) {
    if (cb_load(f) >= need) return true;
    if (__hip_atomic_load(abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) return false;
    __builtin_amdgcn_s_sleep(1);
  }
  __hip_atomic_store(abort_flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  return false;
}

// The diagonal panels are INVERTED in place before the sweep (diag_invert_kernel: L_pp^-1 over L_pp, same block
// layout, one wave per panel, all panels at once), so that the solver's x_p = L_pp^-T y' is 48 independent
// multiply-adds per lane instead of a 48-step substitution with two v_readlane and one v_fma_f64 per dependent step
// (instrumented: 3.7 k of the step's 12 k cycles, plus 2.9 k for fetching and scaling the strided column of L).
// Slot t of the lower block triangle of a panel: block row r6, distance sd from the diagonal block, element e.
constexpr int CB_DIAG_ELEMS = 36 * (PB * (PB + 1) / 2);
constexpr int CB_DIAG_PER_LANE = (CB_DIAG_ELEMS + 63) / 64;
static_assert(CB_DIAG_ELEMS == bandidx::DIAG_ELEMS && PB == bandidx::PB && NB == bandidx::NB && UTP == bandidx::UTP, "band_index.h");

__global__ __launch_bounds__(64) void diag_invert_kernel(BandSet S, int band, int n_solve) {
  const int sysi = blockIdx.y, p = blockIdx.x, lane = threadIdx.x;
  double* Sb = S.s[sysi].Sb;
  const int n_poses = n_solve > 0 ? n_solve : S.s[sysi].n;
  const int k0 = PB * p;
  if (k0 >= n_poses) return;
  const int nb = 6 * min(PB, n_poses - k0);
  __shared__ double sL[NB * (NB + 1)];      // L_pp, dense, row stride NB + 1
  __shared__ double sI[NB * (NB + 1)];      // its inverse
  for (int t = lane; t < NB * (NB + 1); t += 64) {
    sL[t] = 0.0;
    sI[t] = 0.0;
  }
  __syncthreads();
  const double* src[CB_DIAG_PER_LANE];
  double v[CB_DIAG_PER_LANE];
  unsigned have = 0;
#pragma unroll
  for (int j = 0; j < CB_DIAG_PER_LANE; ++j) {
    const int t = lane + 64 * j;
    const long long o_el = bandidx::diag_elem(band, k0, nb, t);
    have |= (unsigned)(o_el >= 0) << j;
    src[j] = o_el >= 0 ? Sb + o_el : Sb;
  }
#pragma unroll
  for (int j = 0; j < CB_DIAG_PER_LANE; ++j) v[j] = *src[j];
#pragma unroll
  for (int j = 0; j < CB_DIAG_PER_LANE; ++j) {
    const int t = lane + 64 * j;
    int r6, sd, e;
    bandidx::diag_slot(t < CB_DIAG_ELEMS ? t : 0, r6, sd, e);
    if ((have >> j) & 1) sL[(6 * r6 + e / 6) * (NB + 1) + 6 * (r6 - sd) + e % 6] = v[j];
  }
  __syncthreads();
  __shared__ double sR[NB];                 // 1 / L_rr
  if (lane < NB) sR[lane] = lane < nb ? 1.0 / sL[lane * (NB + 2)] : 0.0;
  __syncthreads();
  // lane j: column j of the inverse by forward substitution, x_r = (delta_rj - sum_{k<r} L_rk x_k) / L_rr
  if (lane < nb) {
    double x[NB];
#pragma unroll
    for (int r = 0; r < NB; ++r) {
      double acc = r == lane ? 1.0 : 0.0;
#pragma unroll
      for (int k = 0; k < r; ++k) acc -= sL[r * (NB + 1) + k] * x[k];
      x[r] = r >= lane ? acc * sR[r] : 0.0;
      sI[r * (NB + 1) + lane] = x[r];
    }
  }
  __syncthreads();
#pragma unroll
  for (int j = 0; j < CB_DIAG_PER_LANE; ++j) {
    const int t = lane + 64 * j;
    int r6, sd, e;
    bandidx::diag_slot(t < CB_DIAG_ELEMS ? t : 0, r6, sd, e);
    if ((have >> j) & 1)
      Sb[bandidx::diag_elem(band, k0, nb, t)] = sI[(6 * r6 + e / 6) * (NB + 1) + 6 * (r6 - sd) + e % 6];
  }
}

// column `lane` of the diagonal block, pre-scaled: Lp[c] = L[c][lane] / L[lane][lane] for c > lane, else 0
__device__ __forceinline__ void cb_load_diag(const double* __restrict__ Sb, int band, int k0, int nb, int lane,
                                             double (&Lp)[NB], double& dinv) {
#pragma unroll
  for (int c = 0; c < NB; ++c) {
    const long long o_el = bandidx::cb_diag(band, k0, nb, lane, c);
    const double* src = o_el >= 0 ? Sb + o_el : Sb;
    Lp[c] = *src;
    if (o_el < 0) Lp[c] = 0.0;
  }
  const long long o_piv = bandidx::cb_diag_pivot(band, k0, nb, lane);
  const double dg = o_piv >= 0 ? Sb[o_piv] : 1.0;
  dinv = 1.0 / dg;
#pragma unroll
  for (int c = 0; c < NB; ++c) Lp[c] *= dinv;
}

// column `lane` of the inverted diagonal panel: Lp[r] = (L_pp^-1)[r][lane] for r >= lane, else 0.  (Only called
// with band >= PB - 1: every block of the panel's lower triangle is stored.)  One per-lane base pointer, the rest
// of every address is the same for all lanes.
__device__ __forceinline__ void cb_load_inv(const double* __restrict__ Sb, int band, int k0, int nb, int lane,
                                            double (&Lp)[NB]) {
  // entries above the diagonal (and rows >= nb of a short last panel) are not stored: those lanes read the panel's
  // first element instead (bandidx::cb_inv_safe; an earlier version formed an address 2 KB in front of the band for
  // them).  One per-lane base, the rest of every address is the same for all lanes; all 48 loads are issued before
  // the first result is looked at.
  const double* safe = Sb + bandidx::cb_inv_safe(band, k0);
  const double* col = Sb + bandidx::cb_inv_base(band, k0, lane);      // may point in front of the panel: never used alone
#pragma unroll
  for (int r = 0; r < NB; ++r) {
    const double* src = bandidx::cb_inv_stored(nb, lane, r) ? col + bandidx::cb_inv_delta(band, r) : safe;
    Lp[r] = *src;
  }
#pragma unroll
  for (int r = 0; r < NB; ++r)
    if (!(r < nb && lane <= r)) Lp[r] = 0.0;
}

__global__ __launch_bounds__(CB_THREADS) void chol_backsolve_kernel(BandSet S, int band, int n_rhs, int n_groups, int n_solve,
                                                                   int inverted) {
  // workgroup index g inside its system: 0 = solver, w >= 1 serves row groups w, w + n_wg - 1, ...; with two systems
  // (the two halves of the two-sided solve) consecutive blocks alternate between them.  The workgroups spread over
  // all XCDs: everything they exchange goes through agent-scope atomics, i.e. through the memory side, whether or not
  // they share an L2, and ONE XCD holds 24 of these 512-thread blocks -- fewer than configs[2] has row groups, which
  // gave every helper two or three groups per panel and made the helpers the pace of the sweep.
  const int wg = blockIdx.x;
  const int sysi = S.count == 2 ? (wg & 1) : 0;
  const int g = S.count == 2 ? (wg >> 1) : wg;
  const int n_helpers = (int)gridDim.x / S.count - 1;
  const double* __restrict__ Sb = S.s[sysi].Sb;
  double* yv = S.s[sysi].y;
  int* F = S.s[sysi].F;
  int* __restrict__ status = S.s[sysi].status;
  // n_solve > 0: only the leading n_solve poses are back-substituted (the eliminated part of a partial factorisation);
  // the right-hand sides keep the row stride of the whole system
  const int n_poses = n_solve > 0 ? n_solve : S.s[sysi].n;
  const size_t ystride = 6 * (size_t)S.s[sysi].n;
  __shared__ double s_x[BS_MAX_RHS][NB];
  __shared__ double s_part[BS_MAX_RHS][PB][NB];
  __shared__ double s_y[2][BS_MAX_RHS][NB];       // solver: y of this panel / of the next one (look-ahead wave)
  __shared__ double s_inv[NB * LDD];              // solver: the inverted diagonal panel the next step solves with
  __shared__ int s_go, s_go2[2];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int NP = (n_poses + PB - 1) / PB;
  const int kk = tid / NB, oc = tid - NB * kk;    // update task: panel row kk, output oc = 6a + c
  const int a = oc / 6, c = oc - 6 * a;
  int* abort_flag = F + 1;
  // this thread's block row for panel p: row c of the transposed block (8p + kk, 8p - 8g - 8 + a)
  d2a_t l[3];
  bool have;
#define CB_LOAD_ROWS_G(P, G)                                                                           \
  {                                                                                                    \
    const long long o_ = bandidx::cb_rows(band, n_poses, (P), (G), kk, a, c);                          \
    have = o_ >= 0;                                                                                    \
    const d2a_t* src_ = reinterpret_cast<const d2a_t*>(have ? Sb + o_ : Sb);                           \
    l[0] = src_[0]; l[1] = src_[1]; l[2] = src_[2];                                                    \
  }
#define CB_PARTIAL_DOTS()                                                                              \
  for (int q = 0; q < (kk < PB ? n_rhs : 0); ++q) {                                                                    \
    const double* xq = &s_x[q][6 * kk];                                                                \
    const double d = l[0].x * xq[0] + l[0].y * xq[1] + l[1].x * xq[2] + l[1].y * xq[3] + l[2].x * xq[4] + l[2].y * xq[5]; \
    s_part[q][kk][oc] = have ? d : 0.0;                                                                \
  }
#define CB_LOAD_ROWS(P) CB_LOAD_ROWS_G(P, g)
  CB_LOAD_ROWS(NP - 1);
  if (g == 0) {
    // ---- solver ----
    // Two of the step's round trips to the coherence point are taken off its chain: while the solving waves work on
    // panel p, the last wave (idle otherwise: the products use threads < 8 * 48) waits for the flags of every
    // workgroup that adds to y of panel p-1 and loads that y into LDS.  (Those flags only depend on x of panels
    // published in EARLIER steps, so the wait cannot depend on this step's own flag.)
    constexpr int LA = CB_THREADS / 64 - 1;
    if (wave == LA) {              // y of the last panel: nobody adds to it
      const int k0 = PB * (NP - 1), nb = 6 * min(PB, n_poses - k0);
      for (int t = lane; t < NB * n_rhs; t += 64) {
        const int q = t / NB, r = t - NB * q;
        s_y[0][q][r] = r < nb ? __hip_atomic_load(&yv[(size_t)q * ystride + 6 * k0 + r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.0;
      }
      if (lane == 0) s_go2[0] = 1;
    }
    __syncthreads();
    if (inverted) {
      // ---- pipelined sweep (diagonal panels inverted in place) ----
      // The inverted panel of step s+1 travels while step s runs: threads < 8 * 48 request its 2304 elements (six each)
      // at the END of step s-1, park them in LDS at the end of step s, and the solving waves read their column from LDS
      // at the start of step s+1.  Loaded into registers after the step's flag, as before, the 48 loads were a memory
      // round trip on the chain: 4.2 k of the step's 6.9 k cycles (tools/backsolve_timing.py).  The step's last barrier
      // is LDS-only, so that the requests stay in flight across it.
      const int pc = tid % NB, prg = tid / NB;           // prefetch task: column pc, rows prg + 8 i
      double pre[6];
      // the thread's six offsets inside a panel, computed once (cb_inv = panel base + column part + row part)
      int rel[6];
#pragma unroll
      for (int i = 0; i < 6; ++i)
        rel[i] = (int)(bandidx::cb_inv_base(band, 0, pc) + bandidx::cb_inv_delta(band, prg < PB ? prg + 8 * i : 0));
      auto request = [&](int pp) {                      // panel pp's inverse -> pre (masked elements: a safe address)
        const int k0p = PB * pp, nbp = 6 * min(PB, n_poses - k0p);
        const double* base = Sb + bandidx::cb_inv_safe(band, k0p);       // the panel's first element
        const double* pa[6];
#pragma unroll
        for (int i = 0; i < 6; ++i) pa[i] = base + ((prg < PB && bandidx::cb_inv_stored(nbp, pc, prg + 8 * i)) ? rel[i] : 0);
        __builtin_amdgcn_sched_barrier(0);              // every address first, then the loads back to back
#pragma unroll
        for (int i = 0; i < 6; ++i) pre[i] = *pa[i];
      };
      auto park = [&](int pp) {      // after the step's first barrier: the solving waves have their columns in registers
        const int k0p = PB * pp, nbp = 6 * min(PB, n_poses - k0p);
        if (prg < PB) {
#pragma unroll
          for (int i = 0; i < 6; ++i) {
            const int r = prg + 8 * i;
            s_inv[r * LDD + pc] = bandidx::cb_inv_stored(nbp, pc, r) ? pre[i] : 0.0;
          }
        }
      };
      request(NP - 1);
      park(NP - 1);
      if (NP > 1) request(NP - 2);
      __syncthreads();
      for (int s = 0; s < NP; ++s) {
        const int p = NP - 1 - s, k0 = PB * p;
        const int nb = 6 * min(PB, n_poses - k0);
        const int cur = s & 1;
        if (!s_go2[cur]) break;
#ifdef VUS_TIMING
        const bool bm_on = s == 40 && wg == 0 && NP > 60;
#define VUS_BM(k) do { if (bm_on && tid == 0) s_wtm[k] = __builtin_amdgcn_s_memtime(); } while (0)      /* marks land in g_wtm[16 + k] */
#else
#define VUS_BM(k)
#endif
        VUS_BM(0);
        if (wave < n_rhs) {
          double* yq = yv + (size_t)wave * ystride;
          double yr = lane < NB ? s_y[cur][wave][lane] : 0.0;
          const double* invc = &s_inv[lane < NB ? lane : 0];
          double Lc[NB];
#pragma unroll
          for (int r = 0; r < NB; ++r) Lc[r] = invc[r * LDD];
          if (s > 0 && lane < NB) {       // what x of the panel below added (its eight block rows' products)
#pragma unroll
            for (int k2 = 0; k2 < PB; ++k2) yr -= s_part[wave][k2][lane];
          }
          // x_c = sum_r (L_pp^-1)[r][c] y'_r: four independent chains of multiply-adds
          double z4[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
          for (int r = 0; r < NB; ++r) z4[r & 3] += Lc[r] * bcast_lane(yr, r);
          const double z = (z4[0] + z4[1]) + (z4[2] + z4[3]);
          VUS_BM(1);
          if (lane < nb) {
            __hip_atomic_store(&yq[6 * k0 + lane], z, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            s_x[wave][lane] = z;
          } else if (lane < NB) {
            s_x[wave][lane] = 0.0;
          }
          cb_drain();
          VUS_BM(2);
        }
        if (wave == LA && p > 0) {   // look-ahead (after its own solve when all eight waves carry a right-hand side)
          bool ok = true;
          for (int g0 = 1; g0 < n_groups && ok; g0 += 64) {
            const int gg = g0 + lane;
            if (gg < n_groups && s + 1 - gg > 0) ok = cb_wait(F + 2 + gg, s + 1 - gg, abort_flag);
            ok = __all(ok);
          }
#ifdef VUS_TIMING
          if (bm_on && lane == 0) s_wtm[6] = __builtin_amdgcn_s_memtime();
#endif
          if (ok) {
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
            for (int t = lane; t < NB * n_rhs; t += 64) {     // a full panel: NB rows
              const int q = t / NB, r = t - NB * q;
              s_y[cur ^ 1][q][r] = __hip_atomic_load(&yv[(size_t)q * ystride + 6 * (k0 - PB) + r], __ATOMIC_RELAXED,
                                                     __HIP_MEMORY_SCOPE_AGENT);
            }
          }
          if (lane == 0) s_go2[cur ^ 1] = ok;
#ifdef VUS_TIMING
          if (bm_on && lane == 0) s_wtm[7] = __builtin_amdgcn_s_memtime();
#endif
        }
        __syncthreads();
        VUS_BM(3);
        // the step's flag is raised by a lane of the look-ahead wave: the waves that go on to the products below wait for
        // their block rows with s_waitcnt vmcnt(0), which would also wait for this store's round trip (1 k cycles)
        if (tid == 64 * LA) __hip_atomic_store(F, s + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (p > 0) {   // contribution of x_p to the panel right above (kept in LDS, subtracted at the next step)
          VUS_BM(8);
          CB_PARTIAL_DOTS();
          VUS_BM(4);
          park(p - 1);                          // requested a step ago
          VUS_BM(9);
          CB_LOAD_ROWS(p - 1);
          VUS_BM(10);
          if (p > 1) request(p - 2);
          VUS_BM(11);
        }
        lds_barrier();                          // the requests above stay in flight
#ifdef VUS_TIMING
        if (bm_on && tid == 0) {
          s_wtm[5] = __builtin_amdgcn_s_memtime();
          for (int k = 0; k < 12; ++k) g_wtm[16 + k] = s_wtm[k];
        }
#endif
      }
    } else {
    // ---- narrow bands (< 7 poses): the factor itself, 48-step substitution, operands loaded after the step's flag ----
    double Lp[NB];
    double dinv = 1.0;
    if (wave < n_rhs) {
      const int k0 = PB * (NP - 1);
      cb_load_diag(Sb, band, k0, 6 * min(PB, n_poses - k0), lane, Lp, dinv);
    }
    for (int s = 0; s < NP; ++s) {
      const int p = NP - 1 - s, k0 = PB * p;
      const int nb = 6 * min(PB, n_poses - k0);
      const int cur = s & 1;
      if (!s_go2[cur]) break;
      if (wave < n_rhs) {
        double* yq = yv + (size_t)wave * ystride;
        double yr = lane < NB ? s_y[cur][wave][lane] : 0.0;
        if (s > 0 && lane < NB) {       // what x of the panel below added (its eight block rows' products)
#pragma unroll
          for (int k2 = 0; k2 < PB; ++k2) yr -= s_part[wave][k2][lane];
        }
        double z;
        if (inverted) {                 // x_c = sum_r (L_pp^-1)[r][c] y'_r: independent multiply-adds
          z = 0.0;
#pragma unroll
          for (int r = 0; r < NB; ++r) z += Lp[r] * bcast_lane(yr, r);
        } else {                        // narrow bands: substitution.  z_r = y_r / L_rr;  x_c = z_c once every column > c is applied
          z = yr * dinv;
#pragma unroll
          for (int cc = NB - 1; cc >= 0; --cc)
            if (cc < nb) z -= Lp[cc] * bcast_lane(z, cc);
        }
        if (lane < nb) {
          __hip_atomic_store(&yq[6 * k0 + lane], z, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          s_x[wave][lane] = z;
        } else if (lane < NB) {
          s_x[wave][lane] = 0.0;
        }
        cb_drain();
      }
      if (wave == LA && p > 0) {   // look-ahead (after its own solve when all eight waves carry a right-hand side)
        bool ok = true;
        for (int g0 = 1; g0 < n_groups && ok; g0 += 64) {
          const int gg = g0 + lane;
          if (gg < n_groups && s + 1 - gg > 0) ok = cb_wait(F + 2 + gg, s + 1 - gg, abort_flag);
          ok = __all(ok);
        }
        if (ok) {
          __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
          for (int t = lane; t < NB * n_rhs; t += 64) {     // a full panel: NB rows
            const int q = t / NB, r = t - NB * q;
            s_y[cur ^ 1][q][r] = __hip_atomic_load(&yv[(size_t)q * ystride + 6 * (k0 - PB) + r], __ATOMIC_RELAXED,
                                                   __HIP_MEMORY_SCOPE_AGENT);
          }
        }
        if (lane == 0) s_go2[cur ^ 1] = ok;
      }
      __syncthreads();
      if (tid == 0) __hip_atomic_store(F, s + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (p > 0) {   // contribution of x_p to the panel right above (kept in LDS, subtracted at the next step)
        CB_PARTIAL_DOTS();
        if (wave < n_rhs) {
          if (inverted) cb_load_inv(Sb, band, k0 - PB, NB, lane, Lp);
          else cb_load_diag(Sb, band, k0 - PB, NB, lane, Lp, dinv);
        }
        CB_LOAD_ROWS(p - 1);
      }
      __syncthreads();
    }
    }
  } else {
    // ---- row group gg: panels NP-1 .. gg+1 contribute to panels NP-gg-2 .. 0 ----
    // Normally one row group per workgroup (its block rows prefetched one panel ahead).  Bands wider than
    // 8 * n_helpers poses give a workgroup several groups (gg = g, g + n_helpers, ...): all active workgroups
    // must be resident at once for the flag protocol to make progress, so their number is capped.
    const bool single = g + n_helpers >= n_groups;
    for (int s = 0; s < NP - g - 1; ++s) {
      const int p = NP - 1 - s, k0 = PB * p;
      const int nb = 6 * min(PB, n_poses - k0);
      if (tid == 0) s_go = cb_wait(F, s + 1, abort_flag);
      __syncthreads();
      if (!s_go) break;
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
      if (tid < NB * n_rhs) {
        const int q = tid / NB, r = tid - NB * q;
        s_x[q][r] = r < nb ? __hip_atomic_load(&yv[(size_t)q * ystride + 6 * k0 + r], __ATOMIC_RELAXED,
                                               __HIP_MEMORY_SCOPE_AGENT)
                           : 0.0;
      }
      __syncthreads();
      for (int gg = g; gg < n_groups && s < NP - gg - 1; gg += n_helpers) {
        if (!single) CB_LOAD_ROWS_G(p, gg);
        CB_PARTIAL_DOTS();
        __syncthreads();
        if (tid < NB * n_rhs) {
          const int q = tid / NB, r = tid - NB * q;
          double sum = 0.0;
#pragma unroll
          for (int k2 = 0; k2 < PB; ++k2) sum += s_part[q][k2][r];
          const int row = 6 * (k0 - PB * gg - PB) + r;    // >= 0: this group stops at panel gg + 1
          __hip_atomic_fetch_add(&yv[(size_t)q * ystride + row], -sum, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        cb_drain();
        __syncthreads();
        if (tid == 0) __hip_atomic_store(F + 2 + gg, s + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      if (single && s + 1 < NP - g - 1) CB_LOAD_ROWS(p - 1);
    }
  }
#undef CB_LOAD_ROWS
#undef CB_LOAD_ROWS_G
#undef CB_PARTIAL_DOTS
  if (tid == 0 && __hip_atomic_load(abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) status[0] = -1;
}

// ---------------------------------------------------------------------------------------------------------------
// Persistent window factorisation (factor_launches mode 3): the whole chain of panel steps in ONE launch.
//
// Why: a panel step issued as launches costs a TRSM launch (9 us: every workgroup stages L_D and rebuilds its inverse
// blocks) + an update launch (15 us: tile (0,0)'s workgroup stages, multiplies, combines and THEN factors the next
// panel) + two launch boundaries, all of it on the chain of n / 8 dependent steps; the update's ~400 other tiles take
// as long only because they re-read and re-write the whole 7 MB window from memory every step.  Here
//   * the sliding window lives in REGISTERS: tile (I, J) of 8 x 8 poses sits in the MFMA accumulators of one workgroup
//     from its first update to its elimination (band_index.h, win_tile_of: D + 1 choose 2 slots, each hosting exactly
//     one tile at every step; one workgroup per slot serves that slot of both systems of a two-sided solve);
//   * one CRITICAL workgroup per system owns the chain: factor the diagonal tile (panel_factor), build the inverse
//     blocks ONCE and publish them, solve the sub-diagonal tile, update the next diagonal tile, factor again -- it
//     never waits for the bulk of the window, only for the two tiles of the next block row, which their owners hand
//     over one step ahead;
//   * everything that crosses workgroups (the factor L in its final place in Sband, the inverse blocks, the solved
//     rows X, handed-over tiles, right-hand sides, the flags) is written and read ONLY with agent-scope 8-byte atomics
//     (the protocol of chol_backsolve_kernel: store, drain, barrier, flag / poll, barrier, load), every wait is bounded
//     (abort flag -> status -1, never a hang), and the grid is sized so that every workgroup is resident.
// Flags of a system: F[0] = panels published by the critical workgroup, F[1] = abort, F[2 + I] = panels whose solved
// rows of block row I are in memory, F[2 + NT + I] = tiles of block row I handed over (2 = both).
struct WinSys {
  double* Sb;
  double* y;
  int* status;
  double* pub;      // [n_panels][WIN_PUB]: G(1,0), G(2,0), G(2,1), M_0, M_1, M_2 of every panel, 16 x 16 row-major each
  int* F;
  int n;
};
struct WinSet {
  WinSys s[2];
  int count;
  int spin;         // bound of every wait, in polls: sized from the chain length by window_launch
  int fault;        // tests: the first bulk workgroup returns at once (a workgroup that never became resident)
};
constexpr int WIN_PUB = 6 * 256;
constexpr int WIN_LDS_DOUBLES = 3 * UT * ULD + 3 * 16 * MLD + NB + 4 * BS_RHS_MAX * NB;
constexpr int WIN_MIN_BAND = 2 * PB;      // at least two off-diagonal tile distances, else the critical workgroup owns everything

__device__ __forceinline__ double ld_sc1(const double* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_sc1(double* p, double v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// 16-byte agent-scope (sc1) accesses through a buffer descriptor of the whole band: a third of the memory transactions
// of 8-byte atomics for the 48-byte row segments everything here moves, and a masked lane simply addresses past the
// buffer's end (the load returns zeros, the store is dropped: hardware range check, no branch).
typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
typedef __amdgpu_buffer_rsrc_t rsrc_t;
constexpr unsigned WIN_OOB = 0xFFFFFFF0u;
__device__ __forceinline__ rsrc_t win_rsrc(const void* p, long long bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (int)(unsigned)bytes, 0x00020000);
}
__device__ __forceinline__ unsigned win_off(long long off_doubles) { return off_doubles >= 0 ? (unsigned)(8 * off_doubles) : WIN_OOB; }
__device__ __forceinline__ d2a_t ld16_sc1(rsrc_t r, unsigned byte_off) {
  const u32x4_t v = __builtin_amdgcn_raw_buffer_load_b128(r, (int)byte_off, 0, 16);
  return __builtin_bit_cast(d2a_t, v);
}
__device__ __forceinline__ void st16_sc1(rsrc_t r, unsigned byte_off, d2a_t v) {
  __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, v), r, (int)byte_off, 0, 16);
}

// A 48 x 48 tile moves as the 1152 16-byte vectors of bandidx::win_vec, in address order: consecutive lanes carry
// consecutive 16 bytes of a pose row's 2304 contiguous bytes (whole cache lines per wave instruction), 4.5 vectors per
// thread.  The thread's share of the address arithmetic is done once per launch (WinLane); a tile adds its base.
constexpr int WIN_VPT = (bandidx::WIN_VECS + 255) / 256;        // 5
struct WinLane {
  unsigned rel[WIN_VPT];      // bytes from the tile's base (WIN_OOB: the thread has no such vector)
  unsigned pk[WIN_VPT];       // LDS index of element e as a natural tile (bits 0-11) / as solved rows (12-23), ii (24-26), kk - ii + 8 (27-30)
};
struct WinFetch {
  d2a_t v[WIN_VPT];
};
__device__ __forceinline__ WinLane win_lane(int band) {
  WinLane L;
#pragma unroll
  for (int u = 0; u < WIN_VPT; ++u) {
    const int v = threadIdx.x + 256 * u;
    int ii, kk, e;
    bandidx::win_vec_pos(v, ii, kk, e);
    const unsigned nat = (unsigned)((6 * ii + e / 6) * LDD + 6 * kk + e % 6);       // elements (r, c), (r, c + 1)
    const unsigned xt = (unsigned)((6 * ii + e % 6) * ULD + 6 * kk + e / 6);        // elements (r, c), (r + 1, c): transposed blocks
    L.rel[u] = v < bandidx::WIN_VECS ? (unsigned)(8 * bandidx::win_vec_rel(band, v)) : WIN_OOB;
    L.pk[u] = v < bandidx::WIN_VECS ? nat | xt << 12 | (unsigned)ii << 24 | (unsigned)(kk - ii + 8) << 27 : 8u << 27;
  }
  return L;
}
// byte offset of the thread's vector u of tile (pose rows pi0.., pose columns pj0..) in a matrix of n poses, or WIN_OOB
__device__ __forceinline__ unsigned win_vec_off(const WinLane& L, int u, int band, int n, int pi0, int pj0) {
  const unsigned base = (unsigned)(8 * bandidx::win_vec_base(band, pi0, pj0));      // modulo 2^32; base + rel is exact
  const int ii = (int)(L.pk[u] >> 24 & 7), dk = (int)(L.pk[u] >> 27 & 15) - 8;
  return (L.rel[u] != WIN_OOB && bandidx::win_vec_ok(band, n, pi0, pj0, ii, dk)) ? base + L.rel[u] : WIN_OOB;
}
__device__ __forceinline__ void win_fetch(rsrc_t rs, const WinLane& L, int band, int n, int pi0, int pj0, WinFetch& f) {
#pragma unroll
  for (int u = 0; u < WIN_VPT; ++u) f.v[u] = ld16_sc1(rs, win_vec_off(L, u, band, n, pi0, pj0));
}
// XT = false: a tile of the matrix, row-major in LDS (stride LDD); XT = true: solved rows, whose blocks are stored
// transposed (LDS: row-major X, stride ULD).  Vectors the band does not store were fetched as zeros.
template <bool XT>
__device__ __forceinline__ void win_commit(const WinLane& L, const WinFetch& f, double* T) {
#pragma unroll
  for (int u = 0; u < WIN_VPT; ++u) {
    if (L.rel[u] == WIN_OOB) continue;
    double* dst = T + (XT ? L.pk[u] >> 12 & 0xFFF : L.pk[u] & 0xFFF);
    dst[0] = f.v[u].x;
    dst[XT ? ULD : 1] = f.v[u].y;
  }
}
// EAGER (the critical workgroup): all LDS reads requested before the first store waits for its own; the window slots,
// which keep two tiles in accumulators, have no registers to spare for that.
template <bool XT, bool EAGER = false>
__device__ __forceinline__ void win_store(rsrc_t rs, const WinLane& L, int band, int n, int pi0, int pj0, const double* T) {
  d2a_t v[WIN_VPT];
#pragma unroll
  for (int u = 0; u < WIN_VPT; ++u) {      // a thread without a fifth vector reads element 0 (its store is dropped)
    const double* src = T + (XT ? L.pk[u] >> 12 & 0xFFF : L.pk[u] & 0xFFF);
    v[u] = d2a_t{src[0], src[XT ? ULD : 1]};
    if (!EAGER) st16_sc1(rs, win_vec_off(L, u, band, n, pi0, pj0), v[u]);
  }
  if (EAGER) {
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int u = 0; u < WIN_VPT; ++u) st16_sc1(rs, win_vec_off(L, u, band, n, pi0, pj0), v[u]);
  }
}

// acc (an MFMA accumulator set holding MINUS the tile) += Xi * Xjj^T; a diagonal tile keeps its lower MFMA tiles only
__device__ __forceinline__ void win_mfma_update(double4_t (&acc)[UQ], const double* Xi, const double* Xjj, bool diag) {
  const int lane = threadIdx.x & 63, wave = wave_index();
  const int arow = lane & 15, kq = lane >> 4;
#pragma unroll
  for (int q = 0; q < UQ; ++q) {
    const int t = wave + 4 * q;
    const int a = t / UMT, b = t - UMT * a;
    if (t >= UMT * UMT || (diag && b > a)) continue;
    const double* pa = Xi + (16 * a + arow) * ULD + kq;
    const double* pbm = Xjj + (16 * b + arow) * ULD + kq;
#pragma unroll
    for (int s2 = 0; s2 < NB / 4; ++s2) acc[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(pa[4 * s2], pbm[4 * s2], acc[q], 0, 0, 0);
  }
}

// ---- the critical workgroup of one system ----
__device__ void win_critical(const WinSys& B, int band, int NE, int n_rhs, int spin, double* smem, int& s_bad, int& s_go) {
  const int tid = threadIdx.x, lane = tid & 63, wave = wave_index();
  const int arow = lane & 15, kq = lane >> 4;
  const int n = B.n, NT = (n + PB - 1) / PB;
  double* Xi = smem;                       // sub-diagonal tile A -> X; the ring of panel_factor while a panel is factored
  double* Xn = Xi + UT * ULD;              // the next diagonal tile
  double* sL = Xn + UT * ULD;              // the diagonal tile -> L_D -> (G blocks)
  double* sM = sL + UT * ULD;
  double* sInv = sM + 3 * 16 * MLD;
  double* s_z = sInv + NB;                 // solved right-hand sides of the panel  [q][NB]
  double* s_rhs = s_z + BS_RHS_MAX * NB;   // right-hand-side rows of the next panel  [q][NB]
  int* F = B.F;
  int* abort_flag = F + 1;
  int* xflag = F + 2;
  int* hand = F + 2 + NT;
  const size_t ystride = 6 * (size_t)n;
  const rsrc_t rsS = win_rsrc(B.Sb, 8 * bandidx::band_doubles(n, band));
  const rsrc_t rsP = win_rsrc(B.pub, 8ll * WIN_PUB * NE);
  const WinLane WL = win_lane(band);
  {
    WinFetch f0;
    win_fetch(rsS, WL, band, n, 0, 0, f0);
    win_commit<false>(WL, f0, sL);
  }
  for (int e = tid; e < NB * n_rhs; e += 256) {
    const int q = e / NB, c = e - NB * q;
    s_rhs[e] = c < 6 * n ? ld_sc1(B.y + (size_t)q * ystride + c) : 0.0;
  }
  __syncthreads();
  bool ok = true;
  for (int p = 0; p < NE; ++p) {
    const int k0 = PB * p;
    const int pb = min(PB, n - k0), nb = 6 * pb;
    [[maybe_unused]] const bool vus_wm_on = p == 41 && blockIdx.x == 0;
    VUS_WM(0);
    panel_factor<true, true>(B.Sb, n, band, k0, B.y, ystride, n_rhs, B.status, reinterpret_cast<double(*)[64 * 6]>(Xi), s_bad,
                             sL, s_rhs, pb, sL, s_z, sInv);
    const int I = p + 1;
    const bool more = I < NT;
    // the hand-over flag of block row I is read NOW and looked at after the stores below: the round trip of the usual
    // case (handed over long ago) is hidden
    int hand_seen = 0;
    if (more && I >= 2 && tid == 0) hand_seen = __hip_atomic_load(hand + I, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    lds_barrier();
    win_store<false, true>(rsS, WL, band, n, k0, k0, sL);      // the factor and the solved right-hand sides, on their way to memory
    for (int e = tid; e < NB * n_rhs; e += 256) {
      const int q = e / NB, c = e - NB * q;
      if (c < nb) st_sc1(B.y + (size_t)q * ystride + 6 * (size_t)k0 + c, s_z[e]);
    }
    VUS_WM(4);
    // The next block row's two right-most tiles carry every update before step p; their owners handed them over while
    // this panel was being factored.  Their loads are issued NOW and land while the inverse blocks are computed.
    WinFetch fa, fb;
    double vr[2];
    bool okr[2];
    if (more) {
      if (I >= 2) {
        if (tid == 0) s_go = hand_seen >= 2 || cb_wait(hand + I, 2, abort_flag, spin);
        lds_barrier();
        if (!s_go) { ok = false; break; }
      }
      VUS_WM(5);
      win_fetch(rsS, WL, band, n, PB * I, PB * p, fa);        // rows past the matrix's end read as zero
      win_fetch(rsS, WL, band, n, PB * I, PB * I, fb);
#pragma unroll
      for (int u = 0; u < 2; ++u) {       // NB * n_rhs <= 384 right-hand-side elements
        const int e = tid + 256 * u;
        const int q = e / NB, c = e - NB * q;
        okr[u] = e < NB * n_rhs && 6 * PB * I + c < 6 * n;
        vr[u] = ld_sc1(B.y + (okr[u] ? (size_t)q * ystride + 6 * (size_t)PB * I + c : 0));
      }
    }
    if (nb < NB) {                 // the matrix's last panel: identity past its end (sInv[0 .. nb) came from panel_factor)
      if (tid >= nb && tid < NB) sInv[tid] = 1.0;
      lds_barrier();
    }
    block_inverses<true>(sL, sM, sInv);
    VUS_WM(6);
    {
      // the six 16 x 16 blocks G(1,0), G(2,0), G(2,1), M_0, M_1, M_2, row-major: 768 pairs of doubles, 3 per thread
#pragma unroll
      for (int u = 0; u < WIN_PUB / 512; ++u) {
        const int e2 = tid + 256 * u;                  // pair index: block e2 >> 7, row (e2 >> 3) & 15, columns 2 (e2 & 7)
        const int blk = e2 >> 7, r = (e2 >> 3) & 15, c = 2 * (e2 & 7);
        const double* src = blk < 3 ? sL + (16 * (blk == 0 ? 1 : 2) + r) * LDD + 16 * (blk == 2 ? 1 : 0) + c
                                    : sM + 16 * MLD * (blk - 3) + MLD * r + c;
        st16_sc1(rsP, (unsigned)(8 * ((size_t)p * WIN_PUB) + 16 * e2), d2a_t{src[0], src[1]});
      }
    }
    VUS_WM(7);
    if (more) {                    // the next block row's tiles go to LDS while the stores above travel
      win_commit<false>(WL, fa, Xi);
      win_commit<false>(WL, fb, Xn);
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int e = tid + 256 * u;
        if (e < NB * n_rhs) s_rhs[e] = okr[u] ? vr[u] : 0.0;
      }
    }
    cb_drain();                    // the factor, the solved right-hand sides, the inverse blocks are in memory
    __syncthreads();
    if (tid == 0) __hip_atomic_store(F, p + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);    // panel p is out
    VUS_WM(8);
    if (!more) break;
    solve_rows<true>(1, Xi, Xi, sL, sM);         // X of block row p+1
    VUS_WM(9);
    win_store<true, true>(rsS, WL, band, n, PB * I, k0, Xi);      // the solved rows, on their way to memory while the update below runs
    {
      // next diagonal tile -= X X^T: its six lower MFMA tiles, at most two per wave; every element of Xn belongs to
      // one lane
      {
        // tile index t = 3 a + b: wave 0 -> (0,0), (2,1); wave 1 -> (1,0), (2,2); wave 2 -> (1,1); wave 3 -> (2,0).  The
        // two tiles of a wave are loaded together and their (independent) MFMA chains issued alternately.
        const int t0 = wave == 0 ? 0 : wave == 1 ? 3 : wave == 2 ? 4 : 6, t1 = wave == 0 ? 7 : 8;
        const bool two = wave < 2;
        const int a0 = t0 / UMT, b0 = t0 - UMT * a0, a1 = t1 / UMT, b1 = t1 - UMT * a1;
        double4_t acc0, acc1 = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          acc0[r] = -Xn[(16 * a0 + kq + 4 * r) * LDD + 16 * b0 + arow];
          if (two) acc1[r] = -Xn[(16 * a1 + kq + 4 * r) * LDD + 16 * b1 + arow];
        }
#pragma unroll
        for (int h = 0; h < 2; ++h) {      // two halves of the 48 columns: the operands of one half fit the registers left
          constexpr int HS = NB / 8;
          double av0[HS], bv0[HS], av1[HS], bv1[HS];
#pragma unroll
          for (int s2 = 0; s2 < HS; ++s2) {
            const int c = kq + 4 * (HS * h + s2);
            av0[s2] = Xi[(16 * a0 + arow) * ULD + c];
            bv0[s2] = Xi[(16 * b0 + arow) * ULD + c];
            av1[s2] = two ? Xi[(16 * a1 + arow) * ULD + c] : 0.0;
            bv1[s2] = two ? Xi[(16 * b1 + arow) * ULD + c] : 0.0;
          }
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int s2 = 0; s2 < HS; ++s2) {
            acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(av0[s2], bv0[s2], acc0, 0, 0, 0);
            if (two) acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(av1[s2], bv1[s2], acc1, 0, 0, 0);
          }
          __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          Xn[(16 * a0 + kq + 4 * r) * LDD + 16 * b0 + arow] = -acc0[r];
          if (two) Xn[(16 * a1 + kq + 4 * r) * LDD + 16 * b1 + arow] = -acc1[r];
        }
      }
      // its right-hand sides -= X z (the waves with one MFMA tile take them): eight lanes per element, six columns
      // each, summed across the lanes with DPP moves
      if (wave >= 2) {
        const int idx = tid - 128, part = idx & 7;
        for (int e0 = 0; e0 < NB * n_rhs; e0 += 16) {
          const int e = e0 + (idx >> 3);
          const bool on = e < NB * n_rhs;
          const int q = on ? e / NB : 0, r = on ? e - NB * q : 0;
          double xv[6], zv[6];
#pragma unroll
          for (int k = 0; k < 6; ++k) {
            xv[k] = Xi[r * ULD + 6 * part + k];
            zv[k] = s_z[q * NB + 6 * part + k];
          }
          __builtin_amdgcn_sched_barrier(0);
          double sum = 0.0;
#pragma unroll
          for (int k = 0; k < 6; ++k) sum += xv[k] * zv[k];
          sum += dpp_f64<0xB1>(sum);      // quad_perm [1, 0, 3, 2]
          sum += dpp_f64<0x4E>(sum);      // quad_perm [2, 3, 0, 1]
          sum += dpp_f64<0x141>(sum);     // row_half_mirror: the other quad of the eight
          if (on && part == 0) s_rhs[e] -= sum;
        }
      }
    }
    VUS_WM(10);
    cb_drain();
    __syncthreads();
    if (tid == 0) __hip_atomic_store(xflag + I, p + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    VUS_WM(11);
#ifdef VUS_TIMING
    if (vus_wm_on && tid < 12) g_wtm[tid] = s_wtm[tid];
#endif
    double* t_ = sL;
    sL = Xn;
    Xn = t_;
  }
  if (ok && NE < NT) {   // partial factorisation: the Schur complement's first diagonal tile and right-hand sides
    win_store<false>(rsS, WL, band, n, PB * NE, PB * NE, sL);
    for (int e = tid; e < NB * n_rhs; e += 256) {
      const int q = e / NB, c = e - NB * q;
      if (6 * PB * NE + c < 6 * n) st_sc1(B.y + (size_t)q * ystride + 6 * (size_t)PB * NE + c, s_rhs[e]);
    }
  }
}

// ---- one window slot, for every system of the set ----
__device__ void win_bulk(const WinSet& S, int slot, int band, int NE, int n_rhs, double* smem, int& s_go) {
  const int tid = threadIdx.x, lane = tid & 63, wave = wave_index();
  const int arow = lane & 15, kq = lane >> 4;
  const int n = S.s[0].n, NT = (n + PB - 1) / PB;
  const int D = (band + PB - 1) / PB, M = D + 1;
  double* Xi = smem;
  double* Xj = Xi + UT * ULD;
  double* sL = Xj + UT * ULD;
  double* sM = sL + UT * ULD;
  double* s_z = sM + 3 * 16 * MLD + NB;
  double* s_yown = s_z + 2 * BS_RHS_MAX * NB;          // [2 systems][q][NB]: right-hand sides of a hosted diagonal tile
  const size_t ystride = 6 * (size_t)n;
  int hi, lo;
  bandidx::win_slot_pair(slot, hi, lo);
  const WinLane WL = win_lane(band);
  double4_t acc[2][UQ];
#pragma unroll
  for (int y2 = 0; y2 < 2; ++y2)
#pragma unroll
    for (int q = 0; q < UQ; ++q) acc[y2][q] = double4_t{0.0, 0.0, 0.0, 0.0};
  bool live = true;
  for (int p = 0; p <= NE && live; ++p) {
#pragma unroll
    for (int sys = 0; sys < 2; ++sys) {
      if (sys >= S.count || !live) continue;
      const WinSys& B = S.s[sys];
      int I, J;
      bandidx::win_tile_of(hi, lo, M, p, I, J);
      if (I >= NT) continue;                       // no such tile in this matrix
      const int d = I - J;
      if (d <= 1 && p > I - 2) continue;           // the critical workgroup's by now (rows 0 and 1: from the start)
      const int birth = max(I - D, 0);
      int* F = B.F;
      int* abort_flag = F + 1;
      int* xflag = F + 2;
      int* hand = F + 2 + NT;
      double* yown = s_yown + sys * BS_RHS_MAX * NB;
      const rsrc_t rsS = win_rsrc(B.Sb, 8 * bandidx::band_doubles(n, band));
      const rsrc_t rsP = win_rsrc(B.pub, 8ll * WIN_PUB * NE);
      if (p == NE) {
        // after the last step of a partial factorisation: what is still in registers goes back to its place
        if (p > birth) {
          __syncthreads();
#pragma unroll
          for (int q = 0; q < UQ; ++q) {
            const int t = wave + 4 * q;
            const int a = t / UMT, b = t - UMT * a;
            if (t >= UMT * UMT) continue;
#pragma unroll
            for (int r = 0; r < 4; ++r) Xi[(16 * a + kq + 4 * r) * LDD + 16 * b + arow] = -acc[sys][q][r];
          }
          __syncthreads();
          win_store<false>(rsS, WL, band, n, PB * I, PB * J, Xi);
          if (d == 0)
            for (int e = tid; e < NB * n_rhs; e += 256) {
              const int q = e / NB, c = e - NB * q;
              if (6 * PB * I + c < 6 * n) st_sc1(B.y + (size_t)q * ystride + 6 * (size_t)PB * I + c, yown[e]);
            }
        }
        continue;
      }
      const int k0 = PB * p;
      const int pb = min(PB, n - k0);
      if (p == birth) {
        // birth: the tile's entries of the assembled system (nobody has written them in this launch)
#pragma unroll
        for (int q = 0; q < UQ; ++q) {
          const int t = wave + 4 * q;
          const int a = t / UMT, b = t - UMT * a;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const long long o = t < UMT * UMT ? bandidx::win_scalar(band, n, PB * I, PB * J, 16 * a + kq + 4 * r, 16 * b + arow) : -1;
            const double v = B.Sb[o >= 0 ? o : 0];       // unconditional load (masked: element 0), all twelve in flight
            acc[sys][q][r] = o >= 0 ? -v : 0.0;
          }
        }
        if (d == 0)
          for (int e = tid; e < NB * n_rhs; e += 256) {
            const int q = e / NB, c = e - NB * q;
            yown[e] = 6 * PB * I + c < 6 * n ? B.y[(size_t)q * ystride + 6 * (size_t)PB * I + c] : 0.0;
          }
      }
      if (J == p) {
        // ---- elimination of the tile: X = A L_D^-T with the inverse blocks the critical workgroup published ----
        if (tid == 0) s_go = cb_wait(F, p + 1, abort_flag, S.spin);
        __syncthreads();
        if (!s_go) { live = false; continue; }
#pragma unroll
        for (int q = 0; q < UQ; ++q) {
          const int t = wave + 4 * q;
          const int a = t / UMT, b = t - UMT * a;
          if (t >= UMT * UMT) continue;
#pragma unroll
          for (int r = 0; r < 4; ++r) Xi[(16 * a + kq + 4 * r) * ULD + 16 * b + arow] = -acc[sys][q][r];
        }
        {
          d2a_t v[WIN_PUB / 512];
#pragma unroll
          for (int u = 0; u < WIN_PUB / 512; ++u) v[u] = ld16_sc1(rsP, (unsigned)(8 * ((size_t)p * WIN_PUB) + 16 * (tid + 256 * u)));
#pragma unroll
          for (int u = 0; u < WIN_PUB / 512; ++u) {
            const int e2 = tid + 256 * u;
            const int blk = e2 >> 7, r = (e2 >> 3) & 15, c = 2 * (e2 & 7);
            double* dst = blk < 3 ? sL + (16 * (blk == 0 ? 1 : 2) + r) * LDD + 16 * (blk == 2 ? 1 : 0) + c
                                  : sM + 16 * MLD * (blk - 3) + MLD * r + c;
            dst[0] = v[u].x;
            dst[1] = v[u].y;
          }
        }
        __syncthreads();
        solve_rows(1, Xi, Xi, sL, sM);
        win_store<true>(rsS, WL, band, n, PB * I, k0, Xi);
        cb_drain();
        __syncthreads();
        if (tid == 0) __hip_atomic_store(xflag + I, p + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        continue;
      }
      // ---- update with the solved rows of block rows I and J of panel p ----
      if (tid == 0) s_go = cb_wait(xflag + I, p + 1, abort_flag, S.spin) && (d == 0 || cb_wait(xflag + J, p + 1, abort_flag, S.spin));
      __syncthreads();
      if (!s_go) { live = false; continue; }
      {
        WinFetch fi, fj;
        win_fetch(rsS, WL, band, n, PB * I, k0, fi);
        if (d > 0) win_fetch(rsS, WL, band, n, PB * J, k0, fj);
        win_commit<true>(WL, fi, Xi);
        if (d > 0) win_commit<true>(WL, fj, Xj);
      }
      if (d == 0)
        for (int e = tid; e < NB * n_rhs; e += 256) {
          const int q = e / NB, c = e - NB * q;
          s_z[e] = c < 6 * pb ? ld_sc1(B.y + (size_t)q * ystride + 6 * (size_t)k0 + c) : 0.0;
        }
      __syncthreads();
      win_mfma_update(acc[sys], Xi, d == 0 ? Xi : Xj, d == 0);
      if (d == 0)
        for (int e = tid; e < NB * n_rhs; e += 256) {
          const int q = e / NB, r = e - NB * q;
          double sum = 0.0;
#pragma unroll 8
          for (int c = 0; c < NB; ++c) sum += Xi[r * ULD + c] * s_z[q * NB + c];
          yown[e] -= sum;
        }
      if (d <= 1 && p == I - 2) {
        // hand the tile (and a diagonal tile's right-hand sides) over to the critical workgroup
        __syncthreads();             // every wave has read its X fragments: Xi is free
#pragma unroll
        for (int q = 0; q < UQ; ++q) {
          const int t = wave + 4 * q;
          const int a = t / UMT, b = t - UMT * a;
          if (t >= UMT * UMT) continue;
#pragma unroll
          for (int r = 0; r < 4; ++r) Xi[(16 * a + kq + 4 * r) * LDD + 16 * b + arow] = -acc[sys][q][r];
        }
        __syncthreads();
        win_store<false>(rsS, WL, band, n, PB * I, PB * J, Xi);
        __syncthreads();             // yown is complete
        if (d == 0)
          for (int e = tid; e < NB * n_rhs; e += 256) {
            const int q = e / NB, c = e - NB * q;
            if (6 * PB * I + c < 6 * n) st_sc1(B.y + (size_t)q * ystride + 6 * (size_t)PB * I + c, yown[e]);
          }
        cb_drain();
        __syncthreads();
        if (tid == 0) __hip_atomic_fetch_add(hand + I, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      __syncthreads();               // Xi, Xj, s_z are free again
    }
  }
}

// Block map: blocks 0 .. count-1 are the critical workgroups; blocks pad0 .. pad0+count-1 exit at once; the rest are the
// window slots in order.  pad0 = the number of CUs: with two workgroups per CU and blocks dealt to CUs in order, block
// b + n_cu is the one that would share a CU with block b, so the padding leaves each critical workgroup a compute unit
// (its LDS bandwidth, its issue slots) to itself.  Placement is the hardware's business: if it differs, only speed does.
__global__ __launch_bounds__(256, 2) void chol_window_kernel(WinSet S, int band, int NE, int n_rhs, int pad0) {
  extern __shared__ __attribute__((aligned(16))) double win_smem[];
  __shared__ int s_bad, s_go;
  const int bid = blockIdx.x;
  if (bid >= pad0 && bid < pad0 + S.count) return;
  if (S.fault && bid == S.count) return;
  if (bid < S.count) win_critical(S.s[bid], band, NE, n_rhs, S.spin, win_smem, s_bad, s_go);
  else win_bulk(S, bid - S.count - (bid >= pad0 ? S.count : 0), band, NE, n_rhs, win_smem, s_go);
  if (threadIdx.x == 0)
    for (int q = 0; q < S.count; ++q)       // an expired wait of THIS kernel: VUS_STATUS_WINDOW_EXPIRED, the caller may redo the solve launch by launch
      if (__hip_atomic_load(S.s[q].F + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) S.s[q].status[0] = -3;
}

// slots (i, s) with s > i of the first `band` rows lie left of pose 0: never read, kept at zero (one workgroup per row)
__global__ void band_head_zero_kernel(double* __restrict__ Sband, int n_poses, int band) {
  const int i = blockIdx.x;
  double* row = Sband + 36 * ((size_t)i * (band + 1) + (i + 1));
  for (int t = threadIdx.x; t < 36 * (band - i); t += blockDim.x) row[t] = 0.0;
}

__global__ void add_diag_kernel(double* __restrict__ Sband, int n_poses, int band, double value) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t < 6 * n_poses) Sband[36 * (size_t)(t / 6) * (band + 1) + 7 * (t % 6)] += value;
}

__global__ void negate_copy_kernel(const double* __restrict__ src, double* __restrict__ dst, int n) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t < n) dst[t] = -src[t];
}

// ---------------------------------------------------------------------------------------------
// back-substitution, retraction, error evaluation
__global__ __launch_bounds__(256) void backsub_kernel(vus_ba_problem P, const double* __restrict__ W,
                                                      const double* __restrict__ Vinv,
                                                      const double* __restrict__ gl, const double* __restrict__ dp,
                                                      double* __restrict__ dl) {
  const int lane = threadIdx.x & 63;
  const int j = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (j >= P.n_points) return;
  double t[3] = {0, 0, 0};
  for (int a = P.point_ptr[j] + lane; a < P.point_ptr[j + 1]; a += 64) {
    const double* Wa = W + 18 * (size_t)a;
    const double* d = dp + 6 * (size_t)pose_stride(P) * P.obs_pose[a];
#pragma unroll
    for (int rr = 0; rr < 6; ++rr) {
      const double dr = d[rr];
      t[0] += Wa[3 * rr] * dr;
      t[1] += Wa[3 * rr + 1] * dr;
      t[2] += Wa[3 * rr + 2] * dr;
    }
  }
#pragma unroll
  for (int c = 0; c < 3; ++c) t[c] = wave_sum(t[c]);
  if (lane == 0) {
#pragma unroll
    for (int c = 0; c < 3; ++c) t[c] += gl[3 * (size_t)j + c];
    const double* vi = Vinv + 6 * (size_t)j;
#pragma unroll
    for (int c = 0; c < 3; ++c)
      dl[3 * (size_t)j + c] = -(sym3(vi, c, 0) * t[0] + sym3(vi, c, 1) * t[1] + sym3(vi, c, 2) * t[2]);
  }
}

__global__ void retract_kernel(int n_poses, int n_points, int ps, const double* __restrict__ poses,
                               const double* __restrict__ points, const double* __restrict__ dp,
                               const double* __restrict__ dl, double* __restrict__ new_poses,
                               double* __restrict__ new_points) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t < n_poses) {
    double T[12], xi[6], out[12];
    load12(poses + 12 * (size_t)t, T);
#pragma unroll
    for (int k = 0; k < 6; ++k) xi[k] = dp[6 * (size_t)ps * t + k];
    pose_retract(T, xi, out);
#pragma unroll
    for (int k = 0; k < 12; ++k) new_poses[12 * (size_t)t + k] = out[k];
  }
  for (int k = t; k < 3 * n_points; k += gridDim.x * blockDim.x) new_points[k] = points[k] + dl[k];
}

// per point: part_lin[j] = 0.5 sum |r + H1 dp + H2 dl|^2 at the old values,
//            part_new[j] = 0.5 sum |r|^2 at the new values
template <bool WITH_LIN>
__global__ __launch_bounds__(256) void eval_points_kernel(vus_ba_problem P, const double* __restrict__ poses,
                                                          const double* __restrict__ points,
                                                          const double* __restrict__ dp, const double* __restrict__ dl,
                                                          const double* __restrict__ new_poses,
                                                          const double* __restrict__ new_points,
                                                          double* __restrict__ part_lin, double* __restrict__ part_new) {
  const int lane = threadIdx.x & 63;
  const int j = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (j >= P.n_points) return;
  const Calib K = load_calib(P.K, P.inv_sigma);
  const double pn[3] = {new_points[3 * (size_t)j], new_points[3 * (size_t)j + 1], new_points[3 * (size_t)j + 2]};
  double po[3] = {0, 0, 0}, d_l[3] = {0, 0, 0};
  if (WITH_LIN) {
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      po[c] = points[3 * (size_t)j + c];
      d_l[c] = dl[3 * (size_t)j + c];
    }
  }
  double e_lin = 0, e_new = 0;
  for (int a = P.point_ptr[j] + lane; a < P.point_ptr[j + 1]; a += 64) {
    const int i = P.obs_pose[a];
    const double m[3] = {P.meas[3 * (size_t)a], P.meas[3 * (size_t)a + 1], P.meas[3 * (size_t)a + 2]};
    double T[12], r[3];
    load12(new_poses + 12 * (size_t)i, T);
    stereo_factor<false, false>(T, pn, m, K, r, nullptr, nullptr);
    e_new += 0.5 * (r[0] * r[0] + r[1] * r[1] + r[2] * r[2]);
    if (WITH_LIN) {
      double H1[18], H2[9];
      load12(poses + 12 * (size_t)i, T);
      stereo_factor<true, true>(T, po, m, K, r, H1, H2);
      const double* d = dp + 6 * (size_t)pose_stride(P) * i;
#pragma unroll
      for (int rr = 0; rr < 3; ++rr) {
        double t = r[rr];
#pragma unroll
        for (int c = 0; c < 6; ++c) t += H1[6 * rr + c] * d[c];
#pragma unroll
        for (int c = 0; c < 3; ++c) t += H2[3 * rr + c] * d_l[c];
        e_lin += 0.5 * t * t;
      }
    }
  }
  e_new = wave_sum(e_new);
  if (WITH_LIN) e_lin = wave_sum(e_lin);
  if (lane == 0) {
    part_new[j] = e_new;
    if (WITH_LIN) part_lin[j] = e_lin;
  }
}

int check_problem(const vus_ba_problem* P) {
  VUS_REQUIRE(P != nullptr, "problem is null");
  VUS_REQUIRE(P->n_poses >= 1 && P->n_points >= 0 && P->n_obs >= 0 && P->n_priors >= 0,
              "bad sizes: poses=%d points=%d obs=%d priors=%d", P->n_poses, P->n_points, P->n_obs, P->n_priors);
  VUS_REQUIRE(P->K != nullptr, "K is null");
  VUS_REQUIRE(P->inv_sigma > 0.0, "inv_sigma=%g", P->inv_sigma);
  // pose_ptr has n_poses + 1 entries and the per-pose kernels read it even for a graph without a stereo factor
  VUS_REQUIRE(P->pose_ptr != nullptr, "pose_ptr is null");
  if (P->n_points > 0) VUS_REQUIRE(P->point_ptr != nullptr, "point_ptr is null");
  if (P->n_obs > 0)
    VUS_REQUIRE(P->meas && P->obs_pose && P->obs_point && P->point_ptr && P->obs_ppos && P->pobs_lidx,
                "observation arrays are null");
  if (P->n_priors > 0) VUS_REQUIRE(P->prior_pose && P->prior_T && P->prior_w, "prior arrays are null");
  return VUS_OK;
}

inline int cdiv(long long a, int b) { return (int)((a + b - 1) / b); }

}  // namespace

extern "C" long long vus_ba_work_doubles(const vus_ba_problem* P) {
  if (!P) return 0;
  return 2ll * ((long long)P->n_points + 1) + 8;
}

extern "C" int vus_ba_error(const vus_ba_problem* P, const double* poses, const double* points, double* err,
                            double* work, void* stream) {
  if (int rc = check_problem(P)) return rc;
  VUS_REQUIRE(poses && (points || !P->n_points) && err && work, "null buffer");
  hipStream_t st = vus::as_stream(stream);
  const int nL = P->n_points;
  if (nL > 0)
    eval_points_kernel<false><<<cdiv(nL, 4), 256, 0, st>>>(*P, nullptr, nullptr, nullptr, nullptr, poses, points,
                                                          nullptr, work);
  priors_kernel<<<1, 64, 0, st>>>(*P, poses, nullptr, nullptr, nullptr, work + nL, 1);
  reduce_partials_kernel<<<1, 1024, 0, st>>>(work, nL + 1, err);
  VUS_CHECK_LAUNCH("ba_error");
  return VUS_OK;
}

extern "C" int vus_ba_linearize(const vus_ba_problem* P, const double* poses, const double* points, double* W,
                                double* V, double* gl, double* Hpp, double* gp, double* err, double* work,
                                void* stream) {
  if (int rc = check_problem(P)) return rc;
  // a graph without landmarks (priors only) has empty per-landmark / per-observation arrays: those may be null
  VUS_REQUIRE(poses && Hpp && gp && err && work, "null buffer");
  VUS_REQUIRE((points && V && gl) || !P->n_points, "null landmark buffer");
  VUS_REQUIRE(W || !P->n_obs, "null observation buffer");
  hipStream_t st = vus::as_stream(stream);
  const int nL = P->n_points;
  if (nL > 0) lin_points_kernel<<<cdiv(nL, 4), 256, 0, st>>>(*P, poses, points, W, V, gl, work);
  lin_poses_kernel<<<P->n_poses, 256, 0, st>>>(*P, poses, points, Hpp, gp);
  priors_kernel<<<1, 64, 0, st>>>(*P, poses, nullptr, Hpp, gp, work + nL, 0);
  reduce_partials_kernel<<<1, 1024, 0, st>>>(work, nL + 1, err);
  VUS_CHECK_LAUNCH("ba_linearize");
  return VUS_OK;
}

extern "C" int vus_ba_schur(const vus_ba_problem* P, const vus_ba_tiles* T, double lambda, const double* W, const double* V,
                            const double* gl, const double* Hpp, const double* gp, double* Vinv, double* Y, double* Sband,
                            int band_nodes, double* gs, int* counter, void* stream) {
  if (int rc = check_problem(P)) return rc;
  VUS_REQUIRE(T != nullptr, "tile structure is null");
  const int nP = P->n_poses, nL = P->n_points, nO = P->n_obs;
  const int ps = pose_stride(*P);
  VUS_REQUIRE(Hpp && gp && Sband && gs && counter, "null buffer");
  VUS_REQUIRE((V && gl && Vinv) || !nL, "null landmark buffer");
  VUS_REQUIRE(W || !nO, "null observation buffer");
  VUS_REQUIRE(lambda >= 0.0, "lambda=%g", lambda);
  VUS_REQUIRE(T->band >= 0 && T->n_tiles == (nP + 7) / 8 && T->n_units == T->n_tiles * ((T->band + 7) / 8 + 1) && T->n_entries >= 0,
              "tile structure of another problem: band=%d tiles=%d units=%d", T->band, T->n_tiles, T->n_units);
  VUS_REQUIRE(band_nodes >= ps * T->band && band_nodes < ps * nP + 1, "band_nodes=%d against %d poses of tile band, %d nodes",
              band_nodes, T->band, ps * nP);
  VUS_REQUIRE(T->unit_ptr && T->order && (T->entries || T->n_entries == 0), "tile lists are null");
  hipStream_t st = vus::as_stream(stream);
  // With velocity nodes between the poses (ps = 2) the blocks the tile pairs do not cover belong to the inertial
  // factors and start from zero.  With ps = 1 every stored block (i, k), 0 <= k <= i, i - k <= band, is WRITTEN by its
  // tile pair (nothing is accumulated into); the slots left of pose 0 (k < 0) of the first band rows are never read by
  // the solvers (csrc/band_index.h masks them) and only kept finite.
  if (ps > 1) {
    VUS_CHECK_HIP(hipMemsetAsync(Sband, 0, sizeof(double) * 36 * (size_t)nP * ps * (band_nodes + 1), st));
    VUS_CHECK_HIP(hipMemsetAsync(gs, 0, sizeof(double) * 6 * (size_t)nP * ps, st));
  } else if (band_nodes > 0) {
    band_head_zero_kernel<<<min(band_nodes, nP), 256, 0, st>>>(Sband, nP, band_nodes);
  }
  VUS_CHECK_HIP(hipMemsetAsync(counter, 0, sizeof(int), st));
  if (nL > 0) vinv_kernel<<<cdiv(nL, 256), 256, 0, st>>>(nL, lambda, V, Vinv);
  if (nO > 0 && Y != nullptr) ymul_kernel<<<cdiv(nO, 256), 256, 0, st>>>(*P, W, Vinv, Y);   // optional output only
  constexpr int lds = ST_WAVES * ST_WAVE_DOUBLES * (int)sizeof(double);
  // (kernel, device) attribute: set per call -- a host-side table write -- instead of once per process
  VUS_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(schur_tiles_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
  int n_cu = device_cu_count();
  if (n_cu < 8) n_cu = 256;
  int wg = 2 * n_cu;                 // persistent: two workgroups per CU is what their LDS admits
  if (wg > T->n_units) wg = T->n_units;
  schur_tiles_kernel<<<wg, 64 * ST_WAVES, lds, st>>>(*T, nP, ps, band_nodes, lambda, W, Vinv, Hpp, gl, gp, Sband, gs, counter);
  VUS_CHECK_LAUNCH("ba_schur");
  return VUS_OK;
}

extern "C" int vus_ba_add_diag(double* Sband, int n_poses, int band, double value, void* stream) {
  VUS_REQUIRE(Sband != nullptr, "Sband is null");
  VUS_REQUIRE(n_poses >= 1 && band >= 0, "n_poses=%d band=%d", n_poses, band);
  add_diag_kernel<<<cdiv(6ll * n_poses, 256), 256, 0, vus::as_stream(stream)>>>(Sband, n_poses, band, value);
  VUS_CHECK_LAUNCH("ba_add_diag");
  return VUS_OK;
}

namespace {
// band == 0 has no spare slot in Sband: the solver runs alone and only touches F[0], F[1]; one 16-int buffer per
// DEVICE serves (allocated once, under a lock; concurrent band-0 solves may share it: nobody waits on F[0] when
// there are no helpers, and F[1] is only ever raised by an expired wait, which a band-0 solve does not have).
int* flags_fallback() {
  static std::mutex mu;
  static int* buf[64] = {};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return nullptr;
  std::lock_guard<std::mutex> lock(mu);
  if (!buf[dev] && hipMalloc(&buf[dev], 16 * sizeof(int)) != hipSuccess) buf[dev] = nullptr;
  return buf[dev];
}

// Number of workgroups of the cooperative back-substitution that can be resident at once: the occupancy query for
// this kernel times the compute units of the device, less a margin of one workgroup per eight CUs (the query can
// read one block per CU high, MI355X_MICROARCH.md "Residency and cooperative launch").  The waits are bounded, so a
// workgroup that is not resident after all (another stream holding CUs) ends in status -1, not in a hang.
// VUS_CB_MAX_WG (compile time) and the tuning knob VUS_TUNE_CB_MAX_WG (run time, used by the tests to force several
// row groups per workgroup) cap it further.
int backsolve_max_wg() {
  static std::mutex mu;
  static int cached[64] = {};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 1;
  int cap;
  {
    std::lock_guard<std::mutex> lock(mu);
    if (!cached[dev]) {
      int per_cu = 0, n_cu = 0;
      if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, chol_backsolve_kernel, CB_THREADS, 0) != hipSuccess) per_cu = 1;
      if (hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) n_cu = 8;
      int c = per_cu * n_cu - n_cu / 8;      // all of them must be resident at once: leave a margin
      cached[dev] = c < 1 ? 1 : c;
    }
    cap = cached[dev];
  }
  if (cap > CB_MAX_WG) cap = CB_MAX_WG;
  const int v = g_knobs.cb_max_wg.load(std::memory_order_relaxed);
  if (v >= 1 && v < cap) cap = v;
  return cap;
}

// Scratch of the persistent window kernel for one system of n poses of which n_panels panels are eliminated:
// published inverse blocks (WIN_PUB doubles per panel) + flags, in doubles.
size_t window_doubles(int n, int n_panels) {
  const size_t NT = (size_t)(n + PB - 1) / PB;
  return ((size_t)n_panels * WIN_PUB + (2 + 2 * NT + 2) / 2 + 3) & ~(size_t)1;     // even: what follows stays 16-byte aligned
}

// Workgroups of chol_window_kernel that are resident together (occupancy query x CUs, less a margin: the flags
// protocol needs ALL of them running; a workgroup that is not resident after all ends in status -1, not in a hang).
int window_capacity() {
  static std::mutex mu;
  static int cached[64] = {};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 0;
  std::lock_guard<std::mutex> lock(mu);
  if (!cached[dev]) {
    const int lds = WIN_LDS_DOUBLES * (int)sizeof(double);
    int per_cu = 0;
    const int n_cu = device_cu_count();
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(chol_window_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess ||
        hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, chol_window_kernel, 256, lds) != hipSuccess)
      per_cu = 0;
    if (per_cu > 2) per_cu = 2;       // what __launch_bounds__(256, 2) and 76 KB of LDS were sized for
    const int c = per_cu * n_cu - n_cu / 8;
    cached[dev] = c > 0 ? c : -1;
  }
  return cached[dev] > 0 ? cached[dev] : 0;
}

// true if the persistent window kernel can run these systems (geometry, scratch, residency)
bool window_applicable(const BandSet& S, int band, int n_elim) {
  const int n = S.s[0].n;
  if (band < WIN_MIN_BAND || n < 3 * PB || n_elim < PB) return false;
  if (8 * bandidx::band_doubles(n, band) >= (1ll << 32)) return false;      // the kernel addresses the band through 32-bit buffer offsets
  if (n_elim < n && n_elim % PB != 0) return false;
  for (int q = 0; q < S.count; ++q)
    if (!S.s[q].win_pub || !S.s[q].win_F || S.s[q].n != n) return false;
  const int D = (band + PB - 1) / PB, M = D + 1;
  return M * (M + 1) / 2 + 2 * S.count <= window_capacity();
}

int window_launch(const BandSet& S, int band, int n_elim, int n_rhs, hipStream_t st, bool flags_cleared = false) {
  const int n = S.s[0].n, NT = (n + PB - 1) / PB;
  const int NE = n_elim >= n ? NT : n_elim / PB;
  const int D = (band + PB - 1) / PB, M = D + 1;
  WinSet W;
  W.count = S.count;
  // a poll is >= ~1 us (an sc1 load through the L2 and an s_sleep); a healthy chain advances one panel step in 12-30 us,
  // and the longest legitimate wait is a late tile's for the whole chain: 512 polls per step is >= 10x that.  A
  // workgroup that is not resident (something else holds its CU) therefore fails in tens of milliseconds, not seconds.
  W.spin = 16384 + 512 * NE;
  W.fault = g_knobs.win_fault.load(std::memory_order_relaxed);
  for (int q = 0; q < 2; ++q) {
    const BandSys& b = S.s[q < S.count ? q : 0];
    W.s[q] = WinSys{b.Sb, b.y, b.status, b.win_pub, b.win_F, b.n};
  }
  if (!flags_cleared)
    for (int q = 0; q < S.count; ++q)
      VUS_CHECK_HIP(hipMemsetAsync(S.s[q].win_F, 0, sizeof(int) * (size_t)(2 + 2 * NT), st));
  const int lds = WIN_LDS_DOUBLES * (int)sizeof(double);
  VUS_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(chol_window_kernel),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, lds));
  const int n_slots = M * (M + 1) / 2, n_cu = device_cu_count();
  const bool pad = n_cu > 0 && S.count + n_slots > n_cu && g_knobs.win_pad.load(std::memory_order_relaxed) != 0;
  chol_window_kernel<<<S.count + n_slots + (pad ? S.count : 0), 256, lds, st>>>(W, band, NE, n_rhs, pad ? n_cu : 0x7FFFFFFF);
  VUS_CHECK_LAUNCH("ba_band_window");
  return VUS_OK;
}

// Right-looking factorisation launches for the systems of S (identical geometry): panels 0 .. n_elim/PB - 1 are
// eliminated (n_elim == S.s[0].n: the whole matrix; smaller, a multiple of PB: a PARTIAL factorisation that leaves the
// Schur complement of the eliminated poses in the trailing window and the forward-substituted right-hand sides
// in y).  The forward substitution rides along.
int factor_launches(const BandSet& S, int band, int n_elim, int n_rhs, hipStream_t st, hipStream_t st2 = nullptr,
                    bool flags_cleared = false) {
  const int n = S.s[0].n;
  const bool full = n_elim >= n;
  // Three ways to issue a panel step (the knob VUS_TUNE_BAND_MODE = 0 / 1 / 2 forces one; tests and A/B timing):
  //  0  fused launch: every update tile solves its rows itself -- one system;
  //  1  TRSM launch + SYRK launch shared by both systems (rows solved once, light update tiles, one round);
  //  2  the two systems on two STREAMS, a (TRSM, SYRK) launch pair each per panel, issued alternately: the halves of
  //     the two-sided solve are independent chains, so one half's small TRSM launch runs beside the other's update
  //     (band solve at configs[2]: mode 0 4.64 ms, mode 1 4.31 ms, mode 2 4.11 ms; the fused launch per half on two
  //     streams was measured too: 4.15 ms).
  int mode = S.count == 2 ? (st2 ? 2 : 1) : 0;
  //  3  the persistent window kernel: ONE launch for the whole chain of panel steps of every system of S
  //     (chol_window_kernel); the automatic choice wherever it applies (window_applicable).
  {
    const int m = g_knobs.band_mode.load(std::memory_order_relaxed);
    if ((m < 0 || m == 3) && window_applicable(S, band, full ? n : n_elim)) {
      g_knobs.last_mode = 3;
      return window_launch(S, band, full ? n : n_elim, n_rhs, st, flags_cleared);
    }
    if (m == 0 || m == 1 || (m == 2 && S.count == 2 && st2)) mode = m;
  }
  g_knobs.last_mode = mode;
  BandSet one[2];
  hipStream_t sts[2] = {st, st2};
  int n_sets = 1;
  const bool fused = mode == 0;
  if (mode == 2) {
    n_sets = 2;
    for (int q = 0; q < 2; ++q) { one[q].count = 1; one[q].s[0] = S.s[q]; one[q].s[1] = S.s[q]; }
  } else {
    one[0] = S;
  }
  int k0_prev = -1, tiles_prev = 0;
  for (int k0 = 0; k0 < (full ? n : n_elim); k0 += PB) {
    const int pb = n - k0 < PB ? n - k0 : PB;
    const int i_first = k0 + pb;
    int i_last = k0 + pb - 1 + band;
    if (i_last > n - 1) i_last = n - 1;
    const int rows = i_last - i_first + 1;
    const int tiles = rows > 0 ? (rows + UTP - 1) / UTP : 0;
    const int n_update = tiles * (tiles + 1) / 2;
    const int factor_next = (full || k0 + PB < n_elim) ? 1 : 0;
    for (int q = 0; q < n_sets; ++q) {
      const BandSet& B = one[q];
      const int c = B.count;
      hipStream_t s_ = sts[q];
      // panel 0 has a launch of its own; panel p + 1 is factored by tile (0,0) of panel p's update launch
      if (k0 == 0) chol_panel_kernel<<<c, 256, 0, s_>>>(B, band, k0, n_rhs);
      if (!fused) {
        if (tiles > 0) {
          chol_trsm_kernel<<<c * tiles, 256, 0, s_>>>(B, band, k0, n_rhs);
          chol_syrk_kernel<<<c * n_update, 256, 0, s_>>>(B, band, k0, n_rhs, factor_next);
        }
      } else if (n_update + tiles_prev > 0) {
        // update tiles of this panel + the write-back of the previous panel's solved rows
        chol_trsm_update_kernel<<<c * (n_update + tiles_prev), 256, 0, s_>>>(B, band, k0, n_update, k0_prev, n_rhs, factor_next);
      }
      // no row below this panel (band 0, or a band that ends here): nobody has factored the next panel
      if (tiles == 0 && factor_next && i_first < n) chol_panel_kernel<<<c, 256, 0, s_>>>(B, band, i_first, n_rhs);
    }
    k0_prev = k0;
    tiles_prev = tiles;
  }
  if (fused && !full && tiles_prev > 0)   // write-back of the last eliminated panel's solved rows
    for (int q = 0; q < n_sets; ++q)
      chol_trsm_update_kernel<<<one[q].count * tiles_prev, 256, 0, sts[q]>>>(one[q], band, n_elim, 0, k0_prev, n_rhs, 0);
  VUS_CHECK_LAUNCH("ba_band_factor");
  return VUS_OK;
}

// The two-sided solve forks onto an auxiliary stream.  That stream and its fork / join events belong to ONE caller
// stream of ONE device (created on first use, never destroyed): two host threads that solve on their own streams share
// nothing, so neither can re-record an event the other is about to wait on.  The entry's mutex is held from the first
// fork to the last join of a call: two threads that do share a caller stream serialise their (host-side) enqueue
// sections instead of interleaving them.
struct SplitAux {
  int dev = -1;
  hipStream_t key = nullptr;
  hipStream_t s2 = nullptr;
  hipEvent_t fork = nullptr, join = nullptr;
  std::mutex mu;
};
SplitAux* split_aux(hipStream_t st) {
  static std::mutex mu;
  static std::vector<std::unique_ptr<SplitAux>> pool;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return nullptr;
  std::lock_guard<std::mutex> lock(mu);
  for (auto& a : pool)
    if (a->dev == dev && a->key == st) return a.get();
  std::unique_ptr<SplitAux> a(new SplitAux);
  a->dev = dev;
  a->key = st;
  if (hipStreamCreateWithFlags(&a->s2, hipStreamNonBlocking) != hipSuccess) return nullptr;
  if (hipEventCreateWithFlags(&a->fork, hipEventDisableTiming) != hipSuccess ||
      hipEventCreateWithFlags(&a->join, hipEventDisableTiming) != hipSuccess) {
    (void)hipStreamDestroy(a->s2);
    return nullptr;
  }
  pool.push_back(std::move(a));
  return pool.back().get();
}

// One fork .. join section on a SplitAux.  Once fork() has succeeded, the caller's stream is made to wait for the
// auxiliary stream before the section ends -- by join(), or by the destructor on every early return -- so that no
// buffer of the caller is still in use on the auxiliary stream when the call returns.
struct SplitSection {
  SplitAux* a;
  hipStream_t st;
  std::unique_lock<std::mutex> lock;
  bool open = false;
  SplitSection(SplitAux* aux, hipStream_t s) : a(aux), st(s) {
    if (a) lock = std::unique_lock<std::mutex>(a->mu);
  }
  int fork() {      // s2 continues from this point of st
    VUS_CHECK_HIP(hipEventRecord(a->fork, st));
    VUS_CHECK_HIP(hipStreamWaitEvent(a->s2, a->fork, 0));
    open = true;
    return VUS_OK;
  }
  int join() {      // st continues only after everything issued on s2 so far
    open = false;
    VUS_CHECK_HIP(hipEventRecord(a->join, a->s2));
    VUS_CHECK_HIP(hipStreamWaitEvent(st, a->join, 0));
    return VUS_OK;
  }
  ~SplitSection() {
    if (open) (void)join();
  }
};

// Cooperative back-substitution x = L^-T y of every system of S (n_solve > 0: of its leading n_solve poses only).
// The diagonal panels of every system of S inverted in place (see diag_invert_kernel); false = band too narrow
bool diag_invert_launch(const BandSet& S, int band, int n_solve, hipStream_t st) {
  // The inverse of a diagonal panel is a full lower triangle: it fits the panel's own storage only if the band
  // reaches across the panel (band >= 7 poses); narrower bands keep the factor and substitute.
  if (band < PB - 1) return false;
  int np_max = 0;
  for (int q = 0; q < S.count; ++q) {
    const int n_q = n_solve > 0 ? n_solve : S.s[q].n;
    np_max = (n_q + PB - 1) / PB > np_max ? (n_q + PB - 1) / PB : np_max;
  }
  diag_invert_kernel<<<dim3(np_max, S.count), 64, 0, st>>>(S, band, n_solve);
  return true;
}

int backsolve_launch(BandSet S, int band, int n_rhs, int n_solve, hipStream_t st, bool pre_inverted = false,
                     bool flags_cleared = false) {
  // flags of the cooperative sweep live in the unused slots of block row 0 (blocks (0, k < 0))
  const int n_groups = band > 0 ? (band + PB - 1) / PB : 1;
  VUS_REQUIRE(band == 0 || 2 + n_groups <= 72 * band, "band=%d: too many row groups for the flag area", band);
  for (int q = 0; q < S.count; ++q) {
    S.s[q].F = band > 0 ? reinterpret_cast<int*>(S.s[q].Sb + 36) : flags_fallback();
    VUS_REQUIRE(S.s[q].F != nullptr, "no scratch for the solver flags");
    VUS_REQUIRE(band > 0 || S.count == 1, "two band-0 systems cannot share the fallback flags");
    if (!flags_cleared) VUS_CHECK_HIP(hipMemsetAsync(S.s[q].F, 0, sizeof(int) * (size_t)(2 + n_groups), st));
  }
  const int inverted = pre_inverted ? 1 : (int)diag_invert_launch(S, band, n_solve, st);
  // at most backsolve_max_wg() cooperating workgroups in total, so that all of them are resident at once
  int max_wg = backsolve_max_wg() / S.count;
  if (max_wg < 2) max_wg = 2;       // a solver and at least one helper per system (the helper then serves every row group)
  const int n_wg = n_groups < max_wg ? n_groups : max_wg;
  chol_backsolve_kernel<<<n_wg * S.count, CB_THREADS, 0, st>>>(S, band, n_rhs, n_groups, n_solve, inverted);
  VUS_CHECK_LAUNCH("ba_band_backsolve");
  return VUS_OK;
}

// flags_cleared: the caller has zeroed the status word, the window kernel's flags and the back-substitution's flag words
// (the unused slots of block row 0, Sband + 36) in a launch of its own.
int band_solve_impl(double* Sband, int n_nodes, int band, double* y, int n_rhs, int* status, hipStream_t st,
                    double* win_scratch = nullptr, bool flags_cleared = false) {
  if (!flags_cleared) VUS_CHECK_HIP(hipMemsetAsync(status, 0, sizeof(int), st));
  BandSet S;
  S.count = 1;
  S.s[0] = BandSys{Sband, y, status, nullptr, n_nodes};
  if (win_scratch) {      // window_doubles(n_nodes, all panels)
    S.s[0].win_pub = win_scratch;
    S.s[0].win_F = reinterpret_cast<int*>(win_scratch + (size_t)((n_nodes + PB - 1) / PB) * WIN_PUB);
  }
  S.s[1] = S.s[0];
  if (int rc = factor_launches(S, band, n_nodes, n_rhs, st, nullptr, flags_cleared)) return rc;
  return backsolve_launch(S, band, n_rhs, 0, st, false, flags_cleared && band > 0);
}

// ---------------------------------------------------------------------------------------------------------------
// Two-sided ("burn at both ends") solve.  The factorisation is a chain of n/8 dependent panel steps, each as long
// as one workgroup's dependent work (the launch is latency-bound, not flop-bound), so the chain is cut in two:
// poses 0 .. m-1 are eliminated top-down in place (system T = the first m + band block rows of Sband), poses
// n-1 .. n-m bottom-up on a pose-reversed copy (system R), both in the SAME launches, block-interleaved; what is
// left is the dense system of the n - 2m middle poses (>= band of them): its entries are the T window (updated in
// place, original entries included) plus R's window (started from zero: Schur contributions only).  It is factored
// and solved one-sided, its solution is pushed through the two "spikes" (the factor blocks that couple the middle
// to the last eliminated poses of either side) and both halves are back-substituted in one cooperative launch.
// Same arithmetic, the elimination order differs: results agree with the one-sided solve to round-off.
struct SplitPlan {
  int n, band, n_rhs, m, nT, n_mid, bm;
  size_t off_R, off_mid, off_yT, off_yR, off_yM, off_int, off_winT, off_winR, off_winM, total;
};

bool split_plan(int n, int band, int n_rhs, SplitPlan& p) {
  p.n = n; p.band = band; p.n_rhs = n_rhs;
  p.m = band > 0 ? ((n - band) / 2 / PB) * PB : 0;
  if (p.m < PB) return false;
  p.nT = p.m + band;
  p.n_mid = n - 2 * p.m;
  p.bm = band < p.n_mid - 1 ? band : p.n_mid - 1;
  size_t o = 0;
  p.off_R = o;   o += 36 * (size_t)p.nT * (band + 1);
  p.off_mid = o; o += 36 * (size_t)p.n_mid * (p.bm + 1);
  p.off_yT = o;  o += 6 * (size_t)p.nT * n_rhs;
  p.off_yR = o;  o += 6 * (size_t)p.nT * n_rhs;
  p.off_yM = o;  o += 6 * (size_t)p.n_mid * n_rhs;
  p.off_int = o; o += 8;
  // scratch of the persistent window kernel: the two halves (m / PB panels each) and the middle system
  p.off_winT = o; o += window_doubles(p.nT, p.m / PB);
  p.off_winR = o; o += window_doubles(p.nT, p.m / PB);
  p.off_winM = o; o += window_doubles(p.n_mid, (p.n_mid + PB - 1) / PB);
  p.total = o;
  return true;
}

// Every flag word and status word of a two-sided solve, zeroed by its first kernel instead of by a memset launch each
// (nine launches of ~5 us on the solve's critical path).
struct ClearList {
  int* p[8];
  int n[8];
  int count = 0;
  void add(int* q, size_t k) {
    p[count] = q;
    n[count++] = (int)k;
  }
};

// Rb(i', s) = transpose of Sband(n-1-i'+s, s) -- the pose-reversed matrix in the same lower-band layout -- except the
// middle x middle region (both reversed poses >= m), which starts from zero; yT = y[.. nT), yR = reversed y, zero on
// the middle poses.
// cl: the solve's flag and status words (one of them inside Sband: the unused slots of block row 0, which this kernel
// does not read).
__global__ void split_prepare_kernel(const double* Sband, const double* __restrict__ y, SplitPlan p,
                                     double* __restrict__ Rb, double* __restrict__ yT, double* __restrict__ yR, ClearList cl) {
  const size_t t = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  for (int k = 0; k < cl.count; ++k)
    if (blockIdx.x == (unsigned)k % gridDim.x)
      for (int i = threadIdx.x; i < cl.n[k]; i += blockDim.x) cl.p[k][i] = 0;
  const size_t nR = 36 * (size_t)p.nT * (p.band + 1);
  if (t < nR) {
    const int e = (int)(t % 36);
    const size_t blk = t / 36;
    const int s = (int)(blk % (p.band + 1)), ip = (int)(blk / (p.band + 1));
    const int kp = ip - s;
    double v = 0.0;
    if (kp >= 0 && kp < p.m) {
      const int r = e / 6, c = e - 6 * r;
      v = Sband[36 * ((size_t)(p.n - 1 - kp) * (p.band + 1) + s) + 6 * c + r];
    }
    Rb[t] = v;
  }
  const size_t ny = 6 * (size_t)p.nT * p.n_rhs;
  if (t < ny) {
    const int q = (int)(t / (6 * (size_t)p.nT)), rem = (int)(t - (size_t)q * 6 * p.nT);
    const int i = rem / 6, c = rem - 6 * i;
    const double* yq = y + (size_t)q * 6 * p.n;
    yT[t] = yq[6 * (size_t)i + c];
    yR[t] = i < p.m ? yq[6 * (size_t)(p.n - 1 - i) + c] : 0.0;
  }
}

// Mid(u, s) = Sband(m + u, s) [T's window, or untouched original rows past it] + transpose of R's window block;
// diagonal blocks are symmetrised (the factorisation maintains their lower triangles only).  yM likewise.
__global__ void split_mid_kernel(const double* __restrict__ Sband, const double* __restrict__ y, SplitPlan p,
                                 const double* __restrict__ Rb, const double* __restrict__ yT,
                                 const double* __restrict__ yR, double* __restrict__ Mid, double* __restrict__ yM) {
  const size_t t = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  const size_t nM = 36 * (size_t)p.n_mid * (p.bm + 1);
  if (t < nM) {
    const int e = (int)(t % 36);
    const size_t blk = t / 36;
    const int s = (int)(blk % (p.bm + 1)), u = (int)(blk / (p.bm + 1));
    int r = e / 6, c = e - 6 * r;
    double v = 0.0;
    if (s <= u) {
      if (s == 0 && c > r) { const int tmp = r; r = c; c = tmp; }     // mirror the lower triangle of a diagonal block
      v = Sband[36 * ((size_t)(p.m + u) * (p.band + 1) + s) + 6 * r + c];
      const int ip = p.n - 1 - (p.m + u - s);                          // reversed index of the block's column pose
      if (ip < p.nT) {
        const size_t rb = 36 * ((size_t)ip * (p.band + 1) + s);
        v += s == 0 ? Rb[rb + 6 * r + c] : Rb[rb + 6 * c + r];
      }
    }
    Mid[t] = v;
  }
  const size_t ny = 6 * (size_t)p.n_mid * p.n_rhs;
  if (t < ny) {
    const int q = (int)(t / (6 * (size_t)p.n_mid)), rem = (int)(t - (size_t)q * 6 * p.n_mid);
    const int u = rem / 6, c = rem - 6 * u;
    const int i = p.m + u, ip = p.n - 1 - i;
    double v = i < p.nT ? yT[(size_t)q * 6 * p.nT + 6 * (size_t)i + c] : y[(size_t)q * 6 * p.n + 6 * (size_t)i + c];
    if (ip < p.nT) v += yR[(size_t)q * 6 * p.nT + 6 * (size_t)ip + c];
    yM[t] = v;
  }
}

// y_k -= sum_{i >= m, i - k <= band} L(i, k)^T x_i for the eliminated poses k < m next to the middle, both systems
// (blockIdx.y).  The blocks (i >= m, k < m) lie left of pose i's diagonal panel: stored transposed, so that
// (L^T x)[c] = sum_r stored[6 c + r] x[r].  One wave per (pose k, right-hand side); lanes split (i, c).
__global__ __launch_bounds__(64) void split_spike_kernel(const double* __restrict__ Sband, const double* __restrict__ Rb,
                                                         SplitPlan p, const double* __restrict__ yM,
                                                         double* __restrict__ yT, double* __restrict__ yR) {
  const int sysi = blockIdx.y, q = blockIdx.z;
  const int k = p.m - 1 - (int)blockIdx.x;
  if (k < 0) return;
  const double* Sb = sysi == 0 ? Sband : Rb;
  double* ys = (sysi == 0 ? yT : yR) + (size_t)q * 6 * p.nT;
  const double* xm = yM + (size_t)q * 6 * p.n_mid;
  const int lane = threadIdx.x, c = lane % 6, sl = lane / 6;     // 10 slices of poses x 6 outputs (lanes 60..63 idle)
  const int i_hi = min(k + p.band, p.nT - 1);
  double acc = 0.0;
  if (sl < 10)
    for (int i = p.m + sl; i <= i_hi; i += 10) {
      const double* b = Sb + 36 * ((size_t)i * (p.band + 1) + (i - k)) + 6 * c;
      const int u = sysi == 0 ? i - p.m : p.n - 1 - i - p.m;       // middle index of (reversed) pose i
      const double* x = xm + 6 * (size_t)u;
      acc += b[0] * x[0] + b[1] * x[1] + b[2] * x[2] + b[3] * x[3] + b[4] * x[4] + b[5] * x[5];
    }
  __shared__ double s_acc[64];
  s_acc[lane] = sl < 10 ? acc : 0.0;
  __syncthreads();
  if (lane < 6) {
    double t = 0.0;
    for (int j = 0; j < 10; ++j) t += s_acc[6 * j + lane];
    ys[6 * (size_t)k + lane] -= t;
  }
}

__global__ void split_gather_kernel(SplitPlan p, const double* __restrict__ yT, const double* __restrict__ yR,
                                    const double* __restrict__ yM, double* __restrict__ y, const int* __restrict__ st_R,
                                    const int* __restrict__ st_M, int* __restrict__ status) {
  const size_t t = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  if (t < 6 * (size_t)p.n * p.n_rhs) {
    const int q = (int)(t / (6 * (size_t)p.n)), rem = (int)(t - (size_t)q * 6 * p.n);
    const int i = rem / 6, c = rem - 6 * i;
    double v;
    if (i < p.m) v = yT[(size_t)q * 6 * p.nT + 6 * (size_t)i + c];
    else if (i < p.n - p.m) v = yM[(size_t)q * 6 * p.n_mid + 6 * (size_t)(i - p.m) + c];
    else v = yR[(size_t)q * 6 * p.nT + 6 * (size_t)(p.n - 1 - i) + c];
    y[t] = v;
  }
  if (t == 0) {
    // worst status of the three systems: an expired wait (-1) first, else the first non-positive pivot met, reported
    // as a global scalar column + 1 (the column inside its own system for T, mapped back for R and the middle)
    const int a = status[0], b = st_R[0], c = st_M[0];
    int out = a;
    if (a < 0 || b < 0 || c < 0) out = min(a < 0 ? a : 0, min(b < 0 ? b : 0, c < 0 ? c : 0));    // -3 (window) before -1
    else if (a == 0 && b > 0) out = 6 * (p.n - 1 - (b - 1) / 6) + (b - 1) % 6 + 1;
    else if (a == 0 && c > 0) out = 6 * p.m + c;
    status[0] = out;
  }
}

int band_solve_split_impl(double* Sband, int n, int band, double* y, int n_rhs, int* status, double* work, hipStream_t st) {
  SplitPlan p;
  if (!split_plan(n, band, n_rhs, p)) return band_solve_impl(Sband, n, band, y, n_rhs, status, st);
  double* Rb = work + p.off_R;
  double* Mid = work + p.off_mid;
  double* yT = work + p.off_yT;
  double* yR = work + p.off_yR;
  double* yM = work + p.off_yM;
  int* st_R = reinterpret_cast<int*>(work + p.off_int);
  int* st_M = st_R + 2;
  BandSet S;
  S.count = 2;
  S.s[0] = BandSys{Sband, yT, status, nullptr, p.nT};
  S.s[1] = BandSys{Rb, yR, st_R, nullptr, p.nT};
  {
    double* w[2] = {work + p.off_winT, work + p.off_winR};
    for (int q = 0; q < 2; ++q) {
      S.s[q].win_pub = w[q];
      S.s[q].win_F = reinterpret_cast<int*>(w[q] + (size_t)(p.m / PB) * WIN_PUB);
    }
  }
  // Status words, the window kernels' flags (halves and middle system) and the flag words of the halves' and the middle
  // system's back-substitution are zeroed by split_prepare_kernel.  The back-substitution's words are the unused slots of
  // block row 0 (Sb + 36): the top half's are cleared explicitly; the reversed copy and the middle system are written
  // whole by split_prepare_kernel / split_mid_kernel, zeros in every slot the band does not store.
  ClearList cl;
  const int NT_half = (p.nT + PB - 1) / PB, NT_mid = (p.n_mid + PB - 1) / PB;
  double* winM = work + p.off_winM;
  cl.add(status, 1);
  cl.add(st_R, 4);
  cl.add(S.s[0].win_F, 2 + 2 * (size_t)NT_half);
  cl.add(S.s[1].win_F, 2 + 2 * (size_t)NT_half);
  cl.add(reinterpret_cast<int*>(winM + (size_t)NT_mid * WIN_PUB), 2 + 2 * (size_t)NT_mid);
  cl.add(reinterpret_cast<int*>(Sband + 36), 2 + (size_t)((band + PB - 1) / PB));
  const size_t nR = 36 * (size_t)p.nT * (band + 1);
  split_prepare_kernel<<<cdiv((long long)nR, 256), 256, 0, st>>>(Sband, y, p, Rb, yT, yR, cl);
  // the halves are independent chains until the middle system: the pose-reversed one runs on a second stream
  const int mode_knob = g_knobs.band_mode.load(std::memory_order_relaxed);
  SplitSection sec((mode_knob < 0 || mode_knob >= 2) ? split_aux(st) : nullptr, st);
  // (the persistent window kernel serves both halves in one launch on the caller's stream: nothing to fork for)
  const bool halves_on_two_streams = sec.a && !((mode_knob < 0 || mode_knob == 3) && window_applicable(S, band, p.m));
  if (halves_on_two_streams)
    if (int rc = sec.fork()) return rc;
  if (int rc = factor_launches(S, band, p.m, n_rhs, st, halves_on_two_streams ? sec.a->s2 : nullptr, true)) return rc;
  if (halves_on_two_streams)
    if (int rc = sec.join()) return rc;
  // the diagonal panels of both halves are inverted (for their back-substitution) beside the middle system's solve
  bool halves_inverted = false;
  if (sec.a) {
    if (int rc = sec.fork()) return rc;
    halves_inverted = diag_invert_launch(S, band, p.m, sec.a->s2);
  }
  const size_t nM = 36 * (size_t)p.n_mid * (p.bm + 1);
  split_mid_kernel<<<cdiv((long long)nM, 256), 256, 0, st>>>(Sband, y, p, Rb, yT, yR, Mid, yM);
  if (int rc = band_solve_impl(Mid, p.n_mid, p.bm, yM, n_rhs, st_M, st, work + p.off_winM, true)) return rc;
  const int n_spike = band < p.m ? band : p.m;
  split_spike_kernel<<<dim3(n_spike, 2, n_rhs), 64, 0, st>>>(Sband, Rb, p, yM, yT, yR);
  if (sec.open)
    if (int rc = sec.join()) return rc;
  if (int rc = backsolve_launch(S, band, n_rhs, p.m, st, halves_inverted, true)) return rc;     // the eliminated poses of both halves
  split_gather_kernel<<<cdiv(6ll * n * n_rhs, 256), 256, 0, st>>>(p, yT, yR, yM, y, st_R, st_M, status);
  VUS_CHECK_LAUNCH("ba_band_solve_split");
  return VUS_OK;
}
}  // namespace

extern "C" int vus_ba_set_tuning(int knob, int value) {
  switch (knob) {
    case VUS_TUNE_BAND_MODE:
      VUS_REQUIRE(value >= -1 && value <= 3, "band mode %d out of range [-1, 3]", value);
      g_knobs.band_mode = value;
      return VUS_OK;
    case VUS_TUNE_CB_MAX_WG:
      VUS_REQUIRE(value >= 0, "workgroup cap %d is negative", value);
      g_knobs.cb_max_wg = value;
      return VUS_OK;
    case VUS_TUNE_WIN_FAULT:
      VUS_REQUIRE(value == 0 || value == 1, "fault injection knob takes 0 or 1, not %d", value);
      g_knobs.win_fault = value;
      return VUS_OK;
    default:
      return vus::fail(VUS_E_INVALID, "unknown tuning knob %d", knob);
  }
}

extern "C" int vus_ba_get_tuning(int knob) {
  if (knob == VUS_TUNE_BAND_MODE) return g_knobs.band_mode.load();
  if (knob == VUS_TUNE_CB_MAX_WG) return g_knobs.cb_max_wg.load();
  if (knob == VUS_TUNE_LAST_BAND_MODE) return g_knobs.last_mode.load();
  if (knob == VUS_TUNE_WIN_FAULT) return g_knobs.win_fault.load();
  return vus::fail(VUS_E_INVALID, "unknown tuning knob %d", knob);
}

extern "C" int vus_ba_band_solve(double* Sband, int n_poses, int band, const double* gs, double* dp, int* status,
                                 void* stream) {
  VUS_REQUIRE(Sband && gs && dp && status, "null buffer");
  VUS_REQUIRE(n_poses >= 1 && band >= 0, "n_poses=%d band=%d", n_poses, band);
  hipStream_t st = vus::as_stream(stream);
  negate_copy_kernel<<<cdiv(6ll * n_poses, 256), 256, 0, st>>>(gs, dp, 6 * n_poses);
  return band_solve_impl(Sband, n_poses, band, dp, 1, status, st);
}

extern "C" int vus_ba_band_solve_multi(double* Sband, int n_nodes, int band, double* rhs, int n_rhs, int* status,
                                       void* stream) {
  VUS_REQUIRE(Sband && rhs && status, "null buffer");
  VUS_REQUIRE(n_nodes >= 1 && band >= 0, "n_nodes=%d band=%d", n_nodes, band);
  VUS_REQUIRE(n_rhs >= 1 && n_rhs <= BS_MAX_RHS, "n_rhs=%d out of range [1, %d]", n_rhs, BS_MAX_RHS);
  return band_solve_impl(Sband, n_nodes, band, rhs, n_rhs, status, vus::as_stream(stream));
}

extern "C" long long vus_ba_band_solve_work_doubles(int n_nodes, int band, int n_rhs) {
  SplitPlan p;
  if (n_nodes < 1 || band < 0 || n_rhs < 1 || n_rhs > BS_MAX_RHS || !split_plan(n_nodes, band, n_rhs, p)) return 0;
  return (long long)p.total;
}

extern "C" int vus_ba_band_solve_split(double* Sband, int n_poses, int band, const double* gs, double* dp, int* status,
                                       double* work, void* stream) {
  VUS_REQUIRE(Sband && gs && dp && status && work, "null buffer");
  VUS_REQUIRE(n_poses >= 1 && band >= 0, "n_poses=%d band=%d", n_poses, band);
  hipStream_t st = vus::as_stream(stream);
  negate_copy_kernel<<<cdiv(6ll * n_poses, 256), 256, 0, st>>>(gs, dp, 6 * n_poses);
  return band_solve_split_impl(Sband, n_poses, band, dp, 1, status, work, st);
}

extern "C" int vus_ba_band_solve_multi_split(double* Sband, int n_nodes, int band, double* rhs, int n_rhs, int* status,
                                             double* work, void* stream) {
  VUS_REQUIRE(Sband && rhs && status && work, "null buffer");
  VUS_REQUIRE(n_nodes >= 1 && band >= 0, "n_nodes=%d band=%d", n_nodes, band);
  VUS_REQUIRE(n_rhs >= 1 && n_rhs <= BS_MAX_RHS, "n_rhs=%d out of range [1, %d]", n_rhs, BS_MAX_RHS);
  return band_solve_split_impl(Sband, n_nodes, band, rhs, n_rhs, status, work, vus::as_stream(stream));
}

extern "C" int vus_ba_backsub(const vus_ba_problem* P, const double* W, const double* Vinv, const double* gl,
                              const double* dp, double* dl, void* stream) {
  if (int rc = check_problem(P)) return rc;
  VUS_REQUIRE(dp != nullptr, "null buffer");
  VUS_REQUIRE((Vinv && gl && dl) || !P->n_points, "null landmark buffer");
  VUS_REQUIRE(W || !P->n_obs, "null observation buffer");
  if (P->n_points > 0)
    backsub_kernel<<<cdiv(P->n_points, 4), 256, 0, vus::as_stream(stream)>>>(*P, W, Vinv, gl, dp, dl);
  VUS_CHECK_LAUNCH("ba_backsub");
  return VUS_OK;
}

extern "C" int vus_ba_eval_step(const vus_ba_problem* P, const double* poses, const double* points, const double* dp,
                                const double* dl, double* new_poses, double* new_points, double* out, double* work,
                                void* stream) {
  if (int rc = check_problem(P)) return rc;
  VUS_REQUIRE(poses && dp && new_poses && out && work, "null buffer");
  VUS_REQUIRE((points && dl && new_points) || !P->n_points, "null landmark buffer");
  hipStream_t st = vus::as_stream(stream);
  const int nP = P->n_poses, nL = P->n_points;
  retract_kernel<<<cdiv(nP > nL ? nP : (nL < 65536 ? nL : 65536), 256) + 1, 256, 0, st>>>(nP, nL, pose_stride(*P), poses, points, dp, dl,
                                                                                           new_poses, new_points);
  double* part_lin = work;
  double* part_new = work + (nL + 1);
  if (nL > 0)
    eval_points_kernel<true><<<cdiv(nL, 4), 256, 0, st>>>(*P, poses, points, dp, dl, new_poses, new_points, part_lin,
                                                         part_new);
  priors_kernel<<<1, 64, 0, st>>>(*P, poses, dp, nullptr, nullptr, part_lin + nL, 2);
  priors_kernel<<<1, 64, 0, st>>>(*P, new_poses, nullptr, nullptr, nullptr, part_new + nL, 1);
  reduce_partials_kernel<<<1, 1024, 0, st>>>(part_lin, nL + 1, out);
  reduce_partials_kernel<<<1, 1024, 0, st>>>(part_new, nL + 1, out + 1);
  VUS_CHECK_LAUNCH("ba_eval_step");
  return VUS_OK;
}

#ifdef VUS_TIMING
// timing builds only (tools/win_timing.py): the s_memtime marks of the window kernel's critical workgroup, panel 41
extern "C" int vus_debug_read_wtm(unsigned long long* out32) {
  return hipMemcpyFromSymbol(out32, HIP_SYMBOL(g_wtm), sizeof(g_wtm)) == hipSuccess ? 0 : -1;
}
#endif
