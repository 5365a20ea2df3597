// nav.hip -- camera-side navigation factors for gfx950:
// gtsam.ImuFactor over preintegrated measurements (reference batch.py:237-239,289-293), the DVL
// velocity factor (batch.py:196-250, with the correct Jacobians) and PriorFactorVector on velocities
// (batch.py:282).  O(#keyframes) work next to the O(#observations) stereo kernels of ba.hip.
//
// Node layout (vus_ba_problem.pose_stride = 2): node 2i = X(i), node 2i+1 = V(i) padded to 6 dims
// (dims 3..5 are inert: unit diagonal, zero right-hand side); the shared bias B(0) is a 6-wide BORDER of
// the reduced camera system, eliminated after the band solve with 7 right-hand sides.
//
// Factors are evaluated one per thread.  Their J^T J blocks on the camera side go in with f64 atomics (a block
// receives at most two IMU factors, one DVL factor and one prior, so the sums differ between runs by the order of
// at most four addends, ~1e-16 relative); the bias-bias block and the bias gradient, which every IMU factor
// touches, are written per factor and reduced in a fixed order.
#include <cmath>
#include <cstring>
#include "vus_common.h"

namespace {

constexpr int PIM_DT = 0, PIM_DR = 1, PIM_DP = 10, PIM_DV = 13, PIM_DR_DBG = 16, PIM_DP_DBA = 25, PIM_DP_DBG = 34,
              PIM_DV_DBA = 43, PIM_DV_DBG = 52, PIM_BIAS = 61, PIM_N = 148;
constexpr double kEps = 2.220446049250313e-16;
constexpr double kPi = 3.14159265358979323846;

__device__ __forceinline__ void skew(const double* w, double* S) {
  S[0] = 0; S[1] = -w[2]; S[2] = w[1]; S[3] = w[2]; S[4] = 0; S[5] = -w[0]; S[6] = -w[1]; S[7] = w[0]; S[8] = 0;
}
__device__ __forceinline__ void mm(const double* A, const double* B, double* C) {
#pragma unroll
  for (int r = 0; r < 3; ++r)
#pragma unroll
    for (int c = 0; c < 3; ++c) C[3 * r + c] = A[3 * r] * B[c] + A[3 * r + 1] * B[3 + c] + A[3 * r + 2] * B[6 + c];
}
__device__ __forceinline__ void mtm(const double* A, const double* B, double* C) {   // A^T B
#pragma unroll
  for (int r = 0; r < 3; ++r)
#pragma unroll
    for (int c = 0; c < 3; ++c) C[3 * r + c] = A[r] * B[c] + A[3 + r] * B[3 + c] + A[6 + r] * B[6 + c];
}
__device__ __forceinline__ void mv(const double* A, const double* v, double* o) {
#pragma unroll
  for (int r = 0; r < 3; ++r) o[r] = A[3 * r] * v[0] + A[3 * r + 1] * v[1] + A[3 * r + 2] * v[2];
}
__device__ __forceinline__ void mtv(const double* A, const double* v, double* o) {
#pragma unroll
  for (int r = 0; r < 3; ++r) o[r] = A[r] * v[0] + A[3 + r] * v[1] + A[6 + r] * v[2];
}

__device__ void so3_exp(const double* w, double* R) {
  const double th2 = w[0] * w[0] + w[1] * w[1] + w[2] * w[2];
  double W[9], WW[9];
  skew(w, W);
  if (th2 <= kEps) {
#pragma unroll
    for (int i = 0; i < 9; ++i) R[i] = W[i] + (i % 4 == 0 ? 1.0 : 0.0);
    return;
  }
  const double th = sqrt(th2), s = sin(th) / th, sh = sin(0.5 * th), c = 2.0 * sh * sh / th2;
  mm(W, W, WW);
#pragma unroll
  for (int i = 0; i < 9; ++i) R[i] = (i % 4 == 0 ? 1.0 : 0.0) + s * W[i] + c * WW[i];
}
__device__ void so3_log(const double* R, double* w) {
  const double tr = R[0] + R[4] + R[8];
  if (tr + 1.0 < 1e-10) {
    if (fabs(R[8] + 1.0) > 1e-5) { double k = kPi / sqrt(2.0 + 2.0 * R[8]); w[0] = k * R[2]; w[1] = k * R[5]; w[2] = k * (1.0 + R[8]); }
    else if (fabs(R[4] + 1.0) > 1e-5) { double k = kPi / sqrt(2.0 + 2.0 * R[4]); w[0] = k * R[1]; w[1] = k * (1.0 + R[4]); w[2] = k * R[7]; }
    else { double k = kPi / sqrt(2.0 + 2.0 * R[0]); w[0] = k * (1.0 + R[0]); w[1] = k * R[3]; w[2] = k * R[6]; }
    return;
  }
  double mag;
  const double tr3 = tr - 3.0;
  if (tr3 < -1e-7) { const double th = acos((tr - 1.0) / 2.0); mag = th / (2.0 * sin(th)); }
  else mag = 0.5 - tr3 / 12.0;
  w[0] = mag * (R[7] - R[5]); w[1] = mag * (R[2] - R[6]); w[2] = mag * (R[3] - R[1]);
}
__device__ void so3_jr(const double* w, double* J) {
  const double th2 = w[0] * w[0] + w[1] * w[1] + w[2] * w[2];
  double W[9], WW[9];
  skew(w, W); mm(W, W, WW);
  double a, b;
  if (th2 < 1e-10) { a = 0.5 - th2 / 24.0; b = 1.0 / 6.0 - th2 / 120.0; }
  else { const double th = sqrt(th2); a = (1.0 - cos(th)) / th2; b = (th - sin(th)) / (th2 * th); }
#pragma unroll
  for (int i = 0; i < 9; ++i) J[i] = (i % 4 == 0 ? 1.0 : 0.0) - a * W[i] + b * WW[i];
}
__device__ void so3_jr_inv(const double* w, double* J) {
  const double th2 = w[0] * w[0] + w[1] * w[1] + w[2] * w[2];
  double W[9], WW[9];
  skew(w, W); mm(W, W, WW);
  double b;
  if (th2 < 1e-10) b = 1.0 / 12.0 + th2 / 720.0;
  else { const double th = sqrt(th2); b = 1.0 / th2 - (1.0 + cos(th)) / (2.0 * th * sin(th)); }
#pragma unroll
  for (int i = 0; i < 9; ++i) J[i] = (i % 4 == 0 ? 1.0 : 0.0) + 0.5 * W[i] + b * WW[i];
}

// ImuFactor: unwhitened residual r[9] = (theta, p, v) and, when J != nullptr, the Jacobian J[9][24] with
// columns pose_i(6) vel_i(3) pose_j(6) vel_j(3) bias(6).  Forster et al. 2017 / gtsam ImuFactor.
__device__ void imu_factor(const double* Ti, const double* vi, const double* Tj, const double* vj, const double* bias,
                           const double* pim, const double* g, double* r, double* J) {
  const double dt = pim[PIM_DT];
  double dba[3], dbg[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) { dba[k] = bias[k] - pim[PIM_BIAS + k]; dbg[k] = bias[3 + k] - pim[PIM_BIAS + 3 + k]; }
  double phi[3], Ephi[9], dRc[9], dPc[3], dVc[3], t3[3], t3b[3];
  mv(pim + PIM_DR_DBG, dbg, phi);
  so3_exp(phi, Ephi);
  mm(pim + PIM_DR, Ephi, dRc);
  mv(pim + PIM_DP_DBA, dba, t3); mv(pim + PIM_DP_DBG, dbg, t3b);
#pragma unroll
  for (int k = 0; k < 3; ++k) dPc[k] = pim[PIM_DP + k] + t3[k] + t3b[k];
  mv(pim + PIM_DV_DBA, dba, t3); mv(pim + PIM_DV_DBG, dbg, t3b);
#pragma unroll
  for (int k = 0; k < 3; ++k) dVc[k] = pim[PIM_DV + k] + t3[k] + t3b[k];
  const double* Ri = Ti; const double* pi = Ti + 9;
  const double* Rj = Tj; const double* pj = Tj + 9;
  double RjtRi[9], E[9], rR[3];
  mtm(Rj, Ri, RjtRi);
  mm(RjtRi, dRc, E);
  so3_log(E, rR);
  double RidP[3], RidV[3], dpw[3], dvw[3], rP[3], rV[3];
  mv(Ri, dPc, RidP); mv(Ri, dVc, RidV);
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    dpw[k] = pi[k] + vi[k] * dt + 0.5 * g[k] * dt * dt + RidP[k] - pj[k];
    dvw[k] = vi[k] + g[k] * dt + RidV[k] - vj[k];
  }
  mtv(Rj, dpw, rP); mtv(Rj, dvw, rV);
#pragma unroll
  for (int k = 0; k < 3; ++k) { r[k] = rR[k]; r[3 + k] = rP[k]; r[6 + k] = rV[k]; }
  if (!J) return;
  for (int k = 0; k < 9 * 24; ++k) J[k] = 0.0;
  double JrInv[9], JrInvNeg[9], M[9], M2[9], X[9], JrPhi[9];
  const double nrR[3] = {-rR[0], -rR[1], -rR[2]};
  so3_jr_inv(rR, JrInv);
  so3_jr_inv(nrR, JrInvNeg);
#define JSET(row0, col0, Mat, sgn)                                                         \
  for (int a = 0; a < 3; ++a)                                                              \
    for (int b = 0; b < 3; ++b) J[24 * ((row0) + a) + (col0) + b] = (sgn) * (Mat)[3 * a + b]
  for (int a = 0; a < 3; ++a) for (int b = 0; b < 3; ++b) M[3 * a + b] = dRc[3 * b + a];
  mm(JrInv, M, M2);
  JSET(0, 0, M2, 1.0);
  JSET(0, 9, JrInvNeg, -1.0);
  so3_jr(phi, JrPhi);
  mm(JrInv, JrPhi, M); mm(M, pim + PIM_DR_DBG, M2);
  JSET(0, 21, M2, 1.0);
  skew(dPc, X); mm(RjtRi, X, M);
  JSET(3, 0, M, -1.0);
  JSET(3, 3, RjtRi, 1.0);
  for (int a = 0; a < 3; ++a) for (int b = 0; b < 3; ++b) M[3 * a + b] = Rj[3 * b + a] * dt;
  JSET(3, 6, M, 1.0);
  skew(rP, X);
  JSET(3, 9, X, 1.0);
  for (int a = 0; a < 3; ++a) J[24 * (3 + a) + 12 + a] = -1.0;
  mm(RjtRi, pim + PIM_DP_DBA, M); JSET(3, 18, M, 1.0);
  mm(RjtRi, pim + PIM_DP_DBG, M); JSET(3, 21, M, 1.0);
  skew(dVc, X); mm(RjtRi, X, M);
  JSET(6, 0, M, -1.0);
  for (int a = 0; a < 3; ++a) for (int b = 0; b < 3; ++b) M[3 * a + b] = Rj[3 * b + a];
  JSET(6, 6, M, 1.0);
  skew(rV, X);
  JSET(6, 9, X, 1.0);
  JSET(6, 15, M, -1.0);
  mm(RjtRi, pim + PIM_DV_DBA, M); JSET(6, 18, M, 1.0);
  mm(RjtRi, pim + PIM_DV_DBG, M); JSET(6, 21, M, 1.0);
#undef JSET
}

// scratch record of one evaluated factor: whitened Jacobian rows followed by the whitened residual
constexpr int IMU_REC = 9 * 25;   // Jw[9][24] | rw[9] stored as row a: 24 J entries + 1 residual
constexpr int DVL_REC = 3 * 10;   // Jw[3][9]  | rw[3]

// mode 0: Jacobians + residual into the scratch records;  mode 1: error only (part[f]);
// mode 2: linearised error 0.5 |rw + Jw d|^2 with d from (dc, db) AND nothing else (part[f]).
__global__ void nav_imu_kernel(vus_nav_factors N, int n_poses, const double* __restrict__ poses,
                               const double* __restrict__ vels, const double* __restrict__ bias,
                               const double* __restrict__ dc, const double* __restrict__ db,
                               double* __restrict__ rec, double* __restrict__ part, int mode) {
  const int f = blockIdx.x * blockDim.x + threadIdx.x;
  if (f >= N.n_imu) return;
  const int i = N.imu_i[f], j = N.imu_j[f];
  double r[9], J[9 * 24];
  imu_factor(poses + 12 * (size_t)i, vels + 3 * (size_t)i, poses + 12 * (size_t)j, vels + 3 * (size_t)j, bias,
             N.imu_pim + PIM_N * (size_t)f, N.gravity, r, mode == 1 ? nullptr : J);
  const double* W = N.imu_W + 81 * (size_t)f;
  double e = 0.0;
  for (int a = 0; a < 9; ++a) {
    double rw = 0.0;
    for (int k = 0; k < 9; ++k) rw += W[9 * a + k] * r[k];
    if (mode != 1) {
      double* out = rec + IMU_REC * (size_t)f + 25 * a;
      for (int c = 0; c < 24; ++c) {
        double jw = 0.0;
        for (int k = 0; k < 9; ++k) jw += W[9 * a + k] * J[24 * k + c];
        if (mode == 0) out[c] = jw;
        else {
          // step component of column c: pose_i, vel_i, pose_j, vel_j live in node layout, bias in db
          double dcomp;
          if (c < 6) dcomp = dc[6 * (size_t)(2 * i) + c];
          else if (c < 9) dcomp = dc[6 * (size_t)(2 * i + 1) + c - 6];
          else if (c < 15) dcomp = dc[6 * (size_t)(2 * j) + c - 9];
          else if (c < 18) dcomp = dc[6 * (size_t)(2 * j + 1) + c - 15];
          else dcomp = db[c - 18];
          rw += jw * dcomp;
        }
      }
      if (mode == 0) out[24] = rw;
    }
    e += 0.5 * rw * rw;
  }
  part[f] = e;
}

__global__ void nav_dvl_kernel(vus_nav_factors N, const double* __restrict__ poses, const double* __restrict__ vels,
                               const double* __restrict__ dc, double* __restrict__ rec, double* __restrict__ part,
                               int mode) {
  const int f = blockIdx.x * blockDim.x + threadIdx.x;
  if (f >= N.n_dvl) return;
  const int i = N.dvl_pose[f];
  const double* T = poses + 12 * (size_t)i;
  const double* m = N.dvl_meas + 3 * (size_t)f;
  const double w = N.dvl_w[f];
  double Rm[3], X[9], M[9];
  mv(T, m, Rm);
  skew(m, X);
  mm(T, X, M);   // R [m]x
  double e = 0.0;
  for (int a = 0; a < 3; ++a) {
    double rw = w * (Rm[a] - vels[3 * (size_t)i + a]);
    double Jw[9];
    for (int b = 0; b < 3; ++b) { Jw[b] = -w * M[3 * a + b]; Jw[3 + b] = 0.0; Jw[6 + b] = (a == b) ? -w : 0.0; }
    if (mode == 0) {
      double* out = rec + DVL_REC * (size_t)f + 10 * a;
      for (int c = 0; c < 9; ++c) out[c] = Jw[c];
      out[9] = rw;
    } else if (mode == 2) {
      for (int c = 0; c < 6; ++c) rw += Jw[c] * dc[6 * (size_t)(2 * i) + c];
      for (int c = 0; c < 3; ++c) rw += Jw[6 + c] * dc[6 * (size_t)(2 * i + 1) + c];
    }
    e += 0.5 * rw * rw;
  }
  part[f] = e;
}

__global__ void nav_vprior_kernel(vus_nav_factors N, const double* __restrict__ vels, const double* __restrict__ dc,
                                  double* __restrict__ part, int mode) {
  const int f = blockIdx.x * blockDim.x + threadIdx.x;
  if (f >= N.n_vprior) return;
  const int i = N.vprior_idx[f];
  double e = 0.0;
  for (int k = 0; k < 3; ++k) {
    const double w = N.vprior_w[3 * (size_t)f + k];
    double r = w * (vels[3 * (size_t)i + k] - N.vprior_v[3 * (size_t)f + k]);
    if (mode == 2) r += w * dc[6 * (size_t)(2 * i + 1) + k];
    e += 0.5 * r * r;
  }
  part[f] = e;
}

// column of an IMU factor -> (node, dim); node -1 = the bias border
__device__ __forceinline__ void imu_col(int c, int i, int j, int& node, int& dim) {
  if (c < 6) { node = 2 * i; dim = c; }
  else if (c < 9) { node = 2 * i + 1; dim = c - 6; }
  else if (c < 15) { node = 2 * j; dim = c - 9; }
  else if (c < 18) { node = 2 * j + 1; dim = c - 15; }
  else { node = -1; dim = c - 18; }
}

// One workgroup adds the factors' J^T J / J^T r one factor after the other (fixed order => reproducible).
// Thread t < 576 owns entry (c1, c2) of the 24x24 block, threads 576..599 the gradient entries.
// One workgroup per factor.  A camera-side block receives at most two IMU factors, one DVL factor and
// one velocity prior, so these go in with f64 atomics; the bias-bias block and the bias gradient, which
// EVERY IMU factor touches, are written per factor and reduced in a fixed order afterwards.
constexpr int NAV_BIAS_PART = 42;   // 36 (Sbb) + 6 (gb) per IMU factor
__global__ __launch_bounds__(640) void nav_accumulate_imu_kernel(vus_nav_factors N, const double* __restrict__ rec_imu,
                                                                 double* __restrict__ Snav, double* __restrict__ Scb,
                                                                 double* __restrict__ gnav,
                                                                 double* __restrict__ bias_part) {
  const int t = threadIdx.x, f = blockIdx.x;
  const int i = N.imu_i[f], j = N.imu_j[f];
  const double* R = rec_imu + IMU_REC * (size_t)f;
  if (t < 576) {
    const int c1 = t / 24, c2 = t - 24 * c1;
    double h = 0.0;
#pragma unroll
    for (int a = 0; a < 9; ++a) h += R[25 * a + c1] * R[25 * a + c2];
    int n1, d1, n2, d2;
    imu_col(c1, i, j, n1, d1);
    imu_col(c2, i, j, n2, d2);
    if (n1 >= 0 && n2 >= 0) {
      if (n1 >= n2 && n1 - n2 <= 3) unsafeAtomicAdd(&Snav[36 * ((size_t)n1 * 4 + (n1 - n2)) + 6 * d1 + d2], h);
    } else if (n1 >= 0 && n2 < 0) {
      unsafeAtomicAdd(&Scb[36 * (size_t)n1 + 6 * d1 + d2], h);
    } else if (n1 < 0 && n2 < 0) {
      bias_part[NAV_BIAS_PART * (size_t)f + 6 * d1 + d2] = h;
    }
  } else if (t < 600) {
    const int c = t - 576;
    double gsum = 0.0;
#pragma unroll
    for (int a = 0; a < 9; ++a) gsum += R[25 * a + c] * R[25 * a + 24];
    int n1, d1;
    imu_col(c, i, j, n1, d1);
    if (n1 >= 0) unsafeAtomicAdd(&gnav[6 * (size_t)n1 + d1], gsum);
    else bias_part[NAV_BIAS_PART * (size_t)f + 36 + d1] = gsum;
  }
}

__global__ __launch_bounds__(64) void nav_bias_reduce_kernel(int n_imu, const double* __restrict__ bias_part,
                                                            double* __restrict__ Sbb, double* __restrict__ gb) {
  const int e = blockIdx.x, lane = threadIdx.x;
  double acc = 0.0;
  for (int f = lane; f < n_imu; f += 64) acc += bias_part[NAV_BIAS_PART * (size_t)f + e];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
  if (lane == 0) {
    if (e < 36) Sbb[e] = acc;
    else gb[e - 36] = acc;
  }
}

__global__ __launch_bounds__(128) void nav_accumulate_dvl_kernel(vus_nav_factors N, const double* __restrict__ rec_dvl,
                                                                 double* __restrict__ Snav, double* __restrict__ gnav) {
  const int t = threadIdx.x, f = blockIdx.x;
  const int i = N.dvl_pose[f];
  const double* R = rec_dvl + DVL_REC * (size_t)f;
  if (t < 81) {
    const int c1 = t / 9, c2 = t - 9 * c1;
    double h = 0.0;
#pragma unroll
    for (int a = 0; a < 3; ++a) h += R[10 * a + c1] * R[10 * a + c2];
    const int n1 = c1 < 6 ? 2 * i : 2 * i + 1, d1 = c1 < 6 ? c1 : c1 - 6;
    const int n2 = c2 < 6 ? 2 * i : 2 * i + 1, d2 = c2 < 6 ? c2 : c2 - 6;
    if (n1 >= n2) unsafeAtomicAdd(&Snav[36 * ((size_t)n1 * 4 + (n1 - n2)) + 6 * d1 + d2], h);
  } else if (t < 90) {
    const int c = t - 81;
    double gsum = 0.0;
#pragma unroll
    for (int a = 0; a < 3; ++a) gsum += R[10 * a + c] * R[10 * a + 9];
    const int n1 = c < 6 ? 2 * i : 2 * i + 1, d1 = c < 6 ? c : c - 6;
    unsafeAtomicAdd(&gnav[6 * (size_t)n1 + d1], gsum);
  }
}

// velocity priors are diagonal: one thread per coordinate, no conflicts with each other; run after the
// accumulate kernel (same stream)
__global__ void nav_vprior_accumulate_kernel(vus_nav_factors N, const double* __restrict__ vels,
                                             double* __restrict__ Snav, double* __restrict__ gnav) {
  if (blockIdx.x != 0 || threadIdx.x != 0) return;
  for (int f = 0; f < N.n_vprior; ++f) {   // several priors on one velocity are legal: keep it sequential
    const int node = 2 * N.vprior_idx[f] + 1;
    for (int k = 0; k < 3; ++k) {
      const double w = N.vprior_w[3 * (size_t)f + k];
      const double r = w * (vels[3 * (size_t)N.vprior_idx[f] + k] - N.vprior_v[3 * (size_t)f + k]);
      Snav[36 * ((size_t)node * 4) + 7 * k] += w * w;
      gnav[6 * (size_t)node + k] += w * r;
    }
  }
}

__global__ __launch_bounds__(1024) void reduce_kernel(const double* __restrict__ part, int n, double* __restrict__ out) {
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

__global__ void nav_assemble_kernel(int n_nodes, int band, double lambda, const double* __restrict__ Snav,
                                    const double* __restrict__ Scb, const double* __restrict__ gnav,
                                    double* __restrict__ Sband, double* __restrict__ gs, double* __restrict__ rhs) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= 36 * n_nodes) return;
  const int node = t / 36, e = t - 36 * node;
  const int smax = min(3, band);
  for (int s = 0; s <= smax; ++s) {
    double v = Snav[36 * ((size_t)node * 4 + s) + e];
    if (s == 0 && (node & 1) && e % 7 == 0) v += (e / 7 < 3) ? lambda : 1.0;   // velocity node: damping / padding
    Sband[36 * ((size_t)node * (band + 1) + s) + e] += v;
  }
  if (e < 6) {
    const size_t k = 6 * (size_t)node + e;
    const double g = gs[k] + gnav[k];
    gs[k] = g;
    rhs[k] = -g;
  }
  // columns 1..6 of the right-hand sides: the bias coupling, entry (6*node + d, q) = Scb[node][d][q]
  const int d = e / 6, q = e - 6 * d;
  rhs[(size_t)(1 + q) * 6 * n_nodes + 6 * (size_t)node + d] = Scb[t];
}

// (Sbb + lambda I - Scb^T Z) db = -gb - Scb^T z0;  dc = z0 - Z db.   One workgroup.
__global__ __launch_bounds__(1024) void nav_border_kernel(int n_nodes, const double* __restrict__ rhs,
                                                          const double* __restrict__ Scb, const double* __restrict__ Sbb,
                                                          const double* __restrict__ gb, double lambda,
                                                          double* __restrict__ dc, double* __restrict__ db) {
  __shared__ double s_red[16][42];
  __shared__ double s_db[6];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const size_t n = 6 * (size_t)n_nodes;
  double acc[42];   // [a][0..5]: (Scb^T Z)[a][b];  [36 + a]: (Scb^T z0)[a]
#pragma unroll
  for (int k = 0; k < 42; ++k) acc[k] = 0.0;
  for (size_t k = tid; k < n; k += 1024) {
    const size_t node = k / 6;
    const int d = (int)(k - 6 * node);
    const double* c = Scb + 36 * node + 6 * d;   // row d of the node's coupling block = (Scb)[k][0..5]
    const double z0 = rhs[k];
#pragma unroll
    for (int a = 0; a < 6; ++a) {
      acc[36 + a] += c[a] * z0;
#pragma unroll
      for (int b = 0; b < 6; ++b) acc[6 * a + b] += c[a] * rhs[(size_t)(1 + b) * n + k];
    }
  }
#pragma unroll
  for (int k = 0; k < 42; ++k) {
    double v = acc[k];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    if (lane == 0) s_red[wave][k] = v;
  }
  __syncthreads();
  if (tid == 0) {
    double M[36], v[6];
    for (int k = 0; k < 42; ++k) {
      double t = 0.0;
      for (int w = 0; w < 16; ++w) t += s_red[w][k];
      if (k < 36) M[k] = Sbb[k] + ((k % 7 == 0) ? lambda : 0.0) - t;
      else v[k - 36] = -gb[k - 36] - t;
    }
    // Cholesky of the 6x6 SPD border block
    for (int c = 0; c < 6; ++c) {
      double sdiag = M[7 * c];
      for (int k = 0; k < c; ++k) sdiag -= M[6 * c + k] * M[6 * c + k];
      const double l = sqrt(sdiag > 0.0 ? sdiag : 1e-300);
      M[7 * c] = l;
      for (int r = c + 1; r < 6; ++r) {
        double tt = M[6 * r + c];
        for (int k = 0; k < c; ++k) tt -= M[6 * r + k] * M[6 * c + k];
        M[6 * r + c] = tt / l;
      }
    }
    for (int r = 0; r < 6; ++r) { double tt = v[r]; for (int k = 0; k < r; ++k) tt -= M[6 * r + k] * v[k]; v[r] = tt / M[7 * r]; }
    for (int c = 5; c >= 0; --c) { double tt = v[c]; for (int r = c + 1; r < 6; ++r) tt -= M[6 * r + c] * v[r]; v[c] = tt / M[7 * c]; }
    for (int k = 0; k < 6; ++k) { s_db[k] = v[k]; db[k] = v[k]; }
  }
  __syncthreads();
  for (size_t k = tid; k < n; k += 1024) {
    double t = rhs[k];
#pragma unroll
    for (int b = 0; b < 6; ++b) t -= rhs[(size_t)(1 + b) * n + k] * s_db[b];
    dc[k] = t;
  }
}

__global__ void nav_retract_kernel(int n_poses, const double* __restrict__ vels, const double* __restrict__ bias,
                                   const double* __restrict__ dc, const double* __restrict__ db,
                                   double* __restrict__ new_vels, double* __restrict__ new_bias) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t < 3 * n_poses) new_vels[t] = vels[t] + dc[6 * (size_t)(2 * (t / 3) + 1) + t % 3];
  if (t < 6) new_bias[t] = bias[t] + db[t];
}

int check_nav(const vus_nav_factors* N, int n_poses) {
  VUS_REQUIRE(N != nullptr, "nav factors are null");
  VUS_REQUIRE(n_poses >= 1 && N->n_imu >= 0 && N->n_dvl >= 0 && N->n_vprior >= 0, "bad sizes");
  if (N->n_imu > 0) VUS_REQUIRE(N->imu_i && N->imu_j && N->imu_pim && N->imu_W, "imu arrays are null");
  if (N->n_dvl > 0) VUS_REQUIRE(N->dvl_pose && N->dvl_meas && N->dvl_w, "dvl arrays are null");
  if (N->n_vprior > 0) VUS_REQUIRE(N->vprior_idx && N->vprior_v && N->vprior_w, "velocity prior arrays are null");
  return VUS_OK;
}

inline int cdivi(long long a, int b) { return (int)((a + b - 1) / b); }

// error partials of all factor kinds into work[0 .. n_imu + n_dvl + n_vprior), then reduce into out[0]
int nav_errors(const vus_nav_factors* N, int n_poses, const double* poses, const double* vels, const double* bias,
               const double* dc, const double* db, int mode, double* rec_imu, double* rec_dvl, double* part,
               double* out, hipStream_t st) {
  if (N->n_imu > 0)
    nav_imu_kernel<<<cdivi(N->n_imu, 64), 64, 0, st>>>(*N, n_poses, poses, vels, bias, dc, db, rec_imu, part, mode);
  if (N->n_dvl > 0)
    nav_dvl_kernel<<<cdivi(N->n_dvl, 64), 64, 0, st>>>(*N, poses, vels, dc, rec_dvl, part + N->n_imu, mode);
  if (N->n_vprior > 0)
    nav_vprior_kernel<<<cdivi(N->n_vprior, 64), 64, 0, st>>>(*N, vels, dc, part + N->n_imu + N->n_dvl, mode);
  reduce_kernel<<<1, 1024, 0, st>>>(part, N->n_imu + N->n_dvl + N->n_vprior, out);
  VUS_CHECK_LAUNCH("nav_errors");
  return VUS_OK;
}

}  // namespace

// work layout: [records of the IMU factors | records of the DVL factors | error partials | bias partials]
extern "C" long long vus_nav_work_doubles(const vus_nav_factors* N) {
  if (!N) return 0;
  return (long long)IMU_REC * N->n_imu + (long long)DVL_REC * N->n_dvl + N->n_imu + N->n_dvl + N->n_vprior + 8 +
         (long long)NAV_BIAS_PART * N->n_imu;
}

extern "C" int vus_nav_linearize(const vus_nav_factors* N, int n_poses, const double* poses, const double* vels,
                                 const double* bias, double* Snav, double* Scb, double* Sbb, double* gnav, double* gb,
                                 double* err, double* work, void* stream) {
  if (int rc = check_nav(N, n_poses)) return rc;
  VUS_REQUIRE(poses && vels && bias && Snav && Scb && Sbb && gnav && gb && err && work, "null buffer");
  hipStream_t st = vus::as_stream(stream);
  const size_t n_nodes = 2 * (size_t)n_poses;
  VUS_CHECK_HIP(hipMemsetAsync(Snav, 0, sizeof(double) * 36 * 4 * n_nodes, st));
  VUS_CHECK_HIP(hipMemsetAsync(Scb, 0, sizeof(double) * 36 * n_nodes, st));
  VUS_CHECK_HIP(hipMemsetAsync(Sbb, 0, sizeof(double) * 36, st));
  VUS_CHECK_HIP(hipMemsetAsync(gnav, 0, sizeof(double) * 6 * n_nodes, st));
  VUS_CHECK_HIP(hipMemsetAsync(gb, 0, sizeof(double) * 6, st));
  double* rec_imu = work;
  double* rec_dvl = rec_imu + (size_t)IMU_REC * N->n_imu;
  double* part = rec_dvl + (size_t)DVL_REC * N->n_dvl;
  if (int rc = nav_errors(N, n_poses, poses, vels, bias, nullptr, nullptr, 0, rec_imu, rec_dvl, part, err, st)) return rc;
  double* bias_part = part + N->n_imu + N->n_dvl + N->n_vprior + 8;
  if (N->n_imu > 0) {
    nav_accumulate_imu_kernel<<<N->n_imu, 640, 0, st>>>(*N, rec_imu, Snav, Scb, gnav, bias_part);
    nav_bias_reduce_kernel<<<NAV_BIAS_PART, 64, 0, st>>>(N->n_imu, bias_part, Sbb, gb);
  }
  if (N->n_dvl > 0) nav_accumulate_dvl_kernel<<<N->n_dvl, 128, 0, st>>>(*N, rec_dvl, Snav, gnav);
  nav_vprior_accumulate_kernel<<<1, 64, 0, st>>>(*N, vels, Snav, gnav);
  VUS_CHECK_LAUNCH("nav_linearize");
  return VUS_OK;
}

extern "C" int vus_nav_assemble(int n_nodes, int band, double lambda, const double* Snav, const double* Scb,
                                const double* gnav, double* Sband, double* gs, double* rhs, void* stream) {
  VUS_REQUIRE(Snav && Scb && gnav && Sband && gs && rhs, "null buffer");
  VUS_REQUIRE(n_nodes >= 2 && (n_nodes & 1) == 0 && band >= 1 && lambda >= 0.0, "n_nodes=%d band=%d lambda=%g", n_nodes,
              band, lambda);
  nav_assemble_kernel<<<cdivi(36ll * n_nodes, 256), 256, 0, vus::as_stream(stream)>>>(n_nodes, band, lambda, Snav, Scb,
                                                                                    gnav, Sband, gs, rhs);
  VUS_CHECK_LAUNCH("nav_assemble");
  return VUS_OK;
}

extern "C" int vus_nav_border_solve(int n_nodes, const double* rhs, const double* Scb, const double* Sbb,
                                    const double* gb, double lambda, double* dc, double* db, void* stream) {
  VUS_REQUIRE(rhs && Scb && Sbb && gb && dc && db, "null buffer");
  VUS_REQUIRE(n_nodes >= 1, "n_nodes=%d", n_nodes);
  nav_border_kernel<<<1, 1024, 0, vus::as_stream(stream)>>>(n_nodes, rhs, Scb, Sbb, gb, lambda, dc, db);
  VUS_CHECK_LAUNCH("nav_border_solve");
  return VUS_OK;
}

extern "C" int vus_nav_eval_step(const vus_nav_factors* N, int n_poses, const double* poses, const double* vels,
                                 const double* bias, const double* dc, const double* db, const double* new_poses,
                                 double* new_vels, double* new_bias, double* out, double* work, void* stream) {
  if (int rc = check_nav(N, n_poses)) return rc;
  VUS_REQUIRE(poses && vels && bias && dc && db && new_poses && new_vels && new_bias && out && work, "null buffer");
  hipStream_t st = vus::as_stream(stream);
  nav_retract_kernel<<<cdivi(3ll * n_poses + 6, 256), 256, 0, st>>>(n_poses, vels, bias, dc, db, new_vels, new_bias);
  double* rec_imu = work;
  double* rec_dvl = rec_imu + (size_t)IMU_REC * N->n_imu;
  double* part = rec_dvl + (size_t)DVL_REC * N->n_dvl;
  if (int rc = nav_errors(N, n_poses, poses, vels, bias, dc, db, 2, rec_imu, rec_dvl, part, out, st)) return rc;
  return nav_errors(N, n_poses, new_poses, new_vels, new_bias, nullptr, nullptr, 1, rec_imu, rec_dvl, part, out + 1, st);
}

extern "C" int vus_nav_error(const vus_nav_factors* N, int n_poses, const double* poses, const double* vels,
                             const double* bias, double* err, double* work, void* stream) {
  if (int rc = check_nav(N, n_poses)) return rc;
  VUS_REQUIRE(poses && vels && bias && err && work, "null buffer");
  double* rec_imu = work;
  double* rec_dvl = rec_imu + (size_t)IMU_REC * N->n_imu;
  double* part = rec_dvl + (size_t)DVL_REC * N->n_dvl;
  return nav_errors(N, n_poses, poses, vels, bias, nullptr, nullptr, 1, rec_imu, rec_dvl, part, err, vus::as_stream(stream));
}

// ---- host side: IMU preintegration (gtsam does it in C++ inside integrateMeasurement; the Python restatement of it,
// gtsam/imu.py, cost the end-to-end sequence 78 of its 92 ms) ---------------------------------------------------------
namespace {
namespace pim {
constexpr int DT = 0, DR = 1, DP = 10, DV = 13, DR_DBG = 16, DP_DBA = 25, DP_DBG = 34, DV_DBA = 43, DV_DBG = 52, BIAS = 61, COV = 67;

inline void skew(const double* w, double* W) {
  W[0] = 0.0;   W[1] = -w[2]; W[2] = w[1];
  W[3] = w[2];  W[4] = 0.0;   W[5] = -w[0];
  W[6] = -w[1]; W[7] = w[0];  W[8] = 0.0;
}
inline void mul33(const double* A, const double* B, double* C) {
  for (int r = 0; r < 3; ++r)
    for (int c = 0; c < 3; ++c) C[3 * r + c] = A[3 * r] * B[c] + A[3 * r + 1] * B[3 + c] + A[3 * r + 2] * B[6 + c];
}
inline void mul33_tn(const double* A, const double* B, double* C) {      // A^T B
  for (int r = 0; r < 3; ++r)
    for (int c = 0; c < 3; ++c) C[3 * r + c] = A[r] * B[c] + A[3 + r] * B[3 + c] + A[6 + r] * B[6 + c];
}
// exp(w^) and the right Jacobian of SO(3), with the small-angle branches of gtsam/imu.py
inline void expmap_jr(const double* w, double* R, double* Jr) {
  const double th2 = w[0] * w[0] + w[1] * w[1] + w[2] * w[2];
  double W[9], W2[9];
  skew(w, W);
  mul33(W, W, W2);
  double s1, s2;
  if (th2 <= 2.220446049250313e-16) {
    s1 = 1.0;
    s2 = 0.0;
    for (int i = 0; i < 9; ++i) R[i] = (i % 4 == 0 ? 1.0 : 0.0) + W[i];
  } else {
    const double th = std::sqrt(th2), sh = std::sin(0.5 * th);
    s1 = std::sin(th) / th;
    s2 = 2.0 * sh * sh / th2;
    for (int i = 0; i < 9; ++i) R[i] = (i % 4 == 0 ? 1.0 : 0.0) + s1 * W[i] + s2 * W2[i];
  }
  double a, b;
  if (th2 < 1e-10) {
    a = 0.5 - th2 / 24.0;
    b = 1.0 / 6.0 - th2 / 120.0;
  } else {
    const double th = std::sqrt(th2);
    a = (1.0 - std::cos(th)) / th2;
    b = (th - std::sin(th)) / (th2 * th);
  }
  for (int i = 0; i < 9; ++i) Jr[i] = (i % 4 == 0 ? 1.0 : 0.0) - a * W[i] + b * W2[i];
}

void integrate(double* p, const double* smp, const double* acc_cov, const double* gyro_cov, const double* int_cov) {
  const double dt = smp[6];
  const double a[3] = {smp[0] - p[BIAS], smp[1] - p[BIAS + 1], smp[2] - p[BIAS + 2]};
  const double w[3] = {(smp[3] - p[BIAS + 3]) * dt, (smp[4] - p[BIAS + 4]) * dt, (smp[5] - p[BIAS + 5]) * dt};
  double dRinc[9], Jr[9], aX[9], RaX[9];
  expmap_jr(w, dRinc, Jr);
  skew(a, aX);
  double* dR = p + DR;
  mul33(dR, aX, RaX);
  // covariance: A cov A^T + B (acc_cov / dt) B^T + C (gyro_cov / dt) C^T, + int_cov dt on the position block
  double A[81] = {0.0}, Bm[27] = {0.0}, Cm[27] = {0.0};
  for (int i = 0; i < 9; ++i) A[10 * i] = 1.0;
  for (int r = 0; r < 3; ++r)
    for (int c = 0; c < 3; ++c) {
      A[9 * r + c] = dRinc[3 * c + r];
      A[9 * (3 + r) + c] = -0.5 * dt * dt * RaX[3 * r + c];
      A[9 * (6 + r) + c] = -dt * RaX[3 * r + c];
      Bm[3 * (3 + r) + c] = 0.5 * dt * dt * dR[3 * r + c];
      Bm[3 * (6 + r) + c] = dt * dR[3 * r + c];
      Cm[3 * r + c] = dt * Jr[3 * r + c];
    }
  for (int r = 0; r < 3; ++r) A[9 * (3 + r) + 6 + r] = dt;
  double* cov = p + COV;
  double T[81], N[81];
  for (int r = 0; r < 9; ++r)
    for (int c = 0; c < 9; ++c) {
      double v = 0.0;
      for (int k = 0; k < 9; ++k) v += A[9 * r + k] * cov[9 * k + c];
      T[9 * r + c] = v;
    }
  for (int r = 0; r < 9; ++r)
    for (int c = 0; c < 9; ++c) {
      double v = 0.0;
      for (int k = 0; k < 9; ++k) v += T[9 * r + k] * A[9 * c + k];
      N[9 * r + c] = v;
    }
  auto add_noise = [&](const double* G, const double* Q) {      // N += G (Q / dt) G^T, G 9x3
    double GQ[27];
    for (int r = 0; r < 9; ++r)
      for (int c = 0; c < 3; ++c) GQ[3 * r + c] = (G[3 * r] * Q[c] + G[3 * r + 1] * Q[3 + c] + G[3 * r + 2] * Q[6 + c]) / dt;
    for (int r = 0; r < 9; ++r)
      for (int c = 0; c < 9; ++c) N[9 * r + c] += GQ[3 * r] * G[3 * c] + GQ[3 * r + 1] * G[3 * c + 1] + GQ[3 * r + 2] * G[3 * c + 2];
  };
  add_noise(Bm, acc_cov);
  add_noise(Cm, gyro_cov);
  for (int r = 0; r < 3; ++r)
    for (int c = 0; c < 3; ++c) N[9 * (3 + r) + 3 + c] += int_cov[3 * r + c] * dt;
  std::memcpy(cov, N, sizeof(N));
  // bias Jacobians, then the deltas (every update uses the OLD dR, dV)
  double t[9], tmp[9];
  mul33(RaX, p + DR_DBG, t);
  for (int i = 0; i < 9; ++i) {
    p[DP_DBA + i] += p[DV_DBA + i] * dt - 0.5 * dt * dt * dR[i];
    p[DP_DBG + i] += p[DV_DBG + i] * dt - 0.5 * dt * dt * t[i];
    p[DV_DBA + i] -= dt * dR[i];
    p[DV_DBG + i] -= dt * t[i];
  }
  mul33_tn(dRinc, p + DR_DBG, tmp);
  for (int i = 0; i < 9; ++i) p[DR_DBG + i] = tmp[i] - dt * Jr[i];
  double Ra[3];
  for (int r = 0; r < 3; ++r) Ra[r] = dR[3 * r] * a[0] + dR[3 * r + 1] * a[1] + dR[3 * r + 2] * a[2];
  for (int r = 0; r < 3; ++r) {
    p[DP + r] += p[DV + r] * dt + 0.5 * dt * dt * Ra[r];
    p[DV + r] += dt * Ra[r];
  }
  mul33(dR, dRinc, tmp);
  std::memcpy(dR, tmp, sizeof(tmp));
  p[DT] += dt;
}

// W = L^-1, cov = L L^T
bool whitening(const double* cov, double* W) {
  double L[81] = {0.0};
  for (int c = 0; c < 9; ++c) {
    double d = cov[10 * c];
    for (int k = 0; k < c; ++k) d -= L[9 * c + k] * L[9 * c + k];
    if (!(d > 0.0)) return false;
    L[10 * c] = std::sqrt(d);
    for (int r = c + 1; r < 9; ++r) {
      double v = cov[9 * r + c];
      for (int k = 0; k < c; ++k) v -= L[9 * r + k] * L[9 * c + k];
      L[9 * r + c] = v / L[10 * c];
    }
  }
  for (int c = 0; c < 9; ++c)          // column c of L^-1 by forward substitution
    for (int r = 0; r < 9; ++r) {
      double v = (r == c) ? 1.0 : 0.0;
      for (int k = 0; k < r; ++k) v -= L[9 * r + k] * W[9 * k + c];
      W[9 * r + c] = v / L[10 * r];
    }
  return true;
}
}  // namespace pim
}  // namespace

extern "C" int vus_imu_preintegrate(double* pim_rec, const double* samples, int n, const double* acc_cov, const double* gyro_cov,
                                    const double* int_cov, double* whiten) {
  VUS_REQUIRE(pim_rec && acc_cov && gyro_cov && int_cov, "null buffer");
  VUS_REQUIRE(n >= 0 && (samples != nullptr || n == 0), "n=%d", n);
  for (int i = 0; i < n; ++i) {
    VUS_REQUIRE(samples[7 * i + 6] > 0.0, "sample %d: dt=%g is not positive", i, samples[7 * i + 6]);
    pim::integrate(pim_rec, samples + 7 * (size_t)i, acc_cov, gyro_cov, int_cov);
  }
  if (whiten) VUS_REQUIRE(pim::whitening(pim_rec + pim::COV, whiten), "preintegrated covariance is not positive definite");
  return VUS_OK;
}
