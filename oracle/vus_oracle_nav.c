/* vus_oracle_nav.c -- CPU ORACLE (test infrastructure, NOT the product).
 *
 * Inertial / velocity factors on the camera side of the reference's graph (SURVEY.md section 8, rows
 * f1 and f2):
 *   gtsam.PreintegratedImuMeasurements + gtsam.ImuFactor      reference batch.py:90-92,178-193,237-239,289-293
 *   the DVL velocity factor                                   reference batch.py:196-250
 *   gtsam.PriorFactorVector on V(0)                           reference batch.py:282
 *
 * PARITY UNPINNED: GTSAM is un-vendored and unpinned (reference README.md:18,21).  Restated from the
 * published algorithm: Forster, Carlone, Dellaert, Scaramuzza, "On-Manifold Preintegration for Real-Time
 * Visual-Inertial Odometry", TRO 2017 -- the formulation behind gtsam::ManifoldPreintegration /
 * PreintegratedImuMeasurements::integrateMeasurement / ImuFactor::evaluateError (error = NavState_j
 * .localCoordinates(predict(NavState_i, bias)), tangent order (dR, dP, dV), first-order bias
 * correction with the stored Jacobians).  Pinned by finite-difference Jacobian tests and by an exact
 * constant-acceleration / constant-rate trajectory check in tests/.
 *
 * The DVL factor: the reference's Python callback (batch.py:196-233) returns the residual
 * e = R_i m - v_i with Jacobians that are ill-formed for a Pose3 key (3x3 for a 6-dof variable,
 * SURVEY.md D7).  The residual is kept; the Jacobians are the correct analytic ones:
 * de/dv = -I, de/dX = [ -R [m]x , 0 ].
 */
#define _GNU_SOURCE
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include "../include/vus.h"

/* ---- small helpers ---------------------------------------------------------------------------- */
static void skew(const double* w, double* S) {
  S[0] = 0; S[1] = -w[2]; S[2] = w[1];
  S[3] = w[2]; S[4] = 0; S[5] = -w[0];
  S[6] = -w[1]; S[7] = w[0]; S[8] = 0;
}
static void mat3_mul(const double* A, const double* B, double* C) {
  for (int r = 0; r < 3; ++r)
    for (int c = 0; c < 3; ++c) C[3 * r + c] = A[3 * r] * B[c] + A[3 * r + 1] * B[3 + c] + A[3 * r + 2] * B[6 + c];
}
static void mat3_tmul(const double* A, const double* B, double* C) { /* A^T B */
  for (int r = 0; r < 3; ++r)
    for (int c = 0; c < 3; ++c) C[3 * r + c] = A[r] * B[c] + A[3 + r] * B[3 + c] + A[6 + r] * B[6 + c];
}
static void mat3_vec(const double* A, const double* v, double* o) {
  for (int r = 0; r < 3; ++r) o[r] = A[3 * r] * v[0] + A[3 * r + 1] * v[1] + A[3 * r + 2] * v[2];
}
static void mat3_tvec(const double* A, const double* v, double* o) {
  for (int r = 0; r < 3; ++r) o[r] = A[r] * v[0] + A[3 + r] * v[1] + A[6 + r] * v[2];
}

void vus_so3_expmap_cpu(const double* w, double* R) {
  double th2 = w[0] * w[0] + w[1] * w[1] + w[2] * w[2];
  double W[9], WW[9];
  skew(w, W);
  if (th2 <= 2.220446049250313e-16) {
    for (int i = 0; i < 9; ++i) R[i] = W[i] + (i % 4 == 0 ? 1.0 : 0.0);
    return;
  }
  double th = sqrt(th2), s = sin(th) / th, sh = sin(0.5 * th), c = 2.0 * sh * sh / th2;
  mat3_mul(W, W, WW);
  for (int i = 0; i < 9; ++i) R[i] = (i % 4 == 0 ? 1.0 : 0.0) + s * W[i] + c * WW[i];
}

void vus_so3_logmap_cpu(const double* R, double* w) {
  double tr = R[0] + R[4] + R[8];
  if (tr + 1.0 < 1e-10) {
    if (fabs(R[8] + 1.0) > 1e-5) { double k = M_PI / sqrt(2.0 + 2.0 * R[8]); w[0] = k * R[2]; w[1] = k * R[5]; w[2] = k * (1.0 + R[8]); }
    else if (fabs(R[4] + 1.0) > 1e-5) { double k = M_PI / sqrt(2.0 + 2.0 * R[4]); w[0] = k * R[1]; w[1] = k * (1.0 + R[4]); w[2] = k * R[7]; }
    else { double k = M_PI / sqrt(2.0 + 2.0 * R[0]); w[0] = k * (1.0 + R[0]); w[1] = k * R[3]; w[2] = k * R[6]; }
    return;
  }
  double mag, tr3 = tr - 3.0;
  if (tr3 < -1e-7) { double th = acos((tr - 1.0) / 2.0); mag = th / (2.0 * sin(th)); }
  else mag = 0.5 - tr3 / 12.0;
  w[0] = mag * (R[7] - R[5]); w[1] = mag * (R[2] - R[6]); w[2] = mag * (R[3] - R[1]);
}

/* right Jacobian of SO(3) and its inverse (Forster 2017, eq. 8 and Chirikjian) */
void vus_so3_jr_cpu(const double* w, double* J) {
  double th2 = w[0] * w[0] + w[1] * w[1] + w[2] * w[2];
  double W[9], WW[9];
  skew(w, W);
  mat3_mul(W, W, WW);
  double a, b;
  if (th2 < 1e-10) { a = 0.5 - th2 / 24.0; b = 1.0 / 6.0 - th2 / 120.0; }
  else { double th = sqrt(th2); a = (1.0 - cos(th)) / th2; b = (th - sin(th)) / (th2 * th); }
  for (int i = 0; i < 9; ++i) J[i] = (i % 4 == 0 ? 1.0 : 0.0) - a * W[i] + b * WW[i];
}
void vus_so3_jr_inv_cpu(const double* w, double* J) {
  double th2 = w[0] * w[0] + w[1] * w[1] + w[2] * w[2];
  double W[9], WW[9];
  skew(w, W);
  mat3_mul(W, W, WW);
  double b;
  if (th2 < 1e-10) b = 1.0 / 12.0 + th2 / 720.0;
  else { double th = sqrt(th2); b = 1.0 / th2 - (1.0 + cos(th)) / (2.0 * th * sin(th)); }
  for (int i = 0; i < 9; ++i) J[i] = (i % 4 == 0 ? 1.0 : 0.0) + 0.5 * W[i] + b * WW[i];
}

/* ---- preintegration ---------------------------------------------------------------------------
 * pim layout (PIM_DOUBLES = 148 doubles): offsets below.  dR row-major 3x3; the five bias Jacobians
 * 3x3 row-major; bias_hat = (acc, gyro); cov 9x9 row-major in tangent order (theta, p, v). */
#define PIM_DT 0
#define PIM_DR 1
#define PIM_DP 10
#define PIM_DV 13
#define PIM_DR_DBG 16
#define PIM_DP_DBA 25
#define PIM_DP_DBG 34
#define PIM_DV_DBA 43
#define PIM_DV_DBG 52
#define PIM_BIAS 61   /* acc(3), gyro(3) */
#define PIM_COV 67    /* 9x9 covariance, tangent order (theta, p, v) */
#define PIM_DOUBLES 148

int vus_pim_doubles_cpu(void) { return PIM_DOUBLES; }

/* samples [n,7] = (ax, ay, az, wx, wy, wz, dt); bias_hat [6] = (acc, gyro); covariances 3x3 row-major. */
int vus_imu_preintegrate_cpu(const double* samples, int n, const double* bias_hat, const double* acc_cov,
                             const double* gyro_cov, const double* int_cov, double* pim) {
  if (!samples || !bias_hat || !acc_cov || !gyro_cov || !int_cov || !pim || n < 0) return VUS_E_INVALID;
  memset(pim, 0, sizeof(double) * PIM_DOUBLES);
  double* dR = pim + PIM_DR;
  dR[0] = dR[4] = dR[8] = 1.0;
  for (int k = 0; k < 6; ++k) pim[PIM_BIAS + k] = bias_hat[k];
  double* cov = pim + PIM_COV;
  for (int s = 0; s < n; ++s) {
    const double* m = samples + 7 * s;
    const double dt = m[6];
    double acc[3] = {m[0] - bias_hat[0], m[1] - bias_hat[1], m[2] - bias_hat[2]};
    double om[3] = {m[3] - bias_hat[3], m[4] - bias_hat[4], m[5] - bias_hat[5]};
    double th[3] = {om[0] * dt, om[1] * dt, om[2] * dt};
    double dRinc[9], Jr[9], accx[9], Racc[3], RaccX[9], tmp[9], tmp2[9];
    vus_so3_expmap_cpu(th, dRinc);
    vus_so3_jr_cpu(th, Jr);
    skew(acc, accx);
    mat3_vec(dR, acc, Racc);
    mat3_mul(dR, accx, RaccX);                 /* dR [acc]x */
    /* covariance first (uses the state before the update) */
    double A[81], B[27], C[27];
    memset(A, 0, sizeof A); memset(B, 0, sizeof B); memset(C, 0, sizeof C);
    for (int i = 0; i < 9; ++i) A[10 * i] = 1.0;
    for (int r = 0; r < 3; ++r)
      for (int c = 0; c < 3; ++c) {
        A[9 * r + c] = dRinc[3 * c + r];                              /* theta-theta: dRinc^T */
        A[9 * (3 + r) + c] = -0.5 * dt * dt * RaccX[3 * r + c];       /* p wrt theta */
        A[9 * (6 + r) + c] = -dt * RaccX[3 * r + c];                  /* v wrt theta */
        B[3 * (3 + r) + c] = 0.5 * dt * dt * dR[3 * r + c];           /* p wrt acc */
        B[3 * (6 + r) + c] = dt * dR[3 * r + c];                      /* v wrt acc */
        C[3 * r + c] = dt * Jr[3 * r + c];                            /* theta wrt omega */
      }
    for (int r = 0; r < 3; ++r) A[9 * (3 + r) + 6 + r] = dt;          /* p wrt v */
    double AC[81], N[81];
    for (int r = 0; r < 9; ++r)
      for (int c = 0; c < 9; ++c) { double t = 0; for (int k = 0; k < 9; ++k) t += A[9 * r + k] * cov[9 * k + c]; AC[9 * r + c] = t; }
    for (int r = 0; r < 9; ++r)
      for (int c = 0; c < 9; ++c) { double t = 0; for (int k = 0; k < 9; ++k) t += AC[9 * r + k] * A[9 * c + k]; N[9 * r + c] = t; }
    for (int r = 0; r < 9; ++r)
      for (int c = 0; c < 9; ++c) {
        double t = 0;
        for (int a = 0; a < 3; ++a)
          for (int b = 0; b < 3; ++b)
            t += B[3 * r + a] * (acc_cov[3 * a + b] / dt) * B[3 * c + b] + C[3 * r + a] * (gyro_cov[3 * a + b] / dt) * C[3 * c + b];
        N[9 * r + c] += t;
      }
    for (int r = 0; r < 3; ++r)
      for (int c = 0; c < 3; ++c) N[9 * (3 + r) + 3 + c] += int_cov[3 * r + c] * dt;
    memcpy(cov, N, sizeof N);
    /* bias Jacobians (old dR, old dR_dbg) */
    double* dR_dbg = pim + PIM_DR_DBG;
    mat3_mul(RaccX, dR_dbg, tmp);               /* dR [acc]x dR_dbg */
    for (int i = 0; i < 9; ++i) {
      pim[PIM_DP_DBA + i] += pim[PIM_DV_DBA + i] * dt - 0.5 * dt * dt * dR[i];
      pim[PIM_DP_DBG + i] += pim[PIM_DV_DBG + i] * dt - 0.5 * dt * dt * tmp[i];
      pim[PIM_DV_DBA + i] += -dt * dR[i];
      pim[PIM_DV_DBG + i] += -dt * tmp[i];
    }
    mat3_tmul(dRinc, dR_dbg, tmp2);             /* dRinc^T dR_dbg */
    for (int i = 0; i < 9; ++i) dR_dbg[i] = tmp2[i] - dt * Jr[i];
    /* state */
    for (int r = 0; r < 3; ++r) {
      pim[PIM_DP + r] += pim[PIM_DV + r] * dt + 0.5 * dt * dt * Racc[r];
      pim[PIM_DV + r] += dt * Racc[r];
    }
    mat3_mul(dR, dRinc, tmp);
    memcpy(dR, tmp, sizeof tmp);
    pim[PIM_DT] += dt;
  }
  return VUS_OK;
}

/* W = L^-1 with cov = L L^T (lower Cholesky): whitened residual = W r, |W r|^2 = r^T cov^-1 r. */
int vus_sqrt_information_cpu(const double* cov, int n, double* W) {
  if (!cov || !W || n < 1 || n > 16) return VUS_E_INVALID;
  double L[256];
  memset(L, 0, sizeof L);
  for (int c = 0; c < n; ++c) {
    double s = cov[n * c + c];
    for (int k = 0; k < c; ++k) s -= L[n * c + k] * L[n * c + k];
    if (!(s > 0.0)) return VUS_E_INVALID;
    L[n * c + c] = sqrt(s);
    for (int r = c + 1; r < n; ++r) {
      double t = cov[n * r + c];
      for (int k = 0; k < c; ++k) t -= L[n * r + k] * L[n * c + k];
      L[n * r + c] = t / L[n * c + c];
    }
  }
  memset(W, 0, sizeof(double) * n * n);
  for (int j = 0; j < n; ++j) /* forward-substitute the columns of the identity */
    for (int r = j; r < n; ++r) {
      double t = (r == j) ? 1.0 : 0.0;
      for (int k = j; k < r; ++k) t -= L[n * r + k] * W[n * k + j];
      W[n * r + j] = t / L[n * r + r];
    }
  return VUS_OK;
}

/* ---- ImuFactor ---------------------------------------------------------------------------------
 * Unwhitened residual r[9] (theta, p, v) and Jacobian J[9 x 24] (row-major), columns:
 *   0..5 pose_i (omega, u: right perturbation T Exp(xi))   6..8 vel_i   9..14 pose_j   15..17 vel_j
 *   18..23 bias (acc, gyro).   J may be NULL.  g[3] = gravity in the navigation frame. */
void vus_imu_factor_cpu(const double* Ti, const double* vi, const double* Tj, const double* vj, const double* bias,
                        const double* pim, const double* g, double* r, double* J) {
  const double dt = pim[PIM_DT];
  const double* dR = pim + PIM_DR;
  double dba[3], dbg[3];
  for (int k = 0; k < 3; ++k) { dba[k] = bias[k] - pim[PIM_BIAS + k]; dbg[k] = bias[3 + k] - pim[PIM_BIAS + 3 + k]; }
  /* first-order bias correction */
  double phi[3], Ephi[9], dRc[9], dPc[3], dVc[3], t3[3], t3b[3];
  mat3_vec(pim + PIM_DR_DBG, dbg, phi);
  vus_so3_expmap_cpu(phi, Ephi);
  mat3_mul(dR, Ephi, dRc);
  mat3_vec(pim + PIM_DP_DBA, dba, t3); mat3_vec(pim + PIM_DP_DBG, dbg, t3b);
  for (int k = 0; k < 3; ++k) dPc[k] = pim[PIM_DP + k] + t3[k] + t3b[k];
  mat3_vec(pim + PIM_DV_DBA, dba, t3); mat3_vec(pim + PIM_DV_DBG, dbg, t3b);
  for (int k = 0; k < 3; ++k) dVc[k] = pim[PIM_DV + k] + t3[k] + t3b[k];
  const double* Ri = Ti; const double* pi = Ti + 9;
  const double* Rj = Tj; const double* pj = Tj + 9;
  /* rotation residual */
  double RjtRi[9], E[9], rR[3];
  mat3_tmul(Rj, Ri, RjtRi);
  mat3_mul(RjtRi, dRc, E);
  vus_so3_logmap_cpu(E, rR);
  /* position / velocity residuals */
  double RidP[3], RidV[3], dpw[3], dvw[3], rP[3], rV[3];
  mat3_vec(Ri, dPc, RidP); mat3_vec(Ri, dVc, RidV);
  for (int k = 0; k < 3; ++k) {
    dpw[k] = pi[k] + vi[k] * dt + 0.5 * g[k] * dt * dt + RidP[k] - pj[k];
    dvw[k] = vi[k] + g[k] * dt + RidV[k] - vj[k];
  }
  mat3_tvec(Rj, dpw, rP); mat3_tvec(Rj, dvw, rV);
  for (int k = 0; k < 3; ++k) { r[k] = rR[k]; r[3 + k] = rP[k]; r[6 + k] = rV[k]; }
  if (!J) return;
  memset(J, 0, sizeof(double) * 9 * 24);
  double JrInv[9], JrInvNeg[9], nrR[3] = {-rR[0], -rR[1], -rR[2]}, M[9], M2[9], X[9];
  vus_so3_jr_inv_cpu(rR, JrInv);
  vus_so3_jr_inv_cpu(nrR, JrInvNeg);
#define JSET(row0, col0, Mat, sgn) for (int a = 0; a < 3; ++a) for (int b = 0; b < 3; ++b) J[24 * ((row0) + a) + (col0) + b] = (sgn) * (Mat)[3 * a + b]
  /* d rR */
  for (int a = 0; a < 3; ++a) for (int b = 0; b < 3; ++b) M[3 * a + b] = dRc[3 * b + a];   /* dRc^T */
  mat3_mul(JrInv, M, M2);
  JSET(0, 0, M2, 1.0);                                    /* omega_i */
  JSET(0, 9, JrInvNeg, -1.0);                             /* omega_j */
  double JrPhi[9];
  vus_so3_jr_cpu(phi, JrPhi);
  mat3_mul(JrInv, JrPhi, M); mat3_mul(M, pim + PIM_DR_DBG, M2);
  JSET(0, 21, M2, 1.0);                                   /* bias gyro */
  /* d rP */
  skew(dPc, X); mat3_mul(RjtRi, X, M);
  JSET(3, 0, M, -1.0);                                    /* omega_i: -Rj^T Ri [dPc]x */
  JSET(3, 3, RjtRi, 1.0);                                 /* u_i */
  for (int a = 0; a < 3; ++a) for (int b = 0; b < 3; ++b) M[3 * a + b] = Rj[3 * b + a] * dt;   /* Rj^T dt */
  JSET(3, 6, M, 1.0);                                     /* v_i */
  skew(rP, X);
  JSET(3, 9, X, 1.0);                                     /* omega_j: [rP]x */
  for (int a = 0; a < 3; ++a) J[24 * (3 + a) + 12 + a] = -1.0;   /* u_j */
  mat3_mul(RjtRi, pim + PIM_DP_DBA, M); JSET(3, 18, M, 1.0);
  mat3_mul(RjtRi, pim + PIM_DP_DBG, M); JSET(3, 21, M, 1.0);
  /* d rV */
  skew(dVc, X); mat3_mul(RjtRi, X, M);
  JSET(6, 0, M, -1.0);                                    /* omega_i */
  for (int a = 0; a < 3; ++a) for (int b = 0; b < 3; ++b) M[3 * a + b] = Rj[3 * b + a];         /* Rj^T */
  JSET(6, 6, M, 1.0);                                     /* v_i */
  skew(rV, X);
  JSET(6, 9, X, 1.0);                                     /* omega_j */
  JSET(6, 15, M, -1.0);                                   /* v_j */
  mat3_mul(RjtRi, pim + PIM_DV_DBA, M); JSET(6, 18, M, 1.0);
  mat3_mul(RjtRi, pim + PIM_DV_DBG, M); JSET(6, 21, M, 1.0);
#undef JSET
}

/* DVL velocity factor (batch.py:196-233 residual): e = R m - v; Jv [3x3] = -I, JX [3x6] = [-R [m]x, 0]. */
void vus_dvl_factor_cpu(const double* T, const double* v, const double* m, double* e, double* JX, double* Jv) {
  double Rm[3];
  mat3_vec(T, m, Rm);
  for (int k = 0; k < 3; ++k) e[k] = Rm[k] - v[k];
  if (JX) {
    double X[9], M[9];
    skew(m, X);
    mat3_mul(T, X, M);
    for (int a = 0; a < 3; ++a)
      for (int b = 0; b < 3; ++b) { JX[6 * a + b] = -M[3 * a + b]; JX[6 * a + 3 + b] = 0.0; }
  }
  if (Jv) { memset(Jv, 0, sizeof(double) * 9); Jv[0] = Jv[4] = Jv[8] = -1.0; }
}
