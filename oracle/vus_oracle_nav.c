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

/* ---------------------------------------------------------------------------------------------
 * Full-graph Levenberg-Marquardt with navigation factors (oracle only: dense camera-side solve).
 * Camera-side vector: for pose i [X_i (6), V_i (3)] at offset 9 i, then the shared bias (6).
 * Same LM control flow as vus_ba_lm_optimize_cpu (gtsam defaults). */
typedef struct vus_lm_params {
  double lambda_initial, lambda_factor, lambda_upper, lambda_lower, min_model_fidelity;
  double rel_tol, abs_tol, error_tol;
  int max_iterations;
} vus_lm_params;
#define VUS_LM_HIST 128
typedef struct vus_lm_report {
  int iterations, outer, tries, status;
  double initial_error, final_error, final_lambda;
  double err_hist[VUS_LM_HIST];
  double lambda_hist[VUS_LM_HIST];
} vus_lm_report;

int vus_ba_error_cpu(const vus_ba_problem* P, const double* poses, const double* points, double* err);
int vus_ba_linearize_cpu(const vus_ba_problem* P, const double* poses, const double* points, double* W, double* V,
                         double* gl, double* Hpp, double* gp, double* err);
void vus_pose_retract_cpu(const double* T, const double* xi, double* out);
void vus_stereo_factor_cpu(const double* T, const double* p, const double* m, const double* K, double w, double* r,
                           double* H1, double* H2);

/* error of the navigation factors; when H/g are given, also accumulate J^T J and J^T r (dense, ld = nc);
 * when d (camera-side step) is given, return the LINEARISED error 0.5 |r + J d|^2 instead. */
static double nav_terms(const vus_nav_factors* N, int nP, const double* poses, const double* vels, const double* bias,
                        double* H, double* g, const double* d) {
  const int nc = 9 * nP + 6;
  double e = 0;
  for (int f = 0; f < N->n_imu; ++f) {
    const int i = N->imu_i[f], j = N->imu_j[f];
    double r[9], J[9 * 24], rw[9], Jw[9 * 24];
    vus_imu_factor_cpu(poses + 12 * i, vels + 3 * i, poses + 12 * j, vels + 3 * j, bias, N->imu_pim + PIM_DOUBLES * (size_t)f,
                       N->gravity, r, (H || d) ? J : NULL);
    const double* W = N->imu_W + 81 * (size_t)f;
    for (int a = 0; a < 9; ++a) {
      double t = 0;
      for (int k = 0; k < 9; ++k) t += W[9 * a + k] * r[k];
      rw[a] = t;
    }
    int col[24];
    for (int k = 0; k < 9; ++k) { col[k] = 9 * i + k; col[9 + k] = 9 * j + k; }
    for (int k = 0; k < 6; ++k) col[18 + k] = 9 * nP + k;
    if (H || d) {
      for (int a = 0; a < 9; ++a)
        for (int c = 0; c < 24; ++c) {
          double t = 0;
          for (int k = 0; k < 9; ++k) t += W[9 * a + k] * J[24 * k + c];
          Jw[24 * a + c] = t;
        }
    }
    if (d)
      for (int a = 0; a < 9; ++a)
        for (int c = 0; c < 24; ++c) rw[a] += Jw[24 * a + c] * d[col[c]];
    for (int a = 0; a < 9; ++a) e += 0.5 * rw[a] * rw[a];
    if (H)
      for (int c1 = 0; c1 < 24; ++c1) {
        double t = 0;
        for (int a = 0; a < 9; ++a) t += Jw[24 * a + c1] * rw[a];
        g[col[c1]] += t;
        for (int c2 = 0; c2 < 24; ++c2) {
          double h = 0;
          for (int a = 0; a < 9; ++a) h += Jw[24 * a + c1] * Jw[24 * a + c2];
          H[(size_t)col[c1] * nc + col[c2]] += h;
        }
      }
  }
  for (int f = 0; f < N->n_dvl; ++f) {
    const int i = N->dvl_pose[f];
    const double w = N->dvl_w[f];
    double r[3], JX[18], Jv[9];
    vus_dvl_factor_cpu(poses + 12 * i, vels + 3 * i, N->dvl_meas + 3 * f, r, JX, Jv);
    double Jw[3 * 9];
    for (int a = 0; a < 3; ++a) {
      r[a] *= w;
      for (int c = 0; c < 6; ++c) Jw[9 * a + c] = w * JX[6 * a + c];
      for (int c = 0; c < 3; ++c) Jw[9 * a + 6 + c] = w * Jv[3 * a + c];
    }
    if (d)
      for (int a = 0; a < 3; ++a)
        for (int c = 0; c < 9; ++c) r[a] += Jw[9 * a + c] * d[9 * i + c];
    for (int a = 0; a < 3; ++a) e += 0.5 * r[a] * r[a];
    if (H)
      for (int c1 = 0; c1 < 9; ++c1) {
        double t = 0;
        for (int a = 0; a < 3; ++a) t += Jw[9 * a + c1] * r[a];
        g[9 * i + c1] += t;
        for (int c2 = 0; c2 < 9; ++c2) {
          double h = 0;
          for (int a = 0; a < 3; ++a) h += Jw[9 * a + c1] * Jw[9 * a + c2];
          H[(size_t)(9 * i + c1) * nc + 9 * i + c2] += h;
        }
      }
  }
  for (int f = 0; f < N->n_vprior; ++f) {
    const int i = N->vprior_idx[f];
    for (int k = 0; k < 3; ++k) {
      const double w = N->vprior_w[3 * f + k];
      double r = w * (vels[3 * i + k] - N->vprior_v[3 * f + k]);
      if (d) r += w * d[9 * i + 6 + k];
      e += 0.5 * r * r;
      if (H) { g[9 * i + 6 + k] += w * r; H[(size_t)(9 * i + 6 + k) * nc + 9 * i + 6 + k] += w * w; }
    }
  }
  return e;
}

static int dense_cholesky_solve(double* A, int n, double* b) { /* A overwritten; returns 0 ok */
  for (int c = 0; c < n; ++c) {
    double s = A[(size_t)c * n + c];
    for (int k = 0; k < c; ++k) s -= A[(size_t)c * n + k] * A[(size_t)c * n + k];
    if (!(s > 0.0)) return c + 1;
    double l = sqrt(s);
    A[(size_t)c * n + c] = l;
    for (int r = c + 1; r < n; ++r) {
      double t = A[(size_t)r * n + c];
      for (int k = 0; k < c; ++k) t -= A[(size_t)r * n + k] * A[(size_t)c * n + k];
      A[(size_t)r * n + c] = t / l;
    }
  }
  for (int r = 0; r < n; ++r) { double t = b[r]; for (int k = 0; k < r; ++k) t -= A[(size_t)r * n + k] * b[k]; b[r] = t / A[(size_t)r * n + r]; }
  for (int c = n - 1; c >= 0; --c) { double t = b[c]; for (int r = c + 1; r < n; ++r) t -= A[(size_t)r * n + c] * b[r]; b[c] = t / A[(size_t)c * n + c]; }
  return 0;
}

static void sym3_inv(const double* v, double* o) {
  double a = v[0], b = v[1], c = v[2], d = v[3], e = v[4], f = v[5];
  double c00 = d * f - e * e, c01 = c * e - b * f, c02 = b * e - c * d;
  double id = 1.0 / (a * c00 + b * c01 + c * c02);
  o[0] = c00 * id; o[1] = c01 * id; o[2] = c02 * id; o[3] = (a * f - c * c) * id; o[4] = (b * c - a * e) * id; o[5] = (a * d - b * b) * id;
}
static inline double s3(const double* v, int r, int c) { static const int ix[3][3] = {{0, 1, 2}, {1, 3, 4}, {2, 4, 5}}; return v[ix[r][c]]; }

static double nav_total_error(const vus_ba_problem* P, const vus_nav_factors* N, const double* poses, const double* vels,
                              const double* bias, const double* points) {
  double e;
  vus_ba_error_cpu(P, poses, points, &e);
  return e + nav_terms(N, P->n_poses, poses, vels, bias, NULL, NULL, NULL);
}

int vus_nav_total_error_cpu(const vus_ba_problem* P, const vus_nav_factors* N, const double* poses, const double* vels,
                            const double* bias, const double* points, double* err) {
  if (!P || !N || !poses || !vels || !bias || !points || !err) return VUS_E_INVALID;
  err[0] = nav_total_error(P, N, poses, vels, bias, points);
  return VUS_OK;
}

int vus_nav_lm_optimize_cpu(const vus_ba_problem* P, const vus_nav_factors* N, const vus_lm_params* prm, double* poses,
                            double* vels, double* bias, double* points, vus_lm_report* rep) {
  if (!P || !N || !prm || !poses || !vels || !bias || !points || !rep) return VUS_E_INVALID;
  const int nP = P->n_poses, nL = P->n_points, nO = P->n_obs, nc = 9 * nP + 6;
  double* W = malloc(sizeof(double) * 18 * (size_t)(nO + 1));
  double* Y = malloc(sizeof(double) * 18 * (size_t)(nO + 1));
  double* V = malloc(sizeof(double) * 6 * (size_t)(nL + 1));
  double* Vinv = malloc(sizeof(double) * 6 * (size_t)(nL + 1));
  double* gl = malloc(sizeof(double) * 3 * (size_t)(nL + 1));
  double* dl = malloc(sizeof(double) * 3 * (size_t)(nL + 1));
  double* Hpp = malloc(sizeof(double) * 36 * (size_t)nP);
  double* gp = malloc(sizeof(double) * 6 * (size_t)nP);
  double* H = malloc(sizeof(double) * (size_t)nc * nc);
  double* S = malloc(sizeof(double) * (size_t)nc * nc);
  double* g = malloc(sizeof(double) * nc);
  double* d = malloc(sizeof(double) * nc);
  double* nposes = malloc(sizeof(double) * 12 * (size_t)nP);
  double* nvels = malloc(sizeof(double) * 3 * (size_t)nP);
  double* npoints = malloc(sizeof(double) * 3 * (size_t)(nL + 1));
  double nbias[6];
  memset(rep, 0, sizeof *rep);
  double lambda = prm->lambda_initial;
  double current = nav_total_error(P, N, poses, vels, bias, points);
  rep->initial_error = current;
  rep->status = 1;
  while (rep->iterations < prm->max_iterations) {
    double lin_stereo;
    vus_ba_linearize_cpu(P, poses, points, W, V, gl, Hpp, gp, &lin_stereo);
    memset(H, 0, sizeof(double) * (size_t)nc * nc);
    memset(g, 0, sizeof(double) * nc);
    for (int i = 0; i < nP; ++i)
      for (int r = 0; r < 6; ++r) {
        g[9 * i + r] = gp[6 * i + r];
        for (int c = 0; c < 6; ++c) H[(size_t)(9 * i + r) * nc + 9 * i + c] = Hpp[36 * i + 6 * r + c];
      }
    const double lin0 = lin_stereo + nav_terms(N, nP, poses, vels, bias, H, g, NULL);
    double new_error = current;
    int stop_search = 0, accepted = 0;
    for (;;) {
      memcpy(S, H, sizeof(double) * (size_t)nc * nc);
      memcpy(d, g, sizeof(double) * nc);
      for (int k = 0; k < nc; ++k) S[(size_t)k * nc + k] += lambda;
      for (int j = 0; j < nL; ++j) {
        double Vd[6];
        for (int k = 0; k < 6; ++k) Vd[k] = V[6 * j + k];
        Vd[0] += lambda; Vd[3] += lambda; Vd[5] += lambda;
        sym3_inv(Vd, Vinv + 6 * j);
        const double* Vi = Vinv + 6 * j;
        for (int a = P->point_ptr[j]; a < P->point_ptr[j + 1]; ++a) {
          const double* Wa = W + 18 * (size_t)a;
          double* Ya = Y + 18 * (size_t)a;
          for (int rr = 0; rr < 6; ++rr)
            for (int c = 0; c < 3; ++c)
              Ya[3 * rr + c] = Wa[3 * rr] * s3(Vi, 0, c) + Wa[3 * rr + 1] * s3(Vi, 1, c) + Wa[3 * rr + 2] * s3(Vi, 2, c);
          const int ia = P->obs_pose[a];
          for (int rr = 0; rr < 6; ++rr)
            d[9 * ia + rr] -= Ya[3 * rr] * gl[3 * j] + Ya[3 * rr + 1] * gl[3 * j + 1] + Ya[3 * rr + 2] * gl[3 * j + 2];
        }
        for (int a = P->point_ptr[j]; a < P->point_ptr[j + 1]; ++a)
          for (int b = P->point_ptr[j]; b < P->point_ptr[j + 1]; ++b) {
            const int ia = P->obs_pose[a], ib = P->obs_pose[b];
            const double* Ya = Y + 18 * (size_t)a;
            const double* Wb = W + 18 * (size_t)b;
            for (int rr = 0; rr < 6; ++rr)
              for (int c = 0; c < 6; ++c)
                S[(size_t)(9 * ia + rr) * nc + 9 * ib + c] -= Ya[3 * rr] * Wb[3 * c] + Ya[3 * rr + 1] * Wb[3 * c + 1] + Ya[3 * rr + 2] * Wb[3 * c + 2];
          }
      }
      for (int k = 0; k < nc; ++k) d[k] = -d[k];
      const int status = dense_cholesky_solve(S, nc, d);
      ++rep->tries;
      int success = 0;
      if (status == 0) {
        for (int j = 0; j < nL; ++j) {
          double t[3] = {gl[3 * j], gl[3 * j + 1], gl[3 * j + 2]};
          for (int a = P->point_ptr[j]; a < P->point_ptr[j + 1]; ++a) {
            const double* Wa = W + 18 * (size_t)a;
            const double* dd = d + 9 * P->obs_pose[a];
            for (int c = 0; c < 3; ++c)
              for (int rr = 0; rr < 6; ++rr) t[c] += Wa[3 * rr + c] * dd[rr];
          }
          const double* Vi = Vinv + 6 * j;
          for (int c = 0; c < 3; ++c) dl[3 * j + c] = -(s3(Vi, c, 0) * t[0] + s3(Vi, c, 1) * t[1] + s3(Vi, c, 2) * t[2]);
        }
        /* linearised error at the step */
        double lin = nav_terms(N, nP, poses, vels, bias, NULL, NULL, d);
        for (int a = 0; a < nO; ++a) {
          const int i = P->obs_pose[a], j = P->obs_point[a];
          double r[3], H1[18], H2[9];
          vus_stereo_factor_cpu(poses + 12 * i, points + 3 * j, P->meas + 3 * a, P->K, P->inv_sigma, r, H1, H2);
          for (int rr = 0; rr < 3; ++rr) {
            double t = r[rr];
            for (int c = 0; c < 6; ++c) t += H1[6 * rr + c] * d[9 * i + c];
            for (int c = 0; c < 3; ++c) t += H2[3 * rr + c] * dl[3 * j + c];
            lin += 0.5 * t * t;
          }
        }
        for (int q = 0; q < P->n_priors; ++q) { /* pose priors: r + w d, with r from the linearisation */
          const int i = P->prior_pose[q];
          double xi[6];
          extern void vus_pose_local_cpu(const double*, const double*, double*);
          vus_pose_local_cpu(poses + 12 * i, P->prior_T + 12 * q, xi);
          for (int k = 0; k < 6; ++k) {
            const double w = P->prior_w[6 * q + k];
            const double t = -xi[k] * w + w * d[9 * i + k];
            lin += 0.5 * t * t;
          }
        }
        for (int i = 0; i < nP; ++i) {
          vus_pose_retract_cpu(poses + 12 * i, d + 9 * i, nposes + 12 * i);
          for (int k = 0; k < 3; ++k) nvels[3 * i + k] = vels[3 * i + k] + d[9 * i + 6 + k];
        }
        for (int k = 0; k < 6; ++k) nbias[k] = bias[k] + d[9 * nP + k];
        for (int k = 0; k < 3 * nL; ++k) npoints[k] = points[k] + dl[k];
        const double nerr = nav_total_error(P, N, nposes, nvels, nbias, npoints);
        const double lin_change = lin0 - lin;
        if (lin_change >= 0.0) {
          const double cost_change = current - nerr;
          if (lin_change > 2.220446049250313e-16 * lin0) success = cost_change / lin_change > prm->min_model_fidelity;
          if (fabs(cost_change) < prm->rel_tol * current) stop_search = 1;
          if (success) {
            memcpy(poses, nposes, sizeof(double) * 12 * (size_t)nP);
            memcpy(vels, nvels, sizeof(double) * 3 * (size_t)nP);
            memcpy(bias, nbias, sizeof nbias);
            memcpy(points, npoints, sizeof(double) * 3 * (size_t)nL);
            new_error = nerr;
          }
        }
      }
      if (success) { lambda /= prm->lambda_factor; if (lambda < prm->lambda_lower) lambda = prm->lambda_lower; accepted = 1; break; }
      if (stop_search) break;
      lambda *= prm->lambda_factor;
      if (lambda >= prm->lambda_upper) { rep->status = 2; break; }
    }
    if (rep->outer < VUS_LM_HIST) { rep->err_hist[rep->outer] = new_error; rep->lambda_hist[rep->outer] = lambda; }
    ++rep->outer;
    rep->iterations += accepted;
    int converged;
    if (new_error <= prm->error_tol) converged = 1;
    else { const double ad = current - new_error; converged = (ad / current <= prm->rel_tol) || (ad <= prm->abs_tol); }
    current = new_error;
    if (rep->status == 2) break;
    if (converged) { rep->status = 0; break; }
    if (!isfinite(current)) break;
  }
  rep->final_error = current;
  rep->final_lambda = lambda;
  free(W); free(Y); free(V); free(Vinv); free(gl); free(dl); free(Hpp); free(gp); free(H); free(S); free(g); free(d);
  free(nposes); free(nvels); free(npoints);
  return VUS_OK;
}


/* ---------------------------------------------------------------------------------------------
 * Twins of the navigation entry points of include/vus.h (node layout: node 2i = X(i), 2i+1 = V(i)
 * padded to 6, bias as a border) -- used to check the HIP kernels stage by stage. */
static void dense_to_nodes(int nP, const double* H, const double* g, double* Snav, double* Scb, double* Sbb,
                           double* gnav, double* gb) {
  const int nc = 9 * nP + 6, nn = 2 * nP;
  memset(Snav, 0, sizeof(double) * 36 * 4 * (size_t)nn);
  memset(Scb, 0, sizeof(double) * 36 * (size_t)nn);
  memset(gnav, 0, sizeof(double) * 6 * (size_t)nn);
  /* dense index -> (node, dim) */
  for (int r = 0; r < 9 * nP; ++r) {
    const int i = r / 9, k = r % 9, n1 = k < 6 ? 2 * i : 2 * i + 1, d1 = k < 6 ? k : k - 6;
    gnav[6 * n1 + d1] = g[r];
    for (int c = 0; c < 9 * nP; ++c) {
      const int j = c / 9, kk = c % 9, n2 = kk < 6 ? 2 * j : 2 * j + 1, d2 = kk < 6 ? kk : kk - 6;
      if (n1 >= n2 && n1 - n2 <= 3) Snav[36 * ((size_t)n1 * 4 + (n1 - n2)) + 6 * d1 + d2] = H[(size_t)r * nc + c];
    }
    for (int q = 0; q < 6; ++q) Scb[36 * (size_t)n1 + 6 * d1 + q] = H[(size_t)r * nc + 9 * nP + q];
  }
  for (int a = 0; a < 6; ++a) {
    gb[a] = g[9 * nP + a];
    for (int b = 0; b < 6; ++b) Sbb[6 * a + b] = H[(size_t)(9 * nP + a) * nc + 9 * nP + b];
  }
}

int vus_nav_linearize_cpu(const vus_nav_factors* N, int n_poses, const double* poses, const double* vels,
                          const double* bias, double* Snav, double* Scb, double* Sbb, double* gnav, double* gb,
                          double* err, double* work) {
  (void)work;
  if (!N || !poses || !vels || !bias || !Snav || !Scb || !Sbb || !gnav || !gb || !err) return VUS_E_INVALID;
  const int nc = 9 * n_poses + 6;
  double* H = calloc((size_t)nc * nc, sizeof(double));
  double* g = calloc(nc, sizeof(double));
  err[0] = nav_terms(N, n_poses, poses, vels, bias, H, g, NULL);
  dense_to_nodes(n_poses, H, g, Snav, Scb, Sbb, gnav, gb);
  free(H); free(g);
  return VUS_OK;
}

int vus_nav_assemble_cpu(int n_nodes, int band, double lambda, const double* Snav, const double* Scb,
                         const double* gnav, double* Sband, double* gs, double* rhs) {
  if (!Snav || !Scb || !gnav || !Sband || !gs || !rhs || n_nodes < 2 || band < 1) return VUS_E_INVALID;
  const int smax = band < 3 ? band : 3;
  for (int node = 0; node < n_nodes; ++node) {
    for (int s = 0; s <= smax; ++s)
      for (int e = 0; e < 36; ++e) {
        double v = Snav[36 * ((size_t)node * 4 + s) + e];
        if (s == 0 && (node & 1) && e % 7 == 0) v += (e / 7 < 3) ? lambda : 1.0;
        Sband[36 * ((size_t)node * (band + 1) + s) + e] += v;
      }
    for (int d = 0; d < 6; ++d) {
      const size_t k = 6 * (size_t)node + d;
      gs[k] += gnav[k];
      rhs[k] = -gs[k];
      for (int q = 0; q < 6; ++q) rhs[(size_t)(1 + q) * 6 * n_nodes + k] = Scb[36 * (size_t)node + 6 * d + q];
    }
  }
  return VUS_OK;
}

int vus_nav_border_solve_cpu(int n_nodes, const double* rhs, const double* Scb, const double* Sbb, const double* gb,
                             double lambda, double* dc, double* db) {
  if (!rhs || !Scb || !Sbb || !gb || !dc || !db || n_nodes < 1) return VUS_E_INVALID;
  const size_t n = 6 * (size_t)n_nodes;
  double M[36], v[6];
  for (int a = 0; a < 6; ++a) {
    v[a] = -gb[a];
    for (int b = 0; b < 6; ++b) M[6 * a + b] = Sbb[6 * a + b] + (a == b ? lambda : 0.0);
  }
  for (size_t k = 0; k < n; ++k) {
    const double* c = Scb + 36 * (k / 6) + 6 * (k % 6);
    for (int a = 0; a < 6; ++a) {
      v[a] -= c[a] * rhs[k];
      for (int b = 0; b < 6; ++b) M[6 * a + b] -= c[a] * rhs[(size_t)(1 + b) * n + k];
    }
  }
  if (dense_cholesky_solve(M, 6, v)) return VUS_E_INVALID;
  for (int k = 0; k < 6; ++k) db[k] = v[k];
  for (size_t k = 0; k < n; ++k) {
    double t = rhs[k];
    for (int b = 0; b < 6; ++b) t -= rhs[(size_t)(1 + b) * n + k] * v[b];
    dc[k] = t;
  }
  return VUS_OK;
}

static void nodes_to_dense_step(int nP, const double* dc, const double* db, double* d) {
  for (int i = 0; i < nP; ++i) {
    for (int k = 0; k < 6; ++k) d[9 * i + k] = dc[6 * (2 * i) + k];
    for (int k = 0; k < 3; ++k) d[9 * i + 6 + k] = dc[6 * (2 * i + 1) + k];
  }
  for (int k = 0; k < 6; ++k) d[9 * nP + k] = db[k];
}

int vus_nav_eval_step_cpu(const vus_nav_factors* N, int n_poses, const double* poses, const double* vels,
                          const double* bias, const double* dc, const double* db, const double* new_poses,
                          double* new_vels, double* new_bias, double* out, double* work) {
  (void)work;
  if (!N || !poses || !vels || !bias || !dc || !db || !new_poses || !new_vels || !new_bias || !out) return VUS_E_INVALID;
  double* d = malloc(sizeof(double) * (9 * (size_t)n_poses + 6));
  nodes_to_dense_step(n_poses, dc, db, d);
  for (int t = 0; t < 3 * n_poses; ++t) new_vels[t] = vels[t] + dc[6 * (2 * (t / 3) + 1) + t % 3];
  for (int k = 0; k < 6; ++k) new_bias[k] = bias[k] + db[k];
  out[0] = nav_terms(N, n_poses, poses, vels, bias, NULL, NULL, d);
  out[1] = nav_terms(N, n_poses, new_poses, new_vels, new_bias, NULL, NULL, NULL);
  free(d);
  return VUS_OK;
}

int vus_nav_error_cpu(const vus_nav_factors* N, int n_poses, const double* poses, const double* vels,
                      const double* bias, double* err, double* work) {
  (void)work;
  if (!N || !poses || !vels || !bias || !err) return VUS_E_INVALID;
  err[0] = nav_terms(N, n_poses, poses, vels, bias, NULL, NULL, NULL);
  return VUS_OK;
}
