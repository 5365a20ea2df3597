/* vus_oracle_ba_mt.c -- CPU "PORT" of the stereo bundle adjustment for the cpu_baseline leg of bench.py
 * (TEST INFRASTRUCTURE, NOT the product; same rule as the rest of oracle/).
 *
 * The scalar oracle (vus_oracle_ba.c) is written for clarity and checks the HIP kernels; at BASELINE.json
 * configs[2] (2000 keyframes / 50k landmarks / 1.93 M stereo factors) it would be a strawman baseline: one
 * thread, a scalar band Cholesky with div/mod addressing.  BASELINE.md section 2 asks for the CPU path "at 1 thread
 * and at all cores (OpenMP over ... factors for M2)".  This file is that: the same arithmetic as the oracle
 * (tests/test_ba_oracle.py compares them), organised the way a CPU would run it --
 *   linearise      OpenMP over landmarks (W, V, gl, error) and over keyframes (Hpp, gp): no write is shared;
 *   Schur          OpenMP over the non-zero 6x6 blocks of the reduced camera system (vus_ba_structure), each
 *                  block summed in registers and written straight into LAPACK's lower band storage;
 *   reduced solve  LAPACK dpbtrf / dpbtrs (scipy.linalg.cholesky_banded on the multi-threaded OpenBLAS shipped
 *                  with scipy), driven from oracle/ba_port.py;
 *   back-substitution, step evaluation   OpenMP over landmarks.
 * What gtsam.LevenbergMarquardtOptimizer (reference batch.py:337) would do differently -- multifrontal Cholesky
 * under COLAMD instead of an explicit Schur complement -- solves the same linear system.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include "vus_oracle.h"

void vus_stereo_factor_cpu(const double* T, const double* p, const double* m, const double* K, double w,
                           double* r, double* H1, double* H2);
void vus_pose_retract_cpu(const double* T, const double* xi, double* out);
void vus_pose_local_cpu(const double* T, const double* T2, double* xi);

static void sym3_inv(const double* v, double lambda, double* o) {
  const double a = v[0] + lambda, b = v[1], c = v[2], d = v[3] + lambda, e = v[4], f = v[5] + lambda;
  const double c00 = d * f - e * e, c01 = c * e - b * f, c02 = b * e - c * d;
  const double id = 1.0 / (a * c00 + b * c01 + c * c02);
  o[0] = c00 * id; o[1] = c01 * id; o[2] = c02 * id;
  o[3] = (a * f - c * c) * id; o[4] = (b * c - a * e) * id; o[5] = (a * d - b * b) * id;
}

static double prior_terms(const vus_ba_problem* P, const double* poses, const double* dp, double* Hpp, double* gp) {
  double e = 0;
  for (int q = 0; q < P->n_priors; ++q) {
    const int i = P->prior_pose[q];
    double xi[6];
    vus_pose_local_cpu(poses + 12 * (size_t)i, P->prior_T + 12 * (size_t)q, xi);
    for (int k = 0; k < 6; ++k) {
      const double w = P->prior_w[6 * q + k];
      double r = -xi[k] * w;
      if (Hpp) { Hpp[36 * (size_t)i + 7 * k] += w * w; gp[6 * (size_t)i + k] += w * r; }
      if (dp) r += w * dp[6 * (size_t)i + k];
      e += 0.5 * r * r;
    }
  }
  return e;
}

int vus_ba_error_mt_cpu(const vus_ba_problem* P, const double* poses, const double* points, double* err) {
  if (!P || !poses || !points || !err) return VUS_E_INVALID;
  double e = 0;
#pragma omp parallel for reduction(+ : e) schedule(static)
  for (int a = 0; a < P->n_obs; ++a) {
    double r[3];
    vus_stereo_factor_cpu(poses + 12 * (size_t)P->obs_pose[a], points + 3 * (size_t)P->obs_point[a], P->meas + 3 * (size_t)a,
                          P->K, P->inv_sigma, r, NULL, NULL);
    e += 0.5 * (r[0] * r[0] + r[1] * r[1] + r[2] * r[2]);
  }
  err[0] = e + prior_terms(P, poses, NULL, NULL, NULL);
  return VUS_OK;
}

int vus_ba_linearize_mt_cpu(const vus_ba_problem* P, const double* poses, const double* points, double* W, double* V,
                            double* gl, double* Hpp, double* gp, double* err) {
  if (!P || !poses || !points || !W || !V || !gl || !Hpp || !gp || !err) return VUS_E_INVALID;
  double e = 0;
#pragma omp parallel for reduction(+ : e) schedule(dynamic, 64)
  for (int j = 0; j < P->n_points; ++j) {
    double v[6] = {0, 0, 0, 0, 0, 0}, g[3] = {0, 0, 0};
    for (int a = P->point_ptr[j]; a < P->point_ptr[j + 1]; ++a) {
      double r[3], H1[18], H2[9];
      vus_stereo_factor_cpu(poses + 12 * (size_t)P->obs_pose[a], points + 3 * (size_t)j, P->meas + 3 * (size_t)a, P->K,
                            P->inv_sigma, r, H1, H2);
      e += 0.5 * (r[0] * r[0] + r[1] * r[1] + r[2] * r[2]);
      double* Wa = W + 18 * (size_t)P->obs_ppos[a];
      for (int rr = 0; rr < 6; ++rr)
        for (int c = 0; c < 3; ++c) Wa[3 * rr + c] = H1[rr] * H2[c] + H1[6 + rr] * H2[3 + c] + H1[12 + rr] * H2[6 + c];
      int u = 0;
      for (int rr = 0; rr < 3; ++rr)
        for (int c = rr; c < 3; ++c, ++u) v[u] += H2[rr] * H2[c] + H2[3 + rr] * H2[3 + c] + H2[6 + rr] * H2[6 + c];
      for (int c = 0; c < 3; ++c) g[c] += H2[c] * r[0] + H2[3 + c] * r[1] + H2[6 + c] * r[2];
    }
    for (int k = 0; k < 6; ++k) V[6 * (size_t)j + k] = v[k];
    for (int k = 0; k < 3; ++k) gl[3 * (size_t)j + k] = g[k];
  }
#pragma omp parallel for schedule(dynamic, 4)
  for (int i = 0; i < P->n_poses; ++i) {
    double h[36], g[6];
    memset(h, 0, sizeof h);
    memset(g, 0, sizeof g);
    for (int s = P->pose_ptr[i]; s < P->pose_ptr[i + 1]; ++s) {
      const int a = P->pobs_lidx[s];
      double r[3], H1[18];
      vus_stereo_factor_cpu(poses + 12 * (size_t)i, points + 3 * (size_t)P->obs_point[a], P->meas + 3 * (size_t)a, P->K,
                            P->inv_sigma, r, H1, NULL);
      for (int rr = 0; rr < 6; ++rr) {
        for (int c = 0; c < 6; ++c) h[6 * rr + c] += H1[rr] * H1[c] + H1[6 + rr] * H1[6 + c] + H1[12 + rr] * H1[12 + c];
        g[rr] += H1[rr] * r[0] + H1[6 + rr] * r[1] + H1[12 + rr] * r[2];
      }
    }
    memcpy(Hpp + 36 * (size_t)i, h, sizeof h);
    memcpy(gp + 6 * (size_t)i, g, sizeof g);
  }
  err[0] = e + prior_terms(P, poses, NULL, Hpp, gp);
  return VUS_OK;
}

/* Damped landmark elimination.  The reduced camera system leaves in LAPACK lower band storage:
 * ab[(R - C) * n + C] = S(R, C) for R >= C, n = 6 n_poses, (6 band + 6) rows ("kd" = 6 band + 5). */
int vus_ba_schur_mt_cpu(const vus_ba_problem* P, const vus_ba_structure* S, double lambda, const double* W,
                        const double* V, const double* gl, const double* Hpp, const double* gp, double* Vinv, double* Y,
                        double* ab, double* gs) {
  if (!P || !S || !W || !V || !gl || !Hpp || !gp || !Vinv || !Y || !ab || !gs) return VUS_E_INVALID;
  const int nP = P->n_poses;
  const size_t n = 6 * (size_t)nP;
  memset(ab, 0, sizeof(double) * n * (6 * (size_t)S->band + 6));
#pragma omp parallel for schedule(dynamic, 64)
  for (int j = 0; j < P->n_points; ++j) {
    double* vi = Vinv + 6 * (size_t)j;
    sym3_inv(V + 6 * (size_t)j, lambda, vi);
    for (int a = P->point_ptr[j]; a < P->point_ptr[j + 1]; ++a) {
      const double* Ws = W + 18 * (size_t)P->obs_ppos[a];
      double* Ys = Y + 18 * (size_t)P->obs_ppos[a];
      for (int rr = 0; rr < 6; ++rr) {
        const double w0 = Ws[3 * rr], w1 = Ws[3 * rr + 1], w2 = Ws[3 * rr + 2];
        Ys[3 * rr + 0] = w0 * vi[0] + w1 * vi[1] + w2 * vi[2];
        Ys[3 * rr + 1] = w0 * vi[1] + w1 * vi[3] + w2 * vi[4];
        Ys[3 * rr + 2] = w0 * vi[2] + w1 * vi[4] + w2 * vi[5];
      }
    }
  }
#pragma omp parallel for schedule(dynamic, 4)
  for (int i = 0; i < nP; ++i) {
    double g[6];
    for (int k = 0; k < 6; ++k) g[k] = gp[6 * (size_t)i + k];
    for (int s = P->pose_ptr[i]; s < P->pose_ptr[i + 1]; ++s) {
      const int j = P->obs_point[P->pobs_lidx[s]];
      const double* Ys = Y + 18 * (size_t)s;
      for (int rr = 0; rr < 6; ++rr)
        g[rr] -= Ys[3 * rr] * gl[3 * (size_t)j] + Ys[3 * rr + 1] * gl[3 * (size_t)j + 1] + Ys[3 * rr + 2] * gl[3 * (size_t)j + 2];
    }
    for (int k = 0; k < 6; ++k) gs[6 * (size_t)i + k] = g[k];
    for (int rr = 0; rr < 6; ++rr)       /* diagonal block: Hpp + lambda I (the pair sums are subtracted below) */
      for (int c = 0; c <= rr; ++c)
        ab[(size_t)(rr - c) * n + 6 * (size_t)i + c] = Hpp[36 * (size_t)i + 6 * rr + c] + (rr == c ? lambda : 0.0);
  }
#pragma omp parallel for schedule(dynamic, 16)
  for (int q = 0; q < S->n_blocks; ++q) {
    const int i = S->blk_i[q], k = S->blk_k[q];
    double acc[36];
    memset(acc, 0, sizeof acc);
    for (int p = S->blk_ptr[q]; p < S->blk_ptr[q + 1]; ++p) {
      const double* Ya = Y + 18 * (size_t)S->pair_a[p];
      const double* Wb = W + 18 * (size_t)S->pair_b[p];
      for (int rr = 0; rr < 6; ++rr)
        for (int c = 0; c < 6; ++c) acc[6 * rr + c] += Ya[3 * rr] * Wb[3 * c] + Ya[3 * rr + 1] * Wb[3 * c + 1] + Ya[3 * rr + 2] * Wb[3 * c + 2];
    }
    for (int rr = 0; rr < 6; ++rr)
      for (int c = 0; c < 6; ++c) {
        const size_t R = 6 * (size_t)i + rr, C = 6 * (size_t)k + c;
        if (R >= C) ab[(R - C) * n + C] -= acc[6 * rr + c];
      }
  }
  return VUS_OK;
}

int vus_ba_backsub_mt_cpu(const vus_ba_problem* P, const double* W, const double* Vinv, const double* gl,
                          const double* dp, double* dl) {
  if (!P || !W || !Vinv || !gl || !dp || !dl) return VUS_E_INVALID;
#pragma omp parallel for schedule(dynamic, 64)
  for (int j = 0; j < P->n_points; ++j) {
    double t[3] = {gl[3 * (size_t)j], gl[3 * (size_t)j + 1], gl[3 * (size_t)j + 2]};
    for (int a = P->point_ptr[j]; a < P->point_ptr[j + 1]; ++a) {
      const double* Wa = W + 18 * (size_t)P->obs_ppos[a];
      const double* d = dp + 6 * (size_t)P->obs_pose[a];
      for (int rr = 0; rr < 6; ++rr)
        for (int c = 0; c < 3; ++c) t[c] += Wa[3 * rr + c] * d[rr];
    }
    const double* vi = Vinv + 6 * (size_t)j;
    dl[3 * (size_t)j + 0] = -(vi[0] * t[0] + vi[1] * t[1] + vi[2] * t[2]);
    dl[3 * (size_t)j + 1] = -(vi[1] * t[0] + vi[3] * t[1] + vi[4] * t[2]);
    dl[3 * (size_t)j + 2] = -(vi[2] * t[0] + vi[4] * t[1] + vi[5] * t[2]);
  }
  return VUS_OK;
}

int vus_ba_eval_step_mt_cpu(const vus_ba_problem* P, const double* poses, const double* points, const double* dp,
                            const double* dl, double* new_poses, double* new_points, double* out) {
  if (!P || !poses || !points || !dp || !dl || !new_poses || !new_points || !out) return VUS_E_INVALID;
#pragma omp parallel for schedule(static)
  for (int i = 0; i < P->n_poses; ++i) vus_pose_retract_cpu(poses + 12 * (size_t)i, dp + 6 * (size_t)i, new_poses + 12 * (size_t)i);
#pragma omp parallel for schedule(static)
  for (int k = 0; k < 3 * P->n_points; ++k) new_points[k] = points[k] + dl[k];
  double lin = 0, enew = 0;
#pragma omp parallel for reduction(+ : lin, enew) schedule(static)
  for (int a = 0; a < P->n_obs; ++a) {
    const int i = P->obs_pose[a], j = P->obs_point[a];
    double r[3], H1[18], H2[9];
    vus_stereo_factor_cpu(poses + 12 * (size_t)i, points + 3 * (size_t)j, P->meas + 3 * (size_t)a, P->K, P->inv_sigma, r, H1, H2);
    for (int rr = 0; rr < 3; ++rr) {
      double t = r[rr];
      for (int c = 0; c < 6; ++c) t += H1[6 * rr + c] * dp[6 * (size_t)i + c];
      for (int c = 0; c < 3; ++c) t += H2[3 * rr + c] * dl[3 * (size_t)j + c];
      lin += 0.5 * t * t;
    }
    vus_stereo_factor_cpu(new_poses + 12 * (size_t)i, new_points + 3 * (size_t)j, P->meas + 3 * (size_t)a, P->K, P->inv_sigma,
                          r, NULL, NULL);
    enew += 0.5 * (r[0] * r[0] + r[1] * r[1] + r[2] * r[2]);
  }
  out[0] = lin + prior_terms(P, poses, dp, NULL, NULL);
  out[1] = enew + prior_terms(P, new_poses, NULL, NULL, NULL);
  return VUS_OK;
}
