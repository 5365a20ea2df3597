/* vus_oracle_ba.c -- CPU ORACLE (test infrastructure, NOT the product).
 *
 * Plain-C fp64 restatement of the bundle-adjustment arithmetic behind
 *   gtsam.GenericStereoFactor3D(...)                            (reference batch.py:300-305)
 *   gtsam.PriorFactorPose3(...)                                 (reference batch.py:281)
 *   gtsam.LevenbergMarquardtOptimizer(g, v, params).optimize()  (reference batch.py:337)
 *
 * PARITY UNPINNED: GTSAM is a third-party dependency of the reference, un-vendored and with no
 * pinned version (reference README.md:18,21); it is not installed here, and the reference has no
 * tests or golden vectors.  What follows restates GTSAM's PUBLISHED algorithm
 * (gtsam 4.x: StereoCamera::project2, GenericStereoFactor::evaluateError, PriorFactor::evaluateError,
 * Pose3::Expmap/Logmap, LevenbergMarquardtOptimizer::iterate/tryLambda, NonlinearOptimizer::
 * defaultOptimize/checkConvergence) from the maths; it is pinned by finite-difference Jacobian
 * checks, a dense-solve equivalence test and a ground-truth recovery test in tests/, and by
 * self-generated golden vectors in tests/golden/.
 *
 * One deliberate difference from GTSAM's internals, which does not change the solution of the
 * linear system: the damped normal equations are solved by an explicit landmark Schur complement +
 * block-band Cholesky instead of multifrontal elimination under COLAMD.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this.
 */
#define _USE_MATH_DEFINES
#define _GNU_SOURCE
#include <math.h>
#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include "vus_oracle.h"

/* ---------------------------------------------------------------------------------------------
 * small dense helpers
 * ------------------------------------------------------------------------------------------- */
static void sym3_inverse(const double* v /*xx,xy,xz,yy,yz,zz*/, double* o) {
  double a = v[0], b = v[1], c = v[2], d = v[3], e = v[4], f = v[5];
  double c00 = d * f - e * e, c01 = c * e - b * f, c02 = b * e - c * d;
  double det = a * c00 + b * c01 + c * c02;
  double id = 1.0 / det;
  o[0] = c00 * id;
  o[1] = c01 * id;
  o[2] = c02 * id;
  o[3] = (a * f - c * c) * id;
  o[4] = (b * c - a * e) * id;
  o[5] = (a * d - b * b) * id;
}

static inline double sym3_at(const double* v, int r, int c) {
  static const int idx[3][3] = {{0, 1, 2}, {1, 3, 4}, {2, 4, 5}};
  return v[idx[r][c]];
}

/* Rodrigues: gtsam SO3 ExpmapFunctor (nearZero: theta^2 <= DBL_EPSILON -> I + [w]x). */
static void so3_expmap(const double* w, double* R) {
  double th2 = w[0] * w[0] + w[1] * w[1] + w[2] * w[2];
  double Wx[9] = {0, -w[2], w[1], w[2], 0, -w[0], -w[1], w[0], 0};
  if (th2 <= 2.220446049250313e-16) {
    for (int i = 0; i < 9; ++i) R[i] = Wx[i] + (i % 4 == 0 ? 1.0 : 0.0);
    return;
  }
  double th = sqrt(th2);
  double s = sin(th) / th;
  double sh = sin(0.5 * th);
  double c = 2.0 * sh * sh / th2; /* (1 - cos th) / th^2 */
  for (int r = 0; r < 3; ++r)
    for (int cc = 0; cc < 3; ++cc) {
      double ww = 0;
      for (int k = 0; k < 3; ++k) ww += Wx[3 * r + k] * Wx[3 * k + cc];
      R[3 * r + cc] = (r == cc ? 1.0 : 0.0) + s * Wx[3 * r + cc] + c * ww;
    }
}

/* gtsam SO3::Logmap (the 4.0/4.1 form; near pi uses the simple axis formulas). */
static void so3_logmap(const double* R, double* w) {
  double tr = R[0] + R[4] + R[8];
  if (tr + 1.0 < 1e-10) {
    if (fabs(R[8] + 1.0) > 1e-5) {
      double k = M_PI / sqrt(2.0 + 2.0 * R[8]);
      w[0] = k * R[2]; w[1] = k * R[5]; w[2] = k * (1.0 + R[8]);
    } else if (fabs(R[4] + 1.0) > 1e-5) {
      double k = M_PI / sqrt(2.0 + 2.0 * R[4]);
      w[0] = k * R[1]; w[1] = k * (1.0 + R[4]); w[2] = k * R[7];
    } else {
      double k = M_PI / sqrt(2.0 + 2.0 * R[0]);
      w[0] = k * (1.0 + R[0]); w[1] = k * R[3]; w[2] = k * R[6];
    }
    return;
  }
  double mag;
  double tr3 = tr - 3.0;
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

/* Pose3::Expmap(xi = (w, v)) -> (R, t). */
static void se3_expmap(const double* xi, double* R, double* t) {
  const double* w = xi;
  const double* v = xi + 3;
  so3_expmap(w, R);
  double th2 = w[0] * w[0] + w[1] * w[1] + w[2] * w[2];
  if (th2 > 2.220446049250313e-16) {
    double wv = w[0] * v[0] + w[1] * v[1] + w[2] * v[2];
    double c[3] = {w[1] * v[2] - w[2] * v[1], w[2] * v[0] - w[0] * v[2], w[0] * v[1] - w[1] * v[0]};
    for (int r = 0; r < 3; ++r) {
      double Rc = R[3 * r] * c[0] + R[3 * r + 1] * c[1] + R[3 * r + 2] * c[2];
      t[r] = (c[r] - Rc + w[r] * wv) / th2;
    }
  } else {
    t[0] = v[0]; t[1] = v[1]; t[2] = v[2];
  }
}

/* Pose3::Logmap((R, t)) -> xi. */
static void se3_logmap(const double* R, const double* t, double* xi) {
  double w[3];
  so3_logmap(R, w);
  double th = sqrt(w[0] * w[0] + w[1] * w[1] + w[2] * w[2]);
  xi[0] = w[0]; xi[1] = w[1]; xi[2] = w[2];
  if (th < 1e-10) {
    xi[3] = t[0]; xi[4] = t[1]; xi[5] = t[2];
    return;
  }
  double k[3] = {w[0] / th, w[1] / th, w[2] / th};
  double WT[3] = {k[1] * t[2] - k[2] * t[1], k[2] * t[0] - k[0] * t[2], k[0] * t[1] - k[1] * t[0]};
  double WWT[3] = {k[1] * WT[2] - k[2] * WT[1], k[2] * WT[0] - k[0] * WT[2], k[0] * WT[1] - k[1] * WT[0]};
  double tn = tan(0.5 * th);
  for (int r = 0; r < 3; ++r) xi[3 + r] = t[r] - (0.5 * th) * WT[r] + (1.0 - th / (2.0 * tn)) * WWT[r];
}

/* pose (+) xi = T * Exp(xi) */
void vus_pose_retract_cpu(const double* T, const double* xi, double* out) {
  double Re[9], te[3];
  se3_expmap(xi, Re, te);
  for (int r = 0; r < 3; ++r) {
    for (int c = 0; c < 3; ++c)
      out[3 * r + c] = T[3 * r] * Re[c] + T[3 * r + 1] * Re[3 + c] + T[3 * r + 2] * Re[6 + c];
    out[9 + r] = T[9 + r] + (T[3 * r] * te[0] + T[3 * r + 1] * te[1] + T[3 * r + 2] * te[2]);
  }
}

/* localCoordinates: Logmap(T^-1 * T2) */
void vus_pose_local_cpu(const double* T, const double* T2, double* xi) {
  double R[9], t[3];
  for (int r = 0; r < 3; ++r) {
    for (int c = 0; c < 3; ++c)
      R[3 * r + c] = T[r] * T2[c] + T[3 + r] * T2[3 + c] + T[6 + r] * T2[6 + c];
    t[r] = T[r] * (T2[9] - T[9]) + T[3 + r] * (T2[10] - T[10]) + T[6 + r] * (T2[11] - T[11]);
  }
  se3_logmap(R, t, xi);
}

/* GenericStereoFactor: whitened residual r[3], H1[18] (3x6 wrt pose tangent), H2[9] (3x3 wrt point).
 * H1/H2 may be NULL. */
void vus_stereo_factor_cpu(const double* T, const double* p, const double* m, const double* K, double w,
                           double* r, double* H1, double* H2) {
  const double fx = K[0], fy = K[1], cx = K[3], cy = K[4], b = K[5];
  double d0 = p[0] - T[9], d1 = p[1] - T[10], d2 = p[2] - T[11];
  double x = T[0] * d0 + T[3] * d1 + T[6] * d2; /* q = R^T (p - t) */
  double y = T[1] * d0 + T[4] * d1 + T[7] * d2;
  double z = T[2] * d0 + T[5] * d1 + T[8] * d2;
  if (z <= 0.0) { /* StereoCheiralityException, throwCheirality = false */
    r[0] = r[1] = r[2] = 2.0 * fx * w;
    if (H1) memset(H1, 0, 18 * sizeof(double));
    if (H2) memset(H2, 0, 9 * sizeof(double));
    return;
  }
  double d = 1.0 / z;
  double uL = cx + d * fx * x, uR = cx + d * fx * (x - b), v = cy + d * fy * y;
  r[0] = (uL - m[0]) * w;
  r[1] = (uR - m[1]) * w;
  r[2] = (v - m[2]) * w;
  if (!H1 && !H2) return;
  /* J = d(uL,uR,v)/dq, whitened */
  double J[9] = {w * d * fx, 0, -w * d * d * fx * x,
                 w * d * fx, 0, -w * d * d * fx * (x - b),
                 0, w * d * fy, -w * d * d * fy * y};
  if (H2) /* dq/dp = R^T */
    for (int rr = 0; rr < 3; ++rr)
      for (int c = 0; c < 3; ++c)
        H2[3 * rr + c] = J[3 * rr] * T[3 * c] + J[3 * rr + 1] * T[3 * c + 1] + J[3 * rr + 2] * T[3 * c + 2];
  if (H1) { /* dq/dxi = [ [q]x , -I ] */
    double Q[9] = {0, -z, y, z, 0, -x, -y, x, 0};
    for (int rr = 0; rr < 3; ++rr)
      for (int c = 0; c < 3; ++c) {
        H1[6 * rr + c] = J[3 * rr] * Q[c] + J[3 * rr + 1] * Q[3 + c] + J[3 * rr + 2] * Q[6 + c];
        H1[6 * rr + 3 + c] = -J[3 * rr + c];
      }
  }
}

/* PriorFactorPose3: e = -Local(x, prior) = -Logmap(x^-1 prior), H = I (gtsam's PriorFactor). */
static void prior_residual(const double* T, const double* Tp, const double* w6, double* r) {
  double xi[6];
  vus_pose_local_cpu(T, Tp, xi);
  for (int k = 0; k < 6; ++k) r[k] = -xi[k] * w6[k];
}

/* ---------------------------------------------------------------------------------------------
 * the ABI twins
 * ------------------------------------------------------------------------------------------- */
int vus_ba_error_cpu(const vus_ba_problem* P, const double* poses, const double* points, double* err) {
  if (!P || !poses || !points || !err) return VUS_E_INVALID;
  double e = 0;
  for (int a = 0; a < P->n_obs; ++a) {
    double r[3];
    vus_stereo_factor_cpu(poses + 12 * P->obs_pose[a], points + 3 * P->obs_point[a], P->meas + 3 * a, P->K,
                          P->inv_sigma, r, NULL, NULL);
    e += 0.5 * (r[0] * r[0] + r[1] * r[1] + r[2] * r[2]);
  }
  for (int q = 0; q < P->n_priors; ++q) {
    double r[6];
    prior_residual(poses + 12 * P->prior_pose[q], P->prior_T + 12 * q, P->prior_w + 6 * q, r);
    for (int k = 0; k < 6; ++k) e += 0.5 * r[k] * r[k];
  }
  err[0] = e;
  return VUS_OK;
}

int vus_ba_linearize_cpu(const vus_ba_problem* P, const double* poses, const double* points, double* W,
                         double* V, double* gl, double* Hpp, double* gp, double* err) {
  if (!P || !poses || !points || !W || !V || !gl || !Hpp || !gp || !err) return VUS_E_INVALID;
  memset(V, 0, sizeof(double) * 6 * P->n_points);
  memset(gl, 0, sizeof(double) * 3 * P->n_points);
  memset(Hpp, 0, sizeof(double) * 36 * P->n_poses);
  memset(gp, 0, sizeof(double) * 6 * P->n_poses);
  double e = 0;
  for (int a = 0; a < P->n_obs; ++a) {
    int i = P->obs_pose[a], j = P->obs_point[a];
    double r[3], H1[18], H2[9];
    vus_stereo_factor_cpu(poses + 12 * i, points + 3 * j, P->meas + 3 * a, P->K, P->inv_sigma, r, H1, H2);
    e += 0.5 * (r[0] * r[0] + r[1] * r[1] + r[2] * r[2]);
    double* Wa = W + 18 * (size_t)a;
    for (int rr = 0; rr < 6; ++rr)
      for (int c = 0; c < 3; ++c)
        Wa[3 * rr + c] = H1[rr] * H2[c] + H1[6 + rr] * H2[3 + c] + H1[12 + rr] * H2[6 + c];
    int u = 0;
    for (int rr = 0; rr < 3; ++rr)
      for (int c = rr; c < 3; ++c, ++u)
        V[6 * j + u] += H2[rr] * H2[c] + H2[3 + rr] * H2[3 + c] + H2[6 + rr] * H2[6 + c];
    for (int c = 0; c < 3; ++c) gl[3 * j + c] += H2[c] * r[0] + H2[3 + c] * r[1] + H2[6 + c] * r[2];
    for (int rr = 0; rr < 6; ++rr) {
      for (int c = 0; c < 6; ++c)
        Hpp[36 * i + 6 * rr + c] += H1[rr] * H1[c] + H1[6 + rr] * H1[6 + c] + H1[12 + rr] * H1[12 + c];
      gp[6 * i + rr] += H1[rr] * r[0] + H1[6 + rr] * r[1] + H1[12 + rr] * r[2];
    }
  }
  for (int q = 0; q < P->n_priors; ++q) {
    int i = P->prior_pose[q];
    double r[6];
    prior_residual(poses + 12 * i, P->prior_T + 12 * q, P->prior_w + 6 * q, r);
    for (int k = 0; k < 6; ++k) {
      double w = P->prior_w[6 * q + k];
      Hpp[36 * i + 7 * k] += w * w;
      gp[6 * i + k] += w * r[k];
      e += 0.5 * r[k] * r[k];
    }
  }
  err[0] = e;
  return VUS_OK;
}

/* Block structure of the reduced camera system, row by row: the plain statement of what csrc/structure.hip builds
 * (and of ba_pack.build_structure).  Pairs of row i: (slot s of pose i, observation b of the same landmark with
 * pose k <= i); block = (i, k); blocks by ascending k, pairs of a block by ascending s. */
int vus_ba_structure_count_cpu(const vus_ba_problem* P, int band, int* row_blocks, int* row_pairs) {
  if (!P || !row_blocks || !row_pairs || band < 0) return VUS_E_INVALID;
  int* cnt = (int*)calloc((size_t)band + 1, sizeof(int));
  if (!cnt) return VUS_E_INVALID;
  for (int i = 0; i < P->n_poses; ++i) {
    memset(cnt, 0, sizeof(int) * ((size_t)band + 1));
    int np = 0, nb = 0;
    for (int s = P->pose_ptr[i]; s < P->pose_ptr[i + 1]; ++s) {
      const int a = P->pobs_lidx[s];
      for (int b = P->point_ptr[P->obs_point[a]]; b <= a; ++b) {
        const int d = i - P->obs_pose[b];
        if (d < 0 || d > band) { free(cnt); return VUS_E_INVALID; }
        ++cnt[d];
        ++np;
      }
    }
    for (int d = 0; d <= band; ++d) nb += cnt[d] > 0;
    row_blocks[i] = nb;
    row_pairs[i] = np;
  }
  free(cnt);
  return VUS_OK;
}

int vus_ba_structure_fill_cpu(const vus_ba_problem* P, int band, const int* blk_base, const int* pair_base,
                              int* blk_ptr, int* blk_i, int* blk_k, int* pair_a, int* pair_b) {
  if (!P || !blk_base || !pair_base || !blk_ptr || !blk_i || !blk_k || !pair_a || !pair_b || band < 0)
    return VUS_E_INVALID;
  int* cnt = (int*)calloc((size_t)band + 1, sizeof(int));
  int* cur = (int*)calloc((size_t)band + 1, sizeof(int));
  if (!cnt || !cur) { free(cnt); free(cur); return VUS_E_INVALID; }
  for (int i = 0; i < P->n_poses; ++i) {
    memset(cnt, 0, sizeof(int) * ((size_t)band + 1));
    for (int s = P->pose_ptr[i]; s < P->pose_ptr[i + 1]; ++s) {
      const int a = P->pobs_lidx[s];
      for (int b = P->point_ptr[P->obs_point[a]]; b <= a; ++b) ++cnt[i - P->obs_pose[b]];
    }
    int q = blk_base[i], pos = pair_base[i];
    for (int d = band; d >= 0; --d) {          /* ascending k */
      cur[d] = pos;
      if (cnt[d] > 0) {
        blk_i[q] = i;
        blk_k[q] = i - d;
        blk_ptr[q] = pos;
        ++q;
        pos += cnt[d];
      }
    }
    for (int s = P->pose_ptr[i]; s < P->pose_ptr[i + 1]; ++s) {
      const int a = P->pobs_lidx[s];
      for (int b = P->point_ptr[P->obs_point[a]]; b <= a; ++b) {
        const int d = i - P->obs_pose[b];
        pair_a[cur[d]] = s;
        pair_b[cur[d]] = P->obs_ppos[b];
        ++cur[d];
      }
    }
  }
  blk_ptr[blk_base[P->n_poses]] = pair_base[P->n_poses];
  free(cnt);
  free(cur);
  return VUS_OK;
}

/* Twin of vus_ba_schur by the plain per-landmark statement: the tile lists T are a schedule of the SAME sum and are not
 * read (T may be NULL); Sband has band_nodes + 1 blocks per row; W, Y in L-order; pose_stride 1. */
int vus_ba_schur_cpu(const vus_ba_problem* P, const vus_ba_tiles* T, double lambda, const double* W, const double* V,
                     const double* gl, const double* Hpp, const double* gp, double* Vinv, double* Y, double* Sband,
                     int band_nodes, double* gs, int* counter) {
  (void)T; (void)counter;
  if (!P || !W || !V || !gl || !Hpp || !gp || !Vinv || !Y || !Sband || !gs || band_nodes < 0 || P->pose_stride > 1) return VUS_E_INVALID;
  const int B = band_nodes, nP = P->n_poses;
  memset(Sband, 0, sizeof(double) * 36 * (size_t)nP * (B + 1));
  for (int i = 0; i < nP; ++i) {
    double* D = Sband + 36 * (size_t)i * (B + 1);
    for (int k = 0; k < 36; ++k) D[k] = Hpp[36 * i + k];
    for (int k = 0; k < 6; ++k) D[7 * k] += lambda;
    for (int k = 0; k < 6; ++k) gs[6 * i + k] = gp[6 * i + k];
  }
  for (int j = 0; j < P->n_points; ++j) {
    double Vd[6];
    for (int k = 0; k < 6; ++k) Vd[k] = V[6 * j + k];
    Vd[0] += lambda; Vd[3] += lambda; Vd[5] += lambda;
    sym3_inverse(Vd, Vinv + 6 * j);
    const double* Vi = Vinv + 6 * j;
    for (int a = P->point_ptr[j]; a < P->point_ptr[j + 1]; ++a) {
      const double* Wa = W + 18 * (size_t)a;
      double* Ya = Y + 18 * (size_t)a;
      for (int rr = 0; rr < 6; ++rr)
        for (int c = 0; c < 3; ++c)
          Ya[3 * rr + c] = Wa[3 * rr] * sym3_at(Vi, 0, c) + Wa[3 * rr + 1] * sym3_at(Vi, 1, c) +
                           Wa[3 * rr + 2] * sym3_at(Vi, 2, c);
      int ia = P->obs_pose[a];
      for (int rr = 0; rr < 6; ++rr)
        gs[6 * ia + rr] -= Ya[3 * rr] * gl[3 * j] + Ya[3 * rr + 1] * gl[3 * j + 1] + Ya[3 * rr + 2] * gl[3 * j + 2];
    }
    for (int a = P->point_ptr[j]; a < P->point_ptr[j + 1]; ++a) {
      int ia = P->obs_pose[a];
      const double* Ya = Y + 18 * (size_t)a;
      for (int b = P->point_ptr[j]; b <= a; ++b) {
        int ib = P->obs_pose[b];
        if (ia - ib > B || ib > ia || (ia == ib && a != b)) return VUS_E_INVALID;
        const double* Wb = W + 18 * (size_t)b;
        double* blk = Sband + 36 * ((size_t)ia * (B + 1) + (ia - ib));
        for (int rr = 0; rr < 6; ++rr)
          for (int c = 0; c < 6; ++c)
            blk[6 * rr + c] -= Ya[3 * rr] * Wb[3 * c] + Ya[3 * rr + 1] * Wb[3 * c + 1] + Ya[3 * rr + 2] * Wb[3 * c + 2];
      }
    }
  }
  return VUS_OK;
}

int vus_ba_add_diag_cpu(double* Sband, int n_poses, int band, double value) {
  if (!Sband || n_poses < 1 || band < 0) return VUS_E_INVALID;
  for (int i = 0; i < n_poses; ++i)
    for (int k = 0; k < 6; ++k) Sband[36 * (size_t)i * (band + 1) + 7 * k] += value;
  return VUS_OK;
}

/* scalar view of the block band: element (R, C), R >= C, R/6 - C/6 <= B */
static inline double* band_at(double* Sb, int B, int R, int C) {
  int i = R / 6, k = C / 6;
  return Sb + 36 * ((size_t)i * (B + 1) + (i - k)) + 6 * (R % 6) + (C % 6);
}

/* left-looking Cholesky of the block band, lower factor in place; returns 0 or failing column + 1 */
static int band_factor(double* Sband, int n_poses, int band) {
  const int n = 6 * n_poses, B = band;
  for (int C = 0; C < n; ++C) {
    int kc0 = (C / 6 - B) * 6;
    if (kc0 < 0) kc0 = 0;
    double s = *band_at(Sband, B, C, C);
    for (int k = kc0; k < C; ++k) { double l = *band_at(Sband, B, C, k); s -= l * l; }
    if (!(s > 0.0)) return C + 1;
    double lcc = sqrt(s);
    *band_at(Sband, B, C, C) = lcc;
    int Rmax = (C / 6 + B) * 6 + 5;
    if (Rmax > n - 1) Rmax = n - 1;
    for (int R = C + 1; R <= Rmax; ++R) {
      int k0 = (R / 6 - B) * 6;
      if (k0 < 0) k0 = 0;
      double t = *band_at(Sband, B, R, C);
      for (int k = k0; k < C; ++k) t -= *band_at(Sband, B, R, k) * *band_at(Sband, B, C, k);
      *band_at(Sband, B, R, C) = t / lcc;
    }
  }
  return 0;
}

/* x <- L^-T L^-1 x */
static void band_substitute(double* Sband, int n_poses, int band, double* x) {
  const int n = 6 * n_poses, B = band;
  for (int R = 0; R < n; ++R) {
    int k0 = (R / 6 - B) * 6;
    if (k0 < 0) k0 = 0;
    double t = x[R];
    for (int k = k0; k < R; ++k) t -= *band_at(Sband, B, R, k) * x[k];
    x[R] = t / *band_at(Sband, B, R, R);
  }
  for (int C = n - 1; C >= 0; --C) {
    int Rmax = (C / 6 + B) * 6 + 5;
    if (Rmax > n - 1) Rmax = n - 1;
    double t = x[C];
    for (int R = C + 1; R <= Rmax; ++R) t -= *band_at(Sband, B, R, C) * x[R];
    x[C] = t / *band_at(Sband, B, C, C);
  }
}

int vus_ba_band_solve_cpu(double* Sband, int n_poses, int band, const double* gs, double* dp, int* status) {
  if (!Sband || !gs || !dp || !status || n_poses < 1 || band < 0) return VUS_E_INVALID;
  status[0] = band_factor(Sband, n_poses, band);
  if (status[0]) return VUS_OK;
  for (int k = 0; k < 6 * n_poses; ++k) dp[k] = -gs[k];
  band_substitute(Sband, n_poses, band, dp);
  return VUS_OK;
}

int vus_ba_band_solve_multi_cpu(double* Sband, int n_nodes, int band, double* rhs, int n_rhs, int* status) {
  if (!Sband || !rhs || !status || n_nodes < 1 || band < 0 || n_rhs < 1 || n_rhs > 8) return VUS_E_INVALID;
  status[0] = band_factor(Sband, n_nodes, band);
  if (status[0]) return VUS_OK;
  for (int q = 0; q < n_rhs; ++q) band_substitute(Sband, n_nodes, band, rhs + (size_t)q * 6 * n_nodes);
  return VUS_OK;
}

/* Twins of the two-sided solves: the oracle states WHAT is solved (S x = rhs), not in which order the GPU
 * eliminates; `work` is ignored. */
int vus_ba_band_solve_split_cpu(double* Sband, int n_poses, int band, const double* gs, double* dp, int* status,
                                double* work) {
  (void)work;
  return vus_ba_band_solve_cpu(Sband, n_poses, band, gs, dp, status);
}

int vus_ba_band_solve_multi_split_cpu(double* Sband, int n_nodes, int band, double* rhs, int n_rhs, int* status,
                                      double* work) {
  (void)work;
  return vus_ba_band_solve_multi_cpu(Sband, n_nodes, band, rhs, n_rhs, status);
}

int vus_ba_backsub_cpu(const vus_ba_problem* P, const double* W, const double* Vinv, const double* gl,
                       const double* dp, double* dl) {
  if (!P || !W || !Vinv || !gl || !dp || !dl) return VUS_E_INVALID;
  for (int j = 0; j < P->n_points; ++j) {
    double t[3] = {gl[3 * j], gl[3 * j + 1], gl[3 * j + 2]};
    for (int a = P->point_ptr[j]; a < P->point_ptr[j + 1]; ++a) {
      const double* Wa = W + 18 * (size_t)a;
      const double* d = dp + 6 * P->obs_pose[a];
      for (int c = 0; c < 3; ++c)
        for (int rr = 0; rr < 6; ++rr) t[c] += Wa[3 * rr + c] * d[rr];
    }
    const double* Vi = Vinv + 6 * j;
    for (int c = 0; c < 3; ++c)
      dl[3 * j + c] = -(sym3_at(Vi, c, 0) * t[0] + sym3_at(Vi, c, 1) * t[1] + sym3_at(Vi, c, 2) * t[2]);
  }
  return VUS_OK;
}

int vus_ba_eval_step_cpu(const vus_ba_problem* P, const double* poses, const double* points, const double* dp,
                         const double* dl, double* new_poses, double* new_points, double* out) {
  if (!P || !poses || !points || !dp || !dl || !new_poses || !new_points || !out) return VUS_E_INVALID;
  for (int i = 0; i < P->n_poses; ++i) vus_pose_retract_cpu(poses + 12 * i, dp + 6 * i, new_poses + 12 * i);
  for (int k = 0; k < 3 * P->n_points; ++k) new_points[k] = points[k] + dl[k];
  double lin = 0;
  for (int a = 0; a < P->n_obs; ++a) {
    int i = P->obs_pose[a], j = P->obs_point[a];
    double r[3], H1[18], H2[9];
    vus_stereo_factor_cpu(poses + 12 * i, points + 3 * j, P->meas + 3 * a, P->K, P->inv_sigma, r, H1, H2);
    for (int rr = 0; rr < 3; ++rr) {
      double t = r[rr];
      for (int c = 0; c < 6; ++c) t += H1[6 * rr + c] * dp[6 * i + c];
      for (int c = 0; c < 3; ++c) t += H2[3 * rr + c] * dl[3 * j + c];
      lin += 0.5 * t * t;
    }
  }
  for (int q = 0; q < P->n_priors; ++q) {
    int i = P->prior_pose[q];
    double r[6];
    prior_residual(poses + 12 * i, P->prior_T + 12 * q, P->prior_w + 6 * q, r);
    for (int k = 0; k < 6; ++k) {
      double t = r[k] + P->prior_w[6 * q + k] * dp[6 * i + k];
      lin += 0.5 * t * t;
    }
  }
  out[0] = lin;
  return vus_ba_error_cpu(P, new_poses, new_points, out + 1);
}

/* ---------------------------------------------------------------------------------------------
 * Levenberg-Marquardt, gtsam defaults (LevenbergMarquardtParams()):
 *   lambdaInitial 1e-5, lambdaFactor 10, lambdaUpperBound 1e5, lambdaLowerBound 0,
 *   minModelFidelity 1e-3, diagonalDamping false, useFixedLambdaFactor true,
 *   maxIterations 100, relativeErrorTol 1e-5, absoluteErrorTol 1e-5, errorTol 0.
 * ------------------------------------------------------------------------------------------- */
typedef struct vus_lm_params {
  double lambda_initial, lambda_factor, lambda_upper, lambda_lower, min_model_fidelity;
  double rel_tol, abs_tol, error_tol;
  int max_iterations;
} vus_lm_params;

#define VUS_LM_HIST 128
typedef struct vus_lm_report {
  int iterations;    /* accepted steps (gtsam's iterations()) */
  int outer;         /* calls of iterate() = linearisations */
  int tries;         /* linear solves */
  int status;        /* 0 converged, 1 max iterations, 2 lambda upper bound hit */
  double initial_error, final_error, final_lambda;
  double err_hist[VUS_LM_HIST];    /* error after each iterate() */
  double lambda_hist[VUS_LM_HIST]; /* lambda after each outer iteration */
} vus_lm_report;

int vus_ba_lm_optimize_cpu(const vus_ba_problem* P, int band, const vus_lm_params* prm, double* poses,
                           double* points, vus_lm_report* rep) {
  if (!P || !prm || !poses || !points || !rep || band < 0) return VUS_E_INVALID;
  const int nP = P->n_poses, nL = P->n_points, nO = P->n_obs;
  double* W = malloc(sizeof(double) * 18 * (size_t)nO);
  double* Y = malloc(sizeof(double) * 18 * (size_t)nO);
  double* V = malloc(sizeof(double) * 6 * (size_t)nL);
  double* Vinv = malloc(sizeof(double) * 6 * (size_t)nL);
  double* gl = malloc(sizeof(double) * 3 * (size_t)nL);
  double* dl = malloc(sizeof(double) * 3 * (size_t)nL);
  double* Hpp = malloc(sizeof(double) * 36 * (size_t)nP);
  double* gp = malloc(sizeof(double) * 6 * (size_t)nP);
  double* gs = malloc(sizeof(double) * 6 * (size_t)nP);
  double* dp = malloc(sizeof(double) * 6 * (size_t)nP);
  double* Sb = malloc(sizeof(double) * 36 * (size_t)nP * (band + 1));
  double* nposes = malloc(sizeof(double) * 12 * (size_t)nP);
  double* npoints = malloc(sizeof(double) * 3 * (size_t)nL);
  int rc = VUS_OK;
  memset(rep, 0, sizeof *rep);
  double lambda = prm->lambda_initial;
  double current;
  vus_ba_error_cpu(P, poses, points, &current);
  rep->initial_error = current;
  rep->status = 1;
  while (rep->iterations < prm->max_iterations) {
    /* ---- iterate(): linearise once, then search lambda ---- */
    double lin0;
    rc = vus_ba_linearize_cpu(P, poses, points, W, V, gl, Hpp, gp, &lin0);
    if (rc) break;
    double new_error = current;
    int stop_lambda_search = 0, accepted = 0;
    for (;;) {
      int status = 0;
      rc = vus_ba_schur_cpu(P, NULL, lambda, W, V, gl, Hpp, gp, Vinv, Y, Sb, band, gs, NULL);
      if (rc) break;
      vus_ba_band_solve_cpu(Sb, nP, band, gs, dp, &status);
      ++rep->tries;
      int success = 0;
      if (status == 0) {
        vus_ba_backsub_cpu(P, W, Vinv, gl, dp, dl);
        double out[2];
        vus_ba_eval_step_cpu(P, poses, points, dp, dl, nposes, npoints, out);
        double lin_change = lin0 - out[0]; /* oldLinearizedError - newlinearizedError */
        if (lin_change >= 0.0) {           /* "step is valid" */
          double cost_change = current - out[1];
          if (lin_change > 2.220446049250313e-16 * lin0) {
            double fidelity = cost_change / lin_change;
            success = fidelity > prm->min_model_fidelity;
          }
          if (fabs(cost_change) < prm->rel_tol * current) stop_lambda_search = 1;
          if (success) {
            memcpy(poses, nposes, sizeof(double) * 12 * (size_t)nP);
            memcpy(points, npoints, sizeof(double) * 3 * (size_t)nL);
            new_error = out[1];
          }
        }
      }
      if (success) { /* decreaseLambda, fixed factor; the only place gtsam counts an iteration */
        lambda /= prm->lambda_factor;
        if (lambda < prm->lambda_lower) lambda = prm->lambda_lower;
        accepted = 1;
        break;
      }
      if (stop_lambda_search) break;
      lambda *= prm->lambda_factor; /* increaseLambda */
      if (lambda >= prm->lambda_upper) { rep->status = 2; break; }
    }
    if (rc) break;
    if (rep->outer < VUS_LM_HIST) {
      rep->err_hist[rep->outer] = new_error;
      rep->lambda_hist[rep->outer] = lambda;
    }
    ++rep->outer;
    rep->iterations += accepted;
    /* ---- checkConvergence ---- */
    int converged;
    if (new_error <= prm->error_tol) {
      converged = 1;
    } else {
      double abs_dec = current - new_error;
      double rel_dec = abs_dec / current;
      converged = (rel_dec <= prm->rel_tol) || (abs_dec <= prm->abs_tol);
    }
    current = new_error;
    if (rep->status == 2) break;
    if (converged) { rep->status = 0; break; }
    if (!isfinite(current)) break;
  }
  rep->final_error = current;
  rep->final_lambda = lambda;
  free(W); free(Y); free(V); free(Vinv); free(gl); free(dl); free(Hpp); free(gp); free(gs); free(dp);
  free(Sb); free(nposes); free(npoints);
  return rc;
}
