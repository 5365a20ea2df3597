/* vus_oracle_frontend.c -- CPU ORACLE (test infrastructure, NOT the product).
 *
 * Plain-C restatement of the stereo ORB front-end on the hot path.  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library; the product
 * (visual-underwater-slam_amd/) never does.
 *
 * PARITY UNPINNED: the reference repository holds no front-end source, no tests and no golden
 * vectors for this path.  The front-end it wires in is the external
 * `gtsam_vio/ImageProcessorNodelet` (reference launch/stereo.launch:33-55, parameters :37-47),
 * whose output batch.py consumes (batch.py:29,149-154,323).  This file therefore restates the
 * PUBLISHED algorithms named by BASELINE.json's north_star, with every integer choice spelled out:
 *   - FAST-9/16 with score and strict 3x3 non-max suppression: Rosten & Drummond, "Machine
 *     learning for high-speed corner detection", ECCV 2006; threshold 10 = stereo.launch:43.
 *   - intensity-centroid orientation in 30 bins of 12 degrees and 256-bit steered BRIEF on a
 *     smoothed 31x31 patch: Rublee et al., "ORB", ICCV 2011, sec. 3.2 and 4.2.
 *   - brute-force Hamming matching with a row gate (stereo_threshold=5 = stereo.launch:47).
 *   - get_landmarks: batch.py:144-176 (this one IS in the reference and is followed verbatim).
 * PINNED BY THE REFERENCE ITSELF (round 4): vus_triangulate_cpu and vus_emit_stereo_factors_cpu -- get_landmarks and the
 * landmark loop of batch_create, batch.py:144-176, 295-305 -- are checked against outputs of /root/reference/batch.py,
 * run unmodified in the build container (tests/golden/make_reference_fixtures.py -> tests/golden/ref_batch_*.npz;
 * tests/test_reference_fixtures.py): measurements and factor order bit for bit, world points to 1 ulp.
 * For FAST / orientation / rBRIEF / Hamming the oracle is pinned only by the known-answer tests in tests/ (hand-built patches), by
 * self-generated golden vectors under tests/golden/, and -- detector corner set and orientation bin
 * only -- by agreement with scikit-image 0.18.3, an independent implementation of the same published
 * algorithms (tests/test_thirdparty_crosscheck.py).
 *
 * Everything here is written for clarity, one pixel / one keypoint at a time.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#ifdef _OPENMP
#include <omp.h>
#endif
#include "../include/vus.h"
#include "../include/vus_orb_tables.h"

/* Thread count of the image/pair-level OpenMP loops (the cpu_baseline leg states it as `cores`). */
void vus_oracle_set_threads(int n) {
#ifdef _OPENMP
  omp_set_num_threads(n < 1 ? 1 : n);
#else
  (void)n;
#endif
}

static inline int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

/* FAST score of one interior pixel: max over the 16 arcs of 9 contiguous circle pixels of the
 * smallest |difference| on the arc (all brighter, or all darker), minus 1.  That is the largest
 * threshold t for which "9 contiguous pixels all > p + t or all < p - t" still holds. */
static int fast_score_pixel(const uint8_t* im, int pitch, int y, int x) {
  int d[16];
  int p = im[y * pitch + x];
  for (int k = 0; k < 16; ++k)
    d[k] = (int)im[(y + VUS_CIRCLE_DY[k]) * pitch + (x + VUS_CIRCLE_DX[k])] - p;
  int best = 0;
  for (int s = 0; s < 16; ++s) {
    int mn_b = 1 << 20, mn_d = 1 << 20;
    for (int j = 0; j < VUS_FAST_ARC; ++j) {
      int v = d[(s + j) & 15];
      if (v < mn_b) mn_b = v;   /* arc brighter than centre: all d > t  */
      if (-v < mn_d) mn_d = -v; /* arc darker than centre: all -d > t   */
    }
    if (mn_b > best) best = mn_b;
    if (mn_d > best) best = mn_d;
  }
  return best - 1; /* may be -1 when no arc is one-sided */
}

int vus_fast_score_cpu(const uint8_t* img, int n_img, int H, int W, int pitch, int thr,
                       uint8_t* score_out) {
  if (!img || !score_out || n_img < 0 || H < 7 || W < 7 || pitch < W || thr < 1 || thr > 254)
    return VUS_E_INVALID;
  for (int n = 0; n < n_img; ++n) {
    const uint8_t* im = img + (size_t)n * H * pitch;
    uint8_t* sc = score_out + (size_t)n * H * W;
    memset(sc, 0, (size_t)H * W);
    for (int y = 3; y < H - 3; ++y)
      for (int x = 3; x < W - 3; ++x) {
        int s = fast_score_pixel(im, pitch, y, x);
        sc[y * W + x] = (uint8_t)(s >= thr ? s : 0);
      }
  }
  return VUS_OK;
}

int vus_blur7_cpu(const uint8_t* img, int n_img, int H, int W, int pitch, uint8_t* out) {
  if (!img || !out || n_img < 0 || H < 1 || W < 1 || pitch < W) return VUS_E_INVALID;
#pragma omp parallel for schedule(dynamic)
  for (int n = 0; n < n_img; ++n) {
    int32_t* tmp = (int32_t*)malloc((size_t)H * W * sizeof(int32_t));
    const uint8_t* im = img + (size_t)n * H * pitch;
    uint8_t* o = out + (size_t)n * H * W;
    for (int y = 0; y < H; ++y)
      for (int x = 0; x < W; ++x) {
        int32_t acc = 0;
        for (int k = 0; k < 7; ++k) acc += VUS_BLUR_W[k] * im[y * pitch + clampi(x + k - 3, 0, W - 1)];
        tmp[y * W + x] = acc; /* <= 255*256 */
      }
    for (int y = 0; y < H; ++y)
      for (int x = 0; x < W; ++x) {
        int32_t acc = 0;
        for (int k = 0; k < 7; ++k) acc += VUS_BLUR_W[k] * tmp[clampi(y + k - 3, 0, H - 1) * W + x];
        o[y * W + x] = (uint8_t)((acc + 32768) >> 16);
      }
    free(tmp);
  }
  return VUS_OK;
}

/* Candidates come out in raster order here; the GPU appends in any order and only the SET of keys
 * is compared (the following top-K selection sorts them). */
int vus_fast_detect_cpu(const uint8_t* img, int n_img, int H, int W, int pitch, int thr, int border,
                        uint8_t* blur_out, uint32_t* cand_keys, int cand_cap, int* cand_count) {
  if (!img || !cand_keys || !cand_count || cand_cap < 1 || border < 0) return VUS_E_INVALID;
  if ((long long)H * W > (1ll << VUS_KEY_POS_BITS)) return VUS_E_INVALID;
  if (thr < 1 || thr > 254 || H < 7 || W < 7 || pitch < W) return VUS_E_INVALID;
#pragma omp parallel for schedule(dynamic)
  for (int n = 0; n < n_img; ++n) {
    uint8_t* sc = (uint8_t*)malloc((size_t)H * W);
    vus_fast_score_cpu(img + (size_t)n * H * pitch, 1, H, W, pitch, thr, sc);
    int cnt = 0;
    uint32_t* keys = cand_keys + (size_t)n * cand_cap;
    for (int y = border; y < H - border; ++y)
      for (int x = border; x < W - border; ++x) {
        int s = sc[y * W + x];
        if (!s) continue;
        int is_max = 1;
        for (int dy = -1; dy <= 1 && is_max; ++dy)
          for (int dx = -1; dx <= 1; ++dx) {
            if (!dx && !dy) continue;
            int yy = y + dy, xx = x + dx;
            int nb = (yy >= 0 && yy < H && xx >= 0 && xx < W) ? sc[yy * W + xx] : 0;
            if (nb >= s) { is_max = 0; break; }
          }
        if (!is_max) continue;
        if (cnt < cand_cap) keys[cnt] = ((uint32_t)(255 - s) << VUS_KEY_POS_BITS) | (uint32_t)(y * W + x);
        ++cnt;
      }
    cand_count[n] = cnt;
    free(sc);
  }
  if (blur_out) return vus_blur7_cpu(img, n_img, H, W, pitch, blur_out);
  return VUS_OK;
}

/* ---- the adaptive detector (include/vus.h): the plain statement of its three calls ---- */
int vus_fast_threshold_estimate_cpu(const uint8_t* img, int n_img, int H, int W, int pitch, int thr, int border, int max_kp,
                                    int sample_stride, int* hist, int* thr_img) {
  if (!img || !hist || !thr_img || thr < 1 || thr > 254 || border < 0 || max_kp < 1 || sample_stride < 1) return VUS_E_INVALID;
  const int tx = (W + VUS_FAST_TILE_W - 1) / VUS_FAST_TILE_W, ty = (H + VUS_FAST_TILE_H - 1) / VUS_FAST_TILE_H;
  const long long n_tiles = (long long)tx * ty;
  long long n_sampled = (n_tiles - sample_stride / 2 + sample_stride - 1) / sample_stride;
  if (n_sampled < 1) n_sampled = 1;
  const int cap = H * W;                     /* every pixel could be a candidate */
  const int floor = thr > VUS_FAST_SAMPLE_FLOOR ? thr : VUS_FAST_SAMPLE_FLOOR;
  for (int n = 0; n < n_img; ++n) {
    uint32_t* keys = (uint32_t*)malloc(sizeof(uint32_t) * (size_t)cap);
    int cnt = 0;
    int rc = vus_fast_detect_cpu(img + (size_t)n * H * pitch, 1, H, W, pitch, thr, border, NULL, keys, cap, &cnt);
    if (rc) { free(keys); return rc; }
    int* h = hist + 256 * (size_t)n;
    memset(h, 0, sizeof(int) * 256);
    for (int i = 0; i < cnt; ++i) {
      const int pos = (int)(keys[i] & ((1u << VUS_KEY_POS_BITS) - 1u)), sc = 255 - (int)(keys[i] >> VUS_KEY_POS_BITS);
      const int tile = (pos / W / VUS_FAST_TILE_H) * tx + (pos % W) / VUS_FAST_TILE_W;
      /* include/vus.h: only the bins from f = max(thr, VUS_FAST_SAMPLE_FLOOR) on are counted (the survivors at or above f
       * of this detection at thr are the survivors of a detection at f) and only the bins above f can carry the estimate */
      if (sc >= floor && tile >= sample_stride / 2 && (tile - sample_stride / 2) % sample_stride == 0) ++h[sc];
    }
    free(keys);
    long long run = 0;
    int t = 254;
    for (; t > floor; --t) {
      run += h[t];
      if (run * n_tiles * VUS_FAST_MARGIN_DEN >= (long long)max_kp * n_sampled * VUS_FAST_MARGIN_NUM) break;
    }
    thr_img[n] = t > floor ? t : thr;
  }
  return VUS_OK;
}

int vus_fast_detect_adaptive_cpu(const uint8_t* img, int n_img, int H, int W, int pitch, const int* thr_img, int border,
                                 uint8_t* blur_out, uint32_t* cand_keys, int cand_cap, int* cand_count) {
  if (!img || !thr_img || !cand_keys || !cand_count) return VUS_E_INVALID;
  for (int n = 0; n < n_img; ++n) {
    int rc = vus_fast_detect_cpu(img + (size_t)n * H * pitch, 1, H, W, pitch, thr_img[n], border,
                                 blur_out ? blur_out + (size_t)n * H * W : NULL, cand_keys + (size_t)n * cand_cap, cand_cap,
                                 cand_count + n);
    if (rc) return rc;
  }
  return VUS_OK;
}

int vus_fast_detect_retry_cpu(const uint8_t* img, int n_img, int H, int W, int pitch, int thr, const int* thr_img, int max_kp,
                              int border, uint32_t* cand_keys, int cand_cap, int* cand_count, int* retry_list,
                              int* retry_count) {
  if (!img || !thr_img || !cand_keys || !cand_count || !retry_list || !retry_count) return VUS_E_INVALID;
  int m = 0;
  for (int n = 0; n < n_img; ++n)
    if (thr_img[n] > thr && cand_count[n] < max_kp) {
      retry_list[m++] = n;
      int rc = vus_fast_detect_cpu(img + (size_t)n * H * pitch, 1, H, W, pitch, thr, border, NULL,
                                   cand_keys + (size_t)n * cand_cap, cand_cap, cand_count + n);
      if (rc) return rc;
    }
  retry_count[0] = m;
  return VUS_OK;
}

static int cmp_u32(const void* a, const void* b) {
  uint32_t x = *(const uint32_t*)a, y = *(const uint32_t*)b;
  return x < y ? -1 : (x > y ? 1 : 0);
}

int vus_select_topk_cpu(const uint32_t* cand_keys, const int* cand_count, int n_img, int cand_cap,
                        int max_kp, uint32_t* kp_keys, int* kp_count) {
  if (!cand_keys || !cand_count || !kp_keys || !kp_count || max_kp < 1 || cand_cap < 1)
    return VUS_E_INVALID;
  int failed = 0;
#pragma omp parallel for schedule(dynamic)
  for (int n = 0; n < n_img; ++n) {
    int cnt = cand_count[n] < cand_cap ? cand_count[n] : cand_cap;
    uint32_t* tmp = (uint32_t*)malloc((size_t)(cnt > 0 ? cnt : 1) * sizeof(uint32_t));   /* per-image scratch */
    if (!tmp) { failed = 1; continue; }
    memcpy(tmp, cand_keys + (size_t)n * cand_cap, (size_t)cnt * sizeof(uint32_t));
    qsort(tmp, (size_t)cnt, sizeof(uint32_t), cmp_u32);
    int k = cnt < max_kp ? cnt : max_kp;
    uint32_t* o = kp_keys + (size_t)n * max_kp;
    for (int i = 0; i < max_kp; ++i) o[i] = i < k ? tmp[i] : VUS_KEY_INVALID;
    kp_count[n] = k;
    free(tmp);
  }
  return failed ? VUS_E_INVALID : VUS_OK;
}

/* Grid-bucketed selection (include/vus.h: vus_select_grid). */
int vus_select_grid_cpu(const uint32_t* cand_keys, const int* cand_count, int n_img, int cand_cap, int H, int W,
                        int grid_row, int grid_col, int per_cell, int max_kp, uint32_t* kp_keys, int* kp_count) {
  if (!cand_keys || !cand_count || !kp_keys || !kp_count || cand_cap < 1 || H < 1 || W < 1 || grid_row < 1 ||
      grid_col < 1 || per_cell < 1 || max_kp < 1)
    return VUS_E_INVALID;
  for (int n = 0; n < n_img; ++n) {
    const uint32_t* keys = cand_keys + (size_t)n * cand_cap;
    const int cnt = cand_count[n] < cand_cap ? cand_count[n] : cand_cap;
    uint32_t* o = kp_keys + (size_t)n * max_kp;
    int out = 0;
    for (int cell = 0; cell < grid_row * grid_col; ++cell) {
      const int cy = cell / grid_col, cx = cell - cy * grid_col;
      uint32_t last = 0;
      int have_last = 0;
      for (int j = 0; j < per_cell && out < max_kp; ++j) {
        uint32_t best = VUS_KEY_INVALID;
        for (int i = 0; i < cnt; ++i) {
          const uint32_t k = keys[i];
          const int pos = (int)(k & VUS_KEY_POS_MASK), y = pos / W, x = pos - y * W;
          if (y * grid_row / H != cy || x * grid_col / W != cx) continue;
          if (have_last && k <= last) continue;
          if (k < best) best = k;
        }
        if (best == VUS_KEY_INVALID) break;
        o[out++] = best;
        last = best;
        have_last = 1;
      }
    }
    kp_count[n] = out;
    for (int i = out; i < max_kp; ++i) o[i] = VUS_KEY_INVALID;
  }
  return VUS_OK;
}

/* The oracle's OWN steering tables, derived here from first principles with libm and NOT read from the
 * generated header the kernels compile in (include/vus_orb_tables.h: VUS_ANGLE_COS/SIN, VUS_RBRIEF_ROT), so
 * that a wrong generated table cannot pass both sides (tests/test_frontend_oracle.py compares the two):
 *   bin direction k: (cos, sin)(2 pi k / 30) in Q14, rounded to nearest (Rublee et al. 2011, sec. 4.2:
 *   "discretize the angle to increments of 2 pi / 30 (12 degrees)");
 *   rotated test point: (x cos - y sin, x sin + y cos), each rounded half-to-even (rint), of the 256 learned
 *   pairs VUS_RBRIEF_BASE (third-party DATA: the published rBRIEF table). */
static int32_t g_cos_q14[VUS_N_ANGLE_BINS], g_sin_q14[VUS_N_ANGLE_BINS];
static int8_t g_rot[VUS_N_ANGLE_BINS * 256 * 4];
static int g_tables_ready = 0;

static void oracle_tables_init(void) {
  if (g_tables_ready) return;
#pragma omp critical(vus_oracle_tables)
  if (!g_tables_ready) {
    const double two_pi = 6.283185307179586476925286766559;
    for (int k = 0; k < VUS_N_ANGLE_BINS; ++k) {
      const double th = two_pi * (double)k / (double)VUS_N_ANGLE_BINS;
      const double c = cos(th), sn = sin(th);
      g_cos_q14[k] = (int32_t)rint(c * 16384.0);
      g_sin_q14[k] = (int32_t)rint(sn * 16384.0);
      for (int t = 0; t < 256; ++t)
        for (int h = 0; h < 4; h += 2) {
          const double x = (double)VUS_RBRIEF_BASE[4 * t + h], y = (double)VUS_RBRIEF_BASE[4 * t + h + 1];
          g_rot[((size_t)k * 256 + t) * 4 + h] = (int8_t)rint(x * c - y * sn);
          g_rot[((size_t)k * 256 + t) * 4 + h + 1] = (int8_t)rint(x * sn + y * c);
        }
    }
    g_tables_ready = 1;
  }
}

/* which = 0: the oracle's derived tables; 1: the generated header's (what the kernels use).  For the test
 * that compares them.  rot [30*256*4] int8, cosq/sinq [30] int32. */
int vus_oracle_tables_cpu(int which, int8_t* rot, int32_t* cosq, int32_t* sinq) {
  if (!rot || !cosq || !sinq) return VUS_E_INVALID;
  oracle_tables_init();
  memcpy(rot, which ? VUS_RBRIEF_ROT : g_rot, sizeof(g_rot));
  memcpy(cosq, which ? VUS_ANGLE_COS : g_cos_q14, sizeof(g_cos_q14));
  memcpy(sinq, which ? VUS_ANGLE_SIN : g_sin_q14, sizeof(g_sin_q14));
  return VUS_OK;
}

int vus_orient_rbrief_cpu(const uint8_t* img, const uint8_t* blur, int n_img, int H, int W, int pitch,
                          const uint32_t* kp_keys, const int* kp_count, int max_kp,
                          uint64_t* desc_out, uint8_t* angle_out);

/* Twins of vus_orient_order / vus_orient_rbrief_ordered (include/vus.h).  The order is a schedule of the GPU kernel, not
 * arithmetic: any permutation that fixes the unused slots is a valid answer of the first (here: the identity) and the
 * second gives vus_orient_rbrief's outputs whatever the permutation -- after checking that it is one. */
int vus_orient_order_cpu(const uint32_t* kp_keys, const int* kp_count, int n_img, int max_kp, int H, int W, int* order) {
  if (!kp_keys || !kp_count || !order || n_img < 0 || max_kp < 1 || H < 1 || W < 1) return VUS_E_INVALID;
  for (int n = 0; n < n_img; ++n)
    for (int j = 0; j < max_kp; ++j) order[(size_t)n * max_kp + j] = j;
  return VUS_OK;
}

int vus_orient_rbrief_ordered_cpu(const uint8_t* img, const uint8_t* blur, int n_img, int H, int W, int pitch,
                                  const uint32_t* kp_keys, const int* kp_count, int max_kp, const int* order,
                                  uint64_t* desc_out, uint8_t* angle_out) {
  if (!order || !kp_count || max_kp < 1) return VUS_E_INVALID;
  for (int n = 0; n < n_img; ++n) {
    const int cnt = kp_count[n] < max_kp ? kp_count[n] : max_kp;
    uint8_t* seen = (uint8_t*)calloc((size_t)max_kp, 1);
    int ok = seen != NULL;
    for (int j = 0; ok && j < max_kp; ++j) {
      const int o = order[(size_t)n * max_kp + j];
      if (o < 0 || o >= max_kp || seen[o] || (j >= cnt && o != j) || (j < cnt && o >= cnt)) ok = 0;
      else seen[o] = 1;
    }
    free(seen);
    if (!ok) return VUS_E_INVALID;
  }
  return vus_orient_rbrief_cpu(img, blur, n_img, H, W, pitch, kp_keys, kp_count, max_kp, desc_out, angle_out);
}

int vus_orient_rbrief_cpu(const uint8_t* img, const uint8_t* blur, int n_img, int H, int W, int pitch,
                          const uint32_t* kp_keys, const int* kp_count, int max_kp,
                          uint64_t* desc_out, uint8_t* angle_out) {
  if (!img || !blur || !kp_keys || !kp_count || !desc_out || !angle_out) return VUS_E_INVALID;
  oracle_tables_init();
#pragma omp parallel for schedule(dynamic)
  for (int n = 0; n < n_img; ++n) {
    const uint8_t* im = img + (size_t)n * H * pitch;
    const uint8_t* bl = blur + (size_t)n * H * W;
    for (int i = 0; i < max_kp; ++i) {
      uint64_t* d = desc_out + ((size_t)n * max_kp + i) * 4;
      d[0] = d[1] = d[2] = d[3] = 0;
      angle_out[(size_t)n * max_kp + i] = 0;
      if (i >= kp_count[n]) continue;
      uint32_t pos = kp_keys[(size_t)n * max_kp + i] & VUS_KEY_POS_MASK;
      int y = (int)(pos / (uint32_t)W), x = (int)(pos % (uint32_t)W);
      /* intensity centroid on the unsmoothed image */
      int64_t m10 = 0, m01 = 0;
      for (int k = 0; k < VUS_DISC_N; ++k) {
        int v = im[clampi(y + VUS_DISC_DY[k], 0, H - 1) * pitch + clampi(x + VUS_DISC_DX[k], 0, W - 1)];
        m10 += (int64_t)VUS_DISC_DX[k] * v;
        m01 += (int64_t)VUS_DISC_DY[k] * v;
      }
      /* nearest of the 30 bin directions = largest projection; first maximum wins */
      int bin = 0;
      int64_t best = m10 * g_cos_q14[0] + m01 * g_sin_q14[0];
      for (int k = 1; k < VUS_N_ANGLE_BINS; ++k) {
        int64_t pr = m10 * g_cos_q14[k] + m01 * g_sin_q14[k];
        if (pr > best) { best = pr; bin = k; }
      }
      angle_out[(size_t)n * max_kp + i] = (uint8_t)bin;
      const int8_t* pat = g_rot + (size_t)bin * 256 * 4;
      for (int t = 0; t < 256; ++t) {
        int a = bl[clampi(y + pat[4 * t + 1], 0, H - 1) * W + clampi(x + pat[4 * t + 0], 0, W - 1)];
        int b = bl[clampi(y + pat[4 * t + 3], 0, H - 1) * W + clampi(x + pat[4 * t + 2], 0, W - 1)];
        if (a < b) d[t >> 6] |= (uint64_t)1 << (t & 63);
      }
    }
  }
  return VUS_OK;
}

/* Bilinear resize, integer (include/vus.h: vus_resize_bilinear). */
static void resize_coeff(int d, int Ns, int Nd, int* i0, int* i1, int* w) {
  const long long num = (long long)(2 * d + 1) * Ns - Nd, den = 2LL * Nd;
  long long ix = num >= 0 ? num / den : -((-num + den - 1) / den);   /* floor */
  const long long rem = num - ix * den;
  *w = (int)((rem * 2048 + den / 2) / den);
  *i0 = clampi((int)ix, 0, Ns - 1);
  *i1 = clampi((int)ix + 1, 0, Ns - 1);
}

int vus_resize_bilinear_cpu(const uint8_t* src, int n_img, int Hs, int Ws, int pitch_s, uint8_t* dst, int Hd,
                            int Wd, int pitch_d) {
  if (!src || !dst || n_img < 0 || Hs < 1 || Ws < 1 || Hd < 1 || Wd < 1 || pitch_s < Ws || pitch_d < Wd)
    return VUS_E_INVALID;
  for (int n = 0; n < n_img; ++n) {
    const uint8_t* s = src + (size_t)n * Hs * pitch_s;
    uint8_t* d = dst + (size_t)n * Hd * pitch_d;
    for (int y = 0; y < Hd; ++y) {
      int y0, y1, wy;
      resize_coeff(y, Hs, Hd, &y0, &y1, &wy);
      for (int x = 0; x < Wd; ++x) {
        int x0, x1, wx;
        resize_coeff(x, Ws, Wd, &x0, &x1, &wx);
        const int top = (2048 - wx) * s[(size_t)y0 * pitch_s + x0] + wx * s[(size_t)y0 * pitch_s + x1];
        const int bot = (2048 - wx) * s[(size_t)y1 * pitch_s + x0] + wx * s[(size_t)y1 * pitch_s + x1];
        d[(size_t)y * pitch_d + x] = (uint8_t)(((2048 - wy) * top + wy * bot + (1 << 21)) >> 22);
      }
    }
  }
  return VUS_OK;
}

/* Level-major merge of the per-level keypoint lists (include/vus.h: vus_pyramid_append). */
int vus_pyramid_append_cpu(const uint32_t* lvl_keys, const int* lvl_count, const uint64_t* lvl_desc,
                           const uint8_t* lvl_angle, int n_img, int lvl_max_kp, int Hl, int Wl, int level, int H0,
                           int W0, int max_kp, uint32_t* kp_keys, int* kp_count, uint64_t* desc, uint8_t* angle,
                           uint8_t* kp_level, int32_t* kp_xy_q4) {
  if (!lvl_keys || !lvl_count || !lvl_desc || !lvl_angle || !kp_keys || !kp_count || !desc || !angle ||
      Hl < 1 || Wl < 1 || H0 < 1 || W0 < 1 || max_kp < 1 || lvl_max_kp < 1 || level < 0 || level > 255)
    return VUS_E_INVALID;
  for (int n = 0; n < n_img; ++n) {
    const int base = kp_count[n];
    int cnt = lvl_count[n] < lvl_max_kp ? lvl_count[n] : lvl_max_kp;
    if (cnt > max_kp - base) cnt = max_kp - base;
    for (int t = 0; t < cnt; ++t) {
      const uint32_t key = lvl_keys[(size_t)n * lvl_max_kp + t];
      const int pos = (int)(key & VUS_KEY_POS_MASK), y = pos / Wl, x = pos - y * Wl;
      const int xq = (int)((((long long)(2 * x + 1) * 8 * W0 + Wl / 2) / Wl) - 8);
      const int yq = (int)((((long long)(2 * y + 1) * 8 * H0 + Hl / 2) / Hl) - 8);
      const int x0 = clampi((xq + 8) >> 4, 0, W0 - 1), y0 = clampi((yq + 8) >> 4, 0, H0 - 1);
      const size_t o = (size_t)n * max_kp + base + t;
      kp_keys[o] = (key & ~VUS_KEY_POS_MASK) | (uint32_t)(y0 * W0 + x0);
      for (int w = 0; w < 4; ++w) desc[4 * o + w] = lvl_desc[4 * ((size_t)n * lvl_max_kp + t) + w];
      angle[o] = lvl_angle[(size_t)n * lvl_max_kp + t];
      if (kp_level) kp_level[o] = (uint8_t)level;
      if (kp_xy_q4) { kp_xy_q4[2 * o] = xq; kp_xy_q4[2 * o + 1] = yq; }
    }
    kp_count[n] = base + cnt;
  }
  return VUS_OK;
}

int vus_hamming_match_cpu(const uint64_t* desc, const uint32_t* kp_keys, const int* kp_count,
                          int max_kp, int H, int W, const int* q_index, const int* t_index, int n_pairs,
                          int max_dy, int min_disp, int max_disp, int max_dist,
                          int32_t* idx_out, int32_t* dist_out) {
  if (!desc || !kp_keys || !kp_count || !q_index || !t_index || !idx_out || !dist_out || W < 1 || H < 1)
    return VUS_E_INVALID;
#pragma omp parallel for schedule(dynamic)
  for (int p = 0; p < n_pairs; ++p) {
    int qi = q_index[p], ti = t_index[p];
    const uint64_t* dq = desc + (size_t)qi * max_kp * 4;
    const uint64_t* dt = desc + (size_t)ti * max_kp * 4;
    const uint32_t* kq = kp_keys + (size_t)qi * max_kp;
    const uint32_t* kt = kp_keys + (size_t)ti * max_kp;
    for (int i = 0; i < max_kp; ++i) {
      int best = 1 << 20, bidx = -1;
      if (i < kp_count[qi]) {
        uint32_t pq = kq[i] & VUS_KEY_POS_MASK;
        int yq = (int)(pq / (uint32_t)W), xq = (int)(pq % (uint32_t)W);
        for (int j = 0; j < kp_count[ti]; ++j) {
          if (max_dy >= 0) {
            uint32_t pt = kt[j] & VUS_KEY_POS_MASK;
            int yt = (int)(pt / (uint32_t)W), xt = (int)(pt % (uint32_t)W);
            int dy = yq - yt, dx = xq - xt;
            if (dy > max_dy || dy < -max_dy || dx < min_disp || dx > max_disp) continue;
          }
          int dist = 0;
          for (int w = 0; w < 4; ++w) dist += __builtin_popcountll(dq[4 * i + w] ^ dt[4 * j + w]);
          if (dist < best) { best = dist; bidx = j; }
        }
      }
      if (bidx < 0) best = 512;            /* no gated candidate */
      else if (best > max_dist) bidx = -1; /* rejected: the distance is still reported */
      idx_out[(size_t)p * max_kp + i] = bidx;
      dist_out[(size_t)p * max_kp + i] = best;
    }
  }
  return VUS_OK;
}

/* CameraMeasurement emitter: ids along the temporal matches, frame by frame (see include/vus.h). */
/* Mutual-nearest-neighbour filter (include/vus.h: vus_cross_check). */
int vus_cross_check_cpu(const int32_t* idx_fwd, const int32_t* idx_bwd, int n_pairs, int max_kp, int32_t* idx_out) {
  if (!idx_fwd || !idx_bwd || !idx_out || n_pairs < 0 || max_kp < 1) return VUS_E_INVALID;
  for (int p = 0; p < n_pairs; ++p)
    for (int i = 0; i < max_kp; ++i) {
      const int32_t j = idx_fwd[(size_t)p * max_kp + i];
      idx_out[(size_t)p * max_kp + i] = (j >= 0 && j < max_kp && idx_bwd[(size_t)p * max_kp + j] == i) ? j : -1;
    }
  return VUS_OK;
}

int vus_track_ids_cpu(const int32_t* stereo_idx, const int32_t* track_idx, const uint32_t* kp_keys,
                      const int* kp_count, int n_frames, int max_kp, int H, int W, int64_t* ids_out,
                      double* feat_out, int64_t* n_ids_out) {
  if (!stereo_idx || !kp_keys || !kp_count || !ids_out || !feat_out || !n_ids_out || n_frames < 0 || max_kp < 1 ||
      H < 1 || W < 1 || (n_frames > 1 && !track_idx))
    return VUS_E_INVALID;
  int64_t* cur = (int64_t*)malloc(sizeof(int64_t) * (size_t)max_kp);
  int64_t* nxt = (int64_t*)malloc(sizeof(int64_t) * (size_t)max_kp);
  int64_t next_id = 0;
  int n_prev = 0;
  for (int f = 0; f < n_frames; ++f) {
    int nl = kp_count[2 * f] < max_kp ? kp_count[2 * f] : max_kp;
    int nr = kp_count[2 * f + 1] < max_kp ? kp_count[2 * f + 1] : max_kp;
    for (int i = 0; i < max_kp; ++i) nxt[i] = -1;
    if (f > 0)
      for (int ip = 0; ip < n_prev; ++ip) {
        int j = track_idx[(size_t)(f - 1) * max_kp + ip];
        if (j >= 0 && j < nl && cur[ip] >= 0 && nxt[j] < 0) nxt[j] = cur[ip];
      }
    const uint32_t* kl = kp_keys + (size_t)(2 * f) * max_kp;
    const uint32_t* kr = kp_keys + (size_t)(2 * f + 1) * max_kp;
    for (int i = 0; i < max_kp; ++i) {
      int64_t* id = ids_out + (size_t)f * max_kp + i;
      double* ft = feat_out + ((size_t)f * max_kp + i) * 4;
      *id = -1;
      ft[0] = ft[1] = ft[2] = ft[3] = 0.0;
      if (i >= nl) continue;
      int j = stereo_idx[(size_t)f * max_kp + i];
      if (j < 0 || j >= nr) continue;
      if (nxt[i] < 0) nxt[i] = next_id++;
      *id = nxt[i];
      uint32_t pl = kl[i] & VUS_KEY_POS_MASK, pr = kr[j] & VUS_KEY_POS_MASK;
      ft[0] = 2.0 * (double)(pl % (uint32_t)W) / (double)W - 1.0;
      ft[1] = 2.0 * (double)(pl / (uint32_t)W) / (double)H - 1.0;
      ft[2] = 2.0 * (double)(pr % (uint32_t)W) / (double)W - 1.0;
      ft[3] = 2.0 * (double)(pr / (uint32_t)W) / (double)H - 1.0;
    }
    memcpy(cur, nxt, sizeof(int64_t) * (size_t)max_kp);
    n_prev = nl;
  }
  n_ids_out[0] = next_id;
  free(cur); free(nxt);
  return VUS_OK;
}

/* batch.py:152-166, one feature at a time, in double precision, same operation order. */
int vus_triangulate_cpu(const double* feat, int n, const double* cam, const double* Rt, double* out) {
  if (!feat || !cam || !Rt || !out || n < 0) return VUS_E_INVALID;
  double fx = cam[0], fy = cam[1], cx = cam[2], cy = cam[3], baseline = cam[4];
  double res_x = cam[5], res_y = cam[6];
  double f = (fx + fy) / 2.0; /* batch.py:112 */
  for (int i = 0; i < n; ++i) {
    double u0 = feat[4 * i + 0], v0 = feat[4 * i + 1], u1 = feat[4 * i + 2], v1 = feat[4 * i + 3];
    double uL = (u0 + 1) * 0.5 * res_x;              /* :152 */
    double uR = (u1 + 1) * 0.5 * res_x;              /* :153 */
    double v = ((v0 + v1) / 2.0 + 1) * 0.5 * res_y;  /* :154 */
    double d = uR - uL;                              /* :156 (sign as in the reference) */
    double Wd = d / baseline;                        /* :159 */
    double xc = (uL - cx) / Wd;                      /* :160 */
    double yc = (v - cy) / Wd;                       /* :161 */
    double zc = f / Wd;                              /* :162 */
    for (int r = 0; r < 3; ++r)                      /* :166  R @ cam + t */
      out[6 * i + r] = ((Rt[3 * r + 0] * xc + Rt[3 * r + 1] * yc) + Rt[3 * r + 2] * zc) + Rt[9 + r];
    out[6 * i + 3] = uL;
    out[6 * i + 4] = uR;
    out[6 * i + 5] = v;
  }
  return VUS_OK;
}

/* batch_update's get_landmarks per keyframe (batch.py:264-265) + the landmark loop of batch_create (batch.py:295-305),
 * as two plain nested loops in the reference's order: keyframes ascending (none before first_frame: batch.py:280-305
 * has no landmark loop for i == 0), features in message order. */
int vus_emit_stereo_factors_cpu(const int64_t* ids, const double* feat, const double* Rt, const double* cam, int n_frames,
                                int max_kp, int first_frame, long long n_ids, int* frame_base, int* count,
                                int* obs_frame, int64_t* obs_id, double* obs_meas, int64_t* lm_first, double* lm_point) {
  if (!ids || !feat || !Rt || !cam || !frame_base || !count || !obs_frame || !obs_id || !obs_meas || !lm_first ||
      !lm_point || n_frames < 0 || max_kp < 1 || first_frame < 0 || n_ids < 0)
    return VUS_E_INVALID;
  for (long long j = 0; j < n_ids; ++j) lm_first[j] = -1;
  int n = 0;
  for (int f = 0; f < n_frames; ++f) {
    frame_base[f] = n;
    if (f < first_frame) continue;
    for (int i = 0; i < max_kp; ++i) {
      int64_t id = ids[(size_t)f * max_kp + i];
      if (id < 0 || id >= n_ids) continue;
      double tri[6];
      int rc = vus_triangulate_cpu(feat + ((size_t)f * max_kp + i) * 4, 1, cam, Rt + 12 * (size_t)f, tri); /* :265 */
      if (rc) return rc;
      if (lm_first[id] < 0) {                       /* :297 `not initial_estimate.exists(L(id))` */
        lm_first[id] = (int64_t)f * max_kp + i;
        lm_point[3 * id] = tri[0]; lm_point[3 * id + 1] = tri[1]; lm_point[3 * id + 2] = tri[2];   /* :298 */
      }
      obs_frame[n] = f;                             /* :300-305 */
      obs_id[n] = id;
      obs_meas[3 * (size_t)n] = tri[3]; obs_meas[3 * (size_t)n + 1] = tri[4]; obs_meas[3 * (size_t)n + 2] = tri[5];
      ++n;
    }
  }
  frame_base[n_frames] = n;
  count[0] = n;
  return VUS_OK;
}

/* GenericStereoFactor3D's h(X, L) - z at the initial estimate, unwhitened (gtsam StereoCamera::project; the same
 * projection as the BA oracle's stereo factor, vus_oracle_ba.c). */
int vus_stereo_initial_residuals_cpu(const double* Rt, const double* K, const double* lm_point, const int* obs_frame,
                                     const int64_t* obs_id, const double* obs_meas, int n, double* resid) {
  if (n < 0 || (n > 0 && (!Rt || !K || !lm_point || !obs_frame || !obs_id || !obs_meas || !resid))) return VUS_E_INVALID;
  double fx = K[0], fy = K[1], cx = K[3], cy = K[4], b = K[5];
  for (int a = 0; a < n; ++a) {
    const double* T = Rt + 12 * (size_t)obs_frame[a];
    const double* p = lm_point + 3 * (size_t)obs_id[a];
    double d0 = p[0] - T[9], d1 = p[1] - T[10], d2 = p[2] - T[11];
    double x = (T[0] * d0 + T[3] * d1) + T[6] * d2;
    double y = (T[1] * d0 + T[4] * d1) + T[7] * d2;
    double z = (T[2] * d0 + T[5] * d1) + T[8] * d2;
    double* r = resid + 3 * (size_t)a;
    if (!(z > 0.0)) { r[0] = r[1] = r[2] = HUGE_VAL; continue; }
    r[0] = (cx + fx * x / z) - obs_meas[3 * (size_t)a];
    r[1] = (cx + fx * (x - b) / z) - obs_meas[3 * (size_t)a + 1];
    r[2] = (cy + fy * y / z) - obs_meas[3 * (size_t)a + 2];
  }
  return VUS_OK;
}
