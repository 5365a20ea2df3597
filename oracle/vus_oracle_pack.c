/* vus_oracle_pack.c -- CPU twins of the graph-packing entry points of include/vus.h (csrc/pack.hip).
 * TEST INFRASTRUCTURE ONLY.  Plain C with qsort: the definition of what the device kernels must produce -- index
 * plumbing of the reference's factor emission (batch.py:295-305: one factor per observation, keys X(i), L(id)). */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include "../include/vus.h"

typedef struct { uint64_t key; int idx; } kv_t;

static int cmp_kv(const void* a, const void* b) {
  const kv_t* x = (const kv_t*)a; const kv_t* y = (const kv_t*)b;
  if (x->key != y->key) return x->key < y->key ? -1 : 1;
  return x->idx < y->idx ? -1 : (x->idx > y->idx);          /* stable */
}

int vus_keys_to_indices_cpu(const int64_t* keys, int n, int* idx_out, int64_t* uniq_out, int* n_unique, void* work,
                            long long work_bytes) {
  (void)work; (void)work_bytes;
  if (n < 0 || !n_unique || (n > 0 && (!keys || !idx_out || !uniq_out))) return VUS_E_INVALID;
  n_unique[0] = 0;
  if (n == 0) return VUS_OK;
  kv_t* a = (kv_t*)malloc(sizeof(kv_t) * (size_t)n);
  for (int i = 0; i < n; ++i) { a[i].key = (uint64_t)keys[i]; a[i].idx = i; }
  qsort(a, (size_t)n, sizeof(kv_t), cmp_kv);
  int r = -1;
  for (int i = 0; i < n; ++i) {
    if (i == 0 || a[i].key != a[i - 1].key) uniq_out[++r] = (int64_t)a[i].key;
    idx_out[a[i].idx] = r;
  }
  n_unique[0] = r + 1;
  free(a);
  return VUS_OK;
}

int vus_lookup_keys_cpu(const int64_t* sorted_keys, int m, const int64_t* queries, int n, int* idx_out, int* first_miss) {
  if (m < 0 || n < 0 || !first_miss || (n > 0 && (!queries || !idx_out)) || (m > 0 && !sorted_keys)) return VUS_E_INVALID;
  first_miss[0] = 0x7F7F7F7F;
  for (int i = 0; i < n; ++i) {
    int lo = 0, hi = m;
    while (lo < hi) { int mid = (lo + hi) >> 1; if (sorted_keys[mid] < queries[i]) lo = mid + 1; else hi = mid; }
    int hit = lo < m && sorted_keys[lo] == queries[i];
    idx_out[i] = hit ? lo : -1;
    if (!hit && i < first_miss[0]) first_miss[0] = i;
  }
  return VUS_OK;
}

int vus_ba_pack_observations_cpu(const int* obs_pose, const int* obs_point, const double* meas, int n_obs, int n_poses,
                                 int n_points, double* meas_L, int* obs_pose_L, int* obs_point_L, int* point_ptr,
                                 int* obs_ppos, int* pose_ptr, int* pobs_lidx, int* perm, int* flags, int* band, void* work,
                                 long long work_bytes) {
  (void)work; (void)work_bytes;
  if (n_obs < 0 || n_poses < 1 || n_points < 0 || !point_ptr || !pose_ptr || !flags) return VUS_E_INVALID;
  flags[0] = 0;
  if (band) band[0] = 0;
  memset(point_ptr, 0, sizeof(int) * (size_t)(n_points + 1));
  memset(pose_ptr, 0, sizeof(int) * (size_t)(n_poses + 1));
  if (n_obs == 0) return VUS_OK;
  if (!obs_pose || !obs_point || !meas || !meas_L || !obs_pose_L || !obs_point_L || !obs_ppos || !pobs_lidx || !perm)
    return VUS_E_INVALID;
  kv_t* a = (kv_t*)malloc(sizeof(kv_t) * (size_t)n_obs);
  for (int i = 0; i < n_obs; ++i) {
    int p = obs_pose[i], l = obs_point[i];
    if (p < 0 || p >= n_poses || l < 0 || l >= n_points) flags[0] |= 2;
    a[i].key = (uint64_t)(l < 0 ? 0 : l) * (uint64_t)n_poses + (uint64_t)(p < 0 ? 0 : p);
    a[i].idx = i;
  }
  qsort(a, (size_t)n_obs, sizeof(kv_t), cmp_kv);                 /* L-order: by (point, pose) */
  for (int i = 0; i < n_obs; ++i) {
    if (i > 0 && a[i].key == a[i - 1].key) flags[0] |= 1;
    int l = (int)(a[i].key / (uint64_t)n_poses);
    obs_point_L[i] = l;
    obs_pose_L[i] = (int)(a[i].key - (uint64_t)l * (uint64_t)n_poses);
    perm[i] = a[i].idx;
    memcpy(meas_L + 3 * (size_t)i, meas + 3 * (size_t)a[i].idx, 3 * sizeof(double));
    point_ptr[l + 1]++;
    pose_ptr[obs_pose_L[i] + 1]++;
  }
  for (int j = 0; j < n_points; ++j) point_ptr[j + 1] += point_ptr[j];
  for (int i = 0; i < n_poses; ++i) pose_ptr[i + 1] += pose_ptr[i];
  if (band)
    for (int j = 0; j < n_points; ++j)
      if (point_ptr[j + 1] > point_ptr[j]) {
        int span = obs_pose_L[point_ptr[j + 1] - 1] - obs_pose_L[point_ptr[j]];
        if (span > band[0]) band[0] = span;
      }
  /* P-order: L-order rows counting-sorted by pose (stable: points ascend inside a pose) */
  int* fill = (int*)malloc(sizeof(int) * (size_t)n_poses);
  memcpy(fill, pose_ptr, sizeof(int) * (size_t)n_poses);
  for (int i = 0; i < n_obs; ++i) {
    int s = fill[obs_pose_L[i]]++;
    pobs_lidx[s] = i;
    obs_ppos[i] = s;
  }
  free(fill); free(a);
  return VUS_OK;
}

int vus_exclusive_scan_i32_cpu(const int* in, int n, int* out, long long* total) {
  if (n < 0 || !out || !total || (n > 0 && !in)) return VUS_E_INVALID;
  long long run = 0;
  for (int i = 0; i < n; ++i) { out[i] = (int)run; run += in[i]; }
  out[n] = (int)run;
  total[0] = run;
  return VUS_OK;
}

/* ---- tile-pair structure of the landmark elimination (vus_ba_tiles; csrc/pack.hip builds it with two launches and a
 * radix sort): the plain statement -- for every unit in turn, every landmark in turn. */
int vus_ba_tiles_count_cpu(const vus_ba_problem* P, int* lm_entries) {
  if (!P || !lm_entries || P->n_points < 1) return VUS_E_INVALID;
  for (int j = 0; j < P->n_points; ++j) {
    int m = 0, prev = -1;
    for (int a = P->point_ptr[j]; a < P->point_ptr[j + 1]; ++a) {
      const int t = P->obs_pose[a] / 8;
      m += t != prev;
      prev = t;
    }
    lm_entries[j] = m * (m + 1) / 2;
  }
  return VUS_OK;
}

long long vus_ba_tiles_work_bytes_cpu(int n_entries) { (void)n_entries; return 0; }

int vus_ba_tiles_fill_cpu(const vus_ba_problem* P, int band, const int* lm_base, int n_entries, int* unit_ptr, int* entries,
                          int* order, void* work, long long work_bytes) {
  (void)work; (void)work_bytes; (void)lm_base;
  if (!P || band < 0 || n_entries < 0 || !unit_ptr || !order) return VUS_E_INVALID;
  const int n_tiles = (P->n_poses + 7) / 8, dt1 = (band + 7) / 8 + 1, n_units = n_tiles * dt1;
  int* cnt = (int*)calloc((size_t)n_units + 1, sizeof(int));
  /* pass 1 counts, pass 2 places: landmarks ascend inside every unit */
  for (int pass = 0; pass < 2; ++pass) {
    for (int j = 0; j < P->n_points; ++j) {
      const int a0 = P->point_ptr[j], a1 = P->point_ptr[j + 1];
      for (int a = a0; a < a1;) {
        const int tu = P->obs_pose[a] / 8, au = a;
        int mu = 0;
        for (; a < a1 && P->obs_pose[a] / 8 == tu; ++a) mu |= 1 << (P->obs_pose[a] % 8);
        for (int b = a0; b < a;) {
          const int tv = P->obs_pose[b] / 8, bv = b;
          int mv = 0;
          for (; b < a1 && P->obs_pose[b] / 8 == tv; ++b) mv |= 1 << (P->obs_pose[b] % 8);
          const int u = tu * dt1 + (tu - tv);
          if (tu - tv >= dt1) { free(cnt); return VUS_E_INVALID; }
          if (pass == 0) ++cnt[u];
          else {
            int* e = entries + 4 * (size_t)cnt[u]++;
            e[0] = au; e[1] = bv; e[2] = j; e[3] = mu | (mv << 8);
          }
        }
      }
    }
    if (pass == 0) {
      int run = 0;
      for (int u = 0; u <= n_units; ++u) { const int c = u < n_units ? cnt[u] : 0; unit_ptr[u] = run; cnt[u] = run; run += c; }
      if (run != n_entries) { free(cnt); return VUS_E_INVALID; }
    }
  }
  free(cnt);
  /* order: any permutation with non-increasing size CLASS (bit length of the count) is valid; here: stable by class */
  int pos = 0;
  for (int cls = 32; cls >= 0; --cls)
    for (int u = 0; u < n_units; ++u) {
      const int c = unit_ptr[u + 1] - unit_ptr[u];
      int bl = 0;
      for (int v = c; v > 0; v >>= 1) ++bl;
      if (bl == cls) order[pos++] = u;
    }
  return VUS_OK;
}
