// band_index.h -- every address the band-solve kernels of ba.hip form inside the block band `Sband`, as plain
// __host__ __device__ functions of (geometry, thread), so that the SAME arithmetic the kernels run can be swept on the
// CPU over every (n, band, panel, thread) and checked against the buffer's bounds (tests/test_band_index.py compiles
// this header with g++ under AddressSanitizer/UBSan; the GPU pool offers no device sanitizer).
//
// Layout: Sband[n][band + 1][36]; entry (i, s) is the 6x6 block (i, i - s), row-major.  All functions return an offset
// in DOUBLES from Sband, or -1 where the thread is masked (the element is not stored: above the diagonal, left of the
// band, past the last pose).  A masked thread of a kernel that issues its loads unconditionally reads `safe` instead,
// which the corresponding *_safe function returns and which must itself lie inside the buffer.
#pragma once

#if defined(__HIPCC__)
#define VUS_HD __host__ __device__ __forceinline__
#else
#define VUS_HD inline
#endif

namespace bandidx {

constexpr int PB = 8;            // poses per panel
constexpr int NB = 6 * PB;       // scalar columns per panel
constexpr int UTP = 8;           // poses per update tile

VUS_HD long long band_doubles(int n, int band) { return 36ll * n * (band + 1); }

// block (i, k), k <= i <= k + band
VUS_HD long long blk(int band, int i, int k) { return 36ll * ((long long)i * (band + 1) + (i - k)); }

// ---- panel_factor: lane R = scalar row of the panel, kb = 6-column block; 6 contiguous doubles --------------------
VUS_HD bool panel_row_ok(int band, int nb, int R, int kb) {
  const int ii = R / 6;
  return R < nb && kb <= ii && ii - kb <= band;
}
VUS_HD long long panel_row(int band, int k0, int nb, int R, int kb) {
  const int ii = R / 6, rr = R - 6 * ii;
  if (!panel_row_ok(band, nb, R, kb)) return -1;
  return blk(band, k0 + ii, k0 + kb) + 6 * rr;
}

// ---- stage_and_solve: item = (tile, scalar row lr, panel pose kk); 6 contiguous doubles ---------------------------
// tile 0 = the panel's diagonal block, tiles 1 / 2 = 48-row tiles starting at poses pose0_a / pose0_b.
VUS_HD long long stage_item(int band, int k0, int pb, int i_last, int pose0_a, int pose0_b, int n_tiles, int item) {
  constexpr int ITEMS = 6 * UTP * PB;
  const int tile = (item >= ITEMS) + (item >= 2 * ITEMS);
  const int e = item - ITEMS * tile;
  const int lr = e >> 3, kk = e & 7;
  const int ii = lr / 6, rr = lr - 6 * ii;
  const int pose = (tile == 0 ? k0 : tile == 1 ? pose0_a : pose0_b) + ii;
  bool have = item < (1 + n_tiles) * ITEMS && kk < pb;
  if (tile == 0) have = have && ii < pb && kk <= ii && ii - kk <= band;
  else have = have && pose <= i_last && kk >= pose - band - k0;
  return have ? blk(band, pose, k0 + kk) + 6 * rr : -1;
}

// ---- solved rows: element (pose p0 + ii, panel pose kk, column c) of a 48-row tile; 6 contiguous doubles ----------
// (write-back of chol_trsm / chol_trsm_update and stage_solved_tile of chol_syrk: blocks stored TRANSPOSED)
VUS_HD long long solved_item(int band, int k0, int pb, int i_last, int pose0, int item) {
  const int ii = item / (6 * PB), rem = item - 6 * PB * ii;
  const int kk = rem / 6, c = rem - 6 * kk;
  const int pose = pose0 + ii;
  const bool have = pose <= i_last && kk < pb && kk >= pose - band - k0;
  return have ? blk(band, pose, k0 + kk) + 6 * c : -1;
}

// ---- chol_syrk: 16-byte vector v of the 48x48 update tile (pose rows pi0.., pose columns pj0..) -------------------
// For pose row i the eight blocks (i, pj0 .. pj0 + 7) are contiguous: 288 doubles starting at block (i, pj0 + 7).
// mask bit 0 / 1: element 0 / 1 of the vector belongs to the stored lower part.
VUS_HD long long tile_vec(int band, int i_last, int pi0, int pj0, int v, unsigned& mask) {
  const int ii = v / 144, w = v - 144 * ii;
  const int o = w / 18, e = 2 * (w - 18 * o);
  const int i = pi0 + ii, j = pj0 + 7 - o;
  const int rr = e / 6, c = e - 6 * rr;
  mask = 0;
  if (v < UTP * 144 && i <= i_last && j <= i) mask = (j < i) ? 3u : ((c <= rr ? 1u : 0u) | (c + 1 <= rr ? 2u : 0u));
  return mask ? blk(band, i, j) + e : -1;
}

// ---- chol_trsm_update: scalar (row Rr of tile rows pi0.., column Cc of tile columns pj0..) -------------------------
VUS_HD long long tile_scalar(int band, int i_last, int pi0, int pj0, int Rr, int Cc) {
  const int i = pi0 + Rr / 6, rm = Rr % 6;
  const int j = pj0 + Cc / 6, cm = Cc % 6;
  const bool ok = i <= i_last && j <= i && (j < i || cm <= rm);
  return ok ? blk(band, i, j) + 6 * rm + cm : -1;
}

// ---- chol_window: scalar (row Rr of the 48-row tile starting at pose pi0, column Cc of the tile starting at pose pj0)
// of a tile ANYWHERE in the lower band (the tile may straddle the band's edge or the matrix's end): stored iff
// j <= i <= min(n - 1, j + band), lower triangle on the diagonal.
VUS_HD long long win_scalar(int band, int n, int pi0, int pj0, int Rr, int Cc) {
  const int i = pi0 + Rr / 6, rm = Rr % 6;
  const int j = pj0 + Cc / 6, cm = Cc % 6;
  const bool ok = i < n && j <= i && i - j <= band && (j < i || cm <= rm);
  return ok ? blk(band, i, j) + 6 * rm + cm : -1;
}

// the same tile in ADDRESS ORDER: pose row i of the tile is the 2304 contiguous bytes of the blocks (i, pj0 + 7 .. pj0),
// 144 vectors of 16 bytes; vector v = 144 ii + w holds the elements e, e + 1 (e = 2 (w mod 18)) of block
// (pi0 + ii, pj0 + 7 - w / 18).  This is how chol_window_kernel moves tiles: consecutive lanes, consecutive 16 bytes.
// A block of SOLVED rows (stored transposed) has the same addresses: vector v of the rows pose0.. against the panel at
// k0 is win_vec(band, n, pose0, k0, v), its elements e, e + 1 being (column e / 6, rows e % 6, e % 6 + 1) of the block.
constexpr int WIN_VECS = 8 * 144;
VUS_HD void win_vec_pos(int v, int& ii, int& kk, int& e) {
  ii = v / 144;
  const int w = v - 144 * ii, o = w / 18;
  kk = 7 - o;
  e = 2 * (w - 18 * o);
}
VUS_HD long long win_vec(int band, int n, int pi0, int pj0, int v) {
  int ii, kk, e;
  win_vec_pos(v, ii, kk, e);
  const int i = pi0 + ii, j = pj0 + kk;
  const bool ok = v < WIN_VECS && i < n && j <= i && i - j <= band;
  return ok ? blk(band, i, j) + e : -1;
}
// The form the kernel computes: win_vec = win_vec_base(tile) + win_vec_rel(thread's vector), valid iff win_vec_ok; the
// thread keeps rel, ii and dk = kk - ii for the whole launch (checked against win_vec by the CPU sweep).
VUS_HD long long win_vec_base(int band, int pi0, int pj0) { return 36ll * ((long long)pi0 * (band + 1) + (pi0 - pj0 - 7)); }
VUS_HD long long win_vec_rel(int band, int v) {
  const int ii = v / 144;
  return 36ll * ii * (band + 2) + 2 * (v - 144 * ii);
}
VUS_HD bool win_vec_ok(int band, int n, int pi0, int pj0, int ii, int dk) {
  const int dd = pi0 - pj0;
  return pi0 + ii < n && dk <= dd && -dk <= band - dd;     // i < n, j <= i, i - j <= band
}

// ---- chol_window: which tile a window slot hosts at panel step p ----------------------------------------------------
// Tiles are 8 x 8 poses; tile (I, J), J <= I <= J + D, is LIVE during the panel steps [I - D, J] (its first update to its
// own elimination).  With M = D + 1, tile (I, J) is hosted by the slot of the unordered pair {I mod M, J mod M}: the two
// orientations of a pair have complementary lifetimes (M - d and d steps of every M), so a slot hosts exactly one tile at
// every step -- M (M + 1) / 2 slots for the whole sliding window, every one busy at every step.
VUS_HD void win_slot_pair(int slot, int& hi, int& lo) {
  hi = 0;
  while ((hi + 1) * (hi + 2) / 2 <= slot) ++hi;
  lo = slot - hi * (hi + 1) / 2;
}
VUS_HD void win_tile_of(int hi, int lo, int M, int p, int& I, int& J) {
  const int d1 = hi - lo;
  int r = (lo - p) % M;
  if (r < 0) r += M;
  if (r <= M - d1 - 1) {
    J = p + r;
    I = J + d1;
  } else {
    J = p + r + d1 - M;
    I = J + (M - d1);
  }
}

// ---- diag_invert: slot t of the lower block triangle of a panel (block row r6, distance sd, element e) ------------
constexpr int DIAG_ELEMS = 36 * (PB * (PB + 1) / 2);
VUS_HD void diag_slot(int t, int& r6, int& sd, int& e) {
  r6 = 0;
  for (int q = 1; q < PB; ++q) r6 += t >= 36 * (q * (q + 1) / 2);
  const int u = t - 36 * (r6 * (r6 + 1) / 2);
  sd = u / 36;
  e = u - 36 * sd;
}
VUS_HD long long diag_elem(int band, int k0, int nb, int t) {
  int r6, sd, e;
  diag_slot(t < DIAG_ELEMS ? t : 0, r6, sd, e);
  const bool h = t < DIAG_ELEMS && 6 * r6 < nb && sd <= band;
  return h ? 36ll * ((long long)(k0 + r6) * (band + 1) + sd) + e : -1;
}

// ---- chol_backsolve --------------------------------------------------------------------------------------------------
// cb_load_diag: element (row c, column lane) of the panel's diagonal block (substitution path, any band)
VUS_HD long long cb_diag(int band, int k0, int nb, int lane, int c) {
  const int lr6 = lane / 6, lrm = lane - 6 * lr6;
  const bool have = c < nb && lane < c && c / 6 - lr6 <= band;
  return have ? blk(band, k0 + c / 6, k0 + lr6) + 6 * (c % 6) + lrm : -1;
}
VUS_HD long long cb_diag_pivot(int band, int k0, int nb, int lane) {
  const int lr6 = lane / 6, lrm = lane - 6 * lr6;
  return lane < nb ? blk(band, k0 + lr6, k0 + lr6) + 7 * lrm : -1;
}
// cb_load_inv: element (row r, column lane) of the INVERTED diagonal panel (band >= PB - 1 only); masked lanes read
// cb_inv_safe = the panel's first element
VUS_HD long long cb_inv_safe(int band, int k0) { return 36ll * k0 * (band + 1); }
// = cb_inv_base (per lane: element (row 0 of the panel, column lane), which for lane >= 6 lies in front of the panel and
// is only ever used together with a row delta that brings it back inside) + cb_inv_delta (per row, the same for every
// lane: the kernel keeps ONE per-lane pointer and adds wave-uniform offsets -- 48 loads, no per-element 64-bit address
// arithmetic on the sweep's critical path)
VUS_HD long long cb_inv_base(int band, int k0, int lane) {
  const int ln = lane < NB ? lane : 0;
  const int c6 = ln / 6, cm = ln - 6 * c6;
  return 36ll * k0 * (band + 1) + (cm - 36 * c6);
}
VUS_HD long long cb_inv_delta(int band, int r) { return (long long)(r / 6) * (36ll * (band + 2)) + 6 * (r % 6); }
VUS_HD bool cb_inv_stored(int nb, int lane, int r) { return r < nb && lane <= r; }
VUS_HD long long cb_inv(int band, int k0, int nb, int lane, int r) {
  return cb_inv_stored(nb, lane, r) ? cb_inv_base(band, k0, lane) + cb_inv_delta(band, r) : -1;
}
// row c of the transposed block (8 P + kk, 8 P - 8 G - 8 + a): 6 contiguous doubles
VUS_HD long long cb_rows(int band, int n_poses, int P, int G, int kk, int a, int c) {
  const int k0 = PB * P, i = k0 - PB * G - PB + a;
  const bool have = kk < PB && i >= 0 && k0 + kk < n_poses && k0 + kk - i <= band;
  return have ? blk(band, k0 + kk, i) + 6 * c : -1;
}

}  // namespace bandidx
