// frontend.hip -- stereo ORB front-end for gfx950 (MI355X): FAST-9/16 + NMS + smoothing, top-K
// selection, intensity-centroid orientation + rotated BRIEF, brute-force Hamming matching, and the
// get_landmarks triangulation of the reference (batch.py:144-176).
//
// Replaces the external `gtsam_vio/ImageProcessorNodelet` the reference wires in at
// launch/stereo.launch:33-55 (its output is what batch.py:149-154 consumes).  Algorithm
// definitions (integer, bit-exact) are stated in include/vus.h; the CPU oracle under oracle/
// restates them independently for the parity tests.
//
// Design notes (MI355X):
//  * everything is batched over images: grid.z / grid.y = image, so one launch covers the whole
//    resident stream and fills the 256 CUs;
//  * fast_detect stages a 144x32 byte tile (128x24 outputs + halo) in LDS once, as dwords, and
//    derives score, non-max suppression and the 7x7 smoothing from it: the image is read from HBM
//    once; the score map never goes to HBM (candidates leave the CU as 4-byte keys);
//  * every thread works on strips of 4 adjacent pixels read with ds_read_b32/b64 and unpacked in
//    registers (SDWA byte selects): byte-wide LDS reads made the first version LDS-issue bound;
//  * scoring is two-pass: a 10-op necessary test (compass points N/S/E/W) on every pixel, survivors
//    compacted into an LDS work list (DPP wave scan), then the exact score with v_min3/v_max3
//    sliding windows on dense lanes; the smoothing uses v_dot4_u32_u8 on v_alignbyte windows.
//    The integer min/max/SDWA ops issue at ~0.57x the fp32 rate on gfx950 (tools/ubench): the
//    kernel is VALU-issue bound, so the lever is instruction count, not bytes;
//  * select_topk is an exact 4x8-bit MSB radix select + LDS bitonic sort, one workgroup per image;
//  * orient_rbrief uses one 64-lane wave per keypoint: lane-strided disc moments, integer bin
//    choice, and the descriptor words come straight out of __ballot (lane = test bit);
//  * hamming_match keeps the train descriptors in LDS (broadcast reads) and one query per lane.
#include "vus_common.h"
#define VUS_TABLE_QUAL __device__ constexpr
#include "../../include/vus_orb_tables.h"

namespace {

#ifndef VUS_TW
#define VUS_TW 128
#endif
#ifndef VUS_TH
#define VUS_TH 24
#endif
constexpr int TW = VUS_TW;              // output tile width  (1280 = 10 tiles of 128)
constexpr int TH = VUS_TH;              // output tile height (720 = 30 tiles of 24)
#ifndef VUS_AB_TILE   // tools/ab experiments only
static_assert(TW == VUS_FAST_TILE_W && TH == VUS_FAST_TILE_H, "the sampling pattern of vus_fast_threshold_estimate is part of the ABI");
#endif
#ifndef VUS_FAST_WPE
#define VUS_FAST_WPE 1    // second __launch_bounds__ argument of the tile kernels: waves per SIMD the compiler must allow
#endif
#ifndef VUS_NT
#define VUS_NT 256
#endif
constexpr int NTHREADS = VUS_NT;
// LDS images are arrays of dwords = 4 horizontally adjacent pixels ("strips"); every phase reads
// ds_read_b32/b64 and unpacks bytes in registers (byte-wide LDS reads cost ~3x the LDS cycles).
constexpr int IMG_ROWS = TH + 8;        // image rows  y0-4 .. y0+TH+3
constexpr int IMG_DW = (TW + 16) / 4;   // image cols  x0-8 .. x0+TW+7
constexpr int SC_ROWS = TH + 2;         // score rows  y0-1 .. y0+TH
constexpr int SC_DW = (TW + 8) / 4;     // score cols  x0-4 .. x0+TW+3
constexpr int H_ROWS = TH + 6;          // horizontally smoothed rows y0-3 .. y0+TH+2
constexpr int H_DW = TW / 2;            // two u16 per dword
constexpr int STRIPS = TW / 4;

__device__ __forceinline__ int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }
__device__ __forceinline__ int min3i(int a, int b, int c) { return min(min(a, b), c); }
__device__ __forceinline__ int max3i(int a, int b, int c) { return max(max(a, b), c); }
__device__ __forceinline__ int byte_of(uint32_t w, int i) { return (int)((w >> (8 * i)) & 0xFFu); }

// FAST-9/16 score of pixel e (0..3) of a strip.  r[row][j]: 7 image rows x 3 dwords; the strip's
// pixels are bytes 4..7 of each 12-byte row window.  Sliding-window min / max of the 16 circle
// differences over every arc of 9:  w3[k] = op(d[k..k+2]),  w9[k] = op(w3[k], w3[k+3], w3[k+6]).
// Bright arcs need min(d) large, dark arcs need max(d) small (very negative).
template <int E>
__device__ __forceinline__ int fast_score_strip(const uint32_t (&r)[7][3]) {
  constexpr int DX[16] = {0, 1, 2, 3, 3, 3, 2, 1, 0, -1, -2, -3, -3, -3, -2, -1};
  constexpr int DY[16] = {-3, -3, -2, -1, 0, 1, 2, 3, 3, 3, 2, 1, 0, -1, -2, -3};
  const int p = byte_of(r[3][1], E);
  int d[16];
#pragma unroll
  for (int k = 0; k < 16; ++k) {
    const int pos = 4 + E + DX[k];
    d[k] = byte_of(r[3 + DY[k]][pos >> 2], pos & 3) - p;
  }
  int mn3[16], mx3[16];
#pragma unroll
  for (int k = 0; k < 16; ++k) {
    mn3[k] = min3i(d[k], d[(k + 1) & 15], d[(k + 2) & 15]);
    mx3[k] = max3i(d[k], d[(k + 1) & 15], d[(k + 2) & 15]);
  }
  int best_bright = -(1 << 20), best_dark = 1 << 20;
#pragma unroll
  for (int k = 0; k < 16; ++k) {
    best_bright = max(best_bright, min3i(mn3[k], mn3[(k + 3) & 15], mn3[(k + 6) & 15]));
    best_dark = min(best_dark, max3i(mx3[k], mx3[(k + 3) & 15], mx3[(k + 6) & 15]));
  }
  return max(best_bright, -best_dark) - 1;
}

// The same score from the sixteen circle bytes read as BYTES around the pixel (c = its address in the staged tile, rb = bytes
// per staged row): 17 ds_read_u8 instead of 21 ds_read_b32 + 21 v_alignbyte that bring the pixel to a fixed byte first.
__device__ __forceinline__ int fast_score_bytes(const uint8_t* c, int rb) {
  constexpr int DX[16] = {0, 1, 2, 3, 3, 3, 2, 1, 0, -1, -2, -3, -3, -3, -2, -1};
  constexpr int DY[16] = {-3, -3, -2, -1, 0, 1, 2, 3, 3, 3, 2, 1, 0, -1, -2, -3};
  const int p = c[0];
  int d[16];
#pragma unroll
  for (int k = 0; k < 16; ++k) d[k] = (int)c[DY[k] * rb + DX[k]] - p;
  int mn3[16], mx3[16];
#pragma unroll
  for (int k = 0; k < 16; ++k) {
    mn3[k] = min3i(d[k], d[(k + 1) & 15], d[(k + 2) & 15]);
    mx3[k] = max3i(d[k], d[(k + 1) & 15], d[(k + 2) & 15]);
  }
  int best_bright = -(1 << 20), best_dark = 1 << 20;
#pragma unroll
  for (int k = 0; k < 16; ++k) {
    best_bright = max(best_bright, min3i(mn3[k], mn3[(k + 3) & 15], mn3[(k + 6) & 15]));
    best_dark = min(best_dark, max3i(mx3[k], mx3[(k + 3) & 15], mx3[(k + 6) & 15]));
  }
  return max(best_bright, -best_dark) - 1;
}

// The same score on PACKED 16-bit pairs (round 3).  Lane pair (b, 255 - b) of every circle pixel --
// one v_perm_b32 from the row dword and its complement -- so that ONE v_pk_min_i16 chain serves bright and dark arcs:
//   min over an arc of b        = p + (bright arc's min d),      min over an arc of (255 - b) = 255 - max b,
// and the windows double instead of tripling: w2, w4, w8, then w9 = min(w8[k], v[k+8]) -- 4 packed ops per position
// instead of 2 x (min3 + min3/max) on scalars.  score = max(max_k w9.lo - p, max_k w9.hi - (255 - p)) - 1: identical.
typedef short short2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ int fast_score_pk(const uint32_t (&r)[7][3]) {     // pixel = byte 4 of the 12-byte windows
  constexpr int DX[16] = {0, 1, 2, 3, 3, 3, 2, 1, 0, -1, -2, -3, -3, -3, -2, -1};
  constexpr int DY[16] = {-3, -3, -2, -1, 0, 1, 2, 3, 3, 3, 2, 1, 0, -1, -2, -3};
  const int p = byte_of(r[3][1], 0);
  uint32_t nr[7][2];
#pragma unroll
  for (int k = 0; k < 7; ++k) {
    nr[k][0] = ~r[k][0];
    nr[k][1] = ~r[k][1];
  }
  short2_t v[16];
#pragma unroll
  for (int k = 0; k < 16; ++k) {
    const int pos = 4 + DX[k], row = 3 + DY[k], dw = pos >> 2, by = pos & 3;
    // v_perm_b32: selector bytes 0-3 address src1 (the row), 4-7 src0 (its complement), 0x0c = constant zero
    const uint32_t sel = (uint32_t)by | (0x0cu << 8) | ((uint32_t)(4 + by) << 16) | (0x0cu << 24);
    v[k] = __builtin_bit_cast(short2_t, __builtin_amdgcn_perm(nr[row][dw], r[row][dw], sel));
  }
  short2_t w2[16], w4[16], w8[16];
#pragma unroll
  for (int k = 0; k < 16; ++k) w2[k] = __builtin_elementwise_min(v[k], v[(k + 1) & 15]);
#pragma unroll
  for (int k = 0; k < 16; ++k) w4[k] = __builtin_elementwise_min(w2[k], w2[(k + 2) & 15]);
#pragma unroll
  for (int k = 0; k < 16; ++k) w8[k] = __builtin_elementwise_min(w4[k], w4[(k + 4) & 15]);
  short2_t best = __builtin_elementwise_min(w8[0], v[8]);
#pragma unroll
  for (int k = 1; k < 16; ++k) best = __builtin_elementwise_max(best, __builtin_elementwise_min(w8[k], v[(k + 8) & 15]));
  return max((int)best.x - p, (int)best.y - (255 - p)) - 1;
}

// Measured (MI355X, configs[1], A/B builds in one gpurun call, twice): fast_detect 6.23 / 6.25 ms packed against
// 6.17 / 6.17 ms with the scalar v_min3 / v_max3 windows of fast_score_strip -- 95 packed instructions (64 v_pk_min_i16,
// 15 v_pk_max_i16, 16 v_perm_b32) + 17 v_not do not issue faster than the ~160 scalar ones they replace.  Bit-exact
// either way (tests/test_frontend_gpu.py); the scalar form stays the default.
#ifndef VUS_FAST_PK
#define VUS_FAST_PK 0
#endif

template <int E>
__device__ __forceinline__ bool nms_keep(const uint32_t (&c)[3][3]) {
  // pixel E of the strip = byte 4+E of the 12-byte windows; strict maximum of its 8 neighbours
  const int s = byte_of(c[1][1], E);
  int m = 0;
#pragma unroll
  for (int dy = 0; dy < 3; ++dy)
#pragma unroll
    for (int dx = -1; dx <= 1; ++dx) {
      if (dy == 1 && dx == 0) continue;
      const int pos = 4 + E + dx;
      m = max(m, byte_of(c[dy][pos >> 2], pos & 3));
    }
  return s > m;
}

#ifndef VUS_FAST_STRIP
#define VUS_FAST_STRIP 1  // strip-level pre-test from per-dword extrema before the per-pixel one (see pass 1a)
#endif
#ifndef VUS_BLUR_MFMA
#define VUS_BLUR_MFMA 1   // the 7x7 smoothing as two banded int8 GEMMs on the matrix cores (see blur_tile_mfma)
#endif
#ifndef VUS_FAST_LATE_BLUR
#define VUS_FAST_LATE_BLUR 1
#endif
constexpr int VUS_CAND_REGIONS = 8;
#ifndef VUS_FAST_SCORE_BYTES
#define VUS_FAST_SCORE_BYTES 1   // exact score from 17 byte reads around the pixel (2.92 -> 2.89 ms per 1000 stereo frames)
#endif
#ifndef VUS_FAST_MM_BYTES
#define VUS_FAST_MM_BYTES 1   // pass 1a reads the (min, max) pairs as bytes: 3.01 -> 2.96 ms per 1000 stereo frames
#endif
#ifndef VUS_FAST_DIAG
#define VUS_FAST_DIAG 1   // pre-test also on the two diagonal opposite pairs: survivors 33 % -> 25 %, 6.32 -> 6.23 ms
#endif
// ---- the 7 x 7 smoothing of a staged tile on the matrix cores (round 4).  Both passes of the separable filter are
// products with a banded (Toeplitz) weight matrix, and v_mfma_i32_16x16x32_i8 computes a 16 x 16 block of such a
// product over a K window of 32 -- the 22 inputs a 16-wide block needs fit.  Exact integer arithmetic, same result as
// the VALU form (u16 row sums, then sum w H + 32768 >> 16):
//   H pass   C[row][col] = sum_k img[row][k] T[k][col]      A = 8 consecutive image bytes of a row (one ds_read_b64),
//            B = the weights, a per-lane constant.  The bytes are unsigned and the instruction is signed: a ^ 0x80 =
//            a - 128, and 128 * sum(w) = 32768 goes into the accumulator's initial value.
//   V pass   out^T[col][row] = sum_k H^T[col][k] Tv[k][row], with H split into its low and high bytes (two products,
//            recombined as (hi << 8) + lo).  K is only a summation index, so its order is chosen to be the one the
//            H pass leaves in the registers: lane (col, g) holds rows 4g..4g+3 of both 16-row blocks = its 8 K slots;
//            the weight operand is laid out to match.  No LDS round trip between the passes, and the result arrives as
//            4 consecutive pixels of one row per lane = one dword store.
// Cost per tile: 48 MFMAs and ~300 VALU wave-instructions against ~980 for the VALU form (two passes through LDS).
struct BlurMfmaTable {
  uint64_t h[64];      // H pass B operand: byte s of lane (g, n) = w[8 g + s - n - 5]
  uint64_t v[2][64];   // V pass B operand of output rows 16 nb + (lane & 15)
};
constexpr BlurMfmaTable make_blur_mfma_table() {
  constexpr int BW[7] = {18, 33, 49, 56, 49, 33, 18};
  BlurMfmaTable t{};
  for (int l = 0; l < 64; ++l) {
    const int g = l >> 4, m = l & 15;
    for (int sl = 0; sl < 8; ++sl) {
      const int d = 8 * g + sl - m - 5;     // image column 16 j + k feeds output column 16 j + n + 8 - 3 + tap
      if (d >= 0 && d <= 6) t.h[l] |= (uint64_t)BW[d] << (8 * sl);
      const int hrow = sl < 4 ? 4 * g + sl : 16 + 4 * g + (sl - 4);
      for (int nb = 0; nb < 2; ++nb) {
        const int dv = hrow - (16 * nb + m + 1);   // output row ly reads staged rows ly + 1 .. ly + 7
        if (dv >= 0 && dv <= 6) t.v[nb][l] |= (uint64_t)BW[dv] << (8 * sl);
      }
    }
  }
  return t;
}
__device__ __attribute__((aligned(16))) const BlurMfmaTable g_blur_mfma_table = make_blur_mfma_table();
typedef int v4i32_t __attribute__((ext_vector_type(4)));

// The tile's eight 16-column blocks go to the waves j_first, j_first + j_step, ... (all four waves: tid >> 6, 4; the
// late form below: three waves, the fourth waits for the tile's global atomic meanwhile).
__device__ __forceinline__ void blur_tile_mfma(const uint32_t* s_img, uint8_t* __restrict__ blur_out, int n, int H, int W,
                                               int x0, int y0, int tid, int j_first, int j_step) {
  static_assert(IMG_ROWS == 32 && TW % 16 == 0 && (IMG_DW * 4) % 8 == 0, "blur_tile_mfma: K = the 32 staged rows");
  const int l = tid & 63, g = l >> 4, m = l & 15;
  const long bh = (long)g_blur_mfma_table.h[l];
  const long bv[2] = {(long)g_blur_mfma_table.v[0][l], (long)g_blur_mfma_table.v[1][l]};
  const uint8_t* img8 = reinterpret_cast<const uint8_t*>(s_img);
  constexpr unsigned long long SIGN = 0x8080808080808080ull;
#ifndef VUS_BLUR_UNROLL
#define VUS_BLUR_UNROLL 1
#endif
#pragma unroll VUS_BLUR_UNROLL
  for (int j = j_first; j < TW / 16; j += j_step) {
    v4i32_t ch[2];
#pragma unroll
    for (int mb = 0; mb < 2; ++mb) {
      const unsigned long long a =
          *reinterpret_cast<const unsigned long long*>(img8 + (16 * mb + m) * (IMG_DW * 4) + 16 * j + 8 * g) ^ SIGN;
      const v4i32_t c0 = {32768, 32768, 32768, 32768};
      ch[mb] = __builtin_amdgcn_mfma_i32_16x16x32_i8((long)a, bh, c0, 0, 0, 0);
    }
    // the 8 row sums of this lane (0 .. 65280) -> their low bytes and their high bytes, in K-slot order
    uint32_t lo[2], hi[2];
#pragma unroll
    for (int mb = 0; mb < 2; ++mb) {
      const uint32_t t0 = __builtin_amdgcn_perm((uint32_t)ch[mb][1], (uint32_t)ch[mb][0], 0x05010400u);
      const uint32_t t1 = __builtin_amdgcn_perm((uint32_t)ch[mb][3], (uint32_t)ch[mb][2], 0x05010400u);
      lo[mb] = __builtin_amdgcn_perm(t1, t0, 0x05040100u);
      hi[mb] = __builtin_amdgcn_perm(t1, t0, 0x07060302u);
    }
    const long a_lo = (long)((((unsigned long long)lo[1] << 32) | lo[0]) ^ SIGN);
    const long a_hi = (long)((((unsigned long long)hi[1] << 32) | hi[0]) ^ SIGN);
#pragma unroll
    for (int nb = 0; nb < 2; ++nb) {
      const v4i32_t ci = {32768, 32768, 32768, 32768};          // 128 * sum(w)
      const v4i32_t cl = {65536, 65536, 65536, 65536};          // 128 * sum(w) + the rounding half
      const v4i32_t chi = __builtin_amdgcn_mfma_i32_16x16x32_i8(a_hi, bv[nb], ci, 0, 0, 0);
      const v4i32_t clo = __builtin_amdgcn_mfma_i32_16x16x32_i8(a_lo, bv[nb], cl, 0, 0, 0);
      uint32_t q[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) q[r] = ((uint32_t)chi[r] << 8) + (uint32_t)clo[r];   // < 2^24: the pixel is byte 2
      const uint32_t v = __builtin_amdgcn_perm(q[1], q[0], 0x0c0c0602u) | __builtin_amdgcn_perm(q[3], q[2], 0x06020c0cu);
      const int ly = 16 * nb + m, gy = y0 + ly, gx = x0 + 16 * j + 4 * g;
#ifdef VUS_BLUR_EXP_NOSTORE   // timing experiment: everything but the stores (a value nobody produces keeps the result live)
      if (v != 0x12345678u) continue;
#endif
      if (ly < TH && gy < H) {
        uint8_t* o = blur_out + ((size_t)n * H + gy) * W + gx;
        if (gx + 3 < W) {
          __builtin_memcpy(o, &v, 4);
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (gx + e < W) o[e] = (uint8_t)(v >> (8 * e));
        }
      }
    }
  }
}

#ifdef VUS_FAST_DEBUG_COUNT   // counting build (tools/fast_counts.py): tiles, strips listed by pass 1a, pixels listed by pass 1b
__device__ unsigned long long g_fast_dbg[4];
extern "C" int vus_debug_fast_counters(unsigned long long* out, int reset) {
  if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_fast_dbg), sizeof(g_fast_dbg)) != hipSuccess) return -1;
  if (reset) { unsigned long long z[4] = {0, 0, 0, 0}; if (hipMemcpyToSymbol(HIP_SYMBOL(g_fast_dbg), z, sizeof(z)) != hipSuccess) return -1; }
  return 0;
}
#endif
// One 128 x 24 tile of image n.  HIST (with DETECT): the non-max-suppression survivors of the tile are not listed but
// counted by score into hist[256 n + score] (vus_fast_threshold_estimate's sample).
template <bool WRITE_SCORE, bool DETECT, bool BLUR, bool HIST = false, bool REGIONS = false>
__device__ __forceinline__ void fast_tile_body(
    const uint8_t* __restrict__ img, int H, int W, int pitch, int thr, int border,
    uint8_t* __restrict__ score_out, uint8_t* __restrict__ blur_out,
    uint32_t* __restrict__ cand_keys, int cand_cap, int* __restrict__ cand_count, int* __restrict__ hist,
    int n, int tile, int tiles_x) {
  __shared__ __attribute__((aligned(8))) uint32_t s_img[IMG_ROWS * IMG_DW];
  __shared__ uint32_t s_score[(WRITE_SCORE || DETECT) ? SC_ROWS * SC_DW : 1];
  // the strip pre-test's tables (s_mm, s_strip) share the horizontal-blur buffer of the VALU smoothing (not written
  // before pass 2); the matrix-core smoothing has no such buffer
  constexpr bool STRIP = VUS_FAST_STRIP && (WRITE_SCORE || DETECT);
  constexpr bool MFMA_BLUR = VUS_BLUR_MFMA && BLUR && IMG_ROWS == 32;
  constexpr int AUX_DW = STRIP ? (IMG_ROWS * IMG_DW + SC_ROWS * SC_DW + 1) / 2 : 2;
  __shared__ uint32_t s_h[(BLUR && !MFMA_BLUR && H_ROWS * H_DW > AUX_DW) ? H_ROWS * H_DW : AUX_DW];
  __shared__ uint16_t s_work[(WRITE_SCORE || DETECT) ? SC_ROWS * SC_DW * 4 : 2];   // pixels that pass the pre-test
#if VUS_FAST_STRIP
  uint16_t* const s_mm = reinterpret_cast<uint16_t*>(s_h);       // min | max << 8 of each staged dword
  uint16_t* const s_strip = s_mm + IMG_ROWS * IMG_DW;            // strips that pass the strip test
  __shared__ int s_nstrip;
#endif
  __shared__ int s_cnt, s_base, s_nwork;
  // LATE_BLUR (the detection launch proper): the smoothing runs at the END of the tile, on three waves, while the fourth
  // waits for the returning atomic that reserves the tile's slots in the image's candidate list.  The staged image then
  // has to live to the end: the candidate list moves to the strip tables' buffer (dead after pass 1b).  Otherwise the
  // list reuses the image tile, dead after the exact scores.
  constexpr bool LATE_BLUR = MFMA_BLUR && DETECT && !HIST && STRIP && VUS_FAST_LATE_BLUR;
  static_assert(IMG_ROWS * IMG_DW >= TW * TH / 4, "candidate list must fit in the image tile");
  static_assert(!LATE_BLUR || AUX_DW >= TW * TH / 4, "candidate list must fit in the strip tables' buffer");
  uint32_t* const s_keys = LATE_BLUR ? s_h : s_img;

  const int tid = threadIdx.x;
  const int x0 = (tile % tiles_x) * TW, y0 = (tile / tiles_x) * TH;
  const uint8_t* im = img + (size_t)n * H * pitch;

  // stage the tile as dwords; out-of-image pixels replicate the border (what the smoothing wants;
  // FAST never produces a score within 3 pixels of the edge, so it does not care).
  // Thread = fixed dword column, 7 rows per pass: no per-element division, one address increment.
  {
    constexpr int RPP = NTHREADS / IMG_DW;   // rows per pass
    const int col = tid % IMG_DW, r0 = tid / IMG_DW;
    const int gx = x0 - 8 + 4 * col;
    const bool fast = gx >= 0 && gx + 3 < W;   // one dword load, aligned or not (odd-width pyramid levels)
    // tiles whose staged window lies inside the image (all but the frame of edge tiles) need no clamping: one
    // 32-bit offset per thread and a uniform row stride
    const bool inside = x0 >= 8 && x0 + TW + 8 <= W && y0 >= 4 && y0 + TH + 4 <= H;
    if (r0 < RPP) {
      // all loads of the thread are issued before the first is used (round 4: the per-row "load, wait, write" chain was
      // five global-memory latencies long); rows past the tile load a valid row again and are not written
      constexpr int NP = (IMG_ROWS + RPP - 1) / RPP;
      constexpr int LASTP = IMG_ROWS - (NP - 1) * RPP;   // threads with r0 < LASTP have a row in the last pass
      uint32_t v[NP];
      if (inside) {
        const uint32_t off0 = (uint32_t)((y0 - 4 + r0) * pitch + gx);
        const uint32_t step = (uint32_t)(RPP * pitch);
#pragma unroll
        for (int k = 0; k < NP - 1; ++k) __builtin_memcpy(&v[k], im + (off0 + (uint32_t)k * step), 4);
        v[NP - 1] = 0u;
        if (r0 < LASTP) __builtin_memcpy(&v[NP - 1], im + (off0 + (uint32_t)(NP - 1) * step), 4);
      } else {
        const uint8_t* rp[NP];
#pragma unroll
        for (int k = 0; k < NP; ++k) rp[k] = im + (size_t)clampi(y0 - 4 + min(r0 + RPP * k, IMG_ROWS - 1), 0, H - 1) * pitch;
        if (fast) {
#pragma unroll
          for (int k = 0; k < NP; ++k) __builtin_memcpy(&v[k], rp[k] + gx, 4);
        } else {
          const int c0 = clampi(gx, 0, W - 1), c1 = clampi(gx + 1, 0, W - 1), c2 = clampi(gx + 2, 0, W - 1),
                    c3 = clampi(gx + 3, 0, W - 1);
          uint8_t q[NP][4];
#pragma unroll
          for (int k = 0; k < NP; ++k) { q[k][0] = rp[k][c0]; q[k][1] = rp[k][c1]; q[k][2] = rp[k][c2]; q[k][3] = rp[k][c3]; }
#pragma unroll
          for (int k = 0; k < NP; ++k)
            v[k] = (uint32_t)q[k][0] | ((uint32_t)q[k][1] << 8) | ((uint32_t)q[k][2] << 16) | ((uint32_t)q[k][3] << 24);
        }
      }
#pragma unroll
      for (int k = 0; k < NP; ++k) {
        const int row = r0 + RPP * k;
        if (row < IMG_ROWS) {
          s_img[row * IMG_DW + col] = v[k];
#if VUS_FAST_STRIP
          if (WRITE_SCORE || DETECT) {
            const int b0 = byte_of(v[k], 0), b1 = byte_of(v[k], 1), b2 = byte_of(v[k], 2), b3 = byte_of(v[k], 3);
            s_mm[row * IMG_DW + col] = (uint16_t)(min(min(b0, b1), min(b2, b3)) | (max(max(b0, b1), max(b2, b3)) << 8));
          }
#endif
        }
      }
    }
  }
  if (tid == 0) {
    s_cnt = 0; s_nwork = 0;
#if VUS_FAST_STRIP
    s_nstrip = 0;
#endif
  }
  __syncthreads();
#ifdef VUS_FAST_EXIT_AFTER   // timing builds only (tools/ab): the kernel stops after phase N, results are wrong
#define VUS_FAST_EXIT(N) do { if (VUS_FAST_EXIT_AFTER == (N)) return; } while (0)
#else
#define VUS_FAST_EXIT(N)
#endif
  VUS_FAST_EXIT(1);   // staging (+ per-dword extrema)

  if (WRITE_SCORE || DETECT) {
    // Pass 1 -- cheap necessary test on every pixel of the tile plus a ring (the 3x3 non-max
    // suppression needs 1 pixel), one strip of 4 per item.  A 9-long arc of the 16-circle always
    // contains one pixel of every opposite pair, so a corner needs  min over the tested pairs of max(pair) > p + thr
    // or  max over the pairs of min(pair) < p - thr  (pairs tested: N/S, E/W and the two diagonals).  Survivors are compacted into an LDS work list (wave prefix
    // sum with DPP, one LDS atomic per wave) so that pass 2 runs the full score on dense lanes.
#if VUS_FAST_STRIP
    // Pass 1a (round 4) -- an even cheaper necessary test per STRIP, from the per-dword extrema recorded while staging:
    // a pixel of the strip can pass the N/S + E/W pair test only if
    //    max(max N dword, max S dword) > min(strip) + thr  and  max(max W dword, max E dword, b0, b3) > min(strip) + thr
    // (its N/S neighbours are bytes of the dwords above / below, its W/E neighbours lie in the dword to the left plus
    // the strip's first byte / the strip's last byte plus the dword to the right), or the mirrored condition on the
    // dark side.  Measured on the configs[1] frames: 9.0 % of the strips pass at the adaptive threshold where 8.2 % hold
    // a pixel that passes the per-pixel test, 47 % against 45 % at fast_threshold 10 (single frames on the CPU; the counting
    // build on the bench stream says 18.2 % of a tile's strips are listed and 3.9 % of its pixels pass pass 1b).  The per-pixel test then runs on
    // the listed strips only.
    // Thread = fixed strip column, S_RPP rows per pass (no per-item division; the LDS addresses of a pass differ from
    // the first one's by constants).  All passes are evaluated first and listed with ONE LDS atomic per wave.
    constexpr int S_RPP = NTHREADS / SC_DW;                       // 7 rows of 34 strips per pass
    constexpr int NIT = (SC_ROWS + S_RPP - 1) / S_RPP;
    const int sr0 = tid / SC_DW, ss = tid - sr0 * SC_DW;
    const int gx = x0 - 4 + 4 * ss;
    const bool col_ok = sr0 < S_RPP && gx + 3 >= 3 && gx < W - 3;
    const int ci0 = (sr0 + 3) * IMG_DW + ss;
    unsigned long long bal[NIT];
    bool pass[NIT];
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int sr = sr0 + it * S_RPP;
      pass[it] = false;
      if (sr0 < S_RPP && sr < SC_ROWS) {
        const int gy = y0 - 1 + sr;
        if (col_ok && gy >= 3 && gy < H - 3) {
          const int ci = ci0 + it * S_RPP * IMG_DW;
#if VUS_FAST_MM_BYTES   // the (min, max) pairs read as the two bytes they are: ten ds_read_u8 instead of five ds_read_u16 + shifts / masks
          const uint8_t* mm8 = reinterpret_cast<const uint8_t*>(s_mm) + 2 * ci;
          const int ma_lo = mm8[0], ma_hi = mm8[1], pmin = mm8[2], pmax = mm8[3], mc_lo = mm8[4], mc_hi = mm8[5];
          const int mn_lo = mm8[2 - 6 * IMG_DW], mn_hi = mm8[3 - 6 * IMG_DW], ms_lo = mm8[2 + 6 * IMG_DW], ms_hi = mm8[3 + 6 * IMG_DW];
          const uint32_t b = s_img[ci + 1];
          const int b0 = byte_of(b, 0), b3 = byte_of(b, 3);
          const int hi = min(max(mn_hi, ms_hi), max3i(ma_hi, mc_hi, max(b0, b3)));
          const int lo = max(min(mn_lo, ms_lo), min3i(ma_lo, mc_lo, min(b0, b3)));
          pass[it] = hi > pmin + thr || lo < pmax - thr;
#else
          const int ma = s_mm[ci], mb = s_mm[ci + 1], mc = s_mm[ci + 2];
          const int mn = s_mm[ci + 1 - 3 * IMG_DW], ms = s_mm[ci + 1 + 3 * IMG_DW];
          const uint32_t b = s_img[ci + 1];
          const int b0 = byte_of(b, 0), b3 = byte_of(b, 3);
          const int pmin = mb & 0xFF, pmax = mb >> 8;
          const int hi = min(max(mn >> 8, ms >> 8), max3i(ma >> 8, mc >> 8, max(b0, b3)));
          const int lo = max(min(mn & 0xFF, ms & 0xFF), min3i(ma & 0xFF, mc & 0xFF, min(b0, b3)));
          pass[it] = hi > pmin + thr || lo < pmax - thr;
#endif
        }
        s_score[sr * SC_DW + ss] = 0u;
      }
      bal[it] = __ballot(pass[it]);
    }
    int total = 0;
#pragma unroll
    for (int it = 0; it < NIT; ++it) total += __popcll(bal[it]);
    if (total > 0) {
      int base = 0;
      if ((tid & 63) == 0) base = atomicAdd(&s_nstrip, total);
      base = __builtin_amdgcn_readfirstlane(base);
#pragma unroll
      for (int it = 0; it < NIT; ++it) {
        if (pass[it])
          s_strip[base + __builtin_amdgcn_mbcnt_hi((uint32_t)(bal[it] >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bal[it], 0))] =
              (uint16_t)((sr0 + it * S_RPP) * SC_DW + ss);
        base += __popcll(bal[it]);
      }
    }
    __syncthreads();
    VUS_FAST_EXIT(2);   // + strip pre-test
    const int nstrip = s_nstrip;
#ifdef VUS_FAST_DEBUG_COUNT
    if (tid == 0 && BLUR) { atomicAdd(&g_fast_dbg[0], 1ull); atomicAdd(&g_fast_dbg[1], (unsigned long long)nstrip); }
#endif
    // (Measured and not kept, round 4: the same test with one PIXEL per lane -- nine byte reads, no v_alignbyte / v_bfe,
    // work spread over all four waves: bit-exact, 3.00 against 2.92 ms per 1000 stereo frames.  A tile lists 161 of its
    // 884 strips here and 138 of their pixels go on to the exact score, tools/fast_counts.py.)
    for (int j0 = 0; j0 < nstrip; j0 += NTHREADS) {   // uniform trip count (wave scans inside)
      const int j = j0 + tid;
      int mask = 0, idx = 0;
      if (j < nstrip) {
        idx = s_strip[j];
        const int sr = idx / SC_DW, ss = idx - sr * SC_DW;
        const int gx = x0 - 4 + 4 * ss;
        {
#else
    for (int idx0 = 0; idx0 < SC_ROWS * SC_DW; idx0 += NTHREADS) {   // uniform trip count (wave scans inside)
      const int idx = idx0 + tid;
      int mask = 0;
      if (idx < SC_ROWS * SC_DW) {
        const int sr = idx / SC_DW, ss = idx - sr * SC_DW;
        const int gy = y0 - 1 + sr, gx = x0 - 4 + 4 * ss;
        if (gy >= 3 && gy < H - 3 && gx + 3 >= 3 && gx < W - 3) {
#endif
          const uint32_t* cp = &s_img[(sr + 3) * IMG_DW + ss];
          const uint32_t a = cp[0], b = cp[1], c = cp[2];
          const uint32_t nn = s_img[sr * IMG_DW + ss + 1], so = s_img[(sr + 6) * IMG_DW + ss + 1];
          const uint32_t wv = __builtin_amdgcn_alignbyte(b, a, 1);   // bytes x-3 of the 4 pixels
          const uint32_t ev = __builtin_amdgcn_alignbyte(c, b, 3);   // bytes x+3
#if VUS_FAST_DIAG
          // the diagonal opposite pairs of the circle, (x+2,y-2)/(x-2,y+2) and (x+2,y+2)/(x-2,y-2)
          const uint32_t* up = &s_img[(sr + 1) * IMG_DW + ss];
          const uint32_t* dn = &s_img[(sr + 5) * IMG_DW + ss];
          const uint32_t nev = __builtin_amdgcn_alignbyte(up[2], up[1], 2), nwv = __builtin_amdgcn_alignbyte(up[1], up[0], 2);
          const uint32_t sev = __builtin_amdgcn_alignbyte(dn[2], dn[1], 2), swv = __builtin_amdgcn_alignbyte(dn[1], dn[0], 2);
#endif
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int p = byte_of(b, e);
            const int n_ = byte_of(nn, e), s_ = byte_of(so, e), w_ = byte_of(wv, e), e_ = byte_of(ev, e);
            int hi = min(max(n_, s_), max(e_, w_)), lo = max(min(n_, s_), min(e_, w_));
#if VUS_FAST_DIAG
            const int ne = byte_of(nev, e), sw = byte_of(swv, e), se = byte_of(sev, e), nw = byte_of(nwv, e);
            hi = min3i(hi, max(ne, sw), max(se, nw));
            lo = max3i(lo, min(ne, sw), min(se, nw));
#endif
            mask |= (hi > p + thr || lo < p - thr) ? (1 << e) : 0;
          }
          if (gx < 3 || gx + 3 >= W - 3) {   // strips straddling the 3-pixel frame (edge tiles only)
#pragma unroll
            for (int e = 0; e < 4; ++e)
              if (gx + e < 3 || gx + e >= W - 3) mask &= ~(1 << e);
          }
        }
#if !VUS_FAST_STRIP
        s_score[idx] = 0u;
#endif
      }
      const int cnt = __popc(mask);
      int incl = cnt;   // inclusive wave scan (same DPP sequence as wave_sum_i32)
      incl += __builtin_amdgcn_update_dpp(0, incl, 0x111, 0xf, 0xf, true);
      incl += __builtin_amdgcn_update_dpp(0, incl, 0x112, 0xf, 0xf, true);
      incl += __builtin_amdgcn_update_dpp(0, incl, 0x114, 0xf, 0xe, true);
      incl += __builtin_amdgcn_update_dpp(0, incl, 0x118, 0xf, 0xc, true);
      incl += __builtin_amdgcn_update_dpp(0, incl, 0x142, 0xa, 0xf, true);
      incl += __builtin_amdgcn_update_dpp(0, incl, 0x143, 0xc, 0xf, true);
      const int total = __builtin_amdgcn_readlane(incl, 63);
      int base = 0;
      if ((tid & 63) == 0 && total > 0) base = atomicAdd(&s_nwork, total);
      base = __builtin_amdgcn_readfirstlane(base);
      int pos = base + incl - cnt;
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (mask & (1 << e)) s_work[pos++] = (uint16_t)(idx * 4 + e);
    }
  }
  auto blur_rows = [&]() {
    // horizontal 7-tap pass: 4 outputs per item from 3 dwords, two v_dot4_u32_u8 per output
    constexpr uint32_t W0123 = 18u | (33u << 8) | (49u << 16) | (56u << 24);
    constexpr uint32_t W456 = 49u | (33u << 8) | (18u << 16);
    for (int idx = tid; idx < H_ROWS * STRIPS; idx += NTHREADS) {
      const int hr = idx / STRIPS, hs = idx - hr * STRIPS;
      const uint32_t* rp = &s_img[(hr + 1) * IMG_DW + hs + 1];
      const uint32_t a = rp[0], b = rp[1], c = rp[2];
      // output e covers bytes (1+e)..(7+e) of the 12-byte window {a,b,c}
      const uint32_t o0 = __builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(b, a, 1), W0123, 0u, false) +
                          __builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(c, b, 1), W456, 0u, false);
      const uint32_t o1 = __builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(b, a, 2), W0123, 0u, false) +
                          __builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(c, b, 2), W456, 0u, false);
      const uint32_t o2 = __builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(b, a, 3), W0123, 0u, false) +
                          __builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(c, b, 3), W456, 0u, false);
      const uint32_t o3 = __builtin_amdgcn_udot4(b, W0123, 0u, false) + __builtin_amdgcn_udot4(c, W456, 0u, false);
      uint2 out = make_uint2(o0 | (o1 << 16), o2 | (o3 << 16));
      *reinterpret_cast<uint2*>(&s_h[hr * H_DW + 2 * hs]) = out;
    }
  };
#ifdef VUS_FAST_DEBUG_COUNT
  __syncthreads();
  if (tid == 0 && BLUR && (WRITE_SCORE || DETECT)) atomicAdd(&g_fast_dbg[2], (unsigned long long)s_nwork);
#endif
  VUS_FAST_EXIT(3);   // + per-pixel pre-test of the listed strips
  if (BLUR && !MFMA_BLUR && !STRIP) blur_rows();
  if (MFMA_BLUR && !LATE_BLUR) blur_tile_mfma(s_img, blur_out, n, H, W, x0, y0, tid, tid >> 6, NTHREADS / 64);   // reads the staged tile only: no barrier of its own
  __syncthreads();
  VUS_FAST_EXIT(4);   // + smoothing

  if (BLUR && !MFMA_BLUR && STRIP) blur_rows();   // after the barrier: its buffer held the strip tables until here
  if (WRITE_SCORE || DETECT) {
    // Pass 2 -- exact FAST score of the survivors, one pixel per lane.  The 12-byte row windows are
    // re-aligned with v_alignbyte so that the pixel sits at byte 4 and the strip code (E = 0) applies.
    const int nwork = s_nwork;
    uint8_t* score8 = reinterpret_cast<uint8_t*>(s_score);
    for (int j = tid; j < nwork; j += NTHREADS) {
      const int ent = s_work[j];
      const int idx = ent >> 2, e = ent & 3;
      const int sr = idx / SC_DW, ss = idx - sr * SC_DW;
#if VUS_FAST_SCORE_BYTES
      const int sc = fast_score_bytes(reinterpret_cast<const uint8_t*>(s_img) + (sr + 3) * (4 * IMG_DW) + 4 * (ss + 1) + e, 4 * IMG_DW);
#else
      uint32_t r[7][3];
#pragma unroll
      for (int k = 0; k < 7; ++k) {
        const uint32_t* rp = &s_img[(sr + k) * IMG_DW + ss];
        const uint32_t a = rp[0], b = rp[1], c = rp[2];
        r[k][0] = __builtin_amdgcn_alignbyte(b, a, e);
        r[k][1] = __builtin_amdgcn_alignbyte(c, b, e);
        r[k][2] = c >> (8 * e);
      }
#if VUS_FAST_PK
      const int sc = fast_score_pk(r);
#else
      const int sc = fast_score_strip<0>(r);
#endif
#endif
      if (sc >= thr) score8[ent] = (uint8_t)sc;
    }
    __syncthreads();
  }
  VUS_FAST_EXIT(5);   // + exact scores

  if (WRITE_SCORE) {
    for (int idx = tid; idx < TH * STRIPS; idx += NTHREADS) {
      const int ly = idx / STRIPS, ls = idx - ly * STRIPS;
      const int gy = y0 + ly, gx = x0 + 4 * ls;
      const uint32_t v = s_score[(ly + 1) * SC_DW + ls + 1];
      if (gy < H) {
        uint8_t* o = score_out + ((size_t)n * H + gy) * W + gx;
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (gx + e < W) o[e] = (uint8_t)(v >> (8 * e));
      }
    }
  }
  if (BLUR && !MFMA_BLUR) {
    constexpr int BW[7] = {18, 33, 49, 56, 49, 33, 18};
    for (int idx = tid; idx < TH * STRIPS; idx += NTHREADS) {
      const int ly = idx / STRIPS, ls = idx - ly * STRIPS;
      const int gy = y0 + ly, gx = x0 + 4 * ls;
      uint32_t a0 = 0, a1 = 0, a2 = 0, a3 = 0;
#pragma unroll
      for (int k = 0; k < 7; ++k) {
        const uint2 h = *reinterpret_cast<const uint2*>(&s_h[(ly + k) * H_DW + 2 * ls]);
        a0 += (uint32_t)BW[k] * (h.x & 0xFFFFu);
        a1 += (uint32_t)BW[k] * (h.x >> 16);
        a2 += (uint32_t)BW[k] * (h.y & 0xFFFFu);
        a3 += (uint32_t)BW[k] * (h.y >> 16);
      }
      const uint32_t v = ((a0 + 32768u) >> 16) | (((a1 + 32768u) >> 16) << 8) | (((a2 + 32768u) >> 16) << 16) |
                         (((a3 + 32768u) >> 16) << 24);
      if (gy < H) {
        uint8_t* o = blur_out + ((size_t)n * H + gy) * W + gx;
        if (gx + 3 < W) {
          __builtin_memcpy(o, &v, 4);
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (gx + e < W) o[e] = (uint8_t)(v >> (8 * e));
        }
      }
    }
  }
  if (DETECT) {
    // Non-max suppression over the SURVIVOR LIST of pass 2, not over the tile (round 4): with the adaptive threshold a
    // tile keeps a few hundred scored pixels of 3536, yet nearly every 4-pixel strip of a wave's 64 had one in it, so the
    // strip loop ran its 3 x 3-dword windows for the whole tile (~700 of the tile's ~4600 wave-instructions).  A list
    // entry is the pixel's byte index in the score tile: its eight neighbours are eight byte reads.
    const int nwork = s_nwork;
    const uint8_t* score8 = reinterpret_cast<const uint8_t*>(s_score);
    for (int j = tid; j < nwork; j += NTHREADS) {
      const int ent = s_work[j];
      const int s = score8[ent];
      if (s == 0) continue;
      const int sr = ent / (4 * SC_DW), sx = ent - sr * (4 * SC_DW);     // row / byte column inside the score tile
      const int ly = sr - 1, lx = sx - 4;                                 // the ring belongs to the neighbouring tiles
      if (ly < 0 || ly >= TH || lx < 0 || lx >= TW) continue;
      const int gy = y0 + ly, gx = x0 + lx;
      if (gy < border || gy >= H - border || gx < border || gx >= W - border) continue;
      const uint8_t* c = score8 + ent;
      constexpr int RB = 4 * SC_DW;
      const int m = max(max3i(c[-RB - 1], c[-RB], c[-RB + 1]), max(max(c[-1], c[1]), max3i(c[RB - 1], c[RB], c[RB + 1])));
      if (s > m) {                                                         // strict maximum of its 8 neighbours
        if (HIST) {
          atomicAdd(&hist[256 * n + s], 1);       // a sampled tile yields a few dozen survivors: no LDS stage
        } else {
          const int p = atomicAdd(&s_cnt, 1);
          s_keys[p] = ((uint32_t)(255 - s) << VUS_KEY_POS_BITS) | (uint32_t)(gy * W + gx);
        }
      }
    }
    if (!HIST) {
      __syncthreads();
      const int cnt = s_cnt;
      // REGIONS: the image's list is filled as VUS_CAND_REGIONS sub-lists, tile t into region t mod 8, each with its counter in
      // its own first slot (cand_region_merge_kernel compacts them afterwards): the ~300 returning atomics of an image on
      // ONE word serialised at the memory side -- 0.14 of the launch's 2.93 ms (timing build with eight counters on lines
      // of their own)
      const int rcap = cand_cap / VUS_CAND_REGIONS;
      uint32_t* const list = cand_keys + (size_t)n * cand_cap + (REGIONS ? (size_t)(tile & (VUS_CAND_REGIONS - 1)) * rcap : 0);
      int* const counter = REGIONS ? reinterpret_cast<int*>(list) : &cand_count[n];
      if (LATE_BLUR) {
        if (tid == NTHREADS - 64 && cnt > 0) s_base = atomicAdd(counter, cnt);
        if (tid < NTHREADS - 64) blur_tile_mfma(s_img, blur_out, n, H, W, x0, y0, tid, tid >> 6, NTHREADS / 64 - 1);
      } else {
        if (tid == 0 && cnt > 0) s_base = atomicAdd(counter, cnt);
      }
      __syncthreads();
      if (cnt > 0) {
        const int base = s_base + (REGIONS ? 1 : 0), lim = REGIONS ? rcap : cand_cap;
        for (int i = tid; i < cnt; i += NTHREADS)
          if (base + i < lim) list[base + i] = s_keys[i];
      }
    }
  }
}

// XCD-aware block -> (image, tile) map: blocks with equal (blockIdx % 8) share an XCD, so all tiles of one
// image run on ONE XCD and the halo re-reads of neighbouring tiles hit its L2 instead of HBM (speed only).
// thr_img (may be null): per-image thresholds of the adaptive detector (vus_fast_detect_adaptive).
template <bool WRITE_SCORE, bool DETECT, bool BLUR, bool REGIONS = false>
__global__ __launch_bounds__(NTHREADS, VUS_FAST_WPE) void fast_tile_kernel(
    const uint8_t* __restrict__ img, int H, int W, int pitch, int thr, const int* __restrict__ thr_img, int border,
    uint8_t* __restrict__ score_out, uint8_t* __restrict__ blur_out,
    uint32_t* __restrict__ cand_keys, int cand_cap, int* __restrict__ cand_count, int n_img, int tiles_x,
    int tiles_per_img) {
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int n = (slot / tiles_per_img) * 8 + xcd;
  if (n >= n_img) return;
  const int tile = slot - (slot / tiles_per_img) * tiles_per_img;
  fast_tile_body<WRITE_SCORE, DETECT, BLUR, false, REGIONS>(img, H, W, pitch, thr_img ? thr_img[n] : thr, border, score_out, blur_out,
                                                            cand_keys, cand_cap, cand_count, nullptr, n, tile, tiles_x);
}

// The regions of fast_tile_kernel<.., REGIONS>: counters zeroed before the launch ...
__global__ void cand_region_zero_kernel(uint32_t* __restrict__ cand_keys, int cand_cap, int n_img) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t < n_img * VUS_CAND_REGIONS)
    cand_keys[(size_t)(t / VUS_CAND_REGIONS) * cand_cap + (size_t)(t % VUS_CAND_REGIONS) * (cand_cap / VUS_CAND_REGIONS)] = 0u;
}

// ... and compacted after it, one workgroup per image: region r's keys (slots 1 .. of its part of the row) move down to
// where region r - 1's ended -- always to lower addresses than anything still unread, a chunk read whole before it is
// written -- and cand_count[n] becomes the total.  A region that overflowed its share of cand_cap (the image's candidates
// would have to sit in every eighth tile for that to happen below cand_cap) is reported as an overflow of the list:
// cand_count[n] > cand_cap, the unwritten tail filled with VUS_KEY_INVALID.
__global__ __launch_bounds__(256) void cand_region_merge_kernel(uint32_t* __restrict__ cand_keys, int cand_cap, int* __restrict__ cand_count) {
  __shared__ int s_c[VUS_CAND_REGIONS];
  const int n = blockIdx.x, tid = threadIdx.x;
  const int rcap = cand_cap / VUS_CAND_REGIONS;
  uint32_t* row = cand_keys + (size_t)n * cand_cap;
  if (tid < VUS_CAND_REGIONS) s_c[tid] = (int)row[(size_t)tid * rcap];
  __syncthreads();
  int off = 0;
  long long all = 0;
  bool over = false;
  for (int r = 0; r < VUS_CAND_REGIONS; ++r) {
    const int c = s_c[r], m = min(c, rcap - 1);
    all += c;
    over |= c > rcap - 1;
    const uint32_t* src = row + (size_t)r * rcap + 1;
    for (int i0 = 0; i0 < m; i0 += 256) {
      const int i = i0 + tid;
      const uint32_t v = i < m ? src[i] : 0u;
      __syncthreads();
      if (i < m) row[off + i] = v;
      __syncthreads();
    }
    off += m;
  }
  if (over)
    for (int i = off + tid; i < cand_cap; i += 256) row[i] = VUS_KEY_INVALID;
  if (tid == 0) cand_count[n] = over ? (int)max(all, (long long)cand_cap + 1) : off;
}

// ---- the adaptive detector (round 4): the top-K keypoints of an image depend only on pixels whose score reaches s*,
// the K-th best score among the non-max-suppression survivors.  A pixel below s* can neither be selected nor suppress a
// pixel at or above it (suppression needs a neighbour with a score >= its own).  So detection at ANY threshold
// t' <= s* gives the same top K as detection at fast_threshold -- and far fewer pixels pass the pre-test of pass 1 and
// reach the exact score of pass 2, where the kernel's instructions go (configs[1]: s* ~ 128, 25 % of the pixels pass the
// pre-test at 10, ~5 % at 110).  t' is estimated per image from the survivors' score histogram of a SAMPLE of tiles
// (every sample_stride-th tile, at fast_threshold), with a margin; whether the estimate was good enough is CHECKED on
// the device (did the image yield >= K candidates?), and the images that failed are detected again at fast_threshold.
// Bit-identical keypoints by construction, whatever the estimate.
__global__ __launch_bounds__(NTHREADS, VUS_FAST_WPE) void fast_sample_kernel(const uint8_t* __restrict__ img, int H, int W, int pitch,
                                                               int thr, int border, int* __restrict__ hist, int n_img,
                                                               int tiles_x, int tiles_per_img, int stride, int n_sampled) {
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int n = (slot / n_sampled) * 8 + xcd;
  if (n >= n_img) return;
  const int tile = (slot - (slot / n_sampled) * n_sampled) * stride + stride / 2;
  if (tile >= tiles_per_img) return;
  fast_tile_body<false, true, false, true>(img, H, W, pitch, thr, border, nullptr, nullptr, nullptr, 0, nullptr, hist, n, tile, tiles_x);
}

// thr_img[n] = the largest t in (floor, 254] with  (survivors of the sample with score >= t) * n_tiles * den  >=
// max_kp * n_sampled * num  (num / den: the margin), thr if there is none (floor = the threshold the sample was detected
// at, >= thr: the bins below it are empty).  One thread per image.
__global__ __launch_bounds__(256) void fast_pick_threshold_kernel(const int* __restrict__ hist, int n_img, int thr, int floor, int max_kp,
                                                                  long long n_tiles, long long n_sampled, int num, int den,
                                                                  int* __restrict__ thr_img) {
  // one WAVE per image (one thread per image walked its 200 bins as 200 dependent loads: 29 us for 2000 images): lane l
  // holds bins 4 l .. 4 l + 3, a suffix scan over the lanes gives every bin its count(score >= t), the largest
  // qualifying t wins
  const int n = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (n >= n_img) return;
  const int4 h = reinterpret_cast<const int4*>(hist + 256 * (size_t)n)[lane];
  const int hv[4] = {h.x, h.y, h.z, lane == 63 ? 0 : h.w};        // bin 255 is never counted (scores end at 254)
  int above = hv[0] + hv[1] + hv[2] + hv[3];                       // -> sum over the lanes ABOVE this one
  int incl = above;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const int o = __shfl_down(incl, d);
    if (lane + d < 64) incl += o;
  }
  above = incl - above;
  const long long need = (long long)max_kp * n_sampled * num;
  long long run = above;
  int best = -1;
#pragma unroll
  for (int q = 3; q >= 0; --q) {
    run += hv[q];
    const int t = 4 * lane + q;
    if (best < 0 && t > floor && t <= 254 && run * n_tiles * den >= need) best = t;
  }
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) best = max(best, __shfl_xor(best, d));
  if (lane == 0) thr_img[n] = best > floor ? best : thr;
}

// images whose adaptive pass yielded fewer than max_kp candidates although it ran above fast_threshold: listed,
// their counts reset (one workgroup)
__global__ __launch_bounds__(1024) void fast_retry_list_kernel(const int* __restrict__ thr_img, int thr, int max_kp, int n_img,
                                                               int* __restrict__ cand_count, int* __restrict__ retry_list,
                                                               int* __restrict__ retry_count) {
  __shared__ int s_n;
  if (threadIdx.x == 0) s_n = 0;
  __syncthreads();
  for (int n = threadIdx.x; n < n_img; n += 1024)
    if (thr_img[n] > thr && cand_count[n] < max_kp) {
      retry_list[atomicAdd(&s_n, 1)] = n;
      cand_count[n] = 0;
    }
  __syncthreads();
  if (threadIdx.x == 0) retry_count[0] = s_n;
}

// persistent: (listed image, tile) items at fast_threshold; no work, no cost
__global__ __launch_bounds__(NTHREADS, VUS_FAST_WPE) void fast_retry_kernel(const uint8_t* __restrict__ img, int H, int W, int pitch, int thr,
                                                              int border, uint32_t* __restrict__ cand_keys, int cand_cap,
                                                              int* __restrict__ cand_count, const int* __restrict__ retry_list,
                                                              const int* __restrict__ retry_count, int tiles_x, int tiles_per_img) {
  const long long items = (long long)retry_count[0] * tiles_per_img;
  for (long long it = blockIdx.x; it < items; it += gridDim.x) {
    const int n = retry_list[it / tiles_per_img], tile = (int)(it % tiles_per_img);
    fast_tile_body<false, true, false>(img, H, W, pitch, thr, border, nullptr, nullptr, cand_keys, cand_cap, cand_count, nullptr, n,
                                       tile, tiles_x);
    __syncthreads();                 // the tile's LDS images are reused by the next item
  }
}

// ---------------------------------------------------------------------------------------------
// Exact top-K: the max_kp smallest (unique) keys of one image, ascending.
constexpr int SEL_THREADS = 1024;

__global__ __launch_bounds__(SEL_THREADS) void select_topk_kernel(
    const uint32_t* __restrict__ cand_keys, const int* __restrict__ cand_count, int cand_cap,
    int max_kp, int sort_n, uint32_t* __restrict__ kp_keys, int* __restrict__ kp_count) {
  extern __shared__ uint32_t s_sort[];
  __shared__ int s_hist[256];
  __shared__ int s_scan[256];
  __shared__ int s_wtot[4];
  __shared__ uint32_t s_prefix;
  __shared__ int s_k, s_out;
  const int tid = threadIdx.x;
  const int n = blockIdx.x;
  const uint32_t* keys = cand_keys + (size_t)n * cand_cap;
  const int cnt = min(cand_count[n], cand_cap);
  const int K = min(cnt, max_kp);

  uint32_t T = 0xFFFFFFFFu;  // keep everything
  if (cnt > max_kp) {
    // 4 passes of 8 bits, most significant first: afterwards prefix == the K-th smallest key
    if (tid == 0) { s_prefix = 0; s_k = max_kp; }
    uint32_t mask = 0;
    for (int shift = 24; shift >= 0; shift -= 8) {
      for (int i = tid; i < 256; i += SEL_THREADS) s_hist[i] = 0;
      __syncthreads();
      const uint32_t prefix = s_prefix;
      for (int i = tid; i < cnt; i += SEL_THREADS) {
        uint32_t k = keys[i];
        if ((k & mask) == prefix) atomicAdd(&s_hist[(k >> shift) & 255], 1);
      }
      __syncthreads();
      // the bucket that holds the k-th smallest key: parallel inclusive scan of the 256 counts (a one-thread
      // walk over the bins cost 10 us per pass)
      const int kk = s_k;
      __syncthreads();   // everybody has read s_k and s_prefix before they are rewritten
      if (tid < 256) {
        const int h = s_hist[tid];
        int incl = h;   // inclusive scan inside the wave (DPP row shifts / broadcasts)
        incl += __builtin_amdgcn_update_dpp(0, incl, 0x111, 0xf, 0xf, true);
        incl += __builtin_amdgcn_update_dpp(0, incl, 0x112, 0xf, 0xf, true);
        incl += __builtin_amdgcn_update_dpp(0, incl, 0x114, 0xf, 0xe, true);
        incl += __builtin_amdgcn_update_dpp(0, incl, 0x118, 0xf, 0xc, true);
        incl += __builtin_amdgcn_update_dpp(0, incl, 0x142, 0xa, 0xf, true);
        incl += __builtin_amdgcn_update_dpp(0, incl, 0x143, 0xc, 0xf, true);
        if ((tid & 63) == 63) s_wtot[tid >> 6] = incl;
        s_scan[tid] = incl;
      }
      __syncthreads();
      if (tid < 256) {
        int base = 0;
        for (int w = 0; w < (tid >> 6); ++w) base += s_wtot[w];
        const int incl = s_scan[tid] + base, excl = incl - s_hist[tid];
        if (excl < kk && kk <= incl) {   // exactly one bucket
          s_prefix = prefix | ((uint32_t)tid << shift);
          s_k = kk - excl;
        }
      }
      mask |= 0xFFu << shift;
      __syncthreads();
    }
    T = s_prefix;
  }
  for (int i = tid; i < sort_n; i += SEL_THREADS) s_sort[i] = VUS_KEY_INVALID;
  if (tid == 0) s_out = 0;
  __syncthreads();
  for (int i = tid; i < cnt; i += SEL_THREADS) {
    uint32_t k = keys[i];
    if (k <= T) {
      int p = atomicAdd(&s_out, 1);
      if (p < sort_n) s_sort[p] = k;
    }
  }
  __syncthreads();
  // bitonic sort, ascending
  for (int k = 2; k <= sort_n; k <<= 1) {
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int i = tid; i < sort_n; i += SEL_THREADS) {
        int l = i ^ j;
        if (l > i) {
          uint32_t a = s_sort[i], b = s_sort[l];
          bool up = (i & k) == 0;
          if ((a > b) == up) { s_sort[i] = b; s_sort[l] = a; }
        }
      }
      __syncthreads();
    }
  }
  for (int i = tid; i < max_kp; i += SEL_THREADS)
    kp_keys[(size_t)n * max_kp + i] = i < K ? s_sort[i] : VUS_KEY_INVALID;
  if (tid == 0) kp_count[n] = K;
}

// ---------------------------------------------------------------------------------------------
// One wave per keypoint: orientation bin + 256-bit rotated BRIEF.
// Each wave stages the keypoint's two patches in LDS with row-contiguous dword loads (31 rows of the
// image for the centroid disc, 37 rows of the smoothed image for the tests), then gathers from LDS.
constexpr int OR_R = 15;                    // disc radius
constexpr int OR_ROWS = 2 * OR_R + 1;       // 31
#ifndef VUS_OR_VW
#define VUS_OR_VW 2
#endif
constexpr int OR_VW = VUS_OR_VW;               // dwords per lane and load
constexpr int OR_DW = OR_VW == 4 ? 12 : 10;    // 36 bytes cover x-15..x+15 from an aligned start; a multiple of OR_VW is loaded
constexpr int BR_R = VUS_RBRIEF_REACH;      // 18
constexpr int BR_ROWS = 2 * BR_R + 1;       // 37
constexpr int BR_DW = OR_VW == 4 ? 12 : 10;    // 40 bytes cover x-18..x+18 from an aligned start
#ifndef VUS_OR_KPW
#define VUS_OR_KPW 8
#endif
constexpr int OR_KP_PER_WAVE = VUS_OR_KPW;  // keypoints handled sequentially by one wave

// Load a (2*RADIUS+1) x (4*DW)-byte patch around (x, y) into registers: every lane issues all of
// its dword loads back to back (one memory round trip), the caller stores them to LDS afterwards.
// A lane moves VW dwords (dwordx2 / dwordx4 access at dword alignment): the texture path's cost is per
// instruction, not per byte -- 11 dword loads per keypoint took 3.19 ms per 1000 frames, 6 dwordx2 2.82
// (4 dwordx4 of 48-byte rows: 3.04).
template <int VW> struct PatchVec;
template <> struct PatchVec<2> { typedef uint32_t type __attribute__((ext_vector_type(2), aligned(1))); };
template <> struct PatchVec<4> { typedef uint32_t type __attribute__((ext_vector_type(4), aligned(1))); };

// EXACT: the patch starts at column x - RADIUS whatever its alignment (images whose rows are not dword
// aligned, e.g. odd-width pyramid levels); otherwise at the dword boundary below it (aligned loads are
// ~7 % faster on the full-size image: 2.70 vs 2.90 ms per 1000 frames).
template <int RADIUS, int DW, int VW, bool EXACT>
struct PatchRegs {
  static_assert(DW % VW == 0, "row width must be a multiple of the vector width");
  typedef typename PatchVec<VW>::type vec_t;
  static constexpr int ROWS = 2 * RADIUS + 1;
  static constexpr int HW = DW / VW;            // vectors per row
  static constexpr int N = ROWS * HW;
  static constexpr int ITERS = (N + 63) / 64;
  vec_t v[ITERS];
  uint32_t off[ITERS];   // byte offset of this lane's vectors inside an in-image patch (fixed per kernel); unsigned:
                         // scalar base + zero-extended 32-bit lane offset is an addressing mode of global_load

  __device__ __forceinline__ void init(int pitch, int lane) {
#pragma unroll
    for (int u = 0; u < ITERS; ++u) {
      const int t = min(lane + 64 * u, N - 1);
      off[u] = (uint32_t)((t / HW) * pitch + 4 * VW * (t % HW));
    }
  }
  __device__ __forceinline__ void load(const uint8_t* __restrict__ src, int H, int W, int pitch, int y, int x,
                                       int lane) {
    const int xa = EXACT ? x - RADIUS : (x - RADIUS) & ~3;   // start column
    const bool inside = y - RADIUS >= 0 && y + RADIUS < H && xa >= 0 && xa + 4 * DW <= W;
    if (inside) {   // wave-uniform
      const uint8_t* base = src + (size_t)(y - RADIUS) * pitch + xa;
#pragma unroll
      for (int u = 0; u < ITERS; ++u) v[u] = *reinterpret_cast<const vec_t*>(base + (size_t)off[u]);
    } else {        // replicate-clamped, byte by byte (keypoints near the image edge)
#pragma unroll
      for (int u = 0; u < ITERS; ++u) {
        const int t = min(lane + 64 * u, N - 1);
        const int r = t / HW, c = t - r * HW;
        const uint8_t* rp = src + (size_t)clampi(y - RADIUS + r, 0, H - 1) * pitch;
        const int gx = xa + 4 * VW * c;
#pragma unroll
        for (int k = 0; k < VW; ++k)
          v[u][k] = (uint32_t)rp[clampi(gx + 4 * k, 0, W - 1)] | ((uint32_t)rp[clampi(gx + 4 * k + 1, 0, W - 1)] << 8) |
                    ((uint32_t)rp[clampi(gx + 4 * k + 2, 0, W - 1)] << 16) |
                    ((uint32_t)rp[clampi(gx + 4 * k + 3, 0, W - 1)] << 24);
      }
    }
  }
  __device__ __forceinline__ void store(uint32_t* __restrict__ dst, int lane) const {
#pragma unroll
    for (int u = 0; u < ITERS; ++u)
      if (lane + 64 * u < N) {
        uint32_t* d = dst + VW * (lane + 64 * u);
#pragma unroll
        for (int k = 0; k < VW; k += 2) *reinterpret_cast<uint2*>(d + k) = make_uint2(v[u][k], v[u][k + 1]);
      }
  }
};

// Cross-lane helpers of orient_rbrief (round 4, second pass).  A wave works on EIGHT keypoints at once; the per-lane
// partial sums of their centroid moments (16 values) are folded with a reduce-scatter instead of 16 full wave
// reductions: v_permlane32_swap / v_permlane16_swap (gfx950) exchange half-waves / odd-even rows of TWO registers in one
// instruction, so that one add folds two values at once; then one row_ror:8 and three 8-lane all-reduce steps.
// After it the eight lanes 8g..8g+7 hold the totals of keypoint g.
__device__ __forceinline__ int fold32(int x, int y) {   // lanes < 32: x[l] + x[l+32];  lanes >= 32: y[l-32] + y[l]
  auto r = __builtin_amdgcn_permlane32_swap((uint32_t)x, (uint32_t)y, false, false);
  return (int)(r[0] + r[1]);
}
__device__ __forceinline__ int fold16(int x, int y) {   // rows 0, 2: x's row pair summed;  rows 1, 3: y's row pair summed
  auto r = __builtin_amdgcn_permlane16_swap((uint32_t)x, (uint32_t)y, false, false);
  return (int)(r[0] + r[1]);
}
template <int CTRL>
__device__ __forceinline__ int dpp_i32(int v) { return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xf, 0xf, true); }
constexpr int DPP_ROW_ROR8 = 0x128, DPP_HALF_MIRROR = 0x141, DPP_QUAD_XOR1 = 0xB1, DPP_QUAD_XOR2 = 0x4E;
__device__ __forceinline__ int fold8(int x, int y, bool upper8) {   // lanes with bit 3 clear: x over lane pairs (l, l+8); set: y
  const int keep = upper8 ? y : x, give = upper8 ? x : y;
  return keep + dpp_i32<DPP_ROW_ROR8>(give);
}
__device__ __forceinline__ int allsum8(int v) {   // every lane: the sum over its group of eight lanes
  v += dpp_i32<DPP_HALF_MIRROR>(v);
  v += dpp_i32<DPP_QUAD_XOR1>(v);
  v += dpp_i32<DPP_QUAD_XOR2>(v);
  return v;
}
template <int CTRL>
__device__ __forceinline__ long long dpp_max_i64(long long key) {   // every source lane of these patterns exists
  const int lo = dpp_i32<CTRL>((int)(uint32_t)key), hi = dpp_i32<CTRL>((int)(key >> 32));
  const long long ok = (long long)(((unsigned long long)(uint32_t)hi << 32) | (uint32_t)lo);
  return ok > key ? ok : key;
}
__device__ __forceinline__ long long allmax8_i64(long long key) {
  key = dpp_max_i64<DPP_HALF_MIRROR>(key);
  key = dpp_max_i64<DPP_QUAD_XOR1>(key);
  key = dpp_max_i64<DPP_QUAD_XOR2>(key);
  return key;
}

// v_writelane_b32: one lane of a register takes a wave-uniform value (no builtin in this compiler).  Value and lane
// select are both scalar registers and gfx9 encodings read ONE over the constant bus: the select goes through M0, which
// the instruction accepts as its lane operand.  M0 is written and consumed inside the one asm statement; no kernel of
// this file gives the compiler a reason to keep a value of its own there (no LDS-DMA, no indirect register indexing).
__device__ __forceinline__ uint32_t writelane_u32(uint32_t reg, uint32_t value, int lane_sel) {
  asm("s_mov_b32 m0, %2\n\tv_writelane_b32 %0, %1, m0" : "+v"(reg) : "s"(value), "s"(lane_sel));   // M0 is a reserved register: not listable as a clobber
  return reg;
}


// Rotated test pattern re-laid for the kernel: [bin][lane][word] = the two patch byte offsets
// (y * row bytes + x, int16 each) of test 64 * word + lane.  Derived from VUS_RBRIEF_ROT at COMPILE time, so
// the table is part of the code object: no initialisation launch, nothing to order between streams or threads.
struct RotOffTable {
  uint32_t v[VUS_N_ANGLE_BINS * 64 * 4];
};
constexpr RotOffTable make_rot_off_table() {
  RotOffTable r{};
  for (int e = 0; e < VUS_N_ANGLE_BINS * 256; ++e) {
    const int bin = e / 256, test = e - 256 * bin, w = test >> 6, lane = test & 63;
    const int oa = VUS_RBRIEF_ROT[4 * e + 1] * (4 * BR_DW) + VUS_RBRIEF_ROT[4 * e];
    const int ob = VUS_RBRIEF_ROT[4 * e + 3] * (4 * BR_DW) + VUS_RBRIEF_ROT[4 * e + 2];
    r.v[(bin * 64 + lane) * 4 + w] = ((uint32_t)oa & 0xFFFFu) | ((uint32_t)ob << 16);
  }
  return r;
}
__device__ __attribute__((aligned(16))) const RotOffTable g_rot_off_table = make_rot_off_table();

#ifndef VUS_OR_WPE
#define VUS_OR_WPE 4
#endif
#ifndef VUS_OR_GRP
#define VUS_OR_GRP 4
#endif
constexpr int OR_GRP = VUS_OR_GRP;   // centroid patches of this many keypoints are in flight together (4 or 8)
static_assert(OR_VW == 2 && OR_KP_PER_WAVE == 8, "orient_rbrief_kernel is written for dwordx2 patch loads and 8 keypoints per wave");
constexpr int OR_NV = OR_ROWS * (OR_DW / OR_VW);   // 155 dwordx2 vectors of a centroid patch
constexpr int OR_WT = 192;                         // weight entries per byte alignment: one per vector, padded to 3 x 64 lanes

// The disc weights of orient_rbrief_kernel's s_w as a compile-time table: entry (sh, t) = weights of patch vector t (row
// t / 5, dword pair t % 5) for a patch whose first column sits sh bytes into its first dword: .x/.y = (dx + 15) inside
// the disc else 0 (u8 x 4) of the two dwords, .z/.w = 1 inside the disc else 0.
struct DiscWeightTable {
  uint32_t v[4 * OR_WT * 4];
};
constexpr DiscWeightTable make_disc_weight_table() {
  constexpr int umax[16] = {15, 15, 15, 15, 14, 14, 14, 13, 13, 12, 11, 10, 9, 8, 6, 3};
  DiscWeightTable r{};
  for (int e = 0; e < 4 * OR_WT; ++e) {
    const int sh = e / OR_WT, t = e - sh * OR_WT;
    if (t >= OR_NV) continue;
    const int row = t / (OR_DW / 2), c = t - row * (OR_DW / 2);
    const int dy = row - OR_R, um = umax[dy < 0 ? -dy : dy];
    for (int d = 0; d < 2; ++d)
      for (int b = 0; b < 4; ++b) {
        const int dx = 4 * (2 * c + d) + b - sh - OR_R;
        if (dx >= -um && dx <= um) {
          r.v[4 * e + d] |= (uint32_t)(dx + OR_R) << (8 * b);
          r.v[4 * e + 2 + d] |= 1u << (8 * b);
        }
      }
  }
  return r;
}
__device__ __attribute__((aligned(16))) const DiscWeightTable g_disc_weight_table = make_disc_weight_table();

// One wave = eight consecutive keypoints of one image, in two phases.
//  A. orientation of all eight: the 31-row image patches arrive in registers (3 dwordx2 per lane and keypoint, up to
//     12 in flight) and are multiplied right there with the disc weights (v_dot4_u32_u8; the weights of a lane's two
//     dwords are ONE ds_read_b128): no LDS round trip for the patch.  Per-lane partial moments of the eight keypoints
//     -> reduce-scatter (above) -> lane 8g + j holds m10 / m01 of keypoint g and ranks bins 4j .. 4j+3.
//  B. descriptors, keypoint by keypoint: the 37-row patch of the smoothed image goes through LDS (double-buffered; the
//     next keypoint's rows and its bin's test offsets are in flight meanwhile), 8 byte reads + 4 compares per lane; the
//     ballots are written into lanes 4k + w of one register pair, so the eight descriptors leave in ONE 256-byte store.
// Round 4 first pass: one keypoint at a time, 232 vector instructions per keypoint (2.67 ms per 1000 stereo frames).
// One workgroup per image: the image's keypoints counting-sorted by 64 x 64-pixel cell (cells in raster order; inside a
// cell as the atomics fall -- a schedule, not a result).  order [n_img][max_kp]; slots from the image's count on map to
// themselves.
constexpr int OO_CELL = 64, OO_MAX_CELLS = 1024;
__global__ __launch_bounds__(256) void orient_order_kernel(const uint32_t* __restrict__ kp_keys, const int* __restrict__ kp_count,
                                                           int max_kp, int W, int cw, int n_cells, int* __restrict__ order) {
  __shared__ int s_cnt[OO_MAX_CELLS];
  __shared__ int s_wave[4];
  const int n = blockIdx.x, tid = threadIdx.x;
  const int cnt = min(kp_count[n], max_kp);
  const uint32_t* keys = kp_keys + (size_t)n * max_kp;
  int* ord = order + (size_t)n * max_kp;
  for (int r = tid; r < n_cells; r += 256) s_cnt[r] = 0;
  __syncthreads();
  for (int j = tid; j < cnt; j += 256) {
    const uint32_t pos = keys[j] & VUS_KEY_POS_MASK;
    const int y = (int)(pos / (uint32_t)W), x = (int)(pos - (uint32_t)y * (uint32_t)W);
    atomicAdd(&s_cnt[(y / OO_CELL) * cw + (x / OO_CELL)], 1);
  }
  __syncthreads();
  int c[4], sum = 0;      // exclusive scan of the <= 1024 counts: four consecutive cells per thread
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    c[q] = 4 * tid + q < n_cells ? s_cnt[4 * tid + q] : 0;
    sum += c[q];
  }
  int incl = sum;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const int o = __shfl_up(incl, d);
    if ((tid & 63) >= d) incl += o;
  }
  if ((tid & 63) == 63) s_wave[tid >> 6] = incl;
  __syncthreads();
  int base = incl - sum;
  for (int w = 0; w < (tid >> 6); ++w) base += s_wave[w];
#pragma unroll
  for (int q = 0; q < 4; ++q)
    if (4 * tid + q < n_cells) {
      s_cnt[4 * tid + q] = base;   // becomes the cell's fill cursor
      base += c[q];
    }
  __syncthreads();
  for (int j = tid; j < cnt; j += 256) {
    const uint32_t pos = keys[j] & VUS_KEY_POS_MASK;
    const int y = (int)(pos / (uint32_t)W), x = (int)(pos - (uint32_t)y * (uint32_t)W);
    ord[atomicAdd(&s_cnt[(y / OO_CELL) * cw + (x / OO_CELL)], 1)] = j;
  }
  for (int j = cnt + tid; j < max_kp; j += 256) ord[j] = j;
}

// ORDERED: slot s of an image is served with keypoint order[s] (vus_orient_order: the image's keypoints grouped by 64 x 64
// cell, the unused slots mapped to themselves), so that the eight keypoints of a wave and the 32 of a workgroup are
// neighbours and their patch rows share lines in flight; the outputs go to the keypoint's own index.  A schedule only.
template <bool EXACT, bool ORDERED>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(VUS_OR_WPE, 8))) void orient_rbrief_kernel(
    const uint8_t* __restrict__ img, const uint8_t* __restrict__ blur, int H, int W, int pitch,
    const uint32_t* __restrict__ kp_keys, const int* __restrict__ kp_count, int max_kp, const int* __restrict__ order,
    uint64_t* __restrict__ desc_out, uint8_t* __restrict__ angle_out, int n_img, int chunks_per_img) {
  // centroid weights per patch vector (two dwords), for the 4 possible byte alignments of the patch:
  // .x/.y = (dx + 15) inside the disc else 0 (u8 x 4) of the two dwords, .z/.w = 1 inside the disc else 0
  __shared__ uint4 s_w[4 * OR_WT];
  __shared__ __attribute__((aligned(8))) uint32_t s_blur[4][2][BR_ROWS * BR_DW];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  // XCD-aware block -> (image, chunk) map: workgroups are dealt round-robin over the 8 XCDs, so all
  // chunks of one image go to blocks with equal (blockIdx % 8): the image's two 0.9 MB planes then
  // stay in ONE XCD's 4 MiB L2 while its 2000 patches are gathered (placement affects speed only).
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int n = (slot / chunks_per_img) * 8 + xcd;
  const int chunk = slot - (slot / chunks_per_img) * chunks_per_img;
  if (n >= n_img) return;
  {   // the disc weights: a compile-time table, copied (deriving them per workgroup cost ~19 vector instructions per keypoint)
    const uint4* wt = reinterpret_cast<const uint4*>(g_disc_weight_table.v);
#pragma unroll
    for (int i = 0; i < 4 * OR_WT / 256; ++i) s_w[threadIdx.x + 256 * i] = wt[threadIdx.x + 256 * i];
  }
  const uint8_t* im = img + (size_t)n * H * pitch;
  const uint8_t* bl = blur + (size_t)n * H * W;
  const int base_i = (chunk * 4 + wave) * OR_KP_PER_WAVE;       // this wave's keypoints: base_i .. base_i + 7
  const int n_live = clampi(min(kp_count[n], max_kp) - base_i, 0, OR_KP_PER_WAVE);
  int my_y = 0, my_x = 0;                                         // lane k < 8: position of keypoint k
  int my_idx = base_i + lane;                                     // ... and its index in the image's list
  if (ORDERED && lane < OR_KP_PER_WAVE && base_i + lane < max_kp) my_idx = order[(size_t)n * max_kp + base_i + lane];
  if (lane < n_live) {
    const uint32_t pos = kp_keys[(size_t)n * max_kp + my_idx] & VUS_KEY_POS_MASK;
    my_y = (int)(pos / (uint32_t)W);
    my_x = (int)(pos - (uint32_t)my_y * (uint32_t)W);
  }
  // the bins this lane ranks in phase A: 4 (lane & 7) + q
  int bin_cos[4], bin_sin[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int bq = min(4 * (lane & 7) + q, VUS_N_ANGLE_BINS - 1);
    bin_cos[q] = VUS_ANGLE_COS[bq];
    bin_sin[q] = VUS_ANGLE_SIN[bq];
  }
  PatchRegs<OR_R, OR_DW, OR_VW, EXACT> pr[OR_GRP];
  PatchRegs<BR_R, BR_DW, OR_VW, EXACT> pb;
#pragma unroll
  for (int kk = 0; kk < OR_GRP; ++kk) pr[kk].init(pitch, lane);
  pb.init(W, lane);
  int mom_dy[3];
#pragma unroll
  for (int u = 0; u < 3; ++u) mom_dy[u] = min(lane + 64 * u, OR_NV - 1) / (OR_DW / 2) - OR_R;
  __syncthreads();   // the weight table is complete (the only workgroup-level synchronisation of the kernel)
  if (n_live == 0) {   // wave-uniform: nothing but defined contents for the unused slots
    if (lane < 4 * OR_KP_PER_WAVE && base_i + (lane >> 2) < max_kp) desc_out[((size_t)n * max_kp + base_i) * 4 + lane] = 0;
    if (lane < OR_KP_PER_WAVE && base_i + lane < max_kp) angle_out[(size_t)n * max_kp + base_i + lane] = 0;
    return;
  }

  // ---- phase A: centroid moments of the eight keypoints, four at a time
  int pa[8], pq[8];   // per-lane partials of m10 and m01
#pragma unroll
  for (int g = 0; g < 8 / OR_GRP; ++g) {
#pragma unroll
    for (int kk = 0; kk < OR_GRP; ++kk) {
      const int k = OR_GRP * g + kk;
#ifndef VUS_OR_EXP_NORAW
      if (k < n_live)
        pr[kk].load(im, H, W, pitch, __builtin_amdgcn_readlane(my_y, k), __builtin_amdgcn_readlane(my_x, k), lane);
#else
      for (int u = 0; u < 3; ++u) pr[kk].v[u] = typename PatchVec<OR_VW>::type{(uint32_t)lane, (uint32_t)k};
#endif
    }
    if (g == 8 / OR_GRP - 1)   // the first descriptor patch joins the queue behind the last centroid patches
      pb.load(bl, H, W, W, __builtin_amdgcn_readlane(my_y, 0), __builtin_amdgcn_readlane(my_x, 0), lane);
#pragma unroll
    for (int kk = 0; kk < OR_GRP; ++kk) {
      const int k = OR_GRP * g + kk;
      pa[k] = 0;
      pq[k] = 0;
      if (k < n_live) {
        const int sh = EXACT ? 0 : (__builtin_amdgcn_readlane(my_x, k) - OR_R) & 3;
        const uint4* wt = s_w + sh * OR_WT + lane;
        uint32_t sx = 0;
        int si = 0, sy = 0;
#pragma unroll
        for (int u = 0; u < 3; ++u) {
          const uint4 w = wt[64 * u];
          sx = __builtin_amdgcn_udot4(pr[kk].v[u][0], w.x, sx, false);
          sx = __builtin_amdgcn_udot4(pr[kk].v[u][1], w.y, sx, false);
          uint32_t rs = __builtin_amdgcn_udot4(pr[kk].v[u][0], w.z, 0u, false);
          rs = __builtin_amdgcn_udot4(pr[kk].v[u][1], w.w, rs, false);
          si += (int)rs;
          sy += mom_dy[u] * (int)rs;
        }
        pa[k] = (int)sx - OR_R * si;   // sum dx I over this lane's pixels
        pq[k] = sy;                    // sum dy I
      }
    }
  }
  // reduce-scatter: afterwards the lanes of group g = lane / 8 hold the moments of keypoint g
  int m10, m01;
  {
    const bool up8 = (lane & 8) != 0;
    int c[4], d[2];
#pragma unroll
    for (int j = 0; j < 4; ++j) c[j] = fold32(pa[j], pa[j + 4]);        // lanes < 32: keypoint j, else j + 4
#pragma unroll
    for (int j = 0; j < 2; ++j) d[j] = fold16(c[j], c[j + 2]);          // row r: keypoint j + 2 r
    m10 = allsum8(fold8(d[0], d[1], up8));                              // group g: keypoint g
#pragma unroll
    for (int j = 0; j < 4; ++j) c[j] = fold32(pq[j], pq[j + 4]);
#pragma unroll
    for (int j = 0; j < 2; ++j) d[j] = fold16(c[j], c[j + 2]);
    m01 = allsum8(fold8(d[0], d[1], up8));
  }
  // nearest bin direction = largest projection, first maximum wins (integer, exact):
  // |prj| < 2^38, so (prj << 5) | (31 - bin) orders by projection, then by lowest bin, in one 64-bit max
  int my_bin;
  {
    long long best = (long long)(-0x7FFFFFFFFFFFFFFFll - 1);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int bq = 4 * (lane & 7) + q;
      long long key = ((long long)m10 * bin_cos[q] + (long long)m01 * bin_sin[q]) * 32 + (31 - bq);
      if (bq >= VUS_N_ANGLE_BINS) key = (long long)(-0x7FFFFFFFFFFFFFFFll - 1);
      best = key > best ? key : best;
    }
    best = allmax8_i64(best);
    my_bin = 31 - (int)(best & 31);   // a keypoint that is not live has zero moments: bin 0
  }
  {
    const int j_ang = ORDERED ? __shfl(my_idx, lane >> 3) : base_i + (lane >> 3);
    if ((lane & 7) == 0 && base_i + (lane >> 3) < max_kp) angle_out[(size_t)n * max_kp + j_ang] = (uint8_t)my_bin;
  }

  // ---- phase B: the descriptors
  uint32_t wlo = 0, whi = 0;   // lane 4k + w: word w of keypoint k
  const uint4* rot = reinterpret_cast<const uint4*>(g_rot_off_table.v) + lane;
  uint4 to = rot[__builtin_amdgcn_readlane(my_bin, 0) * 64];
  for (int k = 0; k < n_live; ++k) {   // scalar loop
    uint32_t* patch = s_blur[wave][k & 1];
    pb.store(patch, lane);
    const uint4 cto = to;
    const int cx = __builtin_amdgcn_readlane(my_x, k);
    if (k + 1 < n_live) {
#ifndef VUS_OR_EXP_NOBLUR
      pb.load(bl, H, W, W, __builtin_amdgcn_readlane(my_y, k + 1), __builtin_amdgcn_readlane(my_x, k + 1), lane);
#endif
      to = rot[__builtin_amdgcn_readlane(my_bin, 8 * (k + 1)) * 64];
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");   // this wave's LDS operations execute in order
    const int sh_blur = EXACT ? 0 : (cx - BR_R) & 3;   // patch column of x - radius
    const uint8_t* c = reinterpret_cast<const uint8_t*>(patch) + BR_R * (4 * BR_DW) + BR_R + sh_blur;   // the keypoint inside the patch
    const uint32_t tw[4] = {cto.x, cto.y, cto.z, cto.w};
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      const int a = c[(int16_t)(tw[w] & 0xFFFFu)];
      const int b = c[(int16_t)(tw[w] >> 16)];
      const uint64_t word = __ballot(a < b);  // lane l supplies bit l of word w
      wlo = writelane_u32(wlo, (uint32_t)word, 4 * k + w);
      whi = writelane_u32(whi, (uint32_t)(word >> 32), 4 * k + w);
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  }
  const int j_desc = ORDERED ? __shfl(my_idx, lane >> 2) : 0;   // keypoint k's index lives in lane k; lanes 4k .. 4k+3 store its words
  if (lane < 4 * OR_KP_PER_WAVE && base_i + (lane >> 2) < max_kp)
    desc_out[ORDERED ? ((size_t)n * max_kp + j_desc) * 4 + (lane & 3) : ((size_t)n * max_kp + base_i) * 4 + lane] =
        ((uint64_t)whi << 32) | wlo;
}

// ---------------------------------------------------------------------------------------------
// Brute-force Hamming: one query per lane, train descriptors staged in LDS tiles.
#ifndef VUS_MT
#define VUS_MT 512
#endif
constexpr int MT = VUS_MT;  // train tile

template <bool GATE>
__global__ __launch_bounds__(256) void hamming_match_kernel(
    const uint64_t* __restrict__ desc, const uint32_t* __restrict__ kp_keys,
    const int* __restrict__ kp_count, int max_kp, int W, const int* __restrict__ q_index,
    const int* __restrict__ t_index, int max_dy, int min_disp, int max_disp, int max_dist,
    int32_t* __restrict__ idx_out, int32_t* __restrict__ dist_out) {
  __shared__ uint64_t s_desc[MT * 4];
  __shared__ int s_xy[MT];
  const int tid = threadIdx.x;
  const int p = blockIdx.y;
  const int qi = q_index[p], ti = t_index[p];
  const int i = blockIdx.x * 256 + tid;
  const int nq = kp_count[qi], nt = kp_count[ti];
  const bool active = i < nq;
  uint64_t q0 = 0, q1 = 0, q2 = 0, q3 = 0;
  int yq = 0, xq = 0;
  if (active) {
    const uint64_t* dq = desc + ((size_t)qi * max_kp + i) * 4;
    q0 = dq[0]; q1 = dq[1]; q2 = dq[2]; q3 = dq[3];
    uint32_t pq = kp_keys[(size_t)qi * max_kp + i] & VUS_KEY_POS_MASK;
    yq = (int)(pq / (uint32_t)W);
    xq = (int)(pq - (uint32_t)yq * (uint32_t)W);
  }
  int best = 1 << 20, bidx = -1;
  const uint64_t* dt = desc + (size_t)ti * max_kp * 4;
  const uint32_t* kt = kp_keys + (size_t)ti * max_kp;
  for (int t0 = 0; t0 < nt; t0 += MT) {
    const int tn = min(MT, nt - t0);
    __syncthreads();
    for (int k = tid; k < tn * 4; k += 256) s_desc[k] = dt[(size_t)t0 * 4 + k];
    for (int k = tid; k < tn; k += 256) {
      uint32_t pt = kt[t0 + k] & VUS_KEY_POS_MASK;
      int yt = (int)(pt / (uint32_t)W);
      s_xy[k] = (yt << 16) | (int)(pt - (uint32_t)yt * (uint32_t)W);
    }
    __syncthreads();
    if (active) {
      for (int j = 0; j < tn; ++j) {
        int dist = __popcll(q0 ^ s_desc[4 * j]) + __popcll(q1 ^ s_desc[4 * j + 1]) +
                   __popcll(q2 ^ s_desc[4 * j + 2]) + __popcll(q3 ^ s_desc[4 * j + 3]);
        bool ok = true;
        if (GATE) {
          int xy = s_xy[j];
          int dy = yq - (int)((unsigned)xy >> 16), dx = xq - (xy & 0xFFFF);
          ok = dy <= max_dy && dy >= -max_dy && dx >= min_disp && dx <= max_disp;
        }
        if (ok && dist < best) { best = dist; bidx = t0 + j; }
      }
    }
  }
  if (i < max_kp) {
    if (bidx < 0) best = 512;
    else if (best > max_dist) bidx = -1;
    idx_out[(size_t)p * max_kp + i] = bidx;
    dist_out[(size_t)p * max_kp + i] = best;
  }
}

// ---------------------------------------------------------------------------------------------
// Row-gated Hamming (stereo): only train keypoints within max_dy rows of the query can match, so the
// train set is bucketed by image row in LDS (counting sort) and each query lane visits the ~30
// keypoints of its 2*max_dy+1 rows instead of all 2000.  Ties are broken on the train index
// explicitly, so the visiting order does not matter and the result equals the brute-force scan.
__global__ __launch_bounds__(256) void hamming_match_rows_kernel(
    const uint64_t* __restrict__ desc, const uint32_t* __restrict__ kp_keys,
    const int* __restrict__ kp_count, int max_kp, int W, int Hrows, const int* __restrict__ q_index,
    const int* __restrict__ t_index, int max_dy, int min_disp, int max_disp, int max_dist,
    int32_t* __restrict__ idx_out, int32_t* __restrict__ dist_out) {
  extern __shared__ int s_dyn[];
  int* s_start = s_dyn;                         // [Hrows + 1] first slot of every row
  int* s_cursor = s_dyn + (Hrows + 1);          // [Hrows]
  int* s_list = s_cursor + Hrows;               // [max_kp] train indices grouped by row
  int* s_xy = s_list + max_kp;                  // [max_kp] (y << 16) | x per train index
  __shared__ int s_wsum[4];
  const int tid = threadIdx.x;
  const int p = blockIdx.y;
  const int qi = q_index[p], ti = t_index[p];
  const int i = blockIdx.x * 256 + tid;
  const int nq = min(kp_count[qi], max_kp), nt = min(kp_count[ti], max_kp);
  const uint32_t* kt = kp_keys + (size_t)ti * max_kp;
  for (int r = tid; r <= Hrows; r += 256) s_start[r] = 0;
  __syncthreads();
  for (int j = tid; j < nt; j += 256) {
    const uint32_t pt = kt[j] & VUS_KEY_POS_MASK;
    const int yt = min((int)(pt / (uint32_t)W), Hrows - 1);
    s_xy[j] = (yt << 16) | (int)(pt - (uint32_t)(pt / (uint32_t)W) * (uint32_t)W);
    atomicAdd(&s_start[yt + 1], 1);
  }
  __syncthreads();
  // inclusive scan of s_start[1..Hrows] (counts) -> row starts; 256 threads, chunked
  {
    const int per = (Hrows + 255) / 256;
    const int b = 1 + tid * per, e = min(Hrows + 1, b + per);
    int sum = 0;
    for (int r = b; r < e; ++r) sum += s_start[r];
    int incl = sum;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      int v = __shfl_up(incl, o);
      if ((tid & 63) >= o) incl += v;
    }
    if ((tid & 63) == 63) s_wsum[tid >> 6] = incl;
    __syncthreads();
    int base = incl - sum;
    for (int w = 0; w < (tid >> 6); ++w) base += s_wsum[w];
    for (int r = b; r < e; ++r) { base += s_start[r]; s_start[r] = base; }
  }
  __syncthreads();
  for (int r = tid; r < Hrows; r += 256) s_cursor[r] = s_start[r];
  __syncthreads();
  for (int j = tid; j < nt; j += 256) s_list[atomicAdd(&s_cursor[s_xy[j] >> 16], 1)] = j;
  __syncthreads();

  int best = 1 << 20, bidx = -1;
  if (i < nq) {
    const uint64_t* dq = desc + ((size_t)qi * max_kp + i) * 4;
    const uint64_t q0 = dq[0], q1 = dq[1], q2 = dq[2], q3 = dq[3];
    const uint32_t pq = kp_keys[(size_t)qi * max_kp + i] & VUS_KEY_POS_MASK;
    const int yq = (int)(pq / (uint32_t)W), xq = (int)(pq - (uint32_t)yq * (uint32_t)W);
    const int r0 = max(0, yq - max_dy), r1 = min(Hrows - 1, yq + max_dy);
    const uint64_t* dt = desc + (size_t)ti * max_kp * 4;
    if (r0 <= r1) {
      for (int c = s_start[r0]; c < s_start[r1 + 1]; ++c) {
        const int j = s_list[c];
        const int dx = xq - (s_xy[j] & 0xFFFF);
        if (dx < min_disp || dx > max_disp) continue;
        const uint64_t* d = dt + (size_t)j * 4;
        const int dist = __popcll(q0 ^ d[0]) + __popcll(q1 ^ d[1]) + __popcll(q2 ^ d[2]) + __popcll(q3 ^ d[3]);
        if (dist < best || (dist == best && j < bidx)) { best = dist; bidx = j; }
      }
    }
  }
  if (i < max_kp) {
    if (bidx < 0) best = 512;
    else if (best > max_dist) bidx = -1;
    idx_out[(size_t)p * max_kp + i] = bidx;
    dist_out[(size_t)p * max_kp + i] = best;
  }
}

// ---------------------------------------------------------------------------------------------
// CameraMeasurement emitter: persistent ids along the temporal matches.  Frames are inherently
// sequential (ids of frame f depend on frame f-1), so ONE workgroup walks the frames; inside a frame
// everything is parallel over keypoints (LDS atomicMin picks the lowest-index predecessor, a block
// scan numbers the fresh ids in index order).
constexpr int TRK_THREADS = 1024;
__global__ __launch_bounds__(TRK_THREADS) void track_ids_kernel(
    const int32_t* __restrict__ stereo_idx, const int32_t* __restrict__ track_idx,
    const uint32_t* __restrict__ kp_keys, const int* __restrict__ kp_count, int n_frames, int max_kp, int H, int W,
    long long* __restrict__ ids_out, double* __restrict__ feat_out, long long* __restrict__ n_ids_out) {
  extern __shared__ long long s_trk[];
  long long* s_cur = s_trk;                    // [max_kp] ids carried by the previous frame's keypoints
  long long* s_nxt = s_trk + max_kp;           // [max_kp]
  int* s_src = reinterpret_cast<int*>(s_trk + 2 * max_kp);   // [max_kp] lowest predecessor index
  __shared__ int s_part[TRK_THREADS];
  __shared__ long long s_next;
  const int tid = threadIdx.x;
  const int per = (max_kp + TRK_THREADS - 1) / TRK_THREADS;   // consecutive keypoints per thread
  if (tid == 0) s_next = 0;
  int n_prev = 0;
  for (int f = 0; f < n_frames; ++f) {
    const int nl = min(kp_count[2 * f], max_kp), nr = min(kp_count[2 * f + 1], max_kp);
    for (int i = tid; i < max_kp; i += TRK_THREADS) { s_nxt[i] = -1; s_src[i] = 0x7FFFFFFF; }
    __syncthreads();
    if (f > 0)
      for (int ip = tid; ip < n_prev; ip += TRK_THREADS) {
        const int j = track_idx[(size_t)(f - 1) * max_kp + ip];
        if (j >= 0 && j < nl && s_cur[ip] >= 0) atomicMin(&s_src[j], ip);
      }
    __syncthreads();
    for (int j = tid; j < nl; j += TRK_THREADS)
      if (s_src[j] != 0x7FFFFFFF) s_nxt[j] = s_cur[s_src[j]];
    __syncthreads();
    // fresh ids in index order: thread t owns keypoints [t*per, (t+1)*per)
    int local = 0;
    for (int u = 0; u < per; ++u) {
      const int i = tid * per + u;
      if (i < nl) {
        const int j = stereo_idx[(size_t)f * max_kp + i];
        local += (j >= 0 && j < nr && s_nxt[i] < 0) ? 1 : 0;
      }
    }
    s_part[tid] = local;
    __syncthreads();
    for (int o = 1; o < TRK_THREADS; o <<= 1) {   // inclusive Hillis-Steele scan
      const int v = tid >= o ? s_part[tid - o] : 0;
      __syncthreads();
      s_part[tid] += v;
      __syncthreads();
    }
    long long id = s_next + s_part[tid] - local;
    const long long total = s_part[TRK_THREADS - 1];
    const uint32_t* kl = kp_keys + (size_t)(2 * f) * max_kp;
    const uint32_t* kr = kp_keys + (size_t)(2 * f + 1) * max_kp;
    for (int u = 0; u < per; ++u) {
      const int i = tid * per + u;
      if (i >= max_kp) break;
      long long out_id = -1;
      double ft[4] = {0.0, 0.0, 0.0, 0.0};
      if (i < nl) {
        const int j = stereo_idx[(size_t)f * max_kp + i];
        if (j >= 0 && j < nr) {
          if (s_nxt[i] < 0) s_nxt[i] = id++;
          out_id = s_nxt[i];
          const uint32_t pl = kl[i] & VUS_KEY_POS_MASK, pr = kr[j] & VUS_KEY_POS_MASK;
          ft[0] = 2.0 * (double)(pl % (uint32_t)W) / (double)W - 1.0;
          ft[1] = 2.0 * (double)(pl / (uint32_t)W) / (double)H - 1.0;
          ft[2] = 2.0 * (double)(pr % (uint32_t)W) / (double)W - 1.0;
          ft[3] = 2.0 * (double)(pr / (uint32_t)W) / (double)H - 1.0;
        }
      }
      ids_out[(size_t)f * max_kp + i] = out_id;
      double* o = feat_out + ((size_t)f * max_kp + i) * 4;
      o[0] = ft[0]; o[1] = ft[1]; o[2] = ft[2]; o[3] = ft[3];
    }
    __syncthreads();
    for (int i = tid; i < max_kp; i += TRK_THREADS) s_cur[i] = s_nxt[i];
    if (tid == 0) s_next += total;
    n_prev = nl;
    __syncthreads();
  }
  if (tid == 0) n_ids_out[0] = s_next;
}

// ---------------------------------------------------------------------------------------------
// get_landmarks (batch.py:152-166), one feature per thread, fp64, no FMA contraction
// (this file is built with -ffp-contract=off so the result is bit-identical to the scalar order).
__global__ void triangulate_kernel(const double* __restrict__ feat, int n, const double* __restrict__ cam,
                                   const double* __restrict__ Rt, double* __restrict__ out) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const double fx = cam[0], fy = cam[1], cx = cam[2], cy = cam[3], baseline = cam[4];
  const double res_x = cam[5], res_y = cam[6];
  const double f = (fx + fy) / 2.0;
  double u0 = feat[4 * i + 0], v0 = feat[4 * i + 1], u1 = feat[4 * i + 2], v1 = feat[4 * i + 3];
  double uL = (u0 + 1) * 0.5 * res_x;
  double uR = (u1 + 1) * 0.5 * res_x;
  double v = ((v0 + v1) / 2.0 + 1) * 0.5 * res_y;
  double d = uR - uL;
  double Wd = d / baseline;
  double xc = (uL - cx) / Wd, yc = (v - cy) / Wd, zc = f / Wd;
#pragma unroll
  for (int r = 0; r < 3; ++r)
    out[6 * i + r] = ((Rt[3 * r + 0] * xc + Rt[3 * r + 1] * yc) + Rt[3 * r + 2] * zc) + Rt[9 + r];
  out[6 * i + 3] = uL;
  out[6 * i + 4] = uR;
  out[6 * i + 5] = v;
}

// ---------------------------------------------------------------------------------------------
// vus_emit_stereo_factors: the reference's per-keyframe Python loops (get_landmarks per feature, batch.py:149-176; the
// landmark loop of batch_create, batch.py:295-305) for all keyframes at once.
__device__ __forceinline__ void triangulate_one(const double* __restrict__ ft, const double* __restrict__ cam,
                                                const double* __restrict__ Rt, double* out) {
  const double fx = cam[0], fy = cam[1], cx = cam[2], cy = cam[3], baseline = cam[4];
  const double res_x = cam[5], res_y = cam[6];
  const double f = (fx + fy) / 2.0;
  const double uL = (ft[0] + 1) * 0.5 * res_x;
  const double uR = (ft[2] + 1) * 0.5 * res_x;
  const double v = ((ft[1] + ft[3]) / 2.0 + 1) * 0.5 * res_y;
  const double d = uR - uL;
  const double Wd = d / baseline;
  const double xc = (uL - cx) / Wd, yc = (v - cy) / Wd, zc = f / Wd;
#pragma unroll
  for (int r = 0; r < 3; ++r) out[r] = ((Rt[3 * r + 0] * xc + Rt[3 * r + 1] * yc) + Rt[3 * r + 2] * zc) + Rt[9 + r];
  out[3] = uL;
  out[4] = uR;
  out[5] = v;
}

constexpr int EMIT_THREADS = 256;

// pass 1 (one workgroup per keyframe): features per keyframe; first sighting of every id (atomicMin of f * max_kp + i)
__global__ __launch_bounds__(EMIT_THREADS) void emit_count_kernel(const long long* __restrict__ ids, int max_kp,
                                                                  int first_frame, long long n_ids,
                                                                  int* __restrict__ frame_count,
                                                                  unsigned long long* __restrict__ lm_first) {
  __shared__ int s_n;
  const int f = blockIdx.x;
  if (threadIdx.x == 0) s_n = 0;
  __syncthreads();
  int local = 0;
  if (f >= first_frame)
    for (int i = threadIdx.x; i < max_kp; i += EMIT_THREADS) {
      const long long id = ids[(size_t)f * max_kp + i];
      if (id >= 0 && id < n_ids) {
        ++local;
        atomicMin(&lm_first[id], (unsigned long long)f * (unsigned long long)max_kp + (unsigned long long)i);
      }
    }
  atomicAdd(&s_n, local);
  __syncthreads();
  if (threadIdx.x == 0) frame_count[f] = s_n;
}

// exclusive prefix sum of the per-keyframe counts (one workgroup; n_frames is a few thousand at most)
__global__ __launch_bounds__(1024) void emit_scan_kernel(int* __restrict__ frame_base, int n_frames, int* __restrict__ count) {
  __shared__ int s_part[1024];
  const int tid = threadIdx.x;
  const int per = (n_frames + 1023) / 1024;
  int local = 0;
  for (int u = 0; u < per; ++u) {
    const int f = tid * per + u;
    if (f < n_frames) local += frame_base[f];
  }
  s_part[tid] = local;
  __syncthreads();
  for (int o = 1; o < 1024; o <<= 1) {
    const int v = tid >= o ? s_part[tid - o] : 0;
    __syncthreads();
    s_part[tid] += v;
    __syncthreads();
  }
  int run = s_part[tid] - local;
  for (int u = 0; u < per; ++u) {
    const int f = tid * per + u;
    if (f < n_frames) {
      const int c = frame_base[f];
      frame_base[f] = run;
      run += c;
    }
  }
  if (tid == 1023) {
    frame_base[n_frames] = s_part[1023];
    count[0] = s_part[1023];
  }
}

// pass 2 (one workgroup per keyframe): ordered compaction of the keyframe's features behind frame_base[f]
__global__ __launch_bounds__(EMIT_THREADS) void emit_write_kernel(const long long* __restrict__ ids,
                                                                  const double* __restrict__ feat,
                                                                  const double* __restrict__ Rt,
                                                                  const double* __restrict__ cam, int max_kp,
                                                                  int first_frame, long long n_ids,
                                                                  const int* __restrict__ frame_base,
                                                                  int* __restrict__ obs_frame,
                                                                  long long* __restrict__ obs_id,
                                                                  double* __restrict__ obs_meas,
                                                                  long long* __restrict__ lm_first,
                                                                  double* __restrict__ lm_point) {
  __shared__ int s_part[EMIT_THREADS];
  const int f = blockIdx.x, tid = threadIdx.x;
  if (f < first_frame) return;
  const int per = (max_kp + EMIT_THREADS - 1) / EMIT_THREADS;      // consecutive slots per thread
  int local = 0;
  for (int u = 0; u < per; ++u) {
    const int i = tid * per + u;
    if (i < max_kp) {
      const long long id = ids[(size_t)f * max_kp + i];
      local += (id >= 0 && id < n_ids) ? 1 : 0;
    }
  }
  s_part[tid] = local;
  __syncthreads();
  for (int o = 1; o < EMIT_THREADS; o <<= 1) {
    const int v = tid >= o ? s_part[tid - o] : 0;
    __syncthreads();
    s_part[tid] += v;
    __syncthreads();
  }
  int pos = frame_base[f] + s_part[tid] - local;
  for (int u = 0; u < per; ++u) {
    const int i = tid * per + u;
    if (i >= max_kp) break;
    const long long id = ids[(size_t)f * max_kp + i];
    if (id < 0 || id >= n_ids) continue;
    double tri[6];
    triangulate_one(feat + ((size_t)f * max_kp + i) * 4, cam, Rt + 12 * (size_t)f, tri);
    obs_frame[pos] = f;
    obs_id[pos] = id;
    obs_meas[3 * (size_t)pos] = tri[3];
    obs_meas[3 * (size_t)pos + 1] = tri[4];
    obs_meas[3 * (size_t)pos + 2] = tri[5];
    if (lm_first[id] == (long long)f * max_kp + i) {      // batch.py:297-298: the first sighting initialises L(id)
      lm_point[3 * (size_t)id] = tri[0];
      lm_point[3 * (size_t)id + 1] = tri[1];
      lm_point[3 * (size_t)id + 2] = tri[2];
    }
    ++pos;
  }
}

// h(X, L) - z of gtsam's GenericStereoFactor3D (StereoCamera::project: q = R^T (p - t), uL = cx + fx x / z,
// uR = cx + fx (x - b) / z, v = cy + fy y / z), unwhitened, one factor per thread; statement order = the oracle's
__global__ void stereo_initial_residuals_kernel(const double* __restrict__ Rt, const double* __restrict__ K,
                                                const double* __restrict__ lm_point, const int* __restrict__ obs_frame,
                                                const long long* __restrict__ obs_id, const double* __restrict__ obs_meas,
                                                int n, double* __restrict__ resid) {
  const int a = blockIdx.x * blockDim.x + threadIdx.x;
  if (a >= n) return;
  const double* T = Rt + 12 * (size_t)obs_frame[a];
  const double* p = lm_point + 3 * (size_t)obs_id[a];
  const double fx = K[0], fy = K[1], cx = K[3], cy = K[4], b = K[5];
  const double d0 = p[0] - T[9], d1 = p[1] - T[10], d2 = p[2] - T[11];
  const double x = (T[0] * d0 + T[3] * d1) + T[6] * d2;
  const double y = (T[1] * d0 + T[4] * d1) + T[7] * d2;
  const double z = (T[2] * d0 + T[5] * d1) + T[8] * d2;
  double* r = resid + 3 * (size_t)a;
  if (!(z > 0.0)) {
    r[0] = r[1] = r[2] = __builtin_huge_val();
    return;
  }
  r[0] = (cx + fx * x / z) - obs_meas[3 * (size_t)a];
  r[1] = (cx + fx * (x - b) / z) - obs_meas[3 * (size_t)a + 1];
  r[2] = (cy + fy * y / z) - obs_meas[3 * (size_t)a + 2];
}

int check_image_args(const void* img, int n_img, int H, int W, int pitch) {
  VUS_REQUIRE(img != nullptr, "image pointer is null");
  VUS_REQUIRE(n_img >= 0 && n_img <= (1 << 20), "n_img=%d out of range [0, 2^20]", n_img);
  VUS_REQUIRE(H >= 7 && W >= 7, "image %dx%d too small (need >= 7x7)", W, H);
  VUS_REQUIRE(pitch >= W, "pitch %d < W %d", pitch, W);
  VUS_REQUIRE((long long)H * W <= (1ll << VUS_KEY_POS_BITS), "H*W=%lld exceeds 2^24", (long long)H * W);
  return VUS_OK;
}

struct TileGrid {
  unsigned blocks;
  int tiles_x, tiles_per_img;
};
TileGrid tile_grid(int n_img, int H, int W) {
  const int tx = (W + TW - 1) / TW, ty = (H + TH - 1) / TH;
  return TileGrid{(unsigned)(((n_img + 7) / 8) * 8 * tx * ty), tx, tx * ty};
}

// ---------------------------------------------------------------------------------------------
// Ungated brute-force matcher on the matrix cores.  With the descriptor bits as +-1 int8 values,
// dot(q, t) = 256 - 2 * Hamming(q, t) exactly, so argmin Hamming = argmax dot: the 2000 x 2000 x 256-bit
// all-pairs problem of one image pair is an int8 GEMM (v_mfma_i32_32x32x32_i8; the VALU version is
// bound by the v_bcnt issue rate).  Rows of a 32x32 tile = train descriptors (A operand, expanded into
// LDS once per 128-train chunk and shared by the four waves), columns = the wave's 32 queries (B
// operand, expanded once into registers).  A and B use the same lane -> k map, so only the C layout
// (col = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5)) matters.  Epilogue: one key per
// value, (dot << 16) | (0xFFFF - train index), running signed max = smallest distance, ties to the
// lowest train index -- the brute-force result, bit for bit.
typedef int v4i_t __attribute__((ext_vector_type(4)));
typedef int v16i_t __attribute__((ext_vector_type(16)));
#ifndef VUS_HM_CHUNK
#define VUS_HM_CHUNK 128
#endif
constexpr int HM_CHUNK = VUS_HM_CHUNK;        // trains expanded per LDS chunk
constexpr int HM_ROWB = 272;         // bytes per expanded row: 256 + 16 (bank spread of the b128 reads)

// 4 descriptor bits -> 4 bytes of +1 (bit set) / -1
__device__ __forceinline__ int spread_pm1(uint32_t nib) {
  const uint32_t sp = (nib * 0x00204081u) & 0x01010101u;
  return (int)~(sp * 0xFEu);
}

#ifndef VUS_HM_QT
#define VUS_HM_QT 2
#endif
constexpr int HM_QT = VUS_HM_QT;             // 32-query column tiles per wave (A fragments and the chunk expansion are shared)
constexpr int HM_QWG = 4 * 32 * HM_QT;   // queries per workgroup

// 4 descriptor bits -> 4 bytes of +32 (bit set) / -32: the QUERY operand carries the factor 32 of the epilogue's key
__device__ __forceinline__ int spread_pm32(uint32_t nib) {
  const uint32_t sp = (nib * 0x00204081u) & 0x01010101u;
  return (int)((sp * 0xC0u) ^ 0xE0E0E0E0u);
}

// best key of one 32x32 tile for this lane's query.  The accumulators ARE the keys (dot << 5) | (31 - row): the factor
// 32 rides on the query operand (+-32 instead of +-1) and the row term is the accumulators' initial value (round 4: the
// 32 v_lshl_or per tile that formed the keys were a third of the epilogue's vector instructions, and the epilogue, not the
// matrix pipe, was the longer of the two per tile).  A max3 tree, then one conversion to the global key
// (dot << 16) | (0xFFFF - train index).
__device__ __forceinline__ int tile_best(const v16i_t& acc, int tb) {
  int k[16];
#pragma unroll
  for (int reg = 0; reg < 16; ++reg) k[reg] = acc[reg];
  const int m0 = max3i(k[0], k[1], k[2]), m1 = max3i(k[3], k[4], k[5]), m2 = max3i(k[6], k[7], k[8]),
            m3 = max3i(k[9], k[10], k[11]), m4 = max3i(k[12], k[13], k[14]);
  const int m = max(max3i(m0, m1, m2), max3i(m3, m4, k[15]));
  const int row = 31 - (m & 31);
  return (m >> 5) * 65536 + (0xFFFF - (tb + row));
}

__global__ __launch_bounds__(256) void hamming_match_mfma_kernel(const uint32_t* __restrict__ desc32,
                                                                 const int* __restrict__ kp_count, int max_kp,
                                                                 const int* __restrict__ q_index,
                                                                 const int* __restrict__ t_index, int max_dist,
                                                                 int32_t* __restrict__ idx_out,
                                                                 int32_t* __restrict__ dist_out) {
  __shared__ __attribute__((aligned(16))) uint8_t s_t[HM_CHUNK * HM_ROWB];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, h = lane >> 5;
  const int p = blockIdx.y;
  const int qi = q_index[p], ti = t_index[p];
  const int nq = kp_count[qi], nt = kp_count[ti];
  const int q0 = blockIdx.x * HM_QWG + 32 * HM_QT * wave + r;   // column tile c holds query q0 + 32 c
  // B fragments: k-step ks, lane half h <-> descriptor bits [32 ks + 16 h, +16)
  v4i_t bq[HM_QT][8];
#pragma unroll
  for (int c = 0; c < HM_QT; ++c) {
    const int q = q0 + 32 * c;
    const uint32_t* dq = desc32 + ((size_t)qi * max_kp + min(q, max_kp - 1)) * 8;
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
      const uint32_t bits = (q < nq ? dq[ks] : 0u) >> (16 * h);
#pragma unroll
      for (int n = 0; n < 4; ++n) bq[c][ks][n] = spread_pm32((bits >> (4 * n)) & 0xFu);
    }
  }
  // initial value of a tile's accumulators: 31 - (row of the value inside this lane's sixteen), row = (reg & 3) + 8 (reg >> 2)
  v16i_t cinit;
#pragma unroll
  for (int reg = 0; reg < 16; ++reg) cinit[reg] = 31 - ((reg & 3) + 8 * (reg >> 2));
  int best[HM_QT];
#pragma unroll
  for (int c = 0; c < HM_QT; ++c) best[c] = INT_MIN;
  const uint32_t* dt = desc32 + (size_t)ti * max_kp * 8;
  for (int t0 = 0; t0 < nt; t0 += HM_CHUNK) {
    __syncthreads();
#pragma unroll
    for (int u = 0; u < HM_CHUNK * 8 / 256; ++u) {   // (train, 32-bit word) items
      const int item = tid + 256 * u;
      const int tr = item >> 3, wd = item & 7;
      const uint32_t x = t0 + tr < nt ? dt[(size_t)(t0 + tr) * 8 + wd] : 0u;
      v4i_t lo, hi;
#pragma unroll
      for (int n = 0; n < 4; ++n) {
        lo[n] = spread_pm1((x >> (4 * n)) & 0xFu);
        hi[n] = spread_pm1((x >> (16 + 4 * n)) & 0xFu);
      }
      v4i_t* dst = reinterpret_cast<v4i_t*>(s_t + tr * HM_ROWB + 32 * wd);
      dst[0] = lo;
      dst[1] = hi;
    }
    __syncthreads();
    const int ntile = min(HM_CHUNK, nt - t0 + 31) >> 5;
    for (int tile = 0; tile < ntile; ++tile) {
      const uint8_t* arow = s_t + (32 * tile + r) * HM_ROWB + 16 * h;
      v16i_t acc[HM_QT];
#pragma unroll
      for (int ks = 0; ks < 8; ++ks) {
        const v4i_t a = *reinterpret_cast<const v4i_t*>(arow + 32 * ks);
#pragma unroll
        for (int c = 0; c < HM_QT; ++c)
          acc[c] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, bq[c][ks], ks == 0 ? cinit : acc[c], 0, 0, 0);
      }
      const int tb = t0 + 32 * tile + 4 * h;            // train index of this lane's row 0
      if (t0 + 32 * tile + 32 > nt) {                   // wave-uniform: rows past the train count cannot win
#pragma unroll
        for (int reg = 0; reg < 16; ++reg)
          if (tb + (reg & 3) + 8 * (reg >> 2) >= nt) {
#pragma unroll
            for (int c = 0; c < HM_QT; ++c) acc[c][reg] = -(1 << 20);
          }
      }
#pragma unroll
      for (int c = 0; c < HM_QT; ++c) best[c] = max(best[c], tile_best(acc[c], tb));
    }
  }
#pragma unroll
  for (int c = 0; c < HM_QT; ++c) {
    const int b2 = max(best[c], __shfl_xor(best[c], 32));
    const int q = q0 + 32 * c;
    if (h == 0 && q < max_kp) {
      int bidx = -1, bd = 512;
      const int dot = b2 >> 16;
      if (q < nq && nt > 0 && dot >= -256) {
        bd = (256 - dot) >> 1;
        bidx = 0xFFFF - (b2 & 0xFFFF);
        if (bd > max_dist) bidx = -1;
      }
      idx_out[(size_t)p * max_kp + q] = bidx;
      dist_out[(size_t)p * max_kp + q] = bd;
    }
  }
}

// ---------------------------------------------------------------------------------------------
// optional scale pyramid: bilinear resize (11-bit weights, pixel-centre alignment) and the
// level-major merge of per-level keypoints (include/vus.h)
// all quantities fit 32 bits for images up to 2^24 pixels: num < 2 Ns Nd, rem * 2048 < 2^12 Nd
__device__ __forceinline__ void resize_coeff(int d, int Ns, int Nd, int& i0, int& i1, int& w) {
  const int num = (2 * d + 1) * Ns - Nd, den = 2 * Nd;
  const int ix = num >= 0 ? (int)((unsigned)num / (unsigned)den) : -(int)((unsigned)(-num + den - 1) / (unsigned)den);
  const unsigned rem = (unsigned)(num - ix * den);
  w = (int)((rem * 2048u + (unsigned)Nd) / (unsigned)den);
  i0 = min(max(ix, 0), Ns - 1);
  i1 = min(max(ix + 1, 0), Ns - 1);
}

// One thread per 4 destination columns of an RS_ROWS-row strip.  The kernel is bound by the number of
// vector-memory instructions (a wave-wide byte load costs the texture path as much as a dword load), so
// a thread fetches the 12 source bytes that cover its 4 columns as three (unaligned) dwords per source
// row and stores one dword; the byte selectors are row-independent.  Column coefficients are computed
// once per thread, row coefficients once per workgroup (LDS).  Threads whose columns span more than 12
// source bytes (scale > 2) fall back to byte loads.
constexpr int RS_ROWS = 8;
__device__ __forceinline__ uint32_t window_byte(uint32_t w0, uint32_t w1, uint32_t w2, int o) {
  const uint32_t sel = o < 4 ? w0 : (o < 8 ? w1 : w2);
  return (sel >> (8 * (o & 3))) & 0xFFu;
}

constexpr int RS_THREADS = 64;   // 256 columns per workgroup: little waste on rows that are not a multiple of 1024
__global__ __launch_bounds__(RS_THREADS) void resize_bilinear_kernel(const uint8_t* __restrict__ src, int Hs, int Ws,
                                                              int pitch_s, uint8_t* __restrict__ dst, int Hd, int Wd,
                                                              int pitch_d) {
  __shared__ int s_y[RS_ROWS][3];
  const int xb = 4 * (blockIdx.x * RS_THREADS + threadIdx.x);
  const int ys = blockIdx.y * RS_ROWS;
  if (threadIdx.x < RS_ROWS) {
    int y0, y1, wy;
    resize_coeff(min(ys + (int)threadIdx.x, Hd - 1), Hs, Hd, y0, y1, wy);
    s_y[threadIdx.x][0] = y0 * pitch_s;
    s_y[threadIdx.x][1] = y1 * pitch_s;
    s_y[threadIdx.x][2] = wy;
  }
  __syncthreads();
  if (xb >= Wd) return;
  const uint8_t* s = src + (size_t)blockIdx.z * Hs * pitch_s;
  uint8_t* d = dst + (size_t)blockIdx.z * Hd * pitch_d;
  int x0[4], x1[4], wx[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) resize_coeff(min(xb + j, Wd - 1), Ws, Wd, x0[j], x1[j], wx[j]);
  const int start = max(min(x0[0], Ws - 12), 0);
  const bool windowed = Ws >= 12 && x1[3] - start < 12;
  int o0[4], o1[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    o0[j] = x0[j] - start;
    o1[j] = x1[j] - start;
  }
  const int nx = min(4, Wd - xb);
#pragma unroll 2
  for (int r = 0; r < RS_ROWS; ++r) {
    const int y = ys + r;
    if (y >= Hd) break;
    const int r0 = s_y[r][0], r1 = s_y[r][1], wy = s_y[r][2];
    uint32_t out[4];
    if (windowed) {
      uint32_t t[3], b[3];
      __builtin_memcpy(t, s + r0 + start, 12);
      __builtin_memcpy(b, s + r1 + start, 12);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int top = (2048 - wx[j]) * (int)window_byte(t[0], t[1], t[2], o0[j]) + wx[j] * (int)window_byte(t[0], t[1], t[2], o1[j]);
        const int bot = (2048 - wx[j]) * (int)window_byte(b[0], b[1], b[2], o0[j]) + wx[j] * (int)window_byte(b[0], b[1], b[2], o1[j]);
        out[j] = (uint32_t)(((2048 - wy) * top + wy * bot + (1 << 21)) >> 22);
      }
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int top = (2048 - wx[j]) * s[r0 + x0[j]] + wx[j] * s[r0 + x1[j]];
        const int bot = (2048 - wx[j]) * s[r1 + x0[j]] + wx[j] * s[r1 + x1[j]];
        out[j] = (uint32_t)(((2048 - wy) * top + wy * bot + (1 << 21)) >> 22);
      }
    }
    uint8_t* o = d + (size_t)y * pitch_d + xb;
    if (nx == 4) {   // one (possibly unaligned) dword store
      const uint32_t packed = out[0] | (out[1] << 8) | (out[2] << 16) | (out[3] << 24);
      __builtin_memcpy(o, &packed, 4);
    } else {
      for (int j = 0; j < nx; ++j) o[j] = (uint8_t)out[j];
    }
  }
}

__global__ __launch_bounds__(256) void pyramid_append_kernel(const uint32_t* __restrict__ lvl_keys,
                                                             const int* __restrict__ lvl_count,
                                                             const uint64_t* __restrict__ lvl_desc,
                                                             const uint8_t* __restrict__ lvl_angle, int lvl_max_kp,
                                                             int Hl, int Wl, int level, int H0, int W0, int max_kp,
                                                             uint32_t* __restrict__ kp_keys, int* __restrict__ kp_count,
                                                             uint64_t* __restrict__ desc, uint8_t* __restrict__ angle,
                                                             uint8_t* __restrict__ kp_level,
                                                             int32_t* __restrict__ kp_xy_q4) {
  const int n = blockIdx.x;
  const int base = kp_count[n];
  int cnt = min(lvl_count[n], lvl_max_kp);
  cnt = min(cnt, max_kp - base);
  __syncthreads();   // every thread has read the old count
  for (int t = threadIdx.x; t < cnt; t += 256) {
    const size_t li = (size_t)n * lvl_max_kp + t;
    const uint32_t key = lvl_keys[li];
    const int pos = (int)(key & VUS_KEY_POS_MASK), y = pos / Wl, x = pos - y * Wl;
    const int xq = (int)((((long long)(2 * x + 1) * 8 * W0 + Wl / 2) / Wl) - 8);
    const int yq = (int)((((long long)(2 * y + 1) * 8 * H0 + Hl / 2) / Hl) - 8);
    const int x0 = min(max((xq + 8) >> 4, 0), W0 - 1), y0 = min(max((yq + 8) >> 4, 0), H0 - 1);
    const size_t o = (size_t)n * max_kp + base + t;
    kp_keys[o] = (key & ~VUS_KEY_POS_MASK) | (uint32_t)(y0 * W0 + x0);
#pragma unroll
    for (int w = 0; w < 4; ++w) desc[4 * o + w] = lvl_desc[4 * li + w];
    angle[o] = lvl_angle[li];
    if (kp_level) kp_level[o] = (uint8_t)level;
    if (kp_xy_q4) {
      kp_xy_q4[2 * o] = xq;
      kp_xy_q4[2 * o + 1] = yq;
    }
  }
  if (threadIdx.x == 0) kp_count[n] = base + cnt;
}

__global__ void cross_check_kernel(const int32_t* __restrict__ idx_fwd, const int32_t* __restrict__ idx_bwd, int max_kp,
                                   long long total, int32_t* __restrict__ idx_out) {
  const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= total) return;
  const long long row = t / max_kp;
  const int i = (int)(t - row * max_kp);
  const int32_t j = idx_fwd[t];
  idx_out[t] = (j >= 0 && j < max_kp && idx_bwd[row * max_kp + j] == i) ? j : -1;
}

// Grid-bucketed selection: one workgroup per image walks the cells in row-major order; per cell, per_cell
// rounds of "smallest key of the cell above the previous pick" (block-wide min), appended in order.
__global__ __launch_bounds__(SEL_THREADS) void select_grid_kernel(const uint32_t* __restrict__ cand_keys,
                                                                  const int* __restrict__ cand_count, int cand_cap,
                                                                  int H, int W, int grid_row, int grid_col, int per_cell,
                                                                  int max_kp, uint32_t* __restrict__ kp_keys,
                                                                  int* __restrict__ kp_count) {
  __shared__ uint32_t s_min[SEL_THREADS / 64];
  __shared__ uint32_t s_pick;
  const int tid = threadIdx.x, n = blockIdx.x;
  const uint32_t* keys = cand_keys + (size_t)n * cand_cap;
  const int cnt = min(cand_count[n], cand_cap);
  uint32_t* o = kp_keys + (size_t)n * max_kp;
  int out = 0;
  for (int cell = 0; cell < grid_row * grid_col && out < max_kp; ++cell) {
    const int cy = cell / grid_col, cx = cell - cy * grid_col;
    uint32_t last = 0;
    bool have_last = false;
    for (int j = 0; j < per_cell && out < max_kp; ++j) {
      uint32_t best = VUS_KEY_INVALID;
      for (int i = tid; i < cnt; i += SEL_THREADS) {
        const uint32_t k = keys[i];
        const int pos = (int)(k & VUS_KEY_POS_MASK), y = pos / W, x = pos - y * W;
        if (y * grid_row / H == cy && x * grid_col / W == cx && !(have_last && k <= last)) best = min(best, k);
      }
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) best = min(best, (uint32_t)__shfl_xor((int)best, off));
      if ((tid & 63) == 0) s_min[tid >> 6] = best;
      __syncthreads();
      if (tid == 0) {
        uint32_t b = s_min[0];
        for (int w = 1; w < SEL_THREADS / 64; ++w) b = min(b, s_min[w]);
        s_pick = b;
      }
      __syncthreads();
      const uint32_t pick = s_pick;
      __syncthreads();   // s_pick / s_min are rewritten in the next round
      if (pick == VUS_KEY_INVALID) break;
      if (tid == 0) o[out] = pick;
      ++out;
      last = pick;
      have_last = true;
    }
  }
  for (int i = out + tid; i < max_kp; i += SEL_THREADS) o[i] = VUS_KEY_INVALID;
  if (tid == 0) kp_count[n] = out;
}

}  // namespace

extern "C" int vus_fast_score(const uint8_t* img, int n_img, int H, int W, int pitch, int thr,
                              uint8_t* score_out, void* stream) {
  if (int rc = check_image_args(img, n_img, H, W, pitch)) return rc;
  VUS_REQUIRE(score_out != nullptr, "score_out is null");
  VUS_REQUIRE(thr >= 1 && thr <= 254, "thr=%d out of range [1, 254]", thr);
  if (n_img == 0) return VUS_OK;
  const TileGrid g = tile_grid(n_img, H, W);
  fast_tile_kernel<true, false, false><<<g.blocks, NTHREADS, 0, vus::as_stream(stream)>>>(
      img, H, W, pitch, thr, nullptr, 0, score_out, nullptr, nullptr, 0, nullptr, n_img, g.tiles_x, g.tiles_per_img);
  VUS_CHECK_LAUNCH("fast_score");
  return VUS_OK;
}

extern "C" int vus_blur7(const uint8_t* img, int n_img, int H, int W, int pitch, uint8_t* out, void* stream) {
  if (int rc = check_image_args(img, n_img, H, W, pitch)) return rc;
  VUS_REQUIRE(out != nullptr, "out is null");
  if (n_img == 0) return VUS_OK;
  const TileGrid g = tile_grid(n_img, H, W);
  fast_tile_kernel<false, false, true><<<g.blocks, NTHREADS, 0, vus::as_stream(stream)>>>(
      img, H, W, pitch, 1, nullptr, 0, nullptr, out, nullptr, 0, nullptr, n_img, g.tiles_x, g.tiles_per_img);
  VUS_CHECK_LAUNCH("blur7");
  return VUS_OK;
}

extern "C" int vus_fast_detect(const uint8_t* img, int n_img, int H, int W, int pitch, int thr, int border,
                               uint8_t* blur_out, uint32_t* cand_keys, int cand_cap, int* cand_count,
                               void* stream) {
  if (int rc = check_image_args(img, n_img, H, W, pitch)) return rc;
  VUS_REQUIRE(cand_keys != nullptr && cand_count != nullptr, "candidate buffers are null");
  VUS_REQUIRE(cand_cap >= 1, "cand_cap=%d", cand_cap);
  VUS_REQUIRE(thr >= 1 && thr <= 254, "thr=%d out of range [1, 254]", thr);
  VUS_REQUIRE(border >= 0, "border=%d", border);
  if (n_img == 0) return VUS_OK;
  hipStream_t st = vus::as_stream(stream);
  const TileGrid g = tile_grid(n_img, H, W);
  if (blur_out)
    fast_tile_kernel<false, true, true><<<g.blocks, NTHREADS, 0, st>>>(
        img, H, W, pitch, thr, nullptr, border, nullptr, blur_out, cand_keys, cand_cap, cand_count, n_img, g.tiles_x,
        g.tiles_per_img);
  else
    fast_tile_kernel<false, true, false><<<g.blocks, NTHREADS, 0, st>>>(
        img, H, W, pitch, thr, nullptr, border, nullptr, nullptr, cand_keys, cand_cap, cand_count, n_img, g.tiles_x,
        g.tiles_per_img);
  VUS_CHECK_LAUNCH("fast_detect");
  return VUS_OK;
}

extern "C" int vus_fast_threshold_estimate(const uint8_t* img, int n_img, int H, int W, int pitch, int thr, int border,
                                           int max_kp, int sample_stride, int* hist, int* thr_img, void* stream) {
  if (int rc = check_image_args(img, n_img, H, W, pitch)) return rc;
  VUS_REQUIRE(hist != nullptr && thr_img != nullptr, "null buffer");
  VUS_REQUIRE(thr >= 1 && thr <= 254, "thr=%d out of range [1, 254]", thr);
  VUS_REQUIRE(border >= 0 && max_kp >= 1 && sample_stride >= 1, "border=%d max_kp=%d sample_stride=%d", border, max_kp, sample_stride);
  if (n_img == 0) return VUS_OK;
  hipStream_t st = vus::as_stream(stream);
  const TileGrid g = tile_grid(n_img, H, W);
  const int n_sampled = (g.tiles_per_img - sample_stride / 2 + sample_stride - 1) / sample_stride;     // tiles S/2, S/2 + S, ...
  VUS_CHECK_HIP(hipMemsetAsync(hist, 0, sizeof(int) * 256 * (size_t)n_img, st));
  // the sample runs at max(thr, VUS_FAST_SAMPLE_FLOOR): an estimate is only ever taken from the bins above that, and at
  // thr = 10 half of a tile's strips pass the pre-tests (the sample cost 0.16 ms per 1000 stereo frames, a twentieth of the
  // detection it serves)
  const int floor = thr > VUS_FAST_SAMPLE_FLOOR ? thr : VUS_FAST_SAMPLE_FLOOR;
  if (n_sampled > 0)
    fast_sample_kernel<<<(unsigned)(((n_img + 7) / 8) * 8 * n_sampled), NTHREADS, 0, st>>>(
        img, H, W, pitch, floor, border, hist, n_img, g.tiles_x, g.tiles_per_img, sample_stride, n_sampled);
  fast_pick_threshold_kernel<<<(n_img + 3) / 4, 256, 0, st>>>(hist, n_img, thr, floor, max_kp, g.tiles_per_img,
                                                                 n_sampled > 0 ? n_sampled : 1, VUS_FAST_MARGIN_NUM,
                                                                 VUS_FAST_MARGIN_DEN, thr_img);
  VUS_CHECK_LAUNCH("fast_threshold_estimate");
  return VUS_OK;
}

extern "C" int vus_fast_detect_adaptive(const uint8_t* img, int n_img, int H, int W, int pitch, const int* thr_img, int border,
                                        uint8_t* blur_out, uint32_t* cand_keys, int cand_cap, int* cand_count, void* stream) {
  if (int rc = check_image_args(img, n_img, H, W, pitch)) return rc;
  VUS_REQUIRE(cand_keys != nullptr && cand_count != nullptr && thr_img != nullptr, "null buffer");
  VUS_REQUIRE(cand_cap >= 1 && border >= 0, "cand_cap=%d border=%d", cand_cap, border);
  if (n_img == 0) return VUS_OK;
  hipStream_t st = vus::as_stream(stream);
  const TileGrid g = tile_grid(n_img, H, W);
#ifndef VUS_FAST_LDS_PAD
#define VUS_FAST_LDS_PAD 0   // occupancy experiment (tools/ab): dynamic LDS nobody uses = fewer workgroups per CU, same code
#endif
  // with room for it, the list is filled as eight sub-lists and compacted afterwards (fast_tile_body, REGIONS)
  const bool regions = blur_out != nullptr && cand_cap >= 64 * VUS_CAND_REGIONS;
  if (regions) {
    cand_region_zero_kernel<<<(n_img * VUS_CAND_REGIONS + 255) / 256, 256, 0, st>>>(cand_keys, cand_cap, n_img);
    fast_tile_kernel<false, true, true, true><<<g.blocks, NTHREADS, VUS_FAST_LDS_PAD, st>>>(
        img, H, W, pitch, 0, thr_img, border, nullptr, blur_out, cand_keys, cand_cap, cand_count, n_img, g.tiles_x, g.tiles_per_img);
    cand_region_merge_kernel<<<n_img, 256, 0, st>>>(cand_keys, cand_cap, cand_count);
  } else if (blur_out)
    fast_tile_kernel<false, true, true><<<g.blocks, NTHREADS, VUS_FAST_LDS_PAD, st>>>(
        img, H, W, pitch, 0, thr_img, border, nullptr, blur_out, cand_keys, cand_cap, cand_count, n_img, g.tiles_x, g.tiles_per_img);
  else
    fast_tile_kernel<false, true, false><<<g.blocks, NTHREADS, 0, st>>>(
        img, H, W, pitch, 0, thr_img, border, nullptr, nullptr, cand_keys, cand_cap, cand_count, n_img, g.tiles_x, g.tiles_per_img);
  VUS_CHECK_LAUNCH("fast_detect_adaptive");
  return VUS_OK;
}

extern "C" int vus_fast_detect_retry(const uint8_t* img, int n_img, int H, int W, int pitch, int thr, const int* thr_img,
                                     int max_kp, int border, uint32_t* cand_keys, int cand_cap, int* cand_count,
                                     int* retry_list, int* retry_count, void* stream) {
  if (int rc = check_image_args(img, n_img, H, W, pitch)) return rc;
  VUS_REQUIRE(cand_keys && cand_count && thr_img && retry_list && retry_count, "null buffer");
  VUS_REQUIRE(thr >= 1 && thr <= 254 && max_kp >= 1 && cand_cap >= 1 && border >= 0, "thr=%d max_kp=%d cand_cap=%d border=%d", thr,
              max_kp, cand_cap, border);
  if (n_img == 0) return VUS_OK;
  hipStream_t st = vus::as_stream(stream);
  const TileGrid g = tile_grid(n_img, H, W);
  fast_retry_list_kernel<<<1, 1024, 0, st>>>(thr_img, thr, max_kp, n_img, cand_count, retry_list, retry_count);
  int n_cu = 256;
  {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess ||
        n_cu < 1)
      n_cu = 256;
  }
  fast_retry_kernel<<<8 * n_cu, NTHREADS, 0, st>>>(img, H, W, pitch, thr, border, cand_keys, cand_cap, cand_count, retry_list,
                                                   retry_count, g.tiles_x, g.tiles_per_img);
  VUS_CHECK_LAUNCH("fast_detect_retry");
  return VUS_OK;
}

extern "C" int vus_select_topk(const uint32_t* cand_keys, const int* cand_count, int n_img, int cand_cap,
                               int max_kp, uint32_t* kp_keys, int* kp_count, void* stream) {
  VUS_REQUIRE(cand_keys && cand_count && kp_keys && kp_count, "null buffer");
  VUS_REQUIRE(n_img >= 0, "n_img=%d", n_img);
  VUS_REQUIRE(cand_cap >= 1, "cand_cap=%d", cand_cap);
  VUS_REQUIRE(max_kp >= 1 && max_kp <= 8192, "max_kp=%d out of range [1, 8192]", max_kp);
  if (n_img == 0) return VUS_OK;
  int sort_n = 64;
  while (sort_n < max_kp) sort_n <<= 1;
  select_topk_kernel<<<n_img, SEL_THREADS, sort_n * sizeof(uint32_t), vus::as_stream(stream)>>>(
      cand_keys, cand_count, cand_cap, max_kp, sort_n, kp_keys, kp_count);
  VUS_CHECK_LAUNCH("select_topk");
  return VUS_OK;
}

static int orient_launch(const uint8_t* img, const uint8_t* blur, int n_img, int H, int W, int pitch, const uint32_t* kp_keys,
                         const int* kp_count, int max_kp, const int* order, uint64_t* desc_out, uint8_t* angle_out, void* stream) {
  if (int rc = check_image_args(img, n_img, H, W, pitch)) return rc;
  VUS_REQUIRE(blur && kp_keys && kp_count && desc_out && angle_out, "null buffer");
  VUS_REQUIRE(max_kp >= 1, "max_kp=%d", max_kp);
  if (n_img == 0) return VUS_OK;
  const int chunks = (max_kp + 4 * OR_KP_PER_WAVE - 1) / (4 * OR_KP_PER_WAVE);
  const long long blocks = (long long)((n_img + 7) / 8) * 8 * chunks;
  VUS_REQUIRE(blocks < (1ll << 31), "too many workgroups (%lld)", blocks);
  // rows of both planes on dword boundaries -> aligned patch loads; otherwise exact-start (unaligned) loads
  const bool aligned = ((reinterpret_cast<uintptr_t>(img) | reinterpret_cast<uintptr_t>(blur) | (uintptr_t)pitch |
                         (uintptr_t)W | ((uintptr_t)H * (uintptr_t)pitch)) & 3u) == 0;
#define VUS_OR_LAUNCH(EX, ORD)                                                                              \
  orient_rbrief_kernel<EX, ORD><<<(unsigned)blocks, 256, 0, vus::as_stream(stream)>>>(                      \
      img, blur, H, W, pitch, kp_keys, kp_count, max_kp, order, desc_out, angle_out, n_img, chunks)
  if (aligned && order) VUS_OR_LAUNCH(false, true);
  else if (aligned) VUS_OR_LAUNCH(false, false);
  else if (order) VUS_OR_LAUNCH(true, true);
  else VUS_OR_LAUNCH(true, false);
#undef VUS_OR_LAUNCH
  VUS_CHECK_LAUNCH("orient_rbrief");
  return VUS_OK;
}

extern "C" int vus_orient_rbrief(const uint8_t* img, const uint8_t* blur, int n_img, int H, int W, int pitch,
                                 const uint32_t* kp_keys, const int* kp_count, int max_kp,
                                 uint64_t* desc_out, uint8_t* angle_out, void* stream) {
  return orient_launch(img, blur, n_img, H, W, pitch, kp_keys, kp_count, max_kp, nullptr, desc_out, angle_out, stream);
}

extern "C" int vus_orient_order(const uint32_t* kp_keys, const int* kp_count, int n_img, int max_kp, int H, int W, int* order,
                                void* stream) {
  VUS_REQUIRE(kp_keys && kp_count && order, "null buffer");
  VUS_REQUIRE(n_img >= 0 && max_kp >= 1 && H >= 1 && W >= 1, "n_img=%d max_kp=%d H=%d W=%d", n_img, max_kp, H, W);
  if (n_img == 0) return VUS_OK;
  // cells of 64 x 64 pixels; an image with more than 1024 of them gets coarser cells (the order is a schedule: any
  // grouping is valid)
  int cw = (W + OO_CELL - 1) / OO_CELL, ch = (H + OO_CELL - 1) / OO_CELL;
  VUS_REQUIRE((long long)cw * ch <= OO_MAX_CELLS, "image of %d x %d pixels has more than %d cells of %d x %d", W, H, OO_MAX_CELLS,
              OO_CELL, OO_CELL);
  orient_order_kernel<<<n_img, 256, 0, vus::as_stream(stream)>>>(kp_keys, kp_count, max_kp, W, cw, cw * ch, order);
  VUS_CHECK_LAUNCH("orient_order");
  return VUS_OK;
}

extern "C" int vus_orient_rbrief_ordered(const uint8_t* img, const uint8_t* blur, int n_img, int H, int W, int pitch,
                                         const uint32_t* kp_keys, const int* kp_count, int max_kp, const int* order,
                                         uint64_t* desc_out, uint8_t* angle_out, void* stream) {
  VUS_REQUIRE(order != nullptr, "null buffer");
  return orient_launch(img, blur, n_img, H, W, pitch, kp_keys, kp_count, max_kp, order, desc_out, angle_out, stream);
}

extern "C" int vus_hamming_match(const uint64_t* desc, const uint32_t* kp_keys, const int* kp_count,
                                 int max_kp, int H, int W, const int* q_index, const int* t_index, int n_pairs,
                                 int max_dy, int min_disp, int max_disp, int max_dist,
                                 int32_t* idx_out, int32_t* dist_out, void* stream) {
  VUS_REQUIRE(desc && kp_keys && kp_count && q_index && t_index && idx_out && dist_out, "null buffer");
  VUS_REQUIRE(max_kp >= 1 && W >= 1 && W < 65536 && H >= 1 && H < 65536, "max_kp=%d H=%d W=%d", max_kp, H, W);
  VUS_REQUIRE(n_pairs >= 0 && n_pairs <= 65535, "n_pairs=%d out of range [0, 65535]", n_pairs);
  if (n_pairs == 0) return VUS_OK;
  dim3 grid((max_kp + 255) / 256, n_pairs);
  const int h_rows = H;   // the gated kernel buckets the train keypoints by image row
  const size_t lds = sizeof(int) * ((size_t)2 * h_rows + 1 + 2 * (size_t)max_kp);
  if (max_dy >= 0 && h_rows <= 16384 && lds <= 96 * 1024) {
    if (lds > 48 * 1024)
      VUS_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(hamming_match_rows_kernel),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hamming_match_rows_kernel<<<grid, 256, lds, vus::as_stream(stream)>>>(
        desc, kp_keys, kp_count, max_kp, W, h_rows, q_index, t_index, max_dy, min_disp, max_disp, max_dist, idx_out,
        dist_out);
  } else {
    if (max_dy >= 0)
      hamming_match_kernel<true><<<grid, 256, 0, vus::as_stream(stream)>>>(
          desc, kp_keys, kp_count, max_kp, W, q_index, t_index, max_dy, min_disp, max_disp, max_dist, idx_out, dist_out);
    else if (max_kp <= 65535)   // ungated: int8 GEMM formulation on the matrix cores
      hamming_match_mfma_kernel<<<dim3((max_kp + HM_QWG - 1) / HM_QWG, n_pairs), 256, 0, vus::as_stream(stream)>>>(
          reinterpret_cast<const uint32_t*>(desc), kp_count, max_kp, q_index, t_index, max_dist, idx_out, dist_out);
    else
      hamming_match_kernel<false><<<grid, 256, 0, vus::as_stream(stream)>>>(
          desc, kp_keys, kp_count, max_kp, W, q_index, t_index, max_dy, min_disp, max_disp, max_dist, idx_out, dist_out);
  }
  VUS_CHECK_LAUNCH("hamming_match");
  return VUS_OK;
}

extern "C" int vus_track_ids(const int32_t* stereo_idx, const int32_t* track_idx, const uint32_t* kp_keys,
                             const int* kp_count, int n_frames, int max_kp, int H, int W, int64_t* ids_out,
                             double* feat_out, int64_t* n_ids_out, void* stream) {
  VUS_REQUIRE(stereo_idx && kp_keys && kp_count && ids_out && feat_out && n_ids_out, "null buffer");
  VUS_REQUIRE(n_frames >= 0 && max_kp >= 1 && max_kp <= 8192 && H >= 1 && W >= 1, "n_frames=%d max_kp=%d H=%d W=%d",
              n_frames, max_kp, H, W);
  VUS_REQUIRE(n_frames <= 1 || track_idx != nullptr, "track_idx is null");
  const size_t lds = (size_t)max_kp * (2 * sizeof(long long) + sizeof(int));
  if (lds > 48 * 1024)
    VUS_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(track_ids_kernel),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  track_ids_kernel<<<1, TRK_THREADS, lds, vus::as_stream(stream)>>>(
      stereo_idx, track_idx, kp_keys, kp_count, n_frames, max_kp, H, W, reinterpret_cast<long long*>(ids_out), feat_out,
      reinterpret_cast<long long*>(n_ids_out));
  VUS_CHECK_LAUNCH("track_ids");
  return VUS_OK;
}

extern "C" int vus_emit_stereo_factors(const int64_t* ids, const double* feat, const double* Rt, const double* cam,
                                       int n_frames, int max_kp, int first_frame, long long n_ids, int* frame_base,
                                       int* count, int* obs_frame, int64_t* obs_id, double* obs_meas, int64_t* lm_first,
                                       double* lm_point, void* stream) {
  VUS_REQUIRE(ids && feat && Rt && cam && frame_base && count && lm_first && lm_point, "null buffer");
  VUS_REQUIRE(obs_frame && obs_id && obs_meas, "null output buffer");
  VUS_REQUIRE(n_frames >= 0 && n_frames <= 65535 && max_kp >= 1 && first_frame >= 0 && n_ids >= 0,
              "n_frames=%d max_kp=%d first_frame=%d n_ids=%lld", n_frames, max_kp, first_frame, n_ids);
  VUS_REQUIRE((long long)n_frames * max_kp < (1ll << 31), "n_frames * max_kp overflows the int32 factor index");
  hipStream_t st = vus::as_stream(stream);
  if (n_ids > 0) VUS_CHECK_HIP(hipMemsetAsync(lm_first, 0xFF, sizeof(int64_t) * (size_t)n_ids, st));   // = -1 = UINT64_MAX
  if (n_frames == 0) {
    VUS_CHECK_HIP(hipMemsetAsync(count, 0, sizeof(int), st));
    VUS_CHECK_HIP(hipMemsetAsync(frame_base, 0, sizeof(int), st));
    return VUS_OK;
  }
  emit_count_kernel<<<n_frames, EMIT_THREADS, 0, st>>>(reinterpret_cast<const long long*>(ids), max_kp, first_frame, n_ids,
                                                        frame_base, reinterpret_cast<unsigned long long*>(lm_first));
  emit_scan_kernel<<<1, 1024, 0, st>>>(frame_base, n_frames, count);
  emit_write_kernel<<<n_frames, EMIT_THREADS, 0, st>>>(reinterpret_cast<const long long*>(ids), feat, Rt, cam, max_kp,
                                                        first_frame, n_ids, frame_base, obs_frame,
                                                        reinterpret_cast<long long*>(obs_id), obs_meas,
                                                        reinterpret_cast<long long*>(lm_first), lm_point);
  VUS_CHECK_LAUNCH("emit_stereo_factors");
  return VUS_OK;
}

extern "C" int vus_stereo_initial_residuals(const double* Rt, const double* K, const double* lm_point,
                                            const int* obs_frame, const int64_t* obs_id, const double* obs_meas, int n,
                                            double* resid, void* stream) {
  VUS_REQUIRE(n >= 0, "n=%d", n);
  if (n == 0) return VUS_OK;
  VUS_REQUIRE(Rt && K && lm_point && obs_frame && obs_id && obs_meas && resid, "null buffer");
  stereo_initial_residuals_kernel<<<(n + 255) / 256, 256, 0, vus::as_stream(stream)>>>(
      Rt, K, lm_point, obs_frame, reinterpret_cast<const long long*>(obs_id), obs_meas, n, resid);
  VUS_CHECK_LAUNCH("stereo_initial_residuals");
  return VUS_OK;
}

extern "C" int vus_triangulate(const double* feat, int n, const double* cam, const double* Rt, double* out,
                               void* stream) {
  VUS_REQUIRE(feat && cam && Rt && out, "null buffer");
  VUS_REQUIRE(n >= 0, "n=%d", n);
  if (n == 0) return VUS_OK;
  triangulate_kernel<<<(n + 255) / 256, 256, 0, vus::as_stream(stream)>>>(feat, n, cam, Rt, out);
  VUS_CHECK_LAUNCH("triangulate");
  return VUS_OK;
}

extern "C" int vus_resize_bilinear(const uint8_t* src, int n_img, int Hs, int Ws, int pitch_s, uint8_t* dst, int Hd,
                                   int Wd, int pitch_d, void* stream) {
  if (int rc = check_image_args(src, n_img, Hs, Ws, pitch_s)) return rc;
  VUS_REQUIRE(dst != nullptr, "dst is null");
  VUS_REQUIRE(Hd >= 1 && Wd >= 1 && pitch_d >= Wd, "bad destination size %dx%d pitch %d", Hd, Wd, pitch_d);
  if (n_img == 0) return VUS_OK;
  VUS_REQUIRE(n_img <= 65535, "n_img=%d exceeds the grid's z extent", n_img);
  const dim3 grid((Wd + 4 * RS_THREADS - 1) / (4 * RS_THREADS), (Hd + RS_ROWS - 1) / RS_ROWS, n_img);
  resize_bilinear_kernel<<<grid, RS_THREADS, 0, vus::as_stream(stream)>>>(src, Hs, Ws, pitch_s, dst, Hd, Wd, pitch_d);
  VUS_CHECK_LAUNCH("resize_bilinear");
  return VUS_OK;
}

extern "C" int vus_pyramid_append(const uint32_t* lvl_keys, const int* lvl_count, const uint64_t* lvl_desc,
                                  const uint8_t* lvl_angle, int n_img, int lvl_max_kp, int Hl, int Wl, int level,
                                  int H0, int W0, int max_kp, uint32_t* kp_keys, int* kp_count, uint64_t* desc,
                                  uint8_t* angle, uint8_t* kp_level, int32_t* kp_xy_q4, void* stream) {
  VUS_REQUIRE(lvl_keys && lvl_count && lvl_desc && lvl_angle && kp_keys && kp_count && desc && angle, "null buffer");
  VUS_REQUIRE(n_img >= 0 && lvl_max_kp >= 1 && max_kp >= 1, "bad sizes: n_img=%d lvl_max_kp=%d max_kp=%d", n_img,
              lvl_max_kp, max_kp);
  VUS_REQUIRE(Hl >= 1 && Wl >= 1 && H0 >= 1 && W0 >= 1 && (long long)H0 * W0 <= (1ll << VUS_KEY_POS_BITS),
              "bad image sizes: level %dx%d, base %dx%d", Hl, Wl, H0, W0);
  VUS_REQUIRE(level >= 0 && level <= 255, "level=%d", level);
  if (n_img == 0) return VUS_OK;
  pyramid_append_kernel<<<n_img, 256, 0, vus::as_stream(stream)>>>(lvl_keys, lvl_count, lvl_desc, lvl_angle, lvl_max_kp,
                                                                   Hl, Wl, level, H0, W0, max_kp, kp_keys, kp_count,
                                                                   desc, angle, kp_level, kp_xy_q4);
  VUS_CHECK_LAUNCH("pyramid_append");
  return VUS_OK;
}

extern "C" int vus_cross_check(const int32_t* idx_fwd, const int32_t* idx_bwd, int n_pairs, int max_kp,
                               int32_t* idx_out, void* stream) {
  VUS_REQUIRE(idx_fwd && idx_bwd && idx_out, "null buffer");
  VUS_REQUIRE(n_pairs >= 0 && max_kp >= 1, "n_pairs=%d max_kp=%d", n_pairs, max_kp);
  const long long total = (long long)n_pairs * max_kp;
  if (total == 0) return VUS_OK;
  cross_check_kernel<<<(unsigned)((total + 255) / 256), 256, 0, vus::as_stream(stream)>>>(idx_fwd, idx_bwd, max_kp, total,
                                                                                       idx_out);
  VUS_CHECK_LAUNCH("cross_check");
  return VUS_OK;
}

extern "C" int vus_select_grid(const uint32_t* cand_keys, const int* cand_count, int n_img, int cand_cap, int H, int W,
                               int grid_row, int grid_col, int per_cell, int max_kp, uint32_t* kp_keys, int* kp_count,
                               void* stream) {
  VUS_REQUIRE(cand_keys && cand_count && kp_keys && kp_count, "null buffer");
  VUS_REQUIRE(n_img >= 0 && cand_cap >= 1 && max_kp >= 1, "n_img=%d cand_cap=%d max_kp=%d", n_img, cand_cap, max_kp);
  VUS_REQUIRE(H >= 1 && W >= 1 && H < 32768 && W < 32768, "H=%d W=%d", H, W);
  VUS_REQUIRE(grid_row >= 1 && grid_col >= 1 && grid_row <= 256 && grid_col <= 256 && per_cell >= 1,
              "grid %dx%d per_cell=%d", grid_row, grid_col, per_cell);
  if (n_img == 0) return VUS_OK;
  select_grid_kernel<<<n_img, SEL_THREADS, 0, vus::as_stream(stream)>>>(cand_keys, cand_count, cand_cap, H, W, grid_row,
                                                                        grid_col, per_cell, max_kp, kp_keys, kp_count);
  VUS_CHECK_LAUNCH("select_grid");
  return VUS_OK;
}
