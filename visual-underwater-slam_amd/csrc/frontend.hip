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
//  * fast_detect stages a 72x40 byte tile (64x32 outputs + 4-pixel halo) in LDS once and derives
//    score, non-max suppression and the 7x7 smoothing from it: the image is read from HBM once;
//    the score map never goes to HBM (candidates leave the CU as 4-byte keys);
//  * the score is computed branch-free for every pixel with v_min3/v_max3 sliding windows
//    (no divergence on corner density);
//  * select_topk is an exact 4x8-bit MSB radix select + LDS bitonic sort, one workgroup per image;
//  * orient_rbrief uses one 64-lane wave per keypoint: lane-strided disc moments, integer bin
//    choice, and the descriptor words come straight out of __ballot (lane = test bit);
//  * hamming_match keeps the train descriptors in LDS (broadcast reads) and one query per lane.
#include "vus_common.h"
#define VUS_TABLE_QUAL __device__ const
#include "../../include/vus_orb_tables.h"

namespace {

constexpr int TW = 64;      // output tile width
constexpr int TH = 32;      // output tile height
constexpr int HALO = 4;     // 3 (FAST circle / blur taps) + 1 (NMS ring)
constexpr int SW = TW + 2 * HALO;
constexpr int SH = TH + 2 * HALO;
constexpr int SCW = TW + 4;  // score tile row stride (TW+2 used)
constexpr int NTHREADS = 256;

__device__ __forceinline__ int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }
__device__ __forceinline__ int min3i(int a, int b, int c) { return min(min(a, b), c); }
__device__ __forceinline__ int max3i(int a, int b, int c) { return max(max(a, b), c); }

// FAST-9/16 score of the pixel at c (LDS), rows `stride` bytes apart.  Sliding-window min / max of
// the 16 circle differences over every arc of 9:  w3[k] = op(d[k..k+2]),  w9[k] = op(w3[k],
// w3[k+3], w3[k+6]).  Bright arcs need min(d) large, dark arcs need max(d) small (very negative).
__device__ __forceinline__ int fast_score_at(const uint8_t* c, int stride) {
  constexpr int DX[16] = {0, 1, 2, 3, 3, 3, 2, 1, 0, -1, -2, -3, -3, -3, -2, -1};
  constexpr int DY[16] = {-3, -3, -2, -1, 0, 1, 2, 3, 3, 3, 2, 1, 0, -1, -2, -3};
  const int p = c[0];
  int d[16];
#pragma unroll
  for (int k = 0; k < 16; ++k) d[k] = (int)c[DY[k] * stride + DX[k]] - p;
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

template <bool WRITE_SCORE, bool DETECT, bool BLUR>
__global__ __launch_bounds__(NTHREADS) void fast_tile_kernel(
    const uint8_t* __restrict__ img, int H, int W, int pitch, int thr, int border,
    uint8_t* __restrict__ score_out, uint8_t* __restrict__ blur_out,
    uint32_t* __restrict__ cand_keys, int cand_cap, int* __restrict__ cand_count) {
  __shared__ uint8_t s_img[SH * SW];
  __shared__ uint8_t s_score[(TH + 2) * SCW];
  __shared__ uint16_t s_h[(TH + 6) * TW];
  __shared__ uint32_t s_keys[TH * TW / 4];
  __shared__ int s_cnt, s_base;

  const int tid = threadIdx.x;
  const int n = blockIdx.z;
  const int x0 = blockIdx.x * TW, y0 = blockIdx.y * TH;
  const uint8_t* im = img + (size_t)n * H * pitch;

  // stage the tile; out-of-image pixels replicate the border (what the smoothing wants; FAST never
  // produces a score within 3 pixels of the edge, so it does not care)
  for (int idx = tid; idx < SH * SW; idx += NTHREADS) {
    int ly = idx / SW, lx = idx - ly * SW;
    int gy = clampi(y0 - HALO + ly, 0, H - 1), gx = clampi(x0 - HALO + lx, 0, W - 1);
    s_img[idx] = im[(size_t)gy * pitch + gx];
  }
  if (tid == 0) s_cnt = 0;
  __syncthreads();

  if (WRITE_SCORE || DETECT) {
    // score on the tile plus a 1-pixel ring (needed by the 3x3 non-max suppression)
    for (int idx = tid; idx < (TH + 2) * (TW + 2); idx += NTHREADS) {
      int ly = idx / (TW + 2), lx = idx - ly * (TW + 2);
      int gy = y0 - 1 + ly, gx = x0 - 1 + lx;
      int s = 0;
      if (gy >= 3 && gy < H - 3 && gx >= 3 && gx < W - 3) {
        int sc = fast_score_at(&s_img[(ly + 3) * SW + (lx + 3)], SW);
        s = sc >= thr ? sc : 0;
      }
      s_score[ly * SCW + lx] = (uint8_t)s;
    }
  }
  if (BLUR) {
    constexpr int BW[7] = {18, 33, 49, 56, 49, 33, 18};
    // horizontal pass over rows y0-3 .. y0+TH+2
    for (int idx = tid; idx < (TH + 6) * TW; idx += NTHREADS) {
      int ly = idx / TW, lx = idx - ly * TW;
      const uint8_t* r = &s_img[(ly + 1) * SW + (lx + 1)];
      int acc = 0;
#pragma unroll
      for (int k = 0; k < 7; ++k) acc += BW[k] * (int)r[k];
      s_h[idx] = (uint16_t)acc;
    }
  }
  __syncthreads();

  if (WRITE_SCORE) {
    for (int idx = tid; idx < TH * TW; idx += NTHREADS) {
      int ly = idx / TW, lx = idx - ly * TW;
      int gy = y0 + ly, gx = x0 + lx;
      if (gy < H && gx < W) score_out[((size_t)n * H + gy) * W + gx] = s_score[(ly + 1) * SCW + lx + 1];
    }
  }
  if (BLUR) {
    constexpr int BW[7] = {18, 33, 49, 56, 49, 33, 18};
    for (int idx = tid; idx < TH * TW; idx += NTHREADS) {
      int ly = idx / TW, lx = idx - ly * TW;
      int gy = y0 + ly, gx = x0 + lx;
      int acc = 0;
#pragma unroll
      for (int k = 0; k < 7; ++k) acc += BW[k] * (int)s_h[(ly + k) * TW + lx];
      if (gy < H && gx < W) blur_out[((size_t)n * H + gy) * W + gx] = (uint8_t)((acc + 32768) >> 16);
    }
  }
  if (DETECT) {
    for (int idx = tid; idx < TH * TW; idx += NTHREADS) {
      int ly = idx / TW, lx = idx - ly * TW;
      int gy = y0 + ly, gx = x0 + lx;
      const uint8_t* c = &s_score[(ly + 1) * SCW + lx + 1];
      int s = c[0];
      bool keep = s > 0 && gy >= border && gy < H - border && gx >= border && gx < W - border;
      if (keep) {
        int m = max(max3i(c[-SCW - 1], c[-SCW], c[-SCW + 1]), max(c[-1], c[1]));
        m = max(m, max3i(c[SCW - 1], c[SCW], c[SCW + 1]));
        keep = s > m;  // strict maximum of its 8 neighbours
      }
      if (keep) {
        int p = atomicAdd(&s_cnt, 1);
        s_keys[p] = ((uint32_t)(255 - s) << VUS_KEY_POS_BITS) | (uint32_t)(gy * W + gx);
      }
    }
    __syncthreads();
    const int cnt = s_cnt;
    if (tid == 0 && cnt > 0) s_base = atomicAdd(&cand_count[n], cnt);
    __syncthreads();
    if (cnt > 0) {
      const int base = s_base;
      for (int i = tid; i < cnt; i += NTHREADS)
        if (base + i < cand_cap) cand_keys[(size_t)n * cand_cap + base + i] = s_keys[i];
    }
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
      if (tid == 0) {
        int k = s_k, cum = 0, b = 0;
        for (; b < 256; ++b) {
          int h = s_hist[b];
          if (cum + h >= k) break;
          cum += h;
        }
        s_prefix = prefix | ((uint32_t)b << shift);
        s_k = k - cum;
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
__global__ __launch_bounds__(256) void orient_rbrief_kernel(
    const uint8_t* __restrict__ img, const uint8_t* __restrict__ blur, int H, int W, int pitch,
    const uint32_t* __restrict__ kp_keys, const int* __restrict__ kp_count, int max_kp,
    uint64_t* __restrict__ desc_out, uint8_t* __restrict__ angle_out) {
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int n = blockIdx.y;
  const int i = blockIdx.x * 4 + wave;
  if (i >= max_kp) return;
  uint64_t* d = desc_out + ((size_t)n * max_kp + i) * 4;
  if (i >= kp_count[n]) {  // unused slot: defined contents
    if (lane < 4) d[lane] = 0;
    if (lane == 0) angle_out[(size_t)n * max_kp + i] = 0;
    return;
  }
  const uint32_t pos = kp_keys[(size_t)n * max_kp + i] & VUS_KEY_POS_MASK;
  const int y = (int)(pos / (uint32_t)W), x = (int)(pos - (uint32_t)y * (uint32_t)W);
  const uint8_t* im = img + (size_t)n * H * pitch;
  const uint8_t* bl = blur + (size_t)n * H * W;

  // intensity centroid: each lane sums a strided share of the 749 disc pixels
  int m10 = 0, m01 = 0;
  for (int k = lane; k < VUS_DISC_N; k += 64) {
    int dx = VUS_DISC_DX[k], dy = VUS_DISC_DY[k];
    int v = im[(size_t)clampi(y + dy, 0, H - 1) * pitch + clampi(x + dx, 0, W - 1)];
    m10 += dx * v;
    m01 += dy * v;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    m10 += __shfl_xor(m10, o);
    m01 += __shfl_xor(m01, o);
  }
  // nearest bin direction = largest projection, first maximum wins (integer, exact)
  long long pr = lane < VUS_N_ANGLE_BINS
                     ? (long long)m10 * VUS_ANGLE_COS[lane] + (long long)m01 * VUS_ANGLE_SIN[lane]
                     : (long long)(-0x7FFFFFFFFFFFFFFFll - 1);
  int bin = lane;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    long long opr = __shfl_xor(pr, o);
    int obin = __shfl_xor(bin, o);
    if (opr > pr || (opr == pr && obin < bin)) { pr = opr; bin = obin; }
  }
  const char4* pat = reinterpret_cast<const char4*>(VUS_RBRIEF_ROT) + (size_t)bin * 256;
  uint64_t word[4];
#pragma unroll
  for (int w = 0; w < 4; ++w) {
    char4 t = pat[w * 64 + lane];
    int a = bl[(size_t)clampi(y + t.y, 0, H - 1) * W + clampi(x + t.x, 0, W - 1)];
    int b = bl[(size_t)clampi(y + t.w, 0, H - 1) * W + clampi(x + t.z, 0, W - 1)];
    word[w] = __ballot(a < b);  // lane l supplies bit l of word w
  }
  if (lane == 0) {
    d[0] = word[0]; d[1] = word[1]; d[2] = word[2]; d[3] = word[3];
    angle_out[(size_t)n * max_kp + i] = (uint8_t)bin;
  }
}

// ---------------------------------------------------------------------------------------------
// Brute-force Hamming: one query per lane, train descriptors staged in LDS tiles.
constexpr int MT = 1024;  // train tile

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
        if (max_dy >= 0) {
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

int check_image_args(const void* img, int n_img, int H, int W, int pitch) {
  VUS_REQUIRE(img != nullptr, "image pointer is null");
  VUS_REQUIRE(n_img >= 0 && n_img <= 65535, "n_img=%d out of range [0, 65535]", n_img);
  VUS_REQUIRE(H >= 7 && W >= 7, "image %dx%d too small (need >= 7x7)", W, H);
  VUS_REQUIRE(pitch >= W, "pitch %d < W %d", pitch, W);
  VUS_REQUIRE((long long)H * W <= (1ll << VUS_KEY_POS_BITS), "H*W=%lld exceeds 2^24", (long long)H * W);
  return VUS_OK;
}

dim3 tile_grid(int n_img, int H, int W) { return dim3((W + TW - 1) / TW, (H + TH - 1) / TH, n_img); }

}  // namespace

extern "C" int vus_fast_score(const uint8_t* img, int n_img, int H, int W, int pitch, int thr,
                              uint8_t* score_out, void* stream) {
  if (int rc = check_image_args(img, n_img, H, W, pitch)) return rc;
  VUS_REQUIRE(score_out != nullptr, "score_out is null");
  VUS_REQUIRE(thr >= 1 && thr <= 254, "thr=%d out of range [1, 254]", thr);
  if (n_img == 0) return VUS_OK;
  fast_tile_kernel<true, false, false><<<tile_grid(n_img, H, W), NTHREADS, 0, vus::as_stream(stream)>>>(
      img, H, W, pitch, thr, 0, score_out, nullptr, nullptr, 0, nullptr);
  VUS_CHECK_LAUNCH("fast_score");
  return VUS_OK;
}

extern "C" int vus_blur7(const uint8_t* img, int n_img, int H, int W, int pitch, uint8_t* out, void* stream) {
  if (int rc = check_image_args(img, n_img, H, W, pitch)) return rc;
  VUS_REQUIRE(out != nullptr, "out is null");
  if (n_img == 0) return VUS_OK;
  fast_tile_kernel<false, false, true><<<tile_grid(n_img, H, W), NTHREADS, 0, vus::as_stream(stream)>>>(
      img, H, W, pitch, 1, 0, nullptr, out, nullptr, 0, nullptr);
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
  if (blur_out)
    fast_tile_kernel<false, true, true><<<tile_grid(n_img, H, W), NTHREADS, 0, st>>>(
        img, H, W, pitch, thr, border, nullptr, blur_out, cand_keys, cand_cap, cand_count);
  else
    fast_tile_kernel<false, true, false><<<tile_grid(n_img, H, W), NTHREADS, 0, st>>>(
        img, H, W, pitch, thr, border, nullptr, nullptr, cand_keys, cand_cap, cand_count);
  VUS_CHECK_LAUNCH("fast_detect");
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

extern "C" int vus_orient_rbrief(const uint8_t* img, const uint8_t* blur, int n_img, int H, int W, int pitch,
                                 const uint32_t* kp_keys, const int* kp_count, int max_kp,
                                 uint64_t* desc_out, uint8_t* angle_out, void* stream) {
  if (int rc = check_image_args(img, n_img, H, W, pitch)) return rc;
  VUS_REQUIRE(blur && kp_keys && kp_count && desc_out && angle_out, "null buffer");
  VUS_REQUIRE(max_kp >= 1, "max_kp=%d", max_kp);
  if (n_img == 0) return VUS_OK;
  dim3 grid((max_kp + 3) / 4, n_img);
  orient_rbrief_kernel<<<grid, 256, 0, vus::as_stream(stream)>>>(img, blur, H, W, pitch, kp_keys, kp_count,
                                                              max_kp, desc_out, angle_out);
  VUS_CHECK_LAUNCH("orient_rbrief");
  return VUS_OK;
}

extern "C" int vus_hamming_match(const uint64_t* desc, const uint32_t* kp_keys, const int* kp_count,
                                 int max_kp, int W, const int* q_index, const int* t_index, int n_pairs,
                                 int max_dy, int min_disp, int max_disp, int max_dist,
                                 int32_t* idx_out, int32_t* dist_out, void* stream) {
  VUS_REQUIRE(desc && kp_keys && kp_count && q_index && t_index && idx_out && dist_out, "null buffer");
  VUS_REQUIRE(max_kp >= 1 && W >= 1 && W < 65536, "max_kp=%d W=%d", max_kp, W);
  VUS_REQUIRE(n_pairs >= 0 && n_pairs <= 65535, "n_pairs=%d out of range [0, 65535]", n_pairs);
  if (n_pairs == 0) return VUS_OK;
  dim3 grid((max_kp + 255) / 256, n_pairs);
  hamming_match_kernel<<<grid, 256, 0, vus::as_stream(stream)>>>(desc, kp_keys, kp_count, max_kp, W, q_index,
                                                              t_index, max_dy, min_disp, max_disp, max_dist,
                                                              idx_out, dist_out);
  VUS_CHECK_LAUNCH("hamming_match");
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
