// structure.hip -- the block structure of the reduced camera system (vus_ba_structure) built on the device.
//
// What it replaces: the host-side construction that materialises every co-observation pair, sorts the 57 M of them
// (configs[2]) by block key and gathers their slots (ba_pack.build_structure, torch ops: 8 ms of the 41 ms drop-in
// call).  The reference has no counterpart -- GTSAM discovers the same structure inside
// LevenbergMarquardtOptimizer.optimize() (/root/reference/batch.py:337) by symbolic elimination.
//
// One workgroup per pose row i.  The pairs of the row are (s, b): s a P-order slot of pose i (landmark j), b an
// L-order index of the run point_ptr[j] .. L-index(s) of that landmark, i.e. of an observation (k, j) with k <= i.
// They are grouped by block offset d = i - k with a STABLE counting sort whose keys are never stored: a bit matrix
// bits[d][s] marks which slots have a partner at offset d, and the place of (s, d) inside block d is the number of
// set bits below s (popcounts).  The result is bit-identical to the host construction (blocks of a row by ascending
// k, pairs of a block by ascending slot), so the summation order of the Schur kernel -- and with it every result --
// does not depend on which of the two built the lists.
#include "vus_common.h"

namespace {

constexpr int SB_THREADS = 1024;
constexpr int SB_LDS_INTS = 30 * 1024;       // 120 KB of LDS for the per-row tables
constexpr int SB_WORDS_MAX = 32;             // slots of a row handled per pass = 32 * words

// exclusive prefix sums of (hist[e] > 0) and hist[e] over e = 0 .. nd-1 by the first wave; returns the totals
__device__ __forceinline__ void scan_row(const int* hist, int* brank, int* poff, int nd, int lane, int& n_blk,
                                         int& n_pair) {
  const int seg = (nd + 63) >> 6;
  const int e0 = min(lane * seg, nd), e1 = min(e0 + seg, nd);
  int nz = 0, np = 0;
  for (int e = e0; e < e1; ++e) {
    nz += hist[e] > 0;
    np += hist[e];
  }
  int inz = nz, inp = np;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const int a = __shfl_up(inz, off), b = __shfl_up(inp, off);
    if (lane >= off) {
      inz += a;
      inp += b;
    }
  }
  int rz = inz - nz, rp = inp - np;
  for (int e = e0; e < e1; ++e) {
    brank[e] = rz;
    poff[e] = rp;
    rz += hist[e] > 0;
    rp += hist[e];
  }
  n_blk = __shfl(inz, 63);
  n_pair = __shfl(inp, 63);
}

// The partners of a slot are the consecutive L-order entries lo .. a of its landmark; four at a time, so that the
// index loads of a step do not wait for one another.
#define SB_WALK_RUN(LO, A, BODY)                                                    \
  for (int b0_ = (LO); b0_ <= (A); b0_ += 4) {                                      \
    int kq_[4];                                                                     \
    _Pragma("unroll") for (int u_ = 0; u_ < 4; ++u_) kq_[u_] = P.obs_pose[min(b0_ + u_, (A))]; \
    _Pragma("unroll") for (int u_ = 0; u_ < 4; ++u_) {                              \
      const int b = b0_ + u_;                                                       \
      const int d = i - kq_[u_];                                                    \
      if (b <= (A) && d >= 0 && d < nd) { BODY }                                    \
    }                                                                               \
  }

// e = nd - 1 - d: ascending e = ascending k = the block order of the structure
template <bool FILL>
__global__ __launch_bounds__(SB_THREADS) void structure_rows_kernel(vus_ba_problem P, int nd, int words,
                                                                    const int* __restrict__ blk_base,
                                                                    const int* __restrict__ pair_base,
                                                                    int* __restrict__ row_blocks,
                                                                    int* __restrict__ row_pairs,
                                                                    int* __restrict__ blk_ptr, int* __restrict__ blk_i,
                                                                    int* __restrict__ blk_k, int* __restrict__ pair_a,
                                                                    int* __restrict__ pair_b) {
  extern __shared__ int sb[];
  int* hist = sb;                  // [nd] pairs of block e
  int* brank = hist + nd;          // [nd] rank of block e among the row's non-zero blocks
  int* poff = brank + nd;          // [nd] first pair of block e inside the row; advances pass by pass
  unsigned* bits = reinterpret_cast<unsigned*>(poff + nd);   // [nd][words]
  int* wpre = reinterpret_cast<int*>(bits + (size_t)nd * words);   // [nd][words] pairs of block e before word w
  const int i = blockIdx.x, tid = threadIdx.x;
  const int a0 = P.pose_ptr[i], a1 = P.pose_ptr[i + 1];
  for (int e = tid; e < nd; e += SB_THREADS) hist[e] = 0;
  __syncthreads();
  for (int s = a0 + tid; s < a1; s += SB_THREADS) {
    const int a = P.pobs_lidx[s];
    const int lo = P.point_ptr[P.obs_point[a]];
    SB_WALK_RUN(lo, a, atomicAdd(&hist[nd - 1 - d], 1);)
  }
  __syncthreads();
  if (tid < 64) {
    int n_blk, n_pair;
    scan_row(hist, brank, poff, nd, tid, n_blk, n_pair);
    if (!FILL && tid == 0) {
      row_blocks[i] = n_blk;
      row_pairs[i] = n_pair;
    }
  }
  if (!FILL) return;
  __syncthreads();
  const int qb = blk_base[i], pb = pair_base[i];
  for (int e = tid; e < nd; e += SB_THREADS)
    if (hist[e] > 0) {
      const int q = qb + brank[e];
      blk_i[q] = i;
      blk_k[q] = i - (nd - 1 - e);
      blk_ptr[q] = pb + poff[e];
    }
  if (i == P.n_poses - 1 && tid == 0) blk_ptr[blk_base[P.n_poses]] = pair_base[P.n_poses];
  const int span = 32 * words;
  for (int c0 = a0; c0 < a1; c0 += span) {
    const int n = min(span, a1 - c0);
    __syncthreads();               // the previous pass has read its tables
    for (int t = tid; t < nd * words; t += SB_THREADS) bits[t] = 0u;
    __syncthreads();
    for (int o = tid; o < n; o += SB_THREADS) {
      const int a = P.pobs_lidx[c0 + o];
      const int lo = P.point_ptr[P.obs_point[a]];
      const int w = o >> 5;
      const unsigned bit = 1u << (o & 31);
      SB_WALK_RUN(lo, a, atomicOr(&bits[(nd - 1 - d) * words + w], bit);)
    }
    __syncthreads();
    for (int e = tid; e < nd; e += SB_THREADS) {
      int run = poff[e];
      for (int w = 0; w < words; ++w) {
        wpre[e * words + w] = run;
        run += __popc(bits[e * words + w]);
      }
      poff[e] = run;
    }
    __syncthreads();
    for (int o = tid; o < n; o += SB_THREADS) {
      const int s = c0 + o;
      const int a = P.pobs_lidx[s];
      const int w = o >> 5;
      const unsigned below = (1u << (o & 31)) - 1u;
      // backwards from the slot's own observation: step t of neighbouring slots then lands in the same block more
      // often than not (d = 0 first, then mostly the previous keyframe, ...), i.e. at neighbouring places of the lists
      const int lo = P.point_ptr[P.obs_point[a]];
      for (int b0 = a; b0 >= lo; b0 -= 4) {
        int kq[4], pq[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int b = max(b0 - u, lo);
          kq[u] = P.obs_pose[b];
          pq[u] = P.obs_ppos[b];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int d = i - kq[u];
          if (b0 - u >= lo && d >= 0 && d < nd) {
            const int t = (nd - 1 - d) * words + w;
            const int pos = pb + wpre[t] + __popc(bits[t] & below);
            pair_a[pos] = s;
            pair_b[pos] = pq[u];
          }
        }
      }
    }
  }
}

int check(const vus_ba_problem* P, int band, int& nd, int& words) {
  VUS_REQUIRE(P != nullptr, "problem is null");
  VUS_REQUIRE(P->n_poses >= 1 && P->n_obs >= 1 && P->n_points >= 1, "bad sizes: poses=%d points=%d obs=%d", P->n_poses,
              P->n_points, P->n_obs);
  VUS_REQUIRE(P->obs_pose && P->obs_point && P->point_ptr && P->obs_ppos && P->pose_ptr && P->pobs_lidx,
              "observation arrays are null");
  VUS_REQUIRE(band >= 0 && band < P->n_poses, "band=%d", band);
  nd = band + 1;
  words = (SB_LDS_INTS - 3 * nd) / (2 * nd);
  if (words > SB_WORDS_MAX) words = SB_WORDS_MAX;
  VUS_REQUIRE(words >= 1, "band of %d poses is too wide for the device structure builder (limit %d)", band,
              SB_LDS_INTS / 5 - 1);
  return VUS_OK;
}

template <bool FILL>
int prepare(int lds_bytes) {
  // The attribute belongs to (kernel, device): it is set on every call (a host-side table write, no device work)
  // rather than cached in a process-wide static that a second device or a second thread would read wrongly.
  VUS_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(structure_rows_kernel<FILL>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes));
  return VUS_OK;
}

}  // namespace

extern "C" int vus_ba_structure_count(const vus_ba_problem* P, int band, int* row_blocks, int* row_pairs,
                                      void* stream) {
  int nd, words;
  if (int rc = check(P, band, nd, words)) return rc;
  VUS_REQUIRE(row_blocks && row_pairs, "null buffer");
  const int lds = 3 * nd * (int)sizeof(int);
  if (int rc = prepare<false>(lds)) return rc;
  structure_rows_kernel<false><<<P->n_poses, SB_THREADS, lds, vus::as_stream(stream)>>>(
      *P, nd, words, nullptr, nullptr, row_blocks, row_pairs, nullptr, nullptr, nullptr, nullptr, nullptr);
  VUS_CHECK_LAUNCH("ba_structure_count");
  return VUS_OK;
}

extern "C" int vus_ba_structure_fill(const vus_ba_problem* P, int band, const int* blk_base, const int* pair_base,
                                     int* blk_ptr, int* blk_i, int* blk_k, int* pair_a, int* pair_b, void* stream) {
  int nd, words;
  if (int rc = check(P, band, nd, words)) return rc;
  VUS_REQUIRE(blk_base && pair_base && blk_ptr && blk_i && blk_k && pair_a && pair_b, "null buffer");
  const int lds = (3 * nd + 2 * nd * words) * (int)sizeof(int);
  if (int rc = prepare<true>(lds)) return rc;
  structure_rows_kernel<true><<<P->n_poses, SB_THREADS, lds, vus::as_stream(stream)>>>(
      *P, nd, words, blk_base, pair_base, nullptr, nullptr, blk_ptr, blk_i, blk_k, pair_a, pair_b);
  VUS_CHECK_LAUNCH("ba_structure_fill");
  return VUS_OK;
}
