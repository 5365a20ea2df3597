// pack.hip -- graph packing on the GPU: what the reference's batch_create leaves as Python lists of factor objects
// (/root/reference/batch.py:295-305) becomes the structure-of-arrays problem of include/vus.h here, without a host pass
// and without torch's index operators (whose lazily loaded code objects made the FIRST optimize() of a process --
// batch.py:337 calls it exactly once -- six times slower than a warm one).
//
//   vus_keys_to_indices        gtsam keys L(id) of every observation -> compact landmark indices + the sorted unique keys
//   vus_lookup_keys            pose keys X(i) of every observation -> rows of the (sorted) pose table, misses reported
//   vus_ba_pack_observations   (pose, point, measurement) rows in any order -> L-order / P-order arrays, both
//                              permutations and both pointer arrays of vus_ba_problem
//   vus_exclusive_scan_i32     row counts -> offsets + 64-bit total (the structure builder's list sizes)
//
// The sorts are a stable LSD radix sort written here (8-bit digits: per-tile histograms, one scan, a stable scatter that
// ranks equal digits with wave ballots).  Round 3 first used rocPRIM's device radix sort: the same speed per call (0.50 ms
// against 0.56 ms for a configs[2] graph), but its template instantiations made this translation unit's code object 6.8 MB -- 4 MB of
// it mangled names -- and LOADING it cost the first optimize() of a process 23 ms (tools/cold_pack_probe.py); batch.py:337
// calls optimize() exactly once per process.  All results stay on the device.
#include <cstring>
#include "vus_common.h"

namespace {

inline int cdiv(long long a, int b) { return (int)((a + b - 1) / b); }
inline size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

__global__ void iota_kernel(int* __restrict__ a, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) a[i] = i;
}

// rank[i] = inclusive scan of head: the key of sorted position i is the (rank[i] - 1)-th distinct key
__global__ void scatter_ranks_kernel(const unsigned long long* __restrict__ sk, const int* __restrict__ sp,
                                     const int* __restrict__ rank, int n, int* __restrict__ idx_out,
                                     long long* __restrict__ uniq_out, int* __restrict__ n_unique) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int r = rank[i] - 1;
  idx_out[sp[i]] = r;
  if (i == 0 || sk[i] != sk[i - 1]) uniq_out[r] = (long long)sk[i];
  if (i == n - 1) n_unique[0] = r + 1;
}

__device__ __forceinline__ int lower_bound_u64(const unsigned long long* __restrict__ a, int n, unsigned long long key) {
  int lo = 0, hi = n;
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if (a[mid] < key) lo = mid + 1;
    else hi = mid;
  }
  return lo;
}

__device__ __forceinline__ int lower_bound_i32(const int* __restrict__ a, int n, int key) {
  int lo = 0, hi = n;
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if (a[mid] < key) lo = mid + 1;
    else hi = mid;
  }
  return lo;
}

__global__ void lookup_kernel(const long long* __restrict__ sorted_keys, int m, const long long* __restrict__ q, int n,
                              int* __restrict__ idx_out, int* __restrict__ first_miss) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const long long key = q[i];
  int lo = 0, hi = m;
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if (sorted_keys[mid] < key) lo = mid + 1;
    else hi = mid;
  }
  const bool hit = lo < m && sorted_keys[lo] == key;
  idx_out[i] = hit ? lo : -1;
  if (!hit) atomicMin(first_miss, i);
}

__global__ void make_keys_kernel(const int* __restrict__ obs_pose, const int* __restrict__ obs_point, int n, int n_poses,
                                 int n_points, unsigned long long* __restrict__ key, int* __restrict__ flags) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int p = obs_pose[i], l = obs_point[i];
  if (p < 0 || p >= n_poses || l < 0 || l >= n_points) atomicOr(flags, 2);       // index out of range
  key[i] = (unsigned long long)(l < 0 ? 0 : l) * (unsigned long long)n_poses + (unsigned long long)(p < 0 ? 0 : p);
}

// L-order rows from the sorted keys and the permutation
__global__ void gather_L_kernel(const unsigned long long* __restrict__ sk, const int* __restrict__ perm,
                                const double* __restrict__ meas, int n, int n_poses, double* __restrict__ meas_L,
                                int* __restrict__ obs_pose_L, int* __restrict__ obs_point_L, int* __restrict__ flags) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const unsigned long long k = sk[i];
  if (i > 0 && sk[i - 1] == k) atomicOr(flags, 1);      // two factors between the same pose and landmark
  const int l = (int)(k / (unsigned long long)n_poses);
  obs_point_L[i] = l;
  obs_pose_L[i] = (int)(k - (unsigned long long)l * (unsigned long long)n_poses);
  const double* m = meas + 3 * (size_t)perm[i];
  meas_L[3 * (size_t)i] = m[0];
  meas_L[3 * (size_t)i + 1] = m[1];
  meas_L[3 * (size_t)i + 2] = m[2];
}

__global__ void point_ptr_kernel(const unsigned long long* __restrict__ sk, int n, int n_poses, int n_points,
                                 int* __restrict__ point_ptr) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j <= n_points) point_ptr[j] = lower_bound_u64(sk, n, (unsigned long long)j * (unsigned long long)n_poses);
}

// widest keyframe span of a landmark = half-bandwidth (in pose blocks) of the reduced camera system
__global__ void band_kernel(const int* __restrict__ point_ptr, const int* __restrict__ obs_pose_L, int n_points,
                            int* __restrict__ band) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n_points) return;
  const int a0 = point_ptr[j], a1 = point_ptr[j + 1];
  if (a1 > a0) atomicMax(band, obs_pose_L[a1 - 1] - obs_pose_L[a0]);
}

__global__ void pose_ptr_kernel(const int* __restrict__ sorted_pose, int n, int n_poses, int* __restrict__ pose_ptr) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i <= n_poses) pose_ptr[i] = lower_bound_i32(sorted_pose, n, i);
}

__global__ void inverse_perm_kernel(const int* __restrict__ pobs_lidx, int n, int* __restrict__ obs_ppos) {
  const int s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s < n) obs_ppos[pobs_lidx[s]] = s;
}

// out[i] = sum of in[0 .. i) for i <= n (32-bit offsets), total[0] = the sum in 64 bits (overflow is the caller's check)
__global__ __launch_bounds__(1024) void scan_i32_kernel(const int* __restrict__ in, int n, int* __restrict__ out,
                                                        long long* __restrict__ total) {
  __shared__ long long s_part[1024];
  const int tid = threadIdx.x;
  const int per = (n + 1023) / 1024;
  long long local = 0;
  for (int u = 0; u < per; ++u) {
    const int i = tid * per + u;
    if (i < n) local += in[i];
  }
  s_part[tid] = local;
  __syncthreads();
  for (int o = 1; o < 1024; o <<= 1) {
    const long long v = tid >= o ? s_part[tid - o] : 0;
    __syncthreads();
    s_part[tid] += v;
    __syncthreads();
  }
  long long run = s_part[tid] - local;
  for (int u = 0; u < per; ++u) {
    const int i = tid * per + u;
    if (i < n) {
      out[i] = (int)run;
      run += in[i];
    }
  }
  if (tid == 1023) {
    out[n] = (int)s_part[1023];
    total[0] = s_part[1023];
  }
}

// ---- stable LSD radix sort of (key, int value) pairs, 8 bits per pass -------------------------------------------------
// A tile = RS_TILE consecutive elements handled by one workgroup in both kernels of a pass, so that the histogram's
// counts are exactly what the scatter places.  counts / offsets are laid out [digit][tile]: one exclusive scan over the
// whole table gives every (digit, tile) its first output position, and the order digit-major / tile-minor / element
// order inside the tile is what makes the pass stable.
constexpr int RS_THREADS = 256, RS_ITEMS = 16, RS_TILE = RS_THREADS * RS_ITEMS;

// vary |= key ^ key[0] over all keys: a pass whose digit is the same in every key (the upper bytes of gtsam keys
// chr << 56 | index) has nothing to reorder -- its kernels see that in `vary` and the scatter degenerates to a copy.
template <class Key>
__global__ __launch_bounds__(RS_THREADS) void rs_vary_kernel(const Key* __restrict__ keys, int n, unsigned long long* __restrict__ vary) {
  const Key k0 = keys[0];
  unsigned long long v = 0;
  for (long long idx = (long long)blockIdx.x * RS_THREADS + threadIdx.x; idx < n; idx += (long long)gridDim.x * RS_THREADS)
    v |= (unsigned long long)(keys[idx] ^ k0);
  for (int o = 32; o > 0; o >>= 1) v |= __shfl_down(v, o, 64);
  if ((threadIdx.x & 63) == 0 && v != 0) atomicOr(vary, v);
}

__device__ __forceinline__ bool rs_skip(const unsigned long long* vary, int shift, unsigned mask) {
  return ((unsigned)(vary[0] >> shift) & mask) == 0;
}

template <class Key>
__global__ __launch_bounds__(RS_THREADS) void rs_hist_kernel(const Key* __restrict__ keys, int n, int shift, unsigned mask,
                                                             int* __restrict__ counts, int n_tiles,
                                                             const unsigned long long* __restrict__ vary) {
  if (rs_skip(vary, shift, mask)) return;
  __shared__ int h[256];
  const int tid = threadIdx.x;
  h[tid] = 0;
  __syncthreads();
  const long long base = (long long)blockIdx.x * RS_TILE;
#pragma unroll 4
  for (int i = 0; i < RS_ITEMS; ++i) {
    const long long idx = base + i * RS_THREADS + tid;
    if (idx < n) atomicAdd(&h[(unsigned)(keys[idx] >> shift) & mask], 1);
  }
  __syncthreads();
  counts[tid * n_tiles + blockIdx.x] = h[tid];
}

// workgroup d: counts[d][0 .. n_tiles) -> exclusive prefix over the tiles, in place; totals[d] = the digit's count
__global__ __launch_bounds__(RS_THREADS) void rs_scan_kernel(int* __restrict__ counts, int n_tiles, int* __restrict__ totals, int shift,
                                                             unsigned mask, const unsigned long long* __restrict__ vary) {
  if (rs_skip(vary, shift, mask)) return;
  __shared__ int s_t[RS_THREADS];
  const int tid = threadIdx.x;
  int* row = counts + (size_t)blockIdx.x * n_tiles;
  const int per = (n_tiles + RS_THREADS - 1) / RS_THREADS;
  int local = 0;
  for (int u = 0; u < per; ++u) {
    const int i = tid * per + u;
    if (i < n_tiles) local += row[i];
  }
  s_t[tid] = local;
  __syncthreads();
  for (int o = 1; o < RS_THREADS; o <<= 1) {
    const int v = tid >= o ? s_t[tid - o] : 0;
    __syncthreads();
    s_t[tid] += v;
    __syncthreads();
  }
  int run = s_t[tid] - local;
  for (int u = 0; u < per; ++u) {
    const int i = tid * per + u;
    if (i < n_tiles) {
      const int c = row[i];
      row[i] = run;
      run += c;
    }
  }
  if (tid == RS_THREADS - 1) totals[blockIdx.x] = s_t[tid];
}

template <class Key>
__global__ __launch_bounds__(RS_THREADS) void rs_scatter_kernel(const Key* __restrict__ kin, const int* __restrict__ vin,
                                                                Key* __restrict__ kout, int* __restrict__ vout, int n, int shift,
                                                                unsigned mask, const int* __restrict__ tile_prefix,
                                                                const int* __restrict__ totals, int n_tiles,
                                                                const unsigned long long* __restrict__ vary) {
  __shared__ int s_next[256];                    // output position of the tile's next element of each digit
  __shared__ int s_wc[RS_THREADS / 64][256];     // elements of each digit in each wave of the current row of 256
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const long long base = (long long)blockIdx.x * RS_TILE;
  if (rs_skip(vary, shift, mask)) {              // nothing to reorder: the tile is copied
    for (int i = 0; i < RS_ITEMS; ++i) {
      const long long idx = base + i * RS_THREADS + tid;
      if (idx < n) {
        kout[idx] = kin[idx];
        vout[idx] = vin[idx];
      }
    }
    return;
  }
  // first position of digit d = number of elements with a smaller digit: exclusive scan of the 256 totals
  const int tot = totals[tid];
  s_next[tid] = tot;
  __syncthreads();
  for (int o = 1; o < 256; o <<= 1) {
    const int v = tid >= o ? s_next[tid - o] : 0;
    __syncthreads();
    s_next[tid] += v;
    __syncthreads();
  }
  const int first = s_next[tid] - tot;
  __syncthreads();
  s_next[tid] = first + tile_prefix[tid * n_tiles + blockIdx.x];
#pragma unroll
  for (int w = 0; w < RS_THREADS / 64; ++w) s_wc[w][tid] = 0;
  __syncthreads();
  const unsigned long long lt = lane == 0 ? 0ull : (~0ull >> (64 - lane));
  for (int i = 0; i < RS_ITEMS; ++i) {
    const long long idx = base + i * RS_THREADS + tid;
    const bool valid = idx < n;
    const Key key = valid ? kin[idx] : (Key)0;
    const int val = valid ? vin[idx] : 0;
    const unsigned d = (unsigned)(key >> shift) & mask;
    // the lanes of this wave holding the same digit (eight ballots), this lane's place among them
    unsigned long long same = __ballot(valid);
#pragma unroll
    for (int b = 0; b < 8; ++b) {
      const bool bit = (d >> b) & 1u;
      const unsigned long long bal = __ballot(bit);
      same &= bit ? bal : ~bal;
    }
    const int rank_w = __popcll(same & lt);
    if (valid && rank_w == 0) s_wc[wave][d] = __popcll(same);
    __syncthreads();
    if (valid) {
      int before = 0;
#pragma unroll
      for (int w = 0; w < RS_THREADS / 64; ++w) before += w < wave ? s_wc[w][d] : 0;
      const int pos = s_next[d] + before + rank_w;
      kout[pos] = key;
      vout[pos] = val;
    }
    __syncthreads();
    int row = 0;
#pragma unroll
    for (int w = 0; w < RS_THREADS / 64; ++w) {
      row += s_wc[w][tid];
      s_wc[w][tid] = 0;
    }
    s_next[tid] += row;
    __syncthreads();
  }
}

inline int rs_tiles(int n) { return n > 0 ? (n + RS_TILE - 1) / RS_TILE : 1; }
inline int rs_passes(int end_bit) { return end_bit > 0 ? (end_bit + 7) / 8 : 1; }

// scratch of one sort: a key array, a value array (ping-pong partners of the outputs), the count table [256][tiles], the
// digit totals and the word of varying key bits
struct SortTemp {
  void* keys;
  int* vals;
  int* counts;
  int* offsets;          // [0, 256): digit totals; the rest: scratch of the callers' own scans
  long long* total;      // the word of varying key bits during a sort; a scan's 64-bit total otherwise
};

// Sorts the low end_bit bits of the keys, stable.  The inputs are only read; the last pass writes (kout, vout).
template <class Key>
int radix_sort_pairs(const Key* kin, Key* kout, const int* vin, int* vout, int n, int end_bit, const SortTemp& t, hipStream_t st) {
  const int passes = rs_passes(end_bit), nt = rs_tiles(n);
  Key* ktmp = static_cast<Key*>(t.keys);
  unsigned long long* vary = reinterpret_cast<unsigned long long*>(t.total);
  VUS_CHECK_HIP(hipMemsetAsync(vary, 0, sizeof(unsigned long long), st));
  rs_vary_kernel<Key><<<nt < 1024 ? nt : 1024, RS_THREADS, 0, st>>>(kin, n, vary);
  const Key* ks = kin;
  const int* vs = vin;
  for (int p = 0; p < passes; ++p) {
    const bool to_out = ((passes - 1 - p) & 1) == 0;
    Key* kd = to_out ? kout : ktmp;
    int* vd = to_out ? vout : t.vals;
    const int bits = end_bit - 8 * p < 8 ? end_bit - 8 * p : 8;
    const unsigned mask = (1u << bits) - 1u;
    rs_hist_kernel<Key><<<nt, RS_THREADS, 0, st>>>(ks, n, 8 * p, mask, t.counts, nt, vary);
    rs_scan_kernel<<<256, RS_THREADS, 0, st>>>(t.counts, nt, t.offsets, 8 * p, mask, vary);
    rs_scatter_kernel<Key><<<nt, RS_THREADS, 0, st>>>(ks, vs, kd, vd, n, 8 * p, mask, t.counts, t.offsets, nt, vary);
    ks = kd;
    vs = vd;
  }
  VUS_CHECK_LAUNCH("radix_sort_pairs");
  return VUS_OK;
}

// ---- rank of every sorted position among the distinct keys: inclusive scan of the head flags, tile sums first ----------
__global__ __launch_bounds__(RS_THREADS) void head_sums_kernel(const unsigned long long* __restrict__ sk, int n, int* __restrict__ tile_sum) {
  __shared__ int s_w[RS_THREADS / 64];
  const int tid = threadIdx.x;
  const long long base = (long long)blockIdx.x * RS_TILE;
  int c = 0;
#pragma unroll 4
  for (int i = 0; i < RS_ITEMS; ++i) {
    const long long idx = base + i * RS_THREADS + tid;
    if (idx < n) c += (idx == 0 || sk[idx] != sk[idx - 1]) ? 1 : 0;
  }
  for (int o = 32; o > 0; o >>= 1) c += __shfl_down(c, o, 64);
  if ((tid & 63) == 0) s_w[tid >> 6] = c;
  __syncthreads();
  if (tid == 0) {
    int t = 0;
    for (int w = 0; w < RS_THREADS / 64; ++w) t += s_w[w];
    tile_sum[blockIdx.x] = t;
  }
}

// rank[i] = number of heads in [0, i]; a thread owns RS_ITEMS CONSECUTIVE elements of the tile
__global__ __launch_bounds__(RS_THREADS) void head_ranks_kernel(const unsigned long long* __restrict__ sk, int n,
                                                                const int* __restrict__ tile_off, int* __restrict__ rank) {
  __shared__ int s_t[RS_THREADS];
  const int tid = threadIdx.x;
  const long long first = (long long)blockIdx.x * RS_TILE + (long long)tid * RS_ITEMS;
  int c = 0;
  for (int i = 0; i < RS_ITEMS; ++i) {
    const long long idx = first + i;
    if (idx < n) c += (idx == 0 || sk[idx] != sk[idx - 1]) ? 1 : 0;
  }
  s_t[tid] = c;
  __syncthreads();
  for (int o = 1; o < RS_THREADS; o <<= 1) {
    const int v = tid >= o ? s_t[tid - o] : 0;
    __syncthreads();
    s_t[tid] += v;
    __syncthreads();
  }
  int run = tile_off[blockIdx.x] + s_t[tid] - c;
  for (int i = 0; i < RS_ITEMS; ++i) {
    const long long idx = first + i;
    if (idx < n) {
      run += (idx == 0 || sk[idx] != sk[idx - 1]) ? 1 : 0;
      rank[idx] = run;
    }
  }
}


// ---- tile-pair structure of the landmark elimination (vus_ba_tiles, include/vus.h) ------------------------------------
// A landmark's L-order run (poses ascending) falls into SEGMENTS of equal pose tile (pose >> 3); every pair of segments
// (u >= v) is one entry of the unit (tile of u, tile distance): the landmark's rows of the two tiles, as first L-order row
// + 8-bit pose mask each.  Entries are emitted landmark by landmark and sorted by unit with the stable radix sort above:
// inside a unit they ascend by landmark, whatever the launch geometry -- the summation order of the Schur kernel is fixed.
constexpr int TILE_POSES = 8;

__global__ void tiles_count_kernel(const int* __restrict__ point_ptr, const int* __restrict__ obs_pose, int n_points,
                                   int* __restrict__ cnt) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n_points) return;
  int m = 0, prev = -1;
  for (int a = point_ptr[j]; a < point_ptr[j + 1]; ++a) {
    const int t = obs_pose[a] / TILE_POSES;
    m += t != prev;
    prev = t;
  }
  cnt[j] = m * (m + 1) / 2;
}

__global__ void tiles_emit_kernel(const int* __restrict__ point_ptr, const int* __restrict__ obs_pose, int n_points, int dt1,
                                  const int* __restrict__ base, unsigned* __restrict__ key, int4* __restrict__ ent) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n_points) return;
  const int a0 = point_ptr[j], a1 = point_ptr[j + 1];
  int out = base[j];
  for (int a = a0; a < a1;) {
    const int tu = obs_pose[a] / TILE_POSES, au = a;
    int mu = 0;
    for (; a < a1 && obs_pose[a] / TILE_POSES == tu; ++a) mu |= 1 << (obs_pose[a] % TILE_POSES);
    for (int b = a0; b < a;) {                      // the segments up to and including u, in order
      const int tv = obs_pose[b] / TILE_POSES, bv = b;
      int mv = 0;
      for (; b < a1 && obs_pose[b] / TILE_POSES == tv; ++b) mv |= 1 << (obs_pose[b] % TILE_POSES);
      key[out] = (unsigned)(tu * dt1 + (tu - tv));
      ent[out] = make_int4(au, bv, j, mu | (mv << 8));
      ++out;
    }
  }
}

__global__ void tiles_gather_kernel(const int4* __restrict__ ent, const int* __restrict__ perm, int n, int4* __restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = ent[perm[i]];
}

// unit_ptr[u] = first sorted entry with key >= u (u = 0 .. n_units)
__global__ void tiles_ptr_kernel(const unsigned* __restrict__ sk, int n, int n_units, int* __restrict__ unit_ptr) {
  const int u = blockIdx.x * blockDim.x + threadIdx.x;
  if (u > n_units) return;
  int lo = 0, hi = n;
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if (sk[mid] < (unsigned)u) lo = mid + 1;
    else hi = mid;
  }
  unit_ptr[u] = lo;
}

// order[] = the units by descending size class (bit length of the entry count): the persistent workgroups of the Schur
// kernel take them largest first.  Only the schedule depends on it, no result.  One workgroup.
__global__ __launch_bounds__(1024) void tiles_order_kernel(const int* __restrict__ unit_ptr, int n_units, int* __restrict__ order) {
  __shared__ int s_cls[33];
  const int tid = threadIdx.x;
  if (tid < 33) s_cls[tid] = 0;
  __syncthreads();
  for (int u = tid; u < n_units; u += 1024) {
    const int c = unit_ptr[u + 1] - unit_ptr[u];
    atomicAdd(&s_cls[c > 0 ? 31 - __clz(c) + 1 : 0], 1);
  }
  __syncthreads();
  if (tid == 0) {
    int run = 0;
    for (int c = 32; c >= 0; --c) { const int n = s_cls[c]; s_cls[c] = run; run += n; }
  }
  __syncthreads();
  for (int u = tid; u < n_units; u += 1024) {
    const int c = unit_ptr[u + 1] - unit_ptr[u];
    order[atomicAdd(&s_cls[c > 0 ? 31 - __clz(c) + 1 : 0], 1)] = u;
  }
}

int bits_for(unsigned long long max_value) {
  int b = 1;
  while (b < 64 && (max_value >> b) != 0) ++b;
  return b;
}

struct Carve {
  char* p;
  size_t left;
  template <class T>
  T* take(size_t n) {
    const size_t bytes = align256(n * sizeof(T));
    if (bytes > left) return nullptr;
    T* r = reinterpret_cast<T*>(p);
    p += bytes;
    left -= bytes;
    return r;
  }
};

// bytes of a SortTemp for n elements (carved by take_sort_temp)
size_t sort_temp_bytes(int n) {
  const size_t nt = (size_t)rs_tiles(n);
  return align256(8 * (size_t)n) + align256(4 * (size_t)n) + 2 * align256(4 * (256 * nt + 1)) + align256(8);
}

bool take_sort_temp(Carve& c, int n, SortTemp& t) {
  const size_t nt = (size_t)rs_tiles(n);
  t.keys = c.take<unsigned long long>(n);
  t.vals = c.take<int>(n);
  t.counts = c.take<int>(256 * nt + 1);
  t.offsets = c.take<int>(256 * nt + 1);
  t.total = c.take<long long>(1);
  return t.keys && t.vals && t.counts && t.offsets && t.total;
}

}  // namespace

extern "C" long long vus_pack_work_bytes(int n) {
  if (n < 0) return 0;
  const size_t nn = (size_t)(n > 0 ? n : 1);
  // two 8-byte key arrays, four 4-byte index arrays, the sort's own temporary storage
  return (long long)(2 * align256(8 * nn) + 4 * align256(4 * nn) + sort_temp_bytes(n > 0 ? n : 1) + 1024);
}

extern "C" int vus_keys_to_indices(const int64_t* keys, int n, int* idx_out, int64_t* uniq_out, int* n_unique, void* work,
                                   long long work_bytes, void* stream) {
  VUS_REQUIRE(n >= 0, "n=%d", n);
  VUS_REQUIRE(n_unique != nullptr, "n_unique is null");
  hipStream_t st = vus::as_stream(stream);
  if (n == 0) {
    VUS_CHECK_HIP(hipMemsetAsync(n_unique, 0, sizeof(int), st));
    return VUS_OK;
  }
  VUS_REQUIRE(keys && idx_out && uniq_out && work, "null buffer");
  VUS_REQUIRE(work_bytes >= vus_pack_work_bytes(n), "workspace of %lld bytes, %lld needed", work_bytes, vus_pack_work_bytes(n));
  Carve c{static_cast<char*>(work), (size_t)work_bytes};
  unsigned long long* sk = c.take<unsigned long long>(n);
  c.take<unsigned long long>(n);
  int* iota = c.take<int>(n);
  int* sp = c.take<int>(n);
  int* tile_sum = c.take<int>(n);          // rs_tiles(n) + 1 entries used of each
  int* rank = c.take<int>(n);
  SortTemp tmp;
  VUS_REQUIRE(sk && iota && sp && tile_sum && rank && take_sort_temp(c, n, tmp), "workspace too small");
  iota_kernel<<<cdiv(n, 256), 256, 0, st>>>(iota, n);
  // gtsam keys are non-negative (chr << 56 | index): their unsigned order is their order
  if (int rc = radix_sort_pairs<unsigned long long>(reinterpret_cast<const unsigned long long*>(keys), sk, iota, sp, n, 64, tmp, st))
    return rc;
  const int nt = rs_tiles(n);
  head_sums_kernel<<<nt, RS_THREADS, 0, st>>>(sk, n, tile_sum);
  scan_i32_kernel<<<1, 1024, 0, st>>>(tile_sum, nt, tmp.offsets, tmp.total);
  head_ranks_kernel<<<nt, RS_THREADS, 0, st>>>(sk, n, tmp.offsets, rank);
  scatter_ranks_kernel<<<cdiv(n, 256), 256, 0, st>>>(sk, sp, rank, n, idx_out, reinterpret_cast<long long*>(uniq_out), n_unique);
  VUS_CHECK_LAUNCH("keys_to_indices");
  return VUS_OK;
}

extern "C" int vus_lookup_keys(const int64_t* sorted_keys, int m, const int64_t* queries, int n, int* idx_out,
                               int* first_miss, void* stream) {
  VUS_REQUIRE(m >= 0 && n >= 0, "m=%d n=%d", m, n);
  VUS_REQUIRE(first_miss != nullptr, "first_miss is null");
  hipStream_t st = vus::as_stream(stream);
  VUS_CHECK_HIP(hipMemsetAsync(first_miss, 0x7F, sizeof(int), st));      // 0x7F7F7F7F: "no miss"
  if (n == 0) return VUS_OK;
  VUS_REQUIRE((sorted_keys || m == 0) && queries && idx_out, "null buffer");
  lookup_kernel<<<cdiv(n, 256), 256, 0, st>>>(reinterpret_cast<const long long*>(sorted_keys), m,
                                              reinterpret_cast<const long long*>(queries), n, idx_out, first_miss);
  VUS_CHECK_LAUNCH("lookup_keys");
  return VUS_OK;
}

extern "C" int vus_ba_pack_observations(const int* obs_pose, const int* obs_point, const double* meas, int n_obs,
                                        int n_poses, int n_points, double* meas_L, int* obs_pose_L, int* obs_point_L,
                                        int* point_ptr, int* obs_ppos, int* pose_ptr, int* pobs_lidx, int* perm, int* flags,
                                        int* band, void* work, long long work_bytes, void* stream) {
  VUS_REQUIRE(n_obs >= 0 && n_poses >= 1 && n_points >= 0, "n_obs=%d n_poses=%d n_points=%d", n_obs, n_poses, n_points);
  VUS_REQUIRE(point_ptr && pose_ptr && flags, "null buffer");
  hipStream_t st = vus::as_stream(stream);
  VUS_CHECK_HIP(hipMemsetAsync(flags, 0, sizeof(int), st));
  if (band) VUS_CHECK_HIP(hipMemsetAsync(band, 0, sizeof(int), st));
  if (n_obs == 0) {
    VUS_CHECK_HIP(hipMemsetAsync(point_ptr, 0, sizeof(int) * (size_t)(n_points + 1), st));
    VUS_CHECK_HIP(hipMemsetAsync(pose_ptr, 0, sizeof(int) * (size_t)(n_poses + 1), st));
    return VUS_OK;
  }
  VUS_REQUIRE(obs_pose && obs_point && meas && meas_L && obs_pose_L && obs_point_L && obs_ppos && pobs_lidx && perm && work,
              "null buffer");
  VUS_REQUIRE(work_bytes >= vus_pack_work_bytes(n_obs), "workspace of %lld bytes, %lld needed", work_bytes,
              vus_pack_work_bytes(n_obs));
  const int n = n_obs;
  Carve c{static_cast<char*>(work), (size_t)work_bytes};
  unsigned long long* key = c.take<unsigned long long>(n);
  unsigned long long* sk = c.take<unsigned long long>(n);
  int* iota = c.take<int>(n);
  int* spose = c.take<int>(n);
  c.take<int>(n);
  c.take<int>(n);
  SortTemp tmp;
  VUS_REQUIRE(key && sk && iota && spose && take_sort_temp(c, n, tmp), "workspace too small");
  iota_kernel<<<cdiv(n, 256), 256, 0, st>>>(iota, n);
  make_keys_kernel<<<cdiv(n, 256), 256, 0, st>>>(obs_pose, obs_point, n, n_poses, n_points, key, flags);
  // L-order: by (point, pose).  Keys are unique in a valid graph, so the order is total.
  const int kbits = bits_for((unsigned long long)(n_points > 0 ? n_points : 1) * (unsigned long long)n_poses);
  if (int rc = radix_sort_pairs<unsigned long long>(key, sk, iota, perm, n, kbits, tmp, st)) return rc;
  gather_L_kernel<<<cdiv(n, 256), 256, 0, st>>>(sk, perm, meas, n, n_poses, meas_L, obs_pose_L, obs_point_L, flags);
  point_ptr_kernel<<<cdiv(n_points + 1, 256), 256, 0, st>>>(sk, n, n_poses, n_points, point_ptr);
  if (band && n_points > 0) band_kernel<<<cdiv(n_points, 256), 256, 0, st>>>(point_ptr, obs_pose_L, n_points, band);
  // P-order: a STABLE sort of the L-order rows by pose keeps the points ascending inside every pose
  if (int rc = radix_sort_pairs<unsigned int>(reinterpret_cast<const unsigned int*>(obs_pose_L), reinterpret_cast<unsigned int*>(spose),
                                              iota, pobs_lidx, n, bits_for((unsigned long long)n_poses), tmp, st))
    return rc;
  inverse_perm_kernel<<<cdiv(n, 256), 256, 0, st>>>(pobs_lidx, n, obs_ppos);
  pose_ptr_kernel<<<cdiv(n_poses + 1, 256), 256, 0, st>>>(spose, n, n_poses, pose_ptr);
  VUS_CHECK_LAUNCH("ba_pack_observations");
  return VUS_OK;
}

extern "C" int vus_exclusive_scan_i32(const int* in, int n, int* out, long long* total, void* stream) {
  VUS_REQUIRE(n >= 0 && out != nullptr && total != nullptr, "n=%d", n);
  VUS_REQUIRE(in != nullptr || n == 0, "null buffer");
  // n is a number of poses (thousands): one workgroup
  scan_i32_kernel<<<1, 1024, 0, vus::as_stream(stream)>>>(in, n, out, total);
  VUS_CHECK_LAUNCH("exclusive_scan_i32");
  return VUS_OK;
}

extern "C" int vus_ba_tiles_count(const vus_ba_problem* P, int* lm_entries, void* stream) {
  VUS_REQUIRE(P != nullptr && lm_entries != nullptr, "null argument");
  VUS_REQUIRE(P->n_points >= 1 && P->point_ptr && P->obs_pose, "the problem holds no landmark");
  tiles_count_kernel<<<cdiv(P->n_points, 256), 256, 0, vus::as_stream(stream)>>>(P->point_ptr, P->obs_pose, P->n_points, lm_entries);
  VUS_CHECK_LAUNCH("ba_tiles_count");
  return VUS_OK;
}

extern "C" long long vus_ba_tiles_work_bytes(int n_entries) {
  const size_t n = (size_t)(n_entries > 0 ? n_entries : 1);
  // keys in and out, the identity and the permutation, the unsorted entries, the sort's own scratch
  return (long long)(2 * align256(4 * n) + 2 * align256(4 * n) + align256(16 * n) + sort_temp_bytes((int)n) + 1024);
}

extern "C" int vus_ba_tiles_fill(const vus_ba_problem* P, int band, const int* lm_base, int n_entries, int* unit_ptr,
                                 int* entries, int* order, void* work, long long work_bytes, void* stream) {
  VUS_REQUIRE(P != nullptr && lm_base && unit_ptr && order && work, "null argument");
  VUS_REQUIRE(band >= 0 && n_entries >= 0 && (entries || n_entries == 0), "band=%d n_entries=%d", band, n_entries);
  VUS_REQUIRE(work_bytes >= vus_ba_tiles_work_bytes(n_entries), "workspace of %lld bytes, %lld needed", work_bytes,
              vus_ba_tiles_work_bytes(n_entries));
  hipStream_t st = vus::as_stream(stream);
  const int n_tiles = (P->n_poses + TILE_POSES - 1) / TILE_POSES, dt1 = (band + TILE_POSES - 1) / TILE_POSES + 1;
  VUS_REQUIRE((long long)n_tiles * dt1 < (1ll << 31), "%d pose tiles x %d tile distances exceed the unit index", n_tiles, dt1);
  const int n_units = n_tiles * dt1, n = n_entries;
  if (n > 0) {
    Carve c{static_cast<char*>(work), (size_t)work_bytes};
    unsigned* key = c.take<unsigned>(n);
    unsigned* sk = c.take<unsigned>(n);
    int* iota = c.take<int>(n);
    int* perm = c.take<int>(n);
    int4* ent = c.take<int4>(n);
    SortTemp tmp;
    VUS_REQUIRE(key && sk && iota && perm && ent && take_sort_temp(c, n, tmp), "workspace too small");
    iota_kernel<<<cdiv(n, 256), 256, 0, st>>>(iota, n);
    tiles_emit_kernel<<<cdiv(P->n_points, 256), 256, 0, st>>>(P->point_ptr, P->obs_pose, P->n_points, dt1, lm_base, key, ent);
    if (int rc = radix_sort_pairs<unsigned>(key, sk, iota, perm, n, bits_for((unsigned long long)n_units), tmp, st)) return rc;
    tiles_gather_kernel<<<cdiv(n, 256), 256, 0, st>>>(ent, perm, n, reinterpret_cast<int4*>(entries));
    tiles_ptr_kernel<<<cdiv(n_units + 1, 256), 256, 0, st>>>(sk, n, n_units, unit_ptr);
  } else {
    VUS_CHECK_HIP(hipMemsetAsync(unit_ptr, 0, sizeof(int) * (size_t)(n_units + 1), st));
  }
  tiles_order_kernel<<<1, 1024, 0, st>>>(unit_ptr, n_units, order);
  VUS_CHECK_LAUNCH("ba_tiles_fill");
  return VUS_OK;
}
