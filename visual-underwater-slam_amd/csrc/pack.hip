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
// The sorts are rocPRIM's device radix sort (a ROCm library primitive compiled into this library: index plumbing, not
// arithmetic of the path); everything around them is small HIP kernels.  All results stay on the device.
#include <cstring>
#include <rocprim/rocprim.hpp>
#include "vus_common.h"

namespace {

inline int cdiv(long long a, int b) { return (int)((a + b - 1) / b); }
inline size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

__global__ void iota_kernel(int* __restrict__ a, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) a[i] = i;
}

// head[i] = 1 where a new key starts in the sorted sequence
__global__ void head_flags_kernel(const unsigned long long* __restrict__ sk, int n, int* __restrict__ head) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) head[i] = (i == 0 || sk[i] != sk[i - 1]) ? 1 : 0;
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

size_t sort_temp_bytes(int n) {
  size_t b64 = 0, b32 = 0, bs = 0;
  (void)rocprim::radix_sort_pairs(nullptr, b64, (const unsigned long long*)nullptr, (unsigned long long*)nullptr,
                                  (const int*)nullptr, (int*)nullptr, (size_t)n, 0, 64);
  (void)rocprim::radix_sort_pairs(nullptr, b32, (const unsigned int*)nullptr, (unsigned int*)nullptr, (const int*)nullptr,
                                  (int*)nullptr, (size_t)n, 0, 32);
  (void)rocprim::inclusive_scan(nullptr, bs, (const int*)nullptr, (int*)nullptr, (size_t)n, rocprim::plus<int>());
  size_t m = b64 > b32 ? b64 : b32;
  return m > bs ? m : bs;
}

}  // namespace

extern "C" long long vus_pack_work_bytes(int n) {
  if (n < 0) return 0;
  const size_t nn = (size_t)(n > 0 ? n : 1);
  // two 8-byte key arrays, four 4-byte index arrays, the library's own temporary storage
  return (long long)(2 * align256(8 * nn) + 4 * align256(4 * nn) + align256(sort_temp_bytes(n > 0 ? n : 1)) + 1024);
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
  int* head = c.take<int>(n);
  int* rank = c.take<int>(n);
  size_t tb = sort_temp_bytes(n);
  void* tmp = c.take<char>(tb);
  VUS_REQUIRE(sk && iota && sp && head && rank && tmp, "workspace too small");
  iota_kernel<<<cdiv(n, 256), 256, 0, st>>>(iota, n);
  // gtsam keys are non-negative (chr << 56 | index): their unsigned order is their order
  VUS_CHECK_HIP(rocprim::radix_sort_pairs(tmp, tb, reinterpret_cast<const unsigned long long*>(keys), sk, iota, sp, (size_t)n,
                                          0, 64, st));
  head_flags_kernel<<<cdiv(n, 256), 256, 0, st>>>(sk, n, head);
  size_t sb = tb;
  VUS_CHECK_HIP(rocprim::inclusive_scan(tmp, sb, head, rank, (size_t)n, rocprim::plus<int>(), st));
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
  size_t tb = sort_temp_bytes(n);
  void* tmp = c.take<char>(tb);
  VUS_REQUIRE(key && sk && iota && spose && tmp, "workspace too small");
  iota_kernel<<<cdiv(n, 256), 256, 0, st>>>(iota, n);
  make_keys_kernel<<<cdiv(n, 256), 256, 0, st>>>(obs_pose, obs_point, n, n_poses, n_points, key, flags);
  // L-order: by (point, pose).  Keys are unique in a valid graph, so the order is total.
  const int kbits = bits_for((unsigned long long)(n_points > 0 ? n_points : 1) * (unsigned long long)n_poses);
  VUS_CHECK_HIP(rocprim::radix_sort_pairs(tmp, tb, key, sk, iota, perm, (size_t)n, 0, kbits, st));
  gather_L_kernel<<<cdiv(n, 256), 256, 0, st>>>(sk, perm, meas, n, n_poses, meas_L, obs_pose_L, obs_point_L, flags);
  point_ptr_kernel<<<cdiv(n_points + 1, 256), 256, 0, st>>>(sk, n, n_poses, n_points, point_ptr);
  if (band && n_points > 0) band_kernel<<<cdiv(n_points, 256), 256, 0, st>>>(point_ptr, obs_pose_L, n_points, band);
  // P-order: a STABLE sort of the L-order rows by pose keeps the points ascending inside every pose
  size_t tb2 = tb;
  VUS_CHECK_HIP(rocprim::radix_sort_pairs(tmp, tb2, reinterpret_cast<const unsigned int*>(obs_pose_L),
                                          reinterpret_cast<unsigned int*>(spose), iota, pobs_lidx, (size_t)n, 0,
                                          bits_for((unsigned long long)n_poses), st));
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
