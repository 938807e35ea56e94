// Integer bookkeeping of the TILED matching path on the device (round 6) — what pipeline.py / tile_matching.py did with
// torch.argsort / torch.unique / bincount / cumsum / gather (rocPRIM sorts behind a host-language API) around MatcherBase._match_by_tile's
// np.unique (MB:452-459) and get_features_by_tile's boolean masks (MB:1380-1391):
//
//   dim_op_tile_counts        keypoints per tile of one image's merged table (the table capacities are sized from them on the host)
//   dim_op_group_by_tile      the merged table -> per-tile tables (keypoints / descriptors / index map), original order inside a tile
//                             (= the boolean-mask order): a stable grouping = a sort of the keys  tile << 32 | index
//   dim_op_tile_match_keys    a batch of tile-pair match lists -> 64-bit keys  slot << 40 | idx0 << 20 | idx1  in the image's index space
//   dim_op_unique_match_rows  all keys of a phase -> per image pair the UNIQUE rows in lexicographic order (np.unique(axis=0)) + counts
//
// One primitive underneath: an ascending sort of 64-bit keys — 4096-key chunks bitonic-sorted in LDS by one workgroup each, then log2(chunks)
// merge passes in which every key finds its place in the merged run by a binary search in the partner run (rank = own position + partner
// keys below it; equal keys: the left run's go first, so the pass is a permutation).  Same scheme as sp_post.hip's top-k above 4096.
#include "dim_kernels.h"

namespace {
constexpr int SC = 4096;                       // keys per chunk
constexpr unsigned long long SENT = ~0ull;     // padding / dead rows: sorts behind everything

__device__ __forceinline__ void so_bitonic_asc(unsigned long long* keys) {
  const int t = threadIdx.x;
  for (int size = 2; size <= SC; size <<= 1) {
    for (int stride = size >> 1; stride > 0; stride >>= 1) {
      for (int i = t; i < (SC >> 1); i += 1024) {
        const int lo = 2 * i - (i & (stride - 1)), hi = lo + stride;
        const bool asc = ((lo & size) == 0);
        const unsigned long long a = keys[lo], c = keys[hi];
        if (asc ? (a > c) : (a < c)) { keys[lo] = c; keys[hi] = a; }
      }
      __syncthreads();
    }
  }
}
// n_live (device, may be null = all n_pad): chunks / keys past its round-up to whole chunks are not touched
__device__ __forceinline__ int so_live_pad(const int* n_live, int n_pad) {
  if (n_live == nullptr) return n_pad;
  const int v = (*n_live + SC - 1) / SC * SC;
  return v < n_pad ? v : n_pad;
}
__global__ __launch_bounds__(1024) void so_chunk_sort_kernel(unsigned long long* __restrict__ keys, int n_pad, const int* __restrict__ n_live) {
  __shared__ unsigned long long sh[SC];
  const int t = threadIdx.x;
  const size_t c0 = (size_t)blockIdx.x * SC;
  if ((long long)c0 >= so_live_pad(n_live, n_pad)) return;   // (uniform)
  for (int i = t; i < SC; i += 1024) sh[i] = keys[c0 + i];
  __syncthreads();
  so_bitonic_asc(sh);
  for (int i = t; i < SC; i += 1024) keys[c0 + i] = sh[i];
}
// runs of L keys (multiples of SC) merged pairwise: src -> dst
__global__ __launch_bounds__(256) void so_merge_pass_kernel(const unsigned long long* __restrict__ src, unsigned long long* __restrict__ dst, int n_pad,
                                                            int L, const int* __restrict__ n_live) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  const int lim = so_live_pad(n_live, n_pad);
  if (i >= lim) return;
  const int run = i / L, partner = run ^ 1;
  const unsigned long long key = src[i];
  const int p0 = partner * L, p1 = min(lim, p0 + L);
  int cnt = 0;
  if (p0 < lim) {
    const unsigned long long* pr = src + p0;
    int lo = 0, hi = p1 - p0;
    if (run & 1) { while (lo < hi) { const int mid = (lo + hi) >> 1; if (pr[mid] <= key) lo = mid + 1; else hi = mid; } }   // right run: behind its equals
    else { while (lo < hi) { const int mid = (lo + hi) >> 1; if (pr[mid] < key) lo = mid + 1; else hi = mid; } }
    cnt = lo;
  }
  dst[(run & ~1) * L + (i - run * L) + cnt] = key;
}
// sorts a[0 .. n_pad) ascending (n_pad a multiple of SC); returns the buffer that holds the result (a or b)
unsigned long long* so_sort(unsigned long long* a, unsigned long long* b, int n_pad, const int* n_live, hipStream_t s) {
  hipLaunchKernelGGL(so_chunk_sort_kernel, dim3(n_pad / SC), dim3(1024), 0, s, a, n_pad, n_live);
  for (int L = SC; L < n_pad; L <<= 1) {
    hipLaunchKernelGGL(so_merge_pass_kernel, dim3(cdiv(n_pad, 256)), dim3(256), 0, s, (const unsigned long long*)a, b, n_pad, L, n_live);
    unsigned long long* t = a; a = b; b = t;
  }
  return a;
}
__device__ __forceinline__ int so_lower_bound(const unsigned long long* __restrict__ keys, int n, unsigned long long v) {   // first position with keys[pos] >= v
  int lo = 0, hi = n;
  while (lo < hi) { const int mid = (lo + hi) >> 1; if (keys[mid] < v) lo = mid + 1; else hi = mid; }
  return lo;
}

// ---------------------------------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void so_tile_count_kernel(const float* __restrict__ tile_idx, int ld, int n, int n_tiles, int* __restrict__ counts) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const int t = (int)tile_idx[(size_t)i * ld];
  if (t >= 0 && t < n_tiles) atomicAdd(&counts[t], 1);
}
__global__ __launch_bounds__(256) void so_tile_keys_kernel(const float* __restrict__ tile_idx, int ld, int n, int n_pad, unsigned long long* __restrict__ keys) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n_pad) return;
  keys[i] = i < n ? (((unsigned long long)(unsigned)(int)tile_idx[(size_t)i * ld]) << 32) | (unsigned)i : SENT;
}
// one thread per tile: its run in the sorted keys -> start, count (and the table's live count)
__global__ __launch_bounds__(256) void so_tile_runs_kernel(const unsigned long long* __restrict__ keys, int n, int n_tiles, const int* __restrict__ row_of_tile,
                                                           int cap, int* __restrict__ start, int* __restrict__ nt) {
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (t >= n_tiles) return;
  const int a = so_lower_bound(keys, n, (unsigned long long)t << 32), b = so_lower_bound(keys, n, (unsigned long long)(t + 1) << 32);
  start[t] = a;
  const int r = row_of_tile[t];
  if (r >= 0) nt[r] = min(b - a, cap);
}
// one wave per sorted position: keypoint, descriptor row and index into the tile's table
__global__ __launch_bounds__(256) void so_tile_scatter_kernel(const unsigned long long* __restrict__ keys, int n, int n_tiles, const int* __restrict__ row_of_tile,
                                                              const int* __restrict__ start, const float* __restrict__ kp, int ld_kp, const float* __restrict__ desc, int ld_desc,
                                                              int D, int cap,
                                                              float* __restrict__ kt, float* __restrict__ dt, long long* __restrict__ it) {
  const int sidx = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (sidx >= n) return;
  const unsigned long long key = keys[sidx];
  const int t = (int)(key >> 32), i = (int)(key & 0xffffffffull);
  if (t < 0 || t >= n_tiles) return;
  const int r = row_of_tile[t], pos = sidx - start[t];
  if (r < 0 || pos >= cap) return;
  const size_t slot = (size_t)r * cap + pos;
  if (lane < 2) kt[slot * 2 + lane] = kp[(size_t)i * ld_kp + lane];
  if (lane == 2) it[slot] = i;
  for (int d = lane; d < D; d += 64) dt[slot * D + d] = desc[(size_t)i * ld_desc + d];
}

// ---------------------------------------------------------------------------------------------------------------------------------------------
// thread = (tile pair j of the batch, match row k)
__global__ __launch_bounds__(256) void so_match_keys_kernel(const long long* __restrict__ matches, const int* __restrict__ n_matches, const long long* __restrict__ it,
                                                            const int* __restrict__ pidx, const int* __restrict__ slot, int b, int NK, int cap,
                                                            unsigned long long* __restrict__ keys) {
  const int g = blockIdx.x * 256 + threadIdx.x;
  if (g >= b * NK) return;
  const int j = g / NK, k = g - j * NK;
  unsigned long long key = SENT;
  if (k < n_matches[j]) {
    const long long m0 = matches[(size_t)g * 2], m1 = matches[(size_t)g * 2 + 1];
    const long long g0 = it[(size_t)pidx[2 * j] * cap + min(max(m0, 0ll), (long long)cap - 1)];
    const long long g1 = it[(size_t)pidx[2 * j + 1] * cap + min(max(m1, 0ll), (long long)cap - 1)];
    key = ((unsigned long long)slot[j] << 40) | ((unsigned long long)g0 << 20) | (unsigned long long)g1;
  }
  keys[g] = key;
}
// live keys to the front of `out` (order free: a sort follows), count in *n_live
__global__ __launch_bounds__(256) void so_compact_kernel(const unsigned long long* __restrict__ keys, int n, unsigned long long* __restrict__ out, int* __restrict__ n_live) {
  const int i = blockIdx.x * 256 + threadIdx.x, lane = threadIdx.x & 63;
  const unsigned long long key = i < n ? keys[i] : SENT;
  const bool live = key != SENT;
  const unsigned long long m = __ballot(live);
  int base = 0;
  if (lane == 0 && m) base = atomicAdd(n_live, __popcll(m));
  base = __shfl(base, 0);
  if (live) out[base + __popcll(m & ((1ull << lane) - 1ull))] = key;
}
__global__ __launch_bounds__(256) void so_pad_kernel(unsigned long long* __restrict__ keys, int n_pad, const int* __restrict__ n_live) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= *n_live && i < so_live_pad(n_live, n_pad)) keys[i] = SENT;
}
// flag = first of its run of equal keys; per 1024-key block: the number of flags
__global__ __launch_bounds__(1024) void so_flag_count_kernel(const unsigned long long* __restrict__ keys, const int* __restrict__ n_live, int* __restrict__ blk) {
  __shared__ int wsum[16];
  const int t = threadIdx.x, i = blockIdx.x * 1024 + t, n = *n_live;
  const int f = (i < n && (i == 0 || keys[i] != keys[i - 1])) ? 1 : 0;
  const unsigned long long m = __ballot(f);
  if ((t & 63) == 0) wsum[t >> 6] = __popcll(m);
  __syncthreads();
  if (t == 0) {
    int s = 0;
    for (int w = 0; w < 16; ++w) s += wsum[w];
    blk[blockIdx.x] = s;
  }
}
// ONE workgroup: exclusive scan of the block counts in place; total -> *n_unique
__global__ __launch_bounds__(1024) void so_block_scan_kernel(int* __restrict__ blk, int n_blk, int* __restrict__ n_unique) {
  __shared__ int part[1024];
  const int t = threadIdx.x;
  const int per = (n_blk + 1023) / 1024, j0 = t * per, j1 = min(j0 + per, n_blk);
  int sum = 0;
  for (int j = j0; j < j1; ++j) sum += blk[j];
  part[t] = sum;
  __syncthreads();
  for (int off = 1; off < 1024; off <<= 1) {
    const int v = t >= off ? part[t - off] : 0;
    __syncthreads();
    part[t] += v;
    __syncthreads();
  }
  int run = part[t] - sum;
  for (int j = j0; j < j1; ++j) { const int c = blk[j]; blk[j] = run; run += c; }
  if (t == 1023) { *n_unique = part[1023]; blk[n_blk] = part[1023]; }   // (blk has n_blk + 1 entries: the tail serves position n_live == n_pad)
}
// prefix[i] = number of flags in front of position i (i <= n_live: prefix[n_live] = the total)
__global__ __launch_bounds__(1024) void so_prefix_kernel(const unsigned long long* __restrict__ keys, const int* __restrict__ n_live, const int* __restrict__ blk,
                                                         int* __restrict__ prefix) {
  __shared__ int wsum[16];
  const int t = threadIdx.x, lane = t & 63, wv = t >> 6, i = blockIdx.x * 1024 + t, n = *n_live;
  const int f = (i < n && (i == 0 || keys[i] != keys[i - 1])) ? 1 : 0;   // (block n_blk only exists for position n_live == n_pad: no flags)
  const unsigned long long m = __ballot(f);
  if (lane == 0) wsum[wv] = __popcll(m);
  __syncthreads();
  int before = blk[blockIdx.x];
  for (int w = 0; w < wv; ++w) before += wsum[w];
  before += __popcll(m & ((1ull << lane) - 1ull));
  if (i <= n) prefix[i] = before;
}
// thread = sorted position: a flagged key goes to row (its rank among the slot's unique keys) of its image pair
template <typename RowT>
__global__ __launch_bounds__(256) void so_rows_kernel(const unsigned long long* __restrict__ keys, const int* __restrict__ n_live, const int* __restrict__ prefix,
                                                      int n_slots, int cap_m, RowT* __restrict__ rows) {
  const int i = blockIdx.x * 256 + threadIdx.x, n = *n_live;
  if (i >= n) return;
  const unsigned long long key = keys[i];
  if (i > 0 && keys[i - 1] == key) return;
  const int slot = (int)(key >> 40);
  if (slot >= n_slots) return;
  const int pos = prefix[i] - prefix[so_lower_bound(keys, n, (unsigned long long)slot << 40)];
  if (pos < cap_m) {
    rows[((size_t)slot * cap_m + pos) * 2] = (RowT)((key >> 20) & 0xFFFFFull);
    rows[((size_t)slot * cap_m + pos) * 2 + 1] = (RowT)(key & 0xFFFFFull);
  }
}
__global__ __launch_bounds__(256) void so_slot_counts_kernel(const unsigned long long* __restrict__ keys, const int* __restrict__ n_live, const int* __restrict__ prefix,
                                                             int n_slots, int cap_m, int* __restrict__ cnt, int* __restrict__ n_full) {
  const int slot = blockIdx.x * 256 + threadIdx.x, n = *n_live;
  if (slot >= n_slots) return;
  const int full = prefix[so_lower_bound(keys, n, (unsigned long long)(slot + 1) << 40)] - prefix[so_lower_bound(keys, n, (unsigned long long)slot << 40)];
  cnt[slot] = min(full, cap_m);
  if (n_full) n_full[slot] = full;
}
inline size_t so_pad(size_t n) { return (n + SC - 1) / SC * SC; }
}  // namespace

extern "C" {

int dim_op_tile_counts(const float* tile_idx_dev, int ld_tile, int n, int n_tiles, int32_t* counts_dev, void* stream) {
  DIM_REQUIRE(counts_dev && n >= 0 && n_tiles > 0 && ld_tile >= 1 && (n == 0 || tile_idx_dev), "dim_op_tile_counts: bad argument");
  hipStream_t s = (hipStream_t)stream;
  DIM_HIP(hipMemsetAsync(counts_dev, 0, (size_t)n_tiles * sizeof(int), s));
  if (n > 0) hipLaunchKernelGGL(so_tile_count_kernel, dim3(cdiv(n, 256)), dim3(256), 0, s, tile_idx_dev, ld_tile, n, n_tiles, counts_dev);
  DIM_LAUNCH_CHECK();
  return 0;
}

size_t dim_op_group_by_tile_workspace_bytes(int n, int n_tiles) {
  return 2 * so_pad((size_t)(n > 0 ? n : 0)) * 8 + (size_t)(n_tiles > 0 ? n_tiles : 0) * 4 + 256;
}

int dim_op_group_by_tile(const float* tile_idx_dev, int ld_tile, const float* kpts_dev, int ld_kpts, const float* desc_nd_dev, int ld_desc, int n, int D,
                         const int32_t* row_of_tile_dev, int n_tiles, int cap, float* kt_dev, float* dt_dev, long long* it_dev, int32_t* nt_dev, void* workspace, void* stream) {
  DIM_REQUIRE(row_of_tile_dev && kt_dev && dt_dev && it_dev && nt_dev && workspace && n >= 0 && n_tiles > 0 && cap > 0 && D > 0 && ld_tile >= 1 && ld_kpts >= 2 && ld_desc >= D,
              "dim_op_group_by_tile: bad argument");
  if (n == 0) return 0;
  DIM_REQUIRE(tile_idx_dev && kpts_dev && desc_nd_dev, "dim_op_group_by_tile: null table");
  hipStream_t s = (hipStream_t)stream;
  const int n_pad = (int)so_pad((size_t)n);
  unsigned long long* a = (unsigned long long*)workspace;
  unsigned long long* b = a + n_pad;
  int* start = (int*)(b + n_pad);
  hipLaunchKernelGGL(so_tile_keys_kernel, dim3(cdiv(n_pad, 256)), dim3(256), 0, s, tile_idx_dev, ld_tile, n, n_pad, a);
  const unsigned long long* sorted = so_sort(a, b, n_pad, nullptr, s);
  hipLaunchKernelGGL(so_tile_runs_kernel, dim3(cdiv(n_tiles, 256)), dim3(256), 0, s, sorted, n, n_tiles, row_of_tile_dev, cap, start, nt_dev);
  hipLaunchKernelGGL(so_tile_scatter_kernel, dim3(cdiv(n, 4)), dim3(256), 0, s, sorted, n, n_tiles, row_of_tile_dev, (const int*)start, kpts_dev, ld_kpts, desc_nd_dev, ld_desc, D, cap,
                     kt_dev, dt_dev, it_dev);
  DIM_LAUNCH_CHECK();
  return 0;
}

int dim_op_tile_match_keys(const long long* matches_dev, const int32_t* n_matches_dev, const long long* it_dev, const int32_t* pair_rows_dev, const int32_t* slot_dev,
                           int n_pairs, int nk, int cap, unsigned long long* keys_dev, void* stream) {
  DIM_REQUIRE(matches_dev && n_matches_dev && it_dev && pair_rows_dev && slot_dev && keys_dev && n_pairs >= 0 && nk > 0 && cap > 0, "dim_op_tile_match_keys: bad argument");
  if (n_pairs == 0) return 0;
  hipLaunchKernelGGL(so_match_keys_kernel, dim3(cdiv(n_pairs * nk, 256)), dim3(256), 0, (hipStream_t)stream, matches_dev, n_matches_dev, it_dev, pair_rows_dev, slot_dev,
                     n_pairs, nk, cap, keys_dev);
  DIM_LAUNCH_CHECK();
  return 0;
}

size_t dim_op_unique_match_rows_workspace_bytes(long long n) {
  const size_t np = so_pad((size_t)(n > 0 ? n : 0));
  return 2 * np * 8 + (np + 1) * 4 + (np / 1024 + 2) * 4 + 2 * 4 + 256;
}

int dim_op_unique_match_rows(const unsigned long long* keys_dev, long long n, int n_slots, int cap_m, int rows_are_i64, void* rows_dev, int32_t* cnt_dev,
                             int32_t* n_full_dev, void* workspace, void* stream) {
  DIM_REQUIRE(rows_dev && cnt_dev && workspace && n >= 0 && n < (1ll << 30) && n_slots > 0 && n_slots < (1 << 23) && cap_m > 0, "dim_op_unique_match_rows: bad argument");
  hipStream_t s = (hipStream_t)stream;
  if (n == 0) {
    DIM_HIP(hipMemsetAsync(cnt_dev, 0, (size_t)n_slots * sizeof(int), s));
    if (n_full_dev) DIM_HIP(hipMemsetAsync(n_full_dev, 0, (size_t)n_slots * sizeof(int), s));
    return 0;
  }
  DIM_REQUIRE(keys_dev, "dim_op_unique_match_rows: null keys");
  const int n_pad = (int)so_pad((size_t)n), n_blk = n_pad / 1024;
  unsigned long long* a = (unsigned long long*)workspace;
  unsigned long long* b = a + n_pad;
  int* prefix = (int*)(b + n_pad);          // n_pad + 1
  int* blk = prefix + n_pad + 1;            // n_blk + 1
  int* scal = blk + n_blk + 1;              // [0] n_live, [1] n_unique
  DIM_HIP(hipMemsetAsync(scal, 0, 2 * sizeof(int), s));
  hipLaunchKernelGGL(so_compact_kernel, dim3(cdiv((int)n, 256)), dim3(256), 0, s, keys_dev, (int)n, a, scal);
  hipLaunchKernelGGL(so_pad_kernel, dim3(cdiv(n_pad, 256)), dim3(256), 0, s, a, n_pad, (const int*)scal);
  const unsigned long long* sorted = so_sort(a, b, n_pad, scal, s);
  hipLaunchKernelGGL(so_flag_count_kernel, dim3(n_blk), dim3(1024), 0, s, sorted, (const int*)scal, blk);
  hipLaunchKernelGGL(so_block_scan_kernel, dim3(1), dim3(1024), 0, s, blk, n_blk, scal + 1);
  hipLaunchKernelGGL(so_prefix_kernel, dim3(n_blk + 1), dim3(1024), 0, s, sorted, (const int*)scal, (const int*)blk, prefix);
  if (rows_are_i64) hipLaunchKernelGGL(HIP_KERNEL_NAME(so_rows_kernel<long long>), dim3(cdiv(n_pad, 256)), dim3(256), 0, s, sorted, (const int*)scal, (const int*)prefix, n_slots, cap_m, (long long*)rows_dev);
  else hipLaunchKernelGGL(HIP_KERNEL_NAME(so_rows_kernel<int>), dim3(cdiv(n_pad, 256)), dim3(256), 0, s, sorted, (const int*)scal, (const int*)prefix, n_slots, cap_m, (int*)rows_dev);
  hipLaunchKernelGGL(so_slot_counts_kernel, dim3(cdiv(n_slots, 256)), dim3(256), 0, s, sorted, (const int*)scal, (const int*)prefix, n_slots, cap_m, cnt_dev, n_full_dev);
  DIM_LAUNCH_CHECK();
  return 0;
}

}  // extern "C"
