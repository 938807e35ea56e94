// Merging per-tile feature tables into one image-level feature set ON THE DEVICE — the tail of ExtractorBase._extract_by_tile
// (EB:330-390): shift every tile's keypoints by the tile origin, drop those within 2 px of the (unpadded) image border,
// concatenate in tile order, de-duplicate with np.unique(axis=0, return_index=True) — i.e. sort the rows lexicographically by
// (x, y), keep the FIRST occurrence (lowest concatenation index) of every distinct row — and hand back keypoints (N, 2), scores (N),
// tile_idx (N) and descriptors TRANSPOSED to (D, N).  On the host this was 42 ms for the 64 000 keypoints of a 16-tile
// 6000 x 4000 image (np.unique on rows, a fancy-index gather of a 33 MB array, 16 strided transposes) against 12 ms of network.
//
// Keys: surviving keypoints have x, y >= 2 > 0, and for positive floats the IEEE bit pattern orders like the value, so
// key = bits(x) << 32 | bits(y) compared as an unsigned 64-bit integer IS numpy's lexicographic row order; rows that the border
// test drops (and slots past a tile's live count) get the key ~0 and sort behind everything.
// Order: a rank sort (rank = number of rows that precede this one; ties by concatenation index, which is what makes "first
// occurrence" well defined) — O(n^2) compares of scalar-broadcast keys, no temporary storage beyond the rank array, and exactly
// reproducible.  Then one single-workgroup scan compacts the survivors, and a gather writes the four outputs; the (D, N)
// descriptor rows use the live count N as their stride, read from device memory, so the first D * N floats of the output
// buffer are the contiguous array the reference returns and the host never has to know N before the final copy.
#include "dim_kernels.h"

namespace {
constexpr unsigned long long DROPPED = ~0ull;

// thread = (tile, slot): shifted keypoint + sort key
__global__ __launch_bounds__(256) void tm_keys_kernel(const float* __restrict__ kp, const int* __restrict__ n_tab, const int* __restrict__ origins_xy,
                                                      int n_tiles, int cap, int H, int W, float thr, float* __restrict__ kp_shift,
                                                      unsigned long long* __restrict__ keys) {
  const int g = blockIdx.x * 256 + threadIdx.x;
  if (g >= n_tiles * cap) return;
  const int tile = g / cap, slot = g - tile * cap;
  unsigned long long key = DROPPED;
  float x = 0.0f, y = 0.0f;
  if (slot < n_tab[tile]) {
    // fp32 + fp32 of an integer-valued origin, like keypoints + np.array(origin, dtype=float32)
    x = kp[2 * (size_t)g] + (float)origins_xy[2 * tile];
    y = kp[2 * (size_t)g + 1] + (float)origins_xy[2 * tile + 1];
    if (x >= thr && x < (float)W - thr && y >= thr && y < (float)H - thr)
      key = ((unsigned long long)__float_as_uint(x) << 32) | (unsigned long long)__float_as_uint(y);
  }
  kp_shift[2 * (size_t)g] = x; kp_shift[2 * (size_t)g + 1] = y;
  keys[g] = key;
}

// thread = row i, blockIdx.y = one of RANK_PARTS slices of j: rank_i = #{j : key_j < key_i, or key_j == key_i and j < i}, summed over
// the slices with one atomic per thread.  j is wave-uniform (the compared keys come through the scalar cache), and outside the
// workgroup's own 256 rows the tie rule is known per slice — rows before the workgroup count when <=, rows after it when < —
// so the long loops are one 64-bit compare and one add per pair.  Dropped rows are ranked too (behind all survivors), which
// keeps `order` a permutation.
constexpr int RANK_PARTS = 8;
__global__ __launch_bounds__(256) void tm_rank_kernel(const unsigned long long* __restrict__ keys, int n, int* __restrict__ rank) {
  const int i0 = blockIdx.x * 256, i = i0 + threadIdx.x;
  const int per = (n + RANK_PARTS - 1) / RANK_PARTS, ja = blockIdx.y * per, jb = min(n, ja + per);
  const unsigned long long ki = i < n ? keys[i] : DROPPED;
  int cnt = 0;
  for (int j = ja; j < min(jb, i0); ++j) cnt += keys[j] <= ki;
  for (int j = max(ja, i0); j < min(jb, i0 + 256); ++j) { const unsigned long long kj = keys[j]; cnt += (kj < ki) | ((kj == ki) & (j < i)); }
  for (int j = max(ja, i0 + 256); j < jb; ++j) cnt += keys[j] < ki;
  if (i < n && cnt) atomicAdd(&rank[i], cnt);
}
__global__ __launch_bounds__(256) void tm_scatter_kernel(const unsigned long long* __restrict__ keys, const int* __restrict__ rank, int n, int* __restrict__ order,
                                                         unsigned long long* __restrict__ sorted_keys) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) { order[rank[i]] = i; sorted_keys[rank[i]] = keys[i]; }
}

// ONE workgroup: keep[r] = survivor and (not unique, or first of its run of equal keys); exclusive scan -> output slot of rank r.
__global__ __launch_bounds__(1024) void tm_scan_kernel(const unsigned long long* __restrict__ sorted_keys, int n, int unique, int* __restrict__ src_of_out,
                                                       const int* __restrict__ order, int* __restrict__ n_out) {
  __shared__ int wsum[16];
  __shared__ int carry;
  const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
  if (t == 0) carry = 0;
  __syncthreads();
  for (int base = 0; base < n; base += 1024) {
    const int r = base + t;
    int keep = 0;
    if (r < n) {
      const unsigned long long k = sorted_keys[r];
      keep = k != DROPPED && (!unique || r == 0 || sorted_keys[r - 1] != k);
    }
    int incl = keep;   // inclusive scan over the wave, then over the 16 waves
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const int v = __shfl_up(incl, d); if (lane >= d) incl += v; }
    if (lane == 63) wsum[wv] = incl;
    __syncthreads();
    int before = carry;
    for (int w = 0; w < wv; ++w) before += wsum[w];
    if (keep) src_of_out[before + incl - 1] = order ? order[r] : r;
    __syncthreads();
    if (t == 1023) carry = before + incl;
    __syncthreads();
  }
  if (t == 0) *n_out = carry;
}

// thread = output row o (keypoint, score, tile index): the descriptor transpose is the next kernel
__global__ __launch_bounds__(256) void tm_emit_kernel(const int* __restrict__ src_of_out, const int* __restrict__ n_out, const float* __restrict__ kp_shift,
                                                      const float* __restrict__ scores, const float* __restrict__ tile_ids, int cap,
                                                      float* __restrict__ out_kp, float* __restrict__ out_scores, float* __restrict__ out_tidx) {
  const int o = blockIdx.x * 256 + threadIdx.x;
  if (o >= *n_out) return;
  const int g = src_of_out[o];
  out_kp[2 * (size_t)o] = kp_shift[2 * (size_t)g]; out_kp[2 * (size_t)o + 1] = kp_shift[2 * (size_t)g + 1];
  out_scores[o] = scores[g];
  out_tidx[o] = tile_ids[g / cap];
}
// workgroup = 64 output rows x 128 dims.  The 64 source rows are read as rows (each wave: one row per step, lanes along d:
// coalesced) into LDS and written back transposed (lanes along o: coalesced), stride N = the live count.
__global__ __launch_bounds__(256) void tm_desc_kernel(const int* __restrict__ src_of_out, const int* __restrict__ n_out, const float* __restrict__ desc, int D,
                                                      float* __restrict__ out_desc) {
  __shared__ float tile[64 * 129];
  const int N = *n_out, o0 = blockIdx.x * 64, d0 = blockIdx.y * 128, dn = min(128, D - d0);
  if (o0 >= N) return;
  const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
  for (int r = wv; r < 64; r += 4) {
    const int o = o0 + r;
    if (o < N) {
      const float* s = desc + (size_t)src_of_out[o] * D + d0;
      for (int d = lane; d < dn; d += 64) tile[r * 129 + d] = s[d];
    }
  }
  __syncthreads();
  if (o0 + lane < N)
    for (int d = wv; d < dn; d += 4) out_desc[(size_t)(d0 + d) * N + o0 + lane] = tile[lane * 129 + d];
}
}  // namespace

extern "C" {

size_t dim_op_merge_tiles_workspace_bytes(int n_tiles, int cap) {
  const size_t n = (size_t)(n_tiles > 0 ? n_tiles : 0) * (size_t)(cap > 0 ? cap : 0);
  return n * (8 + 8 + 8 + 4 + 4 + 4) + 256;   // keys, sorted keys, shifted keypoints, order, src_of_out, rank
}

int dim_op_merge_tiles(const float* kpts_tab, const float* scores_tab, const float* desc_tab, const int32_t* n_tab, const int32_t* origins_xy,
                       const float* tile_ids, int n_tiles, int cap, int D, int image_h, int image_w, int select_unique, void* workspace,
                       float* out_kpts, float* out_scores, float* out_tile_idx, float* out_desc, int32_t* n_out, void* stream) {
  DIM_REQUIRE(kpts_tab && scores_tab && desc_tab && n_tab && origins_xy && tile_ids && workspace && out_kpts && out_scores && out_tile_idx && out_desc && n_out,
              "dim_op_merge_tiles: null argument");
  DIM_REQUIRE(n_tiles > 0 && cap > 0 && D > 0 && image_h > 0 && image_w > 0 && (long long)n_tiles * cap < (1ll << 30), "dim_op_merge_tiles: bad sizes");
  hipStream_t s = (hipStream_t)stream;
  const int n = n_tiles * cap;
  unsigned long long* keys = (unsigned long long*)workspace;
  unsigned long long* skeys = keys + n;
  float* kp_shift = (float*)(skeys + n);
  int* order = (int*)(kp_shift + 2 * (size_t)n);
  int* src = order + n;
  int* rank = src + n;
  hipLaunchKernelGGL(tm_keys_kernel, dim3(cdiv(n, 256)), dim3(256), 0, s, kpts_tab, n_tab, origins_xy, n_tiles, cap, image_h, image_w, 2.0f, kp_shift, keys);
  if (select_unique) {
    DIM_HIP(hipMemsetAsync(rank, 0, (size_t)n * sizeof(int), s));
    hipLaunchKernelGGL(tm_rank_kernel, dim3(cdiv(n, 256), RANK_PARTS), dim3(256), 0, s, keys, n, rank);
    hipLaunchKernelGGL(tm_scatter_kernel, dim3(cdiv(n, 256)), dim3(256), 0, s, keys, rank, n, order, skeys);
    hipLaunchKernelGGL(tm_scan_kernel, dim3(1), dim3(1024), 0, s, skeys, n, 1, src, order, n_out);
  } else {  // concatenation order: rank r = row r
    hipLaunchKernelGGL(tm_scan_kernel, dim3(1), dim3(1024), 0, s, keys, n, 0, src, (const int*)nullptr, n_out);
  }
  hipLaunchKernelGGL(tm_emit_kernel, dim3(cdiv(n, 256)), dim3(256), 0, s, src, n_out, kp_shift, scores_tab, tile_ids, cap, out_kpts, out_scores, out_tile_idx);
  hipLaunchKernelGGL(tm_desc_kernel, dim3(cdiv(n, 64), cdiv(D, 128)), dim3(256), 0, s, src, n_out, desc_tab, D, out_desc);
  DIM_LAUNCH_CHECK();
  return 0;
}

}  // extern "C"
