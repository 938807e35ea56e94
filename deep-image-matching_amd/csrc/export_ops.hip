// Writers off the critical path: the device half of save_features_h5 (extractors/extractor_base.py:56-99).
// The reference converts every array to float16 on the host (EB:60-67), transposes the descriptors to (D, N) in the
// extractor wrapper (extractors/superpoint.py:121-127) and deflates what is left.  Here the conversion, the (N, D) -> (D, N)
// transpose and the un-padding to the live keypoint count happen in HBM, so that ONE device-to-host copy of fp16 data (half
// the PCIe bytes of the fp32 tables) lands in a pinned slot whose sections are already the byte images of the datasets
// features.h5 stores — the writer threads only deflate.
//
// Slot layout per image, in fp16 elements (slot_halves = cap * (4 + D)):
//   [0, 2 n)                      keypoints   (n, 2)
//   [2 cap, 2 cap + n)            scores      (n,)
//   [3 cap, 3 cap + n)            tile_idx    (n,)   (0 when no tile table is given)
//   [4 cap, 4 cap + D n)          descriptors (D, n), row stride n — a contiguous (D, n) array
// n = min(n_kpts[b], cap).  Conversion is round-to-nearest-even with overflow to inf, exactly numpy's astype(float16).
#include "../../include/dim_hip.h"
#include "dim_common.h"

namespace {

constexpr int PK_T = 64;  // transpose tile: 64 keypoints x 64 descriptor dims

__device__ __forceinline__ unsigned short f2h(float v) {
  const _Float16 h = (_Float16)v;
  unsigned short u;
  __builtin_memcpy(&u, &h, 2);
  return u;
}

// grid (ceil(cap / 64), D / 64 + 1, batch); block 256.  y < D/64: one 64 x 64 descriptor tile through LDS (reads coalesced
// along d, writes coalesced along k); y == D/64: keypoints, scores, tile ids of 64 keypoints.
__global__ __launch_bounds__(256) void pack_features_f16_kernel(const float* __restrict__ kpts, const float* __restrict__ scores,
                                                                const float* __restrict__ desc, const int* __restrict__ n_kpts,
                                                                const int* __restrict__ tile_idx, int cap, int D,
                                                                unsigned short* __restrict__ out) {
  __shared__ float tile[PK_T][PK_T + 1];
  const int b = blockIdx.z, k0 = blockIdx.x * PK_T, t = threadIdx.x;
  const int n = min(n_kpts[b], cap);
  if (k0 >= n) return;
  unsigned short* slot = out + (size_t)b * cap * (4 + D);
  if ((int)blockIdx.y == D / PK_T) {
    if (t < PK_T && k0 + t < n) {
      const int k = k0 + t;
      const float2 p = *(const float2*)(kpts + ((size_t)b * cap + k) * 2);
      slot[2 * k] = f2h(p.x);
      slot[2 * k + 1] = f2h(p.y);
      slot[2 * cap + k] = f2h(scores[(size_t)b * cap + k]);
      slot[3 * cap + k] = f2h(tile_idx ? (float)tile_idx[(size_t)b * cap + k] : 0.0f);
    }
    return;
  }
  const int d0 = blockIdx.y * PK_T;
  const int tx = t & 63, ty = t >> 6;  // 4 rows of 64 per pass
#pragma unroll
  for (int r = 0; r < PK_T; r += 4) {
    const int k = k0 + r + ty;
    tile[r + ty][tx] = k < n ? desc[((size_t)b * cap + k) * D + d0 + tx] : 0.0f;
  }
  __syncthreads();
  unsigned short* dst = slot + 4 * (size_t)cap;
#pragma unroll
  for (int r = 0; r < PK_T; r += 4) {
    const int d = d0 + r + ty, k = k0 + tx;
    if (k < n) dst[(size_t)d * n + k] = f2h(tile[tx][r + ty]);
  }
}

// Verified match lists on the device (matchers/matcher_base.py:287-339): per pair, "fewer than 8 raw matches -> skip", then
// num_inliers < min_inliers or ratio < min_ratio -> skip, else the rows of the inlier mask in order.  One workgroup per pair:
// ballot-free block scan over the mask.  n_ver[p] = -1 for a skipped pair.
__global__ __launch_bounds__(256) void filter_matches_kernel(const long long* __restrict__ matches, const int* __restrict__ n_matches,
                                                             const unsigned char* __restrict__ mask, int nk, int min_inliers,
                                                             double min_ratio, long long* __restrict__ ver, int* __restrict__ n_ver) {
  __shared__ int part[256];
  __shared__ int total;
  const int p = blockIdx.x, t = threadIdx.x;
  const int s = min(n_matches[p], nk);
  const long long* m = matches + (size_t)p * nk * 2;
  const unsigned char* mk = mask + (size_t)p * nk;
  long long* o = ver + (size_t)p * nk * 2;
  const int per = (s + 255) / 256;
  const int lo = min(t * per, s), hi = min(lo + per, s);
  int c = 0;
  for (int i = lo; i < hi; ++i) c += mk[i] ? 1 : 0;
  part[t] = c;
  __syncthreads();
  if (t == 0) {
    int acc = 0;
    for (int i = 0; i < 256; ++i) { const int v = part[i]; part[i] = acc; acc += v; }
    total = acc;
  }
  __syncthreads();
  const int n_in = total;
  // the reference compares Python floats: num_inliers / len(matches) < min_inlier_ratio_per_pair, evaluated in fp64
  const bool drop = s < 8 || n_in < min_inliers || (double)n_in / (double)(s > 0 ? s : 1) < min_ratio;
  if (t == 0) n_ver[p] = drop ? -1 : n_in;
  if (drop) return;
  int w = part[t];
  for (int i = lo; i < hi; ++i)
    if (mk[i]) { o[2 * (size_t)w] = m[2 * (size_t)i]; o[2 * (size_t)w + 1] = m[2 * (size_t)i + 1]; ++w; }
}

// Match tables for the end-of-job exchange (SURVEY §8(e) phase 4): dim_lg_match's (idx0, idx1) int64 rows + fp32 scores ->
// 12-byte (idx0:int32, idx1:int32, score bits) rows, zero beyond the pair's live count, so that ONE all-gather of a flat
// int32 buffer carries counts and rows; and the inverse with the round-robin shard order undone (pair p of the job was
// matched by rank p % world as its (p / world)-th pair; src_of_pair maps p -> row of the gathered buffer).
__global__ __launch_bounds__(256) void pack_match_rows_kernel(const long long* __restrict__ matches, const float* __restrict__ scores,
                                                              const int* __restrict__ n_matches, int nk, int* __restrict__ rows) {
  const int p = blockIdx.y, i = blockIdx.x * 256 + threadIdx.x;
  if (i >= nk) return;
  const bool live = i < n_matches[p];
  const size_t r = (size_t)p * nk + i;
  rows[3 * r] = live ? (int)matches[2 * r] : 0;
  rows[3 * r + 1] = live ? (int)matches[2 * r + 1] : 0;
  rows[3 * r + 2] = live ? __float_as_int(scores[r]) : 0;
}
__global__ __launch_bounds__(256) void unpack_match_rows_kernel(const int* __restrict__ rows, const int* __restrict__ src_of_pair, int nk,
                                                                long long* __restrict__ matches, float* __restrict__ scores) {
  const int p = blockIdx.y, i = blockIdx.x * 256 + threadIdx.x;
  if (i >= nk) return;
  const size_t s = ((size_t)(src_of_pair ? src_of_pair[p] : p) * nk + i) * 3, d = (size_t)p * nk + i;
  matches[2 * d] = rows[s]; matches[2 * d + 1] = rows[s + 1];
  scores[d] = __int_as_float(rows[s + 2]);
}

}  // namespace

size_t dim_pack_features_slot_halves(int cap, int D) { return (size_t)cap * (size_t)(4 + D); }

int dim_op_pack_features_f16(const float* kpts_dev, const float* scores_dev, const float* desc_dev, const int32_t* n_kpts_dev,
                             const int32_t* tile_idx_dev, int batch, int cap, int D, void* out_f16_dev, void* stream) {
  DIM_REQUIRE(kpts_dev && scores_dev && desc_dev && n_kpts_dev && out_f16_dev, "dim_op_pack_features_f16: null argument");
  DIM_REQUIRE(batch > 0 && cap > 0 && D > 0 && D % PK_T == 0, "dim_op_pack_features_f16: D must be a multiple of 64 (got %d)", D);
  hipLaunchKernelGGL(pack_features_f16_kernel, dim3(cdiv(cap, PK_T), D / PK_T + 1, batch), dim3(256), 0, (hipStream_t)stream, kpts_dev,
                     scores_dev, desc_dev, n_kpts_dev, tile_idx_dev, cap, D, (unsigned short*)out_f16_dev);
  DIM_LAUNCH_CHECK();
  return 0;
}

int dim_op_filter_matches(const int64_t* matches_dev, const int32_t* n_matches_dev, const unsigned char* mask_dev, int nk, int n_pairs,
                          int min_inliers, double min_ratio, int64_t* verified_dev, int32_t* n_verified_dev, void* stream) {
  DIM_REQUIRE(matches_dev && n_matches_dev && mask_dev && verified_dev && n_verified_dev, "dim_op_filter_matches: null argument");
  DIM_REQUIRE(nk > 0 && n_pairs > 0, "dim_op_filter_matches: bad sizes");
  hipLaunchKernelGGL(filter_matches_kernel, dim3(n_pairs), dim3(256), 0, (hipStream_t)stream, (const long long*)matches_dev, n_matches_dev,
                     mask_dev, nk, min_inliers, min_ratio, (long long*)verified_dev, n_verified_dev);
  DIM_LAUNCH_CHECK();
  return 0;
}

int dim_op_pack_match_rows(const int64_t* matches_dev, const float* scores_dev, const int32_t* n_matches_dev, int nk, int n_pairs,
                           int32_t* rows_dev, void* stream) {
  DIM_REQUIRE(matches_dev && scores_dev && n_matches_dev && rows_dev && nk > 0 && n_pairs > 0, "dim_op_pack_match_rows: bad arguments");
  hipLaunchKernelGGL(pack_match_rows_kernel, dim3(cdiv(nk, 256), n_pairs), dim3(256), 0, (hipStream_t)stream, (const long long*)matches_dev,
                     scores_dev, n_matches_dev, nk, rows_dev);
  DIM_LAUNCH_CHECK();
  return 0;
}

int dim_op_unpack_match_rows(const int32_t* rows_dev, const int32_t* src_of_pair_dev, int nk, int n_pairs, int64_t* matches_dev,
                             float* scores_dev, void* stream) {
  DIM_REQUIRE(rows_dev && matches_dev && scores_dev && nk > 0 && n_pairs > 0, "dim_op_unpack_match_rows: bad arguments");
  hipLaunchKernelGGL(unpack_match_rows_kernel, dim3(cdiv(nk, 256), n_pairs), dim3(256), 0, (hipStream_t)stream, rows_dev, src_of_pair_dev, nk,
                     (long long*)matches_dev, scores_dev);
  DIM_LAUNCH_CHECK();
  return 0;
}
