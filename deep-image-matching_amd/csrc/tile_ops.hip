// Tile preselection on device (tile_selection, PRESELECTION branch, matchers/matcher_base.py:1054-1133):
// the down-sampling of both images and the per-tile-pair vote count, so that the whole preselection
// (resize -> SuperPoint -> LightGlue -> votes) runs without leaving HBM.
#include "../../include/dim_hip.h"
#include "dim_common.h"

namespace {

// One axis of OpenCV's INTER_AREA decimation table (imgproc/resize.cpp, computeResizeAreaTab, 4.11):
// destination cell d covers source [d*scale, (d+1)*scale); partially covered border pixels get their
// covered fraction, everything is normalised by the cell width.  Evaluated in fp64 like the library,
// weights rounded to fp32.
struct AreaTaps {
  int first, n;        // source index of tap 0, number of taps
  int sx1, sx2;        // fully covered range [sx1, sx2)
  float a_head, a_mid, a_tail;
  bool head, tail;
  __device__ float weight(int i) const { return (head && i == 0) ? a_head : ((tail && i == n - 1) ? a_tail : a_mid); }
};
__device__ __forceinline__ AreaTaps area_taps(int d, int ssize, double scale) {
  const double f1 = d * scale, f2 = f1 + scale;
  const double cell = fmin(scale, (double)ssize - f1);
  int s1 = (int)ceil(f1), s2 = (int)floor(f2);
  s2 = min(s2, ssize - 1);
  s1 = min(s1, s2);
  AreaTaps t;
  t.sx1 = s1; t.sx2 = s2;
  t.head = (double)s1 - f1 > 1e-3;
  t.tail = f2 - (double)s2 > 1e-3;
  t.a_head = (float)(((double)s1 - f1) / cell);
  t.a_mid = (float)(1.0 / cell);
  t.a_tail = (float)(fmin(fmin(f2 - (double)s2, 1.0), cell) / cell);
  t.first = t.head ? s1 - 1 : s1;
  t.n = (s2 - s1) + (t.head ? 1 : 0) + (t.tail ? 1 : 0);
  return t;
}

// thread = destination pixel.  General path: rows are first reduced along x (buf += S * alpha, tap order),
// then accumulated along y (sum += beta * buf), all in fp32 like ResizeArea_Invoker<float, float>.
// Integer-ratio path (ResizeAreaFast): plain sum over the block in row-major order, times 1/area.
__global__ __launch_bounds__(256) void resize_area_kernel(const float* __restrict__ src, int H, int W, float* __restrict__ dst, int h,
                                                          int w, int fast, int div255) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= h * w) return;
  const int dy = i / w, dx = i - dy * w;
  float out;
  if (fast) {
    const int sx = W / w, sy = H / h;
    float sum = 0.f;
    for (int y = 0; y < sy; ++y)
      for (int x = 0; x < sx; ++x) sum += src[(size_t)(dy * sy + y) * W + dx * sx + x];
    out = sum * (float)(1.0 / (double)(sx * sy));
  } else {
    const AreaTaps tx = area_taps(dx, W, (double)W / (double)w), ty = area_taps(dy, H, (double)H / (double)h);
    float sum = 0.f;
    for (int j = 0; j < ty.n; ++j) {
      const float* row = src + (size_t)(ty.first + j) * W + tx.first;
      float buf = 0.f;
      for (int k = 0; k < tx.n; ++k) buf += row[k] * tx.weight(k);
      sum += ty.weight(j) * buf;
    }
    out = sum;
  }
  dst[i] = div255 ? out / 255.0f : out;
}

// thread = match.  Both keypoints are scaled back to full resolution (kp / scale, fp32 like numpy) and
// tested against every tile rectangle with the strict inequalities of points_in_rect (MB:1410-1412).
__global__ __launch_bounds__(256) void tile_votes_kernel(const float* __restrict__ k0, const float* __restrict__ k1,
                                                         const long long* __restrict__ matches, const int* __restrict__ n_matches,
                                                         int max_matches, float scale0, float scale1, const int* __restrict__ org0,
                                                         int T0, const int* __restrict__ org1, int T1, int tw, int th,
                                                         int* __restrict__ votes) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= min(*n_matches, max_matches)) return;
  const long long a = matches[2 * (size_t)i], b = matches[2 * (size_t)i + 1];
  const float x0 = k0[2 * a] / scale0, y0 = k0[2 * a + 1] / scale0;
  const float x1 = k1[2 * b] / scale1, y1 = k1[2 * b + 1] / scale1;
  for (int t0 = 0; t0 < T0; ++t0) {
    const float ox = (float)org0[2 * t0], oy = (float)org0[2 * t0 + 1];
    if (!(x0 > ox && y0 > oy && x0 < ox + (float)tw && y0 < oy + (float)th)) continue;
    for (int t1 = 0; t1 < T1; ++t1) {
      const float px = (float)org1[2 * t1], py = (float)org1[2 * t1 + 1];
      if (x1 > px && y1 > py && x1 < px + (float)tw && y1 < py + (float)th) atomicAdd(&votes[t0 * T1 + t1], 1);
    }
  }
}
}  // namespace

extern "C" {

int dim_op_resize_area_f32(const float* src, int H, int W, float* dst, int h, int w, int div255, void* stream) {
  DIM_REQUIRE(src && dst && H > 0 && W > 0 && h > 0 && w > 0, "dim_op_resize_area_f32: bad arguments");
  DIM_REQUIRE(h <= H && w <= W, "dim_op_resize_area_f32: %dx%d -> %dx%d is not a decimation (INTER_AREA enlargement is not built)", H, W, h, w);
  const int fast = (H % h == 0) && (W % w == 0);
  hipLaunchKernelGGL(resize_area_kernel, dim3(cdiv(h * w, 256)), dim3(256), 0, (hipStream_t)stream, src, H, W, dst, h, w, fast, div255);
  DIM_LAUNCH_CHECK();
  return 0;
}

int dim_op_tile_pair_votes(const float* kpts0_xy, const float* kpts1_xy, const long long* matches, const int* n_matches_dev,
                           int max_matches, float scale0, float scale1, const int* origins0_xy, int T0, const int* origins1_xy, int T1,
                           int tile_w, int tile_h, int* votes, void* stream) {
  DIM_REQUIRE(kpts0_xy && kpts1_xy && matches && n_matches_dev && origins0_xy && origins1_xy && votes, "dim_op_tile_pair_votes: null argument");
  DIM_REQUIRE(max_matches >= 0 && T0 > 0 && T1 > 0 && tile_w > 0 && tile_h > 0 && scale0 > 0.f && scale1 > 0.f, "dim_op_tile_pair_votes: bad sizes");
  DIM_HIP(hipMemsetAsync(votes, 0, (size_t)T0 * T1 * sizeof(int), (hipStream_t)stream));
  if (max_matches == 0) return 0;
  hipLaunchKernelGGL(tile_votes_kernel, dim3(cdiv(max_matches, 256)), dim3(256), 0, (hipStream_t)stream, kpts0_xy, kpts1_xy, matches,
                     n_matches_dev, max_matches, scale0, scale1, origins0_xy, T0, origins1_xy, T1, tile_w, tile_h, votes);
  DIM_LAUNCH_CHECK();
  return 0;
}

}  // extern "C"
