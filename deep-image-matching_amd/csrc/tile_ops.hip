// Tile preselection on device (tile_selection, PRESELECTION branch, matchers/matcher_base.py:1054-1133):
// the down-sampling of both images and the per-tile-pair vote count, so that the whole preselection
// (resize -> SuperPoint -> LightGlue -> votes) runs without leaving HBM.
#include <math.h>

#include "../../include/dim_hip.h"
#include "dim_kernels.h"

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
// cv::resize computes inv_scale = (double)dsize / ssize first and scale = 1. / inv_scale from it (resize.cpp): that is NOT always the
// double nearest to ssize / dsize (1 ulp apart for some non-power-of-two ratios), and a floor / weight can depend on it
__device__ __forceinline__ double cv_scale(int ssize, int dsize) { return 1.0 / ((double)dsize / (double)ssize); }
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
    const AreaTaps tx = area_taps(dx, W, cv_scale(W, w)), ty = area_taps(dy, H, cv_scale(H, h));
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

// cv2.resize(..., INTER_AREA) when the image is NOT decimated on both axes (OpenCV 4.11 imgproc/resize.cpp: "true area
// interpolation is only implemented for the case (scale_x >= 1 && scale_y >= 1); in other cases it is emulated using some
// variant of bilinear"): per destination index d the source index is s = floor(d * scale) and the weight of s + 1 is
// f = (d + 1) - (s + 1) / scale, clamped to 0 when negative, fractional part otherwise; at the last source pixel f = 0.
// Rows are interpolated along x first (S[s] * (1 - f) + S[s + 1] * f in fp32), then along y, like HResizeLinear /
// VResizeLinear<float>.  pairs_from_lowres and the tile preselection up-sample every image whose long side is below
// resize_max / tile_preselection_size this way (pairs_generator.py:141-146, matcher_base.py:1062-1069).
struct LinTap { int s; float f; };
__device__ __forceinline__ LinTap area_linear_tap(int d, int ssize, int dsize) {
  const double inv_scale = (double)dsize / (double)ssize, scale = 1.0 / inv_scale;   // OpenCV's order (cv_scale)
  LinTap t;
  t.s = (int)floor(d * scale);
  float f = (float)((double)(d + 1) - (double)(t.s + 1) * inv_scale);
  t.f = f <= 0.f ? 0.f : f - floorf(f);
  if (t.s >= ssize - 1) { t.f = 0.f; t.s = ssize - 1; }
  return t;
}
__global__ __launch_bounds__(256) void resize_area_linear_kernel(const float* __restrict__ src, int H, int W, float* __restrict__ dst,
                                                                 int h, int w, int div255) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= h * w) return;
  const int dy = i / w, dx = i - dy * w;
  const LinTap tx = area_linear_tap(dx, W, w), ty = area_linear_tap(dy, H, h);
  const int y0 = ty.s, y1 = min(ty.s + 1, H - 1);
  auto hrow = [&](int y) -> float {
    const float* r = src + (size_t)y * W;
    if (tx.s >= W - 1) return r[W - 1];                       // the "dx >= xmax" tail of HResizeLinear: a plain copy
    return r[tx.s] * (1.0f - tx.f) + r[tx.s + 1] * tx.f;
  };
  const float out = hrow(y0) * (1.0f - ty.f) + hrow(y1) * ty.f;
  dst[i] = div255 ? out / 255.0f : out;
}

// cv2.resize(..., INTER_LINEAR) of a single-channel fp32 image (OpenCV 4.11 imgproc/resize.cpp, resizeGeneric_ with
// HResizeLinear / VResizeLinear<float>): pixel centres, fx = (float)((dx + 0.5) * scale - 0.5), s = floor(fx), fx -= s;
// s < 0 -> (0, 0); s >= ssize - 1 -> (ssize - 1, 0) and the horizontal pass copies S[s] there; rows are clamped instead
// (the vertical weights keep their fraction).  This is what utils/image.py:52-57 resize_image switches to when the
// requested size ENLARGES the image (quality HIGHEST, extractor_base.py:392-412 / matcher_base.py:1026-1034).
__device__ __forceinline__ LinTap linear_tap(int d, int ssize, int dsize) {
  const double scale = cv_scale(ssize, dsize);
  float fx = (float)(((double)d + 0.5) * scale - 0.5);
  LinTap t;
  t.s = (int)floorf(fx);
  t.f = fx - (float)t.s;
  return t;
}
__global__ __launch_bounds__(256) void resize_linear_kernel(const float* __restrict__ src, int H, int W, float* __restrict__ dst, int h, int w,
                                                            int div255) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= h * w) return;
  const int dy = i / w, dx = i - dy * w;
  LinTap tx = linear_tap(dx, W, w);
  const LinTap ty = linear_tap(dy, H, h);
  if (tx.s < 0) { tx.s = 0; tx.f = 0.f; }
  const bool copy = tx.s + 1 >= W;   // dx >= xmax
  if (tx.s >= W - 1) { tx.s = W - 1; tx.f = 0.f; }
  const int y0 = min(max(ty.s, 0), H - 1), y1 = min(max(ty.s + 1, 0), H - 1);
  auto hrow = [&](int y) -> float {
    const float* r = src + (size_t)y * W;
    if (copy) return r[tx.s];
    return r[tx.s] * (1.0f - tx.f) + r[tx.s + 1] * tx.f;
  };
  const float out = hrow(y0) * (1.0f - ty.f) + hrow(y1) * ty.f;
  dst[i] = div255 ? out / 255.0f : out;
}

// Tile slicing for _extract_by_tile (extractors/extractor_base.py:279-328: the image is zero padded by Tiler.compute_tiles_by_size,
// unfolded into windows, and every window goes through _frame2tensor's / 255): one pass from the image as it arrived on the
// device ([H][W][C] fp32, 0..255) straight into the extractor's batch [n][th][tw][C]; pixels outside the image are the zero
// padding.  No padded copy of the image, no stack, no separate division pass.
__global__ __launch_bounds__(256) void gather_tiles_kernel(const float* __restrict__ img, int H, int W, int C, const int* __restrict__ origins_xy,
                                                           int th, int tw, float* __restrict__ out, int div255) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  const int tile = blockIdx.y;
  if (i >= (size_t)th * tw * C) return;
  const int c = (int)(i % C);
  const size_t p = i / C;
  const int y = (int)(p / tw), x = (int)(p - (size_t)y * tw);
  const int gy = origins_xy[2 * tile + 1] + y, gx = origins_xy[2 * tile] + x;
  float v = 0.0f;
  if (gy >= 0 && gy < H && gx >= 0 && gx < W) v = img[((size_t)gy * W + gx) * C + c];
  out[(size_t)tile * th * tw * C + i] = div255 ? v / 255.0f : v;
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
// Retrieval pair selection (thirdparty/hloc/pairs_from_retrieval.py:49-70): per query row mask the invalid entries
// (self matches, score < min_score) to -inf and take the top-k — k passes of a workgroup arg-max (ties: lowest index),
// the winner is knocked out in place.  sim is the scratch similarity matrix written by the GEMM that precedes it.
__global__ __launch_bounds__(256) void retrieval_topk_kernel(float* __restrict__ sim, const unsigned char* __restrict__ invalid, int nd, int k,
                                                             float min_score, int use_min, int* __restrict__ idx_out, float* __restrict__ val_out) {
  __shared__ float sv[4];
  __shared__ int si[4];
  const int row = blockIdx.x, t = threadIdx.x;
  float* r = sim + (size_t)row * nd;
  const unsigned char* inv = invalid ? invalid + (size_t)row * nd : nullptr;
  for (int j = t; j < nd; j += 256) {
    const float v = r[j];
    if ((inv && inv[j]) || (use_min && v < min_score)) r[j] = -INFINITY;
  }
  __syncthreads();
  for (int pass = 0; pass < k; ++pass) {
    float bv = -INFINITY; int bi = 0x7fffffff;
    for (int j = t; j < nd; j += 256) { const float v = r[j]; if (v > bv || (v == bv && j < bi)) { bv = v; bi = j; } }
    for (int o = 32; o > 0; o >>= 1) {
      const float ov = __shfl_xor(bv, o); const int oi = __shfl_xor(bi, o);
      if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
    }
    if ((t & 63) == 0) { sv[t >> 6] = bv; si[t >> 6] = bi; }
    __syncthreads();
    if (t == 0) {
      for (int w = 1; w < 4; ++w) if (sv[w] > bv || (sv[w] == bv && si[w] < bi)) { bv = sv[w]; bi = si[w]; }
      const bool ok = bv > -INFINITY && bi < nd;
      idx_out[(size_t)row * k + pass] = ok ? bi : -1;
      val_out[(size_t)row * k + pass] = ok ? bv : -INFINITY;
      if (ok) r[bi] = -INFINITY;
    }
    __syncthreads();
  }
}
}  // namespace

extern "C" {

int dim_op_retrieval_topk(const float* query_dev, int nq, const float* db_dev, int nd, int dim, const unsigned char* invalid_dev, int num_select,
                          float min_score, int use_min_score, float* sim_scratch_dev, int* indices_dev, float* values_dev, void* stream) {
  DIM_REQUIRE(query_dev && db_dev && sim_scratch_dev && indices_dev && values_dev, "dim_op_retrieval_topk: null argument");
  DIM_REQUIRE(nq > 0 && nd > 0 && dim > 0 && dim % 32 == 0 && num_select > 0, "dim_op_retrieval_topk: bad sizes (dim must be a multiple of 32)");
  GemmArgs g;  // sim = query * db^T on the exact fp32 MFMA path (einsum("id,jd->ij"), pairs_from_retrieval.py:108)
  g.A0 = query_dev; g.lda0 = dim; g.B = db_dev; g.ldb = dim; g.bt = 1; g.C = sim_scratch_dev; g.ldc = nd; g.M = nq; g.N = nd; g.K = dim;
  const int rc = launch_gemm(g, 1, (hipStream_t)stream);
  if (rc != 0) return rc;
  hipLaunchKernelGGL(retrieval_topk_kernel, dim3(nq), dim3(256), 0, (hipStream_t)stream, sim_scratch_dev, invalid_dev, nd, num_select, min_score,
                     use_min_score, indices_dev, values_dev);
  DIM_LAUNCH_CHECK();
  return 0;
}

int dim_op_resize_area_f32(const float* src, int H, int W, float* dst, int h, int w, int div255, void* stream) {
  DIM_REQUIRE(src && dst && H > 0 && W > 0 && h > 0 && w > 0, "dim_op_resize_area_f32: bad arguments");
  if (h > H || w > W) {  // not a decimation on both axes: OpenCV's bilinear emulation of INTER_AREA
    hipLaunchKernelGGL(resize_area_linear_kernel, dim3(cdiv(h * w, 256)), dim3(256), 0, (hipStream_t)stream, src, H, W, dst, h, w, div255);
    DIM_LAUNCH_CHECK();
    return 0;
  }
  const int fast = (H % h == 0) && (W % w == 0);
  hipLaunchKernelGGL(resize_area_kernel, dim3(cdiv(h * w, 256)), dim3(256), 0, (hipStream_t)stream, src, H, W, dst, h, w, fast, div255);
  DIM_LAUNCH_CHECK();
  return 0;
}

int dim_op_resize_linear_f32(const float* src, int H, int W, float* dst, int h, int w, int div255, void* stream) {
  DIM_REQUIRE(src && dst && H > 0 && W > 0 && h > 0 && w > 0, "dim_op_resize_linear_f32: bad arguments");
  hipLaunchKernelGGL(resize_linear_kernel, dim3(cdiv(h * w, 256)), dim3(256), 0, (hipStream_t)stream, src, H, W, dst, h, w, div255);
  DIM_LAUNCH_CHECK();
  return 0;
}

int dim_op_gather_tiles_f32(const float* image_dev, int H, int W, int C, const int32_t* origins_xy_dev, int n_tiles, int tile_h, int tile_w,
                            float* out_dev, int div255, void* stream) {
  DIM_REQUIRE(image_dev && origins_xy_dev && out_dev && H > 0 && W > 0 && C > 0 && n_tiles > 0 && tile_h > 0 && tile_w > 0, "dim_op_gather_tiles_f32: bad arguments");
  const size_t per = (size_t)tile_h * tile_w * C;
  hipLaunchKernelGGL(gather_tiles_kernel, dim3((unsigned)((per + 255) / 256), n_tiles), dim3(256), 0, (hipStream_t)stream, image_dev, H, W, C,
                     origins_xy_dev, tile_h, tile_w, out_dev, div255);
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
