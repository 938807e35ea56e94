// Batched geometric verification on the device: a fundamental-matrix RANSAC over the match tables that
// dim_lg_match leaves in HBM, replacing the per-pair host call geometric_verification(...) that follows _match_pairs in
// the reference (utils/geometric_verification.py:45-179, called at matchers/matcher_base.py:311 with cv2's USAC_MAGSAC /
// RANSAC or pydegensac).  Those estimators are randomised and not result-identical to one another, so the contract here
// is the reference's INTERFACE (keypoint pairs + pixel threshold -> F and a boolean inlier mask, "fewer than 8 matches
// -> everything is an inlier") with a deterministic algorithm that oracle/geom_ref.py restates in numpy:
//
//   1. Hartley normalisation of both point sets (centroid, mean distance sqrt 2), fp64.
//   2. `iters` hypotheses from 7-point minimal samples drawn with a counter-based integer hash (seed, pair, hypothesis,
//      draw): null space of the 7x9 system by Gauss-Jordan elimination with full pivoting, det(a F1 + (1 - a) F2) = 0
//      solved in closed form (1 or 3 real roots), every root scored over ALL matches of the pair with the Sampson
//      distance (or the symmetric epipolar distance cv2.RANSAC uses) against threshold^2; best = most inliers, ties to
//      the lowest (hypothesis, root) index.
//   3. Local optimisation, twice: normalised 8-point least squares on the current inlier set (9x9 normal matrix reduced
//      over the workgroup, smallest eigenvector by cyclic Jacobi, rank 2 enforced by removing the smallest singular
//      direction), accepted when it does not lose inliers.
//
// Mapping: phase A runs `splits` workgroups per pair (grid.x = split, grid.y = pair), one hypothesis per thread and
// round, the pair's correspondences staged once in LDS as float4 (x0, y0, x1, y1) and read as wave-wide broadcasts;
// everything is fp64 VALU work (MI355X: 78.6 TFLOP/s vector fp64) — 4096 hypotheses x 3 roots x 2048 matches are
// ~0.8 GFLOP per pair, i.e. ~1 % of the time LightGlue spends on the same pair.  Phase B (one workgroup per pair) picks
// the best split, refines and writes the mask.  No host round trip, no per-pair launch.
#include <math.h>

#include "../../include/dim_hip.h"
#include "dim_common.h"

namespace {
constexpr int GV_MAX_PTS = 4096;   // correspondences per pair held in LDS (64 KB)
constexpr int GV_T = 256;

struct GvArgs {
  const float* kpts; int cap;                 // feature table [n_img][cap][2]
  const int* pair_idx;                        // [P][2] or nullptr (pair p = slots 2p, 2p+1)
  const long long* matches; const int* n_matches; int nk;   // [P][nk][2], [P]
  double thr2; int iters; int splits; unsigned seed; int err_type;   // err_type 0 = Sampson, 1 = symmetric epipolar (max of the two point-line distances)
  int* best_cnt; int* best_id; double* best_F;   // phase A results [P][splits], [P][splits], [P][splits][9]
  unsigned char* mask; int* n_inl; double* F_out;  // outputs [P][nk], [P], [P][9]
};

__device__ __forceinline__ unsigned gv_hash(unsigned seed, unsigned pair, unsigned hyp, unsigned k) {
  unsigned h = seed ^ (pair * 0x9E3779B9u) ^ (hyp * 0x85EBCA6Bu) ^ (k * 0xC2B2AE35u);
  h ^= h >> 16; h *= 0x7feb352du; h ^= h >> 15; h *= 0x846ca68bu; h ^= h >> 16;
  return h;
}

struct Norm { double cx, cy, s; };   // x_n = s * (x - cx)

// residual of one correspondence under F (row-major, acting on raw pixel coordinates: x1^T F x0)
__device__ __forceinline__ double gv_error(const double* F, double x0, double y0, double x1, double y1, int err_type) {
  const double a0 = F[0] * x0 + F[1] * y0 + F[2], b0 = F[3] * x0 + F[4] * y0 + F[5], c0 = F[6] * x0 + F[7] * y0 + F[8];  // F x0: line in image 1
  const double a1 = F[0] * x1 + F[3] * y1 + F[6], b1 = F[1] * x1 + F[4] * y1 + F[7];                                    // F^T x1: line in image 0
  const double e = x1 * a0 + y1 * b0 + c0;
  const double g0 = a0 * a0 + b0 * b0, g1 = a1 * a1 + b1 * b1;
  if (err_type == 0) { const double g = g0 + g1; return g > 0.0 ? e * e / g : 1e300; }
  const double d0 = g0 > 0.0 ? e * e / g0 : 1e300, d1 = g1 > 0.0 ? e * e / g1 : 1e300;
  return fmax(d0, d1);
}

// F (normalised frame) -> raw pixel frame: T1^T F T0 with T = [[s, 0, -s cx], [0, s, -s cy], [0, 0, 1]]
__device__ void gv_denormalise(const double* Fn, const Norm& n0, const Norm& n1, double* F) {
  double M[9];  // Fn * T0
  for (int r = 0; r < 3; ++r) {
    M[3 * r] = Fn[3 * r] * n0.s;
    M[3 * r + 1] = Fn[3 * r + 1] * n0.s;
    M[3 * r + 2] = Fn[3 * r + 2] - n0.s * (Fn[3 * r] * n0.cx + Fn[3 * r + 1] * n0.cy);
  }
  for (int c = 0; c < 3; ++c) {  // T1^T * M
    F[c] = n1.s * M[c];
    F[3 + c] = n1.s * M[3 + c];
    F[6 + c] = M[6 + c] - n1.s * (n1.cx * M[c] + n1.cy * M[3 + c]);
  }
}

__device__ __forceinline__ double gv_det3(const double* F) {
  return F[0] * (F[4] * F[8] - F[5] * F[7]) - F[1] * (F[3] * F[8] - F[5] * F[6]) + F[2] * (F[3] * F[7] - F[4] * F[6]);
}

// real roots of c3 a^3 + c2 a^2 + c1 a + c0 (Numerical Recipes form); returns the count (0..3)
__device__ int gv_cubic(double c3, double c2, double c1, double c0, double* r) {
  const double scale = fmax(fmax(fabs(c3), fabs(c2)), fmax(fabs(c1), fabs(c0)));
  if (!(scale > 0.0)) return 0;
  if (fabs(c3) < 1e-12 * scale) {
    if (fabs(c2) < 1e-12 * scale) { if (fabs(c1) < 1e-12 * scale) return 0; r[0] = -c0 / c1; return 1; }
    const double disc = c1 * c1 - 4.0 * c2 * c0;
    if (disc < 0.0) return 0;
    const double sq = sqrt(disc), q = -0.5 * (c1 + (c1 >= 0.0 ? sq : -sq));
    r[0] = q / c2; if (q != 0.0) { r[1] = c0 / q; return 2; } return 1;
  }
  const double a = c2 / c3, b = c1 / c3, c = c0 / c3;
  const double Q = (a * a - 3.0 * b) / 9.0, R = (2.0 * a * a * a - 9.0 * a * b + 27.0 * c) / 54.0;
  const double Q3 = Q * Q * Q;
  if (R * R < Q3) {
    const double th = acos(R / sqrt(Q3)), m = -2.0 * sqrt(Q);
    r[0] = m * cos(th / 3.0) - a / 3.0;
    r[1] = m * cos((th + 6.283185307179586476925286766559) / 3.0) - a / 3.0;
    r[2] = m * cos((th - 6.283185307179586476925286766559) / 3.0) - a / 3.0;
    return 3;
  }
  const double A = -(R >= 0.0 ? 1.0 : -1.0) * cbrt(fabs(R) + sqrt(R * R - Q3));
  const double B = A != 0.0 ? Q / A : 0.0;
  r[0] = A + B - a / 3.0;
  return 1;
}

// 7-point solver in the normalised frame.  pts: 7 x (x0, y0, x1, y1).  Writes up to 3 candidate matrices; returns the count.
__device__ int gv_seven_point(const double (*p)[4], double (*Fc)[9]) {
  double A[7][9];
  for (int i = 0; i < 7; ++i) {
    const double x0 = p[i][0], y0 = p[i][1], x1 = p[i][2], y1 = p[i][3];
    A[i][0] = x1 * x0; A[i][1] = x1 * y0; A[i][2] = x1; A[i][3] = y1 * x0; A[i][4] = y1 * y0; A[i][5] = y1; A[i][6] = x0; A[i][7] = y0; A[i][8] = 1.0;
  }
  int col[9];
  for (int j = 0; j < 9; ++j) col[j] = j;
  for (int k = 0; k < 7; ++k) {  // Gauss-Jordan, full pivoting over the remaining block
    int pr = k, pc = k; double best = -1.0;
    for (int i = k; i < 7; ++i)
      for (int j = k; j < 9; ++j) { const double v = fabs(A[i][j]); if (v > best) { best = v; pr = i; pc = j; } }
    if (!(best > 1e-12)) return 0;  // degenerate sample
    if (pr != k) for (int j = 0; j < 9; ++j) { const double t = A[k][j]; A[k][j] = A[pr][j]; A[pr][j] = t; }
    if (pc != k) { for (int i = 0; i < 7; ++i) { const double t = A[i][k]; A[i][k] = A[i][pc]; A[i][pc] = t; } const int t = col[k]; col[k] = col[pc]; col[pc] = t; }
    const double inv = 1.0 / A[k][k];
    for (int j = 0; j < 9; ++j) A[k][j] *= inv;
    for (int i = 0; i < 7; ++i) {
      if (i == k) continue;
      const double f = A[i][k];
      if (f != 0.0) for (int j = 0; j < 9; ++j) A[i][j] -= f * A[k][j];
    }
  }
  double F1[9], F2[9];  // null vectors: free column 7 (resp. 8) = 1, pivot column k = -A[k][7] (resp. -A[k][8])
  for (int j = 0; j < 9; ++j) { F1[j] = 0.0; F2[j] = 0.0; }
  F1[col[7]] = 1.0; F2[col[8]] = 1.0;
  for (int k = 0; k < 7; ++k) { F1[col[k]] = -A[k][7]; F2[col[k]] = -A[k][8]; }
  // det(a F1 + (1 - a) F2) is a cubic in a: interpolate through a = 0, 1, -1, 2
  double G[9], pv[4];
  const double av[4] = {0.0, 1.0, -1.0, 2.0};
  for (int t = 0; t < 4; ++t) { for (int j = 0; j < 9; ++j) G[j] = av[t] * F1[j] + (1.0 - av[t]) * F2[j]; pv[t] = gv_det3(G); }
  const double c0 = pv[0], c2 = 0.5 * (pv[1] + pv[2]) - c0, s = 0.5 * (pv[1] - pv[2]), tt = pv[3] - 4.0 * c2 - c0;
  const double c3 = (tt - 2.0 * s) / 6.0, c1 = s - c3;
  double roots[3];
  const int nr = gv_cubic(c3, c2, c1, c0, roots);
  for (int t = 0; t < nr; ++t)
    for (int j = 0; j < 9; ++j) Fc[t][j] = roots[t] * F1[j] + (1.0 - roots[t]) * F2[j];
  return nr;
}

// cyclic Jacobi on a symmetric N x N matrix (row-major, destroyed); V receives the eigenvectors in columns
template <int N>
__device__ void gv_jacobi(double* A, double* V) {
  for (int i = 0; i < N; ++i) for (int j = 0; j < N; ++j) V[i * N + j] = i == j ? 1.0 : 0.0;
  for (int sweep = 0; sweep < 40; ++sweep) {
    double off = 0.0, dia = 0.0;
    for (int i = 0; i < N; ++i) { dia += A[i * N + i] * A[i * N + i]; for (int j = i + 1; j < N; ++j) off += A[i * N + j] * A[i * N + j]; }
    if (!(off > 1e-30 * dia)) break;
    for (int p = 0; p < N - 1; ++p)
      for (int q = p + 1; q < N; ++q) {
        const double apq = A[p * N + q];
        if (apq == 0.0) continue;
        const double theta = (A[q * N + q] - A[p * N + p]) / (2.0 * apq);
        const double t = (theta >= 0.0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
        const double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
        for (int k = 0; k < N; ++k) { const double akp = A[k * N + p], akq = A[k * N + q]; A[k * N + p] = c * akp - s * akq; A[k * N + q] = s * akp + c * akq; }
        for (int k = 0; k < N; ++k) { const double apk = A[p * N + k], aqk = A[q * N + k]; A[p * N + k] = c * apk - s * aqk; A[q * N + k] = s * apk + c * aqk; }
        for (int k = 0; k < N; ++k) { const double vkp = V[k * N + p], vkq = V[k * N + q]; V[k * N + p] = c * vkp - s * vkq; V[k * N + q] = s * vkp + c * vkq; }
      }
  }
}

// block-wide staging shared by both phases: gather the pair's correspondences into LDS, Hartley statistics in fp64
struct GvShared {
  float4 pts[GV_MAX_PTS];
  double red[4][48];
  double F[9];
  int ibest[4], icnt[4];
  Norm n0, n1;
  int n;
};

__device__ double gv_block_sum(GvShared& sh, double v, int slot) {  // all threads call; result valid after the barrier pair
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  if ((threadIdx.x & 63) == 0) sh.red[threadIdx.x >> 6][slot] = v;
  __syncthreads();
  const double r = sh.red[0][slot] + sh.red[1][slot] + sh.red[2][slot] + sh.red[3][slot];
  __syncthreads();
  return r;
}

__device__ void gv_stage(const GvArgs& a, int pair, GvShared& sh) {
  const int t = threadIdx.x;
  const int n = min(min(a.n_matches[pair], a.nk), GV_MAX_PTS);
  const int i0 = a.pair_idx ? a.pair_idx[2 * pair] : 2 * pair, i1 = a.pair_idx ? a.pair_idx[2 * pair + 1] : 2 * pair + 1;
  const float* k0 = a.kpts + (size_t)i0 * a.cap * 2;
  const float* k1 = a.kpts + (size_t)i1 * a.cap * 2;
  const long long* m = a.matches + (size_t)pair * a.nk * 2;
  double sx0 = 0, sy0 = 0, sx1 = 0, sy1 = 0;
  for (int i = t; i < n; i += GV_T) {
    const long long ia = m[2 * i], ib = m[2 * i + 1];
    const float4 p = make_float4(k0[2 * ia], k0[2 * ia + 1], k1[2 * ib], k1[2 * ib + 1]);
    sh.pts[i] = p;
    sx0 += p.x; sy0 += p.y; sx1 += p.z; sy1 += p.w;
  }
  if (t == 0) sh.n = n;
  const double inv = n > 0 ? 1.0 / n : 0.0;
  const double cx0 = gv_block_sum(sh, sx0, 0) * inv, cy0 = gv_block_sum(sh, sy0, 1) * inv;
  const double cx1 = gv_block_sum(sh, sx1, 2) * inv, cy1 = gv_block_sum(sh, sy1, 3) * inv;
  double d0 = 0, d1 = 0;
  for (int i = t; i < n; i += GV_T) {
    const float4 p = sh.pts[i];
    d0 += sqrt((p.x - cx0) * (p.x - cx0) + (p.y - cy0) * (p.y - cy0));
    d1 += sqrt((p.z - cx1) * (p.z - cx1) + (p.w - cy1) * (p.w - cy1));
  }
  const double m0 = gv_block_sum(sh, d0, 4) * inv, m1 = gv_block_sum(sh, d1, 5) * inv;
  if (t == 0) {
    sh.n0.cx = cx0; sh.n0.cy = cy0; sh.n0.s = m0 > 0.0 ? 1.4142135623730951 / m0 : 1.0;
    sh.n1.cx = cx1; sh.n1.cy = cy1; sh.n1.s = m1 > 0.0 ? 1.4142135623730951 / m1 : 1.0;
  }
  __syncthreads();
}

__device__ int gv_count(const GvShared& sh, const double* F, double thr2, int err_type) {
  int c = 0;
  for (int i = 0; i < sh.n; ++i) {
    const float4 p = sh.pts[i];
    c += gv_error(F, p.x, p.y, p.z, p.w, err_type) <= thr2 ? 1 : 0;
  }
  return c;
}

// ---- phase A: hypotheses --------------------------------------------------------------------------------------------
__global__ __launch_bounds__(GV_T) void gv_hypotheses_kernel(GvArgs a) {
  __shared__ GvShared sh;
  const int split = blockIdx.x, pair = blockIdx.y, t = threadIdx.x;
  gv_stage(a, pair, sh);
  const int n = sh.n;
  int my_cnt = -1, my_id = 0x7fffffff;
  double my_F[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
  if (n >= 8) {
    const int per = (a.iters + a.splits - 1) / a.splits;
    const int h0 = split * per, h1 = min(a.iters, h0 + per);
    for (int hyp = h0 + t; hyp < h1; hyp += GV_T) {
      int idx[7];
      bool ok = true;
      for (int j = 0; j < 7 && ok; ++j) {
        int att = 0;
        for (;;) {
          const int cand = (int)(((unsigned long long)gv_hash(a.seed, (unsigned)pair, (unsigned)hyp, (unsigned)(j + 7 * att)) * (unsigned long long)n) >> 32);
          bool dup = false;
          for (int q = 0; q < j; ++q) dup = dup || idx[q] == cand;
          if (!dup) { idx[j] = cand; break; }
          if (++att >= 8) { ok = false; break; }
        }
      }
      if (!ok) continue;
      double p[7][4];
      for (int j = 0; j < 7; ++j) {
        const float4 q = sh.pts[idx[j]];
        p[j][0] = sh.n0.s * (q.x - sh.n0.cx); p[j][1] = sh.n0.s * (q.y - sh.n0.cy);
        p[j][2] = sh.n1.s * (q.z - sh.n1.cx); p[j][3] = sh.n1.s * (q.w - sh.n1.cy);
      }
      double Fc[3][9];
      const int nr = gv_seven_point(p, Fc);
      for (int r = 0; r < nr; ++r) {
        double F[9];
        gv_denormalise(Fc[r], sh.n0, sh.n1, F);
        const int c = gv_count(sh, F, a.thr2, a.err_type);
        const int id = hyp * 3 + r;
        if (c > my_cnt || (c == my_cnt && id < my_id)) { my_cnt = c; my_id = id; for (int j = 0; j < 9; ++j) my_F[j] = F[j]; }
      }
    }
  }
  // workgroup arg-max (count desc, id asc): wave shuffles, then the four wave winners through LDS
  int bc = my_cnt, bi = my_id;
  for (int o = 32; o > 0; o >>= 1) {
    const int oc = __shfl_xor(bc, o), oi = __shfl_xor(bi, o);
    if (oc > bc || (oc == bc && oi < bi)) { bc = oc; bi = oi; }
  }
  if ((t & 63) == 0) { sh.icnt[t >> 6] = bc; sh.ibest[t >> 6] = bi; }
  __syncthreads();
  for (int w = 0; w < 4; ++w) { const int oc = sh.icnt[w], oi = sh.ibest[w]; if (oc > bc || (oc == bc && oi < bi)) { bc = oc; bi = oi; } }
  if (my_cnt == bc && my_id == bi && bc >= 0) {  // exactly one thread owns (count, id)
    const size_t o = (size_t)pair * a.splits + split;
    a.best_cnt[o] = bc; a.best_id[o] = bi;
    for (int j = 0; j < 9; ++j) a.best_F[o * 9 + j] = my_F[j];
  }
  if (t == 0 && bc < 0) { const size_t o = (size_t)pair * a.splits + split; a.best_cnt[o] = -1; a.best_id[o] = 0x7fffffff; }
}

// ---- phase B: pick, refine, write the mask ----------------------------------------------------------------------------
__global__ __launch_bounds__(GV_T) void gv_refine_kernel(GvArgs a) {
  __shared__ GvShared sh;
  __shared__ double Fcur[9];
  __shared__ int cur_cnt;
  const int pair = blockIdx.x, t = threadIdx.x;
  gv_stage(a, pair, sh);
  const int n = sh.n;
  unsigned char* mask = a.mask + (size_t)pair * a.nk;
  if (n < 8) {  // geometric_verification.py:107-110: not enough matches -> F = None, every match is an inlier
    for (int i = t; i < a.nk; i += GV_T) mask[i] = i < n ? 1 : 0;
    if (t == 0) { a.n_inl[pair] = n; for (int j = 0; j < 9; ++j) a.F_out[(size_t)pair * 9 + j] = 0.0; }
    return;
  }
  if (t == 0) {
    int bc = -1, bi = 0x7fffffff, bs = -1;
    for (int s = 0; s < a.splits; ++s) {
      const int c = a.best_cnt[(size_t)pair * a.splits + s], id = a.best_id[(size_t)pair * a.splits + s];
      if (c > bc || (c == bc && id < bi)) { bc = c; bi = id; bs = s; }
    }
    cur_cnt = bc;
    for (int j = 0; j < 9; ++j) Fcur[j] = bs >= 0 ? a.best_F[((size_t)pair * a.splits + bs) * 9 + j] : 0.0;
  }
  __syncthreads();
  if (cur_cnt < 0) {  // every 7-point sample was degenerate (duplicate / collinear correspondences): the estimator has no
    // model.  geometric_verification.py:150-172 keeps an all-ones mask when its estimator fails; so does this path.
    for (int i = t; i < a.nk; i += GV_T) mask[i] = i < n ? 1 : 0;
    if (t == 0) { a.n_inl[pair] = n; for (int j = 0; j < 9; ++j) a.F_out[(size_t)pair * 9 + j] = 0.0; }
    return;
  }
  for (int lo = 0; lo < 2 && cur_cnt >= 8; ++lo) {
    // normal matrix of the normalised 8-point system over the current inliers: 45 upper-triangle sums per thread
    double acc[45];
    for (int j = 0; j < 45; ++j) acc[j] = 0.0;
    for (int i = t; i < n; i += GV_T) {
      const float4 q = sh.pts[i];
      if (!(gv_error(Fcur, q.x, q.y, q.z, q.w, a.err_type) <= a.thr2)) continue;
      const double x0 = sh.n0.s * (q.x - sh.n0.cx), y0 = sh.n0.s * (q.y - sh.n0.cy), x1 = sh.n1.s * (q.z - sh.n1.cx), y1 = sh.n1.s * (q.w - sh.n1.cy);
      const double r[9] = {x1 * x0, x1 * y0, x1, y1 * x0, y1 * y0, y1, x0, y0, 1.0};
      int k = 0;
      for (int u = 0; u < 9; ++u) for (int v = u; v < 9; ++v) acc[k++] += r[u] * r[v];
    }
    for (int j = 0; j < 45; ++j) {
      double v = acc[j];
      for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
      if ((t & 63) == 0) sh.red[t >> 6][j] = v;
    }
    __syncthreads();
    if (t == 0) {
      double M[81], V[81];
      int k = 0;
      for (int u = 0; u < 9; ++u) for (int v = u; v < 9; ++v) { const double s = sh.red[0][k] + sh.red[1][k] + sh.red[2][k] + sh.red[3][k]; M[u * 9 + v] = s; M[v * 9 + u] = s; ++k; }
      gv_jacobi<9>(M, V);
      int jm = 0;
      for (int j = 1; j < 9; ++j) if (M[j * 9 + j] < M[jm * 9 + jm]) jm = j;
      double Fn[9];
      for (int j = 0; j < 9; ++j) Fn[j] = V[j * 9 + jm];
      // rank 2: remove the direction of the smallest singular value (v3 = eigenvector of Fn^T Fn)
      double G[9], W[9];
      for (int u = 0; u < 3; ++u) for (int v = 0; v < 3; ++v) G[u * 3 + v] = Fn[u] * Fn[v] + Fn[3 + u] * Fn[3 + v] + Fn[6 + u] * Fn[6 + v];
      gv_jacobi<3>(G, W);
      int j3 = 0;
      for (int j = 1; j < 3; ++j) if (G[j * 3 + j] < G[j3 * 3 + j3]) j3 = j;
      const double v3[3] = {W[j3], W[3 + j3], W[6 + j3]};
      for (int r = 0; r < 3; ++r) {
        const double fv = Fn[3 * r] * v3[0] + Fn[3 * r + 1] * v3[1] + Fn[3 * r + 2] * v3[2];
        for (int c = 0; c < 3; ++c) Fn[3 * r + c] -= fv * v3[c];
      }
      gv_denormalise(Fn, sh.n0, sh.n1, sh.F);
    }
    __syncthreads();
    int c = 0;
    for (int i = t; i < n; i += GV_T) { const float4 q = sh.pts[i]; c += gv_error(sh.F, q.x, q.y, q.z, q.w, a.err_type) <= a.thr2 ? 1 : 0; }
    const int tot = (int)(gv_block_sum(sh, (double)c, 46) + 0.5);
    if (t == 0 && tot >= cur_cnt) { cur_cnt = tot; for (int j = 0; j < 9; ++j) Fcur[j] = sh.F[j]; }
    __syncthreads();
  }
  for (int i = t; i < a.nk; i += GV_T) {
    unsigned char v = 0;
    if (i < n) { const float4 q = sh.pts[i]; v = gv_error(Fcur, q.x, q.y, q.z, q.w, a.err_type) <= a.thr2 ? 1 : 0; }
    mask[i] = v;
  }
  if (t == 0) {
    a.n_inl[pair] = cur_cnt;
    // scale like OpenCV (F33 = 1) when possible, else unit Frobenius norm
    double nrm = 0.0;
    for (int j = 0; j < 9; ++j) nrm += Fcur[j] * Fcur[j];
    nrm = sqrt(nrm);
    const double sc = fabs(Fcur[8]) > 1e-12 * nrm ? 1.0 / Fcur[8] : (nrm > 0.0 ? 1.0 / nrm : 0.0);
    for (int j = 0; j < 9; ++j) a.F_out[(size_t)pair * 9 + j] = Fcur[j] * sc;
  }
}
}  // namespace

extern "C" {

static size_t gv_scratch(int n_pairs, int splits) {
  const size_t e = (size_t)(n_pairs > 0 ? n_pairs : 0) * (size_t)(splits > 0 ? splits : 1);
  return e * (2 * sizeof(int) + 9 * sizeof(double)) + 64;
}
size_t dim_gv_scratch_bytes(int n_pairs) { return gv_scratch(n_pairs, 16); }

int dim_gv_fundamental(const float* kpts_tab_dev, int cap, const int32_t* pair_idx_dev, const int64_t* matches_dev,
                       const int32_t* n_matches_dev, int nk, int n_pairs, double threshold_px, int iters, int error_type, unsigned seed,
                       void* scratch_dev, size_t scratch_bytes, unsigned char* inlier_mask_dev, int32_t* n_inliers_dev, double* F_dev,
                       void* stream) {
  DIM_REQUIRE(kpts_tab_dev && matches_dev && n_matches_dev && inlier_mask_dev && n_inliers_dev && F_dev && scratch_dev, "dim_gv_fundamental: null argument");
  DIM_REQUIRE(cap > 0 && nk > 0 && nk <= GV_MAX_PTS, "dim_gv_fundamental: nk=%d outside [1,%d]", nk, GV_MAX_PTS);
  DIM_REQUIRE(threshold_px > 0.0 && iters >= 1 && (error_type == 0 || error_type == 1), "dim_gv_fundamental: bad threshold / iters / error_type");
  if (n_pairs <= 0) return 0;
  GvArgs a;
  a.kpts = kpts_tab_dev; a.cap = cap; a.pair_idx = pair_idx_dev; a.matches = (const long long*)matches_dev; a.n_matches = n_matches_dev; a.nk = nk;
  a.thr2 = threshold_px * threshold_px; a.iters = iters; a.seed = seed; a.err_type = error_type;
  // enough workgroups to fill the chip even for a handful of pairs; each split needs a few rounds of 256 hypotheses to pay
  int splits = (512 + n_pairs - 1) / n_pairs;
  splits = splits < 1 ? 1 : (splits > 16 ? 16 : splits);
  while (splits > 1 && (iters + splits - 1) / splits < GV_T) --splits;
  a.splits = splits;
  DIM_REQUIRE(scratch_bytes >= gv_scratch(n_pairs, splits), "dim_gv_fundamental: scratch too small (%zu < %zu)", scratch_bytes, gv_scratch(n_pairs, splits));
  const size_t e = (size_t)n_pairs * splits;
  a.best_F = (double*)scratch_dev; a.best_cnt = (int*)(a.best_F + e * 9); a.best_id = a.best_cnt + e;
  a.mask = inlier_mask_dev; a.n_inl = n_inliers_dev; a.F_out = F_dev;
  hipLaunchKernelGGL(gv_hypotheses_kernel, dim3(splits, n_pairs), dim3(GV_T), 0, (hipStream_t)stream, a);
  hipLaunchKernelGGL(gv_refine_kernel, dim3(n_pairs), dim3(GV_T), 0, (hipStream_t)stream, a);
  DIM_LAUNCH_CHECK();
  return 0;
}

}  // extern "C"
