// SuperPoint detector/descriptor post-processing for gfx950 — the HBM-bound,
// compare/integer half of the extractor (reference SPN:47-98,176-221).
//
//   softmax_d2s   : 65-way softmax per 8x8 cell, drop dustbin, depth-to-space (SPN:176-179)
//   nms           : simple_nms fused into ONE pass per 32x32 tile with a 5r halo
//                   staged in LDS (the reference runs 5 full-res max-pools, SPN:47-63)
//   count/scan/emit: row-major ordered compaction of (s > thr) & border (SPN:183-196)
//   topk          : radix-select + LDS bitonic sort, score-descending (SPN:74-78,199-207)
//   sample        : bilinear sampling of the dense descriptor map at the keypoints with
//                   both L2 normalisations, touching only the <=4 cells per keypoint
//                   (SPN:81-98,215; DIM's fix_sampling variant SPX:16-27)
#include <math.h>

#include "sp_kernels.h"

namespace {

// ---------------------------------------------------------------------------
// one wave per SD_CPW consecutive cells: lane c holds logit c; lane-uniform dustbin.  The loads of all its cells go out first: with one cell
// per wave the kernel ran at (resident waves) / (load latency) = 2 TB/s, a quarter of what its 8.3 MB per 1024^2 map allow.
constexpr int SD_CPW = 4;
__global__ __launch_bounds__(256) void softmax_d2s_kernel(const float* __restrict__ logits, float* __restrict__ smap,
                                                          int n_cells_total, int h, int w) {
  const int lane = threadIdx.x & 63;
  const int cell0 = (blockIdx.x * 4 + (threadIdx.x >> 6)) * SD_CPW;
  if (cell0 >= n_cells_total) return;
  float v[SD_CPW], dust[SD_CPW];
#pragma unroll
  for (int j = 0; j < SD_CPW; ++j) {
    const float* p = logits + (size_t)min(cell0 + j, n_cells_total - 1) * 65;
    v[j] = p[lane];
    dust[j] = p[64];
  }
  const int W8 = w * 8;
#pragma unroll
  for (int j = 0; j < SD_CPW; ++j) {
    const int cell = cell0 + j;
    if (cell >= n_cells_total) break;
    const float m = fmaxf(wave_max(v[j]), dust[j]);
    const float e = expf(v[j] - m);
    const float s = wave_sum(e) + expf(dust[j] - m);
    const int b = cell / (h * w);
    const int rem = cell - b * h * w;
    const int cy = rem / w, cx = rem - cy * w;
    smap[((size_t)b * h * 8 + cy * 8 + (lane >> 3)) * W8 + cx * 8 + (lane & 7)] = e / s;
  }
}

// ---------------------------------------------------------------------------
// simple_nms (SPN:47-63) fused into ONE pass per TILE x TILE output tile with a 5R halo
// (dependency radius R -> 2R -> ... -> 5R) staged in LDS; T = TILE + 10R,
// row stride TS = T|1 (odd).  Every (2R+1)^2 max-pool is separable; a line filter gives each
// thread one 16-long segment of one line, loads 16+2R values into registers once and emits 16
// window maxima (1 LDS read + 1 write per element instead of 2R+1 reads).  Lane -> line
// mapping keeps both passes bank-conflict free: the row pass strides lanes over y (odd TS),
// the column pass over x.  Every stage covers only the region its inputs are exact on (T shrinks by 2R per
// pool: 62 -> 56 -> 50 -> 44 -> 38 -> 32 for R = 3, TILE = 32): a third less work than full-tile passes.  Arrays: s (scores, -inf outside the image; the sign bit
// marks "near a kept maximum" during a suppress-and-recover round), t (row-pass scratch; the mask passes' byte scratch t8 lives in it), kp (keep mask, bytes).
// window maximum as v_max3 chains.  `a > b ? a : b` on floats compiles to v_cmp_gt_f32 + v_cndmask_b32 (+ an s_nop for the VCC hazard)
// per step — ~15 issue slots per output of a 7-wide window, and every instruction of this VALU-bound kernel is time (round-4
// counters: the float passes were ~2/3 of its 16 us per 1024^2 map); fmaxf nests become v_max3_f32.  Scores are finite or -inf
// (never NaN: a softmax of finite logits), so fmaxf == the comparison form.  t3[k] = max(v[k], v[k+1], v[k+2]) is shared by the
// outputs: a 7-wide window is max3(t3[o], t3[o+3], v[o+6]) = 2.25 instructions per output.
__device__ __forceinline__ float nms_mx3(float a, float b, float c) { return __builtin_fmaxf(__builtin_fmaxf(a, b), c); }
__device__ __forceinline__ unsigned char nms_mx3(unsigned char a, unsigned char b, unsigned char c) {
  const unsigned char m = a > b ? a : b;
  return m > c ? m : c;
}
template <int R, int SEG, typename T>
__device__ __forceinline__ void nms_window_max(const T (&v)[SEG + 2 * R], T (&out)[SEG]) {
  constexpr int W = 2 * R + 1;
  if (R == 0) {
#pragma unroll
    for (int o = 0; o < SEG; ++o) out[o] = v[o];
    return;
  }
  T t3[SEG + 2 * R - 2 > 0 ? SEG + 2 * R - 2 : 1];
#pragma unroll
  for (int k = 0; k < SEG + 2 * R - 2; ++k) t3[k] = nms_mx3(v[k], v[k + 1], v[k + 2]);
#pragma unroll
  for (int o = 0; o < SEG; ++o) {
    T m = t3[o];
    // the remaining W - 3 elements o+3 .. o+W-1: whole t3 blocks while they fit, single elements for the rest, two terms per max3
    if (W == 5) m = nms_mx3(m, v[o + 3], v[o + 4]);
    if (W == 7) m = nms_mx3(m, t3[o + 3], v[o + 6]);
    if (W == 9) m = nms_mx3(m, t3[o + 3], t3[o + 6]);
    if (W == 11) { m = nms_mx3(m, t3[o + 3], t3[o + 6]); m = nms_mx3(m, v[o + 9], v[o + 10]); }
    if (W == 13) { m = nms_mx3(m, t3[o + 3], t3[o + 6]); m = nms_mx3(m, t3[o + 9], v[o + 12]); }
    out[o] = m;
  }
}
template <int R, int SEG, int NMS_THREADS, typename T, bool ROW, bool CLAMP0 = false, typename Emit>
__device__ __forceinline__ void nms_line_max(const T* src, int TS, int l0, int l1, int p0, int p1, Emit emit) {
  // lines l0..l1-1, window maxima at positions p0..p1-1 (the window p-R..p+R always lies inside the array: every
  // stage only covers the region its inputs are valid on, which shrinks by R per pool).  SEG outputs per item: chosen
  // with the thread count so that one pass of the tile is a single, well-filled round of items
  static_assert(R <= 6, "window forms up to 13 wide");
  const int nl = l1 - l0, nseg = (p1 - p0 + SEG - 1) / SEG;
  for (int item = threadIdx.x; item < nl * nseg; item += NMS_THREADS) {
    const int line = l0 + item % nl, seg = item / nl;
    const int base = p0 + seg * SEG - R;
    const int stride = ROW ? 1 : TS;
    const int off = ROW ? line * TS : line;
    T v[SEG + 2 * R];
    if (base + SEG + 2 * R - 1 <= p1 - 1 + R) {   // whole segment inside the line: constant offsets from one base address
      const T* p = src + off + base * stride;
#pragma unroll
      for (int k = 0; k < SEG + 2 * R; ++k) v[k] = p[k * stride];
    } else {
#pragma unroll
      for (int k = 0; k < SEG + 2 * R; ++k) {
        const int q = min(base + k, p1 - 1 + R);  // the last segment may be short: re-read the last needed element
        v[k] = src[off + q * stride];
      }
    }
    if (CLAMP0) {   // float lines only: the values enter the pool as max(v, 0) (nms_kernel's marked scores)
#pragma unroll
      for (int k = 0; k < SEG + 2 * R; ++k) v[k] = (T)__builtin_fmaxf((float)v[k], 0.0f);
    }
    T mo[SEG];
    nms_window_max<R, SEG, T>(v, mo);
#pragma unroll
    for (int o = 0; o < SEG; ++o) {
      const int q = p0 + seg * SEG + o;
      if (q < p1) emit(off + q * stride, mo[o]);
    }
  }
}

// TILE 32 / 512 threads: 35 KB of LDS, (32 + 10 R)^2 / 32^2 = 3.75 x the tile area per pass at R = 3.
// TILE 64 / 1024 threads: 80 KB (R = 3), two workgroups per CU, 2.16 x: 42 % less pass work per output pixel.
template <int R, int TILE, int NMS_THREADS, int SEG>
__global__ __launch_bounds__(NMS_THREADS) void nms_kernel(const float* __restrict__ smap, float* __restrict__ out, int H8, int W8,
                                                  int tiles_x) {
  constexpr int HALO = 5 * R, T = TILE + 2 * HALO, TS = T | 1, TT = T * TS;
  // 9 bytes per tile element (round 4; 14 before): the byte scratch of the mask passes lives in the float scratch (never live together), and the
  // "suppressed scores" of a round are not stored — a pixel near a kept maximum carries the mark in the SIGN BIT of its score (scores are
  // softmax outputs >= +0, or -inf outside the image), the pool input is max(s', 0) and the mark is cleared again for the next round.  80 KB
  // instead of 125 KB for the 64 x 64 tile at R = 3: two workgroups per CU, so that one's load / store / barrier phases overlap the other's passes.
  __shared__ float s[TT];
  __shared__ float t[TT];
  __shared__ unsigned char kp[TT];
  unsigned char* const t8 = (unsigned char*)t;
  const int tid = threadIdx.x, b = blockIdx.z;
  const int ty0 = (blockIdx.x / tiles_x) * TILE - HALO, tx0 = (blockIdx.x % tiles_x) * TILE - HALO;
  const float* src = smap + (size_t)b * H8 * W8;
  const float NEG = -INFINITY;

  for (int i = tid; i < T * T; i += NMS_THREADS) {
    const int y = i / T, x = i - y * T, gy = ty0 + y, gx = tx0 + x;
    s[y * TS + x] = (gy >= 0 && gy < H8 && gx >= 0 && gx < W8) ? src[(size_t)gy * W8 + gx] : NEG;
  }
  __syncthreads();
  // region k = [k R, T - k R)^2: what is still exact after k pools.  Row passes run on the rows of the previous
  // region and the columns of the next one, column passes on the next region.
  auto lo = [](int k) { return k * R; };
  auto hi = [&](int k) { return T - k * R; };
  // round 0: keep = (s == P(s)); the column pass compares in its epilogue
  nms_line_max<R, SEG, NMS_THREADS, float, true>(s, TS, lo(0), hi(0), lo(1), hi(1), [&](int j, float m) { t[j] = m; });
  __syncthreads();
  nms_line_max<R, SEG, NMS_THREADS, float, false>(t, TS, lo(1), hi(1), lo(1), hi(1), [&](int j, float m) { kp[j] = (s[j] != NEG && s[j] == m) ? 1 : 0; });
  __syncthreads();
  // two rounds of suppress-and-recover
#pragma unroll
  for (int round = 0; round < 2; ++round) {
    const int k = 1 + 2 * round;  // kp is exact on region k
    nms_line_max<R, SEG, NMS_THREADS, unsigned char, true>(kp, TS, lo(k), hi(k), lo(k + 1), hi(k + 1), [&](int j, unsigned char m) { t8[j] = m; });
    __syncthreads();
    // near = dilate(keep): marked in the sign bit of the score (-inf outside the image stays what it is; the previous round's marks are replaced)
    nms_line_max<R, SEG, NMS_THREADS, unsigned char, false>(t8, TS, lo(k + 1), hi(k + 1), lo(k + 1), hi(k + 1), [&](int j, unsigned char m) {
      const unsigned bits = __float_as_uint(s[j]);
      if (bits != 0xff800000u) s[j] = __uint_as_float((bits & 0x7fffffffu) | (m ? 0x80000000u : 0u));
    });
    __syncthreads();
    // the suppressed scores of the round: rest = near ? 0 : s = max(s', 0) (outside the image 0 instead of -inf: a window around an in-image
    // pixel holds that pixel's own value >= 0, so its maximum is the same)
    nms_line_max<R, SEG, NMS_THREADS, float, true, true>(s, TS, lo(k + 1), hi(k + 1), lo(k + 2), hi(k + 2), [&](int j, float m) { t[j] = m; });
    __syncthreads();
    nms_line_max<R, SEG, NMS_THREADS, float, false>(t, TS, lo(k + 2), hi(k + 2), lo(k + 2), hi(k + 2), [&](int j, float m) {
      const float sv = s[j];   // in the image: >= +0 and unmarked, or marked (sign bit) = near a kept maximum
      if (sv != NEG && (__float_as_uint(sv) >> 31) == 0u && sv == m) kp[j] = 1;
    });
    __syncthreads();
  }
  float* dst = out + (size_t)b * H8 * W8;
  for (int i = tid; i < TILE * TILE; i += NMS_THREADS) {
    const int y = i / TILE, x = i - y * TILE;
    const int gy = ty0 + HALO + y, gx = tx0 + HALO + x;
    if (gy < H8 && gx < W8) {
      const int j = (y + HALO) * TS + x + HALO;
      dst[(size_t)gy * W8 + gx] = kp[j] ? __uint_as_float(__float_as_uint(s[j]) & 0x7fffffffu) : 0.0f;   // (a kept maximum is near itself: clear the mark)
    }
  }
}

// ---------------------------------------------------------------------------
__device__ __forceinline__ bool sp_is_candidate(float v, int y, int x, int H8, int W8, float thr, int border) {
  return v > thr && y >= border && y < H8 - border && x >= border && x < W8 - border;
}

// one wave per score-map row: count candidates
__global__ __launch_bounds__(256) void count_rows_kernel(const float* __restrict__ nms, int* __restrict__ rowcount, int H8,
                                                         int W8, float thr, int border, const float* __restrict__ thr_dev) {
  const int lane = threadIdx.x & 63;
  const int y = blockIdx.x * 4 + (threadIdx.x >> 6), b = blockIdx.y;
  if (y >= H8) return;
  if (thr_dev) thr = thr_dev[b];
  const float* row = nms + ((size_t)b * H8 + y) * W8;
  int c = 0;
  if ((W8 & 3) == 0) {   // 16 bytes per lane, up to four requests in flight (one row of a 1024-wide map): the scalar loop ran at the load latency
    for (int x0 = 0; x0 < W8; x0 += 1024) {
      float4 v[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int x = x0 + 256 * i + 4 * lane;
        v[i] = x < W8 ? *(const float4*)(row + x) : make_float4(0.f, 0.f, 0.f, 0.f);
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int x = x0 + 256 * i + 4 * lane;
        if (x < W8)
          c += (sp_is_candidate(v[i].x, y, x, H8, W8, thr, border) ? 1 : 0) + (sp_is_candidate(v[i].y, y, x + 1, H8, W8, thr, border) ? 1 : 0) +
               (sp_is_candidate(v[i].z, y, x + 2, H8, W8, thr, border) ? 1 : 0) + (sp_is_candidate(v[i].w, y, x + 3, H8, W8, thr, border) ? 1 : 0);
      }
    }
  } else {
    for (int x = lane; x < W8; x += 64) c += sp_is_candidate(row[x], y, x, H8, W8, thr, border) ? 1 : 0;
  }
  c = wave_sum_i(c);
  if (lane == 0) rowcount[(size_t)b * H8 + y] = c;
}

// one block per image: exclusive scan over rows
__global__ __launch_bounds__(1024) void scan_rows_kernel(const int* __restrict__ rowcount, int* __restrict__ rowoff,
                                                         int* __restrict__ ncand, int H8) {
  __shared__ int part[1024];
  const int t = threadIdx.x, b = blockIdx.x;
  const int per = (H8 + 1023) / 1024;
  const int y0 = t * per, y1 = min(y0 + per, H8);
  int sum = 0;
  for (int y = y0; y < y1; ++y) sum += rowcount[(size_t)b * H8 + y];
  part[t] = sum;
  __syncthreads();
  for (int off = 1; off < 1024; off <<= 1) {
    int v = (t >= off) ? part[t - off] : 0;
    __syncthreads();
    part[t] += v;
    __syncthreads();
  }
  int run = part[t] - sum;
  for (int y = y0; y < y1; ++y) {
    rowoff[(size_t)b * H8 + y] = run;
    run += rowcount[(size_t)b * H8 + y];
  }
  if (t == 1023) ncand[b] = part[1023];
}

// one wave per row: ordered emission (row-major == torch.nonzero order, SPN:183-186)
__global__ __launch_bounds__(256) void emit_rows_kernel(const float* __restrict__ nms, const int* __restrict__ rowoff,
                                                        float* __restrict__ cand_score, int* __restrict__ cand_idx, int H8,
                                                        int W8, float thr, int border, const float* __restrict__ thr_dev) {
  const int lane = threadIdx.x & 63;
  const int y = blockIdx.x * 4 + (threadIdx.x >> 6), b = blockIdx.y;
  if (y >= H8) return;
  if (thr_dev) thr = thr_dev[b];
  const float* row = nms + ((size_t)b * H8 + y) * W8;
  const size_t base_img = (size_t)b * H8 * W8;
  int base = rowoff[(size_t)b * H8 + y];
  if ((W8 & 3) == 0) {   // the wide form of the loop below: lane l holds x .. x + 3 of every 256-column group; row-major order = lower lanes first, then the lane's own lower elements
    const unsigned long long lower = (1ull << lane) - 1ull;
    for (int x0 = 0; x0 < W8; x0 += 1024) {
      float4 v[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int x = x0 + 256 * i + 4 * lane;
        v[i] = x < W8 ? *(const float4*)(row + x) : make_float4(0.f, 0.f, 0.f, 0.f);
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int x = x0 + 256 * i + 4 * lane;
        if (x0 + 256 * i >= W8) break;   // wave-uniform
        const float e[4] = {v[i].x, v[i].y, v[i].z, v[i].w};
        bool c[4];
        unsigned long long m[4];
        int before = 0, total = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          c[j] = x < W8 && sp_is_candidate(e[j], y, x + j, H8, W8, thr, border);
          m[j] = __ballot(c[j]);
          before += __popcll(m[j] & lower);
          total += __popcll(m[j]);
        }
        int pos = base + before;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          if (c[j]) {
            cand_score[base_img + pos] = e[j];
            cand_idx[base_img + pos] = y * W8 + x + j;
            ++pos;
          }
        }
        base += total;
      }
    }
    return;
  }
  for (int x0 = 0; x0 < W8; x0 += 64) {
    const int x = x0 + lane;
    const float v = (x < W8) ? row[x] : 0.0f;
    const bool c = (x < W8) && sp_is_candidate(v, y, x, H8, W8, thr, border);
    const unsigned long long m = __ballot(c);
    if (c) {
      const int pos = base + __popcll(m & ((1ull << lane) - 1ull));
      cand_score[base_img + pos] = v;
      cand_idx[base_img + pos] = y * W8 + x;
    }
    base += __popcll(m);
  }
}

// ---------------------------------------------------------------------------
// top-k (SPN:74-78; DKD's n_limit cut ALN:169-173).  key = score bits (positive floats order like uints) << 32 |
// (~idx): descending key order = descending score, ascending index among ties; keys are unique (the index half), and a real
// key is never 0 (~idx != 0), so 0 pads.
//   k <= 4096 : ONE workgroup per image does everything in LDS (topk_kernel).
//   k  > 4096 : (round 6: config/aliked.yaml and config/superpoint+superglue.yaml ask for 8000, ALN:571 allows 20 000)
//               topk_select_big_kernel (the same radix select; the k keys >= the k-th go, unordered, to a global table padded to a
//               multiple of 4096), topk_chunk_sort_kernel (one workgroup per 4096-key chunk: bitonic sort in LDS) and topk_merge_kernel
//               (one thread per key: rank = own position + the number of larger keys in every other chunk, by binary search — unique
//               keys make the rank a permutation, so the output is the same whatever order the gather's atomics produced).
constexpr int TOPK_MAX = 4096;
constexpr int TOPK_BIG_MAX = 32768;

__device__ __forceinline__ unsigned long long topk_key(float score, int idx) {
  return ((unsigned long long)__float_as_uint(score) << 32) | (unsigned)(~idx);
}
__device__ __forceinline__ void topk_emit(unsigned long long key, int W8, float* kp, float* sc, int i) {
  const int idx = (int)(~(unsigned)(key & 0xffffffffull));
  kp[2 * i] = (float)(idx % W8);
  kp[2 * i + 1] = (float)(idx / W8);
  sc[i] = __uint_as_float((unsigned)(key >> 32));
}
// keep all, row-major order (SPN:75-76)
__device__ __forceinline__ void topk_keep_all(const float* cs, const int* ci, int n, int capacity, int W8, float* kp, float* sc, int* n_out) {
  const int t = threadIdx.x;
  const int m = min(n, capacity);
  for (int i = t; i < m; i += 1024) {
    const int idx = ci[i];
    kp[2 * i] = (float)(idx % W8);
    kp[2 * i + 1] = (float)(idx / W8);
    sc[i] = cs[i];
  }
  if (t == 0) *n_out = m;
}

// ---- radix select of the k-th largest key, one byte per pass from the top (1024 threads; returns the key, uniform) ----
// Round 5 (one image per call through the plugin hooks = ONE workgroup on the chip: 125 us of the call's 830): (a) four candidates per thread
// in flight per step instead of one dependent load per step; (b) the passes stop as soon as the selected bin is needed WHOLE (every key of it
// belongs to the top k: the remaining low bytes of the threshold are then 0) — with distinct scores that happens inside the score bytes, and
// the four index bytes (which only order ties of the k-th score) are never walked.  Same selection, same order.
struct TopkSelectShared {
  int hist[256];
  unsigned long long prefix;
  int krem, done;
};
__device__ __forceinline__ unsigned long long topk_radix_select(const float* __restrict__ cs, const int* __restrict__ ci, int n, int k,
                                                                TopkSelectShared& sh) {
  const int t = threadIdx.x;
  if (t == 0) { sh.prefix = 0ull; sh.krem = k; sh.done = 0; }
  __syncthreads();
  for (int byte = 7; byte >= 0; --byte) {
    if (t < 256) sh.hist[t] = 0;
    __syncthreads();
    const unsigned long long prefix = sh.prefix;
    const unsigned long long himask = (byte == 7) ? 0ull : (~0ull << (8 * (byte + 1)));
    // a thread's consecutive hits of one bin are merged into a single atomic: the leading bytes of positive float
    // scores are (nearly) constant, which would otherwise serialise every key of the image on one LDS counter
    int run_bin = -1, run_cnt = 0;
    for (int i0 = t; i0 < n; i0 += 4096) {
      float sv[4]; int iv[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int i = i0 + 1024 * u;
        sv[u] = i < n ? cs[i] : 0.0f;
        iv[u] = i < n ? ci[i] : 0;
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        if (i0 + 1024 * u >= n) break;
        const unsigned long long key = topk_key(sv[u], iv[u]);
        if ((key & himask) == prefix) {
          const int bin = (int)((key >> (8 * byte)) & 0xffull);
          if (bin == run_bin) {
            ++run_cnt;
          } else {
            if (run_cnt) atomicAdd(&sh.hist[run_bin], run_cnt);
            run_bin = bin; run_cnt = 1;
          }
        }
      }
    }
    if (run_cnt) atomicAdd(&sh.hist[run_bin], run_cnt);
    __syncthreads();
    if (t < 64) {  // wave 0: digit d with  above(d) < krem <= above(d) + hist[d],  above(d) = keys in higher bins
      const int krem = sh.krem;
      const int h0 = sh.hist[4 * t], h1 = sh.hist[4 * t + 1], h2 = sh.hist[4 * t + 2], h3 = sh.hist[4 * t + 3];
      const int tot = h0 + h1 + h2 + h3;
      int suf = tot;  // inclusive suffix sum over lanes t..63
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) {
        const int v = __shfl_down(suf, o);
        if (t + o < 64) suf += v;
      }
      const int a3 = suf - tot, a2 = a3 + h3, a1 = a2 + h2, a0 = a1 + h1;  // keys above bins 4t+3 .. 4t
      int d = -1, above = 0, hd = 0;
      if (a3 < krem && krem <= a3 + h3) { d = 4 * t + 3; above = a3; hd = h3; }
      else if (a2 < krem && krem <= a2 + h2) { d = 4 * t + 2; above = a2; hd = h2; }
      else if (a1 < krem && krem <= a1 + h1) { d = 4 * t + 1; above = a1; hd = h1; }
      else if (a0 < krem && krem <= a0 + h0) { d = 4 * t; above = a0; hd = h0; }
      if (d >= 0) {  // exactly one lane (the prefix always holds >= krem keys)
        sh.krem = krem - above;
        sh.prefix = prefix | ((unsigned long long)d << (8 * byte));
        sh.done = (krem - above == hd) ? 1 : 0;   // the whole bin is taken: the threshold's remaining bytes are 0
      }
    }
    __syncthreads();
    if (sh.done) break;   // (uniform)
  }
  return sh.prefix;
}
// the keys >= kth (exactly k of them) to dst[0 .. P) in the order the atomics fall; the caller zeroed dst and *cnt
__device__ __forceinline__ void topk_gather(const float* __restrict__ cs, const int* __restrict__ ci, int n, unsigned long long kth,
                                            unsigned long long* dst, int P, int* cnt) {
  const int t = threadIdx.x;
  for (int i0 = t; i0 < n; i0 += 4096) {
    float sv[4]; int iv[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int i = i0 + 1024 * u;
      sv[u] = i < n ? cs[i] : 0.0f;
      iv[u] = i < n ? ci[i] : 0;
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const unsigned long long key = topk_key(sv[u], iv[u]);
      if (i0 + 1024 * u < n && key >= kth) {
        const int pos = atomicAdd(cnt, 1);
        if (pos < P) dst[pos] = key;
      }
    }
  }
}
// bitonic sort, descending, of P (power of two) keys in LDS by 1024 threads
__device__ __forceinline__ void topk_bitonic_desc(unsigned long long* keys, int P) {
  const int t = threadIdx.x;
  for (int size = 2; size <= P; size <<= 1) {
    for (int stride = size >> 1; stride > 0; stride >>= 1) {
      for (int i = t; i < (P >> 1); i += 1024) {
        const int lo = 2 * i - (i & (stride - 1));  // index with bit `stride` clear
        const int hi = lo + stride;
        const bool desc = ((lo & size) == 0);
        const unsigned long long a = keys[lo], c = keys[hi];
        if (desc ? (a < c) : (a > c)) { keys[lo] = c; keys[hi] = a; }
      }
      __syncthreads();
    }
  }
}

// one block per image
__global__ __launch_bounds__(1024) void topk_kernel(const float* __restrict__ cand_score, const int* __restrict__ cand_idx,
                                                    const int* __restrict__ ncand, int H8, int W8, int k, int capacity,
                                                    int sort_always, float* __restrict__ kpts, float* __restrict__ scores,
                                                    int* __restrict__ n_out) {
  __shared__ unsigned long long keys[TOPK_MAX];
  __shared__ TopkSelectShared sel;
  __shared__ int sh_cnt;
  const int t = threadIdx.x, b = blockIdx.x;
  const int n = ncand[b];
  const float* cs = cand_score + (size_t)b * H8 * W8;
  const int* ci = cand_idx + (size_t)b * H8 * W8;
  float* kp = kpts + (size_t)b * capacity * 2;
  float* sc = scores + (size_t)b * capacity;

  const bool all = k < 0 || n <= k;
  if (all && !(sort_always && k > 0)) {
    topk_keep_all(cs, ci, n, capacity, W8, kp, sc, n_out + b);
    return;
  }
  // sort_always (DKD's top-k mode, ALN:150-151: torch.topk returns its values sorted): fewer candidates than k are ALL taken, score-descending
  const unsigned long long kth = all ? 0ull : topk_radix_select(cs, ci, n, k, sel);
  const int m = all ? n : k;

  // ---- gather the m keys >= kth, pad to a power of two, bitonic sort descending ----
  int P = 1;
  while (P < m) P <<= 1;
  for (int i = t; i < P; i += 1024) keys[i] = 0ull;
  if (t == 0) sh_cnt = 0;
  __syncthreads();
  topk_gather(cs, ci, n, kth, keys, P, &sh_cnt);
  __syncthreads();
  topk_bitonic_desc(keys, P);
  for (int i = t; i < m; i += 1024) topk_emit(keys[i], W8, kp, sc, i);
  if (t == 0) n_out[b] = m;
}

// ---- k > 4096: select + unordered gather into the global key table [batch][P], P = k rounded up to whole chunks ----
__global__ __launch_bounds__(1024) void topk_select_big_kernel(const float* __restrict__ cand_score, const int* __restrict__ cand_idx,
                                                               const int* __restrict__ ncand, int H8, int W8, int k, int capacity, int P,
                                                               int sort_always, unsigned long long* __restrict__ gkeys, float* __restrict__ kpts,
                                                               float* __restrict__ scores, int* __restrict__ n_out) {
  __shared__ TopkSelectShared sel;
  __shared__ int sh_cnt;
  const int t = threadIdx.x, b = blockIdx.x;
  const int n = ncand[b];
  const float* cs = cand_score + (size_t)b * H8 * W8;
  const int* ci = cand_idx + (size_t)b * H8 * W8;
  if (n <= k && !sort_always) {   // keep all; the sort / merge workgroups of this image see n <= k and leave
    topk_keep_all(cs, ci, n, capacity, W8, kpts + (size_t)b * capacity * 2, scores + (size_t)b * capacity, n_out + b);
    return;
  }
  const unsigned long long kth = n <= k ? 0ull : topk_radix_select(cs, ci, n, k, sel);
  unsigned long long* dst = gkeys + (size_t)b * P;
  for (int i = t; i < P; i += 1024) dst[i] = 0ull;
  if (t == 0) sh_cnt = 0;
  __syncthreads();
  topk_gather(cs, ci, n, kth, dst, P, &sh_cnt);
  if (t == 0) n_out[b] = min(n, k);
}
__global__ __launch_bounds__(1024) void topk_chunk_sort_kernel(const int* __restrict__ ncand, int k, int P, int sort_always, unsigned long long* __restrict__ gkeys) {
  __shared__ unsigned long long keys[TOPK_MAX];
  const int t = threadIdx.x, b = blockIdx.y;
  if (ncand[b] <= k && !sort_always) return;   // (uniform)
  unsigned long long* src = gkeys + (size_t)b * P + (size_t)blockIdx.x * TOPK_MAX;
  for (int i = t; i < TOPK_MAX; i += 1024) keys[i] = src[i];
  __syncthreads();
  topk_bitonic_desc(keys, TOPK_MAX);
  for (int i = t; i < TOPK_MAX; i += 1024) src[i] = keys[i];
}
__global__ __launch_bounds__(256) void topk_merge_kernel(const int* __restrict__ ncand, int W8, int k, int capacity, int P, int sort_always,
                                                         const unsigned long long* __restrict__ gkeys, float* __restrict__ kpts,
                                                         float* __restrict__ scores) {
  const int i = blockIdx.x * 256 + threadIdx.x, b = blockIdx.y;
  if ((ncand[b] <= k && !sort_always) || i >= P) return;
  const unsigned long long* g = gkeys + (size_t)b * P;
  const unsigned long long key = g[i];
  if (key == 0ull) return;   // padding
  const int own = i / TOPK_MAX;
  int rank = i - own * TOPK_MAX;   // keys of the own chunk in front of this one
  for (int c = 0; c < P / TOPK_MAX; ++c) {
    if (c == own) continue;
    const unsigned long long* ch = g + (size_t)c * TOPK_MAX;   // descending; the number of keys > key = first position with ch[pos] < key
    int lo = 0, hi = TOPK_MAX;
    while (lo < hi) {
      const int mid = (lo + hi) >> 1;
      if (ch[mid] > key) lo = mid + 1; else hi = mid;
    }
    rank += lo;
  }
  if (rank < k) topk_emit(key, W8, kpts + (size_t)b * capacity * 2, scores + (size_t)b * capacity, rank);
}

// DKD's top-k mode with FEWER maxima than k (ALN:150-151): torch.topk over the border-cleared NMS map then fills up with zero-score pixels.
// Which ones is an artefact of the sort it runs (libstdc++'s heap / introselect on the reference's CPU path, a radix select on its CUDA path:
// different pixels) — here the first k - n non-candidate pixels in row-major order, appended behind the n sorted maxima with score 0.
// One workgroup per image; 1024 pixels per step, wave ballots + a 16-entry LDS scan give every pixel its slot.
__global__ __launch_bounds__(1024) void topk_zero_fill_kernel(const float* __restrict__ nms, int H8, int W8, float thr, int border, int k,
                                                              int capacity, float* __restrict__ kpts, float* __restrict__ scores,
                                                              int* __restrict__ n_out) {
  __shared__ int wave_cnt[16];
  __shared__ int sh_base;
  const int t = threadIdx.x, lane = t & 63, wv = t >> 6, b = blockIdx.x;
  const int n = n_out[b];
  if (n >= k) return;   // (uniform)
  const float* src = nms + (size_t)b * H8 * W8;
  float* kp = kpts + (size_t)b * capacity * 2;
  float* sc = scores + (size_t)b * capacity;
  const int total = H8 * W8;
  if (t == 0) sh_base = n;
  __syncthreads();
  for (int i0 = 0; i0 < total; i0 += 1024) {
    const int base = sh_base;
    if (base >= k) break;   // (uniform)
    const int i = i0 + t;
    const int y = i / W8, x = i - y * W8;
    const bool fill = i < total && !sp_is_candidate(src[min(i, total - 1)], y, x, H8, W8, thr, border);
    const unsigned long long m = __ballot(fill);
    if (lane == 0) wave_cnt[wv] = __popcll(m);
    __syncthreads();
    int before = 0, all = 0;
#pragma unroll
    for (int w = 0; w < 16; ++w) {
      const int c = wave_cnt[w];
      before += w < wv ? c : 0;
      all += c;
    }
    const int pos = base + before + __popcll(m & ((1ull << lane) - 1ull));
    if (fill && pos < k) {
      kp[2 * pos] = (float)x;
      kp[2 * pos + 1] = (float)y;
      sc[pos] = 0.0f;
    }
    __syncthreads();
    if (t == 0) sh_base = base + all;
    __syncthreads();
  }
  if (t == 0) n_out[b] = min(sh_base, k);
}

// ---------------------------------------------------------------------------
// one wave per keypoint; lane owns channels 4*lane .. 4*lane+3 of the 256.
__global__ __launch_bounds__(256) void sample_desc_kernel(const float* __restrict__ dense, const float* __restrict__ kpts,
                                                          const int* __restrict__ n_kpts, float* __restrict__ desc, int h,
                                                          int w, int capacity, int fix_sampling) {
  const int lane = threadIdx.x & 63;
  const int i = blockIdx.x * 4 + (threadIdx.x >> 6), b = blockIdx.y;
  if (i >= n_kpts[b]) return;
  const float kx = kpts[((size_t)b * capacity + i) * 2], ky = kpts[((size_t)b * capacity + i) * 2 + 1];
  float ix, iy;
  if (fix_sampling) {  // SPX:16-27: (k + 0.5) / (w*8), align_corners=False
    float gx = (kx + 0.5f) / ((float)w * 8.0f), gy = (ky + 0.5f) / ((float)h * 8.0f);
    gx = gx * 2.0f - 1.0f; gy = gy * 2.0f - 1.0f;
    ix = ((gx + 1.0f) * (float)w - 1.0f) / 2.0f;
    iy = ((gy + 1.0f) * (float)h - 1.0f) / 2.0f;
  } else {  // SPN:84-94: (k - 4 + 0.5) / (w*8 - 4 - 0.5), align_corners=True
    float gx = (kx - 4.0f + 0.5f) / ((float)(w * 8) - 4.0f - 0.5f), gy = (ky - 4.0f + 0.5f) / ((float)(h * 8) - 4.0f - 0.5f);
    gx = gx * 2.0f - 1.0f; gy = gy * 2.0f - 1.0f;
    ix = ((gx + 1.0f) / 2.0f) * (float)(w - 1);
    iy = ((gy + 1.0f) / 2.0f) * (float)(h - 1);
  }
  const float fx = floorf(ix), fy = floorf(iy);
  const int x0 = (int)fx, y0 = (int)fy;
  // bilinear weights exactly as ATen's grid_sampler: nw, ne, sw, se
  const float wnw = (fx + 1.0f - ix) * (fy + 1.0f - iy), wne = (ix - fx) * (fy + 1.0f - iy);
  const float wsw = (fx + 1.0f - ix) * (iy - fy), wse = (ix - fx) * (iy - fy);
  const float* base = dense + (size_t)b * h * w * 256;
  float acc[4] = {0.f, 0.f, 0.f, 0.f};
  const int cxs[4] = {x0, x0 + 1, x0, x0 + 1};
  const int cys[4] = {y0, y0, y0 + 1, y0 + 1};
  const float wts[4] = {wnw, wne, wsw, wse};
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    const bool in = cxs[c] >= 0 && cxs[c] < w && cys[c] >= 0 && cys[c] < h;  // wave-uniform
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (in) v = *(const float4*)(base + ((size_t)cys[c] * w + cxs[c]) * 256 + lane * 4);
    float ss = wave_sum(v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w);
    const float den = fmaxf(sqrtf(ss), 1e-12f);  // F.normalize of the dense map (SPN:215)
    if (in) {
      acc[0] += (v.x / den) * wts[c]; acc[1] += (v.y / den) * wts[c];
      acc[2] += (v.z / den) * wts[c]; acc[3] += (v.w / den) * wts[c];
    }
  }
  const float ss = wave_sum(acc[0] * acc[0] + acc[1] * acc[1] + acc[2] * acc[2] + acc[3] * acc[3]);
  const float den = fmaxf(sqrtf(ss), 1e-12f);  // SPN:95-97
  *(float4*)(desc + ((size_t)b * capacity + i) * 256 + lane * 4) =
      make_float4(acc[0] / den, acc[1] / den, acc[2] / den, acc[3] / den);
}

}  // namespace

int launch_softmax_d2s(const float* logits, float* smap, int batch, int h, int w, hipStream_t s) {
  const int n = batch * h * w;
  if (n <= 0) return 0;
  hipLaunchKernelGGL(softmax_d2s_kernel, dim3(cdiv(n, 4 * SD_CPW)), dim3(256), 0, s, logits, smap, n, h, w);
  DIM_LAUNCH_CHECK();
  return 0;
}

static int g_nms_big = 1;
int dim_nms_big_tiles() { return g_nms_big; }
void dim_nms_set_big_tiles(int v) { g_nms_big = v; }

int launch_nms(const float* smap, float* out, int batch, int H8, int W8, int radius, hipStream_t s) {
  DIM_REQUIRE(radius >= 0 && radius <= 6, "nms: radius %d unsupported (0..6)", radius);
  if (batch <= 0 || H8 <= 0 || W8 <= 0) return 0;
#define DIM_NMS(RR, TL, TH, SG)                                                                                      \
  {                                                                                                                  \
    const int tx = cdiv(W8, TL), ty = cdiv(H8, TL);                                                                  \
    hipLaunchKernelGGL(HIP_KERNEL_NAME(nms_kernel<RR, TL, TH, SG>), dim3(tx * ty, 1, batch), dim3(TH), 0, s, smap, out, H8, W8, tx); \
  }
  // large maps, radius <= 4: 64 x 64 tiles (less halo recomputation); small maps keep 32 x 32 (more workgroups than CUs)
  const bool big = dim_nms_big_tiles() == 2 || (dim_nms_big_tiles() && (long)cdiv(W8, 64) * cdiv(H8, 64) * batch >= 256);  // 2: forced (tests)
  switch (radius) {
    case 0: DIM_NMS(0, 32, 512, 8) break;
    case 1: if (big) DIM_NMS(1, 64, 1024, 8) else DIM_NMS(1, 32, 512, 8) break;
    case 2: DIM_NMS(2, 32, 512, 8) break;  // (round 2, 14-byte elements: 4 workgroups of 38 KB per CU beat one 64-tile workgroup of 100 KB: 2.5 vs 3.1 us per 512^2 map)
    case 3: if (big) DIM_NMS(3, 64, 1024, 10) else DIM_NMS(3, 32, 512, 8) break;
    case 4: if (big) DIM_NMS(4, 64, 1024, 12) else DIM_NMS(4, 32, 512, 8) break;
    case 5: DIM_NMS(5, 16, 512, 8) break;
    default: DIM_NMS(6, 16, 512, 8) break;
  }
#undef DIM_NMS
  DIM_LAUNCH_CHECK();
  return 0;
}

int launch_select_ex(const float* nms, int batch, int H8, int W8, float thr, const float* thr_dev, int border, int* rowcount,
                     int* rowoff, int* ncand, float* cand_score, int* cand_idx, int count_only, hipStream_t s) {
  if (batch <= 0 || H8 <= 0 || W8 <= 0) return 0;
  hipLaunchKernelGGL(count_rows_kernel, dim3(cdiv(H8, 4), batch), dim3(256), 0, s, nms, rowcount, H8, W8, thr, border, thr_dev);
  hipLaunchKernelGGL(scan_rows_kernel, dim3(batch), dim3(1024), 0, s, (const int*)rowcount, rowoff, ncand, H8);
  if (!count_only)
    hipLaunchKernelGGL(emit_rows_kernel, dim3(cdiv(H8, 4), batch), dim3(256), 0, s, nms, (const int*)rowoff, cand_score, cand_idx, H8, W8, thr, border, thr_dev);
  DIM_LAUNCH_CHECK();
  return 0;
}
int launch_select(const float* nms, int batch, int H8, int W8, float thr, int border, int* rowcount, int* rowoff,
                  int* ncand, float* cand_score, int* cand_idx, hipStream_t s) {
  return launch_select_ex(nms, batch, H8, W8, thr, nullptr, border, rowcount, rowoff, ncand, cand_score, cand_idx, 0, s);
}

size_t topk_scratch_keys(int batch, int k) {   // 8-byte keys of launch_topk's global table (0: the one-workgroup form needs none)
  if (k <= TOPK_MAX) return 0;
  return (size_t)batch * (size_t)(cdiv(k, TOPK_MAX) * TOPK_MAX);
}
int launch_topk(const float* cand_score, const int* cand_idx, const int* ncand, int batch, int H8, int W8, int k,
                int capacity, float* kpts, float* scores, int* n_out, unsigned long long* scratch, int sort_always, hipStream_t s) {
  DIM_REQUIRE(k <= TOPK_BIG_MAX, "topk: max_keypoints %d > %d unsupported", k, TOPK_BIG_MAX);
  if (batch <= 0) return 0;
  if (k <= TOPK_MAX) {
    hipLaunchKernelGGL(topk_kernel, dim3(batch), dim3(1024), 0, s, cand_score, cand_idx, ncand, H8, W8, k, capacity, sort_always, kpts, scores, n_out);
  } else {
    DIM_REQUIRE(scratch != nullptr, "topk: max_keypoints %d needs the key table (topk_scratch_keys)", k);
    DIM_REQUIRE(k <= capacity, "topk: max_keypoints %d > capacity %d", k, capacity);
    const int P = cdiv(k, TOPK_MAX) * TOPK_MAX;
    hipLaunchKernelGGL(topk_select_big_kernel, dim3(batch), dim3(1024), 0, s, cand_score, cand_idx, ncand, H8, W8, k, capacity, P, sort_always, scratch, kpts, scores, n_out);
    hipLaunchKernelGGL(topk_chunk_sort_kernel, dim3(P / TOPK_MAX, batch), dim3(1024), 0, s, ncand, k, P, sort_always, scratch);
    hipLaunchKernelGGL(topk_merge_kernel, dim3(P / 256, batch), dim3(256), 0, s, ncand, W8, k, capacity, P, sort_always, (const unsigned long long*)scratch, kpts, scores);
  }
  DIM_LAUNCH_CHECK();
  return 0;
}
int launch_topk_zero_fill(const float* nms, int batch, int H8, int W8, float thr, int border, int k, int capacity, float* kpts, float* scores,
                          int* n_out, hipStream_t s) {
  DIM_REQUIRE(k <= capacity, "topk fill: k %d > capacity %d", k, capacity);
  DIM_REQUIRE((long long)k <= (long long)H8 * W8, "selected index k out of range: top_k %d > %d x %d pixels", k, H8, W8);   // torch.topk's error (ALN:150)
  if (batch <= 0) return 0;
  hipLaunchKernelGGL(topk_zero_fill_kernel, dim3(batch), dim3(1024), 0, s, nms, H8, W8, thr, border, k, capacity, kpts, scores, n_out);
  DIM_LAUNCH_CHECK();
  return 0;
}

int launch_sample_desc(const float* dense, const float* kpts, const int* n_kpts, float* desc, int batch, int h, int w,
                       int capacity, int fix_sampling, hipStream_t s) {
  if (batch <= 0 || capacity <= 0) return 0;
  hipLaunchKernelGGL(sample_desc_kernel, dim3(cdiv(capacity, 4), batch), dim3(256), 0, s, dense, kpts, n_kpts, desc, h, w, capacity, fix_sampling);
  DIM_LAUNCH_CHECK();
  return 0;
}
