// ALIKED kernels for gfx950 (reference ALN = thirdparty/LightGlue/lightglue/aliked.py).
//
// ALIKED's dense stage is HBM-bound (≈10 kMAC/px against a 512 B/px feature map), its
// channel counts are small (3..128), and BatchNorm runs in TRAINING mode (Q7: statistics of the
// current image), so every conv output needs a global per-channel reduction before it can be
// activated.  Layout NHWC; convolutions are direct VALU kernels whose weights are wave-uniform and
// therefore come through the scalar cache (s_load + v_fmac with an SGPR operand), inputs through an
// LDS halo tile with a (C+4)-dword pixel stride (conflict-free ds_read_b128); MFMA is used only for
// the SDDH GEMMs (gemm.hip).
#include <math.h>

#include "aliked_kernels.h"

namespace {

__device__ __forceinline__ float selu_(float x) {
  // ATen elu kernel: x <= 0 ? (exp(x) - 1) * (alpha*scale) : x * scale
  const float scale = 1.0507009873554804934193349852946f, alpha = 1.6732632423543772848170429916717f;
  return x <= 0.0f ? (exp_le0(x) - 1.0f) * (alpha * scale) : x * scale;
}
__device__ __forceinline__ float act_(float v, int act) {
  if (act == AL_ACT_SELU) return selu_(v);
  if (act == AL_ACT_SIGMOID) return 1.0f / (1.0f + expf(-v));
  return v;
}

// ---------------------------------------------------------------------------
// direct 3x3 conv, 16x16 pixel tile per workgroup, one pixel per thread, all COUT per thread;
// input channels are walked in chunks of CC through an LDS halo tile.
template <int CC, int COUT>
__global__ __launch_bounds__(256) void al_conv3x3_kernel(const float* __restrict__ in, int in_c, int cin_pad,
                                                         const float* __restrict__ w, const float* __restrict__ bias,
                                                         float* __restrict__ out, int out_c, int H, int W, int act, int tiles_x,
                                                         int crop_y, int crop_x, int out_h, int out_w) {
  constexpr int PS = CC + 4;  // pixel stride in dwords
  __shared__ float tile[18 * 18 * PS];
  const int t = threadIdx.x, b = blockIdx.z;
  const int ty0 = (blockIdx.x / tiles_x) * 16, tx0 = (blockIdx.x % tiles_x) * 16;
  const float* src = in + (size_t)b * H * W * in_c;
  const int py = t >> 4, px = t & 15;
  float acc[COUT];
#pragma unroll
  for (int co = 0; co < COUT; ++co) acc[co] = 0.f;
  for (int c0 = 0; c0 < cin_pad; c0 += CC) {
    if (c0 > 0) __syncthreads();
    if (in_c % 4 == 0) {
      for (int i = t; i < 18 * 18 * (CC / 4); i += 256) {
        const int p = i / (CC / 4), q = i % (CC / 4);
        const int gy = ty0 + p / 18 - 1, gx = tx0 + p % 18 - 1;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (gy >= 0 && gy < H && gx >= 0 && gx < W) v = *(const float4*)(src + ((size_t)gy * W + gx) * in_c + c0 + q * 4);
        *(float4*)&tile[p * PS + q * 4] = v;
      }
    } else {
      for (int i = t; i < 18 * 18 * CC; i += 256) {
        const int p = i / CC, c = i % CC;
        const int gy = ty0 + p / 18 - 1, gx = tx0 + p % 18 - 1;
        float v = 0.f;
        if (c < in_c && gy >= 0 && gy < H && gx >= 0 && gx < W) v = src[((size_t)gy * W + gx) * in_c + c];
        tile[p * PS + c] = v;
      }
    }
    __syncthreads();
    for (int tap = 0; tap < 9; ++tap) {
      const float* tp = &tile[((py + tap / 3) * 18 + px + tap % 3) * PS];
      const float* wt = w + ((size_t)tap * cin_pad + c0) * COUT;
#pragma unroll
      for (int c4 = 0; c4 < CC / 4; ++c4) {
        const float4 a = *(const float4*)(tp + c4 * 4);
        const float av[4] = {a.x, a.y, a.z, a.w};
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int co = 0; co < COUT; ++co) acc[co] = fmaf(av[j], wt[(c4 * 4 + j) * COUT + co], acc[co]);
      }
    }
  }
  const int y = ty0 + py - crop_y, x = tx0 + px - crop_x;
  if (ty0 + py < H && tx0 + px < W && y >= 0 && y < out_h && x >= 0 && x < out_w) {
    float* dst = out + (((size_t)b * out_h + y) * out_w + x) * out_c;
#pragma unroll
    for (int co = 0; co < COUT; ++co)
      if (co < out_c) dst[co] = act_(acc[co] + (bias ? bias[co] : 0.f), act);
  }
}

// 1x1 conv: one pixel per thread, weights [CIN][COUT] through the scalar cache.
template <int CIN, int COUT>
__global__ __launch_bounds__(256) void al_conv1x1_kernel(const float* __restrict__ in, const float* __restrict__ w,
                                                         const float* __restrict__ bias, float* __restrict__ out, int n, int act) {
  const int p = blockIdx.x * 256 + threadIdx.x;
  if (p >= n) return;
  float acc[COUT];
#pragma unroll
  for (int co = 0; co < COUT; ++co) acc[co] = bias ? bias[co] : 0.f;
  const float* src = in + (size_t)p * CIN;
#pragma unroll
  for (int c4 = 0; c4 < CIN / 4; ++c4) {
    const float4 a = *(const float4*)(src + c4 * 4);
    const float av[4] = {a.x, a.y, a.z, a.w};
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int co = 0; co < COUT; ++co) acc[co] = fmaf(av[j], w[(c4 * 4 + j) * COUT + co], acc[co]);
  }
  float* dst = out + (size_t)p * COUT;
#pragma unroll
  for (int co = 0; co < COUT; co += 4)
    *(float4*)(dst + co) = make_float4(act_(acc[co], act), act_(acc[co + 1], act), act_(acc[co + 2], act), act_(acc[co + 3], act));
}

// replicate padding to a multiple of 32 (InputPadder, ALN:247-271); gray -> RGB repeat (ALN:679-680)
__global__ __launch_bounds__(256) void al_pad_kernel(const float* __restrict__ img, float* __restrict__ out, int H, int W, int Hp,
                                                     int Wp, int pad_t, int pad_l, int in_ch) {
  const int i = blockIdx.x * 256 + threadIdx.x, b = blockIdx.y;
  if (i >= Hp * Wp) return;
  const int y = i / Wp, x = i - y * Wp;
  const int sy = min(max(y - pad_t, 0), H - 1), sx = min(max(x - pad_l, 0), W - 1);
  const float* s = img + (((size_t)b * H + sy) * W + sx) * in_ch;
  float* d = out + ((size_t)b * Hp * Wp + i) * 3;
  d[0] = s[0]; d[1] = in_ch == 3 ? s[1] : s[0]; d[2] = in_ch == 3 ? s[2] : s[0];
}

__global__ __launch_bounds__(256) void al_avgpool_kernel(const float* __restrict__ in, float* __restrict__ out, int H, int W, int C,
                                                         int k) {
  const int Ho = H / k, Wo = W / k, C4 = C / 4;
  const int i = blockIdx.x * 256 + threadIdx.x, b = blockIdx.y;
  if (i >= Ho * Wo * C4) return;
  const int c4 = i % C4, p = i / C4, y = p / Wo, x = p - y * Wo;
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int dy = 0; dy < k; ++dy)
    for (int dx = 0; dx < k; ++dx) {
      const float4 v = *(const float4*)(in + (((size_t)b * H + y * k + dy) * W + x * k + dx) * C + c4 * 4);
      s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
  const float inv = (float)(k * k);
  *(float4*)(out + (((size_t)b * Ho + y) * Wo + x) * C + c4 * 4) = make_float4(s.x / inv, s.y / inv, s.z / inv, s.w / inv);
}

// ---------------------------------------------------------------------------
// BN training-mode statistics: stage 1 = per-block fp64 partial (sum, sumsq) per channel, stage 2 =
// fixed-order reduction -> alpha = gamma/sqrt(var+eps), beta = bias - mean*alpha (ATen's contiguous
// CPU path: out = x*alpha + beta).  Deterministic (no atomics).
constexpr int BN_BLOCKS = 256;
__global__ __launch_bounds__(256) void al_bn_partial_kernel(const float* __restrict__ x, int n_pixels, int C, double* __restrict__ partial) {
  __shared__ double red[256][2];
  const int b = blockIdx.y, blk = blockIdx.x, t = threadIdx.x;
  const int lanes_per_c = 256 / C;  // C in {8,16,32,64,128} -> 32,16,8,4,2 pixel lanes per channel
  const int c = t % C, pl = t / C;
  const float* src = x + (size_t)b * n_pixels * C;
  double s = 0.0, q = 0.0;
  for (int p = blk * lanes_per_c + pl; p < n_pixels; p += BN_BLOCKS * lanes_per_c) {
    const double v = (double)src[(size_t)p * C + c];
    s += v; q += v * v;
  }
  red[t][0] = s; red[t][1] = q;
  __syncthreads();
  if (t < C) {
    for (int k = 1; k < lanes_per_c; ++k) { s += red[t + k * C][0]; q += red[t + k * C][1]; }
    double* d = partial + (((size_t)b * BN_BLOCKS + blk) * C + t) * 2;
    d[0] = s; d[1] = q;
  }
}
__global__ void al_bn_final_kernel(const double* __restrict__ partial, int n_pixels, int C, const float* __restrict__ gamma,
                                   const float* __restrict__ beta_w, float* __restrict__ alpha, float* __restrict__ beta) {
  const int c = threadIdx.x, b = blockIdx.x;
  if (c >= C) return;
  double s = 0.0, q = 0.0;
  for (int k = 0; k < BN_BLOCKS; ++k) {
    const double* d = partial + (((size_t)b * BN_BLOCKS + k) * C + c) * 2;
    s += d[0]; q += d[1];
  }
  const double mean = s / n_pixels;
  double var = q / n_pixels - mean * mean;
  if (var < 0.0) var = 0.0;
  const float invstd = (float)(1.0 / sqrt(var + 1e-5));
  const float a = invstd * gamma[c];
  alpha[b * C + c] = a;
  beta[b * C + c] = beta_w[c] - (float)mean * a;
}
// y = selu(x*alpha + beta (+ residual))
__global__ __launch_bounds__(256) void al_bn_apply_kernel(const float* __restrict__ x, const float* __restrict__ alpha,
                                                          const float* __restrict__ beta, const float* __restrict__ res,
                                                          float* __restrict__ out, int n_pixels, int C) {
  const int C4 = C / 4;
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  const int b = blockIdx.y;
  if (i >= (size_t)n_pixels * C4) return;
  const int c = (int)(i % C4) * 4;
  const size_t off = (size_t)b * n_pixels * C + i * 4;
  const float4 v = *(const float4*)(x + off);
  const float4 a = *(const float4*)(alpha + b * C + c), bb = *(const float4*)(beta + b * C + c);
  float4 y = make_float4(v.x * a.x + bb.x, v.y * a.y + bb.y, v.z * a.z + bb.z, v.w * a.w + bb.w);
  if (res) {
    const float4 r = *(const float4*)(res + off);
    y.x += r.x; y.y += r.y; y.z += r.z; y.w += r.w;
  }
  *(float4*)(out + off) = make_float4(selu_(y.x), selu_(y.y), selu_(y.z), selu_(y.w));
}

// BatchNorm-apply (+ residual) + SELU and the k x k average pooling that follows it in ONE pass over the raw map: thread =
// (pooled pixel, 4 channels) reads its k x k window once, writes the k x k activated values (the block's output x_i, needed
// at full resolution by the feature aggregation) and their average (the next block's input) — the separate pooling pass
// re-read the whole activated map.  Same arithmetic and summation order as al_bn_apply_kernel followed by al_avgpool_kernel.
template <int K>
__global__ __launch_bounds__(256) void al_bn_apply_pool_kernel(const float* __restrict__ x, const float* __restrict__ alpha,
                                                               const float* __restrict__ beta, const float* __restrict__ res,
                                                               float* __restrict__ out, float* __restrict__ pooled, int H, int W, int C) {
  const int Ho = H / K, Wo = W / K, C4 = C / 4;
  const int i = blockIdx.x * 256 + threadIdx.x, b = blockIdx.y;
  if (i >= Ho * Wo * C4) return;
  const int c4 = i % C4, p = i / C4, y = p / Wo, xx = p - y * Wo;
  const float4 a = *(const float4*)(alpha + b * C + c4 * 4), bb = *(const float4*)(beta + b * C + c4 * 4);
  float4 v[K * K], r[K * K];
#pragma unroll
  for (int d = 0; d < K * K; ++d) {
    const size_t off = (((size_t)b * H + y * K + d / K) * W + xx * K + d % K) * C + c4 * 4;
    v[d] = *(const float4*)(x + off);
    r[d] = res ? *(const float4*)(res + off) : make_float4(0.f, 0.f, 0.f, 0.f);
  }
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
  for (int d = 0; d < K * K; ++d) {
    const size_t off = (((size_t)b * H + y * K + d / K) * W + xx * K + d % K) * C + c4 * 4;
    float4 t = make_float4(v[d].x * a.x + bb.x, v[d].y * a.y + bb.y, v[d].z * a.z + bb.z, v[d].w * a.w + bb.w);
    if (res) { t.x += r[d].x; t.y += r[d].y; t.z += r[d].z; t.w += r[d].w; }
    t = make_float4(selu_(t.x), selu_(t.y), selu_(t.z), selu_(t.w));
    *(float4*)(out + off) = t;
    s.x += t.x; s.y += t.y; s.z += t.z; s.w += t.w;
  }
  const float inv = (float)(K * K);
  *(float4*)(pooled + (((size_t)b * Ho + y) * Wo + xx) * C + c4 * 4) = make_float4(s.x / inv, s.y / inv, s.z / inv, s.w / inv);
}

__global__ __launch_bounds__(256) void al_clamp_kernel(float* __restrict__ x, size_t n, float lim) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) x[i] = fminf(fmaxf(x[i], -lim), lim);
}

// ---------------------------------------------------------------------------
// deformable conv (torchvision.ops.deform_conv2d semantics, call site ALN:322-329) = bilinear gather +
// GEMM: this kernel forms cols[pixel][tap * CIN + c] (the deformed im2col row, K = 9 * CIN) and the
// product with the [9 * CIN][cout] weight runs on the matrix cores (gemm.hip).  Thread = (pixel, tap,
// 4 channels); a wave covers consecutive channels of one tap so every corner read is one contiguous run.
// KROW = row stride of cols: 9 * CIN rounded up to the GEMMs' 32-wide K granule (CIN = 16, aliked-t16's block3: 144 -> 160, the tail zeroed here —
// the buffer is shared by layers of different K).
template <int CIN>
__global__ __launch_bounds__(256) void al_deform_gather_kernel(const float* __restrict__ in, const float* __restrict__ offs, int off_c,
                                                               float* __restrict__ cols, int H, int W, int n_rows) {
  constexpr int C4 = CIN / 4, KROW = (9 * CIN + 31) / 32 * 32;
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= (size_t)n_rows * 9 * C4) return;
  const int c4 = (int)(i % C4);
  const size_t r = i / C4;
  const int tap = (int)(r % 9);
  const size_t gp = r / 9;                  // b * H * W + p
  const int b = (int)(gp / ((size_t)H * W)), p = (int)(gp - (size_t)b * H * W), y = p / W, x = p - y * W;
  const float* src = in + (size_t)b * H * W * CIN + c4 * 4;
  const float* of = offs + gp * off_c;
  const float py = (float)(y - 1 + tap / 3) + of[2 * tap], px = (float)(x - 1 + tap % 3) + of[2 * tap + 1];
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
  if (py > -1.f && py < (float)H && px > -1.f && px < (float)W) {
    const float fy = floorf(py), fx = floorf(px);
    const int y0 = (int)fy, x0 = (int)fx, y1 = y0 + 1, x1 = x0 + 1;
    const float ly = py - fy, lx = px - fx, hy = 1.f - ly, hx = 1.f - lx;
    const float w00 = hy * hx, w01 = hy * lx, w10 = ly * hx, w11 = ly * lx;
    if (y0 >= 0 && x0 >= 0) { const float4 v = *(const float4*)(src + ((size_t)y0 * W + x0) * CIN); s.x += w00 * v.x; s.y += w00 * v.y; s.z += w00 * v.z; s.w += w00 * v.w; }
    if (y0 >= 0 && x1 <= W - 1) { const float4 v = *(const float4*)(src + ((size_t)y0 * W + x1) * CIN); s.x += w01 * v.x; s.y += w01 * v.y; s.z += w01 * v.z; s.w += w01 * v.w; }
    if (y1 <= H - 1 && x0 >= 0) { const float4 v = *(const float4*)(src + ((size_t)y1 * W + x0) * CIN); s.x += w10 * v.x; s.y += w10 * v.y; s.z += w10 * v.z; s.w += w10 * v.w; }
    if (y1 <= H - 1 && x1 <= W - 1) { const float4 v = *(const float4*)(src + ((size_t)y1 * W + x1) * CIN); s.x += w11 * v.x; s.y += w11 * v.y; s.z += w11 * v.z; s.w += w11 * v.w; }
  }
  *(float4*)(cols + gp * KROW + tap * CIN + c4 * 4) = s;
  if (KROW > 9 * CIN && tap == 8) {   // (KROW - 9 CIN) / 4 <= C4 float4s of zero padding, written by the last tap's threads
    if (c4 < (KROW - 9 * CIN) / 4) *(float4*)(cols + gp * KROW + 9 * CIN + c4 * 4) = make_float4(0.f, 0.f, 0.f, 0.f);
  }
}

// ---------------------------------------------------------------------------
// bilinear upsampling with align_corners=True as ATen computes it (upsample_bilinear2d, fp32):
// src = dst * (in-1)/(out-1); lambdas from the float source index.
struct UpIdx { int i0, i1; float l0, l1; };
__device__ __forceinline__ UpIdx up_index(int dst, int in_size, int out_size) {
  const float scale = out_size > 1 ? (float)(in_size - 1) / (float)(out_size - 1) : 0.f;
  const float r = scale * (float)dst;
  UpIdx u;
  u.i0 = (int)r;
  u.i1 = u.i0 + ((u.i0 < in_size - 1) ? 1 : 0);
  u.l1 = r - (float)u.i0;
  u.l0 = 1.f - u.l1;
  return u;
}
// x1234 = cat[selu(conv1(x1)), up2(f2), up8(f3), up32(f4)] (ALN:657-664) and s8 = selu(score_head.0(x1234))
// (ALN:666).  Workgroup = 64 pixels x 4 channel groups: lane (pixel, g) produces the 32 channels of group g
// (g = 0: the 16->32 1x1 conv of x1; g = 1..3: one bilinearly up-sampled map; one wave per group) and its
// share of the 128->8 score projection (reduced over the 4 waves through LDS); the 64 x 128 tile goes through LDS so
// that the 512 B/pixel feature map is written as one contiguous 32-KiB run.  The product path runs
// STORE = false: only s8 is written; the 128-channel map (512 MB per 1024^2 image) is never materialised,
// SDDH re-evaluates it at the cells it touches (feat_pair below).  STORE = true serves the debug tap.
// C1 = channels of x1, G = dim / 4 channels per group: (16, 32) for aliked-n16 / n16rot / n32, (8, 16) for aliked-t16
template <bool STORE, int C1, int G>
__global__ __launch_bounds__(256) void al_assemble_kernel(const float* __restrict__ x1, const float* __restrict__ f2,
                                                          const float* __restrict__ f3, const float* __restrict__ f4,
                                                          const float* __restrict__ w1, const float* __restrict__ ws0,
                                                          float* __restrict__ x1234, float* __restrict__ s8, int Hp, int Wp) {
  __shared__ float tile[STORE ? 64 * (4 * G + 4) : 4];
  __shared__ float spart[3][8][64];
  const int t = threadIdx.x, g = __builtin_amdgcn_readfirstlane(t >> 6), pl = t & 63, b = blockIdx.y;  // g is wave-uniform: weights come through scalar loads
  const int i = blockIdx.x * 64 + pl;
  const bool ok = i < Hp * Wp;
  const int y = ok ? i / Wp : 0, x = ok ? i - y * Wp : 0;
  float sacc[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) sacc[k] = 0.f;
  float* trow = &tile[STORE ? pl * (4 * G + 4) + g * G : 0];
  if (g == 0) {
    const float* src = x1 + ((size_t)b * Hp * Wp + (ok ? i : 0)) * C1;
    float a[C1];
#pragma unroll
    for (int c4 = 0; c4 < C1 / 4; ++c4) { const float4 v = *(const float4*)(src + c4 * 4); a[c4 * 4] = v.x; a[c4 * 4 + 1] = v.y; a[c4 * 4 + 2] = v.z; a[c4 * 4 + 3] = v.w; }
#pragma unroll
    for (int co = 0; co < G; co += 4) {
      float o[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ci = 0; ci < C1; ++ci)
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = fmaf(a[ci], w1[ci * G + co + j], o[j]);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        o[j] = selu_(o[j]);
#pragma unroll
        for (int k = 0; k < 8; ++k) sacc[k] = fmaf(o[j], ws0[(co + j) * 8 + k], sacc[k]);
      }
      if (STORE) *(float4*)(trow + co) = make_float4(o[0], o[1], o[2], o[3]);
    }
  } else {
    const int fac = g == 1 ? 2 : (g == 2 ? 8 : 32);
    const float* map = g == 1 ? f2 : (g == 2 ? f3 : f4);
    const int h = Hp / fac, w = Wp / fac;
    const UpIdx uy = up_index(y, h, Hp), ux = up_index(x, w, Wp);
    const float* base = map + (size_t)b * h * w * G;
    const float* p00 = base + ((size_t)uy.i0 * w + ux.i0) * G;
    const float* p01 = base + ((size_t)uy.i0 * w + ux.i1) * G;
    const float* p10 = base + ((size_t)uy.i1 * w + ux.i0) * G;
    const float* p11 = base + ((size_t)uy.i1 * w + ux.i1) * G;
#pragma unroll
    for (int c = 0; c < G; c += 4) {
      const float4 a = *(const float4*)(p00 + c), bq = *(const float4*)(p01 + c), cq = *(const float4*)(p10 + c), d = *(const float4*)(p11 + c);
      float o[4];
      o[0] = uy.l0 * (ux.l0 * a.x + ux.l1 * bq.x) + uy.l1 * (ux.l0 * cq.x + ux.l1 * d.x);
      o[1] = uy.l0 * (ux.l0 * a.y + ux.l1 * bq.y) + uy.l1 * (ux.l0 * cq.y + ux.l1 * d.y);
      o[2] = uy.l0 * (ux.l0 * a.z + ux.l1 * bq.z) + uy.l1 * (ux.l0 * cq.z + ux.l1 * d.z);
      o[3] = uy.l0 * (ux.l0 * a.w + ux.l1 * bq.w) + uy.l1 * (ux.l0 * cq.w + ux.l1 * d.w);
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int k = 0; k < 8; ++k) sacc[k] = fmaf(o[j], ws0[(G * g + c + j) * 8 + k], sacc[k]);
      if (STORE) *(float4*)(trow + c) = make_float4(o[0], o[1], o[2], o[3]);
    }
  }
  // the reference sums the 128 products of score_head.0 in channel order; here (g0 + g1) + (g2 + g3)
  if (g > 0) {
#pragma unroll
    for (int k = 0; k < 8; ++k) spart[g - 1][k][pl] = sacc[k];
  }
  __syncthreads();
  if (g == 0 && ok) {
#pragma unroll
    for (int k = 0; k < 8; ++k) sacc[k] = (sacc[k] + spart[0][k][pl]) + (spart[1][k][pl] + spart[2][k][pl]);
    float* sd = s8 + ((size_t)b * Hp * Wp + i) * 8;
    *(float4*)sd = make_float4(selu_(sacc[0]), selu_(sacc[1]), selu_(sacc[2]), selu_(sacc[3]));
    *(float4*)(sd + 4) = make_float4(selu_(sacc[4]), selu_(sacc[5]), selu_(sacc[6]), selu_(sacc[7]));
  }
  if (!STORE) return;
  const int p0 = blockIdx.x * 64;
  float* dst = x1234 + ((size_t)b * Hp * Wp + p0) * (4 * G);
#pragma unroll
  for (int k = 0; k < G / 4; ++k) {   // 64 pixels x G float4
    const int idx = t + 256 * k, pix = idx / G, c4 = idx % G;
    if (p0 + pix < Hp * Wp) *(float4*)(dst + (size_t)pix * (4 * G) + c4 * 4) = *(const float4*)&tile[pix * (4 * G + 4) + c4 * 4];
  }
}

// The product path of the same stage.  score_head.0 (1x1, 128 -> 8) and the bilinear up-sampling are both linear, so the
// projection of the three up-sampled groups commutes with the interpolation:
//   W0[32g .. 32g+31] . up(f_g)(y, x) = up(W0[32g ..] . f_g)(y, x)
// q_g = f_g x W0[32g:32g+32] is evaluated ONCE at the map's own resolution (1/4, 1/64, 1/1024 of the pixels; al_conv1x1<32, 8>)
// and the full-resolution pass interpolates 8 instead of 32 channels per map: 12 corner reads of 32 B and 96 FMAs per pixel
// where the unfactored form needs 12 x 128 B and 384 + 768.  Thread = pixel: the 16 -> 32 conv of x1, SELU, its share of the
// projection, the three 8-channel interpolations, SELU.  Rounding differs from the unfactored order by fp32 reassociation only.
template <int C1, int G>
__global__ __launch_bounds__(256) void al_assemble_proj_kernel(const float* __restrict__ x1, const float* __restrict__ q2,
                                                               const float* __restrict__ q3, const float* __restrict__ q4,
                                                               const float* __restrict__ w1, const float* __restrict__ ws0,
                                                               float* __restrict__ s8, int Hp, int Wp) {
  const int i = blockIdx.x * 256 + threadIdx.x, b = blockIdx.y;
  if (i >= Hp * Wp) return;
  const int y = i / Wp, x = i - y * Wp;
  float sacc[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) sacc[k] = 0.f;
  {
    const float* src = x1 + ((size_t)b * Hp * Wp + i) * C1;
    float a[C1];
#pragma unroll
    for (int c4 = 0; c4 < C1 / 4; ++c4) { const float4 v = *(const float4*)(src + c4 * 4); a[c4 * 4] = v.x; a[c4 * 4 + 1] = v.y; a[c4 * 4 + 2] = v.z; a[c4 * 4 + 3] = v.w; }
#pragma unroll
    for (int co = 0; co < G; co += 4) {
      float o[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ci = 0; ci < C1; ++ci)
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = fmaf(a[ci], w1[ci * G + co + j], o[j]);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        o[j] = selu_(o[j]);
#pragma unroll
        for (int k = 0; k < 8; ++k) sacc[k] = fmaf(o[j], ws0[(co + j) * 8 + k], sacc[k]);
      }
    }
  }
  float up[3][8];
#pragma unroll
  for (int g = 0; g < 3; ++g) {
    const int fac = g == 0 ? 2 : (g == 1 ? 8 : 32);
    const float* map = g == 0 ? q2 : (g == 1 ? q3 : q4);
    const int h = Hp / fac, w = Wp / fac;
    const UpIdx uy = up_index(y, h, Hp), ux = up_index(x, w, Wp);
    const float* base = map + (size_t)b * h * w * 8;
    const float* p00 = base + ((size_t)uy.i0 * w + ux.i0) * 8;
    const float* p01 = base + ((size_t)uy.i0 * w + ux.i1) * 8;
    const float* p10 = base + ((size_t)uy.i1 * w + ux.i0) * 8;
    const float* p11 = base + ((size_t)uy.i1 * w + ux.i1) * 8;
#pragma unroll
    for (int c = 0; c < 8; c += 4) {
      const float4 a = *(const float4*)(p00 + c), bq = *(const float4*)(p01 + c), cq = *(const float4*)(p10 + c), d = *(const float4*)(p11 + c);
      up[g][c + 0] = uy.l0 * (ux.l0 * a.x + ux.l1 * bq.x) + uy.l1 * (ux.l0 * cq.x + ux.l1 * d.x);
      up[g][c + 1] = uy.l0 * (ux.l0 * a.y + ux.l1 * bq.y) + uy.l1 * (ux.l0 * cq.y + ux.l1 * d.y);
      up[g][c + 2] = uy.l0 * (ux.l0 * a.z + ux.l1 * bq.z) + uy.l1 * (ux.l0 * cq.z + ux.l1 * d.z);
      up[g][c + 3] = uy.l0 * (ux.l0 * a.w + ux.l1 * bq.w) + uy.l1 * (ux.l0 * cq.w + ux.l1 * d.w);
    }
  }
  float o8[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) o8[k] = selu_((sacc[k] + up[0][k]) + (up[1][k] + up[2][k]));
  float* sd = s8 + ((size_t)b * Hp * Wp + i) * 8;
  *(float4*)sd = make_float4(o8[0], o8[1], o8[2], o8[3]);
  *(float4*)(sd + 4) = make_float4(o8[4], o8[5], o8[6], o8[7]);
}

// ---------------------------------------------------------------------------
// DKD soft-argmax refinement (ALN:176-216): one thread per keypoint.
__global__ __launch_bounds__(256) void al_dkd_refine_kernel(const float* __restrict__ score, const float* __restrict__ kpts_px,
                                                            const int* __restrict__ n_kpts, float* __restrict__ kpts_norm,
                                                            float* __restrict__ disp, float* __restrict__ kscore,
                                                            float* __restrict__ kpts_out, int H, int W, int capacity, int radius) {
  const int i = blockIdx.x * 256 + threadIdx.x, b = blockIdx.y;
  if (i >= n_kpts[b]) return;
  const size_t k = (size_t)b * capacity + i;
  const int x0 = (int)kpts_px[k * 2], y0 = (int)kpts_px[k * 2 + 1];
  const float* sm = score + (size_t)b * H * W;
  const int ks = 2 * radius + 1;
  float mx = -INFINITY;
  for (int dy = -radius; dy <= radius; ++dy)
    for (int dx = -radius; dx <= radius; ++dx) {
      const int yy = y0 + dy, xx = x0 + dx;
      const float v = (yy >= 0 && yy < H && xx >= 0 && xx < W) ? sm[(size_t)yy * W + xx] : 0.f;  // nn.Unfold zero padding
      mx = fmaxf(mx, v);
    }
  float se = 0.f, sx = 0.f, sy = 0.f;
  for (int dy = -radius; dy <= radius; ++dy)
    for (int dx = -radius; dx <= radius; ++dx) {
      const int yy = y0 + dy, xx = x0 + dx;
      const float v = (yy >= 0 && yy < H && xx >= 0 && xx < W) ? sm[(size_t)yy * W + xx] : 0.f;
      const float e = expf((v - mx) / 0.1f);
      se += e; sx += e * (float)dx; sy += e * (float)dy;
    }
  const float rx = sx / se, ry = sy / se;
  float dsum = 0.f;
  for (int dy = -radius; dy <= radius; ++dy)
    for (int dx = -radius; dx <= radius; ++dx) {
      const int yy = y0 + dy, xx = x0 + dx;
      const float v = (yy >= 0 && yy < H && xx >= 0 && xx < W) ? sm[(size_t)yy * W + xx] : 0.f;
      const float e = expf((v - mx) / 0.1f);
      const float ux = ((float)dx - rx) / (float)radius, uy = ((float)dy - ry) / (float)radius;
      const float nrm = sqrtf(ux * ux + uy * uy);
      dsum += e * (nrm * nrm);
    }
  (void)ks;
  const float wx = (float)(W - 1), wy = (float)(H - 1);
  const float kx = ((float)x0 + rx) / wx * 2.f - 1.f, ky = ((float)y0 + ry) / wy * 2.f - 1.f;
  kpts_norm[k * 2] = kx; kpts_norm[k * 2 + 1] = ky;
  disp[k] = dsum / se;
  // bilinear score at the refined position (grid_sample, align_corners=True, zeros padding)
  const float ix = ((kx + 1.f) / 2.f) * wx, iy = ((ky + 1.f) / 2.f) * wy;
  const float fx = floorf(ix), fy = floorf(iy);
  const int xa = (int)fx, ya = (int)fy;
  float val = 0.f;
  const float wts[4] = {(fx + 1.f - ix) * (fy + 1.f - iy), (ix - fx) * (fy + 1.f - iy), (fx + 1.f - ix) * (iy - fy), (ix - fx) * (iy - fy)};
  const int xs[4] = {xa, xa + 1, xa, xa + 1}, ys[4] = {ya, ya, ya + 1, ya + 1};
#pragma unroll
  for (int c = 0; c < 4; ++c)
    if (xs[c] >= 0 && xs[c] < W && ys[c] >= 0 && ys[c] < H) val += sm[(size_t)ys[c] * W + xs[c]] * wts[c];
  kscore[k] = val;
  // ALN:687: wh * (kpts + 1) / 2
  kpts_out[k * 2] = wx * (kx + 1.f) / 2.f;
  kpts_out[k * 2 + 1] = wy * (ky + 1.f) / 2.f;
}

// ---------------------------------------------------------------------------
// SDDH (ALN:503-558).  The 128-channel map x1234 is not stored: each cell a keypoint touches is
// re-evaluated from its sources with the arithmetic of al_assemble_kernel (lane -> channels 2*lane,
// 2*lane+1: lanes 0-15 the 16->32 conv of x1, then one up-sampled map per 16 lanes; aliked-t16: one channel per lane,
// 8 -> 16 conv) and L2-normalised on the fly (F.normalize over the 4 G channels, ALN:669).
// ---- the evaluation is split into "issue the loads" and "finish", so that a wave can keep the gathers of MANY cells in
// flight (the kernels below were latency-bound with one cell's loads per round trip: 1.5 ms for 2 M cells).  Lanes 0 .. C1-1
// (group 0) hold ONE input channel of x1 each — the C1 values reach all lanes through v_readlane (wave-uniform SGPRs) and the
// lane's CPL = G / 16 columns of the C1 -> G conv weights stay in registers for the whole kernel; lanes 16-63 hold the four bilinear
// corners of their up-sampled map.  9 VGPRs per cell in flight.  (CPL = 1: the .y halves are zero throughout.)
struct FeatCell { float x1v; float2 c00, c01, c10, c11; float l0y, l1y, l0x, l1x; };
template <int CPL>
__device__ __forceinline__ float2 feat_load(const float* p) {
  if (CPL == 2) return *(const float2*)p;
  return make_float2(p[0], 0.f);
}
template <int C1, int G>
__device__ __forceinline__ void feat_issue(const AlFeat& F, int b, int Y, int X, int lane, FeatCell& r) {
  constexpr int CPL = G / 16;
  const int g = lane >> 4, cc = (lane & 15) * CPL;
  r.x1v = F.x1[(((size_t)b * F.Hp + Y) * F.Wp + X) * C1 + (lane & (C1 - 1))];
  const int fac = g <= 1 ? 2 : (g == 2 ? 8 : 32);
  const float* map = g <= 1 ? F.f2 : (g == 2 ? F.f3 : F.f4);
  const int h = F.Hp / fac, w = F.Wp / fac;
  const UpIdx uy = up_index(Y, h, F.Hp), ux = up_index(X, w, F.Wp);
  const float* base = map + (size_t)b * h * w * G + cc;
  r.c00 = feat_load<CPL>(base + ((size_t)uy.i0 * w + ux.i0) * G); r.c01 = feat_load<CPL>(base + ((size_t)uy.i0 * w + ux.i1) * G);
  r.c10 = feat_load<CPL>(base + ((size_t)uy.i1 * w + ux.i0) * G); r.c11 = feat_load<CPL>(base + ((size_t)uy.i1 * w + ux.i1) * G);
  r.l0y = uy.l0; r.l1y = uy.l1; r.l0x = ux.l0; r.l1x = ux.l1;
}
// -> this lane's CPL channels of the L2-NORMALISED 4 G-vector of the cell (F.normalize, ALN:669), same arithmetic as al_assemble_kernel
template <int C1, int G>
__device__ __forceinline__ float2 feat_finish(const FeatCell& r, const float2 (&w1r)[C1], int lane) {
  constexpr int CPL = G / 16;
  float o0 = 0.f, o1 = 0.f;
#pragma unroll
  for (int ci = 0; ci < C1; ++ci) {
    const float a = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(r.x1v), ci));
    o0 = fmaf(a, w1r[ci].x, o0); o1 = fmaf(a, w1r[ci].y, o1);
  }
  float2 v;
  if ((lane >> 4) == 0) v = make_float2(selu_(o0), CPL == 2 ? selu_(o1) : 0.f);
  else v = make_float2(r.l0y * (r.l0x * r.c00.x + r.l1x * r.c01.x) + r.l1y * (r.l0x * r.c10.x + r.l1x * r.c11.x),
                       r.l0y * (r.l0x * r.c00.y + r.l1x * r.c01.y) + r.l1y * (r.l0x * r.c10.y + r.l1x * r.c11.y));
  const float den = fmaxf(sqrtf(wave_sum_dpp(v.x * v.x + v.y * v.y)), 1e-12f);   // DPP reduction: no LDS round trips in the gather loop
  return make_float2(v.x / den, v.y / den);
}
template <int C1, int G>
__device__ __forceinline__ void feat_weights(const AlFeat& F, int lane, float2 (&w1r)[C1]) {
  constexpr int CPL = G / 16;
  const int cc = (lane & 15) * CPL;
#pragma unroll
  for (int ci = 0; ci < C1; ++ci) w1r[ci] = feat_load<CPL>(F.w1 + ci * G + cc);
}
// wave per keypoint: the 3x3 patch of normalised features -> patches [kpt][4 G ci][9 cells]; all 9 cells' loads in flight
template <int C1, int G>
__global__ __launch_bounds__(256) void al_sddh_patches_kernel(AlFeat F, const float* __restrict__ kpts_norm,
                                                              const int* __restrict__ n_kpts, float* __restrict__ patches, int H,
                                                              int W, int pad_t, int pad_l, int capacity) {
  constexpr int CPL = G / 16;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int i = blockIdx.x * 4 + wv, b = blockIdx.y;
  if (i >= n_kpts[b]) return;
  float2 w1r[C1];
  feat_weights<C1, G>(F, lane, w1r);
  const size_t k = (size_t)b * capacity + i;
  const float kwx = (kpts_norm[k * 2] / 2.f + 0.5f) * (float)(W - 1), kwy = (kpts_norm[k * 2 + 1] / 2.f + 0.5f) * (float)(H - 1);
  const int xl = (int)kwx, yl = (int)kwy;  // .long()
  int cx = (int)((float)xl - 1.5f + 1.f), cy = (int)((float)yl - 1.5f + 1.f);  // (corner - ps/2 + 1).long(), truncation
  cx = min(max(cx, 0), W - 1 - 3); cy = min(max(cy, 0), H - 1 - 3);
  FeatCell cell[9];
#pragma unroll
  for (int c = 0; c < 9; ++c) feat_issue<C1, G>(F, b, cy + c / 3 + pad_t, cx + c % 3 + pad_l, lane, cell[c]);
  // layout [kpt][ci][ky][kx] flattened as ci*9 + cell to match offset_conv.0.weight (2M, 4G, 3, 3)
  float* dst = patches + k * (4 * G * 9);
#pragma unroll
  for (int c = 0; c < 9; ++c) {
    const float2 v = feat_finish<C1, G>(cell[c], w1r, lane);
    dst[(lane * CPL) * 9 + c] = v.x;
    if (CPL == 2) dst[(lane * CPL + 1) * 9 + c] = v.y;
  }
}
// wave per keypoint: offsets = clamp(W2 * selu(hidden) + b2), then M (16: aliked-n16 / n16rot, 32: aliked-n32) bilinear samples of the
// normalised feature map -> feats [kpt][M][4 G]; the 8 cells of two samples are in flight together
template <int M, int C1, int G>
__global__ __launch_bounds__(256) void al_sddh_sample_kernel(AlFeat F, const float* __restrict__ kpts_norm,
                                                             const int* __restrict__ n_kpts, const float* __restrict__ hidden,
                                                             const float* __restrict__ w2, const float* __restrict__ b2,
                                                             float* __restrict__ feats, int H, int W, int pad_t, int pad_l, int capacity) {
  __shared__ float offs[4][2 * M];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int i = blockIdx.x * 4 + wv, b = blockIdx.y;
  const bool live = i < n_kpts[b];
  const size_t k = (size_t)b * capacity + (live ? i : 0);
  const float max_off = (float)max(H, W) / 4.0f;
  if (lane < 2 * M) {  // offset_conv.2 (1x1, 2M -> 2M) on selu(hidden); w2 is [in][out]
    float o = b2[lane];
    for (int c = 0; c < 2 * M; ++c) o = fmaf(selu_(hidden[k * (2 * M) + c]), w2[c * (2 * M) + lane], o);
    offs[wv][lane] = fminf(fmaxf(o, -max_off), max_off);
  }
  __syncthreads();
  if (!live) return;
  constexpr int CPL = G / 16;
  float2 w1r[C1];
  feat_weights<C1, G>(F, lane, w1r);
  const float wx = (float)(W - 1), wy = (float)(H - 1);
  const float kwx = (kpts_norm[k * 2] / 2.f + 0.5f) * wx, kwy = (kpts_norm[k * 2 + 1] / 2.f + 0.5f) * wy;
  constexpr int SB = 2;   // samples per round: 8 cells in flight (104 VGPRs), 3 waves per SIMD
  for (int p0 = 0; p0 < M; p0 += SB) {
    FeatCell cell[SB][4];
    float wts[SB][4];
    bool in[SB][4];
#pragma unroll
    for (int j = 0; j < SB; ++j) {
      const int p = p0 + j;
      // offset[:, :, 0, 0].view(N, 2, M): channel p = x offset, channel M + p = y offset (ALN:540)
      const float gx = 2.0f * (kwx + offs[wv][p]) / wx - 1.f, gy = 2.0f * (kwy + offs[wv][M + p]) / wy - 1.f;
      const float ix = ((gx + 1.f) / 2.f) * wx, iy = ((gy + 1.f) / 2.f) * wy;
      const float fx = floorf(ix), fy = floorf(iy);
      const int xa = (int)fx, ya = (int)fy;
      wts[j][0] = (fx + 1.f - ix) * (fy + 1.f - iy); wts[j][1] = (ix - fx) * (fy + 1.f - iy);
      wts[j][2] = (fx + 1.f - ix) * (iy - fy); wts[j][3] = (ix - fx) * (iy - fy);
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const int xs = xa + (c & 1), ys = ya + (c >> 1);
        in[j][c] = xs >= 0 && xs < W && ys >= 0 && ys < H;  // wave-uniform
        feat_issue<C1, G>(F, b, (in[j][c] ? ys : 0) + pad_t, (in[j][c] ? xs : 0) + pad_l, lane, cell[j][c]);
      }
    }
#pragma unroll
    for (int j = 0; j < SB; ++j) {
      float o0 = 0.f, o1 = 0.f;
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const float2 v = feat_finish<C1, G>(cell[j][c], w1r, lane);
        if (in[j][c]) { o0 += v.x * wts[j][c]; o1 += v.y * wts[j][c]; }
      }
      if (CPL == 2) *(float2*)(feats + (k * M + p0 + j) * (4 * G) + lane * 2) = make_float2(o0, o1);
      else feats[(k * M + p0 + j) * (4 * G) + lane] = o0;
    }
  }
}
// L2-normalise rows of [batch][capacity][C] (C = 128: two values per lane), wave per row
__global__ __launch_bounds__(256) void al_normalize_rows_kernel(float* __restrict__ x, const int* __restrict__ n_rows, int capacity, int C) {
  const int lane = threadIdx.x & 63, i = blockIdx.x * 4 + (threadIdx.x >> 6), b = blockIdx.y;
  if (i >= n_rows[b]) return;
  float* p = x + ((size_t)b * capacity + i) * C;
  float ss = 0.f;
  for (int c = lane; c < C; c += 64) ss += p[c] * p[c];
  const float den = fmaxf(sqrtf(wave_sum(ss)), 1e-12f);
  for (int c = lane; c < C; c += 64) p[c] = p[c] / den;
}

// mean of a [batch][n] map (fp64 two-stage), for DKD's fallback threshold (ALN:165-168)
__global__ __launch_bounds__(256) void al_mean_partial_kernel(const float* __restrict__ x, int n, double* __restrict__ partial) {
  __shared__ double red[256];
  const int b = blockIdx.y, t = threadIdx.x;
  double s = 0.0;
  for (int i = blockIdx.x * 256 + t; i < n; i += BN_BLOCKS * 256) s += (double)x[(size_t)b * n + i];
  red[t] = s;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) { if (t < o) red[t] += red[t + o]; __syncthreads(); }
  if (t == 0) partial[b * BN_BLOCKS + blockIdx.x] = red[0];
}
__global__ void al_mean_final_kernel(const double* __restrict__ partial, int n, float* __restrict__ mean) {
  const int b = threadIdx.x;
  double s = 0.0;
  for (int k = 0; k < BN_BLOCKS; ++k) s += partial[b * BN_BLOCKS + k];
  mean[b] = (float)(s / n);
}
__global__ void al_pick_threshold_kernel(const int* __restrict__ ncand, const float* __restrict__ mean, float thr, float* __restrict__ out) {
  const int b = threadIdx.x;
  out[b] = (thr > 0.f && ncand[b] > 0) ? thr : mean[b];  // ALN:163-168 (per image: DIM runs batch 1)
}

}  // namespace

// ---------------------------------------------------------------------------
int launch_al_conv3x3(const float* in, int cin, const float* w, const float* bias, float* out, int cout, int batch, int H,
                      int W, int act, int crop_y, int crop_x, int out_h, int out_w, hipStream_t s) {
  const int tx = cdiv(W, 16), ty = cdiv(H, 16);
  dim3 grid(tx * ty, 1, batch);
#define AL_C3(CC, CIP, CO) hipLaunchKernelGGL(HIP_KERNEL_NAME(al_conv3x3_kernel<CC, CO>), grid, dim3(256), 0, s, in, cin, CIP, w, bias, out, cout, H, W, act, tx, crop_y, crop_x, out_h, out_w)
  if (cin == 3 && cout == 16) AL_C3(4, 4, 16);
  else if (cin == 3 && cout == 8) AL_C3(4, 4, 8);        // aliked-t16: 8 / 16 / 32 / 64 channels
  else if (cin == 8 && cout == 8) AL_C3(8, 8, 8);
  else if (cin == 8 && cout == 16) AL_C3(8, 8, 16);
  else if (cin == 16 && cout == 18) AL_C3(16, 16, 20);
  else if (cin == 16 && cout == 16) AL_C3(16, 16, 16);
  else if (cin == 16 && cout == 32) AL_C3(16, 16, 32);
  else if (cin == 32 && cout == 32) AL_C3(32, 32, 32);
  else if (cin == 32 && cout == 18) AL_C3(32, 32, 20);
  else if (cin == 64 && cout == 18) AL_C3(32, 64, 20);
  else if (cin == 128 && cout == 18) AL_C3(32, 128, 20);
  else if (cin == 8 && cout == 4) AL_C3(8, 8, 4);
  else if (cin == 4 && cout == 4) AL_C3(4, 4, 4);
  else if (cin == 4 && cout == 1) AL_C3(4, 4, 4);
  else { dim_set_error("aliked conv3x3: unsupported channels %d -> %d", cin, cout); return -2; }
#undef AL_C3
  DIM_LAUNCH_CHECK();
  return 0;
}
int launch_al_conv1x1(const float* in, int cin, const float* w, const float* bias, float* out, int cout, int n_pixels, int act,
                      hipStream_t s) {
  dim3 grid(cdiv(n_pixels, 256));
#define AL_C1(CI, CO) hipLaunchKernelGGL(HIP_KERNEL_NAME(al_conv1x1_kernel<CI, CO>), grid, dim3(256), 0, s, in, w, bias, out, n_pixels, act)
  if (cin == 16 && cout == 32) AL_C1(16, 32);
  else if (cin == 8 && cout == 16) AL_C1(8, 16);         // aliked-t16
  else if (cin == 16 && cout == 16) AL_C1(16, 16);
  else if (cin == 32 && cout == 16) AL_C1(32, 16);
  else if (cin == 64 && cout == 16) AL_C1(64, 16);
  else if (cin == 16 && cout == 8) AL_C1(16, 8);
  else if (cin == 32 && cout == 32) AL_C1(32, 32);
  else if (cin == 32 && cout == 64) AL_C1(32, 64);
  else if (cin == 64 && cout == 32) AL_C1(64, 32);
  else if (cin == 64 && cout == 128) AL_C1(64, 128);
  else if (cin == 128 && cout == 32) AL_C1(128, 32);
  else if (cin == 32 && cout == 8) AL_C1(32, 8);
  else { dim_set_error("aliked conv1x1: unsupported channels %d -> %d", cin, cout); return -2; }
#undef AL_C1
  DIM_LAUNCH_CHECK();
  return 0;
}
int launch_al_pad_replicate(const float* img, float* out, int batch, int H, int W, int Hp, int Wp, int pad_t, int pad_l, int in_ch,
                            hipStream_t s) {
  hipLaunchKernelGGL(al_pad_kernel, dim3(cdiv(Hp * Wp, 256), batch), dim3(256), 0, s, img, out, H, W, Hp, Wp, pad_t, pad_l, in_ch);
  DIM_LAUNCH_CHECK();
  return 0;
}
int launch_al_avgpool(const float* in, float* out, int batch, int H, int W, int C, int k, hipStream_t s) {
  hipLaunchKernelGGL(al_avgpool_kernel, dim3(cdiv((H / k) * (W / k) * (C / 4), 256), batch), dim3(256), 0, s, in, out, H, W, C, k);
  DIM_LAUNCH_CHECK();
  return 0;
}
int launch_al_bn_stats(const float* x, int batch, int n_pixels, int C, const float* gamma, const float* beta_w, double* partial,
                       float* alpha, float* beta, hipStream_t s) {
  DIM_REQUIRE(C == 8 || C == 16 || C == 32 || C == 64 || C == 128, "aliked bn: C=%d unsupported", C);
  hipLaunchKernelGGL(al_bn_partial_kernel, dim3(BN_BLOCKS, batch), dim3(256), 0, s, x, n_pixels, C, partial);
  hipLaunchKernelGGL(al_bn_final_kernel, dim3(batch), dim3(128), 0, s, (const double*)partial, n_pixels, C, gamma, beta_w, alpha, beta);
  DIM_LAUNCH_CHECK();
  return 0;
}
int launch_al_bn_apply(const float* x, const float* alpha, const float* beta, const float* residual, float* out, int batch,
                       int n_pixels, int C, hipStream_t s) {
  hipLaunchKernelGGL(al_bn_apply_kernel, dim3(cdiv(n_pixels * (C / 4), 256), batch), dim3(256), 0, s, x, alpha, beta, residual, out, n_pixels, C);
  DIM_LAUNCH_CHECK();
  return 0;
}
int launch_al_bn_apply_pool(const float* x, const float* alpha, const float* beta, const float* residual, float* out, float* pooled, int batch,
                            int H, int W, int C, int k, hipStream_t s) {
  DIM_REQUIRE((k == 2 || k == 4) && H % k == 0 && W % k == 0 && C % 4 == 0, "aliked bn_apply_pool: k %d on %dx%d", k, H, W);
  const dim3 grid(cdiv((H / k) * (W / k) * (C / 4), 256), batch);
  if (k == 2) hipLaunchKernelGGL(HIP_KERNEL_NAME(al_bn_apply_pool_kernel<2>), grid, dim3(256), 0, s, x, alpha, beta, residual, out, pooled, H, W, C);
  else hipLaunchKernelGGL(HIP_KERNEL_NAME(al_bn_apply_pool_kernel<4>), grid, dim3(256), 0, s, x, alpha, beta, residual, out, pooled, H, W, C);
  DIM_LAUNCH_CHECK();
  return 0;
}
int launch_al_deform_conv(const float* in, int cin, const float* offsets, int off_c, const float* w, const SplitWeights* wx, unsigned* sat,
                          float* cols, float* out, int cout, int batch, int H, int W, hipStream_t s) {
  DIM_REQUIRE(cout % 32 == 0 && (cin == 16 || cin == 32 || cin == 64 || cin == 128), "deform conv: cin %d cout %d", cin, cout);
  const int rows = batch * H * W, krow = al_deform_krow(cin);
  const dim3 grid((unsigned)(((size_t)rows * 9 * (cin / 4) + 255) / 256));
  if (cin == 16) hipLaunchKernelGGL(HIP_KERNEL_NAME(al_deform_gather_kernel<16>), grid, dim3(256), 0, s, in, offsets, off_c, cols, H, W, rows);
  else if (cin == 32) hipLaunchKernelGGL(HIP_KERNEL_NAME(al_deform_gather_kernel<32>), grid, dim3(256), 0, s, in, offsets, off_c, cols, H, W, rows);
  else if (cin == 64) hipLaunchKernelGGL(HIP_KERNEL_NAME(al_deform_gather_kernel<64>), grid, dim3(256), 0, s, in, offsets, off_c, cols, H, W, rows);
  else hipLaunchKernelGGL(HIP_KERNEL_NAME(al_deform_gather_kernel<128>), grid, dim3(256), 0, s, in, offsets, off_c, cols, H, W, rows);
  DIM_LAUNCH_CHECK();
  GemmArgs g;   // the weight operand has krow rows (zero rows past 9 cin: al_deform_krow)
  g.A0 = cols; g.lda0 = krow; g.B = w; g.ldb = cout; g.C = out; g.ldc = cout; g.M = rows; g.N = cout; g.K = krow;
  if (wx != nullptr) { g.set_split(*wx); g.sat = sat; return launch_gemm_x6(g, 1, s); }   // fp16x3 on the matrix cores
  return launch_gemm(g, 1, s);
}
int launch_al_clamp(float* x, size_t n, float lim, hipStream_t s) {
  hipLaunchKernelGGL(al_clamp_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, x, n, lim);
  DIM_LAUNCH_CHECK();
  return 0;
}
int launch_al_assemble(const float* x1, const float* f2, const float* f3, const float* f4, const float* w1, const float* ws0,
                       float* x1234, float* s8, int c1, int batch, int Hp, int Wp, hipStream_t s) {
  const dim3 grid(cdiv(Hp * Wp, 64), batch);
#define AL_ASM(ST, C1, G) hipLaunchKernelGGL(HIP_KERNEL_NAME(al_assemble_kernel<ST, C1, G>), grid, dim3(256), 0, s, x1, f2, f3, f4, w1, ws0, x1234, s8, Hp, Wp)
  if (c1 == 16) { if (x1234) AL_ASM(true, 16, 32); else AL_ASM(false, 16, 32); }
  else { if (x1234) AL_ASM(true, 8, 16); else AL_ASM(false, 8, 16); }
#undef AL_ASM
  DIM_LAUNCH_CHECK();
  return 0;
}
int launch_al_assemble_proj(const float* x1, const float* q2, const float* q3, const float* q4, const float* w1, const float* ws0, float* s8,
                            int c1, int batch, int Hp, int Wp, hipStream_t s) {
  if (c1 == 16) hipLaunchKernelGGL(HIP_KERNEL_NAME(al_assemble_proj_kernel<16, 32>), dim3(cdiv(Hp * Wp, 256), batch), dim3(256), 0, s, x1, q2, q3, q4, w1, ws0, s8, Hp, Wp);
  else hipLaunchKernelGGL(HIP_KERNEL_NAME(al_assemble_proj_kernel<8, 16>), dim3(cdiv(Hp * Wp, 256), batch), dim3(256), 0, s, x1, q2, q3, q4, w1, ws0, s8, Hp, Wp);
  DIM_LAUNCH_CHECK();
  return 0;
}
int launch_al_dkd_refine(const float* score, const float* kpts_px, const int* n_kpts, float* kpts_norm, float* disp, float* kscore,
                         float* kpts_out, int batch, int H, int W, int capacity, int radius, hipStream_t s) {
  hipLaunchKernelGGL(al_dkd_refine_kernel, dim3(cdiv(capacity, 256), batch), dim3(256), 0, s, score, kpts_px, n_kpts, kpts_norm, disp, kscore, kpts_out, H, W, capacity, radius);
  DIM_LAUNCH_CHECK();
  return 0;
}
int launch_al_sddh_patches(const AlFeat& F, const float* kpts_norm, const int* n_kpts, float* patches, int batch, int H, int W,
                           int pad_t, int pad_l, int capacity, hipStream_t s) {
  if (F.c1 == 16) hipLaunchKernelGGL(HIP_KERNEL_NAME(al_sddh_patches_kernel<16, 32>), dim3(cdiv(capacity, 4), batch), dim3(256), 0, s, F, kpts_norm, n_kpts, patches, H, W, pad_t, pad_l, capacity);
  else hipLaunchKernelGGL(HIP_KERNEL_NAME(al_sddh_patches_kernel<8, 16>), dim3(cdiv(capacity, 4), batch), dim3(256), 0, s, F, kpts_norm, n_kpts, patches, H, W, pad_t, pad_l, capacity);
  DIM_LAUNCH_CHECK();
  return 0;
}
int launch_al_sddh_sample(const AlFeat& F, const float* kpts_norm, const int* n_kpts, const float* off_hidden, const float* w2,
                          const float* b2, float* feats, int M, int batch, int H, int W, int pad_t, int pad_l, int capacity, hipStream_t s) {
  DIM_REQUIRE((M == 16 || M == 32) && (F.c1 == 16 || M == 16), "al_sddh_sample: M = %d sample positions (16 or 32; aliked-t16: 16)", M);
  const dim3 grid(cdiv(capacity, 4), batch);
#define AL_SMP(MM, C1, G) hipLaunchKernelGGL(HIP_KERNEL_NAME(al_sddh_sample_kernel<MM, C1, G>), grid, dim3(256), 0, s, F, kpts_norm, n_kpts, off_hidden, w2, b2, feats, H, W, pad_t, pad_l, capacity)
  if (F.c1 == 8) AL_SMP(16, 8, 16);
  else if (M == 16) AL_SMP(16, 16, 32);
  else AL_SMP(32, 16, 32);
#undef AL_SMP
  DIM_LAUNCH_CHECK();
  return 0;
}
int launch_al_normalize_rows(float* x, const int* n_rows, int batch, int capacity, int C, hipStream_t s) {
  hipLaunchKernelGGL(al_normalize_rows_kernel, dim3(cdiv(capacity, 4), batch), dim3(256), 0, s, x, n_rows, capacity, C);
  DIM_LAUNCH_CHECK();
  return 0;
}
int launch_al_mean(const float* x, int batch, int n, double* partial, float* mean, hipStream_t s) {
  hipLaunchKernelGGL(al_mean_partial_kernel, dim3(BN_BLOCKS, batch), dim3(256), 0, s, x, n, partial);
  hipLaunchKernelGGL(al_mean_final_kernel, dim3(1), dim3(batch), 0, s, (const double*)partial, n, mean);
  DIM_LAUNCH_CHECK();
  return 0;
}
int launch_al_pick_threshold(const int* ncand, const float* mean, float thr, float* thr_out, int batch, hipStream_t s) {
  hipLaunchKernelGGL(al_pick_threshold_kernel, dim3(1), dim3(batch), 0, s, ncand, mean, thr, thr_out);
  DIM_LAUNCH_CHECK();
  return 0;
}
