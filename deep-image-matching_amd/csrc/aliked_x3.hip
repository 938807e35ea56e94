// ALIKED's small-channel convolutions on the fp16 matrix cores at fp32 accuracy (fp16x3, dim_common.h SplitMma<2>).
//
// The dense stage of ALIKED (ALN:644-675) spends its time in four convolutions over full- and half-resolution maps with
// 3 / 16 / 32 channels (block1: 3 -> 16 -> 16 at 1/1; block2: 16 -> 32 -> 32 and the 1x1 down-sample 16 -> 32 at 1/2).  As
// direct fp32 VALU kernels (aliked.hip: one pixel per thread, weights through the scalar cache) they run at 40-50 TFLOP/s —
// compute-bound at 5x their HBM time.  Here the same convolutions are implicit GEMMs on v_mfma_f32_32x32x16_f16:
//
//   M = 32 consecutive pixels of one image row (lane = pixel of the A operand)
//   K = 16 input channels of ONE tap per MFMA step (a 3x3 conv over CIN channels = 9 * CIN / 16 steps; CIN = 3 is padded to 16)
//   N = 32 output channels (COUT = 16 uses half of the tile: the kernels are HBM-bound with it, not matrix-bound)
//
// Activations: the (TH + 2) x 34 halo tile of a TH x 32 output tile is read once as fp32 NHWC, multiplied by the activation
// scale, split into two fp16 pieces and staged in LDS as [pixel][h: CIN/2 dwords | l: CIN/2 dwords | 4 pad] — the CIN + 4 dword
// pixel stride makes every 16-lane group of a ds_read_b128 cover all 64 banks exactly once.  Weights: split on the host per
// output channel (power-of-two scale, dim_kernels.h split_weights), stored in MFMA-fragment order [plane][K/16][k-half][32][8]
// and read through L1 (18-72 KB per layer, reused by every workgroup).  Three cross terms lh, hl, hh per step.
//
// Epilogue: C layout lane = output channel, registers = 16 pixels: x (1 / (weight scale x activation scale)) + bias, stored as
// fp32 NHWC; and — because BatchNorm runs in TRAINING mode (quirk Q7) and needs the per-image statistics of exactly this
// tensor — the per-channel sum and sum of squares of the tile are reduced in-lane (lane = channel), across the two lane
// halves and the four waves, and written as ONE fp64 partial per (workgroup, channel): the separate statistics pass over the
// map (al_bn_partial_kernel: a full extra read of every conv output) disappears.  al_bn_final_tiles_kernel sums the partials
// in a fixed order (deterministic; no atomics).
//
// Range: |activation| <= 4094 is exact (DIM_F16_ACT_LIMIT); the staging tracks what it splits and bumps DIM_SAT_ALIKED
// otherwise — the wrapper then repeats the call on the fp32 VALU path.
#include <math.h>

#include "../../include/dim_hip.h"
#include "aliked_kernels.h"

namespace {
using S = SplitMma<2>;
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float selu_x3(float x) {   // ATen's elu kernel, as aliked.hip selu_
  const float scale = 1.0507009873554804934193349852946f, alpha = 1.6732632423543772848170429916717f;
  const float l2e_hi = 1.44269502162933349609375f, l2e_lo = 1.925962989e-8f;   // exp_le0 of aliked.hip
  const float t = x * l2e_hi;
  float r = fmaf(x, l2e_hi, -t);
  r = fmaf(x, l2e_lo, r);
  const float y = __builtin_amdgcn_exp2f(t);
  return x <= 0.0f ? (fmaf(y, r * 0.693147180559945309417f, y) - 1.0f) * (alpha * scale) : x * scale;
}

// TAPS: 9 (3x3, zero padding 1) or 1 (1x1).  CIN: padded input channels (16 or 32).  TH: tile rows (multiple of 4).
template <int CIN, int TAPS, int TH, bool WREG = true>
__global__ __launch_bounds__(256, (WREG ? 2 : 3)) void al_convx3_kernel(const float* __restrict__ in, int in_c, const u32x4* __restrict__ wfrag,
                                                        const float* __restrict__ inv_ch, const float* __restrict__ bias,
                                                        float* __restrict__ out, int out_c, int H, int W, int tiles_x,
                                                        double* __restrict__ partial, int n_wg, unsigned* __restrict__ sat,
                                                        const float* __restrict__ in_alpha, const float* __restrict__ in_beta) {
  constexpr int R = TAPS == 9 ? 1 : 0, HW_ = 32 + 2 * R, HH = TH + 2 * R;   // halo
  constexpr int PSD = CIN + 4, KS = CIN / 16, MR = TH / 4;
  __shared__ __attribute__((aligned(16))) unsigned tile[HH * HW_ * PSD];
  __shared__ double red[4][32][2];
  const int t = threadIdx.x, b = blockIdx.z, lane = t & 63, wv = t >> 6, lx = lane & 31, half = lane >> 5;
  const int ty0 = (blockIdx.x / tiles_x) * TH, tx0 = (blockIdx.x % tiles_x) * 32;
  const float* src = in + (size_t)b * H * W * in_c;

  // ---- stage the halo tile: fp32 -> two fp16 planes.  ALL global loads of the tile are issued before the first value is
  // used (one exposed HBM round trip per workgroup instead of one per loop iteration: with a load inside every iteration the
  // first version of this kernel spent 25 us per workgroup waiting), and — for CIN = 16 — so are the 18 weight fragments. ----
  constexpr bool WPRE = WREG && (TAPS * KS <= 9);   // all weight fragments held in registers (72 VGPRs) — or streamed one tap ahead through L1
  u32x4 wreg[WPRE ? TAPS * KS : 1][2];
  if (WPRE) {
#pragma unroll
    for (int j = 0; j < TAPS * KS; ++j)
#pragma unroll
      for (int pl = 0; pl < 2; ++pl) wreg[j][pl] = wfrag[(((size_t)pl * (TAPS * KS) + j) * 2 + half) * 32 + lx];
  }
  float vmax = 0.0f;
  if (in_c % 4 == 0) {
    constexpr int Q = CIN / 4, NIT = (HH * HW_ * Q + 255) / 256;
    float4 rv[NIT];
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int i = t + 256 * it, p = i / Q, q = i - p * Q;
      const int gy = ty0 + p / HW_ - R, gx = tx0 + p % HW_ - R;
      const bool ok = i < HH * HW_ * Q && q * 4 < in_c && gy >= 0 && gy < H && gx >= 0 && gx < W;
      rv[it] = ok ? *(const float4*)(src + ((size_t)gy * W + gx) * in_c + q * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int i = t + 256 * it, p = i / Q, q = i - p * Q;
      if (i >= HH * HW_ * Q) break;
      float4 v = rv[it];
      if (in_alpha != nullptr) {  // the producer's BatchNorm + SELU applied on the way in (the padding stays zero: it pads the ACTIVATED map)
        const int gy = ty0 + p / HW_ - R, gx = tx0 + p % HW_ - R;
        if (gy >= 0 && gy < H && gx >= 0 && gx < W) {
          const float4 a = *(const float4*)(in_alpha + b * in_c + q * 4), bb = *(const float4*)(in_beta + b * in_c + q * 4);
          v = make_float4(selu_x3(v.x * a.x + bb.x), selu_x3(v.y * a.y + bb.y), selu_x3(v.z * a.z + bb.z), selu_x3(v.w * a.w + bb.w));
        }
      }
      unsigned p0[2], p1[2];
      S::split(v.x, v.y, S::act_scale(), p0);
      S::split(v.z, v.w, S::act_scale(), p1);
      vmax = sat_track(sat_track(vmax, v.x, v.y), v.z, v.w);
      unsigned* d = &tile[p * PSD + q * 2];
      *(u32x2*)d = u32x2{p0[0], p1[0]};
      *(u32x2*)(d + CIN / 2) = u32x2{p0[1], p1[1]};
    }
  } else {  // in_c = 3 (the RGB image): thread = pixel, channels zero padded to CIN
    constexpr int NIT = (HH * HW_ + 255) / 256;
    float r0[NIT], r1[NIT], r2[NIT];
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int p = t + 256 * it;
      const int gy = ty0 + p / HW_ - R, gx = tx0 + p % HW_ - R;
      const bool ok = p < HH * HW_ && gy >= 0 && gy < H && gx >= 0 && gx < W;
      const float* s3 = src + ((size_t)(ok ? gy : 0) * W + (ok ? gx : 0)) * 3;
      r0[it] = ok ? s3[0] : 0.f; r1[it] = ok ? s3[1] : 0.f; r2[it] = ok ? s3[2] : 0.f;
    }
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int p = t + 256 * it;
      if (p >= HH * HW_) break;
      unsigned pa[2], pb[2];
      S::split(r0[it], r1[it], S::act_scale(), pa);
      S::split(r2[it], 0.0f, S::act_scale(), pb);
      vmax = sat_track(sat_track(vmax, r0[it], r1[it]), r2[it], 0.0f);
      u32x4* d = (u32x4*)&tile[p * PSD];
#pragma unroll
      for (int j = 0; j < CIN / 4; ++j) d[j] = u32x4{0u, 0u, 0u, 0u};
      tile[p * PSD] = pa[0]; tile[p * PSD + 1] = pb[0];
      tile[p * PSD + CIN / 2] = pa[1]; tile[p * PSD + CIN / 2 + 1] = pb[1];
    }
  }
  sat_report(sat, vmax);
  __syncthreads();

  // ---- MFMA loop: wave wv owns rows wv * MR .. wv * MR + MR - 1 of the tile ----
  f32x16 acc[MR];
#pragma unroll
  for (int m = 0; m < MR; ++m)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[m][r] = 0.0f;
  u32x4 fnext[KS][2];
  if (!WPRE) {
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
#pragma unroll
      for (int pl = 0; pl < 2; ++pl) fnext[ks][pl] = wfrag[(((size_t)pl * (TAPS * KS) + ks) * 2 + half) * 32 + lx];
  }
#pragma unroll
  for (int tap = 0; tap < TAPS; ++tap) {
    const int dy = TAPS == 9 ? tap / 3 : 0, dx = TAPS == 9 ? tap % 3 : 0;
    u32x4 fcur[KS][2];
    if (!WPRE) {
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) { fcur[ks][0] = fnext[ks][0]; fcur[ks][1] = fnext[ks][1]; }
      if (tap + 1 < TAPS) {   // the next tap's fragments travel while this tap's MFMAs run
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
#pragma unroll
          for (int pl = 0; pl < 2; ++pl) fnext[ks][pl] = wfrag[(((size_t)pl * (TAPS * KS) + (tap + 1) * KS + ks) * 2 + half) * 32 + lx];
      }
    }
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const u32x4 fb[2] = {WPRE ? wreg[WPRE ? tap * KS + ks : 0][0] : fcur[ks][0], WPRE ? wreg[WPRE ? tap * KS + ks : 0][1] : fcur[ks][1]};
#pragma unroll
      for (int m = 0; m < MR; ++m) {
        const unsigned* ap = &tile[((wv * MR + m + dy) * HW_ + lx + dx) * PSD + ks * 8 + half * 4];
        const u32x4 fh = *(const u32x4*)ap, fl = *(const u32x4*)(ap + CIN / 2);
        const u32x4 fa[2] = {fh, fl};
#pragma unroll
        for (int tm = 0; tm < S::NT; ++tm) acc[m] = S::mma(fa[S::ta(tm)], fb[S::tb(tm)], acc[m]);
      }
    }
  }

  // ---- epilogue: lane = output channel lx; register r of row m = pixel column mfma_row(r, half).  Interior tiles (the
  // common case, workgroup-uniform) store without per-element bounds tests. ----
  const bool cok = lx < out_c;
  const float iv = inv_ch[lx], bv = (bias != nullptr && cok) ? bias[lx] : 0.0f;
  float s1 = 0.0f, s2 = 0.0f;
  const bool interior = ty0 + TH <= H && tx0 + 32 <= W;
  float* const dtile = out + (((size_t)b * H + ty0) * W + tx0) * out_c + lx;
  // (staging the outputs through LDS for 16-byte stores was measured: 411 -> 467 us per full-resolution layer — the two extra
  // barriers and the LDS round trip cost more than the 64-byte store pieces; the direct stores stay)
  if (interior) {
    if (cok) {
#pragma unroll
      for (int m = 0; m < MR; ++m) {
        float* drow = dtile + (size_t)(wv * MR + m) * W * out_c;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float v = acc[m][r] * iv + bv;
          drow[(unsigned)(mfma_row(r, half) * out_c)] = v;
          s1 += v; s2 += v * v;
        }
      }
    }
  } else {
#pragma unroll
    for (int m = 0; m < MR; ++m) {
      const int y = ty0 + wv * MR + m;
      float* drow = dtile + (size_t)(wv * MR + m) * W * out_c;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int col = mfma_row(r, half);
        const float v = acc[m][r] * iv + bv;
        if (cok && y < H && tx0 + col < W) {
          drow[(unsigned)(col * out_c)] = v;
          s1 += v; s2 += v * v;
        }
      }
    }
  }
  if (partial != nullptr) {
    double d1 = (double)s1, d2 = (double)s2;
    d1 = half_pair_sum(d1); d2 = half_pair_sum(d2);
    if (half == 0) { red[wv][lx][0] = d1; red[wv][lx][1] = d2; }
    __syncthreads();
    if (t < 32) {
      const double a1 = (red[0][t][0] + red[1][t][0]) + (red[2][t][0] + red[3][t][0]);
      const double a2 = (red[0][t][1] + red[1][t][1]) + (red[2][t][1] + red[3][t][1]);
      double* d = partial + (((size_t)b * n_wg + blockIdx.x) * 32 + t) * 2;
      d[0] = a1; d[1] = a2;
    }
  }
}

// BatchNorm statistics from the per-workgroup partials [b][n_wg][32][2]: one workgroup per (image, channel); 256 threads walk
// the partials with a fixed stride, then a fixed-order tree — deterministic.  alpha = gamma / sqrt(var + eps),
// beta = bias - mean * alpha (as al_bn_final_kernel).
__global__ __launch_bounds__(256) void al_bn_final_tiles_kernel(const double* __restrict__ partial, int n_wg, int n_pixels, int C,
                                                                const float* __restrict__ gamma, const float* __restrict__ beta_w,
                                                                float* __restrict__ alpha, float* __restrict__ beta) {
  __shared__ double red[256][2];
  const int t = threadIdx.x, c = blockIdx.x, b = blockIdx.y;
  double s = 0.0, q = 0.0;
  for (int k = t; k < n_wg; k += 256) {
    const double* d = partial + (((size_t)b * n_wg + k) * 32 + c) * 2;
    s += d[0]; q += d[1];
  }
  red[t][0] = s; red[t][1] = q;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (t < o) { red[t][0] += red[t + o][0]; red[t][1] += red[t + o][1]; }
    __syncthreads();
  }
  if (t == 0) {
    const double mean = red[0][0] / n_pixels;
    double var = red[0][1] / n_pixels - mean * mean;
    if (var < 0.0) var = 0.0;
    const float invstd = (float)(1.0 / sqrt(var + 1e-5));
    const float a = invstd * gamma[c];
    alpha[b * C + c] = a;
    beta[b * C + c] = beta_w[c] - (float)mean * a;
  }
}
// ---------------------------------------------------------------------------------------------------------------------
// Feature aggregation + score_head.0 (ALN:657-668) with both channel contractions on the matrix cores.
//   f1 = selu(conv1(x1))  (1x1, 16 -> 32)         D1[co][px] = W1^T[co][ci] x x1^T[ci][px]      1 MFMA step  (x 3 split terms)
//   s  = f1 . Ws0[0:32]   (1x1, 32 -> 8)          D2[k][px]  = Ws0^T[k][co] x f1[co][px]        2 MFMA steps (x 3)
//   s8 = selu(s + up2(q2) + up8(q3) + up32(q4))   q_g = f_g . Ws0[32g:32g+32] at the maps' own resolutions (al_assemble_proj_kernel)
// Lane = pixel in both products (the pixel is the N index), so NOTHING is transposed or staged: the B operand of the first
// product is the pixel's own 8 input channels (lane half h: channels 8h .. 8h+7, two float4 loads, split in registers); its
// result leaves lane (px, h) holding 16 of the pixel's 32 f1 channels — rows (r & 3) + 8 (r >> 2) + 4 h of the MFMA C layout —
// which, after SELU and a second split, ARE the B operand of the second product once the contraction index is DEFINED in that
// order: k-slot (step s, half h, j) <-> channel (j & 3) + 8 (2 s + (j >> 2)) + 4 h; the host lays Ws0^T out to match.  The
// second product leaves lane (px, h) with output channels 4h .. 4h+3, which is also how the three 8-channel up-sampled maps
// are shared between the two lanes of a pixel and how s8 is stored (16 B per lane, the wave writes 2 KB contiguously).
// 9 MFMAs per 32 pixels replace 768 of the ~1450 VALU instructions per pixel of the all-VALU kernel.
struct AsmIdx { int i0, i1; float l0, l1; };
__device__ __forceinline__ AsmIdx asm_up_index(int dst, int in_size, int out_size) {   // aliked.hip up_index (ATen upsample_bilinear2d, align_corners)
  const float scale = out_size > 1 ? (float)(in_size - 1) / (float)(out_size - 1) : 0.f;
  const float r = scale * (float)dst;
  AsmIdx u;
  u.i0 = (int)r;
  u.i1 = u.i0 + ((u.i0 < in_size - 1) ? 1 : 0);
  u.l1 = r - (float)u.i0;
  u.l0 = 1.f - u.l1;
  return u;
}
__global__ __launch_bounds__(256, 4) void al_assemble_x3_kernel(const float* __restrict__ x1, const float* __restrict__ q2,
                                                             const float* __restrict__ q3, const float* __restrict__ q4,
                                                             const u32x4* __restrict__ w1f, const float* __restrict__ w1inv,
                                                             const u32x4* __restrict__ w0f, float w0inv, float* __restrict__ s8, int Hp,
                                                             int Wp, unsigned* __restrict__ sat) {
  const int t = threadIdx.x, lane = t & 63, wv = t >> 6, lx = lane & 31, half = lane >> 5, b = blockIdx.y;
  const int npx = Hp * Wp;
  // constant operands: W1^T (A of product 1: lane = co, 8 input channels of half h) and Ws0^T (A of product 2: lane = k (8 live
  // rows of 32), k-slots in the order described above), two fp16 planes each
  u32x4 a1[2], a2[2][2];
#pragma unroll
  for (int pl = 0; pl < 2; ++pl) {
    a1[pl] = w1f[(pl * 2 + half) * 32 + lx];
#pragma unroll
    for (int st = 0; st < 2; ++st) a2[st][pl] = w0f[((pl * 2 + st) * 2 + half) * 32 + lx];
  }
  float inv1[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) inv1[r] = w1inv[mfma_row(r, half)];
  float vmax = 0.0f;
  {   // one 32-pixel M-tile per wave (two per wave cost 174 registers and half the resident waves: the kernel waits on loads)
    const int i = blockIdx.x * 128 + wv * 32 + lx;
    const bool ok = i < npx;
    const int ic = ok ? i : npx - 1;
    const float* src = x1 + ((size_t)b * npx + ic) * 16 + half * 8;
    const float4 va = *(const float4*)src, vb = *(const float4*)(src + 4);
    unsigned p[4][2];
    S::split(va.x, va.y, S::act_scale(), p[0]); S::split(va.z, va.w, S::act_scale(), p[1]);
    S::split(vb.x, vb.y, S::act_scale(), p[2]); S::split(vb.z, vb.w, S::act_scale(), p[3]);
    vmax = sat_track(sat_track(sat_track(sat_track(vmax, va.x, va.y), va.z, va.w), vb.x, vb.y), vb.z, vb.w);
    const u32x4 b1[2] = {u32x4{p[0][0], p[1][0], p[2][0], p[3][0]}, u32x4{p[0][1], p[1][1], p[2][1], p[3][1]}};
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
#pragma unroll
    for (int tm = 0; tm < S::NT; ++tm) acc = S::mma(a1[S::tb(tm)], b1[S::ta(tm)], acc);   // weights are the A operand here: the term order pairs (l, h), (h, l), (h, h)
    float f1[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) f1[r] = selu_x3(acc[r] * inv1[r]);
    unsigned pc[8][2];
#pragma unroll
    for (int j = 0; j < 8; ++j) { S::split(f1[2 * j], f1[2 * j + 1], S::act_scale(), pc[j]); vmax = sat_track(vmax, f1[2 * j], f1[2 * j + 1]); }
    f32x16 acc2;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc2[r] = 0.0f;
#pragma unroll
    for (int st = 0; st < 2; ++st) {
      const u32x4 b2[2] = {u32x4{pc[4 * st][0], pc[4 * st + 1][0], pc[4 * st + 2][0], pc[4 * st + 3][0]},
                           u32x4{pc[4 * st][1], pc[4 * st + 1][1], pc[4 * st + 2][1], pc[4 * st + 3][1]}};
#pragma unroll
      for (int tm = 0; tm < S::NT; ++tm) acc2 = S::mma(a2[st][S::tb(tm)], b2[S::ta(tm)], acc2);
    }
    // lane (px, h): output channels 4h .. 4h+3 = registers 0..3
    const int y = ic / Wp, x = ic - y * Wp;
    float part[3][4];
#pragma unroll
    for (int g = 0; g < 3; ++g) {
      const int fac = g == 0 ? 2 : (g == 1 ? 8 : 32);
      const float* map = g == 0 ? q2 : (g == 1 ? q3 : q4);
      const int h = Hp / fac, w = Wp / fac;
      const AsmIdx uy = asm_up_index(y, h, Hp), ux = asm_up_index(x, w, Wp);
      const float* base = map + (size_t)b * h * w * 8 + half * 4;
      const float4 a = *(const float4*)(base + ((size_t)uy.i0 * w + ux.i0) * 8), bq = *(const float4*)(base + ((size_t)uy.i0 * w + ux.i1) * 8);
      const float4 cq = *(const float4*)(base + ((size_t)uy.i1 * w + ux.i0) * 8), d = *(const float4*)(base + ((size_t)uy.i1 * w + ux.i1) * 8);
      part[g][0] = uy.l0 * (ux.l0 * a.x + ux.l1 * bq.x) + uy.l1 * (ux.l0 * cq.x + ux.l1 * d.x);
      part[g][1] = uy.l0 * (ux.l0 * a.y + ux.l1 * bq.y) + uy.l1 * (ux.l0 * cq.y + ux.l1 * d.y);
      part[g][2] = uy.l0 * (ux.l0 * a.z + ux.l1 * bq.z) + uy.l1 * (ux.l0 * cq.z + ux.l1 * d.z);
      part[g][3] = uy.l0 * (ux.l0 * a.w + ux.l1 * bq.w) + uy.l1 * (ux.l0 * cq.w + ux.l1 * d.w);
    }
    float o[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) o[k] = selu_x3((acc2[k] * w0inv + part[0][k]) + (part[1][k] + part[2][k]));
    if (ok) *(float4*)(s8 + ((size_t)b * npx + i) * 8 + half * 4) = make_float4(o[0], o[1], o[2], o[3]);
  }
  sat_report(sat, vmax);
}
}  // namespace

// Host: the two constant A operands of al_assemble_x3_kernel as fp16x3 fragments [plane][(step)][k-half][32 lanes][8].
//   w1 [16 ci][32 co] -> lane = co, k = ci (8 per half), scaled per co (inverse scales out: inv1[32] incl. the activation scale)
//   ws0 rows 0..31 [co][8] -> lane = k (rows 8..31 zero), k-slot (s, h, j) = channel (j & 3) + 8 (2 s + (j >> 2)) + 4 h, ONE scale
size_t al_assemble_x3_frag_halves() { return (size_t)(2 * 2 * 32 * 8) + (size_t)(2 * 2 * 2 * 32 * 8); }
void al_assemble_x3_prepare(const float* w1_ci_co, const float* ws0_co_k, unsigned short* frag, float* inv1, float* inv0) {
  auto split16 = [](float x, unsigned short* hb, unsigned short* lb) {
    const _Float16 hv = (_Float16)x;
    const _Float16 lv = (_Float16)(x - (float)hv);
    __builtin_memcpy(hb, &hv, 2); __builtin_memcpy(lb, &lv, 2);
  };
  unsigned short* f1 = frag;                       // [pl][half][32][8]
  unsigned short* f0 = frag + 2 * 2 * 32 * 8;      // [pl][step][half][32][8]
  for (int co = 0; co < 32; ++co) {
    float mx = 0.f;
    for (int ci = 0; ci < 16; ++ci) mx = fmaxf(mx, fabsf(w1_ci_co[ci * 32 + co]));
    int e = 0; float sc = 1.0f;
    if (mx > 0.f && mx < INFINITY) { frexpf(mx, &e); sc = ldexpf(1.0f, 14 - e); }
    inv1[co] = 1.0f / (sc * DIM_F16_ACT_SCALE);
    for (int ci = 0; ci < 16; ++ci) {
      const int hf = ci / 8, j = ci % 8;
      split16(w1_ci_co[ci * 32 + co] * sc, &f1[((0 * 2 + hf) * 32 + co) * 8 + j], &f1[((1 * 2 + hf) * 32 + co) * 8 + j]);
    }
  }
  float mx = 0.f;
  for (int i = 0; i < 32 * 8; ++i) mx = fmaxf(mx, fabsf(ws0_co_k[i]));
  int e = 0; float sc = 1.0f;
  if (mx > 0.f && mx < INFINITY) { frexpf(mx, &e); sc = ldexpf(1.0f, 14 - e); }
  *inv0 = 1.0f / (sc * DIM_F16_ACT_SCALE);
  for (size_t i = 0; i < (size_t)2 * 2 * 2 * 32 * 8; ++i) f0[i] = 0;
  for (int k = 0; k < 8; ++k)
    for (int st = 0; st < 2; ++st)
      for (int hf = 0; hf < 2; ++hf)
        for (int j = 0; j < 8; ++j) {
          const int co = (j & 3) + 8 * (2 * st + (j >> 2)) + 4 * hf;
          split16(ws0_co_k[co * 8 + k] * sc, &f0[(((0 * 2 + st) * 2 + hf) * 32 + k) * 8 + j], &f0[(((1 * 2 + st) * 2 + hf) * 32 + k) * 8 + j]);
        }
}

int launch_al_assemble_x3(const float* x1, const float* q2, const float* q3, const float* q4, const void* frag_dev, const float* inv1_dev, float inv0,
                          float* s8, int batch, int Hp, int Wp, hipStream_t s) {
  const u32x4* w1f = (const u32x4*)frag_dev;
  const u32x4* w0f = w1f + 2 * 2 * 32;
  hipLaunchKernelGGL(al_assemble_x3_kernel, dim3(cdiv(Hp * Wp, 128), batch), dim3(256), 0, s, x1, q2, q3, q4, w1f, inv1_dev, w0f, inv0, s8, Hp, Wp,
                     dim_sat_counter(DIM_SAT_ALIKED));
  DIM_LAUNCH_CHECK();
  return 0;
}

size_t al_convx3_partial_doubles(int batch, int H, int W) { return (size_t)batch * cdiv(H, 8) * cdiv(W, 32) * 32 * 2; }

// w: device fragments + inverse scales as split_weights(K = taps * cin_pad, N = cout, n_pad = 32, mode 2) lays them out.
int launch_al_convx3(const float* in, int in_c, int cin_pad, int taps, const SplitWeights& w, const float* bias, float* out, int cout,
                     int batch, int H, int W, double* partial, int* n_wg_out, const float* in_alpha, const float* in_beta, hipStream_t s) {
  DIM_REQUIRE((cin_pad == 16 || cin_pad == 32) && (taps == 1 || taps == 9) && cout <= 32 && w.n_pad == 32 && w.mode == 2,
              "aliked convx3: unsupported shape cin_pad %d taps %d cout %d", cin_pad, taps, cout);
  // dim_tune_set key 10: 16 (default) = 16-row tiles, weights in registers; 17 = 16-row tiles, weights streamed (3 workgroups per CU); 8 = 8-row tiles
  const int th = (cin_pad == 16 && taps == 9 && dim_aliked_tile_rows() >= 16) ? 16 : 8;
  const int tx = cdiv(W, 32), ty = cdiv(H, th);
  const dim3 grid(tx * ty, 1, batch);
  if (n_wg_out) *n_wg_out = tx * ty;
  unsigned* sat = dim_sat_counter(DIM_SAT_ALIKED);
  const u32x4* wf = (const u32x4*)w.dev;
#define AL_X3(CI, TP, ...) hipLaunchKernelGGL(HIP_KERNEL_NAME(al_convx3_kernel<CI, TP, __VA_ARGS__>), grid, dim3(256), 0, s, in, in_c, wf, w.inv_ch(), bias, out, cout, H, W, tx, partial, tx * ty, sat, in_alpha, in_beta)
  if (cin_pad == 16 && taps == 9 && th == 16 && dim_aliked_tile_rows() == 16) AL_X3(16, 9, 16);
  else if (cin_pad == 16 && taps == 9 && th == 16) AL_X3(16, 9, 16, false);
  else if (cin_pad == 16 && taps == 9) AL_X3(16, 9, 8);
  else if (cin_pad == 32 && taps == 9) AL_X3(32, 9, 8);
  else if (cin_pad == 16 && taps == 1) AL_X3(16, 1, 8);
  else AL_X3(32, 1, 8);
#undef AL_X3
  DIM_LAUNCH_CHECK();
  return 0;
}

int launch_al_bn_final_tiles(const double* partial, int n_wg, int batch, int n_pixels, int C, const float* gamma, const float* beta_w,
                             float* alpha, float* beta, hipStream_t s) {
  DIM_REQUIRE(C == 16 || C == 32, "aliked bn (tile partials): C=%d unsupported", C);
  hipLaunchKernelGGL(al_bn_final_tiles_kernel, dim3(C, batch), dim3(256), 0, s, partial, n_wg, n_pixels, C, gamma, beta_w, alpha, beta);
  DIM_LAUNCH_CHECK();
  return 0;
}
