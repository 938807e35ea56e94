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
__device__ __forceinline__ float selu_x3(float x) {   // ATen's elu kernel, as aliked.hip selu_
  const float scale = 1.0507009873554804934193349852946f, alpha = 1.6732632423543772848170429916717f;
  return x <= 0.0f ? (expf(x) - 1.0f) * (alpha * scale) : x * scale;
}

// TAPS: 9 (3x3, zero padding 1) or 1 (1x1).  CIN: padded input channels (16 or 32).  TH: tile rows (multiple of 4).
template <int CIN, int TAPS, int TH>
__global__ __launch_bounds__(256) void al_convx3_kernel(const float* __restrict__ in, int in_c, const u32x4* __restrict__ wfrag,
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

  // ---- stage the halo tile: fp32 -> two fp16 planes ----
  float vmax = 0.0f;
  if (in_c % 4 == 0) {
    constexpr int Q = CIN / 4;
    for (int i = t; i < HH * HW_ * Q; i += 256) {
      const int p = i / Q, q = i - p * Q;
      const int gy = ty0 + p / HW_ - R, gx = tx0 + p % HW_ - R;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (q * 4 < in_c && gy >= 0 && gy < H && gx >= 0 && gx < W) {
        v = *(const float4*)(src + ((size_t)gy * W + gx) * in_c + q * 4);
        if (in_alpha != nullptr) {  // the producer's BatchNorm + SELU applied on the way in (the padding stays zero: it pads the ACTIVATED map)
          const float4 a = *(const float4*)(in_alpha + b * in_c + q * 4), bb = *(const float4*)(in_beta + b * in_c + q * 4);
          v = make_float4(selu_x3(v.x * a.x + bb.x), selu_x3(v.y * a.y + bb.y), selu_x3(v.z * a.z + bb.z), selu_x3(v.w * a.w + bb.w));
        }
      }
      unsigned p0[2], p1[2];
      S::split(v.x, v.y, S::act_scale(), p0);
      S::split(v.z, v.w, S::act_scale(), p1);
      vmax = sat_track(sat_track(vmax, v.x, v.y), v.z, v.w);
      unsigned* d = &tile[p * PSD + q * 2];
      d[0] = p0[0]; d[1] = p1[0];
      d[CIN / 2] = p0[1]; d[CIN / 2 + 1] = p1[1];
    }
  } else {  // in_c = 3 (the RGB image): channel pairs, zero padded to CIN
    constexpr int Q = CIN / 2;
    for (int i = t; i < HH * HW_ * Q; i += 256) {
      const int p = i / Q, q = i - p * Q;
      const int gy = ty0 + p / HW_ - R, gx = tx0 + p % HW_ - R;
      float v0 = 0.f, v1 = 0.f;
      if (gy >= 0 && gy < H && gx >= 0 && gx < W) {
        const float* s3 = src + ((size_t)gy * W + gx) * in_c;
        if (2 * q < in_c) v0 = s3[2 * q];
        if (2 * q + 1 < in_c) v1 = s3[2 * q + 1];
      }
      unsigned pc[2];
      S::split(v0, v1, S::act_scale(), pc);
      vmax = sat_track(vmax, v0, v1);
      tile[p * PSD + q] = pc[0];
      tile[p * PSD + CIN / 2 + q] = pc[1];
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
#pragma unroll
  for (int tap = 0; tap < TAPS; ++tap) {
    const int dy = TAPS == 9 ? tap / 3 : 0, dx = TAPS == 9 ? tap % 3 : 0;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      u32x4 fb[2];
#pragma unroll
      for (int pl = 0; pl < 2; ++pl) fb[pl] = wfrag[(((size_t)pl * (TAPS * KS) + tap * KS + ks) * 2 + half) * 32 + lx];
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

  // ---- epilogue: lane = output channel lx; register r of row m = pixel column mfma_row(r, half) ----
  const bool cok = lx < out_c;
  const float iv = inv_ch[lx], bv = (bias != nullptr && cok) ? bias[lx] : 0.0f;
  float s1 = 0.0f, s2 = 0.0f;
#pragma unroll
  for (int m = 0; m < MR; ++m) {
    const int y = ty0 + wv * MR + m;
    float* drow = out + (((size_t)b * H + y) * W + tx0) * out_c + lx;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int col = mfma_row(r, half);
      const float v = acc[m][r] * iv + bv;
      if (cok && y < H && tx0 + col < W) {
        drow[(size_t)col * out_c] = v;
        s1 += v; s2 += v * v;
      }
    }
  }
  if (partial != nullptr) {
    double d1 = (double)s1, d2 = (double)s2;
    d1 += __shfl_xor(d1, 32); d2 += __shfl_xor(d2, 32);
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

// BatchNorm statistics from the per-workgroup partials [b][n_wg][32][2]: 256 / C lanes per channel walk the workgroups with a
// fixed stride, then a fixed-order sum — deterministic.  alpha = gamma / sqrt(var + eps), beta = bias - mean * alpha (as
// al_bn_final_kernel).
__global__ __launch_bounds__(256) void al_bn_final_tiles_kernel(const double* __restrict__ partial, int n_wg, int n_pixels, int C,
                                                                const float* __restrict__ gamma, const float* __restrict__ beta_w,
                                                                float* __restrict__ alpha, float* __restrict__ beta) {
  __shared__ double red[256][2];
  const int t = threadIdx.x, b = blockIdx.x;
  const int lanes_per_c = 256 / C, c = t % C, pl = t / C;
  double s = 0.0, q = 0.0;
  if (pl < lanes_per_c)
    for (int k = pl; k < n_wg; k += lanes_per_c) {
      const double* d = partial + (((size_t)b * n_wg + k) * 32 + c) * 2;
      s += d[0]; q += d[1];
    }
  red[t][0] = s; red[t][1] = q;
  __syncthreads();
  if (t < C) {
    for (int k = 1; k < lanes_per_c; ++k) { s += red[t + k * C][0]; q += red[t + k * C][1]; }
    const double mean = s / n_pixels;
    double var = q / n_pixels - mean * mean;
    if (var < 0.0) var = 0.0;
    const float invstd = (float)(1.0 / sqrt(var + 1e-5));
    const float a = invstd * gamma[t];
    alpha[b * C + t] = a;
    beta[b * C + t] = beta_w[t] - (float)mean * a;
  }
}
}  // namespace

size_t al_convx3_partial_doubles(int batch, int H, int W) { return (size_t)batch * cdiv(H, 8) * cdiv(W, 32) * 32 * 2; }

// w: device fragments + inverse scales as split_weights(K = taps * cin_pad, N = cout, n_pad = 32, mode 2) lays them out.
int launch_al_convx3(const float* in, int in_c, int cin_pad, int taps, const SplitWeights& w, const float* bias, float* out, int cout,
                     int batch, int H, int W, double* partial, int* n_wg_out, const float* in_alpha, const float* in_beta, hipStream_t s) {
  DIM_REQUIRE((cin_pad == 16 || cin_pad == 32) && (taps == 1 || taps == 9) && cout <= 32 && w.n_pad == 32 && w.mode == 2,
              "aliked convx3: unsupported shape cin_pad %d taps %d cout %d", cin_pad, taps, cout);
  const int th = (cin_pad == 16 && taps == 9) ? 16 : 8;
  const int tx = cdiv(W, 32), ty = cdiv(H, th);
  const dim3 grid(tx * ty, 1, batch);
  if (n_wg_out) *n_wg_out = tx * ty;
  unsigned* sat = dim_sat_counter(DIM_SAT_ALIKED);
  const u32x4* wf = (const u32x4*)w.dev;
#define AL_X3(CI, TP, TH_) hipLaunchKernelGGL(HIP_KERNEL_NAME(al_convx3_kernel<CI, TP, TH_>), grid, dim3(256), 0, s, in, in_c, wf, w.inv_ch(), bias, out, cout, H, W, tx, partial, tx * ty, sat, in_alpha, in_beta)
  if (cin_pad == 16 && taps == 9) AL_X3(16, 9, 16);
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
  hipLaunchKernelGGL(al_bn_final_tiles_kernel, dim3(batch), dim3(256), 0, s, partial, n_wg, n_pixels, C, gamma, beta_w, alpha, beta);
  DIM_LAUNCH_CHECK();
  return 0;
}
