// SuperPoint VGG convolutions for gfx950 (reference: SPN:128-143,161-171).
//
// conv3x3_mfma_kernel — 3x3/s1/p1 conv over NHWC fp32 as an implicit GEMM on
// v_mfma_f32_32x32x2_f32 (M = pixels, N = output channels, K = 9*cin):
//   * workgroup = 4 waves; output tile = 8 rows x 32 columns x 64 channels;
//     wave w owns tile rows 2w and 2w+1 (two 32-pixel MFMA row groups) x two
//     32-channel slabs = 4 accumulators.
//   * K loop: input channels in chunks of 16; per chunk the (8+2)x(32+2) halo
//     tile is staged HWC into LDS with a 17-dword pixel stride (the 32 lanes
//     that read one channel of 32 consecutive pixels hit 32 different banks)
//     and the 9x16x64 weight slice is staged with channel-contiguous rows.
//   * 59,984 B LDS/workgroup -> 2 workgroups per CU; staging of one overlaps
//     the 288 MFMAs/wave/chunk of the other.
//   * epilogue: bias + ReLU; the 2x2 max-pool is entirely in-lane because the
//     MFMA C layout keeps x, x+1 in adjacent registers and rows 2w, 2w+1 in the
//     same wave, so the pre-pool map (268 MB at 1024^2) is never written.
// conv1a_kernel — 1->64 channels, direct VALU conv writing NHWC.
#include "dim_kernels.h"

namespace {
constexpr int TH = 8, TW = 32, IW = TW + 2, IH = TH + 2;

// KC   : input channels per LDS chunk (16: 60 KB LDS, 288 MFMAs/wave between barriers;
//        8: 31 KB LDS, 144 MFMAs).
// PF   : software pipeline — the next chunk's halo tile and weight slice are fetched into
//        registers while the MFMAs of the current chunk run (affordable only at KC = 8:
//        32 staging VGPRs; at KC = 16 the 60 extra VGPRs halve the occupancy).
// OCC  : __launch_bounds__ minimum waves per SIMD.
template <int CIN, int POOL, int KC, int PF, int OCC>
__global__ __launch_bounds__(256, OCC) void conv3x3_mfma_kernel(const float* __restrict__ in, const float* __restrict__ w,
                                                                const float* __restrict__ bias, float* __restrict__ out,
                                                                int H, int W, int cout, int relu, int tiles_x) {
  constexpr int CS = KC + 1;
  constexpr int NIN_TOT = IH * IW * (KC / 4), NIN = (NIN_TOT + 255) / 256, NW_TOT = 9 * KC * 16, NW = (NW_TOT + 255) / 256;
  __shared__ float Is[IH * IW * CS];
  __shared__ float Ws[9 * KC * 64];

  const int t = threadIdx.x;
  const int lane = t & 63, wv = t >> 6, lx = lane & 31, half = lane >> 5;
  const int ty = blockIdx.x / tiles_x, tx = blockIdx.x % tiles_x;
  const int cb = blockIdx.y, b = blockIdx.z;
  const int oy = ty * TH, ox = tx * TW;
  const float* in_b = in + (size_t)b * H * W * CIN;

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

  float4 rin[NIN], rw[NW];
  auto load_chunk = [&](int c0) {
#pragma unroll
    for (int i = 0; i < NIN; ++i) {
      const int idx = t + 256 * i;
      const int p = idx / (KC / 4), q = idx % (KC / 4);
      const int py = p / IW, px = p - py * IW;
      const int gy = oy + py - 1, gx = ox + px - 1;
      rin[i] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (idx < NIN_TOT && gy >= 0 && gy < H && gx >= 0 && gx < W)
        rin[i] = *(const float4*)(in_b + ((size_t)gy * W + gx) * CIN + c0 + q * 4);
    }
#pragma unroll
    for (int i = 0; i < NW; ++i) {
      const int idx = t + 256 * i;
      const int rowi = idx >> 4, q = idx & 15;
      const int tap = rowi / KC, ci = rowi - tap * KC;
      if (NW_TOT % 256 == 0 || idx < NW_TOT) rw[i] = *(const float4*)(w + ((size_t)tap * CIN + c0 + ci) * cout + cb * 64 + q * 4);
    }
  };
  auto store_chunk = [&]() {
#pragma unroll
    for (int i = 0; i < NIN; ++i) {
      const int idx = t + 256 * i;
      if (idx < NIN_TOT) {
        float* d = &Is[(idx / (KC / 4)) * CS + (idx % (KC / 4)) * 4];
        d[0] = rin[i].x; d[1] = rin[i].y; d[2] = rin[i].z; d[3] = rin[i].w;
      }
    }
#pragma unroll
    for (int i = 0; i < NW; ++i) {
      const int idx = t + 256 * i;
      if (NW_TOT % 256 == 0 || idx < NW_TOT) *(float4*)&Ws[(idx >> 4) * 64 + (idx & 15) * 4] = rw[i];
    }
  };

  if (PF) load_chunk(0);
  for (int c0 = 0; c0 < CIN; c0 += KC) {
    if (!PF) load_chunk(c0);
    store_chunk();
    __syncthreads();
    if (PF && c0 + KC < CIN) load_chunk(c0 + KC);
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      const int dy = tap / 3, dx = tap - 3 * (tap / 3);
      const float* i0 = &Is[((2 * wv + dy) * IW + lx + dx) * CS + half];
      const float* i1 = i0 + IW * CS;
      const float* wp = &Ws[(tap * KC + half) * 64 + lx];
#pragma unroll
      for (int s = 0; s < KC / 2; ++s) {
        const float a0 = i0[2 * s], a1 = i1[2 * s];
        const float b0 = wp[2 * s * 64], b1 = wp[2 * s * 64 + 32];
        acc[0][0] = mfma32(a0, b0, acc[0][0]);
        acc[0][1] = mfma32(a0, b1, acc[0][1]);
        acc[1][0] = mfma32(a1, b0, acc[1][0]);
        acc[1][1] = mfma32(a1, b1, acc[1][1]);
      }
    }
    __syncthreads();
  }

#pragma unroll
  for (int n = 0; n < 2; ++n) {
    const int co = cb * 64 + n * 32 + lx;
    const float bv = bias[co];
    if (POOL) {
      const int Ho = H >> 1, Wo = W >> 1;
      const int py = (oy >> 1) + wv;
      float* out_b = out + (size_t)b * Ho * Wo * cout;
#pragma unroll
      for (int r = 0; r < 16; r += 2) {
        const int px = (ox + mfma_row(r, half)) >> 1;
        float v = fmaxf(fmaxf(acc[0][n][r], acc[0][n][r + 1]), fmaxf(acc[1][n][r], acc[1][n][r + 1])) + bv;
        if (relu) v = fmaxf(v, 0.0f);
        if (py < Ho && px < Wo) out_b[((size_t)py * Wo + px) * cout + co] = v;
      }
    } else {
      float* out_b = out + (size_t)b * H * W * cout;
#pragma unroll
      for (int m = 0; m < 2; ++m) {
        const int y = oy + 2 * wv + m;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int x = ox + mfma_row(r, half);
          float v = acc[m][n][r] + bv;
          if (relu) v = fmaxf(v, 0.0f);
          if (y < H && x < W) out_b[((size_t)y * W + x) * cout + co] = v;
        }
      }
    }
  }
}

// 16 lanes cover the 64 output channels of one pixel (4 each, one float4 store),
// so a wave writes 4 consecutive pixels = 1 KiB contiguous.
__global__ __launch_bounds__(256) void conv1a_kernel(const float* __restrict__ in, const float* __restrict__ w,
                                                     const float* __restrict__ bias, float* __restrict__ out, int H,
                                                     int W) {
  const int t = threadIdx.x, cq = t & 15, pl = t >> 4;
  const int y = blockIdx.y, b = blockIdx.z;
  const float* img = in + (size_t)b * H * W;
  float wr[9][4];
#pragma unroll
  for (int k = 0; k < 9; ++k) {
    const float4 v = *(const float4*)(w + k * 64 + cq * 4);
    wr[k][0] = v.x; wr[k][1] = v.y; wr[k][2] = v.z; wr[k][3] = v.w;
  }
  const float4 bv = *(const float4*)(bias + cq * 4);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int x = blockIdx.x * 64 + i * 16 + pl;
    if (x >= W) continue;
    float o0 = 0.f, o1 = 0.f, o2 = 0.f, o3 = 0.f;
#pragma unroll
    for (int k = 0; k < 9; ++k) {
      const int yy = y + k / 3 - 1, xx = x + k % 3 - 1;
      const float p = (yy >= 0 && yy < H && xx >= 0 && xx < W) ? img[(size_t)yy * W + xx] : 0.0f;
      o0 = fmaf(p, wr[k][0], o0); o1 = fmaf(p, wr[k][1], o1);
      o2 = fmaf(p, wr[k][2], o2); o3 = fmaf(p, wr[k][3], o3);
    }
    float4 o = make_float4(fmaxf(o0 + bv.x, 0.f), fmaxf(o1 + bv.y, 0.f), fmaxf(o2 + bv.z, 0.f), fmaxf(o3 + bv.w, 0.f));
    *(float4*)(out + (((size_t)b * H + y) * W + x) * 64 + cq * 4) = o;
  }
}
}  // namespace

static int g_conv_variant = 3;
void dim_conv_set_variant(int v) { g_conv_variant = v; }

int launch_conv3x3(const float* in, const float* w, const float* bias, float* out, int batch, int H, int W, int cin,
                   int cout, int pool, int relu, hipStream_t s) {
  DIM_REQUIRE(cout % 64 == 0, "conv3x3: cout=%d must be a multiple of 64", cout);
  DIM_REQUIRE(cin == 64 || cin == 128, "conv3x3: cin=%d unsupported (64 or 128)", cin);
  if (batch <= 0 || H <= 0 || W <= 0) return 0;
  const int tiles_x = cdiv(W, TW), tiles_y = cdiv(H, TH);
  dim3 grid(tiles_x * tiles_y, cout / 64, batch);
#define DIM_CONV(CI, P, KCV, PFV, OC) \
  hipLaunchKernelGGL(HIP_KERNEL_NAME(conv3x3_mfma_kernel<CI, P, KCV, PFV, OC>), grid, dim3(256), 0, s, in, w, bias, out, H, W, cout, relu, tiles_x)
#define DIM_CONV_V(KCV, PFV, OC)               \
  {                                            \
    if (cin == 64 && pool) DIM_CONV(64, 1, KCV, PFV, OC);  \
    else if (cin == 64) DIM_CONV(64, 0, KCV, PFV, OC);     \
    else if (pool) DIM_CONV(128, 1, KCV, PFV, OC);         \
    else DIM_CONV(128, 0, KCV, PFV, OC);                   \
  }
  switch (g_conv_variant) {  // measured r01 (TFLOP/s, conv1b / conv4a): v0 117/111, v1 121/116, v2 118/105, v3 122/122
    case 1: DIM_CONV_V(8, 0, 3) break;
    case 2: DIM_CONV_V(8, 1, 2) break;
    case 4: DIM_CONV_V(16, 0, 1) break;
    default: DIM_CONV_V(8, 1, 3) break;
  }
#undef DIM_CONV_V
#undef DIM_CONV
  DIM_LAUNCH_CHECK();
  return 0;
}

int launch_conv1a(const float* in, const float* w, const float* bias, float* out, int batch, int H, int W,
                  hipStream_t s) {
  if (batch <= 0 || H <= 0 || W <= 0) return 0;
  dim3 grid(cdiv(W, 64), H, batch);
  hipLaunchKernelGGL(conv1a_kernel, grid, dim3(256), 0, s, in, w, bias, out, H, W);
  DIM_LAUNCH_CHECK();
  return 0;
}
