// 3x3 convolution as a 1-D Winograd F(2, 3) along x on the fp16 matrix cores (fp16x3 split arithmetic, dim_common.h): a pair of
// neighbouring output columns (2j, 2j+1) of one row and kernel row dy needs the four input columns d0..d3 = 2j-1 .. 2j+2:
//     t0 = d0 - d2,  t1 = d1 + d2,  t2 = d2 - d1,  t3 = d1 - d3                 (input transform, fp32, BEFORE the fp16 split)
//     u0 = g0,  u1 = (g0 + g1 + g2) / 2,  u2 = (g0 - g1 + g2) / 2,  u3 = g2     (weight transform, host, fp64 -> fp32 -> split)
//     m_p = sum over (cin, dy) of t_p u_p;    y(2j) = m0 + m1 + m2,  y(2j+1) = m1 - m2 - m3
// = 4 instead of 6 multiplies per output pair, input channel and kernel row: 2/3 of the direct kernel's MFMAs (conv_x6.hip).
// Replaces the same reference lines as conv_x6.hip (SuperPoint's 3x3 convolutions, SPN:161-175); selected by dim_tune_set(15, mask).
//
// Implicit GEMM per position p: M = (row, column pair) of the tile, N = cout, K = 3 dy x cin.  Workgroup = 4 waves, output tile
// 8 rows x 32 columns x 64 channels; WAVE p OWNS POSITION p: 4 M-tiles (2 rows x 16 pairs each) x 2 N-tiles = 8 accumulators.
//   * B operand (transformed weights of position p) goes L2 -> registers in MFMA-fragment order, one contiguous 4-KB run per wave and
//     (chunk, dy) step, requested one step ahead: every fragment is fetched by exactly one wave, no weight image in LDS, no barrier
//     per kernel row;
//   * A operand: the transformed + split halo tile of one 16-channel chunk in LDS, Ip[plane][k-half][pos][row 10][pair 16] 16-byte
//     slots (one ds_read_b128 = one MFMA operand; a wave reads 8 per 24 MFMAs where the direct kernel reads 12);
//   * F1A staging (conv1b): SuperPoint's conv1a (1 -> 64, SPN:161) is evaluated from the image patch for the FOUR columns of a pair's
//     tuple (same fmaf chain as conv1a_kernel / conv_x6.hip), transformed, split and stored — the 64-channel conv1a map never exists;
//   * the four positions of an output live in four waves: after the K loop every wave hands three of its M-tiles over through LDS
//     (the Ip bytes) and finishes one (rows 2w, 2w+1): output transform, 2x2 max-pool in-lane, bias, ReLU, split, store.
// fp16x3 range guard: the transformed activations reach 2 max|d| (t1 = d1 + d2); the staging tracks max|t| of what it splits and
// reports it at the consumer's site, the epilogue guards its outputs as conv_x6.hip does.
#ifdef DIM_RESEARCH   // Winograd F(2,3) conv1b: measured at parity with the direct kernel -> research build only
#include <math.h>
#include <string.h>

#include <type_traits>

#include "dim_kernels.h"

namespace {
constexpr int WG_TW = 32, WG_NP = WG_TW / 2;
constexpr int WG_IMW = WG_TW + 4;                 // image patch of the fused conv1a: halo of the halo (36 columns)
// MT = M-tiles per position = tile rows / 2 (template parameter): 4 = 8-row tiles (halo 10 rows = 2.5 staging rounds per wave),
// 3 = 6-row tiles (8 halo rows = 2 rounds exactly, 6 accumulators, 3 workgroups per CU; wave 3 finishes no M-tile)
constexpr int wg_th(int mt) { return 2 * mt; }
constexpr int wg_rows(int mt) { return (wg_th(mt) + 2) * WG_NP; }     // (row, pair) slots of one (plane, k-half, position)
// slots of one (plane, k-half) block, padded so that the block stride is 16 banks (mod 32): a ds_write_b64 lane group (16 lanes = 4 pairs x
// 4 channel quads) touches both k-halves, and LDS stores are banked (a / 4) mod 32 (MI355X_MICROARCH.md, LDS) — measured before the
// padding: SQ_LDS_BANK_CONFLICT = 18 % of the LDS cycles, all of it the staging stores
constexpr int wg_kh(int mt) { return 4 * wg_rows(mt) + ((4 * wg_rows(mt)) % 8 == 0 ? 4 : 0); }
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
__host__ __device__ constexpr size_t planes_image_pixels_wg(int h, int w) { return ((size_t)h * w + 1) & ~(size_t)1; }  // pixel slots of one pre-split image (conv_x6.hip)
constexpr int WG_WFRAG = 2 * 2 * 2 * 32 * 8;      // 16-bit elements per (cout block, chunk, dy, position): [plane][n][k-half][32 co][8 ci] = 4 KB

__device__ unsigned long long g_wg_phase[16];   // TIMING builds: summed s_memtime deltas of wave 0 per phase (dim_conv_wg_phase_read)

template <int CIN, int POOL, bool POUT, int MT, int PROBE = 0, int STG = 0, bool TIMING = false, int NTILE = 1>
__global__ __launch_bounds__(256, 2) void conv3x3_wg_f1a_kernel(const float* __restrict__ image, const unsigned short* __restrict__ wx,
                                                               const float* __restrict__ bias, float* __restrict__ out, int H, int W, int cout,
                                                               int tiles_x, const float* __restrict__ w1a, const float* __restrict__ b1a,
                                                               const float* __restrict__ inv_ch, unsigned* sat, unsigned* sat_image, int stagger, int n_tiles) {
  static_assert(CIN == 64, "the fused conv1a produces 64 channels");
  using S = SplitMma<2>;
  constexpr int NCHUNK = CIN / 16, NSTEP = NCHUNK * 3;
  constexpr int WG_TH = wg_th(MT), WG_IH = WG_TH + 2, WG_ROWS = wg_rows(MT), WG_KH = wg_kh(MT), WG_IMH = WG_TH + 4;
  __shared__ u32x4 Ip[4 * WG_KH];
  __shared__ float Img[WG_IMH * WG_IMW];
  __shared__ float W1a[9 * 64 + 64];
  // STG 1: conv1a's outputs of one chunk for the halo tile, fp32 [pixel (IH x 34)][16 channels + 4 pad] (80-byte pixel stride:
  // 16-byte aligned float4 reads, neighbouring pixel pairs 40 banks apart)
  constexpr int SPS = 20, NHP = WG_IH * (WG_TW + 2);
  __shared__ float Sx[STG == 1 ? NHP * SPS : 1];

  const int t = threadIdx.x;
  const int lane = t & 63, lx = lane & 31, half = lane >> 5;
  const int wv = __builtin_amdgcn_readfirstlane(t >> 6);   // = this wave's Winograd position
  int tile0;   // first tile of this workgroup (NTILE consecutive tiles of the XCD-banded order: a persistent workgroup keeps the conv1a
               // weights in LDS and fetches the next tile's image patch behind the current tile's work — the per-tile start-up chain
               // global load -> LDS -> barrier -> phase 1 -> barrier was 22 % of a workgroup's time in the phase-timer build)
  {  // XCD-aware order (conv_x6.hip): every XCD works on a contiguous band of the image's tile groups
    const int nt = gridDim.x, xcd = blockIdx.x & 7, j = blockIdx.x >> 3, q = nt >> 3, r = nt & 7;
    tile0 = (xcd * q + min(xcd, r) + j) * NTILE;
  }
  const int cb = blockIdx.y, b = blockIdx.z;
  int oy = (tile0 / tiles_x) * WG_TH, ox = (tile0 % tiles_x) * WG_TW;
  const float* in_b = image + (size_t)b * H * W;

  unsigned long long tph[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tlast = TIMING ? __builtin_readcyclecounter() : 0ull;
  const unsigned long long tstart = tlast;
  auto tick = [&](int ph) {   // TIMING: charge the cycles since the previous tick to phase ph
    if (TIMING) { const unsigned long long now = __builtin_readcyclecounter(); tph[ph] += now - tlast; tlast = now; }
  };
  f32x16 acc[MT][2];
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int n = 0; n < 2; ++n)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.0f;

  // image patch (rows oy - 2 .. oy + TH + 1, columns ox - 2 .. ox + 33; zero outside the image = conv1a's padding): global -> registers
  // (load_img, issued one tile ahead) -> LDS (put_img, with the range guard on the values)
  constexpr int NIMG = (WG_IMH * WG_IMW + 255) / 256;
  float pimg[NIMG];
  auto load_img = [&](int toy, int tox) {
#pragma unroll
    for (int i = 0; i < NIMG; ++i) {
      const int idx = t + 256 * i, r = idx / WG_IMW, cc = idx - r * WG_IMW;
      const int gy = toy + r - 2, gx = tox + cc - 2;
      pimg[i] = (idx < WG_IMH * WG_IMW && gy >= 0 && gy < H && gx >= 0 && gx < W) ? in_b[(size_t)gy * W + gx] : 0.0f;
    }
  };
  auto put_img = [&]() {
    unsigned imax = 0u;
#pragma unroll
    for (int i = 0; i < NIMG; ++i) {
      const int idx = t + 256 * i;
      if (idx < WG_IMH * WG_IMW) Img[idx] = pimg[i];
      imax = max(imax, __float_as_uint(pimg[i]) & 0x7fffffffu);
    }
    if (sat_image != nullptr && imax > 0x3f800000u) atomicAdd(sat_image, 1u);
  };
  load_img(oy, ox);
  // conv1a weights [tap][64] + bias [64], pre-multiplied by the activation scale (a power of two: exact); once per workgroup
  for (int idx = t; idx < 9 * 64 + 64; idx += 256) W1a[idx] = (idx < 9 * 64 ? w1a[idx] : b1a[idx - 9 * 64]) * S::act_scale();

  // B operand: transformed weights of position wv for step s = chunk * 3 + dy, one step ahead of the MFMAs that use them
  u32x4 bw[2][2][2];   // [buffer][plane][n]
  auto load_b = [&](int step, auto buf_t) {
    constexpr int BUF = decltype(buf_t)::value;
    const u32x4* src = (const u32x4*)(wx + (((size_t)cb * NSTEP + step) * 4 + wv) * WG_WFRAG);
#pragma unroll
    for (int pl = 0; pl < 2; ++pl)
#pragma unroll
      for (int n = 0; n < 2; ++n) bw[BUF][pl][n] = src[(pl * 2 + n) * 64 + lane];
  };
  load_b(0, std::integral_constant<int, 0>{});

  // staging items: (halo row r, column pair j, channel quad q) -> conv1a at the tuple's four columns, transform, split, store.
  // 10 x 16 x 4 = 640 items = 2.5 per thread: q and j are fixed per thread, wave w takes rows w, w + 4 and (waves 0, 1) w + 8.
  const int q = t & 3, j = (t >> 2) & 15;
  // does the halo of this tile leave the image?  (wave-uniform: the zero-padding masks cost VALU only on border tiles)
  bool border = false;   // set per tile
  float vmax_in = 0.0f;   // range guard on the transformed activations (scaled units)
  auto stage = [&](int c) {
    float wr[9][4], bv[4];
#pragma unroll
    for (int k = 0; k < 9; ++k) {
      const float4 v = *(const float4*)&W1a[k * 64 + c * 16 + q * 4];
      wr[k][0] = v.x; wr[k][1] = v.y; wr[k][2] = v.z; wr[k][3] = v.w;
    }
    {
      const float4 v = *(const float4*)&W1a[9 * 64 + c * 16 + q * 4];
      bv[0] = v.x; bv[1] = v.y; bv[2] = v.z; bv[3] = v.w;
    }
#pragma unroll
    for (int i = 0; i < (WG_IH + 3) / 4; ++i) {
      const int r = wv + 4 * i;
      if (r >= WG_IH) break;   // wave-uniform
      float v[3][6];
#pragma unroll
      for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int k = 0; k < 3; ++k) {
          const float2 p2 = *(const float2*)&Img[(r + a) * WG_IMW + 2 * j + 2 * k];
          v[a][2 * k] = p2.x; v[a][2 * k + 1] = p2.y;
        }
      float d[4][4];   // [column of the tuple][channel]
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        float o[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int tap = 0; tap < (PROBE == 1 ? 1 : 9); ++tap) {   // PROBE 1 (timing only, wrong results): one tap instead of nine
          const float x = v[tap / 3][k + tap % 3];
#pragma unroll
          for (int e = 0; e < 4; ++e) o[e] = fmaf(x, wr[tap][e], o[e]);
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) d[k][e] = fmaxf(o[e] + bv[e], 0.f);
      }
      if (border) {  // conv1b's zero padding: halo pixels outside the image are 0, not relu(conv1a of the padded image)
        const int gy = oy - 1 + r;
        const bool rowok = gy >= 0 && gy < H;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const int gx = ox - 1 + 2 * j + k;
          const bool ok = rowok && gx >= 0 && gx < W;
#pragma unroll
          for (int e = 0; e < 4; ++e) d[k][e] = ok ? d[k][e] : 0.f;
        }
      }
      float tp[4][4];  // [position][channel]
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        tp[0][e] = d[0][e] - d[2][e];
        tp[1][e] = d[1][e] + d[2][e];
        tp[2][e] = d[2][e] - d[1][e];
        tp[3][e] = d[1][e] - d[3][e];
      }
#pragma unroll
      for (int p = 0; p < 4; ++p) {
        vmax_in = fmaxf(vmax_in, fmaxf(fmaxf(fabsf(tp[p][0]), fabsf(tp[p][1])), fmaxf(fabsf(tp[p][2]), fabsf(tp[p][3]))));
        unsigned h01, l01, h23, l23;
        split2_pk_raw(tp[p][0], tp[p][1], h01, l01);
        split2_pk_raw(tp[p][2], tp[p][3], h23, l23);
        // channels q*4 .. q*4+3 live in k-half q >> 1, dwords (q & 1) * 2, + 1 of the (row, pair) slot
        unsigned* dst = (unsigned*)&Ip[(q >> 1) * WG_KH + p * WG_ROWS + r * WG_NP + j] + (q & 1) * 2;
        *(u32x2*)dst = u32x2{h01, h23};
        *(u32x2*)(dst + 2 * WG_KH * 4) = u32x2{l01, l23};   // plane 1 = the low pieces
      }
    }
  };

  // ---- STG 1: two-phase staging.  Phase 1: conv1a ONCE per halo pixel (the fused form evaluates every column twice: the tuples of
  // neighbouring pairs overlap) on packed fp32 FMAs (v_pk_fma_f32: two channels per instruction; every VALU instruction of this kernel
  // costs ~4 cycles of a SIMD that cannot issue MFMAs meanwhile — measured: 8/9 of the conv1a FMAs removed = -16 % kernel time), into Sx.
  // Phase 2: per (row, pair, channel quad) four float4 reads, input transform, split (v_fma_mix residuals), store. ----
  constexpr int N1 = (NHP * 4 + 255) / 256;
  int s1_img[STG == 1 ? N1 : 1];     // Img offset of the item's pixel, -1 = no such item
  bool s1_ok[STG == 1 ? N1 : 1];     // pixel inside the image (conv1b's zero padding otherwise)
  auto setup_tile = [&]() {   // per tile: border flag, phase-1 item tables
    border = oy == 0 || oy + WG_TH >= H || ox == 0 || ox + WG_TW >= W;
    if (STG == 1) {
#pragma unroll
      for (int i = 0; i < N1; ++i) {
        const int idx = t + 256 * i, p = idx >> 2;
        const int py = p / (WG_TW + 2), px = p - py * (WG_TW + 2);
        const int gy = oy + py - 1, gx = ox + px - 1;
        s1_img[i] = idx < NHP * 4 ? py * WG_IMW + px : -1;
        s1_ok[i] = gy >= 0 && gy < H && gx >= 0 && gx < W;
      }
    }
  };
  auto stage1 = [&](int c) {
    f32x2 wr2[9][2], bv2[2];
#pragma unroll
    for (int k = 0; k < 9; ++k) {
      const float4 v = *(const float4*)&W1a[k * 64 + c * 16 + q * 4];
      wr2[k][0] = f32x2{v.x, v.y}; wr2[k][1] = f32x2{v.z, v.w};
    }
    {
      const float4 v = *(const float4*)&W1a[9 * 64 + c * 16 + q * 4];
      bv2[0] = f32x2{v.x, v.y}; bv2[1] = f32x2{v.z, v.w};
    }
#pragma unroll
    for (int i = 0; i < N1; ++i) {
      const int io = s1_img[i];
      if (io < 0) continue;
      f32x2 o0 = {0.f, 0.f}, o1 = {0.f, 0.f};
#pragma unroll
      for (int tap = 0; tap < (PROBE == 1 ? 1 : 9); ++tap) {
        const float x = Img[io + (tap / 3) * WG_IMW + tap % 3];
        const f32x2 x2 = {x, x};
        o0 = __builtin_elementwise_fma(x2, wr2[tap][0], o0);
        o1 = __builtin_elementwise_fma(x2, wr2[tap][1], o1);
      }
      o0 = o0 + bv2[0]; o1 = o1 + bv2[1];
      float4 d = make_float4(fmaxf(o0[0], 0.f), fmaxf(o0[1], 0.f), fmaxf(o1[0], 0.f), fmaxf(o1[1], 0.f));
      if (border && !s1_ok[i]) d = make_float4(0.f, 0.f, 0.f, 0.f);
      const int p = (t + 256 * i) >> 2;
      *(float4*)&Sx[p * SPS + q * 4] = d;
    }
  };
  auto stage2 = [&](int c) {
    (void)c;
#pragma unroll
    for (int i = 0; i < (WG_IH + 3) / 4; ++i) {
      const int r = wv + 4 * i;
      if (r >= WG_IH) break;   // wave-uniform
      float4 d[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) d[k] = *(const float4*)&Sx[(r * (WG_TW + 2) + 2 * j + k) * SPS + q * 4];
      const float dd[4][4] = {{d[0].x, d[0].y, d[0].z, d[0].w}, {d[1].x, d[1].y, d[1].z, d[1].w}, {d[2].x, d[2].y, d[2].z, d[2].w}, {d[3].x, d[3].y, d[3].z, d[3].w}};
      float tp[4][4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        tp[0][e] = dd[0][e] - dd[2][e];
        tp[1][e] = dd[1][e] + dd[2][e];
        tp[2][e] = dd[2][e] - dd[1][e];
        tp[3][e] = dd[1][e] - dd[3][e];
      }
#pragma unroll
      for (int p = 0; p < 4; ++p) {
        vmax_in = fmaxf(vmax_in, fmaxf(fmaxf(fabsf(tp[p][0]), fabsf(tp[p][1])), fmaxf(fabsf(tp[p][2]), fabsf(tp[p][3]))));
        unsigned h01, l01, h23, l23;
        split2_pk_raw(tp[p][0], tp[p][1], h01, l01);
        split2_pk_raw(tp[p][2], tp[p][3], h23, l23);
        unsigned* dst = (unsigned*)&Ip[(q >> 1) * WG_KH + p * WG_ROWS + r * WG_NP + j] + (q & 1) * 2;
        *(u32x2*)dst = u32x2{h01, h23};
        *(u32x2*)(dst + 2 * WG_KH * 4) = u32x2{l01, l23};
      }
    }
  };

  auto mma_chunk = [&](int c, auto par_t) {
    constexpr int CP = decltype(par_t)::value;   // chunk parity: the buffer of step c * 3 + dy is (CP + dy) & 1
#pragma unroll
    for (int dy = 0; dy < 3; ++dy) {
      constexpr int dummy = 0; (void)dummy;
      const int step = c * 3 + dy;
      if (PROBE == 2) {   // timing probe (wrong results): the weight fragments of step 0 / 1 serve every step — no L2 round trip per step
        if (step == 0) load_b(1, std::integral_constant<int, 1>{});
      } else if ((CP + dy) & 1) { if (step + 1 < NSTEP) load_b(step + 1, std::integral_constant<int, 0>{}); }
      else { if (step + 1 < NSTEP) load_b(step + 1, std::integral_constant<int, 1>{}); }
      u32x4 fa[MT][2];
#pragma unroll
      for (int pl = 0; pl < 2; ++pl)
#pragma unroll
        for (int m = 0; m < MT; ++m) fa[m][pl] = Ip[(pl * 2 + half) * WG_KH + wv * WG_ROWS + (2 * m + dy + (lx >> 4)) * WG_NP + (lx & 15)];
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int tm = 0; tm < S::NT; ++tm)  // smallest cross terms first
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
          for (int n = 0; n < 2; ++n) acc[m][n] = S::mma(fa[m][S::ta(tm)], bw[(CP + dy) & 1][S::tb(tm)][n], acc[m][n]);
      __builtin_amdgcn_s_setprio(0);
    }
  };

  (void)stagger;
  for (int it = 0; it < NTILE; ++it) {
  const int tile = tile0 + it;
  if (tile >= n_tiles) break;   // workgroup-uniform
  oy = (tile / tiles_x) * WG_TH; ox = (tile % tiles_x) * WG_TW;
  setup_tile();
  if (it > 0) {
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int n = 0; n < 2; ++n)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.0f;
  }
  put_img();   // (every wave has left the previous tile's phase 1 long ago: the exchange barriers lie in between)
  __syncthreads();   // Img / W1a complete
  if (NTILE > 1 && it + 1 < NTILE && tile + 1 < n_tiles) load_img(((tile + 1) / tiles_x) * WG_TH, ((tile + 1) % tiles_x) * WG_TW);
  tick(0);
  if (STG == 1) {
    // phase 1 of chunk c + 1 shares a barrier interval with the MFMAs of chunk c (Sx is free once phase 2 has read it): 2 barriers per chunk
    stage1(0);
    __syncthreads();
    tick(0);
    for (int c = 0; c < NCHUNK; c += 2) {
      stage2(c); tick(1);
      __syncthreads(); tick(2);
      mma_chunk(c, std::integral_constant<int, 0>{}); tick(3);
      stage1(c + 1); tick(4);
      __syncthreads(); tick(5);
      stage2(c + 1); tick(1);
      __syncthreads(); tick(2);
      mma_chunk(c + 1, std::integral_constant<int, 1>{}); tick(3);
      if (c + 2 < NCHUNK) stage1(c + 2);
      tick(4);
      __syncthreads(); tick(5);
    }
  } else
  for (int c = 0; c < NCHUNK; c += 2) {
    stage(c); tick(1);
    __syncthreads(); tick(2);
    mma_chunk(c, std::integral_constant<int, 0>{}); tick(3);
    __syncthreads(); tick(5);   // every wave has read chunk c's operands
    stage(c + 1); tick(1);
    __syncthreads(); tick(2);
    mma_chunk(c + 1, std::integral_constant<int, 1>{}); tick(3);
    __syncthreads(); tick(5);
  }

  // ---- output transform across the four waves: wave w finishes M-tile w (tile rows 2w, 2w + 1) ----
  // round k: wave p hands M-tile (p + k) & 3 over (both N-tiles, 8 KB) and picks up M-tile w from wave (w - k) & 3
  f32x4* X = (f32x4*)Ip;   // [wave 4][n 2][register quad 4][lane 64] = 32 KB of the 40.5 KB operand image
  f32x16 y0[2], y1[2];
  auto coef0 = [](int p) { return p == 3 ? 0.0f : 1.0f; };                         // y(2j)   = m0 + m1 + m2
  auto coef1 = [](int p) { return p == 0 ? 0.0f : (p == 1 ? 1.0f : -1.0f); };      // y(2j+1) = m1 - m2 - m3
  {
    const float a0 = coef0(wv), a1 = coef1(wv);
#pragma unroll
    for (int m = 0; m < MT; ++m)
      if (m == wv) {
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
          for (int r = 0; r < 16; ++r) { y0[n][r] = a0 * acc[m][n][r]; y1[n][r] = a1 * acc[m][n][r]; }
      }
  }
#pragma unroll
  for (int k = 1; k < 4; ++k) {
    const int dst = (wv + k) & 3, src = (wv + 4 - k) & 3;
#pragma unroll
    for (int m = 0; m < MT; ++m)
      if (m == dst) {
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            f32x4 v4 = {acc[m][n][4 * g], acc[m][n][4 * g + 1], acc[m][n][4 * g + 2], acc[m][n][4 * g + 3]};
            X[((wv * 2 + n) * 4 + g) * 64 + lane] = v4;
          }
      }
    __syncthreads();
    const float a0 = coef0(src), a1 = coef1(src);
    if (wv < MT)
#pragma unroll
    for (int n = 0; n < 2; ++n)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const f32x4 v4 = X[((src * 2 + n) * 4 + g) * 64 + lane];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          y0[n][4 * g + i] = fmaf(a0, v4[i], y0[n][4 * g + i]);
          y1[n][4 * g + i] = fmaf(a1, v4[i], y1[n][4 * g + i]);
        }
      }
    if (k < 3) __syncthreads();
  }

  tick(6);
  // ---- epilogue (conv_x6.hip's, on the transformed tile): register r of the C layout = (row bit r >> 3, pair jc(r & 7)) ----
  const int Ho = POOL ? (H >> 1) : H, Wo = POOL ? (W >> 1) : W;
  const size_t img_elems = (POUT ? planes_image_pixels_wg(Ho, Wo) : (size_t)Ho * Wo) * cout;
  const int par = lx & 1;
  const unsigned psel = par ? 0x03020706u : 0x05040100u;
  const dim_rsrc rs_f = buf_rsrc(out + (size_t)b * img_elems, POUT ? 0 : img_elems * 4);
  const dim_rsrc rs_p = buf_rsrc(out + (size_t)b * img_elems, POUT ? img_elems * 4 : 0);
  constexpr float OSC = POUT ? DIM_F16_ACT_SCALE : 1.0f;
  auto run_epilogue = [&](auto chk_t) {
    constexpr bool CHK = decltype(chk_t)::value;
    auto put2 = [&](unsigned pix0, int col0, int wlim, int co, unsigned cpart, float v0, float v1) {
      if (POUT) {
        unsigned h, l;
        split2_pk_raw(v0, v1, h, l);
        const unsigned ho = byte_perm(lane_swap1(h), h, psel), lo = byte_perm(lane_swap1(l), l, psel);
        const unsigned pix = pix0 + (unsigned)par;
        unsigned off = (pix >> 1) * ((unsigned)cout * 8u) + (pix & 1u) * 64u + cpart;
        if (CHK) off = (col0 + par < wlim) ? off : DIM_BUF_OOB;
        buf_store_u32(rs_p, off, ho);
        buf_store_u32(rs_p, off + 32u, lo);
      } else {
        const unsigned o0 = (pix0 * (unsigned)cout + (unsigned)co) * 4u;
        buf_store_f32(rs_f, (!CHK || col0 < wlim) ? o0 : DIM_BUF_OOB, v0);
        buf_store_f32(rs_f, (!CHK || col0 + 1 < wlim) ? o0 + (unsigned)cout * 4u : DIM_BUF_OOB, v1);
      }
    };
    float vmax = 0.0f;
#pragma unroll
    for (int n = 0; n < 2; ++n) {
      const int co = cb * 64 + n * 32 + lx;
      const float bvv = bias[co] * OSC;
      const float inv_scale = inv_ch[co] * OSC;
      const unsigned c2 = (unsigned)(co - par);
      const unsigned cpart = ((c2 >> 4) * 64u + (c2 & 15u)) * 2u;
      auto finish = [&](float& v0, float& v1) {
        if (POUT) {
          vmax = fmaxf(vmax, fmaxf(v0, v1));
          v0 = __builtin_amdgcn_fmed3f(v0, 0.0f, 65504.0f);
          v1 = __builtin_amdgcn_fmed3f(v1, 0.0f, 65504.0f);
        } else {
          v0 = fmaxf(v0, 0.0f); v1 = fmaxf(v1, 0.0f);
          vmax = sat_track(vmax, v0, v1);
        }
      };
      if (POOL) {
        const int py = (oy >> 1) + wv, pxb = ox >> 1;
        float pv[8];
#pragma unroll
        for (int r = 0; r < 8; ++r)
          pv[r] = fmaxf(fmaxf(y0[n][r], y1[n][r]), fmaxf(y0[n][r + 8], y1[n][r + 8])) * inv_scale + bvv;
#pragma unroll
        for (int r = 0; r < 8; r += 2) {
          const int q0 = mfma_row(r, half);   // pooled column inside the tile = the pair index; register r + 1 is the next column
          finish(pv[r], pv[r + 1]);
          put2((unsigned)(py * Wo + pxb + q0), pxb + q0, Wo, co, cpart, pv[r], pv[r + 1]);
        }
      } else {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int y = oy + 2 * wv + (r >> 3), x0 = ox + 2 * mfma_row(r & 7, half);
          float v0 = y0[n][r] * inv_scale + bvv, v1 = y1[n][r] * inv_scale + bvv;
          finish(v0, v1);
          put2((unsigned)(y * W + x0), x0, W, co, cpart, v0, v1);
        }
      }
    }
    sat_report(sat, vmax * (1.0f / OSC));
  };
  if (wv < MT) {   // (6-row tiles: the fourth wave owns a position but no output rows)
    if (POOL ? ((ox >> 1) + WG_TW / 2 <= Wo) : (ox + WG_TW <= W)) run_epilogue(std::false_type{});
    else run_epilogue(std::true_type{});
  }
  tick(7);
  if (NTILE > 1 && it + 1 < NTILE && tile + 1 < n_tiles) load_b(0, std::integral_constant<int, 0>{});   // step 0 of the next tile: in flight during its phase 1
  }   // tile loop
  sat_report(sat, vmax_in * (1.0f / DIM_F16_ACT_SCALE));
  if (TIMING) {
    if (t == 0) {
#pragma unroll
      for (int i = 0; i < 8; ++i) atomicAdd(&g_wg_phase[i], tph[i]);
      atomicAdd(&g_wg_phase[8], __builtin_readcyclecounter() - tstart);
      atomicAdd(&g_wg_phase[9], 1ull);
    }
  }
}
}  // namespace

// Host: OIHW fp32 3x3 weights -> Winograd F(2,3)-along-x transformed, fp16x3-split pieces
// [cout/64][cin/16][dy][position 4][plane 2][n 2][k-half 2][32 co][8 ci], followed by the fp32 per-output-channel inverse scales.
// The transform is evaluated in fp64 and rounded once to fp32 (u1, u2 are not fp32-exact sums); the power-of-two scale per output
// channel puts max|u[co]| into [8192, 16384) as prepare_conv_weights_split does for the untransformed weights.
size_t conv_wino_weight_elems(int cin, int cout) { return (size_t)(cout / 64) * (cin / 16) * 3 * 4 * WG_WFRAG + 2 * (size_t)cout; }
void prepare_conv_weights_wino(const float* w_oihw, int cin, int cout, unsigned short* out, SplitWeights* sw) {
  const int nchunk = cin / 16;
  sw->mode = 2;
  sw->scale_off = conv_wino_weight_elems(cin, cout) - 2 * (size_t)cout;
  float* inv = (float*)(out + sw->scale_off);
  for (int co = 0; co < cout; ++co) {
    auto u_of = [&](int ci, int dy, int p) -> float {
      const float* g = w_oihw + (((size_t)co * cin + ci) * 3 + dy) * 3;
      const double g0 = g[0], g1 = g[1], g2 = g[2];
      const double u = p == 0 ? g0 : p == 1 ? 0.5 * (g0 + g1 + g2) : p == 2 ? 0.5 * (g0 - g1 + g2) : g2;
      return (float)u;
    };
    float mx = 0.f;
    for (int ci = 0; ci < cin; ++ci)
      for (int dy = 0; dy < 3; ++dy)
        for (int p = 0; p < 4; ++p) mx = fmaxf(mx, fabsf(u_of(ci, dy, p)));
    float wscale = 1.0f;
    int e2 = 0;
    if (mx > 0.f && mx < INFINITY) { frexpf(mx, &e2); wscale = ldexpf(1.0f, 14 - e2); }
    const float inv_co = 1.0f / (wscale * DIM_F16_ACT_SCALE);
    memcpy(&inv[co], &inv_co, 4);
    const int cb = co / 64, n = (co % 64) / 32, col = co % 32;
    for (int ci = 0; ci < cin; ++ci)
      for (int dy = 0; dy < 3; ++dy)
        for (int p = 0; p < 4; ++p) {
          float x = u_of(ci, dy, p) * wscale;
          const int c = ci / 16, hf = (ci % 16) / 8, e = ci % 8;
          for (int pl = 0; pl < 2; ++pl) {
            const _Float16 hv = (_Float16)x;
            unsigned short bits;
            memcpy(&bits, &hv, 2);
            const size_t idx = ((((((((size_t)(cb * nchunk + c) * 3 + dy) * 4 + p) * 2 + pl) * 2 + n) * 2 + hf) * 32 + col) * 8) + e;
            out[idx] = bits;
            x = x - (float)hv;
          }
        }
  }
}

// conv1a (1 -> 64) + Winograd conv1b (64 -> cout), bias, ReLU, optional 2x2 max-pool; output fp32 NHWC or pre-split planes
int launch_conv3x3_wg_fused1a(const float* image, const float* w1a_tap_cout, const float* b1a, const SplitWeights& wt, const float* bias,
                              float* out, int batch, int H, int W, int cout, int pool, int planes_out, hipStream_t s, unsigned* sat,
                              unsigned* sat_image) {
  const int var = dim_conv_winograd();
  const int mt = ((var >> 4) & 3) == 1 ? 3 : 4, stagger = (var >> 8) & 255;
  DIM_REQUIRE(cout % 64 == 0, "conv3x3_wg fused conv1a: cout=%d must be a multiple of 64", cout);
  DIM_REQUIRE(wt.dev && wt.mode == 2, "conv3x3_wg: Winograd weights not prepared");
  if (batch <= 0 || H <= 0 || W <= 0) return 0;
  const int tiles_x = cdiv(W, WG_TW), tiles_y = cdiv(H, wg_th(mt));
  const int n_tiles = tiles_x * tiles_y;
  const int ntile = (var & 32) ? 8 : ((var & 64) ? 4 : 1);   // persistent workgroups: consecutive tiles per workgroup
  dim3 grid(cdiv(n_tiles, ntile), cout / 64, batch);
  if (pool && planes_out && ntile > 1 && mt == 4) {
    if (ntile == 8) {
      if (var & 8) hipLaunchKernelGGL(HIP_KERNEL_NAME(conv3x3_wg_f1a_kernel<64, 1, true, 4, 0, 1, true, 8>), grid, dim3(256), 0, s, image, wt.dev, bias, out, H, W, cout, tiles_x, w1a_tap_cout, b1a, wt.inv_ch(), sat, sat_image, stagger, n_tiles);
      else if (var & 2) hipLaunchKernelGGL(HIP_KERNEL_NAME(conv3x3_wg_f1a_kernel<64, 1, true, 4, 0, 1, false, 8>), grid, dim3(256), 0, s, image, wt.dev, bias, out, H, W, cout, tiles_x, w1a_tap_cout, b1a, wt.inv_ch(), sat, sat_image, stagger, n_tiles);
      else hipLaunchKernelGGL(HIP_KERNEL_NAME(conv3x3_wg_f1a_kernel<64, 1, true, 4, 0, 0, false, 8>), grid, dim3(256), 0, s, image, wt.dev, bias, out, H, W, cout, tiles_x, w1a_tap_cout, b1a, wt.inv_ch(), sat, sat_image, stagger, n_tiles);
    } else {
      if (var & 2) hipLaunchKernelGGL(HIP_KERNEL_NAME(conv3x3_wg_f1a_kernel<64, 1, true, 4, 0, 1, false, 4>), grid, dim3(256), 0, s, image, wt.dev, bias, out, H, W, cout, tiles_x, w1a_tap_cout, b1a, wt.inv_ch(), sat, sat_image, stagger, n_tiles);
      else hipLaunchKernelGGL(HIP_KERNEL_NAME(conv3x3_wg_f1a_kernel<64, 1, true, 4, 0, 0, false, 4>), grid, dim3(256), 0, s, image, wt.dev, bias, out, H, W, cout, tiles_x, w1a_tap_cout, b1a, wt.inv_ch(), sat, sat_image, stagger, n_tiles);
    }
    DIM_LAUNCH_CHECK();
    return 0;
  }
#define DIM_WG(P, PO, MTV) hipLaunchKernelGGL(HIP_KERNEL_NAME(conv3x3_wg_f1a_kernel<64, P, PO, MTV>), grid, dim3(256), 0, s, image, wt.dev, bias, out, H, W, cout, tiles_x, w1a_tap_cout, b1a, wt.inv_ch(), sat, sat_image, stagger, n_tiles)
#define DIM_WG_MT(P, PO) { if (mt == 3) DIM_WG(P, PO, 3); else DIM_WG(P, PO, 4); }
  if (pool && planes_out && (var & 128) && mt == 4) {   // timing probe 2: no weight-fragment reloads
    if (var & 2) hipLaunchKernelGGL(HIP_KERNEL_NAME(conv3x3_wg_f1a_kernel<64, 1, true, 4, 2, 1>), grid, dim3(256), 0, s, image, wt.dev, bias, out, H, W, cout, tiles_x, w1a_tap_cout, b1a, wt.inv_ch(), sat, sat_image, stagger, n_tiles);
    else hipLaunchKernelGGL(HIP_KERNEL_NAME(conv3x3_wg_f1a_kernel<64, 1, true, 4, 2, 0>), grid, dim3(256), 0, s, image, wt.dev, bias, out, H, W, cout, tiles_x, w1a_tap_cout, b1a, wt.inv_ch(), sat, sat_image, stagger, n_tiles);
  } else if (pool && planes_out && (var & 8) && mt == 4) {   // phase timers (s_memtime deltas of wave 0, dim_conv_wg_phase_read)
    if (var & 2) hipLaunchKernelGGL(HIP_KERNEL_NAME(conv3x3_wg_f1a_kernel<64, 1, true, 4, 0, 1, true>), grid, dim3(256), 0, s, image, wt.dev, bias, out, H, W, cout, tiles_x, w1a_tap_cout, b1a, wt.inv_ch(), sat, sat_image, stagger, n_tiles);
    else hipLaunchKernelGGL(HIP_KERNEL_NAME(conv3x3_wg_f1a_kernel<64, 1, true, 4, 0, 0, true>), grid, dim3(256), 0, s, image, wt.dev, bias, out, H, W, cout, tiles_x, w1a_tap_cout, b1a, wt.inv_ch(), sat, sat_image, stagger, n_tiles);
  } else if (pool && planes_out && (var & 2)) {   // two-phase staging (Sx scratch, packed fp32 conv1a, fma_mix splits)
    if (var & 4) hipLaunchKernelGGL(HIP_KERNEL_NAME(conv3x3_wg_f1a_kernel<64, 1, true, 4, 1, 1>), grid, dim3(256), 0, s, image, wt.dev, bias, out, H, W, cout, tiles_x, w1a_tap_cout, b1a, wt.inv_ch(), sat, sat_image, stagger, n_tiles);   // timing probe
    else if (mt == 3) hipLaunchKernelGGL(HIP_KERNEL_NAME(conv3x3_wg_f1a_kernel<64, 1, true, 3, 0, 1>), grid, dim3(256), 0, s, image, wt.dev, bias, out, H, W, cout, tiles_x, w1a_tap_cout, b1a, wt.inv_ch(), sat, sat_image, stagger, n_tiles);
    else hipLaunchKernelGGL(HIP_KERNEL_NAME(conv3x3_wg_f1a_kernel<64, 1, true, 4, 0, 1>), grid, dim3(256), 0, s, image, wt.dev, bias, out, H, W, cout, tiles_x, w1a_tap_cout, b1a, wt.inv_ch(), sat, sat_image, stagger, n_tiles);
  } else
  if (pool && planes_out && (var & 4)) hipLaunchKernelGGL(HIP_KERNEL_NAME(conv3x3_wg_f1a_kernel<64, 1, true, 4, 1>), grid, dim3(256), 0, s, image, wt.dev, bias, out, H, W, cout, tiles_x, w1a_tap_cout, b1a, wt.inv_ch(), sat, sat_image, stagger, n_tiles);   // timing probe
  else if (pool && planes_out) DIM_WG_MT(1, true)
  else if (pool) DIM_WG_MT(1, false)
  else if (planes_out) DIM_WG_MT(0, true)
  else DIM_WG_MT(0, false)
#undef DIM_WG_MT
#undef DIM_WG
  DIM_LAUNCH_CHECK();
  return 0;
}

// TIMING builds (dim_tune_set(15, .. | 8)): phase cycles of wave 0 summed over the workgroups since the last reset:
// [0] prologue, [1] staging (fused) / phase 2, [2] barrier after it, [3] MFMA steps, [4] phase 1 of the next chunk, [5] barrier after the
// MFMA interval, [6] output transform across the waves, [7] epilogue, [8] whole workgroup, [9] workgroups
extern "C" int dim_conv_wg_phase_read(unsigned long long* host16, int reset) {
  DIM_HIP(hipMemcpyFromSymbol(host16, HIP_SYMBOL(g_wg_phase), 16 * sizeof(unsigned long long)));
  if (reset) {
    unsigned long long z[16] = {0};
    DIM_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_wg_phase), z, sizeof(z)));
  }
  return 0;
}

#endif   // DIM_RESEARCH
