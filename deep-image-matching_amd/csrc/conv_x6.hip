// 3x3 convolution on the 16-bit matrix cores at fp32 accuracy: every fp32 operand is split into 16-bit pieces (SplitMma,
// dim_common.h: fp16x3 = two fp16 pieces of the power-of-two-scaled value x three cross terms, the default; bf16x6 = three
// bf16 pieces x six terms) and each 32x32x16 MFMA step issues the cross terms instead of eight fp32 MFMAs.
//
// Implicit GEMM, M = pixels, N = cout, K = 9 cin.  Workgroup = 4 waves; production shape (MR 4): output tile 16 rows x 32
// columns x 64 channels, wave w owns rows 4w .. 4w+3 x two 32-channel slabs = 8 accumulators, 2 workgroups per CU.
// K loop: 16 input channels per chunk; inside a chunk the weights are staged one kernel ROW (3 taps) at a time:
//   Ip[plane][k-half 2][pixel (TH+2) x 34][8 x 16 bit]   one 16-B ds_read_b128 = one A operand,
//   Wp[plane][dx 3][k-half 2][cout 64][8 x 16 bit]       one 16-B ds_read_b128 = one B operand (double-buffered with WDMA);
// consecutive lanes read consecutive 16-B slots in both images: conflict-free without padding.
// The pre-split weight buffer is laid out so that each (cout block, chunk, kernel row) slice is one contiguous run
// (a straight 16-B-per-lane copy, or an LDS-DMA).  Template parameters below: fused conv1a (F1A), pre-split input / output
// (PIN / POUT), tile height (MR), LDS-DMA weight staging (WDMA).
#include <math.h>
#include <string.h>

#include <type_traits>

#include "dim_kernels.h"

namespace {
constexpr int TW = 32, IW = TW + 2;
constexpr int tile_rows(int mr) { return 4 * mr; }   // output rows per workgroup: 4 waves x MR rows
__host__ __device__ constexpr size_t planes_image_pixels(int h, int w) { return ((size_t)h * w + 1) & ~(size_t)1; }  // pixel slots of one pre-split image
constexpr int w_slice(int npl) { return npl * 3 * 2 * 64 * 8; }              // 16-bit elements per (cb, chunk, dy) slice

// PF: 0 = no software prefetch, 1 = next weight slice fetched into registers behind the MFMAs,
//     2 = additionally the next chunk's halo tile (24 more VGPRs: 2 instead of 3 workgroups per CU).
// F1A: the input is the 1-channel image and the kernel computes SuperPoint's conv1a (1 -> 64, 3x3, bias,
//      ReLU; SPN:161) on the fly while it stages its own halo tile, with the arithmetic of conv1a_kernel
//      (conv.hip: fmaf chain over the 9 taps from 0, then + bias, ReLU): the 268-MB conv1a map of a 1024^2
//      image is neither written nor read back.  CIN must be 64.
// MODE: SplitMma policy (dim_common.h) — 1: three bf16 planes x six cross terms, 2: two fp16 planes x three
//       cross terms (activations scaled by act_scale(), weights pre-scaled; inv_scale undoes both exactly).
// PIN / POUT (mode 2 only): the input / output lives in HBM pre-split, in the bytes of the fp32 NHWC tensor it replaces:
//      per PAIR of pixels (linear index 2i, 2i+1 inside the image; an image occupies an even number of pixel slots) and
//      group of 16 channels: pixel 2i's 16 h pieces, its 16 l pieces, then pixel 2i+1's (fp16 of 16 x the activation,
//      clamped to +-65504): element (pixel, channel c, plane) sits at 16-bit index
//      (pixel / 2) * 4C + (c / 16) * 64 + (pixel % 2) * 32 + plane * 16 + c % 16 — the 16 channels of one K chunk of two
//      neighbouring pixels are ONE 128-byte L2 line (with the pixels' chunks laid out one pixel after the other, every line
//      held chunks c and c + 1 of one pixel, the second half was usually evicted before the K loop came back for it, and
//      the 512^2 x 64 layers fetched 1.66 x their input).  A producer splits each value ONCE in its epilogue; a
//      consumer stages its halo tile with straight 16-byte copies — no per-consumer split VALU (every element
//      used to be split (cout / 64) x 1.33 times, ~5 VALU each) — and the 16 channels of one K chunk are ONE
//      contiguous 64-byte run per pixel.  (Round 1 kept two separate planes: a chunk then read 32 bytes per pixel and
//      plane, half of every 64-byte HBM burst was wasted, and conv2a .. conv4b ran at 4.2-5.3 TB/s of FETCH — HBM-bound
//      on 2.6-6x their input — instead of matrix-core-bound.)
// MR: output rows per wave.  2: tile 8 x 32, 4 accumulators per wave, 3 workgroups per CU.  4: tile 16 x 32, 8 accumulators,
//     2 workgroups per CU — 12 instead of 16 LDS operand reads per 24 MFMAs, half the weight staging per MFMA, 1.20 x
//     instead of 1.33 x halo overhead: the production shapes run it (measured 522 -> 531 pairs/s at a power-limited clock;
//     staging all three kernel rows of weights per barrier pair on top of it was +-0 at MR 4 and 3 % slower at MR 2).
// WDMA (the 16-row fp16x3 kernels): the weight slice of the NEXT kernel row travels L2 -> LDS by LDS-DMA into the other half of
//     a double-buffered Wp while the MFMAs of the current row run: no staging registers, no ds_write pass, one barrier per
//     kernel row instead of two (the slice copy through registers was the largest staging cost: +4 % of the step when
//     switched off in the stage-ablation experiment).
template <int CIN, int POOL, int PF, bool F1A, int MODE, bool PIN, bool POUT, int MR = 2, bool WDMA = false>
__global__ __launch_bounds__(256, ((PF == 2 || MR == 4) ? 2 : 3)) void conv3x3_x6_kernel(const float* __restrict__ in, const unsigned short* __restrict__ wx,
                                                            const float* __restrict__ bias, float* __restrict__ out, int H, int W,
                                                            int cout, int relu, int tiles_x, const float* __restrict__ w1a,
                                                            const float* __restrict__ b1a, const float* __restrict__ inv_ch, unsigned* sat,
                                                            unsigned* sat_image) {
  static_assert(!(PIN || POUT) || MODE == 2, "pre-split planes exist for the fp16x3 mode only");
  static_assert(!(PIN && F1A), "the fused conv1a computes its own input");
  using S = SplitMma<MODE>;
  constexpr int NPL = S::NPL, W_SLICE = w_slice(NPL);
  constexpr int TH = tile_rows(MR), IH = TH + 2, NPIX = IH * IW;  // MR 2: 340 halo pixels, MR 4: 612
  __shared__ u32x4 Ip[NPL * 2 * NPIX];
  constexpr int WSL = NPL * 3 * 2 * 64;   // 16-byte slots of one weight slice
  __shared__ u32x4 Wp[(WDMA ? 2 : 1) * WSL];
  constexpr int IMW = IW + 2, IMH = IH + 2;  // image tile of the fused conv1a: halo of the halo
  __shared__ float Img[F1A ? IMH * IMW : 1];
  __shared__ float W1a[F1A ? 9 * 64 + 64 : 1];  // conv1a weights [tap][64] + bias[64]: read per chunk from LDS, not from L2
  constexpr int NCHUNK = CIN / 16;

  const int t = threadIdx.x;
  const int lane = t & 63, wv = t >> 6, lx = lane & 31, half = lane >> 5;
  // XCD-aware tile order (cdna_hip_programming.md T1): the hardware sends workgroup L to XCD L % 8; giving every XCD a
  // contiguous BAND of the image's tiles (bijective for any tile count) lets neighbouring tiles find each other's halo
  // rows / columns in that XCD's L2 instead of fetching them again.  Speed only.
  int tile;
  {
    const int nt = gridDim.x, xcd = blockIdx.x & 7, j = blockIdx.x >> 3, q = nt >> 3, r = nt & 7;
    tile = xcd * q + min(xcd, r) + j;
  }
  const int ty = tile / tiles_x, tx = tile % tiles_x;
  const int cb = blockIdx.y, b = blockIdx.z;
  const int oy = ty * TH, ox = tx * TW;
  const float* in_b = in + (size_t)b * H * W * (F1A ? 1 : CIN);

  f32x16 acc[MR][2];
#pragma unroll
  for (int i = 0; i < MR; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

  constexpr int W_ITEMS = W_SLICE / 8;              // 16-B weight items per staging step
  constexpr int NWV = (W_ITEMS + 255) / 256;        // per thread (bf16x6: 4.5 -> 5; fp16x3: 3)
  constexpr int NIN = (NPIX * 4 + 255) / 256;       // float4 halo items per thread per chunk (5.3 -> 6)
  u32x4 rw[NWV];
  float4 rin[NIN];
  auto load_w = [&](int c, int dy) {
    const u32x4* src = (const u32x4*)(wx + ((size_t)(cb * NCHUNK + c) * 3 + dy) * W_SLICE);
#pragma unroll
    for (int i = 0; i < NWV; ++i) {
      const int idx = t + 256 * i;
      if (idx < W_ITEMS) rw[i] = src[idx];
    }
  };
  auto store_w = [&]() {
#pragma unroll
    for (int i = 0; i < NWV; ++i) {
      const int idx = t + 256 * i;
      if (idx < W_ITEMS) Wp[idx] = rw[i];
    }
  };
  auto load_in = [&](int c) {
#pragma unroll
    for (int i = 0; i < NIN; ++i) {
      const int idx = t + 256 * i;
      const int p = idx >> 2, q = idx & 3;
      const int py = p / IW, px = p - py * IW;
      const int gy = oy + py - 1, gx = ox + px - 1;
      rin[i] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (PIN) {  // q = plane * 2 + k-half: 8 consecutive channels of one plane = one 16-byte item; the 4 items of a pixel are contiguous
        const unsigned short* pl = (const unsigned short*)in + (size_t)b * planes_image_pixels(H, W) * CIN * 2;
        const int pix = gy * W + gx;
        if (idx < NPIX * 4 && gy >= 0 && gy < H && gx >= 0 && gx < W) rin[i] = *(const float4*)(pl + (size_t)(pix >> 1) * (4 * CIN) + c * 64 + (pix & 1) * 32 + q * 8);
      } else if (idx < NPIX * 4 && gy >= 0 && gy < H && gx >= 0 && gx < W) rin[i] = *(const float4*)(in_b + ((size_t)gy * W + gx) * CIN + c * 16 + q * 4);
    }
  };
  auto store_in = [&]() {
#pragma unroll
    for (int i = 0; i < NIN; ++i) {
      const int idx = t + 256 * i;
      if (idx < NPIX * 4) {
        const int p = idx >> 2, q = idx & 3;
        if (PIN) {  // already split: the item IS the LDS slot of (plane, k-half) = q
          Ip[q * NPIX + p] = __builtin_bit_cast(u32x4, rin[i]);
          continue;
        }
        unsigned p0[NPL], p1[NPL];
        S::split(rin[i].x, rin[i].y, F1A ? 1.0f : S::act_scale(), p0);  // F1A: already scaled (see W1a)
        S::split(rin[i].z, rin[i].w, F1A ? 1.0f : S::act_scale(), p1);
        // channels q*4..q*4+3 live in k-half q>>1, dwords (q&1)*2, +1 of that pixel's 16-B slot
        unsigned* d = (unsigned*)&Ip[(q >> 1) * NPIX + p] + (q & 1) * 2;
#pragma unroll
        for (int pl = 0; pl < NPL; ++pl) { d[pl * 2 * NPIX * 4] = p0[pl]; d[pl * 2 * NPIX * 4 + 1] = p1[pl]; }
      }
    }
  };

  // chunk-invariant part of the fused conv1a: where each staged item sits in the image patch
  int img_off[F1A ? NIN : 1];
  if (F1A) {
#pragma unroll
    for (int i = 0; i < NIN; ++i) {
      const int idx = t + 256 * i, p = idx >> 2;
      const int py = p / IW, px = p - py * IW;
      const int gy = oy + py - 1, gx = ox + px - 1;
      img_off[i] = (idx < NPIX * 4 && gy >= 0 && gy < H && gx >= 0 && gx < W) ? py * IMW + px : -1;
    }
  }
  // fused conv1a: this thread's quad of channels is fixed (q = t & 3); per chunk it needs 9 x 4 weights
  auto conv1a_in = [&](int c) {
    const int q = t & 3;
    float wr[9][4];
#pragma unroll
    for (int k = 0; k < 9; ++k) {
      const float4 v = *(const float4*)&W1a[k * 64 + c * 16 + q * 4];
      wr[k][0] = v.x; wr[k][1] = v.y; wr[k][2] = v.z; wr[k][3] = v.w;
    }
    const float4 bv = *(const float4*)&W1a[9 * 64 + c * 16 + q * 4];
#pragma unroll
    for (int i = 0; i < NIN; ++i) {
      const int io = img_off[i];  // -1: not a halo pixel of this tile, or outside the image (conv1b's zero padding)
      const int o = io < 0 ? 0 : io;
      float o0 = 0.f, o1 = 0.f, o2 = 0.f, o3 = 0.f;
#pragma unroll
      for (int k = 0; k < 9; ++k) {
        const float v = Img[o + (k / 3) * IMW + k % 3];
        o0 = fmaf(v, wr[k][0], o0); o1 = fmaf(v, wr[k][1], o1);
        o2 = fmaf(v, wr[k][2], o2); o3 = fmaf(v, wr[k][3], o3);
      }
      rin[i] = io >= 0 ? make_float4(fmaxf(o0 + bv.x, 0.f), fmaxf(o1 + bv.y, 0.f), fmaxf(o2 + bv.z, 0.f), fmaxf(o3 + bv.w, 0.f))
                       : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  };
  if (F1A) {
    // range guard: the host bound on conv1a's outputs assumes |image| <= 1 (image / 255).  Tracked on the bit patterns of |v|
    // (ordered like the values for non-negative floats; NaN patterns sort above Inf, so a NaN pixel trips the guard too).
    unsigned imax = 0u;
    for (int idx = t; idx < IMH * IMW; idx += 256) {
      const int r = idx / IMW, cc = idx - r * IMW;
      const int gy = oy + r - 2, gx = ox + cc - 2;
      const float v = (gy >= 0 && gy < H && gx >= 0 && gx < W) ? in_b[(size_t)gy * W + gx] : 0.0f;
      Img[idx] = v;
      imax = max(imax, __float_as_uint(v) & 0x7fffffffu);
    }
    if (sat_image != nullptr && imax > 0x3f800000u) atomicAdd(sat_image, 1u);
    // weights and bias pre-multiplied by the activation scale (a power of two: fmaf(v, s w, s acc) = s fmaf(v, w, acc)
    // exactly), so the conv1a outputs come out scaled and their split skips the multiply
    for (int idx = t; idx < 9 * 64 + 64; idx += 256) W1a[idx] = (idx < 9 * 64 ? w1a[idx] : b1a[idx - 9 * 64]) * S::act_scale();
  }

  // WDMA: slice `ph` = (chunk, kernel row) of this cout block, the slices are contiguous in HBM; wave w moves items
  // w * 64 + 256 i + lane (the same item-to-thread mapping as the register path)
  auto dma_w = [&](int ph) {
    const u32x4* src = (const u32x4*)(wx + ((size_t)cb * NCHUNK * 3 + ph) * W_SLICE);
#pragma unroll
    for (int i = 0; i < W_ITEMS / 256; ++i) lds_dma16(src + wv * 64 + 256 * i + lane, &Wp[(ph & 1) * WSL + wv * 64 + 256 * i]);
  };
  static_assert(!WDMA || W_ITEMS % 256 == 0, "LDS-DMA weight staging moves whole 256-item rounds");
  if (WDMA) dma_w(0);
  else if (PF >= 1) load_w(0, 0);
  if (PF == 2 && !F1A) load_in(0);
  for (int c = 0; c < NCHUNK; ++c) {
    __syncthreads();  // every wave is done with the previous chunk's Ip / Wp (and Img is complete)
    if (F1A) conv1a_in(c);
    else if (PF != 2) load_in(c);
    store_in();
    for (int dy = 0; dy < 3; ++dy) {
      const int wbuf = WDMA ? ((c * 3 + dy) & 1) * WSL : 0;
      if (WDMA) {
        // slice (c, dy) was requested one kernel row ago: the barrier's vmcnt(0) retires it for every wave, and every
        // wave has left the row that read the other buffer, which the next slice may now overwrite
        lds_dma_wait_all();
        __syncthreads();
        if (c * 3 + dy + 1 < NCHUNK * 3) dma_w(c * 3 + dy + 1);
      } else {
        if (dy > 0) __syncthreads();  // previous kernel row's weights consumed
        if (PF == 0) load_w(c, dy);
        store_w();
        __syncthreads();
        if (PF >= 1) {  // next slice (and, at the last kernel row, the next halo tile) behind the MFMAs
          if (dy < 2) load_w(c, dy + 1);
          else if (c + 1 < NCHUNK) load_w(c + 1, 0);
        }
      }
      if (PF == 2 && !F1A && dy == 2 && c + 1 < NCHUNK) load_in(c + 1);
      __builtin_amdgcn_s_setprio(1);  // a wave in its matrix phase outranks the co-resident waves that are staging (T5)
#pragma unroll
      for (int dx = 0; dx < 3; ++dx) {
        u32x4 fa[MR][NPL], fb[2][NPL];
#pragma unroll
        for (int p = 0; p < NPL; ++p) {
#pragma unroll
          for (int m = 0; m < MR; ++m) fa[m][p] = Ip[(p * 2 + half) * NPIX + (MR * wv + m + dy) * IW + lx + dx];
#pragma unroll
          for (int n = 0; n < 2; ++n) fb[n][p] = Wp[wbuf + ((p * 3 + dx) * 2 + half) * 64 + n * 32 + lx];
        }
#pragma unroll
        for (int tm = 0; tm < S::NT; ++tm)  // smallest cross terms first
#pragma unroll
          for (int m = 0; m < MR; ++m)
#pragma unroll
            for (int n = 0; n < 2; ++n) acc[m][n] = S::mma(fa[m][S::ta(tm)], fb[n][S::tb(tm)], acc[m][n]);
      }
      __builtin_amdgcn_s_setprio(0);
    }
  }

  // Epilogue through per-image buffer descriptors (dim_common.h): 32-bit byte offsets, no per-element branches — rows past
  // the image end fall outside the descriptor and are dropped by the hardware, a column past the image width selects the
  // out-of-range offset.  With POUT two neighbouring pixels are split together (split2_pk packs two values), adjacent
  // lanes (= adjacent output channels) exchange their packed halves with one DPP move per plane, and every lane stores
  // ONE 4-byte {channel pair} per plane: even lanes the first pixel, odd lanes the second (half the store instructions
  // of 2-byte stores, 4 instead of ~25 instructions per pair of values).
  const int Ho = POOL ? (H >> 1) : H, Wo = POOL ? (W >> 1) : W;
  const size_t img_elems = (POUT ? planes_image_pixels(Ho, Wo) : (size_t)Ho * Wo) * cout;
  const int par = lx & 1;
  const unsigned psel = par ? 0x03020706u : 0x05040100u;   // odd lanes keep the high halves (second pixel), even lanes the low halves
  const dim_rsrc rs_f = buf_rsrc(out + (size_t)b * img_elems, POUT ? 0 : img_elems * 4);
  const dim_rsrc rs_p = buf_rsrc(out + (size_t)b * img_elems, POUT ? img_elems * 4 : 0);  // h and l pieces interleaved per 16 channels
  // POUT: the values arrive already multiplied by the activation scale (folded into the per-channel factor and the bias: a
  // power of two, exact), ReLU and the fp16 clamp are ONE v_med3 (0 .. 65504; the launchers require relu for pre-split
  // outputs), the range guard tracks the scaled value before the clamp.  CHK = false (the tile lies inside the image's
  // columns: every tile but the last of a ragged row) drops the per-element column test.
  // pix0: pixel index of the first value inside the image (the second value is the next pixel); rows past the image end
  // lie past the descriptor and are dropped by the hardware
  constexpr float OSC = POUT ? DIM_F16_ACT_SCALE : 1.0f;
  auto run_epilogue = [&](auto chk_t) {
    constexpr bool CHK = decltype(chk_t)::value;
    auto put2 = [&](unsigned pix0, int col0, int wlim, int co, unsigned cpart, float v0, float v1) {
      if (POUT) {
        unsigned h, l;
        split2_pk_raw(v0, v1, h, l);
        const unsigned ho = byte_perm(lane_swap1(h), h, psel), lo = byte_perm(lane_swap1(l), l, psel);
        const unsigned pix = pix0 + (unsigned)par;   // even lanes store the first pixel's channel pair, odd lanes the second's
        unsigned off = (pix >> 1) * ((unsigned)cout * 8u) + (pix & 1u) * 64u + cpart;
        if (CHK) off = (col0 + par < wlim) ? off : DIM_BUF_OOB;
        buf_store_u32(rs_p, off, ho);
        buf_store_u32(rs_p, off + 32u, lo);   // the l pieces of the group follow its 16 h pieces
      } else {
        const unsigned o0 = (pix0 * (unsigned)cout + (unsigned)co) * 4u;
        buf_store_f32(rs_f, (!CHK || col0 < wlim) ? o0 : DIM_BUF_OOB, v0);
        buf_store_f32(rs_f, (!CHK || col0 + 1 < wlim) ? o0 + (unsigned)cout * 4u : DIM_BUF_OOB, v1);
      }
    };
    float vmax = 0.0f;  // fp16x3 range guard on everything this thread writes (dim_common.h), in units of OSC
#pragma unroll
    for (int n = 0; n < 2; ++n) {
      const int co = cb * 64 + n * 32 + lx;
      const float bv = bias[co] * OSC;
      const float inv_scale = inv_ch[co] * OSC;
      const unsigned c2 = (unsigned)(co - par);  // the even channel of this lane pair
      const unsigned cpart = ((c2 >> 4) * 64u + (c2 & 15u)) * 2u;
      auto finish = [&](float& v0, float& v1) {  // activation, range guard, clamp
        if (POUT) {
          vmax = fmaxf(vmax, fmaxf(v0, v1));
          v0 = __builtin_amdgcn_fmed3f(v0, 0.0f, 65504.0f);
          v1 = __builtin_amdgcn_fmed3f(v1, 0.0f, 65504.0f);
        } else {
          if (relu) { v0 = fmaxf(v0, 0.0f); v1 = fmaxf(v1, 0.0f); }
          vmax = sat_track(vmax, v0, v1);
        }
      };
      if (POOL) {
#pragma unroll
        for (int mp = 0; mp < MR / 2; ++mp) {
          const int py = (oy >> 1) + (MR / 2) * wv + mp, pxb = ox >> 1;
          float pv[8];
#pragma unroll
          for (int r = 0; r < 16; r += 2)
            pv[r >> 1] = fmaxf(fmaxf(acc[2 * mp][n][r], acc[2 * mp][n][r + 1]), fmaxf(acc[2 * mp + 1][n][r], acc[2 * mp + 1][n][r + 1])) * inv_scale + bv;
#pragma unroll
          for (int j = 0; j < 8; j += 2) {
            const int q0 = mfma_row(2 * j, half) >> 1;  // pooled column offset inside the tile; the next value is the next column
            finish(pv[j], pv[j + 1]);
            put2((unsigned)(py * Wo + pxb + q0), pxb + q0, Wo, co, cpart, pv[j], pv[j + 1]);
          }
        }
      } else {
#pragma unroll
        for (int m = 0; m < MR; ++m) {
          const int y = oy + MR * wv + m;
#pragma unroll
          for (int r = 0; r < 16; r += 2) {
            const int x0 = ox + mfma_row(r, half);  // r even: the odd register is the next column
            float v0 = acc[m][n][r] * inv_scale + bv, v1 = acc[m][n][r + 1] * inv_scale + bv;
            finish(v0, v1);
            put2((unsigned)(y * W + x0), x0, W, co, cpart, v0, v1);
          }
        }
      }
    }
    if (MODE == 2) sat_report(sat, vmax * (1.0f / OSC));
  };
  if (POOL ? ((ox >> 1) + TW / 2 <= Wo) : (ox + TW <= W)) run_epilogue(std::false_type{});
  else run_epilogue(std::true_type{});
}

// debug / inspection: pre-split planes back to fp32 (h + l is exact in fp32; / 16 undoes the activation scale)
__global__ __launch_bounds__(256) void planes_to_f32_kernel(const unsigned short* __restrict__ planes, int C, int hw, float* __restrict__ out, size_t n) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;   // i = (image * hw + pixel) * C + channel
  if (i >= n) return;
  const size_t gp = i / C, c = i - gp * C, b = gp / hw, pix = gp - b * hw;
  const size_t o = b * planes_image_pixels(hw, 1) * 2 * C + (pix >> 1) * 4 * C + (c >> 4) * 64 + (pix & 1) * 32 + (c & 15);
  const _Float16 hv = __builtin_bit_cast(_Float16, planes[o]), lv = __builtin_bit_cast(_Float16, planes[o + 16]);
  out[i] = ((float)hv + (float)lv) / DIM_F16_ACT_SCALE;
}

unsigned short host_bf16_rne(float x) {
  unsigned u;
  memcpy(&u, &x, 4);
  if ((u & 0x7f800000u) == 0x7f800000u) return (unsigned short)(u >> 16);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (unsigned short)(u >> 16);
}
}  // namespace

// Host: OIHW fp32 3x3 weights -> [cout/64][cin/16][dy][plane][dx][k-half][64 co][8 ci] 16-bit pieces, followed by
// the fp32 per-output-channel inverse scales [cout] (SplitWeights::scale_off).
// mode 1: three bf16 planes (RNE pieces, scales 1); mode 2: two fp16 planes of w * 2^s_co with 2^s_co chosen per output
// channel so that max|w[co]| lands in [8192, 16384) — inverse scale 1 / (2^s_co * DIM_F16_ACT_SCALE), exact.
static size_t conv_split_piece_elems(int cin, int cout, int mode) { return (size_t)(cout / 64) * (cin / 16) * 3 * w_slice(mode == 2 ? 2 : 3); }
size_t conv_split_weight_elems(int cin, int cout, int mode) { return conv_split_piece_elems(cin, cout, mode) + 2 * (size_t)cout; }
void prepare_conv_weights_split(const float* w_oihw, int cin, int cout, int mode, unsigned short* out, SplitWeights* sw) {
  const int nchunk = cin / 16, npl = mode == 2 ? 2 : 3;
  sw->mode = mode;
  sw->scale_off = conv_split_piece_elems(cin, cout, mode);
  float* inv = (float*)(out + sw->scale_off);
  for (int co = 0; co < cout; ++co) {
    float wscale = 1.0f;
    float inv_co = 1.0f;
    if (mode == 2) {
      float mx = 0.f;
      for (size_t i = 0; i < (size_t)cin * 9; ++i) mx = fmaxf(mx, fabsf(w_oihw[(size_t)co * cin * 9 + i]));
      int e = 0;
      if (mx > 0.f && mx < INFINITY) { frexpf(mx, &e); wscale = ldexpf(1.0f, 14 - e); }  // mx = f * 2^e, f in [0.5, 1) -> mx * wscale in [8192, 16384)
      inv_co = 1.0f / (wscale * DIM_F16_ACT_SCALE);
    }
    memcpy(&inv[co], &inv_co, 4);
    for (int ci = 0; ci < cin; ++ci)
      for (int dy = 0; dy < 3; ++dy)
        for (int dx = 0; dx < 3; ++dx) {
          float x = w_oihw[(((size_t)co * cin + ci) * 3 + dy) * 3 + dx] * wscale;
          const int cb = co / 64, col = co % 64, c = ci / 16, hf = (ci % 16) / 8, e = ci % 8;
          for (int p = 0; p < npl; ++p) {
            unsigned short bits;
            float piece;
            if (mode == 2) {
              const _Float16 hv = (_Float16)x;
              memcpy(&bits, &hv, 2);
              piece = (float)hv;
            } else {
              bits = host_bf16_rne(x);
              const unsigned u = (unsigned)bits << 16;
              memcpy(&piece, &u, 4);
            }
            const size_t idx = ((((((size_t)(cb * nchunk + c) * 3 + dy) * npl + p) * 3 + dx) * 2 + hf) * 64 + col) * 8 + e;
            out[idx] = bits;
            x = x - piece;
          }
        }
  }
}

static int g_conv_x6_variant = 1 | 16;  // bits 0-1: prefetch variant of the bf16x6 kernels; bit 4: 16-row tiles for the production fp16x3 shapes
int dim_conv_x6_variant() { return g_conv_x6_variant; }
void dim_conv_x6_set_variant(int v) { g_conv_x6_variant = v; }

int launch_conv3x3_x6(const float* in, const SplitWeights& wt, const float* bias, float* out, int batch, int H, int W, int cin,
                      int cout, int pool, int relu, hipStream_t s, unsigned* sat) {
  DIM_REQUIRE(cout % 64 == 0, "conv3x3_x6: cout=%d must be a multiple of 64", cout);
  DIM_REQUIRE(cin == 64 || cin == 128, "conv3x3_x6: cin=%d unsupported (64 or 128)", cin);
  DIM_REQUIRE(wt.dev && (wt.mode == 1 || wt.mode == 2), "conv3x3_x6: weights not prepared (mode %d)", wt.mode);
  if (batch <= 0 || H <= 0 || W <= 0) return 0;
  const int tiles_x = cdiv(W, TW), tiles_y = cdiv(H, tile_rows(2));
  dim3 grid(tiles_x * tiles_y, cout / 64, batch);
  const unsigned short* wx = wt.dev;
  const float* inv = wt.inv_ch();
#define DIM_CONV6(CI, P, PFV, MD) hipLaunchKernelGGL(HIP_KERNEL_NAME(conv3x3_x6_kernel<CI, P, PFV, false, MD, false, false>), grid, dim3(256), 0, s, in, wx, bias, out, H, W, cout, relu, tiles_x, (const float*)nullptr, (const float*)nullptr, inv, sat, (unsigned*)nullptr)
#define DIM_CONV6_V(PFV, MD)                              \
  {                                                       \
    if (cin == 64 && pool) DIM_CONV6(64, 1, PFV, MD);     \
    else if (cin == 64) DIM_CONV6(64, 0, PFV, MD);        \
    else if (pool) DIM_CONV6(128, 1, PFV, MD);            \
    else DIM_CONV6(128, 0, PFV, MD);                      \
  }
  if (wt.mode == 2) {
    DIM_CONV6_V(1, 2)
  } else {
    switch (dim_conv_x6_variant() & 3) {
      case 0: DIM_CONV6_V(0, 1) break;
      case 2: DIM_CONV6_V(2, 1) break;
      default: DIM_CONV6_V(1, 1) break;
    }
  }
#undef DIM_CONV6_V
#undef DIM_CONV6
  DIM_LAUNCH_CHECK();
  return 0;
}

// conv1a (1 -> 64) fused into the 64 -> cout convolution that consumes it (SuperPoint conv1a + conv1b, SPN:161-162).
int launch_conv3x3_x6_fused1a(const float* image, const float* w1a_tap_cout, const float* b1a, const SplitWeights& wt, const float* bias,
                              float* out, int batch, int H, int W, int cout, int pool, int relu, int planes_out, hipStream_t s,
                              unsigned* sat, unsigned* sat_image) {
  DIM_REQUIRE(cout % 64 == 0, "conv3x3_x6 fused conv1a: cout=%d must be a multiple of 64", cout);
  DIM_REQUIRE(wt.dev && (wt.mode == 1 || wt.mode == 2), "conv3x3_x6: weights not prepared (mode %d)", wt.mode);
  if (batch <= 0 || H <= 0 || W <= 0) return 0;
  const int var = dim_conv_x6_variant();
  const bool big = wt.mode == 2 && planes_out && pool && (var & 16);   // the production shape has the 16-row tile variant
  const int mr = big ? 4 : 2;
  const int tiles_x = cdiv(W, TW), tiles_y = cdiv(H, tile_rows(mr));
  dim3 grid(tiles_x * tiles_y, cout / 64, batch);
  DIM_REQUIRE(!planes_out || wt.mode == 2, "conv3x3_x6: pre-split output planes exist for the fp16x3 mode only");
  DIM_REQUIRE(!planes_out || relu, "conv3x3_x6: a pre-split output implies ReLU (its clamp starts at 0)");
#define DIM_CONV6F(P, MD, PO, ...) hipLaunchKernelGGL(HIP_KERNEL_NAME(conv3x3_x6_kernel<64, P, 1, true, MD, false, PO, ##__VA_ARGS__>), grid, dim3(256), 0, s, image, wt.dev, bias, out, H, W, cout, relu, tiles_x, w1a_tap_cout, b1a, wt.inv_ch(), sat, sat_image)
  if (big) DIM_CONV6F(1, 2, true, 4, true);
  else if (wt.mode == 2 && planes_out) { if (pool) DIM_CONV6F(1, 2, true); else DIM_CONV6F(0, 2, true); }
  else if (wt.mode == 2) { if (pool) DIM_CONV6F(1, 2, false); else DIM_CONV6F(0, 2, false); }
  else { if (pool) DIM_CONV6F(1, 1, false); else DIM_CONV6F(0, 1, false); }
#undef DIM_CONV6F
  DIM_LAUNCH_CHECK();
  return 0;
}

// fp16x3 convolution whose input and / or output are pre-split fp16 planes (see PIN / POUT above)
int launch_conv3x3_x6_planes(const float* in, const SplitWeights& wt, const float* bias, float* out, int batch, int H, int W, int cin,
                             int cout, int pool, int relu, int planes_in, int planes_out, hipStream_t s, unsigned* sat) {
  DIM_REQUIRE(cout % 64 == 0 && (cin == 64 || cin == 128), "conv3x3_x6 planes: cin=%d cout=%d", cin, cout);
  DIM_REQUIRE(wt.dev && wt.mode == 2, "conv3x3_x6 planes: fp16x3 weights required (mode %d)", wt.mode);
  DIM_REQUIRE(!planes_out || relu, "conv3x3_x6 planes: a pre-split output implies ReLU (its clamp starts at 0)");
  if (batch <= 0 || H <= 0 || W <= 0) return 0;
  const int var = dim_conv_x6_variant();
  // 16-row tiles unless they leave most of the chip idle: ONE image per call (the plugin hooks) gives the 128 x 128 maps of conv4a .. convDa 8 x 4 tiles
  // x 2 channel blocks = 64 workgroups on 256 CUs; 8-row tiles double that (same K order per output: bit-identical; the batched path never gets here)
  const bool big = planes_in && (var & 16) && (long)cdiv(W, TW) * cdiv(H, tile_rows(4)) * (cout / 64) * batch >= 256;
  const int mr = big ? 4 : 2;
  const int tiles_x = cdiv(W, TW), tiles_y = cdiv(H, tile_rows(mr));
  dim3 grid(tiles_x * tiles_y, cout / 64, batch);
#define DIM_CONV6P(CI, P, PI, PO, ...) hipLaunchKernelGGL(HIP_KERNEL_NAME(conv3x3_x6_kernel<CI, P, 1, false, 2, PI, PO, ##__VA_ARGS__>), grid, dim3(256), 0, s, in, wt.dev, bias, out, H, W, cout, relu, tiles_x, (const float*)nullptr, (const float*)nullptr, wt.inv_ch(), sat, (unsigned*)nullptr)
#define DIM_CONV6P_IO(CI, P)                                   \
  {                                                            \
    if (big && planes_out) DIM_CONV6P(CI, P, true, true, 4, true);   \
    else if (big) DIM_CONV6P(CI, P, true, false, 4, true);           \
    else if (planes_in && planes_out) DIM_CONV6P(CI, P, true, true); \
    else if (planes_in) DIM_CONV6P(CI, P, true, false);        \
    else if (planes_out) DIM_CONV6P(CI, P, false, true);       \
    else DIM_CONV6P(CI, P, false, false);                      \
  }
  if (cin == 64 && pool) DIM_CONV6P_IO(64, 1)
  else if (cin == 64) DIM_CONV6P_IO(64, 0)
  else if (pool) DIM_CONV6P_IO(128, 1)
  else DIM_CONV6P_IO(128, 0)
#undef DIM_CONV6P_IO
#undef DIM_CONV6P
  DIM_LAUNCH_CHECK();
  return 0;
}

int launch_planes_to_f32(const void* planes, int batch, int hw, int channels, float* out, hipStream_t s) {
  const size_t n = (size_t)batch * hw * channels;
  if (n == 0) return 0;
  hipLaunchKernelGGL(planes_to_f32_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, (const unsigned short*)planes, channels, hw, out, n);
  DIM_LAUNCH_CHECK();
  return 0;
}
