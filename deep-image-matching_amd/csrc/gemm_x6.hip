// fp32-accurate GEMM on the bf16 matrix cores ("bf16x6"): every fp32 operand is split EXACTLY into three
// bf16 pieces (dim_common.h: split3) and the six leading cross terms are accumulated in fp32 by
// v_mfma_f32_32x32x16_bf16.  Accuracy is fp32-class (hardware probe: 1.3e-7 of sum|a*b| at K = 1024, an
// fp32 fmaf chain gives 1.2e-7) while the MFMA cost per fp32-equivalent 32x32x16 step drops from
// 8 x 64 = 512 cycles (v_mfma_f32_32x32x2_f32) to 6 x 32 = 192 cycles: an effective dense peak of
// 2.5 PF / 6 = 417 TFLOP/s instead of 157.
//
// Same tiling as gemm.hip (128x128 block, 4 waves of 64x64, K chunks of 32).  Activations are split
// while they are staged into LDS ([plane][128][40 bf16]: the 80-byte row stride makes every 16-lane
// group of a ds_read_b128 cover all 64 banks exactly once).  Weights are pre-split on the host and
// stored in HBM directly in MFMA-fragment order, [plane][N/32][K/16][k-half][32 cols][8 k], so a
// wave fetches a B operand with ONE fully coalesced 1-KiB global load and the weights never touch
// LDS (they are L2-resident: at most 1.5 MB per layer).
#include <math.h>
#include <string.h>

#include <vector>

#include "dim_kernels.h"

namespace {
constexpr int KC = 32, RS = 20;  // RS: row stride in dwords (40 bf16)
// BM = 128 (each wave 64 x 64) or, for small problems that would leave most CUs idle, 64 (each wave 32 x 64: twice the
// workgroups, half the MFMA / split work per barrier-to-barrier step of the serial K loop)

// NT: 32-column MFMA tiles per wave.  2: block 128 x 128 (3 workgroups per CU); 4: block 128 x 256, each wave 64 x 128 =
//     8 accumulators (2 per CU) — the activation tile is loaded, split and staged once for twice the MFMAs, and the K loop's
//     two barriers and its exposed prefetch latency are paid per 48 instead of per 24 MFMAs.
// KV (fp16x3, NT 4 only): the q|k|v projection of LightGlue.  The workgroups of column block a.kv_kblock issue their MFMAs with
//     the operands SWAPPED (C^T = W^T X^T: lane = key, registers = 16 output dims), apply the rotary embedding to adjacent
//     dims, split, and store each lane's registers 8j .. 8j+7 as one 16-byte slot of the attention kernel's K tile image;
//     the workgroups of block a.kv_vblock keep the normal orientation — the MFMA C layout of a 32-key tile IS the key
//     order of the V image — and store registers 8u .. 8u+7 of every (dim, step u) as one slot.  No fp32 k / v, no
//     separate pre-split pass (kv_prep_kernel: 155 us per launch, 0.85 GB of traffic).
// KV 1 / 2: the body for column block kv_kblock / kv_vblock (separate instantiations selected per workgroup by
// gemm_x6_qkv_kernel: both MFMA orientations, or two epilogues, inside ONE loop body spill 30 .. 800 registers).
// WN: waves along N.  2: waves 2 x 2, each (BM / 2) rows x 32 NT columns.  4 (the 128 x 256 block): waves 1 x 4, each ALL 128
//     rows x 64 columns (MT 4 x NT 2 = 8 accumulators): a weight fragment is then fetched by exactly one wave — with 2 x 2
//     waves of 64 x 128 every fragment was fetched twice per workgroup and the per-CU vector-memory path (64 B / clk), not
//     the matrix pipe or HBM latency, was what the kernel ran at (an experiment that made every activation prefetch an L2
//     hit changed nothing).  The activation fragments come from LDS, which has the bandwidth to feed 4 waves.
// PROBE (timing experiments only, scripts/gpu_gemm_probe.py; 0 in every product instantiation): bit 0 = the activation rows of all
// workgroups come from the first 2048 rows (cache-resident), bit 1 = every chunk re-reads the weight fragments of chunk 0 (L1 hits),
// bit 2 = no MFMAs, bit 3 = the weight fragments are loaded once, before the K loop.  Results are wrong by design.
// KCH: K values per staged chunk.  32 everywhere in the product; 64 (prototype, dim_tune_set key 14, pipelined wide blocks only) halves the
// number of barrier pairs and staging round trips per MFMA for 16 more prefetch registers and a 36-dword row stride (also conflict-free).
#ifdef DIM_FFN_TIMERS
// Instrumented build only (build.build_variant("ffntime", ["-DDIM_FFN_TIMERS"]); scripts/gpu_ffn_phases.py): s_memtime stamps (100 MHz; gfx950 has
// no SHADER_CYCLES register; the stamp's lgkmcnt(0) wait sits where a barrier or a consumed LDS read waits anyway) at the phase boundaries of the
// fused feed-forward, summed over every wave of every workgroup.
__device__ unsigned long long g_ffn_phase[24];
__device__ __forceinline__ unsigned ft_now() {
  __builtin_amdgcn_sched_barrier(0);
  const unsigned v = (unsigned)__builtin_readcyclecounter();
  __builtin_amdgcn_sched_barrier(0);
  return v;
}
#define FT_DECL unsigned ft_ph[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}; unsigned ft_last = ft_now(); const unsigned long long ft_t0 = __builtin_readcyclecounter(), ft_r0 = __builtin_amdgcn_s_memrealtime();
#define FT_TICK(p) do { if (KV == 4) { const unsigned ft_n = ft_now(); ft_ph[p] += ft_n - ft_last; ft_last = ft_n; } } while (0)
// one workgroup in 16 reports (230 000 waves adding to the same 18 words would stretch the kernel)
#define FT_FLUSH() do { if (KV == 4 && (threadIdx.x & 63) == 0 && (blockIdx.x & 15) == 0) { \
    const unsigned long long ft_t1 = __builtin_readcyclecounter() - ft_t0, ft_r1 = __builtin_amdgcn_s_memrealtime() - ft_r0; \
    for (int i = 0; i < 16; ++i) atomicAdd(&g_ffn_phase[i], (unsigned long long)ft_ph[i]); \
    atomicAdd(&g_ffn_phase[16], ft_t1); atomicAdd(&g_ffn_phase[17], 1ull); atomicAdd(&g_ffn_phase[18], ft_r1); } } while (0)
#else
#define FT_DECL
#define FT_TICK(p)
#define FT_FLUSH()
#endif
// DB (prototype, dim_tune_set key 14 = 33, pipelined blocks only): the staged activation tile is double-buffered in LDS — chunk c + 1 is split and
// written into the other half BEFORE the MFMAs of chunk c, one barrier per chunk instead of two, and the store -> barrier -> fragment-read
// latency chain leaves the critical path.
// ROLL (the fused feed-forward's K loop since round 4; dim_tune_set key 14 = 36 selects the previous one-k-step-at-a-time loop): the weight fragments of column tile n are
// re-requested for the NEXT k-step right after this step's MFMAs on tile n have been issued — every request has the other tiles' MFMAs (3/4 of a
// step) in front of its use, in the same 32 registers; the activation prefetch goes out between the chunk's two steps (requests retire in order).
// STREAM (round 5 prototype, research build only: dim_tune_set key 14 = 63 on the 64 x 128 block of SMALL problems — one pair per call through the plugin
// hooks, fp16x3): no LDS, no barriers.  The idea: a launch with fewer workgroups than the chip has slots runs for as long as ONE workgroup's serial chain,
// and the staged loop's chain is K / 32 x {wait for the activation prefetch, split, ds_write, barrier, wait for the weight fragments, ds_read, 12 MFMAs,
// barrier}.  Here every wave reads the fp32 rows of ITS 32 x 16 A fragment straight from global memory (a lane's 8 consecutive k values = two 16-byte
// loads), splits them in registers into the operand the LDS image would have held, and keeps DIM_STREAM_D k-steps of activations AND weight fragments in
// flight in a register ring (vmcnt(18..22) in the ISA).  Same pieces, same per-accumulator term order, same epilogue: bit-identical results.  MEASURED
// (profiles/r05_ab_small_gemm_stream.jsonl, 4096 rows): 16.6 vs 13.3 us (256 -> 768), 9.8 vs 9.1 (256 -> 256 + residual), 16.9 vs 15.4 (512 -> 512),
// 15.8 vs 15.2 (512 -> 256 + residual); a ring of 8 k-steps is slower still.  The chain was not the bound: both loops cost ~0.38 us per 16-wide k-step
// per workgroup, i.e. the ~24 KB per k-step that the workgroup's four waves pull through the CU's 64 B / clk vector-memory path (every weight fragment
// twice, and here every activation row twice and in half-used cache lines), and a launch is ~4.7 us before it does anything (the duration rocprofv3
// reports for the one-workgroup lg_decide_kernel).  Kept for the record.
#ifndef DIM_STREAM_D
#define DIM_STREAM_D 4
#endif
template <int MODE, int BM, int NT, int KV, int WN, int PROBE = 0, bool PIPE = false, int KCH = 32, bool DB = false, bool ROLL = false, bool STREAM = false, int STREAM_KS = 0, bool BSET = false, int K_T = 0>
__device__ __forceinline__ void gemm_x6_body(const GemmArgs& a, unsigned* Ap, int by) {
  using S = SplitMma<MODE>;
  constexpr int WM = 4 / WN, MT = BM / (32 * WM);  // 32-row MFMA tiles per wave
  constexpr int NPL = S::NPL, NLD = BM * KCH / 1024;  // float4 loads per thread and chunk
  constexpr int KC = KCH, RS = KCH / 2 + 4, Q4_SHIFT = KCH == 32 ? 3 : 4, KSTEPS = KCH / 16;   // (shadow the file-level 32-wide constants)
  static_assert(KCH == 32 || (KCH == 64 && PIPE && PROBE == 0), "64-wide chunks exist for the pipelined K loop");
  static_assert(!DB || (PIPE && KCH == 32 && PROBE == 0), "the double-buffered tile exists for the pipelined 32-wide K loop");
  static_assert(!BSET || (PIPE && KCH == 64 && !DB && !ROLL && !STREAM && PROBE == 0 && NT * (KCH / 16) <= 4), "chunk-level fragment sets exist for the pipelined 64-wide K loop of one-tile waves");
  static_assert(!ROLL || (((KV == 3 || KV == 4) && !PIPE) || (PIPE && KCH == 32 && !DB)) && (PROBE & ~1) == 0, "rolling fragment requests exist for the 64 x 512 blocks and (prototype) the pipelined 32-wide K loop");
  constexpr int ABUF = NPL * BM * RS;   // dwords of one staged activation tile
  constexpr int BN = 32 * NT * WN;
  static_assert(KV == 0 || KV == 3 || KV == 4 || KV == 5 || (MODE == 2 && ((BN == 256 && (BM == 128 || BM == 256)) || (BN == 128 && BM == 32))), "the K|V image epilogue exists for the fp16x3 128 / 256 x 256 blocks and the one-pair 32 x 128 block");
  static_assert(KV != 5 || (MODE == 2 && BM == 128 && NT == 3 && WN == 1), "the detector-head epilogue exists for the fp16x3 128 x 96 block (4 waves x 32 cells, all 65 channels per wave)");
  static_assert(KV != 4 || (MODE == 2 && BN == 512 && (BM == 64 || BM == 128) && WN == 4 && NT == 4), "the fused ffn exists for the fp16x3 64 / 128 x 512 blocks");
  static_assert(KV != 3 || (MODE == 2 && BN == 512 && BM == 64 && WN == 4), "the LayerNorm + GELU epilogue exists for the fp16x3 64 x 512 block");
  static_assert(!STREAM || (MT == 1 && KV == 0 && PROBE == 0 && !PIPE && !DB && !ROLL && KCH == 32), "the streaming K loop exists for one 32-row tile per wave and the plain epilogue");
  const int z = blockIdx.z;
  if (a.flag && a.flag[z >> a.flag_shift] != a.flag_eq) return;
  const int rows = a.rows ? a.rows[z * a.rows_mul + a.rows_off] * a.rows_scale : a.M;
  const int m0 = blockIdx.x * BM, n0 = by * BN;
  if (m0 >= rows || n0 >= a.N) return;
  constexpr bool ffn = KV == 4, kblk = KV == 1 || ffn, vblk = KV == 2, lng = KV == 3;   // (ffn computes its hidden tile transposed, like the K image)

  const int t = threadIdx.x;
  const int lane = t & 63, wv = t >> 6, wm = wv / WN, wn = wv % WN, lx = lane & 31, half = lane >> 5;
  const float* A0 = a.A0 + (size_t)(a.a_idx ? a.a_idx[z] : z) * a.strideA0;
  const float* A1 = a.A1 ? a.A1 + (size_t)z * a.strideA1 : nullptr;
  const int NB = a.n_pad / 32, KS = a.K / 16;
  // this lane's slot inside a [k-half][32 cols] fragment block, for its two 32-column groups
  const u32x4* Bf = (const u32x4*)a.Bx3;
  const size_t nb0 = (size_t)(n0 + wn * (32 * NT)) / 32;

  f32x16 acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

  FT_DECL
  float4 ra[NLD];
  auto load_chunk_r = [&](float4 (&ra)[NLD], int k0) {
    const float* src; int ld, kk0;
    if (A1 == nullptr || k0 < a.ksplit) { src = A0; ld = a.lda0; kk0 = k0; }
    else { src = A1; ld = a.lda1; kk0 = k0 - a.ksplit; }
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
      const int idx = t + 256 * i, row = idx >> Q4_SHIFT, q = idx & ((1 << Q4_SHIFT) - 1);
      // rows past the ragged end re-read the last valid row (always mapped): GEMM rows are independent and the
      // epilogue never stores them, so they need no zeroing — no branch, and no VALU touching the prefetch
      // registers before the split (anything earlier would drag the wait for them into the MFMA phase)
      ra[i] = *(const float4*)(src + (size_t)((PROBE & 1) ? ((m0 + row) & 2047) : min(m0 + row, rows - 1)) * ld + kk0 + q * 4);
    }
  };
  auto load_chunk = [&](int k0) { load_chunk_r(ra, k0); };
  auto store_chunk_r = [&](const float4 (&ra)[NLD], int abuf) {
    unsigned* const Ab = Ap + abuf * ABUF;
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
      const int idx = t + 256 * i, row = idx >> Q4_SHIFT, q = idx & ((1 << Q4_SHIFT) - 1);
      unsigned p0[NPL], p1[NPL];
      if (PROBE & 64) {   // probe bit 6: no split arithmetic — the raw bits go to LDS (what activations pre-split by their producer would at most save)
#pragma unroll
        for (int pl = 0; pl < NPL; ++pl) { p0[pl] = __float_as_uint(pl ? ra[i].y : ra[i].x); p1[pl] = __float_as_uint(pl ? ra[i].w : ra[i].z); }
      } else {
        S::split(ra[i].x, ra[i].y, S::act_scale(), p0);
        S::split(ra[i].z, ra[i].w, S::act_scale(), p1);
      }
      unsigned* d = &Ab[row * RS + q * 2];
#pragma unroll
      for (int pl = 0; pl < NPL; ++pl) { d[pl * BM * RS] = p0[pl]; d[pl * BM * RS + 1] = p1[pl]; }
    }
  };
  auto store_chunk = [&](int abuf = 0) { store_chunk_r(ra, abuf); };

  // one 16-wide k-step of the chunk staged in LDS against the weight fragments fbk[n][plane]
  auto mma_step = [&](int ks, const u32x4 (&fbk)[NT][NPL], int abuf = 0) {
    const unsigned* const Ab = Ap + abuf * ABUF;
    u32x4 fa[MT][NPL];
#pragma unroll
    for (int p = 0; p < NPL; ++p)
#pragma unroll
      for (int m = 0; m < MT; ++m) fa[m][p] = *(const u32x4*)&Ab[(p * BM + wm * (32 * MT) + m * 32 + lx) * RS + ks * 8 + half * 4];
    // cross terms smallest first; the accumulators interleave so no MFMA waits on its predecessor
    if (PROBE & 4) {  // no MFMAs: keep the operands alive
#pragma unroll
      for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int n = 0; n < NT; ++n) acc[m][n][0] += __uint_as_float(fa[m][0][0] ^ fbk[n][1][0]);
    } else if (kblk) {  // transposed tiles: the weights are the A operand
#pragma unroll
      for (int tm = 0; tm < S::NT; ++tm)
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
          for (int n = 0; n < NT; ++n) acc[m][n] = S::mma(fbk[n][S::tb(tm)], fa[m][S::ta(tm)], acc[m][n]);
    } else {
#pragma unroll
      for (int tm = 0; tm < S::NT; ++tm)
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
          for (int n = 0; n < NT; ++n) acc[m][n] = S::mma(fa[m][S::ta(tm)], fbk[n][S::tb(tm)], acc[m][n]);
    }
  };
  auto load_b = [&](int kstep, u32x4 (&fbk)[NT][NPL]) {
#pragma unroll
    for (int p = 0; p < NPL; ++p)
#pragma unroll
      for (int n = 0; n < NT; ++n) fbk[n][p] = Bf[((((size_t)p * NB + nb0 + n) * KS + kstep) * 2 + half) * 32 + lx];
  };

  // ROLL: one k-step with the column tiles outermost; tile n's fragments are replaced by those of k-step `next` as soon as its MFMAs are issued
  // (per accumulator the cross terms keep their order: lh, hl, hh of step s, then of step s + 1 — bit-identical results)
  auto mma_roll = [&](int ks, u32x4 (&fbk)[NT][NPL], int next) {
    u32x4 fa[MT][NPL];
#pragma unroll
    for (int p = 0; p < NPL; ++p)
#pragma unroll
      for (int m = 0; m < MT; ++m) fa[m][p] = *(const u32x4*)&Ap[(p * BM + wm * (32 * MT) + m * 32 + lx) * RS + ks * 8 + half * 4];
#pragma unroll
    for (int n = 0; n < NT; ++n) {
#pragma unroll
      for (int tm = 0; tm < S::NT; ++tm)
#pragma unroll
        for (int m = 0; m < MT; ++m)
          acc[m][n] = kblk ? S::mma(fbk[n][S::tb(tm)], fa[m][S::ta(tm)], acc[m][n]) : S::mma(fa[m][S::ta(tm)], fbk[n][S::tb(tm)], acc[m][n]);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int p = 0; p < NPL; ++p) fbk[n][p] = Bf[((((size_t)p * NB + nb0 + n) * KS + next) * 2 + half) * 32 + lx];
      __builtin_amdgcn_sched_barrier(0);
    }
  };

  if constexpr (STREAM) {
    constexpr int STREAM_D = DIM_STREAM_D;   // k-steps in flight, each 8 + 16 registers
    const int arow = min(m0 + wm * 32 + lx, rows - 1);   // rows past the ragged end re-read the last valid row (never stored)
    const float* const ap0 = A0 + (size_t)arow * a.lda0 + half * 8;
    const float* const ap1 = A1 ? A1 + (size_t)arow * a.lda1 + half * 8 : nullptr;
    float4 ar[STREAM_D][2];
    u32x4 br[STREAM_D][NT][NPL];
    auto fetch = [&](int ks, float4 (&x)[2], u32x4 (&f)[NT][NPL]) {
      const int k0 = ks * 16;
      const float* src = (ap1 == nullptr || k0 < a.ksplit) ? ap0 + k0 : ap1 + (k0 - a.ksplit);
      x[0] = *(const float4*)src;
      x[1] = *(const float4*)(src + 4);
      load_b(ks, f);
    };
#pragma unroll
    for (int d = 0; d < STREAM_D; ++d) fetch(min(d, KS - 1), ar[d], br[d]);
    __builtin_amdgcn_sched_barrier(0);
    // STREAM_KS (16, 32 = K 256, 512: every linear of LightGlue): the loop is unrolled completely — in straight-line code the compiler sizes every
    // wait for exactly the request it needs (vmcnt = what was issued after it); at a loop header it waits for everything in flight, i.e. the ring
    // would drain once per round.  0: run-time K, that loop.
    const int ksn = STREAM_KS ? STREAM_KS : KS;
#pragma unroll
    for (int ks0 = 0; ks0 < ksn; ks0 += STREAM_D) {
#pragma unroll
      for (int d = 0; d < STREAM_D; ++d) {
        if (ks0 + d < ksn) {   // (wave-uniform)
          unsigned pc[4][NPL];
          S::split(ar[d][0].x, ar[d][0].y, S::act_scale(), pc[0]);
          S::split(ar[d][0].z, ar[d][0].w, S::act_scale(), pc[1]);
          S::split(ar[d][1].x, ar[d][1].y, S::act_scale(), pc[2]);
          S::split(ar[d][1].z, ar[d][1].w, S::act_scale(), pc[3]);
          u32x4 fa[NPL];
#pragma unroll
          for (int pl = 0; pl < NPL; ++pl) fa[pl] = u32x4{pc[0][pl], pc[1][pl], pc[2][pl], pc[3][pl]};
#pragma unroll
          for (int tm = 0; tm < S::NT; ++tm)
#pragma unroll
            for (int n = 0; n < NT; ++n) acc[0][n] = S::mma(fa[S::ta(tm)], br[d][n][S::tb(tm)], acc[0][n]);
          __builtin_amdgcn_sched_barrier(0);   // (the scheduler otherwise sinks every request to just in front of its use: 76 registers, vmcnt(0) everywhere)
          if (!STREAM_KS || ks0 + d + STREAM_D < STREAM_KS) fetch(min(ks0 + d + STREAM_D, KS - 1), ar[d], br[d]);   // (run-time K: the tail harmlessly re-reads the last k-step)
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    }
  } else {
  load_chunk(0);
  const int k_end = (PROBE & 32) ? KC : a.K;   // probe bit 5: one K chunk only
  if (PIPE && ROLL) {
    // prototype (dim_tune_set key 14 = 37): the two fragment sets of the pipelined loop refilled tile by tile — set 0 with step s + 2 during
    // step s, set 1 with step s + 3 during step s + 1: every request ~1.75 steps (42 MFMAs) in front of its use instead of one (24), same 32
    // registers; rotated like the fused feed-forward's loop (the wait in front of the split is sized in straight-line code).
    u32x4 fb0[NT][NPL], fb1[NT][NPL];
    load_b(0, fb0);
    load_b(1, fb1);
    store_chunk();
    for (int k0 = 0; k0 < a.K; k0 += KC) {
      __syncthreads();
      const int kst = k0 >> 4;
      mma_roll(0, fb0, min(kst + 2, KS - 2));   // (the last chunk harmlessly re-reads its own fragments)
      __builtin_amdgcn_sched_barrier(0);
      load_chunk(min(k0 + KC, a.K - KC));
      __builtin_amdgcn_sched_barrier(0);
      mma_roll(1, fb1, min(kst + 3, KS - 1));
      __syncthreads();
      if (k0 + KC < a.K) store_chunk();
    }
  } else if (PIPE && BSET) {
    // BSET (round 6; the one-pair 32 x 128 blocks): weight fragments AND activations requested a whole chunk period ahead of their use.  A launch of one LightGlue
    // pair is bound by the latency of its L2 (first touch: MALL) requests times the bytes it keeps in flight — a wave of the step-pipelined loop above has ONE k-step
    // of fragments (2 KB) and one chunk of activations outstanding, ~1/3 of what the CU's 64 B / clk path needs at ~800 clocks of latency (a 32 x 512 block that
    // streams all of ffn.0's weights through one workgroup per CU runs at 25 B / clk: 26 us), and only 3 MFMAs between a fragment request and its use.  With one
    // 32-column tile per wave a whole 64-wide chunk of fragments is 32 registers.  Iteration c: split + stage A(c) [requested in iteration c - 2], barrier, request
    // A(c + 2) and then the fragment set of chunk c + 1, the MFMAs of chunk c [its set was requested in iteration c - 1; vector-memory requests retire in order, so
    // the wait in front of them retires A(c + 1) as well and leaves exactly this iteration's requests in flight], barrier.  Two chunks per loop trip (the register
    // sets swap roles; K must be a multiple of 128).  Same pieces, same per-accumulator term order: bit-identical.
    // K_T (256, 512: every linear of LightGlue) unrolls the loop completely: in straight-line code the compiler sizes every wait for exactly the request it needs;
    // at the header of a real loop it merges the counts of the entry and the back edge and waits for (nearly) everything in flight.  0: run-time K, that loop.
    u32x4 fs0[KSTEPS][NT][NPL], fs1[KSTEPS][NT][NPL];
    float4 rb[NLD];   // activations of the odd chunks (ra: the even ones)
    const int Kt = K_T ? K_T : a.K;
    __builtin_amdgcn_sched_barrier(0);   // (A(0) was requested above: keep it the oldest request)
#pragma unroll
    for (int ks = 0; ks < KSTEPS; ++ks) load_b(ks, fs0[ks]);
    __builtin_amdgcn_sched_barrier(0);
    load_chunk_r(rb, min(KC, Kt - KC));
    __builtin_amdgcn_sched_barrier(0);
    auto chunk = [&](int k0, float4 (&ac)[NLD], const u32x4 (&cur)[KSTEPS][NT][NPL], u32x4 (&nxt)[KSTEPS][NT][NPL]) {
      store_chunk_r(ac, 0);
      __syncthreads();
      load_chunk_r(ac, min(k0 + 2 * KC, Kt - KC));   // (the tail harmlessly re-reads the last chunk)
      __builtin_amdgcn_sched_barrier(0);
      const int kn = min(k0 + KC, Kt - KC);
#pragma unroll
      for (int ks = 0; ks < KSTEPS; ++ks) load_b((kn >> 4) + ks, nxt[ks]);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int ks = 0; ks < KSTEPS; ++ks) mma_step(ks, cur[ks]);
      __syncthreads();
    };
    if constexpr (K_T != 0) {
#pragma unroll
      for (int k0 = 0; k0 < K_T; k0 += 2 * KC) { chunk(k0, ra, fs0, fs1); chunk(k0 + KC, rb, fs1, fs0); }
    } else {
      for (int k0 = 0; k0 < a.K; k0 += 2 * KC) { chunk(k0, ra, fs0, fs1); chunk(k0 + KC, rb, fs1, fs0); }
    }
  } else if (PIPE && DB) {
    u32x4 fb0[NT][NPL], fb1[NT][NPL];
    store_chunk(0);
    load_chunk(min(KC, a.K - KC));
    load_b(0, fb0);
    __syncthreads();
    int abuf = 0;
    for (int k0 = 0; k0 < a.K; k0 += KC, abuf ^= 1) {
      const int kst = k0 >> 4;
      load_b(kst + 1, fb1);
      __builtin_amdgcn_sched_barrier(0);
      store_chunk(abuf ^ 1);                      // chunk c + 1 (held in ra since the previous iteration) into the other half
      __builtin_amdgcn_sched_barrier(0);
      load_chunk(min(k0 + 2 * KC, a.K - KC));     // chunk c + 2 (the tail harmlessly re-reads the last chunk)
      __builtin_amdgcn_sched_barrier(0);
      mma_step(0, fb0, abuf);
      __builtin_amdgcn_sched_barrier(0);
      load_b(min(kst + 2, KS - 2), fb0);
      __builtin_amdgcn_sched_barrier(0);
      mma_step(1, fb1, abuf);
      __syncthreads();   // everyone has read this half and written the other
    }
  } else if (PIPE) {
    // Weight fragments double-buffered per k-STEP: the fragments of step s + 1 are requested before the MFMAs of step s, so the
    // (L2) latency of every request hides behind 3 * MT * NT MFMAs instead of standing in front of them twice per chunk, in the
    // registers the one-chunk-at-a-time form already used (2 steps x NT x NPL).  Vector-memory loads retire in order: the next
    // chunk's activation prefetch is issued AFTER the fragments of this chunk's second step and BEFORE those of the next chunk's
    // first step, which are not waited for until after the next barrier — by then the prefetch is needed anyway.
    u32x4 fb0[NT][NPL], fb1[NT][NPL];
    load_b(0, fb0);
    for (int k0 = 0; k0 < k_end; k0 += KC) {
      store_chunk();
      __syncthreads();
      const int kst = (PROBE & 2) ? 0 : (k0 >> 4);
      if (!(PROBE & 8)) load_b(kst + 1, fb1);
      __builtin_amdgcn_sched_barrier(0);
      load_chunk(min(k0 + KC, a.K - KC));   // unconditional: see below
      __builtin_amdgcn_sched_barrier(0);
      mma_step(0, fb0);
      __builtin_amdgcn_sched_barrier(0);
      if (!(PROBE & 8)) load_b(min(kst + 2, KS - 2), fb0);   // the last chunk harmlessly re-reads its own first step
      __builtin_amdgcn_sched_barrier(0);
      mma_step(1, (PROBE & 8) ? fb0 : fb1);
      if (KSTEPS == 4) {   // 64-wide chunk: two more steps, same alternation (step 2 from fb0, step 3 from fb1)
        __builtin_amdgcn_sched_barrier(0);
        load_b(min(kst + 3, KS - 1), fb1);
        __builtin_amdgcn_sched_barrier(0);
        mma_step(2, fb0);
        __builtin_amdgcn_sched_barrier(0);
        load_b(min(kst + 4, KS - 4), fb0);                   // the next chunk's first step
        __builtin_amdgcn_sched_barrier(0);
        mma_step(3, fb1);
      }
      __syncthreads();
    }
  } else if (ROLL) {
    // (rotated: the split + store of chunk c + 1 closes the iteration, so that the wait in front of it is sized in straight-line code — it
    // leaves the eight fragment requests of the second step in flight; at a loop header the compiler waits for everything)
    u32x4 fbr[NT][NPL];
    load_b(0, fbr);
    FT_TICK(13);
    store_chunk();
    FT_TICK(0);
    for (int k0 = 0; k0 < a.K; k0 += KC) {
      __syncthreads();
      FT_TICK(1);
      const int kst = k0 >> 4;
      mma_roll(0, fbr, kst + 1);
      __builtin_amdgcn_sched_barrier(0);
      load_chunk(min(k0 + KC, a.K - KC));
      __builtin_amdgcn_sched_barrier(0);
      mma_roll(1, fbr, min(kst + 2, KS - 1));   // the last step harmlessly re-reads its own fragments
      FT_TICK(2);
      __syncthreads();
      FT_TICK(3);
      if (k0 + KC < a.K) store_chunk();
      FT_TICK(0);
    }
  } else
  for (int k0 = 0; k0 < k_end; k0 += KC) {
    if (k0 == 0) FT_TICK(13);
    store_chunk();
    FT_TICK(0);
    __syncthreads();
    FT_TICK(1);
    // Issue order matters: vector-memory loads retire in order, so the weight fragments this chunk's MFMAs
    // need are requested BEFORE the next chunk's activation prefetch — the wait in front of the first MFMA
    // then covers the (L2-resident) fragments only and the HBM latency of the prefetch hides behind the MFMAs.
    // (the 64 x 512 LayerNorm block holds 8 accumulators AND 4 column tiles of fragments per step: it requests one k-step's
    // fragments at a time — 32 instead of 64 registers — so that the accumulators can live in the AGPR half of the file)
    constexpr int FKS = (lng || ffn) ? 1 : 2;
    u32x4 fb[FKS][NT][NPL];
    const int kf0 = (PROBE & 2) ? 0 : (k0 >> 4);
    if (!(PROBE & 8) || k0 == 0) {
#pragma unroll
      for (int ks = 0; ks < FKS; ++ks) load_b(kf0 + ks, fb[ks]);
    }
    // unconditional (the last iteration harmlessly re-reads its own chunk): a branch here would make the compiler
    // size every wait in the MFMA phase for the path WITHOUT the prefetch, i.e. wait for the prefetch on the other
    __builtin_amdgcn_sched_barrier(0);
    load_chunk(min(k0 + KC, a.K - KC));
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      if ((lng || ffn) && ks == 1 && !(PROBE & 8)) load_b(kf0 + 1, fb[0]);
      mma_step(ks, fb[(lng || ffn) ? 0 : ks]);
    }
    FT_TICK(2);
    __syncthreads();
    FT_TICK(3);
  }
  }   // (!STREAM)

  if constexpr (ffn) {
    // ================= LightGlue's whole feed-forward in one workgroup (LGN:141-142,159,209): the 64 x 512 hidden tile never
    // leaves the CU.  The K loop above produced it TRANSPOSED (weights as the A operand): lane = token lx (+ 32 m), register r of
    // tile n = hidden unit 128 wn + 32 n + (r & 3) + 8 (r >> 2) + 4 half — exactly the operand layout of the next product
    // (lane = row, 8 consecutive registers = the k values of one lane half), so after LayerNorm + GELU the tile is split in
    // place into fp16 pieces and fed to ffn.3 as the A operand straight from registers.  The k ORDER inside a 16-step differs
    // from the natural one (register j of half h is unit 8 (j >> 2) + 4 h + (j & 3) of the step); ffn.3's weight fragments are
    // packed with the same permutation on the host (split_weights, kperm).  Every wave holds 128 of the 512 hidden units, i.e. a
    // K-slice of ffn.3: the four partial 64 x 256 outputs are reduced through LDS, 64 columns at a time, each wave ending up
    // with one 32 x 32 tile that it finishes (scale, bias, residual, range guard) and stores. =================
    constexpr int TOK = 32 * MT;                       // tokens (rows) of the block: 64, or 128 in the 512-register variant
    float* const prm = (float*)(Ap + NPL * BM * RS);   // [inv0 | bias0 | gamma | beta] x 512 — filled before the K loop
    float* const red = prm + 2048;                     // [2 passes][4 waves][TOK tokens]
    float* const xbuf = red + 8 * TOK;                 // [4 owner tiles][3 sources][4 register quads][64 lanes][4]
    // ---- v = acc * inv + bias in place; LayerNorm statistics per token: registers + lane halves + the four waves ----
    float part[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) part[m] = 0.0f;
#pragma unroll
    for (int n = 0; n < NT; ++n)
#pragma unroll
      for (int rq = 0; rq < 4; ++rq) {
        const int ub = wn * 128 + n * 32 + 8 * rq + 4 * half;
        const float4 iv = *(const float4*)(prm + ub), bv = *(const float4*)(prm + 512 + ub);
#pragma unroll
        for (int m = 0; m < MT; ++m) {
          acc[m][n][4 * rq + 0] = fmaf(acc[m][n][4 * rq + 0], iv.x, bv.x);
          acc[m][n][4 * rq + 1] = fmaf(acc[m][n][4 * rq + 1], iv.y, bv.y);
          acc[m][n][4 * rq + 2] = fmaf(acc[m][n][4 * rq + 2], iv.z, bv.z);
          acc[m][n][4 * rq + 3] = fmaf(acc[m][n][4 * rq + 3], iv.w, bv.w);
          part[m] += (acc[m][n][4 * rq + 0] + acc[m][n][4 * rq + 1]) + (acc[m][n][4 * rq + 2] + acc[m][n][4 * rq + 3]);
        }
      }
    float mean[MT], rstd[MT];
    auto token_total = [&](float (&p)[MT], float* region, float (&out)[MT]) {   // sum over the 512 units of the lane's tokens
#pragma unroll
      for (int m = 0; m < MT; ++m) {
        const float v = half_pair_sum(p[m]);
        if (half == 0) region[wn * TOK + m * 32 + lx] = v;
      }
      __syncthreads();
#pragma unroll
      for (int m = 0; m < MT; ++m) out[m] = (region[m * 32 + lx] + region[TOK + m * 32 + lx]) + (region[2 * TOK + m * 32 + lx] + region[3 * TOK + m * 32 + lx]);
    };
    token_total(part, red, mean);
#pragma unroll
    for (int m = 0; m < MT; ++m) { mean[m] *= 1.0f / 512.0f; part[m] = 0.0f; }
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int n = 0; n < NT; ++n)
#pragma unroll
        for (int r = 0; r < 16; ++r) { const float d = acc[m][n][r] - mean[m]; part[m] = fmaf(d, d, part[m]); }
    token_total(part, red + 4 * TOK, rstd);
#pragma unroll
    for (int m = 0; m < MT; ++m) rstd[m] = 1.0f / sqrtf(rstd[m] * (1.0f / 512.0f) + 1e-5f);
    FT_TICK(4);
    // ---- normalise, GELU, range guard, split: tile (m, n) -> hq[m][n][plane][q] (q = which 8 registers = which k-step) ----
    u32x4 hq[MT][NT][NPL][2];
    float vmax = 0.0f;
#pragma unroll
    for (int n = 0; n < NT; ++n)
#pragma unroll
      for (int rq = 0; rq < 4; ++rq) {
        const int ub = wn * 128 + n * 32 + 8 * rq + 4 * half;
        const float4 gm = *(const float4*)(prm + 1024 + ub), bt = *(const float4*)(prm + 1536 + ub);
#pragma unroll
        for (int m = 0; m < MT; ++m) {
          const f32x2 mean2 = {mean[m], mean[m]}, rstd2 = {rstd[m], rstd[m]};
          const f32x2 half2 = {0.5f, 0.5f}, rt2 = {0.70710678118654752440f, 0.70710678118654752440f};
          f32x2 y0 = (f32x2{acc[m][n][4 * rq], acc[m][n][4 * rq + 1]} - mean2) * rstd2;
          f32x2 y1 = (f32x2{acc[m][n][4 * rq + 2], acc[m][n][4 * rq + 3]} - mean2) * rstd2;
          y0 = __builtin_elementwise_fma(y0, f32x2{gm.x, gm.y}, f32x2{bt.x, bt.y});
          y1 = __builtin_elementwise_fma(y1, f32x2{gm.z, gm.w}, f32x2{bt.z, bt.w});
          const f32x2 e0 = erf2_1ulp(y0 * rt2), e1 = erf2_1ulp(y1 * rt2);
          const f32x2 h0 = y0 * half2, h1 = y1 * half2;
          const f32x2 g0 = __builtin_elementwise_fma(h0, e0, h0), g1 = __builtin_elementwise_fma(h1, e1, h1);
          vmax = sat_track(sat_track(vmax, g0[0], g0[1]), g1[0], g1[1]);
          unsigned p0[NPL], p1[NPL];
          S::split(g0[0], g0[1], S::act_scale(), p0);
          S::split(g1[0], g1[1], S::act_scale(), p1);
#pragma unroll
          for (int pl = 0; pl < NPL; ++pl) { hq[m][n][pl][rq >> 1][2 * (rq & 1)] = p0[pl]; hq[m][n][pl][rq >> 1][2 * (rq & 1) + 1] = p1[pl]; }
        }
      }
    sat_report(a.sat, vmax);
    FT_TICK(5);
    // ---- ffn.3 over this wave's 128 hidden units (8 k-steps: step ks = tile n = ks / 2, registers 8 (ks & 1) ..), 64 output
    // columns per round; weight fragments one k-step ahead ----
    const u32x4* const Bf2 = (const u32x4*)a.B2x3;
    constexpr int NB2 = 8, KS2 = 32;   // 256 columns, 512 hidden units
    const dim_rsrc Cr = buf_rsrc(a.C + (size_t)z * a.strideC, ((size_t)(rows - 1) * a.ldc + 256) * sizeof(float));
    const dim_rsrc Rr = buf_rsrc(a.R + (size_t)z * a.strideR, ((size_t)(rows - 1) * a.ldr + 256) * sizeof(float));
    float vmax2 = 0.0f;
    // output tile of this wave in every round: rows 32 (wn >> 1) .., columns 64 grp + 32 (wn & 1) ..  One per-lane byte offset (first
    // row of the lane half, lane column); the register's row step and the round's column step are wave-uniform and ride in the
    // scalar offset — which the hardware range check ignores, so rows past the ragged end are predicated off explicitly
    const int orow = m0 + 32 * (wn >> 1) + 4 * half;
    const unsigned obase = (unsigned)orow * (unsigned)a.ldc * 4u + (unsigned)(32 * (wn & 1) + lx) * 4u;
    for (int grp = 0; grp < 4; ++grp) {
      f32x16 acc2[MT][2];
#pragma unroll
      for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int ct = 0; ct < 2; ++ct)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc2[m][ct][r] = 0.0f;
      auto load_w3 = [&](int ks, u32x4 (&f)[2][NPL]) {
#pragma unroll
        for (int p = 0; p < NPL; ++p)
#pragma unroll
          for (int ct = 0; ct < 2; ++ct) f[ct][p] = Bf2[((((size_t)p * NB2 + 2 * grp + ct) * KS2 + wn * 8 + ks) * 2 + half) * 32 + lx];
      };
      auto step2 = [&](int n, int q, const u32x4 (&f)[2][NPL]) {      // k-step (tile n, register half q) of ffn.3 on every 32-row tile m
#pragma unroll
        for (int tm = 0; tm < S::NT; ++tm)
#pragma unroll
          for (int ct = 0; ct < 2; ++ct)
#pragma unroll
            for (int m = 0; m < MT; ++m) acc2[m][ct] = S::mma(hq[m][n][S::ta(tm)][q], f[ct][S::tb(tm)], acc2[m][ct]);
      };
      u32x4 fa2[2][NPL], fb2[2][NPL];
      load_w3(0, fa2);
#pragma unroll
      for (int n = 0; n < NT; ++n) {
        load_w3(2 * n + 1, fb2);
        __builtin_amdgcn_sched_barrier(0);
        step2(n, 0, fa2);
        __builtin_amdgcn_sched_barrier(0);
        if (n + 1 < NT) load_w3(2 * n + 2, fa2);
        __builtin_amdgcn_sched_barrier(0);
        step2(n, 1, fb2);
      }
      FT_TICK(6);
      // the 2 MT partial tiles of the round are finished in MT / 2 passes of four (one per wave): pass mp = rows 64 mp .. 64 mp + 63
#pragma unroll
      for (int mp = 0; mp < MT / 2; ++mp) {
      // the residual of the tile this wave will own (tile wn: rows 64 mp + 32 (wn >> 1) .., columns 64 grp + 32 (wn & 1) ..), requested
      // before the exchange so that its latency hides behind it
      const int ocol = 64 * grp + 32 * (wn & 1) + lx;
      const int orow_p = orow + 64 * mp;
      const unsigned obase_p = obase + (unsigned)(64 * mp) * (unsigned)a.ldc * 4u;
      const unsigned goff = (unsigned)__builtin_amdgcn_readfirstlane(grp * 256);
      float rv[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int k = (r & 3) + 8 * (r >> 2);
        rv[r] = orow_p + k < rows ? buf_load_f32_s(Rr, obase_p, goff + (unsigned)k * (unsigned)a.ldc * 4u) : 0.0f;
      }
      // ---- exchange: tile t = (m = 2 mp + (t >> 1), ct = t & 1) belongs to wave t; the other three waves hand over their partial sums ----
#pragma unroll
      for (int t4 = 0; t4 < 4; ++t4)
        if (t4 != wn) {
          const int src = wn < t4 ? wn : wn - 1;
#pragma unroll
          for (int rq = 0; rq < 4; ++rq)
            *(float4*)(xbuf + ((((t4 * 3 + src) * 4 + rq) * 64 + lane) << 2)) =
                make_float4(acc2[2 * mp + (t4 >> 1)][t4 & 1][4 * rq], acc2[2 * mp + (t4 >> 1)][t4 & 1][4 * rq + 1], acc2[2 * mp + (t4 >> 1)][t4 & 1][4 * rq + 2],
                            acc2[2 * mp + (t4 >> 1)][t4 & 1][4 * rq + 3]);
        }
      FT_TICK(7);
      __syncthreads();
      FT_TICK(8);
      float own[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float a0 = (wn & 1) ? acc2[2 * mp][1][r] : acc2[2 * mp][0][r], a1 = (wn & 1) ? acc2[2 * mp + 1][1][r] : acc2[2 * mp + 1][0][r];
        own[r] = (wn >> 1) ? a1 : a0;
      }
#pragma unroll
      for (int src = 0; src < 3; ++src) {
#pragma unroll
        for (int rq = 0; rq < 4; ++rq) {
          const float4 o = *(const float4*)(xbuf + ((((wn * 3 + src) * 4 + rq) * 64 + lane) << 2));
          own[4 * rq] += o.x; own[4 * rq + 1] += o.y; own[4 * rq + 2] += o.z; own[4 * rq + 3] += o.w;
        }
        __builtin_amdgcn_sched_barrier(0);   // one source (16 registers) in flight at a time: the hidden tile occupies half the file
      }
      const float iv3 = a.inv_ch2[ocol], bv3 = a.bias2[ocol];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float v = (own[r] * iv3 + bv3) + rv[r];
        vmax2 = fmaxf(vmax2, fabsf(v));
        const int k = (r & 3) + 8 * (r >> 2);
        if (orow_p + k < rows) buf_store_f32_s(Cr, obase_p, goff + (unsigned)k * (unsigned)a.ldc * 4u, v);
      }
      FT_TICK(9);
      __syncthreads();   // the exchange buffer is rewritten by the next pass / round
      }
      FT_TICK(10);
    }
    sat_report(a.sat2, vmax2);
    FT_FLUSH();
    return;
  }

  if constexpr (KV == 5) {
    // ================= SuperPoint's detector tail in the epilogue of convPb (SPN:176-179; round 6): the block's 128 cells x 65 logits never
    // leave the CU.  Every wave holds ALL channels of its 32 cells (lane = channel lx of column tile n: channels lx, 32 + lx; tile 2's lane 0 =
    // the dustbin), so the 65-way softmax is a max / sum over the 32 lanes of a lane half plus one broadcast value.  The arithmetic is
    // softmax_d2s_kernel's, operation for operation — logit = acc * inv + bias as the plain epilogue stores it; wave_sum's association
    // (channel c + channel c ^ 32 first, then the lane butterflies 16 .. 1); the dustbin's exponential added last — so the score map is
    // bit-identical to the two-kernel path (tests/test_superpoint_emu.py).  The probabilities go through LDS (the staging buffer is free) and
    // leave as 16-byte stores along the image rows: 128 consecutive cells = 1024 consecutive pixels of 8 rows.  No [cells][65] logits in HBM
    // (426 MB written and read per 100 maps), no N = 65 -> 128 padding (96 columns issued instead of 128). =================
    constexpr int HS = 72;                          // floats per cell row in LDS (64 + 8: the halves' rows land 32 banks apart)
    float* const tile = (float*)Ap;                 // [128 cells][HS]
    const float bv0 = a.bias[lx], bv1 = a.bias[32 + lx], bv2 = a.bias[64];
    const float iv0 = a.inv_ch[lx], iv1 = a.inv_ch[32 + lx], iv2 = a.inv_ch[64];
    float vmax = 0.0f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float v0 = acc[0][0][r] * iv0 + bv0, v1 = acc[0][1][r] * iv1 + bv1;
      const float dust = __shfl(acc[0][2][r] * iv2 + bv2, lane & 32);   // column 64 sits in lane 0 of each half
      vmax = sat_track(vmax, v0, v1);
      float m = fmaxf(v0, v1);
#pragma unroll
      for (int o = 16; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
      m = fmaxf(m, dust);
      const float e0 = expf(v0 - m), e1 = expf(v1 - m);
      float sum = e0 + e1;
#pragma unroll
      for (int o = 16; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
      sum += expf(dust - m);
      const int cell = wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
      tile[cell * HS + lx] = e0 / sum;
      tile[cell * HS + 32 + lx] = e1 / sum;
    }
    (void)vmax;   // (logits feed no split product: nothing to guard)
    __syncthreads();
    // depth-to-space: thread -> (cell, half of its 8-pixel rows); channel c = 8 dy + dx (SPN:178-179)
    const int cl = t >> 1, dxq = t & 1, cell = m0 + cl;
    if (cell < rows) {
      const int hw = a.d2s_h * a.d2s_w, b = cell / hw, rem = cell - b * hw, cy = rem / a.d2s_w, cx = rem - cy * a.d2s_w;
      const int W8 = a.d2s_w * 8;
      float* dst = a.d2s_out + ((size_t)b * a.d2s_h * 8 + (size_t)cy * 8) * W8 + cx * 8 + dxq * 4;
#pragma unroll
      for (int dy = 0; dy < 8; ++dy) *(float4*)(dst + (size_t)dy * W8) = *(const float4*)&tile[cl * HS + dy * 8 + dxq * 4];
    }
    return;
  }

  if (kblk || vblk) {
    // ---- K | V tile images (layout: lg_attn_x6.hip).  Rows past the ragged end are copies of the last valid row (finite;
    // the attention kernel masks their scores), tiles past the image capacity are skipped. ----
    u32x4* const img_item = (u32x4*)a.kv_img + (size_t)z * 4 * a.kv_tiles * KV_TILE_STRIDE;
    float vmax = 0.0f;
#pragma unroll
    for (int m = 0; m < MT; ++m) {
      const int tile = (m0 + wm * (32 * MT) + m * 32) >> 5;
      if (tile >= a.kv_tiles) continue;
#pragma unroll
      for (int n = 0; n < NT; ++n) {
        const int nk0 = n0 & 255;                          // (128-column blocks: which half of the 256-column K / V range)
        const int colw = nk0 + wn * (32 * NT) + n * 32;    // first column of the 32-column tile inside the 256-column range (4 heads x 2 tiles)
        const int head = colw >> 6, np = (colw >> 5) & 1;
        u32x4* const img = img_item + ((size_t)head * a.kv_tiles + tile) * KV_TILE_STRIDE;
        const int cbase = (n0 - nk0) + colw;
        if (kblk) {
          // lane = key lx (+ 32-key tile m), register r = output dim cbase + 8 (r >> 2) + 4 half + (r & 3)
          const int key = m0 + wm * (32 * MT) + m * 32 + lx;
          // rows past the ragged end are copies of the last valid row (load_chunk clamps): rotate them with THAT row's table
          // entry — the table is only initialised for live rows, and garbage here would trip the range guard
          const float* e = a.kv_enc ? a.kv_enc + ((size_t)z * a.kv_nmax + min(key, rows - 1)) * 64 : nullptr;
#pragma unroll
          for (int j = 0; j < 2; ++j) {   // registers 8j .. 8j+7 = one slot: dims 16 (2 np + j) + 4 half + {0..3, 8..11} of the head
            float v[8];
#pragma unroll
            for (int g = 0; g < 2; ++g) {
              const int c0 = cbase + 8 * (2 * j + g) + 4 * half;
              const float4 bv = *(const float4*)(a.bias + c0), iv = *(const float4*)(a.inv_ch + c0);
              v[4 * g + 0] = acc[m][n][8 * j + 4 * g + 0] * iv.x + bv.x;
              v[4 * g + 1] = acc[m][n][8 * j + 4 * g + 1] * iv.y + bv.y;
              v[4 * g + 2] = acc[m][n][8 * j + 4 * g + 2] * iv.z + bv.z;
              v[4 * g + 3] = acc[m][n][8 * j + 4 * g + 3] * iv.w + bv.w;
              if (e != nullptr) {  // rotary (LGN:41-54,155-156): pair (d, d+1) <-> frequency d / 2 of the head
                const int f0 = (np * 32 + 8 * (2 * j + g) + 4 * half) >> 1;
                const float2 cs = *(const float2*)(e + f0), sn = *(const float2*)(e + 32 + f0);
                const float t0 = v[4 * g], t1 = v[4 * g + 1], t2 = v[4 * g + 2], t3 = v[4 * g + 3];
                v[4 * g + 0] = t0 * cs.x + (-t1) * sn.x;
                v[4 * g + 1] = t1 * cs.x + t0 * sn.x;
                v[4 * g + 2] = t2 * cs.y + (-t3) * sn.y;
                v[4 * g + 3] = t3 * cs.y + t2 * sn.y;
              }
            }
            unsigned pc[4][NPL];
#pragma unroll
            for (int i = 0; i < 4; ++i) { S::split(v[2 * i], v[2 * i + 1], S::act_scale(), pc[i]); vmax = sat_track(vmax, v[2 * i], v[2 * i + 1]); }
            const int blk = 2 * (2 * np + j) + half;
#pragma unroll
            for (int pl = 0; pl < NPL; ++pl) img[(pl * 8 + blk) * 32 + lx] = u32x4{pc[0][pl], pc[1][pl], pc[2][pl], pc[3][pl]};
          }
        } else {
          // lane = dim lx of the tile, register r = key (r & 3) + 8 (r >> 2) + 4 half: registers 8u .. 8u+7 = slot (u, half, dim)
          const float bv = a.bias[cbase + lx], iv = a.inv_ch[cbase + lx];
#pragma unroll
          for (int u = 0; u < 2; ++u) {
            unsigned pc[4][NPL];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              const float v0 = acc[m][n][8 * u + 2 * i] * iv + bv, v1 = acc[m][n][8 * u + 2 * i + 1] * iv + bv;
              S::split(v0, v1, S::act_scale(), pc[i]);
              vmax = sat_track(vmax, v0, v1);
            }
#pragma unroll
            for (int pl = 0; pl < NPL; ++pl) img[NPL * 256 + ((pl * 2 + u) * 2 + half) * 64 + np * 32 + lx] = u32x4{pc[0][pl], pc[1][pl], pc[2][pl], pc[3][pl]};
          }
        }
      }
    }
    sat_report(a.sat, vmax);
    return;
  }

  if constexpr (lng) {
    // ---- LayerNorm(512) + GELU over the block's full rows (LGN:141-142), then the store.
    // Row statistics: in-lane over the wave's 4 column tiles, DPP over the 16-lane rows, then through LDS (the activation
    // staging buffer is free after the K loop) over the 4 x 4 sixteen-lane groups that share a block row; mean first, then
    // the centred sum of squares — the two-pass form lg_ln_gelu_kernel (and ATen) use. ----
    float* const red = (float*)Ap;                 // [wave 4][lane group 4][32 (m, r)] partial sums
    float bv[NT], iv[NT], gm[NT], bt[NT];
#pragma unroll
    for (int n = 0; n < NT; ++n) {
      const int col = n0 + wn * (32 * NT) + n * 32 + lx;
      bv[n] = a.bias[col]; iv[n] = a.inv_ch[col]; gm[n] = a.ln_gamma[col]; bt[n] = a.ln_beta[col];
    }
    const int q = lane >> 4;
    float rowv[MT * 16];
    auto block_rows = [&](float (&part)[MT * 16], float scale_) {   // part[(m, r)] per lane -> total over the block row, x scale_
#pragma unroll
      for (int j = 0; j < MT * 16; ++j) {
        float v = part[j];
        v += dpp_f(v, 0xB1); v += dpp_f(v, 0x4E); v += dpp_f(v, 0x141); v += dpp_f(v, 0x140);
        if ((lane & 15) == 0) red[(wv * 4 + q) * 32 + j] = v;
      }
      __syncthreads();
      // One row total per lane: lane (lx, half) sums the 8 partials (4 waves x the 2 sixteen-lane groups of its lane half) of row
      // slot j = lx in a fixed order; the 32 totals of a half then reach all of its lanes through v_readlane (wave-uniform values,
      // no second trip through LDS).  (A first version let wave 0 alone form the 64 row totals and publish them through LDS for a
      // second barrier-separated read: correct on the emulator and at one workgroup per CU, but on hardware with two co-resident
      // workgroups ~1 row in 1000 came back with stale statistics — found by the fp64 op test at 204 800 rows,
      // scripts/gpu_ffn_ln_check.py.  Letting every lane read all 256 partials itself was correct but cost 30 % of the kernel.)
      float tot = 0.0f;
#pragma unroll
      for (int w4 = 0; w4 < 4; ++w4) tot += red[(w4 * 4 + 2 * half) * 32 + lx] + red[(w4 * 4 + 2 * half + 1) * 32 + lx];
      tot *= scale_;
#pragma unroll
      for (int j = 0; j < MT * 16; ++j) {
        const float a0 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(tot), j));
        const float a1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(tot), 32 + j));
        part[j] = half ? a1 : a0;
      }
      __syncthreads();
    };
    // v = acc * inv + bias once, in place (inv is a power of two: the fused multiply-add rounds like the separate product and
    // sum); from here on the arithmetic runs on column pairs (n, n + 1) as packed fp32 operations — the epilogue is ~ 5000 VALU
    // instructions per thread against 768 MFMAs, and is what the kernel ran at before the polynomial chains were packed
    static_assert(NT % 2 == 0, "column pairs");
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int n = 0; n < NT; n += 2)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const f32x2 v = __builtin_elementwise_fma(f32x2{acc[m][n][r], acc[m][n + 1][r]}, f32x2{iv[n], iv[n + 1]}, f32x2{bv[n], bv[n + 1]});
          acc[m][n][r] = v[0]; acc[m][n + 1][r] = v[1];
        }
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        f32x2 sm_ = {0.0f, 0.0f};
#pragma unroll
        for (int n = 0; n < NT; n += 2) sm_ += f32x2{acc[m][n][r], acc[m][n + 1][r]};
        rowv[m * 16 + r] = sm_[0] + sm_[1];
      }
    block_rows(rowv, 1.0f / 512.0f);               // rowv = mean of the row
    float sq[MT * 16];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const f32x2 mean2 = {rowv[m * 16 + r], rowv[m * 16 + r]};
        f32x2 s2 = {0.0f, 0.0f};
#pragma unroll
        for (int n = 0; n < NT; n += 2) { const f32x2 d = f32x2{acc[m][n][r], acc[m][n + 1][r]} - mean2; s2 = __builtin_elementwise_fma(d, d, s2); }
        sq[m * 16 + r] = s2[0] + s2[1];
      }
    block_rows(sq, 1.0f / 512.0f);                 // sq = biased variance of the row
    const dim_rsrc Cr = buf_rsrc(a.C + (size_t)z * a.strideC, ((size_t)(rows - 1) * a.ldc + a.N) * sizeof(float));
    const unsigned ldc4 = (unsigned)a.ldc * 4u;
    float vmax = 0.0f;
#pragma unroll
    for (int m = 0; m < MT; ++m) {
      const unsigned row0 = (unsigned)(m0 + m * 32 + 4 * half);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float mean = rowv[m * 16 + r], rstd = 1.0f / sqrtf(sq[m * 16 + r] + 1e-5f);
        const f32x2 mean2 = {mean, mean}, rstd2 = {rstd, rstd};
        const unsigned rb = (row0 + (unsigned)((r & 3) + 8 * (r >> 2))) * ldc4;
#pragma unroll
        for (int n = 0; n < NT; n += 2) {
          const f32x2 d = (f32x2{acc[m][n][r], acc[m][n + 1][r]} - mean2) * rstd2;
          const f32x2 y = __builtin_elementwise_fma(d, f32x2{gm[n], gm[n + 1]}, f32x2{bt[n], bt[n + 1]});
          const f32x2 e = erf2_1ulp(y * f32x2{0.70710678118654752440f, 0.70710678118654752440f});
          const f32x2 hy = y * f32x2{0.5f, 0.5f};
          const f32x2 g = __builtin_elementwise_fma(hy, e, hy);     // 0.5 y (1 + erf(y / sqrt 2))
          vmax = sat_track(vmax, g[0], g[1]);
          buf_store_f32(Cr, rb + (unsigned)(n0 + wn * (32 * NT) + n * 32 + lx) * 4u, g[0]);
          buf_store_f32(Cr, rb + (unsigned)(n0 + wn * (32 * NT) + (n + 1) * 32 + lx) * 4u, g[1]);
        }
      }
    }
    sat_report(a.sat, vmax);
    return;
  }

  // Epilogue through buffer descriptors (dim_common.h): 32-bit offsets, no per-element branches.  The descriptor of C / R
  // ends after the item's last valid row, so rows past the ragged end are dropped (stores) / read as zero (loads) by the
  // hardware; a column past N selects the out-of-range offset.  Per (m, n) tile all 16 residual values are requested
  // before any is used; the uniform decisions (residual? activation?) are hoisted.
  const dim_rsrc Cr = buf_rsrc(a.C + (size_t)z * a.strideC, ((size_t)(rows - 1) * a.ldc + a.N) * sizeof(float));
  const dim_rsrc Rr = buf_rsrc(a.R ? a.R + (size_t)z * a.strideR : a.C, a.R ? ((size_t)(rows - 1) * a.ldr + a.N) * sizeof(float) : 0);
  const bool has_r = a.R != nullptr && !(PROBE & 16);   // probe bit 4: no residual loads, one store per thread
  const unsigned ldc4 = (unsigned)a.ldc * 4u, ldr4 = (unsigned)a.ldr * 4u;
  float vmax = 0.0f;  // fp16x3 range guard on what this thread stores (dim_common.h)
#pragma unroll
  for (int n = 0; n < NT; ++n) {
    const int col = n0 + wn * (32 * NT) + n * 32 + lx;
    const bool colok = col < a.N;
    const int colc = colok ? col : a.N - 1;
    const float bv = a.bias ? a.bias[colc] : 0.0f;
    const float inv = a.inv_ch[colc];  // per-column inverse weight scale (x activation scale)
#pragma unroll
    for (int m = 0; m < MT; ++m) {
      const unsigned row0 = (unsigned)(m0 + wm * (32 * MT) + m * 32 + 4 * half);  // row of register r: row0 + (r & 3) + 8 * (r >> 2)
      const unsigned cbase = colok ? row0 * ldc4 + (unsigned)col * 4u : DIM_BUF_OOB;
      const unsigned rbase = colok ? row0 * ldr4 + (unsigned)col * 4u : DIM_BUF_OOB;
      float v[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) v[r] = acc[m][n][r] * inv + bv;
      if (has_r) {
        float rv[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) rv[r] = buf_load_f32(Rr, rbase + (unsigned)((r & 3) + 8 * (r >> 2)) * ldr4);
#pragma unroll
        for (int r = 0; r < 16; ++r) v[r] += rv[r];
      }
      if (a.relu == 1) {
#pragma unroll
        for (int r = 0; r < 16; ++r) v[r] = fmaxf(v[r], 0.0f);
      } else if (a.relu == 2) {
#pragma unroll
        for (int r = 0; r < 16; ++r) v[r] = v[r] <= 0.0f ? (exp_le0(v[r]) - 1.0f) * 1.7580993408473768599402175208123f : v[r] * 1.0507009873554804934193349852946f;
      }
      // range guard over all 16 values, unconditionally: rows past the ragged end are duplicates of the last valid row
      // (load_chunk clamps) plus a zero residual, padded columns hold the bias of the last valid column
#pragma unroll
      for (int r = 0; r < 16; r += 2) vmax = sat_track(vmax, v[r], v[r + 1]);
      if (PROBE & 16) {
        float sum = 0.0f;
#pragma unroll
        for (int r = 0; r < 16; ++r) sum += v[r];
        if (m == 0 && n == 0) buf_store_f32(Cr, cbase, sum); else vmax = fmaxf(vmax, sum);
        continue;
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) buf_store_f32(Cr, cbase + (unsigned)((r & 3) + 8 * (r >> 2)) * ldc4, v[r]);
    }
  }
  if (MODE == 2) sat_report(a.sat, vmax);
}

// layer_tab (dim_kernels.h): the item's flag selects the layer whose weights it multiplies by; false = the item is not live
__device__ __forceinline__ bool gemm_pick_layer(GemmArgs& a) {
  if (a.layer_tab == nullptr) return true;
  const int fl = a.flag[blockIdx.z >> a.flag_shift];
  if (fl <= 0) return false;
  const GemmLayerTab lt = a.layer_tab[fl - 1];
  a.Bx3 = lt.Bx3; a.inv_ch = lt.inv_ch; a.bias = lt.bias; a.flag = nullptr;
  return true;
}
template <int MODE, int BM, int NT = 2, int WN = 2>
__global__ __launch_bounds__(256, ((BM / (32 * (4 / WN))) * NT >= 16 ? 1 : ((BM / (32 * (4 / WN))) * NT >= 8 ? 2 : 3))) void gemm_x6_kernel(GemmArgs a) {
  __shared__ unsigned Ap[SplitMma<MODE>::NPL * BM * RS];
  if (!gemm_pick_layer(a)) return;
  gemm_x6_body<MODE, BM, NT, 0, WN, 0, (WN == 4)>(a, Ap, (int)blockIdx.y);   // the wide block runs the pipelined K loop
}
#ifdef DIM_RESEARCH
// small problems, fp16x3: the streaming K loop (see STREAM above; prototype, dim_tune_set key 14 = 63); no LDS
template <int KS_T>
__global__ __launch_bounds__(256, 2) void gemm_x6_stream_kernel(GemmArgs a) {
  gemm_x6_body<2, 64, 2, 0, 2, 0, false, 32, false, false, true, KS_T>(a, nullptr, (int)blockIdx.y);
}
#endif
// one LightGlue pair per call (32 x 128 blocks, see launch_gemm_x6): the q|k|v projection writing the attention kernel's K | V tile images from its epilogue (KV above; two 128-column blocks per 256-column K / V range)
constexpr int RS64 = 64 / 2 + 4;   // row stride in dwords of a staged 64-wide chunk (conflict-free like RS)
template <int KCH, bool BSET, int K_T = 0>
__global__ __launch_bounds__(256, 3) void gemm_x6_qkv_small_kernel(GemmArgs a) {
  __shared__ unsigned Ap[2 * 32 * (KCH / 2 + 4)];
  const int by = (int)blockIdx.y, b256 = by >> 1;
  if (b256 == a.kv_kblock) gemm_x6_body<2, 32, 1, 1, 4, 0, true, KCH, false, false, false, 0, BSET, K_T>(a, Ap, by);
  else if (b256 == a.kv_vblock) gemm_x6_body<2, 32, 1, 2, 4, 0, true, KCH, false, false, false, 0, BSET, K_T>(a, Ap, by);
  else gemm_x6_body<2, 32, 1, 0, 4, 0, true, KCH, false, false, false, 0, BSET, K_T>(a, Ap, by);
}
// the plain 32 x 128 block with 64-wide K chunks: half the barrier pairs and staging round trips per MFMA of the 32-wide form (gemm_x6_kernel<2, 32, 1, 4>,
// used when K is not a multiple of 64); bit-identical
template <bool BSET, int K_T = 0>
__global__ __launch_bounds__(256, 3) void gemm_x6_small32_kc64_kernel(GemmArgs a) {
  __shared__ unsigned Ap[2 * 32 * RS64];
  if (!gemm_pick_layer(a)) return;
  gemm_x6_body<2, 32, 1, 0, 4, 0, true, 64, false, false, false, 0, BSET, K_T>(a, Ap, (int)blockIdx.y);
}
// LightGlue's ffn.0 with LayerNorm + GELU in the epilogue: one workgroup owns 64 rows x all 512 columns
__global__ __launch_bounds__(256, 2) void gemm_x6_ffn_ln_kernel(GemmArgs a) {
  __shared__ unsigned Ap[2 * 64 * RS];
  gemm_x6_body<2, 64, 4, 3, 4, 0, true>(a, Ap, 0);
}
// SuperPoint's convPb (256 -> 65) + 65-way softmax + depth-to-space: one workgroup owns 128 cells x all 65 channels (KV = 5 above)
#ifndef DIM_HEAD_WGS
#define DIM_HEAD_WGS 3
#endif
#ifndef DIM_HEAD_PIPE
#define DIM_HEAD_PIPE true
#endif
__global__ __launch_bounds__(256, DIM_HEAD_WGS) void gemm_x6_head_kernel(GemmArgs a) {
  __shared__ unsigned Ap[128 * 72];
  gemm_x6_body<2, 128, 3, 5, 1, 0, DIM_HEAD_PIPE>(a, Ap, 0);
}
#ifdef DIM_RESEARCH   // timing probes (wrong results by design): research build only
template <int PROBE, bool PIPE = false>
__global__ __launch_bounds__(256, 2) void gemm_x6_probe_kernel(GemmArgs a) {
  __shared__ unsigned Ap[2 * 128 * RS];
  gemm_x6_body<2, 128, 2, 0, 4, PROBE, PIPE>(a, Ap, (int)blockIdx.y);
}
template <int PROBE, bool PIPE = false>
__global__ __launch_bounds__(256, 2) void gemm_x6_ffn_ln_probe_kernel(GemmArgs a) {
  __shared__ unsigned Ap[2 * 64 * RS];
  gemm_x6_body<2, 64, 4, 3, 4, PROBE, PIPE>(a, Ap, 0);
}
#endif
// C[z] = A[z] * B[z]^T with BOTH operands fp32 activations (LightGlue's similarity sim = mdesc0 mdesc1^T, LGN:271; K = 256): both
// 128 x 32 chunks are split while they are staged into LDS and both MFMA operands come from there.  128 x 128 block, waves 2 x 2,
// each 64 x 64.  Ragged rows (a.rows) and columns (a.cols) like gemm.hip's bt mode; entries outside stay untouched.  The fp32 MFMA
// GEMM this replaces ran the 107 GFLOP of a 50-pair batch at 86 TFLOP/s (1.25 ms).
template <int MODE>
__global__ __launch_bounds__(256, (MODE == 2 ? 3 : 2)) void gemm_x6_nt_kernel(GemmArgs a) {
  using S = SplitMma<MODE>;
  constexpr int NPL = S::NPL, BM = 128;
  __shared__ unsigned Ap[NPL * BM * RS], Bp[NPL * BM * RS];
  const int z = blockIdx.z;
  if (a.flag) {
    const int fl = a.flag[z >> a.flag_shift];
    if (a.flag_any ? fl <= 0 : fl != a.flag_eq) return;
  }
  const int rows = a.rows ? a.rows[z * a.rows_mul + a.rows_off] * a.rows_scale : a.M;
  const int cols = a.cols ? a.cols[z * a.cols_mul + a.cols_off] : a.N;
  const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BM;
  if (m0 >= rows || n0 >= cols) return;
  const int t = threadIdx.x, lane = t & 63, wv = t >> 6, wm = wv >> 1, wn = wv & 1, lx = lane & 31, half = lane >> 5;
  const float* A = a.A0 + (size_t)z * a.strideA0;
  const float* B = a.B + (size_t)z * a.strideB;
  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;
  float4 ra[4], rb[4];
  auto load_chunk = [&](int k0) {   // rows / columns past the ragged end re-read the last valid one (never stored)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int idx = t + 256 * i, row = idx >> 3, q = idx & 7;
      ra[i] = *(const float4*)(A + (size_t)min(m0 + row, rows - 1) * a.lda0 + k0 + q * 4);
      rb[i] = *(const float4*)(B + (size_t)min(n0 + row, cols - 1) * a.ldb + k0 + q * 4);
    }
  };
  auto store_chunk = [&]() {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int idx = t + 256 * i, row = idx >> 3, q = idx & 7;
      unsigned p0[NPL], p1[NPL], q0[NPL], q1[NPL];
      S::split(ra[i].x, ra[i].y, S::act_scale(), p0); S::split(ra[i].z, ra[i].w, S::act_scale(), p1);
      S::split(rb[i].x, rb[i].y, S::act_scale(), q0); S::split(rb[i].z, rb[i].w, S::act_scale(), q1);
#pragma unroll
      for (int pl = 0; pl < NPL; ++pl) {
        Ap[(pl * BM + row) * RS + q * 2] = p0[pl]; Ap[(pl * BM + row) * RS + q * 2 + 1] = p1[pl];
        Bp[(pl * BM + row) * RS + q * 2] = q0[pl]; Bp[(pl * BM + row) * RS + q * 2 + 1] = q1[pl];
      }
    }
  };
  load_chunk(0);
  for (int k0 = 0; k0 < a.K; k0 += KC) {
    store_chunk();
    __syncthreads();
    load_chunk(min(k0 + KC, a.K - KC));
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      u32x4 fa[2][NPL], fb[2][NPL];
#pragma unroll
      for (int p = 0; p < NPL; ++p)
#pragma unroll
        for (int m = 0; m < 2; ++m) {
          fa[m][p] = *(const u32x4*)&Ap[(p * BM + wm * 64 + m * 32 + lx) * RS + ks * 8 + half * 4];
          fb[m][p] = *(const u32x4*)&Bp[(p * BM + wn * 64 + m * 32 + lx) * RS + ks * 8 + half * 4];
        }
#pragma unroll
      for (int tm = 0; tm < S::NT; ++tm)
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
          for (int n = 0; n < 2; ++n) acc[m][n] = S::mma(fa[m][S::ta(tm)], fb[n][S::tb(tm)], acc[m][n]);
    }
    __syncthreads();
  }
  const float inv = 1.0f / (S::act_scale() * S::act_scale());
  const dim_rsrc Cr = buf_rsrc(a.C + (size_t)z * a.strideC, ((size_t)(rows - 1) * a.ldc + cols) * sizeof(float));
  const unsigned ldc4 = (unsigned)a.ldc * 4u;
#pragma unroll
  for (int n = 0; n < 2; ++n) {
    const int col = n0 + wn * 64 + n * 32 + lx;
#pragma unroll
    for (int m = 0; m < 2; ++m) {
      const unsigned row0 = (unsigned)(m0 + wm * 64 + m * 32 + 4 * half);
      const unsigned cbase = col < cols ? row0 * ldc4 + (unsigned)col * 4u : DIM_BUF_OOB;
#pragma unroll
      for (int r = 0; r < 16; ++r) buf_store_f32(Cr, cbase + (unsigned)((r & 3) + 8 * (r >> 2)) * ldc4, acc[m][n][r] * inv);
    }
  }
}
#ifdef DIM_RESEARCH   // prototypes that lost their A/B: research build only
// ---- round 5: 256-row blocks at the 512-register point (dim_tune_set key 14 = 256): every wave owns 256 rows x 64 columns = 16 accumulators
// (AGPRs), one workgroup per CU, one wave per SIMD.  A weight fragment fetched from L2 then serves twice the rows: the wide blocks stream
// 4.4 TB/s of weight fragments through an L2 that delivers ~8 — the kernels are as much L2- as matrix-bound ----
__global__ __launch_bounds__(256, 1) void gemm_x6_qkv256_kernel(GemmArgs a) {
  __shared__ unsigned Ap[2 * 256 * RS];
  const int by = (int)blockIdx.y;
  if (by == a.kv_kblock) gemm_x6_body<2, 256, 2, 1, 4, 0, true>(a, Ap, by);
  else if (by == a.kv_vblock) gemm_x6_body<2, 256, 2, 2, 4, 0, true>(a, Ap, by);
  else gemm_x6_body<2, 256, 2, 0, 4, 0, true>(a, Ap, by);
}
// ---- prototypes with 64-wide K chunks (dim_tune_set key 14 = 64; measured in round 4): the plain 128 x 256 block and the q|k|v kernel ----
__global__ __launch_bounds__(256, 2) void gemm_x6_wide_kc64_kernel(GemmArgs a) {
  __shared__ unsigned Ap[2 * 128 * RS64];
  gemm_x6_body<2, 128, 2, 0, 4, 0, true, 64>(a, Ap, (int)blockIdx.y);
}
__global__ __launch_bounds__(256, 2) void gemm_x6_qkv_kc64_kernel(GemmArgs a) {
  __shared__ unsigned Ap[2 * 128 * RS64];
  const int by = (int)blockIdx.y;
  if (by == a.kv_kblock) gemm_x6_body<2, 128, 2, 1, 4, 0, true, 64>(a, Ap, by);
  else if (by == a.kv_vblock) gemm_x6_body<2, 128, 2, 2, 4, 0, true, 64>(a, Ap, by);
  else gemm_x6_body<2, 128, 2, 0, 4, 0, true, 64>(a, Ap, by);
}
// ---- prototypes with the activation tile double-buffered in LDS (dim_tune_set key 14 = 33) ----
__global__ __launch_bounds__(256, 2) void gemm_x6_wide_db_kernel(GemmArgs a) {
  __shared__ unsigned Ap[2 * 2 * 128 * RS];
  gemm_x6_body<2, 128, 2, 0, 4, 0, true, 32, true>(a, Ap, (int)blockIdx.y);
}
__global__ __launch_bounds__(256, 2) void gemm_x6_qkv_db_kernel(GemmArgs a) {
  __shared__ unsigned Ap[2 * 2 * 128 * RS];
  const int by = (int)blockIdx.y;
  if (by == a.kv_kblock) gemm_x6_body<2, 128, 2, 1, 4, 0, true, 32, true>(a, Ap, by);
  else if (by == a.kv_vblock) gemm_x6_body<2, 128, 2, 2, 4, 0, true, 32, true>(a, Ap, by);
  else gemm_x6_body<2, 128, 2, 0, 4, 0, true, 32, true>(a, Ap, by);
}
// ---- prototypes with tile-by-tile refilled fragment sets (dim_tune_set key 14 = 37) ----
__global__ __launch_bounds__(256, 2) void gemm_x6_wide_roll_kernel(GemmArgs a) {
  __shared__ unsigned Ap[2 * 128 * RS];
  gemm_x6_body<2, 128, 2, 0, 4, 0, true, 32, false, true>(a, Ap, (int)blockIdx.y);
}
__global__ __launch_bounds__(256, 2) void gemm_x6_qkv_roll_kernel(GemmArgs a) {
  __shared__ unsigned Ap[2 * 128 * RS];
  const int by = (int)blockIdx.y;
  if (by == a.kv_kblock) gemm_x6_body<2, 128, 2, 1, 4, 0, true, 32, false, true>(a, Ap, by);
  else if (by == a.kv_vblock) gemm_x6_body<2, 128, 2, 2, 4, 0, true, 32, false, true>(a, Ap, by);
  else gemm_x6_body<2, 128, 2, 0, 4, 0, true, 32, false, true>(a, Ap, by);
}
// ---- round 6 probe of the one-pair 32 x 128 block (dim_tune_set key 14 = 76): the double-buffered activation tile ----
__global__ __launch_bounds__(256, 3) void gemm_x6_small32_db_kernel(GemmArgs a) {
  __shared__ unsigned Ap[2 * 2 * 32 * RS];
  gemm_x6_body<2, 32, 1, 0, 4, 0, true, 32, true>(a, Ap, (int)blockIdx.y);
}
#endif   // DIM_RESEARCH
// LightGlue's ffn.0 -> LayerNorm -> GELU -> ffn.3 (+ residual) in one kernel: 64 rows per workgroup, the hidden tensor stays on the CU
constexpr int ffn_lds_dwords(int bm) { return 2 * bm * RS + 2048 + 8 * bm + 4 * 3 * 4 * 64 * 4; }
constexpr int FFN_LDS_DWORDS = ffn_lds_dwords(64);
__global__ __launch_bounds__(256, 2) void gemm_x6_ffn_fused_kernel(GemmArgs a) {
  __shared__ unsigned Ap[FFN_LDS_DWORDS];
  float* const prm = (float*)(Ap + 2 * 64 * RS);
  for (int i = threadIdx.x; i < 512; i += 256) {
    prm[i] = a.inv_ch[i]; prm[512 + i] = a.bias[i]; prm[1024 + i] = a.ln_gamma[i]; prm[1536 + i] = a.ln_beta[i];
  }   // (visible to every wave after the K loop's barriers)
  gemm_x6_body<2, 64, 4, 4, 4, 0, false, 32, false, true>(a, Ap, 0);   // rolling fragment requests (round 4: 593 -> 576 us; the k-step-pipelined
}                                                                       // loop needs 32 more registers here: measured slower, 613 vs 592 us)
#ifdef DIM_RESEARCH
// round 5 (dim_tune_set key 14 = 128): the fused feed-forward on 128-row blocks at the 512-register point — 16 ffn.0 accumulators per wave in
// AGPRs, one workgroup per CU.  A 64-row block streams 1.5 MB of weight fragments from L2: 3200 workgroups x 1.5 MB in 558 us = 8.6 TB/s, the
// L2's limit; a 128-row block halves that.
__global__ __launch_bounds__(256, 1) void gemm_x6_ffn_fused128_kernel(GemmArgs a) {
  __shared__ unsigned Ap[ffn_lds_dwords(128)];
  float* const prm = (float*)(Ap + 2 * 128 * RS);
  for (int i = threadIdx.x; i < 512; i += 256) {
    prm[i] = a.inv_ch[i]; prm[512 + i] = a.bias[i]; prm[1024 + i] = a.ln_gamma[i]; prm[1536 + i] = a.ln_beta[i];
  }
  gemm_x6_body<2, 128, 4, 4, 4, 0, false, 32, false, true>(a, Ap, 0);
}
__global__ __launch_bounds__(256, 2) void gemm_x6_ffn_fused_step_kernel(GemmArgs a) {   // round 3's loop (one k-step's fragments at a time), kept for A/B: dim_tune_set(14, 36)
  __shared__ unsigned Ap[FFN_LDS_DWORDS];
  float* const prm = (float*)(Ap + 2 * 64 * RS);
  for (int i = threadIdx.x; i < 512; i += 256) {
    prm[i] = a.inv_ch[i]; prm[512 + i] = a.bias[i]; prm[1024 + i] = a.ln_gamma[i]; prm[1536 + i] = a.ln_beta[i];
  }
  gemm_x6_body<2, 64, 4, 4, 4, 0, false>(a, Ap, 0);
}
__global__ __launch_bounds__(256, 2) void gemm_x6_ffn_fused_roll_probe_kernel(GemmArgs a) {   // timing probe (14 = 35): activations from 2048 cached rows, results wrong
  __shared__ unsigned Ap[FFN_LDS_DWORDS];
  float* const prm = (float*)(Ap + 2 * 64 * RS);
  for (int i = threadIdx.x; i < 512; i += 256) {
    prm[i] = a.inv_ch[i]; prm[512 + i] = a.bias[i]; prm[1024 + i] = a.ln_gamma[i]; prm[1536 + i] = a.ln_beta[i];
  }
  gemm_x6_body<2, 64, 4, 4, 4, 1, false, 32, false, true>(a, Ap, 0);
}
#endif   // DIM_RESEARCH
// LightGlue's q|k|v projection in ONE launch: blockIdx.y selects the column block and with it the code path (plain fp32 /
// transposed K image / V image — three inlined bodies, one register allocation each), so that the 2 or 3 column blocks of
// a row block run next to each other on the same XCD and the activation rows come from HBM once (as separate launches
// the three blocks each re-read them: 116 + 162 + 121 us where the traffic of one pass allows ~ 170).
__global__ __launch_bounds__(256, 2) void gemm_x6_qkv_kernel(GemmArgs a) {
  __shared__ unsigned Ap[2 * 128 * RS];
  const int by = (int)blockIdx.y;
  if (by == a.kv_kblock) gemm_x6_body<2, 128, 2, 1, 4, 0, true>(a, Ap, by);
  else if (by == a.kv_vblock) gemm_x6_body<2, 128, 2, 2, 4, 0, true>(a, Ap, by);
  else gemm_x6_body<2, 128, 2, 0, 4, 0, true>(a, Ap, by);
}
}  // namespace

static int g_gemm_x6_wide = 1;
int dim_gemm_x6_wide() { return g_gemm_x6_wide; }
void dim_gemm_x6_set_wide(int v) { g_gemm_x6_wide = v; }

static bool small_problem(int M, int N, int batch) {  // fewer 128-row workgroups than CUs (never when the wide block is forced: tests)
  return dim_gemm_x6_wide() != 2 && (long)cdiv(M, 128) * cdiv(N, 128) * batch < 256;
}
static bool wide_block(int M, int n_pad, int batch, int split_mode) {
  // dim_gemm_x6_wide(): 2 = forced (tests); 1 = when the launch still fills 2 workgroups per CU
  return split_mode == 2 && dim_gemm_x6_wide() && n_pad % 256 == 0 && (dim_gemm_x6_wide() == 2 || (long)cdiv(M, 128) * (n_pad / 256) * batch >= 512);
}
// launches of at most two 64-row workgroups per CU (one LightGlue pair per call): the 32 x 128 block (see launch_gemm_x6)
static bool small32(int M, int N, int batch, int split_mode) {
  return split_mode == 2 && small_problem(M, N, batch) && (long)cdiv(M, 64) * cdiv(N, 128) * batch <= 512;
}
bool gemm_x6_fuses_kv(int M, int n_pad, int batch, int split_mode) {
  return (!small_problem(M, n_pad, batch) && wide_block(M, n_pad, batch, split_mode)) || (n_pad % 256 == 0 && small32(M, n_pad, batch, split_mode));
}

int launch_gemm_x6_nt(const GemmArgs& a, int batch, int split_mode, hipStream_t s) {
  DIM_REQUIRE(a.bt && a.A0 && a.B && a.C && a.A1 == nullptr && a.bias == nullptr && a.R == nullptr && a.relu == 0, "gemm_x6_nt: plain A * B^T only");
  DIM_REQUIRE(a.K % KC == 0 && a.lda0 % 4 == 0 && a.ldb % 4 == 0 && (split_mode == 1 || split_mode == 2), "gemm_x6_nt: K %% 32, leading dimensions %% 4");
  if (batch <= 0 || a.M <= 0 || a.N <= 0) return 0;
  dim3 grid(cdiv(a.M, 128), cdiv(a.N, 128), batch);
  if (split_mode == 2) hipLaunchKernelGGL(HIP_KERNEL_NAME(gemm_x6_nt_kernel<2>), grid, dim3(256), 0, s, a);
  else hipLaunchKernelGGL(HIP_KERNEL_NAME(gemm_x6_nt_kernel<1>), grid, dim3(256), 0, s, a);
  DIM_LAUNCH_CHECK();
  return 0;
}

int launch_gemm_x6(const GemmArgs& a, int batch, hipStream_t s) {
  DIM_REQUIRE(a.Bx3 != nullptr && !a.bt && (a.split_mode == 1 || a.split_mode == 2), "gemm_x6: needs pre-split [planes][n_pad][K] weights");
  DIM_REQUIRE(a.K % 16 == 0, "gemm_x6: K");
  DIM_REQUIRE(a.K % KC == 0 && (a.A1 == nullptr || a.ksplit % KC == 0), "gemm_x6: K=%d / ksplit=%d must be multiples of %d", a.K, a.ksplit, KC);
  constexpr int BN = 128;
  DIM_REQUIRE(a.n_pad % BN == 0 && a.n_pad >= a.N, "gemm_x6: n_pad=%d must be a multiple of %d covering N=%d", a.n_pad, BN, a.N);
  DIM_REQUIRE(a.lda0 % 4 == 0 && (a.A1 == nullptr || a.lda1 % 4 == 0), "gemm_x6: leading dims must be multiples of 4");
  if (batch <= 0 || a.M <= 0 || a.N <= 0) return 0;
  // research selectors (prototype blocks, timing probes): never for a launch with per-item layer weights — only the product blocks pick a layer
  const int kc_sel = a.layer_tab ? 32 : dim_gemm_kc(), probe_sel = a.layer_tab ? 0 : dim_gemm_probe();
  (void)kc_sel; (void)probe_sel;
  DIM_REQUIRE(a.layer_tab == nullptr || (a.d2s_out == nullptr && a.ln_gamma == nullptr && a.B2x3 == nullptr), "gemm_x6: per-item layer weights exist for the plain blocks");
  if (a.d2s_out != nullptr) {
    DIM_REQUIRE(a.split_mode == 2 && a.N == 65 && a.n_pad >= 96 && a.bias && a.R == nullptr && a.relu == 0 && a.kv_img == nullptr && a.ln_gamma == nullptr && a.A1 == nullptr &&
                batch == 1 && a.d2s_h > 0 && a.d2s_w > 0 && a.M % (a.d2s_h * a.d2s_w) == 0 && (a.d2s_w * 8) % 4 == 0,
                "gemm_x6: the detector-head epilogue needs the fp16x3 256 -> 65 shape over [maps][h][w] cells");
    hipLaunchKernelGGL(gemm_x6_head_kernel, dim3(cdiv(a.M, 128), 1, 1), dim3(256), 0, s, a);
    DIM_LAUNCH_CHECK();
    return 0;
  }
  if (a.ln_gamma != nullptr && a.B2x3 != nullptr) {
    DIM_REQUIRE(a.split_mode == 2 && a.N == 512 && a.n_pad == 512 && a.K % KC == 0 && a.ln_beta && a.bias && a.bias2 && a.inv_ch2 && a.R && a.C && a.relu == 0 &&
                a.kv_img == nullptr && a.ldr == a.ldc && a.strideR == a.strideC,
                "gemm_x6: the fused feed-forward needs the fp16x3 512 -> 256 shapes, a residual laid out like the output and both bias vectors");
#ifdef DIM_RESEARCH
    if (kc_sel == 128) hipLaunchKernelGGL(gemm_x6_ffn_fused128_kernel, dim3(cdiv(a.M, 128), 1, batch), dim3(256), 0, s, a);
    else if (kc_sel == 35) hipLaunchKernelGGL(gemm_x6_ffn_fused_roll_probe_kernel, dim3(cdiv(a.M, 64), 1, batch), dim3(256), 0, s, a);
    else if (kc_sel == 36) hipLaunchKernelGGL(gemm_x6_ffn_fused_step_kernel, dim3(cdiv(a.M, 64), 1, batch), dim3(256), 0, s, a);
    else
#endif
    hipLaunchKernelGGL(gemm_x6_ffn_fused_kernel, dim3(cdiv(a.M, 64), 1, batch), dim3(256), 0, s, a);
    DIM_LAUNCH_CHECK();
    return 0;
  }
  if (a.ln_gamma != nullptr) {
    DIM_REQUIRE(a.split_mode == 2 && a.N == 512 && a.n_pad == 512 && a.ln_beta && a.bias && a.R == nullptr && a.relu == 0 && a.kv_img == nullptr,
                "gemm_x6: the LayerNorm + GELU epilogue needs the fp16x3 512-column ffn.0 shape");
    const dim3 lg(cdiv(a.M, 64), 1, batch);
    switch (probe_sel) {
#ifdef DIM_RESEARCH
      case 1: hipLaunchKernelGGL(HIP_KERNEL_NAME(gemm_x6_ffn_ln_probe_kernel<1>), lg, dim3(256), 0, s, a); break;
      case 2: hipLaunchKernelGGL(HIP_KERNEL_NAME(gemm_x6_ffn_ln_probe_kernel<2>), lg, dim3(256), 0, s, a); break;
      case 4: hipLaunchKernelGGL(HIP_KERNEL_NAME(gemm_x6_ffn_ln_probe_kernel<4>), lg, dim3(256), 0, s, a); break;
      case 8: hipLaunchKernelGGL(HIP_KERNEL_NAME(gemm_x6_ffn_ln_probe_kernel<8>), lg, dim3(256), 0, s, a); break;
      case 9: hipLaunchKernelGGL(HIP_KERNEL_NAME(gemm_x6_ffn_ln_probe_kernel<9>), lg, dim3(256), 0, s, a); break;
      case 100: hipLaunchKernelGGL(HIP_KERNEL_NAME(gemm_x6_ffn_ln_probe_kernel<0, true>), lg, dim3(256), 0, s, a); break;
      case 104: hipLaunchKernelGGL(HIP_KERNEL_NAME(gemm_x6_ffn_ln_probe_kernel<4, true>), lg, dim3(256), 0, s, a); break;
      case 32: hipLaunchKernelGGL(HIP_KERNEL_NAME(gemm_x6_ffn_ln_probe_kernel<32>), lg, dim3(256), 0, s, a); break;
#endif
      default: hipLaunchKernelGGL(gemm_x6_ffn_ln_kernel, lg, dim3(256), 0, s, a);
    }
    DIM_LAUNCH_CHECK();
    return 0;
  }
  DIM_REQUIRE(a.layer_tab == nullptr || (a.flag != nullptr && a.kv_img == nullptr), "gemm_x6: per-item layer weights exist for the plain blocks");
  const bool small = small_problem(a.M, a.N, batch);
  DIM_REQUIRE(a.kv_img == nullptr || gemm_x6_fuses_kv(a.M, a.n_pad, batch, a.split_mode), "gemm_x6: K|V images need the 128 x 256 or the 32 x 128 block (gemm_x6_fuses_kv)");
  if (small && a.kv_img != nullptr) {
    DIM_REQUIRE(a.bias && a.N % 256 == 0 && a.kv_tiles > 0 && a.R == nullptr && a.relu == 0, "gemm_x6: bad K|V image request");
    DIM_REQUIRE(a.kv_kblock >= 0 && a.kv_vblock == a.kv_kblock + 1 && a.kv_vblock == a.N / 256 - 1, "gemm_x6: the K and V blocks must be the last two");
    const dim3 qg(cdiv(a.M, 32), cdiv(a.N, BN), batch);
#ifdef DIM_RESEARCH
    if (kc_sel == 79 && a.K % 64 == 0) hipLaunchKernelGGL(HIP_KERNEL_NAME(gemm_x6_qkv_small_kernel<64, false>), qg, dim3(256), 0, s, a);   // step-pipelined fragments (A/B)
    else
#endif
    if (a.K == 256) hipLaunchKernelGGL(HIP_KERNEL_NAME(gemm_x6_qkv_small_kernel<64, true, 256>), qg, dim3(256), 0, s, a);
    else if (a.K % 128 == 0) hipLaunchKernelGGL(HIP_KERNEL_NAME(gemm_x6_qkv_small_kernel<64, true>), qg, dim3(256), 0, s, a);
    else hipLaunchKernelGGL(HIP_KERNEL_NAME(gemm_x6_qkv_small_kernel<32, false>), qg, dim3(256), 0, s, a);
    DIM_LAUNCH_CHECK();
    return 0;
  }
  if (small) {
    dim3 grid(cdiv(a.M, 64), cdiv(a.N, BN), batch);
#ifdef DIM_RESEARCH   // the streaming K loop (14 = 63): bit-identical, measured SLOWER than the staged loop at every LightGlue shape (see STREAM above)
    // round 6 probes of the small-problem block (same pieces, same per-accumulator term order: bit-identical): 71 = 64 x 128 with waves 1 x 4 (every weight
    // fragment fetched by ONE wave, pipelined K loop), 74 = 64 x 256 with waves 1 x 4, 76 / 78 = the 32 x 128 block with a double-buffered tile / 32-wide chunks, 70 = round 5's block
    if (a.split_mode == 2 && kc_sel == 71) hipLaunchKernelGGL(HIP_KERNEL_NAME(gemm_x6_kernel<2, 64, 1, 4>), grid, dim3(256), 0, s, a);
    else if (a.split_mode == 2 && kc_sel == 70) hipLaunchKernelGGL(HIP_KERNEL_NAME(gemm_x6_kernel<2, 64>), grid, dim3(256), 0, s, a);   // round 5's product block (waves 2 x 2)
    else if (a.split_mode == 2 && kc_sel == 74 && a.n_pad % 256 == 0) hipLaunchKernelGGL(HIP_KERNEL_NAME(gemm_x6_kernel<2, 64, 2, 4>), dim3(grid.x, cdiv(a.N, 256), grid.z), dim3(256), 0, s, a);
    else if (a.split_mode == 2 && kc_sel == 79 && a.K % 64 == 0 && (a.A1 == nullptr || a.ksplit % 64 == 0)) hipLaunchKernelGGL(HIP_KERNEL_NAME(gemm_x6_small32_kc64_kernel<false>), dim3(cdiv(a.M, 32), grid.y, grid.z), dim3(256), 0, s, a);   // 64-wide chunks, step-pipelined fragments
    else if (a.split_mode == 2 && kc_sel == 78) hipLaunchKernelGGL(HIP_KERNEL_NAME(gemm_x6_kernel<2, 32, 1, 4>), dim3(cdiv(a.M, 32), grid.y, grid.z), dim3(256), 0, s, a);   // the 32 x 128 block with 32-wide chunks
    else if (a.split_mode == 2 && kc_sel == 76) hipLaunchKernelGGL(gemm_x6_small32_db_kernel, dim3(cdiv(a.M, 32), grid.y, grid.z), dim3(256), 0, s, a);
    else
    if (a.split_mode == 2 && kc_sel == 63 && a.K == 256) hipLaunchKernelGGL(HIP_KERNEL_NAME(gemm_x6_stream_kernel<16>), grid, dim3(256), 0, s, a);
    else if (a.split_mode == 2 && kc_sel == 63 && a.K == 512) hipLaunchKernelGGL(HIP_KERNEL_NAME(gemm_x6_stream_kernel<32>), grid, dim3(256), 0, s, a);
    else if (a.split_mode == 2 && kc_sel == 63) hipLaunchKernelGGL(HIP_KERNEL_NAME(gemm_x6_stream_kernel<0>), grid, dim3(256), 0, s, a);
    else
#endif
    // fp16x3, launches of at most two 64-row workgroups per CU (one LightGlue pair per call: 2 x 2048 rows): 32 x 128 blocks, waves 1 x 4 — every wave owns all
    // 32 rows x 32 columns, so a weight fragment is fetched by exactly ONE wave (the 2 x 2 waves of the 64-row block fetch each twice through the CU's vector-memory
    // path, the bound of these launches), twice the workgroups, the k-step-pipelined loop.  Bit-identical to the 64-row block (same pieces, same term order per
    // accumulator).  Measured at 4096 rows (profiles/r06_small_gemm_variants.json): 256 -> 768 13.3 -> 10.8 us, 256 -> 512 10.0 -> 8.5, 512 -> 512 15.0 -> 12.9,
    // 512 -> 256 + residual 13.7 -> 9.7.
    // 64-wide K chunks where K allows (every linear of LightGlue): 256 -> 768 10.7, 256 -> 512 8.0, 512 -> 512 12.8, 512 -> 256 + residual 8.8 us.
    if (a.split_mode == 2 && (long)grid.x * grid.y * grid.z <= 512 && a.K % 128 == 0 && (a.A1 == nullptr || a.ksplit % 64 == 0)) {
      const dim3 g32(cdiv(a.M, 32), grid.y, grid.z);
      if (a.K == 256) hipLaunchKernelGGL(HIP_KERNEL_NAME(gemm_x6_small32_kc64_kernel<true, 256>), g32, dim3(256), 0, s, a);
      else if (a.K == 512) hipLaunchKernelGGL(HIP_KERNEL_NAME(gemm_x6_small32_kc64_kernel<true, 512>), g32, dim3(256), 0, s, a);
      else hipLaunchKernelGGL(HIP_KERNEL_NAME(gemm_x6_small32_kc64_kernel<true>), g32, dim3(256), 0, s, a);
    }
    else if (a.split_mode == 2 && (long)grid.x * grid.y * grid.z <= 512) hipLaunchKernelGGL(HIP_KERNEL_NAME(gemm_x6_kernel<2, 32, 1, 4>), dim3(cdiv(a.M, 32), grid.y, grid.z), dim3(256), 0, s, a);
    else if (a.split_mode == 2) hipLaunchKernelGGL(HIP_KERNEL_NAME(gemm_x6_kernel<2, 64>), grid, dim3(256), 0, s, a);
    else hipLaunchKernelGGL(HIP_KERNEL_NAME(gemm_x6_kernel<1, 64>), grid, dim3(256), 0, s, a);
  } else if (wide_block(a.M, a.n_pad, batch, a.split_mode)) {
    dim3 grid(cdiv(a.M, 128), cdiv(a.N, 256), batch);
    if (a.kv_img != nullptr) {
      DIM_REQUIRE(a.bias && a.N % 256 == 0 && a.kv_tiles > 0 && a.R == nullptr && a.relu == 0, "gemm_x6: bad K|V image request");
      DIM_REQUIRE(a.kv_kblock >= 0 && a.kv_vblock == a.kv_kblock + 1 && a.kv_vblock == (int)grid.y - 1, "gemm_x6: the K and V blocks must be the last two");
#ifdef DIM_RESEARCH
      if (kc_sel == 256) hipLaunchKernelGGL(gemm_x6_qkv256_kernel, dim3(cdiv(a.M, 256), grid.y, grid.z), dim3(256), 0, s, a);
      else if (kc_sel == 37 && a.K >= 64) hipLaunchKernelGGL(gemm_x6_qkv_roll_kernel, grid, dim3(256), 0, s, a);
      else if (kc_sel == 33) hipLaunchKernelGGL(gemm_x6_qkv_db_kernel, grid, dim3(256), 0, s, a);
      else if (kc_sel == 64 && a.K % 64 == 0) hipLaunchKernelGGL(gemm_x6_qkv_kc64_kernel, grid, dim3(256), 0, s, a);
      else
#endif
      hipLaunchKernelGGL(gemm_x6_qkv_kernel, grid, dim3(256), 0, s, a);
    } else switch (probe_sel) {
#ifdef DIM_RESEARCH
      case 1: hipLaunchKernelGGL(HIP_KERNEL_NAME(gemm_x6_probe_kernel<1>), grid, dim3(256), 0, s, a); break;
      case 2: hipLaunchKernelGGL(HIP_KERNEL_NAME(gemm_x6_probe_kernel<2>), grid, dim3(256), 0, s, a); break;
      case 4: hipLaunchKernelGGL(HIP_KERNEL_NAME(gemm_x6_probe_kernel<4>), grid, dim3(256), 0, s, a); break;
      case 8: hipLaunchKernelGGL(HIP_KERNEL_NAME(gemm_x6_probe_kernel<8>), grid, dim3(256), 0, s, a); break;
      case 9: hipLaunchKernelGGL(HIP_KERNEL_NAME(gemm_x6_probe_kernel<9>), grid, dim3(256), 0, s, a); break;
      case 100: hipLaunchKernelGGL(HIP_KERNEL_NAME(gemm_x6_probe_kernel<0, true>), grid, dim3(256), 0, s, a); break;
      case 104: hipLaunchKernelGGL(HIP_KERNEL_NAME(gemm_x6_probe_kernel<4, true>), grid, dim3(256), 0, s, a); break;
      case 116: hipLaunchKernelGGL(HIP_KERNEL_NAME(gemm_x6_probe_kernel<16, true>), grid, dim3(256), 0, s, a); break;
      case 164: hipLaunchKernelGGL(HIP_KERNEL_NAME(gemm_x6_probe_kernel<64, true>), grid, dim3(256), 0, s, a); break;   // round 6: no split VALU
      case 172: hipLaunchKernelGGL(HIP_KERNEL_NAME(gemm_x6_probe_kernel<72, true>), grid, dim3(256), 0, s, a); break;   // + weight fragments loaded once
      case 108: hipLaunchKernelGGL(HIP_KERNEL_NAME(gemm_x6_probe_kernel<8, true>), grid, dim3(256), 0, s, a); break;    // pipelined loop, weight fragments once (the bound of any weight-sharing scheme)
      case 180: hipLaunchKernelGGL(HIP_KERNEL_NAME(gemm_x6_probe_kernel<80, true>), grid, dim3(256), 0, s, a); break;   // no split + no epilogue traffic
      case 48: hipLaunchKernelGGL(HIP_KERNEL_NAME(gemm_x6_probe_kernel<48>), grid, dim3(256), 0, s, a); break;
      case 32: hipLaunchKernelGGL(HIP_KERNEL_NAME(gemm_x6_probe_kernel<32>), grid, dim3(256), 0, s, a); break;
      case 16: hipLaunchKernelGGL(HIP_KERNEL_NAME(gemm_x6_probe_kernel<16>), grid, dim3(256), 0, s, a); break;
#endif
      default:
#ifdef DIM_RESEARCH
        if (kc_sel == 256) hipLaunchKernelGGL(HIP_KERNEL_NAME(gemm_x6_kernel<2, 256, 2, 4>), dim3(cdiv(a.M, 256), grid.y, grid.z), dim3(256), 0, s, a);
        else if (kc_sel == 37 && a.K >= 64) hipLaunchKernelGGL(gemm_x6_wide_roll_kernel, grid, dim3(256), 0, s, a);
        else if (kc_sel == 33) hipLaunchKernelGGL(gemm_x6_wide_db_kernel, grid, dim3(256), 0, s, a);
        else if (kc_sel == 64 && a.K % 64 == 0 && (a.A1 == nullptr || a.ksplit % 64 == 0)) hipLaunchKernelGGL(gemm_x6_wide_kc64_kernel, grid, dim3(256), 0, s, a);
        else
#endif
        hipLaunchKernelGGL(HIP_KERNEL_NAME(gemm_x6_kernel<2, 128, 2, 4>), grid, dim3(256), 0, s, a);
    }
  } else {
    dim3 grid(cdiv(a.M, 128), cdiv(a.N, BN), batch);
    if (a.split_mode == 2) hipLaunchKernelGGL(HIP_KERNEL_NAME(gemm_x6_kernel<2, 128>), grid, dim3(256), 0, s, a);
    else hipLaunchKernelGGL(HIP_KERNEL_NAME(gemm_x6_kernel<1, 128>), grid, dim3(256), 0, s, a);
  }
  DIM_LAUNCH_CHECK();
  return 0;
}

// Host: nn.Linear-style operand [K][N] fp32 -> 16-bit planes [plane][n_pad/32][K/16][k-half][32 cols][8 k].
// mode 1: three bf16 planes, round-to-nearest-even pieces (the same rule as split3_pk on the device);
// mode 2: two fp16 planes of w * 2^sw (max|w| scaled into [8192, 16384)), inv_scale = 1 / (2^sw * DIM_F16_ACT_SCALE).
static unsigned short host_bf16_rne(float x) {
  unsigned u;
  memcpy(&u, &x, 4);
  if ((u & 0x7f800000u) == 0x7f800000u) return (unsigned short)(u >> 16);  // inf / nan: truncate
  u += 0x7fffu + ((u >> 16) & 1u);
  return (unsigned short)(u >> 16);
}
static size_t gemm_split_piece_elems(int K, int n_pad, int mode) { return (size_t)(mode == 2 ? 2 : 3) * n_pad * K; }
size_t gemm_split_weight_elems(int K, int n_pad, int mode) { return gemm_split_piece_elems(K, n_pad, mode) + 2 * (size_t)n_pad; }
void split_weights(const float* w_kn, int K, int N, int n_pad, int mode, unsigned short* out, SplitWeights* sw, int kperm) {
  const int NB = n_pad / 32, KS = K / 16, npl = mode == 2 ? 2 : 3;
  sw->mode = mode; sw->n_pad = n_pad; sw->scale_off = gemm_split_piece_elems(K, n_pad, mode);
  for (size_t i = 0; i < gemm_split_weight_elems(K, n_pad, mode); ++i) out[i] = 0;
  // per-column (= output feature) power-of-two scale: max|w[:, n]| lands in [8192, 16384); inverse scales in the tail
  std::vector<float> wsc(n_pad, 1.0f);
  float* inv = (float*)(out + sw->scale_off);
  for (int n = 0; n < n_pad; ++n) {
    float inv_n = 1.0f;
    if (mode == 2) {
      float mx = 0.f;
      if (n < N)
        for (int k = 0; k < K; ++k) mx = fmaxf(mx, fabsf(w_kn[(size_t)k * N + n]));
      int e = 0;
      if (mx > 0.f && mx < INFINITY) { frexpf(mx, &e); wsc[n] = ldexpf(1.0f, 14 - e); }
      inv_n = 1.0f / (wsc[n] * DIM_F16_ACT_SCALE);
    }
    memcpy(&inv[n], &inv_n, 4);
  }
  for (int k = 0; k < K; ++k)
    for (int n = 0; n < N; ++n) {
      float x = w_kn[(size_t)k * N + n] * wsc[n];
      const int nb = n / 32, j = n % 32, ks = k / 16, pos = k % 16;
      const int hf = kperm ? (pos >> 2) & 1 : pos / 8, e = kperm ? ((pos >> 3) << 2) | (pos & 3) : pos % 8;
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
        out[(((((size_t)p * NB + nb) * KS + ks) * 2 + hf) * 32 + j) * 8 + e] = bits;
        x = x - piece;
      }
    }
}

#ifdef DIM_FFN_TIMERS
// instrumented build only: [0..15] summed s_memtime ticks (100 MHz) per phase over all waves, [16] the same for the whole wave, [17] waves that reported (one workgroup in 16), [18] s_memrealtime ticks (100 MHz) for the whole wave
extern "C" int dim_ffn_phase_read(unsigned long long* host24, int reset) {
  DIM_HIP(hipMemcpyFromSymbol(host24, HIP_SYMBOL(g_ffn_phase), 24 * sizeof(unsigned long long)));
  if (reset) {
    const unsigned long long z[24] = {};
    DIM_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_ffn_phase), z, sizeof(z)));
  }
  return 0;
}
#endif
