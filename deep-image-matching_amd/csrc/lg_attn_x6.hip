// LightGlue attention on the bf16 matrix cores at fp32 accuracy ("bf16x6"; see dim_common.h split3_pk,
// gemm_x6.hip).  Same algorithm and dataflow as lg_attn.hip — transposed score tile
// S^T[key][query] = K·Q^T so that every lane owns one query column, probabilities fed back as the B
// operand of O^T[d][query] += V^T·P^T without leaving registers, log2-domain online softmax with
// deferred rescale — but each 32x32x16 step is six v_mfma_f32_32x32x16_bf16 (192 cycles) instead of
// eight v_mfma_f32_32x32x2_f32 (512 cycles).
//
// Operand images (one 16-B ds_read_b128 = one MFMA operand, consecutive lanes -> consecutive slots):
//   Kp[plane][d-block 8][key 32][8 bf16]          A operand of K·Q^T   (k = d); d-block 2s + h holds the head dims
//       16s + 4h + {0,1,2,3, 8,9,10,11} — the 8 registers a lane of gemm_x6.hip's transposed K block owns — and Q uses the
//       same assignment (a dot product does not care in which order its dims are paired with MFMA k-slots)
//   Vp[plane][step 2][k-half 2][d 64][8 keys]     A operand of V^T·P^T (k = key), keys stored in the
//       order in which the MFMA C layout of S^T holds them: register 8u+e of lane-half h holds key
//       (e&3) + 8*(2u + (e>>2)) + 4h, so registers 8u..8u+7 of P ARE the B operand of step u.
//   Q is split once into registers (4 steps x 3 planes), with the rotary embedding applied on load.
// K and V are split ONCE per layer by kv_prep_kernel (rotary applied to K for self-attention) into
// per-tile images laid out exactly like Kp / Vp, so the 16 query blocks that share a (item, head)
// stage a tile with six straight 16-byte copies per thread (no per-block split work, no LDS bank
// conflicts: consecutive lanes write consecutive slots).
#include <math.h>

#include "lg_kernels.h"

namespace {

struct AttnArgs6 {
  const float* q; const float* k; const float* v; float* o;
  int ldq, ldk, ldv, ldo;
  long long sq, sk, sv, so;
  const int* n; const int* done;
  int cross;
  float scale;
  const float* enc;      // [items][nmax][64] cos|sin; rotary is applied to q (and k) iff !cross
  u32x4* kv_img;         // [items][4 heads][tiles][1536] pre-split K|V tile images (kv_prep_kernel)
  int nmax, tiles;
  int splits;            // key range cut into `splits` parts per (query block, head, item): small batches only
  int qblocks, groups;   // workgroups per (item, head) group = qblocks (incl. splits); groups = 4 * items (XCD-aware 1-D grid)
  float* part;           // [items][4][nmax][splits][PART] partial (unnormalised O, running max, running sum)
  int probe;             // timing probes of the shared-score-tile question (dim_tune_set key 12); 0 in the product
  unsigned* sat;         // fp16x3 range guard on the rotated K (the rotation can grow |k| by sqrt 2) and on V
  int q_img;             // cross attention with images written by the projection GEMM (gemm_x6.hip KV): there is no fp32 qk —
                         // the Q operand is this item's own K image, and the softmax scale is applied to the scores instead
};
constexpr int PART = 68;   // 64 output dims + m + l, padded to a 16-byte multiple

constexpr int TILE_STRIDE = KV_TILE_STRIDE;  // slots of 16 B reserved per tile image in HBM (mode 1 fills all 1536)
constexpr int tile_slots(int npl) { return npl * 512; }    // npl * 256 K slots + npl * 256 V slots

// One workgroup per (key tile, head, item): rotary on K (self-attention only, LGN:41-54,155-156), exact
// 3-way bf16 split of K and V, written in the LDS-image order of attn_x6_kernel.
template <int MODE>
__global__ __launch_bounds__(256) void kv_prep_kernel(AttnArgs6 a) {
  using S = SplitMma<MODE>;
  constexpr int NPL = S::NPL, KSL = NPL * 256;
  const int tile = blockIdx.x, head = blockIdx.y, item = blockIdx.z;
  if (a.done[item >> 1] != 0) return;
  const int nk = a.n[item], kt = tile * 32;
  if (kt >= nk) return;
  const int t = threadIdx.x;
  const float* kb = a.k + (size_t)item * a.sk + head * 64;
  const float* vb = a.v + (size_t)item * a.sv + head * 64;
  u32x4* img = a.kv_img + (((size_t)item * 4 + head) * a.tiles + tile) * TILE_STRIDE;
  {  // K: thread (key = t>>3, d-block = t&7 = 2s + h: dims 16s + 4h + {0..3, 8..11})
    const int key = t >> 3, blk = t & 7, d0 = 16 * (blk >> 1) + 4 * (blk & 1);
    float x[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (kt + key < nk) {
      const float* p = kb + (size_t)(kt + key) * a.ldk + d0;
      const float4 x0 = *(const float4*)p, x1 = *(const float4*)(p + 8);
      x[0] = x0.x; x[1] = x0.y; x[2] = x0.z; x[3] = x0.w; x[4] = x1.x; x[5] = x1.y; x[6] = x1.z; x[7] = x1.w;
      if (!a.cross) {
        const float* e = a.enc + ((size_t)item * a.nmax + kt + key) * 64 + (d0 >> 1);  // pairs d0/2, d0/2 + 1, d0/2 + 4, d0/2 + 5
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int f = (i & 1) + 4 * (i >> 1);
          const float c = e[f], sn = e[32 + f], t0 = x[2 * i], t1 = x[2 * i + 1];
          x[2 * i] = t0 * c + (-t1) * sn;
          x[2 * i + 1] = t1 * c + t0 * sn;
        }
      }
    }
    unsigned pc[4][NPL];
    float vmax = 0.0f;
#pragma unroll
    for (int i = 0; i < 4; ++i) { S::split(x[2 * i], x[2 * i + 1], S::act_scale(), pc[i]); vmax = sat_track(vmax, x[2 * i], x[2 * i + 1]); }
    if (MODE == 2) sat_report(a.sat, vmax);
#pragma unroll
    for (int pl = 0; pl < NPL; ++pl) img[(pl * 8 + blk) * 32 + key] = u32x4{pc[0][pl], pc[1][pl], pc[2][pl], pc[3][pl]};
  }
  {  // V: thread (d = t&63, step u = t>>7, k-half h = (t>>6)&1), keys in MFMA-C-layout order
    const int d = t & 63, u = t >> 7, hh = (t >> 6) & 1;
    float x[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int key = (e & 3) + 8 * (2 * u + (e >> 2)) + 4 * hh;
      x[e] = (kt + key < nk) ? vb[(size_t)(kt + key) * a.ldv + d] : 0.0f;
    }
    unsigned pc[4][NPL];
#pragma unroll
    for (int i = 0; i < 4; ++i) S::split(x[2 * i], x[2 * i + 1], S::act_scale(), pc[i]);
#pragma unroll
    for (int pl = 0; pl < NPL; ++pl) img[KSL + ((pl * 2 + u) * 2 + hh) * 64 + d] = u32x4{pc[0][pl], pc[1][pl], pc[2][pl], pc[3][pl]};
  }
}

// fp16x3: 16 KB of LDS and <= 128 VGPRs -> 4 workgroups per CU, so the 1024 workgroups of a 16-item batch
// (16 query blocks x 4 heads x 16 items) run as ONE round on 256 CUs instead of 1.33 rounds at 3 per CU
// PROBE: the round-3 timing probes (dim_tune_set key 12) as a TEMPLATE parameter: as a run-time test around the score MFMAs they forced the
// product kernel to materialise the zeroed score accumulator (16 v_mov per key tile; every VALU instruction is time)
// NBUF: key-tile images in LDS.  2 (the product): the next tile arrives while this one is consumed.  3 (round 5 prototype, research build, dim_tune_set key
// 12 = 22, the key-split launches of SMALL batches — one pair per call through the plugin hooks): two tiles ahead; the wait in front of the barrier then
// retires the OLDEST transfer only (vector-memory operations complete in order: vmcnt(4) leaves the newer tile's four 16-byte copies in flight).  The
// idea: with 2 workgroups per CU instead of 3.6 nothing else covers the L2 / HBM round trip of a 16-KB tile image.  MEASURED (profiles/r05_batch1_calls.json):
// one 2048 x 2048 pair 1.953 / 1.958 ms against 1.920 / 1.918 with one tile ahead — the 2.2 us a workgroup spends per key tile are its own dependent
// chain (12 MFMAs into one score accumulator, the softmax, 12 more), not the transfer.  Bit-identical; kept for the record and for the emulator's
// partial-wait model (tests/test_lightglue_emu.py).
template <int MODE, int PROBE = 0, int NBUF = 2>
__global__ __launch_bounds__(256, (MODE == 2 ? 4 : 2)) void attn_x6_kernel(AttnArgs6 a) {
  static_assert(NBUF == 2 || (NBUF == 3 && MODE == 2 && PROBE == 0), "three tile images exist for the fp16x3 product kernel");
  using S = SplitMma<MODE>;
  constexpr int NPL = S::NPL, TILE_SLOTS = tile_slots(NPL), KSL = NPL * 256, NCP = TILE_SLOTS / 256;
  // K, Q, V and P are multiplied by the (power-of-two) activation scale before the split: exact factors
  const float inv_qk = (1.0f / (S::act_scale() * S::act_scale())) * (a.q_img ? a.scale * 1.44269504088896340736f : 1.0f);
  // XCD-aware mapping of a 1-D grid (cdna_hip_programming.md T1): the hardware sends workgroup L to XCD L % 8, and all the
  // query blocks of one (item, head) group stream the SAME K | V tile images.  Giving every group to ONE XCD (group =
  // (L / 8 / qblocks) * 8 + L % 8) makes 15 of its 16 workgroups hit that XCD's L2 instead of each XCD fetching its own
  // copy (measured before: 3.7 GB of FETCH per launch for 0.4 GB of K | V images).  Speed only: any mapping is correct.
  const int L = blockIdx.x, xcd = L & 7, slot = L >> 3;
  const int group = (slot / a.qblocks) * 8 + xcd, bx = slot % a.qblocks;
  if (group >= a.groups) return;
  const int item = group >> 2, head = group & 3, qb = bx / a.splits, sp = bx - qb * a.splits, q0 = qb * 128;
  if (a.done[item >> 1] != 0) return;
  const int kitem = a.cross ? (item ^ 1) : item;
  const int nq = a.n[item], nk = a.n[kitem];
  if (q0 >= nq) return;

  // two tile images: the next key tile arrives by LDS-DMA (global_load_lds_dwordx4: no staging registers, no ds_write pass)
  // in the other half while this one is consumed — one barrier per tile
  __shared__ u32x4 img[NBUF * TILE_SLOTS];

  const int t = threadIdx.x, lane = t & 63, wv = t >> 6, lx = lane & 31, half = lane >> 5;
  const int qrow = q0 + wv * 32 + lx;
  const bool qok = qrow < nq;

  // Q^T operand: lane (query lx, half) holds d = 16s + 4*half + {0..3, 8..11} for s = 0..3 (the K image's assignment), NPL planes each
  u32x4 qf[4][NPL];
  const float sc = a.scale * 1.44269504088896340736f;  // scores in the log2 domain
  if (a.q_img) {  // the item's own K image IS the split Q operand (cross attention: q and k are the same projection)
    // (a wave whose 32 queries all lie past the ragged end reads the last written tile: its results are never stored)
    const u32x4* qi = a.kv_img + (((size_t)item * 4 + head) * a.tiles + min((q0 + wv * 32) >> 5, (nq - 1) >> 5)) * TILE_STRIDE;
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
      for (int pl = 0; pl < NPL; ++pl) qf[s][pl] = qi[(pl * 8 + 2 * s + half) * 32 + lx];
  } else {
    const float* qp = a.q + (size_t)item * a.sq + (size_t)(qok ? qrow : 0) * a.ldq + head * 64 + half * 4;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      float x[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      if (qok) {
        const float4 x0 = *(const float4*)(qp + 16 * s), x1 = *(const float4*)(qp + 16 * s + 8);
        x[0] = x0.x; x[1] = x0.y; x[2] = x0.z; x[3] = x0.w; x[4] = x1.x; x[5] = x1.y; x[6] = x1.z; x[7] = x1.w;
        if (!a.cross) {  // rotary (LGN:41-54,155): pair i of the lane <-> frequency 8s + 2half + (i & 1) + 4 (i >> 1)
          const float* e = a.enc + ((size_t)item * a.nmax + qrow) * 64 + 8 * s + 2 * half;
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const int f = (i & 1) + 4 * (i >> 1);
            const float c = e[f], sn = e[32 + f], t0 = x[2 * i], t1 = x[2 * i + 1];
            x[2 * i] = t0 * c + (-t1) * sn;
            x[2 * i + 1] = t1 * c + t0 * sn;
          }
        }
      }
      unsigned pc[4][NPL];
#pragma unroll
      for (int i = 0; i < 4; ++i) S::split(x[2 * i] * sc, x[2 * i + 1] * sc, S::act_scale(), pc[i]);
#pragma unroll
      for (int pl = 0; pl < NPL; ++pl) qf[s][pl] = u32x4{pc[0][pl], pc[1][pl], pc[2][pl], pc[3][pl]};
    }
  }
  f32x16 oacc[2];
#pragma unroll
  for (int n = 0; n < 2; ++n)
#pragma unroll
    for (int r = 0; r < 16; ++r) oacc[n][r] = 0.0f;
  float m_run = -INFINITY, l_run = 0.0f;

  // staging: NCP (6 / 4) straight 16-byte copies per thread per tile (images pre-built by kv_prep_kernel)
  const u32x4* src = a.kv_img + ((size_t)kitem * 4 + head) * a.tiles * TILE_STRIDE;
  auto load_tile = [&](int kt, int buf) {  // wave w moves items w * 64 + 256 i + lane
    const u32x4* p = src + (size_t)(kt >> 5) * TILE_STRIDE;
#pragma unroll
    for (int i = 0; i < NCP; ++i) lds_dma16(p + wv * 64 + 256 * i + lane, &img[buf * TILE_SLOTS + wv * 64 + 256 * i]);
  };
  // this workgroup's share of the key tiles (all of them unless the launcher split the key range)
  const int per = ((nk + 31) / 32 + a.splits - 1) / a.splits * 32;
  const int kt0 = sp * per, kt1 = min(nk, kt0 + per);
  if (kt0 < kt1) load_tile(kt0, 0);
  if (NBUF == 3 && kt0 + 32 < kt1) load_tile(kt0 + 32, 1);
#if defined(__AMDGCN__)
  if (NBUF == 3) {
    // the Q fragments of a cross launch are plain loads that nothing consumes before the loop: retire them HERE (the empty asm makes them inputs),
    // or the compiler, which cannot see that the partial wait below covers them, puts a vmcnt(0) in front of every tile's first MFMA
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
      for (int pl = 0; pl < NPL; ++pl) asm volatile("" : "+v"(qf[s][pl]));
  }
#endif
  int buf = 0;
  for (int kt = kt0; kt < kt1; kt += 32, buf = (NBUF == 2 ? buf ^ 1 : (buf == 2 ? 0 : buf + 1))) {
    // the compiler does not order the barrier after the DMA by itself; after it every wave has also left the tile that
    // lived in the buffer the next transfer overwrites
    if (NBUF == 2) {
      lds_dma_wait_all();
      __syncthreads();
      if (kt + 32 < kt1) load_tile(kt + 32, buf ^ 1);
    } else {
      // this tile's transfer is the oldest in flight; the next tile's (if there is one) stays in flight.  lgkmcnt(0): this wave's LDS reads of the
      // previous tile are done (their MFMAs were issued).  A plain s_barrier: __syncthreads() would bring its own vmcnt(0).
      if (kt + 32 < kt1) __builtin_amdgcn_s_waitcnt(0x0074); else __builtin_amdgcn_s_waitcnt(0x0070);
      __builtin_amdgcn_s_barrier();
      if (kt + 64 < kt1) load_tile(kt + 64, buf == 0 ? 2 : buf - 1);
    }
    const u32x4* Kp = img + buf * TILE_SLOTS;
    const u32x4* Vp = Kp + KSL;

    // ---- S^T = K · Q^T : 4 steps x 6 cross terms ----
    f32x16 sacc;
#pragma unroll
    for (int r = 0; r < 16; ++r) sacc[r] = 0.0f;
    // (probe 1 — timing only, results wrong: every second key tile of a cross launch runs WITHOUT its score MFMAs = the 25 % of the
    // launch's MFMAs a score tile shared by the two directions would not execute.  Every second TILE rather than every second
    // item: the XCD-aware mapping above sends the odd items to XCDs 4..7, which would idle while 0..3 set the launch time.)
    if (!(PROBE == 1 && (kt & 32))) {
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        u32x4 kf[NPL];
#pragma unroll
        for (int p = 0; p < NPL; ++p) kf[p] = Kp[(p * 8 + 2 * s + half) * 32 + lx];
#pragma unroll
        for (int tm = 0; tm < S::NT; ++tm) sacc = S::mma(kf[S::ta(tm)], qf[s][S::tb(tm)], sacc);
      }
    }
    // fp16x3: sacc holds scale^2 x the scores; the exact power-of-two factor is applied inside the fused
    // multiply-add of the exponent below (and once to the tile maximum) instead of to all 16 values

    // ---- online softmax (log2 domain, deferred rescale) ----
    if (kt + 32 > nk) {
#pragma unroll
      for (int r = 0; r < 16; ++r)
        if (kt + mfma_row(r, half) >= nk) sacc[r] = -INFINITY;
    }
    float tmax = fmaxf(fmaxf(sacc[0], sacc[1]), sacc[2]);
#pragma unroll
    for (int r = 3; r < 15; r += 2) tmax = fmaxf(fmaxf(tmax, sacc[r]), sacc[r + 1]);
    tmax = fmaxf(tmax, sacc[15]);
    tmax = half_pair_max(tmax) * inv_qk;
    constexpr float RESCALE_LOG2 = 8.0f;
    if (__any(tmax > m_run + RESCALE_LOG2)) {
      const float m_new = fmaxf(m_run, tmax);
      const float alpha = exp2f(m_run - m_new);
      l_run *= alpha;
#pragma unroll
      for (int n = 0; n < 2; ++n)
#pragma unroll
        for (int r = 0; r < 16; ++r) oacc[n][r] *= alpha;
      m_run = m_new;
    }
    const float p_shift = MODE == 2 ? 4.0f : 0.0f;  // log2(DIM_F16_ACT_SCALE)
    float p[16];
    float psum = 0.0f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      // sacc * 2^-k is exact: one rounding, as sacc' - m_run.  fp16x3: the probabilities are produced already multiplied
      // by the activation scale (2^4 added to the exponent), so their split needs neither a multiply nor a clamp
      // (p <= 2^RESCALE_LOG2 * 16 = 4096); l_run then carries the same factor, undone once at the end
      p[r] = __builtin_amdgcn_exp2f(fmaf(sacc[r], inv_qk, p_shift - m_run));
      psum += p[r];
    }
    l_run += psum;

    // ---- O^T += V^T · P^T : registers 8u..8u+7 of p are the B operand of step u ----
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      unsigned pc[4][NPL];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        if (MODE == 2) split2_pk_raw(p[8 * u + 2 * e], p[8 * u + 2 * e + 1], pc[e][0], pc[e][1]);
        else S::split(p[8 * u + 2 * e], p[8 * u + 2 * e + 1], S::act_scale(), pc[e]);
      }
      u32x4 pf[NPL];
#pragma unroll
      for (int pl = 0; pl < NPL; ++pl) pf[pl] = u32x4{pc[0][pl], pc[1][pl], pc[2][pl], pc[3][pl]};
      u32x4 vf[2][NPL];
#pragma unroll
      for (int pl = 0; pl < NPL; ++pl) {
        vf[0][pl] = Vp[((pl * 2 + u) * 2 + half) * 64 + lx];
        vf[1][pl] = Vp[((pl * 2 + u) * 2 + half) * 64 + 32 + lx];
      }
#pragma unroll
      for (int tm = 0; tm < S::NT; ++tm) {
        oacc[0] = S::mma(vf[0][S::ta(tm)], pf[S::tb(tm)], oacc[0]);
        oacc[1] = S::mma(vf[1][S::ta(tm)], pf[S::tb(tm)], oacc[1]);
      }
    }
    // (probe 3 — timing only: the second direction of a shared score tile cannot keep its output across query blocks; per
    // 32-key tile the workgroup would write a 32 x (64 + m + l) partial record.  Every second tile writes a record of that size
    // here: the volume of one direction, spread over all XCDs.)
    if (PROBE == 3 && (kt & 32)) {
      float* pp = a.part + (((((size_t)item * 4 + head) * ((a.tiles + 1) >> 1) + (kt >> 6)) * (a.qblocks / a.splits) + qb) * 256 + t) * 8;
      *(float4*)pp = make_float4(oacc[0][0], oacc[0][1], oacc[0][2], oacc[0][3]);
      *(float4*)(pp + 4) = make_float4(oacc[1][0], oacc[1][1], oacc[1][2], oacc[1][3]);
    }
  }

  // fp16x3: the V·P accumulator holds scale^2 x the sum and l_run scale x the normaliser: divide by scale * l_run
  const float l_both = half_pair_sum(l_run);
  if (a.splits > 1) {  // partial result for attn_combine_kernel
    if (qok) {
      float* pp = a.part + ((((size_t)item * 4 + head) * a.nmax + qrow) * a.splits + sp) * PART;
#pragma unroll
      for (int n = 0; n < 2; ++n)
#pragma unroll
        for (int g = 0; g < 4; ++g)
          *(float4*)(pp + n * 32 + 8 * g + 4 * half) = make_float4(oacc[n][4 * g], oacc[n][4 * g + 1], oacc[n][4 * g + 2], oacc[n][4 * g + 3]);
      if (half == 0) { pp[64] = m_run; pp[65] = l_both; }
    }
    return;
  }
  const float l_tot = l_both * (MODE == 2 ? S::act_scale() : 1.0f);
  if (qok) {
    float* op = a.o + (size_t)item * a.so + (size_t)qrow * a.ldo + head * 64;
#pragma unroll
    for (int n = 0; n < 2; ++n)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        float4 o = make_float4(0.f, 0.f, 0.f, 0.f);  // LGN:103-104: empty key set -> zeros
        if (nk > 0) o = make_float4(oacc[n][4 * g] / l_tot, oacc[n][4 * g + 1] / l_tot, oacc[n][4 * g + 2] / l_tot, oacc[n][4 * g + 3] / l_tot);
        *(float4*)(op + n * 32 + 8 * g + 4 * half) = o;
      }
  }
}
// merges the partial results of a split key range: 16 threads per (query row, head), four output dims each — every load a 16-byte one and all of a thread's
// loads in flight together (until round 6 one wave per (row, head) with 4-byte loads in a dependent loop: 10.5 us per launch of a one-pair batch for 22 MB of
// traffic; the same arithmetic per value, bit-identical context rows)
template <int SPLITS>
__global__ __launch_bounds__(256) void attn_combine_kernel(AttnArgs6 a, float out_scale) {
  const int item = blockIdx.z, head = blockIdx.y, row = blockIdx.x * 16 + (threadIdx.x >> 4), d4 = (threadIdx.x & 15) * 4;
  if (a.done[item >> 1] != 0 || row >= a.n[item]) return;
  const int splits = SPLITS ? SPLITS : a.splits;
  const float* pp = a.part + (((size_t)item * 4 + head) * a.nmax + row) * splits * PART;
  float4 o[SPLITS ? SPLITS : 1];
  float m[SPLITS ? SPLITS : 1], l[SPLITS ? SPLITS : 1];
  float M = -INFINITY, L = 0.f;
  float4 O = make_float4(0.f, 0.f, 0.f, 0.f);
  if (SPLITS) {
#pragma unroll
    for (int s = 0; s < SPLITS; ++s) { o[s] = *(const float4*)(pp + s * PART + d4); m[s] = pp[s * PART + 64]; l[s] = pp[s * PART + 65]; }
#pragma unroll
    for (int s = 0; s < SPLITS; ++s) M = fmaxf(M, m[s]);
#pragma unroll
    for (int s = 0; s < SPLITS; ++s) {
      const float w = (m[s] == -INFINITY) ? 0.f : exp2f(m[s] - M);
      L += w * l[s];
      O.x += w * o[s].x; O.y += w * o[s].y; O.z += w * o[s].z; O.w += w * o[s].w;
    }
  } else {
    for (int s = 0; s < splits; ++s) M = fmaxf(M, pp[s * PART + 64]);
    for (int s = 0; s < splits; ++s) {
      const float ms = pp[s * PART + 64];
      const float w = (ms == -INFINITY) ? 0.f : exp2f(ms - M);
      const float4 os = *(const float4*)(pp + s * PART + d4);
      L += w * pp[s * PART + 65];
      O.x += w * os.x; O.y += w * os.y; O.z += w * os.z; O.w += w * os.w;
    }
  }
  const float den = L * out_scale;
  const float4 r = (L > 0.f) ? make_float4(O.x / den, O.y / den, O.z / den, O.w / den) : make_float4(0.f, 0.f, 0.f, 0.f);  // no keys -> zeros (LGN:103-104)
  *(float4*)(a.o + (size_t)item * a.so + (size_t)row * a.ldo + head * 64 + d4) = r;
}
}  // namespace

int launch_lg_attention_x6(const LgState& st, int cross, hipStream_t s, int kv_ready) {
  AttnArgs6 a;
  const long long is = (long long)st.nmax * 768;
  if (!cross) { a.q = st.qkv; a.k = st.qkv + 256; a.v = st.qkv + 512; }
  else { a.q = st.qkv; a.k = st.qkv; a.v = st.qkv + 256; }
  a.ldq = a.ldk = a.ldv = 768; a.sq = a.sk = a.sv = is;
  a.o = st.ctx; a.ldo = 256; a.so = (long long)st.nmax * 256;
  a.n = st.n_cur; a.done = st.done; a.cross = cross;
  a.scale = 0.125f;
  a.sat = st.sat_qkv;
  a.q_img = (kv_ready && cross) ? 1 : 0;
  a.enc = st.enc; a.kv_img = (u32x4*)st.kv_img; a.nmax = st.nmax; a.tiles = cdiv(st.nmax, 32);
  // small batches leave most CUs idle and make every workgroup walk all key tiles alone: cut the key range
  const int nsel = st.nsel > 0 ? st.nsel : st.nmax;   // rows that can be live in this call: the query blocks / merge rows beyond them would only exit
  const int wgs = cdiv(nsel, 128) * 4 * st.n_items;
  a.part = st.attn_part;
  // ... into the fewest parts (1, 2, 4) that give the launch two workgroups per CU (512).  (Until round 6: 4 parts up to 128 workgroups, 2 up to 256, else 1 —
  // one pair of 2304 keypoints ran 288 workgroups of 36 tiles, 1.1 per CU: 2.01 ms against 1.52 at 2048; three pairs of 2048 ran 384 of 64 tiles.)
  a.splits = (st.attn_part && st.n_items <= st.attn_part_items) ? (wgs >= 512 ? 1 : (wgs >= 256 ? 2 : 4)) : 1;
  a.probe = cross ? dim_attn_probe() : 0;
  if (a.probe == 2) a.splits = 16;  // probe 2: every query block's key range in 16 parts -> the partial-record volume of a shared score tile, written AND merged
  if (dim_attn_probe() == 8 || dim_attn_probe() == 16) { a.probe = 0; if (a.splits > 1) a.splits = dim_attn_probe(); }   // research: a finer key split for small batches (results stay correct)
  const bool two_ahead = dim_attn_probe() == 22;   // research, 22: the key-split launches with TWO tiles of prefetch distance (prototype; results identical, measured slower)
  if (two_ahead) a.probe = 0;
  (void)two_ahead;
  a.qblocks = cdiv(nsel, 128) * a.splits;
  a.groups = 4 * st.n_items;
  dim3 grid((unsigned)(cdiv(a.groups, 8) * 8 * a.qblocks));  // whole rounds of 8 groups, one per XCD (surplus workgroups exit)
  if (dim_precision_mode() == 2) {
    if (!kv_ready) hipLaunchKernelGGL(HIP_KERNEL_NAME(kv_prep_kernel<2>), dim3(a.tiles, 4, st.n_items), dim3(256), 0, s, a);
#ifdef DIM_RESEARCH   // cross-attention timing probes (wrong results by design): research build only
    if (a.probe == 1) hipLaunchKernelGGL(HIP_KERNEL_NAME(attn_x6_kernel<2, 1>), grid, dim3(256), 0, s, a);
    else if (a.probe == 3) hipLaunchKernelGGL(HIP_KERNEL_NAME(attn_x6_kernel<2, 3>), grid, dim3(256), 0, s, a);
    else
    if (a.splits > 1 && two_ahead) hipLaunchKernelGGL(HIP_KERNEL_NAME(attn_x6_kernel<2, 0, 3>), grid, dim3(256), 0, s, a);
    else
#endif
    hipLaunchKernelGGL(HIP_KERNEL_NAME(attn_x6_kernel<2>), grid, dim3(256), 0, s, a);
  } else {
    hipLaunchKernelGGL(HIP_KERNEL_NAME(kv_prep_kernel<1>), dim3(a.tiles, 4, st.n_items), dim3(256), 0, s, a);
    hipLaunchKernelGGL(HIP_KERNEL_NAME(attn_x6_kernel<1>), grid, dim3(256), 0, s, a);
  }
  if (a.splits > 1) {
    const float out_scale = dim_precision_mode() == 2 ? DIM_F16_ACT_SCALE : 1.0f;
    const dim3 cg(cdiv(nsel, 16), 4, st.n_items);
    if (a.splits == 4) hipLaunchKernelGGL(HIP_KERNEL_NAME(attn_combine_kernel<4>), cg, dim3(256), 0, s, a, out_scale);
    else if (a.splits == 2) hipLaunchKernelGGL(HIP_KERNEL_NAME(attn_combine_kernel<2>), cg, dim3(256), 0, s, a, out_scale);
    else hipLaunchKernelGGL(HIP_KERNEL_NAME(attn_combine_kernel<0>), cg, dim3(256), 0, s, a, out_scale);
  }
  DIM_LAUNCH_CHECK();
  return 0;
}
