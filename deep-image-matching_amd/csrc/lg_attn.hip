// LightGlue attention for gfx950: exact-fp32 flash attention on
// v_mfma_f32_32x32x2_f32 (reference: self LGN:102-126,146-159; cross LGN:186-211).
//
// One kernel serves self-attention (keys/values from the same item) and both
// directions of cross-attention (keys/values from the partner item, item^1).
// The reference's CPU cross path builds ONE similarity and soft-maxes it along
// rows and along columns (LGN:197-206); softmax_rows(sim)·v1 and
// softmax_cols(sim)^T·v0 are exactly attention(qk0→qk1,v1) and
// attention(qk1→qk0,v0), so the 64 MB similarity is never materialised.
//
// Per wave: 32 queries.  The score tile is computed TRANSPOSED,
//   S^T[key][query] = K_tile(32x64) · Q^T(64x32),
// so the MFMA C layout puts one query per lane column (lane&31) and 16 keys in
// that lane's registers: the running max / sum are (almost) in-lane reductions
// (one xor-32 shuffle), and — the point of the layout — the probabilities can be
// fed straight back as the B operand of
//   O^T[d][query] += V^T[d][key] · P^T[key][query]
// without leaving registers: register r of lane-half h IS P^T[key=row(r,h)][query],
// the contraction order over keys is simply permuted (row(r,0), row(r,1) per step).
// O^T again has one query per lane, so the online-softmax rescale is in-lane too.
//
// Workgroup = 4 waves = 128 queries of one (item, head); K/V tiles of 32 keys are
// staged once per workgroup in LDS (K with a 65-dword row stride so the 32 lanes
// reading one d of 32 keys hit 32 banks; V rows are read along d: conflict free).
#include <math.h>

#include "lg_kernels.h"

namespace {

struct AttnArgs {
  const float* q; const float* k; const float* v; float* o;
  int ldq, ldk, ldv, ldo;
  long long sq, sk, sv, so;
  const int* n; const int* done;
  int cross;
  float scale;
};

__global__ __launch_bounds__(256, 3) void attn_kernel(AttnArgs a) {
  const int item = blockIdx.z, head = blockIdx.y, q0 = blockIdx.x * 128;
  if (a.done[item >> 1] != 0) return;
  const int kitem = a.cross ? (item ^ 1) : item;
  const int nq = a.n[item], nk = a.n[kitem];
  if (q0 >= nq) return;

  __shared__ float Ks[32 * 65];
  __shared__ float Vs[32 * 64];

  const int t = threadIdx.x, lane = t & 63, wv = t >> 6, lx = lane & 31, half = lane >> 5;
  const int qrow = q0 + wv * 32 + lx;
  const bool qok = qrow < nq;

  float qreg[32];
  {
    const float* qp = a.q + (size_t)item * a.sq + (size_t)(qok ? qrow : 0) * a.ldq + head * 64 + half;
#pragma unroll
    for (int s = 0; s < 32; ++s) qreg[s] = qok ? qp[2 * s] * (a.scale * 1.44269504088896340736f) : 0.0f;
  }
  f32x16 oacc[2];
#pragma unroll
  for (int n = 0; n < 2; ++n)
#pragma unroll
    for (int r = 0; r < 16; ++r) oacc[n][r] = 0.0f;
  float m_run = -INFINITY, l_run = 0.0f;

  const float* kb = a.k + (size_t)kitem * a.sk + head * 64;
  const float* vb = a.v + (size_t)kitem * a.sv + head * 64;

  // software pipeline: tile kt+32 is fetched into registers while tile kt is being consumed
  float4 rk[2], rv[2];
  auto load_tile = [&](int kt) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int idx = t + 256 * i;
      const int key = idx >> 4, q4 = idx & 15;
      rk[i] = make_float4(0.f, 0.f, 0.f, 0.f);
      rv[i] = rk[i];
      if (kt + key < nk) {
        rk[i] = *(const float4*)(kb + (size_t)(kt + key) * a.ldk + q4 * 4);
        rv[i] = *(const float4*)(vb + (size_t)(kt + key) * a.ldv + q4 * 4);
      }
    }
  };
  load_tile(0);
  for (int kt = 0; kt < nk; kt += 32) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int idx = t + 256 * i;
      const int key = idx >> 4, q4 = idx & 15;
      float* d = &Ks[key * 65 + q4 * 4];
      d[0] = rk[i].x; d[1] = rk[i].y; d[2] = rk[i].z; d[3] = rk[i].w;
      *(float4*)&Vs[key * 64 + q4 * 4] = rv[i];
    }
    __syncthreads();
    if (kt + 32 < nk) load_tile(kt + 32);

    f32x16 sacc;
#pragma unroll
    for (int r = 0; r < 16; ++r) sacc[r] = 0.0f;
    const float* kp = &Ks[lx * 65 + half];
#pragma unroll
    for (int s = 0; s < 32; ++s) sacc = mfma32(kp[2 * s], qreg[s], sacc);

    // ---- online softmax in the log2 domain (Q was pre-scaled by scale*log2(e)) ----
    if (kt + 32 > nk) {  // ragged last tile only: mask the missing keys
#pragma unroll
      for (int r = 0; r < 16; ++r)
        if (kt + mfma_row(r, half) >= nk) sacc[r] = -INFINITY;
    }
    float tmax = fmaxf(fmaxf(sacc[0], sacc[1]), sacc[2]);
#pragma unroll
    for (int r = 3; r < 15; r += 2) tmax = fmaxf(fmaxf(tmax, sacc[r]), sacc[r + 1]);
    tmax = fmaxf(tmax, sacc[15]);
    tmax = fmaxf(tmax, __shfl_xor(tmax, 32));
    // Deferred rescale: the running reference m_run is only moved when the tile maximum exceeds it
    // by more than 2^RESCALE_LOG2 (probabilities then stay <= 2^RESCALE_LOG2, far from fp32
    // overflow); mathematically identical after the final division by l.
    constexpr float RESCALE_LOG2 = 8.0f;
    if (__any(tmax > m_run + RESCALE_LOG2)) {
      const float m_new = fmaxf(m_run, tmax);
      const float alpha = exp2f(m_run - m_new);  // first tile: exp2(-inf) = 0
      l_run *= alpha;
#pragma unroll
      for (int n = 0; n < 2; ++n)
#pragma unroll
        for (int r = 0; r < 16; ++r) oacc[n][r] *= alpha;
      m_run = m_new;
    }
    float p[16];
    float psum = 0.0f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      p[r] = __builtin_amdgcn_exp2f(sacc[r] - m_run);
      psum += p[r];
    }
    l_run += psum;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float* vp = &Vs[mfma_row(r, half) * 64 + lx];
      oacc[0] = mfma32(vp[0], p[r], oacc[0]);
      oacc[1] = mfma32(vp[32], p[r], oacc[1]);
    }
    __syncthreads();
  }

  const float l_tot = l_run + __shfl_xor(l_run, 32);
  if (qok) {
    float* op = a.o + (size_t)item * a.so + (size_t)qrow * a.ldo + head * 64;
    const float inv_ok = (nk > 0) ? 1.0f : 0.0f;
#pragma unroll
    for (int n = 0; n < 2; ++n)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        float4 o;
        if (inv_ok != 0.0f)
          o = make_float4(oacc[n][4 * g] / l_tot, oacc[n][4 * g + 1] / l_tot, oacc[n][4 * g + 2] / l_tot, oacc[n][4 * g + 3] / l_tot);
        else
          o = make_float4(0.f, 0.f, 0.f, 0.f);  // LGN:103-104: empty key set -> zeros
        *(float4*)(op + n * 32 + 8 * g + 4 * half) = o;
      }
  }
}
}  // namespace

int launch_lg_attention(const LgState& st, int cross, hipStream_t s, int kv_ready) {
  if (dim_precision_mode() != 0) return launch_lg_attention_x6(st, cross, s, kv_ready);
  AttnArgs a;
  const long long is = (long long)st.nmax * 768;
  if (!cross) {  // qkv = [q(256) | k(256) | v(256)] after the Wqkv row permutation done at load time
    a.q = st.qkv; a.k = st.qkv + 256; a.v = st.qkv + 512;
  } else {       // [qk(256) | v(256)] from the fused to_qk/to_v projection
    a.q = st.qkv; a.k = st.qkv; a.v = st.qkv + 256;
  }
  a.ldq = a.ldk = a.ldv = 768; a.sq = a.sk = a.sv = is;
  a.o = st.ctx; a.ldo = 256; a.so = (long long)st.nmax * 256;
  a.n = st.n_cur; a.done = st.done; a.cross = cross;
  a.scale = 0.125f;  // 64^-0.5 (LGN:123; cross: 64^-0.25 on each side, LGN:198) — a power of two, exact
  dim3 grid(cdiv(st.nmax, 128), 4, st.n_items);
  hipLaunchKernelGGL(attn_kernel, grid, dim3(256), 0, s, a);
  DIM_LAUNCH_CHECK();
  return 0;
}
