// LightGlue non-GEMM kernels for gfx950 (reference LGN = thirdparty/LightGlue/lightglue/lightglue.py).
// All of them are HBM/latency-bound element- or row-wise passes; one wave per
// row wherever a row reduction is needed (wave64 xor-shuffle reductions).
#include <math.h>

#include "lg_kernels.h"

namespace {

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }
// F.logsigmoid(x) = min(x,0) - log1p(exp(-|x|))
__device__ __forceinline__ float logsigmoidf_(float x) { return fminf(x, 0.0f) - log1pf(expf(-fabsf(x))); }

// ---------------------------------------------------------------------------
// init: gather features of (pair, side) from the feature table, normalise keypoints
// (LGN:25-34, size used as given — DIM passes (H, W), Q4), positional encoding
// (LGN:57-70), index / prune / done bookkeeping (LGN:480-488).
// grid: (ceil(nmax/8), items), block 256 = 8 points x 32 frequencies.
__global__ __launch_bounds__(256) void lg_init_kernel(LgState st, const float* __restrict__ kpts_tab,
                                                      const float* __restrict__ desc_tab, const int* __restrict__ n_tab,
                                                      const float* __restrict__ size_tab, const int* __restrict__ pair_idx,
                                                      int cap, int in_dim, const float* __restrict__ Wr, int copy_desc, unsigned* sat) {
  const int item = blockIdx.y;
  const int img = pair_idx ? pair_idx[item] : item;
  const int n = min(min(n_tab[img], st.nmax), cap);   // (a table of `cap` rows per image cannot hold more)
  const int j = threadIdx.x & 31, pt = blockIdx.x * 8 + (threadIdx.x >> 5);
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    st.n_cur[item] = n; st.n_new[item] = n; st.n_orig[item] = n;
    if ((item & 1) == 0) {
      const int img1 = pair_idx ? pair_idx[item + 1] : item + 1;
      const int n1 = min(min(n_tab[img1], st.nmax), cap);
      st.done[item >> 1] = (n == 0 || n1 == 0) ? -1 : 0;  // LGN:491-492 at i = 0 -> stop = 1
      st.cnt_lt[item >> 1] = 0;
    }
  }
  if (pt >= st.nmax) return;
  const size_t row = (size_t)item * st.nmax + pt;
  if (j == 0) { st.ind[row] = pt; st.prune[row] = 1; }
  if (pt >= n) return;
  const float sx = size_tab[img * 2], sy = size_tab[img * 2 + 1];
  const float scale = fmaxf(sx, sy) / 2.0f;
  const float kx = (kpts_tab[((size_t)img * cap + pt) * 2] - sx / 2.0f) / scale;
  const float ky = (kpts_tab[((size_t)img * cap + pt) * 2 + 1] - sy / 2.0f) / scale;
  const float pr = kx * Wr[j * 2] + ky * Wr[j * 2 + 1];
  st.enc[row * 64 + j] = cosf(pr);
  st.enc[row * 64 + 32 + j] = sinf(pr);
  if (copy_desc) {  // input_dim == 256: Identity input_proj (LGN:363-364)
    const float4* src = (const float4*)(desc_tab + ((size_t)img * cap + pt) * in_dim);
    float4* dst = (float4*)(st.desc + row * 256);
    const float4 a = src[j], b = src[j + 32];
    dst[j] = a;
    dst[j + 32] = b;
    // external input: a NaN descriptor must trip the guard too, and fmaxf drops NaN — add the NaN marker explicitly
    const float nanmark = (a.x + a.y + a.z + a.w + b.x + b.y + b.z + b.w) * 0.0f;   // 0 for finite inputs, NaN for NaN / Inf
    sat_report(sat, fmaxf(sat_track(sat_track(0.0f, a.x, a.y), a.z, a.w), sat_track(sat_track(0.0f, b.x, b.y), b.z, b.w)) + nanmark);
  }
}

// ---------------------------------------------------------------------------
// rotary embedding on q and k in place (LGN:41-54,155-156): pairs (t0,t1) ->
// (t0 c - t1 s, t1 c + t0 s).  block 256 = (q|k) x 4 heads x 32 pairs, one row.
__global__ __launch_bounds__(256) void lg_rotary_kernel(LgState st) {
  const int item = blockIdx.y;
  if (st.done[item >> 1] != 0) return;
  const int row = blockIdx.x;
  if (row >= st.n_cur[item]) return;
  const int t = threadIdx.x, i = t & 31, hq = t >> 5;  // hq: 0..3 q heads, 4..7 k heads
  const size_t r = (size_t)item * st.nmax + row;
  float* p = st.qkv + r * 768 + hq * 64 + 2 * i;
  const float c = st.enc[r * 64 + i], s = st.enc[r * 64 + 32 + i];
  const float t0 = p[0], t1 = p[1];
  p[0] = t0 * c + (-t1) * s;
  p[1] = t1 * c + t0 * s;
}

// ---------------------------------------------------------------------------
// LayerNorm(512, eps 1e-5, affine) + exact-erf GELU in place on hid (LGN:141-142).
// one wave per row, 8 values per lane.
__global__ __launch_bounds__(256) void lg_ln_gelu_kernel(LgState st, const float* __restrict__ gamma,
                                                         const float* __restrict__ beta) {
  const int item = blockIdx.y;
  if (st.done[item >> 1] != 0) return;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= st.n_cur[item]) return;
  float* p = st.hid + ((size_t)item * st.nmax + row) * 512;
  float x[8];
  const float4 a = *(const float4*)(p + lane * 4), b = *(const float4*)(p + 256 + lane * 4);
  x[0] = a.x; x[1] = a.y; x[2] = a.z; x[3] = a.w; x[4] = b.x; x[5] = b.y; x[6] = b.z; x[7] = b.w;
  float sum = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i) sum += x[i];
  const float mean = wave_sum(sum) / 512.0f;
  float sq = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i) { const float d = x[i] - mean; sq += d * d; }
  const float rstd = 1.0f / sqrtf(wave_sum(sq) / 512.0f + 1e-5f);
  const float4 g0 = *(const float4*)(gamma + lane * 4), g1 = *(const float4*)(gamma + 256 + lane * 4);
  const float4 b0 = *(const float4*)(beta + lane * 4), b1 = *(const float4*)(beta + 256 + lane * 4);
  const float gm[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w}, bt[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const float y = (x[i] - mean) * rstd * gm[i] + bt[i];
    x[i] = 0.5f * y * (1.0f + erf_1ulp(y * 0.70710678118654752440f));
  }
  *(float4*)(p + lane * 4) = make_float4(x[0], x[1], x[2], x[3]);
  *(float4*)(p + 256 + lane * 4) = make_float4(x[4], x[5], x[6], x[7]);
  float vmax = 0.0f;
#pragma unroll
  for (int i = 0; i < 8; i += 2) vmax = sat_track(vmax, x[i], x[i + 1]);
  sat_report(st.sat_ffn, vmax);
}

// ---------------------------------------------------------------------------
// token confidence (LGN:73-83) and matchability (LGN:277-278) per live point, plus
// the per-pair count of points with confidence < thr (LGN:600-603).
// One wave per point, CONF_ROWS points per wave one after the other (their descriptor loads go out together), one count per WORKGROUP: until round 5
// every low-confidence point did its own atomicAdd on the pair's counter — with weights that are not confident (every synthetic set, and the first
// layers of trained ones) that is 2 x n atomics on ONE address per pair and launch, which the memory system serialises: 0.99 ms per launch at
// 16 tile pairs x 4096 points (22 % of the GPU time of the config-5 benchmark, profiles/r05_config5_kernel_stats_before.csv), ~0.5 ms at 50 pairs x
// 2048.  Now 1 / (4 CONF_ROWS) of them.  Integer counts: the same decisions.
constexpr int CONF_ROWS = 8;
__global__ __launch_bounds__(256) void lg_confidence_kernel(LgState st, const float* __restrict__ w_tok,
                                                            const float* __restrict__ b_tok,
                                                            const float* __restrict__ w_match,
                                                            const float* __restrict__ b_match, float thr, int use_token) {
  __shared__ int blk_cnt[4];
  const int item = blockIdx.y;
  if (st.done[item >> 1] != 0) return;
  const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int row0 = (blockIdx.x * 4 + wv) * CONF_ROWS, n = st.n_cur[item];
  if (blockIdx.x * 4 * CONF_ROWS >= n) return;   // (block-uniform)
  const float4 wm = *(const float4*)(w_match + lane * 4);
  float4 wt = make_float4(0.f, 0.f, 0.f, 0.f);
  if (use_token) wt = *(const float4*)(w_tok + lane * 4);
  float4 d[CONF_ROWS];
#pragma unroll
  for (int i = 0; i < CONF_ROWS; ++i) {
    const size_t r = (size_t)item * st.nmax + min(row0 + i, n - 1);
    d[i] = *(const float4*)(st.desc + r * 256 + lane * 4);
  }
  int cnt = 0;
#pragma unroll
  for (int i = 0; i < CONF_ROWS; ++i) {
    const int row = row0 + i;
    if (row >= n) break;   // (wave-uniform)
    const size_t r = (size_t)item * st.nmax + row;
    // (wave_sum_dpp: DPP row reductions + four v_readlane instead of six ds_bpermute round trips per sum — 16 sums per wave here)
    const float zm = wave_sum_dpp(d[i].x * wm.x + d[i].y * wm.y + d[i].z * wm.z + d[i].w * wm.w) + b_match[0];
    float cf = 0.0f;
    if (use_token) cf = sigmoidf_(wave_sum_dpp(d[i].x * wt.x + d[i].y * wt.y + d[i].z * wt.z + d[i].w * wt.w) + b_tok[0]);
    if (lane == 0) {
      st.mtch[r] = sigmoidf_(zm);
      st.conf[r] = cf;
    }
    cnt += (use_token && cf < thr) ? 1 : 0;   // (cf is wave-uniform: wave_sum leaves the total in every lane)
  }
  if (lane == 0) blk_cnt[wv] = cnt;
  __syncthreads();
  if (threadIdx.x == 0) {
    const int c = (blk_cnt[0] + blk_cnt[1]) + (blk_cnt[2] + blk_cnt[3]);
    if (c) atomicAdd(&st.cnt_lt[item >> 1], c);
  }
}

// one thread per pair: early-stop decision (LGN:593-604) / last-layer close-out.
// mirror (nullptr, or page-locked host memory mapped into the device's address space; n_pairs <= 64 = one wave): the flags as they stand after this layer,
// then — behind a system-scope fence — the call's sequence number in mirror[seq_off]: the host (dim_lg_match, key 18) spins on that word two layers behind
// the device and stops enqueueing layers once every pair has left.  Two 4-byte stores over the host link instead of a copy command + an event per layer.
__global__ void lg_decide_kernel(LgState st, int layer, float depth_conf, int early, int last, int* mirror, int seq_off, int seq) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p < st.n_pairs) {
    if (st.done[p] == 0) {
      if (last) {
        st.done[p] = layer + 1;
      } else if (early) {
        const float num = (float)(st.n_orig[2 * p] + st.n_orig[2 * p + 1]);
        const float ratio = 1.0f - (float)st.cnt_lt[p] / num;
        if (ratio > depth_conf) st.done[p] = layer + 1;
      }
    }
    st.cnt_lt[p] = 0;
    if (mirror != nullptr) mirror[p] = st.done[p];
  }
  if (mirror != nullptr && blockIdx.x == 0) {
    __threadfence_system();
    if (threadIdx.x == 0) __atomic_store_n(mirror + seq_off, seq, __ATOMIC_RELEASE);
  }
}

// ---------------------------------------------------------------------------
// pruning (LGN:501-516,586-591): ordered compaction of the kept points.
// 1) scan: one block per item -> dest[j] (or -1), n_new
__global__ __launch_bounds__(1024) void lg_prune_scan_kernel(LgState st, float keep_thr, float thr, int use_token,
                                                             int pruning_min) {
  __shared__ int part[1024];
  const int item = blockIdx.x, t = threadIdx.x;
  if (st.done[item >> 1] != 0) return;
  const int n = st.n_cur[item];
  if (n <= pruning_min) {  // LGN:501,510: pruning only above pruning_min_kpts (-1 on the CPU path)
    if (t == 0) st.n_new[item] = n;
    return;
  }
  const size_t base = (size_t)item * st.nmax;
  const int per = (n + 1023) / 1024;
  const int j0 = t * per, j1 = min(j0 + per, n);
  int cnt = 0;
  for (int j = j0; j < j1; ++j) {
    bool keep = st.mtch[base + j] > keep_thr;
    if (use_token) keep = keep || (st.conf[base + j] <= thr);
    cnt += keep ? 1 : 0;
  }
  part[t] = cnt;
  __syncthreads();
  for (int off = 1; off < 1024; off <<= 1) {
    const int v = (t >= off) ? part[t - off] : 0;
    __syncthreads();
    part[t] += v;
    __syncthreads();
  }
  int run = part[t] - cnt;
  for (int j = j0; j < j1; ++j) {
    bool keep = st.mtch[base + j] > keep_thr;
    if (use_token) keep = keep || (st.conf[base + j] <= thr);
    st.dest[base + j] = keep ? run : -1;
    run += keep ? 1 : 0;
  }
  if (t == 1023) st.n_new[item] = part[1023];
}
// 2) gather kept rows into scratch (one wave per row), bump prune counters (LGN:509,516)
__global__ __launch_bounds__(256) void lg_prune_gather_kernel(LgState st, int pruning_min) {
  const int item = blockIdx.y;
  if (st.done[item >> 1] != 0) return;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (st.n_cur[item] <= pruning_min || row >= st.n_cur[item]) return;
  const size_t base = (size_t)item * st.nmax;
  // nothing pruned on this side in this layer (every set of weights that is not confident yet; n_cur is committed after the copy-back): the compaction
  // is the identity — only the counters move, the two 1-KB-per-point copies (through scratch and back: 2.5 % of the config-5 benchmark's GPU time) do not
  const bool identity = st.n_new[item] == st.n_cur[item];
  const int d = st.dest[base + row];
  if (d < 0) return;
  if (!identity) {
    *(float4*)(st.tdesc + (base + d) * 256 + lane * 4) = *(const float4*)(st.desc + (base + row) * 256 + lane * 4);
    st.tenc[(base + d) * 64 + lane] = st.enc[(base + row) * 64 + lane];
  }
  if (lane == 0) {
    const int orig = st.ind[base + row];
    if (!identity) st.tind[base + d] = orig;
    st.prune[base + orig] += 1;
  }
}
// 3) copy back
__global__ __launch_bounds__(256) void lg_prune_copyback_kernel(LgState st, int pruning_min) {
  const int item = blockIdx.y;
  if (st.done[item >> 1] != 0) return;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (st.n_cur[item] <= pruning_min || row >= st.n_new[item] || st.n_new[item] == st.n_cur[item]) return;   // (identity: see the gather)
  const size_t r = (size_t)item * st.nmax + row;
  *(float4*)(st.desc + r * 256 + lane * 4) = *(const float4*)(st.tdesc + r * 256 + lane * 4);
  st.enc[r * 64 + lane] = st.tenc[r * 64 + lane];
  if (lane == 0) st.ind[r] = st.tind[r];
}
// 4) commit counts; a side pruned to zero takes the "no keypoints" exit at the top of
//    the next iteration (LGN:491-492) -> stop = layer + 2
__global__ void lg_prune_commit_kernel(LgState st, int layer) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= st.n_pairs || st.done[p] != 0) return;
  const int a = st.n_new[2 * p], b = st.n_new[2 * p + 1];
  st.n_cur[2 * p] = a; st.n_cur[2 * p + 1] = b;
  if (a == 0 || b == 0) st.done[p] = -(layer + 2);
}

// ---------------------------------------------------------------------------
// assignment (LGN:246-275): per-row / per-column softmax statistics of sim and the
// log-sigmoid of the matchability logit, for pairs with done == tag.
// side 0 rows: wave per row (coalesced along the row).
__global__ __launch_bounds__(256) void lg_row_stats_kernel(LgState st, int tag, const float* __restrict__ w_match,
                                                           const float* __restrict__ b_match) {
  const int item = blockIdx.y, p = item >> 1, side = item & 1;
  const int dn = st.done[p];
  if (tag ? dn != tag : dn <= 0) return;   // tag 0: every stopped pair (its layer's weights: lg_api.hip, the deferred assignment)
  if (tag == 0) { w_match += (size_t)(dn - 1) * 256; b_match += dn - 1; }   // [layers][256] / [layers] tables
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int n = st.n_cur[item];
  if (row >= n) return;
  const size_t r = (size_t)item * st.nmax + row;
  // logsigmoid(matchability logit) for this point (both sides)
  const float4 d = *(const float4*)(st.desc + r * 256 + lane * 4);
  const float4 wm = *(const float4*)(w_match + lane * 4);
  const float z = wave_sum(d.x * wm.x + d.y * wm.y + d.z * wm.z + d.w * wm.w) + b_match[0];
  if (lane == 0) st.zls[r] = logsigmoidf_(z);
  if (side != 0) return;
  const int ncol = st.n_cur[item + 1];
  const float* srow = st.sim + ((size_t)p * st.nmax + row) * st.nmax;
  float m = -INFINITY;
  for (int j = lane; j < ncol; j += 64) m = fmaxf(m, srow[j]);
  m = wave_max(m);
  float s = 0.f;
  for (int j = lane; j < ncol; j += 64) s += expf(srow[j] - m);
  s = wave_sum(s);
  if (lane == 0) { st.rmax[r] = m; st.rlse[r] = logf(s); }
}
// side 1 columns: block = 64 columns x 16 row groups (1024 threads); each thread walks every 16th
// row of its column (coalesced across the 64 lanes of a wave), partials merged through LDS.
__global__ __launch_bounds__(1024) void lg_col_stats_kernel(LgState st, int tag) {
  __shared__ float red[16][64];
  const int p = blockIdx.y;
  const int dn = st.done[p];
  if (tag ? dn != tag : dn <= 0) return;   // tag 0: every stopped pair (its layer's weights: lg_api.hip, the deferred assignment)
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  const int col = blockIdx.x * 64 + tx;
  const int nrow = st.n_cur[2 * p], ncol = st.n_cur[2 * p + 1];
  if (blockIdx.x * 64 >= ncol) return;
  const bool ok = col < ncol;
  const float* sp = st.sim + (size_t)p * st.nmax * st.nmax + col;
  float m = -INFINITY;
  if (ok) for (int i = ty; i < nrow; i += 16) m = fmaxf(m, sp[(size_t)i * st.nmax]);
  red[ty][tx] = m;
  __syncthreads();
  m = red[0][tx];
#pragma unroll
  for (int g = 1; g < 16; ++g) m = fmaxf(m, red[g][tx]);
  __syncthreads();
  float s = 0.f;
  if (ok) for (int i = ty; i < nrow; i += 16) s += expf(sp[(size_t)i * st.nmax] - m);
  red[ty][tx] = s;
  __syncthreads();
  if (ty == 0 && ok) {
    s = 0.f;
#pragma unroll
    for (int g = 0; g < 16; ++g) s += red[g][tx];
    const size_t r = (size_t)(2 * p + 1) * st.nmax + col;
    st.rmax[r] = m; st.rlse[r] = logf(s);
  }
}

__device__ __forceinline__ float lg_score(float sim, float rm, float rl, float cm, float cl, float z0, float z1) {
  // scores0 + scores1 + certainties  (LGN:250-254)
  return (((sim - rm) - rl) + ((sim - cm) - cl)) + (z0 + z1);
}
// row argmax (first maximal index, Tensor.max on CPU) — wave per row; optionally dumps the dense matrix.
__global__ __launch_bounds__(256) void lg_row_argmax_kernel(LgState st, int tag, float* __restrict__ dense) {
  const int p = blockIdx.y;
  const int dn = st.done[p];
  if (tag ? dn != tag : dn <= 0) return;   // tag 0: every stopped pair (its layer's weights: lg_api.hip, the deferred assignment)
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int nrow = st.n_cur[2 * p], ncol = st.n_cur[2 * p + 1];
  if (row >= nrow) return;
  const size_t r0 = (size_t)(2 * p) * st.nmax + row, c0 = (size_t)(2 * p + 1) * st.nmax;
  const float rm = st.rmax[r0], rl = st.rlse[r0], z0 = st.zls[r0];
  const float* srow = st.sim + ((size_t)p * st.nmax + row) * st.nmax;
  float best = -INFINITY;
  int bi = 0x7fffffff;
  for (int j = lane; j < ncol; j += 64) {
    const float v = lg_score(srow[j], rm, rl, st.rmax[c0 + j], st.rlse[c0 + j], z0, st.zls[c0 + j]);
    if (dense) dense[((size_t)p * (st.nmax + 1) + row) * (st.nmax + 1) + j] = v;
    if (v > best) { best = v; bi = j; }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float ob = __shfl_xor(best, o);
    const int oi = __shfl_xor(bi, o);
    if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; }
  }
  if (lane == 0) { st.best[r0] = best; st.arg[r0] = bi; }
}
__global__ __launch_bounds__(1024) void lg_col_argmax_kernel(LgState st, int tag) {
  __shared__ float redv[16][64];
  __shared__ int redi[16][64];
  const int p = blockIdx.y;
  const int dn = st.done[p];
  if (tag ? dn != tag : dn <= 0) return;   // tag 0: every stopped pair (its layer's weights: lg_api.hip, the deferred assignment)
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  const int col = blockIdx.x * 64 + tx;
  const int nrow = st.n_cur[2 * p], ncol = st.n_cur[2 * p + 1];
  if (blockIdx.x * 64 >= ncol) return;
  const bool ok = col < ncol;
  const size_t r0 = (size_t)(2 * p) * st.nmax, c = (size_t)(2 * p + 1) * st.nmax + (ok ? col : 0);
  const float cm = st.rmax[c], cl = st.rlse[c], z1 = st.zls[c];
  const float* sp = st.sim + (size_t)p * st.nmax * st.nmax + col;
  float best = -INFINITY;
  int bi = 0x7fffffff;
  if (ok)
    for (int i = ty; i < nrow; i += 16) {
      const float v = lg_score(sp[(size_t)i * st.nmax], st.rmax[r0 + i], st.rlse[r0 + i], cm, cl, st.zls[r0 + i], z1);
      if (v > best) { best = v; bi = i; }
    }
  redv[ty][tx] = best; redi[ty][tx] = bi;
  __syncthreads();
  if (ty == 0 && ok) {
#pragma unroll
    for (int g = 1; g < 16; ++g) {  // first maximal row index wins ties (Tensor.max on CPU)
      const float ov = redv[g][tx];
      const int oi = redi[g][tx];
      if (ov > best || (ov == best && oi < bi)) { best = ov; bi = oi; }
    }
    st.best[c] = best; st.arg[c] = bi;
  }
}

// ---------------------------------------------------------------------------
// filter_matches (LGN:281-297) + compact match list + scatter back to the
// un-pruned index space (LGN:543-566).  One block per pair.
__global__ __launch_bounds__(1024) void lg_finalize_kernel(LgState st, int n_layers, int prune_enabled, float filter_thr,
                                                           int out_cap, long long* __restrict__ matches,
                                                           float* __restrict__ mscores, int* __restrict__ n_matches,
                                                           int* __restrict__ mfull, float* __restrict__ msfull,
                                                           int* __restrict__ stop, int* __restrict__ prune_out) {
  __shared__ int part[1024];
  const int p = blockIdx.x, t = threadIdx.x;
  const int dn = st.done[p];
  const size_t b0 = (size_t)(2 * p) * st.nmax, b1 = b0 + st.nmax;
  const int m_orig = st.n_orig[2 * p], n_orig = st.n_orig[2 * p + 1];
  for (int i = t; i < st.nmax; i += 1024) {
    mfull[b0 + i] = -1; msfull[b0 + i] = 0.0f;
    mfull[b1 + i] = -1; msfull[b1 + i] = 0.0f;
    prune_out[b0 + i] = (i < m_orig) ? (prune_enabled ? st.prune[b0 + i] : n_layers) : 0;
    prune_out[b1 + i] = (i < n_orig) ? (prune_enabled ? st.prune[b1 + i] : n_layers) : 0;
  }
  if (t == 0) stop[p] = dn > 0 ? dn : -dn;
  if (dn <= 0) {  // "no keypoints" exit: no matches
    if (t == 0) n_matches[p] = 0;
    return;
  }
  __syncthreads();
  const int m = st.n_cur[2 * p], n = st.n_cur[2 * p + 1];
  // side 0
  const int per = (m + 1023) / 1024;
  const int i0 = t * per, i1 = min(i0 + per, m);
  int cnt = 0;
  for (int i = i0; i < i1; ++i) {
    const int j = st.arg[b0 + i];
    const bool mutual = st.arg[b1 + j] == i;
    const float ms = mutual ? expf(st.best[b0 + i]) : 0.0f;
    const bool valid = mutual && ms > filter_thr;
    const int oi = st.ind[b0 + i];
    msfull[b0 + oi] = ms;
    mfull[b0 + oi] = valid ? st.ind[b1 + j] : -1;
    cnt += valid ? 1 : 0;
  }
  part[t] = cnt;
  __syncthreads();
  for (int off = 1; off < 1024; off <<= 1) {
    const int v = (t >= off) ? part[t - off] : 0;
    __syncthreads();
    part[t] += v;
    __syncthreads();
  }
  int run = part[t] - cnt;
  for (int i = i0; i < i1; ++i) {
    const int j = st.arg[b0 + i];
    const bool mutual = st.arg[b1 + j] == i;
    const float ms = mutual ? expf(st.best[b0 + i]) : 0.0f;
    if (mutual && ms > filter_thr) {
      if (run < out_cap) {
        matches[((size_t)p * out_cap + run) * 2] = st.ind[b0 + i];
        matches[((size_t)p * out_cap + run) * 2 + 1] = st.ind[b1 + j];
        mscores[(size_t)p * out_cap + run] = ms;
      }
      ++run;
    }
  }
  if (t == 1023) n_matches[p] = min(part[1023], out_cap);
  // side 1 (LGN:292,295)
  for (int j = t; j < n; j += 1024) {
    const int i = st.arg[b1 + j];
    const bool mutual1 = st.arg[b0 + i] == j;
    const bool mutual0 = mutual1;  // arg0[i] == j and arg1[j] == i are the same condition seen from j
    const float ms0 = mutual0 ? expf(st.best[b0 + i]) : 0.0f;
    const bool valid1 = mutual1 && (ms0 > filter_thr);
    const int oj = st.ind[b1 + j];
    msfull[b1 + oj] = mutual1 ? ms0 : 0.0f;
    mfull[b1 + oj] = valid1 ? st.ind[b0 + i] : -1;
  }
}

}  // namespace

// ---------------------------------------------------------------------------
int launch_lg_init(const LgState& st, const float* kpts_tab, const float* desc_tab, const int* n_tab, const float* size_tab,
                   const int* pair_idx, int cap, int in_dim, const float* Wr, int copy_desc, unsigned* sat, hipStream_t s) {
  hipLaunchKernelGGL(lg_init_kernel, dim3(cdiv(st.nmax, 8), st.n_items), dim3(256), 0, s, st, kpts_tab, desc_tab, n_tab,
                     size_tab, pair_idx, cap, in_dim, Wr, copy_desc, sat);
  DIM_LAUNCH_CHECK();
  return 0;
}
int launch_lg_rotary(const LgState& st, hipStream_t s) {
  hipLaunchKernelGGL(lg_rotary_kernel, dim3(st.nmax, st.n_items), dim3(256), 0, s, st);
  DIM_LAUNCH_CHECK();
  return 0;
}
int launch_lg_ln_gelu(const LgState& st, const float* gamma, const float* beta, hipStream_t s) {
  hipLaunchKernelGGL(lg_ln_gelu_kernel, dim3(cdiv(st.nmax, 4), st.n_items), dim3(256), 0, s, st, gamma, beta);
  DIM_LAUNCH_CHECK();
  return 0;
}
int launch_lg_confidence(const LgState& st, const float* w_tok, const float* b_tok, const float* w_match,
                         const float* b_match, float thr, int use_token, hipStream_t s) {
  hipLaunchKernelGGL(lg_confidence_kernel, dim3(cdiv(st.nmax, 4 * CONF_ROWS), st.n_items), dim3(256), 0, s, st, w_tok, b_tok, w_match,
                     b_match, thr, use_token);
  DIM_LAUNCH_CHECK();
  return 0;
}
int launch_lg_decide(const LgState& st, int layer, float depth_conf, int early, int last, hipStream_t s, int* mirror, int seq_off, int seq) {
  DIM_REQUIRE(mirror == nullptr || st.n_pairs <= 64, "lg_decide: the host mirror exists for at most 64 pairs");
  hipLaunchKernelGGL(lg_decide_kernel, dim3(cdiv(st.n_pairs, 64)), dim3(64), 0, s, st, layer, depth_conf, early, last, mirror, seq_off, seq);
  DIM_LAUNCH_CHECK();
  return 0;
}
int launch_lg_prune(const LgState& st, int layer, double width_conf, float thr, int use_token, int pruning_min,
                    hipStream_t s) {
  const float keep_thr = (float)(1.0 - width_conf);  // python double, cast to the tensor dtype at the compare (LGN:588)
  hipLaunchKernelGGL(lg_prune_scan_kernel, dim3(st.n_items), dim3(1024), 0, s, st, keep_thr, thr, use_token, pruning_min);
  hipLaunchKernelGGL(lg_prune_gather_kernel, dim3(cdiv(st.nmax, 4), st.n_items), dim3(256), 0, s, st, pruning_min);
  hipLaunchKernelGGL(lg_prune_copyback_kernel, dim3(cdiv(st.nmax, 4), st.n_items), dim3(256), 0, s, st, pruning_min);
  hipLaunchKernelGGL(lg_prune_commit_kernel, dim3(cdiv(st.n_pairs, 64)), dim3(64), 0, s, st, layer);
  DIM_LAUNCH_CHECK();
  return 0;
}
// ---- the same four passes for the common shape (at most 2048 keypoints; the row stride nmax is always a multiple of 4, lg_api.hip): every element of the
// similarity is read ONCE per pass as part of a 16-byte load with all of a thread's loads in flight together (the generic kernels
// above walk a row twice in 4-byte steps, one dependent load per iteration: 0.38 ms per pass and 50 pairs, 2.2 TB/s).  A row lives in
// 8 float4 registers per lane; a column group of 4 columns per thread keeps online (max, sum) pairs.  exp(x) for x <= 0 is the
// 6-instruction exp_le0 (dim_common.h, ~1.5 ulp). ----
constexpr int ROW_CH_MAX = 16;   // float4 chunks per lane: 64 lanes x 4 x 16 = 4096 columns (8 for rows of up to 2048)
__device__ __forceinline__ float exp_le0_z(float d) { return d > -INFINITY ? exp_le0(d) : 0.0f; }   // exp(d), d <= 0, with exp(-inf) = 0 (exp_le0 itself returns NaN there)
template <int ROW_CH>   // float4 chunks per lane: 8 = rows of up to 2048 live columns, 16 = up to 4096 (round 6: a 2100-keypoint pair fell to the generic kernels)
__global__ __launch_bounds__(256) void lg_row_stats4_kernel(LgState st, int tag, const float* __restrict__ w_match,
                                                            const float* __restrict__ b_match) {
  const int item = blockIdx.y, p = item >> 1, side = item & 1;
  const int dn = st.done[p];
  if (tag ? dn != tag : dn <= 0) return;   // tag 0: every stopped pair (its layer's weights: lg_api.hip, the deferred assignment)
  if (tag == 0) { w_match += (size_t)(dn - 1) * 256; b_match += dn - 1; }   // [layers][256] / [layers] tables
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int n = st.n_cur[item];
  if (row >= n) return;
  const size_t r = (size_t)item * st.nmax + row;
  const float4 d = *(const float4*)(st.desc + r * 256 + lane * 4);
  const float4 wm = *(const float4*)(w_match + lane * 4);
  const float z = wave_sum(d.x * wm.x + d.y * wm.y + d.z * wm.z + d.w * wm.w) + b_match[0];
  if (lane == 0) st.zls[r] = logsigmoidf_(z);
  if (side != 0) return;
  const int ncol = st.n_cur[item + 1];
  const float4* srow = (const float4*)(st.sim + ((size_t)p * st.nmax + row) * st.nmax);
  float4 x[ROW_CH];
#pragma unroll
  for (int c = 0; c < ROW_CH; ++c) {
    const int j = (c * 64 + lane) * 4;
    x[c] = j < ncol ? srow[c * 64 + lane] : make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);   // (the row's padding is readable)
    if (j + 1 >= ncol) x[c].y = -INFINITY;
    if (j + 2 >= ncol) x[c].z = -INFINITY;
    if (j + 3 >= ncol) x[c].w = -INFINITY;
  }
  float m = -INFINITY;
#pragma unroll
  for (int c = 0; c < ROW_CH; ++c) m = fmaxf(m, fmaxf(fmaxf(x[c].x, x[c].y), fmaxf(x[c].z, x[c].w)));
  m = wave_max(m);
  float s = 0.f;
  if (m > -INFINITY) {
#pragma unroll
    for (int c = 0; c < ROW_CH; ++c) s += (exp_le0_z(x[c].x - m) + exp_le0_z(x[c].y - m)) + (exp_le0_z(x[c].z - m) + exp_le0_z(x[c].w - m));
  }
  s = wave_sum(s);
  if (lane == 0) { st.rmax[r] = m; st.rlse[r] = logf(s); }
}
// thread = 4 adjacent columns x every COL_RG-th row; workgroup = 4 COL_CW columns.  COL_CW 16 (round 5; 64 before): 32 instead of 8 workgroups per pair
// at 2048 columns, 32 instead of 128 dependent loop steps per thread — ONE pair per call (the plugin hooks) ran these two passes on 8 of the chip's 256
// CUs (65 + 39 us per pair; the row passes, one wave per row, take 7 + 7).  A row segment of a workgroup is still 256 contiguous bytes.  The SAME shape
// at every batch size: a pair's result does not depend on how many pairs share the launch.
constexpr int COL_CW = 16, COL_RG = 1024 / COL_CW;
__global__ __launch_bounds__(1024) void lg_col_stats4_kernel(LgState st, int tag) {
  __shared__ float4 redm[COL_RG][COL_CW], reds[COL_RG][COL_CW];
  const int p = blockIdx.y;
  const int dn = st.done[p];
  if (tag ? dn != tag : dn <= 0) return;   // tag 0: every stopped pair (its layer's weights: lg_api.hip, the deferred assignment)
  const int tx = threadIdx.x % COL_CW, ty = threadIdx.x / COL_CW;
  const int col = blockIdx.x * (4 * COL_CW) + tx * 4;
  const int nrow = st.n_cur[2 * p], ncol = st.n_cur[2 * p + 1];
  if (blockIdx.x * (4 * COL_CW) >= ncol) return;
  const bool ok = col < ncol;
  const float* sp = st.sim + (size_t)p * st.nmax * st.nmax + col;
  float m[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY}, sm[4] = {0.f, 0.f, 0.f, 0.f};
  if (ok) {
#pragma unroll 4
    for (int i = ty; i < nrow; i += COL_RG) {
      const float4 v = *(const float4*)(sp + (size_t)i * st.nmax);
      const float xv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
      for (int e = 0; e < 4; ++e) {   // online: rescale the running sum when the maximum moves
        const float mn = fmaxf(m[e], xv[e]);
        sm[e] = sm[e] * exp_le0_z(m[e] - mn) + exp_le0(xv[e] - mn);   // first element: 0 * exp(-inf) = 0
        m[e] = mn;
      }
    }
  }
  redm[ty][tx] = make_float4(m[0], m[1], m[2], m[3]); reds[ty][tx] = make_float4(sm[0], sm[1], sm[2], sm[3]);
  __syncthreads();
  if (ty < 4 && ok) {   // thread (tx, e = ty): merges the COL_RG row groups of column col + e
    const int e = ty;
    if (col + e < ncol) {
      float M = -INFINITY;
#pragma unroll 16
      for (int g = 0; g < COL_RG; ++g) M = fmaxf(M, ((const float*)&redm[g][tx])[e]);
      float S = 0.f;
#pragma unroll 16
      for (int g = 0; g < COL_RG; ++g) {
        const float mg = ((const float*)&redm[g][tx])[e];
        if (mg > -INFINITY) S += ((const float*)&reds[g][tx])[e] * exp_le0(mg - M);
      }
      const size_t r = (size_t)(2 * p + 1) * st.nmax + col + e;
      st.rmax[r] = M; st.rlse[r] = logf(S);
    }
  }
}
template <int ROW_CH>
__global__ __launch_bounds__(256) void lg_row_argmax4_kernel(LgState st, int tag, float* __restrict__ dense) {
  const int p = blockIdx.y;
  const int dn = st.done[p];
  if (tag ? dn != tag : dn <= 0) return;   // tag 0: every stopped pair (its layer's weights: lg_api.hip, the deferred assignment)
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int nrow = st.n_cur[2 * p], ncol = st.n_cur[2 * p + 1];
  if (row >= nrow) return;
  const size_t r0 = (size_t)(2 * p) * st.nmax + row, c0 = (size_t)(2 * p + 1) * st.nmax;
  const float rm = st.rmax[r0], rl = st.rlse[r0], z0 = st.zls[r0];
  const float4* srow = (const float4*)(st.sim + ((size_t)p * st.nmax + row) * st.nmax);
  const float4 *cm4 = (const float4*)(st.rmax + c0), *cl4 = (const float4*)(st.rlse + c0), *cz4 = (const float4*)(st.zls + c0);
  float best = -INFINITY;
  int bi = 0x7fffffff;
#pragma unroll
  for (int c = 0; c < ROW_CH; ++c) {
    const int j = (c * 64 + lane) * 4;
    if (j < ncol) {
      const float4 x = srow[c * 64 + lane], cm = cm4[c * 64 + lane], cl = cl4[c * 64 + lane], cz = cz4[c * 64 + lane];
      const float xv[4] = {x.x, x.y, x.z, x.w}, cmv[4] = {cm.x, cm.y, cm.z, cm.w}, clv[4] = {cl.x, cl.y, cl.z, cl.w}, czv[4] = {cz.x, cz.y, cz.z, cz.w};
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (j + e < ncol) {
          const float v = lg_score(xv[e], rm, rl, cmv[e], clv[e], z0, czv[e]);
          if (dense) dense[((size_t)p * (st.nmax + 1) + row) * (st.nmax + 1) + j + e] = v;
          if (v > best) { best = v; bi = j + e; }   // a lane walks its columns in ascending order: the first maximum stays
        }
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float ob = __shfl_xor(best, o);
    const int oi = __shfl_xor(bi, o);
    if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; }
  }
  if (lane == 0) { st.best[r0] = best; st.arg[r0] = bi; }
}
__global__ __launch_bounds__(1024) void lg_col_argmax4_kernel(LgState st, int tag) {
  __shared__ float4 redv[COL_RG][COL_CW];
  __shared__ int redi[COL_RG][COL_CW][4];
  const int p = blockIdx.y;
  const int dn = st.done[p];
  if (tag ? dn != tag : dn <= 0) return;   // tag 0: every stopped pair (its layer's weights: lg_api.hip, the deferred assignment)
  const int tx = threadIdx.x % COL_CW, ty = threadIdx.x / COL_CW;
  const int col = blockIdx.x * (4 * COL_CW) + tx * 4;
  const int nrow = st.n_cur[2 * p], ncol = st.n_cur[2 * p + 1];
  if (blockIdx.x * (4 * COL_CW) >= ncol) return;
  const bool ok = col < ncol;
  const size_t r0 = (size_t)(2 * p) * st.nmax, c = (size_t)(2 * p + 1) * st.nmax + (ok ? col : 0);
  const float4 cm = *(const float4*)(st.rmax + c), cl = *(const float4*)(st.rlse + c), z1 = *(const float4*)(st.zls + c);
  const float cmv[4] = {cm.x, cm.y, cm.z, cm.w}, clv[4] = {cl.x, cl.y, cl.z, cl.w}, z1v[4] = {z1.x, z1.y, z1.z, z1.w};
  const float* sp = st.sim + (size_t)p * st.nmax * st.nmax + col;
  float best[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
  int bi[4] = {0x7fffffff, 0x7fffffff, 0x7fffffff, 0x7fffffff};
  if (ok) {
#pragma unroll 4
    for (int i = ty; i < nrow; i += COL_RG) {
      const float4 x = *(const float4*)(sp + (size_t)i * st.nmax);
      const float rm = st.rmax[r0 + i], rl = st.rlse[r0 + i], z0 = st.zls[r0 + i];
      const float xv[4] = {x.x, x.y, x.z, x.w};
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float v = lg_score(xv[e], rm, rl, cmv[e], clv[e], z0, z1v[e]);
        if (v > best[e]) { best[e] = v; bi[e] = i; }
      }
    }
  }
  redv[ty][tx] = make_float4(best[0], best[1], best[2], best[3]);
#pragma unroll
  for (int e = 0; e < 4; ++e) redi[ty][tx][e] = bi[e];
  __syncthreads();
  if (ty < 4 && ok && col + ty < ncol) {
    const int e = ty;
    float b = -INFINITY;
    int ix = 0x7fffffff;
#pragma unroll 16
    for (int g = 0; g < COL_RG; ++g) {  // first maximal row index wins ties (Tensor.max on CPU)
      const float ov = ((const float*)&redv[g][tx])[e];
      const int oi = redi[g][tx][e];
      if (ov > b || (ov == b && oi < ix)) { b = ov; ix = oi; }
    }
    st.best[c + e] = b; st.arg[c + e] = ix;
  }
}
// (the fast kernels hold a row's live columns in registers: what must fit is the live count's bound nsel, the row stride nmax only has to be a multiple of 4)
static bool assign_fast_shape(const LgState& st) { return (st.nmax & 3) == 0 && (st.nsel > 0 ? st.nsel : st.nmax) <= 256 * ROW_CH_MAX; }
static bool assign_rows_2048(const LgState& st) { return (st.nsel > 0 ? st.nsel : st.nmax) <= 2048; }

int launch_lg_assign_stats(const LgState& st, int tag, const float* w_match, const float* b_match, hipStream_t s) {
  if (assign_fast_shape(st)) {
    if (assign_rows_2048(st)) hipLaunchKernelGGL(HIP_KERNEL_NAME(lg_row_stats4_kernel<8>), dim3(cdiv(st.nmax, 4), st.n_items), dim3(256), 0, s, st, tag, w_match, b_match);
    else hipLaunchKernelGGL(HIP_KERNEL_NAME(lg_row_stats4_kernel<16>), dim3(cdiv(st.nmax, 4), st.n_items), dim3(256), 0, s, st, tag, w_match, b_match);
    hipLaunchKernelGGL(lg_col_stats4_kernel, dim3(cdiv(st.nmax, 4 * COL_CW), st.n_pairs), dim3(1024), 0, s, st, tag);
    DIM_LAUNCH_CHECK();
    return 0;
  }
  hipLaunchKernelGGL(lg_row_stats_kernel, dim3(cdiv(st.nmax, 4), st.n_items), dim3(256), 0, s, st, tag, w_match, b_match);
  hipLaunchKernelGGL(lg_col_stats_kernel, dim3(cdiv(st.nmax, 64), st.n_pairs), dim3(1024), 0, s, st, tag);
  DIM_LAUNCH_CHECK();
  return 0;
}
int launch_lg_assign_argmax(const LgState& st, int tag, float* dense_scores, hipStream_t s) {
  if (assign_fast_shape(st)) {
    if (assign_rows_2048(st)) hipLaunchKernelGGL(HIP_KERNEL_NAME(lg_row_argmax4_kernel<8>), dim3(cdiv(st.nmax, 4), st.n_pairs), dim3(256), 0, s, st, tag, dense_scores);
    else hipLaunchKernelGGL(HIP_KERNEL_NAME(lg_row_argmax4_kernel<16>), dim3(cdiv(st.nmax, 4), st.n_pairs), dim3(256), 0, s, st, tag, dense_scores);
    hipLaunchKernelGGL(lg_col_argmax4_kernel, dim3(cdiv(st.nmax, 4 * COL_CW), st.n_pairs), dim3(1024), 0, s, st, tag);
    DIM_LAUNCH_CHECK();
    return 0;
  }
  hipLaunchKernelGGL(lg_row_argmax_kernel, dim3(cdiv(st.nmax, 4), st.n_pairs), dim3(256), 0, s, st, tag, dense_scores);
  hipLaunchKernelGGL(lg_col_argmax_kernel, dim3(cdiv(st.nmax, 64), st.n_pairs), dim3(1024), 0, s, st, tag);
  DIM_LAUNCH_CHECK();
  return 0;
}
int launch_lg_finalize(const LgState& st, int n_layers, int prune_enabled, float filter_thr, int out_cap,
                       long long* matches, float* mscores, int* n_matches, int* mfull, float* msfull, int* stop,
                       int* prune_out, hipStream_t s) {
  hipLaunchKernelGGL(lg_finalize_kernel, dim3(st.n_pairs), dim3(1024), 0, s, st, n_layers, prune_enabled, filter_thr,
                     out_cap, matches, mscores, n_matches, mfull, msfull, stop, prune_out);
  DIM_LAUNCH_CHECK();
  return 0;
}
