// dim_lg_* : resident LightGlue matcher (C ABI in include/dim_hip.h).
// Replaces LightGlue.__init__/forward (LGN:300-610) as driven by
// LightGlueMatcher._match_pairs (matchers/lightglue.py:102-125), batched over
// pairs with device-side early stop / pruning (no host read-back anywhere).
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <chrono>
#include <thread>
#include <vector>

#include "../../include/dim_hip.h"
#include "lg_kernels.h"

namespace {
struct LayerW {
  float *qkv_w, *qkv_b, *out_w, *out_b, *sffn0_w, *sffn0_b, *sln_w, *sln_b, *sffn3_w, *sffn3_b;
  float *cqkv_w, *cqkv_b, *cout_w, *cout_b, *cffn0_w, *cffn0_b, *cln_w, *cln_b, *cffn3_w, *cffn3_b;
  float *match_w, *match_b, *proj_w, *proj_b, *tok_w, *tok_b;
  // pre-split copies of the GEMM operands for the split modes 1 (bf16x6) and 2 (fp16x3), gemm_x6.hip; index = mode
  SplitWeights qkv_x[3], out_x[3], sffn0_x[3], sffn3_x[3], cqkv_x[3], cout_x[3], cffn0_x[3], cffn3_x[3], proj_x[3];
  // out_proj folded into ffn.0 (split modes): [desc | ctx] * [W1a ; Wout*W1b] + (b1 + bout*W1b)
  SplitWeights sffn0f_x[3], cffn0f_x[3];
  SplitWeights sffn3p_x[3], cffn3p_x[3];   // ffn.3 with the k permutation of the fused feed-forward kernel (gemm_x6.hip, KV == 4)
  float *sffn0f_b, *cffn0f_b;
};
}  // namespace

constexpr int FOLLOW_MAX_PAIRS = 2;   // handles created for at most this many pairs may follow the stop flags on the host
namespace {
// dim_lg_stage_features: the arrays of a pair exactly as features.h5 holds them -> the fp32 (N, D) feature table of dim_lg_match.  One workgroup per
// (32 keypoints, image).  (D, N) descriptors (extractors/superpoint.py:121-127 stores them so) are read along N — 32 consecutive keypoints of one dim =
// one 64- / 128-byte run — and turned through a 32 x 33 LDS tile; fp16 -> fp32 is exact; rows past the live count are zeroed like the host path did.
struct StageArgs {
  const void* kp[2]; const void* ds[2];
  int n[2], kp_f16[2], ds_f16[2], ds_dn[2];
  int cap, D;
  float* kt; float* dt;
};
__device__ __forceinline__ float stage_ld(const void* p, size_t i, int f16) { return f16 ? (float)((const _Float16*)p)[i] : ((const float*)p)[i]; }
__global__ __launch_bounds__(256) void lg_stage_kernel(StageArgs a) {
  __shared__ float tile[32][33];
  const int im = blockIdx.y, r0 = blockIdx.x * 32, t = threadIdx.x, n = a.n[im];
  if (t < 64) {
    const int row = r0 + (t >> 1);
    if (row < a.cap) a.kt[((size_t)im * a.cap + row) * 2 + (t & 1)] = row < n ? stage_ld(a.kp[im], (size_t)row * 2 + (t & 1), a.kp_f16[im]) : 0.0f;
  }
  float* const out = a.dt + (size_t)im * a.cap * a.D;
  const int f16 = a.ds_f16[im];
  if (!a.ds_dn[im]) {   // (N, D): thread -> (row t >> 3, four consecutive dims)
    const int row = r0 + (t >> 3);
    if (row >= a.cap) return;
    for (int d0 = (t & 7) * 4; d0 < a.D; d0 += 32) {
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (row < n) {
        const size_t i = (size_t)row * a.D + d0;
        v = make_float4(stage_ld(a.ds[im], i, f16), stage_ld(a.ds[im], i + 1, f16), stage_ld(a.ds[im], i + 2, f16), stage_ld(a.ds[im], i + 3, f16));
      }
      *(float4*)(out + (size_t)row * a.D + d0) = v;
    }
    return;
  }
  for (int d0 = 0; d0 < a.D; d0 += 32) {   // (D, N): read [dim t >> 3][keypoints 4 (t & 7) ..], write [keypoint t >> 3][dims 4 (t & 7) ..]
    const int dd = t >> 3;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int rr = (t & 7) * 4 + j;
      tile[dd][rr] = (r0 + rr < n) ? stage_ld(a.ds[im], (size_t)(d0 + dd) * n + r0 + rr, f16) : 0.0f;
    }
    __syncthreads();
    const int row = r0 + (t >> 3), c = (t & 7) * 4;
    if (row < a.cap) *(float4*)(out + (size_t)row * a.D + d0 + c) = make_float4(tile[c][t >> 3], tile[c + 1][t >> 3], tile[c + 2][t >> 3], tile[c + 3][t >> 3]);
    __syncthreads();
  }
}
}  // namespace

struct dim_lg {
  DimHandleBase base;   // first member: dim_handle_tune_set
  dim_lg_config cfg;
  int n_layers, input_dim, max_pairs, nmax;
  std::vector<LayerW> L;
  std::vector<float> thr;
  float *inproj_w, *inproj_b, *Wr;
  // the deferred assignment of adaptive depth (dim_lg_match): per-layer tables the kernels index with a pair's stop layer
  float *match_w_all = nullptr, *match_b_all = nullptr;   // [layers][256], [layers]
  GemmLayerTab* proj_tab[3] = {nullptr, nullptr, nullptr};   // [mode][layers]: final_proj's split weights / inverse scales / bias
  // adaptive depth at one or two pairs per call: the host follows the stop flags two layers behind the device (dim_lg_match)
  int* done_host = nullptr;            // page-locked, device-mapped: [layers][max_pairs flags | sequence word] written by lg_decide_kernel
  int call_seq = 0;
  LgState st;
  std::vector<void*> allocs;
};

namespace {
template <typename T>
int dev_alloc(dim_lg* h, T** p, size_t count) {
  void* q = nullptr;
  hipError_t e = hipMalloc(&q, count * sizeof(T) + 256);
  if (e != hipSuccess) {
    dim_set_error("hipMalloc of %zu bytes failed: out of memory (%s)", count * sizeof(T), hipGetErrorString(e));
    return -1;
  }
  h->allocs.push_back(q);
  *p = (T*)q;
  return 0;
}
int upload(dim_lg* h, float** dst, const std::vector<float>& v) {
  if (!dim_all_finite(v.data(), v.size())) { dim_set_error("non-finite value in the weights"); return -1; }
  if (dev_alloc(h, dst, v.size()) != 0) return -1;
  if (hipMemcpy(*dst, v.data(), v.size() * sizeof(float), hipMemcpyHostToDevice) != hipSuccess) {
    dim_set_error("weight upload failed");
    return -1;
  }
  return 0;
}
// [K][N] fp32 GEMM operand -> device split planes for both split modes
int upload_x3(dim_lg* h, SplitWeights* dst, const std::vector<float>& w_kn, int K, int N, int kperm = 0) {
  const int n_pad = (N + 127) / 128 * 128;
  for (int mode = 1; mode <= 2; ++mode) {
    std::vector<unsigned short> host(gemm_split_weight_elems(K, n_pad, mode));
    split_weights(w_kn.data(), K, N, n_pad, mode, host.data(), &dst[mode], kperm);
    unsigned short* d = nullptr;
    if (dev_alloc(h, &d, host.size()) != 0) return -1;
    if (hipMemcpy(d, host.data(), host.size() * 2, hipMemcpyHostToDevice) != hipSuccess) {
      dim_set_error("weight upload failed");
      return -1;
    }
    dst[mode].dev = d; dst[mode].mode = mode; dst[mode].n_pad = n_pad;
  }
  return 0;
}
// ffn.0([x | out_proj(ctx)]) = x*W1a + (ctx*Wout + bout)*W1b + b1 = [x | ctx] * [W1a ; Wout*W1b] + (b1 + bout*W1b):
// folding the 256x256 out_proj into the first FFN layer (products formed in fp64, rounded once to fp32) removes one
// GEMM launch and the message tensor per block.  wout [256][256] and w1 [512][512] are [in][out] operands.
void fold_out_proj(const std::vector<float>& wout, const std::vector<float>& bout, const std::vector<float>& w1,
                   const std::vector<float>& b1, std::vector<float>& wf, std::vector<float>& bf) {
  wf = w1;
  bf = b1;
  for (int o = 0; o < 512; ++o) {
    double sb = b1[o];
    for (int j = 0; j < 256; ++j) sb += (double)bout[j] * w1[(size_t)(256 + j) * 512 + o];
    bf[o] = (float)sb;
  }
  // (row i of Wout * W1b: j outermost so that the inner loop walks W1b's rows contiguously — per (i, o) the products are added in the same order j = 0 .. 255 as
  // the dot-product form, so the folded weights are bit-identical; 33.5 M strided multiply-adds per block made a 9-layer handle take 1.2 s to create)
  std::vector<double> acc(512);
  for (int i = 0; i < 256; ++i) {
    std::fill(acc.begin(), acc.end(), 0.0);
    for (int j = 0; j < 256; ++j) {
      const double a = (double)wout[(size_t)i * 256 + j];
      const float* row = &w1[(size_t)(256 + j) * 512];
      for (int o = 0; o < 512; ++o) acc[o] += a * row[o];
    }
    for (int o = 0; o < 512; ++o) wf[(size_t)(256 + i) * 512 + o] = (float)acc[o];
  }
}
// nn.Linear weight [out][in] -> GEMM operand [in][out] (optionally scaled)
std::vector<float> transpose(const float* w, int out_f, int in_f, float scale = 1.0f) {
  std::vector<float> t((size_t)in_f * out_f);
  for (int o = 0; o < out_f; ++o)
    for (int i = 0; i < in_f; ++i) t[(size_t)i * out_f + o] = w[(size_t)o * in_f + i] * scale;
  return t;
}
std::vector<float> vec(const float* b, int n, float scale = 1.0f) {
  std::vector<float> v(n);
  for (int i = 0; i < n; ++i) v[i] = b[i] * scale;
  return v;
}
}  // namespace

extern "C" {

void dim_lg_destroy(dim_lg* h) {
  if (!h) return;
  for (void* p : h->allocs) hipFree(p);
  if (h->done_host) hipHostFree(h->done_host);
  delete h;
}

int dim_lg_create(const dim_lg_weights* w, const dim_lg_config* cfg, int max_pairs, int max_kpts, dim_lg** out) {
  DIM_REQUIRE(w && cfg && out && w->layers, "dim_lg_create: null argument");
  DIM_REQUIRE(w->n_layers >= 1 && w->n_layers <= 32, "dim_lg_create: n_layers %d", w->n_layers);
  DIM_REQUIRE(w->input_dim > 0 && w->input_dim % 32 == 0, "dim_lg_create: input_dim %d must be a multiple of 32", w->input_dim);
  DIM_REQUIRE((w->input_dim == 256) == (w->input_proj_w == nullptr), "dim_lg_create: input_proj must be given iff input_dim != 256 (LGN:361-364)");
  DIM_REQUIRE(max_pairs > 0 && max_kpts > 0, "dim_lg_create: bad sizes");
  dim_lg* h = new dim_lg();
  h->cfg = *cfg;
  h->n_layers = w->n_layers; h->input_dim = w->input_dim; h->max_pairs = max_pairs;
  h->nmax = (max_kpts + 3) & ~3;
  h->inproj_w = h->inproj_b = nullptr;
  h->thr.assign(w->confidence_thresholds, w->confidence_thresholds + w->n_layers);
#define LG_TRY(x) do { if ((x) != 0) { dim_lg_destroy(h); return -1; } } while (0)
  LG_TRY(upload(h, &h->Wr, vec(w->posenc_Wr, 64)));
  if (w->input_proj_w) {
    LG_TRY(upload(h, &h->inproj_w, transpose(w->input_proj_w, 256, w->input_dim)));
    LG_TRY(upload(h, &h->inproj_b, vec(w->input_proj_b, 256)));
  }
  h->L.resize(w->n_layers);
  for (int i = 0; i < w->n_layers; ++i) {
    const dim_lg_layer_weights& s = w->layers[i];
    LayerW& d = h->L[i];
    // Wqkv rows are interleaved head*192 + dim*3 + {q,k,v} (LGN:153-154); emit columns [q | k | v], head-major.
    std::vector<float> qkv((size_t)256 * 768), qkvb(768);
    for (int which = 0; which < 3; ++which)
      for (int hd = 0; hd < 4; ++hd)
        for (int dd = 0; dd < 64; ++dd) {
          const int o = hd * 192 + dd * 3 + which, c = which * 256 + hd * 64 + dd;
          qkvb[c] = s.self_Wqkv_b[o];
          for (int k = 0; k < 256; ++k) qkv[(size_t)k * 768 + c] = s.self_Wqkv_w[(size_t)o * 256 + k];
        }
    LG_TRY(upload(h, &d.qkv_w, qkv)); LG_TRY(upload(h, &d.qkv_b, qkvb)); LG_TRY(upload_x3(h, d.qkv_x, qkv, 256, 768));
    LG_TRY(upload(h, &d.out_w, transpose(s.self_out_w, 256, 256))); LG_TRY(upload_x3(h, d.out_x, transpose(s.self_out_w, 256, 256), 256, 256)); LG_TRY(upload(h, &d.out_b, vec(s.self_out_b, 256)));
    LG_TRY(upload(h, &d.sffn0_w, transpose(s.self_ffn0_w, 512, 512))); LG_TRY(upload_x3(h, d.sffn0_x, transpose(s.self_ffn0_w, 512, 512), 512, 512)); LG_TRY(upload(h, &d.sffn0_b, vec(s.self_ffn0_b, 512)));
    LG_TRY(upload(h, &d.sln_w, vec(s.self_ln_w, 512))); LG_TRY(upload(h, &d.sln_b, vec(s.self_ln_b, 512)));
    LG_TRY(upload(h, &d.sffn3_w, transpose(s.self_ffn3_w, 256, 512))); LG_TRY(upload_x3(h, d.sffn3_x, transpose(s.self_ffn3_w, 256, 512), 512, 256)); LG_TRY(upload(h, &d.sffn3_b, vec(s.self_ffn3_b, 256)));
    // to_qk and to_v fused into one [256][512] operand: columns [qk | v]
    std::vector<float> cq((size_t)256 * 512), cqb(512);
    for (int o = 0; o < 256; ++o) {
      cqb[o] = s.cross_qk_b[o]; cqb[256 + o] = s.cross_v_b[o];
      for (int k = 0; k < 256; ++k) {
        cq[(size_t)k * 512 + o] = s.cross_qk_w[(size_t)o * 256 + k];
        cq[(size_t)k * 512 + 256 + o] = s.cross_v_w[(size_t)o * 256 + k];
      }
    }
    LG_TRY(upload(h, &d.cqkv_w, cq)); LG_TRY(upload(h, &d.cqkv_b, cqb)); LG_TRY(upload_x3(h, d.cqkv_x, cq, 256, 512));
    LG_TRY(upload(h, &d.cout_w, transpose(s.cross_out_w, 256, 256))); LG_TRY(upload_x3(h, d.cout_x, transpose(s.cross_out_w, 256, 256), 256, 256)); LG_TRY(upload(h, &d.cout_b, vec(s.cross_out_b, 256)));
    LG_TRY(upload(h, &d.cffn0_w, transpose(s.cross_ffn0_w, 512, 512))); LG_TRY(upload_x3(h, d.cffn0_x, transpose(s.cross_ffn0_w, 512, 512), 512, 512)); LG_TRY(upload(h, &d.cffn0_b, vec(s.cross_ffn0_b, 512)));
    LG_TRY(upload(h, &d.cln_w, vec(s.cross_ln_w, 512))); LG_TRY(upload(h, &d.cln_b, vec(s.cross_ln_b, 512)));
    LG_TRY(upload(h, &d.cffn3_w, transpose(s.cross_ffn3_w, 256, 512))); LG_TRY(upload_x3(h, d.cffn3_x, transpose(s.cross_ffn3_w, 256, 512), 512, 256)); LG_TRY(upload(h, &d.cffn3_b, vec(s.cross_ffn3_b, 256)));
    {
      std::vector<float> wf, bf;
      fold_out_proj(transpose(s.self_out_w, 256, 256), vec(s.self_out_b, 256), transpose(s.self_ffn0_w, 512, 512), vec(s.self_ffn0_b, 512), wf, bf);
      LG_TRY(upload_x3(h, d.sffn0f_x, wf, 512, 512)); LG_TRY(upload(h, &d.sffn0f_b, bf));
      fold_out_proj(transpose(s.cross_out_w, 256, 256), vec(s.cross_out_b, 256), transpose(s.cross_ffn0_w, 512, 512), vec(s.cross_ffn0_b, 512), wf, bf);
      LG_TRY(upload_x3(h, d.cffn0f_x, wf, 512, 512)); LG_TRY(upload(h, &d.cffn0f_b, bf));
      LG_TRY(upload_x3(h, d.sffn3p_x, transpose(s.self_ffn3_w, 256, 512), 512, 256, 1));
      LG_TRY(upload_x3(h, d.cffn3p_x, transpose(s.cross_ffn3_w, 256, 512), 512, 256, 1));
    }
    LG_TRY(upload(h, &d.match_w, vec(s.assign_match_w, 256))); LG_TRY(upload(h, &d.match_b, vec(s.assign_match_b, 1)));
    // final_proj / d^0.25 (LGN:268-270): 256^0.25 = 4, a power of two -> folding the scale is exact
    LG_TRY(upload(h, &d.proj_w, transpose(s.assign_proj_w, 256, 256, 0.25f))); LG_TRY(upload_x3(h, d.proj_x, transpose(s.assign_proj_w, 256, 256, 0.25f), 256, 256)); LG_TRY(upload(h, &d.proj_b, vec(s.assign_proj_b, 256, 0.25f)));
    d.tok_w = d.tok_b = nullptr;
    if (s.token_w) { LG_TRY(upload(h, &d.tok_w, vec(s.token_w, 256))); LG_TRY(upload(h, &d.tok_b, vec(s.token_b, 1))); }
    DIM_REQUIRE(i == w->n_layers - 1 || s.token_w, "dim_lg_create: token_confidence.%d missing", i);
  }
  {
    std::vector<float> mw((size_t)w->n_layers * 256), mb(w->n_layers);
    for (int i = 0; i < w->n_layers; ++i) {
      memcpy(&mw[(size_t)i * 256], w->layers[i].assign_match_w, 256 * sizeof(float));
      mb[i] = w->layers[i].assign_match_b[0];
    }
    LG_TRY(upload(h, &h->match_w_all, mw)); LG_TRY(upload(h, &h->match_b_all, mb));
    for (int mode = 1; mode <= 2; ++mode) {
      std::vector<GemmLayerTab> tab(w->n_layers);
      for (int i = 0; i < w->n_layers; ++i) tab[i] = GemmLayerTab{h->L[i].proj_x[mode].dev, h->L[i].proj_x[mode].inv_ch(), h->L[i].proj_b};
      LG_TRY(dev_alloc(h, &h->proj_tab[mode], tab.size()));
      if (hipMemcpy(h->proj_tab[mode], tab.data(), tab.size() * sizeof(GemmLayerTab), hipMemcpyHostToDevice) != hipSuccess) {
        dim_set_error("weight upload failed"); dim_lg_destroy(h); return -1;
      }
    }
  }
  if (max_pairs <= FOLLOW_MAX_PAIRS) {
    void* q = nullptr;
    const size_t nb = (size_t)w->n_layers * (max_pairs + 1) * sizeof(int);
    if (hipHostMalloc(&q, nb, hipHostMallocMapped | hipHostMallocCoherent) == hipSuccess) {   // (without it the calls simply do not follow the stop flags)
      memset(q, 0, nb);
      h->done_host = (int*)q;
    } else {
      (void)hipGetLastError();
    }
  }
  LgState& st = h->st;
  const size_t P = max_pairs, I = 2 * P, N = h->nmax;
  st.n_pairs = max_pairs; st.n_items = 2 * max_pairs; st.nmax = h->nmax; st.nsel = h->nmax;
  LG_TRY(dev_alloc(h, &st.desc, I * N * 256)); LG_TRY(dev_alloc(h, &st.enc, I * N * 64));
  LG_TRY(dev_alloc(h, &st.qkv, I * N * 768)); LG_TRY(dev_alloc(h, &st.ctx, I * N * 256));
  LG_TRY(dev_alloc(h, &st.msg, I * N * 256)); LG_TRY(dev_alloc(h, &st.hid, I * N * 512));
  LG_TRY(dev_alloc(h, &st.md, I * N * 256)); LG_TRY(dev_alloc(h, &st.sim, P * N * N));
  LG_TRY(dev_alloc(h, &st.conf, I * N)); LG_TRY(dev_alloc(h, &st.mtch, I * N)); LG_TRY(dev_alloc(h, &st.zls, I * N));
  LG_TRY(dev_alloc(h, &st.rmax, I * N)); LG_TRY(dev_alloc(h, &st.rlse, I * N)); LG_TRY(dev_alloc(h, &st.best, I * N));
  LG_TRY(dev_alloc(h, &st.arg, I * N)); LG_TRY(dev_alloc(h, &st.n_cur, I)); LG_TRY(dev_alloc(h, &st.n_new, I));
  LG_TRY(dev_alloc(h, &st.n_orig, I)); LG_TRY(dev_alloc(h, &st.ind, I * N)); LG_TRY(dev_alloc(h, &st.dest, I * N));
  LG_TRY(dev_alloc(h, &st.prune, I * N)); LG_TRY(dev_alloc(h, &st.done, P)); LG_TRY(dev_alloc(h, &st.cnt_lt, P));
  { unsigned char* kvp = nullptr; LG_TRY(dev_alloc(h, &kvp, I * 4 * ((N + 31) / 32) * 1536 * 16)); st.kv_img = kvp; }
  st.attn_part_items = (int)(I < 8 ? I : 8);  // key-split attention only pays for <= 4 pairs
  if (dim_attn_probe()) st.attn_part_items = (int)I;   // timing probes (dim_tune_set key 12): 16 partial records per row for every item
  LG_TRY(dev_alloc(h, &st.attn_part, (size_t)st.attn_part_items * 4 * N * (dim_attn_probe() ? 16 : 4) * 68));
  LG_TRY(dev_alloc(h, &st.tdesc, I * N * 256)); LG_TRY(dev_alloc(h, &st.tenc, I * N * 64)); LG_TRY(dev_alloc(h, &st.tind, I * N));
#undef LG_TRY
  *out = h;
  return 0;
}

int dim_lg_match(dim_lg* h, const float* kpts_tab_dev, const float* desc_tab_dev, const int32_t* n_tab_dev,
                 const float* size_tab_dev, int cap, const int32_t* pair_idx_dev, int n_pairs, int64_t* matches_dev,
                 float* mscores_dev, int32_t* n_matches_dev, int32_t* matches01_dev, float* mscores01_dev,
                 int32_t* stop_dev, int32_t* prune01_dev, float* dense_scores_dev, void* stream) {
  DIM_REQUIRE(h && kpts_tab_dev && desc_tab_dev && n_tab_dev && size_tab_dev, "dim_lg_match: null input");
  DimTuneScope tune_scope(&h->base);
  DIM_REQUIRE(matches_dev && mscores_dev && n_matches_dev && matches01_dev && mscores01_dev && stop_dev && prune01_dev, "dim_lg_match: null output");
  DIM_REQUIRE(n_pairs >= 1 && n_pairs <= h->max_pairs, "dim_lg_match: n_pairs %d outside [1,%d]", n_pairs, h->max_pairs);
  DIM_REQUIRE(cap > 0, "dim_lg_match: cap");
  hipStream_t s = (hipStream_t)stream;
  LgState st = h->st;
  st.n_pairs = n_pairs; st.n_items = 2 * n_pairs;
  const int N = st.nmax, I = st.n_items, Lr = h->n_layers;
  // Nsel: no item can have more live rows than the feature table has rows per image (`cap`), whatever the handle's capacity N is — the plugins size their
  // handles to the next power of two, so a 2100-keypoint pair runs on a 4096-row handle.  Launch shapes and kernel selections (the one-pair GEMM blocks, the
  // key-split attention, K | V images from the projection, the fast assignment kernels) follow Nsel; strides and layouts follow N.  (Round 6, found by sweeping the
  // keypoint count on power-of-two handles: 2100 keypoints on a 4096-row handle 2.24 ms, on a handle of its own size ~1.6.)
  const int Nsel = std::min(N, (cap + 3) & ~3);
  st.nsel = Nsel;
  const long long s256 = (long long)N * 256, s512 = (long long)N * 512, s768 = (long long)N * 768;
  const bool early = h->cfg.depth_confidence > 0, prune = h->cfg.width_confidence > 0;  // LGN:480-481
#define LG_RUN(x) do { int rc__ = (x); if (rc__ != 0) return rc__; } while (0)
  const int pmode = dim_precision_mode();  // 2 fp16x3 (default) / 1 bf16x6: fp32-accurate products on the 16-bit matrix cores; 0 fp32 MFMA
  const bool x6 = pmode != 0;
  // fp16x3 range guard (dim_common.h): every producer of a value that a later split consumes reports max|x| > 4094
  auto sat = [&](int site) -> unsigned* { return pmode == 2 ? dim_sat_counter(site) : nullptr; };
  st.sat_qkv = sat(DIM_SAT_LG_QKV); st.sat_ffn = sat(DIM_SAT_LG_FFN);
  LG_RUN(launch_lg_init(st, kpts_tab_dev, desc_tab_dev, n_tab_dev, size_tab_dev, pair_idx_dev, cap, h->input_dim, h->Wr,
                        h->input_dim == 256 ? 1 : 0, sat(DIM_SAT_LG_INPUT), s));
  const bool fold = x6 && dim_fold_out_proj();  // split modes: out_proj folded into ffn.0 (one GEMM less per block)
  // fp16x3 at batch sizes that fill the GPU with 64-row blocks: LayerNorm + GELU run in ffn.0's epilogue (dim_tune_set key 11)
  const bool fuse_ln = pmode == 2 && dim_fuse_ffn_ln() && (dim_fuse_ffn_ln() == 2 || (long)((Nsel + 63) / 64) * I >= 512);   // 2 = forced (tests)
  auto gemm_items = [&](const float* A, int lda, long long sA, const float* A1, int lda1, long long sA1, int ksplit,
                        const float* B, const SplitWeights* Bx, int ldb, const float* bias, const float* R, float* C, int ldc,
                        long long sC, int Nn, int K, int flag_eq, unsigned* sat_ctr = nullptr, const float* ln_g = nullptr,
                        const float* ln_b = nullptr) -> int {
    GemmArgs g;
    g.ln_gamma = ln_g; g.ln_beta = ln_b;
    g.A0 = A; g.lda0 = lda; g.strideA0 = sA; g.A1 = A1; g.lda1 = lda1; g.strideA1 = sA1; g.ksplit = ksplit;
    g.B = B; g.ldb = ldb; g.bias = bias; g.R = R; g.ldr = ldc; g.strideR = sC; g.C = C; g.ldc = ldc; g.strideC = sC;
    g.M = Nsel; g.N = Nn; g.K = K; g.rows = st.n_cur; g.flag = st.done; g.flag_shift = 1; g.flag_eq = flag_eq;
    g.sat = sat_ctr;
    if (x6) { g.set_split(Bx[pmode]); return launch_gemm_x6(g, I, s); }
    return launch_gemm(g, I, s);
  };
  // ... and ffn.3 + the residual in the same kernel (dim_tune_set key 11 = 3 / 4 = forced): the hidden tensor is never stored
  const bool fuse_ffn = fold && pmode == 2 && (dim_fuse_ffn_ln() == 4 || (dim_fuse_ffn_ln() == 3 && (long)((Nsel + 63) / 64) * I >= 512));
  auto ffn_fused = [&](const SplitWeights* w0, const float* b0, const float* lg, const float* lb, const SplitWeights* w3, const float* b3) -> int {
    GemmArgs g;
    g.A0 = st.desc; g.lda0 = 256; g.strideA0 = s256; g.A1 = st.ctx; g.lda1 = 256; g.strideA1 = s256; g.ksplit = 256;
    g.bias = b0; g.ln_gamma = lg; g.ln_beta = lb; g.set_split(w0[2]); g.set_split2(w3[2]); g.bias2 = b3;
    g.R = st.desc; g.ldr = 256; g.strideR = s256; g.C = st.desc; g.ldc = 256; g.strideC = s256;
    g.M = Nsel; g.N = 512; g.K = 512; g.rows = st.n_cur; g.flag = st.done; g.flag_shift = 1; g.flag_eq = 0;
    g.sat = st.sat_ffn; g.sat2 = sat(DIM_SAT_LG_DESC);
    return launch_gemm_x6(g, I, s);
  };
  // fp16x3 at batch sizes that run the 128 x 256 GEMM block: the q|k|v projections write the attention kernel's K | V tile
  // images themselves (rotary + pre-split in the epilogue; dim_tune_set key 8 = 0 keeps the separate kv_prep pass)
  const bool fuse_kv = pmode == 2 && dim_fuse_kv() && gemm_x6_fuses_kv(Nsel, 512, I, 2) && gemm_x6_fuses_kv(Nsel, 768, I, 2);
  auto gemm_qkv = [&](const SplitWeights* Bx, const float* bias, int Nn, int kblock, bool rotary) -> int {
    GemmArgs g;
    g.A0 = st.desc; g.lda0 = 256; g.strideA0 = s256;
    g.bias = bias; g.C = st.qkv; g.ldc = 768; g.strideC = s768;
    g.M = Nsel; g.N = Nn; g.K = 256; g.rows = st.n_cur; g.flag = st.done; g.flag_shift = 1; g.flag_eq = 0;
    g.sat = st.sat_qkv;
    g.kv_img = st.kv_img; g.kv_tiles = (N + 31) / 32; g.kv_kblock = kblock; g.kv_vblock = kblock + 1; g.kv_nmax = N;
    g.kv_enc = rotary ? st.enc : nullptr;
    g.set_split(Bx[2]);
    return launch_gemm_x6(g, I, s);
  };
  if (h->input_dim != 256) {  // input_proj (LGN:473-474) straight from the feature table
    GemmArgs g;
    g.A0 = desc_tab_dev; g.lda0 = h->input_dim; g.strideA0 = (long long)cap * h->input_dim; g.a_idx = pair_idx_dev;
    g.B = h->inproj_w; g.ldb = 256; g.bias = h->inproj_b; g.C = st.desc; g.ldc = 256; g.strideC = s256;
    g.M = Nsel; g.N = 256; g.K = h->input_dim; g.rows = st.n_cur; g.sat = sat(DIM_SAT_LG_INPUT);
    LG_RUN(launch_gemm(g, I, s));
  }
  // ---- assignment (LGN:540-542).  tag = layer + 1: the pairs that stopped at that layer, with its weights (gated launches after every layer that may
  // stop pairs).  tag = 0, the DEFERRED form of adaptive depth (round 6): a stopped pair's descriptors, counts and index tables are frozen — every later
  // launch skips it — so ONE pass after the loop serves every pair, each item reading final_proj / matchability weights of its own stop layer from
  // per-layer tables (GemmLayerTab, match_w_all).  Same arithmetic on the same operands: identical results; 6 launches per call instead of 6 per layer —
  // at one pair per call (the plugin hooks, reference-default confidences) ~48 empty launches of ~3 us each.  Split modes only (dim_tune_set key 17).
  const bool defer = early && x6 && dim_defer_assignment() != 0;
  auto assignment = [&](int tag, const LayerW* w) -> int {
    {
      GemmArgs g;
      g.A0 = st.desc; g.lda0 = 256; g.strideA0 = s256; g.C = st.md; g.ldc = 256; g.strideC = s256; g.ldr = 256; g.strideR = s256;
      g.M = Nsel; g.N = 256; g.K = 256; g.rows = st.n_cur; g.flag = st.done; g.flag_shift = 1; g.flag_eq = tag;
      g.sat = sat(DIM_SAT_LG_DESC);   // guarded: the similarity splits it
      if (tag == 0) { g.set_split(h->L[0].proj_x[pmode]); g.bias = h->L[0].proj_b; g.layer_tab = h->proj_tab[pmode]; }   // (shape fields from layer 0; pointers per item)
      else { g.B = w->proj_w; g.ldb = 256; g.bias = w->proj_b; if (x6) g.set_split(w->proj_x[pmode]); }
      if (x6) LG_RUN(launch_gemm_x6(g, I, s)); else LG_RUN(launch_gemm(g, I, s));
    }
    GemmArgs g;
    g.A0 = st.md; g.lda0 = 256; g.strideA0 = 2 * s256; g.B = st.md + s256; g.ldb = 256; g.strideB = 2 * s256; g.bt = 1;
    g.C = st.sim; g.ldc = N; g.strideC = (long long)N * N; g.M = Nsel; g.N = Nsel; g.K = 256;
    g.rows = st.n_cur; g.rows_mul = 2; g.rows_off = 0; g.cols = st.n_cur; g.cols_mul = 2; g.cols_off = 1;
    g.flag = st.done; g.flag_shift = 0; g.flag_eq = tag; g.flag_any = tag == 0 ? 1 : 0;
    if (x6) LG_RUN(launch_gemm_x6_nt(g, n_pairs, pmode, s));   // split-precision on the 16-bit matrix cores like every other product
    else LG_RUN(launch_gemm(g, n_pairs, s));
    LG_RUN(launch_lg_assign_stats(st, tag, tag == 0 ? h->match_w_all : w->match_w, tag == 0 ? h->match_b_all : w->match_b, s));
    LG_RUN(launch_lg_assign_argmax(st, tag, dense_scores_dev, s));
    return 0;
  };
  // ---- one or two pairs per call (the plugin hooks) with adaptive depth: the HOST follows the stop flags two layers behind the device.  Every launch
  // skips a stopped pair by itself, but a launch that finds nothing to do still costs ~3 us, and reference-default LightGlue stops most pairs of
  // an exhaustive job after 3 - 5 of 9 layers: ~20 launches per skipped layer = more time than the layers that ran.  After layer i's decide the flags
  // land in page-locked host memory (lg_decide_kernel writes them there itself, then the call's sequence number behind a system-scope fence); before
  // layer i + 2 is enqueued the host spins on that word (the device is then working on layer i + 1, which is already in the stream: no bubble as
  // long as a layer lasts longer than the hand-over) and leaves the loop when every pair has stopped.  Results cannot change: the skipped launches
  // would all have returned at their flag test.  (A first version copied the flags with hipMemcpyAsync + an event per layer: the nine blit kernels
  // cost a pair that runs all layers 0.04 ms, profiles/r06_b1_adaptive.json.)  dim_tune_set key 18 = 0: never follow — the call then enqueues
  // everything without touching the host, as batched calls always do.
  bool follow = early && h->done_host != nullptr && n_pairs <= FOLLOW_MAX_PAIRS && dim_follow_stop_flags() != 0;
  const int mstride = h->max_pairs + 1;
  const int seq = follow ? (h->call_seq = (h->call_seq % 0x3fffffff) + 1) : 0;
  for (int i = 0; i < Lr; ++i) {
    const LayerW& w = h->L[i];
    if (follow && i >= 2) {
      const int* m = h->done_host + (size_t)(i - 2) * mstride;
      bool seen = false;
      const auto t_end = std::chrono::steady_clock::now() + std::chrono::milliseconds(500);   // (at worst: then the call simply goes on enqueueing)
      for (long spin = 0;; ++spin) {
        if (__atomic_load_n(m + h->max_pairs, __ATOMIC_ACQUIRE) == seq) { seen = true; break; }
        if ((spin & 1023) == 1023) {
          if (std::chrono::steady_clock::now() > t_end) break;
          std::this_thread::yield();
        }
      }
      bool all = seen;
      for (int p = 0; p < n_pairs; ++p) all = all && m[p] != 0;
      if (all) break;
      if (!seen) follow = false;   // the device is not making progress while we wait (a stream held back by the caller?): enqueue the rest without looking
    }
    // ---- self block (LGN:146-159) ----
    if (fuse_kv) LG_RUN(gemm_qkv(w.qkv_x, w.qkv_b, 768, 1, true));
    else LG_RUN(gemm_items(st.desc, 256, s256, nullptr, 0, 0, 0, w.qkv_w, w.qkv_x, 768, w.qkv_b, nullptr, st.qkv, 768, s768, 768, 256, 0, st.sat_qkv));
    if (!x6) LG_RUN(launch_lg_rotary(st, s));  // the bf16x6 attention applies the rotary embedding while it loads q / pre-splits k
    dim_prof_begin(DIM_PROF_LG_SELF_ATTN, s);
    LG_RUN(launch_lg_attention(st, 0, s, fuse_kv ? 1 : 0));
    dim_prof_end(DIM_PROF_LG_SELF_ATTN, s);
    if (fuse_ffn) LG_RUN(ffn_fused(w.sffn0f_x, w.sffn0f_b, w.sln_w, w.sln_b, w.sffn3p_x, w.sffn3_b));
    else {
    if (fold) {  // out_proj folded into ffn.0: A = [desc | ctx]
      LG_RUN(gemm_items(st.desc, 256, s256, st.ctx, 256, s256, 256, nullptr, w.sffn0f_x, 512, w.sffn0f_b, nullptr, st.hid, 512, s512, 512, 512, 0,
                        fuse_ln ? st.sat_ffn : nullptr, fuse_ln ? w.sln_w : nullptr, fuse_ln ? w.sln_b : nullptr));
    } else {
      LG_RUN(gemm_items(st.ctx, 256, s256, nullptr, 0, 0, 0, w.out_w, w.out_x, 256, w.out_b, nullptr, st.msg, 256, s256, 256, 256, 0));
      LG_RUN(gemm_items(st.desc, 256, s256, st.msg, 256, s256, 256, w.sffn0_w, w.sffn0_x, 512, w.sffn0_b, nullptr, st.hid, 512, s512, 512, 512, 0));
    }
    if (!(fold && fuse_ln)) LG_RUN(launch_lg_ln_gelu(st, w.sln_w, w.sln_b, s));
    LG_RUN(gemm_items(st.hid, 512, s512, nullptr, 0, 0, 0, w.sffn3_w, w.sffn3_x, 256, w.sffn3_b, st.desc, st.desc, 256, s256, 256, 512, 0, sat(DIM_SAT_LG_DESC)));
    }
    // ---- cross block (LGN:186-211) ----
    if (fuse_kv) LG_RUN(gemm_qkv(w.cqkv_x, w.cqkv_b, 512, 0, false));
    else LG_RUN(gemm_items(st.desc, 256, s256, nullptr, 0, 0, 0, w.cqkv_w, w.cqkv_x, 512, w.cqkv_b, nullptr, st.qkv, 768, s768, 512, 256, 0, st.sat_qkv));
    dim_prof_begin(DIM_PROF_LG_CROSS_ATTN, s);
    LG_RUN(launch_lg_attention(st, 1, s, fuse_kv ? 1 : 0));
    dim_prof_end(DIM_PROF_LG_CROSS_ATTN, s);
    if (fuse_ffn) LG_RUN(ffn_fused(w.cffn0f_x, w.cffn0f_b, w.cln_w, w.cln_b, w.cffn3p_x, w.cffn3_b));
    else {
    if (fold) {
      LG_RUN(gemm_items(st.desc, 256, s256, st.ctx, 256, s256, 256, nullptr, w.cffn0f_x, 512, w.cffn0f_b, nullptr, st.hid, 512, s512, 512, 512, 0,
                        fuse_ln ? st.sat_ffn : nullptr, fuse_ln ? w.cln_w : nullptr, fuse_ln ? w.cln_b : nullptr));
    } else {
      LG_RUN(gemm_items(st.ctx, 256, s256, nullptr, 0, 0, 0, w.cout_w, w.cout_x, 256, w.cout_b, nullptr, st.msg, 256, s256, 256, 256, 0));
      LG_RUN(gemm_items(st.desc, 256, s256, st.msg, 256, s256, 256, w.cffn0_w, w.cffn0_x, 512, w.cffn0_b, nullptr, st.hid, 512, s512, 512, 512, 0));
    }
    if (!(fold && fuse_ln)) LG_RUN(launch_lg_ln_gelu(st, w.cln_w, w.cln_b, s));
    LG_RUN(gemm_items(st.hid, 512, s512, nullptr, 0, 0, 0, w.cffn3_w, w.cffn3_x, 256, w.cffn3_b, st.desc, st.desc, 256, s256, 256, 512, 0, sat(DIM_SAT_LG_DESC)));
    }
    // ---- adaptive depth / width (LGN:494-516) ----
    const bool last = (i == Lr - 1);
    if (!last && (early || prune))
      LG_RUN(launch_lg_confidence(st, w.tok_w, w.tok_b, w.match_w, w.match_b, h->thr[i], early ? 1 : 0, s));
    if (last || early)
      LG_RUN(launch_lg_decide(st, i, (float)h->cfg.depth_confidence, early ? 1 : 0, last ? 1 : 0, s, follow ? h->done_host + (size_t)i * mstride : nullptr, h->max_pairs, seq));
    if ((last || early) && !defer) LG_RUN(assignment(i + 1, &w));   // the pairs that stopped at this layer (LGN:540-542)
    if (!last && prune)
      LG_RUN(launch_lg_prune(st, i, h->cfg.width_confidence, h->thr[i], early ? 1 : 0, h->cfg.pruning_min_kpts, s));
  }
  if (defer) LG_RUN(assignment(0, nullptr));
  LG_RUN(launch_lg_finalize(st, Lr, prune ? 1 : 0, (float)h->cfg.filter_threshold, N, (long long*)matches_dev, mscores_dev,
                            n_matches_dev, matches01_dev, mscores01_dev, stop_dev, prune01_dev, s));
#undef LG_RUN
  return 0;
}

int dim_lg_max_kpts(dim_lg* h) { return h ? h->nmax : -1; }

int dim_lg_stage_features(const dim_lg_raw_features* img0, const dim_lg_raw_features* img1, int cap, int D, float* kpts_tab_dev, float* desc_tab_dev,
                          void* stream) {
  DIM_REQUIRE(img0 && img1 && kpts_tab_dev && desc_tab_dev, "dim_lg_stage_features: null argument");
  DIM_REQUIRE(cap > 0 && D > 0 && D % 32 == 0, "dim_lg_stage_features: cap %d, D %d (a multiple of 32)", cap, D);
  StageArgs a;
  const dim_lg_raw_features* im[2] = {img0, img1};
  for (int i = 0; i < 2; ++i) {
    DIM_REQUIRE(im[i]->n >= 0 && im[i]->n <= cap, "dim_lg_stage_features: image %d has %d keypoints, table capacity %d", i, im[i]->n, cap);
    DIM_REQUIRE(im[i]->n == 0 || (im[i]->kpts_dev && im[i]->desc_dev), "dim_lg_stage_features: image %d: null arrays", i);
    DIM_REQUIRE(((size_t)im[i]->kpts_dev & 3) == 0 && ((size_t)im[i]->desc_dev & 15) == 0, "dim_lg_stage_features: image %d: keypoints must be 4-byte, descriptors 16-byte aligned", i);
    a.kp[i] = im[i]->kpts_dev; a.ds[i] = im[i]->desc_dev; a.n[i] = im[i]->n;
    a.kp_f16[i] = im[i]->kpts_f16 ? 1 : 0; a.ds_f16[i] = im[i]->desc_f16 ? 1 : 0; a.ds_dn[i] = im[i]->desc_is_dn ? 1 : 0;
  }
  a.cap = cap; a.D = D; a.kt = kpts_tab_dev; a.dt = desc_tab_dev;
  hipLaunchKernelGGL(lg_stage_kernel, dim3(cdiv(cap, 32), 2), dim3(256), 0, (hipStream_t)stream, a);
  DIM_LAUNCH_CHECK();
  return 0;
}

int dim_lg_debug_desc(dim_lg* h, const float** desc, const int32_t** n_cur, const int32_t** ind) {
  DIM_REQUIRE(h, "dim_lg_debug_desc: null handle");
  if (desc) *desc = h->st.desc;
  if (n_cur) *n_cur = h->st.n_cur;
  if (ind) *ind = h->st.ind;
  return 0;
}

}  // extern "C"
