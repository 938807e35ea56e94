// dim_aliked_* : resident ALIKED extractor (C ABI in include/dim_hip.h).
// Replaces ALIKED.__init__/forward (ALN:561-693) as driven by AlikedExtractor._extract
// (extractors/aliked.py:45-64), quirks Q7 (train-mode BatchNorm) and Q8 (scores = dispersity) included.
#include <stdlib.h>
#include <string.h>

#include <vector>

#include "../../include/dim_hip.h"
#include "aliked_kernels.h"
#include "sp_kernels.h"

struct dim_aliked {
  DimHandleBase base;   // first member: dim_handle_tune_set
  dim_aliked_config cfg;
  int max_batch, max_h, max_w, capacity;
  // weights (device, kernel layouts)
  float *b1c1, *b1c2, *b2c1, *b2c2, *b2ds_w, *b2ds_b;
  float *b3o1_w, *b3o1_b, *b3r1, *b3o2_w, *b3o2_b, *b3r2, *b3ds_w, *b3ds_b;
  float *b4o1_w, *b4o1_b, *b4r1, *b4o2_w, *b4o2_b, *b4r2, *b4ds_w, *b4ds_b;
  float *bn_g[8], *bn_b[8];
  float *hc1, *hc2, *hc3, *hc4, *sh0, *sh2, *sh4, *sh6;
  float *dh_o0_w, *dh_o0_b, *dh_o2_w, *dh_o2_b, *dh_sf, *dh_agg;
  // activations
  float *P, *raw, *act, *x1, *p2, *idn, *x2, *p3, *off, *x3, *p4, *x4, *f2, *f3, *f4, *x1234, *s8, *s4a, *s4b, *score, *nms;
  SplitWeights x_b1c1, x_b1c2, x_b2c1, x_b2c2, x_b2ds;   // fp16x3 fragments of the full- / half-resolution convolutions (aliked_x3.hip)
  double* tile_partial;
  unsigned short* asm_frag; float* asm_inv1; float asm_inv0;   // constant operands of al_assemble_x3_kernel
  SplitWeights g_b3r1, g_b3r2, g_b3ds, g_b4r1, g_b4r2, g_b4ds, g_hc2, g_hc3, g_hc4, g_o0, g_sf, g_agg;   // fp16x3 GEMM operands (gemm_x6.hip)
  float *q2, *q3, *q4;   // score_head.0 projections of f2 / f3 / f4 at their own resolutions (8 channels)
  float *cand_score, *kpts_px, *sc_tmp, *kpts_norm, *kscore, *patches, *hidden, *feats, *feats2, *bn_alpha, *bn_beta, *mean, *thr_eff, *cols;
  double* partial;
  int *cand_idx, *rowcount, *rowoff, *ncand;
  unsigned long long* topk_keys;   // launch_topk's global key table (n_limit > 4096 only)
  int last_hp, last_wp, last_h, last_w, last_batch;
  float* dbg_x1234;  // debug tap only: materialised on request by dim_aliked_debug_buffers
  std::vector<void*> allocs;
};

namespace {
template <typename T>
int dev_alloc(dim_aliked* h, T** p, size_t count) {
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
int upload(dim_aliked* h, float** dst, const std::vector<float>& v) {
  if (!dim_all_finite(v.data(), v.size())) { dim_set_error("non-finite value in the weights"); return -1; }
  if (dev_alloc(h, dst, v.size()) != 0) return -1;
  if (hipMemcpy(*dst, v.data(), v.size() * sizeof(float), hipMemcpyHostToDevice) != hipSuccess) {
    dim_set_error("weight upload failed");
    return -1;
  }
  return 0;
}
// OIHW -> [tap][cin_pad][cout_pad] (zero padded)
std::vector<float> relayout(const float* w, int co, int ci, int k, int ci_pad, int co_pad) {
  std::vector<float> o((size_t)k * k * ci_pad * co_pad, 0.0f);
  for (int a = 0; a < co; ++a)
    for (int b = 0; b < ci; ++b)
      for (int t = 0; t < k * k; ++t) o[((size_t)t * ci_pad + b) * co_pad + a] = w[((size_t)a * ci + b) * k * k + t];
  return o;
}
// [K][N] fp32 conv operand (K = tap * cin_pad + ci) -> fp16x3 MFMA fragments + per-channel inverse scales (n_pad = 32)
int upload_x3(dim_aliked* h, SplitWeights* dst, const std::vector<float>& w_kn, int K, int N) {
  if (!dim_all_finite(w_kn.data(), w_kn.size())) { dim_set_error("non-finite value in the weights"); return -1; }
  std::vector<unsigned short> host(gemm_split_weight_elems(K, 32, 2));
  split_weights(w_kn.data(), K, N, 32, 2, host.data(), dst);
  unsigned short* d = nullptr;
  if (dev_alloc(h, &d, host.size()) != 0) return -1;
  if (hipMemcpy(d, host.data(), host.size() * 2, hipMemcpyHostToDevice) != hipSuccess) { dim_set_error("weight upload failed"); return -1; }
  dst->dev = d; dst->mode = 2; dst->n_pad = 32;
  return 0;
}
// [K][N] fp32 GEMM operand -> fp16x3 planes in gemm_x6's fragment order (n_pad = multiple of 128)
int upload_x3g(dim_aliked* h, SplitWeights* dst, const float* w_kn, int K, int N) {
  const int n_pad = (N + 127) / 128 * 128;
  std::vector<unsigned short> host(gemm_split_weight_elems(K, n_pad, 2));
  split_weights(w_kn, K, N, n_pad, 2, host.data(), dst);
  unsigned short* d = nullptr;
  if (dev_alloc(h, &d, host.size()) != 0) return -1;
  if (hipMemcpy(d, host.data(), host.size() * 2, hipMemcpyHostToDevice) != hipSuccess) { dim_set_error("weight upload failed"); return -1; }
  dst->dev = d; dst->mode = 2; dst->n_pad = n_pad;
  return 0;
}
std::vector<float> padvec(const float* b, int n, int n_pad) {
  std::vector<float> v(n_pad, 0.0f);
  for (int i = 0; i < n; ++i) v[i] = b[i];
  return v;
}
}  // namespace

extern "C" {

void dim_aliked_destroy(dim_aliked* h) {
  if (!h) return;
  for (void* p : h->allocs) hipFree(p);
  if (h->dbg_x1234) hipFree(h->dbg_x1234);
  delete h;
}

int dim_aliked_create(const dim_aliked_weights* w, const dim_aliked_config* cfg, int max_batch, int max_h, int max_w,
                      int capacity, dim_aliked** out) {
  DIM_REQUIRE(w && cfg && out, "dim_aliked_create: null argument");
  const bool normal = cfg->c1 == 16 && cfg->c2 == 32 && cfg->c3 == 64 && cfg->c4 == 128 && cfg->dim == 128 && (cfg->M == 16 || cfg->M == 32);
  const bool tiny = cfg->c1 == 8 && cfg->c2 == 16 && cfg->c3 == 32 && cfg->c4 == 64 && cfg->dim == 64 && cfg->M == 16;
  DIM_REQUIRE((normal || tiny) && cfg->K == 3,
              "dim_aliked_create: geometry (%d,%d,%d,%d,%d,%d,%d) is none of ALN:573-579 (aliked-t16 / n16 / n16rot / n32)", cfg->c1, cfg->c2, cfg->c3,
              cfg->c4, cfg->dim, cfg->K, cfg->M);
  const int c1 = cfg->c1, c2 = cfg->c2, c3 = cfg->c3, c4 = cfg->c4, dim = cfg->dim, G = cfg->dim / 4;
  const int M = cfg->M, M2 = 2 * cfg->M;   // SDDH sample positions; offset channels (ALN:503-519)
  // detection_threshold <= 0 with max_num_keypoints > 0 selects DKD's top-k mode (ALN:624-627: top_k = max_num_keypoints): the
  // max_num_keypoints highest NMS maxima (filled up with zero-score pixels when there are fewer, as torch.topk does); with max_num_keypoints <= 0
  // as well the reference's top_k is <= 0 and DKD thresholds at the image's MEAN score (ALN:161-163), at most n_limit_max = 20000 (ALN:571)
  // of them (or `capacity`, if that is smaller) — both handled in dim_aliked_extract
  DIM_REQUIRE(cfg->nms_radius >= 1 && cfg->nms_radius <= 6, "dim_aliked_create: nms_radius %d", cfg->nms_radius);
  DIM_REQUIRE(capacity > 0 && capacity <= 32768 && cfg->max_num_keypoints <= capacity, "dim_aliked_create: capacity %d (<= 32768) must cover max_num_keypoints %d", capacity, cfg->max_num_keypoints);
  DIM_REQUIRE(max_batch > 0 && max_batch <= 64 && max_h >= 16 && max_w >= 16, "dim_aliked_create: bad sizes");
  dim_aliked* h = new dim_aliked();
  h->cfg = *cfg;
  h->max_batch = max_batch; h->max_h = max_h; h->max_w = max_w; h->capacity = capacity;
#define AL_TRY(x) do { if ((x) != 0) { dim_aliked_destroy(h); return -1; } } while (0)
  // [tap][ci][co] operand of a deformable convolution's GEMM with the rows padded to the K granule (al_deform_krow: aliked-t16's 144 -> 160)
  auto deform_w = [&](const float* w_, int co, int ci) {
    std::vector<float> v = relayout(w_, co, ci, 3, ci, co);
    v.resize((size_t)al_deform_krow(ci) * co, 0.0f);
    return v;
  };
  AL_TRY(upload(h, &h->b1c1, relayout(w->block1_conv1, c1, 3, 3, 4, c1)));
  AL_TRY(upload(h, &h->b1c2, relayout(w->block1_conv2, c1, c1, 3, c1, c1)));
  AL_TRY(upload(h, &h->b2c1, relayout(w->block2_conv1, c2, c1, 3, c1, c2)));
  AL_TRY(upload(h, &h->b2c2, relayout(w->block2_conv2, c2, c2, 3, c2, c2)));
  AL_TRY(upload(h, &h->b2ds_w, relayout(w->block2_ds_w, c2, c1, 1, c1, c2))); AL_TRY(upload(h, &h->b2ds_b, padvec(w->block2_ds_b, c2, c2)));
  AL_TRY(upload(h, &h->b3o1_w, relayout(w->block3_off1_w, 18, c2, 3, c2, 20))); AL_TRY(upload(h, &h->b3o1_b, padvec(w->block3_off1_b, 18, 20)));
  AL_TRY(upload(h, &h->b3r1, deform_w(w->block3_reg1, c3, c2)));
  AL_TRY(upload(h, &h->b3o2_w, relayout(w->block3_off2_w, 18, c3, 3, c3, 20))); AL_TRY(upload(h, &h->b3o2_b, padvec(w->block3_off2_b, 18, 20)));
  AL_TRY(upload(h, &h->b3r2, deform_w(w->block3_reg2, c3, c3)));
  AL_TRY(upload(h, &h->b3ds_w, relayout(w->block3_ds_w, c3, c2, 1, c2, c3))); AL_TRY(upload(h, &h->b3ds_b, padvec(w->block3_ds_b, c3, c3)));
  AL_TRY(upload(h, &h->b4o1_w, relayout(w->block4_off1_w, 18, c3, 3, c3, 20))); AL_TRY(upload(h, &h->b4o1_b, padvec(w->block4_off1_b, 18, 20)));
  AL_TRY(upload(h, &h->b4r1, deform_w(w->block4_reg1, c4, c3)));
  AL_TRY(upload(h, &h->b4o2_w, relayout(w->block4_off2_w, 18, c4, 3, c4, 20))); AL_TRY(upload(h, &h->b4o2_b, padvec(w->block4_off2_b, 18, 20)));
  AL_TRY(upload(h, &h->b4r2, deform_w(w->block4_reg2, c4, c4)));
  AL_TRY(upload(h, &h->b4ds_w, relayout(w->block4_ds_w, c4, c3, 1, c3, c4))); AL_TRY(upload(h, &h->b4ds_b, padvec(w->block4_ds_b, c4, c4)));
  if (normal) {   // the matrix-core forms of the full- / half-resolution convolutions and of the aggregation exist for the 16 / 32-channel geometry
    AL_TRY(upload_x3(h, &h->x_b1c1, relayout(w->block1_conv1, 16, 3, 3, 16, 16), 9 * 16, 16));
    AL_TRY(upload_x3(h, &h->x_b1c2, relayout(w->block1_conv2, 16, 16, 3, 16, 16), 9 * 16, 16));
    AL_TRY(upload_x3(h, &h->x_b2c1, relayout(w->block2_conv1, 32, 16, 3, 16, 32), 9 * 16, 32));
    AL_TRY(upload_x3(h, &h->x_b2c2, relayout(w->block2_conv2, 32, 32, 3, 32, 32), 9 * 32, 32));
    AL_TRY(upload_x3(h, &h->x_b2ds, relayout(w->block2_ds_w, 32, 16, 1, 16, 32), 16, 32));
  }
  {  // the same operands the fp32 GEMMs use ([K][N] row-major), split for the matrix cores
    auto kn = [&](const float* w_, int co, int ci, int k) { return relayout(w_, co, ci, k, ci, co); };
    AL_TRY(upload_x3g(h, &h->g_b3r1, deform_w(w->block3_reg1, c3, c2).data(), al_deform_krow(c2), c3)); AL_TRY(upload_x3g(h, &h->g_b3r2, deform_w(w->block3_reg2, c3, c3).data(), al_deform_krow(c3), c3));
    AL_TRY(upload_x3g(h, &h->g_b4r1, deform_w(w->block4_reg1, c4, c3).data(), al_deform_krow(c3), c4)); AL_TRY(upload_x3g(h, &h->g_b4r2, deform_w(w->block4_reg2, c4, c4).data(), al_deform_krow(c4), c4));
    AL_TRY(upload_x3g(h, &h->g_b4ds, kn(w->block4_ds_w, c4, c3, 1).data(), c3, c4));
    if (normal) {
      AL_TRY(upload_x3g(h, &h->g_b3ds, kn(w->block3_ds_w, 64, 32, 1).data(), 32, 64));
      AL_TRY(upload_x3g(h, &h->g_hc2, kn(w->conv2, 32, 32, 1).data(), 32, 32)); AL_TRY(upload_x3g(h, &h->g_hc3, kn(w->conv3, 32, 64, 1).data(), 64, 32));
      AL_TRY(upload_x3g(h, &h->g_hc4, kn(w->conv4, 32, 128, 1).data(), 128, 32));
    }
    AL_TRY(upload_x3g(h, &h->g_sf, kn(w->desc_sf, dim, dim, 1).data(), dim, dim));
    AL_TRY(upload_x3g(h, &h->g_agg, w->desc_agg, M * dim, dim));
    std::vector<float> o0((size_t)dim * 9 * M2);
    for (int co = 0; co < M2; ++co)
      for (int k = 0; k < dim * 9; ++k) o0[(size_t)k * M2 + co] = w->desc_off0_w[(size_t)co * dim * 9 + k];
    AL_TRY(upload_x3g(h, &h->g_o0, o0.data(), dim * 9, M2));
  }
  if (normal) {
    const std::vector<float> w1 = relayout(w->conv1, 32, 16, 1, 16, 32), ws0 = relayout(w->score0, 8, 128, 1, 128, 8);
    std::vector<unsigned short> frag(al_assemble_x3_frag_halves());
    std::vector<float> inv1(32);
    al_assemble_x3_prepare(w1.data(), ws0.data(), frag.data(), inv1.data(), &h->asm_inv0);
    AL_TRY(dev_alloc(h, &h->asm_frag, frag.size()));
    AL_TRY(upload(h, &h->asm_inv1, inv1));
    if (hipMemcpy(h->asm_frag, frag.data(), frag.size() * 2, hipMemcpyHostToDevice) != hipSuccess) { dim_set_error("weight upload failed"); dim_aliked_destroy(h); return -1; }
  }
  const int bnc[8] = {c1, c1, c2, c2, c3, c3, c4, c4};
  for (int i = 0; i < 8; ++i) {
    AL_TRY(upload(h, &h->bn_g[i], padvec(w->bn_weight[i], bnc[i], bnc[i])));
    AL_TRY(upload(h, &h->bn_b[i], padvec(w->bn_bias[i], bnc[i], bnc[i])));
  }
  AL_TRY(upload(h, &h->hc1, relayout(w->conv1, G, c1, 1, c1, G))); AL_TRY(upload(h, &h->hc2, relayout(w->conv2, G, c2, 1, c2, G)));
  AL_TRY(upload(h, &h->hc3, relayout(w->conv3, G, c3, 1, c3, G))); AL_TRY(upload(h, &h->hc4, relayout(w->conv4, G, c4, 1, c4, G)));
  AL_TRY(upload(h, &h->sh0, relayout(w->score0, 8, dim, 1, dim, 8))); AL_TRY(upload(h, &h->sh2, relayout(w->score2, 4, 8, 3, 8, 4)));
  AL_TRY(upload(h, &h->sh4, relayout(w->score4, 4, 4, 3, 4, 4))); AL_TRY(upload(h, &h->sh6, relayout(w->score6, 1, 4, 3, 4, 4)));
  {  // SDDH: offset_conv.0 (2M,dim,3,3) -> GEMM operand [ci*9+tap][2M]; offset_conv.2 (2M,2M,1,1) -> [in][out]
    std::vector<float> o0((size_t)dim * 9 * M2);
    for (int co = 0; co < M2; ++co)
      for (int k = 0; k < dim * 9; ++k) o0[(size_t)k * M2 + co] = w->desc_off0_w[(size_t)co * dim * 9 + k];
    AL_TRY(upload(h, &h->dh_o0_w, o0)); AL_TRY(upload(h, &h->dh_o0_b, padvec(w->desc_off0_b, M2, M2)));
    AL_TRY(upload(h, &h->dh_o2_w, relayout(w->desc_off2_w, M2, M2, 1, M2, M2))); AL_TRY(upload(h, &h->dh_o2_b, padvec(w->desc_off2_b, M2, M2)));
    AL_TRY(upload(h, &h->dh_sf, relayout(w->desc_sf, dim, dim, 1, dim, dim)));
    AL_TRY(upload(h, &h->dh_agg, padvec(w->desc_agg, M * dim * dim, M * dim * dim)));  // [p][c][d] == GEMM operand [p*dim+c][d]
  }
  const size_t B = max_batch;
  const size_t Hp = ((size_t)max_h + 31) / 32 * 32, Wp = ((size_t)max_w + 31) / 32 * 32, NP = Hp * Wp, cap = capacity;
  AL_TRY(dev_alloc(h, &h->cols, B * NP / 64 * al_deform_krow(c3)));  // deformed im2col rows: block3 (1/8 res, K = 9 * c3) is the largest
  AL_TRY(dev_alloc(h, &h->P, B * NP * 3)); AL_TRY(dev_alloc(h, &h->raw, B * NP * c1)); AL_TRY(dev_alloc(h, &h->act, B * NP * c1));
  AL_TRY(dev_alloc(h, &h->x1, B * NP * c1)); AL_TRY(dev_alloc(h, &h->p2, B * NP / 4 * c1)); AL_TRY(dev_alloc(h, &h->idn, B * NP / 4 * c2));
  AL_TRY(dev_alloc(h, &h->x2, B * NP / 4 * c2)); AL_TRY(dev_alloc(h, &h->p3, B * NP / 64 * c2)); AL_TRY(dev_alloc(h, &h->off, B * NP / 64 * 20));
  AL_TRY(dev_alloc(h, &h->x3, B * NP / 64 * c3)); AL_TRY(dev_alloc(h, &h->p4, B * NP / 1024 * c3)); AL_TRY(dev_alloc(h, &h->x4, B * NP / 1024 * c4));
  AL_TRY(dev_alloc(h, &h->f2, B * NP / 4 * G)); AL_TRY(dev_alloc(h, &h->f3, B * NP / 64 * G)); AL_TRY(dev_alloc(h, &h->f4, B * NP / 1024 * G));
  AL_TRY(dev_alloc(h, &h->q2, B * NP / 4 * 8)); AL_TRY(dev_alloc(h, &h->q3, B * NP / 64 * 8)); AL_TRY(dev_alloc(h, &h->q4, B * NP / 1024 * 8));
  AL_TRY(dev_alloc(h, &h->s8, B * NP * 8)); AL_TRY(dev_alloc(h, &h->s4a, B * NP * 4));
  AL_TRY(dev_alloc(h, &h->s4b, B * NP * 4)); AL_TRY(dev_alloc(h, &h->score, B * NP)); AL_TRY(dev_alloc(h, &h->nms, B * NP));
  AL_TRY(dev_alloc(h, &h->cand_score, B * NP)); AL_TRY(dev_alloc(h, &h->cand_idx, B * NP)); AL_TRY(dev_alloc(h, &h->rowcount, B * Hp));
  AL_TRY(dev_alloc(h, &h->rowoff, B * Hp)); AL_TRY(dev_alloc(h, &h->ncand, B)); AL_TRY(dev_alloc(h, &h->kpts_px, B * cap * 2));
  AL_TRY(dev_alloc(h, &h->sc_tmp, B * cap)); AL_TRY(dev_alloc(h, &h->kpts_norm, B * cap * 2)); AL_TRY(dev_alloc(h, &h->kscore, B * cap));
  AL_TRY(dev_alloc(h, &h->patches, B * cap * dim * 9)); AL_TRY(dev_alloc(h, &h->hidden, B * cap * M2)); AL_TRY(dev_alloc(h, &h->feats, B * cap * M * dim));
  AL_TRY(dev_alloc(h, &h->feats2, B * cap * M * dim)); AL_TRY(dev_alloc(h, &h->bn_alpha, 2 * B * 128)); AL_TRY(dev_alloc(h, &h->bn_beta, 2 * B * 128));   // two slots: a conv's input and output BatchNorm
  AL_TRY(dev_alloc(h, &h->mean, B)); AL_TRY(dev_alloc(h, &h->thr_eff, B)); AL_TRY(dev_alloc(h, &h->partial, B * 256 * 128 * 2));
  AL_TRY(dev_alloc(h, &h->tile_partial, al_convx3_partial_doubles((int)B, (int)Hp, (int)Wp)));
  h->topk_keys = nullptr;
  if (topk_scratch_keys(max_batch, capacity)) AL_TRY(dev_alloc(h, &h->topk_keys, topk_scratch_keys(max_batch, capacity)));
#undef AL_TRY
  *out = h;
  return 0;
}

int dim_aliked_extract(dim_aliked* h, const float* images_dev, int batch, int H, int W, int in_channels, float* kpts_xy_dev,
                       float* scores_dev, float* desc_dev, int32_t* n_kpts_dev, void* stream) {
  DIM_REQUIRE(h && images_dev && kpts_xy_dev && scores_dev && desc_dev && n_kpts_dev, "dim_aliked_extract: null argument");
  DimTuneScope tune_scope(&h->base);
  DIM_REQUIRE(batch >= 1 && batch <= h->max_batch, "dim_aliked_extract: batch %d outside [1,%d]", batch, h->max_batch);
  DIM_REQUIRE(in_channels == 1 || in_channels == 3, "dim_aliked_extract: in_channels %d (1 or 3)", in_channels);
  DIM_REQUIRE(H >= 16 && W >= 16 && H <= h->max_h && W <= h->max_w, "dim_aliked_extract: image %dx%d outside the handle's %dx%d", H, W, h->max_h, h->max_w);
  hipStream_t s = (hipStream_t)stream;
  const int c1 = h->cfg.c1, c2 = h->cfg.c2, c3 = h->cfg.c3, c4 = h->cfg.c4, dim = h->cfg.dim, G = dim / 4;
  const bool tiny = c1 == 8;   // aliked-t16: fp32 VALU kernels for the small-channel convolutions and 1x1 heads (the matrix-core forms are built for 16 / 32 channels)
  // InputPadder(div 32) (ALN:247-271,646-648)
  const int ph = (((H / 32) + 1) * 32 - H) % 32, pw = (((W / 32) + 1) * 32 - W) % 32;
  const int pad_t = ph / 2, pad_l = pw / 2, Hp = H + ph, Wp = W + pw;
  const int H2 = Hp / 2, W2 = Wp / 2, H8 = Hp / 8, W8 = Wp / 8, H32 = Hp / 32, W32 = Wp / 32;
  const int NP = Hp * Wp, r = h->cfg.nms_radius, cap = h->capacity;
#define AL_RUN(x) do { int rc__ = (x); if (rc__ != 0) return rc__; } while (0)
  // 1x1 convolutions on NHWC maps are plain GEMMs over pixels (weights already [cin][cout])
  // GEMMs: fp16x3 on the matrix cores (gemm_x6.hip) under the default arithmetic, plain fp32 MFMA otherwise
  const bool x3 = dim_precision_mode() == 2;
  const bool x3c = x3 && !tiny;
  unsigned* const sat_al = dim_sat_counter(DIM_SAT_ALIKED);
  auto gemm = [&](GemmArgs& g, const SplitWeights& wx, int nb) -> int {
    if (x3 && g.K % 32 == 0) { g.set_split(wx); g.sat = sat_al; return launch_gemm_x6(g, nb, s); }
    return launch_gemm(g, nb, s);
  };
  auto conv1x1 = [&](const float* in, int ci, const float* w1, const SplitWeights& wx, const float* bias, float* out, int co, int npx, int act) -> int {
    if (tiny && !(ci == 32 && co == 64)) return launch_al_conv1x1(in, ci, w1, bias, out, co, npx, act, s);   // K = 16 / N = 16: below the GEMMs' granules
    GemmArgs g;
    g.A0 = in; g.lda0 = ci; g.B = w1; g.ldb = co; g.bias = bias; g.C = out; g.ldc = co; g.M = npx; g.N = co; g.K = ci;
    g.relu = act == AL_ACT_SELU ? 2 : 0;
    return gemm(g, wx, 1);
  };
  auto bn = [&](const float* x, int npx, int C, int i, const float* res, float* dst) -> int {
    int rc = launch_al_bn_stats(x, batch, npx, C, h->bn_g[i], h->bn_b[i], h->partial, h->bn_alpha, h->bn_beta, s);
    if (rc) return rc;
    return launch_al_bn_apply(x, h->bn_alpha, h->bn_beta, res, dst, batch, npx, C, s);
  };
  AL_RUN(launch_al_pad_replicate(images_dev, h->P, batch, H, W, Hp, Wp, pad_t, pad_l, in_channels, s));
  // Full- and half-resolution convolutions: fp16x3 on the matrix cores with the BatchNorm statistics reduced in the conv
  // epilogue (aliked_x3.hip) under the default arithmetic; the fp32 VALU kernels + a separate statistics pass otherwise
  // (dim_tune_set(1, 0 | 1): A/B checks and the range-guard fallback; aliked-t16).
  // conv -> train-mode BN statistics -> (alpha, beta) of BatchNorm layer i in slot `slot` of bn_alpha / bn_beta.  in_bn >= 0: the
  // input is the previous convolution's RAW output and its BatchNorm + SELU (slot in_bn) is applied while the tile is staged.
  const bool fuse_bn = x3c && dim_aliked_fuse_bn();
  auto ab = [&](int slot, float** a, float** bta) { *a = h->bn_alpha + (size_t)slot * batch * 128; *bta = h->bn_beta + (size_t)slot * batch * 128; };
  auto conv_bn = [&](const float* in, int in_c, int cin_pad, int taps, const SplitWeights& wx, const float* wv, float* out, int co,
                     int Hh, int Ww, int i, int slot, int in_bn) -> int {
    float *a, *bt, *ia = nullptr, *ib = nullptr;
    ab(slot, &a, &bt);
    if (in_bn >= 0) ab(in_bn, &ia, &ib);
    if (x3c) {
      int n_wg = 0, rc;
      if (Hh == Hp) dim_prof_begin(DIM_PROF_AL_CONV_FULL, s);
      if ((rc = launch_al_convx3(in, in_c, cin_pad, taps, wx, nullptr, out, co, batch, Hh, Ww, h->tile_partial, &n_wg, ia, ib, s))) return rc;
      if (Hh == Hp) dim_prof_end(DIM_PROF_AL_CONV_FULL, s);
      return launch_al_bn_final_tiles(h->tile_partial, n_wg, batch, Hh * Ww, co, h->bn_g[i], h->bn_b[i], a, bt, s);
    }
    int rc = launch_al_conv3x3(in, in_c, wv, nullptr, out, co, batch, Hh, Ww, AL_ACT_NONE, 0, 0, Hh, Ww, s);
    if (rc) return rc;
    return launch_al_bn_stats(out, batch, Hh * Ww, co, h->bn_g[i], h->bn_b[i], h->partial, a, bt, s);
  };
  auto apply = [&](const float* x, int slot, const float* res, float* dst, int npx, int C) -> int {
    float *a, *bt;
    ab(slot, &a, &bt);
    return launch_al_bn_apply(x, a, bt, res, dst, batch, npx, C, s);
  };
  // the block's last BatchNorm (+ residual) + SELU together with the average pooling in front of the next block
  auto apply_pool = [&](const float* x, int slot, const float* res, float* dst, float* pooled, int Hh, int Ww, int C, int k) -> int {
    float *a, *bt;
    ab(slot, &a, &bt);
    return launch_al_bn_apply_pool(x, a, bt, res, dst, pooled, batch, Hh, Ww, C, k, s);
  };
  // block1 (ConvBlock, ALN:367-393)
  AL_RUN(conv_bn(h->P, 3, 16, 9, h->x_b1c1, h->b1c1, h->raw, c1, Hp, Wp, 0, 0, -1));
  if (fuse_bn) {
    AL_RUN(conv_bn(h->raw, c1, c1, 9, h->x_b1c2, h->b1c2, h->act, c1, Hp, Wp, 1, 1, 0));   // bn1 + SELU of conv1 in the staging; raw output -> act
    AL_RUN(apply_pool(h->act, 1, nullptr, h->x1, h->p2, Hp, Wp, c1, 2));
  } else {
    AL_RUN(apply(h->raw, 0, nullptr, h->act, NP, c1));
    AL_RUN(conv_bn(h->act, c1, c1, 9, h->x_b1c2, h->b1c2, h->raw, c1, Hp, Wp, 1, 1, -1));
    AL_RUN(apply_pool(h->raw, 1, nullptr, h->x1, h->p2, Hp, Wp, c1, 2));
  }
  // block2 (ResBlock, plain convs); its input p2 = avgpool2(x1) was written by apply_pool
  AL_RUN(conv_bn(h->p2, c1, c1, 9, h->x_b2c1, h->b2c1, h->raw, c2, H2, W2, 2, 0, -1));
  if (x3c) AL_RUN(launch_al_convx3(h->p2, c1, c1, 1, h->x_b2ds, h->b2ds_b, h->idn, c2, batch, H2, W2, nullptr, nullptr, nullptr, nullptr, s));
  else AL_RUN(launch_al_conv1x1(h->p2, c1, h->b2ds_w, h->b2ds_b, h->idn, c2, batch * H2 * W2, AL_ACT_NONE, s));  // K = 16 / 8: below the GEMM's K granule
  if (fuse_bn) {
    AL_RUN(conv_bn(h->raw, c2, c2, 9, h->x_b2c2, h->b2c2, h->act, c2, H2, W2, 3, 1, 0));
    AL_RUN(apply_pool(h->act, 1, h->idn, h->x2, h->p3, H2, W2, c2, 4));
  } else {
    AL_RUN(apply(h->raw, 0, nullptr, h->act, H2 * W2, c2));
    AL_RUN(conv_bn(h->act, c2, c2, 9, h->x_b2c2, h->b2c2, h->raw, c2, H2, W2, 3, 1, -1));
    AL_RUN(apply_pool(h->raw, 1, h->idn, h->x2, h->p3, H2, W2, c2, 4));
  }
  // block3 / block4 (ResBlock with DeformableConv2d, ALN:274-330)
  auto dcn_block = [&](const float* x, int Hh, int Ww, int ci, int co, const float* o1w, const float* o1b, const float* r1,
                       const float* o2w, const float* o2b, const float* r2, const float* dsw, const float* dsb, int bni, float* dst,
                       const SplitWeights& xr1, const SplitWeights& xr2, const SplitWeights& xds) -> int {
    const float lim = (float)(Hh > Ww ? Hh : Ww) / 4.0f;
    int rc;
    if ((rc = launch_al_conv3x3(x, ci, o1w, o1b, h->off, 18, batch, Hh, Ww, AL_ACT_NONE, 0, 0, Hh, Ww, s))) return rc;
    if ((rc = launch_al_clamp(h->off, (size_t)batch * Hh * Ww * 18, lim, s))) return rc;
    if ((rc = launch_al_deform_conv(x, ci, h->off, 18, r1, x3 ? &xr1 : nullptr, sat_al, h->cols, h->raw, co, batch, Hh, Ww, s))) return rc;
    if ((rc = bn(h->raw, Hh * Ww, co, bni, nullptr, h->act))) return rc;
    if ((rc = launch_al_conv3x3(h->act, co, o2w, o2b, h->off, 18, batch, Hh, Ww, AL_ACT_NONE, 0, 0, Hh, Ww, s))) return rc;
    if ((rc = launch_al_clamp(h->off, (size_t)batch * Hh * Ww * 18, lim, s))) return rc;
    if ((rc = launch_al_deform_conv(h->act, co, h->off, 18, r2, x3 ? &xr2 : nullptr, sat_al, h->cols, h->raw, co, batch, Hh, Ww, s))) return rc;
    if ((rc = conv1x1(x, ci, dsw, xds, dsb, h->idn, co, batch * Hh * Ww, AL_ACT_NONE))) return rc;
    return bn(h->raw, Hh * Ww, co, bni + 1, h->idn, dst);
  };
  AL_RUN(dcn_block(h->p3, H8, W8, c2, c3, h->b3o1_w, h->b3o1_b, h->b3r1, h->b3o2_w, h->b3o2_b, h->b3r2, h->b3ds_w, h->b3ds_b, 4, h->x3, h->g_b3r1, h->g_b3r2, h->g_b3ds));
  AL_RUN(launch_al_avgpool(h->x3, h->p4, batch, H8, W8, c3, 4, s));
  AL_RUN(dcn_block(h->p4, H32, W32, c3, c4, h->b4o1_w, h->b4o1_b, h->b4r1, h->b4o2_w, h->b4o2_b, h->b4r2, h->b4ds_w, h->b4ds_b, 6, h->x4, h->g_b4r1, h->g_b4r2, h->g_b4ds));
  // feature aggregation + score head (ALN:656-669)
  if (tiny) {
    AL_RUN(launch_al_conv1x1(h->x2, c2, h->hc2, nullptr, h->f2, G, batch * H2 * W2, AL_ACT_SELU, s));
    AL_RUN(launch_al_conv1x1(h->x3, c3, h->hc3, nullptr, h->f3, G, batch * H8 * W8, AL_ACT_SELU, s));
    AL_RUN(launch_al_conv1x1(h->x4, c4, h->hc4, nullptr, h->f4, G, batch * H32 * W32, AL_ACT_SELU, s));
  } else {
    AL_RUN(conv1x1(h->x2, 32, h->hc2, h->g_hc2, nullptr, h->f2, 32, batch * H2 * W2, AL_ACT_SELU));
    AL_RUN(conv1x1(h->x3, 64, h->hc3, h->g_hc3, nullptr, h->f3, 32, batch * H8 * W8, AL_ACT_SELU));
    AL_RUN(conv1x1(h->x4, 128, h->hc4, h->g_hc4, nullptr, h->f4, 32, batch * H32 * W32, AL_ACT_SELU));
  }
  // s8 only; x1234 stays virtual.  The G -> 8 projections of the three up-sampled groups run at the maps' own resolutions
  AL_RUN(launch_al_conv1x1(h->f2, G, h->sh0 + G * 8, nullptr, h->q2, 8, batch * H2 * W2, AL_ACT_NONE, s));
  AL_RUN(launch_al_conv1x1(h->f3, G, h->sh0 + 2 * G * 8, nullptr, h->q3, 8, batch * H8 * W8, AL_ACT_NONE, s));
  AL_RUN(launch_al_conv1x1(h->f4, G, h->sh0 + 3 * G * 8, nullptr, h->q4, 8, batch * H32 * W32, AL_ACT_NONE, s));
  if (x3c) AL_RUN(launch_al_assemble_x3(h->x1, h->q2, h->q3, h->q4, h->asm_frag, h->asm_inv1, h->asm_inv0, h->s8, batch, Hp, Wp, s));
  else AL_RUN(launch_al_assemble_proj(h->x1, h->q2, h->q3, h->q4, h->hc1, h->sh0, h->s8, c1, batch, Hp, Wp, s));
  const AlFeat F{h->x1, h->f2, h->f3, h->f4, h->hc1, Hp, Wp, c1};
  AL_RUN(launch_al_conv3x3(h->s8, 8, h->sh2, nullptr, h->s4a, 4, batch, Hp, Wp, AL_ACT_SELU, 0, 0, Hp, Wp, s));
  AL_RUN(launch_al_conv3x3(h->s4a, 4, h->sh4, nullptr, h->s4b, 4, batch, Hp, Wp, AL_ACT_SELU, 0, 0, Hp, Wp, s));
  AL_RUN(launch_al_conv3x3(h->s4b, 4, h->sh6, nullptr, h->score, 1, batch, Hp, Wp, AL_ACT_SIGMOID, pad_t, pad_l, H, W, s));  // unpad (ALN:672-673)
  // DKD (ALN:123-244): NMS, border, threshold (mean fallback), n_limit, soft-argmax refinement
  AL_RUN(launch_nms(h->score, h->nms, batch, H, W, r, s));
  AL_RUN(launch_al_mean(h->score, batch, H * W, h->partial, h->mean, s));
  if (h->cfg.detection_threshold > 0) {
    AL_RUN(launch_select_ex(h->nms, batch, H, W, (float)h->cfg.detection_threshold, nullptr, r, h->rowcount, h->rowoff, h->ncand, h->cand_score, h->cand_idx, 1, s));
    AL_RUN(launch_al_pick_threshold(h->ncand, h->mean, (float)h->cfg.detection_threshold, h->thr_eff, batch, s));
    AL_RUN(launch_select_ex(h->nms, batch, H, W, 0.f, h->thr_eff, r, h->rowcount, h->rowoff, h->ncand, h->cand_score, h->cand_idx, 0, s));
  } else if (h->cfg.max_num_keypoints <= 0) {
    // no threshold and no top-k (ALN:161-163): masks = nms_scores > mean score of the image
    AL_RUN(launch_al_pick_threshold(h->ncand, h->mean, 0.f, h->thr_eff, batch, s));     // thr <= 0 -> the mean
    AL_RUN(launch_select_ex(h->nms, batch, H, W, 0.f, h->thr_eff, r, h->rowcount, h->rowoff, h->ncand, h->cand_score, h->cand_idx, 0, s));
  } else {
    // top-k mode (ALN:150-151: topk over the border-cleared NMS map): every maximum is a candidate (the map is zero elsewhere, scores are
    // sigmoids > 0) and launch_topk keeps the max_num_keypoints highest, sorted; with FEWER maxima than that, zero-score pixels fill up
    // (launch_topk_zero_fill: which pixels torch.topk takes among the equal zeros is an artefact of its sort, not a rule)
    AL_RUN(launch_select_ex(h->nms, batch, H, W, 0.f, nullptr, r, h->rowcount, h->rowoff, h->ncand, h->cand_score, h->cand_idx, 0, s));
  }
  const bool topk_mode = !(h->cfg.detection_threshold > 0) && h->cfg.max_num_keypoints > 0;
  // n_limit (ALN:624-626): max_num_keypoints, or n_limit_max = 20000 (ALN:571) in the keep-all modes (bounded by the slot)
  const int n_limit = h->cfg.max_num_keypoints > 0 ? h->cfg.max_num_keypoints : (cap < 20000 ? cap : 20000);
  AL_RUN(launch_topk(h->cand_score, h->cand_idx, h->ncand, batch, H, W, n_limit, cap, h->kpts_px, h->sc_tmp, n_kpts_dev, h->topk_keys, topk_mode ? 1 : 0, s));
  if (topk_mode) AL_RUN(launch_topk_zero_fill(h->nms, batch, H, W, 0.f, r, n_limit, cap, h->kpts_px, h->sc_tmp, n_kpts_dev, s));
  // Q8: DIM's "scores" are the dispersities (ALN:682 unpacks DKD's return in the wrong order)
  AL_RUN(launch_al_dkd_refine(h->score, h->kpts_px, n_kpts_dev, h->kpts_norm, scores_dev, h->kscore, kpts_xy_dev, batch, H, W, cap, r, s));
  // SDDH (ALN:503-558)
  AL_RUN(launch_al_sddh_patches(F, h->kpts_norm, n_kpts_dev, h->patches, batch, H, W, pad_t, pad_l, cap, s));
  const int M = h->cfg.M, M2 = 2 * M, PK = dim * 9;
  {
    GemmArgs g;
    g.A0 = h->patches; g.lda0 = PK; g.strideA0 = (long long)cap * PK; g.B = h->dh_o0_w; g.ldb = M2; g.bias = h->dh_o0_b;
    g.C = h->hidden; g.ldc = M2; g.strideC = (long long)cap * M2; g.M = cap; g.N = M2; g.K = PK; g.rows = n_kpts_dev;
    AL_RUN(gemm(g, h->g_o0, batch));
  }
  AL_RUN(launch_al_sddh_sample(F, h->kpts_norm, n_kpts_dev, h->hidden, h->dh_o2_w, h->dh_o2_b, h->feats, M, batch, H, W, pad_t, pad_l, cap, s));
  {
    GemmArgs g;  // sf_conv 1x1 (dim -> dim) + SELU over the M sampled positions of every keypoint
    g.A0 = h->feats; g.lda0 = dim; g.strideA0 = (long long)cap * M * dim; g.B = h->dh_sf; g.ldb = dim;
    g.C = h->feats2; g.ldc = dim; g.strideC = (long long)cap * M * dim; g.M = cap * M; g.N = dim; g.K = dim;
    g.rows = n_kpts_dev; g.rows_scale = M; g.relu = 2;
    AL_RUN(gemm(g, h->g_sf, batch));
  }
  {
    GemmArgs g;  // einsum("ncp,pcd->nd") with agg_weights [p][c][d] == [n][p*dim+c] x [p*dim+c][d]
    g.A0 = h->feats2; g.lda0 = M * dim; g.strideA0 = (long long)cap * M * dim; g.B = h->dh_agg; g.ldb = dim;
    g.C = desc_dev; g.ldc = dim; g.strideC = (long long)cap * dim; g.M = cap; g.N = dim; g.K = M * dim; g.rows = n_kpts_dev;
    AL_RUN(gemm(g, h->g_agg, batch));
  }
  AL_RUN(launch_al_normalize_rows(desc_dev, n_kpts_dev, batch, cap, dim, s));
#undef AL_RUN
  h->last_hp = Hp; h->last_wp = Wp; h->last_h = H; h->last_w = W; h->last_batch = batch;
  return 0;
}

int dim_aliked_debug_buffers(dim_aliked* h, const float** x1234, const float** score_map, int* hp, int* wp, int* pad_t, int* pad_l) {
  DIM_REQUIRE(h, "dim_aliked_debug_buffers: null handle");
  DimTuneScope tune_scope(&h->base);
  if (x1234) {  // the product path never stores the 128-channel map: rebuild it for the last batch
    DIM_REQUIRE(h->last_batch > 0, "dim_aliked_debug_buffers: no extract call yet");
    if (h->dbg_x1234) hipFree(h->dbg_x1234);
    h->dbg_x1234 = nullptr;
    DIM_HIP(hipMalloc((void**)&h->dbg_x1234, (size_t)h->last_batch * h->last_hp * h->last_wp * h->cfg.dim * sizeof(float)));
    if (launch_al_assemble(h->x1, h->f2, h->f3, h->f4, h->hc1, h->sh0, h->dbg_x1234, h->s8, h->cfg.c1, h->last_batch, h->last_hp, h->last_wp, nullptr)) return -1;
    DIM_HIP(hipDeviceSynchronize());
    *x1234 = h->dbg_x1234;
  }
  if (score_map) *score_map = h->score;
  if (hp) *hp = h->last_hp;
  if (wp) *wp = h->last_wp;
  if (pad_t) *pad_t = (h->last_hp - h->last_h) / 2;
  if (pad_l) *pad_l = (h->last_wp - h->last_w) / 2;
  return 0;
}

}  // extern "C"
