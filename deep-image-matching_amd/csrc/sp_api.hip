// dim_sp_* : resident SuperPoint extractor (C ABI in include/dim_hip.h).
// Replaces SuperPoint.__init__/forward (SPN:101-227) as driven by
// SuperPointExtractor._extract (extractors/superpoint.py:107-132).
//
// HBM layout: every activation is NHWC fp32 ([batch][H][W][C]); 1x1 convolutions
// are therefore plain GEMMs over pixels; the descriptor head's output stays
// un-normalised in HBM and is normalised only at the <= 4 cells each keypoint
// touches (sample_desc_kernel), which is algebraically the reference's
// normalise-then-sample (SPN:215,218-221).
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include <vector>

#include "../../include/dim_hip.h"
#include "sp_kernels.h"

struct dim_sp {
  DimHandleBase base;   // first member: dim_handle_tune_set
  dim_sp_config cfg;
  int max_batch, max_h, max_w, capacity;
  // weights (device)
  float* w1a; float* wk[12]; float* bias[12];
  SplitWeights wsw[12];     // Winograd F(2,3)-transformed fp16x3 weights (conv_wg.hip) of the layers that have the variant (conv1b)
  SplitWeights wsp[3][12];  // [precision mode 1 = bf16x6, 2 = fp16x3] pre-split 3x3 weights (conv_x6.hip); empty for conv1a and the 1x1 layers
  // activations
  float *a1, *b1, *a2, *b2, *a3, *b3, *a4, *x, *pa, *logits, *da, *dd, *smap, *nms, *cand_score;
  int *cand_idx, *rowcount, *rowoff, *ncand;
  unsigned long long* topk_keys;   // launch_topk's global key table (max_keypoints > 4096 only)
  int last_h, last_w, last_batch;
  float conv1a_bound; // max over channels of sum|w1a| + |b1a|: bound on conv1a's outputs for |image| <= 1 (fp16x3 range guard)
  bool x_is_planes;   // the last extract stored the encoder output as pre-split planes
  bool head_fused;    // the last extract ran convPb + softmax + depth-to-space as one kernel: h->logits is stale
  float* b1_dbg;      // fp32 copy of conv1b's pooled output (dim_sp_debug_conv1b)
  float* x_dbg;       // fp32 copy of it, built on request by dim_sp_debug_buffers
  std::vector<void*> allocs;
};

namespace {
const int kCin[12] = {1, 64, 64, 64, 64, 128, 128, 128, 128, 256, 128, 256};
const int kCout[12] = {64, 64, 64, 64, 128, 128, 128, 128, 256, 65, 256, 256};
const int kK[12] = {3, 3, 3, 3, 3, 3, 3, 3, 3, 1, 3, 1};

template <typename T>
int dev_alloc(dim_sp* h, T** p, size_t count) {
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
}  // namespace

extern "C" {

void dim_sp_destroy(dim_sp* h) {
  if (!h) return;
  for (void* p : h->allocs) hipFree(p);
  delete h;
}

int dim_sp_create(const dim_sp_weights* w, const dim_sp_config* cfg, int max_batch, int max_h, int max_w, int capacity,
                  dim_sp** out) {
  DIM_REQUIRE(w && cfg && out, "dim_sp_create: null argument");
  DIM_REQUIRE(max_batch > 0 && max_h >= 8 && max_w >= 8, "dim_sp_create: bad sizes");
  DIM_REQUIRE(cfg->max_keypoints != 0 && cfg->max_keypoints >= -1, "\"max_keypoints\" must be positive or \"-1\"");  // SPN:152-154
  DIM_REQUIRE(capacity > 0 && capacity >= cfg->max_keypoints, "dim_sp_create: capacity %d < max_keypoints %d", capacity, cfg->max_keypoints);
  DIM_REQUIRE(cfg->nms_radius >= 0, "nms_radius must be >= 0");  // SPN:49
  dim_sp* h = new dim_sp();
  memset((void*)&h->cfg, 0, sizeof(h->cfg));
  h->cfg = *cfg;
  h->max_batch = max_batch; h->max_h = max_h; h->max_w = max_w; h->capacity = capacity;
  h->last_h = h->last_w = h->last_batch = 0;
  h->b1_dbg = nullptr; h->x_dbg = nullptr; h->head_fused = false;
#define SP_TRY(x) do { if ((x) != 0) { dim_sp_destroy(h); return -1; } } while (0)
  // ---- weights: OIHW (SPN:128-143) -> [tap][cin][cout] / [cin][cout_padded4] ----
  for (int l = 0; l < 12; ++l) {
    const int ci = kCin[l], co = kCout[l], k = kK[l];
    const int co_pad = (co + 3) & ~3;
    if (!dim_all_finite(w->conv_w[l], (size_t)k * k * ci * co) || !dim_all_finite(w->conv_b[l], (size_t)co)) { dim_set_error("dim_sp_create: non-finite value in the weights of layer %d", l); dim_sp_destroy(h); return -1; }
    std::vector<float> host((size_t)k * k * ci * co_pad, 0.0f);
    for (int o = 0; o < co; ++o)
      for (int i = 0; i < ci; ++i)
        for (int t = 0; t < k * k; ++t) host[((size_t)t * ci + i) * co_pad + o] = w->conv_w[l][((size_t)o * ci + i) * k * k + t];
    if (k == 3 && ci >= 64) {
      for (int mode = 1; mode <= 2; ++mode) {
        std::vector<unsigned short> hx(conv_split_weight_elems(ci, co, mode));
        SplitWeights& sw = h->wsp[mode][l];
        prepare_conv_weights_split(w->conv_w[l], ci, co, mode, hx.data(), &sw);
        unsigned short* d = nullptr;
        SP_TRY(dev_alloc(h, &d, hx.size()));
        if (hipMemcpy(d, hx.data(), hx.size() * 2, hipMemcpyHostToDevice) != hipSuccess) { dim_set_error("weight upload failed"); dim_sp_destroy(h); return -1; }
        sw.dev = d; sw.mode = mode;
      }
    }
#ifdef DIM_RESEARCH
    if (l == 1) {  // conv1b: the Winograd variant's weights (dim_tune_set key 15; research build)
      std::vector<unsigned short> hx(conv_wino_weight_elems(ci, co));
      SplitWeights& sw = h->wsw[l];
      prepare_conv_weights_wino(w->conv_w[l], ci, co, hx.data(), &sw);
      unsigned short* d = nullptr;
      SP_TRY(dev_alloc(h, &d, hx.size()));
      if (hipMemcpy(d, hx.data(), hx.size() * 2, hipMemcpyHostToDevice) != hipSuccess) { dim_set_error("weight upload failed"); dim_sp_destroy(h); return -1; }
      sw.dev = d; sw.mode = 2;
    }
#endif
    if (k == 1) {  // the two 1x1 heads (convPb 256 -> 65, convDb 256 -> 256) run on the split GEMM
      std::vector<float> kn((size_t)ci * co);
      for (int o = 0; o < co; ++o)
        for (int i = 0; i < ci; ++i) kn[(size_t)i * co + o] = w->conv_w[l][(size_t)o * ci + i];
      const int n_pad = (co + 127) / 128 * 128;
      for (int mode = 1; mode <= 2; ++mode) {
        std::vector<unsigned short> hx(gemm_split_weight_elems(ci, n_pad, mode));
        SplitWeights& sw = h->wsp[mode][l];
        split_weights(kn.data(), ci, co, n_pad, mode, hx.data(), &sw);
        unsigned short* d = nullptr;
        SP_TRY(dev_alloc(h, &d, hx.size()));
        if (hipMemcpy(d, hx.data(), hx.size() * 2, hipMemcpyHostToDevice) != hipSuccess) { dim_set_error("weight upload failed"); dim_sp_destroy(h); return -1; }
        sw.dev = d; sw.mode = mode; sw.n_pad = n_pad;
      }
    }
    SP_TRY(dev_alloc(h, &h->wk[l], host.size()));
    if (hipMemcpy(h->wk[l], host.data(), host.size() * sizeof(float), hipMemcpyHostToDevice) != hipSuccess) { dim_set_error("weight upload failed"); dim_sp_destroy(h); return -1; }
    std::vector<float> hb(co_pad, 0.0f);
    memcpy(hb.data(), w->conv_b[l], co * sizeof(float));
    SP_TRY(dev_alloc(h, &h->bias[l], hb.size()));
    if (hipMemcpy(h->bias[l], hb.data(), hb.size() * sizeof(float), hipMemcpyHostToDevice) != hipSuccess) { dim_set_error("bias upload failed"); dim_sp_destroy(h); return -1; }
  }
  h->conv1a_bound = 0.0f;
  for (int o = 0; o < 64; ++o) {
    float sum = fabsf(w->conv_b[0][o]);
    for (int t = 0; t < 9; ++t) sum += fabsf(w->conv_w[0][o * 9 + t]);
    h->conv1a_bound = fmaxf(h->conv1a_bound, sum);
  }
  // ---- activations ----
  const size_t B = max_batch, H = max_h, W = max_w;
  const size_t H2 = H / 2, W2 = W / 2, H4 = H2 / 2, W4 = W2 / 2, hh = H4 / 2, ww = W4 / 2;
  h->a1 = nullptr;  // conv1a's 64-channel full-resolution map: only the unfused A/B path needs it, allocated on first use
  SP_TRY(dev_alloc(h, &h->b1, B * dim_planes_image_pixels(H2, W2) * 64));
  SP_TRY(dev_alloc(h, &h->a2, B * dim_planes_image_pixels(H2, W2) * 64));
  SP_TRY(dev_alloc(h, &h->b2, B * dim_planes_image_pixels(H4, W4) * 64));
  SP_TRY(dev_alloc(h, &h->a3, B * dim_planes_image_pixels(H4, W4) * 128));
  SP_TRY(dev_alloc(h, &h->b3, B * dim_planes_image_pixels(hh, ww) * 128));
  SP_TRY(dev_alloc(h, &h->a4, B * dim_planes_image_pixels(hh, ww) * 128));
  SP_TRY(dev_alloc(h, &h->x, B * dim_planes_image_pixels(hh, ww) * 128));
  SP_TRY(dev_alloc(h, &h->pa, B * hh * ww * 256));
  SP_TRY(dev_alloc(h, &h->logits, B * hh * ww * 65));
  SP_TRY(dev_alloc(h, &h->da, B * hh * ww * 256));
  SP_TRY(dev_alloc(h, &h->dd, B * hh * ww * 256));
  SP_TRY(dev_alloc(h, &h->smap, B * hh * ww * 64));
  SP_TRY(dev_alloc(h, &h->nms, B * hh * ww * 64));
  SP_TRY(dev_alloc(h, &h->cand_score, B * hh * ww * 64));
  SP_TRY(dev_alloc(h, &h->cand_idx, B * hh * ww * 64));
  SP_TRY(dev_alloc(h, &h->rowcount, B * hh * 8));
  SP_TRY(dev_alloc(h, &h->rowoff, B * hh * 8));
  SP_TRY(dev_alloc(h, &h->ncand, B));
  h->topk_keys = nullptr;
  if (topk_scratch_keys(max_batch, cfg->max_keypoints)) SP_TRY(dev_alloc(h, &h->topk_keys, topk_scratch_keys(max_batch, cfg->max_keypoints)));
#undef SP_TRY
  *out = h;
  return 0;
}

int dim_sp_extract(dim_sp* h, const float* images_dev, int batch, int H, int W, float* kpts_xy_dev, float* scores_dev,
                   float* desc_dev, int32_t* n_kpts_dev, void* stream) {
  DIM_REQUIRE(h && images_dev && kpts_xy_dev && scores_dev && desc_dev && n_kpts_dev, "dim_sp_extract: null argument");
  DimTuneScope tune_scope(&h->base);
  DIM_REQUIRE(batch >= 1 && batch <= h->max_batch, "dim_sp_extract: batch %d outside [1,%d]", batch, h->max_batch);
  DIM_REQUIRE(H >= 8 && W >= 8 && H <= h->max_h && W <= h->max_w && (size_t)H * W <= (size_t)h->max_h * h->max_w,
              "dim_sp_extract: image %dx%d outside the handle's %dx%d", H, W, h->max_h, h->max_w);
  hipStream_t s = (hipStream_t)stream;
  const int H2 = H / 2, W2 = W / 2, H4 = H2 / 2, W4 = W2 / 2, hh = H4 / 2, ww = W4 / 2;  // floor at every pool (Q12)
  const int H8 = hh * 8, W8 = ww * 8;
#define SP_RUN(x) do { int rc__ = (x); if (rc__ != 0) return rc__; } while (0)
#define SP_SITE(id, x) do { dim_prof_begin(id, s); SP_RUN(x); dim_prof_end(id, s); } while (0)
  const int pmode = dim_precision_mode();  // 2 (default) fp16x3 / 1 bf16x6: fp32-accurate products on the 16-bit matrix cores; 0: fp32 MFMA
  const bool x6 = pmode != 0;
  // fp16x3 range guard (dim_common.h): producers of values that a later split consumes report max|x| > 4094
  unsigned* sat_enc = pmode == 2 ? dim_sat_counter(DIM_SAT_SP_ENCODER) : nullptr;
  unsigned* sat_head = pmode == 2 ? dim_sat_counter(DIM_SAT_SP_HEADS) : nullptr;
  unsigned* sat_img = pmode == 2 ? dim_sat_counter(DIM_SAT_SP_IMAGE) : nullptr;
  if (pmode == 2 && !(h->conv1a_bound <= DIM_F16_ACT_LIMIT)) dim_sat_host_bump(DIM_SAT_SP_IMAGE);  // conv1a's outputs may leave the range
  auto conv = [&](int l, const float* in, float* out, int Hh, int Ww, int ci, int co, int pool) -> int {
    return x6 ? launch_conv3x3_x6(in, h->wsp[pmode][l], h->bias[l], out, batch, Hh, Ww, ci, co, pool, 1, s, l >= 8 ? sat_head : sat_enc)
              : launch_conv3x3(in, h->wk[l], h->bias[l], out, batch, Hh, Ww, ci, co, pool, 1, s);
  };
  // encoder (SPN:161-171)
  // fp16x3 + fused conv1a: the conv-to-conv activations b1 .. a4 are stored as pre-split fp16 planes (conv_x6.hip PIN / POUT):
  // each value is split once by its producer instead of ~1.3 x (cout / 64) times by its consumers
  const bool planes = pmode == 2 && dim_fuse_conv1a() && dim_presplit_activations();
  auto convp = [&](int l, const float* in, float* out, int Hh, int Ww, int ci, int co, int pool, int pin, int pout) -> int {
    return launch_conv3x3_x6_planes(in, h->wsp[2][l], h->bias[l], out, batch, Hh, Ww, ci, co, pool, 1, pin, pout, s, l >= 8 ? sat_head : sat_enc);
  };
#ifdef DIM_RESEARCH
  if (pmode == 2 && dim_fuse_conv1a() && (dim_conv_winograd() & 1)) {  // Winograd F(2,3) along x: 2/3 of the MFMAs (conv_wg.hip)
    // the transformed activations reach 2 x conv1a's output bound: check the doubled bound on the host as the direct path checks the plain one
    if (!(2.0f * h->conv1a_bound <= DIM_F16_ACT_LIMIT)) dim_sat_host_bump(DIM_SAT_SP_IMAGE);
    SP_SITE(DIM_PROF_SP_CONV1B, launch_conv3x3_wg_fused1a(images_dev, h->wk[0], h->bias[0], h->wsw[1], h->bias[1], h->b1, batch, H, W, 64, 1, planes ? 1 : 0, s, sat_enc, sat_img));
  } else
#endif
  if (x6 && dim_fuse_conv1a()) {  // conv1a evaluated inside conv1b's halo staging: its 64-channel full-resolution map never exists
    SP_SITE(DIM_PROF_SP_CONV1B, launch_conv3x3_x6_fused1a(images_dev, h->wk[0], h->bias[0], h->wsp[pmode][1], h->bias[1], h->b1, batch, H, W, 64, 1, 1, planes ? 1 : 0, s, sat_enc, sat_img));
  } else {
    if (!h->a1) SP_RUN(dev_alloc(h, &h->a1, (size_t)h->max_batch * h->max_h * h->max_w * 64));
    SP_SITE(DIM_PROF_SP_CONV1A, launch_conv1a(images_dev, h->wk[0], h->bias[0], h->a1, batch, H, W, s));
    SP_SITE(DIM_PROF_SP_CONV1B, conv(1, h->a1, h->b1, H, W, 64, 64, 1));
  }
  if (planes) {
    SP_SITE(DIM_PROF_SP_CONV2A, convp(2, h->b1, h->a2, H2, W2, 64, 64, 0, 1, 1));
    SP_SITE(DIM_PROF_SP_CONV2B, convp(3, h->a2, h->b2, H2, W2, 64, 64, 1, 1, 1));
    SP_SITE(DIM_PROF_SP_CONV3A, convp(4, h->b2, h->a3, H4, W4, 64, 128, 0, 1, 1));
    SP_SITE(DIM_PROF_SP_CONV3B, convp(5, h->a3, h->b3, H4, W4, 128, 128, 1, 1, 1));
    SP_SITE(DIM_PROF_SP_CONV4A, convp(6, h->b3, h->a4, hh, ww, 128, 128, 0, 1, 1));
    SP_SITE(DIM_PROF_SP_CONV4B, convp(7, h->a4, h->x, hh, ww, 128, 128, 0, 1, 1));  // the encoder output feeds 8 cout blocks of the two heads
  } else {
    SP_SITE(DIM_PROF_SP_CONV2A, conv(2, h->b1, h->a2, H2, W2, 64, 64, 0));
    SP_SITE(DIM_PROF_SP_CONV2B, conv(3, h->a2, h->b2, H2, W2, 64, 64, 1));
    SP_SITE(DIM_PROF_SP_CONV3A, conv(4, h->b2, h->a3, H4, W4, 64, 128, 0));
    SP_SITE(DIM_PROF_SP_CONV3B, conv(5, h->a3, h->b3, H4, W4, 128, 128, 1));
    SP_SITE(DIM_PROF_SP_CONV4A, conv(6, h->b3, h->a4, hh, ww, 128, 128, 0));
    SP_SITE(DIM_PROF_SP_CONV4B, conv(7, h->a4, h->x, hh, ww, 128, 128, 0));
  }
  // detector head (SPN:174-180)
  h->x_is_planes = planes;
  if (planes) SP_SITE(DIM_PROF_SP_CONVPA, convp(8, h->x, h->pa, hh, ww, 128, 256, 0, 1, 0));
  else SP_SITE(DIM_PROF_SP_CONVPA, conv(8, h->x, h->pa, hh, ww, 128, 256, 0));
  {
    GemmArgs g;
    g.A0 = h->pa; g.lda0 = 256; g.B = h->wk[9]; g.ldb = 68; g.bias = h->bias[9];
    g.C = h->logits; g.ldc = 65; g.M = batch * hh * ww; g.N = 65; g.K = 256;
    // fp16x3: convPb + softmax + depth-to-space in one kernel (gemm_x6_head_kernel): the logits are not stored (dim_sp_debug_buffers rebuilds them)
    h->head_fused = pmode == 2 && dim_fuse_sp_head();
    if (h->head_fused) {
      g.set_split(h->wsp[pmode][9]); g.d2s_out = h->smap; g.d2s_h = hh; g.d2s_w = ww;
      SP_RUN(launch_gemm_x6(g, 1, s));
    } else {
      if (x6) { g.set_split(h->wsp[pmode][9]); SP_RUN(launch_gemm_x6(g, 1, s)); }
      else SP_RUN(launch_gemm(g, 1, s));
      SP_RUN(launch_softmax_d2s(h->logits, h->smap, batch, hh, ww, s));
    }
  }
  SP_RUN(launch_nms(h->smap, h->nms, batch, H8, W8, h->cfg.nms_radius, s));
  // selection (SPN:183-210)
  SP_RUN(launch_select(h->nms, batch, H8, W8, h->cfg.keypoint_threshold, h->cfg.remove_borders, h->rowcount, h->rowoff,
                       h->ncand, h->cand_score, h->cand_idx, s));
  SP_RUN(launch_topk(h->cand_score, h->cand_idx, h->ncand, batch, H8, W8, h->cfg.max_keypoints, h->capacity, kpts_xy_dev,
                     scores_dev, n_kpts_dev, h->topk_keys, 0, s));
  // descriptor head (SPN:213-221)
  if (planes) SP_SITE(DIM_PROF_SP_CONVDA, convp(10, h->x, h->da, hh, ww, 128, 256, 0, 1, 0));
  else SP_SITE(DIM_PROF_SP_CONVDA, conv(10, h->x, h->da, hh, ww, 128, 256, 0));
  {
    GemmArgs g;
    g.A0 = h->da; g.lda0 = 256; g.B = h->wk[11]; g.ldb = 256; g.bias = h->bias[11];
    g.C = h->dd; g.ldc = 256; g.M = batch * hh * ww; g.N = 256; g.K = 256;
    if (x6) { g.set_split(h->wsp[pmode][11]); SP_RUN(launch_gemm_x6(g, 1, s)); }
    else SP_RUN(launch_gemm(g, 1, s));
  }
  SP_RUN(launch_sample_desc(h->dd, kpts_xy_dev, n_kpts_dev, desc_dev, batch, hh, ww, h->capacity, h->cfg.fix_sampling, s));
#undef SP_SITE
#undef SP_RUN
  h->last_h = hh; h->last_w = ww; h->last_batch = batch;
  return 0;
}

int dim_sp_debug_buffers(dim_sp* h, const float** encoder, const float** logits, const float** score_map,
                         const float** nms_map, const float** dense_desc, int* h8, int* w8) {
  DIM_REQUIRE(h, "dim_sp_debug_buffers: null handle");
  DimTuneScope tune_scope(&h->base);
  if (encoder) {
    *encoder = h->x;
    if (h->x_is_planes) {  // rebuild fp32 from the planes of the last batch
      if (!h->x_dbg && dev_alloc(h, &h->x_dbg, (size_t)h->max_batch * (h->max_h / 8) * (h->max_w / 8) * 128) != 0) return -1;
      if (launch_planes_to_f32(h->x, h->last_batch, h->last_h * h->last_w, 128, h->x_dbg, nullptr) != 0) return -1;
      DIM_HIP(hipDeviceSynchronize());
      *encoder = h->x_dbg;
    }
  }
  if (logits) {
    if (h->head_fused && h->last_batch > 0) {   // the fused detector tail never stores the logits: rebuild them with the plain GEMM on the last batch's convPa output
      GemmArgs g;
      g.A0 = h->pa; g.lda0 = 256; g.B = h->wk[9]; g.ldb = 68; g.bias = h->bias[9];
      g.C = h->logits; g.ldc = 65; g.M = h->last_batch * h->last_h * h->last_w; g.N = 65; g.K = 256;
      g.set_split(h->wsp[2][9]);
      if (launch_gemm_x6(g, 1, nullptr) != 0) return -1;
      DIM_HIP(hipDeviceSynchronize());
    }
    *logits = h->logits;
  }
  if (score_map) *score_map = h->smap;
  if (nms_map) *nms_map = h->nms;
  if (dense_desc) *dense_desc = h->dd;
  if (h8) *h8 = h->last_h * 8;
  if (w8) *w8 = h->last_w * 8;
  return 0;
}

int dim_sp_debug_conv1b(dim_sp* h, int batch, int H, int W, const float** out_f32, int* h2, int* w2) {
  // fp32 NHWC copy [batch][H/2][W/2][64] of conv1b's pooled output of the last extract (A/B of the convolution variants)
  DIM_REQUIRE(h && out_f32 && batch >= 1 && batch <= h->max_batch, "dim_sp_debug_conv1b: bad argument");
  DimTuneScope tune_scope(&h->base);
  const int H2 = H / 2, W2 = W / 2;
  if (!h->b1_dbg && dev_alloc(h, &h->b1_dbg, (size_t)h->max_batch * (h->max_h / 2) * (h->max_w / 2) * 64) != 0) return -1;
  const bool planes = dim_precision_mode() == 2 && dim_fuse_conv1a() && dim_presplit_activations();
  if (planes) {
    if (launch_planes_to_f32(h->b1, batch, H2 * W2, 64, h->b1_dbg, nullptr) != 0) return -1;
  } else {
    DIM_HIP(hipMemcpy(h->b1_dbg, h->b1, (size_t)batch * H2 * W2 * 64 * sizeof(float), hipMemcpyDeviceToDevice));
  }
  DIM_HIP(hipDeviceSynchronize());
  *out_f32 = h->b1_dbg;
  if (h2) *h2 = H2;
  if (w2) *w2 = W2;
  return 0;
}

int dim_sp_candidate_counts(dim_sp* h, const int32_t** ncand_dev) {
  DIM_REQUIRE(h && ncand_dev, "dim_sp_candidate_counts: null argument");
  *ncand_dev = h->ncand;
  return 0;
}

}  // extern "C"
