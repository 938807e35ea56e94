/* dim_hip.h — C ABI of libdim_hip.so, the MI355X (gfx950) implementation of
 * deep-image-matching's per-pair hot path (SuperPoint extraction + LightGlue
 * matching).  Plain pointers and sizes only; no torch / numpy types.
 *
 * Conventions
 *   - every pointer named *_dev / documented "device" is HBM memory of the
 *     current HIP device; `stream` is a hipStream_t passed as void* (NULL = the
 *     default stream).  No entry point synchronises unless it says so.
 *   - return value 0 = success; non-zero = failure, dim_last_error() returns a
 *     thread-local message (the Python plugin turns it into an exception; an
 *     allocation failure message contains "out of memory" so that the caller's
 *     tile fallback keyed on that substring keeps working —
 *     reference matchers/matcher_base.py:251-256).
 *
 * Reference interfaces replaced (paths relative to the reference repo root,
 * src/deep_image_matching/...):
 *   dim_sp_*  <-  thirdparty/SuperGluePretrainedNetwork/models/superpoint.py:101-227
 *                 (SuperPoint.__init__/forward) as driven by
 *                 extractors/superpoint.py:107-132 (SuperPointExtractor._extract)
 *   dim_lg_*  <-  thirdparty/LightGlue/lightglue/lightglue.py:300-610
 *                 (LightGlue.__init__/forward) as driven by
 *                 matchers/lightglue.py:102-125 (LightGlueMatcher._match_pairs)
 */
#ifndef DIM_HIP_H
#define DIM_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DIM_HIP_ABI_VERSION 2   /* 2 (round 6): handle structs carry a tune header, dim_tune_set keys 16 - 18, keypoint slots up to 32768, dim_lg_stage_features, the dim_op_* sort entries */

const char* dim_last_error(void);
int dim_abi_version(void);
int dim_device_synchronize(void);
/* Process-wide defaults of the choices a user may make (read at the entry of every later dim_* call; set them before the work starts —
 * they are plain ints, not synchronised against calls running on other threads):
 *   key 1  matrix arithmetic: 2 (default) "fp16x3" = fp32-accurate products on the fp16 matrix cores (2-way splits of the power-of-two-scaled
 *          operands x 3 terms; activations exact up to |x| = 4094 and GUARDED beyond, dim_saturation_read), 1 = "bf16x6" (exact 3-way bf16
 *          splits x 6 terms, no range limit), 0 = plain fp32 MFMA;
 *   fusion on / off (every setting gives the same results to fp32 rounding; the defaults are the fast ones):
 *   key 3  1 (default) SuperPoint conv1a evaluated inside conv1b, 0 = separate kernels;
 *   key 4  1 (default) LightGlue out_proj folded into ffn.0 (split modes), 0 = separate GEMMs;
 *   key 5  1 (default) SuperPoint conv-to-conv activations stored pre-split (fp16x3), 0 = fp32;
 *   key 8  1 (default) LightGlue's K | V attention tile images written by the projection GEMM, 0 = separate pre-split pass;
 *   key 9  1 (default) ALIKED BatchNorm + SELU applied while the consumer stages its input, 0 = separate pass;
 *   key 16 1 (default) SuperPoint's detector tail (convPb 256 -> 65, softmax, depth-to-space) as one kernel in fp16x3, 0 = GEMM + softmax kernels
 *          (bit-identical score maps);
 *   key 17 1 (default) adaptive-depth LightGlue (depth_confidence > 0) evaluates the assignment once, after the layer loop, for every pair with the
 *          weights of the layer it stopped at; 0 = gated assignment launches after every layer (same results);
 *   key 18 1 (default) dim_lg_match with adaptive depth on a handle created for at most two pairs (the per-call plugin hooks) reads the stop
 *          flags back two layers behind the device (an event wait per layer) and does not enqueue the layers that no pair needs; 0 = the call
 *          never touches the host, as calls on larger handles always do (same results);
 *   key 11 LightGlue's feed-forward: 3 (default) ffn.0 + LayerNorm + GELU + ffn.3 + residual as one kernel when the launch fills the GPU,
 *          4 = always, 1 / 2 = LayerNorm + GELU in ffn.0's epilogue only (when large / always), 0 = separate kernels;
 *   kernel-shape selection (same results; the defaults pick by problem size, the other values force a shape — used by the tests to reach
 *   the large-batch kernels with small inputs):
 *   key 0  fp32 conv3x3 kernel variant; key 2  split-precision conv variant (bits 0-1 prefetch variant of the bf16x6 kernels, bit 4 (default
 *          set) 16-row tiles with LDS-DMA weight staging); key 6  split-precision GEMM block (1 default: 128 x 256 when the launch fills the
 *          GPU, 0 = always 128 x 128, 2 = always 128 x 256); key 7  simple_nms tiles (1 default: 64 x 64 on large maps, 0 = 32 x 32,
 *          2 = always 64 x 64); key 10  ALIKED fp16x3 convolution tile rows (16 default, 8, 17 = 16 with streamed weights).
 * Keys 12-15 do not exist in this library: they select research prototypes and timing probes that are compiled only into
 * libdim_hip_research.so (build.build_variant("research", ["-DDIM_RESEARCH"]); csrc/dim_kernels.h) and return an error here.
 * dim_handle_tune_set(handle, key, value) overrides a key for ONE extractor / matcher handle (value < 0: back to the process default). */
int dim_tune_set(int key, int value);
int dim_handle_tune_set(void* handle, int key, int value);

/* Per-launch-site timing with HIP events recorded on the launch stream (bench.py's
 * roofline figure).  dim_profile_start(mask) arms the sites whose bits (1 << DIM_PROF_*) are set;
 * dim_profile_stop() waits for the recorded events and returns the summed kernel time and the
 * number of launches. */
enum {
  DIM_PROF_SP_CONV1A = 0, DIM_PROF_SP_CONV1B, DIM_PROF_SP_CONV2A, DIM_PROF_SP_CONV2B, DIM_PROF_SP_CONV3A,
  DIM_PROF_SP_CONV3B, DIM_PROF_SP_CONV4A, DIM_PROF_SP_CONV4B, DIM_PROF_SP_CONVPA, DIM_PROF_SP_CONVPB,
  DIM_PROF_SP_CONVDA, DIM_PROF_SP_CONVDB, DIM_PROF_SP_POST, DIM_PROF_LG_SELF_ATTN, DIM_PROF_LG_CROSS_ATTN,
  DIM_PROF_LG_GEMMS, DIM_PROF_LG_ASSIGN,
  DIM_PROF_AL_CONV_FULL   /* ALIKED's full-resolution 3x3 convolutions (block1.conv1 3 -> 16, block1.conv2 16 -> 16): HBM-bound */
};
int dim_profile_start(unsigned long long site_mask);
int dim_profile_stop(double* total_ms, int* launches);

/* fp16x3 range guard.  The default arithmetic represents every activation as two fp16 pieces of 16*x, exact for
 * |x| <= 4094 and saturating beyond.  Every kernel that produces a value a later split consumes checks max|x| of
 * what it writes and bumps a sticky device counter per launch site when the range is exceeded (the results of that
 * call are then NOT fp32-accurate).  dim_saturation_read synchronises `stream`, copies the DIM_SAT_SITES counters to
 * counts_host (may be NULL), returns their sum in *total and zeroes them when reset != 0.  The Python plugins call it
 * after every extract / match and re-run the call in bf16x6 (no range limit) when the sum is non-zero. */
enum {
  DIM_SAT_SP_IMAGE = 0,   /* |image| > 1 or conv1a's weight bound exceeds the range (fused conv1a+conv1b input) */
  DIM_SAT_SP_ENCODER,     /* outputs of conv1b .. conv4b */
  DIM_SAT_SP_HEADS,       /* outputs of convPa / convDa (inputs of the 1x1 head GEMMs) */
  DIM_SAT_LG_INPUT,       /* descriptors handed to dim_lg_match / input_proj output */
  DIM_SAT_LG_QKV,         /* Wqkv / to_qk|to_v outputs incl. the rotated q, k */
  DIM_SAT_LG_FFN,         /* ffn.0 output after LayerNorm+GELU */
  DIM_SAT_LG_DESC,        /* residual stream after ffn.3 */
  DIM_SAT_OP,             /* operator-level entry points (dim_op_*_x6) */
  DIM_SAT_ALIKED,         /* ALIKED: inputs of the split-precision convolutions / GEMMs (image, BatchNorm+SELU outputs, SDDH samples) */
  DIM_SAT_SITES = 16
};
int dim_saturation_read(unsigned* counts_host, unsigned long long* total, int reset, void* stream);
/* Zeroes the counters without reading them: one memset enqueued on `stream`, no synchronisation (what a guarded call does before it starts, to drop
 * anything an earlier unguarded call left behind). */
int dim_saturation_reset(void* stream);

/* Shader-clock probe: writes {s_memtime (shader cycles), s_memrealtime (100 MHz)} to out_dev[2] on `stream`;
 * two probes around a region give its average shader clock = d(cycles) / d(realtime) * 100 MHz (bench.py). */
int dim_op_read_clocks(unsigned long long* out_dev, void* stream);

/* ------------------------------------------------------------------------ */
/* SuperPoint (reference SPN:101-227)                                       */
/* ------------------------------------------------------------------------ */

/* Weights in the reference's own state_dict layout (SPN:128-143): conv weights
 * OIHW fp32, host pointers; they are re-laid out for the kernels at create. */
typedef struct dim_sp_weights {
  const float* conv_w[12]; /* conv1a,1b,2a,2b,3a,3b,4a,4b,convPa,convPb,convDa,convDb */
  const float* conv_b[12];
} dim_sp_weights;

/* Mirrors SuperPoint.default_config (SPN:112-118) + DIM's fix_sampling switch
 * (extractors/superpoint.py:16-27,56-57). */
typedef struct dim_sp_config {
  int nms_radius;           /* >= 0 */
  float keypoint_threshold; /* s > thr */
  int max_keypoints;        /* -1 = keep all (bounded by capacity); top-k up to 32768 (config/superpoint+superglue.yaml: 8000) */
  int remove_borders;
  int fix_sampling;         /* 0: SPN:81-98 sampler, 1: extractors/superpoint.py:16-27 */
} dim_sp_config;

typedef struct dim_sp dim_sp;

/* Builds a resident extractor for images up to max_h x max_w, max_batch images
 * per call, at most capacity keypoints per image (capacity >= max_keypoints
 * when that is >= 0). */
int dim_sp_create(const dim_sp_weights* w, const dim_sp_config* cfg, int max_batch, int max_h, int max_w,
                  int capacity, dim_sp** out);
void dim_sp_destroy(dim_sp* h);

/* images_dev: [batch][H][W] fp32, values already divided by 255
 *             (extractors/superpoint.py:134-146 _frame2tensor).
 * Outputs (device, caller allocated, slot b at offset b*capacity):
 *   kpts_xy_dev  [batch][capacity][2] fp32  (x, y) pixel coordinates (SPN:210)
 *   scores_dev   [batch][capacity]    fp32
 *   desc_dev     [batch][capacity][256] fp32, row-major (N, D) — the transpose of
 *                the reference's (D, N); the Python plugin returns the .T view
 *   n_kpts_dev   [batch] int32
 * Order of keypoints: score-descending when more than max_keypoints survive
 * (torch.topk, SPN:74-78), row-major (y, x) otherwise (SPN:183-186). */
int dim_sp_extract(dim_sp* h, const float* images_dev, int batch, int H, int W, float* kpts_xy_dev, float* scores_dev,
                   float* desc_dev, int32_t* n_kpts_dev, void* stream);

/* Debug/parity taps of the last dim_sp_extract call (device pointers owned by
 * the handle; valid until the next call).  Layouts: encoder [batch][h][w][128],
 * logits [batch][h*w][65], score_map / nms_map [batch][8h][8w],
 * dense_desc (un-normalised convDb output) [batch][h][w][256]. */
int dim_sp_debug_buffers(dim_sp* h, const float** encoder, const float** logits, const float** score_map,
                         const float** nms_map, const float** dense_desc, int* h8, int* w8);

/* fp32 NHWC copy [batch][H/2][W/2][64] of conv1b's pooled output (SPN:162-163) of the last dim_sp_extract call on (batch, H, W):
 * A/B of the convolution variants (dim_tune_set keys 2, 3, 5).  Synchronises the device. */
int dim_sp_debug_conv1b(dim_sp* h, int batch, int H, int W, const float** out_f32, int* h2, int* w2);

/* Number of NMS survivors above threshold/border per image of the last call
 * (before top-k), device int32 [batch] owned by the handle. */
int dim_sp_candidate_counts(dim_sp* h, const int32_t** ncand_dev);

/* ------------------------------------------------------------------------ */
/* ALIKED (reference ALN = thirdparty/LightGlue/lightglue/aliked.py:561-693, driven by
 * extractors/aliked.py:45-64)                                               */
/* ------------------------------------------------------------------------ */

/* Learnable tensors in the reference's state_dict layout (conv weights OIHW, host pointers;
 * SURVEY.md Appendix A).  BatchNorm running statistics are NOT part of the interface: the
 * reference plugin never calls .eval(), so BatchNorm normalises with the statistics of the
 * current image (quirk Q7, ALX:40-43) and this library does the same. */
typedef struct dim_aliked_weights {
  const float *block1_conv1, *block1_conv2;            /* (16,3,3,3), (16,16,3,3) */
  const float *block2_conv1, *block2_conv2, *block2_ds_w, *block2_ds_b;
  const float *block3_off1_w, *block3_off1_b, *block3_reg1, *block3_off2_w, *block3_off2_b, *block3_reg2, *block3_ds_w, *block3_ds_b;
  const float *block4_off1_w, *block4_off1_b, *block4_reg1, *block4_off2_w, *block4_off2_b, *block4_reg2, *block4_ds_w, *block4_ds_b;
  const float *bn_weight[8], *bn_bias[8];             /* block1.bn1, block1.bn2, block2.bn1, ..., block4.bn2 */
  const float *conv1, *conv2, *conv3, *conv4;          /* 1x1 heads (32,c,1,1) */
  const float *score0, *score2, *score4, *score6;      /* score_head.{0,2,4,6}.weight */
  const float *desc_off0_w, *desc_off0_b, *desc_off2_w, *desc_off2_b, *desc_sf, *desc_agg; /* desc_head.* */
} dim_aliked_weights;

/* ALIKED._default_conf (ALN:562-567) + the geometry row of ALIKED.cfgs (ALN:573-579). */
typedef struct dim_aliked_config {
  int c1, c2, c3, c4, dim, K, M;   /* ALN:573-579: aliked-n16 / n16rot 16,32,64,128,128,3,16; aliked-n32 the same with M = 32; aliked-t16 8,16,32,64,64,3,16 (desc_dev rows are dim floats) */
  int max_num_keypoints;           /* n_limit of DKD (ALN:624-626), up to 32768 (config/aliked.yaml: 8000); <= 0 = n_limit_max 20000 (ALN:571), bounded by capacity */
  double detection_threshold;      /* > 0: threshold mode; <= 0: DKD's top-k mode (exactly max_num_keypoints per image, zero-score pixels filling up,
                                      ALN:150-151) or, with max_num_keypoints <= 0 as well, the mean-score threshold (ALN:161-163) */
  int nms_radius;
} dim_aliked_config;

typedef struct dim_aliked dim_aliked;
int dim_aliked_create(const dim_aliked_weights* w, const dim_aliked_config* cfg, int max_batch, int max_h, int max_w,
                      int capacity, dim_aliked** out);
void dim_aliked_destroy(dim_aliked* h);
/* images_dev: [batch][H][W][in_channels] fp32 (HWC, as the numpy image arrives; already /255,
 *             extractors/aliked.py:66-78), in_channels 3 (RGB) or 1 (repeated to RGB, ALN:679-680).
 * Outputs as dim_sp_extract with D = cfg.dim (128; aliked-t16: 64): kpts (x, y) sub-pixel pixel coordinates (ALN:687),
 * scores = DIM's "scores" i.e. the score DISPERSITIES (quirk Q8), desc row-major (N, dim). */
int dim_aliked_extract(dim_aliked* h, const float* images_dev, int batch, int H, int W, int in_channels, float* kpts_xy_dev,
                       float* scores_dev, float* desc_dev, int32_t* n_kpts_dev, void* stream);
/* Parity taps: un-normalised feature map [batch][Hp][Wp][dim] in the padded frame, score map [batch][H][W]. */
int dim_aliked_debug_buffers(dim_aliked* h, const float** x1234, const float** score_map, int* hp, int* wp, int* pad_t, int* pad_l);

/* ------------------------------------------------------------------------ */
/* LightGlue (reference LGN:300-610)                                        */
/* ------------------------------------------------------------------------ */

/* Per-layer tensors in the reference's state_dict layout (nn.Linear weight
 * [out][in], LGN:361-378), host pointers. */
typedef struct dim_lg_layer_weights {
  const float *self_Wqkv_w, *self_Wqkv_b;   /* transformers.i.self_attn.Wqkv   (768,256) rows = head*192+dim*3+{q,k,v} */
  const float *self_out_w, *self_out_b;     /* transformers.i.self_attn.out_proj (256,256) */
  const float *self_ffn0_w, *self_ffn0_b;   /* ...self_attn.ffn.0 (512,512) */
  const float *self_ln_w, *self_ln_b;       /* ...self_attn.ffn.1 LayerNorm(512) */
  const float *self_ffn3_w, *self_ffn3_b;   /* ...self_attn.ffn.3 (256,512) */
  const float *cross_qk_w, *cross_qk_b;     /* transformers.i.cross_attn.to_qk (256,256) */
  const float *cross_v_w, *cross_v_b;       /* ...to_v */
  const float *cross_out_w, *cross_out_b;   /* ...to_out */
  const float *cross_ffn0_w, *cross_ffn0_b, *cross_ln_w, *cross_ln_b, *cross_ffn3_w, *cross_ffn3_b;
  const float *assign_match_w, *assign_match_b; /* log_assignment.i.matchability (1,256),(1,) */
  const float *assign_proj_w, *assign_proj_b;   /* log_assignment.i.final_proj (256,256) */
  const float *token_w, *token_b;               /* token_confidence.i.token.0 (1,256),(1,); NULL for the last layer */
} dim_lg_layer_weights;

typedef struct dim_lg_weights {
  int n_layers;                       /* 9 */
  int input_dim;                      /* 256 (SuperPoint) or 128 (ALIKED/DISK/SIFT) */
  const float *input_proj_w, *input_proj_b; /* (256,input_dim),(256,) or NULL when input_dim == 256 (LGN:361-364) */
  const float* posenc_Wr;             /* posenc.Wr.weight (32,2) */
  const float* confidence_thresholds; /* buffer (n_layers,) (LGN:581-584) */
  const dim_lg_layer_weights* layers; /* [n_layers] */
} dim_lg_weights;

/* LightGlue._default_conf (LGN:301-314); doubles because the reference compares fp32
 * tensors against Python floats.  pruning_min_kpts: LGN:318-323,606-610 (-1 = the CPU
 * path this library is parity-checked against). */
typedef struct dim_lg_config {
  double depth_confidence; /* early stop, disable with -1 */
  double width_confidence; /* point pruning, disable with -1 */
  double filter_threshold;
  int pruning_min_kpts;
} dim_lg_config;

typedef struct dim_lg dim_lg;

int dim_lg_create(const dim_lg_weights* w, const dim_lg_config* cfg, int max_pairs, int max_kpts, dim_lg** out);
void dim_lg_destroy(dim_lg* h);
/* Row stride (>= max_kpts, multiple of 4) of the per-point output arrays below. */
int dim_lg_max_kpts(dim_lg* h);

/* The conversion half of featuresDict2Lightglue (matchers/lightglue.py:18-64: descriptors (D, N) -> (N, D), :38-43, and
 * torch.as_tensor(v, dtype=torch.float32, device=device), :62) on the device, for ONE pair whose arrays were uploaded exactly as
 * features.h5 holds them (save_features_h5 with as_half, extractors/extractor_base.py:56-99: float16; SuperPoint / ALIKED write their
 * descriptors as (D, N), extractors/superpoint.py:121-127): fills rows [0, n) of slots 0 and 1 of a feature table
 * kpts_tab_dev [2][cap][2], desc_tab_dev [2][cap][D] (fp32, the layout dim_lg_match reads) and zeroes rows [n, cap).  fp16 -> fp32 is exact,
 * so the table equals what the host conversion produced.  Keypoints 4-byte, descriptors 16-byte aligned; D a multiple of 32. */
typedef struct dim_lg_raw_features {
  const void* kpts_dev;   /* (N, 2) row-major, float32 or float16 */
  const void* desc_dev;   /* (N, D) or (D, N) row-major, float32 or float16 */
  int n;                  /* keypoints (<= cap) */
  int kpts_f16, desc_f16; /* 1: float16, 0: float32 */
  int desc_is_dn;         /* 1: (D, N), 0: (N, D) */
} dim_lg_raw_features;
int dim_lg_stage_features(const dim_lg_raw_features* img0, const dim_lg_raw_features* img1, int cap, int D, float* kpts_tab_dev, float* desc_tab_dev,
                          void* stream);

/* Matches n_pairs pairs in one call.  Features come from a device feature table
 * (slot i = image i, as written by dim_sp_extract):
 *   kpts_tab_dev [n_img][cap][2], desc_tab_dev [n_img][cap][input_dim] (row-major (N,D):
 *   what featuresDict2Lightglue produces, matchers/lightglue.py:38-43), n_tab_dev [n_img],
 *   size_tab_dev [n_img][2] = image_size exactly as DIM feeds it ((H, W)).
 * pair_idx_dev [n_pairs][2] int32 = image slots of (image0, image1); NULL = pair p is
 * slots (2p, 2p+1).  Keypoint counts above the handle's max_kpts are truncated.
 * Outputs (device, caller allocated; NK = dim_lg_max_kpts(h)):
 *   matches_dev   [n_pairs][NK][2] int64 — compact (idx0, idx1) list, idx0 ascending (LGN:543-553)
 *   mscores_dev   [n_pairs][NK]
 *   n_matches_dev [n_pairs]
 *   matches01_dev [n_pairs][2][NK] int32 — matches0 / matches1 (-1 = unmatched)
 *   mscores01_dev [n_pairs][2][NK]       — matching_scores0 / matching_scores1
 *   stop_dev      [n_pairs]              — "stop" (LGN:570)
 *   prune01_dev   [n_pairs][2][NK] int32 — prune0 / prune1
 *   dense_scores_dev: NULL, or [n_pairs][NK+1][NK+1] receiving the inner MxN block of the
 *   log assignment matrix (parity tests only).
 * The call only enqueues work on `stream` and never reads results back — with one exception: on a handle created for at most two pairs, with
 * depth_confidence > 0 (adaptive depth), it waits (spinning on mapped page-locked memory, two layers behind the device) for the pairs' stop flags and
 * stops enqueueing layers once every pair has stopped (dim_tune_set key 18 = 0 turns that off; results are identical either way).  A handle must not be
 * used from two threads at once. */
int dim_lg_match(dim_lg* h, const float* kpts_tab_dev, const float* desc_tab_dev, const int32_t* n_tab_dev,
                 const float* size_tab_dev, int cap, const int32_t* pair_idx_dev, int n_pairs, int64_t* matches_dev,
                 float* mscores_dev, int32_t* n_matches_dev, int32_t* matches01_dev, float* mscores01_dev,
                 int32_t* stop_dev, int32_t* prune01_dev, float* dense_scores_dev, void* stream);

/* Parity taps: final descriptors [2*max_pairs][NK][256], live counts and index maps. */
int dim_lg_debug_desc(dim_lg* h, const float** desc, const int32_t** n_cur, const int32_t** ind);

/* ------------------------------------------------------------------------ */
/* operator-level entry points (each is one kernel launch; used by the      */
/* parity tests and available to integrators)                               */
/* ------------------------------------------------------------------------ */

/* C[M][N] = act(A[M][K] * B + bias) (+ residual); B is [K][N] (ldb) or, when
 * b_is_nk != 0, [N][K] (ldb).  K % 32 == 0, leading dims % 4 == 0. */
int dim_op_gemm_f32(const float* A, int lda, const float* B, int ldb, int b_is_nk, const float* bias,
                    const float* residual, int ldr, float* C, int ldc, int M, int N, int K, int relu, void* stream);

/* fp32-accurate GEMM on the 16-bit matrix cores (csrc/gemm_x6.hip).  dim_x3_create pre-splits a host [K][N] fp32
 * operand for the split mode that is active at the call (dim_tune_set key 1: fp16x3 or bf16x6) and returns an OPAQUE
 * handle (host struct owning the device planes [planes][n_pad][K]); dim_op_gemm_x6_f32 computes
 * C = act(A*W + bias) (+ residual), act 0 none / 1 ReLU / 2 SELU, in the handle's mode.  K % 32 == 0. */
int dim_x3_create(const float* w_kn_host, int K, int N, void** out_handle, int* n_pad_out);
void dim_x3_destroy(void* handle);
int dim_op_gemm_x6_f32(const float* A, int lda, const void* w_x3_handle, int n_pad, const float* bias, const float* residual, int ldr,
                       float* C, int ldc, int M, int N, int K, int act, void* stream);

/* LightGlue's ffn.0 -> LayerNorm(512, eps 1e-5) -> erf-GELU (LGN:141-142,157-158) as ONE kernel: C[M][512] =
 * gelu(layer_norm(A[M][K] * W + bias)); w_x3_handle from dim_x3_create(K, 512) under the default arithmetic. */
int dim_op_gemm_x6_ln_gelu_f32(const float* A, int lda, const void* w_x3_handle, const float* bias, const float* ln_gamma, const float* ln_beta,
                               float* C, int ldc, int M, int K, void* stream);

/* C[M][N] = A[M][K] * B[N][K]^T with both operands fp32 activations (LightGlue's similarity, LGN:271), on the 16-bit matrix cores in the
 * active split arithmetic (|A|, |B| <= 4094 under fp16x3: the callers' producers are range-guarded).  K % 32 == 0, lda / ldb % 4 == 0. */
int dim_op_gemm_x6_nt_f32(const float* A, int lda, const float* B, int ldb, float* C, int ldc, int M, int N, int K, void* stream);

/* LightGlue's whole feed-forward (LGN:141-142,159,209) as ONE kernel: C[M][256] = residual + gelu(layer_norm(A[M][K] * W0 + bias0)) *
 * W3 + bias3, the 512-wide hidden tensor kept on the compute unit.  w0 from dim_x3_create(K, 512); w3 from dim_x3_create_kperm(512,
 * 256) (the same split, rows of every 16-step stored in the order the kernel's register-resident operand presents them). */
int dim_x3_create_kperm(const float* w_kn_host, int K, int N, void** out_handle, int* n_pad_out);
int dim_op_ffn_fused_f32(const float* A, int lda, const void* w0_x3_handle, const float* bias0, const float* ln_gamma, const float* ln_beta,
                         const void* w3_x3_kperm_handle, const float* bias3, const float* residual, int ldr, float* C, int ldc, int M, int K,
                         void* stream);

/* 3x3/s1/p1 conv, NHWC fp32, weights [9][cin][cout], bias+ReLU and optional
 * 2x2 max-pool fused (SPN:161-171).  cin in {64,128}, cout % 64 == 0. */
int dim_op_conv3x3_nhwc_f32(const float* in, const float* w_tap_cin_cout, const float* bias, float* out, int batch,
                            int H, int W, int cin, int cout, int pool2x2, int relu, void* stream);

/* simple_nms (SPN:47-63) on [batch][H][W] score maps, radius 0..6; non-maxima -> 0.  Scores must be >= +0 (what the reference feeds it: softmax
 * outputs; ALIKED: sigmoids) — the kernel keeps a per-pixel mark in the sign bit. */
int dim_op_simple_nms_f32(const float* score_map, float* out, int batch, int H, int W, int radius, void* stream);

/* The same convolution on the 16-bit matrix cores at fp32 accuracy (csrc/conv_x6.hip): weights are pre-split from
 * the reference's OIHW fp32 layout by dim_convx6_create for the active split mode (opaque handle, free with
 * dim_x3_destroy). */
int dim_convx6_create(const float* w_oihw_host, int cin, int cout, void** out_handle);
int dim_op_conv3x3_x6_nhwc_f32(const float* in, const void* w_x6_handle, const float* bias, float* out, int batch, int H, int W,
                               int cin, int cout, int pool2x2, int relu, void* stream);

/* conv1a: [batch][H][W] -> [batch][H][W][64], weights [9][64], bias, ReLU. */
int dim_op_conv1a_f32(const float* in, const float* w_tap_cout, const float* bias, float* out, int batch, int H, int W,
                      void* stream);

/* ---- tile preselection on device (csrc/tile_ops.hip) ------------------------------------------------
 * Replaces the host steps of tile_selection's PRESELECTION branch (matchers/matcher_base.py:1054-1133):
 * cv2.resize(img, size, interpolation=cv2.INTER_AREA) of a single-channel fp32 image (MB:1068-1069; the
 * OpenCV 4.11 decimation table restated; an enlarging size takes OpenCV's bilinear emulation of INTER_AREA),
 * optionally followed by frame2tensor's
 * /255 (MB:1393-1398); and the vote count "matches whose two end points fall into tile t0 of image 0 and
 * tile t1 of image 1" (points_in_rect / get_tile_bounding_box, MB:1124-1133, strict inequalities).
 * matches is dim_lg_match's (S,2) int64 list with its device-side count; keypoints are the down-sampled
 * images' (x,y) and are divided by scale0/scale1 in fp32 as numpy does; origins are (x,y) int32 pairs;
 * votes[T0*T1] is zeroed by the call. */
int dim_op_resize_area_f32(const float* src, int H, int W, float* dst, int h, int w, int div255, void* stream);
/* The tail of _extract_by_tile (extractors/extractor_base.py:330-390) on the device: per-tile tables kpts [n_tiles][cap][2], scores
 * [n_tiles][cap], desc [n_tiles][cap][D], live counts n_tab [n_tiles] (device), tile origins (x, y) int32 and the tile numbers as
 * floats (tile_ids, the values of "tile_idx") -> keypoints shifted to image coordinates, those within 2 px of the (image_h,
 * image_w) border dropped, concatenated in tile order and, with select_unique, sorted + de-duplicated exactly like
 * np.unique(kpts, axis=0, return_index=True) (lexicographic by (x, y), first occurrence kept).  Outputs (device, sized for
 * n_tiles * cap rows): out_kpts [N][2], out_scores [N], out_tile_idx [N], out_desc = the contiguous (D, N) array in the first
 * D * N floats, *n_out = N.  workspace: dim_op_merge_tiles_workspace_bytes(n_tiles, cap) bytes of device memory. */
size_t dim_op_merge_tiles_workspace_bytes(int n_tiles, int cap);
int dim_op_merge_tiles(const float* kpts_tab, const float* scores_tab, const float* desc_tab, const int32_t* n_tab, const int32_t* origins_xy,
                       const float* tile_ids, int n_tiles, int cap, int D, int image_h, int image_w, int select_unique, void* workspace,
                       float* out_kpts, float* out_scores, float* out_tile_idx, float* out_desc, int32_t* n_out, void* stream);
/* Integer bookkeeping of tile-wise matching on the device (csrc/sort_ops.hip) — get_features_by_tile's boolean masks (matchers/matcher_base.py:1380-1391)
 * and _match_by_tile's np.unique(matches, axis=0) (:452-459) without host-language sort / unique calls:
 *  (ld_*: row strides in floats — the three columns may be views of one packed [n][4 + D] table)
 *  dim_op_tile_counts      counts[t] = keypoints of the merged table with tile_idx == t (tile_idx: the float values of "tile_idx");
 *  dim_op_group_by_tile    the merged table (tile_idx [n], kpts [n][2], desc [n][D]) -> per-tile tables: tile t goes to table row row_of_tile[t]
 *                          (-1: not needed), its keypoints in their ORIGINAL order (= the boolean-mask order) into kt [rows][cap][2], dt [rows][cap][D],
 *                          it [rows][cap] int64 (index in the merged table), nt [rows] live counts.  The tables must arrive zeroed.
 *  dim_op_tile_match_keys  dim_lg_match's lists of a batch of tile pairs (matches [n_pairs][nk][2] int64, n_matches) -> keys [n_pairs][nk] =
 *                          slot << 40 | it[row0][m0] << 20 | it[row1][m1] (pair_rows [n_pairs][2] = the two table rows, slot = the image pair), ~0 for dead rows;
 *  dim_op_unique_match_rows  n keys -> per slot the UNIQUE (idx0, idx1) rows in lexicographic order (np.unique(axis=0)): rows [n_slots][cap_m][2]
 *                          (int32, or int64 with rows_are_i64), cnt[slot] = min(rows of the slot, cap_m), n_full[slot] (optional) = before that cut. */
int dim_op_tile_counts(const float* tile_idx_dev, int ld_tile, int n, int n_tiles, int32_t* counts_dev, void* stream);
size_t dim_op_group_by_tile_workspace_bytes(int n, int n_tiles);
int dim_op_group_by_tile(const float* tile_idx_dev, int ld_tile, const float* kpts_dev, int ld_kpts, const float* desc_nd_dev, int ld_desc, int n, int D,
                         const int32_t* row_of_tile_dev, int n_tiles, int cap, float* kt_dev, float* dt_dev, long long* it_dev, int32_t* nt_dev, void* workspace, void* stream);
int dim_op_tile_match_keys(const long long* matches_dev, const int32_t* n_matches_dev, const long long* it_dev, const int32_t* pair_rows_dev, const int32_t* slot_dev,
                           int n_pairs, int nk, int cap, unsigned long long* keys_dev, void* stream);
size_t dim_op_unique_match_rows_workspace_bytes(long long n);
int dim_op_unique_match_rows(const unsigned long long* keys_dev, long long n, int n_slots, int cap_m, int rows_are_i64, void* rows_dev, int32_t* cnt_dev,
                             int32_t* n_full_dev, void* workspace, void* stream);
/* Tile slicing of _extract_by_tile (extractors/extractor_base.py:279-328) on the device: image_dev [H][W][C] fp32 as the numpy array
 * arrived (0..255), origins (x, y) int32 per tile (negative / overhanging = the Tiler's zero padding) ->
 * out_dev [n_tiles][tile_h][tile_w][C], optionally / 255 (_frame2tensor). */
int dim_op_gather_tiles_f32(const float* image_dev, int H, int W, int C, const int32_t* origins_xy_dev, int n_tiles, int tile_h, int tile_w,
                            float* out_dev, int div255, void* stream);
/* cv2.resize(img, size, interpolation=cv2.INTER_LINEAR) (pixel-centre bilinear, OpenCV 4.11 restated): what
 * utils/image.py:52-57 resize_image uses when the target size enlarges the image — quality HIGHEST in
 * extractor_base.py:392-412 and tile_selection (matcher_base.py:1026-1034). */
int dim_op_resize_linear_f32(const float* src, int H, int W, float* dst, int h, int w, int div255, void* stream);
int dim_op_tile_pair_votes(const float* kpts0_xy, const float* kpts1_xy, const long long* matches, const int* n_matches_dev,
                           int max_matches, float scale0, float scale1, const int* origins0_xy, int T0, const int* origins1_xy, int T1,
                           int tile_w, int tile_h, int* votes, void* stream);

/* ---- writers off the critical path (csrc/export_ops.hip) -------------------------------------------------------
 * The device half of save_features_h5 (extractors/extractor_base.py:56-99) for a batch of dim_sp_extract / dim_aliked_extract
 * tables: float32 -> float16 (round-to-nearest-even, overflow -> inf: numpy's astype(float16), EB:60-67), descriptors
 * (N, D) -> (D, N) (extractors/superpoint.py:121-127), un-padding to the live counts.  out_f16_dev receives
 * [batch][slot] fp16 with slot = dim_pack_features_slot_halves(cap, D) = cap * (4 + D) elements:
 *   [0, 2n) keypoints (n,2) | [2cap, 2cap+n) scores | [3cap, 3cap+n) tile_idx (zeros when tile_idx_dev is NULL) |
 *   [4cap, 4cap + D n) descriptors (D, n) with row stride n;  n = min(n_kpts[b], cap).  D % 64 == 0.
 * One device-to-host copy of the slots is then the byte image of the datasets features.h5 stores. */
size_t dim_pack_features_slot_halves(int cap, int D);
int dim_op_pack_features_f16(const float* kpts_dev, const float* scores_dev, const float* desc_dev, const int32_t* n_kpts_dev,
                             const int32_t* tile_idx_dev, int batch, int cap, int D, void* out_f16_dev, void* stream);
/* The accept / reject rules that follow the estimator in MatcherBase.match (matchers/matcher_base.py:287-334) on the device
 * tables of dim_lg_match + dim_gv_fundamental: a pair with fewer than 8 raw matches, fewer than min_inliers inliers or an
 * inlier ratio below min_ratio (fp64 quotient, as Python evaluates it) gets n_verified = -1; otherwise verified_dev
 * [n_pairs][nk][2] receives the inlier rows in order and n_verified their count. */
int dim_op_filter_matches(const int64_t* matches_dev, const int32_t* n_matches_dev, const unsigned char* mask_dev, int nk, int n_pairs,
                          int min_inliers, double min_ratio, int64_t* verified_dev, int32_t* n_verified_dev, void* stream);

/* End-of-job exchange of the per-rank match tables (SURVEY §8(e) phase 4; no reference counterpart — the reference is single
 * process): dim_lg_match's (S,2) int64 lists + scores -> [n_pairs][nk][3] int32 rows (idx0, idx1, score bits), zero beyond
 * n_matches, ready for ONE all-gather of a flat int32 buffer; and back to int64 / fp32 tables, reading row src_of_pair[p]
 * of the gathered buffer for output pair p (NULL = identity), which undoes the round-robin shard order. */
int dim_op_pack_match_rows(const int64_t* matches_dev, const float* scores_dev, const int32_t* n_matches_dev, int nk, int n_pairs,
                           int32_t* rows_dev, void* stream);
int dim_op_unpack_match_rows(const int32_t* rows_dev, const int32_t* src_of_pair_dev, int nk, int n_pairs, int64_t* matches_dev,
                             float* scores_dev, void* stream);

/* ---- retrieval pair selection (csrc/tile_ops.hip) --------------------------------------------------------------
 * thirdparty/hloc/pairs_from_retrieval.py:49-70,108-112: sim = einsum("id,jd->ij", query, db) (global descriptors,
 * fp32 MFMA), invalid entries (self matches; score < min_score when use_min_score) -> -inf, torch.topk(num_select) per
 * query row.  indices [nq][num_select] int32 in descending score order (ties: lowest index), -1 where fewer finite
 * entries exist; values likewise.  sim_scratch: nq * nd floats.  dim must be a multiple of 32 (zero-pad otherwise). */
int dim_op_retrieval_topk(const float* query_dev, int nq, const float* db_dev, int nd, int dim, const unsigned char* invalid_dev, int num_select,
                          float min_score, int use_min_score, float* sim_scratch_dev, int* indices_dev, float* values_dev, void* stream);

/* ---- geometric verification on device (csrc/geom_verify.hip) ----------------------------------------------
 * Replaces the per-pair host call geometric_verification(kpts0[matches[:,0]], kpts1[matches[:,1]], method, threshold,
 * confidence) that follows _match_pairs in the reference (utils/geometric_verification.py:45-179, called at
 * matchers/matcher_base.py:311): a batched fundamental-matrix RANSAC straight on dim_lg_match's device outputs.
 * Inputs: the feature table's keypoints [n_img][cap][2], pair_idx [n_pairs][2] (NULL = slots 2p, 2p+1), matches
 * [n_pairs][nk][2] int64 + n_matches [n_pairs] exactly as dim_lg_match wrote them (nk <= 4096).
 * threshold_px as the reference passes it (gv_threshold x quality scale); iters = number of 7-point hypotheses (the
 * reference's max_iters is 10000); error_type 0 = Sampson distance (USAC / pydegensac family), 1 = symmetric
 * epipolar distance (max of the two point-line distances, cv2.RANSAC); seed makes the sampling reproducible.
 * Outputs: inlier_mask [n_pairs][nk] uint8 (0 beyond n_matches), n_inliers [n_pairs], F [n_pairs][9] fp64 row-major
 * acting on pixel coordinates (x1^T F x0 = 0, scaled to F33 = 1 when possible; zeros when the pair has < 8 matches, in
 * which case every match is an inlier as in geometric_verification.py:107-110).
 * scratch: dim_gv_scratch_bytes(n_pairs) bytes of device memory.  The estimator is deterministic and restated in
 * numpy by oracle/geom_ref.py; it is NOT result-identical to cv2's MAGSAC (no two RANSACs are). */
size_t dim_gv_scratch_bytes(int n_pairs);
int dim_gv_fundamental(const float* kpts_tab_dev, int cap, const int32_t* pair_idx_dev, const int64_t* matches_dev,
                       const int32_t* n_matches_dev, int nk, int n_pairs, double threshold_px, int iters, int error_type, unsigned seed,
                       void* scratch_dev, size_t scratch_bytes, unsigned char* inlier_mask_dev, int32_t* n_inliers_dev, double* F_dev,
                       void* stream);

#ifdef __cplusplus
}
#endif
#endif /* DIM_HIP_H */
