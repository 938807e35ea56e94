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

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DIM_HIP_ABI_VERSION 1

const char* dim_last_error(void);
int dim_abi_version(void);
int dim_device_synchronize(void);

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
  int max_keypoints;        /* -1 = keep all (bounded by capacity) */
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

/* Number of NMS survivors above threshold/border per image of the last call
 * (before top-k), device int32 [batch] owned by the handle. */
int dim_sp_candidate_counts(dim_sp* h, const int32_t** ncand_dev);

/* ------------------------------------------------------------------------ */
/* operator-level entry points (each is one kernel launch; used by the      */
/* parity tests and available to integrators)                               */
/* ------------------------------------------------------------------------ */

/* C[M][N] = act(A[M][K] * B + bias) (+ residual); B is [K][N] (ldb) or, when
 * b_is_nk != 0, [N][K] (ldb).  K % 32 == 0, leading dims % 4 == 0. */
int dim_op_gemm_f32(const float* A, int lda, const float* B, int ldb, int b_is_nk, const float* bias,
                    const float* residual, int ldr, float* C, int ldc, int M, int N, int K, int relu, void* stream);

/* 3x3/s1/p1 conv, NHWC fp32, weights [9][cin][cout], bias+ReLU and optional
 * 2x2 max-pool fused (SPN:161-171).  cin in {64,128}, cout % 64 == 0. */
int dim_op_conv3x3_nhwc_f32(const float* in, const float* w_tap_cin_cout, const float* bias, float* out, int batch,
                            int H, int W, int cin, int cout, int pool2x2, int relu, void* stream);

/* conv1a: [batch][H][W] -> [batch][H][W][64], weights [9][64], bias, ReLU. */
int dim_op_conv1a_f32(const float* in, const float* w_tap_cout, const float* bias, float* out, int batch, int H, int W,
                      void* stream);

#ifdef __cplusplus
}
#endif
#endif /* DIM_HIP_H */
