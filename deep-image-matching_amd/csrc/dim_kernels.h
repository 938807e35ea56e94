// Internal launcher interface shared by the SuperPoint and LightGlue pipelines.
// Everything here runs on a caller-provided HIP stream and never synchronises.
#pragma once
#include "dim_common.h"

// ---------------------------------------------------------------------------
// Batched, ragged fp32 GEMM on v_mfma_f32_32x32x2_f32:
//   C[z][m][n] = act( sum_k A[z][m][k] * B[k][n] + bias[n] ) (+ R[z][m][n])
// A may be split along K between two sources (A0 for k < ksplit, A1 after) so
// that LightGlue's ffn(cat[x, message]) never materialises the concatenation
// (reference LGN:159,209).  bt != 0 means B is given as [n][k] row-major (the
// "NT" product used for sim = mdesc0 * mdesc1^T, LGN:271).
// Raggedness: rows of item z = rows[z*rows_mul + rows_off] (device array) when
// rows != nullptr, else M; same for cols (bt mode only).  flag gating: the item
// is skipped unless flag[z >> flag_shift] == flag_eq (flag == nullptr: always run).
// A matrix-core operand pre-split on the host into 16-bit planes (SplitMma policy, dim_common.h):
// mode 1 = three bf16 planes, mode 2 = two fp16 planes of the power-of-two-scaled weights.  The scale is chosen PER
// OUTPUT CHANNEL (max|w[:, co]| lands in [8192, 16384): one outlier weight cannot push the other channels' weights
// towards the subnormal low piece); the exact per-channel factors the accumulator is multiplied by in the epilogue
// (1 / (2^s_co * activation scale); all 1 for mode 1) are fp32 values stored in the tail of the same device buffer,
// `scale_off` 16-bit elements from its start.
struct SplitWeights {
  const unsigned short* dev = nullptr;
  int mode = 0;
  size_t scale_off = 0;
  int n_pad = 0;  // GEMM operands: padded column count
  const float* inv_ch() const { return (const float*)(dev + scale_off); }
};
// per-layer weights of one linear (LightGlue's final_proj): dim_lg_match's deferred assignment lets every item pick the layer its pair stopped at
struct GemmLayerTab { const unsigned short* Bx3; const float* inv_ch; const float* bias; };
struct GemmArgs {
  const float* A0 = nullptr; const float* A1 = nullptr;
  int lda0 = 0, lda1 = 0, ksplit = 0;
  long long strideA0 = 0, strideA1 = 0;
  const int* a_idx = nullptr;  // optional indirection: A0 of item z starts at A0 + a_idx[z]*strideA0
  const float* B = nullptr; int ldb = 0; long long strideB = 0; int bt = 0;
  const unsigned short* Bx3 = nullptr; int n_pad = 0;  // split path: weights pre-split into [planes][n_pad][K] 16-bit pieces
  int split_mode = 1; const float* inv_ch = nullptr;    //   (SplitWeights::mode / per-column inverse scales [n_pad])
  void set_split(const SplitWeights& w) { Bx3 = w.dev; n_pad = w.n_pad; split_mode = w.mode; inv_ch = w.inv_ch(); }
  unsigned* sat = nullptr;  // fp16x3 range guard: counter bumped when max|C| > DIM_F16_ACT_LIMIT (dim_common.h); nullptr = unchecked
  const float* bias = nullptr;
  const float* R = nullptr; int ldr = 0; long long strideR = 0;
  float* C = nullptr; int ldc = 0; long long strideC = 0;
  int M = 0, N = 0, K = 0;
  const int* rows = nullptr; int rows_mul = 1, rows_off = 0, rows_scale = 1;
  const int* cols = nullptr; int cols_mul = 1, cols_off = 0;
  const int* flag = nullptr; int flag_shift = 0, flag_eq = 0;
  // flag_any: run for every item whose flag is > 0 (instead of == flag_eq); layer_tab (plain split-precision blocks only): such an item takes
  // Bx3 / inv_ch / bias from layer_tab[flag - 1]
  int flag_any = 0; const GemmLayerTab* layer_tab = nullptr;
  int relu = 0;  // epilogue activation: 0 none, 1 ReLU, 2 SELU
  // LightGlue q|k|v projections (fp16x3, 128 x 256 blocks only; gemm_x6_fuses_kv()): the 256-column blocks kv_kblock / kv_vblock
  // are not stored as fp32 but written pre-split, in the tile-image layout lg_attn_x6.hip's attention kernel stages from
  // (kv_img: [item][4 heads][kv_tiles][KV_TILE_STRIDE x 16 B]), K with the rotary embedding applied when kv_enc != nullptr
  // ([item][kv_nmax][64] cos | sin) — what kv_prep_kernel did in a separate pass over the fp32 projections.
  void* kv_img = nullptr; int kv_tiles = 0, kv_kblock = -1, kv_vblock = -1, kv_nmax = 0; const float* kv_enc = nullptr;
  // LightGlue ffn.0 (N = 512, fp16x3): when ln_gamma != nullptr the block spans ALL 512 columns of 64 rows and the epilogue applies
  // LayerNorm(512, eps 1e-5, affine) + erf-GELU (LGN:141-142) before the store — the separate lg_ln_gelu pass (a read and a write
  // of the 512-wide hidden tensor) disappears.  Needs N == n_pad == 512, no residual, no activation.
  const float* ln_gamma = nullptr; const float* ln_beta = nullptr;
  // ... and when B2x3 != nullptr the same workgroup goes on to ffn.3 (512 -> 256; weights split with the k permutation of
  // split_weights(kperm = 1)) + bias2 + residual R: C [M][256] = R + gelu(layer_norm(A W + bias)) W2 + bias2, the hidden tensor
  // never stored (gemm_x6_ffn_fused_kernel).  sat guards the hidden values, sat2 the outputs.
  const unsigned short* B2x3 = nullptr; const float* inv_ch2 = nullptr; const float* bias2 = nullptr; unsigned* sat2 = nullptr;
  void set_split2(const SplitWeights& w) { B2x3 = w.dev; inv_ch2 = w.inv_ch(); }
  // SuperPoint's detector tail (fp16x3, N = 65 over M = maps x d2s_h x d2s_w cells, batch 1): when d2s_out != nullptr the block spans all 65
  // channels of 128 cells and the epilogue applies the 65-way softmax, drops the dustbin and stores the 8 x 8 blocks of the score map
  // [maps][8 d2s_h][8 d2s_w] (SPN:176-179) — C (the logits) is not written (gemm_x6_head_kernel).
  float* d2s_out = nullptr; int d2s_h = 0, d2s_w = 0;
};
constexpr int KV_TILE_STRIDE = 3 * 8 * 32 + 3 * 2 * 2 * 64;  // 16-byte slots reserved per (item, head, 32-key tile) image (bf16x6 fills all 1536)
bool gemm_x6_fuses_kv(int M, int n_pad, int batch, int split_mode);  // true when launch_gemm_x6 will honour kv_img for this shape
int launch_gemm(const GemmArgs& a, int batch, hipStream_t s);
// fp32-accurate GEMM on the 16-bit matrix cores (gemm_x6.hip); needs a.set_split(...).
int launch_gemm_x6(const GemmArgs& a, int batch, hipStream_t s);
// C = A * B^T, both operands fp32 activations split on the fly (bt layout of launch_gemm: B [n][k], ragged rows AND cols); the
// producers of A and B must be range-guarded in split_mode 2
int launch_gemm_x6_nt(const GemmArgs& a, int batch, int split_mode, hipStream_t s);
// host: [K][N] fp32 -> the pre-split device layout; elems = planes * n_pad * K 16-bit values + 2 * n_pad for the fp32
// per-column inverse scales in the tail (sw->scale_off); fills sw->mode / n_pad / scale_off (not sw->dev)
size_t gemm_split_weight_elems(int K, int n_pad, int mode);
void split_weights(const float* w_kn, int K, int N, int n_pad, int mode, unsigned short* out, SplitWeights* sw, int kperm = 0);
// kperm = 1: inside every 16-wide k-step, k value 8 (e >> 2) + 4 hf + (e & 3) is stored at (k-half hf, element e) — the order in which
// a transposed MFMA result tile presents its rows when it is reused as an operand (gemm_x6_ffn_fused_kernel)

// ---------------------------------------------------------------------------
// 3x3 / stride 1 / pad 1 convolution over NHWC fp32 images as an implicit GEMM
// on MFMA, bias + ReLU fused, optional fused 2x2/2 max-pool (floor semantics,
// SPN:163-171).  in: [B][H][W][cin], w: [9][cin][cout], out: [B][H'][W'][cout].
int launch_conv3x3(const float* in, const float* w, const float* bias, float* out, int batch, int H, int W,
                   int cin, int cout, int pool, int relu, hipStream_t s);
// fp32-accurate 3x3 conv on the 16-bit matrix cores (conv_x6.hip); weights pre-split by prepare_conv_weights_split.
size_t conv_split_weight_elems(int cin, int cout, int mode);
void prepare_conv_weights_split(const float* w_oihw, int cin, int cout, int mode, unsigned short* out, SplitWeights* sw);
// `sat` (all x6 conv launchers): fp16x3 range-guard counter for the outputs (dim_common.h), nullptr = unchecked
int launch_conv3x3_x6(const float* in, const SplitWeights& wt, const float* bias, float* out, int batch, int H, int W, int cin,
                      int cout, int pool, int relu, hipStream_t s, unsigned* sat = nullptr);
// conv1a (image -> 64 channels, weights [9][64]) computed on the fly inside the following 64 -> cout conv
int launch_conv3x3_x6_fused1a(const float* image, const float* w1a_tap_cout, const float* b1a, const SplitWeights& wt, const float* bias,
                              float* out, int batch, int H, int W, int cout, int pool, int relu, int planes_out, hipStream_t s,
                              unsigned* sat = nullptr, unsigned* sat_image = nullptr);
// fp16x3 only: input and / or output pre-split in the bytes of the fp32 NHWC tensor: per pixel and 16-channel group, 16 h then 16 l fp16 pieces (conv_x6.hip)
int launch_conv3x3_x6_planes(const float* in, const SplitWeights& wt, const float* bias, float* out, int batch, int H, int W, int cin,
                             int cout, int pool, int relu, int planes_in, int planes_out, hipStream_t s, unsigned* sat = nullptr);
// Winograd F(2,3)-along-x variant of the fused conv1a + 3x3 convolution (conv_wg.hip): fp16x3 only, 2/3 of the MFMAs
size_t conv_wino_weight_elems(int cin, int cout);
void prepare_conv_weights_wino(const float* w_oihw, int cin, int cout, unsigned short* out, SplitWeights* sw);
int launch_conv3x3_wg_fused1a(const float* image, const float* w1a_tap_cout, const float* b1a, const SplitWeights& wt, const float* bias,
                              float* out, int batch, int H, int W, int cout, int pool, int planes_out, hipStream_t s, unsigned* sat = nullptr,
                              unsigned* sat_image = nullptr);
// Research build only (-DDIM_RESEARCH = build.build_variant("research") -> lib/libdim_hip_research.so): the default-off prototypes that lost their A/B
// (Winograd conv1b, 64-wide K chunks, double-buffered / tile-refilled GEMM blocks, round 3's feed-forward loop) and the timing probes that give wrong
// results by design.  The product library compiles none of them: the selectors below are constants there and dim_tune_set rejects keys 12-15.
#ifdef DIM_RESEARCH
int dim_conv_winograd();     // dim_tune_set key 15: bit 0 = SuperPoint conv1b (fused conv1a) runs the Winograd F(2,3) kernel (default 0 until measured faster)
#else
static inline int dim_conv_winograd() { return 0; }
#endif
int launch_planes_to_f32(const void* planes, int batch, int hw, int channels, float* out, hipStream_t s);  // [batch][hw pixels][channels]
// a pre-split image occupies an even number of pixel slots (pixels are stored in pairs): size buffers with this
inline size_t dim_planes_image_pixels(int h, int w) { return ((size_t)h * w + 1) & ~(size_t)1; }
int dim_presplit_activations();  // 1 (default): SuperPoint's conv-to-conv activations are stored pre-split (dim_tune_set key 5)
int dim_precision_mode();  // 2 (default): fp16x3, 1: bf16x6 on the 16-bit matrix cores; 0: fp32 MFMA (dim_tune_set key 1)
int dim_aliked_tile_rows();  // dim_tune_set key 10: tile rows (16 | 8) of ALIKED's 16-channel 3x3 matrix-core convolution
int dim_aliked_fuse_bn();   // dim_tune_set key 9: ALIKED folds BatchNorm + SELU into the consuming convolution's staging (default 1)
int dim_fuse_conv1a();     // 1 (default): SuperPoint conv1a is computed inside conv1b (dim_tune_set key 3)
int dim_fuse_sp_head();    // 1 (default): SuperPoint's convPb + softmax + depth-to-space are one kernel in fp16x3 (dim_tune_set key 16)
int dim_fold_out_proj();   // 1 (default): LightGlue out_proj folded into ffn.0's weights in the split modes (dim_tune_set key 4)
#ifdef DIM_RESEARCH
int dim_gemm_kc();           // dim_tune_set key 14: 32 (product); 64 / 33 = prototypes of the wide GEMM blocks, 36 = the fused feed-forward's previous K loop, 35 = timing probe
int dim_gemm_probe();        // 0 (product); timing probes of the wide fp16x3 GEMM blocks (dim_tune_set key 13; gemm_x6.hip PROBE)
int dim_attn_probe();        // 0 (product); 1 / 2 / 3: timing probes of cross attention (dim_tune_set key 12; lg_attn_x6.hip)
#else
static inline int dim_gemm_kc() { return 32; }
static inline int dim_gemm_probe() { return 0; }
static inline int dim_attn_probe() { return 0; }
#endif
int dim_fuse_ffn_ln();       // dim_tune_set key 11.  3 (default): LightGlue's whole feed-forward (ffn.0, LayerNorm, GELU, ffn.3, residual) is one kernel when the
                             // launch fills the GPU with 64-row blocks, 4 = always (tests); 1 / 2: only LayerNorm + GELU in ffn.0's epilogue; 0: separate kernels
int dim_follow_stop_flags();   // 1 (default): dim_lg_match with adaptive depth on a handle for <= 2 pairs follows the stop flags on the host and stops enqueueing layers nobody needs (dim_tune_set key 18)
int dim_defer_assignment();   // 1 (default): adaptive-depth LightGlue runs the assignment ONCE after the layer loop, every pair with its stop layer's weights (dim_tune_set key 17; 0 = gated launches after every layer)
int dim_fuse_kv();           // 1 (default): LightGlue's K | V tile images written by the projection GEMM's epilogue (dim_tune_set key 8)
int dim_nms_big_tiles();     // 1 (default): 64 x 64 NMS tiles on large score maps (dim_tune_set key 7; 2 = forced)
void dim_nms_set_big_tiles(int v);
int dim_gemm_x6_wide();      // 1 (default): 128 x 256 workgroup blocks where n_pad allows (dim_tune_set key 6)
void dim_gemm_x6_set_wide(int v);
void dim_conv_x6_set_variant(int v);  // tuning hook: prefetch variant of conv3x3_x6 (dim_tune_set key 2)
void dim_conv_set_variant(int v);  // tuning hook: selects the conv3x3 kernel variant (see conv.hip)
// conv1a: 1 -> 64 channels, direct (VALU) convolution; in: [B][H][W], w: [9][64].
int launch_conv1a(const float* in, const float* w, const float* bias, float* out, int batch, int H, int W,
                  hipStream_t s);

// ---------------------------------------------------------------------------
// Optional per-launch-site timing with HIP events on the launch stream (used by
// bench.py for the roofline figure; zero cost when disabled).  Site ids are the
// DIM_PROF_* constants of include/dim_hip.h.
void dim_prof_begin(int site, hipStream_t s);
void dim_prof_end(int site, hipStream_t s);
