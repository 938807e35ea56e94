// Launchers of the ALIKED kernels (aliked.hip).  All tensors NHWC fp32.
#pragma once
#include "dim_kernels.h"

enum { AL_ACT_NONE = 0, AL_ACT_SELU = 1, AL_ACT_SIGMOID = 2 };

// out[b][y][x][co] = act( sum in[b][y+dy][x+dx][ci] * w[tap][ci][co] + bias ), zero padding, over an H x W map;
// only the window [crop_y, crop_y+out_h) x [crop_x, crop_x+out_w) is stored (out is [b][out_h][out_w][out_c]).
int launch_al_conv3x3(const float* in, int cin, const float* w, const float* bias, float* out, int cout, int batch, int H,
                      int W, int act, int crop_y, int crop_x, int out_h, int out_w, hipStream_t s);
int launch_al_conv1x1(const float* in, int cin, const float* w, const float* bias, float* out, int cout, int n_pixels,
                      int act, hipStream_t s);
int launch_al_pad_replicate(const float* img, float* out, int batch, int H, int W, int Hp, int Wp, int pad_t, int pad_l,
                            int in_ch, hipStream_t s);
int launch_al_avgpool(const float* in, float* out, int batch, int H, int W, int C, int k, hipStream_t s);
// BatchNorm in TRAINING mode (Q7), statistics per image: stats -> (alpha, beta) with y = x*alpha + beta
int launch_al_bn_stats(const float* x, int batch, int n_pixels, int C, const float* gamma, const float* beta_w,
                       double* partial, float* alpha, float* beta, hipStream_t s);
int launch_al_bn_apply(const float* x, const float* alpha, const float* beta, const float* residual, float* out, int batch,
                       int n_pixels, int C, hipStream_t s);
// bn_apply (+ residual) + SELU fused with the k x k average pooling that follows (k = 2 or 4): writes the activated map AND its pooled copy
int launch_al_bn_apply_pool(const float* x, const float* alpha, const float* beta, const float* residual, float* out, float* pooled, int batch,
                            int H, int W, int C, int k, hipStream_t s);
// deformable 3x3 conv (ALN:274-330): offsets [b][H][W][off_c>=18] (dy,dx per tap, already clamped), no bias
// wx != nullptr: the [9 cin][cout] product runs as fp16x3 on gemm_x6 (range guard counter `sat`), else as fp32 MFMA
int launch_al_deform_conv(const float* in, int cin, const float* offsets, int off_c, const float* w, const SplitWeights* wx, unsigned* sat,
                          float* cols, float* out, int cout, int batch, int H, int W, hipStream_t s);
int launch_al_clamp(float* x, size_t n, float lim, hipStream_t s);
// feature aggregation + first score-head layer (ALN:657-668)
int launch_al_assemble(const float* x1, const float* f2, const float* f3, const float* f4, const float* w1, const float* ws0,
                       float* x1234, float* s8, int c1, int batch, int Hp, int Wp, hipStream_t s);
// the product form of the same stage: q_g = f_g x score_head.0[32g:32g+32] at the maps' own resolutions (8 channels), interpolated
int launch_al_assemble_proj(const float* x1, const float* q2, const float* q3, const float* q4, const float* w1, const float* ws0, float* s8,
                            int c1, int batch, int Hp, int Wp, hipStream_t s);
// DKD sub-pixel refinement (ALN:176-216) and SDDH pieces (ALN:503-558)
int launch_al_dkd_refine(const float* score, const float* kpts_px, const int* n_kpts, float* kpts_norm, float* disp,
                         float* kscore, float* kpts_out, int batch, int H, int W, int capacity, int radius, hipStream_t s);
// sources of the (never materialised) 128-channel feature map x1234, padded frame Hp x Wp
struct AlFeat { const float *x1, *f2, *f3, *f4, *w1; int Hp, Wp, c1; };   // c1 = channels of x1: 16 (dim 128) or 8 (aliked-t16, dim 64)
inline int al_deform_krow(int cin) { return (9 * cin + 31) / 32 * 32; }   // row length of the deformed im2col matrix (K of its GEMM)
int launch_al_sddh_patches(const AlFeat& F, const float* kpts_norm, const int* n_kpts, float* patches, int batch, int H, int W,
                           int pad_t, int pad_l, int capacity, hipStream_t s);
int launch_al_sddh_sample(const AlFeat& F, const float* kpts_norm, const int* n_kpts, const float* off_hidden, const float* w2,
                          const float* b2, float* feats, int M, int batch, int H, int W, int pad_t, int pad_l, int capacity, hipStream_t s);
int launch_al_normalize_rows(float* x, const int* n_rows, int batch, int capacity, int C, hipStream_t s);
int launch_al_mean(const float* x, int batch, int n, double* partial, float* mean, hipStream_t s);
int launch_al_pick_threshold(const int* ncand, const float* mean, float thr, float* thr_out, int batch, hipStream_t s);

// ---- fp16x3 matrix-core path of the small-channel convolutions (aliked_x3.hip) ----
// weights: split_weights(K = taps * cin_pad, N = cout, n_pad = 32, mode 2).  partial (optional): fp64 (sum, sum of squares) per
// (image, workgroup, output channel) for the train-mode BatchNorm that follows; *n_wg_out = workgroups per image.
size_t al_convx3_partial_doubles(int batch, int H, int W);
// in_alpha / in_beta (optional, [batch][in_c]): the input is the RAW output of the previous convolution and selu(x * alpha + beta) —
// its BatchNorm + activation — is applied while the tile is staged (no separate bn_apply pass over the map).
int launch_al_convx3(const float* in, int in_c, int cin_pad, int taps, const SplitWeights& w, const float* bias, float* out, int cout,
                     int batch, int H, int W, double* partial, int* n_wg_out, const float* in_alpha, const float* in_beta, hipStream_t s);
int launch_al_bn_final_tiles(const double* partial, int n_wg, int batch, int n_pixels, int C, const float* gamma, const float* beta_w,
                             float* alpha, float* beta, hipStream_t s);
// feature aggregation + score_head.0 with both channel contractions on the matrix cores (al_assemble_x3_kernel); the constant
// operands are prepared once by al_assemble_x3_prepare (frag: al_assemble_x3_frag_halves() 16-bit values; inv1: 32 floats)
size_t al_assemble_x3_frag_halves();
void al_assemble_x3_prepare(const float* w1_ci_co, const float* ws0_co_k, unsigned short* frag, float* inv1, float* inv0);
int launch_al_assemble_x3(const float* x1, const float* q2, const float* q3, const float* q4, const void* frag_dev, const float* inv1_dev, float inv0,
                          float* s8, int batch, int Hp, int Wp, hipStream_t s);
