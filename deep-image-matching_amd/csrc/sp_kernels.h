// Launchers of the SuperPoint post-processing kernels (sp_post.hip).
#pragma once
#include "dim_kernels.h"

int launch_softmax_d2s(const float* logits, float* smap, int batch, int h, int w, hipStream_t s);
int launch_nms(const float* smap, float* out, int batch, int H8, int W8, int radius, hipStream_t s);
int launch_select(const float* nms, int batch, int H8, int W8, float thr, int border, int* rowcount, int* rowoff,
                  int* ncand, float* cand_score, int* cand_idx, hipStream_t s);
// as launch_select with an optional per-image device threshold and a count-only mode (ALIKED's DKD)
int launch_select_ex(const float* nms, int batch, int H8, int W8, float thr, const float* thr_dev, int border, int* rowcount,
                     int* rowoff, int* ncand, float* cand_score, int* cand_idx, int count_only, hipStream_t s);
// sort_always: score-descending output also when there are fewer candidates than k (else row-major keep-all, SPN:75-76).
// k <= 4096: one workgroup per image; larger k (up to 32768) sorts 4096-key chunks of a global key table `scratch`
// (topk_scratch_keys(batch, k) 8-byte elements; may be null when that is 0)
size_t topk_scratch_keys(int batch, int k);
int launch_topk(const float* cand_score, const int* cand_idx, const int* ncand, int batch, int H8, int W8, int k,
                int capacity, float* kpts, float* scores, int* n_out, unsigned long long* scratch, int sort_always, hipStream_t s);
// after launch_topk(sort_always = 1): images with n_out < k are filled up to k with the first non-candidate pixels in row-major order, score 0
// (DKD's top-k mode, ALN:150-151)
int launch_topk_zero_fill(const float* nms, int batch, int H8, int W8, float thr, int border, int k, int capacity, float* kpts, float* scores,
                          int* n_out, hipStream_t s);
int launch_sample_desc(const float* dense, const float* kpts, const int* n_kpts, float* desc, int batch, int h, int w,
                       int capacity, int fix_sampling, hipStream_t s);
