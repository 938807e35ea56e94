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
int launch_topk(const float* cand_score, const int* cand_idx, const int* ncand, int batch, int H8, int W8, int k,
                int capacity, float* kpts, float* scores, int* n_out, hipStream_t s);
int launch_sample_desc(const float* dense, const float* kpts, const int* n_kpts, float* desc, int batch, int h, int w,
                       int capacity, int fix_sampling, hipStream_t s);
