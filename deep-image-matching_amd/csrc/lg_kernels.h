// Launchers of the LightGlue kernels (lg_kernels.hip, lg_attn.hip).
//
// Everything is batched over ITEMS (item = 2*pair + side) and ragged: the live
// point count of an item is n_cur[item] (device), rows beyond it are never
// touched.  `done[pair]` gates every launch: 0 = still iterating, >0 = stopped
// at that layer (early stop or last layer), <0 = left through the reference's
// "no keypoints" exit (LGN:491-492,518-540).  No launch reads anything back.
#pragma once
#include "dim_kernels.h"

struct LgState {  // device pointers owned by the handle
  int n_pairs, n_items, nmax;
  float* desc;    // [items][nmax][256]
  float* enc;     // [items][nmax][64]   cos[32] | sin[32]
  float* qkv;     // [items][nmax][768]
  float* ctx;     // [items][nmax][256]
  float* msg;     // [items][nmax][256]
  float* hid;     // [items][nmax][512]
  float* md;      // [items][nmax][256]
  float* sim;     // [pairs][nmax][nmax]
  float* conf;    // [items][nmax] token confidence
  float* mtch;    // [items][nmax] matchability (sigmoid)
  float* zls;     // [items][nmax] logsigmoid(matchability logit)
  float* rmax; float* rlse;  // [items][nmax] row (side 0) / column (side 1) softmax stats of sim
  float* best;    // [items][nmax] best score per row / column
  int* arg;       // [items][nmax] argmax per row / column
  int* n_cur; int* n_new; int* n_orig;  // [items]
  int* ind;       // [items][nmax] original index of each live point
  int* dest;      // [items][nmax] compaction destination
  int* prune;     // [items][nmax] (indexed by ORIGINAL point index)
  int* done;      // [pairs]
  int* cnt_lt;    // [pairs]
  float* tdesc; float* tenc; int* tind;  // compaction scratch
  void* kv_img;   // [items][4][ceil(nmax/32)][1536 x 16 B] pre-split K|V tile images (split-precision attention)
  unsigned* sat_qkv = nullptr; unsigned* sat_ffn = nullptr;  // fp16x3 range-guard counters (dim_common.h), set per call
  int nsel = 0;   // upper bound of the live rows of any item in THIS call (<= nmax; dim_lg_match: the feature table's rows per image): launch shapes follow it, strides follow nmax
  float* attn_part; int attn_part_items;  // scratch of the key-split attention used for small batches ([items][4][nmax][4][68])
};

int launch_lg_init(const LgState& st, const float* kpts_tab, const float* desc_tab, const int* n_tab, const float* size_tab,
                   const int* pair_idx, int cap, int in_dim, const float* Wr, int copy_desc, unsigned* sat, hipStream_t s);
int launch_lg_rotary(const LgState& st, hipStream_t s);
// kv_ready: the K | V tile images were already written by the projection GEMM (gemm_x6.hip KV epilogue): no kv_prep pass, and
// cross attention takes its Q operand from the item's own K image
int launch_lg_attention(const LgState& st, int cross, hipStream_t s, int kv_ready = 0);      // dispatches on dim_precision_mode()
int launch_lg_attention_x6(const LgState& st, int cross, hipStream_t s, int kv_ready = 0);   // split-precision variant (lg_attn_x6.hip)
int launch_lg_ln_gelu(const LgState& st, const float* gamma, const float* beta, hipStream_t s);
int launch_lg_confidence(const LgState& st, const float* w_tok, const float* b_tok, const float* w_match,
                         const float* b_match, float thr, int use_token, hipStream_t s);
int launch_lg_decide(const LgState& st, int layer, float depth_conf, int early, int last, hipStream_t s, int* mirror = nullptr, int seq_off = 0, int seq = 0);
int launch_lg_prune(const LgState& st, int layer, double width_conf, float thr, int use_token, int pruning_min,
                    hipStream_t s);
int launch_lg_assign_stats(const LgState& st, int tag, const float* w_match, const float* b_match, hipStream_t s);
int launch_lg_assign_argmax(const LgState& st, int tag, float* dense_scores, hipStream_t s);
int launch_lg_finalize(const LgState& st, int n_layers, int prune_enabled, float filter_thr, int out_cap,
                       long long* matches, float* mscores, int* n_matches, int* mfull, float* msfull, int* stop,
                       int* prune_out, hipStream_t s);
