// C-ABI: error reporting + operator-level entry points (thin wrappers over the
// internal launchers so each kernel can be parity-tested on its own).
#include <stdarg.h>
#include <stdio.h>

#include "../../include/dim_hip.h"
#include "sp_kernels.h"

static thread_local char g_err[1024] = "";

void dim_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

#include <vector>
namespace {
unsigned long long g_prof_mask = 0ull;
std::vector<hipEvent_t> g_prof_ev;  // begin/end pairs
size_t g_prof_used = 0;
}  // namespace

// ---- fp16x3 range guard: one block of sticky counters per device, allocated on first use ----
namespace {
constexpr int kMaxDev = 16;
unsigned* g_sat_dev[kMaxDev] = {nullptr};
bool g_sat_failed[kMaxDev] = {false};
unsigned g_sat_host[DIM_SAT_SITES] = {0};
__global__ void read_clocks_kernel(unsigned long long* out) {
  out[0] = (unsigned long long)__builtin_readcyclecounter();
  out[1] = (unsigned long long)wall_clock64();
}
unsigned* sat_block() {
  int d = 0;
  if (hipGetDevice(&d) != hipSuccess || d < 0 || d >= kMaxDev) return nullptr;
  if (!g_sat_dev[d] && !g_sat_failed[d]) {
    void* p = nullptr;
    if (hipMalloc(&p, DIM_SAT_SITES * sizeof(unsigned)) != hipSuccess || hipMemset(p, 0, DIM_SAT_SITES * sizeof(unsigned)) != hipSuccess) g_sat_failed[d] = true;
    else g_sat_dev[d] = (unsigned*)p;
  }
  return g_sat_dev[d];
}
}  // namespace
unsigned* dim_sat_counter(int site) {
  unsigned* b = sat_block();
  return (b && site >= 0 && site < DIM_SAT_SITES) ? b + site : nullptr;
}
void dim_sat_host_bump(int site) {
  if (site >= 0 && site < DIM_SAT_SITES) g_sat_host[site]++;
}

// process defaults (dim_tune_set) with per-handle overrides (dim_handle_tune_set): see DimTuneScope in dim_common.h
static thread_local const DimTune* g_tune_scope = nullptr;
void dim_tune_scope_set(const DimTune* t) { g_tune_scope = t; }
const DimTune* dim_tune_scope_get() { return g_tune_scope; }
static inline int tuned(int key, int process_default) {
  const DimTune* t = g_tune_scope;
  return (t && t->v[key] >= 0) ? t->v[key] : process_default;
}
static int g_precision_mode = 2;
int dim_precision_mode() { return tuned(1, g_precision_mode); }
static int g_fuse_conv1a = 1;
int dim_fuse_conv1a() { return tuned(3, g_fuse_conv1a); }
static int g_fuse_sp_head = 1;
int dim_fuse_sp_head() { return tuned(16, g_fuse_sp_head); }
static int g_fold_out_proj = 1;
int dim_fold_out_proj() { return tuned(4, g_fold_out_proj); }
static int g_fuse_kv = 1;
int dim_fuse_kv() { return tuned(8, g_fuse_kv); }
static int g_follow_stop = 1;
int dim_follow_stop_flags() { return tuned(18, g_follow_stop); }
static int g_defer_assign = 1;
int dim_defer_assignment() { return tuned(17, g_defer_assign); }
static int g_fuse_ffn_ln = 3;
int dim_fuse_ffn_ln() { return tuned(11, g_fuse_ffn_ln); }
#ifdef DIM_RESEARCH   // research build (build.build_variant("research")): prototype / timing-probe selectors, see dim_kernels.h
static int g_gemm_kc = 32;
int dim_gemm_kc() { return g_gemm_kc; }
static int g_conv_wino = 0;
int dim_conv_winograd() { return g_conv_wino; }
static int g_gemm_probe = 0;
int dim_gemm_probe() { return g_gemm_probe; }
static int g_attn_probe = 0;
int dim_attn_probe() { return g_attn_probe; }
#endif
static int g_al_tile_rows = 16;
int dim_aliked_tile_rows() { return tuned(10, g_al_tile_rows); }
static int g_al_fuse_bn = 1;
int dim_aliked_fuse_bn() { return tuned(9, g_al_fuse_bn); }
static int g_presplit = 1;
int dim_presplit_activations() { return tuned(5, g_presplit); }

void dim_prof_begin(int site, hipStream_t s) {
  if (!((g_prof_mask >> site) & 1ull)) return;
  if (g_prof_used + 2 > g_prof_ev.size()) {
    hipEvent_t a, b;
    if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) return;
    g_prof_ev.push_back(a); g_prof_ev.push_back(b);
  }
  hipEventRecord(g_prof_ev[g_prof_used], s);
}
void dim_prof_end(int site, hipStream_t s) {
  if (!((g_prof_mask >> site) & 1ull) || g_prof_used + 2 > g_prof_ev.size()) return;
  hipEventRecord(g_prof_ev[g_prof_used + 1], s);
  g_prof_used += 2;
}

extern "C" {

int dim_profile_start(unsigned long long site_mask) {
  g_prof_mask = site_mask;
  g_prof_used = 0;
  return 0;
}

int dim_profile_stop(double* total_ms, int* launches) {
  DIM_REQUIRE(total_ms && launches, "dim_profile_stop: null argument");
  double tot = 0.0;
  for (size_t i = 0; i + 1 < g_prof_used; i += 2) {
    DIM_HIP(hipEventSynchronize(g_prof_ev[i + 1]));
    float ms = 0.f;
    DIM_HIP(hipEventElapsedTime(&ms, g_prof_ev[i], g_prof_ev[i + 1]));
    tot += ms;
  }
  *total_ms = tot;
  *launches = (int)(g_prof_used / 2);
  g_prof_mask = 0ull;
  g_prof_used = 0;
  return 0;
}

int dim_saturation_read(unsigned* counts_host, unsigned long long* total, int reset, void* stream) {
  unsigned c[DIM_SAT_SITES] = {0};
  unsigned* b = sat_block();
  if (b) {
    DIM_HIP(hipMemcpyAsync(c, b, sizeof(c), hipMemcpyDeviceToHost, (hipStream_t)stream));
    if (reset) DIM_HIP(hipMemsetAsync(b, 0, sizeof(c), (hipStream_t)stream));
    DIM_HIP(hipStreamSynchronize((hipStream_t)stream));
  }
  unsigned long long tot = 0;
  for (int i = 0; i < DIM_SAT_SITES; ++i) {
    c[i] += g_sat_host[i];
    if (reset) g_sat_host[i] = 0;
    tot += c[i];
    if (counts_host) counts_host[i] = c[i];
  }
  if (total) *total = tot;
  return 0;
}

int dim_saturation_reset(void* stream) {
  unsigned* b = sat_block();
  if (b) DIM_HIP(hipMemsetAsync(b, 0, DIM_SAT_SITES * sizeof(unsigned), (hipStream_t)stream));
  for (int i = 0; i < DIM_SAT_SITES; ++i) g_sat_host[i] = 0;
  return 0;
}

int dim_op_read_clocks(unsigned long long* out_dev, void* stream) {
  DIM_REQUIRE(out_dev, "dim_op_read_clocks: null argument");
  hipLaunchKernelGGL(read_clocks_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, out_dev);
  DIM_LAUNCH_CHECK();
  return 0;
}

const char* dim_last_error(void) { return g_err; }

int dim_abi_version(void) { return DIM_HIP_ABI_VERSION; }

int dim_op_gemm_f32(const float* A, int lda, const float* B, int ldb, int b_is_nk, const float* bias,
                    const float* residual, int ldr, float* C, int ldc, int M, int N, int K, int relu, void* stream) {
  GemmArgs g;
  g.A0 = A; g.lda0 = lda; g.B = B; g.ldb = ldb; g.bt = b_is_nk; g.bias = bias;
  g.R = residual; g.ldr = ldr; g.C = C; g.ldc = ldc; g.M = M; g.N = N; g.K = K; g.relu = relu;
  return launch_gemm(g, 1, (hipStream_t)stream);
}

int dim_op_conv3x3_nhwc_f32(const float* in, const float* w_tap_cin_cout, const float* bias, float* out, int batch,
                            int H, int W, int cin, int cout, int pool2x2, int relu, void* stream) {
  return launch_conv3x3(in, w_tap_cin_cout, bias, out, batch, H, W, cin, cout, pool2x2, relu, (hipStream_t)stream);
}

int dim_op_conv1a_f32(const float* in, const float* w_tap_cout, const float* bias, float* out, int batch, int H,
                      int W, void* stream) {
  return launch_conv1a(in, w_tap_cout, bias, out, batch, H, W, (hipStream_t)stream);
}

int dim_op_simple_nms_f32(const float* score_map, float* out, int batch, int H, int W, int radius, void* stream) {
  return launch_nms(score_map, out, batch, H, W, radius, (hipStream_t)stream);
}

// Op-level handles: a host struct holding the pre-split device operand for the precision mode that was active
// at creation (dim_tune_set key 1: 2 = fp16x3, 1 = bf16x6).
static int x3_create(const float* w_kn_host, int K, int N, void** out_dev, int* n_pad_out, int kperm);
int dim_x3_create(const float* w_kn_host, int K, int N, void** out_dev, int* n_pad_out) { return x3_create(w_kn_host, K, N, out_dev, n_pad_out, 0); }
int dim_x3_create_kperm(const float* w_kn_host, int K, int N, void** out_dev, int* n_pad_out) { return x3_create(w_kn_host, K, N, out_dev, n_pad_out, 1); }
static int x3_create(const float* w_kn_host, int K, int N, void** out_dev, int* n_pad_out, int kperm) {
  DIM_REQUIRE(w_kn_host && out_dev && n_pad_out && K > 0 && N > 0, "dim_x3_create: bad argument");
  const int mode = g_precision_mode == 1 ? 1 : 2;
  const int n_pad = (N + 127) / 128 * 128;
  std::vector<unsigned short> host(gemm_split_weight_elems(K, n_pad, mode));
  SplitWeights* w = new SplitWeights();
  split_weights(w_kn_host, K, N, n_pad, mode, host.data(), w, kperm);
  void* d = nullptr;
  if (hipMalloc(&d, host.size() * 2) != hipSuccess || hipMemcpy(d, host.data(), host.size() * 2, hipMemcpyHostToDevice) != hipSuccess) {
    delete w;
    dim_set_error("dim_x3_create: device allocation / upload failed (out of memory?)");
    return -1;
  }
  w->dev = (const unsigned short*)d; w->mode = mode; w->n_pad = n_pad;
  *out_dev = w; *n_pad_out = n_pad;
  return 0;
}
void dim_x3_destroy(void* handle) {
  if (!handle) return;
  SplitWeights* w = (SplitWeights*)handle;
  hipFree((void*)w->dev);
  delete w;
}
int dim_op_gemm_x6_f32(const float* A, int lda, const void* w_x3, int n_pad, const float* bias, const float* residual, int ldr,
                       float* C, int ldc, int M, int N, int K, int act, void* stream) {
  DIM_REQUIRE(w_x3 && ((const SplitWeights*)w_x3)->n_pad == n_pad, "dim_op_gemm_x6_f32: bad weight handle");
  GemmArgs g;
  g.A0 = A; g.lda0 = lda; g.set_split(*(const SplitWeights*)w_x3); g.bias = bias;
  g.R = residual; g.ldr = ldr; g.C = C; g.ldc = ldc; g.M = M; g.N = N; g.K = K; g.relu = act;
  g.sat = dim_sat_counter(DIM_SAT_OP);
  return launch_gemm_x6(g, 1, (hipStream_t)stream);
}

int dim_op_gemm_x6_ln_gelu_f32(const float* A, int lda, const void* w_x3, const float* bias, const float* ln_gamma, const float* ln_beta,
                               float* C, int ldc, int M, int K, void* stream) {
  DIM_REQUIRE(w_x3 && ((const SplitWeights*)w_x3)->n_pad == 512 && ((const SplitWeights*)w_x3)->mode == 2, "dim_op_gemm_x6_ln_gelu_f32: needs an fp16x3 512-column weight handle");
  DIM_REQUIRE(A && bias && ln_gamma && ln_beta && C, "dim_op_gemm_x6_ln_gelu_f32: null argument");
  GemmArgs g;
  g.A0 = A; g.lda0 = lda; g.set_split(*(const SplitWeights*)w_x3); g.bias = bias; g.ln_gamma = ln_gamma; g.ln_beta = ln_beta;
  g.C = C; g.ldc = ldc; g.M = M; g.N = 512; g.K = K;
  g.sat = dim_sat_counter(DIM_SAT_OP);
  return launch_gemm_x6(g, 1, (hipStream_t)stream);
}

int dim_op_gemm_x6_nt_f32(const float* A, int lda, const float* B, int ldb, float* C, int ldc, int M, int N, int K, void* stream) {
  DIM_REQUIRE(A && B && C && g_precision_mode != 0, "dim_op_gemm_x6_nt_f32: null argument, or the fp32 arithmetic is selected");
  GemmArgs g;
  g.A0 = A; g.lda0 = lda; g.B = B; g.ldb = ldb; g.bt = 1; g.C = C; g.ldc = ldc; g.M = M; g.N = N; g.K = K;
  return launch_gemm_x6_nt(g, 1, g_precision_mode == 1 ? 1 : 2, (hipStream_t)stream);
}

int dim_op_ffn_fused_f32(const float* A, int lda, const void* w0_x3, const float* bias0, const float* ln_gamma, const float* ln_beta,
                         const void* w3_x3_kperm, const float* bias3, const float* residual, int ldr, float* C, int ldc, int M, int K, void* stream) {
  DIM_REQUIRE(w0_x3 && ((const SplitWeights*)w0_x3)->n_pad == 512 && ((const SplitWeights*)w0_x3)->mode == 2, "dim_op_ffn_fused_f32: needs an fp16x3 512-column ffn.0 handle");
  DIM_REQUIRE(w3_x3_kperm && ((const SplitWeights*)w3_x3_kperm)->n_pad == 256 && ((const SplitWeights*)w3_x3_kperm)->mode == 2, "dim_op_ffn_fused_f32: needs an fp16x3 256-column ffn.3 handle (dim_x3_create_kperm)");
  DIM_REQUIRE(A && bias0 && ln_gamma && ln_beta && bias3 && residual && C, "dim_op_ffn_fused_f32: null argument");
  GemmArgs g;
  g.A0 = A; g.lda0 = lda; g.set_split(*(const SplitWeights*)w0_x3); g.bias = bias0; g.ln_gamma = ln_gamma; g.ln_beta = ln_beta;
  g.set_split2(*(const SplitWeights*)w3_x3_kperm); g.bias2 = bias3; g.R = residual; g.ldr = ldr;
  g.C = C; g.ldc = ldc; g.M = M; g.N = 512; g.K = K;
  g.sat = dim_sat_counter(DIM_SAT_OP); g.sat2 = dim_sat_counter(DIM_SAT_OP);
  return launch_gemm_x6(g, 1, (hipStream_t)stream);
}

int dim_convx6_create(const float* w_oihw_host, int cin, int cout, void** out_dev) {
  DIM_REQUIRE(w_oihw_host && out_dev && (cin == 64 || cin == 128) && cout % 64 == 0, "dim_convx6_create: bad argument");
  const int mode = g_precision_mode == 1 ? 1 : 2;
  std::vector<unsigned short> host(conv_split_weight_elems(cin, cout, mode));
  SplitWeights* w = new SplitWeights();
  prepare_conv_weights_split(w_oihw_host, cin, cout, mode, host.data(), w);
  void* d = nullptr;
  if (hipMalloc(&d, host.size() * 2) != hipSuccess || hipMemcpy(d, host.data(), host.size() * 2, hipMemcpyHostToDevice) != hipSuccess) {
    delete w;
    dim_set_error("dim_convx6_create: device allocation / upload failed (out of memory?)");
    return -1;
  }
  w->dev = (const unsigned short*)d; w->mode = mode;
  *out_dev = w;
  return 0;
}
int dim_op_conv3x3_x6_nhwc_f32(const float* in, const void* w_x6, const float* bias, float* out, int batch, int H, int W,
                               int cin, int cout, int pool2x2, int relu, void* stream) {
  DIM_REQUIRE(w_x6, "dim_op_conv3x3_x6_nhwc_f32: null weight handle");
  return launch_conv3x3_x6(in, *(const SplitWeights*)w_x6, bias, out, batch, H, W, cin, cout, pool2x2, relu, (hipStream_t)stream, dim_sat_counter(DIM_SAT_OP));
}

int dim_tune_set(int key, int value) {
  if (key == 0) dim_conv_set_variant(value);
  if (key == 1) g_precision_mode = value;
  if (key == 2) dim_conv_x6_set_variant(value);
  if (key == 3) g_fuse_conv1a = value;
  if (key == 4) g_fold_out_proj = value;
  if (key == 5) g_presplit = value;
  if (key == 6) dim_gemm_x6_set_wide(value);
  if (key == 7) dim_nms_set_big_tiles(value);
  if (key == 8) g_fuse_kv = value;
  if (key == 9) g_al_fuse_bn = value;
  if (key == 10) g_al_tile_rows = value;
  if (key == 11) g_fuse_ffn_ln = value;
  if (key == 16) g_fuse_sp_head = value;
  if (key == 17) g_defer_assign = value;
  if (key == 18) g_follow_stop = value;
#ifdef DIM_RESEARCH
  if (key == 12) g_attn_probe = value;
  if (key == 13) g_gemm_probe = value;
  if (key == 14) g_gemm_kc = value;
  if (key == 15) g_conv_wino = value;
#else
  DIM_REQUIRE(key < 12 || key > 15, "dim_tune_set: key %d selects a research prototype / timing probe that the product library does not contain "
              "(build.build_variant(\"research\", [\"-DDIM_RESEARCH\"]) -> libdim_hip_research.so)", key);
#endif
  DIM_REQUIRE(key >= 0 && key <= 18, "dim_tune_set: unknown key %d", key);
  return 0;
}

int dim_handle_tune_set(void* handle, int key, int value) {
  DimHandleBase* b = (DimHandleBase*)handle;
  DIM_REQUIRE(b != nullptr && b->magic == DIM_HANDLE_MAGIC, "dim_handle_tune_set: not an extractor / matcher handle of this library");
  DIM_REQUIRE(key == 1 || key == 3 || key == 4 || key == 5 || key == 8 || key == 9 || key == 10 || key == 11 || key == 16 || key == 17 || key == 18,
              "dim_handle_tune_set: key %d has no per-handle form (arithmetic 1; fusion 3, 4, 5, 8, 9, 11, 16, 17, 18; ALIKED tile rows 10)", key);
  b->tune.v[key] = value < 0 ? -1 : value;
  return 0;
}

int dim_device_synchronize(void) {
  DIM_HIP(hipDeviceSynchronize());
  return 0;
}

}  // extern "C"
