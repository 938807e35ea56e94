// Shared device/host helpers for the gfx950 (MI355X, CDNA4) kernels.
// wave = 64 lanes everywhere; MFMA = v_mfma_f32_32x32x2_f32 (exact fp32 fmaf
// chain, 157 TFLOP/s dense peak — MI355X_MICROARCH.md "Peak FP32 (matrix)").
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define DIM_WAVE 64

// D = A(32x2) * B(2x32) + C on one wave.
//   A operand: lane l supplies A[i = l&31][k = l>>5]
//   B operand: lane l supplies B[k = l>>5][j = l&31]
//   C/D     : lane l, reg r holds D[row = (r&3) + 8*(r>>2) + 4*(l>>5)][col = l&31]
__device__ __forceinline__ f32x16 mfma32(float a, float b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}
// row index inside a 32x32 MFMA tile held by (lane-half h, register r)
__device__ __forceinline__ int mfma_row(int r, int half) { return (r & 3) + 8 * (r >> 2) + 4 * half; }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
  return v;
}
__device__ __forceinline__ int wave_sum_i(int v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

// ---- host side -------------------------------------------------------------
void dim_set_error(const char* fmt, ...);
#define DIM_HIP(expr)                                                                 \
  do {                                                                                \
    hipError_t e__ = (expr);                                                          \
    if (e__ != hipSuccess) {                                                          \
      dim_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e__), __FILE__, __LINE__); \
      return -1;                                                                      \
    }                                                                                 \
  } while (0)
#define DIM_LAUNCH_CHECK()                                                            \
  do {                                                                                \
    hipError_t e__ = hipGetLastError();                                               \
    if (e__ != hipSuccess) {                                                          \
      dim_set_error("kernel launch failed: %s (%s:%d)", hipGetErrorString(e__), __FILE__, __LINE__); \
      return -1;                                                                      \
    }                                                                                 \
  } while (0)
#define DIM_REQUIRE(cond, ...)            \
  do {                                    \
    if (!(cond)) {                        \
      dim_set_error(__VA_ARGS__);         \
      return -2;                          \
    }                                     \
  } while (0)

static inline int cdiv(int a, int b) { return (a + b - 1) / b; }
