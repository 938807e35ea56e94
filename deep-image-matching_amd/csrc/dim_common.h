// Shared device/host helpers for the gfx950 (MI355X, CDNA4) kernels.
// wave = 64 lanes everywhere; MFMA = v_mfma_f32_32x32x2_f32 (exact fp32 fmaf
// chain, 157 TFLOP/s dense peak — MI355X_MICROARCH.md "Peak FP32 (matrix)").
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

#define DIM_WAVE 64

// D = A(32x2) * B(2x32) + C on one wave.
//   A operand: lane l supplies A[i = l&31][k = l>>5]
//   B operand: lane l supplies B[k = l>>5][j = l&31]
//   C/D     : lane l, reg r holds D[row = (r&3) + 8*(r>>2) + 4*(l>>5)][col = l&31]
__device__ __forceinline__ f32x16 mfma32(float a, float b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}
// D = A(32x16) * B(16x32) + C with bf16 inputs (v_mfma_f32_32x32x16_bf16, 32 cycles/SIMD, 16x the
// fp32 MFMA rate).  Operands are 8 bf16 packed in 4 dwords:
//   A: lane l supplies A[i = l&31][k = 8*(l>>5) + 0..7],  B: lane l supplies B[k = 8*(l>>5) + 0..7][j = l&31]
// (lane maps verified on hardware by scripts/probe/mfma_bf16_probe.hip); C/D as mfma32.
__device__ __forceinline__ f32x16 mfma_bf16(u32x4 a, u32x4 b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}
// 3-way split of two fp32 values into packed bf16 pieces with round-to-nearest-even conversions
// (v_cvt_pk_bf16_f32): x = h + m + l up to 2^-27 |x| (each residual x - h, r1 - m is exact in fp32).
// Products of the pieces are exact in fp32 and the six leading cross terms hh, hm, mh, hl, lh, mm
// carry the product to ~2^-26: fp32-class accuracy (hardware probe: 1.3e-7 of sum|a*b| at K = 1024,
// an fp32 fmaf chain gives 1.2e-7) at 6/16 of the fp32-MFMA cost.  Low half of a dword = first value.
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned cvt_pk_bf16(float a, float b) {
  f32x2 v = {a, b};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
}
__device__ __forceinline__ void split3_pk(float x0, float x1, unsigned& h, unsigned& m, unsigned& l) {
  h = cvt_pk_bf16(x0, x1);
  const float r0 = x0 - __uint_as_float(h << 16), r1 = x1 - __uint_as_float(h & 0xffff0000u);
  m = cvt_pk_bf16(r0, r1);
  const float s0 = r0 - __uint_as_float(m << 16), s1 = r1 - __uint_as_float(m & 0xffff0000u);
  l = cvt_pk_bf16(s0, s1);
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
