// Probe: lane->element maps and accumulation accuracy of v_mfma_f32_32x32x16_bf16 on gfx950.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
#include <string.h>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
__device__ __bf16 f2bf_trunc(float x) { unsigned u = __float_as_uint(x) >> 16; unsigned short s = (unsigned short)u; __bf16 r; memcpy(&r, &s, 2); return r; }
__global__ void probe(const float* A, const float* B, float* D) {  // A [32][16], B [16][32] (values exactly bf16-representable)
  const int l = threadIdx.x, i = l & 31, kb = l >> 5;
  bf16x8 a, b;
  for (int e = 0; e < 8; ++e) { a[e] = f2bf_trunc(A[i * 16 + kb * 8 + e]); b[e] = f2bf_trunc(B[(kb * 8 + e) * 32 + i]); }
  f32x16 c; for (int r = 0; r < 16; ++r) c[r] = 0.f;
  c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
  for (int r = 0; r < 16; ++r) { int row = (r & 3) + 8 * (r >> 2) + 4 * kb; D[row * 32 + i] = c[r]; }
}
// accuracy: 6-term split product of fp32 values vs fp64
__global__ void split6(const float* A, const float* B, float* D, int K) {  // A [32][K], B [K][32]
  const int l = threadIdx.x, i = l & 31, kb = l >> 5;
  f32x16 c; for (int r = 0; r < 16; ++r) c[r] = 0.f;
  for (int k0 = 0; k0 < K; k0 += 16) {
    bf16x8 a[3], b[3];
    for (int e = 0; e < 8; ++e) {
      float x = A[i * K + k0 + kb * 8 + e], y = B[(k0 + kb * 8 + e) * 32 + i];
      for (int p = 0; p < 3; ++p) { unsigned ux = __float_as_uint(x) & 0xffff0000u; a[p][e] = f2bf_trunc(x); x = x - __uint_as_float(ux);
                                    unsigned uy = __float_as_uint(y) & 0xffff0000u; b[p][e] = f2bf_trunc(y); y = y - __uint_as_float(uy); }
    }
    // smallest terms first
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[1], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[2], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2], b[0], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[1], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[0], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[0], c, 0, 0, 0);
  }
  for (int r = 0; r < 16; ++r) { int row = (r & 3) + 8 * (r >> 2) + 4 * kb; D[row * 32 + i] = c[r]; }
}
int main() {
  const int K = 1024;
  float *hA = (float*)malloc(32 * K * 4), *hB = (float*)malloc(K * 32 * 4), *hD = (float*)malloc(32 * 32 * 4);
  // layout probe with small integers (exact in bf16), asymmetric
  for (int i = 0; i < 32; ++i) for (int k = 0; k < 16; ++k) hA[i * 16 + k] = (float)((i * 3 + k * 7) % 13 - 6);
  for (int k = 0; k < 16; ++k) for (int j = 0; j < 32; ++j) hB[k * 32 + j] = (float)((k * 5 + j * 11) % 17 - 8);
  float *dA, *dB, *dD; hipMalloc(&dA, 32 * K * 4); hipMalloc(&dB, K * 32 * 4); hipMalloc(&dD, 32 * 32 * 4);
  hipMemcpy(dA, hA, 32 * 16 * 4, hipMemcpyHostToDevice); hipMemcpy(dB, hB, 16 * 32 * 4, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, dA, dB, dD); hipMemcpy(hD, dD, 32 * 32 * 4, hipMemcpyDeviceToHost);
  int bad = 0; for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) { float s = 0; for (int k = 0; k < 16; ++k) s += hA[i * 16 + k] * hB[k * 32 + j]; if (s != hD[i * 32 + j]) ++bad; }
  printf("layout probe: %d mismatches of 1024\n", bad);
  // accuracy probe
  srand(1); for (int i = 0; i < 32 * K; ++i) { hA[i] = (float)rand() / RAND_MAX * 2 - 1; hB[i] = (float)rand() / RAND_MAX * 2 - 1; }
  hipMemcpy(dA, hA, 32 * K * 4, hipMemcpyHostToDevice); hipMemcpy(dB, hB, K * 32 * 4, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(split6, dim3(1), dim3(64), 0, 0, dA, dB, dD, K); hipMemcpy(hD, dD, 32 * 32 * 4, hipMemcpyDeviceToHost);
  double emax = 0, e32max = 0, smax = 0;
  for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) { double s = 0, sa = 0; float f = 0; for (int k = 0; k < K; ++k) { double p = (double)hA[i * K + k] * hB[k * 32 + j]; s += p; sa += fabs(p); f = fmaf(hA[i * K + k], hB[k * 32 + j], f); }
    double e = fabs(hD[i * 32 + j] - s) / sa, e32 = fabs((double)f - s) / sa; if (e > emax) emax = e; if (e32 > e32max) e32max = e32; if (sa > smax) smax = sa; }
  printf("split6 bf16 MFMA: max err/sum|ab| = %.3e ; fp32 fmaf chain: %.3e (K=%d)\n", emax, e32max, K);
  return 0;
}
