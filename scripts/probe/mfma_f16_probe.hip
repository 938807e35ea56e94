// Probe: v_mfma_f32_32x32x16_f16 on gfx950 — lane->element map, fp16 subnormal inputs, accuracy of the
// 2-way fp16 split x 3 cross terms ("fp16x3") against fp64, the fp32 fmaf chain and bf16x6.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
#include <string.h>
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
__global__ void probe(const float* A, const float* B, float* D) {  // A [32][16], B [16][32], values exactly fp16-representable
  const int l = threadIdx.x, i = l & 31, kb = l >> 5;
  f16x8 a, b;
  for (int e = 0; e < 8; ++e) { a[e] = (_Float16)A[i * 16 + kb * 8 + e]; b[e] = (_Float16)B[(kb * 8 + e) * 32 + i]; }
  f32x16 c; for (int r = 0; r < 16; ++r) c[r] = 0.f;
  c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
  for (int r = 0; r < 16; ++r) { int row = (r & 3) + 8 * (r >> 2) + 4 * kb; D[row * 32 + i] = c[r]; }
}
__device__ __forceinline__ void split2(float x, float scale, _Float16& h, _Float16& l) {
  x = __builtin_amdgcn_fmed3f(x * scale, -65504.f, 65504.f);
  h = (_Float16)x;
  l = (_Float16)(x - (float)h);
}
__global__ void split3(const float* A, const float* B, float* D, int K, float sa, float sb) {  // A [32][K], B [K][32]
  const int l = threadIdx.x, i = l & 31, kb = l >> 5;
  f32x16 c; for (int r = 0; r < 16; ++r) c[r] = 0.f;
  for (int k0 = 0; k0 < K; k0 += 16) {
    f16x8 ah, al, bh, bl;
    for (int e = 0; e < 8; ++e) {
      _Float16 h, lo;
      split2(A[i * K + k0 + kb * 8 + e], sa, h, lo); ah[e] = h; al[e] = lo;
      split2(B[(k0 + kb * 8 + e) * 32 + i], sb, h, lo); bh[e] = h; bl[e] = lo;
    }
    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, c, 0, 0, 0);
  }
  const float inv = 1.0f / (sa * sb);
  for (int r = 0; r < 16; ++r) { int row = (r & 3) + 8 * (r >> 2) + 4 * kb; D[row * 32 + i] = c[r] * inv; }
}
template <int F16>
__global__ void rate(float* out, int iters) {
  f16x8 a, b; bf16x8 a2, b2;
  for (int e = 0; e < 8; ++e) { a[e] = (_Float16)(threadIdx.x * 0.001f + e); b[e] = (_Float16)(e * 0.5f); a2[e] = (__bf16)(float)a[e]; b2[e] = (__bf16)(float)b[e]; }
  f32x16 c0, c1, c2, c3;
  for (int r = 0; r < 16; ++r) c0[r] = c1[r] = c2[r] = c3[r] = 0.f;
  for (int i = 0; i < iters; ++i) {
    if (F16) { c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c0, 0, 0, 0); c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c1, 0, 0, 0);
               c2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c2, 0, 0, 0); c3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c3, 0, 0, 0); }
    else { c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, b2, c0, 0, 0, 0); c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, b2, c1, 0, 0, 0);
           c2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, b2, c2, 0, 0, 0); c3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, b2, c3, 0, 0, 0); }
  }
  float s = 0; for (int r = 0; r < 16; ++r) s += c0[r] + c1[r] + c2[r] + c3[r];
  if (s == 12345.678f) out[0] = s;
}
int main() {
  const int K = 1024;
  float *hA = (float*)malloc(32 * K * 4), *hB = (float*)malloc(K * 32 * 4), *hD = (float*)malloc(32 * 32 * 4);
  for (int i = 0; i < 32; ++i) for (int k = 0; k < 16; ++k) hA[i * 16 + k] = (float)((i * 3 + k * 7) % 13 - 6);
  for (int k = 0; k < 16; ++k) for (int j = 0; j < 32; ++j) hB[k * 32 + j] = (float)((k * 5 + j * 11) % 17 - 8);
  float *dA, *dB, *dD; hipMalloc(&dA, 32 * K * 4); hipMalloc(&dB, K * 32 * 4); hipMalloc(&dD, 32 * 32 * 4);
  hipMemcpy(dA, hA, 32 * 16 * 4, hipMemcpyHostToDevice); hipMemcpy(dB, hB, 16 * 32 * 4, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, dA, dB, dD); hipMemcpy(hD, dD, 32 * 32 * 4, hipMemcpyDeviceToHost);
  int bad = 0; for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) { float s = 0; for (int k = 0; k < 16; ++k) s += hA[i * 16 + k] * hB[k * 32 + j]; if (s != hD[i * 32 + j]) ++bad; }
  printf("layout probe (same map as the bf16 instruction): %d mismatches of 1024\n", bad);
  // subnormal inputs: A = 2^-20 (fp16 subnormal), B = 2^10 -> each product 2^-10, 16 of them = 2^-6
  for (int i = 0; i < 32 * 16; ++i) { hA[i] = ldexpf(1.f, -20); hB[i] = 1024.f; }
  hipMemcpy(dA, hA, 32 * 16 * 4, hipMemcpyHostToDevice); hipMemcpy(dB, hB, 16 * 32 * 4, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, dA, dB, dD); hipMemcpy(hD, dD, 32 * 32 * 4, hipMemcpyDeviceToHost);
  printf("subnormal fp16 inputs: got %g expected %g -> %s\n", hD[0], ldexp(1.0, -6), hD[0] == (float)ldexp(1.0, -6) ? "preserved" : "FLUSHED");
  // accuracy in several magnitude regimes
  const char* names[4] = {"uniform[-1,1] x uniform[-1,1]", "relu-like [0,2) x weights 0.05", "tiny acts 1e-3 x weights 0.05", "mixed 1e-4..1e+2 x weights 0.05"};
  for (int reg = 0; reg < 4; ++reg) {
    srand(1 + reg);
    for (int i = 0; i < 32 * K; ++i) {
      float u = (float)rand() / RAND_MAX, v = (float)rand() / RAND_MAX * 2 - 1;
      hA[i] = reg == 0 ? u * 2 - 1 : reg == 1 ? (u < 0.5f ? 0.f : u * 4 - 2) : reg == 2 ? u * 1e-3f : powf(10.f, u * 6 - 4) * (rand() & 1 ? 1.f : -1.f);
      hB[i] = reg == 0 ? v : v * 0.05f;
    }
    hipMemcpy(dA, hA, 32 * K * 4, hipMemcpyHostToDevice); hipMemcpy(dB, hB, K * 32 * 4, hipMemcpyHostToDevice);
    for (int sai = 0; sai < 3; ++sai) {
      const float sa = sai == 0 ? 1.f : sai == 1 ? 16.f : 256.f, sb = reg == 0 ? 1.f : 8192.f;
      hipLaunchKernelGGL(split3, dim3(1), dim3(64), 0, 0, dA, dB, dD, K, sa, sb); hipMemcpy(hD, dD, 32 * 32 * 4, hipMemcpyDeviceToHost);
      double emax = 0, e32max = 0;
      for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) { double s = 0, sabs = 0; float f = 0; for (int k = 0; k < K; ++k) { double p = (double)hA[i * K + k] * hB[k * 32 + j]; s += p; sabs += fabs(p); f = fmaf(hA[i * K + k], hB[k * 32 + j], f); }
        double e = fabs(hD[i * 32 + j] - s) / sabs, e32 = fabs((double)f - s) / sabs; if (e > emax) emax = e; if (e32 > e32max) e32max = e32; }
      printf("%-34s act scale %5g: fp16x3 max err/sum|ab| = %.3e ; fp32 fmaf chain %.3e\n", names[reg], sa, emax, e32max);
    }
  }
  // issue rate: 4 independent accumulators, one wave per SIMD on every CU
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int f16 = 0; f16 < 2; ++f16) {
    const int iters = 20000; float ms;
    for (int rep = 0; rep < 2; ++rep) {
      hipEventRecord(e0);
      if (f16) hipLaunchKernelGGL(rate<1>, dim3(256 * 4), dim3(64), 0, 0, dD, iters); else hipLaunchKernelGGL(rate<0>, dim3(256 * 4), dim3(64), 0, 0, dD, iters);
      hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
    }
    printf("%s 32x32x16 MFMA: %.1f TFLOP/s dense (1 wave/SIMD, 4 accumulators)\n", f16 ? "f16 " : "bf16", 1024.0 * iters * 4 * 32768.0 / (ms * 1e-3) / 1e12);
  }
  return 0;
}
