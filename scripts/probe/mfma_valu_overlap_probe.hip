// Probe (round 4): how do MFMA and VALU work share one SIMD on gfx950?  The round-4 counters of the convolution kernels show
// MFMA-busy + VALU-active ~ 1 per SIMD (no overlap).  Variants, all with 2 workgroups x 4 waves per CU (= 2 waves per SIMD),
// per wave and period NM MFMAs (8 independent accumulators) and NV dependent-free VALU FMAs:
//   0  phased, in lockstep:      every wave runs [NM MFMAs][NV VALU] per period
//   1  phased, anti-phase:       the second workgroup of each CU runs [NV VALU][NM MFMAs] (starts with the VALU block)
//   2  interleaved in one wave:  per period NM x { 1 MFMA, NV / NM VALU } in program order
//   3  MFMA only     4  VALU only
// prints the time per period per variant: lockstep ~ sum means no cross-wave overlap; anti-phase / interleaved ~ max means the
// pipes do overlap when the instruction streams allow it.
//   hipcc --offload-arch=gfx950 -O3 scripts/probe/mfma_valu_overlap_probe.hip -o /tmp/ovl && /tmp/ovl
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
constexpr int NM = 24, NV = 192;

template <int VAR>
__global__ __launch_bounds__(256, 2) void k(float* out, int periods) {
  f16x8 a, b;
  for (int e = 0; e < 8; ++e) { a[e] = (_Float16)(threadIdx.x * 0.001f + e); b[e] = (_Float16)(e * 0.5f); }
  f32x16 c[8];
  for (int i = 0; i < 8; ++i) for (int r = 0; r < 16; ++r) c[i][r] = 0.f;
  float v[16];
  for (int i = 0; i < 16; ++i) v[i] = threadIdx.x * 0.01f + i;
  const float m1 = 1.0001f, m2 = 0.0003f;
  // which workgroup of the CU am I?  the first 256 workgroups fill slot 0 of every CU, the next 256 slot 1 (dispatch order)
  const bool second = __builtin_amdgcn_readfirstlane((int)((blockIdx.x >> 8) & 1)) != 0;   // scalar: a real branch, not EXEC masking of both paths
  auto mfmas = [&]() {
#pragma unroll
    for (int i = 0; i < NM; ++i) c[i & 7] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c[i & 7], 0, 0, 0);
  };
  auto valus = [&]() {
#pragma unroll
    for (int i = 0; i < NV; ++i) v[i & 15] = __builtin_fmaf(v[i & 15], m1, m2);
  };
  for (int p = 0; p < periods; ++p) {
    if (VAR == 0) { mfmas(); __builtin_amdgcn_sched_barrier(0); valus(); __builtin_amdgcn_sched_barrier(0); }
    if (VAR == 1) {
      if (second) { valus(); __builtin_amdgcn_sched_barrier(0); mfmas(); __builtin_amdgcn_sched_barrier(0); }
      else { mfmas(); __builtin_amdgcn_sched_barrier(0); valus(); __builtin_amdgcn_sched_barrier(0); }
    }
    if (VAR == 2) {
#pragma unroll
      for (int i = 0; i < NM; ++i) {
        c[i & 7] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c[i & 7], 0, 0, 0);
#pragma unroll
        for (int j = 0; j < NV / NM; ++j) v[(i * (NV / NM) + j) & 15] = __builtin_fmaf(v[(i * (NV / NM) + j) & 15], m1, m2);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    if (VAR == 3) { mfmas(); __builtin_amdgcn_sched_barrier(0); }
    if (VAR == 4) { valus(); __builtin_amdgcn_sched_barrier(0); }
  }
  float s = 0;
  for (int i = 0; i < 8; ++i) for (int r = 0; r < 16; ++r) s += c[i][r];
  for (int i = 0; i < 16; ++i) s += v[i];
  if (s == 12345.678f) out[0] = s;
}

// the same with 4 accumulators (<= 128 registers) and 4 workgroups per CU = 4 waves per SIMD: VAR 0 phased lockstep, 2 interleaved
template <int VAR>
__global__ __launch_bounds__(256, 4) void k4(float* out, int periods) {
  f16x8 a, b;
  for (int e = 0; e < 8; ++e) { a[e] = (_Float16)(threadIdx.x * 0.001f + e); b[e] = (_Float16)(e * 0.5f); }
  f32x16 c[4];
  for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) c[i][r] = 0.f;
  float v[16];
  for (int i = 0; i < 16; ++i) v[i] = threadIdx.x * 0.01f + i;
  const float m1 = 1.0001f, m2 = 0.0003f;
  for (int p = 0; p < periods; ++p) {
    if (VAR == 0) {
#pragma unroll
      for (int i = 0; i < NM; ++i) c[i & 3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c[i & 3], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < NV; ++i) v[i & 15] = __builtin_fmaf(v[i & 15], m1, m2);
      __builtin_amdgcn_sched_barrier(0);
    } else {
#pragma unroll
      for (int i = 0; i < NM; ++i) {
        c[i & 3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c[i & 3], 0, 0, 0);
#pragma unroll
        for (int j = 0; j < NV / NM; ++j) v[(i * (NV / NM) + j) & 15] = __builtin_fmaf(v[(i * (NV / NM) + j) & 15], m1, m2);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  }
  float s = 0;
  for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += c[i][r];
  for (int i = 0; i < 16; ++i) s += v[i];
  if (s == 12345.678f) out[0] = s;
}
template <int VAR>
float run4(float* d, int periods) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k4<VAR>, dim3(1024), dim3(256), 0, 0, d, 10);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(k4<VAR>, dim3(1024), dim3(256), 0, 0, d, periods);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  return ms;
}

template <int VAR>
float run(float* d, int periods) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k<VAR>, dim3(512), dim3(256), 0, 0, d, 10);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<VAR>, dim3(512), dim3(256), 0, 0, d, periods);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  return ms;
}
int main() {
  float* d;
  hipMalloc(&d, 4);
  const int P = 20000;
  const char* names[5] = {"phased lockstep", "phased anti-phase", "interleaved in-wave", "MFMA only", "VALU only"};
  float ms[5] = {run<0>(d, P), run<1>(d, P), run<2>(d, P), run<3>(d, P), run<4>(d, P)};
  for (int i = 0; i < 5; ++i)
    printf("{\"variant\": \"%s\", \"ns_per_period\": %.1f, \"mfma_per_period\": %d, \"valu_per_period\": %d}\n", names[i], ms[i] * 1e6 / P, i == 4 ? 0 : NM, i == 3 ? 0 : NV);
  // 4 waves per SIMD (twice the work per SIMD and period: compare ns_per_period / 2 with the rows above)
  const float a4 = run4<0>(d, P), b4 = run4<2>(d, P);
  printf("{\"variant\": \"4 waves per SIMD, phased lockstep\", \"ns_per_period_per_2_waves\": %.1f}\n", a4 * 1e6 / P / 2);
  printf("{\"variant\": \"4 waves per SIMD, interleaved in-wave\", \"ns_per_period_per_2_waves\": %.1f}\n", b4 * 1e6 / P / 2);
  return 0;
}
