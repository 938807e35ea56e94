// Probe (round 5): does single-issue work HIDE in the gap behind a v_mfma_f32_32x32x16_f16 when ONE wave owns a SIMD?
// MI355X_MICROARCH.md (instruction-timing table) says <= 5 single-issue instructions hide per MFMA gap at one wave per SIMD (512-register
// kernel); round 4's probe (mfma_valu_overlap_probe.hip) only covered 2 and 4 waves per SIMD, one filler class, 8 fillers per MFMA, and found
// "time = MFMA + VALU".  This probe pins the instruction stream with `asm volatile` (program order = issue order; the ISA of the loop body is
// committed next to the results):
//   per period 32 x { 1 MFMA (8 rotating accumulators in AGPRs) , NF fillers of class CLS (16 rotating, independent destinations) }
//   NF in {0, 2, 4, 5, 8};  CLS: 0 v_fma_f32, 1 v_cvt_pk_f16_f32, 2 v_exp_f32, 3 ds_read_b128, 4 v_fma_mix_f32 (the fp16 split's residual),
//                                 5 v_pk_fma_f32, 6 v_perm_b32, 7 v_max3_f32
//   WPS = waves per SIMD: 1 (one 256-thread workgroup per CU, forced by 100 KB of LDS) or 2 (two workgroups per CU)
// and a filler-only stream (no MFMA) per class for the issue cost of the filler itself.  Time: s_memtime per wave (cycles per MFMA slot) and
// HIP events (clock).  Operands are zero / tiny so the package stays below its power limit and the clock stays near 2.4 GHz.
//   hipcc --offload-arch=gfx950 -O3 scripts/probe/mfma_filler_1wave_probe.hip -o scripts/probe/filler_probe && scripts/probe/filler_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <string.h>
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;
constexpr int NM = 32;

template <int CLS>
__device__ __forceinline__ void filler(float (&v)[16], f32x4 (&q)[8], f32x2 (&p)[8], int i, float m1, float m2, unsigned lds_addr) {
  float& d = v[i & 15];
  if constexpr (CLS == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(d) : "v"(m1), "v"(m2));
  if constexpr (CLS == 1) asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(d) : "v"(m1), "v"(m2));
  if constexpr (CLS == 2) asm volatile("v_exp_f32 %0, %1" : "=v"(d) : "v"(m2));
  if constexpr (CLS == 3) asm volatile("ds_read_b128 %0, %1" : "=v"(q[i & 7]) : "v"(lds_addr));
  if constexpr (CLS == 4) asm volatile("v_fma_mix_f32 %0, %1, %2, %3 op_sel_hi:[1,0,0]" : "=v"(d) : "v"(m1), "v"(m2), "v"(m1));
  if constexpr (CLS == 5) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(p[i & 7]) : "v"(p[(i + 4) & 7]));
  if constexpr (CLS == 6) asm volatile("v_perm_b32 %0, %1, %2, %3" : "=v"(d) : "v"(m1), "v"(m2), "v"(lds_addr));
  if constexpr (CLS == 7) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(d) : "v"(m1), "v"(m2));
}

// MF = 1: MFMAs + fillers with the accumulators in AGPRs; MF = 2: the same with the accumulators in VGPRs (what hipcc chooses by itself for a kernel
// that fits 256 registers: every production kernel of csrc/ — 8424 of 8856 MFMAs of conv_x6.hip are the VGPR form); MF = 0: fillers only
template <int CLS, int NF, int WPS, int MF>
__global__ __launch_bounds__(256, WPS) void k(float* out, unsigned long long* cyc, int periods, int operands) {
  extern __shared__ float lds[];
  for (int i = threadIdx.x; i < 4096; i += 256) lds[i] = 0.f;
  __syncthreads();
  // operands == 0: zeros (the multipliers do not toggle: the package stays far below its power cap and the clock near 2.4 GHz — issue
  // behaviour in isolation); operands == 1: pseudo-random fp16 values, two alternating operand sets (every MFMA sees new inputs): the regime
  // of the production kernels, where the firmware holds the package at its power cap by lowering the clock
  f16x8 a, b, a2, b2;
  unsigned lcg = threadIdx.x * 2654435761u + blockIdx.x * 40503u + 12345u;
  for (int e = 0; e < 8; ++e) {
    float r[4];
    for (int q = 0; q < 4; ++q) { lcg = lcg * 1664525u + 1013904223u; r[q] = operands ? ((int)(lcg >> 8) % 4096 - 2048) * (1.0f / 1024.0f) : 0.f; }
    a[e] = (_Float16)r[0]; b[e] = (_Float16)r[1]; a2[e] = (_Float16)r[2]; b2[e] = (_Float16)r[3];
  }
  f32x16 c[8];
  for (int i = 0; i < 8; ++i) for (int r = 0; r < 16; ++r) c[i][r] = 0.f;
  float v[16];
  f32x4 q[8];
  f32x2 p[8];
  for (int i = 0; i < 16; ++i) v[i] = threadIdx.x * 1e-6f;
  for (int i = 0; i < 8; ++i) { q[i] = f32x4{0, 0, 0, 0}; p[i] = f32x2{1e-6f, 1e-6f}; }
  float m1 = operands ? -0.9990234f : 1.0f, m2 = operands ? 0.37f : 1e-6f;     // (random mode: a sign-alternating contraction: the filler registers keep toggling, bounded)
  asm volatile("" : "+v"(m1), "+v"(m2));
  const unsigned lds_addr = (threadIdx.x & 63) * 16;     // conflict-free b128 pattern, 1 KB per wave
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int pp = 0; pp < periods; ++pp) {
#pragma unroll
    for (int i = 0; i < NM; ++i) {
      if constexpr (MF == 1) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(c[i & 7]) : "v"((i & 1) ? a2 : a), "v"((i & 1) ? b2 : b));
      if constexpr (MF == 2) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(c[i & 7]) : "v"((i & 1) ? a2 : a), "v"((i & 1) ? b2 : b));
#pragma unroll
      for (int j = 0; j < NF; ++j) filler<CLS>(v, q, p, i * NF + j, m1, m2, lds_addr);
    }
    if constexpr (CLS == 3) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0;
  for (int i = 0; i < 8; ++i) for (int r = 0; r < 16; ++r) s += c[i][r];
  for (int i = 0; i < 16; ++i) s += v[i];
  for (int i = 0; i < 8; ++i) s += q[i][0] + q[i][3] + p[i][0] + p[i][1];
  if (s == 12345.678f) out[0] = s;
  if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}

static float* d_out;
static unsigned long long* d_cyc;
static unsigned long long h_cyc[4096];
static const char* CLSN[8] = {"v_fma_f32", "v_cvt_pk_f16_f32", "v_exp_f32", "ds_read_b128", "v_fma_mix_f32", "v_pk_fma_f32", "v_perm_b32", "v_max3_f32"};

static int g_operands = 0;
template <int CLS, int NF, int WPS, int MF>
void run(int periods) {
  // WPS 1: 100 KB of LDS per workgroup -> one workgroup per CU, grid = 256 (one round); WPS 2: 64 KB -> two per CU, grid = 512
  const int lds = WPS == 1 ? 100 * 1024 : 64 * 1024;
  const int grid = 256 * WPS;
  hipFuncSetAttribute((const void*)k<CLS, NF, WPS, MF>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((k<CLS, NF, WPS, MF>), dim3(grid), dim3(256), lds, 0, d_out, d_cyc, 50, g_operands);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL((k<CLS, NF, WPS, MF>), dim3(grid), dim3(256), lds, 0, d_out, d_cyc, periods, g_operands);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  hipMemcpy(h_cyc, d_cyc, sizeof(unsigned long long) * grid * 4, hipMemcpyDeviceToHost);
  double sum = 0, mx = 0;
  for (int i = 0; i < grid * 4; ++i) { sum += (double)h_cyc[i]; if ((double)h_cyc[i] > mx) mx = (double)h_cyc[i]; }
  const double slots = (double)periods * NM;
  // both clocks are reported: s_memtime ticks per slot (average and slowest wave) and the event time per slot; the MFMA-only row gives the
  // ticks <-> ns relation of the run (32 shader cycles per slot)
  printf("{\"operands\": \"%s\", \"class\": \"%s\", \"fillers_per_mfma\": %d, \"waves_per_simd\": %d, \"mfma\": %d, \"ns_per_slot\": %.3f, \"memtime_ticks_per_slot_avg\": %.4f, "
         "\"memtime_ticks_per_slot_max\": %.4f, \"ms\": %.4f}\n",
         g_operands ? "random" : "zero", CLSN[CLS], NF, WPS, MF, ms * 1e6 / slots, sum / (grid * 4) / slots, mx / slots, ms);
  fflush(stdout);
}

template <int CLS, int WPS>
void sweep(int P) {
  run<CLS, 2, WPS, 1>(P); run<CLS, 4, WPS, 1>(P); run<CLS, 5, WPS, 1>(P); run<CLS, 8, WPS, 1>(P);
  run<CLS, 4, WPS, 0>(P); run<CLS, 8, WPS, 0>(P);
}

int main(int argc, char** argv) {
  hipMalloc(&d_out, 4);
  hipMalloc(&d_cyc, sizeof(unsigned long long) * 4096);
  const int P = argc > 1 ? atoi(argv[1]) : 4000;
  if (argc > 2 && atoi(argv[2]) == 1) {
    // the power-limited regime: long runs (P ~ 400 000: 0.2 - 0.5 s each, the firmware's time scale) on toggling operands, the production
    // occupancy (two waves per SIMD).  Cycles per slot (s_memtime) say whether the fillers still hide; ns per slot what that is worth in time.
    g_operands = 1;
    run<0, 0, 2, 1>(P); run<0, 5, 2, 0>(P); run<0, 5, 2, 1>(P); run<0, 2, 2, 1>(P); run<4, 4, 2, 0>(P); run<4, 4, 2, 1>(P); run<0, 8, 2, 1>(P);
    run<0, 0, 1, 1>(P); run<0, 5, 1, 1>(P);
    g_operands = 0;
    run<0, 0, 2, 1>(P); run<0, 5, 2, 1>(P);
    return 0;
  }
  run<0, 0, 1, 1>(P);      // MFMA only, one wave per SIMD: the floor (32 cycles per slot)
  run<0, 0, 2, 1>(P);      // MFMA only, two waves per SIMD
  sweep<0, 1>(P); sweep<1, 1>(P); sweep<2, 1>(P); sweep<3, 1>(P); sweep<4, 1>(P); sweep<5, 1>(P); sweep<6, 1>(P); sweep<7, 1>(P);
  sweep<0, 2>(P); sweep<2, 2>(P); sweep<3, 2>(P); sweep<4, 2>(P);
  // accumulators in VGPRs (the form the compiler picks for the production kernels) instead of AGPRs: MFMA only, then with fillers
  run<0, 0, 1, 2>(P); run<0, 2, 1, 2>(P); run<0, 4, 1, 2>(P); run<0, 5, 1, 2>(P); run<0, 8, 1, 2>(P);
  run<4, 2, 1, 2>(P); run<4, 4, 1, 2>(P); run<4, 8, 1, 2>(P);
  run<0, 0, 2, 2>(P); run<0, 2, 2, 2>(P); run<0, 4, 2, 2>(P); run<0, 5, 2, 2>(P); run<0, 8, 2, 2>(P);
  run<4, 4, 2, 2>(P); run<4, 8, 2, 2>(P); run<3, 2, 2, 2>(P);
  return 0;
}
