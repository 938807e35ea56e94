// Does the ORDER in which a wave issues the cross terms of a split-precision tile change what the matrix cores sustain under the power cap?  (round 5)
// One wave holds the fragments of the production convolution step (conv_x6.hip, MR 4): A = 4 pixel rows x {h, l}, B = 2 channel slabs x {h, l}, 8 accumulators,
// 24 v_mfma_f32_32x32x16_f16 per step (three cross terms per accumulator, always issued l.h -> h.l -> h.h PER ACCUMULATOR: every order below gives the same bits).
//   order 0: production — term outer, row, slab inner: the A register changes every 2nd MFMA, B every MFMA
//   order 1: B-stationary — slab, term, row inner: B changes every 4th MFMA, A every MFMA
//   order 2: A-stationary — row, term, slab inner: A changes every 2nd / 4th MFMA (Ah[m] serves 4 in a row), B every MFMA
//   order 3: one fixed (A, B) register pair for all 24 (the floor of operand toggling at these values; not a usable schedule)
// Operands: the "relu-split" set of mfma_power_probe.hip (post-ReLU activations x 16 split h / l, weights scaled to [8192, 16384) split h / l).
//   hipcc --offload-arch=gfx950 -O3 -o order_probe mfma_operand_order_probe.hip && ./order_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float float16v __attribute__((ext_vector_type(16)));

#define MFMA(acc, a, b) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b))

template <int ORDER>
__global__ __launch_bounds__(256, 2) void probe(const half8* __restrict__ src, float* __restrict__ out, int iters, unsigned long long* cyc) {
  const int t = threadIdx.x;
  half8 a[4][2], b[2][2];   // [row / slab][plane: 0 = h, 1 = l]
  for (int m = 0; m < 4; ++m) for (int p = 0; p < 2; ++p) a[m][p] = src[((m * 2 + p) * 256 + t) & 4095];
  for (int n = 0; n < 2; ++n) for (int p = 0; p < 2; ++p) b[n][p] = src[((8 + n * 2 + p) * 256 + t) & 4095];
  float16v c[4][2];
  for (int m = 0; m < 4; ++m) for (int n = 0; n < 2; ++n) for (int j = 0; j < 16; ++j) c[m][n][j] = 0.0f;
  // term k of an accumulator: (A plane, B plane) = (l, h), (h, l), (h, h)
  constexpr int TA[3] = {1, 0, 0}, TB[3] = {0, 1, 0};
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
    if (ORDER == 0) {
#pragma unroll
      for (int k = 0; k < 3; ++k)
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
          for (int n = 0; n < 2; ++n) MFMA(c[m][n], a[m][TA[k]], b[n][TB[k]]);
    } else if (ORDER == 1) {
#pragma unroll
      for (int n = 0; n < 2; ++n)
#pragma unroll
        for (int k = 0; k < 3; ++k)
#pragma unroll
          for (int m = 0; m < 4; ++m) MFMA(c[m][n], a[m][TA[k]], b[n][TB[k]]);
    } else if (ORDER == 2) {
#pragma unroll
      for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int k = 0; k < 3; ++k)
#pragma unroll
          for (int n = 0; n < 2; ++n) MFMA(c[m][n], a[m][TA[k]], b[n][TB[k]]);
    } else {
#pragma unroll
      for (int k = 0; k < 3; ++k)
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
          for (int n = 0; n < 2; ++n) MFMA(c[m][n], a[0][0], b[0][0]);
    }
  }
  float s = 0.0f;
  for (int m = 0; m < 4; ++m) for (int n = 0; n < 2; ++n) for (int j = 0; j < 16; ++j) s += c[m][n][j];
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (blockIdx.x == 0 && t == 0) *cyc = t1 - t0;
  if (s == 123.456f) out[blockIdx.x * 256 + t] = s;
}

int main(int argc, char** argv) {
  const int iters = argc > 1 ? atoi(argv[1]) : 700, launches = argc > 2 ? atoi(argv[2]) : 480, group = 40;
  hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
  const int cus = p.multiProcessorCount;
  half8* src; float* out;
  hipMalloc(&src, 4096 * sizeof(half8)); hipMalloc(&out, (size_t)cus * 8 * 256 * 4);
  unsigned long long* cyc; hipMalloc(&cyc, 8);
  std::vector<_Float16> host(4096 * 8);
  auto uni = []() { return rand() / (float)RAND_MAX; };
  srand(1);
  for (int frag = 0; frag < 12; frag += 2)
    for (int e = 0; e < 256 * 8; ++e) {
      const float x = frag < 8 ? (uni() < 0.5f ? 0.0f : 16.0f * 2.0f * uni()) : (uni() - 0.5f) * 2.0f * 16384.0f;
      const _Float16 hi = (_Float16)x;
      host[(size_t)frag * 2048 + e] = hi;
      host[(size_t)(frag + 1) * 2048 + e] = (_Float16)(x - (float)hi);
    }
  hipMemcpy(src, host.data(), host.size() * 2, hipMemcpyHostToDevice);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const char* names[4] = {"production: A every 2nd, B every MFMA", "B-stationary: B every 4th, A every MFMA", "A-stationary: A every 2nd-4th, B every MFMA", "one fixed register pair (floor)"};
  for (int rep = 0; rep < 2; ++rep)
    for (int order = 0; order < 4; ++order) {
      const int grid = cus * 2;   // two workgroups per CU = two waves per SIMD, as the convolutions run
      hipDeviceSynchronize();
      for (int g = 0; g < launches / group; ++g) {
        hipEventRecord(e0);
        for (int l = 0; l < group; ++l) {
          if (order == 0) hipLaunchKernelGGL(probe<0>, dim3(grid), dim3(256), 0, 0, src, out, iters, cyc);
          else if (order == 1) hipLaunchKernelGGL(probe<1>, dim3(grid), dim3(256), 0, 0, src, out, iters, cyc);
          else if (order == 2) hipLaunchKernelGGL(probe<2>, dim3(grid), dim3(256), 0, 0, src, out, iters, cyc);
          else hipLaunchKernelGGL(probe<3>, dim3(grid), dim3(256), 0, 0, src, out, iters, cyc);
        }
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double mfma_per_simd = (double)iters * 24 * 2 * group, flops = mfma_per_simd * 32768.0 * cus * 4;
        unsigned long long hc = 0; hipMemcpy(&hc, cyc, 8, hipMemcpyDeviceToHost);
        if (g == launches / group - 1 || g == launches / group / 2)
          printf("{\"order\": %d, \"what\": \"%s\", \"rep\": %d, \"group\": %d, \"ms_per_launch\": %.4f, \"tflops\": %.1f, \"effective_clock_ghz_if_never_idle\": %.3f, \"wave_cycles_per_mfma\": %.2f}\n", order, names[order], rep, g,
                 ms / group, flops / (ms * 1e-3) / 1e12, mfma_per_simd * 32.0 / (ms * 1e-3) / 1e9, (double)hc / ((double)iters * 24));
      }
    }
  return 0;
}
