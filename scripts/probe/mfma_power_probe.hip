// What the matrix cores of this MI355X sustain under its power management (round 4).
// Every SIMD of the chip runs back-to-back independent v_mfma_f32_32x32x16_f16 (4 accumulators per wave, W waves per SIMD) for ~0.5 s;
// per-launch HIP-event times give the rate at the start and in the steady state.  Operands: all zero / random fp16 (data toggling costs power) /
// random + a ds_read_b128 per MFMA (what a staged kernel adds).  Effective clock = MFMAs per SIMD x 32 cycles / time (the pipe is never idle).
//   hipcc --offload-arch=gfx950 -O3 -o mfma_power_probe mfma_power_probe.hip && ./mfma_power_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float float16v __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

typedef __bf16 bf8 __attribute__((ext_vector_type(8)));
typedef float float4v __attribute__((ext_vector_type(4)));
// the other two instruction shapes at the same operand bits: bf16 32x32x16 and f16 16x16x32 (4 accumulators of 4 registers x 4 chains)
template <int INST>
__global__ __launch_bounds__(256) void probe_inst(const half8* __restrict__ src, float* __restrict__ out, int iters) {
  const int t = threadIdx.x;
  half8 a[4], b[4];
  for (int i = 0; i < 4; ++i) { a[i] = src[(i * 256 + t) & 2047]; b[i] = src[((i + 4) * 256 + t) & 2047]; }
  float s = 0.0f;
  if (INST == 1) {
    float16v c[4];
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 16; ++j) c[i][j] = 0.0f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < 8; ++i)
        c[i & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf8, a[i & 3]), __builtin_bit_cast(bf8, b[(i + (i >> 2)) & 3]), c[i & 3], 0, 0, 0);
    }
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 16; ++j) s += c[i][j];
  } else {
    float4v c[8];
    for (int i = 0; i < 8; ++i) for (int j = 0; j < 4; ++j) c[i][j] = 0.0f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < 16; ++i)   // 16 x (16x16x32 = 16384 flop, 16 cycles) = the work of 8 x 32x32x16
        c[i & 7] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[i & 3], b[(i + (i >> 2)) & 3], c[i & 7], 0, 0, 0);
    }
    for (int i = 0; i < 8; ++i) for (int j = 0; j < 4; ++j) s += c[i][j];
  }
  if (s == 123.456f) out[blockIdx.x * 256 + t] = s;
}

template <int MODE>   // 0: operands from the buffer (zero or random), 1: + one ds_read_b128 per MFMA feeding the A operand
__global__ __launch_bounds__(256) void probe(const half8* __restrict__ src, float* __restrict__ out, int iters) {
  __shared__ u32x4 lds[4 * 64 * 4];
  const int t = threadIdx.x, lane = t & 63, w = t >> 6;
  half8 a[4], b[4];
  for (int i = 0; i < 4; ++i) { a[i] = src[(i * 256 + t) & 2047]; b[i] = src[((i + 4) * 256 + t) & 2047]; }
  for (int i = 0; i < 4; ++i) lds[(w * 4 + i) * 64 + lane] = __builtin_bit_cast(u32x4, a[i]);
  __syncthreads();
  float16v c[4];
  for (int i = 0; i < 4; ++i) for (int j = 0; j < 16; ++j) c[i][j] = 0.0f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      half8 av = a[i & 3];
      if (MODE == 1) av = __builtin_bit_cast(half8, lds[(w * 4 + ((i + it) & 3)) * 64 + lane]);
      c[i & 3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(av, b[(i + (i >> 2)) & 3], c[i & 3], 0, 0, 0);
    }
  }
  float s = 0.0f;
  for (int i = 0; i < 4; ++i) for (int j = 0; j < 16; ++j) s += c[i][j];
  if (s == 123.456f) out[blockIdx.x * 256 + t] = s;
}

int main(int argc, char** argv) {
  const int iters = argc > 1 ? atoi(argv[1]) : 2048, launches = argc > 2 ? atoi(argv[2]) : 480, group = 40;
  hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
  const int cus = p.multiProcessorCount;
  half8* src; float* out;
  hipMalloc(&src, 2048 * sizeof(half8)); hipMalloc(&out, (size_t)cus * 8 * 256 * 4);
  std::vector<_Float16> host(2048 * 8);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  // operand sets: zero; random fp16; "split": what the fp16x3 kernels feed — A = high / low fp16 pieces (alternating fragments) of x 16 post-ReLU
  // activations (half of them zero), B = pieces of weights scaled to [8192, 16384)
  const char* names[3] = {"zero", "random", "relu-split"};
  auto uni = []() { return rand() / (float)RAND_MAX; };
  for (int data = 0; data < 3; ++data)
    for (int mode = 0; mode < 4; ++mode)
      for (int wps = 1; wps <= 2; ++wps) {
        if (mode >= 2 && wps == 1) continue;
        srand(1);
        for (size_t e = 0; e < host.size(); ++e) {
          const int frag = (int)(e / (256 * 8));   // fragments 0..3 = A, 4..7 = B
          if (data == 0) host[e] = (_Float16)0.0f;
          else if (data == 1) host[e] = (_Float16)((uni() - 0.5f) * 0.05f);
          else {
            float x = frag < 4 ? (uni() < 0.5f ? 0.0f : 16.0f * 2.0f * uni()) : (uni() - 0.5f) * 2.0f * 16384.0f;
            const _Float16 hi = (_Float16)x;
            host[e] = (frag & 1) ? (_Float16)(x - (float)hi) : hi;
          }
        }
        hipMemcpy(src, host.data(), host.size() * 2, hipMemcpyHostToDevice);
        const int grid = cus * wps;   // 256 threads = one wave per SIMD per workgroup
        hipDeviceSynchronize();
        for (int g = 0; g < launches / group; ++g) {
          hipEventRecord(e0);
          for (int l = 0; l < group; ++l) {
            if (mode == 1) hipLaunchKernelGGL(probe<1>, dim3(grid), dim3(256), 0, 0, src, out, iters);
            else if (mode == 0) hipLaunchKernelGGL(probe<0>, dim3(grid), dim3(256), 0, 0, src, out, iters);
            else if (mode == 2) hipLaunchKernelGGL(probe_inst<1>, dim3(grid), dim3(256), 0, 0, src, out, iters);
            else hipLaunchKernelGGL(probe_inst<2>, dim3(grid), dim3(256), 0, 0, src, out, iters);
          }
          hipEventRecord(e1); hipEventSynchronize(e1);
          float ms; hipEventElapsedTime(&ms, e0, e1);
          const double mfma_per_simd = (double)iters * 8 * wps * group, flops = mfma_per_simd * 32768.0 * cus * 4;
          if (g == 0 || g == launches / group - 1 || g == launches / group / 2)
            printf("{\"operands\": \"%s\", \"instruction\": \"%s\", \"lds_read_per_mfma\": %d, \"waves_per_simd\": %d, \"group\": %d, \"ms_per_launch\": %.4f, \"tflops\": %.1f, \"effective_clock_ghz\": %.3f}\n",
                   names[data], mode == 2 ? "32x32x16_bf16" : mode == 3 ? "16x16x32_f16" : "32x32x16_f16", mode == 1, wps, g, ms / group, flops / (ms * 1e-3) / 1e12, mfma_per_simd * 32.0 / (ms * 1e-3) / 1e9);
        }
      }
  return 0;
}
