// Batched ragged fp32 GEMM on the gfx950 matrix cores (v_mfma_f32_32x32x2_f32).
//
// Workgroup = 256 threads = 4 waves laid out 2(M) x 2(N); block tile 128x128,
// wave tile 64x64 = 2x2 MFMA tiles (64 accumulator VGPRs), K stepped in chunks
// of 32 through LDS.  LDS images:
//   A  : [128 rows][33]   (row stride 33 dwords -> the 32 lanes of a half-wave
//                          that read one k of 32 consecutive rows hit 32 banks)
//   B  : [32 k][128 cols] (lanes read consecutive columns: conflict free) or,
//        for the NT form, [128 cols][33] like A.
// Every operand read is one ds_read_b32 feeding one MFMA (64 cycles/SIMD), so
// LDS bandwidth is far from limiting; the kernel is MFMA-issue bound once two
// workgroups per CU overlap staging with math.
//
// Serves: SuperPoint 1x1 convolutions on NHWC maps (SPN:175,214) and every
// LightGlue linear layer / similarity product (LGN:153,158,189-209,266-272).
#include <math.h>

#include "dim_kernels.h"

namespace {
constexpr int BM = 128, BN = 128, KC = 32, AS = KC + 1;

template <int BT>
__global__ __launch_bounds__(256, 4) void gemm_mfma_kernel(GemmArgs a) {
  const int z = blockIdx.z;
  if (a.flag && a.flag[z >> a.flag_shift] != a.flag_eq) return;
  const int rows = a.rows ? a.rows[z * a.rows_mul + a.rows_off] * a.rows_scale : a.M;
  const int cols = (BT && a.cols) ? a.cols[z * a.cols_mul + a.cols_off] : a.N;
  const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN;
  if (m0 >= rows || n0 >= cols) return;

  __shared__ float As[BM * AS];
  __shared__ float Bs[BT ? BN * AS : KC * BN];

  const int t = threadIdx.x;
  const int lane = t & 63, wv = t >> 6, wm = wv >> 1, wn = wv & 1;
  const int lx = lane & 31, half = lane >> 5;

  const float* A0 = a.A0 + (size_t)(a.a_idx ? a.a_idx[z] : z) * a.strideA0;
  const float* A1 = a.A1 ? a.A1 + (size_t)z * a.strideA1 : nullptr;
  const float* B = a.B + (size_t)z * a.strideB;

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

  // Software pipeline: the global loads of chunk k+1 are issued before the MFMA loop of chunk k
  // and only consumed (written to LDS) after it, so HBM/L2 latency hides behind 64 MFMAs/wave.
  float4 ra[(BM * KC / 4) / 256], rb[(KC * BN / 4) / 256];
  auto load_chunk = [&](int k0) {
    const float* src; int ld, kk0;
    if (A1 == nullptr || k0 < a.ksplit) { src = A0; ld = a.lda0; kk0 = k0; }
    else { src = A1; ld = a.lda1; kk0 = k0 - a.ksplit; }
#pragma unroll
    for (int i = 0; i < (BM * KC / 4) / 256; ++i) {
      const int idx = t + 256 * i;
      const int row = idx >> 3, q = idx & 7;
      ra[i] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (m0 + row < rows) ra[i] = *(const float4*)(src + (size_t)(m0 + row) * ld + kk0 + q * 4);
    }
    if (BT) {
#pragma unroll
      for (int i = 0; i < (BN * KC / 4) / 256; ++i) {
        const int idx = t + 256 * i;
        const int col = idx >> 3, q = idx & 7;
        rb[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (n0 + col < cols) rb[i] = *(const float4*)(B + (size_t)(n0 + col) * a.ldb + k0 + q * 4);
      }
    } else {
#pragma unroll
      for (int i = 0; i < (KC * BN / 4) / 256; ++i) {
        const int idx = t + 256 * i;
        const int kk = idx >> 5, q = idx & 31;
        rb[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (n0 + q * 4 < a.ldb) rb[i] = *(const float4*)(B + (size_t)(k0 + kk) * a.ldb + n0 + q * 4);
      }
    }
  };
  auto store_chunk = [&]() {
#pragma unroll
    for (int i = 0; i < (BM * KC / 4) / 256; ++i) {
      const int idx = t + 256 * i;
      const int row = idx >> 3, q = idx & 7;
      float* d = &As[row * AS + q * 4];
      d[0] = ra[i].x; d[1] = ra[i].y; d[2] = ra[i].z; d[3] = ra[i].w;
    }
    if (BT) {
#pragma unroll
      for (int i = 0; i < (BN * KC / 4) / 256; ++i) {
        const int idx = t + 256 * i;
        const int col = idx >> 3, q = idx & 7;
        float* d = &Bs[col * AS + q * 4];
        d[0] = rb[i].x; d[1] = rb[i].y; d[2] = rb[i].z; d[3] = rb[i].w;
      }
    } else {
#pragma unroll
      for (int i = 0; i < (KC * BN / 4) / 256; ++i) {
        const int idx = t + 256 * i;
        const int kk = idx >> 5, q = idx & 31;
        *(float4*)&Bs[kk * BN + q * 4] = rb[i];
      }
    }
  };

  load_chunk(0);
  for (int k0 = 0; k0 < a.K; k0 += KC) {
    store_chunk();
    __syncthreads();
    if (k0 + KC < a.K) load_chunk(k0 + KC);
#pragma unroll 4
    for (int s = 0; s < KC / 2; ++s) {
      const int kk = 2 * s + half;
      const float a0 = As[(wm * 64 + lx) * AS + kk];
      const float a1 = As[(wm * 64 + 32 + lx) * AS + kk];
      float b0, b1;
      if (BT) {
        b0 = Bs[(wn * 64 + lx) * AS + kk];
        b1 = Bs[(wn * 64 + 32 + lx) * AS + kk];
      } else {
        b0 = Bs[kk * BN + wn * 64 + lx];
        b1 = Bs[kk * BN + wn * 64 + 32 + lx];
      }
      acc[0][0] = mfma32(a0, b0, acc[0][0]);
      acc[0][1] = mfma32(a0, b1, acc[0][1]);
      acc[1][0] = mfma32(a1, b0, acc[1][0]);
      acc[1][1] = mfma32(a1, b1, acc[1][1]);
    }
    __syncthreads();
  }

  // ---- epilogue: bias, residual, ReLU; lanes 0..31 write 32 consecutive floats ----
  float* C = a.C + (size_t)z * a.strideC;
  const float* R = a.R ? a.R + (size_t)z * a.strideR : nullptr;
  float vmax = 0.0f;  // fp16x3 range guard (dim_common.h) when the output feeds a split-precision kernel (a.sat != nullptr)
#pragma unroll
  for (int n = 0; n < 2; ++n) {
    const int col = n0 + wn * 64 + n * 32 + lx;
    if (col >= cols) continue;
    const float bv = a.bias ? a.bias[col] : 0.0f;
#pragma unroll
    for (int m = 0; m < 2; ++m) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = m0 + wm * 64 + m * 32 + mfma_row(r, half);
        if (row >= rows) continue;
        float v = acc[m][n][r] + bv;
        if (R) v += R[(size_t)row * a.ldr + col];
        if (a.relu == 1) v = fmaxf(v, 0.0f);
        else if (a.relu == 2) v = v <= 0.0f ? (exp_le0(v) - 1.0f) * 1.7580993408473768599402175208123f : v * 1.0507009873554804934193349852946f;
        C[(size_t)row * a.ldc + col] = v;
        if (a.sat) vmax = fmaxf(vmax, fabsf(v));
      }
    }
  }
  sat_report(a.sat, vmax);
}
}  // namespace

int launch_gemm(const GemmArgs& a, int batch, hipStream_t s) {
  DIM_REQUIRE(a.K % KC == 0, "gemm: K=%d must be a multiple of %d", a.K, KC);
  DIM_REQUIRE(a.A1 == nullptr || a.ksplit % KC == 0, "gemm: ksplit=%d must be a multiple of %d", a.ksplit, KC);
  DIM_REQUIRE(a.lda0 % 4 == 0 && a.ldb % 4 == 0 && (a.A1 == nullptr || a.lda1 % 4 == 0), "gemm: leading dims must be multiples of 4");
  if (batch <= 0 || a.M <= 0 || a.N <= 0) return 0;
  dim3 grid(cdiv(a.M, BM), cdiv(a.N, BN), batch);
  if (a.bt)
    hipLaunchKernelGGL(HIP_KERNEL_NAME(gemm_mfma_kernel<1>), grid, dim3(256), 0, s, a);
  else
    hipLaunchKernelGGL(HIP_KERNEL_NAME(gemm_mfma_kernel<0>), grid, dim3(256), 0, s, a);
  DIM_LAUNCH_CHECK();
  return 0;
}
