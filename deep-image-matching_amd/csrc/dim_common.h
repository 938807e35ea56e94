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
// ---- fp16 pieces ("fp16x3") -----------------------------------------------------------------------
// v_mfma_f32_32x32x16_f16 has the bf16 instruction's lane map and rate and keeps fp16 subnormal inputs
// (scripts/probe/mfma_f16_probe.hip).  A 2-way split x = h + l into fp16 pieces (11 + 11 mantissa bits)
// with the three cross terms lh, hl, hh carries a product to ~2^-22 of |x y|: measured on hardware
// 0.7-1.7e-7 of sum|a*b| at K = 1024 where the fp32 fmaf chain gives 1.2-4.5e-7 — fp32-class at 3 instead
// of 6 MFMA passes.  fp16's narrow exponent is handled by power-of-two scales (exact): activations are
// multiplied by DIM_F16_ACT_SCALE before the split and clamped to +-65504 (|x| up to 4094 is exact in
// range; below ~2e-3 the low piece turns subnormal and the element's relative accuracy decays towards
// 2^-11 while its absolute error stays <= 2^-29 — invisible next to any O(0.01+) term of the same dot
// product); weights are pre-scaled per tensor on the host; the accumulator is multiplied by the exact
// inverse in the epilogue.
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(2))) _Float16 f16x2;
#define DIM_F16_ACT_SCALE 16.0f
__device__ __forceinline__ f32x16 mfma_f16(u32x4 a, u32x4 b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}
__device__ __forceinline__ unsigned cvt_pk_f16(float a, float b) {  // round-to-nearest-even
  f32x2 v = {a, b};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, f16x2));
}
// The residual x - (float)h is taken by v_fma_mix_f32, which reads an fp16 half directly (fma(h, -1, x): exact, the same value as the
// convert-and-subtract form the host-compiled test emulator uses): 4 instead of 6 VALU per pair of values.  Every VALU instruction
// of the matrix-core kernels costs ~4 cycles of a SIMD that issues no MFMA meanwhile (round-4 counters: MFMA-busy + VALU-active ~ 1).
__device__ __forceinline__ void split2_residual(float x0, float x1, unsigned h, float& r0, float& r1) {
#if defined(__AMDGCN__) && !defined(DIM_SPLIT_NO_MIX)   // (DIM_SPLIT_NO_MIX: the pre-round-4 form, for A/B builds)
  asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(r0) : "v"(h), "v"(x0));
  asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r1) : "v"(h), "v"(x1));
#else
  const f16x2 hv = __builtin_bit_cast(f16x2, h);
  r0 = x0 - (float)hv[0]; r1 = x1 - (float)hv[1];
#endif
}
__device__ __forceinline__ void split2_pk(float x0, float x1, float scale, unsigned& h, unsigned& l) {
  x0 = __builtin_amdgcn_fmed3f(x0 * scale, -65504.0f, 65504.0f);
  x1 = __builtin_amdgcn_fmed3f(x1 * scale, -65504.0f, 65504.0f);
  h = cvt_pk_f16(x0, x1);
  float r0, r1;
  split2_residual(x0, x1, h, r0, r1);
  l = cvt_pk_f16(r0, r1);
}
// the same split for a value that is already scaled and known to lie inside +-65504 (no multiply, no clamp)
__device__ __forceinline__ void split2_pk_raw(float x0, float x1, unsigned& h, unsigned& l) {
  h = cvt_pk_f16(x0, x1);
  float r0, r1;
  split2_residual(x0, x1, h, r0, r1);
  l = cvt_pk_f16(r0, r1);
}
// Split policy shared by conv_x6 / gemm_x6 / lg_attn_x6: MODE 1 = three bf16 planes, six cross terms;
// MODE 2 = two fp16 planes, three cross terms.  Cross terms are issued smallest first.
template <int MODE> struct SplitMma;
template <> struct SplitMma<1> {
  static constexpr int NPL = 3, NT = 6;
  __device__ static __forceinline__ float act_scale() { return 1.0f; }
  __device__ static __forceinline__ void split(float x0, float x1, float, unsigned (&p)[3]) { split3_pk(x0, x1, p[0], p[1], p[2]); }
  __device__ static __forceinline__ f32x16 mma(u32x4 a, u32x4 b, f32x16 c) { return mfma_bf16(a, b, c); }
  __device__ static __forceinline__ int ta(int t) { const int v[6] = {1, 0, 2, 0, 1, 0}; return v[t]; }
  __device__ static __forceinline__ int tb(int t) { const int v[6] = {1, 2, 0, 1, 0, 0}; return v[t]; }
};
template <> struct SplitMma<2> {
  static constexpr int NPL = 2, NT = 3;
  __device__ static __forceinline__ float act_scale() { return DIM_F16_ACT_SCALE; }
  __device__ static __forceinline__ void split(float x0, float x1, float scale, unsigned (&p)[2]) { split2_pk(x0, x1, scale, p[0], p[1]); }
  __device__ static __forceinline__ f32x16 mma(u32x4 a, u32x4 b, f32x16 c) { return mfma_f16(a, b, c); }
  __device__ static __forceinline__ int ta(int t) { const int v[3] = {1, 0, 0}; return v[t]; }
  __device__ static __forceinline__ int tb(int t) { const int v[3] = {0, 1, 0}; return v[t]; }
};

// exp(x) for x <= 0 to ~1.5 ulp in 6 instructions: 2^(x log2 e) with the product carried in two pieces (t + r), one v_exp_f32
// and a first-order correction 2^r = 1 + r ln 2.  (The library expf spends ~25 instructions on range handling the SELU's
// negative branch never needs; SELU is 40 of the ~1450 instructions per pixel of the feature-aggregation pass.)
__device__ __forceinline__ float exp_le0(float x) {
  const float l2e_hi = 1.44269502162933349609375f, l2e_lo = 1.925962989e-8f;
  const float t = x * l2e_hi;
  float r = fmaf(x, l2e_hi, -t);
  r = fmaf(x, l2e_lo, r);
  const float y = __builtin_amdgcn_exp2f(t);
  return fmaf(y, r * 0.693147180559945309417f, y);
}
// erf to ~1 ulp without branches (both ranges evaluated, one select): minimax polynomials for |x| <= 0.927734375
// (x + x p(x^2)) and beyond (1 - exp(q(|x|))), 13 fma + one v_exp_f32.  The coefficients are the widely circulated two-range
// single-precision minimax fit for erf (split point 0.927734375; published by N. Juffa in his public erff postings and restated in
// several open-source GPU math libraries); they are constants of that fit, not derived here.  Re-verified in this repository by a
// numpy restatement of exactly this evaluation order against scipy's fp64 erf on 6e6 points of [-6, 6] + N(0, 1.5): max error
// 1.25 ulp (at |x| = 0.933, just above the split), 7.5e-8 absolute.  The library erff costs ~45 instructions per value
// in divergent branches, which made the LayerNorm + GELU pass instruction-bound (3.3 TB/s) instead of HBM-bound.
__device__ __forceinline__ float erf_1ulp(float a) {
  const float t = fabsf(a), s = a * a;
  float r = fmaf(-1.72853470e-5f, t, 3.83197126e-4f);
  const float u = fmaf(-3.88396438e-3f, t, 2.42546219e-2f);
  r = fmaf(r, s, u);
  r = fmaf(r, t, -1.06777877e-1f);
  r = fmaf(r, t, -6.34846687e-1f);
  r = fmaf(r, t, -1.28717512e-1f);
  r = fmaf(r, t, -t);
  const float big = copysignf(1.0f - __builtin_amdgcn_exp2f(r * 1.44269504088896340736f), a);
  float q = -5.96761703e-4f;
  q = fmaf(q, s, 4.99119423e-3f);
  q = fmaf(q, s, -2.67681349e-2f);
  q = fmaf(q, s, 1.12819925e-1f);
  q = fmaf(q, s, -3.76125336e-1f);
  q = fmaf(q, s, 1.28379166e-1f);
  return t > 0.927734375f ? big : fmaf(q, a, a);
}

// The same function on a pair of values with the polynomial chains as packed fp32 FMAs (v_pk_fma_f32: two lanes' worth of work per
// issue slot); bit-identical to erf_1ulp per element.  The fused ffn.0 + LayerNorm + GELU epilogue is VALU-bound without it.
__device__ __forceinline__ f32x2 erf2_1ulp(f32x2 a) {
  const f32x2 t = __builtin_elementwise_abs(a), s = a * a;
  auto k = [](float c) { return f32x2{c, c}; };
  f32x2 r = __builtin_elementwise_fma(k(-1.72853470e-5f), t, k(3.83197126e-4f));
  const f32x2 u = __builtin_elementwise_fma(k(-3.88396438e-3f), t, k(2.42546219e-2f));
  r = __builtin_elementwise_fma(r, s, u);
  r = __builtin_elementwise_fma(r, t, k(-1.06777877e-1f));
  r = __builtin_elementwise_fma(r, t, k(-6.34846687e-1f));
  r = __builtin_elementwise_fma(r, t, k(-1.28717512e-1f));
  r = __builtin_elementwise_fma(r, t, -t);
  r = r * k(1.44269504088896340736f);
  f32x2 q = __builtin_elementwise_fma(k(-5.96761703e-4f), s, k(4.99119423e-3f));
  q = __builtin_elementwise_fma(q, s, k(-2.67681349e-2f));
  q = __builtin_elementwise_fma(q, s, k(1.12819925e-1f));
  q = __builtin_elementwise_fma(q, s, k(-3.76125336e-1f));
  q = __builtin_elementwise_fma(q, s, k(1.28379166e-1f));
  const f32x2 small = __builtin_elementwise_fma(q, a, a);
  f32x2 o;
  o[0] = t[0] > 0.927734375f ? copysignf(1.0f - __builtin_amdgcn_exp2f(r[0]), a[0]) : small[0];
  o[1] = t[1] > 0.927734375f ? copysignf(1.0f - __builtin_amdgcn_exp2f(r[1]), a[1]) : small[1];
  return o;
}

// row index inside a 32x32 MFMA tile held by (lane-half h, register r)
__device__ __forceinline__ int mfma_row(int r, int half) { return (r & 3) + 8 * (r >> 2) + 4 * half; }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}
// The same total without the LDS crossbar: __shfl_xor lowers to ds_bpermute_b32 (an LDS-pipe round trip per step, six dependent
// ones per reduction), DPP moves run at VALU rate.  quad_perm (lane ^ 1, lane ^ 2), row_half_mirror and row_mirror give every
// lane the sum of its row of 16; the four row totals are wave-uniform values read with v_readlane.  (Different association
// than wave_sum: use one of the two consistently per quantity.)
__device__ __forceinline__ float dpp_f(float v, const int ctrl) {
  switch (ctrl) {
    case 0xB1: return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xF, 0xF, false));
    case 0x4E: return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xF, 0xF, false));
    case 0x141: return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x141, 0xF, 0xF, false));
    default: return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x140, 0xF, 0xF, false));
  }
}
__device__ __forceinline__ float wave_sum_dpp(float v) {
  v += dpp_f(v, 0xB1);
  v += dpp_f(v, 0x4E);
  v += dpp_f(v, 0x141);
  v += dpp_f(v, 0x140);
  const float r0 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 0)), r1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 16));
  const float r2 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 32)), r3 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 48));
  return (r0 + r1) + (r2 + r3);
}
// max / sum over a lane and its partner in the other lane half (lane ^ 32; the two halves of a 32x32 MFMA tile's column).  __shfl_xor(v, 32) is a
// ds_bpermute_b32: an LDS-pipe round trip in the middle of a dependent chain (the attention's row maximum sits between the score MFMAs and the
// exponentials of EVERY key tile).  gfx950's v_permlane32_swap exchanges the upper half of one register with the lower half of another in one VALU
// instruction: with (v, v) the two results hold {lower value in all lanes} and {upper value in all lanes}.  Same operands, commutative operation:
// the same bits as the shuffle form (which the test emulator runs).
__device__ __forceinline__ float half_pair_max(float v) {
#if defined(__AMDGCN__) && !defined(DIM_NO_PERMLANE)   // (DIM_NO_PERMLANE: the shuffle form, for A/B builds)
  const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
#else
  return fmaxf(v, __shfl_xor(v, 32));
#endif
}
__device__ __forceinline__ float half_pair_sum(float v) {
#if defined(__AMDGCN__) && !defined(DIM_NO_PERMLANE)   // (DIM_NO_PERMLANE: the shuffle form, for A/B builds)
  const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
#else
  return v + __shfl_xor(v, 32);
#endif
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

// ---- buffer (SRD) addressing for epilogues --------------------------------------------------------------------------
// A masked `if (ok) p[i] = v` store costs a branch (s_and_saveexec / s_cbranch_execz) plus 64-bit address arithmetic PER
// ELEMENT: the epilogues of the matrix-core kernels were 30-45 % of their instruction stream that way.  With a buffer
// resource descriptor (wave-uniform base + byte size) the address is a 32-bit byte offset, and anything at or beyond the
// size is dropped (stores) or reads as zero (loads) by the hardware: rows past a ragged end need no test at all, and any
// other mask is one v_cndmask selecting the out-of-range offset DIM_BUF_OOB (cdna_hip_programming.md T8 / T20).
// The base must be wave-uniform (blockIdx-derived); sizes above 4 GiB are avoided by taking the base per image / item.
typedef __amdgpu_buffer_rsrc_t dim_rsrc;
#define DIM_BUF_OOB 0x80000000u   // beyond every descriptor used here (< 2 GiB) and still out of range after adding a tile's offsets
__device__ __forceinline__ dim_rsrc buf_rsrc(const void* base, size_t bytes) {
  // readfirstlane makes the (already uniform) base provably uniform to the compiler: no waterfall loop around each access
  const unsigned long long p = (unsigned long long)base;
  const unsigned lo = __builtin_amdgcn_readfirstlane((int)(unsigned)p), hi = __builtin_amdgcn_readfirstlane((int)(unsigned)(p >> 32));
  void* q = (void*)(((unsigned long long)hi << 32) | lo);
  const unsigned n = bytes > 0x7FFFFFF0ull ? 0x7FFFFFF0u : (unsigned)bytes;
  return __builtin_amdgcn_make_buffer_rsrc(q, (short)0, (int)__builtin_amdgcn_readfirstlane((int)n), 0x00020000);
}
__device__ __forceinline__ void buf_store_f32(dim_rsrc r, unsigned byte_off, float v) { __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), r, (int)byte_off, 0, 0); }
__device__ __forceinline__ void buf_store_u32(dim_rsrc r, unsigned byte_off, unsigned v) { __builtin_amdgcn_raw_buffer_store_b32(v, r, (int)byte_off, 0, 0); }
__device__ __forceinline__ void buf_store_u16(dim_rsrc r, unsigned byte_off, unsigned short v) { __builtin_amdgcn_raw_buffer_store_b16(v, r, (int)byte_off, 0, 0); }
// neighbouring-lane exchange (lane ^ 1) as a DPP quad permute: no LDS traffic, one VALU
__device__ __forceinline__ unsigned lane_swap1(unsigned v) { return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xF, 0xF, false); }
// byte select from {hi, lo} (v_perm_b32): selector byte k in 0..3 picks lo byte k, 4..7 picks hi byte k - 4
__device__ __forceinline__ unsigned byte_perm(unsigned hi, unsigned lo, unsigned sel) { return __builtin_amdgcn_perm(hi, lo, sel); }
// ... with a wave-uniform byte offset in an SGPR on top of the per-lane one.  The hardware range check covers the per-lane offset
// only: keep everything that can run past the end (the row) in byte_off.
__device__ __forceinline__ float buf_load_f32_s(dim_rsrc r, unsigned byte_off, unsigned uniform_off) { return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, (int)byte_off, (int)uniform_off, 0)); }
__device__ __forceinline__ void buf_store_f32_s(dim_rsrc r, unsigned byte_off, unsigned uniform_off, float v) { __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), r, (int)byte_off, (int)uniform_off, 0); }
__device__ __forceinline__ float buf_load_f32(dim_rsrc r, unsigned byte_off) { return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, (int)byte_off, 0, 0)); }

// LDS-DMA (global_load_lds_dwordx4): 16 bytes per lane straight from global memory into LDS at the WAVE-UNIFORM base + lane * 16
// — no staging registers, no ds_write pass.  The transfer is an outstanding vector-memory operation: it is complete after
// s_waitcnt vmcnt(0) (which __syncthreads() emits when one is in flight), and may be read after the barrier that follows.
__device__ __forceinline__ void lds_dma16(const void* gsrc_lane, void* lds_wave_base) {
  __builtin_amdgcn_global_load_lds(gsrc_lane, (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}
// The compiler does NOT order a later LDS read (or the barrier in front of it) after an LDS-DMA by itself: retire this
// wave's outstanding transfers explicitly before the barrier.  s_waitcnt vmcnt(0) (expcnt / lgkmcnt fields left at "any").
__device__ __forceinline__ void lds_dma_wait_all() { __builtin_amdgcn_s_waitcnt(0x0F70); }

// ---- fp16x3 range guard ---------------------------------------------------------------------------
// Every kernel that PRODUCES a value a later fp16x3 split will consume (conv / GEMM epilogues, LayerNorm+GELU,
// the rotated q / k, the external inputs) tracks max|x| of what it writes and bumps a sticky per-site device
// counter when that exceeds the exact range of the split (65504 / DIM_F16_ACT_SCALE = 4094): one v_max3 per two
// outputs in an epilogue and one (never taken) branch per thread.  The host reads the counters through
// dim_saturation_read() and re-runs the call in bf16x6 (no range limit) when any is non-zero.
// Non-finite values: an Inf trips the site that produces it (fmaxf keeps it).  v_max drops a NaN operand, so the epilogue
// trackers do NOT see NaN — they do not need to: with finite weights (checked by every dim_*_create) and finite external
// inputs, every intermediate of these networks is a finite-bounded sum unless an earlier site has already reported an
// Inf, so a NaN can only ENTER through the external inputs, and those two sites (image patch, input descriptors /
// keypoints) track with NaN-sticky forms.
#define DIM_F16_ACT_LIMIT (65504.0f / DIM_F16_ACT_SCALE)
__device__ __forceinline__ float sat_track(float m, float a, float b) { return fmaxf(m, fmaxf(fabsf(a), fabsf(b))); }
__device__ __forceinline__ void sat_report(unsigned* ctr, float m) {
  if (ctr != nullptr && !(m <= DIM_F16_ACT_LIMIT)) atomicAdd(ctr, 1u);  // a NaN maximum counts as out of range
}

// ---- host side -------------------------------------------------------------
void dim_set_error(const char* fmt, ...);
// device pointer to the saturation counter of `site` (DIM_SAT_* in include/dim_hip.h) on the current device; the
// block of counters is allocated and zeroed on first use.  nullptr if the allocation failed (guard disabled).
unsigned* dim_sat_counter(int site);
void dim_sat_host_bump(int site);  // a range violation established on the host (e.g. a weight bound at create time)
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

// ---- per-handle overrides of the dim_tune_set choices (include/dim_hip.h: dim_handle_tune_set) ----
// Every extractor / matcher handle starts with a DimHandleBase; the C-ABI entry points open a DimTuneScope on it, and the accessors
// (dim_precision_mode(), dim_fuse_conv1a(), ...) return the handle's override while the scope is open on this thread, else the process default.
constexpr int DIM_TUNE_KEYS = 19;
constexpr unsigned DIM_HANDLE_MAGIC = 0x44494d48u;   // "DIMH"
struct DimTune {
  int v[DIM_TUNE_KEYS];
  DimTune() { for (int i = 0; i < DIM_TUNE_KEYS; ++i) v[i] = -1; }   // -1 = inherit the process default
};
struct DimHandleBase {
  unsigned magic = DIM_HANDLE_MAGIC;
  DimTune tune;
};
void dim_tune_scope_set(const DimTune* t);     // api_ops.hip (thread-local)
const DimTune* dim_tune_scope_get();
struct DimTuneScope {
  const DimTune* prev;
  explicit DimTuneScope(const DimHandleBase* h) : prev(dim_tune_scope_get()) { dim_tune_scope_set(h ? &h->tune : nullptr); }
  ~DimTuneScope() { dim_tune_scope_set(prev); }
};

static inline int cdiv(int a, int b) { return (a + b - 1) / b; }
// weights must be finite: the range guard's reasoning (above) rests on it, and a NaN checkpoint is a caller error
static inline bool dim_all_finite(const float* p, size_t n) {
  for (size_t i = 0; i < n; ++i) if (!(p[i] - p[i] == 0.0f)) return false;
  return true;
}
