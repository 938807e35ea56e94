// hipemu — TEST-ONLY stand-in for <hip/hip_runtime.h>.
//
// This header is NOT part of the product.  It exists so that the very same
// .hip kernel sources that hipcc compiles for gfx950 can also be compiled
// with the host clang and stepped through on a CPU-only machine, one fiber
// per GPU thread, to check indexing / fragment-layout / barrier logic before
// spending GPU minutes.  The product library (libdim_hip.so) is built by
// hipcc against the real ROCm header and never sees this file; the emulated
// library is only ever loaded by tests under tests/ (never by the package,
// bench.py or smoke()).
//
// Semantics that are emulated faithfully:
//   * wave = 64 lanes, workgroup barriers, wave-level cross-lane ops
//     (shfl / ballot) with lock-step exchange between the lanes of a wave,
//   * v_mfma_f32_32x32x2_f32 and v_mfma_f32_16x16x4_f32 with the gfx950
//     A/B/C/D lane->element maps (cdna_hip_programming.md §3), k-ordered fmaf
//     chain, so a wrong fragment layout fails here exactly as on hardware,
//   * __shared__ (one copy per workgroup; workgroups run one after another).
// Not emulated: timing, caches, memory model (fibers of a block run in a
// fixed or reversed order between sync points: HIPEMU_ORDER=reverse flips the
// order so a missing barrier shows up as a result change).
#pragma once
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <algorithm>

struct dim3 {
  unsigned x, y, z;
  constexpr dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
struct uint3 { unsigned x, y, z; };
struct float2 { float x, y; };
struct float4 { float x, y, z, w; };
struct int2 { int x, y; };
struct int4 { int x, y, z, w; };
static inline float4 make_float4(float x, float y, float z, float w) { return float4{x, y, z, w}; }
static inline float2 make_float2(float x, float y) { return float2{x, y}; }
static inline int2 make_int2(int x, int y) { return int2{x, y}; }

using std::max;
using std::min;
#define __global__
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __launch_bounds__(...)
#define __shared__ static
#define HIP_KERNEL_NAME(...) __VA_ARGS__

typedef int hipError_t;
#define hipSuccess 0
#define hipErrorInvalidValue 1
#define hipErrorOutOfMemory 2
typedef struct hipemu_stream* hipStream_t;
typedef struct hipemu_event* hipEvent_t;
enum hipMemcpyKind { hipMemcpyHostToHost, hipMemcpyHostToDevice, hipMemcpyDeviceToHost, hipMemcpyDeviceToDevice, hipMemcpyDefault };

namespace hipemu {
struct Fiber;
struct Fiber {
  void* sp;
  uint3 tid;
  unsigned flat;     // flat thread id in block
  unsigned lane;     // flat & 63
  unsigned wave;     // flat >> 6
  int state;         // 0 runnable, 1 at barrier, 2 done
  unsigned opcount;  // wave-op sequence number
};
extern Fiber* cur;
extern uint3 g_block, g_bdim, g_gdim;
void barrier();
// Lock-step exchange between the live lanes of the calling lane's wave.
// Deposits `bytes` from `in`; returns a pointer to a [64][stride] byte table of
// every lane's deposit (valid until this lane's next-but-one wave op) and the
// mask of lanes that took part.
const unsigned char* wave_exchange(const void* in, unsigned bytes, unsigned* stride, unsigned long long* present);
void launch(dim3 grid, dim3 block, const std::function<void()>& body);
void wave_collective(const void* in, unsigned bytes, void (*compute)(const unsigned char* tab, unsigned stride, unsigned char* out),
                     unsigned out_bytes, void* my_out);
// LDS-DMA is ASYNCHRONOUS on the hardware and neither the compiler nor __syncthreads() orders later LDS reads after it:
// the emulated transfer snapshots its source bytes at issue and lands only when the issuing thread executes
// s_waitcnt vmcnt(0) (or exits) — a kernel that forgets the wait reads stale LDS here too and fails its parity test.
void dma_issue(void* lds_dst, const void* src, unsigned size);
void dma_retire(unsigned keep_newest = 0);
void set_dma_mode(int early);  // 1: land at issue instead (exposes restaging a buffer other threads still read)
}  // namespace hipemu

#define threadIdx (hipemu::cur->tid)
#define blockIdx (hipemu::g_block)
#define blockDim (hipemu::g_bdim)
#define gridDim (hipemu::g_gdim)
#define warpSize 64

static inline void __syncthreads() { hipemu::barrier(); }
static inline void __builtin_amdgcn_s_barrier() { hipemu::barrier(); }
static inline void __threadfence() {}
static inline void __threadfence_block() {}
static inline void __threadfence_system() {}

template <typename T>
static inline T hipemu_shfl_from(T v, int src_lane_for_me) {
  unsigned stride; unsigned long long present;
  const unsigned char* tab = hipemu::wave_exchange(&v, sizeof(T), &stride, &present);
  T out;
  int s = src_lane_for_me & 63;
  if (!((present >> s) & 1ull)) return v;  // inactive source: undefined on HW; keep own value
  memcpy(&out, tab + (size_t)s * stride, sizeof(T));
  return out;
}
template <typename T> static inline T __shfl(T v, int lane, int width = 64) {
  int me = hipemu::cur->lane;
  int base = me & ~(width - 1);
  return hipemu_shfl_from(v, base + (lane & (width - 1)));
}
template <typename T> static inline T __shfl_xor(T v, int mask, int width = 64) {
  int me = hipemu::cur->lane;
  int base = me & ~(width - 1);
  return hipemu_shfl_from(v, base + (((me & (width - 1)) ^ mask) & (width - 1)));
}
template <typename T> static inline T __shfl_down(T v, unsigned d, int width = 64) {
  int me = hipemu::cur->lane;
  int in = me & (width - 1);
  int src = (in + (int)d < width) ? me + (int)d : me;
  return hipemu_shfl_from(v, src);
}
template <typename T> static inline T __shfl_up(T v, unsigned d, int width = 64) {
  int me = hipemu::cur->lane;
  int in = me & (width - 1);
  int src = (in - (int)d >= 0) ? me - (int)d : me;
  return hipemu_shfl_from(v, src);
}
static inline unsigned long long __ballot(int pred) {
  unsigned stride; unsigned long long present;
  int p = pred ? 1 : 0;
  const unsigned char* tab = hipemu::wave_exchange(&p, sizeof(int), &stride, &present);
  unsigned long long m = 0;
  for (int l = 0; l < 64; ++l)
    if ((present >> l) & 1ull) { int q; memcpy(&q, tab + (size_t)l * stride, 4); if (q) m |= 1ull << l; }
  return m;
}
static inline int __any(int p) { return __ballot(p) != 0ull; }
static inline int __popcll(unsigned long long v) { return __builtin_popcountll(v); }
static inline int __popc(unsigned v) { return __builtin_popcount(v); }
static inline int __clz(int v) { return v == 0 ? 32 : __builtin_clz((unsigned)v); }
static inline int __ffsll(unsigned long long v) { return __builtin_ffsll((long long)v); }

typedef float hipemu_f32x16 __attribute__((ext_vector_type(16)));
typedef float hipemu_f32x4 __attribute__((ext_vector_type(4)));

// v_mfma_f32_32x32x2_f32: A[i=l&31][k=l>>5], B[k=l>>5][j=l&31],
// D: col=l&31, row=(r&3)+8*(r>>2)+4*(l>>5); k-ordered fmaf chain.
static inline hipemu_f32x16 hipemu_mfma_32x32x2(float a, float b, hipemu_f32x16 c) {
  float ab[2] = {a, b};
  unsigned stride; unsigned long long present;
  const unsigned char* tab = hipemu::wave_exchange(ab, sizeof(ab), &stride, &present);
  if (present != ~0ull) { fprintf(stderr, "hipemu: MFMA issued with a partial wave (exec mask %016llx)\n", present); abort(); }
  int l = hipemu::cur->lane;
  int col = l & 31;
  hipemu_f32x16 d = c;
  for (int r = 0; r < 16; ++r) {
    int row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5);
    float acc = c[r];
    for (int k = 0; k < 2; ++k) {
      float av, bv;
      memcpy(&av, tab + (size_t)(row + 32 * k) * stride, 4);
      memcpy(&bv, tab + (size_t)(col + 32 * k) * stride + 4, 4);
      acc = fmaf(av, bv, acc);
    }
    d[r] = acc;
  }
  return d;
}
// v_mfma_f32_16x16x4_f32: A[i=l&15][k=l>>4], B[k=l>>4][j=l&15],
// D: col=l&15, row=(l>>4)*4+r.
static inline hipemu_f32x4 hipemu_mfma_16x16x4(float a, float b, hipemu_f32x4 c) {
  float ab[2] = {a, b};
  unsigned stride; unsigned long long present;
  const unsigned char* tab = hipemu::wave_exchange(ab, sizeof(ab), &stride, &present);
  if (present != ~0ull) { fprintf(stderr, "hipemu: MFMA issued with a partial wave\n"); abort(); }
  int l = hipemu::cur->lane;
  int col = l & 15;
  hipemu_f32x4 d = c;
  for (int r = 0; r < 4; ++r) {
    int row = (l >> 4) * 4 + r;
    float acc = c[r];
    for (int k = 0; k < 4; ++k) {
      float av, bv;
      memcpy(&av, tab + (size_t)(row + 16 * k) * stride, 4);
      memcpy(&bv, tab + (size_t)(col + 16 * k) * stride + 4, 4);
      acc = fmaf(av, bv, acc);
    }
    d[r] = acc;
  }
  return d;
}
// v_mfma_f32_32x32x16_bf16: A[i=l&31][k=8*(l>>5)+e], B[k=8*(l>>5)+e][j=l&31], fp32 accumulate.
typedef __bf16 hipemu_bf16x8 __attribute__((ext_vector_type(8)));
static inline void hipemu_mfma_bf16_compute(const unsigned char* tab, unsigned stride, unsigned char* out) {
  // tab[lane]: 8 bf16 of A, 8 bf16 of B.  out[lane]: 16 floats = sum_k A[row][k] * B[k][col] (fp32 fmaf chain from 0)
  float A[32][16], B[16][32];
  for (int l = 0; l < 64; ++l)
    for (int e = 0; e < 8; ++e) {
      unsigned short ua, ub;
      memcpy(&ua, tab + (size_t)l * stride + 2 * e, 2);
      memcpy(&ub, tab + (size_t)l * stride + 16 + 2 * e, 2);
      unsigned wa = (unsigned)ua << 16, wb = (unsigned)ub << 16;
      memcpy(&A[l & 31][8 * (l >> 5) + e], &wa, 4);
      memcpy(&B[8 * (l >> 5) + e][l & 31], &wb, 4);
    }
  for (int l = 0; l < 64; ++l) {
    float* o = (float*)(out + (size_t)l * 64);
    const int col = l & 31;
    for (int r = 0; r < 16; ++r) {
      const int row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5);
      float acc = 0.f;
      for (int k = 0; k < 16; ++k) acc = fmaf(A[row][k], B[k][col], acc);
      o[r] = acc;
    }
  }
}
static inline hipemu_f32x16 hipemu_mfma_32x32x16_bf16(hipemu_bf16x8 a, hipemu_bf16x8 b, hipemu_f32x16 c) {
  unsigned short ab[16];
  memcpy(ab, &a, 16); memcpy(ab + 8, &b, 16);
  float prod[16];
  hipemu::wave_collective(ab, sizeof(ab), hipemu_mfma_bf16_compute, 64, prod);
  hipemu_f32x16 d = c;
  for (int r = 0; r < 16; ++r) d[r] = c[r] + prod[r];
  return d;
}
#define __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, x, y, z) hipemu_mfma_32x32x16_bf16((a), (b), (c))
// v_mfma_f32_32x32x16_f16: same lane map (scripts/probe/mfma_f16_probe.hip), fp16 inputs incl. subnormals, fp32 accumulate.
typedef _Float16 hipemu_f16x8 __attribute__((ext_vector_type(8)));
static inline void hipemu_mfma_f16_compute(const unsigned char* tab, unsigned stride, unsigned char* out) {
  float A[32][16], B[16][32];
  for (int l = 0; l < 64; ++l)
    for (int e = 0; e < 8; ++e) {
      _Float16 ua, ub;
      memcpy(&ua, tab + (size_t)l * stride + 2 * e, 2);
      memcpy(&ub, tab + (size_t)l * stride + 16 + 2 * e, 2);
      A[l & 31][8 * (l >> 5) + e] = (float)ua;
      B[8 * (l >> 5) + e][l & 31] = (float)ub;
    }
  for (int l = 0; l < 64; ++l) {
    float* o = (float*)(out + (size_t)l * 64);
    const int col = l & 31;
    for (int r = 0; r < 16; ++r) {
      const int row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5);
      float acc = 0.f;
      for (int k = 0; k < 16; ++k) acc = fmaf(A[row][k], B[k][col], acc);
      o[r] = acc;
    }
  }
}
static inline hipemu_f32x16 hipemu_mfma_32x32x16_f16(hipemu_f16x8 a, hipemu_f16x8 b, hipemu_f32x16 c) {
  unsigned short ab[16];
  memcpy(ab, &a, 16); memcpy(ab + 8, &b, 16);
  float prod[16];
  hipemu::wave_collective(ab, sizeof(ab), hipemu_mfma_f16_compute, 64, prod);
  hipemu_f32x16 d = c;
  for (int r = 0; r < 16; ++r) d[r] = c[r] + prod[r];
  return d;
}
#define __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, x, y, z) hipemu_mfma_32x32x16_f16((a), (b), (c))
#define __builtin_amdgcn_sched_barrier(mask) ((void)0)
#define __builtin_amdgcn_s_setprio(p) ((void)0)
#define __builtin_amdgcn_s_sleep(n) ((void)0)
static inline float __builtin_amdgcn_fmed3f(float a, float b, float c) { return fmaxf(fminf(fmaxf(a, b), c), fminf(a, b)); }
#define __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, x, y, z) hipemu_mfma_32x32x2((a), (b), (c))
#define __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, x, y, z) hipemu_mfma_16x16x4((a), (b), (c))

// buffer resource descriptors: (base, bytes); accesses at or beyond `bytes` are dropped / read as zero, like the hardware's
struct hipemu_rsrc { unsigned char* base; unsigned bytes; };
typedef hipemu_rsrc __amdgpu_buffer_rsrc_t;
static inline hipemu_rsrc __builtin_amdgcn_make_buffer_rsrc(void* p, short, int n, int) { return hipemu_rsrc{(unsigned char*)p, (unsigned)n}; }
static inline void __builtin_amdgcn_raw_buffer_store_b32(unsigned v, hipemu_rsrc r, int off, int soff, int) {
  const unsigned o = (unsigned)off + (unsigned)soff; if ((unsigned long long)o + 4 <= r.bytes) memcpy(r.base + o, &v, 4);
}
static inline void __builtin_amdgcn_raw_buffer_store_b16(unsigned short v, hipemu_rsrc r, int off, int soff, int) {
  const unsigned o = (unsigned)off + (unsigned)soff; if ((unsigned long long)o + 2 <= r.bytes) memcpy(r.base + o, &v, 2);
}
static inline unsigned __builtin_amdgcn_raw_buffer_load_b32(hipemu_rsrc r, int off, int soff, int) {
  const unsigned o = (unsigned)off + (unsigned)soff; unsigned v = 0; if ((unsigned long long)o + 4 <= r.bytes) memcpy(&v, r.base + o, 4); return v;
}

// LDS-DMA (global_load_lds_dwordx4 ...): every lane's `size` bytes go to LDS at the wave-uniform base + lane * size + offset
static inline void __builtin_amdgcn_global_load_lds(const void* gptr, __attribute__((address_space(3))) void* lds, unsigned size, int offset, int) {
  hipemu::dma_issue((char*)(unsigned long long)lds + (size_t)hipemu::cur->lane * size + offset, gptr, size);
}
// s_waitcnt simm16 (gfx9 encoding): vmcnt = bits 3:0 | bits 15:14 << 4.  vmcnt(N) retires this thread's LDS-DMA transfers except the N newest
// (vector-memory operations complete in order); 63 = "any" retires nothing.
static inline void __builtin_amdgcn_s_waitcnt(int imm) {
  const unsigned n = ((unsigned)imm & 0xFu) | ((((unsigned)imm >> 14) & 3u) << 4);
  if (n < 63u) hipemu::dma_retire(n);
}

// DPP quad permutes used by the kernels: 0xB1 = quad_perm [1,0,3,2] (lane ^ 1), 0x4E = quad_perm [2,3,0,1] (lane ^ 2)
static inline int __builtin_amdgcn_update_dpp(int old, int src, int ctrl, int, int, bool) {
  (void)old;
  const int me = (int)hipemu::cur->lane;
  if (ctrl == 0x141) return hipemu_shfl_from(src, (me & ~7) | (7 - (me & 7)));      // row_half_mirror: reverse within 8 lanes
  if (ctrl == 0x140) return hipemu_shfl_from(src, (me & ~15) | (15 - (me & 15)));   // row_mirror: reverse within a row of 16
  const int x = ctrl == 0xB1 ? 1 : (ctrl == 0x4E ? 2 : -1);
  if (x < 0) { fprintf(stderr, "hipemu: unsupported dpp_ctrl 0x%x\n", ctrl); abort(); }
  return hipemu_shfl_from(src, me ^ x);
}
static inline unsigned __builtin_amdgcn_perm(unsigned hi, unsigned lo, unsigned sel) {
  const unsigned long long both = ((unsigned long long)hi << 32) | lo;
  unsigned out = 0;
  for (int k = 0; k < 4; ++k) {
    const unsigned s = (sel >> (8 * k)) & 0xff;
    const unsigned b = s < 8 ? (unsigned)((both >> (8 * s)) & 0xff) : (s == 0x0c ? 0u : 0xffu);
    out |= b << (8 * k);
  }
  return out;
}

// atomics: one fiber runs at a time, so plain read-modify-write is atomic.
template <typename T> static inline T atomicAdd(T* p, T v) { T o = *p; *p = o + v; return o; }
template <typename T> static inline T atomicMax(T* p, T v) { T o = *p; *p = o > v ? o : v; return o; }
template <typename T> static inline T atomicMin(T* p, T v) { T o = *p; *p = o < v ? o : v; return o; }
template <typename T> static inline T atomicOr(T* p, T v) { T o = *p; *p = o | v; return o; }
template <typename T> static inline T atomicExch(T* p, T v) { T o = *p; *p = v; return o; }
template <typename T> static inline T atomicCAS(T* p, T cmp, T v) { T o = *p; if (o == cmp) *p = v; return o; }

static inline unsigned long long wall_clock64() { return (unsigned long long)__builtin_readcyclecounter() / 30; }
#define __expf(x) expf(x)
static inline float __builtin_amdgcn_exp2f(float x) { return exp2f(x); }
// only ever applied to wave-uniform values in this code base
static inline int __builtin_amdgcn_readfirstlane(int x) { return x; }
// v_readlane_b32: the value lane `lane` holds, broadcast (wave-uniform on hardware: an SGPR)
static inline int __builtin_amdgcn_readlane(int v, int lane) { return hipemu_shfl_from(v, lane); }
#define __logf(x) logf(x)
static inline float rsqrtf(float x) { return 1.0f / sqrtf(x); }
static inline float __fdividef(float a, float b) { return a / b; }
static inline float __int_as_float(int v) { float f; memcpy(&f, &v, 4); return f; }
static inline int __float_as_int(float f) { int v; memcpy(&v, &f, 4); return v; }
static inline unsigned __float_as_uint(float f) { unsigned v; memcpy(&v, &f, 4); return v; }
static inline float __uint_as_float(unsigned v) { float f; memcpy(&f, &v, 4); return f; }

// ---- host runtime subset -------------------------------------------------
extern "C" {
hipError_t hipMalloc(void** p, size_t n);
hipError_t hipFree(void* p);
#define hipHostMallocMapped 0x2
#define hipHostMallocCoherent 0x40000000
hipError_t hipHostMalloc(void** p, size_t n, unsigned flags);
hipError_t hipHostFree(void* p);
hipError_t hipMemcpy(void* d, const void* s, size_t n, hipMemcpyKind k);
#define HIP_SYMBOL(x) (&(x))
static inline hipError_t hipMemcpyFromSymbol(void* d, const void* sym, size_t n) { memcpy(d, sym, n); return hipSuccess; }
static inline hipError_t hipMemcpyToSymbol(void* sym, const void* s, size_t n) { memcpy(sym, s, n); return hipSuccess; }
hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind k, hipStream_t st);
hipError_t hipMemset(void* d, int v, size_t n);
hipError_t hipMemsetAsync(void* d, int v, size_t n, hipStream_t st);
hipError_t hipStreamSynchronize(hipStream_t st);
hipError_t hipDeviceSynchronize();
hipError_t hipGetLastError();
hipError_t hipPeekAtLastError();
const char* hipGetErrorString(hipError_t e);
hipError_t hipEventCreate(hipEvent_t* e);
hipError_t hipEventDestroy(hipEvent_t e);
hipError_t hipEventRecord(hipEvent_t e, hipStream_t st);
hipError_t hipEventSynchronize(hipEvent_t e);
hipError_t hipEventElapsedTime(float* ms, hipEvent_t a, hipEvent_t b);
hipError_t hipGetDevice(int* d);
hipError_t hipSetDevice(int d);
hipError_t hipGetDeviceCount(int* n);
}
template <typename T> static inline hipError_t hipMalloc(T** p, size_t n) { return hipMalloc((void**)p, n); }

#define hipLaunchKernelGGL(kernel, grid, block, shmem, stream, ...) \
  hipemu::launch((grid), (block), [=]() { kernel(__VA_ARGS__); })
