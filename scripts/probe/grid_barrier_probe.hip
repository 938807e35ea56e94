// Probe (round 6): what does a grid-wide phase boundary cost INSIDE a persistent kernel on MI355X, against the kernel boundary it would replace?
// The batch-1 LightGlue call (one pair per call through the plugin hooks) is ~135 dependent launches of 5 - 35 us; VERDICT r5 next #3 asks for a
// persistent kernel per layer with grid-wide phase barriers.  This measures, per barrier / per launch:
//   barrier variant 0: one monotonic counter (agent-scope release add by thread 0 of every workgroup, acquire spin on the same word)
//   barrier variant 1: per-XCD counters (workgroup b -> XCD b % 8), the last arrival of an XCD adds to the global counter, everybody spins on the global word
//   with G = 256 / 512 / 1024 workgroups of 256 threads (1 / 2 / 4 per CU), each workgroup writing 4 KB of "phase output" per phase that ANOTHER
//   workgroup (on another XCD) reads and checks after the barrier (so the cost includes making the data visible across the 8 L2s);
//   kernel boundary: N dependent launches of an empty 1-workgroup kernel / of a 256-workgroup kernel that writes + reads the same 4 KB per workgroup.
//   hipcc --offload-arch=gfx950 -O3 scripts/probe/grid_barrier_probe.hip -o scripts/probe/grid_barrier_probe && scripts/probe/grid_barrier_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("{\"error\": \"%s at line %d\"}\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

__device__ __forceinline__ unsigned ld_acq(const unsigned* p) { return __hip_atomic_load(p, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ unsigned ld_rlx(const unsigned* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

template <int VAR>
__device__ __forceinline__ void grid_barrier(unsigned* ctr, unsigned phase, unsigned G) {
  __syncthreads();   // every thread's stores of the phase are issued
  if (threadIdx.x == 0) {
    if (VAR == 0) {
      __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
      while (ld_rlx(ctr) < (phase + 1) * G) __builtin_amdgcn_s_sleep(1);
      __atomic_thread_fence(__ATOMIC_ACQUIRE);   // (HIP lowers this to agent scope + buffer_inv sc1 on gfx9)
    } else {
      const unsigned xcd = blockIdx.x & 7, per = G >> 3;
      const unsigned old = __hip_atomic_fetch_add(ctr + 32 * (1 + xcd), 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
      if (old == (phase + 1) * per - 1) __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
      while (ld_rlx(ctr) < (phase + 1) * 8) __builtin_amdgcn_s_sleep(1);
      __atomic_thread_fence(__ATOMIC_ACQUIRE);
    }
  }
  __syncthreads();
}

// every phase: workgroup b writes 4 KB (16 B per thread) tagged with the phase; after the barrier it checks the 4 KB of workgroup (b + 3) % G (another XCD)
template <int VAR>
__global__ __launch_bounds__(256) void persistent_kernel(unsigned* ctr, uint4* data, unsigned* bad, int phases) {
  const unsigned G = gridDim.x, b = blockIdx.x, t = threadIdx.x;
  unsigned nbad = 0;
  for (int ph = 0; ph < phases; ++ph) {
    data[(size_t)b * 256 + t] = uint4{(unsigned)ph, b, t, (unsigned)ph ^ b};
    grid_barrier<VAR>(ctr, 2 * ph, G);
    const unsigned o = (b + 3) % G;
    const uint4 v = data[(size_t)o * 256 + t];
    nbad += (v.x != (unsigned)ph || v.y != o || v.z != t);
    grid_barrier<VAR>(ctr, 2 * ph + 1, G);   // nobody overwrites before everybody has read (a real phase chain alternates buffers; two barriers per phase here)
  }
  if (nbad) atomicAdd(bad, nbad);
}

__global__ void empty_kernel(unsigned* p) { if (p == nullptr) __builtin_trap(); }
__global__ __launch_bounds__(256) void phase_kernel(uint4* data, unsigned* bad, int ph) {
  const unsigned G = gridDim.x, b = blockIdx.x, t = threadIdx.x;
  const unsigned o = (b + 3) % G;
  const uint4 v = data[(size_t)o * 256 + t];   // what the previous launch wrote
  if (ph > 0 && (v.x != (unsigned)(ph - 1) || v.y != o)) atomicAdd(bad, 1u);
  data[(size_t)G * 256 + (size_t)b * 256 + t] = v;
  (void)v;
}
__global__ __launch_bounds__(256) void phase_write_kernel(uint4* data, int ph) {
  data[(size_t)blockIdx.x * 256 + threadIdx.x] = uint4{(unsigned)ph, blockIdx.x, threadIdx.x, 0u};
}

template <int VAR>
static void run_persistent(int G, int phases, unsigned* ctr, uint4* data, unsigned* bad) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  float best = 1e30f;
  unsigned hbad = 0;
  for (int rep = 0; rep < 5; ++rep) {
    CK(hipMemset(ctr, 0, 4096)); CK(hipMemset(bad, 0, 4));
    void* args[] = {&ctr, &data, &bad, &phases};
    CK(hipEventRecord(e0, 0));
    CK(hipLaunchCooperativeKernel((const void*)persistent_kernel<VAR>, dim3(G), dim3(256), args, 0, 0));
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    if (ms < best) best = ms;
    unsigned hb; CK(hipMemcpy(&hb, bad, 4, hipMemcpyDeviceToHost)); hbad += hb;
  }
  printf("{\"what\": \"persistent kernel, barrier variant %d\", \"workgroups\": %d, \"barriers\": %d, \"us_per_barrier\": %.3f, \"stale_reads\": %u}\n", VAR, G, 2 * phases,
         best * 1e3f / (2 * phases), hbad);
}

int main() {
  unsigned *ctr, *bad; uint4* data;
  CK(hipMalloc(&ctr, 4096)); CK(hipMalloc(&bad, 4)); CK(hipMalloc(&data, (size_t)2 * 1024 * 256 * 16));
  const int phases = 500;
  for (int G : {256, 512, 1024}) { run_persistent<0>(G, phases, ctr, data, bad); run_persistent<1>(G, phases, ctr, data, bad); }
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int mode = 0; mode < 3; ++mode) {
    float best = 1e30f;
    const int n = 1000;
    for (int rep = 0; rep < 5; ++rep) {
      CK(hipMemset(bad, 0, 4));
      CK(hipDeviceSynchronize());
      CK(hipEventRecord(e0, 0));
      for (int i = 0; i < n; ++i) {
        if (mode == 0) hipLaunchKernelGGL(empty_kernel, dim3(1), dim3(64), 0, 0, ctr);
        else if (mode == 1) hipLaunchKernelGGL(empty_kernel, dim3(256), dim3(256), 0, 0, ctr);
        else { hipLaunchKernelGGL(phase_write_kernel, dim3(256), dim3(256), 0, 0, data, i); hipLaunchKernelGGL(phase_kernel, dim3(256), dim3(256), 0, 0, data, bad, i + 1); }
      }
      CK(hipEventRecord(e1, 0));
      CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      if (ms < best) best = ms;
    }
    unsigned hb; CK(hipMemcpy(&hb, bad, 4, hipMemcpyDeviceToHost));
    printf("{\"what\": \"%s\", \"launches\": %d, \"us_per_launch\": %.3f, \"stale_reads\": %u}\n",
           mode == 0 ? "dependent launches of an empty 1-workgroup kernel" : mode == 1 ? "dependent launches of an empty 256-workgroup kernel" : "dependent launches: 256 workgroups write 4 KB each, the next launch reads another workgroup's",
           mode == 2 ? 2 * n : n, best * 1e3f / (mode == 2 ? 2 * n : n), hb);
  }
  return 0;
}
