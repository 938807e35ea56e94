// hipemu runtime — TEST-ONLY.  See include/hip/hip_runtime.h for scope.
// One fiber per GPU thread, a hand-rolled x86-64 context switch, a
// round-robin scheduler per workgroup, workgroups executed one after another.
#include <hip/hip_runtime.h>

#include <chrono>
#include <vector>

#if !defined(__x86_64__)
#error "hipemu's context switch is written for x86-64 only"
#endif

extern "C" void hipemu_switch(void** save_sp, void* load_sp);
asm(R"(
.text
.globl hipemu_switch
.type hipemu_switch,@function
hipemu_switch:
  pushq %rbp
  pushq %rbx
  pushq %r12
  pushq %r13
  pushq %r14
  pushq %r15
  movq %rsp, (%rdi)
  movq %rsi, %rsp
  popq %r15
  popq %r14
  popq %r13
  popq %r12
  popq %rbx
  popq %rbp
  ret
.size hipemu_switch,.-hipemu_switch
)");

namespace hipemu {

Fiber* cur = nullptr;
uint3 g_block, g_bdim, g_gdim;

static constexpr size_t kStack = 256 * 1024;
static constexpr unsigned kMaxThreads = 1024;
static constexpr unsigned kSlot = 96;  // bytes per lane per exchange

struct Wave {
  unsigned alive;
  unsigned opid[2];
  unsigned arrived[2];
  unsigned long long present[2];
  unsigned char buf[2][64][kSlot];
  unsigned char res[2][64][64];
};

static void* sched_sp;
static std::vector<Fiber> fibers;
static std::vector<Wave> waves;
static unsigned char* stacks = nullptr;
static const std::function<void()>* body_fn;
static unsigned n_alive, n_at_barrier;
static bool progress;

static void yield() { hipemu_switch(&cur->sp, sched_sp); }

// pending LDS-DMA transfers per thread of the running block (see hip_runtime.h: dma_issue / dma_retire)
struct PendingDma { void* dst; unsigned size; unsigned char data[16]; };
static std::vector<std::vector<PendingDma>> pending_dma;
static int wave_serial = 0;  // set_schedule(1): see the block runner
void set_schedule(int mode) { wave_serial = mode; }
static int dma_early = 0;  // 0: transfers land at the wait (exposes a missing wait: RAW); 1: at issue (exposes restaging a buffer that is still being read: WAR)
void dma_issue(void* lds_dst, const void* src, unsigned size) {
  if (size > 16) { fprintf(stderr, "hipemu: LDS-DMA of %u bytes per lane\n", size); abort(); }
  if (dma_early) { memcpy(lds_dst, src, size); return; }
  if (pending_dma.size() <= cur->flat) pending_dma.resize(cur->flat + 1);
  PendingDma d; d.dst = lds_dst; d.size = size; memcpy(d.data, src, size);
  pending_dma[cur->flat].push_back(d);
}
void set_dma_mode(int early) { dma_early = early; }
void dma_retire(unsigned keep_newest) {   // s_waitcnt vmcnt(keep_newest): in-order completion, the newest `keep_newest` transfers stay in flight
  if (pending_dma.size() <= cur->flat) return;
  std::vector<PendingDma>& q = pending_dma[cur->flat];
  if (q.size() <= keep_newest) return;
  const size_t n = q.size() - keep_newest;
  for (size_t i = 0; i < n; ++i) memcpy(q[i].dst, q[i].data, q[i].size);
  q.erase(q.begin(), q.begin() + n);
}

static void fiber_main() {
  (*body_fn)();
  dma_retire(0);  // a thread's outstanding transfers complete at the latest when it ends
  Fiber* f = cur;
  f->state = 2;
  waves[f->wave].alive--;
  n_alive--;
  progress = true;
  hipemu_switch(&f->sp, sched_sp);
  abort();  // never resumed
}

void barrier() {
  cur->state = 1;
  n_at_barrier++;
  progress = true;
  yield();
}

const unsigned char* wave_exchange(const void* in, unsigned bytes, unsigned* stride, unsigned long long* present) {
  if (bytes > kSlot) { fprintf(stderr, "hipemu: exchange too large\n"); abort(); }
  Fiber* f = cur;
  Wave& w = waves[f->wave];
  unsigned op = ++f->opcount;
  unsigned p = op & 1;
  if (w.opid[p] != op) { w.opid[p] = op; w.arrived[p] = 0; w.present[p] = 0; }
  memcpy(w.buf[p][f->lane], in, bytes);
  w.present[p] |= 1ull << f->lane;
  w.arrived[p]++;
  progress = true;
  while (w.arrived[p] < w.alive) yield();
  *stride = kSlot;
  *present = w.present[p];
  return &w.buf[p][0][0];
}

// Collective with a wave-level result: every lane deposits `bytes`; the LAST lane to arrive runs
// `compute` once over the whole table (writing 64 x out_bytes results); every lane then copies its own.
void wave_collective(const void* in, unsigned bytes, void (*compute)(const unsigned char* tab, unsigned stride, unsigned char* out),
                     unsigned out_bytes, void* my_out) {
  if (bytes > kSlot || out_bytes > 64) { fprintf(stderr, "hipemu: collective too large\n"); abort(); }
  Fiber* f = cur;
  Wave& w = waves[f->wave];
  unsigned op = ++f->opcount;
  unsigned p = op & 1;
  if (w.opid[p] != op) { w.opid[p] = op; w.arrived[p] = 0; w.present[p] = 0; }
  memcpy(w.buf[p][f->lane], in, bytes);
  w.present[p] |= 1ull << f->lane;
  w.arrived[p]++;
  progress = true;
  if (w.arrived[p] >= w.alive) {
    if (w.present[p] != ~0ull) { fprintf(stderr, "hipemu: MFMA issued with a partial wave (exec mask %016llx)\n", w.present[p]); abort(); }
    compute(&w.buf[p][0][0], kSlot, &w.res[p][0][0]);
  } else {
    while (w.arrived[p] < w.alive) yield();
  }
  memcpy(my_out, w.res[p][f->lane], out_bytes);
}

static void run_block(dim3 block) {
  unsigned T = block.x * block.y * block.z;
  if (T > kMaxThreads) { fprintf(stderr, "hipemu: block too large\n"); abort(); }
  if (!stacks) stacks = (unsigned char*)aligned_alloc(4096, kStack * kMaxThreads);
  fibers.assign(T, Fiber{});
  unsigned nw = (T + 63) / 64;
  waves.resize(nw);
  for (unsigned w = 0; w < nw; ++w) {
    waves[w].alive = std::min(64u, T - w * 64);
    waves[w].opid[0] = waves[w].opid[1] = 0xffffffffu;
    waves[w].arrived[0] = waves[w].arrived[1] = 0;
    waves[w].present[0] = waves[w].present[1] = 0;
  }
  for (unsigned t = 0; t < T; ++t) {
    Fiber& f = fibers[t];
    f.flat = t; f.lane = t & 63; f.wave = t >> 6; f.state = 0; f.opcount = 0;
    f.tid.x = t % block.x; f.tid.y = (t / block.x) % block.y; f.tid.z = t / (block.x * block.y);
    uintptr_t top = ((uintptr_t)(stacks + (size_t)(t + 1) * kStack)) & ~(uintptr_t)15;
    void** s = (void**)top;
    *--s = nullptr;               // fake return address of fiber_main (keeps rsp%16==8 at entry)
    *--s = (void*)&fiber_main;    // popped by `ret`
    for (int i = 0; i < 6; ++i) *--s = nullptr;  // rbp rbx r12..r15
    f.sp = (void*)s;
  }
  n_alive = T; n_at_barrier = 0;
  static int reverse = -1;
  if (reverse < 0) {
    const char* e = getenv("HIPEMU_ORDER");
    reverse = (e && !strcmp(e, "reverse")) ? 1 : 0;
    if (e && !strcmp(e, "wave_serial")) wave_serial = 1;   // HIPEMU_ORDER=wave_serial: the adversarial schedule for a whole test run
  }
  while (n_alive) {
    progress = false;
    if (wave_serial) {
      // adversarial schedule: every wave runs as far as it can (to its next barrier or its end) before the next wave
      // moves at all — the waves of a workgroup are maximally out of step between barriers, as they may be on hardware
      for (unsigned w = 0; w < nw; ++w) {
        bool moved = true;
        while (moved) {
          moved = false;
          for (unsigned t = w * 64; t < std::min(T, (w + 1) * 64); ++t) {
            Fiber& f = fibers[t];
            if (f.state != 0) continue;
            cur = &f;
            hipemu_switch(&sched_sp, f.sp);
            moved = true;
          }
        }
      }
    } else
    for (unsigned i = 0; i < T; ++i) {
      unsigned t = reverse ? T - 1 - i : i;
      Fiber& f = fibers[t];
      if (f.state != 0) continue;
      cur = &f;
      hipemu_switch(&sched_sp, f.sp);
    }
    if (n_alive && n_at_barrier == n_alive) {
      for (auto& f : fibers) if (f.state == 1) f.state = 0;
      n_at_barrier = 0;
      progress = true;
    }
    if (!progress && n_alive) {
      fprintf(stderr, "hipemu: deadlock in block (%u,%u,%u): %u alive, %u at barrier (divergent barrier / wave op?)\n",
              g_block.x, g_block.y, g_block.z, n_alive, n_at_barrier);
      abort();
    }
  }
  cur = nullptr;
}

void launch(dim3 grid, dim3 block, const std::function<void()>& body) {
  body_fn = &body;
  g_bdim = uint3{block.x, block.y, block.z};
  g_gdim = uint3{grid.x, grid.y, grid.z};
  for (unsigned z = 0; z < grid.z; ++z)
    for (unsigned y = 0; y < grid.y; ++y)
      for (unsigned x = 0; x < grid.x; ++x) {
        g_block = uint3{x, y, z};
        run_block(block);
      }
}

}  // namespace hipemu

struct hipemu_event { std::chrono::steady_clock::time_point t; };

extern "C" {
// device memory is handed out POISONED (0x7F bytes: 3.4e38 as float — finite, so max-based range tracking sees it, unlike a NaN — and 2139062143 as int): a kernel that reads what nothing wrote — ragged
// tails, scratch it assumes zeroed — produces NaNs / trips the range guard deterministically instead of depending on what
// the allocator happened to return
hipError_t hipMalloc(void** p, size_t n) {
  const size_t bytes = (n + 255) & ~(size_t)255;
  *p = aligned_alloc(256, bytes);
  if (*p) memset(*p, 0x7F, bytes);
  return *p ? hipSuccess : hipErrorOutOfMemory;
}
hipError_t hipFree(void* p) { free(p); return hipSuccess; }
hipError_t hipHostMalloc(void** p, size_t n, unsigned) { *p = calloc(1, (n + 255) & ~(size_t)255); return *p ? hipSuccess : hipErrorOutOfMemory; }
hipError_t hipHostFree(void* p) { free(p); return hipSuccess; }
hipError_t hipMemcpy(void* d, const void* s, size_t n, hipMemcpyKind) { memmove(d, s, n); return hipSuccess; }
hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind, hipStream_t) { memmove(d, s, n); return hipSuccess; }
hipError_t hipMemset(void* d, int v, size_t n) { memset(d, v, n); return hipSuccess; }
hipError_t hipMemsetAsync(void* d, int v, size_t n, hipStream_t) { memset(d, v, n); return hipSuccess; }
hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
hipError_t hipDeviceSynchronize() { return hipSuccess; }
hipError_t hipGetLastError() { return hipSuccess; }
hipError_t hipPeekAtLastError() { return hipSuccess; }
const char* hipGetErrorString(hipError_t e) { return e == hipSuccess ? "hipSuccess" : "hipemu error"; }
hipError_t hipEventCreate(hipEvent_t* e) { *e = new hipemu_event(); return hipSuccess; }
hipError_t hipEventDestroy(hipEvent_t e) { delete e; return hipSuccess; }
hipError_t hipEventRecord(hipEvent_t e, hipStream_t) { e->t = std::chrono::steady_clock::now(); return hipSuccess; }
hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
hipError_t hipEventElapsedTime(float* ms, hipEvent_t a, hipEvent_t b) {
  *ms = std::chrono::duration<float, std::milli>(b->t - a->t).count();
  return hipSuccess;
}
hipError_t hipGetDevice(int* d) { *d = 0; return hipSuccess; }
hipError_t hipSetDevice(int) { return hipSuccess; }
hipError_t hipGetDeviceCount(int* n) { *n = 1; return hipSuccess; }
}

// test switch (ctypes): when the emulated LDS-DMA transfers land — see dma_issue
extern "C" void hipemu_set_dma_mode(int early) { hipemu::set_dma_mode(early); }
// test switch: 0 = all threads advance one step per round (default), 1 = wave after wave between barriers
extern "C" void hipemu_set_schedule(int mode) { hipemu::set_schedule(mode); }
