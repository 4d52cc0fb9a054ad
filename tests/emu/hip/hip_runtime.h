// TEST INFRASTRUCTURE — a CPU stand-in for <hip/hip_runtime.h>, never part of the product.
//
// tests/emu/run_emulated.py compiles every meilisearch_amd/csrc/*.hip as plain C++ (ROCm clang, -I tests/emu) against
// THIS header, so that the CPU test tier executes the very kernel source hipcc compiles for gfx950 — grid / block
// index arithmetic, wave ballots and shuffles, MFMA fragment layouts, LDS (static and dynamic), atomics, the "last
// workgroup publishes" completion protocol — without a GPU (tests/test_kernels_emulated_cpu.py runs the GPU test
// files on it).  It checks kernel LOGIC only: it says nothing about speed, memory-model races between workgroups, or
// code generation.  libmsi.so is never built from it and meilisearch_amd never loads it.
//
// Execution model: the workgroups of a launch run one after the other on the launching thread (launches from
// several host threads are serialised); the threads of a workgroup are fibers that run until they reach a wave
// collective (__ballot, __shfl*, __any, __all, readfirstlane, wave_barrier, MFMA) or __syncthreads.  When no lane of a
// 64-wide wave can run, the lanes waiting at a collective exchange their values (lanes that exited or wait elsewhere
// are inactive, as on the hardware); when nothing in the workgroup can run, the __syncthreads waiters are released
// together.  MFMA: D = A x B + C over the wave with the CDNA3/4 fragment layouts (lane l: A[l % 16][K * (l / 16) ..],
// B[K * (l / 16) ..][l % 16], D[4 * (l / 16) + r][l % 16]); the accumulation order inside one instruction is not the
// hardware's — the scan it serves is a candidate generator whose results are re-scored in reference arithmetic.
// __shared__ variables are `static` (one workgroup at a time).  "Device" memory is host memory filled with 0xCD at
// allocation so that a read of uninitialised memory shows.
#pragma once
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <ucontext.h>
#include <pthread.h>
#include <signal.h>

#include <algorithm>
#include <chrono>
#include <functional>
#include <mutex>
#include <tuple>
#include <vector>

#define MSI_HIP_EMULATED 1
#define __global__
#define __device__
#define __host__
#define __forceinline__ inline
#define __shared__ static
#define __launch_bounds__(...)
#define MSI_DYNAMIC_LDS(name) unsigned char *name = hipemu::g.dyn_lds
#define MSI_ORDER_ATOMICS() ((void)0)
#define MSI_RELEASE_DEVICE() ((void)0)
#define MSI_ACQUIRE_DEVICE() ((void)0)
#define MSI_SLEEP() ((void)0)   /* (workgroups of a launch run in block order: what a workgroup waits for has run) */
#define MSI_UNIFORM(x) (x)
#define __HIP_MEMORY_SCOPE_SINGLETHREAD 1
#define __HIP_MEMORY_SCOPE_WAVEFRONT 2
#define __HIP_MEMORY_SCOPE_WORKGROUP 3
#define __HIP_MEMORY_SCOPE_AGENT 4
#define __HIP_MEMORY_SCOPE_SYSTEM 5

using std::max;
using std::min;

struct dim3 {
  unsigned x, y, z;
  constexpr dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
inline dim3 threadIdx, blockIdx, blockDim, gridDim;
struct alignas(16) ulonglong2 { unsigned long long x, y; };
struct alignas(16) uint4 { unsigned x, y, z, w; };
struct alignas(8) uint2 { unsigned x, y; };
struct alignas(16) float4 { float x, y, z, w; };
struct alignas(8) float2 { float x, y; };
inline ulonglong2 make_ulonglong2(unsigned long long x, unsigned long long y) { return ulonglong2{x, y}; }
inline float4 make_float4(float x, float y, float z, float w) { return float4{x, y, z, w}; }
inline float2 make_float2(float x, float y) { return float2{x, y}; }
inline uint4 make_uint4(unsigned x, unsigned y, unsigned z, unsigned w) { return uint4{x, y, z, w}; }
inline uint2 make_uint2(unsigned x, unsigned y) { return uint2{x, y}; }

// ---- host API --------------------------------------------------------------------------------------------------
typedef int hipError_t;
enum { hipSuccess = 0, hipErrorInvalidValue = 1, hipErrorOutOfMemory = 2, hipErrorNotReady = 600, hipErrorNotSupported = 801 };
typedef struct hipemuStream *hipStream_t;
typedef std::chrono::steady_clock::time_point *hipEvent_t;
enum hipMemcpyKind { hipMemcpyHostToHost, hipMemcpyHostToDevice, hipMemcpyDeviceToHost, hipMemcpyDeviceToDevice, hipMemcpyDefault };
enum { hipStreamNonBlocking = 1, hipHostMallocDefault = 0, hipHostMallocCoherent = 0x40000000, hipHostMallocMapped = 2 };
struct hipDeviceProp_t {
  char gcnArchName[256];
  char name[256];
  int multiProcessorCount;
  size_t totalGlobalMem;
};
inline const char *hipGetErrorString(hipError_t e) { return e == hipSuccess ? "hipSuccess" : e == hipErrorOutOfMemory ? "hipErrorOutOfMemory" : "hipError(emulated)"; }
inline hipError_t hipGetLastError() { return hipSuccess; }
enum hipFuncAttribute { hipFuncAttributeMaxDynamicSharedMemorySize = 8 };
template <typename F>
inline hipError_t hipFuncSetAttribute(F, hipFuncAttribute, int) { return hipSuccess; }
template <typename F>
inline hipError_t hipOccupancyMaxActiveBlocksPerMultiprocessor(int *n, F, int, size_t) { *n = 4; return hipSuccess; }
// MSI_EMU_DEVICES emulated devices (default 1): they share the host's memory and the one launch lock — what differs per
// device is what the host code keeps per device (contexts, streams, stores, the current-device guard of every entry point)
inline int hipemu_device_count() {
  const char *e = getenv("MSI_EMU_DEVICES");
  const int n = e ? atoi(e) : 1;
  return n < 1 ? 1 : (n > 64 ? 64 : n);
}
inline thread_local int hipemu_current_device = 0;
inline hipError_t hipGetDeviceCount(int *n) { *n = hipemu_device_count(); return hipSuccess; }
inline hipError_t hipGetDevice(int *d) { *d = hipemu_current_device; return hipSuccess; }
inline hipError_t hipSetDevice(int d) {
  if (d < 0 || d >= hipemu_device_count()) return hipErrorInvalidValue;
  hipemu_current_device = d;
  return hipSuccess;
}
inline hipError_t hipGetDeviceProperties(hipDeviceProp_t *p, int dev) {
  if (dev < 0 || dev >= hipemu_device_count()) return hipErrorInvalidValue;
  memset(p, 0, sizeof(*p));
  strcpy(p->gcnArchName, "gfx950:emulated-on-cpu");
  strcpy(p->name, "hip emulation (tests/emu)");
  const char *cu = getenv("MSI_EMU_CUS");
  p->multiProcessorCount = cu ? atoi(cu) : 2;  // small grids: the kernels are grid-stride or cover by blocks
  p->totalGlobalMem = (size_t)1 << 32;
  return hipSuccess;
}
struct hipemuStream { int device; };
inline hipError_t hipStreamCreate(hipStream_t *s) {   // a stream belongs to the device that was current when it was made
  *s = (hipStream_t)malloc(sizeof(hipemuStream));
  (*s)->device = hipemu_current_device;
  return hipSuccess;
}
// work enqueued on a stream of another device than the current one is an error on the real runtime
// (hipErrorInvalidResourceHandle / hipErrorContextIsDestroyed): the emulation stops, so that the CPU tier catches a
// multi-device entry point that forgot its device guard
inline void hipemu_check_stream(hipStream_t s, const char *what) {
  if (s && s->device != hipemu_current_device) {
    fprintf(stderr, "hipemu: %s on a stream of device %d while device %d is current\n", what, s->device, hipemu_current_device);
    abort();
  }
}
inline hipError_t hipMalloc(void **p, size_t n) {
  // MSI_EMU_FAIL_MALLOC=<k> (tests: set and cleared between calls): HBM is exhausted after k more allocations — the
  // count starts whenever the variable's value changes
  if (const char *lim = getenv("MSI_EMU_FAIL_MALLOC")) {
    static char seen[32] = "";
    static long left = 0;
    if (strncmp(seen, lim, sizeof(seen) - 1) != 0) {
      strncpy(seen, lim, sizeof(seen) - 1);
      left = atol(lim);
    }
    if (left <= 0) {
      *p = nullptr;
      return hipErrorOutOfMemory;
    }
    --left;
  }
  *p = malloc(n ? n : 1);
  if (!*p) return hipErrorOutOfMemory;
  memset(*p, getenv("MSI_EMU_FILL") ? atoi(getenv("MSI_EMU_FILL")) : 0xCD, n);
  return hipSuccess;
}
template <typename T>
inline hipError_t hipMalloc(T **p, size_t n) { return hipMalloc((void **)p, n); }
inline hipError_t hipFree(void *p) { free(p); return hipSuccess; }
inline hipError_t hipHostMalloc(void **p, size_t n, unsigned = 0) { return hipMalloc(p, n); }
template <typename T>
inline hipError_t hipHostMalloc(T **p, size_t n, unsigned f = 0) { return hipMalloc((void **)p, n); }
inline hipError_t hipHostFree(void *p) { free(p); return hipSuccess; }
inline hipError_t hipMemcpy(void *d, const void *s, size_t n, hipMemcpyKind) { if (n) memmove(d, s, n); return hipSuccess; }
inline hipError_t hipMemcpyAsync(void *d, const void *s, size_t n, hipMemcpyKind, hipStream_t st = nullptr) { hipemu_check_stream(st, "hipMemcpyAsync"); if (n) memmove(d, s, n); return hipSuccess; }
inline hipError_t hipMemset(void *d, int v, size_t n) { if (n) memset(d, v, n); return hipSuccess; }
inline hipError_t hipMemsetAsync(void *d, int v, size_t n, hipStream_t st = nullptr) { hipemu_check_stream(st, "hipMemsetAsync"); if (n) memset(d, v, n); return hipSuccess; }
inline hipError_t hipStreamCreateWithFlags(hipStream_t *s, unsigned) { return hipStreamCreate(s); }
inline hipError_t hipStreamCreateWithPriority(hipStream_t *s, unsigned, int) { return hipStreamCreate(s); }
inline hipError_t hipExtStreamCreateWithCUMask(hipStream_t *s, unsigned, const unsigned *) { return hipStreamCreate(s); }
inline hipError_t hipDeviceGetStreamPriorityRange(int *least, int *greatest) { *least = *greatest = 0; return hipSuccess; }
inline hipError_t hipStreamDestroy(hipStream_t s) { free(s); return hipSuccess; }
inline hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
inline hipError_t hipDeviceSynchronize() { return hipSuccess; }
inline hipError_t hipEventCreate(hipEvent_t *e) { *e = new std::chrono::steady_clock::time_point(); return hipSuccess; }
inline hipError_t hipEventDestroy(hipEvent_t e) { delete e; return hipSuccess; }
enum { hipEventDefault = 0, hipEventBlockingSync = 1, hipEventDisableTiming = 2 };
inline hipError_t hipEventCreateWithFlags(hipEvent_t *e, unsigned) { return hipEventCreate(e); }
inline hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
inline hipError_t hipStreamWaitEvent(hipStream_t s, hipEvent_t, unsigned = 0) { hipemu_check_stream(s, "hipStreamWaitEvent"); return hipSuccess; }   // launches run to completion before they return
inline hipError_t hipEventQuery(hipEvent_t) { return hipSuccess; }   // launches run to completion before they return
inline hipError_t hipEventRecord(hipEvent_t e, hipStream_t = nullptr) { *e = std::chrono::steady_clock::now(); return hipSuccess; }
inline hipError_t hipEventElapsedTime(float *ms, hipEvent_t a, hipEvent_t b) { *ms = std::chrono::duration<float, std::milli>(*b - *a).count(); return hipSuccess; }

// ---- the fiber scheduler -------------------------------------------------------------------------------------------
namespace hipemu {

constexpr size_t STACK_BYTES = 64 * 1024;
enum LaneState { RUNNABLE, WAIT_WAVE, WAIT_BLOCK, DONE };
enum WaveOp { OP_BALLOT, OP_SHFL, OP_SHFL_XOR, OP_SHFL_DOWN, OP_SHFL_UP, OP_ANY, OP_ALL, OP_FIRST, OP_MFMA_F32X4, OP_MFMA_BF16X32, OP_MFMA_I8X64 };

// Context switch: on x86-64 six callee-saved registers and the stack pointer (swapcontext would make two signal-mask
// system calls per switch — most of the run time of this tier); ucontext elsewhere.
#if defined(__x86_64__)
#define HIPEMU_FAST_SWITCH 1
static __attribute__((naked, noinline)) void fiber_switch(void ** /*save_sp: rdi*/, void * /*next_sp: rsi*/) {
  __asm__ volatile(
      "pushq %rbp\n\tpushq %rbx\n\tpushq %r12\n\tpushq %r13\n\tpushq %r14\n\tpushq %r15\n\t"
      "movq %rsp, (%rdi)\n\t"
      "movq %rsi, %rsp\n\t"
      "popq %r15\n\tpopq %r14\n\tpopq %r13\n\tpopq %r12\n\tpopq %rbx\n\tpopq %rbp\n\t"
      "ret\n\t");
}
struct Context {
  void *sp = nullptr;
};
#else
struct Context {
  ucontext_t uc;
};
#endif

struct Lane {
  Context ctx;
  LaneState state = DONE;
  int op = 0;
  int param = 0;
  uint64_t value = 0;   // deposited by the lane, replaced by the result
  float fa[8], fb[8], fc[4];  // MFMA operands of the lane (A and B fragments widened to f32), C in / D out
  signed char ia[16], ib[16];  // v_mfma_i32_16x16x64_i8: the lane's 16 bytes of A and of B
  int ic[4];                   // ... C in / D out
  int tag = 0;                 // MSI_EMU_TAG: what the lane is doing (store accounting)
  void *stack = nullptr;
};

struct Sched {
  Context main;
  std::vector<Lane> lanes;
  const std::function<void()> *body = nullptr;
  unsigned cur = 0, n = 0;
  uint64_t launches = 0, blocks = 0;
  bool in_kernel = false;
  unsigned char *dyn_lds = nullptr;   // the launch's dynamic shared memory (16-byte aligned, poisoned per workgroup)
  size_t dyn_bytes = 0;
};
inline Sched g;
inline std::mutex launch_mu;  // host threads (one search per pool, many in flight) launch one at a time

inline void set_thread(unsigned t) {
  g.cur = t;
  threadIdx.x = t % blockDim.x;
  threadIdx.y = (t / blockDim.x) % blockDim.y;
  threadIdx.z = t / (blockDim.x * blockDim.y);
}

inline void switch_context(Context &from, Context &to) {
#ifdef HIPEMU_FAST_SWITCH
  fiber_switch(&from.sp, to.sp);
#else
  swapcontext(&from.uc, &to.uc);
#endif
}

inline void trampoline() {
  (*g.body)();
  g.lanes[g.cur].state = DONE;
  switch_context(g.lanes[g.cur].ctx, g.main);
  abort();  // a finished lane is never resumed
}

inline void prepare_lane(Lane &l) {
#ifdef HIPEMU_FAST_SWITCH
  // [r15 r14 r13 r12 rbx rbp | return address = trampoline | slot of the caller's return address]: the stack pointer
  // is 8 mod 16 when the trampoline is entered, as after a call
  uintptr_t top = ((uintptr_t)l.stack + STACK_BYTES) & ~(uintptr_t)15;
  void **sp = (void **)(top - 8 * 8);
  for (int i = 0; i < 6; ++i) sp[i] = nullptr;
  sp[6] = (void *)&trampoline;
  sp[7] = nullptr;
  l.ctx.sp = sp;
#else
  getcontext(&l.ctx.uc);
  l.ctx.uc.uc_stack.ss_sp = l.stack;
  l.ctx.uc.uc_stack.ss_size = STACK_BYTES;
  l.ctx.uc.uc_link = &g.main.uc;
  makecontext(&l.ctx.uc, (void (*)())trampoline, 0);
#endif
}

inline void yield_to_scheduler() {
  const unsigned me = g.cur;
  switch_context(g.lanes[me].ctx, g.main);
  set_thread(me);
}

inline uint64_t wave_collective(int op, int param, uint64_t value) {
  Lane &l = g.lanes[g.cur];
  l.op = op;
  l.param = param;
  l.value = value;
  l.state = WAIT_WAVE;
  yield_to_scheduler();
  return g.lanes[g.cur].value;
}

inline void resolve_wave(unsigned w0, unsigned w1) {
  // the lanes of [w0, w1) that wait at a collective, grouped by (op, param): divergent branches resolve separately
  for (;;) {
    int op = -1, param = 0;
    for (unsigned t = w0; t < w1; ++t)
      if (g.lanes[t].state == WAIT_WAVE) { op = g.lanes[t].op; param = g.lanes[t].param; break; }
    if (op < 0) return;
    bool in[64] = {};
    uint64_t val[64] = {};
    for (unsigned t = w0; t < w1; ++t)
      if (g.lanes[t].state == WAIT_WAVE && g.lanes[t].op == op && g.lanes[t].param == param) { in[t - w0] = true; val[t - w0] = g.lanes[t].value; }
    uint64_t ballot = 0;
    bool any = false, all = true;
    for (unsigned i = 0; i < 64; ++i)
      if (in[i]) {
        if (val[i]) { ballot |= 1ull << i; any = true; } else all = false;
      }
    if (op == OP_MFMA_I8X64) {
      // v_mfma_i32_16x16x64_i8: lane l holds A[l % 16][16 * (l / 16) .. + 16) and B[the same k][l % 16] as 16 signed bytes each,
      // D[4 * (l / 16) + r][l % 16] in its r-th accumulator register; exact i32 accumulation
      int D[16][16];
      for (int i = 0; i < 16; ++i)
        for (int j = 0; j < 16; ++j) {
          int acc = 0;
          for (int g4 = 0; g4 < 4; ++g4)
            for (int t = 0; t < 16; ++t) {
              const unsigned la = (unsigned)(g4 * 16 + i), lb = (unsigned)(g4 * 16 + j);
              if (w0 + la < w1 && w0 + lb < w1 && in[la] && in[lb]) acc += (int)g.lanes[w0 + la].ia[t] * (int)g.lanes[w0 + lb].ib[t];
            }
          D[i][j] = acc;
        }
      for (unsigned l = 0; l < w1 - w0; ++l)
        if (in[l]) {
          Lane &ln = g.lanes[w0 + l];
          for (int r = 0; r < 4; ++r) ln.ic[r] += D[4 * (l / 16) + r][l % 16];
          ln.state = RUNNABLE;
        }
      continue;
    }
    if (op == OP_MFMA_F32X4 || op == OP_MFMA_BF16X32) {
      // D = A x B + C over the whole wave (CDNA3/4 ISA fragment layouts): lane l holds A[l % 16][kb .. kb + K),
      // B[kb .. kb + K)[l % 16] with kb = K * (l / 16) (K = 1 for 16x16x4 f32, 8 for 16x16x32 bf16) and
      // D[4 * (l / 16) + r][l % 16] in its r-th accumulator register.  A lane that is not here contributes zeros.
      const int K = op == OP_MFMA_F32X4 ? 1 : 8;
      float D[16][16];
      for (int i = 0; i < 16; ++i)
        for (int j = 0; j < 16; ++j) {
          float acc = 0.f;
          for (int g4 = 0; g4 < 4; ++g4)
            for (int t = 0; t < K; ++t) {
              const unsigned la = (unsigned)(g4 * 16 + i), lb = (unsigned)(g4 * 16 + j);
              if (w0 + la < w1 && w0 + lb < w1 && in[la] && in[lb]) acc += g.lanes[w0 + la].fa[t] * g.lanes[w0 + lb].fb[t];
            }
          D[i][j] = acc;
        }
      for (unsigned l = 0; l < w1 - w0; ++l)
        if (in[l]) {
          Lane &ln = g.lanes[w0 + l];
          for (int r = 0; r < 4; ++r) ln.fc[r] += D[4 * (l / 16) + r][l % 16];
          ln.state = RUNNABLE;
        }
      continue;
    }
    unsigned first_lane = 64;
    for (unsigned i = 0; i < 64; ++i)
      if (in[i]) { first_lane = i; break; }
    for (unsigned i = 0; i < w1 - w0; ++i) {
      if (!in[i]) continue;
      uint64_t r = val[i];
      int src = -1;
      switch (op) {
        case OP_BALLOT: r = ballot; break;
        case OP_ANY: r = any; break;
        case OP_ALL: r = all; break;
        case OP_FIRST: r = val[first_lane]; break;
        case OP_SHFL: src = param & 63; break;
        case OP_SHFL_XOR: src = (int)(i ^ (unsigned)param); break;
        case OP_SHFL_DOWN: src = (int)i + param; break;
        case OP_SHFL_UP: src = (int)i - param; break;
      }
      if (src >= 0 && src < 64 && in[src]) r = val[src];
      g.lanes[w0 + i].value = r;
      g.lanes[w0 + i].state = RUNNABLE;
    }
  }
}

inline void run_block() {
  const unsigned n = g.n;
  for (unsigned t = 0; t < n; ++t) {
    Lane &l = g.lanes[t];
    prepare_lane(l);
    l.tag = 0;
    l.state = RUNNABLE;
  }
  for (;;) {
    bool ran = false;
    for (unsigned t = 0; t < n; ++t)
      if (g.lanes[t].state == RUNNABLE) {
        set_thread(t);
        switch_context(g.main, g.lanes[t].ctx);
        ran = true;
      }
    bool released = false;
    for (unsigned w0 = 0; w0 < n; w0 += 64) {
      const unsigned w1 = std::min(n, w0 + 64);
      bool waiting = false;
      for (unsigned t = w0; t < w1; ++t) waiting |= g.lanes[t].state == WAIT_WAVE;
      if (waiting) { resolve_wave(w0, w1); released = true; }
    }
    if (released) continue;
    bool any_block = false, all_done = true;
    for (unsigned t = 0; t < n; ++t) {
      any_block |= g.lanes[t].state == WAIT_BLOCK;
      all_done &= g.lanes[t].state == DONE;
    }
    if (any_block) {
      for (unsigned t = 0; t < n; ++t)
        if (g.lanes[t].state == WAIT_BLOCK) g.lanes[t].state = RUNNABLE;
      continue;
    }
    if (all_done) return;
    if (!ran) { fprintf(stderr, "hipemu: workgroup cannot make progress\n"); abort(); }
  }
}

inline void launch(dim3 grid, dim3 block, size_t shmem, const std::function<void()> &body) {
  if (shmem > 160 * 1024) { fprintf(stderr, "hipemu: %zu bytes of LDS requested (160 KiB per workgroup on gfx950)\n", shmem); abort(); }
  std::lock_guard<std::mutex> lk(launch_mu);   // one launch at a time, whichever emulated device or host thread it comes from
  // a sampling profiler's SIGPROF (tools/kw_leg.py --emulated with KW_PROFILE) must not unwind a lane's hand-made stack
  struct ProfMask {
    sigset_t old;
    ProfMask() { sigset_t s; sigemptyset(&s); sigaddset(&s, SIGPROF); pthread_sigmask(SIG_BLOCK, &s, &old); }
    ~ProfMask() { pthread_sigmask(SIG_SETMASK, &old, nullptr); }
  } prof_mask;
  if (shmem > g.dyn_bytes) {
    free(g.dyn_lds);
    g.dyn_lds = (unsigned char *)aligned_alloc(16, (shmem + 15) & ~(size_t)15);
    g.dyn_bytes = shmem;
  }
  if (g.in_kernel) { fprintf(stderr, "hipemu: nested launch\n"); abort(); }
  const unsigned n = block.x * block.y * block.z;
  if (!n || n > 1024) { fprintf(stderr, "hipemu: bad block size %u\n", n); abort(); }
  if (g.lanes.size() < n) {
    const size_t old = g.lanes.size();
    g.lanes.resize(n);
    for (size_t t = old; t < n; ++t) g.lanes[t].stack = malloc(STACK_BYTES);
  }
  g.in_kernel = true;
  g.body = &body;
  g.n = n;
  blockDim = block;
  gridDim = grid;
  ++g.launches;
  for (unsigned bz = 0; bz < grid.z; ++bz)
    for (unsigned by = 0; by < grid.y; ++by)
      for (unsigned bx = 0; bx < grid.x; ++bx) {
        blockIdx = dim3(bx, by, bz);
        ++g.blocks;
        if (shmem) memset(g.dyn_lds, 0xCD, shmem);
        run_block();
      }
  g.in_kernel = false;
}

template <typename F, typename... A>
inline void launch_kernel(dim3 grid, dim3 block, size_t shmem, F kernel, A... args) {
  // arguments are evaluated once and copied, as a real launch marshals them
  auto packed = std::make_tuple(args...);
  const std::function<void()> body = [&]() { std::apply(kernel, packed); };
  launch(grid, block, shmem, body);
}

template <typename T>
inline uint64_t to_bits(T v) { uint64_t b = 0; static_assert(sizeof(T) <= 8, ""); memcpy(&b, &v, sizeof(T)); return b; }
template <typename T>
inline T from_bits(uint64_t b) { T v; memcpy(&v, &b, sizeof(T)); return v; }

}  // namespace hipemu

#define hipLaunchKernelGGL(kernel, grid, block, shmem, stream, ...) \
  (hipemu_check_stream(stream, "a kernel launch"), hipemu::launch_kernel(dim3(grid), dim3(block), (size_t)(shmem), kernel, ##__VA_ARGS__))

// ---- device intrinsics ----------------------------------------------------------------------------------------------
inline void __syncthreads() {
  hipemu::g.lanes[hipemu::g.cur].state = hipemu::WAIT_BLOCK;
  hipemu::yield_to_scheduler();
}
inline unsigned long long wall_clock64() { return 0; }
inline void __threadfence() {}
inline void __threadfence_block() {}
inline void __threadfence_system() {}
inline unsigned long long __ballot(int pred) { return hipemu::wave_collective(hipemu::OP_BALLOT, 0, pred != 0); }
inline int __any(int pred) { return (int)hipemu::wave_collective(hipemu::OP_ANY, 0, pred != 0); }
inline int __all(int pred) { return (int)hipemu::wave_collective(hipemu::OP_ALL, 0, pred != 0); }
template <typename T>
inline T __shfl(T v, int src, int = 64) { return hipemu::from_bits<T>(hipemu::wave_collective(hipemu::OP_SHFL, src, hipemu::to_bits(v))); }
template <typename T>
inline T __shfl_xor(T v, int m, int = 64) { return hipemu::from_bits<T>(hipemu::wave_collective(hipemu::OP_SHFL_XOR, m, hipemu::to_bits(v))); }
template <typename T>
inline T __shfl_down(T v, unsigned d, int = 64) { return hipemu::from_bits<T>(hipemu::wave_collective(hipemu::OP_SHFL_DOWN, (int)d, hipemu::to_bits(v))); }
template <typename T>
inline T __shfl_up(T v, unsigned d, int = 64) { return hipemu::from_bits<T>(hipemu::wave_collective(hipemu::OP_SHFL_UP, (int)d, hipemu::to_bits(v))); }
inline void hipemu_wave_barrier() { (void)hipemu::wave_collective(hipemu::OP_ANY, -1, 0); }
template <typename T>
inline T hipemu_readfirstlane(T v) { return hipemu::from_bits<T>(hipemu::wave_collective(hipemu::OP_FIRST, 0, hipemu::to_bits(v))); }
#define __builtin_amdgcn_wave_barrier() hipemu_wave_barrier()
#define __builtin_amdgcn_readfirstlane(v) hipemu_readfirstlane(v)
#define __builtin_amdgcn_s_waitcnt(x) ((void)0)
#define __builtin_amdgcn_sched_barrier(x) ((void)0)
#define __builtin_amdgcn_s_setprio(x) ((void)0)
#if defined(__clang__)
typedef float hipemu_f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 hipemu_bf16x8 __attribute__((ext_vector_type(8)));
inline hipemu_f32x4 hipemu_mfma_f32(float a, float b, hipemu_f32x4 c) {
  hipemu::Lane &l = hipemu::g.lanes[hipemu::g.cur];
  l.fa[0] = a; l.fb[0] = b;
  for (int r = 0; r < 4; ++r) l.fc[r] = c[r];
  (void)hipemu::wave_collective(hipemu::OP_MFMA_F32X4, 0, 0);
  hipemu::Lane &m = hipemu::g.lanes[hipemu::g.cur];
  hipemu_f32x4 d;
  for (int r = 0; r < 4; ++r) d[r] = m.fc[r];
  return d;
}
inline hipemu_f32x4 hipemu_mfma_bf16(hipemu_bf16x8 a, hipemu_bf16x8 b, hipemu_f32x4 c) {
  hipemu::Lane &l = hipemu::g.lanes[hipemu::g.cur];
  for (int t = 0; t < 8; ++t) { l.fa[t] = (float)a[t]; l.fb[t] = (float)b[t]; }
  for (int r = 0; r < 4; ++r) l.fc[r] = c[r];
  (void)hipemu::wave_collective(hipemu::OP_MFMA_BF16X32, 0, 0);
  hipemu::Lane &m = hipemu::g.lanes[hipemu::g.cur];
  hipemu_f32x4 d;
  for (int r = 0; r < 4; ++r) d[r] = m.fc[r];
  return d;
}
typedef int hipemu_i32x4 __attribute__((ext_vector_type(4)));
inline hipemu_i32x4 hipemu_mfma_i8(hipemu_i32x4 a, hipemu_i32x4 b, hipemu_i32x4 c) {
  hipemu::Lane &l = hipemu::g.lanes[hipemu::g.cur];
  __builtin_memcpy(l.ia, &a, 16);
  __builtin_memcpy(l.ib, &b, 16);
  for (int r = 0; r < 4; ++r) l.ic[r] = c[r];
  (void)hipemu::wave_collective(hipemu::OP_MFMA_I8X64, 0, 0);
  hipemu::Lane &m = hipemu::g.lanes[hipemu::g.cur];
  hipemu_i32x4 d;
  for (int r = 0; r < 4; ++r) d[r] = m.ic[r];
  return d;
}
#define __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, c, x, y, z) hipemu_mfma_i8((a), (b), (c))
#define __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, x, y, z) hipemu_mfma_f32((a), (b), (c))
#define __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, x, y, z) hipemu_mfma_bf16((a), (b), (c))
#endif
inline unsigned __funnelshift_r(unsigned lo, unsigned hi, unsigned shift) {
  return (unsigned)((((unsigned long long)hi << 32) | lo) >> (shift & 31));
}
inline unsigned __funnelshift_l(unsigned lo, unsigned hi, unsigned shift) {
  return (unsigned)(((((unsigned long long)hi << 32) | lo) << (shift & 31)) >> 32);
}
inline float __fmul_rn(float a, float b) { return a * b; }
inline float __fadd_rn(float a, float b) { return a + b; }
inline float __fsub_rn(float a, float b) { return a - b; }
inline float __fmaf_rn(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
inline float __fdiv_rn(float a, float b) { return a / b; }
inline float __fsqrt_rn(float a) { return __builtin_sqrtf(a); }
inline int __popc(unsigned v) { return __builtin_popcount(v); }
inline int __popcll(unsigned long long v) { return __builtin_popcountll(v); }
inline int __ffs(int v) { return __builtin_ffs(v); }
inline int __ffsll(long long v) { return __builtin_ffsll(v); }
inline int __ffsll(unsigned long long v) { return __builtin_ffsll((long long)v); }
inline int __clz(int v) { return v ? __builtin_clz((unsigned)v) : 32; }
inline int __clzll(long long v) { return v ? __builtin_clzll((unsigned long long)v) : 64; }
inline long long __double_as_longlong(double d) { return hipemu::from_bits<long long>(hipemu::to_bits(d)); }
inline double __longlong_as_double(long long v) { return hipemu::from_bits<double>(hipemu::to_bits(v)); }
inline unsigned __float_as_uint(float f) { return hipemu::from_bits<unsigned>(hipemu::to_bits(f)); }
inline float __uint_as_float(unsigned u) { return hipemu::from_bits<float>(hipemu::to_bits(u)); }

// fibers never preempt each other, so a plain read-modify-write IS atomic here
template <typename T, typename U>
inline T atomicAdd(T *p, U v) { T o = *p; *p = (T)(o + (T)v); return o; }
template <typename T, typename U>
inline T atomicSub(T *p, U v) { T o = *p; *p = (T)(o - (T)v); return o; }
template <typename T, typename U>
inline T atomicMax(T *p, U v) { T o = *p; if ((T)v > o) *p = (T)v; return o; }
template <typename T, typename U>
inline T atomicMin(T *p, U v) { T o = *p; if ((T)v < o) *p = (T)v; return o; }
template <typename T, typename U>
inline T atomicExch(T *p, U v) { T o = *p; *p = (T)v; return o; }
template <typename T, typename U>
inline T atomicOr(T *p, U v) { T o = *p; *p = (T)(o | (T)v); return o; }
template <typename T, typename U>
inline T atomicAnd(T *p, U v) { T o = *p; *p = (T)(o & (T)v); return o; }
template <typename T, typename U>
inline T atomicXor(T *p, U v) { T o = *p; *p = (T)(o ^ (T)v); return o; }
template <typename T, typename U, typename V>
inline T atomicCAS(T *p, U cmp, V v) { T o = *p; if (o == (T)cmp) *p = (T)v; return o; }
#define __hip_atomic_load(ptr, order, scope) (*(ptr))
// HIPEMU_STORE_BYTES=<file>: bytes stored with device-scope stores (the interpreter's set words, tables, results), by the tag
// the storing lane set last (MSI_EMU_TAG: the command of a list it is executing; 0: none) — written when the process ends
namespace hipemu {
struct StoreBytes {
  unsigned long long by_tag[64] = {};
  const char *path = getenv("HIPEMU_STORE_BYTES");
  ~StoreBytes() {
    if (!path) return;
    FILE *f = fopen(path, "w");
    if (!f) return;
    for (int t = 0; t < 64; ++t)
      if (by_tag[t]) fprintf(f, "tag %2d %s bytes %llu\n", t & 31, t >= 32 ? "compact" : "full   ", by_tag[t]);
    fclose(f);
  }
};
inline StoreBytes store_bytes;
inline void count_store(size_t n) {
  if (store_bytes.path && g.in_kernel) store_bytes.by_tag[g.lanes[g.cur].tag & 63] += n;
}
}  // namespace hipemu
#define MSI_EMU_TAG(x) (hipemu::g.lanes[hipemu::g.cur].tag = (int)(x))
#define __hip_atomic_store(ptr, v, order, scope) ((void)(hipemu::count_store(sizeof(*(ptr))), *(ptr) = (v)))
#define __hip_atomic_fetch_add(ptr, v, order, scope) atomicAdd((ptr), (v))
#define __hip_atomic_fetch_or(ptr, v, order, scope) atomicOr((ptr), (v))
#define __hip_atomic_fetch_and(ptr, v, order, scope) atomicAnd((ptr), (v))
#define __hip_atomic_fetch_max(ptr, v, order, scope) atomicMax((ptr), (v))
#define __hip_atomic_fetch_min(ptr, v, order, scope) atomicMin((ptr), (v))
#define __hip_atomic_exchange(ptr, v, order, scope) atomicExch((ptr), (v))
