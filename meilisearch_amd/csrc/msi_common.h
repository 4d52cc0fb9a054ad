// msi_common.h — internal helpers shared by the libmsi translation units.
// gfx950 (MI355X / CDNA4) only: 64-wide wavefronts are assumed everywhere.
#pragma once
#include "msi_arena.h"
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <atomic>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/msi.h"

#define MSI_WAVE 64

// The dynamic LDS of a kernel (the size is the launch's).  tests/emu pre-defines it for the CPU emulation of the HIP
// runtime, where LDS is a host buffer.
// Ordering without cache maintenance.  On gfx950 an agent-scope fence (__threadfence) is `buffer_wbl2 sc1` +
// `buffer_inv sc1`: it writes back AND invalidates the whole L2 of the XCD the wave runs on — for every kernel resident
// there.  A workgroup that only has to order its own device-scope ATOMICS (which are performed at the device's
// coherence point, not in the L2) before a later atomic needs no more than "my earlier memory operations are
// acknowledged": s_waitcnt vmcnt(0).  The instruction is emitted EXPLICITLY (on gfx9 loads, stores and atomics all count
// in vmcnt; there is no separate store counter): a workgroup-scope fence alone promises no inter-workgroup ordering under
// the memory model and the backend is free to leave the wait out of it — the fence stays for what it does promise, that
// the compiler moves no memory operation across this point.  `make` checks the built ISA for the wait (check_isa.py).
#ifndef MSI_ORDER_ATOMICS
#define MSI_ORDER_ATOMICS()                                      \
  do {                                                           \
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");       \
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");             \
  } while (0)
#endif
// A value every lane of the wave holds alike, moved to a scalar register (uniform branches, scalar address arithmetic).
#ifndef MSI_UNIFORM
#define MSI_UNIFORM(x) ((uint32_t)__builtin_amdgcn_readfirstlane((int)(x)))   // (the builtin is int -> int: unsigned again, or a
                                                                            // 64-bit value pieced together from two of these is sign-extended)
#endif
// What a workgroup WROTE with plain stores must be in memory before another kernel — possibly on another XCD, possibly
// started before this kernel ends — is told about it: write this XCD's L2 back (buffer_wbl2 sc1), but do not invalidate
// it (the acquire half of __threadfence, buffer_inv sc1, is what evicts every resident kernel's cached lines).
#ifndef MSI_RELEASE_DEVICE
#define MSI_RELEASE_DEVICE() __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent")
#endif
// The acquire half on its own (buffer_inv sc1): lines of other XCDs' write-through stores this XCD may still hold are dropped.
#ifndef MSI_ACQUIRE_DEVICE
#define MSI_ACQUIRE_DEVICE() __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent")
#endif
// The CPU emulation of the test tier attributes the device-scope stores of a kernel to what the storing lane is doing (the
// command of a list: tests/emu counts stored bytes per tag when HIPEMU_STORE_BYTES names a file).  Nothing on the device.
#ifndef MSI_EMU_TAG
#define MSI_EMU_TAG(x) ((void)0)
#endif
#ifndef MSI_SLEEP
#define MSI_SLEEP() __builtin_amdgcn_s_sleep(8)
#endif
#ifndef MSI_DYNAMIC_LDS
#define MSI_DYNAMIC_LDS(name) extern __shared__ __attribute__((aligned(16))) unsigned char name[]
#endif

void msi_set_error(const char *fmt, ...);

#define MSI_HIP_TRY(expr)                                                          \
  do {                                                                             \
    hipError_t _e = (expr);                                                        \
    if (_e != hipSuccess) {                                                        \
      msi_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, \
                    __LINE__);                                                     \
      return _e == hipErrorOutOfMemory ? MSI_E_OOM : MSI_E_HIP;                    \
    }                                                                              \
  } while (0)

#define MSI_TRY(expr)              \
  do {                             \
    int32_t _s = (expr);           \
    if (_s != MSI_OK) return _s;   \
  } while (0)

struct msi_ctx {
  int device = 0;
  int n_cu = 0;
  int n_cu_scan = 0;   // CUs the main stream may use when it was created with a CU mask (MSI_SCAN_CUS); 0: all
  hipStream_t stream = nullptr;      // vector stores, docid sets, ranking
  std::mutex mu;                     // serialises use of `stream` + per-object scratch
  hipStream_t stream_aux = nullptr;  // dictionaries: the VALU-bound typo lookup overlaps the
  std::mutex mu_aux;                 // HBM-bound scan when both are enqueued (separate HIP streams)
  bool profiling = false;
  // The caller's handle holds one reference, every object created on the
  // context (msi_vs / msi_dict / msi_bits) one more: msi_ctx_destroy only drops
  // the caller's, so objects may be destroyed after their context in any order.
  std::atomic<int> refs{1};
  // the command-list combiner of the ranked keyword searches (msi_vm.hip), created on first use
  struct msi_vm *vm = nullptr;
  std::mutex vm_mu;
};
void msi_ctx_retain(msi_ctx *ctx);
void msi_ctx_release(msi_ctx *ctx);

// HIP-event bracket around one kernel (only when msi_ctx::profiling).
struct KernelTimer {
  std::vector<std::pair<hipEvent_t, hipEvent_t>> pending, free_;
  hipEvent_t cur_start = nullptr, cur_stop = nullptr;
  hipStream_t st_ = nullptr;
  void begin(msi_ctx *ctx, hipStream_t st = nullptr) {
    if (!ctx->profiling) return;
    st_ = st ? st : ctx->stream;
    if (free_.empty()) {
      hipEvent_t a, b;
      if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) return;
      free_.push_back({a, b});
    }
    cur_start = free_.back().first;
    cur_stop = free_.back().second;
    free_.pop_back();
    (void)hipEventRecord(cur_start, st_);
  }
  void end(msi_ctx *ctx) {
    if (!cur_start) return;
    (void)hipEventRecord(cur_stop, st_);
    pending.push_back({cur_start, cur_stop});
    cur_start = cur_stop = nullptr;
  }
  // stream must be synchronised by the caller
  void drain(uint64_t *launches, double *ms) {
    *launches = 0;
    *ms = 0.0;
    for (auto &p : pending) {
      float t = 0.f;
      if (hipEventElapsedTime(&t, p.first, p.second) == hipSuccess) {
        *ms += t;
        ++*launches;
      }
      free_.push_back(p);
    }
    pending.clear();
  }
  void release() {
    for (auto &p : pending) { (void)hipEventDestroy(p.first); (void)hipEventDestroy(p.second); }
    for (auto &p : free_) { (void)hipEventDestroy(p.first); (void)hipEventDestroy(p.second); }
    pending.clear();
    free_.clear();
  }
};

// Growable device buffer (never shrinks); not thread-safe by itself (callers
// hold msi_ctx::mu).
struct DevBuf {
  void *p = nullptr;
  size_t cap = 0;
  int32_t ensure(size_t bytes) {
    if (bytes <= cap) return MSI_OK;
    if (p) (void)hipFree(p);
    p = nullptr;
    cap = 0;
    size_t want = bytes < 256 ? 256 : bytes;
    hipError_t e = hipMalloc(&p, want);
    if (e != hipSuccess) {
      msi_set_error("hipMalloc(%zu) failed: %s", want, hipGetErrorString(e));
      p = nullptr;
      return MSI_E_OOM;
    }
    cap = want;
    return MSI_OK;
  }
  void release() {
    if (p) (void)hipFree(p);
    p = nullptr;
    cap = 0;
  }
  template <typename T>
  T *as() const {
    return reinterpret_cast<T *>(p);
  }
};

struct DeviceGuard {
  int prev = -1;
  explicit DeviceGuard(int dev) {
    (void)hipGetDevice(&prev);
    if (prev != dev) (void)hipSetDevice(dev);
    else prev = -1;
  }
  ~DeviceGuard() {
    if (prev >= 0) (void)hipSetDevice(prev);
  }
};

// ---- batched CboRoaringBitmap decode (msi_bits.hip; used by msi_keyword.hip) ----------
struct MsiContainer {
  uint32_t key;     // high 16 bits of the docids
  uint32_t type;    // 0 array, 1 bitmap, 2 run
  uint32_t card;    // array: #values, run: #runs
  uint32_t offset;  // byte offset of the body inside the staged buffer
};
constexpr uint64_t MSI_NO_CACHE = ~0ull;
// (a batch lives inside one call; inside msi_keyword_search_ranked its vectors come from the search's arena, msi_arena.h)
struct MsiCboBatch {
  msi_arena::Vec<uint8_t> bytes;            // concatenated Roaring serialisations (each starts 16-byte aligned)
  msi_arena::Vec<MsiContainer> containers;  // their containers, offsets into `bytes`
  msi_arena::Vec<uint32_t> small_ids;       // documents of the <= 7-integer raw values
  // HBM posting cache (msi_vm.h), per container, empty when the batch never touched the cache:
  //   src[i]  != MSI_NO_CACHE: the body is read from the cache at this byte offset (its bytes are NOT in `bytes`);
  //   fill[i] != MSI_NO_CACHE: the workgroup that decodes the body also stores it into the cache at this offset.
  msi_arena::Vec<uint64_t> src, fill;
  msi_arena::Vec<void *> fill_tokens;       // the cache entries this batch fills (msi_pcache_commit once its decode has run)
};
struct msi_bits;
// cache_src / cache_fill: byte offset of the WHOLE serialisation inside the posting cache (MSI_NO_CACHE: none).
bool msi_cbo_batch_append(MsiCboBatch &batch, const uint8_t *bytes, size_t len, uint64_t cache_src = MSI_NO_CACHE,
                          uint64_t cache_fill = MSI_NO_CACHE);
uint64_t msi_cbo_cardinality(const uint8_t *bytes, size_t len);
int32_t msi_bits_decode_batch(msi_bits *p, uint32_t slot, const MsiCboBatch &batch, bool clear);
extern "C" int32_t msi_bits_or_docids_device(msi_bits *p, uint32_t slot, const uint32_t *d_docids, uint64_t n);
msi_ctx *msi_bits_ctx(msi_bits *p);
// Fused set steps of the path search (msi_search.hip), one launch each:
//   and_many:  dst[i] = prefix & cond[i], counts[i] = |dst[i]|   (n <= MSI_BITS_MANY; one completion signal)
//   claim:     bucket |= docs; universe &= ~docs; stack[i] &= ~docs   (docs may be one of the stack slots)
constexpr uint32_t MSI_BITS_MANY = 16;
int32_t msi_bits_and_many_count(msi_bits *p, uint32_t prefix, uint32_t n, const uint32_t *cond, const uint32_t *dst,
                                uint64_t *counts);
int32_t msi_bits_claim(msi_bits *p, uint32_t docs, uint32_t bucket, uint32_t universe, uint32_t n_stack,
                       const uint32_t *stack);
//   paths_claim: for path k = 0..n-1 in order: docs = universe & AND(steps of k); bucket |= docs; universe &= ~docs;
//                counts[k] = |docs|  — a whole cost level of a rule graph in one launch and one completion signal
constexpr uint32_t MSI_BITS_MAX_PATHS = 256, MSI_BITS_MAX_STEPS = 4096;
//   clear_slots: zero up to MSI_BITS_CLEAR_MAX slots in one launch (the engine hands out pre-zeroed slots)
constexpr uint32_t MSI_BITS_CLEAR_MAX = 64;
int32_t msi_bits_clear_slots(msi_bits *p, uint32_t n, const uint32_t *slots);
int32_t msi_bits_paths_claim(msi_bits *p, uint32_t n_paths, const uint32_t *path_off, const uint32_t *step_slots,
                             uint32_t bucket, uint32_t universe, uint64_t *counts);
//   paths_enqueue / paths_collect: several cost levels behind ONE completion wait.  Each enqueued level (<= 64 paths
//   over <= 32 distinct conditions) publishes its counts into its own region (0..MSI_BITS_PATH_REGIONS-1); the
//   levels run in stream order on the same universe, so level k+1 sees what level k left.  paths_collect waits for
//   the last enqueued level and returns counts[region][64].  paths_enqueue returns MSI_E_UNSUPPORTED when the level
//   does not fit the in-argument kernel (the caller then uses msi_bits_paths_claim).
constexpr uint32_t MSI_BITS_PATH_REGIONS = 4, MSI_BITS_REGION_PATHS = 64;
int32_t msi_bits_paths_enqueue(msi_bits *p, uint32_t n_paths, const uint32_t *path_off, const uint32_t *step_slots,
                               uint32_t bucket, uint32_t universe, uint32_t region);
int32_t msi_bits_paths_collect(msi_bits *p, uint32_t n_regions, uint64_t *counts);

// ---- device helpers -------------------------------------------------------

// Monotone map f32 -> u32 (a < b  <=>  ord(a) < ord(b), -0 < +0, NaNs at the ends).
__host__ __device__ inline uint32_t f32_to_ord(float f) {
  uint32_t b;
#if defined(__HIP_DEVICE_COMPILE__)
  b = __float_as_uint(f);
#else
  memcpy(&b, &f, 4);
#endif
  return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__host__ __device__ inline float ord_to_f32(uint32_t o) {
  uint32_t b = (o & 0x80000000u) ? (o & 0x7FFFFFFFu) : ~o;
#if defined(__HIP_DEVICE_COMPILE__)
  return __uint_as_float(b);
#else
  float f;
  memcpy(&f, &b, 4);
  return f;
#endif
}

// Correctly rounded f32 sqrt / divide for the reference-arithmetic ("canonical")
// paths.  HIP's __fsqrt_rn lowers to the ~1-ulp native v_sqrt_f32 and the rounding
// of `a / b` depends on a compiler flag, so both go through f64: the f64 result of
// one sqrt / divide of f32 operands rounds to the correctly rounded f32 (53 >= 2*24+2
// makes the double rounding innocuous).  Not used on any hot path.
__device__ __forceinline__ float msi_sqrt_rn(float x) { return (float)__builtin_sqrt((double)x); }
__device__ __forceinline__ float msi_div_rn(float a, float b) { return (float)((double)a / (double)b); }

static inline uint32_t ceil_div_u32(uint64_t a, uint64_t b) { return (uint32_t)((a + b - 1) / b); }
