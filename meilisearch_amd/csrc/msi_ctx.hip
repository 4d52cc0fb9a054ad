// msi_ctx.hip — context (device + stream), error string, host-side scoring
// arithmetic entry points of include/msi.h.
#include <float.h>
#include <math.h>
#include <stdarg.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <atomic>
#include <numeric>
#include <vector>

#include <execinfo.h>
#include <fcntl.h>
#include <signal.h>
#include <unistd.h>


#include "msi_common.h"
#include "msi_vm.h"

static thread_local char g_err[512] = "";

void msi_set_error(const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

namespace {
// MSI_DEBUG_ABORT=1 (diagnostics): a native backtrace on stderr when the process aborts or faults
void msi_abort_backtrace(int sig) {
  void *frames[64];
  const int n = backtrace(frames, 64);
  const char msg[] = "[msi] fatal signal, native backtrace:\n";
  int fd = 2;   // MSI_DEBUG_ABORT=<path>: into that file (a test runner may have redirected fd 2)
  const char *path = getenv("MSI_DEBUG_ABORT");
  if (path && path[0] == '/') {
    const int f = open(path, O_WRONLY | O_CREAT | O_APPEND, 0644);
    if (f >= 0) fd = f;
  }
  (void)!write(fd, msg, sizeof(msg) - 1);
  backtrace_symbols_fd(frames, n, fd);
  signal(sig, SIG_DFL);
  raise(sig);
}
}  // namespace
// re-armed by the entry points that ask for it (another library may have taken the signal since)
extern "C" void msi_debug_arm_abort_backtrace(void) {
  static const bool on = getenv("MSI_DEBUG_ABORT") != nullptr;
  if (!on) return;
  signal(SIGABRT, msi_abort_backtrace);
  signal(SIGSEGV, msi_abort_backtrace);
  signal(SIGBUS, msi_abort_backtrace);
}
namespace {
struct AbortHook {
  AbortHook() {
    if (getenv("MSI_DEBUG_ABORT")) {
      signal(SIGABRT, msi_abort_backtrace);
      signal(SIGSEGV, msi_abort_backtrace);
    }
  }
} g_abort_hook;
}  // namespace

// The command-list rounds of the keyword searches run on 16 streams; the HIP runtime maps streams onto GPU_MAX_HW_QUEUES
// hardware queues (default 4) and rounds that share a queue serialise: 4 -> 16 queues took the keyword leg from 2.8 k to
// 6.1 k searches/s.  The runtime reads the variable when it STARTS (its first API call), so the library sets
// it when it is LOADED — before any HIP call of a process whose only HIP user it is (the Rust server), and before torch's
// lazy initialisation in the test / bench processes.  A value the host already chose is left alone; MSI_KEEP_HW_QUEUES=1
// keeps the library from touching the environment at all.  (not an environment note for the integrator.)
// (setenv is not thread-safe against a concurrent getenv: a host that dlopen()s the library from a multi-threaded process
// sets the variable itself and MSI_KEEP_HW_QUEUES=1 — include/msi.h at msi_runtime_hw_queues)
static int g_hw_queues_source = 0;   // 1: the constructor below set the variable
__attribute__((constructor(101))) static void msi_preset_hw_queues() {
  if (!getenv("MSI_KEEP_HW_QUEUES") && !getenv("GPU_MAX_HW_QUEUES")) {
    setenv("GPU_MAX_HW_QUEUES", "16", 0);
    g_hw_queues_source = 1;
  }
}

extern "C" {

int32_t msi_runtime_hw_queues(void) {
  const char *hq = getenv("GPU_MAX_HW_QUEUES");
  const int v = hq ? atoi(hq) : 4;   // the runtime's default
  return v > 0 ? v : 4;
}

int32_t msi_runtime_hw_queues_source(void) { return g_hw_queues_source; }

int32_t msi_abi_version(void) { return MSI_ABI_VERSION; }
const char *msi_last_error(void) { return g_err; }

int32_t msi_ctx_create(int32_t device, msi_ctx **out) {
  if (!out) {
    msi_set_error("msi_ctx_create: out is NULL");
    return MSI_E_INVALID;
  }
  *out = nullptr;
  int count = 0;
  hipError_t e = hipGetDeviceCount(&count);
  if (e != hipSuccess || count <= 0) {
    msi_set_error("no HIP device available (%s); libmsi has no CPU fallback",
                  e == hipSuccess ? "device count is 0" : hipGetErrorString(e));
    return MSI_E_NO_DEVICE;
  }
  if (device < 0) {
    const char *lr = getenv("LOCAL_RANK");
    device = lr ? atoi(lr) % count : 0;
  }
  if (device >= count) {
    msi_set_error("device %d out of range (count %d)", device, count);
    return MSI_E_NO_DEVICE;
  }
  hipDeviceProp_t prop;
  MSI_HIP_TRY(hipGetDeviceProperties(&prop, device));
  if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
    msi_set_error("device %d is %s; libmsi is built for gfx950 (MI355X) only", device,
                  prop.gcnArchName);
    return MSI_E_NO_DEVICE;
  }
  DeviceGuard g(device);
  // (hardware queues: msi_preset_hw_queues() below ran when the library was loaded; msi_runtime_hw_queues() reports what is
  // in effect.  MSI_VERBOSE=1 says so on stderr once — a library does not write to its host's stderr unasked: ADVICE r4)
  {
    static std::atomic<bool> said{false};
    if (getenv("MSI_VERBOSE") && msi_runtime_hw_queues() < 16 && !said.exchange(true))
      fprintf(stderr, "libmsi: GPU_MAX_HW_QUEUES is %d: keyword searches in flight share that many hardware queues and their "
              "command-list rounds serialise (measured: less than half the throughput of 16)\n", msi_runtime_hw_queues());
  }
  msi_ctx *c = new msi_ctx();
  c->device = device;
  c->n_cu = prop.multiProcessorCount;
  // The context's main stream carries the vector scan: 256 persistent workgroups that each want a whole CU's LDS; the
  // command-list kernels of the keyword searches run on other streams as thousands of small workgroups.  Giving the
  // scan's stream the highest dispatch priority (MSI_SCAN_STREAM_PRIORITY=1) was measured and is OFF by default: the scan
  // gained 3 % in the overlapped step and the keyword rounds lost more (hybrid step 157 -> 183 ms, r3_bench_variants.txt).
  {
    int least = 0, greatest = 0;
    const char *knob = getenv("MSI_SCAN_STREAM_PRIORITY");
    e = hipErrorNotSupported;
    if (knob && knob[0] == '1' && hipDeviceGetStreamPriorityRange(&least, &greatest) == hipSuccess && greatest != least)
      e = hipStreamCreateWithPriority(&c->stream, hipStreamNonBlocking, greatest);
    // MSI_SCAN_CUS=<n>: the scan's stream may only use n of the device's CUs (hipExtStreamCreateWithCUMask) and the scan
    // sizes its persistent grid for them: the rest stays free for the keyword searches' command lists while a sweep is
    // in flight (a sweep otherwise holds every CU's LDS for 5 ms and a keyword round waits behind it).
    if (const char *cus = getenv("MSI_SCAN_CUS")) {
      const int n = atoi(cus);
      if (e != hipSuccess && n >= 8 && n < c->n_cu) {
        // enabled CUs spread evenly over the mask (whatever the bit -> XCD mapping is, every XCD keeps some free CUs)
        std::vector<uint32_t> mask((size_t)(c->n_cu + 31) / 32, 0);
        int on = 0;
        for (int i = 0; i < c->n_cu; ++i)
          if ((int64_t)(i + 1) * n / c->n_cu > (int64_t)i * n / c->n_cu) { mask[i / 32] |= 1u << (i % 32); ++on; }
        if (hipExtStreamCreateWithCUMask(&c->stream, (uint32_t)mask.size(), mask.data()) == hipSuccess) {
          e = hipSuccess;
          c->n_cu_scan = on;
        }
      }
    }
    if (e != hipSuccess) e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
  }
  if (e == hipSuccess) e = hipStreamCreateWithFlags(&c->stream_aux, hipStreamNonBlocking);
  if (e != hipSuccess) {
    msi_set_error("hipStreamCreate failed: %s", hipGetErrorString(e));
    delete c;
    return MSI_E_HIP;
  }
  *out = c;
  return MSI_OK;
}

void msi_ctx_destroy(msi_ctx *ctx) { msi_ctx_release(ctx); }

}  // extern "C"

void msi_ctx_retain(msi_ctx *ctx) { ctx->refs.fetch_add(1, std::memory_order_relaxed); }

void msi_ctx_release(msi_ctx *ctx) {
  if (!ctx) return;
  if (ctx->refs.fetch_sub(1, std::memory_order_acq_rel) != 1) return;
  if (ctx->vm) msi_vm_destroy(ctx->vm);   // joins the combiner thread; every pool is gone by now
  ctx->vm = nullptr;
  DeviceGuard g(ctx->device);
  if (ctx->stream) {
    (void)hipStreamSynchronize(ctx->stream);
    (void)hipStreamDestroy(ctx->stream);
  }
  if (ctx->stream_aux) {
    (void)hipStreamSynchronize(ctx->stream_aux);
    (void)hipStreamDestroy(ctx->stream_aux);
  }
  delete ctx;
}

extern "C" {

void *msi_ctx_stream(msi_ctx *ctx) { return ctx ? (void *)ctx->stream : nullptr; }
int32_t msi_ctx_device(msi_ctx *ctx) { return ctx ? ctx->device : -1; }

int32_t msi_ctx_set_profiling(msi_ctx *ctx, int32_t enable) {
  if (!ctx) return MSI_E_INVALID;
  std::lock_guard<std::mutex> lk(ctx->mu);
  std::lock_guard<std::mutex> lk2(ctx->mu_aux);
  ctx->profiling = enable != 0;
  return MSI_OK;
}

int32_t msi_ctx_synchronize(msi_ctx *ctx) {
  if (!ctx) return MSI_E_INVALID;
  DeviceGuard g(ctx->device);
  MSI_HIP_TRY(hipStreamSynchronize(ctx->stream));
  MSI_HIP_TRY(hipStreamSynchronize(ctx->stream_aux));
  return MSI_OK;
}

// ---- host-side list merge ------------------------------------------------------

// The tail of VectorStore::nns_by_vector (crates/milli/src/vector/store.rs:1059,1090):
// the per-store result lists are concatenated and sorted by distance.  Also the
// final step of a row-sharded search (one list per GPU).  Ties: ascending docid
// (search/new/tests/cutoff.rs:507-626), which makes the result independent of the
// number of shards.  No device work.
uint32_t msi_merge_topk(const uint32_t *docids, const float *dist, const uint32_t *counts,
                        uint32_t n_lists, uint32_t list_stride, uint32_t k_out,
                        uint32_t *out_docids, float *out_dist) {
  if (!docids || !dist || !counts || (k_out && (!out_docids || !out_dist))) return 0;
  std::vector<uint64_t> pos;
  for (uint32_t l = 0; l < n_lists; ++l) {
    const uint32_t c = std::min(counts[l], list_stride);
    for (uint32_t i = 0; i < c; ++i) pos.push_back((uint64_t)l * list_stride + i);
  }
  auto less = [&](uint64_t a, uint64_t b) {
    const uint32_t oa = f32_to_ord(dist[a]), ob = f32_to_ord(dist[b]);
    if (oa != ob) return oa < ob;
    return docids[a] < docids[b];
  };
  const size_t n = std::min<size_t>(pos.size(), k_out);
  std::partial_sort(pos.begin(), pos.begin() + n, pos.end(), less);
  for (size_t i = 0; i < n; ++i) {
    out_docids[i] = docids[pos[i]];
    out_dist[i] = dist[pos[i]];
  }
  return (uint32_t)n;
}

// ---- scoring arithmetic (host; these run on the Rust caller's thread) -------

// DistributionShift::shift — crates/milli/src/vector/distribution.rs:103-130.
float msi_distribution_shift(float mean, float sigma, float score) {
  const float target_mean = 0.5f, target_sigma = 0.4f;
  volatile float factor = target_sigma / sigma;
  volatile float fm = factor * mean;  // volatile: keep mul and add unfused
  float offset = target_mean - fm;
  volatile float fs = factor * score;
  float s = fs + offset;
  if (s <= 0.0f) s = FLT_EPSILON;
  if (s > 1.0f) s = 1.0f;
  return s;
}

// Rank::global_score — crates/milli/src/score_details.rs:517-546.
double msi_rank_global_score(const uint32_t *ranks, const uint32_t *max_ranks, uint32_t n) {
  uint32_t rank = 1, max_rank = 1;
  for (uint32_t i = 0; i < n; ++i) {
    rank = rank ? rank - 1 : 0;  // saturating_sub(1)
    rank *= max_ranks[i];
    max_rank *= max_ranks[i];
    rank += ranks[i];
  }
  return (double)rank / (double)max_rank;
}

// ScoreDetails::global_score of the details msi_keyword_search_ranked returns for one hit.
double msi_score_details_global_score(const msi_score_detail *details, uint32_t n) {
  uint32_t rank = 1, max_rank = 1;
  for (uint32_t i = 0; i < n; ++i) {
    uint32_t r, m;
    // ScoreDetails::rank() is None for Sort and GeoSort, score_details.rs:103-121
    // ... and for Pin ("a placement directive, not a score", score_details.rs:123,135)
    if (details[i].kind == MSI_SCORE_SORT || details[i].kind == MSI_SCORE_GEO_SORT || details[i].kind == MSI_SCORE_PIN) continue;
    switch (details[i].kind) {
      case MSI_SCORE_TYPO:
        m = details[i].b + 1;
        r = m > details[i].a ? m - details[i].a : 0;
        break;
      case MSI_SCORE_EXACT_WORDS:
        r = details[i].a + 1;
        m = details[i].b + 1;
        break;
      case MSI_SCORE_SKIPPED:  // score_details.rs:118
        r = 0;
        m = 1;
        break;
      default:
        r = details[i].a;
        m = details[i].b;
        break;
    }
    rank = rank ? rank - 1 : 0;  // Rank::merge, score_details.rs:536-546
    rank *= m;
    max_rank *= m;
    rank += r;
  }
  return (double)rank / (double)max_rank;
}

// inject_pins (search/new/bucket_sort.rs:345-377) over merge_positioned_hits_into_page (search/mod.rs:579-625): the pins —
// in resolve_pins' order — are merged into the organic prefix [0, from + length) the bucket sort produced, a pin taking
// the combined index its position names (or the next one when earlier pins pushed it on; or the end of the organic hits
// when they run out: "pumping"), and the page [from, from + length) is cut out of the combined list.
uint32_t msi_inject_pins(const msi_pin *pins, uint32_t n_pins, uint32_t from, uint32_t length, const uint32_t *docids,
                         const msi_score_detail *scores, const uint32_t *n_scores, uint32_t n, uint32_t *out_docids,
                         msi_score_detail *out_scores, uint32_t *out_n_scores) {
  if ((n_pins && !pins) || (n && !docids) || !out_docids) return 0;
  const uint64_t page_end = (uint64_t)from + (uint64_t)length;
  uint32_t written = 0, pi = 0, oi = 0;
  auto put_organic = [&](uint32_t i, bool keep) {
    if (!keep) return;
    out_docids[written] = docids[i];
    if (out_scores && out_n_scores) {
      const uint32_t ns = scores && n_scores ? n_scores[i] : 0;
      if (ns) memcpy(out_scores + (size_t)written * MSI_MAX_SCORE_DETAILS, scores + (size_t)i * MSI_MAX_SCORE_DETAILS, ns * sizeof(msi_score_detail));
      out_n_scores[written] = ns;
    }
    ++written;
  };
  auto put_pin = [&](const msi_pin &p, bool keep) {
    if (!keep) return;
    out_docids[written] = p.docid;
    if (out_scores && out_n_scores) {   // vec![ScoreDetails::Pin { position }]
      out_scores[(size_t)written * MSI_MAX_SCORE_DETAILS] = msi_score_detail{MSI_SCORE_PIN, p.position, 0};
      out_n_scores[written] = 1;
    }
    ++written;
  };
  if (n_pins == 0) {   // (the reference returns the organic hits as they are: the caller asked the bucket sort for the page itself)
    for (uint32_t i = 0; i < n && written < length; ++i) put_organic(i, true);
    return written;
  }
  for (uint64_t combined = 0; combined < page_end; ++combined) {
    const bool keep = combined >= from;
    if (pi < n_pins) {
      if ((uint64_t)pins[pi].position <= combined) put_pin(pins[pi++], keep);
      else if (oi < n) put_organic(oi++, keep);
      else put_pin(pins[pi++], keep);
    } else if (oi < n) {
      put_organic(oi++, keep);
    } else {
      break;
    }
  }
  return written;
}

// compare_scores over ScoreValue::Score sequences — search/hybrid.rs:32-80.
int32_t msi_compare_scores(const double *left, uint32_t n_left, float left_ratio,
                           const double *right, uint32_t n_right, float right_ratio) {
  for (uint32_t i = 0;; ++i) {
    bool hl = i < n_left, hr = i < n_right;
    if (!hl && !hr) return 0;
    if (!hl) return -1;
    if (!hr) return 1;
    double a = left[i] * (double)left_ratio;
    double b = right[i] * (double)right_ratio;
    if (fabs(a - b) <= DBL_EPSILON) continue;
    return a < b ? -1 : 1;
  }
}

}  // extern "C"
