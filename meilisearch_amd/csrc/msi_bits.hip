// msi_bits.hip — S3: dense docid-set algebra in HBM, gfx950.
//
// Device replacement for the RoaringBitmap algebra on milli's ranking path:
// compute_query_term_subset_docids (search/new/resolve_query_graph.rs:33-59),
// the ∩/∪/− of visit_path_condition (graph_based_ranking_rule.rs:383-437) and
// bucket_sort's universe bookkeeping (bucket_sort.rs:23-343).  A pool owns
// n_slots dense sets of n_docs bits; every operation is one HBM-bound pass of
// 16-byte loads/stores (algorithmic bytes = words touched x 8).
//
// Posting lists arrive in milli's on-disk form (CboRoaringBitmapCodec,
// heed_codec/roaring_bitmap/cbo_roaring_bitmap_codec.rs:53-85) and are decoded
// on the device: the host only parses the container headers of the standard
// Roaring serialisation (RoaringFormatSpec: cookies 12346 / 12347).
#include <string.h>

#include <algorithm>
#include <chrono>
#include <vector>

#include "msi_common.h"

typedef unsigned long long u64;

struct msi_bits {
  msi_ctx *ctx = nullptr;
  uint64_t n_docs = 0;
  uint64_t n_words = 0;  // per slot, multiple of 2 (16-byte vector access)
  uint32_t n_slots = 0;
  DevBuf pool, tmp, small, stage, desc, small_ids;
  // chunk summaries of the command-list path (msi_vm.hip): one bit per (65 536-document chunk, slot), chunk-major rows of
  // 16 words; 0 = that chunk of the slot is all zero.  Kept by the command lists; anything else that may write a slot
  // (every direct entry point goes through check_slot / msi_bits_slot_ptr) marks them stale and the next list resets them.
  DevBuf summary;
  bool sum_dirty = false;
  // Completion signalling without a stream synchronisation: the last workgroup of a counting kernel
  // writes {value, sequence number} into fine-grained pinned host memory and the caller polls it.
  volatile uint64_t *h_sig = nullptr;  // [0] value, [1] sequence
  u64 *d_acc = nullptr;                // [0] running sum, [1] workgroups done (self-resetting)
  uint64_t seq = 0;
  // pinned staging ring for posting bytes: decodes are enqueued without waiting for the copy
  uint8_t *h_ring = nullptr;
  size_t ring_cap = 0, ring_pos = 0;
  // The stream and lock every operation of this pool uses: the context's by default (in-stream order with
  // the vector scan and the ranking kernels), a private pair after msi_bits_use_private_stream (one search
  // per pool, many searches in flight).
  hipStream_t stream = nullptr;
  std::mutex own_mu;
  std::mutex *mu = nullptr;
  // One counting operation in flight per pool: the signal cell pair {value, sequence} and the counts regions are per
  // pool, so a caller holds wait_mu from its launch until it has read its result (lock order: wait_mu, then mu).
  std::recursive_mutex wait_mu;  // recursive: a counting entry point may call another one of the same pool
  bool private_stream = false;
  // distinct scratch, per pool (one search per pool): first[v] = the smallest undecided candidate holding value v
  // this round, taken[v] = stamp of the call in which a kept candidate holds v.  Stamps instead of clears: a round
  // writes ((~round_stamp) << 32 | docid) with atomicMin, so a newer round always beats what older rounds left.
  DevBuf dv_first, dv_taken;
  uint32_t dv_cap = 0, dv_round = 0, dv_call = 0;
  // result block of the command-list path (msi_vm.hip): pinned, fine-grained; [0] sequence, [1] first-k count,
  // [2 ..] cardinalities, then the first-k docids
  uint64_t *vm_block = nullptr;
  uint64_t vm_seq = 0;
  uint8_t *vm_stage = nullptr;   // pinned staging of the decode payloads of the list being recorded
  size_t vm_stage_cap = 0;
  // Universe compaction of the ranked keyword search (msi_search.hip, "compact space"): once a search knows its initial
  // universe U0 — typically well under 1 % of the index — every later set is a subset of it and is kept over the RANKS of
  // the documents inside U0 instead of over docids: |U0| bits per set instead of n_docs.  The sets of that space live in
  // a companion pool (created on first use, |U0| <= its n_docs); the tables that map between the spaces — per-chunk
  // cardinalities of U0, exclusive prefix popcounts per word of U0, rank -> docid — are `caux`, written by the VM_RANK
  // commands of a list on THIS pool (msi_vm.hip).
  msi_bits *companion = nullptr;
  DevBuf caux;
  uint64_t caux_cap = 0;
  u64 *slot(uint32_t s) const { return pool.as<u64>() + (uint64_t)s * n_words; }
};

// The facet values of the distinct field per document, CSR in HBM (msi_doc_values_create)
struct msi_doc_values {
  msi_ctx *ctx = nullptr;
  uint64_t n_docs = 0;
  uint32_t n_values = 0;
  DevBuf offsets, values;  // u32 [n_docs + 1], u32 [offsets[n_docs]]
};

// The _geo point of every document, resident in HBM (msi_geo_points_create): [n_docs][2] f64, NaN latitude = none
struct msi_geo_points {
  msi_ctx *ctx = nullptr;
  uint64_t n_docs = 0;
  DevBuf lat_lng;
};

// The facet values of one filterable field and kind (numbers | strings) per document: CSR of u64 sort keys
struct msi_facet_keys {
  msi_ctx *ctx = nullptr;
  uint64_t n_docs = 0;
  DevBuf offsets, keys;  // u32 [n_docs + 1], u64 [offsets[n_docs]]
};

// One u32 order key per document, resident in HBM (msi_doc_keys_create)
struct msi_doc_keys {
  msi_ctx *ctx = nullptr;
  uint64_t n_docs = 0;
  DevBuf keys;
};

// accessors for the other translation units (msi_rank.hip)
msi_ctx *msi_bits_ctx(msi_bits *p) { return p->ctx; }
hipStream_t msi_bits_stream(msi_bits *p) { return p->stream; }
std::mutex &msi_bits_mutex(msi_bits *p) { return *p->mu; }
u64 *msi_bits_slot_ptr(msi_bits *p, uint32_t slot) {   // (other modules' kernels write through this pointer)
  p->sum_dirty = true;
  return p->slot(slot);
}
u64 *msi_bits_pool_base(msi_bits *p) { return p->slot(0); }   // the command lists' own view of the pool
u64 *msi_bits_summary(msi_bits *p) { return p->summary.p ? p->summary.as<u64>() : nullptr; }
bool msi_bits_take_summary_dirty(msi_bits *p) {
  const bool d = p->sum_dirty;
  p->sum_dirty = false;
  return d;
}
void msi_bits_mark_summary_dirty(msi_bits *p) { p->sum_dirty = true; }   // (a list that had taken the flag was dropped)
uint64_t msi_bits_words_per_slot(msi_bits *p) { return p->n_words; }
uint32_t msi_bits_n_slots(msi_bits *p) { return p->n_slots; }
uint64_t msi_bits_n_docs(msi_bits *p) { return p->n_docs; }
uint64_t *msi_bits_vm_block(msi_bits *p) {
  if (!p->vm_block) {
    DeviceGuard g(p->ctx->device);
    void *h = nullptr;
    const size_t bytes = (2 + 1024) * sizeof(uint64_t) + 8192 * sizeof(uint32_t);
    if (hipHostMalloc(&h, bytes, hipHostMallocCoherent) != hipSuccess) return nullptr;
    memset(h, 0, bytes);
    p->vm_block = (uint64_t *)h;
  }
  return p->vm_block;
}
uint64_t msi_bits_vm_next_seq(msi_bits *p) { return ++p->vm_seq; }
// The search thread that owns the pool records into it and waits for its list before recording again, so the buffer is
// never in use by a kernel when it grows.
uint8_t *msi_bits_vm_stage(msi_bits *p, size_t need, size_t keep) {
  if (need <= p->vm_stage_cap) return p->vm_stage;
  DeviceGuard g(p->ctx->device);
  const size_t cap = std::max<size_t>(need * 2, (size_t)4 << 20);
  void *h = nullptr;
  if (hipHostMalloc(&h, cap, hipHostMallocDefault) != hipSuccess) {
    msi_set_error("hipHostMalloc(%zu) for the posting staging buffer failed", cap);
    return nullptr;
  }
  if (keep && p->vm_stage) memcpy(h, p->vm_stage, keep);
  if (p->vm_stage) (void)hipHostFree(p->vm_stage);
  p->vm_stage = (uint8_t *)h;
  p->vm_stage_cap = cap;
  return p->vm_stage;
}
// The companion pool of the compact space and the capacity (documents of U0) it was made for: an eighth of the index
// (beyond that the set words saved no longer pay for the mapping), the whole index for pools of at most one chunk (tests).
uint64_t msi_bits_compact_capacity(const msi_bits *p) {
  if (p->n_docs <= 65536) return p->n_docs;
  return std::max<uint64_t>(65536, (p->n_docs / 8 + 127) & ~127ull);
}
msi_bits *msi_bits_compact_pool(msi_bits *p) {
  if (!p->companion) {
    msi_bits *c = nullptr;
    // MSI_BITS_COMPANION_SLOTS_X=2: the companion holds twice the slots (a set of the compact space is an eighth of a set of
    // the full one; a search's sibling sub-trees are admitted as cooperative tasks by the slots that are free, and more tasks
    // are fewer rounds: 10.79 -> 10.16 lists per fresh query at 10 M documents).  Measured at the same throughput (19.7 k q/s
    // either way, profiles/r6_callers_and_slots.log) for 80 MB more per caller: off by default.
    static const uint32_t mult = getenv("MSI_BITS_COMPANION_SLOTS_X") ? (uint32_t)std::max(1, atoi(getenv("MSI_BITS_COMPANION_SLOTS_X"))) : 1u;
    const uint32_t c_slots = std::min<uint32_t>(1024u, std::max<uint32_t>(p->n_slots, p->n_slots * mult));
    if (msi_bits_create(p->ctx, std::max<uint64_t>(1, msi_bits_compact_capacity(p)), c_slots, &c) != MSI_OK) return nullptr;
    // its creation memsets ran on the context's stream; the command lists run on the combiner's
    {
      std::lock_guard<std::mutex> lk(p->ctx->mu);
      DeviceGuard g(p->ctx->device);
      if (hipStreamSynchronize(p->ctx->stream) != hipSuccess) {
        msi_bits_destroy(c);
        return nullptr;
      }
    }
    p->companion = c;
  }
  return p->companion;
}
// [chunk cardinalities: n_chunks u32, padded to 4][prefix: n_words u32, padded to 4][rank -> docid: capacity u32]
uint32_t *msi_bits_compact_aux(msi_bits *p) {
  const uint64_t n_chunks = (p->n_words + 1023) / 1024;
  const uint64_t words = ((n_chunks + 3) & ~3ull) + ((p->n_words + 3) & ~3ull) + ((msi_bits_compact_capacity(p) + 3) & ~3ull);
  if (!p->caux.p) {
    DeviceGuard g(p->ctx->device);
    if (p->caux.ensure((size_t)words * sizeof(uint32_t)) != MSI_OK) return nullptr;
  }
  return p->caux.as<uint32_t>();
}
// waits for everything enqueued on the pool's own stream (the command-list path runs on the combiner's stream)
int32_t msi_bits_sync(msi_bits *p) {
  std::lock_guard<std::mutex> lk(*p->mu);
  DeviceGuard g(p->ctx->device);
  MSI_HIP_TRY(hipStreamSynchronize(p->stream));
  return MSI_OK;
}
const uint32_t *msi_doc_keys_device(const msi_doc_keys *k) { return k->keys.as<uint32_t>(); }

namespace {

constexpr int BT = 256;

typedef MsiContainer Container;

__global__ void bits_fill_kernel(u64 *__restrict__ dst, uint64_t n_words, uint64_t n_docs, int ones) {
  uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_words) return;
  u64 v = 0;
  if (ones) {
    uint64_t lo = i * 64;
    if (lo + 64 <= n_docs) v = ~0ull;
    else if (lo < n_docs) v = (~0ull) >> (64 - (n_docs - lo));
  }
  dst[i] = v;
}

template <int OP>
__global__ void bits_op_kernel(u64 *__restrict__ dst, const u64 *__restrict__ a,
                               const u64 *__restrict__ b, uint64_t n_pairs) {
  uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  for (; i < n_pairs; i += stride) {
    const ulonglong2 x = reinterpret_cast<const ulonglong2 *>(a)[i];
    const ulonglong2 y = reinterpret_cast<const ulonglong2 *>(b)[i];
    ulonglong2 r;
    if (OP == MSI_BITS_AND) { r.x = x.x & y.x; r.y = x.y & y.y; }
    else if (OP == MSI_BITS_OR) { r.x = x.x | y.x; r.y = x.y | y.y; }
    else if (OP == MSI_BITS_ANDNOT) { r.x = x.x & ~y.x; r.y = x.y & ~y.y; }
    else { r.x = x.x ^ y.x; r.y = x.y ^ y.y; }
    reinterpret_cast<ulonglong2 *>(dst)[i] = r;
  }
}

// Block partial -> device accumulator; the last workgroup publishes the total to the host and
// re-arms the accumulator (no memset, no device-to-host copy, no stream synchronisation).
__device__ __forceinline__ void publish_count(uint32_t c, u64 *__restrict__ acc, volatile uint64_t *__restrict__ sig,
                                              uint64_t seq) {
  __shared__ uint32_t part[BT / 64];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) c += __shfl_down(c, o);
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = c;
  __syncthreads();
  if (threadIdx.x == 0) {
    u64 b = 0;
    for (int i = 0; i < BT / 64; ++i) b += part[i];
    if (b) atomicAdd(&acc[0], b);
    MSI_ORDER_ATOMICS();   // atomics only: no L2 write-back / invalidate (msi_common.h)
    const u64 done = atomicAdd(&acc[1], 1ull);
    if (done == gridDim.x - 1) {
      const u64 total = atomicExch(&acc[0], 0ull);
      acc[1] = 0;
      __hip_atomic_store(const_cast<uint64_t *>(&sig[0]), total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      __hip_atomic_store(const_cast<uint64_t *>(&sig[1]), seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
}

// dst = a OP b, *out += |dst|
template <int OP>
__global__ void bits_op_count_kernel(u64 *__restrict__ dst, const u64 *__restrict__ a,
                                     const u64 *__restrict__ b, uint64_t n_pairs, u64 *__restrict__ acc,
                                     volatile uint64_t *__restrict__ sig, uint64_t seq) {
  uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  uint32_t c = 0;
  for (; i < n_pairs; i += stride) {
    const ulonglong2 x = reinterpret_cast<const ulonglong2 *>(a)[i];
    const ulonglong2 y = reinterpret_cast<const ulonglong2 *>(b)[i];
    ulonglong2 r;
    if (OP == MSI_BITS_AND) { r.x = x.x & y.x; r.y = x.y & y.y; }
    else if (OP == MSI_BITS_OR) { r.x = x.x | y.x; r.y = x.y | y.y; }
    else if (OP == MSI_BITS_ANDNOT) { r.x = x.x & ~y.x; r.y = x.y & ~y.y; }
    else { r.x = x.x ^ y.x; r.y = x.y ^ y.y; }
    reinterpret_cast<ulonglong2 *>(dst)[i] = r;
    c += __popcll(r.x) + __popcll(r.y);
  }
  publish_count(c, acc, sig, seq);
}

// ---- order keys (Sort / Asc / Desc ranking rules, crates/milli/src/search/new/sort.rs:95-233) ------------------------
// One u32 key per document: its rank in the rule's iteration order over the facet values of the field (numbers, then
// strings, each in the rule's direction), 0xFFFFFFFF = the document has no value.  The rule's next bucket is "the
// documents of the universe with the smallest key": one pass finds the minimum, one takes its documents out.
// One document per thread, so a wave covers exactly one 64-bit word of a set and __ballot yields the word.
__global__ void bits_min_key_kernel(const u64 *__restrict__ universe, const uint32_t *__restrict__ keys, uint64_t n_docs,
                                    u64 *__restrict__ best /* max over documents of 0xFFFFFFFF - key; 0 = none */) {
  // grid-stride over whole 64-document words (a wave = one word) and ONE atomic per workgroup: with one atomic per
  // wave on a single address the kernel spent 0.25 ms on 2 M documents (r2_rules_kernel_stats_before.csv)
  __shared__ uint32_t part[BT / 64];
  const uint64_t n_span = ((n_docs + 63) / 64) * 64;
  uint32_t inv = 0;
  for (uint64_t d = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; d < n_span; d += (uint64_t)gridDim.x * blockDim.x) {
    const u64 word = universe[d >> 6];   // wave-uniform
    if (!word) continue;
    if (d < n_docs && ((word >> (d & 63)) & 1ull)) inv = max(inv, 0xFFFFFFFFu - keys[d]);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) inv = max(inv, (uint32_t)__shfl_xor((int)inv, o));
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = inv;
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int i = 1; i < BT / 64; ++i) inv = max(inv, part[i]);
    if (inv) atomicMax(best, (u64)inv);
  }
}

// bucket = {d in universe : key[d] == the minimum found}, universe -= bucket; the last workgroup publishes
// {|bucket|, key, seq} and re-arms `best`.  The grid covers every word of the slot, so `bucket` is fully overwritten.
__global__ void bits_take_key_kernel(u64 *__restrict__ universe, u64 *__restrict__ bucket,
                                     const uint32_t *__restrict__ keys, uint64_t n_docs, uint64_t n_words,
                                     u64 *__restrict__ best, u64 *__restrict__ acc, volatile uint64_t *__restrict__ sig,
                                     uint64_t seq) {
  const uint32_t key = 0xFFFFFFFFu - (uint32_t)__hip_atomic_load(best, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __shared__ uint32_t part[BT / 64];
  uint32_t cnt = 0;  // lane 0 of each wave
  for (uint64_t d = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; d < n_words * 64; d += (uint64_t)gridDim.x * blockDim.x) {
    const uint64_t w = d >> 6;
    const u64 word = universe[w];   // wave-uniform
    u64 mask = 0;
    if (word) {
      const bool hit = d < n_docs && ((word >> (d & 63)) & 1ull) && keys[d] == key;
      mask = __ballot(hit);
    }
    if ((threadIdx.x & 63) == 0) {
      bucket[w] = mask;
      if (mask) universe[w] = word & ~mask;
      cnt += (uint32_t)__popcll(mask);
    }
  }
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = cnt;
  __syncthreads();
  if (threadIdx.x == 0) {
    u64 b = 0;
    for (int i = 0; i < BT / 64; ++i) b += part[i];
    if (b) atomicAdd(&acc[0], b);
    MSI_ORDER_ATOMICS();   // atomics only: no L2 write-back / invalidate (msi_common.h)
    const u64 done = atomicAdd(&acc[1], 1ull);
    if (done == gridDim.x - 1) {
      const u64 total = atomicExch(&acc[0], 0ull);
      acc[1] = 0;
      atomicExch(best, 0ull);  // every workgroup has read it: re-armed for the next call
      __hip_atomic_store(const_cast<uint64_t *>(&sig[0]), total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      __hip_atomic_store(const_cast<uint64_t *>(&sig[2]), (uint64_t)key, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      __hip_atomic_store(const_cast<uint64_t *>(&sig[1]), seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
}

// ---- distinct (crates/milli/src/search/new/distinct.rs:19-62) --------------------------------------------------------
// apply_distinct_rule keeps, in ascending docid order, every candidate that shares no facet value of the distinct
// field with a candidate kept before it: the lexicographically first maximal independent set of the "shares a value"
// graph.  Parallel form, one document per thread (a wave = one 64-bit word of a set, __ballot = the word):
//   propose: every undecided candidate writes its docid into first[v] (atomicMin) for each of its values v;
//   join:    a candidate that owns first[v] for ALL its values is kept and stamps taken[v];
//   propose (next round) first drops the undecided candidates that hold a taken value.
// A candidate is kept in round r iff it is the smallest undecided document of its closed neighbourhood — the rounds
// reproduce the sequential loop exactly.  A single-valued field is decided in one round.
__device__ __forceinline__ u64 dv_pack(uint32_t round, uint32_t docid) { return ((u64)(~round) << 32) | (u64)docid; }

// universe -= holders of a taken value; the survivors propose; the last workgroup publishes {|universe|, kept so far}.
__global__ void bits_distinct_propose_kernel(u64 *__restrict__ undecided, const uint32_t *__restrict__ offsets,
                                             const uint32_t *__restrict__ values, uint64_t n_docs, uint64_t n_words,
                                             u64 *__restrict__ first, const uint32_t *__restrict__ taken,
                                             uint32_t call_stamp, uint32_t round, int first_round,
                                             u64 *__restrict__ kept_acc, u64 *__restrict__ acc,
                                             volatile uint64_t *__restrict__ sig, uint64_t seq) {
  const uint64_t d = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const uint64_t w = d >> 6;
  if (first_round && d == 0) *kept_acc = 0;  // the join kernels of this call run after this launch
  const u64 word = w < n_words ? undecided[w] : 0ull;  // one address per wave
  bool alive = false, pruned = false;
  if (d < n_docs && ((word >> (d & 63)) & 1ull)) {
    const uint32_t lo = offsets[d], hi = offsets[d + 1];
    for (uint32_t i = lo; i < hi && !pruned; ++i) pruned = taken[values[i]] == call_stamp;
    alive = !pruned;
    if (alive)
      for (uint32_t i = lo; i < hi; ++i) atomicMin(&first[values[i]], dv_pack(round, (uint32_t)d));
  }
  const u64 keep = __ballot(alive), drop = __ballot(pruned);
  __shared__ uint32_t part[BT / 64];
  if ((threadIdx.x & 63) == 0) {
    if (drop) undecided[w] = keep;
    part[threadIdx.x >> 6] = (uint32_t)__popcll(keep);
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    u64 b = 0;
    for (int i = 0; i < BT / 64; ++i) b += part[i];
    if (b) atomicAdd(&acc[0], b);
    MSI_ORDER_ATOMICS();   // atomics only: no L2 write-back / invalidate (msi_common.h)
    const u64 done = atomicAdd(&acc[1], 1ull);
    if (done == gridDim.x - 1) {
      const u64 total = atomicExch(&acc[0], 0ull);
      acc[1] = 0;
      const u64 kept = first_round ? 0ull : __hip_atomic_load(kept_acc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(const_cast<uint64_t *>(&sig[0]), total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      __hip_atomic_store(const_cast<uint64_t *>(&sig[2]), kept, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      __hip_atomic_store(const_cast<uint64_t *>(&sig[1]), seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
}

// kept |= the undecided candidates that own every one of their values (first round: kept := them, the slot's old
// content must not survive); they leave `undecided` and stamp their values
__global__ void bits_distinct_join_kernel(u64 *__restrict__ undecided, u64 *__restrict__ kept,
                                          const uint32_t *__restrict__ offsets, const uint32_t *__restrict__ values,
                                          uint64_t n_docs, uint64_t n_words, const u64 *__restrict__ first,
                                          uint32_t *__restrict__ taken, uint32_t call_stamp, uint32_t round,
                                          int first_round, u64 *__restrict__ kept_acc) {
  const uint64_t d = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const uint64_t w = d >> 6;
  const u64 word = w < n_words ? undecided[w] : 0ull;
  bool ok = false;
  if (d < n_docs && ((word >> (d & 63)) & 1ull)) {
    const uint32_t lo = offsets[d], hi = offsets[d + 1];
    const u64 mine = dv_pack(round, (uint32_t)d);
    ok = true;
    for (uint32_t i = lo; i < hi && ok; ++i) ok = first[values[i]] == mine;
    if (ok)
      for (uint32_t i = lo; i < hi; ++i) taken[values[i]] = call_stamp;  // only the owner of a value writes it
  }
  const u64 joined = __ballot(ok);
  if ((threadIdx.x & 63) == 0 && w < n_words) {
    if (first_round) kept[w] = joined;
    else if (joined) kept[w] |= joined;
    if (joined) {
      undecided[w] = word & ~joined;
      atomicAdd(kept_acc, (u64)__popcll(joined));
    }
  }
}

// What the rounds left undecided, in the reference's own order, by one thread: bounds the work on pathological
// inputs (a chain d0 - d1 - d2 - ... of shared values needs one round per two documents).
__global__ void bits_distinct_sequential_kernel(u64 *__restrict__ undecided, u64 *__restrict__ kept,
                                                const uint32_t *__restrict__ offsets, const uint32_t *__restrict__ values,
                                                uint64_t n_words, uint32_t *__restrict__ taken, uint32_t call_stamp,
                                                u64 *__restrict__ kept_acc, volatile uint64_t *__restrict__ sig,
                                                uint64_t seq) {
  if (blockIdx.x != 0 || threadIdx.x != 0) return;
  u64 n = 0;
  for (uint64_t w = 0; w < n_words; ++w) {
    u64 bits = undecided[w];
    if (!bits) continue;
    u64 add = 0;
    while (bits) {
      const uint32_t b = (uint32_t)__ffsll((long long)bits) - 1;
      bits &= bits - 1;
      const uint64_t d = w * 64 + b;
      const uint32_t lo = offsets[d], hi = offsets[d + 1];
      bool free_ = true;
      for (uint32_t i = lo; i < hi && free_; ++i) free_ = taken[values[i]] != call_stamp;
      if (!free_) continue;
      for (uint32_t i = lo; i < hi; ++i) taken[values[i]] = call_stamp;
      add |= 1ull << b;
      ++n;
    }
    undecided[w] = 0;
    if (add) kept[w] |= add;
  }
  const u64 total = *kept_acc + n;
  __hip_atomic_store(const_cast<uint64_t *>(&sig[0]), 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  __hip_atomic_store(const_cast<uint64_t *>(&sig[2]), total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  __hip_atomic_store(const_cast<uint64_t *>(&sig[1]), seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// taken[v] := stamp for the values of the documents of `kept` (msi_bits_distinct_excluded)
__global__ void bits_distinct_mark_kernel(const u64 *__restrict__ kept, const uint32_t *__restrict__ offsets,
                                          const uint32_t *__restrict__ values, uint64_t n_docs, uint64_t n_words,
                                          uint32_t *__restrict__ taken, uint32_t call_stamp) {
  const uint64_t d = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const uint64_t w = d >> 6;
  if (d >= n_docs || w >= n_words || !((kept[w] >> (d & 63)) & 1ull)) return;
  for (uint32_t i = offsets[d]; i < offsets[d + 1]; ++i) taken[values[i]] = call_stamp;
}

// excluded := every document of the index that holds a stamped value (the whole slot is overwritten)
__global__ void bits_distinct_excluded_kernel(u64 *__restrict__ excluded, const uint32_t *__restrict__ offsets,
                                              const uint32_t *__restrict__ values, uint64_t n_docs, uint64_t n_words,
                                              const uint32_t *__restrict__ taken, uint32_t call_stamp) {
  const uint64_t d = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const uint64_t w = d >> 6;
  bool ex = false;
  if (d < n_docs) {
    const uint32_t lo = offsets[d], hi = offsets[d + 1];
    for (uint32_t i = lo; i < hi && !ex; ++i) ex = taken[values[i]] == call_stamp;
  }
  const u64 mask = __ballot(ex);
  if ((threadIdx.x & 63) == 0 && w < n_words) excluded[w] = mask;
}

// ---- GeoSort (crates/milli/src/search/new/geo_sort.rs, documents/geo_sort.rs:150-224) --------------------------------
// distance_between_two_points (lib.rs:388-393) = geoutils 0.5.1's haversine_distance_to: hav(t) = (1 - cos t) / 2 on
// the radian differences, mean radius 6371 km, metres rounded to millimetres.  The operation order is the crate's;
// the translation unit is built with -ffp-contract=off.  Distances are >= 0, so their bit patterns order like them.
struct GeoTarget {
  double phi, cos_phi, lam;  // of the target point, radians
  double margin;
  int ascending;
};
__device__ __forceinline__ double geo_distance_m(const GeoTarget &t, double lat, double lng) {
  const double D2R = 3.14159265358979323846 / 180.0;  // f64::to_radians
  const double phi2 = lat * D2R, lam2 = lng * D2R;
  const double hav_phi = (1.0 - cos(phi2 - t.phi)) / 2.0;
  const double hav_lam = t.cos_phi * cos(phi2) * ((1.0 - cos(lam2 - t.lam)) / 2.0);
  const double total = hav_phi + hav_lam;
  return round(2.0 * 6371e3 * asin(sqrt(total)) * 1000.0) / 1000.0;
}
__device__ __forceinline__ u64 shfl_xor_u64(u64 v, int o) {
  const uint32_t lo = (uint32_t)__shfl_xor((int)(uint32_t)v, o), hi = (uint32_t)__shfl_xor((int)(uint32_t)(v >> 32), o);
  return ((u64)hi << 32) | lo;
}
// sort key of a distance: ascending rules minimise the bits, descending rules the complement; ~0 = no point
__device__ __forceinline__ u64 geo_key(const GeoTarget &t, double dist) {
  const u64 b = (u64)__double_as_longlong(dist);
  return t.ascending ? b : ~b - 1;  // `- 1` keeps ~0 free for "none" (a distance of +0.0 has all-zero bits)
}

// best = min over the documents of `src` that have a point of their distance key (one document per thread)
__global__ void bits_geo_min_kernel(const u64 *__restrict__ src, const double *__restrict__ lat_lng, uint64_t n_docs,
                                    uint64_t n_words, GeoTarget t, u64 *__restrict__ best) {
  __shared__ u64 part[BT / 64];
  u64 k = ~0ull;
  for (uint64_t d = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; d < n_words * 64; d += (uint64_t)gridDim.x * blockDim.x) {
    const u64 word = src[d >> 6];   // wave-uniform
    if (!word) continue;
    if (d < n_docs && ((word >> (d & 63)) & 1ull)) {
      const double lat = lat_lng[2 * d];
      if (lat == lat) {
        const u64 kk = geo_key(t, geo_distance_m(t, lat, lat_lng[2 * d + 1]));
        k = kk < k ? kk : k;
      }
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const u64 other = shfl_xor_u64(k, o);
    k = other < k ? other : k;
  }
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = k;
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int i = 1; i < BT / 64; ++i) k = part[i] < k ? part[i] : k;
    if (k != ~0ull) atomicMin(best, k);   // one atomic per workgroup
  }
}

// dst := {d in src with a point : key_lo <= key(d) <= key_hi} (mode 0; mode 2: dst |= them) or, relative to the extreme found by the min
// kernel, the documents within the error margin of it (mode 1; they also leave `src`, and the smallest docid at the
// extreme itself is collected).  The last workgroup publishes {|dst|, first docid} and re-arms the cells.
__global__ void bits_geo_take_kernel(u64 *__restrict__ src, u64 *__restrict__ dst, const double *__restrict__ lat_lng,
                                     uint64_t n_docs, uint64_t n_words, GeoTarget t, int mode, u64 key_lo, u64 key_hi,
                                     u64 *__restrict__ best, u64 *__restrict__ first, u64 *__restrict__ acc,
                                     volatile uint64_t *__restrict__ sig, uint64_t seq) {
  const u64 kbest = mode == 1 ? __hip_atomic_load(best, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0ull;
  __shared__ uint32_t part[BT / 64];
  __shared__ u64 part_first[BT / 64];
  uint32_t cnt = 0;      // lane 0 of each wave
  u64 mine_first = ~0ull;
  for (uint64_t d = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; d < n_words * 64; d += (uint64_t)gridDim.x * blockDim.x) {
    const uint64_t w = d >> 6;
    const u64 word = src[w];   // wave-uniform
    bool hit = false;
    if (d < n_docs && ((word >> (d & 63)) & 1ull) && (mode != 1 || kbest != ~0ull)) {
      const double lat = lat_lng[2 * d];
      if (lat == lat) {
        const double dist = geo_distance_m(t, lat, lat_lng[2 * d + 1]);
        const u64 k = geo_key(t, dist);
        if (mode != 1) {
          hit = k >= key_lo && k <= key_hi;
        } else {
          const double d0 = __longlong_as_double((long long)(t.ascending ? kbest : ~(kbest + 1)));
          hit = fabs(d0 - dist) <= t.margin;  // documents/geo_sort.rs:181
          if (k == kbest && d < mine_first) mine_first = d;
        }
      }
    }
    const u64 mask = __ballot(hit);
    if ((threadIdx.x & 63) == 0) {
      if (mode == 2) {
        if (mask) dst[w] |= mask;
      } else {
        dst[w] = mask;
      }
      if (mode == 1 && mask) src[w] = word & ~mask;
      cnt += (uint32_t)__popcll(mask);
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const u64 other = shfl_xor_u64(mine_first, o);
    mine_first = other < mine_first ? other : mine_first;
  }
  if ((threadIdx.x & 63) == 0) {
    part[threadIdx.x >> 6] = cnt;
    part_first[threadIdx.x >> 6] = mine_first;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    u64 b = 0, f0 = ~0ull;
    for (int i = 0; i < BT / 64; ++i) {
      b += part[i];
      f0 = part_first[i] < f0 ? part_first[i] : f0;
    }
    if (b) atomicAdd(&acc[0], b);
    if (f0 != ~0ull) atomicMin(first, f0);
    MSI_ORDER_ATOMICS();   // atomics only: no L2 write-back / invalidate (msi_common.h)
    const u64 done = atomicAdd(&acc[1], 1ull);
    if (done == gridDim.x - 1) {
      const u64 total = atomicExch(&acc[0], 0ull);
      acc[1] = 0;
      u64 f = ~0ull, kb = 0;
      if (mode == 1) {
        f = atomicExch(first, ~0ull);
        kb = atomicExch(best, ~0ull);  // every workgroup has read it: re-armed for the next call
      }
      __hip_atomic_store(const_cast<uint64_t *>(&sig[0]), total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      __hip_atomic_store(const_cast<uint64_t *>(&sig[2]), f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      __hip_atomic_store(const_cast<uint64_t *>(&sig[3]), kb, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      __hip_atomic_store(const_cast<uint64_t *>(&sig[1]), seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
}

// The documents of `src` that have a point, with their distance: appended in no particular order (one atomic slot per
// document), all of them counted.
__global__ void bits_geo_list_kernel(const u64 *__restrict__ src, const double *__restrict__ lat_lng, uint64_t n_docs,
                                     uint64_t n_words, GeoTarget t, uint32_t cap, uint32_t *__restrict__ out_ids,
                                     double *__restrict__ out_dist, u64 *__restrict__ counter) {
  for (uint64_t d = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; d < n_words * 64; d += (uint64_t)gridDim.x * blockDim.x) {
    const u64 word = src[d >> 6];   // wave-uniform
    if (!word) continue;
    if (d < n_docs && ((word >> (d & 63)) & 1ull)) {
      const double lat = lat_lng[2 * d];
      if (lat == lat) {
        const u64 at = atomicAdd(counter, 1ull);
        if (at < cap) {
          out_ids[at] = (uint32_t)d;
          out_dist[at] = geo_distance_m(t, lat, lat_lng[2 * d + 1]);
        }
      }
    }
  }
}

// ---- filter leaves (crates/milli/src/search/facet/filter/index_filter.rs:84-340) ---------------------------------------
// dst (|)= {d : some key of d is in [lo, hi]} — or, with a sorted list, is one of its n keys.  One document per thread,
// the wave's ballot is the word of the set.
__global__ void bits_facet_select_kernel(u64 *__restrict__ dst, const uint32_t *__restrict__ offsets,
                                         const u64 *__restrict__ keys, uint64_t n_docs, uint64_t n_words, u64 lo, u64 hi,
                                         const u64 *__restrict__ list, uint64_t n_list, int accumulate) {
  const uint64_t d = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const uint64_t w = d >> 6;
  bool hit = false;
  if (d < n_docs) {
    const uint32_t b = offsets[d], e = offsets[d + 1];
    for (uint32_t i = b; i < e && !hit; ++i) {
      const u64 k = keys[i];
      if (!list) {
        hit = k >= lo && k <= hi;
      } else {
        uint64_t a = 0, z = n_list;  // lower bound
        while (a < z) {
          const uint64_t m = (a + z) >> 1;
          if (list[m] < k) a = m + 1;
          else z = m;
        }
        hit = a < n_list && list[a] == k;
      }
    }
  }
  const u64 mask = __ballot(hit);
  if ((threadIdx.x & 63) == 0 && w < n_words) {
    if (accumulate) {
      if (mask) dst[w] |= mask;
    } else {
      dst[w] = mask;
    }
  }
}

struct ManyArgs {
  const u64 *cond[MSI_BITS_MANY];
  u64 *dst[MSI_BITS_MANY];
  u64 *stack[MSI_BITS_MANY];
};

// dst[i] = prefix & cond[i] with |dst[i]| for every i < n; the last workgroup publishes the n counts.
__global__ void bits_and_many_kernel(ManyArgs a, const u64 *__restrict__ prefix, uint32_t n, uint64_t n_pairs,
                                     u64 *__restrict__ acc, volatile uint64_t *__restrict__ sig, uint64_t seq) {
  uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  uint32_t c[MSI_BITS_MANY];
#pragma unroll
  for (uint32_t k = 0; k < MSI_BITS_MANY; ++k) c[k] = 0;
  for (; i < n_pairs; i += stride) {
    const ulonglong2 x = reinterpret_cast<const ulonglong2 *>(prefix)[i];
#pragma unroll
    for (uint32_t k = 0; k < MSI_BITS_MANY; ++k) {
      if (k < n) {
        const ulonglong2 y = reinterpret_cast<const ulonglong2 *>(a.cond[k])[i];
        ulonglong2 r;
        r.x = x.x & y.x;
        r.y = x.y & y.y;
        reinterpret_cast<ulonglong2 *>(a.dst[k])[i] = r;
        c[k] += __popcll(r.x) + __popcll(r.y);
      }
    }
  }
  __shared__ uint32_t part[MSI_BITS_MANY];
  if (threadIdx.x < MSI_BITS_MANY) part[threadIdx.x] = 0;
  __syncthreads();
#pragma unroll
  for (uint32_t k = 0; k < MSI_BITS_MANY; ++k) {
    if (k < n) {
      uint32_t v = c[k];
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o);
      if ((threadIdx.x & 63) == 0 && v) atomicAdd(&part[k], v);
    }
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (uint32_t k = 0; k < n; ++k)
      if (part[k]) atomicAdd(&acc[2 + k], (u64)part[k]);
    MSI_ORDER_ATOMICS();   // atomics only: no L2 write-back / invalidate (msi_common.h)
    const u64 done = atomicAdd(&acc[1], 1ull);
    if (done == gridDim.x - 1) {
      for (uint32_t k = 0; k < n; ++k) {
        const u64 total = atomicExch(&acc[2 + k], 0ull);
        __hip_atomic_store(const_cast<uint64_t *>(&sig[2 + k]), total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      }
      acc[1] = 0;
      __hip_atomic_store(const_cast<uint64_t *>(&sig[1]), seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
}

// dst[i] &= ~removed with |dst[i]| for every i < n (the universes of the rule stack after a distinct pass)
__global__ void bits_andnot_many_kernel(ManyArgs a, const u64 *__restrict__ removed, uint32_t n, uint64_t n_pairs,
                                        u64 *__restrict__ acc, volatile uint64_t *__restrict__ sig, uint64_t seq) {
  uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  uint32_t c[MSI_BITS_MANY];
#pragma unroll
  for (uint32_t k = 0; k < MSI_BITS_MANY; ++k) c[k] = 0;
  for (; i < n_pairs; i += stride) {
    const ulonglong2 x = reinterpret_cast<const ulonglong2 *>(removed)[i];
#pragma unroll
    for (uint32_t k = 0; k < MSI_BITS_MANY; ++k) {
      if (k < n) {
        ulonglong2 r = reinterpret_cast<const ulonglong2 *>(a.dst[k])[i];
        if ((r.x & x.x) | (r.y & x.y)) {
          r.x &= ~x.x;
          r.y &= ~x.y;
          reinterpret_cast<ulonglong2 *>(a.dst[k])[i] = r;
        }
        c[k] += __popcll(r.x) + __popcll(r.y);
      }
    }
  }
  __shared__ uint32_t part[MSI_BITS_MANY];
  if (threadIdx.x < MSI_BITS_MANY) part[threadIdx.x] = 0;
  __syncthreads();
#pragma unroll
  for (uint32_t k = 0; k < MSI_BITS_MANY; ++k) {
    if (k < n) {
      uint32_t v = c[k];
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o);
      if ((threadIdx.x & 63) == 0 && v) atomicAdd(&part[k], v);
    }
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (uint32_t k = 0; k < n; ++k)
      if (part[k]) atomicAdd(&acc[2 + k], (u64)part[k]);
    MSI_ORDER_ATOMICS();   // atomics only: no L2 write-back / invalidate (msi_common.h)
    const u64 done = atomicAdd(&acc[1], 1ull);
    if (done == gridDim.x - 1) {
      for (uint32_t k = 0; k < n; ++k) {
        const u64 total = atomicExch(&acc[2 + k], 0ull);
        __hip_atomic_store(const_cast<uint64_t *>(&sig[2 + k]), total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      }
      acc[1] = 0;
      __hip_atomic_store(const_cast<uint64_t *>(&sig[1]), seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
}

// bucket |= docs; universe &= ~docs; stack[i] &= ~docs (docs is read before any of them is written)
__global__ void bits_claim_kernel(ManyArgs a, const u64 *docs, u64 *bucket, u64 *universe, uint32_t n_stack,
                                  uint64_t n_pairs) {
  uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  for (; i < n_pairs; i += stride) {
    const ulonglong2 d = reinterpret_cast<const ulonglong2 *>(docs)[i];
    if (!(d.x | d.y)) continue;
    ulonglong2 b = reinterpret_cast<ulonglong2 *>(bucket)[i];
    b.x |= d.x;
    b.y |= d.y;
    reinterpret_cast<ulonglong2 *>(bucket)[i] = b;
    ulonglong2 u = reinterpret_cast<ulonglong2 *>(universe)[i];
    u.x &= ~d.x;
    u.y &= ~d.y;
    reinterpret_cast<ulonglong2 *>(universe)[i] = u;
    for (uint32_t k = 0; k < n_stack; ++k) {
      ulonglong2 s = reinterpret_cast<ulonglong2 *>(a.stack[k])[i];
      s.x &= ~d.x;
      s.y &= ~d.y;
      reinterpret_cast<ulonglong2 *>(a.stack[k])[i] = s;
    }
  }
}

// Paths of one cost level, in DFS order: every 16-byte chunk of documents is independent, so the sequential
// "a path claims what the earlier paths left" is a loop per thread.  steps[path_off[k] .. path_off[k+1]) are the
// condition sets of path k.
__global__ void bits_paths_kernel(const u64 *const *__restrict__ steps, const uint32_t *__restrict__ path_off,
                                  uint32_t n_paths, u64 *__restrict__ bucket, u64 *__restrict__ universe,
                                  uint64_t n_pairs, u64 *__restrict__ acc_counts, u64 *__restrict__ acc,
                                  volatile uint64_t *__restrict__ sig_counts, volatile uint64_t *__restrict__ sig,
                                  uint64_t seq) {
  __shared__ uint32_t cnt[MSI_BITS_MAX_PATHS];
  for (uint32_t k = threadIdx.x; k < n_paths; k += blockDim.x) cnt[k] = 0;
  __syncthreads();
  uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  for (; i < n_pairs; i += stride) {
    ulonglong2 u = reinterpret_cast<ulonglong2 *>(universe)[i];
    if (!(u.x | u.y)) continue;
    ulonglong2 b = reinterpret_cast<ulonglong2 *>(bucket)[i];
    for (uint32_t k = 0; k < n_paths && (u.x | u.y); ++k) {
      ulonglong2 m = u;
      for (uint32_t s = path_off[k]; s < path_off[k + 1] && (m.x | m.y); ++s) {
        const ulonglong2 c = reinterpret_cast<const ulonglong2 *>(steps[s])[i];
        m.x &= c.x;
        m.y &= c.y;
      }
      if (m.x | m.y) {
        b.x |= m.x;
        b.y |= m.y;
        u.x &= ~m.x;
        u.y &= ~m.y;
        atomicAdd(&cnt[k], (uint32_t)(__popcll(m.x) + __popcll(m.y)));
      }
    }
    reinterpret_cast<ulonglong2 *>(bucket)[i] = b;
    reinterpret_cast<ulonglong2 *>(universe)[i] = u;
  }
  __syncthreads();
  for (uint32_t k = threadIdx.x; k < n_paths; k += blockDim.x)
    if (cnt[k]) atomicAdd(&acc_counts[k], (u64)cnt[k]);
  MSI_ORDER_ATOMICS();   // atomics only: no L2 write-back / invalidate (msi_common.h)
  __syncthreads();
  __shared__ bool last;
  if (threadIdx.x == 0) last = atomicAdd(&acc[1], 1ull) == gridDim.x - 1;
  __syncthreads();
  if (last) {
    for (uint32_t k = threadIdx.x; k < n_paths; k += blockDim.x) {
      const u64 total = atomicExch(&acc_counts[k], 0ull);
      __hip_atomic_store(const_cast<uint64_t *>(&sig_counts[k]), total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    MSI_ORDER_ATOMICS();
    __syncthreads();
    if (threadIdx.x == 0) {
      acc[1] = 0;
      __hip_atomic_store(const_cast<uint64_t *>(&sig[1]), seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
}

// The common case of a cost level — few distinct conditions (<= 32), <= 64 paths — with everything in the kernel
// arguments (no descriptor copy) and every condition chunk loaded ONCE, up front and independently, into LDS:
// the path loop then runs on LDS instead of a chain of dependent global loads.
constexpr uint32_t PS_CONDS = 32, PS_PATHS = 64, PS_STEPS = 448, PS_T = 64;
struct PathsSmall {
  const u64 *cond[PS_CONDS];
  uint16_t off[PS_PATHS + 1];
  uint8_t step[PS_STEPS];
};
__global__ __launch_bounds__(PS_T) void bits_paths_small_kernel(PathsSmall a, uint32_t n_conds, uint32_t n_paths,
                                                                 u64 *__restrict__ bucket, u64 *__restrict__ universe,
                                                                 uint64_t n_pairs, u64 *__restrict__ acc_counts,
                                                                 u64 *__restrict__ acc,
                                                                 volatile uint64_t *__restrict__ sig_counts,
                                                                 volatile uint64_t *__restrict__ sig, uint64_t seq) {
  __shared__ ulonglong2 cv[PS_CONDS][PS_T];
  __shared__ uint32_t cnt[PS_PATHS];
  const uint32_t tid = threadIdx.x;
  if (tid < n_paths) cnt[tid] = 0;
  __syncthreads();
  const uint64_t i = (uint64_t)blockIdx.x * PS_T + tid;
  if (i < n_pairs) {
    ulonglong2 u = reinterpret_cast<ulonglong2 *>(universe)[i];
    if (u.x | u.y) {
      for (uint32_t c = 0; c < n_conds; ++c) cv[c][tid] = reinterpret_cast<const ulonglong2 *>(a.cond[c])[i];
      ulonglong2 b = reinterpret_cast<ulonglong2 *>(bucket)[i];
      for (uint32_t k = 0; k < n_paths && (u.x | u.y); ++k) {
        ulonglong2 m = u;
        for (uint32_t s = a.off[k]; s < a.off[k + 1] && (m.x | m.y); ++s) {
          const ulonglong2 c = cv[a.step[s]][tid];
          m.x &= c.x;
          m.y &= c.y;
        }
        if (m.x | m.y) {
          b.x |= m.x;
          b.y |= m.y;
          u.x &= ~m.x;
          u.y &= ~m.y;
          atomicAdd(&cnt[k], (uint32_t)(__popcll(m.x) + __popcll(m.y)));
        }
      }
      reinterpret_cast<ulonglong2 *>(bucket)[i] = b;
      reinterpret_cast<ulonglong2 *>(universe)[i] = u;
    }
  }
  __syncthreads();
  if (tid < n_paths && cnt[tid]) atomicAdd(&acc_counts[tid], (u64)cnt[tid]);
  MSI_ORDER_ATOMICS();   // atomics only: no L2 write-back / invalidate (msi_common.h)
  __syncthreads();
  __shared__ bool last;
  if (tid == 0) last = atomicAdd(&acc[1], 1ull) == gridDim.x - 1;
  __syncthreads();
  if (last) {
    if (tid < n_paths) {
      const u64 total = atomicExch(&acc_counts[tid], 0ull);
      __hip_atomic_store(const_cast<uint64_t *>(&sig_counts[tid]), total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    MSI_ORDER_ATOMICS();
    __syncthreads();
    if (tid == 0) {
      acc[1] = 0;
      __hip_atomic_store(const_cast<uint64_t *>(&sig[1]), seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
}

// dst = (OR_i pool[srcs[i]]) & pool[universe]
__global__ void bits_union_many_kernel(u64 *__restrict__ pool, uint64_t n_words, uint32_t dst,
                                       const uint32_t *__restrict__ srcs, uint32_t n, uint32_t universe) {
  uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  const uint64_t n_pairs = n_words / 2;
  for (; i < n_pairs; i += stride) {
    ulonglong2 acc = make_ulonglong2(0, 0);
    for (uint32_t s = 0; s < n; ++s) {
      const ulonglong2 v = reinterpret_cast<const ulonglong2 *>(pool + (uint64_t)srcs[s] * n_words)[i];
      acc.x |= v.x;
      acc.y |= v.y;
    }
    if (universe != 0xFFFFFFFFu) {
      const ulonglong2 u = reinterpret_cast<const ulonglong2 *>(pool + (uint64_t)universe * n_words)[i];
      acc.x &= u.x;
      acc.y &= u.y;
    }
    reinterpret_cast<ulonglong2 *>(pool + (uint64_t)dst * n_words)[i] = acc;
  }
}

__global__ void bits_count_kernel(const u64 *__restrict__ a, uint64_t n_words, u64 *__restrict__ acc,
                                  volatile uint64_t *__restrict__ sig, uint64_t seq) {
  uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  uint32_t c = 0;
  for (; i < n_words; i += stride) c += __popcll(a[i]);
  publish_count(c, acc, sig, seq);
}

__global__ void bits_set_docids_kernel(u64 *__restrict__ dst, uint64_t n_docs,
                                       const uint32_t *__restrict__ ids, uint64_t n) {
  uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint32_t id = ids[i];
  if ((uint64_t)id < n_docs) atomicOr(&dst[id >> 6], 1ull << (id & 63));
}

// slot |= {docids[i] : i < n, docids[i] < n_docs} — the items of a vector store as a docid set (`_vectors` filter leaf)
__global__ void bits_or_docids_kernel(u64 *__restrict__ slot, uint64_t n_docs, const uint32_t *__restrict__ docids, uint64_t n) {
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
    const uint32_t d = docids[i];
    if (d < n_docs) atomicOr(&slot[d >> 6], 1ull << (d & 63));
  }
}

// slot(first + y * stride) := the documents of list y (device-resident lists, e.g. the top-k rows of a vector
// search: the rerank universes of many queries in two launches)
__global__ void bits_clear_slots_kernel(u64 *__restrict__ pool, uint64_t n_words, uint32_t first, uint32_t stride) {
  u64 *dst = pool + (uint64_t)(first + blockIdx.y * stride) * n_words;
  uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const uint64_t step = (uint64_t)gridDim.x * blockDim.x;
  for (; i < n_words / 2; i += step) reinterpret_cast<ulonglong2 *>(dst)[i] = make_ulonglong2(0, 0);
}
__global__ void bits_set_lists_kernel(u64 *__restrict__ pool, uint64_t n_words, uint64_t n_docs, uint32_t first,
                                      uint32_t stride, const uint32_t *__restrict__ ids, uint32_t list_stride,
                                      const uint32_t *__restrict__ counts) {
  u64 *dst = pool + (uint64_t)(first + blockIdx.y * stride) * n_words;
  const uint32_t n = counts[blockIdx.y] < list_stride ? counts[blockIdx.y] : list_stride;
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const uint32_t id = ids[(uint64_t)blockIdx.y * list_stride + i];
    if ((uint64_t)id < n_docs) atomicOr(&dst[id >> 6], 1ull << (id & 63));
  }
}

struct ClearArgs {
  u64 *slot[MSI_BITS_CLEAR_MAX];
};
__global__ void bits_clear_many_kernel(ClearArgs a, uint64_t n_pairs) {
  u64 *dst = a.slot[blockIdx.y];
  uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const uint64_t step = (uint64_t)gridDim.x * blockDim.x;
  for (; i < n_pairs; i += step) reinterpret_cast<ulonglong2 *>(dst)[i] = make_ulonglong2(0, 0);
}

// One block per Roaring container.
__global__ void bits_decode_roaring_kernel(u64 *__restrict__ dst, uint64_t n_docs,
                                           const uint8_t *__restrict__ bytes,
                                           const Container *__restrict__ cs) {
  const Container c = cs[blockIdx.x];
  const uint64_t base = (uint64_t)c.key << 16;
  const uint8_t *body = bytes + c.offset;
  if (c.type == 0) {
    for (uint32_t i = threadIdx.x; i < c.card; i += blockDim.x) {
      const uint64_t id = base + ((uint32_t)body[2 * i] | ((uint32_t)body[2 * i + 1] << 8));
      if (id < n_docs) atomicOr(&dst[id >> 6], 1ull << (id & 63));
    }
  } else if (c.type == 1) {
    for (uint32_t w = threadIdx.x; w < 1024; w += blockDim.x) {
      u64 v = 0;
      for (int b = 0; b < 8; ++b) v |= (u64)body[8 * w + b] << (8 * b);
      const uint64_t wi = (base >> 6) + w;
      if (v && wi * 64 < n_docs) atomicOr(&dst[wi], v);
    }
  } else {
    for (uint32_t r = 0; r < c.card; ++r) {
      const uint32_t start = (uint32_t)body[4 * r] | ((uint32_t)body[4 * r + 1] << 8);
      const uint32_t len = ((uint32_t)body[4 * r + 2] | ((uint32_t)body[4 * r + 3] << 8)) + 1;
      for (uint32_t i = threadIdx.x; i < len; i += blockDim.x) {
        const uint64_t id = base + start + i;
        if (id < n_docs) atomicOr(&dst[id >> 6], 1ull << (id & 63));
      }
    }
  }
}

// first_k: (1) per-block popcounts, (2) single-block exclusive scan, (3) emit.
__global__ void bits_block_counts_kernel(const u64 *__restrict__ a, uint64_t n_words,
                                         uint32_t *__restrict__ blk) {
  __shared__ uint32_t sh[BT / 64];
  uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  uint32_t c = i < n_words ? __popcll(a[i]) : 0;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) c += __shfl_down(c, o);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = c;
  __syncthreads();
  if (threadIdx.x == 0) {
    uint32_t t = 0;
    for (int w = 0; w < BT / 64; ++w) t += sh[w];
    blk[blockIdx.x] = t;
  }
}

__global__ void bits_scan_counts_kernel(uint32_t *__restrict__ blk, uint32_t n_blocks,
                                        uint32_t *__restrict__ total) {
  // single block; serial over chunks of blockDim.x
  __shared__ uint32_t sh[BT];
  __shared__ uint32_t carry;
  if (threadIdx.x == 0) carry = 0;
  __syncthreads();
  for (uint32_t base = 0; base < n_blocks; base += BT) {
    const uint32_t i = base + threadIdx.x;
    const uint32_t v = i < n_blocks ? blk[i] : 0;
    sh[threadIdx.x] = v;
    __syncthreads();
    for (uint32_t o = 1; o < BT; o <<= 1) {
      uint32_t t = threadIdx.x >= o ? sh[threadIdx.x - o] : 0;
      __syncthreads();
      sh[threadIdx.x] += t;
      __syncthreads();
    }
    const uint32_t incl = sh[threadIdx.x];
    if (i < n_blocks) blk[i] = carry + incl - v;  // exclusive
    __syncthreads();
    if (threadIdx.x == BT - 1) carry += incl;
    __syncthreads();
  }
  if (threadIdx.x == 0) *total = carry;
}

__global__ void bits_emit_first_k_kernel(const u64 *__restrict__ a, uint64_t n_words,
                                         const uint32_t *__restrict__ blk_excl, uint32_t k,
                                         uint32_t *__restrict__ out) {
  __shared__ uint32_t sh[BT];
  const uint32_t blk_base = blk_excl[blockIdx.x];
  if (blk_base >= k) return;
  uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  u64 w = i < n_words ? a[i] : 0;
  const uint32_t c = __popcll(w);
  sh[threadIdx.x] = c;
  __syncthreads();
  for (uint32_t o = 1; o < BT; o <<= 1) {
    uint32_t t = threadIdx.x >= o ? sh[threadIdx.x - o] : 0;
    __syncthreads();
    sh[threadIdx.x] += t;
    __syncthreads();
  }
  uint32_t rank = blk_base + sh[threadIdx.x] - c;
  while (w && rank < k) {
    const uint32_t b = __ffsll((long long)w) - 1;
    out[rank++] = (uint32_t)(i * 64 + b);
    w &= w - 1;
  }
}

uint32_t grid_for(uint64_t n, uint32_t cap_blocks) {
  uint64_t b = (n + BT - 1) / BT;
  if (b < 1) b = 1;
  return (uint32_t)std::min<uint64_t>(b, cap_blocks);
}

// Waits for the counting kernel launched with sequence number `seq`; spins on the pinned signal and
// falls back to a stream synchronisation if the signal does not arrive promptly.
int32_t wait_count(msi_bits *p, uint64_t seq, uint64_t *out) {
  const auto t0 = std::chrono::steady_clock::now();
  for (uint32_t spin = 0;; ++spin) {
    if (__atomic_load_n(const_cast<uint64_t *>(&p->h_sig[1]), __ATOMIC_ACQUIRE) == seq) break;
    if ((spin & 1023) == 1023 &&
        std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() > 2000.0) {
      MSI_HIP_TRY(hipStreamSynchronize(p->stream));
      if (__atomic_load_n(const_cast<uint64_t *>(&p->h_sig[1]), __ATOMIC_ACQUIRE) != seq) {
        msi_set_error("count kernel finished without publishing its result");
        return MSI_E_INTERNAL;
      }
      break;
    }
  }
  *out = __atomic_load_n(const_cast<uint64_t *>(&p->h_sig[0]), __ATOMIC_RELAXED);
  return MSI_OK;
}

int32_t check_slot(const msi_bits *p, uint32_t s, const char *what) {
  if (!p || s >= p->n_slots) {
    msi_set_error("%s: slot %u out of range", what, s);
    return MSI_E_INVALID;
  }
  const_cast<msi_bits *>(p)->sum_dirty = true;   // a direct operation: the command lists' chunk summaries may no longer hold
  return MSI_OK;
}

}  // namespace

extern "C" {

int32_t msi_bits_create(msi_ctx *ctx, uint64_t n_docs, uint32_t n_slots, msi_bits **out) {
  if (!ctx || !out || n_slots == 0) {
    msi_set_error("msi_bits_create: invalid argument");
    return MSI_E_INVALID;
  }
  *out = nullptr;
  DeviceGuard g(ctx->device);
  msi_bits *p = new msi_bits();
  p->ctx = ctx;
  p->stream = ctx->stream;
  p->mu = &ctx->mu;
  p->n_docs = n_docs;
  p->n_words = std::max<uint64_t>(2, ((n_docs + 127) / 128) * 2);
  p->n_slots = n_slots;
  int32_t s = p->pool.ensure((size_t)p->n_words * n_slots * sizeof(u64));
  const uint64_t n_chunks = (p->n_words + 1023) / 1024;
  if (s == MSI_OK && n_slots <= 1024) s = p->summary.ensure((size_t)n_chunks * 16 * sizeof(u64));
  if (s == MSI_OK) s = p->small.ensure(64);
  if (s == MSI_OK) {
    void *h = nullptr;
    if (hipHostMalloc(&h, (6 + MSI_BITS_MANY + MSI_BITS_MAX_PATHS) * sizeof(uint64_t), hipHostMallocCoherent) != hipSuccess ||
        hipMalloc((void **)&p->d_acc, (6 + MSI_BITS_MANY + MSI_BITS_MAX_PATHS) * sizeof(u64)) != hipSuccess ||
        hipMemset(p->d_acc, 0, (6 + MSI_BITS_MANY + MSI_BITS_MAX_PATHS) * sizeof(u64)) != hipSuccess) {
      msi_set_error("msi_bits_create: allocating the completion signal failed");
      s = MSI_E_OOM;
    } else {
      p->h_sig = (volatile uint64_t *)h;
      p->h_sig[0] = p->h_sig[1] = 0;
      // the two cells of the GeoSort kernels rest at ~0 ("no point seen")
      if (hipMemset(p->d_acc + 4 + MSI_BITS_MANY + MSI_BITS_MAX_PATHS, 0xFF, 2 * sizeof(u64)) != hipSuccess ||
          hipStreamSynchronize(nullptr) != hipSuccess)  // the pool's streams do not order with the null stream's memsets
        s = MSI_E_HIP;
    }
  }
  if (s != MSI_OK) {
    p->pool.release();
    p->small.release();
    p->summary.release();
    if (p->h_sig) (void)hipHostFree((void *)p->h_sig);
    if (p->d_acc) (void)hipFree(p->d_acc);
    delete p;
    return s;
  }
  std::lock_guard<std::mutex> lk(ctx->mu);
  hipError_t e = hipMemsetAsync(p->pool.p, 0, (size_t)p->n_words * n_slots * sizeof(u64), ctx->stream);
  // every slot starts all zero, and so do the summaries ("0 = this chunk is empty")
  if (e == hipSuccess && p->summary.p) e = hipMemsetAsync(p->summary.p, 0, (size_t)n_chunks * 16 * sizeof(u64), ctx->stream);
  if (e != hipSuccess) {
    msi_set_error("hipMemsetAsync failed: %s", hipGetErrorString(e));
    p->pool.release();
    delete p;
    return MSI_E_HIP;
  }
  msi_ctx_retain(ctx);
  *out = p;
  return MSI_OK;
}

int32_t msi_bits_use_private_stream(msi_bits *p) {
  if (!p) return MSI_E_INVALID;
  if (p->private_stream) return MSI_OK;
  DeviceGuard g(p->ctx->device);
  {
    std::lock_guard<std::mutex> lk(*p->mu);
    MSI_HIP_TRY(hipStreamSynchronize(p->stream));  // the creation memset ran on the context stream
  }
  hipStream_t st = nullptr;
  MSI_HIP_TRY(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  p->stream = st;
  p->mu = &p->own_mu;
  p->private_stream = true;
  return MSI_OK;
}

void msi_bits_destroy(msi_bits *p) {
  if (!p) return;
  msi_ctx *ctx = p->ctx;
  {
  std::lock_guard<std::mutex> lk(ctx->mu);
  DeviceGuard g(ctx->device);
  (void)hipStreamSynchronize(p->stream);
  p->pool.release();
  p->tmp.release();
  p->small.release();
  p->summary.release();
  p->stage.release();
  p->desc.release();
  p->small_ids.release();
  p->dv_first.release();
  p->dv_taken.release();
  if (p->h_sig) (void)hipHostFree((void *)p->h_sig);
  if (p->d_acc) (void)hipFree(p->d_acc);
  if (p->h_ring) (void)hipHostFree(p->h_ring);
  if (p->vm_block) (void)hipHostFree(p->vm_block);
  if (p->vm_stage) (void)hipHostFree(p->vm_stage);
  if (p->private_stream) (void)hipStreamDestroy(p->stream);
  p->caux.release();
  }
  msi_bits *companion = p->companion;
  delete p;
  if (companion) msi_bits_destroy(companion);
  msi_ctx_release(ctx);
}

int32_t msi_bits_fill(msi_bits *p, uint32_t slot, int32_t ones) {
  MSI_TRY(check_slot(p, slot, "msi_bits_fill"));
  std::lock_guard<std::mutex> lk(*p->mu);
  DeviceGuard g(p->ctx->device);
  hipLaunchKernelGGL(bits_fill_kernel, dim3((uint32_t)((p->n_words + BT - 1) / BT)), dim3(BT), 0,
                     p->stream, p->slot(slot), p->n_words, p->n_docs, ones);
  MSI_HIP_TRY(hipGetLastError());
  return MSI_OK;
}

int32_t msi_bits_set_from_docids(msi_bits *p, uint32_t slot, const uint32_t *docids, uint64_t n) {
  MSI_TRY(check_slot(p, slot, "msi_bits_set_from_docids"));
  if (n && !docids) return MSI_E_INVALID;
  std::lock_guard<std::mutex> lk(*p->mu);
  DeviceGuard g(p->ctx->device);
  hipStream_t st = p->stream;
  MSI_HIP_TRY(hipMemsetAsync(p->slot(slot), 0, p->n_words * sizeof(u64), st));
  if (n) {
    MSI_TRY(p->stage.ensure(n * sizeof(uint32_t)));
    MSI_HIP_TRY(hipMemcpyAsync(p->stage.p, docids, n * sizeof(uint32_t), hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(bits_set_docids_kernel, dim3((uint32_t)((n + BT - 1) / BT)), dim3(BT), 0, st,
                       p->slot(slot), p->n_docs, p->stage.as<uint32_t>(), n);
    MSI_HIP_TRY(hipGetLastError());
    MSI_HIP_TRY(hipStreamSynchronize(st));  // docids is borrowed only for the call
  }
  return MSI_OK;
}

int32_t msi_bits_set_from_docid_lists_device(msi_bits *p, uint32_t first_slot, uint32_t slot_stride,
                                             const uint32_t *d_docids, uint32_t list_stride, const uint32_t *d_counts,
                                             uint32_t n_lists) {
  if (!p || !n_lists || !slot_stride || !d_docids || !d_counts || !list_stride) return MSI_E_INVALID;
  MSI_TRY(check_slot(p, first_slot, "msi_bits_set_from_docid_lists_device"));
  MSI_TRY(check_slot(p, first_slot + (n_lists - 1) * slot_stride, "msi_bits_set_from_docid_lists_device"));
  std::lock_guard<std::mutex> lk(*p->mu);
  DeviceGuard g(p->ctx->device);
  hipStream_t st = p->stream;
  const uint32_t gx = (uint32_t)std::min<uint64_t>(std::max<uint64_t>(1, (p->n_words / 2 + BT - 1) / BT), 64);
  hipLaunchKernelGGL(bits_clear_slots_kernel, dim3(gx, n_lists), dim3(BT), 0, st, p->pool.as<u64>(), p->n_words,
                     first_slot, slot_stride);
  hipLaunchKernelGGL(bits_set_lists_kernel, dim3((list_stride + BT - 1) / BT, n_lists), dim3(BT), 0, st,
                     p->pool.as<u64>(), p->n_words, p->n_docs, first_slot, slot_stride, d_docids, list_stride, d_counts);
  MSI_HIP_TRY(hipGetLastError());
  return MSI_OK;
}

// internal: slot |= the docids of a device-resident ascending list (the items of a vector store)
int32_t msi_bits_or_docids_device(msi_bits *p, uint32_t slot, const uint32_t *d_docids, uint64_t n) {
  MSI_TRY(check_slot(p, slot, "msi_bits_or_docids_device"));
  if (!n) return MSI_OK;
  if (!d_docids) return MSI_E_INVALID;
  std::lock_guard<std::mutex> lk(*p->mu);
  DeviceGuard g(p->ctx->device);
  const uint32_t grid = (uint32_t)std::min<uint64_t>((n + BT - 1) / BT, (uint64_t)std::max(1, p->ctx->n_cu) * 8);
  hipLaunchKernelGGL(bits_or_docids_kernel, dim3(grid), dim3(BT), 0, p->stream, p->slot(slot), p->n_docs, d_docids, n);
  MSI_HIP_TRY(hipGetLastError());
  return MSI_OK;
}

// `_vectors.{embedder}[.fragments.{f} | .userProvided | .documentTemplate | .regenerate]` for ONE embedder
// (search/facet/filter/vector.rs:78-158, evaluate_inner): the items of the embedder's stores as a docid set, minus the
// index's user_provided / skip_regenerate bitmaps as the variant asks.
int32_t msi_bits_vector_filter(msi_bits *p, uint32_t dst, int32_t kind, int32_t embedder_has_fragments,
                               msi_vs *const *stores, uint32_t n_stores, msi_bq *const *bq_stores, uint32_t n_bq_stores,
                               uint32_t user_provided, uint32_t skip_regenerate, uint32_t scratch, int32_t accumulate) {
  if (!p || kind < MSI_VECTOR_FILTER_NONE || kind > MSI_VECTOR_FILTER_REGENERATE || (n_stores && !stores) ||
      (n_bq_stores && !bq_stores)) {
    msi_set_error("msi_bits_vector_filter: invalid argument");
    return MSI_E_INVALID;
  }
  MSI_TRY(check_slot(p, dst, "msi_bits_vector_filter"));
  MSI_TRY(check_slot(p, scratch, "msi_bits_vector_filter"));
  if (scratch == dst) {
    msi_set_error("msi_bits_vector_filter: scratch must differ from dst");
    return MSI_E_INVALID;
  }
  const bool needs_up = kind == MSI_VECTOR_FILTER_FRAGMENT || kind == MSI_VECTOR_FILTER_DOCUMENT_TEMPLATE ||
                        kind == MSI_VECTOR_FILTER_USER_PROVIDED;
  if (needs_up) MSI_TRY(check_slot(p, user_provided, "msi_bits_vector_filter (user_provided)"));
  if (kind == MSI_VECTOR_FILTER_REGENERATE) MSI_TRY(check_slot(p, skip_regenerate, "msi_bits_vector_filter (skip_regenerate)"));
  if (kind == MSI_VECTOR_FILTER_USER_PROVIDED) {   // vector.rs:136-139
    if (accumulate) return msi_bits_op(p, dst, dst, user_provided, MSI_BITS_OR);
    return msi_bits_op(p, dst, user_provided, user_provided, MSI_BITS_OR);
  }
  MSI_TRY(msi_bits_fill(p, scratch, 0));
  // DocumentTemplate on an embedder that has fragments selects nothing (vector.rs:126-129)
  if (!(kind == MSI_VECTOR_FILTER_DOCUMENT_TEMPLATE && embedder_has_fragments)) {
    for (uint32_t i = 0; i < n_stores; ++i) MSI_TRY(msi_vs_items_bits(stores[i], p, scratch));      // stats.documents /
    for (uint32_t i = 0; i < n_bq_stores; ++i) MSI_TRY(msi_bq_items_bits(bq_stores[i], p, scratch));  // items_in_store
    if (kind == MSI_VECTOR_FILTER_FRAGMENT || kind == MSI_VECTOR_FILTER_DOCUMENT_TEMPLATE)
      MSI_TRY(msi_bits_op(p, scratch, scratch, user_provided, MSI_BITS_ANDNOT));                      // :121-124, :131-134
    else if (kind == MSI_VECTOR_FILTER_REGENERATE)
      MSI_TRY(msi_bits_op(p, scratch, scratch, skip_regenerate, MSI_BITS_ANDNOT));                    // :140-145
  }
  if (accumulate) return msi_bits_op(p, dst, dst, scratch, MSI_BITS_OR);
  return msi_bits_op(p, dst, scratch, scratch, MSI_BITS_OR);
}

int32_t msi_bits_set_from_words(msi_bits *p, uint32_t slot, const uint64_t *words, uint64_t n_words) {
  MSI_TRY(check_slot(p, slot, "msi_bits_set_from_words"));
  if (n_words > p->n_words || (n_words && !words)) {
    msi_set_error("msi_bits_set_from_words: %llu words > slot size %llu", (unsigned long long)n_words,
                  (unsigned long long)p->n_words);
    return MSI_E_INVALID;
  }
  std::lock_guard<std::mutex> lk(*p->mu);
  DeviceGuard g(p->ctx->device);
  hipStream_t st = p->stream;
  MSI_HIP_TRY(hipMemsetAsync(p->slot(slot), 0, p->n_words * sizeof(u64), st));
  if (n_words) MSI_HIP_TRY(hipMemcpyAsync(p->slot(slot), words, n_words * sizeof(u64), hipMemcpyHostToDevice, st));
  MSI_HIP_TRY(hipStreamSynchronize(st));
  return MSI_OK;
}

}  // extern "C"

// ---- batched posting-list decode (shared with msi_keyword.hip) ----------------------

// Host-side parse of one CboRoaringBitmapCodec value
// (cbo_roaring_bitmap_codec.rs:53-69): <= 7 integers are raw native-endian u32s,
// otherwise the portable Roaring serialisation (cookies 12346 / 12347), whose
// containers are appended to the batch with offsets into the batch buffer.
// Returns false on a malformed value.
// The container table of one serialisation (offsets relative to its first byte); false: malformed or a raw small value.
bool msi_cbo_parse(const uint8_t *bytes, size_t len, std::vector<MsiContainer> &out) {
  if (len <= 7 * sizeof(uint32_t)) return false;
  MsiCboBatch tmp;
  // (parsed through the batch path with a pretended cache source, so that nothing is copied and offsets are relative)
  if (!msi_cbo_batch_append(tmp, bytes, len, 0, MSI_NO_CACHE)) return false;
  out.assign(tmp.containers.begin(), tmp.containers.end());
  return !out.empty();
}

// A posting the cache already knows (msi_pcache_known, kind 3): its containers join the batch without its bytes.
void msi_cbo_batch_append_known(MsiCboBatch &batch, const MsiContainer *conts, uint32_t n, uint64_t cache_off) {
  const size_t first = batch.containers.size();
  batch.containers.insert(batch.containers.end(), conts, conts + n);
  batch.src.resize(first + n, MSI_NO_CACHE);
  batch.fill.resize(first + n, MSI_NO_CACHE);
  for (uint32_t i = 0; i < n; ++i) batch.src[first + i] = cache_off + conts[i].offset;
}

bool msi_cbo_batch_append(MsiCboBatch &batch, const uint8_t *bytes, size_t len, uint64_t cache_src, uint64_t cache_fill) {
  const size_t THRESHOLD = 7;  // cbo_roaring_bitmap_codec.rs:15
  if (len <= THRESHOLD * sizeof(uint32_t)) {
    // a trailing partial integer is ignored (read_u32 fails)
    for (size_t i = 0; i + 4 <= len; i += 4) {
      uint32_t v;
      memcpy(&v, bytes + i, 4);
      batch.small_ids.push_back(v);
    }
    return true;
  }
  auto rd16 = [&](size_t o) -> uint32_t { return (uint32_t)bytes[o] | ((uint32_t)bytes[o + 1] << 8); };
  auto rd32 = [&](size_t o) -> uint32_t { return rd16(o) | (rd16(o + 2) << 16); };
  size_t pos = 0;
  if (len < 8) return false;
  const uint32_t cookie = rd32(0);
  uint32_t n_cont = 0;
  bool has_runs = false;
  const uint8_t *run_flags = nullptr;
  if ((cookie & 0xFFFF) == 12347) {
    has_runs = true;
    n_cont = (cookie >> 16) + 1;
    pos = 4;
    run_flags = bytes + pos;
    pos += (n_cont + 7) / 8;
  } else if (cookie == 12346) {
    n_cont = rd32(4);
    pos = 8;
  } else {
    return false;
  }
  if (n_cont > 65536 || pos + (size_t)n_cont * 4 > len) return false;
  // every serialisation starts 16-byte aligned, in the staging buffer as in the posting cache: a body then has the
  // same alignment skew in both, and the cache is filled with whole 16-byte blocks
  const bool cached = cache_src != MSI_NO_CACHE;
  if (!cached) batch.bytes.resize((batch.bytes.size() + 15) & ~(size_t)15, 0);
  const size_t base = cached ? 0 : batch.bytes.size();
  const size_t first = batch.containers.size();
  batch.containers.resize(first + n_cont);
  MsiContainer *cs = batch.containers.data() + first;
  for (uint32_t i = 0; i < n_cont; ++i) {
    cs[i].key = rd16(pos + 4 * i);
    cs[i].card = rd16(pos + 4 * i + 2) + 1;
    const bool is_run = has_runs && ((run_flags[i / 8] >> (i % 8)) & 1);
    cs[i].type = is_run ? 2 : (cs[i].card > 4096 ? 1 : 0);
  }
  pos += (size_t)n_cont * 4;
  if (!has_runs || n_cont >= 4) pos += (size_t)n_cont * 4;  // offset header (recomputed below)
  bool ok = true;
  for (uint32_t i = 0; i < n_cont && ok; ++i) {
    if (pos > len) { ok = false; break; }
    cs[i].offset = (uint32_t)(base + pos);
    if (cs[i].type == 0) pos += (size_t)cs[i].card * 2;
    else if (cs[i].type == 1) pos += 8192;
    else {
      if (pos + 2 > len) { ok = false; break; }
      const uint32_t n_runs = rd16(pos);
      cs[i].offset = (uint32_t)(base + pos + 2);
      cs[i].card = n_runs;
      pos += 2 + (size_t)n_runs * 4;
    }
  }
  if (!ok || pos > len || base + len > 0xFFFFFFFFull) {
    batch.containers.resize(first);
    return false;
  }
  if (cached || cache_fill != MSI_NO_CACHE || !batch.src.empty()) {
    batch.src.resize(first + n_cont, MSI_NO_CACHE);
    batch.fill.resize(first + n_cont, MSI_NO_CACHE);
    for (uint32_t i = 0; i < n_cont; ++i) {
      const uint64_t rel = cs[i].offset - base;   // of the body inside its serialisation
      if (cached) batch.src[first + i] = cache_src + rel;
      else if (cache_fill != MSI_NO_CACHE) batch.fill[first + i] = cache_fill + rel;
    }
  }
  if (!cached) batch.bytes.insert(batch.bytes.end(), bytes, bytes + len);
  return true;
}

// Number of documents of a CboRoaringBitmap value without decoding the bodies
// (the descriptive header stores cardinality - 1 per container).
uint64_t msi_cbo_cardinality(const uint8_t *bytes, size_t len) {
  if (len <= 7 * sizeof(uint32_t)) return len / 4;
  auto rd16 = [&](size_t o) -> uint32_t { return (uint32_t)bytes[o] | ((uint32_t)bytes[o + 1] << 8); };
  auto rd32 = [&](size_t o) -> uint32_t { return rd16(o) | (rd16(o + 2) << 16); };
  if (len < 8) return 0;
  const uint32_t cookie = rd32(0);
  uint32_t n_cont = 0;
  size_t pos = 0;
  if ((cookie & 0xFFFF) == 12347) {
    n_cont = (cookie >> 16) + 1;
    pos = 4 + (n_cont + 7) / 8;
  } else if (cookie == 12346) {
    n_cont = rd32(4);
    pos = 8;
  } else {
    return 0;
  }
  if (n_cont > 65536 || pos + (size_t)n_cont * 4 > len) return 0;
  uint64_t c = 0;
  for (uint32_t i = 0; i < n_cont; ++i) c += rd16(pos + 4 * i + 2) + 1;
  return c;
}

// slot := (clear ? {} : slot) ∪ every value appended to the batch.  Takes the context lock.
int32_t msi_bits_and_many_count(msi_bits *p, uint32_t prefix, uint32_t n, const uint32_t *cond, const uint32_t *dst,
                                uint64_t *counts) {
  if (!p || !n || n > MSI_BITS_MANY || !cond || !dst || !counts) return MSI_E_INVALID;
  MSI_TRY(check_slot(p, prefix, "msi_bits_and_many_count"));
  ManyArgs a;
  for (uint32_t k = 0; k < n; ++k) {
    MSI_TRY(check_slot(p, cond[k], "msi_bits_and_many_count"));
    MSI_TRY(check_slot(p, dst[k], "msi_bits_and_many_count"));
    a.cond[k] = p->slot(cond[k]);
    a.dst[k] = p->slot(dst[k]);
  }
  std::lock_guard<std::recursive_mutex> wl(p->wait_mu);
  std::unique_lock<std::mutex> lk(*p->mu);
  DeviceGuard g(p->ctx->device);
  const uint64_t n_pairs = p->n_words / 2;
  const uint64_t seq = ++p->seq;
  hipLaunchKernelGGL(bits_and_many_kernel, dim3(grid_for(n_pairs, (uint32_t)p->ctx->n_cu * 4)), dim3(BT), 0,
                     p->stream, a, p->slot(prefix), n, n_pairs, p->d_acc, p->h_sig, seq);
  MSI_HIP_TRY(hipGetLastError());
  lk.unlock();  // other pools that share the stream keep enqueueing while this one waits
  uint64_t ignored = 0;
  MSI_TRY(wait_count(p, seq, &ignored));
  for (uint32_t k = 0; k < n; ++k) counts[k] = __atomic_load_n(const_cast<uint64_t *>(&p->h_sig[2 + k]), __ATOMIC_RELAXED);
  return MSI_OK;
}

int32_t msi_bits_claim(msi_bits *p, uint32_t docs, uint32_t bucket, uint32_t universe, uint32_t n_stack,
                       const uint32_t *stack) {
  if (!p || n_stack > MSI_BITS_MANY || (n_stack && !stack)) return MSI_E_INVALID;
  MSI_TRY(check_slot(p, docs, "msi_bits_claim"));
  MSI_TRY(check_slot(p, bucket, "msi_bits_claim"));
  MSI_TRY(check_slot(p, universe, "msi_bits_claim"));
  ManyArgs a;
  for (uint32_t k = 0; k < n_stack; ++k) {
    MSI_TRY(check_slot(p, stack[k], "msi_bits_claim"));
    a.stack[k] = p->slot(stack[k]);
  }
  std::lock_guard<std::mutex> lk(*p->mu);
  DeviceGuard g(p->ctx->device);
  const uint64_t n_pairs = p->n_words / 2;
  hipLaunchKernelGGL(bits_claim_kernel, dim3(grid_for(n_pairs, (uint32_t)p->ctx->n_cu * 4)), dim3(BT), 0,
                     p->stream, a, p->slot(docs), p->slot(bucket), p->slot(universe), n_stack, n_pairs);
  MSI_HIP_TRY(hipGetLastError());
  return MSI_OK;
}

// A region of the pinned staging ring (copies from it are truly asynchronous; the stream is only
// waited for when the ring wraps, so a run of decodes is enqueued without a single synchronisation).
static int32_t ring_alloc(msi_bits *p, size_t bytes, uint8_t **out) {
  bytes = (bytes + 255) & ~(size_t)255;
  if (bytes > p->ring_cap) {
    MSI_HIP_TRY(hipStreamSynchronize(p->stream));
    if (p->h_ring) (void)hipHostFree(p->h_ring);
    p->h_ring = nullptr;
    p->ring_cap = 0;
    size_t cap = std::max<size_t>(bytes * 2, (size_t)8 << 20);
    void *h = nullptr;
    if (hipHostMalloc(&h, cap, hipHostMallocDefault) != hipSuccess) {
      msi_set_error("hipHostMalloc(%zu) for the posting staging ring failed", cap);
      return MSI_E_OOM;
    }
    p->h_ring = (uint8_t *)h;
    p->ring_cap = cap;
    p->ring_pos = 0;
  }
  if (p->ring_pos + bytes > p->ring_cap) {
    MSI_HIP_TRY(hipStreamSynchronize(p->stream));
    p->ring_pos = 0;
  }
  *out = p->h_ring + p->ring_pos;
  p->ring_pos += bytes;
  return MSI_OK;
}

int32_t msi_bits_clear_slots(msi_bits *p, uint32_t n, const uint32_t *slots) {
  if (!p || !n || n > MSI_BITS_CLEAR_MAX || !slots) return MSI_E_INVALID;
  ClearArgs a;
  for (uint32_t k = 0; k < n; ++k) {
    MSI_TRY(check_slot(p, slots[k], "msi_bits_clear_slots"));
    a.slot[k] = p->slot(slots[k]);
  }
  std::lock_guard<std::mutex> lk(*p->mu);
  DeviceGuard g(p->ctx->device);
  const uint64_t n_pairs = p->n_words / 2;
  const uint32_t gx = (uint32_t)std::min<uint64_t>(std::max<uint64_t>(1, (n_pairs + BT - 1) / BT), 32);
  hipLaunchKernelGGL(bits_clear_many_kernel, dim3(gx, n), dim3(BT), 0, p->stream, a, n_pairs);
  MSI_HIP_TRY(hipGetLastError());
  return MSI_OK;
}

// Launches bits_paths_small_kernel for one level if it fits (<= 64 paths, <= 448 steps, <= 32 distinct conditions);
// the counts go to region `region` of the pinned signal area.  Caller holds the pool lock.  MSI_E_UNSUPPORTED = does
// not fit.
static int32_t launch_paths_small(msi_bits *p, uint32_t n_paths, const uint32_t *path_off, const uint32_t *step_slots,
                                  uint32_t bucket, uint32_t universe, uint32_t region) {
  const uint32_t n_steps = path_off[n_paths];
  if (n_paths > PS_PATHS || n_steps > PS_STEPS) return MSI_E_UNSUPPORTED;
  PathsSmall a;
  uint32_t slots[PS_CONDS], n_conds = 0;
  for (uint32_t s = 0; s < n_steps; ++s) {
    uint32_t c = 0;
    while (c < n_conds && slots[c] != step_slots[s]) ++c;
    if (c == n_conds) {
      if (n_conds == PS_CONDS) return MSI_E_UNSUPPORTED;
      slots[n_conds] = step_slots[s];
      a.cond[n_conds++] = p->slot(step_slots[s]);
    }
    a.step[s] = (uint8_t)c;
  }
  for (uint32_t k = 0; k <= n_paths; ++k) a.off[k] = (uint16_t)path_off[k];
  const uint64_t n_pairs = p->n_words / 2;
  const uint64_t seq = ++p->seq;
  hipLaunchKernelGGL(bits_paths_small_kernel, dim3((uint32_t)((n_pairs + PS_T - 1) / PS_T)), dim3(PS_T), 0, p->stream, a,
                     n_conds, n_paths, p->slot(bucket), p->slot(universe), n_pairs, p->d_acc + 2 + MSI_BITS_MANY, p->d_acc,
                     p->h_sig + 2 + MSI_BITS_MANY + (size_t)region * MSI_BITS_REGION_PATHS, p->h_sig, seq);
  MSI_HIP_TRY(hipGetLastError());
  return MSI_OK;
}

int32_t msi_bits_paths_enqueue(msi_bits *p, uint32_t n_paths, const uint32_t *path_off, const uint32_t *step_slots,
                               uint32_t bucket, uint32_t universe, uint32_t region) {
  if (!p || !n_paths || !path_off || region >= MSI_BITS_PATH_REGIONS || n_paths > MSI_BITS_REGION_PATHS) return MSI_E_INVALID;
  const uint32_t n_steps = path_off[n_paths];
  if (n_steps && !step_slots) return MSI_E_INVALID;
  MSI_TRY(check_slot(p, bucket, "msi_bits_paths_enqueue"));
  MSI_TRY(check_slot(p, universe, "msi_bits_paths_enqueue"));
  for (uint32_t s = 0; s < n_steps; ++s) MSI_TRY(check_slot(p, step_slots[s], "msi_bits_paths_enqueue"));
  std::lock_guard<std::mutex> lk(*p->mu);
  DeviceGuard g(p->ctx->device);
  return launch_paths_small(p, n_paths, path_off, step_slots, bucket, universe, region);
}

int32_t msi_bits_paths_collect(msi_bits *p, uint32_t n_regions, uint64_t *counts) {
  if (!p || !n_regions || n_regions > MSI_BITS_PATH_REGIONS || !counts) return MSI_E_INVALID;
  std::lock_guard<std::recursive_mutex> wl(p->wait_mu);
  uint64_t seq;
  {
    std::lock_guard<std::mutex> lk(*p->mu);
    seq = p->seq;  // the last level enqueued on this pool; in-stream order makes the earlier ones complete too
  }
  uint64_t ignored = 0;
  MSI_TRY(wait_count(p, seq, &ignored));
  for (uint32_t k = 0; k < n_regions * MSI_BITS_REGION_PATHS; ++k)
    counts[k] = __atomic_load_n(const_cast<uint64_t *>(&p->h_sig[2 + MSI_BITS_MANY + k]), __ATOMIC_RELAXED);
  return MSI_OK;
}

int32_t msi_bits_paths_claim(msi_bits *p, uint32_t n_paths, const uint32_t *path_off, const uint32_t *step_slots,
                             uint32_t bucket, uint32_t universe, uint64_t *counts) {
  if (!p || !n_paths || n_paths > MSI_BITS_MAX_PATHS || !path_off || !counts) return MSI_E_INVALID;
  const uint32_t n_steps = path_off[n_paths];
  if (n_steps > MSI_BITS_MAX_STEPS || (n_steps && !step_slots)) return MSI_E_INVALID;
  MSI_TRY(check_slot(p, bucket, "msi_bits_paths_claim"));
  MSI_TRY(check_slot(p, universe, "msi_bits_paths_claim"));
  for (uint32_t s = 0; s < n_steps; ++s) MSI_TRY(check_slot(p, step_slots[s], "msi_bits_paths_claim"));
  std::lock_guard<std::recursive_mutex> wl(p->wait_mu);
  std::unique_lock<std::mutex> lk(*p->mu);
  DeviceGuard g(p->ctx->device);
  hipStream_t st = p->stream;
  {
    const int32_t st_small = launch_paths_small(p, n_paths, path_off, step_slots, bucket, universe, 0);
    if (st_small == MSI_OK) {
      const uint64_t seq = p->seq;
      lk.unlock();
      uint64_t ignored = 0;
      MSI_TRY(wait_count(p, seq, &ignored));
      for (uint32_t k = 0; k < n_paths; ++k)
        counts[k] = __atomic_load_n(const_cast<uint64_t *>(&p->h_sig[2 + MSI_BITS_MANY + k]), __ATOMIC_RELAXED);
      return MSI_OK;
    }
    if (st_small != MSI_E_UNSUPPORTED) return st_small;
  }
  const size_t steps_bytes = std::max<size_t>(1, n_steps) * sizeof(u64 *), off_bytes = (n_paths + 1) * sizeof(uint32_t);
  if (steps_bytes + off_bytes > p->desc.cap) MSI_TRY(p->desc.ensure(std::max<size_t>(steps_bytes + off_bytes, (size_t)64 << 10)));
  uint8_t *h = nullptr;
  MSI_TRY(ring_alloc(p, steps_bytes + off_bytes, &h));
  u64 **hs = (u64 **)h;
  for (uint32_t s = 0; s < n_steps; ++s) hs[s] = p->slot(step_slots[s]);
  memcpy(h + steps_bytes, path_off, off_bytes);
  MSI_HIP_TRY(hipMemcpyAsync(p->desc.p, h, steps_bytes + off_bytes, hipMemcpyHostToDevice, st));
  const uint64_t n_pairs = p->n_words / 2;
  const uint64_t seq = ++p->seq;
  hipLaunchKernelGGL(bits_paths_kernel, dim3(grid_for(n_pairs, (uint32_t)p->ctx->n_cu * 4)), dim3(BT), 0, st,
                     (const u64 *const *)p->desc.p, (const uint32_t *)((uint8_t *)p->desc.p + steps_bytes), n_paths,
                     p->slot(bucket), p->slot(universe), n_pairs, p->d_acc + 2 + MSI_BITS_MANY, p->d_acc,
                     p->h_sig + 2 + MSI_BITS_MANY, p->h_sig, seq);
  MSI_HIP_TRY(hipGetLastError());
  lk.unlock();
  uint64_t ignored = 0;
  MSI_TRY(wait_count(p, seq, &ignored));
  for (uint32_t k = 0; k < n_paths; ++k)
    counts[k] = __atomic_load_n(const_cast<uint64_t *>(&p->h_sig[2 + MSI_BITS_MANY + k]), __ATOMIC_RELAXED);
  return MSI_OK;
}

int32_t msi_bits_decode_batch(msi_bits *p, uint32_t slot, const MsiCboBatch &batch, bool clear) {
  std::lock_guard<std::mutex> lk(*p->mu);
  DeviceGuard g(p->ctx->device);
  hipStream_t st = p->stream;
  if (clear) MSI_HIP_TRY(hipMemsetAsync(p->slot(slot), 0, p->n_words * sizeof(u64), st));
  const size_t n_cont = batch.containers.size();
  if (n_cont) {
    if (batch.bytes.size() > p->stage.cap)
      MSI_TRY(p->stage.ensure(std::max<size_t>({batch.bytes.size(), p->stage.cap * 2, (size_t)1 << 20})));
    if (n_cont * sizeof(Container) > p->desc.cap)
      MSI_TRY(p->desc.ensure(std::max<size_t>({n_cont * sizeof(Container), p->desc.cap * 2, (size_t)64 << 10})));
    // ONE ring block for both pieces: a second ring_alloc may wrap onto (or, growing the ring, free) the first
    // block before its bytes have been copied
    const size_t bytes_al = (batch.bytes.size() + 255) & ~(size_t)255;
    uint8_t *hb = nullptr;
    MSI_TRY(ring_alloc(p, bytes_al + n_cont * sizeof(Container), &hb));
    uint8_t *hc = hb + bytes_al;
    memcpy(hb, batch.bytes.data(), batch.bytes.size());
    memcpy(hc, batch.containers.data(), n_cont * sizeof(Container));
    MSI_HIP_TRY(hipMemcpyAsync(p->stage.p, hb, batch.bytes.size(), hipMemcpyHostToDevice, st));
    MSI_HIP_TRY(hipMemcpyAsync(p->desc.p, hc, n_cont * sizeof(Container), hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(bits_decode_roaring_kernel, dim3((uint32_t)n_cont), dim3(BT), 0, st, p->slot(slot),
                       p->n_docs, p->stage.as<uint8_t>(), p->desc.as<Container>());
    MSI_HIP_TRY(hipGetLastError());
  }
  if (!batch.small_ids.empty()) {
    const size_t n = batch.small_ids.size();
    MSI_TRY(p->small_ids.ensure(n * sizeof(uint32_t)));
    uint8_t *hs = nullptr;
    MSI_TRY(ring_alloc(p, n * sizeof(uint32_t), &hs));
    memcpy(hs, batch.small_ids.data(), n * sizeof(uint32_t));
    MSI_HIP_TRY(hipMemcpyAsync(p->small_ids.p, hs, n * sizeof(uint32_t), hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(bits_set_docids_kernel, dim3((uint32_t)((n + BT - 1) / BT)), dim3(BT), 0, st, p->slot(slot),
                       p->n_docs, p->small_ids.as<uint32_t>(), n);
    MSI_HIP_TRY(hipGetLastError());
  }
  return MSI_OK;   // in-stream order makes the slot valid for every later operation of this pool
}

extern "C" {

// CboRoaringBitmapCodec::deserialize_from (cbo_roaring_bitmap_codec.rs:53-69).
int32_t msi_bits_set_from_cbo(msi_bits *p, uint32_t slot, const uint8_t *bytes, size_t len) {
  MSI_TRY(check_slot(p, slot, "msi_bits_set_from_cbo"));
  if (len && !bytes) return MSI_E_INVALID;
  MsiCboBatch batch;
  if (!msi_cbo_batch_append(batch, bytes, len)) {
    msi_set_error("msi_bits_set_from_cbo: malformed Roaring serialisation (%zu bytes)", len);
    return MSI_E_INVALID;
  }
  return msi_bits_decode_batch(p, slot, batch, true);
}

int32_t msi_bits_op(msi_bits *p, uint32_t dst, uint32_t a, uint32_t b, int32_t op) {
  MSI_TRY(check_slot(p, dst, "msi_bits_op"));
  MSI_TRY(check_slot(p, a, "msi_bits_op"));
  MSI_TRY(check_slot(p, b, "msi_bits_op"));
  std::lock_guard<std::mutex> lk(*p->mu);
  DeviceGuard g(p->ctx->device);
  const uint64_t n_pairs = p->n_words / 2;
  const dim3 grid(grid_for(n_pairs, (uint32_t)p->ctx->n_cu * 8)), block(BT);
  hipStream_t st = p->stream;
  switch (op) {
    case MSI_BITS_AND:
      hipLaunchKernelGGL(bits_op_kernel<MSI_BITS_AND>, grid, block, 0, st, p->slot(dst), p->slot(a), p->slot(b), n_pairs);
      break;
    case MSI_BITS_OR:
      hipLaunchKernelGGL(bits_op_kernel<MSI_BITS_OR>, grid, block, 0, st, p->slot(dst), p->slot(a), p->slot(b), n_pairs);
      break;
    case MSI_BITS_ANDNOT:
      hipLaunchKernelGGL(bits_op_kernel<MSI_BITS_ANDNOT>, grid, block, 0, st, p->slot(dst), p->slot(a), p->slot(b), n_pairs);
      break;
    case MSI_BITS_XOR:
      hipLaunchKernelGGL(bits_op_kernel<MSI_BITS_XOR>, grid, block, 0, st, p->slot(dst), p->slot(a), p->slot(b), n_pairs);
      break;
    default:
      msi_set_error("msi_bits_op: unknown op %d", op);
      return MSI_E_INVALID;
  }
  MSI_HIP_TRY(hipGetLastError());
  return MSI_OK;
}

int32_t msi_bits_op_count(msi_bits *p, uint32_t dst, uint32_t a, uint32_t b, int32_t op, uint64_t *out_count) {
  MSI_TRY(check_slot(p, dst, "msi_bits_op_count"));
  MSI_TRY(check_slot(p, a, "msi_bits_op_count"));
  MSI_TRY(check_slot(p, b, "msi_bits_op_count"));
  if (!out_count) return MSI_E_INVALID;
  std::lock_guard<std::recursive_mutex> wl(p->wait_mu);
  std::unique_lock<std::mutex> lk(*p->mu);
  DeviceGuard g(p->ctx->device);
  const uint64_t n_pairs = p->n_words / 2;
  const dim3 grid(grid_for(n_pairs, (uint32_t)p->ctx->n_cu * 8)), block(BT);
  hipStream_t st = p->stream;
  const uint64_t seq = ++p->seq;
  u64 *acc = p->d_acc;
  volatile uint64_t *sig = p->h_sig;
  switch (op) {
    case MSI_BITS_AND:
      hipLaunchKernelGGL(bits_op_count_kernel<MSI_BITS_AND>, grid, block, 0, st, p->slot(dst), p->slot(a), p->slot(b), n_pairs, acc, sig, seq);
      break;
    case MSI_BITS_OR:
      hipLaunchKernelGGL(bits_op_count_kernel<MSI_BITS_OR>, grid, block, 0, st, p->slot(dst), p->slot(a), p->slot(b), n_pairs, acc, sig, seq);
      break;
    case MSI_BITS_ANDNOT:
      hipLaunchKernelGGL(bits_op_count_kernel<MSI_BITS_ANDNOT>, grid, block, 0, st, p->slot(dst), p->slot(a), p->slot(b), n_pairs, acc, sig, seq);
      break;
    case MSI_BITS_XOR:
      hipLaunchKernelGGL(bits_op_count_kernel<MSI_BITS_XOR>, grid, block, 0, st, p->slot(dst), p->slot(a), p->slot(b), n_pairs, acc, sig, seq);
      break;
    default:
      msi_set_error("msi_bits_op_count: unknown op %d", op);
      return MSI_E_INVALID;
  }
  MSI_HIP_TRY(hipGetLastError());
  lk.unlock();
  return wait_count(p, seq, out_count);
}

int32_t msi_bits_union_many_and(msi_bits *p, uint32_t dst, const uint32_t *srcs, uint32_t n, uint32_t universe) {
  MSI_TRY(check_slot(p, dst, "msi_bits_union_many_and"));
  if (universe != 0xFFFFFFFFu) MSI_TRY(check_slot(p, universe, "msi_bits_union_many_and"));
  for (uint32_t i = 0; i < n; ++i) MSI_TRY(check_slot(p, srcs[i], "msi_bits_union_many_and"));
  std::lock_guard<std::mutex> lk(*p->mu);
  DeviceGuard g(p->ctx->device);
  hipStream_t st = p->stream;
  MSI_TRY(p->desc.ensure(std::max<uint32_t>(1, n) * sizeof(uint32_t)));
  if (n) MSI_HIP_TRY(hipMemcpyAsync(p->desc.p, srcs, n * sizeof(uint32_t), hipMemcpyHostToDevice, st));
  hipLaunchKernelGGL(bits_union_many_kernel, dim3(grid_for(p->n_words / 2, (uint32_t)p->ctx->n_cu * 8)), dim3(BT),
                     0, st, p->pool.as<u64>(), p->n_words, dst, p->desc.as<uint32_t>(), n, universe);
  MSI_HIP_TRY(hipGetLastError());
  MSI_HIP_TRY(hipStreamSynchronize(st));  // srcs is borrowed; desc is reused
  return MSI_OK;
}

int32_t msi_doc_keys_create(msi_ctx *ctx, const uint32_t *keys, uint64_t n_docs, msi_doc_keys **out) {
  if (!ctx || !out || !n_docs || !keys) {
    msi_set_error("msi_doc_keys_create: invalid argument");
    return MSI_E_INVALID;
  }
  *out = nullptr;
  DeviceGuard g(ctx->device);
  msi_doc_keys *k = new msi_doc_keys();
  k->ctx = ctx;
  k->n_docs = n_docs;
  int32_t s = k->keys.ensure((size_t)n_docs * sizeof(uint32_t));
  if (s == MSI_OK) {
    std::lock_guard<std::mutex> lk(ctx->mu);
    hipError_t e = hipMemcpyAsync(k->keys.p, keys, (size_t)n_docs * sizeof(uint32_t), hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);  // `keys` is borrowed for the call
    if (e != hipSuccess) {
      msi_set_error("msi_doc_keys_create: upload failed: %s", hipGetErrorString(e));
      s = MSI_E_HIP;
    }
  }
  if (s != MSI_OK) {
    k->keys.release();
    delete k;
    return s;
  }
  msi_ctx_retain(ctx);
  *out = k;
  return MSI_OK;
}

void msi_doc_keys_destroy(msi_doc_keys *k) {
  if (!k) return;
  {
    DeviceGuard g(k->ctx->device);
    k->keys.release();
  }
  msi_ctx_release(k->ctx);
  delete k;
}

int32_t msi_bits_order_next(msi_bits *p, const msi_doc_keys *keys, uint32_t universe, uint32_t bucket, uint32_t *out_key,
                            uint64_t *out_count) {
  if (!p || !keys || !out_key || !out_count || universe == bucket) {
    msi_set_error("msi_bits_order_next: invalid argument");
    return MSI_E_INVALID;
  }
  if (keys->ctx != p->ctx || keys->n_docs != p->n_docs) {
    msi_set_error("msi_bits_order_next: the key array (%llu documents) does not belong to this pool (%llu documents)",
                  (unsigned long long)keys->n_docs, (unsigned long long)p->n_docs);
    return MSI_E_INVALID;
  }
  MSI_TRY(check_slot(p, universe, "msi_bits_order_next"));
  MSI_TRY(check_slot(p, bucket, "msi_bits_order_next"));
  std::lock_guard<std::recursive_mutex> wl(p->wait_mu);
  std::unique_lock<std::mutex> lk(*p->mu);
  DeviceGuard g(p->ctx->device);
  hipStream_t st = p->stream;
  u64 *best = p->d_acc + 2 + MSI_BITS_MANY + MSI_BITS_MAX_PATHS;
  // grid-stride kernels: a few workgroups per CU, whole 64-document words per wave
  const uint32_t blocks = (uint32_t)std::min<uint64_t>((p->n_words * 64 + BT - 1) / BT, (uint64_t)p->ctx->n_cu * 4);
  const uint64_t seq = ++p->seq;
  hipLaunchKernelGGL(bits_min_key_kernel, dim3(blocks), dim3(BT), 0, st, p->slot(universe), keys->keys.as<uint32_t>(),
                     p->n_docs, best);
  hipLaunchKernelGGL(bits_take_key_kernel, dim3(blocks), dim3(BT), 0, st, p->slot(universe), p->slot(bucket),
                     keys->keys.as<uint32_t>(), p->n_docs, p->n_words, best, p->d_acc, p->h_sig, seq);
  MSI_HIP_TRY(hipGetLastError());
  lk.unlock();
  MSI_TRY(wait_count(p, seq, out_count));
  *out_key = (uint32_t)__atomic_load_n(const_cast<uint64_t *>(&p->h_sig[2]), __ATOMIC_RELAXED);
  return MSI_OK;
}

int32_t msi_doc_values_create(msi_ctx *ctx, const uint64_t *offsets, const uint32_t *value_ids, uint64_t n_docs,
                              uint32_t n_values, msi_doc_values **out) {
  if (!ctx || !out || !n_docs || !offsets || offsets[0] != 0 || (offsets[n_docs] && !value_ids)) {
    msi_set_error("msi_doc_values_create: invalid argument");
    return MSI_E_INVALID;
  }
  *out = nullptr;
  const uint64_t total = offsets[n_docs];
  if (total >= 0xFFFFFFFFull) {
    msi_set_error("msi_doc_values_create: %llu (document, value) pairs; the device layout holds < 2^32",
                  (unsigned long long)total);
    return MSI_E_UNSUPPORTED;
  }
  std::vector<uint32_t> off32(n_docs + 1);
  for (uint64_t d = 0; d <= n_docs; ++d) {
    if (d && offsets[d] < offsets[d - 1]) {
      msi_set_error("msi_doc_values_create: offsets decrease at document %llu", (unsigned long long)d);
      return MSI_E_INVALID;
    }
    off32[d] = (uint32_t)offsets[d];
  }
  for (uint64_t i = 0; i < total; ++i)
    if (value_ids[i] >= n_values) {
      msi_set_error("msi_doc_values_create: value id %u >= n_values %u", value_ids[i], n_values);
      return MSI_E_INVALID;
    }
  DeviceGuard g(ctx->device);
  msi_doc_values *v = new msi_doc_values();
  v->ctx = ctx;
  v->n_docs = n_docs;
  v->n_values = n_values;
  int32_t st = v->offsets.ensure((size_t)(n_docs + 1) * sizeof(uint32_t));
  if (st == MSI_OK) st = v->values.ensure(std::max<size_t>(1, (size_t)total) * sizeof(uint32_t));
  if (st == MSI_OK) {
    std::lock_guard<std::mutex> lk(ctx->mu);
    hipError_t e = hipMemcpyAsync(v->offsets.p, off32.data(), (size_t)(n_docs + 1) * sizeof(uint32_t), hipMemcpyHostToDevice,
                                  ctx->stream);
    if (e == hipSuccess && total)
      e = hipMemcpyAsync(v->values.p, value_ids, (size_t)total * sizeof(uint32_t), hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);  // the arrays are borrowed for the call
    if (e != hipSuccess) {
      msi_set_error("msi_doc_values_create: upload failed: %s", hipGetErrorString(e));
      st = MSI_E_HIP;
    }
  }
  if (st != MSI_OK) {
    v->offsets.release();
    v->values.release();
    delete v;
    return st;
  }
  msi_ctx_retain(ctx);
  *out = v;
  return MSI_OK;
}

void msi_doc_values_destroy(msi_doc_values *v) {
  if (!v) return;
  {
    DeviceGuard g(v->ctx->device);
    v->offsets.release();
    v->values.release();
  }
  msi_ctx_release(v->ctx);
  delete v;
}

// The per-pool scratch of the distinct kernels, sized for `vals`; hands out the stamps of a new call.  Pool lock held.
static int32_t distinct_scratch(msi_bits *p, const msi_doc_values *vals, uint32_t rounds_needed) {
  hipStream_t st = p->stream;
  const uint32_t need = std::max<uint32_t>(1, vals->n_values);
  if (need > p->dv_cap) {
    MSI_HIP_TRY(hipStreamSynchronize(st));  // kernels in flight may still read the old buffers
    MSI_TRY(p->dv_first.ensure((size_t)need * sizeof(u64)));
    MSI_TRY(p->dv_taken.ensure((size_t)need * sizeof(uint32_t)));
    p->dv_cap = need;
    p->dv_round = 0;
    p->dv_call = 0;
  }
  if (p->dv_round == 0 || p->dv_round > 0xFFFFFFFFu - rounds_needed - 2) {
    MSI_HIP_TRY(hipMemsetAsync(p->dv_first.p, 0xFF, (size_t)p->dv_cap * sizeof(u64), st));
    p->dv_round = 1;
  }
  if (p->dv_call == 0 || p->dv_call == 0xFFFFFFFEu) {
    MSI_HIP_TRY(hipMemsetAsync(p->dv_taken.p, 0, (size_t)p->dv_cap * sizeof(uint32_t), st));
    p->dv_call = 0;
  }
  ++p->dv_call;
  return MSI_OK;
}

static int32_t check_values(const msi_bits *p, const msi_doc_values *vals, const char *what) {
  if (!p || !vals) {
    msi_set_error("%s: invalid argument", what);
    return MSI_E_INVALID;
  }
  if (vals->ctx != p->ctx || vals->n_docs != p->n_docs) {
    msi_set_error("%s: the value table (%llu documents) does not belong to this pool (%llu documents)", what,
                  (unsigned long long)vals->n_docs, (unsigned long long)p->n_docs);
    return MSI_E_INVALID;
  }
  return MSI_OK;
}

int32_t msi_bits_distinct(msi_bits *p, const msi_doc_values *vals, uint32_t candidates, uint32_t remaining,
                          uint32_t excluded, uint64_t *out_remaining, uint32_t *out_rounds) {
  MSI_TRY(check_values(p, vals, "msi_bits_distinct"));
  if (!out_remaining || candidates == remaining || candidates == excluded || remaining == excluded) {
    msi_set_error("msi_bits_distinct: invalid argument (three different slots)");
    return MSI_E_INVALID;
  }
  MSI_TRY(check_slot(p, candidates, "msi_bits_distinct"));
  MSI_TRY(check_slot(p, remaining, "msi_bits_distinct"));
  if (excluded != MSI_BITS_NO_SLOT) MSI_TRY(check_slot(p, excluded, "msi_bits_distinct"));
  std::lock_guard<std::recursive_mutex> wl(p->wait_mu);
  std::unique_lock<std::mutex> lk(*p->mu);
  DeviceGuard g(p->ctx->device);
  hipStream_t st = p->stream;
  MSI_TRY(distinct_scratch(p, vals, MSI_DISTINCT_MAX_ROUNDS + 1));
  const uint32_t call = p->dv_call;
  const uint32_t *off = vals->offsets.as<uint32_t>(), *val = vals->values.as<uint32_t>();
  u64 *first = p->dv_first.as<u64>();
  uint32_t *taken = p->dv_taken.as<uint32_t>();
  u64 *kept_acc = p->d_acc + 3 + MSI_BITS_MANY + MSI_BITS_MAX_PATHS;
  const dim3 grid((uint32_t)((p->n_words * 64 + BT - 1) / BT)), block(BT);
  uint32_t round = p->dv_round++;
  hipLaunchKernelGGL(bits_distinct_propose_kernel, grid, block, 0, st, p->slot(candidates), off, val, p->n_docs, p->n_words,
                     first, taken, call, round, 1, kept_acc, p->d_acc, p->h_sig, ++p->seq);
  uint32_t rounds = 0;
  uint64_t left = 0, kept = 0;
  for (;;) {
    hipLaunchKernelGGL(bits_distinct_join_kernel, grid, block, 0, st, p->slot(candidates), p->slot(remaining), off, val,
                       p->n_docs, p->n_words, first, taken, call, round, rounds == 0 ? 1 : 0, kept_acc);
    round = p->dv_round++;
    const uint64_t seq = ++p->seq;
    hipLaunchKernelGGL(bits_distinct_propose_kernel, grid, block, 0, st, p->slot(candidates), off, val, p->n_docs,
                       p->n_words, first, taken, call, round, 0, kept_acc, p->d_acc, p->h_sig, seq);
    MSI_HIP_TRY(hipGetLastError());
    ++rounds;
    lk.unlock();
    MSI_TRY(wait_count(p, seq, &left));
    kept = __atomic_load_n(const_cast<uint64_t *>(&p->h_sig[2]), __ATOMIC_RELAXED);
    lk.lock();
    if (!left) break;
    if (rounds >= MSI_DISTINCT_MAX_ROUNDS) {
      const uint64_t seq2 = ++p->seq;
      hipLaunchKernelGGL(bits_distinct_sequential_kernel, dim3(1), dim3(64), 0, st, p->slot(candidates), p->slot(remaining),
                         off, val, p->n_words, taken, call, kept_acc, p->h_sig, seq2);
      MSI_HIP_TRY(hipGetLastError());
      lk.unlock();
      MSI_TRY(wait_count(p, seq2, &left));
      kept = __atomic_load_n(const_cast<uint64_t *>(&p->h_sig[2]), __ATOMIC_RELAXED);
      lk.lock();
      rounds |= 0x80000000u;
      break;
    }
  }
  if (excluded != MSI_BITS_NO_SLOT) {
    hipLaunchKernelGGL(bits_distinct_excluded_kernel, grid, block, 0, st, p->slot(excluded), off, val, p->n_docs, p->n_words,
                       taken, call);
    MSI_HIP_TRY(hipGetLastError());
  }
  *out_remaining = kept;
  if (out_rounds) *out_rounds = rounds;
  return MSI_OK;
}

int32_t msi_bits_distinct_excluded(msi_bits *p, const msi_doc_values *vals, uint32_t kept, uint32_t excluded) {
  MSI_TRY(check_values(p, vals, "msi_bits_distinct_excluded"));
  if (kept == excluded) {
    msi_set_error("msi_bits_distinct_excluded: kept and excluded are the same slot");
    return MSI_E_INVALID;
  }
  MSI_TRY(check_slot(p, kept, "msi_bits_distinct_excluded"));
  MSI_TRY(check_slot(p, excluded, "msi_bits_distinct_excluded"));
  std::lock_guard<std::mutex> lk(*p->mu);
  DeviceGuard g(p->ctx->device);
  hipStream_t st = p->stream;
  MSI_TRY(distinct_scratch(p, vals, 0));
  const dim3 grid((uint32_t)((p->n_words * 64 + BT - 1) / BT)), block(BT);
  hipLaunchKernelGGL(bits_distinct_mark_kernel, grid, block, 0, st, p->slot(kept), vals->offsets.as<uint32_t>(),
                     vals->values.as<uint32_t>(), p->n_docs, p->n_words, p->dv_taken.as<uint32_t>(), p->dv_call);
  hipLaunchKernelGGL(bits_distinct_excluded_kernel, grid, block, 0, st, p->slot(excluded), vals->offsets.as<uint32_t>(),
                     vals->values.as<uint32_t>(), p->n_docs, p->n_words, p->dv_taken.as<uint32_t>(), p->dv_call);
  MSI_HIP_TRY(hipGetLastError());
  return MSI_OK;
}

int32_t msi_bits_andnot_many_count(msi_bits *p, uint32_t removed, uint32_t n, const uint32_t *slots, uint64_t *counts) {
  if (!p || !n || n > MSI_BITS_MANY || !slots || !counts) return MSI_E_INVALID;
  MSI_TRY(check_slot(p, removed, "msi_bits_andnot_many_count"));
  ManyArgs a;
  for (uint32_t k = 0; k < n; ++k) {
    MSI_TRY(check_slot(p, slots[k], "msi_bits_andnot_many_count"));
    if (slots[k] == removed) {
      msi_set_error("msi_bits_andnot_many_count: `removed` is one of the slots");
      return MSI_E_INVALID;
    }
    a.dst[k] = p->slot(slots[k]);
  }
  std::lock_guard<std::recursive_mutex> wl(p->wait_mu);
  std::unique_lock<std::mutex> lk(*p->mu);
  DeviceGuard g(p->ctx->device);
  const uint64_t n_pairs = p->n_words / 2;
  const uint64_t seq = ++p->seq;
  hipLaunchKernelGGL(bits_andnot_many_kernel, dim3(grid_for(n_pairs, (uint32_t)p->ctx->n_cu * 4)), dim3(BT), 0, p->stream, a,
                     p->slot(removed), n, n_pairs, p->d_acc, p->h_sig, seq);
  MSI_HIP_TRY(hipGetLastError());
  lk.unlock();
  uint64_t ignored = 0;
  MSI_TRY(wait_count(p, seq, &ignored));
  for (uint32_t k = 0; k < n; ++k) counts[k] = __atomic_load_n(const_cast<uint64_t *>(&p->h_sig[2 + k]), __ATOMIC_RELAXED);
  return MSI_OK;
}

int32_t msi_geo_points_create(msi_ctx *ctx, const double *lat_lng, uint64_t n_docs, msi_geo_points **out) {
  if (!ctx || !out || !n_docs || !lat_lng) {
    msi_set_error("msi_geo_points_create: invalid argument");
    return MSI_E_INVALID;
  }
  *out = nullptr;
  DeviceGuard g(ctx->device);
  msi_geo_points *gp = new msi_geo_points();
  gp->ctx = ctx;
  gp->n_docs = n_docs;
  int32_t st = gp->lat_lng.ensure((size_t)n_docs * 2 * sizeof(double));
  if (st == MSI_OK) {
    std::lock_guard<std::mutex> lk(ctx->mu);
    hipError_t e = hipMemcpyAsync(gp->lat_lng.p, lat_lng, (size_t)n_docs * 2 * sizeof(double), hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);  // borrowed for the call
    if (e != hipSuccess) {
      msi_set_error("msi_geo_points_create: upload failed: %s", hipGetErrorString(e));
      st = MSI_E_HIP;
    }
  }
  if (st != MSI_OK) {
    gp->lat_lng.release();
    delete gp;
    return st;
  }
  msi_ctx_retain(ctx);
  *out = gp;
  return MSI_OK;
}

void msi_geo_points_destroy(msi_geo_points *gp) {
  if (!gp) return;
  {
    DeviceGuard g(gp->ctx->device);
    gp->lat_lng.release();
  }
  msi_ctx_release(gp->ctx);
  delete gp;
}

// one launch of the take kernel in a range mode (0: dst := selection, 2: dst |= selection) -> |selection|
static int32_t geo_range(msi_bits *p, const msi_geo_points *gp, const GeoTarget &t, uint32_t src, uint32_t dst, int mode,
                         u64 key_lo, u64 key_hi, uint64_t *count) {
  std::lock_guard<std::recursive_mutex> wl(p->wait_mu);
  std::unique_lock<std::mutex> lk(*p->mu);
  DeviceGuard g(p->ctx->device);
  u64 *cells = p->d_acc + 4 + MSI_BITS_MANY + MSI_BITS_MAX_PATHS;
  const dim3 grid((uint32_t)std::min<uint64_t>((p->n_words * 64 + BT - 1) / BT, (uint64_t)p->ctx->n_cu * 4)), block(BT);
  const uint64_t seq = ++p->seq;
  hipLaunchKernelGGL(bits_geo_take_kernel, grid, block, 0, p->stream, p->slot(src), p->slot(dst), gp->lat_lng.as<double>(),
                     p->n_docs, p->n_words, t, mode, key_lo, key_hi, cells, cells + 1, p->d_acc, p->h_sig, seq);
  MSI_HIP_TRY(hipGetLastError());
  lk.unlock();
  return wait_count(p, seq, count);
}

int32_t msi_bits_geo_list(msi_bits *p, const msi_geo_points *gp, uint32_t universe, double lat, double lng, uint32_t cap,
                          uint32_t *out_docids, double *out_distance, uint64_t *out_total) {
  if (!p || !gp || !out_total || (cap && (!out_docids || !out_distance))) {
    msi_set_error("msi_bits_geo_list: invalid argument");
    return MSI_E_INVALID;
  }
  if (gp->ctx != p->ctx || gp->n_docs != p->n_docs) {
    msi_set_error("msi_bits_geo_list: the points (%llu documents) do not belong to this pool (%llu documents)",
                  (unsigned long long)gp->n_docs, (unsigned long long)p->n_docs);
    return MSI_E_INVALID;
  }
  MSI_TRY(check_slot(p, universe, "msi_bits_geo_list"));
  const double D2R = 3.14159265358979323846 / 180.0;
  GeoTarget t;
  t.phi = lat * D2R;
  t.cos_phi = cos(t.phi);
  t.lam = lng * D2R;
  t.margin = 0.0;
  t.ascending = 1;
  std::lock_guard<std::mutex> lk(*p->mu);
  DeviceGuard g(p->ctx->device);
  hipStream_t st = p->stream;
  // scratch: [counter u64][ids u32 x cap][distances f64 x cap]
  const size_t ids_at = 16, dist_at = (ids_at + (size_t)cap * 4 + 15) & ~(size_t)15, total = dist_at + (size_t)cap * 8;
  MSI_TRY(p->tmp.ensure(std::max<size_t>(total, 64)));
  uint8_t *base = p->tmp.as<uint8_t>();
  MSI_HIP_TRY(hipMemsetAsync(base, 0, 16, st));
  const dim3 grid((uint32_t)std::min<uint64_t>((p->n_words * 64 + BT - 1) / BT, (uint64_t)p->ctx->n_cu * 4)), block(BT);
  hipLaunchKernelGGL(bits_geo_list_kernel, grid, block, 0, st, p->slot(universe), gp->lat_lng.as<double>(), p->n_docs, p->n_words,
                     t, cap, reinterpret_cast<uint32_t *>(base + ids_at), reinterpret_cast<double *>(base + dist_at),
                     reinterpret_cast<u64 *>(base));
  MSI_HIP_TRY(hipGetLastError());
  u64 n = 0;
  MSI_HIP_TRY(hipMemcpyAsync(&n, base, sizeof(n), hipMemcpyDeviceToHost, st));
  MSI_HIP_TRY(hipStreamSynchronize(st));
  *out_total = n;
  const size_t take = (size_t)std::min<u64>(n, cap);
  if (take) {
    MSI_HIP_TRY(hipMemcpyAsync(out_docids, base + ids_at, take * 4, hipMemcpyDeviceToHost, st));
    MSI_HIP_TRY(hipMemcpyAsync(out_distance, base + dist_at, take * 8, hipMemcpyDeviceToHost, st));
    MSI_HIP_TRY(hipStreamSynchronize(st));
  }
  return MSI_OK;
}

int32_t msi_bits_geo_next(msi_bits *p, const msi_geo_points *gp, uint32_t universe, uint32_t bucket, uint32_t scratch,
                          double lat, double lng, int32_t ascending, uint32_t max_bucket_size, double margin,
                          uint32_t *out_first_docid, uint64_t *out_count) {
  if (!p || !gp || !out_first_docid || !out_count || universe == bucket || universe == scratch || bucket == scratch ||
      !(margin >= 0.0)) {
    msi_set_error("msi_bits_geo_next: invalid argument (three different slots, a margin >= 0)");
    return MSI_E_INVALID;
  }
  if (gp->ctx != p->ctx || gp->n_docs != p->n_docs) {
    msi_set_error("msi_bits_geo_next: the points (%llu documents) do not belong to this pool (%llu documents)",
                  (unsigned long long)gp->n_docs, (unsigned long long)p->n_docs);
    return MSI_E_INVALID;
  }
  MSI_TRY(check_slot(p, universe, "msi_bits_geo_next"));
  MSI_TRY(check_slot(p, bucket, "msi_bits_geo_next"));
  MSI_TRY(check_slot(p, scratch, "msi_bits_geo_next"));
  const double D2R = 3.14159265358979323846 / 180.0;
  GeoTarget t;
  t.phi = lat * D2R;
  t.cos_phi = cos(t.phi);
  t.lam = lng * D2R;
  t.margin = margin;
  t.ascending = ascending ? 1 : 0;
  const uint64_t cap = max_bucket_size ? max_bucket_size : 1000;
  uint64_t count = 0, first = 0, kbest = 0;
  {
    std::lock_guard<std::recursive_mutex> wl(p->wait_mu);
    std::unique_lock<std::mutex> lk(*p->mu);
    DeviceGuard g(p->ctx->device);
    hipStream_t st = p->stream;
    u64 *cells = p->d_acc + 4 + MSI_BITS_MANY + MSI_BITS_MAX_PATHS;
    const dim3 grid((uint32_t)std::min<uint64_t>((p->n_words * 64 + BT - 1) / BT, (uint64_t)p->ctx->n_cu * 4)), block(BT);
    const uint64_t seq = ++p->seq;
    hipLaunchKernelGGL(bits_geo_min_kernel, grid, block, 0, st, p->slot(universe), gp->lat_lng.as<double>(), p->n_docs,
                       p->n_words, t, cells);
    hipLaunchKernelGGL(bits_geo_take_kernel, grid, block, 0, st, p->slot(universe), p->slot(bucket),
                       gp->lat_lng.as<double>(), p->n_docs, p->n_words, t, 1, (u64)0, (u64)0, cells, cells + 1, p->d_acc,
                       p->h_sig, seq);
    MSI_HIP_TRY(hipGetLastError());
    lk.unlock();
    MSI_TRY(wait_count(p, seq, &count));
    first = __atomic_load_n(const_cast<uint64_t *>(&p->h_sig[2]), __ATOMIC_RELAXED);
    kbest = __atomic_load_n(const_cast<uint64_t *>(&p->h_sig[3]), __ATOMIC_RELAXED);
  }
  if (kbest == ~0ull || !count) {  // no document of the universe has a point
    *out_first_docid = 0xFFFFFFFFu;
    *out_count = 0;
    return MSI_OK;
  }
  if (count > cap) {
    // More documents within the margin than a bucket may hold (documents/geo_sort.rs:190-193,203-205): keep the `cap`
    // first in (distance key, docid) order.  T = the smallest key with |{key <= T}| >= cap, by bisection over the
    // key bits; everything below T, then the smallest docids at T; the rest goes back to the universe.
    double d0;
    {
      const u64 b0 = ascending ? kbest : ~(kbest + 1);
      memcpy(&d0, &b0, sizeof(d0));
    }
    const double edge = ascending ? d0 + margin : std::max(0.0, d0 - margin);  // every kept key lies between the two
    u64 edge_bits;
    memcpy(&edge_bits, &edge, sizeof(edge));
    u64 lo = kbest, hi = ascending ? edge_bits : ~edge_bits - 1;
    while (lo < hi) {
      const u64 mid = lo + (hi - lo) / 2;
      uint64_t c = 0;
      MSI_TRY(geo_range(p, gp, t, bucket, scratch, 0, 0, mid, &c));
      if (c >= cap) hi = mid;
      else lo = mid + 1;
    }
    uint64_t below = 0, at = 0;
    if (lo > 0) MSI_TRY(geo_range(p, gp, t, bucket, scratch, 0, 0, lo - 1, &below));
    MSI_TRY(geo_range(p, gp, t, bucket, scratch, 0, lo, lo, &at));
    std::vector<uint32_t> ids((size_t)(cap - below));
    uint32_t n_ids = 0;
    MSI_TRY(msi_bits_first_k(p, scratch, (uint32_t)ids.size(), ids.data(), &n_ids));
    MSI_TRY(msi_bits_set_from_docids(p, scratch, ids.data(), n_ids));
    if (lo > 0) MSI_TRY(geo_range(p, gp, t, bucket, scratch, 2, 0, lo - 1, &below));
    MSI_TRY(msi_bits_op(p, bucket, bucket, scratch, MSI_BITS_ANDNOT));  // what does not fit
    MSI_TRY(msi_bits_op(p, universe, universe, bucket, MSI_BITS_OR));
    MSI_TRY(msi_bits_op(p, bucket, scratch, scratch, MSI_BITS_AND));
    count = cap;
  }
  *out_first_docid = (uint32_t)first;
  *out_count = count;
  return MSI_OK;
}

uint64_t msi_facet_number_key(double value) {
  // f64_into_bytes (heed_codec/facet/value_encoding.rs:5-20) stores -0.0 as +0.0: one key for both
  if (value == 0.0) value = 0.0;
  uint64_t b;
  memcpy(&b, &value, sizeof(b));
  return (b >> 63) ? ~b : (b | 0x8000000000000000ull);  // the order of the doubles (OrderedF64Codec sorts the same way)
}

int32_t msi_facet_keys_create(msi_ctx *ctx, const uint64_t *offsets, const uint64_t *keys, uint64_t n_docs,
                              msi_facet_keys **out) {
  if (!ctx || !out || !n_docs || !offsets || offsets[0] != 0 || (offsets[n_docs] && !keys)) {
    msi_set_error("msi_facet_keys_create: invalid argument");
    return MSI_E_INVALID;
  }
  *out = nullptr;
  const uint64_t total = offsets[n_docs];
  if (total >= 0xFFFFFFFFull) {
    msi_set_error("msi_facet_keys_create: %llu (document, value) pairs; the device layout holds < 2^32",
                  (unsigned long long)total);
    return MSI_E_UNSUPPORTED;
  }
  std::vector<uint32_t> off32(n_docs + 1);
  for (uint64_t d = 0; d <= n_docs; ++d) {
    if (d && offsets[d] < offsets[d - 1]) {
      msi_set_error("msi_facet_keys_create: offsets decrease at document %llu", (unsigned long long)d);
      return MSI_E_INVALID;
    }
    off32[d] = (uint32_t)offsets[d];
  }
  DeviceGuard g(ctx->device);
  msi_facet_keys *f = new msi_facet_keys();
  f->ctx = ctx;
  f->n_docs = n_docs;
  int32_t st = f->offsets.ensure((size_t)(n_docs + 1) * sizeof(uint32_t));
  if (st == MSI_OK) st = f->keys.ensure(std::max<size_t>(1, (size_t)total) * sizeof(u64));
  if (st == MSI_OK) {
    std::lock_guard<std::mutex> lk(ctx->mu);
    hipError_t e = hipMemcpyAsync(f->offsets.p, off32.data(), (size_t)(n_docs + 1) * sizeof(uint32_t), hipMemcpyHostToDevice,
                                  ctx->stream);
    if (e == hipSuccess && total)
      e = hipMemcpyAsync(f->keys.p, keys, (size_t)total * sizeof(u64), hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) {
      msi_set_error("msi_facet_keys_create: upload failed: %s", hipGetErrorString(e));
      st = MSI_E_HIP;
    }
  }
  if (st != MSI_OK) {
    f->offsets.release();
    f->keys.release();
    delete f;
    return st;
  }
  msi_ctx_retain(ctx);
  *out = f;
  return MSI_OK;
}

void msi_facet_keys_destroy(msi_facet_keys *f) {
  if (!f) return;
  {
    DeviceGuard g(f->ctx->device);
    f->offsets.release();
    f->keys.release();
  }
  msi_ctx_release(f->ctx);
  delete f;
}

static int32_t facet_select(msi_bits *p, const msi_facet_keys *f, u64 lo, u64 hi, const uint64_t *list, uint64_t n_list,
                            uint32_t dst, int32_t accumulate, const char *what) {
  if (!p || !f || (n_list && !list)) {
    msi_set_error("%s: invalid argument", what);
    return MSI_E_INVALID;
  }
  if (f->ctx != p->ctx || f->n_docs != p->n_docs) {
    msi_set_error("%s: the key table (%llu documents) does not belong to this pool (%llu documents)", what,
                  (unsigned long long)f->n_docs, (unsigned long long)p->n_docs);
    return MSI_E_INVALID;
  }
  MSI_TRY(check_slot(p, dst, what));
  for (uint64_t i = 1; i < n_list; ++i)
    if (list[i - 1] >= list[i]) {
      msi_set_error("%s: the key list must be strictly ascending", what);
      return MSI_E_NOT_SORTED;
    }
  std::lock_guard<std::mutex> lk(*p->mu);
  DeviceGuard g(p->ctx->device);
  hipStream_t st = p->stream;
  const u64 *d_list = nullptr;
  if (list) {
    // an empty list still selects through the list branch (nothing); the copy must be done before `list` goes away
    MSI_TRY(p->small_ids.ensure(std::max<uint64_t>(1, n_list) * sizeof(u64)));
    if (n_list) MSI_HIP_TRY(hipMemcpyAsync(p->small_ids.p, list, n_list * sizeof(u64), hipMemcpyHostToDevice, st));
    d_list = p->small_ids.as<u64>();
  }
  const dim3 grid((uint32_t)((p->n_words * 64 + BT - 1) / BT)), block(BT);
  hipLaunchKernelGGL(bits_facet_select_kernel, grid, block, 0, st, p->slot(dst), f->offsets.as<uint32_t>(), f->keys.as<u64>(),
                     p->n_docs, p->n_words, lo, hi, d_list, n_list, accumulate ? 1 : 0);
  MSI_HIP_TRY(hipGetLastError());
  if (list) MSI_HIP_TRY(hipStreamSynchronize(st));  // `list` is borrowed; small_ids is shared scratch
  return MSI_OK;
}

int32_t msi_bits_facet_range(msi_bits *p, const msi_facet_keys *f, uint64_t lo, uint64_t hi, uint32_t dst,
                             int32_t accumulate) {
  return facet_select(p, f, lo, hi, nullptr, 0, dst, accumulate, "msi_bits_facet_range");
}

int32_t msi_bits_facet_in(msi_bits *p, const msi_facet_keys *f, const uint64_t *sorted_keys, uint64_t n, uint32_t dst,
                          int32_t accumulate) {
  static const uint64_t none = 0;
  return facet_select(p, f, 1, 0, n ? sorted_keys : &none, n, dst, accumulate, "msi_bits_facet_in");
}

int32_t msi_bits_geo_within(msi_bits *p, const msi_geo_points *gp, uint32_t src, double lat, double lng, double radius_m,
                            uint32_t dst) {
  if (!p || !gp || src == dst || !(radius_m >= 0.0)) {
    msi_set_error("msi_bits_geo_within: invalid argument (two different slots, a radius >= 0)");
    return MSI_E_INVALID;
  }
  if (gp->ctx != p->ctx || gp->n_docs != p->n_docs) {
    msi_set_error("msi_bits_geo_within: the points (%llu documents) do not belong to this pool (%llu documents)",
                  (unsigned long long)gp->n_docs, (unsigned long long)p->n_docs);
    return MSI_E_INVALID;
  }
  MSI_TRY(check_slot(p, src, "msi_bits_geo_within"));
  MSI_TRY(check_slot(p, dst, "msi_bits_geo_within"));
  const double D2R = 3.14159265358979323846 / 180.0;
  GeoTarget t;
  t.phi = lat * D2R;
  t.cos_phi = cos(t.phi);
  t.lam = lng * D2R;
  t.margin = 0.0;
  t.ascending = 1;
  const double edge = radius_m + 2.220446049250313e-16;  // `<= radius + f64::EPSILON`, index_filter.rs:496-497
  u64 hi;
  memcpy(&hi, &edge, sizeof(hi));
  uint64_t ignored = 0;
  return geo_range(p, gp, t, src, dst, 0, 0, hi, &ignored);
}

int32_t msi_bits_count(msi_bits *p, uint32_t slot, uint64_t *out) {
  MSI_TRY(check_slot(p, slot, "msi_bits_count"));
  if (!out) return MSI_E_INVALID;
  std::lock_guard<std::recursive_mutex> wl(p->wait_mu);
  std::unique_lock<std::mutex> lk(*p->mu);
  DeviceGuard g(p->ctx->device);
  hipStream_t st = p->stream;
  const uint64_t seq = ++p->seq;
  hipLaunchKernelGGL(bits_count_kernel, dim3(grid_for(p->n_words, (uint32_t)p->ctx->n_cu * 8)), dim3(BT), 0, st,
                     p->slot(slot), p->n_words, p->d_acc, p->h_sig, seq);
  MSI_HIP_TRY(hipGetLastError());
  lk.unlock();
  return wait_count(p, seq, out);
}

int32_t msi_bits_first_k(msi_bits *p, uint32_t slot, uint32_t k, uint32_t *out_docids, uint32_t *out_n) {
  MSI_TRY(check_slot(p, slot, "msi_bits_first_k"));
  if (!out_n || (k && !out_docids)) return MSI_E_INVALID;
  std::lock_guard<std::mutex> lk(*p->mu);
  DeviceGuard g(p->ctx->device);
  hipStream_t st = p->stream;
  const uint32_t n_blocks = (uint32_t)((p->n_words + BT - 1) / BT);
  MSI_TRY(p->tmp.ensure(((size_t)n_blocks + std::max<uint32_t>(1, k)) * sizeof(uint32_t)));
  uint32_t *blk = p->tmp.as<uint32_t>();
  uint32_t *d_out = blk + n_blocks;
  hipLaunchKernelGGL(bits_block_counts_kernel, dim3(n_blocks), dim3(BT), 0, st, p->slot(slot), p->n_words, blk);
  hipLaunchKernelGGL(bits_scan_counts_kernel, dim3(1), dim3(BT), 0, st, blk, n_blocks, p->small.as<uint32_t>() + 4);
  if (k) hipLaunchKernelGGL(bits_emit_first_k_kernel, dim3(n_blocks), dim3(BT), 0, st, p->slot(slot), p->n_words, blk, k, d_out);
  MSI_HIP_TRY(hipGetLastError());
  uint32_t total = 0;
  MSI_HIP_TRY(hipMemcpyAsync(&total, p->small.as<uint32_t>() + 4, sizeof(uint32_t), hipMemcpyDeviceToHost, st));
  MSI_HIP_TRY(hipStreamSynchronize(st));
  const uint32_t n = std::min(total, k);
  if (n) MSI_HIP_TRY(hipMemcpy(out_docids, d_out, n * sizeof(uint32_t), hipMemcpyDeviceToHost));
  *out_n = n;
  return MSI_OK;
}

int32_t msi_bits_read_words(msi_bits *p, uint32_t slot, uint64_t *out_words) {
  MSI_TRY(check_slot(p, slot, "msi_bits_read_words"));
  if (!out_words) return MSI_E_INVALID;
  std::lock_guard<std::mutex> lk(*p->mu);
  DeviceGuard g(p->ctx->device);
  hipStream_t st = p->stream;
  const uint64_t words = (p->n_docs + 63) / 64;
  if (words) MSI_HIP_TRY(hipMemcpyAsync(out_words, p->slot(slot), words * sizeof(u64), hipMemcpyDeviceToHost, st));
  MSI_HIP_TRY(hipStreamSynchronize(st));
  return MSI_OK;
}

const uint64_t *msi_bits_device_ptr(msi_bits *p, uint32_t slot) {
  if (!p || slot >= p->n_slots) return nullptr;
  return (const uint64_t *)p->slot(slot);
}

}  // extern "C"
