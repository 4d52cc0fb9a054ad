// msi_vm.h — internal: command lists over the docid sets of a msi_bits pool, executed by ONE kernel launch per
// dependency round and SHARED by every keyword search in flight on the context (msi_vm.hip).
//
// Why: the ranked keyword search (msi_search.hip) is a host control flow that needs a cardinality every few set
// operations.  One launch per operation made a 3-term query 211 launches and 80 waits — launch-bound, ~1 % of the
// HBM roofline, slower than its CPU port.  Here a search only RECORDS its operations; at the
// point where it needs a result it submits the recorded list.  A combiner thread packs the lists of all the
// searches that are waiting at that moment into one arena (one H2D copy) and runs them with one launch: a
// workgroup owns one 65 536-document chunk (one Roaring container span, 8 KiB of a set) of one search and
// interprets that search's commands in order.  Every command is chunk-local (set algebra, fused path claims,
// container decode into LDS), so no workgroup ever waits for another; the last workgroup of a list publishes the
// cardinalities (and ordered docids) into the search's pinned result block.
#pragma once
#include <stdint.h>

#include <vector>

struct msi_bits;
struct MsiCboBatch;

enum : uint32_t {
  VM_END = 0,
  VM_FILL,       // dst, ones
  VM_OP,         // dst, a, b, op(MSI_BITS_*)
  VM_OP_COUNT,   // dst, a, b, op, cnt
  VM_CLEAR,      // n, slot...
  VM_CLAIM,      // docs, bucket, universe, n, stack...
  VM_AND_MANY,   // prefix, n, cnt_base, (cond, dst)...
  VM_PATHS,      // n_paths, bucket, universe, cnt_base, n_steps, off[n_paths + 1], step slot...
  VM_SUB_MANY,   // removed, n, cnt_base, slot...
  VM_COUNT,      // slot, cnt
  VM_DECODE,     // dst, overwrite, index of the decode among the list's decodes
  VM_FIRSTK,     // slot, k, cnt, ids base (the set's cardinality comes back too): the first k docids of the set, ascending,
                 // at u32 offset `ids base` of the list's id block; <= MSI_VM_MAX_FK_PHASE per phase, <= MSI_VM_MAX_FIRSTK ids per list
  VM_MINKEY,     // universe, keys lo, keys hi, cell
  VM_TAKEKEY,    // universe, bucket, keys lo, keys hi, cell, cnt, key result index
  VM_SUMMARY_RESET,   // (no operand) the pool's chunk summaries no longer hold: every bit back to "may hold documents"
  // ---- universe compaction (see MsiVmList::geom_docs) ------------------------------------------------------------
  VM_RANK_A,     // slot, aux lo, aux hi: this chunk's cardinality of the set -> aux.chunk_cnt[chunk]
  VM_RANK_B,     // slot, aux lo, aux hi, capacity (next phase): exclusive prefix popcount of every word -> aux.prefix[word],
                 // docid of every document by rank -> aux.c2d[rank] (ranks below the capacity)
  VM_DECODEC,    // dst (compact slot), source: index of the decode among the list's decodes | 0x80000000 + a slot of the FULL
                 // pool.  Only in the WIDE phase of a compact list (one workgroup per chunk of the full space): the
                 // documents of the source that are in U0, as ranks, into the part of dst that belongs to this chunk
};

constexpr uint32_t MSI_VM_MAX_COUNTS = 1024;   // cardinalities one list can ask for
constexpr uint32_t MSI_VM_MAX_PHASES = 4;      // kernel boundaries inside one list (Sort rule: min, then take)
constexpr uint32_t MSI_VM_MAX_FIRSTK = 8192;   // docids one list can read back
constexpr uint32_t MSI_VM_MAX_FK_PHASE = 16;   // first-k commands per phase
constexpr uint32_t MSI_VM_CELLS = 4;

struct MsiVmList {
  std::vector<uint32_t> words;          // commands of all phases, each phase terminated by VM_END
  std::vector<uint32_t> phase_start;    // word offset of each phase (phase_start[0] == 0 once anything is recorded)
  // decode payloads: written by the search thread straight into its pool's PINNED staging buffer and read by the
  // kernel over PCIe (wide aligned loads, each posting byte exactly once) — no copy on the combiner thread, no DMA
  // call per list.  `stage_used` bytes of msi_bits_vm_stage(pool) belong to this list.
  size_t stage_used = 0;
  // descriptors of the decode commands (chunk-bucketed while recording; laid out chunk-major behind the commands when
  // the list is submitted): start[c] .. start[c+1] = the containers of chunk c, two u64 per container
  struct Decode {
    std::vector<uint32_t> start;
    std::vector<uint64_t> c;
  };
  std::vector<Decode> decodes;
  uint32_t data_off = 0;                // word offset of the descriptor block inside `words` once finalised (0: none)
  uint64_t cache_base = 0;              // device address of the posting cache the decode commands refer to (0: none)
  uint32_t n_counts = 0;
  uint32_t firstk_total = 0;            // ids the list's first-k commands ask for (their blocks lie back to back)
  uint32_t fk_in_phase = 0, max_fk_phase = 1;
  // Universe compaction.  A list with geom_docs != 0 is a COMPACT list: it runs on the companion pool of `full_pool`, its
  // sets are geom_docs bits long (the ranks of the documents of U0 = slot `u0_slot` of the full pool; slots are
  // geom_words apart), first-k ids are translated back to docids, and the VM_DECODEC commands recorded into `pre` run
  // first, in a phase of their own with one workgroup per chunk of the FULL space (they read postings / full sets by
  // docid and write ranks).  Hoisting them is safe because their destinations are slots no earlier command of the same
  // list has touched (msi_search.hip keeps slots freed while a list is recorded out of circulation until it has run).
  uint64_t geom_docs = 0;
  msi_bits *full_pool = nullptr;
  uint32_t u0_slot = 0;
  std::vector<uint32_t> pre;
  bool pre_merged = false;
  bool any_fill = false;                // a decode of this list is the FIRST reader of its key: its decode phase stores the bodies
                                        // into the posting cache on its way (the by-rank phase only looks for them when told so)
  bool empty() const { return words.empty() && pre.empty(); }
  void clear() {
    words.clear();
    pre.clear();
    pre_merged = false;
    any_fill = false;
    phase_start.clear();
    stage_used = 0;
    cache_base = 0;
    decodes.clear();
    data_off = 0;
    n_counts = 0;
    firstk_total = 0;
    fk_in_phase = 0;
    max_fk_phase = 1;
  }
  void begin() { if (phase_start.empty()) phase_start.push_back(0); }
  void barrier() {                      // the commands recorded next run after a kernel boundary
    begin();
    words.push_back(VM_END);
    phase_start.push_back((uint32_t)words.size());
    fk_in_phase = 0;
  }
  uint32_t new_counts(uint32_t n) {
    const uint32_t b = n_counts;
    n_counts += n;
    return b;
  }
};

struct MsiVmResult {
  std::vector<uint64_t> counts;         // [n_counts]
  std::vector<uint32_t> firstk;         // the id blocks of the list's VM_FIRSTK commands
};

// Appends a VM_DECODE of `batch` into `dst` (overwrite = the slot's previous content is discarded).  In a compact list
// (l.geom_docs != 0; `pool` = the companion pool): a VM_DECODEC into l.pre, dst := the ranks of the batch's documents in U0.
int32_t msi_vm_record_decode(MsiVmList &l, msi_bits *pool, uint32_t dst, const MsiCboBatch &batch, bool overwrite);
// Compact list: dst := the ranks of the documents of slot `full_slot` of the full pool that are in U0 (into l.pre).
void msi_vm_record_compact(MsiVmList &l, uint32_t dst, uint32_t full_slot);
// Appends, to a list on the FULL pool, the two commands (a kernel boundary between them) that fill the pool's
// compaction tables for U0 = `slot`.
bool msi_vm_record_rank(MsiVmList &l, msi_bits *pool, uint32_t slot);   // false: the pool's rank tables could not be allocated
uint64_t msi_bits_compact_capacity(const msi_bits *p);
msi_bits *msi_bits_compact_pool(msi_bits *p);
uint32_t *msi_bits_compact_aux(msi_bits *p);
// Runs the list on the pool's context (blocks until its results are published).  Appends the decode descriptors to
// l.words (once): clear() the list before recording again.
int32_t msi_vm_run(msi_bits *pool, MsiVmList &l, MsiVmResult *res);
// Combiner statistics of the context the pool lives on: rounds launched, lists executed, then nanoseconds summed over
// lists: queued, packed, (per round) in the launch calls, from the launch calls until the waiter saw its result.
void msi_vm_stats(msi_bits *pool, uint64_t out[6]);

struct msi_vm;
void msi_vm_destroy(msi_vm *vm);

// MSI_SEARCH_CPU_PROFILE=1: where the HOST CPU of the keyword leg goes — thread CPU time (CLOCK_THREAD_CPUTIME_ID: a
// sleeping waiter costs nothing) summed process-wide: [0] searches, [1] whole searches, [2] inside msi_vm_run (submission,
// the poll / futex wait, the wake-up), [3] of it: finalising the list (hoisted decodes, descriptors, accounting),
// [4] typo derivations (the batched dictionary lookup's host side), [5] index callbacks + posting parsing, [6] the
// combiner threads, [7] lists.  msi_search_cpu_profile() hands the sums out (tools/ranked_bench prints them per query).
bool msi_cpu_prof_on();
uint64_t msi_thread_cpu_ns();
void msi_cpu_prof_add(int idx, uint64_t ns);

// ---- HBM posting cache --------------------------------------------------------------------------------------------
// The stored CboRoaringBitmap bytes of the postings a search reads (word_docids, word_fid_docids, word_position_docids,
// word_pair_proximity_docids, ...), kept in HBM under a 128-bit hash of (database, key).  A search that reads a key
// for the first time decodes it out of its pinned staging buffer (the bytes cross PCIe once) and the decoding
// workgroups store the bodies into the cache on the way; every later search — any thread — decodes the same key
// straight from HBM: no host copy, no PCIe.  Owned by the dictionary handle (one per index version: both are
// re-staged when Index::updated_at moves), msi_dict_enable_posting_cache.
struct MsiPostingCache;
struct MsiCacheKey {
  uint64_t a, b;
};
// `view`: msi_search_params::index_view — which view of the index the value was read through (0: the index as it is)
MsiCacheKey msi_cache_key(uint32_t db, const void *s1, size_t n1, const void *s2, size_t n2, uint64_t x, uint64_t y,
                          uint64_t view = 0);
MsiPostingCache *msi_pcache_create(msi_ctx *ctx, uint64_t capacity_bytes);
void msi_pcache_destroy(MsiPostingCache *c);
// -> 1: ready in the cache at *off (decode from there); 2: reserved at *off for THIS caller to fill (pass *token to
// msi_pcache_commit once the list that fills it has run); 0: not cacheable now (being filled by someone else, or full)
int msi_pcache_lookup(MsiPostingCache *c, const MsiCacheKey &k, size_t len, uint64_t *off, void **token);
void msi_pcache_commit(MsiPostingCache *c, void *token);
void msi_pcache_abandon(MsiPostingCache *c, void *token);   // the filling list failed / was dropped: the next reader refills
uint64_t msi_pcache_device_base(const MsiPostingCache *c);
// What the cache KNOWS about a key without asking the index again (valid for the index version the cache belongs to):
// absent from the database | a raw value of <= 7 docids (kept on the host) | a serialisation whose body is ready in HBM,
// with its parsed container table.  A warm search then makes no index callback and parses no posting header at all.
struct MsiContainer;
struct MsiKnownPosting {
  int kind;                      // 1 absent, 2 small ids, 3 body in HBM
  uint64_t off, len, card;       // kind 3: byte offset in the cache, serialised length; cardinality (kinds 2, 3)
  const MsiContainer *conts;     // kind 3: offsets relative to the serialisation
  uint32_t n_conts;
  const uint32_t *small;         // kind 2
  uint32_t n_small;
};
bool msi_pcache_known(MsiPostingCache *c, const MsiCacheKey &k, MsiKnownPosting *out);
void msi_pcache_learn(MsiPostingCache *c, const MsiCacheKey &k, const uint8_t *bytes, size_t len);   // absent (len 0) or small values
void msi_pcache_describe(MsiPostingCache *c, void *token, const uint8_t *bytes, size_t len);         // kind 3, before the commit
void msi_pcache_stats(const MsiPostingCache *c, uint64_t out[4]);   // hits, misses, bytes used, capacity
// ---- staging at index-open (msi_dict_stage_postings) ----------------------------------------------------------------
// The stored values of whole databases put into the cache before the first search: bodies in HBM with their container
// tables parsed, "no such key" and raw small values on the host — what a search's first reader would have left there, for
// every key at once (one reservation and one host-to-device copy per call).  Thread-safe against searches and other calls.
struct MsiStageValue {
  MsiCacheKey key;
  const uint8_t *bytes;   // the stored value (null / len 0: the key is absent)
  size_t len;
};
// -> MSI_OK | MSI_E_OOM (the cache's HBM arena cannot hold the call's bodies: nothing of the call is staged) | MSI_E_INVALID (a value
// that does not parse as a CboRoaringBitmap); out: [bodies staged in HBM, values kept on the host, keys that were already known]
int32_t msi_pcache_stage(MsiPostingCache *c, const MsiStageValue *values, uint64_t n, uint64_t out[3]);
// Every key of database `db` (1 word_docids, 3 word_fid_docids, 4 word_position_docids, ...) of view `view` has been staged:
// a key the cache does not know DOES NOT EXIST in that database — the search asks nobody.
void msi_pcache_set_complete(MsiPostingCache *c, uint64_t view, uint32_t db_mask);
bool msi_pcache_complete(MsiPostingCache *c, uint32_t db, uint64_t view);   // (counts a hit when true)
// Forget what searches left in the cache; what was STAGED stays (bodies, tables, completeness).  No search in flight.
void msi_pcache_reset(MsiPostingCache *c);
void msi_pcache_staged_stats(const MsiPostingCache *c, uint64_t out[4]);   // staged: bodies, host-kept values, HBM bytes, complete-db answers
