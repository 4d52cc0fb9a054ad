// msi_search.hip — the keyword leg of Search::execute() with every graph-based ranking rule.
//
// Host side of the scorer (S3): the query graph and the rule graphs are tiny (tens of nodes) and
// stay on the caller's thread; every docid set is a slot of the msi_bits pool in HBM and every
// set operation is a device kernel (msi_bits.hip): batched CboRoaringBitmap decode of all the
// postings of a condition in one launch, and / or / andnot, fused op + cardinality, ordered
// extraction.  No set is ever materialised on the host.
//
// Restates, in crates/milli/src/search/new/:
//   query_term/{mod.rs,ntypo_subset.rs}                      term subsets
//   query_term/parse_query.rs:227-300                        make_ngram
//   query_term/compute_derivations.rs:21-383                 derivations (device dictionary, batched)
//   query_graph.rs:96-180,200-305,346-440,470-544            from_query, edges, removal order, build_from_paths
//   resolve_query_graph.rs:33-268                            term / phrase / graph docids
//   ranking_rule_graph/build.rs:12-91                        rule graph (edge order = DFS order)
//   ranking_rule_graph/cheapest_paths.rs:94-310              paths of a given cost, nodes_to_skip
//   ranking_rule_graph/{words,typo,proximity,fid,position,exactness}/
//   graph_based_ranking_rule.rs:136-368                      buckets by increasing cost
//   exact_attribute.rs:17-302, bucket_sort.rs:23-460, mod.rs:273-301,510-649
// The reference's DeadEndsCache only prunes paths that resolve to no document; here a path is not
// extended past a prefix whose documents are exhausted, which visits the same non-empty paths in
// the same order.  fid/mod.rs and position/mod.rs push their edges in hash-map order (unspecified);
// here ascending fid / ascending cost.
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <ucontext.h>

#include <algorithm>
#include <unordered_map>
#include <exception>
#include <functional>
#include <chrono>
#include <deque>
#include <map>
#include <memory>
#include <set>
#include <string>
#include <tuple>
#include <vector>

#include "msi_arena.h"
#include "msi_common.h"
#ifndef MSI_SEARCH_DIRECT_ONLY
#include "msi_vm.h"
int32_t msi_bits_sync(msi_bits *p);
bool msi_bits_take_summary_dirty(msi_bits *p);
void msi_bits_mark_summary_dirty(msi_bits *p);
const uint32_t *msi_doc_keys_device(const msi_doc_keys *k);
MsiPostingCache *msi_dict_pcache(const msi_dict *d);
void msi_cbo_batch_append_known(MsiCboBatch &batch, const MsiContainer *conts, uint32_t n, uint64_t cache_off);
#endif

struct msi_dict;
bool msi_dict_word(const msi_dict *d, uint32_t idx, const uint8_t **w, uint32_t *len);
void msi_dict_prefix_range(const msi_dict *d, const uint8_t *prefix, uint32_t plen, uint32_t *lo, uint32_t *hi);
uint32_t msi_bits_n_slots(msi_bits *p);
uint64_t msi_bits_n_docs(msi_bits *p);

namespace {
// every container below lives and dies inside one msi_keyword_search_ranked call: the thread's arena (msi_arena.h)
using msi_arena::Vec;
using msi_arena::Map;
using msi_arena::OrdSet;


constexpr uint32_t MAX_PREFIX_COUNT = 1000, MAX_ONE_TYPO_COUNT = 150, MAX_TWO_TYPOS_COUNT = 50;  // limits.rs
constexpr uint32_t MAX_WORD_LENGTH = 250;                                                        // lib.rs:146
constexpr uint32_t MAX_SYNONYM_PHRASE_COUNT = 50, MAX_SYNONYM_WORD_COUNT = 100;                    // limits.rs
constexpr uint32_t MAX_DISTANCE = 4;                                                             // proximity.rs:7

struct Fail {
  int32_t code;
};

// counters of the last search on this thread (msi_search_last_stats)
struct Stats {
  uint64_t launches = 0, syncs = 0, decodes = 0, callbacks = 0, postings_bytes = 0, paths = 0, buckets = 0;
  double callback_ms = 0, device_wait_ms = 0, total_ms = 0;
};
thread_local Stats g_stats;
// MSI_SEARCH_FK_TRACE: one line per first-k command recorded / delivered and per bucket emitted (debugging aid; read once)
inline bool fk_trace() {
  static const bool on = getenv("MSI_SEARCH_FK_TRACE") != nullptr;
  return on;
}
// process-wide: searches that continued in the compact space, and the documents of their universes (msi_search_compaction_stats)
std::atomic<uint64_t> g_compact_searches{0}, g_compact_docs{0}, g_ranked_searches{0};
std::atomic<uint64_t> g_late_compactions{0}, g_late_compact_docs{0};   // sub-trees of the bucket sort that moved into the space of their bucket
struct Clock {
  std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
  double ms() const { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(); }
};
void ck(int32_t st) {
  if (st != MSI_OK) throw Fail{st};
}
[[noreturn]] void fail(int32_t code, const char *msg) {
  msi_set_error("msi_keyword_search_ranked: %s", msg);
  throw Fail{code};
}

// ---- cooperative tasks ---------------------------------------------------------------------------
// bucket_sort hands every bucket to the next ranking rule; the sub-trees of sibling buckets do not depend on each
// other (a bucket's place in the result list is fixed by the cardinalities before it), only their ORDER matters.  With
// the command-list back end a search pays per dependency round, not per operation — so sibling sub-trees run as
// cooperative tasks on one thread: each task is the ordinary blocking code on its own stack; where it would wait for
// the device it parks, and when every task is parked the ONE list they recorded together runs.  A query with detailed
// scores is then ~25 rounds deep instead of 74 waits long.
struct Tasks {
  struct T {
    ucontext_t uc;
    std::unique_ptr<char[]> stack;
    std::function<void()> fn;
    bool done = false, parked = false;
    std::exception_ptr err;
  };
  static constexpr size_t STACK = 1u << 20;   // (index callbacks run on it: an embedding host language needs room)
  ucontext_t main_uc;
  Vec<std::unique_ptr<T>> all;
  T *cur = nullptr;
  bool abort = false;
  size_t live = 0;
  static void entry(unsigned lo, unsigned hi) {
    T *t = reinterpret_cast<T *>(((uintptr_t)hi << 32) | (uintptr_t)lo);
    try {
      t->fn();
    } catch (...) {
      t->err = std::current_exception();
    }
    t->done = true;
  }
  // Stacks are recycled per caller thread: a fresh megabyte is an mmap, a munmap and a page fault per touched page —
  // for a dozen tasks per query that was a fifth of a millisecond of kernel time.
  static std::vector<std::unique_ptr<char[]>> &stack_pool() {   // (outlives the searches: not the arena's)
    thread_local std::vector<std::unique_ptr<char[]>> pool;
    return pool;
  }
  static std::unique_ptr<char[]> take_stack() {
    auto &pool = stack_pool();
    if (pool.empty()) return std::unique_ptr<char[]>(new char[STACK]);
    std::unique_ptr<char[]> st = std::move(pool.back());
    pool.pop_back();
    return st;
  }
  static void give_stack(std::unique_ptr<char[]> st) {
    auto &pool = stack_pool();
    if (st && pool.size() < 32) pool.push_back(std::move(st));
  }
  ~Tasks() {
    for (auto &t : all) give_stack(std::move(t->stack));
  }
  void spawn(std::function<void()> fn) {
    std::unique_ptr<T> t(new T());
    t->stack = take_stack();
    t->fn = std::move(fn);
    getcontext(&t->uc);
    t->uc.uc_stack.ss_sp = t->stack.get();
    t->uc.uc_stack.ss_size = STACK;
    t->uc.uc_link = &main_uc;
    const uintptr_t a = (uintptr_t)t.get();
    makecontext(&t->uc, (void (*)())entry, 2, (unsigned)(a & 0xFFFFFFFFu), (unsigned)(a >> 32));
    ++live;
    all.push_back(std::move(t));
  }
  // called by the running task where it needs the device: back to the scheduler until the list has run
  void park() {
    T *t = cur;
    t->parked = true;
    swapcontext(&t->uc, &main_uc);
    if (abort) throw Fail{MSI_E_INTERNAL};
  }
  // runs every runnable task until all are parked or finished; -> true when some task is parked (the list must run)
  bool round(std::exception_ptr &first_err) {
    for (size_t i = 0; i < all.size(); ++i) {   // tasks spawned meanwhile run in the same round
      T *t = all[i].get();
      if (t->done || t->parked) continue;
      cur = t;
      swapcontext(&main_uc, &t->uc);
      cur = nullptr;
      if (t->done) {
        --live;
        give_stack(std::move(t->stack));
        t->fn = nullptr;
        if (t->err && !first_err) {
          first_err = t->err;
          abort = true;
        }
      }
    }
    bool parked = false;
    for (auto &t : all) parked = parked || (!t->done && t->parked);
    return parked;
  }
  void release() {
    for (auto &t : all) t->parked = false;
  }
};

// ---- device sets ------------------------------------------------------------------------------
struct SetPool {
  msi_bits *p;
  Vec<uint32_t> free_;
  Vec<uint32_t> clean_;  // free slots that are known to be all zero
  // command-list back end: a slot handed out as "all zero" is only zeroed when something READS it (most are first
  // written whole — the bucket of a cost level, a decode — or never used at all)
  Vec<uint8_t> lazy_zero;
  // compact space (Dev::compact_begin): the decodes of a list are hoisted into a phase of their own that runs FIRST, so a
  // slot freed while a list is being recorded must not be handed out again (as a decode's destination) before that list
  // has run — it waits here
  bool hold = false;
  Vec<uint32_t> held;
  SetPool(msi_bits *p_, uint32_t first) : p(p_), lazy_zero(msi_bits_n_slots(p_), 0) {
    for (uint32_t s = msi_bits_n_slots(p_); s-- > first;) free_.push_back(s);
  }
  void release_held() {
    free_.insert(free_.end(), held.begin(), held.end());
    held.clear();
  }
};
struct SetH {
  SetPool *pool;
  uint32_t slot;
  SetH(SetPool *p_, uint32_t s_) : pool(p_), slot(s_) {}
  ~SetH() {
    pool->lazy_zero[slot] = 0;
    if (pool->hold) pool->held.push_back(slot);
    else pool->free_.push_back(slot);
  }
};
using Set = std::shared_ptr<SetH>;

// Every device operation of a search goes through Dev.  Two back ends:
//   command lists (default; msi_vm.h): operations are RECORDED; at the points where the control flow needs a
//   cardinality (or the ordered docids) the recorded list is submitted and runs — together with the lists of every
//   other search waiting at that moment — as one kernel launch.  A search is then ~80 submissions instead of 211
//   launches + 80 waits, and the launches are shared by all searches in flight;
//   direct (MSI_SEARCH_VM=0, and the host-logic test double which has no kernels): one msi_bits call per operation.
// the paths of one cost level as the kernels take them: path k = slots[off[k] .. off[k+1])
struct PathSlots {
  Vec<uint32_t> off{0}, slots;
  size_t size() const { return off.size() - 1; }
  bool empty() const { return off.size() == 1; }
  void close_path() { off.push_back((uint32_t)slots.size()); }
};

struct Dev {
  SetPool pool;                        // the caller's pool: sets over docids
  // compact spaces the search has left (Ctx::late_leave): a set handle that outlives its space still finds its SetPool
  // (declared before every member that holds sets: destroyed after them)
  Vec<std::unique_ptr<SetPool>> retired_cpools;
  SetPool *cur = &pool;                // the pool every operation works on: `pool`, or — after compact_begin — `cpool`
  std::unique_ptr<SetPool> cpool;      // the companion pool: sets over the ranks of the documents of U0 (universe compaction)
  Set u0_full;                         // U0 in the caller's pool (read by the compact lists' decodes)
  Vec<Set> keep_until_run;     // full-space sets the recorded list still reads
#ifndef MSI_SEARCH_DIRECT_ONLY
  bool vm = true;
  MsiVmList list;
  MsiVmResult res;
  MsiPostingCache *pcache = nullptr;   // HBM posting cache of the index version (msi_dict_enable_posting_cache), or none
  Vec<void *> fills;           // cache entries the RECORDED decodes fill: ready once the list has run
  ~Dev() { drop_list(); }              // a search that ended with a recorded list it never ran (an error unwound it)
  // What is recorded will not run: the cache entries it was to fill go back, and if it had taken the pool's "summaries
  // are stale" flag (open_list) the flag returns — the next search's first list resets them.
  void drop_list() {
    for (void *t : fills) msi_pcache_abandon(pcache, t);
    fills.clear();
    if (!list.empty() && !list.words.empty() && list.words[0] == VM_SUMMARY_RESET) msi_bits_mark_summary_dirty(cur->p);
    list.clear();
  }
  struct PendingFk {
    Set set;   // keeps the slot from being reused before the list has run
    uint32_t k, ci, base;
    std::function<void(const uint32_t *, size_t)> sink;
  };
  Vec<PendingFk> pending_fk;
  // A stored posting value joins a decode batch: from the cache when another search left it there, else from the
  // bytes the callback handed over (and, when there is room, into the cache on the way).
  bool append_posting(MsiCboBatch &b, const MsiCacheKey &k, const uint8_t *bytes, size_t n) {
    if (!vm || !pcache || n <= 7 * sizeof(uint32_t)) return msi_cbo_batch_append(b, bytes, n);
    uint64_t off = 0;
    void *token = nullptr;
    const int r = msi_pcache_lookup(pcache, k, n, &off, &token);
    if (r == 1) return msi_cbo_batch_append(b, bytes, n, off, MSI_NO_CACHE);
    if (r == 2) {
      if (!msi_cbo_batch_append(b, bytes, n, MSI_NO_CACHE, off)) return false;
      msi_pcache_describe(pcache, token, bytes, n);   // its container table: later searches do not even ask the index
      b.fill_tokens.push_back(token);   // committed when the batch's decode has run; a list that fails or is dropped
      return true;                      // abandons them (msi_pcache_abandon): the next reader of the key refills
    }
    return msi_cbo_batch_append(b, bytes, n);
  }
#else
  static constexpr bool vm = false;
#endif
  explicit Dev(msi_bits *p) : pool(p, 0) {
#ifndef MSI_SEARCH_DIRECT_ONLY
    const char *knob = getenv("MSI_SEARCH_VM");
    vm = !(knob && knob[0] == '0');
    if (vm) ck(msi_bits_sync(p));   // whatever the caller enqueued on the pool's own stream is done before a list runs
#endif
  }
#ifndef MSI_SEARCH_DIRECT_ONLY
  // The first command of a list: when something outside the command lists touched the pool since the last one (a direct
  // kernel of this search, the caller between searches), the chunk summaries are stale and the list says so first.
  void open_list() {
    if (!list.empty()) return;
    list.begin();
    if (msi_bits_take_summary_dirty(cur->p)) list.words.push_back(VM_SUMMARY_RESET);
  }
  void rec(std::initializer_list<uint32_t> w) {
    open_list();
    list.words.insert(list.words.end(), w.begin(), w.end());
    static const bool op_trace = getenv("MSI_SEARCH_OP_TRACE") != nullptr;   // debugging aid: every recorded command, in order
    if (op_trace) {
      fprintf(stderr, "[msi op] list %llu %s:", (unsigned long long)g_stats.syncs, compact() ? "compact" : "full");
      for (uint32_t x : w) fprintf(stderr, " %u", x);
      fprintf(stderr, "\n");
    }
  }
  // the slot is about to be READ: a lazily zeroed slot is zeroed now
  void rd(uint32_t slot) {
    if (!cur->lazy_zero[slot]) return;
    cur->lazy_zero[slot] = 0;
    rec({VM_CLEAR, 1u, slot});
  }
  // the slot is about to be overwritten whole
  void wr(uint32_t slot) { cur->lazy_zero[slot] = 0; }
  // submit what was recorded and wait for it: the only blocking point of the command-list back end
  Tasks *tasks = nullptr;   // set while the bucket sort runs as cooperative tasks
  void run() {
    if (list.empty()) return;
    if (tasks && tasks->cur) {   // a task: park; the scheduler runs the shared list when every task is parked
      tasks->park();
      return;
    }
    run_now();
  }
  void run_now() {
    if (list.empty()) return;
    Clock ck_;
    ++g_stats.launches;
    ++g_stats.syncs;
    list.cache_base = msi_pcache_device_base(pcache);
    const int32_t st = msi_vm_run(cur->p, list, &res);
    g_stats.device_wait_ms += ck_.ms();
    list.clear();
    keep_until_run.clear();
    cur->release_held();
    for (void *t : fills) {
      if (st == MSI_OK) msi_pcache_commit(pcache, t);
      else msi_pcache_abandon(pcache, t);
    }
    fills.clear();
    Vec<PendingFk> fk;
    fk.swap(pending_fk);
    ck(st);
    for (PendingFk &f : fk) {   // a set smaller than k filled less of its block
      const size_t n = (size_t)std::min<uint64_t>(res.counts[f.ci], f.k);
      if (fk_trace())
        fprintf(stderr, "[msi fk] deliver slot %u k %u -> %zu ids (count %llu, base %u of %zu)\n", f.set->slot, f.k, n,
                (unsigned long long)res.counts[f.ci], f.base, res.firstk.size());
      f.sink(res.firstk.data() + f.base, n);
    }
  }
  // direct calls (GeoSort, distinct) see everything recorded so far
  void settle() {
    if (!vm) return;
    for (uint32_t sl = 0; sl < cur->lazy_zero.size(); ++sl) rd(sl);
    run();
  }
  uint32_t counts_for(uint32_t n) {   // room for n more cardinalities in this list (else it runs first)
    while (list.n_counts + n > MSI_VM_MAX_COUNTS) run();   // (a loop: other tasks may have filled the next list meanwhile)
    return list.new_counts(n);
  }
  void flush() {   // what is recorded runs now (deferred first-k ids are delivered)
    if (vm) run();
  }
  // The end of a search: the recorded list only has to run when somebody still waits for ids out of it — what else it
  // holds (the subtraction of the last bucket from a universe nobody will read, zeroing, decodes for rules that will not
  // start) dies with the search, and a round that nobody waits for is a round the next search does not queue behind.
  void finish_list() {
    if (!vm) return;
    if (!pending_fk.empty()) {
      run();
      return;
    }
    drop_list();
    keep_until_run.clear();
    cur->release_held();
  }
#else
  void settle() {}
  void flush() {}
  void finish_list() {}
#endif
#ifndef MSI_SEARCH_DIRECT_ONLY
  // ---- universe compaction ------------------------------------------------------------------------------------------
  // At 10 M documents a detailed 3-term search moved ~0.5 GB of set words per query through HBM (PMC, r3_pmc_ranked10):
  // every set operation of every rule swept 1.25 MB sets although the search's universe — the documents that match the
  // query at all — was 60 000 documents, 0.6 % of the index, spread over every chunk.  Every set a rule works with is a
  // subset of that universe U0, so once U0 is known the search continues in the COMPACT SPACE: document = its rank inside
  // U0 (monotone in the docid: ascending order, first-k and every set operation mean the same), a set = |U0| bits.
  // Postings are decoded straight into that space (VM_DECODEC), first-k ids leave it through the rank -> docid table.
  // The tables are filled by two commands that ride in the list that counts U0 (rank_tables).
  static int compact_mode() {   // MSI_SEARCH_COMPACT: 0 off, 1 when it pays (default), 2 always (tests: tiny corpora too)
    static const int m = getenv("MSI_SEARCH_COMPACT") ? atoi(getenv("MSI_SEARCH_COMPACT")) : 1;
    return m;
  }
  // MSI_SEARCH_LATE_COMPACT: 0 off; 1 (default) a sub-tree of the bucket sort moves into the compact space of its bucket
  // where that pays (Ctx::late_enter); 2 (tests, tiny corpora) every bucket that is ranked further does, whatever its size,
  // and the search never compacts its whole universe up front
  static int late_mode() {
    static const int m = getenv("MSI_SEARCH_LATE_COMPACT") ? atoi(getenv("MSI_SEARCH_LATE_COMPACT")) : 1;
    return m;
  }
  bool late_pays(uint64_t n_bucket) const {
    if (late_mode() >= 2) return n_bucket && n_bucket <= msi_bits_compact_capacity(pool.p);
    return n_bucket >= 64 && compact_pays(n_bucket);   // (below that the sub-tree is a handful of commands either way)
  }
  bool compact() const { return cur != &pool; }
  void rank_tables(const Set &u0) {
    rd(u0->slot);
    open_list();
    if (list.phase_start.size() + 1 > MSI_VM_MAX_PHASES) run();
    open_list();
    if (!msi_vm_record_rank(list, pool.p, u0->slot)) fail(MSI_E_OOM, "the rank tables of the compact space could not be allocated");
  }
  bool compact_pays(uint64_t n_u0) const {
    const uint64_t n = msi_bits_n_docs(pool.p);
    if (!n_u0 || n_u0 > msi_bits_compact_capacity(pool.p)) return false;
    return compact_mode() >= 2 || (n > 65536 && n_u0 * 8 <= n);
  }
  bool counted_compact = false;
  void compact_begin(const Set &u0, uint64_t n_u0, bool late = false) {
    if (!list.empty()) run();
    msi_bits *cp = msi_bits_compact_pool(pool.p);
    if (!cp) fail(MSI_E_OOM, "the companion pool of the compact space could not be created");
    if (cpool) retired_cpools.push_back(std::move(cpool));
    cpool.reset(new SetPool(cp, 0));
    cpool->hold = true;
    cur = cpool.get();
    u0_full = u0;
    list.clear();
    list.geom_docs = n_u0;
    list.full_pool = pool.p;
    list.u0_slot = u0->slot;
    if (late) {   // (counted apart: msi_search_compaction_stats is about whole universes — ADVICE r4)
      g_late_compactions.fetch_add(1, std::memory_order_relaxed);
      g_late_compact_docs.fetch_add(n_u0, std::memory_order_relaxed);
    } else if (!counted_compact) {
      counted_compact = true;
      g_compact_searches.fetch_add(1, std::memory_order_relaxed);
      g_compact_docs.fetch_add(n_u0, std::memory_order_relaxed);
    }
  }
  // back to the caller's pool (the sub-tree that lived in the compact space is finished: nothing recorded, nothing pending)
  void compact_end() {
    if (cur == &pool) return;
    list.clear();
    list.geom_docs = 0;
    list.full_pool = nullptr;
    list.u0_slot = 0;
    keep_until_run.clear();
    cpool->release_held();
    retired_cpools.push_back(std::move(cpool));
    cur = &pool;
    u0_full.reset();
  }
  // the compact-space image of a set of the caller's pool (its documents that are in U0, as ranks)
  Set compact_of(const Set &full) {
    Set s = alloc();
    wr(s->slot);
    open_list();
    msi_vm_record_compact(list, s->slot, full->slot);
    keep_until_run.push_back(full);
    return s;
  }
#endif
  Set alloc() {  // content undefined: the caller overwrites every word
#ifndef MSI_SEARCH_DIRECT_ONLY
    if (cur->free_.empty() && cur->clean_.empty() && !cur->held.empty()) {
      if (list.empty()) cur->release_held();
      else run();   // (slots freed while this list was recorded come back once it has run)
    }
#endif
    if (cur->free_.empty()) {
      if (cur->clean_.empty()) fail(MSI_E_OOM, "the msi_bits pool has no free slot left (create it with more slots)");
      const uint32_t s = cur->clean_.back();
      cur->clean_.pop_back();
      return msi_arena::make_shared<SetH>(cur, s);
    }
    const uint32_t s = cur->free_.back();
    cur->free_.pop_back();
    return msi_arena::make_shared<SetH>(cur, s);
  }
  void op(uint32_t d, uint32_t a, uint32_t b, int32_t o) {
#ifndef MSI_SEARCH_DIRECT_ONLY
    if (vm) {
      rd(a);
      rd(b);
      wr(d);
      return rec({VM_OP, d, a, b, (uint32_t)o});
    }
#endif
    ++g_stats.launches;
    ck(msi_bits_op(cur->p, d, a, b, o));
  }
  void fill(uint32_t d, int ones) {
#ifndef MSI_SEARCH_DIRECT_ONLY
    if (vm) {
      wr(d);
      return rec({VM_FILL, d, ones ? 1u : 0u});
    }
#endif
    ++g_stats.launches;
    ck(msi_bits_fill(cur->p, d, ones));
  }
  // Zeroed slots are handed out from a stock that one operation refills MSI_BITS_CLEAR_MAX at a time
  // (instead of one memset per set: a third of the launches of a search were clears).
  Set zeros() {
#ifndef MSI_SEARCH_DIRECT_ONLY
    if (vm) {
      Set s = alloc();
      cur->lazy_zero[s->slot] = 1;
      return s;
    }
#endif
    if (cur->clean_.empty()) {
      uint32_t batch[MSI_BITS_CLEAR_MAX];
      uint32_t n = 0;
      while (n < MSI_BITS_CLEAR_MAX && !cur->free_.empty()) {
        batch[n++] = cur->free_.back();
        cur->free_.pop_back();
      }
      if (!n) fail(MSI_E_OOM, "the msi_bits pool has no free slot left (create it with more slots)");
#ifndef MSI_SEARCH_DIRECT_ONLY
      if (vm) {
        rec({VM_CLEAR, n});
        list.words.insert(list.words.end(), batch, batch + n);
      } else
#endif
      {
        ++g_stats.launches;
        ck(msi_bits_clear_slots(cur->p, n, batch));
      }
      for (uint32_t k = n; k-- > 0;) cur->clean_.push_back(batch[k]);
    }
    const uint32_t s = cur->clean_.back();
    cur->clean_.pop_back();
    return msi_arena::make_shared<SetH>(cur, s);
  }
  Set ones() {
    Set s = alloc();
    fill(s->slot, 1);
    return s;
  }
  Set clone(const Set &a) {
    Set s = alloc();
    op(s->slot, a->slot, a->slot, MSI_BITS_AND);
    return s;
  }
  void and_(const Set &d, const Set &a) { op(d->slot, d->slot, a->slot, MSI_BITS_AND); }
  void or_(const Set &d, const Set &a) { op(d->slot, d->slot, a->slot, MSI_BITS_OR); }
  void sub_(const Set &d, const Set &a) { op(d->slot, d->slot, a->slot, MSI_BITS_ANDNOT); }
  Set and_new(const Set &a, const Set &b, uint64_t *count) {
    Set s = alloc();
    if (count) {
#ifndef MSI_SEARCH_DIRECT_ONLY
      if (vm) {
        const uint32_t c = counts_for(1);
        rd(a->slot);
        rd(b->slot);
        rec({VM_OP_COUNT, s->slot, a->slot, b->slot, (uint32_t)MSI_BITS_AND, c});
        run();
        *count = res.counts[c];
        return s;
      }
#endif
      Clock ck_;
      ++g_stats.launches;
      ++g_stats.syncs;
      ck(msi_bits_op_count(cur->p, s->slot, a->slot, b->slot, MSI_BITS_AND, count));
      g_stats.device_wait_ms += ck_.ms();
    } else {
      op(s->slot, a->slot, b->slot, MSI_BITS_AND);
    }
    return s;
  }
  // dst[i] = prefix & cond[i] with the cardinalities, one operation and one completion wait for all of them
  Vec<std::pair<Set, uint64_t>> and_many(const Set &prefix, const Vec<Set> &conds) {
    Vec<std::pair<Set, uint64_t>> out;
#ifndef MSI_SEARCH_DIRECT_ONLY
    if (vm) {
      for (size_t base = 0; base < conds.size(); base += 256) {
        const uint32_t n = (uint32_t)std::min<size_t>(256, conds.size() - base);
        // (the destinations first: an allocation may run the list recorded so far, or fail — never in mid-command)
        Vec<Set> ds(n);
        for (uint32_t k = 0; k < n; ++k) ds[k] = alloc();
        const uint32_t cb = counts_for(n);
        rd(prefix->slot);
        for (uint32_t k = 0; k < n; ++k) rd(conds[base + k]->slot);
        rec({VM_AND_MANY, prefix->slot, n, cb});
        for (uint32_t k = 0; k < n; ++k) {
          list.words.push_back(conds[base + k]->slot);
          list.words.push_back(ds[k]->slot);
          out.push_back({ds[k], 0});
        }
        run();
        for (uint32_t k = 0; k < n; ++k) out[base + k].second = res.counts[cb + k];
      }
      return out;
    }
#endif
    for (size_t base = 0; base < conds.size(); base += MSI_BITS_MANY) {
      const uint32_t n = (uint32_t)std::min<size_t>(MSI_BITS_MANY, conds.size() - base);
      uint32_t cs[MSI_BITS_MANY], ds[MSI_BITS_MANY];
      uint64_t counts[MSI_BITS_MANY];
      Vec<Set> dst;
      for (uint32_t k = 0; k < n; ++k) {
        dst.push_back(alloc());
        cs[k] = conds[base + k]->slot;
        ds[k] = dst[k]->slot;
      }
      Clock ck_;
      ++g_stats.launches;
      ++g_stats.syncs;
      ck(msi_bits_and_many_count(cur->p, prefix->slot, n, cs, ds, counts));
      g_stats.device_wait_ms += ck_.ms();
      for (uint32_t k = 0; k < n; ++k) out.push_back({dst[k], counts[k]});
    }
    return out;
  }
#ifndef MSI_SEARCH_DIRECT_ONLY
  uint32_t rec_paths(const PathSlots &paths, const Set &bucket, const Set &universe) {
    const uint32_t n = (uint32_t)paths.size();
    const uint32_t cb = counts_for(n);
    const uint32_t n_steps = (uint32_t)paths.slots.size();
    rd(universe->slot);
    for (uint32_t sl : paths.slots) rd(sl);
    // a bucket that was handed out as "all zero" and never touched is written whole by the level: no clear at all
    const uint32_t fresh = cur->lazy_zero[bucket->slot] ? 1u : 0u;
    wr(bucket->slot);
    rec({VM_PATHS, n, bucket->slot, universe->slot, cb, n_steps | (fresh << 31)});
    list.words.insert(list.words.end(), paths.off.begin(), paths.off.end());
    list.words.insert(list.words.end(), paths.slots.begin(), paths.slots.end());
    return cb;
  }
#endif
  // a whole cost level: path k claims universe & AND(its condition sets), in order; returns the cardinalities
  // `ids_k` / `ids`: also the bucket's first ids_k documents, in the same list (Ctx::spec_k); *ids stays null when the
  // list has no room for the command
  Vec<uint64_t> paths_claim(const PathSlots &paths, const Set &bucket, const Set &universe, uint32_t ids_k = 0,
                            std::shared_ptr<Vec<uint32_t>> *ids = nullptr) {
    Vec<uint64_t> counts(paths.size(), 0);
#ifndef MSI_SEARCH_DIRECT_ONLY
    if (vm) {
      const uint32_t cb = rec_paths(paths, bucket, universe);
      if (ids_k && ids) {
        auto box = msi_arena::make_shared<Vec<uint32_t>>();
        if (first_k_if_room(bucket, ids_k, [box](const uint32_t *d, size_t n) { box->assign(d, d + n); })) *ids = box;
      }
      run();
      for (size_t k = 0; k < paths.size(); ++k) counts[k] = res.counts[cb + k];
      return counts;
    }
#endif
    Clock ck_;
    ++g_stats.launches;
    ++g_stats.syncs;
    ck(msi_bits_paths_claim(cur->p, (uint32_t)paths.size(), paths.off.data(), paths.slots.data(), bucket->slot, universe->slot,
                            counts.data()));
    g_stats.device_wait_ms += ck_.ms();
    return counts;
  }
  // Sort rule: bucket := the documents of `universe` with its smallest order key, universe -= bucket
  Set order_next(const msi_doc_keys *keys, const Set &universe, uint32_t *key, uint64_t *count) {
    Set b = alloc();  // fully overwritten by the kernel
#ifndef MSI_SEARCH_DIRECT_ONLY
    if (vm && list.phase_start.size() + 1 <= MSI_VM_MAX_PHASES) {
      const uint64_t kp = (uint64_t)(uintptr_t)msi_doc_keys_device(keys);
      const uint32_t c = counts_for(2);
      if (list.phase_start.size() + 1 > MSI_VM_MAX_PHASES) run();
      rd(universe->slot);
      wr(b->slot);
      rec({VM_MINKEY, universe->slot, (uint32_t)kp, (uint32_t)(kp >> 32), 0});
      list.barrier();   // every chunk has contributed its minimum before any chunk takes
      rec({VM_TAKEKEY, universe->slot, b->slot, (uint32_t)kp, (uint32_t)(kp >> 32), 0, c, c + 1});
      run();
      *count = res.counts[c];
      *key = (uint32_t)res.counts[c + 1];
      if (!*count) *key = 0xFFFFFFFFu;
      return b;
    }
    settle();
#endif
    Clock ck_;
    g_stats.launches += 2;
    ++g_stats.syncs;
    ck(msi_bits_order_next(cur->p, keys, universe->slot, b->slot, key, count));
    direct_done();
    g_stats.device_wait_ms += ck_.ms();
    return b;
  }
  // GeoSort: bucket := the documents of `universe` within the error margin of the nearest / farthest one, universe -=
  // bucket; *first = the document whose point is the bucket's value (0xFFFFFFFF: no document has a point, nothing done)
  Set geo_next(const msi_geo_rule &r, uint32_t cap, double margin, const Set &universe, uint32_t *first, uint64_t *count) {
    Set b = alloc(), scratch = alloc();  // the bucket is fully overwritten by the kernel
    settle();
    Clock ck_;
    g_stats.launches += 2;
    ++g_stats.syncs;
    ck(msi_bits_geo_next(cur->p, r.points, universe->slot, b->slot, scratch->slot, r.lat, r.lng, r.ascending, cap, margin,
                         first, count));
    // The completion signal comes from the LAST WORKGROUP of the take kernel, not from the end of the kernel: the set
    // words the other workgroups stored may still sit in their XCDs' L2 (written back when the kernel ends).  The next
    // command list runs on another stream — it would overtake that write-back and read stale universe / bucket words
    // (seen on the MI355X as a rare wrong bucket with four searches in flight).  As after `distinct`: wait for the stream.
    direct_done();
    g_stats.device_wait_ms += ck_.ms();
    return b;
  }
  // the documents of `universe` that have a point, with their distances (at most `cap`; *total = how many there are)
  void geo_list(const msi_geo_rule &r, const Set &universe, uint32_t cap, Vec<uint32_t> &ids, Vec<double> &dist,
                uint64_t *total) {
    settle();
    Clock ck_;
    ++g_stats.launches;
    ++g_stats.syncs;
    ids.assign(cap, 0);
    dist.assign(cap, 0.0);
    ck(msi_bits_geo_list(cur->p, r.points, universe->slot, r.lat, r.lng, cap, ids.data(), dist.data(), total));
    const size_t n = (size_t)std::min<uint64_t>(*total, cap);
    ids.resize(n);
    dist.resize(n);
    direct_done();
    g_stats.device_wait_ms += ck_.ms();
  }
  // apply_distinct_rule (distinct.rs:19-36) on a COPY of `cands`: {kept candidates, every document of the index that
  // shares a value with one of them}
  std::pair<Set, Set> distinct(const msi_doc_values *vals, const Set &cands, uint64_t *kept) {
    Set work = clone(cands);  // consumed by the rounds; `cands` may be a shared, cached set
    Set rem = alloc(), exc = alloc();
    settle();
    Clock ck_;
    uint32_t rounds = 0;
    ck(msi_bits_distinct(cur->p, vals, work->slot, rem->slot, exc->slot, kept, &rounds));
    rounds &= 0x7FFFFFFFu;
    g_stats.launches += 2 + 2 * rounds;
    g_stats.syncs += rounds;
    g_stats.device_wait_ms += ck_.ms();
    direct_done();
    return {rem, exc};
  }
  Set distinct_excluded(const msi_doc_values *vals, const Set &kept) {
    Set exc = alloc();
    settle();
    g_stats.launches += 2;
    ck(msi_bits_distinct_excluded(cur->p, vals, kept->slot, exc->slot));
    direct_done();
    return exc;
  }
  // a direct call may leave work on the pool's own stream — or, behind a completion signal that its last workgroup
  // raised, dirty lines in an XCD's L2 until the kernel ends: the next list (another stream) must not overtake either
  void direct_done() {
#ifndef MSI_SEARCH_DIRECT_ONLY
    if (vm) ck(msi_bits_sync(cur->p));
#endif
  }
  // sets[i] -= removed, with the new cardinalities: one operation and one wait
  Vec<uint64_t> sub_many(const Set &removed, const Vec<Set> &sets) {
    Vec<uint64_t> out(sets.size(), 0);
#ifndef MSI_SEARCH_DIRECT_ONLY
    if (vm) {
      for (size_t base = 0; base < sets.size(); base += 512) {
        const uint32_t n = (uint32_t)std::min<size_t>(512, sets.size() - base);
        const uint32_t cb = counts_for(n);
        rd(removed->slot);
        for (uint32_t k = 0; k < n; ++k) rd(sets[base + k]->slot);
        rec({VM_SUB_MANY, removed->slot, n, cb});
        for (uint32_t k = 0; k < n; ++k) list.words.push_back(sets[base + k]->slot);
        run();
        for (uint32_t k = 0; k < n; ++k) out[base + k] = res.counts[cb + k];
      }
      return out;
    }
#endif
    for (size_t base = 0; base < sets.size(); base += MSI_BITS_MANY) {
      const uint32_t n = (uint32_t)std::min<size_t>(MSI_BITS_MANY, sets.size() - base);
      uint32_t ss[MSI_BITS_MANY];
      for (uint32_t k = 0; k < n; ++k) ss[k] = sets[base + k]->slot;
      Clock ck_;
      ++g_stats.launches;
      ++g_stats.syncs;
      ck(msi_bits_andnot_many_count(cur->p, removed->slot, n, ss, out.data() + base));
      g_stats.device_wait_ms += ck_.ms();
    }
    return out;
  }
  Set from_docids(const Vec<uint32_t> &ids) {
    Set s = alloc();
#ifndef MSI_SEARCH_DIRECT_ONLY
    if (vm) {
      MsiCboBatch b;
      b.small_ids.assign(ids.begin(), ids.end());
      open_list();
      ck(msi_vm_record_decode(list, cur->p, s->slot, b, true));
      return s;
    }
#endif
    ++g_stats.launches;
    ck(msi_bits_set_from_docids(cur->p, s->slot, ids.empty() ? nullptr : ids.data(), ids.size()));
    return s;
  }
  // the same level WITHOUT the completion wait: its counts land in `region`; false = does not fit, nothing enqueued
  // `pending_levels`: (count base, paths, region) of the levels recorded ahead — the CALLER's (a rule evaluation's) state: the
  // bucket sort's tasks interleave between an enqueue and its collect
  struct PendingLevel { uint32_t base, n, region; };
  using PendingLevels = Vec<PendingLevel>;
  bool paths_enqueue(const PathSlots &paths, const Set &bucket, const Set &universe, uint32_t region,
                     PendingLevels &pending_levels) {
#ifndef MSI_SEARCH_DIRECT_ONLY
    if (vm) {
      if (paths.size() > MSI_BITS_REGION_PATHS || list.n_counts + paths.size() > MSI_VM_MAX_COUNTS) return false;
      pending_levels.push_back(PendingLevel{rec_paths(paths, bucket, universe), (uint32_t)paths.size(), region});
      return true;
    }
#endif
    const int32_t st = msi_bits_paths_enqueue(cur->p, (uint32_t)paths.size(), paths.off.data(), paths.slots.data(), bucket->slot,
                                              universe->slot, region);
    if (st == MSI_E_UNSUPPORTED) return false;
    ck(st);
    ++g_stats.launches;
    return true;
  }
  Vec<uint64_t> paths_collect(uint32_t n_regions, PendingLevels &pending_levels) {  // ONE wait for every level enqueued so far
    Vec<uint64_t> counts((size_t)n_regions * MSI_BITS_REGION_PATHS, 0);
#ifndef MSI_SEARCH_DIRECT_ONLY
    if (vm) {
      run();
      // (levels without a path were not recorded: a level's counts go to the region its caller named)
      for (const PendingLevel &pl : pending_levels)
        if (pl.region < n_regions)
          for (uint32_t k = 0; k < pl.n; ++k) counts[(size_t)pl.region * MSI_BITS_REGION_PATHS + k] = res.counts[pl.base + k];
      pending_levels.clear();
      return counts;
    }
#endif
    Clock ck_;
    ++g_stats.syncs;
    ck(msi_bits_paths_collect(cur->p, n_regions, counts.data()));
    g_stats.device_wait_ms += ck_.ms();
    return counts;
  }
  void claim(const Set &docs, const Set &bucket, const Set &universe, const Vec<Set> &stack) {
#ifndef MSI_SEARCH_DIRECT_ONLY
    if (vm) {
      rd(docs->slot);
      rd(bucket->slot);
      rd(universe->slot);
      for (auto &s_ : stack) rd(s_->slot);
      rec({VM_CLAIM, docs->slot, bucket->slot, universe->slot, (uint32_t)stack.size()});
      for (auto &s_ : stack) list.words.push_back(s_->slot);
      return;
    }
#endif
    uint32_t ss[MSI_BITS_MANY];
    if (stack.size() > MSI_BITS_MANY) fail(MSI_E_INTERNAL, "path longer than the claim kernel supports");
    for (size_t k = 0; k < stack.size(); ++k) ss[k] = stack[k]->slot;
    ++g_stats.launches;
    ck(msi_bits_claim(cur->p, docs->slot, bucket->slot, universe->slot, (uint32_t)stack.size(), ss));
  }
  // the cardinalities of several sets: one list, one completion wait for all of them
  Vec<uint64_t> count_many(const Vec<Set> &sets) {
    Vec<uint64_t> out(sets.size(), 0);
#ifndef MSI_SEARCH_DIRECT_ONLY
    if (vm) {
      for (size_t base = 0; base < sets.size(); base += 64) {
        const uint32_t n = (uint32_t)std::min<size_t>(64, sets.size() - base);
        const uint32_t ci = counts_for(n);
        for (uint32_t i = 0; i < n; ++i) {
          rd(sets[base + i]->slot);
          rec({VM_COUNT, sets[base + i]->slot, ci + i});
        }
        run();
        for (uint32_t i = 0; i < n; ++i) out[base + i] = res.counts[ci + i];
      }
      return out;
    }
#endif
    for (size_t i = 0; i < sets.size(); ++i) out[i] = count(sets[i]);
    return out;
  }
  uint64_t count(const Set &a) {
    uint64_t c = 0;
#ifndef MSI_SEARCH_DIRECT_ONLY
    if (vm) {
      const uint32_t ci = counts_for(1);
      rd(a->slot);
      rec({VM_COUNT, a->slot, ci});
      run();
      return res.counts[ci];
    }
#endif
    Clock ck_;
    ++g_stats.launches;
    ++g_stats.syncs;
    ck(msi_bits_count(cur->p, a->slot, &c));
    g_stats.device_wait_ms += ck_.ms();
    return c;
  }
  Set decode(const MsiCboBatch &b) {
#ifndef MSI_SEARCH_DIRECT_ONLY
    if (vm) {
      if (b.containers.empty() && b.small_ids.empty()) return zeros();
      Set s = alloc();   // overwritten chunk by chunk: no zeroed slot needed
      ++g_stats.decodes;
      g_stats.postings_bytes += b.bytes.size() + 4 * b.small_ids.size();
      wr(s->slot);
      open_list();
      ck(msi_vm_record_decode(list, cur->p, s->slot, b, true));
      fills.insert(fills.end(), b.fill_tokens.begin(), b.fill_tokens.end());
      return s;
    }
#endif
    Set s = zeros();
    if (!b.containers.empty() || !b.small_ids.empty()) {
      Clock ck_;
      ++g_stats.decodes;
      g_stats.launches += (b.containers.empty() ? 0 : 1) + (b.small_ids.empty() ? 0 : 1);
      g_stats.postings_bytes += b.bytes.size() + 4 * b.small_ids.size();
      ck(msi_bits_decode_batch(cur->p, s->slot, b, false));
      g_stats.device_wait_ms += ck_.ms();
    }
    return s;
  }
  // The first k documents of `a` handed to `sink` — with the command-list back end not now but when the list runs next
  // (the ids of a bucket are not needed to go on with the bucket sort: only its cardinality is, and that is known):
  // a leaf bucket costs no wait of its own.  `a` stays alive until then.
  void first_k_later(const Set &a, uint32_t k, std::function<void(const uint32_t *, size_t)> sink) {
#ifndef MSI_SEARCH_DIRECT_ONLY
    if (vm && k <= MSI_VM_MAX_FIRSTK) {
      if (k == 0) {
        sink(nullptr, 0);
        return;
      }
      // (ONE loop over every limit: run() parks a task, and the list it comes back to has been written by the others —
      // checking the first-k limits and then waiting for a free count let a 17th first-k into a phase of 16, whose ids
      // came back as zeros: fuzz seed 12003397 with 16 levels per wait)
      while (list.fk_in_phase >= MSI_VM_MAX_FK_PHASE || list.firstk_total + k > MSI_VM_MAX_FIRSTK ||
             list.n_counts + 1 > MSI_VM_MAX_COUNTS)
        run();
      const uint32_t ci = list.new_counts(1);
      rd(a->slot);
      rec({VM_FIRSTK, a->slot, k, ci, list.firstk_total});
      if (fk_trace())
        fprintf(stderr, "[msi fk] record slot %u k %u base %u (in phase %u, list counts %u)\n", a->slot, k, list.firstk_total,
                list.fk_in_phase, list.n_counts);
      pending_fk.push_back(PendingFk{a, k, ci, list.firstk_total, std::move(sink)});
      list.firstk_total += k;
      ++list.fk_in_phase;
      list.max_fk_phase = std::max(list.max_fk_phase, list.fk_in_phase);
      return;
    }
#endif
    const Vec<uint32_t> ids = first_k(a, k);
    sink(ids.data(), ids.size());
  }
  // The same, but only when the open list takes the command as it stands (never runs a list: for callers that sit
  // between an enqueue and its collect).  false: nothing recorded.
  bool first_k_if_room(const Set &a, uint32_t k, std::function<void(const uint32_t *, size_t)> sink) {
#ifndef MSI_SEARCH_DIRECT_ONLY
    if (!vm || !k || k > MSI_VM_MAX_FIRSTK || list.empty() || list.fk_in_phase >= MSI_VM_MAX_FK_PHASE ||
        list.firstk_total + k > MSI_VM_MAX_FIRSTK || list.n_counts + 1 > MSI_VM_MAX_COUNTS || cur->lazy_zero[a->slot])
      return false;
    first_k_later(a, k, std::move(sink));
    return true;
#else
    (void)a; (void)k; (void)sink;
    return false;
#endif
  }
  Vec<uint32_t> first_k(const Set &a, uint32_t k) {
#ifndef MSI_SEARCH_DIRECT_ONLY
    if (vm && k <= MSI_VM_MAX_FIRSTK) {
      Vec<uint32_t> out;
      first_k_later(a, k, [&out](const uint32_t *ids, size_t n) { out.assign(ids, ids + n); });
      run();
      return out;
    }
    settle();
#endif
    Vec<uint32_t> ids(std::max<uint32_t>(k, 1));
    uint32_t n = 0;
    Clock ck_;
    g_stats.launches += 3;
    ++g_stats.syncs;
    ck(msi_bits_first_k(cur->p, a->slot, k, ids.data(), &n));
    g_stats.device_wait_ms += ck_.ms();
    ids.resize(n);
    return ids;
  }
};

// ---- terms and subsets (query_term/mod.rs, ntypo_subset.rs) -------------------------------------------
using Phrase = Vec<int32_t>;  // word ids, -1 = a stop word inside the phrase

// A set of small ids (word / phrase interner ids, graph nodes) in ascending order: what the reference keeps in
// BTreeSet<u32> / SmallBitmap.  The host logic copies and compares these tens of thousands of times per query (subsets
// inside condition keys, path bookkeeping), so up to six ids live inside the object and copying one allocates nothing.
struct IdSet {
  static constexpr uint32_t INL = 6;
  uint32_t n = 0, cap = INL;
  uint32_t inl[INL];
  uint32_t *heap = nullptr;
  IdSet() {}
  IdSet(std::initializer_list<uint32_t> l) { for (uint32_t x : l) insert(x); }
  IdSet(const IdSet &o) { assign(o); }
  IdSet(IdSet &&o) noexcept { steal(o); }
  IdSet &operator=(const IdSet &o) {
    if (this != &o) { n = 0; assign(o); }
    return *this;
  }
  IdSet &operator=(IdSet &&o) noexcept {
    if (this != &o) { delete[] heap; heap = nullptr; cap = INL; steal(o); }
    return *this;
  }
  ~IdSet() { delete[] heap; }
  const uint32_t *data() const { return heap ? heap : inl; }
  uint32_t *data() { return heap ? heap : inl; }
  const uint32_t *begin() const { return data(); }
  const uint32_t *end() const { return data() + n; }
  bool empty() const { return n == 0; }
  size_t size() const { return n; }
  void clear() { n = 0; }
  size_t count(uint32_t x) const { return std::binary_search(begin(), end(), x) ? 1 : 0; }
  bool insert(uint32_t x) {
    uint32_t *d = data();
    uint32_t *at = std::lower_bound(d, d + n, x);
    if (at != d + n && *at == x) return false;
    const size_t i = at - d;
    if (n == cap) { grow(cap * 2); d = data(); }
    memmove(d + i + 1, d + i, (n - i) * sizeof(uint32_t));
    d[i] = x;
    ++n;
    return true;
  }
  template <class It> void insert(It a, It b) { for (; a != b; ++a) insert((uint32_t)*a); }
  void erase(uint32_t x) {
    uint32_t *d = data();
    uint32_t *at = std::lower_bound(d, d + n, x);
    if (at == d + n || *at != x) return;
    memmove(at, at + 1, (d + n - at - 1) * sizeof(uint32_t));
    --n;
  }
  void swap(IdSet &o) { IdSet t(std::move(o)); o = std::move(*this); *this = std::move(t); }
  int cmp(const IdSet &o) const {  // lexicographic, as std::set / BTreeSet compare
    const uint32_t *a = data(), *b = o.data();
    const uint32_t m = std::min(n, o.n);
    for (uint32_t i = 0; i < m; ++i) if (a[i] != b[i]) return a[i] < b[i] ? -1 : 1;
    return n == o.n ? 0 : (n < o.n ? -1 : 1);
  }
  bool operator==(const IdSet &o) const { return n == o.n && memcmp(data(), o.data(), n * sizeof(uint32_t)) == 0; }
  bool operator!=(const IdSet &o) const { return !(*this == o); }
  bool operator<(const IdSet &o) const { return cmp(o) < 0; }

 private:
  void grow(uint32_t want) {
    uint32_t *h = new uint32_t[want];
    memcpy(h, data(), n * sizeof(uint32_t));
    delete[] heap;
    heap = h;
    cap = want;
  }
  void assign(const IdSet &o) {  // this->n == 0; keeps whatever storage this object already has
    if (o.n > cap) { delete[] heap; heap = new uint32_t[o.n]; cap = o.n; }
    memcpy(data(), o.data(), o.n * sizeof(uint32_t));
    n = o.n;
  }
  void steal(IdSet &o) {  // this object owns no heap block
    n = o.n;
    if (o.heap) { heap = o.heap; cap = o.cap; o.heap = nullptr; o.cap = INL; }
    else memcpy(inl, o.inl, o.n * sizeof(uint32_t));
    o.n = 0;
  }
};

struct NTypo {
  uint8_t kind = 1;  // 0 Nothing, 1 All, 2 Subset
  IdSet words, phrases;
  bool is_empty() const { return kind == 0 || (kind == 2 && words.empty() && phrases.empty()); }
  bool has_word(uint32_t w) const { return kind == 1 || (kind == 2 && words.count(w)); }
  bool has_phrase(uint32_t p) const { return kind == 1 || (kind == 2 && phrases.count(p)); }
  void intersect(const NTypo &o) {
    if (kind == 1) { *this = o; return; }
    if (kind == 0 || o.kind == 1) return;
    if (o.kind == 0) { *this = NTypo{0, {}, {}}; return; }
    IdSet w, p;
    for (uint32_t x : words) if (o.words.count(x)) w.insert(x);
    for (uint32_t x : phrases) if (o.phrases.count(x)) p.insert(x);
    words.swap(w);
    phrases.swap(p);
  }
  // Ordered as the tuple (kind, words, phrases) — written out as ONE three-way pass: these keys are compared tens of
  // thousands of times per query (condition tables, the caches of decoded sets), and Nothing / All carry no sets.
  int cmp(const NTypo &o) const {
    if (kind != o.kind) return kind < o.kind ? -1 : 1;
    if (kind != 2) return 0;
    if (int c = words.cmp(o.words)) return c;
    return phrases.cmp(o.phrases);
  }
  bool operator<(const NTypo &o) const { return cmp(o) < 0; }
  bool operator==(const NTypo &o) const { return cmp(o) == 0; }
};
const NTypo NT_NONE{0, {}, {}};

struct Subset {
  uint32_t term = 0;
  NTypo zero, one, two;
  bool mandatory = false;
  int cmp(const Subset &o) const {   // the order of the tuple (term, zero, one, two, mandatory)
    if (term != o.term) return term < o.term ? -1 : 1;
    if (int c = zero.cmp(o.zero)) return c;
    if (int c = one.cmp(o.one)) return c;
    if (int c = two.cmp(o.two)) return c;
    if (mandatory != o.mandatory) return mandatory < o.mandatory ? -1 : 1;
    return 0;
  }
  bool operator<(const Subset &o) const { return cmp(o) < 0; }
  bool operator==(const Subset &o) const { return cmp(o) == 0; }
};
struct Located {
  Subset subset;
  uint32_t pos_lo = 0, pos_hi = 0, id_lo = 0, id_hi = 0;
  uint32_t n_ids() const { return id_hi - id_lo + 1; }
  int cmp(const Located &o) const {  // the order of the tuple (subset, pos_lo, pos_hi, id_lo, id_hi)
    if (int c = subset.cmp(o.subset)) return c;
    if (pos_lo != o.pos_lo) return pos_lo < o.pos_lo ? -1 : 1;
    if (pos_hi != o.pos_hi) return pos_hi < o.pos_hi ? -1 : 1;
    if (id_lo != o.id_lo) return id_lo < o.id_lo ? -1 : 1;
    if (id_hi != o.id_hi) return id_hi < o.id_hi ? -1 : 1;
    return 0;
  }
  bool operator<(const Located &o) const { return cmp(o) < 0; }
  bool operator==(const Located &o) const { return cmp(o) == 0; }
};

struct Term {
  uint32_t original = 0;
  bool is_ngram = false;
  Vec<uint32_t> ngram_words;
  int32_t phrase = -1;
  uint32_t max_lev = 0;
  bool is_prefix = false;
  int32_t exact = -1;
  Vec<uint32_t> prefix_of, one_typo, two_typos;
  int32_t split_words = -1;
  Vec<uint32_t> synonyms;  // phrase ids
  int32_t use_prefix_db = -1;  // the word itself when it is a key of the word-prefix databases
  bool too_long = false;
};

enum RuleKind { R_WORDS, R_TYPO, R_PROXIMITY, R_FID, R_POSITION, R_EXACTNESS, R_EXACT_ATTRIBUTE, R_ORDER_BY };
enum CondKind { C_TERM, C_TYPO, C_PROX, C_FID, C_POSITION, C_EXACT, C_ANY };

struct Condition {
  int kind = C_TERM;
  Located term;            // destination term
  Located left;            // C_PROX
  bool has_left = false;
  uint32_t x = 0;          // typo count / proximity cost / fid
  bool has_fid = false;    // C_FID
  Vec<uint16_t> positions;
  bool operator<(const Condition &o) const {   // the order of the tuple (kind, term, has_left, left, x, has_fid, positions)
    if (kind != o.kind) return kind < o.kind;
    if (int c = term.cmp(o.term)) return c < 0;
    if (has_left != o.has_left) return has_left < o.has_left;
    if (int c = left.cmp(o.left)) return c < 0;
    if (x != o.x) return x < o.x;
    if (has_fid != o.has_fid) return has_fid < o.has_fid;
    return positions < o.positions;
  }
};

struct GNode {
  int kind = 0;  // 0 start, 1 end, 2 term, 3 deleted
  Located term;
  IdSet preds, succs;
};
struct Graph {
  Vec<GNode> nodes;
  static constexpr uint32_t ROOT = 0, END = 1;
};
using PathSubsets = Vec<std::pair<std::pair<bool, Located>, Located>>;  // (start?, dest) per condition

struct Score {
  uint32_t kind, a, b;
};

struct Resolved {
  Set docs;
  bool has_start = false;
  Located start, end;
};
// The conditions of the edges into one located term (from one adjacent term, for proximity) and what they resolved to.
// A query meets the same (rule, source, destination) in bucket after bucket: built once per query (Ctx::edge_memo); a rule
// evaluation holds the entries it uses, so dropping the memo under memory pressure never pulls them from under it.
struct EdgeSet {
  Vec<std::pair<uint32_t, Condition>> conds;   // (cost, condition), build_edges' order
  Vec<std::unique_ptr<Resolved>> resolved;     // on demand
};
struct EdgeKeyRef {   // a key that only points at its terms: lookups copy nothing
  int rule;
  const Located *dst;
  bool adjacent;   // proximity between adjacent terms: the only case in which the source term shapes the conditions
  const Located *src;
};
struct EdgeKey {
  int rule;
  Located dst;
  bool adjacent;
  Located src;
  EdgeKeyRef ref() const { return {rule, &dst, adjacent, &src}; }
};
struct EdgeKeyLess {
  using is_transparent = void;
  static bool lt(const EdgeKeyRef &a, const EdgeKeyRef &b) {
    if (a.rule != b.rule) return a.rule < b.rule;
    if (int c = a.dst->cmp(*b.dst)) return c < 0;
    if (a.adjacent != b.adjacent) return a.adjacent < b.adjacent;
    return a.adjacent && a.src->cmp(*b.src) < 0;
  }
  bool operator()(const EdgeKey &a, const EdgeKey &b) const { return lt(a.ref(), b.ref()); }
  bool operator()(const EdgeKey &a, const EdgeKeyRef &b) const { return lt(a.ref(), b); }
  bool operator()(const EdgeKeyRef &a, const EdgeKey &b) const { return lt(a, b.ref()); }
};

// ---- the search context ---------------------------------------------------------------------
struct Ctx {
  msi_dict *dict;
  const msi_index_vtable *ix;
  const msi_search_params *prm;
  Dev dev;
  // from + length when the page is small: the LAST rule's evaluations ask for the first ids of the first bucket they
  // plan in the list that computes it (GraphRule::look_ahead) — a leaf bucket's ids then arrive with its cardinality, and
  // the search ends without the one more round that only fetched ids.  0: off.
  uint32_t spec_k = 0;
  Vec<std::string> words;
  std::unordered_map<std::string, uint32_t, std::hash<std::string>, std::equal_to<std::string>,
                     msi_arena::Alloc<std::pair<const std::string, uint32_t>>> word_ids;   // (lookups only; the ids are the order of `words`)
  Vec<Phrase> phrases;
  Map<Phrase, uint32_t> phrase_ids;
  Vec<Term> terms;
  Map<uint32_t, Set> phrase_cache;
  // Document sets that do not depend on a universe, decoded once per search and then only intersected:
  // the rules of a search resolve the same term subsets again and again (every bucket of a rule restarts
  // the rules below it).
  Map<Subset, Set> subset_cache;
  Map<EdgeKey, std::shared_ptr<EdgeSet>, EdgeKeyLess> edge_memo;
  Map<std::tuple<Subset, int, Vec<uint32_t>>, Set> within_cache;
  Map<std::string, Vec<Set>> exact_attr_cache;  // {ExactMatch, MatchesStart, position candidates}
  Map<std::tuple<Subset, Subset, uint32_t, uint32_t>, Set> prox_cache;
  Map<std::pair<uint32_t, bool>, Set> word_cache;
  Map<uint32_t, uint32_t> freq_weight;   // MSI_TERMS_FREQUENCY: removal weight per term id (removal_order_frequency)
  void forget() {   // every cached set (all of them can be recomputed from the index)
    subset_cache.clear();
    within_cache.clear();
    prox_cache.clear();
    word_cache.clear();
    exact_attr_cache.clear();
    phrase_cache.clear();
    edge_memo.clear();
    empty_.reset();
  }
#ifndef MSI_SEARCH_DIRECT_ONLY
  // The search moves into the compact space (Dev::compact_begin): what the caches hold so far — the term subsets, words
  // and phrases the universe was computed from — follows as images (one VM_DECODEC from the full-space slot each);
  // everything decoded from here on is decoded straight into that space.
  void to_compact_space() {
    for (auto &kv : subset_cache) kv.second = dev.compact_of(kv.second);
    for (auto &kv : word_cache) kv.second = dev.compact_of(kv.second);
    for (auto &kv : phrase_cache) kv.second = dev.compact_of(kv.second);
    within_cache.clear();
    prox_cache.clear();
    exact_attr_cache.clear();
    edge_memo.clear();
    empty_.reset();
  }
#endif
#ifndef MSI_SEARCH_DIRECT_ONLY
  // ---- a sub-tree of the bucket sort in the compact space of ITS bucket -----------------------------------------------
  // A search whose universe is too large to compact (a query with one frequent word: the union of its words' documents)
  // still narrows quickly: the first `words` bucket of "the matrix" is the intersection.  When a bucket that is to be
  // ranked by the rules below holds at most an eighth of the index and nothing else of the bucket sort is alive, the rules
  // below run in the compact space of that bucket: the full-space caches are put aside (they serve the rules above again
  // afterwards), the term subsets / words / phrases follow as images, and when the sub-tree has finished the search is
  // back in the caller's pool.  One space at a time (one companion pool, one set of rank tables per pool).
  struct LateStash {
    Map<uint32_t, Set> phrase_cache;
    Map<Subset, Set> subset_cache;
    Map<EdgeKey, std::shared_ptr<EdgeSet>, EdgeKeyLess> edge_memo;
    Map<std::tuple<Subset, int, Vec<uint32_t>>, Set> within_cache;
    Map<std::string, Vec<Set>> exact_attr_cache;
    Map<std::tuple<Subset, Subset, uint32_t, uint32_t>, Set> prox_cache;
    Map<std::pair<uint32_t, bool>, Set> word_cache;
    Set empty_;
  };
  std::unique_ptr<LateStash> late;
  void late_enter(const Set &bucket, uint64_t count) {
    dev.rank_tables(bucket);                 // (rides in the list compact_begin runs before it switches pools)
    dev.compact_begin(bucket, count, true);
    late.reset(new LateStash());
    late->phrase_cache.swap(phrase_cache);
    late->subset_cache.swap(subset_cache);
    late->edge_memo.swap(edge_memo);
    late->within_cache.swap(within_cache);
    late->exact_attr_cache.swap(exact_attr_cache);
    late->prox_cache.swap(prox_cache);
    late->word_cache.swap(word_cache);
    late->empty_.swap(empty_);
    for (auto &kv : late->subset_cache) subset_cache[kv.first] = dev.compact_of(kv.second);
    for (auto &kv : late->word_cache) word_cache[kv.first] = dev.compact_of(kv.second);
    for (auto &kv : late->phrase_cache) phrase_cache[kv.first] = dev.compact_of(kv.second);
  }
  // `unwinding`: an error is on its way up — nothing may block, what was recorded is dropped
  void late_leave(bool unwinding) {
    if (!late) return;
    if (unwinding) {
      dev.drop_list();
      dev.pending_fk.clear();
    } else {
      dev.finish_list();   // ids still to come are fetched; anything else recorded for the finished sub-tree is dropped
    }
    forget();              // the compact-space caches
    phrase_cache.swap(late->phrase_cache);
    subset_cache.swap(late->subset_cache);
    edge_memo.swap(late->edge_memo);
    within_cache.swap(late->within_cache);
    exact_attr_cache.swap(late->exact_attr_cache);
    prox_cache.swap(late->prox_cache);
    word_cache.swap(late->word_cache);
    empty_.swap(late->empty_);
    late.reset();
    dev.compact_end();
  }
#endif
  void relieve() {  // keep the pool from running dry on long queries: drop what can be recomputed
    // (slots freed while a compact list is being recorded wait in `held` until it has run: they are as good as free)
    if (dev.cur->free_.size() + dev.cur->clean_.size() + dev.cur->held.size() >= 48) return;
#ifndef MSI_SEARCH_DIRECT_ONLY
    // another task of the bucket sort may be parked with a reference into these maps: with tasks alive the caches stay
    // (a search that then runs out of slots is re-run sequentially, where this works again)
    if (dev.tasks && dev.tasks->live > 1) return;
#endif
    subset_cache.clear();
    within_cache.clear();
    prox_cache.clear();
    word_cache.clear();
    exact_attr_cache.clear();
    edge_memo.clear();
  }

  // The knobs every rule evaluation asks for, read ONCE per search (a getenv is a walk over the whole environment: with two
  // of them per look-ahead and one per level, ~90 walks per query; the tests that flip them do so between searches)
  int knob_levels_per_wait = -1;    // MSI_SEARCH_LEVELS_PER_WAIT (-1: not set)
  bool knob_fused_off = false;      // MSI_SEARCH_FUSED_LEVELS=0
  bool knob_known_off = false;      // MSI_SEARCH_KNOWN_OUTCOMES=0
  Ctx(msi_dict *d, msi_bits *pool, const msi_index_vtable *i, const msi_search_params *p)
      : dict(d), ix(i), prm(p), dev(pool) {
#ifndef MSI_SEARCH_DIRECT_ONLY
    if (dev.vm) dev.pcache = msi_dict_pcache(d);
#endif
    if (const char *k = getenv("MSI_SEARCH_LEVELS_PER_WAIT")) knob_levels_per_wait = std::max(0, atoi(k));
    const char *f = getenv("MSI_SEARCH_FUSED_LEVELS");
    knob_fused_off = f && f[0] == '0';
    const char *ko = getenv("MSI_SEARCH_KNOWN_OUTCOMES");
    knob_known_off = ko && ko[0] == '0';
  }

  uint32_t word(const std::string &w) {
    auto it = word_ids.find(w);
    if (it != word_ids.end()) return it->second;
    words.push_back(w);
    return word_ids[w] = (uint32_t)words.size() - 1;
  }
  uint32_t phrase(const Phrase &p) {
    auto it = phrase_ids.find(p);
    if (it != phrase_ids.end()) return it->second;
    phrases.push_back(p);
    return phrase_ids[p] = (uint32_t)phrases.size() - 1;
  }

  // -- the index reads (postings arrive as stored bytes and go straight into a decode batch) -----
  // db: which database the value came from; (s1, s2, x, y): its key there — together the posting-cache key
  void take(MsiCboBatch &b, int32_t st, const uint8_t *bytes, size_t n, const char *what, uint32_t db,
            const std::string &s1, const std::string &s2, uint64_t x, uint64_t y) {
    if (st < 0) {
      msi_set_error("msi_keyword_search_ranked: %s callback failed (%d)", what, st);
      throw Fail{MSI_E_INTERNAL};
    }
#ifndef MSI_SEARCH_DIRECT_ONLY
    if (dev.pcache && (!n || !bytes || n <= 7 * sizeof(uint32_t)))   // "no such key" and raw small values: remembered on the host
      msi_pcache_learn(dev.pcache, msi_cache_key(db, s1.data(), s1.size(), s2.data(), s2.size(), x, y, prm->index_view), bytes, bytes ? n : 0);
#endif
    if (!n || !bytes) return;
#ifndef MSI_SEARCH_DIRECT_ONLY
    const bool ok = dev.pcache ? dev.append_posting(b, msi_cache_key(db, s1.data(), s1.size(), s2.data(), s2.size(), x, y, prm->index_view), bytes, n)
                               : msi_cbo_batch_append(b, bytes, n);
#else
    (void)db; (void)s1; (void)s2; (void)x; (void)y;
    const bool ok = msi_cbo_batch_append(b, bytes, n);
#endif
    if (!ok) {
      msi_set_error("msi_keyword_search_ranked: malformed posting list from %s", what);
      throw Fail{MSI_E_INVALID};
    }
  }
  // The posting cache of the index version answers instead of the index (msi_pcache_known): the key is absent, a raw
  // small value, or a serialisation whose body is in HBM and whose container table was parsed when it was first read.
  // -> true: `b` (when given) has what the key holds; *card its cardinality; *present whether the key exists.
  bool from_cache(MsiCboBatch *b, uint32_t db, const std::string &s1, const std::string &s2, uint64_t x, uint64_t y,
                  uint64_t *card, bool *present) {
#ifndef MSI_SEARCH_DIRECT_ONLY
    static const bool off = getenv("MSI_PCACHE_KNOWN") && getenv("MSI_PCACHE_KNOWN")[0] == '0';   // experiments
    if (!dev.vm || !dev.pcache || off) return false;
    MsiKnownPosting kp;
    if (!msi_pcache_known(dev.pcache, msi_cache_key(db, s1.data(), s1.size(), s2.data(), s2.size(), x, y, prm->index_view), &kp)) {
      // a database staged whole at index-open (msi_dict_stage_complete): what the cache does not hold does not exist
      if (!msi_pcache_complete(dev.pcache, db, prm->index_view)) return false;
      if (card) *card = 0;
      if (present) *present = false;
      return true;
    }
    if (card) *card = kp.card;
    if (present) *present = kp.kind != 1;
    if (b) {
      if (kp.kind == 2) b->small_ids.insert(b->small_ids.end(), kp.small, kp.small + kp.n_small);
      else if (kp.kind == 3) msi_cbo_batch_append_known(*b, kp.conts, kp.n_conts, kp.off);
    }
    return true;
#else
    (void)b; (void)db; (void)s1; (void)s2; (void)x; (void)y; (void)card; (void)present;
    return false;
#endif
  }
  struct Cb {
    Clock c;
    ~Cb() {
      ++g_stats.callbacks;
      g_stats.callback_ms += c.ms();
    }
  };
  // (*card, when asked for: how many documents the word has — from the posting cache, or the stored value's header)
  bool add_word(MsiCboBatch &b, uint32_t w, bool original, uint64_t *card = nullptr) {
    const std::string &s = words[w];
    bool present = false;
    if (card) *card = 0;
    if (from_cache(&b, 1, s, std::string(), original ? 1 : 0, 0, card, &present)) return present;
    const uint8_t *bytes = nullptr;
    size_t n = 0;
    Cb cb_;
    const int32_t st = ix->word_docids(ix->user, (const uint8_t *)s.data(), (uint32_t)s.size(), original ? 1 : 0, &bytes, &n);
    take(b, st, bytes, n, "word_docids", 1, s, std::string(), original ? 1 : 0, 0);
    if (card && n && bytes) *card = msi_cbo_cardinality(bytes, n);
    return n != 0;
  }
  bool contains_word(uint32_t w) {   // Index::contains_word: the key exists; nothing is decoded
    const std::string &s = words[w];
    bool present = false;
    if (from_cache(nullptr, 1, s, std::string(), 1, 0, nullptr, &present)) return present;
    const uint8_t *bytes = nullptr;
    size_t n = 0;
    Cb cb_;
    const int32_t st = ix->word_docids(ix->user, (const uint8_t *)s.data(), (uint32_t)s.size(), 1, &bytes, &n);
    if (st < 0) fail(MSI_E_INTERNAL, "word_docids callback failed");
    return n != 0;
  }
  uint64_t add_pair(MsiCboBatch *b, uint32_t prox, uint32_t w1, uint32_t w2) {  // returns the cardinality
    if (!ix->word_pair_proximity_docids) return 0;
    const std::string &l = words[w1], &r = words[w2];
    uint64_t known_card = 0;
    if (from_cache(b, 2, l, r, prox, 0, &known_card, nullptr)) return known_card;
    const uint8_t *bytes = nullptr;
    size_t n = 0;
    Cb cb_;
    const int32_t st = ix->word_pair_proximity_docids(ix->user, prox, (const uint8_t *)l.data(), (uint32_t)l.size(),
                                                      (const uint8_t *)r.data(), (uint32_t)r.size(), &bytes, &n);
    if (st < 0) fail(MSI_E_INTERNAL, "word_pair_proximity_docids callback failed");
    if (!n || !bytes) {
#ifndef MSI_SEARCH_DIRECT_ONLY
      if (dev.pcache) msi_pcache_learn(dev.pcache, msi_cache_key(2, l.data(), l.size(), r.data(), r.size(), prox, 0, prm->index_view), nullptr, 0);
#endif
      return 0;
    }
    const uint64_t card = msi_cbo_cardinality(bytes, n);
    if (b) take(*b, st, bytes, n, "word_pair_proximity_docids", 2, l, r, prox, 0);
    return card;
  }
  void add_word_fid(MsiCboBatch &b, uint32_t w, uint32_t fid) {
    if (!ix->word_fid_docids) fail(MSI_E_INVALID, "the index vtable has no word_fid_docids (attribute / exactness rule)");
    const std::string &s = words[w];
    if (from_cache(&b, 3, s, std::string(), fid, 0, nullptr, nullptr)) return;
    const uint8_t *bytes = nullptr;
    size_t n = 0;
    Cb cb_;
    // the call first: `bytes` / `n` as further arguments of the same call would be read in an unspecified order
    const int32_t st = ix->word_fid_docids(ix->user, (const uint8_t *)s.data(), (uint32_t)s.size(), fid, &bytes, &n);
    take(b, st, bytes, n, "word_fid_docids", 3, s, std::string(), fid, 0);
  }
  void add_word_position(MsiCboBatch &b, uint32_t w, uint32_t pos) {
    if (!ix->word_position_docids)
      fail(MSI_E_INVALID, "the index vtable has no word_position_docids (position / exactness rule)");
    const std::string &s = words[w];
    if (from_cache(&b, 4, s, std::string(), pos, 0, nullptr, nullptr)) return;
    const uint8_t *bytes = nullptr;
    size_t n = 0;
    Cb cb_;
    const int32_t st = ix->word_position_docids(ix->user, (const uint8_t *)s.data(), (uint32_t)s.size(), pos, &bytes, &n);
    take(b, st, bytes, n, "word_position_docids", 4, s, std::string(), pos, 0);
  }
  // sink of the word-prefix callbacks: every stored value goes straight into the decode batch
  struct Sink {
    Ctx *c;
    MsiCboBatch *b;
    bool bad = false;
    // posting-cache key of the i-th value the callback pushes: (db, s1, s2, x, i)
    uint32_t db = 0;
    const std::string *s1 = nullptr, *s2 = nullptr;
    uint64_t x = 0, i = 0;
  };
  static int32_t sink_push(void *sink, const uint8_t *bytes, size_t n) {
    Sink *s = (Sink *)sink;
    const uint64_t i = s->i++;
    if (!n || !bytes || !s->b) return 0;
    bool ok;
#ifndef MSI_SEARCH_DIRECT_ONLY
    static const std::string none;
    if (s->c->dev.pcache && s->db) {
      const std::string &a = s->s1 ? *s->s1 : none, &b2 = s->s2 ? *s->s2 : none;
      ok = s->c->dev.append_posting(*s->b, msi_cache_key(s->db, a.data(), a.size(), b2.data(), b2.size(), s->x, i, s->c->prm->index_view), bytes, n);
    } else
#endif
      ok = msi_cbo_batch_append(*s->b, bytes, n);
    (void)i;
    if (!ok) {
      s->bad = true;
      return -1;
    }
    return 0;
  }
  int32_t finish(Sink &s, int32_t st, const char *what) {
    if (st < 0 || s.bad) {
      msi_set_error("msi_keyword_search_ranked: %s callback failed or returned a malformed posting list (%d)", what, st);
      throw Fail{s.bad ? MSI_E_INVALID : MSI_E_INTERNAL};
    }
    return st;
  }
  int32_t add_prefix(MsiCboBatch *b, uint32_t w, bool original) {  // -> number of stored values (0: no such key)
    if (!ix->word_prefix_docids) return 0;
    const std::string &s = words[w];
    Sink sk{this, b};
    sk.db = original ? 6 : 7;
    sk.s1 = &s;
    Cb cb_;
    return finish(sk, ix->word_prefix_docids(ix->user, (const uint8_t *)s.data(), (uint32_t)s.size(), original ? 1 : 0,
                                             sink_push, &sk), "word_prefix_docids");
  }
  void add_prefix_key(MsiCboBatch &b, uint32_t w, int which, uint32_t key) {
    auto fn = which == 0 ? ix->word_prefix_fid_docids : ix->word_prefix_position_docids;
    if (!fn) fail(MSI_E_INVALID, "the index vtable has word_prefix_docids but not word_prefix_fid/position_docids");
    const std::string &s = words[w];
    Sink sk{this, &b};
    sk.db = which == 0 ? 8 : 9;
    sk.s1 = &s;
    sk.x = key;
    Cb cb_;
    finish(sk, fn(ix->user, (const uint8_t *)s.data(), (uint32_t)s.size(), key, sink_push, &sk), "word_prefix_*_docids");
  }
  void add_prefix_pair(MsiCboBatch &b, uint32_t prox, uint32_t w1, uint32_t prefix2) {
    if (!ix->word_prefix_pair_proximity_docids)
      fail(MSI_E_INVALID, "the index vtable has word_prefix_docids but not word_prefix_pair_proximity_docids");
    const std::string &l = words[w1], &r = words[prefix2];
    Sink sk{this, &b};
    sk.db = 10;
    sk.s1 = &l;
    sk.s2 = &r;
    sk.x = prox;
    Cb cb_;
    finish(sk, ix->word_prefix_pair_proximity_docids(ix->user, prox, (const uint8_t *)l.data(), (uint32_t)l.size(),
                                                      (const uint8_t *)r.data(), (uint32_t)r.size(), sink_push, &sk),
           "word_prefix_pair_proximity_docids");
  }

  struct SynSink {
    Ctx *c;
    Vec<Phrase> out;
  };
  static int32_t syn_push(void *sink, const msi_query_token *ws, uint32_t n) {
    SynSink *s = (SynSink *)sink;
    Phrase p;
    for (uint32_t i = 0; i < n; ++i) p.push_back((int32_t)s->c->word(std::string((const char *)ws[i].word, ws[i].len)));
    s->out.push_back(std::move(p));
    return 0;
  }
  Vec<Phrase> synonyms_of(const Vec<uint32_t> &ws) {
    if (!ix->synonyms) return {};
    Vec<msi_query_token> toks(ws.size());
    Vec<std::string> keep;
    for (uint32_t w : ws) keep.push_back(words[w]);
    for (size_t i = 0; i < ws.size(); ++i) toks[i] = msi_query_token{(const uint8_t *)keep[i].data(), (uint32_t)keep[i].size(), 0};
    SynSink sk{this, {}};
    Cb cb_;
    if (ix->synonyms(ix->user, toks.data(), (uint32_t)toks.size(), syn_push, &sk) < 0)
      fail(MSI_E_INTERNAL, "synonyms callback failed");
    return sk.out;
  }

  Vec<uint16_t> list_of(decltype(msi_index_vtable::word_fids) fn, uint32_t w, const char *what) {
    if (!fn) {
      msi_set_error("msi_keyword_search_ranked: the index vtable has no %s", what);
      throw Fail{MSI_E_INVALID};
    }
    const std::string &s = words[w];
    Vec<uint16_t> out(64);
    Cb cb_;
    for (;;) {
      uint32_t n = 0;
      if (fn(ix->user, (const uint8_t *)s.data(), (uint32_t)s.size(), out.data(), (uint32_t)out.size(), &n) < 0)
        fail(MSI_E_INTERNAL, "word_fids / word_positions callback failed");
      if (n <= out.size()) {
        out.resize(n);
        return out;
      }
      out.resize(n);
    }
  }

  uint32_t budget(const std::string &w) {  // number_of_typos_allowed, parse_query.rs:204-225 (chars, not bytes)
    uint32_t chars = 0;
    for (unsigned char c : w) chars += (c & 0xC0) != 0x80;
    const bool exact = ix->is_exact_word && ix->is_exact_word(ix->user, (const uint8_t *)w.data(), (uint32_t)w.size()) > 0;
    if (!prm->authorize_typos || chars < prm->min_word_len_one_typo || exact) return 0;
    return chars < prm->min_word_len_two_typos ? 1 : 2;
  }

  // partially_initialized_term_from_word, compute_derivations.rs:170-253 (no prefix DB, no synonyms)
  Term term_from_word(const std::string &w, uint32_t max_typo, bool is_prefix, bool is_ngram = false) {
    Term t;
    t.original = word(w);
    if (w.size() > MAX_WORD_LENGTH) {
      t.too_long = true;
      return t;
    }
    t.max_lev = max_typo;
    t.is_prefix = is_prefix;
    if (contains_word(t.original)) t.exact = (int32_t)t.original;  // Index::contains_word
    {  // synonyms of the word: at most 50 phrases and 100 words in total (:217-236)
      uint32_t n_words = 0, n_phr = 0;
      for (Phrase &syn : synonyms_of({t.original})) {
        if (n_phr++ >= MAX_SYNONYM_PHRASE_COUNT) break;
        if (n_words + syn.size() > MAX_SYNONYM_WORD_COUNT) continue;
        n_words += (uint32_t)syn.size();
        t.synonyms.push_back(phrase(syn));
      }
    }
    // word_prefix_docids has the word, or (not for n-grams) exact_word_prefix_docids (:193-205)
    if (is_prefix && add_prefix(nullptr, t.original, !is_ngram) > 0) t.use_prefix_db = (int32_t)t.original;
    if (is_prefix && t.use_prefix_db < 0) {
      // find_zero_typo_prefix_derivations :40-73: the keys of word_docids (= the dictionary) merged in key order with
      // the keys of exact_word_docids, the word itself left out, at most 1000
      uint32_t lo = 0, hi = 0;
      msi_dict_prefix_range(dict, (const uint8_t *)w.data(), (uint32_t)w.size(), &lo, &hi);
      Vec<std::string> exact;
      if (ix->exact_words_with_prefix) {
        SynSink sk{this, {}};
        Cb cb_;
        if (ix->exact_words_with_prefix(ix->user, (const uint8_t *)w.data(), (uint32_t)w.size(), syn_push, &sk) < 0)
          fail(MSI_E_INTERNAL, "exact_words_with_prefix callback failed");
        for (Phrase &p : sk.out)
          if (!p.empty() && p[0] >= 0) exact.push_back(words[(uint32_t)p[0]]);
      }
      size_t ei = 0;
      uint32_t di = lo;
      while ((di < hi || ei < exact.size()) && t.prefix_of.size() < MAX_PREFIX_COUNT) {
        std::string next;
        if (di < hi) {
          const uint8_t *dw;
          uint32_t dl;
          msi_dict_word(dict, di, &dw, &dl);
          next.assign((const char *)dw, dl);
        }
        if (di >= hi || (ei < exact.size() && exact[ei] < next)) {
          next = exact[ei++];
        } else {
          if (ei < exact.size() && exact[ei] == next) ++ei;
          ++di;
        }
        if (next == w) continue;
        t.prefix_of.push_back(word(next));
      }
    }
    return t;
  }

  // compute_fully_if_needed for every term of the query: ONE batched device dictionary lookup
  void compute_derivations() {
    Vec<msi_typo_query> tq;
    Vec<uint32_t> owner;
    for (uint32_t ti = 0; ti < terms.size(); ++ti) {
      Term &t = terms[ti];
      if (t.phrase >= 0 || t.too_long || t.max_lev == 0) continue;
      const std::string &w = words[t.original];
      msi_typo_query q;
      q.word = (const uint8_t *)w.data();
      q.len = (uint32_t)w.size();
      q.max_typos = (uint8_t)t.max_lev;
      q.is_prefix = t.is_prefix ? 1 : 0;
      q._pad = 0;
      tq.push_back(q);
      owner.push_back(ti);
    }
    Vec<uint32_t> one(tq.size() * MAX_ONE_TYPO_COUNT), two(tq.size() * MAX_TWO_TYPOS_COUNT), n1(tq.size()),
        n2(tq.size());
    if (!tq.empty())
      ck(msi_dict_lookup(dict, tq.data(), (uint32_t)tq.size(), MAX_ONE_TYPO_COUNT, MAX_TWO_TYPOS_COUNT, one.data(),
                         n1.data(), two.data(), n2.data()));
    auto dict_word = [&](uint32_t idx) {
      const uint8_t *dw;
      uint32_t dl;
      msi_dict_word(dict, idx, &dw, &dl);
      return word(std::string((const char *)dw, dl));
    };
    for (size_t k = 0; k < owner.size(); ++k) {
      Term &t = terms[owner[k]];
      for (uint32_t i = 0; i < n1[k]; ++i) t.one_typo.push_back(dict_word(one[k * MAX_ONE_TYPO_COUNT + i]));
      if (t.max_lev > 1)
        for (uint32_t i = 0; i < n2[k]; ++i) t.two_typos.push_back(dict_word(two[k * MAX_TWO_TYPOS_COUNT + i]));
    }
    // split words: the most frequent adjacent pair (compute_derivations.rs:363-383), part of the
    // one-typo subterm for every budget; not for phrases; an n-gram (budget <= 1) is not split back
    // into its own words (:296-309)
    for (Term &t : terms) {
      if (t.phrase >= 0 || t.too_long || !ix->word_pair_proximity_docids) continue;
      const std::string w = words[t.original];
      uint64_t best = 0;
      size_t at = 0;
      for (size_t i = 1; i < w.size(); ++i) {
        if ((w[i] & 0xC0) == 0x80) continue;
        const uint64_t f = add_pair(nullptr, 1, word(w.substr(0, i)), word(w.substr(i)));
        if (f > best) {
          best = f;
          at = i;
        }
      }
      if (!best) continue;
      Phrase sp{(int32_t)word(w.substr(0, at)), (int32_t)word(w.substr(at))};
      if (t.max_lev <= 1 && t.is_ngram && t.ngram_words.size() == 2 && (int32_t)t.ngram_words[0] == sp[0] &&
          (int32_t)t.ngram_words[1] == sp[1])
        continue;
      t.split_words = (int32_t)phrase(sp);
    }
  }

  // -- QueryTermSubset ---------------------------------------------------------------------------
  Subset full(uint32_t ti) {
    Subset s;
    s.term = ti;
    return s;
  }
  // -> (kind 0 none / 1 phrase / 2 word, id)
  std::pair<int, uint32_t> exact_term(const Subset &ss) {
    const Term &t = terms[ss.term];
    if (t.is_ngram) return {0, 0};
    if (t.phrase >= 0) return ss.zero.has_phrase((uint32_t)t.phrase) ? std::make_pair(1, (uint32_t)t.phrase) : std::make_pair(0, 0u);
    if (t.exact >= 0) return ss.zero.has_word((uint32_t)t.exact) ? std::make_pair(2, (uint32_t)t.exact) : std::make_pair(0, 0u);
    return {0, 0};
  }
  // QueryTermSubset::use_prefix_db, query_term/mod.rs:183-203 -> word or -1 (original? = !is_ngram)
  int32_t use_prefix_db(const Subset &ss) {
    const Term &t = terms[ss.term];
    if (t.use_prefix_db < 0 || !ss.zero.has_word((uint32_t)t.use_prefix_db)) return -1;
    return t.use_prefix_db;
  }
  OrdSet<std::pair<uint32_t, bool>> all_single_words(const Subset &ss) {  // (word, Word::Original?)
    const Term &t = terms[ss.term];
    const bool orig = !t.is_ngram;
    OrdSet<std::pair<uint32_t, bool>> out;
    if (ss.zero.kind != 0) {
      if (t.exact >= 0 && ss.zero.has_word((uint32_t)t.exact)) out.insert({(uint32_t)t.exact, orig});
      for (uint32_t w : t.prefix_of)
        if (ss.zero.has_word(w)) out.insert({w, orig});
    }
    if (ss.one.kind != 0)
      for (uint32_t w : t.one_typo)
        if (ss.one.has_word(w)) out.insert({w, false});
    if (ss.two.kind != 0)
      for (uint32_t w : t.two_typos)
        if (ss.two.has_word(w)) out.insert({w, false});
    return out;
  }
  IdSet all_phrases(const Subset &ss) {
    const Term &t = terms[ss.term];
    IdSet out;
    if (t.phrase >= 0) out.insert((uint32_t)t.phrase);  // regardless of the zero-typo subset, as the reference
    for (uint32_t p : t.synonyms) out.insert(p);
    if (ss.one.kind != 0 && t.split_words >= 0 && ss.one.has_phrase((uint32_t)t.split_words))
      out.insert((uint32_t)t.split_words);
    return out;
  }
  int32_t original_phrase(const Subset &ss) {
    const Term &t = terms[ss.term];
    return (t.phrase >= 0 && ss.zero.has_phrase((uint32_t)t.phrase)) ? t.phrase : -1;
  }
  uint32_t max_typo_cost(const Subset &ss) {  // query_term/mod.rs:340-370
    const Term &t = terms[ss.term];
    if (t.max_lev == 0) return t.phrase < 0 ? 1 : 0;
    if (t.max_lev == 1) return ss.one.is_empty() ? 0 : 1;
    if (ss.two.is_empty()) return ss.one.is_empty() ? 0 : 1;
    return 2;
  }
  Subset keep_only_exact_term(Subset ss) {
    auto e = exact_term(ss);
    if (e.first == 0) return ss;
    NTypo z{2, {}, {}};
    if (e.first == 1) z.phrases.insert(e.second);
    else z.words.insert(e.second);
    ss.zero = z;
    ss.one = NT_NONE;
    ss.two = NT_NONE;
    return ss;
  }

  // -- docids (resolve_query_graph.rs), all on the device -----------------------------------------
  // Every set returned by the *_full functions is universe-independent, cached for the whole search and
  // SHARED: callers never modify it, they intersect it into a set of their own.
  Set empty_;
  Set empty_set() {
    if (!empty_) empty_ = dev.zeros();
    return empty_;
  }
  Set word_full(uint32_t w, bool original) {
    auto it = word_cache.find({w, original});
    if (it == word_cache.end()) {
      relieve();
      MsiCboBatch b;
      add_word(b, w, original);
      it = word_cache.emplace(std::make_pair(w, original), dev.decode(b)).first;
    }
    return it->second;
  }
  Set pair_union(uint32_t w1, uint32_t w2, uint32_t max_prox) {  // union of pair(prox = 1..max_prox), fresh
    MsiCboBatch b;
    for (uint32_t p = 1; p <= max_prox; ++p) add_pair(&b, p, w1, w2);
    return dev.decode(b);
  }
  // compute_phrase_docids :187-268.  The result is the intersection of the words' documents with, for
  // every window of <= 3 words and every pair (i < j) in it, the documents where the pair is within
  // j - i; the reference's early returns are all "the intersection is already empty".
  Set phrase_docids(uint32_t pid) {
    auto it = phrase_cache.find(pid);
    if (it != phrase_cache.end()) return it->second;
    const Phrase p = phrases[pid];
    Set cand;
    for (int32_t w : p) {
      if (w < 0) continue;
      if (!cand) cand = dev.clone(word_full((uint32_t)w, true));
      else dev.and_(cand, word_full((uint32_t)w, true));
    }
    if (!cand) cand = dev.zeros();
    const size_t win = std::min<size_t>(p.size(), 3);
    for (size_t s = 0; s + win <= p.size() && win > 0; ++s)
      for (size_t a = 0; a < win; ++a) {
        if (p[s + a] < 0) continue;
        for (size_t b = a + 1; b < win; ++b) {
          if (p[s + b] < 0) continue;
          Set m = pair_union((uint32_t)p[s + a], (uint32_t)p[s + b], (uint32_t)(b - a));
          dev.and_(cand, m);
        }
      }
    return phrase_cache[pid] = cand;
  }
  // compute_query_term_subset_docids :33-59 without the universe
  Set subset_full(const Subset &ss) {
    Subset key = ss;
    key.mandatory = false;
    auto it = subset_cache.find(key);
    if (it == subset_cache.end()) {
      relieve();
      MsiCboBatch b;
      uint64_t floor = 0;   // the subset's documents are at least those of its most frequent word
      for (auto &w : all_single_words(ss)) {
        uint64_t card = 0;
        add_word(b, w.first, w.second, &card);
        floor = std::max(floor, card);
      }
      const int32_t pf = use_prefix_db(ss);
      if (pf >= 0) add_prefix(&b, (uint32_t)pf, !terms[ss.term].is_ngram);
      Set d = dev.decode(b);
      for (uint32_t p : all_phrases(ss)) dev.or_(d, phrase_docids(p));
      it = subset_cache.emplace(key, d).first;
      subset_floor[key] = floor;
    }
    return it->second;
  }
  // a lower bound of |subset_full(ss)| known on the host (0: nothing known) — valid once subset_full(ss) has been called
  Map<Subset, uint64_t> subset_floor;
  uint64_t floor_of(const Subset &ss) {
    Subset key = ss;
    key.mandatory = false;
    auto it = subset_floor.find(key);
    return it == subset_floor.end() ? 0 : it->second;
  }
  // ..._within_field_id / ..._within_position :61-130 for one fid (which 0) or a set of positions (which 1):
  // every posting of the condition goes into ONE decode batch
  Set within_full(const Subset &ss, int which, const Vec<uint32_t> &keys) {
    Subset sk = ss;
    sk.mandatory = false;
    auto ck_ = std::make_tuple(sk, which, keys);
    auto it = within_cache.find(ck_);
    if (it != within_cache.end()) return it->second;
    relieve();
    MsiCboBatch b;
    const int32_t pf = use_prefix_db(ss);
    for (uint32_t key : keys) {
      for (auto &w : all_single_words(ss)) {
        if (which == 0) add_word_fid(b, w.first, key);
        else add_word_position(b, w.first, key);
      }
      if (pf >= 0) add_prefix_key(b, (uint32_t)pf, which, key);
    }
    Set d = dev.decode(b);
    for (uint32_t p : all_phrases(ss)) {
      int32_t first = -1;
      for (int32_t w : phrases[p])
        if (w >= 0) {
          first = w;
          break;
        }
      if (first < 0) continue;
      MsiCboBatch fb;
      for (uint32_t key : keys) {
        if (which == 0) add_word_fid(fb, (uint32_t)first, key);
        else add_word_position(fb, (uint32_t)first, key);
      }
      Set f = dev.decode(fb);
      dev.and_(f, phrase_docids(p));
      dev.or_(d, f);
    }
    return within_cache.emplace(ck_, d).first->second;
  }
};

// ---- QueryGraph (query_graph.rs) -----------------------------------------------------------------
void build_initial_edges(Graph &g) {
  for (GNode &n : g.nodes) {
    n.preds.clear();
    n.succs.clear();
  }
  for (uint32_t i = 0; i < g.nodes.size(); ++i) {
    GNode &n = g.nodes[i];
    int end_prev;
    if (n.kind == 2) end_prev = (int)n.term.id_hi;
    else if (n.kind == 0) end_prev = -1;
    else continue;
    IdSet succ;
    int mn = 1 << 30;
    for (uint32_t j = 0; j < g.nodes.size(); ++j) {
      const GNode &m = g.nodes[j];
      int st;
      if (m.kind == 2) st = (int)m.term.id_lo;
      else if (m.kind == 1) st = 1 << 29;
      else continue;
      if (st <= end_prev) continue;
      if (st < mn) {
        mn = st;
        succ.clear();
        succ.insert(j);
      } else if (st == mn) {
        succ.insert(j);
      }
    }
    n.succs = succ;
    for (uint32_t j : succ) g.nodes[j].preds.insert(i);
  }
}

void remove_nodes_keep_edges(Graph &g, const Vec<uint32_t> &ids) {
  for (uint32_t i : ids) {
    const IdSet pr = g.nodes[i].preds, su = g.nodes[i].succs;
    for (uint32_t p : pr) {
      g.nodes[p].succs.erase(i);
      g.nodes[p].succs.insert(su.begin(), su.end());
    }
    for (uint32_t s : su) {
      g.nodes[s].preds.erase(i);
      g.nodes[s].preds.insert(pr.begin(), pr.end());
    }
    g.nodes[i] = GNode();
    g.nodes[i].kind = 3;
  }
}

// removal_order_for_terms_matching_strategy :377-406 — groups of nodes by ascending cost (the largest `order` over the
// node's term ids), first removed first; the last group stays unless a phrase / mandatory term keeps the query alive
template <typename Order>
Vec<IdSet> removal_order(Ctx &c, const Graph &g, Order order) {
  Map<uint32_t, IdSet> groups;
  bool mandatory = false;
  for (uint32_t i = 0; i < g.nodes.size(); ++i) {
    const GNode &n = g.nodes[i];
    if (n.kind != 2) continue;
    if (c.original_phrase(n.term.subset) >= 0 || n.term.subset.mandatory) {
      mandatory = true;
      continue;
    }
    uint32_t cost = 0;
    for (uint32_t id = n.term.id_lo; id <= n.term.id_hi; ++id) cost = std::max(cost, (uint32_t)order(id));
    groups[cost].insert(i);
  }
  Vec<IdSet> res;
  for (auto &kv : groups) res.push_back(kv.second);
  if (!mandatory && !res.empty()) res.pop_back();
  return res;
}

// removal_order_for_terms_matching_strategy_last :346-375
Vec<IdSet> removal_order_last(Ctx &c, const Graph &g) {
  uint32_t first = 255, last = 0;
  for (const GNode &n : g.nodes)
    if (n.kind == 2) {
      last = std::max(last, n.term.id_hi);
      first = std::min(first, n.term.id_lo);
    }
  if (first >= last) return {};
  return removal_order(c, g, [last](uint32_t id) { return 1 + last - id; });
}

// removal_order_for_terms_matching_strategy_frequency :303-344 — per term id the number of documents of the union of
// the nodes that cover it (n-gram nodes count for each of their ids; no document at all counts as the LARGEST
// frequency); the most frequent term gets weight 1 and is removed first, equal frequencies share a weight.  The unions
// and all their cardinalities are ONE command list (VM_OP ... VM_COUNT x terms): one completion wait per call.
Vec<IdSet> removal_order_frequency(Ctx &c, const Graph &g) {
  // The frequencies are those of the whole index (universe None) and the graph is the query's own every time — Words is
  // the first graph-based rule, so nothing above it rebuilds the graph: computed once per search.  (Computed again after
  // the search moved into the compact space they would be frequencies inside U0.)
  if (!c.freq_weight.empty())
    return removal_order(c, g, [&c](uint32_t id) { return c.freq_weight.at(id); });
  Map<uint32_t, Set> term_docids;
  for (const GNode &n : g.nodes) {
    if (n.kind != 2) continue;
    Set d = c.subset_full(n.term.subset);     // compute_query_term_subset_docids(ctx, None, subset): cached, never written
    for (uint32_t id = n.term.id_lo; id <= n.term.id_hi; ++id) {
      auto it = term_docids.find(id);
      if (it == term_docids.end()) term_docids.emplace(id, c.dev.clone(d));
      else c.dev.or_(it->second, d);
    }
  }
  Vec<Set> sets;
  Vec<std::pair<uint32_t, uint64_t>> tf;
  for (auto &kv : term_docids) {
    tf.push_back({kv.first, 0});
    sets.push_back(kv.second);
  }
  const Vec<uint64_t> counts = c.dev.count_many(sets);
  for (size_t i = 0; i < tf.size(); ++i) tf[i].second = counts[i] ? counts[i] : ~0ull;
  std::stable_sort(tf.begin(), tf.end(), [](const std::pair<uint32_t, uint64_t> &a, const std::pair<uint32_t, uint64_t> &b) {
    return a.second > b.second;               // sort_by_key(Reverse(frequency)): stable over ascending term ids
  });
  Map<uint32_t, uint32_t> &weight_of = c.freq_weight;
  uint32_t weight = 1;
  for (size_t i = 0; i < tf.size(); ++i) {
    weight_of[tf[i].first] = weight;
    if (i + 1 < tf.size() && tf[i + 1].second != tf[i].second) ++weight;
  }
  return removal_order(c, g, [&weight_of](uint32_t id) { return weight_of.at(id); });
}

Vec<IdSet> removal_order_of(Ctx &c, const Graph &g, int strategy) {
  if (strategy == MSI_TERMS_LAST) return removal_order_last(c, g);
  if (strategy == MSI_TERMS_FREQUENCY) return removal_order_frequency(c, g);
  return {};
}

uint32_t words_in_phrases_count(Ctx &c, const Graph &g) {
  uint32_t n = 0;
  for (const GNode &nd : g.nodes)
    if (nd.kind == 2) {
      const int32_t p = c.original_phrase(nd.term.subset);
      if (p >= 0)
        for (int32_t w : c.phrases[(uint32_t)p]) n += w >= 0;
    }
  return n;
}

// QueryGraph::build_from_paths :470-544 (nodes shared by (term, suffix of the path))
Graph build_from_paths(const Vec<PathSubsets> &paths) {
  Vec<Vec<Located>> singles;
  for (const PathSubsets &path : paths) {
    Vec<Located> out;
    bool has_prev = false;
    Located prev;
    for (const auto &cond : path) {
      const bool has_start = cond.first.first;
      Located start = cond.first.second;
      if (has_prev) {
        if (has_start) {
          if (start.id_lo == prev.id_lo && start.id_hi == prev.id_hi) {
            start.subset.zero.intersect(prev.subset.zero);
            start.subset.one.intersect(prev.subset.one);
            start.subset.two.intersect(prev.subset.two);
            out.push_back(start);
          } else {
            out.push_back(prev);
            out.push_back(start);
          }
        } else {
          out.push_back(prev);
        }
      } else if (has_start) {
        out.push_back(start);
      }
      prev = cond.second;
      has_prev = true;
    }
    if (has_prev) out.push_back(prev);
    singles.push_back(std::move(out));
  }
  Graph g;
  g.nodes.resize(2);
  g.nodes[0].kind = 0;
  g.nodes[1].kind = 1;
  Map<Vec<Located>, uint32_t> ids;  // keyed by the suffix starting at the term
  Vec<Vec<uint32_t>> id_paths;
  for (const auto &path : singles) {
    Vec<uint32_t> p;
    for (size_t k = 0; k < path.size(); ++k) {
      Vec<Located> suffix(path.begin() + k, path.end());
      auto it = ids.find(suffix);
      if (it == ids.end()) {
        GNode n;
        n.kind = 2;
        n.term = path[k];
        g.nodes.push_back(n);
        it = ids.emplace(std::move(suffix), (uint32_t)g.nodes.size() - 1).first;
      }
      p.push_back(it->second);
    }
    id_paths.push_back(std::move(p));
  }
  for (const auto &p : id_paths) {
    uint32_t prev = 0;
    for (uint32_t i : p) {
      g.nodes[prev].succs.insert(i);
      g.nodes[i].preds.insert(prev);
      prev = i;
    }
    g.nodes[prev].succs.insert(1);
    g.nodes[1].preds.insert(prev);
  }
  return g;
}

// compute_query_graph_docids :133-185
// `universe` may be null: "every document" (no filter, no negative term) — intersecting with it and uniting into it are
// then no operations at all, and the result may be null for the same reason (a graph whose root reaches its end).
// What comes back is only ever READ by the caller: it may be one of the search's shared, cached sets (a term subset's
// documents) or `universe` itself.  (As first written — a zeroed set per node, united with every predecessor, the root a
// copy of the universe — a one-word search recorded a fill, five set operations and two clears over the whole index
// before its first wait, a three-word search eleven and four; the same documents take none and three.)
// *at_least (when asked for): a lower bound of the result's cardinality known on the host without waiting for the device — the
// most frequent word of a term that is united into it (0: nothing known)
Set query_graph_docids(Ctx &c, const Graph &g, const Set &universe, uint64_t *at_least = nullptr) {
  IdSet resolved;
  Map<uint32_t, Set> docs;       // (a null entry: every document)
  Map<uint32_t, uint64_t> floor; // lower bound of |docs[i]| (only tracked through "every document" and unions)
  if (at_least) *at_least = 0;
  Vec<uint32_t> queue{Graph::ROOT};
  size_t guard = 0;
  while (!queue.empty()) {
    if (++guard > 100000) fail(MSI_E_INTERNAL, "query graph is not a DAG");
    const uint32_t i = queue.front();
    queue.erase(queue.begin());
    const GNode &n = g.nodes[i];
    bool ready = true;
    for (uint32_t p : n.preds) ready &= resolved.count(p) != 0;
    if (!ready) {
      queue.push_back(i);
      continue;
    }
    // the union of the predecessors' documents: nothing (no predecessor), one of them as it is, or a set of its own
    Set pd;
    bool everything = false, first = true;
    uint64_t pd_floor = 0;   // a union holds at least what its largest member holds
    for (uint32_t p : n.preds) {
      everything = everything || !docs[p];
      pd_floor = std::max(pd_floor, floor[p]);
    }
    if (!everything) {
      for (uint32_t p : n.preds) {
        if (first) pd = n.preds.size() == 1 ? docs[p] : c.dev.clone(docs[p]);
        else c.dev.or_(pd, docs[p]);
        first = false;
      }
      if (n.preds.empty()) pd = c.empty_set();
    }
    Set nd;
    uint64_t nd_floor = 0;
    if (n.kind == 2) {
      const Set sub = c.subset_full(n.term.subset);
      nd = everything ? sub : c.dev.and_new(sub, pd, nullptr);
      if (everything) nd_floor = c.floor_of(n.term.subset);   // (an intersection with less than everything: nothing known)
    } else if (n.kind == 0) {
      nd = universe;
    } else if (n.kind == 1) {
      if (at_least) *at_least = everything ? 0 : pd_floor;
      return pd;   // (null when `everything`)
    } else {
      fail(MSI_E_INTERNAL, "deleted node reached");
    }
    resolved.insert(i);
    docs[i] = nd;
    floor[i] = nd_floor;
    for (uint32_t s : n.succs)
      if (!resolved.count(s) && std::find(queue.begin(), queue.end(), s) == queue.end()) queue.push_back(s);
  }
  fail(MSI_E_INTERNAL, "query graph has no end node");
}

// ---- rule plug-ins -------------------------------------------------------------------------------
uint32_t cost_from_distance(uint32_t d) {  // position/mod.rs:127-143
  static const uint32_t lim[] = {0, 1, 4, 7, 11, 16, 24, 64, 256, 1024};
  for (uint32_t c = 0; c < 10; ++c)
    if (d <= lim[c]) return c;
  return 10;
}

Vec<std::pair<uint32_t, Condition>> build_edges(Ctx &c, int kind, const Located *src, const Located &dst) {
  Vec<std::pair<uint32_t, Condition>> out;
  out.reserve(8);
  const uint32_t n = dst.n_ids();
  auto cond = [&](int k) {
    Condition x;
    x.kind = k;
    x.term = dst;
    return x;
  };
  switch (kind) {
    case R_WORDS:
      out.push_back({0, cond(C_TERM)});
      break;
    case R_TYPO: {  // typo/mod.rs:41-80
      const uint32_t base = n == 1 ? 0 : n;
      const uint32_t mx = c.max_typo_cost(dst.subset);
      for (uint32_t k = 0; k <= mx; ++k) {
        Condition x = cond(C_TYPO);
        if (k != 0) x.term.subset.zero = NT_NONE;
        if (k != 1) x.term.subset.one = NT_NONE;
        if (k != 2) x.term.subset.two = NT_NONE;
        x.x = k;
        out.emplace_back(k + base, std::move(x));
      }
      break;
    }
    case R_PROXIMITY: {  // proximity/build.rs:10-56
      const uint32_t ng = n - 1;
      if (!src || src->pos_hi + 1 != dst.pos_lo) {
        out.push_back({ng, cond(C_TERM)});
        break;
      }
      for (uint32_t cost = ng; cost < MAX_DISTANCE - 1 + ng; ++cost) {
        Condition x = cond(C_PROX);
        x.left = *src;
        x.has_left = true;
        x.x = cost + 1;
        out.emplace_back(cost, std::move(x));
      }
      out.push_back({MAX_DISTANCE - 1 + ng, cond(C_TERM)});
      break;
    }
    case R_FID: {  // fid/mod.rs:51-121
      OrdSet<uint16_t> fids;
      for (auto &w : c.all_single_words(dst.subset))
        for (uint16_t f : c.list_of(c.ix->word_fids, w.first, "word_fids")) fids.insert(f);
      for (uint32_t p : c.all_phrases(dst.subset))
        for (int32_t w : c.phrases[p])
          if (w >= 0)
            for (uint16_t f : c.list_of(c.ix->word_fids, (uint32_t)w, "word_fids")) fids.insert(f);
      {
        const int32_t pf = c.use_prefix_db(dst.subset);
        if (pf >= 0)
          for (uint16_t f : c.list_of(c.ix->word_prefix_fids, (uint32_t)pf, "word_prefix_fids")) fids.insert(f);
      }
      uint32_t cur_max = 0;
      for (uint16_t f : fids) {
        int32_t weight = -1;
        for (uint32_t i = 0; i < c.prm->n_searchable; ++i)
          if (c.prm->searchable_fids[i] == f) weight = c.prm->searchable_weights[i];
        if (weight < 0) continue;
        cur_max = std::max(cur_max, (uint32_t)weight);
        Condition x = cond(C_FID);
        x.x = f;
        x.has_fid = true;
        out.emplace_back((uint32_t)weight * n, std::move(x));
      }
      if (c.prm->max_weight >= 0 && cur_max < (uint32_t)c.prm->max_weight) {
        Condition x = cond(C_FID);
        out.emplace_back((uint32_t)c.prm->max_weight * n, std::move(x));
      }
      break;
    }
    case R_POSITION: {  // position/mod.rs:50-125
      OrdSet<uint16_t> positions;
      for (auto &w : c.all_single_words(dst.subset))
        for (uint16_t p : c.list_of(c.ix->word_positions, w.first, "word_positions")) positions.insert(p);
      for (uint32_t p : c.all_phrases(dst.subset))
        for (int32_t w : c.phrases[p])
          if (w >= 0) {
            for (uint16_t q : c.list_of(c.ix->word_positions, (uint32_t)w, "word_positions")) positions.insert(q);
            break;
          }
      {
        const int32_t pf = c.use_prefix_db(dst.subset);
        if (pf >= 0)
          for (uint16_t q : c.list_of(c.ix->word_prefix_positions, (uint32_t)pf, "word_prefix_positions")) positions.insert(q);
      }
      Map<uint32_t, Vec<uint16_t>> by_cost;
      for (uint16_t pos : positions) {
        const uint32_t dist = pos > dst.pos_lo ? pos - dst.pos_lo : dst.pos_lo - pos;
        uint32_t cost = 0;
        for (uint32_t i = 0; i < n; ++i) cost += cost_from_distance(dist + i);
        by_cost[cost].push_back(pos);
      }
      for (auto &kv : by_cost) {
        Condition x = cond(C_POSITION);
        x.positions = kv.second;
        out.emplace_back(kv.first, std::move(x));
      }
      if (!by_cost.count(n * 10)) out.push_back({n * 10, cond(C_POSITION)});
      break;
    }
    case R_EXACTNESS:  // exactness/mod.rs:73-87
      out.push_back({0, cond(C_EXACT)});
      out.push_back({n, cond(C_ANY)});
      break;
  }
  return out;
}

// proximity/compute_docids.rs:15-212 (no prefix DB)
Set proximity_full(Ctx &c, const Condition &cd) {
  const uint32_t rn = cd.term.n_ids();
  const uint32_t forward = 1 + cd.x - rn, backward = cd.x - rn;
  Subset lk = cd.left.subset, rk = cd.term.subset;
  lk.mandatory = rk.mandatory = false;
  auto key = std::make_tuple(lk, rk, forward, backward);
  auto hit = c.prox_cache.find(key);
  if (hit != c.prox_cache.end()) return hit->second;
  c.relieve();
  OrdSet<std::pair<int32_t, uint32_t>> lefts, rights;  // (phrase or -1, word)
  for (auto &w : c.all_single_words(cd.left.subset)) lefts.insert({-1, w.first});
  for (uint32_t p : c.all_phrases(cd.left.subset))
    if (c.phrases[p].back() >= 0) lefts.insert({(int32_t)p, (uint32_t)c.phrases[p].back()});
  for (auto &w : c.all_single_words(cd.term.subset)) rights.insert({-1, w.first});
  for (uint32_t p : c.all_phrases(cd.term.subset))
    if (c.phrases[p].front() >= 0) rights.insert({(int32_t)p, (uint32_t)c.phrases[p].front()});
  Set docids;   // (the first group's documents as they are: no zeroed set to unite them into)
  // all the word-word pairs resolve against the same universe: one decode launch for all of them
  Map<std::pair<int32_t, int32_t>, MsiCboBatch> groups;
  const int32_t pf = c.use_prefix_db(cd.term.subset);
  if (pf >= 0)  // compute_prefix_edges :97-147: (left word, right prefix) forward, (right prefix as a word, left word) backward
    for (auto &l : lefts) {
      MsiCboBatch &b = groups[{l.first, -1}];
      c.add_prefix_pair(b, forward, l.second, (uint32_t)pf);
      if (l.first < 0 && backward >= 1) c.add_pair(&b, backward, (uint32_t)pf, l.second);
    }
  for (auto &l : lefts)
    for (auto &r : rights) {
      MsiCboBatch &b = groups[{l.first, r.first}];
      c.add_pair(&b, forward, l.second, r.second);
      if (backward >= 1 && l.first < 0 && r.first < 0) c.add_pair(&b, backward, r.second, l.second);
    }
  for (auto &kv : groups) {
    if (kv.second.containers.empty() && kv.second.small_ids.empty()) continue;
    Set d = c.dev.decode(kv.second);
    if (kv.first.first >= 0) c.dev.and_(d, c.phrase_docids((uint32_t)kv.first.first));
    if (kv.first.second >= 0) c.dev.and_(d, c.phrase_docids((uint32_t)kv.first.second));
    if (!docids) docids = d;
    else c.dev.or_(docids, d);
  }
  if (!docids) docids = c.dev.zeros();
  c.prox_cache.emplace(key, docids);
  return docids;
}

// The documents of a condition WITHOUT the universe (shared, never modified): the path search intersects
// them with a prefix that is already inside the current universe.
Resolved resolve_condition(Ctx &c, const Condition &cd) {
  Resolved r;
  r.end = cd.term;
  switch (cd.kind) {
    case C_TERM:
    case C_TYPO:
    case C_ANY:
      r.docs = c.subset_full(cd.term.subset);
      break;
    case C_FID:
      r.docs = cd.has_fid ? c.within_full(cd.term.subset, 0, {cd.x}) : c.empty_set();
      break;
    case C_POSITION:
      r.docs = cd.positions.empty() ? c.empty_set()
                                    : c.within_full(cd.term.subset, 1, Vec<uint32_t>(cd.positions.begin(), cd.positions.end()));
      break;
    case C_EXACT: {
      r.end.subset = c.keep_only_exact_term(cd.term.subset);
      r.end.subset.mandatory = true;
      auto e = c.exact_term(cd.term.subset);
      if (e.first == 0) r.docs = c.empty_set();
      else if (e.first == 1) r.docs = c.phrase_docids(e.second);
      else r.docs = c.word_full(e.second, true);
      break;
    }
    case C_PROX:
      r.docs = proximity_full(c, cd);
      r.has_start = true;
      r.start = cd.left;
      break;
  }
  return r;
}

// ---- rules ---------------------------------------------------------------------------------------
struct Bucket {
  // The query graph of the bucket, for the rule below — wanted only when that rule actually ranks the bucket (not for the
  // last rule's buckets, empty ones, or those before / after the page), so it is assembled on demand: from the paths
  // that found documents (a graph rule), or it is the producing rule's own graph, which outlives the bucket.
  Graph owned;
  const Graph *same_as = nullptr;
  struct Rule *maker = nullptr;                 // the graph rule that knows what `good` names
  Vec<Vec<int32_t>> good;       // the paths (condition ids) that found documents
  inline const Graph &graph();
  Set docs;
  uint64_t count = 0;
  Score score{0, 0, 0};
  bool universe_reduced = false;  // the rule already removed `docs` from the universe it was given
  std::shared_ptr<Vec<uint32_t>> ids;   // the first min(count, Ctx::spec_k) documents, when the rule asked for them ahead
};

struct Rule {
  int kind;
  int tms;  // -1 none, MSI_TERMS_LAST, MSI_TERMS_ALL, MSI_TERMS_FREQUENCY
  bool leaf = false;   // the last rule of the list (set by the bucket sort's tree): its buckets go straight to the results
  // documents the page still takes from this rule's buckets (set by the bucket sort's tree before every next(); 0: not
  // known — the sequential loop): a bucket that fills it, or takes the rest of the universe, is the last one asked for,
  // and what is left of the universe afterwards is read by nobody
  uint64_t page_room = 0;
  Rule(int k, int t) : kind(k), tms(t) {}
  virtual ~Rule() {}
  virtual void start(Ctx &c, const Set &universe, const Graph &g) = 0;
  virtual bool next(Ctx &c, const Set &universe, uint64_t universe_count, Bucket &out) = 0;
  virtual void end() = 0;
  virtual Rule *fresh() const = 0;   // another instance of the same rule (the bucket sort's tasks each rank with their own)
  virtual Graph graph_of(const Vec<Vec<int32_t>> &) { return Graph(); }   // Bucket::graph
};

const Graph &Bucket::graph() {
  if (maker) {
    owned = maker->graph_of(good);
    maker = nullptr;
  }
  return same_as ? *same_as : owned;
}

struct Edge {
  uint32_t cost;
  int32_t cond;  // -1: unconditional
  uint32_t dest;
  IdSet skip;
};

struct GraphRule : Rule {
  Vec<std::pair<EdgeSet *, uint32_t>> conds;   // condition id -> its entry in an edge set of the query's memo
  Vec<std::shared_ptr<EdgeSet>> held;          // ... which this evaluation keeps alive
  Vec<Vec<Edge>> edges;
  Vec<Vec<uint64_t>> costs;
  Vec<char> costs_done;
  uint64_t next_max_cost = 1, cur_cost = 0;

  // per next_bucket state
  Ctx *cx = nullptr;
  Set uni, bucket;
  uint64_t uni_count = 0, bucket_count = 0;
  struct StackE {
    int32_t cond;
    Set docs;
    bool stale;
    uint64_t count;
  };
  Vec<StackE> stack;
  Vec<Vec<int32_t>> good;
  bool stop = false;
  uint64_t emit_epoch = 0;
  // cost levels evaluated ahead of their turn behind one completion wait (MSI_SEARCH_LEVELS_PER_WAIT > 1)
  struct Ready {
    uint64_t cost;
    Set bucket;
    uint64_t count;
    Vec<Vec<int32_t>> good;
    std::shared_ptr<Vec<uint32_t>> ids;
  };
  std::deque<Ready> ready;
  std::shared_ptr<Vec<uint32_t>> level_ids;   // fused_level: the ids asked for with the level (leaf rule)

  GraphRule(int k, int t) : Rule(k, t) {}
  Rule *fresh() const override { return new GraphRule(kind, tms); }

  void start(Ctx &c, const Set &, const Graph &g) override {
    conds.clear();
    held.clear();
    ready.clear();
    next_max_cost = 1;
    cur_cost = 0;
    Map<uint32_t, std::pair<uint32_t, IdSet>> skip_cost;
    if (tms >= 0) {
      const uint32_t wp = words_in_phrases_count(c, g);
      next_max_cost += wp > 0 ? wp - 1 : 0;
      if (tms == MSI_TERMS_LAST || tms == MSI_TERMS_FREQUENCY) {
        IdSet forbidden;
        for (auto &ns : removal_order_of(c, g, tms)) {
          for (uint32_t n : ns) skip_cost[n] = {1, forbidden};
          forbidden.insert(ns.begin(), ns.end());
        }
      }
    }
    // The reference interns conditions by value (DedupInterner).  The conditions of an edge are a function of the rule,
    // its destination term and — for proximity between adjacent terms only — its source term, and one build_edges call
    // never yields the same condition twice: interning per edge set of the query's memo (keyed by exactly those values)
    // hands out the same ids without comparing condition values.
    Map<EdgeSet *, Vec<std::pair<uint32_t, int32_t>>> interned;
    edges.assign(g.nodes.size(), {});
    for (uint32_t i = 0; i < g.nodes.size(); ++i) {
      const GNode &n = g.nodes[i];
      if (n.kind != 0 && n.kind != 2) continue;
      auto push = [&](Edge e) {  // (dest, cost, cond) is Edge equality, mod.rs:62-70; a node has a handful of edges
        for (const Edge &x : edges[i])
          if (x.dest == e.dest && x.cost == e.cost && x.cond == e.cond) return;
        edges[i].push_back(std::move(e));
      };
      for (uint32_t d : n.succs) {
        const GNode &dn = g.nodes[d];
        if (dn.kind == 1) {
          push(Edge{0, -1, d, {}});
          continue;
        }
        auto sk = skip_cost.find(d);
        if (sk != skip_cost.end()) push(Edge{sk->second.first * dn.term.n_ids(), -1, d, sk->second.second});
        const Located *src = n.kind == 2 ? &n.term : nullptr;
        const bool by_source = kind == R_PROXIMITY && src && src->pos_hi + 1 == dn.term.pos_lo;
        auto known = c.edge_memo.find(EdgeKeyRef{kind, &dn.term, by_source, src});
        if (known == c.edge_memo.end()) {
          auto fresh_set = msi_arena::make_shared<EdgeSet>();
          fresh_set->conds = build_edges(c, kind, src, dn.term);
          fresh_set->resolved.resize(fresh_set->conds.size());
          known = c.edge_memo.emplace(EdgeKey{kind, dn.term, by_source, by_source ? *src : Located()}, std::move(fresh_set)).first;
        }
        EdgeSet *es = known->second.get();
        auto memo = interned.find(es);
        if (memo == interned.end()) {
          held.push_back(known->second);
          Vec<std::pair<uint32_t, int32_t>> ids;
          for (uint32_t k = 0; k < es->conds.size(); ++k) {
            conds.push_back({es, k});
            ids.push_back({es->conds[k].first, (int32_t)conds.size() - 1});
          }
          memo = interned.emplace(es, std::move(ids)).first;
        }
        for (auto &ce : memo->second) push(Edge{ce.first, ce.second, d, {}});
      }
    }
    costs.assign(g.nodes.size(), {});
    costs_done.assign(g.nodes.size(), 0);
    const auto &rc = costs_to_end(Graph::ROOT);
    next_max_cost += rc.empty() ? 0 : rc.back();
  }

  const Vec<uint64_t> &costs_to_end(uint32_t i) {  // find_all_costs_to_end, cheapest_paths.rs:312-340
    if (costs_done[i]) return costs[i];
    costs_done[i] = 1;
    if (i == Graph::END) {
      costs[i] = {0};
      return costs[i];
    }
    Vec<uint64_t> out;
    for (const Edge &e : edges[i])
      for (uint64_t cc : costs_to_end(e.dest)) out.push_back(e.cost + cc);
    std::sort(out.begin(), out.end());
    out.erase(std::unique(out.begin(), out.end()), out.end());
    costs[i] = std::move(out);
    return costs[i];
  }

  bool next(Ctx &c, const Set &universe, uint64_t universe_count, Bucket &out) override {
    const auto &rc = costs[Graph::ROOT];
    auto it = std::lower_bound(rc.begin(), rc.end(), cur_cost);
    if (it == rc.end()) return false;
    const uint64_t cost = *it;
    cur_cost = cost + 1;
    const uint32_t rank = (uint32_t)(next_max_cost - cost), mx = (uint32_t)next_max_cost;
    switch (kind) {  // rank_to_score of each rule + score_details.rs:440-509
      case R_WORDS: out.score = {MSI_SCORE_WORDS, rank, mx}; break;
      case R_TYPO: out.score = {MSI_SCORE_TYPO, mx - rank, mx - 1}; break;
      case R_PROXIMITY: out.score = {MSI_SCORE_PROXIMITY, rank, mx}; break;
      case R_FID: out.score = {MSI_SCORE_FID, rank, mx}; break;
      case R_POSITION: out.score = {MSI_SCORE_POSITION, rank, mx}; break;
      default: out.score = {MSI_SCORE_EXACT_WORDS, rank > 0 ? rank - 1 : 0, mx > 0 ? mx - 1 : 0}; break;
    }
    cx = &c;
    uni = universe;  // worked on in place: what is left after the paths claimed their documents IS universe - bucket
    uni_count = universe_count;
    bucket = c.dev.zeros();
    bucket_count = 0;
    stack.clear();
    good.clear();
    stop = false;
    if (known_outcome(c)) {
      // ONE term, ONE path, ONE condition — "the document holds the term" — and every document of the universe does: the
      // universe of any rule is a subset of the search's, which is query_graph_docids of this very graph (or of a graph this
      // one was reduced from, by paths whose documents are the universe).  The bucket IS the universe: no command, no wait.
      // (the Words and Proximity evaluations of a one-word search: two of its eight dependent rounds)
      good.clear();
      good.push_back({edges[Graph::ROOT][0].cond});
      ++g_stats.paths;
      out.maker = this;
      out.good = std::move(good);
      out.docs = universe;
      out.count = universe_count;
      out.universe_reduced = true;   // (nothing is left of it: the bucket sort's tree does not read it again)
      uni.reset();
      bucket.reset();
      return true;
    }
    if (ready.empty() || ready.front().cost != cost) {
      ready.clear();
      look_ahead(it, rc.end());
    }
    if (!ready.empty()) {
      // this level ran on a copy of the universe: take its documents out of the real one now
      Ready r = std::move(ready.front());
      ready.pop_front();
      bucket = r.bucket;
      bucket_count = r.count;
      good = std::move(r.good);
      out.ids = std::move(r.ids);
      const bool last_asked_for = page_room && (bucket_count == uni_count || bucket_count >= page_room);
      if (bucket_count && !last_asked_for) c.dev.sub_(uni, bucket);
    } else {
      IdSet visited, to_skip;
      level_ids.reset();
      if (!fused_level(cost)) visit(Graph::ROOT, cost, visited, to_skip);
      out.ids = std::move(level_ids);
    }
    out.maker = this;
    out.good = std::move(good);
    out.docs = bucket;
    out.count = bucket_count;
    out.universe_reduced = true;
    uni.reset();
    bucket.reset();
    stack.clear();
    return true;
  }

  // The Words and Proximity rules over a graph of one term node (root -> term -> end: one conditional edge of one cost) inside
  // the bucket sort's tree (page_room is only set there: the sequential loop keeps reading the universe a rule was given).
  // MSI_SEARCH_KNOWN_OUTCOMES=0: off (the tests hold both against the oracle).
  bool known_outcome(Ctx &c) const {
    const bool off = c.knob_known_off;
    // (Typo too when the term has ONE typo level — a word too short for typos, an exact term: its one condition holds the
    // term's zero-typo derivations, which are all the term has)
    if (off || !page_room || !c.dev.vm || (kind != R_WORDS && kind != R_PROXIMITY && kind != R_TYPO)) return false;
    if (costs[Graph::ROOT].size() != 1 || edges[Graph::ROOT].size() != 1) return false;
    const Edge &e = edges[Graph::ROOT][0];
    if (e.cond < 0 || !e.skip.empty() || e.dest == Graph::END || e.dest >= edges.size()) return false;
    const auto &out_edges = edges[e.dest];
    return out_edges.size() == 1 && out_edges[0].dest == Graph::END && out_edges[0].cond < 0 && out_edges[0].cost == 0;
  }

  Graph graph_of(const Vec<Vec<int32_t>> &found) override {
    Vec<PathSubsets> paths;
    paths.reserve(found.size());
    for (auto &p : found) {
      PathSubsets ps;
      ps.reserve(p.size());
      for (int32_t ci : p) {
        const Resolved &r = resolved(ci);
        ps.push_back({{r.has_start, r.start}, r.end});
      }
      paths.push_back(std::move(ps));
    }
    return build_from_paths(paths);
  }

  // All the paths of this cost, in the order the search would visit them, WITHOUT evaluating anything
  // (cheapest_paths.rs:147-310 minus the dead-end pruning).  False when there are too many for one launch.
  struct PathList {  // path k = conds[off[k] .. off[k+1])
    Vec<uint32_t> off{0};
    Vec<int32_t> conds;
    size_t size() const { return off.size() - 1; }
    bool empty() const { return off.size() == 1; }
    Vec<int32_t> path(size_t k) const { return {conds.begin() + off[k], conds.begin() + off[k + 1]}; }
  };
  bool enumerate(uint32_t node, uint64_t remaining, IdSet &visited, const IdSet &to_skip,
                 Vec<int32_t> &cur, PathList &out, size_t &steps) {
    for (const Edge &e : edges[node]) {
      if (remaining < e.cost) continue;
      const uint64_t rem = remaining - e.cost;
      const auto &dc = costs[e.dest];
      if (!std::binary_search(dc.begin(), dc.end(), rem)) continue;
      if (e.cond < 0) {
        if (e.dest == Graph::END) {
          steps += cur.size();
          if (out.size() >= MSI_BITS_MAX_PATHS || steps > MSI_BITS_MAX_STEPS) return false;
          out.conds.insert(out.conds.end(), cur.begin(), cur.end());
          out.off.push_back((uint32_t)out.conds.size());
        } else {
          IdSet ts;
          if (!e.skip.empty()) { ts = to_skip; ts.insert(e.skip.begin(), e.skip.end()); }
          if (!enumerate(e.dest, rem, visited, e.skip.empty() ? to_skip : ts, cur, out, steps)) return false;
        }
        continue;
      }
      if (to_skip.count(e.dest)) continue;
      bool blocked = false;
      for (uint32_t s : e.skip) blocked |= visited.count(s) != 0;
      if (blocked) continue;
      cur.push_back(e.cond);
      visited.insert(e.dest);
      IdSet ts;
      if (!e.skip.empty()) { ts = to_skip; ts.insert(e.skip.begin(), e.skip.end()); }
      const bool ok = enumerate(e.dest, rem, visited, e.skip.empty() ? to_skip : ts, cur, out, steps);
      visited.erase(e.dest);
      cur.pop_back();
      if (!ok) return false;
    }
    return true;
  }

  // MSI_SEARCH_LEVELS_PER_WAIT=n (2..4; default 1 = off): the next n cost levels are enqueued back to back on a
  // COPY of the universe — level k+1 claims from what level k left, exactly what the level-by-level iteration
  // would hand it — and their per-path counts come back behind one completion wait instead of n.  The real universe
  // loses a level's documents when bucket_sort asks for that level, so a search that stops early, a deadline or a
  // score threshold see the same universe as without the look-ahead.  Fills `ready` or leaves it empty (a level
  // that does not fit the in-argument kernel: the caller runs the one-level path).
  void look_ahead(Vec<uint64_t>::const_iterator it, Vec<uint64_t>::const_iterator end) {
    // with command lists a level evaluated ahead is one more command in a list that is submitted anyway, so the
    // look-ahead is on by default there (waits per 3-term query 76 -> 41); the direct back end pays a launch per level.
    // 8 levels per wait since round 3 (4 before): the Position and Fid rules have 20+ cost levels with a few documents
    // each — waits per detailed 3-term query at 10 M documents 17.0 -> 15.1 (12: 14.7, 16: 14.6), the same conditions
    // resolved, 1.3 % more set-operand bytes
    // (the direct back end publishes each level's counts into one of MSI_BITS_PATH_REGIONS regions; a command list has
    // MSI_VM_MAX_COUNTS counts and takes as many levels as fit)
    // 12 since round 6: lists per fresh query 10.79 -> 10.6 at 10 M documents (profiles/r6_queues.log), 10.52 -> 10.29 on the
    // emulated corpus; 16 adds nothing (10.28)
    const int per_wait = std::min<int>(cx->knob_levels_per_wait >= 0 ? cx->knob_levels_per_wait : (cx->dev.vm ? 12 : 1),
                                       cx->dev.vm ? 16 : (int)MSI_BITS_PATH_REGIONS);
    if (per_wait < 2 || cx->knob_fused_off) return;
    // `distinct` removes documents from every universe of the stack whenever a bucket reaches the results
    // (bucket_sort.rs:404-411): a level evaluated ahead on a copy of the universe would not see that
    if (cx->prm->distinct_values) return;
    struct Plan {
      uint64_t cost;
      PathList all;
      PathSlots sets;
    };
    Vec<Plan> plan;
    for (; it != end && (int)plan.size() < per_wait; ++it) {
      Plan pl;
      pl.cost = *it;
      Vec<int32_t> cur;
      IdSet visited, to_skip;
      size_t steps = 0;
      if (!enumerate(Graph::ROOT, pl.cost, visited, to_skip, cur, pl.all, steps)) break;
      if (pl.all.size() > MSI_BITS_REGION_PATHS) break;
      slots_of(pl.all, pl.sets);  // every condition resolved BEFORE the first level is enqueued
      plan.push_back(std::move(pl));
    }
    if (plan.size() < 2) return;
    Set ahead = cx->dev.clone(uni);
    Vec<Set> buckets;
    Dev::PendingLevels pending_levels;
    size_t n_enq = 0;
    for (; n_enq < plan.size(); ++n_enq) {
      buckets.push_back(cx->dev.zeros());
      if (plan[n_enq].all.empty()) continue;  // no path of this cost: an empty bucket, no launch
      if (!cx->dev.paths_enqueue(plan[n_enq].sets, buckets.back(), ahead, (uint32_t)n_enq, pending_levels)) break;
    }
    if (n_enq == 0) return;
    // the last rule: the ids of the first level that has paths ride in this list (Ctx::spec_k)
    std::shared_ptr<Vec<uint32_t>> ahead_ids;
    size_t ahead_level = n_enq;
    if (leaf && cx->spec_k) {
      for (size_t j = 0; j < n_enq && ahead_level == n_enq; ++j)
        if (!plan[j].all.empty()) ahead_level = j;
      if (ahead_level < n_enq) {
        auto box = msi_arena::make_shared<Vec<uint32_t>>();
        if (cx->dev.first_k_if_room(buckets[ahead_level], cx->spec_k,
                                    [box](const uint32_t *ids, size_t n) { box->assign(ids, ids + n); }))
          ahead_ids = box;
      }
    }
    const Vec<uint64_t> counts = cx->dev.paths_collect((uint32_t)n_enq, pending_levels);
    for (size_t j = 0; j < n_enq; ++j) {
      Ready r;
      r.cost = plan[j].cost;
      r.bucket = buckets[j];
      r.count = 0;
      if (j == ahead_level) r.ids = ahead_ids;
      for (size_t k = 0; k < plan[j].all.size(); ++k) {
        const uint64_t n = counts[j * MSI_BITS_REGION_PATHS + k];
        if (!n) continue;
        r.good.push_back(plan[j].all.path(k));
        ++g_stats.paths;
        r.count += n;
      }
      ready.push_back(std::move(r));
    }
  }

  // One launch for the whole cost level: every 16-byte chunk of documents walks the paths in order and a
  // path claims what the earlier paths left (msi_bits_paths_claim) — the same documents per path as the
  // path-by-path search, because claims never cross documents.
  bool fused_level(uint64_t cost) {
    // MSI_SEARCH_FUSED_LEVELS=0 forces the path-by-path search (the fallback for levels with > 256 paths), so
    // that the tests can hold both against the oracle
    if (cx->knob_fused_off) return false;
    PathList all;
    Vec<int32_t> cur;
    IdSet visited, to_skip;
    size_t steps = 0;
    if (!enumerate(Graph::ROOT, cost, visited, to_skip, cur, all, steps)) return false;
    if (all.empty()) return true;
    PathSlots sets;
    slots_of(all, sets);
    level_ids.reset();
    const Vec<uint64_t> counts = cx->dev.paths_claim(sets, bucket, uni, leaf ? cx->spec_k : 0, &level_ids);
    for (size_t k = 0; k < all.size(); ++k) {
      if (!counts[k]) continue;
      good.push_back(all.path(k));
      ++g_stats.paths;
      bucket_count += counts[k];
      uni_count -= counts[k];
    }
    return true;
  }

  // the condition sets of every path, as slots (the sets themselves stay alive in `cache` until end())
  void slots_of(const PathList &all, PathSlots &out) {
    out.off = all.off;
    out.slots.resize(all.conds.size());
    for (size_t i = 0; i < all.conds.size(); ++i) out.slots[i] = resolved(all.conds[i]).docs->slot;
  }
  const Resolved &resolved(int32_t ci) {
    EdgeSet *es = conds[ci].first;
    const uint32_t k = conds[ci].second;
    if (!es->resolved[k]) {
      // resolving may park this task (a full staging ring runs the list first) and another task of the bucket sort may
      // resolve the same condition meanwhile: the first result stays — references handed out are never invalidated
      Resolved r = resolve_condition(*cx, es->conds[k].second);
      if (!es->resolved[k]) es->resolved[k].reset(new Resolved(std::move(r)));
    }
    return *es->resolved[k];
  }

  // cheapest_paths.rs:147-310: edges in insertion order; a conditional edge cannot enter a node that
  // must be skipped, nor be taken once a node named by its skip list was traversed.  All the conditional
  // edges that leave a node are intersected with the path prefix in ONE launch (and one completion wait);
  // a sibling evaluated before an earlier sibling claimed documents is re-intersected when its turn comes.
  void visit(uint32_t node, uint64_t remaining, IdSet &visited, const IdSet &to_skip) {
    Vec<const Edge *> cand;
    for (const Edge &e : edges[node]) {
      if (remaining < e.cost) continue;
      const auto &dc = costs[e.dest];
      if (!std::binary_search(dc.begin(), dc.end(), remaining - e.cost)) continue;
      if (e.cond >= 0) {
        if (to_skip.count(e.dest)) continue;
        bool blocked = false;
        for (uint32_t s : e.skip) blocked |= visited.count(s) != 0;
        if (blocked) continue;
      }
      cand.push_back(&e);
    }
    Vec<Set> cs;
    for (const Edge *e : cand)
      if (e->cond >= 0) cs.push_back(resolved(e->cond).docs);
    // stack entries are subsets of the current universe, so only the first condition needs it
    Vec<std::pair<Set, uint64_t>> pre;
    if (!cs.empty() && !stop) pre = cx->dev.and_many(stack.empty() ? uni : stack.back().docs, cs);
    const uint64_t epoch0 = emit_epoch;
    size_t k = 0;
    for (const Edge *ep : cand) {
      if (stop) return;
      const Edge &e = *ep;
      const uint64_t rem = remaining - e.cost;
      if (e.cond < 0) {
        if (e.dest == Graph::END) {
          emit();
        } else {
          IdSet ts;
          if (!e.skip.empty()) { ts = to_skip; ts.insert(e.skip.begin(), e.skip.end()); }
          visit(e.dest, rem, visited, e.skip.empty() ? to_skip : ts);
        }
        continue;
      }
      Set d = pre[k].first;
      uint64_t cnt = pre[k].second;
      pre[k].first.reset();
      ++k;
      if (!cnt) continue;  // every extension of an empty prefix is empty (sets only shrink)
      if (emit_epoch != epoch0) d = cx->dev.and_new(cs[k - 1], stack.empty() ? uni : stack.back().docs, &cnt);
      if (!cnt) continue;
      stack.push_back({e.cond, d, false, cnt});
      visited.insert(e.dest);
      IdSet ts;
      if (!e.skip.empty()) { ts = to_skip; ts.insert(e.skip.begin(), e.skip.end()); }
      visit(e.dest, rem, visited, e.skip.empty() ? to_skip : ts);
      visited.erase(e.dest);
      stack.pop_back();
    }
  }

  void emit() {
    if (!uni_count) {
      stop = true;
      return;
    }
    uint64_t cnt;
    Vec<int32_t> path;
    for (auto &s : stack) path.push_back(s.cond);
    if (stack.empty()) {  // a path without any condition takes the whole universe
      cnt = uni_count;
      cx->dev.or_(bucket, uni);
      cx->dev.sub_(uni, uni);
    } else {
      StackE &top = stack.back();
      if (top.stale) {
        top.count = cx->dev.count(top.docs);
        top.stale = false;
      }
      cnt = top.count;
      if (!cnt) return;
      Vec<Set> ss;
      for (auto &s : stack) ss.push_back(s.docs);
      cx->dev.claim(top.docs, bucket, uni, ss);
      for (auto &s : stack) s.stale = true;
      top.stale = false;
      top.count = 0;
    }
    good.push_back(std::move(path));
    ++g_stats.paths;
    ++emit_epoch;
    bucket_count += cnt;
    uni_count -= cnt;
    if (!uni_count) stop = true;
  }

  void end() override {
    conds.clear();
    held.clear();
    edges.clear();
  }
};

// exact_attribute.rs:17-302.  The reference's `is_empty()` shortcuts only skip work: an empty candidate
// set yields empty ExactMatch / MatchesStart buckets, which bucket_sort drops.
struct ExactAttributeRule : Rule {
  Graph g;
  int state = 0;  // 0 empty, 1 exact attribute, 2 attribute starts
  Set exact_match, matches_start;  // this iteration's two buckets (already inside the universe)
  uint64_t exact_count = 0, start_count = 0;
  ExactAttributeRule() : Rule(R_EXACT_ATTRIBUTE, -1) {}
  Rule *fresh() const override { return new ExactAttributeRule(); }

  static uint32_t bucketed_position(uint32_t rel) {  // lib.rs:248-262
    if (rel < 16) return rel;
    if (rel < 24) return 24;
    uint32_t p = 1;
    while (p < rel) p <<= 1;
    return p;
  }

  void start(Ctx &c, const Set &universe, const Graph &graph) override {
    g = graph;
    state = 0;
    struct Info {
      uint32_t start_id;
      std::pair<int, uint32_t> exact;
      uint32_t start_pos, n_pos;
    };
    Vec<Info> infos;
    for (const GNode &n : g.nodes) {
      if (n.kind != 2) continue;
      auto e = c.exact_term(n.term.subset);
      if (e.first == 0) continue;
      infos.push_back({n.term.id_lo, e, n.term.pos_lo, n.term.pos_hi - n.term.pos_lo + 1});
    }
    std::stable_sort(infos.begin(), infos.end(), [](const Info &a, const Info &b) { return a.start_id < b.start_id; });
    Vec<Info> ded;
    for (auto &x : infos)
      if (ded.empty() || ded.back().start_id != x.start_id) ded.push_back(x);
    uint32_t count_all = 0;
    for (auto &x : ded) count_all += x.n_pos;
    if (ded.empty() || ded[0].start_id != 0) return;
    uint32_t prev = 0;
    for (auto &x : ded) {
      if (x.start_id < prev || x.start_id - prev > 1) return;
      prev = x.start_id;
    }
    Vec<std::pair<Phrase, uint32_t>> words_positions;
    std::string sig = std::to_string(count_all);
    for (auto &x : ded) {
      Phrase ws;
      if (x.exact.first == 1) ws = c.phrases[x.exact.second];
      else ws.push_back((int32_t)x.exact.second);
      for (int32_t w : ws) sig += "," + std::to_string(w);
      sig += "@" + std::to_string(x.start_pos);
      words_positions.push_back({ws, x.start_pos});
    }
    // Both buckets are universe-independent up to a final intersection:
    //   P = AND_(word, offset) word_position_docids(word, bucketed(start + offset))
    //   per searchable field: S = P AND_word word_fid_docids(word, fid);  W = field_id_word_count_docids(fid, #positions)
    //   ExactMatch = OR_fid (S & W),  MatchesStart = OR_fid (S - W)
    auto hit = c.exact_attr_cache.find(sig);
    if (hit == c.exact_attr_cache.end()) {
      c.relieve();
      if (!c.ix->field_id_word_count_docids)
        fail(MSI_E_INVALID, "the index vtable has no field_id_word_count_docids (exactness rule)");
      Set P;
      for (auto &wp : words_positions)
        for (size_t off = 0; off < wp.first.size(); ++off) {
          if (wp.first[off] < 0) continue;
          MsiCboBatch b;
          c.add_word_position(b, (uint32_t)wp.first[off], bucketed_position(wp.second + (uint32_t)off));
          if (!P) P = c.dev.decode(b);
          else c.dev.and_(P, c.dev.decode(b));
        }
      if (!P) P = c.dev.ones();
      // (no copy of P per field, no zeroed sets to unite into: the first field's sets are the unions so far — five set
      // operations and two clears less per search, over the whole index when the search could not move to a compact space)
      Set e1, e2;
      for (uint32_t i = 0; i < c.prm->n_searchable; ++i) {
        const uint32_t fid = c.prm->searchable_fids[i];
        Set S;
        for (auto &wp : words_positions)
          for (int32_t w : wp.first) {
            if (w < 0) continue;
            MsiCboBatch b;
            c.add_word_fid(b, (uint32_t)w, fid);
            const Set d = c.dev.decode(b);
            if (!S) S = c.dev.and_new(P, d, nullptr);
            else c.dev.and_(S, d);
          }
        if (!S) S = c.dev.clone(P);
        MsiCboBatch wc;
        if (count_all < 255 && !c.from_cache(&wc, 5, std::string(), std::string(), fid, count_all, nullptr, nullptr)) {
          const uint8_t *bytes = nullptr;
          size_t n = 0;
          Ctx::Cb cb_;
          const int32_t st = c.ix->field_id_word_count_docids(c.ix->user, fid, count_all, &bytes, &n);
          c.take(wc, st, bytes, n, "field_id_word_count_docids", 5, std::string(), std::string(), fid, count_all);
        }
        Set W = c.dev.decode(wc);
        Set both = c.dev.and_new(S, W, nullptr);
        if (!e1) e1 = both;
        else c.dev.or_(e1, both);
        c.dev.sub_(S, W);
        if (!e2) e2 = S;
        else c.dev.or_(e2, S);
      }
      if (!e1) e1 = c.dev.zeros();
      if (!e2) e2 = c.dev.zeros();
      c.dev.sub_(e2, e1);  // a document can match exactly in one field and only start another: ExactMatch wins
      hit = c.exact_attr_cache.emplace(sig, Vec<Set>{e1, e2, P}).first;
    }
    // both buckets against this universe in one launch and one wait (they are disjoint, so taking the first out
    // of the universe does not change the second)
    auto both = c.dev.and_many(universe, hit->second);
    // no document has the words at their positions: the reference answers with the single NoExactMatch bucket
    // (State::Empty, exact_attribute.rs:154-170) — same documents either way, one loop iteration instead of three
    if (!both[2].second) return;
    exact_match = both[0].first;
    matches_start = both[1].first;
    exact_count = both[0].second;
    start_count = both[1].second;
    state = 1;
  }

  bool next(Ctx &c, const Set &universe, uint64_t universe_count, Bucket &out) override {
    out.same_as = &g;
    if (state == 0) {
      out.docs = c.dev.clone(universe);
      out.count = universe_count;
      out.score = {MSI_SCORE_EXACT_ATTRIBUTE, 1, 3};
      return true;
    }
    out.docs = state == 1 ? exact_match : matches_start;
    out.count = state == 1 ? exact_count : start_count;
    if (c.prm->distinct_values && out.count) {
      // `candidates &= universe` at every call (exact_attribute.rs:253,275): without `distinct` the universe only
      // loses this rule's own, disjoint buckets between start() and here; with it, whatever a deeper rule excluded
      uint64_t n = 0;
      out.docs = c.dev.and_new(out.docs, universe, &n);
      out.count = n;
    }
    out.score = {MSI_SCORE_EXACT_ATTRIBUTE, state == 1 ? 3u : 2u, 3};
    state = state == 1 ? 2 : 0;
    return true;
  }
  void end() override {
    exact_match.reset();
    matches_start.reset();
    state = 0;
  }
};

// ScoreDetails::global_score of the details pushed so far (score_details.rs:123-154)
// sort.rs:95-233 over a per-document order key array (include/msi.h, msi_doc_keys): every bucket is the set of
// documents that share the smallest key left in the universe; the last one (key 0xFFFFFFFF) holds the documents
// without a value.  The rule does not look at the query: it hands the graph it was given to the next rule, and
// it also orders placeholder searches.
struct OrderByRule : Rule {
  uint32_t idx;
  const msi_doc_keys *keys;
  Graph g;
  OrderByRule(uint32_t i, const msi_doc_keys *k) : Rule(R_ORDER_BY, -1), idx(i), keys(k) {}
  Rule *fresh() const override { return new OrderByRule(idx, keys); }
  // With `distinct` the reference's rule keeps handing out the values its iteration STARTED with: the facet iterator was
  // built over the starting universe, and a value whose documents `distinct` removed meanwhile comes out as an empty
  // bucket (`bucket.candidates &= universe`, sort.rs:214-217) — one more turn of bucket_sort's loop, visible through the
  // deadline's check count.  `left` is that iterator: the starting universe, losing a value's documents per call.
  Set left;
  void start(Ctx &c, const Set &universe, const Graph &graph) override {
    g = graph;
    left.reset();
    if (c.prm->distinct_values) left = c.dev.clone(universe);
  }
  bool next(Ctx &c, const Set &universe, uint64_t universe_count, Bucket &out) override {
    if (!universe_count) return false;
    uint32_t key = 0;
    uint64_t n = 0;
    out.same_as = &g;
    if (left) {
      Set of_value = c.dev.order_next(keys, left, &key, &n);     // the next value of the starting universe ...
      out.docs = c.dev.and_new(of_value, universe, &n);           // ... and what is left of its documents
      out.universe_reduced = false;
    } else {
      out.docs = c.dev.order_next(keys, universe, &key, &n);
      out.universe_reduced = true;
    }
    out.count = n;
    out.score = {MSI_SCORE_SORT, idx, key};
    return true;
  }
  void end() override { left.reset(); }
};

// geo_sort.rs:14-160 over the _geo points in HBM (include/msi.h, msi_geo_points): every bucket is the set of documents
// within the error margin of the nearest (farthest) one left in the universe; when no document of the universe has a
// point the rule answers with the whole universe and no value (geo_sort.rs:149-153).  Like Sort it does not look at
// the query and also orders placeholder searches.
struct GeoSortRule : Rule {
  uint32_t idx;
  msi_geo_rule rule;
  Graph g;
  // documents/geo_sort.rs keeps a cache of candidates in visiting order, filled either from the R-tree (the nearest
  // `cache_size`, in chord-distance order = exact distance order) or ITERATIVELY (every candidate, sorted by its distance
  // truncated to whole metres — docid order inside a metre), and refilled when it runs dry.  Which of the two a fill uses
  // is decided by the strategy and the number of candidates at that moment (:81-94).  Here: `mode` of the current fill
  // and how many of its documents are still to come.
  int mode = 0;              // 0: no fill yet, 1: iterative, 2: R-tree order (the min / take kernels)
  uint64_t cache_left = 0;
  GeoSortRule(uint32_t i, const msi_geo_rule &r) : Rule(R_ORDER_BY, -1), idx(i), rule(r) {}
  Rule *fresh() const override { return new GeoSortRule(idx, rule); }
  void start(Ctx &, const Set &, const Graph &graph) override {
    g = graph;
    mode = 0;
    cache_left = 0;
  }
  bool next(Ctx &c, const Set &universe, uint64_t universe_count, Bucket &out) override {
    if (!universe_count) return false;
    const uint32_t cache_size = c.prm->geo_cache_size ? c.prm->geo_cache_size : 1000;
    const uint32_t cap = c.prm->geo_max_bucket_size ? c.prm->geo_max_bucket_size : 1000;
    const double margin = c.prm->geo_distance_error_margin;
    const int strategy = c.prm->geo_strategy;
    out.same_as = &g;
    Vec<uint32_t> ids;
    Vec<double> dist;
    uint64_t total = 0;
    bool listed = false;
    if (strategy != MSI_GEO_ALWAYS_RTREE && (mode == 0 || cache_left == 0 || mode == 1)) {
      // a fill (or the iterative cache seen through the current universe): the candidates, at most cache_size of them
      c.dev.geo_list(rule, universe, strategy == MSI_GEO_ALWAYS_ITERATIVE ? 0xFFFFFu : cache_size, ids, dist, &total);
      listed = true;
    }
    if (mode == 0 || cache_left == 0) {   // fill_cache :66-133
      const bool use_rtree = strategy == MSI_GEO_ALWAYS_RTREE || (strategy == MSI_GEO_DYNAMIC && total >= cache_size);
      mode = use_rtree ? 2 : 1;
      cache_left = use_rtree ? (listed ? std::min<uint64_t>(total, cache_size) : cache_size) : total;
    }
    // (AlwaysIterative is a test / debugging strategy of the reference; a candidate list past the device list's cap would
    // silently fall through to the exact-distance order below, which orders documents inside one metre differently)
    if (strategy == MSI_GEO_ALWAYS_ITERATIVE && listed && total > ids.size())
      fail(MSI_E_UNSUPPORTED, "GeoSort with the AlwaysIterative strategy lists at most 1 048 575 candidates");
    if (mode == 1 && listed && total <= ids.size()) {
      if (!total) {   // no candidate left: what remains has no point (:161-163)
        out.score = {MSI_SCORE_GEO_SORT, idx, 0xFFFFFFFFu};
        out.docs = c.dev.clone(universe);
        out.count = universe_count;
        return true;
      }
      // sort_by_cached_key(distance as usize) over candidates in docid order: stable (:127-131); a descending rule pops
      // the cache from the back
      Vec<uint32_t> order(ids.size());
      for (uint32_t i = 0; i < order.size(); ++i) order[i] = i;
      std::sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) {
        const uint64_t ka = (uint64_t)dist[a], kb = (uint64_t)dist[b];
        return ka != kb ? ka < kb : ids[a] < ids[b];
      });
      if (!rule.ascending) std::reverse(order.begin(), order.end());
      Vec<uint32_t> bucket;
      const double d0 = dist[order[0]];
      for (uint32_t o : order) {   // next_bucket :166-206
        if (fabs(d0 - dist[o]) > margin) break;
        bucket.push_back(ids[o]);
        if (bucket.size() == cap) break;
      }
      const uint32_t first = ids[order[0]];
      std::sort(bucket.begin(), bucket.end());
      out.docs = c.dev.from_docids(bucket);
      out.count = bucket.size();
      c.dev.sub_(universe, out.docs);
      out.universe_reduced = true;
      out.score = {MSI_SCORE_GEO_SORT, idx, first};
      cache_left -= std::min<uint64_t>(cache_left, bucket.size());
      return true;
    }
    // R-tree order: exact distance order on the device (min, then take within the margin)
    uint32_t first = 0xFFFFFFFFu;
    uint64_t n = 0;
    Set b = c.dev.geo_next(rule, c.prm->geo_max_bucket_size, margin, universe, &first, &n);
    out.score = {MSI_SCORE_GEO_SORT, idx, first};
    if (first == 0xFFFFFFFFu) {
      out.docs = c.dev.clone(universe);
      out.count = universe_count;
      return true;
    }
    out.docs = b;
    out.count = n;
    out.universe_reduced = true;
    cache_left -= std::min<uint64_t>(cache_left, n);
    return true;
  }
  void end() override {}
};

double global_score(const Vec<Score> &scores) {
  Vec<msi_score_detail> d;
  for (const Score &s : scores) d.push_back(msi_score_detail{s.kind, s.a, s.b});
  return msi_score_details_global_score(d.data(), (uint32_t)d.size());
}

// get_ranking_rules_for_query_graph_search, mod.rs:510-649
Vec<std::unique_ptr<Rule>> ranking_rules(const msi_search_params *p) {
  Vec<std::unique_ptr<Rule>> rules;
  bool words = p->strategy == MSI_TERMS_ALL, typo = false, prox = false, attribute = false, attr_rank = false,
       word_pos = false, exact = false;
  uint32_t n_order = 0, n_geo = 0;
  auto add_words = [&]() {
    if (!words) {
      rules.emplace_back(new GraphRule(R_WORDS, p->strategy));
      words = true;
    }
  };
  for (uint32_t i = 0; i < p->n_criteria; ++i) {
    const int32_t c = p->criteria[i];
    if (c == MSI_CRIT_TYPO || c == MSI_CRIT_ATTRIBUTE || c == MSI_CRIT_ATTRIBUTE_RANK || c == MSI_CRIT_WORD_POSITION ||
        c == MSI_CRIT_PROXIMITY || c == MSI_CRIT_EXACTNESS)
      add_words();
    switch (c) {
      case MSI_CRIT_WORDS: add_words(); break;
      case MSI_CRIT_TYPO:
        if (!typo) rules.emplace_back(new GraphRule(R_TYPO, -1));
        typo = true;
        break;
      case MSI_CRIT_PROXIMITY:
        if (!prox) rules.emplace_back(new GraphRule(R_PROXIMITY, -1));
        prox = true;
        break;
      case MSI_CRIT_ATTRIBUTE:
        if (attribute || attr_rank || word_pos) break;
        attribute = true;
        rules.emplace_back(new GraphRule(R_FID, -1));
        rules.emplace_back(new GraphRule(R_POSITION, -1));
        break;
      case MSI_CRIT_ATTRIBUTE_RANK:
        if (attribute || attr_rank) break;
        attr_rank = true;
        rules.emplace_back(new GraphRule(R_FID, -1));
        break;
      case MSI_CRIT_WORD_POSITION:
        if (attribute || word_pos) break;
        word_pos = true;
        rules.emplace_back(new GraphRule(R_POSITION, -1));
        break;
      case MSI_CRIT_EXACTNESS:
        if (exact) break;
        exact = true;
        rules.emplace_back(new ExactAttributeRule());
        rules.emplace_back(new GraphRule(R_EXACTNESS, -1));
        break;
      case MSI_CRIT_ORDER_BY:
        if (n_order < p->n_order_keys && p->order_keys && p->order_keys[n_order])
          rules.emplace_back(new OrderByRule(n_order, p->order_keys[n_order]));
        ++n_order;
        break;
      case MSI_CRIT_GEO_SORT:
        if (n_geo < p->n_geo_rules && p->geo_rules && p->geo_rules[n_geo].points)
          rules.emplace_back(new GeoSortRule(n_geo, p->geo_rules[n_geo]));
        ++n_geo;
        break;
      default: break;  // MSI_CRIT_SORT: expanded by the caller into MSI_CRIT_ORDER_BY / MSI_CRIT_GEO_SORT entries
    }
  }
  return rules;
}

// get_ranking_rules_for_placeholder_search, mod.rs:352-420: only the Sort / Asc / Desc rules
Vec<std::unique_ptr<Rule>> placeholder_rules(const msi_search_params *p) {
  Vec<std::unique_ptr<Rule>> rules;
  uint32_t n_order = 0, n_geo = 0;
  for (uint32_t i = 0; i < p->n_criteria; ++i)
    if (p->criteria[i] == MSI_CRIT_ORDER_BY) {
      if (n_order < p->n_order_keys && p->order_keys && p->order_keys[n_order])
        rules.emplace_back(new OrderByRule(n_order, p->order_keys[n_order]));
      ++n_order;
    } else if (p->criteria[i] == MSI_CRIT_GEO_SORT) {
      if (n_geo < p->n_geo_rules && p->geo_rules && p->geo_rules[n_geo].points)
        rules.emplace_back(new GeoSortRule(n_geo, p->geo_rules[n_geo]));
      ++n_geo;
    }
  return rules;
}

// make_ngram, parse_query.rs:227-300 -> term index or -1
int32_t make_ngram(Ctx &c, const Vec<std::pair<uint32_t, std::pair<uint32_t, uint32_t>>> &ts, size_t lo, size_t hi) {
  for (size_t i = lo; i <= hi; ++i)
    if (c.terms[ts[i].first].phrase >= 0) return -1;
  for (size_t i = lo; i < hi; ++i)
    if (ts[i].second.second + 1 != ts[i + 1].second.first) return -1;
  std::string s;
  Vec<uint32_t> ws;
  for (size_t i = lo; i <= hi; ++i) {
    const Term &t = c.terms[ts[i].first];
    if (t.is_ngram) return -1;
    ws.push_back(t.original);
    s += c.words[t.original];
  }
  if (s.size() > MAX_WORD_LENGTH) return -1;
  const bool is_prefix = c.terms[ts[hi].first].is_prefix;
  const uint32_t b = c.budget(s), n1 = (uint32_t)(hi - lo);
  const uint32_t max_typos = b > n1 ? b - n1 : 0;
  Term t = c.term_from_word(s, max_typos, is_prefix, true);
  for (Phrase &syn : c.synonyms_of(ws)) t.synonyms.push_back(c.phrase(syn));  // parse_query.rs:277-285
  t.is_ngram = true;
  t.ngram_words = ws;
  t.is_prefix = is_prefix;
  t.max_lev = max_typos;
  c.terms.push_back(t);
  return (int32_t)c.terms.size() - 1;
}

void search(Ctx &c, const msi_located_term *lt, uint32_t n_terms, const uint8_t *universe_cbo, size_t universe_len,
            uint32_t *out_docids, msi_score_detail *out_scores, uint32_t *out_n_scores, uint32_t *out_n,
            uint64_t *out_candidates, int32_t *out_degraded) {
  const Clock started;
  uint32_t deadline_checks = 0;
  auto deadline_exceeded = [&]() {  // Deadline::exceeded, lib.rs:211-226
    const msi_search_params *q = c.prm;
    if (q->stop_after >= 0) return deadline_checks++ >= (uint32_t)q->stop_after;
    return q->time_budget_us != 0 && started.ms() * 1e3 > (double)q->time_budget_us;
  };
  const msi_search_params *p = c.prm;
  // ---- located terms -> terms -----------------------------------------------------------------
  Vec<std::pair<uint32_t, std::pair<uint32_t, uint32_t>>> ts;  // (term, positions)
  Vec<uint32_t> negatives;
  for (uint32_t i = 0; i < n_terms; ++i) {
    const msi_located_term &l = lt[i];
    if (!l.words || l.n_words == 0) fail(MSI_E_INVALID, "a located term has no word");
    if (l.is_phrase & MSI_TERM_NEGATIVE) {
      negatives.push_back(i);
      continue;
    }
    if (l.is_phrase & MSI_TERM_PHRASE) {
      Phrase ph;
      std::string desc;
      for (uint32_t k = 0; k < l.n_words; ++k) {
        if (!l.words[k].len) {
          ph.push_back(-1);
          continue;
        }
        const std::string w((const char *)l.words[k].word, l.words[k].len);
        ph.push_back((int32_t)c.word(w));
        if (!desc.empty()) desc += " ";
        desc += w;
      }
      Term t;
      t.original = c.word(desc);
      t.phrase = (int32_t)c.phrase(ph);
      c.terms.push_back(t);
    } else {
      const std::string w((const char *)l.words[0].word, l.words[0].len);
      c.terms.push_back(c.term_from_word(w, c.budget(w), l.words[0].is_prefix != 0));
    }
    ts.push_back({(uint32_t)c.terms.size() - 1, {l.position_start, l.position_end}});
  }
  // ---- QueryGraph::from_query -----------------------------------------------------------------
  Graph g;
  g.nodes.resize(2);
  g.nodes[0].kind = 0;
  g.nodes[1].kind = 1;
  auto add_node = [&](uint32_t ti, uint32_t plo, uint32_t phi, uint32_t ilo, uint32_t ihi) {
    GNode n;
    n.kind = 2;
    n.term.subset = c.full(ti);
    n.term.pos_lo = plo;
    n.term.pos_hi = phi;
    n.term.id_lo = ilo;
    n.term.id_hi = ihi;
    g.nodes.push_back(n);
  };
  for (size_t i = 0; i < ts.size(); ++i) {
    add_node(ts[i].first, ts[i].second.first, ts[i].second.second, (uint32_t)i, (uint32_t)i);
    for (size_t n = 2; n <= 3; ++n) {
      if (i + 1 < n) continue;
      const size_t lo = i + 1 - n;
      const int32_t ng = make_ngram(c, ts, lo, i);
      if (ng >= 0) add_node((uint32_t)ng, ts[lo].second.first, ts[i].second.second, (uint32_t)lo, (uint32_t)i);
    }
  }
  build_initial_edges(g);
#ifndef MSI_SEARCH_DIRECT_ONLY
  {
    const bool cpu_prof = msi_cpu_prof_on();
    const uint64_t t0 = cpu_prof ? msi_thread_cpu_ns() : 0;
    c.compute_derivations();
    if (cpu_prof) msi_cpu_prof_add(4, msi_thread_cpu_ns() - t0);
  }
#else
  c.compute_derivations();
#endif

  // ---- universe (resolve_universe, mod.rs:273-301) ---------------------------------------------
  // (null while it is "every document": a search without a filter and without negative terms never materialises that set —
  // its universe is what its query graph matches, query_graph_docids)
  Set universe;
  uint64_t universe_floor = 0;   // documents the universe holds at least, as far as the host knows before the device has counted
  if (universe_cbo) {
    MsiCboBatch ub;
    if (!msi_cbo_batch_append(ub, universe_cbo, universe_len)) fail(MSI_E_INVALID, "malformed universe bitmap");
    universe = c.dev.decode(ub);
  } else if (!negatives.empty()) {
    universe = c.dev.ones();
  }
  // resolve_negative_words / resolve_negative_phrases (mod.rs:323-351), applied by Search::execute (mod.rs:431-440)
  for (uint32_t i : negatives) {
    const msi_located_term &l = lt[i];
    if (l.is_phrase & MSI_TERM_PHRASE) {
      Phrase ph;
      for (uint32_t k = 0; k < l.n_words; ++k)
        ph.push_back(l.words[k].len ? (int32_t)c.word(std::string((const char *)l.words[k].word, l.words[k].len)) : -1);
      c.dev.sub_(universe, c.phrase_docids(c.phrase(ph)));
    } else {
      c.dev.sub_(universe, c.word_full(c.word(std::string((const char *)l.words[0].word, l.words[0].len)), true));
    }
  }
  // only stop words (or nothing) survived the tokenizer: a placeholder search — no keyword rule applies
  // (mod.rs:770-800): the Sort / Asc / Desc rules alone order the universe, ascending docids when there is none
  const bool placeholder = ts.empty();
  if (!placeholder) {
    Graph reduced = g;
    if (p->strategy == MSI_TERMS_LAST || p->strategy == MSI_TERMS_FREQUENCY) {
      Vec<uint32_t> rm;
      for (auto &ns : removal_order_of(c, g, p->strategy)) rm.insert(rm.end(), ns.begin(), ns.end());
      remove_nodes_keep_edges(reduced, rm);
    }
    uint64_t d_floor = 0;
    Set d = query_graph_docids(c, reduced, universe, &d_floor);
    if (!universe) {
      universe = d;               // (read-only from here on: it may be a shared set)
      universe_floor = d_floor;
    } else if (d) {
      c.dev.and_(universe, d);
    }
  }
  if (!universe) universe = c.dev.ones();
  auto rules = placeholder ? placeholder_rules(p) : ranking_rules(p);
#ifndef MSI_SEARCH_DIRECT_ONLY
  // Universe compaction (Dev::compact_begin): everything below works on subsets of `universe`.  Not with the rules and
  // options that read per-document arrays by docid through direct kernels (Sort / GeoSort keys, distinct values).
  // (nor with a page that reaches past what one list can read back: Dev::first_k then falls back to the direct kernel,
  // which knows nothing of ranks — ADVICE r3: offset 8200 + limit 5 over a compact universe)
  bool may_compact = c.dev.vm && Dev::compact_mode() > 0 && !placeholder && !p->distinct_values && msi_bits_n_slots(c.dev.pool.p) <= 1024 &&
                     (uint64_t)p->from + p->length <= MSI_VM_MAX_FIRSTK;
  for (auto &r : rules) may_compact = may_compact && r->kind != R_ORDER_BY;
  // (the tables ride in the list that counts the universe: no round of its own.  Not for a universe that is known to be too
  // large for the compact space before it is counted — a search for a frequent word: two more sweeps of the index, the
  // prefix table and 5 MB of rank -> docid entries written for nothing, and a second phase in the search's first list)
  const bool tables_now = may_compact && universe_floor <= msi_bits_compact_capacity(c.dev.pool.p);
  if (tables_now) c.dev.rank_tables(universe);
#endif
  uint64_t universe_count = c.dev.count(universe);
#ifndef MSI_SEARCH_DIRECT_ONLY
  if (tables_now && Dev::late_mode() < 2 && c.dev.compact_pays(universe_count)) {
    c.dev.compact_begin(universe, universe_count);
    c.to_compact_space();
    universe = c.dev.ones();     // U0 in its own space: every rank
  }
#endif

  // ---- bucket_sort (bucket_sort.rs:23-343; no distinct, pins, deadline, score threshold) ---------
  const uint32_t from = p->from, length = p->length;
  const bool detailed = p->detailed_scores != 0;
  uint32_t n_out = 0;
  *out_n = 0;
  if (out_candidates) *out_candidates = universe_count;
  const msi_doc_values *dv = p->distinct_values;
  // search/new/mod.rs:894-907: an exhaustive count under `distinct` is what the distinct rule keeps of all_candidates
  auto exhaustive_count = [&](const Set &all) {
    uint64_t kept = 0;
    c.dev.distinct(dv, all, &kept);
    return kept;
  };
  const bool exhaustive_distinct = p->exhaustive_number_hits && dv && out_candidates;
  if (universe_count < from || length == 0) {
    if (exhaustive_distinct && universe_count) *out_candidates = exhaustive_count(universe);
    return;
  }
  if (rules.empty() && dv) {
    // bucket_sort.rs:61-92: the first from + length documents the distinct loop keeps, in docid order; only THEIR
    // values exclude documents from all_candidates
    uint64_t n_kept = 0;
    auto de = c.dev.distinct(dv, universe, &n_kept);
    Set kept = de.first;
    auto ids = c.dev.first_k(kept, (uint32_t)std::min<uint64_t>(n_kept, (uint64_t)from + length));
    Set exc = de.second;
    if (n_kept > (uint64_t)from + length) {
      kept = c.dev.from_docids(ids);
      exc = c.dev.distinct_excluded(dv, kept);
    }
    if (out_candidates) {
      Set all = c.dev.clone(universe);
      c.dev.sub_(all, exc);
      c.dev.or_(all, kept);
      *out_candidates = exhaustive_distinct ? exhaustive_count(all) : c.dev.count(all);
    }
    for (size_t i = from; i < ids.size(); ++i) {
      out_docids[n_out] = ids[i];
      out_n_scores[n_out++] = 0;
    }
    *out_n = n_out;
    return;
  }
  if (rules.empty()) {
    auto ids = c.dev.first_k(universe, from + length);
    for (size_t i = from; i < ids.size(); ++i) {
      out_docids[n_out] = ids[i];
      out_n_scores[n_out++] = 0;
    }
    *out_n = n_out;
    return;
  }
  const size_t nr = rules.size();
  // ---- the bucket sort as a tree of tasks -------------------------------------------------------------------------
  // Without `distinct`, a score threshold or the stop_after test hook, where a bucket's documents land in the result
  // list is known as soon as the bucket exists (the cardinalities before it), so every bucket's sub-tree — the rules
  // below it, on its documents — is independent of its siblings': bucket_sort.rs:187-343 as a recursion, the sub-trees
  // as cooperative tasks (Tasks) that share the search's command list.  Same buckets, same order, same score details
  // as the loop below; the loop stays for the sequential cases (distinct, thresholds, GeoSort's direct kernels).
  {
    const char *knob = getenv("MSI_SEARCH_TASKS");
    const int max_tasks = knob ? atoi(knob) : 24;
    // Sort / GeoSort rules stay sequential: GeoSort works through direct kernels on the pool's stream, and the Sort
    // rule's two-phase command pair shares one accumulator cell per list
    bool direct_rules = false;
    for (auto &r : rules) direct_rules = direct_rules || r->kind == R_ORDER_BY;
    const bool tree = max_tasks > 0 && !dv && !p->has_score_threshold && p->stop_after < 0 && !direct_rules &&
                      !getenv("MSI_SEARCH_TRACE");
    if (tree) {
      const uint64_t page_end = (uint64_t)from + length;
      {
        const char *knob = getenv("MSI_SEARCH_IDS_AHEAD");   // 0: off (tests hold both against the oracle)
        c.spec_k = (page_end <= 64 && !(knob && knob[0] == '0')) ? (uint32_t)page_end : 0;
      }
      bool ids_short = false, degraded = false;
      const bool tree_trace = getenv("MSI_SEARCH_TREE_TRACE") != nullptr;
      // A rule may end before its buckets covered its universe (the reference drops what is left, bucket_sort.rs `back!`,
      // and the documents after it move up): then the places handed out below — cumulative cardinalities — are wrong.
      // Rare (three of 230 000 random searches); the tree notices and the search is done again by the sequential loop.
      bool dropped = false;
      // documents [off, off + count) of the final order, all with the same score details
      auto emit = [&](const Set &docs, uint64_t count, uint64_t off, const Vec<Score> &scores,
                      const std::shared_ptr<Vec<uint32_t>> &known = nullptr) {
        if (!count || off >= page_end || off + count <= from) return;
        const uint64_t skip = off < from ? from - off : 0;
        const uint32_t take = (uint32_t)std::min<uint64_t>(count - skip, page_end - (off + skip));
        const uint32_t at = (uint32_t)(off + skip - from);
        const uint32_t ns = (uint32_t)std::min<size_t>(scores.size(), MSI_MAX_SCORE_DETAILS);
        for (uint32_t i = 0; i < take; ++i) {
          for (uint32_t sdx = 0; sdx < ns; ++sdx)
            out_scores[(size_t)(at + i) * MSI_MAX_SCORE_DETAILS + sdx] = msi_score_detail{scores[sdx].kind, scores[sdx].a, scores[sdx].b};
          out_n_scores[at + i] = ns;
        }
        if (fk_trace())
          fprintf(stderr, "[msi emit] place %llu count %llu -> at %u skip %llu take %u slot %u known %d\n", (unsigned long long)off,
                  (unsigned long long)count, at, (unsigned long long)skip, take, docs->slot, known ? (int)known->size() : -1);
        if (known && known->size() >= skip + take) {   // the rule asked for these ids in the list that counted the bucket
          for (uint32_t i = 0; i < take; ++i) out_docids[at + i] = (*known)[(size_t)skip + i];
          return;
        }
        c.dev.first_k_later(docs, (uint32_t)(skip + take), [out_docids, at, skip, take, &ids_short](const uint32_t *ids, size_t n) {
          if (n < skip + take) ids_short = true;   // cannot happen: `count` came from the device
          for (size_t i = (size_t)skip; i < n && i < skip + take; ++i) out_docids[at + (i - skip)] = ids[i];
        });
      };
#ifndef MSI_SEARCH_DIRECT_ONLY
      Tasks tasks;
      const bool no_gate = getenv("MSI_SEARCH_TASKS_NO_GATE") != nullptr;   // tests: provoke the out-of-slots re-run
      // free slots a task is admitted against (MSI_SEARCH_TASK_SLOTS; a search that runs out of slots is done again, one bucket at a time)
      static const size_t task_slots = getenv("MSI_SEARCH_TASK_SLOTS") ? (size_t)std::max(1, atoi(getenv("MSI_SEARCH_TASK_SLOTS"))) : 48;
      bool coop = c.dev.vm && max_tasks > 1;
      bool late_ok = may_compact && Dev::late_mode() > 0;
      static const bool late_wait = getenv("MSI_SEARCH_LATE_WAIT") && getenv("MSI_SEARCH_LATE_WAIT")[0] == '1';
      bool late_waiting = false;
#else
      const bool coop = false;
#endif
      // rule `cur` ranks `uni` (count documents, first of them at place `off`), bucket_sort.rs:187-343
      std::function<void(size_t, Set, uint64_t, uint64_t, Vec<Score>, const Graph &)> rank;
      rank = [&](size_t cur, Set uni, uint64_t left, uint64_t off, Vec<Score> scores, const Graph &graph) {
        std::unique_ptr<Rule> rule_owner(rules[cur]->fresh());
        Rule *rule = rule_owner.get();
        rule->leaf = cur == nr - 1;
        rule->start(c, uni, graph);
        for (;;) {
          if (left == 0 || off >= page_end) break;                       // the page is full: the loop of :187 ends
          if (!detailed && left == 1) {                                  // :195-203
            emit(uni, left, off, scores);
            break;
          }
          if (deadline_exceeded()) {                                     // :206-264: what is left goes out unranked
            scores.push_back(Score{MSI_SCORE_SKIPPED, 0, 1});
            emit(uni, left, off, scores);
            degraded = true;
            break;
          }
          Bucket b;
          rule->page_room = page_end - off;
          if (!rule->next(c, uni, left, b)) {                            // (a rule normally ends with its universe empty)
            if (left) dropped = true;
            break;
          }
          ++g_stats.buckets;
          if (tree_trace)   // MSI_SEARCH_TREE_TRACE: one line per bucket — after which completion wait it became known
            fprintf(stderr, "[msi tree] wait %llu rule %zu (kind %d) bucket of %llu at place %llu, %llu left, score (%u,%u,%u)%s\n",
                    (unsigned long long)g_stats.syncs, cur, rule->kind, (unsigned long long)b.count, (unsigned long long)off,
                    (unsigned long long)(left - b.count), b.score.kind, b.score.a, b.score.b,
                    b.ids ? ", ids came along" : "");
          // what is left of `uni` is only read by this loop's next round — and there is none when the bucket took the rest
          // of it or fills the page (most rule evaluations of a search end that way: six set operations over the whole
          // index in a search for a frequent word)
          if (!b.universe_reduced && left != b.count && off + b.count < page_end) c.dev.sub_(uni, b.docs);
          left -= b.count;
          Vec<Score> sc = scores;
          sc.push_back(b.score);
          if (cur == nr - 1 || (!detailed && b.count <= 1) || off + b.count < from) {
            emit(b.docs, b.count, off, sc, b.ids);                       // :296-330
          } else if (off < page_end && b.count) {
#ifndef MSI_SEARCH_DIRECT_ONLY
            // The rules below rank this bucket in ITS compact space (Ctx::late_enter) when the search could not compact
            // its whole universe, the bucket is at most an eighth of the index and this is the only task alive — the
            // sub-tree's own tasks are joined before the search returns to the caller's pool.
            // MSI_SEARCH_LATE_WAIT=1 (not measured yet: off): with other tasks alive, ONE task at a time may wait for
            // them to finish and then move its bucket's sub-tree — the others neither wait nor enter meanwhile, so the wait ends
            const bool can_late = late_ok && !c.dev.compact() && c.dev.late_pays(b.count);
            if (can_late && coop && late_wait && !late_waiting && tasks.live > 1) {
              late_waiting = true;
              while (tasks.live > 1) tasks.park();
              late_waiting = false;
            }
            if (can_late && !c.dev.compact() && (!coop || tasks.live == 1)) {
              struct Leave {
                Ctx &c;
                bool done = false;
                ~Leave() { if (!done) c.late_leave(true); }
              } leave{c};
              c.late_enter(b.docs, b.count);
              rank(cur + 1, c.dev.ones(), b.count, off, sc, b.graph());
              if (coop)
                while (tasks.live > 1) tasks.park();   // (a yield: the scheduler runs the others' list and comes back)
              c.late_leave(false);
              leave.done = true;
            } else
            // a task keeps its own working sets alive: only as many tasks as the pool has room for (a rule evaluation
            // is given what relieve() keeps free)
            if (coop && tasks.live < (size_t)max_tasks &&
                (no_gate || c.dev.cur->free_.size() + c.dev.cur->clean_.size() + c.dev.cur->held.size() >= task_slots * (tasks.live + 2))) {
              // (the graph is moved into the task: the Bucket dies at the end of this iteration)
              auto gp = msi_arena::make_shared<Graph>(b.graph());
              tasks.spawn([&rank, cur, docs = b.docs, cnt = b.count, off, sc, gp]() { rank(cur + 1, docs, cnt, off, sc, *gp); });
            } else
#endif
              rank(cur + 1, b.docs, b.count, off, sc, b.graph());
          }
          off += b.count;
        }
        rule->end();
      };
      Set root = c.dev.clone(universe);
#ifndef MSI_SEARCH_DIRECT_ONLY
      if (coop) {
        std::exception_ptr err;
        c.dev.tasks = &tasks;
        tasks.spawn([&]() { rank(0, root, universe_count, 0, {}, g); });
        for (;;) {
          const bool parked = tasks.round(err);
          if (!parked) break;
          if (!tasks.abort) {
            try {
              c.dev.run_now();
            } catch (...) {
              if (!err) err = std::current_exception();
              tasks.abort = true;
            }
          }
          tasks.release();
        }
        c.dev.tasks = nullptr;
        if (err) {
          bool oom = false;
          try {
            std::rethrow_exception(err);
          } catch (const Fail &f) {
            oom = f.code == MSI_E_OOM;
          } catch (...) {
          }
          if (!oom) std::rethrow_exception(err);
          // the tasks together held more sets than the pool has slots: once more, one bucket at a time.  What was
          // recorded runs first — the list holds the zeroing of slots the pool already counts as clean and the decodes
          // of sets the caches already hold; dropping it would leave them undefined.
          if (getenv("MSI_SEARCH_DEBUG")) fprintf(stderr, "[msi] bucket-sort tasks ran out of pool slots: sequential re-run\n");
          c.dev.run_now();
          tasks.all.clear();
          // start over with nothing cached: the tasks were cut off wherever they stood, and only what they were about to do
          // knew which of the sets they had filed were complete
          c.forget();
          coop = false;
          late_ok = false;
          ids_short = false;
          dropped = false;
          rank(0, c.dev.clone(universe), universe_count, 0, {}, g);
        }
      } else
#endif
        rank(0, root, universe_count, 0, {}, g);
      c.dev.finish_list();   // the ids of the last buckets, when any are still to come
      if (!dropped) {
        if (ids_short) fail(MSI_E_INTERNAL, "a bucket held fewer documents than its cardinality said");
        *out_n = (uint32_t)std::min<uint64_t>(length, universe_count - from);
        if (degraded && out_degraded) *out_degraded = 1;
        return;
      }
      if (getenv("MSI_SEARCH_DEBUG")) fprintf(stderr, "[msi] a rule dropped documents: sequential re-run\n");
    }
  }
  Vec<Set> unis(nr);
  Vec<uint64_t> uni_counts(nr, 0);
  Vec<Score> scores;
  unis[0] = c.dev.clone(universe);
  uni_counts[0] = universe_count;
  rules[0]->start(c, universe, g);
  size_t cur = 0;
  uint64_t cur_off = 0;
  Set excluded;  // documents a ranking score threshold removed from all_candidates
  bool ids_short = false;
  auto add = [&](Set cands, uint64_t count) {  // maybe_add_to_results :382-460
    if (dv && count) {
      // apply_distinct_rule, then `universe -= excluded` for every rule of the stack and all_candidates (:404-411)
      const Set given = cands;
      uint64_t n_kept = 0;
      auto de = c.dev.distinct(dv, cands, &n_kept);
      Vec<Set> live;
      Vec<size_t> idx;
      for (size_t i = 0; i < nr; ++i)
        if (unis[i] && unis[i] != given && uni_counts[i]) {  // a universe handed over as the bucket is dropped by the caller
          live.push_back(unis[i]);
          idx.push_back(i);
        }
      if (!live.empty()) {
        const auto counts = c.dev.sub_many(de.second, live);
        for (size_t k = 0; k < idx.size(); ++k) uni_counts[idx[k]] = counts[k];
      }
      c.dev.and_(de.second, universe);  // all_candidates lives inside the initial universe
      if (!excluded) excluded = c.dev.zeros();
      c.dev.or_(excluded, de.second);
      cands = de.first;
      count = n_kept;
    }
    if (!count) return;
    if (excluded) c.dev.sub_(excluded, cands);  // `*all_candidates |= &candidates`
    uint64_t skip = 0;
    if (cur_off < from) {
      if (cur_off + count < from) {
        cur_off += count;
        return;
      }
      skip = from - cur_off;
    }
    const uint32_t take = (uint32_t)std::min<uint64_t>(count - skip, length - n_out);
    if (take) {
      // count is the bucket's exact cardinality: the ids of out[n_out .. n_out + take) are known to exist, their
      // score details are known now; the ids themselves arrive with the next list that runs (finish() at the latest)
      const uint32_t at = n_out;
      for (uint32_t i = 0; i < take; ++i) {
        const uint32_t ns = (uint32_t)std::min<size_t>(scores.size(), MSI_MAX_SCORE_DETAILS);
        for (uint32_t s = 0; s < ns; ++s)
          out_scores[(size_t)n_out * MSI_MAX_SCORE_DETAILS + s] = msi_score_detail{scores[s].kind, scores[s].a, scores[s].b};
        out_n_scores[n_out] = ns;
        ++n_out;
      }
      c.dev.first_k_later(cands, (uint32_t)(skip + take), [out_docids, at, skip, take, &ids_short](const uint32_t *ids, size_t n) {
        if (n < skip + take) ids_short = true;   // cannot happen: `count` came from the device
        for (size_t i = (size_t)skip; i < n && i < skip + take; ++i) out_docids[at + (i - skip)] = ids[i];
      });
    }
    cur_off += count;
  };
  auto exclude = [&](const Set &docs, uint64_t count) {
    if (!count) return;
    if (!excluded) excluded = c.dev.zeros();
    c.dev.or_(excluded, docs);
  };
  auto back = [&]() -> bool {  // false: iteration over
    unis[cur].reset();
    uni_counts[cur] = 0;
    rules[cur]->end();
    if (cur == 0) return false;
    --cur;
    if (scores.size() > cur) scores.pop_back();
    return true;
  };
  const bool trace = getenv("MSI_SEARCH_TRACE") != nullptr;  // one line per bucket on stderr (debugging aid)
  // max_len_to_evaluate, bucket_sort.rs:187-191
  const uint64_t max_len = (p->has_score_threshold && p->exhaustive_number_hits && p->max_total_hits) ? p->max_total_hits : length;
  // all_candidates at the end (and, search/new/mod.rs:894-907, what `distinct` keeps of it for an exhaustive count)
  auto finish = [&]() {
    c.dev.flush();   // the ids of the last buckets
    if (ids_short) fail(MSI_E_INTERNAL, "a bucket held fewer documents than its cardinality said");
    *out_n = n_out;
    if (!out_candidates) return;
    if (exhaustive_distinct) {
      Set all = c.dev.clone(universe);
      if (excluded) c.dev.sub_(all, excluded);
      *out_candidates = exhaustive_count(all);
    } else if (excluded) {
      *out_candidates = universe_count - c.dev.count(excluded);
    }
  };
  while (n_out < max_len) {
    if (uni_counts[cur] == 0 || (!detailed && uni_counts[cur] == 1)) {
      if (uni_counts[cur]) add(unis[cur], uni_counts[cur]);
      if (!back()) break;
      continue;
    }
    if (deadline_exceeded()) {
      // graph-based rules never answer non_blocking_next_bucket (ranking_rules.rs:67-74): what is left of every
      // universe on the stack goes out unranked under a Skipped detail (bucket_sort.rs:206-264)
      for (;;) {
        scores.push_back(Score{MSI_SCORE_SKIPPED, 0, 1});
        if (p->has_score_threshold && global_score(scores) < p->score_threshold) exclude(unis[cur], uni_counts[cur]);
        else add(unis[cur], uni_counts[cur]);
        scores.pop_back();
        unis[cur].reset();
        uni_counts[cur] = 0;
        if (cur == 0) {
          if (out_degraded) *out_degraded = 1;
          finish();
          return;
        }
        rules[cur]->end();
        --cur;
        if (scores.size() > cur) scores.pop_back();
      }
    }
    Bucket b;
    if (!rules[cur]->next(c, unis[cur], uni_counts[cur], b)) {
      if (!back()) break;
      continue;
    }
    ++g_stats.buckets;
    scores.push_back(b.score);
    const bool below = p->has_score_threshold && global_score(scores) < p->score_threshold;
    if (trace)
      fprintf(stderr, "[msi trace] rule %zu kind %d bucket %llu left %llu score (%u,%u,%u) below %d\n", cur, rules[cur]->kind,
              (unsigned long long)b.count, (unsigned long long)(uni_counts[cur] - b.count), b.score.kind, b.score.a,
              b.score.b, (int)below);
    if (!b.universe_reduced) c.dev.sub_(unis[cur], b.docs);
    uni_counts[cur] -= b.count;
    if (cur == nr - 1 || (!detailed && b.count <= 1) || cur_off + b.count < from || below) {
      if (below) {
        exclude(b.docs, b.count);
        exclude(unis[cur], uni_counts[cur]);
      } else {
        add(b.docs, b.count);
      }
      scores.pop_back();
      continue;
    }
    ++cur;
    unis[cur] = b.docs;
    uni_counts[cur] = b.count;
    rules[cur]->start(c, b.docs, b.graph());
  }
  finish();
}

}  // namespace

extern "C" int32_t msi_keyword_search_ranked(msi_dict *dict, msi_bits *pool, const msi_index_vtable *index,
                                             const msi_located_term *terms, uint32_t n_terms,
                                             const msi_search_params *params, const uint8_t *universe_cbo,
                                             size_t universe_len, uint32_t *out_docids, msi_score_detail *out_scores,
                                             uint32_t *out_n_scores, uint32_t *out_n, uint64_t *out_candidates,
                                             int32_t *out_degraded) {
  if (!dict || !pool || !index || !index->word_docids || !params || !out_n || (n_terms && !terms) ||
      n_terms > MSI_RANK_MAX_TERMS || (params->length && (!out_docids || !out_scores || !out_n_scores)) ||
      (params->n_criteria && !params->criteria) ||
      (params->n_searchable && (!params->searchable_fids || !params->searchable_weights))) {
    msi_set_error("msi_keyword_search_ranked: invalid argument (<= %d terms)", MSI_RANK_MAX_TERMS);
    return MSI_E_INVALID;
  }
  if (msi_bits_n_slots(pool) < 64) {
    msi_set_error("msi_keyword_search_ranked: the pool needs at least 64 slots, has %u", msi_bits_n_slots(pool));
    return MSI_E_INVALID;
  }
  *out_n = 0;
  if (out_candidates) *out_candidates = 0;
  if (out_degraded) *out_degraded = 0;
  g_stats = Stats();
  g_ranked_searches.fetch_add(1, std::memory_order_relaxed);
  Clock total;
#ifndef MSI_SEARCH_DIRECT_ONLY
  struct CpuProf {   // MSI_SEARCH_CPU_PROFILE (msi_vm.h): this search's thread CPU time, whichever way it ends
    bool on = msi_cpu_prof_on();
    uint64_t t0 = on ? msi_thread_cpu_ns() : 0;
    ~CpuProf() {
      if (on) {
        msi_cpu_prof_add(1, msi_thread_cpu_ns() - t0);
        msi_cpu_prof_add(0, 1);
        msi_cpu_prof_add(5, (uint64_t)(g_stats.callback_ms * 1e6));   // (wall time of the callbacks: they never sleep)
      }
    }
  } cpu_guard;
#endif
  try {
    msi_arena::ArenaScope host_memory;   // (declared before Ctx: released after everything the search built)
    Ctx c(dict, pool, index, params);
    search(c, terms, n_terms, universe_cbo, universe_len, out_docids, out_scores, out_n_scores, out_n, out_candidates,
           out_degraded);
    g_stats.total_ms = total.ms();
    return MSI_OK;
  } catch (const Fail &f) {
    return f.code;
  } catch (const std::bad_alloc &) {
    msi_set_error("msi_keyword_search_ranked: out of host memory");
    return MSI_E_OOM;
  } catch (const std::exception &e) {
    msi_set_error("msi_keyword_search_ranked: %s", e.what());
    return MSI_E_INTERNAL;
  }
}

// Universe compaction, process-wide: [ranked searches, of them continued in the compact space, documents of their universes summed]
extern "C" int32_t msi_search_compaction_stats(uint64_t out[3]) {
  if (!out) return MSI_E_INVALID;
  out[0] = g_ranked_searches.load();
  out[1] = g_compact_searches.load();
  out[2] = g_compact_docs.load();
  return MSI_OK;
}

// [sub-trees of the bucket sort that continued in the compact space of their bucket, documents of those buckets summed]
extern "C" int32_t msi_search_late_compaction_stats(uint64_t out[2]) {
  if (!out) return MSI_E_INVALID;
  out[0] = g_late_compactions.load();
  out[1] = g_late_compact_docs.load();
  return MSI_OK;
}

// Counters of the last msi_keyword_search_ranked on the calling thread: [launches, syncs, decode batches,
// index callbacks, posting bytes decoded, matching paths, buckets, callback us, device wait us, total us].
extern "C" int32_t msi_search_last_stats(uint64_t out[10]) {
  if (!out) return MSI_E_INVALID;
  const Stats &s = g_stats;
  const uint64_t v[10] = {s.launches, s.syncs, s.decodes, s.callbacks, s.postings_bytes, s.paths, s.buckets,
                          (uint64_t)(s.callback_ms * 1e3), (uint64_t)(s.device_wait_ms * 1e3), (uint64_t)(s.total_ms * 1e3)};
  memcpy(out, v, sizeof(v));
  return MSI_OK;
}
