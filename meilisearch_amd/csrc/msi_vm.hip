// msi_vm.hip — command lists over docid sets: one launch per dependency round, shared by every keyword search in
// flight on the context.  Design and rationale: msi_vm.h; callers: msi_search.hip (`Dev`).
//
// Replaces, for the ranked keyword search, one kernel launch per RoaringBitmap operation of
// crates/milli/src/search/new/{graph_based_ranking_rule.rs:383-437 (visit_path_condition), resolve_query_graph.rs:33-130
// (term / phrase docids), bucket_sort.rs:23-343 (universe bookkeeping), sort.rs:95-233 (next bucket of a Sort rule)} and
// heed_codec/roaring_bitmap/cbo_roaring_bitmap_codec.rs:53-85 (posting decode).
//
// Device side: a workgroup = (list, 65 536-document chunk).  Set words are handled as 16-byte pairs with a FIXED
// thread <-> pair mapping, so consecutive element-wise commands need no barrier (a thread only re-reads what it wrote
// itself); commands with another mapping (container decode through LDS, the one-document-per-thread key commands)
// are fenced by workgroup barriers.  Cardinalities accumulate in LDS and leave the workgroup once, at the end of its
// list; the last workgroup of a list's last phase copies them into the search's pinned result block and stores the
// sequence number with system-scope release — the search thread polls that word, no stream synchronisation.
#include <string.h>
#if defined(__linux__)
#include <linux/futex.h>
#include <sys/prctl.h>
#include <time.h>
#include <sys/syscall.h>
#include <time.h>
#include <unistd.h>
#endif

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <memory>
#include <shared_mutex>
#include <thread>
#include <unordered_map>

#include "msi_common.h"
#include "msi_vm.h"

typedef unsigned long long u64;

msi_ctx *msi_bits_ctx(msi_bits *p);
u64 *msi_bits_slot_ptr(msi_bits *p, uint32_t slot);
u64 *msi_bits_pool_base(msi_bits *p);
u64 *msi_bits_summary(msi_bits *p);
uint64_t msi_bits_words_per_slot(msi_bits *p);
uint64_t msi_bits_n_docs(msi_bits *p);
uint32_t msi_bits_n_slots(msi_bits *p);
uint64_t msi_bits_compact_capacity(const msi_bits *p);
uint32_t *msi_bits_compact_aux(msi_bits *p);
uint64_t *msi_bits_vm_block(msi_bits *p);     // pinned, fine-grained: [0] seq, [1] first-k count, [2..] counts, then ids
uint64_t msi_bits_vm_next_seq(msi_bits *p);
uint8_t *msi_bits_vm_stage(msi_bits *p, size_t need, size_t keep);   // pinned staging of the pool's decode payloads (grows, keeps `keep` bytes)

namespace {

constexpr int VT = 256;                 // threads per workgroup (512 measured 30 % slower: r2 notes in DESIGN §4.7)
constexpr int WPT = 1024 / VT;          // words of a chunk per thread in the ordered emit
constexpr uint32_t CHW = 1024;          // u64 words per chunk (65 536 documents = one Roaring container span)
constexpr uint32_t MAX_SUBS = 64;       // lists per round
constexpr uint32_t SUM_W = 16;                                  // 64-bit words of a chunk's summary row: pools of <= 1024 slots
constexpr uint32_t CMD_LDS = 2048;                             // command words of a phase kept in LDS (longer phases: read from the arena)
// One LDS arena per workgroup, used three ways (the phases of a launch never mix them inside one workgroup):
//   full-space list phases   s_dec (a chunk being decoded) | s_raw (one container body) | s_whole | s_nzw (chunk summaries)
//   wide phase (VM_DECODEC)  s_dec (U0's words of the chunk) | s_raw (the waves' output words) | the chunk's decode descriptors
//   compact command phase    the SET CACHE: slot -> entry map and resolved path steps per wave, then the cached sets
// The set cache is a compile-time option, OFF by default: measured on the MI355X (round 4, profiles/r4_ranked_variants.txt) it
// LOST — 6.6-7.2 k keyword searches/s against 9.5-9.7 k without it at 128 callers, 3.96 against 3.53 ms for one caller.  The
// premise was wrong for compact lists: VM_PATHS is 9 % of a workgroup's time once the sets are a chunk or two long (their
// operands sit in L2; a list's 8-480 paths are a handful of dependent round trips), the cache's bookkeeping per level
// costs more than the round trips it removes, and its LDS took a workgroup per CU away from the wide phase, which is
// where a round's device time goes.  -DMSI_VM_SET_CACHE=1 -DMSI_VM_ARENA_KB=38 builds it (tests: MSI_VM_CACHE=1).
#ifndef MSI_VM_SET_CACHE
#define MSI_VM_SET_CACHE 0
#endif
#ifndef MSI_VM_ARENA_KB
#define MSI_VM_ARENA_KB (MSI_VM_SET_CACHE ? 38 : 23)
#endif
constexpr uint32_t ARENA_BYTES = MSI_VM_ARENA_KB * 1024;
constexpr uint32_t A_RAW = CHW * 8, A_WHOLE = A_RAW + CHW * 8 + 32, A_NZW = A_WHOLE + SUM_W * 64 * 2, A_FULL_END = A_NZW + SUM_W * 64 * 4;
constexpr uint32_t A_DESC = A_WHOLE;                             // wide phase: the chunk's decode descriptors (as much as fits)
constexpr uint32_t DESC_WORDS = (ARENA_BYTES - A_DESC) / 4;
constexpr uint32_t SO_CAP = 1024;                                // path steps of one VM_PATHS resolved ahead per wave (u8 each)
constexpr uint32_t C_MAP = 0, C_SO = C_MAP + (VT / 64) * 1024, C_DATA = C_SO + (VT / 64) * SO_CAP;
constexpr uint32_t CACHE_MAX_ENTRIES = 32;                       // an entry per lane of the bookkeeping registers' low half
static_assert(A_FULL_END <= ARENA_BYTES && (!MSI_VM_SET_CACHE || C_DATA + 8192 <= ARENA_BYTES), "the LDS arena holds every use of it");
constexpr size_t RES_COUNTS = 2;                               // u64 index of counts[0] in the result block
constexpr u64 MSI_VM_RES_FAILED = ~0ull;                       // res[1] of a list whose fused workgroups gave up waiting
// ticks of the 100 MHz wall clock a workgroup of a fused list waits for the list's wide phase before it gives up: 2 s — four
// orders of magnitude above a wide phase under load (tens of microseconds), below the host's own 5 s watchdog
#ifndef MSI_VM_SPIN_LIMIT_TICKS
#define MSI_VM_SPIN_LIMIT_TICKS 200000000ull
#endif
constexpr size_t RES_IDS = RES_COUNTS + MSI_VM_MAX_COUNTS;     // u64 index where the u32 ids start

struct alignas(16) RoundSub {
  u64 pool_base, n_words, n_docs, host_res, seq;
  u64 stage;                              // the pool's pinned staging buffer (decode payloads), device-visible
  u64 cache;                              // device base of the HBM posting cache (0: none)
  uint32_t n_chunks, n_phases;
  uint32_t phase_off[MSI_VM_MAX_PHASES];  // arena word offsets of each phase's first command
  uint32_t list_off;                      // arena word offset of the list's words
  uint32_t data_off;                      // word offset, from list_off, of the chunk-major decode descriptors
  uint32_t state_off;                     // arena word offset of {done[4] u32, cells[4] u64, counts[n_counts] u64, chunk cardinalities[n_chunks] u32}
  uint32_t n_counts;
  uint32_t n_decodes;
  uint32_t n_cmd_words;                   // the list's command words (its decode descriptors follow them)
  uint32_t sum_lo, sum_hi;                // device address of the pool's chunk summaries (0: none), msi_bits_summary
  // compact lists (universe compaction, msi_vm.h): the full pool's compaction tables and slots, U0, and which phases
  // are WIDE (one workgroup per chunk of the full space: `wide_chunks` of them)
  u64 aux, full_base;
  uint32_t full_words, u0_slot, wide_chunks, wide_mask;
  // words per workgroup in the list's command phases: CHW (a Roaring container span) for lists over docids; compact lists
  // take narrower chunks — more workgroups per list, and a chunk of a set small enough that a dozen sets fit in LDS
  uint32_t chw, cache_on;
  // written by the DEVICE (the round's own copy of this struct): a workgroup of a fused list gave up waiting for the list's
  // wide phase (MSI_VM_SPIN_LIMIT_TICKS).  The list still hands in its tickets, and its last workgroup publishes the failure
  // instead of a result count, so that a stall surfaces as MSI_E_INTERNAL on the host and not as a hung device.
  uint32_t failed;
  // workgroups of the list's phase 0 when it is a decode phase of its own: `wide_chunks` (one per chunk of the full space) or,
  // BY RANK (wide_mask bit 30), one per 64 documents of U0
  uint32_t p0_wgs;
};
static_assert(sizeof(RoundSub) % 16 == 0, "RoundSub array stays 16-byte aligned");

// A container as the decode command reads it.  The descriptors of ALL the decode commands of a list travel with its
// commands (arena, one bulk H2D copy per round) laid out CHUNK-MAJOR — for chunk c: how many containers each decode has
// in c, then those containers back to back — so a workgroup reads its own, contiguous, device-resident block.  (With the
// descriptors in the pinned staging buffer every workgroup paid three dependent PCIe reads per decode command: 90 k small
// reads per 64-list round, 0.7 ms — r2_ranked_vm_trace_descriptor_reads_over_pcie.txt.)
struct alignas(16) VmContainer {
  uint32_t meta;       // card (16 bits, array: values, run: runs) | type << 16 | cached << 18
  uint32_t fill_lo;    // low / high half of the posting-cache offset to store the body at; ~0 / ~0 = none
  u64 src;             // byte offset of the body: in the posting cache (cached) or in the pool's pinned staging buffer
};
// high half of the fill offset rides in the upper bits of `meta` (bits 19..31: 13 bits -> 45-bit offsets)

__device__ __forceinline__ uint32_t wave_sum(uint32_t c) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) c += __shfl_xor((int)c, o);
  return c;
}

__device__ __forceinline__ ulonglong2 apply_op(uint32_t op, ulonglong2 x, ulonglong2 y) {
  ulonglong2 r;
  if (op == MSI_BITS_AND) { r.x = x.x & y.x; r.y = x.y & y.y; }
  else if (op == MSI_BITS_OR) { r.x = x.x | y.x; r.y = x.y | y.y; }
  else if (op == MSI_BITS_ANDNOT) { r.x = x.x & ~y.x; r.y = x.y & ~y.y; }
  else { r.x = x.x ^ y.x; r.y = x.y ^ y.y; }
  return r;
}

// bits of word `gw` of a set that are documents (< n_docs)
__device__ __forceinline__ u64 doc_mask(u64 gw, u64 n_docs) {
  const u64 lo = gw * 64;
  if (lo + 64 <= n_docs) return ~0ull;
  if (lo >= n_docs) return 0ull;
  return (~0ull) >> (64 - (n_docs - lo));
}

// Every store of set words (and posting-cache bodies) is WRITE-THROUGH at device scope (global_store ... sc1): what a
// list wrote is read by the search's next list — another kernel, on another stream and other XCDs, launched as soon as
// the host sees this list's results, which can be before this kernel ends.  With write-back stores each workgroup had
// to write its XCD's L2 back before its ticket (buffer_wbl2: 10 M documents, 64 threads: 1345 q/s); written through,
// the ticket only waits for the stores' acknowledgements (2147 q/s).  The line stays valid in this XCD's L2 for the
// commands that follow.
__device__ __forceinline__ void put1(u64 *p, u64 v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void put(ulonglong2 *p, ulonglong2 v) {
  put1(&p->x, v.x);
  put1(&p->y, v.y);
}
__device__ __forceinline__ void put4(uint4 *p, uint4 v) {
  u64 *q = reinterpret_cast<u64 *>(p);
  put1(q, (u64)v.x | ((u64)v.y << 32));
  put1(q + 1, (u64)v.z | ((u64)v.w << 32));
}

__global__ __launch_bounds__(VT, 4) void vm_kernel(uint32_t *__restrict__ arena, uint32_t launch_phase) {
  __shared__ uint32_t s_cnt[MSI_VM_MAX_COUNTS];
  __shared__ __attribute__((aligned(16))) unsigned char s_arena[ARENA_BYTES];
  u64 *const s_dec = reinterpret_cast<u64 *>(s_arena);              // [CHW]
  uint4 *const s_raw = reinterpret_cast<uint4 *>(s_arena + A_RAW);  // one container body (<= 8 KiB) + alignment slack, staged with wide loads
  uint16_t *const s_whole = reinterpret_cast<uint16_t *>(s_arena + A_WHOLE);   // [SUM_W * 64]
  uint32_t *const s_nzw = reinterpret_cast<uint32_t *>(s_arena + A_NZW);       // [SUM_W * 64]
  __shared__ uint32_t s_scan[VT / 64 + 1];
  __shared__ uint32_t s_last;
  __shared__ int s_any;
  __shared__ u64 s_sum[VT / 64][SUM_W];
  __shared__ uint32_t s_cmd[CMD_LDS];          // this phase's command words (read once, coalesced)
  const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // (fields are read one by one: a by-value copy of the struct lands in scratch because phase_off[] is indexed dynamically)
  const RoundSub *const rp = reinterpret_cast<const RoundSub *>(arena + 16) + blockIdx.y;
  struct {
    u64 pool_base, n_words, n_docs, host_res, seq, stage, cache;
    uint32_t n_chunks, n_phases, list_off, data_off, state_off, n_counts, n_decodes;
  } r;
  r.pool_base = rp->pool_base; r.n_words = rp->n_words; r.n_docs = rp->n_docs; r.host_res = rp->host_res; r.seq = rp->seq;
  r.stage = rp->stage; r.cache = rp->cache; r.n_chunks = rp->n_chunks; r.n_phases = rp->n_phases; r.list_off = rp->list_off;
  r.data_off = rp->data_off; r.state_off = rp->state_off; r.n_counts = rp->n_counts; r.n_decodes = rp->n_decodes;
  // A compact list whose phase 0 is WIDE (VM_DECODEC, one workgroup per chunk of the full space) and whose sets are a
  // few chunks long runs its phases 0 and 1 in ONE launch (`fused`): the first wide_chunks workgroups of the list are the
  // wide phase, the workgroups behind them are phase 1 and start once the list's OWN wide phase has handed in all its
  // tickets.  (As two launches every list of a round waited for the slowest wide phase of the round before its
  // commands could start — multi-list rounds took 4-5x a single list's time, profiles/r3_ranked10_trace_two_launches.txt.)
  const bool fused = (rp->wide_mask & 0x80000000u) != 0;
  uint32_t phase = launch_phase, chunk = blockIdx.x;
  if (fused) {
    if (launch_phase == 1) return;               // ran with launch 0
    if (launch_phase == 0 && chunk >= rp->p0_wgs) {
      phase = 1;
      chunk -= rp->p0_wgs;
    }
  }
  const bool wide = ((rp->wide_mask >> phase) & 1u) != 0;       // a compact list's VM_DECODEC phase: chunks of the FULL space
  const bool by_rank = wide && (rp->wide_mask & 0x40000000u) != 0;   // ... or, for a small U0, 64 documents of U0 per workgroup
  const uint32_t my_chunks = wide ? rp->p0_wgs : r.n_chunks;
  if (phase >= r.n_phases || chunk >= my_chunks) return;
  // a waiter that gave up: its workgroup runs NO command (the wide phase's data is incomplete: nothing may be computed from
  // it into pool slots or posting-cache entries that other searches share — ADVICE r5), it only hands in its ticket so
  // that the list's last workgroup can publish the failure
  __shared__ uint32_t s_abandon;
  if (threadIdx.x == 0) s_abandon = 0;
  if (fused && launch_phase == 0 && phase == 1) {
    // the wide workgroups of this list were dispatched before this one (lower block indices): wait for their tickets
    // (bounded: a wait that outlasts MSI_VM_SPIN_LIMIT_TICKS marks the list failed — ADVICE r4)
    const uint32_t *done0 = arena + r.state_off;
    if (threadIdx.x == 0) {
      const u64 t_wait = wall_clock64();
      uint32_t spins = 0;
      while (__hip_atomic_load(done0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < rp->p0_wgs) {
        MSI_SLEEP();
        if ((++spins & 1023u) == 0 && wall_clock64() - t_wait > MSI_VM_SPIN_LIMIT_TICKS) {
          __hip_atomic_store(const_cast<uint32_t *>(&rp->failed), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          s_abandon = 1;
          break;
        }
      }
      // (a sibling workgroup's give-up counts too: the list is failed as a whole)
      if (__hip_atomic_load(&rp->failed, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) s_abandon = 1;
      u64 *const wprof = reinterpret_cast<u64 *>(((u64)arena[3] << 32) | arena[2]);
      if (wprof) {   // MSI_VM_PROFILE: how long the workgroups of fused lists wait for their wide phase
        atomicAdd(&wprof[24], wall_clock64() - t_wait);
        atomicAdd(&wprof[25], 1ull);
      }
    }
    __syncthreads();
    // what they wrote (write-through stores and device-scope atomics) is in memory; drop what this XCD's caches may
    // still hold of those lines
    MSI_ACQUIRE_DEVICE();
  }
  // MSI_VM_PROFILE (diagnostics): thread 0's wall-clock ticks (100 MHz) per opcode, summed over all workgroups
  u64 *const prof = reinterpret_cast<u64 *>(((u64)arena[3] << 32) | arena[2]);
  __shared__ u64 s_prof[16];
  if (prof && tid < 16) s_prof[tid] = 0;
  const u64 t_begin = prof ? wall_clock64() : 0;
  for (uint32_t i = tid; i < r.n_counts; i += VT) s_cnt[i] = 0;
  // The phase's commands go to LDS and every command word is read as a wave-uniform scalar.  (Read from the arena with
  // per-lane loads — the compiler cannot know they are uniform — each command cost two dependent trips to memory, opcode
  // then operands, before its first set word was even requested, and the interpreter branched on vector compares.)
  const uint32_t p_begin = rp->phase_off[phase];
  const uint32_t p_end = phase + 1 < r.n_phases ? rp->phase_off[phase + 1] : r.list_off + rp->n_cmd_words;
  const bool in_lds = p_end - p_begin <= CMD_LDS;
  if (in_lds)
    for (uint32_t i = tid; i < p_end - p_begin; i += VT) s_cmd[i] = arena[p_begin + i];
  __syncthreads();
  const uint32_t *const cmd = in_lds ? s_cmd : arena + p_begin;
  const bool abandoned = s_abandon != 0;   // (written by thread 0 before the barrier above)
  uint32_t pcw = 0;                              // word index of the current command
#define W(i) MSI_UNIFORM(cmd[pcw + (i)])
  const uint32_t chw = wide ? CHW : rp->chw;     // words per workgroup (RoundSub::chw)
  const u64 w0 = (u64)chunk * chw;
  const uint32_t nw = wide ? 0u : (uint32_t)min((u64)chw, r.n_words - w0);   // even: slots are whole 16-byte pairs
  const uint32_t n_pairs = nw / 2;
  u64 *const pool = reinterpret_cast<u64 *>(r.pool_base);
  auto S = [&](uint32_t slot) -> ulonglong2 * { return reinterpret_cast<ulonglong2 *>(pool + (u64)slot * r.n_words + w0); };
  uint32_t *const state = arena + r.state_off;
  u64 *const cells = reinterpret_cast<u64 *>(state + 4);
  u64 *const counts = cells + MSI_VM_CELLS;
  // first-k commands of this phase: {slot, k, cardinality index, ids base} (wave-uniform; kept in LDS for the emit)
  __shared__ uint32_t s_fk[MSI_VM_MAX_FK_PHASE][4];
  uint32_t n_fk = 0;
  uint32_t *const chunk_card = reinterpret_cast<uint32_t *>(counts + r.n_counts);   // first-k: cardinality of the set per chunk
  auto add_count = [&](uint32_t idx, uint32_t c) {
    c = wave_sum(c);
    if (lane == 0 && c) atomicAdd(&s_cnt[idx], c);
  };

  uint32_t cmd_no = 0;
  // ---- set cache (compact command phases) ---------------------------------------------------------------------------
  // Round 3 measured the keyword leg at 552 MB of set operands per query through L2 for 116 MB of HBM traffic: every
  // command loaded its operands from memory again, and a VM_PATHS level re-read the same few dozen condition sets once
  // per path step — a chain of dependent L2 round trips per (path group, pair).  A compact list's chunk of a set is
  // `chw * 8` bytes (2 KiB at the default 256 words), so the sets a list works with FIT IN LDS: an entry per set, the
  // thread that owns pair p of the chunk keeps pair p of every cached set at entry + p — the same thread <-> pair mapping
  // as in memory, so element-wise commands still need no barrier, and the waves of the workgroup never synchronise:
  // each keeps its own copy of the bookkeeping (slot -> entry map in LDS, entry -> slot and last-use stamps in the lanes
  // of two registers) and, every decision being a function of the command stream alone, all copies agree.
  // Write-through: a store goes to memory (the next list, another workgroup's first-k emit read it there) AND to the
  // cached copy; nothing is ever written back, eviction is free.  Sets enter the cache where it pays: the operands of
  // VM_PATHS (conditions, universe, bucket); every other command reads through it and updates what is there.
  const uint32_t ent_pairs = chw / 2;                                   // pairs per cache entry
  const uint32_t n_ent = ARENA_BYTES > C_DATA ? min(CACHE_MAX_ENTRIES, (ARENA_BYTES - C_DATA) / (chw * 8)) : 0u;
  const bool cache_on = MSI_VM_SET_CACHE && rp->cache_on != 0 && !wide && rp->aux != 0 && ent_pairs <= (uint32_t)VT && n_ent >= 4 &&
                        wave * 64 < n_pairs;                            // (a wave that owns no pair of this chunk keeps no cache)
  uint8_t *const c_map = s_arena + C_MAP + wave * 1024;                 // slot -> entry + 1 (0: not cached); pools of <= 1024 slots
  uint8_t *const c_so = s_arena + C_SO + wave * SO_CAP;                 // the current VM_PATHS: step -> entry + 1
  ulonglong2 *const c_data = reinterpret_cast<ulonglong2 *>(s_arena + C_DATA);
  uint32_t v_eslot = 0xFFFFFFFFu, v_estamp = 0;                         // lane e: the slot entry e holds, the command that used it last
  if (cache_on) {
    for (uint32_t i = lane; i < 1024 / 4; i += 64) reinterpret_cast<uint32_t *>(c_map)[i] = 0;
    __builtin_amdgcn_wave_barrier();
  }
  auto c_find = [&](uint32_t slot) -> int {   // the entry that holds `slot`, or -1 (wave-uniform)
    if (!cache_on) return -1;
    __builtin_amdgcn_wave_barrier();
    const int e = (int)MSI_UNIFORM((uint32_t)c_map[slot]) - 1;
    __builtin_amdgcn_wave_barrier();
    return e;
  };
  struct Ref {
    ulonglong2 *g, *l;   // the chunk of the set in memory; its cached copy (or null)
  };
  auto R = [&](uint32_t slot) -> Ref {
    Ref x;
    x.g = S(slot);
    const int e = c_find(slot);
    x.l = e >= 0 ? c_data + (size_t)e * ent_pairs : nullptr;
    return x;
  };
  auto LD = [&](const Ref &x, uint32_t p) -> ulonglong2 {
#ifdef MSI_VM_DEBUG_CACHE
    if (x.l && (x.l[p].x != x.g[p].x || x.l[p].y != x.g[p].y))
      printf("[cache] LD chunk %u cmd %u pair %u: lds %llx %llx mem %llx %llx\n", chunk, cmd_no, p, x.l[p].x, x.l[p].y, x.g[p].x, x.g[p].y);
#endif
    return x.l ? x.l[p] : x.g[p];
  };
  auto ST = [&](const Ref &x, uint32_t p, ulonglong2 v) {
    put(&x.g[p], v);
    if (x.l) x.l[p] = v;
  };
  // `slot` is an operand of the current command: keep it if it is cached, else give it an entry — a free one, or the one
  // used longest ago by an EARLIER command (entries of the current command are never evicted: the command may hold
  // references to them) — and note the fill in lane *n_fill of *v_fill.  Nothing is loaded here: the fills of a command
  // are issued together (c_fill) so that they cost one round trip, not one each.
  uint32_t v_fill = 0;
  auto c_plan = [&](uint32_t slot, uint32_t &n_fill) {
    __builtin_amdgcn_wave_barrier();
    const uint32_t have = MSI_UNIFORM((uint32_t)c_map[slot]);
    if (have) {
      if (lane == have - 1) v_estamp = cmd_no;
      return;
    }
    // key per entry: 0 free, 1 + stamp used by an earlier command, ~0 in use by this command; lanes >= n_ent: ~0
    uint32_t key = lane < n_ent ? (v_eslot == 0xFFFFFFFFu ? 0u : (v_estamp == cmd_no ? 0xFFFFFFFFu : 1u + v_estamp)) : 0xFFFFFFFFu;
    uint32_t mn = key;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mn = min(mn, (uint32_t)__shfl_xor((int)mn, o));
    if (mn == 0xFFFFFFFFu) return;   // every entry belongs to this command: the set is read from memory
    const uint32_t e = (uint32_t)__ffsll((long long)__ballot(key == mn)) - 1;
    const uint32_t old = (uint32_t)__shfl((int)v_eslot, (int)e);
    if (lane == 0) {
      if (old != 0xFFFFFFFFu) c_map[old] = 0;
      c_map[slot] = (uint8_t)(e + 1);
    }
    if (lane == e) {
      v_eslot = slot;
      v_estamp = cmd_no;
    }
    if (lane == n_fill) v_fill = slot | (e << 16);
    ++n_fill;
    __builtin_amdgcn_wave_barrier();
  };
  auto c_fill = [&](uint32_t n_fill) {   // the planned entries <- memory, eight loads in flight per thread
    const bool mine = tid < n_pairs;
    for (uint32_t g0 = 0; g0 < n_fill; g0 += 8) {
      ulonglong2 v[8];
      uint32_t it[8];
#pragma unroll
      for (uint32_t u = 0; u < 8; ++u) {
        it[u] = (uint32_t)__shfl((int)v_fill, (int)min(g0 + u, 63u));
        v[u] = make_ulonglong2(0, 0);
        if (g0 + u < n_fill && mine) v[u] = S(it[u] & 0xFFFFu)[tid];
      }
#pragma unroll
      for (uint32_t u = 0; u < 8; ++u)
        if (g0 + u < n_fill && mine) c_data[(size_t)(it[u] >> 16) * ent_pairs + tid] = v[u];
    }
  };

  // ---- chunk summaries ------------------------------------------------------------------------------------------
  // One bit per (slot, chunk): 0 = this chunk of the slot IS all zero (and its words in memory are zero), 1 = it may hold
  // documents.  84 % of the (command, chunk) executions of a detailed search at 10 M documents work on an empty chunk
  // of their scoping operand (a bucket of three documents occupies three of 153 chunks;
  // profiles/r2_ranked_10m_vm_kernel_opcode_profile.txt): with the bit known, the workgroup neither loads nor stores.
  // The row of this chunk (one bit per slot) is loaded once, kept per wave in LDS (every lane writes the same value to
  // the same word: no barrier), and written back at the end; bits follow conservatively from the operands' bits, and a
  // slot whose chunk was written WHOLE by this list gets its exact bit from what was stored.
  u64 *const sum_row = reinterpret_cast<u64 *>(((u64)rp->sum_hi << 32) | rp->sum_lo);
  const bool sum_on = sum_row != nullptr && !wide;
  if (sum_on) {
    if (lane < SUM_W) s_sum[wave][lane] = sum_row[(u64)chunk * SUM_W + lane];
    for (uint32_t i = tid; i < SUM_W * 64; i += VT) {
      s_whole[i] = 0;
      s_nzw[i] = 0;
    }
  }
  __syncthreads();
  // wide phase (VM_DECODEC): this chunk's words of U0 and their exclusive prefix counts, once for all the phase's commands
  uint32_t c_lo = 0, c_hi = 0;
  uint32_t *const s_desc = reinterpret_cast<uint32_t *>(s_arena + A_DESC);
  const uint32_t *desc_g = nullptr;   // the chunk's descriptor block in memory; its first desc_n words are in s_desc
  uint32_t desc_n = 0;
  const uint32_t desc_st = (r.n_decodes + 1 + 3) & ~3u;   // words of the block's start[] table
  auto DW = [&](uint32_t i) -> uint32_t { return i < desc_n ? s_desc[i] : desc_g[i]; };
  auto DC = [&](uint32_t ci) -> VmContainer {   // container ci of the chunk's block
    const uint32_t w = desc_st + 4 * ci;
    if (w + 4 <= desc_n) return *reinterpret_cast<const VmContainer *>(s_desc + w);
    return *reinterpret_cast<const VmContainer *>(desc_g + w);
  };
  // a 16-bit value of a container body (bodies are 2-byte aligned in every serialisation roaring writes; behind a run
  // cookie with fewer than four containers they can start at an odd offset: bytes then)
  auto ld16 = [&](uintptr_t b0, uint32_t i) -> uint32_t {
    if (b0 & 1) {
      const uint8_t *q = reinterpret_cast<const uint8_t *>(b0) + 2 * (size_t)i;
      return (uint32_t)q[0] | ((uint32_t)q[1] << 8);
    }
    return reinterpret_cast<const uint16_t *>(b0)[i];
  };
  if (wide && !by_rank) {
    const uint32_t full_words = rp->full_words, full_chunks = rp->wide_chunks;
    const uint32_t *prefix = reinterpret_cast<const uint32_t *>(rp->aux) + ((full_chunks + 3) & ~3u);
    const u64 fw0 = (u64)chunk * CHW;
    const uint32_t nwf = (uint32_t)min((u64)CHW, (u64)full_words - fw0);
    const u64 *u0 = reinterpret_cast<const u64 *>(rp->full_base) + (u64)rp->u0_slot * full_words + fw0;
    c_lo = prefix[fw0];
    c_hi = chunk + 1 < full_chunks ? prefix[fw0 + CHW] : (uint32_t)r.n_docs;
    // (a chunk without a document of U0 — most chunks of a universe of a few hundred documents — decodes nothing: its
    // 24 KB of tables stay where they are; the workgroup still fills the posting cache with what it is first to read)
    if (c_hi != c_lo)
      for (uint32_t i = tid; i < CHW; i += VT) {
        s_dec[i] = i < nwf ? u0[i] : 0ull;
        s_cnt[i] = i < nwf ? prefix[fw0 + i] : 0u;
      }
    // this chunk's decode descriptors (one contiguous block of the list: start[n_decodes + 1], padded to 16 bytes, then the
    // 16-byte containers) come to LDS with the tables: read from memory per command they were three dependent loads — block
    // offset, container range, container — in front of every decode's first posting byte
    if (r.n_decodes) {
      const uint32_t *data = arena + r.list_off + r.data_off;
      const uint32_t o0 = data[chunk], o1 = data[chunk + 1];   // 16-byte units
      desc_g = data + 4 * (size_t)o0;
      desc_n = min((o1 - o0) * 4u, DESC_WORDS);
      for (uint32_t i = tid; i < desc_n / 4; i += VT) reinterpret_cast<uint4 *>(s_desc)[i] = reinterpret_cast<const uint4 *>(desc_g)[i];
    }
    __syncthreads();
    if (prof && tid == 0) {
      atomicAdd(&prof[22], 1ull);                          // wide workgroups ...
      atomicAdd(&prof[23], wall_clock64() - t_begin);      // ... and what staging U0's tables + the descriptors cost them
    }
  }
  // (one lane writes, the wave's lanes read: wave_barrier keeps the compiler — and the CPU emulation, whose lanes are
  // fibers — from moving a read across a write)
  auto E = [&](uint32_t slot) -> bool {   // this chunk of `slot` is known to be empty
    if (!sum_on) return false;
    __builtin_amdgcn_wave_barrier();
    const uint32_t b = MSI_UNIFORM((uint32_t)((s_sum[wave][slot >> 6] >> (slot & 63)) & 1ull));
    __builtin_amdgcn_wave_barrier();
    return b == 0;
  };
  auto set_bit = [&](uint32_t slot, bool maybe) {
    if (!sum_on) return;
    __builtin_amdgcn_wave_barrier();
    if (lane == 0) {
      const u64 m = 1ull << (slot & 63), w = s_sum[wave][slot >> 6];
      s_sum[wave][slot >> 6] = maybe ? (w | m) : (w & ~m);
    }
    __builtin_amdgcn_wave_barrier();
  };
  // dst's chunk is written whole by the current command: its exact bit is settled at the end (nz() marks content)
  auto whole = [&](uint32_t slot) {
    set_bit(slot, true);
    if (sum_on && lane == 0) s_whole[slot] = (uint16_t)cmd_no;
  };
  auto partial = [&](uint32_t slot) {   // written in place, pair by pair: only "may hold documents" is known
    set_bit(slot, true);
    if (sum_on && lane == 0) s_whole[slot] = 0;
  };
  // after the store loop of a whole write: did any lane store a document?  (the latest command number wins)
  auto nz = [&](uint32_t slot, bool any) {
    if (!sum_on) return;
    if (__any(any) && lane == 0) atomicMax(&s_nzw[slot], cmd_no);
  };
  // dst := empty.  Its words are only stored when they may hold something.
  auto make_empty = [&](uint32_t slot) {
    if (E(slot)) return;
    const Ref d = R(slot);
    for (uint32_t p = tid; p < n_pairs; p += VT) ST(d, p, make_ulonglong2(0, 0));
    set_bit(slot, false);
    if (sum_on && lane == 0) s_whole[slot] = 0;
  };

  // MSI_VM_PROFILE: how many (command, chunk) executions work on an all-zero chunk of their scoping operand
  auto chunk_empty = [&](uint32_t slot) -> bool {
    const ulonglong2 *a = S(slot);
    int nz = 0;
    for (uint32_t p = tid; p < n_pairs; p += VT) nz |= (a[p].x | a[p].y) != 0;
    if (tid == 0) s_any = 0;
    __syncthreads();
    if (nz) s_any = 1;
    __syncthreads();
    const bool e = s_any == 0;
    __syncthreads();
    return e;
  };
  // The containers of THIS chunk (= blockIdx.x: a chunk of the space the postings are stored in) of decode `di` of the
  // list, OR-ed into s_dec (when want_bits); bodies whose descriptor says so go into the posting cache on the way.
  // -> number of containers here.  Ends with every thread past the last barrier.
  auto decode_chunk = [&](uint32_t di, bool want_bits) -> uint32_t {
    // chunk-major descriptor block of this workgroup (device memory): counts per decode, then the containers
    const uint32_t *data = arena + r.list_off + r.data_off;
    const uint32_t *blk_c = data + 4 * (size_t)data[chunk];          // chunk_off[] is in 16-byte units
    const uint32_t c_first = blk_c[di], n_here = blk_c[di + 1] - c_first;   // start[] of this chunk: n_decodes + 1 entries
    const VmContainer *cs = reinterpret_cast<const VmContainer *>(blk_c + ((r.n_decodes + 1 + 3) & ~3u)) + c_first;
    if (!n_here) return 0;
    __syncthreads();
    if (want_bits)
      for (uint32_t i = tid; i < CHW; i += VT) s_dec[i] = 0;
    __syncthreads();
    for (uint32_t ci = 0; ci < n_here; ++ci) {
      const VmContainer c = cs[ci];
      const uint32_t card = c.meta & 0xFFFFu, type = (c.meta >> 16) & 3u;
      const bool cached = (c.meta >> 18) & 1u;
      const u64 fill_off = ((u64)(c.meta >> 19) << 32) | c.fill_lo;
      const bool do_fill = !(c.fill_lo == 0xFFFFFFFFu && (c.meta >> 19) == 0x1FFFu);
      if (!want_bits && !do_fill) continue;
      // the body is read once, as 16-byte aligned loads (any body alignment) — from the HBM posting cache, or over
      // PCIe from the pinned staging buffer — and decoded from LDS
      const uint32_t len = type == 0 ? 2 * (card + 1) : (type == 1 ? 8192u : 4 * (card + 1));
      const uintptr_t b0 = cached ? (uintptr_t)(r.cache + c.src) : (uintptr_t)(r.stage + c.src);
      const uint32_t skew = (uint32_t)(b0 & 15);
      const uint4 *src = reinterpret_cast<const uint4 *>(b0 - skew);
      const uint32_t n16 = (skew + min(len, 8192u) + 15) / 16;
      uint4 *fill = do_fill ? reinterpret_cast<uint4 *>((uintptr_t)(r.cache + fill_off) - skew) : nullptr;
      for (uint32_t i = tid; i < n16; i += VT) {
        const uint4 v = src[i];
        s_raw[i] = v;
        // first reader of this key: the body goes into the cache on the way (same skew there: serialisations start
        // 16-byte aligned in both places; the partial blocks at the ends carry the neighbouring bytes of the SAME
        // serialisation, so a racing neighbour writes identical values)
        if (fill) put4(&fill[i], v);
      }
      __syncthreads();
      const uint8_t *body = reinterpret_cast<const uint8_t *>(s_raw) + skew;
      if (!want_bits) {
      } else if (type == 0) {
        for (uint32_t i = tid; i < min(card + 1, 4096u); i += VT) {
          const uint32_t v = (uint32_t)body[2 * i] | ((uint32_t)body[2 * i + 1] << 8);
          atomicOr(&s_dec[v >> 6], 1ull << (v & 63));
        }
      } else if (type == 1) {
        for (uint32_t w = tid; w < CHW; w += VT) {
          u64 v = 0;
          for (int b = 0; b < 8; ++b) v |= (u64)body[8 * w + b] << (8 * b);
          if (v) atomicOr(&s_dec[w], v);
        }
      } else {
        for (uint32_t rr = 0; rr < min(card + 1, 2048u); ++rr) {
          const uint32_t start = (uint32_t)body[4 * rr] | ((uint32_t)body[4 * rr + 1] << 8);
          const uint32_t rl = ((uint32_t)body[4 * rr + 2] | ((uint32_t)body[4 * rr + 3] << 8)) + 1;
          for (uint32_t i = tid; i < rl; i += VT) {
            const uint32_t v = start + i;
            if (v < 65536) atomicOr(&s_dec[v >> 6], 1ull << (v & 63));
          }
        }
      }
      __syncthreads();   // s_raw is reused by the next container
    }
    return n_here;
  };
  // ---- wide phase, wave path ---------------------------------------------------------------------------------------
  // A wide phase holds VM_DECODEC commands only (3 words each) and every one of them writes the SAME rank range [c_lo,
  // c_hi) of its destination.  When that range is at most 256 words (always, unless U0 is dense), each WAVE takes
  // commands of its own — its quarter of s_raw as the output words — and runs them without a workgroup barrier: four
  // decodes in flight per workgroup, and a decode is a handful of dependent loads.
  bool wide_done = false;
  // ---- decode phase BY RANK (round 6) ----------------------------------------------------------------------------------
  // The wide phase costs a workgroup per chunk of the FULL space — 153 of them at 10 M documents, ~25 us each — whatever
  // |U0| is, and it reads the postings from THEIR side (every value of an array container is looked up in U0): 36 % of the
  // keyword leg's workgroup time (MSI_VM_PROFILE, profiles/r6_vm_profile.log), most of it for universes of a few hundred
  // or thousand documents.  For those the host asks for this phase instead: one workgroup per 64 documents of U0, lane =
  // document (rank -> docid through the compaction table), wave w takes the commands w, w + 4, ...: the posting's container
  // of the document's chunk is probed for the ONE value (binary search in an array / among the runs, a bit test in a
  // bitmap), the wave's ballot IS the destination's 64-bit word.  No U0 words, no prefix counts, no atomics.
  if (by_rank) {
    wide_done = true;
    const uint32_t total = (uint32_t)r.n_docs;
    const uint32_t rk = chunk * 64 + lane;
    const bool live = rk < total;
    const uint32_t *c2d = reinterpret_cast<const uint32_t *>(rp->aux) + ((rp->wide_chunks + 3) & ~3u) + ((rp->full_words + 3) & ~3u);
    const uint32_t d = live ? c2d[rk] : 0u;
    const uint32_t lo16 = d & 0xFFFFu;
    const uint32_t *blk = nullptr;
    if (live && r.n_decodes) {
      const uint32_t *data = arena + r.list_off + r.data_off;
      blk = data + 4 * (size_t)data[d >> 16];
    }
    const uint32_t n_cmd = (p_end - p_begin) / 3;
    const uint32_t last_word = (uint32_t)(r.n_words - 1);
    const bool fills = (rp->wide_mask & 0x20000000u) != 0;   // a decode of this list is the first reader of its posting
    for (uint32_t k = wave; k < n_cmd; k += VT / 64) {
      if (MSI_UNIFORM(cmd[3 * k]) != VM_DECODEC) break;
      const uint32_t dsts = MSI_UNIFORM(cmd[3 * k + 1]), srcw = MSI_UNIFORM(cmd[3 * k + 2]);
      // the FIRST reader of a posting also stores its bodies into the posting cache (the host commits the entries when the
      // list has run): the wide phase does that per chunk; here the list's workgroups share the chunks of the full space
      if (fills && !(srcw >> 31) && r.n_decodes) {
        const uint32_t *data = arena + r.list_off + r.data_off;
        for (uint32_t fc = chunk; fc < rp->wide_chunks; fc += my_chunks) {
          const uint32_t *fb = data + 4 * (size_t)MSI_UNIFORM(data[fc]);
          const uint32_t f_first = MSI_UNIFORM(fb[srcw]), f_n = MSI_UNIFORM(fb[srcw + 1]) - f_first;
          for (uint32_t ci = 0; ci < f_n; ++ci) {
            const VmContainer c = *reinterpret_cast<const VmContainer *>(fb + desc_st + 4 * (size_t)(f_first + ci));
            if (c.fill_lo == 0xFFFFFFFFu && (c.meta >> 19) == 0x1FFFu) continue;
            const uint32_t card = c.meta & 0xFFFFu, type = (c.meta >> 16) & 3u;
            const u64 fill_off = ((u64)(c.meta >> 19) << 32) | c.fill_lo;
            const uint32_t len = type == 0 ? 2 * (card + 1) : (type == 1 ? 8192u : 4 * (card + 1));
            const uintptr_t b0 = ((c.meta >> 18) & 1u) ? (uintptr_t)(r.cache + c.src) : (uintptr_t)(r.stage + c.src);
            const uint32_t skew = (uint32_t)(b0 & 15);
            const uint4 *src = reinterpret_cast<const uint4 *>(b0 - skew);
            uint4 *fill = reinterpret_cast<uint4 *>((uintptr_t)(r.cache + fill_off) - skew);
            const uint32_t n16 = (skew + min(len, 8192u) + 15) / 16;
            for (uint32_t i = lane; i < n16; i += 64) put4(&fill[i], src[i]);
          }
        }
      }
      bool hit = false;
      if (live) {
        if (srcw >> 31) {
          const u64 *src_slot = reinterpret_cast<const u64 *>(rp->full_base) + (u64)(srcw & 0x7FFFFFFFu) * rp->full_words;
          hit = ((src_slot[d >> 6] >> (d & 63)) & 1ull) != 0;
        } else if (blk) {
          const uint32_t c_first = blk[srcw], n_here = blk[srcw + 1] - c_first;
          for (uint32_t ci = 0; ci < n_here && !hit; ++ci) {
            const VmContainer c = *reinterpret_cast<const VmContainer *>(blk + desc_st + 4 * (size_t)(c_first + ci));
            const uint32_t card = c.meta & 0xFFFFu, type = (c.meta >> 16) & 3u;
            const uintptr_t b0 = ((c.meta >> 18) & 1u) ? (uintptr_t)(r.cache + c.src) : (uintptr_t)(r.stage + c.src);
            if (type == 1) {
              const uint8_t byte = reinterpret_cast<const uint8_t *>(b0)[lo16 >> 3];
              hit = ((byte >> (lo16 & 7)) & 1u) != 0;
            } else if (type == 0) {
              uint32_t lo = 0, hi = min(card + 1, 4096u);   // the value is in [lo, hi) if anywhere
              while (lo < hi) {
                const uint32_t mid = (lo + hi) >> 1, v = ld16(b0, mid);
                if (v == lo16) {
                  hit = true;
                  break;
                }
                if (v < lo16) lo = mid + 1;
                else hi = mid;
              }
            } else {
              uint32_t lo = 0, hi = min(card + 1, 2048u);   // the last run that starts at or before the value
              while (lo < hi) {
                const uint32_t mid = (lo + hi) >> 1;
                if (ld16(b0, 2 * mid) <= lo16) lo = mid + 1;
                else hi = mid;
              }
              if (lo > 0) {
                const uint32_t start = ld16(b0, 2 * (lo - 1)), last = min(65535u, start + ld16(b0, 2 * (lo - 1) + 1));
                hit = lo16 <= last;
              }
            }
          }
        }
      }
      const u64 word = __ballot(hit ? 1 : 0);
      u64 *dst = pool + (u64)dsts * r.n_words;
      if (lane == 0 && chunk <= last_word) put1(&dst[chunk], word);
      // the workgroup of U0's last documents also clears what lies behind them in the slot
      if (chunk + 1 == my_chunks)
        for (u64 gw = (u64)chunk + 1 + lane; gw < r.n_words; gw += 64) put1(&dst[gw], 0ull);
    }
  }
  if (!by_rank && wide && (c_hi == c_lo || ((c_hi - 1) >> 6) - (c_lo >> 6) + 1 <= 256)) {
    wide_done = true;
    const uint32_t full_words = rp->full_words;
    const u64 fw0 = (u64)chunk * CHW;
    const uint32_t nwf = (uint32_t)min((u64)CHW, (u64)full_words - fw0);
    const uint32_t lo = c_lo, hi = c_hi, total = (uint32_t)r.n_docs;
    const uint32_t fwd = lo >> 6, n_out = hi > lo ? ((hi - 1) >> 6) - fwd + 1 : 0;
    u64 *w_out = reinterpret_cast<u64 *>(s_raw) + wave * 256;
    const uint32_t n_cmd = (p_end - p_begin) / 3;   // 3 words per command, then VM_END
    for (uint32_t k = wave; k < n_cmd; k += VT / 64) {
      if (MSI_UNIFORM(cmd[3 * k]) != VM_DECODEC) break;   // (cannot happen: the host records nothing else into a wide phase)
      const uint32_t dsts = MSI_UNIFORM(cmd[3 * k + 1]), srcw = MSI_UNIFORM(cmd[3 * k + 2]);
      const bool from_slot = (srcw >> 31) != 0;
      uint32_t n_here = 0, c_first = 0;
      if (!from_slot) {
        c_first = DW(srcw);
        n_here = DW(srcw + 1) - c_first;
        for (uint32_t ci = 0; ci < n_here; ++ci) {   // first reader of a key: its bodies go into the posting cache
          const VmContainer c = DC(c_first + ci);
          if (c.fill_lo == 0xFFFFFFFFu && (c.meta >> 19) == 0x1FFFu) continue;
          const uint32_t card = c.meta & 0xFFFFu, type = (c.meta >> 16) & 3u;
          const u64 fill_off = ((u64)(c.meta >> 19) << 32) | c.fill_lo;
          const uint32_t len = type == 0 ? 2 * (card + 1) : (type == 1 ? 8192u : 4 * (card + 1));
          const uintptr_t b0 = ((c.meta >> 18) & 1u) ? (uintptr_t)(r.cache + c.src) : (uintptr_t)(r.stage + c.src);
          const uint32_t skew = (uint32_t)(b0 & 15);
          const uint4 *src = reinterpret_cast<const uint4 *>(b0 - skew);
          uint4 *fill = reinterpret_cast<uint4 *>((uintptr_t)(r.cache + fill_off) - skew);
          const uint32_t n16 = (skew + min(len, 8192u) + 15) / 16;
          for (uint32_t i = lane; i < n16; i += 64) put4(&fill[i], src[i]);
        }
      }
      if (!n_out) continue;
      __builtin_amdgcn_wave_barrier();
      for (uint32_t i = lane; i < n_out; i += 64) w_out[i] = 0;
      __builtin_amdgcn_wave_barrier();
      auto rank_bits = [&](uint32_t wi, u64 m) {
        const u64 uw = s_dec[wi];
        const uint32_t base = s_cnt[wi] - (fwd << 6);
        while (m) {
          const uint32_t b = (uint32_t)__ffsll((long long)m) - 1;
          const uint32_t rk = base + (uint32_t)__popcll(uw & ((1ull << b) - 1ull));
          atomicOr(&w_out[rk >> 6], 1ull << (rk & 63));
          m &= m - 1;
        }
      };
      if (from_slot) {
        const u64 *src_slot = reinterpret_cast<const u64 *>(rp->full_base) + (u64)(srcw & 0x7FFFFFFFu) * full_words + fw0;
        for (uint32_t wi = lane; wi < nwf; wi += 64) {
          const u64 uw = s_dec[wi];
          if (!uw) continue;
          const u64 m = src_slot[wi] & uw;
          if (m) rank_bits(wi, m);
        }
      } else {
        for (uint32_t ci = 0; ci < n_here; ++ci) {
          const VmContainer c = DC(c_first + ci);
          const uint32_t card = c.meta & 0xFFFFu, type = (c.meta >> 16) & 3u;
          const uintptr_t b0 = ((c.meta >> 18) & 1u) ? (uintptr_t)(r.cache + c.src) : (uintptr_t)(r.stage + c.src);
          if (type == 0) {   // array: four loads of the wave in flight (512 bytes), then the look-ups
            const uint32_t n = min(card + 1, 4096u);
            for (uint32_t i0 = 0; i0 < n; i0 += 256) {
              uint32_t v[4];
#pragma unroll
              for (uint32_t u = 0; u < 4; ++u) {
                const uint32_t i = i0 + u * 64 + lane;
                v[u] = i < n ? ld16(b0, i) : 0xFFFFFFFFu;
              }
#pragma unroll
              for (uint32_t u = 0; u < 4; ++u)
                if (v[u] != 0xFFFFFFFFu && ((s_dec[v[u] >> 6] >> (v[u] & 63)) & 1ull)) rank_bits(v[u] >> 6, 1ull << (v[u] & 63));
            }
          } else if (type == 1) {
            const bool al8 = (b0 & 7) == 0;
            for (uint32_t wi = lane; wi < CHW; wi += 64) {   // bitmap: only the words where U0 has a document
              const u64 uw = s_dec[wi];
              if (!uw) continue;
              u64 v;
              if (al8) v = reinterpret_cast<const u64 *>(b0)[wi];
              else v = (u64)ld16(b0, 4 * wi) | ((u64)ld16(b0, 4 * wi + 1) << 16) | ((u64)ld16(b0, 4 * wi + 2) << 32) | ((u64)ld16(b0, 4 * wi + 3) << 48);
              const u64 m = v & uw;
              if (m) rank_bits(wi, m);
            }
          } else {
            const uint32_t n_runs = min(card + 1, 2048u);
            for (uint32_t rr = 0; rr < n_runs; ++rr) {
              const uint32_t start = ld16(b0, 2 * rr), last = min(65535u, start + ld16(b0, 2 * rr + 1));
              for (uint32_t wi = (start >> 6) + lane; wi <= (last >> 6); wi += 64) {
                const u64 uw = s_dec[wi];
                if (!uw) continue;
                u64 mask = ~0ull;
                if (wi == (start >> 6)) mask &= ~0ull << (start & 63);
                if (wi == (last >> 6)) mask &= (last & 63) == 63 ? ~0ull : ((1ull << ((last & 63) + 1)) - 1ull);
                const u64 m = uw & mask;
                if (m) rank_bits(wi, m);
              }
            }
          }
        }
      }
      __builtin_amdgcn_wave_barrier();
      u64 *dst = pool + (u64)dsts * r.n_words;
      const bool tail = hi == total;
      for (uint32_t i = lane; i < n_out; i += 64) {
        const u64 gw = (u64)fwd + i;
        const uint32_t lb = lo > gw * 64 ? (uint32_t)(lo - gw * 64) : 0u;
        const uint32_t hb = hi < (gw + 1) * 64 ? (uint32_t)(hi - gw * 64) : 64u;
        u64 own = (hb == 64 ? ~0ull : ((1ull << hb) - 1ull)) & ~((1ull << lb) - 1ull);
        if (tail && i == n_out - 1) own |= hb == 64 ? 0ull : ~((1ull << hb) - 1ull);
        const u64 val = w_out[i];
        if (own == ~0ull) {
          put1(&dst[gw], val);
        } else {
          atomicAnd(&dst[gw], ~own);
          if (val) atomicOr(&dst[gw], val);
        }
      }
      if (tail)
        for (u64 gw = (u64)fwd + n_out + lane; gw < r.n_words; gw += 64) put1(&dst[gw], 0ull);
    }
  }
  for (; !wide_done && !abandoned;) {
    const uint32_t op = W(0);
    if (op == VM_END) break;
    if (prof) {
      uint32_t scope = 0xFFFFFFFFu;
      if (op == VM_PATHS) scope = W(3);
      else if (op == VM_AND_MANY || op == VM_CLAIM || op == VM_SUB_MANY) scope = W(1);
      else if (op == VM_OP || op == VM_OP_COUNT) scope = W(3);
      else if (op == VM_COUNT || op == VM_FIRSTK) scope = W(1);
      if (scope != 0xFFFFFFFFu) {
        const bool e = chunk_empty(scope);
        if (tid == 0) {
          atomicAdd(&prof[20], 1ull);
          if (e) atomicAdd(&prof[21], 1ull);
        }
      }
    }
    const u64 t_op = prof ? wall_clock64() : 0;
#ifdef MSI_VM_DEBUG_CACHE
    { const u64 bb = __ballot(1); if (bb != ~0ull && rp->aux) printf("[sync] chunk %u tid %u before cmd %u op %u ballot %llx\n", chunk, tid, cmd_no + 1, op, bb); }
#endif
    ++cmd_no;
    MSI_EMU_TAG(op + (rp->aux ? 32u : 0u));   // (compact lists: + 32)
    switch (op) {
      case VM_FILL: {
        const Ref d = R(W(1));
        const bool ones = W(2) != 0;
        if (!ones) {
          make_empty(W(1));
          pcw += 3;
          break;
        }
        whole(W(1));
        nz(W(1), true);   // (a chunk past n_docs holds no document: one harmless "maybe")
        for (uint32_t p = tid; p < n_pairs; p += VT) {
          ulonglong2 v = make_ulonglong2(0, 0);
          if (ones) {
            v.x = doc_mask(w0 + 2 * p, r.n_docs);
            v.y = doc_mask(w0 + 2 * p + 1, r.n_docs);
          }
          ST(d, p, v);
        }
        pcw += 3;
        break;
      }
      case VM_OP:
      case VM_OP_COUNT: {
        const Ref d = R(W(1)), a = R(W(2)), b = R(W(3));
        const uint32_t o = W(4), sd = W(1), sa = W(2), sb = W(3);
        uint32_t c = 0;
        const bool ea = E(sa), eb = E(sb);
        const bool res_empty = o == MSI_BITS_AND ? (ea || eb) : (o == MSI_BITS_ANDNOT ? ea : (ea && eb));
        if (res_empty) {
          make_empty(sd);
        } else if (op == VM_OP && sd == sa && eb && (o == MSI_BITS_OR || o == MSI_BITS_ANDNOT || o == MSI_BITS_XOR)) {
          // a |= nothing, a &= ~nothing: unchanged
        } else if (op == VM_OP && sd == sb && ea && (o == MSI_BITS_OR || o == MSI_BITS_XOR)) {
          // b |= nothing
        } else {
          whole(sd);
          u64 any = 0;
          for (uint32_t p = tid; p < n_pairs; p += VT) {
            const ulonglong2 v = apply_op(o, LD(a, p), LD(b, p));
            ST(d, p, v);
            any |= v.x | v.y;
            c += __popcll(v.x) + __popcll(v.y);
          }
          nz(sd, any != 0);
        }
        if (op == VM_OP_COUNT) {
          add_count(W(5), c);
          pcw += 6;
        } else {
          pcw += 5;
        }
        break;
      }
      case VM_CLEAR: {
        const uint32_t n = W(1);
        for (uint32_t k = 0; k < n; ++k) make_empty(W(2 + k));
        pcw += 2 + n;
        break;
      }
      case VM_CLAIM: {  // bucket |= docs; universe &= ~docs; stack[i] &= ~docs   (docs may be one of the stack slots)
        const Ref docs = R(W(1)), bucket = R(W(2)), uni = R(W(3));
        const uint32_t n = W(4);
        if (E(W(1))) {   // nothing to claim in this chunk
          pcw += 5 + n;
          break;
        }
        partial(W(2));   // (the universe and the stack only lose documents: their bits stay)
        // (a thread owns at most CHW / 2 / VT = 2 pairs of a chunk; the claimed documents stay in registers — `docs` may be
        // one of the stack slots and is emptied on the way — and every set is looked up once, by the whole wave)
        constexpr uint32_t PPT = CHW / 2 / VT;
        ulonglong2 dd[PPT];
#pragma unroll
        for (uint32_t j = 0; j < PPT; ++j) {
          const uint32_t p = tid + j * VT;
          dd[j] = p < n_pairs ? LD(docs, p) : make_ulonglong2(0, 0);
          if (!(dd[j].x | dd[j].y)) continue;
          ulonglong2 b = LD(bucket, p), u = LD(uni, p);
          b.x |= dd[j].x; b.y |= dd[j].y;
          u.x &= ~dd[j].x; u.y &= ~dd[j].y;
          ST(bucket, p, b);
          ST(uni, p, u);
        }
        for (uint32_t k = 0; k < n; ++k) {
          const Ref sk = R(W(5 + k));
#pragma unroll
          for (uint32_t j = 0; j < PPT; ++j) {
            const uint32_t p = tid + j * VT;
            if (!(dd[j].x | dd[j].y)) continue;
            ulonglong2 s = LD(sk, p);
            s.x &= ~dd[j].x; s.y &= ~dd[j].y;
            ST(sk, p, s);
          }
        }
        pcw += 5 + n;
        break;
      }
      case VM_AND_MANY: {  // dst[i] = prefix & cond[i], counts[base + i] = |dst[i]|
        const Ref pre = R(W(1));
        const uint32_t n = W(2), base = W(3);
        const bool epre = E(W(1));
        for (uint32_t k = 0; k < n; ++k) {
          const uint32_t sc = W(4 + 2 * k), sd = W(5 + 2 * k);
          if (epre || E(sc)) {
            make_empty(sd);
            continue;
          }
          const Ref cnd = R(sc), d = R(sd);
          uint32_t c = 0;
          whole(sd);
          u64 any = 0;
          for (uint32_t p = tid; p < n_pairs; p += VT) {
            const ulonglong2 x = LD(pre, p), y = LD(cnd, p);
            ulonglong2 v;
            v.x = x.x & y.x; v.y = x.y & y.y;
            ST(d, p, v);
            any |= v.x | v.y;
            c += __popcll(v.x) + __popcll(v.y);
          }
          nz(sd, any != 0);
          add_count(base + k, c);
        }
        pcw += 4 + 2 * n;
        break;
      }
      case VM_PATHS: {  // the paths of one cost level in DFS order: a path claims what the earlier paths left
        const uint32_t n_paths = W(1);
        ulonglong2 *bucket = S(W(2)), *uni = S(W(3));
        const uint32_t base = W(4), n_steps = W(5) & 0x7FFFFFFFu;
        const bool fresh = (W(5) >> 31) != 0;   // the bucket's previous content is garbage: this level writes it whole
        const uint32_t *off = cmd + pcw + 6, *steps = off + n_paths + 1;   // read inside the per-pair loop: plain (LDS) loads
        if (cache_on) {
          // ---- the level out of LDS (compact lists) ------------------------------------------------------------------
          // 1. residency: the universe, the bucket and every condition set of the level get a cache entry (what is cached
          //    already — the conditions of the level before — is only marked as in use); the fills go out together;
          // 2. every step is resolved to its entry once (c_so);
          // 3. the thread streams the level's steps eight at a time — codes, then eight LDS reads in flight — and folds
          //    them into the current path's AND; at a path's end the path claims what the earlier paths left.
          // One L2 round trip per level (none when everything is resident) instead of one per group of four paths.
          const bool mine = tid < n_pairs;
          uint32_t n_fill = 0;
#ifdef MSI_VM_DEBUG_CACHE
          { const u64 bb = __ballot(1); if (lane == 0 && bb != ~0ull) printf("[sync A] wave %u cmd %u ballot %llx\n", wave, cmd_no, bb); }
#endif
          c_plan(W(3), n_fill);
          c_plan(W(2), n_fill);
          for (uint32_t s0 = 0; s0 < n_steps; s0 += 64) {   // mark what is resident, so that planning evicts none of it
            const uint32_t sidx = s0 + lane;
            __builtin_amdgcn_wave_barrier();
            const uint32_t code = sidx < n_steps ? (uint32_t)c_map[steps[sidx]] : 0u;
            uint32_t mask = code ? 1u << (code - 1) : 0u;
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) mask |= (uint32_t)__shfl_xor((int)mask, o);
            if (lane < 32 && ((mask >> lane) & 1u)) v_estamp = cmd_no;
          }
          for (uint32_t s0 = 0; s0 < n_steps; s0 += 64) {
            const uint32_t sidx = s0 + lane;
            const uint32_t sl = sidx < n_steps ? steps[sidx] : 0u;
            __builtin_amdgcn_wave_barrier();
            u64 miss = __ballot(sidx < n_steps && c_map[sl] == 0);
            while (miss) {
              const uint32_t bl = (uint32_t)__ffsll((long long)miss) - 1;
              miss &= miss - 1;
              c_plan((uint32_t)__shfl((int)sl, (int)bl), n_fill);   // (looks the map up again: a set planned a moment ago is there)
            }
          }
          c_fill(n_fill);
#ifdef MSI_VM_DEBUG_CACHE
          { const u64 bb = __ballot(1); if (lane == 0 && bb != ~0ull) printf("[sync B] wave %u cmd %u ballot %llx\n", wave, cmd_no, bb); }
#endif
          __builtin_amdgcn_wave_barrier();
          for (uint32_t si = lane; si < min(n_steps, SO_CAP); si += 64) c_so[si] = c_map[steps[si]];
          __builtin_amdgcn_wave_barrier();
          const Ref ru = R(W(3)), rb = R(W(2));
          ulonglong2 u = make_ulonglong2(0, 0), b = make_ulonglong2(0, 0);
          if (mine) {
            u = LD(ru, tid);
            if (!fresh) b = LD(rb, tid);
          }
          uint32_t k = 0, next_end = n_paths ? MSI_UNIFORM(off[1]) : 0xFFFFFFFFu;
          ulonglong2 acc = make_ulonglong2(~0ull, ~0ull);
          auto path_ends = [&]() {   // path k claims universe & AND(its conditions)
            ulonglong2 m;
            m.x = u.x & acc.x; m.y = u.y & acc.y;
            if (m.x | m.y) {
              b.x |= m.x; b.y |= m.y;
              u.x &= ~m.x; u.y &= ~m.y;
              atomicAdd(&s_cnt[base + k], (uint32_t)(__popcll(m.x) + __popcll(m.y)));
            }
            ++k;
            acc = make_ulonglong2(~0ull, ~0ull);
            next_end = k < n_paths ? MSI_UNIFORM(off[k + 1]) : 0xFFFFFFFFu;
          };
          bool left = true;
          for (uint32_t s0 = 0; s0 < n_steps && left; s0 += 8) {
            uint32_t code[8];
            if (s0 + 8 <= SO_CAP) {
              __builtin_amdgcn_wave_barrier();
              const uint32_t c0 = MSI_UNIFORM(reinterpret_cast<const uint32_t *>(c_so + s0)[0]);
              const uint32_t c1 = MSI_UNIFORM(reinterpret_cast<const uint32_t *>(c_so + s0)[1]);
#pragma unroll
              for (uint32_t i = 0; i < 4; ++i) {
                code[i] = (c0 >> (8 * i)) & 0xFFu;
                code[4 + i] = (c1 >> (8 * i)) & 0xFFu;
              }
            } else {
#pragma unroll
              for (uint32_t i = 0; i < 8; ++i) code[i] = s0 + i < n_steps ? MSI_UNIFORM((uint32_t)c_map[MSI_UNIFORM(steps[s0 + i])]) : 0u;
            }
            ulonglong2 d[8];
#pragma unroll
            for (uint32_t i = 0; i < 8; ++i) {
              d[i] = make_ulonglong2(0, 0);
              if (s0 + i < n_steps) {
                if (code[i]) d[i] = c_data[(size_t)(code[i] - 1) * ent_pairs + tid];
                else if (mine) d[i] = S(MSI_UNIFORM(steps[s0 + i]))[tid];
#ifdef MSI_VM_DEBUG_CACHE
                if (mine) {
                  const ulonglong2 g = S(steps[s0 + i])[tid];
                  if (g.x != d[i].x || g.y != d[i].y)
                    printf("[cache] chunk %u cmd %u step %u slot %u code %u: lds %llx %llx mem %llx %llx\n", chunk, cmd_no, s0 + i, steps[s0 + i], code[i], d[i].x, d[i].y, g.x, g.y);
                }
#endif
              }
            }
#pragma unroll
            for (uint32_t i = 0; i < 8; ++i)
              if (s0 + i < n_steps) {
                while (s0 + i == next_end) path_ends();
                acc.x &= d[i].x; acc.y &= d[i].y;
              }
            left = __any((u.x | u.y) != 0);   // nothing left to claim in this wave's pairs: the rest of the level finds nothing
          }
          while (left && k < n_paths) path_ends();
#ifdef MSI_VM_DEBUG_CACHE
          if (mine) {
            ulonglong2 u2 = ru.g[tid], b2 = fresh ? make_ulonglong2(0, 0) : rb.g[tid];
            for (uint32_t kk = 0; kk < n_paths; ++kk) {
              ulonglong2 a2 = make_ulonglong2(~0ull, ~0ull);
              for (uint32_t ss = off[kk]; ss < off[kk + 1]; ++ss) { a2.x &= S(steps[ss])[tid].x; a2.y &= S(steps[ss])[tid].y; }
              b2.x |= u2.x & a2.x; b2.y |= u2.y & a2.y;
              u2.x &= ~a2.x; u2.y &= ~a2.y;
            }
            if (u2.x != u.x || u2.y != u.y || b2.x != b.x || b2.y != b.y)
              printf("[cache] PATHS chunk %u cmd %u n_paths %u n_steps %u: u %llx/%llx b %llx/%llx\n", chunk, cmd_no, n_paths, n_steps, u.x, u2.x, b.x, b2.x);
          }
#endif
          if (mine) {
            ST(rb, tid, b);
            ST(ru, tid, u);
          }
          pcw += 7 + n_paths + n_steps;
          break;
        }
        // Paths are resolved four at a time: the condition words of the four paths are loaded back to back (no load
        // waits for the result of another), then the paths claim in order.  A serial "load, AND, test, next step" chain
        // made a level of a few hundred steps cost hundreds of microseconds of pure memory latency per workgroup.
        if (E(W(3))) {   // no document of the universe in this chunk: the level finds nothing here
          if (fresh) make_empty(W(2));
          pcw += 7 + n_paths + n_steps;
          break;
        }
        if (fresh) whole(W(2));
        else partial(W(2));
        // A path one of whose condition sets is empty in this chunk finds nothing here: it is neither loaded nor tested
        // (one summary bit per step, read by the lane that owns the path; a ballot makes the verdicts wave-uniform).
        u64 alive[(MSI_BITS_MAX_PATHS + 63) / 64];
#pragma unroll
        for (uint32_t g = 0; g < (MSI_BITS_MAX_PATHS + 63) / 64; ++g) alive[g] = ~0ull;
        if (sum_on) {
          __builtin_amdgcn_wave_barrier();
#pragma unroll
          for (uint32_t g = 0; g < (MSI_BITS_MAX_PATHS + 63) / 64; ++g) {
            const uint32_t k = g * 64 + lane;
            bool ok = false;
            if (k < n_paths) {
              ok = true;
              for (uint32_t s = off[k]; s < off[k + 1]; ++s) {
                const uint32_t sl = steps[s];
                ok = ok && ((s_sum[wave][sl >> 6] >> (sl & 63)) & 1ull) != 0;
              }
            }
            alive[g] = __ballot(ok);
          }
          __builtin_amdgcn_wave_barrier();
        }
        auto live = [&](uint32_t k) -> bool { return k < n_paths && (k >= MSI_BITS_MAX_PATHS || ((alive[k >> 6] >> (k & 63)) & 1ull)); };
        u64 any_b = 0;
        for (uint32_t p = tid; p < n_pairs; p += VT) {
          ulonglong2 u = uni[p];
          if (!(u.x | u.y)) {
            if (fresh) put(&bucket[p], make_ulonglong2(0, 0));
            continue;
          }
          ulonglong2 b = fresh ? make_ulonglong2(0, 0) : bucket[p];
          for (uint32_t k0 = 0; k0 < n_paths && (u.x | u.y); k0 += 4) {
            if (!(live(k0) || live(k0 + 1) || live(k0 + 2) || live(k0 + 3))) continue;
            ulonglong2 acc[4];
#pragma unroll
            for (uint32_t j = 0; j < 4; ++j) {
              acc[j] = make_ulonglong2(~0ull, ~0ull);
              if (live(k0 + j))
                for (uint32_t s = off[k0 + j]; s < off[k0 + j + 1]; ++s) {
                  const ulonglong2 c = S(steps[s])[p];
                  acc[j].x &= c.x; acc[j].y &= c.y;
                }
            }
#pragma unroll
            for (uint32_t j = 0; j < 4; ++j) {
              if (!live(k0 + j)) continue;
              ulonglong2 m;
              m.x = u.x & acc[j].x; m.y = u.y & acc[j].y;
              if (m.x | m.y) {
                b.x |= m.x; b.y |= m.y;
                u.x &= ~m.x; u.y &= ~m.y;
                atomicAdd(&s_cnt[base + k0 + j], (uint32_t)(__popcll(m.x) + __popcll(m.y)));
              }
            }
          }
          put(&bucket[p], b);
          any_b |= b.x | b.y;
          put(&uni[p], u);
        }
        if (fresh) nz(W(2), any_b != 0);
        pcw += 7 + n_paths + n_steps;
        break;
      }
      case VM_SUB_MANY: {  // slot[i] &= ~removed, counts[base + i] = |slot[i]|
        const Ref rm = R(W(1));
        const uint32_t n = W(2), base = W(3);
        const bool erm = E(W(1));
        for (uint32_t k = 0; k < n; ++k) {
          const uint32_t sd = W(4 + k);
          if (E(sd)) continue;                      // nothing to remove from, nothing to count
          const Ref d = R(sd);
          uint32_t c = 0;
          if (erm) {                                // nothing removed here: the cardinality is still asked for
            for (uint32_t p = tid; p < n_pairs; p += VT) {
              const ulonglong2 v = LD(d, p);
              c += __popcll(v.x) + __popcll(v.y);
            }
          } else {
            whole(sd);
            u64 any = 0;
            for (uint32_t p = tid; p < n_pairs; p += VT) {
              const ulonglong2 x = LD(rm, p);
              ulonglong2 v = LD(d, p);
              v.x &= ~x.x; v.y &= ~x.y;
              ST(d, p, v);
              any |= v.x | v.y;
              c += __popcll(v.x) + __popcll(v.y);
            }
            nz(sd, any != 0);
          }
          add_count(base + k, c);
        }
        pcw += 4 + n;
        break;
      }
      case VM_COUNT:
      case VM_FIRSTK: {
        const Ref a = R(W(1));
        uint32_t c = 0;
        if (!E(W(1)))
          for (uint32_t p = tid; p < n_pairs; p += VT) {
            const ulonglong2 v = LD(a, p);
            c += __popcll(v.x) + __popcll(v.y);
          }
        if (op == VM_COUNT) {
          add_count(W(2), c);
          pcw += 3;
        } else {          // slot, k, cnt, ids base: this chunk's cardinality is also kept for the ordered emit
          add_count(W(3), c);
          if (tid < 4 && n_fk < MSI_VM_MAX_FK_PHASE) s_fk[n_fk][tid] = cmd[pcw + 1 + tid];
          ++n_fk;
          pcw += 5;
        }
        break;
      }
      case VM_DECODE: {  // the containers of THIS chunk of every posting of the batch, OR-ed in LDS, written once
        ulonglong2 *d = S(W(1));
        const bool overwrite = W(2) != 0;
        const uint32_t n_here = decode_chunk(W(3), true);   // index of this decode among the list's decodes
        if (n_here) {
          if (overwrite) whole(W(1));
          else partial(W(1));
          u64 any_d = 0;
          for (uint32_t p = tid; p < n_pairs; p += VT) {
            ulonglong2 v;
            v.x = s_dec[2 * p] & doc_mask(w0 + 2 * p, r.n_docs);
            v.y = s_dec[2 * p + 1] & doc_mask(w0 + 2 * p + 1, r.n_docs);
            if (!overwrite) {
              const ulonglong2 o = d[p];
              v.x |= o.x; v.y |= o.y;
            }
            put(&d[p], v);
            any_d |= v.x | v.y;
          }
          if (overwrite) nz(W(1), any_d != 0);
        } else if (overwrite) {
          make_empty(W(1));
        }
        pcw += 4;
        break;
      }
      case VM_RANK_A: {  // universe compaction, first half (a list on the FULL pool): this chunk's cardinality of U0
        const ulonglong2 *a = S(W(1));
        uint32_t *aux = reinterpret_cast<uint32_t *>(((u64)W(3) << 32) | W(2));
        uint32_t c = 0;
        if (!E(W(1)))
          for (uint32_t p = tid; p < n_pairs; p += VT) {
            const ulonglong2 v = a[p];
            c += __popcll(v.x) + __popcll(v.y);
          }
        c = wave_sum(c);
        __syncthreads();
        if (lane == 0) s_scan[wave] = c;
        __syncthreads();
        if (tid == 0) {
          uint32_t t = 0;
          for (int i = 0; i < VT / 64; ++i) t += s_scan[i];
          __hip_atomic_store(&aux[chunk], t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        __syncthreads();
        pcw += 4;
        break;
      }
      case VM_RANK_B: {  // second half (next phase): exclusive prefix popcounts per word of U0, and rank -> docid
        __syncthreads();
        const u64 *a = pool + (u64)W(1) * r.n_words + w0;
        uint32_t *aux = reinterpret_cast<uint32_t *>(((u64)W(3) << 32) | W(2));
        const uint32_t cap = W(4);
        uint32_t *prefix = aux + ((r.n_chunks + 3) & ~3u);
        uint32_t *c2d = prefix + (((uint32_t)r.n_words + 3) & ~3u);
        uint32_t part = 0;
        for (uint32_t c = tid; c < chunk; c += VT) part += __hip_atomic_load(&aux[c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        part = wave_sum(part);
        if (lane == 0) s_scan[wave] = part;
        __syncthreads();
        uint32_t base = 0;
        for (int i = 0; i < VT / 64; ++i) base += s_scan[i];
        __syncthreads();
        u64 w[WPT];
        uint32_t mine = 0;
#pragma unroll
        for (int j = 0; j < WPT; ++j) {   // thread t owns words WPT*t .. WPT*t+WPT-1: ascending across threads
          const uint32_t wi = WPT * tid + j;
          w[j] = wi < nw ? a[wi] : 0ull;
          mine += (uint32_t)__popcll(w[j]);
        }
        uint32_t incl = mine;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
          const uint32_t v = __shfl_up((int)incl, o);
          if ((int)lane >= o) incl += v;
        }
        if (lane == 63) s_scan[wave] = incl;
        __syncthreads();
        uint32_t run = base + incl - mine;
        for (uint32_t ww = 0; ww < wave; ++ww) run += s_scan[ww];
#pragma unroll
        for (int j = 0; j < WPT; ++j) {
          const uint32_t wi = WPT * tid + j;
          if (wi < nw) __hip_atomic_store(&prefix[w0 + wi], run, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          u64 x = w[j];
          while (x) {
            const uint32_t b = (uint32_t)__ffsll((long long)x) - 1;
            if (run < cap) __hip_atomic_store(&c2d[run], (uint32_t)((w0 + wi) * 64 + b), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            ++run;
            x &= x - 1;
          }
        }
        __syncthreads();
        pcw += 5;
        break;
      }
      case VM_DECODEC: {  // compact list, wide phase: workgroup = chunk of the FULL space.  The documents of the source
                          // (a decode batch's containers of this chunk, or this chunk of a full-space slot) that are in
                          // U0 become ranks; this chunk's documents of U0 have the contiguous ranks [lo, hi), the part of
                          // dst this workgroup writes WHOLE (words shared with a neighbouring chunk: its own bits, atomically).
                          // U0's words and prefix counts of this chunk are in LDS (s_dec / s_cnt, staged at the phase's
                          // start); the containers are read straight from memory, a wave per container, no barrier between
                          // them: a value only has to be looked up in U0 and, when it is there, ranked.
        const uint32_t dsts = W(1), srcw = W(2);
        const uint32_t full_words = rp->full_words;
        const u64 fw0 = (u64)chunk * CHW;
        const uint32_t nwf = (uint32_t)min((u64)CHW, (u64)full_words - fw0);
        const uint32_t lo = c_lo, hi = c_hi, total = (uint32_t)r.n_docs;
        const bool from_slot = (srcw >> 31) != 0;
        uint32_t n_here = 0, c_first = 0;
        if (!from_slot) {
          c_first = DW(srcw);
          n_here = DW(srcw + 1) - c_first;
          // first reader of a key: its bodies go into the posting cache, whatever U0 holds in this chunk
          for (uint32_t ci = 0; ci < n_here; ++ci) {
            const VmContainer c = DC(c_first + ci);
            if (c.fill_lo == 0xFFFFFFFFu && (c.meta >> 19) == 0x1FFFu) continue;
            const uint32_t card = c.meta & 0xFFFFu, type = (c.meta >> 16) & 3u;
            const u64 fill_off = ((u64)(c.meta >> 19) << 32) | c.fill_lo;
            const uint32_t len = type == 0 ? 2 * (card + 1) : (type == 1 ? 8192u : 4 * (card + 1));
            const uintptr_t b0 = ((c.meta >> 18) & 1u) ? (uintptr_t)(r.cache + c.src) : (uintptr_t)(r.stage + c.src);
            const uint32_t skew = (uint32_t)(b0 & 15);
            const uint4 *src = reinterpret_cast<const uint4 *>(b0 - skew);
            uint4 *fill = reinterpret_cast<uint4 *>((uintptr_t)(r.cache + fill_off) - skew);
            const uint32_t n16 = (skew + min(len, 8192u) + 15) / 16;
            for (uint32_t i = tid; i < n16; i += VT) put4(&fill[i], src[i]);
          }
        }
        if (hi > lo) {
          u64 *s_out = reinterpret_cast<u64 *>(s_raw);          // [lo >> 6 .. (hi - 1) >> 6]: at most 1025 words
          const uint32_t fwd = lo >> 6, n_out = ((hi - 1) >> 6) - fwd + 1;
          __syncthreads();
          for (uint32_t i = tid; i < n_out; i += VT) s_out[i] = 0;
          __syncthreads();
          // the documents `m` (bits of word wi of this chunk, all of them in U0) -> their ranks
          auto rank_bits = [&](uint32_t wi, u64 m) {
            const u64 uw = s_dec[wi];
            const uint32_t base = s_cnt[wi] - (fwd << 6);
            while (m) {
              const uint32_t b = (uint32_t)__ffsll((long long)m) - 1;
              const uint32_t rk = base + (uint32_t)__popcll(uw & ((1ull << b) - 1ull));
              atomicOr(&s_out[rk >> 6], 1ull << (rk & 63));
              m &= m - 1;
            }
          };
          if (from_slot) {
            const u64 *src_slot = reinterpret_cast<const u64 *>(rp->full_base) + (u64)(srcw & 0x7FFFFFFFu) * full_words + fw0;
            for (uint32_t wi = tid; wi < nwf; wi += VT) {
              const u64 uw = s_dec[wi];
              if (!uw) continue;
              const u64 m = src_slot[wi] & uw;
              if (m) rank_bits(wi, m);
            }
          } else {
            for (uint32_t ci = wave; ci < n_here; ci += VT / 64) {   // a wave per container
              const VmContainer c = DC(c_first + ci);
              const uint32_t card = c.meta & 0xFFFFu, type = (c.meta >> 16) & 3u;
              const uintptr_t b0 = ((c.meta >> 18) & 1u) ? (uintptr_t)(r.cache + c.src) : (uintptr_t)(r.stage + c.src);
              if (type == 0) {          // array: the values themselves
                const uint32_t n = min(card + 1, 4096u);
                for (uint32_t i = lane; i < n; i += 64) {
                  const uint32_t v = ld16(b0, i);
                  if ((s_dec[v >> 6] >> (v & 63)) & 1ull) rank_bits(v >> 6, 1ull << (v & 63));
                }
              } else if (type == 1) {   // bitmap: only the words where U0 has a document are read
                const bool al8 = (b0 & 7) == 0;
                for (uint32_t wi = lane; wi < CHW; wi += 64) {
                  const u64 uw = s_dec[wi];
                  if (!uw) continue;
                  u64 v;
                  if (al8) v = reinterpret_cast<const u64 *>(b0)[wi];
                  else v = (u64)ld16(b0, 4 * wi) | ((u64)ld16(b0, 4 * wi + 1) << 16) | ((u64)ld16(b0, 4 * wi + 2) << 32) | ((u64)ld16(b0, 4 * wi + 3) << 48);
                  const u64 m = v & uw;
                  if (m) rank_bits(wi, m);
                }
              } else {                  // runs: (start, length - 1) pairs
                const uint32_t n_runs = min(card + 1, 2048u);
                for (uint32_t rr = 0; rr < n_runs; ++rr) {
                  const uint32_t start = ld16(b0, 2 * rr), last = min(65535u, start + ld16(b0, 2 * rr + 1));
                  for (uint32_t wi = (start >> 6) + lane; wi <= (last >> 6); wi += 64) {
                    const u64 uw = s_dec[wi];
                    if (!uw) continue;
                    u64 mask = ~0ull;
                    if (wi == (start >> 6)) mask &= ~0ull << (start & 63);
                    if (wi == (last >> 6)) mask &= (last & 63) == 63 ? ~0ull : ((1ull << ((last & 63) + 1)) - 1ull);
                    const u64 m = uw & mask;
                    if (m) rank_bits(wi, m);
                  }
                }
              }
            }
          }
          __syncthreads();
          u64 *dst = pool + (u64)dsts * r.n_words;
          const bool tail = hi == total;   // the last range also owns what lies behind |U0|, up to the end of the slot
          for (uint32_t i = tid; i < n_out; i += VT) {
            const u64 gw = (u64)fwd + i;
            const uint32_t lb = lo > gw * 64 ? (uint32_t)(lo - gw * 64) : 0u;
            const uint32_t hb = hi < (gw + 1) * 64 ? (uint32_t)(hi - gw * 64) : 64u;   // 1 .. 64
            u64 own = (hb == 64 ? ~0ull : ((1ull << hb) - 1ull)) & ~((1ull << lb) - 1ull);
            if (tail && i == n_out - 1) own |= hb == 64 ? 0ull : ~((1ull << hb) - 1ull);
            const u64 val = s_out[i];
            if (own == ~0ull) {
              put1(&dst[gw], val);
            } else {
              atomicAnd(&dst[gw], ~own);
              if (val) atomicOr(&dst[gw], val);
            }
          }
          if (tail)
            for (u64 gw = (u64)fwd + n_out + tid; gw < r.n_words; gw += VT) put1(&dst[gw], 0ull);
          __syncthreads();   // s_out is reused by the next command
        }
        pcw += 3;
        break;
      }
      case VM_MINKEY: {  // Sort rule, first half: the smallest order key among the documents of the universe
        __syncthreads();  // one document per thread from here: other threads' set words must be visible
        const u64 *uni = pool + (u64)W(1) * r.n_words + w0;
        const uint32_t *keys = reinterpret_cast<const uint32_t *>(((u64)W(3) << 32) | W(2));
        uint32_t inv = 0;
        if (!E(W(1)))
          for (uint32_t w = wave; w < nw; w += VT / 64) {
            const u64 word = uni[w];
            if (!word) continue;
            const u64 doc = (w0 + w) * 64 + lane;
            if ((word >> lane) & 1ull) inv = max(inv, 0xFFFFFFFFu - keys[doc]);
          }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) inv = max(inv, (uint32_t)__shfl_xor((int)inv, o));
        if (lane == 0 && inv) atomicMax(&cells[W(4)], (u64)inv);
        pcw += 5;
        break;
      }
      case VM_TAKEKEY: {  // second half (next phase): bucket = the documents with that key, universe -= bucket
        __syncthreads();
        u64 *uni = pool + (u64)W(1) * r.n_words + w0;
        u64 *bucket = pool + (u64)W(2) * r.n_words + w0;
        const uint32_t *keys = reinterpret_cast<const uint32_t *>(((u64)W(4) << 32) | W(3));
        const uint32_t key = 0xFFFFFFFFu - (uint32_t)cells[W(5)];
        uint32_t c = 0;
        partial(W(2));   // (written word by word by lane 0 of each wave: no exact bit)
        for (uint32_t w = wave; w < nw; w += VT / 64) {
          const u64 word = uni[w];
          u64 mask = 0;
          if (word) {
            const u64 doc = (w0 + w) * 64 + lane;
            const bool hit = ((word >> lane) & 1ull) && keys[doc] == key;
            mask = __ballot(hit);
          }
          if (lane == 0) {
            put1(&bucket[w], mask);
            if (mask) put1(&uni[w], word & ~mask);
            c += (uint32_t)__popcll(mask);
          }
        }
        if (lane == 0 && c) atomicAdd(&s_cnt[W(6)], c);
        if (chunk == 0 && tid == 0) s_cnt[W(7)] = key;   // the key itself travels as a "count"
        __syncthreads();
        pcw += 8;
        break;
      }
      case VM_SUMMARY_RESET: {  // something outside the command lists wrote slots of this pool: every bit back to "may hold"
        __syncthreads();
        if (sum_on && lane < SUM_W) s_sum[wave][lane] = ~0ull;
        if (sum_on)
          for (uint32_t i = tid; i < SUM_W * 64; i += VT) s_whole[i] = 0;
        __syncthreads();
        pcw += 1;
        break;
      }
      default:
        pcw = 0xFFFFFFFFu;  // unknown opcode: stop (the host validates what it records)
        break;
    }
    if (pcw == 0xFFFFFFFFu) break;
    if (prof && tid == 0) s_prof[op == VM_DECODEC ? (uint32_t)VM_DECODE : (op >= 15 ? 14u : op)] += wall_clock64() - t_op;
  }
  if (prof && tid == 0) {
    for (int i = 1; i < 15; ++i)
      if (s_prof[i]) atomicAdd(&prof[i], s_prof[i]);
    atomicAdd(&prof[15], wall_clock64() - t_begin);   // the whole interpretation, commands + fetches
    if (wide) atomicAdd(&prof[19], wall_clock64() - t_begin);   // (of it: in wide workgroups)
    atomicAdd(&prof[0], 1ull);                        // workgroups
    atomicMax(&cells[MSI_VM_CELLS - 1], ~t_begin);    // earliest start of a workgroup of this list (phase 0 profile only)
  }
  const u64 t_epi = prof ? wall_clock64() : 0;

#undef W
  // ---- this workgroup's cardinalities leave LDS; the last workgroup of the list publishes --------------------
  __syncthreads();
  if (sum_on) {
    // slots whose chunk this list wrote whole: the bit is exact — empty unless the last such write stored a document
    for (uint32_t i = tid; i < SUM_W * 64; i += VT)
      if (s_whole[i] && s_nzw[i] != (uint32_t)s_whole[i]) atomicAnd(&s_sum[0][i >> 6], ~(1ull << (i & 63)));
    __syncthreads();
    if (tid < SUM_W) put1(&sum_row[(u64)chunk * SUM_W + tid], s_sum[0][tid]);
  }
  if (!wide)   // (a wide phase keeps U0's prefix counts in s_cnt and counts nothing)
    for (uint32_t i = tid; i < r.n_counts; i += VT)
      if (s_cnt[i]) atomicAdd(&counts[i], (u64)s_cnt[i]);
  // Everything this workgroup stored is write-through (put): the ticket below only has to wait until those stores
  // and the cardinalities' atomics are acknowledged — no L2 write-back, no invalidate (a __threadfence here cost both,
  // under every resident kernel, 153 times per list at 10 M documents: r2_ranked10_timeline_before.txt).
  n_fk = min(n_fk, MSI_VM_MAX_FK_PHASE);
  if (n_fk) {
    if (tid < n_fk)
      __hip_atomic_store(&chunk_card[tid * r.n_chunks + chunk], s_cnt[s_fk[tid][2]], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  MSI_ORDER_ATOMICS();
  __syncthreads();
  if (tid == 0) s_last = atomicAdd(&state[phase], 1u) == my_chunks - 1 ? 1u : 0u;
  __syncthreads();
  if (prof && tid == 0) atomicAdd(&prof[18], wall_clock64() - t_epi);   // counts out + ordering + ticket
  if (!s_last) return;
  if (prof && tid == 0) {
    const u64 first = ~__hip_atomic_load(&cells[MSI_VM_CELLS - 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    atomicAdd(&prof[16], wall_clock64() - first);   // first workgroup's start .. last workgroup's ticket
    atomicAdd(&prof[17], 1ull);
  }
  if (n_fk) __threadfence();   // the set words the other workgroups wrote (read below with device-scope loads)
  u64 *res = reinterpret_cast<u64 *>(r.host_res);
  uint32_t emitted = 0;
  // a compact list's sets hold ranks inside U0: the ids leave as docids (rank -> docid table of the full pool)
  const uint32_t *c2d = rp->aux ? reinterpret_cast<const uint32_t *>(rp->aux) + ((rp->wide_chunks + 3) & ~3u) + ((rp->full_words + 3) & ~3u) : nullptr;
  // the first-k commands of THIS phase are emitted by this phase's last workgroup (the ids are in the host's block
  // before the kernel of the list's last phase — a later launch on the same stream — publishes the sequence number)
  for (uint32_t f = 0; f < n_fk; ++f) {
    // ordered emit of the first k documents: chunk cardinalities are known, so only the chunks that contribute are read
    const uint32_t slot = s_fk[f][0], k = s_fk[f][1];
    const uint32_t *cc = chunk_card + f * r.n_chunks;
    uint32_t *ids = reinterpret_cast<uint32_t *>(res + RES_IDS) + s_fk[f][3];
    uint32_t running = 0;
    for (uint32_t c = 0; c < r.n_chunks && running < k; ++c) {
      const uint32_t n_c = __hip_atomic_load(&cc[c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (!n_c) continue;
      const u64 cw0 = (u64)c * chw;
      const uint32_t cnw = (uint32_t)min((u64)chw, r.n_words - cw0);
      const u64 *a = pool + (u64)slot * r.n_words + cw0;
      u64 w[WPT];
      uint32_t mine = 0;
#pragma unroll
      for (int j = 0; j < WPT; ++j) {   // thread t owns words WPT*t .. WPT*t+WPT-1: ascending across threads
        const uint32_t wi = WPT * tid + j;
        w[j] = wi < cnw ? __hip_atomic_load(&a[wi], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0ull;
        mine += (uint32_t)__popcll(w[j]);
      }
      uint32_t incl = mine;   // inclusive scan inside the wave, then across the 4 waves
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) {
        const uint32_t v = __shfl_up((int)incl, o);
        if ((int)lane >= o) incl += v;
      }
      if (lane == 63) s_scan[wave] = incl;
      __syncthreads();
      uint32_t before = running;
      for (uint32_t ww = 0; ww < wave; ++ww) before += s_scan[ww];
      uint32_t rank = before + incl - mine;
#pragma unroll
      for (int j = 0; j < WPT; ++j) {
        u64 x = w[j];
        while (x && rank < k) {
          const uint32_t b = (uint32_t)__ffsll((long long)x) - 1;
          const uint32_t id = (uint32_t)((cw0 + WPT * tid + j) * 64 + b);
          __hip_atomic_store(&ids[rank++], c2d ? c2d[id] : id, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
          x &= x - 1;
        }
      }
      __syncthreads();
      running += n_c;
    }
    emitted += min(running, k);
  }
  if (phase != r.n_phases - 1) return;
  for (uint32_t i = tid; i < r.n_counts; i += VT) {
    const u64 v = __hip_atomic_load(&counts[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(&res[RES_COUNTS + i], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
  // the result block is pinned host memory (uncached on the device): every store above goes straight out; once they
  // are acknowledged the sequence number follows them on the same path
  MSI_ORDER_ATOMICS();
  __syncthreads();
  if (tid == 0) {
    const bool gave_up = __hip_atomic_load(&rp->failed, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0;
    __hip_atomic_store(&res[1], gave_up ? MSI_VM_RES_FAILED : (u64)emitted, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    MSI_ORDER_ATOMICS();
    __hip_atomic_store(&res[0], r.seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

}  // namespace

// ================================================================================================ combiner

// Diagnostic counters that every search thread bumps for every list: striped by thread, summed by the reader — one
// shared cache line per counter took a million contended read-modify-writes per second at 128 callers.
template <int N>
struct StripedCounters {
  static constexpr int STRIPES = 32;
  struct alignas(64) Cell { std::atomic<uint64_t> v[N]; };
  Cell cells[STRIPES];
  StripedCounters() { for (auto &c : cells) for (auto &x : c.v) x.store(0, std::memory_order_relaxed); }
  static unsigned stripe() {
    static std::atomic<unsigned> next{0};
    thread_local const unsigned mine = next.fetch_add(1, std::memory_order_relaxed) % STRIPES;
    return mine;
  }
  void add(int i, uint64_t x) { cells[stripe()].v[i].fetch_add(x, std::memory_order_relaxed); }
  uint64_t sum(int i) const {
    uint64_t t = 0;
    for (const auto &c : cells) t += c.v[i].load(std::memory_order_relaxed);
    return t;
  }
  void reset() { for (auto &c : cells) for (auto &x : c.v) x.store(0, std::memory_order_relaxed); }
};

// A submitted list.  Lives in a slot OWNED BY THE VM (recycled, never freed before the VM), so the combiner may touch it
// at any time; the search thread sleeps on `state` (futex) and is woken by the combiner, which is the only thread that
// polls the GPU's completion words.  (Search threads that spin burn the CPU budget the searches themselves need: on a
// 16-CPU container quota, 64 spinning waiters got the whole process throttled — r2_ranked_2m_vm_v1*.jsonl.)
struct VmSub {
  msi_bits *pool = nullptr;
  const MsiVmList *list = nullptr;
  uint64_t seq = 0;
  volatile uint64_t *blk = nullptr;        // the pool's pinned result block; blk[0] == seq when the list has run
  std::atomic<uint32_t> state{0};          // 0 waiting, 1 done, 2 failed
  uint32_t fused_wgs = 0;                  // waiting workgroups this list holds of the device's budget (guard of `fused`)
  std::atomic<uint32_t> asleep{0};         // the waiter is (about to be) blocked in futex_wait: the combiner must wake it
  int32_t error = MSI_OK;
  char errmsg[192] = "";                   // the combiner thread's error text (msi_last_error is thread-local: the waiter re-issues it)
  int64_t t_submit = 0, t_taken = 0, t_launch = 0, t_done = 0;   // steady-clock ns (diagnostics)
  VmSub *next = nullptr;                   // the combiner's submission stack (VmCombiner::q_head)
};

struct VmCombiner {
  msi_ctx *ctx = nullptr;
  static constexpr int NS = 32;     // rounds in flight AT MOST: a slow list of one round must not hold up the next round's lists
  int ns = 16;                      // ... of which this many are used (MSI_VM_STREAMS: even, 4..NS; half per class of rounds)
  hipStream_t streams[NS] = {};
  std::thread th;
  // Submission takes no lock (round 6).  Every list used to take `mu` twice — to queue itself and to hand its slot back —
  // and the combiner took it once per loop turn: with 256 searches in flight on a 16-CPU quota, pthread_mutex_lock / unlock and
  // the futex calls under them were a quarter of the keyword leg's host CPU (profiles/r6_kw_fresh_profile_before.txt:
  // __lll_lock_wait_private 10 %, __lll_lock_wake_private 8.5 %, pthread_mutex_lock / unlock 6.4 %).  Now a search pushes its
  // VmSub (one per calling thread, never freed) onto an intrusive stack with one compare-exchange; the combiner takes the
  // whole stack with one exchange and reverses it (arrival order).  `mu` / `cv` only serve the combiner's sleep when nothing
  // at all is in flight: the combiner announces `sleeping` BEFORE it re-reads the stack (both seq_cst), a submitter reads
  // `sleeping` AFTER its push — one of the two sees the other — and passes through `mu` before notifying.
  std::mutex mu;
  std::condition_variable cv;
  std::atomic<VmSub *> q_head{nullptr};    // submitted, not yet taken (newest first)
  std::atomic<uint32_t> sleeping{0};       // the combiner waits on cv
  std::atomic<bool> stop{false};
  struct Arena {
    uint8_t *host = nullptr, *dev = nullptr;
    size_t cap = 0;
    hipEvent_t done = nullptr;
    bool in_flight = false;
  } ar[NS];
  std::atomic<uint64_t> rounds{0}, lists{0};
  // where a list's wall time goes (ns, summed over lists): queued until the combiner took it, packed until the launch
  // calls began, launch calls (per round), from the launch calls until the combiner saw the GPU's completion word
  std::atomic<uint64_t> ns_launch_calls{0};
  StripedCounters<3> ns_waiters;   // [queued, packed, after launch]: added by the search threads
  std::atomic<uint32_t> load{0};           // lists submitted and not finished yet
  u64 *d_prof = nullptr;                   // MSI_VM_PROFILE: 16 tick counters in device memory
  std::atomic<int32_t> *fused_wgs = nullptr;   // msi_vm::fused_wgs: the device's waiting workgroups in flight
  int32_t fused_budget = 256;                  // msi_vm::fused_budget
  // The reaper (MSI_VM_REAPER=1; the thread exists when the variable is set at all, the value is read per round): a second
  // thread that only watches the completion words of the rounds in flight and wakes their searches.  The hypothesis: one
  // thread doing everything serves a round in ~120 us (pack 23 lists, one copy, 2-3 launch calls, then 23 futex wake-ups)
  // and is busy 0.66-0.73 of the time at 12-13 k searches/s, so completions are noticed late and lists queue behind the
  // wake-ups.  MEASURED (profiles/r5_reaper.log, 10 M documents, fresh queries, same process): 12.4 / 13.0 k searches/s with
  // it against 12.8 / 13.2 k without at 256 callers, launch -> wake-up 887-913 us per list either way — the time between a
  // round's launch and its searches' wake-up is the device running the round's kernels beside 5 other rounds, not the
  // host noticing late.  Off by default.
  bool split = false;
  std::thread reaper_th;
  std::mutex hand_mu, trace_mu;
  std::condition_variable hand_cv;
  std::vector<VmSub *> handed;             // launched, not yet seen by the reaper (guarded by hand_mu)
  bool reaper_stop = false;                // guarded by hand_mu
  std::atomic<uint32_t> n_inflight{0};     // launched and not finished (either thread's view of "rounds in flight")
  FILE *trace = nullptr;
  void finish(VmSub *s, uint32_t st);
  bool settle(VmSub *s, int64_t t_now);    // -> true when `s` is finished (completion word seen, or given up on after 5 s)
  void run();
  void reap();
};

// The context's combiners: pools are spread over a few of them (a combiner is one thread: packing, launching, noticing
// completions and waking the searches of its pools cost it ~4 us per list, ~250 k lists/s — r2_ranked_2m_vm_v3*.jsonl).
struct msi_vm {
  msi_ctx *ctx = nullptr;
  std::vector<std::unique_ptr<VmCombiner>> comb;
  // Workgroups of fused lists that WAIT for their list's wide phase, over all combiners' rounds in flight.  A waiting
  // workgroup holds a CU slot and does nothing.  Inside ONE kernel that is harmless — a list's waiters follow its own wide
  // workgroups in dispatch order — but rounds run as separate kernels on 16 streams, each XCD dispatches its share of a
  // kernel's workgroups on its own, and nothing orders kernel A's waiters against kernel B's wide workgroups: once the
  // waiters in flight exceed what the device keeps RESIDENT (CUs x workgroups per CU of vm_kernel: 256 x 4 = 1 024 on an
  // MI355X) they can take every slot of an XCD while the wide workgroups they wait for — their own list's, assigned to
  // that XCD — still queue behind them.  That is what round 4 measured as a collapse (9 700 -> 23-170 searches/s under
  // a burst of alike searches with 48-153-chunk lists fused and a budget of 4 096 waiters = 4x the residency; DESIGN
  // 4.7, profiles/r5_fuse_collapse.txt): forward progress then hangs on the hardware scheduler's queue time-slicing.
  // The bound is therefore DERIVED FROM THE RESIDENCY: waiters in flight <= resident workgroups / 4 (MSI_VM_FUSED_WGS_PCT,
  // per cent of the residency, default 25), so that three quarters of every XCD's slots always go to workgroups that do
  // work; a list that does not fit the budget runs its two phases as two launches.  A waiter that still outlasts
  // MSI_VM_SPIN_LIMIT_TICKS fails its list instead of hanging the device (vm_kernel).
  std::atomic<int32_t> fused_wgs{0};
  int32_t resident_wgs = 1024;             // CUs x occupancy of vm_kernel (hipOccupancyMaxActiveBlocksPerMultiprocessor)
  int32_t fused_budget = 256;
};

namespace {

size_t align16(size_t x) { return (x + 15) & ~(size_t)15; }
uint32_t chunks_of(msi_bits *p) { return (uint32_t)((msi_bits_words_per_slot(p) + CHW - 1) / CHW); }

bool arena_ensure(VmCombiner *vm, VmCombiner::Arena &A, size_t bytes) {
  if (bytes <= A.cap) return true;
  if (A.in_flight) {
    (void)hipEventSynchronize(A.done);
    A.in_flight = false;
  }
  if (A.host) (void)hipHostFree(A.host);
  if (A.dev) (void)hipFree(A.dev);
  A.host = A.dev = nullptr;
  A.cap = 0;
  const size_t cap = std::max<size_t>(bytes * 2, (size_t)4 << 20);
  void *h = nullptr, *d = nullptr;
  if (hipHostMalloc(&h, cap, hipHostMallocDefault) != hipSuccess) return false;
  if (hipMalloc(&d, cap) != hipSuccess) {
    (void)hipHostFree(h);
    return false;
  }
  A.host = (uint8_t *)h;
  A.dev = (uint8_t *)d;
  A.cap = cap;
  if (!A.done && hipEventCreateWithFlags(&A.done, hipEventDisableTiming) != hipSuccess) return false;
  return true;
}

}  // namespace

static inline int64_t now_ns() {
  return std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

static void futex_wake_all(std::atomic<uint32_t> *w) {
#if defined(__linux__)
  syscall(SYS_futex, reinterpret_cast<uint32_t *>(w), FUTEX_WAKE_PRIVATE, INT32_MAX, nullptr, nullptr, 0);
#endif
}
static void futex_wait_for(std::atomic<uint32_t> *w, uint32_t expected, long timeout_us) {
#if defined(__linux__)
  struct timespec ts;
  ts.tv_sec = timeout_us / 1000000;
  ts.tv_nsec = (timeout_us % 1000000) * 1000;
  syscall(SYS_futex, reinterpret_cast<uint32_t *>(w), FUTEX_WAIT_PRIVATE, expected, &ts, nullptr, 0);
#else
  (void)w; (void)expected;
  std::this_thread::sleep_for(std::chrono::microseconds(std::min<long>(timeout_us, 50)));
#endif
}

void VmCombiner::finish(VmSub *s, uint32_t st) {
  s->t_done = now_ns();
  if (s->fused_wgs) {
    fused_wgs->fetch_sub((int32_t)s->fused_wgs, std::memory_order_relaxed);
    s->fused_wgs = 0;
  }
  if (trace && s->list) {   // diagnostics: what a list was made of and how long the device took for it
    std::lock_guard<std::mutex> tlk(trace_mu);
    const std::vector<uint32_t> &w = s->list->words;
    uint32_t ops[32] = {0}, max_paths = 0, max_steps = 0, clear_slots = 0;
    const size_t w_end = s->list->data_off ? s->list->data_off : w.size();
    for (size_t i = 0; i < w_end;) {
      const uint32_t op = w[i];
      if (op < 32) ++ops[op];
      switch (op) {
        case VM_END: i += 1; break;
        case VM_FILL: i += 3; break;
        case VM_OP: i += 5; break;
        case VM_OP_COUNT: i += 6; break;
        case VM_CLEAR: clear_slots += w[i + 1]; i += 2 + w[i + 1]; break;
        case VM_CLAIM: i += 5 + w[i + 4]; break;
        case VM_AND_MANY: i += 4 + 2 * w[i + 2]; break;
        case VM_PATHS: max_paths = std::max(max_paths, w[i + 1]); max_steps = std::max(max_steps, w[i + 5] & 0x7FFFFFFFu); i += 7 + w[i + 1] + (w[i + 5] & 0x7FFFFFFFu); break;
        case VM_SUB_MANY: i += 4 + w[i + 2]; break;
        case VM_COUNT: i += 3; break;
        case VM_DECODE: i += 4; break;
        case VM_FIRSTK: i += 5; break;
        case VM_MINKEY: i += 5; break;
        case VM_TAKEKEY: i += 8; break;
        case VM_SUMMARY_RESET: i += 1; break;
        case VM_RANK_A: i += 4; break;
        case VM_RANK_B: i += 5; break;
        case VM_DECODEC: i += 3; break;
        default: i = w_end; break;
      }
    }
    fprintf(trace, "%s%u phases counts %u rankA %u rankB %u decodeC %u | ", s->list->geom_docs ? "compact " : "full ", (unsigned)s->list->phase_start.size(),
            s->list->n_counts, ops[VM_RANK_A], ops[VM_RANK_B], ops[VM_DECODEC]);
    fprintf(trace, "%.1f us words %zu stage %zu fill %u op %u opc %u clear %u(%u slots) claim %u andmany %u paths %u(max %u paths %u steps) sub %u count %u decode %u firstk %u\n",
            (s->t_done - s->t_launch) / 1e3, w_end, s->list->stage_used, ops[VM_FILL], ops[VM_OP], ops[VM_OP_COUNT], ops[VM_CLEAR],
            clear_slots, ops[VM_CLAIM], ops[VM_AND_MANY], ops[VM_PATHS], max_paths, max_steps, ops[VM_SUB_MANY], ops[VM_COUNT],
            ops[VM_DECODE], ops[VM_FIRSTK]);
  }
  s->state.store(st, std::memory_order_seq_cst);
  // a waiter that is still polling needs no system call (the wake-up is the combiner's most expensive step)
  if (s->asleep.load(std::memory_order_seq_cst)) futex_wake_all(&s->state);
}

// one list in flight: has the GPU stored its sequence number into its pool's pinned block?
bool VmCombiner::settle(VmSub *s, int64_t t_now) {
  if (__atomic_load_n(const_cast<uint64_t *>(&s->blk[0]), __ATOMIC_ACQUIRE) == s->seq) {
    finish(s, 1);
    return true;
  }
  if (t_now - s->t_launch > 5000000000ll) {   // 5 s: settle it with the streams, then give up on it
    for (auto stq : streams) (void)hipStreamSynchronize(stq);
    if (__atomic_load_n(const_cast<uint64_t *>(&s->blk[0]), __ATOMIC_ACQUIRE) == s->seq) {
      finish(s, 1);
    } else {
      s->error = MSI_E_INTERNAL;
      finish(s, 2);
    }
    return true;
  }
  return false;
}

// The reaper's loop: the rounds handed over by the launching thread, polled until their lists have run.
void VmCombiner::reap() {
  (void)hipSetDevice(ctx->device);
  const long poll_sleep_ns = (getenv("MSI_VM_POLL_SLEEP_US") ? std::max(0, atoi(getenv("MSI_VM_POLL_SLEEP_US"))) : 20) * 1000l;
  if (poll_sleep_ns > 0) prctl(PR_SET_TIMERSLACK, 1000ul, 0, 0, 0);
  std::vector<VmSub *> mine;
  uint64_t cpu_seen = msi_cpu_prof_on() ? msi_thread_cpu_ns() : 0;
  for (;;) {
    {
      std::unique_lock<std::mutex> lk(hand_mu);
      if (mine.empty()) hand_cv.wait(lk, [&] { return reaper_stop || !handed.empty(); });
      if (reaper_stop && handed.empty() && mine.empty()) return;
      mine.insert(mine.end(), handed.begin(), handed.end());
      handed.clear();
    }
    size_t kept = 0;
    const int64_t t_now = now_ns();
    for (VmSub *s : mine) {
      if (settle(s, t_now)) n_inflight.fetch_sub(1, std::memory_order_release);
      else mine[kept++] = s;
    }
    const bool progress = kept != mine.size();
    mine.resize(kept);
    if (progress && msi_cpu_prof_on()) {   // this thread's CPU joins the combiner's (msi_search_cpu_profile [6])
      const uint64_t now_cpu = msi_thread_cpu_ns();
      msi_cpu_prof_add(6, now_cpu - cpu_seen);
      cpu_seen = now_cpu;
    }
    if (!mine.empty() && !progress) {   // (as the combiner's own polling: a nap under load, a spin when a search's latency is the round trip)
      if (poll_sleep_ns > 0 && load.load(std::memory_order_relaxed) > 3) {
        struct timespec ts = {0, poll_sleep_ns};
        nanosleep(&ts, nullptr);
      } else {
#if defined(__x86_64__)
        __builtin_ia32_pause();
#endif
      }
    }
  }
}

void VmCombiner::run() {
  (void)hipSetDevice(ctx->device);
  std::vector<VmSub *> inflight, taken;
  // Two classes of rounds, each with its own ring of arenas / streams (light: even indices, heavy: odd): a list with
  // many decodes (the postings of a whole condition) or a large level takes 100+ us on the device, the typical list
  // ~10 us; a kernel ends when its slowest list ends and the next kernel on that stream waits for it, so heavy lists
  // get rounds of their own and the light ones keep flowing.
  int cur_of[2] = {0, 1};
  const size_t max_subs = getenv("MSI_VM_MAX_SUBS") ? std::max(1, atoi(getenv("MSI_VM_MAX_SUBS"))) : MAX_SUBS;   // experiments
  // (a compact list's decodes are as heavy as a full list's: they read the same posting bytes)
  auto is_heavy = [](const VmSub *s) { return s->list->decodes.size() > 2 || s->list->words.size() > 600; };
  std::vector<VmSub *> waiting[2];   // taken from the queue, not launched yet (their class's arena is still in use)
  trace = getenv("MSI_VM_TRACE") ? fopen(getenv("MSI_VM_TRACE"), "w") : nullptr;
  const int64_t batch_wait_ns = (getenv("MSI_VM_BATCH_WAIT_US") ? atoi(getenv("MSI_VM_BATCH_WAIT_US")) : 200) * 1000ll;
  const size_t batch_div = getenv("MSI_VM_BATCH_DIV") ? std::max(1, atoi(getenv("MSI_VM_BATCH_DIV"))) : 2;
  const size_t batch_cap = getenv("MSI_VM_BATCH_CAP") ? std::max(1, atoi(getenv("MSI_VM_BATCH_CAP"))) : 32;
  const long poll_sleep_ns = (getenv("MSI_VM_POLL_SLEEP_US") ? std::max(0, atoi(getenv("MSI_VM_POLL_SLEEP_US"))) : 20) * 1000l;
  const bool arena_fifo = getenv("MSI_VM_ARENA_FIFO") && getenv("MSI_VM_ARENA_FIFO")[0] == '1';
  if (poll_sleep_ns > 0) prctl(PR_SET_TIMERSLACK, 1000ul, 0, 0, 0);   // (this thread only: the default 50 us slack would triple the sleep)
  if (getenv("MSI_VM_PROFILE") && hipMalloc((void **)&d_prof, 32 * sizeof(u64)) == hipSuccess) (void)hipMemset(d_prof, 0, 32 * sizeof(u64));
  uint64_t cpu_seen = msi_cpu_prof_on() ? msi_thread_cpu_ns() : 0;
  // one round: pack the lists into the class's next arena, one H2D copy, one launch per phase
  auto launch_round = [&](int cls, std::vector<VmSub *> &batch) {
    const int cur = cur_of[cls];
    Arena &A = ar[cur];
    hipStream_t stream = streams[cur];
    const size_t n_sub = batch.size();
    {   // (read per round, as the knobs below: one process can hold the two forms side by side — tools/kw_leg.py --sweep)
      const char *rk = getenv("MSI_VM_REAPER");   // 1: the reaper notices the round's completion (off by default: no gain measured)
      split = reaper_th.joinable() && rk && rk[0] == '1';
    }
    size_t off = 64 + align16(n_sub * sizeof(RoundSub));
    std::vector<size_t> words_at(n_sub), state_at(n_sub);
    // geometry of a list: its pool's, or — a compact list — geom_docs documents per set on the companion pool
    auto words_of = [](const VmSub *b) -> uint64_t {
      const uint64_t g = b->list->geom_docs;
      return g ? std::max<uint64_t>(2, ((g + 127) / 128) * 2) : msi_bits_words_per_slot(b->pool);
    };
    // words per workgroup in a list's command phases (RoundSub::chw): compact lists take narrower chunks, so that a dozen
    // sets of a chunk fit in the kernel's LDS set cache.  MSI_VM_COMPACT_CHW = 128 | 256 | 512 | 1024 (experiments)
    // (both knobs are read per round, so that one process can measure the variants side by side: tools/ranked_bench RB_VARIANTS)
    // MSI_VM_COMPACT_CHW = 128 | 256 | 512 | 1024 | 2048: that width for every compact list; "auto<N>": 256 words, doubled
    // while the list would have more than N workgroups (a universe of a million documents is 153 chunks of 256 words: as
    // many workgroups as the full space, each with an eighth of the words — workgroup slots, not words, are what runs out)
    uint32_t compact_chw = 256, compact_target = 0;
    if (const char *e = getenv("MSI_VM_COMPACT_CHW")) {
      if (!strncmp(e, "auto", 4)) compact_target = (uint32_t)std::max(1, atoi(e + 4));
      else {
        const int v = atoi(e);
        if (v == 128 || v == 256 || v == 512 || v == 1024 || v == 2048) compact_chw = (uint32_t)v;
      }
    }
    const char *cache_knob = getenv("MSI_VM_CACHE");
    const bool cache_off = !(cache_knob && cache_knob[0] == '1');   // the LDS set cache (builds with MSI_VM_SET_CACHE=1 only): on request
    auto chw_of = [&](const VmSub *b) -> uint32_t {
      if (!b->list->geom_docs) return CHW;
      uint32_t w = compact_chw;
      if (compact_target) {
        const uint64_t words = words_of(b);
        while (w < CHW && (words + w - 1) / w > compact_target) w *= 2;
      }
      return w;
    };
    // lists of at most this many chunks run their wide phase and their commands in one launch (`fused`); 0 = never
    const uint32_t fuse_max_chunks = [] {
      const char *e = getenv("MSI_VM_FUSE_MAX_CHUNKS");
      return e ? (uint32_t)std::max(0, atoi(e)) : 24u;
    }();
    uint32_t max_chunks[MSI_VM_MAX_PHASES] = {0}, max_phases = 1;
    for (size_t i = 0; i < n_sub; ++i) {
      const MsiVmList &l = *batch[i]->list;
      words_at[i] = off;
      off = align16(off + (l.words.size() + 1) * 4);
      state_at[i] = off;
      const uint32_t n_chunks = (uint32_t)((words_of(batch[i]) + chw_of(batch[i]) - 1) / chw_of(batch[i]));
      off = align16(off + 16 + MSI_VM_CELLS * 8 + (size_t)l.n_counts * 8 + (size_t)n_chunks * 4 * l.max_fk_phase);
    }
    int32_t st = MSI_OK;
    if (off > 0xFFFFFFF0ull || !arena_ensure(this, A, off)) {
      msi_set_error("msi_vm: arena of %zu bytes not available", off);
      st = MSI_E_OOM;
    }
    if (st == MSI_OK) {
      memset(A.host, 0, 64);
      reinterpret_cast<uint32_t *>(A.host)[0] = (uint32_t)n_sub;
      reinterpret_cast<uint32_t *>(A.host)[2] = (uint32_t)(uintptr_t)d_prof;
      reinterpret_cast<uint32_t *>(A.host)[3] = (uint32_t)((uintptr_t)d_prof >> 32);
      RoundSub *subs = reinterpret_cast<RoundSub *>(A.host + 64);
      for (size_t i = 0; i < n_sub; ++i) {
        const MsiVmList &l = *batch[i]->list;
        msi_bits *p = batch[i]->pool;
        RoundSub &r = subs[i];
        memset(&r, 0, sizeof(r));
        r.pool_base = (u64)(uintptr_t)msi_bits_pool_base(p);
        r.n_words = words_of(batch[i]);
        r.n_docs = l.geom_docs ? l.geom_docs : msi_bits_n_docs(p);
        r.host_res = (u64)(uintptr_t)batch[i]->blk;
        r.seq = batch[i]->seq;
        r.chw = chw_of(batch[i]);
        r.cache_on = cache_off ? 0u : 1u;
        r.n_chunks = (uint32_t)((r.n_words + r.chw - 1) / r.chw);
        r.n_phases = (uint32_t)l.phase_start.size();
        for (uint32_t ph = 0; ph < r.n_phases; ++ph) r.phase_off[ph] = (uint32_t)(words_at[i] / 4) + l.phase_start[ph];
        r.list_off = (uint32_t)(words_at[i] / 4);
        r.stage = l.stage_used ? (u64)(uintptr_t)msi_bits_vm_stage(p, 0, 0) : 0;
        r.cache = l.cache_base;
        r.state_off = (uint32_t)(state_at[i] / 4);
        r.n_counts = l.n_counts;
        r.data_off = l.data_off;
        r.n_decodes = (uint32_t)l.decodes.size();
        r.n_cmd_words = l.data_off ? l.data_off : (uint32_t)l.words.size() + 1;
        if (l.geom_docs) {
          // a compact list: the full pool's tables and slots; chunk summaries are not kept in the compact space (its sets
          // are a chunk or two long)
          r.aux = (u64)(uintptr_t)msi_bits_compact_aux(l.full_pool);
          r.full_base = (u64)(uintptr_t)msi_bits_pool_base(l.full_pool);
          r.full_words = (uint32_t)msi_bits_words_per_slot(l.full_pool);
          r.u0_slot = l.u0_slot;
          r.wide_chunks = (uint32_t)((r.full_words + CHW - 1) / CHW);
          r.wide_mask = l.pre_merged ? 1u : 0u;
          r.p0_wgs = r.wide_chunks;
          // the decode phase by rank (vm_kernel): a workgroup per 64 documents of U0 instead of one per chunk of the full space,
          // when that is clearly less work — each of its lanes pays a binary search per command where a wide workgroup pays
          // its staging once.  MSI_VM_BY_RANK_MAX_DOCS (experiments; 0: never) caps |U0|.
          static const uint32_t by_rank_max = getenv("MSI_VM_BY_RANK_MAX_DOCS") ? (uint32_t)std::max(0, atoi(getenv("MSI_VM_BY_RANK_MAX_DOCS"))) : 4096u;
          // (MSI_VM_BY_RANK_FORCE=1, tests: also when the full space has fewer chunks than U0 has words — small corpora)
          static const bool by_rank_force = getenv("MSI_VM_BY_RANK_FORCE") && getenv("MSI_VM_BY_RANK_FORCE")[0] == '1';
          if (l.pre_merged && l.geom_docs <= by_rank_max && ((l.geom_docs + 63) / 64 < r.wide_chunks || by_rank_force)) {
            r.wide_mask |= 0x40000000u | (l.any_fill ? 0x20000000u : 0u);
            r.p0_wgs = (uint32_t)std::max<uint64_t>(1, (l.geom_docs + 63) / 64);
          }
          // phases 0 and 1 in one launch (vm_kernel, `fused`): only while the waiting workgroups are few
          static const bool fuse_off = getenv("MSI_VM_FUSE") && getenv("MSI_VM_FUSE")[0] == '0';   // experiments
          // (the waiting workgroups of a list follow its own wide workgroups in dispatch order, so they can only ever wait
          // for workgroups that are already resident: the bound keeps spinning workgroups few, it is not what makes this safe)
          batch[i]->fused_wgs = 0;
          if (l.pre_merged && r.n_phases >= 2 && r.n_chunks <= fuse_max_chunks && r.n_words <= (u64)fuse_max_chunks * 256 && !fuse_off) {
            if (fused_wgs->fetch_add((int32_t)r.n_chunks, std::memory_order_relaxed) + (int32_t)r.n_chunks <= fused_budget) {
              batch[i]->fused_wgs = r.n_chunks;
              r.wide_mask |= 0x80000000u;
            } else {
              fused_wgs->fetch_sub((int32_t)r.n_chunks, std::memory_order_relaxed);   // over the budget: two launches
            }
          }
        } else {
          static const bool sum_off = getenv("MSI_VM_SUMMARY") && getenv("MSI_VM_SUMMARY")[0] == '0';   // diagnostics
          const uint64_t sp = sum_off ? 0 : (uint64_t)(uintptr_t)msi_bits_summary(p);
          r.sum_lo = (uint32_t)sp;
          r.sum_hi = (uint32_t)(sp >> 32);
        }
        memcpy(A.host + words_at[i], l.words.data(), l.words.size() * 4);
        reinterpret_cast<uint32_t *>(A.host + words_at[i])[l.words.size()] = VM_END;
        memset(A.host + state_at[i], 0, 16 + MSI_VM_CELLS * 8 + (size_t)l.n_counts * 8 + (size_t)r.n_chunks * 4 * l.max_fk_phase);
        for (uint32_t ph = 0; ph < r.n_phases; ++ph) {
          if (r.wide_mask & 0x80000000u) {
            if (ph == 0) max_chunks[0] = std::max(max_chunks[0], r.p0_wgs + r.n_chunks);
            if (ph <= 1) continue;
          }
          max_chunks[ph] = std::max(max_chunks[ph], ((r.wide_mask >> ph) & 1u) ? r.p0_wgs : r.n_chunks);
        }
        max_phases = std::max(max_phases, r.n_phases);
      }
      const int64_t t_launch = now_ns();
      for (VmSub *s : batch) s->t_launch = t_launch;
      if (hipMemcpyAsync(A.dev, A.host, off, hipMemcpyHostToDevice, stream) != hipSuccess) st = MSI_E_HIP;
      for (uint32_t ph = 0; ph < max_phases && st == MSI_OK; ++ph) {
        if (!max_chunks[ph]) continue;   // (phase 1 of fused lists ran with launch 0)
        hipLaunchKernelGGL(vm_kernel, dim3(std::max(1u, max_chunks[ph]), (uint32_t)n_sub), dim3(VT), 0, stream,
                           reinterpret_cast<uint32_t *>(A.dev), ph);
        if (hipGetLastError() != hipSuccess) st = MSI_E_HIP;
      }
      if (st == MSI_OK && hipEventRecord(A.done, stream) == hipSuccess) A.in_flight = true;
      ns_launch_calls.fetch_add((uint64_t)(now_ns() - t_launch), std::memory_order_relaxed);
      if (st != MSI_OK) msi_set_error("msi_vm: launching a round of %zu lists failed", n_sub);
    }
    rounds.fetch_add(1, std::memory_order_relaxed);
    lists.fetch_add(n_sub, std::memory_order_relaxed);
    if (msi_cpu_prof_on()) {   // this thread's CPU since the last round
      const uint64_t now_cpu = msi_thread_cpu_ns();
      msi_cpu_prof_add(6, now_cpu - cpu_seen);
      cpu_seen = now_cpu;
    }
    if (st != MSI_OK) {
      for (VmSub *s : batch) {
        s->error = st;
        snprintf(s->errmsg, sizeof(s->errmsg), "%s", msi_last_error());
        finish(s, 2);
      }
    } else {
      n_inflight.fetch_add((uint32_t)batch.size(), std::memory_order_relaxed);
      if (split) {   // the reaper watches them from here on
        {
          std::lock_guard<std::mutex> lk(hand_mu);
          handed.insert(handed.end(), batch.begin(), batch.end());
        }
        hand_cv.notify_one();
      } else {
        inflight.insert(inflight.end(), batch.begin(), batch.end());
      }
    }
    batch.clear();
    cur_of[cls] = (cur + 2) % ns;
  };
  for (;;) {
    taken.clear();
    {
      VmSub *h = q_head.exchange(nullptr, std::memory_order_acquire);
      const bool idle = n_inflight.load(std::memory_order_acquire) == 0 && waiting[0].empty() && waiting[1].empty();
      if (!h && idle) {
        std::unique_lock<std::mutex> lk(mu);
        sleeping.store(1, std::memory_order_seq_cst);
        cv.wait(lk, [&] { return stop.load(std::memory_order_seq_cst) || q_head.load(std::memory_order_seq_cst) != nullptr; });
        sleeping.store(0, std::memory_order_seq_cst);
        lk.unlock();
        h = q_head.exchange(nullptr, std::memory_order_acquire);
      }
      if (!h && idle && stop.load(std::memory_order_acquire)) return;
      for (; h; h = h->next) taken.push_back(h);
      std::reverse(taken.begin(), taken.end());   // (the stack holds the newest first)
    }
    if (!taken.empty()) {
      const int64_t t_taken = now_ns();
      for (VmSub *s : taken) {
        s->t_taken = t_taken;
        waiting[is_heavy(s) ? 1 : 0].push_back(s);
      }
    }
    // the combiner never blocks on the device (it is also the thread that notices completions): a class whose next
    // arena is still in use keeps its lists waiting
    for (int cls = 0; cls < 2; ++cls) {
      if (waiting[cls].empty()) continue;
      // With many searches in flight a round is cheaper per list the more lists it carries (one launch, one copy, one
      // pass over the interpreter's fixed costs): while other rounds keep the device busy, a batch smaller than half of
      // the searches in flight (at most 32) waits up to 200 us for company.  10 M documents, detailed scores, 64 callers:
      // 3 250 -> 4 555 queries/s, p50 19.3 -> 13.3 ms; one caller is never held (nothing else is in flight).
      // (MSI_VM_BATCH_WAIT_US / _DIV / _CAP: experiments; profiles/r2_ranked_10m_batching.txt)
      if (batch_wait_ns && n_inflight.load(std::memory_order_relaxed) != 0 && waiting[cls].size() < std::min<size_t>(batch_cap, load.load(std::memory_order_relaxed) / batch_div) &&
          now_ns() - waiting[cls].front()->t_taken < batch_wait_ns)
        continue;
      // the class's next arena — or, when that one's round is still running, any other arena of the class whose round has
      // finished (rounds differ in length by an order of magnitude: the next in line is not the next to finish)
      int free_at = -1;
      for (int k = 0; k < ns / 2 && free_at < 0; ++k) {
        const int idx = (cur_of[cls] + 2 * k) % ns;
        Arena &Ak = ar[idx];
        if (Ak.in_flight && hipEventQuery(Ak.done) == hipSuccess) Ak.in_flight = false;
        if (!Ak.in_flight) free_at = idx;
        if (arena_fifo) break;   // (MSI_VM_ARENA_FIFO=1: round 5's strict rotation)
      }
      if (free_at < 0) continue;
      cur_of[cls] = free_at;
      std::vector<VmSub *> batch;
      const size_t n = std::min<size_t>(waiting[cls].size(), std::min<size_t>(max_subs, MAX_SUBS));
      batch.assign(waiting[cls].begin(), waiting[cls].begin() + n);
      waiting[cls].erase(waiting[cls].begin(), waiting[cls].begin() + n);
      launch_round(cls, batch);
    }
    // ---- completion: the GPU stored the list's sequence number into its pool's pinned block (the reaper's work when
    // there is one: VmCombiner::reap) -----------------------------------------------------------------------------
    if (!inflight.empty()) {   // (rounds this thread kept: MSI_VM_REAPER=0 when they were launched)
      size_t kept = 0;
      const int64_t t_now = now_ns();
      for (VmSub *s : inflight) {
        if (settle(s, t_now)) n_inflight.fetch_sub(1, std::memory_order_release);
        else inflight[kept++] = s;
      }
      inflight.resize(kept);
    }
    if (taken.empty() && n_inflight.load(std::memory_order_relaxed) != 0) {
      // Nothing new and rounds in flight: the combiner polls their completion words.  With a few searches in flight it
      // spins (their latency is the round trip); under load it sleeps 20 us between polls (MSI_VM_POLL_SLEEP_US, 0 = always
      // spin): measured free on one GPU (8.1 k -> 8.2-8.3 k keyword searches/s, half a CPU less: profiles/r3_pollsleep.txt),
      // and on a host where several GPUs' combiners share the CPUs a spinning thread per GPU is a CPU per GPU.
      if (poll_sleep_ns > 0 && load.load(std::memory_order_relaxed) > 3) {
        struct timespec ts = {0, poll_sleep_ns};
        nanosleep(&ts, nullptr);
      } else {
#if defined(__x86_64__)
        __builtin_ia32_pause();
#endif
      }
    }
  }
}

static msi_vm *vm_of(msi_ctx *ctx) {
  std::lock_guard<std::mutex> lk(ctx->vm_mu);
  if (!ctx->vm) {
    DeviceGuard g(ctx->device);
    msi_vm *vm = new msi_vm();
    vm->ctx = ctx;
    const char *knob = getenv("MSI_VM_COMBINERS");
    const int n = std::max(1, std::min(8, knob ? atoi(knob) : 1));   // one is best: more combiners mean more, smaller rounds
    {   // the budget of waiting workgroups follows from what the device keeps resident (struct msi_vm)
      int occ = 0;
      if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, vm_kernel, VT, 0) != hipSuccess || occ <= 0) occ = 4;
      vm->resident_wgs = std::max(1, ctx->n_cu) * occ;
      const char *pct = getenv("MSI_VM_FUSED_WGS_PCT");   // experiments: 400 = round 4's 4 096 on an MI355X
      vm->fused_budget = (int32_t)((int64_t)vm->resident_wgs * std::max(0, pct ? atoi(pct) : 25) / 100);
    }
    for (int c = 0; c < n; ++c) {
      std::unique_ptr<VmCombiner> cb(new VmCombiner());
      cb->ctx = ctx;
      cb->fused_wgs = &vm->fused_wgs;
      cb->fused_budget = vm->fused_budget;
      if (const char *e = getenv("MSI_VM_STREAMS")) cb->ns = std::max(4, std::min((int)VmCombiner::NS, atoi(e) & ~1));
      // MSI_VM_STREAM_PRIORITY=1 (experiment, round 6): the rounds' streams at the device's highest dispatch priority — beside
      // a vector sweep launched as many SHORT workgroups (MSI_VS_GRID_MULT) a round's workgroups are then dispatched as soon
      // as a sweep's workgroup ends instead of behind the whole sweep
      int prio_least = 0, prio_greatest = 0;
      const char *pk = getenv("MSI_VM_STREAM_PRIORITY");
      const bool high = pk && pk[0] == '1' && hipDeviceGetStreamPriorityRange(&prio_least, &prio_greatest) == hipSuccess && prio_greatest != prio_least;
      for (int i = 0; i < cb->ns; ++i)
        if ((high ? hipStreamCreateWithPriority(&cb->streams[i], hipStreamNonBlocking, prio_greatest)
                  : hipStreamCreateWithFlags(&cb->streams[i], hipStreamNonBlocking)) != hipSuccess) {
          msi_set_error("msi_vm: hipStreamCreate failed");
          delete vm;   // (what was created so far leaks with the failed context; nothing runs on it)
          return nullptr;
        }
      VmCombiner *raw = cb.get();
      if (getenv("MSI_VM_REAPER")) cb->reaper_th = std::thread([raw] { raw->reap(); });   // (an experiment: asleep unless a round is handed to it)
      cb->th = std::thread([raw] { raw->run(); });
      vm->comb.push_back(std::move(cb));
    }
    ctx->vm = vm;
  }
  return ctx->vm;
}

void msi_vm_destroy(msi_vm *vmx) {
  if (!vmx) return;
  for (auto &cb : vmx->comb) {
    VmCombiner *vm = cb.get();
    {
      std::lock_guard<std::mutex> lk(vm->mu);
      vm->stop.store(true, std::memory_order_seq_cst);
    }
    vm->cv.notify_all();
    if (vm->th.joinable()) vm->th.join();   // (returns once nothing is queued, waiting or in flight: the reaper is still there for that)
    {
      std::lock_guard<std::mutex> lk(vm->hand_mu);
      vm->reaper_stop = true;
    }
    vm->hand_cv.notify_all();
    if (vm->reaper_th.joinable()) vm->reaper_th.join();
    DeviceGuard g(vm->ctx->device);
    for (auto st : vm->streams)
      if (st) (void)hipStreamSynchronize(st);
    if (vm->d_prof) {
      u64 t[32] = {0};
      (void)hipMemcpy(t, vm->d_prof, sizeof t, hipMemcpyDeviceToHost);
      static const char *names[16] = {"", "fill", "op", "op_count", "clear", "claim", "and_many", "paths", "sub_many", "count",
                                      "decode", "firstk", "minkey", "takekey", "rank", "all"};
      fprintf(stderr, "msi_vm profile: %llu workgroups, %.1f us each;", (unsigned long long)t[0], t[0] ? t[15] / 100.0 / t[0] : 0.0);
      for (int i = 1; i < 15; ++i)
        if (t[i]) fprintf(stderr, " %s %.1f%%", names[i], 100.0 * t[i] / (double)t[15]);
      fprintf(stderr, "; scoped (command, chunk) executions %llu, of them on an all-zero chunk %llu", (unsigned long long)t[20], (unsigned long long)t[21]);
      fprintf(stderr, "; epilogue %.1f us per workgroup; %llu list-phases, first start to last ticket %.1f us",
              t[0] ? t[18] / 100.0 / t[0] : 0.0, (unsigned long long)t[17], t[17] ? t[16] / 100.0 / t[17] : 0.0);
      fprintf(stderr, "; wide workgroups %llu, %.1f us each (staging %.1f us); the others %.1f us each\n", (unsigned long long)t[22],
              t[22] ? t[19] / 100.0 / t[22] : 0.0, t[22] ? t[23] / 100.0 / t[22] : 0.0,
              t[0] > t[22] ? (t[15] - t[19]) / 100.0 / (t[0] - t[22]) : 0.0);
      if (t[25]) fprintf(stderr, "msi_vm profile: %llu workgroups of fused lists waited %.1f us each for their wide phase\n",
                         (unsigned long long)t[25], t[24] / 100.0 / t[25]);
      (void)hipFree(vm->d_prof);
    }
    for (auto &A : vm->ar) {
      if (A.host) (void)hipHostFree(A.host);
      if (A.dev) (void)hipFree(A.dev);
      if (A.done) (void)hipEventDestroy(A.done);
    }
    for (auto st : vm->streams)
      if (st) (void)hipStreamDestroy(st);
  }
  delete vmx;
}

void msi_vm_stats(msi_bits *pool, uint64_t out[6]) {
  msi_ctx *ctx = msi_bits_ctx(pool);
  std::lock_guard<std::mutex> lk(ctx->vm_mu);
  for (int i = 0; i < 6; ++i) out[i] = 0;
  if (!ctx->vm) return;
  for (auto &cb : ctx->vm->comb) {
    out[0] += cb->rounds.load();
    out[1] += cb->lists.load();
    out[2] += cb->ns_waiters.sum(0);
    out[3] += cb->ns_waiters.sum(1);
    out[4] += cb->ns_launch_calls.load();
    out[5] += cb->ns_waiters.sum(2);
  }
}

extern "C" int32_t msi_bits_vm_stats(msi_bits *pool, uint64_t out[6]) {
  if (!pool || !out) return MSI_E_INVALID;
  msi_vm_stats(pool, out);
  return MSI_OK;
}

int32_t msi_vm_record_decode(MsiVmList &l, msi_bits *pool, uint32_t dst, const MsiCboBatch &batch, bool overwrite) {
  // Bodies that are not in the posting cache go into the pool's pinned staging buffer (read over PCIe by the decoding
  // workgroup, once); the container descriptors are bucketed by chunk (= Roaring key) here and laid out chunk-major
  // when the list is submitted.  The <= 7-document raw values become array containers built here.
  // (a compact list stores ranks, but its decodes read the postings by docid: chunks of the FULL space)
  const uint64_t n_words = msi_bits_words_per_slot(l.geom_docs ? l.full_pool : pool);
  const uint32_t n_chunks = (uint32_t)((n_words + CHW - 1) / CHW);
  std::vector<std::pair<uint32_t, uint16_t>> small;
  small.reserve(batch.small_ids.size());
  for (uint32_t id : batch.small_ids)
    if ((id >> 16) < n_chunks) small.push_back({id >> 16, (uint16_t)(id & 0xFFFF)});
  std::sort(small.begin(), small.end());
  std::vector<MsiContainer> extra;
  std::vector<uint8_t> extra_bytes;
  for (size_t i = 0; i < small.size();) {
    size_t j = i;
    MsiContainer c;
    c.key = small[i].first;
    c.type = 0;
    c.offset = (uint32_t)extra_bytes.size();
    while (j < small.size() && small[j].first == c.key) {
      extra_bytes.push_back((uint8_t)(small[j].second & 0xFF));
      extra_bytes.push_back((uint8_t)(small[j].second >> 8));
      ++j;
    }
    c.card = (uint32_t)(j - i);
    extra.push_back(c);
    i = j;
  }
  // staging: [batch.bytes (16-aligned start)] [extra_bytes (16-aligned start)]
  const size_t at = align16(l.stage_used);
  const size_t extra_at = at + align16(batch.bytes.size());
  const size_t total = extra_at + extra_bytes.size();
  if (total > at) {
    uint8_t *stage = msi_bits_vm_stage(pool, total + 16, l.stage_used);
    if (!stage) return MSI_E_OOM;
    if (!batch.bytes.empty()) memcpy(stage + at, batch.bytes.data(), batch.bytes.size());
    if (!extra_bytes.empty()) memcpy(stage + extra_at, extra_bytes.data(), extra_bytes.size());
    l.stage_used = total;
  }
  MsiVmList::Decode dec;
  dec.start.assign(n_chunks + 1, 0);
  for (const MsiContainer &c : batch.containers)
    if (c.key < n_chunks) ++dec.start[c.key + 1];
  for (const MsiContainer &c : extra) ++dec.start[c.key + 1];
  for (uint32_t k = 0; k < n_chunks; ++k) dec.start[k + 1] += dec.start[k];
  dec.c.assign((size_t)dec.start[n_chunks] * 2, 0);
  std::vector<uint32_t> fill(dec.start.begin(), dec.start.end() - 1);
  const bool has_cache = !batch.src.empty();
  auto put = [&](const MsiContainer &c, bool cached, uint64_t src, uint64_t fill_off) {
    // meta: card - 1 (16 bits) | type << 16 | cached << 18 | fill offset high 13 bits << 19; then fill offset low 32 bits
    const uint32_t card1 = c.type == 1 ? 0u : (c.card ? c.card - 1 : 0u);
    const uint64_t f = fill_off == MSI_NO_CACHE ? ((uint64_t)0x1FFF << 32 | 0xFFFFFFFFull) : fill_off;
    if (fill_off != MSI_NO_CACHE) l.any_fill = true;
    const uint32_t meta = (card1 & 0xFFFFu) | (c.type << 16) | ((cached ? 1u : 0u) << 18) | ((uint32_t)((f >> 32) & 0x1FFF) << 19);
    const size_t at2 = (size_t)fill[c.key]++ * 2;
    dec.c[at2] = (uint64_t)meta | ((uint64_t)(uint32_t)f << 32);
    dec.c[at2 + 1] = src;
  };
  for (size_t i = 0; i < batch.containers.size(); ++i) {
    const MsiContainer &c = batch.containers[i];
    if (c.key >= n_chunks) continue;
    const bool cached = has_cache && batch.src[i] != MSI_NO_CACHE;
    put(c, cached, cached ? batch.src[i] : (uint64_t)at + c.offset, has_cache ? batch.fill[i] : MSI_NO_CACHE);
  }
  for (const MsiContainer &c : extra) put(c, false, (uint64_t)extra_at + c.offset, MSI_NO_CACHE);
  l.begin();
  if (l.geom_docs) {
    if (!overwrite) return MSI_E_UNSUPPORTED;   // (the search only decodes into fresh slots)
    l.pre.push_back(VM_DECODEC);
    l.pre.push_back(dst);
    l.pre.push_back((uint32_t)l.decodes.size());
  } else {
    l.words.push_back(VM_DECODE);
    l.words.push_back(dst);
    l.words.push_back(overwrite ? 1u : 0u);
    l.words.push_back((uint32_t)l.decodes.size());
  }
  l.decodes.push_back(std::move(dec));
  return MSI_OK;
}

void msi_vm_record_compact(MsiVmList &l, uint32_t dst, uint32_t full_slot) {
  l.begin();
  l.pre.push_back(VM_DECODEC);
  l.pre.push_back(dst);
  l.pre.push_back(0x80000000u | full_slot);
}

bool msi_vm_record_rank(MsiVmList &l, msi_bits *pool, uint32_t slot) {
  // (the tables are allocated when a pool first needs them: with HBM exhausted that fails, and commands that write through a
  // null table fault the device — a run of 384 callers beside the C4 store dumped a GPU core here, profiles/r6_step_callers_and_slots.log)
  const uint64_t aux = (uint64_t)(uintptr_t)msi_bits_compact_aux(pool);
  if (!aux) return false;
  l.begin();
  l.words.insert(l.words.end(), {(uint32_t)VM_RANK_A, slot, (uint32_t)aux, (uint32_t)(aux >> 32)});
  l.barrier();
  const uint64_t cap = msi_bits_compact_capacity(pool);
  l.words.insert(l.words.end(), {(uint32_t)VM_RANK_B, slot, (uint32_t)aux, (uint32_t)(aux >> 32), (uint32_t)std::min<uint64_t>(cap, 0xFFFFFFFFull)});
  return true;
}

// A compact list's hoisted VM_DECODEC commands become its phase 0 (the wide phase).
static void merge_pre(MsiVmList &l) {
  if (l.pre.empty() || l.pre_merged) return;
  const uint32_t shift = (uint32_t)l.pre.size() + 1;
  std::vector<uint32_t> w;
  w.reserve(l.pre.size() + 1 + l.words.size());
  w.insert(w.end(), l.pre.begin(), l.pre.end());
  w.push_back(VM_END);
  w.insert(w.end(), l.words.begin(), l.words.end());
  l.words.swap(w);
  if (l.phase_start.empty()) l.phase_start.push_back(0);
  for (uint32_t &x : l.phase_start) x += shift;
  l.phase_start.insert(l.phase_start.begin(), 0u);
  l.pre_merged = true;
}

// Chunk-major layout of the list's decode descriptors, appended to its words (16-byte aligned):
//   chunk_off[n_chunks + 1]  (16-byte units from the block start)
//   per chunk c: start[n_decodes + 1] (containers of decode d in c: start[d] .. start[d+1]), padded to 16 bytes,
//                then those containers, 16 bytes each, decodes back to back
static void finalize_decodes(MsiVmList &l) {
  if (l.decodes.empty() || l.data_off) return;
  const uint32_t D = (uint32_t)l.decodes.size();
  const uint32_t n_chunks = (uint32_t)l.decodes[0].start.size() - 1;
  l.words.push_back(VM_END);                       // the last phase ends here; what follows is data
  while (l.words.size() % 4) l.words.push_back(0);
  l.data_off = (uint32_t)l.words.size();
  const uint32_t hdr_words = (n_chunks + 1 + 3) & ~3u, st_words = (D + 1 + 3) & ~3u;
  std::vector<uint32_t> chunk_off(n_chunks + 1, 0);
  uint32_t at = hdr_words / 4;                     // 16-byte units
  for (uint32_t c = 0; c < n_chunks; ++c) {
    chunk_off[c] = at;
    uint32_t n = 0;
    for (const auto &d : l.decodes) n += d.start[c + 1] - d.start[c];
    at += st_words / 4 + n;
  }
  chunk_off[n_chunks] = at;
  const size_t base = l.words.size();
  l.words.resize(base + (size_t)at * 4, 0);
  uint32_t *w = l.words.data() + base;
  memcpy(w, chunk_off.data(), (n_chunks + 1) * 4);
  for (uint32_t c = 0; c < n_chunks; ++c) {
    uint32_t *blk = w + (size_t)chunk_off[c] * 4;
    uint64_t *cs = reinterpret_cast<uint64_t *>(blk + st_words);
    uint32_t run = 0;
    for (uint32_t d = 0; d < D; ++d) {
      const auto &dec = l.decodes[d];
      blk[d] = run;
      const uint32_t n = dec.start[c + 1] - dec.start[c];
      if (n) memcpy(cs + (size_t)run * 2, dec.c.data() + (size_t)dec.start[c] * 2, (size_t)n * 16);
      run += n;
    }
    blk[D] = run;
  }
}


// Algorithmic bytes of a list: every set operand of every command, whole (slot words x 8), plus the container bodies its
// decodes read and — compact lists — U0's words and prefix counts once per decode.  What the commands ASK the memory
// system for; msi_bits_vm_bytes hands the sums out for the keyword leg's roofline object (bench.py).
static StripedCounters<3> g_vm_bytes;   // [set operands, posting containers, lists]
// MSI_VM_BYTES_BY_OP=<file>: the same model by command and space (full / compact), written when the process ends:
// [space][command] -> commands, sets they sweep, bytes (sets x the list's set size) — where a query's set traffic comes from
struct BytesByOp {
  std::mutex mu;
  uint64_t n[2][32] = {}, sets[2][32] = {}, bytes[2][32] = {}, lists[2] = {}, list_bytes[2] = {};
  const char *path = getenv("MSI_VM_BYTES_BY_OP");
  ~BytesByOp() {
    if (!path) return;
    FILE *f = fopen(path, "w");
    if (!f) return;
    static const char *names[32] = {"end", "fill", "op", "op_count", "clear", "claim", "and_many", "paths", "sub_many", "count", "decode",
                                    "firstk", "minkey", "takekey", "summary_reset", "rank_a", "rank_b", "decodec"};
    for (int sp = 0; sp < 2; ++sp) {
      fprintf(f, "%s lists %llu bytes %llu\n", sp ? "compact" : "full", (unsigned long long)lists[sp], (unsigned long long)list_bytes[sp]);
      for (int op = 0; op < 32; ++op)
        if (n[sp][op])
          fprintf(f, "  %-14s commands %10llu sets %10llu bytes %14llu\n", names[op] ? names[op] : "?", (unsigned long long)n[sp][op],
                  (unsigned long long)sets[sp][op], (unsigned long long)bytes[sp][op]);
    }
    fclose(f);
  }
};
static BytesByOp g_by_op;

static void account_list(msi_bits *pool, const MsiVmList &l) {
  const uint64_t words = l.geom_docs ? std::max<uint64_t>(2, ((l.geom_docs + 127) / 128) * 2) : msi_bits_words_per_slot(pool);
  const uint64_t set_b = words * 8;
  uint64_t sets = 0, wide = 0;
  const std::vector<uint32_t> &w = l.words;
  const bool by_op = g_by_op.path != nullptr;
  uint64_t op_n[32] = {}, op_sets[32] = {};
  for (size_t i = 0; i < w.size();) {
    const uint32_t op_now = w[i];
    const uint64_t sets_before = sets;
    struct Tally {   // (every way out of the switch)
      bool on; uint64_t *n, *s; const uint64_t &sets, &before; uint32_t op;
      ~Tally() { if (on && op < 32) { ++n[op]; s[op] += sets - before; } }
    } tally{by_op, op_n, op_sets, sets, sets_before, op_now};
    switch (w[i]) {
      case VM_END: i += 1; break;
      case VM_FILL: sets += 1; i += 3; break;
      case VM_OP: sets += 3; i += 5; break;
      case VM_OP_COUNT: sets += 3; i += 6; break;
      case VM_CLEAR: sets += w[i + 1]; i += 2 + w[i + 1]; break;
      case VM_CLAIM: sets += 5 + 2 * w[i + 4]; i += 5 + w[i + 4]; break;
      case VM_AND_MANY: sets += 1 + 2 * w[i + 2]; i += 4 + 2 * w[i + 2]; break;
      case VM_PATHS: sets += 4 + (w[i + 5] & 0x7FFFFFFFu); i += 7 + w[i + 1] + (w[i + 5] & 0x7FFFFFFFu); break;
      case VM_SUB_MANY: sets += 1 + 2 * w[i + 2]; i += 4 + w[i + 2]; break;
      case VM_COUNT: sets += 1; i += 3; break;
      case VM_DECODE: sets += 1; i += 4; break;
      case VM_FIRSTK: sets += 1; i += 5; break;
      case VM_MINKEY: sets += 1; i += 5; break;
      case VM_TAKEKEY: sets += 3; i += 8; break;
      case VM_SUMMARY_RESET: i += 1; break;
      case VM_RANK_A: sets += 1; i += 4; break;
      case VM_RANK_B: sets += 2; i += 5; break;   // U0 again + the tables (4 B per word + 4 B per document: about one set)
      case VM_DECODEC: sets += 1; wide += 1; i += 3; break;
      default: i = w.size(); break;
    }
  }
  uint64_t posting = 0;
  for (const auto &d : l.decodes)
    for (size_t c = 0; c + 1 < d.c.size(); c += 2) {
      const uint32_t meta = (uint32_t)d.c[c], type = (meta >> 16) & 3u, card = meta & 0xFFFFu;
      posting += type == 0 ? 2 * (card + 1) : (type == 1 ? 8192u : 4 * (card + 1));
    }
  uint64_t total = sets * set_b;
  // (a wide phase stages U0's words and prefix counts once per workgroup, whatever the number of its decodes)
  if (l.geom_docs && l.full_pool && wide) total += msi_bits_words_per_slot(l.full_pool) * 12;
  g_vm_bytes.add(0, total);
  g_vm_bytes.add(1, posting);
  g_vm_bytes.add(2, 1);
  if (by_op) {
    std::lock_guard<std::mutex> lk(g_by_op.mu);
    const int sp = l.geom_docs ? 1 : 0;
    ++g_by_op.lists[sp];
    g_by_op.list_bytes[sp] += total;
    for (int op = 0; op < 32; ++op) {
      g_by_op.n[sp][op] += op_n[op];
      g_by_op.sets[sp][op] += op_sets[op];
      g_by_op.bytes[sp][op] += op_sets[op] * set_b;
    }
  }
}
extern "C" int32_t msi_bits_vm_bytes(uint64_t out[3]) {
  if (!out) return MSI_E_INVALID;
  for (int i = 0; i < 3; ++i) out[i] = g_vm_bytes.sum(i);
  return MSI_OK;
}

static StripedCounters<8> g_cpu_prof;
static std::atomic<int> g_cpu_prof_switch{-1};   // -1: the environment decides (MSI_SEARCH_CPU_PROFILE), 0 / 1: msi_search_cpu_profile_enable
bool msi_cpu_prof_on() {
  static const bool env_on = getenv("MSI_SEARCH_CPU_PROFILE") != nullptr;
  const int sw = g_cpu_prof_switch.load(std::memory_order_relaxed);
  return sw < 0 ? env_on : sw != 0;
}
extern "C" int32_t msi_search_cpu_profile_enable(int32_t on) {
  g_cpu_prof_switch.store(on ? 1 : 0, std::memory_order_relaxed);
  return MSI_OK;
}
uint64_t msi_thread_cpu_ns() {
  struct timespec ts;
  clock_gettime(CLOCK_THREAD_CPUTIME_ID, &ts);
  return (uint64_t)ts.tv_sec * 1000000000ull + (uint64_t)ts.tv_nsec;
}
void msi_cpu_prof_add(int idx, uint64_t ns) { g_cpu_prof.add(idx, ns); }
extern "C" int32_t msi_search_cpu_profile(uint64_t out[8]) {
  if (!out) return MSI_E_INVALID;
  for (int i = 0; i < 8; ++i) out[i] = g_cpu_prof.sum(i);
  return MSI_OK;
}

int32_t msi_vm_run(msi_bits *pool, MsiVmList &l, MsiVmResult *res) {
  const bool cpu_prof = msi_cpu_prof_on();
  const uint64_t cpu0 = cpu_prof ? msi_thread_cpu_ns() : 0;
  struct CpuProf {   // (every way out of this function)
    bool on;
    uint64_t t0;
    ~CpuProf() {
      if (on) {
        msi_cpu_prof_add(2, msi_thread_cpu_ns() - t0);
        msi_cpu_prof_add(7, 1);
      }
    }
  } cpu_guard{cpu_prof, cpu0};
  merge_pre(l);
  account_list(pool, l);
  finalize_decodes(l);
  if (cpu_prof) msi_cpu_prof_add(3, msi_thread_cpu_ns() - cpu0);
  if (l.n_counts > MSI_VM_MAX_COUNTS || l.phase_start.size() > MSI_VM_MAX_PHASES || l.phase_start.empty()) {
    msi_set_error("msi_vm_run: list outside the supported range (%u counts, %zu phases)", l.n_counts, l.phase_start.size());
    return MSI_E_UNSUPPORTED;
  }
  msi_ctx *ctx = msi_bits_ctx(pool);
  msi_vm *vmx = vm_of(ctx);
  if (!vmx) return MSI_E_HIP;
  VmCombiner *vm = vmx->comb[(((uintptr_t)pool) >> 8) % vmx->comb.size()].get();   // a pool always talks to the same combiner
  volatile uint64_t *blk = msi_bits_vm_block(pool);
  if (!blk) return MSI_E_OOM;
  // one submission record per calling thread (a thread has one list in flight at a time).  It is never freed: the combiner
  // may still read `asleep` of a record whose waiter has already left (finish()), so the memory has to outlive the thread.
  static thread_local VmSub *tl_sub = nullptr;
  if (!tl_sub) tl_sub = new VmSub();
  VmSub *s = tl_sub;
  s->pool = pool;
  s->list = &l;
  s->seq = msi_bits_vm_next_seq(pool);
  s->blk = blk;
  s->error = MSI_OK;
  s->errmsg[0] = 0;
  s->fused_wgs = 0;
  s->state.store(0, std::memory_order_relaxed);
  s->asleep.store(0, std::memory_order_relaxed);
  s->t_submit = now_ns();
  s->t_taken = s->t_launch = s->t_done = 0;
  vm->load.fetch_add(1, std::memory_order_relaxed);
  s->next = vm->q_head.load(std::memory_order_relaxed);
  while (!vm->q_head.compare_exchange_weak(s->next, s, std::memory_order_seq_cst, std::memory_order_relaxed)) {
  }
  if (vm->sleeping.load(std::memory_order_seq_cst)) {
    { std::lock_guard<std::mutex> lk(vm->mu); }   // (the combiner is inside cv.wait once this lock is granted)
    vm->cv.notify_one();
  }
  // A short poll when the combiner is nearly idle (a round trip is then ~25 us), else sleep until the combiner wakes
  // us: under load a round trip takes 100+ us and polling through it would burn the CPU time the searches need.
  const int64_t t0 = now_ns();
  const int64_t spin_ns = vm->load.load(std::memory_order_relaxed) <= 3 ? 30000 : 0;
  uint32_t st;
  while ((st = s->state.load(std::memory_order_acquire)) == 0) {
    if (now_ns() - t0 < spin_ns) {
#if defined(__x86_64__)
      __builtin_ia32_pause();
#endif
      continue;
    }
    s->asleep.store(1, std::memory_order_seq_cst);
    if ((st = s->state.load(std::memory_order_seq_cst)) != 0) break;   // completed between the poll and the flag
    futex_wait_for(&s->state, 0, 2000);
  }
  int32_t ret = MSI_OK;
  if (st == 2) {
    ret = s->error != MSI_OK ? s->error : MSI_E_INTERNAL;
    if (ret == MSI_E_INTERNAL) msi_set_error("msi_vm: a round finished without publishing its results");
    else msi_set_error("%s", s->errmsg[0] ? s->errmsg : "msi_vm: the round of this list failed");
  } else if (__atomic_load_n(const_cast<uint64_t *>(&blk[1]), __ATOMIC_RELAXED) == MSI_VM_RES_FAILED) {
    ret = MSI_E_INTERNAL;
    msi_set_error("msi_vm: the workgroups of a fused list gave up waiting for the list's wide phase (a stalled device?)");
  } else {
    vm->ns_waiters.add(0, (uint64_t)(s->t_taken - s->t_submit));
    vm->ns_waiters.add(1, (uint64_t)(s->t_launch - s->t_taken));
    vm->ns_waiters.add(2, (uint64_t)(s->t_done - s->t_launch));
    if (res) {
      res->counts.resize(l.n_counts);
      for (uint32_t i = 0; i < l.n_counts; ++i)
        res->counts[i] = __atomic_load_n(const_cast<uint64_t *>(&blk[RES_COUNTS + i]), __ATOMIC_RELAXED);
      res->firstk.clear();
      if (l.firstk_total) {   // every first-k command's block of k ids, back to back (a set smaller than k fills less)
        const uint32_t *ids = reinterpret_cast<const uint32_t *>(const_cast<const uint64_t *>(blk) + RES_IDS);
        res->firstk.assign(ids, ids + std::min<uint32_t>(l.firstk_total, MSI_VM_MAX_FIRSTK));
      }
    }
  }
  vm->load.fetch_sub(1, std::memory_order_relaxed);
  return ret;
}


// ================================================================================================ posting cache

namespace {
struct KeyHash {
  size_t operator()(const MsiCacheKey &k) const { return (size_t)(k.a ^ (k.b * 0x9E3779B97F4A7C15ull)); }
};
struct KeyEq {
  bool operator()(const MsiCacheKey &x, const MsiCacheKey &y) const { return x.a == y.a && x.b == y.b; }
};
struct CacheEntry {
  MsiCacheKey key{};
  uint64_t off = 0;
  uint64_t len = 0;
  // what msi_pcache_known hands out (written before `ready` / `host_kind` is released, never changed afterwards)
  std::atomic<uint32_t> host_kind{0};   // 0 nothing, 1 absent, 2 small ids (no body in HBM: off / len unused)
  uint64_t card = 0;
  std::vector<MsiContainer> conts;
  std::vector<uint32_t> small;
  std::atomic<uint32_t> ready{0};   // 0: reserved, being filled | 1: filled | 2: abandoned by a list that failed or was
                                    //    dropped — the next reader of the key takes the reservation over
  bool staged = false;              // put there by msi_pcache_stage (index-open): survives msi_pcache_reset
};
inline uint64_t mix64(uint64_t x) {
  x ^= x >> 33; x *= 0xFF51AFD7ED558CCDull; x ^= x >> 33; x *= 0xC4CEB9FE1A85EC53ull; x ^= x >> 33;
  return x;
}
}  // namespace

struct MsiPostingCache {
  msi_ctx *ctx = nullptr;
  uint8_t *dev = nullptr;
  uint64_t cap = 0;
  std::atomic<uint64_t> used{0};     // bump allocation of the HBM arena
  // The host table: key -> entry.  A search asks ~135 keys and 160 searches ask at once.  Round 3 sharded one reader-writer
  // lock 64 ways; on a stream of fresh queries (round 5) the READ locks were still a fifth of the keyword leg's host CPU
  // (pthread_rwlock_rdlock / unlock: every reader writes the lock word, the line bounces between 13 busy CPUs) and the per-shard
  // hit counters did the same.  Readers now write nothing shared: each shard is an open-addressing table of atomic pointers
  // to entries that never move (an entry is published with a release store once it is initialised; a table that fills up is
  // replaced by one of twice the size and RETIRED, not freed, so a reader that still probes it stays safe — it may miss a key
  // inserted meanwhile, which sends it down the writers' path where the current table is probed again under the shard's
  // mutex).  Entries are removed only with the cache (msi_dict_reset_posting_cache: no search in flight).  Counters are
  // striped by thread.
  static constexpr uint32_t SHARDS = 64;
  struct Table {
    uint64_t mask = 0;
    std::atomic<CacheEntry *> *slots = nullptr;
    explicit Table(uint64_t cap) : mask(cap - 1), slots(new std::atomic<CacheEntry *>[cap]) {
      for (uint64_t i = 0; i < cap; ++i) slots[i].store(nullptr, std::memory_order_relaxed);
    }
    ~Table() { delete[] slots; }
  };
  struct alignas(64) Shard {
    std::atomic<Table *> table{nullptr};
    std::mutex wmu;                      // a key's first appearance (and table growth)
    uint64_t count = 0;
    std::vector<Table *> retired;
    std::vector<CacheEntry *> entries;   // owned
  } shard[SHARDS];
  StripedCounters<2> counters;           // [hits, misses]
  // staging at index-open (msi_pcache_stage): what was staged, and which databases of which view are complete
  std::atomic<uint64_t> staged_bodies{0}, staged_host{0}, staged_bytes{0}, staged_hi{16};
  StripedCounters<1> complete_answers;
  struct Complete {
    std::atomic<uint64_t> view{0};
    std::atomic<uint32_t> mask{0};
  } complete[8];
  std::atomic<uint32_t> n_complete{0};
  std::mutex complete_mu;
  Shard &of(const MsiCacheKey &k) { return shard[(k.b >> 7) % SHARDS]; }
  static uint64_t slot_of(const MsiCacheKey &k) { return (k.a ^ (k.a >> 31)) * 0x9E3779B97F4A7C15ull >> 20; }
  static CacheEntry *probe(const Table *t, const MsiCacheKey &k) {
    for (uint64_t i = slot_of(k) & t->mask;; i = (i + 1) & t->mask) {
      CacheEntry *e = t->slots[i].load(std::memory_order_acquire);
      if (!e) return nullptr;
      if (e->key.a == k.a && e->key.b == k.b) return e;
    }
  }
  // readers: no lock, no shared write
  CacheEntry *find(const MsiCacheKey &k) {
    const Table *t = of(k).table.load(std::memory_order_acquire);
    return t ? probe(t, k) : nullptr;
  }
  // writers, under sh.wmu: the entry of k, created and initialised by `init` (before it becomes visible) when it is new
  template <typename Init>
  CacheEntry *find_or_insert(Shard &sh, const MsiCacheKey &k, bool *created, Init init) {
    Table *t = sh.table.load(std::memory_order_relaxed);
    *created = false;
    if (t) {
      if (CacheEntry *e = probe(t, k)) return e;
    }
    if (!t || (sh.count + 1) * 2 > t->mask + 1) {
      Table *nt = new Table(t ? (t->mask + 1) * 2 : 1024);
      if (t) {
        for (uint64_t i = 0; i <= t->mask; ++i) {
          CacheEntry *e = t->slots[i].load(std::memory_order_relaxed);
          if (!e) continue;
          uint64_t j = slot_of(e->key) & nt->mask;
          while (nt->slots[j].load(std::memory_order_relaxed)) j = (j + 1) & nt->mask;
          nt->slots[j].store(e, std::memory_order_relaxed);
        }
        sh.retired.push_back(t);
      }
      sh.table.store(nt, std::memory_order_release);
      t = nt;
    }
    CacheEntry *e = new CacheEntry();
    e->key = k;
    if (!init(*e)) {   // (nothing to remember: e.g. the HBM arena is full)
      delete e;
      return nullptr;
    }
    sh.entries.push_back(e);
    ++sh.count;
    uint64_t j = slot_of(k) & t->mask;
    while (t->slots[j].load(std::memory_order_relaxed)) j = (j + 1) & t->mask;
    t->slots[j].store(e, std::memory_order_release);
    *created = true;
    return e;
  }
  ~MsiPostingCache() {
    for (auto &sh : shard) {
      for (CacheEntry *e : sh.entries) delete e;
      for (Table *t : sh.retired) delete t;
      delete sh.table.load();
    }
  }
};

// Two independent 64-bit hashes over (database tag, the two strings with their lengths, two integers): a collision
// needs both to agree (2^-128 per pair of keys); the serialisation length is checked on top of it.
MsiCacheKey msi_cache_key(uint32_t db, const void *s1, size_t n1, const void *s2, size_t n2, uint64_t x, uint64_t y,
                          uint64_t view) {
  uint64_t a = 0xCBF29CE484222325ull ^ db, b = 0x84222325CBF29CE4ull + (uint64_t)db * 0x100000001B3ull;
  auto feed = [&](const void *s, size_t n) {
    const uint8_t *p = (const uint8_t *)s;
    a = (a ^ n) * 0x100000001B3ull;
    b = mix64(b + n);
    for (size_t i = 0; i < n; ++i) {
      a = (a ^ p[i]) * 0x100000001B3ull;
      b = (b + p[i]) * 0x9E3779B97F4A7C15ull;
      b ^= b >> 29;
    }
  };
  feed(s1, n1);
  feed(s2, n2);
  a = mix64(a ^ mix64(x + 0x1234567ull)) ^ mix64(y * 0xD6E8FEB86659FD93ull + 1);
  b = mix64(b ^ mix64(y + 0x7654321ull)) ^ mix64(x * 0xA0761D6478BD642Full + 3);
  if (view) {   // (0 leaves the keys of the plain index as they were)
    a = mix64(a ^ mix64(view + 0x51ED27ull));
    b = mix64(b + mix64(view * 0x9E3779B97F4A7C15ull + 5));
  }
  return MsiCacheKey{a, b};
}

MsiPostingCache *msi_pcache_create(msi_ctx *ctx, uint64_t capacity_bytes) {
  if (!ctx || capacity_bytes < 4096) return nullptr;
  DeviceGuard g(ctx->device);
  void *d = nullptr;
  if (hipMalloc(&d, capacity_bytes) != hipSuccess) {
    msi_set_error("hipMalloc(%llu) for the posting cache failed", (unsigned long long)capacity_bytes);
    return nullptr;
  }
  MsiPostingCache *c = new MsiPostingCache();
  c->ctx = ctx;
  c->dev = (uint8_t *)d;
  c->cap = capacity_bytes;
  c->used.store(16);   // offset 0 stays unused
  return c;
}

void msi_pcache_destroy(MsiPostingCache *c) {
  if (!c) return;
  DeviceGuard g(c->ctx->device);
  (void)hipDeviceSynchronize();   // no list that reads or fills the cache is in flight after this
  (void)hipFree(c->dev);
  delete c;
}

int msi_pcache_lookup(MsiPostingCache *c, const MsiCacheKey &k, size_t len, uint64_t *off, void **token) {
  *token = nullptr;
  auto existing = [&](CacheEntry *e) -> int {
    const uint32_t state = e->ready.load(std::memory_order_acquire);
    if (e->host_kind.load(std::memory_order_acquire) == 0 && e->len == len && state == 1) {
      *off = e->off;
      c->counters.add(0, 1);
      return 1;
    }
    c->counters.add(1, 1);
    uint32_t abandoned = 2;
    if (e->host_kind.load(std::memory_order_acquire) == 0 && e->len == len && state == 2 &&
        e->ready.compare_exchange_strong(abandoned, 0, std::memory_order_acq_rel)) {
      *off = e->off;    // the reservation of a list that never ran: this caller fills it
      *token = e;
      return 2;
    }
    return 0;   // being filled by another search (or a length mismatch: never trusted)
  };
  if (CacheEntry *e = c->find(k)) return existing(e);
  MsiPostingCache::Shard &sh = c->of(k);
  std::lock_guard<std::mutex> lk(sh.wmu);
  bool created = false;
  CacheEntry *e = c->find_or_insert(sh, k, &created, [&](CacheEntry &ne) {
    const uint64_t need = ((uint64_t)len + 15 + 16) & ~15ull;   // + one block of slack for the last partial block
    uint64_t at = c->used.load(std::memory_order_relaxed);
    do {
      if (at + need > c->cap) return false;
    } while (!c->used.compare_exchange_weak(at, at + need, std::memory_order_relaxed));
    ne.off = at;
    ne.len = len;
    return true;
  });
  c->counters.add(1, 1);
  if (!e || !created) return 0;   // (somebody else's entry appeared meanwhile: it is being filled; or the arena is full)
  *off = e->off;
  *token = e;   // entries never move
  return 2;
}

void msi_pcache_commit(MsiPostingCache *, void *token) {
  if (token) static_cast<CacheEntry *>(token)->ready.store(1, std::memory_order_release);
}

bool msi_cbo_parse(const uint8_t *bytes, size_t len, std::vector<MsiContainer> &out);
uint64_t msi_cbo_cardinality(const uint8_t *bytes, size_t len);

bool msi_pcache_known(MsiPostingCache *c, const MsiCacheKey &k, MsiKnownPosting *out) {
  const CacheEntry *ep = c->find(k);
  if (!ep) return false;
  const CacheEntry &e = *ep;
  const uint32_t hk = e.host_kind.load(std::memory_order_acquire);
  if (hk == 1 || hk == 2) {
    out->kind = (int)hk;
    out->off = out->len = 0;
    out->card = e.card;
    out->conts = nullptr;
    out->n_conts = 0;
    out->small = e.small.data();
    out->n_small = (uint32_t)e.small.size();
    c->counters.add(0, 1);
    return true;
  }
  if (e.ready.load(std::memory_order_acquire) != 1 || e.conts.empty()) return false;
  out->kind = 3;
  out->off = e.off;
  out->len = e.len;
  out->card = e.card;
  out->conts = e.conts.data();
  out->n_conts = (uint32_t)e.conts.size();
  out->small = nullptr;
  out->n_small = 0;
  c->counters.add(0, 1);
  return true;
}

// the index answered "no such key" (len 0) or a raw value of <= 7 docids: remembered on the host
void msi_pcache_learn(MsiPostingCache *c, const MsiCacheKey &k, const uint8_t *bytes, size_t len) {
  if (len > 7 * sizeof(uint32_t)) return;
  if (c->find(k)) return;   // (known keys are the rule: no lock for them)
  MsiPostingCache::Shard &sh = c->of(k);
  std::lock_guard<std::mutex> lk(sh.wmu);
  bool created = false;
  (void)c->find_or_insert(sh, k, &created, [&](CacheEntry &e) {
    for (size_t i = 0; i + 4 <= len; i += 4) {
      uint32_t v;
      memcpy(&v, bytes + i, 4);
      e.small.push_back(v);
    }
    e.card = e.small.size();
    e.host_kind.store(e.small.empty() ? 1u : 2u, std::memory_order_release);
    return true;
  });
}

// the reserved entry's container table, parsed once from the bytes its first reader holds (before msi_pcache_commit)
void msi_pcache_describe(MsiPostingCache *, void *token, const uint8_t *bytes, size_t len) {
  if (!token) return;
  CacheEntry *e = static_cast<CacheEntry *>(token);
  if (!e->conts.empty()) return;
  std::vector<MsiContainer> cs;
  if (!msi_cbo_parse(bytes, len, cs)) return;
  e->card = msi_cbo_cardinality(bytes, len);
  e->conts.swap(cs);
}

// ---- staging at index-open ------------------------------------------------------------------------------------------
int32_t msi_pcache_stage(MsiPostingCache *c, const MsiStageValue *values, uint64_t n, uint64_t out[3]) {
  if (out) out[0] = out[1] = out[2] = 0;
  if (!c || (!values && n)) return MSI_E_INVALID;
  auto need_of = [](size_t len) { return ((uint64_t)len + 15 + 16) & ~15ull; };   // (as msi_pcache_lookup reserves)
  // the call's bodies: parsed first (a malformed value fails the call before anything is reserved), one reservation
  struct Body {
    uint64_t rel;
    std::vector<MsiContainer> conts;
    uint64_t card;
  };
  std::vector<Body> bodies(n);
  uint64_t total = 0;
  for (uint64_t i = 0; i < n; ++i) {
    const MsiStageValue &v = values[i];
    if (!v.bytes || v.len <= 7 * sizeof(uint32_t)) continue;
    if (!msi_cbo_parse(v.bytes, v.len, bodies[i].conts) || bodies[i].conts.empty()) {
      msi_set_error("msi_dict_stage_postings: value %llu (%zu bytes) is not a CboRoaringBitmap serialisation", (unsigned long long)i, v.len);
      return MSI_E_INVALID;
    }
    bodies[i].card = msi_cbo_cardinality(v.bytes, v.len);
    if (i && values[i - 1].bytes == v.bytes && values[i - 1].len == v.len) {   // the same stored value under two keys: one body
      bodies[i].rel = bodies[i - 1].rel;
      continue;
    }
    bodies[i].rel = total;
    total += need_of(v.len);
  }
  uint64_t base = 0;
  if (total) {
    base = c->used.load(std::memory_order_relaxed);
    do {
      if (base + total > c->cap) {
        msi_set_error("msi_dict_stage_postings: %llu bytes of postings do not fit the posting cache (%llu of %llu bytes used)",
                      (unsigned long long)total, (unsigned long long)base, (unsigned long long)c->cap);
        return MSI_E_OOM;
      }
    } while (!c->used.compare_exchange_weak(base, base + total, std::memory_order_relaxed));
    std::unique_ptr<uint8_t[]> host(new uint8_t[total]);
    for (uint64_t i = 0; i < n; ++i) {
      if (bodies[i].conts.empty()) continue;
      memcpy(host.get() + bodies[i].rel, values[i].bytes, values[i].len);
      memset(host.get() + bodies[i].rel + values[i].len, 0, need_of(values[i].len) - values[i].len);
    }
    DeviceGuard g(c->ctx->device);
    // (a blocking copy: the bodies are in HBM before any entry that points at them becomes visible)
    if (hipMemcpy(c->dev + base, host.get(), total, hipMemcpyHostToDevice) != hipSuccess) {
      msi_set_error("msi_dict_stage_postings: copying %llu bytes to the device failed", (unsigned long long)total);
      return MSI_E_HIP;
    }
    uint64_t hi = c->staged_hi.load(std::memory_order_relaxed);
    while (hi < base + total && !c->staged_hi.compare_exchange_weak(hi, base + total, std::memory_order_relaxed)) {
    }
  }
  uint64_t n_body = 0, n_host = 0, n_known = 0, bytes_body = 0;
  for (uint64_t i = 0; i < n; ++i) {
    const MsiStageValue &v = values[i];
    if (c->find(v.key)) {
      ++n_known;
      continue;
    }
    MsiPostingCache::Shard &sh = c->of(v.key);
    std::lock_guard<std::mutex> lk(sh.wmu);
    bool created = false;
    (void)c->find_or_insert(sh, v.key, &created, [&](CacheEntry &e) {
      e.staged = true;
      if (bodies[i].conts.empty()) {   // absent, or a raw value of <= 7 docids: kept on the host (msi_pcache_learn)
        const size_t len = v.bytes ? v.len : 0;
        for (size_t b = 0; b + 4 <= len; b += 4) {
          uint32_t d;
          memcpy(&d, v.bytes + b, 4);
          e.small.push_back(d);
        }
        e.card = e.small.size();
        e.host_kind.store(e.small.empty() ? 1u : 2u, std::memory_order_release);
        return true;
      }
      e.off = base + bodies[i].rel;
      e.len = v.len;
      e.card = bodies[i].card;
      e.conts.swap(bodies[i].conts);
      e.ready.store(1, std::memory_order_release);
      return true;
    });
    if (!created) ++n_known;
    else if (v.bytes && v.len > 7 * sizeof(uint32_t)) {
      ++n_body;
      bytes_body += v.len;
    } else ++n_host;
  }
  c->staged_bodies.fetch_add(n_body, std::memory_order_relaxed);
  c->staged_host.fetch_add(n_host, std::memory_order_relaxed);
  c->staged_bytes.fetch_add(bytes_body, std::memory_order_relaxed);
  if (out) {
    out[0] = n_body;
    out[1] = n_host;
    out[2] = n_known;
  }
  return MSI_OK;
}

void msi_pcache_set_complete(MsiPostingCache *c, uint64_t view, uint32_t db_mask) {
  if (!c) return;
  std::lock_guard<std::mutex> lk(c->complete_mu);
  const uint32_t n = c->n_complete.load(std::memory_order_relaxed);
  for (uint32_t i = 0; i < n; ++i)
    if (c->complete[i].view.load(std::memory_order_relaxed) == view) {
      c->complete[i].mask.fetch_or(db_mask, std::memory_order_release);
      return;
    }
  if (n >= 8) return;   // (more views than slots: the ninth view's databases are simply asked for as before)
  c->complete[n].view.store(view, std::memory_order_relaxed);
  c->complete[n].mask.store(db_mask, std::memory_order_relaxed);
  c->n_complete.store(n + 1, std::memory_order_release);
}

bool msi_pcache_complete(MsiPostingCache *c, uint32_t db, uint64_t view) {
  if (!c || db >= 32) return false;
  const uint32_t n = c->n_complete.load(std::memory_order_acquire);
  for (uint32_t i = 0; i < n; ++i)
    if (c->complete[i].view.load(std::memory_order_relaxed) == view) {
      if (!((c->complete[i].mask.load(std::memory_order_acquire) >> db) & 1u)) return false;
      c->complete_answers.add(0, 1);
      c->counters.add(0, 1);
      return true;
    }
  return false;
}

void msi_pcache_reset(MsiPostingCache *c) {
  if (!c) return;
  DeviceGuard g(c->ctx->device);
  (void)hipDeviceSynchronize();   // no list that reads or fills the cache is in flight after this
  for (auto &sh : c->shard) {
    std::lock_guard<std::mutex> lk(sh.wmu);
    std::vector<CacheEntry *> kept;
    for (CacheEntry *e : sh.entries) {
      if (e->staged) kept.push_back(e);
      else delete e;
    }
    sh.entries.swap(kept);
    sh.count = sh.entries.size();
    uint64_t cap = 1024;
    while (cap < 2 * (sh.count + 1)) cap *= 2;
    MsiPostingCache::Table *nt = new MsiPostingCache::Table(cap);
    for (CacheEntry *e : sh.entries) {
      uint64_t j = MsiPostingCache::slot_of(e->key) & nt->mask;
      while (nt->slots[j].load(std::memory_order_relaxed)) j = (j + 1) & nt->mask;
      nt->slots[j].store(e, std::memory_order_relaxed);
    }
    MsiPostingCache::Table *old = sh.table.load(std::memory_order_relaxed);
    sh.table.store(nt, std::memory_order_release);
    delete old;   // (no reader in flight: nothing probes it)
    for (MsiPostingCache::Table *t : sh.retired) delete t;
    sh.retired.clear();
  }
  c->used.store(std::max<uint64_t>(16, c->staged_hi.load()));
  c->counters.reset();
}

void msi_pcache_staged_stats(const MsiPostingCache *c, uint64_t out[4]) {
  out[0] = c->staged_bodies.load();
  out[1] = c->staged_host.load();
  out[2] = c->staged_bytes.load();
  out[3] = c->complete_answers.sum(0);
}

// The list that was to fill the entry failed or was dropped: hand the reservation to the next reader of the key.
void msi_pcache_abandon(MsiPostingCache *, void *token) {
  if (token) static_cast<CacheEntry *>(token)->ready.store(2, std::memory_order_release);
}

uint64_t msi_pcache_device_base(const MsiPostingCache *c) { return c ? (uint64_t)(uintptr_t)c->dev : 0; }

void msi_pcache_stats(const MsiPostingCache *c, uint64_t out[4]) {
  out[0] = c->counters.sum(0);
  out[1] = c->counters.sum(1);
  out[2] = c->used.load();
  out[3] = c->cap;
}
