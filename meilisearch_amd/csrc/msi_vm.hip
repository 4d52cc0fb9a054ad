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

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <thread>

#include "msi_common.h"
#include "msi_vm.h"

typedef unsigned long long u64;

msi_ctx *msi_bits_ctx(msi_bits *p);
u64 *msi_bits_slot_ptr(msi_bits *p, uint32_t slot);
uint64_t msi_bits_words_per_slot(msi_bits *p);
uint64_t msi_bits_n_docs(msi_bits *p);
uint32_t msi_bits_n_slots(msi_bits *p);
uint64_t *msi_bits_vm_block(msi_bits *p);     // pinned, fine-grained: [0] seq, [1] first-k count, [2..] counts, then ids
uint64_t msi_bits_vm_next_seq(msi_bits *p);
uint8_t *msi_bits_vm_stage(msi_bits *p, size_t need, size_t keep);   // pinned staging of the pool's decode payloads (grows, keeps `keep` bytes)

namespace {

constexpr int VT = 256;                 // threads per workgroup
constexpr uint32_t CHW = 1024;          // u64 words per chunk (65 536 documents = one Roaring container span)
constexpr uint32_t MAX_SUBS = 64;       // lists per round
constexpr uint32_t NONE = 0xFFFFFFFFu;
constexpr size_t RES_COUNTS = 2;                               // u64 index of counts[0] in the result block
constexpr size_t RES_IDS = RES_COUNTS + MSI_VM_MAX_COUNTS;     // u64 index where the u32 ids start

struct alignas(16) RoundSub {
  u64 pool_base, n_words, n_docs, host_res, seq;
  u64 stage;                              // the pool's pinned staging buffer (decode payloads), device-visible
  uint32_t n_chunks, n_phases;
  uint32_t phase_off[MSI_VM_MAX_PHASES];  // arena word offsets of each phase's first command
  uint32_t list_off;                      // arena word offset of the list's words
  uint32_t _pad0;
  uint32_t state_off;                     // arena word offset of {done[4] u32, cells[4] u64, counts[n_counts] u64, chunk cardinalities[n_chunks] u32}
  uint32_t n_counts;
};
static_assert(sizeof(RoundSub) % 16 == 0, "RoundSub array stays 16-byte aligned");

struct DecodeHdr {  // 16 bytes at the head of a decode payload, then koff[n_chunks + 1], containers, bytes
  uint32_t n_cont, cs_off, bytes_off, _pad;
};

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

__global__ __launch_bounds__(VT) void vm_kernel(uint32_t *__restrict__ arena, uint32_t phase) {
  __shared__ uint32_t s_cnt[MSI_VM_MAX_COUNTS];
  __shared__ u64 s_dec[CHW];
  __shared__ uint4 s_raw[CHW * 8 / 16 + 2];   // one container body (<= 8 KiB) + alignment slack, staged with wide loads
  __shared__ uint32_t s_scan[VT / 64 + 1];
  __shared__ uint32_t s_last;
  const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const RoundSub r = reinterpret_cast<const RoundSub *>(arena + 16)[blockIdx.y];
  const uint32_t chunk = blockIdx.x;
  if (phase >= r.n_phases || chunk >= r.n_chunks) return;
  for (uint32_t i = tid; i < r.n_counts; i += VT) s_cnt[i] = 0;
  __syncthreads();
  const u64 w0 = (u64)chunk * CHW;
  const uint32_t nw = (uint32_t)min((u64)CHW, r.n_words - w0);   // even: slots are whole 16-byte pairs
  const uint32_t n_pairs = nw / 2;
  u64 *const pool = reinterpret_cast<u64 *>(r.pool_base);
  auto S = [&](uint32_t slot) -> ulonglong2 * { return reinterpret_cast<ulonglong2 *>(pool + (u64)slot * r.n_words + w0); };
  uint32_t *const state = arena + r.state_off;
  u64 *const cells = reinterpret_cast<u64 *>(state + 4);
  u64 *const counts = cells + MSI_VM_CELLS;
  const uint8_t *const blob = reinterpret_cast<const uint8_t *>(r.stage);
  const uint32_t *pc = arena + r.phase_off[phase];
  uint32_t fk_cnt = NONE;
  uint32_t *const chunk_card = reinterpret_cast<uint32_t *>(counts + r.n_counts);   // first-k: cardinality of the set per chunk
  auto add_count = [&](uint32_t idx, uint32_t c) {
    c = wave_sum(c);
    if (lane == 0 && c) atomicAdd(&s_cnt[idx], c);
  };

  for (;;) {
    const uint32_t op = pc[0];
    if (op == VM_END) break;
    switch (op) {
      case VM_FILL: {
        ulonglong2 *d = S(pc[1]);
        const bool ones = pc[2] != 0;
        for (uint32_t p = tid; p < n_pairs; p += VT) {
          ulonglong2 v = make_ulonglong2(0, 0);
          if (ones) {
            v.x = doc_mask(w0 + 2 * p, r.n_docs);
            v.y = doc_mask(w0 + 2 * p + 1, r.n_docs);
          }
          d[p] = v;
        }
        pc += 3;
        break;
      }
      case VM_OP:
      case VM_OP_COUNT: {
        ulonglong2 *d = S(pc[1]);
        const ulonglong2 *a = S(pc[2]), *b = S(pc[3]);
        const uint32_t o = pc[4];
        uint32_t c = 0;
        for (uint32_t p = tid; p < n_pairs; p += VT) {
          const ulonglong2 v = apply_op(o, a[p], b[p]);
          d[p] = v;
          c += __popcll(v.x) + __popcll(v.y);
        }
        if (op == VM_OP_COUNT) {
          add_count(pc[5], c);
          pc += 6;
        } else {
          pc += 5;
        }
        break;
      }
      case VM_CLEAR: {
        const uint32_t n = pc[1];
        for (uint32_t k = 0; k < n; ++k) {
          ulonglong2 *d = S(pc[2 + k]);
          for (uint32_t p = tid; p < n_pairs; p += VT) d[p] = make_ulonglong2(0, 0);
        }
        pc += 2 + n;
        break;
      }
      case VM_CLAIM: {  // bucket |= docs; universe &= ~docs; stack[i] &= ~docs   (docs may be one of the stack slots)
        const ulonglong2 *docs = S(pc[1]);
        ulonglong2 *bucket = S(pc[2]), *uni = S(pc[3]);
        const uint32_t n = pc[4];
        for (uint32_t p = tid; p < n_pairs; p += VT) {
          const ulonglong2 dd = docs[p];
          if (!(dd.x | dd.y)) continue;
          ulonglong2 b = bucket[p], u = uni[p];
          b.x |= dd.x; b.y |= dd.y;
          u.x &= ~dd.x; u.y &= ~dd.y;
          bucket[p] = b;
          uni[p] = u;
          for (uint32_t k = 0; k < n; ++k) {
            ulonglong2 *sk = S(pc[5 + k]);
            ulonglong2 s = sk[p];
            s.x &= ~dd.x; s.y &= ~dd.y;
            sk[p] = s;
          }
        }
        pc += 5 + n;
        break;
      }
      case VM_AND_MANY: {  // dst[i] = prefix & cond[i], counts[base + i] = |dst[i]|
        const ulonglong2 *pre = S(pc[1]);
        const uint32_t n = pc[2], base = pc[3];
        for (uint32_t k = 0; k < n; ++k) {
          const ulonglong2 *cnd = S(pc[4 + 2 * k]);
          ulonglong2 *d = S(pc[5 + 2 * k]);
          uint32_t c = 0;
          for (uint32_t p = tid; p < n_pairs; p += VT) {
            const ulonglong2 x = pre[p], y = cnd[p];
            ulonglong2 v;
            v.x = x.x & y.x; v.y = x.y & y.y;
            d[p] = v;
            c += __popcll(v.x) + __popcll(v.y);
          }
          add_count(base + k, c);
        }
        pc += 4 + 2 * n;
        break;
      }
      case VM_PATHS: {  // the paths of one cost level in DFS order: a path claims what the earlier paths left
        const uint32_t n_paths = pc[1];
        ulonglong2 *bucket = S(pc[2]), *uni = S(pc[3]);
        const uint32_t base = pc[4], n_steps = pc[5];
        const uint32_t *off = pc + 6, *steps = off + n_paths + 1;
        for (uint32_t p = tid; p < n_pairs; p += VT) {
          ulonglong2 u = uni[p];
          if (!(u.x | u.y)) continue;
          ulonglong2 b = bucket[p];
          for (uint32_t k = 0; k < n_paths && (u.x | u.y); ++k) {
            ulonglong2 m = u;
            for (uint32_t s = off[k]; s < off[k + 1] && (m.x | m.y); ++s) {
              const ulonglong2 c = S(steps[s])[p];
              m.x &= c.x; m.y &= c.y;
            }
            if (m.x | m.y) {
              b.x |= m.x; b.y |= m.y;
              u.x &= ~m.x; u.y &= ~m.y;
              atomicAdd(&s_cnt[base + k], (uint32_t)(__popcll(m.x) + __popcll(m.y)));
            }
          }
          bucket[p] = b;
          uni[p] = u;
        }
        pc += 7 + n_paths + n_steps;
        break;
      }
      case VM_SUB_MANY: {  // slot[i] &= ~removed, counts[base + i] = |slot[i]|
        const ulonglong2 *rm = S(pc[1]);
        const uint32_t n = pc[2], base = pc[3];
        for (uint32_t k = 0; k < n; ++k) {
          ulonglong2 *d = S(pc[4 + k]);
          uint32_t c = 0;
          for (uint32_t p = tid; p < n_pairs; p += VT) {
            const ulonglong2 x = rm[p];
            ulonglong2 v = d[p];
            v.x &= ~x.x; v.y &= ~x.y;
            d[p] = v;
            c += __popcll(v.x) + __popcll(v.y);
          }
          add_count(base + k, c);
        }
        pc += 4 + n;
        break;
      }
      case VM_COUNT:
      case VM_FIRSTK: {
        const ulonglong2 *a = S(pc[1]);
        uint32_t c = 0;
        for (uint32_t p = tid; p < n_pairs; p += VT) {
          const ulonglong2 v = a[p];
          c += __popcll(v.x) + __popcll(v.y);
        }
        if (op == VM_COUNT) {
          add_count(pc[2], c);
          pc += 3;
        } else {          // slot, k, cnt: this chunk's cardinality is also kept for the ordered emit
          add_count(pc[3], c);
          fk_cnt = pc[3];
          pc += 4;
        }
        break;
      }
      case VM_DECODE: {  // the containers of THIS chunk of every posting of the batch, OR-ed in LDS, written once
        ulonglong2 *d = S(pc[1]);
        const bool overwrite = pc[2] != 0;
        const uint8_t *D = blob + pc[3];
        const DecodeHdr h = *reinterpret_cast<const DecodeHdr *>(D);
        const uint32_t *koff = reinterpret_cast<const uint32_t *>(D + sizeof(DecodeHdr));
        const MsiContainer *cs = reinterpret_cast<const MsiContainer *>(D + h.cs_off);
        const uint8_t *bytes = D + h.bytes_off;
        const uint32_t k0 = koff[chunk], k1 = koff[chunk + 1];
        if (k1 > k0) {
          __syncthreads();
          for (uint32_t i = tid; i < CHW; i += VT) s_dec[i] = 0;
          __syncthreads();
          for (uint32_t ci = k0; ci < k1; ++ci) {
            const MsiContainer c = cs[ci];
            // the body crosses PCIe once, as 16-byte aligned loads (any body alignment), and is decoded from LDS
            const uint32_t len = c.type == 0 ? 2 * c.card : (c.type == 1 ? 8192u : 4 * c.card);
            const uintptr_t b0 = reinterpret_cast<uintptr_t>(bytes + c.offset);
            const uint32_t skew = (uint32_t)(b0 & 15);
            const uint4 *src = reinterpret_cast<const uint4 *>(b0 - skew);
            const uint32_t n16 = (skew + min(len, 8192u) + 15) / 16;
            for (uint32_t i = tid; i < n16; i += VT) s_raw[i] = src[i];
            __syncthreads();
            const uint8_t *body = reinterpret_cast<const uint8_t *>(s_raw) + skew;
            if (c.type == 0) {
              for (uint32_t i = tid; i < min(c.card, 4096u); i += VT) {
                const uint32_t v = (uint32_t)body[2 * i] | ((uint32_t)body[2 * i + 1] << 8);
                atomicOr(&s_dec[v >> 6], 1ull << (v & 63));
              }
            } else if (c.type == 1) {
              for (uint32_t w = tid; w < CHW; w += VT) {
                u64 v = 0;
                for (int b = 0; b < 8; ++b) v |= (u64)body[8 * w + b] << (8 * b);
                if (v) atomicOr(&s_dec[w], v);
              }
            } else {
              for (uint32_t rr = 0; rr < min(c.card, 2048u); ++rr) {
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
          __syncthreads();
          for (uint32_t p = tid; p < n_pairs; p += VT) {
            ulonglong2 v;
            v.x = s_dec[2 * p] & doc_mask(w0 + 2 * p, r.n_docs);
            v.y = s_dec[2 * p + 1] & doc_mask(w0 + 2 * p + 1, r.n_docs);
            if (!overwrite) {
              const ulonglong2 o = d[p];
              v.x |= o.x; v.y |= o.y;
            }
            d[p] = v;
          }
        } else if (overwrite) {
          for (uint32_t p = tid; p < n_pairs; p += VT) d[p] = make_ulonglong2(0, 0);
        }
        pc += 4;
        break;
      }
      case VM_MINKEY: {  // Sort rule, first half: the smallest order key among the documents of the universe
        __syncthreads();  // one document per thread from here: other threads' set words must be visible
        const u64 *uni = pool + (u64)pc[1] * r.n_words + w0;
        const uint32_t *keys = reinterpret_cast<const uint32_t *>(((u64)pc[3] << 32) | pc[2]);
        uint32_t inv = 0;
        for (uint32_t w = wave; w < nw; w += VT / 64) {
          const u64 word = uni[w];
          if (!word) continue;
          const u64 doc = (w0 + w) * 64 + lane;
          if ((word >> lane) & 1ull) inv = max(inv, 0xFFFFFFFFu - keys[doc]);
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) inv = max(inv, (uint32_t)__shfl_xor((int)inv, o));
        if (lane == 0 && inv) atomicMax(&cells[pc[4]], (u64)inv);
        pc += 5;
        break;
      }
      case VM_TAKEKEY: {  // second half (next phase): bucket = the documents with that key, universe -= bucket
        __syncthreads();
        u64 *uni = pool + (u64)pc[1] * r.n_words + w0;
        u64 *bucket = pool + (u64)pc[2] * r.n_words + w0;
        const uint32_t *keys = reinterpret_cast<const uint32_t *>(((u64)pc[4] << 32) | pc[3]);
        const uint32_t key = 0xFFFFFFFFu - (uint32_t)cells[pc[5]];
        uint32_t c = 0;
        for (uint32_t w = wave; w < nw; w += VT / 64) {
          const u64 word = uni[w];
          u64 mask = 0;
          if (word) {
            const u64 doc = (w0 + w) * 64 + lane;
            const bool hit = ((word >> lane) & 1ull) && keys[doc] == key;
            mask = __ballot(hit);
          }
          if (lane == 0) {
            bucket[w] = mask;
            if (mask) uni[w] = word & ~mask;
            c += (uint32_t)__popcll(mask);
          }
        }
        if (lane == 0 && c) atomicAdd(&s_cnt[pc[6]], c);
        if (chunk == 0 && tid == 0) s_cnt[pc[7]] = key;   // the key itself travels as a "count"
        __syncthreads();
        pc += 8;
        break;
      }
      default:
        pc = nullptr;  // unknown opcode: stop (the host validates what it records)
        break;
    }
    if (!pc) break;
  }

  // ---- this workgroup's cardinalities leave LDS; the last workgroup of the list publishes --------------------
  __syncthreads();
  for (uint32_t i = tid; i < r.n_counts; i += VT)
    if (s_cnt[i]) atomicAdd(&counts[i], (u64)s_cnt[i]);
  if (fk_cnt != NONE && tid == 0) chunk_card[chunk] = s_cnt[fk_cnt];
  __threadfence();
  __syncthreads();
  if (tid == 0) s_last = atomicAdd(&state[phase], 1u) == r.n_chunks - 1 ? 1u : 0u;
  __syncthreads();
  if (!s_last || phase != r.n_phases - 1) return;
  __threadfence();
  u64 *res = reinterpret_cast<u64 *>(r.host_res);
  uint32_t emitted = 0;
  if (fk_cnt != NONE) {
    // ordered emit of the first k documents: chunk cardinalities are known, so only the chunks that contribute are read
    const uint32_t *pcf = arena + r.phase_off[phase];
    // find the command again (same walk as above; commands are self-delimiting)
    uint32_t slot = 0, k = 0;
    for (;;) {
      const uint32_t op = pcf[0];
      if (op == VM_END) break;
      if (op == VM_FIRSTK) { slot = pcf[1]; k = pcf[2]; break; }
      switch (op) {
        case VM_FILL: pcf += 3; break;
        case VM_OP: pcf += 5; break;
        case VM_OP_COUNT: pcf += 6; break;
        case VM_CLEAR: pcf += 2 + pcf[1]; break;
        case VM_CLAIM: pcf += 5 + pcf[4]; break;
        case VM_AND_MANY: pcf += 4 + 2 * pcf[2]; break;
        case VM_PATHS: pcf += 7 + pcf[1] + pcf[5]; break;
        case VM_SUB_MANY: pcf += 4 + pcf[2]; break;
        case VM_COUNT: pcf += 3; break;
        case VM_DECODE: pcf += 4; break;
        case VM_MINKEY: pcf += 5; break;
        case VM_TAKEKEY: pcf += 8; break;
        default: pcf += 1; break;
      }
    }
    const uint32_t *cc = chunk_card;
    uint32_t *ids = reinterpret_cast<uint32_t *>(res + RES_IDS);
    uint32_t running = 0;
    for (uint32_t c = 0; c < r.n_chunks && running < k; ++c) {
      const uint32_t n_c = __hip_atomic_load(&cc[c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (!n_c) continue;
      const u64 cw0 = (u64)c * CHW;
      const uint32_t cnw = (uint32_t)min((u64)CHW, r.n_words - cw0);
      const u64 *a = pool + (u64)slot * r.n_words + cw0;
      u64 w[4];
      uint32_t mine = 0;
#pragma unroll
      for (int j = 0; j < 4; ++j) {   // thread t owns words 4t .. 4t+3: ascending across threads
        const uint32_t wi = 4 * tid + j;
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
      for (int j = 0; j < 4; ++j) {
        u64 x = w[j];
        while (x && rank < k) {
          const uint32_t b = (uint32_t)__ffsll((long long)x) - 1;
          ids[rank++] = (uint32_t)((cw0 + 4 * tid + j) * 64 + b);
          x &= x - 1;
        }
      }
      __syncthreads();
      running += n_c;
    }
    emitted = min(running, k);
  }
  for (uint32_t i = tid; i < r.n_counts; i += VT) {
    const u64 v = __hip_atomic_load(&counts[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(&res[RES_COUNTS + i], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
  __threadfence_system();
  __syncthreads();
  if (tid == 0) {
    __hip_atomic_store(&res[1], (u64)emitted, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(&res[0], r.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

}  // namespace

// ================================================================================================ combiner

struct VmSub {
  msi_bits *pool;
  const MsiVmList *list;
  uint64_t seq;
  std::atomic<int32_t> status{1};   // 1 queued, 0 launched, < 0 failed (MSI_E_*)
};

struct msi_vm {
  msi_ctx *ctx = nullptr;
  hipStream_t stream = nullptr;
  std::thread th;
  std::mutex mu;
  std::condition_variable cv;
  std::vector<VmSub *> queue;
  std::atomic<uint32_t> pending{0};
  bool stop = false;
  struct Arena {
    uint8_t *host = nullptr, *dev = nullptr;
    size_t cap = 0;
    hipEvent_t done = nullptr;
    bool in_flight = false;
  } ar[2];
  std::atomic<uint64_t> rounds{0}, lists{0};
  void run();
};

namespace {

size_t align16(size_t x) { return (x + 15) & ~(size_t)15; }
uint32_t chunks_of(msi_bits *p) { return (uint32_t)((msi_bits_words_per_slot(p) + CHW - 1) / CHW); }

bool arena_ensure(msi_vm *vm, msi_vm::Arena &A, size_t bytes) {
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

void msi_vm::run() {
  (void)hipSetDevice(ctx->device);
  std::vector<VmSub *> batch;
  int cur = 0;
  for (;;) {
    batch.clear();
    {
      // searches come back within microseconds of each other: poll briefly before sleeping
      const auto t0 = std::chrono::steady_clock::now();
      while (pending.load(std::memory_order_acquire) == 0) {
        if (std::chrono::steady_clock::now() - t0 > std::chrono::microseconds(200)) break;
#if defined(__x86_64__)
        __builtin_ia32_pause();
#endif
      }
      std::unique_lock<std::mutex> lk(mu);
      cv.wait(lk, [&] { return stop || !queue.empty(); });
      if (stop && queue.empty()) return;
      const size_t n = std::min<size_t>(queue.size(), MAX_SUBS);
      batch.assign(queue.begin(), queue.begin() + n);
      queue.erase(queue.begin(), queue.begin() + n);
      pending.fetch_sub((uint32_t)n, std::memory_order_acq_rel);
    }
    Arena &A = ar[cur];
    // ---- layout ------------------------------------------------------------------------------------------
    const size_t n_sub = batch.size();
    size_t off = 64 + align16(n_sub * sizeof(RoundSub));
    std::vector<size_t> words_at(n_sub), state_at(n_sub);
    uint32_t max_chunks = 1, max_phases = 1;
    for (size_t i = 0; i < n_sub; ++i) {
      const MsiVmList &l = *batch[i]->list;
      words_at[i] = off;
      off = align16(off + (l.words.size() + 1) * 4);
      state_at[i] = off;
      off = align16(off + 16 + MSI_VM_CELLS * 8 + (size_t)l.n_counts * 8 + (size_t)chunks_of(batch[i]->pool) * 4);
    }
    int32_t st = MSI_OK;
    if (off > 0xFFFFFFF0ull || !arena_ensure(this, A, off)) {
      msi_set_error("msi_vm: arena of %zu bytes not available", off);
      st = MSI_E_OOM;
    }
    if (st == MSI_OK && A.in_flight) {   // the round that used this arena two rounds ago
      if (hipEventSynchronize(A.done) != hipSuccess) st = MSI_E_HIP;
      A.in_flight = false;
    }
    if (st == MSI_OK) {
      memset(A.host, 0, 64);
      reinterpret_cast<uint32_t *>(A.host)[0] = (uint32_t)n_sub;
      RoundSub *subs = reinterpret_cast<RoundSub *>(A.host + 64);
      for (size_t i = 0; i < n_sub; ++i) {
        const MsiVmList &l = *batch[i]->list;
        msi_bits *p = batch[i]->pool;
        RoundSub &r = subs[i];
        memset(&r, 0, sizeof(r));
        r.pool_base = (u64)(uintptr_t)msi_bits_slot_ptr(p, 0);
        r.n_words = msi_bits_words_per_slot(p);
        r.n_docs = msi_bits_n_docs(p);
        r.host_res = (u64)(uintptr_t)msi_bits_vm_block(p);
        r.seq = batch[i]->seq;
        r.n_chunks = (uint32_t)((r.n_words + CHW - 1) / CHW);
        r.n_phases = (uint32_t)l.phase_start.size();
        for (uint32_t ph = 0; ph < r.n_phases; ++ph) r.phase_off[ph] = (uint32_t)(words_at[i] / 4) + l.phase_start[ph];
        r.list_off = (uint32_t)(words_at[i] / 4);
        r.stage = l.stage_used ? (u64)(uintptr_t)msi_bits_vm_stage(p, 0, 0) : 0;
        r.state_off = (uint32_t)(state_at[i] / 4);
        r.n_counts = l.n_counts;
        memcpy(A.host + words_at[i], l.words.data(), l.words.size() * 4);
        reinterpret_cast<uint32_t *>(A.host + words_at[i])[l.words.size()] = VM_END;
        memset(A.host + state_at[i], 0, 16 + MSI_VM_CELLS * 8 + (size_t)l.n_counts * 8 + (size_t)r.n_chunks * 4);
        max_chunks = std::max(max_chunks, r.n_chunks);
        max_phases = std::max(max_phases, r.n_phases);
      }
      // From the first launch on a waiter may see its sequence number and leave (its VmSub is on its stack): the
      // combiner does not touch a VmSub after this point unless the round failed — then nothing was published and
      // the waiter is still polling.
      for (VmSub *s : batch) s->status.store(0, std::memory_order_release);
      if (hipMemcpyAsync(A.dev, A.host, off, hipMemcpyHostToDevice, stream) != hipSuccess) st = MSI_E_HIP;
      for (uint32_t ph = 0; ph < max_phases && st == MSI_OK; ++ph) {
        hipLaunchKernelGGL(vm_kernel, dim3(max_chunks, (uint32_t)n_sub), dim3(VT), 0, stream,
                           reinterpret_cast<uint32_t *>(A.dev), ph);
        if (hipGetLastError() != hipSuccess) st = MSI_E_HIP;
      }
      if (st == MSI_OK && hipEventRecord(A.done, stream) == hipSuccess) A.in_flight = true;
      if (st != MSI_OK) msi_set_error("msi_vm: launching a round of %zu lists failed", n_sub);
    }
    rounds.fetch_add(1, std::memory_order_relaxed);
    lists.fetch_add(n_sub, std::memory_order_relaxed);
    if (st != MSI_OK)
      for (VmSub *s : batch) s->status.store(st, std::memory_order_release);
    cur ^= 1;
  }
}

static msi_vm *vm_of(msi_ctx *ctx) {
  std::lock_guard<std::mutex> lk(ctx->vm_mu);
  if (!ctx->vm) {
    DeviceGuard g(ctx->device);
    msi_vm *vm = new msi_vm();
    vm->ctx = ctx;
    if (hipStreamCreateWithFlags(&vm->stream, hipStreamNonBlocking) != hipSuccess) {
      delete vm;
      msi_set_error("msi_vm: hipStreamCreate failed");
      return nullptr;
    }
    vm->th = std::thread([vm] { vm->run(); });
    ctx->vm = vm;
  }
  return ctx->vm;
}

void msi_vm_destroy(msi_vm *vm) {
  if (!vm) return;
  {
    std::lock_guard<std::mutex> lk(vm->mu);
    vm->stop = true;
  }
  vm->cv.notify_all();
  if (vm->th.joinable()) vm->th.join();
  DeviceGuard g(vm->ctx->device);
  (void)hipStreamSynchronize(vm->stream);
  for (auto &A : vm->ar) {
    if (A.host) (void)hipHostFree(A.host);
    if (A.dev) (void)hipFree(A.dev);
    if (A.done) (void)hipEventDestroy(A.done);
  }
  (void)hipStreamDestroy(vm->stream);
  delete vm;
}

void msi_vm_stats(msi_bits *pool, uint64_t *rounds, uint64_t *lists) {
  msi_ctx *ctx = msi_bits_ctx(pool);
  std::lock_guard<std::mutex> lk(ctx->vm_mu);
  *rounds = ctx->vm ? ctx->vm->rounds.load() : 0;
  *lists = ctx->vm ? ctx->vm->lists.load() : 0;
}

int32_t msi_vm_record_decode(MsiVmList &l, msi_bits *pool, uint32_t dst, const MsiCboBatch &batch, bool overwrite) {
  // payload: {n_cont, cs_off, bytes_off}, koff[n_chunks + 1] (containers bucketed by chunk = Roaring key), the
  // containers, the posting bytes, and the <= 7-document raw values as array containers built here
  const uint64_t n_words = msi_bits_words_per_slot(pool);
  const uint32_t n_chunks = (uint32_t)((n_words + CHW - 1) / CHW);
  std::vector<uint32_t> per(n_chunks + 1, 0);
  for (const MsiContainer &c : batch.containers)
    if (c.key < n_chunks) ++per[c.key + 1];
  // raw ids -> per-key arrays
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
    c.offset = (uint32_t)(batch.bytes.size() + extra_bytes.size());
    while (j < small.size() && small[j].first == c.key) {
      extra_bytes.push_back((uint8_t)(small[j].second & 0xFF));
      extra_bytes.push_back((uint8_t)(small[j].second >> 8));
      ++j;
    }
    c.card = (uint32_t)(j - i);
    extra.push_back(c);
    ++per[c.key + 1];
    i = j;
  }
  for (uint32_t k = 0; k < n_chunks; ++k) per[k + 1] += per[k];
  const uint32_t n_cont = per[n_chunks];
  DecodeHdr h;
  h.n_cont = n_cont;
  h.cs_off = (uint32_t)align16(sizeof(DecodeHdr) + (size_t)(n_chunks + 1) * 4);
  h.bytes_off = (uint32_t)align16(h.cs_off + (size_t)n_cont * sizeof(MsiContainer));
  h._pad = 0;
  const size_t at = align16(l.stage_used);
  const size_t total = h.bytes_off + batch.bytes.size() + extra_bytes.size();
  uint8_t *stage = msi_bits_vm_stage(pool, at + total + 16, l.stage_used);
  if (!stage) return MSI_E_OOM;
  l.stage_used = at + total;
  uint8_t *D = stage + at;
  memcpy(D, &h, sizeof(h));
  memcpy(D + sizeof(DecodeHdr), per.data(), (size_t)(n_chunks + 1) * 4);
  MsiContainer *cs = reinterpret_cast<MsiContainer *>(D + h.cs_off);
  std::vector<uint32_t> fill(per.begin(), per.end() - 1);
  for (const MsiContainer &c : batch.containers)
    if (c.key < n_chunks) cs[fill[c.key]++] = c;
  for (const MsiContainer &c : extra) cs[fill[c.key]++] = c;
  if (!batch.bytes.empty()) memcpy(D + h.bytes_off, batch.bytes.data(), batch.bytes.size());
  if (!extra_bytes.empty()) memcpy(D + h.bytes_off + batch.bytes.size(), extra_bytes.data(), extra_bytes.size());
  l.begin();
  l.words.push_back(VM_DECODE);
  l.words.push_back(dst);
  l.words.push_back(overwrite ? 1u : 0u);
  l.words.push_back((uint32_t)at);
  return MSI_OK;
}

int32_t msi_vm_run(msi_bits *pool, const MsiVmList &l, MsiVmResult *res) {
  if (l.n_counts > MSI_VM_MAX_COUNTS || l.phase_start.size() > MSI_VM_MAX_PHASES || l.phase_start.empty()) {
    msi_set_error("msi_vm_run: list outside the supported range (%u counts, %zu phases)", l.n_counts, l.phase_start.size());
    return MSI_E_UNSUPPORTED;
  }
  msi_ctx *ctx = msi_bits_ctx(pool);
  msi_vm *vm = vm_of(ctx);
  if (!vm) return MSI_E_HIP;
  volatile uint64_t *blk = msi_bits_vm_block(pool);
  if (!blk) return MSI_E_OOM;
  VmSub s;
  s.pool = pool;
  s.list = &l;
  s.seq = msi_bits_vm_next_seq(pool);
  {
    std::lock_guard<std::mutex> lk(vm->mu);
    vm->queue.push_back(&s);
    vm->pending.fetch_add(1, std::memory_order_acq_rel);
  }
  vm->cv.notify_one();
  const auto t0 = std::chrono::steady_clock::now();
  bool synced = false;
  for (uint32_t spin = 0;; ++spin) {
    if (__atomic_load_n(const_cast<uint64_t *>(&blk[0]), __ATOMIC_ACQUIRE) == s.seq) break;
    const int32_t st = s.status.load(std::memory_order_acquire);
    if (st < 0) return st;
#if defined(__x86_64__)
    __builtin_ia32_pause();
#endif
    if ((spin & 255) == 255) {
      const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
      if (us > 200.0) std::this_thread::yield();
      if (us > 2e6 && st == 0 && !synced) {   // launched long ago and still no signal: settle it with the stream
        DeviceGuard g(ctx->device);
        MSI_HIP_TRY(hipStreamSynchronize(vm->stream));
        synced = true;
        if (__atomic_load_n(const_cast<uint64_t *>(&blk[0]), __ATOMIC_ACQUIRE) != s.seq) {
          msi_set_error("msi_vm: a round finished without publishing its results");
          return MSI_E_INTERNAL;
        }
        break;
      }
    }
  }
  if (res) {
    res->counts.resize(l.n_counts);
    for (uint32_t i = 0; i < l.n_counts; ++i)
      res->counts[i] = __atomic_load_n(const_cast<uint64_t *>(&blk[RES_COUNTS + i]), __ATOMIC_RELAXED);
    res->firstk.clear();
    if (l.wants_firstk) {
      const uint32_t n = (uint32_t)__atomic_load_n(const_cast<uint64_t *>(&blk[1]), __ATOMIC_RELAXED);
      const uint32_t *ids = reinterpret_cast<const uint32_t *>(const_cast<const uint64_t *>(blk) + RES_IDS);
      res->firstk.assign(ids, ids + std::min<uint32_t>(n, MSI_VM_MAX_FIRSTK));
    }
  }
  return MSI_OK;
}
