// msi_rank.hip — S3: the Words -> Typo bucket sort over dense docid sets, gfx950.
//
// Replaces, for query graphs that are a chain of single-word terms, what
// bucket_sort (crates/milli/src/search/new/bucket_sort.rs:23-343) obtains from
// GraphBasedRankingRule<WordsGraph> followed by GraphBasedRankingRule<TypoGraph>
// (graph_based_ranking_rule.rs:97-378, ranking_rule_graph/words/mod.rs:22-53,
// ranking_rule_graph/typo/mod.rs:23-85):
//
//   Words  with TermsMatchingStrategy::Last the paths keep a prefix of the terms
//          and skip the rest at cost 1 each; the first term is never skipped
//          (query_graph.rs:346-406).  Bucket c = documents that contain the first
//          n-c terms (any derivation) and were not in an earlier bucket; score
//          Words{matching_words: n-c, max_matching_words: n}.  Strategy All: c = 0 only.
//   Typo   inside a Words bucket the query is the kept terms; term i offers edges of
//          cost s = 0..max_typo_cost(i) whose condition is "contains a derivation of
//          term i with exactly s typos"; buckets by increasing total cost, a document
//          lands in the first bucket it matches (the universe shrinks after every
//          good path); score Typo{typo_count: cost, max_typo_count: Σ max_typo_cost}.
//   leaf buckets are appended in ascending docid (bucket_sort.rs:382-460).
//
// The reference walks the ranking-rule graph path by path (DFS + dead-end cache)
// and does Roaring algebra per path.  Here the whole sort key of a document —
// (kept terms, total typos) — is a bit-sliced function of the term posting sets:
//   P_k      = U & A_0 & … & A_{k-1}                       (A_i = union of the levels of term i)
//   F_j(t)   = OR_{s<=min(t,max_i)} F_{j-1}(t-s) & L_j(s)   (documents matching terms 0..j with
//                                                            total cost t by SOME assignment;
//                                                            ∩ distributes over ∪, so this equals the
//                                                            union over all paths of cost t)
//   bucket(k,t) = P_k & ~P_{k+1} & F_{k-1}(t) & ~(F_{k-1}(0) | … | F_{k-1}(t-1))
// evaluated for 64 documents per lane with u64 bit operations: one HBM pass over
// the 3n+1 input sets gives the histogram of all buckets, a second pass
// materialises only the buckets that intersect [from, from+length).
// Algorithmic bytes: (3·n_terms + 1) · n_docs/8 per pass.
#include <string.h>

#include <algorithm>
#include <vector>

#include "msi_common.h"

typedef unsigned long long u64;

// msi_bits internals (msi_bits.hip)
struct msi_bits;
msi_ctx *msi_bits_ctx(msi_bits *p);
u64 *msi_bits_slot_ptr(msi_bits *p, uint32_t slot);
uint64_t msi_bits_words_per_slot(msi_bits *p);
uint32_t msi_bits_n_slots(msi_bits *p);
uint64_t msi_bits_n_docs(msi_bits *p);

namespace {

constexpr int NT_MAX = MSI_RANK_MAX_TERMS;   // words_limit, crates/milli/src/search/mod.rs:111
constexpr int TC_MAX = 2 * NT_MAX;           // largest total typo cost
constexpr int RT = 256;

struct RankArgs {
  const u64 *level[NT_MAX][3];   // nullptr = empty set
  uint32_t max_cost[NT_MAX];
  const u64 *universe;
  uint64_t n_words;
  uint32_t n_terms;
  uint32_t strategy_all;
  uint32_t use_typo;
  // histogram pass
  u64 *hist;                     // [NT_MAX + 1][TC_MAX + 1]
  // materialise pass
  u64 *dst;
  uint32_t sel_k, sel_t;
};

// MATERIALISE = false: histogram of every bucket; true: write bucket (sel_k, sel_t).
template <bool MATERIALISE>
__global__ __launch_bounds__(RT) void rank_words_typo_kernel(RankArgs a) {
  __shared__ uint32_t s_hist[(NT_MAX + 1) * (TC_MAX + 1)];
  if (!MATERIALISE) {
    for (uint32_t i = threadIdx.x; i < (NT_MAX + 1) * (TC_MAX + 1); i += RT) s_hist[i] = 0;
    __syncthreads();
  }
  const uint64_t stride = (uint64_t)gridDim.x * RT;
  for (uint64_t w = (uint64_t)blockIdx.x * RT + threadIdx.x; w < a.n_words; w += stride) {
    const u64 U = a.universe[w];
    u64 L[NT_MAX][3];
    u64 A[NT_MAX + 1];
#pragma unroll
    for (int j = 0; j < NT_MAX; ++j) {
      u64 any = 0;
#pragma unroll
      for (int s = 0; s < 3; ++s) {
        u64 v = 0;
        if (j < (int)a.n_terms && a.level[j][s] && s <= (int)a.max_cost[j]) v = a.level[j][s][w];
        L[j][s] = v;
        any |= v;
      }
      A[j] = any;
    }
    A[NT_MAX] = 0;
    u64 F[TC_MAX + 1];
#pragma unroll
    for (int t = 0; t <= TC_MAX; ++t) F[t] = 0;
    F[0] = U;              // zero terms matched at cost 0, inside the universe
    u64 P = U;             // P_k
    u64 out = 0;
#pragma unroll
    for (int j = 0; j < NT_MAX; ++j) {
      if (j < (int)a.n_terms) {
        // F_j from F_{j-1}
        u64 G[TC_MAX + 1];
#pragma unroll
        for (int t = 0; t <= TC_MAX; ++t) {
          u64 v = F[t] & L[j][0];
          if (t >= 1) v |= F[t - 1] & L[j][1];
          if (t >= 2) v |= F[t - 2] & L[j][2];
          G[t] = v;
        }
#pragma unroll
        for (int t = 0; t <= TC_MAX; ++t) F[t] = G[t];
        P &= A[j];
        const uint32_t k = j + 1;
        const bool last = k == a.n_terms;
        if (last || !a.strategy_all) {
          const u64 next = last ? 0ull : (P & A[j + 1]);
          const u64 D = P & ~next;          // documents whose longest matched prefix is k terms
          if (a.use_typo) {
            u64 seen = 0;
#pragma unroll
            for (int t = 0; t <= TC_MAX; ++t) {
              const u64 b = D & F[t] & ~seen;
              seen |= F[t];
              if (MATERIALISE) {
                if (k == a.sel_k && (uint32_t)t == a.sel_t) out = b;
              } else if (b) {
                atomicAdd(&s_hist[k * (TC_MAX + 1) + t], (uint32_t)__popcll(b));
              }
            }
          } else {
            if (MATERIALISE) {
              if (k == a.sel_k) out = D;
            } else if (D) {
              atomicAdd(&s_hist[k * (TC_MAX + 1)], (uint32_t)__popcll(D));
            }
          }
        }
      }
    }
    if (MATERIALISE) a.dst[w] = out;
  }
  if (!MATERIALISE) {
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < (NT_MAX + 1) * (TC_MAX + 1); i += RT)
      if (s_hist[i]) atomicAdd(&a.hist[i], (u64)s_hist[i]);
  }
}

}  // namespace

extern "C" {

int32_t msi_rank_words_typo(msi_bits *pool, const msi_rank_term *terms, uint32_t n_terms,
                            uint32_t universe_slot, uint32_t scratch_slot, int32_t strategy, int32_t use_typo,
                            uint32_t from, uint32_t length, uint32_t *out_docids,
                            uint32_t *out_matching_words, uint32_t *out_typo_count,
                            uint32_t *out_max_typo_count, uint32_t *out_n, uint64_t *out_candidates) {
  if (!pool || !terms || n_terms == 0 || n_terms > (uint32_t)NT_MAX || !out_n ||
      (length && (!out_docids || !out_matching_words || !out_typo_count || !out_max_typo_count))) {
    msi_set_error("msi_rank_words_typo: invalid argument (1..%d terms)", NT_MAX);
    return MSI_E_INVALID;
  }
  const uint32_t n_slots = msi_bits_n_slots(pool);
  if (universe_slot >= n_slots || scratch_slot >= n_slots || scratch_slot == universe_slot) {
    msi_set_error("msi_rank_words_typo: universe/scratch slot out of range");
    return MSI_E_INVALID;
  }
  RankArgs a;
  memset(&a, 0, sizeof(a));
  for (uint32_t i = 0; i < n_terms; ++i) {
    if (terms[i].max_typo_cost > 2) {
      msi_set_error("msi_rank_words_typo: term %u has max_typo_cost %u > 2", i, terms[i].max_typo_cost);
      return MSI_E_INVALID;
    }
    a.max_cost[i] = terms[i].max_typo_cost;
    for (int s = 0; s < 3; ++s) {
      const uint32_t sl = terms[i].level_slot[s];
      if (sl == MSI_NO_SLOT) continue;
      if (sl >= n_slots || sl == scratch_slot) {
        msi_set_error("msi_rank_words_typo: term %u level %d slot %u invalid", i, s, sl);
        return MSI_E_INVALID;
      }
      a.level[i][s] = msi_bits_slot_ptr(pool, sl);
    }
  }
  msi_ctx *ctx = msi_bits_ctx(pool);
  std::unique_lock<std::mutex> lk(ctx->mu);
  DeviceGuard g(ctx->device);
  hipStream_t st = ctx->stream;
  a.universe = msi_bits_slot_ptr(pool, universe_slot);
  a.n_words = msi_bits_words_per_slot(pool);
  a.n_terms = n_terms;
  a.strategy_all = strategy == MSI_TERMS_ALL;
  a.use_typo = use_typo != 0;
  a.dst = msi_bits_slot_ptr(pool, scratch_slot);
  const size_t hist_n = (size_t)(NT_MAX + 1) * (TC_MAX + 1);
  u64 *d_hist = nullptr;
  MSI_HIP_TRY(hipMalloc(&d_hist, hist_n * sizeof(u64)));
  struct Free {
    void *p;
    ~Free() { (void)hipFree(p); }
  } free_hist{d_hist};
  MSI_HIP_TRY(hipMemsetAsync(d_hist, 0, hist_n * sizeof(u64), st));
  a.hist = d_hist;
  const uint32_t grid = (uint32_t)std::min<uint64_t>((a.n_words + RT - 1) / RT, (uint64_t)ctx->n_cu * 8);
  hipLaunchKernelGGL(rank_words_typo_kernel<false>, dim3(std::max(1u, grid)), dim3(RT), 0, st, a);
  MSI_HIP_TRY(hipGetLastError());
  std::vector<u64> hist(hist_n);
  MSI_HIP_TRY(hipMemcpyAsync(hist.data(), d_hist, hist_n * sizeof(u64), hipMemcpyDeviceToHost, st));
  MSI_HIP_TRY(hipStreamSynchronize(st));
  // bucket order: kept terms descending (Words), total typos ascending (Typo)
  uint64_t total = 0, skipped = 0;
  for (u64 v : hist) total += v;
  if (out_candidates) *out_candidates = total;
  uint32_t written = 0;
  uint32_t max_so_far[NT_MAX + 1];
  max_so_far[0] = 0;
  for (uint32_t i = 0; i < n_terms; ++i) max_so_far[i + 1] = max_so_far[i] + terms[i].max_typo_cost;
  lk.unlock();  // msi_bits_first_k takes the context lock itself
  for (int k = (int)n_terms; k >= 1 && written < length; --k) {
    for (int t = 0; t <= TC_MAX && written < length; ++t) {
      const u64 c = hist[(size_t)k * (TC_MAX + 1) + t];
      if (c == 0) continue;
      if (skipped + c <= from) {  // bucket entirely before `from` (bucket_sort.rs:382-400)
        skipped += c;
        continue;
      }
      const uint64_t skip_here = from > skipped ? from - skipped : 0;
      const uint64_t want = std::min<uint64_t>(c - skip_here, length - written);
      {
        std::lock_guard<std::mutex> lk2(ctx->mu);
        DeviceGuard g2(ctx->device);
        a.sel_k = (uint32_t)k;
        a.sel_t = (uint32_t)t;
        hipLaunchKernelGGL(rank_words_typo_kernel<true>, dim3(std::max(1u, grid)), dim3(RT), 0, st, a);
        MSI_HIP_TRY(hipGetLastError());
      }
      std::vector<uint32_t> ids((size_t)(skip_here + want));
      uint32_t got = 0;
      MSI_TRY(msi_bits_first_k(pool, scratch_slot, (uint32_t)ids.size(), ids.data(), &got));
      for (uint64_t i = skip_here; i < got && written < length; ++i) {
        out_docids[written] = ids[i];
        out_matching_words[written] = (uint32_t)k;
        out_typo_count[written] = (uint32_t)t;
        out_max_typo_count[written] = max_so_far[k];
        ++written;
      }
      skipped += c;
    }
  }
  *out_n = written;
  return MSI_OK;
}

}  // extern "C"
