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
// With n-gram nodes (query_graph.rs:96-180) the chain becomes a DAG over term
// positions and the same recurrence runs over positions: a 2-gram / 3-gram node that
// ends at position p reads F at p-2 / p-3 and adds its base cost 2 / 3
// (typo/mod.rs:41-45); a document's Words bucket is the LARGEST position it reaches.
// Evaluated for 64 documents per lane with u64 bit operations: one HBM pass over
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
hipStream_t msi_bits_stream(msi_bits *p);
std::mutex &msi_bits_mutex(msi_bits *p);
u64 *msi_bits_slot_ptr(msi_bits *p, uint32_t slot);
uint64_t msi_bits_words_per_slot(msi_bits *p);
uint32_t msi_bits_n_slots(msi_bits *p);
uint64_t msi_bits_n_docs(msi_bits *p);

namespace {

constexpr int NT_MAX = MSI_RANK_MAX_TERMS;   // words_limit, crates/milli/src/search/mod.rs:111
constexpr int TC_MAX = 2 * NT_MAX;           // largest total typo cost
constexpr int RT = 256;

// Node of the query graph that ENDS at term position p (1-based end): kind 0 = the
// single term p-1, kind 1 = the 2-gram of terms p-2..p-1, kind 2 = the 3-gram
// (query_graph.rs:96-180).  An n-gram has the base typo cost n (typo/mod.rs:41-45).
// level[s] is ALWAYS a readable pointer (the universe when the set is absent) and
// mask[s] is ~0 / 0, so the loads of a word are unconditional and issue back to back.
struct NodeArg {
  uint32_t level[3];     // offset of the set inside the pool, in 64-bit words (the universe's when absent)
  uint32_t present;      // bit s = level s exists
  uint32_t max_cost;     // 0..2 ; 0xFFFFFFFF = node absent
};

struct RankArgs {
  NodeArg node[NT_MAX][3];   // [end position - 1][kind]
  const u64 *pool;           // slot 0 of the pool; sets are addressed by 32-bit word offsets
  const u64 *universe;
  uint64_t n_words;
  uint32_t n_terms;
  uint32_t strategy_all;
  uint32_t use_typo;
  uint32_t tmax;             // largest total cost any path of this query can have (<= TC_MAX)
  u64 *hist;                 // [NT_MAX + 1][TC_MAX + 1]
  u64 *dst[4];               // materialise: up to 4 buckets per pass
  uint32_t sel_k[4], sel_t[4];
  uint32_t n_sel;
};

enum { MODE_HIST = 0, MODE_STRUCT = 1, MODE_MATERIALISE = 2 };

// MODE_HIST         histogram of every (kept terms, total typo cost) bucket
// MODE_STRUCT       histogram of (kept terms, largest possible cost of a matched path): the
//                   Typo rule's max_typo_count for that Words bucket
// MODE_MATERIALISE  write up to 4 selected buckets to dst[]
// NT / TC: compile-time bounds on the number of terms and on the total cost (the host
// picks the smallest instantiation that fits the query), so small queries keep every
// posting word of a document block in registers and run short cost loops.
template <int MODE, int NT, int TC>
__device__ __forceinline__ void rank_body(const RankArgs &a, uint32_t *s_hist) {
  if (MODE != MODE_MATERIALISE) {
    for (uint32_t i = threadIdx.x; i < (NT_MAX + 1) * (TC_MAX + 1); i += RT) s_hist[i] = 0;
    __syncthreads();
  }
  const int n_terms = (int)a.n_terms;
  const int tmax = (int)a.tmax;
  const uint64_t stride = (uint64_t)gridDim.x * RT;
  for (uint64_t w = (uint64_t)blockIdx.x * RT + threadIdx.x; w < a.n_words; w += stride) {
    // ---- all posting words of this document block, unconditional loads ------------
    const u64 U = a.universe[w];
    u64 L[NT][3][3];
#pragma unroll
    for (int p = 1; p <= NT; ++p)
#pragma unroll
      for (int kind = 0; kind < 3; ++kind)
#pragma unroll
        for (int s = 0; s < 3; ++s)
          if (kind < p) {
            // wave-uniform offset -> scalar register; the load itself is unconditional
            const uint32_t off = __builtin_amdgcn_readfirstlane(a.node[p - 1][kind].level[s]);
            const uint32_t on = __builtin_amdgcn_readfirstlane(a.node[p - 1][kind].present) >> s & 1u;
            const u64 v = a.pool[(uint32_t)(off + (uint32_t)w)];
            L[p - 1][kind][s] = on ? v : 0ull;
          } else {
            L[p - 1][kind][s] = 0ull;
          }
    // ---- sweep 1: which positions does a document reach (any typo level)? --------
    u64 R[NT + 1];
    R[0] = U;
#pragma unroll
    for (int p = 1; p <= NT; ++p) {
      u64 r = 0;
#pragma unroll
      for (int kind = 0; kind < 3; ++kind)
        if (kind < p) r |= R[p - 1 - kind] & (L[p - 1][kind][0] | L[p - 1][kind][1] | L[p - 1][kind][2]);
      R[p] = r;   // masks are 0 beyond n_terms
    }
    // D[p] = documents whose LONGEST matched prefix is p terms (Words bucket n - p)
    u64 later = 0;
    u64 D[NT + 1];
#pragma unroll
    for (int p = NT; p >= 1; --p) {
      u64 d = 0;
      if (p <= n_terms && (p == n_terms || !a.strategy_all)) d = R[p] & ~later;
      D[p] = d;
      later |= R[p];
    }
    // ---- sweep 2: cost DP over positions, rolling window of 4 ---------------------
    u64 F[4][TC + 1];   // F[i] = position p-1-i after the shift below
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int t = 0; t <= TC; ++t) F[i][t] = 0;
    F[0][0] = U;
    u64 out[4] = {0, 0, 0, 0};
#pragma unroll
    for (int p = 1; p <= NT; ++p) {
      if (p <= n_terms) {
        u64 G[TC + 1];
#pragma unroll
        for (int t = 0; t <= TC; ++t) G[t] = 0;
#pragma unroll
        for (int kind = 0; kind < 3; ++kind) {
          if (kind < p) {
            const uint32_t mc = a.node[p - 1][kind].max_cost;
            if (mc != 0xFFFFFFFFu) {
              const int base = kind == 0 ? 0 : kind + 1;
              if (MODE == MODE_STRUCT) {
                const u64 any = L[p - 1][kind][0] | L[p - 1][kind][1] | L[p - 1][kind][2];
                // one pseudo level: the node's largest cost (wave-uniform)
#pragma unroll
                for (int s = 0; s < 3; ++s) {
                  if (s == (int)mc) {
#pragma unroll
                    for (int t = base + s; t <= TC; ++t)
                      if (t <= tmax) G[t] |= F[kind][t - base - s] & any;
                  }
                }
              } else {
#pragma unroll
                for (int s = 0; s < 3; ++s) {
                  if (s <= (int)mc) {
                    const u64 lv = L[p - 1][kind][s];
#pragma unroll
                    for (int t = base + s; t <= TC; ++t)
                      if (t <= tmax) G[t] |= F[kind][t - base - s] & lv;
                  }
                }
              }
            }
          }
        }
        // shift the window: F[0] becomes position p
#pragma unroll
        for (int t = 0; t <= TC; ++t) {
          if (t <= tmax) {
            F[3][t] = F[2][t];
            F[2][t] = F[1][t];
            F[1][t] = F[0][t];
            F[0][t] = G[t];
          }
        }
        const u64 Dp = D[p];
        if (MODE == MODE_STRUCT) {
          // largest structural cost of a path that matches the document
          u64 seen = 0;
#pragma unroll
          for (int t = TC; t >= 0; --t) {
            if (t <= tmax) {
              const u64 b = Dp & F[0][t] & ~seen;
              seen |= F[0][t];
              if (b) atomicAdd(&s_hist[p * (TC_MAX + 1) + t], (uint32_t)__popcll(b));
            }
          }
        } else if (a.use_typo) {
          u64 seen = 0;
#pragma unroll
          for (int t = 0; t <= TC; ++t) {
            if (t <= tmax) {
              const u64 b = Dp & F[0][t] & ~seen;
              seen |= F[0][t];
              if (MODE == MODE_MATERIALISE) {
#pragma unroll
                for (int i = 0; i < 4; ++i)
                  if ((uint32_t)p == a.sel_k[i] && (uint32_t)t == a.sel_t[i]) out[i] = b;
              } else if (b) {
                atomicAdd(&s_hist[p * (TC_MAX + 1) + t], (uint32_t)__popcll(b));
              }
            }
          }
        } else {
          if (MODE == MODE_MATERIALISE) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
              if ((uint32_t)p == a.sel_k[i]) out[i] = Dp;
          } else if (Dp) {
            atomicAdd(&s_hist[p * (TC_MAX + 1)], (uint32_t)__popcll(Dp));
          }
        }
      }
    }
    if (MODE == MODE_MATERIALISE) {
#pragma unroll
      for (int i = 0; i < 4; ++i)
        if (i < (int)a.n_sel) a.dst[i][w] = out[i];
    }
  }
  if (MODE != MODE_MATERIALISE) {
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < (NT_MAX + 1) * (TC_MAX + 1); i += RT)
      if (s_hist[i]) atomicAdd(&a.hist[i], (u64)s_hist[i]);
  }
}

template <int MODE, int NT, int TC>
__global__ __launch_bounds__(RT) void rank_query_graph_kernel(RankArgs a) {
  __shared__ uint32_t s_hist[(NT_MAX + 1) * (TC_MAX + 1)];
  rank_body<MODE, NT, TC>(a, s_hist);
}

// Batched form: blockIdx.y selects the query; its arguments live in HBM and are staged in LDS.
template <int MODE, int NT, int TC>
__global__ __launch_bounds__(RT) void rank_query_graph_batch_kernel(const RankArgs *__restrict__ args,
                                                                    const uint32_t *__restrict__ active) {
  __shared__ uint32_t s_hist[(NT_MAX + 1) * (TC_MAX + 1)];
  __shared__ RankArgs sa;
  const uint32_t q = active ? active[blockIdx.y] : blockIdx.y;
  const uint32_t *src = reinterpret_cast<const uint32_t *>(args + q);
  uint32_t *dst = reinterpret_cast<uint32_t *>(&sa);
  for (uint32_t i = threadIdx.x; i < sizeof(RankArgs) / 4; i += RT) dst[i] = src[i];
  __syncthreads();
  rank_body<MODE, NT, TC>(sa, s_hist);
}

// Size classes: (terms, total cost) bounds of the instantiations.
struct RankClass {
  int nt, tc;
};
constexpr RankClass RANK_CLASSES[3] = {{3, 8}, {6, 14}, {NT_MAX, TC_MAX}};
inline int rank_class_of(uint32_t n_terms, uint32_t tmax) {
  for (int c = 0; c < 3; ++c)
    if ((int)n_terms <= RANK_CLASSES[c].nt && (int)tmax <= RANK_CLASSES[c].tc) return c;
  return 2;
}
#define MSI_RANK_DISPATCH(CLS, CALL)                      \
  switch (CLS) {                                          \
    case 0: { constexpr int NT_ = 3, TC_ = 8; CALL; } break;   \
    case 1: { constexpr int NT_ = 6, TC_ = 14; CALL; } break;  \
    default: { constexpr int NT_ = NT_MAX, TC_ = TC_MAX; CALL; } break; \
  }

// Ascending docids of up to 4 materialised buckets of one query, in bucket order:
// one workgroup walks each bucket's bitmap 256 words at a time (popcount + block
// prefix sum) and stops as soon as skip + want documents went by.
struct ExtractArgs {
  const u64 *slot[4];
  uint32_t skip[4], want[4], out_off[4];
  uint32_t n_sel;
  uint32_t q_out;   // row of the output matrix
};
__global__ __launch_bounds__(RT) void rank_extract_kernel(const ExtractArgs *__restrict__ ex, uint64_t n_words,
                                                          uint32_t length, uint32_t *__restrict__ out) {
  __shared__ uint32_t sh[RT];
  __shared__ uint32_t s_run;
  const ExtractArgs &e = ex[blockIdx.x];   // read through the pointer: no private copy
  for (uint32_t i = 0; i < e.n_sel; ++i) {
    if (threadIdx.x == 0) s_run = 0;
    __syncthreads();
    const uint32_t lim = e.skip[i] + e.want[i];
    for (uint64_t base = 0; base < n_words; base += RT) {
      const uint64_t w = base + threadIdx.x;
      u64 bits = w < n_words ? e.slot[i][w] : 0;
      const uint32_t c = __popcll(bits);
      sh[threadIdx.x] = c;
      __syncthreads();
      for (uint32_t o = 1; o < RT; o <<= 1) {
        const uint32_t t = threadIdx.x >= o ? sh[threadIdx.x - o] : 0;
        __syncthreads();
        sh[threadIdx.x] += t;
        __syncthreads();
      }
      const uint32_t run = s_run;
      uint32_t rank = run + sh[threadIdx.x] - c;
      while (bits && rank < lim) {
        const uint32_t b = __ffsll((long long)bits) - 1;
        if (rank >= e.skip[i]) out[(size_t)e.q_out * length + e.out_off[i] + (rank - e.skip[i])] = (uint32_t)(w * 64 + b);
        ++rank;
        bits &= bits - 1;
      }
      __syncthreads();
      if (threadIdx.x == RT - 1) s_run = run + sh[RT - 1];
      __syncthreads();
      if (s_run >= lim) break;
    }
    __syncthreads();
  }
}

}  // namespace

namespace {

// Validate the nodes and fill the kernel arguments (everything except hist/dst/sel).
int32_t build_args(msi_bits *pool, const msi_rank_node *nodes, uint32_t n_nodes, uint32_t n_terms,
                   uint32_t universe_slot, uint32_t forbidden_slot, int32_t strategy, int32_t use_typo,
                   RankArgs &a) {
  if (!pool || !nodes || n_nodes == 0 || n_terms == 0 || n_terms > (uint32_t)NT_MAX) {
    msi_set_error("msi_rank: invalid argument (1..%d terms)", NT_MAX);
    return MSI_E_INVALID;
  }
  const uint32_t n_slots = msi_bits_n_slots(pool);
  if (universe_slot >= n_slots || forbidden_slot >= n_slots || forbidden_slot == universe_slot) {
    msi_set_error("msi_rank: universe/destination slot out of range");
    return MSI_E_INVALID;
  }
  memset(&a, 0, sizeof(a));
  a.universe = msi_bits_slot_ptr(pool, universe_slot);
  a.pool = msi_bits_slot_ptr(pool, 0);
  const uint64_t wps = msi_bits_words_per_slot(pool);
  if ((uint64_t)n_slots * wps > 0xFFFFFFFFull) {
    msi_set_error("msi_rank: pool larger than 2^32 words (32 GiB)");
    return MSI_E_UNSUPPORTED;
  }
  for (int p = 0; p < NT_MAX; ++p)
    for (int kind = 0; kind < 3; ++kind) {
      a.node[p][kind].max_cost = 0xFFFFFFFFu;
      a.node[p][kind].present = 0;
      for (int s = 0; s < 3; ++s) a.node[p][kind].level[s] = (uint32_t)(universe_slot * wps);  // readable placeholder
    }
  for (uint32_t i = 0; i < n_nodes; ++i) {
    const msi_rank_node &nd = nodes[i];
    if (nd.last_term >= n_terms || nd.first_term > nd.last_term || nd.last_term - nd.first_term > 2 ||
        nd.max_typo_cost > 2) {
      msi_set_error("msi_rank: node %u is not a 1/2/3-gram of the %u terms (or max_typo_cost > 2)", i, n_terms);
      return MSI_E_INVALID;
    }
    NodeArg &na = a.node[nd.last_term][nd.last_term - nd.first_term];
    if (na.max_cost != 0xFFFFFFFFu) {
      msi_set_error("msi_rank: two nodes cover terms %u..%u", nd.first_term, nd.last_term);
      return MSI_E_INVALID;
    }
    na.max_cost = nd.max_typo_cost;
    for (int s = 0; s < 3; ++s) {
      const uint32_t sl = nd.level_slot[s];
      if (sl == MSI_NO_SLOT) continue;
      if (sl >= n_slots || sl == forbidden_slot) {
        msi_set_error("msi_rank: node %u level %d slot %u invalid", i, s, sl);
        return MSI_E_INVALID;
      }
      if ((uint32_t)s > nd.max_typo_cost) continue;   // edges exist for 0..=max_typo_cost only
      na.level[s] = (uint32_t)(sl * wps);
      na.present |= 1u << s;
    }
  }
  if (strategy != MSI_TERMS_LAST && strategy != MSI_TERMS_ALL) {
    // MSI_TERMS_FREQUENCY orders the removals by document frequency: msi_keyword_search_ranked (msi_search.hip)
    msi_set_error("msi_rank: this fast path knows MSI_TERMS_LAST and MSI_TERMS_ALL (use msi_keyword_search_ranked)");
    return MSI_E_UNSUPPORTED;
  }
  a.n_words = msi_bits_words_per_slot(pool);
  a.n_terms = n_terms;
  a.strategy_all = strategy == MSI_TERMS_ALL;
  a.use_typo = use_typo != 0;
  // largest total cost of any path: longest-path DP over positions
  {
    int best[NT_MAX + 1];
    for (int p = 0; p <= NT_MAX; ++p) best[p] = -1;
    best[0] = 0;
    for (uint32_t p = 1; p <= n_terms; ++p)
      for (int kind = 0; kind < 3 && kind < (int)p; ++kind) {
        const NodeArg &na = a.node[p - 1][kind];
        if (na.max_cost == 0xFFFFFFFFu || best[p - 1 - kind] < 0) continue;
        const int c = best[p - 1 - kind] + (kind == 0 ? 0 : kind + 1) + (int)na.max_cost;
        best[p] = std::max(best[p], c);
      }
    int tm = 0;
    for (uint32_t p = 1; p <= n_terms; ++p) tm = std::max(tm, best[p]);
    a.tmax = (uint32_t)std::min(tm, TC_MAX);
  }
  a.dst[0] = msi_bits_slot_ptr(pool, forbidden_slot);
  a.n_sel = 1;
  for (int i = 0; i < 4; ++i) a.sel_k[i] = 0xFFFFFFFFu;
  return MSI_OK;
}

uint32_t rank_grid(msi_ctx *ctx, const RankArgs &a) {
  // single query: fill the chip first (one word per thread until ~4 workgroups per CU),
  // then grow the per-thread share so the histogram set-up and flush amortise
  return std::max(1u, (uint32_t)std::min<uint64_t>((a.n_words + RT - 1) / RT, (uint64_t)ctx->n_cu * 4));
}

void hist_to_buckets_impl(const u64 *hist, const u64 *hist_struct, uint32_t n_terms, std::vector<msi_rank_bucket> &out);
inline void hist_to_buckets(const u64 *hist, const u64 *hist_struct, uint32_t n_terms, std::vector<msi_rank_bucket> &out) {
  hist_to_buckets_impl(hist, hist_struct, n_terms, out);
}

// Histogram passes -> the non-empty buckets in bucket-sort order.
int32_t list_buckets(msi_bits *pool, RankArgs &a, uint32_t n_terms, std::vector<msi_rank_bucket> &out) {
  msi_ctx *ctx = msi_bits_ctx(pool);
  std::lock_guard<std::mutex> lk(msi_bits_mutex(pool));
  DeviceGuard g(ctx->device);
  hipStream_t st = msi_bits_stream(pool);
  const size_t hist_n = (size_t)(NT_MAX + 1) * (TC_MAX + 1);
  u64 *d_hist = nullptr;
  MSI_HIP_TRY(hipMalloc(&d_hist, 2 * hist_n * sizeof(u64)));
  struct Free {
    void *p;
    ~Free() { (void)hipFree(p); }
  } free_hist{d_hist};
  MSI_HIP_TRY(hipMemsetAsync(d_hist, 0, 2 * hist_n * sizeof(u64), st));
  const uint32_t grid = rank_grid(ctx, a);
  a.hist = d_hist;
  const int cls = rank_class_of(a.n_terms, a.tmax);
  MSI_RANK_DISPATCH(cls, hipLaunchKernelGGL((rank_query_graph_kernel<MODE_HIST, NT_, TC_>), dim3(grid), dim3(RT), 0, st, a));
  if (a.use_typo) {
    a.hist = d_hist + hist_n;
    MSI_RANK_DISPATCH(cls, hipLaunchKernelGGL((rank_query_graph_kernel<MODE_STRUCT, NT_, TC_>), dim3(grid), dim3(RT), 0, st, a));
  }
  MSI_HIP_TRY(hipGetLastError());
  std::vector<u64> hist(2 * hist_n);
  MSI_HIP_TRY(hipMemcpyAsync(hist.data(), d_hist, 2 * hist_n * sizeof(u64), hipMemcpyDeviceToHost, st));
  MSI_HIP_TRY(hipStreamSynchronize(st));
  hist_to_buckets(hist.data(), hist.data() + hist_n, n_terms, out);
  return MSI_OK;
}

}  // namespace

namespace {

void hist_to_buckets_impl(const u64 *hist, const u64 *hist_struct, uint32_t n_terms, std::vector<msi_rank_bucket> &out) {
  out.clear();
  const size_t hist_n = (size_t)(NT_MAX + 1) * (TC_MAX + 1);
  (void)hist_n;
  // bucket order: kept terms descending (Words), total typos ascending (Typo)
  for (int k = (int)n_terms; k >= 1; --k) {
    uint32_t max_cost = 0;
    for (int t = TC_MAX; t >= 0; --t)
      if (hist_struct[(size_t)k * (TC_MAX + 1) + t]) {
        max_cost = (uint32_t)t;
        break;
      }
    for (int t = 0; t <= TC_MAX; ++t) {
      const u64 c = hist[(size_t)k * (TC_MAX + 1) + t];
      if (c == 0) continue;
      msi_rank_bucket b;
      b.matching_words = (uint32_t)k;
      b.typo_count = (uint32_t)t;
      b.max_typo_count = max_cost;
      b._pad = 0;
      b.count = c;
      out.push_back(b);
    }
  }
}

int32_t materialise(msi_bits *pool, RankArgs &a, uint32_t k, uint32_t t) {
  msi_ctx *ctx = msi_bits_ctx(pool);
  std::lock_guard<std::mutex> lk(msi_bits_mutex(pool));
  DeviceGuard g(ctx->device);
  a.sel_k[0] = k;
  a.sel_t[0] = t;
  a.n_sel = 1;
  const int cls = rank_class_of(a.n_terms, a.tmax);
  MSI_RANK_DISPATCH(cls, hipLaunchKernelGGL((rank_query_graph_kernel<MODE_MATERIALISE, NT_, TC_>),
                                            dim3(rank_grid(ctx, a)), dim3(RT), 0, msi_bits_stream(pool), a));
  MSI_HIP_TRY(hipGetLastError());
  return MSI_OK;
}

}  // namespace

extern "C" {

int32_t msi_rank_buckets(msi_bits *pool, const msi_rank_node *nodes, uint32_t n_nodes, uint32_t n_terms,
                         uint32_t universe_slot, uint32_t scratch_slot, int32_t strategy, int32_t use_typo,
                         msi_rank_bucket *out_buckets, uint32_t cap, uint32_t *out_n) {
  if (!out_n || (cap && !out_buckets)) {
    msi_set_error("msi_rank_buckets: invalid argument");
    return MSI_E_INVALID;
  }
  RankArgs a;
  MSI_TRY(build_args(pool, nodes, n_nodes, n_terms, universe_slot, scratch_slot, strategy, use_typo, a));
  std::vector<msi_rank_bucket> b;
  MSI_TRY(list_buckets(pool, a, n_terms, b));
  *out_n = (uint32_t)b.size();
  for (uint32_t i = 0; i < b.size() && i < cap; ++i) out_buckets[i] = b[i];
  return MSI_OK;
}

int32_t msi_rank_materialise(msi_bits *pool, const msi_rank_node *nodes, uint32_t n_nodes, uint32_t n_terms,
                             uint32_t universe_slot, int32_t strategy, int32_t use_typo, uint32_t matching_words,
                             uint32_t typo_count, uint32_t dst_slot) {
  RankArgs a;
  MSI_TRY(build_args(pool, nodes, n_nodes, n_terms, universe_slot, dst_slot, strategy, use_typo, a));
  return materialise(pool, a, matching_words, typo_count);
}

int32_t msi_rank_query_graph(msi_bits *pool, const msi_rank_node *nodes, uint32_t n_nodes, uint32_t n_terms,
                             uint32_t universe_slot, uint32_t scratch_slot, int32_t strategy, int32_t use_typo,
                             uint32_t from, uint32_t length, uint32_t *out_docids,
                             uint32_t *out_matching_words, uint32_t *out_typo_count,
                             uint32_t *out_max_typo_count, uint32_t *out_n, uint64_t *out_candidates) {
  if (!out_n || (length && (!out_docids || !out_matching_words || !out_typo_count || !out_max_typo_count))) {
    msi_set_error("msi_rank_query_graph: invalid argument");
    return MSI_E_INVALID;
  }
  RankArgs a;
  MSI_TRY(build_args(pool, nodes, n_nodes, n_terms, universe_slot, scratch_slot, strategy, use_typo, a));
  std::vector<msi_rank_bucket> buckets;
  MSI_TRY(list_buckets(pool, a, n_terms, buckets));
  uint64_t total = 0, skipped = 0;
  for (const auto &b : buckets) total += b.count;
  if (out_candidates) *out_candidates = total;
  uint32_t written = 0;
  for (const auto &b : buckets) {
    if (written >= length) break;
    if (skipped + b.count <= from) {  // bucket entirely before `from` (bucket_sort.rs:382-400)
      skipped += b.count;
      continue;
    }
    const uint64_t skip_here = from > skipped ? from - skipped : 0;
    const uint64_t want = std::min<uint64_t>(b.count - skip_here, length - written);
    MSI_TRY(materialise(pool, a, b.matching_words, b.typo_count));
    std::vector<uint32_t> ids((size_t)(skip_here + want));
    uint32_t got = 0;
    MSI_TRY(msi_bits_first_k(pool, scratch_slot, (uint32_t)ids.size(), ids.data(), &got));
    for (uint64_t i = skip_here; i < got && written < length; ++i) {
      out_docids[written] = ids[i];
      out_matching_words[written] = b.matching_words;
      out_typo_count[written] = b.typo_count;
      out_max_typo_count[written] = b.max_typo_count;
      ++written;
    }
    skipped += b.count;
  }
  *out_n = written;
  return MSI_OK;
}

// Many queries per launch: the histogram passes, the materialise pass and the ordered
// extraction run with blockIdx.y = query, so a batch costs a few launches and
// synchronisations instead of ~8 per query.  Per query: 4 consecutive scratch slots.
int32_t msi_rank_query_graph_batch(msi_bits *pool, const msi_rank_query *queries, uint32_t n_queries,
                                   int32_t strategy, int32_t use_typo, uint32_t from, uint32_t length,
                                   uint32_t *out_docids, uint32_t *out_matching_words, uint32_t *out_typo_count,
                                   uint32_t *out_max_typo_count, uint32_t *out_n, uint64_t *out_candidates) {
  if (!pool || !queries || n_queries == 0 || !out_n ||
      (length && (!out_docids || !out_matching_words || !out_typo_count || !out_max_typo_count))) {
    msi_set_error("msi_rank_query_graph_batch: invalid argument");
    return MSI_E_INVALID;
  }
  const uint32_t n_slots = msi_bits_n_slots(pool);
  std::vector<RankArgs> args(n_queries);
  for (uint32_t q = 0; q < n_queries; ++q) {
    const msi_rank_query &rq = queries[q];
    if (rq.scratch_slot + 4 > n_slots) {
      msi_set_error("msi_rank_query_graph_batch: query %u needs 4 scratch slots from %u", q, rq.scratch_slot);
      return MSI_E_INVALID;
    }
    MSI_TRY(build_args(pool, rq.nodes, rq.n_nodes, rq.n_terms, rq.universe_slot, rq.scratch_slot, strategy, use_typo,
                       args[q]));
    for (int i = 0; i < 4; ++i) args[q].dst[i] = msi_bits_slot_ptr(pool, rq.scratch_slot + i);
  }
  msi_ctx *ctx = msi_bits_ctx(pool);
  std::lock_guard<std::mutex> lk(msi_bits_mutex(pool));
  DeviceGuard g(ctx->device);
  hipStream_t st = msi_bits_stream(pool);
  const size_t hist_n = (size_t)(NT_MAX + 1) * (TC_MAX + 1);
  const size_t sz_args = (size_t)n_queries * sizeof(RankArgs);
  const size_t sz_hist = 2 * (size_t)n_queries * hist_n * sizeof(u64);
  const size_t sz_ex = (size_t)n_queries * sizeof(ExtractArgs);
  const size_t sz_act = (size_t)n_queries * sizeof(uint32_t);
  const size_t sz_out = (size_t)n_queries * std::max(1u, length) * sizeof(uint32_t);
  unsigned char *d_all = nullptr;
  MSI_HIP_TRY(hipMalloc(&d_all, sz_args + sz_hist + sz_ex + sz_act + sz_out + 64));
  struct Free {
    void *p;
    ~Free() { (void)hipFree(p); }
  } free_all{d_all};
  RankArgs *d_args = reinterpret_cast<RankArgs *>(d_all);
  u64 *d_hist = reinterpret_cast<u64 *>(d_all + sz_args);
  ExtractArgs *d_ex = reinterpret_cast<ExtractArgs *>(d_all + sz_args + sz_hist);
  uint32_t *d_act = reinterpret_cast<uint32_t *>(d_all + sz_args + sz_hist + sz_ex);
  uint32_t *d_out = reinterpret_cast<uint32_t *>(d_all + sz_args + sz_hist + sz_ex + sz_act);
  const uint64_t n_words = args[0].n_words;
  const uint32_t gx = std::max(1u, (uint32_t)std::min<uint64_t>((n_words + 8 * RT - 1) / (8 * RT), 2048));
  // ---- pass 1: histograms of every query --------------------------------------------------
  MSI_HIP_TRY(hipMemsetAsync(d_hist, 0, sz_hist, st));
  for (uint32_t q = 0; q < n_queries; ++q) args[q].hist = d_hist + (size_t)q * hist_n;
  MSI_HIP_TRY(hipMemcpyAsync(d_args, args.data(), sz_args, hipMemcpyHostToDevice, st));
  int cls = 0;   // one size class for the batch: the largest any query needs
  for (uint32_t q = 0; q < n_queries; ++q) cls = std::max(cls, rank_class_of(args[q].n_terms, args[q].tmax));
  MSI_RANK_DISPATCH(cls, hipLaunchKernelGGL((rank_query_graph_batch_kernel<MODE_HIST, NT_, TC_>), dim3(gx, n_queries),
                                            dim3(RT), 0, st, d_args, (const uint32_t *)nullptr));
  std::vector<u64> hist(2 * (size_t)n_queries * hist_n);
  MSI_HIP_TRY(hipMemcpyAsync(hist.data(), d_hist, (size_t)n_queries * hist_n * sizeof(u64), hipMemcpyDeviceToHost, st));
  if (use_typo) {
    MSI_HIP_TRY(hipStreamSynchronize(st));  // args is re-uploaded with the second histogram base
    for (uint32_t q = 0; q < n_queries; ++q) args[q].hist = d_hist + ((size_t)n_queries + q) * hist_n;
    MSI_HIP_TRY(hipMemcpyAsync(d_args, args.data(), sz_args, hipMemcpyHostToDevice, st));
    MSI_RANK_DISPATCH(cls, hipLaunchKernelGGL((rank_query_graph_batch_kernel<MODE_STRUCT, NT_, TC_>),
                                              dim3(gx, n_queries), dim3(RT), 0, st, d_args, (const uint32_t *)nullptr));
    MSI_HIP_TRY(hipMemcpyAsync(hist.data() + (size_t)n_queries * hist_n, d_hist + (size_t)n_queries * hist_n,
                               (size_t)n_queries * hist_n * sizeof(u64), hipMemcpyDeviceToHost, st));
  }
  MSI_HIP_TRY(hipGetLastError());
  MSI_HIP_TRY(hipStreamSynchronize(st));
  // ---- plan: which buckets does each query need for [from, from + length)? -------------------
  struct Need {
    uint32_t k, t, max_t, skip, want, out_off;
  };
  std::vector<std::vector<Need>> needs(n_queries);
  for (uint32_t q = 0; q < n_queries; ++q) {
    std::vector<msi_rank_bucket> buckets;
    hist_to_buckets(hist.data() + (size_t)q * hist_n, hist.data() + ((size_t)n_queries + q) * hist_n,
                    queries[q].n_terms, buckets);
    uint64_t total = 0, skipped = 0;
    for (const auto &b : buckets) total += b.count;
    if (out_candidates) out_candidates[q] = total;
    uint32_t written = 0;
    for (const auto &b : buckets) {
      if (written >= length) break;
      if (skipped + b.count <= from) {
        skipped += b.count;
        continue;
      }
      const uint64_t skip_here = from > skipped ? from - skipped : 0;
      const uint32_t want = (uint32_t)std::min<uint64_t>(b.count - skip_here, length - written);
      needs[q].push_back(Need{b.matching_words, b.typo_count, b.max_typo_count, (uint32_t)skip_here, want, written});
      for (uint32_t i = 0; i < want; ++i) {
        out_matching_words[(size_t)q * length + written + i] = b.matching_words;
        out_typo_count[(size_t)q * length + written + i] = b.typo_count;
        out_max_typo_count[(size_t)q * length + written + i] = b.max_typo_count;
      }
      written += want;
      skipped += b.count;
    }
    out_n[q] = written;
  }
  // ---- pass 2: rounds of (materialise <= 4 buckets per query, extract) -------------------------
  std::vector<ExtractArgs> ex(n_queries);
  std::vector<uint32_t> act(n_queries);
  for (size_t round = 0;; ++round) {
    uint32_t n_act = 0;
    for (uint32_t q = 0; q < n_queries; ++q) {
      const size_t b0 = round * 4;
      if (b0 >= needs[q].size()) continue;
      const uint32_t n_sel = (uint32_t)std::min<size_t>(4, needs[q].size() - b0);
      ExtractArgs &e = ex[n_act];
      memset(&e, 0, sizeof(e));
      for (int i = 0; i < 4; ++i) args[q].sel_k[i] = 0xFFFFFFFFu;
      for (uint32_t i = 0; i < n_sel; ++i) {
        const Need &nd = needs[q][b0 + i];
        args[q].sel_k[i] = nd.k;
        args[q].sel_t[i] = nd.t;
        e.slot[i] = args[q].dst[i];
        e.skip[i] = nd.skip;
        e.want[i] = nd.want;
        e.out_off[i] = nd.out_off;
      }
      args[q].n_sel = n_sel;
      e.n_sel = n_sel;
      e.q_out = q;
      act[n_act++] = q;
    }
    if (n_act == 0) break;
    MSI_HIP_TRY(hipMemcpyAsync(d_args, args.data(), sz_args, hipMemcpyHostToDevice, st));
    MSI_HIP_TRY(hipMemcpyAsync(d_act, act.data(), n_act * sizeof(uint32_t), hipMemcpyHostToDevice, st));
    MSI_HIP_TRY(hipMemcpyAsync(d_ex, ex.data(), n_act * sizeof(ExtractArgs), hipMemcpyHostToDevice, st));
    MSI_RANK_DISPATCH(cls, hipLaunchKernelGGL((rank_query_graph_batch_kernel<MODE_MATERIALISE, NT_, TC_>),
                                              dim3(gx, n_act), dim3(RT), 0, st, d_args, (const uint32_t *)d_act));
    hipLaunchKernelGGL(rank_extract_kernel, dim3(n_act), dim3(RT), 0, st, (const ExtractArgs *)d_ex, n_words,
                       length, d_out);
    MSI_HIP_TRY(hipGetLastError());
    MSI_HIP_TRY(hipStreamSynchronize(st));  // the host vectors are reused by the next round
  }
  if (length)
    MSI_HIP_TRY(hipMemcpy(out_docids, d_out, (size_t)n_queries * length * sizeof(uint32_t), hipMemcpyDeviceToHost));
  return MSI_OK;
}

// Chain of single-word terms: the query graph without n-gram nodes.
int32_t msi_rank_words_typo(msi_bits *pool, const msi_rank_term *terms, uint32_t n_terms,
                            uint32_t universe_slot, uint32_t scratch_slot, int32_t strategy, int32_t use_typo,
                            uint32_t from, uint32_t length, uint32_t *out_docids,
                            uint32_t *out_matching_words, uint32_t *out_typo_count,
                            uint32_t *out_max_typo_count, uint32_t *out_n, uint64_t *out_candidates) {
  if (!terms || n_terms == 0 || n_terms > (uint32_t)NT_MAX) {
    msi_set_error("msi_rank_words_typo: invalid argument (1..%d terms)", NT_MAX);
    return MSI_E_INVALID;
  }
  msi_rank_node nodes[NT_MAX];
  for (uint32_t i = 0; i < n_terms; ++i) {
    nodes[i].first_term = nodes[i].last_term = i;
    for (int s = 0; s < 3; ++s) nodes[i].level_slot[s] = terms[i].level_slot[s];
    nodes[i].max_typo_cost = terms[i].max_typo_cost;
  }
  return msi_rank_query_graph(pool, nodes, n_terms, n_terms, universe_slot, scratch_slot, strategy, use_typo, from,
                              length, out_docids, out_matching_words, out_typo_count, out_max_typo_count, out_n,
                              out_candidates);
}

}  // extern "C"
