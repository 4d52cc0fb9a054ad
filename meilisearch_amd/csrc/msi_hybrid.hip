// msi_hybrid.hip — host-side tail of semantic and hybrid search (no device work;
// these run on the caller's thread over <= limit+offset hits).
//
//   msi_vector_sort   VectorSort as the only ranking rule
//                     (crates/milli/src/search/new/vector_sort.rs:58-168): the nns list,
//                     grouped by equal distance, each group intersected with the
//                     remaining universe (a document with several embeddings is emitted
//                     once, at its smallest distance), score = 1 - distance, then
//                     DistributionShift::shift (vector/distribution.rs:103-130).
//   msi_hybrid_merge  ScoreWithRatioResult::merge (search/hybrid.rs:102-235) without pins
//                     and distinct: merge_by(compare_scores(..).is_ge()) of the vector and
//                     keyword lists, first occurrence of a docid wins, skip `from`, take
//                     `length`, count the semantic hits.
//   msi_results_good_enough  Search::results_good_enough (search/hybrid.rs:367-386).
#include <math.h>
#include <float.h>

#include <algorithm>
#include <unordered_set>
#include <vector>

#include "msi_common.h"

extern "C" {

uint32_t msi_vector_sort(const uint32_t *docids, const float *dist, uint32_t n, int32_t has_shift,
                         float mean, float sigma, uint32_t from, uint32_t length,
                         uint32_t *out_docids, float *out_similarity) {
  if (n && (!docids || !dist)) return 0;
  std::unordered_set<uint32_t> seen;
  uint32_t rank = 0, written = 0;
  for (uint32_t i = 0; i < n && written < length; ++i) {
    if (!seen.insert(docids[i]).second) continue;   // already emitted at a smaller distance
    if (rank++ < from) continue;
    volatile float score = 1.0f - dist[i];          // vector_sort.rs:86
    float s = score;
    if (has_shift) s = msi_distribution_shift(mean, sigma, s);
    out_docids[written] = docids[i];
    out_similarity[written] = s;
    ++written;
  }
  return written;
}

uint32_t msi_hybrid_merge(const uint32_t *v_docids, const double *v_scores, const uint32_t *v_off,
                          uint32_t n_v, float v_ratio, const uint32_t *k_docids,
                          const double *k_scores, const uint32_t *k_off, uint32_t n_k, float k_ratio,
                          uint32_t from, uint32_t length, uint32_t *out_docids,
                          uint8_t *out_is_semantic, uint32_t *out_semantic_hit_count) {
  uint32_t iv = 0, ik = 0, rank = 0, written = 0, semantic = 0;
  std::unordered_set<uint32_t> seen;
  while ((iv < n_v || ik < n_k) && written < length) {
    bool take_v;
    if (iv >= n_v) take_v = false;
    else if (ik >= n_k) take_v = true;
    else {
      // itertools::merge_by: take the left (vector) item when the predicate holds
      const int32_t c = msi_compare_scores(v_scores + v_off[iv], v_off[iv + 1] - v_off[iv], v_ratio,
                                           k_scores + k_off[ik], k_off[ik + 1] - k_off[ik], k_ratio);
      take_v = c >= 0;
    }
    const uint32_t id = take_v ? v_docids[iv++] : k_docids[ik++];
    if (!seen.insert(id).second) continue;
    if (rank++ < from) continue;
    out_docids[written] = id;
    if (out_is_semantic) out_is_semantic[written] = take_v ? 1 : 0;
    semantic += take_v ? 1u : 0u;
    ++written;
  }
  if (out_semantic_hit_count) *out_semantic_hit_count = semantic;
  return written;
}

// The hybrid tail for a whole batch, straight from the outputs of msi_vs_search and
// msi_rank_query_graph_batch: vector hit score = similarity = 1 - distance (vector_sort.rs:86),
// keyword hit score = Rank::global_score of [Words{matching_words, n_terms},
// Typo{typo_count, max_typo_count}] (score_details.rs:440-547), then
// ScoreWithRatioResult::merge per query.
int32_t msi_hybrid_merge_batch(const uint32_t *v_docids, const float *v_dist, const uint32_t *v_counts,
                               uint32_t v_stride, const uint32_t *k_docids, const uint32_t *k_matching_words,
                               const uint32_t *k_typo_count, const uint32_t *k_max_typo_count,
                               const uint32_t *k_counts, uint32_t k_stride, const uint32_t *n_terms,
                               uint32_t n_queries, float semantic_ratio, uint32_t from, uint32_t length,
                               uint32_t *out_docids, uint8_t *out_is_semantic, uint32_t *out_counts,
                               uint32_t *out_semantic_hit_counts) {
  if (!v_docids || !v_dist || !v_counts || !k_docids || !k_matching_words || !k_typo_count || !k_max_typo_count ||
      !k_counts || !n_terms || !out_docids || !out_counts) {
    msi_set_error("msi_hybrid_merge_batch: invalid argument");
    return MSI_E_INVALID;
  }
  std::vector<double> vs, ks;
  std::vector<uint32_t> voff, koff;
  for (uint32_t q = 0; q < n_queries; ++q) {
    const uint32_t nv = std::min(v_counts[q], v_stride), nk = std::min(k_counts[q], k_stride);
    vs.resize(nv);
    ks.resize(nk);
    voff.resize(nv + 1);
    koff.resize(nk + 1);
    for (uint32_t i = 0; i < nv; ++i) {
      volatile float sim = 1.0f - v_dist[(size_t)q * v_stride + i];
      vs[i] = (double)sim;
      voff[i] = i;
    }
    voff[nv] = nv;
    for (uint32_t i = 0; i < nk; ++i) {
      const size_t at = (size_t)q * k_stride + i;
      const uint32_t mt = k_max_typo_count[at];
      const uint32_t ranks[2] = {k_matching_words[at], mt + 1 - std::min(k_typo_count[at], mt + 1)};
      const uint32_t maxs[2] = {n_terms[q], mt + 1};
      ks[i] = msi_rank_global_score(ranks, maxs, 2);
      koff[i] = i;
    }
    koff[nk] = nk;
    uint32_t sem = 0;
    out_counts[q] = msi_hybrid_merge(v_docids + (size_t)q * v_stride, vs.data(), voff.data(), nv, semantic_ratio,
                                     k_docids + (size_t)q * k_stride, ks.data(), koff.data(), nk,
                                     1.0f - semantic_ratio, from, length, out_docids + (size_t)q * length,
                                     out_is_semantic ? out_is_semantic + (size_t)q * length : nullptr, &sem);
    if (out_semantic_hit_counts) out_semantic_hit_counts[q] = sem;
  }
  return MSI_OK;
}

int32_t msi_results_good_enough(const double *keyword_global_scores, uint32_t n, uint32_t limit_plus_offset,
                                float semantic_ratio) {
  const double GOOD_ENOUGH_SCORE = 0.45;
  if (n < limit_plus_offset) return 0;
  for (uint32_t i = 0; i < n; ++i)
    if (keyword_global_scores[i] * (double)(1.0f - semantic_ratio) < GOOD_ENOUGH_SCORE) return 0;
  return 1;
}

}  // extern "C"
