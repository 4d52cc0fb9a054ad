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

// ---- federated merge (crates/meilisearch/src/search/federated/weighted_scores.rs:1-46) ------------------------------
namespace {
// WeightedScoreValue::partial_cmp (score_details.rs:57-101): 2 = not comparable
int weighted_cmp(const msi_weighted_value &l, const msi_weighted_value &r) {
  if (l.kind == 0 && r.kind == 0) {
    if (fabs(l.value - r.value) <= 2.220446049250313e-16) return 0;
    return l.value < r.value ? -1 : 1;
  }
  if (l.kind == 1 && r.kind == 1) {
    if (l.asc != r.asc) return 2;
    // compare_sort_values (score_details.rs:576-620): Null (NaN here) is below everything; ascending rules rank the
    // smaller value first, i.e. it is the "greater" hit
    const bool ln = l.value != l.value, rn = r.value != r.value;
    if (ln || rn) return ln && rn ? 0 : (ln ? -1 : 1);
    if (l.value == r.value) return 0;
    const int c = l.value < r.value ? -1 : 1;
    return l.asc ? -c : c;
  }
  if (l.kind == 2 && r.kind == 2) {
    if (l.asc != r.asc) return 2;
    const bool ln = l.value != l.value, rn = r.value != r.value;   // None
    if (ln && rn) return 0;
    if (ln) return -1;
    if (rn) return 1;
    if (fabs(l.value - r.value) <= 2.220446049250313e-16) return 0;
    return l.value < r.value ? -1 : 1;
  }
  return 2;
}
}  // namespace

extern "C" {

int32_t msi_federated_compare(const msi_weighted_value *left, uint32_t n_left, double left_weighted_global_score,
                              const msi_weighted_value *right, uint32_t n_right, double right_weighted_global_score) {
  uint32_t i = 0;
  for (;; ++i) {   // compare_partial
    const bool hl = i < n_left, hr = i < n_right;
    if (!hl && !hr) return 0;
    if (!hl) return -1;
    if (!hr) return 1;
    const int c = weighted_cmp(left[i], right[i]);
    if (c == 0) continue;
    if (c != 2) return c;
    // not comparable: the side with more remaining groups of rules wins; equal -> the weighted global scores decide
    const uint32_t lc = n_left - i - 1, rc = n_right - i - 1;
    if (lc != rc) return lc < rc ? -1 : 1;
    break;
  }
  if (left_weighted_global_score == right_weighted_global_score) return 0;
  return left_weighted_global_score < right_weighted_global_score ? -1 : 1;
}

uint32_t msi_federated_merge_q(uint32_t n_lists, const uint32_t *list_len, const msi_weighted_value *const *values,
                               const uint32_t *const *val_off, const double *const *weighted_global,
                               const uint32_t *const *query_index, uint32_t offset, uint32_t limit, uint32_t *out_list,
                               uint32_t *out_pos) {
  std::vector<uint32_t> at(n_lists, 0);
  uint32_t produced = 0, written = 0;
  auto qi = [&](uint32_t l, uint32_t j) -> uint64_t { return query_index && query_index[l] ? query_index[l][j] : l; };
  while (written < limit) {
    int best = -1;
    for (uint32_t l = 0; l < n_lists; ++l) {
      if (at[l] >= list_len[l]) continue;
      if (best < 0) {
        best = (int)l;
        continue;
      }
      const uint32_t a = at[best], b = at[l];
      const int c = msi_federated_compare(values[l] + val_off[l][b], val_off[l][b + 1] - val_off[l][b], weighted_global[l][b],
                                          values[best] + val_off[best][a], val_off[best][a + 1] - val_off[best][a],
                                          weighted_global[best][a]);
      // the bigger score first; equal: the hit of the EARLIER QUERY (perform.rs:566,609 `left.query_index <
      // right.query_index`), the earlier list when the query is the same
      if (c > 0 || (c == 0 && qi(l, b) < qi((uint32_t)best, a))) best = (int)l;
    }
    if (best < 0) break;
    if (produced >= offset) {
      out_list[written] = (uint32_t)best;
      out_pos[written] = at[best];
      ++written;
    }
    ++produced;
    ++at[best];
  }
  return written;
}

uint32_t msi_federated_merge(uint32_t n_lists, const uint32_t *list_len, const msi_weighted_value *const *values,
                             const uint32_t *const *val_off, const double *const *weighted_global, uint32_t offset,
                             uint32_t limit, uint32_t *out_list, uint32_t *out_pos) {
  return msi_federated_merge_q(n_lists, list_len, values, val_off, weighted_global, nullptr, offset, limit, out_list, out_pos);
}

}  // extern "C"
