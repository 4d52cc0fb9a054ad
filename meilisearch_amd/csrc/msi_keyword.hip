// msi_keyword.hip — the keyword leg of Search::execute() for the Words and Typo ranking
// rules, host orchestration over the device pieces (S2 dictionary, S3 sets + bucket sort).
//
// Restates, on top of an index the CALLER owns (LMDB stays in the reference; postings
// come through msi_index_vtable as the CboRoaringBitmap bytes they are stored as):
//   located_query_terms_from_tokens / QueryGraph::from_query   parse_query.rs:28-202, query_graph.rs:96-180
//       every token, every 2-gram and 3-gram of adjacent tokens (make_ngram, parse_query.rs:227-300)
//   number_of_typos_allowed                                      parse_query.rs:204-225
//   partially_initialized_term_from_word                         compute_derivations.rs:170-253
//       exact word, zero-typo prefix derivations (<= 1000, :40-73)
//   compute_fully_if_needed -> one/two typo derivations          compute_derivations.rs:21-37,264-356
//       ONE batched device dictionary lookup for all nodes of the query
//   split_best_frequency (split words, the one-typo subterm)     compute_derivations.rs:363-383
//   compute_query_term_subset_docids                             resolve_query_graph.rs:33-59
//       per node and typo level: union of the derivations' postings, decoded on the device
//   bucket_sort over [Words, Typo]                               msi_rank_query_graph
// Not handled here (the caller keeps the reference path): phrases, synonyms, the
// word_prefix_docids database (prefix derivations are enumerated from the dictionary
// instead), other ranking rules.
#include <string.h>

#include <algorithm>
#include <string>
#include <vector>

#include "msi_common.h"

struct msi_dict;
bool msi_dict_word(const msi_dict *d, uint32_t idx, const uint8_t **w, uint32_t *len);
void msi_dict_prefix_range(const msi_dict *d, const uint8_t *prefix, uint32_t plen, uint32_t *lo, uint32_t *hi);
uint32_t msi_bits_n_slots(msi_bits *p);

namespace {

constexpr uint32_t MAX_PREFIX_COUNT = 1000;   // search/new/limits.rs:5
constexpr uint32_t MAX_ONE_TYPO_COUNT = 150;  // limits.rs:7
constexpr uint32_t MAX_TWO_TYPOS_COUNT = 50;  // limits.rs:9
constexpr uint32_t MAX_WORD_LENGTH = 250;     // crates/milli/src/lib.rs:146

uint32_t char_count(const std::string &w) {
  uint32_t n = 0;
  for (unsigned char c : w) n += (c & 0xC0) != 0x80;
  return n;
}

struct Node {
  uint32_t first, last;
  std::string word;
  bool is_prefix = false;
  bool is_ngram = false;
  uint32_t budget = 0;
  int32_t lookup = -1;  // index into the batched dictionary lookup
};

}  // namespace

extern "C" {

int32_t msi_keyword_search(msi_dict *dict, msi_bits *pool, const msi_index_vtable *index,
                           const msi_query_token *tokens, uint32_t n_tokens, const msi_keyword_params *params,
                           const uint8_t *universe_cbo, size_t universe_len, uint32_t *out_docids,
                           uint32_t *out_matching_words, uint32_t *out_typo_count,
                           uint32_t *out_max_typo_count, uint32_t *out_n, uint64_t *out_candidates) {
  if (!dict || !pool || !index || !index->word_docids || !tokens || !params || !out_n || n_tokens == 0 ||
      n_tokens > MSI_RANK_MAX_TERMS) {
    msi_set_error("msi_keyword_search: invalid argument (1..%d tokens)", MSI_RANK_MAX_TERMS);
    return MSI_E_INVALID;
  }
  // ---- query graph nodes -------------------------------------------------------------
  std::vector<Node> nodes;
  for (uint32_t last = 0; last < n_tokens; ++last) {
    for (uint32_t size = 1; size <= 3 && size <= last + 1; ++size) {
      Node nd;
      nd.first = last + 1 - size;
      nd.last = last;
      for (uint32_t i = nd.first; i <= last; ++i) {
        if (!tokens[i].word || tokens[i].len == 0) {
          msi_set_error("msi_keyword_search: empty token %u", i);
          return MSI_E_INVALID;
        }
        nd.word.append(reinterpret_cast<const char *>(tokens[i].word), tokens[i].len);
      }
      nd.is_ngram = size > 1;
      if (nd.is_ngram && nd.word.size() > MAX_WORD_LENGTH) continue;  // make_ngram: None
      nd.is_prefix = tokens[last].is_prefix != 0;
      // number_of_typos_allowed (chars, not bytes), n-gram: saturating_sub(n - 1)
      uint32_t b = 0;
      const uint32_t chars = char_count(nd.word);
      const bool exact = index->is_exact_word &&
                         index->is_exact_word(index->user, (const uint8_t *)nd.word.data(), (uint32_t)nd.word.size()) > 0;
      if (params->authorize_typos && chars >= params->min_word_len_one_typo && !exact)
        b = chars < params->min_word_len_two_typos ? 1 : 2;
      b = b > size - 1 ? b - (size - 1) : 0;
      if (nd.word.size() > MAX_WORD_LENGTH) {  // compute_derivations.rs:180-192: no derivations at all
        nd.budget = 0;
        nd.is_prefix = false;
      } else {
        nd.budget = b;
      }
      nodes.push_back(std::move(nd));
    }
  }
  const uint32_t n_slots = msi_bits_n_slots(pool);
  if (n_slots < 2 + 3 * (uint32_t)nodes.size()) {
    msi_set_error("msi_keyword_search: the pool needs %u slots (2 + 3 per query-graph node), has %u",
                  2 + 3 * (uint32_t)nodes.size(), n_slots);
    return MSI_E_INVALID;
  }
  // ---- one batched dictionary lookup for every node with a typo budget ---------------
  std::vector<msi_typo_query> tq;
  for (Node &nd : nodes) {
    if (nd.budget == 0 || nd.word.size() > MAX_WORD_LENGTH) continue;
    nd.lookup = (int32_t)tq.size();
    msi_typo_query q;
    q.word = reinterpret_cast<const uint8_t *>(nd.word.data());
    q.len = (uint32_t)nd.word.size();
    q.max_typos = (uint8_t)nd.budget;
    q.is_prefix = nd.is_prefix ? 1 : 0;
    q._pad = 0;
    tq.push_back(q);
  }
  std::vector<uint32_t> one_idx(tq.size() * MAX_ONE_TYPO_COUNT), two_idx(tq.size() * MAX_TWO_TYPOS_COUNT),
      one_cnt(tq.size()), two_cnt(tq.size());
  if (!tq.empty())
    MSI_TRY(msi_dict_lookup(dict, tq.data(), (uint32_t)tq.size(), MAX_ONE_TYPO_COUNT, MAX_TWO_TYPOS_COUNT,
                            one_idx.data(), one_cnt.data(), two_idx.data(), two_cnt.data()));
  // ---- posting sets per node and typo level ---------------------------------------------
  auto add_word = [&](MsiCboBatch &batch, const uint8_t *w, uint32_t len, int32_t original) -> int32_t {
    const uint8_t *bytes = nullptr;
    size_t n = 0;
    const int32_t st = index->word_docids(index->user, w, len, original, &bytes, &n);
    if (st < 0) {
      msi_set_error("msi_keyword_search: word_docids callback failed (%d)", st);
      return MSI_E_INTERNAL;
    }
    if (n && bytes && !msi_cbo_batch_append(batch, bytes, n)) {
      msi_set_error("msi_keyword_search: malformed posting list for a word of %u bytes", len);
      return MSI_E_INVALID;
    }
    return MSI_OK;
  };
  std::vector<msi_rank_node> rnodes(nodes.size());
  uint32_t slot = 2;
  for (size_t ni = 0; ni < nodes.size(); ++ni) {
    const Node &nd = nodes[ni];
    const uint8_t *w = reinterpret_cast<const uint8_t *>(nd.word.data());
    const uint32_t wl = (uint32_t)nd.word.size();
    MsiCboBatch lv[3];
    if (wl <= MAX_WORD_LENGTH) {
      // zero typos: the word itself (Word::Original unless n-gram, query_term/mod.rs:218-231) ...
      const int32_t orig = nd.is_ngram ? 0 : 1;
      MSI_TRY(add_word(lv[0], w, wl, orig));
      // ... and, for a prefix term, the dictionary words it is a prefix of (compute_derivations.rs:40-73)
      if (nd.is_prefix) {
        uint32_t lo = 0, hi = 0, n_pref = 0;
        msi_dict_prefix_range(dict, w, wl, &lo, &hi);
        for (uint32_t i = lo; i < hi && n_pref < MAX_PREFIX_COUNT; ++i) {
          const uint8_t *dw;
          uint32_t dl;
          msi_dict_word(dict, i, &dw, &dl);
          if (dl == wl) continue;  // the word itself
          MSI_TRY(add_word(lv[0], dw, dl, orig));
          ++n_pref;
        }
      }
      // one / two typos: the device derivations (Word::Derived)
      if (nd.lookup >= 0) {
        for (uint32_t i = 0; i < one_cnt[nd.lookup]; ++i) {
          const uint8_t *dw;
          uint32_t dl;
          msi_dict_word(dict, one_idx[(size_t)nd.lookup * MAX_ONE_TYPO_COUNT + i], &dw, &dl);
          MSI_TRY(add_word(lv[1], dw, dl, 0));
        }
        for (uint32_t i = 0; i < two_cnt[nd.lookup]; ++i) {
          const uint8_t *dw;
          uint32_t dl;
          msi_dict_word(dict, two_idx[(size_t)nd.lookup * MAX_TWO_TYPOS_COUNT + i], &dw, &dl);
          MSI_TRY(add_word(lv[2], dw, dl, 0));
        }
      }
    }
    // split words: the split with the most frequent adjacent pair, part of the one-typo
    // subterm for every budget (compute_derivations.rs:264-356,363-383)
    if (index->word_pair_proximity_docids) {
      uint64_t best = 0;
      size_t best_at = 0;
      for (size_t i = 1; i < nd.word.size(); ++i) {
        if ((nd.word[i] & 0xC0) == 0x80) continue;  // char boundaries only
        const uint8_t *bytes = nullptr;
        size_t n = 0;
        const int32_t st = index->word_pair_proximity_docids(index->user, 1, w, (uint32_t)i, w + i, (uint32_t)(wl - i),
                                                             &bytes, &n);
        if (st < 0) {
          msi_set_error("msi_keyword_search: word_pair_proximity_docids callback failed (%d)", st);
          return MSI_E_INTERNAL;
        }
        if (!n || !bytes) continue;
        const uint64_t f = msi_cbo_cardinality(bytes, n);
        if (f > best) {
          best = f;
          best_at = i;
        }
      }
      bool use_split = best > 0;
      // an n-gram whose split gives back its own component words is not split (:296-309)
      if (use_split && nd.is_ngram && nd.last - nd.first == 1 && best_at == tokens[nd.first].len) use_split = false;
      if (use_split) {
        const uint8_t *bytes = nullptr;
        size_t n = 0;
        index->word_pair_proximity_docids(index->user, 1, w, (uint32_t)best_at, w + best_at, (uint32_t)(wl - best_at),
                                          &bytes, &n);
        if (n && bytes && !msi_cbo_batch_append(lv[1], bytes, n)) {
          msi_set_error("msi_keyword_search: malformed word-pair posting list");
          return MSI_E_INVALID;
        }
      }
    }
    msi_rank_node &rn = rnodes[ni];
    rn.first_term = nd.first;
    rn.last_term = nd.last;
    rn.max_typo_cost = nd.budget <= 1 ? 1 : 2;   // query_term/mod.rs:340-370 (split words allowed)
    for (int s = 0; s < 3; ++s) {
      const bool empty = lv[s].containers.empty() && lv[s].small_ids.empty();
      if (empty) {
        rn.level_slot[s] = MSI_NO_SLOT;
      } else {
        MSI_TRY(msi_bits_decode_batch(pool, slot, lv[s], true));
        rn.level_slot[s] = slot;
      }
      ++slot;
    }
  }
  // ---- universe ---------------------------------------------------------------------------
  if (universe_cbo) {
    MsiCboBatch ub;
    if (!msi_cbo_batch_append(ub, universe_cbo, universe_len)) {
      msi_set_error("msi_keyword_search: malformed universe bitmap");
      return MSI_E_INVALID;
    }
    MSI_TRY(msi_bits_decode_batch(pool, 0, ub, true));
  } else {
    MSI_TRY(msi_bits_fill(pool, 0, 1));
  }
  return msi_rank_query_graph(pool, rnodes.data(), (uint32_t)rnodes.size(), n_tokens, 0, 1, params->strategy,
                              params->use_typo, params->from, params->length, out_docids, out_matching_words,
                              out_typo_count, out_max_typo_count, out_n, out_candidates);
}

}  // extern "C"
