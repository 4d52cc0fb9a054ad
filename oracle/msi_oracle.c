/*
 * msi_oracle.c — CPU restatement of the reference algorithms on milli's
 * query-time scoring path.  TEST INFRASTRUCTURE ONLY: nothing under
 * meilisearch_amd/ (the product) may link, import or call this file; only
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg do.
 *
 * Plain scalar C, written from the reference's behaviour, one function per
 * reference function, each citing the file:line (under /root/reference) it
 * follows.  Build with -ffp-contract=off (see Makefile): the f32 arithmetic
 * below must not be fused.
 *
 * Pinning status (see tests/test_oracle_golden.py):
 *  - cosine distance / similarity: PINNED bit-exactly by the 10 f32 literals of
 *    crates/meilisearch/tests/search/hybrid.rs:296-406,547,758 and
 *    crates/meilisearch/tests/similar/mod.rs:281-335.
 *  - DistributionShift: PINNED by hybrid.rs:540-568.
 *  - Rank::merge / global_score: PINNED by hybrid.rs:313,776,819 and
 *    crates/milli/src/search/new/tests/cutoff.rs:330-470.
 *  - typo derivations: pinned END-TO-END only (typo.rs / typo_tolerance.rs
 *    words).  The third-party crates levenshtein_automata 0.2.1 and fst 0.4.7
 *    are not in /root/reference: their published semantics are restated
 *    (OSA distance over chars, min-over-prefixes for the prefix DFA,
 *    byte-lexicographic stream order); corners beyond the in-tree goldens are
 *    "parity unpinned" (DESIGN.md §oracle).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <float.h>

/* ------------------------------------------------------------------ vectors */

/* arroy/hannoy non-SIMD dot product (`dot_product` scalar path: iter().zip()
 * .map(a*b).sum()): f32 multiply then f32 add, in index order. */
float orc_dot_f32(const float *a, const float *b, uint32_t d) {
  float acc = 0.0f;
  for (uint32_t i = 0; i < d; ++i) {
    float p = a[i] * b[i];
    acc = acc + p;
  }
  return acc;
}

float orc_norm_f32(const float *a, uint32_t d) { return sqrtf(orc_dot_f32(a, a, d)); }

/* arroy 0.6.4 / hannoy 0.1.3 `Cosine::distance` (call sites
 * crates/milli/src/vector/store.rs:1048-1056,1077-1086):
 *   pnqn = pn*qn;  pnqn > f32::EPSILON ? (1 - pq/pnqn)/2 : 0 */
float orc_cosine_distance_pre(float pq, float pn, float qn) {
  float pnqn = pn * qn;
  if (pnqn > FLT_EPSILON) {
    float c = pq / pnqn;
    return (1.0f - c) / 2.0f;
  }
  return 0.0f;
}

float orc_cosine_distance(const float *q, const float *x, uint32_t d) {
  return orc_cosine_distance_pre(orc_dot_f32(x, q, d), orc_norm_f32(x, d), orc_norm_f32(q, d));
}

/* score = 1 - distance (crates/milli/src/search/new/vector_sort.rs:86). */
float orc_similarity(float distance) { return 1.0f - distance; }

typedef struct {
  float dist;
  uint32_t docid;
} orc_hit;

static int hit_less(const orc_hit *a, const orc_hit *b) {
  if (a->dist < b->dist) return 1;
  if (a->dist > b->dist) return 0;
  return a->docid < b->docid;
}

/* Exact (linear-mode) nns_by_vector for one store, store.rs:1036-1093 with the
 * project's tie rule (distance asc, docid asc; cutoff.rs:507-626).  Keeps the k
 * best in a sorted array by insertion. */
void orc_vs_topk(const float *rows, const uint32_t *docids, uint64_t n, uint32_t d,
                 const float *q, uint32_t k, const uint64_t *filter, uint64_t filter_nbits,
                 uint32_t *out_docids, float *out_dist, uint32_t *out_count) {
  orc_hit *best = (orc_hit *)malloc(sizeof(orc_hit) * (k ? k : 1));
  uint32_t cnt = 0;
  float qn = orc_norm_f32(q, d);
  for (uint64_t r = 0; r < n; ++r) {
    uint32_t id = docids[r];
    if (filter) {
      if ((uint64_t)id >= filter_nbits) continue;
      if (!((filter[id >> 6] >> (id & 63)) & 1ull)) continue;
    }
    const float *x = rows + r * (uint64_t)d;
    orc_hit h;
    h.dist = orc_cosine_distance_pre(orc_dot_f32(x, q, d), orc_norm_f32(x, d), qn);
    h.docid = id;
    if (k == 0) continue;
    if (cnt == k && !hit_less(&h, &best[k - 1])) continue;
    uint32_t pos = cnt < k ? cnt : k - 1;
    while (pos > 0 && hit_less(&h, &best[pos - 1])) {
      best[pos] = best[pos - 1];
      --pos;
    }
    best[pos] = h;
    if (cnt < k) ++cnt;
  }
  for (uint32_t i = 0; i < cnt; ++i) {
    out_docids[i] = best[i].docid;
    out_dist[i] = best[i].dist;
  }
  *out_count = cnt;
  free(best);
}

/* DistributionShift::shift, crates/milli/src/vector/distribution.rs:103-130. */
float orc_distribution_shift(float mean, float sigma, float score) {
  float target_mean = 0.5f, target_sigma = 0.4f;
  float factor = target_sigma / sigma;
  float offset = target_mean - (factor * mean);
  float s = factor * score + offset;
  if (s <= 0.0f) s = FLT_EPSILON;
  if (s > 1.0f) s = 1.0f;
  return s;
}

/* -------------------------------------------------------------------- scoring */

/* Rank::merge + Rank::global_score, crates/milli/src/score_details.rs:512-547
 * (u32 arithmetic, saturating_sub on the outer rank). */
double orc_rank_global_score(const uint32_t *ranks, const uint32_t *max_ranks, uint32_t n) {
  uint32_t rank = 1, max_rank = 1;
  for (uint32_t i = 0; i < n; ++i) {
    rank = rank ? rank - 1 : 0;
    rank *= max_ranks[i];
    max_rank *= max_ranks[i];
    rank += ranks[i];
  }
  return (double)rank / (double)max_rank;
}

/* compare_scores restricted to ScoreValue::Score sequences,
 * crates/milli/src/search/hybrid.rs:32-80.  Returns -1 / 0 / +1. */
int32_t orc_compare_scores(const double *l, uint32_t nl, float lr, const double *r, uint32_t nr,
                           float rr) {
  uint32_t i = 0;
  for (;;) {
    int hl = i < nl, hr = i < nr;
    if (!hl && !hr) return 0;
    if (!hl) return -1;
    if (!hr) return 1;
    double a = l[i] * (double)lr, b = r[i] * (double)rr;
    ++i;
    if (fabs(a - b) <= DBL_EPSILON) continue;
    return a < b ? -1 : 1;
  }
}

/* ----------------------------------------------------------------------- typo */

/* UTF-8 → code points (input is valid UTF-8: milli words are Rust `str`). */
uint32_t orc_utf8_decode(const uint8_t *s, uint32_t len, uint32_t *out) {
  uint32_t n = 0, i = 0;
  while (i < len) {
    uint8_t b = s[i];
    uint32_t cp, extra;
    if (b < 0x80) { cp = b; extra = 0; }
    else if (b < 0xE0) { cp = b & 0x1F; extra = 1; }
    else if (b < 0xF0) { cp = b & 0x0F; extra = 2; }
    else { cp = b & 0x07; extra = 3; }
    ++i;
    for (uint32_t e = 0; e < extra && i < len; ++e, ++i) cp = (cp << 6) | (s[i] & 0x3F);
    out[n++] = cp;
  }
  return n;
}

/* number_of_typos_allowed, parse_query.rs:204-225 (char count thresholds,
 * defaults 5 / 9: crates/milli/src/index.rs:46-47). */
uint8_t orc_typo_budget(const uint8_t *word, uint32_t len, uint32_t min_one, uint32_t min_two) {
  uint32_t cps[256];
  if (len > 250) return 0; /* MAX_WORD_LENGTH, compute_derivations.rs:180-192 */
  uint32_t n = orc_utf8_decode(word, len, cps);
  if (n < min_one) return 0;
  if (n < min_two) return 1;
  return 2;
}

/* Optimal-string-alignment (restricted Damerau) distance over code points =
 * what levenshtein_automata 0.2.1 accepts with transposition_cost_one = true
 * (LevBuilder::new(n, true), crates/milli/src/search/mod.rs:32-34).
 * Full (m+1)x(n+1) table; `prefix` returns min_j D[m][j] (build_prefix_dfa's
 * documented distance: the minimum over the prefixes of the candidate). */
static uint32_t osa_table(const uint32_t *q, uint32_t m, const uint32_t *w, uint32_t n,
                          int prefix) {
  uint32_t cols = n + 1;
  uint32_t *D = (uint32_t *)malloc(sizeof(uint32_t) * (size_t)(m + 1) * cols);
  for (uint32_t j = 0; j <= n; ++j) D[j] = j;
  for (uint32_t i = 1; i <= m; ++i) {
    D[i * cols] = i;
    for (uint32_t j = 1; j <= n; ++j) {
      uint32_t cost = q[i - 1] == w[j - 1] ? 0 : 1;
      uint32_t v = D[(i - 1) * cols + (j - 1)] + cost;
      uint32_t a = D[(i - 1) * cols + j] + 1;
      uint32_t b = D[i * cols + (j - 1)] + 1;
      if (a < v) v = a;
      if (b < v) v = b;
      if (i > 1 && j > 1 && q[i - 1] == w[j - 2] && q[i - 2] == w[j - 1]) {
        uint32_t t = D[(i - 2) * cols + (j - 2)] + 1;
        if (t < v) v = t;
      }
      D[i * cols + j] = v;
    }
  }
  uint32_t res = D[m * cols + n];
  if (prefix)
    for (uint32_t j = 0; j <= n; ++j)
      if (D[m * cols + j] < res) res = D[m * cols + j];
  free(D);
  return res;
}

uint32_t orc_osa_distance(const uint8_t *a, uint32_t alen, const uint8_t *b, uint32_t blen,
                          int prefix) {
  uint32_t *qa = (uint32_t *)malloc(sizeof(uint32_t) * (alen + 1));
  uint32_t *wb = (uint32_t *)malloc(sizeof(uint32_t) * (blen + 1));
  uint32_t m = orc_utf8_decode(a, alen, qa), n = orc_utf8_decode(b, blen, wb);
  uint32_t r = osa_table(qa, m, wb, n, prefix);
  free(qa);
  free(wb);
  return r;
}

/* `dfa.distance(state).to_u8()` of build_dfa(word, max, is_prefix) after feeding
 * `w`: Exact(d) for d <= max, AtLeast(max+1) otherwise (search/mod.rs:565-577). */
static uint32_t dfa_distance(const uint32_t *q, uint32_t m, const uint32_t *w, uint32_t n,
                             uint32_t max, int prefix) {
  uint32_t d = osa_table(q, m, w, n, prefix);
  return d <= max ? d : max + 1;
}

static uint32_t first_char_len(const uint8_t *s) {
  uint8_t b = s[0];
  return b < 0x80 ? 1 : b < 0xE0 ? 2 : b < 0xF0 ? 3 : 4;
}

static int starts_with_first(const uint8_t *w, uint32_t wlen, const uint8_t *q, uint32_t c0len) {
  return wlen >= c0len && memcmp(w, q, c0len) == 0;
}

/*
 * find_one_typo_derivations (compute_derivations.rs:75-107) when max_typos==1,
 * find_one_two_typo_derivations (compute_derivations.rs:109-168) when ==2 —
 * the literal loops of the reference over the dictionary in fst stream
 * (= byte-lexicographic) order.  Outputs are dictionary indices.
 */
void orc_typo_lookup(const uint8_t *words, const uint32_t *off, uint32_t n_words,
                     const uint8_t *qword, uint32_t qlen, uint32_t max_typos, int is_prefix,
                     uint32_t cap_one, uint32_t cap_two, uint32_t *out_one, uint32_t *n_one,
                     uint32_t *out_two, uint32_t *n_two) {
  uint32_t q[256], w[256];
  *n_one = 0;
  *n_two = 0;
  if (qlen == 0 || qlen > 250) return;
  uint32_t m = orc_utf8_decode(qword, qlen, q);
  uint32_t c0len = first_char_len(qword);
  if (max_typos <= 1) {
    /* stream of Intersection(StartsWith(c0), DFA_1): :85-106 */
    for (uint32_t i = 0; i < n_words; ++i) {
      const uint8_t *ws = words + off[i];
      uint32_t wl = off[i + 1] - off[i];
      if (wl == 0 || wl > 255) continue;
      if (!starts_with_first(ws, wl, qword, c0len)) continue;
      uint32_t n = orc_utf8_decode(ws, wl, w);
      uint32_t d = dfa_distance(q, m, w, n, 1, is_prefix);
      if (d > 1) continue; /* not accepted by the DFA */
      if (d == 1) {
        if (*n_one < cap_one) out_one[*n_one] = i;
        (*n_one)++;
        if (*n_one >= cap_one) break; /* :99-101 */
      }
    }
    return;
  }
  /* Union(DFA_1 ∩ ¬StartsWith(c0), DFA_2 ∩ StartsWith(c0)): :118-126 */
  for (uint32_t i = 0; i < n_words; ++i) {
    const uint8_t *ws = words + off[i];
    uint32_t wl = off[i + 1] - off[i];
    if (wl == 0 || wl > 255) continue;
    uint32_t n = orc_utf8_decode(ws, wl, w);
    int sw = starts_with_first(ws, wl, qword, c0len);
    int matched = sw ? dfa_distance(q, m, w, n, 2, is_prefix) <= 2
                     : dfa_distance(q, m, w, n, 1, is_prefix) <= 1;
    if (!matched) continue;
    int fin1 = *n_one >= cap_one, fin2 = *n_two >= cap_two; /* :129-134 */
    if (fin1 && fin2) break;
    if (!sw && !fin2) { /* :139-142 */
      out_two[(*n_two)++] = i;
      continue;
    }
    uint32_t d = dfa_distance(q, m, w, n, 2, is_prefix); /* :146 */
    if (d == 1) {
      if (fin1) continue;
      out_one[(*n_one)++] = i;
    } else if (d == 2) {
      if (fin2) continue;
      out_two[(*n_two)++] = i;
    }
  }
}

/* ---- binary-quantised stores ------------------------------------------------------------------
 * Quantisation: bit = (x > 0) — pinned by crates/meilisearch/tests/vector/binary_quantized.rs:67-135 (a stored
 * [-1.2, -2.3, 3.2] reads back as [0, 0, 1]).  Distance: hamming(sign bits of the query, sign bits of the row) / dim —
 * restated from the published definitions of hannoy `Hamming` / arroy `BinaryQuantizedCosine` (third-party crates that
 * are not under /root/reference; no test of the reference holds a distance value): PARITY UNPINNED for the distance.
 * Order: (distance, docid) ascending, as the f32 stores. */
void orc_bq_topk(const float *rows, const uint32_t *docids, uint64_t n, uint32_t dim, const float *q, uint32_t k,
                 const uint64_t *filter_bits, uint64_t filter_nbits, uint32_t *out_docids, float *out_dist,
                 uint32_t *out_cnt) {
  uint32_t *best_h = (uint32_t *)malloc(sizeof(uint32_t) * (k ? k : 1));
  uint32_t cnt = 0;
  for (uint64_t r = 0; r < n; ++r) {
    const uint32_t id = docids[r];
    if (filter_bits && !(id < filter_nbits && ((filter_bits[id >> 6] >> (id & 63)) & 1ull))) continue;
    uint32_t h = 0;
    for (uint32_t c = 0; c < dim; ++c) h += (rows[r * dim + c] > 0.0f) != (q[c] > 0.0f);
    /* insertion by (h, docid) */
    if (cnt == k && !(h < best_h[k - 1] || (h == best_h[k - 1] && id < out_docids[k - 1]))) continue;
    uint32_t pos = cnt < k ? cnt : k - 1;
    while (pos > 0 && (h < best_h[pos - 1] || (h == best_h[pos - 1] && id < out_docids[pos - 1]))) {
      best_h[pos] = best_h[pos - 1];
      out_docids[pos] = out_docids[pos - 1];
      --pos;
    }
    best_h[pos] = h;
    out_docids[pos] = id;
    if (cnt < k) ++cnt;
  }
  for (uint32_t i = 0; i < cnt; ++i) out_dist[i] = (float)best_h[i] / (float)dim;
  *out_cnt = cnt;
  free(best_h);
}
