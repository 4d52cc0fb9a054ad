/*
 * msi_cpubase.c — the CPU baseline leg of bench.py ("cpu_baseline", kind
 * "port").  TEST INFRASTRUCTURE ONLY, like the rest of oracle/.
 *
 * milli itself cannot be built in this image (no rustc/cargo, third-party
 * crates not vendored), so the number reported beside every GPU number is this
 * multi-threaded C restatement of the same path, written the way the reference
 * executes it on a CPU:
 *   - vector scan: exact cosine top-k, SIMD dot product (what arroy/hannoy do
 *     per candidate in linear mode, store.rs:1079-1080), rows split over threads;
 *   - typo lookup: the sorted dictionary is walked like the FST is
 *     (compute_derivations.rs:75-168): one DP column per trie edge, shared
 *     prefixes are never recomputed and dead subtrees are skipped through
 *     precomputed subtree-exit links — the work profile of `fst` ∩ Levenshtein DFA.
 * It is a reported baseline, never a parity checker (summation order is relaxed
 * in the scan; the typo matcher is cross-checked against msi_oracle.c in tests).
 */
#define _GNU_SOURCE
#include <float.h>
#include <math.h>
#include <pthread.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------ vector scan */

static inline float dot_simd(const float *a, const float *b, uint32_t d) {
  float acc[16] = {0};
  uint32_t i = 0;
  for (; i + 16 <= d; i += 16)
    for (int l = 0; l < 16; ++l) acc[l] += a[i + l] * b[i + l];
  float s = 0.f;
  for (int l = 0; l < 16; ++l) s += acc[l];
  for (; i < d; ++i) s += a[i] * b[i];
  return s;
}

typedef struct {
  float dist;
  uint32_t docid;
} hit_t;

static inline int hit_less(const hit_t *a, const hit_t *b) {
  return a->dist < b->dist || (a->dist == b->dist && a->docid < b->docid);
}

static void topk_insert(hit_t *best, uint32_t *cnt, uint32_t k, hit_t h) {
  if (*cnt == k && !hit_less(&h, &best[k - 1])) return;
  uint32_t pos = *cnt < k ? *cnt : k - 1;
  while (pos > 0 && hit_less(&h, &best[pos - 1])) {
    best[pos] = best[pos - 1];
    --pos;
  }
  best[pos] = h;
  if (*cnt < k) ++*cnt;
}

typedef struct {
  const float *rows, *norms, *queries;
  const uint32_t *docids;
  uint64_t r0, r1;
  uint32_t d, nq, k;
  hit_t *best;   /* [nq][k] */
  uint32_t *cnt; /* [nq] */
} vs_job;

static void *vs_worker(void *arg) {
  vs_job *j = (vs_job *)arg;
  float *qn = (float *)malloc(sizeof(float) * j->nq);
  for (uint32_t q = 0; q < j->nq; ++q)
    qn[q] = sqrtf(dot_simd(j->queries + (size_t)q * j->d, j->queries + (size_t)q * j->d, j->d));
  for (uint64_t r = j->r0; r < j->r1; ++r) {
    const float *x = j->rows + r * j->d;
    const float pn = j->norms[r];
    for (uint32_t q = 0; q < j->nq; ++q) {
      const float pq = dot_simd(x, j->queries + (size_t)q * j->d, j->d);
      const float pnqn = pn * qn[q];
      hit_t h;
      h.dist = pnqn > FLT_EPSILON ? (1.0f - pq / pnqn) / 2.0f : 0.0f;
      h.docid = j->docids[r];
      topk_insert(j->best + (size_t)q * j->k, &j->cnt[q], j->k, h);
    }
  }
  free(qn);
  return NULL;
}

void cpb_row_norms(const float *rows, uint64_t n, uint32_t d, float *norms) {
  for (uint64_t r = 0; r < n; ++r) norms[r] = sqrtf(dot_simd(rows + r * d, rows + r * d, d));
}

/* Exact cosine top-k of nq queries over n rows, `threads` threads over rows. */
void cpb_vs_topk_mt(const float *rows, const float *norms, const uint32_t *docids, uint64_t n, uint32_t d,
                    const float *queries, uint32_t nq, uint32_t k, uint32_t threads, uint32_t *out_docids,
                    float *out_dist, uint32_t *out_cnt) {
  if (threads < 1) threads = 1;
  pthread_t *th = (pthread_t *)malloc(sizeof(pthread_t) * threads);
  vs_job *jobs = (vs_job *)calloc(threads, sizeof(vs_job));
  for (uint32_t t = 0; t < threads; ++t) {
    vs_job *j = &jobs[t];
    j->rows = rows; j->norms = norms; j->queries = queries; j->docids = docids;
    j->r0 = n * t / threads; j->r1 = n * (t + 1) / threads;
    j->d = d; j->nq = nq; j->k = k;
    j->best = (hit_t *)malloc(sizeof(hit_t) * (size_t)nq * k);
    j->cnt = (uint32_t *)calloc(nq, sizeof(uint32_t));
    pthread_create(&th[t], NULL, vs_worker, j);
  }
  for (uint32_t t = 0; t < threads; ++t) pthread_join(th[t], NULL);
  for (uint32_t q = 0; q < nq; ++q) {
    hit_t *best = (hit_t *)malloc(sizeof(hit_t) * k);
    uint32_t cnt = 0;
    for (uint32_t t = 0; t < threads; ++t)
      for (uint32_t i = 0; i < jobs[t].cnt[q]; ++i) topk_insert(best, &cnt, k, jobs[t].best[(size_t)q * k + i]);
    for (uint32_t i = 0; i < cnt; ++i) {
      out_docids[(size_t)q * k + i] = best[i].docid;
      out_dist[(size_t)q * k + i] = best[i].dist;
    }
    out_cnt[q] = cnt;
    free(best);
  }
  for (uint32_t t = 0; t < threads; ++t) {
    free(jobs[t].best);
    free(jobs[t].cnt);
  }
  free(jobs);
  free(th);
}

/* ------------------------------------------------------------------- typo lookup */

typedef struct {
  uint32_t n;
  uint32_t *cp;       /* decoded code points of all words */
  uint32_t *cpoff;    /* [n+1] */
  uint8_t *lcp;       /* [n] common leading chars with the previous word (capped 255) */
  uint32_t *exit_[4]; /* exit_[p][i], p=1..3: first j>i with lcp[j] < p (leaves the depth-p subtree) */
} cpb_dict;

static uint32_t utf8_decode(const uint8_t *s, uint32_t len, uint32_t *out) {
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

cpb_dict *cpb_dict_build(const uint8_t *words, const uint32_t *off, uint32_t n) {
  cpb_dict *d = (cpb_dict *)calloc(1, sizeof(cpb_dict));
  d->n = n;
  d->cp = (uint32_t *)malloc(sizeof(uint32_t) * ((size_t)off[n] + 1));
  d->cpoff = (uint32_t *)malloc(sizeof(uint32_t) * ((size_t)n + 1));
  d->lcp = (uint8_t *)calloc((size_t)n + 1, 1);
  uint32_t pos = 0;
  for (uint32_t i = 0; i < n; ++i) {
    d->cpoff[i] = pos;
    pos += utf8_decode(words + off[i], off[i + 1] - off[i], d->cp + pos);
  }
  d->cpoff[n] = pos;
  for (uint32_t i = 1; i < n; ++i) {
    const uint32_t *a = d->cp + d->cpoff[i - 1], *b = d->cp + d->cpoff[i];
    uint32_t la = d->cpoff[i] - d->cpoff[i - 1], lb = d->cpoff[i + 1] - d->cpoff[i], l = 0;
    while (l < la && l < lb && l < 255 && a[l] == b[l]) ++l;
    d->lcp[i] = (uint8_t)l;
  }
  for (int p = 1; p <= 3; ++p) {
    d->exit_[p] = (uint32_t *)malloc(sizeof(uint32_t) * ((size_t)n + 1));
    uint32_t next = n;
    d->exit_[p][n] = n;
    for (uint32_t i = n; i-- > 0;) {
      d->exit_[p][i] = next;           /* first j > i with lcp[j] < p */
      if (d->lcp[i] < p) next = i;
    }
  }
  return d;
}

void cpb_dict_free(cpb_dict *d) {
  if (!d) return;
  free(d->cp); free(d->cpoff); free(d->lcp);
  for (int p = 1; p <= 3; ++p) free(d->exit_[p]);
  free(d);
}

#define MAXQ 256

/* One query: same semantics as orc_typo_lookup (msi_oracle.c), FST-like work. */
static void lookup_one(const cpb_dict *d, const uint8_t *qword, uint32_t qlen, uint32_t max_typos, int is_prefix,
                       uint32_t cap_one, uint32_t cap_two, uint32_t *out_one, uint32_t *n_one, uint32_t *out_two,
                       uint32_t *n_two) {
  *n_one = 0;
  *n_two = 0;
  if (qlen == 0 || qlen > 250 || max_typos == 0) return;
  uint32_t q[MAXQ];
  const uint32_t m = utf8_decode(qword, qlen, q);
  const uint32_t K2 = max_typos >= 2 ? 2 : 1;
  /* col[depth][i], i = 0..m : D[i][depth]; pmin[depth] = min over prefixes of D[m][.] */
  static __thread uint16_t col[260][MAXQ + 1];
  static __thread uint16_t pmin[260];
  static __thread uint16_t cmin[260];
  for (uint32_t i = 0; i <= m; ++i) col[0][i] = (uint16_t)i;
  pmin[0] = (uint16_t)m;
  cmin[0] = 0;
  uint32_t valid = 0; /* columns 0..valid are computed for the current word's prefix */
  uint32_t i = 0;
  while (i < d->n) {
    const uint32_t *w = d->cp + d->cpoff[i];
    const uint32_t wl = d->cpoff[i + 1] - d->cpoff[i];
    if (wl == 0) { ++i; continue; }
    uint32_t start = d->lcp[i] < valid ? d->lcp[i] : valid;
    if (i == 0) start = 0;
    const int sw = w[0] == q[0];
    const uint32_t K = sw ? K2 : 1; /* budget of the automaton that can accept this word */
    if (!sw && max_typos < 2) {     /* other first letter never matches the 1-typo query */
      i = d->exit_[1][i];
      valid = 0;
      continue;
    }
    /* extend columns; a column whose minimum exceeds K kills the whole subtree */
    uint32_t dead_depth = 0;
    uint32_t depth = start;
    /* a previously computed prefix may already be dead for this budget */
    for (uint32_t p = 1; p <= start; ++p)
      if (cmin[p] > K) { dead_depth = p; break; }
    if (!dead_depth) {
      while (depth < wl && depth < 258) {
        const uint32_t j = depth + 1;
        const uint32_t c = w[j - 1];
        uint16_t *cur = col[j], *prv = col[j - 1];
        cur[0] = (uint16_t)j;
        uint16_t mn = cur[0];
        for (uint32_t r = 1; r <= m; ++r) {
          uint16_t v = prv[r - 1] + (q[r - 1] != c);
          if (prv[r] + 1 < v) v = prv[r] + 1;
          if (cur[r - 1] + 1 < v) v = cur[r - 1] + 1;
          if (r > 1 && j > 1 && q[r - 1] == w[j - 2] && q[r - 2] == c && col[j - 2][r - 2] + 1 < v)
            v = col[j - 2][r - 2] + 1;
          cur[r] = v;
          if (v < mn) mn = v;
        }
        cmin[j] = mn;
        pmin[j] = cur[m] < pmin[j - 1] ? cur[m] : pmin[j - 1];
        depth = j;
        if (mn > K) { dead_depth = j; break; }
      }
    }
    valid = depth;
    if (dead_depth) {
      /* in prefix mode an earlier prefix may still have matched */
      int accepted = 0;
      if (is_prefix && pmin[dead_depth - 1 > 0 ? dead_depth - 1 : 0] <= K) accepted = 1;
      if (!accepted) {
        if (dead_depth <= 3) i = d->exit_[dead_depth][i];
        else { ++i; while (i < d->n && d->lcp[i] >= dead_depth) ++i; }
        if (valid >= dead_depth) valid = dead_depth - 1;
        continue;
      }
    }
    uint32_t dist = is_prefix ? pmin[depth] : (depth == wl ? col[wl][m] : 9999);
    if (dead_depth && !is_prefix) dist = 9999;
    if (dist <= K) {
      if (max_typos < 2) {
        if (dist == 1) {
          out_one[(*n_one)++] = i;
          if (*n_one >= cap_one) return;
        }
      } else {
        const int fin1 = *n_one >= cap_one, fin2 = *n_two >= cap_two;
        if (fin1 && fin2) return;
        if (!sw && !fin2) out_two[(*n_two)++] = i;
        else if (dist == 1) { if (!fin1) out_one[(*n_one)++] = i; }
        else if (dist == 2) { if (!fin2) out_two[(*n_two)++] = i; }
      }
    }
    ++i;
  }
}

typedef struct {
  const cpb_dict *d;
  const uint8_t *qbytes;
  const uint32_t *qoff;
  const uint8_t *qflags;
  uint32_t nq, cap_one, cap_two;
  uint32_t *next; /* shared work counter: queries cost 0.01 .. 5 ms each (prefix / 2-typo ones the most), so a
                   * static split would time the unluckiest thread instead of the cores */
  uint32_t *one, *one_cnt, *two, *two_cnt;
} dict_job;

#define DICT_GRAB 4u

static void *dict_worker(void *arg) {
  dict_job *j = (dict_job *)arg;
  for (;;) {
    const uint32_t q0 = __atomic_fetch_add(j->next, DICT_GRAB, __ATOMIC_RELAXED);
    if (q0 >= j->nq) break;
    const uint32_t q1 = q0 + DICT_GRAB < j->nq ? q0 + DICT_GRAB : j->nq;
    for (uint32_t q = q0; q < q1; ++q)
      lookup_one(j->d, j->qbytes + j->qoff[q], j->qoff[q + 1] - j->qoff[q], j->qflags[q] & 3, (j->qflags[q] >> 2) & 1,
                 j->cap_one, j->cap_two, j->one + (size_t)q * j->cap_one, &j->one_cnt[q],
                 j->two + (size_t)q * j->cap_two, &j->two_cnt[q]);
  }
  return NULL;
}

void cpb_dict_lookup_mt(const cpb_dict *d, const uint8_t *qbytes, const uint32_t *qoff, const uint8_t *qflags,
                        uint32_t nq, uint32_t cap_one, uint32_t cap_two, uint32_t threads, uint32_t *one,
                        uint32_t *one_cnt, uint32_t *two, uint32_t *two_cnt) {
  if (threads < 1) threads = 1;
  if (threads > (nq + DICT_GRAB - 1) / DICT_GRAB) threads = (nq + DICT_GRAB - 1) / DICT_GRAB;
  if (threads < 1) threads = 1;
  pthread_t *th = (pthread_t *)malloc(sizeof(pthread_t) * threads);
  dict_job *jobs = (dict_job *)calloc(threads, sizeof(dict_job));
  uint32_t next = 0;
  for (uint32_t t = 0; t < threads; ++t) {
    dict_job *j = &jobs[t];
    j->d = d; j->qbytes = qbytes; j->qoff = qoff; j->qflags = qflags;
    j->nq = nq; j->next = &next;
    j->cap_one = cap_one; j->cap_two = cap_two;
    j->one = one; j->one_cnt = one_cnt; j->two = two; j->two_cnt = two_cnt;
    pthread_create(&th[t], NULL, dict_worker, j);
  }
  for (uint32_t t = 0; t < threads; ++t) pthread_join(th[t], NULL);
  free(jobs);
  free(th);
}
