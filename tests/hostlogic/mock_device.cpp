// TEST DOUBLE — not part of the product, never linked into libmsi.so, never loaded by meilisearch_amd.
//
// Plain-C++ stand-ins for the device-set pool (msi_bits_*) and the device dictionary (msi_dict_*) so that the HOST
// logic of msi_search.hip (query graph, rule graphs, path enumeration, bucket sort, caches) can be exercised by the
// CPU test tier: tests/test_search_hostlogic_cpu.py compiles msi_search.hip together with this file into
// tests/hostlogic/_build/libmsi_hostlogic_test.so and replays the reference snapshots through it.  The GPU tier
// (tests/test_search_gpu.py) runs the same cases through the real kernels.
#include <math.h>
#include <string.h>

#include <algorithm>
#include <string>
#include <vector>

#include "msi_common.h"

struct msi_bits {
  uint64_t n_docs = 0, n_words = 0;
  uint32_t n_slots = 0;
  uint64_t region_counts[MSI_BITS_PATH_REGIONS * MSI_BITS_REGION_PATHS] = {};
  std::vector<uint64_t> pool;
  uint64_t *slot(uint32_t s) { return pool.data() + (uint64_t)s * n_words; }
};


typedef int32_t (*mock_lookup_fn)(const uint8_t *word, uint32_t len, uint32_t max_typos, uint32_t is_prefix, uint32_t cap_one,
                                  uint32_t cap_two, uint32_t *one, uint32_t *n_one, uint32_t *two, uint32_t *n_two);
struct msi_dict {
  std::vector<std::string> words;  // sorted
  mock_lookup_fn lookup = nullptr;
};

struct msi_doc_keys {
  std::vector<uint32_t> keys;
};

struct msi_doc_values {
  std::vector<std::vector<uint32_t>> per_doc;
  uint32_t n_values = 0;
};

struct msi_geo_points {
  std::vector<double> lat_lng;
};

uint32_t msi_bits_n_slots(msi_bits *p) { return p->n_slots; }

bool msi_dict_word(const msi_dict *d, uint32_t idx, const uint8_t **w, uint32_t *len) {
  if (idx >= d->words.size()) return false;
  *w = (const uint8_t *)d->words[idx].data();
  *len = (uint32_t)d->words[idx].size();
  return true;
}
void msi_dict_prefix_range(const msi_dict *d, const uint8_t *prefix, uint32_t plen, uint32_t *lo, uint32_t *hi) {
  const std::string p((const char *)prefix, plen);
  uint32_t a = 0;
  while (a < d->words.size() && d->words[a].compare(0, plen, p) < 0) ++a;
  uint32_t b = a;
  while (b < d->words.size() && d->words[b].size() >= plen && d->words[b].compare(0, plen, p) == 0) ++b;
  *lo = a;
  *hi = b;
}

static uint64_t popcount_slot(msi_bits *p, uint32_t s) {
  uint64_t c = 0;
  for (uint64_t i = 0; i < p->n_words; ++i) c += (uint64_t)__builtin_popcountll(p->slot(s)[i]);
  return c;
}

int32_t msi_bits_decode_batch(msi_bits *p, uint32_t slot, const MsiCboBatch &b, bool clear) {
  uint64_t *dst = p->slot(slot);
  if (clear) std::fill(dst, dst + p->n_words, 0ull);
  auto set = [&](uint64_t id) {
    if (id < p->n_docs) dst[id >> 6] |= 1ull << (id & 63);
  };
  for (const MsiContainer &c : b.containers) {
    const uint8_t *body = b.bytes.data() + c.offset;
    const uint64_t hi = (uint64_t)c.key << 16;
    auto rd16 = [&](size_t o) { return (uint32_t)body[o] | ((uint32_t)body[o + 1] << 8); };
    if (c.type == 0) {
      for (uint32_t i = 0; i < c.card; ++i) set(hi | rd16(2 * i));
    } else if (c.type == 1) {
      // a bitmap container is 1024 little-endian words that line up with the destination words
      const uint64_t w0 = hi >> 6;
      for (uint32_t w = 0; w < 1024 && w0 + w < p->n_words; ++w) {
        uint64_t v;
        memcpy(&v, body + 8 * w, 8);
        const uint64_t base = (w0 + w) * 64;
        if (base + 64 > p->n_docs) v &= base >= p->n_docs ? 0ull : (~0ull >> (64 - (p->n_docs - base)));
        dst[w0 + w] |= v;
      }
    } else {
      for (uint32_t r = 0; r < c.card; ++r) {
        const uint32_t start = rd16(4 * r), len = rd16(4 * r + 2);
        for (uint32_t v = start; v <= start + len; ++v) set(hi | v);
      }
    }
  }
  for (uint32_t id : b.small_ids) set(id);
  return MSI_OK;
}

int32_t msi_bits_and_many_count(msi_bits *p, uint32_t prefix, uint32_t n, const uint32_t *cond, const uint32_t *dst,
                                uint64_t *counts) {
  for (uint32_t k = 0; k < n; ++k) {
    for (uint64_t i = 0; i < p->n_words; ++i) p->slot(dst[k])[i] = p->slot(prefix)[i] & p->slot(cond[k])[i];
    counts[k] = popcount_slot(p, dst[k]);
  }
  return MSI_OK;
}

int32_t msi_bits_claim(msi_bits *p, uint32_t docs, uint32_t bucket, uint32_t universe, uint32_t n_stack,
                       const uint32_t *stack) {
  for (uint64_t i = 0; i < p->n_words; ++i) {
    const uint64_t d = p->slot(docs)[i];
    p->slot(bucket)[i] |= d;
    p->slot(universe)[i] &= ~d;
    for (uint32_t k = 0; k < n_stack; ++k) p->slot(stack[k])[i] &= ~d;
  }
  return MSI_OK;
}

int32_t msi_bits_clear_slots(msi_bits *p, uint32_t n, const uint32_t *slots) {
  for (uint32_t k = 0; k < n; ++k) std::fill(p->slot(slots[k]), p->slot(slots[k]) + p->n_words, 0ull);
  return MSI_OK;
}

int32_t msi_bits_paths_claim(msi_bits *p, uint32_t n_paths, const uint32_t *path_off, const uint32_t *step_slots,
                             uint32_t bucket, uint32_t universe, uint64_t *counts) {
  for (uint32_t k = 0; k < n_paths; ++k) counts[k] = 0;
  for (uint64_t i = 0; i < p->n_words; ++i) {
    uint64_t u = p->slot(universe)[i], b = p->slot(bucket)[i];
    for (uint32_t k = 0; k < n_paths && u; ++k) {
      uint64_t m = u;
      for (uint32_t s = path_off[k]; s < path_off[k + 1]; ++s) m &= p->slot(step_slots[s])[i];
      if (m) {
        b |= m;
        u &= ~m;
        counts[k] += (uint64_t)__builtin_popcountll(m);
      }
    }
    p->slot(universe)[i] = u;
    p->slot(bucket)[i] = b;
  }
  return MSI_OK;
}

// the device's fit rule for a level behind a shared wait: <= 64 paths, <= 448 steps, <= 32 distinct conditions
int32_t msi_bits_paths_enqueue(msi_bits *p, uint32_t n_paths, const uint32_t *path_off, const uint32_t *step_slots,
                               uint32_t bucket, uint32_t universe, uint32_t region) {
  if (!n_paths || region >= MSI_BITS_PATH_REGIONS) return MSI_E_INVALID;
  if (n_paths > MSI_BITS_REGION_PATHS || path_off[n_paths] > 448) return MSI_E_UNSUPPORTED;
  std::vector<uint32_t> distinct;
  for (uint32_t s = 0; s < path_off[n_paths]; ++s)
    if (std::find(distinct.begin(), distinct.end(), step_slots[s]) == distinct.end()) distinct.push_back(step_slots[s]);
  if (distinct.size() > 32) return MSI_E_UNSUPPORTED;
  return msi_bits_paths_claim(p, n_paths, path_off, step_slots, bucket, universe,
                              p->region_counts + (size_t)region * MSI_BITS_REGION_PATHS);
}

int32_t msi_bits_paths_collect(msi_bits *p, uint32_t n_regions, uint64_t *counts) {
  if (!n_regions || n_regions > MSI_BITS_PATH_REGIONS) return MSI_E_INVALID;
  std::copy(p->region_counts, p->region_counts + (size_t)n_regions * MSI_BITS_REGION_PATHS, counts);
  return MSI_OK;
}


extern "C" {

msi_doc_keys *mock_doc_keys_create(const uint32_t *keys, uint64_t n) {
  msi_doc_keys *k = new msi_doc_keys();
  k->keys.assign(keys, keys + n);
  return k;
}
void mock_doc_keys_destroy(msi_doc_keys *k) { delete k; }

int32_t msi_bits_order_next(msi_bits *p, const msi_doc_keys *keys, uint32_t universe, uint32_t bucket, uint32_t *out_key,
                            uint64_t *out_count) {
  if (keys->keys.size() != p->n_docs) return MSI_E_INVALID;
  uint32_t best = 0xFFFFFFFFu;
  for (uint64_t d = 0; d < p->n_docs; ++d)
    if ((p->slot(universe)[d >> 6] >> (d & 63)) & 1ull) best = std::min(best, keys->keys[d]);
  std::fill(p->slot(bucket), p->slot(bucket) + p->n_words, 0ull);
  uint64_t n = 0;
  for (uint64_t d = 0; d < p->n_docs; ++d)
    if (((p->slot(universe)[d >> 6] >> (d & 63)) & 1ull) && keys->keys[d] == best) {
      p->slot(bucket)[d >> 6] |= 1ull << (d & 63);
      p->slot(universe)[d >> 6] &= ~(1ull << (d & 63));
      ++n;
    }
  *out_key = best;
  *out_count = n;
  return MSI_OK;
}

msi_doc_values *mock_doc_values_create(const uint64_t *offsets, const uint32_t *values, uint64_t n_docs, uint32_t n_values) {
  msi_doc_values *v = new msi_doc_values();
  v->n_values = n_values;
  for (uint64_t d = 0; d < n_docs; ++d) v->per_doc.emplace_back(values + offsets[d], values + offsets[d + 1]);
  return v;
}
void mock_doc_values_destroy(msi_doc_values *v) { delete v; }

static bool mock_bit(msi_bits *p, uint32_t s, uint64_t d) { return (p->slot(s)[d >> 6] >> (d & 63)) & 1ull; }

// the reference's loop as it is written (distinct.rs:19-62): the product's parallel rounds must agree with it
int32_t msi_bits_distinct(msi_bits *p, const msi_doc_values *vals, uint32_t candidates, uint32_t remaining, uint32_t excluded,
                          uint64_t *out_remaining, uint32_t *out_rounds) {
  if (vals->per_doc.size() != p->n_docs) return MSI_E_INVALID;
  std::vector<char> taken(vals->n_values, 0);
  std::fill(p->slot(remaining), p->slot(remaining) + p->n_words, 0ull);
  uint64_t n = 0;
  for (uint64_t d = 0; d < p->n_docs; ++d) {
    if (!mock_bit(p, candidates, d)) continue;
    bool skip = false;
    for (uint32_t v : vals->per_doc[d]) skip |= taken[v] != 0;
    if (skip) continue;
    for (uint32_t v : vals->per_doc[d]) taken[v] = 1;
    p->slot(remaining)[d >> 6] |= 1ull << (d & 63);
    ++n;
  }
  std::fill(p->slot(candidates), p->slot(candidates) + p->n_words, 0ull);
  if (excluded != MSI_BITS_NO_SLOT) {
    std::fill(p->slot(excluded), p->slot(excluded) + p->n_words, 0ull);
    for (uint64_t d = 0; d < p->n_docs; ++d)
      for (uint32_t v : vals->per_doc[d])
        if (taken[v]) p->slot(excluded)[d >> 6] |= 1ull << (d & 63);
  }
  *out_remaining = n;
  if (out_rounds) *out_rounds = 1;
  return MSI_OK;
}

int32_t msi_bits_distinct_excluded(msi_bits *p, const msi_doc_values *vals, uint32_t kept, uint32_t excluded) {
  std::vector<char> taken(vals->n_values, 0);
  for (uint64_t d = 0; d < p->n_docs; ++d)
    if (mock_bit(p, kept, d))
      for (uint32_t v : vals->per_doc[d]) taken[v] = 1;
  std::fill(p->slot(excluded), p->slot(excluded) + p->n_words, 0ull);
  for (uint64_t d = 0; d < p->n_docs; ++d)
    for (uint32_t v : vals->per_doc[d])
      if (taken[v]) p->slot(excluded)[d >> 6] |= 1ull << (d & 63);
  return MSI_OK;
}

int32_t msi_bits_andnot_many_count(msi_bits *p, uint32_t removed, uint32_t n, const uint32_t *slots, uint64_t *counts) {
  for (uint32_t k = 0; k < n; ++k) {
    for (uint64_t i = 0; i < p->n_words; ++i) p->slot(slots[k])[i] &= ~p->slot(removed)[i];
    counts[k] = popcount_slot(p, slots[k]);
  }
  return MSI_OK;
}

int32_t msi_bits_set_from_docids(msi_bits *p, uint32_t slot, const uint32_t *docids, uint64_t n) {
  std::fill(p->slot(slot), p->slot(slot) + p->n_words, 0ull);
  for (uint64_t i = 0; i < n; ++i)
    if (docids[i] < p->n_docs) p->slot(slot)[docids[i] >> 6] |= 1ull << (docids[i] & 63);
  return MSI_OK;
}

msi_geo_points *mock_geo_points_create(const double *lat_lng, uint64_t n_docs) {
  msi_geo_points *g = new msi_geo_points();
  g->lat_lng.assign(lat_lng, lat_lng + 2 * n_docs);
  return g;
}
void mock_geo_points_destroy(msi_geo_points *g) { delete g; }

int32_t msi_bits_geo_list(msi_bits *p, const msi_geo_points *gp, uint32_t universe, double lat, double lng, uint32_t cap,
                          uint32_t *out_docids, double *out_distance, uint64_t *out_total) {
  const double D2R = 3.14159265358979323846 / 180.0;
  uint64_t n = 0;
  for (uint64_t d = 0; d < p->n_docs; ++d)
    if (mock_bit(p, universe, d) && gp->lat_lng[2 * d] == gp->lat_lng[2 * d]) {
      if (n < cap) {
        const double lat2 = gp->lat_lng[2 * d], lng2 = gp->lat_lng[2 * d + 1];
        const double phi1 = lat * D2R, phi2 = lat2 * D2R, lam1 = lng * D2R, lam2 = lng2 * D2R;
        const double total = (1.0 - cos(phi2 - phi1)) / 2.0 + cos(phi1) * cos(phi2) * ((1.0 - cos(lam2 - lam1)) / 2.0);
        out_docids[n] = (uint32_t)d;
        out_distance[n] = round(2.0 * 6371e3 * asin(sqrt(total)) * 1000.0) / 1000.0;
      }
      ++n;
    }
  *out_total = n;
  return MSI_OK;
}

// documents/geo_sort.rs:150-224 over a cache in exact (distance, docid) order, written as the loop it is
int32_t msi_bits_geo_next(msi_bits *p, const msi_geo_points *gp, uint32_t universe, uint32_t bucket, uint32_t scratch,
                          double lat, double lng, int32_t ascending, uint32_t max_bucket_size, double margin,
                          uint32_t *out_first_docid, uint64_t *out_count) {
  (void)scratch;
  const double D2R = 3.14159265358979323846 / 180.0;
  auto distance = [&](double lat2, double lng2) {  // lib.rs:388-393 -> geoutils haversine_distance_to
    const double phi1 = lat * D2R, phi2 = lat2 * D2R, lam1 = lng * D2R, lam2 = lng2 * D2R;
    const double total = (1.0 - cos(phi2 - phi1)) / 2.0 + cos(phi1) * cos(phi2) * ((1.0 - cos(lam2 - lam1)) / 2.0);
    return round(2.0 * 6371e3 * asin(sqrt(total)) * 1000.0) / 1000.0;
  };
  std::vector<std::pair<double, uint32_t>> cache;
  for (uint64_t d = 0; d < p->n_docs; ++d)
    if (mock_bit(p, universe, d) && gp->lat_lng[2 * d] == gp->lat_lng[2 * d]) {
      const double dist = distance(gp->lat_lng[2 * d], gp->lat_lng[2 * d + 1]);
      cache.push_back({ascending ? dist : -dist, (uint32_t)d});
    }
  std::fill(p->slot(bucket), p->slot(bucket) + p->n_words, 0ull);
  *out_first_docid = 0xFFFFFFFFu;
  *out_count = 0;
  if (cache.empty()) return MSI_OK;
  std::sort(cache.begin(), cache.end());
  const uint64_t cap = max_bucket_size ? max_bucket_size : 1000;
  uint64_t n = 0;
  for (auto &e : cache) {
    if (fabs(cache[0].first - e.first) > margin || n == cap) break;
    p->slot(bucket)[e.second >> 6] |= 1ull << (e.second & 63);
    p->slot(universe)[e.second >> 6] &= ~(1ull << (e.second & 63));
    ++n;
  }
  *out_first_docid = cache[0].second;
  *out_count = n;
  return MSI_OK;
}

int32_t msi_bits_fill(msi_bits *p, uint32_t slot, int32_t ones) {
  uint64_t *d = p->slot(slot);
  std::fill(d, d + p->n_words, 0ull);
  if (ones) {
    const uint64_t full = p->n_docs / 64;
    std::fill(d, d + full, ~0ull);
    if (p->n_docs % 64) d[full] = ~0ull >> (64 - p->n_docs % 64);
  }
  return MSI_OK;
}

int32_t msi_bits_op(msi_bits *p, uint32_t dst, uint32_t a, uint32_t b, int32_t op) {
  for (uint64_t i = 0; i < p->n_words; ++i) {
    const uint64_t x = p->slot(a)[i], y = p->slot(b)[i];
    p->slot(dst)[i] = op == MSI_BITS_AND ? (x & y) : op == MSI_BITS_OR ? (x | y) : op == MSI_BITS_ANDNOT ? (x & ~y) : (x ^ y);
  }
  return MSI_OK;
}

int32_t msi_bits_op_count(msi_bits *p, uint32_t dst, uint32_t a, uint32_t b, int32_t op, uint64_t *out_count) {
  msi_bits_op(p, dst, a, b, op);
  *out_count = popcount_slot(p, dst);
  return MSI_OK;
}

int32_t msi_bits_count(msi_bits *p, uint32_t slot, uint64_t *out) {
  *out = popcount_slot(p, slot);
  return MSI_OK;
}

int32_t msi_bits_first_k(msi_bits *p, uint32_t slot, uint32_t k, uint32_t *out_docids, uint32_t *out_n) {
  uint32_t n = 0;
  const uint64_t *s = p->slot(slot);
  for (uint64_t w = 0; w < p->n_words && n < k; ++w) {
    uint64_t v = s[w];
    while (v && n < k) {
      out_docids[n++] = (uint32_t)(w * 64 + (uint64_t)__builtin_ctzll(v));
      v &= v - 1;
    }
  }
  *out_n = n;
  return MSI_OK;
}


int32_t msi_dict_lookup(msi_dict *d, const msi_typo_query *q, uint32_t n, uint32_t cap_one, uint32_t cap_two,
                        uint32_t *out_one_idx, uint32_t *out_one_cnt, uint32_t *out_two_idx, uint32_t *out_two_cnt) {
  for (uint32_t i = 0; i < n; ++i) {
    const int32_t st = d->lookup(q[i].word, q[i].len, q[i].max_typos, q[i].is_prefix, cap_one, cap_two,
                                 out_one_idx + (size_t)i * cap_one, out_one_cnt + i, out_two_idx + (size_t)i * cap_two,
                                 out_two_cnt + i);
    if (st != MSI_OK) return st;
  }
  return MSI_OK;
}

// ---- constructors for the test harness -----------------------------------------------------------
msi_bits *mock_bits_create(uint64_t n_docs, uint32_t n_slots) {
  msi_bits *p = new msi_bits();
  p->n_docs = n_docs;
  p->n_words = std::max<uint64_t>(2, ((n_docs + 127) / 128) * 2);
  p->n_slots = n_slots;
  p->pool.assign((size_t)p->n_words * n_slots, 0xDEADBEEFDEADBEEFull);  // stale content must never leak into results
  return p;
}
void mock_bits_destroy(msi_bits *p) { delete p; }
msi_dict *mock_dict_create(const uint8_t *concat, const uint32_t *offsets, uint32_t n, mock_lookup_fn lookup) {
  msi_dict *d = new msi_dict();
  for (uint32_t i = 0; i < n; ++i) d->words.emplace_back((const char *)concat + offsets[i], offsets[i + 1] - offsets[i]);
  d->lookup = lookup;
  return d;
}
void mock_dict_destroy(msi_dict *d) { delete d; }

}  // extern "C"
