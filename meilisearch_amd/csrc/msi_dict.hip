// msi_dict.hip — S2: batched typo-tolerant term lookup over the words
// dictionary staged in HBM, gfx950.
//
// Replaces find_one_typo_derivations / find_one_two_typo_derivations
// (crates/milli/src/search/new/query_term/compute_derivations.rs:75-168), i.e.
// `fst.search_with_state(Levenshtein DFA ∩/∪ StartsWith)` with the 150/50 caps
// (search/new/limits.rs:7-9).  Design:
//
//   HBM layout  the sorted dictionary (what fst.stream() yields) as 16-byte slots (one
//               word per slot, zero padded; longer words keep their bytes in the flat
//               array), a 64-bit char-presence signature and the char / byte lengths
//               per word, and the table of first-char blocks.
//   work shape  the FST ∩ DFA walk never leaves the subtrees the automaton can still
//               accept; the sorted dictionary gives the same pruning as index ranges:
//               (1) words that share the query's first char are ONE range [lo, hi)
//               (first-letter rule): only that range is scanned — lane = word, a
//               10-byte filter (length window + signature), survivors compacted into
//               dense lanes for the banded (5 diagonal) OSA DP over code points;
//               (2) words with another first char match only through one edit on
//               position 0, i.e. they are exact strings (string prefixes under the
//               prefix rule): binary searches, no scan.
//   ordering    one workgroup per query; its waves own contiguous pieces of the range,
//               so ballots give per-wave lists already in fst stream order and their
//               concatenation is the stream; the cap logic runs in the same launch.
//
// Closed form of the reference's sequential cap logic (derived in DESIGN.md):
//   S1/S2 = same-first-char words at distance exactly 1 / 2, X = other-first-char
//   words at distance <= 1, all in dictionary order;
//   two  = first cap_two of (X ∪ S2);  t* = its last element if it is full;
//   one  = first cap_one of (S1 ∪ {x in X : x > t*}).
#include <string.h>

#include <algorithm>
#include <chrono>
#include <condition_variable>
#include <linux/futex.h>
#include <sys/syscall.h>
#include <unistd.h>
#include <climits>
#include <vector>

#include <time.h>

#include "msi_common.h"
#include "msi_vm.h"

typedef unsigned long long u64;

namespace {

constexpr int LW = 4;             // waves per workgroup of the matcher kernel; one workgroup walks one query's dictionary range
constexpr int LT = LW * 64;       // (round 4: 4 waves instead of 8 — a CU then holds as many workgroups as its registers allow waves per
                                  //  SIMD, and the kernel is bound by how many scanning waves are resident: see dict_lookup_kernel)
constexpr int XW = 8;             // waves per workgroup of the other-first-letter kernel
constexpr int XT = XW * 64;
constexpr int QSTRIDE = 256;      // code points reserved per query
constexpr int PQ = 256;           // survivor ring entries per wave (power of two, >= 63 + 64)
constexpr int XB = XT / 2;        // first-char blocks searched per round of the other-first-char step
constexpr int DINF = 1 << 20;
constexpr uint32_t QNONE = 0xFFFFFFFFu;  // "no such query char": never equals a word char

struct alignas(64) QueryMeta {   // one 64-byte line per query
  uint32_t m;        // chars
  uint32_t budget;   // 1 or 2 (0 = skip)
  uint32_t prefix;
  uint32_t qlen;     // bytes
  uint32_t lo, hi;   // dictionary range of the words starting with the query's first char
  u64 sig;           // char-presence signature (bit = sig_bit(code point))
  uint32_t l0, l1;   // byte lengths of the first and second char (l1 = 0: one char only)
  uint32_t q0, q1;   // their code points
  u64 _pad[2];
};
static_assert(sizeof(QueryMeta) == 64, "QueryMeta is one 64-byte line");

// The filter word of a dictionary word: ONE 8-byte load per word in the range scan (round 3 read a 64-bit signature and
// a 16-bit length word: two loads, 10 bytes).  bits 0..7 char count (capped at 255), bit 8 "a word of at least one byte",
// bits 9..63 the char-presence signature: bit 9 + hash55(code point) set for every char of the word.
constexpr u64 FILT_VALID = 1ull << 8;
constexpr u64 FILT_SIG = ~0x1FFull;
// The 55 signature positions: one each for a-z and 0-9 (what nearly every char of a normalised word is: no collisions
// among them), 19 shared by every other code point.  (A plain 55-way hash of all code points let a third more words
// through the filter than round 3's 64-bit one: 2 695 instead of 2 029 matcher pairs per query at C3.)
__host__ __device__ __forceinline__ uint32_t sig_bit(uint32_t cp) {
  if (cp - (uint32_t)'a' < 26u) return 9u + (cp - (uint32_t)'a');
  if (cp - (uint32_t)'0' < 10u) return 9u + 26u + (cp - (uint32_t)'0');
  return 9u + 36u + ((((cp * 0x9E3779B1u) >> 16) * 19u) >> 16);
}
__device__ __forceinline__ u64 pair_key(uint32_t a, uint32_t b) { return (u64)a | ((u64)b << 32); }

__device__ __forceinline__ uint32_t utf8_len(uint32_t b0) {
  return b0 < 0x80 ? 1u : (b0 < 0xE0 ? 2u : (b0 < 0xF0 ? 3u : 4u));
}

// ---- word readers ----------------------------------------------------------

// 16-byte word held in registers as a little-endian byte queue.
template <bool ASCII>
struct SlotReader {
  uint32_t w0, w1, w2, w3;
  __device__ __forceinline__ uint32_t next_char() {
    const uint32_t lo = w0;
    const uint32_t b0 = lo & 0xFF;
    if (ASCII) {
      w0 = __funnelshift_r(w0, w1, 8);
      w1 = __funnelshift_r(w1, w2, 8);
      w2 = __funnelshift_r(w2, w3, 8);
      w3 >>= 8;
      return b0;
    }
    const uint32_t clen = utf8_len(b0);
    const uint32_t b1 = (lo >> 8) & 0x3F, b2 = (lo >> 16) & 0x3F, b3 = (lo >> 24) & 0x3F;
    const uint32_t cp2 = ((b0 & 0x1F) << 6) | b1;
    const uint32_t cp3 = ((b0 & 0x0F) << 12) | (b1 << 6) | b2;
    const uint32_t cp4 = ((b0 & 0x07) << 18) | (b1 << 12) | (b2 << 6) | b3;
    const uint32_t cp = clen == 1 ? b0 : (clen == 2 ? cp2 : (clen == 3 ? cp3 : cp4));
    const uint32_t sh = (clen - 1) * 8;  // 0..24, then a constant 8
    uint32_t a0 = __funnelshift_r(w0, w1, sh), a1 = __funnelshift_r(w1, w2, sh),
             a2 = __funnelshift_r(w2, w3, sh), a3 = w3 >> sh;
    w0 = __funnelshift_r(a0, a1, 8);
    w1 = __funnelshift_r(a1, a2, 8);
    w2 = __funnelshift_r(a2, a3, 8);
    w3 = a3 >> 8;
    return cp;
  }
};

// Word of any length read byte-wise from the flat dictionary bytes (long words).
struct FlatReader {
  const uint8_t *p;
  const uint8_t *end;
  __device__ __forceinline__ uint32_t next_char() {
    if (p >= end) return 0;
    const uint32_t b0 = *p;
    const uint32_t clen = utf8_len(b0);
    uint32_t cp = clen == 1 ? b0 : (clen == 2 ? (b0 & 0x1F) : (clen == 3 ? (b0 & 0x0F) : (b0 & 0x07)));
    for (uint32_t e = 1; e < clen; ++e) cp = (cp << 6) | ((p + e < end ? p[e] : 0) & 0x3F);
    p += clen;
    return cp;
  }
};

__device__ __forceinline__ int wave_max_i32(int v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = max(v, __shfl_xor(v, o));
  return v;
}

// Query code points: every lane of the workgroup works on the SAME query (staged in LDS; reads broadcast).
struct QChars {
  const uint32_t *q;
  int m;
  __device__ __forceinline__ uint32_t at(int t) const {  // t is wave-uniform
    if (t < 0 || t >= m) return QNONE;
    return q[t];
  }
};

// Banded optimal-string-alignment distance (insert / delete / substitute /
// adjacent transposition, each cost 1 — levenshtein_automata 0.2.1 with
// transposition_cost_one = true, crates/milli/src/search/mod.rs:32-34) between
// the workgroup's query (m chars) and the lane's word (nch chars).  The band is 5
// diagonals (|i-j| <= 2); the result is exact whenever the true value is <= K
// (K = 1 or 2) and > K otherwise.  prefix: minimum over the prefixes of the word
// (build_prefix_dfa, search/mod.rs:572-573).  All lanes step through the word
// positions together, so the cell index i = j + d - 2 is wave-uniform and the
// query window slides by one LDS read per step.
template <class Reader>
__device__ __forceinline__ int osa_pair(Reader rd, int nch, bool active, const QChars &qs, int K, bool prefix) {
  constexpr int W = 5, KB2 = 2;
  const int m = qs.m;
  int prev2[W], prev[W], cur[W];
#pragma unroll
  for (int d = 0; d < W; ++d) {
    prev2[d] = DINF;
    const int i = d - KB2;
    prev[d] = (i >= 0 && i <= m) ? i : DINF;
  }
  int best = (prefix && m <= K) ? m : DINF;  // the empty prefix
  const int ne = active ? min(nch, m + K) : 0;
  const int nmax = wave_max_i32(ne);
  // qw[t] = q[j - 4 + t] at step j (0-based chars); cell d uses q[i-1] = qw[d+1], q[i-2] = qw[d]
  uint32_t qw[W + 1];
#pragma unroll
  for (int t = 0; t < W + 1; ++t) qw[t] = qs.at(t - 3);  // step j = 1
  uint32_t wprev = 0xFFFFFFFEu;
  for (int j = 1; j <= nmax; ++j) {
    const uint32_t wc = rd.next_char();
    const bool upd = j <= ne;
    int bandmin = DINF;
#pragma unroll
    for (int d = 0; d < W; ++d) {
      const int i = j + d - KB2;  // wave-uniform
      int v;
      if (i < 0) {
        v = DINF;
      } else if (i == 0) {
        v = j;
      } else {
        const uint32_t qi = qw[d + 1];
        v = prev[d] + (qi != wc ? 1 : 0);
        if (d > 0) v = min(v, cur[d - 1] + 1);
        if (d < W - 1) v = min(v, prev[d + 1] + 1);
        if (i >= 2 && j >= 2) {
          const uint32_t qim = qw[d];
          if (qi == wprev && qim == wc) v = min(v, prev2[d] + 1);
        }
        if (i > m) v = DINF;  // per lane
      }
      cur[d] = v;
      bandmin = min(bandmin, v);
    }
#pragma unroll
    for (int d = 0; d < W; ++d) {
      prev2[d] = upd ? prev[d] : prev2[d];
      prev[d] = upd ? cur[d] : prev[d];
    }
    wprev = upd ? wc : wprev;
    if (prefix) {
      const int dm = m - j + KB2;  // per lane
#pragma unroll
      for (int d = 0; d < W; ++d)
        if (dm == d && upd) best = min(best, cur[d]);
    }
    if (!__any(upd && bandmin <= K)) break;  // every still-running lane is out of budget
#pragma unroll
    for (int t = 0; t < W; ++t) qw[t] = qw[t + 1];
    qw[W] = qs.at(j + 2);
  }
  if (prefix) return best;
  const int dm = m - nch + KB2;  // per lane
  int res = DINF;
#pragma unroll
  for (int d = 0; d < W; ++d)
    if (dm == d) res = prev[d];
  return (active && nch <= m + K) ? res : DINF;
}

// ---- bit-parallel OSA (the survivors' matcher since round 4) ------------------------------------------------------
// Hyyro's bit-vector recurrence for the Damerau / optimal-string-alignment distance ("A bit-vector algorithm for computing
// Levenshtein and Damerau edit distances", 2003): pattern = the workgroup's QUERY (one bit per query char, T = u32 for
// m <= 32, u64 for m <= 64; longer queries take the banded DP above), text = the lane's dictionary word.  A column of the
// DP table is two bit vectors of vertical deltas (VP / VN); one word char advances it in ~20 wave instructions whatever
// m is, against the 78-102 of the five-diagonal band.  The tracked score is D[m][j] — the distance between the whole query
// and the word's first j chars — which is exactly what both automata need: its value at j = |word| (build_dfa), its
// minimum over j (build_prefix_dfa: `sunflowering` costs 0 under the prefix query `sunflower`).
//
// PEq[c] = the query positions holding char c: a direct table for ASCII, an open-addressing table for other code points
// (<= 64 distinct keys in 128 slots), both in LDS, built once per query by the workgroup.
constexpr uint32_t PEQ_EMPTY = 0xFFFFFFFFu;
struct PEq {
  const u64 *ascii;      // [128]
  const uint32_t *key;   // [128]
  const u64 *val;        // [128]
  __device__ __forceinline__ u64 at(uint32_t c) const {
    if (c < 128u) return ascii[c];
    uint32_t h = (c * 0x9E3779B1u) >> 25;
    for (;;) {
      const uint32_t k = key[h];
      if (k == c) return val[h];
      if (k == PEQ_EMPTY) return 0ull;
      h = (h + 1u) & 127u;
    }
  }
};

template <class T, class Reader>
__device__ __forceinline__ int osa_pair_bits(Reader rd, int nch, bool active, const PEq &pq, int m, int K, bool prefix) {
  const T top = (T)1 << (m - 1);
  T VP = (T)(~(T)0), VN = 0, D0 = 0, PMp = 0;   // (bits above m - 1 never reach a bit below them)
  int score = m, best = m;
  const int ne = active ? min(nch, m + K) : 0;
  const int nmax = wave_max_i32(ne);
  for (int j = 1; j <= nmax; ++j) {
    const uint32_t wc = rd.next_char();
    const bool upd = j <= ne;
    const T PM = (T)pq.at(upd ? wc : 0u);
    const T TR = (((T)(~D0) & PM) << 1) & PMp;
    const T D0n = ((((PM & VP) + VP) ^ VP) | PM | VN) | TR;
    const T HP = VN | (T)(~(D0n | VP));
    const T HN = D0n & VP;
    const int sc = score + ((HP & top) ? 1 : 0) - ((HN & top) ? 1 : 0);
    const T X = (T)(HP << 1) | (T)1;
    const T VPn = (T)(HN << 1) | (T)(~(D0n | X));
    const T VNn = D0n & X;
    if (upd) {
      D0 = D0n;
      VP = VPn;
      VN = VNn;
      PMp = PM;
      score = sc;
      best = min(best, sc);
    }
  }
  if (!active) return DINF;
  if (prefix) return best;
  return nch <= m + K ? score : DINF;
}

// ---- lookup kernel -----------------------------------------------------------------

struct DictArgs {
  const uint4 *slots;        // [n_words] first 16 bytes of every word, zero padded
  const u64 *filt;           // [n_words] filter word: char count | valid << 8 | char-presence signature << 9
  const uint16_t *wmeta;     // [n_words] char count | byte length << 8
  const uint8_t *flat;       // concatenated bytes
  const uint32_t *offs;      // [n_words+1]
  const uint32_t *fc_start;  // [n_fc+1] first word of every first-char block (non-empty words), ascending
  uint32_t n_fc, n_words;
  const QueryMeta *qm;       // [nq]
  const uint32_t *qchars;    // [nq][QSTRIDE]
  const uint8_t *qbytes;
  const uint32_t *qoff;
  uint32_t nq;
  uint32_t cap1, cap2, capx;
  uint32_t *wlists;          // [grid][LW][cap1+cap2] per-wave hit lists (distance 1 | distance 2)
  // The unit of work of dict_lookup_kernel is a SLICE of a query's first-letter range (slice_tiles 64-word tiles): ranges
  // differ 20x between first letters, and with one workgroup per query a launch ended with a few workgroups walking the
  // longest ranges (8 192 queries: 2.2 M words/s, 32 768: 4.1 M).  dict_plan_kernel lays the units out.
  const uint32_t *unit_off;  // [nq + 1] first unit of every query; unit_off[nq] = units of the batch
  const uint32_t *unit_q;    // [units] the query of every unit
  uint32_t *slice_done;      // [nq] units of the query that have handed in their lists
  uint32_t *ulists;          // [units][cap1 + cap2] a unit's hits in dictionary order (distance 1 | distance 2), capped
  uint32_t *ucnt;            // [units][2]
  uint32_t slice_tiles;
  uint32_t *xq;              // [nq][capx] other-first-char words at distance <= 1 (dict_other_kernel) ...
  uint32_t *xq_cnt;          // [nq]       ... and how many
  uint32_t *ticket;          // next query to take
  uint32_t m_lo, m_hi;       // this launch matches the queries of m_lo..m_hi chars (the other launch takes the rest)
  const uint32_t *n_long;    // queries above 64 chars in the batch (dict_prep_kernel); null = not consulted
  u64 *pairs;                // stats: (query, word) pairs that reached a DP lane
  u64 *prof;                 // MSI_DICT_PROFILE: thread 0's wall-clock ticks (100 MHz) per phase, summed over queries (else null)
  uint32_t *out_one, *out_one_cnt, *out_two, *out_two_cnt;
  uint32_t defer_caps;       // 1: the cap logic runs in dict_caps_kernel, after dict_other_kernel AND the lookup launches (round 6)
};

// A pattern assembled from up to three pieces of the query's bytes (the one-edit shapes that change the first
// char: delete q0, swap q0 q1, and — behind a dictionary first char — substitute q0 / insert before q0).
struct Pattern {
  const uint8_t *qb;
  uint32_t o0, l0, o1, l1, o2, l2, len;
  __device__ __forceinline__ uint32_t at(uint32_t i) const {
    if (i < l0) return qb[o0 + i];
    i -= l0;
    if (i < l1) return qb[o1 + i];
    return qb[o2 + i - l1];
  }
};

// word[skip..] against the pattern: -1 the tail sorts before every string the pattern is a prefix of, 0 the
// pattern is a prefix of the tail (*exact: they are equal), +1 the tail sorts after all of them.
__device__ __forceinline__ int cmp_tail(const DictArgs &a, uint32_t idx, uint32_t skip, const Pattern &p, bool *exact) {
  const uint32_t o = a.offs[idx] + skip, tl = a.offs[idx + 1] - o;
  const uint32_t n = tl < p.len ? tl : p.len;
  for (uint32_t i = 0; i < n; ++i) {
    const uint32_t wb = a.flat[o + i], pb = p.at(i);
    if (wb != pb) return wb < pb ? -1 : 1;
  }
  if (tl < p.len) return -1;
  *exact = tl == p.len;
  return 0;
}

// The words of [s, e) (all sharing their first `skip` bytes) whose tail equals the pattern (prefix = false) or
// starts with it (prefix = true): a contiguous index range of the sorted dictionary.
__device__ __forceinline__ void find_range(const DictArgs &a, uint32_t s, uint32_t e, uint32_t skip, const Pattern &p,
                                           bool prefix, uint32_t *ra, uint32_t *rb) {
  bool ex = false;
  uint32_t lo = s, hi = e;
  while (lo < hi) {
    const uint32_t mid = lo + (hi - lo) / 2;
    if (cmp_tail(a, mid, skip, p, &ex) < 0) lo = mid + 1;
    else hi = mid;
  }
  *ra = *rb = lo;
  if (lo >= e) return;
  ex = false;
  if (cmp_tail(a, lo, skip, p, &ex) != 0) return;
  if (!prefix) {
    if (ex) *rb = lo + 1;   // the shortest word with this prefix sorts first: an exact match sits at the lower bound
    return;
  }
  uint32_t l2 = lo + 1;
  hi = e;
  while (l2 < hi) {
    const uint32_t mid = l2 + (hi - l2) / 2;
    if (cmp_tail(a, mid, skip, p, &ex) == 0) l2 = mid + 1;
    else hi = mid;
  }
  *rb = l2;
}

// The hit lists of a workgroup's waves as one ascending stream: the waves' pieces of the range are contiguous
// dictionary ranges, so concatenation in wave order is dictionary order.
struct WaveLists {
  const uint32_t *base;
  uint32_t stride, off, cls, w, i;
  const uint32_t (*cnt)[2];
  uint32_t n;    // lists (the units of a query, in order)
  // (device-scope loads: the lists were stored by other workgroups — dict_lookup_body)
  __device__ __forceinline__ uint32_t count_of(uint32_t ww) const {
    return __hip_atomic_load(const_cast<uint32_t *>(&cnt[ww][cls]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  __device__ __forceinline__ void settle() { while (w < n && i >= count_of(w)) { ++w; i = 0; } }
  __device__ __forceinline__ bool done() const { return w >= n; }
  __device__ __forceinline__ uint32_t peek() const {
    return __hip_atomic_load(const_cast<uint32_t *>(&base[(size_t)w * stride + off + i]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  __device__ __forceinline__ void pop() { ++i; settle(); }
};

// One workgroup per query (queries are handed out by a ticket counter), three steps:
//   scan    the words that share the query's first char are ONE index range [lo, hi) of the sorted dictionary
//           (first-letter rule, compute_derivations.rs:120-124); the 8 waves split it into contiguous pieces.
//           filter: lane = word, 10 bytes per word (signature + lengths): length window of K edits and the
//           char-presence signature (every edit introduces at most one new char, so popc(sig_q & ~sig_w) <= K and,
//           without the prefix rule, popc(sig_w & ~sig_q) <= K are necessary).  Survivors are queued with ballots in
//           a per-wave LDS ring; whenever 64 are queued every lane takes ONE word and runs the banded OSA DP against
//           the (wave-uniform) query — dense DP lanes.  Hits are appended in dictionary order to the wave's lists.
//   search  words with ANOTHER first char match only at distance <= 1, through one edit on position 0: substitute
//           q0 (c + q[1..]), insert before q0 (c + q), delete q0 (q[1..]), swap q0 q1 (q1 q0 q[2..]).  These are
//           exact strings (prefix rule: string prefixes), so they are binary searches in the sorted dictionary —
//           two per dictionary first char c, two more for the shapes without c — instead of a scan.
//   caps    the reference's sequential cap logic in closed form (header of this file) over the three lists.
// Units of the batch: query q gets ceil(tiles of its first-letter range / slice_tiles) of them (at least one: the cap logic
// runs for every query).  One workgroup: per-thread sums over contiguous queries, a scan over the threads, then the offsets
// and the unit -> query table.
__global__ __launch_bounds__(1024) void dict_plan_kernel(const QueryMeta *__restrict__ qm, uint32_t nq, uint32_t slice_tiles,
                                                         uint32_t *__restrict__ unit_off, uint32_t *__restrict__ unit_q,
                                                         uint32_t *__restrict__ slice_done, uint32_t max_units) {
  __shared__ uint32_t s_part[1024];
  const uint32_t tid = threadIdx.x;
  const uint32_t per = (nq + 1023) / 1024, q0 = min(nq, tid * per), q1 = min(nq, q0 + per);
  auto units_of = [&](uint32_t q) -> uint32_t {
    const QueryMeta m = qm[q];
    if (m.budget == 0 || m.hi <= m.lo) return 1u;
    const uint32_t n_t = ((m.hi + 63) >> 6) - (m.lo >> 6);
    return (n_t + slice_tiles - 1) / slice_tiles;
  };
  uint32_t mine = 0;
  for (uint32_t q = q0; q < q1; ++q) mine += units_of(q);
  s_part[tid] = mine;
  __syncthreads();
  for (uint32_t o = 1; o < 1024; o <<= 1) {   // inclusive scan
    const uint32_t v = tid >= o ? s_part[tid - o] : 0u;
    __syncthreads();
    s_part[tid] += v;
    __syncthreads();
  }
  uint32_t at = s_part[tid] - mine;
  for (uint32_t q = q0; q < q1; ++q) {
    const uint32_t n = units_of(q);
    unit_off[q] = at;
    slice_done[q] = 0;
    for (uint32_t i = 0; i < n; ++i)
      if (at + i < max_units) unit_q[at + i] = q;
    at += n;
  }
  if (tid == 1023) unit_off[nq] = min(s_part[1023], max_units);   // (the host sized the lists for max_units: never exceeded by construction)
}

// The words with ANOTHER first char than the query's: they match only at distance <= 1, through one edit on position 0 —
// exact strings (string prefixes under the prefix rule), i.e. binary searches in the sorted dictionary.  A kernel of its own
// since round 4: inside dict_lookup_kernel its registers (patterns, comparison loops) were what kept that kernel — bound
// by how many range-scanning waves are resident — at one workgroup per CU.  One workgroup per budget-2 query (ticket);
// the result, in dictionary order, goes to a.xq[q] / a.xq_cnt[q], read by the cap logic of dict_lookup_kernel.
__global__ __launch_bounds__(XT) void dict_other_kernel(DictArgs a) {
  __shared__ uint8_t s_qb[256];
  __shared__ uint32_t s_xr[XB][4];     // per first-char block of the round: (a) range, (c) range
  __shared__ u64 s_xany[XW];
  __shared__ uint32_t s_ext[2][2];     // the two shapes without a dictionary first char: delete q0, swap q0 q1
  __shared__ uint32_t s_query, s_xdone;
  const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (;;) {
    __syncthreads();
    if (tid == 0) {
      s_query = atomicAdd(a.ticket, 1u);
      s_xdone = 0;
    }
    __syncthreads();
    const uint32_t q = s_query;
    if (q >= a.nq) break;
    const QueryMeta qm = a.qm[q];
    const int K = (int)qm.budget;
    const int m = (int)qm.m;
    const bool prefix = qm.prefix != 0;
    uint32_t *const xl = a.xq + (size_t)q * a.capx;
    if (!(K == 2 && a.n_fc > 0)) {
      if (tid == 0) a.xq_cnt[q] = 0;
      continue;
    }
    for (uint32_t i = tid; i < qm.qlen; i += XT) s_qb[i] = a.qbytes[a.qoff[q] + i];
    __syncthreads();
    // ---- search: other first chars, distance <= 1 (only the two-typo automaton accepts them) --------
    uint32_t xn = 0;  // thread 0's
    if (K == 2 && a.n_fc > 0) {
      Pattern pt;
      pt.qb = s_qb;
      if (tid < 2) {
        uint32_t ra = 0, rb = 0;
        if (m >= 2 && qm.q1 != qm.q0) {
          if (tid == 0) {          // delete q0: q[1..]
            pt.o0 = qm.l0; pt.l0 = qm.qlen - qm.l0; pt.o1 = pt.l1 = pt.o2 = pt.l2 = 0;
          } else {                 // swap q0 q1: q1 q0 q[2..]
            pt.o0 = qm.l0; pt.l0 = qm.l1; pt.o1 = 0; pt.l1 = qm.l0; pt.o2 = qm.l0 + qm.l1; pt.l2 = qm.qlen - pt.o2;
          }
          pt.len = pt.l0 + pt.l1 + pt.l2;
          find_range(a, 0, a.n_words, 0, pt, prefix, &ra, &rb);
        }
        s_ext[tid][0] = ra;
        s_ext[tid][1] = rb;
      }
      __syncthreads();
      for (uint32_t base = 0; base < a.n_fc; base += XB) {
        const uint32_t bi = tid >> 1, shape = tid & 1, bb = base + bi;
        uint32_t ra = 0, rb = 0;
        bool flag = false;
        if (bb < a.n_fc) {
          const uint32_t fs = a.fc_start[bb], fe = a.fc_start[bb + 1];
          if (!(fs == qm.lo && qm.hi > qm.lo)) {   // not the query's own first-char block
            const uint32_t skip = utf8_len(a.flat[a.offs[fs]]);
            pt.o0 = shape ? 0 : qm.l0;             // shape 0: substitute q0 (c + q[1..]); shape 1: insert (c + q)
            pt.l0 = qm.qlen - pt.o0;
            pt.o1 = pt.l1 = pt.o2 = pt.l2 = 0;
            pt.len = pt.l0;
            find_range(a, fs, fe, skip, pt, prefix, &ra, &rb);
            flag = rb > ra;
            if (shape == 0)
              flag |= (s_ext[0][1] > s_ext[0][0] && s_ext[0][0] >= fs && s_ext[0][0] < fe) ||
                      (s_ext[1][1] > s_ext[1][0] && s_ext[1][0] >= fs && s_ext[1][0] < fe);
          }
        }
        s_xr[bi][shape * 2] = ra;
        s_xr[bi][shape * 2 + 1] = rb;
        const u64 fm = __ballot(flag);
        if (lane == 0) s_xany[wave] = fm;
        __syncthreads();
        if (tid == 0) {
          // blocks in ascending order; inside a block the (at most four) ranges may nest or overlap
          for (uint32_t w = 0; w < XW && xn < a.capx; ++w) {
            u64 mk = s_xany[w];
            while (mk && xn < a.capx) {
              const uint32_t bit = (uint32_t)__ffsll((long long)mk) - 1;
              const uint32_t k = bit >> 1;
              mk &= ~(3ull << (2 * k));
              const uint32_t b2 = w * 32 + k;
              const uint32_t fs = a.fc_start[base + b2], fe = a.fc_start[base + b2 + 1];
              uint32_t r0[4], r1[4], nr = 0;
              for (uint32_t sh = 0; sh < 2; ++sh)
                if (s_xr[b2][sh * 2 + 1] > s_xr[b2][sh * 2]) { r0[nr] = s_xr[b2][sh * 2]; r1[nr] = s_xr[b2][sh * 2 + 1]; ++nr; }
              for (uint32_t e = 0; e < 2; ++e)
                if (s_ext[e][1] > s_ext[e][0] && s_ext[e][0] >= fs && s_ext[e][0] < fe) { r0[nr] = s_ext[e][0]; r1[nr] = s_ext[e][1]; ++nr; }
              for (uint32_t i = 1; i < nr; ++i)
                for (uint32_t j = i; j > 0 && r0[j] < r0[j - 1]; --j) {
                  const uint32_t x0 = r0[j], x1 = r1[j];
                  r0[j] = r0[j - 1]; r1[j] = r1[j - 1];
                  r0[j - 1] = x0; r1[j - 1] = x1;
                }
              uint32_t cur = 0;
              for (uint32_t i = 0; i < nr; ++i) {
                for (uint32_t x = max(r0[i], cur); x < r1[i] && xn < a.capx; ++x) xl[xn++] = x;
                cur = max(cur, r1[i]);
              }
            }
          }
          if (xn >= a.capx) s_xdone = 1;
        }
        __syncthreads();
        if (s_xdone) break;
      }
    }
    if (tid == 0) a.xq_cnt[q] = xn;
  }
}

// ---- caps: the closed form of compute_derivations.rs:129-163 over a query's unit lists and its other-first-char words.
// Run by ONE thread per query: thread 0 of the workgroup that completed the query's last unit (small batches), or a thread of
// dict_caps_kernel (DictArgs::defer_caps).
__device__ __forceinline__ void dict_caps_of_query(const DictArgs &a, uint32_t q, int K, uint32_t n_sl) {
  const uint32_t stride_w = a.cap1 + a.cap2;
  const uint32_t *const xl = a.xq + (size_t)q * a.capx;
  const uint32_t xn = (K == 2 && a.n_fc > 0) ? a.xq_cnt[q] : 0u;
  const uint32_t *ul0 = a.ulists + (size_t)a.unit_off[q] * stride_w;
  const uint32_t(*uc)[2] = reinterpret_cast<const uint32_t(*)[2]>(a.ucnt + 2 * (size_t)a.unit_off[q]);
  WaveLists s1{ul0, stride_w, 0, 0, 0, 0, uc, n_sl}, s2{ul0, stride_w, a.cap1, 1, 0, 0, uc, n_sl};
  s1.settle();
  s2.settle();
  uint32_t *one = a.out_one + (size_t)q * a.cap1;
  uint32_t *two = a.out_two + (size_t)q * a.cap2;
  uint32_t n1 = 0, n2 = 0, xi = 0;
  // two = first cap2 of (X ∪ S2)
  while (n2 < a.cap2 && !(s2.done() && xi >= xn)) {
    uint32_t v;
    if (s2.done()) v = xl[xi++];
    else if (xi >= xn) { v = s2.peek(); s2.pop(); }
    else if (s2.peek() < xl[xi]) { v = s2.peek(); s2.pop(); }
    else v = xl[xi++];
    two[n2++] = v;
  }
  // one = first cap1 of (S1 ∪ {x in X : x > t*}), t* = the last element of a FULL `two`
  xi = xn;  // `two` never filled: no X word reaches `one`
  if (n2 == a.cap2 && n2 > 0) {
    const uint32_t tstar = two[n2 - 1];
    xi = 0;
    while (xi < xn && xl[xi] <= tstar) ++xi;
  }
  while (n1 < a.cap1 && !(s1.done() && xi >= xn)) {
    uint32_t v;
    if (s1.done()) v = xl[xi++];
    else if (xi >= xn) { v = s1.peek(); s1.pop(); }
    else if (s1.peek() < xl[xi]) { v = s1.peek(); s1.pop(); }
    else v = xl[xi++];
    one[n1++] = v;
  }
  a.out_one_cnt[q] = n1;
  a.out_two_cnt[q] = n2;
}

template <bool BITS>
__device__ __forceinline__ void dict_lookup_body(const DictArgs &a) {
  __shared__ u64 s_pa[128];            // PEq: ASCII chars (BITS)
  __shared__ uint32_t s_pk[128];       //      other code points: keys ...
  __shared__ u64 s_pv[128];            //      ... and their position masks
  __shared__ uint32_t s_q[QSTRIDE];
  __shared__ uint32_t s_pq[LW][PQ];
  __shared__ uint32_t s_wcnt[LW][2];
  __shared__ uint32_t s_query, s_last;
  __shared__ u64 s_t[4];               // MSI_DICT_PROFILE: thread 0's timestamps (LDS: no register lives across the phases for them)
  const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const u64 lower = (1ull << lane) - 1ull;
  const uint32_t stride_w = a.cap1 + a.cap2;
  uint32_t *const wl = a.wlists + ((size_t)blockIdx.x * LW + wave) * stride_w;
  u64 pairs = 0;
  if (a.n_long && *a.n_long == 0) return;   // (the launch for queries above 64 chars: the batch has none)

  for (;;) {
    __syncthreads();  // the previous query's LDS state is no longer read
    if (tid == 0) s_query = atomicAdd(a.ticket, 1u);
    __syncthreads();
    const uint32_t u = s_query;                 // the unit: slice `sl` of query q's first-letter range
    if (u >= a.unit_off[a.nq]) break;
    const uint32_t q = a.unit_q[u];
    const uint32_t sl = u - a.unit_off[q], n_sl = a.unit_off[q + 1] - a.unit_off[q];
    if (a.prof && tid == 0) {
      s_t[0] = wall_clock64();
      s_t[3] = 0;
    }
    const QueryMeta qm = a.qm[q];
    const int K = (int)qm.budget;
    const int m = (int)qm.m;
    if ((uint32_t)m < a.m_lo || (uint32_t)m > a.m_hi) continue;   // the other launch's query
    if (K == 0) {
      if (tid == 0) a.out_one_cnt[q] = a.out_two_cnt[q] = 0;
      continue;
    }
    const bool prefix = qm.prefix != 0;
    for (uint32_t i = tid; i < (uint32_t)m; i += LT) s_q[i] = a.qchars[(size_t)q * QSTRIDE + i];
    constexpr bool bits = BITS;   // (the bit-parallel launch is only given queries of <= 64 chars: no banded code in it)
    if (bits && tid < 128) {
      s_pa[tid] = 0ull;
      s_pk[tid] = PEQ_EMPTY;
      s_pv[tid] = 0ull;
    }
    __syncthreads();
    if (bits && tid < (uint32_t)m) {      // thread = query position
      const uint32_t c = s_q[tid];
      if (c < 128u) {
        atomicOr(&s_pa[c], 1ull << tid);
      } else {
        uint32_t h = (c * 0x9E3779B1u) >> 25;
        for (;;) {   // <= 64 distinct keys in 128 slots: a free slot always turns up
          const uint32_t k = atomicCAS(&s_pk[h], PEQ_EMPTY, c);
          if (k == PEQ_EMPTY || k == c) break;
          h = (h + 1u) & 127u;
        }
        atomicOr(&s_pv[h], 1ull << tid);
      }
    }
    if (bits) __syncthreads();
    QChars qs;
    qs.q = s_q;
    qs.m = m;
    PEq pq;
    pq.ascii = s_pa;
    pq.key = s_pk;
    pq.val = s_pv;

    // ---- scan: the same-first-char range ------------------------------------------------------------
    if (a.prof && tid == 0) s_t[1] = wall_clock64();
    uint32_t cnt0 = 0, cnt1 = 0;      // hits of this wave at distance 1 / 2 (wave-uniform)
    uint32_t head = 0, qn = 0;        // ring state (wave-uniform)
    auto drain = [&](uint32_t n) {
      const u64 t_d0 = a.prof ? wall_clock64() : 0;
      const bool act = lane < n;
      const uint32_t idx = act ? s_pq[wave][(head + lane) & (PQ - 1)] : 0u;
      const uint4 slot = a.slots[idx];
      const uint32_t wm = a.wmeta[idx];
      const int nc = (int)(wm & 0xFF);
      const uint32_t bl = wm >> 8;
      const bool is_long = act && bl > 16;
      const bool is_short = act && !is_long;
      int d = DINF;
      // the matcher: bit-parallel (32 bits for queries of <= 32 chars, else 64) or — queries above 64 chars, and the
      // round-3 kernel kept under MSI_DICT_MATCHER=banded — the five-diagonal DP
      auto pair = [&](auto r, bool on) -> int {
        if constexpr (bits) return m <= 32 ? osa_pair_bits<uint32_t>(r, nc, on, pq, m, K, prefix) : osa_pair_bits<u64>(r, nc, on, pq, m, K, prefix);
        else return osa_pair(r, nc, on, qs, K, prefix);
      };
      if (__ballot(is_short)) {
        if (__ballot(is_short && (uint32_t)nc != bl) == 0) d = pair(SlotReader<true>{slot.x, slot.y, slot.z, slot.w}, is_short);
        else d = pair(SlotReader<false>{slot.x, slot.y, slot.z, slot.w}, is_short);
        if (!is_short) d = DINF;
      }
      if (__ballot(is_long)) {   // words longer than a slot read their bytes from the flat array
        const uint32_t o0 = is_long ? a.offs[idx] : 0, o1 = is_long ? a.offs[idx + 1] : 0;
        const int dl = pair(FlatReader{a.flat + o0, a.flat + o1}, is_long);
        if (is_long) d = dl;
      }
      const uint32_t cat = !act ? 3u : (d == 1 ? 0u : ((K == 2 && d == 2) ? 1u : 3u));
      const u64 h0 = __ballot(cat == 0), h1 = __ballot(cat == 1);
      if (cat == 0) {
        const uint32_t pos = cnt0 + __popcll(h0 & lower);
        if (pos < a.cap1) wl[pos] = idx;
      }
      if (cat == 1) {
        const uint32_t pos = cnt1 + __popcll(h1 & lower);
        if (pos < a.cap2) wl[a.cap1 + pos] = idx;
      }
      cnt0 += __popcll(h0);
      cnt1 += __popcll(h1);
      pairs += n;
      head = (head + n) & (PQ - 1);
      qn -= n;
      if (a.prof && tid == 0) s_t[3] += wall_clock64() - t_d0;
    };
    if (qm.hi > qm.lo) {
      // this unit's tiles of the range
      const uint32_t r_lo = qm.lo >> 6, r_hi = (qm.hi + 63) >> 6;
      const uint32_t t_lo = min(r_hi, r_lo + sl * a.slice_tiles), t_hi = min(r_hi, t_lo + a.slice_tiles), n_t = t_hi - t_lo;
      const uint32_t t0 = t_lo + (uint32_t)((u64)n_t * wave / LW), t1 = t_lo + (uint32_t)((u64)n_t * (wave + 1) / LW);
      // The filter is a stream of 8-byte loads with a ballot behind each: one tile (64 words) per iteration left a wave
      // with a single load in flight — latency-bound at ~2 us per 64 words (r3: 0.15 of the VALU issue peak, 11 % VALU-active).
      // Four tiles' loads are issued before the first is looked at.
      constexpr uint32_t UN = 4;
      bool full = false;
      for (uint32_t tb = t0; tb < t1 && !full; tb += UN) {
        u64 f[UN];
#pragma unroll
        for (uint32_t u = 0; u < UN; ++u) {
          const uint32_t idx = (tb + u) * 64 + lane;
          f[u] = (tb + u < t1 && idx >= qm.lo && idx < qm.hi) ? a.filt[idx] : 0ull;
        }
#pragma unroll
        for (uint32_t u = 0; u < UN; ++u) {
          // both lists of this wave full: nothing it finds later can be among the first cap of the query
          if (cnt0 >= a.cap1 && (K < 2 || cnt1 >= a.cap2)) {
            qn = 0;
            full = true;
            break;
          }
          const u64 fw = f[u];
          const int nc = (int)(fw & 0xFF);
          const uint32_t q_not_w = __popcll(qm.sig & ~fw);
          const uint32_t w_not_q = prefix ? 0u : __popcll((fw & FILT_SIG) & ~qm.sig);
          const bool len_ok = (nc + K >= m) & (prefix | (nc <= m + K));
          const bool act = ((fw & FILT_VALID) != 0) & len_ok & (max(q_not_w, w_not_q) <= (uint32_t)K);
          const u64 ms = __ballot(act);
          if (ms == 0) continue;
          if (act) s_pq[wave][(head + qn + __popcll(ms & lower)) & (PQ - 1)] = (tb + u) * 64 + lane;
          qn += __popcll(ms);
          __builtin_amdgcn_wave_barrier();
          while (qn >= 64) drain(64);
        }
      }
      while (qn > 0) drain(qn < 64 ? qn : 64);
    }
    if (lane == 0) {
      s_wcnt[wave][0] = min(cnt0, a.cap1);
      s_wcnt[wave][1] = min(cnt1, a.cap2);
    }
    if (a.prof && tid == 0) s_t[2] = wall_clock64();

    // (the words with ANOTHER first char at distance <= 1 were found by dict_other_kernel: a.xq / a.xq_cnt)
    __syncthreads();  // s_wcnt and the waves' lists are complete
    // ---- this unit's hits, in dictionary order (wave order), capped: the unit list --------------------------------
    {
      uint32_t *const ul = a.ulists + (size_t)u * stride_w;
      const uint32_t *wl0 = a.wlists + (size_t)blockIdx.x * LW * stride_w;
      uint32_t off0 = 0, off1 = 0;
      for (uint32_t w = 0; w < (uint32_t)LW; ++w) {
        const uint32_t c0 = s_wcnt[w][0], c1 = s_wcnt[w][1];
        // (write-through stores at device scope, as in msi_vm.hip: the unit that completes the query — another workgroup, maybe
        // another XCD — reads them; a __threadfence here is an L2 write-back + invalidate under every resident kernel)
        for (uint32_t i = tid; i < c0; i += LT)
          if (off0 + i < a.cap1) __hip_atomic_store(&ul[off0 + i], wl0[(size_t)w * stride_w + i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        for (uint32_t i = tid; i < c1; i += LT)
          if (off1 + i < a.cap2) __hip_atomic_store(&ul[a.cap1 + off1 + i], wl0[(size_t)w * stride_w + a.cap1 + i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        off0 += c0;
        off1 += c1;
      }
      if (tid == 0) {
        __hip_atomic_store(&a.ucnt[2 * (size_t)u], min(off0, a.cap1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&a.ucnt[2 * (size_t)u + 1], min(off1, a.cap2), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
    // the unit that completes the query runs its cap logic over the units' lists: the counter's atomic follows the stores'
    // acknowledgements (s_waitcnt vmcnt(0)), the reader loads at device scope
    if (n_sl > 1) {
      MSI_ORDER_ATOMICS();
      __syncthreads();
      if (tid == 0) s_last = atomicAdd(&a.slice_done[q], 1u) == n_sl - 1 ? 1u : 0u;
      __syncthreads();
      if (!s_last) continue;
    } else {
      __syncthreads();   // (the only unit of its query: its own stores, read back by thread 0 below)
    }

    // ---- caps: the closed form of compute_derivations.rs:129-163 ------------------------------------
    if (a.defer_caps) continue;   // (dict_caps_kernel: the other-first-char words are being found beside this kernel)
    if (tid == 0) {
      const u64 t_q3 = a.prof ? wall_clock64() : 0;
      dict_caps_of_query(a, q, K, n_sl);
      if (a.prof) {   // [queries, staging + PEq, wave 0's scan (with its drains), of that its drains, search + waiting for the
                      //  slowest wave's scan, caps, whole query]
        const u64 t_q4 = wall_clock64();
        atomicAdd(&a.prof[0], 1ull);
        atomicAdd(&a.prof[1], s_t[1] - s_t[0]);
        atomicAdd(&a.prof[2], s_t[2] - s_t[1]);
        atomicAdd(&a.prof[3], s_t[3]);
        atomicAdd(&a.prof[4], t_q3 - s_t[2]);
        atomicAdd(&a.prof[5], t_q4 - t_q3);
        atomicAdd(&a.prof[6], t_q4 - s_t[0]);
      }
    }
  }
  if (lane == 0 && pairs) atomicAdd(a.pairs, pairs);
}
// What bounds this kernel is how many range-scanning waves a CU holds (each is a chain of 8-byte loads with a ballot behind
// every one: 11 % VALU-active in round 3): MI355X, C3, in-kernel timers — 92 us per query inside its workgroup at ONE
// 8-wave workgroup per CU (131 registers), a launch of 8 192 queries 3.6 ms.  The bit-parallel kernel is compiled for 6
// waves per SIMD (80 registers, three of them spilled on a cold path) = six 4-wave workgroups per CU; the banded one
// (queries above 64 chars, MSI_DICT_MATCHER=banded) keeps what it needs.
// DictArgs::defer_caps (batches of 512 words and more): the cap logic of every query, one thread each, after BOTH the lookup
// launches (the units' lists) and dict_other_kernel (the other-first-char words) — which then runs BESIDE the range scans on a
// stream of its own instead of in front of them (it was 16 % of a launch at 8 192 words: binary searches, latency-bound,
// one workgroup per budget-2 query).
__global__ __launch_bounds__(64) void dict_caps_kernel(DictArgs a) {
  const uint32_t q = blockIdx.x * 64 + threadIdx.x;
  if (q >= a.nq) return;
  const int K = (int)a.qm[q].budget;
  if (K == 0) return;   // (the lookup kernels wrote its zero counts)
  dict_caps_of_query(a, q, K, a.unit_off[q + 1] - a.unit_off[q]);
}

template <bool BITS> __global__ void dict_lookup_kernel(DictArgs a);
template <> __global__ __launch_bounds__(LT, 6) void dict_lookup_kernel<true>(DictArgs a) { dict_lookup_body<true>(a); }
template <> __global__ __launch_bounds__(LT) void dict_lookup_kernel<false>(DictArgs a) { dict_lookup_body<false>(a); }

// ---- query preparation -------------------------------------------------------------

// One thread per query: decode UTF-8, derive the first-char dictionary range.
__global__ void dict_prep_kernel(const uint8_t *__restrict__ qbytes, const uint32_t *__restrict__ qoff,
                                 const uint8_t *__restrict__ qflags, uint32_t nq,
                                 const uint4 *__restrict__ slots, uint32_t n_words,
                                 QueryMeta *__restrict__ qm, uint32_t *__restrict__ qchars, uint32_t *__restrict__ n_long) {
  const uint32_t q = blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= nq) return;
  const uint8_t *s = qbytes + qoff[q];
  const uint32_t len = qoff[q + 1] - qoff[q];
  QueryMeta r;
  r.m = 0;
  r.budget = 0;
  r.prefix = (qflags[q] >> 2) & 1;
  r.qlen = len;
  r._pad[0] = r._pad[1] = 0;
  r.lo = r.hi = 0;
  r.sig = 0;
  r.l0 = r.l1 = 0;
  r.q0 = r.q1 = QNONE;
  const uint32_t bud = qflags[q] & 3;
  if (len >= 1 && len <= 250 && bud >= 1) {  // MAX_WORD_LENGTH, compute_derivations.rs:180-192
    uint32_t *out = qchars + (size_t)q * QSTRIDE;
    uint32_t n = 0, i = 0;
    while (i < len) {
      const uint32_t b0 = s[i];
      uint32_t cl = utf8_len(b0);
      uint32_t cp = cl == 1 ? b0 : (cl == 2 ? (b0 & 0x1F) : (cl == 3 ? (b0 & 0x0F) : (b0 & 0x07)));
      for (uint32_t e = 1; e < cl; ++e) cp = (cp << 6) | ((i + e < len ? s[i + e] : 0) & 0x3F);
      if (i + cl > len) cl = len - i;   // a truncated last char keeps the bytes it has
      if (n == 0) r.l0 = cl;
      if (n == 1) r.l1 = cl;
      i += cl;
      out[n++] = cp;
    }
    r.m = n;
    r.budget = bud > 2 ? 2 : bud;
    for (uint32_t c = 0; c < n; ++c) r.sig |= 1ull << sig_bit(out[c]);
    r.q0 = out[0];
    r.q1 = n >= 2 ? out[1] : QNONE;
    // words starting with a given char: compare the first clen bytes (big endian)
    auto range_of = [&](const uint8_t *c, uint32_t avail, uint32_t &rlo, uint32_t &rhi) {
      const uint32_t clen = utf8_len(c[0]);
      uint32_t key = 0;
      for (uint32_t e = 0; e < clen; ++e) key = (key << 8) | (e < avail ? c[e] : 0);
      auto head = [&](uint32_t idx) -> uint32_t {
        const uint32_t w = slots[idx].x;  // little-endian bytes 0..3
        uint32_t v = 0;
        for (uint32_t e = 0; e < clen; ++e) v = (v << 8) | ((w >> (8 * e)) & 0xFF);
        return v;
      };
      uint32_t lo = 0, hi = n_words;
      while (lo < hi) {
        const uint32_t mid = lo + (hi - lo) / 2;
        if (head(mid) < key) lo = mid + 1;
        else hi = mid;
      }
      rlo = lo;
      hi = n_words;
      uint32_t l2 = lo;
      while (l2 < hi) {
        const uint32_t mid = l2 + (hi - l2) / 2;
        if (head(mid) <= key) l2 = mid + 1;
        else hi = mid;
      }
      rhi = l2;
    };
    range_of(s, len, r.lo, r.hi);
    if (n > 64) atomicAdd(n_long, 1u);
  }
  qm[q] = r;
}

}  // namespace

// ================================================================== host object

struct msi_dict {
  msi_ctx *ctx = nullptr;
  uint32_t n_words = 0;
  uint32_t n_fc = 0;   // first-char blocks
  DevBuf slots, filt, wmeta, flat, offs, fc_start;
  // scratch (guarded by ctx->mu_aux)
  DevBuf qbytes, qoff, qflags, qm, qchars, wlists, xq, xq_cnt, ticket, pairs, out1, out1c, out2, out2c, prof;
  DevBuf unit_off, unit_q, slice_done, ulists, ucnt;
  uint32_t max_block_tiles = 1;   // 64-word tiles of the largest first-letter block (what a query's range can be at most)
  // host entry point (msi_dict_lookup): pinned staging of the packed queries and results, and an event the caller
  // SLEEPS on — with pageable buffers every copy was a staged, spinning call and hipStreamSynchronize a busy-wait: a fifth
  // of the host CPU of the keyword leg at 64 callers (profiles/r3_ranked_cpu_profile_before.txt)
  uint8_t *h_stage = nullptr;
  size_t h_stage_cap = 0;
  hipEvent_t h_done = nullptr;
  // dict_other_kernel's own stream and the two events that fork it off the lookup's stream and join it again (defer_caps)
  hipStream_t other_stream = nullptr;
  hipEvent_t ev_fork = nullptr, ev_join = nullptr;
  uint64_t lookup_launches = 0, dict_bytes = 0;
  KernelTimer match_timer;
  // host copy of the sorted words (prefix ranges, idx -> word for the keyword pipeline)
  std::vector<uint8_t> h_flat;
  std::vector<uint32_t> h_offs;
  // micro-batcher: concurrent msi_dict_lookup callers with the same caps share one launch
  struct Pending {
    const msi_typo_query *queries;
    uint32_t n, cap_one, cap_two;
    uint32_t *one_idx, *one_cnt, *two_idx, *two_cnt;
    int32_t status = MSI_OK;
    std::string error;
    std::atomic<uint32_t> done{0};   // set by the leader LAST: the waiter may return (and its Pending vanish) right after
  };
  std::mutex bmu;
  // Waiters sleep on futex words, not on a condition variable: with 128 searches in flight every arrival used to
  // notify_all() the others so that the leader could re-count the queue, and every woken thread went through the mutex —
  // a quarter of the keyword leg's host CPU was futex traffic (profiles/r3_ranked_arena_profile.txt).
  std::atomic<uint32_t> bgen{0};    // bumped (under bmu) whenever a leader is done: its batch returns, one of the rest leads
  std::atomic<uint32_t> bfill{0};   // bumped (under bmu) by the arrival that fills the queue to the target: the leader's wake-up
  std::vector<Pending *> bqueue;
  bool bleader_active = false;
  uint32_t microbatch_wait_us = 0, microbatch_target = 256;
  uint64_t fused_calls = 0, fused_launches = 0;
  bool values_mode = false;  // msi_dict_create_values: every word carries the sentinel first byte
  MsiPostingCache *pcache = nullptr;   // HBM cache of the index version's stored postings (msi_dict_enable_posting_cache)
};

// accessors for msi_keyword.hip / msi_search.hip
MsiPostingCache *msi_dict_pcache(const msi_dict *d) { return d ? d->pcache : nullptr; }
bool msi_dict_word(const msi_dict *d, uint32_t idx, const uint8_t **w, uint32_t *len) {
  if (!d || idx >= d->n_words) return false;
  *w = d->h_flat.data() + d->h_offs[idx];
  *len = d->h_offs[idx + 1] - d->h_offs[idx];
  return true;
}
// [lo, hi) = dictionary words that start with `prefix` (byte-lexicographic order).
void msi_dict_prefix_range(const msi_dict *d, const uint8_t *prefix, uint32_t plen, uint32_t *lo, uint32_t *hi) {
  auto cmp = [&](uint32_t idx) {  // <0: word < prefix-range, 0: has prefix, >0: beyond
    const uint8_t *w = d->h_flat.data() + d->h_offs[idx];
    const uint32_t wl = d->h_offs[idx + 1] - d->h_offs[idx];
    const int c = memcmp(w, prefix, std::min(wl, plen));
    if (c != 0) return c;
    return wl < plen ? -1 : 0;
  };
  uint32_t a = 0, b = d->n_words;
  while (a < b) {
    const uint32_t m = a + (b - a) / 2;
    if (cmp(m) < 0) a = m + 1; else b = m;
  }
  *lo = a;
  b = d->n_words;
  while (a < b) {
    const uint32_t m = a + (b - a) / 2;
    if (cmp(m) <= 0) a = m + 1; else b = m;
  }
  *hi = a;
}

namespace {

int32_t enqueue_lookup(msi_dict *d, const uint8_t *d_qbytes, const uint32_t *d_qoff, const uint8_t *d_qflags,
                       uint32_t n, uint32_t cap1, uint32_t cap2, uint32_t *d_one, uint32_t *d_one_cnt,
                       uint32_t *d_two, uint32_t *d_two_cnt) {
  msi_ctx *ctx = d->ctx;
  hipStream_t st = ctx->stream_aux;
  if (cap1 == 0 || cap2 == 0 || cap1 > 4096 || cap2 > 4096) {
    msi_set_error("msi_dict_lookup: caps must be in 1..4096");
    return MSI_E_INVALID;
  }
  const uint32_t capx = cap1 + cap2;
  MSI_TRY(d->qm.ensure((size_t)n * sizeof(QueryMeta)));
  MSI_TRY(d->qchars.ensure((size_t)n * QSTRIDE * sizeof(uint32_t)));
  if (d->n_words == 0) {   // nothing can match: no kernel reads an empty dictionary
    MSI_HIP_TRY(hipMemsetAsync(d_one_cnt, 0, (size_t)n * sizeof(uint32_t), st));
    MSI_HIP_TRY(hipMemsetAsync(d_two_cnt, 0, (size_t)n * sizeof(uint32_t), st));
    d->lookup_launches++;
    return MSI_OK;
  }
  // ticket[0], ticket[1]: the two lookup launches' query counters; ticket[2]: queries above 64 chars (dict_prep_kernel);
  // ticket[3]: the other-first-letter kernel's counter
  MSI_TRY(d->ticket.ensure(4 * sizeof(uint32_t)));
  MSI_HIP_TRY(hipMemsetAsync(d->ticket.p, 0, 4 * sizeof(uint32_t), st));
  hipLaunchKernelGGL(dict_prep_kernel, dim3((n + 127) / 128), dim3(128), 0, st, d_qbytes, d_qoff, d_qflags, n,
                     d->slots.as<uint4>(), d->n_words, d->qm.as<QueryMeta>(), d->qchars.as<uint32_t>(), d->ticket.as<uint32_t>() + 2);
  // one workgroup per query in flight (persistent, queries by ticket): as many as the kernel's registers let a CU hold
  // (what the kernels are compiled for: 4 / 6 waves per SIMD = workgroups of 4 waves per CU; tests/test_isa_cpu.py holds the
  // register counts behind these; MSI_DICT_WG_PER_CU overrides for experiments)
  static int wg_per_cu[2] = {0, 0};   // [banded, bit-parallel]
  if (!wg_per_cu[0]) {
    const int knob = getenv("MSI_DICT_WG_PER_CU") ? atoi(getenv("MSI_DICT_WG_PER_CU")) : 0;
    wg_per_cu[1] = knob > 0 ? knob : 6;
    wg_per_cu[0] = knob > 0 ? knob : 4;
  }
  const uint32_t grid_cap = (uint32_t)ctx->n_cu * (uint32_t)std::max(wg_per_cu[0], wg_per_cu[1]);
  // slices of a query's first-letter range (DictArgs::unit_off): 256 tiles = 16 384 words each — wider when the batch is so
  // large that its units' lists would not fit a 128 Mi-entry budget (a range has at most max_block_tiles tiles)
  uint32_t slice_tiles = 256;
  if (const char *e = getenv("MSI_DICT_SLICE_TILES")) slice_tiles = (uint32_t)std::max(1, atoi(e));   // experiments
  auto units_bound = [&](uint32_t st_) { return (uint64_t)n * ((d->max_block_tiles + st_ - 1) / st_); };
  while (units_bound(slice_tiles) * (cap1 + cap2) > (128ull << 20) && slice_tiles < (1u << 24)) slice_tiles *= 2;
  const uint32_t max_units = (uint32_t)std::min<uint64_t>(units_bound(slice_tiles), 0x7FFFFFFFull);
  const uint32_t grid = std::min<uint32_t>(max_units, grid_cap);
  MSI_TRY(d->wlists.ensure((size_t)grid * LW * (cap1 + cap2) * sizeof(uint32_t)));
  MSI_TRY(d->unit_off.ensure(((size_t)n + 1) * sizeof(uint32_t)));
  MSI_TRY(d->unit_q.ensure((size_t)max_units * sizeof(uint32_t)));
  MSI_TRY(d->slice_done.ensure((size_t)n * sizeof(uint32_t)));
  MSI_TRY(d->ulists.ensure((size_t)max_units * (cap1 + cap2) * sizeof(uint32_t)));
  MSI_TRY(d->ucnt.ensure((size_t)max_units * 2 * sizeof(uint32_t)));
  hipLaunchKernelGGL(dict_plan_kernel, dim3(1), dim3(1024), 0, st, d->qm.as<QueryMeta>(), n, slice_tiles, d->unit_off.as<uint32_t>(),
                     d->unit_q.as<uint32_t>(), d->slice_done.as<uint32_t>(), max_units);
  MSI_TRY(d->xq.ensure((size_t)n * capx * sizeof(uint32_t)));
  MSI_TRY(d->xq_cnt.ensure((size_t)n * sizeof(uint32_t)));
  DictArgs a;
  a.slots = d->slots.as<uint4>();
  a.filt = d->filt.as<u64>();
  a.wmeta = d->wmeta.as<uint16_t>();
  a.flat = d->flat.as<uint8_t>();
  a.offs = d->offs.as<uint32_t>();
  a.fc_start = d->fc_start.as<uint32_t>();
  a.n_fc = d->n_fc;
  a.n_words = d->n_words;
  a.qm = d->qm.as<QueryMeta>();
  a.qchars = d->qchars.as<uint32_t>();
  a.qbytes = d_qbytes;
  a.qoff = d_qoff;
  a.nq = n;
  a.cap1 = cap1;
  a.cap2 = cap2;
  a.capx = capx;
  a.wlists = d->wlists.as<uint32_t>();
  a.xq = d->xq.as<uint32_t>();
  a.xq_cnt = d->xq_cnt.as<uint32_t>();
  a.unit_off = d->unit_off.as<uint32_t>();
  a.unit_q = d->unit_q.as<uint32_t>();
  a.slice_done = d->slice_done.as<uint32_t>();
  a.ulists = d->ulists.as<uint32_t>();
  a.ucnt = d->ucnt.as<uint32_t>();
  a.slice_tiles = slice_tiles;
  a.ticket = d->ticket.as<uint32_t>();
  a.pairs = d->pairs.as<u64>();
  a.prof = nullptr;
  if (getenv("MSI_DICT_PROFILE")) {   // diagnostics: where a query's time goes inside the kernel (printed by msi_dict_destroy)
    if (!d->prof.p && d->prof.ensure(8 * sizeof(u64)) == MSI_OK) (void)hipMemsetAsync(d->prof.p, 0, 8 * sizeof(u64), st);
    a.prof = d->prof.as<u64>();
  }
  a.out_one = d_one;
  a.out_one_cnt = d_one_cnt;
  a.out_two = d_two;
  a.out_two_cnt = d_two_cnt;
  // batches (C3, the C4 step's 1 536 words): dict_other_kernel beside the range scans, the caps behind both; a search's own
  // one-to-three-word lookups keep the cap logic inside the lookup kernel (one launch and two event waits less)
  static const uint32_t defer_min = getenv("MSI_DICT_DEFER_CAPS_MIN") ? (uint32_t)atoi(getenv("MSI_DICT_DEFER_CAPS_MIN")) : 512u;
  bool defer = defer_min > 0 && n >= defer_min;
  if (defer && !d->other_stream) {
    if (hipStreamCreateWithFlags(&d->other_stream, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreateWithFlags(&d->ev_fork, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&d->ev_join, hipEventDisableTiming) != hipSuccess) {
      (void)hipGetLastError();
      defer = false;   // (no second stream to be had: the serial order of rounds 4-5)
    }
  }
  a.defer_caps = defer ? 1u : 0u;
  d->match_timer.begin(ctx, st);
  // MSI_DICT_MATCHER=banded: the round-3 five-diagonal DP for every survivor (kept for A/B runs and as the matcher of
  // queries above 64 chars); default: the bit-parallel recurrence
  static const bool banded = getenv("MSI_DICT_MATCHER") && !strcmp(getenv("MSI_DICT_MATCHER"), "banded");
  a.m_lo = 0;
  a.m_hi = 0xFFFFFFFFu;
  a.n_long = nullptr;
  {   // the other-first-letter words of every budget-2 query first: the matcher kernel's cap logic reads them
    DictArgs x = a;
    x.ticket = d->ticket.as<uint32_t>() + 3;
    if (defer) {
      MSI_HIP_TRY(hipEventRecord(d->ev_fork, st));
      MSI_HIP_TRY(hipStreamWaitEvent(d->other_stream, d->ev_fork, 0));
    }
    hipLaunchKernelGGL(dict_other_kernel, dim3(std::min<uint32_t>(n, (uint32_t)ctx->n_cu * 2)), dim3(XT), 0, defer ? d->other_stream : st, x);
    if (defer) MSI_HIP_TRY(hipEventRecord(d->ev_join, d->other_stream));
  }
  if (banded) {
    hipLaunchKernelGGL(dict_lookup_kernel<false>, dim3(std::min<uint32_t>(max_units, (uint32_t)ctx->n_cu * (uint32_t)wg_per_cu[0])), dim3(LT), 0, st, a);
  } else {
    // two launches: the bit-parallel kernel (no banded code in it: fewer registers, more workgroups per CU) for the queries
    // of up to 64 chars, the banded kernel for the rest — its workgroups leave at once when the batch has none
    a.m_hi = 64;
    hipLaunchKernelGGL(dict_lookup_kernel<true>, dim3(std::min<uint32_t>(max_units, (uint32_t)ctx->n_cu * (uint32_t)wg_per_cu[1])), dim3(LT), 0, st, a);
    a.m_lo = 65;
    a.m_hi = 0xFFFFFFFFu;
    a.ticket = d->ticket.as<uint32_t>() + 1;
    a.n_long = d->ticket.as<uint32_t>() + 2;
    hipLaunchKernelGGL(dict_lookup_kernel<false>, dim3(std::min<uint32_t>(max_units, (uint32_t)ctx->n_cu * (uint32_t)wg_per_cu[0])), dim3(LT), 0, st, a);
  }
  if (defer) {
    MSI_HIP_TRY(hipStreamWaitEvent(st, d->ev_join, 0));
    hipLaunchKernelGGL(dict_caps_kernel, dim3((n + 63) / 64), dim3(64), 0, st, a);
  }
  d->match_timer.end(ctx);
  MSI_HIP_TRY(hipGetLastError());
  d->lookup_launches++;
  return MSI_OK;
}

}  // namespace

extern "C" {

int32_t msi_dict_create(msi_ctx *ctx, const uint8_t *words_concat, const uint32_t *offsets, uint32_t n_words,
                        msi_dict **out) {
  if (!ctx || !out || (n_words && (!words_concat || !offsets))) {
    msi_set_error("msi_dict_create: invalid argument");
    return MSI_E_INVALID;
  }
  *out = nullptr;
  // host-side staging: slots, lengths, char-presence signatures, first-char blocks; sortedness check
  std::vector<uint4> slots(std::max<uint32_t>(1, n_words));
  std::vector<u64> filt(std::max<uint32_t>(1, n_words));
  std::vector<uint16_t> wmeta(std::max<uint32_t>(1, n_words));
  std::vector<uint32_t> fc_start;
  uint32_t prev_fc = 0, prev_fcl = 0;   // first char of the previous non-empty word (its bytes, big endian)
  for (uint32_t i = 0; i < n_words; ++i) {
    const uint32_t o = offsets[i], len = offsets[i + 1] - o;
    if (offsets[i + 1] < o || len > 255) {
      msi_set_error("msi_dict_create: word %u has invalid length %u (max 255 bytes)", i, len);
      return MSI_E_INVALID;
    }
    const uint8_t *w = words_concat + o;
    if (i > 0) {
      const uint32_t po = offsets[i - 1], pl = o - po;
      const int c = memcmp(words_concat + po, w, std::min(pl, len));
      if (c > 0 || (c == 0 && pl >= len)) {
        msi_set_error("msi_dict_create: words must be byte-lexicographically sorted and unique (at %u)", i);
        return MSI_E_NOT_SORTED;
      }
    }
    uint8_t buf[16] = {0};
    memcpy(buf, w, std::min<uint32_t>(len, 16));
    memcpy(&slots[i], buf, 16);
    uint32_t chars = 0;
    u64 sig = 0;
    for (uint32_t b = 0; b < len;) {
      const uint32_t b0 = w[b];
      const uint32_t cl = b0 < 0x80 ? 1u : (b0 < 0xE0 ? 2u : (b0 < 0xF0 ? 3u : 4u));
      uint32_t cp = cl == 1 ? b0 : (cl == 2 ? (b0 & 0x1F) : (cl == 3 ? (b0 & 0x0F) : (b0 & 0x07)));
      for (uint32_t e = 1; e < cl; ++e) cp = (cp << 6) | ((b + e < len ? w[b + e] : 0) & 0x3F);   // as the kernels decode
      sig |= 1ull << sig_bit(cp);
      ++chars;
      b += cl;
    }
    filt[i] = sig | (len ? FILT_VALID : 0ull) | std::min<uint32_t>(chars, 255);
    wmeta[i] = (uint16_t)(std::min<uint32_t>(chars, 255) | (len << 8));
    if (len) {
      const uint32_t b0 = w[0];
      const uint32_t cl = std::min<uint32_t>(len, b0 < 0x80 ? 1u : (b0 < 0xE0 ? 2u : (b0 < 0xF0 ? 3u : 4u)));
      uint32_t fc = 0;
      for (uint32_t e = 0; e < cl; ++e) fc = (fc << 8) | w[e];
      if (fc_start.empty() || fc != prev_fc || cl != prev_fcl) fc_start.push_back(i);
      prev_fc = fc;
      prev_fcl = cl;
    }
  }
  const uint32_t n_fc = (uint32_t)fc_start.size();
  fc_start.push_back(n_words);
  uint32_t max_block_words = 1;
  for (uint32_t b = 0; b < n_fc; ++b) max_block_words = std::max(max_block_words, fc_start[b + 1] - fc_start[b]);
  DeviceGuard g(ctx->device);
  std::lock_guard<std::mutex> lk(ctx->mu_aux);
  msi_dict *d = new msi_dict();
  d->ctx = ctx;
  d->n_words = n_words;
  d->n_fc = n_fc;
  d->max_block_tiles = max_block_words / 64 + 2;
  const size_t flat_bytes = n_words ? offsets[n_words] : 0;
  hipStream_t st = ctx->stream_aux;
  int32_t s = MSI_OK;
  auto up = [&](DevBuf &b, const void *src, size_t bytes) {
    if (s != MSI_OK) return;
    s = b.ensure(std::max<size_t>(16, bytes));
    if (s == MSI_OK && bytes) {
      hipError_t e = hipMemcpyAsync(b.p, src, bytes, hipMemcpyHostToDevice, st);
      if (e != hipSuccess) {
        msi_set_error("hipMemcpyAsync failed: %s", hipGetErrorString(e));
        s = MSI_E_HIP;
      }
    }
  };
  up(d->slots, slots.data(), (size_t)n_words * sizeof(uint4));
  up(d->filt, filt.data(), (size_t)n_words * sizeof(u64));
  up(d->wmeta, wmeta.data(), (size_t)n_words * sizeof(uint16_t));
  up(d->flat, words_concat, flat_bytes);
  up(d->offs, offsets, ((size_t)n_words + 1) * sizeof(uint32_t));
  up(d->fc_start, fc_start.data(), fc_start.size() * sizeof(uint32_t));
  if (s == MSI_OK) s = d->pairs.ensure(sizeof(u64));
  if (s == MSI_OK && hipMemsetAsync(d->pairs.p, 0, sizeof(u64), st) != hipSuccess) s = MSI_E_HIP;
  if (s == MSI_OK && hipStreamSynchronize(st) != hipSuccess) {
    msi_set_error("msi_dict_create: stream synchronize failed");
    s = MSI_E_HIP;
  }
  if (s != MSI_OK) {
    DevBuf *bufs[] = {&d->slots, &d->filt, &d->wmeta, &d->flat, &d->offs, &d->fc_start, &d->pairs};
    for (DevBuf *b : bufs) b->release();
    delete d;
    return s;
  }
  d->dict_bytes = (uint64_t)n_words * (sizeof(uint4) + sizeof(u64) + 2) + flat_bytes + ((uint64_t)n_words + 1) * 4;
  d->h_flat.assign(words_concat, words_concat + flat_bytes);
  d->h_offs.assign(offsets, offsets + n_words + 1);
  if (n_words == 0) d->h_offs.assign(1, 0);
  msi_ctx_retain(ctx);
  *out = d;
  return MSI_OK;
}

void msi_dict_destroy(msi_dict *d) {
  if (!d) return;
  msi_ctx *ctx = d->ctx;
  if (d->pcache) msi_pcache_destroy(d->pcache);
  d->pcache = nullptr;
  {
  std::lock_guard<std::mutex> lk(ctx->mu_aux);
  DeviceGuard g(ctx->device);
  (void)hipStreamSynchronize(d->ctx->stream_aux);
  if (d->prof.p) {
    u64 t[8] = {0};
    (void)hipMemcpy(t, d->prof.p, sizeof t, hipMemcpyDeviceToHost);
    if (t[0])
      fprintf(stderr, "msi_dict profile: %llu queries, %.1f us each in their workgroup: staging %.1f, wave 0's range scan %.1f (of it matcher drains %.1f), "
              "other first letters + waiting for the slowest wave %.1f, caps %.1f\n", (unsigned long long)t[0], t[6] / 100.0 / t[0], t[1] / 100.0 / t[0],
              t[2] / 100.0 / t[0], t[3] / 100.0 / t[0], t[4] / 100.0 / t[0], t[5] / 100.0 / t[0]);
    d->prof.release();
  }
  DevBuf *bufs[] = {&d->slots, &d->filt, &d->wmeta, &d->flat, &d->offs, &d->fc_start, &d->qbytes, &d->qoff,
                    &d->qflags, &d->qm, &d->qchars, &d->wlists, &d->xq, &d->xq_cnt, &d->unit_off, &d->unit_q, &d->slice_done, &d->ulists, &d->ucnt, &d->ticket, &d->pairs,
                    &d->out1, &d->out1c, &d->out2, &d->out2c};
  for (DevBuf *b : bufs) b->release();
  d->match_timer.release();
  if (d->h_stage) (void)hipHostFree(d->h_stage);
  if (d->h_done) (void)hipEventDestroy(d->h_done);
  if (d->other_stream) {
    (void)hipStreamSynchronize(d->other_stream);
    (void)hipStreamDestroy(d->other_stream);
  }
  if (d->ev_fork) (void)hipEventDestroy(d->ev_fork);
  if (d->ev_join) (void)hipEventDestroy(d->ev_join);
  delete d;
  }
  msi_ctx_release(ctx);
}

uint32_t msi_dict_len(const msi_dict *d) { return d ? d->n_words : 0; }

int32_t msi_dict_enable_posting_cache(msi_dict *d, uint64_t capacity_bytes) {
  if (!d) return MSI_E_INVALID;
  std::lock_guard<std::mutex> lk(d->bmu);
  if (d->pcache) return MSI_OK;
  d->pcache = msi_pcache_create(d->ctx, capacity_bytes);
  return d->pcache ? MSI_OK : MSI_E_OOM;
}

// Forgets everything the posting cache holds (a fresh cache of the same capacity): what a server does when it wants the
// memory back, and how the bench measures a cold-cache pass.  Only while no search on this dictionary is in flight.
int32_t msi_dict_reset_posting_cache(msi_dict *d) {
  if (!d) return MSI_E_INVALID;
  std::lock_guard<std::mutex> lk(d->bmu);
  if (!d->pcache) return MSI_OK;
  msi_pcache_reset(d->pcache);   // (what msi_dict_stage_postings put there stays: it is the index, not a search's leftovers)
  return MSI_OK;
}

// Index-open staging of stored postings (north_star: "staged once into HBM"; the reference reads every posting from the
// mmap on demand, db_cache.rs:50-84 — one lookup, no copy: with the databases staged a search's lookup is one probe of the
// cache's host table and its decode reads HBM).
int32_t msi_dict_stage_postings(msi_dict *d, uint64_t index_view, const msi_staged_posting *values, uint64_t n_values,
                                uint64_t out_counts[3]) {
  if (out_counts) out_counts[0] = out_counts[1] = out_counts[2] = 0;
  if (!d || (!values && n_values)) {
    msi_set_error("msi_dict_stage_postings: invalid argument");
    return MSI_E_INVALID;
  }
  if (!d->pcache) {
    msi_set_error("msi_dict_stage_postings: the dictionary has no posting cache (msi_dict_enable_posting_cache first)");
    return MSI_E_INVALID;
  }
  std::vector<MsiStageValue> v(n_values);
  for (uint64_t i = 0; i < n_values; ++i) {
    const msi_staged_posting &s = values[i];
    if (s.db == 0 || s.db >= 32 || (s.key1_len && !s.key1) || (s.key2_len && !s.key2) || (s.n && !s.bytes)) {
      msi_set_error("msi_dict_stage_postings: value %llu is malformed", (unsigned long long)i);
      return MSI_E_INVALID;
    }
    v[i].key = msi_cache_key(s.db, s.key1, s.key1_len, s.key2, s.key2_len, s.x, s.y, index_view);
    v[i].bytes = s.n ? s.bytes : nullptr;
    v[i].len = s.n;
  }
  return msi_pcache_stage(d->pcache, v.data(), n_values, out_counts);
}

int32_t msi_dict_stage_complete(msi_dict *d, uint64_t index_view, uint32_t db_mask) {
  if (!d || !d->pcache || (db_mask & 1u)) {
    msi_set_error("msi_dict_stage_complete: invalid argument (no posting cache, or database 0)");
    return MSI_E_INVALID;
  }
  msi_pcache_set_complete(d->pcache, index_view, db_mask);
  return MSI_OK;
}

int32_t msi_dict_staged_stats(msi_dict *d, uint64_t out[4]) {
  if (!d || !out) return MSI_E_INVALID;
  out[0] = out[1] = out[2] = out[3] = 0;
  if (d->pcache) msi_pcache_staged_stats(d->pcache, out);
  return MSI_OK;
}

int32_t msi_dict_posting_cache_stats(msi_dict *d, uint64_t out[4]) {
  if (!d || !out) return MSI_E_INVALID;
  out[0] = out[1] = out[2] = out[3] = 0;
  if (d->pcache) msi_pcache_stats(d->pcache, out);
  return MSI_OK;
}

int32_t msi_dict_lookup_device(msi_dict *d, const uint8_t *d_qbytes, const uint32_t *d_qoff, const uint8_t *d_qflags,
                               uint32_t n, uint32_t cap_one, uint32_t cap_two, uint32_t *d_out_one_idx,
                               uint32_t *d_out_one_cnt, uint32_t *d_out_two_idx, uint32_t *d_out_two_cnt) {
  if (!d || !n || !d_qbytes || !d_qoff || !d_qflags || !d_out_one_idx || !d_out_one_cnt || !d_out_two_idx ||
      !d_out_two_cnt) {
    msi_set_error("msi_dict_lookup_device: invalid argument");
    return MSI_E_INVALID;
  }
  std::lock_guard<std::mutex> lk(d->ctx->mu_aux);
  DeviceGuard g(d->ctx->device);
  return enqueue_lookup(d, d_qbytes, d_qoff, d_qflags, n, cap_one, cap_two, d_out_one_idx, d_out_one_cnt,
                        d_out_two_idx, d_out_two_cnt);
}

}  // extern "C"

static int32_t dict_lookup_direct(msi_dict *d, const msi_typo_query *queries, uint32_t n, uint32_t cap_one,
                                  uint32_t cap_two, uint32_t *out_one_idx, uint32_t *out_one_cnt,
                                  uint32_t *out_two_idx, uint32_t *out_two_cnt);

static void dict_futex_wake(std::atomic<uint32_t> *w, int n) {
  syscall(SYS_futex, reinterpret_cast<uint32_t *>(w), FUTEX_WAKE_PRIVATE, n, nullptr, nullptr, 0);
}
static void dict_futex_wait(std::atomic<uint32_t> *w, uint32_t expected, long timeout_us) {
  struct timespec ts;
  ts.tv_sec = timeout_us / 1000000;
  ts.tv_nsec = (timeout_us % 1000000) * 1000;
  syscall(SYS_futex, reinterpret_cast<uint32_t *>(w), FUTEX_WAIT_PRIVATE, expected, &ts, nullptr, 0);
}

// Micro-batcher: a search derives the typos of its <= 10 words (+ n-grams) with one small
// lookup; under load many searches do so at once (4 x cores spawn_blocking threads).  The
// first caller leads: it waits until `microbatch_target` words are queued or
// `microbatch_wait_us` elapsed, runs ONE launch for every queued request with its caps and
// hands the rows out.  Requests with other caps wait for the next leader.
static int32_t dict_lookup_fused(msi_dict *d, const msi_typo_query *queries, uint32_t n, uint32_t cap_one,
                                 uint32_t cap_two, uint32_t *one_idx, uint32_t *one_cnt, uint32_t *two_idx,
                                 uint32_t *two_cnt) {
  msi_dict::Pending me;
  me.queries = queries;
  me.n = n;
  me.cap_one = cap_one;
  me.cap_two = cap_two;
  me.one_idx = one_idx;
  me.one_cnt = one_cnt;
  me.two_idx = two_idx;
  me.two_cnt = two_cnt;
  std::unique_lock<std::mutex> lk(d->bmu);
  d->bqueue.push_back(&me);
  auto queued_words = [&] {
    uint32_t w = 0;
    for (auto *p : d->bqueue) w += p->n;
    return w;
  };
  auto finished = [&]() -> int32_t {
    if (me.status != MSI_OK) msi_set_error("%s", me.error.c_str());
    return me.status;
  };
  for (;;) {
    if (d->bleader_active) {
      const bool filled = queued_words() >= d->microbatch_target;
      if (filled) d->bfill.fetch_add(1, std::memory_order_release);
      const uint32_t gen = d->bgen.load(std::memory_order_acquire);
      lk.unlock();
      if (filled) dict_futex_wake(&d->bfill, 1);
      while (!me.done.load(std::memory_order_acquire) && d->bgen.load(std::memory_order_acquire) == gen)
        dict_futex_wait(&d->bgen, gen, 2000);
      if (me.done.load(std::memory_order_acquire)) return finished();
      lk.lock();                       // the leader is done and my request was not in its batch (other caps, or I came late)
      if (me.done.load(std::memory_order_acquire)) return finished();
      continue;
    }
    d->bleader_active = true;
    if (queued_words() < d->microbatch_target) {
      const uint32_t fill = d->bfill.load(std::memory_order_acquire);
      lk.unlock();
      dict_futex_wait(&d->bfill, fill, d->microbatch_wait_us);
      lk.lock();
    }
    // take every request with MY caps (always includes me); leave the others queued
    std::vector<msi_dict::Pending *> batch, rest;
    for (auto *p : d->bqueue) (p->cap_one == cap_one && p->cap_two == cap_two ? batch : rest).push_back(p);
    d->bqueue.swap(rest);
    lk.unlock();
    uint32_t total = 0;
    for (auto *p : batch) total += p->n;
    std::vector<msi_typo_query> q(total);
    std::vector<uint32_t> o1((size_t)total * cap_one), o2((size_t)total * cap_two), c1(total), c2(total);
    size_t off = 0;
    for (auto *p : batch) {
      memcpy(q.data() + off, p->queries, (size_t)p->n * sizeof(msi_typo_query));
      off += p->n;
    }
    const int32_t st = dict_lookup_direct(d, q.data(), total, cap_one, cap_two, o1.data(), c1.data(), o2.data(), c2.data());
    const std::string err = st == MSI_OK ? std::string() : std::string(msi_last_error());
    off = 0;
    for (auto *p : batch) {
      if (st == MSI_OK) {
        memcpy(p->one_idx, o1.data() + off * cap_one, (size_t)p->n * cap_one * sizeof(uint32_t));
        memcpy(p->two_idx, o2.data() + off * cap_two, (size_t)p->n * cap_two * sizeof(uint32_t));
        memcpy(p->one_cnt, c1.data() + off, (size_t)p->n * sizeof(uint32_t));
        memcpy(p->two_cnt, c2.data() + off, (size_t)p->n * sizeof(uint32_t));
      }
      off += p->n;
    }
    lk.lock();
    d->fused_calls += batch.size();
    d->fused_launches += 1;
    for (auto *p : batch) {
      if (p == &me) continue;
      p->status = st;
      p->error = err;
      p->done.store(1, std::memory_order_release);   // (p may be gone from here on)
    }
    d->bleader_active = false;
    d->bgen.fetch_add(1, std::memory_order_release);
    lk.unlock();
    dict_futex_wake(&d->bgen, INT32_MAX);
    if (st != MSI_OK) msi_set_error("%s", err.c_str());
    return st;   // my own request was in the batch
  }
}

extern "C" {

// ---- facet search (search/facet/search.rs:122-190) ----------------------------------------------
// `fst.search(build_dfa(query, typos, is_prefix = true))` over a facet's values: every value with a prefix within
// `typos` edits of the query — no first-letter rule, no per-class caps, distance 0 included.  The derivation
// kernel applies the first-letter rule by dictionary ranges and classes; staging every value behind one common
// sentinel byte makes all first letters equal, so the same kernel answers this question unchanged (the
// sentinel costs no edit and does not change the order).
static const uint8_t VALUES_SENTINEL = 0x01;

int32_t msi_dict_create_values(msi_ctx *ctx, const uint8_t *values_concat, const uint32_t *offsets, uint32_t n_values,
                               msi_dict **out) {
  if (!ctx || !out || (n_values && (!values_concat || !offsets))) {
    msi_set_error("msi_dict_create_values: invalid argument");
    return MSI_E_INVALID;
  }
  std::vector<uint8_t> concat;
  std::vector<uint32_t> offs(1, 0);
  concat.reserve((n_values ? offsets[n_values] : 0) + n_values);
  for (uint32_t i = 0; i < n_values; ++i) {
    if (offsets[i + 1] < offsets[i] || offsets[i + 1] - offsets[i] > 254) {
      msi_set_error("msi_dict_create_values: value %u has invalid length (max 254 bytes)", i);
      return MSI_E_INVALID;
    }
    concat.push_back(VALUES_SENTINEL);
    concat.insert(concat.end(), values_concat + offsets[i], values_concat + offsets[i + 1]);
    offs.push_back((uint32_t)concat.size());
  }
  MSI_TRY(msi_dict_create(ctx, concat.data(), offs.data(), n_values, out));
  (*out)->values_mode = true;
  return MSI_OK;
}

int32_t msi_dict_search_values(msi_dict *d, const uint8_t *query, uint32_t len, uint32_t max_typos, uint32_t cap,
                               uint32_t *out_idx, uint32_t *out_n, int32_t *out_truncated) {
  if (!d || !d->values_mode || (len && !query) || !out_n || (cap && !out_idx) || max_typos > 2 || len > 249) {
    msi_set_error("msi_dict_search_values: invalid argument (dictionary from msi_dict_create_values, typos 0..2, "
                  "query <= 249 bytes)");
    return MSI_E_INVALID;
  }
  std::vector<uint8_t> q(1, VALUES_SENTINEL);
  q.insert(q.end(), query, query + len);
  std::vector<uint32_t> found;
  bool truncated = false;
  uint32_t lo = 0, hi = 0;  // distance 0: the values the query is a prefix of
  msi_dict_prefix_range(d, q.data(), (uint32_t)q.size(), &lo, &hi);
  for (uint32_t i = lo; i < hi; ++i) found.push_back(i);
  if (max_typos > 0 && d->n_words) {
    const uint32_t c = std::min<uint32_t>(std::max<uint32_t>(cap, 1), 4096);
    std::vector<uint32_t> one(c), two(c);
    uint32_t n1 = 0, n2 = 0;
    msi_typo_query tq;
    tq.word = q.data();
    tq.len = (uint32_t)q.size();
    tq.max_typos = (uint8_t)max_typos;
    tq.is_prefix = 1;
    tq._pad = 0;
    MSI_TRY(msi_dict_lookup(d, &tq, 1, c, c, one.data(), &n1, two.data(), &n2));
    truncated = n1 >= c || n2 >= c;
    found.insert(found.end(), one.begin(), one.begin() + n1);
    if (max_typos > 1) found.insert(found.end(), two.begin(), two.begin() + n2);
  }
  std::sort(found.begin(), found.end());  // FST stream order
  found.erase(std::unique(found.begin(), found.end()), found.end());
  if (found.size() > cap) truncated = true;
  const uint32_t n = (uint32_t)std::min<size_t>(found.size(), cap);
  for (uint32_t i = 0; i < n; ++i) out_idx[i] = found[i];
  *out_n = n;
  if (out_truncated) *out_truncated = truncated ? 1 : 0;
  return MSI_OK;
}

int32_t msi_dict_set_microbatch(msi_dict *d, uint32_t max_wait_us, uint32_t target_words) {
  if (!d) return MSI_E_INVALID;
  std::lock_guard<std::mutex> lk(d->bmu);
  d->microbatch_wait_us = max_wait_us;
  if (target_words) d->microbatch_target = target_words;
  return MSI_OK;
}

int32_t msi_dict_microbatch_stats(msi_dict *d, uint64_t *out_fused_calls, uint64_t *out_fused_launches) {
  if (!d || !out_fused_calls || !out_fused_launches) return MSI_E_INVALID;
  std::lock_guard<std::mutex> lk(d->bmu);
  *out_fused_calls = d->fused_calls;
  *out_fused_launches = d->fused_launches;
  return MSI_OK;
}

int32_t msi_dict_lookup(msi_dict *d, const msi_typo_query *queries, uint32_t n, uint32_t cap_one, uint32_t cap_two,
                        uint32_t *out_one_idx, uint32_t *out_one_cnt, uint32_t *out_two_idx, uint32_t *out_two_cnt) {
  if (!d || (n && (!queries || !out_one_idx || !out_one_cnt || !out_two_idx || !out_two_cnt))) {
    msi_set_error("msi_dict_lookup: invalid argument");
    return MSI_E_INVALID;
  }
  if (n == 0) return MSI_OK;
  if (d->microbatch_wait_us && n < d->microbatch_target)
    return dict_lookup_fused(d, queries, n, cap_one, cap_two, out_one_idx, out_one_cnt, out_two_idx, out_two_cnt);
  return dict_lookup_direct(d, queries, n, cap_one, cap_two, out_one_idx, out_one_cnt, out_two_idx, out_two_cnt);
}

}  // extern "C"

static int32_t dict_lookup_direct(msi_dict *d, const msi_typo_query *queries, uint32_t n, uint32_t cap_one,
                                  uint32_t cap_two, uint32_t *out_one_idx, uint32_t *out_one_cnt,
                                  uint32_t *out_two_idx, uint32_t *out_two_cnt) {
  std::vector<uint8_t> bytes, flags(n);
  std::vector<uint32_t> off(n + 1, 0);
  for (uint32_t i = 0; i < n; ++i) {
    const msi_typo_query &q = queries[i];
    if (q.len && !q.word) {
      msi_set_error("msi_dict_lookup: query %u has a NULL word", i);
      return MSI_E_INVALID;
    }
    bytes.insert(bytes.end(), q.word, q.word + q.len);
    off[i + 1] = (uint32_t)bytes.size();
    flags[i] = (uint8_t)((q.max_typos > 2 ? 2 : q.max_typos) | (q.is_prefix ? 4 : 0));
  }
  std::lock_guard<std::mutex> lk(d->ctx->mu_aux);
  DeviceGuard g(d->ctx->device);
  hipStream_t st = d->ctx->stream_aux;
  MSI_TRY(d->qbytes.ensure(std::max<size_t>(16, bytes.size())));
  MSI_TRY(d->qoff.ensure((n + 1) * sizeof(uint32_t)));
  MSI_TRY(d->qflags.ensure(n));
  MSI_TRY(d->out1.ensure((size_t)n * cap_one * sizeof(uint32_t)));
  MSI_TRY(d->out2.ensure((size_t)n * cap_two * sizeof(uint32_t)));
  MSI_TRY(d->out1c.ensure(n * sizeof(uint32_t)));
  MSI_TRY(d->out2c.ensure(n * sizeof(uint32_t)));
  // pinned staging: [bytes | off | flags] in, [one | two | counts one | counts two] out
  auto a16 = [](size_t x) { return (x + 15) & ~(size_t)15; };
  const size_t in_off = a16(bytes.size()), in_fl = in_off + a16((n + 1) * sizeof(uint32_t)), in_end = in_fl + a16(n);
  const size_t o1 = in_end, o2 = o1 + a16((size_t)n * cap_one * 4), c1 = o2 + a16((size_t)n * cap_two * 4), c2 = c1 + a16(n * 4),
               total = c2 + a16(n * 4);
  if (total > d->h_stage_cap) {
    if (d->h_stage) (void)hipHostFree(d->h_stage);
    d->h_stage = nullptr;
    d->h_stage_cap = 0;
    void *h = nullptr;
    MSI_HIP_TRY(hipHostMalloc(&h, total * 2, hipHostMallocDefault));
    d->h_stage = (uint8_t *)h;
    d->h_stage_cap = total * 2;
  }
  if (!d->h_done) MSI_HIP_TRY(hipEventCreateWithFlags(&d->h_done, hipEventDisableTiming));
  uint8_t *h = d->h_stage;
  if (!bytes.empty()) memcpy(h, bytes.data(), bytes.size());
  memcpy(h + in_off, off.data(), (n + 1) * sizeof(uint32_t));
  memcpy(h + in_fl, flags.data(), n);
  if (!bytes.empty()) MSI_HIP_TRY(hipMemcpyAsync(d->qbytes.p, h, bytes.size(), hipMemcpyHostToDevice, st));
  MSI_HIP_TRY(hipMemcpyAsync(d->qoff.p, h + in_off, (n + 1) * sizeof(uint32_t), hipMemcpyHostToDevice, st));
  MSI_HIP_TRY(hipMemcpyAsync(d->qflags.p, h + in_fl, n, hipMemcpyHostToDevice, st));
  MSI_TRY(enqueue_lookup(d, d->qbytes.as<uint8_t>(), d->qoff.as<uint32_t>(), d->qflags.as<uint8_t>(), n, cap_one,
                         cap_two, d->out1.as<uint32_t>(), d->out1c.as<uint32_t>(), d->out2.as<uint32_t>(),
                         d->out2c.as<uint32_t>()));
  MSI_HIP_TRY(hipMemcpyAsync(h + o1, d->out1.p, (size_t)n * cap_one * sizeof(uint32_t), hipMemcpyDeviceToHost, st));
  MSI_HIP_TRY(hipMemcpyAsync(h + o2, d->out2.p, (size_t)n * cap_two * sizeof(uint32_t), hipMemcpyDeviceToHost, st));
  MSI_HIP_TRY(hipMemcpyAsync(h + c1, d->out1c.p, n * sizeof(uint32_t), hipMemcpyDeviceToHost, st));
  MSI_HIP_TRY(hipMemcpyAsync(h + c2, d->out2c.p, n * sizeof(uint32_t), hipMemcpyDeviceToHost, st));
  MSI_HIP_TRY(hipEventRecord(d->h_done, st));
  // Wait by polling the event with short sleeps: hipStreamSynchronize busy-waits (a fifth of the keyword leg's host CPU
  // went there), a blocking-sync event sleeps until an interrupt that took up to hundreds of milliseconds on this box.
  for (uint32_t spin = 0;; ++spin) {
    const hipError_t q = hipEventQuery(d->h_done);
    if (q == hipSuccess) break;
    if (q != hipErrorNotReady) MSI_HIP_TRY(q);
    if (spin >= 4) {
      struct timespec ts = {0, 20000};
      nanosleep(&ts, nullptr);
    }
  }
  memcpy(out_one_idx, h + o1, (size_t)n * cap_one * sizeof(uint32_t));
  memcpy(out_two_idx, h + o2, (size_t)n * cap_two * sizeof(uint32_t));
  memcpy(out_one_cnt, h + c1, n * sizeof(uint32_t));
  memcpy(out_two_cnt, h + c2, n * sizeof(uint32_t));
  return MSI_OK;
}

extern "C" {

int32_t msi_dict_match_time(msi_dict *d, uint64_t *out_launches, double *out_ms_total) {
  if (!d || !out_launches || !out_ms_total) return MSI_E_INVALID;
  std::lock_guard<std::mutex> lk(d->ctx->mu_aux);
  DeviceGuard g(d->ctx->device);
  MSI_HIP_TRY(hipStreamSynchronize(d->ctx->stream_aux));
  d->match_timer.drain(out_launches, out_ms_total);
  return MSI_OK;
}

int32_t msi_dict_get_stats(const msi_dict *d, msi_dict_stats *out) {
  if (!d || !out) return MSI_E_INVALID;
  out->lookup_launches = d->lookup_launches;
  out->dict_bytes = d->dict_bytes;
  out->pairs_scanned = 0;
  if (d->pairs.p) {
    std::lock_guard<std::mutex> lk(d->ctx->mu_aux);
    DeviceGuard g(d->ctx->device);
    u64 v = 0;
    if (hipStreamSynchronize(d->ctx->stream_aux) == hipSuccess &&
        hipMemcpy(&v, d->pairs.p, sizeof(u64), hipMemcpyDeviceToHost) == hipSuccess)
      out->pairs_scanned = v;
  }
  return MSI_OK;
}

}  // extern "C"
