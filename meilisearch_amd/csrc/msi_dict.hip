// msi_dict.hip — S2: batched typo-tolerant term lookup over the words
// dictionary staged in HBM, gfx950.
//
// Replaces find_one_typo_derivations / find_one_two_typo_derivations
// (crates/milli/src/search/new/query_term/compute_derivations.rs:75-168), i.e.
// `fst.search_with_state(Levenshtein DFA ∩/∪ StartsWith)` with the 150/50 caps
// (search/new/limits.rs:7-9).  Design:
//
//   HBM layout  the sorted dictionary is staged as 16-byte slots (one word per
//               slot, zero padded) + byte/char lengths; words longer than 16
//               bytes keep their bytes in a flat side array and are matched by a
//               second, much smaller launch of the same kernel.
//   dict_match  one lane per dictionary word, 64 consecutive words per wave tile,
//               words live in registers as a 128-bit byte queue while the wave
//               loops over the queries of its chunk.  Distance = banded (2k+1
//               diagonals) optimal-string-alignment DP over code points, all lanes
//               in lock step so the query characters are wave-uniform scalars.
//               The first-letter rule turns into index ranges: the words that
//               share the query's first char are one contiguous range [lo,hi) of
//               the sorted dictionary; outside it only "distance <= 1" can match
//               and a 3-compare prefilter removes almost every word.
//   ordering    every wave owns a contiguous dictionary segment, so ballots +
//               v_mbcnt give per-segment lists already in fst stream order;
//               dict_finalize concatenates segments and applies the cap logic.
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
#include <vector>

#include "msi_common.h"

typedef unsigned long long u64;

namespace {

constexpr int QC = 64;            // queries per workgroup chunk (6-bit query slot in a pair)
constexpr int DW = 4;             // waves per workgroup (all on the same query chunk)
constexpr int QSTRIDE = 256;      // code points reserved per query
constexpr int QL = 32;            // query code points staged in LDS (longer queries read HBM)
constexpr int PQ = 256;           // pair-queue ring entries per wave (power of two, >= 63 + 128)
constexpr int DINF = 1 << 20;
constexpr uint32_t QNONE = 0xFFFFFFFFu;  // "no such query char": never equals a word char
constexpr uint32_t WNONE = 0xFFFFFFFDu;  // "no such word char"

struct alignas(64) QueryMeta {   // one 64-byte scalar load per (tile, query) step
  uint32_t m;        // chars
  uint32_t budget;   // 1 or 2 (0 = skip)
  uint32_t prefix;
  uint32_t _pad;
  uint32_t lo, hi;   // dictionary range of the words starting with the query's first char
  u64 sig;           // char-presence signature (bit = sig_bit(code point))
  // (first, second) char pairs of the four one-edit shapes that change the first char
  u64 p12;           // (q1, q2): substitute q0 -> w[1..2];  delete q0 -> w[0..1]
  u64 p01;           // (q0, q1): insert before q0 -> w[1..2]
  u64 p10;           // (q1, q0): swap q0 q1 -> w[0..1]
  u64 _pad2;
};
static_assert(sizeof(QueryMeta) == 64, "QueryMeta is one 64-byte line");

// 6-bit hash of a code point for the char-presence signatures.
__host__ __device__ __forceinline__ uint32_t sig_bit(uint32_t cp) { return (cp * 0x9E3779B1u) >> 26; }
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

// Query code points of the lane's OWN query: the first QL from LDS, the rest from HBM.
struct QChars {
  const uint32_t *lds;   // this lane's row of the staged chunk
  const uint32_t *glb;   // this lane's row of a.qchars
  int m;
  __device__ __forceinline__ uint32_t at(int t) const {  // t is wave-uniform
    if (t < 0 || t >= m) return QNONE;
    return t < QL ? lds[t] : glb[t];
  }
};

// Banded optimal-string-alignment distance (insert / delete / substitute /
// adjacent transposition, each cost 1 — levenshtein_automata 0.2.1 with
// transposition_cost_one = true, crates/milli/src/search/mod.rs:32-34) between
// the lane's query (m chars) and the lane's word (nch chars): every lane of the
// wave works on its OWN (query, word) pair.  The band is 5 diagonals (|i-j| <= 2)
// for every lane; the result is exact whenever the true value is <= K (K = 1 or 2
// per lane) and > K otherwise.  prefix: minimum over the prefixes of the word
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

// First three chars (WNONE when absent) and char-presence signature of the lane's word.
struct WordKeys {
  u64 w01, w12, sig;
};
template <class Reader>
__device__ __forceinline__ WordKeys scan_word(Reader rd, int nc, int nmax) {
  uint32_t wc0 = WNONE, wc1 = WNONE, wc2 = WNONE;
  u64 sig = 0;
  for (int j = 0; j < nmax; ++j) {
    const uint32_t c = rd.next_char();
    if (j < nc) {
      sig |= 1ull << sig_bit(c);
      wc0 = j == 0 ? c : wc0;
      wc1 = j == 1 ? c : wc1;
      wc2 = j == 2 ? c : wc2;
    }
  }
  WordKeys k;
  k.w01 = pair_key(wc0, wc1);
  k.w12 = pair_key(wc1, wc2);
  k.sig = sig;
  return k;
}

// ---- matching kernel -----------------------------------------------------------

struct DictArgs {
  const uint4 *slots;       // [n_words]
  const uint8_t *blen;      // [n_words] byte length
  const uint8_t *nchars;    // [n_words] char count
  const uint8_t *flat;      // concatenated bytes
  const uint32_t *offs;     // [n_words+1]
  const uint32_t *long_idx; // [n_long] indices of words longer than 16 bytes (ascending)
  uint32_t n_words;
  uint32_t n_items;         // words handled by this launch (n_words or n_long)
  const QueryMeta *qm;      // [nq]
  const uint32_t *qchars;   // [nq][QSTRIDE]
  uint32_t nq;
  uint32_t nseg;            // dictionary segments (a multiple of DW)
  uint32_t cap1, cap2, capx;
  uint32_t *lists;          // [nq][nseg][cap1+cap2+capx]
  uint32_t *cnts;           // [nq][nseg][3]
  u64 *pairs;               // stats: (query, word) pairs that reached a DP lane
};

// Two phases per 64-word tile, both with every lane busy:
//   filter  lane = word; the wave walks the queries of its chunk (wave-uniform
//           query data) and applies the length / first-char tests; survivors are
//           appended, in lane order, to a per-wave ring of (word lane, query, class)
//           pairs with ballots (no atomics);
//   match   whenever 64 pairs are queued (and at the end of the tile) every lane
//           takes ONE pair — its own word and its own query — and runs the banded
//           OSA DP, so the DP lanes are dense instead of following the ~5 % of the
//           (word, query) grid that survives the filter.
// Hits are appended to the per-(query, segment, class) lists in dictionary order:
// pairs of one (query, class) are contiguous in the ring and ordered by word.
template <bool LONG>
__global__ __launch_bounds__(DW * 64) void dict_match_kernel(DictArgs a, const QueryMeta *__restrict__ qmeta) {
  __shared__ uint32_t s_qch[QC][QL];          // staged query code points
  __shared__ uint32_t s_qmb[QC];              // m | budget << 16 | prefix << 24
  __shared__ uint4 s_slot[DW][64];            // the wave's current tile
  __shared__ uint32_t s_wmeta[DW][64];        // nc | bl << 8
  __shared__ uint32_t s_widx[DW][64];
  __shared__ uint32_t s_pq[DW][PQ];
  __shared__ uint32_t s_cnt[DW][QC][3];
  const uint32_t lane = threadIdx.x & 63;
  const uint32_t wave = threadIdx.x >> 6;
  const uint32_t bps = a.nseg / DW;           // workgroups per query chunk
  const uint32_t chunk = blockIdx.x / bps;
  const uint32_t seg = (blockIdx.x % bps) * DW + wave;
  const uint32_t q_begin = chunk * QC;
  const uint32_t nqc = min((uint32_t)QC, a.nq - q_begin);
  for (uint32_t i = threadIdx.x; i < nqc * QL; i += DW * 64)
    s_qch[i / QL][i % QL] = a.qchars[(size_t)(q_begin + i / QL) * QSTRIDE + (i % QL)];
  for (uint32_t i = threadIdx.x; i < nqc; i += DW * 64) {
    const QueryMeta &q = a.qm[q_begin + i];
    s_qmb[i] = q.m | (q.budget << 16) | (q.prefix << 24);
  }
  for (uint32_t i = lane; i < QC * 3; i += 64) (&s_cnt[wave][0][0])[i] = 0;
  __syncthreads();

  const uint32_t n_tiles = (a.n_items + 63) / 64;
  const uint32_t t0 = (uint32_t)((u64)n_tiles * seg / a.nseg);
  const uint32_t t1 = (uint32_t)((u64)n_tiles * (seg + 1) / a.nseg);
  const uint32_t stride_l = a.cap1 + a.cap2 + a.capx;
  const u64 lower = (1ull << lane) - 1ull;
  u64 pairs = 0;
  uint32_t head = 0, qn = 0;  // ring state (wave-uniform)
  bool tile_ascii = false;

  // ---- match phase: lanes [0, n) take one queued pair each ---------------------
  auto drain = [&](uint32_t n) {
    const bool act = lane < n;
    const uint32_t desc = s_pq[wave][(head + lane) & (PQ - 1)];
    const uint32_t wl = desc & 63, ql = (desc >> 6) & 63, cls = (desc >> 12) & 1;
    const uint32_t qlc = act ? ql : 0;
    const uint4 slot = s_slot[wave][wl];
    const uint32_t wm = s_wmeta[wave][wl];
    const uint32_t widx = s_widx[wave][wl];
    const int nc = (int)(wm & 0xFF);
    const uint32_t qmb = s_qmb[qlc];
    const int m = (int)(qmb & 0xFFFF);
    const uint32_t budget = (qmb >> 16) & 0xFF;
    const bool prefix = (qmb >> 24) != 0;
    const int K = cls ? 1 : (int)budget;
    QChars qs;
    qs.lds = &s_qch[qlc][0];
    qs.glb = a.qchars + (size_t)(q_begin + qlc) * QSTRIDE;
    qs.m = act ? m : 0;
    int d;
    if (LONG) {
      const uint32_t o0 = act ? a.offs[widx] : 0, o1 = act ? a.offs[widx + 1] : 0;
      FlatReader r{a.flat + o0, a.flat + o1};
      d = osa_pair(r, nc, act, qs, K, prefix);
    } else if (tile_ascii) {
      SlotReader<true> r{slot.x, slot.y, slot.z, slot.w};
      d = osa_pair(r, nc, act, qs, K, prefix);
    } else {
      SlotReader<false> r{slot.x, slot.y, slot.z, slot.w};
      d = osa_pair(r, nc, act, qs, K, prefix);
    }
    // category: 0 = S1 (same first char, distance 1), 1 = S2 (distance 2), 2 = X
    uint32_t cat = 3;
    if (act) {
      if (cls == 0) cat = d == 1 ? 0u : ((budget == 2 && d == 2) ? 1u : 3u);
      else cat = d <= 1 ? 2u : 3u;
    }
    // pairs of one query are contiguous: segment = run of equal query slots
    const uint32_t prev_ql = __shfl_up(ql, 1);
    const bool is_start = act && (lane == 0 || prev_ql != ql);
    const u64 starts = __ballot(is_start);
    const u64 upto = lower | (1ull << lane);
    const uint32_t p = 63 - __clzll((long long)(starts & upto));      // segment start (act lanes only)
    const u64 above = starts & ~upto;
    const uint32_t nxt = above ? (uint32_t)__ffsll((long long)above) - 1 : n;
    const u64 segmask = (nxt >= 64 ? ~0ull : ((1ull << nxt) - 1ull)) & ~((1ull << p) - 1ull);
    uint32_t *lst = a.lists + ((size_t)(q_begin + qlc) * a.nseg + seg) * stride_l;
#pragma unroll
    for (uint32_t c = 0; c < 3; ++c) {
      const u64 hb = __ballot(cat == c);
      if (hb == 0) continue;
      const uint32_t cap = c == 0 ? a.cap1 : (c == 1 ? a.cap2 : a.capx);
      const uint32_t off = c == 0 ? 0 : (c == 1 ? a.cap1 : a.cap1 + a.cap2);
      const uint32_t base = act ? s_cnt[wave][qlc][c] : 0;
      if (cat == c) {
        const uint32_t pos = base + __popcll(hb & segmask & lower);
        if (pos < cap) lst[off + pos] = widx;
      }
      __builtin_amdgcn_wave_barrier();
      if (is_start) s_cnt[wave][qlc][c] = base + __popcll(hb & segmask);
    }
    __builtin_amdgcn_wave_barrier();
    pairs += n;
    head = (head + n) & (PQ - 1);
    qn -= n;
  };

  for (uint32_t t = t0; t < t1; ++t) {
    const uint32_t item = t * 64 + lane;
    const bool in_range = item < a.n_items;
    uint32_t idx = 0xFFFFFFFFu;
    if (in_range) idx = LONG ? a.long_idx[item] : item;
    uint32_t bl = 0, nc = 0;
    uint4 slot = make_uint4(0, 0, 0, 0);
    if (in_range) {
      bl = a.blen[idx];
      nc = a.nchars[idx];
      slot = a.slots[idx];
    }
    // words longer than a slot belong to the LONG launch; empty words never match
    const bool mine = in_range && bl > 0 && (LONG ? true : bl <= 16);
    tile_ascii = !LONG && (__ballot(mine && nc != bl) == 0);
    s_slot[wave][lane] = slot;
    s_wmeta[wave][lane] = nc | (bl << 8);
    s_widx[wave][lane] = idx;
    // per-word filter keys: first three chars as (w0,w1) / (w1,w2) pairs and the
    // char-presence signature of the whole word
    WordKeys wk;
    {
      const int ncm = mine ? (int)nc : 0;
      const int nmax = wave_max_i32(ncm);
      if (LONG) {
        const uint32_t o0 = mine ? a.offs[idx] : 0, o1 = mine ? a.offs[idx + 1] : 0;
        wk = scan_word(FlatReader{a.flat + o0, a.flat + o1}, ncm, nmax);
      } else if (tile_ascii) {
        wk = scan_word(SlotReader<true>{slot.x, slot.y, slot.z, slot.w}, ncm, nmax);
      } else {
        wk = scan_word(SlotReader<false>{slot.x, slot.y, slot.z, slot.w}, ncm, nmax);
      }
    }
    const u64 w01 = wk.w01, w12 = wk.w12, wsig = wk.sig;
    // dictionary index range covered by this tile (ascending within the tile)
    const uint32_t idx_first = __builtin_amdgcn_readfirstlane(__shfl(idx, 0));
    uint32_t idx_last;
    {
      const u64 bm = __ballot(in_range);
      const int last_lane = 63 - __clzll((long long)bm);
      idx_last = __builtin_amdgcn_readfirstlane(__shfl(idx, last_lane));
    }
    __builtin_amdgcn_wave_barrier();

    // ---- filter phase: query data is wave-uniform (scalar loads), tests branch-free --
    for (uint32_t ql = 0; ql < nqc; ++ql) {
      const QueryMeta qm = qmeta[q_begin + ql];   // one 64-byte line, scalar loads
      const uint32_t budget = qm.budget;
      if (budget == 0) continue;
      const uint32_t lo = qm.lo, hi = qm.hi;
      const bool tile_hits_s = (idx_last >= lo) & (idx_first < hi);
      const bool tile_all_s = (idx_first >= lo) & (idx_last < hi);
      if (!tile_hits_s & (budget != 2)) continue;   // one-typo words only live in [lo, hi)
      const int m = (int)qm.m;
      const bool prefix = qm.prefix != 0;
      const int K = (int)budget;
      const bool same_first = mine & (idx >= lo) & (idx < hi);
      // every char of the query but <= K must occur in the word (and, without the
      // prefix rule, vice versa): each edit introduces at most one new char
      const uint32_t q_not_w = __popcll(qm.sig & ~wsig);
      const uint32_t w_not_q = prefix ? 0u : __popcll(wsig & ~qm.sig);
      const uint32_t sigd = max(q_not_w, w_not_q);
      // length window for K edits: nc + K >= m, and nc <= m + K unless the prefix rule applies
      const int ncK_s = (int)nc + K, ncK_x = (int)nc + 1;
      const bool len_s = (ncK_s >= m) & (prefix | ((int)nc <= m + K));
      const bool len_x = (ncK_x >= m) & (prefix | ((int)nc <= m + 1));
      const bool act_s = tile_hits_s & same_first & len_s & (sigd <= (uint32_t)K);
      // A word with another first char can only be at distance <= 1 through ONE edit
      // that touches position 0: substitute q0 (w[1..] = q[1..]), delete q0 (w = q[1..]),
      // insert before q0 (w[1..] = q), or swap q0 q1 (w = q1 q0 q[2..]).  Two chars of
      // each shape are tested here; the DP decides.  (Queries under 3 chars: no char test.)
      const bool shape = (m < 3) | (w12 == qm.p12) | (w01 == qm.p12) | (w12 == qm.p01) | (w01 == qm.p10);
      const bool act_x = (budget == 2) & !tile_all_s & mine & !same_first & len_x & (sigd <= 1u) & shape;
      const u64 ms = __ballot(act_s);
      const u64 mx = __ballot(act_x);
      if ((ms | mx) == 0) continue;
      if (ms) {
        if (act_s) s_pq[wave][(head + qn + __popcll(ms & lower)) & (PQ - 1)] = lane | (ql << 6);
        qn += __popcll(ms);
      }
      if (mx) {
        if (act_x) s_pq[wave][(head + qn + __popcll(mx & lower)) & (PQ - 1)] = lane | (ql << 6) | (1u << 12);
        qn += __popcll(mx);
      }
      __builtin_amdgcn_wave_barrier();
      while (qn >= 64) drain(64);
    }
    while (qn > 0) drain(qn < 64 ? qn : 64);  // the ring refers to this tile's lanes
    __builtin_amdgcn_wave_barrier();
  }
  __builtin_amdgcn_wave_barrier();
  for (uint32_t i = lane; i < nqc * 3; i += 64) {
    const uint32_t ql = i / 3, c = i % 3;
    const uint32_t cap = c == 0 ? a.cap1 : (c == 1 ? a.cap2 : a.capx);
    a.cnts[((size_t)(q_begin + ql) * a.nseg + seg) * 3 + c] = min(s_cnt[wave][ql][c], cap);
  }
  if (lane == 0 && pairs) atomicAdd(a.pairs, pairs);
}

// ---- query preparation -------------------------------------------------------------

// One thread per query: decode UTF-8, derive the first-char dictionary range.
__global__ void dict_prep_kernel(const uint8_t *__restrict__ qbytes, const uint32_t *__restrict__ qoff,
                                 const uint8_t *__restrict__ qflags, uint32_t nq,
                                 const uint4 *__restrict__ slots, uint32_t n_words,
                                 QueryMeta *__restrict__ qm, uint32_t *__restrict__ qchars) {
  const uint32_t q = blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= nq) return;
  const uint8_t *s = qbytes + qoff[q];
  const uint32_t len = qoff[q + 1] - qoff[q];
  QueryMeta r;
  r.m = 0;
  r.budget = 0;
  r.prefix = (qflags[q] >> 2) & 1;
  r._pad = 0;
  r._pad2 = 0;
  r.lo = r.hi = 0;
  r.sig = 0;
  r.p12 = r.p01 = r.p10 = 0;
  const uint32_t bud = qflags[q] & 3;
  if (len >= 1 && len <= 250 && bud >= 1) {  // MAX_WORD_LENGTH, compute_derivations.rs:180-192
    uint32_t *out = qchars + (size_t)q * QSTRIDE;
    uint32_t n = 0, i = 0;
    while (i < len) {
      const uint32_t b0 = s[i];
      const uint32_t cl = utf8_len(b0);
      uint32_t cp = cl == 1 ? b0 : (cl == 2 ? (b0 & 0x1F) : (cl == 3 ? (b0 & 0x0F) : (b0 & 0x07)));
      for (uint32_t e = 1; e < cl; ++e) cp = (cp << 6) | ((i + e < len ? s[i + e] : 0) & 0x3F);
      i += cl;
      out[n++] = cp;
    }
    r.m = n;
    r.budget = bud > 2 ? 2 : bud;
    for (uint32_t c = 0; c < n; ++c) r.sig |= 1ull << sig_bit(out[c]);
    const uint32_t q0 = out[0], q1 = n >= 2 ? out[1] : QNONE, q2 = n >= 3 ? out[2] : QNONE;
    r.p12 = pair_key(q1, q2);
    r.p01 = pair_key(q0, q1);
    r.p10 = pair_key(q1, q0);
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
  }
  qm[q] = r;
}

// ---- finalisation --------------------------------------------------------------------

// Ascending list of dictionary indices.
struct ArrStream {
  const uint32_t *p;
  uint32_t n, i;
  __device__ bool done() const { return i >= n; }
  __device__ uint32_t peek() const { return p[i]; }
  __device__ void pop() { ++i; }
};

struct MergedStream {  // main ∪ long, ascending
  ArrStream a, b;
  __device__ bool done() const { return a.done() && b.done(); }
  __device__ uint32_t peek() const {
    if (a.done()) return b.peek();
    if (b.done()) return a.peek();
    const uint32_t x = a.peek(), y = b.peek();
    return x < y ? x : y;
  }
  __device__ void pop() {
    if (a.done()) { b.pop(); return; }
    if (b.done()) { a.pop(); return; }
    if (a.peek() < b.peek()) a.pop(); else b.pop();
  }
};

struct FinalArgs {
  const QueryMeta *qm;
  const uint32_t *lists, *cnts;       // main launch
  const uint32_t *llists, *lcnts;     // long-word launch
  uint32_t nq, nseg, lnseg, cap1, cap2, capx;
  uint32_t *comp;                     // [nq][2][cap1+cap2+capx] compacted class lists
  uint32_t *out_one, *out_one_cnt, *out_two, *out_two_cnt;
};

constexpr int FIN_THREADS = 256;

// One workgroup per query.  Phase 1 (parallel): the per-segment lists of each class
// are concatenated (segments are contiguous dictionary ranges, so concatenation in
// segment order is ascending) into one list per (launch, class) with a block-wide
// prefix sum over the segment counts.  Phase 2 (one lane): the reference's cap
// logic in its closed form over those six short lists.
__global__ __launch_bounds__(FIN_THREADS) void dict_finalize_kernel(FinalArgs f) {
  __shared__ uint32_t s_scan[3][FIN_THREADS];
  __shared__ uint32_t s_tot[2][3];
  const uint32_t q = blockIdx.x;
  const uint32_t tid = threadIdx.x;
  const uint32_t st = f.cap1 + f.cap2 + f.capx;
  const uint32_t caps[3] = {f.cap1, f.cap2, f.capx};
  const uint32_t offs[3] = {0, f.cap1, f.cap1 + f.cap2};
  uint32_t *comp = f.comp + (size_t)q * 2 * st;
  const bool live = f.qm[q].budget != 0;
  if (live) {
    for (uint32_t src = 0; src < 2; ++src) {
      const uint32_t ns = src ? f.lnseg : f.nseg;
      const uint32_t *lists = (src ? f.llists : f.lists) + (size_t)q * ns * st;
      const uint32_t *cnts = (src ? f.lcnts : f.cnts) + (size_t)q * ns * 3;
      const uint32_t per = (ns + FIN_THREADS - 1) / FIN_THREADS;
      const uint32_t seg0 = min(ns, tid * per), seg1 = min(ns, seg0 + per);
      uint32_t loc[3] = {0, 0, 0};
      for (uint32_t seg = seg0; seg < seg1; ++seg) {
        loc[0] += cnts[seg * 3 + 0];
        loc[1] += cnts[seg * 3 + 1];
        loc[2] += cnts[seg * 3 + 2];
      }
      __syncthreads();  // s_scan reuse
      for (int c = 0; c < 3; ++c) s_scan[c][tid] = loc[c];
      __syncthreads();
      for (uint32_t o = 1; o < FIN_THREADS; o <<= 1) {
        uint32_t v[3];
        for (int c = 0; c < 3; ++c) v[c] = tid >= o ? s_scan[c][tid - o] : 0;
        __syncthreads();
        for (int c = 0; c < 3; ++c) s_scan[c][tid] += v[c];
        __syncthreads();
      }
      for (int c = 0; c < 3; ++c) {
        uint32_t pos = s_scan[c][tid] - loc[c];  // exclusive
        if (loc[c] && pos < caps[c]) {
          for (uint32_t seg = seg0; seg < seg1 && pos < caps[c]; ++seg) {
            const uint32_t n = cnts[seg * 3 + c];
            const uint32_t *src_l = lists + (size_t)seg * st + offs[c];
            for (uint32_t i = 0; i < n && pos < caps[c]; ++i) comp[src * st + offs[c] + pos++] = src_l[i];
          }
        }
        if (tid == FIN_THREADS - 1) s_tot[src][c] = min(s_scan[c][tid], caps[c]);
      }
    }
  }
  __syncthreads();
  if (tid != 0) return;
  uint32_t *one = f.out_one + (size_t)q * f.cap1;
  uint32_t *two = f.out_two + (size_t)q * f.cap2;
  uint32_t n1 = 0, n2 = 0;
  if (live) {
    __threadfence_block();
    auto mk = [&](uint32_t cls) {
      MergedStream s;
      s.a = ArrStream{comp + offs[cls], s_tot[0][cls], 0};
      s.b = ArrStream{comp + st + offs[cls], s_tot[1][cls], 0};
      return s;
    };
    // two = first cap2 of (X ∪ S2)
    uint32_t tstar = 0xFFFFFFFFu;
    {
      MergedStream s2 = mk(1), sx = mk(2);
      while (n2 < f.cap2 && !(s2.done() && sx.done())) {
        uint32_t v;
        if (s2.done()) { v = sx.peek(); sx.pop(); }
        else if (sx.done()) { v = s2.peek(); s2.pop(); }
        else if (s2.peek() < sx.peek()) { v = s2.peek(); s2.pop(); }
        else { v = sx.peek(); sx.pop(); }
        two[n2++] = v;
      }
      if (n2 == f.cap2 && n2 > 0) tstar = two[n2 - 1];
    }
    // one = first cap1 of (S1 ∪ {x in X : x > t*})
    {
      MergedStream s1 = mk(0), sx = mk(2);
      if (tstar == 0xFFFFFFFFu) {
        while (!sx.done()) sx.pop();  // two never filled: no X word reaches `one`
      } else {
        while (!sx.done() && sx.peek() <= tstar) sx.pop();
      }
      while (n1 < f.cap1 && !(s1.done() && sx.done())) {
        uint32_t v;
        if (s1.done()) { v = sx.peek(); sx.pop(); }
        else if (sx.done()) { v = s1.peek(); s1.pop(); }
        else if (s1.peek() < sx.peek()) { v = s1.peek(); s1.pop(); }
        else { v = sx.peek(); sx.pop(); }
        one[n1++] = v;
      }
    }
  }
  f.out_one_cnt[q] = n1;
  f.out_two_cnt[q] = n2;
}

}  // namespace

// ================================================================== host object

struct msi_dict {
  msi_ctx *ctx = nullptr;
  uint32_t n_words = 0, n_long = 0;
  DevBuf slots, blen, nchars, flat, offs, long_idx;
  // scratch (guarded by ctx->mu_aux)
  DevBuf qbytes, qoff, qflags, qm, qchars, lists, cnts, llists, lcnts, comp, pairs, out1, out1c, out2, out2c;
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
    bool done = false;
  };
  std::mutex bmu;
  std::condition_variable bcv;
  std::vector<Pending *> bqueue;
  bool bleader_active = false;
  uint32_t microbatch_wait_us = 0, microbatch_target = 256;
  uint64_t fused_calls = 0, fused_launches = 0;
  bool values_mode = false;  // msi_dict_create_values: every word carries the sentinel first byte
};

// accessors for msi_keyword.hip
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
  const uint32_t stride_l = cap1 + cap2 + capx;
  MSI_TRY(d->qm.ensure((size_t)n * sizeof(QueryMeta)));
  MSI_TRY(d->qchars.ensure((size_t)n * QSTRIDE * sizeof(uint32_t)));
  hipLaunchKernelGGL(dict_prep_kernel, dim3((n + 127) / 128), dim3(128), 0, st, d_qbytes, d_qoff, d_qflags, n,
                     d->slots.as<uint4>(), d->n_words, d->qm.as<QueryMeta>(), d->qchars.as<uint32_t>());
  const uint32_t nchunks = (n + QC - 1) / QC;
  const uint32_t target_waves = (uint32_t)ctx->n_cu * 20;  // 5 waves per SIMD
  auto seg_for = [&](uint32_t n_items) -> uint32_t {
    const uint32_t n_tiles = std::max<uint32_t>(1, (n_items + 63) / 64);
    uint32_t nseg = (target_waves + nchunks - 1) / nchunks;
    // bound the scratch lists to ~1 GiB (of 288 GB)
    const uint64_t per_seg = (uint64_t)n * stride_l * sizeof(uint32_t);
    const uint32_t mem_cap = (uint32_t)std::max<uint64_t>(1, (1024ull << 20) / std::max<uint64_t>(1, per_seg));
    nseg = std::min(nseg, mem_cap);
    nseg = std::max<uint32_t>(1, std::min(nseg, n_tiles));
    return ((nseg + DW - 1) / DW) * DW;  // the DW waves of a workgroup share a query chunk
  };
  const uint32_t nseg = seg_for(d->n_words);
  const uint32_t lnseg = seg_for(d->n_long);
  MSI_TRY(d->lists.ensure((size_t)n * nseg * stride_l * sizeof(uint32_t)));
  MSI_TRY(d->cnts.ensure((size_t)n * nseg * 3 * sizeof(uint32_t)));
  MSI_TRY(d->llists.ensure((size_t)n * lnseg * stride_l * sizeof(uint32_t)));
  MSI_TRY(d->lcnts.ensure((size_t)n * lnseg * 3 * sizeof(uint32_t)));
  MSI_TRY(d->comp.ensure((size_t)n * 2 * stride_l * sizeof(uint32_t)));
  MSI_HIP_TRY(hipMemsetAsync(d->lcnts.p, 0, (size_t)n * lnseg * 3 * sizeof(uint32_t), st));
  MSI_HIP_TRY(hipMemsetAsync(d->cnts.p, 0, (size_t)n * nseg * 3 * sizeof(uint32_t), st));
  DictArgs a;
  a.slots = d->slots.as<uint4>();
  a.blen = d->blen.as<uint8_t>();
  a.nchars = d->nchars.as<uint8_t>();
  a.flat = d->flat.as<uint8_t>();
  a.offs = d->offs.as<uint32_t>();
  a.long_idx = d->long_idx.as<uint32_t>();
  a.n_words = d->n_words;
  a.qm = d->qm.as<QueryMeta>();
  a.qchars = d->qchars.as<uint32_t>();
  a.nq = n;
  a.cap1 = cap1;
  a.cap2 = cap2;
  a.capx = capx;
  a.pairs = d->pairs.as<u64>();
  if (d->n_words) {
    a.n_items = d->n_words;
    a.nseg = nseg;
    a.lists = d->lists.as<uint32_t>();
    a.cnts = d->cnts.as<uint32_t>();
    d->match_timer.begin(ctx, st);
    hipLaunchKernelGGL(dict_match_kernel<false>, dim3((nseg / DW) * nchunks), dim3(DW * 64), 0, st, a, a.qm);
    d->match_timer.end(ctx);
  }
  if (d->n_long) {
    a.n_items = d->n_long;
    a.nseg = lnseg;
    a.lists = d->llists.as<uint32_t>();
    a.cnts = d->lcnts.as<uint32_t>();
    hipLaunchKernelGGL(dict_match_kernel<true>, dim3((lnseg / DW) * nchunks), dim3(DW * 64), 0, st, a, a.qm);
  }
  FinalArgs f;
  f.qm = d->qm.as<QueryMeta>();
  f.lists = d->lists.as<uint32_t>();
  f.cnts = d->cnts.as<uint32_t>();
  f.llists = d->llists.as<uint32_t>();
  f.lcnts = d->lcnts.as<uint32_t>();
  f.nq = n;
  f.nseg = nseg;
  f.lnseg = lnseg;
  f.cap1 = cap1;
  f.cap2 = cap2;
  f.capx = capx;
  f.comp = d->comp.as<uint32_t>();
  f.out_one = d_one;
  f.out_one_cnt = d_one_cnt;
  f.out_two = d_two;
  f.out_two_cnt = d_two_cnt;
  hipLaunchKernelGGL(dict_finalize_kernel, dim3(n), dim3(FIN_THREADS), 0, st, f);
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
  // host-side staging: slots, lengths, long-word list; sortedness check
  std::vector<uint4> slots(std::max<uint32_t>(1, n_words));
  std::vector<uint8_t> blen(std::max<uint32_t>(1, n_words)), nch(std::max<uint32_t>(1, n_words));
  std::vector<uint32_t> long_idx;
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
    blen[i] = (uint8_t)len;
    uint32_t chars = 0;
    for (uint32_t b = 0; b < len; ++b) chars += (w[b] & 0xC0) != 0x80;
    nch[i] = (uint8_t)chars;
    if (len > 16) long_idx.push_back(i);
  }
  DeviceGuard g(ctx->device);
  std::lock_guard<std::mutex> lk(ctx->mu_aux);
  msi_dict *d = new msi_dict();
  d->ctx = ctx;
  d->n_words = n_words;
  d->n_long = (uint32_t)long_idx.size();
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
  up(d->blen, blen.data(), n_words);
  up(d->nchars, nch.data(), n_words);
  up(d->flat, words_concat, flat_bytes);
  up(d->offs, offsets, ((size_t)n_words + 1) * sizeof(uint32_t));
  up(d->long_idx, long_idx.data(), long_idx.size() * sizeof(uint32_t));
  if (s == MSI_OK) s = d->pairs.ensure(sizeof(u64));
  if (s == MSI_OK && hipMemsetAsync(d->pairs.p, 0, sizeof(u64), st) != hipSuccess) s = MSI_E_HIP;
  if (s == MSI_OK && hipStreamSynchronize(st) != hipSuccess) {
    msi_set_error("msi_dict_create: stream synchronize failed");
    s = MSI_E_HIP;
  }
  if (s != MSI_OK) {
    DevBuf *bufs[] = {&d->slots, &d->blen, &d->nchars, &d->flat, &d->offs, &d->long_idx, &d->pairs};
    for (DevBuf *b : bufs) b->release();
    delete d;
    return s;
  }
  d->dict_bytes = (uint64_t)n_words * (sizeof(uint4) + 2) + flat_bytes + ((uint64_t)n_words + 1) * 4;
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
  {
  std::lock_guard<std::mutex> lk(ctx->mu_aux);
  DeviceGuard g(ctx->device);
  (void)hipStreamSynchronize(d->ctx->stream_aux);
  DevBuf *bufs[] = {&d->slots, &d->blen, &d->nchars, &d->flat, &d->offs, &d->long_idx, &d->qbytes, &d->qoff,
                    &d->qflags, &d->qm, &d->qchars, &d->lists, &d->cnts, &d->llists, &d->lcnts, &d->comp, &d->pairs,
                    &d->out1, &d->out1c, &d->out2, &d->out2c};
  for (DevBuf *b : bufs) b->release();
  d->match_timer.release();
  delete d;
  }
  msi_ctx_release(ctx);
}

uint32_t msi_dict_len(const msi_dict *d) { return d ? d->n_words : 0; }

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
  for (;;) {
    if (d->bleader_active) {
      d->bcv.notify_all();
      d->bcv.wait(lk, [&] { return me.done || !d->bleader_active; });
      if (me.done) {
        if (me.status != MSI_OK) msi_set_error("%s", me.error.c_str());
        return me.status;
      }
    }
    d->bleader_active = true;
    const auto deadline = std::chrono::steady_clock::now() + std::chrono::microseconds(d->microbatch_wait_us);
    d->bcv.wait_until(lk, deadline, [&] { return queued_words() >= d->microbatch_target; });
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
      p->status = st;
      p->error = err;
      p->done = true;
    }
    d->bleader_active = false;
    lk.unlock();
    d->bcv.notify_all();
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
  if (!bytes.empty()) MSI_HIP_TRY(hipMemcpyAsync(d->qbytes.p, bytes.data(), bytes.size(), hipMemcpyHostToDevice, st));
  MSI_HIP_TRY(hipMemcpyAsync(d->qoff.p, off.data(), (n + 1) * sizeof(uint32_t), hipMemcpyHostToDevice, st));
  MSI_HIP_TRY(hipMemcpyAsync(d->qflags.p, flags.data(), n, hipMemcpyHostToDevice, st));
  MSI_TRY(enqueue_lookup(d, d->qbytes.as<uint8_t>(), d->qoff.as<uint32_t>(), d->qflags.as<uint8_t>(), n, cap_one,
                         cap_two, d->out1.as<uint32_t>(), d->out1c.as<uint32_t>(), d->out2.as<uint32_t>(),
                         d->out2c.as<uint32_t>()));
  MSI_HIP_TRY(hipMemcpyAsync(out_one_idx, d->out1.p, (size_t)n * cap_one * sizeof(uint32_t), hipMemcpyDeviceToHost, st));
  MSI_HIP_TRY(hipMemcpyAsync(out_two_idx, d->out2.p, (size_t)n * cap_two * sizeof(uint32_t), hipMemcpyDeviceToHost, st));
  MSI_HIP_TRY(hipMemcpyAsync(out_one_cnt, d->out1c.p, n * sizeof(uint32_t), hipMemcpyDeviceToHost, st));
  MSI_HIP_TRY(hipMemcpyAsync(out_two_cnt, d->out2c.p, n * sizeof(uint32_t), hipMemcpyDeviceToHost, st));
  MSI_HIP_TRY(hipStreamSynchronize(st));
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
