// msi_vs.hip — S1: exact cosine k-NN over one vector store, gfx950.
//
// Replaces VectorStore::nns_by_vector / nns_by_item for one store
// (crates/milli/src/vector/store.rs:615-675,1036-1093) in its exact (linear
// scan) mode.  Design:
//
//   HBM layout   rows are re-tiled at upload into MFMA-fragment order: a tile is
//                16 rows, split into KB = dpad/16 blocks of 1 KiB; block (t,kb)
//                holds float4 #l (l = g*16+i) = row 16t+i, columns 16kb+4g..+3.
//                One wave-wide 16-byte load therefore reads one contiguous KiB
//                and lands directly in v_mfma_f32_16x16x4_f32's A layout.
//   vs_scan      streams every (allowed) tile once per batch of up to 48 queries:
//                D[16 rows][16 queries] += A·B per 16-query tile, query fragments
//                in LDS (the only LDS use).  Two epilogues:
//                  dense   every score is written to a [query][row] matrix (the
//                          strided sample pass, and stores too small to sample);
//                  sparse  scores are compared with a per-query threshold and the
//                          rare survivors are appended to per-query global lists.
//   thresholds   the dense sample pass (~2·sqrt(K'·N) rows) gives each query the
//                r-th best sampled score; r is chosen so that fewer than K'
//                survivors is a <1e-5 event, and that event is detected
//                (survivors < K') and re-run exhaustively — never silently wrong.
//   vs_select    radix-select of the K' best 64-bit keys (score desc, row asc).
//   vs_rescore   recomputes the K' candidates with the REFERENCE arithmetic
//                (sequential f32 mul+add, arroy/hannoy's scalar path), orders
//                them by (distance, docid) and proves that no unselected row can
//                reach the k-th place (error bound on the fast scan); if the
//                proof fails the query is re-run exhaustively (vs_exhaustive).
//
// The fast scan's summation order never reaches the caller: every returned
// distance is the reference's scalar f32 arithmetic.
#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <chrono>
#include <condition_variable>
#include <memory>

#include "msi_common.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
// The rows of a sweep are read once: non-temporal 16-byte loads (MSI_VS_NT=0 at build time: plain loads, for comparison)
#ifndef MSI_VS_NT
#define MSI_VS_NT 1
#endif
__device__ inline float4 vs_stream_load(const float4 *p) {
#if MSI_VS_NT && !defined(MSI_HIP_EMULATED)
  const f32x4 v = __builtin_nontemporal_load(reinterpret_cast<const f32x4 *>(p));
  return make_float4(v[0], v[1], v[2], v[3]);
#else
  return *p;
#endif
}
#define MSI_VS_STREAM_LOAD(p) vs_stream_load(p)
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef unsigned long long u64;
__device__ inline i32x4 vs_stream_load_i8(const i32x4 *p) {
#if MSI_VS_NT && !defined(MSI_HIP_EMULATED)
  return __builtin_nontemporal_load(p);
#else
  return *p;
#endif
}

// Where the 16-byte piece (row i of the tile, column group g) of a KiB block of the f32 / bf16 ROW tiles lives, in pieces.
// Rounds 1-5: g * 16 + i — the lane order of the MFMA A operand, so the sweep's lane l read piece l.  Round 6: i * 4 + g —
// a row's four pieces of a block are one 64-byte sector.  The sweep's wave still reads the same contiguous KiB (lane l reads
// piece MSI_TILE_PIECE(l & 15, l >> 4): a permutation inside the block), but everything that reads ONE row — the second
// opinion on the candidates, the reference rescoring, the exhaustive pass, get_vector — now touches a quarter of the
// sectors: 3 KB per 768-float row instead of 12.  MSI_VS_TILE_ROW_SECTORS=0 at build time: the old order
// (A/B measurements).  The int8 copy and the queries' fragments keep the operand order.
#ifndef MSI_VS_TILE_ROW_SECTORS
#define MSI_VS_TILE_ROW_SECTORS 1
#endif
#if MSI_VS_TILE_ROW_SECTORS
#define MSI_TILE_PIECE(i, g) ((i) * 4u + (g))
#else
#define MSI_TILE_PIECE(i, g) ((g) * 16u + (i))
#endif

namespace {

constexpr int SCAN_WAVES = 8;            // waves per workgroup (512 threads)
constexpr int SCAN_GROUP = 8;            // KiB blocks per software-pipeline stage
constexpr uint32_t KP_MAX = 2048;        // K' supported by select/rescore (k <= 2048; k + max(12, k/4) candidates are rescored)
constexpr int SEL_THREADS = 256;
constexpr int SEL_SORTCAP = 2048;        // u64 keys sorted in LDS by vs_select
constexpr int SEL_UNROLL = 8;            // independent key loads in flight per thread in the selection's passes
constexpr int QT = 16;                   // queries per MFMA tile
constexpr int NQT_MAX = 12;              // query tiles per HBM sweep (12 on the int8 copy, 6 with the bf16x2 contraction, 3 otherwise)
constexpr int NQT_F32_MAX = 6;           // ... of the sweeps over the f32 rows
constexpr int NQ_MAX = QT * NQT_MAX;     // queries per HBM sweep
constexpr int CNT_PAD = 32;              // u32 stride of the per-query counters (one 128-B line each)
constexpr size_t LDS_MAX = 160 * 1024;

__device__ __forceinline__ u64 make_key_desc(float s, uint32_t row) {
  return ((u64)(~f32_to_ord(s)) << 32) | row;
}
__device__ __forceinline__ float key_desc_score(u64 key) {
  return ord_to_f32(~(uint32_t)(key >> 32));
}

// ------------------------------------------------------------------ upload path

// Row-major chunk -> tiled layout.  One thread per output float4.
__global__ void vs_tile_rows_kernel(const float *__restrict__ rows, uint64_t row0, uint64_t n_chunk,
                                    uint32_t dim, uint32_t KB, float4 *__restrict__ tiles) {
  uint64_t idx = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;  // float4 index inside chunk
  uint64_t chunk_tiles = (n_chunk + 15) / 16;
  uint64_t total = chunk_tiles * KB * 64;
  if (idx >= total) return;
  uint32_t lane = idx & 63;
  uint64_t blk = idx >> 6;
  uint32_t kb = blk % KB;
  uint64_t t = blk / KB;
  uint32_t i = lane & 15, g = lane >> 4;
  uint64_t r = t * 16 + i;
  uint32_t k0 = kb * 16 + g * 4;
  float v[4] = {0.f, 0.f, 0.f, 0.f};
  if (r < n_chunk) {
    const float *src = rows + r * (uint64_t)dim;
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (k0 + j < dim) v[j] = src[k0 + j];
  }
  tiles[(row0 / 16) * KB * 64 + blk * 64 + MSI_TILE_PIECE(i, g)] = make_float4(v[0], v[1], v[2], v[3]);
}

// bf16 store: row-major f32 chunk -> bf16 (round to nearest even) tiles.  Block (t,kb)
// holds, for lane l = g*16+i, the 8 bf16 = row 16t+i, columns 32kb+8g..+7: one
// wave-wide 16-byte load is the A operand of v_mfma_f32_16x16x32_bf16 as it is.
__global__ void vs_tile_rows_bf16_kernel(const float *__restrict__ rows, uint64_t row0, uint64_t n_chunk,
                                         uint32_t dim, uint32_t KB, bf16x8 *__restrict__ tiles) {
  uint64_t idx = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;  // 16-byte slot inside chunk
  uint64_t chunk_tiles = (n_chunk + 15) / 16;
  uint64_t total = chunk_tiles * KB * 64;
  if (idx >= total) return;
  uint32_t lane = idx & 63;
  uint64_t blk = idx >> 6;
  uint32_t kb = blk % KB;
  uint64_t t = blk / KB;
  uint32_t i = lane & 15, g = lane >> 4;
  uint64_t r = t * 16 + i;
  uint32_t k0 = kb * 32 + g * 8;
  bf16x8 v;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    float x = 0.f;
    if (r < n_chunk && k0 + j < dim) x = rows[r * (uint64_t)dim + k0 + j];
    v[j] = (__bf16)x;
  }
  tiles[(row0 / 16) * KB * 64 + blk * 64 + MSI_TILE_PIECE(i, g)] = v;
}

// Element (row, column) of the tiled store as f32, for the canonical (sequential) paths.
template <bool S16>
__device__ __forceinline__ float tile_elem(const void *__restrict__ tiles, uint32_t KB, uint32_t row, uint32_t k) {
  if (S16) {
    const __bf16 *p = reinterpret_cast<const __bf16 *>(tiles);
    const uint64_t slot = ((uint64_t)(row >> 4) * KB + (k >> 5)) * 64 + MSI_TILE_PIECE(row & 15, (k >> 3) & 3);
    return (float)p[slot * 8 + (k & 7)];
  }
  const float *p = reinterpret_cast<const float *>(tiles);
  const uint64_t slot = ((uint64_t)(row >> 4) * KB + (k >> 4)) * 64 + MSI_TILE_PIECE(row & 15, (k >> 2) & 3);
  return p[slot * 4 + (k & 3)];
}

// Canonical row norms of a bf16 store (values are the rounded ones).
__global__ void vs_row_norms_bf16_kernel(const void *__restrict__ tiles, uint64_t n_rows_total, uint32_t KB,
                                         uint32_t dim, float *__restrict__ norm, float *__restrict__ inv_norm) {
  uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  uint64_t padded = ((n_rows_total + 15) / 16) * 16;
  if (r >= padded) return;
  if (r >= n_rows_total) {
    norm[r] = 0.f;
    inv_norm[r] = 0.f;
    return;
  }
  float acc = 0.f;
  for (uint32_t k = 0; k < dim; ++k) {
    const float x = tile_elem<true>(tiles, KB, (uint32_t)r, k);
    acc = __fadd_rn(acc, __fmul_rn(x, x));
  }
  const float n = msi_sqrt_rn(acc);
  norm[r] = n;
  inv_norm[r] = 1.0f / n;
}

// Canonical row norms from the tiled layout: pn = sqrt(sum_k x_k*x_k), sequential
// f32 mul+add in column order (arroy/hannoy scalar path).  One thread per row.
__global__ void vs_row_norms_kernel(const float4 *__restrict__ tiles, uint64_t row0,
                                    uint64_t n_rows_total, uint32_t KB, float *__restrict__ norm,
                                    float *__restrict__ inv_norm) {
  uint64_t r = row0 + (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  uint64_t padded = ((n_rows_total + 15) / 16) * 16;
  if (r >= padded) return;
  if (r >= n_rows_total) {
    norm[r] = 0.f;
    inv_norm[r] = 0.f;
    return;
  }
  uint64_t t = r >> 4;
  uint32_t i = r & 15;
  const float4 *base = tiles + t * KB * 64;
  float acc = 0.f;
  for (uint32_t kb = 0; kb < KB; ++kb) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      float4 v = base[(uint64_t)kb * 64 + MSI_TILE_PIECE(i, (uint32_t)g)];
      acc = __fadd_rn(acc, __fmul_rn(v.x, v.x));
      acc = __fadd_rn(acc, __fmul_rn(v.y, v.y));
      acc = __fadd_rn(acc, __fmul_rn(v.z, v.z));
      acc = __fadd_rn(acc, __fmul_rn(v.w, v.w));
    }
  }
  float n = msi_sqrt_rn(acc);
  norm[r] = n;
  inv_norm[r] = 1.0f / n;  // +inf for zero rows: always "degenerate" in the scan
}

__global__ void vs_check_sorted_kernel(const uint32_t *__restrict__ docids, uint64_t n,
                                       uint32_t *__restrict__ bad) {
  uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i + 1 < n && docids[i] >= docids[i + 1]) *bad = 1;
}

// Incremental update (msi_vs_update): new 16-byte slot (row r', block kb, column group g) := the same (kb, g) slot of
// the row it comes from — a row of the old store, or one of the freshly tiled added rows (bit 31 of map[r']).  The
// tiled layout only permutes whole slots when rows move, so the kernel is the same for f32 and bf16 rows.
__global__ void vs_regather_kernel(const uint4 *__restrict__ old_tiles, const uint4 *__restrict__ add_tiles,
                                   const uint32_t *__restrict__ old_docids, const uint32_t *__restrict__ add_docids,
                                   const uint32_t *__restrict__ map, uint64_t n_new, uint32_t KB,
                                   uint4 *__restrict__ new_tiles, uint32_t *__restrict__ new_docids) {
  const uint64_t idx = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const uint64_t total = ((n_new + 15) / 16) * KB * 64;
  if (idx >= total) return;
  const uint32_t lane = idx & 63, i = lane & 15, g = lane >> 4;
  const uint64_t blk = idx >> 6;
  const uint32_t kb = (uint32_t)(blk % KB);
  const uint64_t r = (blk / KB) * 16 + i;
  uint4 v = make_uint4(0, 0, 0, 0);
  uint32_t docid = 0xFFFFFFFFu;
  if (r < n_new) {
    const uint32_t m = map[r], src = m & 0x7FFFFFFFu;
    const uint4 *from = (m & 0x80000000u) ? add_tiles : old_tiles;
    v = from[((uint64_t)(src >> 4) * KB + kb) * 64 + MSI_TILE_PIECE(src & 15, g)];
    docid = (m & 0x80000000u) ? add_docids[src] : old_docids[src];
  }
  new_tiles[blk * 64 + MSI_TILE_PIECE(i, g)] = v;
  if (kb == 0 && g == 0) new_docids[r] = docid;  // padding rows of the last tile included
}

// ------------------------------------------------------------- query preparation

// One workgroup per query slot j (0 .. 16*nqt): queries row-major [nq][dim] ->
//   qfrag [nqt][KB][64] float4 — MFMA B fragments (tile t = j/16, lane l = g*16 + j%16:
//         query j, columns 16kb+4g..+3),
//   qrow  [16*nqt][dpad] — zero-padded rows for the canonical rescoring,
//   canonical |q| (sequential f32 mul+add, then correctly rounded sqrt), its
//   reciprocal and the threshold that marks a row degenerate (pn*qn <= EPS).
__global__ __launch_bounds__(256) void vs_prep_queries_kernel(
    const float *__restrict__ q, uint32_t nq, uint32_t dim, uint32_t KB, float4 *__restrict__ qfrag,
    bf16x8 *__restrict__ qfrag_bf, float *__restrict__ qrow, float *__restrict__ qn,
    float *__restrict__ inv_qn, float *__restrict__ degth, uint32_t store16) {
  MSI_DYNAMIC_LDS(smem);
  float *row = reinterpret_cast<float *>(smem);  // [dpad]
  const uint32_t j = blockIdx.x;
  const bool rows16 = store16 == 1;   // (store16 == 2: f32 rows, hi-only query fragments — the bf16x2 contraction)
  const uint32_t dpad = rows16 ? KB * 32 : KB * 16;
  for (uint32_t k = threadIdx.x; k < dpad; k += blockDim.x)
    row[k] = (j < nq && k < dim) ? q[(uint64_t)j * dim + k] : 0.f;
  __syncthreads();
  for (uint32_t k = threadIdx.x; k < dpad; k += blockDim.x) qrow[(uint64_t)j * dpad + k] = row[k];
  const uint32_t t = j / QT, jj = j % QT;
  if (store16 == 0) {
    for (uint32_t idx = threadIdx.x; idx < KB * 4; idx += blockDim.x) {
      const uint32_t kb = idx >> 2, g = idx & 3;
      const float4 v = *reinterpret_cast<const float4 *>(row + kb * 16 + g * 4);
      qfrag[((uint64_t)t * KB + kb) * 64 + g * 16 + jj] = v;
    }
  }
  // bf16x3 fragments: block pair p = (2p, 2p+1), lane (g, jj) holds the 8 columns
  // 32p+4g..+3 and 32p+16+4g..+3 (the same 8 a row lane holds after two 16-byte
  // loads), split as hi = bf16(x), lo = bf16(x - hi); layout [t][KB/2][hi|lo][64].
  // (bf16 store: KB counts 32-column blocks and lane (g, jj) holds columns 32p+8g..+7)
  const uint32_t n_pairs = rows16 ? KB : KB / 2;
  for (uint32_t idx = threadIdx.x; idx < n_pairs * 4; idx += blockDim.x) {
    const uint32_t p = idx >> 2, g = idx & 3;
    bf16x8 hi, lo;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float x = rows16 ? row[p * 32 + g * 8 + e] : row[p * 32 + (e >> 2) * 16 + g * 4 + (e & 3)];
      const __bf16 h = (__bf16)x;
      hi[e] = h;
      lo[e] = (__bf16)(x - (float)h);
    }
    if (store16 == 2) {   // bf16x2: the hi halves only, [t][KB/2][64]
      qfrag_bf[((uint64_t)t * n_pairs + p) * 64 + g * 16 + jj] = hi;
      continue;
    }
    const uint64_t base = (((uint64_t)t * n_pairs + p) * 2) * 64 + g * 16 + jj;
    qfrag_bf[base] = hi;
    qfrag_bf[base + 64] = lo;
  }
  if (threadIdx.x == 0) {
    float acc = 0.f;
    for (uint32_t k = 0; k < dim; ++k) acc = __fadd_rn(acc, __fmul_rn(row[k], row[k]));
    const float n = msi_sqrt_rn(acc);
    qn[j] = n;
    inv_qn[j] = n > 0.f ? 1.0f / n : 0.f;
    // row is (conservatively) degenerate when pn*qn <= EPS  <=>  1/pn >= qn/EPS
    degth[j] = n * (0.999f / FLT_EPSILON);
  }
}

// ------------------------------------------------------------------- filter path

// The ITEMS a filtered sweep visits (round 6: row-granular).  An item is 16 entries of `rows`: the rows of one 16 x 16 MFMA
// tile, each entry a row index of the store, bit 31 set for a row that is loaded but not allowed (0xFFFFFFFF: padding).
// hannoy's linear mode only touches candidate rows (vector/store.rs:1079-1080); rounds 1-5 streamed every 16-row tile that
// held an allowed row — 8 / 15 / 16 times the allowed rows' bytes at 10 / 1 / 0.1 %.  Since the row tiles
// keep a row's pieces of a KiB block in one 64-byte sector (MSI_TILE_PIECE) a wave can gather 16 ARBITRARY rows into the A
// operand at sector granularity, so a workgroup of this kernel (8 192 consecutive rows) writes its allowed rows either
//   compacted  ceil(allowed / 16) items of allowed rows only (gathered: ~2.5 TB/s of useful bytes), or
//   as tiles   one item per 16-row tile that holds an allowed row, its other rows flagged (streamed: ~5.8 TB/s),
// whichever moves fewer bytes per unit of bandwidth: compacted when allowed rows x gather_num < tile rows x gather_den
// (MSI_VS_GATHER_PCT, default 250 = 2.5: below ~40 % density on uniform filters; 0: always, >= 1600: never).  One global atomic per workgroup
// reserves its slice; items land in workgroup arrival order, which only permutes the order rows are visited in.
// small: [0] items, [1] allowed rows, [2] items written compacted, [3] items written as tiles (msi_vs_filter_stats).
constexpr int FT_SUB = 8;
constexpr uint32_t FROW_OFF = 0x80000000u;   // entry flag: the row is not allowed (padding: every bit set)
__device__ __forceinline__ uint32_t frow_id(uint32_t e) { return e == 0xFFFFFFFFu ? 0u : (e & 0x7FFFFFFFu); }
__device__ __forceinline__ uint32_t frow_safe(uint32_t e) { return (e >> 31) ? 0u : e; }   // (per-row arrays are not padded to tiles)
__global__ __launch_bounds__(1024) void vs_filter_rows_kernel(
    const uint32_t *__restrict__ docids, uint64_t n_rows, const u64 *__restrict__ fbits,
    uint64_t nbits, uint32_t *__restrict__ rows, uint32_t *__restrict__ small, uint32_t gather_num, uint32_t gather_den) {
  __shared__ uint32_t s_rows[FT_SUB * 16], s_tiles[FT_SUB * 16];
  __shared__ uint32_t s_base, s_compact, s_total, s_items;
  const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const uint64_t padded = ((n_rows + 15) / 16) * 16;
  const uint64_t blk_row0 = (uint64_t)blockIdx.x * (1024ull * FT_SUB);
  u64 bal[FT_SUB];  // wave-uniform: allowed rows of this wave's 4 tiles, per sub-chunk
#pragma unroll
  for (int u = 0; u < FT_SUB; ++u) {
    const uint64_t r = blk_row0 + (uint64_t)u * 1024 + threadIdx.x;
    bool ok = false;
    if (r < n_rows) {
      const uint32_t id = docids[r];
      if ((uint64_t)id < nbits) ok = (fbits[id >> 6] >> (id & 63)) & 1ull;
    }
    const u64 b = __ballot(ok);
    bal[u] = b;
    uint32_t c = 0;
#pragma unroll
    for (int t = 0; t < 4; ++t) c += ((b >> (16 * t)) & 0xFFFFull) != 0;
    if (lane == 0) {
      s_tiles[u * 16 + wave] = c;
      s_rows[u * 16 + wave] = (uint32_t)__popcll(b);
    }
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    uint32_t tot_t = 0, tot_r = 0;
    for (int i = 0; i < FT_SUB * 16; ++i) {
      const uint32_t ct = s_tiles[i], cr = s_rows[i];
      s_tiles[i] = tot_t;
      s_rows[i] = tot_r;
      tot_t += ct;
      tot_r += cr;
    }
    const bool compact = (u64)tot_r * gather_num < (u64)tot_t * 16ull * gather_den;
    const uint32_t items = compact ? (tot_r + 15) / 16 : tot_t;
    s_compact = compact ? 1u : 0u;
    s_total = tot_r;
    s_items = items;
    s_base = items ? atomicAdd(&small[0], items) : 0;
    if (items) {
      atomicAdd(&small[1], tot_r);
      atomicAdd(&small[compact ? 2 : 3], items);
    }
  }
  __syncthreads();
  const uint64_t base = (uint64_t)s_base * 16;
  if (s_compact) {
#pragma unroll
    for (int u = 0; u < FT_SUB; ++u) {
      const u64 b = bal[u];
      if ((b >> lane) & 1ull) {
        const uint64_t r = blk_row0 + (uint64_t)u * 1024 + threadIdx.x;
        const uint32_t pos = s_rows[u * 16 + wave] + (uint32_t)__popcll(b & ((1ull << lane) - 1ull));
        rows[base + pos] = (uint32_t)r;
      }
    }
    for (uint32_t i = s_total + threadIdx.x; i < s_items * 16; i += 1024) rows[base + i] = 0xFFFFFFFFu;
    return;
  }
  const uint32_t ts = lane >> 4;  // tile slot of this lane inside the wave-row
#pragma unroll
  for (int u = 0; u < FT_SUB; ++u) {
    const uint64_t r = blk_row0 + (uint64_t)u * 1024 + threadIdx.x;
    const u64 b = bal[u];
    if (r < padded && ((b >> (16 * ts)) & 0xFFFFull)) {
      uint32_t before = 0;
      for (uint32_t t = 0; t < ts; ++t) before += ((b >> (16 * t)) & 0xFFFFull) != 0;
      rows[base + (uint64_t)(s_tiles[u * 16 + wave] + before) * 16 + (lane & 15)] = (uint32_t)r | (((b >> lane) & 1ull) ? 0u : FROW_OFF);
    }
  }
}

// ------------------------------------------------------------------------ scan

struct ScanArgs {
  const float4 *tiles;
  const float *inv_norm;
  const float4 *qfrag;           // [nqt][KB][64]
  const float *theta;            // [NQ_MAX] sparse: pass if !(score < theta)
  const float *degth;            // [NQ_MAX]
  const uint32_t *n_items_ptr;   // number of items: 16-entry groups of `rows` (tiles of the store when rows == null)
  const uint32_t *rows;          // nullable (FILT): the items of a filtered sweep, vs_filter_rows_kernel
  u64 *gkeys;                    // sparse: [NQ_MAX][capg]
  uint32_t *gcnt;                // sparse: [NQ_MAX][CNT_PAD]
  uint32_t *overflow;            // set to 1 if a global list overflowed
  float *dense;                  // dense: [NQ_MAX][dstride], entry = item*16 + row_in_tile
  uint64_t n_rows;
  uint32_t dstride;
  uint32_t capg;
  uint32_t KB;
  uint32_t stride;               // 1 = every item, S = every S-th item (sample pass)
  uint32_t nq;
};

// NQT = 16-query tiles per sweep; DENSE selects the epilogue; BF3 the contraction:
//   false  v_mfma_f32_16x16x4_f32 on the f32 rows (exact products, 1/16 of the bf16 rate);
//   true   bf16x3: rows and queries are split x = hi + lo (two bf16 each) in registers and
//          hi·hi + hi·lo + lo·hi runs on v_mfma_f32_16x16x32_bf16 — 3 instructions of 8
//          passes per 32 columns instead of 8 of 8 passes, ~5x less matrix time, so 48
//          queries per sweep stay HBM-bound.  The dropped lo·lo term and the bf16 rounding
//          of lo cost <= 3·2^-18 relative per product, which the exactness proof's eps
//          carries; returned distances never see it (canonical rescoring).
//   S16    the store holds bf16 rows (half the HBM bytes per row): the 16-byte loads ARE
//          the bf16 A operands, x·(q_hi + q_lo) needs 2 MFMAs per 32 columns and no
//          conversion; LDS holds q_hi and q_lo (2 KiB per 32 columns and query tile).
// MATH: 0 = f32 MFMA, 1 = bf16x3 (hi.hi + hi.lo + lo.hi), 2 = bf16x2 (row hi/lo x query hi: half the LDS per query)
// FILT: the sweep visits the items of `a.rows` — lane (i, g) gathers the pieces of row rows[16 item + i] (a 64-byte sector per
// row and KiB block), the epilogue reads the rows' ids, flags and inverse norms through the same list
template <int WAVES, int NQT, bool DENSE, int MATH, bool S16, bool FILT>
__global__ __launch_bounds__(WAVES * 64) void vs_scan_kernel(ScanArgs a) {
  MSI_DYNAMIC_LDS(smem);
  const uint32_t tid = threadIdx.x;
  const uint32_t lane = tid & 63;
  const uint32_t wave = tid >> 6;
  const uint32_t KB = a.KB;

  float4 *qf = reinterpret_cast<float4 *>(smem);
  for (uint32_t i = tid; i < (MATH == 2 ? NQT * KB * 32 : NQT * KB * 64 * (S16 ? 2 : 1)); i += WAVES * 64) qf[i] = a.qfrag[i];
  __syncthreads();

  const uint32_t qj = lane & 15;   // this lane's query inside a tile (D column)
  const uint32_t g = lane >> 4;    // this lane's row group: rows 4g..4g+3 of the tile
  float th[NQT], dth[NQT];
#pragma unroll
  for (int t = 0; t < NQT; ++t) {
    th[t] = DENSE ? 0.f : a.theta[t * QT + qj];
    dth[t] = a.degth[t * QT + qj];
  }

  // this wave's contiguous share of the item list
  const uint32_t n_all = *a.n_items_ptr;
  const uint32_t n_items = (n_all + a.stride - 1) / a.stride;
  const uint64_t gw = (uint64_t)blockIdx.x * WAVES + wave;
  const uint64_t GW = (uint64_t)gridDim.x * WAVES;
  const uint32_t it0 = (uint32_t)((uint64_t)n_items * gw / GW);
  const uint32_t it1 = (uint32_t)((uint64_t)n_items * (gw + 1) / GW);
  if (it0 >= it1) return;

  const uint32_t GPT = KB / SCAN_GROUP;  // pipeline groups per tile

  auto tile_of = [&](uint32_t it) -> uint32_t { return it * a.stride; };

  // ---- epilogue -----------------------------------------------------------------
  auto epilogue = [&](uint32_t tile, uint32_t it, const f32x4(&acc)[NQT][2]) {
    const uint64_t row0 = (uint64_t)tile * 16 + g * 4;
    uint32_t allowed = 0xF;
    uint32_t ids[4] = {(uint32_t)row0, (uint32_t)row0 + 1u, (uint32_t)row0 + 2u, (uint32_t)row0 + 3u};
    float iv[4];
    if constexpr (FILT) {
      const uint4 e4 = *reinterpret_cast<const uint4 *>(a.rows + (uint64_t)tile * 16 + g * 4);   // (tile: the item's place in the list)
      const uint32_t ee[4] = {e4.x, e4.y, e4.z, e4.w};
      allowed = 0;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        ids[r] = frow_id(ee[r]);
        if (!(ee[r] >> 31)) allowed |= 1u << r;
        iv[r] = (ee[r] >> 31) ? 0.f : a.inv_norm[ids[r]];
      }
    } else {
      const float4 inv = *reinterpret_cast<const float4 *>(a.inv_norm + row0);
      if (row0 + 4 > a.n_rows) allowed = row0 >= a.n_rows ? 0u : ((1u << (a.n_rows - row0)) - 1u);
      iv[0] = inv.x, iv[1] = inv.y, iv[2] = inv.z, iv[3] = inv.w;
    }
#pragma unroll
    for (int t = 0; t < NQT; ++t) {
      float s[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float v = (acc[t][0][r] + acc[t][1][r]) * iv[r];
        if (iv[r] >= dth[t]) v = FLT_MAX;   // pn*qn <= EPS: reference distance is 0 (best)
        if (!(v == v)) v = FLT_MAX;         // NaN: let the canonical rescoring decide
        s[r] = v;
      }
      const uint32_t q = t * QT + qj;
      if (DENSE) {
        float4 o;
        o.x = (allowed & 1u) ? s[0] : -INFINITY;
        o.y = (allowed & 2u) ? s[1] : -INFINITY;
        o.z = (allowed & 4u) ? s[2] : -INFINITY;
        o.w = (allowed & 8u) ? s[3] : -INFINITY;
        *reinterpret_cast<float4 *>(a.dense + (uint64_t)q * a.dstride + (uint64_t)it * 16 + g * 4) = o;
      } else {
        uint32_t pass = 0;
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (!(s[r] < th[t])) pass |= 1u << r;
        pass &= allowed;
        if (q >= a.nq) pass = 0;
        if (pass) {  // rare: the thresholds leave ~1e3 survivors per query and sweep
          uint32_t slot = atomicAdd(&a.gcnt[q * CNT_PAD], (uint32_t)__popc(pass));
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            if (pass & (1u << r)) {
              if (slot < a.capg) a.gkeys[(uint64_t)q * a.capg + slot] = make_key_desc(s[r], ids[r]);
              else *a.overflow = 1;
              ++slot;
            }
          }
        }
      }
    }
  };

  float4 xa[SCAN_GROUP], xb[SCAN_GROUP];
  f32x4 acc[NQT][2];
#pragma unroll
  for (int t = 0; t < NQT; ++t) {
    acc[t][0] = (f32x4){0.f, 0.f, 0.f, 0.f};
    acc[t][1] = (f32x4){0.f, 0.f, 0.f, 0.f};
  }

  uint32_t it_load = it0, sub_load = 0;   // next group to load
  uint32_t it_cmp = it0, sub_cmp = 0;     // next group to compute
  uint32_t tile_load = tile_of(it_load);
  uint32_t tile_cmp = tile_load;

  // FILT: where this lane's row of the item being loaded starts (its piece of block 0)
  // (the NEXT item's entry is requested when an item's loads begin: the list read is off the address path)
  const float4 *row_load = a.tiles;
  uint32_t entry_next = 0;
  auto entry_of = [&](uint32_t item) -> uint32_t { return a.rows[(uint64_t)item * 16 + (lane & 15u)]; };
  auto row_ptr_of = [&](uint32_t entry) -> const float4 * {
    const uint32_t row = frow_id(entry);
    return a.tiles + (uint64_t)(row >> 4) * KB * 64 + MSI_TILE_PIECE(row & 15u, lane >> 4);
  };
  if constexpr (FILT) {
    row_load = row_ptr_of(entry_of(tile_load));
    if (it_load + 1 < it1) entry_next = entry_of(tile_of(it_load + 1));
  }
  auto load_group = [&](float4(&x)[SCAN_GROUP]) {
    // (lane l holds row l & 15, column group l >> 4 of the block: MSI_TILE_PIECE says where that piece lives)
    const float4 *p = FILT ? row_load + (uint64_t)sub_load * SCAN_GROUP * 64
                           : a.tiles + ((uint64_t)tile_load * KB + (uint64_t)sub_load * SCAN_GROUP) * 64 + MSI_TILE_PIECE(lane & 15u, lane >> 4);
#pragma unroll
    for (int u = 0; u < SCAN_GROUP; ++u) x[u] = MSI_VS_STREAM_LOAD(p + u * 64);
    if (++sub_load == GPT) {
      sub_load = 0;
      ++it_load;
      if (it_load < it1) {
        tile_load = tile_of(it_load);
        if constexpr (FILT) {
          row_load = row_ptr_of(entry_next);
          if (it_load + 1 < it1) entry_next = entry_of(tile_of(it_load + 1));
        }
      }
    }
  };
  auto compute_group = [&](const float4(&x)[SCAN_GROUP]) {
    if (S16) {
      // LDS: [t][KB][hi|lo][64] bf16x8; this group covers blocks sub_cmp*8 .. +7
      const bf16x8 *qb = reinterpret_cast<const bf16x8 *>(qf) + (size_t)sub_cmp * SCAN_GROUP * 128 + lane;
#pragma unroll
      for (int u = 0; u < SCAN_GROUP; ++u) {
        bf16x8 xa8;
        __builtin_memcpy(&xa8, &x[u], 16);
#pragma unroll
        for (int t = 0; t < NQT; ++t) {
          const bf16x8 *qt = qb + (size_t)t * KB * 128 + u * 128;
          acc[t][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xa8, qt[0], acc[t][0], 0, 0, 0);
          acc[t][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xa8, qt[64], acc[t][1], 0, 0, 0);
        }
      }
    } else if (MATH == 2) {
      // LDS: [t][KB/2][64] bf16x8 — the queries' hi halves only; the row is split in registers as in bf16x3 and
      // hi.hi + lo.hi is accumulated: the query's rounding (<= 2^-8 relative) is the whole first-order error, the
      // exactness proof carries it (scan_eps) and more candidates are rescored (K').  Half the LDS per query: 96
      // queries share a sweep at d = 768.
      const bf16x8 *qb = reinterpret_cast<const bf16x8 *>(qf) + (size_t)sub_cmp * (SCAN_GROUP / 2) * 64 + lane;
#pragma unroll
      for (int u = 0; u < SCAN_GROUP; u += 2) {
        const float v[8] = {x[u].x, x[u].y, x[u].z, x[u].w, x[u + 1].x, x[u + 1].y, x[u + 1].z, x[u + 1].w};
        bf16x8 hi, lo;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const __bf16 h = (__bf16)v[e];
          hi[e] = h;
          lo[e] = (__bf16)(v[e] - (float)h);
        }
#pragma unroll
        for (int t = 0; t < NQT; ++t) {
          const bf16x8 qh = qb[(size_t)t * (KB / 2) * 64 + (u / 2) * 64];
          acc[t][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(hi, qh, acc[t][0], 0, 0, 0);
          acc[t][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(lo, qh, acc[t][1], 0, 0, 0);
        }
      }
    } else if (MATH == 1) {
      // LDS: [t][KB/2][hi|lo][64] bf16x8; this group covers pairs sub_cmp*4 .. +3
      const bf16x8 *qb = reinterpret_cast<const bf16x8 *>(qf) + (size_t)sub_cmp * (SCAN_GROUP / 2) * 128 + lane;
#pragma unroll
      for (int u = 0; u < SCAN_GROUP; u += 2) {
        const float v[8] = {x[u].x, x[u].y, x[u].z, x[u].w, x[u + 1].x, x[u + 1].y, x[u + 1].z, x[u + 1].w};
        bf16x8 hi, lo;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const __bf16 h = (__bf16)v[e];
          hi[e] = h;
          lo[e] = (__bf16)(v[e] - (float)h);
        }
#pragma unroll
        for (int t = 0; t < NQT; ++t) {
          const bf16x8 *qt = qb + (size_t)t * (KB / 2) * 128 + (u / 2) * 128;
          const bf16x8 qh = qt[0], ql = qt[64];
          acc[t][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(hi, qh, acc[t][0], 0, 0, 0);
          acc[t][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(hi, ql, acc[t][1], 0, 0, 0);
          acc[t][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(lo, qh, acc[t][1], 0, 0, 0);
        }
      }
    } else {
      const float4 *qp = qf + (size_t)sub_cmp * SCAN_GROUP * 64 + lane;
#pragma unroll
      for (int u = 0; u < SCAN_GROUP; ++u) {
#pragma unroll
        for (int t = 0; t < NQT; ++t) {
          const float4 q = qp[(size_t)t * KB * 64 + u * 64];
          // two independent accumulators per query tile: no dependent-MFMA stall
          acc[t][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(x[u].x, q.x, acc[t][0], 0, 0, 0);
          acc[t][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(x[u].y, q.y, acc[t][1], 0, 0, 0);
          acc[t][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(x[u].z, q.z, acc[t][0], 0, 0, 0);
          acc[t][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(x[u].w, q.w, acc[t][1], 0, 0, 0);
        }
      }
    }
    if (++sub_cmp == GPT) {
      epilogue(tile_cmp, it_cmp, acc);
#pragma unroll
      for (int t = 0; t < NQT; ++t) {
        acc[t][0] = (f32x4){0.f, 0.f, 0.f, 0.f};
        acc[t][1] = (f32x4){0.f, 0.f, 0.f, 0.f};
      }
      sub_cmp = 0;
      ++it_cmp;
      if (it_cmp < it1) tile_cmp = tile_of(it_cmp);
    }
  };

  // Software pipeline: loads run one group (8 KiB per wave) ahead of the MFMAs.
  load_group(xa);
  for (;;) {
    bool more = it_load < it1;
    if (more) load_group(xb);
    compute_group(xa);
    if (!more) break;
    more = it_load < it1;
    if (more) load_group(xa);
    compute_group(xb);
    if (!more) break;
  }
}


// ------------------------------------------------------------ int8 candidate sweep
//
// Round 5.  The f32 sweep runs at the machine's streaming ceiling (0.77-0.82 of 8 TB/s): what is left is to stream fewer
// bytes.  Beside its f32 rows a store keeps an INT8 COPY of them — every row divided by its norm and quantised with its
// own scale, dpad bytes per row instead of 4 dpad — and level 0 of the search sweeps THAT copy on v_mfma_i32_16x16x64_i8
// (twice the bf16 rate, exact integer accumulation).  The sweep is still exhaustive (every allowed row is scored) and it
// is still only a CANDIDATE GENERATOR: the K' best rows are rescored from the f32 rows in the reference's arithmetic and
// the exactness proof runs with the quantisation's own bound, so the answers are the same bits as before; what it cannot
// prove falls through to the f32 sweeps (levels of effort, msi_vs::level).
//
//   row      u = x / |x| (the canonical norm), s_x = max|u_i| / 127, xq_i = rint(u_i / s_x), e_x = |u - s_x xq|  (per row)
//   query    v = q / |q|,                      s_q = max|v_i| / 127, qq_i = rint(v_i / s_q), e_q = |v - s_q qq|
//   fast     cos ~ s_x s_q (xq . qq)            — the integer dot product is exact in i32 (d * 127^2 < 2^31 for d <= 133 000)
//   bound    u.v - s_x s_q xq.qq = r_x.v + (s_x xq).r_q,  |r_x.v| <= e_x |v|,  |(s_x xq).r_q| <= (|u| + e_x) e_q   (Cauchy-Schwarz)
//            => |cos - fast| <= e_x + e_q + e_x e_q (+ the f32 terms every level carries): eps of query j uses ITS e_q and the
//            largest e_x of the store (a device scalar, maintained by the quantiser).  On N(0,1) rows e_x ~ 0.008 whatever d.
//
// HBM layout of the copy: tile t = 16 rows = KB8 = dpad / 64 blocks of 1 KiB; in block (t, kb) lane l = g*16+i owns the 16
// bytes of row 16t+i, columns 64kb+16g..+15 — one wave-wide 16-byte load is one contiguous KiB and IS the A operand.  The
// queries' fragments (same k-order, so the lane -> k map of the instruction never matters) sit in LDS: 1 KiB per 64 columns and
// 16 queries — 12 KiB per query tile at d = 768, so 192 queries share a sweep (the bf16x2 sweep: 96).  A wave holds RT row
// tiles at a time so that every 1-KiB query fragment read from LDS feeds RT instructions (one MFMA per read would need
// twice the LDS bandwidth a CU has).
//
// Algorithmic bytes: dpad + 8 per row and sweep (the int8 row, its scale, its inverse norm).

struct Scan8Args {
  const i32x4 *tiles8;
  const float *scale8;           // [rows] s_x (NaN: the row could not be quantised — always a candidate)
  const float *inv_norm;         // [rows] (the degenerate-row rule of the f32 sweeps)
  const uint32_t *i8small;       // [0] largest e_x, [1] largest inverse norm of the store (ordered bits)
  const i32x4 *qfrag8;           // [nqt][KB8][64]
  const float *sqs;              // [NQ_MAX] s_q * |q|: fast score = dot * s_x * sqs, in the f32 sweeps' unit (x.q / |x|)
  const float *theta;
  const float *degth;
  const uint32_t *n_items_ptr;
  const uint32_t *rows;          // nullable (FILT): the items of a filtered sweep, as ScanArgs::rows
  // Rounds 1-5's tile list and allowed-row masks of a filtered sweep: ALWAYS null since round 6 (filtered sweeps go through
  // `rows`).  They stay because the unfiltered kernel's code depends on their uniform branches: without them the same source
  // compiles to 256 registers and 22 spilled ones instead of 230 and none (llvm's scheduling of the epilogue; tools/isa_excerpt.py)
  const uint32_t *list;
  const uint16_t *tmask;
  u64 *gkeys;
  uint32_t *gcnt;
  uint32_t *overflow;
  float *dense;
  uint64_t n_rows;
  uint32_t dstride;
  uint32_t capg;
  uint32_t KB8;
  uint32_t stride;
  uint32_t nq;
};

// the 16 consecutive columns 16m .. 16m + 15 of row i of a tile: KiB block m of an f32 tile (four pieces), half of KiB block
// m / 2 of a bf16 tile (two pieces of eight)
template <bool S16>
__device__ __forceinline__ void vs_row_chunk16(const void *__restrict__ tile_base, uint32_t i, uint32_t m, float (&e)[16]) {
  if constexpr (S16) {
    const bf16x8 *base = reinterpret_cast<const bf16x8 *>(tile_base) + (uint64_t)(m >> 1) * 64;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const bf16x8 v = base[MSI_TILE_PIECE(i, (m & 1u) * 2u + (uint32_t)h)];
#pragma unroll
      for (int j = 0; j < 8; ++j) e[h * 8 + j] = (float)v[j];
    }
  } else {
    const float4 *base = reinterpret_cast<const float4 *>(tile_base) + (uint64_t)m * 64;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const float4 v = base[MSI_TILE_PIECE(i, (uint32_t)g)];
      e[g * 4] = v.x, e[g * 4 + 1] = v.y, e[g * 4 + 2] = v.z, e[g * 4 + 3] = v.w;
    }
  }
}

// f32 / bf16 tiles -> the int8 copy, one workgroup of 256 threads per tile: thread (i = tid & 15, c = tid >> 4) works on row
// i, 16-column chunks c, c + 16, ...  (KB: KiB blocks of the source tile — dpad / 16 of f32 rows, dpad / 32 of bf16 rows)
template <bool S16>
__global__ __launch_bounds__(256) void vs_quantize_rows_kernel(const void *__restrict__ tiles, const float *__restrict__ inv_norm,
                                                               uint64_t n_rows, uint32_t KB, i32x4 *__restrict__ tiles8,
                                                               float *__restrict__ scale8, uint32_t *__restrict__ ex_max_ord) {
  __shared__ float s_red[16][17];
  __shared__ float s_row[16];
  __shared__ int s_bad[16];
  const uint32_t tid = threadIdx.x, i = tid & 15, c = tid >> 4;
  const uint64_t t = blockIdx.x;
  const uint64_t r = t * 16 + i;
  const float inv = r < n_rows ? inv_norm[r] : 0.f;
  const uint32_t NM = S16 ? KB * 2 : KB;   // 16-column chunks of a row
  const uint32_t KB8 = NM / 4;
  if (tid < 16) s_bad[tid] = 0;
  __syncthreads();
  const void *base = reinterpret_cast<const char *>(tiles) + t * KB * 1024;
  float amax = 0.f;
  bool bad = !(inv == inv) || inv == INFINITY;
  for (uint32_t m = c; m < NM; m += 16) {
    float x[16];
    vs_row_chunk16<S16>(base, i, m, x);
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const float e = x[j] * inv;
      if (!(fabsf(e) <= FLT_MAX)) bad = true;   // NaN or infinity
      amax = fmaxf(amax, fabsf(e));
    }
  }
  if (bad) s_bad[i] = 1;
  s_red[i][c] = amax;
  __syncthreads();
  if (c == 0) {
    float m = 0.f;
    for (int j = 0; j < 16; ++j) m = fmaxf(m, s_red[i][j]);
    s_row[i] = m;
  }
  __syncthreads();
  const float rmax = s_row[i];
  const bool row_bad = s_bad[i] != 0;
  const float sx = rmax > 0.f ? rmax / 127.0f : 0.f;
  const float isx = rmax > 0.f ? 127.0f / rmax : 0.f;
  float res = 0.f;
  for (uint32_t m = c; m < NM; m += 16) {
    int q[16];
    float x[16];
    vs_row_chunk16<S16>(base, i, m, x);
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const float e = x[j] * inv;
      float qf = rintf(e * isx);
      qf = fminf(127.f, fmaxf(-127.f, qf));
      if (row_bad) qf = 0.f;
      const float d = __fsub_rn(e, __fmul_rn(sx, qf));
      res = __fadd_rn(res, __fmul_rn(d, d));
      q[j] = (int)qf;
    }
    i32x4 w;
#pragma unroll
    for (int g = 0; g < 4; ++g)
      w[g] = (q[4 * g] & 255) | ((q[4 * g + 1] & 255) << 8) | ((q[4 * g + 2] & 255) << 16) | ((q[4 * g + 3] & 255) << 24);
    tiles8[(t * KB8 + (m >> 2)) * 64 + MSI_TILE_PIECE(i, m & 3u)] = w;
  }
  __syncthreads();
  s_red[i][c] = res;
  __syncthreads();
  if (c == 0) {
    float tot = 0.f;
    for (int j = 0; j < 16; ++j) tot += s_red[i][j];
    // e_x, rounded up: the f32 evaluation of the residual (relative 2^-20 at these lengths) and what u itself carries
    const float ex = msi_sqrt_rn(tot) * 1.001f + 2e-6f;
    if (r < n_rows) {
      scale8[r] = row_bad ? NAN : sx;
      if (!row_bad) atomicMax(ex_max_ord, __float_as_uint(ex));   // (non-negative floats order as their bits)
      // [1]: the largest inverse norm of the store (+inf with a zero row): a sweep whose queries' degenerate-row thresholds
      // all lie above it never loads inv_norm (vs_scan_i8_kernel)
      if (inv == inv && inv > 0.f) atomicMax(ex_max_ord + 1, __float_as_uint(inv));
      if (!(inv == inv)) atomicMax(ex_max_ord + 1, 0x7F800000u);
    } else {
      scale8[r] = 0.f;
    }
  }
}

// The queries of a sweep -> int8 fragments, score scales and each query's proof bound.  Runs behind vs_prep_queries_kernel
// (qrow = the zero-padded rows, qn / inv_qn = canonical norm and reciprocal); one workgroup of 256 threads per query.
__global__ __launch_bounds__(256) void vs_prep_queries_i8_kernel(const float *__restrict__ qrow, const float *__restrict__ qn,
                                                                 const float *__restrict__ inv_qn, uint32_t nq, uint32_t dpad,
                                                                 i32x4 *__restrict__ qfrag8, float *__restrict__ sqs,
                                                                 float *__restrict__ epsq, const uint32_t *__restrict__ ex_max_ord,
                                                                 float eps_base) {
  __shared__ float s_red[256];
  const uint32_t j = blockIdx.x, tid = threadIdx.x;
  const uint32_t KB8 = dpad / 64;
  const float inv = j < nq ? inv_qn[j] : 0.f;
  const float *row = qrow + (uint64_t)j * dpad;
  float amax = 0.f;
  for (uint32_t k = tid; k < dpad; k += 256) amax = fmaxf(amax, fabsf(row[k] * inv));
  s_red[tid] = amax;
  __syncthreads();
  for (uint32_t o = 128; o > 0; o >>= 1) {
    if (tid < o) s_red[tid] = fmaxf(s_red[tid], s_red[tid + o]);
    __syncthreads();
  }
  const float vmax = s_red[0];
  __syncthreads();
  const bool finite = vmax <= FLT_MAX;   // (false for NaN / infinity: the query then proves nothing at this level)
  const float sq = (finite && vmax > 0.f) ? vmax / 127.0f : 0.f;
  const float isq = (finite && vmax > 0.f) ? 127.0f / vmax : 0.f;
  const uint32_t t = j / QT, jj = j % QT;
  float res = 0.f;
  for (uint32_t m = tid; m < dpad / 16; m += 256) {   // 16-column chunk m = (kb8 = m / 4, g = m % 4)
    int q[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const float v = row[m * 16 + e] * inv;
      float qf = finite ? rintf(v * isq) : 0.f;
      qf = fminf(127.f, fmaxf(-127.f, qf));
      const float d = __fsub_rn(v, __fmul_rn(sq, qf));
      res = __fadd_rn(res, __fmul_rn(d, d));
      q[e] = (int)qf;
    }
    i32x4 w;
#pragma unroll
    for (int g = 0; g < 4; ++g)
      w[g] = (q[4 * g] & 255) | ((q[4 * g + 1] & 255) << 8) | ((q[4 * g + 2] & 255) << 16) | ((q[4 * g + 3] & 255) << 24);
    qfrag8[((uint64_t)t * KB8 + (m >> 2)) * 64 + (m & 3) * 16 + jj] = w;
  }
  s_red[tid] = res;
  __syncthreads();
  for (uint32_t o = 128; o > 0; o >>= 1) {
    if (tid < o) s_red[tid] = s_red[tid] + s_red[tid + o];
    __syncthreads();
  }
  if (tid == 0) {
    const float eq = msi_sqrt_rn(s_red[0]) * 1.001f + 2e-6f;
    const float ex = __uint_as_float(*ex_max_ord);
    sqs[j] = sq * (j < nq ? qn[j] : 0.f);
    epsq[j] = finite ? (ex + eq + ex * eq) * 1.002f + eps_base : INFINITY;
  }
}

// NQT 16-query tiles per sweep, RT row tiles per wave and step, GS KiB-blocks per software-pipeline stage (GS divides KB8).
template <int WAVES, int NQT, int RT, int GS, bool DENSE, bool FILT>
__global__ __launch_bounds__(WAVES * 64) void vs_scan_i8_kernel(Scan8Args a) {
  MSI_DYNAMIC_LDS(smem);
  const uint32_t tid = threadIdx.x;
  const uint32_t lane = tid & 63;
  const uint32_t wave = tid >> 6;
  const uint32_t KB8 = a.KB8;
  i32x4 *qf = reinterpret_cast<i32x4 *>(smem);   // [NQT][KB8][64]
  for (uint32_t i = tid; i < NQT * KB8 * 64; i += WAVES * 64) qf[i] = a.qfrag8[i];
  __syncthreads();

  // per-query constants of the epilogue stay in LDS (three registers per query tile otherwise: the accumulators need them)
  float *s_th = reinterpret_cast<float *>(smem + (size_t)NQT * KB8 * 64 * sizeof(i32x4));   // [NQT * QT] thresholds
  float *s_dth = s_th + NQT * QT, *s_sq = s_dth + NQT * QT, *s_thq = s_sq + NQT * QT;
  for (uint32_t i = tid; i < NQT * QT; i += WAVES * 64) {
    const float th = DENSE ? 0.f : a.theta[i], sq = a.sqs[i];
    s_th[i] = th;
    s_dth[i] = a.degth[i];
    s_sq[i] = sq;
    // the sparse epilogue's test in the unit of (integer dot) x (row scale): score >= th  <=>  dot * s_x >= th / sq, taken a few
    // ulps low so that rounding can only ADD survivors (their real scores are computed when they are kept); a query beyond
    // nq keeps nothing, one whose scale is 0 scores 0 everywhere
    float thq = sq > 0.f ? th / sq : (th <= 0.f ? -INFINITY : INFINITY);
    if (sq > 0.f && thq == thq && fabsf(thq) <= FLT_MAX) thq -= fabsf(thq) * 4.8e-7f;
    if (i >= a.nq) thq = INFINITY;
    s_thq[i] = thq;
  }
  __syncthreads();
  // a row is degenerate for query j when 1 / |row| >= degth[j] (pn*qn <= EPS: the reference's distance is 0): if the store's
  // largest inverse norm lies below every threshold of this sweep (always, on sane embeddings) the test cannot fire and the
  // epilogue does not load the rows' inverse norms at all — one dependent global load per tile less in front of the scores
  bool check_deg = false;
  {
    const float inv_max = __uint_as_float(a.i8small[1]);
    for (uint32_t i = 0; i < min(a.nq, (uint32_t)(NQT * QT)); ++i)
      if (inv_max >= s_dth[i]) check_deg = true;
  }

  const uint32_t qj = lane & 15;   // this lane's query inside a tile (D column)
  const uint32_t g = lane >> 4;    // this lane's row group: rows 4g..4g+3 of the tile
  const uint32_t n_all = *a.n_items_ptr;
  const uint32_t n_items = (n_all + a.stride - 1) / a.stride;
  const uint64_t gw = (uint64_t)blockIdx.x * WAVES + wave;
  const uint64_t GW = (uint64_t)gridDim.x * WAVES;
  const uint32_t it0 = (uint32_t)((uint64_t)n_items * gw / GW);
  const uint32_t it1 = (uint32_t)((uint64_t)n_items * (gw + 1) / GW);
  if (it0 >= it1) return;
  const uint32_t GPT = KB8 / GS;

  // (FILT: the "tile" of an item is its place in the list of items, a.rows + 16 tile)
  auto tile_of = [&](uint32_t it) -> uint32_t {
    const uint32_t idx = (it < it1 ? it : it1 - 1) * a.stride;   // (the last group of a wave may be short: its spare slots re-read the last tile)
    if constexpr (FILT) return idx;
    return a.list ? a.list[idx] : idx;
  };

  auto epilogue = [&](uint32_t tile, uint32_t it, const i32x4(&acc)[NQT], const float4 sc) {
    const uint64_t row0 = (uint64_t)tile * 16 + g * 4;
    uint32_t allowed = 0xF;
    uint32_t fids[4] = {0u, 0u, 0u, 0u};   // FILT: the rows' indices (else row0 + r)
    float iv[4] = {0.f, 0.f, 0.f, 0.f};
    auto id_of = [&](int r) -> uint32_t { return FILT ? fids[r] : (uint32_t)(row0 + r); };
    if constexpr (FILT) {
      const uint4 e4 = *reinterpret_cast<const uint4 *>(a.rows + (uint64_t)tile * 16 + g * 4);
      const uint32_t ee[4] = {e4.x, e4.y, e4.z, e4.w};
      allowed = 0;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        fids[r] = frow_id(ee[r]);
        if (!(ee[r] >> 31)) allowed |= 1u << r;
        if (check_deg && !(ee[r] >> 31)) iv[r] = a.inv_norm[fids[r]];
      }
    } else {
      float4 inv = make_float4(0.f, 0.f, 0.f, 0.f);
      if (check_deg) inv = *reinterpret_cast<const float4 *>(a.inv_norm + row0);
      if (a.tmask) allowed = (a.tmask[tile] >> (g * 4)) & 0xF;
      else if (row0 + 4 > a.n_rows) allowed = row0 >= a.n_rows ? 0u : ((1u << (a.n_rows - row0)) - 1u);
      iv[0] = inv.x, iv[1] = inv.y, iv[2] = inv.z, iv[3] = inv.w;
    }
    const float sx[4] = {sc.x, sc.y, sc.z, sc.w};
    if (!DENSE && !check_deg) {
      // The common case of the full sweep, three instructions per score: convert, scale by the row, compare with the query's
      // threshold in that unit (a NaN — a row that could not be quantised — passes).  Everything else happens for survivors only.
#pragma unroll
      for (int t = 0; t < NQT; ++t) {
        const uint32_t q = t * QT + qj;
        const float thq = s_thq[q];
        float f[4];
        uint32_t pass = 0;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          f[r] = (float)acc[t][r] * sx[r];
          if (!(f[r] < thq)) pass |= 1u << r;
        }
        pass &= allowed;
        if (pass) {
          if (q < a.nq) {
            const float sq_t = s_sq[q], th_t = s_th[q];
            uint32_t keep = 0;
            float sv[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              float v = f[r] * sq_t;
              if (!(v == v)) v = FLT_MAX;
              sv[r] = v;
              if ((pass & (1u << r)) && !(v < th_t)) keep |= 1u << r;
            }
            if (keep) {
              uint32_t slot = atomicAdd(&a.gcnt[q * CNT_PAD], (uint32_t)__popc(keep));
#pragma unroll
              for (int r = 0; r < 4; ++r) {
                if (keep & (1u << r)) {
                  if (slot < a.capg) a.gkeys[(uint64_t)q * a.capg + slot] = make_key_desc(sv[r], id_of(r));
                  else *a.overflow = 1;
                  ++slot;
                }
              }
            }
          }
        }
      }
      return;
    }
#pragma unroll
    for (int t = 0; t < NQT; ++t) {
      float s[4];
      const uint32_t q = t * QT + qj;
      const float sq_t = s_sq[q], dth_t = s_dth[q], th_t = s_th[q];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float v = (float)acc[t][r] * sx[r] * sq_t;
        if (check_deg && iv[r] >= dth_t) v = FLT_MAX;   // pn*qn <= EPS: reference distance is 0 (best)
        if (!(v == v)) v = FLT_MAX;        // a row (or query) that could not be quantised: the canonical rescoring decides
        s[r] = v;
      }
      if (DENSE) {
        float4 o;
        o.x = (allowed & 1u) ? s[0] : -INFINITY;
        o.y = (allowed & 2u) ? s[1] : -INFINITY;
        o.z = (allowed & 4u) ? s[2] : -INFINITY;
        o.w = (allowed & 8u) ? s[3] : -INFINITY;
        *reinterpret_cast<float4 *>(a.dense + (uint64_t)q * a.dstride + (uint64_t)it * 16 + g * 4) = o;
      } else {
        uint32_t pass = 0;
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (!(s[r] < th_t)) pass |= 1u << r;
        pass &= allowed;
        if (q >= a.nq) pass = 0;
        if (pass) {
          uint32_t slot = atomicAdd(&a.gcnt[q * CNT_PAD], (uint32_t)__popc(pass));
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            if (pass & (1u << r)) {
              if (slot < a.capg) a.gkeys[(uint64_t)q * a.capg + slot] = make_key_desc(s[r], id_of(r));
              else *a.overflow = 1;
              ++slot;
            }
          }
        }
      }
    }
  };

  i32x4 xa[GS][RT], xb[GS][RT];
  i32x4 acc[RT][NQT];
#pragma unroll
  for (int r = 0; r < RT; ++r)
#pragma unroll
    for (int t = 0; t < NQT; ++t) acc[r][t] = (i32x4){0, 0, 0, 0};

  uint32_t it_load = it0, sub_load = 0;   // first item of the group being loaded, next stage of it
  uint32_t it_cmp = it0, sub_cmp = 0;
  uint32_t tl[RT], tc[RT];
#pragma unroll
  for (int r = 0; r < RT; ++r) tl[r] = tc[r] = tile_of(it0 + r);
  // the rows' scales travel with the first stage of their tile group, into the scale registers of the SAME buffer (sca with
  // xa, scb with xb: a buffer is never loaded again before it was computed, whatever the number of stages per group); the
  // group's first compute stage moves them to scc, where the epilogue finds them
  float4 sca[RT], scb[RT], scc[RT];
#pragma unroll
  for (int r = 0; r < RT; ++r) sca[r] = scb[r] = scc[r] = make_float4(0.f, 0.f, 0.f, 0.f);

  // FILT: where this lane's row of each item being loaded starts (its piece of block 0); an item's entries are read when its
  // first stage is requested (the rows' ids for the loads, the four rows of the lane's D fragment for their scales)
  const uint32_t piece_of_lane = MSI_TILE_PIECE(lane & 15u, lane >> 4);   // (where this lane's piece of a KiB block lives)
  const i32x4 *rowp[FILT ? RT : 1];
  if constexpr (FILT) {
#pragma unroll
    for (int r = 0; r < RT; ++r) rowp[r] = a.tiles8;
  }
  auto load_group = [&](i32x4(&x)[GS][RT], float4(&scn)[RT]) {
    if (sub_load == 0) {
#pragma unroll
      for (int r = 0; r < RT; ++r) {
        if constexpr (FILT) {
          const uint32_t *ent = a.rows + (uint64_t)tl[r] * 16;
          const uint32_t row = frow_id(ent[lane & 15u]);
          rowp[r] = a.tiles8 + (uint64_t)(row >> 4) * KB8 * 64 + MSI_TILE_PIECE(row & 15u, lane >> 4);
          const uint4 e4 = *reinterpret_cast<const uint4 *>(ent + g * 4);
          // (a row that is not allowed is loaded all the same — it shares its tile's sectors — but nobody reads its score)
          scn[r] = make_float4(a.scale8[frow_safe(e4.x)], a.scale8[frow_safe(e4.y)], a.scale8[frow_safe(e4.z)], a.scale8[frow_safe(e4.w)]);
        } else {
          scn[r] = *reinterpret_cast<const float4 *>(a.scale8 + (uint64_t)tl[r] * 16 + g * 4);
        }
      }
    }
#pragma unroll
    for (int r = 0; r < RT; ++r) {
      const i32x4 *p;
      if constexpr (FILT) p = rowp[r] + (uint64_t)sub_load * GS * 64;
      else p = a.tiles8 + ((uint64_t)tl[r] * KB8 + (uint64_t)sub_load * GS) * 64 + piece_of_lane;
#pragma unroll
      for (int u = 0; u < GS; ++u) x[u][r] = vs_stream_load_i8(p + u * 64);
    }
    if (++sub_load == GPT) {
      sub_load = 0;
      it_load += RT;
      if (it_load < it1) {
#pragma unroll
        for (int r = 0; r < RT; ++r) tl[r] = tile_of(it_load + r);
      }
    }
  };
  auto compute_group = [&](const i32x4(&x)[GS][RT], const float4(&scn)[RT]) {
    const i32x4 *qb = qf + (size_t)sub_cmp * GS * 64 + lane;
#pragma unroll
    for (int u = 0; u < GS; ++u) {
#pragma unroll
      for (int t = 0; t < NQT; ++t) {
        const i32x4 b = qb[((size_t)t * KB8 + u) * 64];
#pragma unroll
        for (int r = 0; r < RT; ++r) acc[r][t] = __builtin_amdgcn_mfma_i32_16x16x64_i8(x[u][r], b, acc[r][t], 0, 0, 0);
      }
      // (the query fragments of ONE block in flight at a time: hoisting every block's LDS reads to the top of the group
      // costs 16 registers per query tile and block, and the accumulators need them)
      __builtin_amdgcn_sched_barrier(0);
    }
    if (sub_cmp == 0) {
#pragma unroll
      for (int r = 0; r < RT; ++r) scc[r] = scn[r];   // (this group's scales: requested with its first stage, one stage ago)
    }
    if (++sub_cmp == GPT) {
#pragma unroll
      for (int r = 0; r < RT; ++r) {
        if (it_cmp + r < it1) epilogue(tc[r], it_cmp + r, acc[r], scc[r]);
#pragma unroll
        for (int t = 0; t < NQT; ++t) acc[r][t] = (i32x4){0, 0, 0, 0};
      }
      sub_cmp = 0;
      it_cmp += RT;
      if (it_cmp < it1) {
#pragma unroll
        for (int r = 0; r < RT; ++r) tc[r] = tile_of(it_cmp + r);
      }
    }
  };

  // The wait for a stage's rows comes BEFORE the next stage's loads are issued (`landed`: an empty asm that uses every register
  // of the stage, so the compiler's s_waitcnt sits there with nothing younger in flight).  Left to itself the compiler waits at the
  // first MFMA that reads the stage — after the next stage's loads were issued — and, having lost count of the loads in flight at
  // the epilogue's conditional stores, it waits for ALL of them: the next stage's too, i.e. no overlap of loads and MFMAs at all
  // (measured: 2.06 ms per 128-query sweep of 10 M x 768; rocprofv3 / ISA in profiles/r5_i8_sweep.txt).
  auto landed = [&](i32x4(&x)[GS][RT], float4(&scn)[RT]) {
#if !defined(MSI_HIP_EMULATED)
#pragma unroll
    for (int u = 0; u < GS; ++u)
#pragma unroll
      for (int r = 0; r < RT; ++r) asm volatile("" : "+v"(x[u][r]));
#pragma unroll
    for (int r = 0; r < RT; ++r) asm volatile("" : "+v"(scn[r].x), "+v"(scn[r].y), "+v"(scn[r].z), "+v"(scn[r].w));
#else
    (void)x;
    (void)scn;
#endif
  };
  load_group(xa, sca);
  for (;;) {
    landed(xa, sca);
    bool more = it_load < it1;
    if (more) load_group(xb, scb);
    compute_group(xa, sca);
    if (!more) break;
    landed(xb, scb);
    more = it_load < it1;
    if (more) load_group(xa, sca);
    compute_group(xb, scb);
    if (!more) break;
  }
}

// ---------------------------------------------------------------------- select

// Block-wide bitonic sort (ascending) of n (power of two) keys in LDS.
__device__ void block_bitonic_sort(u64 *buf, uint32_t n) {
  for (uint32_t k = 2; k <= n; k <<= 1) {
    for (uint32_t j = k >> 1; j > 0; j >>= 1) {
      for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) {
        uint32_t p = i ^ j;
        if (p > i) {
          bool asc = (i & k) == 0;
          u64 x = buf[i], y = buf[p];
          if ((x > y) == asc) {
            buf[i] = y;
            buf[p] = x;
          }
        }
      }
      __syncthreads();
    }
  }
}

__device__ __forceinline__ uint32_t next_pow2(uint32_t v) {
  uint32_t p = 1;
  while (p < v) p <<= 1;
  return p;
}

// Where vs_select reads its candidates from: a list of keys (sparse scan, the
// exhaustive path) or a dense score matrix row (dense scan) whose entry i belongs
// to row tile_of(i/16)*16 + i%16; -inf entries (disallowed rows) become the
// sentinel key ~0, which sorts last and is dropped by the callers.
struct KeySrc {
  const u64 *keys;
  const float *dense;
  const uint32_t *rows;   // nullable: the items of a filtered sweep (vs_filter_rows_kernel)
  uint32_t stride;
  uint32_t base = 0;   // dense: entry i of this source is entry base + i of the row (vs_select_part_kernel's slices)
  __device__ __forceinline__ u64 get(uint32_t i) const {
    if (keys) return keys[i];
    i += base;
    const float s = dense[i];
    if (s == -INFINITY) return ~0ull;
    const uint32_t item = (i >> 4) * stride;
    return make_key_desc(s, rows ? frow_id(rows[(uint64_t)item * 16 + (i & 15)]) : item * 16 + (i & 15));
  }
};

// Leaves the min(c,K) smallest keys of src[0..c), ascending, in sbuf[0..) and
// returns their count.  K <= KP_MAX, sbuf has SEL_SORTCAP entries, hist 2048.
__device__ uint32_t block_select_smallest(const KeySrc &src, uint32_t c, uint32_t K, u64 *sbuf,
                                          uint32_t *hist, uint32_t *sh) {
  const uint32_t tid = threadIdx.x;
  if (c <= SEL_SORTCAP) {
    uint32_t n = next_pow2(c < 2 ? 2 : c);
    for (uint32_t i = tid; i < n; i += blockDim.x) sbuf[i] = i < c ? src.get(i) : ~0ull;
    __syncthreads();
    block_bitonic_sort(sbuf, n);
    return c < K ? c : K;
  }
  // radix select on 11-bit digits, most significant first
  u64 prefix = 0;          // decided high bits (right-aligned)
  uint32_t pbits = 0;
  uint32_t krem = K;       // rank still to find inside the current prefix
  for (uint32_t level = 0; level < 6; ++level) {
    const uint32_t dbits = level < 5 ? 11 : 9;
    const uint32_t shift = 64 - pbits - dbits;
    for (uint32_t i = tid; i < 2048; i += blockDim.x) hist[i] = 0;
    __syncthreads();
    // (eight independent loads in flight per thread: with one load per iteration a pass over the ~200 k sampled scores of a
    // query was a chain of 790 memory round trips per thread — most of this kernel's 0.3-0.7 ms, profiles/r5_i8_vector_leg_*)
    for (uint32_t i0 = tid; i0 < c; i0 += SEL_UNROLL * blockDim.x) {
      u64 kk[SEL_UNROLL];
#pragma unroll
      for (int u = 0; u < SEL_UNROLL; ++u) {
        const uint32_t i = i0 + u * blockDim.x;
        kk[u] = i < c ? src.get(i) : 0ull;
      }
#pragma unroll
      for (int u = 0; u < SEL_UNROLL; ++u) {
        if (i0 + u * blockDim.x < c && (pbits == 0 || (kk[u] >> (64 - pbits)) == prefix))
          atomicAdd(&hist[(uint32_t)(kk[u] >> shift) & ((1u << dbits) - 1u)], 1u);
      }
    }
    __syncthreads();
    {
      // the bin that holds the krem-th smallest key: every thread sums 8 bins, the 256 sums are scanned through LDS
      // (eight doubling steps), and the one thread whose bins straddle krem walks them.  (One thread walking all
      // 2 048 bins was most of this kernel's time: ~50 us per level.)
      const uint32_t nb = 1u << dbits;
      uint32_t loc[8], mine = 0;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const uint32_t b = tid * 8 + e;
        loc[e] = b < nb ? hist[b] : 0u;
        mine += loc[e];
      }
      static_assert(SEL_THREADS * 8 == 2048 && SEL_SORTCAP * 2 >= 2 * SEL_THREADS,
                    "the bin scan below is laid out for 256 threads x 8 bins and two rows of 256 words aliased onto sbuf");
      uint32_t *scan = reinterpret_cast<uint32_t *>(sbuf);   // (sbuf is not in use yet: 2 x 256 words of it)
      scan[tid] = mine;
      __syncthreads();
      uint32_t src_off = 0;
      for (uint32_t d = 1; d < 256; d <<= 1) {
        const uint32_t v = scan[src_off + tid] + (tid >= d ? scan[src_off + tid - d] : 0u);
        scan[(src_off ^ 256) + tid] = v;
        src_off ^= 256;
        __syncthreads();
      }
      const uint32_t incl = scan[src_off + tid], excl = incl - mine;
      const uint32_t total = scan[src_off + 255];
      if (tid == 0) sh[0] = 0xFFFFFFFFu;
      __syncthreads();
      if (excl < krem && krem <= incl) {   // exactly one thread (krem >= 1)
        uint32_t cum = excl;
        int e = 0;
        for (; e < 7; ++e) {
          if (cum + loc[e] >= krem) break;
          cum += loc[e];
        }
        sh[0] = tid * 8 + e;
        sh[1] = cum;       // entries of this prefix strictly below bin b
        sh[2] = loc[e];
      }
      __syncthreads();
      if (sh[0] == 0xFFFFFFFFu && tid == 0) {   // krem above the total (c < K cannot happen here: c > SEL_SORTCAP >= K)
        sh[0] = nb - 1;
        sh[1] = total;
        sh[2] = hist[nb - 1];
      }
      __syncthreads();
    }
    const uint32_t b = sh[0], below = sh[1], inbin = sh[2];
    const u64 newprefix = (prefix << dbits) | b;
    const uint32_t gather = (K - krem) + below + inbin;  // keys with top bits <= newprefix
    __syncthreads();
    if (gather <= SEL_SORTCAP || level == 5) {
      if (tid == 0) sh[3] = 0;
      __syncthreads();
      for (uint32_t i0 = tid; i0 < c; i0 += SEL_UNROLL * blockDim.x) {
        u64 kk[SEL_UNROLL];
#pragma unroll
        for (int u = 0; u < SEL_UNROLL; ++u) {
          const uint32_t i = i0 + u * blockDim.x;
          kk[u] = i < c ? src.get(i) : ~0ull;
        }
#pragma unroll
        for (int u = 0; u < SEL_UNROLL; ++u) {
          if (i0 + u * blockDim.x < c && (kk[u] >> shift) <= newprefix) {
            uint32_t slot = atomicAdd(&sh[3], 1u);
            if (slot < SEL_SORTCAP) sbuf[slot] = kk[u];
          }
        }
      }
      __syncthreads();
      uint32_t got = sh[3] < SEL_SORTCAP ? sh[3] : SEL_SORTCAP;
      uint32_t n = next_pow2(got < 2 ? 2 : got);
      for (uint32_t i = got + tid; i < n; i += blockDim.x) sbuf[i] = ~0ull;
      __syncthreads();
      block_bitonic_sort(sbuf, n);
      return got < K ? got : K;
    }
    prefix = newprefix;
    pbits += dbits;
    krem -= below;
  }
  return 0;  // unreachable
}

// Number of leading non-sentinel keys among sbuf[0..got) (sorted ascending).
__device__ uint32_t block_count_valid(const u64 *sbuf, uint32_t got, uint32_t *sh_valid) {
  if (threadIdx.x == 0) *sh_valid = 0;
  __syncthreads();
  for (uint32_t i = threadIdx.x; i < got; i += blockDim.x)
    if (sbuf[i] != ~0ull) atomicAdd(sh_valid, 1u);
  __syncthreads();
  return *sh_valid;
}

struct SelectArgs {
  // input: sparse lists ...
  const u64 *gkeys;            // [NQ_MAX][capg]
  const uint32_t *gcnt;        // [NQ_MAX][CNT_PAD]
  uint32_t capg;
  // ... or the dense score matrix of a dense scan
  const float *dense;          // [NQ_MAX][dstride]; null = sparse input
  uint32_t dstride;
  const uint32_t *n_items_ptr;
  const uint32_t *rows;
  uint32_t stride;
  // output
  uint32_t K;                  // mode 0: the threshold rank r, mode 1: K'
  u64 *sel_keys;               // [NQ_MAX][KP_MAX]   (mode 1)
  uint32_t *sel_cnt;           // [NQ_MAX]           (mode 1)
  float *theta;                // [NQ_MAX]           (mode 0: threshold for the sparse pass)
  int mode;                    // 0 = threshold only, 1 = keep the keys
  uint32_t fixed_c;            // sparse input: every query has this many keys (0: gcnt says) — the lists vs_select_part_kernel left
};

__global__ __launch_bounds__(SEL_THREADS) void vs_select_kernel(SelectArgs a) {
  __shared__ u64 sbuf[SEL_SORTCAP];
  __shared__ uint32_t hist[2048];
  __shared__ uint32_t sh[4];
  __shared__ uint32_t sh_valid;
  const uint32_t j = blockIdx.x;
  KeySrc src;
  uint32_t c;
  if (a.dense) {
    src.keys = nullptr;
    src.dense = a.dense + (uint64_t)j * a.dstride;
    src.rows = a.rows;
    src.stride = a.stride;
    const uint32_t n_all = *a.n_items_ptr;
    c = ((n_all + a.stride - 1) / a.stride) * 16;
  } else {
    src.keys = a.gkeys + (uint64_t)j * a.capg;
    src.dense = nullptr;
    src.rows = nullptr;
    src.stride = 1;
    c = a.fixed_c ? a.fixed_c : a.gcnt[j * CNT_PAD];
    if (c > a.capg) c = a.capg;
  }
  const uint32_t got = block_select_smallest(src, c, a.K, sbuf, hist, sh);
  __syncthreads();
  const uint32_t valid = block_count_valid(sbuf, got, &sh_valid);
  if (a.mode == 0) {
    if (threadIdx.x == 0) a.theta[j] = valid >= a.K ? key_desc_score(sbuf[a.K - 1]) : -INFINITY;
  } else {
    for (uint32_t i = threadIdx.x; i < valid; i += blockDim.x) a.sel_keys[(uint64_t)j * KP_MAX + i] = sbuf[i];
    if (threadIdx.x == 0) a.sel_cnt[j] = valid;
  }
}

// The threshold of the sparse pass sits in front of every sweep, and one workgroup per query walking its ~200 k sampled scores
// (10 M rows, K' = 1 024) two or three times was 0.3-0.75 ms of it with half the CUs idle (profiles/r5_i8_vector_leg_kernel_
// stats.csv: vs_select_kernel).  The K smallest keys of a row are among the K smallest of each of its slices: gridDim.x
// workgroups per query select inside one slice each and leave their K best in sel_keys[j][slice][K] (free until the chunk's
// second selection); vs_select_kernel then reads those gridDim.x * K keys (SelectArgs::fixed_c) instead of the row.
__global__ __launch_bounds__(SEL_THREADS) void vs_select_part_kernel(SelectArgs a) {
  __shared__ u64 sbuf[SEL_SORTCAP];
  __shared__ uint32_t hist[2048];
  __shared__ uint32_t sh[4];
  const uint32_t p = blockIdx.x, P = gridDim.x, j = blockIdx.y;
  const uint32_t n_all = *a.n_items_ptr;
  const uint32_t c = ((n_all + a.stride - 1) / a.stride) * 16;
  const uint32_t lo = (uint32_t)((uint64_t)c * p / P), hi = (uint32_t)((uint64_t)c * (p + 1) / P);
  KeySrc src;
  src.keys = nullptr;
  src.dense = a.dense + (uint64_t)j * a.dstride;
  src.rows = a.rows;
  src.stride = a.stride;
  src.base = lo;
  const uint32_t got = block_select_smallest(src, hi - lo, a.K, sbuf, hist, sh);
  __syncthreads();
  u64 *out = a.sel_keys + (uint64_t)j * KP_MAX + (uint64_t)p * a.K;
  for (uint32_t i = threadIdx.x; i < a.K; i += blockDim.x) out[i] = i < got ? sbuf[i] : ~0ull;
}

// ---- the sparse pass's threshold from two histogram passes (round 6) --------------------------------------------------
// The threshold only has to be CONSERVATIVE: any value at or below the r-th best sampled score keeps the guarantee the
// rank r was chosen for (a lower threshold can only add survivors).  So nothing is selected or sorted: the sampled scores
// of a query are counted by the top 11 bits of their descending key (workgroups over slices of the row, LDS histograms
// merged into a global one), the bin that holds rank r is picked, the scores of THAT bin are counted by the next 11 bits,
// and the threshold is the smallest score of the 22-bit bucket that holds rank r — at most 2^-13 relative below the exact
// one.  Four short launches that use the whole chip instead of one workgroup per query radix-selecting and sorting
// (vs_select_kernel: 0.3-0.75 ms per 128-query chunk at C4 in round 5; in slices, with a sort per slice: 0.15-0.2 ms).
struct ThrArgs {
  const float *dense;            // [NQ_MAX][dstride]
  uint32_t dstride;
  const uint32_t *n_items_ptr;
  uint32_t stride;
  uint32_t rank;                 // r
  uint32_t *hist;                // [2][NQ_MAX][2048]
  uint32_t *pick;                // [NQ_MAX][2]: the level-0 bin, the rank still to find inside it
  float *theta;                  // [NQ_MAX]
};
__device__ __forceinline__ uint32_t thr_inv(float s) { return ~f32_to_ord(s); }   // ascending = best score first (make_key_desc)

template <int LEVEL>
__global__ __launch_bounds__(256) void vs_thr_hist_kernel(ThrArgs a) {
  __shared__ uint32_t h[2048];
  const uint32_t p = blockIdx.x, P = gridDim.x, j = blockIdx.y, tid = threadIdx.x;
  for (uint32_t i = tid; i < 2048; i += 256) h[i] = 0;
  const uint32_t n_all = *a.n_items_ptr;
  const uint32_t c = ((n_all + a.stride - 1) / a.stride) * 16;
  const uint32_t lo = (uint32_t)((uint64_t)c * p / P), hi = (uint32_t)((uint64_t)c * (p + 1) / P);
  const uint32_t b0 = LEVEL ? a.pick[j * 2] : 0u;
  __syncthreads();
  if (LEVEL && b0 == 0xFFFFFFFFu) return;   // fewer than r scores: the threshold is -inf already
  const float *row = a.dense + (uint64_t)j * a.dstride;
  for (uint32_t i0 = lo + tid; i0 < hi; i0 += 8 * 256) {
    float v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = i0 + u * 256 < hi ? row[i0 + u * 256] : -INFINITY;
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      if (v[u] == -INFINITY) continue;   // a disallowed row (or the tail)
      const uint32_t k = thr_inv(v[u]);
      if (LEVEL == 0) atomicAdd(&h[k >> 21], 1u);
      else if ((k >> 21) == b0) atomicAdd(&h[(k >> 10) & 2047u], 1u);
    }
  }
  __syncthreads();
  uint32_t *g = a.hist + ((size_t)LEVEL * NQ_MAX + j) * 2048;
  for (uint32_t i = tid; i < 2048; i += 256)
    if (h[i]) atomicAdd(&g[i], h[i]);
}

// one workgroup per query: the bin of `hist` that holds rank r (LEVEL 0: -> pick) / the threshold (LEVEL 1)
template <int LEVEL>
__global__ __launch_bounds__(256) void vs_thr_pick_kernel(ThrArgs a) {
  __shared__ uint32_t scan[512];
  __shared__ uint32_t found[2];
  const uint32_t j = blockIdx.x, tid = threadIdx.x;
  const uint32_t *g = a.hist + ((size_t)LEVEL * NQ_MAX + j) * 2048;
  const uint32_t b0 = LEVEL ? a.pick[j * 2] : 0u;
  const uint32_t r = LEVEL ? a.pick[j * 2 + 1] : a.rank;
  if (LEVEL && b0 == 0xFFFFFFFFu) {
    if (tid == 0) a.theta[j] = -INFINITY;
    return;
  }
  uint32_t loc[8], mine = 0;
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    loc[e] = g[tid * 8 + e];
    mine += loc[e];
  }
  scan[tid] = mine;
  if (tid == 0) found[0] = 0xFFFFFFFFu;
  __syncthreads();
  uint32_t src = 0;
  for (uint32_t d = 1; d < 256; d <<= 1) {
    const uint32_t v = scan[src + tid] + (tid >= d ? scan[src + tid - d] : 0u);
    scan[(src ^ 256) + tid] = v;
    src ^= 256;
    __syncthreads();
  }
  const uint32_t incl = scan[src + tid], excl = incl - mine;
  if (r >= 1 && excl < r && r <= incl) {   // exactly one thread
    uint32_t cum = excl;
    int e = 0;
    for (; e < 7; ++e) {
      if (cum + loc[e] >= r) break;
      cum += loc[e];
    }
    found[0] = tid * 8 + e;
    found[1] = r - cum;   // rank inside the bin (>= 1)
  }
  __syncthreads();
  if (tid == 0) {
    if (LEVEL == 0) {
      a.pick[j * 2] = found[0];   // 0xFFFFFFFF: fewer than r scores in the sample
      a.pick[j * 2 + 1] = found[0] == 0xFFFFFFFFu ? 0u : found[1];
    } else {
      // the smallest score of the 22-bit bucket: every score of the bucket (the r-th best among them) passes !(s < theta)
      const uint32_t inv = found[0] == 0xFFFFFFFFu ? 0xFFFFFFFFu : ((b0 << 21) | (found[0] << 10) | 0x3FFu);
      a.theta[j] = found[0] == 0xFFFFFFFFu ? -INFINITY : ord_to_f32(~inv);
    }
  }
}

// --------------------------------------------------------------------- rescore

// Reference arithmetic for one (row, query) pair from the tiled layout:
// sequential f32 mul+add in column order, then arroy/hannoy's cosine distance.
template <bool S16>
__device__ __forceinline__ float canonical_dot_t(const void *__restrict__ tiles, uint32_t KB, uint32_t row,
                                                 const float *__restrict__ q, uint32_t dpad) {
  // The additions are one dependent chain (the reference's order), the LOADS are not: a row's 16-byte pieces sit 256 B
  // apart (four per KiB block), so one load per step of the chain was one memory round trip per 4 (8) columns — 192 of
  // them for a row of 768 floats, 0.52 ms per 128-query chunk of C4 in vs_rescore_dots_kernel
  // (profiles/r5_i8_vector_leg_kernel_stats.csv).  CD_UNR blocks = 4 CD_UNR pieces are requested before the first is used.
  constexpr int CD_UNR = 8;
  float acc = 0.f;
  if (S16) {
    const bf16x8 *base = reinterpret_cast<const bf16x8 *>(tiles) + (uint64_t)(row >> 4) * KB * 64;
    const uint32_t ri = row & 15;
    for (uint32_t kb0 = 0; kb0 < KB; kb0 += CD_UNR) {
      bf16x8 v[CD_UNR][4];
#pragma unroll
      for (int u = 0; u < CD_UNR; ++u)
#pragma unroll
        for (int g = 0; g < 4; ++g)
          if (kb0 + u < KB) v[u][g] = base[(uint64_t)(kb0 + u) * 64 + MSI_TILE_PIECE(ri, (uint32_t)g)];
#pragma unroll
      for (int u = 0; u < CD_UNR; ++u) {
        if (kb0 + u < KB) {
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const float *qq = q + (kb0 + u) * 32 + g * 8;
#pragma unroll
            for (int e = 0; e < 8; ++e) acc = __fadd_rn(acc, __fmul_rn((float)v[u][g][e], qq[e]));
          }
        }
      }
    }
    return acc;
  }
  const float4 *base = reinterpret_cast<const float4 *>(tiles) + (uint64_t)(row >> 4) * KB * 64;
  const uint32_t ri = row & 15;
  for (uint32_t kb0 = 0; kb0 < KB; kb0 += CD_UNR) {
    float4 v[CD_UNR][4];
#pragma unroll
    for (int u = 0; u < CD_UNR; ++u)
#pragma unroll
      for (int g = 0; g < 4; ++g)
        if (kb0 + u < KB) v[u][g] = base[(uint64_t)(kb0 + u) * 64 + MSI_TILE_PIECE(ri, (uint32_t)g)];
#pragma unroll
    for (int u = 0; u < CD_UNR; ++u) {
      if (kb0 + u < KB) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const float *qq = q + (kb0 + u) * 16 + g * 4;
          acc = __fadd_rn(acc, __fmul_rn(v[u][g].x, qq[0]));
          acc = __fadd_rn(acc, __fmul_rn(v[u][g].y, qq[1]));
          acc = __fadd_rn(acc, __fmul_rn(v[u][g].z, qq[2]));
          acc = __fadd_rn(acc, __fmul_rn(v[u][g].w, qq[3]));
        }
      }
    }
  }
  return acc;
}
__device__ __forceinline__ float canonical_dot(const void *__restrict__ tiles, uint32_t KB, uint32_t row,
                                               const float *__restrict__ q, uint32_t dpad, bool s16) {
  return s16 ? canonical_dot_t<true>(tiles, KB, row, q, dpad) : canonical_dot_t<false>(tiles, KB, row, q, dpad);
}

__device__ __forceinline__ float canonical_distance(float pq, float pn, float qn) {
  const float pnqn = __fmul_rn(pn, qn);
  if (pnqn > FLT_EPSILON) {
    const float c = msi_div_rn(pq, pnqn);
    return __fsub_rn(1.0f, c) * 0.5f;  // /2 is exact
  }
  return 0.0f;
}

struct RescoreArgs {
  const void *tiles;
  uint32_t dpad;
  uint32_t s16;
  const float *norm;
  const uint32_t *docids;
  const float *qrow;       // [NQ_MAX][dpad]
  const float *qn;         // [NQ_MAX]
  const float *inv_qn;     // [NQ_MAX]
  const u64 *sel_keys;     // [NQ_MAX][KP_MAX]
  const uint32_t *sel_cnt; // [NQ_MAX]
  uint32_t KB;
  uint32_t kp;
  uint32_t k;
  float eps;               // bound on |fast cos - reference cos|
  const float *eps_q;      // nullable [NQ_MAX]: added to eps per query (the int8 sweep: the query's own quantisation bound)
  uint32_t *out_docids;    // [nq][k]
  float *out_dist;         // [nq][k]
  uint32_t *out_counts;    // [nq]
  uint32_t *inexact;       // [nq]
  const uint32_t *overflow;
  const float *theta;      // nullable: thresholds the sparse pass ran with
  float *refined;          // nullable [NQ_MAX][KP_MAX]: the candidates' refined cosines (vs_refine_kernel); -inf: cannot matter
};

// Which of the K' selected candidates need the reference arithmetic at all.  They arrive ordered by fast score; at least k
// rows have a reference cosine >= (k-th fast cosine) - eps, so a candidate whose fast cosine lies more than 2 eps below the
// k-th fast cosine cannot be among the k nearest (its reference cosine is below that of k other rows).  With the int8
// sweep's eps (~1.8e-2) and K' = 1 024 that is most of them: on i.i.d. rows ~200 candidates are rescored per query, not 1 024.
// (The exactness proof is unchanged: it speaks about the rows that were NOT selected.)
__device__ __forceinline__ bool candidate_matters(const RescoreArgs &a, uint32_t j, uint32_t i, uint32_t cnt) {
  if (i < a.k || cnt < a.k || a.k == 0) return true;
  const float eps = a.eps + (a.eps_q ? a.eps_q[j] : 0.0f);
  const float inv = a.inv_qn[j];
  const float ck = key_desc_score(a.sel_keys[(uint64_t)j * KP_MAX + a.k - 1]) * inv;
  const float ci = key_desc_score(a.sel_keys[(uint64_t)j * KP_MAX + i]) * inv;
  if (!(fabsf(ck) < 1e30f)) return true;    // k degenerate rows (distance 0 by definition) in front: a row at cosine 1 ties with them
  // Rows keyed FLT_MAX (degenerate, or not quantisable: NaN scale) sit in front and obey no bound on their reference cosine:
  // with m < k of them only k - m rows back the claim above, so nothing is pruned then
  const float c0 = key_desc_score(a.sel_keys[(uint64_t)j * KP_MAX]) * inv;
  if (!(fabsf(c0) < 1e30f)) return true;
  return !(ci < ck - 2.0f * eps - 1e-6f);   // (1e-6: a strictly larger distance, never a tie that the docid would decide; NaN: keep)
}

// ---- second opinion on the candidates (round 6) ---------------------------------------------------------------------
// The int8 sweep's bound is wide (eps ~ 1.8e-2 in cosine): at 10 M rows EVERY one of the K' = 1 024 selected candidates
// lies within 2 eps of the 20th, so round 5 ran the reference's sequential dot product — one dependent chain of 768
// additions over a row whose 16-byte pieces lie 256 B apart — for all 131 072 (candidate, query) pairs of a chunk: 0.51 ms of
// a 2.6 ms chunk, whatever the access pattern (one thread per row: 64 tiles per wave-load; 16 lanes per row with the
// products in LDS: too few chains in flight — both measured, profiles/r6_vector_leg_c4_kernel_stats*.csv).
// Now the candidates first get a PARALLEL f32 dot product (16 lanes share a row: its pieces side by side, partial sums,
// a shuffle tree): the same rounded products as the reference adds, in another order, so
//     |refined cos - reference cos| <= eps32 = (2 dpad + 64) u + 1e-6          (two orders of summing the same terms)
// and the reference arithmetic is only run for the candidates whose refined cosine lies within 2 eps32 of the k-th best
// refined cosine (k of them have a reference cosine >= that - eps32; anything below by more than 2 eps32 + 1e-6 has a
// strictly larger distance than k candidates) — a few dozen per query instead of 1 024.  Degenerate rows (distance 0 by the
// reference's pn*qn <= EPS rule) and NaNs always go to the reference arithmetic.  Every returned distance is still the
// reference's; the proof about UNSELECTED rows (vs_rescore_kernel's epilogue) is untouched.
constexpr uint32_t RD_WGS_PER_QUERY = 8;
__device__ __forceinline__ float refine_eps(uint32_t dpad) { return (2.0f * (float)dpad + 64.0f) * 5.9604645e-8f * 1.01f + 1e-6f; }

__global__ __launch_bounds__(SEL_THREADS) void vs_refine_kernel(RescoreArgs a) {
  MSI_DYNAMIC_LDS(dyn);
  const uint32_t dpad = a.dpad;
  float *qs = reinterpret_cast<float *>(dyn);          // [dpad]
  __shared__ uint32_t s_last;
  const uint32_t j = blockIdx.y, tid = threadIdx.x;
  const uint32_t cnt = a.sel_cnt[j];
  if (tid == 0) s_last = 0;
  for (uint32_t i = tid; i < dpad; i += blockDim.x) qs[i] = a.qrow[(uint64_t)j * dpad + i];
  __syncthreads();
  {   // the candidates that matter by the SWEEP's bound (ordered by fast score: a prefix); the others can never matter
    uint32_t last = 0;
    for (uint32_t i = tid; i < cnt; i += blockDim.x) {
      if (candidate_matters(a, j, i, cnt)) last = i + 1;
      else if (blockIdx.x == 0) a.refined[(uint64_t)j * KP_MAX + i] = -INFINITY;
    }
    if (last) atomicMax(&s_last, last);
  }
  __syncthreads();
  const uint32_t m = s_last;
  const uint32_t grp = tid >> 4, l16 = tid & 15;
  const float qn = a.qn[j];
  const bool s16 = a.s16 != 0;
  const uint32_t pieces = a.KB * 4;                    // 16-byte pieces of a row
  for (uint32_t i0 = blockIdx.x * 16; i0 < m; i0 += gridDim.x * 16) {   // (uniform trip count: the shuffles below see whole waves)
    const uint32_t i = i0 + grp;
    const bool active = i < m && candidate_matters(a, j, i, cnt);
    const uint32_t row = active ? (uint32_t)a.sel_keys[(uint64_t)j * KP_MAX + i] : 0u;
    float part = 0.f;
    if (!active) {
    } else if (s16) {
      const bf16x8 *base = reinterpret_cast<const bf16x8 *>(a.tiles) + (uint64_t)(row >> 4) * a.KB * 64;
      for (uint32_t p0 = l16; p0 < pieces; p0 += 16 * 4) {
        bf16x8 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const uint32_t p = p0 + u * 16;
          if (p < pieces) v[u] = base[(uint64_t)(p >> 2) * 64 + MSI_TILE_PIECE(row & 15u, p & 3u)];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const uint32_t p = p0 + u * 16;
          if (p >= pieces) continue;
          const uint32_t col = (p >> 2) * 32 + (p & 3) * 8;
#pragma unroll
          for (int e = 0; e < 8; ++e) part = __fadd_rn(part, __fmul_rn((float)v[u][e], qs[col + e]));
        }
      }
    } else {
      const float4 *base = reinterpret_cast<const float4 *>(a.tiles) + (uint64_t)(row >> 4) * a.KB * 64;
      for (uint32_t p0 = l16; p0 < pieces; p0 += 16 * 6) {
        float4 v[6];
#pragma unroll
        for (int u = 0; u < 6; ++u) {
          const uint32_t p = p0 + u * 16;
          if (p < pieces) v[u] = base[(uint64_t)(p >> 2) * 64 + MSI_TILE_PIECE(row & 15u, p & 3u)];
        }
#pragma unroll
        for (int u = 0; u < 6; ++u) {
          const uint32_t p = p0 + u * 16;
          if (p >= pieces) continue;
          const uint32_t col = (p >> 2) * 16 + (p & 3) * 4;
          part = __fadd_rn(part, __fmul_rn(v[u].x, qs[col]));
          part = __fadd_rn(part, __fmul_rn(v[u].y, qs[col + 1]));
          part = __fadd_rn(part, __fmul_rn(v[u].z, qs[col + 2]));
          part = __fadd_rn(part, __fmul_rn(v[u].w, qs[col + 3]));
        }
      }
    }
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) part = __fadd_rn(part, __shfl_xor(part, o));   // (o < 16: inside the group)
    if (active && l16 == 0) {
      const float pnqn = __fmul_rn(a.norm[row], qn);
      float c = INFINITY;                                // degenerate (distance 0 by definition): always to the reference arithmetic
      if (pnqn > FLT_EPSILON) c = msi_div_rn(part, pnqn);
      if (!(c == c)) c = INFINITY;                       // NaN: likewise
      a.refined[(uint64_t)j * KP_MAX + i] = c;
    }
  }
}

__global__ __launch_bounds__(SEL_THREADS) void vs_rescore_kernel(RescoreArgs a) {
  MSI_DYNAMIC_LDS(dyn);
  u64 *sbuf = reinterpret_cast<u64 *>(dyn);                              // [KP_MAX]: candidates by refined cosine
  u64 *sbuf2 = sbuf + KP_MAX;                                            // [KP_MAX]: the rescored ones by reference distance
  float *qs = reinterpret_cast<float *>(dyn + 2 * KP_MAX * sizeof(u64)); // [dpad]
  __shared__ uint32_t s_last;
  const uint32_t j = blockIdx.x;
  const uint32_t dpad = a.dpad;
  for (uint32_t i = threadIdx.x; i < dpad; i += blockDim.x) qs[i] = a.qrow[(uint64_t)j * dpad + i];
  const uint32_t cnt = a.sel_cnt[j];
  if (threadIdx.x == 0) s_last = 0;
  // (1) the candidates ordered by refined cosine, best first (ties: by position = by fast score)
  const uint32_t n1 = next_pow2(cnt < 2 ? 2 : cnt);
  for (uint32_t i = threadIdx.x; i < n1; i += blockDim.x) {
    u64 key = ~0ull;
    if (i < cnt) {
      float c = a.refined ? a.refined[(uint64_t)j * KP_MAX + i] : INFINITY;   // (no refinement: everything is rescored)
      if (c == INFINITY) c = FLT_MAX;
      key = make_key_desc(c, i);
    }
    sbuf[i] = key;
  }
  __syncthreads();
  block_bitonic_sort(sbuf, n1);
  // (2) which of them need the reference arithmetic: a prefix of that order
  {
    const float eps2 = 2.0f * refine_eps(dpad) + 1e-6f;
    float tau = cnt >= a.k && a.k > 0 ? key_desc_score(sbuf[a.k - 1]) : -INFINITY;
    // a degenerate / NaN candidate in front obeys no bound on its reference cosine: then nothing is pruned
    if (cnt && !(fabsf(key_desc_score(sbuf[0])) < 1e30f)) tau = -INFINITY;
    uint32_t last = 0;
    for (uint32_t p = threadIdx.x; p < cnt; p += blockDim.x) {
      const float c = key_desc_score(sbuf[p]);
      if (c == -INFINITY) continue;                        // outside the sweep's own window (vs_refine_kernel)
      if (p < a.k || tau == -INFINITY || !(c < tau - eps2)) last = p + 1;
    }
    if (last) atomicMax(&s_last, last);
  }
  __syncthreads();
  const uint32_t m = s_last;
  const uint32_t n = next_pow2(m < 2 ? 2 : m);
  const float qn = a.qn[j];
  for (uint32_t p = threadIdx.x; p < n; p += blockDim.x) {
    u64 key = ~0ull;
    if (p < m && key_desc_score(sbuf[p]) != -INFINITY) {
      const uint32_t i = (uint32_t)sbuf[p];
      const uint32_t row = (uint32_t)a.sel_keys[(uint64_t)j * KP_MAX + i];
      const float pq = canonical_dot(a.tiles, a.KB, row, qs, dpad, a.s16 != 0);
      const float d = canonical_distance(pq, a.norm[row], qn);
      key = ((u64)f32_to_ord(d) << 32) | row;  // rows ascend with docids
    }
    sbuf2[p] = key;
  }
  __syncthreads();
  block_bitonic_sort(sbuf2, n);
  sbuf = sbuf2;
  const uint32_t out_n = cnt < a.k ? cnt : a.k;
  for (uint32_t i = threadIdx.x; i < a.k; i += blockDim.x) {
    uint32_t id = 0xFFFFFFFFu;
    float d = INFINITY;
    if (i < out_n) {
      const u64 key = sbuf[i];
      id = a.docids[(uint32_t)key];
      d = ord_to_f32((uint32_t)(key >> 32));
    }
    a.out_docids[(uint64_t)j * a.k + i] = id;
    a.out_dist[(uint64_t)j * a.k + i] = d;
  }
  if (threadIdx.x == 0) {
    a.out_counts[j] = out_n;
    // Exactness proof.  Unselected rows have fast cos <= cmin, hence reference
    // cos <= cmin + eps, hence reference distance >= (1 - cmin - eps)/2 - 2e-7.
    uint32_t bad = *a.overflow ? 1u : 0u;
    // a thresholded pass that kept fewer than K' rows may have cut real neighbours
    if (a.theta && a.theta[j] > -INFINITY && cnt < a.kp) bad = 1;
    if (cnt == a.kp && out_n > 0) {
      const float smin = key_desc_score(a.sel_keys[(uint64_t)j * KP_MAX + cnt - 1]);
      const float cmin = smin * a.inv_qn[j];
      const float eps = a.eps + (a.eps_q ? a.eps_q[j] : 0.0f);   // (+inf: a query that could not be quantised proves nothing)
      const float bound = (1.0f - cmin - eps) * 0.5f - 2e-7f;
      const float dk = ord_to_f32((uint32_t)(sbuf[out_n - 1] >> 32));
      if (!(dk < bound)) bad = 1;
    }
    if (a.inexact) a.inexact[j] = bad;
  }
}

// ------------------------------------------------------------------ exhaustive

// Reference distance of EVERY allowed row for one query (fallback when the
// exactness proof fails, e.g. more than K' rows tie at the cut).
__global__ void vs_exhaustive_kernel(const void *__restrict__ tiles, const float *__restrict__ norm,
                                     const uint32_t *__restrict__ docids, uint64_t n_rows, uint32_t KB,
                                     const float *__restrict__ qrow, const float *__restrict__ qn_p,
                                     uint32_t qj, const u64 *__restrict__ fbits, uint64_t nbits,
                                     u64 *__restrict__ keys, uint32_t dpad, uint32_t s16) {
  MSI_DYNAMIC_LDS(dyn);
  float *qs = reinterpret_cast<float *>(dyn);
  for (uint32_t i = threadIdx.x; i < dpad; i += blockDim.x) qs[i] = qrow[(uint64_t)qj * dpad + i];
  __syncthreads();
  uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= n_rows) return;
  bool ok = true;
  if (fbits) {
    uint32_t id = docids[r];
    ok = (uint64_t)id < nbits && ((fbits[id >> 6] >> (id & 63)) & 1ull);
  }
  u64 key = ~0ull;
  if (ok) {
    const float pq = canonical_dot(tiles, KB, (uint32_t)r, qs, dpad, s16 != 0);
    const float d = canonical_distance(pq, norm[r], qn_p[qj]);
    key = ((u64)f32_to_ord(d) << 32) | (uint32_t)r;
  }
  keys[r] = key;
}

__global__ __launch_bounds__(SEL_THREADS) void vs_exhaustive_select_kernel(
    const u64 *__restrict__ keys, uint32_t c, uint32_t k, const uint32_t *__restrict__ docids,
    uint32_t *__restrict__ out_docids, float *__restrict__ out_dist, uint32_t *__restrict__ out_count) {
  __shared__ u64 sbuf[SEL_SORTCAP];
  __shared__ uint32_t hist[2048];
  __shared__ uint32_t sh[4];
  __shared__ uint32_t sh_valid;
  KeySrc src;
  src.keys = keys;
  src.dense = nullptr;
  src.rows = nullptr;
  src.stride = 1;
  const uint32_t got = block_select_smallest(src, c, k, sbuf, hist, sh);
  __syncthreads();
  // disallowed rows carry key ~0: drop them
  const uint32_t out_n = block_count_valid(sbuf, got, &sh_valid);
  for (uint32_t i = threadIdx.x; i < k; i += blockDim.x) {
    uint32_t id = 0xFFFFFFFFu;
    float d = INFINITY;
    if (i < out_n) {
      id = docids[(uint32_t)sbuf[i]];
      d = ord_to_f32((uint32_t)(sbuf[i] >> 32));
    }
    out_docids[i] = id;
    out_dist[i] = d;
  }
  if (threadIdx.x == 0) *out_count = out_n;
}

__global__ void vs_gather_row_kernel(const void *__restrict__ tiles, uint32_t KB, uint32_t row,
                                     uint32_t dim, float *__restrict__ out, uint32_t s16) {
  uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= dim) return;
  out[k] = s16 ? tile_elem<true>(tiles, KB, row, k) : tile_elem<false>(tiles, KB, row, k);
}

// k-way merge of per-shard result lists on the device (row-sharded search: after the
// all-gather every rank holds [n_lists][n_queries][k] (distance, docid) lists).  One
// workgroup per query sorts the n_lists*k keys (ord(distance) << 32 | docid) in LDS and
// writes the k best — the concatenate + sort tail of store.rs:1059,1090.
__global__ __launch_bounds__(SEL_THREADS) void vs_merge_lists_kernel(
    const uint32_t *__restrict__ docids, const float *__restrict__ dist, const uint32_t *__restrict__ counts,
    uint32_t n_lists, uint32_t n_queries, uint32_t k, uint32_t *__restrict__ out_docids,
    float *__restrict__ out_dist, uint32_t *__restrict__ out_counts) {
  __shared__ u64 sbuf[SEL_SORTCAP];
  const uint32_t j = blockIdx.x;
  const uint32_t total = n_lists * k;
  const uint32_t n = next_pow2(total < 2 ? 2 : total);
  for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) {
    u64 key = ~0ull;
    if (i < total) {
      const uint32_t l = i / k, r = i % k;
      const uint32_t c = min(counts[(size_t)l * n_queries + j], k);
      if (r < c) {
        const size_t at = ((size_t)l * n_queries + j) * k + r;
        key = ((u64)f32_to_ord(dist[at]) << 32) | docids[at];
      }
    }
    sbuf[i] = key;
  }
  __syncthreads();
  block_bitonic_sort(sbuf, n);
  uint32_t valid = 0;
  for (uint32_t l = 0; l < n_lists; ++l) valid += min(counts[(size_t)l * n_queries + j], k);
  const uint32_t out_n = valid < k ? valid : k;
  for (uint32_t i = threadIdx.x; i < k; i += blockDim.x) {
    const bool ok = i < out_n;
    out_docids[(size_t)j * k + i] = ok ? (uint32_t)sbuf[i] : 0xFFFFFFFFu;
    out_dist[(size_t)j * k + i] = ok ? ord_to_f32((uint32_t)(sbuf[i] >> 32)) : INFINITY;
  }
  if (threadIdx.x == 0) out_counts[j] = out_n;
}

}  // namespace

// ============================================================== host-side object

struct msi_vs {
  msi_ctx *ctx = nullptr;
  uint32_t dim = 0, dpad = 0, KB = 0;
  uint32_t nqt_max = 1;            // query tiles per sweep the LDS admits for this dim
  uint32_t nqt3_max = 1;           // ... with both halves of the queries in LDS (the bf16x3 second opinion of a bf16x2 store)
  bool bf3 = true;                 // contraction of the fast scan: bf16 MFMA (default) or f32 MFMA
  bool bf2 = false;                // ... bf16x2 (queries' hi halves only in LDS: twice the queries per sweep) instead of bf16x3
  bool s16 = false;                // rows stored as bf16 (MSI_VS_BF16)
  // the int8 copy of the rows and its sweep (f32 stores; MSI_VS_I8=0: none) — level 0 of the search when present
  bool i8 = false;                 // the store keeps the copy
  bool i8_dropped = false;         // ... it did, until its memory could not be had (finish_upload; msi_vs_stats::i8_bytes_per_tile reads 0 then)
  bool i8_now = false;             // the chunk being planned / enqueued sweeps it (set per sweep by the levels of effort)
  uint32_t KB8 = 0, nqt8_max = 1;  // KiB blocks per tile of the copy; query tiles per sweep over it
  DevBuf tiles8, scale8, i8small;  // [n_tiles][KB8][1 KiB] | s_x per row | {largest e_x of the store, as ordered bits}
  uint64_t i8_sweeps = 0, i8_scan_tiles = 0, device_rerun_queries = 0;
  uint64_t n_rows = 0, n_tiles = 0;
  DevBuf tiles, norm, inv_norm, docids;
  DevBuf tiles_next, docids_next, norm_next, inv_norm_next, add_tiles, add_docids, row_map;  // msi_vs_update builds the next store beside the current one
  std::vector<uint32_t> h_docids;  // for get_vector's binary search
  // scratch (guarded by ctx->mu)
  DevBuf qraw, qfrag, qfrag_bf, qfrag8, qrow, qsmall, gkeys, gcnt, gsmall, sel_keys, dense, frows, fsmall, fbits, out_docids,
      out_dist, exh_keys, rowtmp, resc_keys, rerun_q, rerun_flags, thr;
  // the second scratch set and stream of the device entry point's pipeline (msi_vs_search_device): allocated on first use
  struct Scratch2 {
    DevBuf qfrag, qfrag_bf, qfrag8, qrow, qsmall, gkeys, gcnt, gsmall, sel_keys, dense, resc_keys;
  } scr2;
  hipStream_t aux_stream = nullptr;
  hipEvent_t ev_start = nullptr, ev_done = nullptr, ev_pre[2] = {nullptr, nullptr}, ev_main[2] = {nullptr, nullptr};
  uint64_t pipelined_calls = 0;
  uint32_t capg = 0;
  uint32_t scan_grid = 0, scan_grid8 = 0;
  // stats
  uint64_t scan_launches = 0, scan_tiles = 0, exhaustive_reruns = 0, filtered_calls = 0;
  // How a query is proven adapts to the data (host entry point).  The proof needs every row whose fast score lies within
  // 2 x eps of the k-th neighbour among the K' rescored candidates; eps is a WORST-CASE bound (3.9e-3 in cosine for bf16x2,
  // 2.9e-4 for bf16x3 at d = 768: the f32 accumulation bound dominates) — on i.i.d. rows a few dozen rows lie that close and
  // K' = k + max(108, 3k) proves every query in the 96-query bf16x2 sweep, on CLUSTERED embeddings (a thousand rows within
  // 1e-2 of each other at 10 M rows / 10 k clusters) hundreds do and it proves none: round 3 then answered every such query
  // exhaustively, 30 ms each (measured round 4: 29.9 queries/s at C4's size).  Levels of effort, each tried on the queries the
  // one before could not prove: 0 = the store's contraction (bf16x2 by default), K' as above; 1 = the same contraction with
  // K' = KP_MAX candidates (2 048 rows rescored per query: 2 % of a sweep's bytes); 2 = bf16x3 with K' = KP_MAX (bf16x2 stores
  // only); then the exhaustive pass.  The level a sweep STARTS at follows the running share of queries its level could
  // not prove (level_ema > 1/4: one level up for the next 256 sweeps, then one probing sweep a level down).  Results do not
  // depend on any of it: every answer is the canonical rescoring of proven candidates.  MSI_VS_ADAPT=0: always start at 0.
  uint32_t level = 0, level_left = 0;
  float level_ema = 0.0f;
  bool big_slack = false;          // (read by enqueue_search: K' = KP_MAX)
  uint64_t second_opinion_queries = 0, x3_first_sweeps = 0, x2_sweeps = 0, level_sweeps[5] = {0, 0, 0, 0, 0};
  KernelTimer scan_timer;
  // micro-batcher: concurrent unfiltered msi_vs_search calls are fused into one sweep
  struct Pending {
    const float *queries;
    uint32_t n, k;
    uint32_t *out_docids;
    float *out_dist;
    uint32_t *out_counts;
    int32_t status = MSI_OK;
    std::string error;
    bool done = false;
  };
  std::mutex bmu;
  std::condition_variable bcv;
  std::vector<Pending *> bqueue;
  uint32_t bqueued_queries = 0;
  bool bleader_active = false;
  uint32_t microbatch_wait_us = 0;   // 0 = off
  uint64_t fused_calls = 0, fused_sweeps = 0;
  uint32_t sweep_split = 1;          // msi_vs_set_sweep_split
};

namespace {

// layout of the small scratch arrays
struct Small {
  float *qn, *inv_qn, *degth, *theta_inf, *theta, *sqs, *epsq;
  uint32_t *sel_cnt, *n_tiles, *n_items, *overflow, *bad, *inexact, *counts;
};

Small small_of(msi_vs *vs) {
  Small s;
  float *f = vs->qsmall.as<float>();
  s.qn = f;
  s.inv_qn = f + NQ_MAX;
  s.degth = f + 2 * NQ_MAX;
  s.theta_inf = f + 3 * NQ_MAX;
  s.theta = f + 4 * NQ_MAX;
  s.sqs = f + 5 * NQ_MAX;
  s.epsq = f + 6 * NQ_MAX;
  uint32_t *u = vs->gsmall.as<uint32_t>();
  s.sel_cnt = u;
  s.inexact = u + NQ_MAX;
  s.counts = u + 2 * NQ_MAX;
  s.n_tiles = u + 3 * NQ_MAX;
  s.n_items = u + 3 * NQ_MAX + 1;
  s.overflow = u + 3 * NQ_MAX + 2;
  s.bad = u + 3 * NQ_MAX + 3;
  return s;
}

size_t scan_lds_bytes(uint32_t KB, uint32_t nqt, bool s16 = false, bool bf2 = false) {
  if (bf2) return (size_t)nqt * KB * 32 * sizeof(float4);
  return (size_t)nqt * KB * 64 * sizeof(float4) * (s16 ? 2 : 1);
}

size_t scan8_lds_bytes(uint32_t KB8, uint32_t nqt) { return (size_t)nqt * KB8 * 64 * sizeof(i32x4) + 4 * (size_t)nqt * QT * sizeof(float); }

// Threshold rank r for the sample pass: the smallest r for which "fewer than kp
// rows of the whole store beat the r-th best of a p-fraction sample" has
// probability C(kp+r-1, r) p^r <= 1e-5 (a detected, exhaustively re-run event).
uint32_t threshold_rank(uint32_t kp, double p) {
  for (uint32_t r = 1; r < kp; ++r) {
    double lg = 0.0;  // log C(kp+r-1, r) + r log p
    for (uint32_t i = 1; i <= r; ++i) lg += log((double)(kp - 1 + i) / (double)i);
    lg += r * log(p);
    if (lg <= log(1e-5)) return r;
  }
  return kp;
}

// Bound on |fast cos - reference cos| assumed by the exactness proof: f32 accumulation
// of n terms (gamma_n with a factor 2 for the MFMA adder tree; n = dpad, 2·dpad or 3·dpad
// products per row) plus the bf16 split terms (2^-18 relative per dropped/rounded part).
float scan_eps(const msi_vs *vs) {
  const float u = 5.9604645e-8f, h = 3.8146973e-6f;
  if (vs->s16) return (4.0f * (float)vs->dpad + 64.0f) * u + 1.0f * h;
  // bf16x2: the query is rounded to bf16 (relative error <= 2^-8 per element, so |x.(q - q_hi)| <= 2^-8 |x||q|), the
  // row's lo.lo term does not exist and its lo is rounded (2^-16); two products of dpad terms each
  if (vs->bf2) return (4.0f * (float)vs->dpad + 64.0f) * u + 0.00390625f * 1.01f + 2.0f * 1.52587890625e-5f;
  if (vs->bf3) return (6.0f * (float)vs->dpad + 64.0f) * u + 3.0f * h;
  return (2.0f * (float)vs->dpad + 32.0f) * u;
}

int32_t ensure_scratch(msi_vs *vs) {
  MSI_TRY(vs->qraw.ensure((size_t)NQ_MAX * vs->dim * sizeof(float)));
  MSI_TRY(vs->qfrag.ensure((size_t)NQT_MAX * vs->KB * 64 * sizeof(float4)));
  MSI_TRY(vs->qfrag_bf.ensure((size_t)NQT_MAX * vs->KB * 64 * sizeof(float4) * (vs->s16 ? 2 : 1)));
  MSI_TRY(vs->qrow.ensure((size_t)NQ_MAX * vs->dpad * sizeof(float)));
  MSI_TRY(vs->qsmall.ensure(7 * NQ_MAX * sizeof(float)));
  if (vs->i8) MSI_TRY(vs->qfrag8.ensure((size_t)NQT_MAX * vs->KB8 * 64 * sizeof(i32x4)));
  MSI_TRY(vs->gsmall.ensure((3 * NQ_MAX + 8) * sizeof(uint32_t)));
  MSI_TRY(vs->gcnt.ensure((size_t)NQ_MAX * CNT_PAD * sizeof(uint32_t)));
  MSI_TRY(vs->sel_keys.ensure((size_t)NQ_MAX * KP_MAX * sizeof(u64)));
  return MSI_OK;
}

int32_t finish_upload(msi_vs *vs, uint64_t n_rows, const char *what);
int32_t tile_rows(msi_vs *vs, const float *rows, bool rows_on_device, uint64_t n_rows, void *dst);

int32_t upload_common(msi_vs *vs, const uint32_t *docids, bool docids_on_device, const float *rows,
                      bool rows_on_device, uint64_t n_rows) {
  msi_ctx *ctx = vs->ctx;
  hipStream_t st = ctx->stream;
  if (n_rows > 0xFFFFFFF0ull) {
    msi_set_error("msi_vs_upload: n_rows %llu exceeds the u32 row index space", (unsigned long long)n_rows);
    return MSI_E_UNSUPPORTED;
  }
  const uint64_t n_tiles = (n_rows + 15) / 16;
  const uint64_t padded = n_tiles * 16;
  MSI_TRY(vs->tiles.ensure(std::max<uint64_t>(1, n_tiles) * vs->KB * 64 * sizeof(float4)));
  MSI_TRY(vs->norm.ensure(std::max<uint64_t>(16, padded) * sizeof(float)));
  MSI_TRY(vs->inv_norm.ensure(std::max<uint64_t>(16, padded) * sizeof(float)));
  MSI_TRY(vs->docids.ensure(std::max<uint64_t>(16, padded) * sizeof(uint32_t)));
  MSI_TRY(ensure_scratch(vs));
  MSI_HIP_TRY(hipMemsetAsync(vs->docids.p, 0xFF, std::max<uint64_t>(16, padded) * sizeof(uint32_t), st));
  if (n_rows) {
    MSI_HIP_TRY(hipMemcpyAsync(vs->docids.p, docids, n_rows * sizeof(uint32_t),
                               docids_on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, st));
  }
  // sortedness check on device
  Small s = small_of(vs);
  MSI_HIP_TRY(hipMemsetAsync(s.bad, 0, sizeof(uint32_t), st));
  if (n_rows > 1)
    hipLaunchKernelGGL(vs_check_sorted_kernel, dim3(ceil_div_u32(n_rows, 256)), dim3(256), 0, st,
                       vs->docids.as<uint32_t>(), n_rows, s.bad);
  MSI_TRY(tile_rows(vs, rows, rows_on_device, n_rows, vs->tiles.p));
  return finish_upload(vs, n_rows, "msi_vs_upload");
}

// Row-major f32 rows -> the tiled layout at `dst` (f32 or bf16 slots), in chunks: bounded staging for host rows.
int32_t tile_rows(msi_vs *vs, const float *rows, bool rows_on_device, uint64_t n_rows, void *dst) {
  hipStream_t st = vs->ctx->stream;
  const uint64_t chunk_rows = rows_on_device ? n_rows : std::max<uint64_t>(16, ((64ull << 20) / (vs->dim * 4ull)) & ~15ull);
  if (!rows_on_device) MSI_TRY(vs->rowtmp.ensure(std::min<uint64_t>(chunk_rows, std::max<uint64_t>(n_rows, 1)) * vs->dim * sizeof(float)));
  for (uint64_t r0 = 0; r0 < n_rows; r0 += chunk_rows) {
    const uint64_t nc = std::min(chunk_rows, n_rows - r0);
    const float *src;
    if (rows_on_device) {
      src = rows + r0 * vs->dim;
    } else {
      MSI_HIP_TRY(hipMemcpyAsync(vs->rowtmp.p, rows + r0 * vs->dim, nc * vs->dim * sizeof(float),
                                 hipMemcpyHostToDevice, st));
      src = vs->rowtmp.as<float>();
    }
    const uint64_t total = ((nc + 15) / 16) * vs->KB * 64;
    if (vs->s16)
      hipLaunchKernelGGL(vs_tile_rows_bf16_kernel, dim3((uint32_t)((total + 255) / 256)), dim3(256), 0, st, src,
                         r0, nc, vs->dim, vs->KB, reinterpret_cast<bf16x8 *>(dst));
    else
      hipLaunchKernelGGL(vs_tile_rows_kernel, dim3((uint32_t)((total + 255) / 256)), dim3(256), 0, st, src,
                         r0, nc, vs->dim, vs->KB, reinterpret_cast<float4 *>(dst));
    if (!rows_on_device) MSI_HIP_TRY(hipStreamSynchronize(st));  // rowtmp is reused
  }
  MSI_HIP_TRY(hipGetLastError());
  return MSI_OK;
}

// Norms, tile count, the host copy of the docids, candidate lists: what follows the tiling of a store's rows.
int32_t finish_upload(msi_vs *vs, uint64_t n_rows, const char *what) {
  msi_ctx *ctx = vs->ctx;
  hipStream_t st = ctx->stream;
  const uint64_t n_tiles = (n_rows + 15) / 16;
  const uint64_t padded = n_tiles * 16;
  Small s = small_of(vs);
  if (padded && vs->s16)
    hipLaunchKernelGGL(vs_row_norms_bf16_kernel, dim3((uint32_t)((padded + 255) / 256)), dim3(256), 0, st,
                       vs->tiles.p, n_rows, vs->KB, vs->dim, vs->norm.as<float>(), vs->inv_norm.as<float>());
  else if (padded)
    hipLaunchKernelGGL(vs_row_norms_kernel, dim3((uint32_t)((padded + 255) / 256)), dim3(256), 0, st,
                       vs->tiles.as<float4>(), (uint64_t)0, n_rows, vs->KB, vs->norm.as<float>(),
                       vs->inv_norm.as<float>());
  if (vs->i8) {
    // the int8 copy follows the f32 rows (whole: an update re-gathers every tile anyway).  It is an accelerator, not the
    // store: when its memory cannot be had — an update holds two f32 copies at this point — the store goes on WITHOUT it
    // (level 0 becomes the f32 contraction) instead of failing an upload / emptying a store that was serving
    if (vs->tiles8.ensure(std::max<uint64_t>(1, n_tiles) * vs->KB8 * 64 * sizeof(i32x4)) != MSI_OK ||
        vs->scale8.ensure(std::max<uint64_t>(16, padded) * sizeof(float)) != MSI_OK || vs->i8small.ensure(64) != MSI_OK) {
      (void)hipGetLastError();   // (the failed allocation's sticky error)
      vs->tiles8.release();
      vs->scale8.release();
      vs->i8 = false;
      vs->i8_dropped = true;
    }
  }
  if (vs->i8) {
    MSI_HIP_TRY(hipMemsetAsync(vs->i8small.p, 0, 64, st));
    if (n_tiles && vs->s16)
      hipLaunchKernelGGL(vs_quantize_rows_kernel<true>, dim3((uint32_t)n_tiles), dim3(256), 0, st, (const void *)vs->tiles.p,
                         vs->inv_norm.as<float>(), n_rows, vs->KB, vs->tiles8.as<i32x4>(), vs->scale8.as<float>(),
                         vs->i8small.as<uint32_t>());
    else if (n_tiles)
      hipLaunchKernelGGL(vs_quantize_rows_kernel<false>, dim3((uint32_t)n_tiles), dim3(256), 0, st, (const void *)vs->tiles.p,
                         vs->inv_norm.as<float>(), n_rows, vs->KB, vs->tiles8.as<i32x4>(), vs->scale8.as<float>(),
                         vs->i8small.as<uint32_t>());
  }
  uint32_t nt32 = (uint32_t)n_tiles;
  MSI_HIP_TRY(hipMemcpyAsync(s.n_tiles, &nt32, sizeof(uint32_t), hipMemcpyHostToDevice, st));
  float ninf[NQ_MAX];
  for (int i = 0; i < NQ_MAX; ++i) ninf[i] = -INFINITY;  // "everything passes"
  MSI_HIP_TRY(hipMemcpyAsync(s.theta_inf, ninf, sizeof(ninf), hipMemcpyHostToDevice, st));
  uint32_t bad = 0;
  MSI_HIP_TRY(hipMemcpyAsync(&bad, s.bad, sizeof(uint32_t), hipMemcpyDeviceToHost, st));
  vs->h_docids.resize(n_rows);
  if (n_rows)
    MSI_HIP_TRY(hipMemcpyAsync(vs->h_docids.data(), vs->docids.p, n_rows * sizeof(uint32_t),
                               hipMemcpyDeviceToHost, st));
  MSI_HIP_TRY(hipStreamSynchronize(st));
  MSI_HIP_TRY(hipGetLastError());
  if (bad) {
    vs->n_rows = 0;
    vs->n_tiles = 0;
    vs->h_docids.clear();  // a rejected upload leaves an EMPTY store: get_vector / search_by_item / update must not see its list
    msi_set_error("%s: docids must be strictly ascending", what);
    return MSI_E_NOT_SORTED;
  }
  vs->n_rows = n_rows;
  vs->n_tiles = n_tiles;
  // sparse-pass candidate lists: room for 256 Ki keys per query (the thresholds
  // leave ~1e3; an overflow is detected and re-run exhaustively)
  vs->capg = 1u << 18;
  MSI_TRY(vs->gkeys.ensure((size_t)NQ_MAX * vs->capg * sizeof(u64)));
  return MSI_OK;
}

// Launch one sweep.  `dense` selects the epilogue, nqt the number of 16-query tiles.
void launch_scan(msi_vs *vs, const ScanArgs &sa, uint32_t nqt, bool dense, hipStream_t st = nullptr, uint32_t grid_wgs = 0) {
  const size_t lds = scan_lds_bytes(vs->KB, nqt, vs->s16, vs->bf2);
  const dim3 grid(grid_wgs ? grid_wgs : vs->scan_grid), block(SCAN_WAVES * 64);
  if (!st) st = vs->ctx->stream;
#define MSI_SCAN_LAUNCH(N, D, B, S)                                                                            \
  do {                                                                                                         \
    if (sa.rows) hipLaunchKernelGGL((vs_scan_kernel<SCAN_WAVES, N, D, B, S, true>), grid, block, lds, st, sa); \
    else hipLaunchKernelGGL((vs_scan_kernel<SCAN_WAVES, N, D, B, S, false>), grid, block, lds, st, sa);        \
  } while (0)
#define MSI_SCAN_CASE(N)                                              \
  case N:                                                             \
    if (vs->s16) {                                                    \
      if (dense) MSI_SCAN_LAUNCH(N, true, 1, true);                   \
      else MSI_SCAN_LAUNCH(N, false, 1, true);                        \
    } else if (dense) {                                               \
      if (vs->bf2) MSI_SCAN_LAUNCH(N, true, 2, false);                \
      else if (vs->bf3) MSI_SCAN_LAUNCH(N, true, 1, false);           \
      else MSI_SCAN_LAUNCH(N, true, 0, false);                        \
    } else {                                                          \
      if (vs->bf2) MSI_SCAN_LAUNCH(N, false, 2, false);               \
      else if (vs->bf3) MSI_SCAN_LAUNCH(N, false, 1, false);          \
      else MSI_SCAN_LAUNCH(N, false, 0, false);                       \
    }                                                                 \
    break;
#define MSI_SCAN_CASE2(N)                                             \
  case N:                                                             \
    if (dense) MSI_SCAN_LAUNCH(N, true, 2, false);                    \
    else MSI_SCAN_LAUNCH(N, false, 2, false);                         \
    break;
  switch (nqt) {
    MSI_SCAN_CASE(1)
    MSI_SCAN_CASE(2)
    MSI_SCAN_CASE(3)
    MSI_SCAN_CASE2(4)   // (only the bf16x2 contraction admits more than 3 query tiles)
    MSI_SCAN_CASE2(5)
    MSI_SCAN_CASE2(6)
  }
#undef MSI_SCAN_CASE2
#undef MSI_SCAN_LAUNCH
#undef MSI_SCAN_CASE
}

// The int8 sweep: instantiated for 1 / 2 / 3 / 4 / 6 / 8 / 12 query tiles (a sweep of fewer queries runs the next size up: its
// spare query slots are zero fragments whose scores nobody keeps), RT row tiles per wave step by the accumulators that leaves
// room for, GS = the largest of 4 / 3 / 2 that divides KB8 (dpad is a multiple of 128, so KB8 is even).
uint32_t scan8_nqt_of(uint32_t nqt) {
  static const uint32_t sizes[] = {1, 2, 3, 4, 6, 8, 12};
  for (uint32_t v : sizes)
    if (nqt <= v) return v;
  return 12;
}
uint32_t scan8_gs_of(uint32_t KB8) { return KB8 % 4 == 0 ? 4u : (KB8 % 3 == 0 ? 3u : 2u); }
#define MSI_SCAN8_RT(N) ((N) <= 3 ? 4 : ((N) <= 4 ? 3 : 2))
#define MSI_SCAN8_EACH(X) X(1) X(2) X(3) X(4) X(6) X(8) X(12)
template <int GS>
int32_t scan8_set_attributes() {
#define MSI_X(N)                                                                                                          \
  for (const void *fn : {reinterpret_cast<const void *>(&vs_scan_i8_kernel<SCAN_WAVES, N, MSI_SCAN8_RT(N), GS, true, false>),      \
                         reinterpret_cast<const void *>(&vs_scan_i8_kernel<SCAN_WAVES, N, MSI_SCAN8_RT(N), GS, false, false>),     \
                         reinterpret_cast<const void *>(&vs_scan_i8_kernel<SCAN_WAVES, N, MSI_SCAN8_RT(N), GS, true, true>),       \
                         reinterpret_cast<const void *>(&vs_scan_i8_kernel<SCAN_WAVES, N, MSI_SCAN8_RT(N), GS, false, true>)})     \
    if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_MAX) != hipSuccess) return MSI_E_HIP;
  MSI_SCAN8_EACH(MSI_X)
#undef MSI_X
  return MSI_OK;
}
template <int GS>
void launch_scan8_gs(const Scan8Args &sa, uint32_t n, bool dense, dim3 grid, dim3 block, size_t lds, hipStream_t st) {
  switch (n) {
#define MSI_X(N)                                                                                                         \
  case N:                                                                                                                \
    if (sa.rows) {                                                                                                       \
      if (dense) hipLaunchKernelGGL((vs_scan_i8_kernel<SCAN_WAVES, N, MSI_SCAN8_RT(N), GS, true, true>), grid, block, lds, st, sa);   \
      else hipLaunchKernelGGL((vs_scan_i8_kernel<SCAN_WAVES, N, MSI_SCAN8_RT(N), GS, false, true>), grid, block, lds, st, sa);        \
    } else if (dense) hipLaunchKernelGGL((vs_scan_i8_kernel<SCAN_WAVES, N, MSI_SCAN8_RT(N), GS, true, false>), grid, block, lds, st, sa); \
    else hipLaunchKernelGGL((vs_scan_i8_kernel<SCAN_WAVES, N, MSI_SCAN8_RT(N), GS, false, false>), grid, block, lds, st, sa);      \
    break;
    MSI_SCAN8_EACH(MSI_X)
#undef MSI_X
  }
}
// Experiments (MSI_VS_I8_VARIANT=<n>, read per sweep; 8 query tiles only): other shapes of the same kernel — waves per
// workgroup, row tiles per wave step, KiB blocks per pipeline stage.  0 / unset: the default shape.
#define MSI_SCAN8_VARIANTS(X) X(1, 16, 1, 4) X(2, 8, 1, 4) X(3, 8, 2, 2) X(4, 16, 1, 2) X(5, 16, 2, 2) X(6, 8, 4, 2) X(7, 16, 2, 1)
bool launch_scan8_variant(int variant, uint32_t KB8, const Scan8Args &sa, bool dense, uint32_t n_wg, size_t lds, hipStream_t st) {
  if (sa.rows) return false;   // (the experiments' shapes exist for unfiltered sweeps only)
  switch (variant) {
#define MSI_X(V, W, R, G)                                                                                                 \
  case V: {                                                                                                               \
    if (KB8 % G) return false;                                                                                            \
    static bool attr = false;                                                                                             \
    if (!attr) {                                                                                                          \
      (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&vs_scan_i8_kernel<W, 8, R, G, true, false>),                      \
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_MAX);                                \
      (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&vs_scan_i8_kernel<W, 8, R, G, false, false>),                     \
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_MAX);                                \
      attr = true;                                                                                                        \
    }                                                                                                                     \
    if (dense) hipLaunchKernelGGL((vs_scan_i8_kernel<W, 8, R, G, true, false>), dim3(n_wg), dim3(W * 64), lds, st, sa);            \
    else hipLaunchKernelGGL((vs_scan_i8_kernel<W, 8, R, G, false, false>), dim3(n_wg), dim3(W * 64), lds, st, sa);                \
    return true;                                                                                                          \
  }
    MSI_SCAN8_VARIANTS(MSI_X)
#undef MSI_X
  }
  return false;
}
// msi_vs_set_sweep_split(vs, n) / MSI_VS_GRID_MULT=<n> (read per sweep): a full sweep of the int8 copy as n times the
// workgroups, each with 1 / n of the rows — short workgroups (~90 us at n = 16, C4) that let the keyword searches' rounds in
// between when both legs of a hybrid step share the device, at the price of n times the query fragments' LDS fills (the
// sweep alone: 67 -> 63 k q/s at n = 16).  Measured (profiles/r6_overlap_short_wgs.log): the overlapped step 61.5 -> 58.6-59.2 ms.
static uint32_t scan_grid_mult(const msi_vs *vs) {
  const char *e = getenv("MSI_VS_GRID_MULT");
  return e ? (uint32_t)std::max(1, std::min(64, atoi(e))) : std::max(1u, vs->sweep_split);
}
void launch_scan8(msi_vs *vs, const Scan8Args &sa, uint32_t nqt, bool dense, hipStream_t st = nullptr, uint32_t grid_wgs = 0) {
  const uint32_t n = scan8_nqt_of(nqt);
  const size_t lds = scan8_lds_bytes(vs->KB8, n);
  const dim3 grid(grid_wgs ? grid_wgs : vs->scan_grid8 * (dense ? 1u : scan_grid_mult(vs))), block(SCAN_WAVES * 64);
  if (!st) st = vs->ctx->stream;
  if (n == 8) {
    const char *v = getenv("MSI_VS_I8_VARIANT");
    if (v && atoi(v) > 0 && launch_scan8_variant(atoi(v), vs->KB8, sa, dense, grid.x, lds, st)) return;
  }
  const uint32_t gs = scan8_gs_of(vs->KB8);
  if (gs == 4) launch_scan8_gs<4>(sa, n, dense, grid, block, lds, st);
  else if (gs == 3) launch_scan8_gs<3>(sa, n, dense, grid, block, lds, st);
  else launch_scan8_gs<2>(sa, n, dense, grid, block, lds, st);
}

// One chunk of <= 16*nqt_max queries (one HBM sweep): planned against the scratch set that is current when it is planned,
// then enqueued in three stages —
//   pre   queries -> MFMA fragments, norms; the strided sample sweep and the thresholds it yields; counters zeroed
//   main  the full sweep (the HBM-bound kernel)
//   post  selection of the K' best candidates, canonical rescoring, exactness proof, outputs
// enqueue_search() runs the three back to back on the context's stream; msi_vs_search_device() runs `pre` and `post` on
// the store's second stream, double-buffered, so that they overlap the neighbouring chunks' `main` (the sweep is 89 %
// of a chunk's time; the rest was serial in front of and behind it).
struct Chunk {
  ScanArgs sa;
  Scan8Args s8;
  bool i8 = false;            // this chunk sweeps the int8 copy
  SelectArgs se;
  RescoreArgs ra;
  const float *d_queries = nullptr;
  Small s;
  float4 *qfrag = nullptr;
  bf16x8 *qfrag_bf = nullptr;
  i32x4 *qfrag8 = nullptr;
  float *qrow = nullptr;
  void *gcnt = nullptr;
  uint32_t nq = 0, nqt = 0, k = 0, kp = 0, stride = 1, thr_rank = 0;
  uint64_t n_tiles = 0, dense_items = 0;
  bool filtered = false;
  uint32_t grid_main = 0, grid_sample = 0;   // 0 = the store's grid (one workgroup per CU); the pipeline sets both
};

// the items a filtered sweep visits (vs_filter_rows_kernel; the same for every chunk of a call)
int32_t enqueue_filter(msi_vs *vs, const u64 *d_fbits, uint64_t nbits, hipStream_t st) {
  if (vs->n_rows >= 0x7FFFFFFFull) {
    msi_set_error("msi_vs_search: a candidate filter over a store of 2^31 rows or more is not supported");
    return MSI_E_UNSUPPORTED;
  }
  MSI_TRY(vs->frows.ensure(vs->n_tiles * 16 * sizeof(uint32_t)));
  MSI_TRY(vs->fsmall.ensure(64));
  MSI_HIP_TRY(hipMemsetAsync(vs->fsmall.p, 0, 4 * sizeof(uint32_t), st));
  const uint64_t padded = vs->n_tiles * 16;
  const uint64_t rows_per_block = 1024ull * FT_SUB;
  // MSI_VS_GATHER_PCT (read per call; measurements and tests): what a gathered row costs in units of a streamed row, in per
  // cent (default 250; 0: every region compacts its allowed rows; 1600 and above: none does — rounds 1-5's tile-granular sweep)
  const char *e = getenv("MSI_VS_GATHER_PCT");
  const uint32_t pct = e ? (uint32_t)std::max(0, atoi(e)) : 250u;
  hipLaunchKernelGGL(vs_filter_rows_kernel, dim3((uint32_t)((padded + rows_per_block - 1) / rows_per_block)),
                     dim3(1024), 0, st, vs->docids.as<uint32_t>(), vs->n_rows, d_fbits, nbits,
                     vs->frows.as<uint32_t>(), vs->fsmall.as<uint32_t>(), pct, 100u);
  vs->filtered_calls++;
  return MSI_OK;
}

int32_t plan_chunk(msi_vs *vs, Chunk &c, const float *d_queries, uint32_t nq, uint32_t k, bool filtered,
                   uint32_t *d_out_docids, float *d_out_dist, uint32_t *d_out_counts, uint32_t *d_inexact) {
  if (k > KP_MAX) {
    msi_set_error("msi_vs_search: k=%u above the supported maximum %u", k, KP_MAX);
    return MSI_E_UNSUPPORTED;
  }
  c.s = small_of(vs);
  const Small &s = c.s;
  c.d_queries = d_queries;
  c.nq = nq;
  c.k = k;
  c.filtered = filtered;
  // candidates rescored beyond k: every row whose fast score lies within twice the proof's eps of the k-th must be among
  // them — a handful with bf16x3 (eps ~ 1e-5), a few dozen to a hundred with bf16x2 (eps ~ 4e-3)
  // (the int8 sweep: eps ~ 1.8e-2 in cosine — on i.i.d. rows ~200 rows of 10 M lie within it of the 20th neighbour, a few
  // per cent of the queries see 500+; the count grows with the store, so K' does too: 256 / 512 / 1 024 candidates, of which
  // only those within 2 eps of the k-th are rescored (candidate_matters); the next level takes K' = KP_MAX)
  c.i8 = vs->i8 && vs->i8_now;
  const uint32_t slack = vs->big_slack ? KP_MAX
                         : c.i8 ? std::max<uint32_t>(vs->n_rows > 5000000 ? 1004u : (vs->n_rows > 1500000 ? 492u : 236u), 3 * k)
                                : (vs->bf2 ? std::max<uint32_t>(108, 3 * k) : std::max<uint32_t>(12, k / 4));
  const uint32_t kp = c.kp = std::min<uint32_t>(k + slack, KP_MAX);
  c.nqt = (nq + QT - 1) / QT;
  c.qfrag = vs->qfrag.as<float4>();
  c.qfrag_bf = vs->qfrag_bf.as<bf16x8>();
  c.qrow = vs->qrow.as<float>();
  c.qfrag8 = vs->qfrag8.as<i32x4>();
  c.gcnt = vs->gcnt.p;
  const uint32_t *frows = nullptr;
  const uint32_t *n_items_ptr = s.n_tiles;
  if (filtered) {
    frows = vs->frows.as<uint32_t>();
    n_items_ptr = vs->fsmall.as<uint32_t>();
  }
  // plan: small stores are scored densely in one sweep; larger ones get a dense
  // strided sample sweep (thresholds) followed by the sparse full sweep.
  const uint64_t n_tiles = c.n_tiles = vs->n_tiles;
  const uint64_t n_rows_pad = n_tiles * 16;
  uint32_t stride = 1;
  uint32_t thr_rank = kp;
  if (n_rows_pad > 32768) {
    const double s_rows = std::min<double>((double)n_rows_pad / 4.0,
                                           std::max(4096.0, 2.0 * sqrt((double)kp * (double)n_rows_pad)));
    stride = (uint32_t)std::max<uint64_t>(1, (uint64_t)((double)n_rows_pad / s_rows));
    if (stride > 1) thr_rank = std::min(kp, threshold_rank(kp, 1.0 / (double)stride));
  }
  c.stride = stride;
  c.thr_rank = thr_rank;
  const uint64_t dense_items = c.dense_items = (n_tiles + stride - 1) / stride;
  const uint32_t dstride = (uint32_t)(dense_items * 16);
  MSI_TRY(vs->dense.ensure((size_t)NQ_MAX * std::max<uint32_t>(16, dstride) * sizeof(float)));

  ScanArgs &sa = c.sa;
  sa.tiles = vs->tiles.as<float4>();
  sa.inv_norm = vs->inv_norm.as<float>();
  sa.qfrag = (vs->bf3 || vs->s16) ? vs->qfrag_bf.as<float4>() : vs->qfrag.as<float4>();
  sa.theta = s.theta_inf;
  sa.degth = s.degth;
  sa.n_items_ptr = n_items_ptr;
  sa.rows = frows;
  sa.gkeys = vs->gkeys.as<u64>();
  sa.gcnt = vs->gcnt.as<uint32_t>();
  sa.overflow = s.overflow;
  sa.dense = vs->dense.as<float>();
  sa.n_rows = vs->n_rows;
  sa.dstride = dstride;
  sa.capg = vs->capg;
  sa.KB = vs->KB;
  sa.stride = stride;
  sa.nq = nq;
  if (c.i8) {
    Scan8Args &s8 = c.s8;
    s8.tiles8 = vs->tiles8.as<i32x4>();
    s8.scale8 = vs->scale8.as<float>();
    s8.inv_norm = sa.inv_norm;
    s8.i8small = vs->i8small.as<uint32_t>();
    s8.qfrag8 = vs->qfrag8.as<i32x4>();
    s8.sqs = s.sqs;
    s8.theta = sa.theta;
    s8.degth = sa.degth;
    s8.n_items_ptr = sa.n_items_ptr;
    s8.rows = sa.rows;
    s8.list = nullptr;
    s8.tmask = nullptr;
    s8.gkeys = sa.gkeys;
    s8.gcnt = sa.gcnt;
    s8.overflow = sa.overflow;
    s8.dense = sa.dense;
    s8.n_rows = sa.n_rows;
    s8.dstride = sa.dstride;
    s8.capg = sa.capg;
    s8.KB8 = vs->KB8;
    s8.stride = sa.stride;
    s8.nq = nq;
  }
  SelectArgs &se = c.se;
  se.gkeys = vs->gkeys.as<u64>();
  se.gcnt = vs->gcnt.as<uint32_t>();
  se.capg = vs->capg;
  se.dense = vs->dense.as<float>();
  se.dstride = dstride;
  se.n_items_ptr = n_items_ptr;
  se.rows = frows;
  se.stride = stride;
  se.sel_keys = vs->sel_keys.as<u64>();
  se.sel_cnt = s.sel_cnt;
  se.theta = s.theta;
  se.fixed_c = 0;
  // rescore with the reference arithmetic, order, prove exactness
  RescoreArgs &ra = c.ra;
  ra.tiles = vs->tiles.p;
  ra.dpad = vs->dpad;
  ra.s16 = vs->s16 ? 1u : 0u;
  ra.norm = vs->norm.as<float>();
  ra.docids = vs->docids.as<uint32_t>();
  ra.qrow = vs->qrow.as<float>();
  ra.qn = s.qn;
  ra.inv_qn = s.inv_qn;
  ra.sel_keys = vs->sel_keys.as<u64>();
  ra.sel_cnt = s.sel_cnt;
  ra.KB = vs->KB;
  ra.kp = kp;
  ra.k = k;
  // bound on |fast cos - reference cos|: f32 accumulation of n terms (gamma_n, n = dpad or
  // 3·dpad, with a factor 2 for the MFMA adder tree) + the bf16x3 split's 3·2^-18 per product
  ra.eps = c.i8 ? 0.0f : scan_eps(vs);   // (the int8 sweep's bound is per query: s.epsq, written by vs_prep_queries_i8_kernel)
  ra.eps_q = c.i8 ? s.epsq : nullptr;
  ra.out_docids = d_out_docids;
  ra.out_dist = d_out_dist;
  ra.out_counts = d_out_counts;
  ra.inexact = d_inexact;
  ra.overflow = s.overflow;
  ra.theta = stride == 1 ? nullptr : s.theta;
  // (MSI_VS_REFINE=0: every candidate inside the sweep's window goes to the reference arithmetic, as in round 5)
  static const bool refine_off = getenv("MSI_VS_REFINE") && getenv("MSI_VS_REFINE")[0] == '0';
  ra.refined = nullptr;
  if (k > 0 && !refine_off) {
    MSI_TRY(vs->resc_keys.ensure((size_t)NQ_MAX * KP_MAX * sizeof(u64)));
    ra.refined = vs->resc_keys.as<float>();
  }
  MSI_TRY(vs->thr.ensure(((size_t)2 * NQ_MAX * 2048 + (size_t)NQ_MAX * 2) * sizeof(uint32_t)));
  return MSI_OK;
}

int32_t chunk_pre(msi_vs *vs, Chunk &c, hipStream_t st) {
  hipLaunchKernelGGL(vs_prep_queries_kernel, dim3(c.nqt * QT), dim3(256), (size_t)vs->dpad * sizeof(float), st,
                     c.d_queries, c.nq, vs->dim, vs->KB, c.qfrag, c.qfrag_bf, c.qrow, c.s.qn, c.s.inv_qn, c.s.degth,
                     vs->s16 ? 1u : (vs->bf2 ? 2u : 0u));
  if (c.i8) {
    // the f32 terms every level carries (reference accumulation of dpad terms, the scales' roundings) + slack
    const float eps_base = (4.0f * (float)vs->dpad + 64.0f) * 5.9604645e-8f + 1e-5f;
    hipLaunchKernelGGL(vs_prep_queries_i8_kernel, dim3(c.nqt * QT), dim3(256), 0, st, c.qrow, c.s.qn, c.s.inv_qn, c.nq, vs->dpad,
                       c.qfrag8, c.s.sqs, c.s.epsq, vs->i8small.as<uint32_t>(), eps_base);
  }
  MSI_HIP_TRY(hipMemsetAsync(c.s.overflow, 0, sizeof(uint32_t), st));
  if (c.stride > 1) {
    // sample sweep -> thresholds
    if (c.i8) launch_scan8(vs, c.s8, c.nqt, true, st, c.grid_sample);
    else launch_scan(vs, c.sa, c.nqt, true, st, c.grid_sample);
    vs->scan_launches++;
    vs->scan_tiles += c.dense_items;
    if (c.i8) vs->i8_scan_tiles += c.dense_items;
    c.se.K = c.thr_rank;
    c.se.mode = 0;
    // a large sample is selected from in slices, by the whole chip (vs_select_part_kernel); MSI_VS_SELECT_PARTS=0: one
    // workgroup per query as before
    // MSI_VS_THRESHOLD=select: the exact r-th best sampled score (round 5's selection; in slices when the sample is large);
    // default: the conservative threshold of two histogram passes (vs_thr_hist_kernel)
    static const bool thr_select = getenv("MSI_VS_THRESHOLD") && !strcmp(getenv("MSI_VS_THRESHOLD"), "select");
    if (!thr_select) {
      ThrArgs ta;
      ta.dense = c.se.dense;
      ta.dstride = c.se.dstride;
      ta.n_items_ptr = c.se.n_items_ptr;
      ta.stride = c.se.stride;
      ta.rank = c.thr_rank;
      ta.hist = vs->thr.as<uint32_t>();
      ta.pick = ta.hist + (size_t)2 * NQ_MAX * 2048;
      ta.theta = c.se.theta;
      MSI_HIP_TRY(hipMemsetAsync(ta.hist, 0, (size_t)2 * NQ_MAX * 2048 * sizeof(uint32_t), st));
      // slices so that ~1 000 workgroups share the pass (one workgroup per query left half the chip idle)
      const uint32_t nqp = c.nqt * QT;
      const uint32_t slices = std::max<uint32_t>(1, std::min<uint32_t>(64, (1024 + nqp - 1) / nqp));
      hipLaunchKernelGGL(vs_thr_hist_kernel<0>, dim3(slices, nqp), dim3(256), 0, st, ta);
      hipLaunchKernelGGL(vs_thr_pick_kernel<0>, dim3(nqp), dim3(256), 0, st, ta);
      hipLaunchKernelGGL(vs_thr_hist_kernel<1>, dim3(slices, nqp), dim3(256), 0, st, ta);
      hipLaunchKernelGGL(vs_thr_pick_kernel<1>, dim3(nqp), dim3(256), 0, st, ta);
      MSI_HIP_TRY(hipMemsetAsync(c.gcnt, 0, (size_t)NQ_MAX * CNT_PAD * sizeof(uint32_t), st));
      MSI_HIP_TRY(hipGetLastError());
      return MSI_OK;
    }
    static const int parts_knob = getenv("MSI_VS_SELECT_PARTS") ? atoi(getenv("MSI_VS_SELECT_PARTS")) : -1;
    uint32_t parts = parts_knob >= 0 ? (uint32_t)parts_knob : 32u;
    parts = std::min<uint32_t>(parts, KP_MAX / std::max<uint32_t>(1, c.thr_rank));
    if (c.dense_items * 16 < 16384) parts = 0;
    if (parts >= 2) {
      hipLaunchKernelGGL(vs_select_part_kernel, dim3(parts, c.nqt * QT), dim3(SEL_THREADS), 0, st, c.se);
      SelectArgs fin = c.se;
      fin.dense = nullptr;
      fin.gkeys = c.se.sel_keys;
      fin.capg = KP_MAX;
      fin.fixed_c = parts * c.thr_rank;
      hipLaunchKernelGGL(vs_select_kernel, dim3(c.nqt * QT), dim3(SEL_THREADS), 0, st, fin);
    } else {
      hipLaunchKernelGGL(vs_select_kernel, dim3(c.nqt * QT), dim3(SEL_THREADS), 0, st, c.se);
    }
    MSI_HIP_TRY(hipMemsetAsync(c.gcnt, 0, (size_t)NQ_MAX * CNT_PAD * sizeof(uint32_t), st));
  }
  MSI_HIP_TRY(hipGetLastError());
  return MSI_OK;
}

int32_t chunk_main(msi_vs *vs, Chunk &c, hipStream_t st) {
  msi_ctx *ctx = vs->ctx;
  if (c.stride == 1) {
    // dense main sweep; the K' best come straight out of the score matrix
    vs->scan_timer.begin(ctx, st);
    if (c.i8) launch_scan8(vs, c.s8, c.nqt, true, st, c.grid_main);
    else launch_scan(vs, c.sa, c.nqt, true, st, c.grid_main);
    vs->scan_timer.end(ctx);
  } else {
    // full sweep, sparse epilogue
    c.sa.theta = c.s8.theta = c.s.theta;
    c.sa.stride = c.s8.stride = 1;
    vs->scan_timer.begin(ctx, st);
    if (c.i8) launch_scan8(vs, c.s8, c.nqt, false, st, c.grid_main);
    else launch_scan(vs, c.sa, c.nqt, false, st, c.grid_main);
    vs->scan_timer.end(ctx);
    c.se.dense = nullptr;
  }
  if (c.i8) {
    vs->i8_sweeps++;
    vs->i8_scan_tiles += c.n_tiles;
  }
  vs->scan_launches++;
  vs->scan_tiles += c.n_tiles;
  MSI_HIP_TRY(hipGetLastError());
  return MSI_OK;
}

int32_t chunk_post(msi_vs *vs, Chunk &c, hipStream_t st) {
  c.se.K = c.kp;
  c.se.mode = 1;
  hipLaunchKernelGGL(vs_select_kernel, dim3(c.nq), dim3(SEL_THREADS), 0, st, c.se);
  if (c.ra.refined)
    hipLaunchKernelGGL(vs_refine_kernel, dim3(RD_WGS_PER_QUERY, c.nq), dim3(SEL_THREADS), (size_t)vs->dpad * sizeof(float), st, c.ra);
  if (c.k > 0)
    hipLaunchKernelGGL(vs_rescore_kernel, dim3(c.nq), dim3(SEL_THREADS),
                       2 * KP_MAX * sizeof(u64) + (size_t)vs->dpad * sizeof(float), st, c.ra);
  MSI_HIP_TRY(hipGetLastError());
  return MSI_OK;
}

// Enqueue the full pipeline for <= 16*nqt_max queries already in device memory, on the context's stream.
// d_fbits nullable.  Outputs are device pointers.
int32_t enqueue_search(msi_vs *vs, const float *d_queries, uint32_t nq, uint32_t k, const u64 *d_fbits,
                       uint64_t nbits, uint32_t *d_out_docids, float *d_out_dist, uint32_t *d_out_counts,
                       uint32_t *d_inexact) {
  hipStream_t st = vs->ctx->stream;
  const bool filtered = d_fbits && vs->n_rows;
  if (filtered) MSI_TRY(enqueue_filter(vs, d_fbits, nbits, st));
  Chunk c;
  MSI_TRY(plan_chunk(vs, c, d_queries, nq, k, filtered, d_out_docids, d_out_dist, d_out_counts, d_inexact));
  MSI_TRY(chunk_pre(vs, c, st));
  MSI_TRY(chunk_main(vs, c, st));
  return chunk_post(vs, c, st);
}

// Several chunks: `pre` and `post` of a chunk run on the store's second stream against two scratch sets, `main` (the
// sweep) on the context's stream —
//   second stream   pre(0) | pre(1) post(0) | pre(2) post(1) | ...        | post(n-1)
//   context stream          main(0)         | main(1)        | ... main(n-1)          | (waits for post(n-1))
// main(i) waits for pre(i); post(i) waits for main(i); pre(i+2) follows post(i) on the same stream, so a scratch set is
// rewritten only when the chunk that used it is finished.  The outputs are complete when the context's stream is.
void swap_scratch(msi_vs *vs) {
  std::swap(vs->qfrag, vs->scr2.qfrag);
  std::swap(vs->qfrag_bf, vs->scr2.qfrag_bf);
  std::swap(vs->qfrag8, vs->scr2.qfrag8);
  std::swap(vs->qrow, vs->scr2.qrow);
  std::swap(vs->qsmall, vs->scr2.qsmall);
  std::swap(vs->gkeys, vs->scr2.gkeys);
  std::swap(vs->gcnt, vs->scr2.gcnt);
  std::swap(vs->gsmall, vs->scr2.gsmall);
  std::swap(vs->sel_keys, vs->scr2.sel_keys);
  std::swap(vs->dense, vs->scr2.dense);
  std::swap(vs->resc_keys, vs->scr2.resc_keys);
}

int32_t search_device_pipelined_impl(msi_vs *vs, const float *d_queries, uint32_t n_queries, uint32_t k, const u64 *d_fbits,
                                     uint64_t nbits, uint32_t *d_out_docids, float *d_out_dist, uint32_t *d_out_counts,
                                     uint32_t *d_inexact);
int32_t search_device_pipelined(msi_vs *vs, const float *d_queries, uint32_t n_queries, uint32_t k, const u64 *d_fbits,
                                uint64_t nbits, uint32_t *d_out_docids, float *d_out_dist, uint32_t *d_out_counts,
                                uint32_t *d_inexact) {
  const int32_t st = search_device_pipelined_impl(vs, d_queries, n_queries, k, d_fbits, nbits, d_out_docids, d_out_dist,
                                                  d_out_counts, d_inexact);
  if (st != MSI_OK) {
    // an error path left work on the second stream that the context's stream never joined: settle both before the caller
    // sees the failure (its buffers may be freed next) — ADVICE r4
    if (vs->aux_stream) (void)hipStreamSynchronize(vs->aux_stream);
    (void)hipStreamSynchronize(vs->ctx->stream);
  }
  return st;
}
int32_t search_device_pipelined_impl(msi_vs *vs, const float *d_queries, uint32_t n_queries, uint32_t k, const u64 *d_fbits,
                                     uint64_t nbits, uint32_t *d_out_docids, float *d_out_dist, uint32_t *d_out_counts,
                                     uint32_t *d_inexact) {
  msi_ctx *ctx = vs->ctx;
  hipStream_t A = ctx->stream;
  if (!vs->aux_stream) {
    MSI_HIP_TRY(hipStreamCreateWithFlags(&vs->aux_stream, hipStreamNonBlocking));
    for (hipEvent_t *e : {&vs->ev_start, &vs->ev_done, &vs->ev_pre[0], &vs->ev_pre[1], &vs->ev_main[0], &vs->ev_main[1]})
      MSI_HIP_TRY(hipEventCreateWithFlags(e, hipEventDisableTiming));
  }
  hipStream_t B = vs->aux_stream;
  // the second scratch set (the first call allocates it)
  swap_scratch(vs);
  int32_t st2 = ensure_scratch(vs);
  if (st2 == MSI_OK) st2 = vs->gkeys.ensure((size_t)NQ_MAX * vs->capg * sizeof(u64));
  const Small s2 = small_of(vs);
  swap_scratch(vs);
  MSI_TRY(st2);
  const Small s1 = small_of(vs);
  // whatever the caller enqueued on the context's stream (the queries, the filter) is done before the second stream reads it
  MSI_HIP_TRY(hipEventRecord(vs->ev_start, A));
  MSI_HIP_TRY(hipStreamWaitEvent(B, vs->ev_start, 0));
  // the store's constants of the small arrays (tile count, "everything passes" thresholds) follow into the second set
  MSI_HIP_TRY(hipMemcpyAsync(s2.n_tiles, s1.n_tiles, sizeof(uint32_t), hipMemcpyDeviceToDevice, B));
  MSI_HIP_TRY(hipMemcpyAsync(s2.theta_inf, s1.theta_inf, NQ_MAX * sizeof(float), hipMemcpyDeviceToDevice, B));
  const bool filtered = d_fbits && vs->n_rows;
  if (filtered) MSI_TRY(enqueue_filter(vs, d_fbits, nbits, B));
  const uint32_t step = vs->nqt_max * QT;
  static const uint32_t spare = getenv("MSI_VS_SPARE_CUS") ? (uint32_t)std::max(0, atoi(getenv("MSI_VS_SPARE_CUS"))) : 0u;
  Chunk ch[2];
  uint32_t i = 0;
  for (uint32_t q0 = 0; q0 < n_queries; q0 += step, ++i) {
    const uint32_t nq = std::min(step, n_queries - q0);
    const uint32_t set = i & 1;
    if (set) swap_scratch(vs);
    const int32_t pst = plan_chunk(vs, ch[set], d_queries + (size_t)q0 * vs->dim, nq, k, filtered, d_out_docids + (size_t)q0 * k,
                                   d_out_dist + (size_t)q0 * k, d_out_counts + q0, d_inexact ? d_inexact + q0 : nullptr);
    if (set) swap_scratch(vs);
    MSI_TRY(pst);
    // A sweep's workgroup takes a whole CU (its LDS), so nothing of the second stream could run beside it: the sweeps
    // leave `spare` CUs free (one per XCD) and the sample sweep of the next chunk is cut to fit them; selection and
    // rescoring (40 / 19 KB of LDS per workgroup) find room there too.  MSI_VS_SPARE_CUS=0 (default): the sweeps keep every CU.
    if (spare && vs->scan_grid > 4 * spare) {
      ch[set].grid_main = vs->scan_grid - spare;
      ch[set].grid_sample = spare;
    }
    MSI_TRY(chunk_pre(vs, ch[set], B));
    MSI_HIP_TRY(hipEventRecord(vs->ev_pre[set], B));
    if (i >= 1) {
      MSI_HIP_TRY(hipStreamWaitEvent(B, vs->ev_main[set ^ 1], 0));
      MSI_TRY(chunk_post(vs, ch[set ^ 1], B));
    }
    MSI_HIP_TRY(hipStreamWaitEvent(A, vs->ev_pre[set], 0));
    MSI_TRY(chunk_main(vs, ch[set], A));
    MSI_HIP_TRY(hipEventRecord(vs->ev_main[set], A));
  }
  const uint32_t last = (i - 1) & 1;
  MSI_HIP_TRY(hipStreamWaitEvent(B, vs->ev_main[last], 0));
  MSI_TRY(chunk_post(vs, ch[last], B));
  MSI_HIP_TRY(hipEventRecord(vs->ev_done, B));
  MSI_HIP_TRY(hipStreamWaitEvent(A, vs->ev_done, 0));
  vs->pipelined_calls++;
  return MSI_OK;
}

// The levels of effort of a store, cheapest first (msi_vs::level): the int8 candidate sweep when the store keeps the copy;
// the store's f32 contraction (bf16x2 by default) with the usual K'; the same with K' = KP_MAX; bf16x3 with K' = KP_MAX
// (bf16x2 stores).  What the last level cannot prove is answered exhaustively.
struct Level { bool i8, x2, big; };
constexpr uint32_t MAX_LEVELS = 5;
uint32_t vs_levels(const msi_vs *vs, Level out[MAX_LEVELS]) {
  uint32_t n = 0;
  if (vs->i8) {
    out[n++] = Level{true, vs->bf2, false};   // the int8 sweep, K' = k + 1004 ...
    out[n++] = Level{true, vs->bf2, true};    // ... and with K' = KP_MAX for the few queries with more rows inside its bound
  }
  out[n++] = Level{false, vs->bf2, false};
  out[n++] = Level{false, vs->bf2, true};
  if (vs->bf2) out[n++] = Level{false, false, true};
  return n;
}
// MSI_VS_FIRST_LEVEL=<n> | f32 (read per call; measurements and tests): searches start at level n at least; "f32" = the first
// level that sweeps the f32 rows (rounds 1-4's level 0)
uint32_t vs_first_level(const Level *levels, uint32_t n_levels, uint32_t k) {
  const char *e = getenv("MSI_VS_FIRST_LEVEL");
  if (!e) {
    // The int8 level's bound is wide (eps ~ 1.8e-2 in cosine): the rows inside 2 eps of the k-th neighbour number several
    // times k, and K' = k + 3 k candidates is what it is planned with (plan_chunk).  Above KP_MAX / 4 neighbours that no longer
    // fits the KP_MAX candidates a query can rescore — the level would prove nothing and every query would sweep twice (C5's
    // k = 1 000): such searches start at the first level that sweeps the stored rows.
    if (4ull * k > KP_MAX)
      for (uint32_t i = 0; i < n_levels; ++i)
        if (!levels[i].i8) return i;
    return 0u;
  }
  if (!strcmp(e, "f32")) {
    for (uint32_t i = 0; i < n_levels; ++i)
      if (!levels[i].i8) return i;
    return 0u;
  }
  return (uint32_t)std::max(0, atoi(e));
}
uint32_t vs_level_batch(const msi_vs *vs, const Level &l) {
  if (l.i8) return vs->nqt8_max * QT;
  return ((vs->bf2 && !l.x2) ? vs->nqt3_max : vs->nqt_max) * QT;
}

int32_t exhaustive_one(msi_vs *vs, uint32_t qj, uint32_t k, const u64 *d_fbits, uint64_t nbits,
                       uint32_t *d_out_docids, float *d_out_dist, uint32_t *d_out_count) {
  hipStream_t st = vs->ctx->stream;
  Small s = small_of(vs);
  MSI_TRY(vs->exh_keys.ensure(std::max<uint64_t>(1, vs->n_rows) * sizeof(u64)));
  if (vs->n_rows)
    hipLaunchKernelGGL(vs_exhaustive_kernel, dim3((uint32_t)((vs->n_rows + 255) / 256)), dim3(256),
                       (size_t)vs->dpad * sizeof(float), st, vs->tiles.p, vs->norm.as<float>(),
                       vs->docids.as<uint32_t>(), vs->n_rows, vs->KB, vs->qrow.as<float>(), s.qn, qj, d_fbits,
                       nbits, vs->exh_keys.as<u64>(), vs->dpad, vs->s16 ? 1u : 0u);
  hipLaunchKernelGGL(vs_exhaustive_select_kernel, dim3(1), dim3(SEL_THREADS), 0, st, vs->exh_keys.as<u64>(),
                     (uint32_t)vs->n_rows, k, vs->docids.as<uint32_t>(), d_out_docids, d_out_dist, d_out_count);
  MSI_HIP_TRY(hipGetLastError());
  vs->exhaustive_reruns++;
  return MSI_OK;
}

}  // namespace

extern "C" {

int32_t msi_vs_create(msi_ctx *ctx, uint32_t dim, msi_vs **out) {
  return msi_vs_create_typed(ctx, dim, MSI_VS_F32, out);
}

int32_t msi_vs_create_typed(msi_ctx *ctx, uint32_t dim, int32_t storage, msi_vs **out) {
  if (!ctx || !out || dim == 0 || (storage != MSI_VS_F32 && storage != MSI_VS_BF16)) {
    msi_set_error("msi_vs_create: invalid argument");
    return MSI_E_INVALID;
  }
  *out = nullptr;
  const bool s16 = storage == MSI_VS_BF16;
  // a tile is KB blocks of 1 KiB; KB must be a multiple of SCAN_GROUP
  const uint32_t dpad = s16 ? ((dim + 255) / 256) * 256 : ((dim + 127) / 128) * 128;
  const uint32_t KB = s16 ? dpad / 32 : dpad / 16;
  if (scan_lds_bytes(KB, 1, s16) > LDS_MAX) {
    msi_set_error("msi_vs_create: dim %u needs %zu B of LDS for one query tile (max %zu)", dim,
                  scan_lds_bytes(KB, 1, s16), LDS_MAX);
    return MSI_E_UNSUPPORTED;
  }
  // contraction of the fast scan (f32 rows), MSI_VS_SCAN_MATH: "bf16x2" (default: the queries' hi halves only in LDS —
  // twice the queries per sweep, a wider proof margin, more rescored candidates, bf16x3 as second opinion), "bf16x3",
  // "f32"
  const char *math = getenv("MSI_VS_SCAN_MATH");
  const bool bf2 = !s16 && !(math && (strcmp(math, "bf16x3") == 0 || strcmp(math, "f32") == 0));
  const uint32_t nqt_cap = bf2 ? (uint32_t)NQT_F32_MAX : 3u;
  uint32_t nqt_max = 1;
  while (nqt_max < nqt_cap && scan_lds_bytes(KB, nqt_max + 1, s16, bf2) <= LDS_MAX) ++nqt_max;
  // the bf16x3 second opinion keeps hi AND lo halves of the queries in LDS: twice the bytes per query tile, so its
  // capacity is its own (d = 1024: KB = 64 -> 2 tiles of 64 KiB, not 3; d = 1536: 1 tile)
  uint32_t nqt3_max = 1;
  while (nqt3_max < 3u && scan_lds_bytes(KB, nqt3_max + 1, s16, false) <= LDS_MAX) ++nqt3_max;
  if (const char *e = getenv("MSI_VS_MAX_QUERY_TILES")) {  // tuning/testing knob
    const int v = atoi(e);
    if (v >= 1 && (uint32_t)v < nqt_max) nqt_max = (uint32_t)v;
    if (v >= 1 && (uint32_t)v < nqt3_max) nqt3_max = (uint32_t)v;
  }
  DeviceGuard g(ctx->device);
  const void *fns[] = {
#define MSI_F(N, D, B)                                                                     \
  reinterpret_cast<const void *>(&vs_scan_kernel<SCAN_WAVES, N, D, B, false, false>),      \
      reinterpret_cast<const void *>(&vs_scan_kernel<SCAN_WAVES, N, D, B, false, true>)
#define MSI_G(N, D)                                                                        \
  reinterpret_cast<const void *>(&vs_scan_kernel<SCAN_WAVES, N, D, 1, true, false>),       \
      reinterpret_cast<const void *>(&vs_scan_kernel<SCAN_WAVES, N, D, 1, true, true>)
      MSI_F(1, true, true), MSI_F(1, true, false), MSI_F(1, false, true), MSI_F(1, false, false),
      MSI_F(2, true, true), MSI_F(2, true, false), MSI_F(2, false, true), MSI_F(2, false, false),
      MSI_F(3, true, true), MSI_F(3, true, false), MSI_F(3, false, true), MSI_F(3, false, false),
      MSI_G(1, true), MSI_G(1, false), MSI_G(2, true), MSI_G(2, false), MSI_G(3, true), MSI_G(3, false),
      MSI_F(1, true, 2), MSI_F(1, false, 2), MSI_F(2, true, 2), MSI_F(2, false, 2), MSI_F(3, true, 2), MSI_F(3, false, 2),
      MSI_F(4, true, 2), MSI_F(4, false, 2), MSI_F(5, true, 2), MSI_F(5, false, 2), MSI_F(6, true, 2), MSI_F(6, false, 2)
#undef MSI_F
#undef MSI_G
  };
  for (const void *fn : fns) {
    hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_MAX);
    if (e != hipSuccess) {
      msi_set_error("hipFuncSetAttribute(vs_scan) failed: %s", hipGetErrorString(e));
      return MSI_E_HIP;
    }
  }
  msi_vs *vs = new msi_vs();
  vs->ctx = ctx;
  msi_ctx_retain(ctx);
  vs->dim = dim;
  vs->dpad = dpad;
  vs->KB = KB;
  vs->nqt_max = nqt_max;
  vs->nqt3_max = nqt3_max;
  vs->s16 = s16;
  vs->bf3 = !(math && strcmp(math, "f32") == 0);
  vs->bf2 = bf2;
  // workgroups per CU: two when the query fragments leave room (more loads in flight)
  uint32_t wg_per_cu = scan_lds_bytes(KB, nqt_max, s16, bf2) <= LDS_MAX / 2 ? 2 : 1;
  if (const char *e = getenv("MSI_VS_WG_PER_CU")) {
    const int v = atoi(e);
    if (v >= 1 && v <= 4) wg_per_cu = (uint32_t)v;
  }
  vs->scan_grid = (uint32_t)(ctx->n_cu_scan ? ctx->n_cu_scan : ctx->n_cu) * wg_per_cu;
  // the int8 copy (f32 stores): MSI_VS_I8=0 keeps the store without it; MSI_VS_I8_QUERY_TILES caps the queries per sweep
  {
    const char *e8 = getenv("MSI_VS_I8");
    // (round 6: bf16 stores keep the copy too — half the bytes of their rows per sweep; MSI_VS_I8_BF16=0: only f32 stores)
    const char *e16 = getenv("MSI_VS_I8_BF16");
    vs->i8 = !(s16 && e16 && e16[0] == '0') && !(e8 && e8[0] == '0') && (dpad % 64) == 0 && (uint64_t)dpad * 127ull * 127ull < (1ull << 31);
    if (vs->i8) {
      vs->KB8 = dpad / 64;
      // 8 query tiles (128 queries) per sweep by default: with 12 the accumulators (2 row tiles x 12 x 4 registers) and the
      // row buffers no longer fit the 256 registers a wave of a 512-thread workgroup has (139 spilled in the sparse epilogue)
      uint32_t cap8 = 8;
      if (const char *e = getenv("MSI_VS_I8_QUERY_TILES")) cap8 = (uint32_t)std::max(1, std::min<int>(NQT_MAX, atoi(e)));
      vs->nqt8_max = 1;
      while (vs->nqt8_max < cap8 && scan8_lds_bytes(vs->KB8, scan8_nqt_of(vs->nqt8_max + 1)) <= LDS_MAX) ++vs->nqt8_max;
      if (scan8_lds_bytes(vs->KB8, 1) > LDS_MAX) vs->i8 = false;
    }
    if (vs->i8) {
      const uint32_t gs = scan8_gs_of(vs->KB8);
      const int32_t st8 = gs == 4 ? scan8_set_attributes<4>() : (gs == 3 ? scan8_set_attributes<3>() : scan8_set_attributes<2>());
      if (st8 != MSI_OK) {
        msi_set_error("hipFuncSetAttribute(vs_scan_i8) failed");
        msi_ctx_release(ctx);
        delete vs;
        return st8;
      }
      uint32_t wg8 = scan8_lds_bytes(vs->KB8, scan8_nqt_of(vs->nqt8_max)) <= LDS_MAX / 2 ? 2 : 1;
      if (const char *e = getenv("MSI_VS_WG_PER_CU")) {
        const int v = atoi(e);
        if (v >= 1 && v <= 4) wg8 = (uint32_t)v;
      }
      vs->scan_grid8 = (uint32_t)(ctx->n_cu_scan ? ctx->n_cu_scan : ctx->n_cu) * wg8;
    }
  }
  *out = vs;
  return MSI_OK;
}

void msi_vs_destroy(msi_vs *vs) {
  if (!vs) return;
  msi_ctx *ctx = vs->ctx;
  {
    std::lock_guard<std::mutex> lk(ctx->mu);
    DeviceGuard g(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    DevBuf *bufs[] = {&vs->tiles, &vs->norm, &vs->inv_norm, &vs->docids, &vs->qraw, &vs->qfrag, &vs->qfrag_bf, &vs->qrow,
                      &vs->qsmall, &vs->gkeys, &vs->gcnt, &vs->gsmall, &vs->sel_keys, &vs->dense, &vs->frows,
                      &vs->fbits, &vs->out_docids, &vs->out_dist, &vs->exh_keys, &vs->rowtmp, &vs->resc_keys,
                      &vs->tiles_next, &vs->docids_next, &vs->norm_next, &vs->inv_norm_next, &vs->add_tiles, &vs->add_docids, &vs->row_map,
                      &vs->tiles8, &vs->scale8, &vs->i8small, &vs->qfrag8, &vs->rerun_q, &vs->rerun_flags, &vs->scr2.qfrag8};
    for (DevBuf *b : bufs) b->release();
    DevBuf *bufs2[] = {&vs->scr2.qfrag, &vs->scr2.qfrag_bf, &vs->scr2.qrow, &vs->scr2.qsmall, &vs->scr2.gkeys, &vs->scr2.gcnt,
                       &vs->scr2.gsmall, &vs->scr2.sel_keys, &vs->scr2.dense, &vs->scr2.resc_keys, &vs->fsmall};
    for (DevBuf *b : bufs2) b->release();
    if (vs->aux_stream) {
      (void)hipStreamSynchronize(vs->aux_stream);
      (void)hipStreamDestroy(vs->aux_stream);
    }
    for (hipEvent_t e : {vs->ev_start, vs->ev_done, vs->ev_pre[0], vs->ev_pre[1], vs->ev_main[0], vs->ev_main[1]})
      if (e) (void)hipEventDestroy(e);
    vs->scan_timer.release();
    delete vs;
  }
  msi_ctx_release(ctx);
}

int32_t msi_vs_upload(msi_vs *vs, const uint32_t *docids, const float *rows, uint64_t n_rows) {
  if (!vs || (n_rows && (!docids || !rows))) {
    msi_set_error("msi_vs_upload: invalid argument");
    return MSI_E_INVALID;
  }
  std::lock_guard<std::mutex> lk(vs->ctx->mu);
  DeviceGuard g(vs->ctx->device);
  return upload_common(vs, docids, false, rows, false, n_rows);
}

int32_t msi_vs_upload_device(msi_vs *vs, const uint32_t *d_docids, const float *d_rows, uint64_t n_rows) {
  if (!vs || (n_rows && (!d_docids || !d_rows))) {
    msi_set_error("msi_vs_upload_device: invalid argument");
    return MSI_E_INVALID;
  }
  std::lock_guard<std::mutex> lk(vs->ctx->mu);
  DeviceGuard g(vs->ctx->device);
  return upload_common(vs, d_docids, true, d_rows, true, n_rows);
}

// SURVEY §8 f2 — what a committed update does to a store (update/new/indexer/write.rs:65-74,157: del_item /
// add_item per document, then a rebuild of the ANN structure) without sending the store over PCIe again: only the
// docid lists and the ADDED rows travel; the device tiles the added rows, then re-gathers the whole store into its
// next buffer in one pass (every 16-byte slot moves as it is), recomputes the norms and swaps.
int32_t msi_vs_update(msi_vs *vs, const uint32_t *remove_docids, uint64_t n_remove, const uint32_t *add_docids,
                      const float *add_rows, uint64_t n_add) {
  if (!vs || (n_remove && !remove_docids) || (n_add && (!add_docids || !add_rows))) {
    msi_set_error("msi_vs_update: invalid argument");
    return MSI_E_INVALID;
  }
  for (uint64_t i = 1; i < n_remove; ++i)
    if (remove_docids[i - 1] >= remove_docids[i]) {
      msi_set_error("msi_vs_update: remove_docids must be strictly ascending");
      return MSI_E_NOT_SORTED;
    }
  for (uint64_t i = 1; i < n_add; ++i)
    if (add_docids[i - 1] >= add_docids[i]) {
      msi_set_error("msi_vs_update: add_docids must be strictly ascending");
      return MSI_E_NOT_SORTED;
    }
  std::lock_guard<std::mutex> lk(vs->ctx->mu);
  DeviceGuard g(vs->ctx->device);
  hipStream_t st = vs->ctx->stream;
  // the next store's rows in docid order: an old row that is neither removed nor replaced, or an added row
  const std::vector<uint32_t> &old = vs->h_docids;
  std::vector<uint32_t> map;
  map.reserve(old.size() + n_add);
  uint64_t io = 0, ir = 0, ia = 0;
  while (io < old.size() || ia < n_add) {
    const bool take_add = ia < n_add && (io >= old.size() || add_docids[ia] <= old[io]);
    if (take_add) {
      if (io < old.size() && add_docids[ia] == old[io]) ++io;  // replaced
      map.push_back(0x80000000u | (uint32_t)ia);
      ++ia;
      continue;
    }
    while (ir < n_remove && remove_docids[ir] < old[io]) ++ir;
    if (!(ir < n_remove && remove_docids[ir] == old[io])) map.push_back((uint32_t)io);
    ++io;
  }
  const uint64_t n_new = map.size();
  if (n_new > 0x7FFFFFF0ull || n_add > 0x7FFFFFF0ull) {
    msi_set_error("msi_vs_update: %llu rows exceed the row index space of an update", (unsigned long long)n_new);
    return MSI_E_UNSUPPORTED;
  }
  const uint64_t n_tiles = (n_new + 15) / 16, padded = n_tiles * 16, add_tiles = (n_add + 15) / 16;
  const size_t slot = sizeof(uint4);
  MSI_TRY(vs->tiles_next.ensure(std::max<uint64_t>(1, n_tiles) * vs->KB * 64 * slot));
  MSI_TRY(vs->docids_next.ensure(std::max<uint64_t>(16, padded) * sizeof(uint32_t)));
  MSI_TRY(vs->add_tiles.ensure(std::max<uint64_t>(1, add_tiles) * vs->KB * 64 * slot));
  MSI_TRY(vs->add_docids.ensure(std::max<uint64_t>(1, n_add) * sizeof(uint32_t)));
  MSI_TRY(vs->row_map.ensure(std::max<uint64_t>(1, n_new) * sizeof(uint32_t)));
  // (the live store's norm arrays are not touched before the swap: a buffer that grows is a new allocation, and a
  // failure further down must leave the current store whole)
  MSI_TRY(vs->norm_next.ensure(std::max<uint64_t>(16, padded) * sizeof(float)));
  MSI_TRY(vs->inv_norm_next.ensure(std::max<uint64_t>(16, padded) * sizeof(float)));
  MSI_TRY(ensure_scratch(vs));
  if (n_add) {
    MSI_TRY(tile_rows(vs, add_rows, false, n_add, vs->add_tiles.p));
    MSI_HIP_TRY(hipMemcpyAsync(vs->add_docids.p, add_docids, n_add * sizeof(uint32_t), hipMemcpyHostToDevice, st));
  }
  if (n_new) MSI_HIP_TRY(hipMemcpyAsync(vs->row_map.p, map.data(), n_new * sizeof(uint32_t), hipMemcpyHostToDevice, st));
  MSI_HIP_TRY(hipMemsetAsync(vs->docids_next.p, 0xFF, std::max<uint64_t>(16, padded) * sizeof(uint32_t), st));
  const uint64_t total = n_tiles * vs->KB * 64;
  if (total)
    hipLaunchKernelGGL(vs_regather_kernel, dim3((uint32_t)((total + 255) / 256)), dim3(256), 0, st,
                       vs->tiles.as<uint4>(), vs->add_tiles.as<uint4>(), vs->docids.as<uint32_t>(),
                       vs->add_docids.as<uint32_t>(), vs->row_map.as<uint32_t>(), n_new, vs->KB,
                       vs->tiles_next.as<uint4>(), vs->docids_next.as<uint32_t>());
  MSI_HIP_TRY(hipGetLastError());
  MSI_HIP_TRY(hipStreamSynchronize(st));  // `map` and the borrowed lists are done with; searches hold the same lock
  std::swap(vs->tiles, vs->tiles_next);
  std::swap(vs->docids, vs->docids_next);
  std::swap(vs->norm, vs->norm_next);
  std::swap(vs->inv_norm, vs->inv_norm_next);
  Small s = small_of(vs);
  int32_t fin = hipMemsetAsync(s.bad, 0, sizeof(uint32_t), st) == hipSuccess ? MSI_OK : MSI_E_HIP;
  if (fin == MSI_OK) fin = finish_upload(vs, n_new, "msi_vs_update");
  if (fin != MSI_OK) {   // the new store has no valid norms: better empty than wrong
    vs->n_rows = 0;
    vs->n_tiles = 0;
    vs->h_docids.clear();
  }
  return fin;
}

uint64_t msi_vs_len(const msi_vs *vs) { return vs ? vs->n_rows : 0; }
uint32_t msi_vs_dim(const msi_vs *vs) { return vs ? vs->dim : 0; }
uint32_t msi_vs_max_batch(const msi_vs *vs) { return vs ? (vs->i8 ? vs->nqt8_max : vs->nqt_max) * QT : 0; }

int32_t msi_vs_get_vector(msi_vs *vs, uint32_t docid, float *out_row, int32_t *out_found) {
  if (!vs || !out_row || !out_found) {
    msi_set_error("msi_vs_get_vector: invalid argument");
    return MSI_E_INVALID;
  }
  std::lock_guard<std::mutex> lk(vs->ctx->mu);
  DeviceGuard g(vs->ctx->device);
  auto it = std::lower_bound(vs->h_docids.begin(), vs->h_docids.end(), docid);
  if (it == vs->h_docids.end() || *it != docid) {
    *out_found = 0;
    return MSI_OK;
  }
  const uint32_t row = (uint32_t)(it - vs->h_docids.begin());
  hipStream_t st = vs->ctx->stream;
  MSI_TRY(vs->qraw.ensure((size_t)NQ_MAX * vs->dim * sizeof(float)));
  hipLaunchKernelGGL(vs_gather_row_kernel, dim3(ceil_div_u32(vs->dim, 256)), dim3(256), 0, st,
                     vs->tiles.p, vs->KB, row, vs->dim, vs->qraw.as<float>(), vs->s16 ? 1u : 0u);
  MSI_HIP_TRY(hipMemcpyAsync(out_row, vs->qraw.p, vs->dim * sizeof(float), hipMemcpyDeviceToHost, st));
  MSI_HIP_TRY(hipStreamSynchronize(st));
  *out_found = 1;
  return MSI_OK;
}

// VectorStore::nns_by_item for one store (store.rs:615-637,980-1034): the query is the
// item's own stored vector; Similar::execute (search/similar.rs:67-153) passes a filter
// that excludes the item itself.
int32_t msi_vs_search_by_item(msi_vs *vs, uint32_t docid, uint32_t k, const uint64_t *filter_bits,
                              uint64_t filter_nbits, uint32_t *out_docids, float *out_dist, uint32_t *out_count,
                              int32_t *out_found) {
  if (!vs || !out_count || !out_found || (k && (!out_docids || !out_dist))) {
    msi_set_error("msi_vs_search_by_item: invalid argument");
    return MSI_E_INVALID;
  }
  std::vector<float> v(vs->dim);
  MSI_TRY(msi_vs_get_vector(vs, docid, v.data(), out_found));
  *out_count = 0;
  if (!*out_found) return MSI_OK;   // the item has no vector in this store
  return msi_vs_search(vs, v.data(), 1, k, filter_bits, filter_nbits, nullptr, out_docids, out_dist, out_count);
}

int32_t msi_vs_search_device(msi_vs *vs, const float *d_queries, uint32_t n_queries, uint32_t k,
                             const uint64_t *d_filter_bits, uint64_t filter_nbits, uint32_t *d_out_docids,
                             float *d_out_dist, uint32_t *d_out_counts, uint32_t *d_inexact) {
  if (!vs || !d_queries || n_queries == 0 || !d_out_docids || !d_out_dist || !d_out_counts) {
    msi_set_error("msi_vs_search_device: invalid argument");
    return MSI_E_INVALID;
  }
  std::lock_guard<std::mutex> lk(vs->ctx->mu);
  DeviceGuard g(vs->ctx->device);
  if (vs->n_rows == 0 || k == 0) {
    MSI_HIP_TRY(hipMemsetAsync(d_out_counts, 0, n_queries * sizeof(uint32_t), vs->ctx->stream));
    if (d_inexact) MSI_HIP_TRY(hipMemsetAsync(d_inexact, 0, n_queries * sizeof(uint32_t), vs->ctx->stream));
    return MSI_OK;
  }
  // MSI_VS_PIPELINE=1: the chunks' stages on two streams (search_device_pipelined, the store's f32 contraction only).  Off by
  // default — measured at C4 (768 queries, 8 sweeps): 42.25 ms against 42.05 ms on one stream; a sweep's workgroup holds its
  // CU's whole LDS, so the second stream's kernels wait for the sweep anyway (profiles/r4_vs_pipeline.txt)
  const char *pipe_knob = getenv("MSI_VS_PIPELINE");   // (read per call: tests switch it)
  const bool pipeline_on = pipe_knob && pipe_knob[0] == '1';
  // The pipeline answers as the OLD contract does — asynchronous, f32 contraction only, unproven queries merely flagged in
  // d_inexact — so it is taken only when the caller asked for that contract (MSI_VS_DEVICE_RERUN=0): under the default
  // contract ("every query is answered, d_inexact reads 0") a caller would otherwise be handed unproven lists it has been
  // told it may trust.
  {
    const char *rr0 = getenv("MSI_VS_DEVICE_RERUN");
    const bool old_contract = rr0 && rr0[0] == '0';
    if (n_queries > vs->nqt_max * QT && pipeline_on && old_contract)
      return search_device_pipelined(vs, d_queries, n_queries, k, (const u64 *)d_filter_bits, filter_nbits, d_out_docids, d_out_dist,
                                     d_out_counts, d_inexact);
  }
  // Levels of effort, as the host entry point runs them (vs_search_direct) — and since round 5 this entry point ALWAYS
  // ANSWERS too (store.rs:638-675 does): the first pass sweeps every chunk at the store's current level; the queries it
  // could not prove are gathered and re-run level by level, then exhaustively, by the library itself.  That costs one
  // synchronisation of the context's stream per call (the flags have to be read); MSI_VS_DEVICE_RERUN=0 restores the old
  // contract (asynchronous, unproven queries only reported through d_inexact).
  static const bool adapt = !(getenv("MSI_VS_ADAPT") && getenv("MSI_VS_ADAPT")[0] == '0');
  const char *rr = getenv("MSI_VS_DEVICE_RERUN");
  const bool rerun = !(rr && rr[0] == '0');
  hipStream_t st = vs->ctx->stream;
  const bool store_x2 = vs->bf2;
  Level levels[MAX_LEVELS];
  const uint32_t n_levels = vs_levels(vs, levels);
  const uint32_t l0 = std::min(std::max(adapt ? vs->level : 0u, vs_first_level(levels, n_levels, k)), n_levels - 1);
  auto with_level = [&](const Level &l, auto &&fn) -> int32_t {
    vs->bf2 = l.x2;
    vs->big_slack = l.big;
    vs->i8_now = l.i8;
    const int32_t r = fn();
    vs->bf2 = store_x2;
    vs->big_slack = false;
    vs->i8_now = false;
    return r;
  };
  uint32_t *flags = d_inexact;
  if (rerun) {
    MSI_TRY(vs->rerun_flags.ensure((size_t)n_queries * sizeof(uint32_t)));
    flags = vs->rerun_flags.as<uint32_t>();
  }
  const uint32_t step = vs_level_batch(vs, levels[l0]);
  for (uint32_t q0 = 0; q0 < n_queries; q0 += step) {
    const uint32_t nq = std::min(step, n_queries - q0);
    MSI_TRY(with_level(levels[l0], [&] {
      return enqueue_search(vs, d_queries + (size_t)q0 * vs->dim, nq, k, (const u64 *)d_filter_bits, filter_nbits,
                            d_out_docids + (size_t)q0 * k, d_out_dist + (size_t)q0 * k, d_out_counts + q0, flags ? flags + q0 : nullptr);
    }));
    ++vs->level_sweeps[l0];
  }
  if (!rerun) return MSI_OK;
  std::vector<uint32_t> h_flags(n_queries);
  MSI_HIP_TRY(hipMemcpyAsync(h_flags.data(), flags, (size_t)n_queries * sizeof(uint32_t), hipMemcpyDeviceToHost, st));
  MSI_HIP_TRY(hipStreamSynchronize(st));
  if (d_inexact) MSI_HIP_TRY(hipMemsetAsync(d_inexact, 0, (size_t)n_queries * sizeof(uint32_t), st));   // every query is answered
  std::vector<uint32_t> pending;
  for (uint32_t j = 0; j < n_queries; ++j)
    if (h_flags[j]) pending.push_back(j);
  if (adapt && n_queries >= (uint32_t)QT) {   // the same running share as the host entry point keeps
    vs->level_ema = 0.5f * vs->level_ema + 0.5f * (float)pending.size() / (float)n_queries;
    if (vs->level_left > 0 && --vs->level_left == 0 && vs->level > 0) {
      --vs->level;
      vs->level_ema = 0.0f;
    } else if (vs->level_ema > 0.25f && vs->level + 1 < n_levels) {
      ++vs->level;
      vs->level_left = 256;
      vs->level_ema = 0.0f;
    } else if (vs->level_ema > 0.25f) {
      vs->level_left = 256;
    }
  }
  if (pending.empty()) return MSI_OK;
  vs->device_rerun_queries += pending.size();
  const uint32_t kk = std::max<uint32_t>(1, k);
  MSI_TRY(vs->out_docids.ensure((size_t)NQ_MAX * kk * sizeof(uint32_t)));
  MSI_TRY(vs->out_dist.ensure((size_t)NQ_MAX * kk * sizeof(float)));
  MSI_TRY(vs->rerun_q.ensure((size_t)NQ_MAX * vs->dim * sizeof(float)));
  Small s = small_of(vs);
  // (a first pass that already ran at the last level is repeated at it for the pending queries: the exhaustive pass reads
  // the query rows the sweep before it prepared)
  for (uint32_t lvl = std::min(l0 + 1, n_levels - 1); lvl < n_levels && !pending.empty(); ++lvl) {
    const uint32_t sub = vs_level_batch(vs, levels[lvl]);
    std::vector<uint32_t> next;
    for (size_t f0 = 0; f0 < pending.size(); f0 += sub) {
      const uint32_t nf = (uint32_t)std::min<size_t>(sub, pending.size() - f0);
      for (uint32_t i = 0; i < nf; ++i)
        MSI_HIP_TRY(hipMemcpyAsync(vs->rerun_q.as<float>() + (size_t)i * vs->dim, d_queries + (size_t)pending[f0 + i] * vs->dim,
                                   (size_t)vs->dim * sizeof(float), hipMemcpyDeviceToDevice, st));
      MSI_TRY(with_level(levels[lvl], [&] {
        return enqueue_search(vs, vs->rerun_q.as<float>(), nf, k, (const u64 *)d_filter_bits, filter_nbits,
                              vs->out_docids.as<uint32_t>(), vs->out_dist.as<float>(), s.counts, s.inexact);
      }));
      ++vs->level_sweeps[lvl];
      uint32_t f2[NQ_MAX];
      MSI_HIP_TRY(hipMemcpyAsync(f2, s.inexact, nf * sizeof(uint32_t), hipMemcpyDeviceToHost, st));
      MSI_HIP_TRY(hipStreamSynchronize(st));
      for (uint32_t i = 0; i < nf; ++i) {
        const uint32_t j = pending[f0 + i];
        if (f2[i]) {
          if (lvl + 1 == n_levels) {
            MSI_TRY(exhaustive_one(vs, i, k, (const u64 *)d_filter_bits, filter_nbits, d_out_docids + (size_t)j * k,
                                   d_out_dist + (size_t)j * k, d_out_counts + j));
          } else {
            next.push_back(j);
          }
          continue;
        }
        MSI_HIP_TRY(hipMemcpyAsync(d_out_docids + (size_t)j * k, vs->out_docids.as<uint32_t>() + (size_t)i * k, (size_t)k * sizeof(uint32_t),
                                   hipMemcpyDeviceToDevice, st));
        MSI_HIP_TRY(hipMemcpyAsync(d_out_dist + (size_t)j * k, vs->out_dist.as<float>() + (size_t)i * k, (size_t)k * sizeof(float),
                                   hipMemcpyDeviceToDevice, st));
        MSI_HIP_TRY(hipMemcpyAsync(d_out_counts + j, s.counts + i, sizeof(uint32_t), hipMemcpyDeviceToDevice, st));
      }
      MSI_HIP_TRY(hipStreamSynchronize(st));   // (the staging rows and result buffers are reused by the next batch)
    }
    pending.swap(next);
  }
  return MSI_OK;
}

}  // extern "C"

static int32_t vs_search_direct(msi_vs *vs, const float *queries, uint32_t n_queries, uint32_t k,
                                const uint64_t *filter_bits, uint64_t filter_nbits,
                                const volatile int32_t *cancel, uint32_t *out_docids, float *out_dist,
                                uint32_t *out_counts);

// Micro-batcher (SURVEY §8 b: callers are up to 4 x cores tokio spawn_blocking threads,
// crates/meilisearch/src/search/federated/perform.rs:224).  The first caller to arrive
// becomes the leader: it waits until a full sweep worth of queries is queued or
// `microbatch_wait_us` elapsed, runs ONE search for everybody with the largest k
// (a top-k list is a prefix of the top-k' list for k <= k', so every caller gets its exact
// answer) and hands the rows out.
static int32_t vs_search_fused(msi_vs *vs, const float *queries, uint32_t n_queries, uint32_t k,
                               uint32_t *out_docids, float *out_dist, uint32_t *out_counts) {
  msi_vs::Pending me;
  me.queries = queries;
  me.n = n_queries;
  me.k = k;
  me.out_docids = out_docids;
  me.out_dist = out_dist;
  me.out_counts = out_counts;
  std::unique_lock<std::mutex> lk(vs->bmu);
  vs->bqueue.push_back(&me);
  vs->bqueued_queries += n_queries;
  const uint32_t full = vs->nqt_max * QT;
  if (vs->bleader_active) {
    vs->bcv.notify_all();  // the leader may now have a full sweep
    vs->bcv.wait(lk, [&] { return me.done || !vs->bleader_active; });
    if (me.done) {
      if (me.status != MSI_OK) msi_set_error("%s", me.error.c_str());
      return me.status;
    }
    // the previous leader left without taking this request: lead the next batch
  }
  vs->bleader_active = true;
  const auto deadline = std::chrono::steady_clock::now() + std::chrono::microseconds(vs->microbatch_wait_us);
  vs->bcv.wait_until(lk, deadline, [&] { return vs->bqueued_queries >= full; });
  std::vector<msi_vs::Pending *> batch;
  batch.swap(vs->bqueue);
  vs->bqueued_queries = 0;
  lk.unlock();
  // one fused search
  uint32_t total = 0, kmax = 0;
  for (auto *p : batch) {
    total += p->n;
    kmax = std::max(kmax, p->k);
  }
  std::vector<float> q((size_t)total * vs->dim);
  std::vector<uint32_t> d((size_t)total * std::max(1u, kmax)), c(total);
  std::vector<float> s((size_t)total * std::max(1u, kmax));
  size_t off = 0;
  for (auto *p : batch) {
    memcpy(q.data() + off * vs->dim, p->queries, (size_t)p->n * vs->dim * sizeof(float));
    off += p->n;
  }
  const int32_t st = vs_search_direct(vs, q.data(), total, kmax, nullptr, 0, nullptr, d.data(), s.data(), c.data());
  const std::string err = st == MSI_OK ? std::string() : std::string(msi_last_error());
  off = 0;
  for (auto *p : batch) {
    if (st == MSI_OK) {
      for (uint32_t j = 0; j < p->n; ++j) {
        const uint32_t cnt = std::min(c[off + j], p->k);
        p->out_counts[j] = cnt;
        if (p->k) {
          memcpy(p->out_docids + (size_t)j * p->k, d.data() + (off + j) * kmax, (size_t)cnt * sizeof(uint32_t));
          memcpy(p->out_dist + (size_t)j * p->k, s.data() + (off + j) * kmax, (size_t)cnt * sizeof(float));
        }
      }
    }
    off += p->n;
  }
  lk.lock();
  vs->fused_calls += batch.size();
  vs->fused_sweeps += (total + full - 1) / full;
  for (auto *p : batch) {
    p->status = st;
    p->error = err;
    p->done = true;
  }
  vs->bleader_active = false;
  lk.unlock();
  vs->bcv.notify_all();
  if (st != MSI_OK) msi_set_error("%s", err.c_str());
  return st;
}

extern "C" {

int32_t msi_merge_topk_device(msi_ctx *ctx, const uint32_t *d_docids, const float *d_dist,
                              const uint32_t *d_counts, uint32_t n_lists, uint32_t n_queries, uint32_t k,
                              uint32_t *d_out_docids, float *d_out_dist, uint32_t *d_out_counts) {
  if (!ctx || !d_docids || !d_dist || !d_counts || !d_out_docids || !d_out_dist || !d_out_counts || k == 0 ||
      n_lists == 0) {
    msi_set_error("msi_merge_topk_device: invalid argument");
    return MSI_E_INVALID;
  }
  if ((uint64_t)n_lists * k > SEL_SORTCAP) {
    msi_set_error("msi_merge_topk_device: n_lists*k = %llu above %d (merge on the host: msi_merge_topk)",
                  (unsigned long long)n_lists * k, SEL_SORTCAP);
    return MSI_E_UNSUPPORTED;
  }
  if (n_queries == 0) return MSI_OK;
  std::lock_guard<std::mutex> lk(ctx->mu);
  DeviceGuard g(ctx->device);
  hipLaunchKernelGGL(vs_merge_lists_kernel, dim3(n_queries), dim3(SEL_THREADS), 0, ctx->stream, d_docids, d_dist,
                     d_counts, n_lists, n_queries, k, d_out_docids, d_out_dist, d_out_counts);
  MSI_HIP_TRY(hipGetLastError());
  return MSI_OK;
}

int32_t msi_vs_set_sweep_split(msi_vs *vs, uint32_t n) {
  if (!vs || n == 0 || n > 64) {
    msi_set_error("msi_vs_set_sweep_split: n must be in 1..64");
    return MSI_E_INVALID;
  }
  std::lock_guard<std::mutex> lk(vs->ctx->mu);
  vs->sweep_split = n;
  return MSI_OK;
}

int32_t msi_vs_set_microbatch(msi_vs *vs, uint32_t max_wait_us) {
  if (!vs) return MSI_E_INVALID;
  std::lock_guard<std::mutex> lk(vs->bmu);
  vs->microbatch_wait_us = max_wait_us;
  return MSI_OK;
}

int32_t msi_vs_microbatch_stats(msi_vs *vs, uint64_t *out_fused_calls, uint64_t *out_fused_sweeps) {
  if (!vs || !out_fused_calls || !out_fused_sweeps) return MSI_E_INVALID;
  std::lock_guard<std::mutex> lk(vs->bmu);
  *out_fused_calls = vs->fused_calls;
  *out_fused_sweeps = vs->fused_sweeps;
  return MSI_OK;
}

int32_t msi_vs_search(msi_vs *vs, const float *queries, uint32_t n_queries, uint32_t k,
                      const uint64_t *filter_bits, uint64_t filter_nbits, const volatile int32_t *cancel,
                      uint32_t *out_docids, float *out_dist, uint32_t *out_counts) {
  if (!vs || (n_queries && (!queries || !out_counts)) || (n_queries && k && (!out_docids || !out_dist))) {
    msi_set_error("msi_vs_search: invalid argument");
    return MSI_E_INVALID;
  }
  if (k > KP_MAX) {
    msi_set_error("msi_vs_search: k=%u above the supported maximum %u", k, KP_MAX);
    return MSI_E_UNSUPPORTED;
  }
  // unfiltered, uncancellable small requests may share a sweep with concurrent callers
  if (vs->microbatch_wait_us && !filter_bits && !cancel && n_queries && n_queries < vs->nqt_max * QT && k)
    return vs_search_fused(vs, queries, n_queries, k, out_docids, out_dist, out_counts);
  return vs_search_direct(vs, queries, n_queries, k, filter_bits, filter_nbits, cancel, out_docids, out_dist,
                          out_counts);
}

}  // extern "C"

static int32_t vs_search_direct(msi_vs *vs, const float *queries, uint32_t n_queries, uint32_t k,
                                const uint64_t *filter_bits, uint64_t filter_nbits,
                                const volatile int32_t *cancel, uint32_t *out_docids, float *out_dist,
                                uint32_t *out_counts) {
  if (k > KP_MAX) {
    msi_set_error("msi_vs_search: k=%u above the supported maximum %u", k, KP_MAX);
    return MSI_E_UNSUPPORTED;
  }
  std::lock_guard<std::mutex> lk(vs->ctx->mu);
  DeviceGuard g(vs->ctx->device);
  hipStream_t st = vs->ctx->stream;
  Small s = small_of(vs);
  const u64 *d_fbits = nullptr;
  if (filter_bits) {
    const size_t words = (size_t)((filter_nbits + 63) / 64);
    MSI_TRY(vs->fbits.ensure(std::max<size_t>(1, words) * sizeof(u64)));
    if (words) MSI_HIP_TRY(hipMemcpyAsync(vs->fbits.p, filter_bits, words * sizeof(u64), hipMemcpyHostToDevice, st));
    d_fbits = vs->fbits.as<u64>();
  }
  const uint32_t kk = std::max<uint32_t>(1, k);
  MSI_TRY(vs->out_docids.ensure((size_t)NQ_MAX * kk * sizeof(uint32_t)));
  MSI_TRY(vs->out_dist.ensure((size_t)NQ_MAX * kk * sizeof(float)));
  static const bool adapt = !(getenv("MSI_VS_ADAPT") && getenv("MSI_VS_ADAPT")[0] == '0');
  // the levels of effort of this store (msi_vs::level, vs_levels)
  const bool store_x2 = vs->bf2;
  Level levels[MAX_LEVELS];
  const uint32_t n_levels = vs_levels(vs, levels);
  auto batch_of = [&](const Level &l) { return vs_level_batch(vs, l); };
  // one sweep at level `l` for the nf queries whose rows are already in vs->qraw: results in vs->out_*, flags / counts in h_*
  auto sweep = [&](const Level &l, uint32_t nf, uint32_t *h_flags, uint32_t *h_counts) -> int32_t {
    vs->bf2 = l.x2;
    vs->big_slack = l.big;
    vs->i8_now = l.i8;
    const int32_t st1 = enqueue_search(vs, vs->qraw.as<float>(), nf, k, d_fbits, filter_nbits, vs->out_docids.as<uint32_t>(),
                                       vs->out_dist.as<float>(), s.counts, s.inexact);
    vs->bf2 = store_x2;
    vs->big_slack = false;
    vs->i8_now = false;
    MSI_TRY(st1);
    MSI_HIP_TRY(hipMemcpyAsync(h_flags, s.inexact, nf * sizeof(uint32_t), hipMemcpyDeviceToHost, st));
    MSI_HIP_TRY(hipMemcpyAsync(h_counts, s.counts, nf * sizeof(uint32_t), hipMemcpyDeviceToHost, st));
    MSI_HIP_TRY(hipStreamSynchronize(st));
    return MSI_OK;
  };
  uint32_t step = 0;
  for (uint32_t q0 = 0; q0 < n_queries; q0 += step) {
    if (cancel && *cancel) {
      msi_set_error("msi_vs_search: cancelled");
      return MSI_E_CANCELLED;
    }
    const uint32_t l0 = std::min(std::max(adapt ? vs->level : 0u, vs_first_level(levels, n_levels, k)), n_levels - 1);
    step = batch_of(levels[l0]);
    const uint32_t nq = std::min<uint32_t>(step, n_queries - q0);
    if (k == 0 || vs->n_rows == 0) {
      for (uint32_t j = 0; j < nq; ++j) out_counts[q0 + j] = 0;
      continue;
    }
    MSI_HIP_TRY(hipMemcpyAsync(vs->qraw.p, queries + (size_t)q0 * vs->dim, (size_t)nq * vs->dim * sizeof(float),
                               hipMemcpyHostToDevice, st));
    uint32_t h_flags[NQ_MAX], h_counts[NQ_MAX];
    MSI_TRY(sweep(levels[l0], nq, h_flags, h_counts));
    ++vs->level_sweeps[l0];
    if (levels[l0].i8) {}
    else if (levels[l0].x2) ++vs->x2_sweeps;
    else if (store_x2) ++vs->x3_first_sweeps;
    // what this sweep proved leaves now (the re-runs below prepare their own query rows and reuse the result buffers)
    MSI_HIP_TRY(hipMemcpyAsync(out_docids + (size_t)q0 * k, vs->out_docids.p, (size_t)nq * k * sizeof(uint32_t),
                               hipMemcpyDeviceToHost, st));
    MSI_HIP_TRY(hipMemcpyAsync(out_dist + (size_t)q0 * k, vs->out_dist.p, (size_t)nq * k * sizeof(float),
                               hipMemcpyDeviceToHost, st));
    MSI_HIP_TRY(hipStreamSynchronize(st));
    std::vector<uint32_t> pending;
    for (uint32_t j = 0; j < nq; ++j) {
      out_counts[q0 + j] = h_counts[j];
      if (h_flags[j]) pending.push_back(j);
    }
    if (adapt && nq >= (uint32_t)QT) {   // (a sweep of at least one query tile says something about the data)
      vs->level_ema = 0.5f * vs->level_ema + 0.5f * (float)pending.size() / (float)nq;
      if (vs->level_left > 0 && --vs->level_left == 0 && vs->level > 0) {
        --vs->level;                     // the probing sweep a level down comes next
        vs->level_ema = 0.0f;
      } else if (vs->level_ema > 0.25f && vs->level + 1 < n_levels) {
        ++vs->level;
        vs->level_left = 256;
        vs->level_ema = 0.0f;
      } else if (vs->level_ema > 0.25f) {
        vs->level_left = 256;            // (already at the last level: stay)
      }
    }
    if (!levels[l0].i8 && levels[l0].x2 && !levels[l0].big) vs->second_opinion_queries += pending.size();
    // What the sweep that JUST ran (its query rows are still prepared in vs->qrow, row i = query idx[i] of this chunk) could
    // not prove and no further level can: answered exhaustively in the reference arithmetic, results straight to the caller.
    auto exhaustive_now = [&](const std::vector<uint32_t> &rows_i, const std::vector<uint32_t> &idx) -> int32_t {
      for (size_t t = 0; t < rows_i.size(); ++t) {
        if (cancel && *cancel) {
          msi_set_error("msi_vs_search: cancelled");
          return MSI_E_CANCELLED;
        }
        const uint32_t i = rows_i[t], j = idx[t];
        MSI_TRY(exhaustive_one(vs, i, k, d_fbits, filter_nbits, vs->out_docids.as<uint32_t>() + (size_t)i * k,
                               vs->out_dist.as<float>() + (size_t)i * k, s.counts + i));
        const size_t row = (size_t)(q0 + j) * k;
        MSI_HIP_TRY(hipMemcpyAsync(out_docids + row, vs->out_docids.as<uint32_t>() + (size_t)i * k, (size_t)k * sizeof(uint32_t),
                                   hipMemcpyDeviceToHost, st));
        MSI_HIP_TRY(hipMemcpyAsync(out_dist + row, vs->out_dist.as<float>() + (size_t)i * k, (size_t)k * sizeof(float),
                                   hipMemcpyDeviceToHost, st));
        MSI_HIP_TRY(hipMemcpyAsync(out_counts + q0 + j, s.counts + i, sizeof(uint32_t), hipMemcpyDeviceToHost, st));
      }
      MSI_HIP_TRY(hipStreamSynchronize(st));
      return MSI_OK;
    };
    if (l0 + 1 == n_levels && !pending.empty()) {   // the first sweep ran at the last level already
      MSI_TRY(exhaustive_now(pending, pending));
      pending.clear();
    }
    // the queries the sweep could not prove: level by level
    for (uint32_t lvl = l0 + 1; lvl < n_levels && !pending.empty(); ++lvl) {
      const uint32_t sub = batch_of(levels[lvl]);
      std::vector<uint32_t> next;
      for (size_t f0 = 0; f0 < pending.size(); f0 += sub) {
        const uint32_t nf = (uint32_t)std::min<size_t>(sub, pending.size() - f0);
        if (cancel && *cancel) {
          msi_set_error("msi_vs_search: cancelled");
          return MSI_E_CANCELLED;
        }
        for (uint32_t i = 0; i < nf; ++i)
          MSI_HIP_TRY(hipMemcpyAsync(vs->qraw.as<float>() + (size_t)i * vs->dim,
                                     queries + (size_t)(q0 + pending[f0 + i]) * vs->dim, (size_t)vs->dim * sizeof(float),
                                     hipMemcpyHostToDevice, st));
        uint32_t f2[NQ_MAX], c2[NQ_MAX];
        MSI_TRY(sweep(levels[lvl], nf, f2, c2));
        ++vs->level_sweeps[lvl];
        std::vector<uint32_t> left_i, left_j;
        for (uint32_t i = 0; i < nf; ++i) {
          const uint32_t j = pending[f0 + i];
          if (f2[i]) {
            if (lvl + 1 == n_levels) {
              left_i.push_back(i);
              left_j.push_back(j);
            } else {
              next.push_back(j);
            }
            continue;
          }
          const size_t row = (size_t)(q0 + j) * k;
          MSI_HIP_TRY(hipMemcpyAsync(out_docids + row, vs->out_docids.as<uint32_t>() + (size_t)i * k, (size_t)k * sizeof(uint32_t),
                                     hipMemcpyDeviceToHost, st));
          MSI_HIP_TRY(hipMemcpyAsync(out_dist + row, vs->out_dist.as<float>() + (size_t)i * k, (size_t)k * sizeof(float),
                                     hipMemcpyDeviceToHost, st));
          out_counts[q0 + j] = c2[i];
        }
        MSI_HIP_TRY(hipStreamSynchronize(st));
        if (!left_i.empty()) MSI_TRY(exhaustive_now(left_i, left_j));
      }
      pending.swap(next);
    }
  }
  return MSI_OK;
}

extern "C" {

// Test instrumentation: the fast scan's raw scores (dot / |row|, before any
// thresholding) of every row for <= msi_vs_max_batch() host queries, and the bound
// `eps` the exactness proof assumes for |fast cos - reference cos|.
int32_t msi_vs_debug_fast_scores(msi_vs *vs, const float *queries, uint32_t n_queries, float *out_scores,
                                 float *out_eps) {
  if (!vs || !queries || !out_scores || n_queries == 0 || n_queries > vs->nqt_max * QT) {
    msi_set_error("msi_vs_debug_fast_scores: invalid argument");
    return MSI_E_INVALID;
  }
  if (vs->n_rows > (1u << 22)) {
    msi_set_error("msi_vs_debug_fast_scores: store too large for the debug path");
    return MSI_E_UNSUPPORTED;
  }
  std::lock_guard<std::mutex> lk(vs->ctx->mu);
  DeviceGuard g(vs->ctx->device);
  hipStream_t st = vs->ctx->stream;
  Small s = small_of(vs);
  const uint32_t nqt = (n_queries + QT - 1) / QT;
  const uint32_t dstride = (uint32_t)(vs->n_tiles * 16);
  MSI_TRY(vs->dense.ensure((size_t)NQ_MAX * std::max<uint32_t>(16, dstride) * sizeof(float)));
  MSI_HIP_TRY(hipMemcpyAsync(vs->qraw.p, queries, (size_t)n_queries * vs->dim * sizeof(float), hipMemcpyHostToDevice, st));
  hipLaunchKernelGGL(vs_prep_queries_kernel, dim3(nqt * QT), dim3(256), (size_t)vs->dpad * sizeof(float), st,
                     vs->qraw.as<float>(), n_queries, vs->dim, vs->KB, vs->qfrag.as<float4>(),
                     vs->qfrag_bf.as<bf16x8>(), vs->qrow.as<float>(), s.qn, s.inv_qn, s.degth, vs->s16 ? 1u : (vs->bf2 ? 2u : 0u));
  ScanArgs sa;
  memset(&sa, 0, sizeof(sa));
  sa.tiles = vs->tiles.as<float4>();
  sa.inv_norm = vs->inv_norm.as<float>();
  sa.qfrag = (vs->bf3 || vs->s16) ? vs->qfrag_bf.as<float4>() : vs->qfrag.as<float4>();
  sa.theta = s.theta_inf;
  sa.degth = s.degth;
  sa.n_items_ptr = s.n_tiles;
  sa.gkeys = vs->gkeys.as<u64>();
  sa.gcnt = vs->gcnt.as<uint32_t>();
  sa.overflow = s.overflow;
  sa.dense = vs->dense.as<float>();
  sa.n_rows = vs->n_rows;
  sa.dstride = dstride;
  sa.capg = vs->capg;
  sa.KB = vs->KB;
  sa.stride = 1;
  sa.nq = n_queries;
  // MSI_VS_DEBUG_I8=1 (tests): the int8 sweep's scores and the largest per-query bound of the batch
  const char *d8 = getenv("MSI_VS_DEBUG_I8");
  const bool use8 = vs->i8 && d8 && d8[0] == '1';
  if (use8) {
    const float eps_base = (4.0f * (float)vs->dpad + 64.0f) * 5.9604645e-8f + 1e-5f;
    hipLaunchKernelGGL(vs_prep_queries_i8_kernel, dim3(nqt * QT), dim3(256), 0, st, vs->qrow.as<float>(), s.qn, s.inv_qn, n_queries,
                       vs->dpad, vs->qfrag8.as<i32x4>(), s.sqs, s.epsq, vs->i8small.as<uint32_t>(), eps_base);
    Scan8Args s8;
    memset(&s8, 0, sizeof(s8));
    s8.tiles8 = vs->tiles8.as<i32x4>();
    s8.scale8 = vs->scale8.as<float>();
    s8.inv_norm = sa.inv_norm;
    s8.i8small = vs->i8small.as<uint32_t>();
    s8.qfrag8 = vs->qfrag8.as<i32x4>();
    s8.sqs = s.sqs;
    s8.theta = sa.theta;
    s8.degth = sa.degth;
    s8.n_items_ptr = sa.n_items_ptr;
    s8.gkeys = sa.gkeys;
    s8.gcnt = sa.gcnt;
    s8.overflow = sa.overflow;
    s8.dense = sa.dense;
    s8.n_rows = sa.n_rows;
    s8.dstride = dstride;
    s8.capg = sa.capg;
    s8.KB8 = vs->KB8;
    s8.stride = 1;
    s8.nq = n_queries;
    if (vs->n_rows) launch_scan8(vs, s8, nqt, true);
  } else if (vs->n_rows) {
    launch_scan(vs, sa, nqt, true);
  }
  MSI_HIP_TRY(hipGetLastError());
  for (uint32_t j = 0; j < n_queries; ++j)
    if (vs->n_rows)
      MSI_HIP_TRY(hipMemcpyAsync(out_scores + (size_t)j * vs->n_rows, vs->dense.as<float>() + (size_t)j * dstride,
                                 vs->n_rows * sizeof(float), hipMemcpyDeviceToHost, st));
  float h_epsq[NQ_MAX];
  if (use8) MSI_HIP_TRY(hipMemcpyAsync(h_epsq, s.epsq, n_queries * sizeof(float), hipMemcpyDeviceToHost, st));
  MSI_HIP_TRY(hipStreamSynchronize(st));
  if (out_eps) {
    *out_eps = scan_eps(vs);
    if (use8) {
      *out_eps = 0.f;
      for (uint32_t j = 0; j < n_queries; ++j) *out_eps = std::max(*out_eps, h_epsq[j]);
    }
  }
  return MSI_OK;
}

int32_t msi_vs_scan_time(msi_vs *vs, uint64_t *out_launches, double *out_ms_total) {
  if (!vs || !out_launches || !out_ms_total) return MSI_E_INVALID;
  std::lock_guard<std::mutex> lk(vs->ctx->mu);
  DeviceGuard g(vs->ctx->device);
  MSI_HIP_TRY(hipStreamSynchronize(vs->ctx->stream));
  vs->scan_timer.drain(out_launches, out_ms_total);
  return MSI_OK;
}

int32_t msi_vs_filter_stats(msi_vs *vs, uint64_t out[4]) {
  if (!vs || !out) return MSI_E_INVALID;
  out[0] = out[1] = out[2] = out[3] = 0;
  if (!vs->filtered_calls || !vs->fsmall.p) return MSI_OK;
  std::lock_guard<std::mutex> lk(vs->ctx->mu);
  DeviceGuard g(vs->ctx->device);
  MSI_HIP_TRY(hipStreamSynchronize(vs->ctx->stream));
  if (vs->aux_stream) MSI_HIP_TRY(hipStreamSynchronize(vs->aux_stream));
  uint32_t h[4] = {0, 0, 0, 0};
  MSI_HIP_TRY(hipMemcpy(h, vs->fsmall.p, sizeof(h), hipMemcpyDeviceToHost));
  for (int i = 0; i < 4; ++i) out[i] = h[i];
  return MSI_OK;
}

int32_t msi_vs_get_stats(const msi_vs *vs, msi_vs_stats *out) {
  if (!vs || !out) return MSI_E_INVALID;
  out->scan_launches = vs->scan_launches;
  out->scan_tiles = vs->scan_tiles;
  out->exhaustive_reruns = vs->exhaustive_reruns;
  out->second_opinion_queries = vs->second_opinion_queries;
  out->x3_first_sweeps = vs->x3_first_sweeps;
  out->x2_sweeps = vs->x2_sweeps;
  // (level_sweeps of the ABI: the f32 levels; a store with an int8 copy counts its level 0 in i8_sweeps)
  for (int i = 0; i < 3; ++i) out->level_sweeps[i] = vs->level_sweeps[i + (vs->i8 ? 2 : 0)];
  out->bytes_per_tile = (uint64_t)vs->KB * 1024;
  out->i8_bytes_per_tile = vs->i8 ? (uint64_t)vs->KB8 * 1024 + 16 * 8 : 0;
  out->i8_sweeps = vs->i8_sweeps;
  out->i8_scan_tiles = vs->i8_scan_tiles;
  out->device_rerun_queries = vs->device_rerun_queries;
  out->i8_queries_per_sweep = vs->i8 ? vs->nqt8_max * QT : 0;
  out->f32_queries_per_sweep = vs->nqt_max * QT;
  return MSI_OK;
}

// The store's items as a docid set: pool[slot] |= {docids of the rows} (filter/vector.rs: items_in_store / aggregate_stats)
int32_t msi_vs_items_bits(msi_vs *vs, msi_bits *pool, uint32_t slot) {
  if (!vs || !pool) return MSI_E_INVALID;
  if (msi_bits_ctx(pool) != vs->ctx) {
    msi_set_error("msi_vs_items_bits: the pool and the store live on different contexts");
    return MSI_E_INVALID;
  }
  return msi_bits_or_docids_device(pool, slot, vs->docids.as<uint32_t>(), vs->n_rows);
}

}  // extern "C"
