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
//   vs_scan      streams every (allowed) tile once: D[16 rows][16 queries] +=
//                A·B with the query fragments in LDS; the epilogue scales by
//                1/|row|, compares against a per-query threshold and only the rare
//                survivors take the slow path into a per-wave LDS candidate list.
//   thresholds   a strided sample pass (sqrt(K'·N) rows) gives each query a
//                valid lower bound on its K'-th best score, so the main pass
//                keeps ~K'·N/S rows per query instead of warming up per wave.
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
#include <string.h>

#include <algorithm>

#include "msi_common.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned long long u64;

namespace {

constexpr int SCAN_WAVES = 8;            // waves per workgroup (512 threads)
constexpr int SCAN_CAP = 64;             // per-wave, per-query LDS candidate slots
constexpr int SCAN_GROUP = 8;            // KiB blocks per software-pipeline stage
constexpr uint32_t LOCAL_KP_MAX = SCAN_CAP - 16;  // K' up to which waves self-compact
constexpr uint32_t KP_MAX = 1024;        // K' supported by select/rescore
constexpr int SEL_THREADS = 256;
constexpr int SEL_SORTCAP = 2048;        // u64 keys sorted in LDS by vs_select
constexpr int QT = 16;                   // queries per pass (one MFMA tile)

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
  tiles[(row0 / 16) * KB * 64 + idx] = make_float4(v[0], v[1], v[2], v[3]);
}

// Canonical row norms from the tiled layout: pn = sqrtf(sum_k x_k*x_k), sequential
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
  const float4 *base = tiles + t * KB * 64 + i;
  float acc = 0.f;
  for (uint32_t kb = 0; kb < KB; ++kb) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      float4 v = base[(uint64_t)kb * 64 + g * 16];
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

// ------------------------------------------------------------- query preparation

// queries row-major [nq][dim] -> MFMA B fragments [KB][64] float4 (lane l=g*16+j:
// query j, columns 16kb+4g..+3), canonical |q|, thresholds for degenerate rows.
__global__ void vs_prep_queries_kernel(const float *__restrict__ q, uint32_t nq, uint32_t dim,
                                       uint32_t KB, float4 *__restrict__ qfrag,
                                       float *__restrict__ qrow /*[QT][KB*16]*/,
                                       float *__restrict__ qn, float *__restrict__ inv_qn,
                                       float *__restrict__ degth) {
  uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
  uint32_t total = KB * 64;
  uint32_t dpad = KB * 16;
  for (uint32_t idx = tid; idx < total; idx += gridDim.x * blockDim.x) {
    uint32_t lane = idx & 63, kb = idx >> 6;
    uint32_t j = lane & 15, g = lane >> 4;
    uint32_t k0 = kb * 16 + g * 4;
    float v[4] = {0.f, 0.f, 0.f, 0.f};
    if (j < nq)
      for (int c = 0; c < 4; ++c)
        if (k0 + c < dim) v[c] = q[(uint64_t)j * dim + k0 + c];
    qfrag[idx] = make_float4(v[0], v[1], v[2], v[3]);
  }
  for (uint32_t idx = tid; idx < QT * dpad; idx += gridDim.x * blockDim.x) {
    uint32_t j = idx / dpad, k = idx % dpad;
    qrow[idx] = (j < nq && k < dim) ? q[(uint64_t)j * dim + k] : 0.f;
  }
  if (tid < QT) {
    float acc = 0.f;
    if (tid < nq)
      for (uint32_t k = 0; k < dim; ++k) {
        float x = q[(uint64_t)tid * dim + k];
        acc = __fadd_rn(acc, __fmul_rn(x, x));
      }
    float n = msi_sqrt_rn(acc);
    qn[tid] = n;
    inv_qn[tid] = n > 0.f ? 1.0f / n : 0.f;
    // row is (conservatively) degenerate when pn*qn <= EPS  <=>  1/pn >= qn/EPS
    degth[tid] = n * (0.999f / FLT_EPSILON);
  }
}

// ------------------------------------------------------------------- filter path

// Per-tile 16-bit "row allowed" masks + compacted list of tiles with any allowed
// row.  One thread per row; a wave covers 4 tiles.
__global__ void vs_filter_tiles_kernel(const uint32_t *__restrict__ docids, uint64_t n_rows,
                                       const u64 *__restrict__ fbits, uint64_t nbits,
                                       uint16_t *__restrict__ tmask, uint32_t *__restrict__ list,
                                       uint32_t *__restrict__ n_items) {
  uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  uint64_t padded = ((n_rows + 15) / 16) * 16;
  bool ok = false;
  if (r < n_rows) {
    uint32_t id = docids[r];
    if ((uint64_t)id < nbits) ok = (fbits[id >> 6] >> (id & 63)) & 1ull;
  }
  u64 b = __ballot(ok);
  uint32_t lane = threadIdx.x & 63;
  if (r < padded && (lane & 15) == 0) {
    uint16_t m = (uint16_t)((b >> lane) & 0xFFFFull);
    uint64_t t = r >> 4;
    tmask[t] = m;
    if (m) {
      uint32_t slot = atomicAdd(n_items, 1u);
      list[slot] = (uint32_t)t;
    }
  }
}

// ------------------------------------------------------------------------ scan

struct ScanArgs {
  const float4 *tiles;
  const float *inv_norm;
  const float4 *qfrag;
  const float *theta;            // [QT] pass if !(score < theta)
  const float *degth;            // [QT]
  const uint32_t *n_items_ptr;   // number of entries of `list` (or tiles when list==null)
  const uint32_t *list;          // nullable: active tile ids
  const uint16_t *tmask;         // nullable: per-tile allowed-row masks
  u64 *gkeys;                    // [QT][capg]
  uint32_t *gcnt;                // [QT]
  uint32_t *overflow;            // set to 1 if a global buffer overflowed
  uint64_t n_rows;
  uint32_t capg;
  uint32_t KB;
  uint32_t stride;               // 1 = every item, S = every S-th item (sample pass)
  uint32_t kp;                   // K'
};

// 64-lane bitonic sort, ascending, one key per lane.
__device__ __forceinline__ u64 wave_sort64(u64 key, uint32_t lane) {
#pragma unroll
  for (uint32_t k = 2; k <= 64; k <<= 1) {
#pragma unroll
    for (uint32_t j = k >> 1; j > 0; j >>= 1) {
      u64 other = __shfl_xor(key, (int)j);
      bool up = ((lane & k) == 0);
      bool lower = ((lane & j) == 0);
      bool take_min = (up == lower);
      u64 mn = key < other ? key : other;
      u64 mx = key < other ? other : key;
      key = take_min ? mn : mx;
    }
  }
  return key;
}

template <int WAVES>
__global__ __launch_bounds__(WAVES * 64) void vs_scan_kernel(ScanArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const uint32_t tid = threadIdx.x;
  const uint32_t lane = tid & 63;
  const uint32_t wave = tid >> 6;
  const uint32_t KB = a.KB;

  float4 *qf = reinterpret_cast<float4 *>(smem);
  size_t off = (size_t)KB * 64 * sizeof(float4);
  volatile u64 *ckeys = reinterpret_cast<volatile u64 *>(smem + off) + (size_t)wave * QT * SCAN_CAP;
  off += (size_t)WAVES * QT * SCAN_CAP * sizeof(u64);
  volatile uint32_t *ccnt = reinterpret_cast<volatile uint32_t *>(smem + off) + wave * QT;
  off += (size_t)WAVES * QT * sizeof(uint32_t);
  volatile float *cth = reinterpret_cast<volatile float *>(smem + off) + wave * QT;

  for (uint32_t i = tid; i < KB * 64; i += WAVES * 64) qf[i] = a.qfrag[i];
  if (lane < QT) {
    ccnt[lane] = 0;
    cth[lane] = a.theta[lane];
  }
  __syncthreads();

  const uint32_t qj = lane & 15;   // this lane's query (D column)
  const uint32_t g = lane >> 4;    // this lane's row group: rows 4g..4g+3 of the tile
  float th = cth[qj];
  const float dth = a.degth[qj];
  const bool local_mode = a.kp <= LOCAL_KP_MAX;

  // this wave's contiguous share of the item list
  const uint32_t n_all = *a.n_items_ptr;
  const uint32_t n_items = (n_all + a.stride - 1) / a.stride;
  const uint64_t gw = (uint64_t)blockIdx.x * WAVES + wave;
  const uint64_t GW = (uint64_t)gridDim.x * WAVES;
  const uint32_t it0 = (uint32_t)((uint64_t)n_items * gw / GW);
  const uint32_t it1 = (uint32_t)((uint64_t)n_items * (gw + 1) / GW);

  const uint32_t GPT = KB / SCAN_GROUP;  // pipeline groups per tile

  auto tile_of = [&](uint32_t it) -> uint32_t {
    uint32_t idx = it * a.stride;
    return a.list ? a.list[idx] : idx;
  };

  // ---- epilogue: scale, threshold, rare slow path -------------------------
  auto epilogue = [&](uint32_t tile, f32x4 acc) {
    const uint64_t row0 = (uint64_t)tile * 16 + g * 4;
    const float4 inv = *reinterpret_cast<const float4 *>(a.inv_norm + row0);
    uint32_t allowed = 0xF;
    if (a.tmask) allowed = (a.tmask[tile] >> (g * 4)) & 0xF;
    else if (row0 + 4 > a.n_rows) allowed = row0 >= a.n_rows ? 0u : ((1u << (a.n_rows - row0)) - 1u);
    float s[4];
    s[0] = acc[0] * inv.x;
    s[1] = acc[1] * inv.y;
    s[2] = acc[2] * inv.z;
    s[3] = acc[3] * inv.w;
    const float iv[4] = {inv.x, inv.y, inv.z, inv.w};
    uint32_t pass = 0;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      if (iv[r] >= dth) s[r] = FLT_MAX;          // pn*qn <= EPS: reference distance is 0
      if (!(s[r] < th)) pass |= 1u << r;         // NaN passes; fixed up below
    }
    pass &= allowed;
    if (__ballot(pass != 0) == 0) return;
    // slow path (rare): append survivors to this wave's list for query qj
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      if (pass & (1u << r)) {
        float sc = s[r];
        if (!(sc == sc)) sc = FLT_MAX;
        uint32_t slot = atomicAdd(const_cast<uint32_t *>(&ccnt[qj]), 1u);
        ckeys[qj * SCAN_CAP + slot] = make_key_desc(sc, (uint32_t)(row0 + r));
      }
    }
    __builtin_amdgcn_wave_barrier();
    uint32_t c = lane < QT ? ccnt[lane] : 0;
    u64 need = __ballot(c > (uint32_t)(SCAN_CAP - 16));
    while (need) {
      const uint32_t j = (uint32_t)__ffsll((long long)need) - 1;
      need &= need - 1;
      const uint32_t cj = ccnt[j];
      u64 key = lane < cj ? ckeys[j * SCAN_CAP + lane] : ~0ull;
      if (local_mode) {
        key = wave_sort64(key, lane);
        if (lane < a.kp) ckeys[j * SCAN_CAP + lane] = key;
        u64 kth = __shfl(key, (int)(a.kp - 1));
        if (lane == 0) {
          ccnt[j] = cj < a.kp ? cj : a.kp;
          if (cj >= a.kp) cth[j] = key_desc_score(kth);
        }
      } else {
        // flush mode (large K'): move the list to the global buffer unchanged
        uint32_t base = 0;
        if (lane == 0) base = atomicAdd(&a.gcnt[j], cj);
        base = __shfl(base, 0);
        if (lane < cj) {
          if (base + lane < a.capg) a.gkeys[(uint64_t)j * a.capg + base + lane] = key;
          else *a.overflow = 1;
        }
        if (lane == 0) ccnt[j] = 0;
      }
      __builtin_amdgcn_wave_barrier();
    }
    th = cth[qj];
  };

  if (it0 < it1) {
    float4 xa[SCAN_GROUP], xb[SCAN_GROUP];
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};

    uint32_t it_load = it0, sub_load = 0;   // next group to load
    uint32_t it_cmp = it0, sub_cmp = 0;     // next group to compute
    uint32_t tile_load = tile_of(it_load);
    uint32_t tile_cmp = tile_load;

    auto load_group = [&](float4(&x)[SCAN_GROUP]) {
      const float4 *p = a.tiles + ((uint64_t)tile_load * KB + (uint64_t)sub_load * SCAN_GROUP) * 64 + lane;
#pragma unroll
      for (int u = 0; u < SCAN_GROUP; ++u) x[u] = p[u * 64];
      if (++sub_load == GPT) {
        sub_load = 0;
        ++it_load;
        if (it_load < it1) tile_load = tile_of(it_load);
      }
    };
    auto compute_group = [&](const float4(&x)[SCAN_GROUP]) {
      const float4 *qp = qf + (size_t)sub_cmp * SCAN_GROUP * 64 + lane;
#pragma unroll
      for (int u = 0; u < SCAN_GROUP; ++u) {
        const float4 q = qp[u * 64];
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(x[u].x, q.x, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(x[u].y, q.y, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(x[u].z, q.z, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(x[u].w, q.w, acc, 0, 0, 0);
      }
      if (++sub_cmp == GPT) {
        epilogue(tile_cmp, acc);
        acc = (f32x4){0.f, 0.f, 0.f, 0.f};
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

  // ---- flush this wave's lists to the global per-query buffers ---------------
  __builtin_amdgcn_wave_barrier();
  for (uint32_t j = 0; j < QT; ++j) {
    const uint32_t cj = ccnt[j];
    if (cj == 0) continue;
    uint32_t base = 0;
    if (lane == 0) base = atomicAdd(&a.gcnt[j], cj);
    base = __shfl(base, 0);
    if (lane < cj) {
      if (base + lane < a.capg) a.gkeys[(uint64_t)j * a.capg + base + lane] = ckeys[j * SCAN_CAP + lane];
      else *a.overflow = 1;
    }
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

// Leaves the min(c,K) smallest keys of keys[0..c), ascending, in sbuf[0..) and
// returns their count.  K <= KP_MAX, sbuf has SEL_SORTCAP entries, hist 2048.
__device__ uint32_t block_select_smallest(const u64 *__restrict__ keys, uint32_t c, uint32_t K,
                                          u64 *sbuf, uint32_t *hist, uint32_t *sh) {
  const uint32_t tid = threadIdx.x;
  if (c <= SEL_SORTCAP) {
    uint32_t n = next_pow2(c < 2 ? 2 : c);
    for (uint32_t i = tid; i < n; i += blockDim.x) sbuf[i] = i < c ? keys[i] : ~0ull;
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
    for (uint32_t i = tid; i < c; i += blockDim.x) {
      u64 key = keys[i];
      if (pbits == 0 || (key >> (64 - pbits)) == prefix)
        atomicAdd(&hist[(uint32_t)(key >> shift) & ((1u << dbits) - 1u)], 1u);
    }
    __syncthreads();
    if (tid == 0) {
      uint32_t cum = 0, b = 0;
      const uint32_t nb = 1u << dbits;
      for (; b < nb; ++b) {
        if (cum + hist[b] >= krem) break;
        cum += hist[b];
      }
      if (b == nb) b = nb - 1;  // c < K cannot happen here (c > SEL_SORTCAP >= K)
      sh[0] = b;
      sh[1] = cum;       // entries of this prefix strictly below bin b
      sh[2] = hist[b];
    }
    __syncthreads();
    const uint32_t b = sh[0], below = sh[1], inbin = sh[2];
    const u64 newprefix = (prefix << dbits) | b;
    const uint32_t gather = (K - krem) + below + inbin;  // keys with top bits <= newprefix
    __syncthreads();
    if (gather <= SEL_SORTCAP || level == 5) {
      if (tid == 0) sh[3] = 0;
      __syncthreads();
      for (uint32_t i = tid; i < c; i += blockDim.x) {
        u64 key = keys[i];
        if ((key >> shift) <= newprefix) {
          uint32_t slot = atomicAdd(&sh[3], 1u);
          if (slot < SEL_SORTCAP) sbuf[slot] = key;
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

struct SelectArgs {
  const u64 *gkeys;    // [QT][capg]
  uint32_t *gcnt;      // [QT]; reset to 0 on exit
  uint32_t capg;
  uint32_t kp;
  u64 *sel_keys;       // [QT][KP_MAX]   (mode 1)
  uint32_t *sel_cnt;   // [QT]           (mode 1)
  float *theta;        // [QT]           (mode 0: threshold for the main pass)
  int mode;            // 0 = threshold only, 1 = keep the keys
};

__global__ __launch_bounds__(SEL_THREADS) void vs_select_kernel(SelectArgs a) {
  __shared__ u64 sbuf[SEL_SORTCAP];
  __shared__ uint32_t hist[2048];
  __shared__ uint32_t sh[4];
  const uint32_t j = blockIdx.x;
  uint32_t c = a.gcnt[j];
  if (c > a.capg) c = a.capg;
  const uint32_t got = block_select_smallest(a.gkeys + (uint64_t)j * a.capg, c, a.kp, sbuf, hist, sh);
  __syncthreads();
  if (a.mode == 0) {
    if (threadIdx.x == 0) a.theta[j] = got >= a.kp ? key_desc_score(sbuf[a.kp - 1]) : -INFINITY;
  } else {
    for (uint32_t i = threadIdx.x; i < got; i += blockDim.x) a.sel_keys[(uint64_t)j * KP_MAX + i] = sbuf[i];
    if (threadIdx.x == 0) a.sel_cnt[j] = got;
  }
  if (threadIdx.x == 0) a.gcnt[j] = 0;
}

// --------------------------------------------------------------------- rescore

// Reference arithmetic for one (row, query) pair from the tiled layout:
// sequential f32 mul+add in column order, then arroy/hannoy's cosine distance.
__device__ __forceinline__ float canonical_dot(const float4 *__restrict__ tiles, uint32_t KB,
                                               uint32_t row, const float *__restrict__ q) {
  const float4 *base = tiles + (uint64_t)(row >> 4) * KB * 64 + (row & 15);
  float acc = 0.f;
  for (uint32_t kb = 0; kb < KB; ++kb) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const float4 v = base[(uint64_t)kb * 64 + g * 16];
      const float *qq = q + kb * 16 + g * 4;
      acc = __fadd_rn(acc, __fmul_rn(v.x, qq[0]));
      acc = __fadd_rn(acc, __fmul_rn(v.y, qq[1]));
      acc = __fadd_rn(acc, __fmul_rn(v.z, qq[2]));
      acc = __fadd_rn(acc, __fmul_rn(v.w, qq[3]));
    }
  }
  return acc;
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
  const float4 *tiles;
  const float *norm;
  const uint32_t *docids;
  const float *qrow;       // [QT][dpad]
  const float *qn;         // [QT]
  const float *inv_qn;     // [QT]
  const u64 *sel_keys;     // [QT][KP_MAX]
  const uint32_t *sel_cnt; // [QT]
  uint32_t KB;
  uint32_t kp;
  uint32_t k;
  float eps;               // bound on |fast cos - reference cos|
  uint32_t *out_docids;    // [nq][k]
  float *out_dist;         // [nq][k]
  uint32_t *out_counts;    // [nq]
  uint32_t *inexact;       // [nq]
  const uint32_t *overflow;
};

__global__ __launch_bounds__(SEL_THREADS) void vs_rescore_kernel(RescoreArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char dyn[];
  u64 *sbuf = reinterpret_cast<u64 *>(dyn);                          // [KP_MAX]
  float *qs = reinterpret_cast<float *>(dyn + KP_MAX * sizeof(u64)); // [dpad]
  const uint32_t j = blockIdx.x;
  const uint32_t dpad = a.KB * 16;
  for (uint32_t i = threadIdx.x; i < dpad; i += blockDim.x) qs[i] = a.qrow[(uint64_t)j * dpad + i];
  const uint32_t cnt = a.sel_cnt[j];
  const uint32_t n = next_pow2(cnt < 2 ? 2 : cnt);
  __syncthreads();
  const float qn = a.qn[j];
  for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) {
    u64 key = ~0ull;
    if (i < cnt) {
      const uint32_t row = (uint32_t)a.sel_keys[(uint64_t)j * KP_MAX + i];
      const float pq = canonical_dot(a.tiles, a.KB, row, qs);
      const float d = canonical_distance(pq, a.norm[row], qn);
      key = ((u64)f32_to_ord(d) << 32) | row;  // rows ascend with docids
    }
    sbuf[i] = key;
  }
  __syncthreads();
  block_bitonic_sort(sbuf, n);
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
    if (cnt == a.kp && out_n > 0) {
      const float smin = key_desc_score(a.sel_keys[(uint64_t)j * KP_MAX + cnt - 1]);
      const float cmin = smin * a.inv_qn[j];
      const float bound = (1.0f - cmin - a.eps) * 0.5f - 2e-7f;
      const float dk = ord_to_f32((uint32_t)(sbuf[out_n - 1] >> 32));
      if (!(dk < bound)) bad = 1;
    }
    if (a.inexact) a.inexact[j] = bad;
  }
}

// ------------------------------------------------------------------ exhaustive

// Reference distance of EVERY allowed row for one query (fallback when the
// exactness proof fails, e.g. more than K' rows tie at the cut).
__global__ void vs_exhaustive_kernel(const float4 *__restrict__ tiles, const float *__restrict__ norm,
                                     const uint32_t *__restrict__ docids, uint64_t n_rows, uint32_t KB,
                                     const float *__restrict__ qrow, const float *__restrict__ qn_p,
                                     uint32_t qj, const u64 *__restrict__ fbits, uint64_t nbits,
                                     u64 *__restrict__ keys) {
  extern __shared__ __attribute__((aligned(16))) unsigned char dyn[];
  float *qs = reinterpret_cast<float *>(dyn);
  const uint32_t dpad = KB * 16;
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
    const float pq = canonical_dot(tiles, KB, (uint32_t)r, qs);
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
  uint32_t got = block_select_smallest(keys, c, k, sbuf, hist, sh);
  __syncthreads();
  // disallowed rows carry key ~0: drop them
  __shared__ uint32_t valid;
  if (threadIdx.x == 0) valid = 0;
  __syncthreads();
  for (uint32_t i = threadIdx.x; i < got; i += blockDim.x)
    if (sbuf[i] != ~0ull) atomicAdd(&valid, 1u);
  __syncthreads();
  const uint32_t out_n = valid;
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

__global__ void vs_gather_row_kernel(const float4 *__restrict__ tiles, uint32_t KB, uint32_t row,
                                     uint32_t dim, float *__restrict__ out) {
  uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= dim) return;
  const float4 v = tiles[((uint64_t)(row >> 4) * KB + (k >> 4)) * 64 + ((k >> 2) & 3) * 16 + (row & 15)];
  const float c[4] = {v.x, v.y, v.z, v.w};
  out[k] = c[k & 3];
}

}  // namespace

// ============================================================== host-side object

struct msi_vs {
  msi_ctx *ctx = nullptr;
  uint32_t dim = 0, dpad = 0, KB = 0;
  uint64_t n_rows = 0, n_tiles = 0;
  DevBuf tiles, norm, inv_norm, docids;
  std::vector<uint32_t> h_docids;  // for get_vector's binary search
  // scratch (guarded by ctx->mu)
  DevBuf qraw, qfrag, qrow, qsmall /* qn, inv_qn, degth, theta[2] */, gkeys, gsmall, sel_keys,
      tmask, tlist, fbits, out_docids, out_dist, out_small, exh_keys, rowtmp;
  uint32_t capg = 0;
  uint32_t scan_grid = 0;
  uint32_t waves = SCAN_WAVES;
  // stats
  uint64_t scan_launches = 0, scan_tiles = 0, exhaustive_reruns = 0;
  KernelTimer scan_timer;
};

namespace {

// layout of the small scratch arrays
struct Small {
  float *qn, *inv_qn, *degth, *theta_inf, *theta;
  uint32_t *gcnt, *sel_cnt, *n_tiles, *n_items, *overflow, *inexact, *counts, *bad;
};

Small small_of(msi_vs *vs) {
  Small s;
  float *f = vs->qsmall.as<float>();
  s.qn = f;
  s.inv_qn = f + QT;
  s.degth = f + 2 * QT;
  s.theta_inf = f + 3 * QT;
  s.theta = f + 4 * QT;
  uint32_t *u = vs->gsmall.as<uint32_t>();
  s.gcnt = u;
  s.sel_cnt = u + QT;
  s.n_tiles = u + 2 * QT;
  s.n_items = u + 2 * QT + 1;
  s.overflow = u + 2 * QT + 2;
  s.bad = u + 2 * QT + 3;
  s.inexact = u + 3 * QT;
  s.counts = u + 4 * QT;
  return s;
}

size_t scan_lds_bytes(uint32_t KB, uint32_t waves) {
  return (size_t)KB * 64 * sizeof(float4) + (size_t)waves * QT * SCAN_CAP * sizeof(u64) +
         (size_t)waves * QT * (sizeof(uint32_t) + sizeof(float));
}

void launch_scan(msi_vs *vs, const ScanArgs &sa);

int32_t ensure_scratch(msi_vs *vs) {
  MSI_TRY(vs->qraw.ensure((size_t)QT * vs->dim * sizeof(float)));
  MSI_TRY(vs->qfrag.ensure((size_t)vs->KB * 64 * sizeof(float4)));
  MSI_TRY(vs->qrow.ensure((size_t)QT * vs->dpad * sizeof(float)));
  MSI_TRY(vs->qsmall.ensure(5 * QT * sizeof(float)));
  MSI_TRY(vs->gsmall.ensure(6 * QT * sizeof(uint32_t)));
  MSI_TRY(vs->sel_keys.ensure((size_t)QT * KP_MAX * sizeof(u64)));
  return MSI_OK;
}

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
  // re-tile in chunks (bounded staging memory for host uploads)
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
    hipLaunchKernelGGL(vs_tile_rows_kernel, dim3((uint32_t)((total + 255) / 256)), dim3(256), 0, st, src,
                       r0, nc, vs->dim, vs->KB, vs->tiles.as<float4>());
    if (!rows_on_device) MSI_HIP_TRY(hipStreamSynchronize(st));  // rowtmp is reused
  }
  if (padded)
    hipLaunchKernelGGL(vs_row_norms_kernel, dim3((uint32_t)((padded + 255) / 256)), dim3(256), 0, st,
                       vs->tiles.as<float4>(), (uint64_t)0, n_rows, vs->KB, vs->norm.as<float>(),
                       vs->inv_norm.as<float>());
  uint32_t nt32 = (uint32_t)n_tiles;
  MSI_HIP_TRY(hipMemcpyAsync(s.n_tiles, &nt32, sizeof(uint32_t), hipMemcpyHostToDevice, st));
  float ninf[QT];
  for (int i = 0; i < QT; ++i) ninf[i] = -INFINITY;  // threshold of the first pass
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
    msi_set_error("msi_vs_upload: docids must be strictly ascending");
    return MSI_E_NOT_SORTED;
  }
  vs->n_rows = n_rows;
  vs->n_tiles = n_tiles;
  // per-query global candidate buffers: every wave may flush SCAN_CAP keys per query
  vs->scan_grid = (uint32_t)ctx->n_cu;
  vs->capg = vs->scan_grid * vs->waves * SCAN_CAP * 2;
  MSI_TRY(vs->gkeys.ensure((size_t)QT * vs->capg * sizeof(u64)));
  MSI_HIP_TRY(hipMemsetAsync(vs->gsmall.p, 0, 2 * QT * sizeof(uint32_t), st));  // gcnt, sel_cnt
  return MSI_OK;
}

// Enqueue the full pipeline for <= QT queries already in device memory.
// d_fbits nullable.  Outputs are device pointers.
int32_t enqueue_search(msi_vs *vs, const float *d_queries, uint32_t nq, uint32_t k, const u64 *d_fbits,
                       uint64_t nbits, uint32_t *d_out_docids, float *d_out_dist, uint32_t *d_out_counts,
                       uint32_t *d_inexact) {
  msi_ctx *ctx = vs->ctx;
  hipStream_t st = ctx->stream;
  Small s = small_of(vs);
  const uint32_t slack = std::max<uint32_t>(12, k / 4);
  uint32_t kp = k + slack;
  if (kp > KP_MAX) kp = KP_MAX;
  if (k > KP_MAX) {
    msi_set_error("msi_vs_search: k=%u above the supported maximum %u", k, KP_MAX);
    return MSI_E_UNSUPPORTED;
  }
  // 1. queries
  hipLaunchKernelGGL(vs_prep_queries_kernel, dim3(16), dim3(256), 0, st, d_queries, nq, vs->dim, vs->KB,
                     vs->qfrag.as<float4>(), vs->qrow.as<float>(), s.qn, s.inv_qn, s.degth);
  MSI_HIP_TRY(hipMemsetAsync(s.overflow, 0, sizeof(uint32_t), st));
  // 2. filter
  const uint32_t *list = nullptr;
  const uint16_t *tmask = nullptr;
  const uint32_t *n_items_ptr = s.n_tiles;
  if (d_fbits && vs->n_rows) {
    MSI_TRY(vs->tmask.ensure(vs->n_tiles * sizeof(uint16_t)));
    MSI_TRY(vs->tlist.ensure(vs->n_tiles * sizeof(uint32_t)));
    MSI_HIP_TRY(hipMemsetAsync(s.n_items, 0, sizeof(uint32_t), st));
    const uint64_t padded = vs->n_tiles * 16;
    hipLaunchKernelGGL(vs_filter_tiles_kernel, dim3((uint32_t)((padded + 255) / 256)), dim3(256), 0, st,
                       vs->docids.as<uint32_t>(), vs->n_rows, d_fbits, nbits, vs->tmask.as<uint16_t>(),
                       vs->tlist.as<uint32_t>(), s.n_items);
    list = vs->tlist.as<uint32_t>();
    tmask = vs->tmask.as<uint16_t>();
    n_items_ptr = s.n_items;
  }
  ScanArgs sa;
  sa.tiles = vs->tiles.as<float4>();
  sa.inv_norm = vs->inv_norm.as<float>();
  sa.qfrag = vs->qfrag.as<float4>();
  sa.degth = s.degth;
  sa.n_items_ptr = n_items_ptr;
  sa.list = list;
  sa.tmask = tmask;
  sa.gkeys = vs->gkeys.as<u64>();
  sa.gcnt = s.gcnt;
  sa.overflow = s.overflow;
  sa.n_rows = vs->n_rows;
  sa.capg = vs->capg;
  sa.KB = vs->KB;
  sa.kp = kp;
  SelectArgs se;
  se.gkeys = vs->gkeys.as<u64>();
  se.gcnt = s.gcnt;
  se.capg = vs->capg;
  se.kp = kp;
  se.sel_keys = vs->sel_keys.as<u64>();
  se.sel_cnt = s.sel_cnt;
  se.theta = s.theta;
  // 3. sample pass -> thresholds.  Sample ~sqrt(K'*N) rows, skipped for small stores.
  const uint64_t n_tiles = vs->n_tiles;
  uint32_t stride = 1;
  if (n_tiles) {
    const double s_rows = sqrt((double)kp * (double)(n_tiles * 16)) * 2.0;
    const uint64_t s_tiles = std::max<uint64_t>(64, (uint64_t)(s_rows / 16.0));
    if (n_tiles >= s_tiles * 8) stride = (uint32_t)(n_tiles / s_tiles);
  }
  const float *theta = s.theta_inf;
  if (stride > 1) {
    sa.theta = s.theta_inf;
    sa.stride = stride;
    launch_scan(vs, sa);
    se.mode = 0;
    hipLaunchKernelGGL(vs_select_kernel, dim3(QT), dim3(SEL_THREADS), 0, st, se);
    theta = s.theta;
    vs->scan_launches++;
    vs->scan_tiles += n_tiles / stride;
  }
  // 4. main pass
  sa.theta = theta;
  sa.stride = 1;
  vs->scan_timer.begin(ctx);
  launch_scan(vs, sa);
  vs->scan_timer.end(ctx);
  vs->scan_launches++;
  vs->scan_tiles += n_tiles;
  // 5. select K' best per query
  se.mode = 1;
  hipLaunchKernelGGL(vs_select_kernel, dim3(QT), dim3(SEL_THREADS), 0, st, se);
  // 6. rescore with the reference arithmetic, order, prove exactness
  RescoreArgs ra;
  ra.tiles = vs->tiles.as<float4>();
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
  ra.eps = (2.0f * (float)vs->dpad + 32.0f) * 5.9604645e-8f;
  ra.out_docids = d_out_docids;
  ra.out_dist = d_out_dist;
  ra.out_counts = d_out_counts;
  ra.inexact = d_inexact;
  ra.overflow = s.overflow;
  if (k > 0)
    hipLaunchKernelGGL(vs_rescore_kernel, dim3(nq), dim3(SEL_THREADS),
                       KP_MAX * sizeof(u64) + (size_t)vs->dpad * sizeof(float), st, ra);
  MSI_HIP_TRY(hipGetLastError());
  return MSI_OK;
}

void launch_scan(msi_vs *vs, const ScanArgs &sa) {
  const size_t lds = scan_lds_bytes(vs->KB, vs->waves);
  if (vs->waves == 8)
    hipLaunchKernelGGL(vs_scan_kernel<8>, dim3(vs->scan_grid), dim3(8 * 64), lds, vs->ctx->stream, sa);
  else
    hipLaunchKernelGGL(vs_scan_kernel<4>, dim3(vs->scan_grid), dim3(4 * 64), lds, vs->ctx->stream, sa);
}

int32_t exhaustive_one(msi_vs *vs, uint32_t qj, uint32_t k, const u64 *d_fbits, uint64_t nbits,
                       uint32_t *d_out_docids, float *d_out_dist, uint32_t *d_out_count) {
  hipStream_t st = vs->ctx->stream;
  Small s = small_of(vs);
  MSI_TRY(vs->exh_keys.ensure(std::max<uint64_t>(1, vs->n_rows) * sizeof(u64)));
  if (vs->n_rows)
    hipLaunchKernelGGL(vs_exhaustive_kernel, dim3((uint32_t)((vs->n_rows + 255) / 256)), dim3(256),
                       (size_t)vs->dpad * sizeof(float), st, vs->tiles.as<float4>(), vs->norm.as<float>(),
                       vs->docids.as<uint32_t>(), vs->n_rows, vs->KB, vs->qrow.as<float>(), s.qn, qj, d_fbits,
                       nbits, vs->exh_keys.as<u64>());
  hipLaunchKernelGGL(vs_exhaustive_select_kernel, dim3(1), dim3(SEL_THREADS), 0, st, vs->exh_keys.as<u64>(),
                     (uint32_t)vs->n_rows, k, vs->docids.as<uint32_t>(), d_out_docids, d_out_dist, d_out_count);
  MSI_HIP_TRY(hipGetLastError());
  vs->exhaustive_reruns++;
  return MSI_OK;
}

}  // namespace

extern "C" {

int32_t msi_vs_create(msi_ctx *ctx, uint32_t dim, msi_vs **out) {
  if (!ctx || !out || dim == 0) {
    msi_set_error("msi_vs_create: invalid argument");
    return MSI_E_INVALID;
  }
  *out = nullptr;
  const uint32_t dpad = ((dim + 127) / 128) * 128;  // KB multiple of SCAN_GROUP
  const uint32_t KB = dpad / 16;
  uint32_t waves = SCAN_WAVES;
  if (scan_lds_bytes(KB, waves) > 160 * 1024) waves = 4;
  if (scan_lds_bytes(KB, waves) > 160 * 1024) {
    msi_set_error("msi_vs_create: dim %u needs %zu B of LDS for one query tile (max 163840)", dim,
                  scan_lds_bytes(KB, waves));
    return MSI_E_UNSUPPORTED;
  }
  DeviceGuard g(ctx->device);
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&vs_scan_kernel<8>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  if (e == hipSuccess)
    e = hipFuncSetAttribute(reinterpret_cast<const void *>(&vs_scan_kernel<4>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  if (e != hipSuccess) {
    msi_set_error("hipFuncSetAttribute(vs_scan) failed: %s", hipGetErrorString(e));
    return MSI_E_HIP;
  }
  msi_vs *vs = new msi_vs();
  vs->ctx = ctx;
  msi_ctx_retain(ctx);
  vs->dim = dim;
  vs->dpad = dpad;
  vs->KB = KB;
  vs->waves = waves;
  *out = vs;
  return MSI_OK;
}

void msi_vs_destroy(msi_vs *vs) {
  if (!vs) return;
  msi_ctx *ctx = vs->ctx;
  {
  std::lock_guard<std::mutex> lk(ctx->mu);
  DeviceGuard g(ctx->device);
  (void)hipStreamSynchronize(vs->ctx->stream);
  DevBuf *bufs[] = {&vs->tiles, &vs->norm, &vs->inv_norm, &vs->docids, &vs->qraw, &vs->qfrag, &vs->qrow,
                    &vs->qsmall, &vs->gkeys, &vs->gsmall, &vs->sel_keys, &vs->tmask, &vs->tlist, &vs->fbits,
                    &vs->out_docids, &vs->out_dist, &vs->out_small, &vs->exh_keys, &vs->rowtmp};
  for (DevBuf *b : bufs) b->release();
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

uint64_t msi_vs_len(const msi_vs *vs) { return vs ? vs->n_rows : 0; }
uint32_t msi_vs_dim(const msi_vs *vs) { return vs ? vs->dim : 0; }

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
  MSI_TRY(vs->qraw.ensure((size_t)QT * vs->dim * sizeof(float)));
  hipLaunchKernelGGL(vs_gather_row_kernel, dim3(ceil_div_u32(vs->dim, 256)), dim3(256), 0, st,
                     vs->tiles.as<float4>(), vs->KB, row, vs->dim, vs->qraw.as<float>());
  MSI_HIP_TRY(hipMemcpyAsync(out_row, vs->qraw.p, vs->dim * sizeof(float), hipMemcpyDeviceToHost, st));
  MSI_HIP_TRY(hipStreamSynchronize(st));
  *out_found = 1;
  return MSI_OK;
}

int32_t msi_vs_search_device(msi_vs *vs, const float *d_queries, uint32_t n_queries, uint32_t k,
                             const uint64_t *d_filter_bits, uint64_t filter_nbits, uint32_t *d_out_docids,
                             float *d_out_dist, uint32_t *d_out_counts, uint32_t *d_inexact) {
  if (!vs || !d_queries || n_queries == 0 || n_queries > QT || !d_out_docids || !d_out_dist || !d_out_counts) {
    msi_set_error("msi_vs_search_device: invalid argument (1..16 queries per call)");
    return MSI_E_INVALID;
  }
  std::lock_guard<std::mutex> lk(vs->ctx->mu);
  DeviceGuard g(vs->ctx->device);
  if (vs->n_rows == 0 || k == 0) {
    MSI_HIP_TRY(hipMemsetAsync(d_out_counts, 0, n_queries * sizeof(uint32_t), vs->ctx->stream));
    if (d_inexact) MSI_HIP_TRY(hipMemsetAsync(d_inexact, 0, n_queries * sizeof(uint32_t), vs->ctx->stream));
    return MSI_OK;
  }
  return enqueue_search(vs, d_queries, n_queries, k, (const u64 *)d_filter_bits, filter_nbits, d_out_docids,
                        d_out_dist, d_out_counts, d_inexact);
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
  MSI_TRY(vs->out_docids.ensure((size_t)QT * kk * sizeof(uint32_t)));
  MSI_TRY(vs->out_dist.ensure((size_t)QT * kk * sizeof(float)));
  for (uint32_t q0 = 0; q0 < n_queries; q0 += QT) {
    if (cancel && *cancel) {
      msi_set_error("msi_vs_search: cancelled");
      return MSI_E_CANCELLED;
    }
    const uint32_t nq = std::min<uint32_t>(QT, n_queries - q0);
    if (k == 0 || vs->n_rows == 0) {
      for (uint32_t j = 0; j < nq; ++j) out_counts[q0 + j] = 0;
      continue;
    }
    MSI_HIP_TRY(hipMemcpyAsync(vs->qraw.p, queries + (size_t)q0 * vs->dim, (size_t)nq * vs->dim * sizeof(float),
                               hipMemcpyHostToDevice, st));
    MSI_TRY(enqueue_search(vs, vs->qraw.as<float>(), nq, k, d_fbits, filter_nbits, vs->out_docids.as<uint32_t>(),
                           vs->out_dist.as<float>(), s.counts, s.inexact));
    uint32_t h_inexact[QT];
    MSI_HIP_TRY(hipMemcpyAsync(h_inexact, s.inexact, nq * sizeof(uint32_t), hipMemcpyDeviceToHost, st));
    MSI_HIP_TRY(hipStreamSynchronize(st));
    for (uint32_t j = 0; j < nq; ++j) {
      if (!h_inexact[j]) continue;
      if (cancel && *cancel) {
        msi_set_error("msi_vs_search: cancelled");
        return MSI_E_CANCELLED;
      }
      MSI_TRY(exhaustive_one(vs, j, k, d_fbits, filter_nbits, vs->out_docids.as<uint32_t>() + (size_t)j * k,
                             vs->out_dist.as<float>() + (size_t)j * k, s.counts + j));
    }
    MSI_HIP_TRY(hipMemcpyAsync(out_docids + (size_t)q0 * k, vs->out_docids.p, (size_t)nq * k * sizeof(uint32_t),
                               hipMemcpyDeviceToHost, st));
    MSI_HIP_TRY(hipMemcpyAsync(out_dist + (size_t)q0 * k, vs->out_dist.p, (size_t)nq * k * sizeof(float),
                               hipMemcpyDeviceToHost, st));
    MSI_HIP_TRY(hipMemcpyAsync(out_counts + q0, s.counts, nq * sizeof(uint32_t), hipMemcpyDeviceToHost, st));
    MSI_HIP_TRY(hipStreamSynchronize(st));
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

int32_t msi_vs_get_stats(const msi_vs *vs, msi_vs_stats *out) {
  if (!vs || !out) return MSI_E_INVALID;
  out->scan_launches = vs->scan_launches;
  out->scan_tiles = vs->scan_tiles;
  out->exhaustive_reruns = vs->exhaustive_reruns;
  out->bytes_per_tile = (uint64_t)vs->KB * 1024;
  return MSI_OK;
}

}  // extern "C"
