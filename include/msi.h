/*
 * msi.h — C ABI of libmsi.so: the MI355X-native query-time scoring path for
 * Meilisearch's `milli` (vector k-NN scan, typo-tolerant term lookup, dense
 * docid-set algebra, scoring arithmetic).
 *
 * This is the drop-in boundary (SURVEY.md §8 b).  There is no FFI in the
 * reference today: the seams are Rust method calls inside crates/milli.  Every
 * entry point below cites the reference interface (file:line under
 * /root/reference) that a Rust shim would route to it; INTEGRATION.md shows
 * the `extern "C"` block and the safe wrappers a milli maintainer would add.
 *
 * Conventions
 *  - every function returns int32_t status: 0 = MSI_OK, negative = MSI_E_*;
 *    msi_last_error() returns a thread-local, NUL-terminated description.
 *  - opaque handles are owned by the library and released by *_destroy.
 *  - inputs are borrowed for the duration of the call; outputs are caller
 *    allocated.  No exception ever crosses the boundary.
 *  - all entry points are thread-safe (callers are tokio spawn_blocking
 *    threads, crates/meilisearch/src/search/federated/perform.rs:224).
 *    Calls on one context serialise on the context's stream lock.
 *  - "_device" variants take device pointers and enqueue on the context's HIP
 *    stream without synchronising (multi-GPU gather, benchmarks).
 *  - there is NO CPU fallback inside the library: if no gfx950 device is
 *    usable msi_ctx_create fails with MSI_E_NO_DEVICE.
 */
#ifndef MSI_H
#define MSI_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MSI_ABI_VERSION 3   /* 2: msi_search_params grew (geo_strategy, geo_cache_size, index_view), round 3; 3: msi_vs_stats grew, msi_runtime_hw_queues, round 5 */

enum {
  MSI_OK = 0,
  MSI_E_INVALID = -1,     /* bad argument                                  */
  MSI_E_NO_DEVICE = -2,   /* no usable HIP device / wrong architecture     */
  MSI_E_HIP = -3,         /* a HIP runtime call failed                     */
  MSI_E_OOM = -4,         /* device or host allocation failed              */
  MSI_E_UNSUPPORTED = -5, /* valid request outside the implemented range   */
  MSI_E_CANCELLED = -6,   /* the caller's cancel flag was raised           */
  MSI_E_NOT_SORTED = -7,  /* input that must be sorted/unique is not       */
  MSI_E_INTERNAL = -8
};

typedef struct msi_ctx msi_ctx;   /* one per (process, GPU)                 */
typedef struct msi_vs msi_vs;     /* one vector store (one arroy/hannoy index) */
typedef struct msi_dict msi_dict; /* one words-FST dictionary in HBM        */
typedef struct msi_bits msi_bits; /* a pool of dense docid sets in HBM      */

/* ------------------------------------------------------------------ context */

int32_t msi_abi_version(void);
const char *msi_last_error(void);

/* device < 0: use $LOCAL_RANK if set, else device 0. */
int32_t msi_ctx_create(int32_t device, msi_ctx **out);
void msi_ctx_destroy(msi_ctx *ctx);
/* The context's main hipStream_t (as void*): vector stores, docid sets, ranking.
 * Dictionaries (msi_dict_*) run on a second stream of the context so that the
 * VALU-bound typo lookup overlaps the HBM-bound scan; msi_ctx_synchronize waits
 * for both. */
void *msi_ctx_stream(msi_ctx *ctx);
int32_t msi_ctx_synchronize(msi_ctx *ctx);
int32_t msi_ctx_device(msi_ctx *ctx);
/* When enabled, the dominant kernels (vs_scan's main pass, dict_match) are
 * bracketed by HIP events on the context stream; msi_vs_scan_time /
 * msi_dict_match_time synchronise and return the accumulated kernel time. */
int32_t msi_ctx_set_profiling(msi_ctx *ctx, int32_t enable);
/* Hardware queues the HIP runtime maps this process's streams onto (GPU_MAX_HW_QUEUES; the runtime's default is 4).
 * The keyword searches' command-list rounds run on 16 streams and rounds that share a queue serialise, so libmsi
 * sets GPU_MAX_HW_QUEUES=16 when it is LOADED (a library constructor: before the runtime's first call reads it)
 * unless the host process already chose a value or set MSI_KEEP_HW_QUEUES=1.  Below 16 the keyword leg measured less
 * than half its throughput; an integrator checks this once at start-up (nothing is written to stderr unasked:
 * MSI_VERBOSE=1 prints the warning).
 * What the return value means: the number the ENVIRONMENT asks of the runtime.  It is what the runtime uses only if the
 * runtime had not started when the variable was set — true for a process that links libmsi (the constructor runs before
 * main) or sets the variable itself; NOT guaranteed for a process that dlopen()s libmsi after its first HIP call (another
 * HIP user, torch used before the import): then the runtime keeps the value it started with.  msi_runtime_hw_queues_source
 * says where the number comes from: 0 = the host's own setting (or MSI_KEEP_HW_QUEUES: the runtime's default 4 if unset),
 * 1 = set by libmsi's constructor at load time (in effect only under the condition above).
 * dlopen() from an already multi-threaded process: setenv is not thread-safe against concurrent getenv — such a host sets
 * GPU_MAX_HW_QUEUES=16 in its own environment before it starts and MSI_KEEP_HW_QUEUES=1, and the constructor touches
 * nothing. */
int32_t msi_runtime_hw_queues(void);
int32_t msi_runtime_hw_queues_source(void);

/* ------------------------------------------------- S1: vector k-NN (cosine) */
/*
 * Replaces VectorStore::nns_by_vector / nns_by_item for ONE store
 * (crates/milli/src/vector/store.rs:615-675, 1036-1093): the `limit` nearest
 * rows to each query under arroy/hannoy's cosine distance
 *     d = pn*qn > f32::EPSILON ? (1 - pq/(pn*qn)) / 2 : 0        (all f32)
 * restricted to an optional candidate set, returned ascending by
 * (distance, docid).  The scan is exact (the reference's linear mode,
 * store.rs:22-27,1079-1080); distances are the reference's scalar f32
 * arithmetic (sequential mul+add dot product), so results are reproducible
 * bit for bit on any device count.
 */
int32_t msi_vs_create(msi_ctx *ctx, uint32_t dim, msi_vs **out);
/* Storage type of the rows in HBM.  MSI_VS_BF16 is a build-side choice (the
 * reference stores f32 or 1-bit quantised vectors, store.rs:1095-1109): rows are
 * rounded to bf16 (nearest even) at upload, every distance is the reference
 * arithmetic on the ROUNDED rows, HBM bytes per row halve (BASELINE.json config 5). */
enum { MSI_VS_F32 = 0, MSI_VS_BF16 = 1 };
int32_t msi_vs_create_typed(msi_ctx *ctx, uint32_t dim, int32_t storage, msi_vs **out);
void msi_vs_destroy(msi_vs *vs);

/* Replace the store's contents.  `docids[n_rows]` strictly ascending (one row
 * per document and store, as in arroy), `rows` row-major [n_rows][dim] f32.
 * Host pointers; data is copied to HBM and re-tiled for the scan kernel. */
int32_t msi_vs_upload(msi_vs *vs, const uint32_t *docids, const float *rows,
                      uint64_t n_rows);
/* Same with device pointers (synchronises the context stream before return). */
int32_t msi_vs_upload_device(msi_vs *vs, const uint32_t *d_docids,
                             const float *d_rows, uint64_t n_rows);

/* SURVEY §8 f2 — apply a committed update to the store without sending it over PCIe again (what
 * update/new/indexer/write.rs:65-74,157 does to arroy / hannoy: del_item / add_item per document, then a rebuild):
 * the documents of `remove_docids` leave the store (unknown docids are ignored, as del_item of a missing item is),
 * the rows of `add_docids` enter it — a docid the store already holds is REPLACED.  Both lists strictly ascending;
 * `add_rows` row-major [n_add][dim] f32.  Host pointers, borrowed for the call.  Only the two lists, the added rows
 * and a 4-byte-per-row gather map travel; the device tiles the added rows and re-gathers the store into its next
 * buffer in one pass (2 x store bytes of HBM traffic; the store occupies twice its size during the call).
 * Searches on the same context are serialised with it; results afterwards equal those of msi_vs_upload of the
 * resulting rows bit for bit. */
int32_t msi_vs_update(msi_vs *vs, const uint32_t *remove_docids, uint64_t n_remove, const uint32_t *add_docids,
                      const float *add_rows, uint64_t n_add);

uint64_t msi_vs_len(const msi_vs *vs);
uint32_t msi_vs_dim(const msi_vs *vs);
/* Queries answered by ONE sweep of the store through HBM (16, 32 or 48: as many
 * 16-query MFMA tiles as the store's dimension leaves room for in LDS).  Larger
 * batches are processed in chunks of this size; a micro-batching caller should
 * aim for multiples of it. */
uint32_t msi_vs_max_batch(const msi_vs *vs);

/* store.rs:1004-1034 (nns_by_item reads the item's stored vector first). */
int32_t msi_vs_get_vector(msi_vs *vs, uint32_t docid, float *out_row,
                          int32_t *out_found);

/*
 * queries      [n_queries][dim] f32, row major (host)
 * k            results per query (limit = from + length, vector_sort.rs:68-71)
 * filter_bits  nullable; bit i (LSB-first in 64-bit words) set = docid i allowed
 *              (dense form of the RoaringBitmap `filter`, store.rs:641-643)
 * cancel       nullable; polled between kernel phases; non-zero → MSI_E_CANCELLED
 *              (hannoy's cancellation closure, store.rs:1085-1086)
 * out_docids   [n_queries][k]   out_dist [n_queries][k]   out_counts [n_queries]
 */
int32_t msi_vs_search(msi_vs *vs, const float *queries, uint32_t n_queries,
                      uint32_t k, const uint64_t *filter_bits,
                      uint64_t filter_nbits, const volatile int32_t *cancel,
                      uint32_t *out_docids, float *out_dist,
                      uint32_t *out_counts);

/* nns_by_item for one store (store.rs:615-637,980-1034): the query is the stored vector of
 * `docid` (*out_found = 0 and no results when the store has none).  Similar::execute
 * (search/similar.rs:67-153) passes a filter without the item itself. */
int32_t msi_vs_search_by_item(msi_vs *vs, uint32_t docid, uint32_t k,
                              const uint64_t *filter_bits, uint64_t filter_nbits,
                              uint32_t *out_docids, float *out_dist, uint32_t *out_count,
                              int32_t *out_found);

/* SURVEY §8 f2 — the dictionaries straight from milli's `fst::Set` bytes (main["words-fst"],
 * crates/milli/src/index.rs:1225-1243 `Index::words_fst`; facet-id-string-fst,
 * search/facet/search.rs:122-190), so that the shim passes `set.as_fst().as_bytes()` (a borrow of the LMDB
 * page) instead of streaming every key through Rust: a host-side decoder of the `fst` 0.4 format version 3
 * (bounds-checked, checksum-verified; anything malformed -> MSI_E_INVALID, another version ->
 * MSI_E_UNSUPPORTED, and the caller keeps its own FST path).
 * msi_fst_decode: keys in stream (= byte-lexicographic) order, flat as msi_dict_create takes them.  With
 * out_concat = NULL only *out_n_words / *out_n_bytes are produced (size the buffers, then call again);
 * out_offsets has cap_words + 1 entries. */
#define MSI_FST_SKIP_CHECKSUM 1u /* flags: the caller verified the footer CRC already */
int32_t msi_fst_decode(const uint8_t *fst, size_t len, uint32_t flags, uint8_t *out_concat, uint64_t cap_bytes,
                       uint32_t *out_offsets, uint32_t cap_words, uint32_t *out_n_words, uint64_t *out_n_bytes);
/* = msi_fst_decode + msi_dict_create / msi_dict_create_values */
int32_t msi_dict_create_from_fst(msi_ctx *ctx, const uint8_t *fst, size_t len, msi_dict **out);
int32_t msi_dict_create_values_from_fst(msi_ctx *ctx, const uint8_t *fst, size_t len, msi_dict **out);

/* Micro-batching of concurrent callers (each tokio spawn_blocking search thread
 * calls msi_vs_search with ONE query): with max_wait_us > 0, unfiltered calls that
 * arrive within that window are fused into one HBM sweep (up to msi_vs_max_batch()
 * queries; the largest k of the group is computed and every caller receives its
 * prefix, which is its exact answer).  0 (default) = every call runs on its own. */
int32_t msi_vs_set_microbatch(msi_vs *vs, uint32_t max_wait_us);
/* How finely a full sweep of the store is cut into workgroups: n = 1 (default) = one persistent workgroup per CU slot, each
 * streaming 1 / grid of the rows for the whole sweep (1.4 ms at 10 M x 768); n = 2..64 = n times as many workgroups, each n
 * times shorter.  For a host that runs the vector searches BESIDE other device work of the same process (a server's
 * concurrent queries: one query's vector search beside the others' keyword searches — execute_hybrid, hybrid.rs:264-340,
 * runs both for every hybrid query): short workgroups let that work's kernels in between instead of behind the sweep; the sweep on its own loses ~6 % at n = 16 (the query fragments are staged
 * into LDS once per workgroup).  Answers do not depend on it. */
int32_t msi_vs_set_sweep_split(msi_vs *vs, uint32_t n);
int32_t msi_vs_microbatch_stats(msi_vs *vs, uint64_t *out_fused_calls,
                                uint64_t *out_fused_sweeps);

/* Device-pointer variant: all pointers are device memory.
 * CONTRACT (since ABI 3, round 5): the call ALWAYS ANSWERS, as VectorStore::nns_by_vector does (store.rs:638-675) — the
 * queries whose top-k the first sweep could not prove are gathered and re-run by the library, level by level and finally
 * exhaustively, so on return every list is the exact one and `d_inexact[n_queries]` (nullable) reads 0 everywhere.  The
 * price is ONE hipStreamSynchronize of msi_ctx_stream() inside every call (the proof flags have to be read): the call is
 * not asynchronous — when it returns the outputs are complete on the stream (a re-run's copies may still be in flight:
 * order later work on msi_ctx_stream() or msi_ctx_synchronize()).  A host that relied on the call returning before the
 * sweeps ran must move its own work onto another thread / stream.
 * MSI_VS_DEVICE_RERUN=0 in the environment restores the OLD contract for the whole process: work is only enqueued, nothing
 * is synchronised, and `d_inexact` receives 1 where the exactness proof failed — the caller must re-run those queries
 * (msi_vs_search).  MSI_VS_PIPELINE=1 (two-stream chunk pipeline, f32 contraction only) is honoured under the old contract
 * only: it flags unproven queries, it does not re-run them. */
int32_t msi_vs_search_device(msi_vs *vs, const float *d_queries,
                             uint32_t n_queries, uint32_t k,
                             const uint64_t *d_filter_bits,
                             uint64_t filter_nbits, uint32_t *d_out_docids,
                             float *d_out_dist, uint32_t *d_out_counts,
                             uint32_t *d_inexact);

/*
 * Host-side merge of `n_lists` ascending result lists (list l occupies
 * docids/dist[l*list_stride .. + counts[l])) into the `k_out` best by
 * (distance, docid): the concatenate + sort_unstable_by_key tail of
 * nns_by_vector over an embedder's stores (store.rs:1059,1090), and the final
 * step of a row-sharded search (one list per GPU after the all-gather).
 * Returns the number of entries written.  No device work.
 */
uint32_t msi_merge_topk(const uint32_t *docids, const float *dist,
                        const uint32_t *counts, uint32_t n_lists,
                        uint32_t list_stride, uint32_t k_out,
                        uint32_t *out_docids, float *out_dist);

/* The same merge on the device for a row-sharded search: after the all-gather (RCCL
 * over xGMI) every rank holds d_docids/d_dist [n_lists][n_queries][k] and d_counts
 * [n_lists][n_queries]; one workgroup per query writes the k best by (distance, docid).
 * n_lists*k <= 2048.  Enqueued on msi_ctx_stream(), not synchronised. */
int32_t msi_merge_topk_device(msi_ctx *ctx, const uint32_t *d_docids, const float *d_dist,
                              const uint32_t *d_counts, uint32_t n_lists, uint32_t n_queries,
                              uint32_t k, uint32_t *d_out_docids, float *d_out_dist,
                              uint32_t *d_out_counts);

/* ---------------------------------------------------------------- multi-GPU (SURVEY §8 e) */
/*
 * The reference is ONE server process fanning searches out from spawn_blocking threads
 * (crates/meilisearch/src/search/federated/perform.rs:224): the multi-device entry points are part of this ABI.
 * msi_group_create: one context per device of `devices` + one RCCL communicator per device (ncclCommInitAll) — the
 * in-process form.  msi_group_create_rank: the one-process-per-GPU form (bench.py under torchrun): rank 0 obtains
 * msi_group_unique_id(), the launcher hands the 128 bytes to every rank, each rank joins with its own context.
 * RCCL is loaded with dlopen here: a box without librccl.so gets MSI_E_UNSUPPORTED from these calls and nothing else
 * changes.
 * msi_vs_group (in-process): MSI_GROUP_REPLICATE — every device holds all rows and answers its slice of a query
 * batch, no exchange step (query sharding, the default for stores that fit one GPU: 10 M x 768 f32 = 31 GB of 288);
 * MSI_GROUP_SHARD_ROWS — contiguous row ranges per device, every device scans its rows for the whole batch, ONE
 * ncclAllGather of the packed per-device lists ({distance, docid}[B][k] + counts[B]) over xGMI, k-way merge on the
 * device (msi_merge_topk_device): the concatenate + sort_unstable_by_key of store.rs:1059,1090.  devices x k <= 2048.
 * Results are identical to a single-device store holding all the rows (tie rule included) in both modes.
 * msi_group_allgather (per-rank form): d_recv := the `bytes` of every rank's d_send in rank order, enqueued on the
 * context's stream.
 */
typedef struct msi_group msi_group;
typedef struct msi_vs_group msi_vs_group;
enum { MSI_GROUP_REPLICATE = 0, MSI_GROUP_SHARD_ROWS = 1 };
int32_t msi_group_create(const int32_t *devices, uint32_t n, msi_group **out);
int32_t msi_group_unique_id(uint8_t out_id[128]);
int32_t msi_group_create_rank(msi_ctx *ctx, uint32_t rank, uint32_t world, const uint8_t id[128], msi_group **out);
void msi_group_destroy(msi_group *group);
uint32_t msi_group_size(const msi_group *group);
msi_ctx *msi_group_ctx(msi_group *group, uint32_t i);
int32_t msi_group_allgather(msi_group *group, const void *d_send, size_t bytes, void *d_recv);
int32_t msi_vs_group_create(msi_group *group, uint32_t dim, int32_t storage, int32_t mode, msi_vs_group **out);
void msi_vs_group_destroy(msi_vs_group *vs);
int32_t msi_vs_group_upload(msi_vs_group *vs, const uint32_t *docids, const float *rows, uint64_t n_rows);
int32_t msi_vs_group_search(msi_vs_group *vs, const float *queries, uint32_t n_queries, uint32_t k, uint32_t *out_docids,
                            float *out_dist, uint32_t *out_counts);

/* Introspection for benchmarks/tests. */
typedef struct msi_vs_stats {
  uint64_t scan_launches;      /* vs_scan kernel launches so far (sample + full sweeps) */
  uint64_t scan_tiles;         /* 16-row tiles streamed by those launches  */
  uint64_t exhaustive_reruns;  /* queries that needed the exhaustive path  */
  uint64_t bytes_per_tile;     /* algorithmic HBM bytes per tile           */
  /* (ABI 2) host entry point, bf16x2 stores: queries the 96-query bf16x2 sweep could not prove and that took the bf16x3
   * second opinion; sweeps that ran bf16x3 FIRST because most recent queries needed it (clustered data); bf16x2 sweeps */
  uint64_t second_opinion_queries, x3_first_sweeps, x2_sweeps;
  /* sweeps by level of effort (msi_vs.hip, msi_vs::level): the store's contraction with the usual K' | the same with
   * K' = 2048 rescored candidates | bf16x3 with K' = 2048 */
  uint64_t level_sweeps[3];
  /* (ABI 3) the int8 candidate sweep (level 0 of an f32 store that keeps the int8 copy of its rows: msi_vs.hip): algorithmic
   * HBM bytes per 16-row tile of it (0: the store has no copy), full sweeps of it so far, queries that msi_vs_search_device
   * re-ran at a higher level of effort itself, queries per sweep of the copy / of the f32 rows, tiles of the copy streamed
   * (sample + full sweeps: the part of scan_tiles that read the copy) */
  uint64_t i8_bytes_per_tile, i8_sweeps, device_rerun_queries, i8_queries_per_sweep, f32_queries_per_sweep, i8_scan_tiles;
} msi_vs_stats;
int32_t msi_vs_get_stats(const msi_vs *vs, msi_vs_stats *out);
/* Test instrumentation: the fast scan's raw scores (dot / |row|; -inf for padding)
 * of every row for <= msi_vs_max_batch() host queries — out_scores [n_queries][len] —
 * and the bound the exactness proof assumes for |fast cos - reference cos|. */
int32_t msi_vs_debug_fast_scores(msi_vs *vs, const float *queries, uint32_t n_queries,
                                 float *out_scores, float *out_eps);
/* Accumulated duration of the full-sweep vs_scan launches recorded while
 * profiling was enabled (HIP events on the launch stream); resets the counters. */
int32_t msi_vs_scan_time(msi_vs *vs, uint64_t *out_launches, double *out_ms_total);
/* The last filtered search of this store, as the device counted it (synchronises the store's streams):
 *   out[0] items its sweeps visited (an item = 16 rows of one MFMA tile; x 16 x the row's bytes = what a sweep moves),
 *   out[1] allowed rows, out[2] items written as COMPACTED allowed rows (gathered at 64-byte-sector granularity),
 *   out[3] items written as whole 16-row tiles that hold an allowed row (streamed).
 * hannoy's linear mode scores only the candidates (vector/store.rs:1079-1080); a sweep here visits the allowed rows plus,
 * where a region of the store is dense in them, their tile neighbours (msi_vs.hip: vs_filter_rows_kernel). */
int32_t msi_vs_filter_stats(msi_vs *vs, uint64_t out[4]);

/* ----------------------------------- S1': binary-quantised vector stores */
/*
 * A binary-quantised embedder (`binaryQuantized: true`) lives in arroy's BinaryQuantizedCosine / hannoy's Hamming
 * databases (crates/milli/src/vector/store.rs:1095-1109): one sign bit per dimension.  msi_bq keeps those bits in HBM as
 * bit planes (dim / 8 bytes per row and sweep: 32x less traffic than the f32 rows) and answers nns_by_vector exactly:
 * the query is quantised like a row, rows are ranked by Hamming distance, ties by ascending docid, at most k <= 2048.
 *   quantisation  bit = (x > 0) — PINNED by the reference (tests/vector/binary_quantized.rs:67-135: a stored
 *                 [-1.2, -2.3, 3.2] reads back as [0, 0, 1]); msi_bq_get_vector returns that 0.0 / 1.0 form;
 *   distance      out_dist = hamming / dim (the cosine distance (1 - cos)/2 of the +-1 vectors; hannoy's Hamming is the
 *                 same count un-normalised) — restated from the crates' published definitions, NOT pinned by any
 *                 literal of the reference: parity unpinned for the value, pinned for the order it induces only as far
 *                 as both crates rank by the Hamming distance.
 * docids strictly ascending as for msi_vs; filter_bits as in msi_vs_search.
 */
typedef struct msi_bq msi_bq;
int32_t msi_bq_create(msi_ctx *ctx, uint32_t dim, msi_bq **out);
void msi_bq_destroy(msi_bq *bq);
int32_t msi_bq_upload(msi_bq *bq, const uint32_t *docids, const float *rows, uint64_t n_rows);
int32_t msi_bq_upload_device(msi_bq *bq, const uint32_t *d_docids, const float *d_rows, uint64_t n_rows);
uint64_t msi_bq_len(const msi_bq *bq);
uint32_t msi_bq_dim(const msi_bq *bq);
int32_t msi_bq_get_vector(msi_bq *bq, uint32_t docid, float *out_row, int32_t *out_found);
int32_t msi_bq_search(msi_bq *bq, const float *queries, uint32_t n_queries, uint32_t k, const uint64_t *filter_bits,
                      uint64_t filter_nbits, uint32_t *out_docids, float *out_dist, uint32_t *out_counts);

/* --------------------------------------------- S2: typo-tolerant term lookup */
/*
 * Replaces find_one_typo_derivations / find_one_two_typo_derivations
 * (crates/milli/src/search/new/query_term/compute_derivations.rs:75-168):
 * Levenshtein-DFA(transposition=1 edit) ∩ words FST, with the first-letter
 * rule and the 150/50 caps (search/new/limits.rs:7-9).
 */
int32_t msi_dict_create(msi_ctx *ctx, const uint8_t *words_concat,
                        const uint32_t *offsets /* n_words+1 */,
                        uint32_t n_words /* byte-lexicographic, unique, UTF-8 */,
                        msi_dict **out);
void msi_dict_destroy(msi_dict *dict);
uint32_t msi_dict_len(const msi_dict *dict);

typedef struct msi_typo_query {
  const uint8_t *word; /* normalised query word, UTF-8, not NUL terminated */
  uint32_t len;        /* bytes, 1..250 (MAX_WORD_LENGTH, milli/src/lib.rs) */
  uint8_t max_typos;   /* 1 or 2 (budget 0 never reaches the dictionary)   */
  uint8_t is_prefix;   /* build_prefix_dfa vs build_dfa (search/mod.rs:565-577) */
  uint16_t _pad;
} msi_typo_query;

/*
 * out_one_idx [n][cap_one], out_two_idx [n][cap_two]: dictionary indices in
 * ascending (= byte-lexicographic = fst stream) order; out_*_cnt [n].
 * cap_one/cap_two are MAX_ONE_TYPO_COUNT / MAX_TWO_TYPOS_COUNT (150 / 50).
 */
int32_t msi_dict_lookup(msi_dict *dict, const msi_typo_query *queries,
                        uint32_t n, uint32_t cap_one, uint32_t cap_two,
                        uint32_t *out_one_idx, uint32_t *out_one_cnt,
                        uint32_t *out_two_idx, uint32_t *out_two_cnt);

/* Facet search (crates/milli/src/search/facet/search.rs:122-190): the values of one facet's FST that
 * `fst.search(build_dfa(query, typos, is_prefix = true))` streams — every value with a prefix within max_typos
 * (0..2) edits of the query, distance 0 included, no first-letter rule, no per-class caps — as indices in stream
 * order.  The same derivation kernel answers it: the values are staged behind a common sentinel byte, which
 * makes all first letters equal.  max_typos is the caller's (query.len() < 5: 0, < 9: 1, else 2 — bytes here,
 * search.rs:147-155).  *out_truncated = 1 when more than `cap` (or more than 4096 per distance class) matched. */
int32_t msi_dict_create_values(msi_ctx *ctx, const uint8_t *values_concat, const uint32_t *offsets,
                               uint32_t n_values, msi_dict **out);
int32_t msi_dict_search_values(msi_dict *dict, const uint8_t *query, uint32_t len, uint32_t max_typos,
                               uint32_t cap, uint32_t *out_idx, uint32_t *out_n, int32_t *out_truncated);

/* Micro-batching of concurrent callers (every search derives the typos of its few words
 * with one small lookup): with max_wait_us > 0, calls of fewer than `target_words` words
 * that arrive within that window and use the same caps are fused into one launch. */
int32_t msi_dict_set_microbatch(msi_dict *dict, uint32_t max_wait_us, uint32_t target_words);
int32_t msi_dict_microbatch_stats(msi_dict *dict, uint64_t *out_fused_calls,
                                  uint64_t *out_fused_launches);

/* Packed device-side form: words are concatenated in `d_qbytes`, described by
 * d_qoff[n+1], d_qflags[n] = max_typos | is_prefix<<2.  Outputs are device
 * memory; work is enqueued on msi_ctx_stream() and NOT synchronised. */
int32_t msi_dict_lookup_device(msi_dict *dict, const uint8_t *d_qbytes,
                               const uint32_t *d_qoff, const uint8_t *d_qflags,
                               uint32_t n, uint32_t cap_one, uint32_t cap_two,
                               uint32_t *d_out_one_idx, uint32_t *d_out_one_cnt,
                               uint32_t *d_out_two_idx, uint32_t *d_out_two_cnt);

/* HBM posting cache of one index version (owned by its dictionary handle: both are re-staged when Index::updated_at
 * moves, so the cache can never serve a stale posting).  msi_keyword_search_ranked decodes the stored
 * CboRoaringBitmap values of word_docids / word_fid_docids / word_position_docids / word_pair_proximity_docids /
 * field_id_word_count_docids and the word-prefix databases on the device; with the cache enabled, the first search
 * that reads a key leaves its bytes in HBM (`capacity_bytes` of device memory, allocated here; when it is full new
 * keys are simply not cached) and every later search decodes that key from HBM — no host copy, no PCIe transfer.
 * The role LMDB's page cache + DatabaseCache (search/new/db_cache.rs) play for the reference.
 * Keys are 128-bit hashes of (database, key); the value length is checked as well.
 * stats: [hits, misses, bytes used, capacity]. */
int32_t msi_dict_enable_posting_cache(msi_dict *dict, uint64_t capacity_bytes);
int32_t msi_dict_posting_cache_stats(msi_dict *dict, uint64_t out[4]);
/* Everything the SEARCHES left in the cache is forgotten (what msi_dict_stage_postings staged stays: it is the index).
 * Only while no search on `dict` is in flight. */
int32_t msi_dict_reset_posting_cache(msi_dict *dict);

/* Staging the postings at index-open ("the FST bytes and document embeddings are staged once into HBM": the postings too).
 * The reference reads a posting from the LMDB mmap when a search first asks for it (db_cache.rs:50-84: one lookup, no
 * copy); without staging, msi_keyword_search_ranked does the same through the index vtable and keeps the bytes in the HBM
 * cache.  With the databases staged — the shim walks word_docids, word_fid_docids, word_position_docids (and whatever else
 * it wants resident) once when the index opens or its updated_at moves — a search's read is one probe of the cache's host
 * table and its decode reads HBM: no callback, no host copy, no PCIe, for the first search as for the millionth.
 * A value is exactly what the vtable's callback of that database would hand over for that key:
 *   db 1  word_docids(word = key1, original = x)        x = 1: word_docids ∪ exact_word_docids, x = 0: word_docids only
 *   db 2  word_pair_proximity_docids(proximity = x, left = key1, right = key2)
 *   db 3  word_fid_docids(word = key1, fid = x)
 *   db 4  word_position_docids(word = key1, position = x)
 *   db 5  field_id_word_count_docids(fid = x, count = y)
 * (n = 0: "no such key" is remembered as well).  Values of <= 7 docids (CboRoaringBitmap's raw form) stay on the host.
 * Consecutive values that point at the SAME bytes (word_docids under x = 0 and x = 1 of an index without exact attributes)
 * share one body in HBM.
 * index_view: msi_search_params::index_view of the searches that will read them (0: the index as it is).
 * Thread-safe: several threads may stage different slices at once, searches may run meanwhile.  One call reserves its
 * bodies' HBM in one piece and copies them with one transfer: hand over megabytes per call, not single values.
 * out_counts (nullable): [bodies staged in HBM, values kept on the host, keys the cache already knew]. */
enum { MSI_DB_WORD_DOCIDS = 1, MSI_DB_WORD_PAIR_PROXIMITY = 2, MSI_DB_WORD_FID = 3, MSI_DB_WORD_POSITION = 4,
       MSI_DB_FIELD_ID_WORD_COUNT = 5 };
typedef struct msi_staged_posting {
  uint32_t db;
  const uint8_t *key1;
  uint32_t key1_len;
  const uint8_t *key2;
  uint32_t key2_len;
  uint64_t x, y;
  const uint8_t *bytes; /* the stored CboRoaringBitmap value */
  size_t n;
} msi_staged_posting;
int32_t msi_dict_stage_postings(msi_dict *dict, uint64_t index_view, const msi_staged_posting *values,
                                uint64_t n_values, uint64_t out_counts[3]);
/* Every key of the databases in db_mask (bit MSI_DB_*) of that view has been staged: a key the cache does not hold does
 * not exist — msi_keyword_search_ranked answers "absent" itself instead of asking the vtable. */
int32_t msi_dict_stage_complete(msi_dict *dict, uint64_t index_view, uint32_t db_mask);
/* [bodies staged in HBM, values kept on the host, stored bytes of the staged bodies, reads answered "absent" by a complete db] */
int32_t msi_dict_staged_stats(msi_dict *dict, uint64_t out[4]);

typedef struct msi_dict_stats {
  uint64_t lookup_launches;
  uint64_t pairs_scanned;   /* (query, word) pairs that reached a kernel lane */
  uint64_t dict_bytes;      /* HBM bytes of the staged dictionary            */
} msi_dict_stats;
int32_t msi_dict_get_stats(const msi_dict *dict, msi_dict_stats *out);
int32_t msi_dict_match_time(msi_dict *dict, uint64_t *out_launches, double *out_ms_total);

/* ------------------------------------------------- S3: dense docid-set algebra */
/*
 * Device replacement for the RoaringBitmap algebra of the ranking-rule graph
 * (compute_query_term_subset_docids, resolve_query_graph.rs:33-59; the ∩/∪/−
 * of visit_path_condition, graph_based_ranking_rule.rs:383-437; bucket_sort's
 * universe bookkeeping, bucket_sort.rs:23-343).  A pool holds `n_slots` dense
 * sets of `n_docs` bits each; slot ids are the handles.
 */
int32_t msi_bits_create(msi_ctx *ctx, uint64_t n_docs, uint32_t n_slots,
                        msi_bits **out);
void msi_bits_destroy(msi_bits *pool);
/* By default a pool works on its context's stream (in order with the vector scan that may read one of its
 * slots as a filter).  A pool that serves one keyword search at a time can take a private stream so that
 * many searches (one pool each, one caller thread each) are in flight on the device together.  Call right
 * after msi_bits_create. */
int32_t msi_bits_use_private_stream(msi_bits *pool);
/* slot := {docids}; docids need not be sorted. */
int32_t msi_bits_set_from_docids(msi_bits *pool, uint32_t slot,
                                 const uint32_t *docids, uint64_t n);
/* slot := decode of a CboRoaringBitmapCodec value
 * (heed_codec/roaring_bitmap/cbo_roaring_bitmap_codec.rs:53-85). */
/* slot(first_slot + i * slot_stride) := the documents of the i-th device-resident list
 * (d_docids[i * list_stride .. + min(d_counts[i], list_stride)); ids >= n_docs are ignored) — e.g. the top-k of a
 * vector search as the universes of a ranking-rule rerank, without a trip through the host.  The lists must have
 * been produced on the pool's stream (its context's stream unless msi_bits_use_private_stream). */
int32_t msi_bits_set_from_docid_lists_device(msi_bits *pool, uint32_t first_slot, uint32_t slot_stride,
                                             const uint32_t *d_docids, uint32_t list_stride,
                                             const uint32_t *d_counts, uint32_t n_lists);
int32_t msi_bits_set_from_cbo(msi_bits *pool, uint32_t slot,
                              const uint8_t *bytes, size_t len);
/* SURVEY §8 f3 — the Sort / Asc / Desc ranking rules (crates/milli/src/search/new/sort.rs:95-233) without a facet
 * database walk per query: one u32 ORDER KEY per document, resident in HBM.  key[docid] = rank of the facet value
 * the rule's iteration meets first for that document (ascending_facet_sort / descending_facet_sort over
 * facet_id_f64_docids, then facet_id_string_docids: numbers, then strings, each in the rule's direction; for a
 * document with several values the first one met, i.e. its smallest for `asc`, its largest for `desc`),
 * 0xFFFFFFFF = no value.  Built by the shim once per (index update, field, direction); the rank -> value table
 * stays with the shim, which turns the key of a hit back into ScoreDetails::Sort{value}.
 * msi_bits_order_next = the rule's next_bucket: bucket := the documents of `universe` that share its smallest
 * key, universe -= bucket; *out_key that key (0xFFFFFFFF: what was left has no value — the rule's last, Null,
 * bucket), *out_count = |bucket| (0 only for an empty universe). */
typedef struct msi_doc_keys msi_doc_keys;
int32_t msi_doc_keys_create(msi_ctx *ctx, const uint32_t *keys /* [n_docs] */, uint64_t n_docs, msi_doc_keys **out);
void msi_doc_keys_destroy(msi_doc_keys *keys);
int32_t msi_bits_order_next(msi_bits *pool, const msi_doc_keys *keys, uint32_t universe, uint32_t bucket,
                            uint32_t *out_key, uint64_t *out_count);
/* SURVEY §8 a10 / f3 — the `distinct` attribute (crates/milli/src/search/new/distinct.rs:19-62, applied by
 * maybe_add_to_results, bucket_sort.rs:399-415, and by the rule-less path bucket_sort.rs:61-92) without a facet
 * database walk per candidate: the facet values of the distinct field, PER DOCUMENT, resident in HBM as CSR —
 * value_ids[offsets[d] .. offsets[d+1]) are the ids (< n_values) of the values of document d, numbers and strings
 * alike (facet_number_values + facet_string_values of field_id_docid_facet_f64s / _strings); two documents share
 * a value of the field iff they share an id.  Built by the shim once per (index update, field).
 *
 * msi_bits_distinct = apply_distinct_rule: `remaining` := the candidates the reference's ascending-docid loop
 * keeps — candidate d is kept iff no smaller kept candidate shares a value with it (a document without a value
 * is always kept) — i.e. the lexicographically first maximal independent set of the "shares a value" graph,
 * computed in parallel rounds (a candidate that is the smallest undecided holder of each of its values is kept;
 * the holders of its values are dropped; one round for a single-valued field, a sequential single-thread kernel
 * finishes pathological chains after MSI_DISTINCT_MAX_ROUNDS).  `candidates` is CONSUMED (left empty).
 * `excluded` (or MSI_BITS_NO_SLOT) := every document OF THE INDEX that holds a value of a kept candidate
 * (DistinctOutput::excluded: the kept candidates themselves are in it when they have a value).
 * *out_rounds (nullable): parallel rounds run; bit 31 set when the sequential kernel finished the job.
 * msi_bits_distinct_excluded: `excluded` := every document that shares a value with a document of `kept`
 * (distinct_single_docid over a set; the rule-less path keeps only the first from+length documents).
 * msi_bits_andnot_many_count: slots[i] &= ~removed with the new cardinalities (n <= 16), one launch and one wait:
 * `for universe in ranking_rule_universes { *universe -= &excluded }`. */
typedef struct msi_doc_values msi_doc_values;
#define MSI_BITS_NO_SLOT 0xFFFFFFFFu
#define MSI_DISTINCT_MAX_ROUNDS 32
int32_t msi_doc_values_create(msi_ctx *ctx, const uint64_t *offsets /* [n_docs + 1], offsets[0] = 0 */,
                              const uint32_t *value_ids /* [offsets[n_docs]] */, uint64_t n_docs, uint32_t n_values,
                              msi_doc_values **out);
void msi_doc_values_destroy(msi_doc_values *values);
int32_t msi_bits_distinct(msi_bits *pool, const msi_doc_values *values, uint32_t candidates, uint32_t remaining,
                          uint32_t excluded, uint64_t *out_remaining, uint32_t *out_rounds);
int32_t msi_bits_distinct_excluded(msi_bits *pool, const msi_doc_values *values, uint32_t kept, uint32_t excluded);
int32_t msi_bits_andnot_many_count(msi_bits *pool, uint32_t removed, uint32_t n, const uint32_t *slots,
                                   uint64_t *out_counts);
/* SURVEY §8 f3 — the GeoSort ranking rule (crates/milli/src/search/new/geo_sort.rs:14-160 over
 * documents/geo_sort.rs:66-224) without an R-tree walk or a facet-database read per candidate: the `_geo` point of
 * every document resident in HBM (16 bytes per document: lat, lng as f64; a NaN latitude = the document is not in
 * geo_faceted_documents_ids).  msi_bits_geo_next = the rule's next_bucket for the target point (lat, lng):
 * bucket := the documents of `universe` with a point whose distance (distance_between_two_points, lib.rs:388-393:
 * geoutils' haversine, metres rounded to mm) is within `distance_error_margin` of the nearest (ascending) /
 * farthest (descending) one, at most `max_bucket_size` of them — the nearest / farthest first, ascending docids among
 * equal distances; universe -= bucket; *out_first_docid = the document the bucket starts with (ScoreDetails::GeoSort's
 * `value` is ITS point; the smallest docid among equally distant ones), *out_count = |bucket|.
 * When no document of the universe has a point: *out_first_docid = 0xFFFFFFFF, *out_count = 0, nothing is changed —
 * the rule then answers with the whole universe and value None (geo_sort.rs:149-153).
 * Exact distance order is what both of the reference's strategies compute when distances differ by more than a
 * metre (its own tests assert that they agree: tests/geo_sort.rs:30-66); below that the iterative strategy's
 * truncation to whole metres makes bucket boundaries depend on docid order — not reproduced. */
typedef struct msi_geo_points msi_geo_points;
int32_t msi_geo_points_create(msi_ctx *ctx, const double *lat_lng /* [n_docs][2] */, uint64_t n_docs,
                              msi_geo_points **out);
void msi_geo_points_destroy(msi_geo_points *points);
int32_t msi_bits_geo_next(msi_bits *pool, const msi_geo_points *points, uint32_t universe, uint32_t bucket,
                          uint32_t scratch /* a third slot: used when more than max_bucket_size documents fit */,
                          double lat, double lng, int32_t ascending, uint32_t max_bucket_size,
                          double distance_error_margin, uint32_t *out_first_docid, uint64_t *out_count);
/* The documents of `universe` that have a point, with their distance to (lat, lng) — at most `cap` of them, in no
 * particular order — and how many there are in all (*out_total; the list is complete when *out_total <= cap).  What the
 * iterative strategy of documents/geo_sort.rs:120-131 sorts: the ranked search orders a small candidate set on the host
 * exactly as the reference does (distance truncated to metres, then docid). */
int32_t msi_bits_geo_list(msi_bits *pool, const msi_geo_points *points, uint32_t universe, double lat, double lng,
                          uint32_t cap, uint32_t *out_docids, double *out_distance, uint64_t *out_total);
/* SURVEY §8 f1 — the LEAVES of a filter on the device (crates/milli/src/search/facet/filter/index_filter.rs:84-340,
 * value_bounds.rs): instead of walking the facet levels of facet_id_f64_docids / facet_id_string_docids per
 * condition (facet_range_search.rs) and densifying the resulting Roaring bitmap for the scan, the facet values of a
 * filterable field live in HBM per document — CSR of u64 SORT KEYS, one table per (field, kind):
 *   numbers: msi_facet_number_key(x), a monotone map f64 -> u64 (x < y <=> key(x) < key(y); -0.0 and +0.0 share
 *            one key, as f64_into_bytes stores both as +0.0, facet/value_encoding.rs:5-7; non-finite values are never
 *            indexed by milli and must not be passed),
 *   strings: the rank of the normalised value (normalize_facet, lib.rs:442-444) among the field's distinct values in
 *            byte order — what facet_id_string_fst enumerates — so that a string range, `STARTS WITH` (the range
 *            [prefix, prefix with its last byte + 1), index_filter.rs:198-250) and `=` are rank intervals the shim
 *            finds with two dictionary lookups, and `CONTAINS` / `IN` are rank lists.
 * msi_bits_facet_range: dst := {d : some key of d lies in [lo, hi]} (accumulate != 0: dst |= ...).  Exclusive bounds
 * are the neighbouring keys (lo + 1 / hi - 1); an empty interval (lo > hi) selects nothing (index_filter.rs:315-321).
 * msi_bits_facet_in: the same for a strictly ascending list of keys.  AND / OR / NOT of the expression are
 * msi_bits_op over the leaves' slots; EXISTS / IS NULL / IS EMPTY are the index's own bitmaps (msi_bits_set_from_cbo).
 * msi_bits_geo_within: dst := the documents of `src` whose _geo point is within radius_m + f64::EPSILON metres of
 * (lat, lng) — `_geoRadius`, index_filter.rs:465-503 (distance_between_two_points over every point instead of the
 * R-tree's nearest-neighbour walk; `_geoBoundingBox` is two msi_bits_facet_range over the _geo.lat / _geo.lng
 * tables, as index_filter.rs:531-690 does with its own Between conditions). */
typedef struct msi_facet_keys msi_facet_keys;
uint64_t msi_facet_number_key(double value);
int32_t msi_facet_keys_create(msi_ctx *ctx, const uint64_t *offsets /* [n_docs + 1] */, const uint64_t *keys,
                              uint64_t n_docs, msi_facet_keys **out);
void msi_facet_keys_destroy(msi_facet_keys *keys);
int32_t msi_bits_facet_range(msi_bits *pool, const msi_facet_keys *keys, uint64_t lo, uint64_t hi, uint32_t dst,
                             int32_t accumulate);
int32_t msi_bits_facet_in(msi_bits *pool, const msi_facet_keys *keys, const uint64_t *sorted_keys, uint64_t n,
                          uint32_t dst, int32_t accumulate);
int32_t msi_bits_geo_within(msi_bits *pool, const msi_geo_points *points, uint32_t src, double lat, double lng,
                            double radius_m, uint32_t dst);
int32_t msi_bits_set_from_words(msi_bits *pool, uint32_t slot,
                                const uint64_t *words, uint64_t n_words);
int32_t msi_bits_fill(msi_bits *pool, uint32_t slot, int32_t ones);
enum { MSI_BITS_AND = 0, MSI_BITS_OR = 1, MSI_BITS_ANDNOT = 2, MSI_BITS_XOR = 3 };
/* dst := a OP b */
int32_t msi_bits_op(msi_bits *pool, uint32_t dst, uint32_t a, uint32_t b,
                    int32_t op);
/* dst = a OP b and its cardinality in one pass (the `&` + `is_empty()` of visit_path_condition,
 * graph_based_ranking_rule.rs:383-437). */
int32_t msi_bits_op_count(msi_bits *pool, uint32_t dst, uint32_t a, uint32_t b, int32_t op,
                          uint64_t *out_count);
/* dst := (OR of srcs[0..n)) AND universe   (universe == UINT32_MAX: no AND) */
int32_t msi_bits_union_many_and(msi_bits *pool, uint32_t dst,
                                const uint32_t *srcs, uint32_t n,
                                uint32_t universe);
int32_t msi_bits_count(msi_bits *pool, uint32_t slot, uint64_t *out);
/* first (ascending) `k` docids of the set; returns how many were written. */
int32_t msi_bits_first_k(msi_bits *pool, uint32_t slot, uint32_t k,
                         uint32_t *out_docids, uint32_t *out_n);
int32_t msi_bits_read_words(msi_bits *pool, uint32_t slot, uint64_t *out_words);
/* Device address of a slot (n_docs bits, 64-bit words) — e.g. as the
 * `d_filter_bits` of msi_vs_search_device. */
const uint64_t *msi_bits_device_ptr(msi_bits *pool, uint32_t slot);

/* ------------------------------------- S3: ranking-rule bucket sort (Words, Typo) */
/*
 * bucket_sort (crates/milli/src/search/new/bucket_sort.rs:23-343) over the
 * ranking rules Words (graph_based_ranking_rule.rs + ranking_rule_graph/words/
 * mod.rs:22-53) and Typo (ranking_rule_graph/typo/mod.rs:23-85) for a query graph
 * that is a chain of single-word terms.  Each term hands over, as slots of a
 * msi_bits pool, the documents that contain one of its derivations with exactly
 * 0 / 1 / 2 typos — what compute_query_term_subset_docids
 * (resolve_query_graph.rs:33-130) returns for the zero / one (incl. split words)
 * / two typo subsets — and its max_typo_cost (query_term/mod.rs:340-370).
 * Returns documents [from, from+length) of the bucket order: matched-word prefix
 * longest first (Words; strategy Last drops terms from the end, never the first
 * one — query_graph.rs:346-406; strategy All keeps only full matches), then
 * total typos ascending (Typo, when use_typo), then ascending docid
 * (bucket_sort.rs:382-460), with the ScoreDetails of both rules:
 *   Words{matching_words, max_matching_words = n_terms}
 *   Typo{typo_count, max_typo_count = sum of max_typo_cost over the kept terms}
 */
#define MSI_RANK_MAX_TERMS 10          /* words_limit, crates/milli/src/search/mod.rs:111 */
#define MSI_NO_SLOT 0xFFFFFFFFu
/* TermsMatchingStrategy (crates/milli/src/search/mod.rs:538-556).  MSI_TERMS_FREQUENCY: the terms are dropped in order of
 * decreasing document frequency (query_graph.rs:303-344) — msi_keyword_search_ranked only; the [Words, Typo] fast path
 * (msi_rank_query_graph / msi_keyword_search) answers MSI_E_UNSUPPORTED for it. */
enum { MSI_TERMS_LAST = 0, MSI_TERMS_ALL = 1, MSI_TERMS_FREQUENCY = 2 };
typedef struct msi_rank_term {
  uint32_t level_slot[3];  /* pool slot of the 0 / 1 / 2 typo documents, or MSI_NO_SLOT */
  uint32_t max_typo_cost;  /* 0..2 */
} msi_rank_term;
/* The same over the full query graph: nodes are the single terms plus the 2-gram and
 * 3-gram nodes of adjacent terms (query_graph.rs:96-180, make_ngram
 * parse_query.rs:227-300); an n-gram node has the base typo cost n
 * (typo/mod.rs:41-45) and matching_words counts the terms it covers.  At most one
 * node per (first_term, last_term), last_term - first_term <= 2.
 * max_typo_count is the largest cost of a path that matches a document of the Words
 * bucket (exact for chains; with n-grams the reference derives it from the paths its
 * DFS found non-empty, which can be smaller when one path shadows another). */
typedef struct msi_rank_node {
  uint32_t first_term, last_term; /* 0-based, inclusive */
  uint32_t level_slot[3];
  uint32_t max_typo_cost;
} msi_rank_node;
int32_t msi_rank_query_graph(msi_bits *pool, const msi_rank_node *nodes,
                             uint32_t n_nodes, uint32_t n_terms, uint32_t universe_slot,
                             uint32_t scratch_slot, int32_t strategy, int32_t use_typo,
                             uint32_t from, uint32_t length, uint32_t *out_docids,
                             uint32_t *out_matching_words, uint32_t *out_typo_count,
                             uint32_t *out_max_typo_count, uint32_t *out_n,
                             uint64_t *out_candidates);
/* The two steps of msi_rank_query_graph on their own, for rule lists that continue after
 * Typo (Proximity, Attribute, … stay on the reference's CPU path): the buckets of
 * [Words, Typo] in bucket-sort order with their sizes and ScoreDetails, and one bucket
 * as a docid set in `dst_slot` (read it back with msi_bits_read_words, or keep it in HBM
 * as the universe of the next rule). */
typedef struct msi_rank_bucket {
  uint32_t matching_words, typo_count, max_typo_count, _pad;
  uint64_t count;
} msi_rank_bucket;
int32_t msi_rank_buckets(msi_bits *pool, const msi_rank_node *nodes, uint32_t n_nodes,
                         uint32_t n_terms, uint32_t universe_slot, uint32_t scratch_slot,
                         int32_t strategy, int32_t use_typo, msi_rank_bucket *out_buckets,
                         uint32_t cap, uint32_t *out_n);
int32_t msi_rank_materialise(msi_bits *pool, const msi_rank_node *nodes, uint32_t n_nodes,
                             uint32_t n_terms, uint32_t universe_slot, int32_t strategy,
                             int32_t use_typo, uint32_t matching_words, uint32_t typo_count,
                             uint32_t dst_slot);
/* Batched form for serving throughput: many queries per launch (histograms,
 * materialisation and ordered extraction run with one grid row per query), so a batch
 * costs a handful of launches and synchronisations.  Outputs are [n_queries][length]
 * (rows padded; out_n[q] valid entries).  Each query owns 4 consecutive scratch slots. */
typedef struct msi_rank_query {
  const msi_rank_node *nodes;
  uint32_t n_nodes, n_terms, universe_slot, scratch_slot;
} msi_rank_query;
int32_t msi_rank_query_graph_batch(msi_bits *pool, const msi_rank_query *queries,
                                   uint32_t n_queries, int32_t strategy, int32_t use_typo,
                                   uint32_t from, uint32_t length, uint32_t *out_docids,
                                   uint32_t *out_matching_words, uint32_t *out_typo_count,
                                   uint32_t *out_max_typo_count, uint32_t *out_n,
                                   uint64_t *out_candidates);
int32_t msi_rank_words_typo(msi_bits *pool, const msi_rank_term *terms,
                            uint32_t n_terms, uint32_t universe_slot,
                            uint32_t scratch_slot, int32_t strategy,
                            int32_t use_typo, uint32_t from, uint32_t length,
                            uint32_t *out_docids, uint32_t *out_matching_words,
                            uint32_t *out_typo_count, uint32_t *out_max_typo_count,
                            uint32_t *out_n, uint64_t *out_candidates);

/* ------------------------------------------ keyword leg of Search::execute (host) */
/*
 * The keyword search for the ranking rules [Words, Typo] end to end, on top of an index
 * the caller owns: tokens (already normalised single words, <= 10; words_limit) ->
 * query graph with 2-/3-gram nodes (parse_query.rs:28-300, query_graph.rs:96-180) ->
 * typo budgets (parse_query.rs:204-225) -> derivations: ONE batched msi_dict lookup,
 * zero-typo prefix derivations from the dictionary (compute_derivations.rs:40-73),
 * split words (:363-383) -> posting sets per node and typo level
 * (resolve_query_graph.rs:33-59; postings arrive as the CboRoaringBitmap bytes they
 * are stored as and are decoded on the device) -> msi_rank_query_graph.
 * Phrases, synonyms, the word-prefix databases and the other ranking rules stay on
 * the reference path.  The pool needs 2 + 3·(number of graph nodes) <= 83 slots; slots
 * 0 and 1 are the universe and scratch.
 */
/* Receives one stored CboRoaringBitmap value; returns 0, or negative to make the callback stop. */
typedef int32_t (*msi_posting_sink)(void *sink, const uint8_t *bytes, size_t n);
struct msi_query_token;
typedef int32_t (*msi_synonym_sink)(void *sink, const struct msi_query_token *words, uint32_t n_words);
typedef struct msi_index_vtable {
  void *user;
  /* Posting list of `word` as CboRoaringBitmap bytes (valid until the next callback):
   * original != 0: word_docids ∪ exact_word_docids (Word::Original, db_cache.rs), else
   * word_docids only (Word::Derived).  *n = 0 when absent.  Negative return = error. */
  int32_t (*word_docids)(void *user, const uint8_t *word, uint32_t len, int32_t original,
                         const uint8_t **bytes, size_t *n);
  /* word_pair_proximity_docids(proximity, left, right); nullable = no split words. */
  int32_t (*word_pair_proximity_docids)(void *user, uint32_t proximity, const uint8_t *left,
                                        uint32_t left_len, const uint8_t *right,
                                        uint32_t right_len, const uint8_t **bytes, size_t *n);
  /* exact_words FST membership; nullable. */
  int32_t (*is_exact_word)(void *user, const uint8_t *word, uint32_t len);
  /* The reads of the attribute / position / exactness rules (msi_keyword_search_ranked only;
   * nullable when those rules are not in the criteria).  db_cache.rs:533-552,629-650:
   * word_fid_docids(word, fid), word_position_docids(word, bucketed position) as stored bytes;
   * the fids / bucketed positions a word has entries for (db_cache.rs:575-599, prefix_iter over the
   * same databases; *n receives the count, at most `cap` written);
   * field_id_word_count_docids(fid, count) (exact_attribute.rs:187-200). */
  int32_t (*word_fid_docids)(void *user, const uint8_t *word, uint32_t len, uint32_t fid,
                             const uint8_t **bytes, size_t *n);
  int32_t (*word_position_docids)(void *user, const uint8_t *word, uint32_t len, uint32_t position,
                                  const uint8_t **bytes, size_t *n);
  int32_t (*word_fids)(void *user, const uint8_t *word, uint32_t len, uint16_t *out, uint32_t cap,
                       uint32_t *n);
  int32_t (*word_positions)(void *user, const uint8_t *word, uint32_t len, uint16_t *out, uint32_t cap,
                            uint32_t *n);
  int32_t (*field_id_word_count_docids)(void *user, uint32_t fid, uint32_t count,
                                        const uint8_t **bytes, size_t *n);
  /* The word-prefix databases (a prefix term whose word is a key of word_prefix_docids uses them instead of
   * enumerating its derivations: compute_derivations.rs:193-205, query_term/mod.rs:183-203).  All nullable
   * (= the index has no prefix databases).  A lookup may have to hand over several stored values (tolerant +
   * exact database, or every key of a prefix_iter), so these push each value into the sink the engine
   * passes; the return value is the number of values pushed (0 = the key does not exist), negative = error:
   *   word_prefix_docids(prefix, original)       word_prefix_docids (+ exact_word_prefix_docids if original)
   *   word_prefix_fid_docids(prefix, fid), word_prefix_position_docids(prefix, position)
   *   word_prefix_pair_proximity_docids(prox, word1, prefix2): every word_pair_proximity_docids value whose key
   *       starts with (prox, word1, prefix2…)   (db_cache.rs:451-520)
   *   word_prefix_fids / word_prefix_positions: as word_fids / word_positions. */
  int32_t (*word_prefix_docids)(void *user, const uint8_t *prefix, uint32_t len, int32_t original,
                                msi_posting_sink push, void *sink);
  int32_t (*word_prefix_fid_docids)(void *user, const uint8_t *prefix, uint32_t len, uint32_t fid,
                                    msi_posting_sink push, void *sink);
  int32_t (*word_prefix_position_docids)(void *user, const uint8_t *prefix, uint32_t len, uint32_t position,
                                         msi_posting_sink push, void *sink);
  int32_t (*word_prefix_pair_proximity_docids)(void *user, uint32_t proximity, const uint8_t *word1,
                                               uint32_t len1, const uint8_t *prefix2, uint32_t len2,
                                               msi_posting_sink push, void *sink);
  int32_t (*word_prefix_fids)(void *user, const uint8_t *prefix, uint32_t len, uint16_t *out, uint32_t cap,
                              uint32_t *n);
  int32_t (*word_prefix_positions)(void *user, const uint8_t *prefix, uint32_t len, uint16_t *out,
                                   uint32_t cap, uint32_t *n);
  /* index.synonyms.get(words) (compute_derivations.rs:217-236 for one word, parse_query.rs:277-285 for the
   * words of an n-gram): pushes every synonym as its tokenised words, in stored order.  Nullable. */
  int32_t (*synonyms)(void *user, const struct msi_query_token *words, uint32_t n_words,
                      msi_synonym_sink push, void *sink);
  /* The keys of exact_word_docids that start with `prefix`, in key order (the words of exact attributes are not
   * in the words FST / msi_dict): find_zero_typo_prefix_derivations merges them with the word_docids keys
   * (compute_derivations.rs:40-73).  Pushes each word as a one-token list.  Nullable = no exact attributes. */
  int32_t (*exact_words_with_prefix)(void *user, const uint8_t *prefix, uint32_t len, msi_synonym_sink push,
                                     void *sink);
} msi_index_vtable;
typedef struct msi_query_token {
  const uint8_t *word;
  uint32_t len;
  uint32_t is_prefix; /* the last token of a query that does not end with a separator */
} msi_query_token;
typedef struct msi_keyword_params {
  uint32_t authorize_typos;         /* index.authorize_typos */
  uint32_t min_word_len_one_typo;   /* 5 */
  uint32_t min_word_len_two_typos;  /* 9 */
  int32_t strategy;                 /* MSI_TERMS_LAST | MSI_TERMS_ALL */
  int32_t use_typo;                 /* Typo rule present after Words */
  uint32_t from, length;
} msi_keyword_params;
int32_t msi_keyword_search(msi_dict *dict, msi_bits *pool, const msi_index_vtable *index,
                           const msi_query_token *tokens, uint32_t n_tokens,
                           const msi_keyword_params *params, const uint8_t *universe_cbo,
                           size_t universe_len, uint32_t *out_docids,
                           uint32_t *out_matching_words, uint32_t *out_typo_count,
                           uint32_t *out_max_typo_count, uint32_t *out_n,
                           uint64_t *out_candidates);

/* ------------------------- keyword leg with every graph-based ranking rule (host + device sets) */
/*
 * bucket_sort (bucket_sort.rs:23-343) over the rule list get_ranking_rules_for_query_graph_search
 * builds from index.criteria (search/new/mod.rs:510-649): Words, Typo, Proximity, Attribute (Fid +
 * Position), AttributeRank, WordPosition, Exactness (ExactAttribute + Exactness); Sort / Asc / Desc are
 * skipped (not keyword rules).  Every rule is the generic GraphBasedRankingRule
 * (graph_based_ranking_rule.rs:136-368) over ranking_rule_graph/{words,typo,proximity,fid,position,
 * exactness}; between rules the query graph is rebuilt from the paths that matched
 * (QueryGraph::build_from_paths, query_graph.rs:470-544).  Terms: single words (typo derivations from the
 * device dictionary, prefix derivations, split words, 2-/3-grams) and quoted phrases
 * (compute_phrase_docids, resolve_query_graph.rs:187-268).  The control flow (small graphs) runs on the
 * caller's thread; every docid set lives in the msi_bits pool and every set operation — posting decode,
 * union, intersection, difference, cardinality, ordered extraction — is a device kernel.
 * Pins: msi_inject_pins around this call (the reference injects them after the bucket sort too).
 * The tokenizer stays with the caller: it hands over the located terms of
 * located_query_terms_from_tokens (parse_query.rs:28-202); stop words are its business (dropped, or empty
 * tokens inside a phrase); n_terms = 0 (only stop words) is a placeholder search: the universe in docid order.
 */
enum {
  MSI_CRIT_WORDS = 0, MSI_CRIT_TYPO = 1, MSI_CRIT_PROXIMITY = 2, MSI_CRIT_ATTRIBUTE = 3,
  MSI_CRIT_ATTRIBUTE_RANK = 4, MSI_CRIT_WORD_POSITION = 5, MSI_CRIT_EXACTNESS = 6,
  MSI_CRIT_SORT = 7,     /* ignored: the shim expands Criterion::Sort / Asc / Desc into MSI_CRIT_ORDER_BY entries */
  MSI_CRIT_ORDER_BY = 8, /* one Sort rule (mod.rs:366-376,640-720: one per sorted field, a field only once); the
                          * i-th MSI_CRIT_ORDER_BY of the list uses params->order_keys[i] */
  MSI_CRIT_GEO_SORT = 9  /* one GeoSort rule (mod.rs:690-712: AscDesc::{Asc,Desc}(Member::Geo(point))); the i-th
                          * MSI_CRIT_GEO_SORT of the list uses params->geo_rules[i] */
};
enum { /* ScoreDetails variants, score_details.rs:10-27 */
  MSI_SCORE_WORDS = 0,           /* a = matching_words, b = max_matching_words */
  MSI_SCORE_TYPO = 1,            /* a = typo_count, b = max_typo_count */
  MSI_SCORE_PROXIMITY = 2,       /* a = rank, b = max_rank */
  MSI_SCORE_FID = 3,             /* a = rank, b = max_rank */
  MSI_SCORE_POSITION = 4,        /* a = rank, b = max_rank */
  MSI_SCORE_EXACT_ATTRIBUTE = 5, /* a = 3 ExactMatch | 2 MatchesStart | 1 NoExactMatch, b = 3 */
  MSI_SCORE_EXACT_WORDS = 6,     /* a = matching_words, b = max_matching_words */
  MSI_SCORE_SKIPPED = 7,         /* the deadline cut the ranking short here; rank 0 of 1 */
  MSI_SCORE_SORT = 8,            /* a = index into order_keys, b = the bucket's order key (0xFFFFFFFF: Null); no rank
                                  * (score_details.rs:103-121: Sort does not enter the global score) */
  MSI_SCORE_GEO_SORT = 9,        /* a = index into geo_rules, b = the docid whose point is the bucket's `value`
                                  * (0xFFFFFFFF: None — no _geo); no rank either */
  MSI_SCORE_PIN = 10             /* a = position: the only detail of a pinned hit (msi_inject_pins); no rank, not part of
                                  * the global score (score_details.rs:123,135) */
};
#define MSI_MAX_SCORE_DETAILS 16 /* the 7 keyword rules + the Sort / GeoSort rules of a request; a longer rule list is cut here */
typedef struct msi_score_detail {
  uint32_t kind, a, b;
} msi_score_detail;
typedef struct msi_located_term {
  const msi_query_token *words; /* one word, or the words of a quoted phrase (len 0 = a stop word) */
  uint32_t n_words;
  uint32_t is_phrase;   /* bit 0: quoted phrase; bit 1 (MSI_TERM_NEGATIVE): `-word` / `-"phrase"` — its documents are
                         * removed from the universe (search/mod.rs:431-440) and it is no term of the query graph */
  uint32_t position_start, position_end; /* parse_query.rs:60-120: +1 per word, +7 more over a hard separator */
} msi_located_term;
#define MSI_TERM_PHRASE 1u
#define MSI_TERM_NEGATIVE 2u
typedef struct msi_geo_rule {
  const msi_geo_points *points;
  double lat, lng;   /* the target point of the request's _geoPoint(lat, lng) */
  int32_t ascending;
} msi_geo_rule;
typedef struct msi_search_params {
  uint32_t authorize_typos, min_word_len_one_typo, min_word_len_two_typos;
  int32_t strategy;                    /* MSI_TERMS_LAST | MSI_TERMS_ALL | MSI_TERMS_FREQUENCY */
  const int32_t *criteria;             /* index.criteria as MSI_CRIT_* */
  uint32_t n_criteria;
  const uint16_t *searchable_fids;     /* searchable_fields_ids, with their weights (fieldids_weights_map) */
  const uint16_t *searchable_weights;
  uint32_t n_searchable;
  int32_t max_weight;                  /* max_searchable_attribute_weight, -1 = None (index.rs:689-698) */
  uint32_t from, length;
  int32_t detailed_scores;             /* ScoringStrategy::Detailed (else Skip) */
  /* The search cutoff (Deadline, lib.rs:150-231; bucket_sort.rs:206-264): once the budget is spent, what is
   * left of every rule's universe is returned unranked with a Skipped score detail and *out_degraded = 1.
   * time_budget_us: 0 = none.  stop_after: >= 0 = "exceeded from the (n+1)-th check on" (the reference's test
   * hook Deadline::with_stop_after, which then ignores the time budget), -1 = off. */
  uint64_t time_budget_us;
  int32_t stop_after;
  int32_t has_score_threshold;
  /* ranking_score_threshold (bucket_sort.rs:286-306): a bucket whose ScoreDetails::global_score so far is below
   * it is dropped together with what is left of that rule's universe; *out_candidates excludes both. */
  double score_threshold;
  /* Sort / Asc / Desc rules: the key array of the i-th MSI_CRIT_ORDER_BY criterion (NULL / 0 when there is none).
   * They also order a placeholder search (no term survived: mod.rs:352-420). */
  const msi_doc_keys *const *order_keys;
  uint32_t n_order_keys;
  /* The distinct field of the request or of the index (distinct_fid, distinct.rs:130-147): NULL = none.  Every
   * bucket that reaches the results goes through msi_bits_distinct; what it excludes leaves every rule's universe
   * and *out_candidates (bucket_sort.rs:399-415).  Also on placeholder / rule-less searches (:61-92). */
  const msi_doc_values *distinct_values;
  /* GeoSort rules (one per MSI_CRIT_GEO_SORT criterion, in order) and the request's GeoSortParameter
   * (documents/geo_sort.rs:12-30: max_bucket_size — 0 = its default, 1000 — and distance_error_margin, metres, 1.0 by
   * default in the reference: passed as it is). */
  const msi_geo_rule *geo_rules;
  uint32_t n_geo_rules;
  uint32_t geo_max_bucket_size;
  double geo_distance_error_margin;
  /* exhaustive_number_hits / max_total_hits of bucket_sort (bucket_sort.rs:187-191): with a score threshold AND an
   * exhaustive count the rules keep running past the page — up to max_total_hits hits (0 = None) — so that every bucket
   * below the threshold leaves *out_candidates; with a distinct field the count is what the distinct rule keeps of
   * all_candidates (search/new/mod.rs:894-907). */
  int32_t exhaustive_number_hits;
  uint32_t max_total_hits;
  /* GeoSortStrategy of the request (documents/geo_sort.rs:32-63): MSI_GEO_DYNAMIC (the reference's default,
   * Dynamic(1000)) walks the R-tree when a fill finds at least geo_cache_size candidates and sorts them ITERATIVELY below
   * that — by distance truncated to whole metres, docid order inside a metre (:127-131) — which decides the order of
   * documents closer than a metre to each other; geo_cache_size 0 = 1000. */
  int32_t geo_strategy;
  uint32_t geo_cache_size;
  /* Which VIEW of the index the callbacks answer for; 0 = the index as it is.  A request with attributesToSearchOn reads
   * the same databases through other functions (search/new/mod.rs:140-222, db_cache.rs:208-345,540-575: word_docids becomes
   * the union of word_fid_docids over the restricted tolerant fields, exact_word_docids the union over the restricted exact
   * fields, the prefix databases likewise, word_fid_docids of a field outside the restriction is absent): the shim's
   * callbacks answer that way and name the view here — any value that identifies the restriction, e.g. a hash of the
   * field list.  The engine keys everything it remembers about stored values (the HBM posting cache of the index version,
   * what it knows about absent keys) by (view, key), so searches under different restrictions never read each other's
   * postings. */
  uint64_t index_view;
} msi_search_params;
enum { MSI_GEO_DYNAMIC = 0, MSI_GEO_ALWAYS_ITERATIVE = 1, MSI_GEO_ALWAYS_RTREE = 2 };
/* out_scores: [length][MSI_MAX_SCORE_DETAILS], out_n_scores: [length].  The pool needs at least 64 free
 * slots above slot 0 (more for long queries: one per live condition of every active rule). */
int32_t msi_keyword_search_ranked(msi_dict *dict, msi_bits *pool, const msi_index_vtable *index,
                                  const msi_located_term *terms, uint32_t n_terms,
                                  const msi_search_params *params, const uint8_t *universe_cbo,
                                  size_t universe_len, uint32_t *out_docids, msi_score_detail *out_scores,
                                  uint32_t *out_n_scores, uint32_t *out_n, uint64_t *out_candidates,
                                  int32_t *out_degraded /* nullable */);

/* Pins (dynamic search rules).  The reference resolves a request's pins before the ranking (resolve_pins,
 * dynamic_search_rules.rs:73-96: every surviving pin's document LEAVES the universe), runs the bucket sort for the organic
 * prefix [0, from + length) when there are pins (bucket_sort.rs:45-50) and merges the pins into it (inject_pins,
 * bucket_sort.rs:345-377, over merge_positioned_hits_into_page, search/mod.rs:579-625).  The shim does the same around
 * msi_keyword_search_ranked: pinned documents out of universe_cbo, params.from = 0, params.length = from + length, then
 * this call with the organic hits it got.  pins: in resolve_pins' order (the merge consumes them front to back; a pin whose
 * position lies beyond the organic hits is pumped forward).  scores / n_scores / out_scores / out_n_scores nullable
 * ([..][MSI_MAX_SCORE_DETAILS] rows as msi_keyword_search_ranked writes them); a pinned hit carries the single detail
 * {MSI_SCORE_PIN, position}.  out_*: caller-allocated, `length` entries.  Returns the number of hits of the page.
 * (all_candidates gains the pins: the caller adds n_pins to its count.)  The hybrid path merges its pins with the same
 * function after ScoreWithRatioResult::merge (hybrid.rs:239-260): call this on msi_hybrid_merge's list the same way. */
typedef struct msi_pin {
  uint32_t position;
  uint32_t docid;
} msi_pin;
uint32_t msi_inject_pins(const msi_pin *pins, uint32_t n_pins, uint32_t from, uint32_t length,
                         const uint32_t *docids, const msi_score_detail *scores, const uint32_t *n_scores,
                         uint32_t n, uint32_t *out_docids, msi_score_detail *out_scores, uint32_t *out_n_scores);

/* ScoreDetails::global_score over the details of one hit (score_details.rs:123-154, ranks per variant
 * :103-121: Typo -> (max_typo_count + 1 - typo_count, max_typo_count + 1), ExactWords -> (matching + 1, max + 1),
 * ExactAttribute -> (a, 3)): the keyword score value msi_hybrid_merge weighs against the semantic one. */
double msi_score_details_global_score(const msi_score_detail *details, uint32_t n);

/* Diagnostics of the command-list combiner of the pool's context (msi_keyword_search_ranked records its set operations
 * and submits them as lists; the lists of all searches waiting at that moment share one kernel launch): [rounds launched,
 * lists executed, then nanoseconds summed over lists: queued before the combiner took the list, packing, (per round)
 * inside the HIP launch calls, from the launch calls until the caller saw its result]. */
int32_t msi_bits_vm_stats(msi_bits *pool, uint64_t out[6]);
/* Diagnostics, process-wide: what the command lists executed so far ASKED the memory system for — [bytes of set operands
 * (every operand of every command, whole), bytes of posting containers their decodes read, lists] — the algorithmic
 * bytes of the keyword leg's roofline object (bench.py: keyword_roofline). */
int32_t msi_bits_vm_bytes(uint64_t out[3]);

/* Diagnostics: counters of the last msi_keyword_search_ranked on the calling thread —
 * [kernel launches, stream syncs, decode batches, index callbacks, posting bytes decoded, matching paths,
 *  buckets, callback microseconds, device-wait microseconds, total microseconds]. */
int32_t msi_search_last_stats(uint64_t out[10]);
/* Diagnostics, process-wide, collected while MSI_SEARCH_CPU_PROFILE is set in the environment: HOST CPU (thread CPU time,
 * nanoseconds; a sleeping waiter costs none) of the keyword leg — [searches, whole searches, inside the command-list
 * submission + wait, of it finalising the list, typo derivations, index callbacks (wall), the combiner threads, lists]. */
int32_t msi_search_cpu_profile(uint64_t out[8]);
/* Switches that collection on / off while the process runs (the environment variable is read once, at the first search). */
int32_t msi_search_cpu_profile_enable(int32_t on);
/* Diagnostics, process-wide: [ranked keyword searches, of them continued in the COMPACT SPACE (once a search knows its
 * universe — the documents that match the query at all — every later set is kept over the ranks of the documents inside
 * it, |universe| bits instead of n_docs: DESIGN.md §4.7.2), documents of those universes summed]. */
int32_t msi_search_compaction_stats(uint64_t out[3]);
/* [sub-trees of the bucket sort that continued in the compact space of their own bucket, documents of those buckets summed]
 * — counted apart from msi_search_compaction_stats, whose [1] and [2] are about whole universes only. */
int32_t msi_search_late_compaction_stats(uint64_t out[2]);

/* ---------------------------------------------------- scoring arithmetic (host) */
/* DistributionShift::shift (crates/milli/src/vector/distribution.rs:103-130). */
float msi_distribution_shift(float mean, float sigma, float score);
/* Rank::merge folded over (rank,max_rank) pairs then local_score
 * (crates/milli/src/score_details.rs:512-547). */
double msi_rank_global_score(const uint32_t *ranks, const uint32_t *max_ranks,
                             uint32_t n);
/* compare_scores restricted to ScoreValue::Score lists
 * (crates/milli/src/search/hybrid.rs:32-80): returns -1/0/+1. */
int32_t msi_compare_scores(const double *left, uint32_t n_left, float left_ratio,
                           const double *right, uint32_t n_right,
                           float right_ratio);

/* ------------------------------------------- semantic / hybrid result tail (host) */
/* VectorSort as the only ranking rule (search/new/vector_sort.rs:58-168) over the
 * output of msi_vs_search / msi_merge_topk: first occurrence of a docid wins,
 * similarity = 1 - distance, optional DistributionShift; returns [from, from+length). */
uint32_t msi_vector_sort(const uint32_t *docids, const float *dist, uint32_t n,
                         int32_t has_shift, float mean, float sigma, uint32_t from,
                         uint32_t length, uint32_t *out_docids, float *out_similarity);
/* ScoreWithRatioResult::merge (search/hybrid.rs:102-235; no pins, no distinct).
 * Each hit has a list of score values (ScoreDetails::score_values: one value per run
 * of rank-based rules, score_details.rs:156-175), hit i of the vector list owns
 * v_scores[v_off[i] .. v_off[i+1]).  v_ratio = semantic_ratio, k_ratio = 1 - it. */
uint32_t msi_hybrid_merge(const uint32_t *v_docids, const double *v_scores,
                          const uint32_t *v_off, uint32_t n_v, float v_ratio,
                          const uint32_t *k_docids, const double *k_scores,
                          const uint32_t *k_off, uint32_t n_k, float k_ratio,
                          uint32_t from, uint32_t length, uint32_t *out_docids,
                          uint8_t *out_is_semantic, uint32_t *out_semantic_hit_count);
/* The same for a batch, straight from the outputs of msi_vs_search (v_*: [n_queries][v_stride])
 * and msi_rank_query_graph_batch (k_*: [n_queries][k_stride]); scores are computed here:
 * similarity = 1 - distance, keyword = Rank::global_score of [Words, Typo]. */
int32_t msi_hybrid_merge_batch(const uint32_t *v_docids, const float *v_dist,
                               const uint32_t *v_counts, uint32_t v_stride,
                               const uint32_t *k_docids, const uint32_t *k_matching_words,
                               const uint32_t *k_typo_count, const uint32_t *k_max_typo_count,
                               const uint32_t *k_counts, uint32_t k_stride,
                               const uint32_t *n_terms, uint32_t n_queries, float semantic_ratio,
                               uint32_t from, uint32_t length, uint32_t *out_docids,
                               uint8_t *out_is_semantic, uint32_t *out_counts,
                               uint32_t *out_semantic_hit_counts);
/* Federated search merge (crates/meilisearch/src/search/federated/weighted_scores.rs:1-46, called from
 * federated/perform.rs when the hits of several queries / indexes are interleaved): orders two hits by their
 * WeightedScoreValue sequences (ScoreDetails::weighted_score_values, score_details.rs:177-199: the rank-based rules of a
 * hit merged into local scores and multiplied by the query's federation weight; a semantic hit contributes
 * VectorSort(similarity * weight)), falling back to the weighted GLOBAL scores when the sequences are not comparable.
 * A value is {kind, asc, value}: kind 0 = WeightedScore / VectorSort (compared with the f64::EPSILON window of
 * score_details.rs:61-67), 1 = Sort on a NUMBER (value = the number; asc = the rule's direction), 2 = GeoSort
 * (value = distance, NaN = None).  String sort values do not cross this boundary (the shim compares them).
 * Returns -1 / 0 / +1 for left <, ==, > right in the order "greater = ranks first" of weighted_scores::compare. */
typedef struct msi_weighted_value {
  uint32_t kind, asc;
  double value;
} msi_weighted_value;
int32_t msi_federated_compare(const msi_weighted_value *left, uint32_t n_left, double left_weighted_global_score,
                              const msi_weighted_value *right, uint32_t n_right, double right_weighted_global_score);
/* Merge of `n_lists` result lists, each already in its own ranking order (perform.rs: merge_index_global_results'
 * k-way merge by weighted_scores::compare): hit j of list l has values[val_off[l][j] .. val_off[l][j+1]) and a weighted
 * global score; writes the first `limit` hits after `offset` as (list, position) pairs.  Device top-k lists (msi_vs /
 * msi_bq output, similarity * weight as a single VectorSort value) and keyword hits (msi_keyword_search_ranked score
 * details -> one WeightedScore per run of rank rules) feed it without another sort.  Returns the number written. */
uint32_t msi_federated_merge(uint32_t n_lists, const uint32_t *list_len, const msi_weighted_value *const *values,
                             const uint32_t *const *val_off, const double *const *weighted_global, uint32_t offset,
                             uint32_t limit, uint32_t *out_list, uint32_t *out_pos);

/* ... with the query each hit came from (query_index[l][j]; a NULL table or row = the list's own number): hits that
 * compare Equal leave in the order of their queries — `Ordering::Equal => left.query_index < right.query_index`,
 * perform.rs:566 (merge_index_local_results) and :609 (merge_index_global_results) — which a list-per-index merge cannot
 * tell from the list order alone when the tied hits of two indexes come from interleaved queries. */
uint32_t msi_federated_merge_q(uint32_t n_lists, const uint32_t *list_len, const msi_weighted_value *const *values,
                               const uint32_t *const *val_off, const double *const *weighted_global,
                               const uint32_t *const *query_index, uint32_t offset, uint32_t limit, uint32_t *out_list,
                               uint32_t *out_pos);

/* Search::results_good_enough (search/hybrid.rs:367-386). */
int32_t msi_results_good_enough(const double *keyword_global_scores, uint32_t n,
                                uint32_t limit_plus_offset, float semantic_ratio);

/* ---- `_vectors` filter leaf (search/facet/filter/vector.rs:49-158) ----------------------------------------
 * `_vectors EXISTS`-style conditions select documents by which vector stores hold an item for them.  The stores already
 * live in HBM, so the leaf is built there: msi_vs_items_bits / msi_bq_items_bits OR a store's docids into a slot (what
 * VectorStore::items_in_store / aggregate_stats().documents return, store.rs), and msi_bits_vector_filter is
 * evaluate_inner (vector.rs:78-158) for ONE embedder:
 *   NONE               the documents of the embedder's stores                                   (:146-150)
 *   FRAGMENT           the items of the fragment's store(s) minus user_provided                  (:101-125)
 *   DOCUMENT_TEMPLATE  nothing when the embedder has fragments, else documents - user_provided   (:126-135)
 *   USER_PROVIDED      the index's user_provided bitmap of the embedder                          (:136-139)
 *   REGENERATE         documents - skip_regenerate                                               (:140-145)
 * `user_provided` / `skip_regenerate` are slots holding EmbeddingStatus's bitmaps (msi_bits_set_from_cbo); `scratch` is
 * a slot the call may overwrite.  evaluate() (vector.rs:49-76) is the shim's loop: accumulate != 0 over the named
 * embedder (or all of them), then one msi_bits_op AND with the universe.  Unknown embedder / fragment names are the
 * shim's errors (it owns the embedding configs). */
enum { MSI_VECTOR_FILTER_NONE = 0, MSI_VECTOR_FILTER_FRAGMENT = 1, MSI_VECTOR_FILTER_DOCUMENT_TEMPLATE = 2,
       MSI_VECTOR_FILTER_USER_PROVIDED = 3, MSI_VECTOR_FILTER_REGENERATE = 4 };
int32_t msi_vs_items_bits(msi_vs *store, msi_bits *pool, uint32_t slot);
int32_t msi_bq_items_bits(msi_bq *store, msi_bits *pool, uint32_t slot);
int32_t msi_bits_vector_filter(msi_bits *pool, uint32_t dst, int32_t kind, int32_t embedder_has_fragments,
                               msi_vs *const *stores, uint32_t n_stores, msi_bq *const *bq_stores, uint32_t n_bq_stores,
                               uint32_t user_provided, uint32_t skip_regenerate, uint32_t scratch, int32_t accumulate);

#ifdef __cplusplus
}
#endif
#endif /* MSI_H */
