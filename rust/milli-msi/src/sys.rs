//! Raw bindings: one item per declaration of include/msi.h (ABI version 1).
#![allow(non_camel_case_types)]
use std::os::raw::{c_char, c_void};

#[repr(C)] pub struct msi_ctx { _p: [u8; 0] }
#[repr(C)] pub struct msi_vs { _p: [u8; 0] }
#[repr(C)] pub struct msi_dict { _p: [u8; 0] }
#[repr(C)] pub struct msi_bits { _p: [u8; 0] }
#[repr(C)] pub struct msi_bq       { _p: [u8; 0] }
#[repr(C)] pub struct msi_group    { _p: [u8; 0] }
#[repr(C)] pub struct msi_vs_group { _p: [u8; 0] }
/// One value of a federated hit's ordering key (search/federated/weighted_scores.rs: WeightedScoreValue)
#[repr(C)] #[derive(Clone, Copy)]
pub struct msi_weighted_value { pub kind: u32, pub asc: u32, pub value: f64 }
pub const MSI_GROUP_REPLICATE: i32 = 0;
pub const MSI_GROUP_SHARD_ROWS: i32 = 1;
pub const MSI_VECTOR_FILTER_NONE: i32 = 0;
pub const MSI_VECTOR_FILTER_FRAGMENT: i32 = 1;
pub const MSI_VECTOR_FILTER_DOCUMENT_TEMPLATE: i32 = 2;
pub const MSI_VECTOR_FILTER_USER_PROVIDED: i32 = 3;
pub const MSI_VECTOR_FILTER_REGENERATE: i32 = 4;
#[repr(C)] pub struct msi_doc_keys { _p: [u8; 0] }
#[repr(C)] pub struct msi_doc_values { _p: [u8; 0] }
#[repr(C)] pub struct msi_geo_points { _p: [u8; 0] }
#[repr(C)] pub struct msi_facet_keys { _p: [u8; 0] }
#[repr(C)] #[derive(Clone, Copy)]
pub struct msi_geo_rule { pub points: *const msi_geo_points, pub lat: f64, pub lng: f64, pub ascending: i32 }
pub const MSI_CRIT_GEO_SORT: i32 = 9;
pub const MSI_SCORE_GEO_SORT: u32 = 9;
pub const MSI_BITS_NO_SLOT: u32 = 0xFFFF_FFFF;

pub const MSI_ABI_VERSION: i32 = 3; // include/msi.h: MSI_ABI_VERSION
pub const MSI_OK: i32 = 0;
pub const MSI_E_INVALID: i32 = -1;
pub const MSI_E_NO_DEVICE: i32 = -2;
pub const MSI_E_HIP: i32 = -3;
pub const MSI_E_OOM: i32 = -4;
pub const MSI_E_UNSUPPORTED: i32 = -5;
pub const MSI_E_CANCELLED: i32 = -6;
pub const MSI_E_NOT_SORTED: i32 = -7;
pub const MSI_E_INTERNAL: i32 = -8;

pub const MSI_VS_F32: i32 = 0;
pub const MSI_VS_BF16: i32 = 1;
pub const MSI_BITS_AND: i32 = 0;
pub const MSI_BITS_OR: i32 = 1;
pub const MSI_BITS_ANDNOT: i32 = 2;
pub const MSI_BITS_XOR: i32 = 3;
pub const MSI_TERMS_LAST: i32 = 0;
pub const MSI_TERMS_ALL: i32 = 1;
pub const MSI_TERMS_FREQUENCY: i32 = 2;
pub const MSI_NO_SLOT: u32 = 0xFFFF_FFFF;
pub const MSI_RANK_MAX_TERMS: usize = 10;

#[repr(C)]
pub struct msi_typo_query { pub word: *const u8, pub len: u32, pub max_typos: u8, pub is_prefix: u8, pub _pad: u16 }
#[repr(C)] #[derive(Clone, Copy)]
pub struct msi_rank_node { pub first_term: u32, pub last_term: u32, pub level_slot: [u32; 3], pub max_typo_cost: u32 }
#[repr(C)] #[derive(Clone, Copy, Default)]
pub struct msi_rank_bucket { pub matching_words: u32, pub typo_count: u32, pub max_typo_count: u32, pub _pad: u32, pub count: u64 }
#[repr(C)]
pub struct msi_rank_query { pub nodes: *const msi_rank_node, pub n_nodes: u32, pub n_terms: u32, pub universe_slot: u32, pub scratch_slot: u32 }
#[repr(C)]
pub struct msi_query_token { pub word: *const u8, pub len: u32, pub is_prefix: u32 }
#[repr(C)]
pub struct msi_keyword_params {
    pub authorize_typos: u32, pub min_word_len_one_typo: u32, pub min_word_len_two_typos: u32,
    pub strategy: i32, pub use_typo: i32, pub from: u32, pub length: u32,
}
pub type word_docids_fn = unsafe extern "C" fn(*mut c_void, *const u8, u32, i32, *mut *const u8, *mut usize) -> i32;
pub type pair_docids_fn = unsafe extern "C" fn(*mut c_void, u32, *const u8, u32, *const u8, u32, *mut *const u8, *mut usize) -> i32;
pub type exact_word_fn = unsafe extern "C" fn(*mut c_void, *const u8, u32) -> i32;
pub type word_key_docids_fn = unsafe extern "C" fn(*mut c_void, *const u8, u32, u32, *mut *const u8, *mut usize) -> i32;
pub type word_keys_fn = unsafe extern "C" fn(*mut c_void, *const u8, u32, *mut u16, u32, *mut u32) -> i32;
pub type msi_posting_sink = unsafe extern "C" fn(*mut c_void, *const u8, usize) -> i32;
pub type msi_synonym_sink = unsafe extern "C" fn(*mut c_void, *const msi_query_token, u32) -> i32;
pub type prefix_docids_fn = unsafe extern "C" fn(*mut c_void, *const u8, u32, i32, msi_posting_sink, *mut c_void) -> i32;
pub type prefix_key_docids_fn = unsafe extern "C" fn(*mut c_void, *const u8, u32, u32, msi_posting_sink, *mut c_void) -> i32;
pub type prefix_pair_docids_fn = unsafe extern "C" fn(*mut c_void, u32, *const u8, u32, *const u8, u32, msi_posting_sink, *mut c_void) -> i32;
pub type synonyms_fn = unsafe extern "C" fn(*mut c_void, *const msi_query_token, u32, msi_synonym_sink, *mut c_void) -> i32;
pub type exact_prefix_fn = unsafe extern "C" fn(*mut c_void, *const u8, u32, msi_synonym_sink, *mut c_void) -> i32;
pub type fid_count_docids_fn = unsafe extern "C" fn(*mut c_void, u32, u32, *mut *const u8, *mut usize) -> i32;
#[repr(C)]
pub struct msi_index_vtable {
    pub user: *mut c_void,
    pub word_docids: Option<word_docids_fn>,
    pub word_pair_proximity_docids: Option<pair_docids_fn>,
    pub is_exact_word: Option<exact_word_fn>,
    pub word_fid_docids: Option<word_key_docids_fn>,
    pub word_position_docids: Option<word_key_docids_fn>,
    pub word_fids: Option<word_keys_fn>,
    pub word_positions: Option<word_keys_fn>,
    pub field_id_word_count_docids: Option<fid_count_docids_fn>,
    pub word_prefix_docids: Option<prefix_docids_fn>,
    pub word_prefix_fid_docids: Option<prefix_key_docids_fn>,
    pub word_prefix_position_docids: Option<prefix_key_docids_fn>,
    pub word_prefix_pair_proximity_docids: Option<prefix_pair_docids_fn>,
    pub word_prefix_fids: Option<word_keys_fn>,
    pub word_prefix_positions: Option<word_keys_fn>,
    pub synonyms: Option<synonyms_fn>,
    pub exact_words_with_prefix: Option<exact_prefix_fn>,
}
pub const MSI_MAX_SCORE_DETAILS: usize = 16;
#[repr(C)] #[derive(Clone, Copy, Default)]
pub struct msi_score_detail { pub kind: u32, pub a: u32, pub b: u32 }
#[repr(C)]
pub struct msi_located_term { pub words: *const msi_query_token, pub n_words: u32, pub is_phrase: u32, pub position_start: u32, pub position_end: u32 }
#[repr(C)]
pub struct msi_search_params {
    pub authorize_typos: u32, pub min_word_len_one_typo: u32, pub min_word_len_two_typos: u32, pub strategy: i32,
    pub criteria: *const i32, pub n_criteria: u32,
    pub searchable_fids: *const u16, pub searchable_weights: *const u16, pub n_searchable: u32,
    pub max_weight: i32, pub from: u32, pub length: u32, pub detailed_scores: i32,
    pub time_budget_us: u64, pub stop_after: i32, pub has_score_threshold: i32, pub score_threshold: f64,
    pub order_keys: *const *const msi_doc_keys, pub n_order_keys: u32,
    pub distinct_values: *const msi_doc_values,
    pub geo_rules: *const msi_geo_rule, pub n_geo_rules: u32, pub geo_max_bucket_size: u32,
    pub geo_distance_error_margin: f64,
    pub exhaustive_number_hits: i32, pub max_total_hits: u32,
    pub geo_strategy: i32, pub geo_cache_size: u32,
    pub index_view: u64,
}
pub const MSI_GEO_DYNAMIC: i32 = 0;
pub const MSI_GEO_ALWAYS_ITERATIVE: i32 = 1;
pub const MSI_GEO_ALWAYS_RTREE: i32 = 2;

extern "C" {
    pub fn msi_abi_version() -> i32;
    pub fn msi_bits_geo_list(pool: *mut msi_bits, points: *const msi_geo_points, universe: u32, lat: f64, lng: f64, cap: u32,
                             out_docids: *mut u32, out_distance: *mut f64, out_total: *mut u64) -> i32;
    pub fn msi_search_compaction_stats(out: *mut u64) -> i32;
    pub fn msi_search_late_compaction_stats(out: *mut u64) -> i32;   // [sub-trees moved into their bucket's space, documents summed]
    pub fn msi_bits_vm_bytes(out: *mut u64) -> i32;
    pub fn msi_last_error() -> *const c_char;
    pub fn msi_ctx_create(device: i32, out: *mut *mut msi_ctx) -> i32;
    pub fn msi_ctx_destroy(ctx: *mut msi_ctx);
    pub fn msi_ctx_synchronize(ctx: *mut msi_ctx) -> i32;
    pub fn msi_runtime_hw_queues() -> i32;

    pub fn msi_vs_create(ctx: *mut msi_ctx, dim: u32, out: *mut *mut msi_vs) -> i32;
    pub fn msi_vs_create_typed(ctx: *mut msi_ctx, dim: u32, storage: i32, out: *mut *mut msi_vs) -> i32;
    pub fn msi_vs_destroy(vs: *mut msi_vs);
    pub fn msi_vs_update(vs: *mut msi_vs, remove_docids: *const u32, n_remove: u64, add_docids: *const u32,
                         add_rows: *const f32, n_add: u64) -> i32;
    pub fn msi_vs_upload(vs: *mut msi_vs, docids: *const u32, rows: *const f32, n_rows: u64) -> i32;
    pub fn msi_vs_len(vs: *const msi_vs) -> u64;
    pub fn msi_vs_dim(vs: *const msi_vs) -> u32;
    pub fn msi_vs_max_batch(vs: *const msi_vs) -> u32;
    pub fn msi_vs_get_vector(vs: *mut msi_vs, docid: u32, out_row: *mut f32, found: *mut i32) -> i32;
    pub fn msi_vs_search(vs: *mut msi_vs, queries: *const f32, n_queries: u32, k: u32, filter_bits: *const u64,
                         filter_nbits: u64, cancel: *const i32, out_docids: *mut u32, out_dist: *mut f32,
                         out_counts: *mut u32) -> i32;
    pub fn msi_vs_search_by_item(vs: *mut msi_vs, docid: u32, k: u32, filter_bits: *const u64, filter_nbits: u64,
                                 out_docids: *mut u32, out_dist: *mut f32, out_count: *mut u32, out_found: *mut i32) -> i32;
    pub fn msi_vs_set_microbatch(vs: *mut msi_vs, max_wait_us: u32) -> i32;
    pub fn msi_merge_topk(docids: *const u32, dist: *const f32, counts: *const u32, n_lists: u32, list_stride: u32,
                          k_out: u32, out_docids: *mut u32, out_dist: *mut f32) -> u32;

    pub fn msi_dict_create(ctx: *mut msi_ctx, words_concat: *const u8, offsets: *const u32, n_words: u32,
                           out: *mut *mut msi_dict) -> i32;
    pub fn msi_dict_destroy(d: *mut msi_dict);
    pub fn msi_dict_len(d: *const msi_dict) -> u32;
    pub fn msi_dict_lookup(d: *mut msi_dict, queries: *const msi_typo_query, n: u32, cap_one: u32, cap_two: u32,
                           out_one_idx: *mut u32, out_one_cnt: *mut u32, out_two_idx: *mut u32, out_two_cnt: *mut u32) -> i32;
    pub fn msi_dict_create_values(ctx: *mut msi_ctx, values_concat: *const u8, offsets: *const u32, n_values: u32,
                                  out: *mut *mut msi_dict) -> i32;
    pub fn msi_doc_keys_create(ctx: *mut msi_ctx, keys: *const u32, n_docs: u64, out: *mut *mut msi_doc_keys) -> i32;
    pub fn msi_doc_keys_destroy(k: *mut msi_doc_keys);
    pub fn msi_bits_order_next(p: *mut msi_bits, keys: *const msi_doc_keys, universe: u32, bucket: u32,
                               out_key: *mut u32, out_count: *mut u64) -> i32;
    pub fn msi_bits_device_ptr(p: *mut msi_bits, slot: u32) -> *const u64;
    pub fn msi_bits_set_from_words(p: *mut msi_bits, slot: u32, words: *const u64, n_words: u64) -> i32;
    pub fn msi_vs_search_device(vs: *mut msi_vs, d_queries: *const f32, n_queries: u32, k: u32, d_filter_bits: *const u64,
                                filter_nbits: u64, d_out_docids: *mut u32, d_out_dist: *mut f32, d_out_counts: *mut u32,
                                d_inexact: *mut u32) -> i32;
    pub fn msi_compare_scores(left: *const f64, n_left: u32, left_ratio: f32, right: *const f64, n_right: u32,
                              right_ratio: f32) -> i32;
    pub fn msi_facet_number_key(value: f64) -> u64;
    pub fn msi_facet_keys_create(ctx: *mut msi_ctx, offsets: *const u64, keys: *const u64, n_docs: u64,
                                 out: *mut *mut msi_facet_keys) -> i32;
    pub fn msi_facet_keys_destroy(k: *mut msi_facet_keys);
    pub fn msi_bits_facet_range(p: *mut msi_bits, keys: *const msi_facet_keys, lo: u64, hi: u64, dst: u32, accumulate: i32) -> i32;
    pub fn msi_bits_facet_in(p: *mut msi_bits, keys: *const msi_facet_keys, sorted_keys: *const u64, n: u64, dst: u32,
                             accumulate: i32) -> i32;
    pub fn msi_bits_geo_within(p: *mut msi_bits, points: *const msi_geo_points, src: u32, lat: f64, lng: f64, radius_m: f64,
                               dst: u32) -> i32;
    pub fn msi_geo_points_create(ctx: *mut msi_ctx, lat_lng: *const f64, n_docs: u64, out: *mut *mut msi_geo_points) -> i32;
    pub fn msi_geo_points_destroy(g: *mut msi_geo_points);
    pub fn msi_bits_geo_next(p: *mut msi_bits, points: *const msi_geo_points, universe: u32, bucket: u32, scratch: u32,
                             lat: f64, lng: f64, ascending: i32, max_bucket_size: u32, distance_error_margin: f64,
                             out_first_docid: *mut u32, out_count: *mut u64) -> i32;
    pub fn msi_doc_values_create(ctx: *mut msi_ctx, offsets: *const u64, value_ids: *const u32, n_docs: u64, n_values: u32,
                                 out: *mut *mut msi_doc_values) -> i32;
    pub fn msi_doc_values_destroy(v: *mut msi_doc_values);
    pub fn msi_bits_distinct(p: *mut msi_bits, values: *const msi_doc_values, candidates: u32, remaining: u32, excluded: u32,
                             out_remaining: *mut u64, out_rounds: *mut u32) -> i32;
    pub fn msi_bits_distinct_excluded(p: *mut msi_bits, values: *const msi_doc_values, kept: u32, excluded: u32) -> i32;
    pub fn msi_bits_andnot_many_count(p: *mut msi_bits, removed: u32, n: u32, slots: *const u32, out_counts: *mut u64) -> i32;
    pub fn msi_fst_decode(fst: *const u8, len: usize, flags: u32, out_concat: *mut u8, cap_bytes: u64,
                          out_offsets: *mut u32, cap_words: u32, out_n_words: *mut u32, out_n_bytes: *mut u64) -> i32;
    pub fn msi_dict_create_from_fst(ctx: *mut msi_ctx, fst: *const u8, len: usize, out: *mut *mut msi_dict) -> i32;
    pub fn msi_dict_create_values_from_fst(ctx: *mut msi_ctx, fst: *const u8, len: usize, out: *mut *mut msi_dict) -> i32;
    pub fn msi_dict_search_values(d: *mut msi_dict, query: *const u8, len: u32, max_typos: u32, cap: u32,
                                  out_idx: *mut u32, out_n: *mut u32, out_truncated: *mut i32) -> i32;
    pub fn msi_bits_set_from_docid_lists_device(p: *mut msi_bits, first_slot: u32, slot_stride: u32, d_docids: *const u32,
                                                list_stride: u32, d_counts: *const u32, n_lists: u32) -> i32;
    pub fn msi_dict_set_microbatch(d: *mut msi_dict, max_wait_us: u32, target_words: u32) -> i32;
    /// HBM cache of the index version's stored postings (hot keys are decoded from HBM, not over PCIe).
    pub fn msi_dict_enable_posting_cache(d: *mut msi_dict, capacity_bytes: u64) -> i32;
    /// out: [hits, misses, bytes used, capacity]
    pub fn msi_dict_posting_cache_stats(d: *mut msi_dict, out: *mut u64) -> i32;

    pub fn msi_bits_create(ctx: *mut msi_ctx, n_docs: u64, n_slots: u32, out: *mut *mut msi_bits) -> i32;
    pub fn msi_bits_destroy(p: *mut msi_bits);
    pub fn msi_bits_set_from_cbo(p: *mut msi_bits, slot: u32, bytes: *const u8, len: usize) -> i32;
    pub fn msi_bits_set_from_docids(p: *mut msi_bits, slot: u32, docids: *const u32, n: u64) -> i32;
    pub fn msi_bits_fill(p: *mut msi_bits, slot: u32, ones: i32) -> i32;
    pub fn msi_bits_op(p: *mut msi_bits, dst: u32, a: u32, b: u32, op: i32) -> i32;
    pub fn msi_bits_union_many_and(p: *mut msi_bits, dst: u32, srcs: *const u32, n: u32, universe: u32) -> i32;
    pub fn msi_bits_count(p: *mut msi_bits, slot: u32, out: *mut u64) -> i32;
    pub fn msi_bits_first_k(p: *mut msi_bits, slot: u32, k: u32, out: *mut u32, out_n: *mut u32) -> i32;
    pub fn msi_bits_read_words(p: *mut msi_bits, slot: u32, out_words: *mut u64) -> i32;

    pub fn msi_rank_query_graph(p: *mut msi_bits, nodes: *const msi_rank_node, n_nodes: u32, n_terms: u32,
                                universe_slot: u32, scratch_slot: u32, strategy: i32, use_typo: i32, from: u32,
                                length: u32, out_docids: *mut u32, out_matching_words: *mut u32,
                                out_typo_count: *mut u32, out_max_typo_count: *mut u32, out_n: *mut u32,
                                out_candidates: *mut u64) -> i32;
    pub fn msi_rank_query_graph_batch(p: *mut msi_bits, queries: *const msi_rank_query, n_queries: u32, strategy: i32,
                                      use_typo: i32, from: u32, length: u32, out_docids: *mut u32,
                                      out_matching_words: *mut u32, out_typo_count: *mut u32,
                                      out_max_typo_count: *mut u32, out_n: *mut u32, out_candidates: *mut u64) -> i32;
    pub fn msi_rank_buckets(p: *mut msi_bits, nodes: *const msi_rank_node, n_nodes: u32, n_terms: u32,
                            universe_slot: u32, scratch_slot: u32, strategy: i32, use_typo: i32,
                            out_buckets: *mut msi_rank_bucket, cap: u32, out_n: *mut u32) -> i32;
    pub fn msi_rank_materialise(p: *mut msi_bits, nodes: *const msi_rank_node, n_nodes: u32, n_terms: u32,
                                universe_slot: u32, strategy: i32, use_typo: i32, matching_words: u32,
                                typo_count: u32, dst_slot: u32) -> i32;
    pub fn msi_keyword_search(d: *mut msi_dict, p: *mut msi_bits, index: *const msi_index_vtable,
                              tokens: *const msi_query_token, n_tokens: u32, params: *const msi_keyword_params,
                              universe_cbo: *const u8, universe_len: usize, out_docids: *mut u32,
                              out_matching_words: *mut u32, out_typo_count: *mut u32, out_max_typo_count: *mut u32,
                              out_n: *mut u32, out_candidates: *mut u64) -> i32;

    pub fn msi_keyword_search_ranked(d: *mut msi_dict, p: *mut msi_bits, index: *const msi_index_vtable,
                                     terms: *const msi_located_term, n_terms: u32, params: *const msi_search_params,
                                     universe_cbo: *const u8, universe_len: usize, out_docids: *mut u32,
                                     out_scores: *mut msi_score_detail, out_n_scores: *mut u32, out_n: *mut u32,
                                     out_candidates: *mut u64, out_degraded: *mut i32) -> i32;
    pub fn msi_score_details_global_score(details: *const msi_score_detail, n: u32) -> f64;
    pub fn msi_bits_use_private_stream(p: *mut msi_bits) -> i32;
    pub fn msi_bits_op_count(p: *mut msi_bits, dst: u32, a: u32, b: u32, op: i32, out_count: *mut u64) -> i32;

    pub fn msi_vector_sort(docids: *const u32, dist: *const f32, n: u32, has_shift: i32, mean: f32, sigma: f32,
                           from: u32, length: u32, out_docids: *mut u32, out_similarity: *mut f32) -> u32;
    pub fn msi_hybrid_merge(v_docids: *const u32, v_scores: *const f64, v_off: *const u32, n_v: u32, v_ratio: f32,
                            k_docids: *const u32, k_scores: *const f64, k_off: *const u32, n_k: u32, k_ratio: f32,
                            from: u32, length: u32, out_docids: *mut u32, out_is_semantic: *mut u8,
                            out_semantic_hit_count: *mut u32) -> u32;
    pub fn msi_results_good_enough(keyword_global_scores: *const f64, n: u32, limit_plus_offset: u32,
                                   semantic_ratio: f32) -> i32;
    pub fn msi_distribution_shift(mean: f32, sigma: f32, score: f32) -> f32;
    pub fn msi_rank_global_score(ranks: *const u32, max_ranks: *const u32, n: u32) -> f64;
    // ---- round 2 ---------------------------------------------------------------------------------------------------
    // binary-quantised stores (vector/store.rs:1095-1109)
    pub fn msi_bq_create(ctx: *mut msi_ctx, dim: u32, out: *mut *mut msi_bq) -> i32;
    pub fn msi_bq_destroy(bq: *mut msi_bq);
    pub fn msi_bq_upload(bq: *mut msi_bq, docids: *const u32, rows: *const f32, n_rows: u64) -> i32;
    pub fn msi_bq_upload_device(bq: *mut msi_bq, d_docids: *const u32, d_rows: *const f32, n_rows: u64) -> i32;
    pub fn msi_bq_len(bq: *const msi_bq) -> u64;
    pub fn msi_bq_dim(bq: *const msi_bq) -> u32;
    pub fn msi_bq_get_vector(bq: *mut msi_bq, docid: u32, out_row: *mut f32, out_found: *mut i32) -> i32;
    pub fn msi_bq_search(bq: *mut msi_bq, queries: *const f32, n_queries: u32, k: u32, filter_bits: *const u64,
                         filter_nbits: u64, out_docids: *mut u32, out_dist: *mut f32, out_counts: *mut u32) -> i32;
    // federated merge (search/federated/weighted_scores.rs, perform.rs)
    pub fn msi_federated_compare(left: *const msi_weighted_value, n_left: u32, left_weighted_global_score: f64,
                                 right: *const msi_weighted_value, n_right: u32, right_weighted_global_score: f64) -> i32;
    pub fn msi_federated_merge(n_lists: u32, list_len: *const u32, values: *const *const msi_weighted_value,
                               val_off: *const *const u32, weighted_global: *const *const f64, offset: u32, limit: u32,
                               out_list: *mut u32, out_pos: *mut u32) -> u32;
    pub fn msi_federated_merge_q(n_lists: u32, list_len: *const u32, values: *const *const msi_weighted_value,
                                 val_off: *const *const u32, weighted_global: *const *const f64,
                                 query_index: *const *const u32, offset: u32, limit: u32, out_list: *mut u32,
                                 out_pos: *mut u32) -> u32;
    // multi-GPU (RCCL inside the library)
    pub fn msi_group_create(devices: *const i32, n: u32, out: *mut *mut msi_group) -> i32;
    pub fn msi_group_unique_id(out_id: *mut u8) -> i32;                       // 128 bytes
    pub fn msi_group_create_rank(ctx: *mut msi_ctx, rank: u32, world: u32, id: *const u8, out: *mut *mut msi_group) -> i32;
    pub fn msi_group_destroy(group: *mut msi_group);
    pub fn msi_group_size(group: *const msi_group) -> u32;
    pub fn msi_group_ctx(group: *mut msi_group, i: u32) -> *mut msi_ctx;
    pub fn msi_group_allgather(group: *mut msi_group, d_send: *const c_void, bytes: usize, d_recv: *mut c_void) -> i32;
    pub fn msi_vs_group_create(group: *mut msi_group, dim: u32, storage: i32, mode: i32, out: *mut *mut msi_vs_group) -> i32;
    pub fn msi_vs_group_destroy(vs: *mut msi_vs_group);
    pub fn msi_vs_group_upload(vs: *mut msi_vs_group, docids: *const u32, rows: *const f32, n_rows: u64) -> i32;
    pub fn msi_vs_group_search(vs: *mut msi_vs_group, queries: *const f32, n_queries: u32, k: u32, out_docids: *mut u32,
                               out_dist: *mut f32, out_counts: *mut u32) -> i32;
    // `_vectors` filter leaf (search/facet/filter/vector.rs:49-158)
    pub fn msi_vs_items_bits(store: *mut msi_vs, pool: *mut msi_bits, slot: u32) -> i32;
    pub fn msi_bq_items_bits(store: *mut msi_bq, pool: *mut msi_bits, slot: u32) -> i32;
    pub fn msi_bits_vector_filter(pool: *mut msi_bits, dst: u32, kind: i32, embedder_has_fragments: i32,
                                  stores: *const *mut msi_vs, n_stores: u32, bq_stores: *const *mut msi_bq, n_bq_stores: u32,
                                  user_provided: u32, skip_regenerate: u32, scratch: u32, accumulate: i32) -> i32;
    // command-list back end statistics: rounds, lists, ns queued / packed / in launch calls / after launch
    pub fn msi_bits_vm_stats(pool: *mut msi_bits, out: *mut u64) -> i32;
}
