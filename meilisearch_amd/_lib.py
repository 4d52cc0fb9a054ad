"""ctypes binding of include/msi.h.  Loud failure if libmsi.so is missing."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

OK = 0
ERRORS = {
    -1: "MSI_E_INVALID", -2: "MSI_E_NO_DEVICE", -3: "MSI_E_HIP", -4: "MSI_E_OOM",
    -5: "MSI_E_UNSUPPORTED", -6: "MSI_E_CANCELLED", -7: "MSI_E_NOT_SORTED", -8: "MSI_E_INTERNAL",
}


class MsiError(RuntimeError):
    def __init__(self, status, message):
        super().__init__(f"{ERRORS.get(status, status)}: {message}")
        self.status = status


class TypoQuery(C.Structure):
    _fields_ = [("word", C.c_void_p), ("len", C.c_uint32), ("max_typos", C.c_uint8),
                ("is_prefix", C.c_uint8), ("_pad", C.c_uint16)]


class RankTerm(C.Structure):
    _fields_ = [("level_slot", C.c_uint32 * 3), ("max_typo_cost", C.c_uint32)]


class RankQuery(C.Structure):
    _fields_ = [("nodes", C.c_void_p), ("n_nodes", C.c_uint32), ("n_terms", C.c_uint32),
                ("universe_slot", C.c_uint32), ("scratch_slot", C.c_uint32)]


class RankBucket(C.Structure):
    _fields_ = [("matching_words", C.c_uint32), ("typo_count", C.c_uint32), ("max_typo_count", C.c_uint32),
                ("_pad", C.c_uint32), ("count", C.c_uint64)]


class RankNode(C.Structure):
    _fields_ = [("first_term", C.c_uint32), ("last_term", C.c_uint32), ("level_slot", C.c_uint32 * 3),
                ("max_typo_cost", C.c_uint32)]


WORD_DOCIDS_FN = C.CFUNCTYPE(C.c_int32, C.c_void_p, C.POINTER(C.c_uint8), C.c_uint32, C.c_int32,
                            C.POINTER(C.POINTER(C.c_uint8)), C.POINTER(C.c_size_t))
PAIR_DOCIDS_FN = C.CFUNCTYPE(C.c_int32, C.c_void_p, C.c_uint32, C.POINTER(C.c_uint8), C.c_uint32,
                            C.POINTER(C.c_uint8), C.c_uint32, C.POINTER(C.POINTER(C.c_uint8)), C.POINTER(C.c_size_t))
EXACT_WORD_FN = C.CFUNCTYPE(C.c_int32, C.c_void_p, C.POINTER(C.c_uint8), C.c_uint32)


WORD_KEY_DOCIDS_FN = C.CFUNCTYPE(C.c_int32, C.c_void_p, C.POINTER(C.c_uint8), C.c_uint32, C.c_uint32,
                                C.POINTER(C.POINTER(C.c_uint8)), C.POINTER(C.c_size_t))
WORD_KEYS_FN = C.CFUNCTYPE(C.c_int32, C.c_void_p, C.POINTER(C.c_uint8), C.c_uint32, C.POINTER(C.c_uint16), C.c_uint32,
                          C.POINTER(C.c_uint32))
FID_COUNT_DOCIDS_FN = C.CFUNCTYPE(C.c_int32, C.c_void_p, C.c_uint32, C.c_uint32, C.POINTER(C.POINTER(C.c_uint8)),
                                 C.POINTER(C.c_size_t))


POSTING_SINK_FN = C.CFUNCTYPE(C.c_int32, C.c_void_p, C.POINTER(C.c_uint8), C.c_size_t)
PREFIX_DOCIDS_FN = C.CFUNCTYPE(C.c_int32, C.c_void_p, C.POINTER(C.c_uint8), C.c_uint32, C.c_int32, POSTING_SINK_FN, C.c_void_p)
PREFIX_KEY_DOCIDS_FN = C.CFUNCTYPE(C.c_int32, C.c_void_p, C.POINTER(C.c_uint8), C.c_uint32, C.c_uint32, POSTING_SINK_FN,
                                  C.c_void_p)
PREFIX_PAIR_DOCIDS_FN = C.CFUNCTYPE(C.c_int32, C.c_void_p, C.c_uint32, C.POINTER(C.c_uint8), C.c_uint32,
                                   C.POINTER(C.c_uint8), C.c_uint32, POSTING_SINK_FN, C.c_void_p)


SYNONYM_SINK_FN = C.CFUNCTYPE(C.c_int32, C.c_void_p, C.c_void_p, C.c_uint32)
SYNONYMS_FN = C.CFUNCTYPE(C.c_int32, C.c_void_p, C.c_void_p, C.c_uint32, SYNONYM_SINK_FN, C.c_void_p)


EXACT_PREFIX_FN = C.CFUNCTYPE(C.c_int32, C.c_void_p, C.POINTER(C.c_uint8), C.c_uint32, SYNONYM_SINK_FN, C.c_void_p)


class IndexVtable(C.Structure):
    _fields_ = [("user", C.c_void_p), ("word_docids", WORD_DOCIDS_FN),
                ("word_pair_proximity_docids", PAIR_DOCIDS_FN), ("is_exact_word", EXACT_WORD_FN),
                ("word_fid_docids", WORD_KEY_DOCIDS_FN), ("word_position_docids", WORD_KEY_DOCIDS_FN),
                ("word_fids", WORD_KEYS_FN), ("word_positions", WORD_KEYS_FN),
                ("field_id_word_count_docids", FID_COUNT_DOCIDS_FN),
                ("word_prefix_docids", PREFIX_DOCIDS_FN), ("word_prefix_fid_docids", PREFIX_KEY_DOCIDS_FN),
                ("word_prefix_position_docids", PREFIX_KEY_DOCIDS_FN),
                ("word_prefix_pair_proximity_docids", PREFIX_PAIR_DOCIDS_FN),
                ("word_prefix_fids", WORD_KEYS_FN), ("word_prefix_positions", WORD_KEYS_FN),
                ("synonyms", SYNONYMS_FN), ("exact_words_with_prefix", EXACT_PREFIX_FN)]


class ScoreDetail(C.Structure):
    _fields_ = [("kind", C.c_uint32), ("a", C.c_uint32), ("b", C.c_uint32)]


class LocatedTerm(C.Structure):
    _fields_ = [("words", C.c_void_p), ("n_words", C.c_uint32), ("is_phrase", C.c_uint32),
                ("position_start", C.c_uint32), ("position_end", C.c_uint32)]


class SearchParams(C.Structure):
    _fields_ = [("authorize_typos", C.c_uint32), ("min_word_len_one_typo", C.c_uint32),
                ("min_word_len_two_typos", C.c_uint32), ("strategy", C.c_int32), ("criteria", C.c_void_p),
                ("n_criteria", C.c_uint32), ("searchable_fids", C.c_void_p), ("searchable_weights", C.c_void_p),
                ("n_searchable", C.c_uint32), ("max_weight", C.c_int32), ("from_", C.c_uint32),
                ("length", C.c_uint32), ("detailed_scores", C.c_int32), ("time_budget_us", C.c_uint64),
                ("stop_after", C.c_int32), ("has_score_threshold", C.c_int32), ("score_threshold", C.c_double),
                ("order_keys", C.c_void_p), ("n_order_keys", C.c_uint32), ("distinct_values", C.c_void_p), ("geo_rules", C.c_void_p), ("n_geo_rules", C.c_uint32),
                ("geo_max_bucket_size", C.c_uint32), ("geo_distance_error_margin", C.c_double),
                ("exhaustive_number_hits", C.c_int32), ("max_total_hits", C.c_uint32),
                ("geo_strategy", C.c_int32), ("geo_cache_size", C.c_uint32), ("index_view", C.c_uint64)]


class GeoRule(C.Structure):
    _fields_ = [("points", C.c_void_p), ("lat", C.c_double), ("lng", C.c_double), ("ascending", C.c_int32)]


class QueryToken(C.Structure):
    _fields_ = [("word", C.c_void_p), ("len", C.c_uint32), ("is_prefix", C.c_uint32)]


class KeywordParams(C.Structure):
    _fields_ = [("authorize_typos", C.c_uint32), ("min_word_len_one_typo", C.c_uint32),
                ("min_word_len_two_typos", C.c_uint32), ("strategy", C.c_int32), ("use_typo", C.c_int32),
                ("from_", C.c_uint32), ("length", C.c_uint32)]


class VsStats(C.Structure):
    _fields_ = [("scan_launches", C.c_uint64), ("scan_tiles", C.c_uint64),
                ("exhaustive_reruns", C.c_uint64), ("bytes_per_tile", C.c_uint64),
                ("second_opinion_queries", C.c_uint64), ("x3_first_sweeps", C.c_uint64), ("x2_sweeps", C.c_uint64),
                ("level_sweeps", C.c_uint64 * 3),
                ("i8_bytes_per_tile", C.c_uint64), ("i8_sweeps", C.c_uint64), ("device_rerun_queries", C.c_uint64),
                ("i8_queries_per_sweep", C.c_uint64), ("f32_queries_per_sweep", C.c_uint64), ("i8_scan_tiles", C.c_uint64)]


class DictStats(C.Structure):
    _fields_ = [("lookup_launches", C.c_uint64), ("pairs_scanned", C.c_uint64),
                ("dict_bytes", C.c_uint64)]


def lib_path():
    return os.path.join(_HERE, "libmsi.so")


# name -> (restype, argtypes); every symbol include/msi.h declares
_VP = C.c_void_p
_U32, _U64, _I32, _F32, _F64 = C.c_uint32, C.c_uint64, C.c_int32, C.c_float, C.c_double
PROTOTYPES = {
    "msi_abi_version": (_I32, []),
    "msi_last_error": (C.c_char_p, []),
    "msi_ctx_create": (_I32, [_I32, C.POINTER(_VP)]),
    "msi_ctx_destroy": (None, [_VP]),
    "msi_ctx_stream": (_VP, [_VP]),
    "msi_ctx_synchronize": (_I32, [_VP]),
    "msi_ctx_device": (_I32, [_VP]),
    "msi_ctx_set_profiling": (_I32, [_VP, _I32]),
    "msi_runtime_hw_queues": (_I32, []),
    "msi_runtime_hw_queues_source": (_I32, []),
    "msi_vs_create": (_I32, [_VP, _U32, C.POINTER(_VP)]),
    "msi_vs_create_typed": (_I32, [_VP, _U32, _I32, C.POINTER(_VP)]),
    "msi_vs_destroy": (None, [_VP]),
    "msi_vs_upload": (_I32, [_VP, _VP, _VP, _U64]),
    "msi_vs_upload_device": (_I32, [_VP, _VP, _VP, _U64]),
    "msi_vs_update": (_I32, [_VP, _VP, _U64, _VP, _VP, _U64]),
    "msi_vs_len": (_U64, [_VP]),
    "msi_vs_dim": (_U32, [_VP]),
    "msi_vs_max_batch": (_U32, [_VP]),
    "msi_vs_get_vector": (_I32, [_VP, _U32, _VP, C.POINTER(_I32)]),
    "msi_vs_search": (_I32, [_VP, _VP, _U32, _U32, _VP, _U64, _VP, _VP, _VP, _VP]),
    "msi_vs_search_by_item": (_I32, [_VP, _U32, _U32, _VP, _U64, _VP, _VP, C.POINTER(_U32), C.POINTER(_I32)]),
    "msi_vs_set_microbatch": (_I32, [_VP, _U32]),
    "msi_vs_set_sweep_split": (_I32, [_VP, _U32]),
    "msi_vs_microbatch_stats": (_I32, [_VP, C.POINTER(_U64), C.POINTER(_U64)]),
    "msi_vs_search_device": (_I32, [_VP, _VP, _U32, _U32, _VP, _U64, _VP, _VP, _VP, _VP]),
    "msi_merge_topk": (_U32, [_VP, _VP, _VP, _U32, _U32, _U32, _VP, _VP]),
    "msi_merge_topk_device": (_I32, [_VP, _VP, _VP, _VP, _U32, _U32, _U32, _VP, _VP, _VP]),
    "msi_vs_get_stats": (_I32, [_VP, C.POINTER(VsStats)]),
    "msi_vs_debug_fast_scores": (_I32, [_VP, _VP, _U32, _VP, C.POINTER(_F32)]),
    "msi_vs_scan_time": (_I32, [_VP, C.POINTER(_U64), C.POINTER(_F64)]),
    "msi_vs_filter_stats": (_I32, [_VP, C.POINTER(_U64)]),
    "msi_group_create": (_I32, [_VP, _U32, C.POINTER(_VP)]),
    "msi_group_unique_id": (_I32, [_VP]),
    "msi_group_create_rank": (_I32, [_VP, _U32, _U32, _VP, C.POINTER(_VP)]),
    "msi_group_destroy": (None, [_VP]),
    "msi_group_size": (_U32, [_VP]),
    "msi_group_ctx": (_VP, [_VP, _U32]),
    "msi_group_allgather": (_I32, [_VP, _VP, C.c_size_t, _VP]),
    "msi_vs_group_create": (_I32, [_VP, _U32, _I32, _I32, C.POINTER(_VP)]),
    "msi_vs_group_destroy": (None, [_VP]),
    "msi_vs_group_upload": (_I32, [_VP, _VP, _VP, _U64]),
    "msi_vs_group_search": (_I32, [_VP, _VP, _U32, _U32, _VP, _VP, _VP]),
    "msi_bq_create": (_I32, [_VP, _U32, C.POINTER(_VP)]),
    "msi_bq_destroy": (None, [_VP]),
    "msi_bq_upload": (_I32, [_VP, _VP, _VP, _U64]),
    "msi_bq_upload_device": (_I32, [_VP, _VP, _VP, _U64]),
    "msi_bq_len": (_U64, [_VP]),
    "msi_bq_dim": (_U32, [_VP]),
    "msi_bq_get_vector": (_I32, [_VP, _U32, _VP, C.POINTER(_I32)]),
    "msi_bq_search": (_I32, [_VP, _VP, _U32, _U32, _VP, _U64, _VP, _VP, _VP]),
    "msi_vs_items_bits": (_I32, [_VP, _VP, _U32]),
    "msi_bq_items_bits": (_I32, [_VP, _VP, _U32]),
    "msi_bits_vector_filter": (_I32, [_VP, _U32, _I32, _I32, _VP, _U32, _VP, _U32, _U32, _U32, _U32, _I32]),
    "msi_federated_compare": (_I32, [_VP, _U32, _F64, _VP, _U32, _F64]),
    "msi_federated_merge": (_U32, [_U32, _VP, _VP, _VP, _VP, _U32, _U32, _VP, _VP]),
    "msi_federated_merge_q": (_U32, [_U32, _VP, _VP, _VP, _VP, _VP, _U32, _U32, _VP, _VP]),
    "msi_dict_create": (_I32, [_VP, _VP, _VP, _U32, C.POINTER(_VP)]),
    "msi_dict_destroy": (None, [_VP]),
    "msi_dict_len": (_U32, [_VP]),
    "msi_dict_lookup": (_I32, [_VP, C.POINTER(TypoQuery), _U32, _U32, _U32, _VP, _VP, _VP, _VP]),
    "msi_dict_create_values": (_I32, [_VP, _VP, _VP, _U32, C.POINTER(_VP)]),
    "msi_doc_keys_create": (_I32, [_VP, _VP, C.c_uint64, C.POINTER(_VP)]),
    "msi_doc_keys_destroy": (None, [_VP]),
    "msi_bits_order_next": (_I32, [_VP, _VP, _U32, _U32, C.POINTER(_U32), C.POINTER(C.c_uint64)]),
    "msi_facet_number_key": (_U64, [_F64]),
    "msi_facet_keys_create": (_I32, [_VP, _VP, _VP, C.c_uint64, C.POINTER(_VP)]),
    "msi_facet_keys_destroy": (None, [_VP]),
    "msi_bits_facet_range": (_I32, [_VP, _VP, _U64, _U64, _U32, _I32]),
    "msi_bits_facet_in": (_I32, [_VP, _VP, _VP, _U64, _U32, _I32]),
    "msi_bits_geo_within": (_I32, [_VP, _VP, _U32, _F64, _F64, _F64, _U32]),
    "msi_geo_points_create": (_I32, [_VP, _VP, C.c_uint64, C.POINTER(_VP)]),
    "msi_geo_points_destroy": (None, [_VP]),
    "msi_bits_geo_next": (_I32, [_VP, _VP, _U32, _U32, _U32, _F64, _F64, _I32, _U32, _F64, C.POINTER(_U32),
                                 C.POINTER(C.c_uint64)]),
    "msi_doc_values_create": (_I32, [_VP, _VP, _VP, C.c_uint64, _U32, C.POINTER(_VP)]),
    "msi_doc_values_destroy": (None, [_VP]),
    "msi_bits_distinct": (_I32, [_VP, _VP, _U32, _U32, _U32, C.POINTER(C.c_uint64), C.POINTER(_U32)]),
    "msi_bits_distinct_excluded": (_I32, [_VP, _VP, _U32, _U32]),
    "msi_bits_andnot_many_count": (_I32, [_VP, _U32, _U32, _VP, _VP]),
    "msi_fst_decode": (_I32, [_VP, C.c_size_t, _U32, _VP, C.c_uint64, _VP, _U32, C.POINTER(_U32), C.POINTER(C.c_uint64)]),
    "msi_dict_create_from_fst": (_I32, [_VP, _VP, C.c_size_t, C.POINTER(_VP)]),
    "msi_dict_create_values_from_fst": (_I32, [_VP, _VP, C.c_size_t, C.POINTER(_VP)]),
    "msi_dict_search_values": (_I32, [_VP, _VP, _U32, _U32, _U32, _VP, C.POINTER(_U32), C.POINTER(_I32)]),
    "msi_dict_set_microbatch": (_I32, [_VP, _U32, _U32]),
    "msi_dict_microbatch_stats": (_I32, [_VP, C.POINTER(_U64), C.POINTER(_U64)]),
    "msi_dict_enable_posting_cache": (_I32, [_VP, _U64]),
    "msi_dict_posting_cache_stats": (_I32, [_VP, C.POINTER(_U64)]),
    "msi_dict_reset_posting_cache": (_I32, [_VP]),
    "msi_inject_pins": (_U32, [_VP, _U32, _U32, _U32, _VP, _VP, _VP, _U32, _VP, _VP, _VP]),
    "msi_dict_stage_postings": (_I32, [_VP, _U64, _VP, _U64, C.POINTER(_U64)]),
    "msi_dict_stage_complete": (_I32, [_VP, _U64, _U32]),
    "msi_dict_staged_stats": (_I32, [_VP, C.POINTER(_U64)]),
    "msi_dict_lookup_device": (_I32, [_VP, _VP, _VP, _VP, _U32, _U32, _U32, _VP, _VP, _VP, _VP]),
    "msi_dict_get_stats": (_I32, [_VP, C.POINTER(DictStats)]),
    "msi_dict_match_time": (_I32, [_VP, C.POINTER(_U64), C.POINTER(_F64)]),
    "msi_bits_create": (_I32, [_VP, _U64, _U32, C.POINTER(_VP)]),
    "msi_bits_destroy": (None, [_VP]),
    "msi_bits_use_private_stream": (_I32, [_VP]),
    "msi_bits_vm_stats": (_I32, [_VP, C.POINTER(_U64)]),
    "msi_bits_set_from_docids": (_I32, [_VP, _U32, _VP, _U64]),
    "msi_bits_set_from_docid_lists_device": (_I32, [_VP, _U32, _U32, _VP, _U32, _VP, _U32]),
    "msi_bits_set_from_cbo": (_I32, [_VP, _U32, _VP, C.c_size_t]),
    "msi_bits_set_from_words": (_I32, [_VP, _U32, _VP, _U64]),
    "msi_bits_fill": (_I32, [_VP, _U32, _I32]),
    "msi_bits_op": (_I32, [_VP, _U32, _U32, _U32, _I32]),
    "msi_bits_op_count": (_I32, [_VP, _U32, _U32, _U32, _I32, C.POINTER(_U64)]),
    "msi_bits_union_many_and": (_I32, [_VP, _U32, _VP, _U32, _U32]),
    "msi_bits_count": (_I32, [_VP, _U32, C.POINTER(_U64)]),
    "msi_bits_first_k": (_I32, [_VP, _U32, _U32, _VP, C.POINTER(_U32)]),
    "msi_bits_read_words": (_I32, [_VP, _U32, _VP]),
    "msi_bits_device_ptr": (_VP, [_VP, _U32]),
    "msi_rank_query_graph": (_I32, [_VP, _VP, _U32, _U32, _U32, _U32, _I32, _I32, _U32, _U32, _VP, _VP, _VP, _VP,
                                    C.POINTER(_U32), C.POINTER(_U64)]),
    "msi_rank_query_graph_batch": (_I32, [_VP, _VP, _U32, _I32, _I32, _U32, _U32, _VP, _VP, _VP, _VP, _VP, _VP]),
    "msi_rank_buckets": (_I32, [_VP, _VP, _U32, _U32, _U32, _U32, _I32, _I32, _VP, _U32, C.POINTER(_U32)]),
    "msi_rank_materialise": (_I32, [_VP, _VP, _U32, _U32, _U32, _I32, _I32, _U32, _U32, _U32]),
    "msi_rank_words_typo": (_I32, [_VP, _VP, _U32, _U32, _U32, _I32, _I32, _U32, _U32, _VP, _VP, _VP, _VP,
                                   C.POINTER(_U32), C.POINTER(_U64)]),
    "msi_vector_sort": (_U32, [_VP, _VP, _U32, _I32, _F32, _F32, _U32, _U32, _VP, _VP]),
    "msi_hybrid_merge": (_U32, [_VP, _VP, _VP, _U32, _F32, _VP, _VP, _VP, _U32, _F32, _U32, _U32, _VP, _VP,
                                C.POINTER(_U32)]),
    "msi_hybrid_merge_batch": (_I32, [_VP, _VP, _VP, _U32, _VP, _VP, _VP, _VP, _VP, _U32, _VP, _U32, _F32, _U32, _U32,
                                      _VP, _VP, _VP, _VP]),
    "msi_results_good_enough": (_I32, [_VP, _U32, _U32, _F32]),
    "msi_keyword_search": (_I32, [_VP, _VP, C.POINTER(IndexVtable), C.POINTER(QueryToken), _U32,
                                  C.POINTER(KeywordParams), _VP, C.c_size_t, _VP, _VP, _VP, _VP,
                                  C.POINTER(_U32), C.POINTER(_U64)]),
    "msi_keyword_search_ranked": (_I32, [_VP, _VP, C.POINTER(IndexVtable), C.POINTER(LocatedTerm), _U32,
                                         C.POINTER(SearchParams), _VP, C.c_size_t, _VP, _VP, _VP,
                                         C.POINTER(_U32), C.POINTER(_U64), C.POINTER(_I32)]),
    "msi_search_last_stats": (_I32, [C.POINTER(_U64)]),
    "msi_search_cpu_profile": (_I32, [C.POINTER(_U64)]),
    "msi_search_cpu_profile_enable": (_I32, [_I32]),
    "msi_search_compaction_stats": (_I32, [C.POINTER(_U64)]),
    "msi_search_late_compaction_stats": (_I32, [C.POINTER(_U64)]),
    "msi_bits_vm_bytes": (_I32, [C.POINTER(_U64)]),
    "msi_bits_geo_list": (_I32, [_VP, _VP, _U32, C.c_double, C.c_double, _U32, _VP, _VP, C.POINTER(_U64)]),
    "msi_score_details_global_score": (_F64, [_VP, _U32]),
    "msi_distribution_shift": (_F32, [_F32, _F32, _F32]),
    "msi_rank_global_score": (_F64, [_VP, _VP, _U32]),
    "msi_compare_scores": (_I32, [_VP, _U32, _F32, _VP, _U32, _F32]),
}


def _share_torch_hip_runtime():
    """One HIP runtime per process.  PyTorch wheels bundle their own
    libamdhip64.so (SONAME libamdhip64.so.7) and ask for it as "libamdhip64.so";
    libmsi.so asks for "libamdhip64.so.7".  If libmsi is loaded first the dynamic
    loader resolves it to /opt/rocm's copy and a later `import torch` maps a SECOND
    runtime ("No HIP GPUs are available").  When torch is installed, map its copy
    first so that both resolve to the same object (torch is not imported here and
    is not required: without it libmsi uses the system ROCm runtime)."""
    import importlib.util
    import sys
    if "torch" in sys.modules:
        return
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.submodule_search_locations:
        return
    cand = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
    if os.path.exists(cand):
        try:
            C.CDLL(cand, mode=C.RTLD_GLOBAL)
        except OSError:
            pass


def on_device(tensor):
    """The tensor's memory is what libmsi may be handed as a device pointer: a CUDA (HIP) tensor — or any tensor when the
    test tier has swapped in the CPU-emulated build of the library (tests/emu: "device" pointers are host pointers)."""
    return bool(tensor.is_cuda) or type(_LIB).__name__ == "EmulatedLib"


def lib():
    """Load libmsi.so.  No fallback: a missing library is an error."""
    global _LIB
    if _LIB is None:
        path = lib_path()
        if not os.path.exists(path):
            raise ImportError(
                f"{path} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc --offload-arch=gfx950).  meilisearch_amd has no CPU fallback.")
        _share_torch_hip_runtime()
        L = C.CDLL(path)
        for name, (res, args) in PROTOTYPES.items():
            fn = getattr(L, name)  # AttributeError if the ABI is incomplete
            fn.restype = res
            fn.argtypes = args
        if L.msi_abi_version() != 3:
            raise ImportError(f"libmsi ABI {L.msi_abi_version()} != 3")
        _LIB = L
    return _LIB


def abi_version():
    return lib().msi_abi_version()


def check(status):
    if status != OK:
        raise MsiError(status, lib().msi_last_error().decode("utf-8", "replace"))
