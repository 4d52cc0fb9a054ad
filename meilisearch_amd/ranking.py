"""Binding + host-side mirror of the keyword ranking seam (S3): bucket sort over the
Words and Typo ranking rules (crates/milli/src/search/new/bucket_sort.rs:23-343,
graph_based_ranking_rule.rs, ranking_rule_graph/{words,typo}/mod.rs) on dense docid
sets in HBM."""
import ctypes as C

import numpy as np

from ._lib import RankNode, RankTerm, check, lib
from .device import np_ptr

NO_SLOT = 0xFFFFFFFF
TERMS_LAST, TERMS_ALL = 0, 1
MAX_TERMS = 10


def bucket_sort_words_typo(pool, terms, universe_slot, scratch_slot, strategy=TERMS_LAST, use_typo=True,
                           offset=0, limit=20):
    """terms: [(slot0|None, slot1|None, slot2|None, max_typo_cost)] in query order.
    Returns ([(docid, matching_words, typo_count, max_typo_count)], n_candidates)."""
    n = len(terms)
    arr = (RankTerm * max(n, 1))()
    for i, (s0, s1, s2, mc) in enumerate(terms):
        for j, s in enumerate((s0, s1, s2)):
            arr[i].level_slot[j] = NO_SLOT if s is None else int(s)
        arr[i].max_typo_cost = int(mc)
    ids = np.zeros(max(limit, 1), dtype=np.uint32)
    words = np.zeros(max(limit, 1), dtype=np.uint32)
    typos = np.zeros(max(limit, 1), dtype=np.uint32)
    maxt = np.zeros(max(limit, 1), dtype=np.uint32)
    out_n = C.c_uint32(0)
    cand = C.c_uint64(0)
    check(lib().msi_rank_words_typo(pool._h, arr, n, universe_slot, scratch_slot, strategy, 1 if use_typo else 0,
                                    offset, limit, np_ptr(ids), np_ptr(words), np_ptr(typos), np_ptr(maxt),
                                    C.byref(out_n), C.byref(cand)))
    k = out_n.value
    return [(int(ids[i]), int(words[i]), int(typos[i]), int(maxt[i])) for i in range(k)], int(cand.value)


def bucket_sort_query_graph(pool, nodes, n_terms, universe_slot, scratch_slot, strategy=TERMS_LAST, use_typo=True,
                            offset=0, limit=20):
    """nodes: [(first_term, last_term, slot0|None, slot1|None, slot2|None, max_typo_cost)] — the single
    terms plus the 2-gram / 3-gram nodes of the query graph (query_graph.rs:96-180)."""
    arr = (RankNode * max(len(nodes), 1))()
    for i, (a, b, s0, s1, s2, mc) in enumerate(nodes):
        arr[i].first_term, arr[i].last_term = int(a), int(b)
        for j, s in enumerate((s0, s1, s2)):
            arr[i].level_slot[j] = NO_SLOT if s is None else int(s)
        arr[i].max_typo_cost = int(mc)
    ids = np.zeros(max(limit, 1), dtype=np.uint32)
    words = np.zeros(max(limit, 1), dtype=np.uint32)
    typos = np.zeros(max(limit, 1), dtype=np.uint32)
    maxt = np.zeros(max(limit, 1), dtype=np.uint32)
    out_n = C.c_uint32(0)
    cand = C.c_uint64(0)
    check(lib().msi_rank_query_graph(pool._h, arr, len(nodes), n_terms, universe_slot, scratch_slot, strategy,
                                     1 if use_typo else 0, offset, limit, np_ptr(ids), np_ptr(words), np_ptr(typos),
                                     np_ptr(maxt), C.byref(out_n), C.byref(cand)))
    k = out_n.value
    return [(int(ids[i]), int(words[i]), int(typos[i]), int(maxt[i])) for i in range(k)], int(cand.value)
