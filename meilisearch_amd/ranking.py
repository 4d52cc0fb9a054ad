"""Binding + host-side mirror of the keyword ranking seam (S3): bucket sort over the
Words and Typo ranking rules (crates/milli/src/search/new/bucket_sort.rs:23-343,
graph_based_ranking_rule.rs, ranking_rule_graph/{words,typo}/mod.rs) on dense docid
sets in HBM."""
import ctypes as C

import numpy as np

from ._lib import (GeoRule, EXACT_WORD_FN, FID_COUNT_DOCIDS_FN, PAIR_DOCIDS_FN, PREFIX_DOCIDS_FN, PREFIX_KEY_DOCIDS_FN,
                   PREFIX_PAIR_DOCIDS_FN, SYNONYMS_FN, EXACT_PREFIX_FN, WORD_DOCIDS_FN, WORD_KEY_DOCIDS_FN, WORD_KEYS_FN,
                   IndexVtable, KeywordParams, LocatedTerm, QueryToken, RankBucket, ScoreDetail, SearchParams,
                   RankNode, RankQuery, RankTerm, check, lib)
from .device import np_ptr

NO_SLOT = 0xFFFFFFFF
TERMS_LAST, TERMS_ALL, TERMS_FREQUENCY = 0, 1, 2      # include/msi.h: MSI_TERMS_*


def strategy_of(tms):
    """"last" | "all" | "frequency" (the request's matchingStrategy) -> MSI_TERMS_*"""
    return {"last": TERMS_LAST, "all": TERMS_ALL, "frequency": TERMS_FREQUENCY}[tms]
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


def _node_array(nodes):
    arr = (RankNode * max(len(nodes), 1))()
    for i, (a, b, s0, s1, s2, mc) in enumerate(nodes):
        arr[i].first_term, arr[i].last_term = int(a), int(b)
        for j, s in enumerate((s0, s1, s2)):
            arr[i].level_slot[j] = NO_SLOT if s is None else int(s)
        arr[i].max_typo_cost = int(mc)
    return arr


def rank_buckets(pool, nodes, n_terms, universe_slot, scratch_slot, strategy=TERMS_LAST, use_typo=True):
    """Buckets of [Words, Typo] in order: [(matching_words, typo_count, max_typo_count, count)]."""
    out = (RankBucket * 256)()
    n = C.c_uint32(0)
    check(lib().msi_rank_buckets(pool._h, _node_array(nodes), len(nodes), n_terms, universe_slot, scratch_slot,
                                 strategy, 1 if use_typo else 0, out, 256, C.byref(n)))
    return [(out[i].matching_words, out[i].typo_count, out[i].max_typo_count, int(out[i].count))
            for i in range(min(n.value, 256))]


def rank_materialise(pool, nodes, n_terms, universe_slot, dst_slot, matching_words, typo_count,
                     strategy=TERMS_LAST, use_typo=True):
    """One bucket as a docid set in `dst_slot` (the universe handed to the next ranking rule)."""
    check(lib().msi_rank_materialise(pool._h, _node_array(nodes), len(nodes), n_terms, universe_slot, strategy,
                                     1 if use_typo else 0, matching_words, typo_count, dst_slot))


class RankBatch:
    """A prepared batch for msi_rank_query_graph_batch: queries = [(nodes, n_terms, universe_slot,
    first_of_4_scratch_slots)]; `run` can be called repeatedly (bench)."""

    def __init__(self, pool, queries):
        self.pool = pool
        self.n = len(queries)
        self._node_arrays = [_node_array(q[0]) for q in queries]
        self._q = (RankQuery * max(self.n, 1))()
        for i, (nodes, n_terms, uni, scratch) in enumerate(queries):
            self._q[i].nodes = C.cast(self._node_arrays[i], C.c_void_p)
            self._q[i].n_nodes = len(nodes)
            self._q[i].n_terms = n_terms
            self._q[i].universe_slot = uni
            self._q[i].scratch_slot = scratch

    def run(self, strategy=TERMS_LAST, use_typo=True, offset=0, limit=20):
        n, L = self.n, max(limit, 1)
        self.ids = np.zeros((n, L), dtype=np.uint32)
        self.words = np.zeros((n, L), dtype=np.uint32)
        self.typos = np.zeros((n, L), dtype=np.uint32)
        self.maxt = np.zeros((n, L), dtype=np.uint32)
        self.counts = np.zeros(n, dtype=np.uint32)
        self.cand = np.zeros(n, dtype=np.uint64)
        check(lib().msi_rank_query_graph_batch(self.pool._h, self._q, n, strategy, 1 if use_typo else 0, offset, limit,
                                               np_ptr(self.ids), np_ptr(self.words), np_ptr(self.typos),
                                               np_ptr(self.maxt), np_ptr(self.counts), np_ptr(self.cand)))
        return self

    def rows(self):
        """Results of the last run as [[(docid, matching_words, typo_count, max_typo_count)]], candidates."""
        return [[(int(self.ids[q, i]), int(self.words[q, i]), int(self.typos[q, i]), int(self.maxt[q, i]))
                 for i in range(int(self.counts[q]))] for q in range(self.n)], self.cand.tolist()


class IndexCallbacks:
    """Adapter from a Python index object to msi_index_vtable.  `index` provides
    word_docids_bytes(word: str, original: bool) -> bytes|None,
    pair_docids_bytes(prox: int, left: str, right: str) -> bytes|None and
    is_exact_word(word: str) -> bool — the LMDB gets of db_cache.rs on the Rust side."""

    def __init__(self, index):
        self.index = index
        self._keep = None   # the bytes handed out stay alive until the next callback

        def hand_out(data, out_bytes, out_n):
            if not data:
                out_n[0] = 0
                return 0
            self._keep = C.create_string_buffer(data, len(data))
            out_bytes[0] = C.cast(self._keep, C.POINTER(C.c_uint8))
            out_n[0] = len(data)
            return 0

        def word_docids(user, w, n, original, out_bytes, out_n):
            try:
                return hand_out(index.word_docids_bytes(bytes(w[:n]).decode("utf-8"), bool(original)), out_bytes, out_n)
            except Exception:
                return -1

        def pair_docids(user, prox, l, ln, r, rn, out_bytes, out_n):
            try:
                return hand_out(index.pair_docids_bytes(prox, bytes(l[:ln]).decode("utf-8"),
                                                        bytes(r[:rn]).decode("utf-8")), out_bytes, out_n)
            except Exception:
                return -1

        def is_exact(user, w, n):
            try:
                return 1 if index.is_exact_word(bytes(w[:n]).decode("utf-8")) else 0
            except Exception:
                return 0
        def word_fid(user, w, n, fid, out_bytes, out_n):
            try:
                return hand_out(index.word_fid_docids_bytes(bytes(w[:n]).decode("utf-8"), fid), out_bytes, out_n)
            except Exception:
                return -1

        def word_position(user, w, n, pos, out_bytes, out_n):
            try:
                return hand_out(index.word_position_docids_bytes(bytes(w[:n]).decode("utf-8"), pos), out_bytes, out_n)
            except Exception:
                return -1

        def keys(getter):
            def fn(user, w, n, out, cap, out_n):
                try:
                    vals = getter(bytes(w[:n]).decode("utf-8"))
                    out_n[0] = len(vals)
                    for i, v in enumerate(vals[:cap]):
                        out[i] = v
                    return 0
                except Exception:
                    return -1
            return fn

        def fid_count(user, fid, count, out_bytes, out_n):
            try:
                return hand_out(index.fid_word_count_docids_bytes(fid, count), out_bytes, out_n)
            except Exception:
                return -1
        self._fns = (WORD_DOCIDS_FN(word_docids), PAIR_DOCIDS_FN(pair_docids), EXACT_WORD_FN(is_exact))
        full = all(hasattr(index, m) for m in ("word_fid_docids_bytes", "word_position_docids_bytes", "word_fids",
                                                "word_positions", "fid_word_count_docids_bytes"))
        if full:
            self._fns += (WORD_KEY_DOCIDS_FN(word_fid), WORD_KEY_DOCIDS_FN(word_position),
                          WORD_KEYS_FN(keys(index.word_fids)), WORD_KEYS_FN(keys(index.word_positions)),
                          FID_COUNT_DOCIDS_FN(fid_count))
        if full and hasattr(index, "word_prefix_docids_values"):
            def push_all(values, push, sink):
                n = 0
                for data in values or ():
                    buf = C.create_string_buffer(data, len(data))
                    if push(sink, C.cast(buf, C.POINTER(C.c_uint8)), len(data)) < 0:
                        return -1
                    n += 1
                return n

            def pfx_docids(user, w, n, original, push, sink):
                try:
                    return push_all(index.word_prefix_docids_values(bytes(w[:n]).decode("utf-8"), bool(original)), push, sink)
                except Exception:
                    return -1

            def pfx_fid(user, w, n, fid, push, sink):
                try:
                    return push_all(index.word_prefix_fid_docids_values(bytes(w[:n]).decode("utf-8"), fid), push, sink)
                except Exception:
                    return -1

            def pfx_pos(user, w, n, pos, push, sink):
                try:
                    return push_all(index.word_prefix_position_docids_values(bytes(w[:n]).decode("utf-8"), pos), push, sink)
                except Exception:
                    return -1

            def pfx_pair(user, prox, l, ln, r, rn, push, sink):
                try:
                    return push_all(index.word_prefix_pair_values(prox, bytes(l[:ln]).decode("utf-8"),
                                                                  bytes(r[:rn]).decode("utf-8")), push, sink)
                except Exception:
                    return -1
            def synonyms(user, words, n, push, sink):
                try:
                    toks = C.cast(words, C.POINTER(QueryToken))
                    key = [C.string_at(toks[i].word, toks[i].len).decode("utf-8") for i in range(n)]
                    for syn in index.get_synonyms(key):
                        arr = (QueryToken * max(len(syn), 1))()
                        keep = []
                        for i, w in enumerate(syn):
                            b = w.encode("utf-8")
                            buf = C.create_string_buffer(b, len(b))
                            keep.append(buf)
                            arr[i].word = C.cast(buf, C.c_void_p)
                            arr[i].len = len(b)
                        if push(sink, C.cast(arr, C.c_void_p), len(syn)) < 0:
                            return -1
                    return 0
                except Exception:
                    return -1
            self._fns += (PREFIX_DOCIDS_FN(pfx_docids), PREFIX_KEY_DOCIDS_FN(pfx_fid), PREFIX_KEY_DOCIDS_FN(pfx_pos),
                          PREFIX_PAIR_DOCIDS_FN(pfx_pair), WORD_KEYS_FN(keys(index.get_word_prefix_fids)),
                          WORD_KEYS_FN(keys(index.get_word_prefix_positions)), SYNONYMS_FN(synonyms))
            if hasattr(index, "exact_words_with_prefix"):
                def exact_prefix(user, w, n, push, sink):
                    try:
                        for word in index.exact_words_with_prefix(bytes(w[:n]).decode("utf-8")):
                            b = word.encode("utf-8")
                            buf = C.create_string_buffer(b, len(b))
                            tok = QueryToken(C.cast(buf, C.c_void_p), len(b), 0)
                            if push(sink, C.cast(C.pointer(tok), C.c_void_p), 1) < 0:
                                return -1
                        return 0
                    except Exception:
                        return -1
                self._fns += (EXACT_PREFIX_FN(exact_prefix),)
        self.vtable = IndexVtable(None, *self._fns)


def keyword_search(gdict, pool, callbacks, words, last_is_prefix=True, strategy=TERMS_LAST, use_typo=True,
                   offset=0, limit=20, authorize_typos=True, min_one=5, min_two=9, universe_cbo=None):
    """msi_keyword_search: the keyword leg for the rules [Words, Typo] on the product path.
    words: normalised single-word tokens in query order."""
    n = len(words)
    toks = (QueryToken * max(n, 1))()
    keep = []
    for i, w in enumerate(words):
        b = w.encode("utf-8")
        buf = C.create_string_buffer(b, len(b))
        keep.append(buf)
        toks[i].word = C.cast(buf, C.c_void_p)
        toks[i].len = len(b)
        toks[i].is_prefix = 1 if (last_is_prefix and i == n - 1) else 0
    params = KeywordParams(1 if authorize_typos else 0, min_one, min_two, strategy, 1 if use_typo else 0, offset, limit)
    ids = np.zeros(max(limit, 1), dtype=np.uint32)
    mw = np.zeros(max(limit, 1), dtype=np.uint32)
    tc = np.zeros(max(limit, 1), dtype=np.uint32)
    mt = np.zeros(max(limit, 1), dtype=np.uint32)
    out_n = C.c_uint32(0)
    cand = C.c_uint64(0)
    ub = None
    if universe_cbo is not None:
        ub = np.frombuffer(universe_cbo, dtype=np.uint8)
    check(lib().msi_keyword_search(gdict._h, pool._h, C.byref(callbacks.vtable), toks, n, C.byref(params),
                                   np_ptr(ub) if ub is not None else None, 0 if ub is None else ub.size,
                                   np_ptr(ids), np_ptr(mw), np_ptr(tc), np_ptr(mt), C.byref(out_n), C.byref(cand)))
    k = out_n.value
    return [(int(ids[i]), int(mw[i]), int(tc[i]), int(mt[i])) for i in range(k)], int(cand.value)


CRITERIA = {"words": 0, "typo": 1, "proximity": 2, "attribute": 3, "attributeRank": 4, "wordPosition": 5,
            "exactness": 6, "sort": 7, "orderBy": 8, "geoSort": 9}
SCORE_KINDS = ["Words", "Typo", "Proximity", "Fid", "Position", "ExactAttribute", "ExactWords", "Skipped", "Sort", "GeoSort"]
NO_ORDER_KEY = 0xFFFFFFFF


def expand_sort_criteria(criteria, sort=None):
    """What the shim does with Criterion::Sort / Asc / Desc (search/new/mod.rs:366-376,640-720): `sort` of the list
    becomes one rule per field of the request's sort list, `asc:f` / `desc:f` one rule, a field is sorted only once.
    A sort entry whose field is ("_geoPoint", lat, lng) becomes a "geoSort" entry (mod.rs:690-712; never deduplicated).
    -> (criteria with "orderBy" / "geoSort" entries, [(field, ascending)] of the orderBy entries in order); the geo
    entries, in order, are geo_sort_entries(sort)."""
    out, order, fields, sort_done = [], [], set(), False
    for c in criteria:
        if c == "sort":
            if not sort_done:
                sort_done = True
                for f, d in sort or ():
                    if is_geo_point(f):
                        out.append("geoSort")
                    elif f not in fields:
                        fields.add(f)
                        out.append("orderBy")
                        order.append((f, d == "asc"))
        elif c.startswith(("asc:", "desc:")):
            d, f = c.split(":", 1)
            if f not in fields:
                fields.add(f)
                out.append("orderBy")
                order.append((f, d == "asc"))
        else:
            out.append(c)
    return out, order


def is_geo_point(field):
    return isinstance(field, (tuple, list)) and len(field) == 3 and field[0] == "_geoPoint"


def geo_sort_entries(criteria, sort):
    """[(lat, lng, ascending)] of the "geoSort" entries expand_sort_criteria produces, in order."""
    return [(float(f[1]), float(f[2]), d == "asc") for f, d in (sort or ()) if is_geo_point(f)] if "sort" in criteria else []


MAX_SCORE_DETAILS = 16


def keyword_search_ranked(gdict, pool, callbacks, terms, criteria, strategy=TERMS_LAST, offset=0, limit=20,
                          detailed=False, searchable_fids=(), searchable_weights=(), max_weight=None,
                          authorize_typos=True, min_one=5, min_two=9, universe_cbo=None, time_budget_us=0,
                          stop_after=None, return_degraded=False, score_threshold=None, order_keys=(), distinct_values=None,
                          geo_rules=(), geo_max_bucket_size=0, geo_distance_error_margin=1.0, geo_strategy=("dynamic", 1000),
                          exhaustive=False,
                          max_total_hits=None, index_view=0, _entry=None):
    """msi_keyword_search_ranked: bucket sort over every graph-based ranking rule of `criteria`.
    terms: [(words, is_phrase, position_start, position_end, is_prefix)] — the located query terms
    (words: [str | None], None = a stop word inside a phrase; an optional 6th element True marks a negative term).
    order_keys: the DocKeys of the "orderBy" entries of `criteria`, in order (expand_sort_criteria).
    distinct_values: the DocValues of the distinct field (None: no distinct).
    geo_rules: [(GeoPoints, lat, lng, ascending)] of the "geoSort" entries of `criteria`, in order.
    -> ([(docid, [(kind name, a, b)])], candidates)."""
    n = len(terms)
    lt = (LocatedTerm * max(n, 1))()
    keep = []
    for i, term in enumerate(terms):
        words, is_phrase, ps, pe, is_prefix = term[:5]
        negative = len(term) > 5 and term[5]
        toks = (QueryToken * len(words))()
        for k, w in enumerate(words):
            b = (w or "").encode("utf-8")
            buf = C.create_string_buffer(b, len(b))
            keep.append(buf)
            toks[k].word = C.cast(buf, C.c_void_p)
            toks[k].len = len(b)
            toks[k].is_prefix = 1 if (is_prefix and not is_phrase) else 0
        keep.append(toks)
        lt[i].words = C.cast(toks, C.c_void_p)
        lt[i].n_words = len(words)
        lt[i].is_phrase = (1 if is_phrase else 0) | (2 if negative else 0)
        lt[i].position_start, lt[i].position_end = ps, pe
    crit = np.array([CRITERIA[c] for c in criteria], dtype=np.int32)
    fids = np.array(list(searchable_fids), dtype=np.uint16)
    wts = np.array(list(searchable_weights), dtype=np.uint16)
    prm = SearchParams(1 if authorize_typos else 0, min_one, min_two, strategy, np_ptr(crit) if crit.size else None,
                       crit.size, np_ptr(fids) if fids.size else None, np_ptr(wts) if wts.size else None, fids.size,
                       -1 if max_weight is None else int(max_weight), offset, limit, 1 if detailed else 0,
                       int(time_budget_us), -1 if stop_after is None else int(stop_after),
                       0 if score_threshold is None else 1, 0.0 if score_threshold is None else float(score_threshold))
    if order_keys:
        okeys = (C.c_void_p * len(order_keys))(*[k._h for k in order_keys])
        prm.order_keys, prm.n_order_keys = C.cast(okeys, C.c_void_p), len(order_keys)
    if distinct_values is not None:
        prm.distinct_values = distinct_values._h
    if geo_rules:
        garr = (GeoRule * len(geo_rules))()
        for i, (pts, lat, lng, asc) in enumerate(geo_rules):
            garr[i].points, garr[i].lat, garr[i].lng, garr[i].ascending = pts._h, float(lat), float(lng), 1 if asc else 0
        prm.geo_rules, prm.n_geo_rules = C.cast(garr, C.c_void_p), len(geo_rules)
    prm.geo_max_bucket_size, prm.geo_distance_error_margin = int(geo_max_bucket_size), float(geo_distance_error_margin)
    prm.exhaustive_number_hits, prm.max_total_hits = (1 if exhaustive else 0), int(max_total_hits or 0)
    # GeoSortStrategy (documents/geo_sort.rs:32-63): ("dynamic" | "iterative" | "rtree", cache size)
    prm.geo_strategy = {"dynamic": 0, "iterative": 1, "rtree": 2}[geo_strategy[0]]
    prm.geo_cache_size = int(geo_strategy[1])
    prm.index_view = int(index_view)      # attributesToSearchOn: the view of the index the callbacks answer for (include/msi.h)
    L = max(limit, 1)
    ids = np.zeros(L, dtype=np.uint32)
    sc = (ScoreDetail * (L * MAX_SCORE_DETAILS))()
    nsc = np.zeros(L, dtype=np.uint32)
    out_n, cand, degraded = C.c_uint32(0), C.c_uint64(0), C.c_int32(0)
    ub = np.frombuffer(universe_cbo, dtype=np.uint8) if universe_cbo is not None else None
    entry = _entry or lib().msi_keyword_search_ranked   # _entry: the CPU test tier's host-logic build (tests/hostlogic)
    check(entry(gdict._h, pool._h, C.byref(callbacks.vtable), lt, n, C.byref(prm),
                                          np_ptr(ub) if ub is not None else None, 0 if ub is None else ub.size,
                                          np_ptr(ids), C.cast(sc, C.c_void_p), np_ptr(nsc), C.byref(out_n),
                                          C.byref(cand), C.byref(degraded)))
    hits = []
    for i in range(out_n.value):
        det = [(SCORE_KINDS[sc[i * MAX_SCORE_DETAILS + k].kind], int(sc[i * MAX_SCORE_DETAILS + k].a),
                int(sc[i * MAX_SCORE_DETAILS + k].b)) for k in range(int(nsc[i]))]
        hits.append((int(ids[i]), det))
    if return_degraded:
        return hits, int(cand.value), bool(degraded.value)
    return hits, int(cand.value)


def search_last_stats():
    """Counters of the last keyword_search_ranked on this thread."""
    v = (C.c_uint64 * 10)()
    check(lib().msi_search_last_stats(v))
    names = ["launches", "syncs", "decode_batches", "callbacks", "posting_bytes", "paths", "buckets", "callback_us",
             "device_wait_us", "total_us"]
    return dict(zip(names, [int(x) for x in v]))


def score_details_global_score(details):
    """ScoreDetails::global_score of one hit's [(kind name, a, b)]."""
    arr = (ScoreDetail * max(len(details), 1))()
    for i, (k, a, b) in enumerate(details):
        arr[i].kind, arr[i].a, arr[i].b = SCORE_KINDS.index(k), a, b
    return float(lib().msi_score_details_global_score(C.cast(arr, C.c_void_p), len(details)))
