"""CPU tier for the HOST logic of msi_keyword_search_ranked (query graph, rule graphs, path enumeration, bucket sort,
caches, deadline, threshold): meilisearch_amd/csrc/msi_search.hip is compiled together with a plain-C++ test double
of the device-set pool and the device dictionary (tests/hostlogic/mock_device.cpp — NOT part of the product, never
loaded by meilisearch_amd) and the reference's snapshot searches are replayed through it.  The GPU tier
(tests/test_search_gpu.py) replays the same cases through the real kernels."""
import ctypes as C
import json
import os
import subprocess

import numpy as np
import pytest

from meilisearch_amd import _lib
from meilisearch_amd import ranking as R
from oracle import oracle as O
from tests.toy_milli import ToyMilli, query_terms

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BUILD = os.path.join(ROOT, "tests", "hostlogic", "_build")
SO = os.path.join(BUILD, "libmsi_hostlogic_test.so")
SOURCES = [os.path.join(ROOT, "tests", "hostlogic", "mock_device.cpp"),
           os.path.join(ROOT, "meilisearch_amd", "csrc", "msi_search.hip"),
           os.path.join(ROOT, "meilisearch_amd", "csrc", "msi_common.h"), os.path.join(ROOT, "include", "msi.h")]
FIX = json.load(open(os.path.join(ROOT, "tests", "golden", "ranking_snapshots.json")))

LOOKUP_FN = C.CFUNCTYPE(C.c_int32, C.POINTER(C.c_uint8), C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32,
                        C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.POINTER(C.c_uint32))


@pytest.fixture(scope="module")
def hostlib():
    return load_hostlib()


def load_hostlib():
    _lib.lib()   # the product library first (one HIP runtime; msi_set_error / parsers come from it)
    if not os.path.exists(SO) or any(os.path.getmtime(s) > os.path.getmtime(SO) for s in SOURCES + [_lib.lib_path()]):
        os.makedirs(BUILD, exist_ok=True)
        subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O1", "-std=c++17", "-fPIC", "-shared",
                               "-DMSI_SEARCH_DIRECT_ONLY",   # the test double has no kernels: one call per set operation
                               "-I" + os.path.join(ROOT, "meilisearch_amd", "csrc"), "-I" + os.path.join(ROOT, "include"),
                               SOURCES[0], SOURCES[1], "-L" + os.path.join(ROOT, "meilisearch_amd"), "-lmsi",
                               "-Wl,-rpath," + os.path.join(ROOT, "meilisearch_amd"), "-o", SO])
    L = C.CDLL(SO)
    L.msi_keyword_search_ranked.restype, L.msi_keyword_search_ranked.argtypes = _lib.PROTOTYPES["msi_keyword_search_ranked"]
    L.mock_bits_create.restype, L.mock_bits_create.argtypes = C.c_void_p, [C.c_uint64, C.c_uint32]
    L.mock_bits_destroy.restype, L.mock_bits_destroy.argtypes = None, [C.c_void_p]
    L.mock_dict_create.restype, L.mock_dict_create.argtypes = C.c_void_p, [C.c_void_p, C.c_void_p, C.c_uint32, LOOKUP_FN]
    L.mock_dict_destroy.restype, L.mock_dict_destroy.argtypes = None, [C.c_void_p]
    L.mock_doc_keys_create.restype, L.mock_doc_keys_create.argtypes = C.c_void_p, [C.c_void_p, C.c_uint64]
    L.mock_doc_keys_destroy.restype, L.mock_doc_keys_destroy.argtypes = None, [C.c_void_p]
    L.mock_doc_values_create.restype, L.mock_doc_values_create.argtypes = C.c_void_p, [C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint32]
    L.mock_doc_values_destroy.restype, L.mock_doc_values_destroy.argtypes = None, [C.c_void_p]
    L.mock_geo_points_create.restype, L.mock_geo_points_create.argtypes = C.c_void_p, [C.c_void_p, C.c_uint64]
    L.mock_geo_points_destroy.restype, L.mock_geo_points_destroy.argtypes = None, [C.c_void_p]
    return L


def make_harness(L, index, **kw):
    return getattr(L, "harness_cls", MockHarness)(L, index, **kw)


class Handle:
    def __init__(self, h):
        self._h = C.c_void_p(h)


class MockHarness:
    """The same call the GPU harness makes, against the host-logic build."""

    def __init__(self, L, index, n_slots=512):
        self.L, self.index = L, index
        self.entry = L.msi_keyword_search_ranked
        self.dict = self.dict_create(index)
        self.pool = self.pool_create(max(index.n_docs, 1), n_slots)
        self.cb = R.IndexCallbacks(index)

    # the device objects: test doubles here; the product's own on the device (tests/test_zzz_distinct_gpu.py:
    # DeviceHarness overrides these hooks — and that file also runs on the CPU emulation of the HIP runtime,
    # tests/test_kernels_emulated_cpu.py)
    def dict_create(self, index):
        L = self.L
        dic = O.Dictionary(index.words)
        words = index.words

        def lookup(w, n, max_typos, is_prefix, cap1, cap2, one, n1, two, n2):
            try:
                a, b = O.typo_lookup(dic, bytes(w[:n]), max_typos, bool(is_prefix), cap1, cap2)
                for i, v in enumerate(a):
                    one[i] = int(v)
                for i, v in enumerate(b):
                    two[i] = int(v)
                n1[0], n2[0] = len(a), len(b)
                return 0
            except Exception:   # noqa: BLE001
                return -8
        self._cb = LOOKUP_FN(lookup)
        bs = [w.encode() for w in words]
        self._concat = np.frombuffer(b"".join(bs) or b"\0", dtype=np.uint8).copy()
        self._offs = np.zeros(len(bs) + 1, dtype=np.uint32)
        if bs:
            np.cumsum([len(b) for b in bs], out=self._offs[1:])
        return Handle(L.mock_dict_create(self._concat.ctypes.data, self._offs.ctypes.data, len(bs), self._cb))

    def dict_destroy(self, d):
        self.L.mock_dict_destroy(d._h)

    def pool_create(self, n_docs, n_slots):
        return Handle(self.L.mock_bits_create(n_docs, n_slots))

    def pool_destroy(self, pool):
        self.L.mock_bits_destroy(pool._h)

    def keys_create(self, arr):
        return Handle(self.L.mock_doc_keys_create(arr.ctypes.data_as(C.c_void_p), arr.size))

    def keys_destroy(self, h):
        self.L.mock_doc_keys_destroy(h._h)

    def values_create(self, per_doc, n_values):
        offsets = np.zeros(len(per_doc) + 1, dtype=np.uint64)
        np.cumsum([len(v) for v in per_doc], out=offsets[1:])
        flat = np.array([x for v in per_doc for x in v] or [0], dtype=np.uint32)
        return Handle(self.L.mock_doc_values_create(offsets.ctypes.data_as(C.c_void_p), flat.ctypes.data_as(C.c_void_p),
                                                    len(per_doc), n_values))

    def values_destroy(self, h):
        self.L.mock_doc_values_destroy(h._h)

    def points_create(self, lat_lng):
        return Handle(self.L.mock_geo_points_create(lat_lng.ctypes.data_as(C.c_void_p), lat_lng.shape[0]))

    def points_destroy(self, h):
        self.L.mock_geo_points_destroy(h._h)

    def search(self, query, tms="last", criteria=None, offset=0, limit=20, detailed=False, stop_after=None, sort=None,
               distinct=None, extra_terms=(), **kw):
        """sort: the request's [(field, "asc" | "desc")]; Sort details come back as the oracle writes them:
        ("Sort", field, ascending, ("Number", x) | ("String", s) | ("Null",)).  distinct: the distinct field."""
        ix = self.index
        dv = self.values_create(*ix.distinct_values(distinct)) if distinct else None
        geo = R.geo_sort_entries(criteria if criteria is not None else ix.criteria, sort)
        points = None
        if geo:
            lat_lng = np.full((max(ix.n_docs, 1), 2), np.nan)
            for d, pt in ix.geo_points.items():
                lat_lng[d] = pt
            points = self.points_create(lat_lng)
        crit, order = R.expand_sort_criteria(criteria if criteria is not None else ix.criteria, sort)
        handles, tables = [], []
        for field, asc in order:
            keys, values = ix.order_keys(field, asc)
            arr = np.array(keys, dtype=np.uint32)
            handles.append(self.keys_create(arr))
            tables.append((field, asc, values))
        try:
            out = R.keyword_search_ranked(
                self.dict, self.pool, self.cb, query_terms(query, stop_words=ix.stop_words) + list(extra_terms), crit,
                strategy=R.strategy_of(tms), offset=offset, limit=limit, detailed=detailed,
                searchable_fids=ix.searchable_fids, searchable_weights=[ix.weights[f] for f in ix.searchable_fids],
                max_weight=ix.max_weight, authorize_typos=ix.authorize_typos, min_one=ix.min_one, min_two=ix.min_two,
                stop_after=stop_after, order_keys=handles, distinct_values=dv, _entry=self.entry,
                geo_rules=[(points, lat, lng, asc) for lat, lng, asc in geo], **kw)
        finally:
            for h in handles:
                self.keys_destroy(h)
            if dv is not None:
                self.values_destroy(dv)
            if points is not None:
                self.points_destroy(points)

        def detail(s):
            if s[0] == "GeoSort":      # (GeoSort, rule index, docid of the bucket's first document): the value is the shim's
                lat, lng, asc = geo[s[1]]
                return ("GeoSort", (lat, lng), asc, None if s[2] == R.NO_ORDER_KEY else ix.geo_points[s[2]])
            return sort_detail(s, tables)
        return ([(d, [detail(s) for s in sc]) for d, sc in out[0]],) + tuple(out[1:])

    def close(self):
        self.pool_destroy(self.pool)
        self.dict_destroy(self.dict)


def sort_detail(s, tables):
    """(Sort, rule index, order key) -> the oracle's ("Sort", field, ascending, value): the rank -> value table is
    the shim's."""
    if s[0] != "Sort":
        return s
    field, asc, values = tables[s[1]]
    if s[2] == R.NO_ORDER_KEY:
        return ("Sort", field, asc, ("Null",))
    kind, v = values[s[2]]
    return ("Sort", field, asc, ("Number", v) if kind == "n" else ("String", v))


def debug_score(s):
    k = s[0]
    if k == "Sort":
        v = s[3]
        val = "Null" if v[0] == "Null" else (f"Number({float(v[1])!r})" if v[0] == "Number" else f'String("{v[1]}")')
        return f'Sort(Sort{{field_name:"{s[1]}",ascending:{str(s[2]).lower()},redacted:false,value:{val},}},)'
    if k == "GeoSort":     # ("GeoSort", target point, ascending, point of the bucket's first document | None)
        val = "None" if s[3] is None else f"Some([{float(s[3][0])!r},{float(s[3][1])!r},],)"
        return (f"GeoSort(GeoSort{{target_point:[{float(s[1][0])!r},{float(s[1][1])!r},],ascending:{str(s[2]).lower()},"
                f"value:{val},}},)")
    if k == "Words":
        return f"Words(Words{{matching_words:{s[1]},max_matching_words:{s[2]},}},)"
    if k == "Typo":
        return f"Typo(Typo{{typo_count:{s[1]},max_typo_count:{s[2]},}},)"
    if k == "ExactWords":
        return f"ExactWords(ExactWords{{matching_words:{s[1]},max_matching_words:{s[2]},}},)"
    if k == "ExactAttribute":
        return "ExactAttribute(%s,)" % {3: "ExactMatch", 2: "MatchesStart", 1: "NoExactMatch"}[s[1]]
    if k == "Skipped":
        return "Skipped"
    return f"{k}(Rank{{rank:{s[1]},max_rank:{s[2]},}},)"


def build_index(cfg, **extra):
    return ToyMilli(cfg["docs"], searchable=cfg.get("searchable"), exact_attributes=cfg.get("exact_attributes", ()),
                    exact_words=cfg.get("exact_words", ()), criteria=cfg.get("criteria"),
                    min_one=cfg.get("min_one", 5), min_two=cfg.get("min_two", 9),
                    authorize_typos=cfg.get("authorize_typos", True), synonyms=cfg.get("synonyms"),
                    stop_words=cfg.get("stop_words", ()), distinct=cfg.get("distinct"), **extra)


@pytest.mark.parametrize("fused,per_wait", [("1", "1"), ("0", "1"), ("1", "2"), ("1", "4")],
                         ids=["level-at-once", "path-by-path", "2-levels-per-wait", "4-levels-per-wait"])
def test_reference_snapshots_through_the_host_logic(hostlib, monkeypatch, fused, per_wait):
    """All 94 reference searches, with the whole-level evaluation, with the path-by-path fallback, and with several
    cost levels evaluated ahead behind one completion wait."""
    monkeypatch.setenv("MSI_SEARCH_FUSED_LEVELS", fused)
    monkeypatch.setenv("MSI_SEARCH_LEVELS_PER_WAIT", per_wait)
    harnesses, n = {}, 0
    cases = [c for c in FIX["cases"] if not c.get("needs")]
    for case in cases:
        if case["index"] not in harnesses:
            harnesses[case["index"]] = make_harness(hostlib, build_index(FIX["indexes"][case["index"]]))
        h = harnesses[case["index"]]
        hits, _ = h.search(case["query"], tms=case["tms"], offset=case["offset"], limit=case["limit"],
                           detailed=case["detailed"], stop_after=case.get("stop_after"), sort=case.get("sort"),
                           distinct=case.get("distinct") or h.index.distinct_field)
        ids = [d for d, _ in hits]
        if case["ids"] is not None:
            assert ids == case["ids"], case["src"]
        if case.get("scores"):
            assert "[" + "".join("[" + "".join(debug_score(s) + "," for s in sc) + "]," for _, sc in hits) + "]" == case["scores"]
        if case.get("ids_scores"):
            got = "[" + "".join(f"({d},[" + "".join(debug_score(s) + "," for s in sc) + "],)," for d, sc in hits) + "]"
            assert got == case["ids_scores"], case["src"]
        if case.get("global_scores"):
            assert [f"{R.score_details_global_score(sc):.4f}" for _, sc in hits] == case["global_scores"]
        n += 1
    assert n == len(cases) >= 108   # 94 keyword searches + the 5 of sort.rs + the 9 of distinct.rs
    for h in harnesses.values():
        h.close()


@pytest.mark.parametrize("per_wait", ["1", "3"], ids=["one-level-per-wait", "3-levels-per-wait"])
def test_host_logic_matches_oracle_on_random_corpora(hostlib, monkeypatch, per_wait, rulesets=slice(0, 4), thresholds=(100, 3)):
    monkeypatch.setenv("MSI_SEARCH_LEVELS_PER_WAIT", per_wait)
    from oracle import ranking_oracle as RO
    import tests.test_search_gpu as G
    docs = G.random_corpus(31, 200)
    for criteria in G.RULESETS[rulesets]:
        for pt in thresholds:
            index = ToyMilli(docs, searchable=["title", "body"], criteria=criteria, prefix_threshold=pt)
            dic = O.Dictionary(index.words)

            def lookup(word, max_typos, is_prefix):
                one, two = O.typo_lookup(dic, word, max_typos, is_prefix)
                return [index.words[i] for i in one], [index.words[i] for i in two]
            h = make_harness(hostlib, index)
            for q in G.QUERIES:
                for tms in ("last", "all", "frequency"):
                    want_ids, want_sc, want_cand = RO.search(RO.Ctx(index, lookup), q, tms=tms, criteria=criteria, length=25,
                                                             detailed=True)
                    hits, cand = h.search(q, tms=tms, criteria=criteria, limit=25, detailed=True)
                    assert [d for d, _ in hits] == want_ids, (criteria, q, tms)
                    assert [[tuple(s) for s in sc] for _, sc in hits] == [[G.oracle_score(s) for s in sc] for sc in want_sc]
                    assert cand == len(want_cand)
            h.close()


@pytest.mark.parametrize("per_wait", ["1", "4"], ids=["one-level-per-wait", "4-levels-per-wait"])
def test_long_queries_match_the_oracle(hostlib, monkeypatch, per_wait):
    """6 / 8 / 10-term queries (n-gram nodes, many paths per level: both evaluation modes get used)."""
    monkeypatch.setenv("MSI_SEARCH_LEVELS_PER_WAIT", per_wait)
    import random
    from oracle import ranking_oracle as RO
    import tests.test_search_gpu as G
    rng = random.Random(3)
    index = ToyMilli(G.random_corpus(41, 400), searchable=["title", "body"])
    dic = O.Dictionary(index.words)

    def lookup(word, max_typos, is_prefix):
        one, two = O.typo_lookup(dic, word, max_typos, is_prefix)
        return [index.words[i] for i in one], [index.words[i] for i in two]
    h = make_harness(hostlib, index, n_slots=2048)
    for n in (6, 8, 10):
        for _ in range(4):
            q = " ".join(rng.choice(G.VOCAB) for _ in range(n))
            for tms in ("last", "all", "frequency"):
                want_ids, want_sc, want_cand = RO.search(RO.Ctx(index, lookup), q, tms=tms, length=20, detailed=True)
                hits, cand = h.search(q, tms=tms, limit=20, detailed=True)
                assert [d for d, _ in hits] == want_ids, (q, tms)
                assert [[tuple(s) for s in sc] for _, sc in hits] == [[G.oracle_score(s) for s in sc] for sc in want_sc]
                assert cand == len(want_cand)
    h.close()


def test_differential_fuzz_smoke(hostlib):
    """A short run of tools/fuzz_ranked_hostlogic.py (the long runs are manual: 79 k cases without a mismatch)."""
    env = dict(os.environ, PYTHONPATH=ROOT)
    out = subprocess.run([os.sys.executable, os.path.join(ROOT, "tools", "fuzz_ranked_hostlogic.py"), "900", "8"],
                         capture_output=True, text=True, env=env, timeout=300)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    assert " bad 0" in out.stdout.strip().splitlines()[-1], out.stdout[-2000:]


def sortable_corpus(seed, n_docs):
    """random_corpus plus facet fields: a float with ties and gaps, a string, a mixed number / string field and a
    multi-valued one (a document is placed at the first of its values the rule's iteration meets)."""
    import random
    import tests.test_search_gpu as G
    rng = random.Random(seed * 7 + 1)
    docs = G.random_corpus(seed, n_docs)
    for d in docs:
        if rng.random() < 0.85:
            d["price"] = rng.choice([1, 2, 2.5, 3, 10, 10, 99.5, 1000])
        if rng.random() < 0.7:
            d["color"] = rng.choice(["red", "green", "blue", "Blue", "ultra violet"])
        if rng.random() < 0.6:
            d["mixed"] = rng.choice([0, 7, "seven", "zero", 3.5])
        if rng.random() < 0.5:
            d["sizes"] = [rng.choice([36, 38, 40, 42, "xl"]) for _ in range(rng.randint(1, 3))]
    return docs


SORT_SETUPS = [
    (["words", "typo", "proximity", "attributeRank", "sort", "wordPosition", "exactness"], [("price", "asc")]),
    (["sort", "words", "typo"], [("color", "desc"), ("price", "asc")]),
    (["words", "sort", "proximity"], [("sizes", "desc")]),
    (["words", "sort"], [("sizes", "asc"), ("mixed", "asc")]),
    (["words", "typo", "desc:price", "exactness"], None),
    (["asc:mixed", "words", "sort"], [("mixed", "desc"), ("color", "asc")]),   # a field is sorted only once
    (["words", "typo"], [("price", "asc")]),                                    # no `sort` criterion: the list is ignored
]


@pytest.mark.parametrize("per_wait", ["1", "3"], ids=["one-level-per-wait", "3-levels-per-wait"])
def test_sort_rules_match_the_oracle(hostlib, monkeypatch, per_wait):
    """Sort / Asc / Desc as order-key rules (msi_bits_order_next), between graph-based rules and on placeholder
    searches: hits, score details (value per bucket, Null last) and candidates against the oracle."""
    from oracle import ranking_oracle as RO
    import tests.test_search_gpu as G
    monkeypatch.setenv("MSI_SEARCH_LEVELS_PER_WAIT", per_wait)
    index = ToyMilli(sortable_corpus(5, 250), searchable=["title", "body"])
    dic = O.Dictionary(index.words)

    def lookup(word, max_typos, is_prefix):
        one, two = O.typo_lookup(dic, word, max_typos, is_prefix)
        return [index.words[i] for i in one], [index.words[i] for i in two]
    h = make_harness(hostlib, index)
    n_sorted = 0
    for criteria, sort in SORT_SETUPS:
        for q in ["", "the", "quick fox", "sun fl", "\"lazy dog\"", "brwn fox jumps", "winter holi"]:
            for detailed, offset, limit in ((True, 0, 30), (False, 0, 12), (True, 17, 9)):
                want_ids, want_sc, want_cand = RO.search(RO.Ctx(index, lookup), q, tms="last", criteria=criteria,
                                                         offset=offset, length=limit, detailed=detailed, sort=sort)
                hits, cand = h.search(q, criteria=criteria, offset=offset, limit=limit, detailed=detailed, sort=sort)
                assert [d for d, _ in hits] == want_ids, (criteria, sort, q, detailed, offset)
                assert [[tuple(s) for s in sc] for _, sc in hits] == [[G.oracle_score(s) for s in sc] for sc in want_sc]
                assert cand == len(want_cand)
                n_sorted += any(s[0] == "Sort" for _, sc in hits for s in sc)
    assert n_sorted > 60
    h.close()


DISTINCT_SETUPS = [
    (["words", "typo", "proximity", "attributeRank", "wordPosition", "exactness"], None),
    (["words", "typo", "proximity", "attributeRank", "sort", "wordPosition", "exactness"], [("price", "asc")]),
    (["sort", "words", "typo"], [("color", "desc"), ("price", "asc")]),
    (["words", "exactness"], None),
    ([], None),                                                                  # no ranking rule at all
]


@pytest.mark.parametrize("per_wait", ["1", "3"], ids=["one-level-per-wait", "3-levels-per-wait"])
def test_distinct_matches_the_oracle(hostlib, monkeypatch, per_wait, fields=("color", "sizes", "mixed", "price"),
                                     setups=DISTINCT_SETUPS):
    """`distinct` (search/new/distinct.rs; bucket_sort.rs:61-92,399-415) over single-valued, multi-valued and mixed
    number / string fields, under every kind of rule and on rule-less searches: hits, score details and
    all_candidates against the oracle; the exclusions reach every universe of the rule stack."""
    from oracle import ranking_oracle as RO
    import tests.test_search_gpu as G
    monkeypatch.setenv("MSI_SEARCH_LEVELS_PER_WAIT", per_wait)
    index = ToyMilli(sortable_corpus(9, 220), searchable=["title", "body"])
    dic = O.Dictionary(index.words)

    def lookup(word, max_typos, is_prefix):
        one, two = O.typo_lookup(dic, word, max_typos, is_prefix)
        return [index.words[i] for i in one], [index.words[i] for i in two]
    h = make_harness(hostlib, index)
    n = 0
    for field in fields:
        for criteria, sort in setups:
            for q in ["", "the", "quick fox", "sun fl", "\"lazy dog\"", "brwn fox jumps"]:
                for detailed, offset, limit, threshold, exhaustive in (
                        (True, 0, 30, None, False), (False, 0, 7, None, False), (True, 5, 9, None, True), (True, 0, 20, 0.6, False),
                        (True, 0, 5, 0.6, True), (True, 500, 5, None, True)):
                    if threshold is not None and not criteria:
                        continue
                    mth = 1000 if exhaustive else None      # exhaustive_number_hits with max_total_hits (pagination.maxTotalHits)
                    want_ids, want_sc, want_cand = RO.search(RO.Ctx(index, lookup), q, tms="last", criteria=criteria,
                                                             offset=offset, length=limit, detailed=detailed, sort=sort,
                                                             distinct=field, threshold=threshold, exhaustive=exhaustive,
                                                             max_total_hits=mth)
                    hits, cand = h.search(q, criteria=criteria, offset=offset, limit=limit, detailed=detailed, sort=sort,
                                          distinct=field, score_threshold=threshold, exhaustive=exhaustive, max_total_hits=mth)
                    assert [d for d, _ in hits] == want_ids, (field, criteria, sort, q, detailed, offset, threshold)
                    assert [[tuple(s) for s in sc] for _, sc in hits] == [[G.oracle_score(s) for s in sc] for sc in want_sc]
                    assert cand == len(want_cand), (field, criteria, sort, q, detailed, offset, threshold)
                    n += 1
    assert n >= 600 or setups is not DISTINCT_SETUPS
    h.close()


def test_sort_under_distinct_hands_out_the_values_distinct_emptied(hostlib):
    """sort.rs:214-217 (`bucket.candidates &= universe`): the Sort rule's facet iterator was built over the universe the
    rule STARTED with, so a value whose documents `distinct` removed meanwhile still comes out — as an empty bucket, one
    more turn of bucket_sort's loop.  Only the number of deadline checks shows it: searches cut off after 0..6 loop
    iterations (`stop_after`, the reference's own cutoff tests' device) must stop at the same bucket as the oracle —
    same hits, same `Skipped` details, same degraded flag.  (Until round 3 the product's rule skipped those values.)"""
    from oracle import ranking_oracle as RO
    import tests.test_search_gpu as G
    index = ToyMilli(sortable_corpus(9, 220), searchable=["title", "body"])
    dic = O.Dictionary(index.words)

    def lookup(word, max_typos, is_prefix):
        one, two = O.typo_lookup(dic, word, max_typos, is_prefix)
        return [index.words[i] for i in one], [index.words[i] for i in two]
    h = make_harness(hostlib, index)
    n = degraded = 0
    for field in ("color", "sizes", "price"):
        for criteria, sort in ((["sort", "words", "typo"], [("price", "asc")]), (["words", "sort", "proximity"], [("color", "desc")]),
                               (["sort"], [("sizes", "asc"), ("price", "desc")]), (["desc:price", "words"], None)):
            for q in ["", "the", "quick fox", "brwn fox jumps"]:
                for stop_after in (0, 1, 2, 3, 4, 6):
                    want = RO.search(RO.Ctx(index, lookup), q, tms="last", criteria=criteria, offset=0, length=12, detailed=True,
                                     sort=sort, distinct=field, stop_after=stop_after)
                    hits, cand, deg = h.search(q, criteria=criteria, offset=0, limit=12, detailed=True, sort=sort, distinct=field,
                                               stop_after=stop_after, return_degraded=True)
                    assert [d for d, _ in hits] == want[0], (field, criteria, sort, q, stop_after)
                    assert [[geo_score(s) for s in sc] for _, sc in hits] == \
                        [[geo_score(G.oracle_score(s)) if s[0] != "Skipped" else ("Skipped", 0, 1) for s in sc] for sc in want[1]], \
                        (field, criteria, q, stop_after)
                    assert cand == len(want[2])
                    degraded += int(bool(deg))
                    n += 1
    assert n == 288 and degraded >= 100
    h.close()


@pytest.mark.parametrize("exact_attributes", [(), ("body",)], ids=["no-exact-attribute", "body-exact"])
def test_attributes_to_search_on_views_match_the_oracle(hostlib, exact_attributes):
    """`attributesToSearchOn` (search/new/mod.rs:140-222): the request reads the index through a restricted VIEW — word_docids
    becomes the union of word_fid_docids over the restricted tolerant fields, exact_word_docids the union over the restricted
    exact fields, the prefix databases likewise, word_fid_docids outside the restriction is absent (db_cache.rs:208-345,
    540-575).  The shim's callbacks answer that way (tests/toy_milli.py: ToyMilli.restricted) and name the view in
    msi_search_params::index_view; hits, score details and candidate counts against the oracle reading the same view."""
    from oracle import ranking_oracle as RO
    import tests.test_search_gpu as G
    docs = G.random_corpus(21, 260)
    for i, d in enumerate(docs):
        d["tags"] = " ".join(G.VOCAB[(i * 7 + k * 3) % len(G.VOCAB)] for k in range(i % 4))
    index = ToyMilli(docs, searchable=["title", "body", "tags"], exact_attributes=exact_attributes, prefix_threshold=3)
    dic = O.Dictionary(index.words)

    def lookup(word, max_typos, is_prefix):
        one, two = O.typo_lookup(dic, word, max_typos, is_prefix)
        return [index.words[i] for i in one], [index.words[i] for i in two]
    n = differs = 0
    for attrs in (["title"], ["body"], ["tags", "title"], ["unknown"], ["*"], ["body", "tags"]):
        view = index.restricted(attrs)
        h = make_harness(hostlib, view)
        for criteria in (None, ["words", "attribute", "exactness"], ["typo", "proximity", "wordPosition"]):
            for q in ["quick fox", "the lazy dog", "sun fl", "brwn fox jumps", "\"lazy dog\" summer", "su"]:
                for tms in ("last", "all", "frequency"):
                    want = RO.search(RO.Ctx(view, lookup), q, tms=tms, criteria=criteria, offset=0, length=25, detailed=True)
                    hits, cand = h.search(q, tms=tms, criteria=criteria, offset=0, limit=25, detailed=True,
                                          index_view=getattr(view, "index_view", 0))
                    assert [d for d, _ in hits] == want[0], (attrs, criteria, q, tms)
                    assert [[tuple(s_) for s_ in sc] for _, sc in hits] == [[G.oracle_score(s_) for s_ in sc] for sc in want[1]], (attrs, q)
                    assert cand == len(want[2]), (attrs, criteria, q, tms)
                    if view is not index:
                        plain = RO.search(RO.Ctx(index, lookup), q, tms=tms, criteria=criteria, offset=0, length=25, detailed=True)
                        differs += int(plain[0] != want[0] or len(plain[2]) != len(want[2]))
                    n += 1
        h.close()
    assert n == 6 * 3 * 6 * 3 and differs >= 100      # (the restriction changes most answers: the comparison is not vacuous)


GEO = json.load(open(os.path.join(ROOT, "tests", "golden", "geo_snapshots.json")))


def test_geo_sort_rs_through_the_host_logic(hostlib):
    """The 18 searches of the reference's geo_sort.rs (docids and the score details of every hit) and the properties of
    its max-bucket-size test, through msi_keyword_search_ranked."""
    for case in GEO["cases"]:
        cfg = GEO["indexes"][case["index"]]
        index = ToyMilli(cfg["docs"], criteria=cfg["criteria"])
        h = make_harness(hostlib, index)
        sort = [(tuple(f) if isinstance(f, list) else f, d) for f, d in case["sort"]]
        # the strategy settings the reference itself asserts to agree on this data (tests/geo_sort.rs:30-66)
        for strategy in (("dynamic", 1000), ("iterative", 1000), ("rtree", 1000)):
            hits, _ = h.search(case["query"], limit=20, detailed=True, sort=sort, geo_strategy=strategy)
            assert [index.docs[d]["id"] for d, _ in hits] == case["ids"], (case["src"], case["sort"], strategy)
            assert "[" + "".join("[" + "".join(debug_score(s) + "," for s in sc) + "]," for _, sc in hits) + "]" == case["scores"]
        if "with_following_ranking_rules" in case["src"] and case["sort"][0][1] == "asc":
            hits, _ = h.search(case["query"], limit=20, detailed=True, sort=sort, geo_max_bucket_size=2)
            ext = [index.docs[d]["id"] for d, _ in hits]
            assert len(ext) == 15 and all(6 <= i <= 11 for i in ext[:6]) and all(12 <= i <= 15 for i in ext[6:10])
            assert ext[10:] == [1, 4, 3, 2, 5]
        h.close()


def geo_corpus(seed, n_docs):
    """sortable_corpus plus _geo: places shared by many documents, neighbours decimetres to metres apart (the error
    margin), a fifth of the documents without a point."""
    import random
    rng = random.Random(seed * 13 + 5)
    docs = sortable_corpus(seed, n_docs)
    places = [(rng.uniform(-80, 80), rng.uniform(-179, 179)) for _ in range(9)]
    for d in docs:
        r = rng.random()
        if r < 0.2:
            continue
        lat, lng = rng.choice(places) if r < 0.7 else (rng.uniform(-89, 89), rng.uniform(-180, 180))
        if 0.45 < r < 0.7:
            lat += rng.choice([2e-6, 5e-6, 2e-5, 1e-4])
        d["_geo"] = {"lat": lat, "lng": lng}
    return docs


GEO_SETUPS = [
    (["words", "sort", "typo"], [(("_geoPoint", 10.0, 20.0), "asc")], {}),
    (["words", "sort", "typo"], [(("_geoPoint", -35.5, 140.25), "desc"), ("price", "asc")], {}),
    (["sort", "words", "proximity"], [("color", "asc"), (("_geoPoint", 0.0, 179.5), "asc")], {}),
    (["words", "typo", "sort"], [(("_geoPoint", 48.85, 2.35), "asc"), ("price", "desc")], {"geo_max_bucket_size": 4}),
    (["words", "sort"], [(("_geoPoint", 48.85, 2.35), "desc")], {"geo_distance_error_margin": 5000000.0}),
]


def test_geo_sort_matches_the_oracle(hostlib, setups=GEO_SETUPS, with_distinct=True):
    """GeoSort between graph-based rules, before / after Sort rules, on placeholder searches, with a small bucket cap,
    a huge error margin and `distinct`: hits, score details (the value of every bucket) and all_candidates against
    the oracle under the reference's three strategies — Dynamic(1000), the default (240 documents: every fill is
    iterative, distances truncated to metres and docid order inside a metre decide neighbours closer than the margin),
    always-iterative and always-rtree (exact distance order, the min / take kernels)."""
    from oracle import ranking_oracle as RO
    import tests.test_search_gpu as G
    index = ToyMilli(geo_corpus(3, 240), searchable=["title", "body"])
    dic = O.Dictionary(index.words)

    def lookup(word, max_typos, is_prefix):
        one, two = O.typo_lookup(dic, word, max_typos, is_prefix)
        return [index.words[i] for i in one], [index.words[i] for i in two]
    h = make_harness(hostlib, index)
    n = 0
    for criteria, sort, geo in setups:
        for q in ["", "the", "quick fox", "sun fl", "brwn fox jumps"]:
            for detailed, offset, limit, distinct in ((True, 0, 40, None), (False, 0, 9, None), (True, 11, 9, None),
                                                       (True, 0, 25, "color")):
                if distinct and not with_distinct:
                    continue
                for strategy in (("dynamic", 1000), ("rtree", 1000), ("iterative", 1000)):
                    n += geo_case(RO, G, index, lookup, h, criteria, sort, geo, q, detailed, offset, limit, distinct, strategy)
    assert n >= 100 or setups is not GEO_SETUPS
    h.close()


def geo_case(RO, G, index, lookup, h, criteria, sort, geo, q, detailed, offset, limit, distinct, strategy):
    if True:
        if True:
            if True:
                RO.GEO_PARAMS.clear()
                RO.GEO_PARAMS.update(strategy=strategy)
                if "geo_max_bucket_size" in geo:
                    RO.GEO_PARAMS["max_bucket_size"] = geo["geo_max_bucket_size"]
                if "geo_distance_error_margin" in geo:
                    RO.GEO_PARAMS["distance_error_margin"] = geo["geo_distance_error_margin"]
                try:
                    want_ids, want_sc, want_cand = RO.search(RO.Ctx(index, lookup), q, tms="last", criteria=criteria,
                                                             offset=offset, length=limit, detailed=detailed, sort=sort,
                                                             distinct=distinct)
                finally:
                    RO.GEO_PARAMS.clear()
                hits, cand = h.search(q, criteria=criteria, offset=offset, limit=limit, detailed=detailed, sort=sort,
                                      distinct=distinct, geo_strategy=strategy, **geo)
                assert [d for d, _ in hits] == want_ids, (criteria, sort, geo, q, detailed, offset, distinct, strategy)
                assert [[geo_score(s) for s in sc] for _, sc in hits] == [[geo_score(G.oracle_score(s)) for s in sc] for sc in want_sc]
                assert cand == len(want_cand)
    return 1


def geo_score(s):
    s = tuple(s)
    return (s[0], tuple(s[1]), s[2], None if s[3] is None else tuple(s[3])) if s[0] == "GeoSort" else s


CRIT = json.load(open(os.path.join(ROOT, "tests", "golden", "criteria_fixtures.json")))
CRIT_DOCS = json.load(open(os.path.join(ROOT, "tests", "golden", "filter_fixtures.json")))["docs"]


def criteria_cases(every=1):
    return [c for i, c in enumerate(CRIT["cases"]) if c["name"] != "criteria_mixup" or i % every == 0]


def test_oracle_replays_the_reference_criteria_tests():
    """crates/milli/tests/search/query_criteria.rs: "hello world america" over test_set.ndjson with its synonyms, the 14
    `test_criterion!` cases and the 120 criteria orders of `criteria_mixup`; the expected order comes from the rank
    columns of the dataset (the reference's own expected_order helper)."""
    from oracle import ranking_oracle as RO
    assert len(CRIT["cases"]) == 134
    for case in CRIT["cases"]:
        index = ToyMilli(CRIT_DOCS, searchable=CRIT["searchable"], criteria=case["criteria"], synonyms=CRIT["synonyms"])
        dic = O.Dictionary(index.words)

        def lookup(word, max_typos, is_prefix):
            one, two = O.typo_lookup(dic, word, max_typos, is_prefix)
            return [index.words[i] for i in one], [index.words[i] for i in two]
        ids, _, _ = RO.search(RO.Ctx(index, lookup), CRIT["query"], tms=case["tms"], criteria=case["criteria"], length=17,
                              sort=[tuple(x) for x in case["sort"]])
        assert [index.docs[d]["id"] for d in ids] == case["ids"], (case["name"], case["criteria"])


def test_reference_criteria_tests_through_the_host_logic(hostlib, every=1):
    for case in criteria_cases(every):
        index = ToyMilli(CRIT_DOCS, searchable=CRIT["searchable"], criteria=case["criteria"], synonyms=CRIT["synonyms"])
        h = make_harness(hostlib, index)
        hits, _ = h.search(CRIT["query"], tms=case["tms"], criteria=case["criteria"], limit=17,
                           sort=[tuple(x) for x in case["sort"]])
        assert [index.docs[d]["id"] for d, _ in hits] == case["ids"], (case["name"], case["criteria"])
        h.close()


def test_reference_distinct_integration_tests(hostlib):
    """crates/milli/tests/search/distinct.rs: 19 `test_distinct!` cases over test_set.ndjson — the literal candidates
    counts (3 distinct tags, 7 distinct asc_desc_ranks; exhaustive or not), the first document of every value in
    ranking order, offsets on the rule-less path, a limit of 0 — oracle and product."""
    from oracle import ranking_oracle as RO
    assert len(CRIT["distinct_cases"]) == 19
    for case in CRIT["distinct_cases"]:
        index = ToyMilli(CRIT_DOCS, searchable=CRIT["searchable"], criteria=case["criteria"], synonyms=CRIT["synonyms"])
        dic = O.Dictionary(index.words)

        def lookup(word, max_typos, is_prefix):
            one, two = O.typo_lookup(dic, word, max_typos, is_prefix)
            return [index.words[i] for i in one], [index.words[i] for i in two]
        ids, _, cand = RO.search(RO.Ctx(index, lookup), CRIT["query"], criteria=case["criteria"], offset=case["offset"],
                                 length=case["limit"], distinct=case["distinct"], exhaustive=case["exhaustive"])
        assert [index.docs[d]["id"] for d in ids] == case["ids"] and len(cand) == case["candidates"], case["name"]
        h = make_harness(hostlib, index)
        hits, n_cand = h.search(CRIT["query"], criteria=case["criteria"], offset=case["offset"], limit=case["limit"],
                                distinct=case["distinct"], exhaustive=case["exhaustive"])
        assert [index.docs[d]["id"] for d, _ in hits] == case["ids"], case["name"]
        assert n_cand == case["candidates"], case["name"]
        h.close()


def test_reference_typo_tolerance_and_phrase_integration_tests(hostlib):
    """crates/milli/tests/search/typo_tolerance.rs and phrase_search.rs over test_set.ndjson: hit counts under the typo
    thresholds (min word length for 1 / 2 typos), exact words, exact attributes, and a phrase of stop words — oracle
    and product."""
    from oracle import ranking_oracle as RO
    two = [{"id": 1, "data": "zealand"}, {"id": 2, "data": "zearand"}]
    cases = [  # (docs, settings, criteria, query, tms, hits)
        (CRIT_DOCS, {}, ["typo"], "zeal", "last", 1),                                     # typo_tolerance.rs:37-43
        (CRIT_DOCS, {}, ["typo"], "zean", "last", 0),                                     # :54-60: 4 letters, no typo by default
        (CRIT_DOCS, {"min_one": 4}, ["typo"], "zean", "last", 1),                         # :68-95
        (CRIT_DOCS, {}, ["typo"], "zealand", "last", 1),                                  # :117-123
        (CRIT_DOCS, {}, ["typo"], "zealemd", "last", 0),                                  # :134-140: 7 letters, one typo at most
        (CRIT_DOCS, {"min_two": 7}, ["typo"], "zealemd", "last", 1),                      # :148-175
        (two, {"searchable": None}, None, "zealand", "last", 2),                          # :249-255
        (two, {"searchable": None, "exact_words": ["zealand"]}, None, "zealand", "last", 1),   # :263-292
        (CRIT_DOCS, {}, ["typo"], "antebelum", "last", 1),                                # :315-321
        (CRIT_DOCS, {"exact_attributes": ["description"]}, ["typo"], "antebelum", "last", 0),  # :330-356
        (CRIT_DOCS, {"stop_words": ["a", "an", "the", "of"]}, [], '"the use of force"', "all", 1),   # phrase_search.rs:27-58
        (CRIT_DOCS, {"stop_words": ["a", "an", "the", "of"]}, ["proximity", "attribute", "exactness"], '"the use of force"', "all", 1),
    ]
    for docs, settings, criteria, q, tms, n_hits in cases:
        kw = dict(settings)
        searchable = kw.pop("searchable", CRIT["searchable"])
        index = ToyMilli(docs, searchable=searchable, criteria=criteria, synonyms=CRIT["synonyms"] if docs is CRIT_DOCS else None, **kw)
        dic = O.Dictionary(index.words)

        def lookup(word, max_typos, is_prefix):
            one, two_ = O.typo_lookup(dic, word, max_typos, is_prefix)
            return [index.words[i] for i in one], [index.words[i] for i in two_]
        ids, _, _ = RO.search(RO.Ctx(index, lookup), q, tms=tms, criteria=index.criteria, length=10)
        assert len(ids) == n_hits, (q, settings, "oracle")
        h = make_harness(hostlib, index)
        hits, _ = h.search(q, tms=tms, limit=10)
        assert [d for d, _ in hits] == ids, (q, settings)
        h.close()


# seeds of tools/fuzz_ranked_hostlogic.py (the corpus / settings / queries of seed + 1) that once disagreed with the oracle
FUZZ_REGRESSIONS = {
    # a rule ended before its buckets covered its universe (bucket_sort.rs `back!` drops what is left): the bucket sort's
    # tree of tasks had handed out result places by cumulative cardinality and left holes
    25606: "a rule that drops documents, first page", 33401: "the same below the third hit", 38598: "... and beyond the tenth",
}


def run_fuzz_seeds(seeds, *mode):
    import subprocess, sys
    for seed in seeds:
        out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fuzz_ranked_hostlogic.py"), str(seed), "0.01", *mode],
                             cwd=ROOT, capture_output=True, text=True, timeout=600)
        tail = out.stdout[-2000:] + out.stderr[-2000:]
        assert out.returncode == 0 and "cases 6 bad 0" in out.stdout, tail   # (six searches per seed)


def test_reference_matching_strategy_literals_through_the_host_logic(hostlib):
    """crates/meilisearch/tests/search/matching_strategy.rs (tests/golden/matching_strategy_fixtures.json): last / all /
    frequency, three searches each — the only literals the reference holds for TermsMatchingStrategy::Frequency."""
    import json
    fix = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "matching_strategy_fixtures.json")))
    index = ToyMilli(fix["documents"])
    h = make_harness(hostlib, index)
    for case in fix["cases"]:
        hits, _ = h.search(case["query"], tms=case["strategy"], limit=20)
        assert [index.docs[d]["id"] for d, _ in hits] == case["ids"], case
    h.close()


def test_fuzz_regressions():
    run_fuzz_seeds(FUZZ_REGRESSIONS)
