"""msi_keyword_search_ranked (the product: host rule graphs + device docid sets) against
  1. the reference's own snapshots (tests/golden/ranking_snapshots.json: docid order and the score details
     of every hit for the searches of search/new/tests/*.rs ), and
  2. the CPU oracle (oracle/ranking_oracle.py, itself pinned to the same snapshots) on random corpora,
     all rule lists, both terms-matching strategies."""
import json
import os
import random

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

FIX = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "ranking_snapshots.json")))
UNSUPPORTED = {}


class Harness:
    def __init__(self, index, n_slots=512):
        import meilisearch_amd as ma
        from meilisearch_amd import ranking as R
        self.R, self.index = R, index
        self.ctx = ma.Context(0)
        self.dict = ma.GpuDictionary(self.ctx, [w.encode() for w in index.words])
        self.pool = ma.BitsPool(self.ctx, max(index.n_docs, 1), n_slots)
        self.cb = R.IndexCallbacks(index)

    def search(self, query, tms="last", criteria=None, offset=0, limit=20, detailed=False, stop_after=None):
        from tests.toy_milli import query_terms
        R, ix = self.R, self.index
        return R.keyword_search_ranked(
            self.dict, self.pool, self.cb, query_terms(query, stop_words=ix.stop_words), criteria if criteria is not None else ix.criteria,
            strategy=R.strategy_of(tms), offset=offset, limit=limit, detailed=detailed,
            searchable_fids=ix.searchable_fids, searchable_weights=[ix.weights[f] for f in ix.searchable_fids],
            max_weight=ix.max_weight, authorize_typos=ix.authorize_typos, min_one=ix.min_one, min_two=ix.min_two,
            stop_after=stop_after)


def debug_score(s):
    k = s[0]
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


def build_index(cfg):
    from tests.toy_milli import ToyMilli
    return ToyMilli(cfg["docs"], searchable=cfg.get("searchable"), exact_attributes=cfg.get("exact_attributes", ()),
                    exact_words=cfg.get("exact_words", ()), criteria=cfg.get("criteria"),
                    min_one=cfg.get("min_one", 5), min_two=cfg.get("min_two", 9),
                    authorize_typos=cfg.get("authorize_typos", True), synonyms=cfg.get("synonyms"),
                    stop_words=cfg.get("stop_words", ()), distinct=cfg.get("distinct"))


_H = {}


def harness_for(key):
    if key not in _H:
        _H[key] = Harness(build_index(FIX["indexes"][key]))
    return _H[key]


# distinct.rs cases are pinned in the oracle only: `distinct` (and Sort) are not built in the product yet
CASES = [c for c in FIX["cases"] if not FIX["indexes"][c["index"]].get("unsupported") and c["query"] not in UNSUPPORTED
         and not c.get("needs") and not c.get("sort") and not c.get("distinct") and not FIX["indexes"][c["index"]].get("distinct")]


@pytest.mark.parametrize("case", CASES, ids=[f'{c["src"].split("::")[1]}:{c["query"]}' for c in CASES])
def test_reference_snapshot(case):
    h = harness_for(case["index"])
    hits, _ = h.search(case["query"], tms=case["tms"], offset=case["offset"], limit=case["limit"],
                       detailed=case["detailed"], stop_after=case.get("stop_after"))
    ids = [d for d, _ in hits]
    if case.get("global_scores"):
        assert [f"{h.R.score_details_global_score(sc):.4f}" for _, sc in hits] == case["global_scores"]
    if case["ids"] is not None:
        assert ids == case["ids"]
    if case.get("scores"):
        got = "[" + "".join("[" + "".join(debug_score(s) + "," for s in sc) + "]," for _, sc in hits) + "]"
        assert got == case["scores"]
    if case.get("ids_scores"):
        got = "[" + "".join(f"({d},[" + "".join(debug_score(s) + "," for s in sc) + "],)," for d, sc in hits) + "]"
        assert got == case["ids_scores"]


VOCAB = ("the quick brown fox jumps over lazy dog sun flower sunflower summer winter holiday beautiful "
         "delicious sweet dessert network interconnection quack quickest brownish foxes jump dogs").split()


def random_corpus(seed, n_docs):
    rng = random.Random(seed)
    docs = []
    for i in range(n_docs):
        def text(lo, hi):
            ws = [rng.choice(VOCAB) for _ in range(rng.randint(lo, hi))]
            out = []
            for w in ws:
                out.append(w + (". " if rng.random() < 0.05 else " "))
            return "".join(out).strip()
        docs.append({"id": i, "title": text(1, 5), "body": text(3, 30)})
    return docs


RULESETS = [
    ["words", "typo", "proximity", "attributeRank", "sort", "wordPosition", "exactness"],   # the default criteria
    ["words", "proximity"], ["words", "typo", "proximity"], ["attribute"], ["exactness"], ["words", "exactness", "typo"],
    ["typo", "words"], ["proximity", "typo"],
]
QUERIES = ["the qui", "brown f", "s", "sun fl", "lazy d", "the quick brown fox", "sunflower", "sun flower holiday", "quick fox jumps over the lazy dog",
           "beautiful summer", "delicious sweet dessert", "\"quick brown\" fox", "the \"lazy dog\" jumps",
           "quik brwn fox", "network interconection", "winter holi", "fox", "dog the"]


@pytest.mark.parametrize("seed,prefix_threshold", [(1, 100), (2, 100), (3, 3)])
def test_matches_oracle_on_random_corpora(seed, prefix_threshold):
    """prefix_threshold = 3 gives the toy corpus word-prefix databases (use_prefix_db terms)."""
    from oracle import oracle as O
    from oracle import ranking_oracle as RO
    from tests.toy_milli import ToyMilli
    docs = random_corpus(seed, 300)
    checked = 0
    for criteria in RULESETS:
        index = ToyMilli(docs, searchable=["title", "body"], criteria=criteria, prefix_threshold=prefix_threshold)
        assert (len(index.prefixes) > 0) == (prefix_threshold == 3)
        dic = O.Dictionary(index.words)

        def lookup(word, max_typos, is_prefix):
            one, two = O.typo_lookup(dic, word, max_typos, is_prefix)
            return [index.words[i] for i in one], [index.words[i] for i in two]
        h = Harness(index)
        for q in QUERIES:
            for tms in ("last", "all", "frequency"):
                for detailed, offset in ((True, 0), (False, 3)):
                    want_ids, want_sc, want_cand = RO.search(RO.Ctx(index, lookup), q, tms=tms, criteria=criteria,
                                                             offset=offset, length=25, detailed=detailed)
                    hits, cand = h.search(q, tms=tms, criteria=criteria, offset=offset, limit=25, detailed=detailed)
                    assert [d for d, _ in hits] == want_ids, (criteria, q, tms, detailed, offset)
                    assert [[tuple(s) for s in sc] for _, sc in hits] == \
                        [[oracle_score(s) for s in sc] for sc in want_sc], (criteria, q, tms)
                    assert cand == len(want_cand)
                    checked += 1
    assert checked == len(RULESETS) * len(QUERIES) * 6


def oracle_score(s):
    if s[0] == "ExactAttribute":
        return ("ExactAttribute", {"ExactMatch": 3, "MatchesStart": 2, "NoExactMatch": 1}[s[1]], 3)
    return tuple(s)


def test_concurrent_searches_on_private_streams():
    """One pool with a private stream per caller thread: the results of 4 threads x 6 queries x 2 strategies
    equal the single-threaded ones (completion signals, staging rings and slot allocators are per pool)."""
    import threading
    import meilisearch_amd as ma
    from tests.toy_milli import ToyMilli
    index = ToyMilli(random_corpus(7, 400), searchable=["title", "body"])
    base = Harness(index)
    want = {(q, tms): base.search(q, tms=tms, detailed=True) for q in QUERIES[:6] for tms in ("last", "all")}
    errors = []

    def worker(k):
        try:
            h = Harness.__new__(Harness)
            h.R, h.index, h.ctx, h.dict = base.R, index, base.ctx, base.dict
            h.pool = ma.BitsPool(base.ctx, index.n_docs, 512, private_stream=True)
            h.cb = base.R.IndexCallbacks(index)
            for (q, tms), w in want.items():
                got = h.search(q, tms=tms, detailed=True)
                if got != w:
                    errors.append((k, q, tms))
        except Exception as e:   # noqa: BLE001
            errors.append((k, repr(e)))
    ths = [threading.Thread(target=worker, args=(k,)) for k in range(4)]
    for t in ths:
        t.start()
    for t in ths:
        t.join()
    assert not errors, errors[:5]


def test_negative_words_and_phrases():
    """`-word` / `-"a phrase"`: their documents leave the universe before anything else (search/mod.rs:431-440)."""
    from oracle import oracle as O
    from oracle import ranking_oracle as RO
    from tests.toy_milli import ToyMilli, query_terms
    index = ToyMilli(random_corpus(11, 300), searchable=["title", "body"])
    dic = O.Dictionary(index.words)

    def lookup(word, max_typos, is_prefix):
        one, two = O.typo_lookup(dic, word, max_typos, is_prefix)
        return [index.words[i] for i in one], [index.words[i] for i in two]
    h = Harness(index)
    R = h.R
    for q, negs in (("the quick fox", ["brown"]), ("sun flower", [("lazy", "dog")]), ("dog", ["the", ("quick", "brown")])):
        want_ids, want_sc, want_cand = RO.search(RO.Ctx(index, lookup), q, tms="last", detailed=True, negatives=negs)
        terms = query_terms(q)
        for ng in negs:
            terms.append(([ng], False, 0, 0, False, True) if isinstance(ng, str) else (list(ng), True, 0, 0, False, True))
        hits, cand = R.keyword_search_ranked(h.dict, h.pool, h.cb, terms, index.criteria, detailed=True,
                                             searchable_fids=index.searchable_fids,
                                             searchable_weights=[index.weights[f] for f in index.searchable_fids],
                                             max_weight=index.max_weight)
        assert [d for d, _ in hits] == want_ids and cand == len(want_cand)
        assert want_ids, "the case should keep some documents"


SETTINGS = [
    dict(exact_attributes=["title"]),
    dict(exact_words=["quick", "sunflower"], min_one=4, min_two=7),
    dict(authorize_typos=False),
    dict(synonyms={"fast": ["quick"], "sunflower": ["sun flower", "helianthus"], "lazy dog": ["sleepy hound"]},
         stop_words=["the", "over"]),
    dict(prefix_threshold=2, exact_attributes=["body"]),
]


@pytest.mark.parametrize("settings", SETTINGS, ids=[",".join(s_) for s_ in SETTINGS])
def test_matches_oracle_under_index_settings(settings):
    """Exact attributes / exact words / typo thresholds / synonyms + stop words / prefix databases on a three-field
    corpus, default criteria, both strategies, with and without an offset."""
    from oracle import oracle as O
    from oracle import ranking_oracle as RO
    from tests.toy_milli import ToyMilli, query_terms
    rng = random.Random(5)
    docs = []
    for d in random_corpus(21, 250):
        d["tags"] = " ".join(rng.choice(VOCAB) for _ in range(rng.randint(0, 3)))
        docs.append(d)
    index = ToyMilli(docs, searchable=["title", "body", "tags"], **settings)
    dic = O.Dictionary(index.words)

    def lookup(word, max_typos, is_prefix):
        one, two = O.typo_lookup(dic, word, max_typos, is_prefix)
        return [index.words[i] for i in one], [index.words[i] for i in two]
    h = Harness(index)
    queries = QUERIES + ["fast brown fox", "sunflower holiday", "the lazy dog jumps", "quick", "qu", "\"sun flower\" the"]
    for q in queries:
        for tms in ("last", "all", "frequency"):
            for detailed, offset in ((True, 0), (False, 2)):
                want_ids, want_sc, want_cand = RO.search(RO.Ctx(index, lookup), q, tms=tms, offset=offset, length=30,
                                                         detailed=detailed)
                hits, cand = h.search(q, tms=tms, offset=offset, limit=30, detailed=detailed)
                assert [d for d, _ in hits] == want_ids, (q, tms, detailed, offset)
                assert [[tuple(s) for s in sc] for _, sc in hits] == [[oracle_score(s) for s in sc] for sc in want_sc], (q, tms)
                assert cand == len(want_cand)


def test_ranking_score_threshold():
    """bucket_sort.rs:286-306: buckets whose global score so far is below the threshold are dropped together with
    the rest of that rule's universe; the candidate count excludes them."""
    from oracle import oracle as O
    from oracle import ranking_oracle as RO
    from tests.toy_milli import ToyMilli, query_terms
    index = ToyMilli(random_corpus(13, 300), searchable=["title", "body"])
    dic = O.Dictionary(index.words)

    def lookup(word, max_typos, is_prefix):
        one, two = O.typo_lookup(dic, word, max_typos, is_prefix)
        return [index.words[i] for i in one], [index.words[i] for i in two]
    h = Harness(index)
    seen_cut = 0
    for q in ("the quick brown fox", "sun flower holiday", "quick brwn fox", "lazy dog the"):
        for thr in (0.2, 0.5, 0.8, 0.95):
            for detailed in (True, False):
                want_ids, want_sc, want_cand = RO.search(RO.Ctx(index, lookup), q, tms="last", length=300,
                                                         detailed=detailed, threshold=thr)
                hits, cand = h.R.keyword_search_ranked(
                    h.dict, h.pool, h.cb, query_terms(q), index.criteria, limit=300, detailed=detailed,
                    searchable_fids=index.searchable_fids,
                    searchable_weights=[index.weights[f] for f in index.searchable_fids], max_weight=index.max_weight,
                    score_threshold=thr)
                assert [d for d, _ in hits] == want_ids, (q, thr, detailed)
                assert cand == len(want_cand), (q, thr, detailed)
                base = RO.search(RO.Ctx(index, lookup), q, tms="last", length=300, detailed=detailed)
                seen_cut += len(want_cand) < len(base[2])
    assert seen_cut > 4


def test_path_by_path_fallback_matches_too(monkeypatch):
    """Cost levels with more than 256 paths fall back to the path-by-path search (batched sibling intersections +
    claim kernel); MSI_SEARCH_FUSED_LEVELS=0 forces it everywhere: the reference snapshots must still replay."""
    monkeypatch.setenv("MSI_SEARCH_FUSED_LEVELS", "0")
    for case in CASES[::3]:
        test_reference_snapshot(case)


def test_known_outcomes_off_matches_too(monkeypatch):
    """A one-word search's Words / Proximity / single-level Typo evaluations are answered on the host (the bucket is the
    universe: msi_search.hip, GraphRule::known_outcome); MSI_SEARCH_KNOWN_OUTCOMES=0 sends them through the device again.
    Every other test runs with the shortcut on; the reference snapshots and a random corpus (its one-word queries under every
    rule order and matching strategy) must replay without it too."""
    monkeypatch.setenv("MSI_SEARCH_KNOWN_OUTCOMES", "0")
    for case in CASES[::2]:
        test_reference_snapshot(case)
    test_matches_oracle_on_random_corpora(2, 100)


def test_reference_matching_strategy_literals_on_the_device():
    """crates/meilisearch/tests/search/matching_strategy.rs: the hit ids of its 9 searches (three per strategy; the only
    literals the reference holds for TermsMatchingStrategy::Frequency) through msi_keyword_search_ranked."""
    from tests.toy_milli import ToyMilli
    fix = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "matching_strategy_fixtures.json")))
    index = ToyMilli(fix["documents"])
    h = Harness(index)
    assert sum(1 for c in fix["cases"] if c["strategy"] == "frequency") == 3
    for case in fix["cases"]:
        hits, _ = h.search(case["query"], tms=case["strategy"], limit=20)
        assert [index.docs[d]["id"] for d, _ in hits] == case["ids"], case
