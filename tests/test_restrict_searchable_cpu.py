"""`attributesToSearchOn` pinned by the reference's own tests: the 14 flat-document tests of
crates/meilisearch/tests/search/restrict_searchable.rs (tests/golden/restrict_searchable_fixtures.json, extracted by
tests/golden/make_restrict_searchable_fixtures.py: 16 searches — hit counts, and the hits' ids / titles in order for the
Words / Typo / Attribute / Exactness rule-order tests and the phrase test) through the oracle reading the restricted view of
the toy index (tests/toy_milli.py: ToyMilli.restricted = db_cache.rs:208-345,540-575) and through the product's host logic
with the view named in msi_search_params::index_view."""
import json
import os

import pytest

from oracle import oracle as O
from oracle import ranking_oracle as RO
from tests.toy_milli import ToyMilli

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FIX = json.load(open(os.path.join(ROOT, "tests", "golden", "restrict_searchable_fixtures.json")))


def replay(case, run):
    """Applies the test's settings changes and searches in order; run(index, view, params) -> internal docids."""
    settings = {}
    n = 0
    for ev in case["events"]:
        if "settings" in ev:
            settings.update(ev["settings"])
            continue
        sa = settings.get("searchableAttributes")
        typo = settings.get("typoTolerance", {})
        index = ToyMilli(case["documents"], searchable=None if sa in (None, ["*"]) else sa,
                         exact_words=[w.lower() for w in typo.get("disableOnWords", [])],
                         exact_attributes=typo.get("disableOnAttributes", ()))
        p = ev["search"]
        view = index.restricted(p["attributesToSearchOn"])
        ids = run(index, view, p)
        want = ev["want"]
        if "n_hits" in want:
            assert len(ids) == want["n_hits"], (case["src"], p)
        if "hits" in want:
            field = next(iter(want["hits"][0]))
            assert [index.docs[d][field] for d in ids] == [h[field] for h in want["hits"]], (case["src"], p)
        n += 1
    return n


@pytest.mark.parametrize("case", FIX["cases"], ids=[c["src"].split("::")[1] for c in FIX["cases"]])
def test_reference_restrict_searchable_through_the_oracle(case):
    def run(index, view, p):
        dic = O.Dictionary(index.words)

        def lookup(word, max_typos, is_prefix):
            one, two = O.typo_lookup(dic, word, max_typos, is_prefix)
            return [index.words[i] for i in one], [index.words[i] for i in two]
        ids, _, _ = RO.search(RO.Ctx(view, lookup), p["q"], tms=p.get("matchingStrategy", "last"), offset=0, length=20)
        return ids
    assert replay(case, run) >= 1


def test_reference_restrict_searchable_through_the_host_logic():
    import tests.test_search_hostlogic_cpu as H
    L = H.load_hostlib()
    n = 0
    for case in FIX["cases"]:
        def run(index, view, p):
            h = H.make_harness(L, view)
            try:
                hits, _ = h.search(p["q"], tms=p.get("matchingStrategy", "last"), offset=0, limit=20,
                                   index_view=getattr(view, "index_view", 0))
            finally:
                h.close()
            return [d for d, _ in hits]
        n += replay(case, run)
    assert n == 16
