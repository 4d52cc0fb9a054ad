"""TermsMatchingStrategy::Frequency (query_graph.rs:303-344, graph_based_ranking_rule.rs:174-190, mod.rs:288-292) in the
oracle, pinned by the only literals the reference holds for it: the hit ids of
crates/meilisearch/tests/search/matching_strategy.rs (tests/golden/matching_strategy_fixtures.json, extracted by
tests/golden/make_matching_strategy_fixtures.py) — 9 searches, three per strategy."""
import json
import os

import pytest

from oracle import ranking_oracle as R
from tests.test_ranking_oracle_snapshots import make_ctx
from tests.toy_milli import ToyMilli

FIX = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "matching_strategy_fixtures.json")))


def build_index():
    return ToyMilli(FIX["documents"])


@pytest.mark.parametrize("case", FIX["cases"], ids=[f'{c["query"]}:{c["strategy"]}' for c in FIX["cases"]])
def test_reference_matching_strategy_literals(case):
    index = build_index()
    ids, _, _ = R.search(make_ctx(index), case["query"], tms=case["strategy"], length=20)
    assert [index.docs[d]["id"] for d in ids] == case["ids"]


def test_frequency_weights():
    """The removal order itself on a corpus where the frequencies are known: the most frequent term leaves first, equal
    frequencies leave together, a term without any document counts as the most frequent of all."""
    docs = [{"id": i, "t": " ".join(w for w, every in (("common", 1), ("medium", 3), ("rare", 10), ("rare2", 10)) if i % every == 0)}
            for i in range(60)]
    index = ToyMilli(docs)
    ctx = make_ctx(index)
    terms = R.parse_query(ctx, "rare common medium rare2 absent ")       # trailing separator: no prefix term
    g = R.QueryGraph.from_query(ctx, terms)
    order = g.removal_order_frequency(ctx)
    term_of = {i: ctx.terms[n.term.subset.term].original for i, n in enumerate(g.nodes) if n.kind == "term"}
    names = [sorted(term_of[i] for i in grp if " " not in term_of[i] and term_of[i] in ("rare", "common", "medium", "rare2", "absent"))
             for grp in order]
    # n-gram nodes ride along with the heaviest of their terms; look only at the single words
    flat = [n for n in names if n]
    assert flat[0] == ["absent"] and flat[1] == ["common"] and flat[2] == ["medium"]
    # rare and rare2 tie: the last group is kept (popped), so they never appear
    assert all("rare" not in n and "rare2" not in n for n in flat)
