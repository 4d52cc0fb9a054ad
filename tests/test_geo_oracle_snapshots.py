"""The oracle's GeoSort rule (oracle/ranking_oracle.py: GeoSortRule, a restatement of search/new/geo_sort.rs over
documents/geo_sort.rs) pinned to the 18 searches of the reference's geo_sort.rs — docids and the score details of
every hit — under the four strategy settings the reference itself asserts to agree (iterative / rtree, cache of 2 /
1000), plus the properties test_geo_sort_reached_max_bucket_size states."""
import json
import os

import pytest

from oracle import oracle as O, ranking_oracle as RO
from tests.test_search_hostlogic_cpu import debug_score
from tests.toy_milli import ToyMilli

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GEO = json.load(open(os.path.join(ROOT, "tests", "golden", "geo_snapshots.json")))
STRATEGIES = [("iterative", 2), ("iterative", 1000), ("rtree", 2), ("rtree", 1000), ("dynamic", 1000)]


def geo_index(key):
    cfg = GEO["indexes"][key]
    return ToyMilli(cfg["docs"], criteria=cfg["criteria"])


def oracle_search(index, case, **geo):
    dic = O.Dictionary(index.words)

    def lookup(word, max_typos, is_prefix):
        one, two = O.typo_lookup(dic, word, max_typos, is_prefix)
        return [index.words[i] for i in one], [index.words[i] for i in two]
    RO.GEO_PARAMS.clear()
    RO.GEO_PARAMS.update(geo)
    try:
        sort = [(tuple(f) if isinstance(f, list) else f, d) for f, d in case["sort"]]
        return RO.search(RO.Ctx(index, lookup), case["query"], criteria=index.criteria, length=20, detailed=True, sort=sort)
    finally:
        RO.GEO_PARAMS.clear()


def render(ids, scores):
    return "[" + "".join("[" + "".join(debug_score(s) + "," for s in sc) + "]," for sc in scores) + "]"


@pytest.mark.parametrize("strategy", STRATEGIES, ids=[f"{k}-{n}" for k, n in STRATEGIES])
def test_oracle_replays_geo_sort_rs(strategy):
    assert len(GEO["cases"]) == 18
    for case in GEO["cases"]:
        index = geo_index(case["index"])
        ids, scores, _ = oracle_search(index, case, strategy=strategy)
        assert [index.docs[d]["id"] for d in ids] == case["ids"], (case["src"], case["sort"])
        assert render(ids, scores) == case["scores"], (case["src"], case["sort"])


@pytest.mark.parametrize("strategy", [("iterative", 1000), ("rtree", 1000)])
def test_oracle_max_bucket_size(strategy):
    """geo_sort.rs::test_geo_sort_reached_max_bucket_size: with buckets of at most 2 documents the following Desc(score)
    rule no longer orders the documents of one place, but the places stay in distance order and what has no _geo is last."""
    case = next(c for c in GEO["cases"] if "with_following_ranking_rules" in c["src"] and c["sort"][0][1] == "asc")
    index = geo_index(case["index"])
    ids, _, _ = oracle_search(index, case, strategy=strategy, max_bucket_size=2)
    ext = [index.docs[d]["id"] for d in ids]
    assert len(ext) == 15 and all(6 <= i <= 11 for i in ext[:6]) and all(12 <= i <= 15 for i in ext[6:10])
    assert ext[10:] == [1, 4, 3, 2, 5]
