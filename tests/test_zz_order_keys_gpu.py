"""Sort / Asc / Desc ranking rules on the device: msi_bits_order_next against numpy, then the rules
inside the ranked keyword search against the reference's sort.rs snapshots and the oracle on random corpora.
The host side of this is held in the CPU tier (tests/test_search_hostlogic_cpu.py)."""
import json
import os

import numpy as np
import pytest

import meilisearch_amd as ma
from meilisearch_amd import ranking as R

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FIX = json.load(open(os.path.join(ROOT, "tests", "golden", "ranking_snapshots.json")))
NONE = 0xFFFFFFFF


@pytest.mark.parametrize("n_docs", [1, 63, 64, 65, 1000, 200003])
def test_order_next_against_numpy(n_docs):
    ctx = ma.Context(0)
    rng = np.random.default_rng(n_docs)
    pool = ma.BitsPool(ctx, n_docs, 4)
    keys = rng.integers(0, max(2, n_docs // 9), n_docs).astype(np.uint32)
    keys[rng.random(n_docs) < 0.2] = NONE
    dk = ma.DocKeys(ctx, keys)
    universe = np.nonzero(rng.random(n_docs) < 0.6)[0].astype(np.uint32)
    pool.set_from_docids(0, universe)
    pool.fill(1, True)                                   # stale content in the bucket slot must not survive
    left = set(universe.tolist())
    for _ in range(40):
        key, n = pool.order_next(dk, 0, 1)
        if not left:
            assert n == 0 and pool.count(1) == 0
            break
        want_key = min(int(keys[d]) for d in left)
        want = sorted(d for d in left if int(keys[d]) == want_key)
        assert (key, n) == (want_key, len(want))
        assert pool.to_docids(1).tolist() == want
        left -= set(want)
        assert pool.to_docids(0).tolist() == sorted(left)
    # a universe whose documents have no value at all: one Null bucket
    pool.set_from_docids(2, np.nonzero(keys == NONE)[0].astype(np.uint32))
    key, n = pool.order_next(dk, 2, 3)
    assert key == NONE and n == int((keys == NONE).sum()) and pool.count(2) == 0
    with pytest.raises(ma.MsiError):
        pool.order_next(ma.DocKeys(ctx, np.zeros(n_docs + 1, np.uint32)), 0, 1)


class SortHarness:
    def __init__(self, index):
        import tests.test_search_gpu as G
        self.h = G.Harness(index)
        self.index = index

    def search(self, query, tms="last", criteria=None, offset=0, limit=20, detailed=False, sort=None):
        from tests.test_search_hostlogic_cpu import sort_detail
        from tests.toy_milli import query_terms
        ix, h = self.index, self.h
        crit, order = R.expand_sort_criteria(criteria if criteria is not None else ix.criteria, sort)
        keys, tables = [], []
        for field, asc in order:
            k, values = ix.order_keys(field, asc)
            keys.append(ma.DocKeys(h.ctx, np.array(k, dtype=np.uint32)))
            tables.append((field, asc, values))
        hits, cand = R.keyword_search_ranked(
            h.dict, h.pool, h.cb, query_terms(query, stop_words=ix.stop_words), crit,
            strategy=R.strategy_of(tms), offset=offset, limit=limit, detailed=detailed,
            searchable_fids=ix.searchable_fids, searchable_weights=[ix.weights[f] for f in ix.searchable_fids],
            max_weight=ix.max_weight, authorize_typos=ix.authorize_typos, min_one=ix.min_one, min_two=ix.min_two,
            order_keys=keys)
        return [(d, [sort_detail(s, tables) for s in sc]) for d, sc in hits], cand


def test_sort_rs_snapshots():
    import tests.test_search_gpu as G
    from tests.test_search_hostlogic_cpu import debug_score
    cases = [c for c in FIX["cases"] if c.get("sort") and not c.get("distinct") and not c.get("needs")
             and not FIX["indexes"][c["index"]].get("distinct")]
    assert len(cases) >= 5
    for case in cases:
        h = SortHarness(G.build_index(FIX["indexes"][case["index"]]))
        hits, _ = h.search(case["query"], tms=case["tms"], offset=case["offset"], limit=case["limit"],
                           detailed=case["detailed"], sort=case["sort"])
        assert [d for d, _ in hits] == case["ids"], case["src"]
        if case.get("scores"):
            assert "[" + "".join("[" + "".join(debug_score(s) + "," for s in sc) + "]," for _, sc in hits) + "]" == case["scores"]


def test_sort_rules_match_the_oracle_on_the_device():
    from oracle import oracle as O, ranking_oracle as RO
    import tests.test_search_gpu as G
    from tests.test_search_hostlogic_cpu import SORT_SETUPS, sortable_corpus
    from tests.toy_milli import ToyMilli
    index = ToyMilli(sortable_corpus(5, 250), searchable=["title", "body"])
    dic = O.Dictionary(index.words)

    def lookup(word, max_typos, is_prefix):
        one, two = O.typo_lookup(dic, word, max_typos, is_prefix)
        return [index.words[i] for i in one], [index.words[i] for i in two]
    h = SortHarness(index)
    for criteria, sort in SORT_SETUPS:
        for q in ["", "quick fox", "sun fl", "brwn fox jumps"]:
            for detailed, offset, limit in ((True, 0, 30), (False, 17, 9)):
                want_ids, want_sc, want_cand = RO.search(RO.Ctx(index, lookup), q, tms="last", criteria=criteria,
                                                         offset=offset, length=limit, detailed=detailed, sort=sort)
                hits, cand = h.search(q, criteria=criteria, offset=offset, limit=limit, detailed=detailed, sort=sort)
                assert [d for d, _ in hits] == want_ids, (criteria, sort, q, detailed, offset)
                assert [[tuple(s) for s in sc] for _, sc in hits] == [[G.oracle_score(s) for s in sc] for sc in want_sc]
                assert cand == len(want_cand)
