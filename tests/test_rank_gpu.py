"""S3 parity: the device Words -> Typo bucket sort against (1) the reference's own
snapshot literals (crates/milli/src/search/new/tests/typo.rs) replayed on a toy index,
with the typo derivations coming from the device dictionary (S2 -> S3 end to end), and
(2) a brute-force per-document restatement on random corpora."""
import numpy as np
import pytest

import meilisearch_amd as ma
from meilisearch_amd import ranking as R
from toy_index import ToyIndex, brute_force_graph_order, brute_force_order

pytestmark = pytest.mark.gpu

# crates/milli/src/search/new/tests/typo.rs:27-148 (field "text" only; docs 24/25 use another field)
TYPO_RS_DOCS = {
    0: "the quick brown fox jumps over the lazy dog",
    1: "the quick brown foxes jump over the lazy dog",
    2: "the quick brown fax sends a letter to the dog",
    3: "the quickest brownest fox jumps over the laziest dog",
    4: "a fox doesn't quack, that crown goes to the duck.",
    5: "the quicker browner fox jumped over the lazier dog",
    6: "the extravagant fox skyrocketed over the languorous dog",
    7: "the quick brown fox jumps over the lazy",
    8: "the quick brown fox jumps over the",
    9: "the quick brown fox jumps over",
    10: "the quick brown fox jumps",
    11: "the quick brown fox",
    12: "the quick brown",
    13: "the quick",
    14: "netwolk interconections sunflawar",
    15: "network interconnections sunflawer",
    16: "network interconnection sunflower",
    17: "network interconnection sun flower",
    18: "network interconnection sunflowering",
    19: "network interconnection sun flowering",
    20: "network interconnection sunflowar",
    21: "the fast brownish fox jumps over the lackadaisical dog",
    22: "the quick brown fox jumps over the lackadaisical dog",
    23: "the quivk brown fox jumps over the lazy dog",
}


class Harness:
    def __init__(self, ctx, docs, n_slots=64):
        self.ctx = ctx
        self.idx = ToyIndex(docs)
        self.gdict = ma.GpuDictionary(ctx, words=self.idx.words)
        self.pool = ma.BitsPool(ctx, self.idx.n_docs, n_slots)

    def lookup(self, word, budget, is_prefix):
        one, two = self.gdict.lookup([(word, budget, is_prefix)])[0]
        return one.tolist(), two.tolist()

    def search(self, query, strategy_all=False, use_typo=True, offset=0, limit=100, **kw):
        words = query.split()
        sets = [self.idx.term_sets(w, i == len(words) - 1, self.lookup, **kw) for i, w in enumerate(words)]
        slot = 2
        terms = []
        for z, o, t, mc in sets:
            sl = []
            for s in (z, o, t):
                self.pool.set_from_docids(slot, np.array(sorted(s), dtype=np.uint32))
                sl.append(slot)
                slot += 1
            terms.append((sl[0], sl[1], sl[2], mc))
        self.pool.set_from_docids(0, np.array(sorted(self.idx.docs), dtype=np.uint32))   # universe
        got, cand = R.bucket_sort_words_typo(self.pool, terms, 0, 1, R.TERMS_ALL if strategy_all else R.TERMS_LAST,
                                             use_typo, offset, limit)
        exp = brute_force_order(self.idx.n_docs, sets, set(self.idx.docs), strategy_all, use_typo)
        assert cand == len(exp)
        assert got == exp[offset:offset + limit], (query, got[:8], exp[:8])
        return got


    def search_graph(self, query, strategy_all=False, use_typo=True, offset=0, limit=100, **kw):
        """Full query graph (terms + 2-grams + 3-grams)."""
        words = query.split()
        gn = self.idx.graph_nodes(words, self.lookup, **kw)
        slot, nodes = 2, []
        for first, last, z, o, t, mc in gn:
            sl = []
            for s in (z, o, t):
                if s:
                    self.pool.set_from_docids(slot, np.array(sorted(s), dtype=np.uint32))
                    sl.append(slot)
                    slot += 1
                else:
                    sl.append(None)
            nodes.append((first, last, sl[0], sl[1], sl[2], mc))
        self.pool.set_from_docids(0, np.array(sorted(self.idx.docs), dtype=np.uint32))
        got, cand = R.bucket_sort_query_graph(self.pool, nodes, len(words), 0, 1,
                                              R.TERMS_ALL if strategy_all else R.TERMS_LAST, use_typo, offset, limit)
        exp = brute_force_graph_order(gn, len(words), set(self.idx.docs), strategy_all, use_typo)
        assert cand == len(exp)
        assert got == exp[offset:offset + limit], (query, got[:8], exp[:8])
        return got


def test_reference_snapshots_with_ngrams(ctx):
    h = Harness(ctx, TYPO_RS_DOCS, n_slots=128)
    # typo.rs:576-594 + snapshots/…typo_bucketing-8.snap: "sun flower" also matches the 2-gram
    # "sunflower" at base cost 2 (criteria [Typo], strategy All)
    got = h.search_graph("network interconnection sun flower", strategy_all=True)
    assert [g[0] for g in got] == [17, 19, 16, 18, 20, 15]
    assert [g[2] for g in got] == [0, 0, 2, 2, 3, 4] and all(g[3] == 6 for g in got)
    # the chain-only snapshots are unchanged by the n-gram nodes
    got = h.search_graph("the quick brown fox jumps over the lazy dog")
    assert [g[0] for g in got] == [0, 23, 7, 8, 9, 22, 10, 11, 1, 2, 12, 13, 4, 3, 5, 6, 21]
    assert got[0][1:] == (9, 0, 9) and got[1][1:] == (9, 1, 9) and got[2][1:] == (8, 0, 8)
    got = h.search_graph("network interconnection sunflower", strategy_all=True)
    assert [g[0] for g in got] == [16, 18, 17, 20, 15, 14] and [g[2] for g in got] == [0, 0, 1, 1, 2, 5]
    # typo.rs:432-459 test_ngram_typos: a 2gram may carry one typo, a 3gram none
    assert [g[0] for g in h.search_graph("the extra lagant fox skyrocketed over the languorous dog", True, False)] == [6]
    assert [g[0] for g in h.search_graph("the ex tra lagant fox skyrocketed over the languorous dog", True, False)] == []


@pytest.mark.parametrize("seed,n_docs,n_terms", [(11, 400, 2), (12, 3000, 4), (13, 50000, 7), (14, 999, 10)])
def test_random_query_graph_vs_brute_force(ctx, seed, n_docs, n_terms):
    rng = np.random.default_rng(seed)
    universe = set(np.nonzero(rng.random(n_docs) < 0.95)[0].tolist())
    gn = []
    for last in range(n_terms):
        for size in (1, 2, 3):
            first = last - size + 1
            if first < 0 or (size > 1 and rng.random() < 0.3):
                continue
            dens = (0.6, 0.3, 0.2) if size == 1 else (0.15, 0.1, 0.05)
            lv = [set(np.nonzero(rng.random(n_docs) < p)[0].tolist()) for p in dens]
            gn.append((first, last, lv[0], lv[1], lv[2], int(rng.integers(0, 3))))
    pool = ma.BitsPool(ctx, n_docs, 3 * len(gn) + 2)
    slot, nodes = 2, []
    for first, last, z, o, t, mc in gn:
        sl = []
        for s in (z, o, t):
            pool.set_from_docids(slot, np.array(sorted(s), dtype=np.uint32))
            sl.append(slot)
            slot += 1
        nodes.append((first, last, sl[0], sl[1], sl[2], mc))
    pool.set_from_docids(0, np.array(sorted(universe), dtype=np.uint32))
    for strategy_all in (False, True):
        for use_typo in (True, False):
            exp = brute_force_graph_order(gn, n_terms, universe, strategy_all, use_typo)
            for off, lim in [(0, 60), (23, 500)]:
                got, cand = R.bucket_sort_query_graph(pool, nodes, n_terms, 0, 1,
                                                      R.TERMS_ALL if strategy_all else R.TERMS_LAST, use_typo, off, lim)
                assert cand == len(exp)
                assert got == exp[off:off + lim], (strategy_all, use_typo, off, lim)


def test_buckets_and_materialise_for_rules_after_typo(ctx):
    # the hand-over to the rules that stay on the CPU path: bucket list + bucket bitmaps
    h = Harness(ctx, TYPO_RS_DOCS, n_slots=128)
    words = "the quick brown fox jumps over the lazy dog".split()
    gn = h.idx.graph_nodes(words, h.lookup)
    slot, nodes = 2, []
    for first, last, z, o, t, mc in gn:
        sl = []
        for s in (z, o, t):
            if s:
                h.pool.set_from_docids(slot, np.array(sorted(s), dtype=np.uint32))
                sl.append(slot)
                slot += 1
            else:
                sl.append(None)
        nodes.append((first, last, sl[0], sl[1], sl[2], mc))
    h.pool.set_from_docids(0, np.array(sorted(h.idx.docs), dtype=np.uint32))
    exp = brute_force_graph_order(gn, len(words), set(h.idx.docs), False, True)
    buckets = R.rank_buckets(h.pool, nodes, len(words), 0, 1)
    assert sum(b[3] for b in buckets) == len(exp)
    flat = []
    for mw, tc, mt, cnt in buckets:
        R.rank_materialise(h.pool, nodes, len(words), 0, 100, mw, tc)
        ids = h.pool.to_docids(100).tolist()
        assert len(ids) == cnt
        flat += [(d, mw, tc, mt) for d in ids]
    assert flat == exp
    assert buckets[0][:3] == (9, 0, 9) and buckets[1][:3] == (9, 1, 9)


def test_batched_rank_equals_single_queries(ctx):
    # 13 random query graphs over one pool, ranked in ONE batch: every row must equal the
    # single-query call (which is checked against the brute force above)
    rng = np.random.default_rng(21)
    n_docs, nq = 30000, 13
    per_query_slots = 3 * 12 + 1 + 4
    pool = ma.BitsPool(ctx, n_docs, nq * per_query_slots + 2)
    queries, graphs = [], []
    slot = 0
    for q in range(nq):
        n_terms = int(rng.integers(1, 6))
        uni, scratch = slot, slot + 1
        slot += 5
        pool.set_from_docids(uni, np.nonzero(rng.random(n_docs) < 0.9)[0].astype(np.uint32))
        nodes = []
        for last in range(n_terms):
            for size in (1, 2, 3):
                first = last - size + 1
                if first < 0 or (size > 1 and rng.random() < 0.5):
                    continue
                dens = (0.5, 0.2, 0.1) if size == 1 else (0.05, 0.03, 0.02)
                sl = []
                for p_ in dens:
                    if rng.random() < 0.15:
                        sl.append(None)
                        continue
                    pool.set_from_docids(slot, np.nonzero(rng.random(n_docs) < p_)[0].astype(np.uint32))
                    sl.append(slot)
                    slot += 1
                nodes.append((first, last, sl[0], sl[1], sl[2], int(rng.integers(0, 3))))
        queries.append((nodes, n_terms, uni, scratch))
    batch = R.RankBatch(pool, queries)
    for strategy in (R.TERMS_LAST, R.TERMS_ALL):
        for use_typo in (True, False):
            for off, lim in [(0, 20), (7, 300), (0, 1)]:
                got, cand = batch.run(strategy, use_typo, off, lim).rows()
                for q, (nodes, n_terms, uni, scratch) in enumerate(queries):
                    e, ec = R.bucket_sort_query_graph(pool, nodes, n_terms, uni, scratch, strategy, use_typo, off, lim)
                    assert cand[q] == ec and got[q] == e, (q, strategy, use_typo, off, lim)


def test_reference_snapshots_typo_rs(ctx):
    h = Harness(ctx, TYPO_RS_DOCS)
    # typo.rs:462-516: criteria [Typo] (Words is inserted implicitly, search/new/mod.rs:536-551), strategy Last
    got = h.search("the quick brown fox jumps over the lazy dog")
    assert [g[0] for g in got] == [0, 23, 7, 8, 9, 22, 10, 11, 1, 2, 12, 13, 4, 3, 5, 6, 21]
    # snapshots/…typo_ranking_rule_not_preceded_by_words_ranking_rule-2.snap (first entries)
    assert got[0][1:] == (9, 0, 9) and got[1][1:] == (9, 1, 9) and got[2][1:] == (8, 0, 8) and got[3][1:] == (7, 0, 7)
    # typo.rs:521-594 test_typo_bucketing: criteria [Words] then [Typo], strategy All
    got = h.search("network interconnection sunflower", strategy_all=True, use_typo=False)
    assert [g[0] for g in got] == [14, 15, 16, 17, 18, 20]
    got = h.search("network interconnection sunflower", strategy_all=True)
    assert [g[0] for g in got] == [16, 18, 17, 20, 15, 14]
    # snapshots/…typo_bucketing-5.snap: typo counts 0,0,1,1,2,5 of max 5
    assert [g[2] for g in got] == [0, 0, 1, 1, 2, 5] and all(g[3] == 5 for g in got)
    # typo.rs:176-233 test_default_typo (criteria [Words], strategy All)
    assert [g[0] for g in h.search("the quick brown fox jumps over the lazy dog", True, False)] == [0, 23]
    assert [g[0] for g in h.search("the quack brown fox jumps over the lazy dog", True, False)] == [0]
    assert [g[0] for g in h.search("the quicest brownest fox jummps over the laziest dog", True, False)] == [3]
    # typo.rs:150-174 test_no_typo
    assert [g[0] for g in h.search("the quick brown fox jumps over the lazy dog", True, False,
                                   authorize_typos=False)] == [0]
    # typo.rs:252-323 test_typo_exact_word
    ew = ("quick", "quack", "sunflower")
    assert [g[0] for g in h.search("the quick brown fox jumps over the lazy dog", True, False, exact_words=ew)] == [0]
    assert [g[0] for g in h.search("the quack brown fox jumps over the lazy dog", True, False, exact_words=ew)] == []
    assert [g[0] for g in h.search("network interconnection sunflower", True, False, exact_words=ew)] == [16, 17, 18]


def test_offset_limit_and_pages(ctx):
    h = Harness(ctx, TYPO_RS_DOCS)
    full = h.search("the quick brown fox jumps over the lazy dog")
    for off, lim in [(0, 1), (1, 3), (2, 5), (5, 100), (16, 4), (17, 4), (3, 0)]:
        assert h.search("the quick brown fox jumps over the lazy dog", offset=off, limit=lim) == full[off:off + lim]


@pytest.mark.parametrize("seed,n_docs,n_terms", [(1, 300, 3), (2, 5000, 5), (3, 70000, 10), (4, 64, 1)])
def test_random_corpus_vs_brute_force(ctx, seed, n_docs, n_terms):
    rng = np.random.default_rng(seed)
    pool = ma.BitsPool(ctx, n_docs, 3 * n_terms + 2)
    universe = set(np.nonzero(rng.random(n_docs) < 0.9)[0].tolist())
    sets, terms, slot = [], [], 2
    for i in range(n_terms):
        mc = int(rng.integers(0, 3))
        lv = [set(np.nonzero(rng.random(n_docs) < p)[0].tolist()) for p in (0.5, 0.3, 0.2)]
        if rng.random() < 0.2:
            lv[1] = set()
        sets.append((lv[0], lv[1], lv[2], mc))
        sl = []
        for s in lv:
            if s or rng.random() < 0.5:
                pool.set_from_docids(slot, np.array(sorted(s), dtype=np.uint32))
                sl.append(slot)
            else:
                sl.append(None)      # MSI_NO_SLOT = empty set
            slot += 1
        terms.append((sl[0], sl[1], sl[2], mc))
    pool.set_from_docids(0, np.array(sorted(universe), dtype=np.uint32))
    for strategy_all in (False, True):
        for use_typo in (True, False):
            exp = brute_force_order(n_docs, sets, universe, strategy_all, use_typo)
            for off, lim in [(0, 50), (37, 200), (max(0, len(exp) - 5), 10)]:
                got, cand = R.bucket_sort_words_typo(pool, terms, 0, 1, R.TERMS_ALL if strategy_all else R.TERMS_LAST,
                                                     use_typo, off, lim)
                assert cand == len(exp)
                assert got == exp[off:off + lim], (strategy_all, use_typo, off, lim)


def test_errors(ctx):
    pool = ma.BitsPool(ctx, 100, 4)
    with pytest.raises(ma.MsiError):
        R.bucket_sort_words_typo(pool, [], 0, 1)
    with pytest.raises(ma.MsiError):
        R.bucket_sort_words_typo(pool, [(2, None, None, 0)] * 11, 0, 1)
    with pytest.raises(ma.MsiError):
        R.bucket_sort_words_typo(pool, [(9, None, None, 0)], 0, 1)


def test_hybrid_search_end_to_end(ctx, oracle):
    """Search::execute_hybrid (search/hybrid.rs:264-366) on a toy index, every stage on the
    product path: keyword = device dictionary -> device bucket sort -> Rank::global_score;
    semantic = device k-NN -> msi_vector_sort; then msi_hybrid_merge.  Checked against the
    same pipeline built from the CPU oracle + the literal Python restatements."""
    from test_hybrid_cpu import py_merge
    h = Harness(ctx, TYPO_RS_DOCS)
    rng = np.random.default_rng(7)
    dim = 16
    ids = np.array(sorted(TYPO_RS_DOCS), dtype=np.uint32)
    emb = rng.standard_normal((ids.size, dim)).astype(np.float32)
    store = ma.GpuStore(ctx, dim)
    store.upload(ids, emb)
    qv = rng.standard_normal(dim).astype(np.float32)
    limit = 8
    # keyword leg (criteria [Words, Typo], strategy Last)
    kw = h.search("the quick brown fox", limit=limit)
    kw_hits = [(d, [ma.scoring.rank_global_score([(w, 4), (mt + 1 - t, mt + 1)])]) for d, w, t, mt in kw]
    # semantic leg
    d, s, c = store.search(qv[None, :], limit)
    vd, vs = ma.scoring.vector_sort(d[0, :c[0]], s[0, :c[0]], 0, limit)
    e_ids, e_dist = oracle.vs_topk(emb, ids, qv, limit)
    assert vd.tolist() == e_ids.tolist()
    assert vs.tolist() == [float(np.float32(1.0) - x) for x in e_dist]
    v_hits = [(int(a), [float(b)]) for a, b in zip(vd, vs)]
    for ratio in (0.0, 0.2, 0.5, 0.8, 1.0):
        got = ma.scoring.hybrid_merge(v_hits, kw_hits, ratio, 0, limit)
        assert got == py_merge(v_hits, kw_hits, ratio, 0, limit), ratio
        assert len(got[0]) == limit and len({x for x, _ in got[0]}) == limit
    assert ma.scoring.hybrid_merge(v_hits, kw_hits, 1.0, 0, limit)[1] == limit      # all semantic
    assert [x for x, _ in ma.scoring.hybrid_merge(v_hits, kw_hits, 0.0, 0, 4)[0]] == [k_[0] for k_ in kw[:4]]


def test_keyword_search_product_path(ctx):
    """msi_keyword_search: tokens -> query graph -> batched device derivations -> posting sets
    decoded on the device -> device bucket sort, all inside libmsi; the index is reached
    through msi_index_vtable callbacks that return CboRoaringBitmap bytes.  Same reference
    snapshots as above (typo.rs), plus the harness' brute-force order."""
    h = Harness(ctx, TYPO_RS_DOCS, n_slots=128)
    cb = R.IndexCallbacks(h.idx)

    def run(query, strategy_all=False, use_typo=True, **kw):
        got, cand = R.keyword_search(h.gdict, h.pool, cb, query.split(), True,
                                     R.TERMS_ALL if strategy_all else R.TERMS_LAST, use_typo, 0, 100, **kw)
        words = query.split()
        gn = h.idx.graph_nodes(words, h.lookup, h.idx.exact_words, kw.get("authorize_typos", True))
        exp = brute_force_graph_order(gn, len(words), set(h.idx.docs), strategy_all, use_typo)
        assert cand == len(exp) and got == exp, (query, got[:6], exp[:6])
        return got
    got = run("the quick brown fox jumps over the lazy dog")
    assert [g[0] for g in got] == [0, 23, 7, 8, 9, 22, 10, 11, 1, 2, 12, 13, 4, 3, 5, 6, 21]     # typo.rs:476
    assert got[0][1:] == (9, 0, 9) and got[1][1:] == (9, 1, 9) and got[2][1:] == (8, 0, 8)
    got = run("network interconnection sunflower", True)
    assert [g[0] for g in got] == [16, 18, 17, 20, 15, 14] and [g[2] for g in got] == [0, 0, 1, 1, 2, 5]
    got = run("network interconnection sun flower", True)
    assert [g[0] for g in got] == [17, 19, 16, 18, 20, 15] and [g[2] for g in got] == [0, 0, 2, 2, 3, 4]
    assert [g[0] for g in run("the quack brown fox jumps over the lazy dog", True, False)] == [0]
    assert [g[0] for g in run("the quicest brownest fox jummps over the laziest dog", True, False)] == [3]
    assert [g[0] for g in run("the quick brown fox jumps over the lazy dog", True, False, authorize_typos=False)] == [0]
    h.idx.exact_words = ("quick", "quack", "sunflower")                                            # typo.rs:252-323
    assert [g[0] for g in run("the quack brown fox jumps over the lazy dog", True, False)] == []
    assert [g[0] for g in run("network interconnection sunflower", True, False)] == [16, 17, 18]
    h.idx.exact_words = ()
    # a filtered universe (the `filter` of a search arrives as a bitmap)
    from toy_index import cbo_bytes
    uni = [d for d in TYPO_RS_DOCS if d % 2 == 0]
    got, cand = R.keyword_search(h.gdict, h.pool, cb, "the quick brown fox".split(), True, R.TERMS_LAST, True, 0, 100,
                                 universe_cbo=cbo_bytes(uni))
    assert got and all(g[0] % 2 == 0 for g in got)
