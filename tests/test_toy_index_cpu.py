"""CPU check of the test harness itself: the toy index + host term derivation +
brute-force Words->Typo order reproduce the reference's snapshot literals when the
typo derivations come from the CPU oracle (so a GPU failure in test_rank_gpu.py is a
device bug, not a harness bug)."""
from toy_index import ToyIndex, brute_force_graph_order, brute_force_order
from test_rank_gpu import TYPO_RS_DOCS


def order(idx, orc, query, strategy_all=False, use_typo=True, **kw):
    odic = orc.Dictionary(idx.words)

    def lookup(w, b, p):
        o, t = orc.typo_lookup(odic, w, b, p)
        return o.tolist(), t.tolist()
    words = query.split()
    sets = [idx.term_sets(w, i == len(words) - 1, lookup, **kw) for i, w in enumerate(words)]
    return brute_force_order(idx.n_docs, sets, set(idx.docs), strategy_all, use_typo)


def test_snapshots_with_cpu_oracle(oracle):
    idx = ToyIndex(TYPO_RS_DOCS)
    got = order(idx, oracle, "the quick brown fox jumps over the lazy dog")
    assert [g[0] for g in got] == [0, 23, 7, 8, 9, 22, 10, 11, 1, 2, 12, 13, 4, 3, 5, 6, 21]   # typo.rs:476
    assert got[0][1:] == (9, 0, 9) and got[1][1:] == (9, 1, 9) and got[2][1:] == (8, 0, 8)
    got = order(idx, oracle, "network interconnection sunflower", True)
    assert [g[0] for g in got] == [16, 18, 17, 20, 15, 14]                                       # typo.rs:560
    assert [g[2] for g in got] == [0, 0, 1, 1, 2, 5] and all(g[3] == 5 for g in got)             # typo_bucketing-5.snap
    got = order(idx, oracle, "network interconnection sunflower", True, False)
    assert [g[0] for g in got] == [14, 15, 16, 17, 18, 20]                                       # typo.rs:533


def test_ngram_snapshots_with_cpu_oracle(oracle):
    idx = ToyIndex(TYPO_RS_DOCS)
    odic = oracle.Dictionary(idx.words)

    def lookup(w, b, p):
        o, t = oracle.typo_lookup(odic, w, b, p)
        return o.tolist(), t.tolist()

    def graph_order(query, strategy_all=False, use_typo=True):
        words = query.split()
        return brute_force_graph_order(idx.graph_nodes(words, lookup), len(words), set(idx.docs), strategy_all, use_typo)
    got = graph_order("network interconnection sun flower", True)
    assert [g[0] for g in got] == [17, 19, 16, 18, 20, 15]                           # typo.rs:579
    assert [g[2] for g in got] == [0, 0, 2, 2, 3, 4] and all(g[3] == 6 for g in got)  # typo_bucketing-8.snap
    got = graph_order("the quick brown fox jumps over the lazy dog")
    assert [g[0] for g in got] == [0, 23, 7, 8, 9, 22, 10, 11, 1, 2, 12, 13, 4, 3, 5, 6, 21]
    assert got[0][1:] == (9, 0, 9) and got[2][1:] == (8, 0, 8)


def test_toy_indexer_against_the_index_milli_wrote():
    """The reference ships one LMDB index written by milli (v1.12 upgrade test; tests/golden/index_blobs.json holds
    its databases): the toy indexer the ranking oracle runs on must produce the same word / exact-word / word-fid /
    word-position / field-word-count / word-pair-proximity databases from the same two documents — positions across
    array values (+8), bucketed positions, stop words keeping their position but not counted, numbers split at the
    dot, exact attributes going to exact_word_docids only.
    Known difference of that (v1.12-era) file, excluded below: its word-pair-proximity database still pairs the stop
    word "un"; today's extractor shares the word extractor's tokenizer
    (extract_word_pair_proximity_docids.rs:76-84), which drops stop words."""
    import json
    import os
    import unicodedata
    from tests.toy_milli import ToyMilli
    fx = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "index_blobs.json")))["index"]
    order = [fx["fields"][str(i)] for i in range(5)]

    def normalise(v):      # charabia's Latin normalisation of this corpus: lowercase + strip diacritics
        if isinstance(v, list):
            return [normalise(x) for x in v]
        if isinstance(v, str):
            return "".join(c for c in unicodedata.normalize("NFD", v.lower()) if not unicodedata.combining(c))
        return v
    docs = [{k: normalise(d[k]) for k in order} for d in fx["documents"]]
    toy = ToyMilli(docs, searchable=order, exact_attributes=fx["exact_attributes"], stop_words=fx["stop_words"])
    assert [toy.fields[n] for n in order] == [0, 1, 2, 3, 4]
    db = fx["databases"]
    assert {w: sorted(s) for w, s in toy.word_docids.items()} == {w: ids for w, ids in db["word_docids"]}
    assert {w: sorted(s) for w, s in toy.exact_word_docids.items()} == {w: ids for w, ids in db["exact_word_docids"]}
    assert {k: sorted(s) for k, s in toy.word_fid_docids.items()} == {(w, f): ids for w, f, ids in db["word_fid_docids"]}
    assert {k: sorted(s) for k, s in toy.word_position_docids.items()} == \
        {(w, p): ids for w, p, ids in db["word_position_docids"]}
    assert {k: sorted(s) for k, s in toy.fid_word_count.items()} == \
        {(f, c): ids for f, c, ids in db["field_id_word_count_docids"]}
    stop = set(fx["stop_words"])
    want = {(p, a, b): ids for p, a, b, ids in db["word_pair_proximity_docids"] if a not in stop and b not in stop}
    assert len(want) == 24
    assert {k: sorted(s) for k, s in toy.pair.items()} == want
