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
