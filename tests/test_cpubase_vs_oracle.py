"""The CPU baseline (oracle/msi_cpubase.c, FST-like work profile, threads) must
agree with the literal restatement (oracle/msi_oracle.c).  Runs without a GPU."""
import numpy as np

from meilisearch_amd import synth


def test_typo_lookup_equivalence(oracle, cpubase):
    words = synth.make_dictionary(6000, seed=3)
    # add long and unicode words
    words = sorted(set(words + ["internationalisationally", "антидисестаблишментарианизм", "ünïcödé",
                                "quick", "quack", "quickest", "quicklyquickly"]), key=lambda w: w.encode())
    concat, off = synth.flatten_words(words)
    dic = oracle.Dictionary.from_flat(concat, off)
    cpu = cpubase.CpuDictionary(concat, off)
    queries = synth.make_typo_queries(words, 300, seed=11)
    queries += [("quick", 1, False), ("quic", 1, True), ("zuickest", 2, False), ("ünïcodé", 1, False),
                ("internationalisationaly", 2, False), ("q", 1, True), ("ab", 2, True)]
    for caps in [(150, 50), (3, 2)]:
        got = cpu.lookup(queries, cap_one=caps[0], cap_two=caps[1], threads=4)
        for (w, b, p), (g1, g2) in zip(queries, got):
            e1, e2 = oracle.typo_lookup(dic, w, b, p, cap_one=caps[0], cap_two=caps[1])
            assert g1.tolist() == e1.tolist(), (w, b, p, caps)
            assert g2.tolist() == e2.tolist(), (w, b, p, caps)


def test_vector_scan_close_to_oracle(oracle, cpubase):
    rows = synth.make_embeddings(3000, 96, seed=5)
    ids = np.arange(3000, dtype=np.uint32) * 2 + 1
    scan = cpubase.CpuVectorScan(rows, ids)
    qs = synth.make_embeddings(5, 96, seed=6)
    d, s, c = scan.search(qs, 10, threads=3)
    for j in range(5):
        e_ids, e_dist = oracle.vs_topk(rows, ids, qs[j], 10)
        assert c[j] == 10
        # relaxed summation order: same docids unless a near-tie, distances within 1e-5
        assert np.allclose(np.sort(s[j]), np.sort(e_dist), atol=1e-5)
        assert len(set(d[j].tolist()) & set(e_ids.tolist())) >= 9
