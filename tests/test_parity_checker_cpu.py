"""The full-size parity checker (oracle/parity.py) against the plain oracle on sizes the oracle scans whole:
the chunked candidate scheme must return exactly orc_vs_topk's answer, and must notice a wrong one."""
import numpy as np

from meilisearch_amd import synth
from oracle import parity


def test_chunked_checker_equals_the_oracle(oracle):
    rows = synth.make_embeddings(30000, 64, seed=1)
    q = synth.make_embeddings(4, 64, seed=2)
    ids = np.arange(30000, dtype=np.uint32) * 2 + 5
    chk = parity.TopkChecker(q, 20)
    for c0 in range(0, 30000, 7000):
        chk.add_chunk(ids[c0:c0 + 7000], rows[c0:c0 + 7000])
    gi = np.zeros((4, 20), np.uint32)
    gd = np.zeros((4, 20), np.float32)
    for j in range(4):
        gi[j], gd[j] = oracle.vs_topk(rows, ids, q[j], 20)
    v = chk.verdict(gi, gd, np.full(4, 20))
    assert v["mismatches"] == 0 and v["rows"] == 30000 and v["candidate_margin"] > 1e-4
    gi[2, 5], gi[2, 6] = gi[2, 6], gi[2, 5]
    assert chk.verdict(gi, gd, np.full(4, 20))["mismatches"] == 1
    gi[2, 5], gi[2, 6] = gi[2, 6], gi[2, 5]
    gd[1, 0] = np.nextafter(gd[1, 0], np.float32(2))
    assert chk.verdict(gi, gd, np.full(4, 20))["mismatches"] == 1


def test_typo_checker(oracle):
    words = synth.make_dictionary(3000, seed=3)
    concat, off = synth.flatten_words(words)
    dic = oracle.Dictionary.from_flat(concat, off)
    queries = synth.make_typo_queries(words, 40, seed=4)
    got = [oracle.typo_lookup(dic, *q) for q in queries]
    assert parity.check_typo_lookup(concat, off, queries, got, threads=4)["mismatches"] == 0
    got[7] = (np.append(got[7][0], np.uint32(1)), got[7][1])
    assert parity.check_typo_lookup(concat, off, queries, got, threads=4)["mismatches"] == 1
