#!/usr/bin/env python
"""Regenerates tests/golden/*.json|npz.

Two kinds of fixtures:
  reference_literals.json  literals transcribed from the reference's OWN tests for this
                           path (file:line under /root/reference in every entry) — the
                           values that pin the oracle (tests/test_oracle_golden.py);
  oracle_*.npz             outputs of the CPU oracle (oracle/msi_oracle.c, pinned by the
                           literals above) on small seeded inputs; the -m gpu tests check
                           the HIP path against them through the C ABI, so parity does not
                           depend on the oracle being rebuilt identically on the GPU box.
The reference itself (Rust + un-vendored crates) cannot be built or imported here, so
no fixture is produced by running it.  Run from the repo root:  python tests/golden/make_fixtures.py
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

LITERALS = {
    "similarity": [  # _rankingScore = 1 - distance = (1 + cos)/2, f32
        {"q": [1, 1], "x": [2, 3], "score": 0.990290343761444, "src": "crates/meilisearch/tests/search/hybrid.rs:296-330"},
        {"q": [1, 1], "x": [1, 2], "score": 0.974341630935669, "src": "crates/meilisearch/tests/search/hybrid.rs:296-330"},
        {"q": [1, 1], "x": [1, 3], "score": 0.9472135901451112, "src": "crates/meilisearch/tests/search/hybrid.rs:296-330"},
        {"q": [1, 0], "x": [2, 3], "score": 0.7773500680923462, "src": "crates/meilisearch/tests/search/hybrid.rs:758"},
        {"q": [1, 0], "x": [1, 2], "score": 0.7236068248748779, "src": "crates/meilisearch/tests/search/hybrid.rs:758"},
        {"q": [1, 0], "x": [1, 3], "score": 0.6581138968467712, "src": "crates/meilisearch/tests/search/hybrid.rs:758"},
        {"q": [-0.5, 0.3, 0.85], "x": [0.1, 0.6, 0.8], "score": 0.890957772731781, "src": "crates/meilisearch/tests/similar/mod.rs:281-335"},
        {"q": [-0.5, 0.3, 0.85], "x": [0.6, 0.8, -0.2], "score": 0.39060014486312866, "src": "crates/meilisearch/tests/similar/mod.rs:281-335"},
        {"q": [-0.5, 0.3, 0.85], "x": [0.7, 0.7, -0.4], "score": 0.2819308042526245, "src": "crates/meilisearch/tests/similar/mod.rs:281-335"},
        {"q": [-0.5, 0.3, 0.85], "x": [0.8, 0.4, -0.5], "score": 0.1662663221359253, "src": "crates/meilisearch/tests/similar/mod.rs:281-335"},
    ],
    "distribution_shift": {"mean": 0.998, "sigma": 0.01,
                           "in": [0.990290343761444, 0.974341630935669, 0.9472135901451112],
                           "out": [0.19161224365234375, 1.1920928955078125e-07, 1.1920928955078125e-07],
                           "src": "crates/meilisearch/tests/search/hybrid.rs:540-568; crates/milli/src/vector/distribution.rs:103-130"},
    "tie_order": {"rows": [[0.1, 0.1], [-0.1, 0.1], [0.1, -0.1], [-0.1, -0.1]], "q": [1, -1],
                  "ids": [2, 0, 3, 1], "similarities": [1.0, 0.5, 0.5, 0.0],
                  "src": "crates/milli/src/search/new/tests/cutoff.rs:507-626"},
    "rank_global_score": [
        {"ranks": [[3, 3], [4, 4]], "score": "1.0000", "src": "crates/milli/src/search/new/tests/cutoff.rs:330-470 (Words 3/3, Typo 0 of 3)"},
        {"ranks": [[3, 3], [3, 4]], "score": "0.9167", "src": "same (Typo 1 of 3)"},
        {"ranks": [[3, 3], [2, 4]], "score": "0.8333", "src": "same (Typo 2 of 3)"},
        {"ranks": [[2, 3], [3, 3]], "score": "0.6667", "src": "same (Words 2/3, Typo 0 of 2)"},
    ],
    "typo_budget": [  # number_of_typos_allowed: thresholds count chars (5 / 9)
        {"word": "dogg", "budget": 0}, {"word": "doggy", "budget": 1}, {"word": "café!", "budget": 1},
        {"word": "собак", "budget": 1}, {"word": "doggydogg", "budget": 2},
        {"src": "crates/milli/src/search/new/query_term/parse_query.rs:204-225,408-478"},
    ],
    "typo_words": {  # end-to-end pins of the matcher semantics
        "dictionary": "the quick brown fox jumps over the lazy dog quickest quickly quack quickbrownfox zeal zealand zealot",
        "cases": [
            {"q": "quack", "typos": 1, "one_contains": ["quick"], "src": "crates/milli/src/search/new/tests/typo.rs:176-233 (replace)"},
            {"q": "quicest", "typos": 1, "one_contains": ["quickest"], "src": "typo.rs:176-233 (missing letter)"},
            {"q": "jummps", "typos": 1, "one_contains": ["jumps"], "src": "typo.rs:176-233 (extra letter)"},
            {"q": "zuickest", "typos": 2, "two_contains": ["quickest"], "one_excludes": ["quickest"],
             "src": "typo.rs:9 (a typo on the first letter counts as two typos)"},
            {"q": "zean", "typos": 1, "one_contains": ["zeal"], "src": "crates/milli/tests/search/typo_tolerance.rs:37-175"},
            {"q": "zealemd", "typos": 2, "two_contains": ["zealand"], "src": "crates/milli/tests/search/typo_tolerance.rs:37-175"},
        ],
    },
}


def main():
    with open(os.path.join(HERE, "reference_literals.json"), "w") as f:
        json.dump(LITERALS, f, indent=1, ensure_ascii=False)
    from meilisearch_amd import synth
    from oracle import oracle as orc
    orc.build()
    # vector k-NN
    n, dim, k = 4096, 64, 20
    rows = synth.make_embeddings(n, dim, seed=101)
    ids = (np.arange(n, dtype=np.uint32) * 5 + 2)
    qs = synth.make_embeddings(8, dim, seed=102)
    allowed = ids[np.random.default_rng(103).random(n) < 0.07]
    out_ids, out_dist, f_ids, f_dist = [], [], [], []
    from meilisearch_amd.vector_store import dense_filter
    fb, nb = dense_filter(allowed, nbits=int(ids.max()) + 1)
    for j in range(qs.shape[0]):
        a, b = orc.vs_topk(rows, ids, qs[j], k)
        out_ids.append(a)
        out_dist.append(b)
        a, b = orc.vs_topk(rows, ids, qs[j], k, fb, nb)
        f_ids.append(a)
        f_dist.append(b)
    np.savez_compressed(os.path.join(HERE, "oracle_vs_topk.npz"), seed_rows=101, seed_queries=102, seed_filter=103,
                        n=n, dim=dim, k=k, ids=np.stack(out_ids), dist=np.stack(out_dist),
                        filtered_ids=np.stack(f_ids), filtered_dist=np.stack(f_dist), allowed=allowed)
    # typo derivations
    words = synth.make_dictionary(3000, seed=201)
    concat, off = synth.flatten_words(words)
    odic = orc.Dictionary.from_flat(concat, off)
    queries = synth.make_typo_queries(words, 96, seed=202)
    ones, twos = [], []
    for w, b, p in queries:
        e1, e2 = orc.typo_lookup(odic, w, b, p)
        ones.append(e1.tolist())
        twos.append(e2.tolist())
    with open(os.path.join(HERE, "oracle_typo_lookup.json"), "w") as f:
        json.dump({"seed_dictionary": 201, "n_words": 3000, "seed_queries": 202,
                   "queries": [[w, int(b), bool(p)] for w, b, p in queries], "one": ones, "two": twos}, f)
    print("fixtures written to", HERE)


if __name__ == "__main__":
    main()
