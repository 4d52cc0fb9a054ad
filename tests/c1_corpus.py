"""BASELINE.json configs[0] (C1): keyword-only search, plumbing.  TEST INFRASTRUCTURE (used by bench.py --config c1 and
tests): the reference's movies.json workload (workloads/search/movies.json: queries "", "Batman returns", "the", "t",
limit 100) needs a remote dataset, so the corpus is restated synthetically as SURVEY §8 d prescribes — 32 k documents,
title 3-6 words, overview 20-60 words, Zipf(1.07) over a 60 k-word vocabulary, seed 42 — and indexed by the toy
indexer the ranking replays run on (tests/toy_milli.py, pinned to milli's own databases).  Queries: the workload's four
shapes (placeholder, two words, the most frequent word, a one-letter prefix) + sampled 1-3 word queries with 0-2 edits.

CPU side (the reported baseline, kind "port"): oracle/ranking_oracle.py end to end (typo derivations from
oracle/msi_oracle.c).  Product side: msi_keyword_search_ranked on the same index; every query's hits are compared."""
import json
import os
import statistics
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def make_vocabulary(n_words, seed):
    from meilisearch_amd import synth
    return synth.make_dictionary(n_words, seed=seed, digits=0.0, two_byte=0.0, mean_len=7.0, sd_len=2.0, max_len=14)


def make_corpus(n_docs=32_000, vocab_size=60_000, seed=42):
    rng = np.random.default_rng(seed)
    vocab = make_vocabulary(vocab_size, seed)
    order = rng.permutation(vocab_size)                       # frequency rank -> word
    p = 1.0 / np.arange(1, vocab_size + 1) ** 1.07
    p /= p.sum()
    docs = []
    tl = rng.integers(3, 7, n_docs)
    ol = rng.integers(20, 61, n_docs)
    draws = rng.choice(vocab_size, size=int(tl.sum() + ol.sum()), p=p)
    pos = 0
    for i in range(n_docs):
        t = " ".join(vocab[order[j]] for j in draws[pos:pos + tl[i]])
        pos += tl[i]
        o = " ".join(vocab[order[j]] for j in draws[pos:pos + ol[i]])
        pos += ol[i]
        docs.append({"id": i, "title": t, "overview": o})
    frequent = [vocab[order[j]] for j in range(2000)]
    return docs, frequent


def make_queries(frequent, n_sampled, seed=7):
    from meilisearch_amd import synth
    rng = np.random.default_rng(seed)
    # the four shapes of workloads/search/movies.json
    qs = ["", f"{frequent[40]} {frequent[90]}", frequent[0], frequent[0][0]]
    for _ in range(n_sampled):
        n = int(rng.integers(1, 4))
        ws = [synth.edit_word(frequent[int(rng.integers(0, len(frequent)))], int(rng.integers(0, 3)), rng) for _ in range(n)]
        qs.append(" ".join(ws))
    return qs


def run(args, env):
    """bench.py --config c1."""
    from oracle import oracle as O
    from oracle import ranking_oracle as RO
    from tests.toy_milli import ToyMilli, query_terms
    ma = env.ma
    from meilisearch_amd import ranking as R
    n_docs = args.rows or 32_000
    limit = 100
    t0 = time.time()
    docs, frequent = make_corpus(n_docs)
    index = ToyMilli(docs, searchable=["title", "overview"])
    build_s = time.time() - t0
    queries = make_queries(frequent, args.queries or 60)
    dic = O.Dictionary(index.words)

    def lookup(word, max_typos, is_prefix):
        one, two = O.typo_lookup(dic, word, max_typos, is_prefix)
        return [index.words[i] for i in one], [index.words[i] for i in two]
    # ---- CPU oracle end to end ----------------------------------------------------------------------------
    cpu_lat, expected = [], []
    for q in queries:
        s0 = time.perf_counter()
        out = RO.search(RO.Ctx(index, lookup), q, tms="last", offset=0, length=limit)
        cpu_lat.append((time.perf_counter() - s0) * 1e3)
        expected.append(list(out[0]))   # (ids, score details, all_candidates)
    # ---- product ------------------------------------------------------------------------------------------
    gdict = ma.GpuDictionary(env.ctx, [w.encode() for w in index.words])
    pool = ma.BitsPool(env.ctx, max(index.n_docs, 1), 1024)
    cb = R.IndexCallbacks(index)

    def product(q):
        hits, _ = R.keyword_search_ranked(
            gdict, pool, cb, query_terms(q, stop_words=index.stop_words), index.criteria, strategy=R.TERMS_LAST, offset=0,
            limit=limit, searchable_fids=index.searchable_fids, searchable_weights=[index.weights[f] for f in index.searchable_fids],
            max_weight=index.max_weight, authorize_typos=index.authorize_typos, min_one=index.min_one, min_two=index.min_two)
        return [d for d, _ in hits]
    for q in queries[:4]:
        product(q)                                   # warm the adapter's posting cache like the oracle's
    got, lat = [], []
    t0 = time.perf_counter()
    for _ in range(max(1, args.steps // 10)):
        got = []
        for q in queries:
            s0 = time.perf_counter()
            got.append(product(q))
            lat.append((time.perf_counter() - s0) * 1e3)
    elapsed = time.perf_counter() - t0
    mism = [q for q, g, e in zip(queries, got, expected) if g != e]
    return {
        "metric": "keyword-only search queries/sec (C1 plumbing: 32k-document synthetic corpus, default criteria, limit 100)",
        "value": round(len(lat) / elapsed, 2), "unit": "queries/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(elapsed / len(lat) * 1e3, 4), "p50_latency_ms": round(statistics.median(lat), 4),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u64 docid sets (integer)",
        "data": "synthetic corpus restating workloads/search/movies.json (SURVEY §8 d): Zipf(1.07) over 60k words, seed 42",
        "config": {"workload": f"C1: {n_docs} documents, {len(queries)} queries (4 workload shapes + sampled 1-3 words with 0-2 edits), "
                               "one caller thread through the Python vtable adapter (ctypes callbacks: plumbing, not the serving path)",
                   "index_build_seconds": round(build_s, 1), "dictionary_words": len(index.words)},
        "roofline": {"kernel": "vm_kernel (docid-set command lists)", "bound": "hbm", "achieved": None, "peak": 8000.0,
                     "unit": "GB/s", "frac": None, "traffic": None,
                     "note": "plumbing config: 4 KB sets, latency-bound by construction (SURVEY §8 d: no GPU number asked)"},
        "cpu_baseline": {"value": round(len(queries) / (sum(cpu_lat) / 1e3), 2), "unit": "queries/s", "cores": 1, "kind": "port",
                         "sample": f"oracle/ranking_oracle.py + oracle/msi_oracle.c end to end over the same {len(queries)} queries "
                                   f"(pure Python sets, one thread); p50 {statistics.median(cpu_lat):.2f} ms"},
        "parity": {"checked_queries": len(queries), "mismatches": len(mism), "first_mismatches": mism[:4],
                   "checker": "oracle/ranking_oracle.py (pinned to the reference's 108 snapshot searches): docid lists identical"},
    }


if __name__ == "__main__":
    t0 = time.time()
    docs, frequent = make_corpus(int(sys.argv[1]) if len(sys.argv) > 1 else 32_000)
    print("corpus", time.time() - t0)
    from tests.toy_milli import ToyMilli
    t0 = time.time()
    ix = ToyMilli(docs, searchable=["title", "overview"])
    print("index", time.time() - t0, len(ix.words))
    print(make_queries(frequent, 5))
