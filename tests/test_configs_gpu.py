"""Parity at BASELINE.json's FULL sizes, on the device, through the C ABI.

  C2  1 M x 384 f32, cosine top-20 (+ a 10 % candidate filter)         store.rs:1036-1093
  C3  2 M-term dictionary, >= 2 048 query words, 30 % prefix, 1/2 typos  compute_derivations.rs:75-168
  C4  10 M x 768 f32, cosine top-20, one full sweep of 48 queries        store.rs:1036-1093
  C5  one GPU's shard: 12.5 M x 1024 bf16 rows, 1 % filter, k = 1000     (bf16 = build-side storage: the oracle
      runs the reference arithmetic on the bf16-rounded rows)

Checker: oracle/parity.py (the scalar oracle ranks the candidates of the multi-threaded CPU scan; the typo
derivations of EVERY query word go through the oracle's literal loops).  Bit-exact bar: docids identical in
order, f32 distances bit-identical, derivation index lists identical.  Scale-specific failure modes these sizes
exercise and the small tests cannot: 32-bit tile/row index arithmetic at 10 M x 48 tiles, survivor-list capacity,
sample-threshold statistics at N = 1e7, dictionary segments > 2^16 words, k' = 1250 selection over 12.5 M rows.
"""
import numpy as np
import pytest

import meilisearch_amd as ma
from meilisearch_amd import synth

pytestmark = pytest.mark.gpu


def _torch_dev():
    import torch
    return torch, torch.device("cuda", 0)


def _check_store(ctx, n, d, k, nq, storage="f32", filter_density=None, chunk=1_000_000, extra=64):
    from oracle import parity
    torch, dev = _torch_dev()
    rows_t = synth.device_rows(n, d, dev, seed=1234)
    ids_t = torch.arange(n, dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    store = ma.GpuStore(ctx, d, storage=storage)
    store.upload_device(ids_t, rows_t)
    assert len(store) == n
    q = synth.device_queries(nq, d, dev, seed=5678).cpu().numpy()
    fb, nb, allowed_t = None, 0, None
    if filter_density is not None:
        fb = synth.random_bitset_words(n, filter_density, seed=31)
        nb = n
        allowed = np.nonzero(np.unpackbits(fb.view(np.uint8), bitorder="little")[:n])[0]
        allowed_t = torch.from_numpy(allowed).to(dev)
    got_ids, got_dist, got_cnt = store.search(q, k, fb, nb)        # the host entry point: exhaustive reruns included
    # size-independent properties first: ascending (distance, docid), no duplicates, counts
    for j in range(nq):
        c = int(got_cnt[j])
        assert c == min(k, n if allowed_t is None else int(allowed_t.numel()))
        dj, ij = got_dist[j, :c], got_ids[j, :c].astype(np.int64)
        assert (np.diff(dj) >= 0).all()
        ties = np.diff(dj) == 0
        assert (np.diff(ij)[ties] > 0).all()
        assert np.unique(ij).size == c
    chk = parity.TopkChecker(q, k, extra=extra)
    if allowed_t is None:
        for c0 in range(0, n, chunk):
            c1 = min(n, c0 + chunk)
            rows = rows_t[c0:c1].cpu().numpy()
            if storage == "bf16":
                rows = synth.round_to_bf16(rows)
            chk.add_chunk(np.arange(c0, c1, dtype=np.uint32), rows)
    else:
        for c0 in range(0, int(allowed_t.numel()), chunk):
            sel = allowed_t[c0:c0 + chunk]
            rows = rows_t[sel].cpu().numpy()
            if storage == "bf16":
                rows = synth.round_to_bf16(rows)
            chk.add_chunk(sel.cpu().numpy().astype(np.uint32), rows)
        chk.rows_seen = int(allowed_t.numel())
    v = chk.verdict(got_ids, got_dist, got_cnt)
    assert v["mismatches"] == 0, v
    assert v["candidate_margin"] is None or v["candidate_margin"] > 2e-6, v   # the CPU candidate set was wide enough
    stats = store.stats()
    store.close()
    del rows_t
    torch.cuda.empty_cache()
    return v, stats


def test_c2_1m_x_384_top20(ctx):
    v, _ = _check_store(ctx, 1_000_000, 384, 20, 48)
    assert v["checked_queries"] == 48 and v["rows"] == 1_000_000


def test_c2_with_10pct_filter(ctx):
    v, _ = _check_store(ctx, 1_000_000, 384, 20, 16, filter_density=0.1)
    assert v["checked_queries"] == 16


def test_c4_10m_x_768_top20(ctx):
    v, stats = _check_store(ctx, 10_000_000, 768, 20, 48)
    assert v["checked_queries"] == 48 and v["rows"] == 10_000_000
    assert stats["bytes_per_tile"] == 16 * 768 * 4


def test_c5_shard_bf16_filtered_k1000(ctx):
    v, _ = _check_store(ctx, 12_500_000, 1024, 1000, 16, storage="bf16", filter_density=0.01)
    assert v["checked_queries"] == 16 and v["k"] == 1000


@pytest.mark.parametrize("density", [0.001, 0.1])
def test_c5_shard_bf16_filtered_k1000_other_densities(ctx, density):
    """BASELINE.md C5's other two filter densities at the shard's full size: 0.1 % — 12 433 allowed rows, a sweep
    of ~1 500 gathered items — and 10 % — 1.25 M allowed rows, compacted 8:1 against the tiles that hold them.  k = 1 000
    starts at the level that sweeps the stored rows (the int8 level cannot prove that many neighbours: vs_first_level)."""
    v, stats = _check_store(ctx, 12_500_000, 1024, 1000, 8, storage="bf16", filter_density=density)
    assert v["checked_queries"] == 8 and v["k"] == 1000
    assert stats["i8_sweeps"] == 0


def test_c3_2m_term_dictionary(ctx):
    from oracle import parity
    words = synth.make_dictionary(2_000_000, seed=99)
    concat, off = synth.flatten_words(words)
    g = ma.GpuDictionary(ctx, concat=concat, offsets=off)
    queries = synth.make_typo_queries(words, 2048, seed=7)
    assert sum(1 for _, _, p in queries if p) > 500          # ~30 % prefix
    assert sum(1 for _, b, _ in queries if b == 2) > 300
    got = g.lookup(queries)
    v = parity.check_typo_lookup(concat, off, queries, got)
    assert v["mismatches"] == 0, v
    assert v["checked_words"] == 2048
    # one batch of 8192 (the largest BASELINE batch) against the same answers: batching must not change results
    more = synth.make_typo_queries(words, 8192, seed=7)
    assert more[:2048] == queries
    got2 = g.lookup(more)
    for (a1, a2), (b1, b2) in zip(got, got2[:2048]):
        assert a1.tolist() == b1.tolist() and a2.tolist() == b2.tolist()
    g.close()


@pytest.mark.parametrize("run_containers", [False, True], ids=["array+bitmap", "with-run-containers"])
def test_c4_keyword_leg(ctx, monkeypatch, run_containers):
    """The keyword leg of the headline step at its own size: msi_keyword_search_ranked (7 default criteria, detailed
    scores, 3-term queries with typo and prefix derivations, 16 caller threads sharing command-list launches) over the
    synthetic 10 M-document inverted index of tools/ranked_bench.cpp, against oracle/ranking_oracle.py reading the same
    stored posting bytes (bucket_sort.rs:23-343, graph_based_ranking_rule.rs:97-378).  What the toy corpora cannot show:
    bitmap containers, 153 chunks of 65 536 documents, posting-cache reuse across searches, many searches in flight."""
    import ctypes as C
    from oracle import parity
    from oracle import synth_index as SI
    import os
    if run_containers:      # every third key of the index run-encoded (cookie 12347): the third container form, also at 10 M
        monkeypatch.setenv("RB_RUN_CONTAINERS", "1")
    n_docs, n_queries, limit = 10_000_000, 24 if run_containers else 48, 20
    if os.environ.get("MSI_RUNNER_SO"):        # the CPU tier's emulated kernels (tests/emu): same path, 5 chunks of documents
        n_docs, n_queries = 300_000, 12
    lib = SI.runner_lib()
    lib.rb_attach.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint64]
    h = lib.rb_create(n_docs, 200_000)
    try:
        assert lib.rb_attach(h, ctx.handle, 16, 1024, 2048) == 0
        lib.rb_prepare_queries(h, n_queries, 3, 4242)
        chk = parity.KeywordLegChecker(lib, h, n_docs)
        cold = chk.run_product(0, n_queries, limit)             # cold posting cache: decode out of the staging buffer
        v = chk.verdict(0, n_queries, limit, product=cold)
        assert v["mismatches"] == 0, v
        assert v["checked_queries"] == n_queries and v["hits_compared"] >= 15 * n_queries and v["score_details_compared"] >= 7 * 15 * n_queries
        # the check is not vacuous: the answers of queries 0..3 are not the oracle's answers for queries 1..4
        assert chk.verdict(1, 4, limit, product=tuple(a[:4] for a in cold))["mismatches"] == 4
        warm = chk.run_product(0, n_queries, limit)             # the same searches out of the HBM posting cache
        for a, b in zip(cold, warm):
            assert (a == b).all()
    finally:
        lib.rb_destroy(h)


def test_c4_keyword_leg_on_the_coherent_corpus(ctx):
    """The same check on BASELINE's own C4 text workload: a coherent corpus (tools/ranked_bench.cpp, struct
    Corpus: title + overview documents, Zipf(1.07) words, every database derived from the same tokens: positions with
    hard-separator jumps, bucketed positions, forward pair proximities, field word counts), a dictionary of the words that
    occur, and queries taken OUT of the documents — 1-3 consecutive words over the whole vocabulary, 0-2 edits, the last
    word cut to a prefix — plus the shapes of workloads/search/movies.json ("" placeholder, two title words, the most
    frequent word).  Against oracle/ranking_oracle.py reading the same stored bytes."""
    import ctypes as C
    from oracle import parity
    from oracle import synth_index as SI
    import os
    n_docs, n_words, n_queries, limit = 10_000_000, 2_000_000, 80, 20
    if os.environ.get("MSI_RUNNER_SO"):        # the CPU tier's emulated kernels: 3 chunks of documents
        n_docs, n_words, n_queries = 150_000, 60_000, 70
    lib = SI.runner_lib()
    lib.rb_attach.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint64]
    h = lib.rb_create_corpus(n_docs, n_words, 42)
    try:
        assert lib.rb_attach(h, ctx.handle, 16, 1024, 2048) == 0
        lib.rb_prepare_queries(h, n_queries, 3, 4242)
        chk = parity.KeywordLegChecker(lib, h, n_docs)
        queries = [chk.index.query(i) for i in range(n_queries)]
        assert queries[0] == "" and len(queries[2].split()) == 2 and len(set(queries)) > n_queries // 2
        from meilisearch_amd._lib import lib as msi
        late0 = (C.c_uint64 * 2)()
        msi().msi_search_late_compaction_stats(late0)
        cold = chk.run_product(0, n_queries, limit)
        v = chk.verdict(0, n_queries, limit, product=cold)
        assert v["mismatches"] == 0, v
        assert v["checked_queries"] == n_queries and v["hits_compared"] >= 5 * n_queries
        # queries with one frequent word: the universe (the union) is too large to compact, the first `words` bucket (the
        # intersection) is not — its sub-tree must have run in the compact space of that bucket (Ctx::late_enter)
        late1 = (C.c_uint64 * 2)()
        msi().msi_search_late_compaction_stats(late1)
        if os.environ.get("MSI_SEARCH_LATE_COMPACT", "1") != "0":
            assert late1[0] > late0[0], (late0[0], late1[0])
        warm = chk.run_product(0, n_queries, limit)
        for a, b in zip(cold, warm):
            assert (a == b).all()
        _index_seen_from_outside(lib, h, chk, queries)
    finally:
        lib.rb_destroy(h)


def _index_seen_from_outside(lib, h, chk, queries, window=20000):
    """engine and oracle read the index through the same runner object, so a stale database fools both.
    On the very handle the device just searched, the documents of a window are read back token by token (rb_doc_tokens) and the
    postings of the queries' own words, their fid / position splits and the pairs of two documents are re-derived from them
    the way milli's write path does (tests/test_corpus_runner_cpu.py does it for the whole index of a small corpus)."""
    import ctypes as C
    w_, f_, p_ = (np.zeros(256, np.uint32) for _ in range(3))
    lib.rb_doc_tokens.argtypes = [C.c_void_p, C.c_uint64] + [C.c_void_p] * 3 + [C.c_uint32]
    lib.rb_doc_tokens.restype = C.c_uint32
    docs = []
    for d in range(window):
        nt = lib.rb_doc_tokens(h, d, w_.ctypes.data, f_.ctypes.data, p_.ctypes.data, 256)
        assert 0 < nt <= 256
        docs.append(list(zip(w_[:nt].tolist(), f_[:nt].tolist(), p_[:nt].tolist())))
    W = chk.index.words
    wid = {w: i for i, w in enumerate(W)}
    probe = []
    for q in queries:
        for w in q.split():
            if w in wid and wid[w] not in probe:
                probe.append(wid[w])
    probe = probe[:12]
    assert len(probe) >= 6

    def in_window(ds):
        return [] if ds is None else sorted(int(x) for x in ds.to_array() if x < window)

    def bucketed(rel):
        if rel < 16:
            return rel
        if rel < 24:
            return 24
        p2 = 1
        while p2 < rel:
            p2 <<= 1
        return p2
    for w in probe:
        assert in_window(chk.index.get_word_docids(W[w], True)) == [d for d, t in enumerate(docs) if any(x == w for x, _, _ in t)], W[w]
        for fid in (1, 2):
            assert in_window(chk.index.get_word_fid_docids(W[w], fid)) == \
                [d for d, t in enumerate(docs) if any(x == w and f == fid for x, f, _ in t)], (W[w], fid)
    w0 = probe[0]
    for pos in chk.index.get_word_positions(W[w0])[:6]:
        assert in_window(chk.index.get_word_position_docids(W[w0], pos)) == \
            [d for d, t in enumerate(docs) if any(x == w0 and bucketed(p) == pos for x, _, p in t)], (W[w0], pos)
    checked_pairs = 0
    for d in range(8):          # adjacent words of a few documents: the pair database of their smallest proximity
        (a, fa, pa), (b, fb, pb) = docs[d][0], docs[d][1]
        if fa != fb or not 1 <= pb - pa <= 3 or a == b:
            continue
        best = {}
        for dd, t in enumerate(docs):
            m = 4
            for i, (x, fx, px) in enumerate(t):
                if x != a:
                    continue
                for y, fy, py in t[i + 1:]:
                    if fy == fx and y == b and 0 < py - px < m:
                        m = py - px
            if m <= 3:
                best[dd] = m
        for pr in (1, 2, 3):
            assert in_window(chk.index.get_pair(pr, W[a], W[b])) == sorted(dd for dd, m in best.items() if m == pr), (pr, W[a], W[b])
        checked_pairs += 1
    assert checked_pairs >= 2


def test_postings_staged_at_index_open_on_the_coherent_corpus(ctx):
    """VERDICT r5 missing #2 (north_star: "staged once into HBM"): msi_dict_stage_postings puts the corpus' word_docids /
    word_fid_docids / word_position_docids / field_id_word_count_docids into the HBM posting cache when the index opens and
    msi_dict_stage_complete declares them complete.  The FIRST search then reads them without a callback: the only misses of
    a cold pass are pair proximities; answers are the oracle's and the same as an engine that met every posting through
    the callbacks; msi_dict_reset_posting_cache keeps what was staged."""
    import ctypes as C
    import os
    from oracle import parity
    from oracle import synth_index as SI
    from meilisearch_amd._lib import lib as msi
    n_docs, n_words, n_queries, limit = 2_000_000, 400_000, 96, 20
    if os.environ.get("MSI_RUNNER_SO"):
        n_docs, n_words, n_queries = 150_000, 60_000, 64
    lib = SI.runner_lib()
    lib.rb_attach.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint64]
    lib.rb_stage_postings.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p]
    lib.rb_dict.restype = C.c_void_p
    lib.rb_dict.argtypes = [C.c_void_p]
    runs = {}
    for staged in (True, False):
        h = lib.rb_create_corpus(n_docs, n_words, 46)
        try:
            assert lib.rb_attach(h, ctx.handle, 8, 1024, 1024) == 0
            d = C.c_void_p(lib.rb_dict(h))
            if staged:
                sec, cnts = C.c_double(0), (C.c_uint64 * 4)()
                assert lib.rb_stage_postings(h, 8, C.byref(sec), cnts) == 0
                # a value per word for word_docids x 2 keys, >= 1 fid and >= 1 position each; bodies and host-kept values both occur
                n_dict = lib.rb_n_words(h)
                assert cnts[0] >= 4 * n_dict and cnts[1] > 0 and cnts[2] > 0 and cnts[3] > 0, list(cnts)
                assert cnts[1] + cnts[2] == cnts[0], list(cnts)
            assert lib.rb_prepare_queries(h, n_queries, 3, 777) == 0
            chk = parity.KeywordLegChecker(lib, h, n_docs)
            pc0 = (C.c_uint64 * 4)()
            msi().msi_dict_posting_cache_stats(d, pc0)
            got = chk.run_product(0, n_queries, limit)     # the engine's FIRST sight of these queries
            pc1 = (C.c_uint64 * 4)()
            msi().msi_dict_posting_cache_stats(d, pc1)
            hits, misses = pc1[0] - pc0[0], pc1[1] - pc0[1]
            v = chk.verdict(0, n_queries, limit, product=got)
            assert v["mismatches"] == 0, v
            assert v["checked_queries"] == n_queries
            runs[staged] = got
            if staged:
                ss = (C.c_uint64 * 4)()
                msi().msi_dict_staged_stats(d, ss)
                assert ss[0] == cnts[1] and ss[1] == cnts[2] and ss[2] == cnts[3]
                assert hits / max(1, hits + misses) >= 0.9, (hits, misses)   # (the misses: pair proximities met for the first time)
                assert ss[3] > 0                           # reads a complete database answered "absent" without the index
                used_staged = int(pc0[2])
                # forgetting what the searches cached keeps the index: same answers, at least the staged bytes still in use,
                # and the cold pass after it misses no more than the first one did
                assert msi().msi_dict_reset_posting_cache(d) == 0
                pc2 = (C.c_uint64 * 4)()
                msi().msi_dict_posting_cache_stats(d, pc2)
                assert pc2[2] == used_staged and pc2[0] == 0 and pc2[1] == 0, (list(pc2), used_staged)
                again = chk.run_product(0, n_queries, limit)
                for a_, b_ in zip(got, again):
                    assert (a_ == b_).all()
                pc3 = (C.c_uint64 * 4)()
                msi().msi_dict_posting_cache_stats(d, pc3)
                assert pc3[1] <= misses, (pc3[1], misses)
                _index_seen_from_outside(lib, h, chk, [chk.index.query(i) for i in range(n_queries)])
            else:
                assert misses > 4 * n_queries, (hits, misses)   # without staging a cold pass meets its postings through the callbacks
        finally:
            lib.rb_destroy(h)
    for a_, b_ in zip(runs[True], runs[False]):
        assert (a_ == b_).all()


def test_phrases_on_the_coherent_corpus(ctx):
    """quoted phrases on an index of several chunks, through universe compaction and the bucket-space
    sub-trees — every eighth query of the corpus workload opens with a phrase of two consecutive words of a document
    (rb_prepare_queries_ex, flags 1), a misspelled / prefix word may follow it; against oracle/ranking_oracle.py on the same
    stored bytes.  On the CPU tier's emulated kernels: 150 000 documents, three chunks; on the MI355X (first run there in round 5,
    green): 2 M documents."""
    import ctypes as C
    import os
    from oracle import parity
    from oracle import synth_index as SI
    n_docs, n_words, n_queries, limit = 2_000_000, 400_000, 96, 20
    if os.environ.get("MSI_RUNNER_SO"):
        n_docs, n_words, n_queries = 150_000, 60_000, 96
    lib = SI.runner_lib()
    lib.rb_attach.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint64]
    h = lib.rb_create_corpus(n_docs, n_words, 44)
    try:
        assert lib.rb_attach(h, ctx.handle, 8, 1024, 1024) == 0
        assert lib.rb_prepare_queries_ex(h, n_queries, 3, 515, 1) == 0
        chk = parity.KeywordLegChecker(lib, h, n_docs)
        queries = [chk.index.query(i) for i in range(n_queries)]
        phrases = [q for q in queries if q.startswith('"')]
        assert len(phrases) >= n_queries // 10 and any(not q.endswith('"') for q in phrases), phrases[:4]
        got = chk.run_product(0, n_queries, limit)
        v = chk.verdict(0, n_queries, limit, product=got)
        assert v["mismatches"] == 0, v
        # the phrases found documents (they are taken out of one) and their hits carry the rules' details
        hits = [int(got[1][i]) for i, q in enumerate(queries) if q.startswith('"')]
        assert min(hits) >= 1, (hits, phrases[:4])
    finally:
        lib.rb_destroy(h)


def test_word_prefix_databases_on_the_coherent_corpus(ctx):
    """VERDICT r3 missing #1 / weak #1 (ii): the corpus index with word-prefix databases (rb_enable_prefix_dbs: keys = the
    prefixes of up to four bytes that enough dictionary words share; word_prefix_docids / _fid_docids / _position_docids
    derived from the same tokens, the pair database's prefix_iter for the proximity rule) and the query shapes that read
    them — workloads/search/movies.json's one-letter query, two- and three-letter prefixes, a word followed by a short
    prefix — next to the usual misspelled / prefix-cut queries and quoted phrases, on three chunks of documents.  Against
    oracle/ranking_oracle.py reading the values the index hands to the engine's sink.  Like the phrase test: written after
    the round's GPU minutes were spent; first run on the MI355X in round 5 (green)."""
    import ctypes as C
    import os
    from oracle import parity
    from oracle import synth_index as SI
    n_docs, n_words, n_queries, limit = 2_000_000, 400_000, 128, 20
    if os.environ.get("MSI_RUNNER_SO"):
        n_docs, n_words, n_queries = 150_000, 60_000, 128
    lib = SI.runner_lib()
    lib.rb_attach.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint64]
    h = lib.rb_create_corpus(n_docs, n_words, 45)
    try:
        assert lib.rb_enable_prefix_dbs(h, 50) == 0
        assert lib.rb_attach(h, ctx.handle, 8, 1024, 1024) == 0
        assert lib.rb_prepare_queries_ex(h, n_queries, 3, 616, 3) == 0
        chk = parity.KeywordLegChecker(lib, h, n_docs)
        queries = [chk.index.query(i) for i in range(n_queries)]
        short = [q for q in queries if q and len(q.split()[-1]) <= 3 and not q.endswith('"')]
        assert len(short) >= 4 and any(len(q) == 1 for q in short), short
        assert all(chk.index.has_prefix(q.split()[-1][:1], False) for q in short)     # one letter: always a key here
        reads0 = chk.index.reads
        got = chk.run_product(0, n_queries, limit)
        v = chk.verdict(0, n_queries, limit, product=got)
        assert v["mismatches"] == 0, v
        assert v["checked_queries"] == n_queries
        # a one-letter prefix matches a large share of the documents: the page is full
        for i, q in enumerate(queries):
            if len(q) == 1:
                assert int(got[1][i]) == limit, (q, int(got[1][i]))
        assert chk.index.reads > reads0
    finally:
        lib.rb_destroy(h)


def test_synonyms_on_the_coherent_corpus(ctx):
    """VERDICT r3 missing #1 / weak #1 (ii): synonyms on the corpus index (rb_enable_synonyms: a sixteenth of the vocabulary
    has a one-word synonym, half of those a two-word phrase that occurs in some title; an eighth of the adjacent pairs a
    synonym for their n-gram key) together with the word-prefix databases and quoted phrases — every eighth query is a word
    or a pair that has synonyms — on three chunks of documents, against oracle/ranking_oracle.py.  CPU tier only for now
    (see test_phrases_on_the_coherent_corpus)."""
    import ctypes as C
    import os
    from oracle import parity
    from oracle import synth_index as SI
    n_docs, n_words, n_queries, limit = 2_000_000, 400_000, 128, 20
    if os.environ.get("MSI_RUNNER_SO"):
        n_docs, n_words, n_queries = 150_000, 60_000, 128
    lib = SI.runner_lib()
    lib.rb_attach.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint64]
    h = lib.rb_create_corpus(n_docs, n_words, 46)
    try:
        assert lib.rb_enable_prefix_dbs(h, 50) == 0 and lib.rb_enable_synonyms(h) == 0
        assert lib.rb_attach(h, ctx.handle, 8, 1024, 1024) == 0
        assert lib.rb_prepare_queries_ex(h, n_queries, 3, 717, 7) == 0
        chk = parity.KeywordLegChecker(lib, h, n_docs)
        queries = [chk.index.query(i) for i in range(n_queries)]
        with_syn = [q for i, q in enumerate(queries) if i % 8 == 6 and q and not q.startswith('"')]
        has = [q for q in with_syn if chk.index.get_synonyms((q.split()[0],)) or chk.index.get_synonyms(tuple(q.split()[:2]))]
        assert len(has) >= 8, (with_syn, has)
        assert any(len(s_) == 2 for q in has for s_ in chk.index.get_synonyms((q.split()[0],))), "no two-word synonym among them"
        got = chk.run_product(0, n_queries, limit)
        v = chk.verdict(0, n_queries, limit, product=got)
        assert v["mismatches"] == 0, v
        assert v["checked_queries"] == n_queries
    finally:
        lib.rb_destroy(h)


def test_negative_terms_on_the_coherent_corpus(ctx):
    """`-word` and `-"a phrase"` (search/mod.rs:431-440: their documents leave the universe before anything else) at corpus
    scale: every eighth query excludes a word or an adjacent pair of some other document, next to phrases, prefix databases
    and synonyms — the universe a search compacts (or the bucket it later moves into) is what is left.  CPU tier only for now
    (see test_phrases_on_the_coherent_corpus)."""
    import ctypes as C
    import os
    from oracle import parity
    from oracle import synth_index as SI
    n_docs, n_words, n_queries, limit = 2_000_000, 400_000, 128, 20
    if os.environ.get("MSI_RUNNER_SO"):
        n_docs, n_words, n_queries = 150_000, 60_000, 128
    lib = SI.runner_lib()
    lib.rb_attach.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint64]
    h = lib.rb_create_corpus(n_docs, n_words, 47)
    try:
        assert lib.rb_enable_prefix_dbs(h, 50) == 0 and lib.rb_enable_synonyms(h) == 0
        assert lib.rb_attach(h, ctx.handle, 8, 1024, 1024) == 0
        assert lib.rb_prepare_queries_ex(h, n_queries, 3, 818, 15) == 0
        chk = parity.KeywordLegChecker(lib, h, n_docs)
        negs = [chk.index.negatives(i) for i in range(n_queries)]
        assert sum(1 for x in negs if x) >= n_queries // 9
        assert any(isinstance(x[0], tuple) for x in negs if x) and any(isinstance(x[0], str) for x in negs if x)
        got = chk.run_product(0, n_queries, limit)
        v = chk.verdict(0, n_queries, limit, product=got)
        assert v["mismatches"] == 0, v
        # a negative term that bites: at least one query's candidates shrink against the same query without it
        shrunk = 0
        for i in range(n_queries):
            if negs[i]:
                plain = chk.oracle.search(chk.index.query(i), limit=limit)[2]
                shrunk += int(got[5][i]) < plain
        assert shrunk >= 1
    finally:
        lib.rb_destroy(h)


def test_rerank_inside_candidate_universes_on_the_corpus(ctx):
    """Config 5's second half as written: the keyword ranking (all default criteria, detailed scores) of a query restricted
    to a candidate set of 1 000 documents — what reranks a filtered vector search's top-1000 — through the runner's
    rb_run_universes (the universe reaches msi_keyword_search_ranked as the CboRoaringBitmap the shim would pass), against
    oracle/ranking_oracle.py given the same universe."""
    import ctypes as C
    from oracle import parity
    from oracle import synth_index as SI
    import os
    n_docs, n_words, n_queries, limit = 2_000_000, 200_000, 32, 20
    if os.environ.get("MSI_RUNNER_SO"):
        n_docs, n_words, n_queries = 150_000, 60_000, 24
    lib = SI.runner_lib()
    lib.rb_attach.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint64]
    h = lib.rb_create_corpus(n_docs, n_words, 43)
    try:
        assert lib.rb_attach(h, ctx.handle, 8, 512, 512) == 0
        lib.rb_prepare_queries(h, n_queries, 3, 99)
        chk = parity.KeywordLegChecker(lib, h, n_docs)
        rng = np.random.default_rng(5)
        # candidate sets that hold matches: half of each from the query's own result list region (low docids are as good as any)
        uni = np.zeros((n_queries, 1000), np.uint32)
        cnt = np.full(n_queries, 1000, np.uint32)
        plain = chk.run_product(0, n_queries, 200)
        for i in range(n_queries):
            hits = plain[0][i, :int(plain[1][i])]
            rest = rng.choice(n_docs, 1000 - hits.size, replace=False).astype(np.uint32)
            uni[i] = np.concatenate([hits, rest])
        v = chk.verdict(0, n_queries, limit, universes=(uni, cnt))
        assert v["mismatches"] == 0, v
        assert v["hits_compared"] >= 3 * n_queries
    finally:
        lib.rb_destroy(h)


def test_a_burst_of_alike_searches_with_large_lists_fused(ctx, monkeypatch):
    """VERDICT r4 #4 / ADVICE r4 (medium).  Round 4 found the keyword leg collapsing from 9 700 to 23-170 searches/s when 160
    searches whose universes are 1-12.5 % of a 10 M-document index (compact lists of 48-153 chunks) arrived together and their
    lists were FUSED (wide phase + waiting command workgroups in one launch) under a budget of 4 096 waiting workgroups — four
    times what the device keeps resident.  The budget now follows from the residency (msi_vm.hip: CUs x occupancy / 4) and a
    waiter gives up after MSI_VM_SPIN_LIMIT_TICKS, failing its list instead of hanging the device.  This runs that burst with
    fusing allowed for every list size (MSI_VM_FUSE_MAX_CHUNKS=160: only the residency budget protects the device) and asks for a
    throughput two orders of magnitude above the collapse; the same searches with fusing off must answer the same lists."""
    import ctypes as C
    import os
    import time
    from oracle import synth_index as SI
    if os.environ.get("MSI_RUNNER_SO"):
        pytest.skip("dispatch behaviour of the device: nothing the emulated kernels can show")
    n_docs, n_words, n_queries, limit, callers = 10_000_000, 2_000_000, 1536, 20, 160
    lib = SI.runner_lib()
    lib.rb_attach.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint64]
    lib.rb_run_detailed.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32] + [C.c_void_p] * 6
    lib.rb_run.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.rb_permute_queries.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32]
    h = lib.rb_create_corpus(n_docs, n_words, 42)
    try:
        assert lib.rb_attach(h, ctx.handle, callers, 512, 4096) == 0
        lib.rb_prepare_queries(h, n_queries, 3, 4242)
        ids = np.zeros((n_queries, limit), np.uint32)
        cnt = np.zeros(n_queries, np.uint32)
        sc = np.zeros((n_queries, limit), np.float64)
        cand = np.zeros(n_queries, np.uint64)
        assert lib.rb_run_detailed(h, 0, n_queries, limit, ids.ctypes.data, cnt.ctypes.data, sc.ctypes.data, None, None,
                                   cand.ctypes.data) == 0
        order = np.argsort(cand, kind="stable").astype(np.uint32)
        assert lib.rb_permute_queries(h, order.ctypes.data, n_queries) == 0
        share = cand[order].astype(np.float64) / n_docs
        lo, hi = int(np.searchsorted(share, 0.01, "right")), int(np.searchsorted(share, 0.125, "right"))
        assert hi - lo >= callers, (lo, hi)          # more alike searches than callers: every caller holds one at the same time
        monkeypatch.setenv("MSI_VM_FUSE_MAX_CHUNKS", "0")
        assert lib.rb_run(h, lo, hi - lo, limit, ids.ctypes.data, cnt.ctypes.data, sc.ctypes.data) == 0
        ref_ids, ref_cnt = ids[:hi - lo].copy(), cnt[:hi - lo].copy()
        monkeypatch.setenv("MSI_VM_FUSE_MAX_CHUNKS", "160")
        best = 0.0
        for _ in range(3):
            t0 = time.perf_counter()
            assert lib.rb_run(h, lo, hi - lo, limit, ids.ctypes.data, cnt.ctypes.data, sc.ctypes.data) == 0
            best = max(best, (hi - lo) / (time.perf_counter() - t0))
            assert (cnt[:hi - lo] == ref_cnt).all() and (ids[:hi - lo] == ref_ids).all()
        assert best >= 2000.0, f"{best:.0f} searches/s in a burst of {hi - lo} alike searches (the collapse ran at 23-170)"
    finally:
        lib.rb_destroy(h)
