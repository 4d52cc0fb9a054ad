"""The coherent corpus of tools/ranked_bench.cpp (the keyword workload of bench.py's C4 / C5 lines and of the corpus parity
tests) against a brute-force re-derivation of its databases.  The parity checks hold the ENGINE against the oracle on the
bytes the runner hands out — both read the same bytes, so a wrong database goes unnoticed there (round 4: a cache keyed by the
index's address served a second corpus the key sets of a destroyed one).  Here the documents themselves are read back
(rb_doc_tokens) and every database is rebuilt the way milli's write path does (tests/toy_milli.py restates it:
extract_word_docids.rs:76-99, extract_word_pair_proximity_docids.rs:232-233,504-515, lib.rs:248-262 bucketed positions),
then compared with the stored values decoded by oracle/docset.py.  No device: the runner's index needs none."""
import ctypes as C
import os

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def bucketed(rel):
    if rel < 16:
        return rel
    if rel < 24:
        return 24
    p = 1
    while p < rel:
        p <<= 1
    return p


@pytest.fixture(scope="module")
def corpora():
    if not os.path.exists(os.path.join(ROOT, "tools", "bin", "libmsi_rankedbench.so")):
        import __graft_entry__
        __graft_entry__.build()
    from oracle import synth_index as SI
    lib = SI.runner_lib()
    made = []

    def make(n_docs, n_words, seed, prefix_threshold=0):
        h = lib.rb_create_corpus(n_docs, n_words, seed)
        if prefix_threshold:
            assert lib.rb_enable_prefix_dbs(h, prefix_threshold) == 0
        made.append(h)
        ix = SI.SynthIndex(lib, h, n_docs)
        docs = []
        w, f, p = (np.zeros(256, np.uint32) for _ in range(3))
        for d in range(n_docs):
            n = lib.rb_doc_tokens(h, d, w.ctypes.data, f.ctypes.data, p.ctypes.data, 256)
            assert 0 < n <= 256
            docs.append(list(zip(w[:n].tolist(), f[:n].tolist(), p[:n].tolist())))
        return ix, docs
    yield make
    for h in made:
        lib.rb_destroy(h)


def ids_of(ds):
    return [] if ds is None else sorted(int(x) for x in ds.to_array())


def brute(docs, n_words):
    word, fid, pos, pair, count = {}, {}, {}, {}, {}
    for d, toks in enumerate(docs):
        per_field = {}
        for w, f, p in toks:
            word.setdefault(w, set()).add(d)
            fid.setdefault((w, f), set()).add(d)
            pos.setdefault((w, bucketed(p)), set()).add(d)
            per_field.setdefault(f, []).append((w, p))
        best = {}
        for f, seq in per_field.items():
            count.setdefault((f, len(seq)), set()).add(d)
            for i, (a, pa) in enumerate(seq):
                for b, pb in seq[i + 1:]:
                    if pb - pa > 3:
                        break                       # positions only grow inside a field
                    if pb > pa:
                        best[(a, b)] = min(best.get((a, b), 4), pb - pa)
        for (a, b), pr in best.items():
            if 1 <= pr <= 3:
                pair.setdefault((pr, a, b), set()).add(d)
    return word, fid, pos, pair, count


def test_databases_are_what_the_documents_say(corpora):
    ix, docs = corpora(3000, 2500, 7)
    W = ix.words
    word, fid, pos, pair, count = brute(docs, len(W))
    assert sorted(word) == list(range(len(W))), "the dictionary is the words that occur"
    assert W == sorted(W)
    rng = np.random.default_rng(1)
    for w in rng.choice(len(W), 400, replace=False).tolist() + [max(word, key=lambda k: len(word[k]))]:
        assert ids_of(ix.get_word_docids(W[w], True)) == sorted(word[w]), W[w]
        fids = sorted({f for (x, f) in fid if x == w})
        assert ix.get_word_fids(W[w]) == fids
        for f in (1, 2):
            assert ids_of(ix.get_word_fid_docids(W[w], f)) == sorted(fid.get((w, f), ())), (W[w], f)
        poss = sorted({p for (x, p) in pos if x == w})
        assert ix.get_word_positions(W[w]) == poss
        for p in poss:
            assert ids_of(ix.get_word_position_docids(W[w], p)) == sorted(pos[(w, p)]), (W[w], p)
    keys = list(pair)
    for k in rng.choice(len(keys), 600, replace=False):
        pr, a, b = keys[k]
        for q in (1, 2, 3):     # a pair sits in the database of its SMALLEST proximity only
            assert ids_of(ix.get_pair(q, W[a], W[b])) == sorted(pair.get((q, a, b), ())), (q, W[a], W[b])
    assert ix.get_pair(1, W[0], W[0] + "zz") is None
    for (f, n), ds in count.items():
        if n <= 30:
            assert ids_of(ix.get_fid_word_count_docids(f, n)) == sorted(ds), (f, n)
    # titles are 3-6 words, overviews 20-60: hard separators only inside overviews (+8 positions)
    assert {f for toks in docs for _, f, _ in toks} == {1, 2}
    assert all(3 <= sum(1 for _, f, _ in toks if f == 1) <= 6 for toks in docs)
    assert any(b - a == 8 for toks in docs for (_, fa, a), (_, fb, b) in zip(toks, toks[1:]) if fa == fb == 2)


def test_a_second_corpus_does_not_see_the_first_ones_key_sets():
    """Round 4's runner bug, pinned: a corpus created after another one was destroyed (the allocator likes to hand out the
    same address again) answers from its own documents — the key sets were cached per index ADDRESS."""
    from oracle import synth_index as SI
    lib = SI.runner_lib()
    w, f, p = (np.zeros(256, np.uint32) for _ in range(3))
    for seed in (11, 12, 13, 11):
        h = lib.rb_create_corpus(800, 900, seed)
        try:
            ix = SI.SynthIndex(lib, h, 800)
            docs = []
            for d in range(800):
                n = lib.rb_doc_tokens(h, d, w.ctypes.data, f.ctypes.data, p.ctypes.data, 256)
                docs.append(list(zip(w[:n].tolist(), f[:n].tolist(), p[:n].tolist())))
            _, fid, pos, _, _ = brute(docs, len(ix.words))
            for wi in range(0, len(ix.words), 5):
                assert ix.get_word_fids(ix.words[wi]) == sorted({f_ for (x, f_) in fid if x == wi}), (seed, wi)
                assert ix.get_word_positions(ix.words[wi]) == sorted({p_ for (x, p_) in pos if x == wi}), (seed, wi)
        finally:
            lib.rb_destroy(h)


def test_word_prefix_databases_are_the_union_of_their_words(corpora):
    ix, docs = corpora(3000, 2500, 9, prefix_threshold=20)
    W = ix.words
    word, fid, pos, pair, _ = brute(docs, len(W))
    seen = 0
    for pfx in sorted({w[:n] for w in W for n in (1, 2, 3, 4)}):
        members = [i for i, w in enumerate(W) if w.startswith(pfx)]
        is_key = len(members) >= 20
        assert ix.has_prefix(pfx, False) == is_key, pfx
        if not is_key:
            assert ix.get_word_prefix_docids(pfx, True) is None
            continue
        seen += 1
        if seen % 5:
            continue                                   # every fifth key in full
        assert ids_of(ix.get_word_prefix_docids(pfx, True)) == sorted(set().union(*(word[m] for m in members))), pfx
        fids = sorted({f for (x, f) in fid if x in set(members)})
        assert ix.get_word_prefix_fids(pfx) == fids
        for f in fids:
            want = sorted(set().union(*(fid.get((m, f), set()) for m in members)))
            assert ids_of(ix.get_word_prefix_fid_docids(pfx, f)) == want, (pfx, f)
        poss = sorted({p for (x, p) in pos if x in set(members)})
        assert ix.get_word_prefix_positions(pfx) == poss
        for p in poss[:6]:
            want = sorted(set().union(*(pos.get((m, p), set()) for m in members)))
            assert ids_of(ix.get_word_prefix_position_docids(pfx, p)) == want, (pfx, p)
    assert seen >= 20
    assert ix.has_prefix("zzzzz", False) is False     # five bytes: never a key
    # the pair database's prefix_iter: (proximity, word1, prefix2...) = the union over the words of the prefix
    rng = np.random.default_rng(3)
    keys = list(pair)
    for k in rng.choice(len(keys), 60, replace=False):
        pr, a, b = keys[k]
        pfx = W[b][:2]
        want = sorted(set().union(*(pair.get((pr, a, m), set()) for m in range(len(W)) if W[m].startswith(pfx))))
        assert ids_of(ix.get_word_prefix_pair(pr, W[a], pfx)) == want, (pr, W[a], pfx)
