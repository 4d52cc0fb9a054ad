"""oracle/docset.py (test infrastructure): the numpy docid-set type the ranking oracle runs on at 10 M documents.

  1. operation by operation against the built-in set (random programs, thresholds chosen so that sets keep
     crossing between the sparse and the dense representation);
  2. the reference's own snapshot searches (tests/golden/ranking_snapshots.json) replayed through
     oracle/ranking_oracle.py in DocSet mode: same docids and score details as the snapshots — so the oracle that
     checks the 10 M-document keyword leg (tests/test_configs_gpu.py::test_c4_keyword_leg, bench.py's parity object)
     is the oracle the snapshots pin, not a second restatement;
  3. the CboRoaringBitmap decoder against the encoders of the test tier and the bytes milli itself wrote."""
import json
import os
import random

import numpy as np
import pytest

from oracle import docset
from oracle import ranking_oracle as R
from tests.test_ranking_oracle_snapshots import (CASES, FIX, build_index, debug_ids_scores, debug_scores, make_ctx)


class DocSetIndex:
    """Any index of the test tier with every docid set it hands out turned into the oracle's current DocSet type."""

    def __init__(self, index, cls):
        self._index, self._cls = index, cls

    def __getattr__(self, name):
        v = getattr(self._index, name)
        if not callable(v):
            return v
        cls = self._cls

        def wrapped(*a, **kw):
            r = v(*a, **kw)
            return cls(r) if isinstance(r, (set, frozenset)) else r
        return wrapped


@pytest.mark.parametrize("seed", range(6))
def test_docset_equals_set_operation_by_operation(seed):
    rng = random.Random(seed)
    n = 700
    D = docset.docset_type(n, dense_above=12, sparse_below=6)

    def draw():
        k = rng.choice([0, 1, 3, 8, 40, 300])
        return set(rng.sample(range(n), k))
    pairs = [(draw(), None) for _ in range(6)]
    pairs = [(s, D(s)) for s, _ in pairs]
    for step in range(600):
        i, j = rng.randrange(len(pairs)), rng.randrange(len(pairs))
        (a, da), (b, db) = pairs[i], pairs[j]
        op = rng.randrange(16)
        if op == 0:
            r = (a & b, da & db)
        elif op == 1:
            r = (a | b, da | db)
        elif op == 2:
            r = (a - b, da - db)
        elif op == 3:
            a &= b if i != j else set(b)
            da &= db if i != j else D(db)
            r = (a, da)
        elif op == 4:
            a |= b
            da |= db
            r = (a, da)
        elif op == 5:
            if i != j:
                a -= b
                da -= db
            r = (a, da)
        elif op == 6:          # a copy must not follow later in-place writes of the original (and vice versa)
            c, dc = set(a), D(da)
            a |= b
            da |= db
            assert sorted(dc) == sorted(c)
            c -= b
            dc -= db
            assert sorted(da) == sorted(a)
            r = (c, dc)
        elif op == 7:
            x = rng.randrange(n)
            a.add(x)
            da.add(x)
            r = (a, da)
        elif op == 8:
            x = rng.randrange(n)
            a.discard(x)
            da.discard(x)
            r = (a, da)
        elif op == 9:
            a.update(b)
            da.update(db)
            r = (a, da)
        elif op == 10:
            if i != j:
                a.difference_update(b)
                da.difference_update(db)
            r = (a, da)
        elif op == 11:         # mixed operands: a built-in set on either side
            r = (a & b, a & db)
            assert isinstance(r[1], D)
        elif op == 12:
            r = (a - b, a - db)
        elif op == 13:
            r = (a | b, da | b)
        elif op == 14:
            assert (a <= b) == (da <= db) and (a >= b) == (da >= db) and (a == b) == (da == db)
            assert (a <= b) == (a <= db) and (a <= b) == (da <= b)
            continue
        else:
            x = rng.randrange(n)
            assert (x in a) == (x in da)
            continue
        s, d = r
        assert list(d) == sorted(s) and len(d) == len(s) and bool(d) == bool(s), (step, op)
        pairs[rng.randrange(len(pairs))] = (set(s), D(d))
    full = D.full()
    assert len(full) == n and list(full) == list(range(n))


@pytest.mark.parametrize("thresholds", [(4096, 2048), (3, 2)], ids=["sparse", "mostly-dense"])
def test_reference_snapshots_through_the_oracle_in_docset_mode(thresholds):
    """Every snapshot search of the reference (108 with Sort / distinct) with the oracle's docid sets as DocSets."""
    done = 0
    indexes = {}
    for case in CASES:
        cfg = FIX["indexes"][case["index"]]
        if cfg.get("unsupported") or case.get("needs"):
            continue
        if case["index"] not in indexes:
            indexes[case["index"]] = build_index(cfg)
        index = indexes[case["index"]]
        D = docset.docset_type(max(index.n_docs, 1), *thresholds)
        with R.use_docset(D):
            ctx = make_ctx(DocSetIndex(index, D))
            ids, scores, cand = R.search(ctx, case["query"], tms=case["tms"], offset=case["offset"], length=case["limit"],
                                         detailed=case["detailed"], stop_after=case.get("stop_after"),
                                         distinct=case.get("distinct") or index.distinct_field, sort=case.get("sort"))
        want_ids, want_scores, want_cand = R.search(make_ctx(index), case["query"], tms=case["tms"], offset=case["offset"],
                                                    length=case["limit"], detailed=case["detailed"],
                                                    stop_after=case.get("stop_after"),
                                                    distinct=case.get("distinct") or index.distinct_field, sort=case.get("sort"))
        assert R.DocSet is set
        assert ids == want_ids and scores == want_scores and sorted(cand) == sorted(want_cand), case["query"]
        if case["ids"] is not None:
            assert ids == case["ids"]
        if case.get("scores"):
            assert debug_scores(scores) == case["scores"]
        if case.get("ids_scores"):
            assert debug_ids_scores(ids, scores) == case["ids_scores"]
        done += 1
    assert done >= 100


def test_random_corpora_in_docset_mode():
    """Random corpora x rule lists x strategies: DocSet mode == set mode (docids, score details, candidates)."""
    from oracle import oracle as O
    from tests.toy_milli import ToyMilli
    rng = random.Random(11)
    vocab = ["quick", "quack", "brown", "fox", "foxes", "jumps", "jumped", "lazy", "dog", "dogs", "summer", "winter",
             "holiday", "holidays", "network", "interconnection", "sweet", "dessert", "the", "a", "of"]
    docs = [{"id": i, "title": " ".join(rng.choices(vocab, k=rng.randint(1, 5))),
             "body": " ".join(rng.choices(vocab, k=rng.randint(3, 30)))} for i in range(400)]
    rulesets = [["words", "typo", "proximity", "attributeRank", "sort", "wordPosition", "exactness"],
                ["typo", "words", "exactness"], ["proximity", "attribute"]]
    queries = ["quick brown fox", "the lazy dog jumps", "quik brwn", "holi", "sweet desert summer", "\"lazy dog\" fox"]
    for criteria in rulesets:
        index = ToyMilli(docs, searchable=["title", "body"], criteria=criteria)
        dic = O.Dictionary(index.words)

        def lookup(word, max_typos, is_prefix):
            one, two = O.typo_lookup(dic, word, max_typos, is_prefix)
            return [index.words[i] for i in one], [index.words[i] for i in two]
        for th in ((4096, 2048), (5, 3)):
            D = docset.docset_type(index.n_docs, *th)
            for q in queries:
                for tms in ("last", "all"):
                    for detailed in (True, False):
                        want = R.search(R.Ctx(index, lookup), q, tms=tms, criteria=criteria, length=30, detailed=detailed)
                        with R.use_docset(D):
                            got = R.search(R.Ctx(DocSetIndex(index, D), lookup), q, tms=tms, criteria=criteria, length=30,
                                           detailed=detailed)
                        assert got[0] == want[0] and got[1] == want[1] and sorted(got[2]) == sorted(want[2]), (criteria, q, tms)


def test_cbo_decoder():
    from meilisearch_amd import synth
    from tests.toy_index import cbo_bytes
    rng = np.random.default_rng(5)
    for n, hi in ((0, 10), (1, 10), (7, 1 << 20), (8, 1 << 20), (300, 1 << 17), (5000, 70000), (60000, 1 << 18)):
        ids = np.unique(rng.integers(0, hi, size=n)).astype(np.uint32)
        assert docset.decode_cbo(cbo_bytes(set(ids.tolist())) if n else b"").tolist() == ids.tolist()
        assert docset.decode_cbo(synth.cbo_serialize(ids)).tolist() == ids.tolist()
    # run containers (cookie 12347), as roaring-rs writes them after optimize(): with and without the offset header
    import struct

    def runs_bytes(conts):
        n = len(conts)
        out = bytearray(struct.pack("<I", 12347 | ((n - 1) << 16)) + bytes([0xFF] * ((n + 7) // 8)))
        for key, runs in conts:
            out += struct.pack("<HH", key, sum(l + 1 for _, l in runs) - 1)
        if n >= 4:
            out += b"\0" * (4 * n)
        for _, runs in conts:
            out += struct.pack("<H", len(runs))
            for s_, l in runs:
                out += struct.pack("<HH", s_, l)
        return bytes(out)
    for conts in ([(0, [(10, 4989), (6000, 0)]), (1, [(4464, 99)]), (3, [(0, 65535)])],
                  [(k, [(k, 10), (100 + k, 3)]) for k in range(5)]):
        want = [(key << 16) + s_ + i for key, runs in conts for s_, l in runs for i in range(l + 1)]
        assert docset.decode_roaring(runs_bytes(conts)).tolist() == want
    # bytes milli itself wrote
    blobs = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "index_blobs.json")))
    n = 0
    for b in blobs["bitmaps"]:
        raw = bytes.fromhex(b["hex"])
        if b["codec"] == "cbo":
            got = docset.decode_cbo(raw)
            assert got.tolist() == np.frombuffer(raw, "<u4").tolist()
        else:
            assert docset.decode_roaring(raw).tolist() == list(range(int(b["n_documents"])))
        n += 1
    assert n >= 70
