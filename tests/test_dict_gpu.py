"""S2 parity: libmsi's typo lookup (through the C ABI) against the literal CPU
restatement of compute_derivations.rs:75-168.  Bar: identical index lists."""
import numpy as np
import pytest

import meilisearch_amd as ma
from meilisearch_amd import synth

pytestmark = pytest.mark.gpu


def compare(oracle, words, queries, caps=(150, 50), ctx=None):
    concat, off = synth.flatten_words(words)
    odic = oracle.Dictionary.from_flat(concat, off)
    gdic = ma.GpuDictionary(ctx, concat=concat, offsets=off)
    assert len(gdic) == len(words)
    got = gdic.lookup(queries, cap_one=caps[0], cap_two=caps[1])
    for (w, b, p), (g1, g2) in zip(queries, got):
        e1, e2 = oracle.typo_lookup(odic, w, b, p, cap_one=caps[0], cap_two=caps[1])
        assert g1.tolist() == e1.tolist(), ("one", w, b, p, caps, g1[:8], e1[:8])
        assert g2.tolist() == e2.tolist(), ("two", w, b, p, caps, g2[:8], e2[:8])
    return gdic


def test_reference_words(ctx, oracle):
    # corpus words of crates/milli/src/search/new/tests/typo.rs and tests/search/typo_tolerance.rs
    words = sorted(set(
        "the quick brown fox jumps over the lazy dog quickest quickly quack quickbrownfox brow browny "
        "brownie foxes jumped jump jumper lazily dogs zeal zealand zealot zoo netwolk network wolk wol "
        "zean zealemd".split()), key=lambda w: w.encode())
    g = ma.GpuDictionary(ctx, words=words)
    assert "quick" in g.find_one_typo_derivations("quack", False)
    assert "quickest" in g.find_one_typo_derivations("quicest", False)
    assert "jumps" in g.find_one_typo_derivations("jummps", False)
    one, two = g.find_one_two_typo_derivations("zuickest", False)
    assert "quickest" in two and "quickest" not in one          # first-letter typo = 2 typos
    assert g.find_one_typo_derivations("zuickest", False) == []
    assert "quick" not in g.find_one_typo_derivations("quick", False)
    # typo_tolerance.rs:37-175: zeal/zean (1 typo), zealand/zealemd (2 typos)
    assert "zeal" in g.find_one_typo_derivations("zean", False)
    one, two = g.find_one_two_typo_derivations("zealemd", False)
    assert "zealand" in two
    queries = [(w, b, p) for w in ["quack", "quicest", "jummps", "zuickest", "quick", "zean", "zealemd",
                                   "brow", "netwolk", "qu", "t"] for b in (1, 2) for p in (False, True)]
    compare(oracle, words, queries, ctx=ctx)
    compare(oracle, words, queries, caps=(2, 1), ctx=ctx)


def test_cap_interplay(ctx, oracle):
    words = sorted(["aello", "bello", "cello", "dello", "hallo", "hella", "hello", "hellos", "jello",
                    "hxllo", "hexlo", "helxo", "hellx", "helol", "ehllo", "yello", "zello"],
                   key=lambda w: w.encode())
    queries = [("hello", 2, False), ("hello", 2, True), ("hello", 1, False), ("hell", 1, True)]
    for caps in [(150, 50), (3, 2), (1, 1), (2, 5), (5, 1)]:
        compare(oracle, words, queries, caps=caps, ctx=ctx)


def test_synthetic_dictionary_all_paths(ctx, oracle):
    words = synth.make_dictionary(20000, seed=17)
    extra = ["internationalisationally", "internationalization", "антидисестаблишментарианизм",
             "ünïcödé", "日本語のテキスト", "日本語", "😀smile", "a", "ab", "abc",
             "pneumonoultramicroscopicsilicovolcanoconiosis", "x" * 250]
    words = sorted(set(words + extra), key=lambda w: w.encode())
    queries = synth.make_typo_queries(words, 400, seed=19)
    queries += [("internationalisationaly", 2, False), ("internationalizatio", 2, True),
                ("антидисестаблишментарианизн", 2, False), ("ünïcodé", 1, False), ("日本誤", 1, False),
                ("日本語のテキス", 2, True), ("😀smil", 1, True), ("pneumonoultramicroscopicsilicovolcanoconiosi", 2, False),
                ("x" * 249, 2, False), ("y" + "x" * 249, 2, False), ("q", 1, True), ("ab", 2, True), ("a", 2, False),
                ("z" * 251, 2, False), ("", 1, False)]
    for caps in [(150, 50), (4, 3)]:
        g = compare(oracle, words, queries, caps=caps, ctx=ctx)
    assert g.stats()["pairs_scanned"] > 0


def test_batch_sizes_and_determinism(ctx, oracle):
    words = synth.make_dictionary(5000, seed=23)
    queries = synth.make_typo_queries(words, 1500, seed=29)     # > one chunk, many segments
    g = compare(oracle, words, queries, ctx=ctx)
    a = g.lookup(queries[:100])
    b = g.lookup(queries[:100])
    for (a1, a2), (b1, b2) in zip(a, b):
        assert a1.tolist() == b1.tolist() and a2.tolist() == b2.tolist()
    # one query at a time gives the same answer as the batch
    for i in (0, 17, 99):
        s1, s2 = g.lookup([queries[i]])[0]
        assert s1.tolist() == a[i][0].tolist() and s2.tolist() == a[i][1].tolist()


@pytest.mark.parametrize("alphabet,seed", [("ab", 1), ("abc", 2), ("aé", 3), ("aбc日", 4)])
def test_dense_hit_regime_tiny_alphabet(ctx, oracle, alphabet, seed):
    # words over a 2-4 letter alphabet: thousands of words within distance 2 of any query, so
    # the caps, the first-letter classes, the signature/shape filters and the in-order
    # appends are all saturated (multi-byte letters take the non-ASCII tile path)
    rng = np.random.default_rng(seed)
    words = set()
    while len(words) < 6000:
        L = int(rng.integers(3, 14))
        words.add("".join(alphabet[int(i)] for i in rng.integers(0, len(alphabet), L)))
    words = sorted(words, key=lambda w: w.encode())
    queries = []
    for _ in range(120):
        L = int(rng.integers(5, 13))
        w = "".join(alphabet[int(i)] for i in rng.integers(0, len(alphabet), L))
        queries.append((w, 1 if L < 9 else 2, bool(rng.random() < 0.4)))
    queries += [(w, 2, False) for w in words[::997]] + [(w[:6], 2, True) for w in words[::1499] if len(w) >= 6]
    for caps in [(150, 50), (7, 3), (1000, 1000)]:
        compare(oracle, words, queries, caps=caps, ctx=ctx)


def test_microbatcher_fuses_concurrent_lookups(ctx, oracle):
    import threading
    words = synth.make_dictionary(8000, seed=31)
    concat, off = synth.flatten_words(words)
    odic = oracle.Dictionary.from_flat(concat, off)
    g = ma.GpuDictionary(ctx, concat=concat, offsets=off)
    g.set_microbatch(20000, 10 ** 6)
    queries = synth.make_typo_queries(words, 36 * 3, seed=33)
    out = [None] * 36
    barrier = threading.Barrier(36)

    def worker(j):
        barrier.wait()
        caps = (150, 50) if j % 4 else (9, 4)          # a second caps group in the same window
        out[j] = (caps, g.lookup(queries[3 * j:3 * j + 3], cap_one=caps[0], cap_two=caps[1]))
    th = [threading.Thread(target=worker, args=(j,)) for j in range(36)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    for j in range(36):
        caps, got = out[j]
        for (w, b, p), (g1, g2) in zip(queries[3 * j:3 * j + 3], got):
            e1, e2 = oracle.typo_lookup(odic, w, b, p, cap_one=caps[0], cap_two=caps[1])
            assert g1.tolist() == e1.tolist() and g2.tolist() == e2.tolist()
    st = g.microbatch_stats()
    assert st["fused_calls"] == 36 and st["fused_launches"] < 36, st


def test_errors(ctx):
    with pytest.raises(ma.MsiError) as e:
        ma.GpuDictionary(ctx, words=["b", "a"])
    assert "MSI_E_NOT_SORTED" in str(e.value)
    with pytest.raises(ma.MsiError):
        ma.GpuDictionary(ctx, words=["a", "a"])
    g = ma.GpuDictionary(ctx, words=[])
    assert g.lookup([("hello", 2, False)])[0][0].size == 0


def test_facet_search_values_match_the_prefix_dfa():
    """search/facet/search.rs:122-190: `fst.search(build_dfa(query, typos, is_prefix = true))` over a facet's values —
    every value with a prefix within `typos` OSA edits of the query, distance 0 included, NO first-letter rule,
    stream order.  Checked value by value against the oracle's prefix distance."""
    import meilisearch_amd as ma
    from oracle import oracle as O
    rng = np.random.default_rng(17)
    letters = list("abcdeilnorst")
    values = set()
    while len(values) < 1500:
        n = int(rng.integers(1, 14))
        w = "".join(rng.choice(letters) for _ in range(n))
        if rng.random() < 0.1:
            w += " " + "".join(rng.choice(letters) for _ in range(int(rng.integers(2, 8))))
        values.add(w)
    values |= {"é" + "".join(rng.choice(letters) for _ in range(4)) for _ in range(20)}
    values = sorted(values, key=lambda v: v.encode())
    ctx = ma.Context(0)
    d = ma.GpuDictionary(ctx, values, facet_values=True)
    queries = ["", "a", "sta", "rose", "lion", "stone", "alert", "oriental", "é", "ébcd", "xyz", "tionals", "rest in"]
    queries += [values[int(i)][: int(rng.integers(1, 9))] for i in rng.integers(0, len(values), 12)]
    checked = 0
    for q in queries:
        for typos in (0, 1, 2):
            got, trunc = d.search_values(q, typos, cap=4000)
            want = [i for i, v in enumerate(values) if O.osa_distance(q, v, prefix=True) <= typos]
            assert not trunc
            assert got.tolist() == want, (q, typos)
            checked += len(want)
    assert checked > 2000
    got, trunc = d.search_values("a", 1, cap=5)
    assert trunc and len(got) == 5
