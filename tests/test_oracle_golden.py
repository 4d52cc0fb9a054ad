"""Pins the CPU oracle against the literals the reference's own tests hold for
this path (SURVEY.md §8 c).  Runs without a GPU."""
import numpy as np
import pytest

f32 = np.float32

# crates/meilisearch/tests/search/hybrid.rs:47-68 (docs), :296-406,:547,:758 (scores)
# crates/meilisearch/tests/similar/mod.rs:19-43 (docs), :281-335 (scores, query = id 143)
SIM_GOLDENS = [
    ([1, 1], [2, 3], 0.990290343761444), ([1, 1], [1, 2], 0.974341630935669),
    ([1, 1], [1, 3], 0.9472135901451112),
    ([1, 0], [2, 3], 0.7773500680923462), ([1, 0], [1, 2], 0.7236068248748779),
    ([1, 0], [1, 3], 0.6581138968467712),
    ([-0.5, 0.3, 0.85], [0.1, 0.6, 0.8], 0.890957772731781),
    ([-0.5, 0.3, 0.85], [0.6, 0.8, -0.2], 0.39060014486312866),
    ([-0.5, 0.3, 0.85], [0.7, 0.7, -0.4], 0.2819308042526245),
    ([-0.5, 0.3, 0.85], [0.8, 0.4, -0.5], 0.1662663221359253),
]


@pytest.mark.parametrize("q,x,gold", SIM_GOLDENS)
def test_similarity_literals_bit_exact(oracle, q, x, gold):
    d = oracle.cosine_distance(q, x)
    sim = oracle.similarity(d)
    # _rankingScore is the f32 similarity widened to f64 and printed
    assert f32(sim) == f32(gold), (sim, gold)
    assert abs(sim - gold) <= 1e-5  # north_star tolerance


def test_distribution_shift_literals(oracle):
    # hybrid.rs:540-568: mean 0.998, sigma 0.01 over the [1,1] query scores
    sims = [oracle.similarity(oracle.cosine_distance([1, 1], x)) for x in ([2, 3], [1, 2], [1, 3])]
    got = [oracle.distribution_shift(0.998, 0.01, s) for s in sims]
    assert f32(got[0]) == f32(0.19161224365234375)
    assert f32(got[1]) == f32(1.1920928955078125e-7)
    assert f32(got[2]) == f32(1.1920928955078125e-7)


def test_tie_order_and_missing_vectors(oracle):
    # crates/milli/src/search/new/tests/cutoff.rs:507-626: query [1,-1]; docs 0..3
    # have vectors, doc 4 has none -> IDs [2,0,3,1], similarities 1.0,0.5,0.5,0.0
    rows = np.array([[0.1, 0.1], [-0.1, 0.1], [0.1, -0.1], [-0.1, -0.1]], dtype=f32)
    ids, dist = oracle.vs_topk(rows, np.arange(4, dtype=np.uint32), [1, -1], 10)
    assert ids.tolist() == [2, 0, 3, 1]
    sims = [oracle.similarity(float(d)) for d in dist]
    assert np.allclose(sims, [1.0, 0.5, 0.5, 0.0], atol=1e-6)


def test_rank_merge_literals(oracle):
    # cutoff.rs:111-172: Words{3,3}+Typo{k,3} and Words{2,3}+Typo{0,2}
    def words(m, mx):
        return (m, mx)

    def typo(t, mx):
        return (mx + 1 - t, mx + 1)  # Typo::rank, score_details.rs:492-497

    cases = [([words(3, 3), typo(0, 3)], "1.0000"), ([words(3, 3), typo(1, 3)], "0.9167"),
             ([words(3, 3), typo(2, 3)], "0.8333"), ([words(2, 3), typo(0, 2)], "0.6667")]
    for pairs, lit in cases:
        assert f"{oracle.rank_global_score(pairs):.4f}" == lit
    assert oracle.rank_global_score([]) == 1.0


def test_compare_scores(oracle):
    # hybrid.rs:32-80
    assert oracle.compare_scores([0.5], 0.5, [0.5], 0.5) == 0
    assert oracle.compare_scores([0.9], 0.5, [0.5], 0.5) == 1
    assert oracle.compare_scores([0.5, 0.1], 1.0, [0.5, 0.2], 1.0) == -1
    assert oracle.compare_scores([0.5], 1.0, [0.5, 0.2], 1.0) == -1
    assert oracle.compare_scores([], 1.0, [], 1.0) == 0
    # the f64::EPSILON window
    assert oracle.compare_scores([0.5 + 1e-17], 1.0, [0.5], 1.0) == 0


def test_typo_budget_thresholds(oracle):
    # parse_query.rs:408-478: thresholds count chars (5 / 9), not bytes
    assert oracle.typo_budget("dogg") == 0
    assert oracle.typo_budget("doggy") == 1
    assert oracle.typo_budget("café") == 0       # 4 chars, 5 bytes
    assert oracle.typo_budget("собак") == 1      # 5 chars, 10 bytes
    assert oracle.typo_budget("sobakasob") == 2
    assert oracle.typo_budget("a" * 251) == 0


def test_osa_distance_semantics(oracle):
    # restricted Damerau: a transposition costs 1, but edited substrings are not re-edited
    assert oracle.osa_distance("ca", "abc") == 3
    assert oracle.osa_distance("quick", "quikc") == 1
    assert oracle.osa_distance("quack", "quick") == 1
    assert oracle.osa_distance("quicest", "quickest") == 1
    assert oracle.osa_distance("jummps", "jumps") == 1
    assert oracle.osa_distance("héllo", "hello") == 1   # chars, not bytes
    assert oracle.osa_distance("quic", "quickest", prefix=True) == 0
    assert oracle.osa_distance("quac", "quickest", prefix=True) == 1


# words of the typo tests: crates/milli/src/search/new/tests/typo.rs:30-170 (corpus)
TYPO_DICT = sorted(set(
    "the quick brown fox jumps over the lazy dog quickest quickly quack quickbrownfox "
    "brow browny brownie foxes jumped jump jumper lazily dogs zeal zealand zealot zoo "
    "netwolk network wolk wol".split()), key=lambda w: w.encode())


def test_typo_lookup_reference_words(oracle):
    dic = oracle.Dictionary(TYPO_DICT)

    def one_two(word, budget, prefix=False):
        a, b = oracle.typo_lookup(dic, word, budget, prefix)
        return [TYPO_DICT[i] for i in a], [TYPO_DICT[i] for i in b]

    # typo.rs:176-233 test_default_typo: 1 typo: replace / missing / extra letter
    assert "quick" in one_two("quack", 1)[0]
    assert "quickest" in one_two("quicest", 1)[0]
    assert "jumps" in one_two("jummps", 1)[0]
    # typo.rs property 6: a typo on the first letter counts as two typos
    one, two = one_two("netwolk", 1)
    assert "network" in one
    one, two = one_two("zuickest", 2)
    assert "quickest" in two and "quickest" not in one
    one, _ = one_two("zuickest", 1)
    assert one == []
    # exact match is never a derivation (d == 0 ignored, compute_derivations.rs:91,148)
    assert "quick" not in one_two("quick", 1)[0]


def test_typo_cap_interplay(oracle):
    # compute_derivations.rs:129-163 with small caps: once `two` is full, an
    # other-first-letter word is classified by the 2-typo DFA and lands in `one`
    words = sorted(["aello", "bello", "cello", "dello", "hallo", "hella", "hello", "hellos", "jello"],
                   key=lambda w: w.encode())
    dic = oracle.Dictionary(words)
    one, two = oracle.typo_lookup(dic, "hello", 2, False, cap_one=3, cap_two=2)
    assert [words[i] for i in two] == ["aello", "bello"]
    # cello, dello arrive after `two` is full -> distance 1 -> `one`; then hallo fills it
    assert [words[i] for i in one] == ["cello", "dello", "hallo"]


def test_norms_arroy_stored_for_real_embeddings():
    """The reference's own index (v1.12 upgrade test) holds two 384-d all-MiniLM-L6-v2 embeddings as arroy item
    leaves {norm: f32, vector}: the oracle's f32 norm (sequential accumulation, the crate's scalar path) must equal
    the norm arroy computed at indexing time, bit for bit; and the distance between the two documents is pinned to
    the f64 value within the north-star tolerance."""
    import ctypes as C
    import json
    import os
    import numpy as np
    from oracle import oracle as O
    fx = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "index_blobs.json")))
    items = fx["arroy_items"]
    assert [it["docid"] for it in items] == [0, 1]
    vecs = [np.frombuffer(bytes.fromhex(it["vector_f32_hex"]), dtype="<f4") for it in items]
    lib = O.lib()
    lib.orc_norm_f32.restype = C.c_float
    for it, v in zip(items, vecs):
        stored = np.frombuffer(bytes.fromhex(it["norm_f32_hex"]), dtype="<f4")[0]
        v = np.ascontiguousarray(v)
        got = np.float32(lib.orc_norm_f32(v.ctypes.data_as(C.POINTER(C.c_float)), C.c_uint32(v.size)))
        assert got.view(np.uint32) == stored.view(np.uint32)
    a, b = (v.astype(np.float64) for v in vecs)
    exact = (1.0 - a @ b / np.sqrt((a @ a) * (b @ b))) / 2.0
    assert abs(O.cosine_distance(vecs[0], vecs[1]) - exact) < 1e-6
    assert 0.0 < exact < 0.5       # two dog descriptions: similar, not identical
