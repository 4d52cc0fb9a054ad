"""tests/golden/: (1) the reference's own literals pin the oracle (CPU); (2) the HIP
path reproduces the committed oracle outputs and the literals through the C ABI (GPU)."""
import json
import os

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = os.path.join(HERE, "golden")
f32 = np.float32


def literals():
    return json.load(open(os.path.join(GOLD, "reference_literals.json")))


# ------------------------------------------------------------------ CPU: oracle vs literals

def test_oracle_matches_reference_literals(oracle):
    L = literals()
    for e in L["similarity"]:
        sim = oracle.similarity(oracle.cosine_distance(e["q"], e["x"]))
        assert f32(sim) == f32(e["score"]), e
        assert abs(sim - e["score"]) <= 1e-5
    ds = L["distribution_shift"]
    for x, y in zip(ds["in"], ds["out"]):
        assert f32(oracle.distribution_shift(ds["mean"], ds["sigma"], x)) == f32(y)
    t = L["tie_order"]
    ids, dist = oracle.vs_topk(np.array(t["rows"], dtype=f32), np.arange(4, dtype=np.uint32), t["q"], 10)
    assert ids.tolist() == t["ids"]
    for e in L["rank_global_score"]:
        assert f"{oracle.rank_global_score([tuple(p) for p in e['ranks']]):.4f}" == e["score"]
    for e in L["typo_budget"]:
        if "word" in e:
            assert oracle.typo_budget(e["word"]) == e["budget"], e
    tw = L["typo_words"]
    words = sorted(set(tw["dictionary"].split()), key=lambda w: w.encode())
    odic = oracle.Dictionary(words)
    for c in tw["cases"]:
        one, two = oracle.typo_lookup(odic, c["q"], c["typos"], c.get("prefix", False))
        one_w = [words[i] for i in one]
        two_w = [words[i] for i in two]
        for w in c.get("one_contains", []):
            assert w in one_w, c
        for w in c.get("two_contains", []):
            assert w in two_w, c
        for w in c.get("one_excludes", []):
            assert w not in one_w, c
        for w in c.get("two_excludes", []):
            assert w not in two_w, c


def test_fixtures_are_reproducible(oracle):
    """The committed oracle outputs are what the oracle computes today."""
    from meilisearch_amd import synth
    z = np.load(os.path.join(GOLD, "oracle_vs_topk.npz"))
    rows = synth.make_embeddings(int(z["n"]), int(z["dim"]), seed=int(z["seed_rows"]))
    ids = (np.arange(int(z["n"]), dtype=np.uint32) * 5 + 2)
    qs = synth.make_embeddings(8, int(z["dim"]), seed=int(z["seed_queries"]))
    for j in (0, 7):
        a, b = oracle.vs_topk(rows, ids, qs[j], int(z["k"]))
        assert a.tolist() == z["ids"][j].tolist()
        assert b.view(np.uint32).tolist() == z["dist"][j].view(np.uint32).tolist()


# ------------------------------------------------------------------ GPU: HIP path vs fixtures

@pytest.mark.gpu
def test_gpu_vs_topk_fixture(ctx):
    import meilisearch_amd as ma
    from meilisearch_amd import synth
    z = np.load(os.path.join(GOLD, "oracle_vs_topk.npz"))
    n, dim, k = int(z["n"]), int(z["dim"]), int(z["k"])
    rows = synth.make_embeddings(n, dim, seed=int(z["seed_rows"]))
    ids = (np.arange(n, dtype=np.uint32) * 5 + 2)
    qs = synth.make_embeddings(8, dim, seed=int(z["seed_queries"]))
    st = ma.GpuStore(ctx, dim)
    st.upload(ids, rows)
    d, s, c = st.search(qs, k)
    assert (d == z["ids"]).all()
    assert (s.view(np.uint32) == z["dist"].view(np.uint32)).all()
    fb, nb = ma.dense_filter(z["allowed"], nbits=int(ids.max()) + 1)
    d, s, c = st.search(qs, k, fb, nb)
    assert (d == z["filtered_ids"]).all()
    assert (s.view(np.uint32) == z["filtered_dist"].view(np.uint32)).all()


@pytest.mark.gpu
def test_gpu_typo_fixture(ctx):
    import meilisearch_amd as ma
    from meilisearch_amd import synth
    g = json.load(open(os.path.join(GOLD, "oracle_typo_lookup.json")))
    words = synth.make_dictionary(g["n_words"], seed=g["seed_dictionary"])
    gd = ma.GpuDictionary(ctx, words=words)
    got = gd.lookup([(w, b, p) for w, b, p in g["queries"]])
    for (g1, g2), e1, e2 in zip(got, g["one"], g["two"]):
        assert g1.tolist() == e1 and g2.tolist() == e2


@pytest.mark.gpu
def test_gpu_reference_literals(ctx):
    import meilisearch_amd as ma
    L = literals()
    for e in L["similarity"]:
        st = ma.GpuStore(ctx, len(e["q"]))
        st.upload([7], [e["x"]])
        d, s, c = st.search(np.array([e["q"]], dtype=f32), 1)
        assert d[0, 0] == 7 and f32(1.0) - s[0, 0] == f32(e["score"])
    tw = L["typo_words"]
    words = sorted(set(tw["dictionary"].split()), key=lambda w: w.encode())
    gd = ma.GpuDictionary(ctx, words=words)
    for c in tw["cases"]:
        if c["typos"] == 1:
            one, two = gd.find_one_typo_derivations(c["q"], c.get("prefix", False)), []
        else:
            one, two = gd.find_one_two_typo_derivations(c["q"], c.get("prefix", False))
        for w in c.get("one_contains", []):
            assert w in one, c
        for w in c.get("two_contains", []):
            assert w in two, c
        for w in c.get("one_excludes", []):
            assert w not in one, c
        for w in c.get("two_excludes", []):
            assert w not in two, c
