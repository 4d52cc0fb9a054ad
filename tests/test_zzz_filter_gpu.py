"""Filter leaves on the device: msi_bits_facet_range / msi_bits_facet_in / msi_bits_geo_within behind the
filter walk of tests/toy_filter.py against the reference's own filter tests (46 cases of filters.rs) and against the
oracle (oracle/filter_oracle.py) on random corpora and random expressions."""
import json
import os
import random

import numpy as np
import pytest

import meilisearch_amd as ma
from oracle import filter_oracle as FO
from tests.toy_filter import DeviceFilter, parse
from tests.toy_milli import ToyMilli

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FIX = json.load(open(os.path.join(ROOT, "tests", "golden", "filter_fixtures.json")))


def test_reference_filter_tests_on_the_device():
    index = ToyMilli(FIX["docs"], searchable=["title", "description"])
    ctx = ma.Context(0)
    dev = DeviceFilter(ma, ctx, index)
    for case in FIX["cases"]:
        tree = ("and", [("or", [parse(f) for f in grp]) for grp in case["groups"]])
        got = dev.evaluate(tree)
        assert sorted(index.docs[d]["id"] for d in got) == case["ids"], case["name"]
    dev.close()


def filter_corpus(seed, n_docs):
    rng = random.Random(seed)
    colors = ["red", "Green", "BLUE", "blue", "ultra violet", "écru", "ecru", "e", "rouge", "réséda", ""]
    docs = []
    for i in range(n_docs):
        d = {"id": i, "title": rng.choice(["hello world", "quick fox", "lazy dog"])}
        if rng.random() < 0.85:
            d["price"] = rng.choice([0, -0.5, 1, 2, 2.5, 3, 10, 10, 99.5, 1000, 1e300, -3])
        if rng.random() < 0.8:
            d["color"] = rng.choice(colors)
        if rng.random() < 0.5:
            d["sizes"] = [rng.choice([36, 38, 40, 42, "xl", "XL", "m"]) for _ in range(rng.randint(0, 3))]
        if rng.random() < 0.3:
            d["opt"] = rng.choice([None, "", [], {}, "x", 5])
        if rng.random() < 0.7:
            d["_geo"] = {"lat": rng.uniform(-89, 89), "lng": rng.choice([rng.uniform(-180, 180), 179.9, -179.9, 2.35])}
        docs.append(d)
    return docs


LEAVES = ["price = 10", "price != 10", "price > 2", "price >= 2.5", "price < 3", "price <= -0.5", "price 1 TO 10",
          "price 10 TO 1", "price > 1e300", "price >= 1e300", "price < -3", "price = red", "color = blue", "color = Blue",
          "color != écru", "color > green", "color <= e", "color e TO f", "color STARTS WITH e", "color STARTS WITH é",
          "color STARTS WITH ''", "color CONTAINS u", "color CONTAINS violet", "color IN[red, blue, 7]", "color NOT IN[red]",
          "sizes = 38", "sizes = xl", "sizes > 38", "sizes IN[36, m]", "sizes 37 TO 41", "opt EXISTS", "opt NOT EXISTS",
          "opt IS NULL", "opt IS NOT NULL", "opt IS EMPTY", "opt = 5", "opt = x", "missing = 1", "missing EXISTS",
          "_geoRadius(48.85, 2.35, 2000000)", "_geoRadius(0, 179.95, 500000)", "_geoRadius(10, 10, 0)",
          "_geoBoundingBox([60, 40], [-10, -20])", "_geoBoundingBox([45, -170], [-45, 170])", "_geoBoundingBox([89, 180], [-89, -180])"]


@pytest.mark.parametrize("n_docs", [1, 64, 65, 700])
def test_leaves_and_expressions_match_the_oracle(n_docs):
    index = ToyMilli(filter_corpus(n_docs, n_docs), searchable=["title"])
    ctx = ma.Context(0)
    dev = DeviceFilter(ma, ctx, index)
    for f in LEAVES:
        tree = parse(f)
        assert dev.evaluate(tree) == sorted(FO.evaluate(index, tree)), f
    rng = random.Random(n_docs)
    for _ in range(60):
        def expr(depth):
            r = rng.random()
            if depth == 0 or r < 0.35:
                return rng.choice(LEAVES)
            if r < 0.5:
                return "NOT (" + expr(depth - 1) + ")"
            op = " AND " if r < 0.75 else " OR "
            return "(" + op.join(expr(depth - 1) for _ in range(rng.randint(2, 3))) + ")"
        f = expr(3)
        tree = parse(f)
        assert dev.evaluate(tree) == sorted(FO.evaluate(index, tree)), f
    dev.close()


def test_kernel_against_numpy_and_argument_checks():
    ctx = ma.Context(0)
    rng = np.random.default_rng(3)
    n = 5003
    per_doc = [sorted(set(int(x) for x in rng.integers(0, 200, rng.integers(0, 4)))) for _ in range(n)]
    fk = ma.FacetKeys(ctx, per_doc)
    pool = ma.BitsPool(ctx, n, 3)
    for lo, hi in ((0, 199), (50, 60), (60, 50), (199, 2 ** 64 - 1), (0, 0)):
        pool.fill(0, True)
        pool.facet_range(fk, lo, hi, 0)
        assert pool.to_docids(0).tolist() == [d for d, v in enumerate(per_doc) if any(lo <= x <= hi for x in v)]
    pool.set_from_docids(1, np.array([1, 2, 3], dtype=np.uint32))
    pool.facet_range(fk, 10, 12, 1, accumulate=True)
    assert pool.to_docids(1).tolist() == sorted({1, 2, 3} | {d for d, v in enumerate(per_doc) if any(10 <= x <= 12 for x in v)})
    for keys in ([], [7], [3, 50, 51, 199, 4000]):
        pool.fill(0, True)
        pool.facet_in(fk, keys, 0)
        assert pool.to_docids(0).tolist() == [d for d, v in enumerate(per_doc) if any(x in keys for x in v)]
    xs = [-1e300, -2.5, -0.0, 0.0, 1e-300, 1.0, 1.5, 1e300]
    ks = [ma.facet_number_key(x) for x in xs]
    assert ks == sorted(ks) and ks[2] == ks[3]     # the order of the doubles; -0.0 and +0.0 share one key (f64_into_bytes, facet/value_encoding.rs:5-7)
    assert len(set(ks)) == len(ks) - 1
    with pytest.raises(ma.MsiError):
        pool.facet_in(fk, [5, 5], 0)
    with pytest.raises(ma.MsiError):
        pool.facet_range(ma.FacetKeys(ctx, per_doc + [[]]), 0, 1, 0)


def test_vectors_filter_leaf(ctx):
    """`_vectors` conditions (filter/vector.rs:49-158): the items of f32 and binary-quantised stores as docid sets on the
    device, every variant, two embedders OR-ed and intersected with a universe — against the plain-set restatement."""
    rng = np.random.default_rng(21)
    n_docs = 70_000
    pool = ma.BitsPool(ctx, n_docs, 8)

    def store(kind, p):
        ids = np.nonzero(rng.random(n_docs) < p)[0].astype(np.uint32)
        st = ma.GpuStore(ctx, 8) if kind == "f32" else ma.GpuBqStore(ctx, 8)
        st.upload(ids, rng.standard_normal((ids.size, 8)).astype(np.float32))
        return st, set(ids.tolist())

    a0, a0_ids = store("f32", 0.3)
    a1, a1_ids = store("f32", 0.01)
    b0, b0_ids = store("bq", 0.2)
    up_a = set(rng.choice(sorted(a0_ids), 500, replace=False).tolist()) | {5, 69_999}
    sr_a = set(rng.choice(n_docs, 3000, replace=False).tolist())
    up_b, sr_b = set(), set(rng.choice(sorted(b0_ids), 50, replace=False).tolist())
    universe = set(rng.choice(n_docs, 40_000, replace=False).tolist())
    UP_A, SR_A, UP_B, SR_B, UNI, DST, SCR = 0, 1, 2, 3, 4, 5, 6
    for slot, ids in ((UP_A, up_a), (SR_A, sr_a), (UP_B, up_b), (SR_B, sr_b), (UNI, universe)):
        pool.set_from_docids(slot, sorted(ids))
    for kind in ("none", "fragment", "documentTemplate", "userProvided", "regenerate"):
        for frag_a in (False, True):
            # embedder A: two f32 stores (a fragment condition looks at one of them), embedder B: one binary-quantised store
            a_stores = [a1] if kind == "fragment" else [a0, a1]
            a_items = [a1_ids] if kind == "fragment" else [a0_ids, a1_ids]
            pool.vector_filter(DST, kind, stores=a_stores, has_fragments=frag_a, user_provided=UP_A, skip_regenerate=SR_A,
                               scratch=SCR)
            assert set(pool.to_docids(DST).tolist()) == FO.vector_filter(kind, frag_a, a_items, up_a, sr_a), (kind, frag_a)
            pool.vector_filter(DST, kind, bq_stores=[b0], has_fragments=False, user_provided=UP_B, skip_regenerate=SR_B,
                               scratch=SCR, accumulate=True)
            pool.op(DST, DST, UNI, 0)  # AND
            want = FO.vector_filter_all([(frag_a, a_items, up_a, sr_a), (False, [b0_ids], up_b, sr_b)], kind, universe)
            assert set(pool.to_docids(DST).tolist()) == want, (kind, frag_a)
    with pytest.raises(ma.MsiError):
        pool.vector_filter(DST, "none", stores=[a0], scratch=DST)
    pool.close()
