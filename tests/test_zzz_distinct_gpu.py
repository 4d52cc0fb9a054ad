"""`distinct` on the device: msi_bits_distinct / msi_bits_distinct_excluded /
msi_bits_andnot_many_count against the reference's sequential loop (search/new/distinct.rs:19-62 restated below),
then distinct inside the ranked keyword search against the reference's distinct.rs snapshots and the oracle.
The same bodies run in the CPU tier on the emulated kernels (tests/test_kernels_emulated_cpu.py)."""
import json
import os

import numpy as np
import pytest

import meilisearch_amd as ma
from meilisearch_amd import bits as B

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def sequential_distinct(per_doc, candidates):
    """apply_distinct_rule, distinct.rs:19-36 -> (remaining, excluded)."""
    holders = {}
    for d, vs in enumerate(per_doc):
        for v in vs:
            holders.setdefault(v, set()).add(d)
    remaining, excluded = [], set()
    for d in sorted(candidates):
        if d in excluded:
            continue
        for v in per_doc[d]:
            excluded |= holders[v]
        remaining.append(d)
    return remaining, excluded


def make_values(rng, n_docs, kind):
    if kind == "single":          # one value per document, a fifth without
        n_values = max(1, n_docs // 4)
        per_doc = [[int(rng.integers(0, n_values))] if rng.random() < 0.8 else [] for _ in range(n_docs)]
    elif kind == "multi":         # 0..3 values: rounds > 1
        n_values = max(1, n_docs // 3)
        per_doc = [sorted(set(int(x) for x in rng.integers(0, n_values, rng.integers(0, 4)))) for _ in range(n_docs)]
    elif kind == "chain":         # document i holds {i, i + 1}: one round per two documents -> the sequential kernel
        n_values = n_docs + 1
        per_doc = [[i, i + 1] for i in range(n_docs)]
    else:                         # everything shares one value
        n_values = 1
        per_doc = [[0] for _ in range(n_docs)]
    return per_doc, n_values


@pytest.mark.parametrize("kind", ["single", "multi", "chain", "same"])
@pytest.mark.parametrize("n_docs", [1, 63, 64, 65, 1000, 5003])
def test_distinct_against_the_sequential_loop(n_docs, kind):
    ctx = ma.Context(0)
    rng = np.random.default_rng(n_docs * 7 + len(kind))
    per_doc, n_values = make_values(rng, n_docs, kind)
    dv = ma.DocValues(ctx, per_doc, n_values)
    pool = ma.BitsPool(ctx, n_docs, 6)
    for density in (1.0, 0.5, 0.02):
        cand = np.nonzero(rng.random(n_docs) < density)[0].astype(np.uint32)
        want_rem, want_exc = sequential_distinct(per_doc, cand.tolist())
        for with_excluded in (True, False):
            pool.set_from_docids(0, cand)
            pool.fill(1, True)                       # stale content of the output slots must not survive
            pool.fill(2, True)
            n, rounds, sequential = pool.distinct(dv, 0, 1, 2 if with_excluded else B.NO_UNIVERSE)
            assert pool.to_docids(1).tolist() == want_rem
            assert n == len(want_rem)
            assert pool.count(0) == 0                # candidates are consumed
            if with_excluded:
                assert pool.to_docids(2).tolist() == sorted(want_exc)
            if kind in ("single", "same") or not len(cand):
                assert rounds == 1 and not sequential
            if kind == "chain" and density == 1.0 and n_docs >= 1000:
                assert sequential                    # n_docs / 2 rounds needed: the single-thread kernel finished it
        # the rule-less path keeps a prefix of `remaining`: excluded of a subset
        kept = want_rem[:max(1, len(want_rem) // 3)]
        pool.set_from_docids(3, np.array(kept, dtype=np.uint32))
        pool.fill(4, True)
        pool.distinct_excluded(dv, 3, 4)
        holders = sequential_distinct(per_doc, kept)[1]
        assert pool.to_docids(4).tolist() == sorted(holders)
    # universes of the rule stack: slots[i] &= ~removed with their cardinalities
    sets = [set(np.nonzero(rng.random(n_docs) < p)[0].tolist()) for p in (0.9, 0.5, 0.1, 0.0)]
    for i, s_ in enumerate(sets):
        pool.set_from_docids(i, np.array(sorted(s_), dtype=np.uint32))
    removed = set(np.nonzero(rng.random(n_docs) < 0.3)[0].tolist())
    pool.set_from_docids(5, np.array(sorted(removed), dtype=np.uint32))
    counts = pool.andnot_many_count(5, [0, 1, 2, 3])
    for i, s_ in enumerate(sets):
        assert pool.to_docids(i).tolist() == sorted(s_ - removed)
        assert counts[i] == len(s_ - removed)
    with pytest.raises(ma.MsiError):
        pool.distinct(ma.DocValues(ctx, per_doc + [[]], n_values), 0, 1, 2)     # another index's table
    with pytest.raises(ma.MsiError):
        pool.distinct(dv, 0, 0, 2)
    with pytest.raises(ma.MsiError):
        ma.DocValues(ctx, [[n_values]], n_values)                               # value id out of range


def test_many_calls_share_the_scratch_without_clearing_it():
    """The per-pool scratch is never cleared between calls (stamps): 300 calls over changing candidates and two
    value tables of different sizes."""
    ctx = ma.Context(0)
    rng = np.random.default_rng(5)
    n_docs = 700
    tables = []
    for kind in ("single", "multi"):
        per_doc, n_values = make_values(rng, n_docs, kind)
        tables.append((per_doc, ma.DocValues(ctx, per_doc, n_values)))
    pool = ma.BitsPool(ctx, n_docs, 3)
    for it in range(300):
        per_doc, dv = tables[it % 2 if it > 20 else 0]
        cand = np.nonzero(rng.random(n_docs) < rng.random())[0].astype(np.uint32)
        pool.set_from_docids(0, cand)
        n, _, _ = pool.distinct(dv, 0, 1, 2)
        want_rem, want_exc = sequential_distinct(per_doc, cand.tolist())
        assert pool.to_docids(1).tolist() == want_rem and n == len(want_rem)
        assert pool.to_docids(2).tolist() == sorted(want_exc)


# ---- distinct inside the ranked keyword search, on the device ---------------------------------------------------------
def device_lib():
    """What the host-logic test bodies take as their library: every object is the product's, on the device."""
    from types import SimpleNamespace
    from meilisearch_amd import _lib
    import tests.test_search_hostlogic_cpu as H

    class DeviceHarness(H.MockHarness):
        def dict_create(self, index):
            return ma.GpuDictionary(self.L.ctx, [w.encode() for w in index.words])

        def dict_destroy(self, d):
            d.close()

        def pool_create(self, n_docs, n_slots):
            return ma.BitsPool(self.L.ctx, n_docs, n_slots, private_stream=getattr(self.L, "private_streams", False))

        def pool_destroy(self, pool):
            pool.close()

        def keys_create(self, arr):
            return ma.DocKeys(self.L.ctx, arr)

        def keys_destroy(self, h):
            h.close()

        def values_create(self, per_doc, n_values):
            return ma.DocValues(self.L.ctx, per_doc, n_values)

        def values_destroy(self, h):
            h.close()

        def points_create(self, lat_lng):
            return ma.GeoPoints(self.L.ctx, lat_lng)

        def points_destroy(self, h):
            h.close()

    return SimpleNamespace(ctx=ma.Context(0), harness_cls=DeviceHarness,
                           msi_keyword_search_ranked=_lib.lib().msi_keyword_search_ranked)


def test_reference_snapshots_with_distinct_and_sort_on_the_device(monkeypatch):
    """All 108 reference searches — the 9 of distinct.rs and the 5 of sort.rs included — through libmsi on the GPU."""
    import tests.test_search_hostlogic_cpu as H
    H.test_reference_snapshots_through_the_host_logic(device_lib(), monkeypatch, "1", "1")


def test_distinct_matches_the_oracle_on_the_device(monkeypatch):
    import tests.test_search_hostlogic_cpu as H
    H.test_distinct_matches_the_oracle(device_lib(), monkeypatch, "1")


def test_reference_criteria_tests_on_the_device():
    """query_criteria.rs: 14 criterion cases + every 4th of the 120 criteria orders over test_set.ndjson (synonyms, real text)."""
    import tests.test_search_hostlogic_cpu as H
    H.test_reference_criteria_tests_through_the_host_logic(device_lib(), every=4)


def test_reference_distinct_integration_tests_on_the_device():
    import tests.test_search_hostlogic_cpu as H
    H.test_reference_distinct_integration_tests(device_lib())


def test_reference_typo_tolerance_and_phrase_integration_tests_on_the_device():
    import tests.test_search_hostlogic_cpu as H
    H.test_reference_typo_tolerance_and_phrase_integration_tests(device_lib())


def test_concurrent_searches_with_distinct_sort_and_geo():
    """One pool with a private stream per caller thread, the per-document arrays shared: 4 threads x 14 searches with
    `distinct`, Sort and GeoSort rules equal the single-threaded results (the distinct scratch, the GeoSort cells and
    the completion signals are per pool)."""
    import threading
    import tests.test_search_hostlogic_cpu as H
    from tests.toy_milli import ToyMilli
    index = ToyMilli(H.geo_corpus(11, 300), searchable=["title", "body"])
    L = device_lib()
    L.private_streams = True
    jobs = []
    for criteria, sort, geo in H.GEO_SETUPS[:4]:
        for q, distinct in (("", "color"), ("the", None), ("quick fox", "sizes"), ("sun fl", "color")):
            jobs.append((q, criteria, sort, distinct, geo))
    jobs = jobs[:14]
    base = H.make_harness(L, index)
    want = [base.search(q, criteria=c, sort=s, distinct=d, limit=25, detailed=True, **g) for q, c, s, d, g in jobs]
    base.close()
    errors = []

    def worker(k):
        try:
            h = H.make_harness(L, index)
            for rep in range(2):
                for j in range(len(jobs)):
                    q, c, s, d, g = jobs[(j + k) % len(jobs)]
                    got = h.search(q, criteria=c, sort=s, distinct=d, limit=25, detailed=True, **g)
                    if got != want[(j + k) % len(jobs)]:
                        errors.append((k, q, c, s, d))
            h.close()
        except Exception as e:  # noqa: BLE001
            errors.append((k, repr(e)))
    threads = [threading.Thread(target=worker, args=(k,)) for k in range(4)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors[:3]
