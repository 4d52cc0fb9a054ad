"""GeoSort on the device: msi_bits_geo_next against the bucket rule of documents/geo_sort.rs:150-224
restated over exact distances, then the rule inside the ranked keyword search against the reference's geo_sort.rs
snapshots and the oracle.  The same bodies run in the CPU tier on the emulated kernels (tests/test_kernels_emulated_cpu.py)."""
import json
import os

import numpy as np
import pytest

import meilisearch_amd as ma
from oracle import ranking_oracle as RO

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def make_points(rng, n_docs):
    """A fifth without _geo; places with many documents (same point), clusters a few decimetres to metres apart
    (the error margin), the poles, the antimeridian."""
    pts = np.full((n_docs, 2), np.nan)
    places = np.column_stack([rng.uniform(-89, 89, 12), rng.uniform(-179, 179, 12)])
    for d in range(n_docs):
        r = rng.random()
        if r < 0.2:
            continue
        if r < 0.5:
            pts[d] = places[rng.integers(0, len(places))]
        elif r < 0.7:
            base = places[rng.integers(0, len(places))]
            pts[d] = base + np.array([rng.choice([0, 2e-6, 5e-6, 2e-5, 1e-4]), 0.0])
        elif r < 0.72:
            pts[d] = [rng.choice([90.0, -90.0]), rng.uniform(-180, 180)]
        elif r < 0.74:
            pts[d] = [rng.uniform(-5, 5), rng.choice([179.9999, -179.9999, 180.0, -180.0])]
        else:
            pts[d] = [rng.uniform(-90, 90), rng.uniform(-180, 180)]
    return pts


def expected_buckets(pts, universe, target, ascending, cap, margin):
    """next_bucket over a cache in exact (distance, docid) order: [(first docid | None, sorted docids)]"""
    left = sorted(universe)
    geo = [d for d in left if not np.isnan(pts[d][0])]
    dist = {d: RO.distance_between_two_points(target, pts[d]) for d in geo}
    order = sorted(geo, key=(lambda d: (dist[d], d)) if ascending else (lambda d: (-dist[d], d)))
    out, i = [], 0
    while i < len(order):
        d0 = dist[order[i]]
        j = i
        while j < len(order) and abs(d0 - dist[order[j]]) <= margin and j - i < cap:
            j += 1
        out.append((order[i], sorted(order[i:j])))
        i = j
    rest = sorted(set(left) - set(geo))
    out.append((None, rest))      # what has no point: the whole remaining universe, value None
    return out


@pytest.mark.parametrize("ascending", [True, False], ids=["asc", "desc"])
@pytest.mark.parametrize("n_docs", [1, 63, 64, 65, 1000, 3001])
def test_geo_next_against_the_bucket_rule(n_docs, ascending):
    ctx = ma.Context(0)
    rng = np.random.default_rng(n_docs * 2 + ascending)
    pts = make_points(rng, n_docs)
    gp = ma.GeoPoints(ctx, pts)
    pool = ma.BitsPool(ctx, n_docs, 4)
    for target, cap, margin in (((48.85, 2.35), 1000, 1.0), ((0.0, 0.0), 3, 1.0), ((-33.9, 151.2), 1000, 0.0),
                                ((89.5, -170.0), 7, 30.0)):
        universe = np.nonzero(rng.random(n_docs) < 0.8)[0].astype(np.uint32)
        pool.set_from_docids(0, universe)
        left = set(universe.tolist())
        for first, docs in expected_buckets(pts, universe.tolist(), target, ascending, cap, margin):
            pool.fill(1, True)               # stale content of the output slots must not survive
            pool.fill(2, True)
            got_first, n = pool.geo_next(gp, 0, 1, 2, target[0], target[1], ascending, cap, margin)
            if first is None:
                assert got_first is None and n == 0
                assert pool.to_docids(0).tolist() == docs          # the universe is left as it is
                break
            assert (got_first, n) == (first, len(docs)), (target, cap, margin)
            assert pool.to_docids(1).tolist() == docs
            left -= set(docs)
            assert pool.to_docids(0).tolist() == sorted(left)
    with pytest.raises(ma.MsiError):
        pool.geo_next(ma.GeoPoints(ctx, np.zeros((n_docs + 1, 2))), 0, 1, 2, 0.0, 0.0)
    with pytest.raises(ma.MsiError):
        pool.geo_next(gp, 0, 1, 1, 0.0, 0.0)


# ---- GeoSort inside the ranked keyword search, on the device -------------------------------------------------------------
def test_geo_sort_rs_on_the_device():
    import tests.test_search_hostlogic_cpu as H
    from tests.test_zzz_distinct_gpu import device_lib
    H.test_geo_sort_rs_through_the_host_logic(device_lib())


def test_geo_sort_matches_the_oracle_on_the_device():
    import tests.test_search_hostlogic_cpu as H
    from tests.test_zzz_distinct_gpu import device_lib
    H.test_geo_sort_matches_the_oracle(device_lib())
