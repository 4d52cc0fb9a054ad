"""SURVEY §8 f4 — binary-quantised stores on the device (msi_bq) against the oracle (orc_bq_topk).
Pinned by the reference: the quantisation (tests/vector/binary_quantized.rs:67-135: [-1.2, -2.3, 3.2] reads back as
[0, 0, 1], [2.5, 1.5, -130] as [1, 1, 0]).  The distance VALUE (hamming / dim) is restated from the published
definitions of hannoy's Hamming / arroy's BinaryQuantizedCosine: parity unpinned for it (oracle header)."""
import numpy as np
import pytest

import meilisearch_amd as ma
from meilisearch_amd import synth

pytestmark = pytest.mark.gpu


def test_reference_quantisation_literals(ctx):
    st = ma.GpuBqStore(ctx, 3)
    st.upload([0, 1], [[-1.2, -2.3, 3.2], [2.5, 1.5, -130]])
    assert st.get_vector(0).tolist() == [0.0, 0.0, 1.0]
    assert st.get_vector(1).tolist() == [1.0, 1.0, 0.0]
    assert st.get_vector(7) is None
    # query bits 111: doc 0 = 001 (2 off), doc 1 = 110 (1 off)
    d, s, c = st.search(np.array([[1, 1, 1]], dtype=np.float32), 5)
    assert c[0] == 2 and d[0, :2].tolist() == [1, 0] and np.allclose(s[0, :2], [1 / 3, 2 / 3])


@pytest.mark.parametrize("n,dim,k", [(1, 3, 2), (70, 64, 10), (1000, 96, 20), (5003, 130, 50), (20000, 768, 20), (3000, 8, 500)])
def test_random_vs_oracle(ctx, oracle, n, dim, k):
    rows = synth.make_embeddings(n, dim, seed=n + dim)
    ids = np.arange(n, dtype=np.uint32) * 3 + 7
    qs = synth.make_embeddings(40, dim, seed=1000 + n)       # > 32: two sweeps
    st = ma.GpuBqStore(ctx, dim)
    st.upload(ids, rows)
    assert len(st) == n
    d, s, c = st.search(qs, k)
    for j in range(qs.shape[0]):
        e_ids, e_dist = oracle.bq_topk(rows, ids, qs[j], k)
        m = int(c[j])
        assert m == e_ids.size
        assert d[j, :m].tolist() == e_ids.tolist(), (j, d[j, :8], e_ids[:8])
        assert s[j, :m].view(np.uint32).tolist() == e_dist.view(np.uint32).tolist()


def test_filter_ties_and_zero_components(ctx, oracle):
    # low dimension: 16 codes only, thousands of ties at every distance -> the docid tie rule decides everything
    rng = np.random.default_rng(5)
    rows = rng.integers(-1, 2, (6000, 4)).astype(np.float32)          # zeros quantise to bit 0 (x > 0 is false)
    ids = np.arange(6000, dtype=np.uint32)
    st = ma.GpuBqStore(ctx, 4)
    st.upload(ids, rows)
    allowed = np.nonzero(rng.random(6000) < 0.3)[0]
    fb, nb = ma.dense_filter(allowed.tolist(), 6000)
    qs = np.array([[1, 1, 1, 1], [0, 0, 0, 0], [-1, 1, 0, 2]], dtype=np.float32)
    for flt in (None, (fb, nb)):
        d, s, c = st.search(qs, 100, *(flt or ()))
        for j in range(3):
            e_ids, e_dist = oracle.bq_topk(rows, ids, qs[j], 100, *(flt or ()))
            assert d[j, :c[j]].tolist() == e_ids.tolist()
            assert s[j, :c[j]].tolist() == e_dist.tolist()
    with pytest.raises(ma.MsiError):
        st.upload([3, 2], np.zeros((2, 4), np.float32))


@pytest.mark.parametrize("n,dim,k,sample", [(5003, 130, 50, 1024), (9000, 64, 20, 2048), (20000, 768, 20, 1024),
                                            (4100, 8, 500, 1024)])
def test_one_sweep_form_vs_oracle(ctx, oracle, monkeypatch, n, dim, k, sample):
    """Large stores are answered by ONE sweep bounded by a sample's k-th distance (msi_bq.hip header); the sample size
    knob brings that path down to sizes the oracle checks in seconds.  40 queries: a full batch of 32 and one of 8."""
    monkeypatch.setenv("MSI_BQ_SAMPLE_ROWS", str(sample))
    rows = synth.make_embeddings(n, dim, seed=n + dim)
    ids = np.arange(n, dtype=np.uint32) * 3 + 7
    qs = synth.make_embeddings(40, dim, seed=1000 + n)
    st = ma.GpuBqStore(ctx, dim)
    st.upload(ids, rows)
    for nq in (40, 1, 3):
        d, s, c = st.search(qs[:nq], k)
        for j in range(nq):
            e_ids, e_dist = oracle.bq_topk(rows, ids, qs[j], k)
            assert int(c[j]) == e_ids.size and d[j, :c[j]].tolist() == e_ids.tolist(), (nq, j)
            assert s[j, :c[j]].view(np.uint32).tolist() == e_dist.view(np.uint32).tolist()


def test_one_sweep_form_falls_back_when_it_cannot_answer(ctx, oracle, monkeypatch):
    """Mass ties at the k-th distance (more than the select kernel orders), a filter that leaves fewer than k rows in
    the sample, and a filter that leaves fewer than k rows at all: the exhaustive form answers, same results."""
    monkeypatch.setenv("MSI_BQ_SAMPLE_ROWS", "1024")
    rng = np.random.default_rng(9)
    rows = rng.integers(-1, 2, (30000, 2)).astype(np.float32)      # 4 codes: > 4096 rows tie at every distance
    ids = np.arange(30000, dtype=np.uint32)
    st = ma.GpuBqStore(ctx, 2)
    st.upload(ids, rows)
    qs = np.array([[1, 1], [-1, 1], [0, 0]], dtype=np.float32)
    rare = np.nonzero(rng.random(30000) < 0.002)[0]                 # ~60 allowed rows, ~2 of them in the sample
    few = rare[:7]
    half = np.nonzero(rng.random(30000) < 0.5)[0]
    for allowed, k in ((None, 10), (None, 2000), (rare, 20), (few, 20), (half, 300)):
        flt = () if allowed is None else ma.dense_filter(allowed.tolist(), 30000)
        d, s, c = st.search(qs, k, *flt)
        for j in range(3):
            e_ids, e_dist = oracle.bq_topk(rows, ids, qs[j], k, *flt)
            assert int(c[j]) == e_ids.size and d[j, :c[j]].tolist() == e_ids.tolist(), (k, j)
            assert s[j, :c[j]].tolist() == e_dist.tolist()
