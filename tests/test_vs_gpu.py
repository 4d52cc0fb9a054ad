"""S1 parity: libmsi's vector k-NN (through the C ABI) against the CPU oracle.
Bar: identical docids in identical order, distances bit-identical (the device
rescoring uses the reference's scalar f32 arithmetic), goldens within 1e-5."""
import os

import numpy as np
import pytest

import meilisearch_amd as ma
from meilisearch_amd import synth

pytestmark = pytest.mark.gpu
f32 = np.float32


def check_against_oracle(oracle, store, rows, ids, queries, k, fb=None, nb=0):
    d, s, c = store.search(queries, k, fb, nb)
    for j in range(queries.shape[0]):
        e_ids, e_dist = oracle.vs_topk(rows, ids, queries[j], k, fb, nb)
        n = int(c[j])
        assert n == e_ids.size, (j, n, e_ids.size)
        assert d[j, :n].tolist() == e_ids.tolist(), (j, d[j, :n][:8], e_ids[:8])
        assert s[j, :n].view(np.uint32).tolist() == e_dist.view(np.uint32).tolist(), j


def test_reference_literals(ctx, oracle):
    # crates/meilisearch/tests/search/hybrid.rs:47-68,296-406,758 ; similar/mod.rs:19-43,281-335
    st = ma.GpuStore(ctx, 2)
    st.upload([1, 2, 3], [[1, 3], [1, 2], [2, 3]])
    d, s, c = st.search(np.array([[1, 1], [1, 0]], dtype=f32), 3)
    assert d[0].tolist() == [3, 2, 1] and d[1].tolist() == [3, 2, 1]
    sims = 1.0 - s.astype(np.float64)
    gold = [[0.990290343761444, 0.974341630935669, 0.9472135901451112],
            [0.7773500680923462, 0.7236068248748779, 0.6581138968467712]]
    assert np.abs(sims - np.array(gold)).max() <= 1e-5
    assert ((f32(1.0) - s) == np.array(gold, dtype=f32)).all()  # bit-exact as f32
    st3 = ma.GpuStore(ctx, 3)
    ids = [143, 166428, 287947, 299537, 522681]
    vecs = [[-0.5, 0.3, 0.85], [0.7, 0.7, -0.4], [0.8, 0.4, -0.5], [0.6, 0.8, -0.2], [0.1, 0.6, 0.8]]
    st3.upload(ids, vecs)
    v = st3.get_vector(143)
    assert v.tolist() == f32(vecs[0]).tolist()
    fb, nb = ma.dense_filter([i for i in ids if i != 143])   # Similar::execute removes the item
    d, s, c = st3.search(v[None, :], 4, fb, nb)
    assert d[0].tolist() == [522681, 299537, 166428, 287947]
    gold3 = f32([0.890957772731781, 0.39060014486312866, 0.2819308042526245, 0.1662663221359253])
    assert ((f32(1.0) - s[0]) == gold3).all()


def test_tie_order_ascending_docid(ctx, oracle):
    # crates/milli/src/search/new/tests/cutoff.rs:507-626
    st = ma.GpuStore(ctx, 2)
    rows = np.array([[0.1, 0.1], [-0.1, 0.1], [0.1, -0.1], [-0.1, -0.1]], dtype=f32)
    st.upload([0, 1, 2, 3], rows)
    d, s, c = st.search(np.array([[1, -1]], dtype=f32), 10)
    assert c[0] == 4 and d[0, :4].tolist() == [2, 0, 3, 1]
    assert np.allclose(1.0 - s[0, :4], [1.0, 0.5, 0.5, 0.0], atol=1e-6)


@pytest.mark.parametrize("n,dim,k", [(1, 7, 5), (15, 3, 4), (16, 16, 16), (1000, 96, 20), (5003, 130, 1),
                                      (20000, 384, 20), (4097, 64, 100), (3000, 32, 500)])
def test_random_vs_oracle(ctx, oracle, n, dim, k):
    rows = synth.make_embeddings(n, dim, seed=n + dim)
    ids = (np.arange(n, dtype=np.uint32) * 3 + 7)
    qs = synth.make_embeddings(5, dim, seed=1000 + n)
    st = ma.GpuStore(ctx, dim)
    st.upload(ids, rows)
    assert len(st) == n
    check_against_oracle(oracle, st, rows, ids, qs, k)


def test_large_k_and_k_above_store_size(ctx, oracle):
    rows = synth.make_embeddings(5000, 48, seed=71)
    ids = np.arange(5000, dtype=np.uint32)
    st = ma.GpuStore(ctx, 48)
    st.upload(ids, rows)
    qs = synth.make_embeddings(3, 48, seed=72)
    check_against_oracle(oracle, st, rows, ids, qs, 1000)          # K' = 1250
    small = ma.GpuStore(ctx, 48)
    small.upload(ids[:13], rows[:13])
    d, s, c = small.search(qs, 50)                                  # fewer rows than k
    assert c.tolist() == [13, 13, 13]
    check_against_oracle(oracle, small, rows[:13], ids[:13], qs, 50)
    big = synth.make_embeddings(90000, 48, seed=73)                 # sparse path with a large k
    st2 = ma.GpuStore(ctx, 48)
    st2.upload(np.arange(90000, dtype=np.uint32), big)
    check_against_oracle(oracle, st2, big, np.arange(90000, dtype=np.uint32), qs[:2], 700)


def test_more_than_one_query_tile(ctx, oracle):
    rows = synth.make_embeddings(6000, 128, seed=21)
    ids = np.arange(6000, dtype=np.uint32)
    qs = synth.make_embeddings(37, 128, seed=22)
    st = ma.GpuStore(ctx, 128)
    st.upload(ids, rows)
    check_against_oracle(oracle, st, rows, ids, qs, 20)


@pytest.mark.parametrize("selectivity", [0.5, 0.05, 0.002])
def test_filtered_search(ctx, oracle, selectivity):
    n, dim = 30000, 64
    rows = synth.make_embeddings(n, dim, seed=31)
    ids = np.sort(np.random.default_rng(5).choice(100000, n, replace=False)).astype(np.uint32)
    rng = np.random.default_rng(int(selectivity * 1e6))
    allowed = ids[rng.random(n) < selectivity]
    # also docids that are not in the store, and ids beyond nbits are not allowed
    fb, nb = ma.dense_filter(list(allowed) + [1, 5], nbits=90000)
    qs = synth.make_embeddings(4, dim, seed=32)
    st = ma.GpuStore(ctx, dim)
    st.upload(ids, rows)
    check_against_oracle(oracle, st, rows, ids, qs, 20, fb, nb)
    # empty filter
    fb0, nb0 = ma.dense_filter([], nbits=64)
    d, s, c = st.search(qs, 20, fb0, nb0)
    assert c.tolist() == [0, 0, 0, 0]


@pytest.mark.parametrize("storage", ["f32", "bf16"])
@pytest.mark.parametrize("gather_pct", ["0", "250", "1600"])
def test_filtered_sweeps_are_row_granular(ctx, oracle, storage, gather_pct, monkeypatch):
    """Round 6: a filtered sweep visits ITEMS of 16 rows — the allowed rows of a region compacted (gathered at
    64-byte-sector granularity from the row-sector tiles) or its tiles that hold an allowed row, whichever is cheaper
    (vs_filter_rows_kernel; MSI_VS_GATHER_PCT: 0 = always compacted, 1600 and above = always tiles as in rounds 1-5).  hannoy's linear
    mode scores only candidates (vector/store.rs:1079-1080).  Every density x mode against the oracle — the sample pass and
    the sparse pass (store above 32 768 rows), the int8 level of the f32 store, the bf16 store's f32 levels, a filter
    denser in one half of the store than in the other (regions decide for themselves), k above the allowed rows — and the
    device's own count of what it visited."""
    monkeypatch.setenv("MSI_VS_GATHER_PCT", gather_pct)
    emulated = bool(os.environ.get("MSI_RUNNER_SO"))      # the CPU tier (tests/emu): the same paths on a store just above the
    n, dim = (36000, 64) if emulated else (70000, 128)   # size where the sample + sparse passes begin (32 768 rows)
    rows = synth.make_embeddings(n, dim, seed=131)
    if storage == "bf16":
        rows = synth.round_to_bf16(rows)
    ids = (np.arange(n, dtype=np.uint32) * 3 + 7).astype(np.uint32)
    st = ma.GpuStore(ctx, dim, storage=storage)
    st.upload(ids, rows)
    qs = synth.make_embeddings(5, dim, seed=132)
    rng = np.random.default_rng(133)
    for density in ((0.5, 0.1, 0.01) if emulated else (0.5, 0.1, 0.01, 0.001)):
        keep = rng.random(n) < density
        if density == 0.1:
            keep[: n // 2] = rng.random(n // 2) < 0.9     # a dense half and a sparse half
        allowed = ids[keep]
        fb, nb = ma.dense_filter(list(allowed), nbits=int(ids[-1]) + 1)
        check_against_oracle(oracle, st, rows, ids, qs, 20, fb, nb)
        fs = st.filter_stats()
        assert fs["allowed_rows"] == allowed.size, (density, fs)
        assert fs["items"] == fs["compact_items"] + fs["tile_items"]
        tiles_with_allowed = int(np.count_nonzero(np.add.reduceat(keep.astype(np.int64), np.arange(0, n, 16))))
        if gather_pct == "1600":
            assert fs["compact_items"] == 0 and fs["items"] == tiles_with_allowed, (density, fs)
        elif gather_pct == "0":
            # every region compacts: at most one padded item per region of 8 192 rows above the allowed rows' own items
            assert fs["tile_items"] == 0 and fs["items"] <= (allowed.size + 15) // 16 + (n + 8191) // 8192, (density, fs)
        else:
            assert fs["items"] <= tiles_with_allowed, (density, fs)
            if density <= 0.01:
                assert fs["tile_items"] == 0 and fs["items"] * 16 <= 2 * allowed.size + 16 * ((n + 8191) // 8192), (density, fs)
    # more neighbours asked for than rows allowed
    few = ids[rng.choice(n, 11, replace=False)]
    fb, nb = ma.dense_filter(list(few), nbits=int(ids[-1]) + 1)
    check_against_oracle(oracle, st, rows, ids, qs, 20, fb, nb)


def test_duplicates_zero_rows_and_zero_query(ctx, oracle):
    dim = 48
    base = synth.make_embeddings(50, dim, seed=41)
    rows = np.concatenate([np.repeat(base[:1], 1500, axis=0),     # 1 500 identical rows: ties > K' (128; 1 024 on the int8 level)
                           base, np.zeros((5, dim), dtype=f32),   # zero vectors: distance 0 by definition
                           (base[:20] * f32(1e-30))])              # tiny norms: pn*qn <= EPS
    n = rows.shape[0]
    ids = np.arange(n, dtype=np.uint32) + 10
    st = ma.GpuStore(ctx, dim)
    st.upload(ids, rows)
    qs = np.concatenate([base[:1] * f32(2.0), synth.make_embeddings(2, dim, seed=42),
                         np.zeros((1, dim), dtype=f32)])
    before = st.stats()
    check_against_oracle(oracle, st, rows, ids, qs, 20)
    after = st.stats()
    # 1 500 ties are more than the usual K' holds (also the int8 level's 1 024): since round 4 the query is re-run with K' = 2048
    # rescored candidates (msi_vs.hip, levels of effort) and proven there — no exhaustive pass
    assert after["exhaustive_reruns"] == before["exhaustive_reruns"]
    if after["i8_bytes_per_tile"]:
        assert after["i8_sweeps"] - before["i8_sweeps"] >= 2       # the int8 sweep, then the int8 sweep with K' = 2048
        assert after["level_sweeps"] == before["level_sweeps"]     # ... which settled it: no sweep of the f32 rows
    else:
        assert after["level_sweeps"][1] > before["level_sweeps"][1]
    # ... and more ties than ANY K' holds still end in the exhaustive pass (reference arithmetic for every row)
    rows2 = np.concatenate([np.repeat(base[:1], 2100, axis=0), base])
    ids2 = np.arange(rows2.shape[0], dtype=np.uint32) + 10
    st2 = ma.GpuStore(ctx, dim)
    st2.upload(ids2, rows2)
    b2 = st2.stats()["exhaustive_reruns"]
    check_against_oracle(oracle, st2, rows2, ids2, qs[:2], 20)
    assert st2.stats()["exhaustive_reruns"] > b2


def test_large_scan_and_sample_pass(ctx, oracle):
    # large enough for the strided sample pass (thresholds) to run
    n, dim = 300000, 128
    rows = synth.make_embeddings(n, dim, seed=51)
    ids = np.arange(n, dtype=np.uint32)
    qs = synth.make_embeddings(16, dim, seed=52)
    st = ma.GpuStore(ctx, dim)
    st.upload(ids, rows)
    s0 = st.stats()
    check_against_oracle(oracle, st, rows, ids, qs[:3], 20)
    s1 = st.stats()
    assert s1["scan_launches"] - s0["scan_launches"] == 2      # sample + main
    assert s1["exhaustive_reruns"] == s0["exhaustive_reruns"]


def test_three_query_tiles_sparse_and_filtered(ctx, oracle):
    # 40 queries = three 16-query MFMA tiles in ONE sweep (dim 128 leaves room for 48),
    # on a store large enough for the sample + sparse passes; then the same with a filter
    n, dim = 120000, 128
    rows = synth.make_embeddings(n, dim, seed=81)
    ids = np.arange(n, dtype=np.uint32) * 2 + 1
    qs = synth.make_embeddings(40, dim, seed=82)
    st = ma.GpuStore(ctx, dim)
    st.upload(ids, rows)
    # 128 queries per sweep of the int8 copy (level 0); the f32 sweeps: 96 (bf16x2, the default: the queries' hi halves only in
    # LDS), 48 with MSI_VS_SCAN_MATH=bf16x3
    f32_batch = 48 if os.environ.get("MSI_VS_SCAN_MATH") in ("bf16x3", "f32") else 96
    assert st.stats()["f32_queries_per_sweep"] == f32_batch
    assert st.max_batch == (128 if st.stats()["i8_bytes_per_tile"] else f32_batch)
    s0 = st.stats()
    check_against_oracle(oracle, st, rows, ids, qs, 20)
    s1 = st.stats()
    assert s1["scan_launches"] - s0["scan_launches"] == 2      # one sample + one full sweep for all 40
    assert s1["exhaustive_reruns"] == s0["exhaustive_reruns"]
    rng = np.random.default_rng(83)
    for sel in (0.3, 0.01, 0.0005):
        allowed = ids[rng.random(n) < sel]
        fb, nb = ma.dense_filter(allowed, nbits=int(ids.max()) + 1)
        check_against_oracle(oracle, st, rows, ids, qs[:19], 20, fb, nb)
    # a single query (15 idle columns in the MFMA tile) and an odd batch
    check_against_oracle(oracle, st, rows, ids, qs[:1], 20)
    check_against_oracle(oracle, st, rows, ids, qs[:17], 5)


@pytest.mark.parametrize("dim,scale", [(768, 1.0), (384, 1.0), (100, 1e3), (1024, 1e-3)])
def test_fast_scan_error_is_inside_the_proof_bound(ctx, dim, scale):
    # the exactness proof is only sound if |fast cos - reference cos| <= eps for EVERY row;
    # check it against an f64 reference, including heavy-tailed rows and a skewed query
    n = 3000
    rng = np.random.default_rng(dim)
    rows = (rng.standard_normal((n, dim)) * scale).astype(f32)
    rows[:200] *= rng.lognormal(0, 3, size=(200, dim)).astype(f32)       # wild dynamic range
    rows[200:300] = np.abs(rows[200:300])                                   # no cancellation
    qs = rng.standard_normal((20, dim)).astype(f32)
    qs[1] = np.abs(qs[1])
    qs[2] *= rng.lognormal(0, 3, size=dim).astype(f32)
    st = ma.GpuStore(ctx, dim)
    st.upload(np.arange(n, dtype=np.uint32), rows)
    fast, eps = st.debug_fast_scores(qs)
    r64, q64 = rows.astype(np.float64), qs.astype(np.float64)
    ref = (q64 @ r64.T) / (np.linalg.norm(q64, axis=1)[:, None] * np.linalg.norm(r64, axis=1)[None, :])
    got = fast.astype(np.float64) / np.linalg.norm(q64, axis=1)[:, None]
    err = np.abs(got - ref).max()
    assert err <= eps, (err, eps)
    # bf16x3 / f32: second-order bounds with room to spare.  bf16x2 rounds the query to bf16: its bound is first order
    # (2^-8 |x||q|) and a row dominated by one coordinate realises half of it
    comfortable = 0.25 if os.environ.get("MSI_VS_SCAN_MATH") in ("bf16x3", "f32") else 0.6
    assert err <= comfortable * eps, ("the bound should be comfortable", err, eps)


@pytest.mark.parametrize("dim,scale", [(768, 1.0), (384, 1.0), (100, 1e3), (1024, 1e-3), (64, 1.0), (130, 1.0), (256, 1.0), (200, 1.0)])
def test_int8_sweep_error_is_inside_its_proof_bound(ctx, dim, scale, monkeypatch):
    """Round 5: level 0 sweeps an int8 copy of the rows (msi_vs.hip: every row divided by its norm, quantised with its own
    scale).  Its proof is only sound if |fast cos - reference cos| <= eps of the query for EVERY row, with eps = the store's
    largest row residual + the query's own residual + their product (Cauchy-Schwarz): checked against an f64 reference on
    heavy-tailed rows (one dominant coordinate: the coarsest quantisation a row can get), rows without cancellation, a skewed
    query — and the bound is not loose either (the worst row comes within a factor of a few of it)."""
    monkeypatch.setenv("MSI_VS_DEBUG_I8", "1")
    n = 3000
    rng = np.random.default_rng(dim + 5)
    rows = (rng.standard_normal((n, dim)) * scale).astype(f32)
    rows[:200] *= rng.lognormal(0, 3, size=(200, dim)).astype(f32)
    rows[200:300] = np.abs(rows[200:300])
    rows[300:310] = 0
    rows[300:310, 3] = 1.0                                                  # one-hot rows: quantised exactly
    qs = rng.standard_normal((20, dim)).astype(f32)
    qs[1] = np.abs(qs[1])
    qs[2] *= rng.lognormal(0, 3, size=dim).astype(f32)
    qs[3] = rows[5]
    st = ma.GpuStore(ctx, dim)
    st.upload(np.arange(n, dtype=np.uint32), rows)
    assert st.stats()["i8_bytes_per_tile"] == ((dim + 127) // 128) * 128 * 16 + 128
    fast, eps = st.debug_fast_scores(qs)
    r64, q64 = rows.astype(np.float64), qs.astype(np.float64)
    ref = (q64 @ r64.T) / (np.linalg.norm(q64, axis=1)[:, None] * np.linalg.norm(r64, axis=1)[None, :])
    got = fast.astype(np.float64) / np.linalg.norm(q64, axis=1)[:, None]
    err = np.abs(got - ref).max()
    assert 0.0 < eps < 0.2, eps
    assert err <= eps, (err, eps)
    assert err >= eps / 40.0, ("a bound this far above the worst row would waste candidates", err, eps)


def test_int8_level_proves_iid_rows_and_hands_crowds_to_the_f32_levels(ctx, oracle):
    """The levels of effort with the int8 copy in front: on i.i.d. rows level 0 proves every query (no f32 sweep at all); a crowd
    of rows closer to each other than the int8 bound is handed to the f32 levels and settled there without the exhaustive pass;
    a store created with MSI_VS_I8=0 answers the same lists (the copy never changes an answer)."""
    n, dim = 50000, 96
    rows = synth.make_embeddings(n, dim, seed=301)
    ids = np.arange(n, dtype=np.uint32) * 2 + 5
    qs = synth.make_embeddings(40, dim, seed=302)
    st = ma.GpuStore(ctx, dim)
    st.upload(ids, rows)
    s0 = st.stats()
    if not s0["i8_bytes_per_tile"]:
        pytest.skip("the store has no int8 copy (MSI_VS_I8=0)")
    check_against_oracle(oracle, st, rows, ids, qs, 20)
    s1 = st.stats()
    assert s1["i8_sweeps"] - s0["i8_sweeps"] == 1                  # 40 queries: one sweep of the copy (after one sample sweep)
    assert s1["scan_launches"] - s0["scan_launches"] == 2
    assert s1["level_sweeps"] == s0["level_sweeps"] and s1["exhaustive_reruns"] == s0["exhaustive_reruns"]
    # a crowd: 2 500 rows within ~1e-3 of each other in cosine around the query direction (inside the int8 bound AND more than
    # any K' of the int8 levels; the f32 levels tell them apart)
    rng = np.random.default_rng(303)
    v = rng.standard_normal(dim).astype(f32)
    rows2 = rows.copy()
    rows2[:2500] = v[None, :] + 0.05 * rng.standard_normal((2500, dim)).astype(f32) * (np.linalg.norm(v) / np.sqrt(dim))
    st.upload(ids, rows2)
    s2 = st.stats()
    check_against_oracle(oracle, st, rows2, ids, v[None, :].astype(f32), 20)
    s3 = st.stats()
    assert s3["i8_sweeps"] > s2["i8_sweeps"] and sum(s3["level_sweeps"]) > sum(s2["level_sweeps"])
    assert s3["exhaustive_reruns"] == s2["exhaustive_reruns"]


@pytest.mark.parametrize("n,dim,k", [(37, 5, 10), (3000, 96, 20), (70000, 300, 20), (40000, 1024, 50)])
def test_bf16_store_vs_oracle_on_rounded_rows(ctx, oracle, n, dim, k):
    # BASELINE.json config 5 stores bf16: parity is against the f32 oracle on the ROUNDED rows
    rows = synth.make_embeddings(n, dim, seed=n + dim + 1)
    rb = synth.round_to_bf16(rows)
    ids = np.arange(n, dtype=np.uint32) * 2
    qs = synth.make_embeddings(19, dim, seed=n + 2)
    st = ma.GpuStore(ctx, dim, storage="bf16")
    st.upload(ids, rows)                       # rounded on the device at upload
    assert (st.get_vector(int(ids[n // 2])) == rb[n // 2]).all()
    check_against_oracle(oracle, st, rb, ids, qs, k)
    allowed = ids[np.random.default_rng(3).random(n) < 0.2]
    fb, nb = ma.dense_filter(allowed, nbits=int(ids.max()) + 1)
    check_against_oracle(oracle, st, rb, ids, qs[:5], k, fb, nb)
    # the proof bound holds for the bf16 path as well
    fast, eps = st.debug_fast_scores(qs[:4])
    r64, q64 = rb.astype(np.float64), qs[:4].astype(np.float64)
    ref = (q64 @ r64.T) / (np.linalg.norm(q64, axis=1)[:, None] * np.maximum(np.linalg.norm(r64, axis=1), 1e-300)[None, :])
    got = fast.astype(np.float64) / np.linalg.norm(q64, axis=1)[:, None]
    assert np.abs(got - ref).max() <= 0.25 * eps


def test_row_sharded_search_with_device_merge(ctx, oracle):
    # three row shards on one GPU stand in for three GPUs: per-shard device search, then the
    # device k-way merge of the "all-gathered" lists must equal the unsharded oracle
    import torch
    from meilisearch_amd.distributed import merge_topk_device, row_range
    n, dim, k, B, W = 9001, 40, 15, 21, 3
    rows = synth.make_embeddings(n, dim, seed=111)
    rows[4000:4050] = rows[100]                       # ties that straddle shards
    ids = np.arange(n, dtype=np.uint32) * 2 + 5
    qs = synth.make_embeddings(B, dim, seed=112)
    dev = torch.device("cuda", ctx.device)
    g_ids = torch.zeros((W, B, k), dtype=torch.int32, device=dev)
    g_dist = torch.zeros((W, B, k), dtype=torch.float32, device=dev)
    g_cnt = torch.zeros((W, B), dtype=torch.int32, device=dev)
    q_t = torch.from_numpy(qs).to(dev)
    stores = []
    for r in range(W):
        r0, r1 = row_range(n, r, W)
        st = ma.GpuStore(ctx, dim)
        st.upload(ids[r0:r1], rows[r0:r1])
        st.search_device(q_t, k, g_ids[r], g_dist[r], g_cnt[r])
        stores.append(st)
    m_ids = torch.zeros((B, k), dtype=torch.int32, device=dev)
    m_dist = torch.zeros((B, k), dtype=torch.float32, device=dev)
    m_cnt = torch.zeros(B, dtype=torch.int32, device=dev)
    merge_topk_device(ctx, g_ids, g_dist, g_cnt, m_ids, m_dist, m_cnt)
    ctx.synchronize()
    got_ids = m_ids.cpu().numpy().view(np.uint32)
    got_dist = m_dist.cpu().numpy()
    for j in range(B):
        e_ids, e_dist = oracle.vs_topk(rows, ids, qs[j], k)
        assert int(m_cnt[j]) == e_ids.size and got_ids[j].tolist() == e_ids.tolist()
        assert got_dist[j].view(np.uint32).tolist() == e_dist.view(np.uint32).tolist()


def test_microbatcher_fuses_concurrent_callers(ctx, oracle):
    # 40 threads, one query each (the shape of milli's spawn_blocking searches): every caller
    # must get exactly its own answer, and the sweeps must be shared
    import threading
    n, dim = 60000, 64
    rows = synth.make_embeddings(n, dim, seed=91)
    ids = np.arange(n, dtype=np.uint32)
    qs = synth.make_embeddings(40, dim, seed=92)
    ks = [20 if j % 3 else 7 for j in range(40)]        # mixed k inside one fused sweep
    st = ma.GpuStore(ctx, dim)
    st.upload(ids, rows)
    st.set_microbatch(20000)
    out = [None] * 40
    barrier = threading.Barrier(40)

    def worker(j):
        barrier.wait()
        out[j] = st.search(qs[j:j + 1], ks[j])
    th = [threading.Thread(target=worker, args=(j,)) for j in range(40)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    for j in range(40):
        d, s, c = out[j]
        e_ids, e_dist = oracle.vs_topk(rows, ids, qs[j], ks[j])
        assert c[0] == e_ids.size and d[0, :c[0]].tolist() == e_ids.tolist()
        assert s[0, :c[0]].view(np.uint32).tolist() == e_dist.view(np.uint32).tolist()
    stats = st.microbatch_stats()
    assert stats["fused_calls"] == 40
    assert stats["fused_sweeps"] < 40, stats             # sweeps were shared
    # filtered / multi-sweep calls bypass the batcher and still work with it enabled
    fb, nb = ma.dense_filter(ids[::5])
    check_against_oracle(oracle, st, rows, ids, qs[:3], 10, fb, nb)
    st.set_microbatch(0)
    check_against_oracle(oracle, st, rows, ids, qs[:2], 10)


def test_objects_may_outlive_their_context():
    c2 = ma.Context(0)
    st = ma.GpuStore(c2, 8)
    st.upload([1, 2, 3], np.ones((3, 8), dtype=f32))
    d, s, c = st.search(np.ones((1, 8), dtype=f32), 2)
    c2.close()          # drops the caller's reference only
    st.close()          # the store's reference frees the context


def test_round_trip_and_idempotence(ctx):
    # size-independent properties: every stored row's nearest neighbour is itself
    # (distance ~0) and a repeated search returns identical bits
    n, dim = 50000, 96
    rows = synth.make_embeddings(n, dim, seed=61)
    st = ma.GpuStore(ctx, dim)
    st.upload(np.arange(n, dtype=np.uint32), rows)
    pick = np.array([0, 1, 777, 4999, n - 1])
    d, s, c = st.search(rows[pick], 3)
    assert d[:, 0].tolist() == pick.tolist()
    assert np.abs(s[:, 0]).max() <= 1e-6
    assert (np.diff(s.astype(np.float64), axis=1) >= 0).all()     # sorted ascending
    d2, s2, c2 = st.search(rows[pick], 3)
    assert (d == d2).all() and (s.view(np.uint32) == s2.view(np.uint32)).all()
    for p in pick:
        assert (st.get_vector(int(p)) == rows[p]).all()
    assert st.get_vector(n + 5) is None


def test_multi_store_vector_store_mirror(ctx, oracle):
    # documents with several embeddings: store i holds each doc's i-th vector
    # (store.rs:752-786); results are concatenated and sorted (store.rs:1036-1062)
    dim = 24
    rng = np.random.default_rng(71)
    emb = {int(d): [rng.standard_normal(dim).astype(f32) for _ in range(1 + (d % 3))] for d in range(0, 400, 2)}
    vs = ma.VectorStore(ctx, dim)
    vs.add_documents(emb)
    q = rng.standard_normal(dim).astype(f32)
    got = vs.nns_by_vector(q, 10)
    exp = []
    for sid in range(3):
        docs = [d for d in emb if len(emb[d]) > sid]
        rows = np.stack([emb[d][sid] for d in docs])
        e_ids, e_dist = oracle.vs_topk(rows, np.array(docs, dtype=np.uint32), q, 10)
        exp += list(zip(e_ids.tolist(), e_dist.tolist()))
    exp.sort(key=lambda t: (t[1], t[0]))
    assert [g[0] for g in got] == [e[0] for e in exp]
    assert [f32(g[1]) for g in got] == [f32(e[1]) for e in exp]
    assert len(vs.item_vectors(4)) == 1 + (4 % 3)
    sim = vs.nns_by_item(4, 5, filter_docids=[d for d in emb if d != 4])
    assert all(d != 4 for d, _ in sim) and len(sim) == 5 * (1 + (4 % 3))


def test_errors(ctx):
    st = ma.GpuStore(ctx, 8)
    with pytest.raises(ma.MsiError) as e:
        st.upload([3, 2, 1], np.zeros((3, 8), dtype=f32))
    assert "MSI_E_NOT_SORTED" in str(e.value)
    st.upload([1, 2, 3], np.ones((3, 8), dtype=f32))
    with pytest.raises(ma.MsiError) as e:
        st.search(np.ones((1, 8), dtype=f32), 5000)
    assert "MSI_E_UNSUPPORTED" in str(e.value)
    cancel = np.array([1], dtype=np.int32)
    with pytest.raises(ma.MsiError) as e:
        st.search(np.ones((1, 8), dtype=f32), 2, cancel=cancel)
    assert "MSI_E_CANCELLED" in str(e.value)
    d, s, c = st.search(np.ones((2, 8), dtype=f32), 0)
    assert c.tolist() == [0, 0]


def test_six_query_tiles_in_one_sweep_and_the_bf16x3_second_opinion(ctx, oracle, monkeypatch):
    """(The f32 levels on their own: the store is created without the int8 copy.)  The default contraction (bf16x2: the queries' hi halves only in LDS) takes 96 queries per sweep; its proof margin is
    2^-8 wide, so a query with a crowd of near-equal scores at the top is re-run through the bf16x3 contraction before
    anything is answered exhaustively."""
    if os.environ.get("MSI_VS_SCAN_MATH") in ("bf16x3", "f32"):
        pytest.skip("the 96-query sweep is the bf16x2 contraction's")
    monkeypatch.setenv("MSI_VS_I8", "0")
    n, dim = 60000, 128
    rows = synth.make_embeddings(n, dim, seed=91)
    ids = np.arange(n, dtype=np.uint32) * 3
    qs = synth.make_embeddings(90, dim, seed=92)
    st = ma.GpuStore(ctx, dim)
    st.upload(ids, rows)
    assert st.max_batch == 96
    s0 = st.stats()
    check_against_oracle(oracle, st, rows, ids, qs, 20)          # 90 queries: six tiles, one sample + one main sweep
    s1 = st.stats()
    assert s1["scan_launches"] - s0["scan_launches"] == 2
    # a crowd: 300 rows within ~4e-3 of each other in cosine around the query direction
    rng = np.random.default_rng(93)
    v = rng.standard_normal(dim).astype(f32)
    crowd = v[None, :] + 0.9 * rng.standard_normal((300, dim)).astype(f32) * (np.linalg.norm(v) / np.sqrt(dim)) * 0.1
    rows2 = rows.copy()
    rows2[:300] = crowd
    st.upload(ids, rows2)
    q = v[None, :].astype(f32)
    s2 = st.stats()
    check_against_oracle(oracle, st, rows2, ids, q, 20)
    s3 = st.stats()
    assert s3["scan_launches"] - s2["scan_launches"] >= 3         # the bf16x3 re-run swept again
    assert s3["exhaustive_reruns"] == s2["exhaustive_reruns"]    # ... and settled it


@pytest.mark.parametrize("dim,batch2,batch3", [(1024, 80, 32), (1536, 48, 16)])
def test_second_opinion_capacity_at_wide_dims(ctx, oracle, dim, batch2, batch3, monkeypatch):
    """(The f32 levels on their own: the store is created without the int8 copy.)  ADVICE r2 (high): the bf16x3 second opinion keeps both halves of the queries in LDS, so at d > 768 it takes fewer
    query tiles per sweep than the bf16x2 main pass (d = 1024: 2 tiles of 64 KiB; d = 1536: 1 tile of 96 KiB) — with
    more flagged queries in one step than that, the re-run must go sub-batch by sub-batch instead of asking for 192 KiB
    of LDS.  Every query of the step points into a crowd of near-equal scores, so every one of them is flagged."""
    if os.environ.get("MSI_VS_SCAN_MATH") in ("bf16x3", "f32"):
        pytest.skip("the second opinion is the bf16x2 contraction's")
    monkeypatch.setenv("MSI_VS_I8", "0")
    n = 6000
    rng = np.random.default_rng(dim)
    rows = synth.make_embeddings(n, dim, seed=95)
    v = rng.standard_normal(dim).astype(f32)
    # (round 4: 2 600 rows whose distance to the query direction spreads over 2e-4 .. 7e-3 — more than K' = 2048 of them inside
    # the bf16x2 proof's margin, so also the second level of effort (bf16x2, K' = 2048) fails and the bf16x3 level is reached;
    # a few hundred inside the bf16x3 margin, so that level proves every query)
    scale = np.exp(rng.uniform(np.log(0.02), np.log(0.12), 2600)).astype(f32)
    rows[:2600] = v[None, :] + scale[:, None] * rng.standard_normal((2600, dim)).astype(f32) * (np.linalg.norm(v) / np.sqrt(dim))
    ids = np.arange(n, dtype=np.uint32) * 2 + 1
    st = ma.GpuStore(ctx, dim)
    st.upload(ids, rows)
    assert st.max_batch == batch2
    nq = batch2                       # one full step, more flagged queries than one second-opinion sweep holds
    assert nq > batch3
    qs = (v[None, :] * (1.0 + 0.01 * np.arange(nq, dtype=f32))[:, None]).astype(f32)     # same direction: same crowd
    s0 = st.stats()
    check_against_oracle(oracle, st, rows, ids, qs, 20)
    s1 = st.stats()
    # one dense pass (<= 32 Ki rows), one with K' = 2048, then ceil(flagged / batch3) bf16x3 passes: more than one sub-batch
    assert s1["scan_launches"] - s0["scan_launches"] >= 1 + 1 + 2
    assert s1["level_sweeps"][2] - s0["level_sweeps"][2] >= 2
    assert s1["exhaustive_reruns"] == s0["exhaustive_reruns"]


class _DeviceArrays:
    """Arrays the device entry points can be handed: torch tensors on the context's device — or, on the CPU tier's emulated
    kernels (tests/emu: device memory is host memory, no torch device), plain numpy arrays."""

    def __init__(self, ctx):
        self.emulated = bool(os.environ.get("MSI_RUNNER_SO"))
        self.ctx = ctx
        self.keep = []

    def put(self, a):
        import ctypes as C
        if self.emulated:
            a = np.ascontiguousarray(a).copy()
            self.keep.append(a)
            return a, C.c_void_p(a.ctypes.data)
        import torch
        t = torch.from_numpy(np.ascontiguousarray(a)).to(torch.device("cuda", self.ctx.device))
        torch.cuda.synchronize()
        self.keep.append(t)
        return t, C.c_void_p(t.data_ptr())

    def get(self, h):
        return h if self.emulated else h.cpu().numpy()


@pytest.mark.parametrize("filtered", [False, True])
def test_device_entry_point_pipelines_its_chunks(ctx, oracle, filtered, monkeypatch):
    """msi_vs_search_device with more queries than one sweep admits: the chunks' preparation / selection / rescoring run on
    the store's second stream against two scratch sets while the neighbouring chunks' sweeps run on the context's stream
    (msi_vs.hip: search_device_pipelined).  5 chunks (the scratch sets alternate 0 1 0 1 0), a ragged last chunk, with and
    without a filter; every list against the oracle, and identical to what the host entry point (one stream, one scratch
    set) answers.  Twice: the second call reuses the sets the first one left behind."""
    import ctypes as C
    from meilisearch_amd._lib import check, lib
    monkeypatch.setenv("MSI_VS_PIPELINE", "1")     # (off by default: it measured no gain — profiles/r4_vs_pipeline.txt)
    # the pipeline reports unproven queries instead of re-running them: it is taken under the old contract only (ADVICE r5;
    # include/msi.h at the prototype) — under the default contract the knob is ignored and the call answers every query itself
    monkeypatch.setenv("MSI_VS_DEVICE_RERUN", "0")
    n, dim, k = 20000, 64, 10
    rows = synth.make_embeddings(n, dim, seed=77)
    rows[7000:7040] = rows[13]                                  # ties
    ids = (np.arange(n, dtype=np.uint32) * 3 + 1)
    st = ma.GpuStore(ctx, dim)
    st.upload(ids, rows)
    step = lib().msi_vs_max_batch(st._h)
    nq = 4 * step + 7
    qs = synth.make_embeddings(nq, dim, seed=78)
    fb = nb = None
    if filtered:
        allowed = ids[np.random.default_rng(9).random(n) < 0.3]
        fb, nb = ma.dense_filter(list(allowed), nbits=int(ids[-1]) + 1)
    dv = _DeviceArrays(ctx)
    q_h, q_p = dv.put(qs)
    f_p = None
    if filtered:
        _, f_p = dv.put(np.ascontiguousarray(fb, dtype=np.uint64))
    h_ids, h_dist, h_cnt = st.search(qs, k, fb, nb or 0)
    for _ in range(2):
        o_ids, o_ids_p = dv.put(np.zeros((nq, k), np.uint32))
        o_dist, o_dist_p = dv.put(np.zeros((nq, k), np.float32))
        o_cnt, o_cnt_p = dv.put(np.zeros(nq, np.uint32))
        o_inx, o_inx_p = dv.put(np.zeros(nq, np.uint32))
        check(lib().msi_vs_search_device(st._h, q_p, nq, k, f_p, nb or 0, o_ids_p, o_dist_p, o_cnt_p, o_inx_p))
        ctx.synchronize()
        g_ids, g_dist, g_cnt, g_inx = (dv.get(x) for x in (o_ids, o_dist, o_cnt, o_inx))
        assert not g_inx.any()
        assert (np.asarray(g_cnt).astype(np.uint32) == h_cnt).all()
        for j in range(nq):
            c = int(h_cnt[j])
            assert np.asarray(g_ids[j][:c]).view(np.uint32).tolist() == h_ids[j][:c].tolist(), j
            assert np.asarray(g_dist[j][:c]).view(np.uint32).tolist() == h_dist[j][:c].view(np.uint32).tolist(), j
    for j in list(range(0, nq, 37)) + [nq - 1]:
        e_ids, e_dist = oracle.vs_topk(rows, ids, qs[j], k, fb, nb or 0)
        assert int(h_cnt[j]) == e_ids.size and h_ids[j][:e_ids.size].tolist() == e_ids.tolist(), j
        assert h_dist[j][:e_ids.size].view(np.uint32).tolist() == e_dist.view(np.uint32).tolist(), j


@pytest.mark.parametrize("first_level", [None, "f32"])
def test_device_entry_point_answers_its_unproven_queries_itself(ctx, oracle, monkeypatch, first_level):
    """msi_vs_search_device used to REPORT the queries its sweep could not prove (d_inexact) and leave the re-run
    to its caller; store.rs:638-675 always answers, and so does the entry point now — the queries a pass cannot prove are gathered
    on the device and re-run level by level, then exhaustively, inside the call.  Three kinds of query in one call of more than
    one sweep: plain ones (proven at the first level), ones that point into a crowd of 2 500 near-equal rows (the f32 levels settle
    them) and ones that point into 2 100 identical rows (more ties than any K': the exhaustive pass).  Every list against the
    oracle; d_inexact comes back all zero.  Once from the store's first level (the int8 sweep when it has the copy), once with
    the search starting at the f32 level (MSI_VS_FIRST_LEVEL=f32)."""
    import ctypes as C
    from meilisearch_amd._lib import check, lib
    if first_level:
        monkeypatch.setenv("MSI_VS_FIRST_LEVEL", first_level)
    n, dim, k = 30000, 64, 10
    rng = np.random.default_rng(401)
    rows = synth.make_embeddings(n, dim, seed=402)
    v = rng.standard_normal(dim).astype(f32)
    rows[:2500] = v[None, :] + 0.05 * rng.standard_normal((2500, dim)).astype(f32) * (np.linalg.norm(v) / np.sqrt(dim))
    w = rng.standard_normal(dim).astype(f32)
    rows[3000:5100] = w[None, :]
    ids = np.arange(n, dtype=np.uint32) * 2 + 3
    st = ma.GpuStore(ctx, dim)
    st.upload(ids, rows)
    step = lib().msi_vs_max_batch(st._h)
    nq = step + 21
    qs = synth.make_embeddings(nq, dim, seed=403)
    crowd = [3, step - 1, step + 5]
    ties = [0, 40, step + 20]
    for j in crowd:
        qs[j] = v * f32(1.0 + 0.01 * j)
    for j in ties:
        qs[j] = w * f32(2.0 + 0.5 * j)
    dv = _DeviceArrays(ctx)
    _, q_p = dv.put(qs)
    s0 = st.stats()
    o_ids, o_ids_p = dv.put(np.zeros((nq, k), np.uint32))
    o_dist, o_dist_p = dv.put(np.zeros((nq, k), np.float32))
    o_cnt, o_cnt_p = dv.put(np.zeros(nq, np.uint32))
    o_inx, o_inx_p = dv.put(np.ones(nq, np.uint32))
    check(lib().msi_vs_search_device(st._h, q_p, nq, k, None, 0, o_ids_p, o_dist_p, o_cnt_p, o_inx_p))
    ctx.synchronize()
    g_ids, g_dist, g_cnt, g_inx = (np.asarray(dv.get(x)) for x in (o_ids, o_dist, o_cnt, o_inx))
    s1 = st.stats()
    assert not g_inx.any()
    assert s1["device_rerun_queries"] - s0["device_rerun_queries"] >= len(crowd) + len(ties)
    assert s1["exhaustive_reruns"] - s0["exhaustive_reruns"] >= len(ties)
    for j in sorted(set(crowd + ties + list(range(0, nq, 11)))):
        e_ids, e_dist = oracle.vs_topk(rows, ids, qs[j], k)
        assert int(g_cnt[j]) == e_ids.size, j
        assert g_ids[j][:e_ids.size].view(np.uint32).tolist() == e_ids.tolist(), j
        assert g_dist[j][:e_ids.size].view(np.uint32).tolist() == e_dist.view(np.uint32).tolist(), j
    # the old contract on request: unproven queries only reported
    monkeypatch.setenv("MSI_VS_DEVICE_RERUN", "0")
    o_inx2, o_inx2_p = dv.put(np.zeros(nq, np.uint32))
    check(lib().msi_vs_search_device(st._h, q_p, nq, k, None, 0, o_ids_p, o_dist_p, o_cnt_p, o_inx2_p))
    ctx.synchronize()
    flagged = np.nonzero(np.asarray(dv.get(o_inx2)))[0].tolist()
    assert set(ties) <= set(flagged)
