#!/usr/bin/env python
"""Per-config measurements of BASELINE.md §3 (C2: vector scan, C3: typo lookup)
on one MI355X.  bench.py stays the driver's contract (the C4 line); this tool
prints one JSON line per (config, batch size) so the numbers quoted in DESIGN.md
and profiles/ can be regenerated:

    python tools/bench_configs.py c2 [--rows 1000000 --dim 384]
    python tools/bench_configs.py c3 [--dict-words 2000000]

Inputs are resident in HBM before the timed region; results stay on the device
(the D2H of B*k*8 bytes is included in bench.py, not here).
"""
import argparse
import json
import os
import statistics
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def timed(fn, sync, reps, warm=2):
    for _ in range(warm):
        fn()
    sync()
    lat = []
    t0 = time.perf_counter()
    for _ in range(reps):
        s0 = time.perf_counter()
        fn()
        sync()
        lat.append((time.perf_counter() - s0) * 1e3)
    return (time.perf_counter() - t0) / reps * 1e3, statistics.median(lat)


def run_c2(args):
    import torch
    import meilisearch_amd as ma
    dev = torch.device("cuda", 0)
    ctx = ma.Context(0)
    n, d, k = args.rows, args.dim, args.k
    gen = torch.Generator(device=dev)
    gen.manual_seed(1234)
    rows = torch.empty((n, d), dtype=torch.float32, device=dev).normal_(generator=gen)
    ids = torch.arange(n, dtype=torch.int32, device=dev)
    store = ma.GpuStore(ctx, d, storage=args.storage)
    store.upload_device(ids, rows)
    del rows
    gq = torch.Generator(device=dev)
    gq.manual_seed(5678)
    q = torch.empty((1024, d), dtype=torch.float32, device=dev).normal_(generator=gq)
    bytes_per_sweep = ((n + 15) // 16) * store.stats()["bytes_per_tile"]
    for B in args.batches:
        out_ids = torch.zeros((B, k), dtype=torch.int32, device=dev)
        out_dist = torch.zeros((B, k), dtype=torch.float32, device=dev)
        out_cnt = torch.zeros(B, dtype=torch.int32, device=dev)
        inexact = torch.zeros(B, dtype=torch.int32, device=dev)

        def step():
            store.search_device(q[:B], k, out_ids, out_dist, out_cnt, inexact)

        ctx.set_profiling(False)
        ms, p50 = timed(step, ctx.synchronize, args.reps)
        ctx.set_profiling(True)
        store.scan_time()
        step()
        ctx.synchronize()
        ln, lms = store.scan_time()
        scan_ms = lms / max(1, ln)
        # the same batch through the HOST entry point (H2D of the queries, D2H of the results,
        # stream synchronisations): the PCIe-inclusive rate
        qh = q[:B].cpu().numpy()
        host_ms, _ = timed(lambda: store.search(qh, k), lambda: None, max(3, args.reps // 2))
        print(json.dumps({
            "config": "C2", "storage": args.storage, "host_api_ms_per_batch": round(host_ms, 4),
            "host_api_qps": round(B / host_ms * 1e3, 1), "rows": n, "dim": d, "k": k, "batch": B,
            "ms_per_batch": round(ms, 4), "p50_ms": round(p50, 4), "qps": round(B / ms * 1e3, 1),
            "sweeps_per_batch": (B + store.max_batch - 1) // store.max_batch, "scan_kernel_ms": round(scan_ms, 4),
            "scan_GBps": round(bytes_per_sweep / (scan_ms * 1e-3) / 1e9, 1),
            "scan_frac_of_8TBps": round(bytes_per_sweep / (scan_ms * 1e-3) / 8e12, 4),
            "inexact": int(inexact.sum().item())}), flush=True)


def run_c3(args):
    import torch
    import meilisearch_amd as ma
    from meilisearch_amd import synth
    from oracle import cpubase  # the CPU baseline leg only
    dev = torch.device("cuda", 0)
    ctx = ma.Context(0)
    words = synth.make_dictionary(args.dict_words, seed=99)
    concat, off = synth.flatten_words(words)
    gdict = ma.GpuDictionary(ctx, concat=concat, offsets=off)
    all_q = synth.make_typo_queries(words, max(args.batches), seed=7)
    cpu = None
    if not args.no_cpu:
        cores = cpubase.host_threads()
        cdict = cpubase.CpuDictionary(concat, off)
        sample = all_q[:args.cpu_sample]
        qb, qoff, qfl = cpubase.pack_queries(sample)
        cdict.lookup_packed(qb, qoff[:9], qfl[:8], threads=cores)
        t0 = time.perf_counter()
        cdict.lookup_packed(qb, qoff, qfl, threads=cores)
        t_all = time.perf_counter() - t0
        t0 = time.perf_counter()
        qb1, qoff1, qfl1 = cpubase.pack_queries(sample[:64])
        cdict.lookup_packed(qb1, qoff1, qfl1, threads=1)
        t_one = (time.perf_counter() - t0) / 64
        cpu = {"cores": cores, "words_per_s_all_cores": round(len(sample) / t_all, 1),
               "words_per_s_one_core": round(1.0 / t_one, 1), "sample_words": len(sample)}
    for B in args.batches:
        tq = all_q[:B]
        qb, qoff, qfl = ma.pack_queries(tq)
        qb_t = torch.from_numpy(qb).to(dev)
        qoff_t = torch.from_numpy(qoff.astype(np.int32)).to(dev)
        qfl_t = torch.from_numpy(qfl).to(dev)
        one_t = torch.zeros((B, 150), dtype=torch.int32, device=dev)
        two_t = torch.zeros((B, 50), dtype=torch.int32, device=dev)
        one_c = torch.zeros(B, dtype=torch.int32, device=dev)
        two_c = torch.zeros(B, dtype=torch.int32, device=dev)

        def step():
            gdict.lookup_device(qb_t, qoff_t, qfl_t, B, one_t, one_c, two_t, two_c)

        ctx.set_profiling(False)
        p0 = gdict.stats()["pairs_scanned"]
        step()
        ctx.synchronize()
        pairs = gdict.stats()["pairs_scanned"] - p0
        ms, p50 = timed(step, ctx.synchronize, args.reps)
        ctx.set_profiling(True)
        gdict.match_time()
        step()
        ctx.synchronize()
        ln, lms = gdict.match_time()
        out = {"config": "C3", "dict_words": len(words), "batch": B, "ms_per_batch": round(ms, 4),
               "p50_ms": round(p50, 4), "words_per_s": round(B / ms * 1e3, 1),
               "match_kernel_ms": round(lms / max(1, ln), 4), "dp_pairs_per_batch": int(pairs),
               "dict_pairs_per_s": round(B * len(words) / ms * 1e3, 1),
               "hits_one": int(one_c.sum().item()), "hits_two": int(two_c.sum().item())}
        if cpu:
            out["cpu_baseline"] = cpu
            out["x_cpu_all_cores"] = round(out["words_per_s"] / cpu["words_per_s_all_cores"], 2)
        print(json.dumps(out), flush=True)


def run_bq(args):
    """exact Hamming k-NN over a binary-quantised store (`rows` x `dim`, sign bits as bit planes in HBM):
    milliseconds per batch, queries/s, and the two rooflines of the one sweep a batch of <= 32 queries makes — HBM
    (rows x (dim/8 + 4) bytes once per sweep) and VALU (4 operations per 64 bits per query: v_xor + v_bcnt on each half;
    1024 SIMDs x 16 lanes x 2.4 GHz = 39.3 T lane-operations/s)."""
    import torch
    import meilisearch_amd as ma
    from meilisearch_amd import synth
    dev = torch.device("cuda", 0)
    ctx = ma.Context(0)
    n, d, k = args.rows, args.dim, args.k
    rows = synth.device_rows(n, d, dev, seed=1234)
    ids = torch.arange(n, dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    st = ma.GpuBqStore(ctx, d)
    st.upload_device(ids, rows)
    del rows
    q = synth.device_queries(64, d, dev, seed=5678).cpu().numpy()
    W = (d + 63) // 64
    for B in (1, 8, 32, 64):
        ms, p50 = timed(lambda: st.search(q[:B], k), lambda: None, args.reps)
        sweeps = (B + 31) // 32
        by = sweeps * n * (W * 8 + 4)
        ops = n * W * 4 * B
        print(json.dumps({"config": "bq", "rows": n, "dim": d, "k": k, "batch": B, "ms_per_batch": round(ms, 4), "p50_ms": round(p50, 4),
                          "qps": round(B / ms * 1e3, 1), "store_MB": round(n * W * 8 / 1e6, 1),
                          "algorithmic_GBps": round(by / ms / 1e6, 1), "frac_of_8TBps": round(by / ms / 1e6 / 8000, 4),
                          "valu_Tops": round(ops / ms / 1e9, 2), "frac_of_39.3_Tops": round(ops / ms / 1e9 / 39.3, 4)}), flush=True)


def run_filtered(args):
    """Filtered vector search (the C5 shape on one GPU's shard, f32): random candidate
    bitsets of 10 % / 1 % / 0.1 % of the documents, resident in HBM (msi_bits slot), 48
    queries per sweep; only tiles that hold an allowed row are streamed."""
    import torch
    import meilisearch_amd as ma
    dev = torch.device("cuda", 0)
    ctx = ma.Context(0)
    n, d, k = args.rows, args.dim, args.k
    gen = torch.Generator(device=dev)
    gen.manual_seed(1234)
    rows = torch.empty((n, d), dtype=torch.float32, device=dev).normal_(generator=gen)
    ids = torch.arange(n, dtype=torch.int32, device=dev)
    store = ma.GpuStore(ctx, d)
    store.upload_device(ids, rows)
    del rows
    B = store.max_batch
    gq = torch.Generator(device=dev)
    gq.manual_seed(5678)
    q = torch.empty((B, d), dtype=torch.float32, device=dev).normal_(generator=gq)
    out_ids = torch.zeros((B, k), dtype=torch.int32, device=dev)
    out_dist = torch.zeros((B, k), dtype=torch.float32, device=dev)
    out_cnt = torch.zeros(B, dtype=torch.int32, device=dev)
    inexact = torch.zeros(B, dtype=torch.int32, device=dev)
    pool = ma.BitsPool(ctx, n, 1)
    rng = np.random.default_rng(31)
    bytes_per_sweep = ((n + 15) // 16) * store.stats()["bytes_per_tile"]
    for sel in (1.0, 0.1, 0.01, 0.001):
        words = (n + 63) // 64
        bits = rng.random(words * 64) < sel
        pool.set_from_words(0, np.packbits(bits, bitorder="little").view(np.uint64))
        fptr = pool.device_ptr(0)

        def step():
            store.search_device(q, k, out_ids, out_dist, out_cnt, inexact, filter_ptr=fptr, filter_nbits=n)

        ms, p50 = timed(step, ctx.synchronize, args.reps)
        ctx.set_profiling(True)
        store.scan_time()
        step()
        ctx.synchronize()
        ln, lms = store.scan_time()
        ctx.set_profiling(False)
        print(json.dumps({"config": "filtered", "rows": n, "dim": d, "k": k, "batch": B, "selectivity": sel,
                          "ms_per_batch": round(ms, 4), "qps": round(B / ms * 1e3, 1),
                          "full_sweep_kernel_ms": round(lms / max(1, ln), 4),
                          "store_GB": round(bytes_per_sweep / 1e9, 2),
                          "min_results": int(out_cnt.min().item()), "inexact": int(inexact.sum().item())}), flush=True)


def run_c5(args):
    """One GPU's shard of C5 (BASELINE.json configs[4]: 100 M x 1024 bf16 over 8 GPUs = 12.5 M rows
    per GPU): filtered vector search (random 10 % / 1 % / 0.1 % candidate bitsets resident in HBM)
    with k = 1000, then the ranking-rule rerank: Words -> Typo bucket sort of a 3-term keyword
    query restricted to each query's top-1000 (universe = those docids), top 20 returned."""
    import torch
    import meilisearch_amd as ma
    from meilisearch_amd import ranking as R
    dev = torch.device("cuda", 0)
    ctx = ma.Context(0)
    n, d, k = args.rows, args.dim, 1000
    store = ma.GpuStore(ctx, d, "bf16")
    gen = torch.Generator(device=dev)
    gen.manual_seed(1234)
    rows = torch.empty((n, d), dtype=torch.float32, device=dev).normal_(generator=gen)
    ids = torch.arange(n, dtype=torch.int32, device=dev)
    store.upload_device(ids, rows)
    del rows
    torch.cuda.empty_cache()
    B = store.max_batch
    gq = torch.Generator(device=dev)
    gq.manual_seed(5678)
    q = torch.empty((B, d), dtype=torch.float32, device=dev).normal_(generator=gq)
    out_ids = torch.zeros((B, k), dtype=torch.int32, device=dev)
    out_dist = torch.zeros((B, k), dtype=torch.float32, device=dev)
    out_cnt = torch.zeros(B, dtype=torch.int32, device=dev)
    inexact = torch.zeros(B, dtype=torch.int32, device=dev)
    nt = 3
    FILTER, POST, UNI = 0, 1, 1 + 3 * nt
    pool = ma.BitsPool(ctx, n, UNI + 5 * B)
    rng = np.random.default_rng(31)
    words = (n + 63) // 64
    dens = [(0.30, 0.05, 0.02), (0.10, 0.02, 0.01), (0.02, 0.005, 0.002)]
    terms, slot = [], POST
    for i in range(nt):
        sl = []
        for p in dens[i % 3]:
            bits = rng.random(words * 64) < p
            pool.set_from_words(slot, np.packbits(bits, bitorder="little").view(np.uint64))
            sl.append(slot)
            slot += 1
        terms.append((sl[0], sl[1], sl[2], 2 if i % 2 else 1))
    nodes = [(i, i, t[0], t[1], t[2], t[3]) for i, t in enumerate(terms)]
    batch = R.RankBatch(pool, [(nodes, nt, UNI + 5 * i, UNI + 5 * i + 1) for i in range(B)])
    bytes_per_sweep = ((n + 15) // 16) * store.stats()["bytes_per_tile"]
    for sel in (0.1, 0.01, 0.001):
        bits = rng.random(words * 64) < sel
        pool.set_from_words(FILTER, np.packbits(bits, bitorder="little").view(np.uint64))
        fptr = pool.device_ptr(FILTER)

        def scan():
            store.search_device(q, k, out_ids, out_dist, out_cnt, inexact, filter_ptr=fptr, filter_nbits=n)

        def step():
            scan()
            # the top-1000 of every query become the rerank universes on the device (same stream: no sync, no copy)
            pool.set_from_docid_lists_device(UNI, 5, out_ids, out_cnt)
            return batch.run(R.TERMS_LAST, True, 0, 20)

        ms_scan, _ = timed(scan, ctx.synchronize, args.reps)
        ms, p50 = timed(step, ctx.synchronize, args.reps)
        res = step()
        ctx.set_profiling(True)
        store.scan_time()
        scan()
        ctx.synchronize()
        ln, lms = store.scan_time()
        ctx.set_profiling(False)
        print(json.dumps({"config": "c5_shard", "rows": n, "dim": d, "storage": "bf16", "k": k, "batch": B,
                          "selectivity": sel, "scan_ms_per_batch": round(ms_scan, 4),
                          "scan_qps": round(B / ms_scan * 1e3, 1),
                          "ms_per_batch_with_rerank": round(ms, 4), "p50_ms": round(p50, 4),
                          "qps_with_rerank": round(B / ms * 1e3, 1),
                          "scan_kernel_ms": round(lms / max(1, ln), 4), "store_GB": round(bytes_per_sweep / 1e9, 2),
                          "min_results": int(out_cnt.min().item()), "inexact": int(inexact.sum().item()),
                          "rerank_returned_min": int(res.counts.min()),
                          "rerank_candidates_mean": float(np.mean(res.cand))}), flush=True)


def run_ranked(args):
    """The keyword leg with the default criteria [words, typo, proximity, attributeRank, sort, wordPosition,
    exactness] (msi_keyword_search_ranked) over a synthetic index of `rows` documents: host rule graphs,
    every docid set in HBM.  Postings come through the Python vtable adapter (cached bytes), so the host
    share includes ctypes callback overhead a Rust shim would not have."""
    import torch  # noqa: F401  (one HIP runtime per process)
    import meilisearch_amd as ma
    from meilisearch_amd import ranking as R
    from meilisearch_amd import synth
    ctx = ma.Context(0)
    n = args.rows
    words = synth.make_dictionary(args.dict_words, seed=99)
    index = synth.SynthIndex(n, words)
    gdict = ma.GpuDictionary(ctx, [w.encode() for w in index.words])
    pool = ma.BitsPool(ctx, n, args.slots)
    cb = R.IndexCallbacks(index)
    rng = np.random.default_rng(11)
    by_rank = sorted((w for w in index.words if w.isascii() and w.isalpha() and 4 <= len(w) <= 9),
                     key=lambda w: index.rank[w])
    criteria = ["words", "typo", "proximity", "attributeRank", "sort", "wordPosition", "exactness"]
    for nt in (1, 2, 3, 5):
        lat, hits_n, cands, stats = [], [], [], []
        for q in range(args.reps):
            ws = [by_rank[int(rng.integers(0, 300))] for _ in range(nt)]
            terms = [([w], False, i, i, i == nt - 1) for i, w in enumerate(ws)]

            def go():
                return R.keyword_search_ranked(gdict, pool, cb, terms, criteria, strategy=R.TERMS_LAST, limit=20,
                                               searchable_fids=index.searchable_fids,
                                               searchable_weights=[index.weights[f] for f in index.searchable_fids],
                                               max_weight=index.max_weight)
            go()                       # fills the posting cache of the synthetic index
            t0 = time.perf_counter()
            hits, cand = go()
            lat.append((time.perf_counter() - t0) * 1e3)
            stats.append(R.search_last_stats())
            hits_n.append(len(hits))
            cands.append(cand)
        print(json.dumps({"config": "ranked", "docs": n, "terms": nt, "criteria": criteria, "queries": args.reps,
                          "p50_ms": round(statistics.median(lat), 3), "mean_ms": round(statistics.mean(lat), 3),
                          "max_ms": round(max(lat), 3), "hits_min": min(hits_n),
                          "candidates_mean": float(np.mean(cands)), "set_bytes": ((n + 127) // 128) * 16,
                          "mean_stats": {k: round(float(np.mean([s_[k] for s_ in stats])), 1) for k in stats[0]}}),
              flush=True)


    # serving throughput: one caller thread per in-flight search, one pool (private stream) per thread
    import threading
    nt = 3
    queries = []
    for q in range(64):
        ws = [by_rank[int(rng.integers(0, 300))] for _ in range(nt)]
        queries.append([([w], False, i, i, i == nt - 1) for i, w in enumerate(ws)])

    def run_one(pool_, cb_, terms):
        return R.keyword_search_ranked(gdict, pool_, cb_, terms, criteria, strategy=R.TERMS_LAST, limit=20,
                                       searchable_fids=index.searchable_fids,
                                       searchable_weights=[index.weights[f] for f in index.searchable_fids],
                                       max_weight=index.max_weight)
    for t in queries:
        run_one(pool, cb, t)                   # fills the posting cache of the synthetic index
    for n_threads in (1, 2, 4, 8, 16):
        pools = [ma.BitsPool(ctx, n, args.slots, private_stream=True) for _ in range(n_threads)]
        cbs = [R.IndexCallbacks(index) for _ in range(n_threads)]
        nxt, lock, lats = [0], threading.Lock(), []

        def worker(k):
            while True:
                with lock:
                    i = nxt[0]
                    nxt[0] += 1
                if i >= len(queries) * 2:
                    return
                t0 = time.perf_counter()
                run_one(pools[k], cbs[k], queries[i % len(queries)])
                lats.append((time.perf_counter() - t0) * 1e3)
        t0 = time.perf_counter()
        ths = [threading.Thread(target=worker, args=(k,)) for k in range(n_threads)]
        for th in ths:
            th.start()
        for th in ths:
            th.join()
        dt = time.perf_counter() - t0
        print(json.dumps({"config": "ranked_throughput", "docs": n, "terms": nt, "threads": n_threads,
                          "queries": len(lats), "queries_per_s": round(len(lats) / dt, 1),
                          "p50_ms": round(statistics.median(lats), 3), "max_ms": round(max(lats), 3)}), flush=True)
        del pools


def run_rank(args):
    """Words -> Typo bucket sort over dense docid sets (S3): n_terms query terms with
    random zero/one/two-typo posting sets over `rows` documents, top-`k`."""
    import torch  # noqa: F401  (one HIP runtime per process)
    import meilisearch_amd as ma
    from meilisearch_amd import ranking as R
    ctx = ma.Context(0)
    n, nt = args.rows, args.terms
    rng = np.random.default_rng(77)
    pool = ma.BitsPool(ctx, n, 3 * nt + 2)
    words = (n + 63) // 64
    dens = [(0.30, 0.05, 0.02), (0.10, 0.02, 0.01), (0.02, 0.005, 0.002)]
    terms, slot = [], 2
    for i in range(nt):
        sl = []
        for p in dens[i % 3]:
            bits = rng.random(words * 64) < p
            pool.set_from_words(slot, np.packbits(bits, bitorder="little").view(np.uint64))
            sl.append(slot)
            slot += 1
        terms.append((sl[0], sl[1], sl[2], 2 if i % 2 else 1))
    pool.fill(0, True)
    for strategy, name in ((R.TERMS_LAST, "last"), (R.TERMS_ALL, "all")):
        def step():
            return R.bucket_sort_words_typo(pool, terms, 0, 1, strategy, True, 0, args.k)
        ms, p50 = timed(step, ctx.synchronize, args.reps)
        got, cand = step()
        sets_bytes = (3 * nt + 1) * words * 8
        print(json.dumps({"config": "rank", "docs": n, "terms": nt, "strategy": name, "k": args.k,
                          "ms_per_query": round(ms, 4), "p50_ms": round(p50, 4), "candidates": cand,
                          "returned": len(got), "algorithmic_bytes_per_pass": sets_bytes,
                          "first": got[:3]}), flush=True)


def run_rules(args):
    """Per-bucket cost of the rules that read per-document arrays: Sort (order keys), distinct
    (single- and multi-valued CSR) and GeoSort over `rows` documents — milliseconds per call and algorithmic
    GB/s (DESIGN §4.8-4.10 give the byte counts).  No torch needed."""
    import meilisearch_amd as ma
    ctx = ma.Context(0)
    n = args.rows
    rng = np.random.default_rng(11)
    pool = ma.BitsPool(ctx, n, 8)
    words = (n + 63) // 64

    def set_density(slot, p):
        bits = rng.random(words * 64) < p
        bits[n:] = False
        pool.set_from_words(slot, np.packbits(bits, bitorder="little").view(np.uint64))
    out = []
    # Sort: 1000 distinct keys, a tenth without a value
    keys = rng.integers(0, 1000, n).astype(np.uint32)
    keys[rng.random(n) < 0.1] = 0xFFFFFFFF
    dk = ma.DocKeys(ctx, keys)
    for dens in (1.0, 0.1, 0.001):
        def step():
            set_density(0, dens)
            return pool.order_next(dk, 0, 1)
        set_density(0, dens)
        t = []
        for _ in range(args.reps):
            set_density(0, dens)
            ctx.synchronize()
            t0 = time.perf_counter()
            pool.order_next(dk, 0, 1)
            t.append((time.perf_counter() - t0) * 1e3)
        ms = statistics.median(t)
        by = 2 * (n / 8 + 4 * n * dens)
        out.append({"rule": "sort", "universe_density": dens, "p50_ms_per_bucket": round(ms, 4),
                    "algorithmic_GBps": round(by / ms / 1e6, 1)})
    # distinct: single-valued (n / 4 values) and multi-valued (0..3 of n / 3 values)
    single = [[int(v)] for v in rng.integers(0, max(1, n // 4), n)]
    multi_counts = rng.integers(0, 4, n)
    multi_vals = rng.integers(0, max(1, n // 3), int(multi_counts.sum()))
    multi, o = [], 0
    for c in multi_counts.tolist():
        multi.append(sorted(set(multi_vals[o:o + c].tolist())))
        o += c
    for name, per_doc, nv in (("single", single, max(1, n // 4)), ("multi", multi, max(1, n // 3))):
        dv = ma.DocValues(ctx, per_doc, nv)
        for dens in (1.0, 0.01):
            t, rounds = [], 0
            for _ in range(args.reps):
                set_density(0, dens)
                ctx.synchronize()
                t0 = time.perf_counter()
                _, rounds, seq = pool.distinct(dv, 0, 1, 2)
                ctx.synchronize()
                t.append((time.perf_counter() - t0) * 1e3)
            out.append({"rule": "distinct", "values": name, "candidate_density": dens, "rounds": rounds, "sequential": seq,
                        "p50_ms_per_application": round(statistics.median(t), 4)})
        dv.close()
    # GeoSort: uniform points, a fifth without
    pts = np.column_stack([rng.uniform(-90, 90, n), rng.uniform(-180, 180, n)])
    pts[rng.random(n) < 0.2] = np.nan
    gp = ma.GeoPoints(ctx, pts)
    for dens in (1.0, 0.01):
        t = []
        for _ in range(args.reps):
            set_density(0, dens)
            ctx.synchronize()
            t0 = time.perf_counter()
            pool.geo_next(gp, 0, 1, 2, 48.85, 2.35)
            t.append((time.perf_counter() - t0) * 1e3)
        ms = statistics.median(t)
        out.append({"rule": "geoSort", "universe_density": dens, "p50_ms_per_bucket": round(ms, 4),
                    "algorithmic_GBps": round(2 * (n / 8 + 16 * n * dens * 0.8) / ms / 1e6, 1)})
    for o_ in out:
        print(json.dumps(dict(config="rules", docs=n, **o_)), flush=True)


def run_update(args):
    """msi_vs_update against a full re-upload: `rows` x `dim`, 1 % of the rows replaced + 1 % removed +
    1 % added per commit.  Reports milliseconds per commit, the bytes that crossed PCIe and the HBM rate of the
    re-gather (2 x store bytes)."""
    import meilisearch_amd as ma
    ctx = ma.Context(0)
    n, d = args.rows, args.dim
    rng = np.random.default_rng(5)
    rows = rng.standard_normal((n, d)).astype(np.float32)
    ids = (np.arange(n, dtype=np.uint32) * 2)
    st = ma.GpuStore(ctx, d, storage=args.storage)
    t0 = time.perf_counter()
    st.upload(ids, rows)
    ctx.synchronize()
    full_ms = (time.perf_counter() - t0) * 1e3
    m = max(1, n // 100)
    t = []
    for rep in range(args.reps):
        rm = np.sort(rng.choice(ids, m, replace=False))
        repl = np.sort(rng.choice(ids, m, replace=False))
        new = np.sort(rng.choice(np.arange(1, 2 * n, 2, dtype=np.uint32), m, replace=False))
        ad = np.union1d(repl, new).astype(np.uint32)
        ad_rows = rng.standard_normal((ad.size, d)).astype(np.float32)
        ctx.synchronize()
        t0 = time.perf_counter()
        st.update(rm, ad, ad_rows)
        ctx.synchronize()
        t.append((time.perf_counter() - t0) * 1e3)
    ms = statistics.median(t)
    elem = 2 if args.storage == "bf16" else 4
    store_bytes = len(st) * d * elem
    print(json.dumps({"config": "update", "rows": n, "dim": d, "storage": args.storage, "full_upload_ms": round(full_ms, 2),
                      "update_p50_ms": round(ms, 3), "pcie_bytes_per_update": int(ad.size * d * 4 + 4 * (rm.size + ad.size + len(st))),
                      "pcie_bytes_full_upload": int(n * d * 4 + 4 * n),
                      "regather_GBps": round(2 * store_bytes / ms / 1e6, 1), "rows_after": len(st)}), flush=True)


def run_group(args):
    """One process driving every visible GPU (msi_group_create: ncclCommInitAll, one context and one host thread per
    device inside msi_vs_group_search) — the form a single meilisearch process would use.  Both modes at C2's shape:
    REPLICATE (every GPU holds the store, the query batch is split) and SHARD_ROWS (every GPU holds a row range, one packed
    all-gather of the per-shard top-k + device merge).  Host entry point: queries and results cross PCIe.  A parity
    check against the oracle on 8 queries precedes the timing.  On a one-GPU box this is a world of one."""
    import ctypes as C
    import torch
    import meilisearch_amd as ma
    from meilisearch_amd import _lib, synth
    from oracle import oracle as O
    L = _lib.lib()
    n_dev = args.devices or torch.cuda.device_count()
    n, d, k = args.rows, args.dim, args.k
    rows = synth.make_embeddings(n, d, seed=1234)
    ids = np.arange(n, dtype=np.uint32)
    ptr = lambda a: a.ctypes.data_as(C.c_void_p)
    devs = (C.c_int32 * n_dev)(*range(n_dev))
    g = C.c_void_p()
    _lib.check(L.msi_group_create(devs, n_dev, C.byref(g)))
    for mode, name in ((0, "replicate"), (1, "shard_rows")):
        vs = C.c_void_p()
        _lib.check(L.msi_vs_group_create(g, d, 0, mode, C.byref(vs)))
        _lib.check(L.msi_vs_group_upload(vs, ptr(ids), ptr(rows), n))
        q = synth.make_embeddings(max(args.batches), d, seed=5678)
        out_d = np.zeros((max(args.batches), k), np.uint32)
        out_s = np.zeros((max(args.batches), k), np.float32)
        cnt = np.zeros(max(args.batches), np.uint32)
        _lib.check(L.msi_vs_group_search(vs, ptr(q), 8, k, ptr(out_d), ptr(out_s), ptr(cnt)))
        bad = 0
        for j in range(8):
            e_ids, e_dist = O.vs_topk(rows, ids, q[j], k)
            bad += int(out_d[j].tolist() != e_ids.tolist() or out_s[j].view(np.uint32).tolist() != e_dist.view(np.uint32).tolist())
        for B in args.batches:
            ms, p50 = timed(lambda: _lib.check(L.msi_vs_group_search(vs, ptr(q), B, k, ptr(out_d), ptr(out_s), ptr(cnt))),
                            lambda: None, args.reps)
            print(json.dumps({"config": "group", "mode": name, "devices": n_dev, "rows": n, "dim": d, "k": k, "batch": B,
                              "ms_per_batch": round(ms, 4), "p50_ms": round(p50, 4), "qps": round(B / ms * 1e3, 1),
                              "includes": "H2D of the queries, the searches on every device, the exchange, D2H of the results",
                              "parity": {"checked_queries": 8, "mismatches": bad,
                                         "checker": "oracle/msi_oracle.c orc_vs_topk: docids in order, f32 distances bit-identical"}}),
                  flush=True)
        L.msi_vs_group_destroy(vs)
    L.msi_group_destroy(g)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("config", choices=["c2", "c3", "rank", "filtered", "c5", "ranked", "rules", "update", "bq", "group"])
    ap.add_argument("--terms", type=int, default=3)
    ap.add_argument("--storage", choices=["f32", "bf16"], default="f32")
    ap.add_argument("--rows", type=int, default=1_000_000)
    ap.add_argument("--dim", type=int, default=384)
    ap.add_argument("--k", type=int, default=20)
    ap.add_argument("--dict-words", type=int, default=2_000_000)
    ap.add_argument("--batches", type=int, nargs="*", default=None)
    ap.add_argument("--reps", type=int, default=10)
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--cpu-sample", type=int, default=2048)
    ap.add_argument("--slots", type=int, default=1024)
    ap.add_argument("--devices", type=int, default=0, help="group: GPUs to drive from this process (default: all visible)")
    args = ap.parse_args()
    if args.batches is None:
        args.batches = [1, 16, 48, 240] if args.config != "c3" else [1, 64, 1024, 8192]
    {"c2": run_c2, "c3": run_c3, "rank": run_rank, "filtered": run_filtered, "c5": run_c5, "ranked": run_ranked,
     "rules": run_rules, "update": run_update, "bq": run_bq, "group": run_group}[args.config](args)


if __name__ == "__main__":
    main()
