#!/usr/bin/env python
"""bench.py — hybrid-search hot path on MI355X (BASELINE.json metric:
queries/sec + p50 latency, 10M-doc index, 768-d hybrid / 2-typo).

One "step" = one batch of Q hybrid queries through the GPU hot path:
  * Q query vectors -> exact cosine top-k over the N x d f32 store (vs_scan,
    16 queries per HBM sweep), results rescored with the reference arithmetic;
  * Q x words_per_query query words -> typo derivations (1/2 typos by char
    count, 30 % prefix) over the D-term dictionary (dict_match);
  * results copied to host.
Inputs (store, dictionary, query batch) are resident in HBM before the timed
region.  NOT in the step: the keyword ranking-rule bucket sort and the final
hybrid merge (not on the device yet — see DESIGN.md "out of scope this round").

N GPUs: one process per GPU (torch.distributed / RCCL).  Default sharding is the
north_star's: the query stream is sharded, every rank holds a replica of the
index ("scaling": "weak"), per-rank top-k lists are all-gathered over xGMI.
"""
import argparse
import json
import os
import statistics
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--rows", type=int, default=10_000_000)
    ap.add_argument("--dim", type=int, default=768)
    ap.add_argument("--k", type=int, default=20)
    ap.add_argument("--queries", type=int, default=96, help="hybrid queries per step per GPU")
    ap.add_argument("--words-per-query", type=int, default=2)
    ap.add_argument("--dict-words", type=int, default=2_000_000)
    ap.add_argument("--storage", choices=["f32", "bf16"], default="f32",
                    help="row storage in HBM (f32 = the reference's; bf16 = BASELINE.json config 5's build-side choice)")
    ap.add_argument("--shard", choices=["queries", "rows"], default="queries",
                    help="queries: replicas, each rank answers its own batch (weak scaling, the default); "
                         "rows: every rank holds rows/world of the store and scans it for the SAME batch, "
                         "all-gather of per-shard top-k + device merge (strong scaling)")
    ap.add_argument("--no-typo", action="store_true")
    ap.add_argument("--no-rank", action="store_true",
                    help="leave the keyword ranking (Words->Typo bucket sort) and the hybrid merge out of the step")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample-rows", type=int, default=200_000)
    ap.add_argument("--cpu-sample-words", type=int, default=512)
    return ap.parse_args()


def main():
    args = parse_args()
    import torch

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    assert world == args.gpus or world == 1, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)

    import meilisearch_amd as ma
    from meilisearch_amd import synth

    ctx = ma.Context(local_rank)
    n, d, k, Q = args.rows, args.dim, args.k, args.queries
    n_total = n
    row_sharded = args.shard == "rows" and world > 1
    if row_sharded:
        from meilisearch_amd.distributed import row_range
        r0, r1 = row_range(n_total, rank, world)
        n = r1 - r0          # this rank's shard; docids stay global

    # ---- vector store: rows ~ N(0,1)^d, seed 1234, generated in HBM -----------
    t_setup = time.time()
    gen = torch.Generator(device=dev)
    gen.manual_seed(1234 + (rank if row_sharded else 0))   # distinct rows per shard, the same store per replica
    rows_t = torch.empty((n, d), dtype=torch.float32, device=dev)
    chunk = 1_000_000
    for c0 in range(0, n, chunk):
        c1 = min(n, c0 + chunk)
        rows_t[c0:c1].normal_(generator=gen)
    ids_t = torch.arange(n, dtype=torch.int32, device=dev)
    if row_sharded:
        ids_t += r0                    # docids stay global: shard `rank` holds [r0, r1)
    torch.cuda.synchronize()
    store = ma.GpuStore(ctx, d, storage=args.storage)
    store.upload_device(ids_t, rows_t)
    cpu_rows = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu_rows = rows_t[:min(n, args.cpu_sample_rows)].cpu().numpy()
    del rows_t
    torch.cuda.empty_cache()

    # query vectors (seed 5678 + rank; the same batch on every rank when rows are sharded)
    gq = torch.Generator(device=dev)
    gq.manual_seed(5678 + (0 if row_sharded else rank))
    q_t = torch.empty((Q, d), dtype=torch.float32, device=dev).normal_(generator=gq)
    out_ids = torch.zeros((Q, k), dtype=torch.int32, device=dev)
    out_dist = torch.zeros((Q, k), dtype=torch.float32, device=dev)
    out_cnt = torch.zeros(Q, dtype=torch.int32, device=dev)
    inexact = torch.zeros(Q, dtype=torch.int32, device=dev)

    # ---- dictionary + query words --------------------------------------------
    gdict = None
    n_words_q = 0
    if not args.no_typo:
        words = synth.make_dictionary(args.dict_words, seed=99)
        concat, off = synth.flatten_words(words)
        gdict = ma.GpuDictionary(ctx, concat=concat, offsets=off)
        tq = synth.make_typo_queries(words, Q * args.words_per_query, seed=7 + rank)
        n_words_q = len(tq)
        qb, qoff, qfl = ma.pack_queries(tq)
        qb_t = torch.from_numpy(qb).to(dev)
        qoff_t = torch.from_numpy(qoff.astype(np.int32)).to(dev)
        qfl_t = torch.from_numpy(qfl).to(dev)
        one_t = torch.zeros((n_words_q, 150), dtype=torch.int32, device=dev)
        two_t = torch.zeros((n_words_q, 50), dtype=torch.int32, device=dev)
        one_c = torch.zeros(n_words_q, dtype=torch.int32, device=dev)
        two_c = torch.zeros(n_words_q, dtype=torch.int32, device=dev)
    # ---- keyword ranking leg: Words -> Typo bucket sort over dense posting sets ---------
    # Every query has `words_per_query` terms (+ their 2-gram node); a term offers the documents
    # matching it with 0 / 1 / 2 typos.  18 seeded random posting sets (densities 1 % / 0.2 % /
    # 0.05 % of the documents, 0.01 % for n-grams) are shared by the queries in different
    # combinations; all sets are resident in HBM before the timed region, like the store.
    rank_batch = None
    if not args.no_rank:
        from meilisearch_amd import ranking as R
        n_docs_rank = n_total if row_sharded else n
        wpq = max(1, min(args.words_per_query, 3))
        n_sets = 18
        pool = ma.BitsPool(ctx, n_docs_rank, 1 + n_sets + 4 * Q)
        pool.fill(0, True)
        words64 = (n_docs_rank + 63) // 64
        rng_r = np.random.default_rng(4242)
        dens = [0.01, 0.002, 0.0005] * 5 + [0.0001] * 3
        gbits = torch.Generator(device=dev)
        gbits.manual_seed(4242)
        for si in range(n_sets):
            # 64 Bernoulli(p) bits per word: OR of sparse random positions is cheap to make on the device
            bits = (torch.rand((words64, 64), device=dev, generator=gbits) < dens[si])
            weights = (2 ** torch.arange(0, 63, device=dev, dtype=torch.int64))
            w = (bits[:, :63].to(torch.int64) * weights).sum(dim=1)
            w = torch.where(bits[:, 63], w | torch.tensor(-2 ** 63, device=dev, dtype=torch.int64), w)
            pool.set_from_words(1 + si, w.cpu().numpy().view(np.uint64))
            del bits, w
        rqueries = []
        for qi in range(Q):
            nodes = []
            for ti in range(wpq):
                base = 1 + 3 * int(rng_r.integers(0, 5))
                nodes.append((ti, ti, base, base + 1, base + 2, 2 if rng_r.random() < 0.5 else 1))
                if ti >= 1:
                    nodes.append((ti - 1, ti, 1 + 15 + int(rng_r.integers(0, 3)), None, None, 1))
            rqueries.append((nodes, wpq, 0, 1 + n_sets + 4 * qi))
        rank_batch = R.RankBatch(pool, rqueries)
    torch.cuda.synchronize()
    setup_s = time.time() - t_setup

    gather_buf = gather_ids = None
    if world > 1:
        gather_buf = torch.zeros((world, Q, k), dtype=torch.float32, device=dev)
        gather_ids = torch.zeros((world, Q, k), dtype=torch.int32, device=dev)
        gather_cnt = torch.zeros((world, Q), dtype=torch.int32, device=dev)
        m_ids = torch.zeros((Q, k), dtype=torch.int32, device=dev)
        m_dist = torch.zeros((Q, k), dtype=torch.float32, device=dev)
        m_cnt = torch.zeros(Q, dtype=torch.int32, device=dev)

    n_terms_arr = np.full(Q, max(1, min(args.words_per_query, 3)), dtype=np.uint32)

    def keyword_and_merge(res):
        """Keyword leg: one batched Words->Typo bucket sort for the Q queries, then the hybrid
        merge (ScoreWithRatioResult::merge, semanticRatio 0.5) of every query's two lists."""
        if rank_batch is None:
            return res
        rank_batch.run(R.TERMS_LAST, True, 0, k)
        merged = ma.scoring.hybrid_merge_batch(res[0].numpy().view(np.uint32), res[1].numpy(),
                                               res[2].numpy().view(np.uint32), rank_batch.ids, rank_batch.words,
                                               rank_batch.typos, rank_batch.maxt, rank_batch.counts, n_terms_arr,
                                               0.5, 0, k)
        return res + (merged,)

    def step():
        store.search_device(q_t, k, out_ids, out_dist, out_cnt, inexact)  # ceil(Q / max_batch) HBM sweeps
        if gdict is not None:
            gdict.lookup_device(qb_t, qoff_t, qfl_t, n_words_q, one_t, one_c, two_t, two_c)
        ctx.synchronize()
        if row_sharded:
            import torch.distributed as dist
            from meilisearch_amd.distributed import merge_topk_device
            # the one exchange step: per-shard top-k (Q*k*8 B per rank) over xGMI, then the
            # k-way merge kernel; every rank ends up with the global top-k
            dist.all_gather_into_tensor(gather_buf, out_dist)
            dist.all_gather_into_tensor(gather_ids, out_ids)
            dist.all_gather_into_tensor(gather_cnt, out_cnt)
            torch.cuda.current_stream().synchronize()
            merge_topk_device(ctx, gather_ids, gather_buf, gather_cnt, m_ids, m_dist, m_cnt)
            ctx.synchronize()
            res = (m_ids.cpu(), m_dist.cpu(), m_cnt.cpu())
            if gdict is not None:
                res += (one_c.cpu(), two_c.cpu(), one_t.cpu(), two_t.cpu())
            return keyword_and_merge(res)
        # results to the host (what the Rust caller receives)
        res = (out_ids.cpu(), out_dist.cpu(), out_cnt.cpu())
        if gdict is not None:
            res += (one_c.cpu(), two_c.cpu(), one_t.cpu(), two_t.cpu())
        res = keyword_and_merge(res)
        if world > 1:
            import torch.distributed as dist
            # the one exchange step: per-rank top-k lists (Q*k*8 bytes) over xGMI (RCCL)
            dist.all_gather_into_tensor(gather_buf, out_dist)
            dist.all_gather_into_tensor(gather_ids, out_ids)
            torch.cuda.current_stream().synchronize()
        return res

    for _ in range(args.warmup):
        step()
    ctx.set_profiling(True)
    store.scan_time()
    if gdict is not None:
        gdict.match_time()

    def sync_all():
        ctx.synchronize()
        torch.cuda.synchronize()
        if world > 1:
            import torch.distributed as dist
            dist.barrier()

    sync_all()
    lat = []
    t0 = time.perf_counter()
    for _ in range(args.steps):
        s0 = time.perf_counter()
        res = step()
        lat.append((time.perf_counter() - s0) * 1e3)
    sync_all()
    elapsed = time.perf_counter() - t0
    if world > 1:
        import torch.distributed as dist
        tt = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())

    scan_n, scan_ms = store.scan_time()
    match_n, match_ms = gdict.match_time() if gdict is not None else (0, 0.0)
    stats = store.stats()
    n_inexact = int(inexact.sum().item())

    if world > 1:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()
    if rank != 0:
        return
    total_queries = Q * (1 if row_sharded else world) * args.steps
    qps = total_queries / elapsed
    algo_bytes = ((n + 15) // 16) * stats["bytes_per_tile"]  # one sweep of the tiled store
    scan_avg_ms = scan_ms / max(1, scan_n)
    achieved = algo_bytes / (scan_avg_ms * 1e-3) / 1e9 if scan_n else 0.0
    # HBM traffic of the dominant kernel from the PMC pass committed under profiles/
    # (rocprofv3 --pmc FETCH_SIZE on this same command; counters cannot be read in-process)
    traffic, traffic_src = None, None
    pmc_path = os.path.join(ROOT, "profiles", "r1_pmc_fetch.json")
    if os.path.exists(pmc_path) and n == 10_000_000 and d == 768 and args.storage == "f32":
        try:
            kernels = json.load(open(pmc_path))["kernels"]
            for name, e in kernels.items():
                if name.startswith("vs_scan_kernel") and "false" in name.split(",")[2] and "FETCH_SIZE" in e:
                    traffic = round(e["hbm_read_bytes_per_launch_avg"] / 1e9, 3)
                    traffic_src = f"profiles/r1_pmc_fetch.json ({name}; FETCH_SIZE KB x 1024 x 2, gfx950 correction)"
        except Exception:
            traffic = None
    out = {
        "metric": "hybrid-search hot path queries/sec (10M-doc index, 768-d cosine top-20 + 2-typo term lookup)",
        "value": round(qps, 2),
        "unit": "queries/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(elapsed / args.steps * 1e3, 4),
        "p50_latency_ms": round(statistics.median(lat), 4),
        "higher_is_better": True,
        "scaling": "strong" if row_sharded else "weak",
        "vs_baseline": None,
        "dtype": "f32" if args.storage == "f32" else "bf16 rows, f32 arithmetic",
        "data": "synthetic (rows N(0,1) seed 1234; dictionary seed 99; query words seed 7; BASELINE.md C4/C3)",
        "config": {
            "workload": f"C4 on one GPU per rank: {n} docs x {d}-d {args.storage} exact cosine top-{k} "
                        f"+ {args.words_per_query} typo-tolerant words/query over a {args.dict_words}-term dictionary",
            "queries_per_step_per_gpu": Q,
            "words_per_step_per_gpu": n_words_q,
            "queries_per_hbm_sweep": store.max_batch,
            "scan_math": os.environ.get("MSI_VS_SCAN_MATH", "bf16x3") + " candidate scan (f32 rows in HBM, f32 accumulate)"
                         " + exact f32 reference rescoring of K' candidates with an exactness proof",
            "sharding": ("rows sharded (%d per GPU of %d), same query batch on every GPU, all_gather of per-shard "
                         "top-k (RCCL) + device k-way merge" % (n, n_total)) if row_sharded else
                        "queries sharded, index replicated per GPU, all_gather of per-rank top-k (RCCL)",
            "step_includes": ["vs_scan + select + reference rescoring", "dict_match + cap logic", "D2H of results"]
                             + ([] if args.no_rank else ["Words->Typo bucket sort (batched, %d terms + n-grams per query)"
                                                         % args.words_per_query, "hybrid merge (semanticRatio 0.5)"]),
            "step_excludes": ["ranking-rule bucket sort", "hybrid merge"] if args.no_rank else
                             ["ranking rules after Typo (reference CPU path)"],
            "inexact_queries_last_step": n_inexact,
            "setup_seconds": round(setup_s, 1),
        },
        "roofline": {
            "kernel": "vs_scan_kernel (main pass)",
            "bound": "hbm",
            "achieved": round(achieved, 1),
            "peak": 8000.0,
            "unit": "GB/s",
            "frac": round(achieved / 8000.0, 4),
            "traffic": traffic,
            "traffic_unit": "GB per launch (HBM reads, PMC)",
            "traffic_source": traffic_src,
            "algorithmic_bytes_per_launch": algo_bytes,
            "avg_launch_ms": round(scan_avg_ms, 4),
            "launches_timed": scan_n,
        },
        "dict_match": {"launches_timed": match_n, "avg_launch_ms": round(match_ms / max(1, match_n), 4),
                       "words_per_launch": n_words_q},
    }
    if cpu_rows is not None:
        out["cpu_baseline"] = cpu_baseline(args, cpu_rows, n, d, k,
                                           words if gdict is not None else None,
                                           concat if gdict is not None else None,
                                           off if gdict is not None else None)
    print(json.dumps(out))


def cpu_baseline(args, cpu_rows, n, d, k, words, concat, off):
    """The CPU restatement (oracle/msi_cpubase.c, kind "port": milli cannot be built
    here) timed on this box's host cores on a bounded sample of the same workload."""
    from meilisearch_amd import synth
    from oracle import cpubase
    cores = cpubase.host_threads()
    sample = cpu_rows.shape[0]
    scan = cpubase.CpuVectorScan(cpu_rows, np.arange(sample, dtype=np.uint32))
    q = synth.make_embeddings(16, d, seed=5678)
    scan.search(q[:2], k, threads=cores)  # warm
    t0 = time.perf_counter()
    scan.search(q, k, threads=cores)
    t_vec_sample = (time.perf_counter() - t0) / 16.0
    t_vec = t_vec_sample * (n / sample)  # exact scan is linear in N
    t_word = 0.0
    if words is not None:
        cdict = cpubase.CpuDictionary(concat, off)
        tq = synth.make_typo_queries(words, args.cpu_sample_words, seed=7)
        qb, qoff, qfl = cpubase.pack_queries(tq)
        cdict.lookup_packed(qb[:], qoff[:9], qfl[:8], threads=cores)  # warm
        t0 = time.perf_counter()
        cdict.lookup_packed(qb, qoff, qfl, threads=cores)
        t_word = (time.perf_counter() - t0) / len(tq)
    per_query = t_vec + args.words_per_query * t_word
    return {
        "value": round(1.0 / per_query, 3),
        "unit": "queries/s",
        "cores": cores,
        "kind": "port",
        "sample": f"vector: 16 queries x {sample} rows x {d}-d (all {cores} threads), scaled x{n / sample:.0f} to {n} rows; "
                  f"typo: {args.cpu_sample_words} words over the full {args.dict_words}-term dictionary",
        "vector_queries_per_s": round(1.0 / t_vec, 3),
        "typo_words_per_s": round(1.0 / t_word, 1) if t_word else None,
    }


if __name__ == "__main__":
    main()
